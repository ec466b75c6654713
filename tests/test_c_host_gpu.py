"""The C ABI used from a host with no Python and no torch in it: tests/c_abi/pred_host.cpp is compiled with hipcc
against include/peanut_hip.h + libpeanut_hip.so, fed a state dict and an input through plain files, and must
reproduce the Python path's output bit for bit (same library, same stream semantics, caller-owned hipMalloc buffers)."""
import os
import shutil
import struct
import subprocess
from types import SimpleNamespace

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _write_state_dict(path, tensors):
    with open(path, "wb") as f:
        f.write(struct.pack("<i", len(tensors)))
        for name, a in tensors:
            nb = name.encode()
            shape = list(a.shape) + [0] * (4 - a.ndim)
            f.write(struct.pack("<i", len(nb)) + nb + struct.pack("<i", a.ndim) + struct.pack("<4q", *shape))
            f.write(np.ascontiguousarray(a, np.float32).tobytes())


@pytest.mark.parametrize("precision", ["fp32", "bf16x6", "fp16x3"])
def test_cpp_host_reproduces_python_path(tmp_path, precision):
    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    if not os.path.exists(hipcc):
        pytest.skip("hipcc not available on this box")
    from peanut_amd import _lib
    from peanut_amd.prediction import PEANUT_Prediction_Model
    from peanut_amd.weights import PredCfg, make_seeded_state_dict, select_inference_tensors
    lib_dir = os.path.join(ROOT, "peanut_amd")
    exe = str(tmp_path / "pred_host")
    r = subprocess.run([hipcc, "--offload-arch=gfx950", "-O2", "-std=c++17", "-I", os.path.join(ROOT, "include"),
                        os.path.join(ROOT, "tests", "c_abi", "pred_host.cpp"), "-L", lib_dir, "-lpeanut_hip",
                        f"-Wl,-rpath,{lib_dir}", "-o", exe], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-3000:]
    cfg = PredCfg()
    sd = make_seeded_state_dict(cfg, 0)
    _write_state_dict(tmp_path / "weights.bin", select_inference_tensors(sd, cfg))
    g = torch.Generator().manual_seed(4)
    x = (torch.rand((2, cfg.in_channels, 88, 120), generator=g) > 0.7).float()
    x.numpy().tofile(tmp_path / "input.bin")
    r = subprocess.run([exe, str(tmp_path / "weights.bin"), str(tmp_path / "input.bin"), str(tmp_path / "output.bin"),
                        "2", str(cfg.in_channels), "88", "120", "1", str(_lib.PRECISIONS[precision])],
                       capture_output=True, text=True, timeout=600)
    assert r.returncode == 0 and r.stdout.startswith("ok"), r.stdout + r.stderr
    got = np.fromfile(tmp_path / "output.bin", np.float32).reshape(2, 6, 88, 120)
    m = PEANUT_Prediction_Model(SimpleNamespace(sem_gpu_id=0), state_dict=sd, cfg=cfg, precision=precision)
    want = m.get_prediction_batch(x.cuda(), apply_sigmoid=True).cpu().numpy()
    assert np.array_equal(got, want)
    assert 0.0 < got.min() and got.max() < 1.0


def test_cpp_detector_host_reproduces_python_path(tmp_path):
    """The whole Mask R-CNN (peanut_rcnn_create + peanut_rcnn_inference) from a torch-free C++ host: detections, scores,
    boxes, classes and pasted masks equal the Python path's (same library underneath) bit for bit."""
    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    if not os.path.exists(hipcc):
        pytest.skip("hipcc not available on this box")
    from peanut_amd.rcnn import MaskRCNN
    from peanut_amd.rcnn_weights import RcnnCfg, front_keys, make_seeded_rcnn_state_dict, roi_head_keys
    lib_dir = os.path.join(ROOT, "peanut_amd")
    exe = str(tmp_path / "rcnn_host")
    r = subprocess.run([hipcc, "--offload-arch=gfx950", "-O2", "-std=c++17", "-I", os.path.join(ROOT, "include"),
                        os.path.join(ROOT, "tests", "c_abi", "rcnn_host.cpp"), "-L", lib_dir, "-lpeanut_hip",
                        f"-Wl,-rpath,{lib_dir}", "-o", exe], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-3000:]
    cfg = RcnnCfg(depth=50, rpn_pre_nms_topk=300, rpn_post_nms_topk=100, detections_per_image=20, score_thresh_test=0.3)
    sd = make_seeded_rcnn_state_dict(cfg, seed=7)
    keys = [k for k, _ in list(front_keys(cfg)) + list(roi_head_keys(cfg))]
    _write_state_dict(tmp_path / "weights.bin", [(k, sd[k].numpy()) for k in keys])
    g = torch.Generator().manual_seed(11)
    B, H, W = 2, 192, 256
    img = torch.randint(0, 256, (B, H, W, 3), generator=g, dtype=torch.uint8)
    img.numpy().tofile(tmp_path / "image.bin")
    pre = str(tmp_path / "out")
    r = subprocess.run([exe, str(tmp_path / "weights.bin"), str(tmp_path / "image.bin"), pre, str(B), str(H), str(W), "50", "300", "100",
                        "20", "0.3"], capture_output=True, text=True, timeout=900)
    assert r.returncode == 0 and r.stdout.startswith("ok"), r.stdout + r.stderr
    counts = np.fromfile(pre + ".counts.bin", np.int32)
    want = MaskRCNN(cfg, sd).inference(img.cuda())
    assert counts.tolist() == [len(w["scores"]) for w in want] and counts.sum() > 0
    n = int(counts.sum())
    boxes = np.fromfile(pre + ".boxes.bin", np.float32).reshape(n, 4)
    scores = np.fromfile(pre + ".scores.bin", np.float32)
    classes = np.fromfile(pre + ".classes.bin", np.int32)
    masks = np.fromfile(pre + ".masks.bin", np.uint8).reshape(n, H, W)
    assert np.array_equal(boxes, torch.cat([w["pred_boxes"] for w in want]).cpu().numpy())
    assert np.array_equal(scores, torch.cat([w["scores"] for w in want]).cpu().numpy())
    assert np.array_equal(classes, torch.cat([w["pred_classes"] for w in want]).cpu().numpy().astype(np.int32))
    assert np.array_equal(masks.astype(bool), torch.cat([w["pred_masks"] for w in want]).cpu().numpy())
