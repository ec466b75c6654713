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


@pytest.mark.parametrize("precision", ["fp32", "bf16x6"])
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
