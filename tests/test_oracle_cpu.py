"""CPU checks: the oracle restatement reproduces the committed golden vectors (generated from the
reference's own sources by oracle/gen_golden.py), and the host-side schema logic is consistent."""
import os

import numpy as np
import pytest
import torch

from oracle import pspnet_ref
from peanut_amd import weights as W


def test_flop_table_matches_baseline():
    cfg = W.PredCfg()
    # BASELINE.md sec. 3 (GFLOP/map)
    assert abs(W.conv_flops_per_map(cfg, 240, 240) / 1e9 - 78.526) < 1e-3
    assert abs(W.conv_flops_per_map(cfg, 480, 480) / 1e9 - 313.788) < 1e-3
    assert abs(W.conv_flops_per_map(cfg, 960, 960) / 1e9 - 1254.837) < 1e-3
    assert abs(W.conv_flops_per_map(W.PredCfg(in_channels=13), 480, 480) / 1e9 - 313.755) < 1e-3
    assert abs(W.conv_flops_per_map(W.PredCfg(in_channels=25), 960, 960) / 1e9 - 1256.297) < 1e-3


def test_state_dict_schema():
    cfg = W.PredCfg()
    sd = W.make_seeded_state_dict(cfg, 0, with_aux=True)
    assert len(sd) == 370
    n_params = sum(v.numel() for k, v in sd.items()
                   if not k.endswith(("running_mean", "running_var", "num_batches_tracked")))
    assert n_params == 48968652
    tensors = W.select_inference_tensors(sd, cfg)
    assert [k for k, _ in tensors] == [k for k, _ in W.inference_keys(cfg)]
    assert not any(k.startswith("auxiliary_head") for k, _ in tensors)
    sd2 = W.make_seeded_state_dict(cfg, 0)
    assert all(torch.equal(sd[k], sd2[k]) for k in sd2)       # seed-reproducible
    bad = dict(sd)
    del bad["backbone.layer3.2.conv2.weight"]
    with pytest.raises(KeyError):
        W.select_inference_tensors(bad, cfg)
    bad = dict(sd)
    bad["decode_head.conv_seg.weight"] = torch.zeros(7, 512, 1, 1)
    with pytest.raises(ValueError):
        W.select_inference_tensors(bad, cfg)


def test_mmcv_checkpoint_roundtrip(tmp_path):
    """Loader accepts the mmcv layout: {'meta': {'CLASSES':...}, 'state_dict': ...}, optional
    'module.' prefix, extra auxiliary_head.* keys (inference.py:33-35)."""
    cfg = W.PredCfg()
    sd = W.make_seeded_state_dict(cfg, 3, with_aux=True)
    ck = {"meta": {"CLASSES": ("a", "b", "c", "d", "e", "f")},
          "state_dict": {"module." + k: v for k, v in sd.items()}}
    p = tmp_path / "ckpt.pth"
    torch.save(ck, p)
    sd2, meta = W.load_mmcv_checkpoint(str(p))
    assert meta["CLASSES"][0] == "a"
    assert set(sd2) == set(sd)
    assert len(W.select_inference_tensors(sd2, cfg)) == len(W.inference_keys(cfg))


def test_cfg_file_parsing(tmp_path):
    p = tmp_path / "cfg.py"
    p.write_text(
        "norm_cfg = dict(type='BN', requires_grad=True)\n"
        "model = dict(type='EncoderDecoder', backbone=dict(type='ResNetV1c', depth=50, num_stages=4,"
        " dilations=(1, 1, 2, 4), strides=(1, 2, 1, 1), contract_dilation=True, in_channels=14),"
        " decode_head=dict(type='PSPHead', in_channels=2048, in_index=3, channels=512,"
        " pool_scales=(1, 2, 3, 6), num_classes=6, align_corners=False), test_cfg=dict(mode='whole'))\n")
    cfg = W.pred_cfg_from_file(str(p))
    assert cfg == W.PredCfg()
    p.write_text(p.read_text().replace("mode='whole'", "mode='slide'"))
    with pytest.raises(ValueError):
        W.pred_cfg_from_file(str(p))


def test_oracle_reproduces_golden(golden_dir):
    z = np.load(os.path.join(golden_dir, "pspnet_golden.npz"))
    cases = sorted({k.split("/")[0] for k in z.files})
    assert {"cfg1_240", "odd_100", "rect_72x104", "cin25_64"} <= set(cases)
    torch.set_num_threads(min(8, os.cpu_count() or 1))
    for case in cases:
        if case == "cfg1_240" and os.environ.get("PEANUT_FAST_TESTS"):
            continue
        cfg = W.PredCfg(in_channels=int(z[f"{case}/c_in"]))
        sd = W.make_seeded_state_dict(cfg, int(z[f"{case}/weight_seed"]))
        x = torch.from_numpy(z[f"{case}/input"].astype(np.float32))
        got = pspnet_ref.forward_batch(sd, x, cfg).numpy()
        # same code, same machine class: bit-exact here; 1e-5 leaves room for other CPUs' kernels
        assert np.abs(got - z[f"{case}/logits"]).max() <= 1e-5, case


def test_oracle_reproduces_the_four_map_480_golden(golden_dir):
    """tests/golden/pspnet_b4_480_golden.npz (oracle/gen_golden.py --round3b): four 480 x 480 maps of the benchmark's recipe
    through the reference's own model files, logits at rows 1::4 / cols 2::4 -- the oracle on ONE of them (the CPU suite
    stays short), and the input recipe itself (the fixture stores the seed, not the maps)."""
    from bench import synth_maps
    z = np.load(os.path.join(golden_dir, "pspnet_b4_480_golden.npz"))
    x = synth_maps(4, 14, 480, "cpu", seed0=int(z["b4_480/input_seed"]))
    assert float(x.double().sum()) == float(z["b4_480/input_sum"]), "bench.synth_maps changed: regenerate the fixture"
    if os.environ.get("PEANUT_FAST_TESTS"):
        return
    cfg = W.PredCfg()
    sd = W.make_seeded_state_dict(cfg, 0)
    torch.set_num_threads(min(8, os.cpu_count() or 1))
    got = pspnet_ref.forward_batch(sd, x[2:3], cfg).numpy()[:, :, 1::4, 2::4]
    assert np.abs(got - z["b4_480/logits32_sub"][2:3]).max() <= 1e-5
    # ... the agent's deployed 720 x 720 window
    x7 = synth_maps(1, 14, 720, "cpu", seed0=int(z["win_720/input_seed"]))
    assert float(x7.double().sum()) == float(z["win_720/input_sum"])
    got7 = pspnet_ref.forward_batch(sd, x7, cfg).numpy()[:, :, 1::4, 2::4]
    assert np.abs(got7 - z["win_720/logits32_sub"]).max() <= 1e-5
    # ... and the config-5-size map of the same file (960 x 960, 25 channels, weights of seed 1)
    cfg5 = W.PredCfg(in_channels=int(z["cfg5_960/c_in"]))
    sd5 = W.make_seeded_state_dict(cfg5, int(z["cfg5_960/weight_seed"]))
    x5 = synth_maps(1, 25, 960, "cpu", seed0=int(z["cfg5_960/input_seed"]))
    assert float(x5.double().sum()) == float(z["cfg5_960/input_sum"])
    got5 = pspnet_ref.forward_batch(sd5, x5, cfg5).numpy()[:, :, 1::4, 2::4]
    assert np.abs(got5 - z["cfg5_960/logits32_sub"]).max() <= 2e-5


def test_oracle_distance_to_fp64_reference(golden_dir):
    """The fp64 golden logits (the reference's model files run in float64, oracle/gen_golden.py) put a number on
    'fp32-class': the oracle -- bit-identical to the reference's fp32 CPU path -- is within 1e-5 of them."""
    import numpy as np
    import torch
    from oracle import pspnet_ref
    from peanut_amd.weights import PredCfg, make_seeded_state_dict
    z = np.load(os.path.join(golden_dir, "pspnet_golden.npz"))
    z64 = np.load(os.path.join(golden_dir, "pspnet_fp64_golden.npz"))
    cfg = PredCfg()
    sd = make_seeded_state_dict(cfg, 0)
    x = torch.from_numpy(z["odd_100/input"].astype(np.float32))
    got = pspnet_ref.forward_batch(sd, x, cfg).numpy().astype(np.float64)
    err = float(np.abs(got - z64["odd_100/logits64"]).max())
    assert err <= 1e-5 and abs(err - float(z64["odd_100/fp32_cpu_reference_max_abs"])) <= 1e-9


def _variant_cfg(z, name):
    over = {}
    for k in z.files:
        if k.startswith(f"{name}/cfg_"):
            v = z[k]
            over[k.split("cfg_", 1)[1]] = tuple(int(t) for t in v) if v.ndim else (bool(v) if v.dtype == bool else int(v))
    return W.PredCfg(**over)


@pytest.mark.parametrize("name", ["align_corners", "pool124_k9_c20", "os16_no_contract"])
def test_oracle_reproduces_variant_config_goldens(golden_dir, name):
    """nav/pred_model_cfg.py with some fields edited (align_corners = True; pool_scales (1, 2, 4) with 9 classes and 20
    input channels; an output-stride-16 backbone without contracted dilation): logits from the reference's own model files
    (oracle/gen_golden.py: gen_pspnet_variants) vs the restatement."""
    z = np.load(os.path.join(golden_dir, "pspnet_golden_variants.npz"))
    cfg = _variant_cfg(z, name)
    sd = W.make_seeded_state_dict(cfg, int(z[f"{name}/weight_seed"]))
    x = torch.from_numpy(z[f"{name}/input"].astype(np.float32))
    got = pspnet_ref.forward_batch(sd, x, cfg).numpy()
    assert got.shape == z[f"{name}/logits"].shape
    assert np.abs(got - z[f"{name}/logits"]).max() <= 1e-5
