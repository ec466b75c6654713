"""ORACLE support (test infrastructure): generate the committed golden vectors under tests/golden/
by running THE REFERENCE'S OWN SOURCES from /root/reference (see oracle/ref_import.py), and assert
that the oracle restatements (oracle/pspnet_ref.py, oracle/mapping_ref.py) reproduce them.

Run in the build container only (needs /root/reference):   python -m oracle.gen_golden
Nothing here travels to or runs on the GPU box; only the .npz outputs do.
"""
from __future__ import annotations

import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

from oracle import pspnet_ref, ref_import  # noqa: E402
from peanut_amd.weights import PredCfg, make_seeded_state_dict  # noqa: E402

GOLDEN = os.path.join(ROOT, "tests", "golden")

# (case name, c_in, weight seed, B, H, W, input seed)
PSP_CASES = [
    ("cfg1_240", 14, 0, 1, 240, 240, 0),      # BASELINE.json configs[0]
    ("b2_96", 14, 0, 2, 96, 96, 1),
    ("odd_100", 14, 0, 1, 100, 100, 2),       # ceil-mode arithmetic, non-integer bilinear scale
    ("rect_72x104", 14, 0, 2, 72, 104, 3),
    ("cin25_64", 25, 1, 1, 64, 64, 4),        # config 5's channel count, different weights
]


def psp_input(b, c, h, w, seed):
    """SURVEY.md sec. 8d config-1 style: binary sparse map, x = (rand > 0.7)."""
    g = torch.Generator().manual_seed(seed)
    return (torch.rand((b, c, h, w), generator=g) > 0.7).float()


def gen_pspnet(report):
    out = {}
    models = {}
    for name, c_in, wseed, b, h, w, iseed in PSP_CASES:
        cfg = PredCfg(in_channels=c_in)
        key = (c_in, wseed)
        if key not in models:
            m = ref_import.build_reference_model(in_channels=c_in)
            sd = make_seeded_state_dict(cfg, wseed, with_aux=True)
            m.load_state_dict(sd, strict=True)
            models[key] = (m, sd)
        m, sd = models[key]
        x = psp_input(b, c_in, h, w, iseed)
        t0 = time.time()
        ref = np.stack(ref_import.reference_forward(m, x))          # list of [6,H,W] -> [B,6,H,W]
        dt = time.time() - t0
        mine = pspnet_ref.forward_batch(sd, x, cfg).numpy()
        err = float(np.abs(ref - mine).max())
        assert err <= 1e-5, f"{name}: oracle restatement deviates from the reference by {err}"
        # single-map API path (run_inference / get_prediction) for the first map
        one = pspnet_ref.run_inference(sd, x[0].numpy(), cfg)[0]
        assert float(np.abs(one - ref_import.reference_forward(m, x[:1])[0]).max()) <= 1e-5
        out[f"{name}/input"] = x.numpy().astype(np.uint8)
        out[f"{name}/logits"] = ref.astype(np.float32)
        out[f"{name}/c_in"] = np.int64(c_in)
        out[f"{name}/weight_seed"] = np.int64(wseed)
        report["pspnet"][name] = dict(shape=[b, c_in, h, w], restatement_max_abs=err,
                                      logits_absmax=float(np.abs(ref).max()), ref_seconds=round(dt, 3))
        print(f"[pspnet] {name}: ref vs restatement max-abs {err:.2e}, |logit|max {np.abs(ref).max():.2f}")
    np.savez_compressed(os.path.join(GOLDEN, "pspnet_golden.npz"), **out)


def gen_pspnet_fp64(report):
    """The reference model run in float64 (same files, ``model.double()``): the yardstick for 'fp32-class' --
    how far is an implementation from the exact result, compared with the reference's own fp32 CPU path?"""
    out = {}
    cfg = PredCfg(in_channels=14)
    m = ref_import.build_reference_model(in_channels=14)
    m.load_state_dict(make_seeded_state_dict(cfg, 0, with_aux=True), strict=True)
    m = m.double()
    z = np.load(os.path.join(GOLDEN, "pspnet_golden.npz"))
    for name in ("b2_96", "odd_100"):
        x = torch.from_numpy(z[f"{name}/input"]).double()
        ref64 = np.stack(ref_import.reference_forward(m, x)).astype(np.float64)
        err32 = float(np.abs(ref64 - z[f"{name}/logits"].astype(np.float64)).max())
        out[f"{name}/logits64"] = ref64
        out[f"{name}/fp32_cpu_reference_max_abs"] = np.float64(err32)
        report["pspnet_fp64"][name] = dict(fp32_cpu_reference_vs_fp64_max_abs=err32)
        print(f"[pspnet fp64] {name}: the reference's fp32 CPU path is {err32:.2e} from its fp64 self")
    np.savez_compressed(os.path.join(GOLDEN, "pspnet_fp64_golden.npz"), **out)


# nav/pred_model_cfg.py fields beyond their committed values (round 3): (name, PredCfg overrides, B, H, W, weight seed)
PSP_VARIANTS = [
    ("align_corners", dict(align_corners=True), 2, 72, 88, 11),
    ("pool124_k9_c20", dict(pool_scales=(1, 2, 4), num_classes=9, in_channels=20), 1, 96, 96, 12),
    ("os16_no_contract", dict(strides=(1, 2, 2, 1), dilations=(1, 1, 1, 2), contract_dilation=False), 2, 96, 80, 13),
]


def gen_pspnet_variants(report):
    """pspnet_golden_variants.npz: the reference's own model files built from nav/pred_model_cfg.py with some of its fields
    edited (align_corners, pool_scales / num_classes / in_channels, an output-stride-16 backbone), seeded weights and inputs."""
    out = {}
    for name, over, b, h, w, wseed in PSP_VARIANTS:
        cfg = PredCfg(**over)
        bb = {k: over[k] for k in ("strides", "dilations", "contract_dilation") if k in over}
        dh = {k: over[k] for k in ("pool_scales", "num_classes", "align_corners") if k in over}
        m = ref_import.build_reference_model(in_channels=cfg.in_channels, backbone=bb, decode_head=dh)
        sd = make_seeded_state_dict(cfg, wseed, with_aux=True)
        m.load_state_dict(sd, strict=True)
        x = psp_input(b, cfg.in_channels, h, w, wseed)
        ref = np.stack(ref_import.reference_forward(m, x))
        mine = pspnet_ref.forward_batch(sd, x, cfg).numpy()
        err = float(np.abs(ref - mine).max())
        assert err <= 1e-5, f"{name}: oracle restatement deviates from the reference by {err}"
        out[f"{name}/input"] = x.numpy().astype(np.uint8)
        out[f"{name}/logits"] = ref.astype(np.float32)
        out[f"{name}/weight_seed"] = np.int64(wseed)
        for k, v in over.items():
            out[f"{name}/cfg_{k}"] = np.asarray(v)
        report["pspnet"][name] = dict(shape=[b, cfg.in_channels, h, w], restatement_max_abs=err, logits_absmax=float(np.abs(ref).max()))
        print(f"[pspnet] {name}: ref vs restatement max-abs {err:.2e}, |logit|max {np.abs(ref).max():.2f}")
    np.savez_compressed(os.path.join(GOLDEN, "pspnet_golden_variants.npz"), **out)


def gen_pspnet_round2(report):
    """Round-2 additions, written to their own files so that the round-1 fixtures stay byte-identical:
    * pspnet_golden_c25.npz -- config 5's channel count (C_in = 25) at 240x240 (one map);
    * pspnet_fp64_480_golden.npz -- the reference model in float64 on ONE 480x480 map of the benchmark's synthetic
      recipe (bench.synth_maps, seed 4242), logits kept at rows 1::4, cols 2::4 only (the full fp64 tensor is 11 MB),
      together with the same sub-grid of the reference's fp32 CPU path."""
    from bench import synth_maps
    cfg = PredCfg(in_channels=25)
    m = ref_import.build_reference_model(in_channels=25)
    sd = make_seeded_state_dict(cfg, 1, with_aux=True)
    m.load_state_dict(sd, strict=True)
    x = psp_input(1, 25, 240, 240, 5)
    ref = np.stack(ref_import.reference_forward(m, x))
    mine = pspnet_ref.forward_batch(sd, x, cfg).numpy()
    err = float(np.abs(ref - mine).max())
    assert err <= 1e-5, f"cin25_240: oracle restatement deviates from the reference by {err}"
    np.savez_compressed(os.path.join(GOLDEN, "pspnet_golden_c25.npz"), **{
        "cin25_240/input": x.numpy().astype(np.uint8), "cin25_240/logits": ref.astype(np.float32),
        "cin25_240/c_in": np.int64(25), "cin25_240/weight_seed": np.int64(1)})
    report["pspnet"]["cin25_240"] = dict(shape=[1, 25, 240, 240], restatement_max_abs=err, logits_absmax=float(np.abs(ref).max()))
    print(f"[pspnet] cin25_240: ref vs restatement max-abs {err:.2e}")

    cfg = PredCfg(in_channels=14)
    m = ref_import.build_reference_model(in_channels=14)
    m.load_state_dict(make_seeded_state_dict(cfg, 0, with_aux=True), strict=True)
    x = synth_maps(1, 14, 480, "cpu", seed0=4242)
    ref32 = np.stack(ref_import.reference_forward(m, x)).astype(np.float64)
    ref64 = np.stack(ref_import.reference_forward(m.double(), x.double())).astype(np.float64)
    sub = (slice(None), slice(None), slice(1, None, 4), slice(2, None, 4))
    err32 = float(np.abs(ref64 - ref32).max())
    np.savez_compressed(os.path.join(GOLDEN, "pspnet_fp64_480_golden.npz"), **{
        "cfg2_480/input_seed": np.int64(4242), "cfg2_480/logits64_sub": ref64[sub],
        "cfg2_480/logits32_sub": ref32[sub].astype(np.float32), "cfg2_480/input_sum": np.float64(x.double().sum().item()),
        "cfg2_480/fp32_cpu_reference_max_abs": np.float64(err32)})
    report["pspnet_fp64"]["cfg2_480"] = dict(fp32_cpu_reference_vs_fp64_max_abs=err32)
    print(f"[pspnet fp64] cfg2_480: the reference's fp32 CPU path is {err32:.2e} from its fp64 self")


def gen_pspnet_round3b(report):
    """Round-3 (last sessions) addition, its own file: pspnet_b4_480_golden.npz -- FOUR 480x480 maps of the benchmark's
    synthetic recipe (bench.synth_maps, seeds 777..780) through the reference's own model files in fp32, logits kept at rows
    1::4, cols 2::4.  At this batch the planner of the HIP path runs the larger Winograd tiles (F(5x5) in the dilation-4
    layers, F(6x6) in the PSP bottleneck), so those forms are held against reference-generated numbers, not only against
    the oracle."""
    from bench import synth_maps
    cfg = PredCfg(in_channels=14)
    m = ref_import.build_reference_model(in_channels=14)
    sd = make_seeded_state_dict(cfg, 0, with_aux=True)
    m.load_state_dict(sd, strict=True)
    x = synth_maps(4, 14, 480, "cpu", seed0=777)
    ref32 = np.stack(ref_import.reference_forward(m, x)).astype(np.float32)
    mine = pspnet_ref.forward_batch(make_seeded_state_dict(cfg, 0), x, cfg).numpy()
    err = float(np.abs(ref32 - mine).max())
    assert err <= 1e-5, f"b4_480: oracle restatement deviates from the reference by {err}"
    sub = (slice(None), slice(None), slice(1, None, 4), slice(2, None, 4))
    np.savez_compressed(os.path.join(GOLDEN, "pspnet_b4_480_golden.npz"), **{
        "b4_480/input_seed": np.int64(777), "b4_480/logits32_sub": ref32[sub],
        "b4_480/input_sum": np.float64(x.double().sum().item())})
    report["pspnet"]["b4_480"] = dict(shape=[4, 14, 480, 480], restatement_max_abs=err, logits_absmax=float(np.abs(ref32).max()))
    print(f"[pspnet] b4_480: ref vs restatement max-abs {err:.2e}, |logit| max {np.abs(ref32).max():.2f}")

    # config 5 at its real size: ONE 960x960 map with 25 input channels (weights of seed 1, bench.synth_maps seed 53 -- map 3 of
    # the batch test_config5_full_size_properties builds), sub-grid rows 1::4 / cols 2::4, into the same file
    cfg = PredCfg(in_channels=25)
    m = ref_import.build_reference_model(in_channels=25)
    m.load_state_dict(make_seeded_state_dict(cfg, 1, with_aux=True), strict=True)
    x5 = synth_maps(1, 25, 960, "cpu", seed0=53)
    ref5 = np.stack(ref_import.reference_forward(m, x5)).astype(np.float32)
    z = dict(np.load(os.path.join(GOLDEN, "pspnet_b4_480_golden.npz")))
    z.update({"cfg5_960/input_seed": np.int64(53), "cfg5_960/logits32_sub": ref5[sub], "cfg5_960/c_in": np.int64(25),
              "cfg5_960/weight_seed": np.int64(1), "cfg5_960/input_sum": np.float64(x5.double().sum().item())})
    np.savez_compressed(os.path.join(GOLDEN, "pspnet_b4_480_golden.npz"), **z)
    # the agent's deployed prediction window (agent_state.py:345-373: a 720 x 720 crop of the full map, one map per call)
    cfg = PredCfg(in_channels=14)
    m = ref_import.build_reference_model(in_channels=14)
    m.load_state_dict(make_seeded_state_dict(cfg, 0, with_aux=True), strict=True)
    x7 = synth_maps(1, 14, 720, "cpu", seed0=7200)
    ref7 = np.stack(ref_import.reference_forward(m, x7)).astype(np.float32)
    z.update({"win_720/input_seed": np.int64(7200), "win_720/logits32_sub": ref7[sub],
              "win_720/input_sum": np.float64(x7.double().sum().item())})
    np.savez_compressed(os.path.join(GOLDEN, "pspnet_b4_480_golden.npz"), **z)
    report["pspnet"]["win_720"] = dict(shape=[1, 14, 720, 720], logits_absmax=float(np.abs(ref7).max()))
    print(f"[pspnet] win_720: one 720x720 map through the reference, |logit| max {np.abs(ref7).max():.2f}")
    report["pspnet"]["cfg5_960"] = dict(shape=[1, 25, 960, 960], logits_absmax=float(np.abs(ref5).max()))
    print(f"[pspnet] cfg5_960: one 960x960 x 25 map through the reference, |logit| max {np.abs(ref5).max():.2f}")


def gen_pspnet_round6(report):
    """Round-6 addition, its own file: pspnet_b32_480_golden.npz -- the HEADLINE's exact batch: the 32 maps bench.py times at
    N = 1 (bench.synth_maps, seed0 = 0 = rank 0's shard, 480 x 480, 14 channels) through the reference's own model files in fp32
    (4 maps per call: the reference is a batch-independent forward), logits kept at rows 3::8, cols 5::8 (2.8 MB).  At 32 maps
    every layer runs on the kernel the benchmark times it on (layer4 conv1 moves to the 256 x 256 persistent kernel only at this
    batch), so the timed kernel assignment itself is held against reference-generated numbers."""
    from bench import synth_maps
    cfg = PredCfg(in_channels=14)
    m = ref_import.build_reference_model(in_channels=14)
    sd = make_seeded_state_dict(cfg, 0, with_aux=True)
    m.load_state_dict(sd, strict=True)
    x = synth_maps(32, 14, 480, "cpu", seed0=0)
    sub = (slice(None), slice(None), slice(3, None, 8), slice(5, None, 8))
    t0 = time.time()
    parts = [np.stack(ref_import.reference_forward(m, x[i:i + 4])).astype(np.float32)[sub] for i in range(0, 32, 4)]
    ref = np.concatenate(parts, 0)
    dt = time.time() - t0
    mine = pspnet_ref.forward_batch(make_seeded_state_dict(cfg, 0), x[28:32], cfg).numpy()[sub]
    err = float(np.abs(ref[28:32] - mine).max())
    assert err <= 1e-5, f"b32_480: oracle restatement deviates from the reference by {err}"
    np.savez_compressed(os.path.join(GOLDEN, "pspnet_b32_480_golden.npz"), **{
        "b32_480/input_seed": np.int64(0), "b32_480/logits32_sub": ref,
        "b32_480/input_sum": np.float64(x.double().sum().item()),
        "b32_480/input_sum_per_map": x.double().sum(dim=(1, 2, 3)).numpy()})
    report["pspnet"]["b32_480"] = dict(shape=[32, 14, 480, 480], restatement_max_abs_last4=err, logits_absmax=float(np.abs(ref).max()),
                                       ref_seconds=round(dt, 1), sub_grid="rows 3::8, cols 5::8")
    print(f"[pspnet] b32_480: 32 maps through the reference in {dt:.0f} s, ref vs restatement (maps 28-31) {err:.2e}, "
          f"|logit| max {np.abs(ref).max():.2f}")


def main():
    if "--round6" in sys.argv:       # only the round-6 fixture (the headline's own 32 maps)
        torch.set_num_threads(8)
        report = {"pspnet": {}}
        gen_pspnet_round6(report)
        with open(os.path.join(GOLDEN, "golden_report_r6.json"), "w") as f:
            json.dump(report, f, indent=1, sort_keys=True)
        return
    if "--round3b" in sys.argv:      # only the fixture of the last sessions of round 3
        report = {"pspnet": {}}
        gen_pspnet_round3b(report)
        with open(os.path.join(GOLDEN, "golden_report_r3b.json"), "w") as f:
            json.dump(report, f, indent=1, sort_keys=True)
        return
    if "--round3" in sys.argv:       # only the round-3 fixtures (variant configs; mapping flags: oracle.gen_golden_mapping)
        report = {"pspnet": {}, "mapping": {}}
        gen_pspnet_variants(report)
        with open(os.path.join(GOLDEN, "golden_report_r3.json"), "w") as f:
            json.dump(report, f, indent=1, sort_keys=True)
        return
    if "--round2" in sys.argv:       # only the added fixtures (the round-1 files are left untouched)
        report = {"pspnet": {}, "pspnet_fp64": {}}
        gen_pspnet_round2(report)
        with open(os.path.join(GOLDEN, "golden_report_r2.json"), "w") as f:
            json.dump(report, f, indent=1, sort_keys=True)
        return
    assert ref_import.reference_available(), "needs /root/reference"
    os.makedirs(GOLDEN, exist_ok=True)
    torch.set_num_threads(8)
    report = {"pspnet": {}, "pspnet_fp64": {}, "mapping": {}, "torch": torch.__version__}
    gen_pspnet(report)
    gen_pspnet_fp64(report)
    gen_pspnet_round2(report)
    gen_pspnet_variants(report)
    from oracle import gen_golden_agent, gen_golden_mapping
    gen_golden_mapping.generate(report)
    gen_golden_agent.generate(report)
    with open(os.path.join(GOLDEN, "golden_report.json"), "w") as f:
        json.dump(report, f, indent=1, sort_keys=True)


if __name__ == "__main__":
    main()
