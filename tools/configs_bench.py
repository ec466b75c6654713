#!/usr/bin/env python3
"""Every BASELINE.json configuration besides the headline one, measured the way bench.py measures the headline: a rate,
a `roofline` for the kernel that bounds the stage and a `cpu_baseline` from the matching oracle on the host cores.
bench.py calls `measure_configs()` at N = 1 and prints the result under "configs"; standalone:

    python tools/configs_bench.py [1,3,4,5,mapping]

  1        one 240 x 240, 6-category map, B = 1 (the reference's own CPU-runnable case): latency
  3        Mask R-CNN R-101-FPN on 640 x 480 RGB frames, B = 16 and B = 1 (SemanticPredMaskRCNN.get_prediction)
  4        the per-step pipeline (detector + observation formatting + projection, map prediction + goal selection every
           10th step) on synthetic frames -- this GPU's share of the 8-episode job
  5        960 x 960, 25-channel maps, B = 8 -- this GPU's share of the 64-map job -- + the all-gather of the predictions
  mapping  Semantic_Mapping.forward alone (stage 2)
Inputs are synthetic and resident in HBM, weights seeded random-init (there are no checkpoints / HM3D recordings here).
The oracle/ modules are used as bench.py's cpu_baseline leg uses them: timed on the host, never part of a GPU figure."""
from __future__ import annotations

import json
import os
import sys
import time
from types import SimpleNamespace

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "tools")):
    if p not in sys.path:
        sys.path.insert(0, p)

FP32_PEAK_TFLOPS = 157.3     # MI355X_MICROARCH.md: v_mfma_f32_32x32x2_f32
HBM_PEAK_GBS = 8000.0


def _cpu_threads() -> int:
    return int(os.environ.get("PEANUT_CPU_THREADS", min(16, os.cpu_count() or 1)))


def _pred_roofline(model, x, B, reps=5):
    """dominant kernel family of a prediction forward (executed FLOPs / summed launch time, HIP events on the launch stream)"""
    rows = model.model.profile(x, repeats=reps)
    fam = {}
    for name, kern, ms, fl, by in rows:
        f = fam.setdefault(kern, {"ms": 0.0, "flops": 0.0, "launches": 0})
        f["ms"] += ms
        f["flops"] += fl
        f["launches"] += 1
    k, f = max(((k, f) for k, f in fam.items() if f["flops"] > 0), key=lambda kv: kv[1]["ms"])
    ach = f["flops"] / (f["ms"] * 1e-3) / 1e12
    total_ms = sum(v["ms"] for v in fam.values())
    return {"bound": "mfma", "kernel": k, "achieved": round(ach, 2), "peak": FP32_PEAK_TFLOPS, "unit": "TFLOP/s",
            "frac": round(ach / FP32_PEAK_TFLOPS, 4), "traffic": None, "launches_per_step": f["launches"],
            "avg_launch_ms": round(f["ms"] / f["launches"], 4), "share_of_step_time": round(f["ms"] / total_ms, 3),
            "whole_forward_tflops_executed": round(sum(r[3] for r in rows) / (total_ms * 1e-3) / 1e12, 2),
            "note": "executed FLOPs of the family / its summed launch time, per-op HIP events (peanut_pred_probe_*)"}


def _pred_cpu(cfg, sd, size, runs=3):
    from bench import cpu_model_name, synth_maps
    from oracle import pspnet_ref
    n = _cpu_threads()
    torch.set_num_threads(n)
    x = synth_maps(1, cfg.in_channels, size, "cpu", seed0=10_000)
    pspnet_ref.forward_batch(sd, x, cfg)
    best = float("inf")
    for _ in range(runs):
        t0 = time.perf_counter()
        pspnet_ref.forward_batch(sd, x, cfg)
        best = min(best, time.perf_counter() - t0)
    return {"value": round(1.0 / best, 4), "unit": "maps/s", "cores": n, "kind": "port", "cpu_model": cpu_model_name(),
            "sample": f"oracle/pspnet_ref.py, one [1,{cfg.in_channels},{size},{size}] map, 1 warm-up + best of {runs}"}


def config1(dev):
    from bench import synth_maps
    from peanut_amd.prediction import PEANUT_Prediction_Model
    from peanut_amd.weights import PredCfg, make_seeded_state_dict
    cfg = PredCfg(in_channels=10)                     # 4 + 6 categories
    sd = make_seeded_state_dict(cfg, seed=0)
    m = PEANUT_Prediction_Model(SimpleNamespace(sem_gpu_id=dev.index), state_dict=sd, cfg=cfg)
    x = synth_maps(1, cfg.in_channels, 240, dev, seed0=1)
    out = torch.empty((1, cfg.num_classes, 240, 240), device=dev)
    for _ in range(5):
        m.get_prediction_batch(x, out=out)
    torch.cuda.synchronize()
    reps = 100
    t0 = time.perf_counter()
    for _ in range(reps):
        m.get_prediction_batch(x, out=out)
    torch.cuda.synchronize()
    ms = (time.perf_counter() - t0) / reps * 1e3
    roof = _pred_roofline(m, x, 1)
    cpu = _pred_cpu(cfg, sd, 240)
    return {"workload": "config 1: one 240x240, 6-category (10-channel) partial map -> prediction forward, batch 1",
            "metric": "maps/s (latency-bound: a chain of ~130 dependent launches)", "value": round(1e3 / ms, 1), "unit": "maps/s",
            "ms_per_map": round(ms, 3), "dtype": "f32", "roofline": roof, "cpu_baseline": cpu,
            "speedup_vs_cpu_baseline": round(1e3 / ms / cpu["value"], 1)}


def config5(dev):
    from bench import synth_maps
    from peanut_amd import dist as pdist
    from peanut_amd.prediction import PEANUT_Prediction_Model
    from peanut_amd.weights import PredCfg, make_seeded_state_dict
    cfg = PredCfg(in_channels=25)
    sd = make_seeded_state_dict(cfg, seed=0)
    B, S = 8, 960
    m = PEANUT_Prediction_Model(SimpleNamespace(sem_gpu_id=dev.index), state_dict=sd, cfg=cfg)
    x = synth_maps(B, cfg.in_channels, S, dev, seed0=0)
    out = torch.empty((B, cfg.num_classes, S, S), device=dev)
    for _ in range(2):
        m.get_prediction_batch(x, out=out)
    torch.cuda.synchronize()
    reps = 5
    t0 = time.perf_counter()
    for _ in range(reps):
        m.get_prediction_batch(x, out=out)
    torch.cuda.synchronize()
    ms = (time.perf_counter() - t0) / reps * 1e3
    roof = _pred_roofline(m, x, B, reps=2)
    # the collation of the predictions: with ONE rank (this bench line) peanut_amd.dist.allgather_maps is the identity and RCCL is
    # not involved -- say so instead of timing a no-op under the collective's name; at N > 1 bench.py times the real all-gather
    import torch.distributed as tdist
    world = tdist.get_world_size() if tdist.is_available() and tdist.is_initialized() else 1
    gather_ms, path = None, "identity (1 rank: RCCL not used; the N > 1 line of bench.py times peanut_allgather_maps)"
    if world > 1:
        path = "peanut_allgather_maps (RCCL)"
        try:
            pdist.allgather_maps(out)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            pdist.allgather_maps(out)
            torch.cuda.synchronize()
            gather_ms = (time.perf_counter() - t0) * 1e3
        except Exception as e:   # noqa: BLE001 -- reporting only
            path = f"failed: {e}"
    cpu = _pred_cpu(cfg, sd, S, runs=2)
    return {"workload": "config 5: 960x960, 25-channel (4+21) maps, batch 8 = one GPU's share of the 64-map job",
            "metric": "maps/s", "value": round(B / ms * 1e3, 2), "unit": "maps/s", "ms_per_step": round(ms, 3), "dtype": "f32",
            "roofline": roof, "cpu_baseline": cpu, "speedup_vs_cpu_baseline": round(B / ms * 1e3 / cpu["value"], 1),
            "allgather_maps_ms": None if gather_ms is None else round(gather_ms, 3),
            "allgather_maps_bytes_per_rank": int(out.numel() * 4), "allgather_maps_path": path}


def _rcnn_roofline(net, img):
    rows = net.probe_front(img, reps=3)
    fam = {}
    for name, kern, ms, fl in rows:
        f = fam.setdefault(kern, {"ms": 0.0, "flops": 0.0, "launches": 0})
        f["ms"] += ms
        f["flops"] += fl
        f["launches"] += 1
    total_ms = sum(v["ms"] for v in fam.values())
    # the dominant family among the launches whose executed FLOPs ARE the direct-form count (no Winograd transform in the op)
    k, f = max(((k, f) for k, f in fam.items() if f["flops"] > 0 and not k.startswith("wino+")), key=lambda kv: kv[1]["ms"])
    ach = f["flops"] / (f["ms"] * 1e-3) / 1e12
    wino_ms = sum(v["ms"] for kk, v in fam.items() if kk.startswith("wino+"))
    return {"bound": "mfma", "kernel": k, "achieved": round(ach, 2), "peak": FP32_PEAK_TFLOPS, "unit": "TFLOP/s",
            "frac": round(ach / FP32_PEAK_TFLOPS, 4), "traffic": None, "launches_per_step": f["launches"],
            "avg_launch_ms": round(f["ms"] / f["launches"], 4), "share_of_front_end_time": round(f["ms"] / total_ms, 3),
            "winograd_ops_share_of_front_end_time": round(wino_ms / total_ms, 3), "front_end_ms": round(total_ms, 3),
            "front_end_tflops_nominal": round(sum(r[3] for r in rows) / (total_ms * 1e-3) / 1e12, 2),
            "note": "front end (R-101 + FPN + RPN heads) of a batch, per-op HIP events (peanut_rcnn_probe_front); the family is "
                    "the dominant one among the ops that execute their direct-form FLOPs; Winograd ops (3 launches each) "
                    "are counted by share only"}


def _rcnn_post_roofline(net, img, cfg, reps=3):
    """The back half of stage 1 (proposal selection, ROI heads, mask paste -- segmentation.py:41-62 around
    GeneralizedRCNN.inference) stage by stage: HIP events at the stage boundaries of peanut_rcnn_semantic
    (peanut_rcnn_set_stage_timing); MFMA stages against the fp32 matrix peak on their direct-form FLOPs, HBM stages against 8 TB/s
    on their algorithmic bytes."""
    net.set_stage_timing(True)
    acc = {}
    dets = []
    for _ in range(reps):
        net.semantic(img, cfg.num_classes, 0.5, 0.5, None)
        for name, bound, ms, work in net.stage_times():
            a = acc.setdefault(name, {"bound": bound, "ms": 0.0, "work": work})
            a["ms"] += ms / reps
        dets.append(sum(net.last_detection_counts) / img.shape[0])
    net.set_stage_timing(False)
    B = img.shape[0]
    props = float(net.debug_stage("prop_count", (B,), torch.int32).float().mean())
    stages = {}
    for name, a in acc.items():
        e = {"bound": a["bound"], "ms": round(a["ms"], 4)}
        if name == "front_end" and a["ms"] > 0:
            # the front end's work figure is the NOMINAL direct-form count (its Winograd layers execute ~a third of it): no
            # fraction here -- the front end's roofline is the `roofline` object next to this one (per-op probe)
            e["achieved_nominal"] = round(a["work"] / (a["ms"] * 1e-3) / 1e12, 2)
            e["unit"] = "TFLOP/s (nominal direct-form FLOPs)"
        elif a["bound"] == "mfma" and a["ms"] > 0:
            e["achieved"] = round(a["work"] / (a["ms"] * 1e-3) / 1e12, 2)
            e["unit"], e["peak"] = "TFLOP/s", FP32_PEAK_TFLOPS
            e["frac"] = round(e["achieved"] / FP32_PEAK_TFLOPS, 4)
        elif a["bound"] == "hbm" and a["ms"] > 0:
            e["achieved"] = round(a["work"] / (a["ms"] * 1e-3) / 1e9, 1)
            e["unit"], e["peak"] = "GB/s", HBM_PEAK_GBS
            e["frac"] = round(e["achieved"] / HBM_PEAK_GBS, 4)
        stages[name] = e
    back = [k for k in stages if k != "front_end"]
    back_ms = sum(stages[k]["ms"] for k in back)
    # composite: time-weighted mean of the stages' own fractions (the host read counts as 0: nothing runs on the GPU under it)
    comp = sum(stages[k]["ms"] * stages[k].get("frac", 0.0) for k in back) / back_ms if back_ms > 0 else None
    return {"stages": stages, "back_half_ms": round(back_ms, 3), "front_end_ms": stages.get("front_end", {}).get("ms"),
            "back_half_frac_time_weighted": None if comp is None else round(comp, 4),
            "proposals_per_image": round(props, 1), "detections_per_image": round(sum(dets) / len(dets), 1),
            "note": "per-stage HIP events inside peanut_rcnn_semantic; mfma stages: executed FLOPs (the front end's row: nominal direct-form FLOPs) / time against 157.3 TFLOP/s; hbm "
                    "stages: algorithmic bytes (operands once, result once) / time against 8 TB/s; seeded random weights: the "
                    "detection count (hence the mask head's work) is whatever those weights produce"}


def config3(dev, with_cpu=True):
    from bench import cpu_model_name
    from peanut_amd.rcnn import MaskRCNN
    from peanut_amd.rcnn_weights import RcnnCfg, make_seeded_rcnn_state_dict
    cfg = RcnnCfg(score_thresh_test=0.5)
    sd = make_seeded_rcnn_state_dict(cfg, 0)
    net = MaskRCNN(cfg, sd, device=dev)
    g = torch.Generator().manual_seed(3)
    res = {}
    for B, reps in ((16, 5), (1, 20)):
        img = torch.randint(0, 256, (B, 480, 640, 3), generator=g, dtype=torch.uint8).to(dev)
        for _ in range(2):
            net.semantic(img, cfg.num_classes, 0.5, 0.5, None)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(reps):
            net.semantic(img, cfg.num_classes, 0.5, 0.5, None)
        torch.cuda.synchronize()
        ms = (time.perf_counter() - t0) / reps * 1e3
        res[B] = (ms, _rcnn_roofline(net, img), _rcnn_post_roofline(net, img, cfg))
    cpu = None
    if with_cpu:
        from oracle import rcnn_ref
        n = _cpu_threads()
        torch.set_num_threads(n)
        img1 = torch.randint(0, 256, (1, 480, 640, 3), generator=g, dtype=torch.uint8)
        # one warm-up + best of two inside a ~30 s budget (a run is ~7 s on 16 cores); the runs' spread goes into the line
        runs = []
        t_budget = time.perf_counter()
        for i in range(3):
            t0 = time.perf_counter()
            rcnn_ref.inference(sd, img1, cfg)
            runs.append(time.perf_counter() - t0)
            if i >= 1 and time.perf_counter() - t_budget > 30.0:
                break
        timed = runs[1:] if len(runs) > 1 else runs
        dt = min(timed)
        cpu = {"value": round(1.0 / dt, 4), "unit": "images/s", "cores": n, "kind": "port", "cpu_model": cpu_model_name(),
               "runs_s": [round(r, 3) for r in runs],
               "sample": "oracle/rcnn_ref.py inference (restated detectron2 v0.6 definitions; parity unpinned), ONE 640x480 frame, "
                         f"1 warm-up + best of {len(timed)} (runs_s lists every run, the warm-up first)"}
    ms16, roof16, post16 = res[16]
    ms1, roof1, post1 = res[1]
    out = {"workload": "config 3: Mask R-CNN R-101-FPN (cat9 yaml) inference + per-category mask accumulation "
                       "(SemanticPredMaskRCNN.get_prediction) on 640x480 RGB frames, batch 16",
           "metric": "images/s", "value": round(16 / ms16 * 1e3, 1), "unit": "images/s", "ms_per_batch": round(ms16, 2), "dtype": "f32",
           "roofline": roof16, "post": post16, "cpu_baseline": cpu,
           "proposals_per_image": post16["proposals_per_image"], "detections_per_image": post16["detections_per_image"],
           "batch1": {"ms_per_frame": round(ms1, 3), "images_per_s": round(1e3 / ms1, 1), "roofline": roof1, "post": post1}}
    if cpu:
        out["speedup_vs_cpu_baseline"] = round(out["value"] / cpu["value"], 1)
    return out, net, (sd, cfg)


def mapping_stage(dev):
    from bench import cpu_model_name
    from oracle import mapping_ref, mapping_scenes
    from peanut_amd.mapping import Semantic_Mapping
    args = SimpleNamespace(device=dev, frame_height=120, frame_width=160, map_resolution=5, map_size_cm=4800,
                           global_downscaling=2, vision_range=100, hfov=79.0, du_scale=1, cat_pred_threshold=5.0,
                           exp_pred_threshold=1.0, map_pred_threshold=0.1, num_sem_categories=10, camera_height=0.88)
    sm = Semantic_Mapping(args)
    mcfg = mapping_ref.MapCfg()
    frames = mapping_scenes.make_sequence(5, 16)
    obs_c = [torch.from_numpy(mapping_scenes.frame_to_obs(f))[None] for f in frames]
    rel_c = [torch.from_numpy(f["pose"]) for f in frames]
    obs_g, rel_g = [o.to(dev) for o in obs_c], [r.to(dev) for r in rel_c]
    n = _cpu_threads()
    torch.set_num_threads(n)
    mc, pc = torch.zeros(14, 480, 480), torch.tensor([12.0, 12.0, 0.0])
    mg, pg = torch.zeros(14, 480, 480, device=dev), torch.tensor([12.0, 12.0, 0.0], device=dev)
    worst, t_cpu = 0.0, 0.0
    for o, r, og, rg in zip(obs_c, rel_c, obs_g, rel_g):
        t0 = time.perf_counter()
        _, mc, _, pc = mapping_ref.forward(o, r, mc, pc, mcfg)
        t_cpu += time.perf_counter() - t0
        _, mg, _, _ = sm(og, rg, mg, pg, None)
        worst = max(worst, (mg.cpu() - mc).abs().max().item())
    torch.cuda.synchronize()
    reps = 20
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        for og, rg in zip(obs_g, rel_g):
            _, mg, _, _ = sm(og, rg, mg, pg, None)
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / (reps * len(frames))
    by = 1.07e6 + 2 * 14 * 480 * 480 * 4                      # obs + maps_last in + map_pred out
    gbs = by / (ms * 1e-3) / 1e9
    cpu_ms = 1e3 * t_cpu / len(frames)
    return {"workload": "stage 2: Semantic_Mapping.forward, one 120x160 frame into the 14x480x480 local map (8 launches)",
            "metric": "steps/s", "value": round(1e3 / ms, 1), "unit": "steps/s", "ms_per_step": round(ms, 4), "dtype": "f32",
            "max_abs_vs_oracle": worst,
            "roofline": {"bound": "hbm", "kernel": "mapping step (8 launches of csrc/mapping.hip)", "achieved": round(gbs, 1),
                         "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(gbs / HBM_PEAK_GBS, 4), "traffic": None,
                         "algorithmic_bytes_per_step": round(by),
                         "note": "27 MB algorithmic per step / time of the whole step (events on the launch stream); eight "
                                 "dependent launches of 5-25 us each: launch-latency-bound, far from the HBM roof"},
            "cpu_baseline": {"value": round(1e3 / cpu_ms, 2), "unit": "steps/s", "cores": n, "kind": "port", "cpu_model": cpu_model_name(),
                             "sample": f"oracle/mapping_ref.py (bit-identical to the reference's module here), {len(frames)} chained frames"},
            "speedup_vs_cpu_baseline": round(cpu_ms / ms, 1)}


def _pred_b1(dev, size=720):
    """one size x size map through the prediction forward at batch 1 (the agent's own operating point, agent_state.py:345-373):
    latency and the whole forward's executed FLOPs against the fp32 matrix peak"""
    from bench import synth_maps
    from peanut_amd.prediction import PEANUT_Prediction_Model
    from peanut_amd.weights import PredCfg, make_seeded_state_dict
    cfg = PredCfg()
    m = PEANUT_Prediction_Model(SimpleNamespace(sem_gpu_id=dev.index), state_dict=make_seeded_state_dict(cfg, 0), cfg=cfg)
    x = synth_maps(1, cfg.in_channels, size, dev, seed0=5)
    out = torch.empty((1, cfg.num_classes, size, size), device=dev)
    for _ in range(5):
        m.get_prediction_batch(x, out=out)
    torch.cuda.synchronize()
    reps = 50
    t0 = time.perf_counter()
    for _ in range(reps):
        m.get_prediction_batch(x, out=out)
    torch.cuda.synchronize()
    ms = (time.perf_counter() - t0) / reps * 1e3
    rows = m.model.profile(x, repeats=3)
    executed = sum(r[3] for r in rows)
    del m
    return {"ms": round(ms, 3), "executed_gflop": round(executed / 1e9, 2), "tflops_executed": round(executed / (ms * 1e-3) / 1e12, 2),
            "frac": round(executed / (ms * 1e-3) / 1e12 / FP32_PEAK_TFLOPS, 4)}


def _pipeline_in_a_fresh_process(flags, dev):
    """tools/bench_pipeline.py as a child process: the per-step pipeline the way an agent process runs it -- its handles and buffers
    allocated once, in a process that has done nothing else.  Inside a process that has already built and freed other models (bench.py
    after its headline and the detector configuration) the same loop alternates between 148 and 138-140 steps/s from one run to the
    next (round 6, profiles/r9f: it follows where the re-allocated workspaces land, not the code).  Returns the tool's JSON line as a
    dict, or None."""
    import subprocess
    env = dict(os.environ)
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    if dev is not None and dev.index is not None and "HIP_VISIBLE_DEVICES" not in env and dev.index != 0:
        env["HIP_VISIBLE_DEVICES"] = str(dev.index)
    try:
        r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "bench_pipeline.py")] + flags, env=env, stdout=subprocess.PIPE,
                           stderr=subprocess.DEVNULL, timeout=600, text=True)
        lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
        if r.returncode != 0 or not lines:
            return None
        out = json.loads(lines[-1])
        out["measured_in"] = "a fresh child process (tools/bench_pipeline.py " + " ".join(flags) + ")"
        return out
    except (OSError, ValueError, subprocess.TimeoutExpired):
        return None


def config4(dev, c3=None, mapping=None, cpu_legs=None):
    """one GPU's share of config 4: the bench_pipeline loop with the detector, 2 episodes x 40 frames"""
    import bench_pipeline as bp
    out = _pipeline_in_a_fresh_process(["--episodes", "2", "--frames", "40", "--detector"], dev)
    if out is None:
        out = bp.run_pipeline(episodes=2, frames=40, precision="fp32", detector=True, goal=True, dev=dev)
        out["measured_in"] = "this process"
    out["workload"] = ("config 4: per-step pipeline on synthetic 640x480 frames -- Mask R-CNN + mask accumulation + observation "
                       "formatting + map projection every step, 720x720 map prediction + long-term goal selection on every 10th "
                       "step and on every step within goal_reached_dist of the goal; 2 episodes x 40 frames on ONE GPU (= its share of the 8-episode job: episodes are independent)")
    out["metric"] = "steps/s"
    out["value"], out["unit"], out["dtype"] = out["steps_per_s"], "steps/s", "f32"
    # where a step goes, stage by stage (each measured on its own, above or here), and a composite fraction: the time-weighted
    # mean of the stages' own roofline fractions (detector: front end on its nominal FLOPs + back half stage by stage; mapping: HBM;
    # prediction: executed FLOPs of a batch-1 720 x 720 forward; goal selection: an iterative solver with no roofline of its own,
    # counted at 0)
    p720 = _pred_b1(dev, 720)
    # prediction + goal selection run on every 10th step AND on every step within goal_reached_dist of the current goal
    # (agent_state.py:240-245): the share of steps that predict is the run's own count, not 1/10
    share = out.get("predictions_per_step") or 0.1
    stages = {"prediction_720_per_step": {"ms": round(p720["ms"] * share, 4), "ms_per_call": p720["ms"], "frac": p720["frac"],
                                          "tflops_executed": p720["tflops_executed"], "calls_per_step": share}}
    if c3 is not None:
        b1 = c3["batch1"]
        front_ms = b1["roofline"].get("front_end_ms") or 0.0
        front_frac = (b1["roofline"].get("front_end_tflops_nominal") or 0.0) / FP32_PEAK_TFLOPS
        back_ms = b1["post"]["back_half_ms"]
        back_frac = b1["post"]["back_half_frac_time_weighted"] or 0.0
        det_ms = b1["ms_per_frame"]
        stages["detector"] = {"ms": det_ms, "front_end_ms": front_ms, "front_end_frac_nominal": round(front_frac, 4),
                              "back_half_ms": back_ms, "back_half_frac": back_frac,
                              "frac": round((front_ms * front_frac + back_ms * back_frac) / max(front_ms + back_ms, 1e-9), 4)}
    if mapping is not None:
        stages["mapping"] = {"ms": mapping["ms_per_step"], "frac": mapping["roofline"]["frac"]}
    # goal selection: its geodesic field runs NEXT TO the prediction forward (peanut_goal_select_begin), so what it adds to a step is
    # the pair's device time minus the forward alone; the serial cost per call comes from a second, short run with the overlap off
    serial = _pipeline_in_a_fresh_process(["--episodes", "1", "--frames", "40", "--serial-goal"], dev) or \
        bp.run_pipeline(episodes=1, frames=40, precision="fp32", detector=False, goal=True, dev=dev, goal_overlap=False)
    pair = out.get("prediction_plus_goal_ms_per_call")
    if pair:
        added = max(pair - p720["ms"], 0.0)
        stages["goal_selection_per_step"] = {"ms": round(added * share, 4), "ms_added_per_call_next_to_the_forward": round(added, 3),
                                             "prediction_plus_goal_ms_per_call": pair,
                                             "ms_per_call_on_its_own": serial.get("goal_selection_ms_per_call"),
                                             "rounds_per_call": out.get("goal_selection_rounds_per_call"),
                                             "passes_per_call": out.get("goal_selection_passes_per_call"),
                                             "calls_unconverged": out.get("goal_selection_calls_unconverged"), "frac": 0.0}
    accounted = sum(v["ms"] for v in stages.values())
    stages["other (observation formatting, map bookkeeping, host)"] = {"ms": round(max(out["ms_per_step"] - accounted, 0.0), 4), "frac": 0.0}
    comp = sum(v["ms"] * v["frac"] for v in stages.values()) / max(out["ms_per_step"], 1e-9)
    out["stages"] = stages
    out["roofline"] = {"bound": "mfma", "kernel": "composite of the step's stages", "frac": round(comp, 4),
                       "note": "time-weighted mean of the stages' own roofline fractions over one step (ms_per_step); stage rows under "
                               "'stages' (detector and mapping from their own configs above, the batch-1 720x720 prediction measured "
                               "here, goal selection = what the pair prediction + goal adds to the forward, from the pipeline's own "
                               "events; both run on predictions_per_step of the steps)"}
    if cpu_legs:
        per_step = cpu_legs["detector_s"] + cpu_legs["mapping_s"] + cpu_legs["pred720_s"] * share
        out["cpu_baseline"] = {"value": round(1.0 / per_step, 4), "unit": "steps/s", "cores": _cpu_threads(), "kind": "port",
                               "sample": "composed from the oracles' host times: rcnn_ref one frame + mapping_ref one step + "
                                         "pspnet_ref one 720x720 map x predictions_per_step (goal selection not counted)", **{k: round(v, 4) for k, v in cpu_legs.items()}}
        out["speedup_vs_cpu_baseline"] = round(out["value"] / out["cpu_baseline"]["value"], 1)
    return out


def measure_configs(which=("1", "3", "4", "5", "mapping"), dev=None, with_cpu=True):
    dev = dev or torch.device("cuda", torch.cuda.current_device())
    res = {}
    t_all = time.perf_counter()
    if "1" in which:
        res["1"] = config1(dev)
    if "5" in which:
        res["5"] = config5(dev)
    torch.cuda.empty_cache()
    if "mapping" in which or "4" in which:
        res["mapping"] = mapping_stage(dev)
    c3, cpu_legs = None, None
    if "3" in which or "4" in which:
        c3, net, _ = config3(dev, with_cpu=with_cpu)
        res["3"] = c3
        del net
        torch.cuda.empty_cache()
        if with_cpu and c3.get("cpu_baseline"):
            from peanut_amd.weights import PredCfg, make_seeded_state_dict
            cfg = PredCfg()
            p720 = _pred_cpu(cfg, make_seeded_state_dict(cfg, 0), 720, runs=1)
            cpu_legs = {"detector_s": 1.0 / c3["cpu_baseline"]["value"], "mapping_s": 1.0 / res["mapping"]["cpu_baseline"]["value"],
                        "pred720_s": 1.0 / p720["value"]}
    if "4" in which:
        res["4"] = config4(dev, c3, res.get("mapping"), cpu_legs)
    res["seconds"] = round(time.perf_counter() - t_all, 1)
    return res


if __name__ == "__main__":
    sel = tuple(sys.argv[1].split(",")) if len(sys.argv) > 1 else ("1", "3", "4", "5", "mapping")
    print(json.dumps(measure_configs(sel)))
