#!/usr/bin/env python3
"""Measure the HIP map-projection step on the GPU box: error vs the CPU oracle on seeded sequences
and steps/s (HIP events over a chained sequence), next to the oracle's CPU time.
The oracle is used here exactly as in bench.py's cpu_baseline leg and the tests: as the checker and the timed CPU
baseline, never as part of what is measured on the GPU."""
import json
import os
import sys
import time
from types import SimpleNamespace

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import mapping_ref, mapping_scenes  # noqa: E402
from peanut_amd.mapping import Semantic_Mapping  # noqa: E402


def main():
    args = SimpleNamespace(device=torch.device("cuda:0"), frame_height=120, frame_width=160, map_resolution=5,
                           map_size_cm=4800, global_downscaling=2, vision_range=100, hfov=79.0, du_scale=1,
                           cat_pred_threshold=5.0, exp_pred_threshold=1.0, map_pred_threshold=0.1,
                           num_sem_categories=10, camera_height=0.88)
    sm = Semantic_Mapping(args)
    cfg = mapping_ref.MapCfg()
    frames = mapping_scenes.make_sequence(5, 16)
    obs_c = [torch.from_numpy(mapping_scenes.frame_to_obs(f))[None] for f in frames]
    rel_c = [torch.from_numpy(f["pose"]) for f in frames]
    obs_g = [o.cuda() for o in obs_c]
    rel_g = [r.cuda() for r in rel_c]
    # accuracy
    mg, pg = torch.zeros(14, 480, 480, device="cuda"), torch.tensor([12.0, 12.0, 0.0], device="cuda")
    mc, pc = torch.zeros(14, 480, 480), torch.tensor([12.0, 12.0, 0.0])
    worst, worst_pose, t_cpu = 0.0, 0.0, 0.0
    for o, r, og, rg in zip(obs_c, rel_c, obs_g, rel_g):
        t0 = time.perf_counter()
        _, mc, _, pc = mapping_ref.forward(o, r, mc, pc, cfg)
        t_cpu += time.perf_counter() - t0
        _, mg, _, _ = sm(og, rg, mg, pg, None)
        worst = max(worst, (mg.cpu() - mc).abs().max().item())
        worst_pose = max(worst_pose, (pg.cpu() - pc).abs().max().item())
    # throughput: chained steps, events on the launch stream
    torch.cuda.synchronize()
    reps = 20
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    mg = torch.zeros(14, 480, 480, device="cuda")
    for og, rg in zip(obs_g, rel_g):
        _, mg, _, _ = sm(og, rg, mg, pg, None)
    torch.cuda.synchronize()
    e0.record()
    for _ in range(reps):
        for og, rg in zip(obs_g, rel_g):
            _, mg, _, _ = sm(og, rg, mg, pg, None)
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / (reps * len(frames))
    print(json.dumps({"stage": "map projection (Semantic_Mapping.forward)", "frames": len(frames),
                      "max_abs_vs_oracle": worst, "pose_max_abs": worst_pose, "gpu_ms_per_step": round(ms, 4),
                      "gpu_steps_per_s": round(1e3 / ms, 1), "cpu_oracle_ms_per_step": round(1e3 * t_cpu / len(frames), 2),
                      "cpu_threads": torch.get_num_threads(),
                      "algorithmic_bytes_per_step": 1.07e6 + 2 * 14 * 480 * 480 * 4}))


if __name__ == "__main__":
    main()
