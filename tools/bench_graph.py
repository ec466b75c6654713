#!/usr/bin/env python3
"""Launch-bound cases with and without hipGraph replay (peanut_pred_use_graph / peanut_map_use_graph):
the B=1 map-prediction forwards (~85 launches each) and the map-projection step (10 launches).
Wall clock around a synchronised loop (host launch cost is the point), persistent input/output buffers."""
import json
import os
import sys
import time
from types import SimpleNamespace

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from peanut_amd.mapping import Semantic_Mapping  # noqa: E402
from peanut_amd.prediction import PEANUT_Prediction_Model  # noqa: E402
from peanut_amd.weights import PredCfg, make_seeded_state_dict  # noqa: E402


def timed(fn, reps):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps * 1e3


def main():
    side = torch.cuda.Stream()        # HIP cannot capture the legacy default stream
    with torch.cuda.stream(side):
        run()


def run():
    cfg = PredCfg()
    m = PEANUT_Prediction_Model(SimpleNamespace(sem_gpu_id=0), state_dict=make_seeded_state_dict(cfg, 0), cfg=cfg)
    for name, s, b, reps in (("config1 240x240 B=1", 240, 1, 50), ("480x480 B=1", 480, 1, 50), ("deployed 720x720 B=1", 720, 1, 50),
                             ("headline 480x480 B=32", 480, 32, 10)):
        x = (torch.rand((b, 14, s, s), device="cuda") > 0.7).float()
        y = torch.empty((b, 6, s, s), device="cuda")
        row = {"case": name}
        for g in (False, True, False, True):
            m.model.use_graph(g)
            key = "graph_ms" if g else "plain_ms"
            v = round(timed(lambda: m.get_prediction_batch(x, out=y), reps), 3)
            row[key] = min(row.get(key, 1e9), v)
        print(json.dumps(row), flush=True)
    args = SimpleNamespace(device=torch.device("cuda:0"), frame_height=120, frame_width=160, map_resolution=5,
                           map_size_cm=4800, global_downscaling=2, vision_range=100, hfov=79.0, du_scale=1,
                           cat_pred_threshold=5.0, exp_pred_threshold=1.0, map_pred_threshold=0.1,
                           num_sem_categories=10, camera_height=0.88)
    sm = Semantic_Mapping(args)
    obs = torch.zeros(1, 14, 120, 160, device="cuda")
    obs[0, 3] = 150.0 + 100.0 * torch.rand(120, 160, device="cuda")
    obs[0, 4:] = (torch.rand(10, 120, 160, device="cuda") > 0.97).float()
    rel = torch.tensor([0.05, 0.0, 0.02], device="cuda")
    bufs = [torch.zeros(14, 480, 480, device="cuda") for _ in range(2)]
    fp = torch.zeros(1, 100, 100, device="cuda")
    pose = torch.tensor([12.0, 12.0, 0.0], device="cuda")
    state = {"i": 0}

    def step():
        i = state["i"]
        sm(obs, rel, bufs[i % 2], pose, None, out=(fp, bufs[(i + 1) % 2]))
        state["i"] = i + 1

    row = {"case": "map projection step (10 launches)"}
    for g in (False, True):
        sm.use_graph(g)
        row["graph_ms" if g else "plain_ms"] = round(timed(step, 400), 4)
    print(json.dumps(row), flush=True)


if __name__ == "__main__":
    main()
