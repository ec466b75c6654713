#!/usr/bin/env python3
"""Experiment: one batch-32 forward vs two batch-16 forwards on two streams (half-batches are independent: the HBM-bound
kernels of one half can run under the MFMA-bound GEMMs of the other).  Usage: tools/exp_two_streams.py [precision ...]"""
import json
import os
import sys
import time
from types import SimpleNamespace

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from bench import synth_maps  # noqa: E402
from peanut_amd.prediction import PEANUT_Prediction_Model  # noqa: E402
from peanut_amd.weights import PredCfg, make_seeded_state_dict  # noqa: E402

cfg = PredCfg()
sd = make_seeded_state_dict(cfg, 0)
dev = torch.device("cuda:0")
B, S, steps = 32, 480, 10
x = synth_maps(B, cfg.in_channels, S, dev)
for prec in sys.argv[1:] or ["fp32", "bf16x6"]:
    for parts in (1, 2, 4):
        models = [PEANUT_Prediction_Model(SimpleNamespace(sem_gpu_id=0), state_dict=sd, cfg=cfg, precision=prec) for _ in range(parts)]
        streams = [torch.cuda.Stream() for _ in range(parts)]
        xs = [x[i * (B // parts):(i + 1) * (B // parts)].contiguous() for i in range(parts)]
        outs = [torch.empty((B // parts, cfg.num_classes, S, S), device=dev) for _ in range(parts)]

        def step():
            for m, st, xi, oi in zip(models, streams, xs, outs):
                with torch.cuda.stream(st):
                    m.get_prediction_batch(xi, apply_sigmoid=True, out=oi)

        for _ in range(3):
            step()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(steps):
            step()
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        print(json.dumps({"precision": prec, "streams": parts, "maps_per_s": round(B * steps / dt, 1), "ms_per_32": round(dt / steps * 1e3, 3)}), flush=True)
        del models
