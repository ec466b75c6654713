#!/usr/bin/env python3
"""Sweep torch CPU thread counts / batch sizes for the oracle forward (480x480) on this host, to give
the CPU baseline its best configuration (the oracle is the thing timed: this is the cpu_baseline leg of bench.py,
swept over thread counts)."""
import os, sys, time, json
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import pspnet_ref
from peanut_amd.weights import PredCfg, make_seeded_state_dict
cfg = PredCfg(); sd = make_seeded_state_dict(cfg, 0)
res = []
for threads in (8, 16, 32, 64, 128):
    torch.set_num_threads(threads)
    for b in (1, 4):
        x = (torch.rand(b, 14, 480, 480) > 0.7).float()
        pspnet_ref.forward_batch(sd, x, cfg)
        t0 = time.perf_counter(); n = 0
        while n < 3 and time.perf_counter() - t0 < 12:
            pspnet_ref.forward_batch(sd, x, cfg); n += 1
        dt = (time.perf_counter() - t0) / n
        res.append(dict(threads=threads, batch=b, maps_per_s=round(b / dt, 3)))
        print(res[-1], flush=True)
print(json.dumps(res))
