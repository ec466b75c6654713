#!/usr/bin/env python3
"""Latency / throughput of the map-prediction forward on the BASELINE.json configs that are parity-test
cases rather than the headline bench line (config 1: 240x240 B=1; deployed window 720x720 B=1;
config 5's per-GPU share: 960x960, 25 channels, B=8), HIP events on the launch stream."""
import json
import os
import sys
from types import SimpleNamespace

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from peanut_amd.prediction import PEANUT_Prediction_Model  # noqa: E402
from peanut_amd.weights import PredCfg, conv_flops_per_map, make_seeded_state_dict  # noqa: E402

CONFIGS = [("config1 240x240 B=1", 14, 1, 240), ("deployed 720x720 B=1", 14, 1, 720),
           ("480x480 B=1", 14, 1, 480), ("headline 480x480 B=32", 14, 32, 480),
           ("480x480 C=13 (4+9) B=32", 13, 32, 480),
           ("config5 share 960x960 C=25 B=8", 25, 8, 960)]


def main():
    out = []
    for prec in ("fp32", "bf16x6", "bf16x3"):
        models = {}
        for name, c, b, s in CONFIGS:
            cfg = PredCfg(in_channels=c)
            if c not in models:
                models[c] = PEANUT_Prediction_Model(SimpleNamespace(sem_gpu_id=0), state_dict=make_seeded_state_dict(cfg, 0),
                                                    cfg=cfg, precision=prec)
            m = models[c]
            x = (torch.rand((b, c, s, s), device="cuda") > 0.7).float()
            y = torch.empty((b, 6, s, s), device="cuda")
            for _ in range(3):
                m.get_prediction_batch(x, out=y)
            torch.cuda.synchronize()
            reps = 20 if b * s * s <= 32 * 480 * 480 // 4 else 8
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(reps):
                m.get_prediction_batch(x, out=y)
            e1.record()
            torch.cuda.synchronize()
            ms = e0.elapsed_time(e1) / reps
            row = dict(config=name, precision=prec, ms_per_forward=round(ms, 3), maps_per_s=round(b / ms * 1e3, 1),
                       nominal_tflops=round(b * conv_flops_per_map(cfg, s, s) / ms / 1e9, 1),
                       workspace_gb=round(m.model.workspace_bytes(b, s, s) / 2**30, 2))
            out.append(row)
            print(json.dumps(row), flush=True)
        del models
    if len(sys.argv) > 1:
        json.dump(out, open(sys.argv[1], "w"), indent=1)


if __name__ == "__main__":
    main()
