#!/usr/bin/env python3
"""Max-abs distance of a precision mode's logits from the committed reference golden vectors (tests/golden/pspnet_golden.npz)
and from the float64 run of the reference model (pspnet_fp64_golden.npz).  Usage: tools/check_mode.py bf16x6 [fp32 ...]"""
import os
import sys
from types import SimpleNamespace

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from peanut_amd.prediction import PEANUT_Prediction_Model  # noqa: E402
from peanut_amd.weights import PredCfg, make_seeded_state_dict  # noqa: E402

z = np.load(os.path.join(ROOT, "tests", "golden", "pspnet_golden.npz"))
z64 = np.load(os.path.join(ROOT, "tests", "golden", "pspnet_fp64_golden.npz"))
cfg = PredCfg()
sd = make_seeded_state_dict(cfg, 0)
for mode in sys.argv[1:] or ["fp32"]:
    m = PEANUT_Prediction_Model(SimpleNamespace(sem_gpu_id=0), state_dict=sd, cfg=cfg, precision=mode)
    out = {}
    for case in ("cfg1_240", "b2_96", "odd_100", "rect_72x104"):
        x = torch.from_numpy(z[f"{case}/input"].astype(np.float32)).cuda()
        got = m.get_prediction_batch(x, apply_sigmoid=False).cpu().numpy()
        out[case] = float(np.abs(got - z[f"{case}/logits"]).max())
        if f"{case}/logits64" in z64:
            out[case + "_vs_fp64"] = float(np.abs(got.astype(np.float64) - z64[f"{case}/logits64"]).max())
    print(mode, {k: f"{v:.2e}" for k, v in out.items()}, flush=True)
    del m
