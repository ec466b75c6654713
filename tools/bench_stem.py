#!/usr/bin/env python3
"""Operator-level timing of the deep stem's 3x3 convs at the headline benchmark's shapes (B = 32, 480 x 480 maps):
    tools/bench_stem.py [reps]         one JSON line per layer; PEANUT_PATCH_MINTILES=0 gives the conv_igemm baseline.
Inputs are relu(N(0,1)); 3 warm-ups, `reps` timed launches between two events on the launch stream.  ZERO=1: all-zero operands
(clock / power probe).  LAYERS=stem.3,stem.6 restricts the set (PMC passes)."""
import json
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from peanut_amd import _lib  # noqa: E402
from peanut_amd.ops import FusedConv  # noqa: E402

LAYERS = [  # name, (B, H, W) of the input, cin_pad, cin_real, cout, stride
    ("stem.0", (32, 480, 480), 16, 14, 32, 2),
    ("stem.3", (32, 240, 240), 32, 32, 32, 1),
    ("stem.6", (32, 240, 240), 32, 32, 64, 1),
]
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 20
only = os.environ.get("LAYERS", "")
g = torch.Generator().manual_seed(0)
for name, (b, h, w), cpad, creal, cout, stride in LAYERS:
    if only and name not in only.split(","):
        continue
    x = torch.relu(torch.randn((b, h, w, cpad), generator=g))
    x[..., creal:] = 0
    x = x.cuda()
    wt = torch.randn((cout, creal, 3, 3), generator=g) * (2.0 / (creal * 9)) ** 0.5
    if os.environ.get("ZERO") == "1":
        x.zero_()
        wt.zero_()
    conv = FusedConv(wt, None, None, stride=stride, padding=1, relu=True, conv_algo="direct", cin_pad=cpad)
    for _ in range(3):
        conv(x)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        y = conv(x)
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / reps
    ho, wo = y.shape[1], y.shape[2]
    fl = 2.0 * b * ho * wo * cpad * 9 * cout
    by = 4.0 * (x.numel() + y.numel())
    print(json.dumps({"layer": name, "kernel": _lib.load().peanut_last_conv_kernel().decode(), "ms": round(ms, 4),
                      "tflops_padded_k": round(fl / ms / 1e9, 1), "gb_s": round(by / ms / 1e6)}), flush=True)
    del conv
