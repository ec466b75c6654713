#!/usr/bin/env python3
"""Operator-level timing of the pointwise-conv GEMMs at the headline benchmark's shapes (B = 32, 480 x 480 maps):
    tools/bench_gemm.py [precision ...]            one JSON line per (shape, precision)
Inputs are post-ReLU-like (relu of N(0,1)); 3 warm-ups, 20 timed launches between two events on the launch stream."""
import json
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from peanut_amd import _lib  # noqa: E402
from peanut_amd.ops import FusedConv  # noqa: E402

SHAPES = [  # name, rows M (as B x H x W), cin, cout, residual
    ("layer1.conv3", (32, 120, 120), 64, 256, True),
    ("layer2.conv3", (32, 60, 60), 128, 512, True),
    ("layer3.conv1", (32, 60, 60), 1024, 256, False),
    ("layer3.conv3", (32, 60, 60), 256, 1024, True),
    ("layer4.0.conv1", (32, 60, 60), 1024, 512, False),
    ("layer4.conv1", (32, 60, 60), 2048, 512, False),
    ("layer4.conv3", (32, 60, 60), 512, 2048, True),
    ("layer3.0.conv3ds", (32, 60, 60), 768, 1024, False),      # K = 256 + 512 as one source (the fused conv3 + downsample GEMM)
    ("layer4.0.conv3ds", (32, 60, 60), 1536, 2048, False),
]
OPTS = json.loads(os.environ.get("OPTS", "{}"))      # tuning options of the handles (csrc/options.h), e.g. OPTS='{"pw256wp_mink": 0}'
only = os.environ.get("SHAPES", "")
precisions = sys.argv[1:] or ["fp32", "bf16x6"]
g = torch.Generator().manual_seed(0)
for name, (b, h, w), cin, cout, residual in SHAPES:
    if only and name not in only.split(","):
        continue
    x = torch.relu(torch.randn((b, h, w, cin), generator=g)).cuda()
    wt = torch.randn((cout, cin, 1, 1), generator=g) * (2.0 / cin) ** 0.5
    if os.environ.get("ZERO") == "1":          # DVFS probe: all-zero operands draw less power (MI355X_MICROARCH.md)
        x.zero_()
        wt.zero_()
    res = torch.randn((b, h, w, cout), generator=g).cuda() if residual else None
    for prec in precisions:
        conv = FusedConv(wt, None, None, relu=True, precision=prec, options=OPTS)
        for _ in range(3):
            conv(x, residual=res)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20):
            conv(x, residual=res)
        e1.record()
        torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / 20
        fl = 2.0 * b * h * w * cin * cout
        by = 4.0 * b * h * w * (cin + cout * (2 if residual else 1))
        print(json.dumps({"shape": name, "M": b * h * w, "K": cin, "N": cout, "precision": prec,
                          "kernel": _lib.load().peanut_last_conv_kernel().decode(), "ms": round(ms, 4),
                          "tflops": round(fl / ms / 1e9, 1), "gb_s": round(by / ms / 1e6), "opts": OPTS}), flush=True)
        del conv
