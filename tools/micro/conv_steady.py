#!/usr/bin/env python3
"""Steady-state rate of the fp32 GEMM kernels with tile turnover taken out: one full wave of 128x128 tiles
(512 workgroup slots) and a very long K, vs the same FLOPs as many short tiles."""
import json, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from peanut_amd.ops import FusedConv


def run(name, M, cin, cout, reps=5, data="randn"):
    g = torch.Generator().manual_seed(0)
    w = torch.randn((cout, cin, 1, 1), generator=g) * 0.02
    conv = FusedConv(w, None, None)
    x = torch.randn((1, 1, M, cin), device="cuda")
    if data == "relu":
        x = torch.relu(x)
    elif data == "zeros":
        x.zero_()
    elif data == "binary":
        x = (x > 0.5).float()
    for _ in range(2):
        conv(x)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        conv(x)
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / reps
    tiles = (M // 128) * ((cout + 127) // 128)
    waves = tiles / 512
    nkt = cin // 32
    print(json.dumps({"case": name, "M": M, "K": cin, "N": cout, "ms": round(ms, 4), "tflops": round(2.0 * M * cin * cout / ms / 1e9, 1),
                      "tiles": tiles, "waves_of_512": round(waves, 2),
                      "us_per_ktile_pair": round(ms * 1e3 / max(waves, 1) / nkt, 3), "data": data, "pw_glds": os.environ.get("PEANUT_PW_GLDS", "1")}), flush=True)


for data in ("randn", "relu", "binary", "zeros"):
    run("layer4 conv1 shape", 115200, 2048, 512, data=data)
    run("one wave, K=8192", 65536, 8192, 128, data=data)
