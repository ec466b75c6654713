#!/usr/bin/env python3
"""BASELINE.json config 3 as far as it is built: Mask R-CNN R-101-FPN front end (preprocess + backbone +
FPN + RPN head) on 640x480 RGB frames, batch 16, one MI355X; HIP events on the launch stream."""
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from peanut_amd.rcnn import MaskRCNNFront  # noqa: E402
from peanut_amd.rcnn_weights import RcnnCfg, make_seeded_rcnn_state_dict  # noqa: E402


def main():
    cfg = RcnnCfg()
    sd = make_seeded_rcnn_state_dict(cfg, 0)
    B = int(sys.argv[1]) if len(sys.argv) > 1 else 16
    img = torch.randint(0, 256, (B, 480, 640, 3), dtype=torch.uint8, device="cuda")
    for prec in ("fp32", "bf16x3"):
        m = MaskRCNNFront(cfg, sd, precision=prec)
        plan = m.plan(B, 480, 640)
        for _ in range(2):
            m.forward_front(img, want_pyramid=False)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        reps = 5
        e0.record()
        for _ in range(reps):
            m.forward_front(img, want_pyramid=False)
        e1.record()
        torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / reps
        print(json.dumps({"workload": f"config 3 (front end only): R-101-FPN + RPN head, {B} x 640x480 RGB -> 800x1088",
                          "precision": prec, "ms_per_batch": round(ms, 2), "images_per_s": round(B / ms * 1e3, 1),
                          "gflop_per_image": round(plan["flops_per_image"] / 1e9, 1),
                          "tflops": round(B * plan["flops_per_image"] / ms / 1e9, 1),
                          "workspace_gb": round(plan["workspace_bytes"] / 2**30, 2)}), flush=True)
        del m


if __name__ == "__main__":
    main()
