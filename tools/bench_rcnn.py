#!/usr/bin/env python3
"""BASELINE.json config 3: Mask R-CNN R-101-FPN (cat9 yaml) inference on 640x480 RGB frames, batch 16, one
MI355X.  Whole ``SemanticPredMaskRCNN`` device path: preprocess + backbone + FPN + RPN + proposal selection +
ROI heads + mask paste + per-category accumulation (ONE peanut_rcnn_semantic call); wall clock around a synchronised
loop (the call reads the detection counts back once).  Seeded random weights: the
number of detections (hence ROI-head work) is whatever those weights produce -- reported alongside."""
import json
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from peanut_amd.rcnn import MaskRCNN  # noqa: E402
from peanut_amd.rcnn_weights import RcnnCfg, make_seeded_rcnn_state_dict  # noqa: E402
from peanut_amd.segmentation import accumulate_instances  # noqa: E402


def main():
    B = int(sys.argv[1]) if len(sys.argv) > 1 else 16
    algo = sys.argv[2] if len(sys.argv) > 2 else "auto"
    cfg = RcnnCfg(score_thresh_test=0.5)
    sd = make_seeded_rcnn_state_dict(cfg, 0)
    img = torch.randint(0, 256, (B, 480, 640, 3), dtype=torch.uint8, device="cuda")
    for prec in [q for q in os.environ.get("PRECS", "fp32,bf16x6,fp16x3,bf16x3").split(",") if q]:
        m = MaskRCNN(cfg, sd, precision=prec, conv_algo=algo)

        def step_masks():      # instance masks materialised, then accumulated (what a caller that wants the masks pays)
            res = m.inference(img)
            sem = [accumulate_instances(r["pred_masks"], r["pred_classes"], r["scores"], cfg.num_classes, 0.5, 0.5, None) for r in res]
            return res, sem

        def step():            # SemanticPredMaskRCNN.get_prediction: one peanut_rcnn_semantic call
            return m.semantic(img, cfg.num_classes, 0.5, 0.5, None)

        reps = 5
        for _ in range(2):
            res, _ = step_masks()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(reps):
            res, _ = step_masks()
        torch.cuda.synchronize()
        masks_ms = (time.perf_counter() - t0) / reps * 1e3
        for _ in range(2):
            step()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(reps):
            step()
        torch.cuda.synchronize()
        ms = (time.perf_counter() - t0) / reps * 1e3
        t0 = time.perf_counter()
        for _ in range(reps):
            m.forward_front(img)
        torch.cuda.synchronize()
        front_ms = (time.perf_counter() - t0) / reps * 1e3
        print(json.dumps({"workload": f"config 3: Mask R-CNN R-101-FPN full inference + mask accumulation, {B} x 640x480 RGB",
                          "precision": prec, "conv_algo": algo, "ms_per_batch": round(ms, 2), "images_per_s": round(B / ms * 1e3, 1),
                          "front_end_ms": round(front_ms, 2), "proposal_roi_paste_ms": round(ms - front_ms, 2),
                          "with_instance_masks_ms": round(masks_ms, 2),
                          "proposals_per_image": round(float(m.debug_stage("prop_count", (B,), torch.int32).float().mean()), 1),
                          "detections_per_image": round(sum(len(r["scores"]) for r in res) / B, 1)}), flush=True)
        del m


if __name__ == "__main__":
    main()
