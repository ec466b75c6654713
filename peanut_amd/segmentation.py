"""Stage 1 boundary: ``SemanticPredMaskRCNN`` (nav/agent/utils/segmentation.py:28-62).

Only the part of the reference that lives IN the reference is built here: the per-instance score
gating and mask accumulation (``get_prediction`` :47-60).  The Mask R-CNN itself is detectron2
(not vendored, not installed, weights not shipped), so the detector is an injected
callable returning the three tensors ``DefaultPredictor(img)["instances"]`` would provide."""
from __future__ import annotations

from typing import Callable, Optional, Tuple

import numpy as np
import torch

from . import _lib


def accumulate_instances(pred_masks: torch.Tensor, pred_classes: torch.Tensor, scores: torch.Tensor,
                         n_cats: int, sem_pred_prob_thr: float, goal_thr: float,
                         goal_cat: Optional[int]) -> torch.Tensor:
    """segmentation.py:46-60 as one device launch: [n,H,W] bool/uint8 masks, [n] classes, [n] scores
    -> float32 [H,W,n_cats+1] (channel n_cats stays zero).  No host sync."""
    lib = _lib.load()
    if not pred_masks.is_cuda:
        raise _lib.PeanutHipError("accumulate_instances needs HIP tensors (no CPU fallback)")
    n, H, W = pred_masks.shape
    masks = pred_masks.to(torch.uint8).contiguous()
    classes = pred_classes.to(torch.int32).contiguous()
    sc = scores.to(torch.float32).contiguous()
    out = torch.empty((H, W, n_cats + 1), dtype=torch.float32, device=pred_masks.device)
    with torch.cuda.device(pred_masks.device):
        rc = lib.peanut_seg_accumulate(masks.data_ptr() if n else None, classes.data_ptr() if n else None,
                                       sc.data_ptr() if n else None, n, H, W, n_cats, float(sem_pred_prob_thr),
                                       float(goal_thr), -1 if goal_cat is None else int(goal_cat), out.data_ptr(),
                                       _lib.current_stream_ptr(pred_masks.device))
    _lib.check(rc, "peanut_seg_accumulate")
    return out


class SemanticPredMaskRCNN():
    """Same call surface as the reference class: ``get_prediction(img_rgb_uint8[H,W,3], depth=None,
    goal_cat=None) -> (np.float32 [H,W,n_cats+1], img_bgr)``.  ``detector(img_bgr)`` must return
    ``(pred_masks [n,H,W], pred_classes [n], scores [n])`` as HIP tensors (what
    ``DefaultPredictor(img)["instances"]`` holds, segmentation.py:45)."""

    def __init__(self, args, detector: Callable[[np.ndarray], Tuple[torch.Tensor, torch.Tensor, torch.Tensor]],
                 n_cats: int = 9):
        self.args = args
        self.n_cats = n_cats          # cfg.MODEL.ROI_HEADS.NUM_CLASSES (mask_rcnn_R_101_cat9.yaml:193)
        self.predictor = detector

    def get_prediction(self, img, depth=None, goal_cat=None):
        args = self.args
        img = img[:, :, ::-1]                                   # RGB -> BGR (segmentation.py:44)
        masks, classes, scores = self.predictor(img)
        semantic_input = accumulate_instances(masks, classes, scores, self.n_cats, args.sem_pred_prob_thr,
                                              args.goal_thr, goal_cat)
        return semantic_input.cpu().numpy(), img
