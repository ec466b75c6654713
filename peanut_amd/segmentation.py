"""Stage 1 boundary: ``SemanticPredMaskRCNN`` (nav/agent/utils/segmentation.py:28-62).

``get_prediction`` = detector + per-instance score gating and mask accumulation (:47-60).  The reference
delegates the detector to detectron2's ``DefaultPredictor`` (not vendored); here it is
``peanut_amd.rcnn.MaskRCNN`` on HIP by default, or any injected callable returning the three tensors
``DefaultPredictor(img)["instances"]`` would provide."""
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
    masks = (pred_masks.view(torch.uint8) if pred_masks.dtype == torch.bool else pred_masks.to(torch.uint8)).contiguous()
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


class HipDetector:
    """``DefaultPredictor(cfg)`` stand-in: ``detector(img_bgr uint8 [H,W,3]) -> (pred_masks, pred_classes,
    scores)`` computed by ``peanut_amd.rcnn.MaskRCNN`` (HIP front end + ROI stages)."""

    def __init__(self, cfg, state_dict, device="cuda:0", precision: str = "fp32"):
        from .rcnn import MaskRCNN
        self.device = torch.device(device)
        self.net = MaskRCNN(cfg, state_dict, device=device, precision=precision)

    def batch(self, imgs_bgr: torch.Tensor):
        """uint8 [B,H,W,3] device tensor -> list of (pred_masks, pred_classes, scores)."""
        return [(r["pred_masks"], r["pred_classes"], r["scores"]) for r in self.net.inference(imgs_bgr)]

    def __call__(self, img_bgr):
        x = torch.from_numpy(np.ascontiguousarray(img_bgr)) if isinstance(img_bgr, np.ndarray) else img_bgr
        return self.batch(x.to(self.device)[None])[0]

    def semantic(self, img_bgr, n_cats: int, sem_pred_prob_thr: float, goal_thr: float, goal_cat=None) -> torch.Tensor:
        """One frame straight to the per-category mask sums [H,W,n_cats+1] (``peanut_rcnn_semantic``)."""
        x = torch.from_numpy(np.ascontiguousarray(img_bgr)) if isinstance(img_bgr, np.ndarray) else img_bgr
        return self.net.semantic(x.to(self.device)[None], n_cats, sem_pred_prob_thr, goal_thr, [goal_cat])[0]


class SemanticPredMaskRCNN():
    """Same call surface as the reference class (segmentation.py:28-62): ``SemanticPredMaskRCNN(args)``,
    ``get_prediction(img_rgb_uint8[H,W,3], depth=None, goal_cat=None) -> (np.float32 [H,W,n_cats+1], img_bgr)``.

    ``args`` carries ``seg_model_wts`` (detectron2 checkpoint), ``sem_pred_prob_thr`` (also the detector's
    SCORE_THRESH_TEST, segmentation.py:33), ``goal_thr`` and ``sem_gpu_id``.  A different ``detector`` callable
    returning ``(pred_masks [n,H,W], pred_classes [n], scores [n])`` HIP tensors may be injected instead
    (what ``DefaultPredictor(img)["instances"]`` holds, segmentation.py:45)."""

    def __init__(self, args, detector: Optional[Callable[[np.ndarray], Tuple[torch.Tensor, torch.Tensor, torch.Tensor]]] = None,
                 n_cats: Optional[int] = None, rcnn_cfg=None, state_dict=None, precision: str = "fp32"):
        self.args = args
        if detector is None:
            from dataclasses import replace
            from .rcnn_weights import RcnnCfg, load_detectron2_checkpoint
            cfg = replace(rcnn_cfg or RcnnCfg(), score_thresh_test=float(args.sem_pred_prob_thr))
            sd = state_dict if state_dict is not None else load_detectron2_checkpoint(args.seg_model_wts)
            detector = HipDetector(cfg, sd, device=f"cuda:{int(getattr(args, 'sem_gpu_id', 0))}", precision=precision)
            n_cats = cfg.num_classes if n_cats is None else n_cats
        self.n_cats = 9 if n_cats is None else n_cats   # cfg.MODEL.ROI_HEADS.NUM_CLASSES (mask_rcnn_R_101_cat9.yaml:193)
        self.predictor = detector

    def get_prediction(self, img, depth=None, goal_cat=None):
        args = self.args
        img = img[:, :, ::-1]                                   # RGB -> BGR (segmentation.py:44)
        if isinstance(self.predictor, HipDetector):             # detector + accumulation in one library call
            return self.predictor.semantic(img, self.n_cats, args.sem_pred_prob_thr, args.goal_thr, goal_cat).cpu().numpy(), img
        masks, classes, scores = self.predictor(img)
        semantic_input = accumulate_instances(masks, classes, scores, self.n_cats, args.sem_pred_prob_thr,
                                              args.goal_thr, goal_cat)
        return semantic_input.cpu().numpy(), img
