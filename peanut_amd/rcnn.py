"""Mask R-CNN front end on the HIP library: preprocessing + ResNet-101-FPN + RPN head of the detector
``SemanticPredMaskRCNN`` builds via detectron2 (nav/agent/utils/segmentation.py:30-38).  detectron2 is
third party and absent; see oracle/rcnn_ref.py for what parity is (and is not) pinned against.
``MaskRCNNFront`` exposes the dense front end, ``MaskRCNN`` (below) the whole inference path."""
from __future__ import annotations

import ctypes as C
from typing import Dict, List, Optional, Tuple

import numpy as np
import torch

from . import _lib
from .rcnn_weights import RcnnCfg, front_keys


class MaskRCNNFront:
    def __init__(self, cfg: RcnnCfg, state_dict: Dict[str, torch.Tensor], device="cuda:0", precision: str = "fp32",
                 conv_algo: str = "auto"):
        if not torch.cuda.is_available():
            raise _lib.PeanutHipError("MaskRCNNFront needs a HIP device (no CPU fallback)")
        self.cfg, self.device, self.precision = cfg, torch.device(device), precision
        self._lib = _lib.load()
        tensors = []
        for key, shape in list(front_keys(cfg)) + list(self._extra_keys(cfg, state_dict)):
            if key not in state_dict:
                raise KeyError(f"checkpoint is missing '{key}'")
            t = state_dict[key]
            if tuple(t.shape) != tuple(shape):
                raise ValueError(f"'{key}' has shape {tuple(t.shape)}, expected {shape}")
            tensors.append((key.encode(), np.ascontiguousarray(t.detach().float().cpu().numpy())))
        arr = (_lib.TensorC * len(tensors))()
        for i, (name, a) in enumerate(tensors):
            arr[i].name, arr[i].data, arr[i].ndim = name, a.ctypes.data, a.ndim
            for d in range(a.ndim):
                arr[i].shape[d] = a.shape[d]
        c = _lib.RcnnCfgC()
        c.depth, c.stem_out, c.res2_out, c.stride_in_1x1 = cfg.depth, cfg.stem_out, cfg.res2_out, int(cfg.stride_in_1x1)
        c.fpn_out, c.num_anchors, c.min_size, c.max_size = cfg.fpn_out, cfg.num_anchors, cfg.min_size, cfg.max_size
        c.size_divisibility, c.bn_eps, c.precision = cfg.size_divisibility, cfg.bn_eps, _lib.PRECISIONS[precision]
        c.conv_algo = _lib.CONV_ALGOS[conv_algo]
        self.conv_algo = conv_algo
        for i in range(3):
            c.pixel_mean[i], c.pixel_std[i] = cfg.pixel_mean[i], cfg.pixel_std[i]
        # proposal generator / ROI heads (read by peanut_rcnn_inference)
        for i, v in enumerate(cfg.anchor_sizes):
            c.anchor_sizes[i] = float(v)
        for i, v in enumerate(cfg.aspect_ratios):
            c.aspect_ratios[i] = float(v)
        c.rpn_pre_nms_topk, c.rpn_post_nms_topk, c.rpn_nms_thresh = cfg.rpn_pre_nms_topk, cfg.rpn_post_nms_topk, cfg.rpn_nms_thresh
        c.num_classes, c.box_pooler_resolution, c.mask_pooler_resolution = cfg.num_classes, cfg.box_pooler_resolution, cfg.mask_pooler_resolution
        c.fc_dim, c.mask_conv_dim, c.num_mask_convs = cfg.fc_dim, cfg.mask_conv_dim, cfg.num_mask_convs
        for i in range(4):
            c.rpn_bbox_weights[i], c.roi_bbox_weights[i] = cfg.rpn_bbox_weights[i], cfg.roi_bbox_weights[i]
        c.score_thresh_test, c.nms_thresh_test = cfg.score_thresh_test, cfg.nms_thresh_test
        c.detections_per_image, c.mask_threshold = cfg.detections_per_image, cfg.mask_threshold
        self._h = C.c_void_p()
        with _lib.default_options(), torch.cuda.device(self.device):     # (the option lock: see _lib.default_options)
            _lib.check(self._lib.peanut_rcnn_create(C.byref(self._h), C.byref(c), arr, len(tensors)), "peanut_rcnn_create")

    def _extra_keys(self, cfg, state_dict):
        """State-dict entries beyond the front end that the library should also receive (none here)."""
        return []

    def __del__(self):
        h = getattr(self, "_h", None)
        if h is not None and h.value:
            try:
                self._lib.peanut_rcnn_destroy(h)
            except Exception:  # pragma: no cover
                pass
            self._h = C.c_void_p()

    def plan(self, b: int, h: int, w: int):
        """-> dict(resized=(h,w), padded=(h,w), levels=[(h,w)]*5, workspace_bytes, flops_per_image)"""
        r, p, lv = (C.c_int * 2)(), (C.c_int * 2)(), (C.c_int * 10)()
        ws, fl = C.c_size_t(0), C.c_double(0)
        _lib.check(self._lib.peanut_rcnn_plan(self._h, b, h, w, C.byref(r), C.byref(p), C.byref(lv), C.byref(ws),
                                              C.byref(fl)), "peanut_rcnn_plan")
        return dict(resized=(r[0], r[1]), padded=(p[0], p[1]), levels=[(lv[2 * i], lv[2 * i + 1]) for i in range(5)],
                    workspace_bytes=ws.value, flops_per_image=fl.value)

    def forward_front(self, img_bgr: torch.Tensor, want_pyramid: bool = True) -> Tuple[List[torch.Tensor], ...]:
        """img_bgr uint8 [B,H,W,3] on the device -> (pyramid p2..p6 [B,h,w,256], objectness [B,h,w,A],
        deltas [B,h,w,4A]) as NHWC tensors; enqueued on the current stream."""
        assert img_bgr.is_cuda and img_bgr.dtype == torch.uint8 and img_bgr.dim() == 4 and img_bgr.shape[3] == 3
        img_bgr = img_bgr.contiguous()
        b, h, w, _ = img_bgr.shape
        lv = self.plan(b, h, w)["levels"]
        A, F = self.cfg.num_anchors, self.cfg.fpn_out
        mk = lambda ch: [torch.empty((b, hh, ww, ch), dtype=torch.float32, device=img_bgr.device) for hh, ww in lv]  # noqa: E731
        pyr = mk(F) if want_pyramid else None

        def mk_joined(ch):      # the five levels as views of ONE buffer, level after level: what lets the library run the RPN head
            flat = torch.empty((sum(b * hh * ww for hh, ww in lv) * ch,), dtype=torch.float32, device=img_bgr.device)   # as one chain
            out, off = [], 0
            for hh, ww in lv:
                n = b * hh * ww * ch
                out.append(flat[off:off + n].view(b, hh, ww, ch))
                off += n
            return out
        obj, dl = mk_joined(A), mk_joined(4 * A)
        ptrs = lambda ts: (C.c_void_p * 5)(*[t.data_ptr() for t in ts]) if ts is not None else None  # noqa: E731
        with torch.cuda.device(img_bgr.device):
            rc = self._lib.peanut_rcnn_forward_front(self._h, img_bgr.data_ptr(), b, h, w, ptrs(pyr), ptrs(obj), ptrs(dl),
                                                     _lib.current_stream_ptr(img_bgr.device))
        _lib.check(rc, "peanut_rcnn_forward_front")
        return pyr, obj, dl

    def probe_front(self, img_bgr: torch.Tensor, reps: int = 3):
        """[(op name, kernel family, mean ms, direct-form conv flops)] of the front end (``peanut_rcnn_probe_front``: HIP
        events on the launch stream after every op) -- what bench.py's stage-1 roofline is computed from."""
        assert img_bgr.is_cuda and img_bgr.dtype == torch.uint8 and img_bgr.dim() == 4 and img_bgr.shape[3] == 3
        img_bgr = img_bgr.contiguous()
        b, h, w, _ = img_bgr.shape
        cap = 2048
        names, kernels = (C.c_char_p * cap)(), (C.c_char_p * cap)()
        ms, fl = (C.c_double * cap)(), (C.c_double * cap)()
        with torch.cuda.device(img_bgr.device):
            n = self._lib.peanut_rcnn_probe_front(self._h, img_bgr.data_ptr(), b, h, w, reps, cap, names, kernels, ms, fl,
                                                  _lib.current_stream_ptr(img_bgr.device))
        if n < 0:
            _lib.check(n, "peanut_rcnn_probe_front")
        return [(names[i].decode(), kernels[i].decode(), ms[i], fl[i]) for i in range(min(n, cap))]


# =========================================================================================================
# Full inference: proposal selection + ROI heads + mask pasting.
#
# detectron2 runs these stages as Python/torch glue around three native operators (ROIAlign, nms, the paste
# resample) and dense layers.  Here the whole pipeline is ONE C entry (peanut_rcnn_inference / peanut_rcnn_semantic,
# csrc/rcnn_post.hip): every selection stage is a HIP kernel.  The three operators are also exported on their own
# (peanut_nms_segments / peanut_roi_align / peanut_paste_masks) and wrapped below; the torch-glue form of the
# pipeline that the stage tests bisect with lives in tests/rcnn_glue.py, not in the product package.
# =========================================================================================================


def nms_keep_segments(boxes_sorted: torch.Tensor, cats: Optional["torch.Tensor"], counts: List[int], thr: float) -> torch.Tensor:
    """peanut_nms_segments: keep mask (bool) for ``len(counts)`` independent box lists stored back to back, each
    already sorted by descending score (one list per image: a single pair of launches for the whole batch)."""
    lib = _lib.load()
    n = boxes_sorted.shape[0]
    assert n == sum(counts)
    keep = torch.empty((n,), dtype=torch.uint8, device=boxes_sorted.device)
    if n == 0:
        return keep.bool()
    b = boxes_sorted.contiguous().float()
    c = None if cats is None else cats.to(torch.int32).contiguous()
    offs = (C.c_int * (len(counts) + 1))()
    for i, k in enumerate(counts):
        offs[i + 1] = offs[i] + int(k)
    ws_bytes = sum(lib.peanut_nms_workspace_bytes(int(k)) for k in counts)
    ws = torch.empty((max(ws_bytes, 8),), dtype=torch.uint8, device=b.device)
    with torch.cuda.device(b.device):
        rc = lib.peanut_nms_segments(b.data_ptr(), None if c is None else c.data_ptr(), offs, len(counts), float(thr),
                                     ws.data_ptr(), keep.data_ptr(), _lib.current_stream_ptr(b.device))
    _lib.check(rc, "peanut_nms_segments")
    return keep.bool()


def nms_keep(boxes_sorted: torch.Tensor, cats: Optional["torch.Tensor"], thr: float) -> torch.Tensor:
    """peanut_nms: keep mask (bool) for boxes already sorted by descending score."""
    return nms_keep_segments(boxes_sorted, cats, [boxes_sorted.shape[0]], thr)


def roi_align_pyramid(pyr: List[torch.Tensor], rois: torch.Tensor, levels: torch.Tensor, pooled: int) -> torch.Tensor:
    """ROIPooler(ROIAlignV2, sampling_ratio 0) over NHWC p2..p5: rois [N,5] (batch, x0,y0,x1,y1) -> [N,P,P,C]."""
    lib = _lib.load()
    n, Cc = rois.shape[0], pyr[0].shape[3]
    out = torch.empty((n, pooled, pooled, Cc), dtype=torch.float32, device=rois.device)
    if n == 0:
        return out
    feats = (C.c_void_p * 4)(*[t.data_ptr() for t in pyr[:4]])
    hw = (C.c_int * 8)(*[d for t in pyr[:4] for d in (t.shape[1], t.shape[2])])
    scales = (C.c_float * 4)(*[1.0 / (4 * 2 ** l) for l in range(4)])
    r = rois.contiguous().float()
    lv = levels.to(torch.int32).contiguous()
    with torch.cuda.device(rois.device):
        rc = lib.peanut_roi_align(feats, hw, scales, 4, Cc, r.data_ptr(), lv.data_ptr(), n, pooled, 0, 1, out.data_ptr(),
                                  _lib.current_stream_ptr(rois.device))
    _lib.check(rc, "peanut_roi_align")
    return out


def paste_masks(probs: torch.Tensor, boxes: torch.Tensor, hw, thr: float) -> torch.Tensor:
    """peanut_paste_masks: probs [n,M,M], boxes [n,4] in output pixels -> bool [n,H,W]."""
    lib = _lib.load()
    H, W = hw
    n = probs.shape[0]
    out = torch.empty((n, H, W), dtype=torch.uint8, device=probs.device)
    if n:
        p, b = probs.contiguous().float(), boxes.contiguous().float()
        with torch.cuda.device(probs.device):
            rc = lib.peanut_paste_masks(p.data_ptr(), b.data_ptr(), n, p.shape[1], H, W, float(thr), out.data_ptr(),
                                        _lib.current_stream_ptr(probs.device))
        _lib.check(rc, "peanut_paste_masks")
    return out.bool()


class MaskRCNN(MaskRCNNFront):
    """``GeneralizedRCNN.inference`` + ``detector_postprocess`` as configured by mask_rcnn_R_101_cat9.yaml:
    what ``DefaultPredictor(img)["instances"]`` yields (segmentation.py:45), batched."""

    def _extra_keys(self, cfg, state_dict):
        from .rcnn_weights import roi_head_keys
        return roi_head_keys(cfg)          # the library builds the ROI heads for peanut_rcnn_inference

    def __init__(self, cfg: RcnnCfg, state_dict, device="cuda:0", precision: str = "fp32", conv_algo: str = "auto"):
        from .rcnn_weights import roi_head_keys
        for key, shape in roi_head_keys(cfg):
            if key not in state_dict or tuple(state_dict[key].shape) != tuple(shape):
                raise KeyError(f"checkpoint is missing or mis-shapes '{key}'")
        super().__init__(cfg, state_dict, device=device, precision=precision, conv_algo=conv_algo)

    def inference(self, img_bgr: torch.Tensor, want_masks: bool = True):
        """img_bgr uint8 [B,H,W,3] (device) -> list of dict(pred_boxes [n,4], scores [n], pred_classes [n],
        pred_masks bool [n,H,W]) at the original resolution (detector_postprocess): ONE call of the library's
        ``peanut_rcnn_inference`` (csrc/rcnn_post.hip) -- proposal selection, box head, detection selection, mask head
        and pasting all run as HIP kernels behind the C ABI; the only host read is the detection count per image."""
        assert img_bgr.is_cuda and img_bgr.dtype == torch.uint8 and img_bgr.dim() == 4 and img_bgr.shape[3] == 3
        img_bgr = img_bgr.contiguous()
        B, H, W, _ = img_bgr.shape
        D = self.cfg.detections_per_image
        dev = img_bgr.device
        boxes = torch.empty((B * D, 4), dtype=torch.float32, device=dev)
        scores = torch.empty((B * D,), dtype=torch.float32, device=dev)
        classes = torch.empty((B * D,), dtype=torch.int32, device=dev)
        masks = torch.empty((B * D, H, W), dtype=torch.uint8, device=dev) if want_masks else None
        n_det = (C.c_int * B)()
        with torch.cuda.device(dev):
            rc = self._lib.peanut_rcnn_inference(self._h, img_bgr.data_ptr(), B, H, W, n_det, boxes.data_ptr(), scores.data_ptr(),
                                                 classes.data_ptr(), None if masks is None else masks.data_ptr(),
                                                 _lib.current_stream_ptr(dev))
        _lib.check(rc, "peanut_rcnn_inference")
        out, start = [], 0
        for b in range(B):
            n = int(n_det[b])
            sl = slice(start, start + n)
            out.append(dict(pred_boxes=boxes[sl], scores=scores[sl], pred_classes=classes[sl].long(),
                            pred_masks=(masks[sl].view(torch.bool) if masks is not None else None)))   # 0/1 bytes: a view, not a copy
            start += n
        return out

    def preprocess(self, img_bgr: torch.Tensor) -> torch.Tensor:
        """uint8 [B,H,W,3] (device) -> float32 [B,3,Hp,Wp]: PIL-bilinear resize, normalisation, padding (``peanut_rcnn_preprocess``)."""
        assert img_bgr.is_cuda and img_bgr.dtype == torch.uint8 and img_bgr.dim() == 4 and img_bgr.shape[3] == 3
        img_bgr = img_bgr.contiguous()
        B, H, W, _ = img_bgr.shape
        Hp, Wp = self.plan(B, H, W)["padded"]
        out = torch.empty((B, 3, Hp, Wp), dtype=torch.float32, device=img_bgr.device)
        with torch.cuda.device(img_bgr.device):
            rc = self._lib.peanut_rcnn_preprocess(self._h, img_bgr.data_ptr(), B, H, W, out.data_ptr(), _lib.current_stream_ptr(img_bgr.device))
        _lib.check(rc, "peanut_rcnn_preprocess")
        return out

    def semantic(self, img_bgr: torch.Tensor, n_cats: int, sem_pred_prob_thr: float, goal_thr: float, goal_cats=None):
        """``SemanticPredMaskRCNN.get_prediction`` for a batch (segmentation.py:41-62): img_bgr uint8 [B,H,W,3] (device)
        -> float32 [B,H,W,n_cats+1] per-category sums of the gated instance masks, in ONE ``peanut_rcnn_semantic`` call
        (the instance masks are evaluated per pixel and never written).  ``goal_cats``: per-frame goal category or None."""
        assert img_bgr.is_cuda and img_bgr.dtype == torch.uint8 and img_bgr.dim() == 4 and img_bgr.shape[3] == 3
        img_bgr = img_bgr.contiguous()
        B, H, W, _ = img_bgr.shape
        D = self.cfg.detections_per_image
        dev = img_bgr.device
        boxes = torch.empty((B * D, 4), dtype=torch.float32, device=dev)
        scores = torch.empty((B * D,), dtype=torch.float32, device=dev)
        classes = torch.empty((B * D,), dtype=torch.int32, device=dev)
        sem = torch.empty((B, H, W, n_cats + 1), dtype=torch.float32, device=dev)
        n_det = (C.c_int * B)()
        goals = None
        if goal_cats is not None:
            goals = (C.c_int32 * B)(*[-1 if g is None else int(g) for g in goal_cats])
        with torch.cuda.device(dev):
            rc = self._lib.peanut_rcnn_semantic(self._h, img_bgr.data_ptr(), B, H, W, int(n_cats), float(sem_pred_prob_thr), float(goal_thr),
                                                goals, sem.data_ptr(), n_det, boxes.data_ptr(), scores.data_ptr(), classes.data_ptr(), None,
                                                _lib.current_stream_ptr(dev))
        _lib.check(rc, "peanut_rcnn_semantic")
        self.last_detection_counts = list(n_det)      # detections per frame of this call (the call's one host read)
        return sem

    def set_stage_timing(self, on: bool = True) -> None:
        """Record a HIP event at every stage boundary of the following ``inference`` / ``semantic`` calls
        (``peanut_rcnn_set_stage_timing``; read with ``stage_times``)."""
        _lib.check(self._lib.peanut_rcnn_set_stage_timing(self._h, int(bool(on))), "peanut_rcnn_set_stage_timing")

    def stage_times(self):
        """[(stage, bound, ms, work)] of the last timed call: bound is "mfma" (work = FLOPs), "hbm" (work = algorithmic bytes)
        or "host" (``peanut_rcnn_stage_times``)."""
        cap = 16
        names, bounds = (C.c_char_p * cap)(), (C.c_char_p * cap)()
        ms, work = (C.c_double * cap)(), (C.c_double * cap)()
        n = self._lib.peanut_rcnn_stage_times(self._h, cap, names, bounds, ms, work)
        if n < 0:
            _lib.check(n, "peanut_rcnn_stage_times")
        return [(names[i].decode(), bounds[i].decode(), float(ms[i]), float(work[i])) for i in range(min(n, cap))]

    def debug_stage(self, name: str, shape, dtype=torch.float32) -> torch.Tensor:
        """Copy of a stage buffer of the last ``inference`` call (peanut_rcnn_debug_stage)."""
        ptr, nbytes = C.c_void_p(), C.c_size_t()
        _lib.check(self._lib.peanut_rcnn_debug_stage(self._h, name.encode(), C.byref(ptr), C.byref(nbytes)), "peanut_rcnn_debug_stage")
        out = torch.empty(shape, dtype=dtype, device=self.device)
        need = out.numel() * out.element_size()
        if need > nbytes.value:
            raise ValueError(f"stage {name} holds {nbytes.value} bytes, {need} requested")
        import ctypes
        hip = ctypes.CDLL("libamdhip64.so")
        torch.cuda.synchronize(self.device)
        rc = hip.hipMemcpy(C.c_void_p(out.data_ptr()), ptr, C.c_size_t(need), 3)    # hipMemcpyDeviceToDevice
        if rc != 0:
            raise _lib.PeanutHipError(f"hipMemcpy failed ({rc})")
        return out
