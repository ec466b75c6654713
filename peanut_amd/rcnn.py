"""Mask R-CNN front end on the HIP library: preprocessing + ResNet-101-FPN + RPN head of the detector
``SemanticPredMaskRCNN`` builds via detectron2 (nav/agent/utils/segmentation.py:30-38).  detectron2 is
third party and absent; see oracle/rcnn_ref.py for what parity is (and is not) pinned against.  The
proposal / ROI stages are not built yet, so this class exposes the dense front end only."""
from __future__ import annotations

import ctypes as C
from typing import Dict, List, Tuple

import numpy as np
import torch

from . import _lib
from .rcnn_weights import RcnnCfg, front_keys


class MaskRCNNFront:
    def __init__(self, cfg: RcnnCfg, state_dict: Dict[str, torch.Tensor], device="cuda:0", precision: str = "fp32"):
        if not torch.cuda.is_available():
            raise _lib.PeanutHipError("MaskRCNNFront needs a HIP device (no CPU fallback)")
        self.cfg, self.device, self.precision = cfg, torch.device(device), precision
        self._lib = _lib.load()
        tensors = []
        for key, shape in front_keys(cfg):
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
        for i in range(3):
            c.pixel_mean[i], c.pixel_std[i] = cfg.pixel_mean[i], cfg.pixel_std[i]
        self._h = C.c_void_p()
        with torch.cuda.device(self.device):
            _lib.check(self._lib.peanut_rcnn_create(C.byref(self._h), C.byref(c), arr, len(tensors)), "peanut_rcnn_create")

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
        obj, dl = mk(A), mk(4 * A)
        ptrs = lambda ts: (C.c_void_p * 5)(*[t.data_ptr() for t in ts]) if ts is not None else None  # noqa: E731
        with torch.cuda.device(img_bgr.device):
            rc = self._lib.peanut_rcnn_forward_front(self._h, img_bgr.data_ptr(), b, h, w, ptrs(pyr), ptrs(obj), ptrs(dl),
                                                     _lib.current_stream_ptr(img_bgr.device))
        _lib.check(rc, "peanut_rcnn_forward_front")
        return pyr, obj, dl
