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
        with torch.cuda.device(self.device):
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
        obj, dl = mk(A), mk(4 * A)
        ptrs = lambda ts: (C.c_void_p * 5)(*[t.data_ptr() for t in ts]) if ts is not None else None  # noqa: E731
        with torch.cuda.device(img_bgr.device):
            rc = self._lib.peanut_rcnn_forward_front(self._h, img_bgr.data_ptr(), b, h, w, ptrs(pyr), ptrs(obj), ptrs(dl),
                                                     _lib.current_stream_ptr(img_bgr.device))
        _lib.check(rc, "peanut_rcnn_forward_front")
        return pyr, obj, dl


# =========================================================================================================
# Full inference: proposal selection + ROI heads + mask pasting.
#
# detectron2 itself runs these stages as Python/torch glue around three native operators (ROIAlign, nms,
# the paste resample) and dense layers.  The same split is kept: the dense layers go through the fused
# conv kernel (FC layers as 1x1 convs over [N,1,1,K]), the three operators are the HIP kernels of
# csrc/rcnn_ops.hip, and the glue (sorting, top-k, box arithmetic, softmax, gathers) stays torch ops on
# device tensors -- no host round trip until the masks are returned.
# =========================================================================================================
import math

from .ops import FusedConv

_SCALE_CLAMP = math.log(1000.0 / 16)


def _cell_anchors(size, ratios, device):
    out = []
    for r in ratios:
        w = math.sqrt(size * size / r)
        h = r * w
        out.append([-w / 2.0, -h / 2.0, w / 2.0, h / 2.0])
    return torch.tensor(out, dtype=torch.float32, device=device)


def grid_anchors(hw, stride, size, ratios, device):
    """DefaultAnchorGenerator, offset 0 (yaml :41-57): [(h*w*A), 4] ordered (y, x, a)."""
    h, w = hw
    sx = torch.arange(0, w * stride, step=stride, dtype=torch.float32, device=device)
    sy = torch.arange(0, h * stride, step=stride, dtype=torch.float32, device=device)
    yy, xx = torch.meshgrid(sy, sx, indexing="ij")
    shifts = torch.stack((xx.reshape(-1), yy.reshape(-1), xx.reshape(-1), yy.reshape(-1)), dim=1)
    return (shifts.view(-1, 1, 4) + _cell_anchors(size, ratios, device).view(1, -1, 4)).reshape(-1, 4)


def apply_deltas(deltas, boxes, weights):
    """Box2BoxTransform.apply_deltas."""
    widths = boxes[:, 2] - boxes[:, 0]
    heights = boxes[:, 3] - boxes[:, 1]
    ctr_x = boxes[:, 0] + 0.5 * widths
    ctr_y = boxes[:, 1] + 0.5 * heights
    wx, wy, ww, wh = weights
    dx, dy = deltas[:, 0::4] / wx, deltas[:, 1::4] / wy
    dw = torch.clamp(deltas[:, 2::4] / ww, max=_SCALE_CLAMP)
    dh = torch.clamp(deltas[:, 3::4] / wh, max=_SCALE_CLAMP)
    pcx = dx * widths[:, None] + ctr_x[:, None]
    pcy = dy * heights[:, None] + ctr_y[:, None]
    pw, ph = torch.exp(dw) * widths[:, None], torch.exp(dh) * heights[:, None]
    return torch.stack((pcx - 0.5 * pw, pcy - 0.5 * ph, pcx + 0.5 * pw, pcy + 0.5 * ph), dim=-1).reshape(deltas.shape)


def clip_boxes(b, hw):
    h, w = hw
    return torch.stack((b[:, 0].clamp(0, w), b[:, 1].clamp(0, h), b[:, 2].clamp(0, w), b[:, 3].clamp(0, h)), dim=1)


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


def batched_nms_segments(items, thr):
    """``items``: per image (boxes [n,4], scores [n], categories [n]).  torchvision.ops.batched_nms semantics per
    image, evaluated for all images at once; returns per image the kept indices in decreasing-score order."""
    orders = [torch.argsort(s, descending=True, stable=True) for _, s, _ in items]
    if not items:
        return []
    boxes = torch.cat([b[o] for (b, _, _), o in zip(items, orders)], 0)
    cats = torch.cat([c[o] for (_, _, c), o in zip(items, orders)], 0)
    counts = [len(o) for o in orders]
    keep = nms_keep_segments(boxes, cats, counts, thr)
    out, start = [], 0
    for o, k in zip(orders, counts):
        out.append(o[keep[start:start + k]])
        start += k
    return out


def batched_nms(boxes, scores, cats, thr):
    """torchvision.ops.batched_nms semantics: kept indices in decreasing-score order."""
    return batched_nms_segments([(boxes, scores, cats)], thr)[0]


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


def assign_levels(boxes, min_level=2, max_level=5, canonical_box_size=224, canonical_level=4):
    sizes = torch.sqrt((boxes[:, 2] - boxes[:, 0]) * (boxes[:, 3] - boxes[:, 1]))
    lv = torch.floor(canonical_level + torch.log2(sizes / canonical_box_size + 1e-8))
    return torch.clamp(lv, min=min_level, max=max_level).to(torch.int64) - min_level


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
        self._anchor_cache = {}
        sd = {k: v.detach().float().cpu() for k, v in state_dict.items() if k.startswith("roi_heads.")}
        P, Fo, K = cfg.box_pooler_resolution, cfg.fpn_out, cfg.num_classes
        # fc1 consumes ROIAlign output flattened NHWC ((y*P+x)*C + c); detectron2 flattens NCHW (c*P*P + y*P + x)
        w1 = sd["roi_heads.box_head.fc1.weight"].view(cfg.fc_dim, Fo, P, P).permute(0, 2, 3, 1).reshape(cfg.fc_dim, -1)
        lin = lambda w, b, relu: FusedConv(w[:, :, None, None], None, b, relu=relu, precision=precision,  # noqa: E731
                                           device=self.device)
        self.fc1 = lin(w1, sd["roi_heads.box_head.fc1.bias"], True)
        self.fc2 = lin(sd["roi_heads.box_head.fc2.weight"], sd["roi_heads.box_head.fc2.bias"], True)
        self.cls_score = lin(sd["roi_heads.box_predictor.cls_score.weight"], sd["roi_heads.box_predictor.cls_score.bias"], False)
        self.bbox_pred = lin(sd["roi_heads.box_predictor.bbox_pred.weight"], sd["roi_heads.box_predictor.bbox_pred.bias"], False)
        self.mask_fcn = [FusedConv(sd[f"roi_heads.mask_head.mask_fcn{i + 1}.weight"], None,
                                   sd[f"roi_heads.mask_head.mask_fcn{i + 1}.bias"], padding=1, relu=True, precision=precision,
                                   conv_algo=conv_algo, device=self.device)
                         for i in range(cfg.num_mask_convs)]
        # ConvTranspose2d(k=2, s=2) = four 1x1 convs, one per output sub-pixel (dy,dx): rows (dy*2+dx)*C + n
        wd, bd = sd["roi_heads.mask_head.deconv.weight"], sd["roi_heads.mask_head.deconv.bias"]
        wd4 = wd.permute(2, 3, 1, 0).reshape(4 * wd.shape[1], wd.shape[0])        # [(dy,dx,n), c]
        self.deconv = lin(wd4, bd.repeat(4), True)
        self.mask_pred = FusedConv(sd["roi_heads.mask_head.predictor.weight"], None, sd["roi_heads.mask_head.predictor.bias"],
                                   precision=precision, device=self.device)

    # ---- RPN.predict_proposals + find_top_rpn_proposals ----
    def proposals(self, obj: List[torch.Tensor], deltas: List[torch.Tensor], image_hw):
        """obj[l] [B,h,w,A], deltas[l] [B,h,w,4A] (NHWC, as forward_front returns them) ->
        per image (boxes [n,4], objectness logits [n])."""
        cfg = self.cfg
        B = obj[0].shape[0]
        sc, dls, ans, lv = [], [], [], []
        for l, (o, d) in enumerate(zip(obj, deltas)):
            _, h, w, A = o.shape
            key = (l, h, w, str(o.device))
            if key not in self._anchor_cache:                         # anchors depend on the level geometry only
                self._anchor_cache[key] = grid_anchors((h, w), 4 * 2 ** l, cfg.anchor_sizes[l], cfg.aspect_ratios, o.device)
            logits = o.reshape(B, -1)
            k = min(logits.shape[1], cfg.rpn_pre_nms_topk)
            s, idx = logits.sort(descending=True, dim=1)
            s, idx = s[:, :k], idx[:, :k]
            dls.append(d.reshape(B, -1, 4).gather(1, idx[:, :, None].expand(-1, -1, 4)))     # decode only the top-k
            ans.append(self._anchor_cache[key][idx.reshape(-1)].view(B, k, 4))
            sc.append(s)
            lv.append(torch.full((k,), l, dtype=torch.int64, device=o.device))
        sc, lv = torch.cat(sc, 1), torch.cat(lv, 0)
        n_all = sc.shape[1]
        # one decode + clip + validity pass for all levels and images
        pr = apply_deltas(torch.cat(dls, 1).reshape(-1, 4), torch.cat(ans, 1).reshape(-1, 4), cfg.rpn_bbox_weights)
        valid = torch.isfinite(pr).all(1) & torch.isfinite(sc.reshape(-1))
        pr = clip_boxes(pr, image_hw)
        ok = (valid & ((pr[:, 2] - pr[:, 0]) > 0) & ((pr[:, 3] - pr[:, 1]) > 0)).view(B, n_all)
        pr = pr.view(B, n_all, 4)
        items = [(pr[n][ok[n]], sc[n][ok[n]], lv[ok[n]]) for n in range(B)]
        keeps = batched_nms_segments(items, cfg.rpn_nms_thresh)          # all images in one pair of launches
        return [(b[k[:cfg.rpn_post_nms_topk]], s[k[:cfg.rpn_post_nms_topk]]) for (b, s, _), k in zip(items, keeps)]

    # ---- StandardROIHeads._forward_box (inference) ----
    def box_branch(self, pyr: List[torch.Tensor], rois: torch.Tensor):
        """rois [N,5] -> (class logits [N,K+1], box deltas [N,4K])."""
        x = roi_align_pyramid(pyr, rois, assign_levels(rois[:, 1:]), self.cfg.box_pooler_resolution)
        n, P = x.shape[0], self.cfg.box_pooler_resolution
        if n == 0:          # no valid proposal in the whole batch: empty Instances, like detectron2
            K = self.cfg.num_classes
            return x.new_zeros((0, K + 1)), x.new_zeros((0, 4 * K))
        x = x.reshape(n, 1, 1, P * P * self.cfg.fpn_out)
        x = self.fc2(self.fc1(x))
        return self.cls_score(x).reshape(x.shape[0], -1), self.bbox_pred(x).reshape(x.shape[0], -1)

    def detections_batch(self, per_image, image_hw):
        """fast_rcnn_inference: ``per_image`` = list of (boxes [R,4K] decoded, scores [R,K+1] softmax) ->
        list of (boxes [n,4], scores [n], classes [n])."""
        cfg = self.cfg
        items, cls = [], []
        for boxes, scores in per_image:
            valid = torch.isfinite(boxes).all(1) & torch.isfinite(scores).all(1)
            boxes, scores = boxes[valid], scores[valid]
            scores = scores[:, :-1]
            K = boxes.shape[1] // 4
            boxes = clip_boxes(boxes.reshape(-1, 4), image_hw).view(-1, K, 4)
            mask = scores > cfg.score_thresh_test
            inds = mask.nonzero()
            items.append((boxes[mask], scores[mask], inds[:, 1]))
        keeps = batched_nms_segments(items, cfg.nms_thresh_test)
        return [(b[k[:cfg.detections_per_image]], s[k[:cfg.detections_per_image]], c[k[:cfg.detections_per_image]])
                for (b, s, c), k in zip(items, keeps)]

    def detections(self, boxes, scores, image_hw):
        """fast_rcnn_inference_single_image: boxes [R,4K] decoded, scores [R,K+1] softmax."""
        return self.detections_batch([(boxes, scores)], image_hw)[0]

    # ---- MaskRCNNConvUpsampleHead + mask_rcnn_inference ----
    def mask_branch(self, pyr: List[torch.Tensor], rois: torch.Tensor, classes: torch.Tensor) -> torch.Tensor:
        """rois [N,5], classes [N] -> mask probabilities [N,2P,2P] of each instance's predicted class."""
        n, P = rois.shape[0], self.cfg.mask_pooler_resolution
        if n == 0:
            return torch.zeros((0, 2 * P, 2 * P), dtype=torch.float32, device=rois.device)
        x = roi_align_pyramid(pyr, rois, assign_levels(rois[:, 1:]), P)
        for conv in self.mask_fcn:
            x = conv(x)
        x = self.deconv(x)                                                  # [N,P,P,4*C]: (dy,dx,c)
        Cc = x.shape[3] // 4
        logits = self.mask_pred(x.reshape(n, P, P * 4, Cc))                 # [N,P,P*4,K]
        Kc = logits.shape[3]
        logits = logits.reshape(n, P, P, 2, 2, Kc).permute(0, 5, 1, 3, 2, 4).reshape(n, Kc, 2 * P, 2 * P)
        return logits[torch.arange(n, device=rois.device), classes].sigmoid()

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
        return sem

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

    def inference_glue(self, img_bgr: torch.Tensor):
        """The same pipeline with the selection stages as torch ops around the operator exports (the form the stage
        tests bisect with); ``inference`` is the product path."""
        cfg = self.cfg
        B, H, W, _ = img_bgr.shape
        nh, nw = self.plan(B, H, W)["resized"]
        pyr, obj, dl = self.forward_front(img_bgr)
        props = self.proposals(obj, dl, (nh, nw))
        rois = torch.cat([torch.cat([torch.full((len(b), 1), float(n), device=b.device), b], 1) for n, (b, _) in enumerate(props)], 0)
        cls_logits, box_deltas = self.box_branch(pyr, rois)
        probs = torch.softmax(cls_logits, dim=-1)
        dec = apply_deltas(box_deltas, rois[:, 1:], cfg.roi_bbox_weights)
        per_image, start = [], 0
        for b, _ in props:
            per_image.append((dec[start:start + len(b)], probs[start:start + len(b)]))
            start += len(b)
        dets = self.detections_batch(per_image, (nh, nw))
        mrois = torch.cat([torch.cat([torch.full((len(b), 1), float(n), device=b.device), b], 1) for n, (b, _, _) in enumerate(dets)], 0)
        mprobs = self.mask_branch(pyr, mrois, torch.cat([c for _, _, c in dets], 0))
        out, start = [], 0
        scale = torch.tensor([W / nw, H / nh, W / nw, H / nh], device=img_bgr.device)
        for n, (b, s, c) in enumerate(dets):
            mp = mprobs[start:start + len(b)]
            start += len(b)
            ob = clip_boxes(b * scale, (H, W))
            ne = ((ob[:, 2] - ob[:, 0]) > 0) & ((ob[:, 3] - ob[:, 1]) > 0)
            ob, s, c, mp = ob[ne], s[ne], c[ne], mp[ne]
            out.append(dict(pred_boxes=ob, scores=s, pred_classes=c, pred_masks=paste_masks(mp, ob, (H, W), cfg.mask_threshold),
                            proposals=props[n][0], mask_probs=mp))
        return out
