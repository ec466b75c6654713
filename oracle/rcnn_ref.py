"""ORACLE (test infrastructure, not product code) -- CPU restatement, in plain fp32 PyTorch, of the
FRONT END of the Mask R-CNN the reference runs through detectron2's ``DefaultPredictor``
(nav/agent/utils/segmentation.py:31-38,45): test-time preprocessing, ResNet-101-FPN backbone and the
RPN head, as configured by nav/agent/utils/COCO-InstSeg/mask_rcnn_R_101_cat9.yaml.

**PARITY UNPINNED against the reference** (partially pinned against detectron2's own published unit tests: ROIAlign
and anchor generation reproduce the known-answer vectors of tests/layers/test_roi_align.py::test_forward_output and
tests/modeling/test_anchor_generator.py::test_default_anchor_generator, see tests/test_oracles_cpu.py).
The arithmetic lives in detectron2 (third party; peanut.Dockerfile:15 installs the
cu111/torch1.10 wheel => v0.6), which is not vendored, not installed, has no network route here, and whose
fine-tuned weights are a Drive link.  The reference has no test or golden vector at this boundary.  This
file therefore restates detectron2 v0.6's PUBLISHED module definitions (``DefaultPredictor.__call__``,
``ResizeShortestEdge``, ``GeneralizedRCNN.preprocess_image``, ``BasicStem``, ``BottleneckBlock``,
``FrozenBatchNorm2d``, ``FPN`` + ``LastLevelMaxPool``, ``StandardRPNHead``) under that yaml; it cannot be
validated against the reference itself.  One known approximation: detectron2 resizes the uint8 image with
PIL's fixed-point BILINEAR filter; here it is float bilinear (half-pixel centres) rounded back to uint8.
Only tests/, smoke() and bench baselines may import this module."""
from __future__ import annotations

import math
from typing import Dict, List

import numpy as np
import torch
import torch.nn.functional as F

from peanut_amd.rcnn_weights import RcnnCfg, padded_hw, resized_hw


_PIL_PRECISION_BITS = 32 - 8 - 2


def pil_bilinear_coeffs(in_size: int, out_size: int):
    """Pillow ``precompute_coeffs`` + ``normalize_coeffs_8bpc`` (src/libImaging/Resample.c) for the bilinear filter:
    per output index the first source index, the tap count and the fixed-point weights (22 fractional bits)."""
    scale = in_size / out_size
    filterscale = max(scale, 1.0)
    support = 1.0 * filterscale
    ksize = int(math.ceil(support)) * 2 + 1
    ik = np.zeros((out_size, ksize), np.int64)
    bounds = np.zeros((out_size, 2), np.int64)
    ss = 1.0 / filterscale
    for xx in range(out_size):
        center = (xx + 0.5) * scale
        xmin = max(int(center - support + 0.5), 0)
        xmax = min(int(center + support + 0.5), in_size) - xmin
        w = []
        for x in range(xmax):
            a = abs((x + xmin - center + 0.5) * ss)
            w.append(1.0 - a if a < 1.0 else 0.0)
        ww = sum(w)         # left to right, as the C loop adds them
        for x in range(xmax):
            v = w[x] / ww if ww != 0.0 else w[x]
            ik[xx, x] = int(-0.5 + v * (1 << _PIL_PRECISION_BITS)) if v < 0 else int(0.5 + v * (1 << _PIL_PRECISION_BITS))
        bounds[xx] = (xmin, xmax)
    return ik, bounds


def _pil_resample_axis(img: np.ndarray, out_size: int, axis: int) -> np.ndarray:
    ik, b = pil_bilinear_coeffs(img.shape[axis], out_size)
    x = np.moveaxis(img, axis, 0).astype(np.int64)
    acc = np.full((out_size,) + x.shape[1:], 1 << (_PIL_PRECISION_BITS - 1), np.int64)
    for t in range(ik.shape[1]):                                              # taps; rows past a bound carry weight 0
        src = np.minimum(b[:, 0] + t, x.shape[0] - 1)
        acc += x[src] * ik[:, t].reshape((-1,) + (1,) * (x.ndim - 1))
    return np.moveaxis(np.clip(acc >> _PIL_PRECISION_BITS, 0, 255).astype(np.uint8), 0, axis)


def pil_resize_bilinear_u8(img: np.ndarray, nh: int, nw: int) -> np.ndarray:
    """``PIL.Image.fromarray(img).resize((nw, nh), Image.BILINEAR)`` restated (what detectron2's ResizeTransform
    applies to uint8 images): Pillow's two-pass ImagingResample -- horizontal, then vertical over the uint8 result of
    the first pass, fixed-point coefficients, round half up, clip.  uint8 [H,W,C] -> uint8 [nh,nw,C].  Equal to
    Pillow's own output (tests/test_oracles_cpu.py, tests/golden/pil_resize_golden.npz)."""
    t = _pil_resample_axis(img, nw, 1) if nw != img.shape[1] else img
    return _pil_resample_axis(t, nh, 0) if nh != img.shape[0] else t


def preprocess(img_bgr_u8: torch.Tensor, cfg: RcnnCfg) -> torch.Tensor:
    """uint8 [B,H,W,3] BGR -> float32 [B,3,Hp,Wp]: resize (yaml :28-30; PIL bilinear as ResizeShortestEdge applies it),
    (x - PIXEL_MEAN) / PIXEL_STD (:82-89), zero-pad to a multiple of 32 (ImageList.from_tensors with the FPN's
    size_divisibility)."""
    b, h, w, _ = img_bgr_u8.shape
    nh, nw = resized_hw(h, w, cfg)
    x = torch.stack([torch.from_numpy(pil_resize_bilinear_u8(im.numpy(), nh, nw)) for im in img_bgr_u8])
    x = x.permute(0, 3, 1, 2).float()
    mean = torch.tensor(cfg.pixel_mean).view(1, 3, 1, 1)
    std = torch.tensor(cfg.pixel_std).view(1, 3, 1, 1)
    x = (x - mean) / std
    ph, pw = padded_hw(nh, nw, cfg)
    return F.pad(x, (0, pw - nw, 0, ph - nh), value=0.0)


def _conv(sd, name, x, stride=1, pad=0, norm=True, relu=False, eps=1e-5):
    """detectron2 ``Conv2d`` wrapper: conv (bias only without norm) -> FrozenBatchNorm2d -> activation."""
    y = F.conv2d(x, sd[f"{name}.weight"], None if norm else sd[f"{name}.bias"], stride=stride, padding=pad)
    if norm:   # FrozenBatchNorm2d.forward: scale = weight * (var + eps).rsqrt(); bias = bias - mean * scale
        scale = sd[f"{name}.norm.weight"] * (sd[f"{name}.norm.running_var"] + eps).rsqrt()
        bias = sd[f"{name}.norm.bias"] - sd[f"{name}.norm.running_mean"] * scale
        y = y * scale.reshape(1, -1, 1, 1) + bias.reshape(1, -1, 1, 1)
    return F.relu(y) if relu else y


def bottom_up(sd, x: torch.Tensor, cfg: RcnnCfg) -> Dict[str, torch.Tensor]:
    """``ResNet.forward``: BasicStem (7x7 s2 + FrozenBN + ReLU, max_pool2d 3/2/1) then res2..res5 of
    BottleneckBlocks with the stride on the 1x1 (STRIDE_IN_1X1, yaml :111)."""
    x = _conv(sd, "backbone.bottom_up.stem.conv1", x, 2, 3, True, True, cfg.bn_eps)
    x = F.max_pool2d(x, kernel_size=3, stride=2, padding=1)
    outs = {}
    cin, cout = cfg.stem_out, cfg.res2_out
    for si, nb in enumerate(cfg.blocks):
        for bi in range(nb):
            p = f"backbone.bottom_up.res{si + 2}.{bi}"
            s = 2 if (bi == 0 and si > 0) else 1
            s1, s3 = (s, 1) if cfg.stride_in_1x1 else (1, s)
            shortcut = _conv(sd, f"{p}.shortcut", x, s, 0, True, False, cfg.bn_eps) if cin != cout else x
            out = _conv(sd, f"{p}.conv1", x, s1, 0, True, True, cfg.bn_eps)
            out = _conv(sd, f"{p}.conv2", out, s3, 1, True, True, cfg.bn_eps)
            out = _conv(sd, f"{p}.conv3", out, 1, 0, True, False, cfg.bn_eps)
            x = F.relu(out + shortcut)
            cin = cout
        outs[f"res{si + 2}"] = x
        cout *= 2
    return outs


def fpn(sd, feats: Dict[str, torch.Tensor]) -> Dict[str, torch.Tensor]:
    """``FPN.forward`` (FUSE_TYPE sum, NORM '', yaml :62-70) + ``LastLevelMaxPool`` (p6 from p5)."""
    prev = _conv(sd, "backbone.fpn_lateral5", feats["res5"], norm=False)
    res = {"p5": _conv(sd, "backbone.fpn_output5", prev, 1, 1, norm=False)}
    for lvl in (4, 3, 2):
        top_down = F.interpolate(prev, scale_factor=2.0, mode="nearest")
        lateral = _conv(sd, f"backbone.fpn_lateral{lvl}", feats[f"res{lvl}"], norm=False)
        prev = lateral + top_down
        res[f"p{lvl}"] = _conv(sd, f"backbone.fpn_output{lvl}", prev, 1, 1, norm=False)
    res["p6"] = F.max_pool2d(res["p5"], kernel_size=1, stride=2, padding=0)
    return res


def rpn_head(sd, feats: Dict[str, torch.Tensor]):
    """``StandardRPNHead.forward`` over p2..p6 (yaml :233-238): shared 3x3 conv + ReLU, 1x1 objectness (A=3)
    and 1x1 anchor deltas (4A)."""
    obj, deltas = [], []
    for k in ("p2", "p3", "p4", "p5", "p6"):
        t = _conv(sd, "proposal_generator.rpn_head.conv", feats[k], 1, 1, norm=False, relu=True)
        obj.append(_conv(sd, "proposal_generator.rpn_head.objectness_logits", t, norm=False))
        deltas.append(_conv(sd, "proposal_generator.rpn_head.anchor_deltas", t, norm=False))
    return obj, deltas


def forward_front(sd, img_bgr_u8: torch.Tensor, cfg: RcnnCfg):
    """-> (dict p2..p6 NCHW, [objectness logits per level], [anchor deltas per level])."""
    with torch.no_grad():
        x = preprocess(img_bgr_u8, cfg)
        p = fpn(sd, bottom_up(sd, x, cfg))
        obj, deltas = rpn_head(sd, p)
    return p, obj, deltas


# =========================================================================================================
# Proposal generator and ROI heads (detectron2 v0.6: RPN.predict_proposals / find_top_rpn_proposals,
# DefaultAnchorGenerator, Box2BoxTransform, ROIPooler + ROIAlign(aligned=True), FastRCNNConvFCHead,
# FastRCNNOutputLayers.inference / fast_rcnn_inference_single_image, MaskRCNNConvUpsampleHead,
# mask_rcnn_inference, detector_postprocess, paste_masks_in_image), restated in plain torch on the CPU.
# Same "parity unpinned" status as the front end above.
# =========================================================================================================
import math


def cell_anchors(size: float, ratios):
    out = []
    for r in ratios:
        w = math.sqrt(size * size / r)
        h = r * w
        out.append([-w / 2.0, -h / 2.0, w / 2.0, h / 2.0])
    return torch.tensor(out, dtype=torch.float32)


def grid_anchors(hw, stride: int, size: float, ratios) -> torch.Tensor:
    """DefaultAnchorGenerator (offset 0): [(h*w*A), 4] ordered (y, x, a)."""
    h, w = hw
    sx = torch.arange(0, w * stride, step=stride, dtype=torch.float32)
    sy = torch.arange(0, h * stride, step=stride, dtype=torch.float32)
    yy, xx = torch.meshgrid(sy, sx, indexing="ij")
    shifts = torch.stack((xx.reshape(-1), yy.reshape(-1), xx.reshape(-1), yy.reshape(-1)), dim=1)
    return (shifts.view(-1, 1, 4) + cell_anchors(size, ratios).view(1, -1, 4)).reshape(-1, 4)


SCALE_CLAMP = math.log(1000.0 / 16)


def apply_deltas(deltas: torch.Tensor, boxes: torch.Tensor, weights) -> torch.Tensor:
    """Box2BoxTransform.apply_deltas: deltas [N, k*4], boxes [N, 4]."""
    deltas = deltas.float()
    boxes = boxes.to(deltas.dtype)
    widths = boxes[:, 2] - boxes[:, 0]
    heights = boxes[:, 3] - boxes[:, 1]
    ctr_x = boxes[:, 0] + 0.5 * widths
    ctr_y = boxes[:, 1] + 0.5 * heights
    wx, wy, ww, wh = weights
    dx = deltas[:, 0::4] / wx
    dy = deltas[:, 1::4] / wy
    dw = torch.clamp(deltas[:, 2::4] / ww, max=SCALE_CLAMP)
    dh = torch.clamp(deltas[:, 3::4] / wh, max=SCALE_CLAMP)
    pcx = dx * widths[:, None] + ctr_x[:, None]
    pcy = dy * heights[:, None] + ctr_y[:, None]
    pw = torch.exp(dw) * widths[:, None]
    ph = torch.exp(dh) * heights[:, None]
    out = torch.stack((pcx - 0.5 * pw, pcy - 0.5 * ph, pcx + 0.5 * pw, pcy + 0.5 * ph), dim=-1)
    return out.reshape(deltas.shape)


def clip_boxes(b: torch.Tensor, hw) -> torch.Tensor:
    h, w = hw
    return torch.stack((b[:, 0].clamp(0, w), b[:, 1].clamp(0, h), b[:, 2].clamp(0, w), b[:, 3].clamp(0, h)), dim=1)


def nms_sorted(boxes: torch.Tensor, cats: torch.Tensor, thr: float) -> torch.Tensor:
    """Greedy NMS of score-sorted boxes, suppression only within equal categories; returns keep mask."""
    n = boxes.shape[0]
    keep = torch.ones(n, dtype=torch.bool)
    area = (boxes[:, 2] - boxes[:, 0]) * (boxes[:, 3] - boxes[:, 1])
    for i in range(n):
        if not keep[i]:
            continue
        lt = torch.maximum(boxes[i, :2], boxes[i + 1:, :2])
        rb = torch.minimum(boxes[i, 2:], boxes[i + 1:, 2:])
        wh = (rb - lt).clamp(min=0)
        inter = wh[:, 0] * wh[:, 1]
        iou = inter / (area[i] + area[i + 1:] - inter)
        keep[i + 1:] &= ~((iou > thr) & (cats[i + 1:] == cats[i]))
    return keep


def batched_nms(boxes, scores, cats, thr):
    """torchvision.ops.batched_nms semantics: indices of kept boxes sorted by decreasing score."""
    order = torch.argsort(scores, descending=True, stable=True)
    keep = nms_sorted(boxes[order], cats[order], thr)
    return order[keep]


def rpn_proposals(obj: List[torch.Tensor], deltas: List[torch.Tensor], image_hw, cfg: RcnnCfg):
    """obj[l] [B,A,h,w], deltas[l] [B,4A,h,w] (NCHW as the modules emit them) -> per image (boxes, logits)."""
    B = obj[0].shape[0]
    strides = [4, 8, 16, 32, 64]
    sc, pr, lv = [], [], []
    for l, (o, d) in enumerate(zip(obj, deltas)):
        h, w = o.shape[2:]
        A = o.shape[1]
        anchors = grid_anchors((h, w), strides[l], cfg.anchor_sizes[l], cfg.aspect_ratios)
        logits = o.permute(0, 2, 3, 1).flatten(1)                                   # [B, h*w*A]
        dl = d.view(B, A, 4, h, w).permute(0, 3, 4, 1, 2).flatten(1, -2)            # [B, h*w*A, 4]
        props = apply_deltas(dl.reshape(-1, 4), anchors.unsqueeze(0).expand(B, -1, -1).reshape(-1, 4),
                             cfg.rpn_bbox_weights).view(B, -1, 4)
        k = min(logits.shape[1], cfg.rpn_pre_nms_topk)
        s, idx = logits.sort(descending=True, dim=1)
        sc.append(s[:, :k])
        pr.append(props[torch.arange(B)[:, None], idx[:, :k]])
        lv.append(torch.full((k,), l, dtype=torch.int64))
    sc, pr, lv = torch.cat(sc, 1), torch.cat(pr, 1), torch.cat(lv, 0)
    out = []
    for n in range(B):
        boxes, scores, lvl = pr[n], sc[n], lv
        valid = torch.isfinite(boxes).all(1) & torch.isfinite(scores)
        boxes, scores, lvl = boxes[valid], scores[valid], lvl[valid]
        boxes = clip_boxes(boxes, image_hw)
        ne = ((boxes[:, 2] - boxes[:, 0]) > 0) & ((boxes[:, 3] - boxes[:, 1]) > 0)
        boxes, scores, lvl = boxes[ne], scores[ne], lvl[ne]
        keep = batched_nms(boxes, scores, lvl, cfg.rpn_nms_thresh)[:cfg.rpn_post_nms_topk]
        out.append((boxes[keep], scores[keep]))
    return out


def assign_levels(boxes: torch.Tensor, min_level=2, max_level=5, canonical_box_size=224, canonical_level=4):
    sizes = torch.sqrt((boxes[:, 2] - boxes[:, 0]) * (boxes[:, 3] - boxes[:, 1]))
    lv = torch.floor(canonical_level + torch.log2(sizes / canonical_box_size + 1e-8))
    return torch.clamp(lv, min=min_level, max=max_level).to(torch.int64) - min_level


def _bilinear(feat, y, x):
    C, H, W = feat.shape
    if y < -1.0 or y > H or x < -1.0 or x > W:
        return torch.zeros(C)
    y, x = max(y, 0.0), max(x, 0.0)
    yl, xl = int(y), int(x)
    if yl >= H - 1:
        yh = yl = H - 1
        y = float(yl)
    else:
        yh = yl + 1
    if xl >= W - 1:
        xh = xl = W - 1
        x = float(xl)
    else:
        xh = xl + 1
    ly, lx = y - yl, x - xl
    hy, hx = 1.0 - ly, 1.0 - lx
    return hy * hx * feat[:, yl, xl] + hy * lx * feat[:, yl, xh] + ly * hx * feat[:, yh, xl] + ly * lx * feat[:, yh, xh]


def roi_align(feat: torch.Tensor, rois: torch.Tensor, scale: float, P: int, sampling_ratio=0, aligned=True):
    """ROIAlign (detectron2/torchvision kernel, loop form): feat [B,C,H,W], rois [N,5] -> [N,C,P,P]."""
    out = torch.zeros((rois.shape[0], feat.shape[1], P, P))
    off = 0.5 if aligned else 0.0
    for n, r in enumerate(rois.tolist()):
        f = feat[int(r[0])]
        sw, sh, ew, eh = (np.float32(r[1]) * np.float32(scale) - np.float32(off), np.float32(r[2]) * np.float32(scale) - np.float32(off),
                          np.float32(r[3]) * np.float32(scale) - np.float32(off), np.float32(r[4]) * np.float32(scale) - np.float32(off))
        rw, rh = np.float32(ew - sw), np.float32(eh - sh)
        if not aligned:
            rw, rh = max(rw, np.float32(1.0)), max(rh, np.float32(1.0))
        bh, bw = np.float32(rh / np.float32(P)), np.float32(rw / np.float32(P))
        gh = sampling_ratio if sampling_ratio > 0 else int(math.ceil(rh / np.float32(P)))
        gw = sampling_ratio if sampling_ratio > 0 else int(math.ceil(rw / np.float32(P)))
        count = max(gh * gw, 1)
        for ph in range(P):
            for pw in range(P):
                acc = torch.zeros(feat.shape[1])
                for iy in range(gh):
                    y = float(np.float32(sh + np.float32(ph) * bh + np.float32(iy + 0.5) * bh / np.float32(gh)))
                    for ix in range(gw):
                        x = float(np.float32(sw + np.float32(pw) * bw + np.float32(ix + 0.5) * bw / np.float32(gw)))
                        acc += _bilinear(f, y, x)
                out[n, :, ph, pw] = acc / count
    return out


import numpy as np  # noqa: E402  (used by roi_align's fp32 coordinate arithmetic)


def roi_align_vec(feat: torch.Tensor, rois: torch.Tensor, scale: float, P: int, sampling_ratio=0, aligned=True, chunk=64):
    """The same ROIAlign, vectorised over rois / bins / channels -- the arithmetic of ``roi_align`` above operation by
    operation (fp32 coordinates, weights formed in double and rounded to fp32, the four corner terms added left to right,
    samples accumulated in (iy, ix) order, one division by the sample count), so the two agree to the bit
    (tests/test_oracles_cpu.py); rois are grouped by their sampling grid (gh, gw).  For inputs the loop form cannot
    finish in reasonable time (1000 proposals per image)."""
    N, C = rois.shape[0], feat.shape[1]
    H, W = feat.shape[2:]
    out = torch.zeros((N, C, P, P))
    if N == 0:
        return out
    f32 = torch.float32
    r = rois.to(f32)
    sc, off, Pf = torch.tensor(scale, dtype=f32), torch.tensor(0.5 if aligned else 0.0, dtype=f32), torch.tensor(float(P), dtype=f32)
    sw, sh, ew, eh = r[:, 1] * sc - off, r[:, 2] * sc - off, r[:, 3] * sc - off, r[:, 4] * sc - off
    rw, rh = ew - sw, eh - sh
    if not aligned:
        rw, rh = rw.clamp(min=1.0), rh.clamp(min=1.0)
    bh, bw = rh / Pf, rw / Pf
    gh = torch.ceil(rh / Pf).to(torch.int64) if sampling_ratio <= 0 else torch.full((N,), sampling_ratio, dtype=torch.int64)
    gw = torch.ceil(rw / Pf).to(torch.int64) if sampling_ratio <= 0 else torch.full((N,), sampling_ratio, dtype=torch.int64)
    bidx = r[:, 0].to(torch.int64)
    flat = feat.permute(0, 2, 3, 1).reshape(feat.shape[0] * H * W, C)         # [B*H*W, C]
    pidx = torch.arange(P, dtype=f32)

    def axis(start, binsz, g, i, size):
        """coordinate of sample i of every bin along one axis -> (valid, low index, high index, l (double), h (double))"""
        gf = torch.tensor(float(g), dtype=f32)
        c = start[:, None] + pidx[None, :] * binsz[:, None] + (torch.tensor(i + 0.5, dtype=f32) * binsz / gf)[:, None]   # [n, P] fp32
        valid = ~((c < -1.0) | (c > size))
        c = c.clamp(min=0.0)
        lo = c.to(torch.int64)                                  # int(y) of a non-negative float
        top = lo >= size - 1
        lo = torch.where(top, torch.full_like(lo, size - 1), lo)
        hi = torch.where(top, lo, lo + 1)
        cd = torch.where(top, lo.to(torch.float64), c.to(torch.float64))
        l = cd - lo.to(torch.float64)
        return valid, lo, hi, l, 1.0 - l

    keys = gh * 100000 + gw
    for key in torch.unique(keys).tolist():
        sel_all = torch.nonzero(keys == key).flatten()
        g_h, g_w = int(key // 100000), int(key % 100000)
        count = max(g_h * g_w, 1)
        for c0 in range(0, len(sel_all), chunk):
            sel = sel_all[c0:c0 + chunk]
            n = len(sel)
            base = (bidx[sel] * (H * W))[:, None, None]
            acc = torch.zeros((n, P, P, C))
            for iy in range(g_h):
                vy, yl, yh, ly, hy = axis(sh[sel], bh[sel], g_h, iy, H)
                for ix in range(g_w):
                    vx, xl, xh, lx, hx = axis(sw[sel], bw[sel], g_w, ix, W)
                    valid = (vy[:, :, None] & vx[:, None, :])[..., None]          # [n, P, P, 1]

                    def corner(yi, xi, wy, wx):
                        v = flat[(base + yi[:, :, None] * W + xi[:, None, :]).reshape(-1)].view(n, P, P, C)
                        wgt = (wy[:, :, None] * wx[:, None, :]).to(f32)[..., None]  # double product, then fp32
                        return wgt * v
                    val = corner(yl, xl, hy, hx) + corner(yl, xh, hy, lx) + corner(yh, xl, ly, hx) + corner(yh, xh, ly, lx)
                    acc = acc + torch.where(valid, val, torch.zeros(()))
            out[sel] = (acc / count).permute(0, 3, 1, 2)
    return out


def roi_pool(pyr: Dict[str, torch.Tensor], rois: torch.Tensor, P: int, vectorised: bool = False) -> torch.Tensor:
    """ROIPooler over p2..p5 (scales 1/4..1/32), ROIAlignV2, sampling_ratio 0."""
    lv = assign_levels(rois[:, 1:])
    out = torch.zeros((rois.shape[0], pyr["p2"].shape[1], P, P))
    fn = roi_align_vec if vectorised else roi_align
    for l, k in enumerate(("p2", "p3", "p4", "p5")):
        inds = torch.nonzero(lv == l).flatten()
        if len(inds):
            out[inds] = fn(pyr[k], rois[inds], 1.0 / (4 * 2 ** l), P, 0, True)
    return out


def box_head(sd, feats: torch.Tensor):
    x = feats.flatten(1)
    x = F.relu(F.linear(x, sd["roi_heads.box_head.fc1.weight"], sd["roi_heads.box_head.fc1.bias"]))
    x = F.relu(F.linear(x, sd["roi_heads.box_head.fc2.weight"], sd["roi_heads.box_head.fc2.bias"]))
    scores = F.linear(x, sd["roi_heads.box_predictor.cls_score.weight"], sd["roi_heads.box_predictor.cls_score.bias"])
    deltas = F.linear(x, sd["roi_heads.box_predictor.bbox_pred.weight"], sd["roi_heads.box_predictor.bbox_pred.bias"])
    return scores, deltas


def fast_rcnn_inference_single_image(boxes, scores, image_hw, cfg: RcnnCfg):
    """boxes [R, K*4] (decoded), scores [R, K+1] (softmax) -> (boxes [n,4], scores [n], classes [n])."""
    valid = torch.isfinite(boxes).all(1) & torch.isfinite(scores).all(1)
    boxes, scores = boxes[valid], scores[valid]
    scores = scores[:, :-1]
    K = boxes.shape[1] // 4
    boxes = clip_boxes(boxes.reshape(-1, 4), image_hw).view(-1, K, 4)
    mask = scores > cfg.score_thresh_test
    inds = mask.nonzero()
    boxes, scores = boxes[mask], scores[mask]
    keep = batched_nms(boxes, scores, inds[:, 1], cfg.nms_thresh_test)[:cfg.detections_per_image]
    return boxes[keep], scores[keep], inds[keep][:, 1]


def mask_head(sd, feats: torch.Tensor, cfg: RcnnCfg):
    x = feats
    for i in range(cfg.num_mask_convs):
        x = F.relu(F.conv2d(x, sd[f"roi_heads.mask_head.mask_fcn{i + 1}.weight"], sd[f"roi_heads.mask_head.mask_fcn{i + 1}.bias"], padding=1))
    x = F.relu(F.conv_transpose2d(x, sd["roi_heads.mask_head.deconv.weight"], sd["roi_heads.mask_head.deconv.bias"], stride=2))
    return F.conv2d(x, sd["roi_heads.mask_head.predictor.weight"], sd["roi_heads.mask_head.predictor.bias"])


def paste_values(masks: torch.Tensor, boxes: torch.Tensor, hw) -> torch.Tensor:
    """_do_paste_mask (skip_empty=False): masks [N,M,M] probs resampled into the image, float [N,H,W]."""
    H, W = hw
    N = masks.shape[0]
    if N == 0:
        return torch.zeros((0, H, W), dtype=torch.float32)
    x0, y0, x1, y1 = torch.split(boxes, 1, dim=1)
    img_y = torch.arange(0, H, dtype=torch.float32) + 0.5
    img_x = torch.arange(0, W, dtype=torch.float32) + 0.5
    img_y = (img_y - y0) / (y1 - y0) * 2 - 1
    img_x = (img_x - x0) / (x1 - x0) * 2 - 1
    gx = img_x[:, None, :].expand(N, H, W)
    gy = img_y[:, :, None].expand(N, H, W)
    return F.grid_sample(masks[:, None].float(), torch.stack([gx, gy], dim=3), align_corners=False)[:, 0]


def paste_masks(masks: torch.Tensor, boxes: torch.Tensor, hw, thr: float) -> torch.Tensor:
    """paste_masks_in_image: probabilities -> bool [N,H,W] (threshold 0.5 in detector_postprocess)."""
    return paste_values(masks, boxes, hw) >= thr


def inference(sd, img_bgr_u8: torch.Tensor, cfg: RcnnCfg, vectorised: bool = False):
    """GeneralizedRCNN.inference + detector_postprocess for a batch; list of dicts (pred_boxes, scores,
    pred_classes, pred_masks bool [n,H,W]) at the ORIGINAL image resolution.  ``vectorised``: ROIAlign through
    ``roi_align_vec`` (same values; needed at the full 1000 proposals per image)."""
    B, H, W, _ = img_bgr_u8.shape
    nh, nw = resized_hw(H, W, cfg)
    pyr, obj, deltas = forward_front(sd, img_bgr_u8, cfg)
    results = []
    with torch.no_grad():
        props = rpn_proposals(obj, deltas, (nh, nw), cfg)
        for n in range(B):
            pb = props[n][0]
            rois = torch.cat([torch.full((len(pb), 1), float(n)), pb], 1)
            scores, dl = box_head(sd, roi_pool(pyr, rois, cfg.box_pooler_resolution, vectorised))
            boxes = apply_deltas(dl, pb, cfg.roi_bbox_weights)
            b, s, c = fast_rcnn_inference_single_image(boxes, F.softmax(scores, dim=-1), (nh, nw), cfg)
            rois = torch.cat([torch.full((len(b), 1), float(n)), b], 1)
            logits = mask_head(sd, roi_pool(pyr, rois, cfg.mask_pooler_resolution, vectorised), cfg)
            probs = logits[torch.arange(len(b)), c].sigmoid() if len(b) else logits[:, 0]
            sx, sy = W / nw, H / nh
            ob = b * torch.tensor([sx, sy, sx, sy])
            ob = clip_boxes(ob, (H, W))
            ne = ((ob[:, 2] - ob[:, 0]) > 0) & ((ob[:, 3] - ob[:, 1]) > 0)
            ob, s, c, probs = ob[ne], s[ne], c[ne], probs[ne]
            results.append(dict(pred_boxes=ob, scores=s, pred_classes=c, pred_masks=paste_masks(probs, ob, (H, W), cfg.mask_threshold),
                                proposals=pb, mask_probs=probs))
    return results
