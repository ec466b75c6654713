"""Torch-glue form of the Mask R-CNN ROI stages, for BISECTING the HIP detector in tests (not product code).

``peanut_amd.rcnn.MaskRCNN.inference`` is one C entry (``peanut_rcnn_inference``: every selection stage a HIP kernel).  This
subclass runs the same pipeline with the selection stages as torch ops around the library's operator exports
(``peanut_nms`` / ``peanut_roi_align`` / ``peanut_paste_masks`` and ``FusedConv`` for the head layers) -- the round-1 form,
kept so that a stage test can feed each stage the oracle's upstream tensors and so that
``test_c_entry_matches_the_stagewise_glue`` can compare the two end to end."""
import math
from typing import List

import torch

from peanut_amd.ops import FusedConv
from peanut_amd.rcnn import MaskRCNN, nms_keep_segments, paste_masks, roi_align_pyramid
from peanut_amd.rcnn_weights import RcnnCfg

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


def assign_levels(boxes, min_level=2, max_level=5, canonical_box_size=224, canonical_level=4):
    sizes = torch.sqrt((boxes[:, 2] - boxes[:, 0]) * (boxes[:, 3] - boxes[:, 1]))
    lv = torch.floor(canonical_level + torch.log2(sizes / canonical_box_size + 1e-8))
    return torch.clamp(lv, min=min_level, max=max_level).to(torch.int64) - min_level



class GlueMaskRCNN(MaskRCNN):
    def __init__(self, cfg: RcnnCfg, state_dict, device="cuda:0", precision: str = "fp32", conv_algo: str = "auto"):
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

