"""State-dict schema and seeded synthetic weights of the Mask R-CNN front end (ResNet-101-FPN backbone +
RPN head) that ``SemanticPredMaskRCNN`` builds through detectron2's ``DefaultPredictor``
(nav/agent/utils/segmentation.py:30-38) from ``COCO-InstSeg/mask_rcnn_R_101_cat9.yaml``.

detectron2 is a third-party dependency of the reference (peanut.Dockerfile:15, the cu111/torch1.10
wheel index => v0.6); it is neither vendored nor installed here and its weights are a Drive link, so
the key names below follow detectron2 v0.6's published module layout (``build_resnet_fpn_backbone``,
``BasicStem``, ``BottleneckBlock``, ``FPN``, ``StandardRPNHead``) as configured by that yaml."""
from __future__ import annotations

from collections import OrderedDict
from dataclasses import dataclass
from typing import List, Tuple

import torch


@dataclass(frozen=True)
class RcnnCfg:
    """mask_rcnn_R_101_cat9.yaml fields the front end reads."""
    depth: int = 101                                   # RESNETS.DEPTH (:103)
    stem_out: int = 64                                 # RESNETS.STEM_OUT_CHANNELS (:110)
    res2_out: int = 256                                # RESNETS.RES2_OUT_CHANNELS (:108)
    stride_in_1x1: bool = True                         # RESNETS.STRIDE_IN_1X1 (:111)
    fpn_out: int = 256                                 # FPN.OUT_CHANNELS (:70)
    num_anchors: int = 3                               # ANCHOR_GENERATOR.ASPECT_RATIOS (:46-49), one size per level
    min_size: int = 800                                # INPUT.MIN_SIZE_TEST (:30)
    max_size: int = 1333                               # INPUT.MAX_SIZE_TEST (:28)
    size_divisibility: int = 32                        # FPN backbone
    pixel_mean: Tuple[float, float, float] = (103.53, 116.28, 123.675)   # BGR (:82-85)
    pixel_std: Tuple[float, float, float] = (1.0, 1.0, 1.0)              # (:86-89)
    bn_eps: float = 1e-5                               # FrozenBatchNorm2d default
    # proposal generator (RPN, :224-253) and anchors (:41-57)
    anchor_sizes: Tuple[int, ...] = (32, 64, 128, 256, 512)
    aspect_ratios: Tuple[float, ...] = (0.5, 1.0, 2.0)
    rpn_pre_nms_topk: int = 1000
    rpn_post_nms_topk: int = 1000
    rpn_nms_thresh: float = 0.7
    rpn_bbox_weights: Tuple[float, ...] = (1.0, 1.0, 1.0, 1.0)
    # ROI heads (:163-223, :312)
    num_classes: int = 9
    box_pooler_resolution: int = 7
    mask_pooler_resolution: int = 14
    fc_dim: int = 1024
    mask_conv_dim: int = 256
    num_mask_convs: int = 4
    roi_bbox_weights: Tuple[float, ...] = (10.0, 10.0, 5.0, 5.0)
    score_thresh_test: float = 0.05
    nms_thresh_test: float = 0.5
    detections_per_image: int = 100
    mask_threshold: float = 0.5                        # detector_postprocess default

    @property
    def blocks(self) -> Tuple[int, int, int, int]:
        return {50: (3, 4, 6, 3), 101: (3, 4, 23, 3), 152: (3, 8, 36, 3)}[self.depth]


@dataclass
class RConv:
    name: str
    cin: int
    cout: int
    k: int
    stride: int = 1
    pad: int = 0
    norm: bool = True      # FrozenBN (True) or plain bias (False)
    relu: bool = False


def backbone_convs(cfg: RcnnCfg) -> List[RConv]:
    """Every conv of stem + res2..res5 in forward order (shortcut first inside a block)."""
    out = [RConv("backbone.bottom_up.stem.conv1", 3, cfg.stem_out, 7, 2, 3, True, True)]
    cin, bott, cout = cfg.stem_out, cfg.res2_out // 4, cfg.res2_out
    for si, nb in enumerate(cfg.blocks):
        for bi in range(nb):
            p = f"backbone.bottom_up.res{si + 2}.{bi}"
            s = 2 if (bi == 0 and si > 0) else 1
            s1, s3 = (s, 1) if cfg.stride_in_1x1 else (1, s)
            if cin != cout:
                out.append(RConv(f"{p}.shortcut", cin, cout, 1, s, 0, True, False))
            out.append(RConv(f"{p}.conv1", cin, bott, 1, s1, 0, True, True))
            out.append(RConv(f"{p}.conv2", bott, bott, 3, s3, 1, True, True))
            out.append(RConv(f"{p}.conv3", bott, cout, 1, 1, 0, True, False))
            cin = cout
        bott, cout = bott * 2, cout * 2
    return out


def head_convs(cfg: RcnnCfg) -> List[RConv]:
    out = []
    chans = [cfg.res2_out * 2 ** i for i in range(4)]
    for lvl, c in zip((2, 3, 4, 5), chans):
        out.append(RConv(f"backbone.fpn_lateral{lvl}", c, cfg.fpn_out, 1, 1, 0, False, False))
        out.append(RConv(f"backbone.fpn_output{lvl}", cfg.fpn_out, cfg.fpn_out, 3, 1, 1, False, False))
    out.append(RConv("proposal_generator.rpn_head.conv", cfg.fpn_out, cfg.fpn_out, 3, 1, 1, False, True))
    out.append(RConv("proposal_generator.rpn_head.objectness_logits", cfg.fpn_out, cfg.num_anchors, 1, 1, 0, False, False))
    out.append(RConv("proposal_generator.rpn_head.anchor_deltas", cfg.fpn_out, cfg.num_anchors * 4, 1, 1, 0, False, False))
    return out


def roi_head_keys(cfg: RcnnCfg) -> List[Tuple[str, Tuple[int, ...]]]:
    """StandardROIHeads: FastRCNNConvFCHead (NUM_FC 2), FastRCNNOutputLayers, MaskRCNNConvUpsampleHead."""
    p, f, k = cfg.box_pooler_resolution, cfg.fc_dim, cfg.num_classes
    keys = [("roi_heads.box_head.fc1.weight", (f, cfg.fpn_out * p * p)), ("roi_heads.box_head.fc1.bias", (f,)),
            ("roi_heads.box_head.fc2.weight", (f, f)), ("roi_heads.box_head.fc2.bias", (f,)),
            ("roi_heads.box_predictor.cls_score.weight", (k + 1, f)), ("roi_heads.box_predictor.cls_score.bias", (k + 1,)),
            ("roi_heads.box_predictor.bbox_pred.weight", (4 * k, f)), ("roi_heads.box_predictor.bbox_pred.bias", (4 * k,))]
    c = cfg.mask_conv_dim
    cin = cfg.fpn_out
    for i in range(cfg.num_mask_convs):
        keys += [(f"roi_heads.mask_head.mask_fcn{i + 1}.weight", (c, cin, 3, 3)), (f"roi_heads.mask_head.mask_fcn{i + 1}.bias", (c,))]
        cin = c
    keys += [("roi_heads.mask_head.deconv.weight", (c, c, 2, 2)), ("roi_heads.mask_head.deconv.bias", (c,)),
             ("roi_heads.mask_head.predictor.weight", (k, c, 1, 1)), ("roi_heads.mask_head.predictor.bias", (k,))]
    return keys


def front_keys(cfg: RcnnCfg) -> List[Tuple[str, Tuple[int, ...]]]:
    keys = []
    for c in backbone_convs(cfg) + head_convs(cfg):
        keys.append((f"{c.name}.weight", (c.cout, c.cin, c.k, c.k)))
        if c.norm:
            for s in ("weight", "bias", "running_mean", "running_var"):
                keys.append((f"{c.name}.norm.{s}", (c.cout,)))
        else:
            keys.append((f"{c.name}.bias", (c.cout,)))
    return keys


def make_seeded_rcnn_state_dict(cfg: RcnnCfg = RcnnCfg(), seed: int = 0) -> "OrderedDict[str, torch.Tensor]":
    """He-normal conv weights (stem scaled by 1/64 for pixel-valued inputs, FPN/RPN by 1/2), FrozenBN weight ~ U(0.75,1.25) (U(0.2,0.3) on each block's conv3 so the
    33-block trunk stays O(1)), bias / mean ~ N(0, 0.1), var ~ U(0.75, 1.25); one generator in key order."""
    g = torch.Generator().manual_seed(int(seed))
    sd: "OrderedDict[str, torch.Tensor]" = OrderedDict()
    for c in backbone_convs(cfg) + head_convs(cfg):
        fan_in = c.cin * c.k * c.k
        std = (2.0 / fan_in) ** 0.5
        if c.name.endswith("stem.conv1"):
            std *= 1.0 / 64.0          # inputs are mean-subtracted 8-bit pixel values (|x| ~ 64)
        if "fpn_" in c.name or "rpn_head" in c.name:
            std *= 0.5                 # no norm layers in the FPN / RPN head: keep the pyramid O(1)
        sd[f"{c.name}.weight"] = torch.randn((c.cout, c.cin, c.k, c.k), generator=g) * std
        if c.norm:
            lo, hi = (0.2, 0.3) if c.name.endswith("conv3") else (0.75, 1.25)
            sd[f"{c.name}.norm.weight"] = torch.rand((c.cout,), generator=g) * (hi - lo) + lo
            sd[f"{c.name}.norm.bias"] = torch.randn((c.cout,), generator=g) * 0.1
            sd[f"{c.name}.norm.running_mean"] = torch.randn((c.cout,), generator=g) * 0.1
            sd[f"{c.name}.norm.running_var"] = torch.rand((c.cout,), generator=g) * 0.5 + 0.75
        else:
            sd[f"{c.name}.bias"] = torch.randn((c.cout,), generator=g) * 0.1
    for key, shape in roi_head_keys(cfg):
        if key.endswith(".bias"):
            sd[key] = torch.randn(shape, generator=g) * 0.1
        else:
            fan_in = 1
            for d in shape[1:]:
                fan_in *= d
            std = (2.0 / fan_in) ** 0.5
            if "bbox_pred" in key or "anchor" in key:
                std *= 0.2             # small box refinements
            if "cls_score" in key:
                std *= 0.35            # keep the softmax un-saturated: distinct scores, no ties in the sorts
            if "mask_head.predictor" in key:
                std *= 2.0             # mask logits of both signs
            sd[key] = torch.randn(shape, generator=g) * std
    return sd


def resized_hw(h: int, w: int, cfg: RcnnCfg) -> Tuple[int, int]:
    """detectron2 ``ResizeShortestEdge.get_output_shape`` (short edge -> min_size, long edge capped at
    max_size, round half up)."""
    size = cfg.min_size * 1.0
    scale = size / min(h, w)
    newh, neww = (size, scale * w) if h < w else (scale * h, size)
    if max(newh, neww) > cfg.max_size:
        s = cfg.max_size * 1.0 / max(newh, neww)
        newh, neww = newh * s, neww * s
    return int(newh + 0.5), int(neww + 0.5)


def padded_hw(h: int, w: int, cfg: RcnnCfg) -> Tuple[int, int]:
    d = cfg.size_divisibility
    return (h + d - 1) // d * d, (w + d - 1) // d * d


def load_detectron2_checkpoint(path: str) -> "OrderedDict[str, torch.Tensor]":
    """Read the file ``cfg.MODEL.WEIGHTS`` names (segmentation.py:34, arguments.py:32): a torch pickle
    holding either ``{'model': {...}}`` (DetectionCheckpointer's layout; values may be numpy arrays for
    converted model-zoo files) or a bare state dict.  Returns fp32 CPU tensors under detectron2's
    parameter names (``backbone.bottom_up.*``, ``proposal_generator.*``, ``roi_heads.*``)."""
    import numpy as np
    blob = torch.load(path, map_location="cpu", weights_only=False)
    sd = blob["model"] if isinstance(blob, dict) and "model" in blob and isinstance(blob["model"], dict) else blob
    out = OrderedDict()
    for k, v in sd.items():
        if isinstance(v, np.ndarray):
            v = torch.from_numpy(v)
        if torch.is_tensor(v) and v.dtype.is_floating_point:
            out[k[7:] if k.startswith("module.") else k] = v.float().contiguous()
    return out
