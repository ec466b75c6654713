"""ORACLE (test infrastructure, not product code) -- CPU restatement, in plain fp32 PyTorch, of the
FRONT END of the Mask R-CNN the reference runs through detectron2's ``DefaultPredictor``
(nav/agent/utils/segmentation.py:31-38,45): test-time preprocessing, ResNet-101-FPN backbone and the
RPN head, as configured by nav/agent/utils/COCO-InstSeg/mask_rcnn_R_101_cat9.yaml.

**PARITY UNPINNED.**  The arithmetic lives in detectron2 (third party; peanut.Dockerfile:15 installs the
cu111/torch1.10 wheel => v0.6), which is not vendored, not installed, has no network route here, and whose
fine-tuned weights are a Drive link.  The reference has no test or golden vector at this boundary.  This
file therefore restates detectron2 v0.6's PUBLISHED module definitions (``DefaultPredictor.__call__``,
``ResizeShortestEdge``, ``GeneralizedRCNN.preprocess_image``, ``BasicStem``, ``BottleneckBlock``,
``FrozenBatchNorm2d``, ``FPN`` + ``LastLevelMaxPool``, ``StandardRPNHead``) under that yaml; it cannot be
validated against the reference itself.  One known approximation: detectron2 resizes the uint8 image with
PIL's fixed-point BILINEAR filter; here it is float bilinear (half-pixel centres) rounded back to uint8.
Only tests/, smoke() and bench baselines may import this module."""
from __future__ import annotations

from typing import Dict, List

import torch
import torch.nn.functional as F

from peanut_amd.rcnn_weights import RcnnCfg, padded_hw, resized_hw


def preprocess(img_bgr_u8: torch.Tensor, cfg: RcnnCfg) -> torch.Tensor:
    """uint8 [B,H,W,3] BGR -> float32 [B,3,Hp,Wp]: resize (yaml :28-30), (x - PIXEL_MEAN) / PIXEL_STD
    (:82-89), zero-pad to a multiple of 32 (ImageList.from_tensors with the FPN's size_divisibility)."""
    b, h, w, _ = img_bgr_u8.shape
    nh, nw = resized_hw(h, w, cfg)
    x = img_bgr_u8.permute(0, 3, 1, 2).float()
    x = F.interpolate(x, size=(nh, nw), mode="bilinear", align_corners=False)
    x = torch.floor(x + 0.5).clamp(0, 255)                                    # back to uint8 values (PIL output)
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
