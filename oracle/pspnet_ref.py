"""ORACLE (test infrastructure, not product code) -- CPU restatement of the reference's
map-prediction forward in plain fp32 PyTorch, without mmcv/mmseg.

Only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg may
import this module; the product path (``peanut_amd``) never does and fails loudly when
its HIP library is missing.

Pinned by: ``oracle/gen_golden.py`` runs the reference's own model files
(``/root/reference/prediction/mmseg/...`` built from ``nav/pred_model_cfg.py``) on seeded
weights/inputs in the build container, asserts this restatement equals them (<=1e-5
max-abs), and commits the outputs under ``tests/golden/``.  The reference holds no numeric
golden vector of its own for this path, so those vectors are the pin.

Each function cites the reference lines it follows (paths relative to /root/reference).
"""
from __future__ import annotations

from typing import Dict, List

import numpy as np
import torch
import torch.nn.functional as F

from peanut_amd.weights import PredCfg, net_spec, ConvSpec  # schema only (no compute)


def _conv_bn_act(x: torch.Tensor, sd: Dict[str, torch.Tensor], c: ConvSpec, eps: float,
                 relu: bool | None = None) -> torch.Tensor:
    """mmcv ``ConvModule`` / ``build_conv_layer``+``build_norm_layer`` wiring: Conv2d(bias =
    not with_norm) -> BatchNorm2d(eps=1e-5, eval) -> ReLU  (call sites
    prediction/mmseg/models/decode_heads/psp_head.py:39-46,86-93;
    prediction/mmseg/models/backbones/resnet.py:164-209,595-623)."""
    bias = None if c.bn else sd[f"{c.name}.bias"]
    y = F.conv2d(x, sd[f"{c.name}.weight"], bias, stride=c.stride, padding=c.pad, dilation=c.dil)
    if c.bn:
        y = F.batch_norm(y, sd[f"{c.bn}.running_mean"], sd[f"{c.bn}.running_var"],
                         sd[f"{c.bn}.weight"], sd[f"{c.bn}.bias"], training=False, eps=eps)
    if c.relu if relu is None else relu:
        y = F.relu(y)
    return y


def backbone_forward(sd, x: torch.Tensor, cfg: PredCfg) -> List[torch.Tensor]:
    """``ResNetV1c.forward`` (resnet.py:659-674): deep stem (resnet.py:591-624), MaxPool2d(3,2,1)
    (:638), then four ``ResLayer``s of ``Bottleneck`` blocks (resnet.py:267-307)."""
    ns = net_spec(cfg)
    for c in ns.stem:
        x = _conv_bn_act(x, sd, c, cfg.bn_eps)
    x = F.max_pool2d(x, kernel_size=3, stride=2, padding=1)
    outs = []
    for blocks in ns.layers:
        for b in blocks:
            identity = x
            out = _conv_bn_act(x, sd, b.conv1, cfg.bn_eps)
            out = _conv_bn_act(out, sd, b.conv2, cfg.bn_eps)
            out = _conv_bn_act(out, sd, b.conv3, cfg.bn_eps)        # BN only (resnet.py:289-290)
            if b.down is not None:
                identity = _conv_bn_act(x, sd, b.down, cfg.bn_eps)  # res_layer.py:55-64
            x = F.relu(out + identity)                              # resnet.py:298,305
        outs.append(x)
    return outs


def ppm_forward(sd, x: torch.Tensor, cfg: PredCfg) -> List[torch.Tensor]:
    """``PPM.forward`` (psp_head.py:48-59): AdaptiveAvgPool2d(k) -> 1x1 ConvModule -> bilinear
    resize to x's size with ``align_corners`` from the head cfg (False)."""
    ns = net_spec(cfg)
    outs = []
    for c, k in zip(ns.ppm, cfg.pool_scales):
        p = F.adaptive_avg_pool2d(x, k)
        p = _conv_bn_act(p, sd, c, cfg.bn_eps)
        p = F.interpolate(p, size=x.shape[2:], mode="bilinear", align_corners=cfg.align_corners)
        outs.append(p)
    return outs


def decode_head_forward(sd, feats: List[torch.Tensor], cfg: PredCfg) -> torch.Tensor:
    """``PSPHead.forward`` (psp_head.py:95-117): select in_index=3 (decode_head.py:154-179),
    cat [x, ppm...] (order matters), 3x3 bottleneck, ``cls_seg`` (decode_head.py:225-230;
    Dropout2d is an eval no-op)."""
    ns = net_spec(cfg)
    x = feats[3]
    cat = torch.cat([x] + ppm_forward(sd, x, cfg), dim=1)
    y = _conv_bn_act(cat, sd, ns.bottleneck, cfg.bn_eps)
    return _conv_bn_act(y, sd, ns.conv_seg, cfg.bn_eps)


def encode_decode(sd, x: torch.Tensor, cfg: PredCfg) -> torch.Tensor:
    """``EncoderDecoder.encode_decode`` (encoder_decoder.py:70-80) followed by
    ``whole_inference`` with rescale=True (encoder_decoder.py:203-223): the slice to
    img_shape and the second resize to ori_shape are identities here because the test
    pipeline is an identity.  The fork returns RAW LOGITS -- no softmax
    (encoder_decoder.py:248) and no argmax (:262-271)."""
    feats = backbone_forward(sd, x, cfg)
    out = decode_head_forward(sd, feats, cfg)
    out = F.interpolate(out, size=x.shape[2:], mode="bilinear", align_corners=cfg.align_corners)
    # whole_inference: resize(seg_logit, size=ori_shape) at the same size is an exact identity
    # for bilinear align_corners=False (source coordinate == destination coordinate).
    return out


def taps(sd, x: torch.Tensor, cfg: PredCfg) -> Dict[str, torch.Tensor]:
    """Per-stage intermediate tensors for kernel-level bisecting (NCHW)."""
    ns = net_spec(cfg)
    t: Dict[str, torch.Tensor] = {}
    y = x
    for i, c in enumerate(ns.stem):
        y = _conv_bn_act(y, sd, c, cfg.bn_eps)
        t[f"stem{i}"] = y
    t["pool"] = F.max_pool2d(y, 3, 2, 1)
    feats = backbone_forward(sd, x, cfg)
    for i, f in enumerate(feats):
        t[f"layer{i + 1}"] = f
    f4 = feats[3]
    table = []
    for c, k in zip(ns.ppm, cfg.pool_scales):
        p = _conv_bn_act(F.adaptive_avg_pool2d(f4, k), sd, c, cfg.bn_eps)
        table.append(p.flatten(2))                       # [B,512,k*k]
    t["ppm_table"] = torch.cat(table, dim=2)             # [B,512,50]
    cat = torch.cat([f4] + ppm_forward(sd, f4, cfg), dim=1)
    bt = _conv_bn_act(cat, sd, ns.bottleneck, cfg.bn_eps)
    t["bottleneck"] = bt
    t["logits_lowres"] = _conv_bn_act(bt, sd, ns.conv_seg, cfg.bn_eps)
    return t


def run_inference(sd, full_map: np.ndarray, cfg: PredCfg) -> List[np.ndarray]:
    """``run_inference`` (nav/agent/prediction.py:112-137): MapFromArray CHW->HWC float32
    (:51-52), MultiScaleFlipAug(ratio 1.0, no flip) + Resize(keep_ratio) at the same size +
    ImageToTensor HWC->CHW (test_time_aug.py:113-135, transforms.py:267-284,
    formatting.py:94-98) = identity, no normalisation; collate to batch 1; forward with
    return_loss=False, rescale=True; ``simple_test`` returns list(np.ndarray)
    (encoder_decoder.py:260-271)."""
    img = np.ascontiguousarray(full_map.transpose(1, 2, 0).astype(np.float32))    # MapFromArray
    x = torch.from_numpy(img.transpose(2, 0, 1).copy())[None]                     # ImageToTensor
    with torch.no_grad():
        y = encode_decode(sd, x, cfg)
    return list(y.cpu().numpy())


def get_prediction(sd, full_map: np.ndarray, cfg: PredCfg) -> np.ndarray:
    """``PEANUT_Prediction_Model.get_prediction`` (prediction.py:155-158):
    ``scipy.special.expit(run_inference(...)[0])``."""
    from scipy.special import expit
    return expit(run_inference(sd, full_map, cfg)[0])


def forward_batch(sd, x: torch.Tensor, cfg: PredCfg, sigmoid: bool = False) -> torch.Tensor:
    """Batched logits (``simple_test`` accepts N>1, base.py:83-89)."""
    with torch.no_grad():
        y = encode_decode(sd, x, cfg)
        return torch.sigmoid(y) if sigmoid else y
