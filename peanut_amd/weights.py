"""State-dict schema, seeded synthetic weights and mmcv-checkpoint reading for the
PEANUT map-prediction network (PSPNet: ResNet-50-V1c-D8 + PSP head).

The key names are the ones the reference model produces (mmcv / mmseg 0.26 layout):
``backbone.stem.*`` from ``prediction/mmseg/models/backbones/resnet.py:591-624``,
``backbone.layerL.i.{conv,bn}{1,2,3}`` / ``downsample.{0,1}`` from
``resnet.py:164-209`` and ``prediction/mmseg/models/utils/res_layer.py:43-95``,
``decode_head.psp_modules.k.1.{conv,bn}`` / ``decode_head.bottleneck.{conv,bn}``
from ``prediction/mmseg/models/decode_heads/psp_head.py:36-93`` and
``decode_head.conv_seg`` from ``decode_head.py:93``.  ``auxiliary_head.*`` keys
exist in real checkpoints (``nav/pred_model_cfg.py:29-40``) but are training-only.

Neither weight file of the reference ships with it (Google-Drive links only), so
tests and the benchmark use :func:`make_seeded_state_dict` -- a documented,
seed-reproducible recipe with non-trivial BatchNorm statistics.
"""
from __future__ import annotations

from collections import OrderedDict
from dataclasses import dataclass, field
from typing import Dict, Iterable, List, Tuple

import numpy as np
import torch


@dataclass(frozen=True)
class PredCfg:
    """The fields of ``nav/pred_model_cfg.py:2-42`` the inference path reads."""

    in_channels: int = 14
    num_classes: int = 6
    stem_channels: int = 64
    base_channels: int = 64
    stage_blocks: Tuple[int, ...] = (3, 4, 6, 3)
    strides: Tuple[int, ...] = (1, 2, 1, 1)
    dilations: Tuple[int, ...] = (1, 1, 2, 4)
    contract_dilation: bool = True
    pool_scales: Tuple[int, ...] = (1, 2, 3, 6)
    head_channels: int = 512
    align_corners: bool = False
    bn_eps: float = 1e-5

    @property
    def feat_channels(self) -> int:
        return self.base_channels * 8 * 4


def pred_cfg_from_file(path: str) -> PredCfg:
    """Read an mmcv-style python config (``nav/pred_model_cfg.py``) without mmcv.

    Mirrors ``Config.fromfile`` (``nav/agent/prediction.py:146``) for the keys
    the forward needs; the file is plain python assignments, so it is exec'd.
    """
    ns: Dict[str, object] = {}
    with open(path, "r") as f:
        exec(compile(f.read(), path, "exec"), ns)  # noqa: S102 - same trust model as mmcv
    model = ns["model"]
    bb, dh = model["backbone"], model["decode_head"]
    if model.get("type") != "EncoderDecoder" or bb.get("type") != "ResNetV1c" \
            or dh.get("type") != "PSPHead":
        raise ValueError("only EncoderDecoder(ResNetV1c, PSPHead) is supported, got "
                         f"{model.get('type')}({bb.get('type')}, {dh.get('type')})")
    if bb.get("depth") != 50:
        raise ValueError(f"only depth=50 is supported, got {bb.get('depth')}")
    mode = (model.get("test_cfg") or {}).get("mode", "whole")
    if mode != "whole":
        raise ValueError(f"only test_cfg.mode='whole' is supported, got {mode!r}")
    return PredCfg(
        in_channels=int(bb.get("in_channels", 3)),
        num_classes=int(dh["num_classes"]),
        strides=tuple(bb.get("strides", (1, 2, 2, 2))),
        dilations=tuple(bb.get("dilations", (1, 1, 1, 1))),
        contract_dilation=bool(bb.get("contract_dilation", False)),
        pool_scales=tuple(dh.get("pool_scales", (1, 2, 3, 6))),
        head_channels=int(dh["channels"]),
        align_corners=bool(dh.get("align_corners", False)),
    )


@dataclass
class ConvSpec:
    """One conv(+BN) on the inference path."""

    name: str          # state-dict prefix of the conv weight (without '.weight')
    bn: str            # state-dict prefix of its BatchNorm ('' -> none, conv has bias)
    cin: int
    cout: int
    k: int
    stride: int = 1
    pad: int = 0
    dil: int = 1
    relu: bool = True


@dataclass
class BlockSpec:
    conv1: ConvSpec
    conv2: ConvSpec
    conv3: ConvSpec
    down: ConvSpec | None


@dataclass
class NetSpec:
    stem: List[ConvSpec] = field(default_factory=list)
    layers: List[List[BlockSpec]] = field(default_factory=list)
    ppm: List[ConvSpec] = field(default_factory=list)
    bottleneck: ConvSpec | None = None
    conv_seg: ConvSpec | None = None


def net_spec(cfg: PredCfg) -> NetSpec:
    """Layer table of the inference path."""
    ns = NetSpec()
    sc = cfg.stem_channels
    ns.stem = [
        ConvSpec("backbone.stem.0", "backbone.stem.1", cfg.in_channels, sc // 2, 3, 2, 1),
        ConvSpec("backbone.stem.3", "backbone.stem.4", sc // 2, sc // 2, 3, 1, 1),
        ConvSpec("backbone.stem.6", "backbone.stem.7", sc // 2, sc, 3, 1, 1),
    ]
    inplanes = sc
    for li, nblocks in enumerate(cfg.stage_blocks):
        planes = cfg.base_channels * 2 ** li
        stride, dilation = cfg.strides[li], cfg.dilations[li]
        # res_layer.py:67-74: first block's dilation is halved under contract_dilation
        first_dil = dilation // 2 if (dilation > 1 and cfg.contract_dilation) else dilation
        blocks = []
        for bi in range(nblocks):
            p = f"backbone.layer{li + 1}.{bi}"
            s = stride if bi == 0 else 1
            d = first_dil if bi == 0 else dilation
            down = None
            if bi == 0 and (stride != 1 or inplanes != planes * 4):
                down = ConvSpec(f"{p}.downsample.0", f"{p}.downsample.1", inplanes,
                                planes * 4, 1, stride, 0, 1, relu=False)
            blocks.append(BlockSpec(
                ConvSpec(f"{p}.conv1", f"{p}.bn1", inplanes, planes, 1),
                ConvSpec(f"{p}.conv2", f"{p}.bn2", planes, planes, 3, s, d, d),
                ConvSpec(f"{p}.conv3", f"{p}.bn3", planes, planes * 4, 1, relu=False),
                down))
            inplanes = planes * 4
        ns.layers.append(blocks)
    hc = cfg.head_channels
    for i, _ in enumerate(cfg.pool_scales):
        p = f"decode_head.psp_modules.{i}.1"
        ns.ppm.append(ConvSpec(f"{p}.conv", f"{p}.bn", inplanes, hc, 1))
    ns.bottleneck = ConvSpec("decode_head.bottleneck.conv", "decode_head.bottleneck.bn",
                             inplanes + len(cfg.pool_scales) * hc, hc, 3, 1, 1)
    ns.conv_seg = ConvSpec("decode_head.conv_seg", "", hc, cfg.num_classes, 1, relu=False)
    return ns


def iter_convs(ns: NetSpec) -> Iterable[ConvSpec]:
    yield from ns.stem
    for blocks in ns.layers:
        for b in blocks:
            yield b.conv1
            yield b.conv2
            yield b.conv3
            if b.down is not None:
                yield b.down
    yield from ns.ppm
    yield ns.bottleneck
    yield ns.conv_seg


def inference_keys(cfg: PredCfg) -> List[Tuple[str, Tuple[int, ...]]]:
    """(key, shape) of every tensor the inference path needs, in layer order."""
    out: List[Tuple[str, Tuple[int, ...]]] = []
    for c in iter_convs(net_spec(cfg)):
        out.append((f"{c.name}.weight", (c.cout, c.cin, c.k, c.k)))
        if c.bn:
            for s in ("weight", "bias", "running_mean", "running_var"):
                out.append((f"{c.bn}.{s}", (c.cout,)))
        else:
            out.append((f"{c.name}.bias", (c.cout,)))
    return out


def conv_flops_per_map(cfg: PredCfg, h: int, w: int) -> float:
    """2 x MACs of every conv on the inference path for one HxW map."""

    def o(n, k, s, p, d):
        return (n + 2 * p - d * (k - 1) - 1) // s + 1

    macs = 0
    ns = net_spec(cfg)
    ch, cw = h, w
    for c in ns.stem:
        ch, cw = o(ch, c.k, c.stride, c.pad, c.dil), o(cw, c.k, c.stride, c.pad, c.dil)
        macs += ch * cw * c.cout * c.cin * c.k * c.k
    ch, cw = o(ch, 3, 2, 1, 1), o(cw, 3, 2, 1, 1)  # maxpool resnet.py:638
    for blocks in ns.layers:
        for b in blocks:
            macs += ch * cw * b.conv1.cout * b.conv1.cin
            oh = o(ch, 3, b.conv2.stride, b.conv2.pad, b.conv2.dil)
            ow = o(cw, 3, b.conv2.stride, b.conv2.pad, b.conv2.dil)
            macs += oh * ow * b.conv2.cout * b.conv2.cin * 9
            macs += oh * ow * b.conv3.cout * b.conv3.cin
            if b.down is not None:
                macs += oh * ow * b.down.cout * b.down.cin
            ch, cw = oh, ow
    for c, k in zip(ns.ppm, cfg.pool_scales):
        macs += k * k * c.cout * c.cin
    macs += ch * cw * ns.bottleneck.cout * ns.bottleneck.cin * 9
    macs += ch * cw * ns.conv_seg.cout * ns.conv_seg.cin
    return 2.0 * macs


def make_seeded_state_dict(cfg: PredCfg = PredCfg(), seed: int = 0,
                           with_aux: bool = False) -> "OrderedDict[str, torch.Tensor]":
    """Seed-reproducible synthetic weights with non-trivial BN statistics.

    Recipe (restated so that it does not depend on the reference
    ctor): conv weights ~ N(0, 2/fan_in) (He), BN ``weight ~ U(0.75, 1.25)``,
    ``bias ~ N(0, 0.1)``, ``running_mean ~ N(0, 0.1)``, ``running_var ~ U(0.75, 1.25)``,
    ``conv_seg.bias ~ N(0, 0.1)``.  The last BN of each bottleneck block gets
    ``weight ~ U(0.2, 0.3)`` so the residual trunk stays O(1) through 16 blocks.
    One ``torch.Generator(seed)`` is consumed in :func:`inference_keys` order.
    """
    g = torch.Generator().manual_seed(int(seed))
    sd: "OrderedDict[str, torch.Tensor]" = OrderedDict()

    def randn(shape, std):
        return torch.randn(shape, generator=g, dtype=torch.float32) * std

    def uni(shape, lo, hi):
        return torch.rand(shape, generator=g, dtype=torch.float32) * (hi - lo) + lo

    def add_conv(c: ConvSpec):
        fan_in = c.cin * c.k * c.k
        sd[f"{c.name}.weight"] = randn((c.cout, c.cin, c.k, c.k), (2.0 / fan_in) ** 0.5)
        if c.bn:
            last = c.bn.endswith(".bn3")
            sd[f"{c.bn}.weight"] = uni((c.cout,), 0.2, 0.3) if last else uni((c.cout,), 0.75, 1.25)
            sd[f"{c.bn}.bias"] = randn((c.cout,), 0.1)
            sd[f"{c.bn}.running_mean"] = randn((c.cout,), 0.1)
            sd[f"{c.bn}.running_var"] = uni((c.cout,), 0.75, 1.25)
            sd[f"{c.bn}.num_batches_tracked"] = torch.tensor(0, dtype=torch.int64)
        else:
            sd[f"{c.name}.bias"] = randn((c.cout,), 0.1)

    for c in iter_convs(net_spec(cfg)):
        add_conv(c)
    if with_aux:  # nav/pred_model_cfg.py:29-40 -- present in checkpoints, unused at inference
        add_conv(ConvSpec("auxiliary_head.convs.0.conv", "auxiliary_head.convs.0.bn",
                          cfg.base_channels * 16, 256, 3, 1, 1))
        add_conv(ConvSpec("auxiliary_head.conv_seg", "", 256, cfg.num_classes, 1))
    return sd


def load_mmcv_checkpoint(path: str) -> Tuple["OrderedDict[str, torch.Tensor]", dict]:
    """Read an mmcv checkpoint the way ``load_checkpoint(model, path, map_location='cpu')``
    + ``checkpoint['meta']['CLASSES']`` do (``prediction/mmseg/apis/inference.py:33-35``).

    Accepts ``{'meta':..., 'state_dict':...}`` or a bare state dict, strips an optional
    ``module.`` prefix (mmcv ``load_state_dict`` convention) and returns (state_dict, meta).
    """
    ckpt = torch.load(path, map_location="cpu", weights_only=False)
    meta = {}
    if isinstance(ckpt, dict) and "state_dict" in ckpt:
        meta = ckpt.get("meta", {}) or {}
        sd = ckpt["state_dict"]
    else:
        sd = ckpt
    out: "OrderedDict[str, torch.Tensor]" = OrderedDict()
    for k, v in sd.items():
        if k.startswith("module."):
            k = k[len("module."):]
        out[k] = v
    return out, meta


def select_inference_tensors(sd: Dict[str, torch.Tensor], cfg: PredCfg
                             ) -> List[Tuple[str, np.ndarray]]:
    """Validate ``sd`` against the schema and return contiguous fp32 arrays in
    :func:`inference_keys` order; ``auxiliary_head.*`` and ``num_batches_tracked``
    are ignored, anything missing or mis-shaped raises ``KeyError``/``ValueError``
    (strict like ``load_checkpoint`` would report it)."""
    out = []
    for key, shape in inference_keys(cfg):
        if key not in sd:
            raise KeyError(f"checkpoint is missing '{key}'")
        t = sd[key]
        if tuple(t.shape) != tuple(shape):
            raise ValueError(f"'{key}' has shape {tuple(t.shape)}, expected {shape}")
        out.append((key, np.ascontiguousarray(t.detach().to(torch.float32).cpu().numpy())))
    return out
