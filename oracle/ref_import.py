"""ORACLE support (test infrastructure) -- import the reference's OWN python sources from
/root/reference in the build container, so golden vectors can be generated from them.

* Stage 2 (``nav/agent/mapping.py`` + ``utils/depth_utils.py`` + ``utils/model.py``) imports
  unmodified: it needs only torch / numpy / matplotlib.
* Stage 3 (the mmseg fork) cannot ``import mmseg`` because ``mmcv`` (mmcv-full==1.6.0,
  peanut.Dockerfile:18) is not installed and there is no network.  The ten model source files
  on the inference path are loaded UNMODIFIED BY FILE PATH under a minimal ``mmcv`` stand-in
  that only wires ``torch.nn`` modules the way mmcv 1.6.0 documents it: ``build_conv_layer``
  -> ``nn.Conv2d``, ``build_norm_layer`` -> ``('bn'+postfix, nn.BatchNorm2d(eps=1e-5))``,
  ``ConvModule`` = conv(bias = not with_norm) -> bn -> ReLU, ``Registry.build`` = look up
  ``type`` and call the class.  Every multiply-add is therefore executed by the reference's
  own ``forward`` code on torch's CPU kernels; the stand-in contributes no arithmetic.
  DESIGN.md states this limitation of the pin.

This module reads /root/reference and therefore never runs on the GPU box: nothing under
``tests/ -m gpu``, ``smoke()`` or ``bench.py`` imports it.
"""
from __future__ import annotations

import importlib.util
import os
import sys
import types

import torch
import torch.nn as nn

REF = os.environ.get("PEANUT_REFERENCE", "/root/reference")
MMSEG = os.path.join(REF, "prediction", "mmseg")


def reference_available() -> bool:
    return os.path.isdir(MMSEG) and os.path.isdir(os.path.join(REF, "nav"))


# --------------------------------------------------------------------------------------
# mmcv stand-in (wiring only)
# --------------------------------------------------------------------------------------
class _Registry:
    def __init__(self, name, parent=None, **_):
        self.name, self.parent, self._mods = name, parent, {}

    def register_module(self, name=None, force=False, module=None):
        if isinstance(name, type):      # used as bare decorator
            self._mods[name.__name__] = name
            return name

        def deco(cls):
            self._mods[name or cls.__name__] = cls
            return cls
        return deco

    def get(self, key):
        r = self
        while r is not None:
            if key in r._mods:
                return r._mods[key]
            r = r.parent
        return None

    def build(self, cfg, default_args=None):
        args = dict(cfg)
        for k, v in (default_args or {}).items():
            args.setdefault(k, v)
        typ = args.pop("type")
        cls = self.get(typ) if isinstance(typ, str) else typ
        if cls is None:
            raise KeyError(f"{typ} is not in the {self.name} registry")
        return cls(**args)


class _Cfg(dict):
    """attribute-style dict (what mmcv.Config hands to the model ctors)."""
    __getattr__ = dict.get

    def __setattr__(self, k, v):
        self[k] = v


def to_cfg(o):
    if isinstance(o, dict):
        return _Cfg({k: to_cfg(v) for k, v in o.items()})
    if isinstance(o, (list, tuple)):
        return type(o)(to_cfg(v) for v in o)
    return o


class _BaseModule(nn.Module):
    def __init__(self, init_cfg=None):
        super().__init__()
        self.init_cfg = init_cfg

    def init_weights(self):
        pass


class _Sequential(_BaseModule, nn.Sequential):
    def __init__(self, *args, init_cfg=None):
        _BaseModule.__init__(self, init_cfg)
        nn.Sequential.__init__(self, *args)


def _build_conv_layer(cfg, *args, **kwargs):
    assert cfg is None or cfg.get("type") in (None, "Conv2d", "Conv")
    return nn.Conv2d(*args, **kwargs)


def _build_norm_layer(cfg, num_features, postfix=""):
    assert cfg["type"] == "BN", cfg
    layer = nn.BatchNorm2d(num_features, eps=cfg.get("eps", 1e-5))
    for p in layer.parameters():
        p.requires_grad = cfg.get("requires_grad", True)
    return "bn" + str(postfix), layer


def _build_plugin_layer(*a, **k):
    raise NotImplementedError("plugins are not on the PEANUT inference path")


class _ConvModule(nn.Module):
    """conv -> norm -> act with mmcv's attribute names ``conv`` / ``bn`` / ``activate``."""

    def __init__(self, in_channels, out_channels, kernel_size, stride=1, padding=0, dilation=1,
                 groups=1, bias="auto", conv_cfg=None, norm_cfg=None, act_cfg=dict(type="ReLU"),
                 inplace=True, **_):
        super().__init__()
        self.with_norm = norm_cfg is not None
        self.with_activation = act_cfg is not None
        if bias == "auto":
            bias = not self.with_norm
        self.conv = _build_conv_layer(conv_cfg, in_channels, out_channels, kernel_size,
                                      stride=stride, padding=padding, dilation=dilation,
                                      groups=groups, bias=bias)
        if self.with_norm:
            self.norm_name, norm = _build_norm_layer(norm_cfg, out_channels)
            self.add_module(self.norm_name, norm)
        if self.with_activation:
            assert act_cfg["type"] == "ReLU"
            self.activate = nn.ReLU(inplace=inplace)

    def forward(self, x):
        x = self.conv(x)
        if self.with_norm:
            x = getattr(self, self.norm_name)(x)
        if self.with_activation:
            x = self.activate(x)
        return x


def _passthrough_deco(*dargs, **dkw):
    def deco(fn):
        return fn
    return deco


def _install_mmcv_standin():
    if "mmcv" in sys.modules and not getattr(sys.modules["mmcv"], "_peanut_standin", False):
        return  # a real mmcv is present: use it
    mk = lambda n: sys.modules.setdefault(n, types.ModuleType(n))  # noqa: E731
    mmcv = mk("mmcv")
    mmcv._peanut_standin = True
    cnn, bricks, breg = mk("mmcv.cnn"), mk("mmcv.cnn.bricks"), mk("mmcv.cnn.bricks.registry")
    runner, utils, pw = mk("mmcv.runner"), mk("mmcv.utils"), mk("mmcv.utils.parrots_wrapper")
    cnn.MODELS = _Registry("model")
    breg.ATTENTION = _Registry("attention")
    cnn.build_conv_layer, cnn.build_norm_layer = _build_conv_layer, _build_norm_layer
    cnn.build_plugin_layer, cnn.ConvModule = _build_plugin_layer, _ConvModule
    cnn.bricks, bricks.registry = bricks, breg
    runner.BaseModule, runner.Sequential = _BaseModule, _Sequential
    runner.auto_fp16, runner.force_fp32 = _passthrough_deco, _passthrough_deco
    utils.Registry = _Registry
    pw._BatchNorm = nn.modules.batchnorm._BatchNorm
    utils.parrots_wrapper = pw
    mmcv.cnn, mmcv.runner, mmcv.utils = cnn, runner, utils


def _load(modname: str, relpath: str):
    spec = importlib.util.spec_from_file_location(modname, os.path.join(MMSEG, relpath))
    mod = importlib.util.module_from_spec(spec)
    sys.modules[modname] = mod
    spec.loader.exec_module(mod)
    return mod


_REF_BUILDER = None


def load_reference_mmseg():
    """Load the reference's inference-path model files; returns their ``models.builder``."""
    global _REF_BUILDER
    if _REF_BUILDER is not None:
        return _REF_BUILDER
    if not reference_available():
        raise RuntimeError(f"reference checkout not found at {REF}")
    _install_mmcv_standin()

    def pkg(name):
        m = types.ModuleType(name)
        m.__path__ = []
        sys.modules[name] = m
        return m

    for n in ("mmseg", "mmseg.models", "mmseg.models.utils", "mmseg.models.backbones",
              "mmseg.models.decode_heads", "mmseg.models.segmentors", "mmseg.models.losses",
              "mmseg.core", "mmseg.core.utils", "mmseg.ops"):
        pkg(n)
    ops = _load("mmseg.ops.wrappers", "ops/wrappers.py")
    sys.modules["mmseg.ops"].resize = ops.resize
    misc = _load("mmseg.core.utils.misc", "core/utils/misc.py")
    sys.modules["mmseg.core"].add_prefix = misc.add_prefix
    sys.modules["mmseg.core"].build_pixel_sampler = lambda cfg, **kw: None
    sys.modules["mmseg.models.losses"].accuracy = lambda *a, **k: None
    builder = _load("mmseg.models.builder", "models/builder.py")
    sys.modules["mmseg.models"].builder = builder

    @builder.LOSSES.register_module()
    class MyLoss(nn.Module):   # nav/agent/prediction.py:86-109, training-only; ctor must exist
        def __init__(self, reduction="mean", loss_weight=1.0):
            super().__init__()

    rl = _load("mmseg.models.utils.res_layer", "models/utils/res_layer.py")
    sys.modules["mmseg.models.utils"].ResLayer = rl.ResLayer
    _load("mmseg.models.backbones.resnet", "models/backbones/resnet.py")
    _load("mmseg.models.decode_heads.decode_head", "models/decode_heads/decode_head.py")
    _load("mmseg.models.decode_heads.psp_head", "models/decode_heads/psp_head.py")
    _load("mmseg.models.decode_heads.fcn_head", "models/decode_heads/fcn_head.py")
    _load("mmseg.models.segmentors.base", "models/segmentors/base.py")
    _load("mmseg.models.segmentors.encoder_decoder", "models/segmentors/encoder_decoder.py")
    _REF_BUILDER = builder
    return builder


def build_reference_model(cfg_path: str | None = None, in_channels: int | None = None, backbone: dict | None = None,
                          decode_head: dict | None = None):
    """``init_segmentor`` without a checkpoint (prediction/mmseg/apis/inference.py:29-39).  ``backbone`` / ``decode_head``:
    fields of nav/pred_model_cfg.py overridden before the build (the variants a maintainer could edit in that file)."""
    builder = load_reference_mmseg()
    cfg_path = cfg_path or os.path.join(REF, "nav", "pred_model_cfg.py")
    ns = {}
    exec(compile(open(cfg_path).read(), cfg_path, "exec"), ns)
    model_cfg = to_cfg(ns["model"])
    model_cfg.pretrained = None                      # inference.py:29
    model_cfg.backbone.pretrained = None
    model_cfg.train_cfg = None                       # inference.py:30
    if in_channels is not None:
        model_cfg.backbone.in_channels = in_channels
    for k, v in (backbone or {}).items():
        setattr(model_cfg.backbone, k, v)
    for k, v in (decode_head or {}).items():
        setattr(model_cfg.decode_head, k, v)
        if k in ("align_corners", "num_classes") and getattr(model_cfg, "auxiliary_head", None) is not None:
            setattr(model_cfg.auxiliary_head, k, v)
    model = builder.build_segmentor(model_cfg, test_cfg=None)
    model.eval()
    return model


def reference_forward(model, x: torch.Tensor):
    """``model(return_loss=False, rescale=True, img=[x], img_metas=[[meta]*N])``
    (nav/agent/prediction.py:135-136) -> list of np.float32 [K,H,W]."""
    n, c, h, w = x.shape
    meta = dict(ori_shape=(h, w, c), img_shape=(h, w, c), pad_shape=(h, w, c),
                scale_factor=1.0, flip=False)
    with torch.no_grad():
        return model(return_loss=False, rescale=True, img=[x], img_metas=[[meta] * n])


def load_reference_mapping():
    """Import ``Semantic_Mapping`` (nav/agent/mapping.py) unmodified."""
    nav = os.path.join(REF, "nav")
    if nav not in sys.path:
        sys.path.insert(0, nav)
    import matplotlib
    matplotlib.use("Agg")
    from agent.mapping import Semantic_Mapping  # type: ignore
    return Semantic_Mapping
