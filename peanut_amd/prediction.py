"""Drop-in for ``nav/agent/prediction.py`` of the reference: same names, argument meaning and
error behaviour, but the model is the HIP library (no mmcv / mmseg, no CPU fallback).

    PEANUT_Prediction_Model(args).get_prediction(full_map) -> np.float32 [num_classes, H, W]

mirrors ``prediction.py:140-158`` (``init_segmentor`` + ``run_inference`` + ``expit``).  Added on
top (the reference is batch-1 only): ``get_prediction_batch`` for device tensors.
"""
from __future__ import annotations

import ctypes as C
import os
from typing import Dict, List, Optional

import numpy as np
import torch

from . import _lib
from .weights import (PredCfg, load_mmcv_checkpoint, pred_cfg_from_file,
                      select_inference_tensors)


def sigmoid(x):
    """``prediction.py:22-23`` (scipy.special.expit) for host arrays."""
    return 1.0 / (1.0 + np.exp(-np.asarray(x, dtype=np.float32)))


class HipSegmentor:
    """What ``init_segmentor`` returns in the reference (an eval-mode EncoderDecoder on a CUDA
    device, ``prediction/mmseg/apis/inference.py:12-40``), here a handle of the HIP library."""

    def __init__(self, cfg: PredCfg, state_dict: Dict[str, torch.Tensor], device="cuda:0",
                 classes=None, precision: str = "fp32", fold_ppm: bool = True, conv_algo: str = "auto",
                 options: Optional[Dict[str, int]] = None):
        if not torch.cuda.is_available():
            raise _lib.PeanutHipError("PEANUT_Prediction_Model needs a HIP device (no CPU fallback)")
        self.cfg = cfg
        self.device = torch.device(device)
        self.CLASSES = classes
        if precision not in _lib.PRECISIONS:
            raise ValueError(f"precision must be one of {sorted(_lib.PRECISIONS)}, got {precision!r}")
        if conv_algo not in _lib.CONV_ALGOS:
            raise ValueError(f"conv_algo must be one of {sorted(_lib.CONV_ALGOS)}, got {conv_algo!r}")
        self.precision = precision
        self.conv_algo = conv_algo
        self._lib = _lib.load()
        tensors = select_inference_tensors(state_dict, cfg)     # raises on missing/mis-shaped keys
        arr = (_lib.TensorC * len(tensors))()
        keep = []
        for i, (name, a) in enumerate(tensors):
            keep.append((name.encode(), a))
            arr[i].name = keep[-1][0]
            arr[i].data = a.ctypes.data
            arr[i].ndim = a.ndim
            for d in range(a.ndim):
                arr[i].shape[d] = a.shape[d]
        c = _lib.PredCfgC()
        c.in_channels, c.num_classes = cfg.in_channels, cfg.num_classes
        for i in range(4):
            c.strides[i], c.dilations[i] = cfg.strides[i], cfg.dilations[i]
        c.contract_dilation = int(cfg.contract_dilation)
        for i, k in enumerate(cfg.pool_scales):
            c.pool_scales[i] = k
        c.n_pool_scales = len(cfg.pool_scales)
        c.head_channels, c.align_corners, c.bn_eps = cfg.head_channels, int(cfg.align_corners), cfg.bn_eps
        c.precision = _lib.PRECISIONS[precision]
        c.fold_ppm = int(fold_ppm)
        c.conv_algo = _lib.CONV_ALGOS[conv_algo]
        self.fold_ppm = bool(fold_ppm)
        self._h = C.c_void_p()
        # tuning options of THIS handle (csrc/options.h): the library snapshots its process defaults at creation, so the
        # create-time ones (Winograd forms, packing tiles) are set as defaults around the create call only
        self.options = dict(options or {})
        with _lib.default_options(**self.options), torch.cuda.device(self.device):
            _lib.check(self._lib.peanut_pred_create(C.byref(self._h), C.byref(c), arr, len(tensors)),
                       "peanut_pred_create")

    def __del__(self):
        h = getattr(self, "_h", None)
        if h is not None and h.value:
            try:
                self._lib.peanut_pred_destroy(h)
            except Exception:  # pragma: no cover - interpreter shutdown
                pass
            self._h = C.c_void_p()

    # nn.Module-ish surface the reference touches (prediction.py:150-152)
    def eval(self):
        return self

    def set_option(self, key: str, value: int) -> None:
        """Change one run-time tuning option of this handle (csrc/options.h; e.g. ``pw256_mink``, ``ppm_overlap``): its cached
        launch plans are dropped and rebuilt under the new policy.  Create-time options are refused by the library."""
        _lib.check(self._lib.peanut_pred_set_option(self._h, key.encode(), int(value)), "peanut_pred_set_option")
        self.options[key] = int(value)

    def get_option(self, key: str) -> int:
        v = C.c_longlong()
        _lib.check(self._lib.peanut_pred_get_option(self._h, key.encode(), C.byref(v)), "peanut_pred_get_option")
        return int(v.value)

    def forward_logits(self, x: torch.Tensor, apply_sigmoid: bool = False,
                       out: Optional[torch.Tensor] = None) -> torch.Tensor:
        """x: float32 NCHW on ``self.device`` -> [B, num_classes, H, W] on the same device,
        enqueued on torch's current stream (no synchronisation)."""
        if x.dim() != 4 or x.shape[1] != self.cfg.in_channels:
            raise ValueError(f"expected [B,{self.cfg.in_channels},H,W], got {tuple(x.shape)}")
        if x.dtype != torch.float32 or not x.is_cuda:
            raise ValueError("input must be a float32 tensor on the HIP device")
        x = x.contiguous()
        b, _, h, w = x.shape
        if out is None:
            out = torch.empty((b, self.cfg.num_classes, h, w), dtype=torch.float32, device=x.device)
        elif tuple(out.shape) != (b, self.cfg.num_classes, h, w) or not out.is_contiguous():
            raise ValueError("out has the wrong shape or is not contiguous")
        with torch.cuda.device(x.device):
            rc = self._lib.peanut_pred_forward(self._h, x.data_ptr(), out.data_ptr(), b, h, w,
                                               int(apply_sigmoid), _lib.current_stream_ptr(x.device))
        _lib.check(rc, "peanut_pred_forward")
        return out

    def check_range(self, y: torch.Tensor) -> torch.Tensor:
        """fp16x3 only: an activation outside fp16's exponent range turns the output into NaN (include/peanut_hip.h);
        the host-facing entry points (``get_prediction``, ``run_inference``: they synchronise anyway) call this and raise
        instead of handing NaN probabilities to the planner.  ``forward_logits`` / ``get_prediction_batch`` stay
        asynchronous and unchecked."""
        if self.precision == "fp16x3" and not bool(torch.isfinite(y).all()):
            raise FloatingPointError("precision='fp16x3': a value left fp16's range (|x| >= 65520 in an emulated layer -- for "
                                     "the Winograd layers that is the transformed input, up to ~100 x the activations) and the "
                                     "output is NaN; run this model with precision='bf16x6' (fp32's exponent range) or 'fp32'")
        return y

    def workspace_bytes(self, b: int, h: int, w: int) -> int:
        return int(self._lib.peanut_pred_workspace_bytes(self._h, b, h, w))

    def flops_per_map(self, h: int, w: int) -> float:
        return float(self._lib.peanut_pred_flops_per_map(self._h, h, w))

    # ---- test / profiling hooks ----
    def debug_keep(self, keep: bool = True):
        _lib.check(self._lib.peanut_pred_debug_keep(self._h, int(keep)), "peanut_pred_debug_keep")

    def debug_tensor(self, name: str) -> torch.Tensor:
        """Copy of a named NHWC intermediate of the last forward (needs debug_keep(True))."""
        dims = (C.c_int * 4)()
        _lib.check(self._lib.peanut_pred_debug_read(self._h, name.encode(), None, 0, C.byref(dims), None),
                   "peanut_pred_debug_read")
        out = torch.empty(tuple(dims), dtype=torch.float32, device=self.device)
        with torch.cuda.device(self.device):
            _lib.check(self._lib.peanut_pred_debug_read(self._h, name.encode(), out.data_ptr(), out.numel(),
                                                        C.byref(dims), _lib.current_stream_ptr(self.device)),
                       "peanut_pred_debug_read")
        return out

    def use_graph(self, enable: bool = True):
        """Replay the launch sequence of each (shape, input buffer, output buffer) combination as one hipGraph
        (captured on its second use).  Pays off for small batches, where ~80 launches per forward dominate; keep
        the input/output tensors alive and reuse them (``out=``) so that the cached graphs are hit, and run under a
        non-default stream (``with torch.cuda.stream(torch.cuda.Stream())``): HIP cannot capture the NULL stream,
        where the launches simply stay plain."""
        _lib.check(self._lib.peanut_pred_use_graph(self._h, int(enable)), "peanut_pred_use_graph")

    def probe_enable(self, enable: bool = True):
        """Per-op HIP-event probe inside forward (see include/peanut_hip.h)."""
        _lib.check(self._lib.peanut_pred_probe_enable(self._h, int(enable)), "peanut_pred_probe_enable")

    def probe_collect(self):
        """-> (n_forwards, [(op name, kernel family, total ms over the forwards, executed flops per launch,
        algorithmic HBM bytes per launch)])"""
        n = 256
        names, kernels = (C.c_char_p * n)(), (C.c_char_p * n)()
        ms, fl, by = (C.c_double * n)(), (C.c_double * n)(), (C.c_double * n)()
        nf = C.c_int(0)
        cnt = self._lib.peanut_pred_probe_collect(self._h, n, names, kernels, ms, fl, by, C.byref(nf))
        if cnt < 0:
            _lib.check(cnt, "peanut_pred_probe_collect")
        return nf.value, [(names[i].decode(), kernels[i].decode(), float(ms[i]), float(fl[i]), float(by[i]))
                          for i in range(min(cnt, n))]

    def profile(self, x: torch.Tensor, repeats: int = 1):
        """[(op name, kernel family, mean ms, flops, bytes)] of a forward, via the event probe."""
        self.probe_enable(True)
        try:
            for _ in range(repeats):
                self.forward_logits(x)
            nf, rows = self.probe_collect()
        finally:
            self.probe_enable(False)
        return [(a, k, ms / max(nf, 1), f, by) for a, k, ms, f, by in rows]


def init_segmentor(config, checkpoint=None, device="cuda:0", state_dict=None,
                   precision: str = "fp32", fold_ppm: bool = True, conv_algo: str = "auto",
                   options: Optional[Dict[str, int]] = None) -> HipSegmentor:
    """``prediction/mmseg/apis/inference.py:12-40``: config path (or PredCfg) + mmcv checkpoint.
    ``state_dict`` lets tests/benchmarks pass seeded weights instead of a checkpoint file; ``options`` are the handle's
    tuning options (csrc/options.h, e.g. ``{"wino_m": 6}``)."""
    if isinstance(config, str):
        cfg = pred_cfg_from_file(config)
    elif isinstance(config, PredCfg):
        cfg = config
    else:
        raise TypeError(f"config must be a filename or PredCfg object, but got {type(config)}")
    classes = None
    if checkpoint is not None:
        state_dict, meta = load_mmcv_checkpoint(checkpoint)
        classes = meta.get("CLASSES")          # inference.py:35
    if state_dict is None:
        raise ValueError("init_segmentor needs a checkpoint (or an explicit state_dict): the HIP "
                         "model has no random-init mode")
    return HipSegmentor(cfg, state_dict, device=device, classes=classes, precision=precision,
                        fold_ppm=fold_ppm, conv_algo=conv_algo, options=options)


def run_inference(model: HipSegmentor, full_map: np.ndarray) -> List[np.ndarray]:
    """``prediction.py:112-137``.  The MapFromArray -> MultiScaleFlipAug(1.0, no flip) -> Resize
    -> ImageToTensor -> Collect pipeline is an identity CHW->HWC->CHW float32 conversion with no
    normalisation (``pred_model_cfg.py:57-69``), so the map goes to the device as is.  Returns the
    list of raw-logit arrays ``simple_test`` returns (encoder_decoder.py:260-271)."""
    if full_map.ndim != 3:
        raise ValueError(f"full_map must be [C,H,W], got shape {full_map.shape}")
    x = torch.from_numpy(np.ascontiguousarray(full_map, dtype=np.float32))[None].to(model.device)
    y = model.check_range(model.forward_logits(x, apply_sigmoid=False))
    return list(y.cpu().numpy())               # .cpu() synchronises, like simple_test's


class PEANUT_Prediction_Model():
    """``nav/agent/prediction.py:140-158``.  ``args`` needs ``pred_model_wts``,
    ``pred_model_cfg`` and ``sem_gpu_id`` (``nav/arguments.py``); ``state_dict`` may replace the
    checkpoint file (seeded weights for tests/benchmarks)."""

    def __init__(self, args, state_dict=None, cfg: Optional[PredCfg] = None, precision: Optional[str] = None,
                 fold_ppm: Optional[bool] = None, conv_algo: Optional[str] = None,
                 options: Optional[Dict[str, int]] = None):
        self.args = args
        ckpt = getattr(args, "pred_model_wts", None) if state_dict is None else None
        if cfg is None:
            cfg_path = getattr(args, "pred_model_cfg", None)
            cfg = pred_cfg_from_file(cfg_path) if cfg_path else PredCfg()
        device = ("cuda:" + str(args.sem_gpu_id)) if args is not None and hasattr(args, "sem_gpu_id") \
            else "cuda:0"
        if precision is None:
            precision = getattr(args, "pred_precision", None) or os.environ.get("PEANUT_PRECISION", "fp32")
        if fold_ppm is None:
            fold_ppm = os.environ.get("PEANUT_FOLD_PPM", "1") != "0"
        if conv_algo is None:
            conv_algo = os.environ.get("PEANUT_CONV_ALGO", "auto")
        # precision="auto": the fastest fp32-class arithmetic that is safe for THIS model -- fp16x3, escalated once and for
        # good to bf16x6 (fp32's exponent range, six products) the first time an activation leaves fp16's range.  The
        # escalation happens at the host-facing entry point, where the result is checked anyway; it is announced.
        self._escalate = None
        if precision == "auto":
            precision = "fp16x3"
            self._escalate = dict(config=cfg, checkpoint=ckpt, device=device, state_dict=state_dict, precision="bf16x6",
                                  fold_ppm=fold_ppm, conv_algo=conv_algo, options=options)
        self.model = init_segmentor(cfg, checkpoint=ckpt, device=device, state_dict=state_dict,
                                    precision=precision, fold_ppm=fold_ppm, conv_algo=conv_algo, options=options)
        self.model.eval()
        self.model.cfg = cfg

    def get_prediction(self, full_map):
        """np.float32 [C,H,W] partial map -> np.float32 [num_classes,H,W] probabilities
        (``sigmoid(result[0])``, prediction.py:157-158; the sigmoid runs fused on the device)."""
        x = torch.from_numpy(np.ascontiguousarray(full_map, dtype=np.float32))[None].to(self.model.device)
        return self._checked_forward(x, True, None)[0].cpu().numpy()

    def _checked_forward(self, x, apply_sigmoid, out):
        try:
            return self.model.check_range(self.model.forward_logits(x, apply_sigmoid=apply_sigmoid, out=out))
        except FloatingPointError:
            if self._escalate is None:
                raise
            import warnings
            warnings.warn("precision='auto': an activation left fp16's exponent range; this model runs in bf16x6 from now on")
            kw, self._escalate = self._escalate, None
            cfg = kw.pop("config")
            self.model = init_segmentor(cfg, **kw)
            self.model.eval()
            self.model.cfg = cfg
            return self.model.forward_logits(x, apply_sigmoid=apply_sigmoid, out=out)

    def get_prediction_batch(self, maps: torch.Tensor, apply_sigmoid: bool = True,
                             out: Optional[torch.Tensor] = None) -> torch.Tensor:
        """Batched, device-resident variant: [B,C,H,W] float32 on the HIP device -> [B,K,H,W]
        (no host round trip, enqueued on the current stream).  With ``precision="auto"`` -- until the model has escalated
        to bf16x6 -- the result is range-checked, which synchronises; every explicit precision stays asynchronous."""
        if self._escalate is not None:
            return self._checked_forward(maps, apply_sigmoid, out)
        return self.model.forward_logits(maps, apply_sigmoid=apply_sigmoid, out=out)
