"""Operator-level handles of the HIP library (thin ctypes wrappers, device tensors in/out)."""
from __future__ import annotations

import ctypes as C
from typing import Optional

import numpy as np
import torch

from . import _lib


def round_up(n: int, m: int) -> int:
    return (n + m - 1) // m * m


class FusedConv:
    """conv (+per-channel scale/shift = folded BatchNorm) (+residual) (+ReLU) on NHWC fp32 --
    the mmcv ``ConvModule`` / Bottleneck conv+bn(+relu) pattern
    (prediction/mmseg/models/backbones/resnet.py:267-307) as ONE kernel launch.

    weight: [cout, cin, kh, kw] (OIHW, host or device tensor); the NHWC input must carry
    ``cin_pad`` channels (multiple of 16; default round_up(cin, 16)), padded channels are ignored.
    ``conv_algo='auto'`` runs stride-1 3x3 layers with >= 128 input channels as Winograd F(4x4,3x3)
    (fp32 transforms, csrc/winograd.hip; the prediction planner goes down to 64 channels and picks F(5x5) / F(6x6) per
    shape, a single conv keeps F(4x4)); ``'direct'`` always uses the direct implicit GEMM.  ``options``: tuning options of
    this handle (csrc/options.h), e.g. ``{"pw256w_mintiles": 256}``.
    """

    def __init__(self, weight: torch.Tensor, scale: Optional[torch.Tensor] = None,
                 shift: Optional[torch.Tensor] = None, stride: int = 1, padding: int = 0,
                 dilation: int = 1, relu: bool = False, cin_pad: Optional[int] = None,
                 precision: str = "fp32", conv_algo: str = "auto", device=None, options=None):
        """``device``: HIP device the weights are uploaded to (default: the current one); inputs must live there."""
        self._lib = _lib.load()
        self.device = torch.device("cuda", torch.cuda.current_device()) if device is None else torch.device(device)
        w = np.ascontiguousarray(weight.detach().cpu().numpy(), dtype=np.float32)
        cout, cin, kh, kw = w.shape
        self.cout, self.cin, self.kh, self.kw = cout, cin, kh, kw
        self.stride, self.padding, self.dilation = stride, padding, dilation
        self.cin_pad = cin_pad if cin_pad is not None else round_up(cin, 16)
        sc = None if scale is None else np.ascontiguousarray(scale.detach().cpu().numpy(), dtype=np.float32)
        sh = None if shift is None else np.ascontiguousarray(shift.detach().cpu().numpy(), dtype=np.float32)
        self._h = C.c_void_p()
        with _lib.default_options(**(options or {})), torch.cuda.device(self.device):      # the library allocates on the current device
            _lib.check(self._lib.peanut_conv_create(
                C.byref(self._h), w.ctypes.data, None if sc is None else sc.ctypes.data,
                None if sh is None else sh.ctypes.data, cout, cin, self.cin_pad, kh, kw, stride, padding,
                dilation, int(relu), _lib.PRECISIONS[precision], _lib.CONV_ALGOS[conv_algo]), "peanut_conv_create")

    def __del__(self):
        h = getattr(self, "_h", None)
        if h is not None and h.value:
            try:
                self._lib.peanut_conv_destroy(h)
            except Exception:  # pragma: no cover
                pass
            self._h = C.c_void_p()

    def set_option(self, key: str, value: int) -> None:
        """Change one run-time tuning option of this conv handle (kernel gates; csrc/options.h)."""
        _lib.check(self._lib.peanut_conv_set_option(self._h, key.encode(), int(value)), "peanut_conv_set_option")

    def out_hw(self, h: int, w: int):
        def o(n, k):
            return (n + 2 * self.padding - self.dilation * (k - 1) - 1) // self.stride + 1
        return o(h, self.kh), o(w, self.kw)

    def __call__(self, x: torch.Tensor, x2: Optional[torch.Tensor] = None,
                 residual: Optional[torch.Tensor] = None) -> torch.Tensor:
        """x: [B,H,W,c1] NHWC float32 on the HIP device (c1 == cin_pad, or c1 + x2.shape[3] ==
        cin_pad for the two-source concat form); returns [B,Ho,Wo,cout]."""
        assert x.is_cuda and x.dtype == torch.float32 and x.is_contiguous() and x.dim() == 4
        if x.device != self.device:
            raise ValueError(f"input lives on {x.device}, the conv's weights on {self.device}")
        b, h, w, c1 = x.shape
        c2 = 0
        if x2 is not None:
            assert x2.is_cuda and x2.is_contiguous() and x2.shape[:3] == x.shape[:3]
            c2 = x2.shape[3]
        if c1 + c2 != self.cin_pad:
            raise ValueError(f"input carries {c1 + c2} channels, conv was built for {self.cin_pad}")
        ho, wo = self.out_hw(h, w)
        y = torch.empty((b, ho, wo, self.cout), dtype=torch.float32, device=x.device)
        if residual is not None:
            assert residual.is_contiguous() and tuple(residual.shape) == tuple(y.shape)
        with torch.cuda.device(x.device):
            rc = self._lib.peanut_conv_forward(
                self._h, x.data_ptr(), None if x2 is None else x2.data_ptr(), c1,
                None if residual is None else residual.data_ptr(), y.data_ptr(), b, h, w,
                _lib.current_stream_ptr(x.device))
        _lib.check(rc, "peanut_conv_forward")
        return y


def to_nhwc_padded(x_nchw: torch.Tensor, cpad: int) -> torch.Tensor:
    """Test helper (torch ops): NCHW -> NHWC with zero channel padding."""
    b, c, h, w = x_nchw.shape
    y = torch.zeros((b, h, w, cpad), dtype=x_nchw.dtype, device=x_nchw.device)
    y[..., :c] = x_nchw.permute(0, 2, 3, 1)
    return y.contiguous()
