"""ORACLE (test infrastructure): Python face of oracle/fmm_ref.c -- a stand-in for ``skfmm.distance`` with the
call shape the reference uses (masked array in, masked array out; nav/agent/agent_state.py:391,
nav/agent/utils/fmm_planner.py:65,73).  PARITY UNPINNED: scikit-fmm==2019.1.30 (peanut.Dockerfile:8) is absent;
see the header of fmm_ref.c for what is restated.  The C file is compiled with gcc on first use (or by
``__graft_entry__.build()``) into oracle/_build/libfmm_ref.so."""
from __future__ import annotations

import ctypes as C
import os
import subprocess
import sys

import numpy as np
from numpy import ma

HERE = os.path.dirname(os.path.abspath(__file__))
SRC = os.path.join(HERE, "fmm_ref.c")
LIB = os.path.join(HERE, "_build", "libfmm_ref.so")
_LIB = None


def build(force: bool = False) -> str:
    if force or not os.path.exists(LIB) or os.path.getmtime(LIB) < os.path.getmtime(SRC):
        os.makedirs(os.path.dirname(LIB), exist_ok=True)
        tmp = LIB + f".tmp{os.getpid()}"
        subprocess.check_call(["gcc", "-O2", "-shared", "-fPIC", "-o", tmp, SRC, "-lm"])
        os.replace(tmp, LIB)
    return LIB


def _lib():
    global _LIB
    if _LIB is None:
        _LIB = C.CDLL(build())
        _LIB.fmm_ref_distance.restype = C.c_int
        _LIB.fmm_ref_distance.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p]
    return _LIB


def distance(phi, dx=1, order=2):
    """``skfmm.distance(phi, dx=1)`` for a 2-D (masked) array: signed distance to the zero contour of phi, masked
    cells excluded, unreached cells masked in the result (pfmm.py: pre_process_args / post_process_result)."""
    if dx != 1:
        raise NotImplementedError("the reference only calls skfmm.distance with dx=1")
    mask = ma.getmaskarray(phi) if isinstance(phi, ma.MaskedArray) else np.zeros(np.shape(phi), bool)
    data = np.ascontiguousarray(ma.getdata(phi), dtype=np.float64)
    if data.ndim != 2:
        raise NotImplementedError("2-D only")
    m8 = np.ascontiguousarray(mask, dtype=np.uint8)
    out = np.empty_like(data)
    rc = _lib().fmm_ref_distance(data.ctypes.data, m8.ctypes.data, data.shape[0], data.shape[1], int(order), out.ctypes.data)
    if rc == 2:
        raise ValueError("the array phi contains no zero contour (no zero level set)")
    big = out == sys.float_info.max
    if big.any():
        out[big] = 0
        return ma.MaskedArray(out, big)
    return out
