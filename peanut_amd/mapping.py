"""Drop-in for ``nav/agent/mapping.py``: ``Semantic_Mapping(args)`` with the reference's call
signature and return tuple, executed by the HIP library (``peanut_map_*``); no CPU fallback."""
from __future__ import annotations

import ctypes as C

import torch
import torch.nn as nn

from . import _lib

_ARG_FIELDS = ("frame_height", "frame_width", "map_resolution", "map_size_cm", "global_downscaling",
               "vision_range", "hfov", "du_scale", "cat_pred_threshold", "exp_pred_threshold",
               "map_pred_threshold", "num_sem_categories", "camera_height")


class Semantic_Mapping(nn.Module):
    """``Semantic_Mapping`` (nav/agent/mapping.py:10-179).  ``args`` carries the fields the reference
    ctor reads (:15-37) plus ``device``.  ``forward`` keeps the reference's tensor contract:

        fp_map_pred [1,V,V], map_pred [C,M,M], pose_pred [3], current_pose [3]
            = sem_map_module(obs [1,C,h,w], pose_obs [3], maps_last [C,M,M], poses_last [3], agent_states)

    ``poses_last`` is updated IN PLACE and both returned poses alias it, exactly like the reference
    (its ``get_new_pose_batch`` mutates the view it is given, :143-160).  Everything stays on the
    device and is enqueued on the current stream; there is no host synchronisation."""

    def __init__(self, args):
        super().__init__()
        self.args = args
        self.device = getattr(args, "device", torch.device("cuda:0"))
        missing = [f for f in _ARG_FIELDS if not hasattr(args, f)]
        if missing:
            raise AttributeError(f"args is missing {missing}")
        if not torch.cuda.is_available():
            raise _lib.PeanutHipError("Semantic_Mapping needs a HIP device (no CPU fallback)")
        self._lib = _lib.load()
        c = _lib.MapCfgC()
        for f in _ARG_FIELDS:
            setattr(c, f, getattr(args, f))
        self.num_sem_categories = args.num_sem_categories
        self._h = C.c_void_p()
        with _lib.default_options(), torch.cuda.device(self.device):     # (the option lock: see _lib.default_options)
            _lib.check(self._lib.peanut_map_create(C.byref(self._h), C.byref(c)), "peanut_map_create")
        dims = (C.c_int * 4)()
        _lib.check(self._lib.peanut_map_dims(self._h, C.byref(dims)), "peanut_map_dims")
        self.channels, self.map_cells, self.vision_range, self.n_points = (int(d) for d in dims)

    def __del__(self):
        h = getattr(self, "_h", None)
        if h is not None and h.value:
            try:
                self._lib.peanut_map_destroy(h)
                h.value = None       # (no attribute assignment: nn.Module.__setattr__ is gone at interpreter shutdown)
            except Exception:  # pragma: no cover
                pass

    @staticmethod
    def _chk(t, shape, name):
        if not (t.is_cuda and t.dtype == torch.float32 and tuple(t.shape) == tuple(shape)):
            raise ValueError(f"{name} must be a float32 HIP tensor of shape {tuple(shape)}, got "
                             f"{t.dtype} {tuple(t.shape)} on {t.device}")

    def use_graph(self, enable: bool = True):
        """Replay the ten launches of a step as one hipGraph per distinct set of buffers (captured on second use).
        Only effective when the caller passes persistent ``out=`` buffers (e.g. two map buffers used in turn)."""
        _lib.check(self._lib.peanut_map_use_graph(self._h, int(enable)), "peanut_map_use_graph")

    def forward(self, obs, pose_obs, maps_last, poses_last, agent_states=None, out=None):
        Cc, M, V = self.channels, self.map_cells, self.vision_range
        h, w = self.args.frame_height, self.args.frame_width
        self._chk(obs, (1, Cc, h, w), "obs")
        self._chk(pose_obs, (3,), "pose_obs")
        self._chk(maps_last, (Cc, M, M), "maps_last")
        self._chk(poses_last, (3,), "poses_last")
        if not poses_last.is_contiguous():
            raise ValueError("poses_last must be contiguous (it is updated in place)")
        obs, pose_obs, maps_last = obs.contiguous(), pose_obs.contiguous(), maps_last.contiguous()
        if out is None:
            fp_map_pred = torch.empty((1, V, V), dtype=torch.float32, device=obs.device)
            map_pred = torch.empty((Cc, M, M), dtype=torch.float32, device=obs.device)
        else:                                   # caller-owned (fp_map_pred, map_pred), must not alias maps_last
            fp_map_pred, map_pred = out
            self._chk(fp_map_pred, (1, V, V), "out[0]")
            self._chk(map_pred, (Cc, M, M), "out[1]")
            if not (fp_map_pred.is_contiguous() and map_pred.is_contiguous()):
                raise ValueError("out buffers must be contiguous")
        with torch.cuda.device(obs.device):
            rc = self._lib.peanut_map_forward(self._h, obs.data_ptr(), pose_obs.data_ptr(), maps_last.data_ptr(),
                                              poses_last.data_ptr(), fp_map_pred.data_ptr(), map_pred.data_ptr(),
                                              _lib.current_stream_ptr(obs.device))
        _lib.check(rc, "peanut_map_forward")
        return fp_map_pred, map_pred, poses_last, poses_last
