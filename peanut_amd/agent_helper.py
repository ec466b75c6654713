"""Observation formatting of the reference's ``Agent_Helper`` (nav/agent/agent_helper.py:166-217),
device-resident: ``preprocess_obs`` produces the ``[1, 4+ncat, h, w]`` tensor that
``Semantic_Mapping`` consumes, with one HIP launch (``peanut_preprocess_obs``) instead of the
reference's per-column Python loop on the host."""
from __future__ import annotations

import torch

from . import _lib


def preprocess_obs(rgb: torch.Tensor, depth: torch.Tensor, sem_seg_pred: torch.Tensor, args) -> torch.Tensor:
    """``_preprocess_obs`` + ``_preprocess_depth``: rgb uint8 [H,W,3], depth float32 [H,W,1] or [H,W]
    (simulator units in [0,1], 0 = invalid), sem_seg_pred float32 [H,W,ncat] -- all HIP tensors --
    -> float32 [1, 3+1+ncat, frame_height, frame_width].  ``args``: env_frame_width, frame_width,
    min_depth, max_depth (nav/arguments.py:44-66)."""
    lib = _lib.load()
    if not (rgb.is_cuda and depth.is_cuda and sem_seg_pred.is_cuda):
        raise _lib.PeanutHipError("preprocess_obs needs HIP tensors (no CPU fallback)")
    if depth.dim() == 3:
        depth = depth[:, :, 0]
    H, W = depth.shape
    ds = args.env_frame_width // args.frame_width          # agent_helper.py:185
    ncat = sem_seg_pred.shape[2]
    rgb = rgb.to(torch.uint8).contiguous()
    depth = depth.to(torch.float32).contiguous()
    sem = sem_seg_pred.to(torch.float32).contiguous()
    obs = torch.empty((1, 4 + ncat, H // ds, W // ds), dtype=torch.float32, device=depth.device)
    with torch.cuda.device(depth.device):
        rc = lib.peanut_preprocess_obs(rgb.data_ptr(), depth.data_ptr(), sem.data_ptr(), H, W, ncat, ds,
                                       float(args.min_depth), float(args.max_depth), obs.data_ptr(),
                                       _lib.current_stream_ptr(depth.device))
    _lib.check(rc, "peanut_preprocess_obs")
    return obs
