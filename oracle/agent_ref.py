"""ORACLE (test infrastructure, not product code): host-side restatements around the hot path --
the observation formatting of Agent_Helper (nav/agent/agent_helper.py:175-217) in NumPy, plus the
small deterministic helpers the agent-state fixtures use.  No reference import here (this file travels
to the GPU box); ``oracle/gen_golden_agent.py`` is what runs the reference's own Agent_State."""
from __future__ import annotations

from argparse import Namespace

import numpy as np


def disk(radius, dtype=np.uint8):
    """skimage.morphology.disk: footprint of all pixels with x^2 + y^2 <= radius^2."""
    L = np.arange(-radius, radius + 1)
    X, Y = np.meshgrid(L, L)
    return np.array((X ** 2 + Y ** 2) <= radius ** 2, dtype=dtype)


def fake_pattern(seed=123, n=6, size=720):
    return np.random.RandomState(seed).uniform(-1.0, 1.0, size=(n, size, size)).astype(np.float32)


class FakePrediction:
    """Deterministic stand-in for PEANUT_Prediction_Model.get_prediction: depends on the crop it is
    given, so that a misplaced crop / pad shows up in target_pred."""

    def __init__(self, size=720):
        self.pattern = fake_pattern(size=size)

    def get_prediction(self, m):
        sel = m[[0, 1, 4, 5, 6, 7]].astype(np.float32)
        return (np.tanh(sel + self.pattern) * np.float32(0.5) + np.float32(0.5)).astype(np.float32)


def agent_args(**over):
    """nav/arguments.py defaults of the fields the hot-path callers read."""
    a = dict(seed=1, cuda=False, sem_gpu_id=0, num_sem_categories=10, map_size_cm=4800, map_resolution=5,
             global_downscaling=2, only_explore=1, col_rad=4, grid_resolution=24, num_local_steps=20,
             switch_step=0, update_goal_freq=10, goal_reached_dist=75, prediction_window=720, visualize=0,
             frame_height=120, frame_width=160, env_frame_height=480, env_frame_width=640, vision_range=100,
             hfov=79.0, du_scale=1, cat_pred_threshold=5.0, exp_pred_threshold=1.0, map_pred_threshold=0.1,
             camera_height=0.88, min_depth=0.5, max_depth=5.0, sem_pred_prob_thr=0.95, goal_thr=0.985,
             dist_weight_temperature=500, timestep_limit=499,
             select_goal=False)     # peanut_amd-only switch: the round-1 fixtures drive Agent_State without update_global_goal
    a.update(over)
    return Namespace(**a)


def preprocess_depth_ref(depth, min_d, max_d):
    """``Agent_Helper._preprocess_depth`` (agent_helper.py:197-217), statement for statement."""
    depth = depth[:, :, 0] * 1
    for i in range(depth.shape[1]):
        invalid = depth[:, i] == 0.
        if np.mean(invalid) > 0.9:
            depth[:, i][invalid] = depth[:, i].max()
        else:
            depth[:, i][invalid] = 100.0
    mask2 = depth > 0.99
    depth[mask2] = 0.
    mask1 = depth == 0
    depth[mask1] = 100.0
    depth = min_d * 100.0 + depth * (max_d - min_d) * 100.0
    return depth


def preprocess_obs_ref(rgb, depth, sem_seg_pred, args):
    """``Agent_Helper._preprocess_obs`` (agent_helper.py:175-195) after the segmentation call.  The
    reference resizes RGB with PIL NEAREST (agent_helper.py:57-59,187), which for an integer factor ds
    selects source pixel ds*i + ds//2 -- the same pixels as the [ds//2::ds] slicing used for depth/sem."""
    depth = preprocess_depth_ref(depth, args.min_depth, args.max_depth)
    ds = args.env_frame_width // args.frame_width
    if ds != 1:
        rgb = rgb[ds // 2::ds, ds // 2::ds]
        depth = depth[ds // 2::ds, ds // 2::ds]
        sem_seg_pred = sem_seg_pred[ds // 2::ds, ds // 2::ds]
    depth = np.expand_dims(depth, axis=2)
    return np.concatenate((rgb, depth, sem_seg_pred), axis=2).transpose(2, 0, 1)
