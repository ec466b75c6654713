"""The pose helpers of the reference that the per-step path uses (nav/agent/utils/pose.py:4-21: `get_l2_distance`,
`get_rel_pose_change`; the planner-side `get_new_pose` / `threshold_poses` are not part of it), host-side NumPy like the original.

Arithmetic note: Habitat hands over ``gps`` / ``compass`` as float32 arrays and the reference's environment is
NumPy 1.x, where ``float32 scalar ** 2`` (scalar with a Python number) is evaluated in float64 and
``float64 scalar * float32 array`` in float32.  The functions below spell those widths out so that the result does
not depend on the NumPy version installed (NumPy >= 2 would keep everything in float32): the distance is formed
in float64, everything that touches the ``(1,)`` compass array is float32."""
from __future__ import annotations

import numpy as np


def get_l2_distance(x1, x2, y1, y2):
    """pose.py:4-8; float64 like NumPy 1.x evaluates ``scalar ** 2`` / ``** 0.5``."""
    return (np.float64(x1 - x2) ** 2 + np.float64(y1 - y2) ** 2) ** 0.5


def get_rel_pose_change(pos2, pos1):
    """pose.py:11-21: motion from pos1 to pos2 expressed in pos1's frame -> (dx, dy, do)."""
    x1, y1, o1 = pos1
    x2, y2, o2 = pos2
    theta = np.arctan2(y2 - y1, x2 - x1) - o1
    dist = get_l2_distance(x1, x2, y1, y2)
    if isinstance(theta, np.ndarray) and theta.dtype == np.float32:
        dist = np.float32(dist)          # float64 scalar x float32 array -> float32 loop (value-based casting)
    dx = dist * np.cos(theta)
    dy = dist * np.sin(theta)
    do = o2 - o1
    return dx, dy, do


class PoseTracker:
    """The pose bookkeeping of ``PEANUT_Agent`` (nav/agent/peanut_agent.py:70-95): simulator location from the
    ``gps`` / ``compass`` sensors and the per-step relative pose change that ``Semantic_Mapping`` integrates."""

    def __init__(self):
        self.last_sim_location = None

    def reset(self):
        self.last_sim_location = None

    @staticmethod
    def get_sim_location(obs):
        """peanut_agent.py:77-84: (x, y, o) = (gps[0], -gps[1], compass wrapped to (-pi, pi])."""
        x = obs['gps'][0]
        y = -obs['gps'][1]
        o = obs['compass']
        if o > np.pi:
            o = o - 2 * np.pi          # the reference's `o -= 2*pi` mutates the observation in place; a copy here
        return x, y, o

    def get_pose_change(self, obs):
        """peanut_agent.py:86-95."""
        curr_sim_pose = self.get_sim_location(obs)
        if self.last_sim_location is not None:
            dx, dy, do = get_rel_pose_change(curr_sim_pose, self.last_sim_location)
            dx, dy, do = dx[0], dy[0], do[0]
        else:
            dx, dy, do = 0, 0, 0
        self.last_sim_location = curr_sim_pose
        return dx, dy, do

    def get_info(self, obs):
        """peanut_agent.py:70-75."""
        dx, dy, do = self.get_pose_change(obs)
        return {'sensor_pose': [dx, dy, do]}
