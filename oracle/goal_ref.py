"""ORACLE (test infrastructure, not product code): CPU restatement of the long-term goal selection of the reference,
``Agent_State.update_global_goal`` (nav/agent/agent_state.py:376-415), and of ``FMMPlanner.set_goal /
set_multi_goal`` (nav/agent/utils/fmm_planner.py:55-75), in NumPy + scipy.ndimage + oracle/fmm_ref (the
scikit-fmm stand-in -- PARITY UNPINNED for that piece, see fmm_ref.c).

``skimage.morphology.binary_dilation(image, footprint)`` is ``scipy.ndimage.binary_dilation(image,
structure=footprint)`` (scikit-image's implementation is that one call; zero border)."""
from __future__ import annotations

import numpy as np
from numpy import ma
from scipy import ndimage as ndi

from oracle import fmm_ref
from oracle.agent_ref import disk


def binary_dilation(image, footprint=None):
    return ndi.binary_dilation(np.asarray(image) != 0, structure=footprint)


def mark_visited(visited_vis, prev_rc, cur_rc, half=2):
    """Test-harness stand-in for the trail Agent_Helper draws into ``visited_vis`` between consecutive agent cells
    (agent_helper.py:262-266, vu.draw_line): a (2*half+1)-wide bar of samples along the segment, full-map cells."""
    (r0, c0), (r1, c1) = prev_rc, cur_rc
    n = max(abs(r1 - r0), abs(c1 - c0), 1)
    H, W = visited_vis.shape
    for k in range(n + 1):
        r = int(round(r0 + (r1 - r0) * k / n))
        c = int(round(c0 + (c1 - c0) * k / n))
        visited_vis[max(r - half, 0):min(r + half + 1, H), max(c - half, 0):min(c + half + 1, W)] = 1
    return visited_vis


def traversible_map(full_obstacle, selem, collision_map, visited_vis):
    """agent_state.py:382-386."""
    trav = binary_dilation(np.rint(full_obstacle), selem) != True  # noqa: E712
    trav[collision_map == 1] = 0
    trav[visited_vis == 1] = 1
    return trav


def geodesic_distance(trav, seed_rc):
    """agent_state.py:388-393: masked FMM from the agent's cell; +inf where masked or unreachable."""
    traversible_ma = ma.masked_values(trav * 1, 0)
    traversible_ma[seed_rc[0], seed_rc[1]] = 0
    dd = fmm_ref.distance(traversible_ma, dx=1)
    dd = ma.filled(dd, np.max(dd) + 1)
    dd[np.where(dd == np.max(dd))] = np.inf
    return dd


class GoalSelector:
    """State that ``update_global_goal`` carries across calls (``dd_wt``, ``last_global_goal``, ``global_goals``)."""

    def __init__(self, args, full_hw, col_rad=None):
        self.args = args
        self.full_w, self.full_h = full_hw
        self.selem = disk(int(args.col_rad if col_rad is None else col_rad))
        self.reset()

    def reset(self):
        self.dd_wt = None
        self.value = None
        self.last_global_goal = None
        self.global_goals = None
        self.dd = None

    def update(self, full_obstacle, lmb, loc_rc, target_pred, collision_map, visited_vis):
        """full_obstacle = full_map[0] [W,H]; lmb = (gx1, gx2, gy1, gy2); loc_rc = agent cell in the local map;
        target_pred [w,h].  Returns the new ``global_goals``."""
        args = self.args
        trav = traversible_map(full_obstacle, self.selem, collision_map, visited_vis)
        r = int(np.clip(loc_rc[0] + lmb[0], 0, self.full_w - 1))
        c = int(np.clip(loc_rc[1] + lmb[2], 0, self.full_h - 1))
        dd = geodesic_distance(trav, (r, c))
        self.dd = dd
        temperature = args.dist_weight_temperature / args.map_resolution
        with np.errstate(over="ignore"):
            dd_wt = np.exp(-dd / temperature)[lmb[0]:lmb[1], lmb[2]:lmb[3]]
        if np.sum(dd_wt) < 10 and self.dd_wt is not None:      # stuck inside an obstacle: keep the last weights
            dd_wt = self.dd_wt
        if args.dist_weight_temperature == -1:
            value = target_pred
        elif args.dist_weight_temperature == 0:
            dd = dd.copy()
            dd[np.where(dd < 60)] = np.inf
            value = np.exp(-dd / 100.)[lmb[0]:lmb[1], lmb[2]:lmb[3]]
        else:
            value = target_pred * dd_wt
        self.dd_wt = dd_wt
        self.value = value
        new_global_goal = [np.unravel_index(value.argmax(), value.shape)]
        if new_global_goal != self.last_global_goal:
            self.last_global_goal = self.global_goals
            self.global_goals = new_global_goal
        return self.global_goals


def fmm_set_goal(traversible, goal_rc):
    """FMMPlanner.set_goal (fmm_planner.py:55-67, scale 1, no auto_improve)."""
    traversible_ma = ma.masked_values(traversible * 1, 0)
    traversible_ma[int(goal_rc[0]), int(goal_rc[1])] = 0
    dd = fmm_ref.distance(traversible_ma, dx=1)
    return ma.filled(dd, np.max(dd) + 1)


def fmm_set_multi_goal(traversible, goal_map):
    """FMMPlanner.set_multi_goal (fmm_planner.py:69-75)."""
    traversible_ma = ma.masked_values(traversible * 1, 0)
    traversible_ma[goal_map == 1] = 0
    dd = fmm_ref.distance(traversible_ma, dx=1)
    return ma.filled(dd, np.max(dd) + 1)
