"""Long-term goal selection on the HIP device: ctypes face of csrc/goal.hip.

``GoalSelector`` carries what ``Agent_State.update_global_goal`` (nav/agent/agent_state.py:376-415) carries across
calls (``dd_wt`` inside the library handle; ``last_global_goal`` / ``global_goals`` here, as Python lists like the
reference).  ``FMMPlanner`` mirrors the distance-transform half of nav/agent/utils/fmm_planner.py (``set_goal``,
``set_multi_goal``, ``get_short_term_goal``).  scikit-fmm's role is taken by a fixed-point solver of the same
second-order discretisation; parity is pinned against oracle/fmm_ref.c only (scikit-fmm is absent: PARITY UNPINNED)."""
from __future__ import annotations

import ctypes as C
from typing import Optional

import numpy as np
import torch

from . import _lib


def _u8(t: Optional[torch.Tensor], device) -> Optional[torch.Tensor]:
    if t is None:
        return None
    if not isinstance(t, torch.Tensor):
        t = torch.from_numpy(np.ascontiguousarray(t))
    return (t != 0).to(device=device, dtype=torch.uint8).contiguous() if t.dtype != torch.uint8 else t.to(device).contiguous()


def _ident(t):
    """What makes two map arguments 'the same input': the storage they view (data pointer, shape, strides, dtype, device) --
    not the Python object (full_map[0] is a new view object on every call)."""
    if t is None:
        return None
    if isinstance(t, torch.Tensor):
        return ("t", t.data_ptr(), tuple(t.shape), tuple(t.stride()), t.dtype, str(t.device))
    a = np.asarray(t)
    return ("n", a.__array_interface__["data"][0], a.shape, a.strides, a.dtype.str)


class GeodesicSolver:
    """peanut_goal_t: scratch for one full-map size + the collision-disk radius."""

    def __init__(self, full_h: int, full_w: int, col_rad: int = 4, device="cuda:0"):
        if not torch.cuda.is_available():
            raise _lib.PeanutHipError("goal selection needs a HIP device (no CPU fallback)")
        self._lib = _lib.load()
        self.device = torch.device(device)
        self.H, self.W, self.col_rad = int(full_h), int(full_w), int(col_rad)
        self.begun_matches = 0            # selects that took over the field a select_begin had started (same storage)
        self._begun = None
        self._h = C.c_void_p()
        # (an empty default_options block = the option lock: a handle snapshots the process defaults while it is created, and must
        # not do so in the middle of another thread's `with default_options(...)`)
        with _lib.default_options(), torch.cuda.device(self.device):
            _lib.check(self._lib.peanut_goal_create(C.byref(self._h), self.H, self.W, self.col_rad), "peanut_goal_create")

    def __del__(self):
        h = getattr(self, "_h", None)
        if h is not None and h.value:
            try:
                self._lib.peanut_goal_destroy(h)
            except Exception:  # pragma: no cover
                pass
            self._h = C.c_void_p()

    @property
    def rounds(self) -> int:
        return int(self._lib.peanut_goal_rounds(self._h))

    @property
    def passes(self) -> int:
        return int(self._lib.peanut_goal_passes(self._h))

    @property
    def converged(self) -> bool:
        """False when the last solve stopped at the ordering-pass cap with its last pass still changing tiles."""
        return bool(self._lib.peanut_goal_converged(self._h))

    def reset(self):
        _lib.check(self._lib.peanut_goal_reset(self._h), "peanut_goal_reset")

    def select_begin(self, full_obstacle, collision_map, visited_vis, lmb, loc_rc):
        """``peanut_goal_select_begin`` on the current stream: the map inputs of the ``select`` that follows are complete HERE; its
        traversible map, initialisation and first relaxation rounds start now, on the solver's own stream, beside what is
        enqueued after this call (the prediction forward that produces ``target_pred``).  The converted inputs are kept so that
        ``select`` hands the library the same buffers."""
        obst = full_obstacle.to(self.device, torch.float32).contiguous()
        col, vis = _u8(collision_map, self.device), _u8(visited_vis, self.device)
        bounds = (C.c_int * 4)(*[int(v) for v in lmb])
        with torch.cuda.device(self.device):
            rc = self._lib.peanut_goal_select_begin(self._h, obst.data_ptr(), None if col is None else col.data_ptr(),
                                                    None if vis is None else vis.data_ptr(), C.byref(bounds), int(loc_rc[0]), int(loc_rc[1]),
                                                    _lib.current_stream_ptr(self.device))
        _lib.check(rc, "peanut_goal_select_begin")
        self._begun = (tuple(_ident(t) for t in (full_obstacle, collision_map, visited_vis)), (obst, col, vis))

    def traversible(self, full_obstacle: torch.Tensor, collision_map=None, visited_vis=None) -> torch.Tensor:
        """agent_state.py:382-386 -> uint8 [H,W] (1 = traversible)."""
        obst = full_obstacle.to(self.device, torch.float32).contiguous()
        col, vis = _u8(collision_map, self.device), _u8(visited_vis, self.device)
        out = torch.empty((self.H, self.W), dtype=torch.uint8, device=self.device)
        with torch.cuda.device(self.device):
            rc = self._lib.peanut_goal_traversible(self._h, obst.data_ptr(), None if col is None else col.data_ptr(),
                                                   None if vis is None else vis.data_ptr(), out.data_ptr(),
                                                   _lib.current_stream_ptr(self.device))
        _lib.check(rc, "peanut_goal_traversible")
        return out

    def distance(self, traversible: torch.Tensor, goal=None, goal_mask=None, fill_max_plus_one: bool = False) -> torch.Tensor:
        """Geodesic distance (cells, float64 [H,W]) to ``goal`` = (r, c) or to the cells of ``goal_mask``; masked /
        unreachable cells: +inf, or max + 1 (``ma.filled(dd, np.max(dd) + 1)``)."""
        trav = _u8(traversible, self.device)
        gm = _u8(goal_mask, self.device)
        out = torch.empty((self.H, self.W), dtype=torch.float64, device=self.device)
        gr, gc = (-1, -1) if goal is None else (int(goal[0]), int(goal[1]))
        with torch.cuda.device(self.device):
            rc = self._lib.peanut_fmm_distance(self._h, trav.data_ptr(), None if gm is None else gm.data_ptr(), gr, gc,
                                               int(fill_max_plus_one), out.data_ptr(), _lib.current_stream_ptr(self.device))
        _lib.check(rc, "peanut_fmm_distance")
        self._note_convergence()
        return out

    def _note_convergence(self):
        """The ordering passes of the second-order stage run until a pass changes nothing, under a hard ceiling (library
        option ``fmm_max_passes``, default 24: the agent's maps need 6-10); a solve that hit the ceiling with its last pass
        still changing tiles returns the last iterate, not the fixed point (measured effect of stopping after six: a few
        hundredths of a cell).  Said once per solver instead of passing silently."""
        if not self.converged and not getattr(self, "_warned_unconverged", False):
            import warnings
            self._warned_unconverged = True
            warnings.warn(f"GeodesicSolver: the ordering passes stopped at their cap ({self.passes}) before reaching their fixed "
                          "point; the field is the last iterate (raise the library option fmm_max_passes to iterate further)")

    def select(self, full_obstacle, collision_map, visited_vis, lmb, loc_rc, target_pred, dist_weight_temperature: float,
               map_resolution: int, want_dist: bool = False, want_value: bool = False):
        """One ``update_global_goal`` evaluation -> dict(goal=(r, c), value_max, wt_sum, kept_last, rounds[, dist, value])."""
        tp = None if target_pred is None else target_pred.to(self.device, torch.float32).contiguous()
        # `begun` stays referenced until the C call has returned: the side stream may still be reading its buffers.  The inputs of
        # select_begin are recognised by storage (data pointer, shape, strides, dtype, device), not by object identity: Agent_State
        # passes full_map[0], a fresh view object on every call
        begun, self._begun = getattr(self, "_begun", None), None
        if begun is not None and begun[0] == tuple(_ident(t) for t in (full_obstacle, collision_map, visited_vis)):
            obst, col, vis = begun[1]          # the buffers select_begin handed over (a converted input is converted once)
            self.begun_matches += 1
        else:
            obst = full_obstacle.to(self.device, torch.float32).contiguous()
            col, vis = _u8(collision_map, self.device), _u8(visited_vis, self.device)
        lw, lh = int(lmb[1] - lmb[0]), int(lmb[3] - lmb[2])
        if tp is not None and tuple(tp.shape) != (lw, lh):
            raise ValueError(f"target_pred must be [{lw},{lh}], got {tuple(tp.shape)}")
        bounds = (C.c_int * 4)(*[int(v) for v in lmb])
        goal, stats = (C.c_int * 2)(), (C.c_double * 4)()
        dist = torch.empty((self.H, self.W), dtype=torch.float64, device=self.device) if want_dist else None
        value = torch.empty((lw, lh), dtype=torch.float64, device=self.device) if want_value else None
        with torch.cuda.device(self.device):
            rc = self._lib.peanut_goal_select(self._h, obst.data_ptr(), None if col is None else col.data_ptr(),
                                              None if vis is None else vis.data_ptr(), C.byref(bounds), int(loc_rc[0]), int(loc_rc[1]),
                                              None if tp is None else tp.data_ptr(), float(dist_weight_temperature), int(map_resolution),
                                              C.byref(goal), C.byref(stats), None if dist is None else dist.data_ptr(),
                                              None if value is None else value.data_ptr(), _lib.current_stream_ptr(self.device))
        _lib.check(rc, "peanut_goal_select")
        self._note_convergence()
        out = dict(goal=(int(goal[0]), int(goal[1])), value_max=float(stats[0]), wt_sum=float(stats[1]), kept_last=bool(stats[2]),
                   rounds=int(stats[3]), passes=self.passes, converged=self.converged)
        if want_dist:
            out["dist"] = dist
        if want_value:
            out["value"] = value
        return out


class GoalSelector:
    """``update_global_goal`` with its cross-call state (agent_state.py:376-415)."""

    def __init__(self, args, full_hw, device="cuda:0"):
        self.args = args
        self.solver = GeodesicSolver(full_hw[0], full_hw[1], int(args.col_rad), device=device)
        self.reset()

    def reset(self):
        self.solver.reset()
        self.last_global_goal = None
        self.global_goals = None
        self.last = None

    def update(self, full_obstacle, lmb, loc_rc, target_pred, collision_map=None, visited_vis=None, **want):
        args = self.args
        res = self.solver.select(full_obstacle, collision_map, visited_vis, lmb, loc_rc, target_pred,
                                 float(getattr(args, "dist_weight_temperature", 500)), int(args.map_resolution), **want)
        self.last = res
        new_global_goal = [res["goal"]]
        if new_global_goal != self.last_global_goal:          # avoid repeating the last goal (:412-415)
            self.last_global_goal = self.global_goals
            self.global_goals = new_global_goal
        return self.global_goals


def get_mask(sx, sy, scale, step_size):
    """fmm_planner.py:8-22."""
    size = int(step_size // scale) * 2 + 1
    mask = np.zeros((size, size))
    for i in range(size):
        for j in range(size):
            d2 = ((i + 0.5) - (size // 2 + sx)) ** 2 + ((j + 0.5) - (size // 2 + sy)) ** 2
            if (step_size - 1) ** 2 < d2 <= step_size ** 2:
                mask[i, j] = 1
    mask[size // 2, size // 2] = 1
    return mask


def get_dist(sx, sy, scale, step_size):
    """fmm_planner.py:25-36."""
    size = int(step_size // scale) * 2 + 1
    mask = np.zeros((size, size)) + 1e-10
    for i in range(size):
        for j in range(size):
            d2 = ((i + 0.5) - (size // 2 + sx)) ** 2 + ((j + 0.5) - (size // 2 + sy)) ** 2
            if d2 <= step_size ** 2:
                mask[i, j] = max(5, d2 ** 0.5)
    return mask


class FMMPlanner():
    """nav/agent/utils/fmm_planner.py:39-116 for ``scale == 1`` (the only value the agent uses,
    agent_helper.py:374-376): the geodesic field comes from the HIP solver and stays on the device
    (``fmm_dist_dev``); ``fmm_dist`` is its host copy, fetched on first use, for ``get_short_term_goal``'s
    (2 du + 1)^2 window arithmetic, which is kept as the reference's NumPy."""

    def __init__(self, traversible, scale=1, step_size=5, solver: Optional[GeodesicSolver] = None, device="cuda:0"):
        if scale != 1:
            raise NotImplementedError("scale != 1 needs cv2.resize in the reference; the agent always passes 1")
        self.scale = scale
        self.step_size = step_size
        self.traversible = traversible
        h, w = traversible.shape
        self.solver = solver if solver is not None else GeodesicSolver(h, w, 0, device=device)
        self.du = int(self.step_size / (self.scale * 1.))
        self.fmm_dist_dev = None
        self._host = None

    @property
    def fmm_dist(self):
        if self._host is None and self.fmm_dist_dev is not None:
            self._host = self.fmm_dist_dev.cpu().numpy()
        return self._host

    def set_goal(self, goal, auto_improve=False):
        if auto_improve:
            raise NotImplementedError("auto_improve (_find_nearest_goal) is planner glue outside the hot path")
        goal_x, goal_y = int(goal[0] / (self.scale * 1.)), int(goal[1] / (self.scale * 1.))
        self.fmm_dist_dev = self.solver.distance(self.traversible, goal=(goal_x, goal_y), fill_max_plus_one=True)
        self._host = None

    def set_multi_goal(self, goal_map):
        self.fmm_dist_dev = self.solver.distance(self.traversible, goal_mask=(torch.as_tensor(np.asarray(goal_map)) == 1)
                                                 if not isinstance(goal_map, torch.Tensor) else (goal_map == 1), fill_max_plus_one=True)
        self._host = None

    def get_short_term_goal(self, state):
        """fmm_planner.py:77-116, statement for statement."""
        scale = self.scale * 1.
        state = [x / scale for x in state]
        dx, dy = state[0] - int(state[0]), state[1] - int(state[1])
        mask = get_mask(dx, dy, scale, self.step_size)
        dist_mask = get_dist(dx, dy, scale, self.step_size)
        state = [int(x) for x in state]
        fmm_dist = self.fmm_dist
        dist = np.pad(fmm_dist, self.du, 'constant', constant_values=fmm_dist.shape[0] ** 2)
        subset = dist[state[0]:state[0] + 2 * self.du + 1, state[1]:state[1] + 2 * self.du + 1]
        assert subset.shape[0] == 2 * self.du + 1 and subset.shape[1] == 2 * self.du + 1, \
            "Planning error: unexpected subset shape {}".format(subset.shape)
        subset *= mask
        subset += (1 - mask) * fmm_dist.shape[0] ** 2
        distance = subset[self.du, self.du]
        stop = bool(subset[self.du, self.du] < 0.25 * 100 / 5.)
        subset -= subset[self.du, self.du]
        ratio1 = subset / dist_mask
        subset[ratio1 < -1.5] = 1
        (stg_x, stg_y) = np.unravel_index(np.argmin(subset), subset.shape)
        replan = bool(subset[stg_x, stg_y] > -0.0001)
        return (stg_x + state[0] - self.du) * scale, (stg_y + state[1] - self.du) * scale, distance, stop, replan
