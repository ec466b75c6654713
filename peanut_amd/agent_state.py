"""Device-resident counterpart of the hot-path CALLERS in the reference's ``Agent_State``
(nav/agent/agent_state.py): map/pose bookkeeping around ``Semantic_Mapping`` and
``PEANUT_Prediction_Model``.  Same attribute and method names, same call order, same arithmetic;
differences, all on purpose:

* ``full_map`` / ``local_map`` live on the HIP device for the whole episode and
  ``update_prediction`` crops, predicts, pads and masks ON the device (the reference copies the
  14x720x720 crop to the host, back to the GPU inside ``run_inference`` and the result back again,
  agent_state.py:361 -> prediction.py:128-131,268);
* long-term goal selection (``update_global_goal`` :376-415) runs on the device too (csrc/goal.hip: dilation,
  geodesic field, distance weights and the argmax; scikit-fmm's fast marching replaced by a fixed-point solver of
  the same discretisation).  ``collision_map`` / ``visited_vis``, which the reference
  reads from ``self.helper`` (the CPU planner's bookkeeping), are attributes here (uint8 HIP tensors, zero until a
  planner writes them);
* ``update_goal_map`` (:418-446, scikit-image erosion of the goal category) is CPU planner glue and is not here.

The 12-byte pose read-back per step (`local_pose.cpu()`, :276) is kept: the integer cell indices it
yields drive the host-side decisions exactly as in the reference.  The eight small tensor operations that follow it
(:281-296: clear the location channel, trajectory square, explored-area footprints) are one launch
(``peanut_map_mark_agent``), and the sensor pose goes up through a pinned staging buffer."""
from __future__ import annotations

import ctypes as C

import numpy as np
import torch

from . import _lib
from .mapping import Semantic_Mapping
from .prediction import PEANUT_Prediction_Model


def default_args(**over):
    """The flags of nav/arguments.py the hot-path callers read, at their defaults (``argparse.Namespace``), for
    drivers that run without the reference's argument parser (tools/bench_pipeline.py)."""
    from argparse import Namespace
    a = dict(seed=1, cuda=False, sem_gpu_id=0, num_sem_categories=10, map_size_cm=4800, map_resolution=5,
             global_downscaling=2, only_explore=1, col_rad=4, grid_resolution=24, num_local_steps=20,
             switch_step=0, update_goal_freq=10, goal_reached_dist=75, prediction_window=720, visualize=0,
             frame_height=120, frame_width=160, env_frame_height=480, env_frame_width=640, vision_range=100,
             hfov=79.0, du_scale=1, cat_pred_threshold=5.0, exp_pred_threshold=1.0, map_pred_threshold=0.1,
             camera_height=0.88, min_depth=0.5, max_depth=5.0, sem_pred_prob_thr=0.95, goal_thr=0.985,
             dist_weight_temperature=500, timestep_limit=499)
    a.update(over)
    return Namespace(**a)


def disk(radius, dtype=np.uint8):
    """``skimage.morphology.disk`` (all pixels with x^2 + y^2 <= r^2); scikit-image is only needed
    for this footprint (agent_state.py:85-86)."""
    L = np.arange(-radius, radius + 1)
    X, Y = np.meshgrid(L, L)
    return np.array((X ** 2 + Y ** 2) <= radius ** 2, dtype=dtype)


class Agent_State:
    """Hot-path subset of ``Agent_State`` (agent_state.py:26-454)."""

    def __init__(self, args, prediction_model=None, state_dict=None):
        self.args = args
        self.device = args.device = torch.device("cuda:" + str(args.sem_gpu_id))
        self.nc = 4 + args.num_sem_categories
        self.map_size = args.map_size_cm // args.map_resolution
        self.full_w, self.full_h = self.map_size, self.map_size
        self.local_w = int(self.full_w / args.global_downscaling)
        self.local_h = int(self.full_h / args.global_downscaling)
        self.full_map = torch.zeros(self.nc, self.full_w, self.full_h, dtype=torch.float32, device=self.device)
        self.local_map = torch.zeros(self.nc, self.local_w, self.local_h, dtype=torch.float32, device=self.device)
        self.full_pose = torch.zeros(3, dtype=torch.float32, device=self.device)
        self.local_pose = torch.zeros(3, dtype=torch.float32, device=self.device)
        self.origins = np.zeros((3))
        self.lmb = np.zeros((4)).astype(int)
        self.planner_pose_inputs = np.zeros((7))
        self.sem_map_module = Semantic_Mapping(args).to(self.device)
        self.sem_map_module.eval()
        if prediction_model is not None:
            self.prediction_model = prediction_model
        elif getattr(args, "only_explore", 0) == 0:
            self.prediction_model = PEANUT_Prediction_Model(args, state_dict=state_dict)
        else:
            self.prediction_model = None
        self.selem = disk(args.col_rad)
        sel = np.where(disk(args.col_rad + 1) > 0)
        self.selem_idx = sel
        self._selem_r = torch.from_numpy(sel[0].astype(np.int64)).to(self.device)
        self._selem_c = torch.from_numpy(sel[1].astype(np.int64)).to(self.device)
        self._selem_mask = torch.from_numpy(np.ascontiguousarray(disk(args.col_rad + 1))).to(self.device)     # uint8 [(2R+1)^2]
        self._pose_host = torch.zeros(3, dtype=torch.float32).pin_memory()      # staging of the per-step sensor pose
        self.target_pred = None
        self.global_goals = [[0, 0]]
        self.dist_to_goal = float("inf")
        # long-term goal selection (agent_state.py:376-415); the planner-side maps live here as HIP tensors
        self.collision_map = torch.zeros((self.full_w, self.full_h), dtype=torch.uint8, device=self.device)
        self.visited_vis = torch.zeros((self.full_w, self.full_h), dtype=torch.uint8, device=self.device)
        self._goal = None
        self.last_global_goal = None
        self.value_max = None

    # ---- agent_state.py:94-105 ----
    def reset(self):
        self.l_step = 0
        self.step = 0
        self.goal_cat = -1
        self.found_goal = False
        self.init_map_and_pose()
        self.target_pred = None
        self.last_global_goal = None
        self.collision_map.zero_()              # Agent_Helper.reset (agent_helper.py:114-115)
        self.visited_vis.zero_()
        if self._goal is not None:
            self._goal.reset()                  # self.dd_wt = None

    # ---- agent_state.py:154-178 ----
    def get_local_map_boundaries(self, agent_loc, local_sizes, full_sizes):
        loc_r, loc_c = agent_loc
        local_w, local_h = local_sizes
        full_w, full_h = full_sizes
        if self.args.global_downscaling > 1:
            gx1, gy1 = loc_r - local_w // 2, loc_c - local_h // 2
            gx1, gy1 = gx1 - gx1 % self.args.grid_resolution, gy1 - gy1 % self.args.grid_resolution
            gx2, gy2 = gx1 + local_w, gy1 + local_h
            if gx1 < 0:
                gx1, gx2 = 0, local_w
            if gx2 > full_w:
                gx1, gx2 = full_w - local_w, full_w
            if gy1 < 0:
                gy1, gy2 = 0, local_h
            if gy2 > full_h:
                gy1, gy2 = full_h - local_h, full_h
        else:
            gx1, gx2, gy1, gy2 = 0, full_w, 0, full_h
        return [gx1, gx2, gy1, gy2]

    def _rebind_local(self):
        """lmb / origins / local view / local pose from the full pose (shared tail of
        init_map_and_pose :197-210 and update_full_map :318-331)."""
        args = self.args
        locs = self.full_pose.cpu().numpy()
        r, c = locs[1], locs[0]
        loc_r, loc_c = [int(r * 100.0 / args.map_resolution), int(c * 100.0 / args.map_resolution)]
        self.lmb = self.get_local_map_boundaries((loc_r, loc_c), (self.local_w, self.local_h),
                                                 (self.full_w, self.full_h))
        self.planner_pose_inputs[3:] = self.lmb
        self.origins = np.array([self.lmb[2] * args.map_resolution / 100.0,
                                 self.lmb[0] * args.map_resolution / 100.0, 0.])
        self.local_map = self.full_map[:, self.lmb[0]:self.lmb[1], self.lmb[2]:self.lmb[3]]
        self.local_pose = self.full_pose - torch.from_numpy(self.origins).to(self.device).float()
        return locs, loc_r, loc_c

    # ---- agent_state.py:181-210 ----
    def init_map_and_pose(self):
        args = self.args
        self.full_map.fill_(0.)
        self.full_pose.fill_(0.)
        self.full_pose[:2] = self.args.map_size_cm / 100.0 / 2.0
        locs = self.full_pose.cpu().numpy()
        self.planner_pose_inputs[:3] = locs
        r, c = locs[1], locs[0]
        loc_r, loc_c = [int(r * 100.0 / args.map_resolution), int(c * 100.0 / args.map_resolution)]
        self.full_map[2:4, loc_r - 1:loc_r + 2, loc_c - 1:loc_c + 2] = 1.0
        self._rebind_local()

    def _map_step(self, obs):
        """``_, self.local_map, _, self.local_pose = self.sem_map_module(obs, self.poses,
        self.local_map, self.local_pose, self)`` (:114-115, :273-274).  The HIP module needs contiguous
        inputs; a view into full_map is copied first, like the reference's `maps_last[None,:]` cat."""
        lm = self.local_map if self.local_map.is_contiguous() else self.local_map.contiguous()
        lp = self.local_pose if self.local_pose.is_contiguous() else self.local_pose.contiguous()
        _, self.local_map, _, self.local_pose = self.sem_map_module(obs, self.poses, lm, lp, self)

    # ---- agent_state.py:108-150 (map part) ----
    def init_with_obs(self, obs, infos):
        self.l_step = 0
        self.step = 0
        self.poses = torch.from_numpy(np.asarray(infos['sensor_pose'])).float().to(self.device)
        self._map_step(obs)
        self.locs = self.local_pose.cpu().numpy()
        r, c = self.locs[1], self.locs[0]
        loc_r, loc_c = [int(r * 100.0 / self.args.map_resolution), int(c * 100.0 / self.args.map_resolution)]
        self.local_map[2:4, loc_r - 1:loc_r + 2, loc_c - 1:loc_c + 2] = 1.
        rgoal = [0.1, 0.1]
        self.global_goals = [[int(rgoal[0] * self.local_w), int(rgoal[1] * self.local_h)]]
        self.global_goals = [[min(x, int(self.local_w - 1)), min(y, int(self.local_h - 1))]
                             for x, y in self.global_goals]

    # ---- agent_state.py:268-300 ----
    def update_local_map(self, obs):
        args = self.args
        self._map_step(obs)
        locs = self.local_pose.cpu().numpy()
        self.planner_pose_inputs[:3] = locs + self.origins
        r, c = locs[1], locs[0]
        loc_r = int(r * 100.0 / args.map_resolution)
        loc_c = int(c * 100.0 / args.map_resolution)
        traj_rad = 2
        self.dist_to_goal = np.sqrt((loc_r - (self.global_goals[0][0])) ** 2 +
                                    (loc_c - (self.global_goals[0][1])) ** 2) * args.map_resolution
        centres = [(loc_r, loc_c)]
        if self.dist_to_goal < args.goal_reached_dist:
            centres.append((self.global_goals[0][0], self.global_goals[0][1]))
        self._mark_agent(loc_r, loc_c, traj_rad, centres)
        self.loc_r = loc_r
        self.loc_c = loc_c

    def _upload_pose(self, sensor_pose):
        """``torch.from_numpy(np.asarray(infos['sensor_pose'])).float().to(device)`` (:225) through one pinned buffer: the copy is
        asynchronous, and the buffer is free again by the next step because update_local_map reads the pose back (a device
        synchronisation) after the projection that consumes it."""
        self._pose_host.copy_(torch.from_numpy(np.asarray(sensor_pose, dtype=np.float64)).float())
        return self._pose_host.to(self.device, non_blocking=True)

    def _mark_agent(self, loc_r, loc_c, traj_rad, centres):
        """(:281-296) in one launch (``peanut_map_mark_agent``)::

            self.local_map[2, :, :].fill_(0.)
            self.local_map[2:4, loc_r - traj_rad:loc_r + traj_rad + 1, loc_c - traj_rad:loc_c + traj_rad + 1] = 1.
            self.local_map[1][self.selem_idx[0] - int(args.col_rad + 1) + r, self.selem_idx[1] - int(args.col_rad + 1) + c] = 1.
                                                                   # for (r, c) in centres: the agent, and the goal once reached

        with Python's slice clamping and torch's index rules (negative wraps, out of range raises IndexError)."""
        lm = self.local_map
        if not lm.is_contiguous():            # (a view into full_map right after _rebind_local; _map_step replaces it)
            raise RuntimeError("local_map must be contiguous here")
        if lm.dtype != torch.float32 or lm.shape[1] != lm.shape[2]:
            # peanut_map_mark_agent takes ONE side length (row / column decomposition, plane stride, index checks) and fp32 planes:
            # the reference's local map is square (agent_state.py:56-60: local_w = local_h); anything else is refused, not mis-indexed
            raise ValueError(f"_mark_agent: the local map must be a square fp32 [C, M, M] tensor, got {tuple(lm.shape)} {lm.dtype}")
        m = int(lm.shape[1])
        rad = int(self.args.col_rad + 1)
        r0, r1, _ = slice(loc_r - traj_rad, loc_r + traj_rad + 1).indices(m)
        c0, c1, _ = slice(loc_c - traj_rad, loc_c + traj_rad + 1).indices(int(lm.shape[2]))
        for cr, cc in centres:
            if cr - rad < -m or cr + rad >= m or cc - rad < -m or cc + rad >= m:
                raise IndexError(f"explored-area footprint around ({cr}, {cc}) leaves the {m} x {m} local map")
        flat = (C.c_int * (2 * len(centres)))(*[int(v) for rc in centres for v in rc])
        with torch.cuda.device(self.device):
            rc = _lib.load().peanut_map_mark_agent(lm.data_ptr(), int(lm.shape[0]), m, r0, r1, c0, c1, self._selem_mask.data_ptr(), rad,
                                                   len(centres), C.byref(flat), _lib.current_stream_ptr(self.device))
        _lib.check(rc, "peanut_map_mark_agent")

    # ---- agent_state.py:303-338 ----
    def update_full_map(self):
        args = self.args
        self.full_map[:, self.lmb[0]:self.lmb[1], self.lmb[2]:self.lmb[3]] = self.local_map
        self.full_pose = self.local_pose + torch.from_numpy(self.origins).to(self.device).float()
        self._rebind_local()
        locs = self.local_pose.cpu().numpy()
        r, c = locs[1], locs[0]
        self.loc_r = int(r * 100.0 / args.map_resolution)
        self.loc_c = int(c * 100.0 / args.map_resolution)

    # ---- agent_state.py:345-373, on the device ----
    def update_prediction(self, goal_follows=False):
        args = self.args
        self.full_map[:, self.lmb[0]:self.lmb[1], self.lmb[2]:self.lmb[3]] = self.local_map
        if goal_follows and getattr(args, "goal_overlap", True):
            # update_global_goal follows at once (update_state, :240-245): its geodesic field needs the map as it is NOW, not
            # the prediction -- begun here, the solver runs it on its own stream beside the forward below (include/peanut_hip.h)
            self._goal_solver().select_begin(self.full_map[0], self.collision_map, self.visited_vis, self.lmb, (self.loc_r, self.loc_c))
        if self.full_w == args.prediction_window and self.full_h == args.prediction_window:
            object_preds = self.prediction_model.get_prediction_batch(self.full_map[None].contiguous())[0]
        else:
            x1 = self.full_w // 2 - args.prediction_window // 2
            x2 = x1 + args.prediction_window
            y1 = self.full_h // 2 - args.prediction_window // 2
            y2 = y1 + args.prediction_window
            crop = self.full_map[:, x1:x2, y1:y2].contiguous()[None]
            preds = self.prediction_model.get_prediction_batch(crop)[0]
            object_preds = torch.zeros((preds.shape[0], self.full_w, self.full_h), dtype=preds.dtype,
                                       device=preds.device)
            object_preds[:, x1:x2, y1:y2] = preds
        target = self.goal_cat
        target_pred = object_preds[target, self.lmb[0]:self.lmb[1], self.lmb[2]:self.lmb[3]]
        target_pred = target_pred * (self.local_map[1] < 0.5)        # unexplored regions only
        self.target_pred = target_pred

    def _goal_solver(self):
        if self._goal is None:
            from .goal import GeodesicSolver
            self._goal = GeodesicSolver(self.full_w, self.full_h, int(self.args.col_rad), device=self.device)
        return self._goal

    # ---- agent_state.py:376-415, on the device ----
    def update_global_goal(self):
        """Geodesic-distance-weighted argmax of the target prediction (csrc/goal.hip).  Needs ``target_pred`` (from
        ``update_prediction``) unless ``dist_weight_temperature == 0``."""
        args = self.args
        res = self._goal_solver().select(self.full_map[0], self.collision_map, self.visited_vis, self.lmb, (self.loc_r, self.loc_c),
                                self.target_pred, float(getattr(args, "dist_weight_temperature", 500)), int(args.map_resolution))
        self.value_max = res["value_max"]
        self.goal_rounds = res["rounds"]
        self.goal_passes, self.goal_converged = self._goal.passes, self._goal.converged
        new_global_goal = [res["goal"]]
        if new_global_goal != self.last_global_goal:      # avoid repeating the last goal
            self.last_global_goal = self.global_goals
            self.global_goals = new_global_goal

    # ---- agent_state.py:449-454 ----
    def inc_step(self):
        args = self.args
        self.l_step += 1
        self.step += 1
        self.l_step = self.step % args.num_local_steps

    # ---- perception half of update_state (agent_state.py:213-245) ----
    def update_state(self, obs, infos):
        """Map update -> (every num_local_steps) full-map update -> (every update_goal_freq steps,
        at step 0, or near the goal) prediction + long-term goal selection (:240-245; ``args.select_goal = False``
        skips the latter).  Returns whether a prediction ran.  ``update_goal_map`` and the planner-input assembly
        (:246-263) are CPU planner glue outside the hot path."""
        args = self.args
        self.goal_cat = infos['goal_cat_id']
        self.poses = self._upload_pose(infos['sensor_pose'])
        self.update_local_map(obs)
        if self.l_step == args.num_local_steps - 1:
            self.l_step = 0
            self.update_full_map()
        predicted = False
        if (self.step % args.update_goal_freq == args.update_goal_freq - 1 or self.step == 0 or
                self.dist_to_goal < args.goal_reached_dist) and self.step >= args.switch_step \
                and self.prediction_model is not None:
            select = getattr(args, "select_goal", True)
            self.update_prediction(goal_follows=select)
            if select:
                self.update_global_goal()
            predicted = True
        self.inc_step()
        return predicted
