"""Habitat-free counterpart of the reference's agent facade ``PEANUT_Agent`` (nav/agent/peanut_agent.py:15-95):
same ``reset`` / ``act`` / ``get_info`` / ``get_sim_location`` / ``get_pose_change`` surface and call order,
driven by recorded observation tuples instead of a live ``habitat.Env``.

Per step (peanut_agent.py:38-68): pose change from ``gps`` / ``compass`` -> goal id through ``hm3d_to_coco``
(constants.py:23-31) -> segmentation + observation formatting (``Agent_Helper.preprocess_inputs``,
agent_helper.py:166-195; here the HIP detector / mask accumulation / ``peanut_preprocess_obs``) ->
``Agent_State.init_with_obs`` on the first frame -> ``Agent_State.update_state``.  The local FMM planner that turns
the planner inputs into a motor action (``Agent_Helper.plan_act``, agent_helper.py:130-159) needs the simulator to
close the loop and is not part of the hot path; ``act`` therefore returns the planner inputs themselves."""
from __future__ import annotations

from typing import Callable, Dict, Optional

import numpy as np
import torch

from .agent_helper import preprocess_obs
from .agent_state import Agent_State
from .pose import PoseTracker
from .segmentation import accumulate_instances

# nav/constants.py:21-31
hm3d_names = {0: "chair", 1: "bed", 2: "plant", 3: "toilet", 4: "tv_monitor", 5: "sofa"}
hm3d_to_coco = {0: 0, 1: 3, 2: 2, 3: 4, 4: 5, 5: 1}


class PEANUT_Agent:
    def __init__(self, args, task_config=None, detector: Optional[Callable] = None, prediction_model=None,
                 state_dict=None):
        """``detector``: callable(img_bgr uint8 [H,W,3] HIP tensor) -> (masks, classes, scores), e.g.
        ``peanut_amd.segmentation.HipDetector``; without one the observations must carry canned ``instances``
        (peanut_amd/episodes.py) or a ready ``obs`` tensor."""
        self.args = args
        self.agent_states = Agent_State(args, prediction_model=prediction_model, state_dict=state_dict)
        self.device = self.agent_states.device
        self.detector = detector
        self.pose = PoseTracker()
        self.first_obs = True
        self.total_episodes = 0
        self.timestep = 0
        self.goal_cat = -1

    # ---- peanut_agent.py:29-36 ----
    def reset(self):
        self.agent_states.reset()
        self.pose.reset()
        self.first_obs = True
        self.step = 0
        self.timestep = 0
        self.total_episodes += 1

    @property
    def last_sim_location(self):
        return self.pose.last_sim_location

    # ---- peanut_agent.py:70-95 ----
    def get_info(self, obs):
        return self.pose.get_info(obs)

    def get_sim_location(self, obs):
        return self.pose.get_sim_location(obs)

    def get_pose_change(self, obs):
        return self.pose.get_pose_change(obs)

    # ---- Agent_Helper.preprocess_inputs (agent_helper.py:166-195) on the device ----
    def _preprocess(self, observations: Dict, goal_cat: int) -> torch.Tensor:
        if "obs" in observations:
            return observations["obs"].to(self.device)
        args = self.args
        rgb = torch.as_tensor(np.ascontiguousarray(observations["rgb"])).to(self.device)
        depth = torch.as_tensor(np.ascontiguousarray(observations["depth"], dtype=np.float32)).to(self.device)
        if "instances" in observations:
            masks, classes, scores = (torch.as_tensor(np.ascontiguousarray(a)).to(self.device) for a in observations["instances"])
        elif self.detector is not None:
            masks, classes, scores = self.detector(rgb.flip(-1))                       # RGB -> BGR (segmentation.py:44)
        else:
            raise ValueError("observation carries no instances and the agent has no detector")
        sem = accumulate_instances(masks, classes, scores, args.num_sem_categories - 1, args.sem_pred_prob_thr,
                                   args.goal_thr, goal_cat)
        return preprocess_obs(rgb, depth, sem, args)

    # ---- peanut_agent.py:38-68 ----
    def act(self, observations: Dict):
        self.timestep += 1
        if self.timestep > getattr(self.args, "timestep_limit", 499):     # always stop at episode end
            return {'action': 0}
        goal = int(np.asarray(observations['objectgoal']).reshape(-1)[0])
        info = self.get_info(observations)
        info['goal_name'] = hm3d_names[goal]
        goal = hm3d_to_coco[goal]
        self.goal_cat = goal                                               # Agent_Helper.set_goal_cat
        obs = self._preprocess(observations, goal)
        info['goal_cat_id'] = goal
        if self.first_obs:
            self.agent_states.init_with_obs(obs, info)
            self.first_obs = False
        predicted = self.agent_states.update_state(obs, info)
        st = self.agent_states
        return {'predicted': bool(predicted), 'sensor_pose': info['sensor_pose'], 'goal_name': info['goal_name'],
                'pose_pred': st.planner_pose_inputs.copy(), 'global_goals': [list(g) for g in st.global_goals]}
