"""ORACLE support (test infrastructure): golden vectors for H-3 -- the pose bookkeeping of PEANUT_Agent
(nav/agent/peanut_agent.py:70-95 get_info / get_sim_location / get_pose_change over
nav/agent/utils/pose.py:11-21) and the episode loop order of nav/collect.py:44-59 (reset per episode, map update
every step, prediction at step 0 and every update_goal_freq steps) -- produced by the reference's OWN classes:

* ``PEANUT_Agent`` is imported from /root/reference/nav/agent/peanut_agent.py under import-time stubs for
  ``habitat`` (base class only) and ``agent.agent_helper`` (cv2 / detectron2 / skimage; its only use on this path is
  ``reset``), and its ``get_info`` is what turns the (gps, compass) readings into ``sensor_pose``;
* ``Agent_State`` is the reference's own (as in oracle/gen_golden_agent.py), with the fake prediction model.

The frames are oracle/mapping_scenes.py sequences; their per-frame motion is integrated into simulator
(gps, compass) readings here, and only the readings are what the agent sees.  Runs in the build container only.

    python -m oracle.gen_golden_pose
"""
from __future__ import annotations

import math
import os
import sys
import types

import numpy as np
import torch

from oracle import gen_golden_agent, mapping_scenes, ref_import
from oracle.agent_ref import FakePrediction, agent_args

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLDEN = os.path.join(ROOT, "tests", "golden")


def load_reference_agent():
    Agent_State = gen_golden_agent.load_reference_agent_state()
    hab = types.ModuleType("habitat")
    hab.Agent = object
    hab.Config = object
    sys.modules.setdefault("habitat", hab)
    helper = types.ModuleType("agent.agent_helper")

    class Agent_Helper:          # stub: only reset() is reached on this path
        def __init__(self, *a, **k):
            pass

        def reset(self):
            pass

    helper.Agent_Helper = Agent_Helper
    sys.modules["agent.agent_helper"] = helper
    from agent.peanut_agent import PEANUT_Agent  # type: ignore
    return PEANUT_Agent, Agent_State


def readings_from_motion(frames, x0=1.5, y0=-0.75, o0=0.4, dtype=np.float32):
    """Integrate per-frame agent-frame motion (dx, dy, do) into Habitat-style sensor readings:
    gps = (x, -y), compass = o wrapped to [0, 2 pi).  The first reading is the start pose."""
    x, y, o = float(x0), float(y0), float(o0)
    gps, comp = [], []
    for i, f in enumerate(frames):
        if i > 0:
            dx, dy, do = (float(v) for v in f["pose"])
            x, y = x + dx * math.cos(o) - dy * math.sin(o), y + dx * math.sin(o) + dy * math.cos(o)
            o = o + do
        gps.append([x, -y])
        comp.append([o % (2 * math.pi)])
    return np.asarray(gps, dtype), np.asarray(comp, dtype)


def generate(report):
    PEANUT_Agent, Agent_State = load_reference_agent()
    args = agent_args()
    ag = object.__new__(PEANUT_Agent)            # the ctor needs a habitat task config; set the fields it sets
    ag.agent_states = Agent_State(args)
    ag.agent_states.prediction_model = FakePrediction(args.prediction_window)
    ag.agent_helper = sys.modules["agent.agent_helper"].Agent_Helper()
    ag.last_sim_location = None
    ag.first_obs = True
    ag.total_episodes = 0
    ag.args = args
    ag.timestep = 0
    st = ag.agent_states
    episodes = [dict(seed=11, n=45, goal=1, scale=3.0, o0=0.4), dict(seed=12, n=30, goal=4, scale=2.0, o0=5.9)]
    out = {"n_episodes": np.int64(len(episodes))}
    for e, spec in enumerate(episodes):
        frames = mapping_scenes.make_sequence(seed=spec["seed"], n_frames=spec["n"])
        for f in frames:
            f["pose"][0] = np.float32(f["pose"][0] * spec["scale"])
        gps, comp = readings_from_motion(frames, o0=spec["o0"])
        ag.reset()                                # peanut_agent.py:29-36 (Agent_State.reset + pose reset)
        rec = dict(sp=[], lmb=[], loc=[], sums=[], poses=[], pred=[])
        for i, fr in enumerate(frames):
            observations = {"gps": gps[i].copy(), "compass": comp[i].copy(), "objectgoal": np.array([spec["goal"]])}
            info = ag.get_info(observations)      # peanut_agent.py:70-75
            rec["sp"].append(np.asarray([float(v) for v in info["sensor_pose"]], np.float64))
            obs = torch.from_numpy(mapping_scenes.frame_to_obs(fr))[None].to(st.device)
            info["goal_cat_id"] = {0: 0, 1: 3, 2: 2, 3: 4, 4: 5, 5: 1}[spec["goal"]]
            if ag.first_obs:
                st.init_with_obs(obs, info)
                ag.first_obs = False
            # perception half of Agent_State.update_state (agent_state.py:213-245)
            st.goal_cat = info["goal_cat_id"]
            st.poses = torch.from_numpy(np.asarray(info["sensor_pose"])).float().to(st.device)
            st.update_local_map(obs)
            if st.l_step == args.num_local_steps - 1:
                st.l_step = 0
                st.update_full_map()
            predicted = False
            if (st.step % args.update_goal_freq == args.update_goal_freq - 1 or st.step == 0
                    or st.dist_to_goal < args.goal_reached_dist) and st.step >= args.switch_step:
                st.update_prediction()
                predicted = True
            rec["lmb"].append(np.array(st.lmb, np.int64))
            rec["loc"].append(np.array([st.loc_r, st.loc_c], np.int64))
            rec["sums"].append(st.local_map.double().sum((1, 2)).cpu().numpy())
            rec["poses"].append(st.local_pose.cpu().numpy().copy())
            rec["pred"].append(predicted)
            st.inc_step()
        full = st.full_map.cpu().numpy()
        idx = np.flatnonzero(full)
        out.update({f"ep{e}_seed": np.int64(spec["seed"]), f"ep{e}_n": np.int64(spec["n"]), f"ep{e}_goal": np.int64(spec["goal"]),
                    f"ep{e}_scale": np.float64(spec["scale"]), f"ep{e}_gps": gps, f"ep{e}_compass": comp,
                    f"ep{e}_sensor_pose": np.stack(rec["sp"]), f"ep{e}_lmb": np.stack(rec["lmb"]),
                    f"ep{e}_loc": np.stack(rec["loc"]), f"ep{e}_channel_sums": np.stack(rec["sums"]),
                    f"ep{e}_local_pose": np.stack(rec["poses"]), f"ep{e}_predicted": np.array(rec["pred"]),
                    f"ep{e}_full_idx": idx.astype(np.int32), f"ep{e}_full_val": full.reshape(-1)[idx].astype(np.float32)})
        print(f"[pose] episode {e}: {spec['n']} frames, predictions at {list(np.flatnonzero(rec['pred']))}, "
              f"full-map nnz {idx.size}")
    # pose arithmetic alone, float64 readings (NumPy-version independent) incl. the compass wrap
    rng = np.random.RandomState(5)
    g64 = np.cumsum(rng.uniform(-0.3, 0.3, size=(64, 2)), 0)
    c64 = (np.cumsum(rng.uniform(-0.6, 0.6, size=(64, 1)), 0) + 3.0) % (2 * math.pi)
    ag.last_sim_location = None
    sp64 = [np.asarray([float(v) for v in ag.get_info({"gps": g64[i].copy(), "compass": c64[i].copy()})["sensor_pose"]])
            for i in range(64)]
    out.update(f64_gps=g64, f64_compass=c64, f64_sensor_pose=np.stack(sp64))
    np.savez_compressed(os.path.join(GOLDEN, "pose_golden.npz"), **out)
    report["pose"] = dict(episodes=[dict(s) for s in episodes], numpy=np.__version__)


if __name__ == "__main__":
    rep = {}
    generate(rep)
    print(rep)
