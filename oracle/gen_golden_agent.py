"""ORACLE support (test infrastructure): golden vectors for the hot-path CALLERS in Agent_State
(nav/agent/agent_state.py: init_with_obs :115-150, init_map_and_pose :181-210, update_local_map
:268-300, update_full_map :303-338, update_prediction :345-373, inc_step) by running the reference's
own class on a seeded synthetic episode.

``agent_state.py`` imports gym, skfmm and skimage at module level (all absent here, none used by the
methods above except ``skimage.morphology.disk`` in the ctor); they are replaced by empty import-time
stubs plus a 4-line ``disk`` following scikit-image's definition.  ``agent.prediction`` (mmcv/mmseg) is
replaced by a deterministic fake model, because these fixtures pin the crop / pad / mask / window
bookkeeping around the model, not the model.  Runs in the build container only."""
from __future__ import annotations

import os
import sys
import types
import numpy as np
import torch

from oracle import mapping_scenes, ref_import
from oracle.agent_ref import FakePrediction, agent_args, disk

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLDEN = os.path.join(ROOT, "tests", "golden")


def load_reference_agent_state():
    nav = os.path.join(ref_import.REF, "nav")
    if nav not in sys.path:
        sys.path.insert(0, nav)
    for name in ("gym", "skfmm", "skimage"):
        sys.modules.setdefault(name, types.ModuleType(name))
    morph = types.ModuleType("skimage.morphology")
    morph.disk = disk
    sys.modules["skimage.morphology"] = morph
    sys.modules["skimage"].morphology = morph
    fake_pred = types.ModuleType("agent.prediction")
    fake_pred.PEANUT_Prediction_Model = FakePrediction
    import matplotlib
    matplotlib.use("Agg")
    import agent  # noqa: F401  (namespace package of the reference)
    sys.modules["agent.prediction"] = fake_pred
    from agent.agent_state import Agent_State  # type: ignore
    return Agent_State


def drive(state, frames, goal_cat=2, record=None):
    """The perception half of PEANUT_Agent.act / Agent_State.update_state (peanut_agent.py:38-68,
    agent_state.py:213-245) without goal selection and planning."""
    args = state.args
    state.reset()
    for i, fr in enumerate(frames):
        obs = torch.from_numpy(mapping_scenes.frame_to_obs(fr))[None].to(state.device)
        infos = {"sensor_pose": [float(v) for v in fr["pose"]], "goal_cat_id": goal_cat}
        if i == 0:
            state.init_with_obs(obs, infos)
        state.goal_cat = infos["goal_cat_id"]
        state.poses = torch.from_numpy(np.asarray(infos["sensor_pose"])).float().to(state.device)
        state.update_local_map(obs)
        if state.l_step == args.num_local_steps - 1:
            state.l_step = 0
            state.update_full_map()
        predicted = False
        if (state.step % args.update_goal_freq == args.update_goal_freq - 1 or state.step == 0
                or state.dist_to_goal < args.goal_reached_dist) and state.step >= args.switch_step:
            state.update_prediction()
            predicted = True
        if record is not None:
            record(i, state, predicted)
        state.inc_step()


def _episode(Agent_State, over, seed, n_frames, goal_cat):
    args = agent_args(**over)
    st = Agent_State(args)
    st.prediction_model = FakePrediction(args.prediction_window)
    frames = mapping_scenes.make_sequence(seed=seed, n_frames=n_frames)
    # make the agent travel: larger forward motion so that the local window is re-centred at step 19/39
    for f in frames:
        f["pose"][0] = np.float32(f["pose"][0] * 3.0)
    rec = dict(lmb=[], loc=[], sums=[], pred_steps=[], pred_sum=[], pred_sq=[], poses=[])
    last_pred = {}

    def record(i, s, predicted):
        rec["lmb"].append(np.array(s.lmb, np.int64))
        rec["loc"].append(np.array([s.loc_r, s.loc_c], np.int64))
        rec["sums"].append(s.local_map.double().sum((1, 2)).cpu().numpy())
        rec["poses"].append(s.local_pose.cpu().numpy().copy())
        if predicted:
            rec["pred_steps"].append(i)
            tp = np.asarray(s.target_pred, np.float64)
            rec["pred_sum"].append(tp.sum())
            rec["pred_sq"].append((tp * tp).sum())
            last_pred["v"] = np.asarray(s.target_pred, np.float32).copy()
            last_pred["step"] = i

    drive(st, frames, goal_cat=goal_cat, record=record)
    full = st.full_map.cpu().numpy()
    idx = np.flatnonzero(full)
    out = {"seed": np.int64(seed), "n_frames": np.int64(n_frames), "goal_cat": np.int64(goal_cat),
           "lmb": np.stack(rec["lmb"]), "loc": np.stack(rec["loc"]), "channel_sums": np.stack(rec["sums"]),
           "local_pose": np.stack(rec["poses"]), "pred_steps": np.array(rec["pred_steps"], np.int64),
           "pred_sum": np.array(rec["pred_sum"]), "pred_sq": np.array(rec["pred_sq"]),
           "last_target_pred": last_pred["v"],
           "last_pred_step": np.int64(last_pred["step"]),
           "full_idx": idx.astype(np.int32), "full_val": full.reshape(-1)[idx].astype(np.float32)}
    stats = dict(frames=n_frames, prediction_steps=rec["pred_steps"],
                 lmb_changes=int((np.diff(np.stack(rec["lmb"]), axis=0) != 0).any(1).sum()), full_nnz=int(idx.size))
    print(f"[agent_state] {over or 'defaults'}: {n_frames} frames, predictions at {rec['pred_steps']}, lmb changed "
          f"{stats['lmb_changes']}x, full-map nnz {idx.size}")
    return out, stats


# nav/arguments.py values other than the defaults (round 3): a prediction window SMALLER than the local map (crop instead
# of pad, agent_state.py:355-361), shorter local / goal-update periods, another goal category
VARIANT = dict(prediction_window=400, num_local_steps=10, update_goal_freq=7, goal_reached_dist=50)


def generate(report):
    Agent_State = load_reference_agent_state()
    out, stats = _episode(Agent_State, {}, 7, 45, 2)
    np.savez_compressed(os.path.join(GOLDEN, "agent_state_golden.npz"), **out)
    report["agent_state"] = stats
    generate_variant(report, Agent_State)


def generate_variant(report, Agent_State=None):
    Agent_State = Agent_State or load_reference_agent_state()
    out, stats = _episode(Agent_State, VARIANT, 8, 36, 5)
    for k, v in VARIANT.items():
        out[f"arg_{k}"] = np.int64(v)
    np.savez_compressed(os.path.join(GOLDEN, "agent_state_golden_v2.npz"), **out)
    report["agent_state_v2"] = stats
