#!/usr/bin/env python3
"""Where one step of config 4 goes on the HOST side (round 5): the loop of tools/bench_pipeline.py with the detector, each section
of a step timed on its own between device synchronisations (so the sections do not overlap: their sum is an upper bound of the
step), next to the un-instrumented step time.  python tools/step_profile.py [frames]"""
import json
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import bench_pipeline as bp  # noqa: E402
from peanut_amd.agent_helper import preprocess_obs  # noqa: E402
from peanut_amd.agent_state import Agent_State, default_args  # noqa: E402
from peanut_amd.rcnn_weights import RcnnCfg, make_seeded_rcnn_state_dict  # noqa: E402
from peanut_amd.segmentation import HipDetector  # noqa: E402
from peanut_amd.weights import PredCfg, make_seeded_state_dict  # noqa: E402


def main():
    frames = int(sys.argv[1]) if len(sys.argv) > 1 else 60
    dev = torch.device("cuda", 0)
    args = default_args(only_explore=0, sem_gpu_id=0, pred_precision="fp32", select_goal=True)
    st = Agent_State(args, state_dict=make_seeded_state_dict(PredCfg(), 0))
    rcfg = RcnnCfg(score_thresh_test=0.5)
    det = HipDetector(rcfg, make_seeded_rcnn_state_dict(rcfg, 0), device=dev)
    ep = bp.synth_episode(1000, frames, dev)
    for fr in ep:
        for k in ("masks", "classes", "scores"):
            fr.pop(k)
    acc = {}

    def timed(name, fn):
        torch.cuda.synchronize()
        t = time.perf_counter()
        r = fn()
        torch.cuda.synchronize()
        acc.setdefault(name, []).append((time.perf_counter() - t) * 1e3)
        return r

    for rep in range(2):
        st.reset()
        acc.clear()
        for i, fr in enumerate(ep):
            bgr = timed("flip", lambda: fr["rgb"].flip(-1))
            sem = timed("detector.semantic", lambda: det.semantic(bgr, args.num_sem_categories - 1, args.sem_pred_prob_thr, args.goal_thr, 3))
            obs = timed("preprocess_obs", lambda: preprocess_obs(fr["rgb"], fr["depth"], sem, args))
            infos = {"sensor_pose": fr["sensor_pose"], "goal_cat_id": 3}
            if i == 0:
                st.init_with_obs(obs, infos)
            # labelled by what update_state DID: besides every update_goal_freq-th step it predicts whenever the agent is within
            # goal_reached_dist of its long-term goal (agent_state.py:240-245), which the synthetic episodes hit often
            predicted = timed("update_state", lambda: st.update_state(obs, infos))
            acc.setdefault("update_state (prediction step)" if predicted else "update_state (plain step)", []).append(acc["update_state"].pop())
        acc.pop("update_state", None)
    out = {k: {"n": len(v), "mean_ms": round(sum(v) / len(v), 4)} for k, v in acc.items()}
    per_step = sum(sum(v) for v in acc.values()) / frames
    rec = {"sections": out, "sum_per_step_ms": round(per_step, 3)}
    if "--fine" in sys.argv:
        rec["plain_step_parts"] = plain_step_parts(st, ep, args, det)
    print(json.dumps(rec))


def plain_step_parts(st, ep, args, det):
    """The plain (no prediction) update_state split at its own statements (agent_state.py: update_local_map), each between device
    synchronisations, and the same statements timed on the host only (no synchronisation: what the CPU spends enqueueing)."""
    parts, host = {}, {}

    def timed(name, fn):
        torch.cuda.synchronize()
        t = time.perf_counter()
        r = fn()
        h = (time.perf_counter() - t) * 1e3
        torch.cuda.synchronize()
        parts.setdefault(name, []).append((time.perf_counter() - t) * 1e3)
        host.setdefault(name, []).append(h)
        return r

    st.reset()
    for i, fr in enumerate(ep):
        sem = det.semantic(fr["rgb"].flip(-1), args.num_sem_categories - 1, args.sem_pred_prob_thr, args.goal_thr, 3)
        obs = preprocess_obs(fr["rgb"], fr["depth"], sem, args)
        infos = {"sensor_pose": fr["sensor_pose"], "goal_cat_id": 3}
        if i == 0:
            st.init_with_obs(obs, infos)
        pred_step = (st.step % args.update_goal_freq == args.update_goal_freq - 1 or st.step == 0)
        if pred_step or st.l_step == args.num_local_steps - 1:
            st.update_state(obs, infos)
            continue
        st.goal_cat = 3
        st.poses = timed("upload_pose", lambda: st._upload_pose(infos["sensor_pose"]))
        timed("map_step", lambda: st._map_step(obs))
        locs = timed("pose_readback", lambda: st.local_pose.cpu().numpy())
        r, c = locs[1], locs[0]
        loc_r, loc_c = int(r * 100.0 / args.map_resolution), int(c * 100.0 / args.map_resolution)
        st.planner_pose_inputs[:3] = locs + st.origins
        timed("mark_agent", lambda: st._mark_agent(loc_r, loc_c, 2, [(loc_r, loc_c)]))
        st.loc_r, st.loc_c = loc_r, loc_c
        st.inc_step()
    return {k: {"n": len(v), "synced_ms": round(sum(v) / len(v), 4), "host_ms": round(sum(host[k]) / len(v), 4)} for k, v in parts.items()}


if __name__ == "__main__":
    main()
