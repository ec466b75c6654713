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
            pred_step = (st.step % args.update_goal_freq == args.update_goal_freq - 1 or st.step == 0)
            timed("update_state (prediction step)" if pred_step else "update_state (plain step)", lambda: st.update_state(obs, infos))
    out = {k: {"n": len(v), "mean_ms": round(sum(v) / len(v), 4)} for k, v in acc.items()}
    per_step = sum(sum(v) for v in acc.values()) / frames
    print(json.dumps({"sections": out, "sum_per_step_ms": round(per_step, 3)}))


if __name__ == "__main__":
    main()
