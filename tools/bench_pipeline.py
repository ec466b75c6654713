#!/usr/bin/env python3
"""Config 4 of BASELINE.json on synthetic frames: the full per-step perception loop of the agent
(instance-mask accumulation -> observation formatting -> map projection -> every update_goal_freq steps
the 720x720 map-prediction forward), one episode per rank, episodes sharded over the GPUs of the node
like the reference's --start_ep/--end_ep.  By default frames carry synthetic instance masks/classes/scores in
the detector's output format; with --detector the Mask R-CNN R-101-FPN itself (peanut_amd.rcnn, seeded random
weights, score threshold lowered so that it returns detections) runs on every 640x480 frame as well.

    python tools/bench_pipeline.py [--episodes 8] [--frames 100] [--detector]
    python -m torch.distributed.run --nproc-per-node N tools/bench_pipeline.py ...
"""
import argparse
import json
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from peanut_amd import dist as pdist  # noqa: E402
from peanut_amd.agent_state import Agent_State  # noqa: E402
from peanut_amd.replay import episode_shard, run_episode  # noqa: E402
from peanut_amd.weights import PredCfg, make_seeded_state_dict  # noqa: E402


def synth_episode(seed, n_frames, dev):
    g = torch.Generator().manual_seed(seed)
    frames = []
    for i in range(n_frames):
        depth = torch.full((480, 640, 1), 0.35) + torch.rand((480, 640, 1), generator=g) * 0.02
        depth[200:330, 150:330] = 0.15 + 0.01 * torch.rand((130, 180, 1), generator=g)       # a closer box
        masks = torch.zeros((4, 480, 640), dtype=torch.bool)
        masks[0, 260:420, 80:260] = True
        masks[1, 200:330, 150:330] = True
        masks[2, 300:460, 400:560] = True
        masks[3, 20:90, 500:620] = True
        frames.append(dict(rgb=torch.randint(0, 256, (480, 640, 3), generator=g, dtype=torch.uint8).to(dev),
                           depth=depth.to(dev), masks=masks.to(dev), classes=torch.tensor([0, 3, 5, 8]).to(dev),
                           scores=torch.tensor([0.99, 0.97, 0.96, 0.4]).to(dev),
                           sensor_pose=[0.25 if i % 4 else 0.0, 0.0, 0.0 if i % 4 else 0.5236]))
    return frames


def run_pipeline(episodes, frames, precision="fp32", detector=False, goal=True, dev=None, rank=0, world=1, goal_overlap=True):
    """The timed loop; returns the result dict on every rank (rank 0's is the one to print).  goal_overlap (the agent's default):
    the geodesic field of the goal selection runs next to the prediction forward (peanut_goal_select_begin); False: one after the
    other, and the goal selection is timed on its own (one extra device synchronisation per prediction step)."""
    dev = dev or torch.device("cuda", torch.cuda.current_device())
    from peanut_amd.agent_state import default_args   # nav/arguments.py defaults
    args = default_args(only_explore=0, sem_gpu_id=dev.index, pred_precision=precision, select_goal=goal, goal_overlap=goal_overlap)
    st = Agent_State(args, state_dict=make_seeded_state_dict(PredCfg(), 0))
    goal_ms, goal_n, goal_rounds, goal_passes, goal_unconverged = [0.0], [0], [0], [0], [0]
    spans = []          # (event before update_prediction, event after update_global_goal): the pair's device time, no host sync
    if goal:
        inner, inner_pred = st.update_global_goal, st.update_prediction

        def timed_pred(**kw):
            e0 = torch.cuda.Event(enable_timing=True)
            e0.record()
            spans.append([e0, None])
            inner_pred(**kw)

        def timed():
            if not goal_overlap:      # serial order: the goal selection on its own (it synchronises anyway: the goal cell goes to the host)
                torch.cuda.synchronize()
            t = time.perf_counter()
            inner()
            goal_ms[0] += (time.perf_counter() - t) * 1e3
            e1 = torch.cuda.Event(enable_timing=True)
            e1.record()
            if spans and spans[-1][1] is None:
                spans[-1][1] = e1
            goal_n[0] += 1
            goal_rounds[0] += st.goal_rounds
            goal_passes[0] += st.goal_passes
            goal_unconverged[0] += 0 if st.goal_converged else 1
        st.update_global_goal = timed
        st.update_prediction = timed_pred
    mine = episode_shard(episodes)
    eps = {e: synth_episode(1000 + e, frames, dev) for e in mine}
    det = None
    if detector:
        from peanut_amd.rcnn_weights import RcnnCfg, make_seeded_rcnn_state_dict
        from peanut_amd.segmentation import HipDetector
        rcfg = RcnnCfg(score_thresh_test=0.5)
        det = HipDetector(rcfg, make_seeded_rcnn_state_dict(rcfg, 0), device=dev, precision=precision)
        for fr in (f for e in mine for f in eps[e]):
            for k in ("masks", "classes", "scores"):
                fr.pop(k)
    if mine:
        run_episode(st, eps[mine[0]][:12], goal_cat=3, detector=det)     # warm-up (plans, workspaces)
    # Garbage of whatever ran in this process before (an earlier run's Agent_State / detector handles, bench.py's other
    # configurations) must not be collected INSIDE the timed loop: a handle's destructor frees device memory, which synchronises the
    # device (measured, round 6: the second of two runs in one process 140 steps/s against 148.5 for the first; profiles/r9f)
    import gc
    gc.collect()
    torch.cuda.synchronize()
    pdist.barrier()
    goal_ms[0], goal_n[0], goal_rounds[0], goal_passes[0], goal_unconverged[0] = 0.0, 0, 0, 0, 0
    del spans[:]
    t0 = time.perf_counter()
    n_pred = 0
    for e in mine:
        n_pred += run_episode(st, eps[e], goal_cat=3, detector=det)
    torch.cuda.synchronize()
    pdist.barrier()
    dt = pdist.max_over_ranks(time.perf_counter() - t0, device=dev)
    steps = episodes * frames
    span_ms = [a.elapsed_time(b) for a, b in spans if b is not None]
    seg = "Mask R-CNN R-101-FPN inference + mask accumulation" if detector else "seg-accumulate (canned instance masks)"
    gtxt = " + long-term goal selection (geodesic field on the 960x960 map)" if goal else ""
    return {"workload": f"config 4: {episodes} synthetic episodes x {frames} frames, {seg} + "
                        f"obs formatting + map projection per step, 720x720 map prediction{gtxt} every 10th step and on every step "
                        f"within goal_reached_dist of the current goal (agent_state.py:240-245)",
            "n_gpus": world, "steps_per_s": round(steps / dt, 1), "ms_per_step": round(dt / steps * 1e3, 3),
            "predictions_rank0": n_pred, "steps_rank0": len(mine) * frames,
            "predictions_per_step": round(n_pred / max(len(mine) * frames, 1), 4), "precision": precision,
            "goal_overlap": bool(goal and goal_overlap),
            "prediction_plus_goal_ms_per_call": round(sum(span_ms) / len(span_ms), 3) if span_ms else None,
            "goal_selection_ms_per_call": round(goal_ms[0] / goal_n[0], 3) if goal_n[0] and not goal_overlap else None,
            "goal_selection_rounds_per_call": round(goal_rounds[0] / goal_n[0], 1) if goal_n[0] else None,
            "goal_selection_passes_per_call": round(goal_passes[0] / goal_n[0], 1) if goal_n[0] else None,
            "goal_selection_calls_unconverged": goal_unconverged[0] if goal_n[0] else None}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--episodes", type=int, default=8)
    ap.add_argument("--frames", type=int, default=100)
    ap.add_argument("--precision", default="fp32")
    ap.add_argument("--detector", action="store_true", help="run Mask R-CNN on every frame instead of canned masks")
    ap.add_argument("--no-goal", action="store_true", help="skip the long-term goal selection (round-1 behaviour of this tool)")
    ap.add_argument("--serial-goal", action="store_true", help="goal selection after the prediction forward instead of next to it")
    a = ap.parse_args()
    rank, local_rank, world = pdist.init_process_group()
    dev = torch.device("cuda", torch.cuda.current_device())
    res = run_pipeline(a.episodes, a.frames, a.precision, a.detector, not a.no_goal, dev, rank, world, goal_overlap=not a.serial_goal)
    if rank == 0:
        print(json.dumps(res))


if __name__ == "__main__":
    main()
