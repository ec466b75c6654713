"""ORACLE support (test infrastructure): golden episode for the long-term goal selection
produced by the reference's OWN ``Agent_State.update_global_goal`` (nav/agent/agent_state.py:376-415), imported from
/root/reference as in oracle/gen_golden_agent.py, with its two absent third-party calls bound to stand-ins:
``skimage.morphology.binary_dilation`` -> scipy.ndimage.binary_dilation (what scikit-image itself calls) and
``skfmm.distance`` -> oracle/fmm_ref (restated scikit-fmm; PARITY UNPINNED for that piece).  The collision / visited
maps the planner side would maintain (Agent_Helper, out of scope) are seeded synthetic arrays.

Also asserts that oracle/goal_ref.py (the restatement that travels to the GPU box) reproduces the run exactly.

    python -m oracle.gen_golden_goal
"""
from __future__ import annotations

import os
import sys
import types

import numpy as np
import torch

from oracle import fmm_ref, gen_golden_agent, goal_ref, mapping_scenes
from oracle.agent_ref import FakePrediction, agent_args

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLDEN = os.path.join(ROOT, "tests", "golden")


def helper_maps(shape, seed=21):
    """Synthetic Agent_Helper.collision_map / visited_vis (agent_helper.py:114-115): a few collision strips and a
    visited trail through the map centre."""
    rng = np.random.RandomState(seed)
    col = np.zeros(shape)
    vis = np.zeros(shape)
    c = shape[0] // 2
    for _ in range(6):
        r0, c0 = c + rng.randint(-60, 60), c + rng.randint(-60, 60)
        col[r0:r0 + rng.randint(2, 10), c0:c0 + rng.randint(2, 4)] = 1
    for k in range(40):
        vis[c - 2 + k // 4:c + 1 + k // 4, c - 1 + k:c + 2 + k] = 1
    return col, vis


def generate(report):
    Agent_State = gen_golden_agent.load_reference_agent_state()
    import agent.agent_state as ras          # the reference module: bind its third-party names
    ras.skfmm.distance = fmm_ref.distance
    ras.skimage.morphology.binary_dilation = goal_ref.binary_dilation
    args = agent_args(dist_weight_temperature=500, only_explore=1)
    st = Agent_State(args)
    st.prediction_model = FakePrediction(args.prediction_window)
    helper = types.SimpleNamespace()
    helper.collision_map, helper.visited_vis = helper_maps((st.full_w, st.full_h))
    st.helper = helper
    frames = mapping_scenes.make_sequence(seed=31, n_frames=45)
    for f in frames:
        f["pose"][0] = np.float32(f["pose"][0] * 3.0)
    sel = goal_ref.GoalSelector(args, (st.full_w, st.full_h))
    rec = dict(goals=[], steps=[], dd_probe=[], wt_sum=[], argmax=[], dd_max=[], dd_reach=[])
    probes = [(480, 480), (470, 520), (520, 430), (400, 560), (600, 600), (300, 480)]
    last = {}

    prev = {}

    def record(i, s, predicted):
        cur = (int(s.loc_r + s.lmb[0]), int(s.loc_c + s.lmb[2]))
        goal_ref.mark_visited(helper.visited_vis, prev.get("rc", cur), cur)      # the planner's trail (every step)
        prev["rc"] = cur
        if not predicted:
            return
        s.update_global_goal()                                     # the reference's own method
        mine = sel.update(s.full_map[0].cpu().numpy(), s.lmb, (s.loc_r, s.loc_c), np.asarray(s.target_pred),
                          helper.collision_map, helper.visited_vis)
        assert [tuple(int(v) for v in g) for g in mine] == [tuple(int(v) for v in g) for g in s.global_goals], i
        assert np.array_equal(sel.value, s.value)
        rec["steps"].append(i)
        rec["goals"].append([int(s.global_goals[0][0]), int(s.global_goals[0][1])])
        rec["argmax"].append([int(v) for v in np.unravel_index(s.value.argmax(), s.value.shape)])
        dd = sel.dd
        rec["dd_probe"].append([dd[p] for p in probes])
        fin = np.isfinite(dd)
        rec["dd_max"].append(dd[fin].max())
        rec["dd_reach"].append(int(fin.sum()))
        rec["wt_sum"].append(float(np.sum(s.dd_wt)))
        last.update(dd=dd.copy(), value=np.asarray(s.value).copy(), step=i)

    gen_golden_agent.drive(st, frames, goal_cat=2, record=record)
    dd = last["dd"]
    out = {"seed": np.int64(31), "n_frames": np.int64(45), "goal_cat": np.int64(2), "helper_seed": np.int64(21),
           "pred_steps": np.array(rec["steps"], np.int64), "global_goals": np.array(rec["goals"], np.int64),
           "value_argmax": np.array(rec["argmax"], np.int64), "probes": np.array(probes, np.int64),
           "dd_probe": np.array(rec["dd_probe"]), "dd_max": np.array(rec["dd_max"]), "dd_reach": np.array(rec["dd_reach"], np.int64),
           "wt_sum": np.array(rec["wt_sum"]), "last_step": np.int64(last["step"]),
           "last_dd_f32": np.where(np.isfinite(dd), dd, -1.0).astype(np.float32),       # -1 = masked / unreachable
           "last_value_max": np.float64(last["value"].max())}
    np.savez_compressed(os.path.join(GOLDEN, "goal_golden.npz"), **out)
    report["goal"] = dict(pred_steps=rec["steps"], goals=rec["goals"], reach=rec["dd_reach"])
    print(f"[goal] predictions at {rec['steps']}; goals {rec['goals']}; reachable cells {rec['dd_reach']}")


if __name__ == "__main__":
    rep = {}
    generate(rep)
