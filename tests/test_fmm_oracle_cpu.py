"""The scikit-fmm restatement (oracle/fmm_ref.c; PARITY UNPINNED -- scikit-fmm itself is absent) against what IS
published about `skfmm.distance`: the docstring example of scikit-fmm's `distance` (3x3, phi = 1 with a -1 centre),
plus properties of the eikonal solution, and the goal-selection restatement (oracle/goal_ref.py) against the golden
episode produced by the reference's own Agent_State.update_global_goal (oracle/gen_golden_goal.py)."""
import os

import numpy as np
from numpy import ma

from oracle import fmm_ref, goal_ref


def test_skfmm_docstring_example():
    """scikit-fmm, skfmm/pfmm.py, docstring of distance():
        >>> phi = np.ones((3, 3)); phi[1, 1] = -1
        >>> skfmm.distance(phi)
        array([[ 1.20710678,  0.5       ,  1.20710678],
               [ 0.5       , -0.35355339,  0.5       ],
               [ 1.20710678,  0.5       ,  1.20710678]])"""
    phi = np.ones((3, 3))
    phi[1, 1] = -1
    want = np.array([[1.20710678, 0.5, 1.20710678], [0.5, -0.35355339, 0.5], [1.20710678, 0.5, 1.20710678]])
    np.testing.assert_allclose(fmm_ref.distance(phi), want, atol=5e-9)


def test_point_source_free_space():
    t = ma.masked_values(np.ones((161, 161)), 0)
    t[80, 80] = 0
    d = fmm_ref.distance(t, dx=1)
    yy, xx = np.mgrid[:161, :161]
    e = np.hypot(yy - 80, xx - 80)
    assert not ma.is_masked(d)
    assert np.abs(d[80, 80:100] - np.arange(20.0)).max() < 1e-9      # along the axes the scheme is exact
    assert np.abs(d - e).max() < 0.35                              # second order: well under half a cell
    assert np.abs(fmm_ref.distance(t, dx=1, order=1) - e).max() > 1.0
    assert np.allclose(d, d.T) and np.allclose(d, d[::-1]) and np.allclose(d, d[:, ::-1])


def test_masked_corridor_and_unreachable_pocket():
    trav = np.zeros((40, 60))
    trav[5, 5:50] = 1          # corridor east ...
    trav[5:30, 49] = 1         # ... then south
    trav[35:38, 10:20] = 1     # pocket the front cannot reach
    t = ma.masked_values(trav * 1, 0)
    t[5, 5] = 0
    d = fmm_ref.distance(t, dx=1)
    assert abs(d[5, 49] - 44.0) < 1e-9 and abs(d[29, 49] - 68.0) < 1e-9    # path length around the corner
    m = ma.getmaskarray(d)
    assert m[36, 15] and m[0, 0] and not m[5, 30]                  # unreached / masked cells come back masked
    filled = goal_ref.fmm_set_goal(trav, (5, 5))
    assert abs(filled[36, 15] - 69.0) < 1e-9 and filled[0, 0] == filled[36, 15]      # ma.filled(dd, max + 1)
    gm = np.zeros_like(trav)
    gm[5, 5] = gm[29, 49] = 1
    d2 = goal_ref.fmm_set_multi_goal(trav, gm)
    assert abs(d2[5, 49] - 24.0) < 1e-9 and abs(d2[5, 27] - 22.0) < 1e-9


def test_goal_restatement_matches_reference_golden(golden_dir):
    """oracle/goal_ref.GoalSelector replays the golden episode's goal sequence given the recorded distance field of
    the last prediction step (the full episode replay is the GPU test's job; here: the weighting / argmax / 'avoid the
    last goal' bookkeeping on a synthetic value map)."""
    z = np.load(os.path.join(golden_dir, "goal_golden.npz"))
    assert list(z["pred_steps"]) == sorted(z["pred_steps"]) and len(z["global_goals"]) == len(z["pred_steps"])
    dd = z["last_dd_f32"].astype(np.float64)
    assert (dd >= 0).sum() == int(z["dd_reach"][-1])
    from types import SimpleNamespace
    sel = goal_ref.GoalSelector(SimpleNamespace(col_rad=4, dist_weight_temperature=500, map_resolution=5), (64, 64))
    obst = np.zeros((64, 64), np.float32)
    obst[20:44, 30] = 1.0                                          # a wall between the agent and the better target
    tp = np.zeros((32, 32))
    tp[5, 28] = 1.0                                                # beyond the wall: far geodesically
    tp[28, 4] = 0.9                                                # same side as the agent
    zero = np.zeros((64, 64))
    g = sel.update(obst, (16, 48, 16, 48), (16, 2), tp, zero, zero)
    assert [tuple(int(v) for v in x) for x in g] == [(5, 28)] or [tuple(int(v) for v in x) for x in g] == [(28, 4)]
    w_far, w_near = sel.dd_wt[5, 28], sel.dd_wt[28, 4]
    assert (w_far * 1.0 > w_near * 0.9) == (tuple(int(v) for v in g[0]) == (5, 28))
    g2 = sel.update(obst, (16, 48, 16, 48), (16, 2), tp, zero, zero)          # same argmax again: goal list unchanged
    assert g2 == g


def test_adjacent_equal_seeds_use_the_second_order_term():
    """updatePointOrderTwo admits the second upwind neighbour when it is NOT FARTHER than the first (`<=`): next to two
    adjacent seeds the cell in line with them solves (9/4) u^2 = 1 -> 2/3, where a lone seed gives the first-order 1.
    (FMMPlanner.set_multi_goal's goal blobs, fmm_planner.py:67-75; the region feeds get_short_term_goal's stop test.)"""
    t = ma.masked_values(np.ones((9, 21)), 0)
    t[4, 9] = 0
    t[4, 10] = 0                                      # two seeds side by side in a row
    d = fmm_ref.distance(t, dx=1)
    assert d[4, 9] == 0 and d[4, 10] == 0
    assert abs(d[4, 11] - 2.0 / 3.0) < 1e-12 and abs(d[4, 8] - 2.0 / 3.0) < 1e-12     # in line with the pair
    assert abs(d[3, 9] - 1.0) < 1e-12 and abs(d[5, 10] - 1.0) < 1e-12                 # across the pair: one seed per axis
    lone = ma.masked_values(np.ones((9, 21)), 0)
    lone[4, 9] = 0
    dl = fmm_ref.distance(lone, dx=1)
    assert abs(dl[4, 10] - 1.0) < 1e-12 and abs(dl[4, 8] - 1.0) < 1e-12 and abs(dl[3, 9] - 1.0) < 1e-12
    # mirror symmetry about the pair (column c <-> 19 - c) holds to a few hundredths of a cell only: the j = -1 direction is
    # examined first and its value2 survives (the library's loop), and equal keys leave the heap in insertion order
    assert np.abs(d[:, 0:9] - d[:, 19:10:-1]).max() < 0.05
