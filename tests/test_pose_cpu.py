"""H-3 host logic on CPU: the pose bookkeeping of PEANUT_Agent (peanut_amd/pose.py) against golden vectors produced by
the reference's own ``PEANUT_Agent.get_info`` (oracle/gen_golden_pose.py), and the recorded-episode file format."""
import os

import numpy as np

from peanut_amd import episodes as E
from peanut_amd.pose import PoseTracker, get_rel_pose_change


def _run(gps, compass):
    tr = PoseTracker()
    return np.stack([np.asarray([float(v) for v in tr.get_info({"gps": gps[i].copy(), "compass": compass[i].copy()})
                                 ["sensor_pose"]]) for i in range(len(gps))])


def test_pose_change_float64_readings_bit_exact(golden_dir):
    z = np.load(os.path.join(golden_dir, "pose_golden.npz"))
    got = _run(z["f64_gps"], z["f64_compass"])
    assert np.array_equal(got, z["f64_sensor_pose"])          # float64 arithmetic is NumPy-version independent
    assert (z["f64_compass"] > np.pi).any(), "fixture must exercise the compass wrap"


def test_pose_change_float32_readings(golden_dir):
    """Habitat's float32 readings: the reference's NumPy 1.x forms the distance in float64, the golden was generated
    under NumPy 2 (all float32) -- one float32 rounding apart at most: tolerance 2 ulp of the step length."""
    z = np.load(os.path.join(golden_dir, "pose_golden.npz"))
    for e in range(int(z["n_episodes"])):
        got = _run(z[f"ep{e}_gps"], z[f"ep{e}_compass"])
        ref = z[f"ep{e}_sensor_pose"]
        assert np.array_equal(got[0], [0, 0, 0])
        assert np.abs(got - ref).max() <= 2.5e-7 * max(1.0, np.abs(ref).max()), np.abs(got - ref).max()
        assert np.array_equal(got[:, 2], ref[:, 2])           # `do` is one float32 subtraction either way


def test_pose_helpers():
    dx, dy, do = get_rel_pose_change((1.0, 1.0, 0.5), (0.0, 0.0, 0.0))
    assert abs(dx - 1.0) < 1e-12 and abs(dy - 1.0) < 1e-12 and do == 0.5


def test_recorded_episode_roundtrip(tmp_path):
    rng = np.random.RandomState(0)
    frames = []
    for t in range(3):
        n = t                    # 0, 1, 2 instances
        frames.append(dict(rgb=rng.randint(0, 255, (8, 12, 3)).astype(np.uint8), depth=rng.rand(8, 12, 1).astype(np.float32),
                           gps=rng.rand(2).astype(np.float32), compass=rng.rand(1).astype(np.float32),
                           objectgoal=np.array([t]), instances=(rng.rand(n, 8, 12) > 0.5, np.arange(n), rng.rand(n))))
    p = str(tmp_path / "ep.npz")
    E.save_episode(p, frames)
    ep = E.load_episode(p)
    obs = list(E.iter_observations(ep))
    assert len(obs) == 3
    for f, o in zip(frames, obs):
        for k in E.KEYS:
            assert np.array_equal(np.asarray(f[k]).reshape(o[k].shape), o[k])
        assert np.array_equal(o["instances"][0].astype(bool), f["instances"][0])
        assert o["instances"][1].dtype == np.int32 and len(o["instances"][2]) == len(f["instances"][1])
    np.savez(str(tmp_path / "bad.npz"), rgb=np.zeros((1, 2, 2, 3), np.uint8))
    try:
        E.load_episode(str(tmp_path / "bad.npz"))
        raise AssertionError("expected ValueError")
    except ValueError:
        pass
