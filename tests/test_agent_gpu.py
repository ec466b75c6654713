"""Hot-path callers on the device against the golden episode produced
by the reference's own Agent_State (tests/golden/agent_state_golden.npz, oracle/gen_golden_agent.py)
and against the NumPy restatement of the observation formatting (oracle/agent_ref.py)."""
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


class FakePredictionGPU:
    """Device twin of oracle.agent_ref.FakePrediction (same formula on HIP tensors)."""

    def __init__(self, pattern):
        self.pattern = torch.from_numpy(pattern).cuda()

    def get_prediction_batch(self, maps, apply_sigmoid=True, out=None):
        sel = maps[:, [0, 1, 4, 5, 6, 7]]
        return torch.tanh(sel + self.pattern[None]) * 0.5 + 0.5


@pytest.mark.parametrize("golden", ["agent_state_golden.npz", "agent_state_golden_v2.npz"])
def test_agent_state_episode_matches_reference(golden_dir, golden):
    """Default nav/arguments.py values, and (v2) a prediction window smaller than the local map (the crop branch of
    agent_state.py:355-361), 10-step local periods, goal updates every 7 steps, another goal category."""
    from oracle import mapping_scenes
    from oracle.agent_ref import agent_args, fake_pattern
    from peanut_amd.agent_state import Agent_State
    z = np.load(os.path.join(golden_dir, golden))
    args = agent_args(**{k[4:]: int(z[k]) for k in z.files if k.startswith("arg_")})
    st = Agent_State(args, prediction_model=FakePredictionGPU(fake_pattern(size=args.prediction_window)))
    frames = mapping_scenes.make_sequence(seed=int(z["seed"]), n_frames=int(z["n_frames"]))
    for f in frames:
        f["pose"][0] = np.float32(f["pose"][0] * 3.0)
    goal = int(z["goal_cat"])
    st.reset()
    pred_steps, last_pred = [], None
    for i, fr in enumerate(frames):
        obs = torch.from_numpy(mapping_scenes.frame_to_obs(fr))[None].cuda()
        infos = {"sensor_pose": [float(v) for v in fr["pose"]], "goal_cat_id": goal}
        if i == 0:
            st.init_with_obs(obs, infos)
        step_before = st.step
        predicted = st.update_state(obs, infos)
        assert st.step == step_before + 1
        assert list(st.lmb) == list(z["lmb"][i]), f"step {i}: local map boundaries"
        assert [st.loc_r, st.loc_c] == list(z["loc"][i]), f"step {i}: agent cell"
        np.testing.assert_allclose(st.local_pose.cpu().numpy(), z["local_pose"][i], rtol=0, atol=2e-5)
        sums = st.local_map.double().sum((1, 2)).cpu().numpy()
        np.testing.assert_allclose(sums, z["channel_sums"][i], rtol=1e-6, atol=0.05, err_msg=f"step {i}")
        if predicted:
            pred_steps.append(i)
            tp = st.target_pred.double()
            k = len(pred_steps) - 1
            assert abs(float(tp.sum()) - z["pred_sum"][k]) <= 1e-5 * abs(z["pred_sum"][k]) + 0.05
            assert abs(float((tp * tp).sum()) - z["pred_sq"][k]) <= 1e-5 * abs(z["pred_sq"][k]) + 0.05
            last_pred = st.target_pred.cpu().numpy()
    assert pred_steps == list(z["pred_steps"])
    assert int(z["last_pred_step"]) == pred_steps[-1]
    assert np.abs(last_pred - z["last_target_pred"]).max() <= 1e-4
    # full_map has the same write history on both sides (update_full_map at the end of each local period,
    # update_prediction's write-back at every prediction step), so it is compared as is
    full = st.full_map.cpu().numpy().reshape(-1)
    ref = np.zeros_like(full)
    ref[z["full_idx"]] = z["full_val"]
    # 5e-5 is the per-step bound of tests/test_mapping_gpu.py (device sin/cos vs SLEEF in the pose); fractional cells
    # that were warped again and again carry it forward: 6.0e-5 on 31 cells of the 36-frame v2 episode, whose poses move
    # three times as far per step
    assert np.abs(full - ref).max() <= (5e-5 if golden == "agent_state_golden.npz" else 1e-4)


def test_preprocess_obs_matches_reference_loop():
    from types import SimpleNamespace
    from oracle.agent_ref import preprocess_obs_ref
    from peanut_amd.agent_helper import preprocess_obs
    rng = np.random.RandomState(0)
    H, W, ncat = 480, 640, 10
    depth = rng.uniform(0.0, 1.1, size=(H, W, 1)).astype(np.float32)
    depth[rng.uniform(size=(H, W, 1)) < 0.1] = 0.0          # scattered invalid pixels
    depth[:, 100:140] = 0.0                                  # fully invalid columns (> 90 %)
    depth[:470, 300:320] = 0.0                               # mostly invalid columns with a few valid rows
    depth[:, 500:520][depth[:, 500:520] > 0.5] = 0.995       # too-far pixels
    rgb = rng.randint(0, 256, size=(H, W, 3)).astype(np.uint8)
    sem = (rng.uniform(size=(H, W, ncat)) > 0.9).astype(np.float32)
    args = SimpleNamespace(env_frame_width=W, frame_width=160, min_depth=0.5, max_depth=5.0)
    ref = preprocess_obs_ref(rgb.astype(np.float32), depth.copy(), sem, args).astype(np.float32)
    got = preprocess_obs(torch.from_numpy(rgb).cuda(), torch.from_numpy(depth).cuda(), torch.from_numpy(sem).cuda(), args)
    assert got.shape == (1, 14, 120, 160)
    assert np.array_equal(got[0].cpu().numpy(), ref)         # bit-exact (fp32, same operation order)


def test_replay_loop_runs_full_pipeline():
    """collect.py order on synthetic raw frames: seg accumulation -> obs formatting -> projection ->
    prediction (real HIP PSPNet, seeded weights) every update_goal_freq steps."""
    from oracle.agent_ref import agent_args
    from peanut_amd.agent_state import Agent_State
    from peanut_amd.replay import run_episode
    from peanut_amd.weights import PredCfg, make_seeded_state_dict
    args = agent_args(only_explore=0, prediction_window=240, map_size_cm=2400)
    sd = make_seeded_state_dict(PredCfg(), 0)
    st = Agent_State(args, state_dict=sd)
    g = torch.Generator().manual_seed(0)
    frames = []
    for i in range(12):
        depth = torch.full((480, 640, 1), 0.3) + torch.rand((480, 640, 1), generator=g) * 0.01   # ~1.85 m wall
        masks = torch.zeros((3, 480, 640), dtype=torch.bool)
        masks[0, 280:400, 100:220] = True     # below the horizon: lands in the agent-height z range
        masks[1, 150:260, 300:420] = True
        masks[2, 10:60, 500:600] = True
        frames.append(dict(rgb=torch.randint(0, 256, (480, 640, 3), generator=g, dtype=torch.uint8).cuda(),
                           depth=depth.cuda(), masks=masks.cuda(), classes=torch.tensor([1, 4, 7]).cuda(),
                           scores=torch.tensor([0.99, 0.97, 0.5]).cuda(), sensor_pose=[0.1, 0.0, 0.05 if i % 3 else 0.0]))
    n_pred = run_episode(st, frames, goal_cat=1)
    assert n_pred == 2                                        # steps 0 and 9
    tp = st.target_pred
    assert tp.shape == (240, 240) and bool(torch.isfinite(tp).all())
    assert float(tp.max()) <= 1.0 and float(tp.min()) >= 0.0
    assert float(st.local_map[1].sum()) > 0                   # something was explored
    assert float(st.local_map[4 + 1].sum()) > 0               # class-1 instance reached the map
    assert float(st.local_map[4 + 7].sum()) == 0              # score 0.5 < 0.95 was gated out


def test_replay_loop_with_the_hip_detector():
    """Same loop with the detector in it: frames carry only rgb + depth, a (small, seeded) MaskRCNN produces the
    instance masks -- the map state must equal the run where the same detector's outputs were precomputed and fed
    as canned masks (the loop adds nothing but the BGR flip of segmentation.py:44)."""
    from oracle.agent_ref import agent_args
    from peanut_amd.agent_state import Agent_State
    from peanut_amd.rcnn_weights import RcnnCfg, make_seeded_rcnn_state_dict
    from peanut_amd.replay import run_episode
    from peanut_amd.segmentation import HipDetector
    from peanut_amd.weights import PredCfg, make_seeded_state_dict
    # thresholds sit inside the score range these seeded weights produce (0.20 .. 0.23 on the antialiased 128 x 171
    # frames), so that the score gate cuts through the detections
    rcfg = RcnnCfg(depth=50, min_size=128, max_size=256, rpn_pre_nms_topk=60, rpn_post_nms_topk=40,
                   detections_per_image=10, score_thresh_test=0.15)
    det = HipDetector(rcfg, make_seeded_rcnn_state_dict(rcfg, 7))
    args = agent_args(only_explore=0, prediction_window=240, map_size_cm=2400, sem_pred_prob_thr=0.205, goal_thr=0.212)
    sd = make_seeded_state_dict(PredCfg(), 0)
    g = torch.Generator().manual_seed(3)
    raw = []
    for i in range(4):
        depth = torch.full((480, 640, 1), 0.3) + torch.rand((480, 640, 1), generator=g) * 0.01
        raw.append(dict(rgb=torch.randint(0, 256, (480, 640, 3), generator=g, dtype=torch.uint8).cuda(), depth=depth.cuda(),
                        sensor_pose=[0.1, 0.0, 0.0]))
    canned = []
    for fr in raw:
        m, c, s = det(fr["rgb"].flip(-1))
        assert m.shape[1:] == (480, 640) and len(c) == len(s) == m.shape[0] > 0
        canned.append(dict(fr, masks=m, classes=c, scores=s))
    a = Agent_State(args, state_dict=sd)
    b = Agent_State(args, state_dict=sd)
    assert run_episode(a, raw, goal_cat=5, detector=det) == run_episode(b, canned, goal_cat=5) == 1
    assert float(a.local_map[4:].sum()) > 0                 # detections reached the map
    assert torch.equal(a.local_map, b.local_map) and torch.equal(a.target_pred, b.target_pred)
    with pytest.raises(ValueError):
        run_episode(a, raw, goal_cat=5)                       # no masks and no detector


def test_agent_loop_from_gps_compass_matches_reference(golden_dir):
    """H-3: ``PEANUT_Agent.act`` driven by recorded (gps, compass, objectgoal) readings over two episodes back to
    back -- pose change from the sensors, hm3d -> coco goal mapping, per-episode reset, map update every step,
    prediction at step 0 and every 10th step -- against the reference's own PEANUT_Agent.get_info + Agent_State
    (tests/golden/pose_golden.npz, oracle/gen_golden_pose.py)."""
    from oracle import mapping_scenes
    from oracle.agent_ref import agent_args, fake_pattern
    from peanut_amd.peanut_agent import PEANUT_Agent
    z = np.load(os.path.join(golden_dir, "pose_golden.npz"))
    args = agent_args()
    agent = PEANUT_Agent(args, prediction_model=FakePredictionGPU(fake_pattern(size=args.prediction_window)))
    for e in range(int(z["n_episodes"])):
        frames = mapping_scenes.make_sequence(seed=int(z[f"ep{e}_seed"]), n_frames=int(z[f"ep{e}_n"]))
        agent.reset()
        assert agent.total_episodes == e + 1 and agent.last_sim_location is None
        st = agent.agent_states
        for i, fr in enumerate(frames):
            observations = {"gps": z[f"ep{e}_gps"][i], "compass": z[f"ep{e}_compass"][i],
                            "objectgoal": np.array([int(z[f"ep{e}_goal"])]),
                            "obs": torch.from_numpy(mapping_scenes.frame_to_obs(fr))[None]}
            out = agent.act(observations)
            np.testing.assert_allclose(np.asarray(out["sensor_pose"], np.float64), z[f"ep{e}_sensor_pose"][i], rtol=0,
                                       atol=3e-7)
            assert out["predicted"] == bool(z[f"ep{e}_predicted"][i]), f"episode {e} step {i}: prediction schedule"
            assert list(st.lmb) == list(z[f"ep{e}_lmb"][i]), f"episode {e} step {i}: local map boundaries"
            assert [st.loc_r, st.loc_c] == list(z[f"ep{e}_loc"][i]), f"episode {e} step {i}: agent cell"
            np.testing.assert_allclose(st.local_pose.cpu().numpy(), z[f"ep{e}_local_pose"][i], rtol=0, atol=3e-5)
            sums = st.local_map.double().sum((1, 2)).cpu().numpy()
            np.testing.assert_allclose(sums, z[f"ep{e}_channel_sums"][i], rtol=1e-6, atol=0.05, err_msg=f"ep {e} step {i}")
        assert st.goal_cat == {0: 0, 1: 3, 2: 2, 3: 4, 4: 5, 5: 1}[int(z[f"ep{e}_goal"])]
        full = st.full_map.cpu().numpy().reshape(-1)
        ref = np.zeros_like(full)
        ref[z[f"ep{e}_full_idx"]] = z[f"ep{e}_full_val"]
        assert np.abs(full - ref).max() <= 5e-5, f"episode {e}: full map (a stale map would mean reset() failed)"


def test_recorded_episode_files_drive_the_replay(tmp_path):
    """The on-disk (rgb, depth, gps, compass, objectgoal [+ instances]) episode format feeds replay.run_recorded_shard
    with collect.py's --start_ep/--end_ep window; timestep_limit stops the agent like peanut_agent.py:41-43."""
    from oracle.agent_ref import agent_args
    from peanut_amd import episodes as E
    from peanut_amd.peanut_agent import PEANUT_Agent
    from peanut_amd.replay import run_recorded_shard
    rng = np.random.RandomState(3)
    paths = []
    for e in range(3):
        frames = []
        x, o = 0.0, 0.0
        for t in range(12):
            x += 0.1
            masks = np.zeros((2, 480, 640), np.uint8)
            masks[0, 280:400, 100:220] = 1
            masks[1, 300:420, 300:420] = 1
            frames.append(dict(rgb=rng.randint(0, 255, (480, 640, 3)).astype(np.uint8),
                               depth=np.full((480, 640, 1), 0.4, np.float32), gps=np.array([x, 0.0], np.float32),
                               compass=np.array([o], np.float32), objectgoal=np.array([e]),
                               instances=(masks, np.array([0, 3], np.int32), np.array([0.99, 0.97], np.float32))))
        p = str(tmp_path / f"ep{e}.npz")
        E.save_episode(p, frames)
        paths.append(p)
    args = agent_args(only_explore=0, prediction_window=240, map_size_cm=2400, timestep_limit=11)
    from peanut_amd.weights import PredCfg, make_seeded_state_dict
    agent = PEANUT_Agent(args, state_dict=make_seeded_state_dict(PredCfg(), seed=0))
    seen = []
    done = run_recorded_shard(agent, paths, start_ep=1, end_ep=3, on_episode=lambda i, n: seen.append(i))
    assert sorted(done) == [1, 2] and seen == [1, 2]
    assert all(n == 2 for n in done.values())            # steps 0 and 9 of the 11 acted frames (the 12th is past the limit)
    assert agent.total_episodes == 2 and agent.timestep == 12
    assert float(agent.agent_states.local_map[4].sum()) > 0 and agent.agent_states.target_pred is not None


def test_map_bookkeeping_in_one_launch_equals_the_tensor_operations():
    """peanut_map_mark_agent (Agent_State._mark_agent) against the reference's statements (agent_state.py:281-296) run as
    torch operations, bit for bit: in the middle of the map, at its borders (Python's slice clamping, torch's wrapping of
    negative indices), with and without the second footprint (goal reached); indices torch refuses raise IndexError before anything is written."""
    from oracle.agent_ref import agent_args
    from peanut_amd.agent_state import Agent_State
    args = agent_args()
    st = Agent_State(args, prediction_model=None)
    m, off = st.local_w, int(args.col_rad + 1)
    g = torch.Generator().manual_seed(3)
    base = torch.rand((st.nc, m, m), generator=g).cuda()

    def reference(lm, loc_r, loc_c, centres):
        lm[2, :, :].fill_(0.)
        lm[2:4, loc_r - 2:loc_r + 3, loc_c - 2:loc_c + 3] = 1.
        for r, c in centres:
            lm[1][st._selem_r - off + r, st._selem_c - off + c] = 1.
        return lm

    cases = [((240, 240), None), ((100, 377), (300, 20)), ((off, off), None), ((m - 1 - off, m - 1 - off), (off, m - 1 - off)),
             ((1, 2), None),            # the footprint wraps to the far rows / columns; the square's slice starts negative: empty
             ((0, m - off - 1), None), ((3, 3), (2, 1)), ((m - 1 - off, 0), None)]
    for (loc, goal) in cases:
        centres = [loc] + ([goal] if goal else [])
        want = reference(base.clone(), loc[0], loc[1], centres)
        st.local_map = base.clone()
        st._mark_agent(loc[0], loc[1], 2, centres)
        assert torch.equal(st.local_map, want), (loc, goal)
    st.local_map = base.clone()
    with pytest.raises(IndexError):        # (a footprint row >= m: torch's device-side index check would abort the process here)
        st._mark_agent(m - 2, 100, 2, [(m - 2, 100)])
    assert torch.equal(st.local_map, base)
