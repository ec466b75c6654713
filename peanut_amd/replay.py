"""Habitat-free replay of the reference's episode/step loop (nav/collect.py:44-84 +
PEANUT_Agent.act, nav/agent/peanut_agent.py:38-68) over recorded -- or synthetic -- frame tuples:
per step  seg mask accumulation -> observation formatting -> map projection -> (every
update_goal_freq steps) map prediction, with episodes sharded over ranks exactly like the
reference's ``--start_ep/--end_ep`` flags.  Planning/acting needs the simulator and is not here."""
from __future__ import annotations

from typing import Callable, Dict, Iterable, List, Optional


from . import dist as pdist
from .agent_helper import preprocess_obs
from .agent_state import Agent_State
from .segmentation import accumulate_instances


def run_episode(state: Agent_State, frames: Iterable[Dict], goal_cat: int,
                on_step: Optional[Callable[[int, Agent_State, bool], None]] = None, detector=None) -> int:
    """frames: dicts with either a ready ``obs`` [1,C,h,w] HIP tensor or the raw tuple
    (``rgb`` u8 [H,W,3], ``depth`` [H,W,1], and instance ``masks``/``classes``/``scores`` unless a ``detector``
    -- e.g. ``peanut_amd.segmentation.HipDetector`` -- is given, which is then run on the BGR-flipped frame as
    ``SemanticPredMaskRCNN.get_prediction`` does, segmentation.py:44-45), plus ``sensor_pose`` = (dx, dy, do)
    (peanut_agent.py:70-95).  Returns the number of predictions."""
    args = state.args
    state.reset()
    n_pred = 0
    for i, fr in enumerate(frames):
        if "obs" in fr:
            obs = fr["obs"]
        else:
            if "masks" not in fr and detector is None:
                raise ValueError("frame carries no instance masks and no detector was given")
            if "masks" not in fr and hasattr(detector, "semantic"):     # detector + accumulation in one library call
                sem = detector.semantic(fr["rgb"].flip(-1), args.num_sem_categories - 1, args.sem_pred_prob_thr, args.goal_thr,
                                        goal_cat)                       # RGB -> BGR
            else:
                if "masks" not in fr:
                    fr = dict(fr)
                    fr["masks"], fr["classes"], fr["scores"] = detector(fr["rgb"].flip(-1))
                sem = accumulate_instances(fr["masks"], fr["classes"], fr["scores"], args.num_sem_categories - 1,
                                           args.sem_pred_prob_thr, args.goal_thr, goal_cat)
            obs = preprocess_obs(fr["rgb"], fr["depth"], sem, args)
        infos = {"sensor_pose": fr["sensor_pose"], "goal_cat_id": goal_cat}
        if i == 0:
            state.init_with_obs(obs, infos)
        predicted = state.update_state(obs, infos)
        n_pred += int(predicted)
        if on_step is not None:
            on_step(i, state, predicted)
    return n_pred


def run_recorded_episode(agent, episode, on_step: Optional[Callable[[int, Dict], None]] = None) -> int:
    """One pass of the reference's inner loop (nav/collect.py:44-59: ``hab_env.reset(); nav_agent.reset();
    while not episode_over: action = nav_agent.act(observations)``) over a recorded episode
    (peanut_amd/episodes.py: the (rgb, depth, gps, compass, objectgoal) tuples Habitat produced).  ``agent`` is a
    ``peanut_amd.peanut_agent.PEANUT_Agent``; ``episode`` a path or a loaded dict.  Returns the number of
    predictions."""
    from . import episodes as E
    ep = E.load_episode(episode) if isinstance(episode, str) else episode
    agent.reset()
    n_pred = 0
    for i, observations in enumerate(E.iter_observations(ep)):
        out = agent.act(observations)
        n_pred += int(out.get('predicted', False))
        if on_step is not None:
            on_step(i, out)
    return n_pred


def run_recorded_shard(agent, episode_paths: List[str], start_ep: int = 0, end_ep: int = -1,
                       on_episode: Optional[Callable[[int, int], None]] = None) -> Dict[int, int]:
    """The outer loop of nav/collect.py:40-84 with its ``--start_ep/--end_ep`` window (:37-39,50): every episode
    before ``end_ep`` is iterated (the reference resets the env for skipped ones too), only those in
    [start_ep, end_ep) are run.  Returns {episode index: predictions}."""
    num_episodes = len(episode_paths)
    end = end_ep if end_ep > 0 else num_episodes
    done = {}
    ep_i = 0
    while ep_i < min(num_episodes, end):
        if start_ep <= ep_i < end:
            done[ep_i] = run_recorded_episode(agent, episode_paths[ep_i])
            if on_episode is not None:
                on_episode(ep_i, done[ep_i])
        ep_i += 1
    return done


def episode_shard(n_episodes: int) -> List[int]:
    """Episode ids owned by this rank (contiguous ``[start_ep, end_ep)`` like nav/collect.py:50)."""
    rank, _, world = pdist.env_rank_world()
    s, e = pdist.shard_range(n_episodes, rank, world)
    return list(range(s, e))
