"""Recorded-episode format of the replay harness: the observation tuple Habitat's
ObjectNav task hands to ``PEANUT_Agent.act`` (nav/agent/peanut_agent.py:38-68), one array per sensor with the
time axis first, in a single ``.npz`` per episode:

    rgb        uint8   [T,H,W,3]   observations['rgb']
    depth      float32 [T,H,W,1]   observations['depth']   (simulator units in [0,1], 0 = invalid)
    gps        float32 [T,2]       observations['gps']
    compass    float32 [T,1]       observations['compass']
    objectgoal int64   [T,1]       observations['objectgoal']  (HM3D goal id, hm3d_to_coco maps it)

Optional, for replays without a detector: ``inst_offsets`` int64 [T+1], ``inst_masks`` uint8 [n,H,W],
``inst_classes`` int32 [n], ``inst_scores`` float32 [n] -- the instances ``DefaultPredictor`` produced on each
frame (segmentation.py:45), stored back to back."""
from __future__ import annotations

from typing import Dict, Iterator, List, Optional

import numpy as np

KEYS = ("rgb", "depth", "gps", "compass", "objectgoal")


def save_episode(path: str, frames: List[Dict]) -> None:
    """frames: observation dicts as Habitat yields them (``KEYS``), optionally with ``instances`` =
    (masks [n,H,W], classes [n], scores [n])."""
    out = {
        "rgb": np.stack([np.asarray(f["rgb"], np.uint8) for f in frames]),
        "depth": np.stack([np.asarray(f["depth"], np.float32).reshape(f["rgb"].shape[0], f["rgb"].shape[1], 1) for f in frames]),
        "gps": np.stack([np.asarray(f["gps"], np.float32).reshape(2) for f in frames]),
        "compass": np.stack([np.asarray(f["compass"], np.float32).reshape(1) for f in frames]),
        "objectgoal": np.stack([np.asarray(f["objectgoal"], np.int64).reshape(1) for f in frames]),
    }
    if all("instances" in f for f in frames):
        off, ms, cs, ss = [0], [], [], []
        for f in frames:
            m, c, s = (np.asarray(a) for a in f["instances"])
            off.append(off[-1] + len(c))
            ms.append(m.astype(np.uint8).reshape(len(c), *out["rgb"].shape[1:3]))
            cs.append(c.astype(np.int32))
            ss.append(s.astype(np.float32))
        out.update(inst_offsets=np.asarray(off, np.int64), inst_masks=np.concatenate(ms), inst_classes=np.concatenate(cs),
                   inst_scores=np.concatenate(ss))
    np.savez_compressed(path, **out)


def load_episode(path: str) -> Dict[str, np.ndarray]:
    z = np.load(path)
    ep = {k: z[k] for k in z.files}
    missing = [k for k in KEYS if k not in ep]
    if missing:
        raise ValueError(f"{path}: not a recorded episode, missing {missing}")
    T = ep["rgb"].shape[0]
    if any(ep[k].shape[0] != T for k in KEYS):
        raise ValueError(f"{path}: sensors disagree on the number of frames")
    return ep


def iter_observations(ep: Dict[str, np.ndarray], start: int = 0, stop: Optional[int] = None) -> Iterator[Dict]:
    """Yield per-step observation dicts in Habitat's form (what ``act`` receives)."""
    T = ep["rgb"].shape[0]
    for t in range(start, T if stop is None else min(stop, T)):
        obs = {k: ep[k][t] for k in KEYS}
        if "inst_offsets" in ep:
            a, b = int(ep["inst_offsets"][t]), int(ep["inst_offsets"][t + 1])
            obs["instances"] = (ep["inst_masks"][a:b], ep["inst_classes"][a:b], ep["inst_scores"][a:b])
        yield obs
