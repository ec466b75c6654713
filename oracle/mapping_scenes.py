"""ORACLE support (test infrastructure): seeded synthetic observation sequences for the
map-projection path -- analytic floor / wall / box depth images in the units
``_preprocess_depth`` produces (nav/agent/agent_helper.py:197-217: cm = 50 + d*450, invalid or
too-far pixels = 45050), plus rectangular semantic masks.  Shapes follow nav/arguments.py
(120x160 frames, 10 semantic channels)."""
from __future__ import annotations

import math
from typing import Dict, List

import numpy as np

FAR_CM = 50.0 + 100.0 * 450.0      # what invalid / >0.99 depth turns into


def _floor_depth(h, w, hfov, cam_h_cm, floor_h_cm):
    xc, zc = (w - 1.0) / 2.0, (h - 1.0) / 2.0
    f = (w / 2.0) / math.tan(math.radians(hfov / 2.0))
    gz = np.arange(h - 1, -1, -1, dtype=np.float64)[:, None].repeat(w, 1)   # flipped row index
    below = zc - gz                                                          # > 0 below the horizon
    with np.errstate(divide="ignore", invalid="ignore"):
        d = np.where(below > 0, (cam_h_cm - floor_h_cm) * f / below, np.inf)
    return d


def make_sequence(seed: int = 0, n_frames: int = 8, h: int = 120, w: int = 160, ncat: int = 10,
                  hfov: float = 79.0, cam_h_cm: float = 88.0) -> List[Dict[str, np.ndarray]]:
    """Frames: dict(depth f32 [h,w] in cm (quantised to 1/4 cm so fixtures compress), sem u8
    [ncat,h,w], pose f32 [3] = (dx m, dy m, dtheta rad)).  Frame 3 looks at a raised floor (30 cm,
    triggers the low-stairs branch mapping.py:94), frame 5 has a far/invalid band (points outside
    the grid) and overlapping instance masks (value 2)."""
    rng = np.random.RandomState(seed)
    frames = []
    for i in range(n_frames):
        floor_h = 30.0 if i == 3 else 0.0
        d = _floor_depth(h, w, hfov, cam_h_cm, floor_h)
        wall = rng.uniform(180.0, 420.0)
        # a slanted wall: depth varies linearly across columns
        slope = rng.uniform(-0.6, 0.6)
        wall_d = wall + slope * (np.arange(w)[None, :] - w / 2.0)
        d = np.minimum(d, wall_d)
        # two boxes closer than the wall
        for _ in range(2):
            r0, c0 = rng.randint(20, h - 50), rng.randint(5, w - 45)
            hh, ww = rng.randint(15, 45), rng.randint(15, 40)
            bd = rng.uniform(70.0, wall * 0.8)
            d[r0:r0 + hh, c0:c0 + ww] = np.minimum(d[r0:r0 + hh, c0:c0 + ww], bd)
        d = d + rng.uniform(-1.0, 1.0, size=d.shape)              # sensor noise
        d = np.clip(d, 50.0, 495.5)
        if i == 5:
            d[:, 120:] = FAR_CM                                    # invalid band -> outside the grid
        d = np.round(d * 4.0) / 4.0
        sem = np.zeros((ncat, h, w), np.uint8)
        for _ in range(3):
            k = rng.randint(0, ncat - 1)
            r0, c0 = rng.randint(10, h - 40), rng.randint(5, w - 40)
            sem[k, r0:r0 + rng.randint(10, 35), c0:c0 + rng.randint(10, 35)] += 1
        if i in (3, 5):
            sem[4, 60:100, 40:90] += 1                             # 'toilet' row (feat[0,5]) region
        if i == 5:
            sem[1, 30:60, 30:70] += 1
            sem[1, 40:70, 50:90] += 1                              # overlapping instances -> value 2
        pose = np.array([rng.uniform(0.0, 0.3), rng.uniform(-0.05, 0.05),
                         rng.choice([0.0, math.radians(30.0), -math.radians(30.0), rng.uniform(-0.2, 0.2)])],
                        np.float32)
        frames.append(dict(depth=d.astype(np.float32), sem=sem, pose=pose))
    return frames


def frame_to_obs(frame, ncat: int = 10) -> np.ndarray:
    """[4+ncat,h,w] float32 observation: RGB (unused by the mapping) zero, ch 3 depth, ch 4.. sem."""
    h, w = frame["depth"].shape
    obs = np.zeros((4 + ncat, h, w), np.float32)
    obs[3] = frame["depth"]
    obs[4:] = frame["sem"].astype(np.float32)
    return obs
