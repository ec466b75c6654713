"""The reference's on-disk semantic-map sequence format: ``.npz`` with key
``maps`` = uint8 ``[T, 4+ncat, W, H]`` written by nav/collect_maps.py:67-68,80-87 (``full_map * 255``
truncated to uint8, 20 snapshots at steps 25, 50, ..., 500) and read back by
``LoadMapFromFile`` (prediction/train_prediction_model.py:63-68: ``maps[t_idx] / 255.``).  Lets real
PEANUT map datasets drive the benchmark / parity runs."""
from __future__ import annotations

from typing import Sequence

import numpy as np
import torch

SAVE_STEPS = list(range(25, 525, 25))     # collect_maps.py:52


def encode_map(full_map) -> np.ndarray:
    """``(full_map.cpu().numpy() * 255).astype(np.uint8)`` (collect_maps.py:80-81; truncation)."""
    if isinstance(full_map, torch.Tensor):
        full_map = full_map.detach().cpu().numpy()
    return (full_map * 255).astype(np.uint8)


def keep_sequence(full_map_seq: np.ndarray) -> bool:
    """Dataset filter of collect_maps.py:86: some semantics seen and > 4000 explored cells."""
    return bool(np.sum(full_map_seq[:, 4:]) > 0 and np.sum(full_map_seq[:, 1]) > 4000)


def save_map_sequence(path: str, maps: Sequence) -> None:
    """np.savez_compressed(path, maps=uint8[T,C,W,H]) (collect_maps.py:87)."""
    seq = np.stack([m if isinstance(m, np.ndarray) and m.dtype == np.uint8 else encode_map(m) for m in maps])
    np.savez_compressed(path, maps=seq)


def load_map_sequence(path: str) -> np.ndarray:
    """uint8 [T,C,W,H]; accepts the .npz (key 'maps') and bare .npy forms LoadMapFromFile accepts
    (train_prediction_model.py:63-65)."""
    maps = np.load(path)
    if path[-1] == "z":
        maps = maps["maps"]
    return np.asarray(maps)


def model_input(maps: np.ndarray, t_idx: int, device=None) -> torch.Tensor:
    """float32 [1,C,W,H] in [0,1] (= ``maps[t_idx].astype(np.float32) / 255.``, :66-68) ready for
    ``PEANUT_Prediction_Model.get_prediction_batch``."""
    x = torch.from_numpy(maps[t_idx].astype(np.float32) / np.float32(255.0))[None]
    return x.to(device) if device is not None else x


def target_from_sequence(maps: np.ndarray, t_idx: int, goal_channels: Sequence[int] = range(4, 10)) -> np.ndarray:
    """Training target of the reference (train_prediction_model.py:85-89): last snapshot's goal
    channels masked to what is unexplored at ``t_idx``; uint8 [W,H,len(goal_channels)]."""
    img = maps[t_idx].transpose(1, 2, 0).astype(np.float32) / 255.
    mask = (img[:, :, 1] > 0)
    return (maps[-1, list(goal_channels)] * (1 - mask)).transpose(1, 2, 0)
