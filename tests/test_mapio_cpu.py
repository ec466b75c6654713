"""The .npz semantic-map sequence format of the reference (collect_maps.py / LoadMapFromFile)."""
import numpy as np
import torch

from peanut_amd import mapio


def test_roundtrip_and_scaling(tmp_path):
    rng = np.random.RandomState(0)
    maps = [rng.uniform(0, 1, size=(14, 96, 96)).astype(np.float32) for _ in range(3)]
    maps[1][1] = 1.0
    p = str(tmp_path / "f00001.npz")
    mapio.save_map_sequence(p, [torch.from_numpy(m) for m in maps])
    seq = mapio.load_map_sequence(p)
    assert seq.dtype == np.uint8 and seq.shape == (3, 14, 96, 96)
    assert np.array_equal(seq[0], (maps[0] * 255).astype(np.uint8))          # truncation, not rounding
    x = mapio.model_input(seq, 1)
    assert x.shape == (1, 14, 96, 96) and x.dtype == torch.float32
    assert float(x[0, 1].min()) == 1.0 and float(x.max()) <= 1.0
    assert mapio.keep_sequence(seq)
    assert not mapio.keep_sequence(np.zeros_like(seq))
    tgt = mapio.target_from_sequence(seq, 1)
    assert tgt.shape == (96, 96, 6) and float(tgt.sum()) == 0.0             # t=1 is fully explored
    assert mapio.SAVE_STEPS[0] == 25 and mapio.SAVE_STEPS[-1] == 500 and len(mapio.SAVE_STEPS) == 20


def test_npy_form(tmp_path):
    seq = np.zeros((2, 14, 8, 8), np.uint8)
    p = str(tmp_path / "m.npy")
    np.save(p, seq)
    assert mapio.load_map_sequence(p).shape == (2, 14, 8, 8)
