"""ORACLE support (test infrastructure): golden vectors for the map-projection path, produced by
the reference's own ``Semantic_Mapping`` (imported unmodified from /root/reference/nav) on seeded
synthetic sequences; also asserts that oracle/mapping_ref.py is bit-identical to it.
Run via ``python -m oracle.gen_golden`` (build container only)."""
from __future__ import annotations

import os
from argparse import Namespace

import numpy as np
import torch

from oracle import mapping_ref, mapping_scenes, ref_import

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLDEN = os.path.join(ROOT, "tests", "golden")

SEQS = [("seq0", 0, 8), ("seq1", 1, 6)]
# 22 semantic categories (> 16): the other set of full-height categories (mapping.py:107-113), 26-channel maps
SEQS_C22 = [("c22_seq0", 2, 5)]


def reference_module(cfg: mapping_ref.MapCfg):
    SM = ref_import.load_reference_mapping()
    args = Namespace(device=torch.device("cpu"), frame_height=cfg.frame_height, frame_width=cfg.frame_width,
                     map_resolution=cfg.map_resolution, map_size_cm=cfg.map_size_cm,
                     global_downscaling=cfg.global_downscaling, vision_range=cfg.vision_range, hfov=cfg.hfov,
                     du_scale=cfg.du_scale, cat_pred_threshold=cfg.cat_pred_threshold,
                     exp_pred_threshold=cfg.exp_pred_threshold, map_pred_threshold=cfg.map_pred_threshold,
                     num_sem_categories=cfg.num_sem_categories, camera_height=cfg.camera_height)
    return SM(args).eval()


# non-default flags of mapping.py:15-37 (round 3): (name, MapCfg overrides, scene seed, frames)
FLAG_VARIANTS = [
    ("du2", dict(du_scale=2), 4, 5),                                       # depth sub-sampled, semantics average-pooled 2x2
    ("vr64_map2400", dict(vision_range=64, map_size_cm=2400), 5, 5),       # smaller egocentric window, 240 x 240 local map
    ("f96x128", dict(frame_height=96, frame_width=128, hfov=90.0, camera_height=1.25), 6, 6),   # another camera
    ("res4_thresholds", dict(map_resolution=4, map_size_cm=3840, cat_pred_threshold=2.0, exp_pred_threshold=2.0,
                             map_pred_threshold=0.5, global_downscaling=1, vision_range=80), 7, 5),   # cell size, thresholds, no downscaling
]


def generate(report):
    _generate(report, mapping_ref.MapCfg(), SEQS, "mapping_golden.npz")
    _generate(report, mapping_ref.MapCfg(num_sem_categories=22), SEQS_C22, "mapping_golden_c22.npz")
    out = {}
    for name, over, seed, n in FLAG_VARIANTS:
        cfg = mapping_ref.MapCfg(**over)
        part = _generate(report, cfg, [(name, seed, n)], None)
        for k, v in over.items():
            part[f"{name}/cfg_{k}"] = np.array(v)
        out.update(part)
    np.savez_compressed(os.path.join(GOLDEN, "mapping_golden_flags.npz"), **out)


def _generate(report, cfg, seqs, fname):
    sm = reference_module(cfg)
    out = {}
    torch.set_grad_enabled(False)
    for name, seed, n in seqs:
        frames = mapping_scenes.make_sequence(seed, n, h=cfg.frame_height, w=cfg.frame_width, ncat=cfg.num_sem_categories,
                                              hfov=cfg.hfov, cam_h_cm=cfg.camera_height * 100.0)
        M, C = cfg.map_cells, 4 + cfg.num_sem_categories
        maps_ref = torch.zeros(C, M, M)
        maps_mine = torch.zeros(C, M, M)
        # agent starts at the centre of the local map, heading 0 (agent_state.py init_map_and_pose)
        pose_ref = torch.tensor([cfg.local_size_cm / 100.0 / 2.0, cfg.local_size_cm / 100.0 / 2.0, 0.0])
        pose_mine = pose_ref.clone()
        sums, nnz, poses, fps, stairs = [], [], [], [], []
        worst = 0.0
        for fr in frames:
            obs = torch.from_numpy(mapping_scenes.frame_to_obs(fr, ncat=cfg.num_sem_categories))[None]
            rel = torch.from_numpy(fr["pose"])
            fp_r, maps_ref, _, pose_ref = sm(obs, rel, maps_ref, pose_ref, None)
            fp_m, maps_mine, _, pose_mine = mapping_ref.forward(obs, rel, maps_mine, pose_mine, cfg)
            assert torch.equal(fp_r, fp_m), name
            assert torch.equal(maps_ref, maps_mine), f"{name}: restatement deviates from the reference"
            assert torch.equal(pose_ref, pose_mine), name
            worst = max(worst, float((maps_ref - maps_mine).abs().max()))
            sums.append(maps_ref.double().sum((1, 2)).numpy())
            nnz.append((maps_ref != 0).sum((1, 2)).numpy())
            poses.append(pose_ref.numpy().copy())
            fps.append(np.packbits(fp_r.numpy().astype(bool)))
            coords = mapping_ref.point_cloud_std(obs[:, 3], cfg)
            feat = torch.ones(1, 1 + cfg.num_sem_categories, (obs.shape[2] // cfg.du_scale) * (obs.shape[3] // cfg.du_scale))
            feat[:, 1:] = torch.nn.AvgPool2d(cfg.du_scale)(obs[:, 4:]).reshape(1, cfg.num_sem_categories, -1)
            stairs.append(bool(mapping_ref.stairs_mask(coords, feat).any()))
        final = maps_ref.numpy()
        idx = np.flatnonzero(final)
        out[f"{name}/depth"] = np.stack([f["depth"] for f in frames])
        out[f"{name}/sem"] = np.stack([f["sem"] for f in frames])
        out[f"{name}/pose_obs"] = np.stack([f["pose"] for f in frames])
        out[f"{name}/channel_sums"] = np.stack(sums)
        out[f"{name}/channel_nnz"] = np.stack(nnz)
        out[f"{name}/poses"] = np.stack(poses)
        out[f"{name}/fp_map_bits"] = np.stack(fps)
        out[f"{name}/final_idx"] = idx.astype(np.int32)
        out[f"{name}/final_val"] = final.reshape(-1)[idx].astype(np.float32)
        out[f"{name}/stairs_branch"] = np.array(stairs)
        report["mapping"][name] = dict(frames=n, restatement_max_abs=worst, final_nnz=int(idx.size),
                                       stairs_frames=[i for i, s in enumerate(stairs) if s])
        print(f"[mapping] {name}: {n} frames, restatement bit-identical, final nnz {idx.size}, "
              f"stairs branch taken in frames {[i for i, s in enumerate(stairs) if s]}")
    if fname:
        np.savez_compressed(os.path.join(GOLDEN, fname), **out)
    return out
