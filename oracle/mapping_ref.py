"""ORACLE (test infrastructure, not product code) -- CPU restatement of the reference's
egocentric -> allocentric semantic-map projection, ``Semantic_Mapping.forward``
(nav/agent/mapping.py:52-179 with nav/agent/utils/depth_utils.py and nav/agent/utils/model.py),
in plain fp32 PyTorch.  Only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s
``cpu_baseline`` leg may import it.

Pinned by: ``oracle/gen_golden_mapping.py`` imports the reference's ``Semantic_Mapping`` UNMODIFIED
(it needs only torch/numpy/matplotlib) in the build container, drives it with seeded synthetic
frame sequences, asserts this restatement is bit-identical, and commits the results under
``tests/golden/mapping_golden.npz``.  The reference has no test of its own for this path.

The float32 operation ORDER of the reference is kept on purpose (e.g. ``((gx - xc) * d) / f``,
``/ res`` then ``- 50`` then ``/ 100`` then ``* 2``): the HIP kernels follow the same order.
"""
from __future__ import annotations

import itertools
from dataclasses import dataclass
from typing import Tuple

import numpy as np
import torch
import torch.nn.functional as F


@dataclass(frozen=True)
class MapCfg:
    """The ``args`` fields ``Semantic_Mapping.__init__`` reads (mapping.py:15-37) at the defaults
    of nav/arguments.py."""
    frame_height: int = 120
    frame_width: int = 160
    map_resolution: int = 5
    map_size_cm: int = 4800
    global_downscaling: int = 2
    vision_range: int = 100
    hfov: float = 79.0
    du_scale: int = 1
    cat_pred_threshold: float = 5.0
    exp_pred_threshold: float = 1.0
    map_pred_threshold: float = 0.1
    num_sem_categories: int = 10
    camera_height: float = 0.88

    # derived (mapping.py:20-35, 102-103)
    @property
    def local_size_cm(self) -> int:
        return self.map_size_cm // self.global_downscaling

    @property
    def map_cells(self) -> int:
        return self.local_size_cm // self.map_resolution

    @property
    def max_h(self) -> int:
        return int(360 / self.map_resolution)

    @property
    def min_h(self) -> int:
        return int(-40 / self.map_resolution)

    @property
    def z_bins(self) -> int:
        return self.max_h - self.min_h

    @property
    def agent_height(self) -> float:
        return self.camera_height * 100.0

    @property
    def min_z(self) -> int:
        return int(25 / self.map_resolution - self.min_h)

    @property
    def max_z(self) -> int:
        return int((self.agent_height + 1) / self.map_resolution - self.min_h)

    @property
    def shift_x(self) -> int:
        return self.vision_range * self.map_resolution // 2

    @property
    def all_height_cats(self) -> Tuple[int, ...]:
        """Feature rows that use the all-height projection (mapping.py:107-113)."""
        return (1 + 5, 1 + 2) if self.num_sem_categories <= 16 else (1 + 3, 1 + 9, 1 + 14)


def camera_matrix(cfg: MapCfg):
    """depth_utils.py:27-34."""
    xc = (cfg.frame_width - 1.0) / 2.0
    zc = (cfg.frame_height - 1.0) / 2.0
    f = (cfg.frame_width / 2.0) / np.tan(np.deg2rad(cfg.hfov / 2.0))
    return xc, zc, f


def point_cloud_std(depth: torch.Tensor, cfg: MapCfg) -> torch.Tensor:
    """depth [1,h,w] (cm) -> normalised coordinates [1,3,h*w] in grid units of [-1,1]
    (depth_utils.py:129-195 with elevation 0 and shift_loc=[250,0,pi/2] -> both rotations are the
    identity; mapping.py:59-88)."""
    xc, zc, f = camera_matrix(cfg)
    h, w = depth.shape[-2:]
    gx = torch.arange(w)[None, None, :].expand(1, h, w)                    # column index
    gz = torch.arange(h - 1, -1, -1)[None, :, None].expand(1, h, w)        # row index, flipped
    s = cfg.du_scale
    d = depth[:, ::s, ::s]
    X = (gx[:, ::s, ::s] - xc) * d / f
    Z = (gz[:, ::s, ::s] - zc) * d / f
    xyz = torch.stack((X, d, Z), dim=3)
    eye = torch.from_numpy(np.eye(3)).float()
    xyz = torch.matmul(xyz.reshape(-1, 3), eye.transpose(1, 0)).reshape(xyz.shape)   # R(elevation 0)
    xyz[..., 2] = xyz[..., 2] + cfg.agent_height
    xyz = torch.matmul(xyz.reshape(-1, 3), eye.transpose(1, 0)).reshape(xyz.shape)   # R(pi/2 - pi/2)
    xyz[..., 0] += cfg.shift_x
    xyz[..., 1] += 0
    xyz = xyz.float()
    vr, res = cfg.vision_range, cfg.map_resolution
    xyz[..., :2] = xyz[..., :2] / res
    xyz[..., :2] = (xyz[..., :2] - vr // 2.) / vr * 2.
    xyz[..., 2] = xyz[..., 2] / res
    xyz[..., 2] = (xyz[..., 2] - (cfg.max_h + cfg.min_h) // 2.) / (cfg.max_h - cfg.min_h) * 2.
    xyz = xyz.permute(0, 3, 1, 2)
    return xyz.reshape(xyz.shape[0], xyz.shape[1], xyz.shape[2] * xyz.shape[3])


def stairs_mask(coords: torch.Tensor, feat: torch.Tensor) -> torch.Tensor:
    """mapping.py:90-97: when the 3 % z-quantile of the in-range points is above 0.2 and more than
    20 % of them lie in (0.2, 0.7), points below 0.7 that are not 'toilet' are thrown out of range."""
    z = coords[0, 2, :]
    my = z[(z > -1) & (z < 1)] * 2 + 1.6
    if len(my) > 0 and torch.quantile(my, 0.03) > 0.2 and torch.sum((my > 0.2) & (my < 0.7)) > 0.2 * len(my):
        below = z * 2 + 1.6 < 0.7
        no_toilet = feat[0, 1 + 4] == 0
        return below & no_toilet
    return torch.zeros_like(z, dtype=torch.bool)


def splat(feat: torch.Tensor, coords: torch.Tensor, dims: Tuple[int, int, int]) -> torch.Tensor:
    """``splat_feat_nd`` (depth_utils.py:198-252): trilinear splat where the WHOLE grid is rounded
    (half-to-even) after each of the 8 corner passes; a corner index is 'safe' iff 0 < ix < dim
    (strict), unsafe corners get weight 0 and index component 0."""
    B, Fch, _ = feat.shape
    n = dims[0] * dims[1] * dims[2]
    grid = torch.zeros(B, Fch, n, dtype=torch.float32)
    pos_dim, wts_dim = [], []
    for d in range(3):
        pos = coords[:, [d], :] * dims[d] / 2 + dims[d] / 2
        pd, wd = [], []
        for ix in (0, 1):
            p_ix = torch.floor(pos) + ix
            safe = ((p_ix > 0) & (p_ix < dims[d])).type(pos.dtype)
            w_ix = (1 - torch.abs(pos - p_ix)) * safe
            pd.append(p_ix * safe)
            wd.append(w_ix)
        pos_dim.append(pd)
        wts_dim.append(wd)
    for corner in itertools.product((0, 1), repeat=3):
        wts = torch.ones_like(wts_dim[0][0])
        index = torch.zeros_like(wts_dim[0][0])
        for d in range(3):
            index = index * dims[d] + pos_dim[d][corner[d]]
            wts = wts * wts_dim[d][corner[d]]
        grid.scatter_add_(2, index.long().expand(-1, Fch, -1), feat * wts)
        grid = torch.round(grid)
    return grid.view(B, Fch, *dims)


def integrate_pose(pose: torch.Tensor, rel: torch.Tensor) -> torch.Tensor:
    """``get_new_pose_batch`` (mapping.py:143-158); pose [1,3] = (x m, y m, theta deg) is updated IN
    PLACE (the reference's returned ``pose_pred`` aliases it), rel = (dx, dy, dtheta rad)."""
    k = 57.29577951308232
    pose[:, 1] += rel[:, 0] * torch.sin(pose[:, 2] / k) + rel[:, 1] * torch.cos(pose[:, 2] / k)
    pose[:, 0] += rel[:, 0] * torch.cos(pose[:, 2] / k) - rel[:, 1] * torch.sin(pose[:, 2] / k)
    pose[:, 2] += rel[:, 2] * k
    pose[:, 2] = torch.fmod(pose[:, 2] - 180.0, 360.0) + 180.0
    pose[:, 2] = torch.fmod(pose[:, 2] + 180.0, 360.0) - 180.0
    return pose


def affine_grids(st_pose: torch.Tensor, size) -> Tuple[torch.Tensor, torch.Tensor]:
    """``get_grid`` (model.py:7-43): rotation grid then translation grid from F.affine_grid with its
    DEFAULT align_corners=False (the sampling below uses align_corners=True -- the reference's
    inconsistent pairing is reproduced, not fixed)."""
    x, y, t = st_pose[:, 0], st_pose[:, 1], st_pose[:, 2]
    t = t * np.pi / 180.
    c, s = t.cos(), t.sin()
    z, o = torch.zeros_like(c), torch.ones_like(c)
    theta1 = torch.stack([torch.stack([c, -s, z], 1), torch.stack([s, c, z], 1)], 1)
    theta2 = torch.stack([torch.stack([o, -z, x], 1), torch.stack([z, o, y], 1)], 1)
    return F.affine_grid(theta1, torch.Size(size)), F.affine_grid(theta2, torch.Size(size))


def forward(obs: torch.Tensor, pose_obs: torch.Tensor, maps_last: torch.Tensor, poses_last: torch.Tensor,
            cfg: MapCfg = MapCfg()):
    """``Semantic_Mapping.forward`` (mapping.py:52-179).
    obs [1,4+ncat,h,w] (ch 3 = depth in cm, ch 4.. = semantic), pose_obs [3] (dx, dy, dtheta),
    maps_last [4+ncat,M,M], poses_last [3] (mutated in place like the reference).
    Returns (fp_map_pred [1,vr,vr], map_pred [4+ncat,M,M], pose_pred [3], current_pose [3])."""
    pose_obs, maps_last, poses_last_b = pose_obs[None, :], maps_last[None, :], poses_last[None, :]
    bs, c, h, w = obs.shape
    ncat = cfg.num_sem_categories
    coords = point_cloud_std(obs[:, 3, :, :], cfg)
    feat = torch.ones(1, 1 + ncat, (h // cfg.du_scale) * (w // cfg.du_scale))
    feat[:, 1:, :] = torch.nn.AvgPool2d(cfg.du_scale)(obs[:, 4:, :, :]).reshape(bs, c - 4, -1)     # mapping.py:80-82
    mask = stairs_mask(coords, feat)
    coords[:, :, mask] = 99999
    vr, zb = cfg.vision_range, cfg.z_bins
    voxels = splat(feat, coords, (vr, vr, zb)).transpose(2, 3)
    agent_h = voxels[..., cfg.min_z:cfg.max_z].sum(4)
    all_h = voxels.sum(4)
    for f in cfg.all_height_cats:
        agent_h[:, f] = all_h[:, f]
    fp_map = torch.clamp(agent_h[:, 0:1] / cfg.map_pred_threshold, min=0.0, max=1.0)
    fp_exp = torch.clamp(all_h[:, 0:1] / cfg.exp_pred_threshold, min=0.0, max=1.0)
    M = cfg.map_cells
    agent_view = torch.zeros(bs, c, M, M)
    x1 = cfg.local_size_cm // (cfg.map_resolution * 2) - vr // 2
    y1 = cfg.local_size_cm // (cfg.map_resolution * 2)
    agent_view[:, 0:1, y1:y1 + vr, x1:x1 + vr] = fp_map
    agent_view[:, 1:2, y1:y1 + vr, x1:x1 + vr] = fp_exp
    agent_view[:, 4:, y1:y1 + vr, x1:x1 + vr] = torch.clamp(agent_h[:, 1:] / cfg.cat_pred_threshold, min=0.0, max=1.0)
    current = integrate_pose(poses_last_b, pose_obs)
    st = current.clone().detach()
    half = cfg.local_size_cm // (cfg.map_resolution * 2)
    st[:, :2] = -(st[:, :2] * 100.0 / cfg.map_resolution - half) / half
    st[:, 2] = 90. - st[:, 2]
    rot, trans = affine_grids(st, agent_view.size())
    rotated = F.grid_sample(agent_view, rot, align_corners=True)
    translated = F.grid_sample(rotated, trans, align_corners=True)
    map_pred, _ = torch.max(torch.cat((maps_last.unsqueeze(1), translated.unsqueeze(1)), 1), 1)
    return fp_map[0], map_pred[0], poses_last_b[0], current[0]
