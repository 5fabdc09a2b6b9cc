"""Seeded synthetic inputs for the tracking hot path.

The reference ships no data, weights or fixtures, so every test,
smoke run and benchmark uses inputs generated here from ``numpy.random.default_rng``.
This module only GENERATES data (numpy / torch-CPU); it runs no part of the hot path.

LM-only scenes (BASELINE config 1 style): N 3-D points uniform in the object box, a
SIMPLE_RADIAL camera f = 1.2 max(w, h), a ground-truth pose looking at the box from a
distance that gives ~50 % image fill, band-limited random feature fields (sigma = 4 px
at each level's resolution) with C = 32/128/128 channels plus a smooth confidence, and
reference observations sampled from the same raw fields at the GT projection.
"""
from __future__ import annotations

import math
from dataclasses import dataclass, field
from typing import Dict, List, Optional, Sequence, Tuple

import numpy as np
import torch
from scipy.ndimage import gaussian_filter

from .geometry import Camera, Pose

# premier_protein render box in ngp coordinates (reference config/premier_protein.sh:14)
PREMIER_PROTEIN_AABB = [[0.359, -0.248, 0.047], [0.627, 0.223, 0.574]]
CRACKER_BOX_AABB = [[0.0, 0.0, 0.0], [1.0, 1.0, 1.0]]  # config/cracker_box.sh:3
NGP_SCALE, NGP_OFFSET = 0.33, 0.5  # instant-ngp nerf->ngp convention


def ngp_aabb_to_nerf_box(aabb) -> Tuple[np.ndarray, np.ndarray]:
    """Box corners in NeRF ("transforms.json") coordinates: invert
    p_ngp = cycle(p_nerf * 0.33 + 0.5) with cycle (x,y,z) <- (y,z,x)."""
    lo, hi = np.asarray(aabb[0], float), np.asarray(aabb[1], float)
    lo_n, hi_n = (lo - NGP_OFFSET) / NGP_SCALE, (hi - NGP_OFFSET) / NGP_SCALE
    # ngp = (nerf_y, nerf_z, nerf_x)  ->  nerf = (ngp_z, ngp_x, ngp_y)
    perm = [2, 0, 1]
    return lo_n[perm], hi_n[perm]


def look_at_pose(eye: np.ndarray, target: np.ndarray, up=np.array([0.0, 0.0, 1.0])):
    """World->camera (R, t), camera looking along +z (COLMAP/pixloc convention), y down."""
    fwd = target - eye
    fwd = fwd / np.linalg.norm(fwd)
    right = np.cross(fwd, up)
    if np.linalg.norm(right) < 1e-6:
        right = np.cross(fwd, np.array([0.0, 1.0, 0.0]))
    right /= np.linalg.norm(right)
    down = np.cross(fwd, right)
    R = np.stack([right, down, fwd], 0)  # rows = camera axes in world coords
    t = -R @ eye
    return R, t


def rodrigues(w: np.ndarray) -> np.ndarray:
    th = np.linalg.norm(w)
    if th < 1e-12:
        return np.eye(3)
    k = w / th
    K = np.array([[0, -k[2], k[1]], [k[2], 0, -k[0]], [-k[1], k[0], 0]])
    return np.eye(3) + math.sin(th) * K + (1 - math.cos(th)) * (K @ K)


def perturb_pose(R, t, rng, rot_deg: float, trans: float, center: Optional[np.ndarray] = None):
    """Rotate the camera about the object centre by ``rot_deg`` around a random axis and shift
    it by ``trans``: keeps the object in view."""
    axis = rng.normal(size=3)
    axis /= np.linalg.norm(axis)
    dR = rodrigues(axis * math.radians(rot_deg))
    d = rng.normal(size=3)
    d = d / np.linalg.norm(d) * trans
    if center is None:
        center = np.zeros(3)
    # x_cam = R (dR (x - c) + c) + t + d
    Rn = R @ dR
    tn = R @ (center - dR @ center) + t + d
    return Rn, tn


def smooth_field(rng, C: int, h: int, w: int, sigma: float) -> np.ndarray:
    """Band-limited Gaussian random field, unit variance per channel, [C, h, w] float32."""
    f = rng.standard_normal((C, h, w)).astype(np.float32)
    f = gaussian_filter(f, sigma=(0, sigma, sigma), mode="wrap")
    f /= f.std(axis=(1, 2), keepdims=True) + 1e-12
    return f


def bilinear_chw(fmap: np.ndarray, pts: np.ndarray) -> np.ndarray:
    """Plain numpy bilinear sample (pixel centres at integers); pts inside the map."""
    C, h, w = fmap.shape
    x, y = pts[:, 0], pts[:, 1]
    x0 = np.clip(np.floor(x).astype(int), 0, w - 2)
    y0 = np.clip(np.floor(y).astype(int), 0, h - 2)
    ax, ay = (x - x0)[None], (y - y0)[None]
    out = (
        fmap[:, y0, x0] * (1 - ax) * (1 - ay)
        + fmap[:, y0, x0 + 1] * ax * (1 - ay)
        + fmap[:, y0 + 1, x0] * (1 - ax) * ay
        + fmap[:, y0 + 1, x0 + 1] * ax * ay
    )
    return out.T.astype(np.float32)


@dataclass
class LMScene:
    width: int
    height: int
    p3d: np.ndarray  # [N,3] float64 (SfM/world units)
    camera: Camera  # full-resolution query camera (pixel-centre convention)
    R_gt: np.ndarray
    t_gt: np.ndarray
    R_init: np.ndarray
    t_init: np.ndarray
    scales: List[Tuple[float, float]]  # per level (1/stride, 1/stride)
    feats_query: List[torch.Tensor]  # per level [(C+1), h, w] raw descriptor + confidence
    feats_ref: List[torch.Tensor]  # per level [N, C+1] raw observation + confidence
    dims: Tuple[int, ...] = (32, 128, 128)
    strides: Tuple[int, ...] = (1, 4, 16)
    center: np.ndarray = field(default_factory=lambda: np.zeros(3))

    @property
    def T_gt(self) -> Pose:
        return Pose.from_Rt(self.R_gt, self.t_gt)

    @property
    def T_init(self) -> Pose:
        return Pose.from_Rt(self.R_init, self.t_init)


def make_lm_scene(
    seed: int = 1001,
    width: int = 320,
    height: int = 240,
    n_points: int = 2048,
    aabb=PREMIER_PROTEIN_AABB,
    dims: Sequence[int] = (32, 128, 128),
    strides: Sequence[int] = (1, 4, 16),
    init_rot_deg: float = 3.0,
    init_trans: float = 0.02,
    sigma_px: float = 4.0,
    fill: float = 0.5,
    k1: float = 0.0,
) -> LMScene:
    rng = np.random.default_rng(seed)
    lo, hi = ngp_aabb_to_nerf_box(aabb)
    p3d = rng.uniform(lo, hi, size=(n_points, 3))
    center = 0.5 * (lo + hi)
    extent = float(np.max(hi - lo))
    f = 1.2 * max(width, height)
    dist = f * extent / (fill * min(width, height))
    direction = rng.normal(size=3)
    direction[2] = abs(direction[2]) * 0.3
    direction /= np.linalg.norm(direction)
    eye = center + direction * dist
    R_gt, t_gt = look_at_pose(eye, center)
    R_init, t_init = perturb_pose(R_gt, t_gt, rng, init_rot_deg, init_trans, center)
    colmap_cam = dict(model="SIMPLE_RADIAL", width=width, height=height,
                      params=np.array([f, width / 2.0, height / 2.0, k1]))
    camera = Camera.from_colmap(colmap_cam)

    feats_query, feats_ref, scales = [], [], []
    p3d_cam = p3d @ R_gt.T + t_gt
    for C_, s in zip(dims, strides):
        h, w = height // s, width // s
        raw = smooth_field(rng, C_, h, w, sigma_px)
        amp = np.exp(0.3 * smooth_field(rng, 1, h, w, 2 * sigma_px))
        raw = raw * amp
        conf = 1.0 / (1.0 + np.exp(-smooth_field(rng, 1, h, w, 2 * sigma_px)))
        fq = np.concatenate([raw, conf], 0).astype(np.float32)
        sc = (1.0 / s, 1.0 / s)
        cam_l = camera.scale(sc)
        p2d, _ = cam_l.world2image(torch.from_numpy(p3d_cam).float())
        p2d = p2d.numpy().astype(np.float64)
        p2d[:, 0] = np.clip(p2d[:, 0], 0, w - 1.001)
        p2d[:, 1] = np.clip(p2d[:, 1], 0, h - 1.001)
        obs = bilinear_chw(fq, p2d)
        feats_query.append(torch.from_numpy(fq))
        feats_ref.append(torch.from_numpy(obs))
        scales.append(sc)
    return LMScene(width, height, p3d, camera, R_gt, t_gt, R_init, t_init, scales, feats_query,
                   feats_ref, tuple(dims), tuple(strides), center)
