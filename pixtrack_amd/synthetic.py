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
from typing import List, Optional, Sequence, Tuple

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


# ---------------------------------------------------------------------------
# Synthetic NeRF ("seeded synthetic hash-grid NeRF + a procedural density
# prior so the object is opaque inside the AABB").  Pure data generation.
# ---------------------------------------------------------------------------


def ngp_grid_layout(n_levels=16, log2_hashmap=19, base_res=16, per_level_scale=1.51572):
    """(scale, res, offset, size, hashed) per level -- tiny-cuda-nn GridEncoding sizes
    (reproduces the 13,074,912 parameters of `notebooks/Render YCB GT Poses .ipynb:147`)."""
    out, off, T = [], 0, 1 << log2_hashmap
    for l in range(n_levels):
        scale = 2.0 ** (l * math.log2(per_level_scale)) * base_res - 1.0
        res = int(math.ceil(scale)) + 1
        n = res**3
        n = T if n > T else n
        n = min((n + 7) // 8 * 8, T)
        out.append((scale, res, off, n, res**3 > n))
        off += n
    return out, off


def shape_value(x_ngp: np.ndarray, aabb) -> np.ndarray:
    """Procedural object: a super-ellipsoid filling ~90 % of the render box; > 0 inside."""
    lo, hi = np.asarray(aabb[0], np.float64), np.asarray(aabb[1], np.float64)
    c, r = 0.5 * (lo + hi), 0.45 * (hi - lo)
    q = np.abs((x_ngp - c) / r)
    return 1.0 - np.sum(q**4, axis=-1) ** 0.25


SHAPE_LEVEL = 3  # finest dense level (56^3 vertices): carries the shape indicator
SHAPE_SHARPNESS = 3.0
DENSITY_LOGIT = 10.0  # density = exp(+-10) inside / outside
C3_GAIN = 1.0


def scene_scale_for_box(aabb) -> float:
    """aabb_scale of the synthetic snapshot for a render box.  The shape indicator lives on the finest DENSE hash
    level (56^3 vertices across the scene cube), i.e. one vertex per 4 / 55 = 0.073 units at the default scale 4:
    a box thinner than 2 vertex spacings (config/roncelli_blankk.sh: 0.079 in y) would hold no vertex with a
    positive indicator and render as empty space.  Such boxes get the scene cube of scale 2 ([-0.5, 1.5]^3, every
    OBJ_AABB of config/*.sh lies inside it), which halves the spacing; all other boxes keep scale 4 (and with it
    the seeds and fixtures of rounds 1-2)."""
    lo, hi = np.asarray(aabb[0], float), np.asarray(aabb[1], float)
    thin = float(np.min(hi - lo)) < 2.0 * 4.0 / 55.0
    fits2 = bool((lo > -0.45).all() and (hi < 1.45).all())
    return 2.0 if (thin and fits2) else 4.0


def make_synthetic_nerf(seed: int = 11, aabb=PREMIER_PROTEIN_AABB, aabb_scale: float = 4.0, cascades: int = 3):
    from .ngp import NerfSnapshot

    rng = np.random.default_rng(seed)
    layout, n_entries = ngp_grid_layout()
    grid = rng.uniform(-0.5, 0.5, size=(n_entries, 2)).astype(np.float32)
    # shape indicator on the vertices of the dense level SHAPE_LEVEL, feature 0
    scale, res, off, size, hashed = layout[SHAPE_LEVEL]
    assert not hashed
    g = np.arange(res, dtype=np.float64)
    gx, gy, gz = np.meshgrid(g, g, g, indexing="ij")
    # vertex g sits at warped x_w = (g - 0.5) / scale; ngp coords = x_w * aabb_scale + scene_lo
    scene_lo = 0.5 - aabb_scale / 2
    P = np.stack([gx, gy, gz], -1)
    xw = (P - 0.5) / scale
    x_ngp = xw * aabb_scale + scene_lo
    val = np.clip(SHAPE_SHARPNESS * shape_value(x_ngp, aabb), -1.0, 1.0)
    idx = (gx + gy * res + gz * res * res).astype(np.int64)
    grid[off + idx.ravel(), 0] = val.ravel()

    # MLPs (row-major [out][in]); feature index = 2 * level + f
    def he(o, i, gain=1.0):
        return rng.normal(size=(o, i)) * math.sqrt(2.0 / i) * gain

    d1 = he(64, 32, 2.0)
    d1[0, :] = 0.0
    d1[1, :] = 0.0
    d1[0, 2 * SHAPE_LEVEL] = 4.0
    d1[1, 2 * SHAPE_LEVEL] = -4.0
    d2 = he(16, 64, 2.0)
    d2[0, :] = 0.0
    d2[0, 0], d2[0, 1] = DENSITY_LOGIT / 4.0, -DENSITY_LOGIT / 4.0
    d2[1:, 0:2] = 0.0  # the geometry features do not see the density logit
    c1 = he(64, 32, 2.0)
    c1[:, 0] = 0.0  # colour does not see the density logit either
    c1[:, 16:] *= 0.15  # weak view dependence
    # last two colour layers in antisymmetric pairs: relu(a) - relu(-a) = a keeps the colour
    # logits zero-mean (there are no biases in these MLPs), so colours spread around 0.5
    w = he(32, 64, 1.0)
    w -= w.mean(axis=1, keepdims=True)
    c2 = np.concatenate([w, -w], 0)
    v = rng.normal(size=(3, 32)) * C3_GAIN / math.sqrt(32)
    c3 = np.zeros((16, 64))
    c3[:3, :32] = v
    c3[:3, 32:] = -v
    mlp = np.concatenate([m.astype(np.float16).ravel() for m in (d1, d2, c1, c2, c3)])

    # occupancy: cascade c covers [0.5 - 2^c/2, 0.5 + 2^c/2]^3 with 128^3 cells.  instant-ngp
    # marks a cell occupied when density * MIN_STEP > 0.01 (its minimum optical thickness);
    # density = exp(DENSITY_LOGIT * interp(level-3 indicator)), evaluated here at the 8 corners
    # of every cell from the same vertex values the renderer interpolates.
    vol = val.astype(np.float32)  # [res, res, res] indexed [gx, gy, gz]

    def indicator(x_ngp_pts):
        q = ((x_ngp_pts - scene_lo) / aabb_scale) * scale + 0.5
        q = np.clip(q, 0.0, res - 1.000001)
        i0 = np.floor(q).astype(np.int64)
        f = (q - i0).astype(np.float32)
        out = np.zeros(q.shape[:-1], np.float32)
        for dx in (0, 1):
            for dy in (0, 1):
                for dz in (0, 1):
                    w = ((f[..., 0] if dx else 1 - f[..., 0]) * (f[..., 1] if dy else 1 - f[..., 1])
                         * (f[..., 2] if dz else 1 - f[..., 2]))
                    out += w * vol[i0[..., 0] + dx, i0[..., 1] + dy, i0[..., 2] + dz]
        return out

    thresh = math.log(0.01 / (math.sqrt(3) / 1024)) / DENSITY_LOGIT - 0.02
    G = 128
    occ_bits = np.zeros(cascades * G**3, np.uint8)
    for c in range(cascades):
        span = 2.0**c
        e = np.arange(G + 1) / G
        ez, ey, ex = np.meshgrid(e, e, e, indexing="ij")  # corner grid indexed [z, y, x]
        corners = (np.stack([ex, ey, ez], -1) - 0.5) * span + 0.5
        ind = indicator(corners.reshape(-1, 3)).reshape(G + 1, G + 1, G + 1)
        m = ind[:-1, :-1, :-1]
        for dz in (0, 1):
            for dy in (0, 1):
                for dx in (0, 1):
                    m = np.maximum(m, ind[dz:G + dz, dy:G + dy, dx:G + dx])
        # linear index = (z * G + y) * G + x
        occ_bits[c * G**3:(c + 1) * G**3] = (m > thresh).ravel().astype(np.uint8)
    occupancy = np.packbits(occ_bits, bitorder="little")
    return NerfSnapshot(grid=grid.astype(np.float16), mlp=mlp, occupancy=occupancy, cascades=cascades,
                        aabb_scale=aabb_scale)


# ---------------------------------------------------------------------------
# Full-pipeline synthetic object (BASELINE configs 2-4): NeRF snapshot + SfM model
# (reference cameras, surface points with tracks) + nerf2sfm + network weights + a
# ground-truth camera trajectory.  The query frames themselves are NeRF renders at the GT
# poses; they are produced on the device by `render_query_frames` (setup, not timed).
# ---------------------------------------------------------------------------

# p_ngp = A p_sfm * 0.33 + 0.5 for the identity-like nerf2sfm below (see sfm_to_nerf_pose
# followed by instant-ngp's nerf_matrix_to_ngp)
_A_SFM_TO_NGP = np.array([[1.0, 0.0, 0.0], [0.0, 0.0, -1.0], [0.0, 1.0, 0.0]])


def ngp_to_sfm_points(p_ngp: np.ndarray) -> np.ndarray:
    return ((p_ngp - NGP_OFFSET) / NGP_SCALE) @ _A_SFM_TO_NGP  # A^T applied to row vectors


def sfm_to_ngp_points(p_sfm: np.ndarray) -> np.ndarray:
    return (p_sfm @ _A_SFM_TO_NGP.T) * NGP_SCALE + NGP_OFFSET


def surface_points(rng, n: int, aabb, level: float = 0.12):
    """Points (and outward normals) on the iso-surface shape_value == level, ngp coordinates."""
    lo, hi = np.asarray(aabb[0], float), np.asarray(aabb[1], float)
    c, r = 0.5 * (lo + hi), 0.45 * (hi - lo)
    u = rng.normal(size=(n, 3))
    u /= np.linalg.norm(u, axis=1, keepdims=True)
    q = u / r  # direction in normalised coordinates
    s = (1.0 - level) / (np.sum(np.abs(q) ** 4, axis=1) ** 0.25)
    p = c + u * s[:, None]
    qn = (p - c) / r
    nrm = np.sign(qn) * np.abs(qn) ** 3 / r
    nrm /= np.linalg.norm(nrm, axis=1, keepdims=True)
    return p, nrm


# SfM camera 1 of the reference's real assets (width, height, f, cx, cy); see make_tracking_assets
REF_CAMERA_YCB = (3072, 3072, 2700.0, 1536.0, 1536.0)   # scripts/create_sfm_from_obj.py:154-159; x 0.3 (pixloc_tracker_ycb.py:89)
REF_CAMERA_PHONE = (1920, 1440, 1500.0, 960.0, 720.0)   # a 4:3 phone frame as colmap2ingp.py:226 quotes them; x 0.5 (r9:145-152)
REF_CAMERA_12MP = (4032, 3024, 3150.0, 2016.0, 1512.0)  # 12-MP stills; x 0.5 = 2016 x 1512 -> resized to 1024 x 768 by the extractor
YCB_QUERY_FXY = (1066.778, 1067.487)                    # YCB-Video intrinsics (scripts/create_sfm_from_obj.py:162-163)


def make_tracking_assets(seed: int = 1002, width: int = 640, height: int = 480, n_frames: int = 200,
                         aabb=PREMIER_PROTEIN_AABB, n_points: int = 5600, n_refs: int = 16,
                         step_deg: float = 0.5, jitter_deg: float = 0.3, jitter_trans: float = 0.003,
                         unet_seed: int = 7, reference_scale: float = 0.5, ref_camera=None, query_f=None):
    """Returns the dict PixLocPoseTrackerR9(assets=...) consumes plus 'gt_poses' [(R, t)] and
    'query_camera' (COLMAP dict).  All seeded.

    ``ref_camera`` = (width, height, f, cx, cy): SfM camera 1 as the reference's own assets have it, instead of the
    default "query size / reference_scale" stand-in - REF_CAMERA_YCB (3072 x 3072, f 2700: scripts/create_sfm_from_obj.py:
    154-159; x 0.3 -> a 921 x 921 reference render), REF_CAMERA_PHONE (1920 x 1440; x 0.5 -> 960 x 720), REF_CAMERA_12MP
    (4032 x 3024; x 0.5 -> 2016 x 1512, which the extractor resizes to 1024 x 768).  ``query_f`` = focal length (fx) of
    the query camera (YCB: 1066.778); the orbit's distance follows it so that the object fills the QUERY as before."""
    from .model3d import Model3D
    from .unet import make_synthetic_unet_weights
    from .utils.colmap import ColmapCamera, ColmapImage, ColmapPoint3D, rotmat2qvec

    rng = np.random.default_rng(seed)
    snapshot = make_synthetic_nerf(seed + 10, aabb, aabb_scale=scene_scale_for_box(aabb))
    nerf2sfm = {"up": np.array([0.0, 0.0, 1.0]), "centroid": np.zeros(3), "avglen": 3.0, "totp": np.zeros(3),
                "R": np.eye(4)}
    p_ngp, n_ngp = surface_points(rng, n_points, aabb)
    p_sfm = ngp_to_sfm_points(p_ngp)
    n_sfm = n_ngp @ _A_SFM_TO_NGP  # rotation only
    lo, hi = np.asarray(aabb[0], float), np.asarray(aabb[1], float)
    center = ngp_to_sfm_points((0.5 * (lo + hi))[None])[0]
    extent = float(np.max(hi - lo)) / NGP_SCALE
    f_q = 1.2 * max(width, height) if query_f is None else float(query_f)
    dist = f_q * extent / (0.5 * min(width, height))

    # reference (mapping) cameras: one COLMAP camera whose size x reference_scale is the query size
    # ("reference camera 2x query resolution x reference_scale 0.5"; 1/0.3 for the YCB policy)
    if ref_camera is None:
        Wr, Hr = int(round(width / reference_scale)), int(round(height / reference_scale))
        fr, cxr, cyr = 1.2 * max(Wr, Hr), Wr / 2.0, Hr / 2.0
    else:
        Wr, Hr, fr, cxr, cyr = int(ref_camera[0]), int(ref_camera[1]), *(float(x) for x in ref_camera[2:5])
    cameras = {1: ColmapCamera(1, "SIMPLE_RADIAL", Wr, Hr, np.array([fr, cxr, cyr, 0.0]))}
    up_axis = np.array([0.0, 0.0, 1.0])  # sfm z is the object's long axis (ngp -y ... +y)
    images, obs = {}, {i: [] for i in range(n_points)}
    for k in range(n_refs):
        az = 2 * math.pi * k / n_refs
        el = math.radians(15.0 * math.sin(3 * az))
        d = np.array([math.cos(az) * math.cos(el), math.sin(az) * math.cos(el), math.sin(el)])
        eye = center + d * dist
        Rk, tk = look_at_pose(eye, center, up=up_axis)
        pc = p_sfm @ Rk.T + tk
        facing = np.einsum("ij,ij->i", n_sfm, eye[None] - p_sfm) > 0.15 * np.linalg.norm(eye[None] - p_sfm, axis=1)
        uv = pc[:, :2] / pc[:, 2:3] * fr + np.array([cxr, cyr])
        vis = facing & (pc[:, 2] > 0) & (uv[:, 0] > 2) & (uv[:, 0] < Wr - 2) & (uv[:, 1] > 2) & (uv[:, 1] < Hr - 2)
        ids = np.nonzero(vis)[0]
        for j, pid in enumerate(ids):
            obs[int(pid)].append((k + 1, j))
        images[k + 1] = ColmapImage(k + 1, rotmat2qvec(Rk), tk, 1, f"mapping/{k + 1:04d}.png", uv[ids], ids.astype(np.int64))
    points3D = {}
    for pid in range(n_points):
        tr = obs[pid]
        points3D[pid] = ColmapPoint3D(pid, p_sfm[pid], np.array([128, 128, 128]), 0.5,
                                      np.array([a for a, _ in tr], np.int64), np.array([b for _, b in tr], np.int64))
    model3d = Model3D(model=(cameras, images, points3D))

    weights = make_synthetic_unet_weights(unet_seed)
    wrng = np.random.default_rng(seed + 20)
    for i in range(3):  # damping constants of the three per-level optimizers (absent checkpoint)
        weights[f"optimizer.{i}.dampingnet.const"] = torch.from_numpy(wrng.uniform(-2.5, -1.5, size=6)).float()

    # ground-truth trajectory: orbit about the object's long axis from reference image 1,
    # plus a small per-frame random twist (<= ~1 deg, <= ~5e-3 units: tracking-sized motion)
    R0, t0 = images[1].qvec2rotmat(), images[1].tvec
    gt = []
    Rc, tc = perturb_pose(R0, t0, rng, 0.8, 0.004, center)
    for i in range(n_frames):
        gt.append((Rc.copy(), tc.copy()))
        dR = rodrigues(up_axis * math.radians(step_deg))
        Rn = Rc @ dR
        tn = Rc @ (center - dR @ center) + tc
        Rc, tc = perturb_pose(Rn, tn, rng, jitter_deg * rng.uniform(), jitter_trans * rng.uniform(), center)
    qcam = dict(model="SIMPLE_RADIAL", width=width, height=height,
                params=np.array([f_q, width / 2.0, height / 2.0, 0.0]))
    return dict(model3d=model3d, nerf2sfm=nerf2sfm, snapshot=snapshot, weights=weights, covis=None, aabb=aabb,
                upright_ref_img="mapping/0001.png", gt_poses=gt, query_camera=qcam, center=center,
                width=width, height=height)


def render_query_frames(assets, testbed, noise_sigma: float = 2.0, seed: int = 5, first_frame_sigma: float = 12.0,
                        cold_start_indices=(0,)):
    """Query frames = NeRF renders at the GT poses (+ Gaussian noise, sigma in 8-bit levels),
    float32 HWC 0..255 device tensors, as ImageIterator would hand them over.  Setup only.

    ``first_frame_sigma``: the cold-start frame is a worse observation than the rest.  The
    reference gates every later frame on ``cost <= 1.1 x (cost of the first frame)`` and never
    updates the pose again after one failure (pixloc_tracker_r9.py:258-268);
    on real data the cold start (unmasked, far initialisation) is the expensive frame.  Here every
    frame is a noisy render of the same NeRF, the steady-state costs scatter +-20 % around the
    first one and 39 of 40 scene seeds trip the gate within 20 frames - after which the tracker
    stops tracking and runs a cheaper path.  A noisier first frame restores the margin the
    reference's policy assumes, so every frame of a synthetic sequence exercises the full path."""
    from .utils.ingp_utils import sfm_to_nerf_pose
    from .visualization.run_vis_on_poses import get_nerf_image_device, rgba_to_u8

    cam = Camera.from_colmap(assets["query_camera"])
    g = torch.Generator(device="cpu").manual_seed(seed)
    frames = []
    for (Rg, tg) in assets["gt_poses"]:
        wIc = np.eye(4)
        wIc[:3, :3], wIc[:3, 3] = Rg, tg
        nerf_pose = sfm_to_nerf_pose(assets["nerf2sfm"], np.linalg.inv(wIc))
        u8 = rgba_to_u8(get_nerf_image_device(testbed, nerf_pose, cam), 0.0)
        img = u8.float()
        # (``cold_start_indices``: positions in the returned list that a tracker will cold-start on)
        sigma = first_frame_sigma if (len(frames) in cold_start_indices and first_frame_sigma is not None) else noise_sigma
        if sigma > 0:
            noise = torch.randn(img.shape, generator=g) * sigma
            img = (img + noise.to(img.device)).clamp_(0, 255).round_()
        frames.append(img.contiguous())
    return frames


def write_object_dir(assets, object_path, query_dir=None, frames=None, ngp_layout: bool = True,
                     pixloc_checkpoint: bool = True):
    """Writes ``assets`` in the reference's on-disk layout (SURVEY 8b "CLI to preserve"):
    ``pixtrack/{aug_nerf_sfm/aug_sfm/*.bin, pixsfm/dataset/nerf2sfm.pkl,
    instant-ngp/snapshots/weights.msgpack, pixloc_megadepth.pt}`` and, when ``frames`` (HWC
    0..255 arrays/tensors) are given, ``query_dir/%06d.png``.  ``ngp_layout`` stores the snapshot
    in instant-ngp's params_binary layout, ``pixloc_checkpoint`` stores the weights as a pixloc
    ``{"model": state_dict}`` checkpoint, so the CLI run exercises both importers."""
    import pickle
    from pathlib import Path

    import msgpack

    from .ngp import save_snapshot, to_instant_ngp
    from .unet import to_pixloc_state_dict
    from .utils.colmap import write_model_binary

    root = Path(object_path) / "pixtrack"
    sfm = root / "aug_nerf_sfm" / "aug_sfm"
    sfm.mkdir(parents=True, exist_ok=True)
    m = assets["model3d"]
    write_model_binary(str(sfm), m.cameras, m.dbs, m.points3D)
    (root / "pixsfm" / "dataset").mkdir(parents=True, exist_ok=True)
    with open(root / "pixsfm" / "dataset" / "nerf2sfm.pkl", "wb") as f:
        pickle.dump(assets["nerf2sfm"], f)
    snaps = root / "instant-ngp" / "snapshots"
    snaps.mkdir(parents=True, exist_ok=True)
    if ngp_layout:
        (snaps / "weights.msgpack").write_bytes(msgpack.packb(to_instant_ngp(assets["snapshot"]), use_bin_type=True))
    else:
        save_snapshot(str(snaps / "weights.msgpack"), assets["snapshot"])
    w = assets["weights"]
    torch.save({"model": to_pixloc_state_dict(w)} if pixloc_checkpoint else w, root / "pixloc_megadepth.pt")
    if frames is not None:
        from PIL import Image

        q = Path(query_dir)
        q.mkdir(parents=True, exist_ok=True)
        for i, fr in enumerate(frames):
            a = fr.detach().cpu().numpy() if torch.is_tensor(fr) else np.asarray(fr)
            Image.fromarray(np.clip(np.rint(a), 0, 255).astype(np.uint8)).save(q / f"{i:06d}.png")
