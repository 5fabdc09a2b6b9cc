"""CPU ORACLE (test infrastructure, NOT product code) for the instant-ngp style NeRF
inference renderer on the pixtrack hot path.

PARITY UNPINNED: the renderer is NVlabs/instant-ngp (`.gitmodules:7-11`, branch master,
`checkout = b551bf1`) + tiny-cuda-nn, reached through `pyngp`; none of that source is
under /root/reference.  The only reference-side facts are the call
sites -- pixtrack/visualization/run_vis_on_poses.py:28-57 (fov from fx, camera matrix,
render(w,h,spp=8,linear=True), Depth/Shade), pixtrack/utils/ingp_utils.py:22-44 (render
settings) -- and the model shape printed in `notebooks/Render YCB GT Poses .ipynb:147-150`
(hash grid L=16,F=2,T=2^19,N_min=16,b=1.51572 -> 13,074,912 params, which
`grid_level_layout` below reproduces exactly; MLPs 32->64->16 and 32->64->64->16).
Everything else restates the published instant-ngp inference algorithm (SURVEY.md
Appendix B); where recollection was uncertain the choice made here IS the
specification the HIP kernel is tested against:

* rays through pixel centres, principal point at the image centre, fov from fx on x;
* radial k1 lens handled by 8 fixed-point un-distortion iterations;
* per-(pixel, spp) start jitter u in [0,1) from an integer hash (`start_jitter`);
* dt = clamp(t * cone_angle, sqrt(3)/1024, sqrt(3)/1024 * 2^(cascades-1) * 8)   [calc_dt: matches upstream's
  MIN_/MAX_CONE_STEPSIZE up to the cascade count - builder decision: the model's, not the compile-time 8; no effect
  inside a render box, DESIGN.md section 4];
* cascade of a sample = min(max_cascade, max(mip_from_pos, frexp-exponent of dt * 2 * 128))   [mip_from_dt: matches
  upstream since round 5; rounds 1-4 used dt * 128];
* mip_from_pos = clamp(frexp-exponent of max|pos - 0.5| + 1, 0, max_cascade)   [matches];
* advance_to_next_voxel: t += calc_dt(t) repeated until t passes the DDA exit of the current cascade cell
  (distance_to_next_voxel with the floor(p + 0.5 + 0.5 sign(d)) form)   [matches the b551bf1-era do / while loop; later
  upstream versions step analytically in "stepping space" - builder decision: the loop, the checkout is from 2022];
* occupancy bitfield: one bit per 128^3 cell, x fastest, cascades concatenated;
* empty cells are skipped by stepping t in dt increments to the next voxel border;
* hash grid: tcnn layout (dense below 2^19 entries, else the 3-prime xor hash),
  trilinear, fp16 table, fp16 output; SH degree 4 of the view direction (fp16)   [the trilinear sum here is separate
  fp32 multiplies and adds; the HIP kernel's may contract them (one rounding less per corner, round 5) - the fp16
  rounding of the two features that follows hides it for all but a few packed values; the MARCH arithmetic stays
  operation for operation this file's, which is what makes ray and sample counts EQUAL in the tests];
* MLPs: fp16 weights and activations, fp32 accumulation, ReLU hidden, density =
  exp(out[0]), rgb = sigmoid(out[0..2]);
* compositing front to back, stop when transmittance < min_transmittance with the
  accumulated colour renormalised by 1/alpha (instant-ngp's early-out);
* Shade mode: every finished ray's composited (premultiplied) colour goes through
  `srgb_to_linear` before it is accumulated over spp, unless the snapshot was trained in
  linear colours (instant-ngp's shade_kernel_nerf: `if (!train_in_linear_colors && mode ==
  Shade) rgb = srgb_to_linear(rgb)`; pixtrack's PNG datasets are not HDR, so the conversion
  is ON for them) -- `NgpModel.linear_colors`, default False = convert.  Depth mode: none;
* output = premultiplied linear RGBA averaged over spp, composited over
  background.rgb * background.a (transparent for pixtrack's [255,255,255,0]).

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this.
"""
from __future__ import annotations

import math
import os
from dataclasses import dataclass, field
from typing import Dict, List, Tuple

import numpy as np

F32 = np.float32
GRID = 128
MIN_STEP = F32(math.sqrt(3.0) / 1024.0)


@dataclass
class NgpModel:
    n_levels: int = 16
    n_features: int = 2
    log2_hashmap: int = 19
    base_res: int = 16
    per_level_scale: float = 1.51572
    cascades: int = 3
    aabb_scale: float = 4.0
    cone_angle: float = 1.0 / 256.0
    depth_scale: float = 1.0
    linear_colors: bool = False  # True: the network's colours are linear already (HDR training set): no srgb_to_linear
    grid: np.ndarray = None  # float16 [n_entries_total, n_features]
    mlp: Dict[str, np.ndarray] = None  # float16: d1 [64,32], d2 [16,64], c1 [64,32], c2 [64,64], c3 [16,64]
    occupancy: np.ndarray = None  # uint8 bitfield [cascades * 128^3 / 8]


def grid_level_layout(m: NgpModel) -> List[Tuple[float, int, int, int, bool]]:
    """Per level: (scale, resolution, offset, size, hashed) -- tiny-cuda-nn GridEncoding."""
    out, off = [], 0
    T = 1 << m.log2_hashmap
    for l in range(m.n_levels):
        scale = 2.0 ** (l * math.log2(m.per_level_scale)) * m.base_res - 1.0
        res = int(math.ceil(scale)) + 1
        n = res**3
        n = T if n > T else n
        n = (n + 7) // 8 * 8
        n = min(n, T)
        out.append((scale, res, off, n, res**3 > n))
        off += n
    return out


# Switches for scripts/renderer_decisions.py ONLY (the measured effect of every recall-level decision of
# this file).  The defaults ARE the specification; no test or product path changes them.
VARIANT = {"jitter": "hash", "undistort_iters": 8, "max_step_cascades": None, "mip_dt_factor": 2 * GRID}


def max_step(m: NgpModel) -> np.float32:
    c = m.cascades if VARIANT["max_step_cascades"] is None else int(VARIANT["max_step_cascades"])
    return F32(MIN_STEP * F32(2 ** (c - 1)) * F32(1024 // GRID))


def start_jitter(pixel_index: np.ndarray, spp_index: int) -> np.ndarray:
    """u in [0,1): 24-bit integer hash of (pixel, sample), exact in float32."""
    if VARIANT["jitter"] == "none":  # (decision table: every pass starts at the box entry)
        return np.zeros(pixel_index.shape, np.float32)
    if VARIANT["jitter"] == "other":  # (decision table: an unrelated sequence of the same quality)
        pixel_index = pixel_index * 2654435761 + 12345
    h = (pixel_index.astype(np.uint64) * 747796405 + np.uint64(spp_index) * 2891336453 + 1) & 0xFFFFFFFF
    h ^= h >> 16
    h = (h * 0x7FEB352D) & 0xFFFFFFFF
    h ^= h >> 15
    h = (h * 0x846CA68B) & 0xFFFFFFFF
    h ^= h >> 16
    return ((h >> 8).astype(np.float32) * F32(1.0 / 16777216.0)).astype(np.float32)


def nerf_matrix_to_ngp(nerf_c2w: np.ndarray, scale: float = 0.33, offset: float = 0.5) -> np.ndarray:
    """instant-ngp nerf_matrix_to_ngp: flip y,z columns, scale+offset the origin, cycle
    the world axes (x,y,z) <- (y,z,x).  Input/outputs 3x4."""
    m = np.array(nerf_c2w[:3, :4], dtype=np.float64).copy()
    m[:, 1] *= -1
    m[:, 2] *= -1
    m[:, 3] = m[:, 3] * scale + offset
    return m[[1, 2, 0], :]


def srgb_to_linear(c: np.ndarray) -> np.ndarray:
    """instant-ngp common_device.cuh srgb_to_linear, element-wise, float32."""
    c = np.asarray(c, np.float32)
    hi = np.power((np.maximum(c, F32(0.04045)) + F32(0.055)) / F32(1.055), F32(2.4)).astype(np.float32)
    return np.where(c <= F32(0.04045), c / F32(12.92), hi).astype(np.float32)


def calc_dt(t, cone_angle, lo, hi):
    return np.minimum(np.maximum(t * F32(cone_angle), lo), hi).astype(np.float32)


def mip_from_pos(pos: np.ndarray, cascades: int) -> np.ndarray:
    maxval = np.max(np.abs(pos - F32(0.5)), axis=-1)
    _, e = np.frexp(maxval)
    return np.clip(e + 1, 0, cascades - 1).astype(np.int32)


def mip_from_dt(dt: np.ndarray, pos: np.ndarray, cascades: int) -> np.ndarray:
    """instant-ngp nerf_device.cuh mip_from_dt: `dt *= 2 * NERF_GRIDSIZE(); if (dt < 1) return mip_from_pos;
    frexpf(dt, &e); return min(max_cascade, max(e, mip))` - the cascade whose cell a step of dt spans HALF of.  (The
    `dt < 1` guard is what max(e, mip) does anyway: e <= 0 there and mip >= 0.)  Rounds 1-4 had the factor as
    NERF_GRIDSIZE() - one cascade finer wherever t >= 1: VARIANT["mip_dt_factor"] = 128 renders that."""
    _, e = np.frexp(dt * F32(VARIANT["mip_dt_factor"]))
    return np.minimum(cascades - 1, np.maximum(e, mip_from_pos(pos, cascades))).astype(np.int32)


def occupied(m: NgpModel, pos: np.ndarray, mip: np.ndarray) -> np.ndarray:
    """Bit test of the cascade-`mip` cell containing pos (float32 [n,3])."""
    scale = np.ldexp(F32(1.0), -mip).astype(np.float32)[:, None]  # 2^-mip
    p = (pos - F32(0.5)) * scale + F32(0.5)
    idx = np.floor(p * F32(GRID)).astype(np.int64)
    inside = np.all((idx >= 0) & (idx < GRID), axis=-1)
    idx = np.clip(idx, 0, GRID - 1)
    lin = (idx[:, 2] * GRID + idx[:, 1]) * GRID + idx[:, 0] + mip.astype(np.int64) * GRID**3
    bits = (m.occupancy[lin >> 3] >> (lin & 7).astype(np.uint8)) & 1
    return inside & (bits != 0)


def advance_to_next_voxel(t, pos, d, idir, mip, cone_angle, lo, hi):
    """Step t in dt increments until it passes the border of the current cascade cell."""
    res = np.ldexp(F32(GRID), -mip).astype(np.float32)[:, None]  # cells per unit at this mip
    p = (res * (pos - F32(0.5))).astype(np.float32)
    with np.errstate(invalid="ignore"):
        tx = (np.floor(p + F32(0.5) + F32(0.5) * np.sign(d).astype(np.float32)) - p) * idir
    tx = np.where(d != 0, tx, np.float32(np.inf))  # axes the ray does not move along
    tmin = np.min(tx, axis=-1)
    t_target = (t + np.maximum(tmin / res[:, 0], F32(0.0))).astype(np.float32)
    t = t.copy()
    todo = np.ones_like(t, dtype=bool)
    while todo.any():
        t[todo] = (t[todo] + calc_dt(t[todo], cone_angle, lo, hi)).astype(np.float32)
        todo &= t < t_target
    return t


# ---- encodings -------------------------------------------------------------

PRIMES = (np.uint32(1), np.uint32(2654435761), np.uint32(805459861))


def hash_grid_encode(m: NgpModel, x: np.ndarray) -> np.ndarray:
    """x float32 [n,3] in [0,1]; returns float16 [n, n_levels*n_features]."""
    n = x.shape[0]
    out = np.zeros((n, m.n_levels * m.n_features), np.float16)
    for l, (scale, res, off, size, hashed) in enumerate(grid_level_layout(m)):
        pos = (x * F32(scale)).astype(np.float32) + F32(0.5)
        pg = np.floor(pos)
        frac = (pos - pg).astype(np.float32)
        pg = pg.astype(np.int64).astype(np.uint32)
        acc = np.zeros((n, m.n_features), np.float32)
        for corner in range(8):
            w = np.ones(n, np.float32)
            cg = np.zeros((n, 3), np.uint32)
            for dim in range(3):
                if corner & (1 << dim):
                    w = w * frac[:, dim]
                    cg[:, dim] = pg[:, dim] + np.uint32(1)
                else:
                    w = w * (F32(1.0) - frac[:, dim])
                    cg[:, dim] = pg[:, dim]
            if hashed:
                idx = (cg[:, 0] * PRIMES[0]) ^ (cg[:, 1] * PRIMES[1]) ^ (cg[:, 2] * PRIMES[2])
            else:
                idx = cg[:, 0] + cg[:, 1] * np.uint32(res) + cg[:, 2] * np.uint32(res * res)
            idx = (idx % np.uint32(size)).astype(np.int64) + off
            acc += w[:, None] * m.grid[idx].astype(np.float32)
        out[:, l * m.n_features:(l + 1) * m.n_features] = acc.astype(np.float16)
    return out


def sh4(d: np.ndarray) -> np.ndarray:
    """Real spherical harmonics up to degree 4 (16 coefficients), tiny-cuda-nn ordering."""
    x, y, z = d[:, 0], d[:, 1], d[:, 2]
    xy, xz, yz, x2, y2, z2 = x * y, x * z, y * z, x * x, y * y, z * z
    o = np.empty((d.shape[0], 16), np.float32)
    o[:, 0] = 0.28209479177387814
    o[:, 1] = -0.48860251190291987 * y
    o[:, 2] = 0.48860251190291987 * z
    o[:, 3] = -0.48860251190291987 * x
    o[:, 4] = 1.0925484305920792 * xy
    o[:, 5] = -1.0925484305920792 * yz
    o[:, 6] = 0.94617469575755997 * z2 - 0.31539156525251999
    o[:, 7] = -1.0925484305920792 * xz
    o[:, 8] = 0.54627421529603959 * x2 - 0.54627421529603959 * y2
    o[:, 9] = 0.59004358992664352 * y * (-3.0 * x2 + y2)
    o[:, 10] = 2.8906114426405538 * xy * z
    o[:, 11] = 0.45704579946446572 * y * (1.0 - 5.0 * z2)
    o[:, 12] = 0.3731763325901154 * z * (5.0 * z2 - 3.0)
    o[:, 13] = 0.45704579946446572 * x * (1.0 - 5.0 * z2)
    o[:, 14] = 1.4453057213202769 * z * (x2 - y2)
    o[:, 15] = 0.59004358992664352 * x * (-x2 + 3.0 * y2)
    return o.astype(np.float16)


_MIN_GEMM_ROWS = 128


def _layer(w16: np.ndarray, x16: np.ndarray, relu: bool) -> np.ndarray:
    """fp16 operands, fp32 accumulation.  Batches below 128 rows are zero-padded to 128: this image's OpenBLAS sends
    small products (M <= 18 .. 75 for these shapes) through a different kernel with another summation order, which
    would make a sample's value depend on how many other rays happen to be alive with it - and a render tiled over
    processes (render_parallel) differ from the serial one in the last bits."""
    n = x16.shape[0]
    x32 = x16.astype(np.float32)
    if n < _MIN_GEMM_ROWS:
        x32 = np.concatenate([x32, np.zeros((_MIN_GEMM_ROWS - n, x32.shape[1]), np.float32)], 0)
    y = (x32 @ w16.astype(np.float32).T)[:n]
    if relu:
        y = np.maximum(y, 0)
    return y


def network(m: NgpModel, pos_unit: np.ndarray, dirs: np.ndarray):
    """pos_unit: warped position in [0,1]^3; dirs: unit view directions.
    Returns (density [n], rgb [n,3]) float32."""
    feat = hash_grid_encode(m, pos_unit)
    h = _layer(m.mlp["d1"], feat, True).astype(np.float16)
    dout = _layer(m.mlp["d2"], h, False)
    density = np.exp(dout[:, 0]).astype(np.float32)
    cin = np.concatenate([dout.astype(np.float16), sh4(dirs)], 1)
    h = _layer(m.mlp["c1"], cin, True).astype(np.float16)
    h = _layer(m.mlp["c2"], h, True).astype(np.float16)
    cout = _layer(m.mlp["c3"], h, False)
    rgb = (1.0 / (1.0 + np.exp(-cout[:, :3]))).astype(np.float32)
    return density, rgb


# ---- renderer --------------------------------------------------------------


@dataclass
class View:
    cam: np.ndarray  # 3x4 camera-to-world in ngp coordinates
    focal: float
    width: int
    height: int
    spp: int = 8
    k1: float = 0.0
    aabb_min: Tuple[float, float, float] = (0.0, 0.0, 0.0)
    aabb_max: Tuple[float, float, float] = (1.0, 1.0, 1.0)
    background: Tuple[float, float, float, float] = (255.0, 255.0, 255.0, 0.0)
    min_transmittance: float = 1e-7
    mode: int = 0  # 0 Shade, 1 Depth


def generate_rays(v: View, rows=None):
    """Rays of the given image rows (an index array; default: all), row-major; every ray's arithmetic is per pixel."""
    W, H = v.width, v.height
    ry = np.arange(H, dtype=np.float32) if rows is None else np.asarray(rows).astype(np.float32)
    px, py = np.meshgrid(np.arange(W, dtype=np.float32), ry)
    u = ((px + F32(0.5)) / F32(W)).astype(np.float32)
    w_ = ((py + F32(0.5)) / F32(H)).astype(np.float32)
    dx = ((u - F32(0.5)) * F32(W) / F32(v.focal)).astype(np.float32).ravel()
    dy = ((w_ - F32(0.5)) * F32(H) / F32(v.focal)).astype(np.float32).ravel()
    if v.k1 != 0.0:
        xu, yu = dx.copy(), dy.copy()
        for _ in range(int(VARIANT["undistort_iters"])):
            r2 = (xu * xu + yu * yu).astype(np.float32)
            s = (F32(1.0) + F32(v.k1) * r2).astype(np.float32)
            xu, yu = (dx / s).astype(np.float32), (dy / s).astype(np.float32)
        dx, dy = xu, yu
    cam = np.asarray(v.cam, dtype=np.float32)
    d = (dx[:, None] * cam[None, :, 0] + dy[:, None] * cam[None, :, 1]).astype(np.float32) + cam[None, :, 2]
    d = d.astype(np.float32)
    nrm = np.sqrt((d[:, 0] * d[:, 0] + d[:, 1] * d[:, 1]).astype(np.float32) + d[:, 2] * d[:, 2]).astype(np.float32)
    d = (d / nrm[:, None]).astype(np.float32)
    o = np.broadcast_to(cam[:, 3], d.shape).astype(np.float32)
    c2 = cam[:, 2]
    fn = np.sqrt((c2[0] * c2[0] + c2[1] * c2[1]).astype(np.float32) + c2[2] * c2[2]).astype(np.float32)
    fwd = (c2 / fn).astype(np.float32)
    return o, d, fwd


def ray_aabb(o, d, lo, hi):
    idir = (F32(1.0) / d).astype(np.float32)
    t0 = ((lo - o) * idir).astype(np.float32)
    t1 = ((hi - o) * idir).astype(np.float32)
    tmin = np.max(np.minimum(t0, t1), axis=-1)
    tmax = np.min(np.maximum(t0, t1), axis=-1)
    return tmin, tmax, idir


def _row_list(v: View, rows) -> np.ndarray:
    if rows is None:
        return np.arange(v.height, dtype=np.int64)
    if isinstance(rows, tuple) and len(rows) == 2:
        return np.arange(int(rows[0]), int(rows[1]), dtype=np.int64)
    return np.asarray(rows, np.int64).reshape(-1)


def render(m: NgpModel, v: View, return_stats: bool = False, rows=None):
    """Returns float32 [H, W, 4] linear premultiplied RGBA.  ``rows``: only these image rows - (r0, r1) for
    r0 <= y < r1, or a list of row indices - as [len(rows), W, 4]; every ray is computed exactly as in the full render
    (rays are independent and keep their global pixel index for the start jitter): the unit render_parallel deals to
    worker processes.  The spp passes of a pixel are marched TOGETHER as independent rays of one batch (an eighth of the
    numpy calls of a pass-by-pass loop) and summed in pass order at the end: the arithmetic of every ray and the
    order of the spp mean are those of a sequential loop over the passes.
    With DEFAULT_PROCS > 1 (``$PXT_ORACLE_PROCS``; the fixture generators and bench.py's cpu_baseline set it) a
    full render is dealt to that many processes - same image, bit for bit."""
    if rows is None and DEFAULT_PROCS > 1 and not _PAR.get("worker"):
        return render_parallel(m, v, DEFAULT_PROCS, return_stats)
    rl = _row_list(v, rows)
    o1, d1, fwd = generate_rays(v, rl)
    n1 = o1.shape[0]
    half = F32(m.aabb_scale / 2.0)
    scene_lo, scene_hi = F32(0.5) - half, F32(0.5) + half
    lo = np.maximum(np.asarray(v.aabb_min, np.float32), scene_lo)
    hi = np.minimum(np.asarray(v.aabb_max, np.float32), scene_hi)
    tmin1, tmax1, idir1 = ray_aabb(o1, d1, lo, hi)
    hit1 = tmax1 > np.maximum(tmin1, F32(0.0))
    dt_lo, dt_hi = MIN_STEP, max_step(m)
    pix1 = (rl[:, None] * v.width + np.arange(v.width, dtype=np.int64)[None, :]).reshape(-1)
    zdot1 = ((d1[:, 0] * fwd[0] + d1[:, 1] * fwd[1]).astype(np.float32) + d1[:, 2] * fwd[2]).astype(np.float32)
    inv_s = F32(1.0 / m.aabb_scale)
    S = int(v.spp)
    # ray r = s * n1 + i: pass s of pixel i
    o, d, idir = np.tile(o1, (S, 1)), np.tile(d1, (S, 1)), np.tile(idir1, (S, 1))
    tmax, zdot = np.tile(tmax1, S), np.tile(zdot1, S)
    n = n1 * S
    t0 = (np.maximum(tmin1, F32(0.0)) + F32(1e-6)).astype(np.float32)
    dt0 = calc_dt(t0, m.cone_angle, dt_lo, dt_hi)
    t = np.concatenate([(t0 + start_jitter(pix1, s) * dt0).astype(np.float32) for s in range(S)])
    alive = np.tile(hit1, S)
    T = np.ones(n, np.float32)
    rgba = np.zeros((n, 4), np.float32)
    n_samples = 0
    while alive.any():
        idx = np.nonzero(alive)[0]
        # -- find the next occupied sample of every live ray
        ti = t[idx]
        searching = np.ones(idx.shape[0], bool)
        found = np.zeros(idx.shape[0], bool)
        while searching.any():
            k = np.nonzero(searching)[0]
            tk = ti[k]
            pos = (o[idx[k]] + tk[:, None] * d[idx[k]]).astype(np.float32)
            out_of_box = tk >= tmax[idx[k]]
            dtk = calc_dt(tk, m.cone_angle, dt_lo, dt_hi)
            mip = mip_from_dt(dtk, pos, m.cascades)
            occ = occupied(m, pos, mip) & ~out_of_box
            found[k[occ]] = True
            searching[k[occ | out_of_box]] = False
            adv = ~(occ | out_of_box)
            if adv.any():
                ka = k[adv]
                ti[ka] = advance_to_next_voxel(tk[adv], pos[adv], d[idx[ka]], idir[idx[ka]], mip[adv],
                                               m.cone_angle, dt_lo, dt_hi)
        t[idx] = ti
        alive[idx[~found]] = False
        idx = idx[found]
        if idx.size == 0:
            break
        # -- evaluate + composite one sample per live ray
        ti = t[idx]
        pos = (o[idx] + ti[:, None] * d[idx]).astype(np.float32)
        dti = calc_dt(ti, m.cone_angle, dt_lo, dt_hi)
        unit = ((pos - scene_lo) * inv_s).astype(np.float32)
        density, rgb = network(m, unit, d[idx])
        n_samples += idx.size
        if v.mode == 1:
            depth = (ti * zdot[idx] * F32(m.depth_scale)).astype(np.float32)
            rgb = np.repeat(depth[:, None], 3, 1)
        alpha = (F32(1.0) - np.exp(-density * dti)).astype(np.float32)
        wgt = (alpha * T[idx]).astype(np.float32)
        rgba[idx, :3] += wgt[:, None] * rgb
        rgba[idx, 3] += wgt
        T[idx] = (T[idx] * (F32(1.0) - alpha)).astype(np.float32)
        done = T[idx] < F32(v.min_transmittance)
        if done.any():
            di = idx[done]
            rgba[di] = rgba[di] / rgba[di, 3:4]
            alive[di] = False
        t[idx] = (ti + dti).astype(np.float32)
    if v.mode == 0 and not m.linear_colors:
        rgba[:, :3] = srgb_to_linear(rgba[:, :3])
    out = np.zeros((n1, 4), np.float32)
    for s in range(S):  # the sequential mean over the passes: ((r0 + r1) + r2) + ...
        out += rgba[s * n1:(s + 1) * n1]
    out /= F32(v.spp)
    bg = np.asarray(v.background, np.float32)
    a = out[:, 3:4]
    out[:, :3] += bg[:3] * bg[3] * (1 - a)
    out[:, 3:4] = a + bg[3] * (1 - a)
    img = out.reshape(rl.shape[0], v.width, 4)
    if return_stats:
        return img, {"samples": int(n_samples), "rays_hit": int(hit1.sum()) * v.spp}
    return img


# ---- the same render on several host cores ------------------------------------------------------
_PAR = {}
_POOL = {}
DEFAULT_PROCS = int(os.environ.get("PXT_ORACLE_PROCS", "1") or 1)


def _worker_init():
    """One BLAS thread per worker process (the workers ARE the parallelism)."""
    _PAR["worker"] = True
    try:
        import threadpoolctl

        _PAR["_limit"] = threadpoolctl.threadpool_limits(1)
    except Exception:  # pragma: no cover - threadpoolctl is optional
        pass


def _render_rows_job(job):
    v, rows = job
    return rows, render(_PAR["m"], v, return_stats=True, rows=rows)


def close_pool():
    """Ends the worker processes render_parallel keeps between calls."""
    pool = _POOL.pop("pool", None)
    degraded = _POOL.get("degraded")
    _POOL.clear()
    if degraded:
        _POOL["degraded"] = True
    _PAR.pop("m", None)
    if pool is not None:
        pool.terminate()
        pool.join()


import atexit

atexit.register(close_pool)  # (a pool left to the interpreter's shutdown dies noisily in Pool.__del__)


def render_parallel(m: NgpModel, v: View, procs: int, return_stats: bool = False, rows_per_job: int = 0):
    """render() with the image rows dealt to ``procs`` forked worker processes.  The workers are forked once per
    (model, procs) - they see the model copy-on-write - and kept for later calls (a frame needs two renders; forking
    from a process with a GPU context mapped costs tens of ms per worker); close_pool() ends them.  Rays are
    independent, so the image and the counts are those of the serial render - bit for bit (the small-batch GEMM
    padding in _layer is what makes a sample's value independent of the batch it sits in;
    tests/test_oracle_kats.py asserts the equality here, bench.py's cpu_baseline on the box it runs on)."""
    import multiprocessing as mp

    procs = max(1, int(procs))
    if procs == 1:
        return render(m, v, return_stats)
    # (the workers see the model and this module's switches as they were at the fork: all of it is in the key)
    key = (id(m), procs, bool(m.linear_colors), m.cascades, float(m.cone_angle), tuple(sorted((k, str(x)) for k, x in VARIANT.items())))
    if _POOL.get("key") != key:
        close_pool()
        _PAR["m"] = m
        _POOL["pool"] = mp.get_context("fork").Pool(procs, initializer=_worker_init)
        _POOL["key"] = key
        _POOL["model"] = m  # (keeps id(m) from being recycled while the pool lives)
    # one job per worker, rows dealt round-robin (rows k, k + P, k + 2P, ...): the object's rows - far more expensive than
    # rows of background - spread evenly, and every job is one large batch (the march loop's cost per step is mostly
    # Python / numpy call overhead).  ``rows_per_job`` > 0: contiguous bands of that many rows instead (tests).
    if rows_per_job:
        sets = [np.arange(r, min(r + rows_per_job, v.height)) for r in range(0, v.height, rows_per_job)]
    else:
        sets = [np.arange(k, v.height, procs) for k in range(min(procs, v.height))]
    # The workers are forked from a process that may hold a GPU context and OpenMP / BLAS thread pools (bench.py's
    # cpu_baseline runs after the GPU loop): a child can inherit a held lock and never answer.  The wait is
    # bounded; after a time-out the pool is dropped and this and every later render of the process run serially, so a
    # baseline figure always comes out (it then says `render_processes: 1`).
    if _POOL.get("degraded"):
        return render(m, v, return_stats)
    try:
        parts = _POOL["pool"].map_async(_render_rows_job, [(v, rs) for rs in sets], chunksize=1).get(
            timeout=float(os.environ.get("PXT_ORACLE_POOL_TIMEOUT", "900")))
    except mp.TimeoutError:
        close_pool()
        _POOL["degraded"] = True
        return render(m, v, return_stats)
    img = np.empty((v.height, v.width, 4), np.float32)
    stats = {"samples": 0, "rays_hit": 0}
    for rs, (part, st) in parts:
        img[rs] = part
        stats["samples"] += st["samples"]
        stats["rays_hit"] += st["rays_hit"]
    return (img, stats) if return_stats else img
