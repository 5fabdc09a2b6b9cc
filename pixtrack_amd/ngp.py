"""Host mirror of the instant-ngp ``Testbed`` surface pixtrack drives.

Reference call sites: construction + render settings pixtrack/utils/ingp_utils.py:22-44;
per-render state pixtrack/visualization/run_vis_on_poses.py:28-57 (``fov``,
``set_nerf_camera_matrix``, ``render_mode``, ``render(w, h, spp, linear)``).

The renderer itself is one HIP kernel behind ``pxt_ngp_render`` (csrc/pxt_ngp.hip).
``render()`` keeps pyngp's contract (host float32 H x W x 4); ``render_device()`` is the
fast path that leaves the frame on the GPU for the UNet (no PCIe round trip).
"""
from __future__ import annotations

import ctypes as C
import math
import os
from dataclasses import dataclass
from enum import IntEnum
from typing import Dict, Optional

import numpy as np
import torch

from . import _lib
from .ops import ops

N_MLP_PARAMS = 64 * 32 + 16 * 64 + 64 * 32 + 64 * 64 + 16 * 64
MLP_SHAPES = [("d1", 64, 32), ("d2", 16, 64), ("c1", 64, 32), ("c2", 64, 64), ("c3", 16, 64)]


class RenderMode(IntEnum):
    Shade = 0
    Depth = 1


class TestbedMode(IntEnum):
    Nerf = 0


@dataclass
class NerfSnapshot:
    """What an instant-ngp snapshot holds, as far as inference needs it."""

    grid: np.ndarray  # float16 [n_entries, 2]
    mlp: np.ndarray  # float16 [10240]: d1 | d2 | c1 | c2 | c3, each row-major [out][in]
    occupancy: np.ndarray  # uint8 bitfield [cascades * 128^3 / 8], x fastest
    n_levels: int = 16
    n_features: int = 2
    log2_hashmap: int = 19
    base_res: int = 16
    per_level_scale: float = 1.51572
    cascades: int = 3
    aabb_scale: float = 4.0
    cone_angle: float = 1.0 / 256.0
    scale: float = 0.33  # nerf -> ngp coordinate scale
    offset: float = 0.5
    k1: float = 0.0  # training lens (render_with_camera_distortion)
    # instant-ngp's shade_kernel_nerf passes a finished ray's colour through srgb_to_linear unless the snapshot was
    # trained in linear colours (`m_nerf.training.linear_colors` = the dataset's is_hdr).  pixtrack trains on PNGs,
    # so its snapshots convert: False = convert (render(..., linear=True) then returns linear light).
    linear_colors: bool = False

    def mlp_dict(self) -> Dict[str, np.ndarray]:
        out, o = {}, 0
        for name, r, c in MLP_SHAPES:
            out[name] = self.mlp[o : o + r * c].reshape(r, c)
            o += r * c
        return out


def save_snapshot(path: str, snap: NerfSnapshot) -> None:
    """msgpack container with instant-ngp-like keys (encoding / network / snapshot)."""
    import msgpack

    d = {
        "encoding": {"otype": "HashGrid", "n_levels": snap.n_levels, "n_features_per_level": snap.n_features,
                     "log2_hashmap_size": snap.log2_hashmap, "base_resolution": snap.base_res,
                     "per_level_scale": snap.per_level_scale},
        "network": {"otype": "FullyFusedMLP", "n_neurons": 64, "n_hidden_layers": 1},
        "rgb_network": {"otype": "FullyFusedMLP", "n_neurons": 64, "n_hidden_layers": 2},
        "snapshot": {
            "version": 1,
            "grid_binary": snap.grid.astype(np.float16).tobytes(),
            "mlp_binary": snap.mlp.astype(np.float16).tobytes(),
            "density_grid_binary": snap.occupancy.tobytes(),
            "density_grid_size": 128,
            "nerf": {"aabb_scale": snap.aabb_scale, "cascades": snap.cascades, "cone_angle_constant": snap.cone_angle,
                     "linear_colors": bool(snap.linear_colors),
                     "dataset": {"scale": snap.scale, "offset": [snap.offset] * 3, "k1": snap.k1}},
        },
    }
    with open(path, "wb") as f:
        f.write(msgpack.packb(d, use_bin_type=True))


def load_snapshot_file(path: str) -> NerfSnapshot:
    """Reads ``weights.msgpack``: this package's own container (save_snapshot) or an instant-ngp
    snapshot (``snapshot.params_binary``; see from_instant_ngp)."""
    import msgpack

    with open(path, "rb") as f:
        d = msgpack.unpackb(f.read(), raw=False, strict_map_key=False)
    enc, s = d["encoding"], d["snapshot"]
    if "grid_binary" not in s:
        return from_instant_ngp(d)
    nerf = s["nerf"]
    return NerfSnapshot(
        grid=np.frombuffer(s["grid_binary"], np.float16).reshape(-1, enc["n_features_per_level"]).copy(),
        mlp=np.frombuffer(s["mlp_binary"], np.float16).copy(),
        occupancy=np.frombuffer(s["density_grid_binary"], np.uint8).copy(),
        n_levels=enc["n_levels"], n_features=enc["n_features_per_level"], log2_hashmap=enc["log2_hashmap_size"],
        base_res=enc["base_resolution"], per_level_scale=enc["per_level_scale"], cascades=nerf["cascades"],
        aabb_scale=nerf["aabb_scale"], cone_angle=nerf["cone_angle_constant"], scale=nerf["dataset"]["scale"],
        offset=nerf["dataset"]["offset"][0], k1=nerf["dataset"].get("k1", 0.0),
        linear_colors=bool(nerf.get("linear_colors", False)),
    )


# ---------------------------------------------------------------------------------------------
# instant-ngp's own snapshot layout (reference ingp_utils.py:27 load_snapshot
# of `instant-ngp/snapshots/weights.msgpack`).  instant-ngp (NVlabs/instant-ngp @ b551bf1, an
# un-vendored submodule: .gitmodules:7-11) is absent from the reference tree and no real snapshot
# exists here, so this follows its published layout from recall and is UNVERIFIED against a real
# file; the round trip through to_instant_ngp below is what the tests pin.
#
#   snapshot.params_binary : fp16 [n_params] = density MLP | rgb MLP | hash grid (| dir enc: none)
#       MLP matrices row-major [out][in]: 64x32, 16x64 | 64x32, 64x64, 16x64 (tiny-cuda-nn
#       FullyFusedMLP); rgb input = density outputs (16) then SH (16); grid = levels
#       concatenated, F halves per entry (tiny-cuda-nn GridEncoding) -> 13,074,912 in total.
#   snapshot.density_grid_binary : [cascades][128^3] density per cell in Morton order (x bit 0),
#       fp16 or fp32 by byte count; occupancy bit = density > min(0.01, mean over cascade 0),
#       then each coarser cascade ORs in the 2x2x2 max-pool of the finer one over its centre
#       half (instant-ngp update_density_grid_mean_and_bitfield / bitfield_max_pool).
#   snapshot.nerf.dataset.{scale, offset, aabb_scale, is_hdr, metadata[0].camera_distortion}
#   colour space: instant-ngp sets training.linear_colors = dataset.is_hdr when it loads the training set and its
#       shade kernel converts sRGB -> linear unless that flag is set; read here as nerf.linear_colors if the key
#       exists, else dataset.is_hdr, else False (LDR images: convert).
_NGP_GRID = 128
_NGP_MIN_OPTICAL_THICKNESS = 0.01
_MLP_PARAMS = sum(r * c for _, r, c in MLP_SHAPES)


def _morton_index_table() -> np.ndarray:
    """morton[z, y, x] for a 128^3 grid: bit i of x -> bit 3i, y -> 3i+1, z -> 3i+2."""
    def spread(v):
        out = np.zeros_like(v)
        for i in range(7):
            out |= ((v >> i) & 1) << (3 * i)
        return out

    a = spread(np.arange(_NGP_GRID, dtype=np.int64))
    return (a[None, None, :]) | (a[None, :, None] << 1) | (a[:, None, None] << 2)


def _grid_entries(n_levels, log2_hashmap, base_res, per_level_scale) -> int:
    total, T = 0, 1 << log2_hashmap
    for l in range(n_levels):
        scale = 2.0 ** (l * math.log2(per_level_scale)) * base_res - 1.0
        res = int(math.ceil(scale)) + 1
        n = min(res**3, T)
        total += min((n + 7) // 8 * 8, T)
    return total


def occupancy_from_density_grid(density: np.ndarray) -> np.ndarray:
    """density: float [cascades, 128^3] in Morton order -> uint8 bitfield, x fastest."""
    cascades = density.shape[0]
    morton = _morton_index_table().reshape(-1)
    mean = float(np.maximum(density[0].astype(np.float64), 0.0).mean())
    thresh = min(_NGP_MIN_OPTICAL_THICKNESS, mean)
    G, h, q = _NGP_GRID, _NGP_GRID // 2, _NGP_GRID // 4
    levels = []
    for c in range(cascades):
        occ = (density[c].astype(np.float32)[morton] > thresh).reshape(G, G, G)  # [z, y, x]
        if c > 0:
            pooled = levels[-1].reshape(h, 2, h, 2, h, 2).any(axis=(1, 3, 5))
            occ[q:q + h, q:q + h, q:q + h] |= pooled
        levels.append(occ)
    bits = np.stack(levels).reshape(-1)
    return np.packbits(bits, bitorder="little")


def from_instant_ngp(d: Dict) -> NerfSnapshot:
    """Unpacked instant-ngp snapshot dict -> NerfSnapshot (layout in the comment above)."""
    enc, s = d["encoding"], d["snapshot"]
    if enc.get("otype", "HashGrid") != "HashGrid":
        raise _lib.PxtError(f"unsupported position encoding {enc.get('otype')!r}: the renderer implements HashGrid")
    net, rgb = d.get("network", {}), d.get("rgb_network", {})
    if (net.get("n_neurons", 64), net.get("n_hidden_layers", 1), rgb.get("n_neurons", 64),
            rgb.get("n_hidden_layers", 2)) != (64, 1, 64, 2):
        raise _lib.PxtError("unsupported MLP shape: the renderer implements 64-wide 1+2 hidden layer networks "
                            "(instant-ngp configs/nerf/base.json)")
    if s.get("params_type", "__half") != "__half":
        raise _lib.PxtError(f"unsupported params_type {s.get('params_type')!r} (expected __half)")
    nerf = s.get("nerf", {})
    ds = nerf.get("dataset", {})
    aabb_scale = float(ds.get("aabb_scale", nerf.get("aabb_scale", 1)))
    n_levels, F = int(enc.get("n_levels", 16)), int(enc.get("n_features_per_level", 2))
    log2_T, base = int(enc.get("log2_hashmap_size", 19)), int(enc.get("base_resolution", 16))
    pls = float(enc.get("per_level_scale", 0.0))
    if pls <= 0.0:  # instant-ngp derives it from desired_resolution (2048) x aabb_scale
        desired = float(enc.get("desired_resolution", 2048.0))
        pls = math.exp(math.log(desired * aabb_scale / base) / (n_levels - 1))
    params = np.frombuffer(s["params_binary"], np.float16)
    n_grid = _grid_entries(n_levels, log2_T, base, pls) * F
    if params.size != _MLP_PARAMS + n_grid:
        raise _lib.PxtError(f"params_binary holds {params.size} values; expected {_MLP_PARAMS} MLP + {n_grid} "
                            "hash-grid values for this encoding config")
    cells = _NGP_GRID**3
    raw = s["density_grid_binary"]
    expect_c = int(round(math.log2(max(aabb_scale, 1.0)))) + 1
    for c, dt in ((expect_c, np.float16), (expect_c, np.float32), (8, np.float16), (8, np.float32)):
        if len(raw) == c * cells * np.dtype(dt).itemsize:
            density = np.frombuffer(raw, dt).reshape(c, cells)[:expect_c]
            break
    else:
        raise _lib.PxtError(f"density_grid_binary has {len(raw)} bytes: not [cascades][128^3] fp16/fp32")
    offset = ds.get("offset", [0.5, 0.5, 0.5])
    if max(offset) != min(offset):
        raise _lib.PxtError(f"per-axis dataset offset {offset} is not supported")
    k1 = 0.0
    meta = ds.get("metadata") or []
    if meta and isinstance(meta[0], dict):
        k1 = _training_lens_k1(meta[0])
    return NerfSnapshot(
        grid=params[_MLP_PARAMS:].reshape(-1, F).copy(), mlp=params[:_MLP_PARAMS].copy(),
        occupancy=occupancy_from_density_grid(density), n_levels=n_levels, n_features=F, log2_hashmap=log2_T,
        base_res=base, per_level_scale=pls, cascades=expect_c, aabb_scale=aabb_scale,
        cone_angle=0.0 if aabb_scale <= 1.0 else 1.0 / 256.0, scale=float(ds.get("scale", 0.33)),
        offset=float(offset[0]), k1=k1,
        linear_colors=bool(nerf.get("linear_colors", ds.get("is_hdr", False))))


def _training_lens_k1(meta0: Dict) -> float:
    """Radial k1 of the training lens from metadata[0].  Two encodings are accepted, because the
    layout cannot be checked against instant-ngp here (its source is not in the reference tree):
    ``{"mode": m, "params": [k1, k2, p1, p2, ...]}`` and named keys (``k1`` / ``k2`` / ``p1`` / ``p2``;
    ``ftheta_*`` marks a lens this renderer does not model).  A lens record that is present but
    unreadable raises instead of silently rendering without distortion."""
    cd = meta0.get("camera_distortion")
    if cd is None:
        cd = meta0.get("lens")
    if cd is None:
        return 0.0
    if not isinstance(cd, dict):
        raise _lib.PxtError(f"snapshot lens record of type {type(cd).__name__} is not understood")
    if any(str(k).startswith("ftheta") for k in cd) or str(cd.get("mode", "")).lower() in ("ftheta", "2", "fisheye"):
        raise _lib.PxtError("the snapshot was trained with an f-theta / fisheye lens, which this renderer does not model")
    if "k1" in cd:
        return float(cd["k1"])
    if "params" in cd:
        prm = list(cd["params"]) or [0.0]
        mode = cd.get("mode", 1)
        if mode in (0, "0", "Perspective", "perspective", None):
            return 0.0
        if mode in (1, "1", "OpenCV", "opencv"):
            return float(prm[0])
        # instant-ngp's other lens modes (LatLong, OpenCVFisheye, Equirectangular, ...) are not radial-k1 lenses
        raise _lib.PxtError(f"the snapshot's lens mode {mode!r} is not modelled by this renderer "
                            "(perspective and OpenCV radial k1 are)")
    if not cd:
        return 0.0
    raise _lib.PxtError(f"snapshot lens record has none of the known keys (k1 / params): {sorted(map(str, cd))}")


def to_instant_ngp(snap: NerfSnapshot) -> Dict:
    """NerfSnapshot -> dict in instant-ngp's snapshot layout (inverse of from_instant_ngp up to
    the density values: occupied cells get density 1, free cells 0)."""
    cells = _NGP_GRID**3
    bits = np.unpackbits(snap.occupancy, bitorder="little")[: snap.cascades * cells].reshape(snap.cascades, cells)
    morton = _morton_index_table().reshape(-1)
    density = np.zeros((snap.cascades, cells), np.float16)
    for c in range(snap.cascades):
        density[c, morton] = bits[c].astype(np.float16)
    params = np.concatenate([snap.mlp.astype(np.float16).reshape(-1), snap.grid.astype(np.float16).reshape(-1)])
    return {
        "encoding": {"otype": "HashGrid", "n_levels": snap.n_levels, "n_features_per_level": snap.n_features,
                     "log2_hashmap_size": snap.log2_hashmap, "base_resolution": snap.base_res,
                     "per_level_scale": snap.per_level_scale},
        "dir_encoding": {"otype": "Composite", "nested": [
            {"n_dims_to_encode": 3, "otype": "SphericalHarmonics", "degree": 4},
            {"otype": "Identity", "n_bins": 4, "degree": 4}]},
        "network": {"otype": "FullyFusedMLP", "activation": "ReLU", "output_activation": "None", "n_neurons": 64,
                    "n_hidden_layers": 1},
        "rgb_network": {"otype": "FullyFusedMLP", "activation": "ReLU", "output_activation": "None",
                        "n_neurons": 64, "n_hidden_layers": 2},
        "snapshot": {
            "version": 1, "n_params": int(params.size), "params_type": "__half",
            "params_binary": params.tobytes(), "density_grid_size": _NGP_GRID,
            "density_grid_binary": density.tobytes(),
            "nerf": {"dataset": {"scale": snap.scale, "offset": [snap.offset] * 3, "aabb_scale": snap.aabb_scale,
                                 "is_hdr": bool(snap.linear_colors),
                                 "metadata": [{"camera_distortion": {"mode": 1, "params": [snap.k1, 0.0, 0.0, 0.0]}}]}},
        },
    }


def nerf_matrix_to_ngp(nerf_c2w: np.ndarray, scale: float, offset: float) -> np.ndarray:
    """instant-ngp's nerf_matrix_to_ngp (what set_nerf_camera_matrix applies): flip the y/z
    camera axes, scale + offset the origin, cycle the world axes (x,y,z) <- (y,z,x)."""
    m = np.array(nerf_c2w, dtype=np.float64)[:3, :4].copy()
    m[:, 1] *= -1
    m[:, 2] *= -1
    m[:, 3] = m[:, 3] * scale + offset
    return m[[1, 2, 0], :]


class _Aabb:
    def __init__(self):
        self.min = [0.0, 0.0, 0.0]
        self.max = [1.0, 1.0, 1.0]


class _NerfSettings:
    def __init__(self):
        self.sharpen = 0.0
        self.render_with_camera_distortion = False
        self.rendering_min_transmittance = 0.01
        self.cone_angle_constant = 1.0 / 256.0


class Testbed:
    """pyngp.Testbed stand-in (inference only) backed by the HIP renderer."""

    __test__ = False  # not a pytest class

    def __init__(self, mode=TestbedMode.Nerf, device: Optional[torch.device] = None):
        self.mode = mode
        self.device = torch.device(device if device is not None else "cuda:0")
        self.nerf = _NerfSettings()
        self.background_color = [0.0, 0.0, 0.0, 1.0]
        self.snap_to_pixel_centers = False
        self.fov_axis = 0
        self.fov = 50.0
        self.shall_train = False
        self.render_aabb = _Aabb()
        self.exposure = 0.0
        self.render_mode = RenderMode.Shade
        self._cam_ngp = np.eye(4)[:3]
        self._ctx = None
        self._ctx_side = None  # a second renderer context over the same snapshot (see _side_ctx_int)
        self._snap: Optional[NerfSnapshot] = None
        self._stats = None
        self.stats_accum = None
        self.n_renders = 0
        self._pipelines = 0  # what set_pipelines() asked for (0: the library default); per-call overrides restore THIS
        self._cam_ring = None  # pinned camera records of pose-driven renders (see _next_cam_out)
        self._cam_next = 0

    def _next_cam_out(self) -> torch.Tensor:
        """A pinned 16-float record for the camera kernel of a pose-driven render.  The kernel receives the raw
        pointer, so torch's host allocator cannot know when the block is free again; the records therefore live
        as long as the testbed and are recycled round-robin - eight of them, at most two are written per frame and
        a frame's kernels have finished (its LM result was read) long before eight more renders are queued."""
        if self._cam_ring is None:
            self._cam_ring = [torch.zeros(16, dtype=torch.float32).pin_memory() for _ in range(8)]
        buf = self._cam_ring[self._cam_next]
        self._cam_next = (self._cam_next + 1) % len(self._cam_ring)
        buf.zero_()
        return buf

    # class-attribute style access used by pixtrack: testbed.render_mode.Depth
    RenderMode = RenderMode

    def __del__(self):
        try:
            for name in ("_ctx", "_ctx_side"):
                if getattr(self, name, None):
                    _lib.lib().pxt_ngp_destroy(getattr(self, name))
                    setattr(self, name, None)
        except Exception:
            pass

    def _create_ctx(self, snap: NerfSnapshot):
        L = _lib.lib()
        model = _lib.NgpModel(snap.n_levels, snap.n_features, snap.log2_hashmap, snap.base_res, snap.per_level_scale,
                              snap.cascades, snap.aabb_scale, snap.cone_angle, 1.0 / snap.scale,
                              1 if snap.linear_colors else 0)
        grid = np.ascontiguousarray(snap.grid.astype(np.float16))
        mlp = np.ascontiguousarray(snap.mlp.astype(np.float16))
        occ = np.ascontiguousarray(snap.occupancy.astype(np.uint8))
        ctx = C.c_void_p()
        with torch.cuda.device(self.device):
            _lib.check(
                L.pxt_ngp_create(C.byref(model), grid.ctypes.data, grid.size, mlp.ctypes.data, mlp.size,
                                 occ.ctypes.data, occ.size, C.byref(ctx)), "pxt_ngp_create")
        return ctx

    def load_snapshot(self, path_or_snapshot):
        snap = path_or_snapshot if isinstance(path_or_snapshot, NerfSnapshot) else load_snapshot_file(str(path_or_snapshot))
        if self.device.type != "cuda":
            raise _lib.PxtError("the NeRF renderer needs a ROCm device; no CPU path exists")
        ctx = self._create_ctx(snap)
        for name in ("_ctx", "_ctx_side"):
            if getattr(self, name):
                _lib.lib().pxt_ngp_destroy(getattr(self, name))
                setattr(self, name, None)
        self._ctx, self._snap = ctx, snap
        self.nerf.cone_angle_constant = snap.cone_angle

    def _side_ctx_int(self) -> int:
        """A second context over the same snapshot (pxt_ngp_create_shared: its own ray lists, counters and camera slot; the
        30 MB of tables are SHARED with the first): what the second of a frame's two renders of different views - the
        tracker's mask at the query camera, its reference image at the reference camera - runs through, both in one
        chain of launches (render_frame_pair_device)."""
        assert self._ctx is not None, "load_snapshot first"
        if self._ctx_side is None:
            ctx = C.c_void_p()
            with torch.cuda.device(self.device):
                _lib.check(_lib.lib().pxt_ngp_create_shared(self._ctx, C.byref(ctx)), "pxt_ngp_create_shared")
            self._ctx_side = ctx
        c = self._ctx_side
        return int(c.value) if hasattr(c, "value") else int(c)

    def set_nerf_camera_matrix(self, nerf_c2w_3x4):
        assert self._snap is not None, "load_snapshot first"
        self._cam_ngp = nerf_matrix_to_ngp(np.asarray(nerf_c2w_3x4), self._snap.scale, self._snap.offset)

    def _view(self) -> list:
        """The 25-float view record of torch.ops.pixtrack.ngp_render (ops.VIEW_FLOATS): camera 3x4,
        focal, k1, render box min / max, background RGBA, minimum transmittance."""
        return ([float(x) for x in np.asarray(self._cam_ngp, np.float32).reshape(-1)]
                + [0.0, float(self._snap.k1) if self.nerf.render_with_camera_distortion else 0.0]
                + [float(x) for x in self.render_aabb.min] + [float(x) for x in self.render_aabb.max]
                + [float(x) for x in self.background_color] + [float(self.nerf.rendering_min_transmittance)])

    def _view_for(self, width: int, height: int, fov=None) -> list:
        v = self._view()
        res = width if self.fov_axis == 0 else height
        v[12] = float(np.float32(0.5 * res / math.tan(0.5 * math.radians(self.fov if fov is None else fov))))
        return v

    def render_device(self, width: int, height: int, spp: int = 8, linear: bool = True,
                      collect_stats: bool = False, side: bool = False, pipelines: int = 0) -> torch.Tensor:
        """float32 [H, W, 4] on the device, linear premultiplied RGBA.  ``side`` / ``pipelines``: as
        render_from_pose_device (two renders of different views side by side on two streams)."""
        assert linear, "pixtrack renders with linear=True (run_vis_on_poses.py:51)"
        assert self._ctx is not None, "load_snapshot first"
        if not self.snap_to_pixel_centers:
            raise _lib.PxtError("only snap_to_pixel_centers=True is implemented (ingp_utils.py:36)")
        out = torch.empty(height, width, 4, device=self.device, dtype=torch.float32)
        stats = self.stats_accum  # running totals across launches when set (bench)
        if collect_stats:
            stats = torch.zeros(4, dtype=torch.int64, device=self.device)
        ctx = self._side_ctx_int() if side else self._ctx_int()
        if pipelines and not side:
            _lib.check(_lib.lib().pxt_ngp_set_pipelines(self._ctx, int(pipelines)), "pxt_ngp_set_pipelines")
        try:
            ops.ngp_render(ctx, self._view_for(width, height), int(width), int(height), int(spp),
                           int(self.render_mode), out, stats)
        finally:
            if pipelines and not side:
                _lib.check(_lib.lib().pxt_ngp_set_pipelines(self._ctx, self._pipelines), "pxt_ngp_set_pipelines")
        if collect_stats:
            self._stats = stats
        self.n_renders += 1
        return out

    def render_both_device(self, width: int, height: int, spp: int = 8):
        """(Shade RGBA, Depth RGBA) of the current view from ONE march; each equals what
        render_device returns in the corresponding render_mode, bit for bit."""
        assert self._ctx is not None, "load_snapshot first"
        if not self.snap_to_pixel_centers:
            raise _lib.PxtError("only snap_to_pixel_centers=True is implemented (ingp_utils.py:36)")
        rgba = torch.empty(height, width, 4, device=self.device, dtype=torch.float32)
        depth = torch.empty(height, width, 4, device=self.device, dtype=torch.float32)
        ops.ngp_render_both(self._ctx_int(), self._view_for(width, height), int(width), int(height), int(spp),
                            rgba, depth, self.stats_accum)
        self.n_renders += 1
        return rgba, depth

    def pose_conversion(self, nerf2sfm) -> list:
        """The 27 doubles pxt_ngp_render_both_from_pose needs to turn a world->camera pose into this testbed's
        camera: nerf2sfm centroid, 3 / avglen, R (4x4 row-major), totp, then the snapshot's scale and offset."""
        off = np.broadcast_to(np.asarray(self._snap.offset, np.float64), (3,))
        return ([float(x) for x in np.asarray(nerf2sfm["centroid"], np.float64).reshape(3)]
                + [3.0 / float(nerf2sfm["avglen"])]
                + [float(x) for x in np.asarray(nerf2sfm["R"], np.float64).reshape(16)]
                + [float(x) for x in np.asarray(nerf2sfm["totp"], np.float64).reshape(3)]
                + [float(self._snap.scale)] + [float(x) for x in off])

    def render_both_from_pose_device(self, width: int, height: int, spp: int, pose_record: torch.Tensor, conv: list):
        """render_both_device for a pose the host has not seen yet: `pose_record` is the pinned record of an
        ENQUEUED refinement (optimizer.PendingLM.buf); the camera is derived from it on the device, in stream
        order.  Returns (rgba, depth, cam_out): cam_out (pinned, 16 floats) receives the 12 camera floats the
        render used and, last, cam_out[12] = 1."""
        assert self._ctx is not None, "load_snapshot first"
        if not self.snap_to_pixel_centers:
            raise _lib.PxtError("only snap_to_pixel_centers=True is implemented (ingp_utils.py:36)")
        rgba = torch.empty(height, width, 4, device=self.device, dtype=torch.float32)
        depth = torch.empty(height, width, 4, device=self.device, dtype=torch.float32)
        cam_out = self._next_cam_out()
        ops.ngp_render_both_from_pose(self._ctx_int(), self._view_for(width, height), pose_record, conv, int(width),
                                      int(height), int(spp), 0, rgba, depth, cam_out, self.stats_accum)
        self.n_renders += 1
        return rgba, depth, cam_out

    def render_from_pose_device(self, width: int, height: int, spp: int, pose_record: torch.Tensor, conv: list,
                                side: bool = False, pipelines: int = 0):
        """render_device (in the current render_mode) for a pose that is still on the device; see
        render_both_from_pose_device.  Returns (rgba, cam_out).  ``side``: through the second context
        (_side_ctx_int; the caller puts the call on another stream); ``pipelines``: ray slices rendered side by
        side by THIS call (0 = the default)."""
        assert self._ctx is not None, "load_snapshot first"
        if not self.snap_to_pixel_centers:
            raise _lib.PxtError("only snap_to_pixel_centers=True is implemented (ingp_utils.py:36)")
        out = torch.empty(height, width, 4, device=self.device, dtype=torch.float32)
        cam_out = self._next_cam_out()
        ctx = self._side_ctx_int() if side else self._ctx_int()
        if pipelines and not side:
            _lib.check(_lib.lib().pxt_ngp_set_pipelines(self._ctx, int(pipelines)), "pxt_ngp_set_pipelines")
        try:
            ops.ngp_render_both_from_pose(ctx, self._view_for(width, height), pose_record, conv, int(width),
                                          int(height), int(spp), int(self.render_mode), out, None, cam_out, self.stats_accum)
        finally:
            if pipelines and not side:
                _lib.check(_lib.lib().pxt_ngp_set_pipelines(self._ctx, self._pipelines), "pxt_ngp_set_pipelines")
        self.n_renders += 1
        return out, cam_out

    def camera_slot(self, side: bool = False) -> int:
        """Device address of the context's 12-float camera slot (pxt_ngp_camera_slot): the LM kernel's epilogue writes
        the next render's camera there (ops.lm_refine cam_slots) and render_frame_device(from_slot=True) reads it."""
        ctx = self._side_ctx_int() if side else self._ctx_int()
        return int(_lib.lib().pxt_ngp_camera_slot(ctx))

    def render_frame_device(self, width: int, height: int, spp: int = 8, mode: int = 2, from_slot: bool = False,
                            side: bool = False, pipelines: int = 0, want_float: bool = False):
        """One render whose last kernel writes what the tracking loop consumes (pxt_ngp_render_frame): returns a dict with
        ``rgb_u8`` uint8 [H, W, 3] (modes 0 / 2: get_nerf_image's image of the Shade render) and ``depth_nz`` uint8 [H, W]
        (modes 1 / 2: get_mask's `uint8(depth * 255) != 0` plane), plus the float images ``rgba`` / ``depth`` when
        ``want_float``.  mode 0 Shade, 1 Depth, 2 both from one march.  ``from_slot``: the camera is whatever the LM
        kernel ahead in the stream wrote into camera_slot(side) - the render of a pose the host has not seen yet."""
        assert self._ctx is not None, "load_snapshot first"
        if not self.snap_to_pixel_centers:
            raise _lib.PxtError("only snap_to_pixel_centers=True is implemented (ingp_utils.py:36)")
        dev, out = self.device, {}
        if mode != 1:
            out["rgb_u8"] = torch.empty(height, width, 3, device=dev, dtype=torch.uint8)
        if mode != 0:
            out["depth_nz"] = torch.empty(height, width, device=dev, dtype=torch.uint8)
        if want_float:
            out["rgba"] = torch.empty(height, width, 4, device=dev, dtype=torch.float32)
            if mode == 2:
                out["depth"] = torch.empty(height, width, 4, device=dev, dtype=torch.float32)
        ctx = self._side_ctx_int() if side else self._ctx_int()
        if pipelines and not side:
            _lib.check(_lib.lib().pxt_ngp_set_pipelines(self._ctx, int(pipelines)), "pxt_ngp_set_pipelines")
        try:
            ops.ngp_render_frame(ctx, self._view_for(width, height), int(width), int(height), int(spp), int(mode),
                                 bool(from_slot), out.get("rgba"), out.get("depth"), out.get("rgb_u8"),
                                 out.get("depth_nz"), self.stats_accum)
        finally:
            if pipelines and not side:
                _lib.check(_lib.lib().pxt_ngp_set_pipelines(self._ctx, self._pipelines), "pxt_ngp_set_pipelines")
        self.n_renders += 1
        return out

    @staticmethod
    def render_frame_batch_device(testbeds, sizes, spp: int = 8, mode=2, from_slot: bool = False, workspace=None,
                                  sides=None, fovs=None):
        """render_frame_device for K renders as ONE staged chain of launches (pxt_ngp_render_frame_batch): K testbeds (K objects,
        each with its own NeRF), or - ``sides[k]`` true - a testbed's second context (the second of a frame's two renders of one
        NeRF).  ``sizes[k]`` = (width, height); ``mode``: one int for all or one per render; ``fovs[k]`` (optional): the
        render's field of view instead of the testbed's current one.  Every testbed's current camera (or, with
        ``from_slot``, the context's camera slot) is used as render_frame_device would.  Returns one dict per render
        (``rgb_u8``, ``depth_nz`` as its mode provides), bit for bit what K render_frame_device calls return.
        ``workspace``: a device uint8 tensor of batch_workspace_bytes(K) the caller keeps per stream (made here when None;
        K <= 2 needs none: the records travel as kernel arguments)."""
        K = len(testbeds)
        assert K >= 1 and len(sizes) == K
        modes = [int(mode)] * K if isinstance(mode, int) else [int(m) for m in mode]
        assert len(modes) == K
        dev = testbeds[0].device
        outs, views, flat_sizes, ctxs = [], [], [], []
        for k, (tb, (w, h)) in enumerate(zip(testbeds, sizes)):
            assert tb._ctx is not None, "load_snapshot first"
            if not tb.snap_to_pixel_centers:
                raise _lib.PxtError("only snap_to_pixel_centers=True is implemented (ingp_utils.py:36)")
            o = {}
            if modes[k] != 1:
                o["rgb_u8"] = torch.empty(h, w, 3, device=dev, dtype=torch.uint8)
            if modes[k] != 0:
                o["depth_nz"] = torch.empty(h, w, device=dev, dtype=torch.uint8)
            outs.append(o)
            views += tb._view_for(w, h, None if fovs is None else fovs[k])
            flat_sizes += [int(w), int(h)]
            ctxs.append(tb._side_ctx_int() if (sides is not None and sides[k]) else tb._ctx_int())
        if workspace is None:
            workspace = torch.empty(Testbed.batch_workspace_bytes(K) if K > 2 else 256, dtype=torch.uint8, device=dev)
        stats = [tb.stats_accum for tb in testbeds] if all(tb.stats_accum is not None for tb in testbeds) else []
        ops.ngp_render_frame_batch(ctxs, views, flat_sizes, int(spp), modes, bool(from_slot),
                                   [o["rgb_u8"] for o in outs if "rgb_u8" in o],
                                   [o["depth_nz"] for o in outs if "depth_nz" in o], workspace, stats)
        for tb in testbeds:
            tb.n_renders += 1
        return outs

    def render_frame_pair_device(self, depth_view, shade_view, spp: int = 8, from_slot: bool = False, workspace=None):
        """A frame's two renders of DIFFERENT cameras at one pose - ``depth_view`` = (width, height, fov) of the mask's Depth
        render (the query camera), ``shade_view`` of the reference image's Shade render (SfM camera 1 x reference_scale) - as
        ONE chain of launches on the current stream (pixtrack/pose_trackers/pixloc_tracker_r9.py:145-152, 207-214 render them
        one after the other through one testbed).  The Depth render runs through this testbed's first context, the Shade
        render through its second (shared tables).  Returns (depth_nz, rgb_u8), bit for bit what render_frame_device
        returns for each."""
        (dw, dh, dfov), (sw, sh, sfov) = depth_view, shade_view
        outs = Testbed.render_frame_batch_device([self, self], [(dw, dh), (sw, sh)], spp, mode=[1, 0], from_slot=from_slot,
                                                 workspace=workspace, sides=[False, True], fovs=[dfov, sfov])
        return outs[0]["depth_nz"], outs[1]["rgb_u8"]

    @staticmethod
    def batch_workspace_bytes(n: int) -> int:
        return int(_lib.lib().pxt_ngp_batch_workspace_bytes(int(n)))

    def _ctx_int(self) -> int:
        return int(self._ctx.value) if hasattr(self._ctx, "value") else int(self._ctx)

    def render(self, width: int, height: int, spp: int = 8, linear: bool = True) -> np.ndarray:
        return self.render_device(width, height, spp, linear).cpu().numpy()

    def set_pipelines(self, n: int = 0):
        """Number of ray slices (pipes) a large render is cut into (0: default of 2; 1: every launch carries one stage).
        Per-call `pipelines=` overrides of render_device / render_from_pose_device return to this value afterwards."""
        _lib.check(_lib.lib().pxt_ngp_set_pipelines(self._ctx, int(n)), "pxt_ngp_set_pipelines")
        self._pipelines = int(n)

    def timing_enable(self, every_nth: int = 1):
        """HIP events around the launches that carry a shade stage (ngp_stage_kernel) of every ``every_nth``-th render (0 / False: off)."""
        _lib.check(_lib.lib().pxt_ngp_timing_enable(self._ctx, int(every_nth)), "pxt_ngp_timing_enable")

    def timing_read(self):
        """(total ms, launches) of the shade-carrying launches since the last read (HIP events on the render stream)."""
        ms, n = C.c_float(0), C.c_int32(0)
        _lib.check(_lib.lib().pxt_ngp_timing_read(self._ctx, C.byref(ms), C.byref(n)), "pxt_ngp_timing_read")
        return float(ms.value), int(n.value)

    def read_stats(self):
        """(samples composited, rays that hit the box) of the last render_device(collect_stats=True)."""
        s = self._stats.cpu().tolist()
        return {"samples": s[0], "rays_hit": s[1]}
