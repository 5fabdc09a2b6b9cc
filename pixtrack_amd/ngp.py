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
from dataclasses import dataclass
from enum import IntEnum
from typing import Dict, Optional, Sequence

import numpy as np
import torch

from . import _lib

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
                     "dataset": {"scale": snap.scale, "offset": [snap.offset] * 3, "k1": snap.k1}},
        },
    }
    with open(path, "wb") as f:
        f.write(msgpack.packb(d, use_bin_type=True))


def load_snapshot_file(path: str) -> NerfSnapshot:
    import msgpack

    with open(path, "rb") as f:
        d = msgpack.unpackb(f.read(), raw=False)
    enc, s = d["encoding"], d["snapshot"]
    if "grid_binary" not in s:
        raise _lib.PxtError(
            "this is not a pixtrack_amd snapshot (instant-ngp's params_binary layout is not "
            "wired up yet: SURVEY.md section 8f rank 2)")
    nerf = s["nerf"]
    return NerfSnapshot(
        grid=np.frombuffer(s["grid_binary"], np.float16).reshape(-1, enc["n_features_per_level"]).copy(),
        mlp=np.frombuffer(s["mlp_binary"], np.float16).copy(),
        occupancy=np.frombuffer(s["density_grid_binary"], np.uint8).copy(),
        n_levels=enc["n_levels"], n_features=enc["n_features_per_level"], log2_hashmap=enc["log2_hashmap_size"],
        base_res=enc["base_resolution"], per_level_scale=enc["per_level_scale"], cascades=nerf["cascades"],
        aabb_scale=nerf["aabb_scale"], cone_angle=nerf["cone_angle_constant"], scale=nerf["dataset"]["scale"],
        offset=nerf["dataset"]["offset"][0], k1=nerf["dataset"].get("k1", 0.0),
    )


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
        self._snap: Optional[NerfSnapshot] = None
        self._stats = None
        self.stats_accum = None
        self.n_renders = 0

    # class-attribute style access used by pixtrack: testbed.render_mode.Depth
    RenderMode = RenderMode

    def __del__(self):
        try:
            if self._ctx:
                _lib.lib().pxt_ngp_destroy(self._ctx)
                self._ctx = None
        except Exception:
            pass

    def load_snapshot(self, path_or_snapshot):
        snap = path_or_snapshot if isinstance(path_or_snapshot, NerfSnapshot) else load_snapshot_file(str(path_or_snapshot))
        if self.device.type != "cuda":
            raise _lib.PxtError("the NeRF renderer needs a ROCm device; no CPU path exists")
        L = _lib.lib()
        model = _lib.NgpModel(snap.n_levels, snap.n_features, snap.log2_hashmap, snap.base_res, snap.per_level_scale,
                              snap.cascades, snap.aabb_scale, snap.cone_angle, 1.0 / snap.scale)
        grid = np.ascontiguousarray(snap.grid.astype(np.float16))
        mlp = np.ascontiguousarray(snap.mlp.astype(np.float16))
        occ = np.ascontiguousarray(snap.occupancy.astype(np.uint8))
        ctx = C.c_void_p()
        with torch.cuda.device(self.device):
            _lib.check(
                L.pxt_ngp_create(C.byref(model), grid.ctypes.data, grid.size, mlp.ctypes.data, mlp.size,
                                 occ.ctypes.data, occ.size, C.byref(ctx)), "pxt_ngp_create")
        if self._ctx:
            L.pxt_ngp_destroy(self._ctx)
        self._ctx, self._snap = ctx, snap
        self.nerf.cone_angle_constant = snap.cone_angle

    def set_nerf_camera_matrix(self, nerf_c2w_3x4):
        assert self._snap is not None, "load_snapshot first"
        self._cam_ngp = nerf_matrix_to_ngp(np.asarray(nerf_c2w_3x4), self._snap.scale, self._snap.offset)

    def _view(self, width: int, height: int, spp: int) -> _lib.NgpView:
        v = _lib.NgpView()
        v.cam[:] = [float(x) for x in np.asarray(self._cam_ngp, np.float32).reshape(-1)]
        res = width if self.fov_axis == 0 else height
        v.focal = float(np.float32(0.5 * res / math.tan(0.5 * math.radians(self.fov))))
        v.k1 = float(self._snap.k1) if self.nerf.render_with_camera_distortion else 0.0
        v.aabb_min[:] = [float(x) for x in self.render_aabb.min]
        v.aabb_max[:] = [float(x) for x in self.render_aabb.max]
        v.background[:] = [float(x) for x in self.background_color]
        v.min_transmittance = float(self.nerf.rendering_min_transmittance)
        v.width, v.height, v.spp = int(width), int(height), int(spp)
        v.mode = int(self.render_mode)
        return v

    def render_device(self, width: int, height: int, spp: int = 8, linear: bool = True,
                      collect_stats: bool = False) -> torch.Tensor:
        """float32 [H, W, 4] on the device, linear premultiplied RGBA."""
        assert linear, "pixtrack renders with linear=True (run_vis_on_poses.py:51)"
        assert self._ctx is not None, "load_snapshot first"
        if not self.snap_to_pixel_centers:
            raise _lib.PxtError("only snap_to_pixel_centers=True is implemented (ingp_utils.py:36)")
        out = torch.empty(height, width, 4, device=self.device, dtype=torch.float32)
        stats = self.stats_accum  # running totals across launches when set (bench)
        if collect_stats:
            stats = torch.zeros(4, dtype=torch.int64, device=self.device)
        v = self._view(width, height, spp)
        _lib.check(
            _lib.lib().pxt_ngp_render(self._ctx, C.byref(v), out.data_ptr(), _lib.dptr(stats),
                                      _lib.stream_ptr(self.device)), "pxt_ngp_render")
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
        v = self._view(width, height, spp)
        _lib.check(
            _lib.lib().pxt_ngp_render_both(self._ctx, C.byref(v), rgba.data_ptr(), depth.data_ptr(),
                                           _lib.dptr(self.stats_accum), _lib.stream_ptr(self.device)),
            "pxt_ngp_render_both")
        self.n_renders += 1
        return rgba, depth

    def render(self, width: int, height: int, spp: int = 8, linear: bool = True) -> np.ndarray:
        return self.render_device(width, height, spp, linear).cpu().numpy()

    def timing_enable(self, enable: bool = True):
        _lib.check(_lib.lib().pxt_ngp_timing_enable(self._ctx, int(enable)), "pxt_ngp_timing_enable")

    def timing_read(self):
        """(total ms, launches) of ngp_encode_kernel since the last read (HIP events on the render stream)."""
        ms, n = C.c_float(0), C.c_int32(0)
        _lib.check(_lib.lib().pxt_ngp_timing_read(self._ctx, C.byref(ms), C.byref(n)), "pxt_ngp_timing_read")
        return float(ms.value), int(n.value)

    def read_stats(self):
        """(samples composited, rays that hit the box, rays finished by the straggler kernel) of
        the last render_device(collect_stats=True)."""
        s = self._stats.cpu().tolist()
        return {"samples": s[0], "rays_hit": s[1], "tail_rays": s[2], "encoded_slots": s[3]}
