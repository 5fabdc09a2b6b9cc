"""ctypes binding of libpixtrack_hip.so (include/pixtrack_hip.h).

This is the whole Python<->native seam: plain pointers, sizes and a stream handle.
torch only supplies device memory (``tensor.data_ptr()``) and the current HIP stream.
There is NO fallback: if the library is missing or a call fails, this raises.
"""
from __future__ import annotations

import ctypes as C
import os
from pathlib import Path
from typing import Optional

import torch

_HERE = Path(__file__).resolve().parent
# PIXTRACK_HIP_LIB points at another build of the same ABI (A/B experiments, packaging)
LIB_PATH = Path(os.environ["PIXTRACK_HIP_LIB"]) if os.environ.get("PIXTRACK_HIP_LIB") else _HERE / "libpixtrack_hip.so"

PXT_MAX_LEVELS = 8
PXT_LM_LOG_STRIDE = 20
PXT_E_TIMEOUT = -3
ABI_VERSION = 11
PXT_LM_MAX_BATCH = 16
PXT_UNET_MAX_BATCH = 16
PXT_NGP_MAX_BATCH = 16


class PxtError(RuntimeError):
    """Infrastructure failure of the native path (never a tracking failure)."""


class LmLevel(C.Structure):
    _fields_ = [
        ("fmap", C.c_void_p),
        ("fref", C.c_void_p),
        ("h", C.c_int32),
        ("w", C.c_int32),
        ("C", C.c_int32),
        ("cstride", C.c_int32),
        ("cam", C.c_float * 10),
        ("ndist", C.c_int32),
        ("lambda_", C.c_float * 6),
    ]


class LmConf(C.Structure):
    _fields_ = [
        ("num_iters", C.c_int32),
        ("pad", C.c_int32),
        ("loss", C.c_int32),
        ("loss_alpha", C.c_float),
        ("loss_scale", C.c_float),
        ("grad_stop", C.c_float),
        ("dt_stop", C.c_float),
        ("dR_stop", C.c_float),
        ("min_valid", C.c_int32),
        ("n_workgroups", C.c_int32),
        ("spin_limit", C.c_int32),
        ("path", C.c_int32),
    ]


class SampleLevel(C.Structure):
    _fields_ = [
        ("fmap", C.c_void_p),
        ("out", C.c_void_p),
        ("h", C.c_int32),
        ("w", C.c_int32),
        ("C", C.c_int32),
        ("cstride", C.c_int32),
        ("cam", C.c_float * 10),
        ("ndist", C.c_int32),
        ("x0", C.c_int32),
        ("y0", C.c_int32),
        ("full_w", C.c_int32),
        ("full_h", C.c_int32),
    ]


class NgpModel(C.Structure):
    _fields_ = [
        ("n_levels", C.c_int32),
        ("n_features", C.c_int32),
        ("log2_hashmap", C.c_int32),
        ("base_res", C.c_int32),
        ("per_level_scale", C.c_float),
        ("grid_cascades", C.c_int32),
        ("aabb_scale", C.c_float),
        ("cone_angle", C.c_float),
        ("depth_scale", C.c_float),
        ("linear_colors", C.c_int32),
    ]


class NgpView(C.Structure):
    _fields_ = [
        ("cam", C.c_float * 12),
        ("focal", C.c_float),
        ("k1", C.c_float),
        ("aabb_min", C.c_float * 3),
        ("aabb_max", C.c_float * 3),
        ("background", C.c_float * 4),
        ("min_transmittance", C.c_float),
        ("width", C.c_int32),
        ("height", C.c_int32),
        ("spp", C.c_int32),
        ("mode", C.c_int32),
    ]


class NgpOutputs(C.Structure):
    _fields_ = [("rgba", C.c_void_p), ("depth_rgba", C.c_void_p), ("rgb_u8", C.c_void_p), ("depth_nz", C.c_void_p)]


class LmCamera(C.Structure):
    _fields_ = [("conv27", C.c_double * 27), ("cam_slot", C.c_void_p * 2), ("cam_out13", C.c_void_p)]


class LmProblem(C.Structure):
    _fields_ = [("p3d", C.c_void_p), ("point_mask", C.c_void_p), ("n_points", C.c_int32),
                ("levels_host", C.POINTER(LmLevel)), ("n_levels", C.c_int32), ("T_init_host", C.POINTER(C.c_float)),
                ("out", C.c_void_p), ("log", C.c_void_p), ("workspace", C.c_void_p), ("cam_host", C.POINTER(LmCamera))]


_lib: Optional[C.CDLL] = None

# name -> (restype, argtypes); every symbol include/pixtrack_hip.h declares.
_VP, _I32, _I64 = C.c_void_p, C.c_int32, C.c_int64
PROTOTYPES = {
    "pxt_version": (C.c_int, []),
    "pxt_last_error": (C.c_char_p, []),
    "pxt_device_cus": (C.c_int, [C.POINTER(C.c_int)]),
    "pxt_lm_refine": (
        C.c_int,
        [_VP, _VP, _I32, C.POINTER(LmLevel), _I32, _VP, C.POINTER(LmConf), _VP, _VP, _VP, _VP],
    ),
    "pxt_lm_refine_cam": (
        C.c_int,
        [_VP, _VP, _I32, C.POINTER(LmLevel), _I32, _VP, C.POINTER(LmConf), _VP, _VP, _VP, C.POINTER(LmCamera), _VP],
    ),
    "pxt_lm_workspace_bytes": (_I64, []),
    "pxt_lm_batch_workspace_bytes": (_I64, [_I32]),
    "pxt_lm_refine_batch": (C.c_int, [C.POINTER(LmProblem), _I32, C.POINTER(LmConf), _VP, _VP]),
    "pxt_sample_sparse": (C.c_int, [_VP, _I32, _VP, C.POINTER(SampleLevel), _I32, _I32, _I32, _VP, _VP]),
    "pxt_unet_create": (C.c_int, [_VP, _I64, C.POINTER(_VP)]),
    "pxt_unet_destroy": (C.c_int, [_VP]),
    "pxt_unet_workspace_bytes": (_I64, [_VP, _I32, _I32]),
    "pxt_unet_forward": (
        C.c_int,
        [_VP, _VP, _I32, _VP, _I32, _I32, C.POINTER(_VP), C.POINTER(_I32), _I32, _VP, _VP],
    ),
    "pxt_unet_set_defer_join": (C.c_int, [_VP, _I32]),
    "pxt_unet_set_batch_plan": (C.c_int, [_VP, _I32]),
    "pxt_unet_set_tile_skip": (C.c_int, [_VP, _I32]),
    "pxt_unet_pair_join": (C.c_int, [_VP, _VP]),
    "pxt_unet_activation_stats": (C.c_int, [_VP, _I32, _I32, _VP, _VP, _VP]),
    "pxt_unet_workspace_bytes_batch": (_I64, [_VP, _I32, _I32, _I32]),
    "pxt_unet_forward_batch": (
        C.c_int,
        [_VP, _I32, C.POINTER(_VP), C.POINTER(_I32), C.POINTER(_VP), _I32, _I32, C.POINTER(_VP), C.POINTER(_I32),
         C.POINTER(_I32), _VP, _VP],
    ),
    "pxt_unet_workspace_bytes_pair": (_I64, [_VP, C.POINTER(_I32), C.POINTER(_I32)]),
    "pxt_unet_forward_pair": (
        C.c_int,
        [_VP, C.POINTER(_VP), C.POINTER(_I32), C.POINTER(_VP), C.POINTER(_I32), C.POINTER(_I32), C.POINTER(_VP),
         C.POINTER(_I32), C.POINTER(_I32), _VP, _VP],
    ),
    "pxt_conv3x3_nhwc_f16": (C.c_int, [_VP, _I32, _I32, _I32, _VP, _VP, _I32, _I32, _VP, _VP]),
    "pxt_conv3x3_packed_bytes": (_I64, [_I32, _I32]),
    "pxt_conv3x3_pack_weights": (C.c_int, [_VP, _I32, _I32, _VP, _VP]),
    "pxt_conv3x3_packed": (C.c_int, [_VP, _I32, _I32, _I32, _VP, _VP, _I32, _I32, _VP, _VP, _I32, _I32, _VP, _I64, _VP]),
    "pxt_ngp_create": (C.c_int, [C.POINTER(NgpModel), _VP, _I64, _VP, _I64, _VP, _I64, C.POINTER(_VP)]),
    "pxt_ngp_destroy": (C.c_int, [_VP]),
    "pxt_ngp_render": (C.c_int, [_VP, C.POINTER(NgpView), _VP, _VP, _VP]),
    "pxt_ngp_render_both": (C.c_int, [_VP, C.POINTER(NgpView), _VP, _VP, _VP, _VP]),
    "pxt_ngp_render_both_from_pose": (C.c_int, [_VP, C.POINTER(NgpView), _VP, C.POINTER(C.c_double), _VP, _VP, _VP, _VP, _VP]),
    "pxt_ngp_render_frame": (C.c_int, [_VP, C.POINTER(NgpView), _I32, _I32, C.POINTER(NgpOutputs), _VP, _VP]),
    "pxt_ngp_camera_slot": (_VP, [_VP]),
    "pxt_ngp_batch_workspace_bytes": (_I64, [_I32]),
    "pxt_ngp_render_frame_batch": (C.c_int, [C.POINTER(_VP), C.POINTER(NgpView), _I32, C.POINTER(_I32), _I32,
                                              C.POINTER(NgpOutputs), C.POINTER(_VP), _VP, _VP]),
    "pxt_ngp_create_shared": (C.c_int, [_VP, C.POINTER(_VP)]),
    "pxt_ngp_set_pipelines": (C.c_int, [_VP, _I32]),
    "pxt_ngp_timing_enable": (C.c_int, [_VP, _I32]),
    "pxt_ngp_timing_read": (C.c_int, [_VP, C.POINTER(C.c_float), C.POINTER(_I32)]),
    "pxt_ngp_query": (C.c_int, [_VP, _VP, _VP, _I32, _VP, _VP]),
    "pxt_depth_mask": (C.c_int, [_VP, _I32, _I32, _I32, _I32, _VP, _VP, _VP]),
    "pxt_depth_mask_plane": (C.c_int, [_VP, _I32, _I32, _I32, _I32, _VP, _VP, _VP]),
    "pxt_rgba_to_u8": (C.c_int, [_VP, _I32, _I32, C.c_float, _VP, _VP]),
    "pxt_resize_linear": (C.c_int, [_VP, _I32, _I32, _I32, _VP, _I32, _I32, _VP]),
    "pxt_resize_activity": (C.c_int, [_VP, _VP, _I32, _I32, _I32, _I32, _VP, _VP]),
}


def lib() -> C.CDLL:
    """Loads the native library once; raises PxtError if it is absent or stale."""
    global _lib
    if _lib is not None:
        return _lib
    if not LIB_PATH.exists():
        raise PxtError(
            f"{LIB_PATH} not built. Run `python -m pixtrack_amd._build` (hipcc, gfx950). "
            "There is no CPU fallback for the product path."
        )
    L = C.CDLL(str(LIB_PATH), mode=os.RTLD_NOW | os.RTLD_LOCAL)
    missing = []
    for name, (res, args) in PROTOTYPES.items():
        try:
            fn = getattr(L, name)
        except AttributeError:
            missing.append(name)  # calling it later raises AttributeError (loud, no fallback)
            continue
        fn.restype = res
        fn.argtypes = args
    L._pxt_missing = missing
    if L.pxt_version() != ABI_VERSION:
        raise PxtError(f"ABI mismatch: library {L.pxt_version()} != binding {ABI_VERSION}; rebuild")
    _lib = L
    return L


def check(code: int, what: str) -> None:
    if code != 0:
        msg = lib().pxt_last_error().decode(errors="replace")
        raise PxtError(f"{what} failed with code {code}: {msg}")


def stream_ptr(device: torch.device) -> int:
    """The raw hipStream_t of torch's current stream on ``device``."""
    return int(torch.cuda.current_stream(device).cuda_stream)


def require_gpu(t: torch.Tensor, name: str) -> None:
    if not t.is_cuda:
        raise PxtError(f"{name} must live on a ROCm device (got {t.device}); no CPU path exists")


def host_pose12(T) -> "C.Array":
    """12 host floats (row-major R, then t) from a Pose / tensor / array, for *_host arguments."""
    data = T.as12() if hasattr(T, "as12") else T
    if torch.is_tensor(data):
        data = data.detach().cpu().reshape(-1).tolist()
    vals = [float(x) for x in data]
    assert len(vals) == 12
    return (C.c_float * 12)(*vals)


def dptr(t: Optional[torch.Tensor]) -> Optional[int]:
    return None if t is None else t.data_ptr()
