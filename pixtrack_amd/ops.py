"""``torch.ops.pixtrack.*``: the C-ABI entry points of libpixtrack_hip.so (include/pixtrack_hip.h)
registered as ``torch.library`` custom ops on the ROCm device (dispatch key CUDA == HIP on a
ROCm build of torch).  This is the seam SURVEY.md 8(b) / BASELINE north_star ask for: Python host
code hands ``torch.Tensor``s to the ops, the op bodies are the only place where ctypes touches the
hot path (pointers, sizes, the current HIP stream).

Reference call sites the ops stand in for:
  lm_refine            pixloc BaseRefiner.refine_pose_using_features -> opt.run per level
                       (pixtrack/localization/pixloc_pose_refiners.py:255-262)
  sample_sparse        PoseTrackerRefiner.interp_sparse_observations (:327-368, interpolator :351)
  unet_forward_batch   self.model({"image": ...}) (pixtrack/localization/feature_extractor.py:48)
  ngp_render[_both]    testbed.render(w, h, spp, True) (pixtrack/visualization/run_vis_on_poses.py:51)
  depth_mask           get_mask morphology (pixtrack/pose_trackers/pixloc_tracker_r9.py:207-214)
  rgba_to_u8           get_nerf_image's alpha threshold / *255 / uint8 (run_vis_on_poses.py:52-54)
  resize_linear        pixloc resize(image, size, max, "linear") (feature_extractor.py:45)

All ops are out-variants (they write into tensors the caller allocated and mutate nothing else), so
the caller decides buffer reuse.  Context handles (``pxt_unet*`` / ``pxt_ngp*``) travel as ints.
There is no CPU implementation: the ops are registered for the CUDA(HIP) key only, so a call with
CPU tensors fails in the dispatcher, and a missing library raises ``PxtError`` at first use.
"""
from __future__ import annotations

import ctypes as C
from typing import List, Optional, Sequence

import torch

from . import _lib

_NS = "pixtrack"
_DEF = torch.library.Library(_NS, "DEF")

SCHEMAS = {
    "lm_refine": (
        "(Tensor p3d, Tensor? point_mask, Tensor[] fmaps, Tensor[] frefs, int[] channels, float[] cameras, "
        "int[] ndist, float[] lambdas, float[] T_init, int num_iters, int pad, int loss, float loss_alpha, "
        "float loss_scale, float grad_stop, float dt_stop, float dR_stop, int min_valid, int n_workgroups, "
        "Tensor(a!) record, Tensor(b!) workspace, bool want_log, int spin_limit=0, float[]? cam_conv=None, "
        "int[]? cam_slots=None, Tensor(c!)? cam_out=None, int lm_path=0) -> ()"),
    # K problems in one persistent launch (pxt_lm_refine_batch): per-problem tensors as lists, per-level tensors and
    # records flattened problem-major (n_levels[k] entries each), conf shared; n_workgroups is per problem
    "lm_refine_batch": (
        "(Tensor[] p3d, Tensor?[] point_masks, int[] n_levels, Tensor[] fmaps, Tensor[] frefs, int[] channels, "
        "float[] cameras, int[] ndist, float[] lambdas, float[] T_init, int num_iters, int pad, int loss, float loss_alpha, "
        "float loss_scale, float grad_stop, float dt_stop, float dR_stop, int min_valid, int n_workgroups, "
        "Tensor(a!)[] records, Tensor(b!)[] workspaces, Tensor(c!) batch_workspace, bool want_log, int spin_limit=0, "
        "float[]? cam_conv=None, int[]? cam_slots=None, Tensor(d!)[]? cam_outs=None, int lm_path=0, "
        "int[]? cam_enabled=None) -> ()"),
    "sample_sparse": (
        "(Tensor p3d, float[] T, Tensor[] fmaps, int[] channels, float[] cameras, int[] ndist, int pad, "
        "bool normalize, Tensor(a!)[] outs, Tensor(b!) valid, int[]? windows=None) -> ()"),
    "unet_forward_batch": (
        "(int ctx, Tensor[] images, Tensor?[] masks, bool[] normalize, Tensor(a!)[] outs, Tensor(b!) workspace) -> ()"),
    "ngp_render": (
        "(int ctx, float[] view, int width, int height, int spp, int mode, Tensor(a!) out, Tensor(b!)? stats) -> ()"),
    "ngp_render_both": (
        "(int ctx, float[] view, int width, int height, int spp, Tensor(a!) rgba, Tensor(b!) depth, "
        "Tensor(c!)? stats) -> ()"),
    "ngp_render_both_from_pose": (
        "(int ctx, float[] view, Tensor pose_record, float[] conv, int width, int height, int spp, int mode, "
        "Tensor(a!) rgba, Tensor(b!)? depth, Tensor(c!) cam_out, Tensor(d!)? stats) -> ()"),
    "ngp_render_frame": (
        "(int ctx, float[] view, int width, int height, int spp, int mode, bool camera_from_slot, Tensor(a!)? rgba, "
        "Tensor(b!)? depth, Tensor(c!)? rgb_u8, Tensor(d!)? depth_nz, Tensor(e!)? stats) -> ()"),
    # K renders of K different contexts as one chain of launches (pxt_ngp_render_frame_batch): views = K x 25 floats, sizes =
    # K x (width, height), modes = K x {0 Shade, 1 Depth, 2 both}; rgb_u8: one tensor per render whose mode is 0 / 2, in render
    # order; depth_nz: one per render whose mode is 1 / 2
    "ngp_render_frame_batch": (
        "(int[] ctxs, float[] views, int[] sizes, int spp, int[] modes, bool camera_from_slot, Tensor(a!)[] rgb_u8, "
        "Tensor(b!)[] depth_nz, Tensor(c!) workspace, Tensor(d!)[] stats) -> ()"),
    "depth_mask": "(Tensor depth_rgba, int n_erode, int n_dilate, Tensor(a!) mask, Tensor(b!) scratch) -> ()",
    "depth_mask_plane": "(Tensor depth_nz, int n_erode, int n_dilate, Tensor(a!) mask) -> ()",
    "rgba_to_u8": "(Tensor rgba, float alpha_thresh, Tensor(a!) out) -> ()",
    "resize_linear": "(Tensor src, Tensor(a!) dst) -> ()",
    "resize_activity": "(Tensor? mask, Tensor? image_u8, int H, int W, Tensor(a!) active) -> ()",
    "conv3x3_nhwc_f16": "(Tensor x, Tensor weight, Tensor bias, bool relu, Tensor(a!) out) -> ()",
}
for _name, _schema in SCHEMAS.items():
    _DEF.define(_name + _schema)

VIEW_FLOATS = 12 + 2 + 3 + 3 + 4 + 1  # cam 3x4, focal, k1, aabb_min, aabb_max, background, min_transmittance


def _stream(t: torch.Tensor) -> int:
    return int(torch.cuda.current_stream(t.device).cuda_stream)


def _f32c(t: torch.Tensor, what: str) -> torch.Tensor:
    if t.dtype != torch.float32 or not t.is_contiguous():
        raise _lib.PxtError(f"{what} must be a contiguous float32 tensor (got {t.dtype}, contiguous={t.is_contiguous()})")
    return t


# ------------------------------------------------------------------------------------------- LM
def _lm_refine(p3d, point_mask, fmaps, frefs, channels, cameras, ndist, lambdas, T_init, num_iters, pad, loss,
               loss_alpha, loss_scale, grad_stop, dt_stop, dR_stop, min_valid, n_workgroups, record, workspace,
               want_log, spin_limit=0, cam_conv=None, cam_slots=None, cam_out=None, lm_path=0):
    L = _lib.lib()
    n_levels = len(fmaps)
    if not (1 <= n_levels <= _lib.PXT_MAX_LEVELS) or len(frefs) != n_levels or len(channels) != n_levels:
        raise _lib.PxtError(f"lm_refine: {n_levels} levels (1..{_lib.PXT_MAX_LEVELS}), {len(frefs)} frefs, {len(channels)} channels")
    if len(cameras) != 10 * n_levels or len(lambdas) != 6 * n_levels or len(ndist) != n_levels or len(T_init) != 12:
        raise _lib.PxtError("lm_refine: cameras need 10, lambdas 6 floats per level, T_init 12 floats")
    _f32c(p3d, "p3d")
    n = int(p3d.shape[0])
    arr = (_lib.LmLevel * n_levels)()
    for i in range(n_levels):
        fm, fr = _f32c(fmaps[i], "fmap"), _f32c(frefs[i], "fref")
        h, w, cs = fm.shape
        if tuple(fr.shape) != (n, cs):
            raise _lib.PxtError(f"lm_refine: fref of level {i} is {tuple(fr.shape)}, expected {(n, cs)}")
        arr[i].fmap, arr[i].fref = fm.data_ptr(), fr.data_ptr()
        arr[i].h, arr[i].w, arr[i].C, arr[i].cstride = h, w, int(channels[i]), cs
        arr[i].cam[:] = cameras[10 * i:10 * i + 10]
        arr[i].ndist = int(ndist[i])
        arr[i].lambda_[:] = lambdas[6 * i:6 * i + 6]
    conf = _lib.LmConf()
    conf.num_iters, conf.pad, conf.loss = int(num_iters), int(pad), int(loss)
    conf.loss_alpha, conf.loss_scale = float(loss_alpha), float(loss_scale)
    conf.grad_stop, conf.dt_stop, conf.dR_stop = float(grad_stop), float(dt_stop), float(dR_stop)
    conf.min_valid, conf.n_workgroups, conf.spin_limit = int(min_valid), int(n_workgroups), int(spin_limit)
    conf.path = int(lm_path)
    nh = 16 + _lib.PXT_MAX_LEVELS
    need = nh + (n_levels * int(num_iters) * _lib.PXT_LM_LOG_STRIDE if want_log else 0)
    if record.dtype != torch.float32 or record.numel() < need or not record.is_contiguous():
        raise _lib.PxtError(f"lm_refine: record needs {need} contiguous float32 values")
    if not (record.is_cuda or record.is_pinned()):
        raise _lib.PxtError("lm_refine: record must be device memory or pinned host memory")
    if point_mask is not None and (point_mask.dtype != torch.uint8 or not point_mask.is_contiguous()):
        raise _lib.PxtError("lm_refine: point_mask must be contiguous uint8")
    base = record.data_ptr()
    T0 = (C.c_float * 12)(*[float(x) for x in T_init])
    if cam_conv is not None:  # the kernel's epilogue also derives the next render's camera (pxt_lm_refine_cam)
        if len(cam_conv) != 27:
            raise _lib.PxtError("lm_refine: cam_conv holds 27 doubles (Testbed.pose_conversion)")
        slots = [int(x) for x in (cam_slots or [])]
        if len(slots) > 2 or (not slots and cam_out is None):
            raise _lib.PxtError("lm_refine: at most two camera slots; at least one slot or cam_out")
        cam = _lib.LmCamera()
        cam.conv27[:] = [float(x) for x in cam_conv]
        for k in range(2):
            cam.cam_slot[k] = slots[k] if k < len(slots) and slots[k] else None
        cam.cam_out13 = None
        if cam_out is not None:
            if cam_out.dtype != torch.float32 or cam_out.numel() < 13 or not (cam_out.is_cuda or cam_out.is_pinned()):
                raise _lib.PxtError("lm_refine: cam_out must be a pinned host (or device) float32 tensor of >= 13 elements")
            cam.cam_out13 = cam_out.data_ptr()
        _lib.check(
            L.pxt_lm_refine_cam(p3d.data_ptr(), _lib.dptr(point_mask), n, arr, n_levels, T0, C.byref(conf), base,
                                base + 4 * nh if want_log else None, workspace.data_ptr(), C.byref(cam), _stream(p3d)),
            "pxt_lm_refine_cam")
        return
    _lib.check(
        L.pxt_lm_refine(p3d.data_ptr(), _lib.dptr(point_mask), n, arr, n_levels, T0, C.byref(conf), base,
                        base + 4 * nh if want_log else None, workspace.data_ptr(), _stream(p3d)),
        "pxt_lm_refine")


def _lm_refine_batch(p3d, point_masks, n_levels, fmaps, frefs, channels, cameras, ndist, lambdas, T_init, num_iters, pad,
                     loss, loss_alpha, loss_scale, grad_stop, dt_stop, dR_stop, min_valid, n_workgroups, records,
                     workspaces, batch_workspace, want_log, spin_limit=0, cam_conv=None, cam_slots=None, cam_outs=None,
                     lm_path=0, cam_enabled=None):
    L = _lib.lib()
    K = len(p3d)
    n_levels = [int(x) for x in n_levels]
    nl_tot = sum(n_levels)
    if not (1 <= K <= _lib.PXT_LM_MAX_BATCH) or len(n_levels) != K or len(records) != K or len(workspaces) != K \
            or len(point_masks) != K or len(T_init) != 12 * K:
        raise _lib.PxtError(f"lm_refine_batch: {K} problems (1..{_lib.PXT_LM_MAX_BATCH}) need one record, workspace, mask slot "
                            "and 12 pose floats each")
    if len(fmaps) != nl_tot or len(frefs) != nl_tot or len(channels) != nl_tot or len(cameras) != 10 * nl_tot \
            or len(lambdas) != 6 * nl_tot or len(ndist) != nl_tot:
        raise _lib.PxtError("lm_refine_batch: per level one fmap, fref, channel count, ndist, 10 camera and 6 lambda floats")
    conf = _lib.LmConf()
    conf.num_iters, conf.pad, conf.loss = int(num_iters), int(pad), int(loss)
    conf.loss_alpha, conf.loss_scale = float(loss_alpha), float(loss_scale)
    conf.grad_stop, conf.dt_stop, conf.dR_stop = float(grad_stop), float(dt_stop), float(dR_stop)
    conf.min_valid, conf.n_workgroups, conf.spin_limit = int(min_valid), int(n_workgroups), int(spin_limit)
    conf.path = int(lm_path)
    nh = 16 + _lib.PXT_MAX_LEVELS
    if cam_conv is not None and (len(cam_conv) != 27 * K or len(cam_slots or []) != 2 * K
                                 or (cam_outs is not None and len(cam_outs) != K)):
        raise _lib.PxtError("lm_refine_batch: cam_conv holds 27 doubles, cam_slots 2 pointers (0 = none) per problem")
    probs = (_lib.LmProblem * K)()
    keep = []  # ctypes records the problem array points to
    li = 0
    for k in range(K):
        _f32c(p3d[k], "p3d")
        n = int(p3d[k].shape[0])
        nl = n_levels[k]
        if not (1 <= nl <= _lib.PXT_MAX_LEVELS):
            raise _lib.PxtError(f"lm_refine_batch: problem {k} has {nl} levels")
        arr = (_lib.LmLevel * nl)()
        for i in range(nl):
            fm, fr = _f32c(fmaps[li], "fmap"), _f32c(frefs[li], "fref")
            h, w, cs = fm.shape
            if tuple(fr.shape) != (n, cs):
                raise _lib.PxtError(f"lm_refine_batch: fref of problem {k} level {i} is {tuple(fr.shape)}, expected {(n, cs)}")
            arr[i].fmap, arr[i].fref = fm.data_ptr(), fr.data_ptr()
            arr[i].h, arr[i].w, arr[i].C, arr[i].cstride = h, w, int(channels[li]), cs
            arr[i].cam[:] = cameras[10 * li:10 * li + 10]
            arr[i].ndist = int(ndist[li])
            arr[i].lambda_[:] = lambdas[6 * li:6 * li + 6]
            li += 1
        rec = records[k]
        need = nh + (nl * int(num_iters) * _lib.PXT_LM_LOG_STRIDE if want_log else 0)
        if rec.dtype != torch.float32 or rec.numel() < need or not rec.is_contiguous() or not (rec.is_cuda or rec.is_pinned()):
            raise _lib.PxtError(f"lm_refine_batch: record {k} needs {need} contiguous float32 values in device or pinned memory")
        mk = point_masks[k]
        if mk is not None and (mk.dtype != torch.uint8 or not mk.is_contiguous() or mk.numel() != n):
            raise _lib.PxtError("lm_refine_batch: point masks are contiguous uint8 [n_points]")
        T0 = (C.c_float * 12)(*[float(x) for x in T_init[12 * k:12 * k + 12]])
        q = probs[k]
        q.p3d, q.point_mask, q.n_points = p3d[k].data_ptr(), _lib.dptr(mk), n
        q.levels_host, q.n_levels, q.T_init_host = arr, nl, T0
        q.out = rec.data_ptr()
        q.log = rec.data_ptr() + 4 * nh if want_log else None
        q.workspace = workspaces[k].data_ptr()
        cam = None
        if cam_conv is not None and (cam_enabled is None or cam_enabled[k]):
            cam = _lib.LmCamera()
            cam.conv27[:] = [float(x) for x in cam_conv[27 * k:27 * k + 27]]
            for j in range(2):
                cam.cam_slot[j] = int(cam_slots[2 * k + j]) or None
            cam.cam_out13 = None
            if cam_outs is not None:
                co = cam_outs[k]
                if co.dtype != torch.float32 or co.numel() < 13 or not (co.is_cuda or co.is_pinned()):
                    raise _lib.PxtError("lm_refine_batch: cam_outs are pinned host (or device) float32 tensors of >= 13 elements")
                cam.cam_out13 = co.data_ptr()
            q.cam_host = C.pointer(cam)
        keep.append((arr, T0, cam))
    if batch_workspace.numel() * batch_workspace.element_size() < int(L.pxt_lm_batch_workspace_bytes(K)):
        raise _lib.PxtError("lm_refine_batch: batch_workspace smaller than pxt_lm_batch_workspace_bytes(K)")
    _lib.check(L.pxt_lm_refine_batch(probs, K, C.byref(conf), batch_workspace.data_ptr(), _stream(p3d[0])),
               "pxt_lm_refine_batch")


# ------------------------------------------------------------------------------------ sampling
def _sample_sparse(p3d, T, fmaps, channels, cameras, ndist, pad, normalize, outs, valid, windows=None):
    L = _lib.lib()
    n_levels = len(fmaps)
    if len(outs) != n_levels or len(cameras) != 10 * n_levels or len(T) != 12:
        raise _lib.PxtError("sample_sparse: one out tensor and 10 camera floats per level, T of 12 floats")
    if windows is not None and len(windows) != 4 * n_levels:
        raise _lib.PxtError("sample_sparse: windows holds (x0, y0, full_w, full_h) per level")
    _f32c(p3d, "p3d")
    n = int(p3d.shape[0])
    arr = (_lib.SampleLevel * n_levels)()
    for i in range(n_levels):
        fm, out = _f32c(fmaps[i], "fmap"), _f32c(outs[i], "out")
        h, w, cs = fm.shape
        if tuple(out.shape) != (n, cs):
            raise _lib.PxtError(f"sample_sparse: out of level {i} is {tuple(out.shape)}, expected {(n, cs)}")
        arr[i].fmap, arr[i].out = fm.data_ptr(), out.data_ptr()
        arr[i].h, arr[i].w, arr[i].C, arr[i].cstride = h, w, int(channels[i]), cs
        arr[i].cam[:] = cameras[10 * i:10 * i + 10]
        arr[i].ndist = int(ndist[i])
        if windows is not None:  # (x0, y0, full_w, full_h) per level: the map is a window of the full level
            arr[i].x0, arr[i].y0, arr[i].full_w, arr[i].full_h = (int(x) for x in windows[4 * i:4 * i + 4])
    if valid.dtype != torch.uint8 or valid.numel() != n:
        raise _lib.PxtError("sample_sparse: valid must be uint8 [n_points]")
    T12 = (C.c_float * 12)(*[float(x) for x in T])
    _lib.check(L.pxt_sample_sparse(p3d.data_ptr(), n, T12, arr, n_levels, int(pad), int(bool(normalize)),
                                   valid.data_ptr(), _stream(p3d)), "pxt_sample_sparse")


# ---------------------------------------------------------------------------------------- UNet
def _unet_forward_batch(ctx, images, masks, normalize, outs, workspace):
    L = _lib.lib()
    n = len(images)
    if n == 0 or len(masks) != n or len(normalize) != n or len(outs) != 3 * n:
        raise _lib.PxtError("unet_forward_batch: per image one mask slot, one normalize flag and three output maps")
    sizes = [(int(im.shape[0]), int(im.shape[1])) for im in images]
    H, W = sizes[0]
    # two images of different sizes (a frame's reference render and its query with real assets): two single-image
    # passes side by side (pxt_unet_forward_pair); any other batch holds images of one size
    pair = n == 2 and sizes[1] != sizes[0]
    for im, mk, hw in zip(images, masks, sizes):
        if im.dim() != 3 or im.shape[2] != 3 or not im.is_contiguous() or im.dtype not in (torch.float32, torch.uint8):
            raise _lib.PxtError("unet_forward_batch: images are contiguous HWC, 3 channels, float32 or uint8")
        if hw != (H, W) and not pair:
            raise _lib.PxtError("unet_forward_batch: a batch of more than two holds images of one size")
        if mk is not None and (mk.dtype != torch.uint8 or tuple(mk.shape) != hw or not mk.is_contiguous()):
            raise _lib.PxtError("unet_forward_batch: masks are contiguous uint8 [H, W]")
    Hs = (C.c_int32 * 2)(sizes[0][0], sizes[-1][0])
    Ws = (C.c_int32 * 2)(sizes[0][1], sizes[-1][1])
    need = int(L.pxt_unet_workspace_bytes_pair(ctx, Hs, Ws)) if pair else int(L.pxt_unet_workspace_bytes_batch(ctx, n, H, W))
    if need <= 0:
        raise _lib.PxtError(f"images {sizes} are not supported by the 4-level encoder")
    if workspace.numel() * workspace.element_size() < need:
        raise _lib.PxtError(f"unet_forward_batch: workspace holds {workspace.numel()} bytes, {need} needed")
    for o in outs:
        _f32c(o, "out map")
    imgs = (C.c_void_p * n)(*[im.data_ptr() for im in images])
    is_u8 = (C.c_int32 * n)(*[int(im.dtype == torch.uint8) for im in images])
    mks = (C.c_void_p * n)(*[_lib.dptr(m) for m in masks])
    norm = (C.c_int32 * n)(*[int(bool(x)) for x in normalize])
    ptrs = (C.c_void_p * (3 * n))(*[o.data_ptr() for o in outs])
    cs = (C.c_int32 * 3)(*[int(o.shape[2]) for o in outs[:3]])
    if pair:
        _lib.check(L.pxt_unet_forward_pair(ctx, imgs, is_u8, mks, Hs, Ws, ptrs, cs, norm, workspace.data_ptr(),
                                           _stream(images[0])), "pxt_unet_forward_pair")
    else:
        _lib.check(L.pxt_unet_forward_batch(ctx, n, imgs, is_u8, mks, H, W, ptrs, cs, norm, workspace.data_ptr(),
                                            _stream(images[0])), "pxt_unet_forward_batch")


def _conv3x3(x, weight, bias, relu, out):
    H, W, Cin = (int(s) for s in x.shape)
    Cout = int(weight.shape[0])
    _lib.check(_lib.lib().pxt_conv3x3_nhwc_f16(x.data_ptr(), H, W, Cin, weight.data_ptr(), bias.data_ptr(), Cout,
                                               int(bool(relu)), out.data_ptr(), _stream(x)), "pxt_conv3x3_nhwc_f16")


# ---------------------------------------------------------------------------------------- NeRF
def _view(view: Sequence[float], width, height, spp, mode) -> "_lib.NgpView":
    if len(view) != VIEW_FLOATS:
        raise _lib.PxtError(f"ngp view record has {len(view)} floats, expected {VIEW_FLOATS}")
    v = _lib.NgpView()
    v.cam[:] = view[0:12]
    v.focal, v.k1 = view[12], view[13]
    v.aabb_min[:] = view[14:17]
    v.aabb_max[:] = view[17:20]
    v.background[:] = view[20:24]
    v.min_transmittance = view[24]
    v.width, v.height, v.spp, v.mode = int(width), int(height), int(spp), int(mode)
    return v


def _check_frame(t: torch.Tensor, width, height, what):
    _f32c(t, what)
    if tuple(t.shape) != (height, width, 4):
        raise _lib.PxtError(f"{what} must be [{height}, {width}, 4] (got {tuple(t.shape)})")


def _ngp_render(ctx, view, width, height, spp, mode, out, stats):
    _check_frame(out, width, height, "out")
    v = _view(view, width, height, spp, mode)
    _lib.check(_lib.lib().pxt_ngp_render(ctx, C.byref(v), out.data_ptr(), _lib.dptr(stats), _stream(out)),
               "pxt_ngp_render")


def _ngp_render_both(ctx, view, width, height, spp, rgba, depth, stats):
    _check_frame(rgba, width, height, "rgba")
    _check_frame(depth, width, height, "depth")
    v = _view(view, width, height, spp, 0)
    _lib.check(_lib.lib().pxt_ngp_render_both(ctx, C.byref(v), rgba.data_ptr(), depth.data_ptr(), _lib.dptr(stats),
                                              _stream(rgba)), "pxt_ngp_render_both")


def _ngp_render_both_from_pose(ctx, view, pose_record, conv, width, height, spp, mode, rgba, depth, cam_out, stats):
    """pxt_ngp_render_both with the camera derived ON THE DEVICE from `pose_record` (pinned float32, >= 12 floats:
    the record pxt_lm_refine writes), so that the render can be enqueued behind the LM launch.  `cam_out`: pinned
    float32 [>= 13], receives the camera + a completion word.  `depth` given: Shade + Depth in one march (`mode`
    ignored); None: one render in `mode` (0 Shade, 1 Depth) into `rgba`."""
    _check_frame(rgba, width, height, "rgba")
    if depth is not None:
        _check_frame(depth, width, height, "depth")
    for t, n, what in ((pose_record, 12, "pose_record"), (cam_out, 13, "cam_out")):
        if t.dtype != torch.float32 or t.numel() < n or t.is_cuda or not t.is_pinned():
            raise _lib.PxtError(f"{what} must be a pinned host float32 tensor of >= {n} elements")
    if len(conv) != 27:
        raise _lib.PxtError("conv holds 27 doubles (nerf2sfm centroid, 3/avglen, R, totp, snapshot scale, offset)")
    v = _view(view, width, height, spp, int(mode) if depth is None else 0)
    cv = (C.c_double * 27)(*[float(x) for x in conv])
    _lib.check(_lib.lib().pxt_ngp_render_both_from_pose(ctx, C.byref(v), pose_record.data_ptr(), cv, cam_out.data_ptr(),
                                                        rgba.data_ptr(), _lib.dptr(depth), _lib.dptr(stats),
                                                        _stream(rgba)), "pxt_ngp_render_both_from_pose")


def _ngp_render_frame(ctx, view, width, height, spp, mode, camera_from_slot, rgba, depth, rgb_u8, depth_nz, stats):
    """pxt_ngp_render_frame: one render (mode 0 Shade / 1 Depth / 2 both in one march) whose last kernel also writes the
    8-bit planes the tracker consumes - `rgb_u8` uint8 [H, W, 3] (get_nerf_image's image), `depth_nz` uint8 [H, W]
    (get_mask's `uint8(depth * 255) != 0`) - so that the float images (`rgba`, `depth`) are optional; with
    `camera_from_slot` the camera comes from the context's slot (written by lm_refine's epilogue: cam_slots)."""
    width, height, mode = int(width), int(height), int(mode)
    if mode not in (0, 1, 2):
        raise _lib.PxtError("ngp_render_frame: mode is 0 (Shade), 1 (Depth) or 2 (both)")
    for t, what in ((rgba, "rgba"), (depth, "depth")):
        if t is not None:
            _check_frame(t, width, height, what)
    if rgb_u8 is not None and (rgb_u8.dtype != torch.uint8 or tuple(rgb_u8.shape) != (height, width, 3)
                               or not rgb_u8.is_contiguous() or not rgb_u8.is_cuda):
        raise _lib.PxtError("ngp_render_frame: rgb_u8 is a contiguous device uint8 [H, W, 3]")
    if depth_nz is not None and (depth_nz.dtype != torch.uint8 or tuple(depth_nz.shape) != (height, width)
                                 or not depth_nz.is_contiguous() or not depth_nz.is_cuda):
        raise _lib.PxtError("ngp_render_frame: depth_nz is a contiguous device uint8 [H, W]")
    ref = next((t for t in (rgba, depth, rgb_u8, depth_nz) if t is not None), None)
    if ref is None:
        raise _lib.PxtError("ngp_render_frame: no output")
    o = _lib.NgpOutputs(_lib.dptr(rgba), _lib.dptr(depth), _lib.dptr(rgb_u8), _lib.dptr(depth_nz))
    v = _view(view, width, height, spp, 0 if mode == 2 else mode)
    _lib.check(_lib.lib().pxt_ngp_render_frame(ctx, C.byref(v), mode, int(bool(camera_from_slot)), C.byref(o),
                                               _lib.dptr(stats), _stream(ref)), "pxt_ngp_render_frame")


def _ngp_render_frame_batch(ctxs, views, sizes, spp, modes, camera_from_slot, rgb_u8, depth_nz, workspace, stats):
    """pxt_ngp_render_frame_batch: K renders of K renderer contexts - a frame's Depth + Shade pair, or K objects in lock-step -
    as ONE staged chain of launches; bit for bit K calls of ngp_render_frame.  8-bit outputs only (what the tracker consumes)."""
    K, modes = len(ctxs), [int(m) for m in modes]
    if not (1 <= K <= _lib.PXT_NGP_MAX_BATCH) or len(views) != K * VIEW_FLOATS or len(sizes) != 2 * K or len(modes) != K:
        raise _lib.PxtError(f"ngp_render_frame_batch: {K} renders (1..{_lib.PXT_NGP_MAX_BATCH}) need {VIEW_FLOATS} view floats, "
                            "(width, height) and a mode each")
    if any(m not in (0, 1, 2) for m in modes):
        raise _lib.PxtError("ngp_render_frame_batch: a mode is 0 (Shade), 1 (Depth) or 2 (both)")
    if len(rgb_u8) != sum(m != 1 for m in modes) or len(depth_nz) != sum(m != 0 for m in modes):
        raise _lib.PxtError("ngp_render_frame_batch: one rgb_u8 per render of mode 0 / 2, one depth_nz per render of mode 1 / 2")
    L = _lib.lib()
    need = int(L.pxt_ngp_batch_workspace_bytes(K)) if K > 2 else 0  # (<= 2 renders: the records travel as kernel arguments)
    if workspace.dtype != torch.uint8 or not workspace.is_cuda or not workspace.is_contiguous() or workspace.numel() < need:
        raise _lib.PxtError(f"ngp_render_frame_batch: workspace is a contiguous device uint8 tensor of >= {need} bytes")
    if len(stats) not in (0, K) or any(t.dtype != torch.int64 or t.numel() < 4 or not t.is_contiguous() for t in stats):
        raise _lib.PxtError("ngp_render_frame_batch: stats is empty or one contiguous int64 [4] per render")
    vs = (_lib.NgpView * K)()
    outs = (_lib.NgpOutputs * K)()
    cp = (C.c_void_p * K)(*[int(c) for c in ctxs])
    mp = (C.c_int32 * K)(*modes)
    sp = (C.c_void_p * K)()
    i_u8 = i_nz = 0
    for k in range(K):
        w, h = int(sizes[2 * k]), int(sizes[2 * k + 1])
        vs[k] = _view(views[k * VIEW_FLOATS:(k + 1) * VIEW_FLOATS], w, h, spp, 0 if modes[k] == 2 else modes[k])
        u8 = nz = None
        if modes[k] != 1:
            u8, i_u8 = rgb_u8[i_u8], i_u8 + 1
        if modes[k] != 0:
            nz, i_nz = depth_nz[i_nz], i_nz + 1
        if u8 is not None and (u8.dtype != torch.uint8 or tuple(u8.shape) != (h, w, 3) or not u8.is_contiguous()
                               or u8.device != workspace.device):
            raise _lib.PxtError("ngp_render_frame_batch: rgb_u8 is a contiguous device uint8 [H, W, 3]")
        if nz is not None and (nz.dtype != torch.uint8 or tuple(nz.shape) != (h, w) or not nz.is_contiguous()
                               or nz.device != workspace.device):
            raise _lib.PxtError("ngp_render_frame_batch: depth_nz is a contiguous device uint8 [H, W]")
        outs[k] = _lib.NgpOutputs(None, None, _lib.dptr(u8), _lib.dptr(nz))
        sp[k] = stats[k].data_ptr() if stats else None
    _lib.check(L.pxt_ngp_render_frame_batch(cp, vs, K, mp, int(bool(camera_from_slot)), outs,
                                            sp if stats else None, workspace.data_ptr() if workspace.numel() else None,
                                            _stream(workspace)),
               "pxt_ngp_render_frame_batch")


# ----------------------------------------------------------------------------------- image ops
def _depth_mask_plane(depth_nz, n_erode, n_dilate, mask):
    if depth_nz.dtype != torch.uint8 or depth_nz.dim() != 2 or not depth_nz.is_contiguous():
        raise _lib.PxtError("depth_mask_plane: depth_nz is a contiguous uint8 [H, W]")
    H, W = int(depth_nz.shape[0]), int(depth_nz.shape[1])
    if mask.dtype != torch.uint8 or tuple(mask.shape) != (H, W):
        raise _lib.PxtError("depth_mask_plane: mask is uint8 [H, W]")
    tmp = None
    if 2 * (int(n_erode) + int(n_dilate)) > 16:
        tmp = torch.empty(2 * H * W, dtype=torch.uint8, device=depth_nz.device)
    _lib.check(_lib.lib().pxt_depth_mask_plane(depth_nz.data_ptr(), H, W, int(n_erode), int(n_dilate), mask.data_ptr(),
                                               _lib.dptr(tmp), _stream(depth_nz)), "pxt_depth_mask_plane")



def _depth_mask(depth_rgba, n_erode, n_dilate, mask, scratch):
    _f32c(depth_rgba, "depth_rgba")
    H, W = int(depth_rgba.shape[0]), int(depth_rgba.shape[1])
    if mask.dtype != torch.uint8 or tuple(mask.shape) != (H, W) or scratch.numel() < 2 * H * W:
        raise _lib.PxtError("depth_mask: mask is uint8 [H, W], scratch holds 2*H*W bytes")
    _lib.check(_lib.lib().pxt_depth_mask(depth_rgba.data_ptr(), H, W, int(n_erode), int(n_dilate), mask.data_ptr(),
                                         scratch.data_ptr(), _stream(depth_rgba)), "pxt_depth_mask")


def _rgba_to_u8(rgba, alpha_thresh, out):
    _f32c(rgba, "rgba")
    H, W = int(rgba.shape[0]), int(rgba.shape[1])
    if out.dtype != torch.uint8 or tuple(out.shape) != (H, W, 3):
        raise _lib.PxtError("rgba_to_u8: out is uint8 [H, W, 3]")
    _lib.check(_lib.lib().pxt_rgba_to_u8(rgba.data_ptr(), H, W, float(alpha_thresh), out.data_ptr(), _stream(rgba)),
               "pxt_rgba_to_u8")


def _resize_linear(src, dst):
    _f32c(src, "src")
    _f32c(dst, "dst")
    H, W, Cc = (int(s) for s in src.shape)
    Ho, Wo, Co = (int(s) for s in dst.shape)
    if Co != Cc:
        raise _lib.PxtError("resize_linear: channel counts differ")
    _lib.check(_lib.lib().pxt_resize_linear(src.data_ptr(), H, W, Cc, dst.data_ptr(), Ho, Wo, _stream(src)),
               "pxt_resize_linear")


def _resize_activity(mask, image_u8, H, W, active):
    if (mask is None) == (image_u8 is None):
        raise _lib.PxtError("resize_activity: exactly one of mask / image_u8")
    src = mask if mask is not None else image_u8
    if src.dtype != torch.uint8 or not src.is_contiguous() or active.dtype != torch.uint8 or not active.is_contiguous():
        raise _lib.PxtError("resize_activity: contiguous uint8 tensors")
    Ho, Wo = int(active.shape[0]), int(active.shape[1])
    _lib.check(_lib.lib().pxt_resize_activity(_lib.dptr(mask), _lib.dptr(image_u8), int(H), int(W), Ho, Wo, active.data_ptr(),
                                              _stream(active)), "pxt_resize_activity")


_IMPLS = {
    "lm_refine": _lm_refine,
    "lm_refine_batch": _lm_refine_batch,
    "sample_sparse": _sample_sparse,
    "unet_forward_batch": _unet_forward_batch,
    "conv3x3_nhwc_f16": _conv3x3,
    "ngp_render": _ngp_render,
    "ngp_render_both": _ngp_render_both,
    "ngp_render_both_from_pose": _ngp_render_both_from_pose,
    "ngp_render_frame": _ngp_render_frame,
    "ngp_render_frame_batch": _ngp_render_frame_batch,
    "depth_mask": _depth_mask,
    "depth_mask_plane": _depth_mask_plane,
    "rgba_to_u8": _rgba_to_u8,
    "resize_linear": _resize_linear,
    "resize_activity": _resize_activity,
}
for _name, _fn in _IMPLS.items():
    _DEF.impl(_name, _fn, "CUDA")  # CUDA dispatch key == HIP device on torch-ROCm; nothing for CPU

ops = getattr(torch.ops, _NS)


def op_names() -> List[str]:
    return sorted(SCHEMAS)
