"""CPU ORACLE (test infrastructure, NOT product code) for ONE tracked frame: the policy of
PixLocPoseTrackerR9.refine (reference pixtrack/pose_trackers/pixloc_tracker_r9.py:216-275)
composed from the three stage oracles -- NeRF reference/depth renders (ngp_oracle), UNet
pyramids (unet_oracle), sparse sampling + LM (lm_oracle).

PARITY UNPINNED (see the stage oracles' headers).  This is "the reference PyTorch-CPU
path" that BASELINE.json asks to be timed next to the GPU path; bench.py times it on a
bounded sample, the `-m gpu` frame test runs it at a small resolution.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this.
"""
from __future__ import annotations

import math
import time
from typing import Dict, Optional

import numpy as np
import torch

from . import lm_oracle as LO
from . import ngp_oracle as NO
from . import unet_oracle as UO

FLIP_YZ = np.diag([1.0, -1.0, -1.0, 1.0])


def sfm_to_nerf_pose(nerf2sfm, cIw):
    """reference pixtrack/utils/ingp_utils.py:47-63."""
    p = np.asarray(cIw, np.float64) @ FLIP_YZ
    p = p[[1, 0, 2, 3], :]
    p[2, :] *= -1
    p[0:3, 3] -= nerf2sfm["centroid"]
    p[0:3, 3] *= 3.0 / nerf2sfm["avglen"]
    p = nerf2sfm["R"] @ p
    p[0:3, 3] -= nerf2sfm["totp"]
    return p


def colmap_camera_to_pix(cam) -> torch.Tensor:
    """pixloc Camera.from_colmap: SIMPLE_RADIAL -> (w,h,f,f,cx-.5,cy-.5,k1,0); OPENCV (the YCB iterator's
    camera, reference pixtrack/utils/io.py:60-66) -> (w,h,fx,fy,cx-.5,cy-.5,k1,k2,p1,p2), missing
    distortion parameters read as zeros."""
    p = [float(x) for x in cam["params"]]
    if cam.get("model", "SIMPLE_RADIAL") == "OPENCV":
        fx, fy, cx, cy = p[:4]
        dist = (p[4:8] + [0.0] * 4)[:4]
        return torch.tensor([cam["width"], cam["height"], fx, fy, cx - 0.5, cy - 0.5] + dist, dtype=torch.float32)
    f, cx, cy, k1 = p
    return torch.tensor([cam["width"], cam["height"], f, f, cx - 0.5, cy - 0.5, k1, 0.0], dtype=torch.float32)


def ngp_model(snapshot) -> NO.NgpModel:
    """The oracle's model of a snapshot, built once per snapshot object (ngp_oracle.render_parallel keeps its worker
    processes per model)."""
    m = getattr(snapshot, "_oracle_model", None)
    if m is None:
        m = snapshot._oracle_model = _ngp_model(snapshot)
    return m


def _ngp_model(snapshot) -> NO.NgpModel:
    return NO.NgpModel(grid=snapshot.grid, mlp=snapshot.mlp_dict(), occupancy=snapshot.occupancy,
                       cascades=snapshot.cascades, aabb_scale=snapshot.aabb_scale, cone_angle=snapshot.cone_angle,
                       depth_scale=1.0 / snapshot.scale, linear_colors=bool(getattr(snapshot, "linear_colors", False)))


def nerf_view(snapshot, nerf2sfm, aabb, R, t, cam: torch.Tensor, mode: int, spp: int = 8) -> NO.View:
    """get_nerf_image's camera set-up (run_vis_on_poses.py:28-47): fov from fx on x."""
    wIc = np.eye(4)
    wIc[:3, :3], wIc[:3, 3] = R, t
    nerf_pose = sfm_to_nerf_pose(nerf2sfm, np.linalg.inv(wIc))
    W, H = int(cam[0]), int(cam[1])
    fov = math.degrees(math.atan(W / (float(cam[2]) * 2)) * 2)
    focal = float(np.float32(0.5 * W / math.tan(0.5 * math.radians(fov))))
    return NO.View(cam=NO.nerf_matrix_to_ngp(nerf_pose, snapshot.scale, snapshot.offset), focal=focal, width=W,
                   height=H, spp=spp, k1=snapshot.k1, aabb_min=tuple(aabb[0]), aabb_max=tuple(aabb[1]), mode=mode)


def to_u8(rgba: np.ndarray) -> np.ndarray:
    """run_vis_on_poses.py:52-54 with alpha_thresh 0: (rgb * 255).astype(uint8) (wraps)."""
    return ((rgba[..., :3] * np.float32(255.0)).astype(np.int64) & 255).astype(np.uint8)


def morph5(img: np.ndarray, erode: bool) -> np.ndarray:
    H, W = img.shape
    pad = np.full((H + 4, W + 4), 255 if erode else 0, dtype=np.uint8)
    pad[2:-2, 2:-2] = img
    stack = [pad[dy:dy + H, dx:dx + W] for dy in range(5) for dx in range(5)]
    return (np.min if erode else np.max)(np.stack(stack, 0), axis=0)


def depth_mask(depth_rgba: np.ndarray) -> np.ndarray:
    """get_mask (pixloc_tracker_r9.py:207-214): != 0, erode 5x5 x1, dilate 5x5 x5."""
    m = (to_u8(depth_rgba)[..., 0] != 0).astype(np.uint8)
    m = morph5(m, True)
    for _ in range(5):
        m = morph5(m, False)
    return m


def fragile_depth_pixels(depth_rgba: np.ndarray, margin: float = 0.05) -> np.ndarray:
    """Pixels whose `uint8(depth * 255) != 0` decision (get_mask; the uint8 cast wraps mod 256,
    run_vis_on_poses.py:53-54) lies within ``margin`` grey levels of flipping: v = depth * 255 near 1 from
    either side, or near a multiple of 256.  Exact zeros (rays that miss) and tiny values cannot flip.
    Only at these pixels may fp32 summation order inside the MLPs change a mask bit."""
    v = depth_rgba[..., 0].astype(np.float64) * 255.0
    m = np.mod(v, 256.0)
    return (np.abs(v - 1.0) < margin) | ((v > 128.0) & ((m < margin) | (m > 256.0 - margin) | (np.abs(m - 1.0) < margin)))


def track_frame(assets: Dict, R: np.ndarray, t: np.ndarray, query_image: np.ndarray, ref_id: int,
                multiscale=(1,), use_mask: bool = True, lm_conf: Optional[LO.LMConf] = None,
                timings: Optional[Dict[str, float]] = None, spp: int = 8, keep: Optional[Dict] = None,
                reference_scale: float = 0.5, aabb=None, query_camera=None):
    """One frame from pose (R, t): returns dict(success, R, t, cost, mask).  ``query_image``
    float32 HWC 0..255.  Follows refine(): mask -> dynamic reference -> per scale
    {UNet(ref) -> sparse sample, UNet(query) -> LM over 3 levels coarse->fine}."""
    tm = timings if timings is not None else {}

    def tick(name, t0):
        tm[name] = tm.get(name, 0.0) + (time.perf_counter() - t0)

    lm_conf = lm_conf or LO.LMConf()
    model3d, snapshot, nerf2sfm, weights = assets["model3d"], assets["snapshot"], assets["nerf2sfm"], assets["weights"]
    aabb = assets["aabb"] if aabb is None else aabb  # (the YCB policy renders inside the SfM points' box)
    ngp = ngp_model(snapshot)
    qcam = colmap_camera_to_pix(assets["query_camera"] if query_camera is None else query_camera)
    mask = None
    img = np.asarray(query_image, np.float32)
    if use_mask:
        t0 = time.perf_counter()
        depth = NO.render(ngp, nerf_view(snapshot, nerf2sfm, aabb, R, t, qcam, 1, spp))
        mask = depth_mask(depth)
        if keep is not None:
            keep["depth_rgba"] = depth
        img = img * mask[..., None].astype(np.float32)
        tick("nerf_depth", t0)
    # dynamic reference render with SfM camera 1 scaled by reference_scale (r9: 0.5, YCB: 0.3)
    c1 = model3d.cameras[1]
    ref_cam_full = colmap_camera_to_pix(dict(width=c1.width, height=c1.height, params=c1.params))
    ref_cam = LO.camera_scale(ref_cam_full, reference_scale)
    t0 = time.perf_counter()
    ref_rgba = NO.render(ngp, nerf_view(snapshot, nerf2sfm, aabb, R, t, ref_cam, 0, spp))
    ref_img = to_u8(ref_rgba).astype(np.float32)
    if keep is not None:
        keep["ref_rgba"] = ref_rgba
    tick("nerf_ref", t0)
    # points observed by the reference image with track length >= 3
    im = model3d.dbs[ref_id]
    ids = [int(p) for p in im.point3D_ids if p != -1 and len(model3d.points3D[int(p)].image_ids) >= 3]
    p3d = torch.from_numpy(np.array([model3d.points3D[p].xyz for p in ids], np.float32))
    lambdas = [LO.damping_lambda(weights[f"optimizer.{i}.dampingnet.const"].float()) for i in range(3)]
    Rt, tt = torch.from_numpy(np.asarray(R, np.float64)), torch.from_numpy(np.asarray(t, np.float64))
    R_ref, t_ref = Rt.float(), tt.float()  # the reference is rendered AND sampled at the frame's start pose
    log = LO.LMLog()
    ret = {"success": False}
    for scale in multiscale:
        t0 = time.perf_counter()
        f_ref, sc_ref, c_ref = UO.extractor_call(weights, ref_img, scale)
        maps_ref = [torch.cat([f, c], 0) for f, c in zip(f_ref, c_ref)]
        tick("unet_ref", t0)
        t0 = time.perf_counter()
        obs, valid = LO.interp_sparse_observations(maps_ref, sc_ref, ref_cam_full, reference_scale, R_ref, t_ref, p3d, lm_conf.pad)
        tick("sample", t0)
        t0 = time.perf_counter()
        f_q, sc_q, c_q = UO.extractor_call(weights, img, scale)
        maps_q = [torch.cat([f, c], 0) for f, c in zip(f_q, c_q)]
        tick("unet_query", t0)
        t0 = time.perf_counter()
        ret = LO.refine_pose_using_features(maps_q, sc_q, qcam, Rt, tt, obs, p3d, lambdas, lm_conf, mask=valid, log=log)
        tick("lm", t0)
        if not ret["success"]:
            break
        Rt, tt = ret["R"], ret["t"]
    cost = float(np.mean([c[-1] for c in log.costs if len(c)])) if log.costs else float("nan")
    return dict(success=ret["success"], R=ret.get("R"), t=ret.get("t"), cost=cost, mask=mask, iters=log.num_iters,
                n_points=len(ids), log=log)


def covisibility(model3d) -> Dict[int, Dict[int, int]]:
    """extract_covisibility (pixtrack/utils/hloc_utils.py:28-47): {image id: {other image id: number of shared 3-D
    points}}, the inner dictionaries in the order the shared points' tracks are met (nearest_reference's ties keep it)."""
    out = {}
    for image_id, image in model3d.dbs.items():
        counts: Dict[int, int] = {}
        for pid in image.point3D_ids:
            if pid == -1:
                continue
            for other in model3d.points3D[int(pid)].image_ids:
                if int(other) != image_id:
                    counts[int(other)] = counts.get(int(other), 0) + 1
        if counts:
            out[image_id] = counts
    return out


def nearest_reference(model3d, covis, curr_ref: int, R_qry: np.ndarray, n_shared: int = 50) -> int:
    """update_reference_ids (pixloc_tracker_r9.py:120-143): the current reference and its covisible images
    (> 50 shared points) ranked by the geodesic distance of their rotation to the current pose; K = 1."""
    def gdist(Ra, Rb):
        c = (np.trace(Ra @ Rb.T) - 1.0) / 2.0
        return float(np.arccos(np.clip(c, -1.0, 1.0)))

    cand = {curr_ref: gdist(R_qry, model3d.dbs[curr_ref].qvec2rotmat())}
    for k, v in covis[curr_ref].items():
        if v > n_shared:
            cand[k] = gdist(R_qry, model3d.dbs[k].qvec2rotmat())
    return sorted(cand, key=lambda r: cand[r])[0]


def track_sequence(assets: Dict, frames, spp: int = 8, covis=None, lm_conf: Optional[LO.LMConf] = None,
                   keep_masks: bool = True):
    """PixLocPoseTrackerR9.refine over CONSECUTIVE frames (pixloc_tracker_r9.py:216-275): what is carried from one
    frame to the next is the pose (updated only by an accepted frame, :258-265), the success flag (a failed frame
    makes the next one run UNMASKED at the scales last set, :218-225), the cost threshold (1.1 x the first frame's
    cost, frozen, :251-256) and the reference id (:120-143).  With ``covis`` (covisibility(model3d)) the reference id
    follows the reference's policy: every frame after the first renders a new dynamic reference (THRESH = 0, :171-203),
    extracts its sparse features at the 3-D points of the reference id IN FORCE (create_dynamic_reference_image,
    :153-158) and only then moves the id to the covisible image (> 50 shared points) whose rotation is nearest to the
    frame's start pose (update_reference_ids, :120-143) - so a switch takes effect one frame later; the refinement
    reads its points from the stored features (pixloc_pose_refiners.py:243-250).  Without ``covis`` the id stays the
    upright reference.  ``frames``: float32 HWC 0..255.  Returns one record per frame."""
    model3d = assets["model3d"]
    ref_id = model3d.name2id[assets["upright_ref_img"]]
    im = model3d.dbs[ref_id]
    R, t = im.qvec2rotmat(), np.asarray(im.tvec, np.float64)  # relocalize(): the upright reference pose (:95-106)
    cold, success, thr, multiscale = True, True, None, (1,)
    out = []
    for i, frame in enumerate(frames):
        use_mask = False
        if cold:
            multiscale, cold = (4, 1), False
        elif success:
            multiscale, use_mask = (1,), True
        keep = {}
        res = track_frame(assets, R, t, np.asarray(frame, np.float32), ref_id, multiscale=multiscale, use_mask=use_mask,
                          lm_conf=lm_conf, spp=spp, keep=keep)
        ref_used = ref_id
        if i > 0 and covis is not None:  # (after this frame's features were taken at the old id's points)
            ref_id = nearest_reference(model3d, covis, ref_id, np.asarray(R))
        cost = res["cost"]
        if thr is None:
            thr = cost + 0.1 * cost
        ok = bool(res["success"] and cost <= thr)
        rec = dict(frame=i, multiscale=tuple(multiscale), masked=use_mask, lm_success=bool(res["success"]), success=ok,
                   cost=cost, cost_threshold=thr, ref_id=ref_used, R_start=np.asarray(R, np.float64).copy(),
                   t_start=np.asarray(t, np.float64).copy(), iters=list(res["iters"]), n_points=res["n_points"],
                   R=res["R"].numpy() if res["R"] is not None else None,
                   t=res["t"].numpy() if res["t"] is not None else None,
                   mask=res["mask"] if keep_masks else None, depth_rgba=keep.get("depth_rgba"))
        out.append(rec)
        if ok:
            R, t = res["R"].numpy().astype(np.float64), res["t"].numpy().astype(np.float64)
        success = ok
    return out
