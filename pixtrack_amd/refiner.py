"""PoseTrackerLocalizer / PoseTrackerRefiner -- the owners of the drop-in boundary
(reference pixtrack/localization/pixloc_pose_refiners.py:28-396; pixloc BaseRefiner,
SURVEY.md Appendix A.4).

Same class names, method names, argument meaning and return dictionaries as the
reference; inside, the three device stages are the HIP kernels of this package:

* ``dense_feature_extraction``  -> pxt_unet_forward (feature_extractor.extract_packed)
* ``interp_sparse_observations`` -> pxt_sample_sparse (all levels, one launch)
* ``refine_pose_using_features`` -> pxt_lm_refine     (all levels, one persistent launch)

Per-point Python lists of the reference (``p3did_to_feat`` built by slicing N tensors per
level per frame, :359-366) are replaced by packed [N, cstride] tensors plus a validity
mask; the list views the reference exposes are materialised lazily on access only.
"""
from __future__ import annotations

import ctypes as C
import logging
from typing import Dict, List, Optional, Sequence, Tuple

import math

import numpy as np
import torch

from . import _lib
from .ops import ops
from .feature_extractor import PixTrackFeatureExtractor
from .geometry import Camera, Pose
from .model3d import Model3D
from .optimizer import LevelPack, PixTrackOptimizer, cstride_for
from .unet import OUTPUT_DIMS, UNet
from .utils.conf import Conf, merge

logger = logging.getLogger(__name__)
import os as _os

_WINDOW_ENV = _os.environ.get("PXT_REF_WINDOW", "1") != "0"  # A/B knob: 0 = the reference pass always on the whole render


class SparseReferenceFeatures(dict):
    """features_dicts[...]["features"][str(scale)] entry.

    Dict keys kept from the reference: "p3dids" (ids valid on all levels) and
    "p3did_to_feat" (per point tuple of per-level [C+1] tensors).  Both are derived lazily
    from the packed device buffers the kernels use: ``packed[l]`` [N, cstride] (descriptor
    L2-normalised, confidence at channel C), ``valid`` uint8 [N], ``p3dids_all``, ``p3d``."""

    def __init__(self, packed, valid, p3dids_all, p3d, dims):
        super().__init__()
        self.packed, self.valid, self.p3dids_all, self.p3d, self.dims = packed, valid, p3dids_all, p3d, dims

    def __missing__(self, key):
        if key == "p3dids":
            keep = self.valid.cpu().bool().tolist()
            self[key] = [p for p, k in zip(self.p3dids_all, keep) if k]
        elif key == "p3did_to_feat":
            keep = self.valid.cpu().bool()
            idx = torch.nonzero(keep)[:, 0].tolist()
            self[key] = [tuple(self.packed[l][i, : c + 1] for l, c in enumerate(self.dims)) for i in idx]
        else:
            raise KeyError(key)
        return self[key]


class PoseTrackerRefiner:
    base_default_config = dict(
        layer_indices=None,
        min_matches_db=10,
        num_dbs=1,
        min_track_length=3,
        min_points_opt=10,
        point_selection="all",
        average_observations=False,
        normalize_descriptors=True,
        compute_uncertainty=True,
    )
    default_config = dict(
        multiscale=None,
        # optional {image_scale: [pyramid levels]}: restricts the levels optimised at an image scale.
        # Not a reference feature: it exists for BASELINE configs[4]'s "4-level pyramid" stress
        # workload ({4: [2], 1: [2, 1, 0]}: the 1/64 map of the downscaled image + the three native ones)
        level_plan=None,
        reference_window=True,  # encode only the window of the reference render the sampled points depend on (see reference_window)
        filter_covisibility=False,
        do_pose_approximation=False,
        do_inlier_ranking=False,
    )
    tracker = None

    def __init__(self, device, optimizer: List[PixTrackOptimizer], model3d: Model3D,
                 feature_extractor: PixTrackFeatureExtractor, paths, conf, global_descriptors=None):
        self.global_descriptors = global_descriptors
        self.reference_scale = 1.0
        self.choices = {}
        self.features_dicts: Dict = {}
        self.device = torch.device(device)
        self.optimizer = optimizer
        self.model3d = model3d
        self.feature_extractor = feature_extractor
        self.paths = paths
        self.conf = merge(self.base_default_config, self.default_config, conf or {})
        assert self.conf.normalize_descriptors and self.conf.compute_uncertainty
        self.query_mask: Optional[torch.Tensor] = None  # device uint8 [H,W], set by the tracker
        self._p3d_cache: Dict[int, Tuple[List[int], torch.Tensor]] = {}
        self._p3d_host: Dict[int, np.ndarray] = {}
        self._ws = torch.zeros(int(_lib.lib().pxt_lm_workspace_bytes()), dtype=torch.uint8, device=self.device)
        self.last_lm = []  # LMResult per image scale of the last refine (costs for the tracker gate)

    # ---- logging hooks (pixloc BaseRefiner) -----------------------------------
    def log_dense(self, **kwargs):
        if self.tracker is not None:
            self.tracker.log_dense(**kwargs)

    def log_optim(self, **kwargs):
        if self.tracker is not None:
            self.tracker.log_optim_done(**kwargs)

    # ---- reference-image points ------------------------------------------------
    def _points_of(self, dbids: Sequence[int]) -> Tuple[List[int], torch.Tensor]:
        """3-D points observed by dbids[0] with a long enough track (static per ref id)."""
        key = int(dbids[0])
        if key not in self._p3d_cache:
            p3did_to_dbids = self.model3d.get_p3did_to_dbids(
                list(dbids), None, None, self.conf.point_selection, self.conf.min_track_length)
            dbid_to_p3dids = self.model3d.get_dbid_to_p3dids(p3did_to_dbids)
            p3dids = dbid_to_p3dids.get(dbids[0], [])
            xyz = np.array([self.model3d.points3D[p].xyz for p in p3dids], dtype=np.float32).reshape(-1, 3)
            self._p3d_cache[key] = (p3dids, torch.from_numpy(xyz).to(self.device))
            self._p3d_host[key] = xyz.astype(np.float64)
        return self._p3d_cache[key]

    # ---- the reference pass on a window of the reference render ---------------------------------------------
    # The reference computes the reference image's dense maps only to sample them at the projected 3-D points
    # (pixloc_pose_refiners.py:282-290, then `del features_ref_dense` :236,316).  A sample depends on the input
    # pixels within the pyramid's dependency radius of its 2 x 2 texels: 130 px for the stride-1 level (through the
    # 1/16 bottleneck and back up), 122 / 90 px for the other two (interval propagation through the 17 convolutions,
    # four pools and four bilinear up-samplings; scripts/unet_dependency_radius.py).  With real assets the reference
    # render is large and mostly background (921 x 921 for the YCB objects: SfM camera 3072 x 3072 x 0.3,
    # scripts/create_sfm_from_obj.py:154-159): the pass then runs on the window [bounding box of the projected points
    # +- WINDOW_MARGIN], aligned to the coarsest stride so that pooling and up-sampling pair the same pixels, and the
    # sampler reads it as a window of the full maps (pxt_sample_level.x0 ..).  Every sampled texel's whole dependency
    # cone lies inside the window, so it sees exactly the inputs it sees in the full pass; what differs is only which
    # tile configuration / split-K factor a layer of that size takes, i.e. fp32 summation order.
    WINDOW_MARGIN = 160   # >= 135 (dependency radius) + 1 (the bilinear sample's second texel) + 16 (alignment), a multiple of 16
    WINDOW_MIN_SAVING = 0.35  # a window is used when it drops at least this share of the image's pixels

    def reference_window(self, dbids, pose: Optional[Pose], reference_image):
        """-> (image to encode, window or None): window = (x0, y0, full_w, full_h) in image pixels."""
        memo = self.__dict__.get("_window_memo")
        if memo is not None and memo[0] is reference_image and memo[1] is pose:
            return memo[2], memo[3]
        out = (reference_image, None)
        if self.conf.get("reference_window", True) and _WINDOW_ENV and torch.is_tensor(reference_image) and pose is not None \
                and (self.conf.multiscale or [1]) == [1]:
            H, W = int(reference_image.shape[0]), int(reference_image.shape[1])
            ex = self.feature_extractor
            if ex.target_size(H, W, 1)[:2] == (H, W):  # (no resize on the way into the network)
                self._points_of(dbids)
                xyz = self._p3d_host[int(dbids[0])]
                image = self.model3d.dbs[dbids[0]]
                cam = self._reference_camera_of(image.camera_id)
                R, t = pose.numpy()
                p = xyz @ np.asarray(R, np.float64).T + np.asarray(t, np.float64)
                front = p[:, 2] > 1e-3
                if front.any():
                    c10 = [float(x) for x in cam.as10().tolist()]  # w, h, fx, fy, cx, cy, k1, k2, p1, p2
                    fx, fy, cx, cy, k1, k2, p1, p2 = c10[2:10]
                    xn, yn = p[front, 0] / p[front, 2], p[front, 1] / p[front, 2]
                    r2 = xn * xn + yn * yn  # the camera model of the sampler (pxt_common.h project_point), in float64
                    rad = k1 * r2 + k2 * r2 * r2
                    xd = xn + xn * rad + 2.0 * p1 * xn * yn + p2 * (r2 + 2.0 * xn * xn)
                    yd = yn + yn * rad + 2.0 * p2 * xn * yn + p1 * (r2 + 2.0 * yn * yn)
                    u, v = fx * xd + cx, fy * yd + cy
                    inside = (u >= 0) & (v >= 0) & (u <= W - 1) & (v <= H - 1)
                    if inside.any():
                        m = self.WINDOW_MARGIN
                        x0 = max(0, int(np.floor(u[inside].min()) - m) // 16 * 16)
                        y0 = max(0, int(np.floor(v[inside].min()) - m) // 16 * 16)
                        x1 = min(W, -(-(int(np.ceil(u[inside].max())) + 1 + m) // 16) * 16)
                        y1 = min(H, -(-(int(np.ceil(v[inside].max())) + 1 + m) // 16) * 16)
                        # (the crop itself must pass through the extractor unresized too: with resize_by = "max_force" a crop
                        # smaller than conf.resize would be force-upscaled and the window arithmetic would be off)
                        if (x1 - x0) * (y1 - y0) <= (1.0 - self.WINDOW_MIN_SAVING) * W * H and x1 - x0 >= 64 and y1 - y0 >= 64 \
                                and ex.target_size(y1 - y0, x1 - x0, 1)[:2] == (y1 - y0, x1 - x0):
                            out = (reference_image[y0:y1, x0:x1].contiguous(), (x0, y0, W, H))
        self._window_memo = (reference_image, pose, out[0], out[1])
        return out

    def _reference_camera_of(self, camera_id) -> Camera:
        ck = (camera_id, float(self.reference_scale))
        cams_memo = self.__dict__.setdefault("_ref_cameras", {})  # model cameras are static: build each once
        camera = cams_memo.get(ck)
        if camera is None:
            camera = cams_memo[ck] = Camera.from_colmap(self.model3d.cameras[camera_id]).scale(self.reference_scale)
        return camera

    def warm_reference_points(self, dbids: Optional[Sequence[int]] = None) -> None:
        """Builds the per-reference point tables (static data) up front, so that a switch of the
        reference image during tracking does not stall a frame on a Python loop over its points."""
        for dbid in (dbids if dbids is not None else list(self.model3d.dbs)):
            self._points_of([int(dbid)])

    # ---- dense features ----------------------------------------------------------
    def dense_feature_extraction(self, image, name: str, image_scale: int = 1, mask=None, normalize=False):
        """-> (HWC maps [h,w,cstride] x3 with the confidence as channel C, scales)."""
        maps, scales = self.feature_extractor.extract_packed(image, image_scale, mask, normalize)
        if self.tracker is not None and getattr(self.tracker, "debug", 0) >= 2:
            feats = [m[..., :c].permute(2, 0, 1) for m, c in zip(maps, OUTPUT_DIMS)]
            weight = [m[..., c : c + 1].permute(2, 0, 1) for m, c in zip(maps, OUTPUT_DIMS)]
            self.log_dense(name=name, image=image, image_scale=image_scale, features=feats, scales=scales,
                           weight=weight)
        return maps, scales

    # ---- sparse reference observations --------------------------------------------
    def interp_sparse_observations(self, feature_maps: List[torch.Tensor], feature_scales, image_id: int,
                                   p3dids: List[int], pose: Optional[Pose] = None,
                                   p3d: Optional[torch.Tensor] = None, window=None) -> SparseReferenceFeatures:
        """``window`` = (x0, y0, full_w, full_h) in image pixels: the maps are those of that window of the reference
        image (reference_window); sampling addresses them as windows of the full levels."""
        image = self.model3d.dbs[image_id]
        camera = self._reference_camera_of(image.camera_id)
        T_w2cam = Pose.from_colmap(image) if pose is None else pose
        if p3d is None:
            p3d = torch.from_numpy(np.array([self.model3d.points3D[p].xyz for p in p3dids], np.float32)).to(self.device)
        n = int(p3d.shape[0])
        outs, cams, ndist = [], [], []
        for fm, sc in zip(feature_maps, feature_scales):
            outs.append(torch.empty(n, fm.shape[2], device=self.device, dtype=torch.float32))
            cam_l = camera.scale(sc)
            cams += cam_l.as10().tolist()
            ndist.append(int(cam_l._data.shape[-1] - 6))
        valid = torch.empty(n, dtype=torch.uint8, device=self.device)
        pad = self.optimizer[0].interpolator.pad
        T12 = T_w2cam.as12().detach().cpu().reshape(-1).tolist()
        windows = None
        if window is not None:
            x0, y0, fw, fh = window
            windows = []
            for fm, stride in zip(feature_maps, self.feature_extractor.model.scales):
                # level l of the full image: the sizes UNet.level_shapes gives it; the window starts at x0 / stride
                lh, lw = self.feature_extractor.model.level_shapes(fh, fw)[len(windows) // 4]
                windows += [x0 // stride, y0 // stride, lw, lh]
        ops.sample_sparse(p3d, T12, list(feature_maps), list(OUTPUT_DIMS[:len(feature_maps)]), cams, ndist, int(pad),
                          True, outs, valid, windows)
        return SparseReferenceFeatures(outs, valid, list(p3dids), p3d, OUTPUT_DIMS)

    def extract_reference_features(self, dbids, pose: Optional[Pose] = None, reference_image=None):
        multiscales = self.conf.multiscale or [1]
        if reference_image is None:
            raise NotImplementedError("static reference images on disk (r5/r7 mode) are out of scope: "
                                      "r9 always passes the NeRF render (pixloc_tracker_r9.py:157-159)")
        p3dids, p3d = self._points_of(dbids)
        ref_img = self.model3d.dbs[dbids[0]]
        if pose is None:
            pose = Pose.from_Rt(ref_img.qvec2rotmat(), ref_img.tvec)
        features = {}
        for image_scale in multiscales:
            image, window = self.reference_window(dbids, pose, reference_image)
            maps, scales = self.dense_feature_extraction(image, ref_img.name, image_scale)
            self.last_reference_wh = self.feature_extractor.last_input_wh  # (w, h) the reference pass ran at
            features[str(image_scale)] = self.interp_sparse_observations(maps, scales, dbids[0], p3dids, pose, p3d, window)
        return features

    # ---- pre-extracted reference features (reference :175-198) ---------------------------
    # File layout of the reference's reader: group[str(ref_id)][str(scale)]["p3dids"] (int ids) and
    # group[str(ref_id)][str(scale)][str(level)]["p3did_to_feat"] ([n, C_level + 1]: descriptor then
    # confidence, row i belongs to p3dids[i]).  The reference tree has a reader but no writer; both
    # directions live here.  Containers: `reference_features.h5` when h5py is importable (it is not
    # in this image), else `reference_features.npz` with the same hierarchy as "/"-joined keys.
    def _feature_cache_paths(self):
        from pathlib import Path

        dumps = Path(self.paths.get("dumps", "."))
        return dumps / "reference_features.h5", dumps / "reference_features.npz"

    def read_features(self, ref_id) -> Dict[str, SparseReferenceFeatures]:
        h5_path, npz_path = self._feature_cache_paths()
        multiscales = self.conf.multiscale or [1]
        arrays = {}
        if h5_path.exists():
            try:
                import h5py
            except ImportError as e:
                raise _lib.PxtError(f"{h5_path} needs h5py, which this environment lacks; convert it to "
                                    f"{npz_path.name} (same keys, '/'-joined)") from e
            with h5py.File(str(h5_path), "r") as f:
                for scale in multiscales:
                    g = f[str(ref_id)][str(scale)]
                    arrays[f"{scale}/p3dids"] = np.array(g["p3dids"])
                    for level in (k for k in g.keys() if k != "p3dids"):
                        arrays[f"{scale}/{level}/p3did_to_feat"] = np.array(g[level]["p3did_to_feat"])
        elif npz_path.exists():
            with np.load(npz_path) as z:
                pre = f"{ref_id}/"
                arrays = {k[len(pre):]: z[k] for k in z.files if k.startswith(pre)}
        else:
            raise FileNotFoundError(f"no reference feature cache at {h5_path} or {npz_path}")
        out = {}
        for scale in multiscales:
            p3dids = [int(x) for x in arrays[f"{scale}/p3dids"].tolist()]
            packed = []
            for level, c in enumerate(OUTPUT_DIMS):
                feat = torch.from_numpy(np.asarray(arrays[f"{scale}/{level}/p3did_to_feat"], np.float32))
                assert feat.shape == (len(p3dids), c + 1), (feat.shape, len(p3dids), c)
                rec = torch.zeros(len(p3dids), cstride_for(c))
                d = feat[:, :c]
                rec[:, :c] = d / d.norm(dim=1, keepdim=True).clamp_min(1e-12)  # A.4: L2-normalised reference
                rec[:, c] = feat[:, c]
                packed.append(rec.to(self.device))
            xyz = np.array([self.model3d.points3D[p].xyz for p in p3dids], np.float32).reshape(-1, 3)
            valid = torch.ones(len(p3dids), dtype=torch.uint8, device=self.device)
            out[str(scale)] = SparseReferenceFeatures(packed, valid, p3dids, torch.from_numpy(xyz).to(self.device),
                                                      OUTPUT_DIMS)
        return out

    def write_features(self, features_by_ref: Dict, path=None) -> str:
        """Writes {ref_id: {str(scale): SparseReferenceFeatures}} (e.g. what extract_reference_features
        returns) in the layout read_features reads; only points valid on every level are stored.
        Descriptors are stored as held here (L2-normalised; the reader normalises again)."""
        h5_path, npz_path = self._feature_cache_paths()
        arrays = {}
        for ref_id, per_scale in features_by_ref.items():
            for scale, ref in per_scale.items():
                keep = ref.valid.cpu().bool()
                ids = [p for p, k in zip(ref.p3dids_all, keep.tolist()) if k]
                arrays[f"{ref_id}/{scale}/p3dids"] = np.asarray(ids, np.int64)
                for level, c in enumerate(ref.dims):
                    arrays[f"{ref_id}/{scale}/{level}/p3did_to_feat"] = ref.packed[level].cpu()[keep][:, : c + 1].numpy()
        try:
            import h5py
        except ImportError:
            target = str(path or npz_path)
            np.savez(target, **arrays)
            return target if target.endswith(".npz") else target + ".npz"
        target = str(path or h5_path)
        with h5py.File(target, "w") as f:
            for k, v in arrays.items():
                f.create_dataset(k, data=v)
        return target

    # ---- refinement ------------------------------------------------------------------
    def refine(self, qname: str, qcamera: Camera, pose_init: Pose, dbids: List[int], loc=None,
               image_query=None, pose: Optional[Pose] = None, reference_images=None, dynamic_id=None) -> Dict:
        fail = {"success": False, "T_init": pose_init, "dbids": dbids}
        p3dids, _ = self._points_of(dbids)
        if len(p3dids) < self.conf.min_points_opt:
            logger.debug("Not enough valid 3D points to optimize")
            return fail
        ret = self.refine_query_pose(qname, qcamera, pose_init, dbids, self.conf.multiscale, image_query, pose,
                                     reference_images, dynamic_id)
        return {**ret, "dbids": dbids}

    def refine_query_pose(self, qname, qcamera: Camera, T_init: Pose, dbids, multiscales=None, image_query=None,
                          pose=None, reference_images=None, dynamic_id=None) -> Dict:
        if multiscales is None:
            multiscales = [1]
        if reference_images is not None:
            raise NotImplementedError("raw reference images per query (r6/r8 mode) are out of scope")
        if dynamic_id is None:
            # static references (the r5/r7 trackers' mode, reference :243-248): pre-extracted sparse
            # features of the reference image, read once from the cache file and kept
            ref_id = dbids[0]
            if ref_id in self.features_dicts:
                features_dict = self.features_dicts[ref_id]
            else:
                features_dict = self.read_features(ref_id)
            self.features_dicts[ref_id] = features_dict
        else:
            features_dict = self.features_dicts[dynamic_id]["features"]
        self.last_lm = []
        ret = {"success": False, "T_init": T_init}
        for image_scale in multiscales:
            ref = features_dict[str(image_scale)]
            maps_q, scales_q = self.dense_feature_extraction(image_query, qname, image_scale, mask=self.query_mask,
                                                             normalize=True)
            # Infrastructure errors (HIP, spin bound) raise PxtError out of here; only the
            # algorithmic failure becomes success=False (SURVEY 8b error convention; the
            # reference's bare `except:` at :259-265 would have hidden both).
            plan = self.conf.level_plan
            levels = None if not plan else plan.get(image_scale, plan.get(str(image_scale)))
            ret = self.refine_pose_using_features(maps_q, scales_q, qcamera, T_init, ref, levels)
            if not ret["success"]:
                logger.info(f"Optimization failed for query {qname}")
                break
            T_init = ret["T_refined"]
        return ret

    def refine_pose_using_features(self, features_query: List[torch.Tensor], scales_query, qcamera: Camera,
                                   T_init: Pose, ref: SparseReferenceFeatures, levels=None) -> Dict:
        """pixloc BaseRefiner.refine_pose_using_features on packed buffers: coarse -> fine,
        optimizer[level] per level, all levels in one kernel launch.  ``levels`` (optional) keeps
        only those pyramid levels (conf.level_plan)."""
        prob = self.lm_problem(features_query, scales_query, qcamera, T_init, ref, levels)
        pending = PixTrackOptimizer.refine_levels(ref.p3d, prob["packs"], T_init, prob["conf"], self._ws, mask=ref.valid,
                                                  camera=prob["camera"])
        # the refinement is enqueued, its result not yet awaited: a caller may queue work behind it that reads the
        # pose record on the device (the tracker's next render, pixloc_tracker_r9._render_ahead)
        hook = getattr(self, "after_lm_enqueued", None)
        if hook is not None:
            hook(pending)
        return self.lm_finish(prob, pending.result())

    def lm_problem(self, features_query, scales_query, qcamera: Camera, T_init: Pose, ref: SparseReferenceFeatures,
                   levels=None) -> Dict:
        """Everything one refinement's launch needs (levels in execution order, the native configuration, the camera
        record of a render to be queued behind the launch): what refine_pose_using_features enqueues by itself and what
        the lock-step multi-object tracker collects from K refiners for ONE batched launch (pxt_lm_refine_batch)."""
        n_levels = len(features_query)
        order = [l for l in reversed(range(n_levels)) if levels is None or l in levels]
        packs = []
        for level in order:
            opt = self.optimizer[level] if isinstance(self.optimizer, (list, tuple)) else self.optimizer
            packs.append(LevelPack(features_query[level], ref.packed[level], OUTPUT_DIMS[level],
                                   qcamera.scale(scales_query[level]), opt.dampingnet()))
        opt0 = self.optimizer[order[0]] if isinstance(self.optimizer, (list, tuple)) else self.optimizer
        # (a tracker that will queue the next frame's render behind this launch asks for the camera of that render in
        # the kernel's epilogue: pixloc_tracker_r9._lm_camera)
        cam_provider = getattr(self, "lm_camera", None)
        return {"order": order, "packs": packs, "conf": opt0.native_conf(), "T_init": T_init, "ref": ref,
                "camera": cam_provider() if cam_provider is not None else None, "workspace": self._ws}

    def lm_finish(self, prob: Dict, res) -> Dict:
        """The host side after a refinement's result has arrived: log replay into the tracker hooks, the
        ``{"success", "T_refined", "diff_R", "diff_t"}`` dictionary of pixloc's refine_pose_using_features."""
        order, packs, T_init, ref = prob["order"], prob["packs"], prob["T_init"], prob["ref"]
        self.last_lm.append(res)
        # replay the iteration log into the tracker hooks, level by level (only when someone listens:
        # DebugTracker ignores everything below debug level 1, tracker.py:33-34)
        listening = self.tracker is not None and getattr(self.tracker, "debug", 1) >= 1
        T_level = T_init
        for k, level in enumerate(order if listening else []):
            if res.iters[k] == 0:
                break
            opt = self.optimizer[level] if isinstance(self.optimizer, (list, tuple)) else self.optimizer
            opt.replay_log(res, k, T_level)
            T_level = Pose(res.log[k, res.iters[k] - 1, 8:20].clone())
            self.log_optim(i=k, T_opt=T_level, fail=res.failed, level=level, p3d=None, p3d_ids=ref.p3dids_all,
                           T_init=T_init, camera=packs[k].camera)
        ret = {"T_init": T_init}
        if res.failed:
            return {**ret, "success": False}
        T_opt = Pose(res.T.as12().double())  # already on the host (pixloc: T_opt.cpu().double())
        # (T_init^-1 @ T_opt).magnitude() in float64 as pixloc does, on 24 Python floats: this runs
        # between the LM result and the next frame's first launch
        a, b = T_init.as12().detach().cpu().double().tolist(), T_opt.as12().tolist()
        cos = min(1.0, max(-1.0, (sum(a[i] * b[i] for i in range(9)) - 1.0) / 2.0))  # trace(R0^T R1)
        dR = abs(math.acos(cos)) / math.pi * 180.0
        d = (b[9] - a[9], b[10] - a[10], b[11] - a[11])
        dt = math.sqrt(sum((a[j] * d[0] + a[3 + j] * d[1] + a[6 + j] * d[2]) ** 2 for j in range(3)))  # |R0^T d|
        return {**ret, "success": True, "T_refined": T_opt, "diff_R": dR, "diff_t": dt}

    # ---- one refinement in two halves (lock-step multi-object tracking) -------------------------------
    def begin_refine(self, qname: str, qcamera: Camera, pose_init: Pose, dbids: List[int], image_query, dynamic_id):
        """refine() -> refine_query_pose() of ONE image scale up to, but without, the LM launch: returns
        ("done", ret) when the reference's early exits apply (too few points), else ("lm", problem) - the problem goes
        into a batched launch and comes back through finish_refine.  Same calls in the same order as refine()."""
        fail = {"success": False, "T_init": pose_init, "dbids": dbids}
        p3dids, _ = self._points_of(dbids)
        if len(p3dids) < self.conf.min_points_opt:
            logger.debug("Not enough valid 3D points to optimize")
            return "done", fail
        multiscales = self.conf.multiscale or [1]
        assert len(multiscales) == 1, "lock-step refinement covers one image scale (cold starts run through refine())"
        image_scale = multiscales[0]
        features_dict = self.features_dicts[dynamic_id]["features"]
        self.last_lm = []
        ref = features_dict[str(image_scale)]
        maps_q, scales_q = self.dense_feature_extraction(image_query, qname, image_scale, mask=self.query_mask,
                                                         normalize=True)
        plan = self.conf.level_plan
        levels = None if not plan else plan.get(image_scale, plan.get(str(image_scale)))
        prob = self.lm_problem(maps_q, scales_q, qcamera, pose_init, ref, levels)
        prob["dbids"], prob["qname"] = dbids, qname
        return "lm", prob

    def finish_refine(self, prob: Dict, res) -> Dict:
        ret = self.lm_finish(prob, res)
        if not ret["success"]:
            logger.info(f"Optimization failed for query {prob['qname']}")
        return {**ret, "dbids": prob["dbids"]}


class Paths(dict):
    """pixloc.utils.data.Paths stand-in: attribute dict + add_prefixes."""

    def __getattr__(self, k):
        try:
            return self[k]
        except KeyError as e:
            raise AttributeError(k) from e

    def add_prefixes(self, dataset, dumps, results):
        from pathlib import Path

        out = Paths(self)
        for k in ("query_images", "reference_images", "reference_sfm", "query_list"):
            if out.get(k) is not None:
                out[k] = Path(dataset) / out[k] if k in ("query_images",) else Path(dumps) / out[k]
        out["dumps"] = Path(dumps)
        out["results_dir"] = Path(results)
        return out


class PoseTrackerLocalizer:
    """Builds model3d + extractor + the three per-level optimizers + refiner
    (reference pixloc_pose_refiners.py:28-118).  ``experiment`` weights come either from a
    dict of tensors (``conf["weights"]``: UNet names of pixtrack_amd.unet + "optimizer.{i}.
    dampingnet.const") or from ``conf["weights_path"]`` (a torch file of that dict, or a pixloc
    ``checkpoint_best.tar``; see unet.load_weights)."""

    def __init__(self, paths, conf, device: Optional[torch.device] = None, model3d: Optional[Model3D] = None):
        if device is None:
            if not torch.cuda.is_available():
                raise _lib.PxtError("no ROCm device: the tracking path has no CPU implementation")
            device = torch.device("cuda:0")
        self.device = torch.device(device)
        self.model3d = model3d if model3d is not None else Model3D(paths.reference_sfm)
        self.queries = {}
        conf = Conf(conf)
        weights = conf.get("weights")
        if weights is None:
            from .unet import load_weights

            weights = load_weights(conf["weights_path"])
        conf_optim = merge({"num_iters": 100}, conf.get("optimizer", {}))
        extractor = UNet(weights, self.device)
        optimizer = []
        for i in range(3):
            opt = PixTrackOptimizer(conf_optim, device=self.device)
            key = f"optimizer.{i}.dampingnet.const"
            if key in weights:
                opt.dampingnet.const = torch.as_tensor(weights[key]).float().cpu()
            optimizer.append(opt)
        self.paths = paths
        self.conf = conf
        self.optimizer = optimizer
        self.extractor = PixTrackFeatureExtractor(extractor, self.device, conf.get("features", {}).get("preprocessing", {}))
        self.refiner = PoseTrackerRefiner(self.device, self.optimizer, self.model3d, self.extractor, paths,
                                          conf.get("refinement", {}))
        self.logs = None

    def run_query(self, name: str, camera: Camera, pose_init: Pose, reference_images, image_query=None,
                  pose: Optional[Pose] = None, reference_images_raw=None, dynamic_id=None):
        loc = None if self.logs is None else self.logs[name]
        return self.refiner.refine(name, camera, pose_init, reference_images, loc=loc, image_query=image_query,
                                   pose=pose, reference_images=reference_images_raw, dynamic_id=dynamic_id)
