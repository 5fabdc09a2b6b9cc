"""PixLocPoseTrackerR9 -- the production tracker and its CLI
(reference pixtrack/pose_trackers/pixloc_tracker_r9.py:32-318).

Per-frame policy kept from the reference: cold start refines at image scales [4, 1] from the upright
reference pose; afterwards scale [1] with the query masked by the dilated NeRF depth silhouette of the previous pose; the
reference view is re-rendered with the NeRF at the current pose every frame (the THRESH = 0 dynamic-reference cache never
hits, Appendix D.2); success = optimiser success AND cost <= 1.1 x first-frame cost.

Device residency (the MI355X-first part): the NeRF frames, the mask, the query image, both feature pyramids and the sparse
reference features never leave HBM; the LM kernel writes the pose and the iteration log into pinned host memory.

CLI (unchanged): --object_path P --query DIR --out_dir DIR [--frames N] [--debug 0/1/2];
env UPRIGHT_REF_IMG, OBJ_AABB; outputs poses.pkl, trackers.pkl.
"""
from __future__ import annotations

import argparse
import ast
import os
import pickle as pkl
from pathlib import Path

import numpy as np
import torch
from scipy.spatial.transform import Rotation as R

from .. import _lib
from ..ops import ops
from ..geometry import Camera as PixCamera, Pose
from ..model3d import Model3D, extract_covisibility
from ..refiner import Paths, PoseTrackerLocalizer
from ..tracker import DebugTracker
from ..utils.colmap import ColmapCamera
from ..utils.ingp_utils import initialize_ingp, load_nerf2sfm, sfm_to_nerf_pose
from ..utils.io import ArrayIterator, ImageIterator
from ..utils.pose_utils import (geodesic_distance_for_rotations, geodesic_distances_to,
                                get_camera_in_world_from_pixpose)
from ..visualization.run_vis_on_poses import get_nerf_image_device, rgba_to_u8
from .base_pose_tracker import PoseTracker


def infer_camera_from_image(width: int, height: int) -> ColmapCamera:
    """pycolmap.infer_camera_from_image without EXIF: SIMPLE_RADIAL, f = 1.2 max(w, h),
    principal point at the image centre, k = 0."""
    f = 1.2 * max(width, height)
    return ColmapCamera(None, "SIMPLE_RADIAL", int(width), int(height), np.array([f, width / 2.0, height / 2.0, 0.0]))


class PixLocPoseTrackerR9(PoseTracker):
    def __init__(self, object_path, data_path, loc_path, eval_path, debug=0, device=None, assets=None):
        """``assets`` (optional) supplies everything that otherwise comes from disk, for the
        synthetic runs: dict(model3d, nerf2sfm, snapshot, weights, covis=None, aabb, upright_ref_img)."""
        default_paths = Paths(query_images="query/", reference_images=loc_path, reference_sfm="aug_sfm",
                              query_list="*_with_intrinsics.txt", global_descriptors="features.h5",
                              retrieval_pairs="pairs_query.txt", results="pixloc_object.txt")
        pixloc_conf = {
            "experiment": "pixloc_megadepth",
            "features": {},
            "optimizer": {"num_iters": 150, "pad": 1},
            "refinement": {"num_dbs": 1, "multiscale": [1], "point_selection": "all",
                           "normalize_descriptors": True, "average_observations": False,
                           "do_pose_approximation": False},
        }
        self.debug = debug
        self.device = torch.device(device if device is not None else "cuda:0")
        paths = default_paths.add_prefixes(Path(data_path), Path(loc_path), Path(eval_path))
        if assets is not None:
            pixloc_conf["weights"] = assets["weights"]
            model3d = assets["model3d"]
        else:
            pixloc_conf["weights_path"] = os.environ.get(
                "PIXTRACK_WEIGHTS", str(Path(object_path) / "pixtrack/pixloc_megadepth.pt"))
            model3d = Model3D(paths.reference_sfm)
        self.localizer = PoseTrackerLocalizer(paths, pixloc_conf, device=self.device, model3d=model3d)
        self.eval_path = eval_path
        covis_path = Path(paths.reference_sfm) / "covis.pkl"
        if assets is not None and assets.get("covis") is not None:
            self.covis = assets["covis"]
        elif assets is None and os.path.isfile(covis_path):
            with open(covis_path, "rb") as f:
                self.covis = pkl.load(f)
        else:
            self.covis = extract_covisibility(self.localizer.model3d)
            if assets is None:
                with open(str(covis_path), "wb") as f:
                    pkl.dump(self.covis, f)
        self.pose_history = {}
        self.pose_tracker_history = {}
        self.cold_start = True
        self.pose = None
        self.reference_ids = self._initial_reference_ids(assets)
        self.reference_scale = 0.5
        self.localizer.refiner.reference_scale = self.reference_scale
        if assets is not None:
            self.nerf2sfm = assets["nerf2sfm"]
            snapshot = assets["snapshot"]
        else:
            self.nerf2sfm = load_nerf2sfm(str(Path(data_path) / "nerf2sfm.pkl"))
            snapshot = str(Path(object_path) / "pixtrack/instant-ngp/snapshots/weights.msgpack")
        self.testbed = initialize_ingp(snapshot, self._render_aabb(assets), device=self.device)
        self.localizer.refiner.warm_reference_points()  # static per-reference tables, off the frame path
        self.dynamic_id, self.cache_hit, self.cost_threshold, self.camera = None, False, None, None
        self.hits = self.misses = self.relocalization_count = 0
        self.success = True
        self.spp = 8  # run_vis_on_poses.py:29
        self.batch_frame_images = True  # reference render + masked query in one batched UNet pass
        self._fused_reference = None  # (pose object, uint8 image) handed from get_mask to get_reference_image
        self._ref_cam_cache = self._coincide_cache = None
        self.keep_feature_history = False  # the reference leaks one entry per frame (Appendix D.2)
        self.steady_multiscale = [1]  # image scales of a tracked (non-cold-start) frame (:223)
        # A frame's mask (Depth, query camera) and reference image (Shade, SfM camera 1 x reference_scale) are rendered at
        # the same pose: ONE march when the two cameras coincide (fuse_identical_views; same size and fx: all
        # get_nerf_image reads), else both renders as ONE chain of launches (Testbed.render_frame_pair_device); the
        # renderer's last kernel writes the 8-bit reference image and the mask's `!= 0` plane itself.
        self.fuse_identical_views = True
        # The next frame's renders need only this frame's pose: with `render_ahead` they are enqueued
        # BEHIND this frame's LM launch, their camera taken from the slot(s) the LM kernel's epilogue fills, instead of
        # after the host has read the result, run the policy and converted the pose (~0.1 ms of GPU idle per frame).  They
        # are consumed only if the host, once it knows the pose, arrives at the same 12 camera floats; otherwise (failed
        # frame, cost gate, relocalisation, a different last bit) the frame renders as before.  PXT_RENDER_AHEAD=0: off.
        self.render_ahead = os.environ.get("PXT_RENDER_AHEAD", "1") != "0"
        self._ahead_cam = None   # the pinned camera record the LM epilogue of this frame writes
        self._ahead = None       # the render queued behind the last LM launch
        self._ahead_ok = None    # ... once verified: (pose object it is valid for, mask, uint8 reference image)
        # queued renders: consumed / camera record never arrived within the poll bound / host and device disagree on a camera
        # bit / view settings changed between enqueue and use
        self.renders_ahead_used = self.renders_ahead_dropped = self.renders_ahead_rejected = self.renders_ahead_stale = 0

    # ------------------------------------------------------------------ per-variant set-up
    def _initial_reference_ids(self, assets):
        """r9: the upright reference image named by $UPRIGHT_REF_IMG (:77-78)."""
        upright_ref_img = assets["upright_ref_img"] if assets is not None else os.environ["UPRIGHT_REF_IMG"]
        return [self.localizer.model3d.name2id[upright_ref_img]]

    def _render_aabb(self, assets):
        """r9: the render box is $OBJ_AABB (:85-86)."""
        return assets["aabb"] if assets is not None else ast.literal_eval(os.environ["OBJ_AABB"])

    # ------------------------------------------------------------------ relocalisation
    def relocalize(self, query):
        if self.cold_start:
            self.camera = self.get_query_camera(query)
            self.cold_start = False
        if self.pose is None:
            ref_img = self.localizer.model3d.dbs[self.reference_ids[0]]
            self.pose = Pose.from_Rt(ref_img.qvec2rotmat(), ref_img.tvec)
        self.relocalization_count += 1

    def start_segment(self, pose_init: Pose):
        """Cold start of a frame SEGMENT that does not begin at the video's first frame (BASELINE configs[4]: one video cut
        into per-GPU segments).  The reference has no such entry point - its only cold start is the upright reference pose
        of frame 0 (:95-106) - so a segment head needs a pose from outside (a relocaliser; the synthetic runs pass a
        perturbed ground truth).  What follows is the reference's cold-start policy: image scales [4, 1], no mask, nearest
        reference image by rotation, cost threshold frozen again from this segment's first frame."""
        self.pose, self.cold_start, self.success = pose_init, True, True
        self.cost_threshold, self.dynamic_id, self.cache_hit = None, None, False
        R_qry = pose_init.numpy()[0]
        dbs = self.localizer.model3d.dbs
        self.reference_ids = sorted(dbs, key=lambda r: geodesic_distance_for_rotations(R_qry, dbs[r].qvec2rotmat()))[:1]

    def get_query_camera(self, query):
        """Camera of the query stream from the frame size (EXIF-less pycolmap heuristic)."""
        image = query[1] if isinstance(query, tuple) else query
        if isinstance(image, (str, os.PathLike)):
            from ..utils.io import read_image

            image = read_image(image)
        h, w = int(image.shape[0]), int(image.shape[1])
        return PixCamera.from_colmap(infer_camera_from_image(w, h))

    # ------------------------------------------------------------------ reference selection
    def update_reference_ids(self):
        if self.cache_hit == True:  # noqa: E712  (kept: see Appendix D.2)
            return self.reference_ids
        # candidates: the current reference, then its covisible images (> 50 shared points), ranked by the
        # geodesic distance of their rotation to the current pose; ties keep this order (stable sort)
        curr = self.reference_ids[0]
        cache = self.__dict__.setdefault("_ref_candidates", {})  # static per model: rotations of each reference's candidates
        cand = cache.get(curr)
        if cand is None:
            dbs = self.localizer.model3d.dbs
            ids = list(dict.fromkeys([curr] + [k for k, v in self.covis[curr].items() if v > 50]))
            cand = cache[curr] = (ids, np.stack([dbs[r].qvec2rotmat() for r in ids]))
        ids, rots = cand
        gd = geodesic_distances_to(self.pose.numpy()[0], rots)
        self.reference_ids = [ids[int(np.argmin(gd))]]
        return self.reference_ids

    # ------------------------------------------------------------------ NeRF reference + mask
    def _nerf_pose(self, pose):
        return sfm_to_nerf_pose(self.nerf2sfm, get_camera_in_world_from_pixpose(pose))

    def _reference_camera(self):
        key = float(self.reference_scale)
        if self._ref_cam_cache is None or self._ref_cam_cache[0] != key:
            self._ref_cam_cache = (key, PixCamera.from_colmap(self.localizer.model3d.cameras[1]).scale(key))
        return self._ref_cam_cache[1]

    def _views_coincide(self) -> bool:
        """Mask and reference cameras agree in (width, height, fx) - all that get_nerf_image reads from a camera
        (run_vis_on_poses.py:30-36); the answer only changes with the camera object or the reference scale."""
        if not self.fuse_identical_views or self.camera is None:
            return False
        key = (id(self.camera), float(self.reference_scale))
        if self._coincide_cache is None or self._coincide_cache[0] != key:
            vk = lambda cam: (int(cam.size[0]), int(cam.size[1]), float(cam.f[0]))
            self._coincide_cache = (key, vk(self._reference_camera()) == vk(self.camera))
        return self._coincide_cache[1]

    def get_reference_image(self, pose) -> torch.Tensor:
        """uint8 [H,W,3] NeRF render at ``pose`` with SfM camera 1 scaled by reference_scale."""
        if self._fused_reference is not None and self._fused_reference[0] is pose:
            img = self._fused_reference[1]
            self._fused_reference = None
            return img
        ref_camera = self._reference_camera()
        rgba = get_nerf_image_device(self.testbed, self._nerf_pose(pose), ref_camera, spp=self.spp)
        return rgba_to_u8(rgba, 0.0)

    def create_dynamic_reference_image(self, pose):
        nerf_img = self.get_reference_image(pose)
        # (the reference hashes str(R); the id is only a dictionary key - the bytes of R serve, without numpy's
        # array printer on the per-frame host path)
        dynamic_id = hash(pose.numpy()[0].tobytes())
        features = self.localizer.refiner.extract_reference_features(self.reference_ids, pose, nerf_img)
        return dynamic_id, features

    def get_dynamic_id(self, pose):
        features_dicts = self.localizer.refiner.features_dicts
        if self.dynamic_id is None:
            self.dynamic_id, features = self.create_dynamic_reference_image(self.pose)
            features_dicts[self.dynamic_id] = {"pose": self.pose, "features": features}
            return self.dynamic_id
        # THRESH = 0 in the reference: a geodesic distance is never < 0, so every frame after
        # the first is a miss and renders a fresh reference (pixloc_tracker_r9.py:171-203).
        self.THRESH = 0
        self.cache_hit = False
        old_id = self.dynamic_id
        self.dynamic_id, features = self.create_dynamic_reference_image(self.pose)
        if not self.keep_feature_history:
            features_dicts.pop(old_id, None)
        features_dicts[self.dynamic_id] = {"pose": self.pose, "features": features,
                                           "ref_ids": self.update_reference_ids()}
        self.misses += 1
        self.cache_hit = True
        return self.dynamic_id

    def get_mask(self, pose) -> torch.Tensor:
        """uint8 [H,W] on the device: depth render != 0, erode 5x5 x1, dilate 5x5 x5."""
        if self._ahead_ok is not None and self._ahead_ok[0] is pose:
            _, mask, ref_u8, views = self._ahead_ok
            self._ahead_ok = None
            # (it also baked in focal length, size, spp, lens, render box, background and minimum transmittance as they were
            # when it was enqueued: it stands in for this frame's render only if they are what this frame would use)
            if views == self._ahead_views_now():
                self._fused_reference = (pose, ref_u8)
                self.renders_ahead_used += 1
                return mask
            self.renders_ahead_stale += 1
        self._ahead_ok = None
        return self._mask_and_reference(pose, from_slot=False)[0]

    def _frame_views(self):
        """(width, height, fov in degrees on x) of the frame's renders as get_nerf_image sets them up
        (run_vis_on_poses.py:30-38): [query camera] when mask and reference cameras coincide (one march yields both
        images), else [query camera (Depth), reference camera (Shade)]."""
        import math

        def fov_of(cam):
            w, h = (int(v) for v in cam.size)
            return w, h, math.atan(w / (float(cam.f[0]) * 2)) * 2 * 180 / np.pi

        return [fov_of(self.camera)] if self._views_coincide() else [fov_of(self.camera), fov_of(self._reference_camera())]

    def _mask_and_reference(self, pose, from_slot: bool):
        """The frame's mask and 8-bit reference image at one pose (reference :145-152, :207-214): ONE march when the two
        views coincide, otherwise the Depth render (query camera) and the Shade render (reference camera) as one chain of
        launches on this stream.  ``from_slot``: the camera is the one the LM kernel ahead in the stream derived from its
        final pose (render-ahead); otherwise the host's conversion of ``pose``.  Sets ``_fused_reference`` when a pose
        object is given; returns (mask, ref_u8)."""
        tb, spp = self.testbed, int(self.spp)
        if not from_slot:
            tb.set_nerf_camera_matrix(np.asarray(self._nerf_pose(pose))[:3, :])
        views = self._frame_views()
        width, height, fov_q = views[0]
        if len(views) == 1:
            tb.fov = fov_q
            out = tb.render_frame_device(width, height, spp, mode=2, from_slot=from_slot)
            ref_u8, nz = out["rgb_u8"], out["depth_nz"]
        else:
            nz, ref_u8 = tb.render_frame_pair_device(views[0], views[1], spp, from_slot=from_slot)
        mask = self._mask_of(nz)
        if pose is not None:
            self._fused_reference = (pose, ref_u8)
        return mask, ref_u8

    def _mask_of(self, nz):
        """get_mask's morphology on the `uint8(depth * 255) != 0` plane a render wrote: erode 5x5 once, dilate 5x5 five
        times (:211-213)."""
        mask = torch.empty(nz.shape[0], nz.shape[1], dtype=torch.uint8, device=self.device)
        ops.depth_mask_plane(nz, 1, 5, mask)
        return mask

    def _lm_camera(self):
        """What the refiner hands to the LM launch of a frame whose next render will be queued behind it: the pose ->
        camera conversion constants, the renderer camera slot(s) to fill, and a pinned record for the host's check."""
        conv = self.__dict__.get("_pose_conv")
        if conv is None:
            conv = self._pose_conv = self.testbed.pose_conversion(self.nerf2sfm)
        slots = [self.testbed.camera_slot()]
        if not self._views_coincide():  # the pair's Shade render runs through the testbed's second context
            slots.append(self.testbed.camera_slot(side=True))
        self._ahead_cam = self.testbed._next_cam_out()
        return conv, slots, self._ahead_cam

    # ------------------------------------------------------------------ the next frame's render, ahead of the host
    def _render_ahead(self, pending):
        """Called between the LM launch and the wait for its result: the mask + reference render(s) of the NEXT frame, whose
        camera the LM kernel's epilogue writes into the renderer's camera slot(s)."""
        views = self._ahead_views_now()
        mask, ref_u8 = self._mask_and_reference(None, from_slot=True)
        self._ahead = ([self._ahead_cam], mask, ref_u8, views)

    def _render_ahead_request(self):
        """For a caller that merges the queued renders of several trackers into one batched chain (lock-step tracking,
        Testbed.render_frame_batch_device): (width, height, spp) of this tracker's one-march mask + reference render with
        the testbed's view set for it - or None when the frame's render is not of that kind (two different cameras), in
        which case _render_ahead() is to be called as usual."""
        if not self._views_coincide():
            return None
        width, height, self.testbed.fov = self._frame_views()[0]
        return int(width), int(height), int(self.spp)

    def _render_ahead_accept(self, out):
        """What _render_ahead() does after its render, for a render that was part of a batched chain."""
        self._ahead = ([self._ahead_cam], self._mask_of(out["depth_nz"]), out["rgb_u8"], self._ahead_views_now())

    def _ahead_views_now(self):
        """What a render queued now bakes in besides the camera pose, per render of the frame: focal length, lens, render
        box, background, minimum transmittance (floats 12.. of the view record), image size and spp - the keys
        get_mask / get_reference_image would render with right now."""
        spp = int(self.spp)
        return [(tuple(self.testbed._view_for(w, h, fov)[12:]), w, h, spp) for w, h, fov in self._frame_views()]

    def _verify_render_ahead(self, success: bool):
        """The render queued behind the LM launch is kept for the next frame only if the pose was accepted and the
        host, going through the reference's own conversion chain, arrives at the 12 camera floats the device used."""
        ahead, self._ahead = self._ahead, None
        self._ahead_ok = None
        if ahead is None or not success:
            return
        cams, mask, ref_u8, views = ahead
        self.testbed.set_nerf_camera_matrix(np.asarray(self._nerf_pose(self.pose))[:3, :])
        want = np.asarray(self.testbed._cam_ngp, np.float32).reshape(-1)
        for cam_out in cams:
            got = cam_out.numpy()
            for _ in range(200000):  # the camera kernel runs right behind the LM kernel whose result is already here
                if got[12] != 0.0:
                    break
            else:  # never arrived: the queued render is dropped (and counted), the frame renders as usual
                self.renders_ahead_dropped += 1
                return
            if not np.array_equal(want.view(np.uint32), got[:12].view(np.uint32)):
                self.renders_ahead_rejected += 1
                return
        self._ahead_ok = (self.pose, mask, ref_u8, views)

    # ------------------------------------------------------------------ one frame
    def refine(self, query):
        query_path, query_image = query
        refiner = self.localizer.refiner
        self._frame_setup(query)
        self.dynamic_id = self.get_dynamic_id(self.pose)
        trackers, rets, costs = {}, {}, {}
        for ref_id in self.reference_ids:
            pose_init = self._frame_pose_init()
            tracker = DebugTracker(refiner, self.debug)
            ret = self.localizer.run_query(query_path, self.camera, pose_init, [ref_id], image_query=query_image,
                                           pose=self.pose, reference_images_raw=None, dynamic_id=self.dynamic_id)
            rets[ref_id] = ret
            trackers[ref_id] = tracker
            costs[ref_id] = self._frame_cost()
        return self._frame_policy(query_path, rets, costs, trackers)

    def _frame_setup(self, query, lockstep: bool = False) -> str:
        """The head of refine() (reference :216-227): cold start -> image scales [4, 1] from the relocalised pose; a
        tracked frame -> scale [1] with the query masked by the silhouette at the last pose; after a failed frame no
        mask and whatever scales were last set (Appendix D.3).  Arms the extractor's two-image pass and the render that
        rides behind the LM launch.  Returns "cold" / "steady" / "plain".  ``lockstep``: the frame's two UNet passes
        and its LM launch are batched with other objects' by the caller (multi_object_tracker.py), which hands the
        maps over through the extractor's preload() - nothing is staged here."""
        query_path, query_image = query
        refiner = self.localizer.refiner
        refiner.query_mask = None
        kind = "plain"
        if self.cold_start:
            refiner.conf.multiscale = [4, 1]
            self.relocalize(query)
            self.cold_start = False
            kind = "cold"
        elif self.success:
            refiner.conf.multiscale = list(self.steady_multiscale)
            refiner.query_mask = self.get_mask(self.pose)  # multiplied inside the first conv

        # The masked query is fully known here, before the reference render is encoded: announce it so both images of
        # the frame go through the UNet in one batched pass.
        if not lockstep and self.batch_frame_images and refiner.conf.multiscale == [1]:
            refiner.feature_extractor.stage(query_image, 1, refiner.query_mask, True)
        else:
            refiner.feature_extractor.unstage()
        # one refinement, at full scale, of a tracked frame: its LM launch can carry the next frame's render(s)
        # behind it (one march when mask and reference views coincide, two renders otherwise)
        steady = (kind != "cold" and self.success and refiner.query_mask is not None
                  and refiner.conf.multiscale == [1] and len(self.reference_ids) == 1)
        self._ahead = None
        refiner.after_lm_enqueued = self._render_ahead if (self.render_ahead and steady) else None
        refiner.lm_camera = self._lm_camera if (self.render_ahead and steady) else None
        return "steady" if steady else kind

    def _frame_pose_init(self) -> Pose:
        rotation, translation = self.pose.numpy()
        rotation = R.from_matrix(rotation).as_matrix()
        return Pose.from_Rt(rotation, translation)

    def _frame_cost(self) -> float:
        """mean over (scale, level) runs of the LAST logged cost; taken from the kernel log so it
        does not depend on --debug (the reference yields NaN with --debug 0, Appendix D.1)"""
        last = [c[-1] for res in self.localizer.refiner.last_lm for c in res.costs if len(c)]
        return float(np.mean(last)) if last else float("nan")

    def _frame_policy(self, query_path, rets, costs, trackers) -> bool:
        """The tail of refine() (reference :251-275): best reference by cost, the cost gate frozen from the first
        frame, pose update, history."""
        refiner = self.localizer.refiner
        avg_cost = costs[self.reference_ids[-1]]
        if hasattr(self, "pbar") and hasattr(self.pbar, "set_description"):
            self.pbar.set_description(f"Cost: {avg_cost}, Relocalizations: {self.relocalization_count}")
        best_ref_id = min(costs, key=costs.get)
        ret = rets[best_ref_id]

        if self.cost_threshold is None:
            thresh = min(costs.values())
            self.cost_threshold = thresh + 0.1 * thresh

        success = bool(ret["success"] and min(costs.values()) <= self.cost_threshold)
        if success:
            self.pose = ret["T_refined"]
        self.success = success
        refiner.after_lm_enqueued = None
        refiner.lm_camera = None
        self._verify_render_ahead(success)
        # (`success` stays what the refiner said, as in the reference's poses.pkl; `tracked` - an added key - is the tracker's
        # decision: refiner success AND the cost gate.  bench.py and the pose gather count THIS one.)
        ret["tracked"] = success
        ret["camera"] = self.camera
        ret["reference_ids"] = self.reference_ids
        ret["query_path"] = query_path
        ret["cost"] = costs[best_ref_id]
        img_name = os.path.basename(str(query_path))
        self.pose_history[img_name] = ret
        self.pose_tracker_history[img_name] = trackers[best_ref_id]
        return success

    def get_query_frame_iterator(self, image_folder, max_frames):
        return image_folder if isinstance(image_folder, (ArrayIterator, ImageIterator)) else ImageIterator(image_folder, max_frames)

    def save_poses(self, pixloc_pickles: bool = False):
        """poses.pkl (reference :281-284).  ``pixloc_pickles`` writes Pose/Camera under pixloc's
        class path so the reference's own tools (run_vis_on_poses.py, GetMetrics.ipynb) load it."""
        _dump(self.pose_history, os.path.join(self.eval_path, "poses.pkl"), pixloc_pickles)


def _dump(obj, path, pixloc_pickles: bool):
    if pixloc_pickles:
        from ..utils.io import dump_reference_pickle

        dump_reference_pickle(obj, path)
    else:
        with open(path, "wb") as f:
            pkl.dump(obj, f)


def main(argv=None):
    parser = argparse.ArgumentParser()
    parser.add_argument("--object_path", type=Path)
    parser.add_argument("--query", type=Path)
    parser.add_argument("--out_dir", type=Path)
    parser.add_argument("--frames", type=int, default=None)
    parser.add_argument("--debug", type=int, default=0)
    parser.add_argument("--pixloc_pickles", action="store_true",
                        help="write poses.pkl/trackers.pkl with pixloc's Pose/Camera class paths")
    args = parser.parse_args(argv)
    data_path = args.object_path / "pixtrack/pixsfm/dataset"
    eval_path = args.out_dir
    loc_path = args.object_path / "pixtrack/aug_nerf_sfm"
    os.makedirs(eval_path, exist_ok=True)
    if torch.cuda.is_available():
        from ..parallel import bind_to_device_numa

        bind_to_device_numa(0)
    tracker = PixLocPoseTrackerR9(object_path=str(args.object_path), data_path=str(data_path),
                                  eval_path=str(eval_path), loc_path=str(loc_path), debug=args.debug)
    import gc

    gc.collect()
    gc.freeze()   # the assets stay; a generation-2 collection inside the frame loop is a 10 ms
    gc.disable()  # stall of the host that feeds the GPU
    try:
        tracker.run(args.query, max_frames=args.frames if args.frames is not None else np.inf)
    finally:
        gc.enable()
    tracker.save_poses(args.pixloc_pickles)
    print("Cache hits: %d, misses: %d" % (tracker.hits, tracker.misses))
    _dump(tracker.pose_tracker_history, os.path.join(tracker.eval_path, "trackers.pkl"), args.pixloc_pickles)
    print("Done")


if __name__ == "__main__":
    main()
