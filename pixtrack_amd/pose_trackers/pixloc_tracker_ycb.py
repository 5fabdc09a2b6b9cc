"""PixLocPoseTrackerYCB -- the YCB-Video policy variant (reference
pixtrack/pose_trackers/pixloc_tracker_ycb.py:37-345).  Same kernels as r9,
different host policy:

* ground-truth pose and camera initialise the track and re-initialise it after a failure
  (`relocalize`, :101-115); the reference image is the nearest of ALL covisibility keys by
  rotation (`set_reference_ids`, :117-130);
* the query is masked EVERY frame (:249-250), image scales stay [1] (:57-59),
  `reference_scale = 0.3` (:89);
* success = optimiser success AND t_err < 10 cm AND r_err < 10 deg, where the error is that
  of `self.pose` BEFORE this frame's update (:280-290, :297-303); `ret["gt_pose"]` is kept.

* needs neither $UPRIGHT_REF_IMG nor $OBJ_AABB: the reference image comes from the GT pose, the
  render box from the SfM points (`get_nerf_aabb_from_sfm`, :92).

Frames are (path, image, gt_pose: Pose, gt_camera: Camera) tuples (reference
pixtrack/utils/io.py:13-72 `YCBVideoIterator`).  CLI (:306-345): --object_path P --query "7:10"
--out_dir D [--frames N] [--debug]; the dataset root is $YCB_ROOT or --ycb_root (the reference
hard-codes /data/ycb/).
"""
from __future__ import annotations

import argparse
import os
from pathlib import Path

import numpy as np
from scipy.spatial.transform import Rotation as R

from ..geometry import Pose
from ..tracker import DebugTracker
from ..utils.ingp_utils import get_nerf_aabb_from_sfm
from ..utils.io import YCBVideoIterator
from ..utils.pose_utils import geodesic_distance_for_rotations
from .pixloc_tracker_r9 import PixLocPoseTrackerR9, _dump


class GTFrameIterator:
    """In-memory stand-in for YCBVideoIterator: yields (path, image, gt_pose, gt_camera)."""

    def __init__(self, names, images, gt_poses, camera):
        self.names, self.images, self.gt_poses, self.camera = list(names), list(images), list(gt_poses), camera
        self.idx = 0

    def __iter__(self):
        return self

    def __len__(self):
        return len(self.names)

    def __next__(self):
        if self.idx > len(self) - 1:
            raise StopIteration
        i = self.idx
        self.idx += 1
        return self.names[i], self.images[i], self.gt_poses[i], self.camera


class PixLocPoseTrackerYCB(PixLocPoseTrackerR9):
    def __init__(self, data_path, loc_path, eval_path, object_path, debug=False, device=None, assets=None,
                 ycb_root=None):
        self.object_path = object_path
        self.ycb_root = Path(ycb_root or os.environ.get("YCB_ROOT", "/data/ycb/"))
        super().__init__(object_path, data_path, loc_path, eval_path, debug=int(debug), device=device, assets=assets)
        self.reference_scale = 0.3
        self.localizer.refiner.reference_scale = self.reference_scale
        self.localizer.refiner.conf.multiscale = [1]
        self.reference_ids = None
        self.gt_pose = None
        self.gt_camera = None
        self.t_err = self.r_err = float("nan")

    def _initial_reference_ids(self, assets):
        return None  # chosen from the GT pose at the first relocalisation (:117-130)

    def _render_aabb(self, assets):
        """The YCB variant renders inside the box of the SfM points (:92), never $OBJ_AABB."""
        return get_nerf_aabb_from_sfm(self.localizer.model3d, self.nerf2sfm)

    def relocalize(self, query_path):
        if self.cold_start:
            self.camera = self.gt_camera
            self.cold_start = False
        self.pose = self.gt_pose
        self.set_reference_ids()
        self.relocalization_count += 1

    def set_reference_ids(self):
        R_qry = self.pose.numpy()[0]
        dbs = self.localizer.model3d.dbs
        gdists = {ref: geodesic_distance_for_rotations(R_qry, dbs[ref].qvec2rotmat()) for ref in self.covis}
        self.reference_ids = sorted(gdists, key=lambda x: gdists[x])[:1]
        return self.reference_ids

    def refine(self, query):
        query_path, query_image, gt_pose, gt_camera = query
        self.gt_pose, self.gt_camera = gt_pose, gt_camera
        refiner = self.localizer.refiner
        if self.cold_start:
            self.relocalize(query_path)
            self.cold_start = False
        refiner.query_mask = self.get_mask(self.pose)  # every frame
        # the masked query is fully known here: announce it so that it goes through the UNet beside the reference render
        if self.batch_frame_images and list(refiner.conf.multiscale or [1]) == [1]:
            refiner.feature_extractor.stage(query_image, 1, refiner.query_mask, True)
        else:
            refiner.feature_extractor.unstage()
        # the next frame's mask + reference renders need only this frame's pose: queued behind the LM launch, used next
        # frame if the pose is accepted and the host arrives at the same camera (pixloc_tracker_r9._render_ahead)
        steady = self.render_ahead and list(refiner.conf.multiscale or [1]) == [1] and len(self.reference_ids) == 1
        self._ahead = None
        refiner.after_lm_enqueued = self._render_ahead if steady else None
        refiner.lm_camera = self._lm_camera if steady else None

        self.dynamic_id = self.get_dynamic_id(self.pose)
        rotation, translation = self.pose.numpy()
        rotation = R.from_matrix(rotation).as_matrix()
        trackers, rets, costs = {}, {}, {}
        for ref_id in self.reference_ids:
            pose_init = Pose.from_Rt(rotation, translation)
            tracker = DebugTracker(refiner, self.debug)
            ret = self.localizer.run_query(query_path, self.camera, pose_init, [ref_id], image_query=query_image,
                                           pose=self.pose, reference_images_raw=None, dynamic_id=self.dynamic_id)
            rets[ref_id], trackers[ref_id] = ret, tracker
            last = [c[-1] for res in refiner.last_lm for c in res.costs if len(c)]
            costs[ref_id] = float(np.mean(last)) if last else float("nan")
        best_ref_id = min(costs, key=costs.get)
        ret = rets[best_ref_id]
        self.calculate_error()
        ret["camera"] = self.camera
        ret["reference_ids"] = self.reference_ids
        ret["query_path"] = query_path
        ret["gt_pose"] = gt_pose
        ret["cost"] = costs[best_ref_id]
        success = bool(ret["success"] and self.t_err < 10 and self.r_err < 10)
        if success:
            self.pose = ret["T_refined"]
        ret["success"] = success
        self.success = success
        refiner.after_lm_enqueued = None
        refiner.lm_camera = None
        self._verify_render_ahead(success)
        img_name = os.path.basename(str(query_path))
        self.pose_history[img_name] = ret
        self.pose_tracker_history[img_name] = trackers[best_ref_id]
        return success

    def calculate_error(self):
        gt_R, gt_T = self.gt_pose.numpy()
        pr_R, pr_T = self.pose.numpy()
        self.t_err = float(np.linalg.norm(gt_T - pr_T) * 100.0)
        self.r_err = float(geodesic_distance_for_rotations(gt_R, pr_R) * 180 / np.pi)
        if hasattr(self, "pbar") and hasattr(self.pbar, "set_description"):
            self.pbar.set_description(f"Translation error: {self.t_err:.2f}cm, Rotation error: {self.r_err:.2f} degrees, "
                                      f"relocalizations: {self.relocalization_count}")

    def get_query_frame_iterator(self, path, max_frames):
        if isinstance(path, (GTFrameIterator, YCBVideoIterator)):
            return path
        it = YCBVideoIterator(object_path=Path(self.object_path), expression=path, ycb_path=self.ycb_root)
        if max_frames is not None and np.isfinite(max_frames):
            it.frames = it.frames[: int(max_frames)]
        return it


def main(argv=None):
    """Reference command line (pixloc_tracker_ycb.py:306-345)."""
    parser = argparse.ArgumentParser()
    parser.add_argument("--object_path", type=Path)
    parser.add_argument("--query", default="7:10")
    parser.add_argument("--out_dir", default="ycb_7")
    parser.add_argument("--frames", type=int, default=None)
    parser.add_argument("--debug", action="store_true", default=False)
    parser.add_argument("--ycb_root", default=None, help="YCB-Video root (default $YCB_ROOT or /data/ycb/)")
    parser.add_argument("--pixloc_pickles", action="store_true")
    args = parser.parse_args(argv)
    obj_path = args.object_path
    eval_path = Path(args.out_dir)
    os.makedirs(eval_path, exist_ok=True)
    tracker = PixLocPoseTrackerYCB(data_path=str(obj_path / "pixtrack/pixsfm/dataset"), eval_path=str(eval_path),
                                   loc_path=str(obj_path / "pixtrack/aug_nerf_sfm"), object_path=obj_path,
                                   debug=args.debug, ycb_root=args.ycb_root)
    tracker.run(args.query, max_frames=args.frames)
    print("Relocalization count: ", tracker.relocalization_count)
    tracker.save_poses(args.pixloc_pickles)
    print("Cache hits: %d, misses: %d" % (tracker.hits, tracker.misses))
    _dump(tracker.pose_tracker_history, os.path.join(tracker.eval_path, "trackers.pkl"), args.pixloc_pickles)
    print("Done")


if __name__ == "__main__":
    main()
