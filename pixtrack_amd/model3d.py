"""pixloc ``Model3D`` equivalent: the COLMAP model wrapper pixtrack reads through
``localizer.model3d`` (pixtrack/localization/pixloc_pose_refiners.py:41,151-153,212,335-341;
pixtrack/pose_trackers/pixloc_tracker_r9.py:78,100-102,126,148).  SURVEY.md Appendix A.6."""
from __future__ import annotations

from collections import defaultdict
from typing import Dict, List, Optional


from .utils.colmap import read_model


class Model3D:
    def __init__(self, path=None, model=None):
        cameras, images, points3D = read_model(path) if model is None else model
        self.cameras, self.dbs, self.points3D = cameras, images, points3D
        self.name2id = {im.name: i for i, im in self.dbs.items()}

    def get_p3did_to_dbids(self, dbids: List[int], loc: Optional[Dict] = None, inliers=None,
                           point_selection: str = "all", min_track_length: int = 3) -> Dict[int, set]:
        """3-D point id -> observing dbids among ``dbids``, for points whose track is long enough."""
        if point_selection != "all":
            raise NotImplementedError("pixtrack uses point_selection='all' (pixloc_tracker_r9.py:53)")
        out = defaultdict(set)
        for dbid in dbids:
            ids = self.dbs[dbid].point3D_ids
            for p3did in ids[ids != -1]:
                out[int(p3did)].add(dbid)
        return {i: v for i, v in out.items() if len(self.points3D[i].image_ids) >= min_track_length}

    def get_dbid_to_p3dids(self, p3did_to_dbids: Dict[int, set]) -> Dict[int, List[int]]:
        out = defaultdict(list)
        for p3id, obs in p3did_to_dbids.items():
            for dbid in obs:
                out[dbid].append(p3id)
        return dict(out)


def extract_covisibility(model3d: "Model3D") -> Dict[int, Dict[int, int]]:
    """{image_id: {other_image_id: n_shared_points}} (reference pixtrack/utils/hloc_utils.py:28-47)."""
    covis = {}
    for iid, im in model3d.dbs.items():
        counts = defaultdict(int)
        ids = im.point3D_ids
        for pid in ids[ids != -1]:
            for other in model3d.points3D[int(pid)].image_ids:
                if int(other) != iid:
                    counts[int(other)] += 1
        covis[iid] = dict(counts)
    return covis
