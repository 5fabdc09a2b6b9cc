"""BASELINE configs[3] on its real workload: every OBJ_AABB of the reference's config/*.sh
(bottle ... spirit_level; values in pixtrack_amd/configs/objects.json) must render and track -
including roncelli_blankk.sh (a slab 0.079 thick in y) and motor_core.sh (y bounds written
max-first).  Small frames (the boxes, not the resolution, are what varies between objects)."""
import numpy as np
import pytest
import torch

from pixtrack_amd import parallel
from pixtrack_amd.pose_trackers.pixloc_tracker_r9 import PixLocPoseTrackerR9
from pixtrack_amd.synthetic import make_tracking_assets, render_query_frames

pytestmark = pytest.mark.gpu

OBJECTS = parallel.load_object_configs()
# (the cold start refines at image scale 4 first: below ~128 x 96 that level is a 16 x 12 image and no point is valid)
# (and the bottle, 0.49 high x 0.18 wide, needs more than 128 x 96 pixels to pin the rotation about its long axis)
SIZES = [(352, 264), (160, 120), (192, 144), (224, 168), (256, 192), (288, 216), (320, 240), (128, 96)]


def test_object_table_is_the_references():
    assert [o["name"] for o in OBJECTS] == ["bottle", "cracker_box", "gimble", "motor_core", "pickle_rick",
                                            "premier_protein", "roncelli_blankk", "spirit_level"]
    for o in OBJECTS:
        lo, hi = np.asarray(o["aabb"][0]), np.asarray(o["aabb"][1])
        assert (lo < hi).all(), o["name"]  # motor_core's swapped y bounds are sorted by the loader
    slab = next(o for o in OBJECTS if o["name"] == "roncelli_blankk")
    assert abs((slab["aabb"][1][1] - slab["aabb"][0][1]) - 0.079) < 1e-9


@pytest.mark.parametrize("k", range(len(OBJECTS)))
def test_every_config_box_renders_and_tracks(device, k):
    obj = OBJECTS[k]
    w, h = SIZES[k]
    n = 6
    assets = make_tracking_assets(seed=1100 + k, width=w, height=h, n_frames=n, aabb=obj["aabb"], n_points=3000)
    assets["aabb"] = obj["OBJ_AABB"]  # the tracker receives the box as config/*.sh writes it ($OBJ_AABB, :85-86)
    tr = PixLocPoseTrackerR9("", "", "", "/tmp", debug=0, device=device, assets=assets)
    tr.spp = 2
    assert (np.asarray(tr.testbed.render_aabb.min) < np.asarray(tr.testbed.render_aabb.max)).all()
    frames = render_query_frames(assets, tr.testbed)
    # the object must be visible in the query frames (a box that clips everything renders background only)
    cover = float((frames[1].sum(-1) > 30).float().mean())
    assert cover > 0.005, (obj["name"], cover)  # (the 0.079-thick slab covers ~2 % of its frames)
    names = [f"{i:06d}.png" for i in range(n)]
    for i in range(n):
        tr.run_single_frame((names[i], frames[i]))
    torch.cuda.synchronize()
    ok = [bool(tr.pose_history[nm].get("success")) for nm in names]
    assert all(ok), (obj["name"], ok)
    m = tr.localizer.refiner.query_mask  # the last frame's silhouette mask: neither empty nor the whole image
    assert m is not None and 0.005 < float((m != 0).float().mean()) < 0.99, obj["name"]
    for i in range(1, n):
        Rr, tt = tr.pose_history[names[i]]["T_refined"].numpy()
        Rg, tg = assets["gt_poses"][i]
        rot = float(np.arccos(np.clip((np.trace(Rr @ Rg.T) - 1) / 2, -1, 1)))
        # (against the SYNTHETIC ground truth, at these small frames and spp 2: the tracker's own gates are the test above;
        # this only catches a track that drifted away - 2 % of the camera distance, 3 degrees)
        assert rot < 5e-2 and float(np.linalg.norm(tt - tg)) < 2e-2 * max(1.5, float(np.linalg.norm(tg))), (obj["name"], i, rot)
