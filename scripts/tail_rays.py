"""Rays left for the straggler kernel (after the wavefront rounds) along the synthetic orbit, and the render's time:
    python scripts/tail_rays.py [frame ...]"""
import sys, time
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import numpy as np, torch
from pixtrack_amd.geometry import Camera
from pixtrack_amd.pose_trackers.pixloc_tracker_r9 import PixLocPoseTrackerR9
from pixtrack_amd.synthetic import make_tracking_assets
from pixtrack_amd.utils.ingp_utils import sfm_to_nerf_pose
from pixtrack_amd.visualization.run_vis_on_poses import get_nerf_image_device

dev = torch.device("cuda:0")
assets = make_tracking_assets(seed=1002, n_frames=330)
tr = PixLocPoseTrackerR9("", "", "", "/tmp", debug=0, device=dev, assets=assets)
cam = Camera.from_colmap(assets["query_camera"])
tb = tr.testbed
for i in ([int(v) for v in sys.argv[1:]] or (0, 30, 60, 100, 150, 200, 250, 300)):
    Rg, tg = assets["gt_poses"][i]
    wIc = np.eye(4); wIc[:3, :3], wIc[:3, 3] = Rg, tg
    nerf_pose = sfm_to_nerf_pose(assets["nerf2sfm"], np.linalg.inv(wIc))
    import math
    w, h = (int(v) for v in cam.size)
    tb.fov = math.atan(w / (float(cam.f[0]) * 2)) * 2 * 180 / np.pi
    tb.set_nerf_camera_matrix(np.asarray(nerf_pose)[:3, :])
    out = tb.render_device(w, h, 8, True, collect_stats=True)
    torch.cuda.synchronize()
    st = tb.read_stats()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(5):
        tb.render_both_device(w, h, 8)
    e1.record(); torch.cuda.synchronize()
    print(f"frame {i:3d}: samples {st['samples']:8d} rays_hit {st['rays_hit']:8d} tail_rays(first render) {st['tail_rays']:7d}  "
          f"render_both {e0.elapsed_time(e1)/5:.3f} ms", flush=True)
