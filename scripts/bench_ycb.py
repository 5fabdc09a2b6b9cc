"""Frames/s of the YCB policy (pixloc_tracker_ycb: mask and reference image are two renders of different cameras, every
frame, no render queued ahead) on the synthetic unit-cube object at 640x480:  python scripts/bench_ycb.py [frames]"""
import sys, time
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import numpy as np, torch
from pixtrack_amd.geometry import Camera, Pose
from pixtrack_amd.pose_trackers import pixloc_tracker_ycb as ycb
from pixtrack_amd.synthetic import CRACKER_BOX_AABB, make_tracking_assets, render_query_frames

n = int(sys.argv[1]) if len(sys.argv) > 1 else 60
dev = torch.device("cuda:0")
assets = make_tracking_assets(seed=1005, width=640, height=480, n_frames=n, aabb=CRACKER_BOX_AABB, reference_scale=0.3, n_points=5600)
tr = ycb.PixLocPoseTrackerYCB("", "", "/tmp", "003_cracker_box", device=dev, assets=assets)
f = float(assets["query_camera"]["params"][0])
cam = Camera.from_colmap(dict(model="OPENCV", width=640, height=480, params=np.array([f, f, 319.5, 239.5])))
frames = render_query_frames(assets, tr.testbed)
gts = [Pose.from_Rt(*p) for p in assets["gt_poses"]]
ok = 0
for i in range(n):
    if i == 10:
        torch.cuda.synchronize(); t0 = time.perf_counter()
    ok += bool(tr.refine((f"{i+1:06d}-color.png", frames[i], gts[i], cam)))
torch.cuda.synchronize(); dt = time.perf_counter() - t0
print(f"ycb policy: {(n - 10) / dt:.1f} frames/s over {n - 10} frames ({ok}/{n} tracked)")
print({k: getattr(tr, k, None) for k in ("renders_ahead_used", "renders_ahead_dropped", "renders_ahead_rejected", "renders_ahead_stale")})
if len(sys.argv) > 2:
    import cProfile, pstats
    pr = cProfile.Profile(); pr.enable()
    for i in range(n - 20, n):
        tr.refine((f"{i+1:06d}-color.png", frames[i], gts[i], cam))
    torch.cuda.synchronize(); pr.disable()
    pstats.Stats(pr).sort_stats("cumulative").print_stats(28)
