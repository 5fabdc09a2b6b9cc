"""Two-render tracking rate (mask and reference rendered separately), alone and with another tracker's contexts
(and their streams) alive in the process.  python scripts/two_renders.py [n_other_trackers]"""
import sys, time
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import torch
from pixtrack_amd.pose_trackers.pixloc_tracker_r9 import PixLocPoseTrackerR9
from pixtrack_amd.synthetic import make_tracking_assets, render_query_frames
dev = torch.device("cuda:0")
n = 65
assets = make_tracking_assets(seed=1002, n_frames=n)
others = []
for _ in range(int(sys.argv[1]) if len(sys.argv) > 1 else 0):
    o = PixLocPoseTrackerR9("", "", "", "/tmp", debug=0, device=dev, assets=assets)
    fr = render_query_frames(assets, o.testbed)
    for i in range(6): o.run_single_frame((f"{i:06d}.png", fr[i]))
    others.append(o)
for fuse in (True, False):
    tr = PixLocPoseTrackerR9("", "", "", "/tmp", debug=0, device=dev, assets=assets)
    tr.fuse_identical_views = fuse
    frames = render_query_frames(assets, tr.testbed)
    for i in range(5): tr.run_single_frame((f"{i:06d}.png", frames[i]))
    torch.cuda.synchronize(); t0 = time.perf_counter(); ts = [t0]
    for i in range(5, n):
        tr.run_single_frame((f"{i:06d}.png", frames[i])); ts.append(time.perf_counter())
    torch.cuda.synchronize()
    import numpy as np
    d = np.diff(ts) * 1e3
    print("other trackers %d, fused views %s: %.1f frames/s; frame ms p10 %.2f p50 %.2f p90 %.2f max %.2f" % (
        len(others), fuse, (n - 5) / (time.perf_counter() - t0), np.percentile(d, 10), np.percentile(d, 50), np.percentile(d, 90), d.max()))
