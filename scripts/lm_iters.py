import sys, time
sys.path.insert(0, "/root/repo")
import numpy as np, torch
from pixtrack_amd.pose_trackers.pixloc_tracker_r9 import PixLocPoseTrackerR9
from pixtrack_amd.synthetic import make_tracking_assets, render_query_frames
dev = torch.device("cuda:0")
n = 40
assets = make_tracking_assets(seed=1002, n_frames=n)
tr = PixLocPoseTrackerR9("", "", "", "/tmp", debug=0, device=dev, assets=assets)
frames = render_query_frames(assets, tr.testbed)
its = []
for i in range(n):
    tr.run_single_frame((f"{i:06d}.png", frames[i]))
    its.append([r.iters for r in tr.localizer.refiner.last_lm])
print("iterations per level (coarse->fine), frames 5..:", its[5:15])
a = np.array([x[0] for x in its[5:]])
print("mean iters per level", a.mean(0), "total per frame", a.sum(1).mean())
