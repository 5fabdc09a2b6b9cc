import sys, time
sys.path.insert(0, "/root/repo")
import torch
from pixtrack_amd.pose_trackers.pixloc_tracker_r9 import PixLocPoseTrackerR9
from pixtrack_amd.synthetic import make_tracking_assets, render_query_frames
dev = torch.device("cuda:0")
n = 65
assets = make_tracking_assets(seed=1002, n_frames=n)
tr = PixLocPoseTrackerR9("", "", "", "/tmp", debug=0, device=dev, assets=assets)
tr.fuse_identical_views = False
frames = render_query_frames(assets, tr.testbed)
for i in range(5): tr.run_single_frame((f"{i:06d}.png", frames[i]))
torch.cuda.synchronize(); t0 = time.perf_counter()
for i in range(5, n): tr.run_single_frame((f"{i:06d}.png", frames[i]))
torch.cuda.synchronize(); print("two renders: %.1f frames/s" % ((n - 5) / (time.perf_counter() - t0)))
