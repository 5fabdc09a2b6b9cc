import os, sys, time
sys.path.insert(0, "/root/repo")
import numpy as np, torch
from pixtrack_amd.pose_trackers.pixloc_tracker_r9 import PixLocPoseTrackerR9
from pixtrack_amd.synthetic import make_tracking_assets, render_query_frames
dev = torch.device("cuda:0")
n = 45
assets = make_tracking_assets(seed=1002, n_frames=n)
res = {}
for ahead in (False, True, False, True):
    tr = PixLocPoseTrackerR9("", "", "", "/tmp", debug=0, device=dev, assets=assets)
    tr.render_ahead = ahead
    frames = render_query_frames(assets, tr.testbed)
    for i in range(5): tr.run_single_frame((f"{i:06d}.png", frames[i]))
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for i in range(5, n): tr.run_single_frame((f"{i:06d}.png", frames[i]))
    torch.cuda.synchronize(); dt = time.perf_counter() - t0
    P = np.stack([np.concatenate([tr.pose_history[f"{i:06d}.png"]["T_refined"].numpy()[0].ravel(), tr.pose_history[f"{i:06d}.png"]["T_refined"].numpy()[1]]) for i in range(n)])
    print("render_ahead=%s: %.1f frames/s, used %d of %d" % (ahead, (n - 5) / dt, tr.renders_ahead_used, n))
    res.setdefault(ahead, P)
print("pose histories identical:", np.array_equal(res[False], res[True]))
