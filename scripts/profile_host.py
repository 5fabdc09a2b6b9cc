"""cProfile of the per-frame host path (Python/ctypes overhead around the kernels)."""
import cProfile, pstats, sys, io
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import torch
from pixtrack_amd.pose_trackers.pixloc_tracker_r9 import PixLocPoseTrackerR9
from pixtrack_amd.synthetic import make_tracking_assets, render_query_frames

dev = torch.device("cuda:0")
n = 45
assets = make_tracking_assets(seed=1002, n_frames=n)
tr = PixLocPoseTrackerR9("", "", "", "/tmp", debug=0, device=dev, assets=assets)
frames = render_query_frames(assets, tr.testbed)
names = [f"{i:06d}.png" for i in range(n)]
for i in range(5):
    tr.run_single_frame((names[i], frames[i]))
torch.cuda.synchronize()
pr = cProfile.Profile()
pr.enable()
for i in range(5, n):
    tr.run_single_frame((names[i], frames[i]))
torch.cuda.synchronize()
pr.disable()
s = io.StringIO()
pstats.Stats(pr, stream=s).sort_stats("cumulative").print_stats(38)
print(s.getvalue()[:9000])
