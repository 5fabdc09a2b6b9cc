"""Where a tracked frame's wall time goes besides kernels: GPU-idle turnaround between the LM
result and the next frame's first launch, host time per section (no profiler attached)."""
import sys, time
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import numpy as np, torch
from pixtrack_amd import optimizer as O, ngp as NGP, unet as U, refiner as RF
from pixtrack_amd.pose_trackers.pixloc_tracker_r9 import PixLocPoseTrackerR9
from pixtrack_amd.synthetic import make_tracking_assets, render_query_frames

dev = torch.device("cuda:0")
n = 65
assets = make_tracking_assets(seed=1002, n_frames=n)
tr = PixLocPoseTrackerR9("", "", "", "/tmp", debug=0, device=dev, assets=assets)
frames = render_query_frames(assets, tr.testbed)
names = [f"{i:06d}.png" for i in range(n)]

marks = []  # (label, host_time, event)
def mark(label):
    e = torch.cuda.Event(enable_timing=True); e.record()
    marks.append((label, time.perf_counter(), e))

def wrap(obj, name, label):
    f = getattr(obj, name)
    def g(*a, **k):
        mark(label + ":in")
        r = f(*a, **k)
        mark(label + ":out")
        return r
    setattr(obj, name, g)

wrap(NGP.Testbed, "render_both_device", "render")
wrap(U.UNet, "forward_packed_batch", "unet")
wrap(O.PixTrackOptimizer, "refine_levels", "lm_launch")
wrap(O.PendingLM, "result", "lm_result")
wrap(RF.PoseTrackerRefiner, "interp_sparse_observations", "sample")
wrap(RF.PoseTrackerRefiner, "dense_feature_extraction", "dense")
wrap(RF.PoseTrackerRefiner, "refine_pose_using_features", "rpuf")
wrap(RF.PoseTrackerRefiner, "refine", "refine")

for i in range(5):
    tr.run_single_frame((names[i], frames[i]))
torch.cuda.synchronize()
marks.clear()
t0 = time.perf_counter()
for i in range(5, n):
    mark("frame:in")
    tr.run_single_frame((names[i], frames[i]))
torch.cuda.synchronize()
t1 = time.perf_counter()
nf = n - 5
print(f"frame {1e3*(t1-t0)/nf:.3f} ms")
# consecutive mark pairs: host dt and gpu dt
agg = {}
for (la, ta, ea), (lb, tb, eb) in zip(marks[:-1], marks[1:]):
    key = f"{la} -> {lb}"
    h, g = 1e3 * (tb - ta), ea.elapsed_time(eb)
    a = agg.setdefault(key, [0.0, 0.0, 0])
    a[0] += h; a[1] += g; a[2] += 1
print(f"{'section':46s} {'host ms/frame':>14s} {'gpu ms/frame':>14s}  n/frame")
for k, (h, g, c) in agg.items():
    print(f"{k:46s} {h/nf:14.3f} {g/nf:14.3f}  {c/nf:.1f}")
