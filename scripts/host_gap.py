"""Host time on the frame's critical path: from the moment the LM record is read (pose known) to the moment the
next frame's first NeRF kernel has been enqueued.  python scripts/host_gap.py"""
import sys, time
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import numpy as np, torch
from pixtrack_amd import ops, optimizer
from pixtrack_amd.pose_trackers.pixloc_tracker_r9 import PixLocPoseTrackerR9
from pixtrack_amd.synthetic import make_tracking_assets, render_query_frames

dev = torch.device("cuda:0")
n = 80
assets = make_tracking_assets(seed=1002, n_frames=n)
tr = PixLocPoseTrackerR9("", "", "", "/tmp", debug=0, device=dev, assets=assets)
frames = render_query_frames(assets, tr.testbed)
names = [f"{i:06d}.png" for i in range(n)]
T = {"res_end": [], "ren_in": [], "ren_out": []}
orig_result = optimizer.PendingLM.result
def result(self):
    r = orig_result(self); T["res_end"].append(time.perf_counter()); return r
optimizer.PendingLM.result = result
orig_render = torch.ops.pixtrack.ngp_render_both
import pixtrack_amd.ngp as ngp_mod
class _W:
    def __call__(self, *a, **k):
        T["ren_in"].append(time.perf_counter()); r = orig_render(*a, **k); T["ren_out"].append(time.perf_counter()); return r
ngp_mod.ops.ngp_render_both = _W()
marks = {}
def wrap(obj, name, key):
    f = getattr(obj, name)
    def g(*a, **k):
        marks.setdefault(key + "_in", []).append(time.perf_counter())
        r = f(*a, **k)
        marks.setdefault(key + "_out", []).append(time.perf_counter())
        return r
    setattr(obj, name, g)
wrap(tr, "refine", "refine")
wrap(tr, "update_reference_ids", "upd")
wrap(tr, "get_mask", "mask")
wrap(tr.testbed, "set_nerf_camera_matrix", "cam")
t0 = time.perf_counter()
for i in range(n):
    tr.run_single_frame((names[i], frames[i]))
torch.cuda.synchronize()
wall = (time.perf_counter() - t0) / n
res, rin, rout = (np.array(T[k]) for k in ("res_end", "ren_in", "ren_out"))
idx = np.searchsorted(res, rin) - 1
ok = idx >= 0
gap = (rin[ok] - res[idx[ok]]) * 1e6
print("frames/s %.1f (%.3f ms); pose known -> render call: med %.1f us (p10 %.1f, p90 %.1f); render call (47 launches): med %.1f us"
      % (1 / wall, wall * 1e3, np.median(gap[10:]), np.percentile(gap[10:], 10), np.percentile(gap[10:], 90), np.median((rout - rin)[10:]) * 1e6))

def seg(a, b):
    A, B = np.array(marks[a] if a in marks else T[a]), np.array(marks[b] if b in marks else T[b])
    idx = np.searchsorted(A, B) - 1
    ok = idx >= 0
    d = (B[ok] - A[idx[ok]]) * 1e6
    return "%6.1f us" % np.median(d[10:])
print("result end -> refine returns ", seg("res_end", "refine_out"))
print("refine returns -> refs updated", seg("refine_out", "upd_out"))
print("refs updated -> next get_mask ", seg("upd_out", "mask_in"))
print("get_mask in -> camera set     ", seg("mask_in", "cam_out"))
print("camera set -> render op call  ", seg("cam_out", "ren_in"))
