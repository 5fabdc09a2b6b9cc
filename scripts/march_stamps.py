"""In-kernel timeline of ngp_march_kernel (library built with -DPXT_EXP_STAMPS=1, PIXTRACK_HIP_LIB):
per-wave start / after-ray-load / end stamps of pipeline 0 in one round.  python scripts/march_stamps.py [round]"""
import ctypes, math, sys
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import numpy as np, torch
from pixtrack_amd import _lib
from pixtrack_amd.ngp import Testbed, RenderMode
from pixtrack_amd.synthetic import PREMIER_PROTEIN_AABB, look_at_pose, make_synthetic_nerf

rnd = int(sys.argv[1]) if len(sys.argv) > 1 else 0
dev = torch.device("cuda:0")
tb = Testbed(device=dev); tb.load_snapshot(make_synthetic_nerf(11))
tb.background_color = [255, 255, 255, 0.0]; tb.snap_to_pixel_centers = True
tb.nerf.rendering_min_transmittance = 1e-7
tb.render_aabb.min, tb.render_aabb.max = PREMIER_PROTEIN_AABB
lo, hi = np.array(PREMIER_PROTEIN_AABB); c = 0.5 * (lo + hi)
eye = c + np.array([0.9, 0.5, 0.3]) / np.linalg.norm([0.9, 0.5, 0.3]) * 1.69
R, _ = look_at_pose(eye, c, up=np.array([0, 1.0, 0]))
tb._cam_ngp = np.concatenate([R.T, eye[:, None]], 1)
W, H = 640, 480
tb.fov = math.degrees(2 * math.atan(W / (2 * 1.2 * W)))
tb.render_mode = RenderMode.Shade
L = _lib.lib()
fn = L.pxt_debug_ngp_stamps
fn.argtypes = [ctypes.c_void_p, ctypes.c_int64, ctypes.c_int32]
assert fn(None, 0, rnd) == 0
for _ in range(4):
    tb.render_both_device(W, H, 8)
torch.cuda.synchronize()
st = np.zeros((16384, 8), np.uint64)
assert fn(st.ctypes.data, st.nbytes, rnd) == 0
st = st.astype(np.int64)
used = st[:, 5] > 0
print("round", rnd, ": waves launched", int((st[:, 0] > 0).sum()), "waves with rays", int(used.sum()), "iterations max", int(st[:, 5].max()))
t0 = st[st[:, 0] > 0, 0].min()
a = st[used]
span = (st[st[:, 0] > 0, 4].max() - t0) / 100.0
print("kernel span %.1f us (realtime)" % span)
dur = (a[:, 4] - a[:, 0]) / 100.0
print("wave duration us: min %.1f med %.1f p90 %.1f max %.1f" % (dur.min(), np.median(dur), np.percentile(dur, 90), dur.max()))
print("wave start us: med %.1f p90 %.1f max %.1f" % tuple(np.percentile((a[:, 0] - t0) / 100.0, [50, 90, 100])))
print("wave end us: med %.1f p90 %.1f p99 %.1f max %.1f" % tuple(np.percentile((a[:, 4] - t0) / 100.0, [50, 90, 99, 100])))
ld = a[:, 2] - a[:, 1]; tot = a[:, 3] - a[:, 1]
print("ticks: ray load med %d p90 %d ; total med %d p90 %d max %d" % (np.median(ld), np.percentile(ld, 90), np.median(tot), np.percentile(tot, 90), tot.max()))
hist, edges = np.histogram((a[:, 4] - t0) / 100.0, bins=12)
print("end-time histogram:", list(zip(np.round(edges[:-1]).astype(int), hist)))
hops, incs = a[:, 6], a[:, 7] + 0
print("max-lane hops per wave: med %d p90 %d p99 %d max %d ; dt increments: med %d p90 %d max %d" % (
    np.median(hops), np.percentile(hops, 90), np.percentile(hops, 99), hops.max(), np.median(incs), np.percentile(incs, 90), incs.max()))
cc = np.corrcoef(hops, tot)[0, 1]
fit = np.polyfit(hops, tot, 1)
print("ticks vs hops: corr %.3f, ticks ~= %.0f * hops + %.0f" % (cc, fit[0], fit[1]))
slow = np.argsort(tot)[-8:]
print("slowest waves (ticks, hops, incs):", [(int(tot[i]), int(hops[i]), int(incs[i])) for i in slow])
