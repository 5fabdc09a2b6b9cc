"""Do renders keep their bits beside another HIP stream's kernels?  (profiles/r06_experiments.md section 8.)

Round 6 found that they did not: with UNet passes - more precisely, the third convolution kernel - running on another stream,
about one render in six came back with a few 4 x 2 pixel blocks changed.  This is the two-stream harness that reproduced it
in seconds and then served every experiment of the hunt: per iteration one piece of neighbour work on stream B and four
renders on stream A, every output plane compared with the render made alone.

    python scripts/race_harness.py                        # neighbour: an eight-image UNet batch pass
    DBG_MODE=conv DBG_CONV=240,320,128,128 python ...     # neighbour: one convolution layer (H, W, Cin, Cout), 12 launches
    DBG_MODE=matmul | copy | render                       # an fp16 8192^3 library GEMM / 256-MB copies / nothing
    DBG_RMODE=0|1|2  DBG_PAIR=1  DBG_FROM_SLOT=1          # Shade / Depth / both; the Depth + Shade pair chain; camera from the slot
    DBG_SERIAL=1  DBG_ONE_STREAM=1  DBG_SYNC_EACH=1       # no overlap / one stream / a stream synchronisation after every render
    DBG_NOTRACK=1                                         # no tracker frames first: one camera in the whole process
    DBG_ITERS=150  DBG_SHOW=4                             # iterations; how many differing renders to print in detail
"""
import math
import os
import sys
from pathlib import Path

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import numpy as np, torch
from pixtrack_amd import parallel
from pixtrack_amd.pose_trackers.pixloc_tracker_r9 import PixLocPoseTrackerR9
from pixtrack_amd.synthetic import make_tracking_assets, render_query_frames
dev = torch.device("cuda:0")
objs = parallel.load_object_configs()
W, H = 640, 480
assets = make_tracking_assets(seed=1002, width=W, height=H, n_frames=4, aabb=objs[0]["aabb"])
tr = PixLocPoseTrackerR9("", "", "", "/tmp", debug=0, device=dev, assets=assets)
if os.environ.get("DBG_NOTRACK"):
    # no render before the harness' own: ONE camera ever exists in this process
    frames = [(torch.rand(H, W, 3, device=dev) * 255).round() for _ in range(4)]
    from pixtrack_amd.geometry import Pose
    tr.pose = Pose.from_Rt(*assets["gt_poses"][0])
    tr.camera = tr.get_query_camera(("x", frames[0]))
else:
    frames = render_query_frames(assets, tr.testbed, first_frame_sigma=24.0)
    for i in range(3): tr.run_single_frame((f"{i:06d}.png", frames[i]))
torch.cuda.synchronize()
tb = tr.testbed
model = tr.localizer.extractor.model
mask = tr.localizer.refiner.query_mask if not os.environ.get('DBG_NOTRACK') else torch.ones(H, W, dtype=torch.uint8, device=dev)
ref_u8 = frames[1].clamp(0, 255).to(torch.uint8).contiguous()
items = [(ref_u8, None, False), (frames[2].contiguous(), mask, True)] * 4
tb.set_nerf_camera_matrix(np.asarray(tr._nerf_pose(tr.pose))[:3, :])
w, h, fov = tr._frame_views()[0]; tb.fov = fov
mode = os.environ.get("DBG_MODE", "unet")
sA, sB = torch.cuda.Stream(), torch.cuda.Stream()
if os.environ.get("DBG_ONE_STREAM"): sB = sA
A16 = torch.randn(8192, 8192, device=dev, dtype=torch.float16)
from pixtrack_amd.ops import ops
if mode.startswith("conv"):
    ch, cw_, cin, cout = [int(x) for x in os.environ["DBG_CONV"].split(",")]
    cx = torch.randn(ch, cw_, cin, device=dev, dtype=torch.float16); cw = (torch.randn(cout, 3, 3, cin, device=dev) * 0.05).half()
    cb = torch.zeros(cout, device=dev); co = torch.empty(ch, cw_, cout, device=dev, dtype=torch.float16)
big1 = torch.randn(64 << 20, device=dev); big2 = torch.empty_like(big1)
rmode = int(os.environ.get("DBG_RMODE", "2"))
import math
fov_r = math.degrees(2 * math.atan(960 / (2 * 750.0)))
def render():
    if os.environ.get("DBG_PAIR"):
        nz, u8 = tb.render_frame_pair_device((w, h, fov), (960, 720, fov_r), 8)
        return {"depth_nz": nz.clone(), "rgb_u8": u8.clone()}
    o = tb.render_frame_device(w, h, 8, mode=rmode, want_float=True, from_slot=bool(os.environ.get('DBG_FROM_SLOT')))
    return {k: v.clone() for k, v in o.items() if v is not None}
with torch.cuda.stream(sA):
    ref = render()
torch.cuda.synchronize()
bad = 0; total = 0; shown = 0
tb.stats_accum = torch.zeros(4, dtype=torch.int64, device=dev)
for it in range(int(os.environ.get("DBG_ITERS", "150"))):
    if mode == "unet":
        with torch.cuda.stream(sB):
            model.set_batch_plan(True); outs = model.forward_packed_batch(items); model.set_batch_plan(False)
    elif mode.startswith("conv"):
        with torch.cuda.stream(sB):
            for _ in range(int(os.environ.get("DBG_CONV_REPS", "12"))): ops.conv3x3_nhwc_f16(cx, cw, cb, True, co)
    elif mode == "matmul":
        with torch.cuda.stream(sB):
            for _ in range(3): mm = A16 @ A16
    elif mode == "copy":
        with torch.cuda.stream(sB):
            for _ in range(40): big2.copy_(big1)
    elif mode == "render":
        pass
    res = []
    if os.environ.get("DBG_SERIAL"):  # the renders wait for the aggressor: no overlap, still two streams
        ev = torch.cuda.Event(); ev.record(sB); sA.wait_event(ev)
    with torch.cuda.stream(sA):
        for k in range(4):
            res.append(render())
            if os.environ.get("DBG_SYNC_EACH"): sA.synchronize()
    torch.cuda.synchronize()
    for r in res:
        total += 1
        d = {k: int((r[k] != ref[k]).reshape(r[k].shape[0], r[k].shape[1], -1).any(-1).sum()) for k in r if r[k] is not None and ref[k] is not None}
        if any(d.values()):
            bad += 1
            if shown < int(os.environ.get('DBG_SHOW', '0')):
                shown += 1
                k0 = "rgba" if "rgba" in r else list(r)[0]
                dm = (r[k0] != ref[k0]).reshape(H, W, -1).any(-1); ys, xs = torch.nonzero(dm, as_tuple=True)
                print("iteration", it, "pixels differing per plane", d, "rows", sorted(set(ys.tolist()))[:6], "x", int(xs.min()), int(xs.max()))
                for y, x in list(zip(ys.tolist(), xs.tolist()))[:6]:
                    print("     ", y, x, [round(v, 6) for v in r[k0][y, x].tolist()], [round(v, 6) for v in ref[k0][y, x].tolist()], "depth", float(r["depth"][y, x].reshape(-1)[0]) if "depth" in r else None, float(ref["depth"][y, x].reshape(-1)[0]) if "depth" in ref else None)
print("mode", mode, "renders differing from the first:", bad, "of", total, "keys", list(ref))
