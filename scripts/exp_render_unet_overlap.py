"""Experiment: does the query image's UNet pass hide under the NeRF render?  Times (a) render alone, (b) one-image
UNet alone, (c) both concurrently on two streams, (d) the frame's device work as shipped (render -> two concurrent
UNet passes) and (e) the alternative (render || query pass -> reference pass)."""
import sys
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import torch
from pixtrack_amd import _lib
from pixtrack_amd.ops import ops
from pixtrack_amd.optimizer import cstride_for
from pixtrack_amd.pose_trackers.pixloc_tracker_r9 import PixLocPoseTrackerR9
from pixtrack_amd.synthetic import make_tracking_assets, render_query_frames
from pixtrack_amd.unet import OUTPUT_DIMS

dev = torch.device("cuda:0")
assets = make_tracking_assets(seed=1002, n_frames=8)
tr = PixLocPoseTrackerR9("", "", "", "/tmp", debug=0, device=dev, assets=assets)
frames = render_query_frames(assets, tr.testbed)
for i in range(6):
    tr.run_single_frame((f"{i:06d}.png", frames[i]))
torch.cuda.synchronize()
tb = tr.testbed
from pixtrack_amd.unet import UNet, make_synthetic_unet_weights
net = UNet(make_synthetic_unet_weights(7), dev)
H, W = 480, 640
a = (torch.rand(H, W, 3, device=dev) * 255).to(torch.uint8)
b = torch.rand(H, W, 3, device=dev) * 255
m = (torch.rand(H, W, device=dev) > 0.3).to(torch.uint8)
shapes = net.level_shapes(H, W)
mk = lambda: [torch.empty(h, w, cstride_for(c), device=dev) for (h, w), c in zip(shapes, OUTPUT_DIMS)]
L = _lib.lib()
ws2 = torch.empty(int(L.pxt_unet_workspace_bytes_batch(net._ctx, 2, H, W)), dtype=torch.uint8, device=dev)
wsa, wsb = (torch.empty(int(L.pxt_unet_workspace_bytes_batch(net._ctx, 1, H, W)), dtype=torch.uint8, device=dev) for _ in range(2))
ctx = int(net._ctx.value)
oa, ob = mk(), mk()
side = torch.cuda.Stream(device=dev)
main = torch.cuda.current_stream(dev)


def render():
    return tb.render_both_device(W, H, 8)


def unet_q():
    ops.unet_forward_batch(ctx, [a], [None], [False], oa, wsa)


def unet_r():
    ops.unet_forward_batch(ctx, [b], [m], [True], ob, wsb)


def both_concurrent():
    side.wait_stream(main)
    with torch.cuda.stream(side):
        unet_q()
    render()
    main.wait_stream(side)


def shipped():
    render()
    ops.unet_forward_batch(ctx, [a, b], [None, m], [False, True], oa + ob, ws2)


def alternative():
    side.wait_stream(main)
    with torch.cuda.stream(side):
        unet_q()
    render()
    unet_r()
    main.wait_stream(side)


for name, fn in (("render", render), ("unet 1 image", unet_q), ("render || unet 1", both_concurrent), ("shipped: render -> 2 passes", shipped),
                 ("alt: render || query pass -> ref pass", alternative), ("shipped", shipped), ("alt", alternative)):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(40):
        fn()
    e1.record()
    torch.cuda.synchronize()
    print(f"{name:40s} {e0.elapsed_time(e1) / 40:.3f} ms", flush=True)
