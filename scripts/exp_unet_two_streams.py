"""Experiment: the frame's two UNet images as ONE batched pass vs TWO single-image passes on two streams."""
import sys
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import torch
from pixtrack_amd.ops import ops
from pixtrack_amd.optimizer import cstride_for
from pixtrack_amd.unet import OUTPUT_DIMS, UNet, make_synthetic_unet_weights

dev = torch.device("cuda:0")
net = UNet(make_synthetic_unet_weights(7), dev)
H, W = 480, 640
a = (torch.rand(H, W, 3, device=dev) * 255).to(torch.uint8)
b = torch.rand(H, W, 3, device=dev) * 255
m = (torch.rand(H, W, device=dev) > 0.3).to(torch.uint8)
shapes = net.level_shapes(H, W)
mk = lambda: [torch.empty(h, w, cstride_for(c), device=dev) for (h, w), c in zip(shapes, OUTPUT_DIMS)]
ws2 = torch.empty(int(__import__("pixtrack_amd")._lib.lib().pxt_unet_workspace_bytes_batch(net._ctx, 2, H, W)), dtype=torch.uint8, device=dev)
wsa, wsb = (torch.empty(int(__import__("pixtrack_amd")._lib.lib().pxt_unet_workspace_bytes_batch(net._ctx, 1, H, W)), dtype=torch.uint8, device=dev) for _ in range(2))
ctx = int(net._ctx.value)
oa, ob = mk(), mk()
side = torch.cuda.Stream(device=dev)

def batched():
    ops.unet_forward_batch(ctx, [a, b], [None, m], [False, True], oa + ob, ws2)

def two_streams():
    main = torch.cuda.current_stream(dev)
    side.wait_stream(main)
    ops.unet_forward_batch(ctx, [a], [None], [False], oa, wsa)
    with torch.cuda.stream(side):
        ops.unet_forward_batch(ctx, [b], [m], [True], ob, wsb)
    main.wait_stream(side)

def sequential():
    ops.unet_forward_batch(ctx, [a], [None], [False], oa, wsa)
    ops.unet_forward_batch(ctx, [b], [m], [True], ob, wsb)

for name, fn in (("batched", batched), ("two streams", two_streams), ("sequential", sequential), ("batched", batched), ("two streams", two_streams)):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(30):
        fn()
    e1.record()
    torch.cuda.synchronize()
    print(f"{name:12s} {e0.elapsed_time(e1) / 30:.3f} ms per image pair", flush=True)
