"""Runs a few two-image UNet passes (the frame's reference render + masked query, as the tracker batches them) so that
rocprofv3 --kernel-trace can record them; scripts/unet_timeline.py then prints one pass's dispatches of BOTH streams:
    cd /tmp && TMPDIR=/tmp rocprofv3 --kernel-trace -d OUT -o ut -- python scripts/unet_pass_timeline.py
    python scripts/unet_timeline.py OUT/ut_results.db"""
import sys
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import torch
from pixtrack_amd.unet import UNet, make_synthetic_unet_weights

dev = torch.device("cuda:0")
net = UNet(make_synthetic_unet_weights(7), dev)
H, W = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (480, 640)
a = torch.rand(H, W, 3, device=dev) * 255
b = torch.rand(H, W, 3, device=dev) * 255
m = (torch.rand(H, W, device=dev) > 0.3).to(torch.uint8)
for _ in range(8):
    net.forward_packed_batch([(a, None, True), (b, m, True)])
    torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(10):
    net.forward_packed_batch([(a, None, True), (b, m, True)])
e1.record()
torch.cuda.synchronize()
print(f"two-image pass: {e0.elapsed_time(e1) / 10:.3f} ms")
# the same pass with the host AHEAD of the device: a 3 ms spin kernel occupies the stream while the host enqueues the
# pass's ~70 launches, so the event pair brackets device time only
ts = []
for _ in range(6):
    torch.cuda._sleep(int(3e-3 * 2.0e9))
    e0.record()
    net.forward_packed_batch([(a, None, True), (b, m, True)])
    e1.record()
    torch.cuda.synchronize()
    ts.append(e0.elapsed_time(e1))
print("two-image pass, host ahead: " + " ".join(f"{t:.3f}" for t in ts) + " ms")
