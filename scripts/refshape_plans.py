"""The UNet pair pass at the reference's own reference-image shapes: per-layer launch plans
(PXT_CONV_DEBUG=1 prints them, once per pass: run this script with that variable set) and the pass's time, for
921x921 || 640x480 (YCB: SfM camera 3072x3072 x 0.3), 960x720 || 640x480 (r9, phone frames), 1024x768 || 640x480
(12-MP stills after the extractor's resize) and the benchmark's 640x480 || 640x480.

    PXT_CONV_DEBUG=1 python scripts/refshape_plans.py > profiles/r04_conv_plans_refshape.log 2>&1"""
import sys
from pathlib import Path

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import torch

from pixtrack_amd.unet import UNet, make_synthetic_unet_weights

dev = torch.device("cuda:0")
net = UNet(make_synthetic_unet_weights(7), dev)
g = torch.Generator().manual_seed(1)
import os
ORDER = ((480, 640), (921, 921), (720, 960), (768, 1024), (480, 640)) if os.environ.get("REV") else ((921, 921), (720, 960), (768, 1024), (480, 640))
for (h0, w0) in ORDER:
    a = (torch.rand(h0, w0, 3, generator=g) * 255).to(torch.uint8).to(dev)   # the reference render (uint8)
    b = (torch.rand(480, 640, 3, generator=g) * 255).to(dev)                  # the query frame (float32)
    m = (torch.rand(480, 640, generator=g) > 0.4).to(torch.uint8).to(dev)
    print(f"==== reference {w0}x{h0} || query 640x480", flush=True)
    sys.stderr.flush()
    net.forward_packed_batch([(a, None, False), (b, m, True)])  # (with PXT_CONV_DEBUG the plans of this pass follow on stderr)
    torch.cuda.synchronize()
    sys.stderr.flush()
    import os
    dbg = os.environ.pop("PXT_CONV_DEBUG", None)  # (the library read it once: the variable only silences nothing; timing below prints too)
    for _ in range(3):
        net.forward_packed_batch([(a, None, False), (b, m, True)])
    torch.cuda.synchronize()
    times = []
    for _ in range(25):  # per-pass events and the median: the caching allocator's occasional hipMalloc is not the pass
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        net.forward_packed_batch([(a, None, False), (b, m, True)])
        e1.record()
        torch.cuda.synchronize()
        times.append(e0.elapsed_time(e1))
    gf = 241.4 * (h0 * w0 + 480 * 640) / (480 * 640)
    ms = sorted(times)[len(times) // 2]
    print(f"pair pass {ms:.3f} ms median of 25 (min {min(times):.3f})  ({gf:.0f} GFLOP -> {gf / ms:.0f} TFLOP/s)", flush=True)
    if dbg is not None:
        os.environ["PXT_CONV_DEBUG"] = dbg
