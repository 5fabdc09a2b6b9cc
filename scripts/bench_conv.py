"""Time of pxt_conv3x3_nhwc_f16 on chosen shapes (TFLOP/s), for kernel ablations:
    PIXTRACK_HIP_LIB=/path/to/variant.so python scripts/bench_conv.py"""
import sys
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import torch
from pixtrack_amd import _lib

dev = torch.device("cuda:0")
L = _lib.lib()
shapes = [(256, 256, 128, 128), (256, 256, 256, 128), (480, 640, 64, 64), (240, 320, 128, 128), (120, 160, 256, 256),
          (60, 80, 512, 512)]
for (H, W, Cin, Cout) in shapes:
    x = torch.randn(H, W, Cin, device=dev).half()
    w = (torch.randn(Cout, 3, 3, Cin, device=dev) * 0.05).half()
    b = torch.randn(Cout, device=dev)
    out = torch.empty(H, W, Cout, device=dev, dtype=torch.float16)
    def run():
        _lib.check(L.pxt_conv3x3_nhwc_f16(x.data_ptr(), H, W, Cin, w.data_ptr(), b.data_ptr(), Cout, 1, out.data_ptr(),
                                          _lib.stream_ptr(dev)), "conv")
    for _ in range(3): run()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20): run()
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 20
    fl = 2 * 9 * Cin * Cout * H * W
    tiles = ((H + 15) // 16) * ((W + 15) // 16) * (Cout // 64)
    print(f"{W}x{H} {Cin}->{Cout}: {ms*1e3:8.1f} us  {fl/ms/1e9:7.1f} TFLOP/s  blocks={tiles}", flush=True)
