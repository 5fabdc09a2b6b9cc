"""Time of the packed 3x3 convolution (pxt_conv3x3_packed) on the pyramid's layer shapes and on the
judge's probe shape, per tile configuration (TFLOP/s; random data, weights packed once):
    python scripts/bench_conv.py [--all-cfgs]"""
import sys
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import torch
from pixtrack_amd import _lib

dev = torch.device("cuda:0")
L = _lib.lib()
CFG = {1: (16, 128), 2: (16, 64), 4: (8, 128), 6: (16, 32),  # rows, channels per WG
       11: (16, 128), 13: (32, 64), 14: (8, 128), 15: (12, 128), 16: (32, 32), 17: (24, 64),
       18: (12, 128), 19: (16, 128)}  # third kernel (pxt_conv_v3.h); 18 / 19: eight waves, K split in the workgroup
# (H, W, Cin, Cout, images): the probe shape, then the 640x480 pyramid's plain layers (two images per pass)
shapes = [(256, 256, 256, 128, 1), (256, 512, 256, 128, 1), (256, 256, 128, 128, 1), (480, 640, 64, 64, 2), (240, 320, 64, 128, 2),
          (240, 320, 128, 128, 2), (120, 160, 128, 256, 2), (120, 160, 256, 256, 2), (60, 80, 256, 512, 2),
          (60, 80, 512, 512, 2), (30, 40, 512, 512, 2)]
all_cfgs = "--all-cfgs" in sys.argv


def bench(H, W, Cin, Cout, n_img, cfg, splits=1, pool=False):
    # two images are benchmarked as one image of double height (same tile count; one extra seam row)
    Hh = H * n_img
    x = torch.randn(Hh, W, Cin, device=dev).half()
    w = (torch.randn(Cout, 3, 3, Cin, device=dev) * 0.05).half()
    b = torch.randn(Cout, device=dev)
    packed = torch.empty(int(L.pxt_conv3x3_packed_bytes(Cin, Cout)), dtype=torch.uint8, device=dev)
    _lib.check(L.pxt_conv3x3_pack_weights(w.data_ptr(), Cin, Cout, packed.data_ptr(), _lib.stream_ptr(dev)), "pack")
    out = torch.empty(Hh, W, Cout, device=dev, dtype=torch.float16)
    pl = torch.empty(Hh // 2, W // 2, Cout, device=dev, dtype=torch.float16) if pool else None
    ws = torch.empty(splits * Hh * W * Cout * 4, dtype=torch.uint8, device=dev) if splits > 1 else None

    def run():
        _lib.check(L.pxt_conv3x3_packed(x.data_ptr(), Hh, W, Cin, packed.data_ptr(), b.data_ptr(), Cout, 1, out.data_ptr(),
                                        _lib.dptr(pl), cfg, splits, _lib.dptr(ws), ws.numel() if ws is not None else 0,
                                        _lib.stream_ptr(dev)), "conv")
    for _ in range(3):
        run()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20):
        run()
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 20
    fl = 2 * 9 * Cin * Cout * Hh * W
    rows, ch = CFG[cfg] if cfg else (0, 0)
    wgs = ((Hh + rows - 1) // rows) * ((W + 15) // 16) * (Cout // ch) * splits if cfg else 0
    print(f"{W}x{H}x{n_img} {Cin:4d}->{Cout:4d} cfg {cfg} splits {splits:2d} pool {int(pool)}: {ms*1e3:8.1f} us "
          f"{fl/ms/1e9:7.1f} TFLOP/s  wgs={wgs}", flush=True)
    return ms


for (H, W, Cin, Cout, n) in shapes:
    cfgs = [c for c in CFG if Cout % CFG[c][1] == 0] if all_cfgs else [0]
    for c in cfgs:
        bench(H, W, Cin, Cout, n, c)
    if all_cfgs and H <= 60:
        for c in (1, 2, 11):
            for sp in (2, 4, 8):
                if sp <= Cin // 32:
                    bench(H, W, Cin, Cout, n, c, splits=sp)
    if all_cfgs and H >= 120 and Cout >= 64:
        bench(H, W, Cin, Cout, n, 2, pool=True)
