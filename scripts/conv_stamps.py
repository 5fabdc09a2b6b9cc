"""In-kernel timeline of the packed 3x3 convolution (needs a library built with -DPXT_EXP_STAMPS=1,
selected with PIXTRACK_HIP_LIB): per-workgroup s_memtime stamps at kernel start, after the first halo
chunk is staged, after every 32-channel chunk, after the loop and after the epilogue.
    python scripts/conv_stamps.py H W Cin Cout cfg"""
import ctypes, sys
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import numpy as np, torch
from pixtrack_amd import _lib

H, W, Cin, Cout, cfg = (int(v) for v in sys.argv[1:6])
dev = torch.device("cuda:0")
L = _lib.lib()
x = torch.randn(H, W, Cin, device=dev).half()
w = (torch.randn(Cout, 3, 3, Cin, device=dev) * 0.05).half()
b = torch.randn(Cout, device=dev)
packed = torch.empty(int(L.pxt_conv3x3_packed_bytes(Cin, Cout)), dtype=torch.uint8, device=dev)
_lib.check(L.pxt_conv3x3_pack_weights(w.data_ptr(), Cin, Cout, packed.data_ptr(), _lib.stream_ptr(dev)), "pack")
out = torch.empty(H, W, Cout, device=dev, dtype=torch.float16)
for _ in range(int(sys.argv[6]) if len(sys.argv) > 6 else 5):
    _lib.check(L.pxt_conv3x3_packed(x.data_ptr(), H, W, Cin, packed.data_ptr(), b.data_ptr(), Cout, 1, out.data_ptr(), None,
                                    cfg, 1, None, 0, _lib.stream_ptr(dev)), "conv")
torch.cuda.synchronize()
fn = L.pxt_debug_read_stamps
fn.argtypes = [ctypes.c_void_p, ctypes.c_int64]
st = np.zeros((8192, 16), np.uint64)
assert fn(st.ctypes.data, st.nbytes) == 0
rows = {1: 16, 2: 16, 4: 8, 6: 16, 11: 16, 13: 32, 14: 8, 15: 12, 16: 32}[cfg]
ch = {1: 128, 2: 64, 4: 128, 6: 32, 11: 128, 13: 64, 14: 128, 15: 128, 16: 32}[cfg]
n = ((H + rows - 1) // rows) * ((W + 15) // 16) * (Cout // ch)
st = st[:min(n, 8192)].astype(np.int64)
t0 = st[:, 0].min()
nch = Cin // (16 if cfg in (13, 16) else 32)
print(f"{n} workgroups; stamps in s_memtime ticks (shader clock): kernel span {st[:, 3].max() - t0}")
rel = st - t0
def q(v): return "min %7d  med %7d  max %7d" % (v.min(), np.median(v), v.max())
print("start       ", q(rel[:, 0]))
print("prologue    ", q(st[:, 1] - st[:, 0]))
prev = st[:, 1]
for c in range(min(nch, 11)):
    print(f"chunk {c:2d}    ", q(st[:, 4 + c] - prev)); prev = st[:, 4 + c]
print("epilogue    ", q(st[:, 3] - st[:, 2]))
if st[:, 12].max() > 0:
    print("  barrier   ", q(st[:, 12] - st[:, 2]))
    print("  bias+pack ", q(st[:, 13] - st[:, 12]))
    print("  rest      ", q(st[:, 3] - st[:, 13]))
print("total per wg", q(st[:, 3] - st[:, 0]))
print("end         ", q(rel[:, 3]))
rt = st[:, 15] - st[:, 14]
print("realtime (100 MHz) per wg", q(rt), " -> s_memtime ticks per us: %.1f" % np.median((st[:, 3] - st[:, 0]) / (rt / 100.0)))
print("kernel span by realtime: %.2f us" % ((st[:, 15].max() - st[:, 14].min()) / 100.0))
if nch >= 6:
    mid = (st[:, 4 + 4] - st[:, 4 + 0]) / 4.0  # mean ticks per chunk over chunks 1-4
    idx = np.arange(len(st))
    print("mean chunk 1-4 by wg % 8 (XCD round-robin):", np.round([mid[idx % 8 == k].mean() for k in range(8)]).astype(int))
    print("mean chunk 1-4 by wg // (n/4):", np.round([mid[(idx * 4) // len(st) == k].mean() for k in range(4)]).astype(int))
    print("by (wg // 8) % 8:", np.round([mid[(idx // 8) % 8 == k].mean() for k in range(8)]).astype(int))
    srt = np.argsort(mid)
    print("fastest wgs:", srt[:16], "slowest:", srt[-16:])
