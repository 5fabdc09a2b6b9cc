"""In-kernel timeline of the LAST convolution of a UNet pass (the 640x480 decoder layer 128 -> 32 with the fused fine
head): every conv kernel of a -DPXT_EXP_STAMPS=1 library writes its workgroups' s_memtime stamps into one buffer, so
after a pass the buffer holds the last layer's.  PIXTRACK_HIP_LIB=.../libpxt_stamps.so python scripts/unet_last_layer_stamps.py [n_img]"""
import ctypes, sys
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import numpy as np, torch
from pixtrack_amd import _lib
from pixtrack_amd.unet import UNet, make_synthetic_unet_weights

dev = torch.device("cuda:0")
net = UNet(make_synthetic_unet_weights(7), dev)
H, W = 480, 640
n_img = int(sys.argv[1]) if len(sys.argv) > 1 else 1
imgs = [torch.rand(H, W, 3, device=dev) * 255 for _ in range(n_img)]
for _ in range(4):
    if n_img == 1:
        net.forward_packed(imgs[0], None, True)
    else:
        net.forward_packed_batch([(im, None, True) for im in imgs])
    torch.cuda.synchronize()
L = _lib.lib()
fn = L.pxt_debug_read_stamps
fn.argtypes = [ctypes.c_void_p, ctypes.c_int64]
st = np.zeros((8192, 16), np.uint64)
assert fn(st.ctypes.data, st.nbytes) == 0
n = (H // 16) * (W // 16)
st = st[:n].astype(np.int64)
t0 = st[:, 0].min()
def q(v): return "min %7d  med %7d  max %7d" % (v.min(), np.median(v), v.max())
print(f"{n} workgroups of the last layer; kernel span {st[:, 3].max() - t0} ticks")
print("start       ", q(st[:, 0] - t0))
print("prologue    ", q(st[:, 1] - st[:, 0]))
prev = st[:, 1]
for c in range(4):
    print(f"chunk {c:2d}    ", q(st[:, 4 + c] - prev)); prev = st[:, 4 + c]
print("epilogue    ", q(st[:, 3] - st[:, 2]))
print("total per wg", q(st[:, 3] - st[:, 0]))
rt = st[:, 15] - st[:, 14]
print("realtime (100 MHz) per wg", q(rt), " -> ticks per us: %.1f" % np.median((st[:, 3] - st[:, 0]) / (rt / 100.0)))
print("kernel span by realtime: %.2f us" % ((st[:, 15].max() - st[:, 14].min()) / 100.0))
order = np.argsort(st[:, 0])
print("start times of the 1st / 300th / 600th / 900th / last workgroup (us):",
      [round(float((st[order[i], 14] - st[:, 14].min()) / 100.0), 1) for i in (0, 299, 599, 899, n - 1)])
