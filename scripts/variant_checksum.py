"""SHA-256 digests of one NeRF render pair, its depth mask and a two-image UNet pass on seeded inputs: run under
different run-time knobs (PXT_NGP_FUSE_COMPACT_MARCH, PXT_UNET_STREAMS, PXT_MASK_BYTES, PXT_NGP_PIPES) the digests
must not change (tests/test_variants_gpu.py).  python scripts/variant_checksum.py [W H]"""
import hashlib, math, sys
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import numpy as np, torch
from pixtrack_amd.ngp import Testbed
from pixtrack_amd.ops import ops
from pixtrack_amd.synthetic import PREMIER_PROTEIN_AABB, look_at_pose, make_synthetic_nerf
from pixtrack_amd.unet import UNet, make_synthetic_unet_weights

W, H = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (640, 480)
dev = torch.device("cuda:0")
tb = Testbed(device=dev); tb.load_snapshot(make_synthetic_nerf(11))
tb.background_color = [255, 255, 255, 0.0]; tb.snap_to_pixel_centers = True
tb.nerf.rendering_min_transmittance = 1e-7
tb.render_aabb.min, tb.render_aabb.max = PREMIER_PROTEIN_AABB
lo, hi = np.array(PREMIER_PROTEIN_AABB); c = 0.5 * (lo + hi)
eye = c + np.array([0.9, 0.5, 0.3]) / np.linalg.norm([0.9, 0.5, 0.3]) * 1.69
R, _ = look_at_pose(eye, c, up=np.array([0, 1.0, 0]))
tb._cam_ngp = np.concatenate([R.T, eye[:, None]], 1)
tb.fov = math.degrees(2 * math.atan(W / (2 * 1.2 * W)))
rgba, depth = tb.render_both_device(W, H, 8)
mask = torch.empty(H, W, dtype=torch.uint8, device=dev); tmp = torch.empty(2 * H * W, dtype=torch.uint8, device=dev)
ops.depth_mask(depth, 1, 5, mask, tmp)
net = UNet(make_synthetic_unet_weights(7), dev)
g = torch.Generator(device="cpu").manual_seed(5)
a = (torch.rand(H, W, 3, generator=g) * 255).to(torch.uint8).to(dev)
b = (torch.rand(H, W, 3, generator=g) * 255).to(dev)
outs = net.forward_packed_batch([(a, None, False), (b, mask, True)])
torch.cuda.synchronize()
def dg(t): return hashlib.sha256(t.detach().cpu().numpy().tobytes()).hexdigest()[:16]
print("DIGEST rgba", dg(rgba)); print("DIGEST depth", dg(depth)); print("DIGEST mask", dg(mask))
for i, per in enumerate(outs):
    for k, m in enumerate(per): print("DIGEST unet%d_%d" % (i, k), dg(m))
