import sys, os, threading, math, hashlib
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from pixtrack_amd.ngp import Testbed
from pixtrack_amd.ops import ops
from pixtrack_amd.synthetic import PREMIER_PROTEIN_AABB, look_at_pose, make_synthetic_nerf
dev = torch.device("cuda:0")
W, H, spp = 320, 240, 4
def mk(seed, d):
    tb = Testbed(device=dev); tb.load_snapshot(make_synthetic_nerf(seed))
    tb.background_color = [255, 255, 255, 0.0]; tb.snap_to_pixel_centers = True
    tb.nerf.rendering_min_transmittance = 1e-7
    tb.render_aabb.min, tb.render_aabb.max = PREMIER_PROTEIN_AABB
    lo, hi = np.array(PREMIER_PROTEIN_AABB); c = 0.5 * (lo + hi)
    eye = c + np.array(d) / np.linalg.norm(d) * 1.7
    R, _ = look_at_pose(eye, c, up=np.array([0, 1.0, 0]))
    tb._cam_ngp = np.concatenate([R.T, eye[:, None]], 1)
    tb.fov = math.degrees(2 * math.atan(W / (2 * 1.2 * W)))
    return tb
tbs = [mk(11 + k, d) for k, d in enumerate(([0.9, 0.5, 0.3], [0.1, 0.3, 1.0], [-0.8, 0.2, 0.1]))]
def one(tb, with_mask):
    o = tb.render_frame_device(W, H, spp, mode=2)
    if with_mask:
        m = torch.empty(H, W, dtype=torch.uint8, device=dev)
        ops.depth_mask_plane(o["depth_nz"], 1, 5, m)
        return o["rgb_u8"], m
    return o["rgb_u8"], o["depth_nz"]
def dg(ts):
    return hashlib.sha1(b"".join(t.cpu().numpy().tobytes() for t in ts)).hexdigest()[:10]
with_mask = len(sys.argv) > 1 and sys.argv[1] == "mask"
want = [dg(one(tb, with_mask)) for tb in tbs]
bad = [0, 0, 0]
def work(k):
    st = torch.cuda.Stream(dev)
    with torch.cuda.stream(st):
        outs = []
        for i in range(150):
            outs.append(one(tbs[k], with_mask))
            if len(outs) == 10:
                st.synchronize()
                bad[k] += sum(dg(o) != want[k] for o in outs)
                outs = []
ts = [threading.Thread(target=work, args=(k,)) for k in range(3)]
[t.start() for t in ts]; [t.join() for t in ts]
print("mask" if with_mask else "render", "coop", os.environ.get("PXT_NGP_COOP"), "mismatching renders per thread (of 150):", bad)
