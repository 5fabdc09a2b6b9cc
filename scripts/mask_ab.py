import os, sys, subprocess, numpy as np, torch
sys.path.insert(0, "/root/repo")
from pixtrack_amd.ops import ops
dev = torch.device("cuda:0")
def run(depth, ne, nd):
    H, W = depth.shape[:2]
    mask = torch.empty(H, W, dtype=torch.uint8, device=dev); tmp = torch.empty(2 * H * W, dtype=torch.uint8, device=dev)
    ops.depth_mask(depth, ne, nd, mask, tmp); torch.cuda.synchronize(); return mask.cpu().numpy()
rng = np.random.default_rng(int(sys.argv[1]))
outs = []
for (H, W) in ((480, 640), (144, 192), (37, 53), (16, 64), (101, 260)):
    for (ne, nd) in ((1, 5), (0, 0), (2, 2), (1, 0), (0, 3), (3, 5)):
        d = np.zeros((H, W, 4), np.float32)
        blobs = rng.random((H, W)) < 0.02
        d[..., 0] = np.where(rng.random((H, W)) < 0.5, rng.random((H, W)) * 0.01, rng.random((H, W)))
        d[..., 0] *= (rng.random((H, W)) < 0.7)
        cy, cx = H // 2, W // 2
        yy, xx = np.mgrid[:H, :W]
        d[..., 0] = np.where((yy - cy) ** 2 + (xx - cx) ** 2 < (min(H, W) // 3) ** 2, d[..., 0] + 0.3, d[..., 0] * blobs)
        outs.append(run(torch.from_numpy(d).to(dev), ne, nd))
np.save(sys.argv[2], np.concatenate([o.ravel() for o in outs]))
print("sum", sum(int(o.sum()) for o in outs))
