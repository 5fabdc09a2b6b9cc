import sys, ctypes as C
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import numpy as np, torch
from pixtrack_amd import _lib
from pixtrack_amd.ngp import Testbed, NerfSnapshot
from pixtrack_amd.synthetic import make_synthetic_nerf, PREMIER_PROTEIN_AABB
from oracle import ngp_oracle as NO

dev = torch.device("cuda:0")
base = make_synthetic_nerf(11)
rng = np.random.default_rng(0)
lo, hi = np.array(PREMIER_PROTEIN_AABB); c = 0.5 * (lo + hi); r = 0.3 * (hi - lo)
n = 256
x = (c + rng.uniform(-1, 1, size=(n, 3)) * r).astype(np.float32)
d = rng.normal(size=(n, 3)); d = (d / np.linalg.norm(d, axis=1, keepdims=True)).astype(np.float32)

def run(snap, tag):
    tb = Testbed(device=dev); tb.load_snapshot(snap)
    out = torch.zeros(n, 4, device=dev)
    xd, dd = torch.from_numpy(x).to(dev), torch.from_numpy(d).to(dev)
    _lib.check(_lib.lib().pxt_ngp_query(tb._ctx, xd.data_ptr(), dd.data_ptr(), n, out.data_ptr(), _lib.stream_ptr(dev)), "q")
    torch.cuda.synchronize()
    got = out.cpu().numpy()
    m = NO.NgpModel(grid=snap.grid, mlp=snap.mlp_dict(), occupancy=snap.occupancy, cascades=3, aabb_scale=4.0)
    unit = ((x - np.float32(0.5 - 2.0)) * np.float32(0.25)).astype(np.float32)
    den, rgb = NO.network(m, unit, d)
    ref = np.concatenate([np.log(den)[:, None], rgb], 1)
    err = np.abs(got - ref)
    lanes = np.arange(n) % 64
    print(tag, 'max err logit/r/g/b', err.max(0).round(4), '| lanes<32:', err[lanes < 32].max(0).round(4), '| lanes>=32:', err[lanes >= 32].max(0).round(4))

run(base, 'full      ')
mlp = base.mlp_dict()
def rebuild(**kw):
    m2 = {k: v.copy() for k, v in mlp.items()}
    m2.update(kw)
    return NerfSnapshot(grid=base.grid, mlp=np.concatenate([m2[k].ravel() for k in ('d1', 'd2', 'c1', 'c2', 'c3')]), occupancy=base.occupancy)
c1 = mlp['c1'].copy(); c1[:, 16:] = 0
run(rebuild(c1=c1), 'no-SH     ')
c1 = mlp['c1'].copy(); c1[:, :16] = 0
run(rebuild(c1=c1), 'SH-only   ')
# random dense weights everywhere
rg = np.random.default_rng(5)
rnd = {k: (rg.normal(size=v.shape) * (1.0 / v.shape[1]) ** 0.5).astype(np.float16) for k, v in mlp.items()}
run(NerfSnapshot(grid=base.grid, mlp=np.concatenate([rnd[k].ravel() for k in ('d1', 'd2', 'c1', 'c2', 'c3')]), occupancy=base.occupancy), 'random    ')
