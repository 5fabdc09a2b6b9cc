"""Micro-benchmark of the fused LM kernel: time per call / per iteration vs grid size."""
import sys, time
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import torch
from pixtrack_amd import _lib
from pixtrack_amd.optimizer import LevelPack, PixTrackOptimizer, cstride_for
from pixtrack_amd.synthetic import make_lm_scene


def l2n(x, dim):
    return x / x.norm(dim=dim, keepdim=True).clamp_min(1e-12)


def main():
    dev = torch.device("cuda:0")
    import os
    W, H, N = 640, 480, 2048
    if len(sys.argv) > 2:
        N = int(sys.argv[2])
    sc = make_lm_scene(seed=1001, width=W, height=H, n_points=N, sigma_px=2.0)
    lam = [10.0 ** (-6 + torch.sigmoid(torch.full((6,), -2.0)) * 11) for _ in range(3)]
    packs = []
    for level in reversed(range(3)):
        fq = sc.feats_query[level]
        Cc = fq.shape[0] - 1
        cs = cstride_for(Cc)
        fmap = torch.zeros(fq.shape[1], fq.shape[2], cs)
        fmap[..., :Cc] = l2n(fq[:-1], 0).permute(1, 2, 0)
        fmap[..., Cc] = fq[-1]
        fr = sc.feats_ref[level]
        fref = torch.zeros(N, cs)
        fref[:, :Cc] = l2n(fr[:, :-1], 1)
        fref[:, Cc] = fr[:, -1]
        packs.append(LevelPack(fmap.to(dev), fref.to(dev), Cc, sc.camera.scale(sc.scales[level]), lam[level]))
    p3d = torch.from_numpy(sc.p3d).float().to(dev)
    ws = torch.zeros(int(_lib.lib().pxt_lm_workspace_bytes()), dtype=torch.uint8, device=dev)
    import ctypes, numpy as np
    grid = int(sys.argv[1]) if len(sys.argv) > 1 else 64
    conf = dict(num_iters=40, pad=1, n_workgroups=grid, grad_stop_criteria=0.0, dt_stop_criteria=0.0, dR_stop_criteria=0.0,
                lm_path=int(os.environ.get("PXT_LM_PATH", "0")))
    nc = PixTrackOptimizer(conf).native_conf()
    for want_log in (True, False):
        for _ in range(3):
            res = PixTrackOptimizer.refine_levels(p3d, packs, sc.T_init, nc, ws, want_log=want_log).result()
        torch.cuda.synchronize()
        fn = _lib.lib().pxt_debug_lm_stamps
        fn.argtypes = [ctypes.c_void_p, ctypes.c_int64]
        st = np.zeros((256, 16), np.uint64)
        assert fn(st.ctypes.data, st.nbytes) == 0
        st = st[:res.total_iters].astype(np.int64)
        names = ["accumulate + park", "publish stores landed", "atomic arrive returned", "spin until all arrived", "all-gather of partials",
                 "fold partials", "solve + update (+ log)"]
        print(f"grid {grid}, log to pinned host: {want_log}; s_memtime ticks per iteration, by level (coarse, mid, fine):")
        for k in range(7):
            d = st[:, k + 1] - st[:, k]
            print("  %-26s %s" % (names[k], [int(np.median(d[l * 40 + 2:(l + 1) * 40])) for l in range(3)]))
        for nm, a, b in (("  acc: start -> p3d landed", 0, 8), ("  acc: -> texels landed", 8, 9), ("  acc: -> reduced + normal eq.", 9, 10), ("  acc: rest of rounds", 10, 11)):
            d = st[:, b] - st[:, a]
            print("  %-30s %s" % (nm, [int(np.median(d[l * 40 + 2:(l + 1) * 40])) for l in range(3)]))
        print("  rounds (wave 0):", [int(np.median(st[l * 40 + 2:(l + 1) * 40, 12])) for l in range(3)])
        tot = st[1:, 0] - st[:-1, 0]
        print("  %-26s %s" % ("iteration", [int(np.median(tot[l * 40 + 2:(l + 1) * 40 - 1])) for l in range(3)]))


if __name__ == "__main__":
    main()
