"""Micro-benchmark of pxt_lm_refine_batch: K copies of one problem (own buffers), time per call vs K and grid per problem.
    python scripts/bench_lm_batch.py [N] [K list, e.g. 1,8] [grid list, e.g. 0,32]
    (PXT_LM_BATCH_MAP=1: contiguous workgroups per problem; PXT_LM_XCD_LOCAL=1: the K = 8 exchange through one XCD's L2)"""
import sys
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import torch
from pixtrack_amd import _lib
from pixtrack_amd.optimizer import LevelPack, PixTrackOptimizer, cstride_for
from pixtrack_amd.synthetic import make_lm_scene


def l2n(x, dim):
    return x / x.norm(dim=dim, keepdim=True).clamp_min(1e-12)


class Ref:
    pass


def main():
    dev = torch.device("cuda:0")
    W, H = 640, 480
    N = int(sys.argv[1]) if len(sys.argv) > 1 else 2341
    lam = [10.0 ** (-6 + torch.sigmoid(torch.full((6,), -2.0)) * 11) for _ in range(3)]
    n_ws = int(_lib.lib().pxt_lm_workspace_bytes())
    probs = []
    for k in range(8):
        sc = make_lm_scene(seed=1001 + k, width=W, height=H, n_points=N, sigma_px=2.0)
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
        r = Ref()
        r.p3d = torch.from_numpy(sc.p3d).float().to(dev)
        r.valid = None
        probs.append({"ref": r, "packs": packs, "T_init": sc.T_init, "workspace": torch.zeros(n_ws, dtype=torch.uint8, device=dev),
                      "camera": None})
    bws = torch.zeros(int(_lib.lib().pxt_lm_batch_workspace_bytes(8)), dtype=torch.uint8, device=dev)
    for stops in ("default", "never"):
        for K in ([int(x) for x in sys.argv[2].split(",")] if len(sys.argv) > 2 else (1, 2, 4, 8)):
            for grid in ([int(x) for x in sys.argv[3].split(",")] if len(sys.argv) > 3 else (0, 16, 24, 32, 40, 48, 64)):
                conf = dict(num_iters=150 if stops == "default" else 40, pad=1, n_workgroups=grid)
                if stops == "never":
                    conf.update(grad_stop_criteria=0.0, dt_stop_criteria=0.0, dR_stop_criteria=0.0)
                nc = PixTrackOptimizer(conf).native_conf()
                for _ in range(3):
                    hs = PixTrackOptimizer.refine_levels_batch(probs[:K], nc, bws)
                    res = [h.result() for h in hs]
                torch.cuda.synchronize()
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                reps = 10
                e0.record()
                for _ in range(reps):
                    PixTrackOptimizer.refine_levels_batch(probs[:K], nc, bws, want_log=False)
                e1.record()
                torch.cuda.synchronize()
                ms = e0.elapsed_time(e1) / reps
                it = max(r.total_iters for r in res)
                print(f"stops={stops} K={K} grid/problem={grid:3d} iters(max)={it:3d} {ms*1e3:9.1f} us/call "
                      f"{ms*1e3/K:8.1f} us/problem {ms*1e3/max(it,1):7.2f} us/iter(max)", flush=True)


if __name__ == "__main__":
    main()
