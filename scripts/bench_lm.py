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
    W, H, N = 640, 480, 2048
    if len(sys.argv) > 1:
        N = int(sys.argv[1])
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
    for stops in ("default", "never"):
        for grid in (8, 16, 32, 64, 96, 128, 192, 256):
            conf = dict(num_iters=150 if stops == "default" else 40, pad=1, n_workgroups=grid)
            if stops == "never":
                conf.update(grad_stop_criteria=0.0, dt_stop_criteria=0.0, dR_stop_criteria=0.0)
            opt = PixTrackOptimizer(conf)
            nc = opt.native_conf()
            for _ in range(3):
                res = PixTrackOptimizer.refine_levels(p3d, packs, sc.T_init, nc, ws).result()
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            reps = 20
            e0.record()
            for _ in range(reps):
                pend = PixTrackOptimizer.refine_levels(p3d, packs, sc.T_init, nc, ws, want_log=False)
            e1.record()
            torch.cuda.synchronize()
            ms = e0.elapsed_time(e1) / reps
            it = res.total_iters
            print(f"stops={stops} grid={grid:4d} iters={res.iters} total={it} {ms*1e3:9.1f} us/call "
                  f"{ms*1e3/max(it,1):7.2f} us/iter", flush=True)


if __name__ == "__main__":
    main()
