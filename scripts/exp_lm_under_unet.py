"""Experiment: what would an LM launch cost if it ran BESIDE the UNet's decoder instead of after it?  Times the
two-image UNet pass alone, the LM call alone, one after the other, and the LM call on a second stream released when
the UNet pass is roughly 45 % / 70 % done (approximated by a timed delay kernel on the LM stream)."""
import sys
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import torch
from pixtrack_amd import _lib
from pixtrack_amd.ops import ops
from pixtrack_amd.optimizer import LevelPack, PixTrackOptimizer, cstride_for
from pixtrack_amd.synthetic import make_lm_scene
from pixtrack_amd.unet import OUTPUT_DIMS, UNet, make_synthetic_unet_weights

dev = torch.device("cuda:0")
W, H, N = 640, 480, 2048
sc = make_lm_scene(seed=1001, width=W, height=H, n_points=N, sigma_px=2.0)
l2n = lambda x, dim: x / x.norm(dim=dim, keepdim=True).clamp_min(1e-12)
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
ws_lm = torch.zeros(int(_lib.lib().pxt_lm_workspace_bytes()), dtype=torch.uint8, device=dev)
nc = PixTrackOptimizer(dict(num_iters=150, pad=1)).native_conf()

net = UNet(make_synthetic_unet_weights(7), dev)
a = (torch.rand(H, W, 3, device=dev) * 255).to(torch.uint8)
b = torch.rand(H, W, 3, device=dev) * 255
m = (torch.rand(H, W, device=dev) > 0.3).to(torch.uint8)
shapes = net.level_shapes(H, W)
mk = lambda: [torch.empty(h, w, cstride_for(c), device=dev) for (h, w), c in zip(shapes, OUTPUT_DIMS)]
ws2 = torch.empty(int(_lib.lib().pxt_unet_workspace_bytes_batch(net._ctx, 2, H, W)), dtype=torch.uint8, device=dev)
ctx = int(net._ctx.value)
oa, ob = mk(), mk()
side = torch.cuda.Stream(device=dev)
main = torch.cuda.current_stream(dev)


def unet():
    ops.unet_forward_batch(ctx, [a, b], [None, m], [False, True], oa + ob, ws2)


def lm(levels=packs):
    return PixTrackOptimizer.refine_levels(p3d, levels, sc.T_init, nc, ws_lm, want_log=False)


def serial():
    unet(); lm()


def make_overlapped(delay_us, levels):
    def f():
        side.wait_stream(main)
        with torch.cuda.stream(side):
            if delay_us:
                torch.cuda._sleep(int(delay_us * 2100))  # ~cycles at 2.1 GHz
            lm(levels)
        unet()
        main.wait_stream(side)
    return f


res = lm().result()
print("LM iterations per level (coarse..fine):", res.iters)
cases = [("unet", unet), ("lm all levels", lm), ("lm coarse+mid", lambda: lm(packs[:2])), ("lm fine", lambda: lm(packs[2:])),
         ("unet -> lm (serial)", serial),
         ("unet || lm coarse+mid @0us", make_overlapped(0, packs[:2])),
         ("unet || lm coarse+mid @350us", make_overlapped(350, packs[:2])),
         ("unet || lm coarse+mid @550us", make_overlapped(550, packs[:2])),
         ("unet || lm all @350us", make_overlapped(350, packs))]
for name, fn in cases + cases[4:6]:
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(30):
        fn()
    e1.record()
    torch.cuda.synchronize()
    print(f"{name:34s} {e0.elapsed_time(e1) / 30 * 1e3:8.1f} us", flush=True)
