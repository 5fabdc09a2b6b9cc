"""K renders (the eight config/*.sh boxes, 640x480, spp 8, Shade + Depth in one march) one after the other - two pipelines
each (the one-object tracker's setting) or one (the lock-step groups' setting) - against ONE batched chain
(pxt_ngp_render_frame_batch), and two batched chains of K / 2 side by side on two streams (what two lock-step groups do).
PXT_NGP_BATCH_GRID=<n> overrides the per-object grid of the batched grid-stride kernels."""
import math
import sys
from pathlib import Path

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import numpy as np
import torch

from pixtrack_amd import parallel
from pixtrack_amd.ngp import Testbed
from pixtrack_amd.synthetic import look_at_pose, make_synthetic_nerf


def testbed(dev, seed, aabb, W, direction):
    tb = Testbed(device=dev)
    tb.load_snapshot(make_synthetic_nerf(seed))
    tb.background_color = [255, 255, 255, 0.0]
    tb.snap_to_pixel_centers = True
    tb.nerf.rendering_min_transmittance = 1e-7
    tb.render_aabb.min, tb.render_aabb.max = aabb
    tb.fov = math.degrees(2 * math.atan(W / (2 * 1.2 * W)))
    lo, hi = np.array(aabb)
    c, ext = 0.5 * (lo + hi), float(np.linalg.norm(hi - lo))
    d = np.asarray(direction, np.float64)
    eye = c + d / np.linalg.norm(d) * ext * 1.1
    R, _ = look_at_pose(eye, c, up=np.array([0, 1.0, 0]))
    tb._cam_ngp = np.concatenate([R.T, eye[:, None]], 1)
    tb.stats_accum = torch.zeros(4, dtype=torch.int64, device=dev)
    return tb


def timed(fn, n=6, warm=2):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n


def main():
    dev = torch.device("cuda:0")
    W, H, spp = 640, 480, 8
    objs = parallel.load_object_configs()
    dirs = [[0.9, 0.5, 0.3], [0.1, 0.3, 1.0], [-0.8, 0.2, 0.1], [0.3, 0.9, -0.2], [-0.4, -0.5, 0.8], [0.7, -0.1, -0.7],
            [-0.2, 0.6, 0.7], [0.5, 0.5, 0.5]]
    tbs = [testbed(dev, 31 + k, o["aabb"], W, dirs[k % 8]) for k, o in enumerate(objs)]
    for tb in tbs:
        tb.render_frame_device(W, H, spp, mode=2)
    torch.cuda.synchronize()
    samples = [int(tb.stats_accum[0]) for tb in tbs]
    print("samples per render (M):", [round(s / 1e6, 2) for s in samples], "total", round(sum(samples) / 1e6, 2), flush=True)
    side = torch.cuda.Stream(dev)
    for K in (2, 4, 8):
        sub = tbs[:K]
        sizes = [(W, H)] * K
        ws = torch.empty(Testbed.batch_workspace_bytes(K), dtype=torch.uint8, device=dev)
        ws2 = torch.empty(Testbed.batch_workspace_bytes(K), dtype=torch.uint8, device=dev)
        t2 = timed(lambda: [tb.render_frame_device(W, H, spp, mode=2) for tb in sub])
        t1 = timed(lambda: [tb.render_frame_device(W, H, spp, mode=2, pipelines=1) for tb in sub])
        tb_ = timed(lambda: Testbed.render_frame_batch_device(sub, sizes, spp, mode=2, workspace=ws))

        def two_streams():
            ev = torch.cuda.Event()
            ev.record()
            with torch.cuda.stream(side):
                side.wait_event(ev)
                Testbed.render_frame_batch_device(sub[K // 2:], sizes[K // 2:], spp, mode=2, workspace=ws2)
                ej = torch.cuda.Event()
                ej.record(side)
            Testbed.render_frame_batch_device(sub[:K // 2], sizes[:K // 2], spp, mode=2, workspace=ws)
            torch.cuda.current_stream().wait_event(ej)

        def two_streams_single():
            ev = torch.cuda.Event()
            ev.record()
            with torch.cuda.stream(side):
                side.wait_event(ev)
                for tb in sub[K // 2:]:
                    tb.render_frame_device(W, H, spp, mode=2, pipelines=1)
                ej = torch.cuda.Event()
                ej.record(side)
            for tb in sub[:K // 2]:
                tb.render_frame_device(W, H, spp, mode=2, pipelines=1)
            torch.cuda.current_stream().wait_event(ej)

        tss = timed(two_streams_single)
        tbb = timed(two_streams)
        print(f"K={K}: one after the other, 2 pipelines {t2:.3f} ms ({t2 / K:.3f} per render) | 1 pipeline {t1:.3f} ({t1 / K:.3f}) | "
              f"two streams of single renders {tss:.3f} ({tss / K:.3f}) | ONE batched chain {tb_:.3f} ({tb_ / K:.3f}) | "
              f"two batched chains on two streams {tbb:.3f} ({tbb / K:.3f})", flush=True)


if __name__ == "__main__":
    main()
