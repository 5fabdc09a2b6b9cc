"""One render's chain in isolation: ms per render of the benchmark object (premier_protein box, 640x480, spp 8) for
  both   Shade + Depth in one march (the headline frame's render),
  pair   a frame's Depth 640x480 + Shade 960x720 (the r9 phone shape) as one chain (render_frame_pair_device),
  two    the same two renders one after the other.
The chain's knobs are environment variables read once per process (PXT_NGP_PIPES, PXT_NGP_G_SHADE / _MARCH / _INIT / _COMPACT,
PXT_NGP_ROUNDS, PXT_NGP_TAIL_GRID): run one process per setting.

    python scripts/bench_render_chain.py [n_views]
"""
import math
import os
import sys
from pathlib import Path

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import numpy as np
import torch

from pixtrack_amd.ngp import Testbed
from pixtrack_amd.synthetic import PREMIER_PROTEIN_AABB, look_at_pose, make_synthetic_nerf


def main():
    dev = torch.device("cuda:0")
    n_views = int(sys.argv[1]) if len(sys.argv) > 1 else 4
    tb = Testbed(device=dev)
    tb.load_snapshot(make_synthetic_nerf(1012, PREMIER_PROTEIN_AABB))
    tb.background_color = [255, 255, 255, 0.0]
    tb.snap_to_pixel_centers = True
    tb.nerf.rendering_min_transmittance = 1e-7
    tb.render_aabb.min, tb.render_aabb.max = PREMIER_PROTEIN_AABB
    lo, hi = np.array(PREMIER_PROTEIN_AABB)
    c, ext = 0.5 * (lo + hi), float(np.max(hi - lo))
    W, H, spp = 640, 480, 8
    fq = 1.2 * W
    fov_q = math.degrees(2 * math.atan(W / (2 * fq)))
    fov_r = math.degrees(2 * math.atan(960 / (2 * 750.0)))
    dist = fq * ext / (0.5 * H)
    tb.stats_accum = torch.zeros(4, dtype=torch.int64, device=dev)
    res = {"both": [], "pair": [], "two": []}
    samples = []
    for v in range(n_views):
        az = 2 * math.pi * v / n_views + 0.3
        d = np.array([math.cos(az), 0.25 * math.sin(3 * az), math.sin(az)])
        eye = c + d / np.linalg.norm(d) * dist
        R, _ = look_at_pose(eye, c, up=np.array([0, 1.0, 0]))
        tb._cam_ngp = np.concatenate([R.T, eye[:, None]], 1)
        tb.fov = fov_q

        def timed(fn, n=20, warm=3):
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

        tb.stats_accum.zero_()
        o = tb.render_frame_device(W, H, spp, mode=2, want_float=True)
        torch.cuda.synchronize()
        samples.append(int(tb.stats_accum[0]))
        if v == 0:  # digests: a change of the renderer that is meant to keep every bit can be checked against an older build
            import hashlib

            nz, u8 = tb.render_frame_pair_device((W, H, fov_q), (960, 720, fov_r), spp)
            torch.cuda.synchronize()
            dg = lambda t: hashlib.sha256(t.detach().cpu().numpy().tobytes()).hexdigest()[:12]
            print("digests", {k: dg(t) for k, t in o.items()}, "pair", dg(nz), dg(u8), "stats", tb.stats_accum.tolist(), flush=True)
        res["both"].append(timed(lambda: tb.render_frame_device(W, H, spp, mode=2)))
        res["pair"].append(timed(lambda: tb.render_frame_pair_device((W, H, fov_q), (960, 720, fov_r), spp)))

        def two():
            tb.fov = fov_q
            tb.render_frame_device(W, H, spp, mode=1)
            tb.fov = fov_r
            tb.render_frame_device(960, 720, spp, mode=0)
            tb.fov = fov_q

        res["two"].append(timed(two))
    knobs = {k: v for k, v in os.environ.items() if k.startswith("PXT_NGP_")}
    print("knobs", knobs, "| samples/render (M)", round(float(np.mean(samples)) / 1e6, 2),
          "| ms per render:", {k: round(float(np.mean(v)), 4) for k, v in res.items()},
          {k: [round(x, 3) for x in v] for k, v in res.items()}, flush=True)


if __name__ == "__main__":
    main()
