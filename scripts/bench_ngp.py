"""Time of one NeRF render (640x480, spp 8) and sample statistics."""
import sys, math
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import numpy as np, torch
from pixtrack_amd.ngp import Testbed, RenderMode
from pixtrack_amd.synthetic import PREMIER_PROTEIN_AABB, look_at_pose, make_synthetic_nerf

def main():
    dev = torch.device("cuda:0")
    snap = make_synthetic_nerf(11)
    tb = Testbed(device=dev); tb.load_snapshot(snap)
    tb.background_color = [255, 255, 255, 0.0]; tb.snap_to_pixel_centers = True
    tb.nerf.rendering_min_transmittance = 1e-7
    tb.render_aabb.min, tb.render_aabb.max = PREMIER_PROTEIN_AABB
    lo, hi = np.array(PREMIER_PROTEIN_AABB); c = 0.5 * (lo + hi)
    import os
    for dist in (1.69,):
        eye = c + np.array([0.9, 0.5, 0.3]) / np.linalg.norm([0.9, 0.5, 0.3]) * dist
        R, _ = look_at_pose(eye, c, up=np.array([0, 1.0, 0]))
        tb._cam_ngp = np.concatenate([R.T, eye[:, None]], 1)
        W, H = 640, 480
        tb.fov = math.degrees(2 * math.atan(W / (2 * 1.2 * W)))
        for mode in (RenderMode.Shade, RenderMode.Depth):
            tb.render_mode = mode
            out = tb.render_device(W, H, 8, True, collect_stats=True)
            st = tb.read_stats()
            for _ in range(2): tb.render_device(W, H, 8, True)
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(5): tb.render_device(W, H, 8, True)
            e1.record(); torch.cuda.synchronize()
            ms = e0.elapsed_time(e1) / 5
            cov = float((out[..., 3] > 0.5).float().mean())
            print(f"dist={dist} mode={mode.name}: {ms:.3f} ms  samples={st['samples']/1e6:.2f}M rays_hit={st['rays_hit']/1e6:.2f}M "
                  f"coverage={cov:.2f} {st['samples']/ms/1e6:.2f} Gsamples/s", flush=True)

if __name__ == "__main__":
    main()
