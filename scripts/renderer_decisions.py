"""DESIGN.md section 4's table: the measured effect of every recall-level decision of the NeRF renderer's
specification (oracle/ngp_oracle.py), on the benchmark's view (BASELINE configs[1]: 640 x 480, spp 8, the synthetic
premier_protein-style object at the ground-truth pose of frame 0).  CPU only; the oracle's rows are dealt to
$PXT_ORACLE_PROCS processes (bit-identical to the serial render).

    PXT_ORACLE_PROCS=8 python scripts/renderer_decisions.py            # ~3 min on 8 cores

For each decision the view is rendered both ways (Shade and Depth) and the script reports the max / mean absolute
difference of the RGBA image, of the 8-bit reference image the UNet receives, and the number of mask bits that flip
(get_mask: uint8 depth != 0, erode, 5 x dilate)."""
import sys
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
import numpy as np

from oracle import frame_oracle as FO
from oracle import ngp_oracle as NO
from pixtrack_amd.synthetic import make_tracking_assets

W, H, SPP = 640, 480, 8


def renders(assets, k1=0.0, linear_colors=False):
    ngp = FO.ngp_model(assets["snapshot"])
    ngp.linear_colors = linear_colors
    R, t = assets["gt_poses"][0]
    qcam = FO.colmap_camera_to_pix(assets["query_camera"])
    out = []
    for mode in (0, 1):
        v = FO.nerf_view(assets["snapshot"], assets["nerf2sfm"], assets["aabb"], R, t, qcam, mode, SPP)
        v.k1 = k1
        out.append(NO.render(ngp, v, return_stats=True))
    (rgba, st), (depth, _) = out
    return rgba, depth, st


def compare(name, base, var, note=""):
    (r0, d0, s0), (r1, d1, s1) = base, var
    d = np.abs(r0 - r1)
    u0, u1 = FO.to_u8(r0).astype(np.int32), FO.to_u8(r1).astype(np.int32)
    du = np.abs(u0 - u1)
    m0, m1 = FO.depth_mask(d0), FO.depth_mask(d1)
    dd = np.abs(d0[..., 0] - d1[..., 0])
    print(f"| {name} | {d.max():.3g} / {d.mean():.3g} | {du.max()} / {du.mean():.3g} | {dd.max():.3g} / {dd.mean():.3g} | "
          f"{int((m0 != m1).sum())} of {m0.size} | {s0['samples']} / {s1['samples']} | {note} |", flush=True)


def main():
    assets = make_tracking_assets(seed=1002, width=W, height=H, n_frames=2)
    print("| decision (specification -> alternative) | RGBA max / mean | 8-bit reference image max / mean (grey levels) | "
          "depth max / mean | mask bits flipped | samples | note |")
    print("|---|---|---|---|---|---|---|")
    base = renders(assets)
    compare("srgb_to_linear on finished rays (on -> off: a snapshot trained in linear colours)", base,
            renders(assets, linear_colors=True), "systematic: the reference image the UNet sees changes everywhere")
    NO.VARIANT["mip_dt_factor"] = 128
    compare("mip_from_dt factor (2 x NERF_GRIDSIZE = 256 -> 128, rounds 1-4)", base, renders(assets),
            "one cascade finer wherever t >= 1: fewer samples, thinner silhouette")
    NO.VARIANT["mip_dt_factor"] = 256
    NO.VARIANT["jitter"] = "other"
    compare("start-jitter hash (builder's 24-bit hash -> an unrelated sequence)", base, renders(assets))
    NO.VARIANT["jitter"] = "none"
    compare("start jitter (hash -> none: every pass starts at the box entry)", base, renders(assets))
    NO.VARIANT["jitter"] = "hash"
    NO.VARIANT["max_step_cascades"] = 8
    compare("max_step (from the model's 3 cascades -> instant-ngp's compile-time 8)", base, renders(assets),
            "dt = t / 256 never reaches either bound inside the render box")
    NO.VARIANT["max_step_cascades"] = None
    k1 = 0.0045691  # the SIMPLE_RADIAL k1 colmap2ingp.py:226 quotes
    lens = renders(assets, k1=k1)
    for iters in (4, 16):
        NO.VARIANT["undistort_iters"] = iters
        compare(f"un-distortion iterations with k1 = {k1} (8 -> {iters})", lens, renders(assets, k1=k1))
    NO.VARIANT["undistort_iters"] = 8
    compare(f"(for scale: lens k1 = 0 -> {k1})", base, lens)


if __name__ == "__main__":
    main()
