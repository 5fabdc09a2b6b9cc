"""Generates tests/golden/sequence_640x480.npz: FIVE consecutive frames of BASELINE configs[1] (640x480, spp 8)
tracked by the CPU oracle's restatement of the per-frame policy (oracle/frame_oracle.track_sequence,
reference pixtrack/pose_trackers/pixloc_tracker_r9.py:216-275):

    frame 0  cold start, image scales [4, 1], no mask; freezes the cost threshold at 1.1 x its cost
    frame 1  steady: scale [1], query masked by the dilated depth silhouette of frame 0's pose
    frame 2  steady
    frame 3  a deliberately BAD query (uniform noise): the LM runs, the cost gate rejects the frame,
             the pose is NOT updated and the success flag drops (:258-268)
    frame 4  the frame after a failure: unmasked, from frame 2's pose

CPU only (~14 full-size numpy renders).  Everything is seeded; the query frames are ORACLE renders at the
ground-truth poses + rounded Gaussian noise.  tests/test_sequence_golden_gpu.py runs the same frames through
PixLocPoseTrackerR9.run_single_frame (render-ahead ON) and compares frame by frame.

    python scripts/make_sequence_golden.py
"""
import sys
import time
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
import numpy as np

from oracle import frame_oracle as FO
from oracle import ngp_oracle as NO
from pixtrack_amd.model3d import extract_covisibility
from pixtrack_amd.synthetic import make_tracking_assets

OUT = ROOT / "tests" / "golden" / "sequence_640x480.npz"
SEED, W, H, SPP, N = 1002, 640, 480, 8, 5
BAD = 3


def main():
    t_all = time.time()
    assets = make_tracking_assets(seed=SEED, width=W, height=H, n_frames=N)
    ngp = FO.ngp_model(assets["snapshot"])
    qcam = FO.colmap_camera_to_pix(assets["query_camera"])
    rng = np.random.default_rng(SEED + 177)
    frames = []
    for i, (Rg, tg) in enumerate(assets["gt_poses"]):
        t0 = time.time()
        if i == BAD:
            img = rng.integers(0, 256, size=(H, W, 3)).astype(np.uint8)
        else:
            rgba = NO.render(ngp, FO.nerf_view(assets["snapshot"], assets["nerf2sfm"], assets["aabb"], Rg, tg, qcam, 0, SPP))
            u8 = FO.to_u8(rgba).astype(np.float32)
            sigma = 12.0 if i == 0 else 2.0  # the cold-start frame is the worse observation (synthetic.render_query_frames)
            img = np.clip(np.rint(u8 + rng.normal(size=u8.shape) * sigma), 0, 255).astype(np.uint8)
        frames.append(img)
        print("query", i, round(time.time() - t0, 1), "s", flush=True)
    covis = extract_covisibility(assets["model3d"])
    t0 = time.time()
    recs = FO.track_sequence(assets, [f.astype(np.float32) for f in frames], spp=SPP, covis=covis)
    print("tracked", round(time.time() - t0, 1), "s", flush=True)
    out = {"seed": SEED, "width": W, "height": H, "spp": SPP, "n_frames": N, "bad_frame": BAD,
           "queries": np.stack(frames), "gt_R": np.stack([g[0] for g in assets["gt_poses"]]),
           "gt_t": np.stack([g[1] for g in assets["gt_poses"]])}
    for r in recs:
        i = r["frame"]
        print(i, r["multiscale"], "masked" if r["masked"] else "unmasked", "lm", r["lm_success"], "ok", r["success"],
              "cost", r["cost"], "thr", r["cost_threshold"], r["iters"], flush=True)
        out[f"f{i}_success"] = r["success"]
        out[f"f{i}_lm_success"] = r["lm_success"]
        out[f"f{i}_masked"] = r["masked"]
        out[f"f{i}_multiscale"] = np.array(r["multiscale"])
        out[f"f{i}_cost"] = r["cost"]
        out[f"f{i}_cost_threshold"] = r["cost_threshold"]
        out[f"f{i}_iters"] = np.array(r["iters"])
        out[f"f{i}_R_start"], out[f"f{i}_t_start"] = r["R_start"], r["t_start"]
        if r["R"] is not None:
            out[f"f{i}_R"], out[f"f{i}_t"] = r["R"], r["t"]
        if r["mask"] is not None:
            out[f"f{i}_mask_bits"] = np.packbits(r["mask"].astype(np.uint8))
            fr = FO.fragile_depth_pixels(r["depth_rgba"])
            out[f"f{i}_depth_fragile_count"] = int(fr.sum())
            out[f"f{i}_depth_u8_nonzero_bits"] = np.packbits((FO.to_u8(r["depth_rgba"])[..., 0] != 0).astype(np.uint8))
    assert recs[0]["success"] and recs[1]["success"] and recs[2]["success"], "the sequence must track up to the bad frame"
    assert recs[BAD]["lm_success"] and not recs[BAD]["success"], "the bad frame must be rejected by the COST gate"
    assert not recs[BAD + 1]["masked"] and np.array_equal(recs[BAD + 1]["R_start"], recs[BAD]["R_start"])
    np.savez_compressed(OUT, **out)
    print("wrote", OUT, round(OUT.stat().st_size / 1e6, 2), "MB; total", round(time.time() - t_all, 1), "s")


if __name__ == "__main__":
    main()
