"""Generates tests/golden/objects8_160x120.npz (+ objects8_seq12.npz): every render box of the reference's
config/*.sh (values in pixtrack_amd/configs/objects.json) tracked by the CPU ORACLE - VERDICT r5 "Missing #2": until
round 6 only the premier_protein box and the unit cube had oracle fixtures, the other six boxes were checked against
the synthetic ground truth with loose bounds.

Per object (oracle/frame_oracle.track_sequence = the policy of pixtrack/pose_trackers/pixloc_tracker_r9.py:216-275):
    frame 0  cold start from the upright reference pose, image scales [4, 1], no mask
    frame 1  steady: scale [1], query masked by the dilated depth silhouette at frame 0's pose
    frame 2  steady
at 160 x 120 (192 x 144 for the bottle: 0.49 high x 0.18 wide needs the pixels to pin its long axis), spp 2.
The query frames are ORACLE renders at the ground-truth poses + rounded Gaussian noise (sigma 12 on the cold-start
frame, 2 afterwards), so nothing of the fixture comes from the HIP path.

``--seq``: for bottle and roncelli_blankk (the two objects whose lock-step AND solo HIP tracks drifted 0.14 / 0.22 rad
from the synthetic ground truth over 60 steps, profiles/r05_bench_objects8.json) a 12-frame oracle sequence at the same
size: does the ORACLE drift as the HIP path does?  The answer (per-frame errors against ground truth of both paths) is
printed by tests/test_objects8_golden_gpu.py and recorded in DESIGN.md section 6.

``--switch``: roncelli_blankk for 48 frames -> tests/golden/roncelli_switch48.npz: around frame 38 the orbit's rotation
comes nearer to the neighbouring mapping image than to the upright one and update_reference_ids
(pixloc_tracker_r9.py:120-143) moves the reference id - one frame later the refinement runs on ANOTHER image's 3-D
points (the thin slab's large face, seen at a grazing angle).  The reference-switch path had no oracle fixture until
round 6; it is also where the HIP track of this box stalls at full size (profiles/r06_drift_probe_roncelli_blankk.log).

CPU only; one process per object.

    python scripts/make_objects8_golden.py            # the eight 3-frame records
    python scripts/make_objects8_golden.py --seq      # + the two 12-frame sequences
    python scripts/make_objects8_golden.py --switch   # only the 48-frame reference-switch sequence
"""
import multiprocessing as mp
import sys
import time
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
import numpy as np

OUT = ROOT / "tests" / "golden" / "objects8_160x120.npz"
OUT_SEQ = ROOT / "tests" / "golden" / "objects8_seq12.npz"
SPP, N_POINTS, SEED0 = 2, 3000, 1100
SEQ_OBJECTS, SEQ_FRAMES = ("bottle", "roncelli_blankk"), 12
OUT_SWITCH, SWITCH_OBJECT, SWITCH_FRAMES = ROOT / "tests" / "golden" / "roncelli_switch48.npz", "roncelli_blankk", 48


def size_of(name):
    return (192, 144) if name == "bottle" else (160, 120)


def track_object(job):
    k, name, aabb, n_frames, first_sigma = job
    import torch

    torch.set_num_threads(1)
    from oracle import frame_oracle as FO
    from oracle import ngp_oracle as NO
    from pixtrack_amd.synthetic import make_tracking_assets

    W, H = size_of(name)
    t_all = time.time()
    assets = make_tracking_assets(seed=SEED0 + k, width=W, height=H, n_frames=n_frames, aabb=aabb, n_points=N_POINTS)
    ngp = FO.ngp_model(assets["snapshot"])
    qcam = FO.colmap_camera_to_pix(assets["query_camera"])
    rng = np.random.default_rng(SEED0 + k + 177)
    frames = []
    for i, (Rg, tg) in enumerate(assets["gt_poses"]):
        rgba = NO.render(ngp, FO.nerf_view(assets["snapshot"], assets["nerf2sfm"], assets["aabb"], Rg, tg, qcam, 0, SPP))
        u8 = FO.to_u8(rgba).astype(np.float32)
        sigma = first_sigma if i == 0 else 2.0
        frames.append(np.clip(np.rint(u8 + rng.normal(size=u8.shape) * sigma), 0, 255).astype(np.uint8))
    covis = FO.covisibility(assets["model3d"])
    recs = FO.track_sequence(assets, [f.astype(np.float32) for f in frames], spp=SPP, covis=covis)
    out = {"width": W, "height": H, "seed": SEED0 + k, "n_frames": n_frames, "first_sigma": first_sigma,
           "queries": np.stack(frames), "gt_R": np.stack([g[0] for g in assets["gt_poses"]]),
           "gt_t": np.stack([g[1] for g in assets["gt_poses"]])}
    log = []
    for r in recs:
        i = r["frame"]
        out[f"f{i}_success"] = r["success"]
        out[f"f{i}_lm_success"] = r["lm_success"]
        out[f"f{i}_masked"] = r["masked"]
        out[f"f{i}_cost"] = r["cost"]
        out[f"f{i}_cost_threshold"] = r["cost_threshold"]
        out[f"f{i}_iters"] = np.array(r["iters"])
        out[f"f{i}_ref_id"], out[f"f{i}_n_points"] = r["ref_id"], r["n_points"]
        out[f"f{i}_R_start"], out[f"f{i}_t_start"] = r["R_start"], r["t_start"]
        if r["R"] is not None:
            out[f"f{i}_R"], out[f"f{i}_t"] = r["R"], r["t"]
            Rg, tg = assets["gt_poses"][i]
            rot = float(np.arccos(np.clip((np.trace(r["R"] @ Rg.T) - 1) / 2, -1, 1)))
            out[f"f{i}_rot_err_gt"], out[f"f{i}_trans_err_gt"] = rot, float(np.linalg.norm(r["t"] - tg))
        if r["mask"] is not None and n_frames <= SEQ_FRAMES:
            out[f"f{i}_mask_bits"] = np.packbits(r["mask"].astype(np.uint8))
            out[f"f{i}_depth_fragile_count"] = int(FO.fragile_depth_pixels(r["depth_rgba"]).sum())
            out[f"f{i}_depth_u8_nonzero_bits"] = np.packbits((FO.to_u8(r["depth_rgba"])[..., 0] != 0).astype(np.uint8))
            # the sample set of the mask's depth render (spp as tracked): what the HIP render must visit, exactly
            _, st = NO.render(ngp, FO.nerf_view(assets["snapshot"], assets["nerf2sfm"], assets["aabb"], r["R_start"],
                                                r["t_start"], qcam, 1, SPP), return_stats=True)
            out[f"f{i}_depth_samples"], out[f"f{i}_depth_rays_hit"] = st["samples"], st["rays_hit"]
        log.append((i, r["multiscale"], r["masked"], r["lm_success"], r["success"], round(r["cost"], 5),
                    round(out.get(f"f{i}_rot_err_gt", float("nan")), 5), r["iters"], "ref", r["ref_id"], r["n_points"]))
    return name, out, log, round(time.time() - t_all, 1)


def main():
    from pixtrack_amd import parallel

    objs = parallel.load_object_configs()
    if "--switch" in sys.argv:
        k = [o["name"] for o in objs].index(SWITCH_OBJECT)
        name, rec, log, secs = track_object((k, SWITCH_OBJECT, objs[k]["aabb"], SWITCH_FRAMES, 24.0))  # (sigma 24 on the cold start: the gate's margin, as in bench.py --config objects8)
        for line in log:
            print("   ", *line)
        ids = [int(rec[f"f{i}_ref_id"]) for i in range(SWITCH_FRAMES)]
        assert len(set(ids)) >= 2, ids
        # no frame within 5 % of the cost gate: the HIP path's costs differ from the oracle's in the fourth digit (fp16
        # activations), a decision must not hang on that
        assert all(abs(float(rec[f"f{i}_cost"]) / float(rec[f"f{i}_cost_threshold"]) - 1.0) > 0.05 for i in range(1, SWITCH_FRAMES))
        np.savez_compressed(OUT_SWITCH, spp=SPP, n_points=N_POINTS, **{f"{name}/{key}": v for key, v in rec.items()})
        print("wrote", OUT_SWITCH, round(OUT_SWITCH.stat().st_size / 1e6, 2), "MB,", secs, "s; reference ids", ids)
        return
    seq = "--seq" in sys.argv
    jobs = [(k, o["name"], o["aabb"], 3, 12.0) for k, o in enumerate(objs)]
    if seq:
        jobs += [(k, o["name"], o["aabb"], SEQ_FRAMES, 12.0) for k, o in enumerate(objs) if o["name"] in SEQ_OBJECTS]
    with mp.get_context("spawn").Pool(min(len(jobs), 8)) as pool:
        results = pool.map(track_object, jobs, chunksize=1)
    out3, out12 = {"names": np.array([o["name"] for o in objs]), "spp": SPP, "n_points": N_POINTS}, {"spp": SPP, "n_points": N_POINTS}
    for (k, name, _, n_frames, _), (nm, rec, log, secs) in zip(jobs, results):
        dst = out3 if n_frames == 3 else out12
        for key, v in rec.items():
            dst[f"{name}/{key}"] = v
        print(name, n_frames, "frames", secs, "s")
        for line in log:
            print("   ", *line)
        if n_frames == 3:
            assert all(rec[f"f{i}_success"] for i in range(3)), (name, "the oracle must track its three frames")
    np.savez_compressed(OUT, **out3)
    print("wrote", OUT, round(OUT.stat().st_size / 1e6, 2), "MB")
    if seq:
        out12["names"] = np.array(list(SEQ_OBJECTS))
        np.savez_compressed(OUT_SEQ, **out12)
        print("wrote", OUT_SEQ, round(OUT_SEQ.stat().st_size / 1e6, 2), "MB")


if __name__ == "__main__":
    main()
