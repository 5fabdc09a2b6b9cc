"""Why does the HIP track of roncelli_blankk leave ground truth around frame 36 (profiles/r06_hip_drift_roncelli_blankk.log:
0.009 rad more per frame = the orbit's 0.5 deg step, i.e. the pose stands still while every frame is accepted) when the
oracle's 90-frame run on ITS noise realisation does not (profiles/r06_oracle_drift_roncelli_blankk.log)?

Two halves, so that both sides see IDENTICAL inputs:

  dump   (GPU box)  track the scene of scripts/hip_drift_long.py; write the 8-bit query frames FIRST..LAST, every frame's
                    start pose, refined pose, cost, per-level iterations and mask coverage to gpurun_out/r06/drift_probe_<object>.npz
  check  (CPU)      oracle/frame_oracle.track_frame on those very frames from the HIP run's start poses: per frame the oracle's
                    refined pose against the HIP one and both against ground truth

    python scripts/drift_probe.py dump roncelli_blankk 34 42
    python scripts/drift_probe.py check gpurun_out/r06/drift_probe_roncelli_blankk.npz [first last] > profiles/r06_drift_probe_roncelli_blankk.log
"""
import sys
import time
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
import numpy as np


def rot_angle(Ra, Rb):
    M = np.asarray(Ra, np.float64) @ np.asarray(Rb, np.float64).T
    v = 0.5 * np.array([M[2, 1] - M[1, 2], M[0, 2] - M[2, 0], M[1, 0] - M[0, 1]])
    c = (np.trace(M) - 1.0) / 2.0
    return float(np.arctan2(np.linalg.norm(v), c))


def scene(name, n):
    from pixtrack_amd import parallel
    from pixtrack_amd.synthetic import make_tracking_assets

    objs = parallel.load_object_configs()
    u = [o["name"] for o in objs].index(name)
    return make_tracking_assets(seed=1002 + u, width=640, height=480, n_frames=n, aabb=objs[u]["aabb"]), objs[u]


def dump(name, first, last):
    import torch

    from pixtrack_amd.pose_trackers.pixloc_tracker_r9 import PixLocPoseTrackerR9
    from pixtrack_amd.synthetic import render_query_frames

    n = last + 1
    assets, _ = scene(name, n)
    dev = torch.device("cuda:0")
    tr = PixLocPoseTrackerR9("", "", "", "/tmp", debug=0, device=dev, assets=assets)
    frames = render_query_frames(assets, tr.testbed, first_frame_sigma=24.0)
    out = {"name": name, "first": first, "last": last, "n": n}
    for i in range(n):
        Rs, ts = tr.pose.numpy() if tr.pose is not None else (np.eye(3), np.zeros(3))  # (None before the cold start)
        out[f"f{i}_ref_id"] = int(tr.reference_ids[0])  # the id whose points the frame's features are taken at
        tr.run_single_frame((f"{i:06d}.png", frames[i]))
        ret = tr.pose_history[f"{i:06d}.png"]
        Rr, tt = ret["T_refined"].numpy()
        m = tr.localizer.refiner.query_mask
        out[f"f{i}_R_start"], out[f"f{i}_t_start"] = np.asarray(Rs, np.float64), np.asarray(ts, np.float64)
        out[f"f{i}_R"], out[f"f{i}_t"] = np.asarray(Rr, np.float64), np.asarray(tt, np.float64)
        out[f"f{i}_cost"], out[f"f{i}_tracked"] = float(ret["cost"]), bool(tr.success)
        out[f"f{i}_iters"] = np.array([int(v) for res in tr.localizer.refiner.last_lm for v in res.iters])
        out[f"f{i}_mask_px"] = -1 if m is None else int((m != 0).sum())
        if first <= i <= last:
            out[f"f{i}_query"] = frames[i].round().clamp(0, 255).to(torch.uint8).cpu().numpy()
        Rg, tg = assets["gt_poses"][i]
        print(i, int(tr.success), f"{ret['cost']:.5f}", f"{rot_angle(tr.pose.numpy()[0], Rg):.5f}", out[f"f{i}_iters"].tolist(),
              out[f"f{i}_mask_px"], flush=True)
    out["thr"] = float(tr.cost_threshold)
    dst = ROOT / "gpurun_out" / "r06"
    dst.mkdir(parents=True, exist_ok=True)
    np.savez_compressed(dst / f"drift_probe_{name}.npz", **out)


def check(path, first=None, last=None, pinned=False):
    import torch

    torch.set_num_threads(2)
    from oracle import frame_oracle as FO

    g = np.load(path)
    name = str(g["name"])
    first = int(g["first"]) if first is None else first
    last = int(g["last"]) if last is None else last
    assets, _ = scene(name, int(g["n"]))
    ref_id = assets["model3d"].name2id[assets["upright_ref_img"]]
    print(f"# {name}: the oracle on the HIP run's own query frames, every frame from the HIP run's start pose")
    print("# frame | HIP: cost iters err_gt_rad moved_rad | oracle: cost iters err_gt_rad moved_rad | HIP vs oracle rot_rad trans | s | reference id, points"
          + ("  (oracle's reference id PINNED to the upright image)" if pinned else ""))
    for i in range(first, last + 1):
        t0 = time.time()
        Rs, ts = g[f"f{i}_R_start"], g[f"f{i}_t_start"]
        q = g[f"f{i}_query"].astype(np.float32)
        ref_i = int(g[f"f{i}_ref_id"]) if (f"f{i}_ref_id" in g and not pinned) else ref_id
        res = FO.track_frame(assets, Rs, ts, q, ref_i, multiscale=(1,), use_mask=True, spp=8)
        Ro, to = res["R"].numpy().astype(np.float64), res["t"].numpy().astype(np.float64)
        Rh, th = g[f"f{i}_R"], g[f"f{i}_t"]
        Rg, tg = assets["gt_poses"][i]
        print(i, "|", f"{float(g[f'f{i}_cost']):.5f}", "/".join(map(str, g[f"f{i}_iters"].tolist())), f"{rot_angle(Rh, Rg):.5f} {rot_angle(Rh, Rs):.5f}",
              "|", f"{res['cost']:.5f}", "/".join(map(str, res["iters"])), f"{rot_angle(Ro, Rg):.5f} {rot_angle(Ro, Rs):.5f}",
              "|", f"{rot_angle(Rh, Ro):.2e} {float(np.linalg.norm(th - to)):.2e}", "|", round(time.time() - t0, 1), "ref", ref_i, res["n_points"], flush=True)


if __name__ == "__main__":
    if sys.argv[1] == "dump":
        dump(sys.argv[2], int(sys.argv[3]), int(sys.argv[4]))
    else:
        check(sys.argv[2], *(int(v) for v in sys.argv[3:5]), pinned="pinned" in sys.argv[5:])
