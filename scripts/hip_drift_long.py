"""The HIP tracker on the scenes of scripts/oracle_drift_long.py (same assets, same ground-truth orbit, the bench's noise
levels: sigma 24 on the cold-start frame, 2 afterwards; query frames = HIP renders at the ground-truth poses): one line per
frame, rotation / translation error against ground truth - to be read beside profiles/r06_oracle_drift_<object>.log.

    python scripts/hip_drift_long.py roncelli_blankk 90 > profiles/r06_hip_drift_roncelli_blankk.log
"""
import sys
from pathlib import Path

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import numpy as np
import torch

from pixtrack_amd import parallel
from pixtrack_amd.pose_trackers.pixloc_tracker_r9 import PixLocPoseTrackerR9
from pixtrack_amd.synthetic import make_tracking_assets, render_query_frames


def main():
    name, n = sys.argv[1], int(sys.argv[2])
    # a third argument: another seed of render_query_frames' noise generator (the bench's is 5), or "numpy" = the noise
    # realisation of scripts/oracle_drift_long.py's own run
    noise = sys.argv[3] if len(sys.argv) > 3 else "5"
    objs = parallel.load_object_configs()
    u = [o["name"] for o in objs].index(name)
    dev = torch.device("cuda:0")
    assets = make_tracking_assets(seed=1002 + u, width=640, height=480, n_frames=n, aabb=objs[u]["aabb"])
    tr = PixLocPoseTrackerR9("", "", "", "/tmp", debug=0, device=dev, assets=assets)
    if noise == "numpy":
        clean = render_query_frames(assets, tr.testbed, noise_sigma=0.0, first_frame_sigma=0.0)
        rng = np.random.default_rng(1002 + u + 177)
        frames = [torch.from_numpy(np.clip(np.rint(c.cpu().numpy() + rng.normal(size=tuple(c.shape)) * (24.0 if i == 0 else 2.0)), 0, 255)
                                   .astype(np.float32)).to(dev) for i, c in enumerate(clean)]
    else:
        frames = render_query_frames(assets, tr.testbed, first_frame_sigma=24.0, seed=int(noise))
    print(f"# {name} box {objs[u]['aabb']} 640x480 spp 8 frames {n} (HIP path, HIP-rendered queries, noise realisation: {noise})")
    print("# frame lm_ok tracked cost thr rot_err_gt_rad trans_err_gt iters ref_id n_points")
    for i in range(n):
        ref_used = int(tr.reference_ids[0])  # (the frame's features are taken at this id's points; the id moves afterwards)
        tr.run_single_frame((f"{i:06d}.png", frames[i]))
        ret = tr.pose_history[f"{i:06d}.png"]
        R, t = tr.pose.numpy()
        Rg, tg = assets["gt_poses"][i]
        rot = float(np.arccos(np.clip((np.trace(R @ Rg.T) - 1) / 2, -1, 1)))
        print(i, int(bool(ret["success"])), int(bool(tr.success)), f"{ret['cost']:.5f} {tr.cost_threshold:.5f} {rot:.5f} {float(np.linalg.norm(t - tg)):.5f}",
              "/".join(str(int(v)) for res in tr.localizer.refiner.last_lm for v in res.iters), ref_used,
              int(tr.localizer.refiner._points_of([ref_used])[1].shape[0]), flush=True)


if __name__ == "__main__":
    main()
