import sys
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import numpy as np, torch
from oracle import frame_oracle as FO
from pixtrack_amd.pose_trackers.pixloc_tracker_r9 import PixLocPoseTrackerR9
from pixtrack_amd.synthetic import make_tracking_assets, render_query_frames
from pixtrack_amd.utils.pose_utils import geodesic_distance_for_rotations
dev = torch.device("cuda:0")
assets = make_tracking_assets(seed=1011, width=128, height=96, n_frames=2, n_points=3000)
tr = PixLocPoseTrackerR9("", "", "", "/tmp", debug=1, device=dev, assets=assets)
tr.spp = 2
frames = render_query_frames(assets, tr.testbed)
tr.run_single_frame(("0.png", frames[0]))
lm = tr.localizer.refiner.last_lm
print("HIP iters", [r.iters for r in lm], "costs", [[round(c[0],5) for c in r.costs] for r in lm], [[round(c[-1],5) for c in r.costs] for r in lm])
ref = assets["model3d"].dbs[1]
for ms in ((4,), (4, 1)):
    want = FO.track_frame(assets, ref.qvec2rotmat(), ref.tvec, frames[0].cpu().numpy(), 1, multiscale=ms, use_mask=False, spp=2)
    print("oracle", ms, want["iters"], want["cost"], want["n_points"])
R, t = tr.pose_history["0.png"]["T_refined"].numpy()
print("rot diff", geodesic_distance_for_rotations(R, want["R"].numpy()), np.linalg.norm(t - want["t"].numpy()))
Rg, tg = assets["gt_poses"][0]
print("HIP vs gt", geodesic_distance_for_rotations(R, Rg), "oracle vs gt", geodesic_distance_for_rotations(want["R"].numpy(), Rg))
print("init vs gt", geodesic_distance_for_rotations(ref.qvec2rotmat(), Rg))
for l in range(len(lm)):
    for k in range(3):
        print("HIP scale", l, "lvl", k, [round(float(x), 6) for x in lm[l].log[k, :lm[l].iters[k], 0]], [round(float(x),4) for x in lm[l].log[k, :lm[l].iters[k], 2]], [round(float(x),5) for x in lm[l].log[k, :lm[l].iters[k], 3]])
lg = want["log"]
for i in range(len(lg.costs)):
    print("ORC stage", i, [round(c, 6) for c in lg.costs[i]], [round(c, 4) for c in lg.dR[i]], [round(c, 5) for c in lg.dt[i]])
