"""Full-size (640x480, spp 8) parity of one tracked frame: HIP path vs the CPU frame oracle on
identical inputs (BASELINE.json: 1e-3 rad / 1e-3 units).  Takes a few minutes of host CPU."""
import sys, time
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import numpy as np, torch
from oracle import frame_oracle as FO
from pixtrack_amd.pose_trackers.pixloc_tracker_r9 import PixLocPoseTrackerR9
from pixtrack_amd.synthetic import make_tracking_assets, render_query_frames
from pixtrack_amd.utils.pose_utils import geodesic_distance_for_rotations

torch.set_num_threads(32)
dev = torch.device("cuda:0")
assets = make_tracking_assets(seed=1002, n_frames=3)
tr = PixLocPoseTrackerR9("", "", "", "/tmp", debug=1, device=dev, assets=assets)
frames = render_query_frames(assets, tr.testbed)
tr.run_single_frame(("0.png", frames[0]))
R0, t0 = tr.pose.numpy()
ref_id = tr.reference_ids[0]
tr.run_single_frame(("1.png", frames[1]))
ret = tr.pose_history["1.png"]
t = time.time()
tm = {}
want = FO.track_frame(assets, R0, t0, frames[1].cpu().numpy(), ref_id, multiscale=(1,), use_mask=True, timings=tm)
Rr, tt = ret["T_refined"].numpy()
print("oracle seconds", round(time.time() - t, 1), {k: round(v, 2) for k, v in tm.items()})
print("HIP  : success", ret["success"], "cost", ret["cost"], "iters", [r.iters for r in tr.localizer.refiner.last_lm])
print("ORCL : success", want["success"], "cost", want["cost"], "iters", want["iters"])
print("rot diff [rad]", geodesic_distance_for_rotations(Rr, want["R"].numpy()), "trans diff", float(np.linalg.norm(tt - want["t"].numpy())))
mask = tr.get_mask(torch.no_grad() and __import__("pixtrack_amd.geometry", fromlist=["Pose"]).Pose.from_Rt(R0, t0)).cpu().numpy()
print("mask mismatching pixels", int((mask != want["mask"]).sum()), "of", mask.size)
