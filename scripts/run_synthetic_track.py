"""Track a synthetic sequence end to end on the GPU and report accuracy vs GT + stage times."""
import sys, time, argparse
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import numpy as np, torch
from pixtrack_amd.synthetic import make_tracking_assets, render_query_frames
from pixtrack_amd.pose_trackers.pixloc_tracker_r9 import PixLocPoseTrackerR9
from pixtrack_amd.utils.pose_utils import geodesic_distance_for_rotations

def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--frames", type=int, default=20)
    ap.add_argument("--width", type=int, default=640)
    ap.add_argument("--height", type=int, default=480)
    ap.add_argument("--seed", type=int, default=1002)
    args = ap.parse_args()
    dev = torch.device("cuda:0")
    t0 = time.time()
    assets = make_tracking_assets(seed=args.seed, width=args.width, height=args.height, n_frames=args.frames)
    print(f"assets: {time.time()-t0:.1f}s", flush=True)
    tr = PixLocPoseTrackerR9("", "", "", "/tmp", debug=1, device=dev, assets=assets)
    t0 = time.time()
    frames = render_query_frames(assets, tr.testbed)
    torch.cuda.synchronize()
    print(f"query frames: {time.time()-t0:.1f}s", flush=True)
    names = [f"{i:06d}.png" for i in range(len(frames))]
    tr.pbar = None
    times = []
    for i, fr in enumerate(zip(names, frames)):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        tr.run_single_frame(fr)
        torch.cuda.synchronize(); times.append(time.perf_counter() - t0)
        ret = tr.pose_history[names[i]]
        Rg, tg = assets["gt_poses"][i]
        if ret["success"]:
            Rr, tt = ret["T_refined"].numpy()
            rot = geodesic_distance_for_rotations(Rr, Rg); tra = float(np.linalg.norm(tt - tg))
        else:
            rot = tra = float("nan")
        its = [r.iters for r in tr.localizer.refiner.last_lm]
        print(f"frame {i}: ok={tr.success} cost={ret['cost']:.5f} thr={tr.cost_threshold:.5f} rot_err={rot:.5f} rad "
              f"trans_err={tra:.5f} iters={its} refs={tr.reference_ids} {times[-1]*1e3:.1f} ms", flush=True)
    print(f"mean frame time (excluding first): {np.mean(times[1:])*1e3:.2f} ms -> {1/np.mean(times[1:]):.1f} fps")

if __name__ == "__main__":
    main()
