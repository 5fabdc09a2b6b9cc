#!/usr/bin/env python
"""K objects of config/*.sh on one GPU: lock-step (MultiObjectTracker) against K solo runs, same frames.

    python scripts/bench_multiobj.py [--objects 8] [--steps 30] [--warmup 5] [--lm-grid 0] [--per-image-plan]
                                     [--max-unet-batch 16] [--no-solo] [--phases]
Prints one JSON line: aggregate frames/s of both, per-object frames/s, pose differences, phase times."""
import argparse
import gc
import json
import sys
import time
from pathlib import Path

import numpy as np
import torch

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))

from pixtrack_amd import parallel  # noqa: E402
from pixtrack_amd.pose_trackers.multi_object_tracker import MultiObjectTracker  # noqa: E402
from pixtrack_amd.pose_trackers.pixloc_tracker_r9 import PixLocPoseTrackerR9  # noqa: E402
from pixtrack_amd.synthetic import make_tracking_assets, render_query_frames  # noqa: E402


def poses(tr, names):
    out = []
    for nm in names:
        ret = tr.pose_history[nm]
        ok = parallel.frame_tracked(ret)
        T = ret["T_refined"] if ok else ret["T_init"]
        out.append(np.concatenate([T.as12().double().numpy().reshape(-1), [float(ok)]]))
    return np.stack(out)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--objects", type=int, default=8)
    ap.add_argument("--steps", type=int, default=30)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--width", type=int, default=640)
    ap.add_argument("--height", type=int, default=480)
    ap.add_argument("--lm-grid", type=int, default=0)
    ap.add_argument("--per-image-plan", action="store_true")
    ap.add_argument("--max-unet-batch", type=int, default=16)
    ap.add_argument("--no-solo", action="store_true")
    ap.add_argument("--phases", action="store_true")
    ap.add_argument("--groups", type=int, default=1)
    ap.add_argument("--render-pipelines", type=int, default=-1)
    ap.add_argument("--render-streams", type=int, default=1)
    args = ap.parse_args()
    dev = torch.device("cuda:0")
    objs = parallel.load_object_configs()
    K, n = args.objects, args.warmup + args.steps
    names = [f"{i:06d}.png" for i in range(n)]
    assets = [make_tracking_assets(seed=1002 + k, width=args.width, height=args.height, n_frames=n, aabb=objs[k % 8]["aabb"])
              for k in range(K)]

    def fresh(k):
        return PixLocPoseTrackerR9("", "", "", "/tmp", debug=0, device=dev, assets=assets[k])

    trackers = [fresh(k) for k in range(K)]
    frames = [render_query_frames(assets[k], trackers[k].testbed) for k in range(K)]
    out = {"objects": K, "steps": args.steps, "lm_grid": args.lm_grid, "per_image_plan": args.per_image_plan,
           "max_unet_batch": args.max_unet_batch, "groups": args.groups}
    solo_poses = None
    if not args.no_solo:
        fps, solo_poses = [], []
        for k in range(K):
            tr = fresh(k)
            gc.collect()
            gc.disable()
            for i in range(args.warmup):
                tr.run_single_frame((names[i], frames[k][i]))
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for i in range(args.warmup, n):
                tr.run_single_frame((names[i], frames[k][i]))
            torch.cuda.synchronize()
            fps.append(args.steps / (time.perf_counter() - t0))
            gc.enable()
            solo_poses.append(poses(tr, names))
        out["solo_fps"] = [round(f, 1) for f in fps]
        out["solo_aggregate_fps"] = round(K * args.steps / sum(args.steps / f for f in fps), 1)  # one after the other
    multi = MultiObjectTracker(trackers, lm_workgroups=args.lm_grid, per_image_plan=args.per_image_plan,
                               max_unet_batch=args.max_unet_batch, n_groups=args.groups,
                               render_pipelines=args.render_pipelines, render_streams=args.render_streams)
    gc.collect()
    gc.disable()
    for i in range(args.warmup):
        multi.run_single_frames([(names[i], frames[k][i]) for k in range(K)])
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(args.warmup, n):
        multi.run_single_frames([(names[i], frames[k][i]) for k in range(K)])
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    gc.enable()
    out["lockstep_aggregate_fps"] = round(K * args.steps / dt, 1)
    out["lockstep_ms_per_step"] = round(dt / args.steps * 1e3, 3)
    out["tracked_ok"] = int(sum(poses(tr, names[args.warmup:])[:, 12].sum() for tr in trackers))
    out["renders_ahead_used"] = [int(tr.renders_ahead_used) for tr in trackers]
    if solo_poses is not None:
        d = [float(np.abs(poses(tr, names)[:, :12] - sp[:, :12]).max()) for tr, sp in zip(trackers, solo_poses)]
        out["max_pose_diff_vs_solo"] = [float(f"{x:.3g}") for x in d]
        out["speedup_vs_solo_aggregate"] = round(out["lockstep_aggregate_fps"] / out["solo_aggregate_fps"], 3)
    if args.phases:  # a separate untimed pass: one synchronised step with event marks between the phases
        extra = []
        multi.set_groups(1)
        for tr in trackers:
            tr.render_ahead = False
        for rep in range(4):
            i = n - 1
            multi.timing = {}
            # (re-run the last frame: poses are at their fixed point, the work per step is the same)
            multi.run_single_frames([(names[i], frames[k][i]) for k in range(K)])
            torch.cuda.synchronize()
            t = multi.timing
            order = ["step_begin", "renders_enqueued", "unet_enqueued", "sampling_enqueued", "lm_enqueued", "step_end"]
            extra.append({b: round(t[a][0].elapsed_time(t[b][0]), 3) for a, b in zip(order[:-1], order[1:])})
        multi.timing = None
        out["phase_ms_render_ahead_off"] = extra[-1]
    print(json.dumps(out), flush=True)


if __name__ == "__main__":
    main()
