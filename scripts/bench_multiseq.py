"""Throughput mode, outside the headline bench: S independent sequences tracked concurrently on
ONE GPU (one Python thread + one HIP stream per sequence, SURVEY 8e "thread + HIP stream per
unit").  Frames of one sequence stay strictly sequential; the sequences fill each other's gaps
(the host turnaround after each LM, the LM kernel's 64 of 256 CUs, the NeRF's late rounds).
    python scripts/bench_multiseq.py [S] [frames]"""
import sys
import threading
import time
from pathlib import Path

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import torch

from pixtrack_amd import optimizer
from pixtrack_amd.pose_trackers.pixloc_tracker_r9 import PixLocPoseTrackerR9
from pixtrack_amd.synthetic import make_tracking_assets, render_query_frames


def main():
    S = int(sys.argv[1]) if len(sys.argv) > 1 else 2
    steps = int(sys.argv[2]) if len(sys.argv) > 2 else 60
    warm = 5
    dev = torch.device("cuda:0")
    optimizer.PendingLM.poll = S == 1
    seqs = []
    for k in range(S):
        assets = make_tracking_assets(seed=1002 + k, n_frames=warm + steps)
        tr = PixLocPoseTrackerR9("", "", "", "/tmp", debug=0, device=dev, assets=assets)
        frames = render_query_frames(assets, tr.testbed)
        seqs.append((tr, frames, torch.cuda.Stream(device=dev)))
    torch.cuda.synchronize()
    gate = threading.Barrier(S + 1)
    ok = [0] * S

    def work(k):
        tr, frames, stream = seqs[k]
        with torch.cuda.stream(stream):
            for i in range(warm):
                tr.run_single_frame((f"{i:06d}.png", frames[i]))
            stream.synchronize()
            gate.wait()
            for i in range(warm, warm + steps):
                tr.run_single_frame((f"{i:06d}.png", frames[i]))
                ok[k] += int(tr.success)
            stream.synchronize()
        gate.wait()

    threads = [threading.Thread(target=work, args=(k,)) for k in range(S)]
    for t in threads:
        t.start()
    gate.wait()
    t0 = time.perf_counter()
    gate.wait()
    dt = time.perf_counter() - t0
    for t in threads:
        t.join()
    print(f"sequences={S} frames={S * steps} tracked_ok={sum(ok)} {S * steps / dt:.1f} frames/s aggregate "
          f"({dt / steps * 1e3:.3f} ms per frame-slot, {steps / dt:.1f} frames/s per sequence)")


if __name__ == "__main__":
    main()
