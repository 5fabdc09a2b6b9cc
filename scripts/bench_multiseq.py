"""Throughput mode, outside the headline bench: S independent sequences tracked concurrently on
ONE GPU (one Python thread + one HIP stream per sequence, SURVEY 8e "thread + HIP stream per
unit").  Frames of one sequence stay strictly sequential; the sequences fill each other's gaps
(the host turnaround after each LM, the LM kernel's 64 of 256 CUs, the NeRF's late rounds).
    python scripts/bench_multiseq.py [S] [frames] [gil_switch_interval]
    python scripts/bench_multiseq.py --procs S [frames]     # one PROCESS per sequence (no shared interpreter lock)"""
import sys
import threading
import time
from pathlib import Path

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import torch

from pixtrack_amd import optimizer
from pixtrack_amd.pose_trackers.pixloc_tracker_r9 import PixLocPoseTrackerR9
from pixtrack_amd.synthetic import make_tracking_assets, render_query_frames


def one_sequence(k, steps, warm, go_file, seed0=1002):
    """--procs worker: warm up, report ready, wait for the parent's start signal, track, print the interval."""
    import os

    dev = torch.device("cuda:0")
    assets = make_tracking_assets(seed=seed0 + k, n_frames=warm + steps)
    tr = PixLocPoseTrackerR9("", "", "", "/tmp", debug=0, device=dev, assets=assets)
    frames = render_query_frames(assets, tr.testbed)
    for i in range(warm):
        tr.run_single_frame((f"{i:06d}.png", frames[i]))
    torch.cuda.synchronize()
    print("READY", flush=True)
    while not os.path.exists(go_file):
        time.sleep(0.0005)
    t0 = time.time()
    ok = 0
    for i in range(warm, warm + steps):
        tr.run_single_frame((f"{i:06d}.png", frames[i]))
        ok += int(tr.success)
    torch.cuda.synchronize()
    print(f"DONE {t0:.6f} {time.time():.6f} {ok}", flush=True)


def main_procs(S, steps):
    import os
    import subprocess
    import tempfile

    go = os.path.join(tempfile.mkdtemp(), "go")
    procs = [subprocess.Popen([sys.executable, __file__, "--worker", str(k), str(steps), go], stdout=subprocess.PIPE, text=True)
             for k in range(S)]
    for p in procs:
        assert p.stdout.readline().strip() == "READY"
    open(go, "w").close()
    t0s, t1s, oks = [], [], 0
    for p in procs:
        line = p.stdout.readline().split()
        assert line[0] == "DONE", line
        t0s.append(float(line[1])); t1s.append(float(line[2])); oks += int(line[3])
        p.wait()
    dt = max(t1s) - min(t0s)
    print(f"processes={S} frames={S * steps} tracked_ok={oks} {S * steps / dt:.1f} frames/s aggregate "
          f"({steps / dt:.1f} frames/s per sequence; one process per sequence on one GPU)")


def main():
    if len(sys.argv) > 1 and sys.argv[1] == "--worker":
        return one_sequence(int(sys.argv[2]), int(sys.argv[3]), 5, sys.argv[4])
    if len(sys.argv) > 1 and sys.argv[1] == "--procs":
        return main_procs(int(sys.argv[2]), int(sys.argv[3]) if len(sys.argv) > 3 else 60)
    S = int(sys.argv[1]) if len(sys.argv) > 1 else 2
    steps = int(sys.argv[2]) if len(sys.argv) > 2 else 60
    warm = 5
    dev = torch.device("cuda:0")
    # CPython hands the GIL over every 5 ms by default: a thread that needs 20 us to enqueue its next launch then
    # waits behind another thread's Python for milliseconds while its stream runs dry
    sys.setswitchinterval(float(sys.argv[3]) if len(sys.argv) > 3 else 5e-5)
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
