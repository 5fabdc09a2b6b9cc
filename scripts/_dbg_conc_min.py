import sys, os, threading
root = sys.argv[2] if len(sys.argv) > 2 else os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, root)
import numpy as np, torch
from pixtrack_amd import optimizer
from pixtrack_amd.pose_trackers.pixloc_tracker_r9 import PixLocPoseTrackerR9
from pixtrack_amd.synthetic import make_tracking_assets, render_query_frames
device = torch.device("cuda:0")
NWG = int(os.environ.get("DBG_LMGRID", "0"))
if NWG:
    optimizer.PixTrackOptimizer.default_conf["n_workgroups"] = NWG
S, n = 3, 24
seqs = []
for k in range(S):
    assets = make_tracking_assets(seed=1040 + k, width=320, height=240, n_frames=n, n_points=4000)
    tr = PixLocPoseTrackerR9("", "", "", "/tmp", debug=0, device=device, assets=assets)
    tr.spp = 4
    seqs.append((assets, tr, render_query_frames(assets, tr.testbed), torch.cuda.Stream(device=device)))
torch.cuda.synchronize()
alone = []
for assets, tr, frames, _ in seqs:
    solo = PixLocPoseTrackerR9("", "", "", "/tmp", debug=0, device=device, assets=assets)
    solo.spp = 4
    for i, f in enumerate(frames):
        solo.run_single_frame((f"{i:06d}.png", f))
    alone.append([solo.pose_history[f"{i:06d}.png"]["T_refined"].numpy() for i in range(n)])
optimizer.PendingLM.poll = False
SYNC = os.environ.get("DBG_SYNC", "")
def sync(): torch.cuda.current_stream().synchronize()
def hook(obj, name, before=False, after=False):
    f = getattr(obj, name)
    def w(*a, **k):
        if before: sync()
        r = f(*a, **k)
        if after: sync()
        return r
    setattr(obj, name, w)
for _, tr, _, _ in seqs:
    if SYNC == "before_render": hook(tr, "_mask_and_reference", before=True)
    if SYNC == "after_render": hook(tr, "_mask_and_reference", after=True)
    if SYNC == "after_sample": hook(tr.localizer.refiner, "interp_sparse_observations", after=True)
    if SYNC == "before_sample": hook(tr.localizer.refiner, "interp_sparse_observations", before=True)
    if SYNC == "frame_end": hook(tr, "run_single_frame", after=True)
    if SYNC == "after_unet": hook(tr.localizer.extractor.model, "forward_packed_batch", after=True)
    if SYNC == "before_unet": hook(tr.localizer.extractor.model, "forward_packed_batch", before=True)
LOCK = threading.Lock()
from pixtrack_amd import _lib as _L
LK = os.environ.get("DBG_LOCK", "")
if LK:
    lib = _L.lib()
    names = [n for n in dir(lib) if n.startswith("pxt_") and (LK == "all" or any(n.startswith("pxt_" + p_) for p_ in LK.split(",")))]
    CL = threading.Lock()
    for nme in names:
        f = getattr(lib, nme)
        def mk(f):
            def w(*a):
                with CL:
                    return f(*a)
            return w
        setattr(lib, nme, mk(f))
    print("locked", len(names), "entry points:", LK)
def work(k):
    _, tr, frames, stream = seqs[k]
    with torch.cuda.stream(stream):
        for i, f in enumerate(frames):
            if SYNC == "hostlock":
                with LOCK:
                    tr.run_single_frame((f"{i:06d}.png", f))
            elif SYNC == "hostlock_sync":
                with LOCK:
                    tr.run_single_frame((f"{i:06d}.png", f))
                    stream.synchronize()
            else:
                tr.run_single_frame((f"{i:06d}.png", f))
        stream.synchronize()
ts = [threading.Thread(target=work, args=(k,)) for k in range(S)]
[t.start() for t in ts]; [t.join() for t in ts]
for k, (assets, tr, frames, _) in enumerate(seqs):
    bad = [i for i in range(n) if not np.array_equal(tr.pose_history[f"{i:06d}.png"]["T_refined"].numpy()[0], alone[k][i][0])]
    mx = max(float(np.abs(tr.pose_history[f"{i:06d}.png"]["T_refined"].numpy()[0] - alone[k][i][0]).max()) for i in range(n))
    print(sys.argv[1], SYNC, LK, "seq", k, "mismatching frames", bad[:5], "max abs R diff %.2e" % mx)
