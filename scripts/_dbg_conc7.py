import sys, os, threading
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from pixtrack_amd import optimizer
from pixtrack_amd.pose_trackers.pixloc_tracker_r9 import PixLocPoseTrackerR9
from pixtrack_amd.synthetic import make_tracking_assets, render_query_frames
device = torch.device("cuda:0")
S, n = 3, 24
REP = []
def instrument(tr, k):
    orig = tr._mask_and_reference
    def wrapped(pose, from_slot):
        m, r = orig(pose, from_slot)
        m2, r2 = orig(pose, from_slot)   # the same render again, right behind it
        m3, r3 = orig(pose, from_slot)
        REP.append((k, len(tr.pose_history), r.clone(), r2.clone(), r3.clone(), m.clone(), m2.clone(), m3.clone()))
        return m, r
    tr._mask_and_reference = wrapped
seqs = []
for k in range(S):
    assets = make_tracking_assets(seed=1040 + k, width=320, height=240, n_frames=n, n_points=4000)
    tr = PixLocPoseTrackerR9("", "", "", "/tmp", debug=0, device=device, assets=assets)
    tr.spp = 4; tr.render_ahead = False
    seqs.append((assets, tr, render_query_frames(assets, tr.testbed), torch.cuda.Stream(device=device)))
    instrument(tr, k)
torch.cuda.synchronize()
optimizer.PendingLM.poll = False
def work(k):
    _, tr, frames, stream = seqs[k]
    with torch.cuda.stream(stream):
        for i, f in enumerate(frames):
            tr.run_single_frame((f"{i:06d}.png", f))
        stream.synchronize()
ts = [threading.Thread(target=work, args=(k,)) for k in range(S)]
[t.start() for t in ts]; [t.join() for t in ts]
bad = 0
for k, i, r, r2, r3, m, m2, m3 in REP:
    e12, e13, e23 = torch.equal(r, r2), torch.equal(r, r3), torch.equal(r2, r3)
    if not (e12 and e13):
        bad += 1
        d = (r.int() - (r2 if e23 else r3).int()).abs().sum(-1) > 0
        ys, xs = torch.nonzero(d, as_tuple=True)
        print("seq", k, "frame", i, "three renders of one camera: 1==2", e12, "1==3", e13, "2==3", e23, "| masks equal", torch.equal(m, m2), torch.equal(m, m3),
              "| differing pixels", int(d.sum()), "first", list(zip(xs.tolist(), ys.tolist()))[:4])
print("renders repeated:", len(REP), "disagreeing triples:", bad)
