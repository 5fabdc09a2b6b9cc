import sys, os, threading
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from pixtrack_amd import optimizer
from pixtrack_amd.pose_trackers.pixloc_tracker_r9 import PixLocPoseTrackerR9
from pixtrack_amd.synthetic import make_tracking_assets, render_query_frames
device = torch.device("cuda:0")
S, n = 3, 24
def instrument(tr, store):
    orig = tr._mask_and_reference
    def wrapped(pose, from_slot):
        m, r = orig(pose, from_slot)
        torch.cuda.current_stream().synchronize()
        st = tr.testbed.stats_accum.tolist(); tr.testbed.stats_accum.zero_()
        store[len(tr.pose_history)] = (m.clone(), r.clone(), np.asarray(tr.testbed._cam_ngp, np.float32).copy(), bool(from_slot), st)
        return m, r
    tr._mask_and_reference = wrapped
seqs, stores, sstores = [], [{} for _ in range(S)], [{} for _ in range(S)]
for k in range(S):
    assets = make_tracking_assets(seed=1040 + k, width=320, height=240, n_frames=n, n_points=4000)
    tr = PixLocPoseTrackerR9("", "", "", "/tmp", debug=0, device=device, assets=assets)
    tr.spp = 4; tr.render_ahead = False
    tr.testbed.stats_accum = torch.zeros(4, dtype=torch.int64, device=device)
    seqs.append((assets, tr, render_query_frames(assets, tr.testbed), torch.cuda.Stream(device=device)))
    instrument(tr, stores[k])
torch.cuda.synchronize()
alone = []
for k, (assets, tr, frames, _) in enumerate(seqs):
    solo = PixLocPoseTrackerR9("", "", "", "/tmp", debug=0, device=device, assets=assets)
    solo.spp = 4; solo.render_ahead = False
    solo.testbed.stats_accum = torch.zeros(4, dtype=torch.int64, device=device)
    instrument(solo, sstores[k])
    for i, f in enumerate(frames):
        solo.run_single_frame((f"{i:06d}.png", f))
    alone.append([solo.pose_history[f"{i:06d}.png"]["T_refined"].numpy() for i in range(n)])
optimizer.PendingLM.poll = False
def work(k):
    _, tr, frames, stream = seqs[k]
    with torch.cuda.stream(stream):
        for i, f in enumerate(frames):
            tr.run_single_frame((f"{i:06d}.png", f))
        stream.synchronize()
ts = [threading.Thread(target=work, args=(k,)) for k in range(S)]
[t.start() for t in ts]; [t.join() for t in ts]
for k, (assets, tr, frames, _) in enumerate(seqs):
    badp = [i for i in range(n) if not np.array_equal(tr.pose_history[f"{i:06d}.png"]["T_refined"].numpy()[0], alone[k][i][0])]
    badr, badm, badc = [], [], []
    for i in sorted(stores[k]):
        a, b = stores[k][i], sstores[k].get(i)
        if b is None: continue
        if not torch.equal(a[1], b[1]): badr.append(i)
        if not torch.equal(a[0], b[0]): badm.append(i)
        if not np.array_equal(a[2], b[2]): badc.append(i)
    drops = [(i, stores[k][i][4][1], stores[k][i][4][3]) for i in sorted(stores[k]) if stores[k][i][4][1] != stores[k][i][4][3]]
    sdrops = [(i, sstores[k][i][4][1], sstores[k][i][4][3]) for i in sorted(sstores[k]) if sstores[k][i][4][1] != sstores[k][i][4][3]]
    print("seq", k, "frames where rays in list != rays finished: threads", drops[:4], "solo", sdrops[:4])
    print("seq", k, "first pose mismatch", badp[:1], "| ref image differs at frames", badr[:3], "mask", badm[:3], "host camera", badc[:3])
    if badr:
        i = badr[0]
        d = (stores[k][i][1].int() - sstores[k][i][1].int()).abs()
        # is the threaded image of frame i the solo image of another frame?
        same_as = [j for j in sstores[k] if torch.equal(stores[k][i][1], sstores[k][j][1])]
        ys, xs = torch.nonzero(d.sum(-1) > 0, as_tuple=True)
        for y, x in list(zip(ys.tolist(), xs.tolist()))[:6]:
            prev = sstores[k].get(i - 1)
            print("       px", (x, y), "threaded", stores[k][i][1][y, x].tolist(), "solo", sstores[k][i][1][y, x].tolist(),
                  "solo prev frame", prev[1][y, x].tolist() if prev else None, "mask thr/solo", int(stores[k][i][0][y, x]), int(sstores[k][i][0][y, x]))
        print("     frame", i, "pixels differing", int((d.sum(-1) > 0).sum()), "max", int(d.max()), "| equals the solo image of frames", same_as)
