"""The N > 1 path on CPU: world_size-2 gloo processes shard sequences round-robin, keep no
communication in the loop, and all-gather ragged pose records at the end."""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from pixtrack_amd import parallel
from pixtrack_amd.geometry import Pose


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, ws, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(ws), LOCAL_RANK=str(rank))
    r, w, lr = parallel.init_from_env("gloo")
    assert (r, w) == (rank, ws) and parallel.world() == (rank, ws)
    units = parallel.shard_units(5, rank, ws)
    # a fake per-rank pose history: rank r tracked 3 + r frames of each of its units
    names, hist = [], {}
    for u in units:
        for f in range(3 + rank):
            n = f"u{u}_f{f}"
            names.append(n)
            T = Pose(torch.arange(12, dtype=torch.float64) + 100 * u + f)
            hist[n] = {"success": f % 2 == 0, "T_init": T, "T_refined": T, "cost": 0.5 * u + f}
    rec = parallel.pack_pose_records(hist, names)
    got = parallel.gather_pose_records(rec)
    tmax = parallel.max_over_ranks(float(rank + 1))
    q.put((rank, units, [g.shape for g in got], [g.sum().item() for g in got], rec.sum().item(), tmax))
    dist.barrier()
    dist.destroy_process_group()


def test_shard_units_round_robin():
    assert parallel.shard_units(8, 0, 8) == [0] and parallel.shard_units(8, 7, 8) == [7]
    assert parallel.shard_units(5, 0, 2) == [0, 2, 4] and parallel.shard_units(5, 1, 2) == [1, 3]
    assert sorted(sum((parallel.shard_units(11, r, 4) for r in range(4)), [])) == list(range(11))


def test_single_process_is_a_noop_world():
    rec = torch.zeros(3, parallel.RECORD, dtype=torch.float64)
    assert parallel.world() == (0, 1)
    assert parallel.gather_pose_records(rec)[0] is rec and parallel.max_over_ranks(2.5) == 2.5


@pytest.mark.timeout(120)
def test_two_rank_gloo_gather():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=100) for _ in range(2))
    for p in procs:
        p.join(timeout=30)
        assert p.exitcode == 0
    (r0, u0, shapes0, sums0, own0, t0), (r1, u1, shapes1, sums1, own1, t1) = res
    assert u0 == [0, 2, 4] and u1 == [1, 3]
    assert shapes0 == shapes1 == [torch.Size([9, 14]), torch.Size([8, 14])]  # ragged: 3*3 and 2*4 frames
    assert sums0 == sums1 and abs(sums0[0] - own0) < 1e-9 and abs(sums0[1] - own1) < 1e-9
    assert t0 == t1 == 2.0


def _seg_worker(rank, ws, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(ws), LOCAL_RANK=str(rank))
    parallel.init_from_env("gloo")
    n_frames = 11
    segs_all = [parallel.shard_segments(n_frames, ws, r, 3) for r in range(ws)]
    mine = [i for (a, b) in segs_all[rank] for i in range(a, b)]
    hist = {}
    for i in mine:  # frame i's record carries i, so the stitched video can be checked against the frame index
        T = Pose(torch.full((12,), float(i), dtype=torch.float64))
        hist[f"{i:06d}.png"] = {"success": True, "T_init": T, "T_refined": T, "cost": float(rank)}
    rec = parallel.pack_pose_records(hist, [f"{i:06d}.png" for i in mine])
    video = parallel.stitch_segments(parallel.gather_pose_records(rec), segs_all, n_frames)
    q.put((rank, segs_all[rank], video[:, 0].tolist(), video[:, 13].tolist()))
    dist.barrier()
    dist.destroy_process_group()


def test_shard_segments_cover_the_video_once():
    for n, ws, seg in [(100, 8, 0), (11, 2, 3), (5, 8, 0), (64, 4, 7)]:
        allsegs = sum((parallel.shard_segments(n, ws, r, seg) for r in range(ws)), [])
        frames = sorted(i for a, b in allsegs for i in range(a, b))
        assert frames == list(range(n)), (n, ws, seg)
    assert parallel.shard_segments(0, 4, 0) == []
    assert parallel.shard_segments(100, 8, 0) == [(0, 13)] and parallel.shard_segments(100, 8, 7) == [(91, 100)]


@pytest.mark.timeout(120)
def test_two_rank_segment_sharding_stitches_in_frame_order():
    """BASELINE configs[4]: one video in contiguous per-rank segments, stitched by the final gather."""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_seg_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=100) for _ in range(2))
    for p in procs:
        p.join(timeout=30)
        assert p.exitcode == 0
    assert res[0][1] == [(0, 3), (6, 9)] and res[1][1] == [(3, 6), (9, 11)]
    for _, _, frame_ids, owner in res:
        assert frame_ids == [float(i) for i in range(11)]
        assert owner == [0.0] * 3 + [1.0] * 3 + [0.0] * 3 + [1.0] * 2


def test_object_configs_match_the_reference_shell_files():
    objs = parallel.load_object_configs()
    assert len(objs) == 8 and {o["name"] for o in objs} >= {"premier_protein", "cracker_box", "gimble"}
    pp = next(o for o in objs if o["name"] == "premier_protein")
    assert pp["OBJ_AABB"] == [[0.359, -0.248, 0.047], [0.627, 0.223, 0.574]] and pp["UPRIGHT_REF_IMG"] == "mapping/IMG_2520.png"
    for o in objs:  # boxes come out ordered even where the shell file has them swapped (motor_core)
        assert all(a < b for a, b in zip(*o["aabb"]))


def _objects_worker(rank, ws, port, q):
    """bench.py --config objects8 on N < 8 ranks: rank r carries objects r, r + N, ... of config/*.sh; the per-object pose
    records of a rank travel as one block (objects in the rank's order, frames within an object) through the one gather."""
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(ws), LOCAL_RANK=str(rank))
    parallel.init_from_env("gloo")
    objs = parallel.load_object_configs()
    units = parallel.shard_units(len(objs), rank, ws)
    steps = 3
    recs = []
    for u in units:
        hist = {f"{i:06d}.png": {"success": True, "T_init": Pose(torch.zeros(12, dtype=torch.float64)),
                                  "T_refined": Pose(torch.full((12,), float(10 * u + i), dtype=torch.float64)), "cost": float(u)}
                for i in range(steps)}
        recs.append(parallel.pack_pose_records(hist, sorted(hist)))
    gathered = parallel.gather_pose_records(torch.cat(recs))
    units_all = parallel.gather_objects(units)
    q.put((rank, units, [objs[u]["name"] for u in units], units_all, [g.shape[0] for g in gathered],
           [g[:, 13].tolist() for g in gathered]))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.timeout(120)
@pytest.mark.parametrize("ws", [2, 4])
def test_eight_objects_dealt_to_fewer_ranks(ws):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_objects_worker, args=(r, ws, port, q)) for r in range(ws)]
    for p in procs:
        p.start()
    out = sorted(q.get(timeout=100) for _ in range(ws))
    for p in procs:
        p.join(30)
        assert p.exitcode == 0
    assert sorted(u for o in out for u in o[1]) == list(range(8))       # every object on exactly one rank
    assert all(len(o[1]) == 8 // ws for o in out)                          # 8 / N per rank
    assert out[0][2][0] == "bottle" and out[1][2][0] == "cracker_box"      # rank r starts with object r
    for o in out:
        assert o[3] == [list(range(r, 8, ws)) for r in range(ws)]          # every rank knows every rank's objects
        assert o[4] == [3 * (8 // ws)] * ws                                # 3 frames of each of its objects per rank
        # the cost column carries the object index: rank r's block is its objects in order, 3 frames each
        assert o[5] == [[float(u) for u in range(r, 8, ws) for _ in range(3)] for r in range(ws)]
