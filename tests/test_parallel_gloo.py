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
