"""Multi-GPU sharding of the tracking workload.

The reference is single-process, single-GPU (pixloc_pose_refiners.py:35-39).  Frames of one
video are sequentially dependent (frame t starts from frame t-1's pose, mask, reference
render and cost threshold: pixloc_tracker_r9.py:224-229,258-265), so the unit of parallel
work is a SEQUENCE (one object / one video / one independently seeded segment).  Units are
dealt round-robin to ranks, one process per GPU; there is no communication inside the
tracking loop.  The only collective is the final gather of the per-frame pose records
(12 pose floats + success + cost = 14 float64 per frame), done with torch.distributed
(backend "nccl" == RCCL on ROCm; "gloo" in the CPU tests).
"""
from __future__ import annotations

import os
from typing import List, Sequence

import torch
import torch.distributed as dist

RECORD = 14  # R (9) | t (3) | success | cost


def world() -> tuple:
    if dist.is_available() and dist.is_initialized():
        return dist.get_rank(), dist.get_world_size()
    return 0, 1


def shard_units(n_units: int, rank: int, world_size: int) -> List[int]:
    """Static round-robin assignment of sequence indices to ranks."""
    return list(range(rank, n_units, world_size))


def shard_segments(n_frames: int, world_size: int, rank: int, segment_len: int = 0) -> List[tuple]:
    """Contiguous frame segments [start, stop) of ONE video dealt round-robin to ranks (BASELINE
    configs[4] "frames sharded over 8 GPUs").  Frames of a video depend on their predecessor
    (pixloc_tracker_r9.py:216-229), so every segment head is a cold start: results at segment heads
    differ from a sequential run - the benchmark says so next to its number.
    ``segment_len`` 0 = one segment per rank."""
    if n_frames <= 0:
        return []
    if segment_len <= 0:
        segment_len = -(-n_frames // max(world_size, 1))
    starts = list(range(0, n_frames, segment_len))
    return [(s, min(n_frames, s + segment_len)) for i, s in enumerate(starts) if i % world_size == rank]


def stitch_segments(gathered: Sequence[torch.Tensor], segments_per_rank: Sequence[Sequence[tuple]], n_frames: int):
    """Per-rank record blocks (each rank's segments concatenated in order) -> one [n_frames, 14]
    tensor in frame order."""
    out = torch.zeros(n_frames, RECORD, dtype=torch.float64)
    for recs, segs in zip(gathered, segments_per_rank):
        k = 0
        for (a, b) in segs:
            out[a:b] = recs[k:k + (b - a)]
            k += b - a
    return out


def load_object_configs() -> List[dict]:
    """The per-object records of the reference's config/*.sh (values committed as data in
    configs/objects.json); boxes are returned with min <= max per axis."""
    import json
    from pathlib import Path

    objs = json.loads((Path(__file__).parent / "configs" / "objects.json").read_text())["objects"]
    for o in objs:
        lo, hi = o["OBJ_AABB"]
        o["aabb"] = [[min(a, b) for a, b in zip(lo, hi)], [max(a, b) for a, b in zip(lo, hi)]]
    return objs


def frame_tracked(ret: dict) -> bool:
    """The tracker's decision for a frame: refiner success AND the policy's gate (r9: cost <= 1.1 x the first frame's,
    pixloc_tracker_r9.py:258-263; `tracked` in the record) - not the refiner's `success` alone, which the reference's
    poses.pkl carries (a frame the gate rejected keeps `success: True` and a `T_refined` the tracker did NOT adopt)."""
    return bool(ret.get("tracked", ret.get("success")))


def pack_pose_records(history: dict, names: Sequence[str]) -> torch.Tensor:
    """[n_frames, 14] float64 from a tracker's pose_history."""
    out = torch.zeros(len(names), RECORD, dtype=torch.float64)
    for i, n in enumerate(names):
        ret = history[n]
        ok = frame_tracked(ret)
        T = ret.get("T_refined", ret["T_init"]) if ok else ret["T_init"]
        out[i, :12] = T.as12().detach().cpu().double()
        out[i, 12] = 1.0 if ok else 0.0
        out[i, 13] = float(ret.get("cost", float("nan")))
    return out


def gather_pose_records(records: torch.Tensor, device=None) -> List[torch.Tensor]:
    """All ranks receive every rank's [n_i, 14] records (ragged n_i handled by padding)."""
    rank, ws = world()
    if ws == 1:
        return [records]
    dev = device if device is not None else records.device
    n = torch.tensor([records.shape[0]], dtype=torch.int64, device=dev)
    ns = [torch.zeros_like(n) for _ in range(ws)]
    dist.all_gather(ns, n)
    nmax = int(max(int(x) for x in ns))
    pad = torch.zeros(nmax, RECORD, dtype=torch.float64, device=dev)
    pad[: records.shape[0]] = records.to(dev)
    outs = [torch.zeros_like(pad) for _ in range(ws)]
    dist.all_gather(outs, pad)
    return [o[: int(k)].cpu() for o, k in zip(outs, ns)]


def gather_objects(obj) -> list:
    """Every rank receives every rank's (small, picklable) record, in rank order."""
    rank, ws = world()
    if ws == 1:
        return [obj]
    out = [None] * ws
    dist.all_gather_object(out, obj)
    return out


def max_over_ranks(value: float, device=None) -> float:
    rank, ws = world()
    if ws == 1:
        return value
    t = torch.tensor([value], dtype=torch.float64, device=device if device is not None else "cpu")
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t)


def init_from_env(backend: str = "nccl"):
    """torchrun contract: RANK / LOCAL_RANK / WORLD_SIZE / MASTER_ADDR / MASTER_PORT."""
    ws = int(os.environ.get("WORLD_SIZE", "1"))
    if ws > 1 and not dist.is_initialized():
        dist.init_process_group(backend=backend, rank=int(os.environ["RANK"]), world_size=ws)
    return int(os.environ.get("RANK", "0")), ws, int(os.environ.get("LOCAL_RANK", "0"))


def bind_to_device_numa(device_index: int = 0):
    """Pins this process to the CPUs of the NUMA node its GPU hangs off (one process per GPU on a
    two-socket host: the per-frame host turnaround - launches, the pinned-memory completion word -
    is 30 % slower from the far socket).  Returns the node, or None when the topology is not
    exposed; never raises."""
    try:
        p = torch.cuda.get_device_properties(device_index)
        bdf = f"{int(getattr(p, 'pci_domain_id', 0)):04x}:{int(p.pci_bus_id):02x}:{int(p.pci_device_id):02x}.0"
        with open(f"/sys/bus/pci/devices/{bdf}/numa_node") as f:
            node = int(f.read().strip())
        if node < 0:
            return None
        with open(f"/sys/devices/system/node/node{node}/cpulist") as f:
            spec = f.read().strip()
        cpus = set()
        for part in spec.split(","):
            a, _, b = part.partition("-")
            cpus.update(range(int(a), int(b or a) + 1))
        cpus &= os.sched_getaffinity(0)
        if not cpus:
            return None
        os.sched_setaffinity(0, cpus)
        return node
    except Exception:
        return None
