"""Batch sharding of independent graphs over the GPUs of one node.

Graphs in a batch never interact (every op of EGNN.forward is batched over the leading B
dimension), so the multi-GPU path is a contiguous split of B with NO collective inside forward.
One process per GPU (`torch.distributed`, backend "nccl" = RCCL over xGMI on ROCm; "gloo" in the CPU
tests).  Parameters are replicated: load the same state_dict on every rank or call
`broadcast_parameters`.  Outputs stay sharded by default (what data-parallel callers want);
`gather_batch` is the optional epilogue all-gather along B.
"""
from __future__ import annotations

import torch
import torch.distributed as dist


def shard_bounds(batch: int, rank: int, world: int):
    """Contiguous split of `batch` graphs: the first (batch % world) ranks get one extra graph."""
    if world < 1 or not (0 <= rank < world):
        raise ValueError(f"bad rank/world {rank}/{world}")
    base, rem = divmod(batch, world)
    start = rank * base + min(rank, rem)
    return start, start + base + (1 if rank < rem else 0)


def shard_batch(rank: int, world: int, *tensors, shared=()):
    """Slice every per-graph tensor (leading dim B) to this rank's graphs.  `None` entries pass through.  Tensors that
    all graphs share -- a 2-D `(N, N)` adjacency -- are named explicitly in `shared` (compared by identity) and are
    replicated, not split; there is no shape heuristic (a `(B, N)` mask with B == N looks exactly like an adjacency).
    Every other tensor must have the batch as its leading dimension."""
    shared_ids = {id(t) for t in shared if t is not None}
    per_graph = [t for t in tensors if t is not None and id(t) not in shared_ids]
    if not per_graph:
        raise ValueError("shard_batch needs at least one per-graph tensor")
    bsz = per_graph[0].shape[0]
    lo, hi = shard_bounds(bsz, rank, world)
    out = []
    for t in tensors:
        if t is None:
            out.append(None)
        elif id(t) in shared_ids:
            out.append(t)                       # replicated on every rank
        else:
            if t.shape[0] != bsz:
                raise ValueError(f"per-graph tensor with leading dim {t.shape[0]} != batch {bsz} "
                                 f"(pass tensors shared by all graphs via shared=(...))")
            out.append(t[lo:hi])
    return out


def broadcast_parameters(module: torch.nn.Module, src: int = 0, group=None):
    """Replicate rank `src`'s parameters and buffers on every rank (one small broadcast per tensor)."""
    for t in list(module.parameters()) + list(module.buffers()):
        dist.broadcast(t.data, src=src, group=group)


def gather_batch(local: torch.Tensor, batch: int, group=None):
    """All-gather per-rank output shards back to (batch, ...) on every rank.  Shards may differ by one
    graph, so they are padded to the largest shard for the collective and trimmed afterwards."""
    world = dist.get_world_size(group)
    sizes = [shard_bounds(batch, r, world) for r in range(world)]
    cap = max(hi - lo for lo, hi in sizes)
    pad = local
    if local.shape[0] < cap:
        pad = torch.cat([local, local.new_zeros((cap - local.shape[0],) + tuple(local.shape[1:]))], dim=0)
    bufs = [torch.empty_like(pad) for _ in range(world)]
    dist.all_gather(bufs, pad.contiguous(), group=group)
    return torch.cat([buf[: hi - lo] for buf, (lo, hi) in zip(bufs, sizes)], dim=0)


def pin_to_local_numa_node(device_index: int | None = None):
    """Best effort: restrict this process's threads to the CPU cores of the NUMA node its GPU hangs off (kernel launches and the
    pinned-memory status read then stay on the socket next to the device: one process per GPU on a two-socket 8-GPU node).  Reads the
    GPU's PCI address from the device properties and the node's core list from sysfs; returns the core set, or None when either is
    unavailable (containers without sysfs topology, a single-node host) -- never raises."""
    import os
    try:
        idx = torch.cuda.current_device() if device_index is None else device_index
        pr = torch.cuda.get_device_properties(idx)
        bdf = f"{pr.pci_domain_id:04x}:{pr.pci_bus_id:02x}:{pr.pci_device_id:02x}.0"
        node = int(open(f"/sys/bus/pci/devices/{bdf}/numa_node").read())
        if node < 0:
            return None
        cpus = set()
        for part in open(f"/sys/devices/system/node/node{node}/cpulist").read().strip().split(","):
            lo, _, hi = part.partition("-")
            cpus.update(range(int(lo), int(hi or lo) + 1))
        cpus &= os.sched_getaffinity(0)
        if not cpus:
            return None
        os.sched_setaffinity(0, cpus)
        return cpus
    except Exception:
        return None
