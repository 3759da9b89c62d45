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


def shard_batch(rank: int, world: int, *tensors, batched_adj: bool = True):
    """Slice every per-graph tensor (leading dim B) to this rank's graphs.  `None` entries pass through;
    a 2-D adjacency (shared by all graphs) is replicated, not split."""
    out = []
    bsz = next(t.shape[0] for t in tensors if t is not None)
    lo, hi = shard_bounds(bsz, rank, world)
    for t in tensors:
        if t is None:
            out.append(None)
        elif t.dim() == 2 and t.dtype == torch.bool and t.shape[0] == t.shape[1] and t.shape[0] != bsz:
            out.append(t)                       # (N,N) adjacency shared across the batch
        else:
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
