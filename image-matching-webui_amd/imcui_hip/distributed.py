"""Multi-GPU sharding of the pair list.

Image pairs are independent units: one process per GPU, a static contiguous shard of the pair
list per rank, weights replicated, NO data-path collective inside the model.  The only exchange
is one all-gather of the fixed-stride match table per step (16 KiB per pair at K = 2048), which
is latency- not bandwidth-bound on xGMI.  backend "nccl" is RCCL on ROCm; "gloo" is used by the
CPU tests.  The reference's only multi-GPU mechanism is replica-per-GPU Ray actors
(imcui/api/server.py:42-66) with no inter-replica traffic.
"""
from __future__ import annotations

import torch
import torch.distributed as dist


def shard_bounds(num_pairs: int, rank: int, world: int) -> tuple[int, int]:
    """Contiguous block of the pair list owned by `rank` (sizes differ by at most one)."""
    base, rem = divmod(num_pairs, world)
    start = rank * base + min(rank, rem)
    return start, start + base + (1 if rank < rem else 0)


def padded_shard_size(num_pairs: int, world: int) -> int:
    return -(-num_pairs // world)


def gather_match_tables(local_table: torch.Tensor, num_pairs: int, group=None) -> torch.Tensor:
    """All-gather the per-rank match tables [n_local, S] into the global table [num_pairs, S].

    Every rank contributes a block padded to ceil(num_pairs / world) rows so a single
    `all_gather_into_tensor` (one RCCL call) suffices; padding rows are dropped afterwards.
    """
    world = dist.get_world_size(group)
    per = padded_shard_size(num_pairs, world)
    stride = local_table.shape[1]
    send = local_table
    if local_table.shape[0] != per:
        send = local_table.new_zeros((per, stride))
        send[: local_table.shape[0]] = local_table
    recv = local_table.new_empty((world * per, stride))
    dist.all_gather_into_tensor(recv, send.contiguous(), group=group)
    parts = []
    for r in range(world):
        s, e = shard_bounds(num_pairs, r, world)
        parts.append(recv[r * per : r * per + (e - s)])
    return torch.cat(parts, 0)


def run_sharded(pairs_fn, num_pairs: int, match_fn, group=None) -> torch.Tensor:
    """`pairs_fn(start, end)` yields this rank's inputs, `match_fn(inputs)` returns its
    [n_local, S] int32 match table; returns the gathered global table on every rank."""
    rank, world = dist.get_rank(group), dist.get_world_size(group)
    s, e = shard_bounds(num_pairs, rank, world)
    table = match_fn(pairs_fn(s, e))
    return gather_match_tables(table, num_pairs, group)
