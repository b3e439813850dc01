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


class TableGather:
    """THE exchange step of the path: all-gather of the per-rank fixed-stride match tables
    (`all_gather_into_tensor`: one RCCL call on GPUs, gloo in the CPU tests), issued asynchronously -- PyTorch runs the
    collective on its own communication stream behind the producing kernels, so the compute stream goes straight on to
    the next batch; two receive buffers alternate and a buffer is only reused after the collective that last wrote it
    has completed.  `bench.py` (every workload that has a table) and `gather_match_tables` / `run_sharded` below all go
    through this class; with world == 1 it hands the local table back (`force` keeps the collective even then: the
    single-rank RCCL check on a one-GPU box, tests/test_gpu_rccl_single_rank.py)."""

    def __init__(self, world: int, rows: int, stride: int, dtype, device, group=None, force: bool = False):
        self.world, self.group = world, group
        self.on = world > 1 or force
        self.bufs = [torch.empty((world * rows, stride), dtype=dtype, device=device) for _ in range(2)] if self.on else []
        self.work = [None, None]
        self.i = 0

    def __call__(self, table: torch.Tensor) -> torch.Tensor:
        """Start the gather of this step's table; returns the receive buffer (valid after `finish()` or after the
        second following call)."""
        if not self.on:
            return table
        j = self.i & 1
        if self.work[j] is not None:
            self.work[j].wait()
        self.work[j] = dist.all_gather_into_tensor(self.bufs[j], table.contiguous(), group=self.group, async_op=True)
        self.i += 1
        return self.bufs[j]

    def finish(self) -> None:
        for w in self.work:
            if w is not None:
                w.wait()
        self.work = [None, None]


def gather_match_tables(local_table: torch.Tensor, num_pairs: int, group=None) -> torch.Tensor:
    """All-gather the per-rank match tables [n_local, S] into the global table [num_pairs, S].

    Every rank contributes a block padded to ceil(num_pairs / world) rows so a single
    `all_gather_into_tensor` (one `TableGather` step) suffices; padding rows are dropped afterwards.
    """
    world = dist.get_world_size(group)
    per = padded_shard_size(num_pairs, world)
    stride = local_table.shape[1]
    send = local_table
    if local_table.shape[0] != per:
        send = local_table.new_zeros((per, stride))
        send[: local_table.shape[0]] = local_table
    if world == 1:
        return send[:num_pairs]
    tg = TableGather(world, per, stride, local_table.dtype, local_table.device, group)
    recv = tg(send)
    tg.finish()
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
