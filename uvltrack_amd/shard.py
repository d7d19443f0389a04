"""Multi-GPU sharding of the per-frame path: independent per-sequence batch shards, weights replicated,
one all-gather of the per-shard boxes per step.

The reference runs one sequence per worker with `gpu_id = worker_id % num_gpu` and no communication
(lib/test/evaluation/running.py:96-100,168-171); here one process per GPU owns a contiguous shard of the
sequences and the decoded boxes are gathered over RCCL (backend "nccl" on ROCm) so every rank -- in particular
rank 0 -- sees the boxes of all sequences after each step.  Backend-agnostic: the CPU tests use gloo.
"""
from __future__ import annotations

from typing import List, Tuple

import torch
import torch.distributed as dist


def shard_range(n_sequences: int, rank: int, world: int) -> Tuple[int, int]:
    """Contiguous shard [lo, hi) of `n_sequences` owned by `rank`; the first (n % world) ranks get one extra."""
    if world <= 0 or not (0 <= rank < world):
        raise ValueError("bad rank/world %d/%d" % (rank, world))
    base, extra = divmod(n_sequences, world)
    lo = rank * base + min(rank, extra)
    return lo, lo + base + (1 if rank < extra else 0)


def shard_sizes(n_sequences: int, world: int) -> List[int]:
    return [shard_range(n_sequences, r, world)[1] - shard_range(n_sequences, r, world)[0] for r in range(world)]


class BoxGatherer:
    """All-gather of [n_local, 4] boxes into [n_total, 4] in global sequence order, double-buffered so the collective
    of step i overlaps the forward pass of step i+1 (the gather runs on the process group's own stream)."""

    def __init__(self, n_sequences: int, device, group=None):
        self.group = group
        self.world = dist.get_world_size(group) if dist.is_initialized() else 1
        self.rank = dist.get_rank(group) if dist.is_initialized() else 0
        self.n = n_sequences
        self.sizes = shard_sizes(n_sequences, self.world)
        self.lo, self.hi = shard_range(n_sequences, self.rank, self.world)
        self.pad = max(self.sizes) if self.sizes else 0          # ragged shards are padded to the largest one
        self._in = [torch.zeros(self.pad, 4, device=device) for _ in range(2)]
        self._out = [torch.zeros(self.world * self.pad, 4, device=device) for _ in range(2)]
        self._pending = [None, None]

    def submit(self, step: int, local_boxes: torch.Tensor):
        """Enqueue the gather of this step's boxes ([n_local, 4]); returns immediately."""
        k = step & 1
        if self._pending[k] is not None:
            self._pending[k].wait()
        n_local = self.hi - self.lo
        if tuple(local_boxes.shape) != (n_local, 4):
            raise ValueError("expected [%d, 4] local boxes, got %s" % (n_local, tuple(local_boxes.shape)))
        self._in[k][:n_local].copy_(local_boxes)
        if dist.is_initialized():          # also a group of ONE rank: the collective leg (RCCL on GPUs) is the same code at every world size
            self._pending[k] = dist.all_gather_into_tensor(self._out[k], self._in[k], group=self.group, async_op=True)
        else:
            self._out[k].copy_(self._in[k])

    def result(self, step: int) -> torch.Tensor:
        """Boxes of every sequence for `step`, [n_sequences, 4] in global order (waits for that step's gather)."""
        k = step & 1
        if self._pending[k] is not None:
            self._pending[k].wait()
            self._pending[k] = None
        parts = [self._out[k][r * self.pad: r * self.pad + self.sizes[r]] for r in range(self.world)]
        return torch.cat(parts, dim=0)

    def drain(self):
        for k in range(2):
            if self._pending[k] is not None:
                self._pending[k].wait()
                self._pending[k] = None
