"""Multi-GPU sharding of the per-frame path: independent per-sequence batch shards, weights replicated,
the per-shard boxes all-gathered in groups of a few steps.

The reference runs one sequence per worker with `gpu_id = worker_id % num_gpu` and no communication
(lib/test/evaluation/running.py:96-100,168-171); here one process per GPU owns a contiguous shard of the
sequences and the decoded boxes are gathered over RCCL (backend "nccl" on ROCm) so every rank -- in particular
rank 0 -- sees the boxes of all sequences of every step.  Backend-agnostic: the CPU tests use gloo.
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
    """All-gather of [n_local, 4] boxes into [n_total, 4] in global sequence order.  The boxes of `every` consecutive steps travel in ONE
    collective ([every, n_local, 4] per rank), issued in stream order on the caller's stream (a synchronous c10d call: the stream waits for
    the collective, the host does not): the ranks meet once per group instead of once per frame, and a consumer sees a step's boxes at
    most `every` - 1 frames late (result() of a step that is still waiting gathers what is there).

    Why not one overlapped collective per step (the form of rounds 1-3: async_op=True, double-buffered, waited for two steps later)?
    Measured on one MI355X under torch.distributed.run (tools/probes/gather_cost.py, UVLTrack-B one sequence, us per step): no gather 771,
    the box copy alone 773, the overlapped per-step gather 860 (+11.5 %) -- and 853-856 with the same asynchronous call every EIGHTH step:
    what costs is a second queue that stays live beside the frame's ~96 dependent launches, not the collective; the synchronous call on
    the frame's own stream costs 782 (+1.4 %) per step, a group of eight a quarter of a percent."""

    def __init__(self, n_sequences: int, device, group=None, every: int = 8):
        self.group = group
        self.world = dist.get_world_size(group) if dist.is_initialized() else 1
        self.rank = dist.get_rank(group) if dist.is_initialized() else 0
        self.n = n_sequences
        self.sizes = shard_sizes(n_sequences, self.world)
        self.lo, self.hi = shard_range(n_sequences, self.rank, self.world)
        self.pad = max(self.sizes) if self.sizes else 0          # ragged shards are padded to the largest one
        self.every = max(1, int(every))
        self._ring = torch.zeros(self.every, self.pad, 4, device=device)
        self._out = torch.zeros(self.world, self.every, self.pad, 4, device=device)
        self._first, self._count = 0, 0                          # steps [first, first + count) wait in the ring
        self._done_first, self._done_n = 0, 0                    # steps [done_first, done_first + done_n) are in _out
        self.collectives = 0

    def submit(self, step: int, local_boxes: torch.Tensor) -> bool:
        """Take this step's boxes ([n_local, 4]); True when that completed a group and its collective has been enqueued."""
        n_local = self.hi - self.lo
        if tuple(local_boxes.shape) != (n_local, 4):
            raise ValueError("expected [%d, 4] local boxes, got %s" % (n_local, tuple(local_boxes.shape)))
        if self._count == 0:
            self._first = step
        elif step != self._first + self._count:
            raise ValueError("steps must be submitted in order: expected %d, got %d" % (self._first + self._count, step))
        self._ring[self._count, :n_local].copy_(local_boxes)
        self._count += 1
        if self._count == self.every:
            self.flush()
            return True
        return False

    def flush(self):
        """Gather whatever waits in the ring (every rank calls this at the same steps)."""
        if self._count == 0:
            return
        if dist.is_initialized():          # also a group of ONE rank: the collective leg (RCCL on GPUs) is the same code at every world size
            dist.all_gather_into_tensor(self._out.view(self.world * self.every, self.pad, 4), self._ring, group=self.group)   # rank-major concatenation
        else:
            self._out[0].copy_(self._ring)
        self._done_first, self._done_n = self._first, self._count
        self._count = 0
        self.collectives += 1

    def result(self, step: int) -> torch.Tensor:
        """Boxes of every sequence for `step`, [n_sequences, 4] in global order (a step still in the ring is gathered now)."""
        if self._count and self._first <= step < self._first + self._count:
            self.flush()
        s = step - self._done_first
        if not 0 <= s < self._done_n:
            raise ValueError("step %d is not among the gathered steps [%d, %d)" % (step, self._done_first, self._done_first + self._done_n))
        parts = [self._out[r, s, :self.sizes[r]] for r in range(self.world)]
        return torch.cat(parts, dim=0)

    def drain(self):
        self.flush()
