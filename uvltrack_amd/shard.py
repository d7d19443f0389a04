"""Multi-GPU sharding of the per-frame path: independent per-sequence batch shards, weights replicated,
the per-shard boxes all-gathered once per step -- or, for sub-millisecond frames, once per group of steps.

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


GATHER_EVERY_SHORT_FRAMES = 8        # steps per collective when a step is shorter than GATHER_PER_STEP_FROM_MS
GATHER_PER_STEP_FROM_MS = 2.0        # from this step time on: one all-gather per step, as SURVEY.md section 8e describes


def choose_every(step_ms: float) -> int:
    """Steps per collective from the measured step time (every rank must pass the SAME figure, e.g. the maximum over ranks).
    A step of >= 2 ms (BASELINE configs[4]: 8 UVLTrack-L sequences per GPU, ~6 ms) takes one all-gather per step -- the collective is
    far below 1 % of it and every consumer sees step t's boxes with step t.  Sub-millisecond frames (one UVLTrack-B sequence: ~96
    dependent launches of 5-10 us) group eight steps: there a stream-ordered collective per step costs 1.4 % and an overlapped one 11 %
    (tools/probes/gather_cost.py), and the boxes arrive up to seven frames late."""
    return 1 if step_ms >= GATHER_PER_STEP_FROM_MS else GATHER_EVERY_SHORT_FRAMES


class BoxGatherer:
    """All-gather of [n_local, 4] boxes into [n_total, 4] in global sequence order.  The boxes of `every` consecutive steps travel in ONE
    collective ([every, n_local, 4] per rank), issued in stream order on the caller's stream (a synchronous c10d call: the stream waits for
    the collective, the host does not).  every = 1 is SURVEY.md section 8e's "one all-gather per step"; with every > 1 the ranks meet once
    per group and a consumer sees a step's boxes at most `every` - 1 frames late (`lateness`).  `choose_every` picks it from the step time.

    Collectives are only ever issued by submit() (when a group completes) and by flush() / drain(), which EVERY rank must call at the
    same steps; result() never communicates -- asking for a step that still waits in the ring raises (a rank-local flush there would
    deadlock the others).  The last TWO gathered groups stay readable.

    Why not one overlapped collective per step (the form of rounds 1-3: async_op=True, double-buffered, waited for two steps later)?
    Measured on one MI355X under torch.distributed.run (tools/probes/gather_cost.py, UVLTrack-B one sequence, us per step): no gather 771,
    the box copy alone 773, the overlapped per-step gather 860 (+11.5 %) -- and 853-856 with the same asynchronous call every EIGHTH step:
    what costs is a second queue that stays live beside the frame's ~96 dependent launches, not the collective; the synchronous call on
    the frame's own stream costs 782 (+1.4 %) per step, a group of eight a quarter of a percent."""

    def __init__(self, n_sequences: int, device, group=None, every: int = 1):
        self.group = group
        self.world = dist.get_world_size(group) if dist.is_initialized() else 1
        self.rank = dist.get_rank(group) if dist.is_initialized() else 0
        self.n = n_sequences
        self.sizes = shard_sizes(n_sequences, self.world)
        self.lo, self.hi = shard_range(n_sequences, self.rank, self.world)
        self.pad = max(self.sizes) if self.sizes else 0          # ragged shards are padded to the largest one
        self.every = max(1, int(every))
        self.lateness = self.every - 1                           # worst case, in steps, between a step and the arrival of its boxes
        self._ring = torch.zeros(self.every, self.pad, 4, device=device)
        self._out = [torch.zeros(self.world, self.every, self.pad, 4, device=device) for _ in range(2)]
        self._first, self._count = 0, 0                          # steps [first, first + count) wait in the ring
        self._done = [(0, 0), (0, 0)]                            # per output buffer: steps [first, first + n) it holds
        self._cur = 1                                            # index of the buffer the LAST collective wrote
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
        """Gather whatever waits in the ring.  COLLECTIVE: every rank calls this at the same steps."""
        if self._count == 0:
            return
        self._cur ^= 1
        out = self._out[self._cur]
        if dist.is_initialized():          # also a group of ONE rank: the collective leg (RCCL on GPUs) is the same code at every world size
            dist.all_gather_into_tensor(out.view(self.world * self.every, self.pad, 4), self._ring, group=self.group)   # rank-major concatenation
        else:
            out[0].copy_(self._ring)
        self._done[self._cur] = (self._first, self._count)
        self._count = 0
        self.collectives += 1

    def result(self, step: int) -> torch.Tensor:
        """Boxes of every sequence for `step`, [n_sequences, 4] in global order; the step must be in one of the last two gathered groups."""
        if self._count and self._first <= step < self._first + self._count:
            raise ValueError("step %d still waits in the ring (group of %d, %d submitted): its collective is issued by submit() of the "
                             "group's last step or by flush() / drain() on EVERY rank" % (step, self.every, self._count))
        for b in (self._cur, self._cur ^ 1):
            first, n = self._done[b]
            if first <= step < first + n:
                parts = [self._out[b][r, step - first, :self.sizes[r]] for r in range(self.world)]
                return torch.cat(parts, dim=0)
        raise ValueError("step %d is not among the gathered steps %s" % (step, [(f, f + n) for f, n in self._done if n]))

    def drain(self):
        self.flush()
