"""The N>1 path of bench.py on CPU: world_size 2 over gloo (shard ownership + box all-gather)."""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from uvltrack_amd.shard import BoxGatherer, shard_range, shard_sizes


def test_shard_ranges_cover_everything():
    for n in (1, 7, 8, 63, 64, 65):
        for w in (1, 2, 3, 8):
            spans = [shard_range(n, r, w) for r in range(w)]
            assert spans[0][0] == 0 and spans[-1][1] == n
            assert all(spans[i][1] == spans[i + 1][0] for i in range(w - 1))
            assert max(shard_sizes(n, w)) - min(shard_sizes(n, w)) <= 1
    with pytest.raises(ValueError):
        shard_range(8, 2, 2)


def _worker(rank, world, port, n_seq, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        g = BoxGatherer(n_seq, torch.device("cpu"))          # every = 1: SURVEY 8e's one all-gather per step
        lo, hi = shard_range(n_seq, rank, world)
        ok = g.every == 1 and g.lateness == 0
        for step in range(5):
            # box of sequence s at step t is a known function of (s, t)
            local = torch.stack([torch.tensor([s, step, s * step, 1.0]) for s in range(lo, hi)]) if hi > lo else torch.zeros(0, 4)
            g.submit(step, local)
            ok &= g.collectives == step + 1
            if step >= 1:           # the previous step's boxes stay readable beside this step's
                got = g.result(step - 1)
                exp = torch.stack([torch.tensor([s, step - 1, s * (step - 1), 1.0]) for s in range(n_seq)])
                ok &= bool(torch.equal(got, exp))
        g.drain()
        got = g.result(4)
        exp = torch.stack([torch.tensor([s, 4.0, s * 4.0, 1.0]) for s in range(n_seq)])
        ok &= bool(torch.equal(got, exp))
        q.put((rank, ok))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("n_seq", [2, 5])
def test_box_gather_world2_gloo(n_seq):
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, n_seq, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = [q.get(timeout=120) for _ in procs]
    for p in procs:
        p.join(timeout=60)
    assert sorted(r for r, _ in res) == [0, 1]
    assert all(ok for _, ok in res)


def _worker_groups(rank, world, port, n_seq, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        g = BoxGatherer(n_seq, torch.device("cpu"), every=8)
        lo, hi = shard_range(n_seq, rank, world)
        box = lambda s, t: torch.tensor([s, t, s * t, 1.0])
        ok, done = True, []
        for step in range(19):
            local = torch.stack([box(s, step) for s in range(lo, hi)]) if hi > lo else torch.zeros(0, 4)
            if g.submit(step, local):              # a group of eight steps has just been gathered: all of its steps can be read
                done.append(step)
                for t in range(step - 7, step + 1):
                    ok &= bool(torch.equal(g.result(t), torch.stack([box(s, t) for s in range(n_seq)])))
        ok &= done == [7, 15] and g.collectives == 2
        try:
            g.result(17)                           # still in the ring: result() never communicates (a rank-local flush would deadlock the others)
            ok = False
        except ValueError:
            pass
        ok &= g.collectives == 2
        ok &= bool(torch.equal(g.result(2), torch.stack([box(s, 2) for s in range(n_seq)])))      # the previous group stays readable
        g.drain()                                  # the three steps left in the ring travel in a third collective
        ok &= g.collectives == 3
        for t in (16, 17, 18, 9, 15):              # the last two groups
            ok &= bool(torch.equal(g.result(t), torch.stack([box(s, t) for s in range(n_seq)])))
        ok &= g.lateness == 7
        try:
            g.result(3)                            # an earlier group's slot has been reused
            ok = False
        except ValueError:
            pass
        q.put((rank, ok))
    finally:
        dist.destroy_process_group()


def test_box_gather_groups_of_eight_steps_world2_gloo():
    """The form bench.py runs: the boxes of eight steps per collective, a partial group at the end of a run."""
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker_groups, args=(r, 2, port, 5, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = [q.get(timeout=120) for _ in procs]
    for p in procs:
        p.join(timeout=60)
    assert sorted(r for r, _ in res) == [0, 1]
    assert all(ok for _, ok in res)
