"""bench.py's own timing / submit / drain loop (`bench.timed_blocks`, the code the driver runs with --gpus N) driven on CPU:
world_size 2 over gloo with a stub in place of the HIP engine.  Every rank checks, step by step, that the gathered boxes are
those of all sequences in global order (per-rank seeds), one step late as the loop delivers them, and that the block timing is
the maximum over ranks.  No scaling curve has been measured on hardware (no multi-GPU node was available to the build or the
driver so far); this covers the path by construction."""
import os
import socket
import sys
import time

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _boxes(seq, step):
    g = torch.Generator().manual_seed(1000 * seq + step)
    return torch.rand(4, generator=g)


def _worker(rank, world, port, B, every, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    sys.path.insert(0, ROOT)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        import bench
        from uvltrack_amd.shard import BoxGatherer
        dev = torch.device("cpu")
        env = bench.DistEnv(dist, dev)
        local = torch.zeros(B, 4)
        seen = []

        def step_fn(i):                        # the stub "frame": this rank's sequences rank*B .. rank*B+B-1 at step i
            for j in range(B):
                local[j] = _boxes(rank * B + j, i)
            if rank == 1:
                time.sleep(0.002)              # the slower rank sets the block time

        def on_result(i, boxes):
            exp = torch.stack([_boxes(s, i) for s in range(world * B)])
            seen.append((i, bool(torch.equal(boxes, exp))))

        gatherer = BoxGatherer(world * B, dev, every=every)
        steps, warmup, blocks = 6, 3, 2
        local_times = []
        times = bench.timed_blocks(step_fn, lambda: local, gatherer, env, lambda: None, steps, warmup, blocks, on_result=on_result, local_times=local_times)
        # the line's diagnostics (outside the timed blocks): every rank's own step time, and the value = all ranks' frames / the SLOWEST rank's time
        stats = bench.rank_step_times(env, lambda: step_fn(0), lambda: None, n=5)
        value = bench.throughput(world, B, steps, times)
        ok = len(times) == blocks and all(t >= steps * 0.002 * 0.9 for t in times)          # rank 0 reports rank 1's time too
        ok &= [i for i, _ in seen] == list(range(warmup + blocks * steps)) and all(f for _, f in seen)
        # one collective per step, or one per group of `every` steps + one per partial group at the end of a run (warm-up, each block)
        runs = [warmup] + [steps] * blocks
        ok &= gatherer.collectives == sum(-(-n // every) for n in runs)
        q.put((rank, ok, times, stats, value))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("every", [1, 8, 4])
def test_bench_loop_world2_gloo(every):
    """Both cadences of the box all-gather: per step (steps of >= 2 ms: BASELINE configs[4]) and per group (sub-millisecond frames)."""
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, 3, every, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = [q.get(timeout=180) for _ in procs]
    for p in procs:
        p.join(timeout=60)
    assert sorted(r[0] for r in res) == [0, 1]
    assert all(r[1] for r in res), res
    (_, _, t0, st0, v0), (_, _, t1, st1, v1) = sorted(res)
    assert t0 == t1                                # the max over ranks is what both report
    # per-rank step times: the same list on both ranks, rank 1 (which sleeps 2 ms per step) is the slowest and at least 2 ms
    assert st0 == st1 and st0["slowest_rank"] == 1 and len(st0["per_rank"]) == 2
    assert st0["per_rank"][1] >= 2.0 * 0.9 and st0["per_rank"][0] < st0["per_rank"][1]
    # value = (frames of ALL ranks) / (time of the slowest rank): 2 ranks x 3 sequences x 6 steps per block
    import numpy as np
    frames = 2 * 3 * 6
    assert v0 == v1 == frames / float(np.median(t0))
    assert v0 <= frames / (2.0e-3 * 6)      # never better than the straggler's 2 ms per step allows (no lower bound: gloo's collectives on a loaded CI host are slow)


def test_bench_loop_single_process():
    sys.path.insert(0, ROOT)
    import bench
    calls = []
    local = []
    times = bench.timed_blocks(lambda i: calls.append(i), lambda: None, None, bench.NoDist(), lambda: None, 5, 2, 3, local_times=local)
    assert len(times) == 3 and calls == list(range(2 + 3 * 5)) and len(local) == 3
    st = bench.rank_step_times(bench.NoDist(), lambda: None, lambda: None, n=3)
    assert st["per_rank"] and st["slowest_rank"] == 0 and st["min"] == st["max"]
    assert bench.throughput(1, 4, 5, [2.0, 1.0, 4.0]) == 4 * 5 / 2.0


def test_self_launcher_builds_one_rank_per_gpu(monkeypatch):
    """`python bench.py --gpus N` as a plain process re-launches itself through torch.distributed.run on 127.0.0.1 (what the
    reference's evaluation does with one worker per GPU, lib/test/evaluation/running.py:153-171): N ranks, the same flags,
    dmabuf IPC in the environment.  The command is checked here; a GPU test runs it for real with one rank."""
    sys.path.insert(0, ROOT)
    import subprocess
    import bench
    seen = {}

    def fake_call(cmd, env=None):
        seen["cmd"], seen["env"] = cmd, env
        return 0
    monkeypatch.setattr(subprocess, "call", fake_call)
    monkeypatch.setattr(sys, "argv", ["bench.py", "--gpus", "8", "--model", "L", "--batch", "8", "--steps", "5", "--dist"])
    assert bench.self_launch(8) == 0
    cmd = seen["cmd"]
    assert cmd[1:4] == ["-m", "torch.distributed.run", "--nnodes=1"]
    assert cmd[cmd.index("--nproc-per-node") + 1] == "8" and cmd[cmd.index("--master-addr") + 1] == "127.0.0.1"
    assert int(cmd[cmd.index("--master-port") + 1]) > 0
    tail = cmd[cmd.index(os.path.join(ROOT, "bench.py")) + 1:]
    assert tail == ["--gpus", "8", "--model", "L", "--batch", "8", "--steps", "5"]          # --dist is consumed by the launcher
    assert seen["env"]["HSA_ENABLE_IPC_MODE_LEGACY"] == "0"


def _worker_one(port, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    sys.path.insert(0, ROOT)
    dist.init_process_group("gloo", rank=0, world_size=1)
    try:
        from uvltrack_amd.shard import BoxGatherer
        g = BoxGatherer(3, torch.device("cpu"), every=1)
        ok = True
        for i in range(4):
            b = torch.stack([_boxes(s, i) for s in range(3)])
            g.submit(i, b)
            ok &= bool(torch.equal(g.result(i), b))
            ok &= g.collectives == i + 1                   # a real collective was enqueued, also for a group of one rank
        g.drain()
        q.put(ok)
    finally:
        dist.destroy_process_group()


def _worker_cadence(rank, world, port, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    sys.path.insert(0, ROOT)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        import bench
        env = bench.DistEnv(dist, torch.device("cpu"))
        # rank 0 steps in 0.2 ms, rank 1 in 3 ms: alone rank 0 would group eight steps, rank 1 would not -- the slowest rank decides for both
        slow = bench.pick_gather_every(lambda: time.sleep(0.0002 if rank == 0 else 0.003), lambda: None, env)
        fast = bench.pick_gather_every(lambda: time.sleep(0.0002), lambda: None, env)
        forced = bench.pick_gather_every(lambda: None, lambda: None, env, forced=4)
        q.put((rank, slow, fast, forced))
    finally:
        dist.destroy_process_group()


def test_ranks_agree_on_the_gather_cadence_world2_gloo():
    """bench.pick_gather_every: the cadence comes from the slowest rank's step time (max over ranks), so the ranks cannot build different groups."""
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker_cadence, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=120) for _ in procs)
    for p in procs:
        p.join(timeout=60)
    assert [r[1:] for r in res] == [(1, 8, 4), (1, 8, 4)], res


def test_gather_cadence_follows_the_step_time():
    """one all-gather per step wherever a step is long enough for it to be free (configs[4]: ~6 ms); groups of eight only
    for sub-millisecond frames.  The bench line reports the cadence and the worst-case lateness."""
    from uvltrack_amd.shard import choose_every
    assert choose_every(6.2) == 1 and choose_every(2.0) == 1
    assert choose_every(0.73) == 8 and choose_every(1.9) == 8


def test_gatherer_runs_the_collective_for_a_group_of_one():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    p = ctx.Process(target=_worker_one, args=(port, q))
    p.start()
    assert q.get(timeout=120)
    p.join(timeout=60)
