#!/usr/bin/env python3
"""Benchmark of the MI355X UVLTrack per-frame forward pass (`forward_test`), reference loop:
tracking/profile_model.py:30-47 (warm-up, then K timed forwards, sync only at both ends).

    python bench.py [--gpus N] [--steps K] [--warmup W] [--batch B] [--model B|L] [--mode NLBBOX|NL|BBOX]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
        bench.py --gpus N --steps K --warmup W
    python bench.py --gpus 8 --model L --batch 8          # BASELINE configs[4]: 64 UVLTrack-L sequences, 8 per GPU; a plain
                                                          # process with --gpus N > 1 launches its own N ranks (one per GPU)

A step = one pass of the hot path over one batch of synthetic frames per GPU (inputs resident in HBM, weights
from the deterministic generator).  Each rank owns its own sequences (independent per-sequence
shards, weights replicated); the per-shard boxes of every step are all-gathered over RCCL, eight steps per collective, in stream order
(uvltrack_amd.shard.BoxGatherer: an overlapped collective per step cost the one-sequence frame 11 %, this form a quarter of a percent).
A timed BLOCK is exactly `--steps` steps between barrier + synchronize on both sides, maximum over ranks; the line reports the
MEDIAN of `--blocks` such blocks (default: as many as fit in about three seconds, 3..400), so a 20-step run of a 0.7 ms frame is
not a 15 ms sample.  Rank 0 prints ONE JSON line.

Launch forms.  Under `torch.distributed.run` (WORLD_SIZE in the environment) every rank joins an RCCL process group -- also when
there is one rank, so the collective leg runs on a single GPU too.  A plain `python bench.py --gpus N` with N > 1 (or `--dist`)
re-launches itself through `python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1`, the way the
reference's evaluation spawns one worker per GPU (lib/test/evaluation/running.py:153-171); rank 0 still prints the one line.
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

PEAK_BF16_TFLOPS = 2500.0      # MI355X dense bf16 MFMA peak (MI355X_MICROARCH.md)
PEAK_HBM_GBS = 8000.0


def parse_args(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=300)
    ap.add_argument("--warmup", type=int, default=50)
    ap.add_argument("--blocks", type=int, default=0, help="timed blocks of --steps steps (0 = about one second's worth, 3..25)")
    ap.add_argument("--batch", type=int, default=1, help="sequences per GPU advanced one frame per step")
    ap.add_argument("--model", default="B", choices=["B", "L"])
    ap.add_argument("--template-size", type=int, default=256)
    ap.add_argument("--search-size", type=int, default=None)
    ap.add_argument("--mode", default="NLBBOX", choices=["BBOX", "NL", "NLBBOX"])
    ap.add_argument("--skip-text", action="store_true", help="BBOX mode only: do not run the text branch (configs[1])")
    ap.add_argument("--reuse-text", action="store_true", help="NOT the headline: frames after the first take the text branch (BERT embedding + "
                    "pre-fusion layers) from the workspace, as the tracker does for a fixed sentence (uvl_inputs.reuse_text); eager launch only")
    ap.add_argument("--launch", default="eager", choices=["eager", "graph"],
                    help="eager: the frame's kernels are launched each step (one stream for one sequence, two for several; fastest on ROCm 7.2, "
                         "where hipGraphLaunch costs more host time per node than a plain launch); graph: hipGraph replay")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-batched", action="store_true", help="skip the extra `batched` block (UVLTrack-L z256/x384, 8 sequences: the per-GPU load of BASELINE configs[4])")
    ap.add_argument("--cpu-frames", type=int, default=8)
    ap.add_argument("--cpu-threads", type=int, default=32)
    ap.add_argument("--profile-json", default=None, help="also write the per-kernel breakdown to this file")
    ap.add_argument("--dist", action="store_true", help="run under torch.distributed.run even with --gpus 1 (RCCL process group of one rank)")
    ap.add_argument("--gather-every", type=int, default=0, help="steps per box all-gather (0 = from the step time: 1 when a step is >= 2 ms, else 8; uvltrack_amd/shard.py)")
    ap.add_argument("--tune", action="append", default=[], metavar="KEY=VALUE",
                    help="tools only: uvl_tune_set(KEY, VALUE) on the engine's handle (include/uvltrack_hip.h: uvl_tuning); recorded in the line")
    return ap.parse_args(argv)


def build_spec(model, template_size, search_size):
    from uvltrack_amd.spec import spec_b, spec_l
    if model == "B":
        return spec_b(template_size, search_size or 256)
    return spec_l(template_size, search_size or 384)


def _cpu_model():
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name"):
                return line.split(":", 1)[1].strip()
    except OSError:
        pass
    return "unknown"


def cpu_baseline(spec, args, flags):
    """The reference's CPU path, restated: oracle/uvl_oracle_torch.py runs forward_test with the ATen operators the reference's eager
    forward runs (F.linear / layer_norm / conv2d / softmax / gelu: the loop of tracking/profile_model.py:38-47), timed on this host's cores
    on a bounded sample (kind "port-torch").  The numpy / OpenBLAS oracle is timed beside it on a smaller sample (`numpy_port`)."""
    import torch
    from oracle import uvl_oracle as O
    from oracle import uvl_oracle_torch as OT
    from uvltrack_amd import weightgen as wg
    sd = wg.make_state_dict(spec, 0, include_unused=False)
    inp = wg.make_inputs(spec, batch=1, seed=100, flags=flags[:1])
    threads = max(1, min(os.cpu_count() or 1, args.cpu_threads))
    prev = torch.get_num_threads()
    torch.set_num_threads(threads)
    try:
        sdt = OT.prepare(sd)
        run_t = lambda: OT.forward_test(sdt, spec, inp["template"], inp["search"], inp["ids"], inp["mask"], inp["prompt"], inp["flag"], prepared=True)
        run_t()
        t0 = time.perf_counter()
        run_t()
        one = time.perf_counter() - t0
        n = int(min(60, max(args.cpu_frames, round(12.0 / max(one, 1e-3)))))          # ~12 s of CPU work, at least --cpu-frames frames
        t0 = time.perf_counter()
        for _ in range(n):
            run_t()
        dt = time.perf_counter() - t0
        used = torch.get_num_threads()
    finally:
        torch.set_num_threads(prev)
    # the numpy oracle beside it (OpenBLAS with one thread per host core thrashes on these small GEMMs: cap the pool)
    run_n = lambda: O.forward_test(sd, spec, inp["template"], inp["search"], inp["ids"], inp["mask"], inp["prompt"], inp["flag"])
    try:
        from threadpoolctl import threadpool_limits
        limiter = threadpool_limits(limits=threads)
    except Exception:
        limiter = None
    try:
        run_n()
        nn = max(2, min(4, args.cpu_frames))
        t0 = time.perf_counter()
        for _ in range(nn):
            run_n()
        dtn = time.perf_counter() - t0
    finally:
        if limiter is not None:
            limiter.restore_original_limits()
    return {"value": n / dt, "unit": "frames/s", "cores": used, "kind": "port-torch", "cpu": _cpu_model(), "host_cores": os.cpu_count(),
            "sample": "%d frames of the same workload (batch 1, fp32, torch %s CPU operators = the reference's eager ATen path restated in "
                      "oracle/uvl_oracle_torch.py, %d intra-op threads of %d host cores)" % (n, torch.__version__.split("+")[0], used, os.cpu_count()),
            "numpy_port": {"value": nn / dtn, "unit": "frames/s", "cores": threads, "sample": "%d frames, numpy / OpenBLAS oracle" % nn},
            # the REAL reference (eager PyTorch fp32, imported from /root/reference) cannot travel to the GPU box; measured at survey time:
            "reference_survey": {"value": 4.8, "unit": "frames/s", "cores": 8,
                                 "source": "BASELINE.md section 3: reference forward_test, UVLTrack-B z256/x256/T40, 8 vCPU Xeon (Icelake), torch 2.10 MKL/oneDNN, 8 threads"}}


class DistEnv:
    """The few collective operations the timing loop needs; `NoDist` when there is one process."""

    def __init__(self, dist, device):
        self.dist, self.device = dist, device
        self.world = dist.get_world_size()

    def barrier(self):
        self.dist.barrier()

    def max_over_ranks(self, seconds: float) -> float:
        import torch
        tt = torch.tensor([seconds], device=self.device, dtype=torch.float64)
        self.dist.all_reduce(tt, op=self.dist.ReduceOp.MAX)
        return float(tt.item())

    def gather_floats(self, x: float):
        """One value per rank, in rank order, on every rank (outside the timed blocks: the per-rank step times of the line)."""
        import torch
        mine = torch.tensor([x], device=self.device, dtype=torch.float64)
        out = [torch.zeros_like(mine) for _ in range(self.world)]
        self.dist.all_gather(out, mine)
        return [float(t.item()) for t in out]


class NoDist:
    world = 1

    def barrier(self):
        pass

    def max_over_ranks(self, seconds: float) -> float:
        return seconds

    def gather_floats(self, x: float):
        return [x]


def throughput(world: int, per_rank_units: int, steps: int, block_seconds) -> float:
    """`value` of the line: the units ALL ranks processed in a block / the block's time -- which is the maximum over ranks (timed_blocks), so a
    straggler rank lowers the whole job's figure.  Median over the blocks."""
    return world * per_rank_units * steps / float(np.median(block_seconds))


def rank_step_times(env, step_fn, sync, n: int = 10):
    """Per-rank step time (ms): every rank runs `n` steps on its own -- no collective in them, so a rank cannot hide behind the others' wait -- and the figures are
    gathered with ONE all_gather outside the timed blocks.  A straggler GPU shows up in the one line (min / median / max, the slowest rank's index, the spread);
    `value` of the line follows the slowest rank (throughput())."""
    sync()
    t0 = time.perf_counter()
    for _ in range(n):
        step_fn()
    sync()
    mine = (time.perf_counter() - t0) / n * 1e3
    per = env.gather_floats(mine)
    return {"per_rank": [round(x, 4) for x in per], "min": min(per), "median": float(np.median(per)), "max": max(per), "slowest_rank": int(np.argmax(per)),
            "spread": (max(per) - min(per)) / max(min(per), 1e-9), "steps": n}


def timed_blocks(step_fn, local_boxes_fn, gatherer, env, sync, steps: int, warmup: int, blocks: int, on_result=None, local_times=None):
    """THE loop of this benchmark -- also what tests/test_bench_loop_gloo.py drives with a stub step over gloo.
    step_fn(i) enqueues one frame of this rank's sequences; local_boxes_fn() is the [n_local, 4] tensor it leaves; gatherer
    (uvltrack_amd.shard.BoxGatherer, or None for one process) all-gathers them, a group of steps per collective.  on_result(i, boxes)
    is called on every rank, in step order, with the gathered boxes of step i once its group is complete (and at the end of a run).  Returns the list of block times (seconds, maximum over ranks), `blocks`
    entries (0 = choose from the first block: about three seconds in total, 3..400).  local_times (a list, optional) receives THIS rank's own time of every
    block: from the common start to the end of its own work, before the closing barrier."""
    counter = [0]
    delivered = [-1]

    def deliver_upto(i):
        if gatherer is not None and on_result is not None:
            for j in range(delivered[0] + 1, i + 1):
                on_result(j, gatherer.result(j))
        delivered[0] = max(delivered[0], i)

    def one(i):
        step_fn(i)
        if gatherer is not None and gatherer.submit(i, local_boxes_fn()):
            deliver_upto(i)

    def finish(last):
        if gatherer is not None:
            gatherer.drain()
            if last >= 0:
                deliver_upto(last)

    def run(n):
        first = counter[0]
        for k in range(n):
            one(first + k)
        counter[0] += n
        finish(counter[0] - 1 if n > 0 else -1)

    run(warmup)
    sync()
    times = []
    want = blocks
    while True:
        env.barrier()
        sync()
        t0 = time.perf_counter()
        run(steps)
        sync()
        if local_times is not None:
            local_times.append(time.perf_counter() - t0)
        env.barrier()
        sync()
        times.append(env.max_over_ranks(time.perf_counter() - t0))
        if want <= 0:
            want = int(min(400, max(3, round(3.0 / max(times[0], 1e-6)))))
        if len(times) >= want:
            return times


def pick_gather_every(step_fn, sync, env, forced: int = 0) -> int:
    """Steps per box all-gather (uvltrack_amd/shard.py::choose_every) from a few untimed steps: the SLOWEST rank's step time decides, so that every rank builds
    the same groups (a rank that disagreed would wait in a collective the others never issue).  tests/test_bench_loop_gloo.py drives it over gloo with stub steps."""
    from uvltrack_amd.shard import choose_every
    if forced > 0:
        return forced
    for _ in range(3):
        step_fn()
    sync()
    t0 = time.perf_counter()
    for _ in range(5):
        step_fn()
    sync()
    return choose_every(env.max_over_ranks((time.perf_counter() - t0) / 5) * 1e3)


def measure(torch, dev, env, rank, spec, B, flag_val, args, skip_text, reuse_text, use_graph, steps, warmup, blocks, seed, eng=None, max_batch=None):
    """Build an engine for `spec` (or take `eng`, built for the same spec), run the timed blocks; returns (times, engine, targs, outs)."""
    from uvltrack_amd import weightgen as wg
    from uvltrack_amd.engine import HipEngine
    from uvltrack_amd.shard import BoxGatherer
    flags = [flag_val] * B
    if eng is None:
        eng = HipEngine(spec, dev, max_batch=max(B, max_batch or 1))
        for kv in args.tune:
            k, v = kv.split("=", 1)
            if k.strip().startswith("debug."):          # uvl_debug_set keys (A/B aids that are not launch heuristics), e.g. debug.fork_text=0
                eng.debug_set(k.strip()[6:], int(v))
            else:
                eng.tune_set(k.strip(), int(v))
        eng.load_state_dict(wg.make_state_dict(spec, 0, include_unused=False))
    inp = wg.make_inputs(spec, batch=B, seed=seed + rank, flags=flags)      # every rank advances its own sequences
    t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)
    targs = (t(inp["template"]), t(inp["search"]), t(inp["ids"]), t(inp["mask"]), t(inp["prompt"]), t(inp["flag"]))
    if not use_graph:
        outs = eng.alloc_outputs(B)
        step_fn = eng.make_eager_step(*targs, skip_text=skip_text, outs=outs, reuse_text=reuse_text)
    else:
        _, outs = eng.capture(*targs, skip_text=skip_text)
        step_fn = eng.replay
    # RCCL all-gather of the per-shard boxes: rank r owns sequences [r*B, (r+1)*B).  One collective per step wherever a
    # step is >= 2 ms (configs[4]); sub-millisecond frames send eight steps per collective (shard.choose_every): the cadence comes from a
    # few untimed steps, the slowest rank's figure, so that every rank builds the same groups
    gatherer = None
    if isinstance(env, DistEnv):                      # also with ONE rank under torch.distributed.run
        every = pick_gather_every(step_fn, torch.cuda.synchronize, env, int(getattr(args, "gather_every", 0) or 0))
        gatherer = BoxGatherer(env.world * B, dev, every=every)
    eng.local_times = []
    times = timed_blocks(lambda i: step_fn(), lambda: outs["pred_boxes"].view(B, 4), gatherer, env, torch.cuda.synchronize, steps, warmup, blocks, local_times=eng.local_times)
    finite = bool(torch.isfinite(outs["bbox_map"]).all().item()) and bool(torch.isfinite(outs["logits"]).all().item())
    if not finite:
        raise SystemExit("bench.py: the forward pass produced non-finite outputs -- timing of a broken path is not reported")
    eng.gather_every = gatherer.every if gatherer is not None else None
    eng.rank_ms = rank_step_times(env, step_fn, torch.cuda.synchronize)          # (a collective: every rank, outside the timed blocks)
    return times, eng, targs, outs


def kernel_rooflines(eng, targs, spec, B, frame_ms, skip_text, reuse_text, single_stream, model):
    """HIP events around every launch of one profiled frame, on the launching stream -> per (site, kernel) entries; returns the
    `roofline` block of the kernel instantiation with the largest summed time and the `roofline_attention` block."""
    if reuse_text:
        eng.forward(*targs)                       # leaves the text rows the profiled frame reuses
    eng.forward(*targs, skip_text=skip_text, profile=True, reuse_text=reuse_text)
    prof = eng.profile_entries()
    # A HIP event pair around a launch also times the dispatch of that launch.  One-sequence frames are a single stream of
    # back-to-back kernels, so the bracketing cost per launch is known exactly on average:
    #   (sum of all event-pair times - duration of the un-instrumented frame) / launches.
    # It is taken off every launch so that avg_launch_us is comparable with rocprofv3's kernel durations
    # (profiles/*_bench_kernel_stats.csv); frames with a second stream (batch > 1) report the raw event-pair time.
    n_launch = sum(e["launches"] for e in prof)
    event_overhead_ms = max(0.0, (sum(e["ms"] for e in prof) - frame_ms) / max(n_launch, 1)) if single_stream else 0.0
    by_kernel = {}
    for e in prof:
        k = by_kernel.setdefault(e["kernel"], dict(ms=0.0, flops=0.0, bytes=0.0, bytes_weights=0.0, launches=0, sites=[]))
        k["ms"] += e["ms"]; k["flops"] += e["flops"]; k["bytes"] += e["bytes"]; k["launches"] += e["launches"]
        k["bytes_weights"] += e.get("bytes_weights", e["bytes"])
        k["sites"].append(e["site"])

    def block(name, d):
        raw_avg_ms = d["ms"] / max(d["launches"], 1)
        avg_ms = max(raw_avg_ms - event_overhead_ms, 0.25 * raw_avg_ms)
        tf = d["flops"] / d["launches"] / (avg_ms * 1e-3) / 1e12 if d["flops"] > 0 else 0.0
        # TWO byte counts.  `bytes_weights` = SURVEY 8(d)'s algorithmic count (weights touched once, activations cache-resident; attention:
        # q, k, v in + o out): `bound` / `achieved` / `frac` are formed on IT, so the figure is comparable across rounds (rounds 1-3: 0.074-0.078
        # for the same ~9.1 us kernel that round 4's wider count reported as 0.183).  `bytes` = that + what the kernel must also move (A rows,
        # the bf16 / f32 output, the f32 read-modify-write of the residual epilogue, once per split-K slab): reported beside it as frac_hbm_moved.
        gbs_moved = d["bytes"] / d["launches"] / (avg_ms * 1e-3) / 1e9
        gbs = d["bytes_weights"] / d["launches"] / (avg_ms * 1e-3) / 1e9
        frac_mfma, frac_hbm = tf / PEAK_BF16_TFLOPS, gbs / PEAK_HBM_GBS
        # HBM bytes per launch of that kernel from the committed rocprofv3 PMC passes (2*FETCH_SIZE + WRITE_SIZE, see
        # profiles/*_pmc_summary.md); null when no PMC pass of this kernel instantiation on this workload has been committed
        traffic, traffic_source = None, None
        try:
            pdir = os.path.join(ROOT, "profiles")
            for fn in sorted(f for f in os.listdir(pdir) if f.endswith("_pmc_traffic.json")):
                doc = json.load(open(os.path.join(pdir, fn)))
                wl = doc.get("workload", {"model": "B", "batch": 1})
                tb = doc.get("bytes_per_launch", {})
                if name in tb and wl.get("batch") == B and wl.get("model") == model:
                    traffic, traffic_source = tb[name], "profiles/" + fn
        except OSError:
            pass
        bound = "mfma" if frac_mfma >= frac_hbm else "hbm"
        # neither roof within 1/0.15 of this kernel: it waits for latency (dependent round trips of a 5-10 us launch), and the
        # label says so; achieved / peak / frac stay those of the nearer roof
        regime = "latency" if max(frac_mfma, frac_hbm) < 0.15 else bound
        return {"bound": bound, "regime": regime, "kernel": name, "sites": sorted(set(d["sites"])), "launches_per_frame": d["launches"],
                "avg_launch_us": avg_ms * 1e3, "avg_launch_us_event_pair": raw_avg_ms * 1e3, "event_overhead_us": event_overhead_ms * 1e3,
                "flops_per_launch": d["flops"] / max(d["launches"], 1), "bytes_per_launch": d["bytes_weights"] / max(d["launches"], 1),
                "bytes_weights_per_launch": d["bytes_weights"] / max(d["launches"], 1), "frac_hbm_weights": frac_hbm,
                "bytes_moved_per_launch": d["bytes"] / max(d["launches"], 1), "achieved_gbs_moved": gbs_moved, "frac_hbm_moved": gbs_moved / PEAK_HBM_GBS,
                "achieved": tf if bound == "mfma" else gbs, "peak": PEAK_BF16_TFLOPS if bound == "mfma" else PEAK_HBM_GBS,
                "unit": "TFLOP/s" if bound == "mfma" else "GB/s", "frac": max(frac_mfma, frac_hbm),
                "achieved_tflops": tf, "frac_mfma": frac_mfma, "achieved_gbs": gbs, "frac_hbm": frac_hbm, "traffic": traffic,
                # `traffic` is NOT measured in this run: HBM bytes per launch (2 x FETCH_SIZE + WRITE_SIZE) from the committed rocprofv3
                # --pmc passes of this same command, looked up by kernel instantiation and workload
                "traffic_source": traffic_source}

    dom_name, dom = max(by_kernel.items(), key=lambda kv: kv[1]["ms"])
    roofline = block(dom_name, dom)
    attn = {k: v for k, v in by_kernel.items() if k.startswith("attn") and any(s_ == "attention" for s_ in v["sites"])}
    roofline_attention = None
    if attn:
        an, ad = max(attn.items(), key=lambda kv: kv[1]["ms"])
        roofline_attention = block(an, ad)
    return roofline, roofline_attention, by_kernel, prof, n_launch


def self_launch(n: int) -> int:
    """Re-run this command line as `n` ranks, one per GPU, through torch.distributed.run on 127.0.0.1 (a free port); returns the
    launcher's exit code.  Rank 0 of the child job prints the JSON line on the inherited stdout."""
    import socket
    import subprocess
    with socket.socket() as so:
        so.bind(("127.0.0.1", 0))
        port = so.getsockname()[1]
    envp = dict(os.environ)
    envp.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    argv = [a for a in sys.argv[1:] if a != "--dist"]
    if "--gpus" not in " ".join(argv):
        argv += ["--gpus", str(n)]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(n), "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + argv
    return subprocess.call(cmd, env=envp)


def main():
    args = parse_args()
    import torch
    import torch.distributed as dist

    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")     # dmabuf IPC: what RCCL needs on this driver (exported by the harness too)
    under_launcher = "WORLD_SIZE" in os.environ and "RANK" in os.environ
    if not under_launcher and (args.gpus > 1 or args.dist):
        raise SystemExit(self_launch(max(1, args.gpus)))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if under_launcher and world != max(1, args.gpus):
        raise SystemExit("--gpus %d but the launcher started %d rank(s)" % (args.gpus, world))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a HIP device: the product path has no CPU fallback")
    if local_rank >= torch.cuda.device_count():
        raise SystemExit("rank %d has no GPU: %d device(s) visible, --gpus %d" % (rank, torch.cuda.device_count(), args.gpus))
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    if under_launcher:                                             # one rank included: the RCCL leg is the same code at every N
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group(backend="nccl", device_id=dev)     # RCCL on ROCm
    env = DistEnv(dist, dev) if under_launcher else NoDist()

    spec = build_spec(args.model, args.template_size, args.search_size)
    B = args.batch
    flag_val = {"BBOX": 0, "NL": 1, "NLBBOX": 2}[args.mode]
    skip_text = bool(args.skip_text and args.mode == "BBOX")
    reuse_text = bool(args.reuse_text and not skip_text)
    use_graph = args.launch == "graph"

    times, eng, targs, outs = measure(torch, dev, env, rank, spec, B, flag_val, args, skip_text, reuse_text, use_graph,
                                      args.steps, args.warmup, args.blocks, seed=1000)
    elapsed = float(np.median(times))
    rank_ms = eng.rank_ms

    if rank == 0:
        fps = throughput(world, B, args.steps, times)
        frame_ms = elapsed / args.steps * 1e3
        single_stream = (B == 1 or skip_text or reuse_text) and not use_graph
        roofline, roofline_attention, by_kernel, prof, n_launch = kernel_rooflines(eng, targs, spec, B, frame_ms, skip_text, reuse_text, single_stream, args.model)
        flops_frame = spec.flops_per_frame(skip_text=skip_text, reuse_text=reuse_text)
        from uvltrack_amd.spec import state_dict_schema
        weight_bytes = 2.0 * sum(int(np.prod(s)) for n, s in state_dict_schema(spec, False).items()
                                 if len(s) >= 2 and "embeddings" not in n and "pos_embed" not in n)
        line = {
            "metric": "tracker FPS (frames/sec) UVLTrack-%s forward_test" % args.model,
            "value": fps, "unit": "frames/s", "n_gpus": world, "rccl_ranks": world if under_launcher else 0,
            "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": elapsed / args.steps * 1e3, "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "bf16", "data": "synthetic",
            "blocks": len(times), "block_ms": [round(x * 1e3, 3) for x in times], "statistic": "median of the blocks (each: exactly `steps` steps, max over ranks)",
            # every rank's own ms per step (ten steps without a collective, gathered outside the timed region): `value` follows the SLOWEST rank
            "rank_ms_per_step": rank_ms,
            "config": {"workload": "UVLTrack-%s z%d/x%d/T%d %s%s, %d sequence(s)/GPU, batch shards + RCCL all-gather of boxes" % (
                args.model, spec.template_size, spec.search_size, spec.text_len, args.mode, " (text branch skipped)" if skip_text else (" (text branch reused from the first frame)" if reuse_text else ""), B),
                "per_gpu_batch": B, "global_batch": B * world, "tokens_visual": spec.nv, "tokens_joint": spec.nj,
                "gflop_per_frame": flops_frame / 1e9, "launches_per_frame": n_launch,
                "launch": "hipgraph" if use_graph else ("eager-1-stream" if single_stream else "eager-2-streams"), "parallelism": "dp%d" % world,
                # the box all-gather's cadence (uvltrack_amd/shard.py): steps per collective and the worst-case lateness of a step's boxes,
                # in steps; null without a process group (one plain process: nothing to gather)
                "gather_every": eng.gather_every, "gather_lateness_steps": None if eng.gather_every is None else eng.gather_every - 1},
            "frame_model_tflops": flops_frame * fps / 1e12,
            "frame_mfma_frac": flops_frame * fps / 1e12 / (PEAK_BF16_TFLOPS * world),
            "frame_hbm_frac": (weight_bytes * (args.steps / elapsed)) / 1e9 / PEAK_HBM_GBS,
            "outputs_finite": True,
            "roofline": roofline,
        }
        if args.tune:
            line["config"]["tune"] = list(args.tune)              # NOT the default path: heuristics overridden for an A/B
        if roofline_attention is not None:
            line["roofline_attention"] = roofline_attention
    eng.close()
    del eng, targs, outs

    # ---- the other BASELINE configs, one process only (the driver's record then holds all five): configs[1] = BBOX mode with the text branch skipped, configs[3] =
    #      one UVLTrack-L sequence z256/x384, `batched` = the per-GPU load of configs[4] (8 UVLTrack-L sequences), the regime where MFMA fractions mean something.
    #      Each: >= 30 blocks of the same number of steps, median, with its own launch count and frame-level roofline fractions.
    def extra(model, mode, skip, Bx, eng_x, label, seed):
        xspec = build_spec(model, 256, None)
        xsteps = max(5, min(args.steps, 40))
        xt, xeng, xtargs, _ = measure(torch, dev, env, rank, xspec, Bx, {"BBOX": 0, "NL": 1, "NLBBOX": 2}[mode], args, skip, False, False, xsteps, max(3, min(args.warmup, 8)),
                                      max(30, args.blocks), seed=seed, eng=eng_x, max_batch=8 if model == "L" else 1)
        xel = float(np.median(xt))
        xfps = throughput(1, Bx, xsteps, xt)
        xflops = xspec.flops_per_frame(skip_text=skip)
        from uvltrack_amd.spec import state_dict_schema as sds
        xwb = 2.0 * sum(int(np.prod(sh)) for n, sh in sds(xspec, False).items() if len(sh) >= 2 and "embeddings" not in n and "pos_embed" not in n and not (skip and ".bert." in n))
        xroof, xattn, _, _, xn = kernel_rooflines(xeng, xtargs, xspec, Bx, xel / xsteps * 1e3, skip, False, Bx == 1 or skip, model)
        blk = {"workload": label, "value": xfps, "unit": "frames/s", "ms_per_step": xel / xsteps * 1e3, "steps": xsteps, "blocks": len(xt),
               "block_ms_min_max": [round(min(xt) * 1e3, 3), round(max(xt) * 1e3, 3)],
               "gflop_per_frame": xflops / 1e9, "frame_model_tflops": xflops * xfps / 1e12, "frame_mfma_frac": xflops * xfps / 1e12 / PEAK_BF16_TFLOPS,
               "frame_hbm_frac": xwb * (xfps / Bx) / 1e9 / PEAK_HBM_GBS, "launches_per_frame": xn, "roofline": xroof, "roofline_attention": xattn}
        return blk, xeng

    if world == 1 and not args.no_batched and not (args.model == "L" and B == 8):
        c1, e1 = extra("B", "BBOX", True, 1, None, "UVLTrack-B z256/x256 BBOX, text branch skipped, 1 sequence (BASELINE configs[1])", 3000)
        e1.close()
        c3, eL = extra("L", "NLBBOX", False, 1, None, "UVLTrack-L z256/x384/T40 NLBBOX, 1 sequence (BASELINE configs[3])", 4000)
        line["configs"] = {"configs[1]": c1, "configs[3]": c3}
        line["batched"], eL = extra("L", "NLBBOX", False, 8, eL, "UVLTrack-L z256/x384/T40 NLBBOX, 8 sequences on this GPU (per-GPU load of BASELINE configs[4])", 2000)
        eL.close()

    if rank == 0:
        if not args.no_cpu_baseline and world == 1:
            line["cpu_baseline"] = cpu_baseline(spec, args, [flag_val] * B)
        if args.profile_json:
            with open(args.profile_json, "w") as f:
                json.dump({"by_kernel": by_kernel, "sites": prof, "line": line}, f, indent=1)
        print(json.dumps(line))
    if under_launcher:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
