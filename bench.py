#!/usr/bin/env python3
"""Benchmark of the MI355X UVLTrack per-frame forward pass (`forward_test`), reference loop:
tracking/profile_model.py:30-47 (warm-up, then K timed forwards, sync only at both ends).

    python bench.py [--gpus N] [--steps K] [--warmup W] [--batch B] [--model B|L] [--mode NLBBOX|NL|BBOX]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
        bench.py --gpus N --steps K --warmup W

A step = one pass of the hot path over one batch of synthetic frames per GPU (inputs resident in HBM, weights
from the deterministic generator, hipGraph replay of the captured frame).  Each rank owns its own sequences
(independent per-sequence shards, weights replicated); after every step the per-shard boxes
are all-gathered over RCCL on a side stream.  Rank 0 prints ONE JSON line.
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


def parse_args():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=300)
    ap.add_argument("--warmup", type=int, default=50)
    ap.add_argument("--batch", type=int, default=1, help="sequences per GPU advanced one frame per step")
    ap.add_argument("--model", default="B", choices=["B", "L"])
    ap.add_argument("--template-size", type=int, default=256)
    ap.add_argument("--search-size", type=int, default=None)
    ap.add_argument("--mode", default="NLBBOX", choices=["BBOX", "NL", "NLBBOX"])
    ap.add_argument("--skip-text", action="store_true", help="BBOX mode only: do not run the text branch (configs[1])")
    ap.add_argument("--reuse-text", action="store_true", help="NOT the headline: frames after the first take the text branch (BERT embedding + "
                    "pre-fusion layers) from the workspace, as the tracker does for a fixed sentence (uvl_inputs.reuse_text); eager launch only")
    ap.add_argument("--launch", default="eager", choices=["eager", "graph"],
                    help="eager: the frame's ~100-150 kernels are launched each step (one stream for one sequence, two for several) (fastest on ROCm 7.2, where "
                         "hipGraphLaunch costs more host time per node than a plain launch); graph: hipGraph replay")
    ap.add_argument("--no-graph", action="store_true", help="alias of --launch eager")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-frames", type=int, default=8)
    ap.add_argument("--cpu-threads", type=int, default=32)
    ap.add_argument("--profile-json", default=None, help="also write the per-kernel breakdown to this file")
    return ap.parse_args()


def build_spec(args):
    from uvltrack_amd.spec import spec_b, spec_l
    if args.model == "B":
        return spec_b(args.template_size, args.search_size or 256)
    return spec_l(args.template_size, args.search_size or 384)


def cpu_baseline(spec, args, flags):
    """The numpy oracle (a port of the reference's algorithm) timed on this host's cores on a bounded sample."""
    from oracle import uvl_oracle as O
    from uvltrack_amd import weightgen as wg
    sd = wg.make_state_dict(spec, 0, include_unused=False)
    inp = wg.make_inputs(spec, batch=1, seed=100, flags=flags[:1])
    run = lambda: O.forward_test(sd, spec, inp["template"], inp["search"], inp["ids"], inp["mask"], inp["prompt"], inp["flag"])
    # OpenBLAS with one thread per host core (256 on the GPU box) thrashes on these small GEMMs: cap the pool
    threads = min(os.cpu_count() or 1, args.cpu_threads)
    try:
        from threadpoolctl import threadpool_limits
        limiter = threadpool_limits(limits=threads)
    except Exception:
        limiter, threads = None, os.cpu_count()
    try:
        run()
        n = max(1, args.cpu_frames)
        t0 = time.perf_counter()
        for _ in range(n):
            run()
        dt = time.perf_counter() - t0
    finally:
        if limiter is not None:
            limiter.restore_original_limits()
    return {"value": n / dt, "unit": "frames/s", "cores": threads, "kind": "port",
            "sample": "%d frames of the same workload (batch 1, fp32 numpy/OpenBLAS oracle, %d BLAS threads of %d host cores)" % (
                n, threads, os.cpu_count())}


def main():
    args = parse_args()
    import torch
    import torch.distributed as dist

    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")     # dmabuf IPC: what RCCL needs on this driver (exported by the harness too)
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != max(1, args.gpus):
        if world == 1 and args.gpus > 1:
            raise SystemExit("--gpus %d needs `python -m torch.distributed.run --nproc-per-node %d bench.py ...`" % (args.gpus, args.gpus))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a HIP device: the product path has no CPU fallback")
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group(backend="nccl", device_id=dev)     # RCCL on ROCm

    from uvltrack_amd import weightgen as wg
    from uvltrack_amd.engine import HipEngine

    spec = build_spec(args)
    B = args.batch
    flag_val = {"BBOX": 0, "NL": 1, "NLBBOX": 2}[args.mode]
    flags = [flag_val] * B
    skip_text = bool(args.skip_text and args.mode == "BBOX")
    reuse_text = bool(args.reuse_text and not skip_text)

    eng = HipEngine(spec, dev, max_batch=max(B, 1))
    eng.load_state_dict(wg.make_state_dict(spec, 0, include_unused=False))
    # every rank advances its own sequences: different synthetic frames per rank
    inp = wg.make_inputs(spec, batch=B, seed=1000 + rank, flags=flags)
    t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)
    targs = (t(inp["template"]), t(inp["search"]), t(inp["ids"]), t(inp["mask"]), t(inp["prompt"]), t(inp["flag"]))

    use_graph = args.launch == "graph" and not args.no_graph
    if not use_graph:
        outs = eng.alloc_outputs(B)
        step_fn = eng.make_eager_step(*targs, skip_text=skip_text, outs=outs, reuse_text=reuse_text)
    else:
        _, outs = eng.capture(*targs, skip_text=skip_text)
        step_fn = eng.replay

    # RCCL all-gather of the per-shard boxes: rank r owns sequences [r*B, (r+1)*B); the gather of step i
    # overlaps the forward pass of step i+1 (uvltrack_amd/shard.py, covered on CPU by tests/test_shard_gloo.py)
    from uvltrack_amd.shard import BoxGatherer
    gatherer = BoxGatherer(world * B, dev) if world > 1 else None

    def step(i):
        step_fn()
        if gatherer is not None:
            gatherer.submit(i, outs["pred_boxes"].view(B, 4))

    def drain():
        if gatherer is not None:
            gatherer.drain()

    for i in range(args.warmup):
        step(i)
    drain()
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(args.steps):
        step(i)
    drain()
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    if world > 1:
        tt = torch.tensor([elapsed], device=dev, dtype=torch.float64)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        elapsed = float(tt.item())

    finite = bool(torch.isfinite(outs["bbox_map"]).all().item()) and bool(torch.isfinite(outs["logits"]).all().item())
    if not finite:
        raise SystemExit("bench.py: the forward pass produced non-finite outputs -- timing of a broken path is not reported")

    if rank == 0:
        frames = world * B * args.steps
        fps = frames / elapsed
        # ---- roofline of the dominant kernel: HIP events around every launch, on the launching stream ----
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
        single_stream = (B == 1) and os.environ.get("UVL_PAIR_TEXT", "1") != "0" and not use_graph
        frame_ms = elapsed / args.steps * 1e3
        event_overhead_ms = max(0.0, (sum(e["ms"] for e in prof) - frame_ms) / max(n_launch, 1)) if single_stream else 0.0
        by_kernel = {}
        for e in prof:
            k = by_kernel.setdefault(e["kernel"], dict(ms=0.0, flops=0.0, bytes=0.0, launches=0, sites=[]))
            k["ms"] += e["ms"]; k["flops"] += e["flops"]; k["bytes"] += e["bytes"]; k["launches"] += e["launches"]
            k["sites"].append(e["site"])
        dom_name, dom = max(by_kernel.items(), key=lambda kv: kv[1]["ms"])
        raw_avg_ms = dom["ms"] / max(dom["launches"], 1)
        avg_ms = max(raw_avg_ms - event_overhead_ms, 0.25 * raw_avg_ms)
        achieved = dom["flops"] / dom["launches"] / (avg_ms * 1e-3) / 1e12 if dom["flops"] > 0 else 0.0
        # HBM bytes per launch of that kernel from the committed rocprofv3 PMC passes (2*FETCH_SIZE + WRITE_SIZE, see
        # profiles/*_pmc_summary.md); null when no PMC pass of this kernel instantiation has been committed
        traffic = None
        try:
            pdir = os.path.join(ROOT, "profiles")
            for fn in sorted(f for f in os.listdir(pdir) if f.endswith("_pmc_traffic.json")):
                tb = json.load(open(os.path.join(pdir, fn))).get("bytes_per_launch", {})
                if dom_name in tb and B == 1 and args.model == "B":
                    traffic = tb[dom_name]
        except OSError:
            pass
        roofline = {"bound": "mfma", "kernel": dom_name, "sites": sorted(set(dom["sites"])), "launches_per_frame": dom["launches"],
                    "avg_launch_us": avg_ms * 1e3, "avg_launch_us_event_pair": raw_avg_ms * 1e3, "event_overhead_us": event_overhead_ms * 1e3,
                    "flops_per_launch": dom["flops"] / max(dom["launches"], 1),
                    "achieved": achieved, "peak": PEAK_BF16_TFLOPS, "unit": "TFLOP/s", "frac": achieved / PEAK_BF16_TFLOPS,
                    "traffic": traffic}
        flops_frame = spec.flops_per_frame(skip_text=skip_text, reuse_text=reuse_text)
        weight_bytes = 2.0 * sum(int(np.prod(s)) for n, s in __import__("uvltrack_amd.spec", fromlist=["x"]).state_dict_schema(spec, False).items()
                                 if len(s) >= 2 and "embeddings" not in n and "pos_embed" not in n)
        line = {
            "metric": "tracker FPS (frames/sec) UVLTrack-%s forward_test" % args.model,
            "value": fps, "unit": "frames/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": elapsed / args.steps * 1e3, "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "bf16", "data": "synthetic",
            "config": {"workload": "UVLTrack-%s z%d/x%d/T%d %s%s, %d sequence(s)/GPU, batch shards + RCCL all-gather of boxes" % (
                args.model, spec.template_size, spec.search_size, spec.text_len, args.mode, " (text branch skipped)" if skip_text else (" (text branch reused from the first frame)" if reuse_text else ""), B),
                "per_gpu_batch": B, "global_batch": B * world, "tokens_visual": spec.nv, "tokens_joint": spec.nj,
                "gflop_per_frame": flops_frame / 1e9, "launch": "hipgraph" if use_graph else ("eager-1-stream" if (B == 1 and not skip_text and os.environ.get("UVL_PAIR_TEXT", "1") != "0") or skip_text or reuse_text else "eager-2-streams"), "parallelism": "dp%d" % world},
            "frame_model_tflops": flops_frame * fps / 1e12,
            "frame_mfma_frac": flops_frame * fps / 1e12 / (PEAK_BF16_TFLOPS * world),
            "frame_hbm_frac": (weight_bytes * (args.steps / elapsed)) / 1e9 / PEAK_HBM_GBS,
            "outputs_finite": finite,
            "roofline": roofline,
        }
        if not args.no_cpu_baseline and world == 1:
            line["cpu_baseline"] = cpu_baseline(spec, args, flags)
        if args.profile_json:
            with open(args.profile_json, "w") as f:
                json.dump({"by_kernel": by_kernel, "sites": prof, "line": line}, f, indent=1)
        print(json.dumps(line))
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
