"""Micro-benchmark of the fused attention kernel (QK^T + softmax + PV, head_dim 64) on the frame's shapes."""
import ctypes as C
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from uvltrack_amd import _native  # noqa: E402

lib = _native.load()
TUNE = _native.UvlTuning()      # per-call overrides of the launch heuristics (no process-global tuning state)
def p(t):
    return C.c_void_p(t.data_ptr())


def main():
    """usage: attn_bench.py [B H N] [--cfgs 3,8,9]   (cfg -1 = the library's heuristic)"""
    st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    shapes = [(1, 12, 553), (8, 12, 553), (64, 12, 553), (1, 16, 681), (8, 16, 681), (8, 16, 873), (32, 16, 681), (32, 16, 873), (64, 16, 681)]
    args = sys.argv[1:]
    cfgs = [-1]
    if "--cfgs" in args:
        i = args.index("--cfgs")
        cfgs = [int(c) for c in args[i + 1].split(",")]
        del args[i:i + 2]
    if "--wgs" in args:                       # persistent workgroups of attn_p64_kernel (cfg 11)
        i = args.index("--wgs")
        TUNE.attn_wgs = int(args[i + 1])
        del args[i:i + 2]
    if len(args) >= 3:
        shapes = [(int(args[0]), int(args[1]), int(args[2]))]
    for B, H, N in shapes:
        Npad = (N + 63) // 64 * 64
        q = (torch.randn(B, H, Npad, 64, device="cuda") * 0.18033688).bfloat16()     # pre-scaled by log2(e)/8, as the frame's QKV GEMM leaves it
        k = torch.randn(B, H, Npad, 64, device="cuda").bfloat16()
        vt = torch.randn(B, H, 64, Npad, device="cuda").bfloat16()
        add = torch.zeros(B, Npad, device="cuda")
        o = torch.empty(B * N, H * 64, device="cuda", dtype=torch.bfloat16)
        fn = lambda: lib.uvl_attention(p(q), p(k), p(vt), p(add), p(o), B, H, N, Npad, 1, TUNE.ref(), st)
        # rounds INTERLEAVED over the configurations, best round per configuration: the first thing a process times runs 5-10 % slower
        # than the same launch a second later (clocks), so "cfg a, then cfg b" flatters b (guide section 5.4 rule 24)
        best = {c: 1e30 for c in cfgs}
        for cfg in cfgs:
            TUNE.attn_cfg = cfg
            for _ in range(10):
                fn()
        torch.cuda.synchronize()
        for _rep in range(4):
            for cfg in cfgs:
                TUNE.attn_cfg = cfg
                a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                it = 30
                fn()
                a.record()
                for _ in range(it):
                    fn()
                b.record()
                torch.cuda.synchronize()
                best[cfg] = min(best[cfg], a.elapsed_time(b) / it * 1e3)
        flops = 4.0 * N * N * H * 64 * B
        for cfg in cfgs:
            us = best[cfg]
            print("attention B=%3d H=%2d N=%4d cfg %2d  %8.1f us  %7.1f TFLOP/s  (%.1f %% of 2500)" % (B, H, N, cfg, us, flops / us / 1e6, flops / us / 1e6 / 25), flush=True)
    TUNE.attn_cfg = -1


if __name__ == "__main__":
    main()
