"""Micro-benchmark of the GEMM kernel on the frame's shapes: tile configuration x split-K sweep.
Usage (GPU box): python tools/gemm_bench.py [batch]"""
import ctypes as C
import sys

import torch

sys.path.insert(0, __import__("os").path.dirname(__import__("os").path.dirname(__import__("os").path.abspath(__file__))))
from uvltrack_amd import _native  # noqa: E402

lib = _native.load()
TUNE = _native.UvlTuning()      # per-call overrides of the launch heuristics (no process-global tuning state)
CFG = {0: "64x64 ns4", 4: "64x64 ns3", 7: "64x64 ns2", 8: "128x64 ns3", 9: "128x64 ns2", 10: "64x128 ns2", 2: "128x128 ns3", 6: "128x128 ns2"}


def p(t):
    return C.c_void_p(t.data_ptr())


def timeit(fn, iters=40):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(iters):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / iters * 1e3


def main():
    B = int(sys.argv[1]) if len(sys.argv) > 1 else 1
    M = 553 * B
    st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    shapes = [("qkv", M, 2304, 768, "bf16"), ("fc1", M, 3072, 768, "bf16"), ("proj", M, 768, 768, "f32"), ("fc2", M, 768, 3072, "f32")]
    for name, M_, N, K, kind in shapes:
        x = torch.randn(M_, K, device="cuda").bfloat16()
        w = (torch.randn(N, K, device="cuda") / K ** 0.5).bfloat16()
        bias = torch.randn(N, device="cuda")
        flops = 2.0 * M_ * N * K
        for cfg in CFG:
            if cfg in (2, 3, 6, 10) and N % 128:
                continue
            TUNE.gemm_cfg = cfg
            res = []
            if kind == "bf16":
                y = torch.empty(M_, N, device="cuda", dtype=torch.bfloat16)
                us = timeit(lambda: lib.uvl_linear(p(x), p(w), p(bias), p(y), M_, N, K, 0, 0, 0, TUNE.ref(), st))
                res.append("sk1 %6.1f us %6.1f TF" % (us, flops / us / 1e6))
            else:
                for sk in (1, 2, 4):
                    if (K // 64) % sk:
                        continue
                    slabs = torch.empty(sk, M_, N, device="cuda")
                    us = timeit(lambda: lib.uvl_linear_splitk(p(x), p(w), p(bias), p(slabs), M_, N, K, sk, TUNE.ref(), st))
                    res.append("sk%d %6.1f us %6.1f TF" % (sk, us, flops / us / 1e6))
            print("%-5s M=%5d N=%4d K=%4d  %-12s %s" % (name, M_, N, K, CFG[cfg], " | ".join(res)))
    TUNE.gemm_cfg = -1


if __name__ == "__main__":
    main()
