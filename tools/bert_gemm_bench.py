"""Text-branch GEMMs (M = 40 rows per sequence): ring depth / tile sweep.  Usage: python tools/bert_gemm_bench.py [batch]"""
import ctypes as C
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from uvltrack_amd import _native  # noqa: E402

lib = _native.load()
TUNE = _native.UvlTuning()      # per-call overrides of the launch heuristics (no process-global tuning state)
p = lambda t: C.c_void_p(t.data_ptr())


def timeit(fn, iters=50):
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
    M = 40 * B
    st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    for name, N, K, f32 in (("bert_qkv", 2304, 768, 0), ("bert_ao", 768, 768, 1), ("bert_i", 3072, 768, 0), ("bert_o", 768, 3072, 1)):
        x = torch.randn(M, K, device="cuda").bfloat16()
        w = (torch.randn(N, K, device="cuda") / K ** 0.5).bfloat16()
        bias = torch.randn(N, device="cuda")
        y = torch.empty(M, N, device="cuda", dtype=torch.float32 if f32 else torch.bfloat16)
        res = []
        for cfg, nm in ((4, "64x64 ns3"), (0, "64x64 ns4"), (5, "64x64 ns6"), (7, "64x64 ns2")):
            TUNE.gemm_cfg = cfg
            us = timeit(lambda: lib.uvl_linear(p(x), p(w), p(bias), p(y), M, N, K, 0, f32, 0, TUNE.ref(), st))
            res.append("%s %.1f us" % (nm, us))
            if f32:
                for sk in (2, 4):
                    if (K // 64) % sk:
                        continue
                    slabs = torch.empty(sk, M, N, device="cuda")
                    us = timeit(lambda: lib.uvl_linear_splitk(p(x), p(w), p(bias), p(slabs), M, N, K, sk, TUNE.ref(), st))
                    res.append("%s sk%d %.1f us" % (nm, sk, us))
        TUNE.gemm_cfg = -1
        print("%-8s M=%d N=%d K=%d | %s" % (name, M, N, K, " | ".join(res)))


if __name__ == "__main__":
    main()
