"""Whole rounds of 256 x 256 tiles on persistent workgroups (gemm_sk = 2) beside the tile grid (cfg 30): does the output burst of a
round overlap the next round's K loop?  Usage (GPU box): python tools/sk_rounds.py"""
import ctypes as C
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from uvltrack_amd import _native  # noqa: E402

if "--lib" in sys.argv:
    i = sys.argv.index("--lib")
    _native.LIB_PATH = os.path.abspath(sys.argv[i + 1])
    del sys.argv[i:i + 2]
lib = _native.load()
p = lambda t: C.c_void_p(t.data_ptr())


def timeit(fn, iters=30):
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
    st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    nb = lib.uvl_gemm_scratch_bytes()
    scratch = torch.zeros((nb,), dtype=torch.uint8, device="cuda")
    for rounds in (1, 2, 3, 4):
        for N, K in ((4096, 1024), (4096, 256), (1024, 4096)):
            M = rounds * 256 * 256 // (N // 256)
            x = torch.randn(M, K, device="cuda").bfloat16()
            w = (torch.randn(N, K, device="cuda") / K ** 0.5).bfloat16()
            bias = torch.randn(N, device="cuda")
            out = []
            for act, f32 in ((0, 0), (1, 0), (0, 1)):
                y = torch.zeros(M, N, device="cuda", dtype=torch.float32 if f32 else torch.bfloat16)
                for label, kw in (("grid", dict(gemm_cfg=30, gemm_sk=0)), ("persist", dict(gemm_cfg=35, gemm_sk=2))):
                    t = _native.UvlTuning(**kw)
                    us = timeit(lambda: lib.uvl_linear_ws(p(x), p(w), None, p(bias), p(y), M, N, K, act, f32, f32, t.ref(), p(scratch), nb, st))
                    out.append("%s %.1f" % (label, us))
            print("rounds %d M=%5d N=%4d K=%4d | bias: %s %s | gelu: %s %s | f32acc: %s %s" % ((rounds, M, N, K) + tuple(out)), flush=True)


if __name__ == "__main__":
    main()
