"""Micro-benchmark: GEMM with LayerNorm folded in (uvl_linear_ln / uvl_linear_residual) beside the plain kernels.
Usage (GPU box): python tools/lnfold_bench.py"""
import ctypes as C
import sys

import torch

sys.path.insert(0, __import__("os").path.dirname(__import__("os").path.dirname(__import__("os").path.abspath(__file__))))
from uvltrack_amd import _native  # noqa: E402

lib = _native.load()


def p(t):
    return C.c_void_p(t.data_ptr()) if t is not None else None


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
    for M, D in ((5448, 1024), (7304, 1024), (17696, 768)):
        for name, N in (("qkv", 3 * D), ("fc1", 4 * D)):
            x = torch.randn(M, D, device="cuda") * 2 + 0.3
            w = torch.randn(N, D, device="cuda") / D ** 0.5
            b = torch.randn(N, device="cuda")
            g = 1 + 0.1 * torch.randn(D, device="cuda")
            be = 0.1 * torch.randn(D, device="cuda")
            wf = torch.empty(N, D, dtype=torch.bfloat16, device="cuda")
            bf = torch.empty(N, device="cuda")
            cs = torch.empty(N, device="cuda")
            lib.uvl_fold_ln_linear(p(w), p(b), p(g), p(be), p(wf), p(bf), p(cs), N, D, st)
            xb = torch.empty(M, D, dtype=torch.bfloat16, device="cuda")
            stt = torch.empty(M, D // 64, 2, device="cuda")
            lib.uvl_row_stats(p(x), p(xb), p(stt), M, D, st)
            xn = torch.nn.functional.layer_norm(x, (D,), g, be).bfloat16()
            wb = w.bfloat16()
            y = torch.empty(M, N, dtype=torch.bfloat16, device="cuda")
            act = 1 if name == "fc1" else 0
            fl = 2.0 * M * N * D
            t0 = timeit(lambda: lib.uvl_linear(p(xn), p(wb), p(b), p(y), M, N, D, act, 0, 0, st))
            t1 = timeit(lambda: lib.uvl_linear_ln(p(xb), p(stt), p(wf), p(bf), p(cs), C.c_float(1e-6), p(y), M, N, D, act, st))
            t2 = timeit(lambda: lib.uvl_linear(p(xb), p(wf), p(bf), p(y), M, N, D, act, 0, 0, st))
            print("%-4s M=%5d N=%4d K=%4d  plain %6.1f us %6.1f TF | ln-fold %6.1f us %6.1f TF | plain kernel on the raw rows %6.1f us" % (name, M, N, D, t0, fl / t0 / 1e6, t1, fl / t1 / 1e6, t2))
        for name, K in (("proj", D), ("fc2", 4 * D)):
            a = torch.randn(M, K, device="cuda").bfloat16()
            w = (torch.randn(D, K, device="cuda") / K ** 0.5).bfloat16()
            b = torch.randn(D, device="cuda")
            x = torch.randn(M, D, device="cuda")
            xb = torch.empty(M, D, dtype=torch.bfloat16, device="cuda")
            stt = torch.empty(M, D // 64, 2, device="cuda")
            fl = 2.0 * M * D * K
            t0 = timeit(lambda: lib.uvl_linear(p(a), p(w), p(b), p(x), M, D, K, 0, 1, 1, st))
            t1 = timeit(lambda: lib.uvl_linear_residual(p(a), p(w), p(b), p(x), p(xb), p(stt), None, None, M, 0, M, D, K, st))
            print("%-4s M=%5d N=%4d K=%4d  x+= %6.1f us %6.1f TF | x+= with rows+stats %6.1f us %6.1f TF" % (name, M, D, K, t0, fl / t0 / 1e6, t1, fl / t1 / 1e6))


if __name__ == "__main__":
    main()
