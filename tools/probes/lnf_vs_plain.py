"""Isolated launch time of the LayerNorm-folded consumer GEMM (uvl_linear_lnf) against the plain 64 x 64 GEMM it replaces (uvl_linear), same shapes, dependent chain
of launches on one stream.  Usage (GPU box): python tools/probes/lnf_vs_plain.py"""
import ctypes as C
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from uvltrack_amd import _native

if os.environ.get('UVL_LIB'):
    _native.LIB_PATH = os.path.abspath(os.environ['UVL_LIB'])
lib = _native.load()
p = lambda t: C.c_void_p(t.data_ptr())
st = lambda: C.c_void_p(torch.cuda.current_stream().cuda_stream)


def bench(fn, n=300):
    for _ in range(20):
        fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / n * 1e3


for (M, N, K) in [(553, 3072, 768), (553, 2304, 768), (873, 4096, 1024), (873, 3072, 1024), (40, 3072, 768)]:
    x = torch.randn(M, K, device="cuda")
    xb = x.bfloat16()
    w = (torch.randn(N, K, device="cuda") / K ** 0.5).bfloat16()
    b = torch.randn(N, device="cuda")
    cs = w.float().sum(-1).contiguous()
    r = x.reshape(M, K // 64, 2, 32)
    stt = torch.stack([r.sum(-1), (r * r).sum(-1)], -1).permute(1, 0, 2, 3).contiguous()
    y = torch.empty(M, N, dtype=torch.bfloat16, device="cuda")
    t_plain = bench(lambda: lib.uvl_linear(p(xb), p(w), p(b), p(y), M, N, K, 1, 0, 0, None, st()))
    t_lnf = bench(lambda: lib.uvl_linear_lnf(p(xb), p(stt), p(w), p(b), p(cs), C.c_float(1e-6), p(y), M, N, K, 1, None, st()))
    print("M %4d N %4d K %4d   plain %.2f us   lnf %.2f us   (%+.2f)" % (M, N, K, t_plain, t_lnf, t_lnf - t_plain), flush=True)
