"""K / tile-count sweeps of the direct-to-register GEMM (cfg 36) beside cfg 30 / 31 and hipBLASLt: separates the cost of a K tile from what a
tile pays outside its loop.  Usage (GPU box): python tools/dr_sweep.py [--lib PATH] [k|m|shapes]"""
import ctypes as C
import os
import sys

import torch
import torch.nn.functional as F

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from uvltrack_amd import _native  # noqa: E402

if "--lib" in sys.argv:
    i = sys.argv.index("--lib")
    _native.LIB_PATH = os.path.abspath(sys.argv[i + 1])
    del sys.argv[i:i + 2]
lib = _native.load()
p = lambda t: C.c_void_p(t.data_ptr())


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


def run(M, N, K, cfgs=(30, 31, 36), act=0, f32=0, vendor=True):
    st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    x = torch.randn(M, K, device="cuda").bfloat16()
    w = (torch.randn(N, K, device="cuda") / K ** 0.5).bfloat16()
    wp = torch.empty_like(w)
    lib.uvl_pack_weight(p(w), p(wp), N, K, st)
    bias = torch.randn(N, device="cuda")
    y = torch.zeros(M, N, device="cuda", dtype=torch.float32 if f32 else torch.bfloat16)
    flops = 2.0 * M * N * K
    out = []
    for cfg in cfgs:
        t = _native.UvlTuning(gemm_cfg=cfg)
        us = timeit(lambda: lib.uvl_linear_pk(p(x), p(w), p(wp), p(bias), p(y), M, N, K, act, f32, f32, t.ref(), st))
        out.append("cfg%d %6.1f us %5.0f TF" % (cfg, us, flops / us / 1e6))
    if vendor:
        bb = bias.bfloat16()
        us = timeit(lambda: F.linear(x, w, bb))
        out.append("hipBLASLt %6.1f us %5.0f TF" % (us, flops / us / 1e6))
    print("M=%6d N=%5d K=%5d act=%d f32=%d | %s" % (M, N, K, act, f32, " | ".join(out)), flush=True)


def main():
    what = sys.argv[1] if len(sys.argv) > 1 else "k"
    if what == "k":
        for K in (128, 256, 512, 1024, 2048, 4096):
            run(8192, 4096, K)              # 1024 tiles of 128 x 256 = two full rounds of 512 slots; 512 of 256 x 256 = two rounds
    elif what == "m":
        for M in (2048, 4096, 5376, 5448, 6144, 6664, 6984, 8192, 10240, 12288, 16384):
            run(M, 3072, 1024)
    else:
        for B, D, ntok in ((8, 768, 553), (32, 768, 553), (32, 1024, 681), (16, 1024, 873)):
            M = B * ntok
            for name, N, K, act, f32 in (("qkv", 3 * D, D, 0, 0), ("fc1", 4 * D, D, 1, 0), ("proj", D, D, 0, 1), ("fc2", D, 4 * D, 0, 1)):
                run(M, N, K, act=act, f32=f32)


if __name__ == "__main__":
    main()
