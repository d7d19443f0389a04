"""Batched-regime yardstick: this library's GEMM / attention kernels beside the vendor libraries reached through
torch (hipBLASLt matmul, SDPA) on the frame's shapes.  Measurement aid only -- the product never calls torch math.
Usage (GPU box): python tools/lib_compare.py [batch ...] | cfg4"""
import ctypes as C
import os
import sys

import torch
import torch.nn.functional as F

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from uvltrack_amd import _native  # noqa: E402

lib = _native.load()
TUNE = _native.UvlTuning()      # per-call overrides of the launch heuristics (no process-global tuning state)
CFGS = (4, 7, 9, 10, 6, 11, 30, 31, 36)
GMS = (0, 8)


def p(t):
    return C.c_void_p(t.data_ptr())


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


def gemms(B, D, ntok):
    st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    M = ntok * B
    for name, N, K in (("qkv", 3 * D, D), ("fc1", 4 * D, D), ("proj", D, D), ("fc2", D, 4 * D)):
        x = torch.randn(M, K, device="cuda").bfloat16()
        w = (torch.randn(N, K, device="cuda") / K ** 0.5).bfloat16()
        wp = torch.empty_like(w)                      # the fragment-native image cfg 36 loads from (a model handle keeps one per ViT weight)
        lib.uvl_pack_weight(p(w), p(wp), N, K, st)
        bias = torch.randn(N, device="cuda")
        y = torch.empty(M, N, device="cuda", dtype=torch.bfloat16)
        flops = 2.0 * M * N * K
        call = lambda: lib.uvl_linear_pk(p(x), p(w), p(wp), p(bias), p(y), M, N, K, 0, 0, 0, TUNE.ref(), st)
        bb = bias.bfloat16()
        vendor = lambda: F.linear(x, w, bb)
        # "ours auto" and hipBLASLt: three interleaved rounds, median (neither gets the cooler chip)
        TUNE.gemm_cfg = -1
        TUNE.gemm_gm = -1
        ta, tv = [], []
        for _ in range(3):
            ta.append(timeit(call, 25))
            tv.append(timeit(vendor, 25))
        mine, ven = sorted(ta)[1], sorted(tv)[1]
        best = (mine, "auto")
        allc = []
        for cfg in CFGS:
            if (cfg in (2, 3, 6, 10, 12, 13, 15) and N % 128) or (cfg in (11, 14, 30, 31, 36) and N % 256):
                continue
            TUNE.gemm_cfg = cfg
            for gm in GMS:
                TUNE.gemm_gm = gm
                us = timeit(call)
                allc.append("%d/g%d:%.0f" % (cfg, gm, flops / us / 1e6))
                if us < best[0]:
                    best = (us, "cfg%d/g%d" % (cfg, gm))
        TUNE.gemm_gm = -1
        TUNE.gemm_cfg = -1
        print("gemm %-4s M=%6d N=%5d K=%5d | ours auto %7.1f us %6.1f TF | ours best %-9s %7.1f us %6.1f TF | hipBLASLt %7.1f us %6.1f TF"
              % (name, M, N, K, mine, flops / mine / 1e6, best[1], best[0], flops / best[0] / 1e6, ven, flops / ven / 1e6), " ".join(allc), flush=True)


def attn(B, H, N):
    st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    Npad = (N + 63) // 64 * 64
    q = (torch.randn(B, H, Npad, 64, device="cuda") * 0.18033688).bfloat16()     # pre-scaled by log2(e)/8, as the frame's QKV GEMM leaves it
    k = torch.randn(B, H, Npad, 64, device="cuda").bfloat16()
    vt = torch.randn(B, H, 64, Npad, device="cuda").bfloat16()
    add = torch.zeros(B, Npad, device="cuda")
    o = torch.empty(B * N, H * 64, device="cuda", dtype=torch.bfloat16)
    flops = 4.0 * N * N * H * 64 * B
    TUNE.attn_cfg = -1
    mine = timeit(lambda: lib.uvl_attention(p(q), p(k), p(vt), p(add), p(o), B, H, N, Npad, 1, TUNE.ref(), st))
    q2, k2, v2 = (torch.randn(B, H, N, 64, device="cuda").bfloat16() for _ in range(3))
    ven = timeit(lambda: F.scaled_dot_product_attention(q2, k2, v2))
    print("attn B=%3d H=%2d N=%4d | ours %7.1f us %6.1f TF | torch SDPA (no mask) %7.1f us %6.1f TF"
          % (B, H, N, mine, flops / mine / 1e6, ven, flops / ven / 1e6), flush=True)


def main():
    if sys.argv[1:2] == ["cfg4"]:
        # the per-GPU shard of BASELINE configs[4] at the shapes its GEMMs run: 8 UVLTrack-L sequences at z256/x384 = 8 x 833 rows
        # in the visual layers (M = 6664) and 8 x 873 rows in the fusion layers (M = 6984); M = 5448 = 8 x 681 (z128) beside them
        for ntok in (833, 873, 681):
            gemms(8, 1024, ntok)
        attn(8, 16, 833)
        attn(8, 16, 873)
        return
    batches = [int(a) for a in sys.argv[1:]] or [8, 32]
    for B in batches:
        gemms(B, 768, 553)
        gemms(B, 1024, 681)
        attn(B, 12, 553)
        attn(B, 16, 681)
        attn(B, 16, 873)


if __name__ == "__main__":
    main()
