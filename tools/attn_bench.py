"""Micro-benchmark of the fused attention kernel (QK^T + softmax + PV, head_dim 64) on the frame's shapes."""
import ctypes as C
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from uvltrack_amd import _native  # noqa: E402

lib = _native.load()


def p(t):
    return C.c_void_p(t.data_ptr())


def main():
    st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    shapes = [(1, 12, 553), (8, 12, 553), (64, 12, 553), (1, 16, 681), (8, 16, 681), (8, 16, 873), (64, 16, 681)]
    if len(sys.argv) > 3:
        shapes = [(int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3]))]
    if len(sys.argv) > 4:
        lib.uvl_tune_set(b"attn_cfg", int(sys.argv[4]))
    for B, H, N in shapes:
        Npad = (N + 63) // 64 * 64
        q = torch.randn(B, H, Npad, 64, device="cuda").bfloat16()
        k = torch.randn(B, H, Npad, 64, device="cuda").bfloat16()
        vt = torch.randn(B, H, 64, Npad, device="cuda").bfloat16()
        add = torch.zeros(B, Npad, device="cuda")
        o = torch.empty(B * N, H * 64, device="cuda", dtype=torch.bfloat16)
        fn = lambda: lib.uvl_attention(p(q), p(k), p(vt), p(add), p(o), B, H, N, Npad, st)
        for _ in range(5):
            fn()
        torch.cuda.synchronize()
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        it = 30
        a.record()
        for _ in range(it):
            fn()
        b.record()
        torch.cuda.synchronize()
        us = a.elapsed_time(b) / it * 1e3
        flops = 4.0 * N * N * H * 64 * B
        print("attention B=%3d H=%2d N=%4d  %8.1f us  %7.1f TFLOP/s  (%.1f %% of 2500)" % (B, H, N, us, flops / us / 1e6, flops / us / 1e6 / 25))


if __name__ == "__main__":
    main()
