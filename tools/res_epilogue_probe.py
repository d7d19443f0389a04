"""What the f32 read-modify-write epilogue of the residual GEMMs costs, and what requesting the residual rows inside the K loop (uvl_tuning.res_pre)
recovers: proj / fc2 of 8 UVLTrack-L sequences on cfg 31 (128 x 256 tiles, one round of 220 workgroups), interleaved rounds, medians.
Forms: bf16 store (no residual) | f32 store | f32 in-place, rows loaded in the epilogue (res_pre = 0) | rows requested in the K loop (default).
`--rot N`: N rotating (x, y) buffer sets so that operands and residual rows are not cache-resident from the previous iteration.
Usage (GPU box): python tools/res_epilogue_probe.py [--rot 6] [M ...]"""
import ctypes as C
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from uvltrack_amd import _native  # noqa: E402

lib = _native.load()
p = lambda t: C.c_void_p(t.data_ptr())


def main():
    args = sys.argv[1:]
    rot = 1
    if "--rot" in args:
        i = args.index("--rot")
        rot = int(args[i + 1])
        del args[i:i + 2]
    Ms = [int(a) for a in args] or [6984, 6664]
    st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    D = 1024
    for M in Ms:
        for name, N, K in (("proj", D, D), ("fc2", D, 4 * D)):
            xs = [torch.randn(M, K, device="cuda").bfloat16() for _ in range(rot)]
            w = (torch.randn(N, K, device="cuda") / K ** 0.5).bfloat16()
            bias = torch.randn(N, device="cuda")
            yf = [torch.zeros(M, N, device="cuda") for _ in range(rot)]
            yb = [torch.zeros(M, N, device="cuda", dtype=torch.bfloat16) for _ in range(rot)]
            forms = [("bf16", dict(gemm_cfg=31), 0, 0), ("f32", dict(gemm_cfg=31), 1, 0), ("rmw/epi", dict(gemm_cfg=31, res_pre=0), 1, 1),
                     ("rmw/loop", dict(gemm_cfg=31), 1, 1)]
            tunes = {f[0]: _native.UvlTuning(**f[1]) for f in forms}
            k = [0]

            def call(label, f32, acc):
                j = k[0] % rot
                k[0] += 1
                y = yf[j] if f32 else yb[j]
                lib.uvl_linear(p(xs[j]), p(w), p(bias), p(y), M, N, K, 0, f32, acc, tunes[label].ref(), st)

            res = {f[0]: [] for f in forms}
            for _ in range(5):
                for label, _, f32, acc in forms:
                    for _ in range(4):
                        call(label, f32, acc)
                    torch.cuda.synchronize()
                    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                    a.record()
                    for _ in range(24):
                        call(label, f32, acc)
                    b.record()
                    torch.cuda.synchronize()
                    res[label].append(a.elapsed_time(b) / 24 * 1e3)
            print("%-4s M=%5d N=%4d K=%4d rot=%d | %s" % (name, M, N, K, rot, "  ".join("%s %.1f" % (kk, sorted(v)[len(v) // 2]) for kk, v in res.items())), flush=True)


if __name__ == "__main__":
    main()
