"""The GEMM forms of many-sequence frames (cfg 30 / 31 tile grids, cfg 36 direct-to-register; round 4 also timed the since-removed
split-tile schedule with this script, profiles/r04_sk_bench.txt) beside hipBLASLt on the shapes of 8 UVLTrack-L sequences: bias epilogue and
the frame's epilogue, three interleaved rounds, medians.  SK_ONLY=c31,dr restricts the forms; --lib PATH loads a variant build.
Usage (GPU box): python tools/gemm_forms_bench.py [M ...]"""
import ctypes as C
import os
import sys

import torch
import torch.nn.functional as F

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from uvltrack_amd import _native  # noqa: E402

if "--lib" in sys.argv:                      # a variant build of the library (development A/B)
    i = sys.argv.index("--lib")
    _native.LIB_PATH = os.path.abspath(sys.argv[i + 1])
    del sys.argv[i:i + 2]
lib = _native.load()


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
    Ms = [int(a) for a in sys.argv[1:]] or [6664, 6984, 5448]
    st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    D = 1024
    for M in Ms:
        for name, N, K, act, f32 in (("qkv", 3 * D, D, 0, 0), ("fc1", 4 * D, D, 1, 0), ("proj", D, D, 0, 1), ("fc2", D, 4 * D, 0, 1)):
            x = torch.randn(M, K, device="cuda").bfloat16()
            w = (torch.randn(N, K, device="cuda") / K ** 0.5).bfloat16()
            bias = torch.randn(N, device="cuda")
            wp = torch.empty_like(w)
            lib.uvl_pack_weight(p(w), p(wp), N, K, st)
            y = torch.zeros(M, N, device="cuda", dtype=torch.float32 if f32 else torch.bfloat16)
            flops = 2.0 * M * N * K
            forms = (("auto", {}), ("c30", dict(gemm_cfg=30)), ("c31", dict(gemm_cfg=31)), ("dr", dict(gemm_cfg=36)),
                     ("c31wt", dict(gemm_cfg=31, res_store=2)), ("drwt", dict(gemm_cfg=36, res_store=2)))
            if os.environ.get("SK_ONLY"):
                forms = [f for f in forms if f[0] in os.environ["SK_ONLY"].split(",")]
            cases = []
            for label, kw in forms:
                t = _native.UvlTuning(**kw)
                for ep_label, a_, f_, acc_ in (("bias", 0, 0, 0), ("frame", act, f32, f32)):
                    yy = y if f_ == f32 else torch.zeros(M, N, device="cuda", dtype=torch.float32 if f_ else torch.bfloat16)
                    cases.append(("%s/%s" % (label, ep_label), (lambda t=t, yy=yy, a_=a_, f_=f_, acc_=acc_: lib.uvl_linear_pk(p(x), p(w), p(wp), p(bias), p(yy), M, N, K, a_, f_, acc_, t.ref(), st))))
            # three interleaved rounds, median: the order of the cases and the clock the part grants do not favour one of them
            res = {c[0]: [] for c in cases}
            for _ in range(3):
                for name_, fn in cases:
                    res[name_].append(timeit(fn, 25))
            row = ["%s %.1f" % (k, sorted(v)[1]) for k, v in res.items()]
            bb = bias.bfloat16()
            ven = timeit(lambda: F.linear(x, w, bb))
            print("%-4s M=%5d N=%4d K=%4d | %s | hipBLASLt(bias) %.1f us %.0f TF" % (name, M, N, K, "  ".join(row), ven, flops / ven / 1e6), flush=True)


if __name__ == "__main__":
    main()
