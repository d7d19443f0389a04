"""A/B of the GEMM ring shapes on the batched frame's shapes: 128x128 tiles with 64-wide K stages and two ring stages (cfg 6, the
product's choice at M >= 6144) against 32-wide K stages with 3..6 stages (cfg 18, 16, 17, 19), bf16 / GELU / f32-residual
epilogues, checked against torch.   Usage (GPU box): python tools/gemm_ring_ab.py"""
import ctypes as C
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from uvltrack_amd import _native  # noqa: E402

lib = _native.load()
TUNE = _native.UvlTuning()      # per-call overrides of the launch heuristics (no process-global tuning state)
p = lambda t: C.c_void_p(t.data_ptr())
st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
CFGS = {6: "128x128 k64 ns2", 20: "128x128 ns2 +2 producers", 21: "128x128 ns2 +4 producers", 18: "128x128 k32 ns3", 16: "128x128 k32 ns4", 17: "128x128 k32 ns5", 19: "128x128 k32 ns6", 11: "256x256 k64 ns2"}
if "--short" in sys.argv:
    sys.argv.remove("--short")
    CFGS = {k: CFGS[k] for k in (6, 20, 21, 11)}


def timeit(fn, iters=30):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    best = 1e9
    for _ in range(3):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(iters):
            fn()
        b.record()
        torch.cuda.synchronize()
        best = min(best, a.elapsed_time(b) / iters * 1e3)
    return best


shapes = [("L8 qkv", 6984, 3072, 1024, 0), ("L8 fc1", 6984, 4096, 1024, 1), ("L8 proj", 6984, 1024, 1024, 0), ("L8 fc2", 6984, 1024, 4096, 0),
          ("B32 fc1", 17696, 3072, 768, 1), ("B32 fc2", 17696, 768, 3072, 0), ("B32 qkv", 17696, 2304, 768, 0)]
if len(sys.argv) > 1:
    shapes = shapes[:int(sys.argv[1])]
for name, M, N, K, act in shapes:
    x = torch.randn(M, K, device="cuda").bfloat16()
    w = (torch.randn(N, K, device="cuda") / K ** 0.5 + torch.linspace(-0.02, 0.03, N, device="cuda")[:, None]).bfloat16()
    bias = torch.randn(N, device="cuda")
    ref = x.float() @ w.float().t() + bias
    if act == 1:
        ref = torch.nn.functional.gelu(ref)
    flops = 2.0 * M * N * K
    row = []
    for cfg, label in CFGS.items():
        if (cfg == 11 and N % 256) or N % 128:
            continue
        TUNE.gemm_cfg = cfg
        y = torch.full((M, N), float("nan"), device="cuda", dtype=torch.bfloat16)
        fn = lambda: lib.uvl_linear(p(x), p(w), p(bias), p(y), M, N, K, act, 0, 0, TUNE.ref(), st)
        us = timeit(fn)
        err = (y.float() - ref).abs()
        ok = bool((err <= 1e-2 * ref.abs() + 2e-2).all())
        row.append("%s %6.1f us %6.1f TF%s" % (label, us, flops / us / 1e6, "" if ok else " WRONG(max %.3g)" % float(err.max())))
    print("%-8s M=%5d N=%4d K=%4d | %s" % (name, M, N, K, " | ".join(row)), flush=True)
TUNE.gemm_cfg = -1
