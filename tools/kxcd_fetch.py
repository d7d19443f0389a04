"""FETCH_SIZE of the one-sequence split-K GEMMs with the tile map (gemm_kxcd 0) and the K-slice map (1):
    rocprofv3 --kernel-trace --pmc FETCH_SIZE -d out -o k --output-format csv -- python tools/kxcd_fetch.py
launches, in order: fc2 (553 x 768 x 3072, split 4) x 20 with the tile map, x 20 with the K-slice map, then proj (553 x 768 x 768, split 2) the same."""
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
for name, M, N, K, sk in (("fc2", 553, 768, 3072, 4), ("proj", 553, 768, 768, 2)):
    x = torch.randn(M, K, device="cuda").bfloat16()
    w = (torch.randn(N, K, device="cuda") / K ** 0.5).bfloat16()
    bias = torch.randn(N, device="cuda")
    slabs = torch.empty(sk, M, N, device="cuda")
    for mode in (0, 1):
        TUNE.gemm_kxcd = mode
        for _ in range(20):
            lib.uvl_linear_splitk(p(x), p(w), p(bias), p(slabs), M, N, K, sk, TUNE.ref(), st)
        torch.cuda.synchronize()
TUNE.gemm_kxcd = -1
