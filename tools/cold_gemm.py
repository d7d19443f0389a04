"""How much of an in-frame batch-1 GEMM is cold weights?  Times uvl_linear (fc1 shape) after (a) nothing (weights L2/MALL warm),
(b) a 64 MB fill (evicts the 32 MB of L2, mostly keeps the 256 MB memory-side cache), (c) a 1 GB fill (evicts everything)."""
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
M, N, K = 553, 3072, 768
x = torch.randn(M, K, device="cuda").bfloat16()
w = (torch.randn(N, K, device="cuda") / K ** 0.5).bfloat16()
bias = torch.randn(N, device="cuda")
y = torch.empty(M, N, device="cuda", dtype=torch.bfloat16)
small = torch.empty(64 << 20, dtype=torch.uint8, device="cuda")
big = torch.empty(1 << 30, dtype=torch.uint8, device="cuda")


def run(evict):
    ts = []
    for _ in range(30):
        if evict is not None:
            evict.fill_(1)
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        lib.uvl_linear(p(x), p(w), p(bias), p(y), M, N, K, 1, 0, 0, TUNE.ref(), st)
        b.record()
        torch.cuda.synchronize()
        ts.append(a.elapsed_time(b) * 1e3)
    ts.sort()
    return ts[len(ts) // 2]


for name, ev in (("warm", None), ("after 64 MB fill (L2 evicted)", small), ("after 1 GB fill (all caches evicted)", big), ("warm", None)):
    print("fc1 553x3072x768, %-38s median event-pair time %.1f us" % (name, run(ev)))
