"""Does fc2 find fc1's output (57 MB of bf16 rows at configs[4]) in a cache?  fc1 -> fc2 chains over rotating buffer sets (every set's inputs are cold), against the two
GEMMs alone on rotating (cold) and single (warm) operands."""
import ctypes as C, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from uvltrack_amd import _native
lib = _native.load()
p = lambda t: C.c_void_p(t.data_ptr())
st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
M, D, F = 6984, 1024, 4096
rot = 5
xs = [torch.randn(M, D, device="cuda").bfloat16() for _ in range(rot)]
hs = [torch.empty(M, F, device="cuda", dtype=torch.bfloat16) for _ in range(rot)]
rs = [torch.zeros(M, D, device="cuda") for _ in range(rot)]
w1 = (torch.randn(F, D, device="cuda") / 32).bfloat16(); b1 = torch.randn(F, device="cuda")
w2 = (torch.randn(D, F, device="cuda") / 64).bfloat16(); b2 = torch.randn(D, device="cuda")
w1p = torch.empty_like(w1); lib.uvl_pack_weight(p(w1), p(w1p), F, D, st)
w2p = torch.empty_like(w2); lib.uvl_pack_weight(p(w2), p(w2p), D, F, st)
fc1 = lambda i, j: lib.uvl_linear_pk(p(xs[i % rot]), p(w1), p(w1p), p(b1), p(hs[j % rot]), M, F, D, 1, 0, 0, None, st)
fc2 = lambda i, j: lib.uvl_linear_pk(p(hs[i % rot]), p(w2), p(w2p), p(b2), p(rs[j % rot]), M, D, F, 0, 1, 1, None, st)
def timeit(fn, n=40):
    for i in range(5): fn(i)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for i in range(n): fn(i)
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3
for rep in range(3):
    a = timeit(lambda i: fc1(i, i))
    b = timeit(lambda i: fc2(i, i))
    bw = timeit(lambda i: fc2(0, 0))
    aw = timeit(lambda i: fc1(0, 0))
    c = timeit(lambda i: (fc1(i, i), fc2(i, i)))
    print("fc1 rotating %.1f us (one buffer %.1f) | fc2 rotating %.1f us (one buffer %.1f) | fc1 -> fc2 chain, rotating sets: %.1f us = fc1 + %.1f" % (a, aw, b, bw, c, c - a), flush=True)
