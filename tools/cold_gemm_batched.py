"""Why does fc1 of the UVLTrack-L x 8 frame take 73 us when the same launch takes 58 us in a tight loop?  Times uvl_linear on the frame's
shape with (a) the same buffers every launch (everything stays in the 256-MB memory-side cache), (b) the OUTPUT rotating over 8 buffers
(458 MB: every launch writes lines the cache does not hold), (c) the INPUTS rotating (activations + weights cold), (d) both.
    python tools/cold_gemm_batched.py [M N K act]"""
import ctypes as C
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from uvltrack_amd import _native  # noqa: E402

lib = _native.load()
TUNE = _native.UvlTuning()
p = lambda t: C.c_void_p(t.data_ptr())
st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
M, N, K, act = (int(a) for a in sys.argv[1:5]) if len(sys.argv) > 4 else (6984, 4096, 1024, 1)
R = 8
xs = [torch.randn(M, K, device="cuda").bfloat16() for _ in range(R)]
ws = [(torch.randn(N, K, device="cuda") / K ** 0.5).bfloat16() for _ in range(R)]
bias = torch.randn(N, device="cuda")
ys = [torch.empty(M, N, device="cuda", dtype=torch.bfloat16) for _ in range(R)]


def run(rot_in, rot_out, iters=40):
    for i in range(R):
        lib.uvl_linear(p(xs[i if rot_in else 0]), p(ws[i if rot_in else 0]), p(bias), p(ys[i if rot_out else 0]), M, N, K, act, 0, 0, TUNE.ref(), st)
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for i in range(iters):
        j = i % R
        lib.uvl_linear(p(xs[j if rot_in else 0]), p(ws[j if rot_in else 0]), p(bias), p(ys[j if rot_out else 0]), M, N, K, act, 0, 0, TUNE.ref(), st)
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / iters * 1e3


flops = 2.0 * M * N * K
for name, ri, ro in (("same buffers", False, False), ("output rotating (8 x %.0f MB)" % (M * N * 2 / 1e6), False, True),
                     ("inputs rotating", True, False), ("inputs and output rotating", True, True), ("same buffers", False, False)):
    us = min(run(ri, ro) for _ in range(3))
    print("%d x %d x %d act %d, %-34s %.1f us  %.0f TFLOP/s" % (M, N, K, act, name, us, flops / us / 1e6))
