"""Is a 256 x 256 tile with the K extent split in two (slabs) worth building for the fc2 residual GEMM of configs[4]?  Timing stand-ins with the kernels at hand:
cfg 31 (128 x 256, the product form, in-place f32 accumulate) at the fc2 shape; cfg 30 (256 x 256) at the same shape (112 workgroups); cfg 30 at M = 2 x 6984,
K = 2048, f32 store (224 workgroups of 256 x 256 x 2048: the traffic and work per workgroup of a two-way K split writing two slabs)."""
import ctypes as C, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from uvltrack_amd import _native
lib = _native.load()
p = lambda t: C.c_void_p(t.data_ptr())
st = C.c_void_p(torch.cuda.current_stream().cuda_stream)

def run(M, N, K, cfg, acc, rot=6, pre=-1):
    xs = [torch.randn(M, K, device="cuda").bfloat16() for _ in range(rot)]
    ws = [(torch.randn(N, K, device="cuda") / K ** 0.5).bfloat16() for _ in range(rot)]
    b = torch.randn(N, device="cuda")
    ys = [torch.zeros(M, N, device="cuda") for _ in range(rot)]
    t = _native.UvlTuning(gemm_cfg=cfg, res_pre=pre)
    def f(i):
        rc = lib.uvl_linear(p(xs[i % rot]), p(ws[i % rot]), p(b), p(ys[i % rot]), M, N, K, 0, 1, acc, t.ref(), st)
        assert rc == 0
    for i in range(5): f(i)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for i in range(60): f(i)
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / 60 * 1e3

for rep in range(2):
    print("fc2 6984x1024x4096  cfg31 acc: %.1f us | cfg31 acc pre=2: %.1f | cfg30 acc: %.1f us | cfg30 store: %.1f | stand-in cfg30 13968x1024x2048 store: %.1f us | cfg31 13968x1024x2048 store %.1f" % (
        run(6984, 1024, 4096, 31, 1), run(6984, 1024, 4096, 31, 1, pre=2), run(6984, 1024, 4096, 30, 1), run(6984, 1024, 4096, 30, 0), run(13968, 1024, 2048, 30, 0), run(13968, 1024, 2048, 31, 0)), flush=True)
    print("proj 6984x1024x1024 cfg31 acc: %.1f us | cfg30 acc %.1f | stand-in cfg30 13968x1024x512 store: %.1f" % (run(6984, 1024, 1024, 31, 1), run(6984, 1024, 1024, 30, 1), run(13968, 1024, 512, 30, 0)), flush=True)
