"""The f32 read-modify-write epilogue on WARM against COLD residual rows: the same C buffer every launch (stays in the memory-side cache) against
12 rotating buffers (343 MB), cfg 31 / cfg 36, write-through stores as in the frame.  Usage (GPU box): python tools/probes/rmw_cold.py"""
import ctypes as C, os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from uvltrack_amd import _native
lib = _native.load()
p = lambda t: C.c_void_p(t.data_ptr())
st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
M, R = 6984, 12
def timeit(fn, it=36):
    for i in range(R): fn(i)
    torch.cuda.synchronize(); a, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True); a.record()
    for i in range(it): fn(i)
    e.record(); torch.cuda.synchronize(); return a.elapsed_time(e) / it * 1e3
for name, N, K in (("proj", 1024, 1024), ("fc2", 1024, 4096)):
    xs = [torch.randn(M, K, device="cuda").bfloat16() for _ in range(R)]
    w = (torch.randn(N, K, device="cuda") / K ** 0.5).bfloat16(); b = torch.randn(N, device="cuda")
    wp = torch.empty_like(w); lib.uvl_pack_weight(p(w), p(wp), N, K, st)
    ys = [torch.zeros(M, N, device="cuda") for _ in range(R)]
    out = []
    for cfg in (31, 36):
        t = _native.UvlTuning(gemm_cfg=cfg, res_store=2)
        res = {}
        for label, rot in (("warm", 0), ("cold", 1)):
            res[label] = sorted(timeit(lambda i: lib.uvl_linear_pk(p(xs[i % R if rot else 0]), p(w), p(wp), p(b), p(ys[i % R if rot else 0]), M, N, K, 0, 1, 1, t.ref(), st)) for _ in range(3))[1]
        out.append("cfg %d warm %.1f cold %.1f" % (cfg, res["warm"], res["cold"]))
    print(name, " | ".join(out), flush=True)
