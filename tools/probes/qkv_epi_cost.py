"""What the QKV scatter epilogue (q, k rows + V^T token-contiguous) costs beside the plain bf16 store, same GEMM, cfg 36 / 31.
Usage (GPU box): python tools/probes/qkv_epi_cost.py"""
import ctypes as C, os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from uvltrack_amd import _native
if "--lib" in sys.argv:
    _native.LIB_PATH = os.path.abspath(sys.argv[sys.argv.index("--lib") + 1])
lib = _native.load()
p = lambda t: C.c_void_p(t.data_ptr())
st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
B, N, D = 8, 873, 1024
Npad = 896
M = B * N
x = torch.randn(M, D, device="cuda").bfloat16(); w = (torch.randn(3 * D, D, device="cuda") / 32).bfloat16(); b = torch.randn(3 * D, device="cuda")
wp = torch.empty_like(w); lib.uvl_pack_weight(p(w), p(wp), 3 * D, D, st)
y = torch.empty(M, 3 * D, device="cuda", dtype=torch.bfloat16)
q = torch.zeros(B, D // 64, Npad, 64, device="cuda", dtype=torch.bfloat16); k = torch.zeros_like(q); vt = torch.zeros(B, D // 64, 64, Npad, device="cuda", dtype=torch.bfloat16)
def timeit(fn, it=30):
    for _ in range(5): fn()
    torch.cuda.synchronize(); a, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True); a.record()
    for _ in range(it): fn()
    e.record(); torch.cuda.synchronize(); return a.elapsed_time(e) / it * 1e3
for cfg in (36, 31):
    t = _native.UvlTuning(gemm_cfg=cfg)
    res = {"plain": [], "qkv": []}
    for _ in range(3):
        res["plain"].append(timeit(lambda: lib.uvl_linear_pk(p(x), p(w), p(wp), p(b), p(y), M, 3 * D, D, 0, 0, 0, t.ref(), st)))
        res["qkv"].append(timeit(lambda: lib.uvl_qkv_project_pk(p(x), p(w), p(wp), p(b), p(q), p(k), p(vt), B, N, Npad, D, C.c_float(0.18), t.ref(), st)))
    print("cfg %d: plain bf16 store %.1f us, QKV scatter %.1f us" % (cfg, sorted(res["plain"])[1], sorted(res["qkv"])[1]))
