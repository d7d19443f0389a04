"""QKV projection of 8 UVLTrack-L sequences on gemm_dr_kernel<2>: product vs a variant library (timing probes).  python tools/probes/qkv_probe.py libA.so libB.so"""
import ctypes as C, sys, os
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from uvltrack_amd import _native
libs = {os.path.basename(p): C.CDLL(os.path.abspath(p)) for p in sys.argv[1:]}
p = lambda t: C.c_void_p(t.data_ptr())
st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
D, H = 1024, 16
for M, T in ((6984, 873), (6664, 833)):
    Npad = (T + 63) // 64 * 64
    x = torch.randn(M, D, device="cuda").bfloat16(); w = (torch.randn(3 * D, D, device="cuda") / 32).bfloat16(); b = torch.randn(3 * D, device="cuda")
    wp = torch.empty_like(w)
    q = torch.zeros(8, H, Npad, 64, device="cuda", dtype=torch.bfloat16); k = torch.zeros_like(q); vt = torch.zeros(8, H, 64, Npad, device="cuda", dtype=torch.bfloat16)
    t = _native.UvlTuning(gemm_cfg=36)
    res = {n: [] for n in libs}
    for n, lib in libs.items():
        lib.uvl_pack_weight(p(w), p(wp), 3 * D, D, st)
    for _ in range(5):
        for n, lib in libs.items():
            f = lambda: lib.uvl_qkv_project_pk(p(x), p(w), p(wp), p(b), p(q), p(k), p(vt), 8, T, Npad, D, C.c_float(0.18033688), t.ref(), st)
            for _ in range(3): f()
            torch.cuda.synchronize()
            a, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record()
            for _ in range(20): f()
            e.record(); torch.cuda.synchronize()
            res[n].append(a.elapsed_time(e) / 20 * 1e3)
    print("M=%d: " % M + "  ".join("%s %.2f us" % (n, sorted(v)[len(v) // 2]) for n, v in res.items()), flush=True)
