"""fc2 / proj of configs[4] with the in-place f32 residual epilogue: the eight-wave 128 x 256 kernel (cfg 31, the product choice) against gemm_dr_kernel<1> (cfg 36,
uvl_tuning.gemm_dr = 1), isolated, interleaved, buffers rotating."""
import ctypes as C, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from uvltrack_amd import _native
lib = _native.load()
p = lambda t: C.c_void_p(t.data_ptr())
st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
for (M, N, K, what) in ((6984, 1024, 4096, "fc2"), (6984, 1024, 1024, "proj"), (6664, 1024, 4096, "fc2 (7 x 952)")):
    rot = 4
    xs = [torch.randn(M, K, device="cuda").bfloat16() for _ in range(rot)]
    w = (torch.randn(N, K, device="cuda") / K ** 0.5).bfloat16(); b = torch.randn(N, device="cuda")
    wp = torch.empty_like(w); lib.uvl_pack_weight(p(w), p(wp), N, K, st)
    ys = [torch.zeros(M, N, device="cuda") for _ in range(rot)]
    res = {}
    for rep in range(3):
        for name, t in (("cfg 31", _native.UvlTuning(gemm_dr=0)), ("cfg 31 pre=2", _native.UvlTuning(gemm_dr=0, res_pre=2)), ("cfg 36", _native.UvlTuning(gemm_dr=1))):
            f = lambda i: lib.uvl_linear_pk(p(xs[i % rot]), p(w), p(wp), p(b), p(ys[i % rot]), M, N, K, 0, 1, 1, t.ref(), st)
            for i in range(5): assert f(i) == 0
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for i in range(40): f(i)
            e1.record(); torch.cuda.synchronize()
            res.setdefault(name, []).append(e0.elapsed_time(e1) / 40 * 1e3)
    print("%-14s M=%d N=%d K=%d | " % (what, M, N, K) + " | ".join("%s: %s us" % (k, " ".join("%.1f" % v for v in vs)) for k, vs in res.items()), flush=True)
