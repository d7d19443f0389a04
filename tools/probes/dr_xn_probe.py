"""(Needs the experiment it measured: an N-part tile order in gemm_dr_body selected by gemm_gm = 100 + parts; not in the tree.  Result: profiles/r05_summary.md.)
gemm_dr tile order: grouped (8 M tiles x all N panels per group, cut into 8 runs) against N parts (uvl_tuning.gemm_gm = 100 + parts), QKV / fc1 of configs[4], isolated, interleaved."""
import ctypes as C, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from uvltrack_amd import _native
lib = _native.load()
p = lambda t: C.c_void_p(t.data_ptr())
st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
only = int(sys.argv[1]) if len(sys.argv) > 1 else None
for (M, N, K, act, what) in ((6984, 3072, 1024, 0, "QKV shape"), (6984, 4096, 1024, 1, "fc1")):
    rot = 4
    xs = [torch.randn(M, K, device="cuda").bfloat16() for _ in range(rot)]
    w = (torch.randn(N, K, device="cuda") / K ** 0.5).bfloat16(); b = torch.randn(N, device="cuda")
    wp = torch.empty_like(w); lib.uvl_pack_weight(p(w), p(wp), N, K, st)
    ys = [torch.empty(M, N, device="cuda", dtype=torch.bfloat16) for _ in range(rot)]
    res, outs = {}, {}
    for rep in range(3 if only is None else 1):
        for gm in ((-1, 102, 104, 16) if only is None else (only,)):
            t = _native.UvlTuning(gemm_cfg=36, gemm_gm=gm)
            f = lambda i: lib.uvl_linear_pk(p(xs[i % rot]), p(w), p(wp), p(b), p(ys[i % rot]), M, N, K, act, 0, 0, t.ref(), st)
            for i in range(4): assert f(i) == 0
            torch.cuda.synchronize()
            outs[gm] = ys[0].clone()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for i in range(40): f(i)
            e1.record(); torch.cuda.synchronize()
            res.setdefault(gm, []).append(e0.elapsed_time(e1) / 40 * 1e3)
    for gm in outs: assert torch.equal(outs[gm], list(outs.values())[0]), gm
    print("%-10s | " % what + " | ".join("gemm_gm %d: %s us" % (k, " ".join("%.1f" % v for v in vs)) for k, vs in res.items()), flush=True)
