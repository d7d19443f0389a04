import ctypes as C, os, sys, math, torch
sys.path.insert(0, "/root/repo")
from uvltrack_amd import _native
if "--lib" in sys.argv:
    i = sys.argv.index("--lib"); _native.LIB_PATH = os.path.abspath(sys.argv[i + 1])
lib = _native.load()
p = lambda t: C.c_void_p(t.data_ptr())
st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
ok = True
for M, N, K in ((777, 512, 192), (300, 256, 64), (6200, 3072, 768), (513, 768, 448), (6984, 1024, 4096), (130, 256, 1024), (1000, 256, 128), (999, 256, 320)):
    g = torch.Generator().manual_seed(1)
    x = torch.randn(M, K, generator=g).cuda().bfloat16(); w = (torch.randn(N, K, generator=g) / math.sqrt(K)).cuda().bfloat16(); b = torch.randn(N, generator=g).cuda()
    wp = torch.empty_like(w); lib.uvl_pack_weight(p(w), p(wp), N, K, st)
    y = torch.full((M, N), float("nan"), dtype=torch.bfloat16, device="cuda")
    t = _native.UvlTuning(gemm_cfg=36)
    rc = lib.uvl_linear_pk(p(x), p(w), p(wp), p(b), p(y), M, N, K, 0, 0, 0, t.ref(), st); torch.cuda.synchronize()
    ref = x.float() @ w.float().t() + b
    err = (y.float() - ref).abs(); good = bool((err <= 1e-2 * ref.abs() + 2e-2).all())
    ok &= good
    print(M, N, K, rc, "ok" if good else "BAD %g" % float(err.max()))
print("ALL OK" if ok else "FAILED")
