"""What does the epilogue of gemm_dr_kernel cost a launch?  Product library against a -DDR_NOEPI variant (tiles end behind the K loop; tools/probes/libuvl_dr_noepi.so),
same process, interleaved.  build (CPU): python tools/probes/dr_epi_probe.py --build      run (GPU box): python tools/probes/dr_epi_probe.py"""
import ctypes as C, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from uvltrack_amd import build as B
VAR = os.path.join(ROOT, "tools", "probes", "libuvl_dr_noepi.so")
if "--build" in sys.argv:
    B.build(force=False, verbose=False)
    obj = os.path.join(ROOT, "tools", "probes", "dr_noepi.o")
    src = os.path.join(B.CSRC, "gemm_dr.hip")
    cmd = ["/opt/rocm/bin/hipcc"] + B._flags_for("gemm_dr.hip") + ["-c", src, "-o", obj]
    subprocess.run(cmd + ["-DDR_NOEPI"], check=True)
    objs = [obj] + [os.path.join(B.HERE, "build", s.replace(".hip", ".o")) for s in B.SOURCES if s != "gemm_dr.hip"]
    subprocess.run(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-shared", "-fPIC", "-o", VAR] + objs, check=True)
    print("built", VAR); sys.exit(0)
import torch
from uvltrack_amd import _native
libs = {"product": _native.load(), "no epilogue": C.CDLL(VAR)}
p = lambda t: C.c_void_p(t.data_ptr())
st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
for (M, N, K, act, what) in ((6984, 3072, 1024, 0, "QKV shape, bf16 store"), (6984, 4096, 1024, 1, "fc1, GELU"), (6984, 4096, 1024, 0, "fc1 shape, bf16 store")):
    x = torch.randn(M, K, device="cuda").bfloat16(); w = (torch.randn(N, K, device="cuda") / K ** 0.5).bfloat16(); b = torch.randn(N, device="cuda")
    wp = torch.empty_like(w); libs["product"].uvl_pack_weight(p(w), p(wp), N, K, st)
    y = torch.empty(M, N, device="cuda", dtype=torch.bfloat16)
    t = _native.UvlTuning(gemm_cfg=36)
    res = {k: [] for k in libs}
    for rep in range(4):
        for k, lib in libs.items():
            for _ in range(5): lib.uvl_linear_pk(p(x), p(w), p(wp), p(b), p(y), M, N, K, act, 0, 0, t.ref(), st)
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(40): lib.uvl_linear_pk(p(x), p(w), p(wp), p(b), p(y), M, N, K, act, 0, 0, t.ref(), st)
            e1.record(); torch.cuda.synchronize()
            res[k].append(e0.elapsed_time(e1) / 40 * 1e3)
    tiles = ((M + 127) // 128) * (N // 256)
    print("%-24s %4d tiles | " % (what, tiles) + " | ".join("%s %s us" % (k, " ".join("%.1f" % v for v in vs)) for k, vs in res.items()), flush=True)
