"""Phase timeline of attn_pp_kernel (development aid): shader-clock stamps of wave 0 (half 0) and wave 4 (half 1) of workgroup 0 at
the start / end of every phase of their first item, from the -DATTN_WGTRACE build (tools/attn_wgtrace.py --build).
    python tools/attn_pp_trace.py B H N"""
import ctypes as C
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
LIB = os.path.join(ROOT, "tools", "probes", "libuvl_wgtrace%s.so" % (("_" + os.environ["PP_VARIANT"]) if os.environ.get("PP_VARIANT") else ""))


def main():
    import torch
    from uvltrack_amd import _native
    lib = C.CDLL(LIB)
    Bn, H, N = (int(a) for a in sys.argv[1:4])
    Npad = (N + 63) // 64 * 64
    q = (torch.randn(Bn, H, Npad, 64, device="cuda") * 0.18033688).bfloat16()
    k = torch.randn(Bn, H, Npad, 64, device="cuda").bfloat16()
    vt = torch.randn(Bn, H, 64, Npad, device="cuda").bfloat16()
    add = torch.zeros(Bn, Npad, device="cuda")
    o = torch.empty(Bn * N, H * 64, device="cuda", dtype=torch.bfloat16)
    p = lambda t: C.c_void_p(t.data_ptr())
    st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    TUNE = _native.UvlTuning(attn_cfg=12)
    for _ in range(4):
        lib.uvl_attention(p(q), p(k), p(vt), p(add), p(o), Bn, H, N, Npad, 1, TUNE.ref(), st)
    torch.cuda.synchronize()
    n = 1024 // 6 + 1
    buf = (C.c_ulonglong * (n * 6))()
    assert lib.uvl_debug_attn_wgtrace(buf, n) == 0
    a = np.frombuffer(buf, dtype=np.uint64)[:1024].astype(np.int64).reshape(2, 64, 8)
    nt = (N + 63) // 64
    t0 = a[0, 0, 0]
    names = ["M0 start", "M0 end", "V0 start", "V0 end", "M1 start", "M1 end", "V1 start", "V1 end"]
    print("B=%d H=%d N=%d %s: %d key tiles; cycles since half 0's first M phase" % (Bn, H, N, os.environ.get("PP_VARIANT", ""), nt))
    if "-v" in sys.argv:
        print("        " + "".join("%10s" % s for s in names))
        for h in range(2):
            for t in range(nt):
                print(" h%d t%2d " % (h, t) + "".join("%10d" % (a[h, t, k] - t0) for k in range(8)))
    for h in range(2):
        d = a[h, 1:nt - 1]
        print(" half %d, tiles 1..%d: M0 %.0f  (barrier %.0f)  V0 %.0f  (barrier %.0f)  M1 %.0f  (barrier %.0f)  V1 %.0f  (barrier to next M0 %.0f); tile %.0f cycles" % (
            h, nt - 2, (d[:, 1] - d[:, 0]).mean(), (d[:, 2] - d[:, 1]).mean(), (d[:, 3] - d[:, 2]).mean(), (d[:, 4] - d[:, 3]).mean(),
            (d[:, 5] - d[:, 4]).mean(), (d[:, 6] - d[:, 5]).mean(), (d[:, 7] - d[:, 6]).mean(), (a[h, 2:nt, 0] - a[h, 1:nt - 1, 7]).mean(),
            (a[h, 2:nt, 0] - a[h, 1:nt - 1, 0]).mean()))


if __name__ == "__main__":
    main()
