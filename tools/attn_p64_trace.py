"""Phase timeline of attn_p64_kernel (development aid): shader-clock stamps of wave 0 of workgroups 0 and 256 (the two that usually share
CU 0) at the phase boundaries of every key tile of their first item, from the -DATTN_WGTRACE build (tools/attn_wgtrace.py --build).
    python tools/attn_p64_trace.py B H N        (PP_VARIANT=<name> picks an ablation library)"""
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
    TUNE = _native.UvlTuning(attn_cfg=11)
    for _ in range(4):
        lib.uvl_attention(p(q), p(k), p(vt), p(add), p(o), Bn, H, N, Npad, 1, TUNE.ref(), st)
    torch.cuda.synchronize()
    n = 8192
    buf = (C.c_ulonglong * (n * 6))()
    assert lib.uvl_debug_attn_wgtrace(buf, n) == 0
    raw = np.frombuffer(buf, dtype=np.uint64).astype(np.int64)
    a = raw[300000 // 8: 300000 // 8 + 1024].reshape(2, 64, 8)
    nt = (N + 63) // 64
    t0 = a[0, 0, 0]
    names = ["M0 start", "M0 end", "V0 start", "V0 end", "M1 start", "M1 end", "barrier", "V1 end"]
    print("B=%d H=%d N=%d %s: %d key tiles; cycles since workgroup 0's first M phase" % (Bn, H, N, os.environ.get("PP_VARIANT", ""), nt))
    if "-v" in sys.argv:
        print("        " + "".join("%10s" % s for s in names))
        for h in range(2):
            for t in range(nt):
                print(" w%d t%2d " % (h, t) + "".join("%10d" % (a[h, t, k] - t0) for k in range(8)))
    if "onestamp" in os.environ.get("PP_VARIANT", ""):
        for h in range(2):
            print(" workgroup %3d: tile %.0f cycles (M0 start to M0 start, tiles 1..%d)" % (256 * h, (a[h, 2:nt, 0] - a[h, 1:nt - 1, 0]).mean(), nt - 2))
        return
    for h in range(2):
        d = a[h, 1:nt - 1]
        print(" workgroup %3d, tiles 1..%d: M0 %.0f  V0 %.0f  M1 %.0f  wait+barrier %.0f  V1 %.0f  (stamps ~%.0f); tile %.0f cycles" % (
            256 * h, nt - 2, (d[:, 1] - d[:, 0]).mean(), (d[:, 3] - d[:, 2]).mean(), (d[:, 5] - d[:, 4]).mean(), (d[:, 6] - d[:, 5]).mean(),
            (d[:, 7] - d[:, 6]).mean(), ((d[:, 2] - d[:, 1]) + (d[:, 4] - d[:, 3])).mean() / 2 * 7, (a[h, 2:nt, 0] - a[h, 1:nt - 1, 0]).mean()))


if __name__ == "__main__":
    main()
