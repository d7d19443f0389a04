"""Do the two workgroups that share a CU in gemm_dr_kernel overlap one's epilogue with the other's K loop?  Needs a variant build with
-DGEMM_DR_TRACE (per workgroup: start, K loop done, end in 100 MHz ticks, HW_ID, XCC_ID).  One launch of the fc1 shape of 8 UVLTrack-L
sequences; per CU the intervals are listed and the share of epilogue time that ran while another workgroup of the same CU was in its K loop.
Usage (GPU box): python tools/dr_wgtrace.py uvltrack_amd/build/libuvl_drtrace.so [M N K]"""
import ctypes as C, os, sys, collections
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from uvltrack_amd import _native
_native.LIB_PATH = os.path.abspath(sys.argv[1])
lib = _native.load()
M, N, K = (int(a) for a in sys.argv[2:5]) if len(sys.argv) >= 5 else (6984, 4096, 1024)
p = lambda t: C.c_void_p(t.data_ptr())
st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
x = torch.randn(M, K, device="cuda").bfloat16(); w = (torch.randn(N, K, device="cuda") / K ** 0.5).bfloat16(); b = torch.randn(N, device="cuda")
wp = torch.empty_like(w); lib.uvl_pack_weight(p(w), p(wp), N, K, st)
y = torch.empty(M, N, device="cuda", dtype=torch.bfloat16)
t = _native.UvlTuning(gemm_cfg=36)
for _ in range(3):
    lib.uvl_linear_pk(p(x), p(w), p(wp), p(b), p(y), M, N, K, 1, 0, 0, t.ref(), st)
torch.cuda.synchronize()
ntile = ((M + 127) // 128) * (N // 256)
n = min(4096, 8 * ((ntile + 7) // 8))
buf = (C.c_ulonglong * (4 * n))()
lib.uvl_debug_dr_trace.argtypes = [C.c_void_p, C.c_int]
assert lib.uvl_debug_dr_trace(buf, n) == 0
recs = []
for i in range(n):
    t0, t1, t2, hw = buf[4 * i], buf[4 * i + 1], buf[4 * i + 2], buf[4 * i + 3]
    if t0 == 0: continue
    xcc, hwid = hw >> 32, hw & 0xffffffff
    cu = (xcc, (hwid >> 13) & 7, (hwid >> 12) & 1, (hwid >> 8) & 15)      # (XCC, SE, SH, CU)
    recs.append((cu, t0, t1, t2, i))
base = min(r[1] for r in recs)
by_cu = collections.defaultdict(list)
for cu, t0, t1, t2, i in recs:
    by_cu[cu].append(((t0 - base) / 100.0, (t1 - base) / 100.0, (t2 - base) / 100.0, i))
print("%d workgroups on %d CUs; launch span %.1f us" % (len(recs), len(by_cu), max(r[3] - base for r in recs) / 100.0))
kl = [r[2] - r[1] for r in recs]; ep = [r[3] - r[2] for r in recs]
print("K loop (incl. prologue) per workgroup: median %.1f us, epilogue: median %.1f us" % (sorted(kl)[len(kl) // 2] / 100.0, sorted(ep)[len(ep) // 2] / 100.0))
tot_ep = ov_ep = 0.0
for cu, L in by_cu.items():
    for (a0, a1, a2, _) in L:
        tot_ep += a2 - a1
        for (b0, b1, b2, _) in L:
            if (b0, b1, b2) == (a0, a1, a2): continue
            ov_ep += max(0.0, min(a2, b1) - max(a1, b0))      # my epilogue [a1, a2) against the other's K loop [b0, b1)
print("epilogue time that ran under another workgroup's K loop on the same CU: %.0f %%" % (100.0 * ov_ep / max(tot_ep, 1e-9)))
for cu in sorted(by_cu)[:3]:
    print("CU", cu, " ".join("[%d: %.1f|%.1f|%.1f]" % (i, a, b_, c) for a, b_, c, i in sorted(by_cu[cu])))
