"""Whole-chip workgroup timeline of attn_w64_kernel (development aid): builds a -DATTN_WGTRACE variant of the library (into
tools/probes/libuvl_wgtrace.so; the product library is untouched), runs one attention launch and prints, from the constant-clock
stamps wave 0 of every workgroup took (entry, loop start, loop end, exit), where the launch's time goes: dispatch ramp, prologue, tile
loop, epilogue, and the hand-over between successive workgroups of one CU slot.
    build:  python tools/attn_wgtrace.py --build        run (GPU box):  python tools/attn_wgtrace.py B H N [cfg]"""
import ctypes as C
import os
import subprocess
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from uvltrack_amd import build as B  # noqa: E402

LIB = os.path.join(ROOT, "tools", "probes", "libuvl_wgtrace.so")


def build():
    objs = []
    for src in B.SOURCES:
        obj = os.path.join(ROOT, "tools", "probes", "wgtrace_" + src.replace(".hip", ".o"))
        subprocess.run(["/opt/rocm/bin/hipcc"] + B.FLAGS + ["-DATTN_WGTRACE", "-c", os.path.join(B.CSRC, src), "-o", obj], check=True)
        objs.append(obj)
    subprocess.run(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-shared", "-fPIC", "-o", LIB] + objs, check=True)
    print("built", LIB)


def main():
    if "--build" in sys.argv:
        return build()
    import torch
    lib = C.CDLL(LIB)
    Bn, H, N = (int(a) for a in sys.argv[1:4])
    cfg = int(sys.argv[4]) if len(sys.argv) > 4 else 10
    Npad = (N + 63) // 64 * 64
    q = (torch.randn(Bn, H, Npad, 64, device="cuda") * 0.18033688).bfloat16()
    k = torch.randn(Bn, H, Npad, 64, device="cuda").bfloat16()
    vt = torch.randn(Bn, H, 64, Npad, device="cuda").bfloat16()
    add = torch.zeros(Bn, Npad, device="cuda")
    o = torch.empty(Bn * N, H * 64, device="cuda", dtype=torch.bfloat16)
    p = lambda t: C.c_void_p(t.data_ptr())
    st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    from uvltrack_amd import _native
    TUNE = _native.UvlTuning(attn_cfg=cfg)
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
    for _ in range(5):
        lib.uvl_attention(p(q), p(k), p(vt), p(add), p(o), Bn, H, N, Npad, 1, TUNE.ref(), st)
    torch.cuda.synchronize()
    ev[0].record()
    lib.uvl_attention(p(q), p(k), p(vt), p(add), p(o), Bn, H, N, Npad, 1, TUNE.ref(), st)
    ev[1].record()
    torch.cuda.synchronize()
    nwg = ((((N + 63) // 64) + 3) // 4) * H * Bn
    nwg8 = 8 * ((nwg + 7) // 8)
    n = min(nwg8, 8192)
    buf = (C.c_ulonglong * (n * 6))()
    assert lib.uvl_debug_attn_wgtrace(buf, n) == 0
    a = np.frombuffer(buf, dtype=np.uint64).reshape(n, 6).astype(np.int64)
    a = a[a[:, 3] > 0]
    t = a[:, :4] * 10.0                                    # ns (100 MHz)
    t0 = t[:, 0].min()
    hw = a[:, 4] & 0xffffffff
    xcc = (a[:, 4] >> 32) & 0xf
    cu = (hw >> 8) & 0xf
    sh = (hw >> 12) & 0x1
    se = (hw >> 13) & 0x7
    slot = xcc * 4096 + se * 512 + sh * 256 + cu * 16      # one CU
    print("B=%d H=%d N=%d cfg %d: %d workgroups traced, launch %.1f us by events, %.1f us first entry -> last exit" %
          (Bn, H, N, cfg, len(a), ev[0].elapsed_time(ev[1]) * 1e3, (t[:, 3].max() - t0) / 1e3))
    q50 = lambda x: "%.2f / %.2f / %.2f" % tuple(np.percentile(x, [10, 50, 90]) / 1e3)
    print("  us (10 / 50 / 90 %%):  entry after first entry %s | prologue %s | tile loop %s | epilogue %s | whole item %s" %
          (q50(t[:, 0] - t0), q50(t[:, 1] - t[:, 0]), q50(t[:, 2] - t[:, 1]), q50(t[:, 3] - t[:, 2]), q50(t[:, 3] - t[:, 0])))
    # per CU: successive workgroups
    gaps, per_cu = [], {}
    for i in range(len(a)):
        per_cu.setdefault(int(slot[i]), []).append((t[i, 0], t[i, 3]))
    busy = []
    for s_, lst in per_cu.items():
        lst.sort()
        ends = []
        for (b_, e_) in lst:
            # a workgroup takes over from the co-resident one that ended latest before its entry
            prev = [x for x in ends if x <= b_]
            if prev:
                gaps.append(b_ - max(prev))
                ends.remove(max(prev))
            ends.append(e_)
        busy.append(sum(e_ - b_ for b_, e_ in lst))
    span = t[:, 3].max() - t0
    print("  %d CUs seen, workgroups per CU %.2f; hand-over gap (exit of a workgroup -> entry of the next on that CU) us 10/50/90 %%: %s" %
          (len(per_cu), len(a) / len(per_cu), q50(np.array(gaps)) if gaps else "-"))
    print("  sum of item times per CU / (2 x span): %.1f %%" % (100.0 * np.mean(busy) / (2 * span)))
    order = np.argsort(t[:, 0])
    print("  first 6 and last 6 workgroups by entry (us): entry, loop start, loop end, exit, xcc/se/cu")
    for i in list(order[:6]) + list(order[-6:]):
        print("   %7.2f %7.2f %7.2f %7.2f   %d/%d/%d" % ((t[i, 0] - t0) / 1e3, (t[i, 1] - t0) / 1e3, (t[i, 2] - t0) / 1e3, (t[i, 3] - t0) / 1e3, xcc[i], se[i], cu[i]))


if __name__ == "__main__":
    main()
