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


def build_variants(names):
    """timing-only ablations / schedule options of attn_p64_kernel (P64_ABL, P64_OPT of tools/gen/attn_p64_gen.py): libuvl_wgtrace_<name>.so, attention.hip recompiled,
    the other objects shared with the plain trace build"""
    probes = os.path.join(ROOT, "tools", "probes")
    for name in names:
        d = os.path.join(probes, "abl_" + name)
        os.makedirs(d, exist_ok=True)
        parts = name.split("+")
        env = dict(os.environ, P64_ABL=",".join(x for x in parts if "=" not in x),
                   P64_OPT=",".join(x for x in parts if "=" in x))
        subprocess.run([sys.executable, os.path.join(ROOT, "tools", "gen", "attn_p64_gen.py"), "--trace", os.path.join(d, "attn_p64_asm_trace.inc")], check=True, env=env)
        obj = os.path.join(d, "attention.o")
        extra = ["-DATTN_P64_NOFALLBACK"] if ("nofallback" in parts or "mfma16" in parts) else []
        subprocess.run(["/opt/rocm/bin/hipcc"] + B.FLAGS + ["-DATTN_WGTRACE"] + extra + ["-I", d, "-c", os.path.join(B.CSRC, "attention.hip"), "-o", obj], check=True)
        objs = [obj] + [os.path.join(probes, "wgtrace_" + src.replace(".hip", ".o")) for src in B.SOURCES if src != "attention.hip"]
        subprocess.run(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-shared", "-fPIC", "-o", os.path.join(probes, "libuvl_wgtrace_%s.so" % name)] + objs, check=True)
        print("built variant", name)


def build():
    if "--variants" in sys.argv:
        return build_variants(sys.argv[sys.argv.index("--variants") + 1].split(":"))
    subprocess.run([sys.executable, os.path.join(ROOT, "tools", "gen", "attn_p64_gen.py"), "--trace", os.path.join(ROOT, "tools", "probes", "attn_p64_asm_trace.inc")], check=True)
    objs = []
    for src in B.SOURCES:
        obj = os.path.join(ROOT, "tools", "probes", "wgtrace_" + src.replace(".hip", ".o"))
        subprocess.run(["/opt/rocm/bin/hipcc"] + B.FLAGS + ["-DATTN_WGTRACE", "-I", os.path.join(ROOT, "tools", "probes"), "-c", os.path.join(B.CSRC, src), "-o", obj], check=True)
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
    t5 = a[:, 5] * 10.0                                    # cfg 11: stores issued (the exit stamp is behind their completion)
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
    if cfg == 11:
        G = int(os.environ.get("P64_WGS", "512"))
        nt = (N + 63) // 64
        nqb = (nt + 3) // 4
        cnt = nwg8 // 8
        vv = np.nonzero(np.frombuffer(buf, dtype=np.uint64).reshape(n, 6)[:, 3] > 0)[0]
        L = (vv & 7) * cnt + (vv >> 3)
        qbs = L % nqb
        rnd = vv // G
        loop = (t[:, 2] - t[:, 1]) / 1e3
        for r in range(int(rnd.max()) + 1):
            print("   round %d: tile loop us by query block: %s" % (r, "  ".join("qb%d %.1f/%.1f/%.1f (n=%d)" % ((q,) + tuple(np.percentile(loop[(rnd == r) & (qbs == q)], [10, 50, 90])) + (int(((rnd == r) & (qbs == q)).sum()),)) for q in range(nqb) if ((rnd == r) & (qbs == q)).any())))
        first = rnd == 0
        print("   round 0 median loop us by XCD, first / second workgroup of a CU: " + "  ".join("x%d %.1f/%.1f" % (x, np.median(loop[first & (xcc == x) & (vv < G // 2)]), np.median(loop[first & (xcc == x) & (vv >= G // 2)])) for x in range(8)))
        pro = (t[:, 1] - t[:, 0]) / 1e3
        print("   round 0 median prologue us by XCD: " + "  ".join("x%d %.1f/%.1f" % (x, np.median(pro[first & (xcc == x) & (vv < G // 2)]), np.median(pro[first & (xcc == x) & (vv >= G // 2)])) for x in range(8)))
        slow = np.argsort(-loop)[:8]
        print("   slowest loops: " + "  ".join("v%d r%d qb%d xcc%d %.1fus@%.1f" % (vv[i], rnd[i], qbs[i], xcc[i], loop[i], (t[i, 1] - t0) / 1e3) for i in slow))
        print("  cfg 11: epilogue arithmetic + store issue %s | store / surplus-round drain %s" % (q50(t5 - t[:, 2]), q50(t[:, 3] - t5)))
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
