"""Where a one-sequence GEMM launch spends its time (development aid): a -DGLDS_TRACE variant of gemm.hip (tools/probes/libuvl_gldstrace.so; the product library is untouched) stamps the
100 MHz clock in wave 0 of every workgroup at the seams of gemm_glds_body: entry | tile decoded, addresses ready | prologue DMA issued | tile 0 landed + barrier | tile 1 | K loop done | epilogue done.
    build (CPU): python tools/glds_trace.py --build        run (GPU box): python tools/glds_trace.py M N K [f32_slabs splitk]"""
import ctypes as C
import os
import subprocess
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from uvltrack_amd import build as B  # noqa: E402

ABLS = [0, 100] if os.environ.get('GLDS_ORD') else [0, 1, 2, 3, 4, 8, 12, 15]        # 100 / 200: -DGLDS_ORDER=1 / 2        # (0 first: the product loop)
LIBF = os.path.join(ROOT, "tools", "probes", "libuvl_gldstrace%d.so")


def build():
    B.build(force=False, verbose=False)
    procs = []
    for abl in ABLS:
        obj = os.path.join(ROOT, "tools", "probes", "gldstrace_gemm%d.o" % abl)
        procs.append((abl, obj, subprocess.Popen(["/opt/rocm/bin/hipcc"] + B.FLAGS + ["-DGLDS_TRACE", "-DGLDS_ABL=%d" % (abl % 100), "-DGLDS_ORDER=%d" % (abl // 100), "-c", os.path.join(B.CSRC, "gemm.hip"), "-o", obj])))
        if len(procs) % 4 == 0:
            for _, _, pr in procs[-4:]:
                assert pr.wait() == 0
    for abl, obj, pr in procs:
        assert pr.wait() == 0
        objs = [obj] + [os.path.join(B.HERE, "build", s.replace(".hip", ".o")) for s in B.SOURCES if s != "gemm.hip"]
        subprocess.run(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-shared", "-fPIC", "-o", LIBF % abl] + objs, check=True)
        print("built", LIBF % abl)


def frame(model):
    """The same stamps INSIDE the frame: every workgroup of every gemm_glds launch adds its phase times to per-epilogue accumulators (g_glds_acc) over 50 replays of one
    sequence's frame (order 1 library = the product loop); the 0-order library beside it."""
    import torch
    from uvltrack_amd import _native
    import bench
    from uvltrack_amd import weightgen as wg
    names = ["decode", "issue", "tile0", "tile1", "loop", "epilogue"]
    for abl in (100, 0):
        _native.LIB_PATH = LIBF % abl
        _native._lib = None
        from uvltrack_amd.engine import HipEngine
        dev = torch.device("cuda:0")
        spec = bench.build_spec(model, 128, 256)
        eng = HipEngine(spec, dev, max_batch=1)
        lib = eng.lib
        assert os.path.samefile(lib._name, LIBF % abl), (lib._name, LIBF % abl)
        lib.uvl_debug_glds_acc.argtypes = [C.c_void_p, C.c_int]
        eng.load_state_dict(wg.make_state_dict(spec, 0, include_unused=False))
        inp = wg.make_inputs(spec, batch=1, seed=1, flags=[2])
        t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)
        eng.capture(t(inp["template"]), t(inp["search"]), t(inp["ids"]), t(inp["mask"]), t(inp["prompt"]), t(inp["flag"]))
        for _ in range(10):
            eng.replay()
        torch.cuda.synchronize()
        buf = (C.c_ulonglong * 64)()
        assert lib.uvl_debug_glds_acc(buf, 1) == 0
        n = 50
        for _ in range(n):
            eng.replay()
        torch.cuda.synchronize()
        assert lib.uvl_debug_glds_acc(buf, 1) == 0
        a = np.frombuffer(buf, dtype=np.uint64).astype(np.float64).reshape(8, 8)
        print("--- loop order %d, UVLTrack-%s x 1, per workgroup of the frame's 64 x 64 GEMM launches (us; 100 MHz ticks summed over every workgroup, %d frames)" % (abl // 100, model, n))
        for row, lab in ((0, "bf16/GELU epilogue"), (1, "f32 slabs"), (2, "QKV scatter"), (4, "rider bf16"), (5, "rider f32"), (6, "rider QKV")):
            if a[row, 6] == 0:
                continue
            per = a[row, :6] / a[row, 6] / 100.0
            print("   %-20s %6.0f workgroups/frame, %4.1f K tiles each: " % (lab, a[row, 6] / n, a[row, 7] / a[row, 6]) + "  ".join("%s %.2f" % (nm, v) for nm, v in zip(names, per)) + "   | total %.2f" % per.sum())
        eng.close()
        del eng


def main():
    if "--build" in sys.argv:
        return build()
    if "--frame" in sys.argv:
        return frame("L" if "L" in sys.argv else "B")
    import torch
    M, N, K = (int(a) for a in sys.argv[1:4])
    sk = int(sys.argv[4]) if len(sys.argv) > 4 and sys.argv[4].isdigit() else 0
    sys.path.insert(0, ROOT)
    from uvltrack_amd import _native
    if "--orders" in sys.argv:                    # loop orders of the 64 x 64 tile (-DGLDS_ORDER), product loop otherwise
        for rep in range(2):
            for abl in ABLS:
                print("--- order %d" % (abl // 100))
                one(torch, C.CDLL(LIBF % abl), M, N, K, sk, _native.UvlTuning(gemm_cfg=4))
        return
    for abl in ABLS:
        print("--- ablation %d (1 no MFMA, 2 no fragment reads, 4 no LDS-DMA in the loop, 8 no barrier)" % abl)
        one(torch, C.CDLL(LIBF % abl), M, N, K, sk, None)


def one(torch, lib, M, N, K, sk, tune):
    tr = tune.ref() if tune is not None else None
    p = lambda t: C.c_void_p(t.data_ptr())
    st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    names = ["decode", "issue", "tile0", "tile1", "loop", "epilogue"]
    for cold in (0,) if os.environ.get('GLDS_ORD') else (0, 1):
        x = torch.randn(M, K, device="cuda").bfloat16()
        ws = [(torch.randn(N, K, device="cuda") / K ** 0.5).bfloat16() for _ in range(40 if cold else 1)]      # cold: a weight nobody has read (rotating through > 256 MB)
        bias = torch.randn(N, device="cuda")
        y = torch.empty(M, N, device="cuda", dtype=torch.bfloat16)
        slabs = torch.empty(max(sk, 1), M, N, device="cuda")
        rows = []
        for it in range(12):
            w = ws[it % len(ws)]
            if sk:
                rc = lib.uvl_linear_splitk(p(x), p(w), p(bias), p(slabs), M, N, K, sk, tr, st)
            else:
                rc = lib.uvl_linear(p(x), p(w), p(bias), p(y), M, N, K, 0, 0, 0, tr, st)
            assert rc == 0
            torch.cuda.synchronize()
            n = 2048
            buf = (C.c_ulonglong * (n * 8))()
            assert lib.uvl_debug_glds_trace(buf, n) == 0
            a = np.frombuffer(buf, dtype=np.uint64).astype(np.int64).reshape(n, 8)
            nwg = int((a[:, 6] > 0).sum()) if it == 0 else nwg
            if it >= 4:
                rows.append(a[:nwg].copy())
        a = np.stack(rows).astype(np.float64)             # [its, wg, 8]   (the 100 MHz counters of different XCDs are offset against each other: compare within an XCD only)
        span = np.mean([(a[:, x::8, 6].max(axis=1) - a[:, x::8, 0].min(axis=1)).mean() for x in range(8)]) / 100.0
        skew = np.mean([(a[:, x::8, 0].max(axis=1) - a[:, x::8, 0].min(axis=1)).mean() for x in range(8)]) / 100.0
        seg = [(a[:, :, k + 1] - a[:, :, k]).mean() / 100.0 for k in range(6)]
        ev = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
        ev[0].record()
        for it in range(200):
            w = ws[it % len(ws)]
            if sk:
                lib.uvl_linear_splitk(p(x), p(w), p(bias), p(slabs), M, N, K, sk, tr, st)
            else:
                lib.uvl_linear(p(x), p(w), p(bias), p(y), M, N, K, 0, 0, 0, tr, st)
        ev[1].record()
        torch.cuda.synchronize()
        print("M=%d N=%d K=%d splitk=%d %s: %d workgroups; back-to-back launches %.2f us each; within an XCD: first entry -> last exit %.2f us, entry skew %.2f us" % (
            M, N, K, sk, "rotating weights" if cold else "one weight", nwg, ev[0].elapsed_time(ev[1]) * 5.0, span, skew))
        print("   mean per workgroup (us): " + "  ".join("%s %.2f" % (nm, v) for nm, v in zip(names, seg)) + "   | total %.2f" % sum(seg), flush=True)


if __name__ == "__main__":
    main()
