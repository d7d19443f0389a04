"""Phase timeline of gemm_pipe_body (development aid): builds a trace variant of the library (-DGEMM_TRACE, into
tools/probes/libuvl_gtrace.so; the product library is untouched), runs one GEMM and prints the shader-clock ticks that wave 0 (wave
group 0) and wave 4 (group 1) of workgroup 0 spent per phase: [barrier-2 wait -> start | fragment reads + counted wait | barrier 1 +
lgkmcnt | MFMAs (+ LDS-DMA issue)].
    build (CPU container or GPU box):  python tools/gemm_trace.py --build
    run (GPU box):                     python tools/gemm_trace.py M N K cfg"""
import ctypes as C
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from uvltrack_amd import build as B  # noqa: E402

LIB = os.path.join(ROOT, "tools", "probes", "libuvl_gtrace.so")


def build():
    objs = []
    for src in B.SOURCES:
        obj = os.path.join(ROOT, "tools", "probes", "gtrace_" + src.replace(".hip", ".o"))
        subprocess.run(["/opt/rocm/bin/hipcc"] + B.FLAGS + ["-DGEMM_TRACE", "-c", os.path.join(B.CSRC, src), "-o", obj], check=True)
        objs.append(obj)
    subprocess.run(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-shared", "-fPIC", "-o", LIB] + objs, check=True)
    print("built", LIB)


def main():
    if "--build" in sys.argv:
        return build()
    import torch
    from uvltrack_amd import _native
    lib = C.CDLL(LIB)
    M, N, K, cfg = (int(a) for a in sys.argv[1:5])
    x = torch.randn(M, K, device="cuda").bfloat16()
    w = (torch.randn(N, K, device="cuda") / K ** 0.5).bfloat16()
    bias = torch.randn(N, device="cuda")
    y = torch.empty(M, N, device="cuda", dtype=torch.bfloat16)
    p = lambda t: C.c_void_p(t.data_ptr())
    st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    tune = _native.UvlTuning(gemm_cfg=cfg)
    for _ in range(3):
        rc = lib.uvl_linear(p(x), p(w), p(bias), p(y), M, N, K, 0, 0, 0, C.byref(tune), st)
        assert rc == 0
    torch.cuda.synchronize()
    ref = x.float() @ w.float().t() + bias
    err = (y.float() - ref).abs()
    print("check:", "ok" if bool((err <= 1e-2 * ref.abs() + 2e-2).all()) else "WRONG (max %.3g)" % float(err.max()))
    n_u32 = 2 * 64 * 4 * 4 + 16
    buf = (C.c_uint * n_u32)()
    assert lib.uvl_debug_gemm_trace(buf) == 0
    if "--raw" in sys.argv:          # absolute stamps of both wave groups side by side (ticks since group 0's first stamp)
        t00 = buf[0]
        for i in range(8, 20):
            a = [(buf[i * 4 + j] - t00) & 0xffffffff for j in range(4)]
            b = [(buf[64 * 4 * 4 + i * 4 + j] - t00) & 0xffffffff for j in range(4)]
            print("   phase %3d  group 0: start %6d wait-done %6d mfma-start %6d mfma-end %6d | group 1: %6d %6d %6d %6d" % ((i,) + tuple(a) + tuple(b)))
    for g in range(2):
        n = buf[2 * 64 * 4 * 4 + g]
        base = g * 64 * 4 * 4
        print("wave group %d: %d phases (4 per K tile); ticks per phase: [bar2->start, reads+wait, bar1+lgkm, MFMAs]  total" % (g, n))
        tot = [0, 0, 0, 0]
        prev3 = None
        for i in range(n):
            t0, t1, t2, t3 = (buf[base + i * 4 + j] for j in range(4))
            d = [((t0 - prev3) & 0xffffffff) if prev3 is not None else 0, (t1 - t0) & 0xffffffff, (t2 - t1) & 0xffffffff, (t3 - t2) & 0xffffffff]
            prev3 = t3
            if i >= 4 and i < n - 8:
                tot = [a + b for a, b in zip(tot, d)]
            if i < 24 or i >= n - 8:
                print("   phase %3d (tile %2d q%d): %5d %5d %5d %5d   %5d" % (i, i // 4, i % 4, d[0], d[1], d[2], d[3], sum(d)))
        cnt = max(1, n - 12)
        print("   steady-state mean per phase:   %5.0f %5.0f %5.0f %5.0f   %5.0f   (per K tile %.0f ticks)" % (
            tot[0] / cnt, tot[1] / cnt, tot[2] / cnt, tot[3] / cnt, sum(tot) / cnt, 4 * sum(tot) / cnt))


if __name__ == "__main__":
    main()
