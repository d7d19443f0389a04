"""Phase timeline of attn_w64_kernel (development aid): builds a trace variant of the library (-DATTN_TRACE, into
tools/probes/libuvl_trace.so; the product library is untouched), runs one attention launch and prints the shader-clock ticks that
wave 0 of workgroup 0 spent per phase, summed over its key tiles.
    build (CPU container or GPU box):  python tools/attn_trace.py --build
    run (GPU box):                     python tools/attn_trace.py B H N [cfg]"""
import ctypes as C
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from uvltrack_amd import build as B  # noqa: E402

LIB = os.path.join(ROOT, "tools", "probes", "libuvl_trace.so")


def build():
    objs = []
    for src in B.SOURCES:
        obj = os.path.join(ROOT, "tools", "probes", "trace_" + src.replace(".hip", ".o"))
        subprocess.run(["/opt/rocm/bin/hipcc"] + B.FLAGS + ["-DATTN_TRACE"] + (["-DATTN_TRACE_CAL"] if "--cal" in sys.argv else []) + ["-c", os.path.join(B.CSRC, src), "-o", obj], check=True)
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
    TUNE = _native.UvlTuning(attn_cfg=cfg)      # per-call override (no process-global tuning state)
    for _ in range(3):
        lib.uvl_attention(p(q), p(k), p(vt), p(add), p(o), Bn, H, N, Npad, 1, TUNE.ref(), st)
    torch.cuda.synchronize()
    buf = (C.c_ulonglong * (16 + 8 * 32))()
    assert lib.uvl_debug_attn_trace(buf) == 0
    names = ["between tiles", "DMA wait + barrier", "K reads + phase 2 (QK block 0)", "phase 3 (QK block 1 | softmax 0)",
             "phase 4 (PV block 0 | softmax 1)", "phase 5 (PV block 1 | DMA issue)", "prologue", "epilogue"]
    nt = buf[9]
    total = buf[8]
    print("B=%d H=%d N=%d cfg %d: %d key tiles, %d ticks for the item (wave 0 of workgroup 0)" % (Bn, H, N, cfg, nt, total))
    for i, n in enumerate(names):
        per = buf[i] / nt if i < 6 else buf[i]
        print("  %-40s %9d ticks  %5.1f %%   %s" % (n, buf[i], 100.0 * buf[i] / total, ("%.0f per tile" % per) if i < 6 else ""))
    print("  per tile: [top, wait+barrier, K+phase 2, phase 3, phase 4, phase 5]")
    for t in range(min(int(nt) + 1, 32)):
        print("   tile %2d: %s" % (t, " ".join("%5d" % buf[16 + t * 8 + i] for i in range(6))))


if __name__ == "__main__":
    main()
