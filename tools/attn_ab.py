"""A/B of generator options of attn_p64_kernel (tools/gen/attn_p64_gen.py: P64_OPT) on ONE box, rounds interleaved over the variants.
    build (CPU):  python tools/attn_ab.py --build rsum=0 rsum=1 [...]      -> tools/probes/libuvl_ab_<name>.so (attention.hip recompiled per variant,
                                                                            the other objects are the product's uvltrack_amd/build/*.o)
    run (GPU):    python tools/attn_ab.py --run rsum=0 rsum=1 [--shapes 8x16x873,32x16x873] [--check]
`--check` compares every variant's output with the first one's (max abs difference).  The committed attn_p64_asm.inc is restored after a build."""
import ctypes as C
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from uvltrack_amd import build as B  # noqa: E402

PROBES = os.path.join(ROOT, "tools", "probes")
GEN = os.path.join(ROOT, "tools", "gen", "attn_p64_gen.py")


def lib_of(name):
    return os.path.join(PROBES, "libuvl_ab_%s.so" % name.replace("=", "").replace(",", "_").replace(":", ""))


def build(names):
    B.build(force=False, verbose=False)
    try:
        for name in names:
            # "rsum=2", "rsum=1,abl:norsum" ...: key=value -> P64_OPT, abl:x -> P64_ABL (timing-only: results wrong, the exact pass is compiled out)
            parts = [] if name == "default" else name.split(",")
            abl = [x[4:] for x in parts if x.startswith("abl:")]
            env = dict(os.environ, P64_OPT=",".join(x for x in parts if not x.startswith("abl:")), P64_ABL=",".join(abl))
            subprocess.run([sys.executable, GEN], check=True, env=env, capture_output=True)
            obj = os.path.join(PROBES, "ab_%s_attention.o" % name.replace("=", "").replace(",", "_").replace(":", ""))
            subprocess.run(["/opt/rocm/bin/hipcc"] + B.FLAGS + (["-DATTN_P64_NOFALLBACK"] if abl else []) + ["-c", os.path.join(B.CSRC, "attention.hip"), "-o", obj], check=True, capture_output=True)
            objs = [obj] + [os.path.join(B.HERE, "build", s.replace(".hip", ".o")) for s in B.SOURCES if s != "attention.hip"]
            subprocess.run(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-shared", "-fPIC", "-o", lib_of(name)] + objs, check=True)
            print("built", lib_of(name))
    finally:
        subprocess.run([sys.executable, GEN], check=True, env=dict(os.environ, P64_OPT="", P64_ABL=""), capture_output=True)


def run(names, shapes, check):
    import torch
    from uvltrack_amd import _native
    libs = {n: C.CDLL(lib_of(n)) for n in names}
    tune = _native.UvlTuning(attn_cfg=11)
    p = lambda t: C.c_void_p(t.data_ptr())
    st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    for Bn, H, N in shapes:
        Npad = (N + 63) // 64 * 64
        g = torch.Generator(device="cuda").manual_seed(1)
        q = (torch.randn(Bn, H, Npad, 64, device="cuda", generator=g) * 0.18033688).bfloat16()
        k = torch.randn(Bn, H, Npad, 64, device="cuda", generator=g).bfloat16()
        vt = torch.randn(Bn, H, 64, Npad, device="cuda", generator=g).bfloat16()
        add = torch.zeros(Bn, Npad, device="cuda")
        outs = {n: torch.zeros(Bn * N, H * 64, device="cuda", dtype=torch.bfloat16) for n in names}
        fn = {n: (lambda n=n: libs[n].uvl_attention(p(q), p(k), p(vt), p(add), p(outs[n]), Bn, H, N, Npad, 1, tune.ref(), st)) for n in names}
        for n in names:
            for _ in range(10):
                assert fn[n]() == 0
        torch.cuda.synchronize()
        best = {n: 1e30 for n in names}
        for _rep in range(5):
            for n in names:
                a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                fn[n]()
                a.record()
                for _ in range(30):
                    fn[n]()
                b.record()
                torch.cuda.synchronize()
                best[n] = min(best[n], a.elapsed_time(b) / 30 * 1e3)
        fl = 4.0 * N * N * 64 * H * Bn
        line = "  ".join("%s %.2f us %.0f TF" % (n, best[n], fl / best[n] / 1e6) for n in names)
        if check:
            ref = outs[names[0]].float()
            line += "   max|diff| vs %s: " % names[0] + " ".join("%.3g" % float((outs[n].float() - ref).abs().max()) for n in names[1:])
            # and against fp32 torch on one head
            s = (q[0, 0, :N].float() @ k[0, 0, :N].float().t()) * 0.6931471805599453
            o = torch.softmax(s, -1) @ vt[0, 0, :, :N].float().t()
            line += "   vs torch (b0,h0): " + " ".join("%.3g" % float((outs[n][:N, :64].float() - o).abs().max()) for n in names)
        print("%d x %d x %d:  %s" % (Bn, H, N, line), flush=True)


if __name__ == "__main__":
    args = sys.argv[1:]
    shapes = [(8, 16, 873), (32, 16, 873), (32, 16, 681), (8, 12, 553), (64, 12, 553)]
    if "--shapes" in args:
        i = args.index("--shapes")
        shapes = [tuple(int(x) for x in sh.split("x")) for sh in args[i + 1].split(",")]
        del args[i:i + 2]
    check = "--check" in args
    args = [a for a in args if a != "--check"]
    if args and args[0] == "--build":
        build(args[1:])
    elif args and args[0] == "--run":
        run(args[1:], shapes, check)
    else:
        print(__doc__)
