"""A/B of the batched GEMM forms on the frames' own shapes, beside hipBLASLt (through torch, a yardstick only): the 128x128 product
loop (cfg 6), its producer-wave form (21), the plain 256x256 tile (11), the phase-pipelined 256-wide tiles (30: 256x256,
31: 128x256; gemm.hip::gemm_pipe_body; 32 / 33: the same with 32x32x16 MFMAs).  Every configuration is bit-checked against torch before it is timed; rounds are
interleaved in one process and the best of them is reported (guide section 5.4 rule 24).
Usage (GPU box): python tools/gemm_pipe_ab.py [--epi bf16|gelu|f32acc|qkv] [--cfgs 6,11,30] [--shapes L8,B32,...]"""
import ctypes as C
import os
import sys

import torch
import torch.nn.functional as F

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from uvltrack_amd import _native  # noqa: E402

lib = _native.load()
p = lambda t: C.c_void_p(t.data_ptr())
st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
LABEL = {4: "64x64 ring", 7: "64x64 2st", 9: "128x64", 10: "64x128", 6: "128x128", 21: "128x128+4prod", 11: "256x256 plain", 30: "pipe 256x256", 31: "pipe 128x256", 32: "pipe256 mi32", 33: "pipe128 mi32", -1: "auto"}


def arg(name, default):
    if name in sys.argv:
        return sys.argv[sys.argv.index(name) + 1]
    return default


def time_once(fn, iters):
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(iters):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / iters * 1e3


SETS = {
    "L8": [("L8 qkv", 6984, 3072, 1024), ("L8 fc1", 6984, 4096, 1024), ("L8 proj", 6984, 1024, 1024), ("L8 fc2", 6984, 1024, 4096)],
    "L8z128": [("L8' qkv", 5448, 3072, 1024), ("L8' fc1", 5448, 4096, 1024), ("L8' proj", 5448, 1024, 1024), ("L8' fc2", 5448, 1024, 4096)],
    "B32": [("B32 qkv", 17696, 2304, 768), ("B32 fc1", 17696, 3072, 768), ("B32 proj", 17696, 768, 768), ("B32 fc2", 17696, 768, 3072)],
    "L32": [("L32 qkv", 21792, 3072, 1024), ("L32 fc1", 21792, 4096, 1024), ("L32 proj", 21792, 1024, 1024), ("L32 fc2", 21792, 1024, 4096)],
    # 1024 tiles of 256x256 = exactly four rounds of 256 CUs: time(K) = 4 x (fixed cost of a tile + K/64 x cost of a K step)
    "KS": [("K%d" % k, 16384, 4096, k) for k in (128, 256, 512, 1024, 2048, 4096, 8192)],
    # one UVLTrack-L sequence (BASELINE configs[3]): 833 visual rows, 873 in the fusion layers
    "L1": [("L1 qkv", 873, 3072, 1024), ("L1 fc1", 873, 4096, 1024), ("L1 qkv833", 833, 3072, 1024), ("L1 fc1 833", 833, 4096, 1024)],
    "B8": [("B8 qkv", 4424, 2304, 768), ("B8 fc1", 4424, 3072, 768), ("B8 proj", 4424, 768, 768), ("B8 fc2", 4424, 768, 3072)],
}


def main():
    epi = arg("--epi", "bf16")
    cfgs = [int(c) for c in arg("--cfgs", "6,21,11,30,31").split(",")]
    shapes = sum((SETS[s] for s in arg("--shapes", "L8,B32,L32").split(",")), [])
    rounds, iters = int(arg("--rounds", "3")), int(arg("--iters", "20"))
    for name, M, N, K in shapes:
        g = torch.Generator(device="cuda").manual_seed(1)
        x = torch.randn(M, K, device="cuda", generator=g).bfloat16()
        w = (torch.randn(N, K, device="cuda", generator=g) / K ** 0.5 + torch.linspace(-0.02, 0.03, N, device="cuda")[:, None]).bfloat16()
        bias = torch.randn(N, device="cuda", generator=g)
        ref = x.float() @ w.float().t() + bias
        flops = 2.0 * M * N * K
        fns, notes = {}, {}
        for cfg in cfgs:
            if cfg in (11, 30, 31, 32) and N % 256:
                continue
            if cfg in (6, 21) and N % 128:
                continue
            tune = _native.UvlTuning(gemm_cfg=cfg) if cfg >= 0 else _native.UvlTuning()
            if epi in ("bf16", "gelu"):
                act = 1 if epi == "gelu" else 0
                y = torch.full((M, N), float("nan"), device="cuda", dtype=torch.bfloat16)
                fn = (lambda y=y, tune=tune, act=act: lib.uvl_linear(p(x), p(w), p(bias), p(y), M, N, K, act, 0, 0, tune.ref(), st))
                rc = fn()
                torch.cuda.synchronize()
                want = F.gelu(ref) if act else ref
                err = (y.float() - want).abs()
                ok = rc == 0 and bool((err <= 1e-2 * want.abs() + 2e-2).all())
            elif epi == "f32acc":
                y0 = torch.randn(M, N, device="cuda", generator=g)
                y = y0.clone()
                fn = (lambda y=y, tune=tune: lib.uvl_linear(p(x), p(w), p(bias), p(y), M, N, K, 0, 1, 1, tune.ref(), st))
                rc = fn()
                torch.cuda.synchronize()
                ok = rc == 0 and float((y - (ref + y0)).abs().max()) < 2e-3
            else:   # qkv scatter epilogue: N = 3 D
                D = N // 3
                Hh, Bq = D // 64, 8
                ntok = M // Bq
                Mq = ntok * Bq
                npad = (ntok + 63) // 64 * 64
                q = torch.zeros(Bq, Hh, npad, 64, device="cuda", dtype=torch.bfloat16)
                k = torch.zeros_like(q)
                vt = torch.zeros(Bq, Hh, 64, npad, device="cuda", dtype=torch.bfloat16)
                fn = (lambda tune=tune, q=q, k=k, vt=vt: lib.uvl_qkv_project(p(x), p(w), p(bias), p(q), p(k), p(vt), Bq, ntok, npad, D, C.c_float(1.0), tune.ref(), st))
                rc = fn()
                torch.cuda.synchronize()
                r = ref[:Mq].reshape(Bq, ntok, 3, Hh, 64).permute(2, 0, 3, 1, 4)
                ok = rc == 0
                for got, want in ((q[:, :, :ntok], r[0]), (k[:, :, :ntok], r[1]), (vt[:, :, :, :ntok].transpose(2, 3), r[2])):
                    ok &= bool(((got.float() - want).abs() <= 1e-2 * want.abs() + 2e-2).all())
            fns[cfg] = fn
            notes[cfg] = "" if ok else " WRONG"
        bb = bias.bfloat16()
        fns["blaslt"] = lambda: F.linear(x, w, bb)
        best = {k: 1e9 for k in fns}
        for k, fn in fns.items():
            for _ in range(3):
                fn()
        torch.cuda.synchronize()
        for _ in range(rounds):
            for k, fn in fns.items():
                best[k] = min(best[k], time_once(fn, iters))
        row = ["%s %6.1f us %5.0f TF%s" % (LABEL.get(k, k), best[k], flops / best[k] / 1e6, notes.get(k, "")) for k in fns]
        print("%-8s M=%5d N=%4d K=%4d %-6s | %s" % (name, M, N, K, epi, " | ".join(row)), flush=True)


if __name__ == "__main__":
    main()
