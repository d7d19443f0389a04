"""The GELU of the GEMM epilogues (uvltrack_amd/csrc/common.h::gelu_erf_poly2 / gelu_erf_poly4) is a clamp + degree-6 polynomial, not
the exact erf form of the reference (`nn.GELU`, /root/reference/lib/models/backbones/utils.py:63-69; the BERT intermediate activation,
bert_backbone.py:118-124).  This file pins it:
  * CPU: the coefficients and the clamp point are READ from common.h and restated in numpy float32 (same Horner order, fused
    multiply-adds emulated through float64); |error| against 0.5 x (1 + erf(x / sqrt 2)) on a dense grid, the relative error for x > 0,
    and the exact saturation beyond the clamp (the advisor's round-4 finding: a residual slope of -8.6e-6 x below x = -3.8).
  * GPU: uvl_linear(act = 1) on an identity weight (the pre-activation IS the input) agrees with the numpy polynomial to bf16 rounding.
"""
import ctypes as C
import os
import re

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
F32 = np.float32


def _constants():
    src = open(os.path.join(ROOT, "uvltrack_amd", "csrc", "common.h")).read()
    body = src[src.index("f32x2 gelu_erf_poly2(f32x2 x)"):src.index("gelu_erf_poly4")]
    coef = [float(m) for m in re.findall(r"f32x2\{(-?[0-9.]+e[-+][0-9]+)f,", body)]
    clamp = float(re.search(r"#define GELU_POLY_CLAMP ([0-9.]+)f", src).group(1))
    assert len(coef) == 7, coef
    # the two-pair form must carry the same numbers
    body4 = src[src.index("f32x4 gelu_erf_poly4(f32x4 x)"):src.index("sigmoidf_")]
    coef4 = [float(m) for m in re.findall(r"f32x2\{(-?[0-9.]+e[-+][0-9]+)f,", body4)]
    assert sorted(coef4) == sorted(coef + coef)                  # each coefficient once per chain
    return [F32(c) for c in coef], F32(clamp)


def _fma(a, b, c):
    return (a.astype(np.float64) * np.float64(b) + np.float64(c)).astype(F32) if np.ndim(b) == 0 else \
        (a.astype(np.float64) * b.astype(np.float64) + np.float64(c)).astype(F32)


def gelu_poly(x):
    """numpy float32 restatement of gelu_erf_poly2, operation for operation"""
    coef, clamp = _constants()
    x = np.asarray(x, F32)
    xc = np.clip(x, -clamp, clamp).astype(F32)
    t = (xc * xc).astype(F32)
    q = _fma(t, coef[0], coef[1])
    for c in coef[2:]:
        q = _fma(q, t, c)
    e = (xc * q).astype(F32)
    hx = (x * F32(0.5)).astype(F32)
    return (hx.astype(np.float64) * e.astype(np.float64) + hx.astype(np.float64)).astype(F32)


def gelu_exact(x):
    from scipy.special import erf
    x = np.asarray(x, np.float64)
    return 0.5 * x * (1.0 + erf(x / np.sqrt(2.0)))


def test_polynomial_gelu_error_bounds():
    x = np.linspace(-12.0, 12.0, 960001).astype(F32)
    err = np.abs(gelu_poly(x).astype(np.float64) - gelu_exact(x))
    assert err.max() <= 2.8e-4, (err.max(), x[err.argmax()])
    inside = np.abs(x) <= 3.8                                    # the fitted range: the round-4 figure
    assert err[inside].max() <= 2.5e-4
    pos = x > 0.05
    rel = err[pos] / gelu_exact(x[pos])
    assert rel.max() <= 1.5e-4, rel.max()


def test_polynomial_gelu_saturates_exactly():
    """beyond the clamp erf is exactly +-1 in float32: gelu(x) = x for large x and exactly 0 for very negative x -- no residual slope"""
    _, clamp = _constants()
    big = np.array([clamp, 5.0, 30.0, 100.0, 1e4], F32)
    assert np.array_equal(gelu_poly(big), big)
    assert np.array_equal(gelu_poly(-big), np.zeros_like(big))
    x = np.linspace(-1e4, -4.0, 100001).astype(F32)
    assert np.abs(gelu_poly(x).astype(np.float64) - gelu_exact(x)).max() <= 2.8e-4      # the true value there: -1.3e-4 at -4, -> 0


def _bf16_round(a):
    """float32 -> nearest-even bfloat16 -> float32"""
    u = np.asarray(a, F32).view(np.uint32).astype(np.uint64)
    u = (u + 0x7FFF + ((u >> 16) & 1)) & 0xFFFF0000
    return u.astype(np.uint32).view(F32)


@pytest.mark.gpu
@pytest.mark.parametrize("M", [64, 2304])        # the 64 x 64-tile kernel and a many-row launch
def test_hip_gelu_epilogue_is_that_polynomial(M):
    import torch
    from uvltrack_amd import _native
    lib = _native.load()
    K = N = 64
    g = torch.Generator().manual_seed(11)
    # pre-activations that are bf16 values: with an identity weight and a zero bias the GEMM reproduces them exactly
    x = torch.cat([torch.randn(M * K // 2, generator=g) * 2.0, torch.linspace(-9.0, 9.0, M * K - M * K // 2)]).reshape(M, K).bfloat16()
    w = torch.eye(N, K).bfloat16().cuda()
    b = torch.zeros(N).cuda()
    xd = x.cuda()
    y = torch.full((M, N), float("nan"), dtype=torch.bfloat16, device="cuda")
    p = lambda t: C.c_void_p(t.data_ptr())
    rc = lib.uvl_linear(p(xd), p(w), p(b), p(y), M, N, K, 1, 0, 0, None, C.c_void_p(torch.cuda.current_stream().cuda_stream))
    assert rc == 0, lib.uvl_last_error().decode()
    torch.cuda.synchronize()
    pre = x.float().numpy()
    want = gelu_poly(pre)
    got = y.float().cpu().numpy()
    # one bf16 step of slack: the device's fused multiply-adds round once, the emulation through float64 twice (rare halfway cases)
    ulp = np.maximum(np.abs(want), 2.0 ** -126) * 2.0 ** -7
    assert np.all(np.abs(got - _bf16_round(want)) <= ulp), float(np.abs(got - _bf16_round(want)).max())
    # and it is NOT silently the exact erf form where the two differ by more than a bf16 step (small negative outputs near the clamp)
    exact = gelu_exact(pre).astype(F32)
    differs = np.abs(_bf16_round(exact) - _bf16_round(want)) > 0
    assert differs.any() and np.array_equal(got[differs], _bf16_round(want)[differs]) or not differs.any()
