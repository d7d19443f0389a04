"""GPU parity of the LayerNorm-free kernel forms (round 6) against plain PyTorch fp32 references, through the C ABI:
the finishing residual GEMM (uvl_linear_fin: x (+)= a W^T + b completed in the launch, bf16 rows + partial statistics, optional post-LayerNorm
residual), the LayerNorm fold of a weight (uvl_fold_ln_linear) and the consumer GEMMs that read un-normalised bf16 rows (uvl_linear_lnf,
uvl_qkv_project_lnf).  Reference ops: block.py:29-32 (x + attn(norm1(x)), x + mlp(norm2(x))), bert_backbone.py:335-339,376-380 (post-LN)."""
import ctypes as C
import math

import pytest
import torch

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def lib():
    from uvltrack_amd import _native
    assert torch.cuda.is_available(), "GPU tests need a HIP device"
    return _native.load()


def _stream():
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


def _p(t):
    return C.c_void_p(t.data_ptr()) if t is not None else None


def _rand(shape, seed, scale=1.0):
    g = torch.Generator(device="cpu").manual_seed(seed)
    return (torch.randn(shape, generator=g) * scale).cuda()


def _chk(rc, lib):
    assert rc == 0, lib.uvl_last_error().decode()


def _partials(rows):
    """[M, D] f32 -> [D/64, M, 2, 2]: (sum, sum of squares) per 32 columns, planes of column-block pairs (fold.h::st_off)."""
    M, D = rows.shape
    r = rows.float().reshape(M, D // 64, 2, 32)
    return torch.stack([r.sum(-1), (r * r).sum(-1)], dim=-1).permute(1, 0, 2, 3).contiguous()


def _rows_with_mean_and_outliers(M, D, seed):
    """Residual-stream-like rows: non-zero mean, a few outlier channels (what a LayerNorm fold must survive)."""
    x = _rand((M, D), seed, 1.5) + 0.7
    x[:, 5] *= 20.0
    x[:, D // 2 + 3] -= 15.0
    return x


def _fin_tune(w):
    from uvltrack_amd import _native
    return _native.UvlTuning(fin_w=w).ref() if w is not None else None


@pytest.mark.parametrize("fw", [None, 0, 1], ids=["auto", "w64", "w32"])
@pytest.mark.parametrize("M,N,K,acc", [(553, 768, 768, 1), (553, 768, 3072, 1), (513, 768, 768, 0), (873, 1024, 4096, 1), (873, 1024, 1024, 1),
                                       (40, 768, 3072, 1), (29, 128, 128, 1), (21, 128, 512, 1), (1106, 768, 3072, 1), (1, 64, 128, 0), (70, 192, 256, 1)])
def test_linear_fin(lib, M, N, K, acc, fw):
    """x (+)= a W^T + b finished in the launch; bf16(x) and the rows' per-32-column partials come with it.  Both tile forms (64 x 64 on two wave
    groups, 64 x 32 on four -- the latter needs K % 256 == 0 and falls back otherwise)."""
    tune = _fin_tune(fw)
    a = _rand((M, K), 1).bfloat16()
    w = (_rand((N, K), 2, 1.0 / math.sqrt(K)) + torch.linspace(-0.02, 0.03, N).cuda()[:, None]).bfloat16()
    b = _rand((N,), 3, 0.5)
    x0 = _rand((M, N), 4) if acc else torch.full((M, N), float("nan"), device="cuda")
    x = x0.clone()
    xn = torch.full((M, N), float("nan"), dtype=torch.bfloat16, device="cuda")
    st = torch.full((N // 64, M, 2, 2), float("nan"), device="cuda")
    _chk(lib.uvl_linear_fin(_p(a), _p(w), _p(b), _p(x), _p(xn), _p(st), M, N, K, acc, None, None, None, C.c_float(0.0), None, tune, _stream()), lib)
    torch.cuda.synchronize()
    ref = a.float() @ w.float().t() + b + (x0 if acc else 0.0)
    assert (x - ref).abs().max().item() < 2e-3
    assert torch.equal(xn, x.bfloat16()), "bf16 copy is not the rounding of the stored f32 rows"
    want = _partials(x)
    assert torch.allclose(st, want, rtol=1e-5, atol=1e-4), (st - want).abs().max().item()
    # bit-reproducible: no atomics, fixed summation order
    x2 = x0.clone()
    _chk(lib.uvl_linear_fin(_p(a), _p(w), _p(b), _p(x2), _p(xn), _p(st), M, N, K, acc, None, None, None, C.c_float(0.0), None, tune, _stream()), lib)
    torch.cuda.synchronize()
    assert torch.equal(x, x2)


@pytest.mark.parametrize("fw", [0, 1], ids=["w64", "w32"])
@pytest.mark.parametrize("M,N,K", [(40, 768, 768), (40, 768, 3072), (8, 128, 512), (120, 1024, 4096)])
def test_linear_fin_post_layernorm_residual(lib, M, N, K, fw):
    """BERT's post-LN residual (bert_backbone.py:335-339): the stored rows are pre-norm u, the residual added is LayerNorm(u), its statistics from u's partials."""
    a = _rand((M, K), 11).bfloat16()
    w = _rand((N, K), 12, 1.0 / math.sqrt(K)).bfloat16()
    b = _rand((N,), 13, 0.5)
    u = _rows_with_mean_and_outliers(M, N, 14)
    g = _rand((N,), 15) * 0.2 + 1.0
    be = _rand((N,), 16) * 0.1
    st_u = _partials(u)
    x = u.clone()
    xn = torch.empty((M, N), dtype=torch.bfloat16, device="cuda")
    st = torch.empty((N // 64, M, 2, 2), device="cuda")
    copy = torch.full((M, N), float("nan"), device="cuda")
    _chk(lib.uvl_linear_fin(_p(a), _p(w), _p(b), _p(x), _p(xn), _p(st), M, N, K, 1, _p(st_u), _p(g), _p(be), C.c_float(1e-12), _p(copy), _fin_tune(fw), _stream()), lib)
    torch.cuda.synchronize()
    ln = torch.nn.functional.layer_norm(u, (N,), g, be, 1e-12)
    # f32 statistics from the partials (one-pass variance): a few 1e-5 on normalised values up to ~20
    assert (copy - ln).abs().max().item() < 2e-4
    ref = a.float() @ w.float().t() + b + ln
    assert (x - ref).abs().max().item() < 2e-3
    assert torch.equal(xn, x.bfloat16())


@pytest.mark.parametrize("form", [None, 0, 1, 2], ids=["auto", "64x64", "64x128/2", "64x128/3"])
@pytest.mark.parametrize("M,N,K,act", [(553, 3072, 768, 1), (553, 2304, 768, 0), (40, 3072, 768, 1), (873, 4096, 1024, 1), (29, 512, 128, 1), (1106, 3072, 768, 1), (65, 64, 256, 0)])
def test_linear_lnf_matches_layernorm_then_linear(lib, M, N, K, act, form):
    """norm -> Linear (-> GELU) as ONE GEMM on the un-normalised bf16 rows, against LayerNorm(fp32) -> Linear in fp32, and no worse than ~1.5x the
    LayerNorm-kernel path (uvl_layernorm -> uvl_linear) on rows with a non-zero mean and 20x outlier channels."""
    x = _rows_with_mean_and_outliers(M, K, 21)
    g = _rand((K,), 22) * 0.2 + 1.0
    be = _rand((K,), 23) * 0.1
    w = _rand((N, K), 24, 1.0 / math.sqrt(K)) + torch.linspace(-0.02, 0.03, N).cuda()[:, None]
    b = _rand((N,), 25, 0.5)
    eps = 1e-6
    ref = torch.nn.functional.layer_norm(x, (K,), g, be, eps) @ w.t() + b
    if act:
        ref = torch.nn.functional.gelu(ref)
    # the fold
    wf = torch.empty((N, K), dtype=torch.bfloat16, device="cuda")
    bf = torch.empty((N,), device="cuda")
    cs = torch.empty((N,), device="cuda")
    _chk(lib.uvl_fold_ln_linear(_p(w), _p(b), _p(g), _p(be), _p(wf), _p(bf), _p(cs), N, K, _stream()), lib)
    torch.cuda.synchronize()
    assert torch.equal(wf, (w * g).bfloat16())
    assert torch.allclose(cs, wf.float().sum(-1), rtol=1e-5, atol=1e-4)
    assert torch.allclose(bf, b + w @ be, rtol=1e-5, atol=1e-5)
    xb = x.bfloat16()
    st = _partials(x)
    y = torch.full((M, N), float("nan"), dtype=torch.bfloat16, device="cuda")
    from uvltrack_amd import _native
    tune = _native.UvlTuning(lnf_w=form).ref() if form is not None else None      # tile form of the folded GEMM (64 x 128 needs N % 128 == 0, else 64 x 64)
    _chk(lib.uvl_linear_lnf(_p(xb), _p(st), _p(wf), _p(bf), _p(cs), C.c_float(eps), _p(y), M, N, K, act, tune, _stream()), lib)
    # the LayerNorm-kernel path on the same operands
    xn = torch.empty((M, K), dtype=torch.bfloat16, device="cuda")
    _chk(lib.uvl_layernorm(_p(x), _p(g), _p(be), C.c_float(eps), _p(xn), None, M, K, _stream()), lib)
    y2 = torch.empty((M, N), dtype=torch.bfloat16, device="cuda")
    wb = w.bfloat16()
    _chk(lib.uvl_linear(_p(xn), _p(wb), _p(b), _p(y2), M, N, K, act, 0, 0, None, _stream()), lib)
    torch.cuda.synchronize()
    assert bool(torch.isfinite(y.float()).all())
    rms_fold = (y.float() - ref).pow(2).mean().sqrt().item()
    rms_ln = (y2.float() - ref).pow(2).mean().sqrt().item()
    print("rms error: fold %.3e   LayerNorm kernel + GEMM %.3e" % (rms_fold, rms_ln))
    assert rms_fold <= 1.5 * rms_ln + 1e-3, (rms_fold, rms_ln)
    err = (y.float() - ref).abs()
    assert bool((err <= 2e-2 * ref.abs() + 8e-2).all()), "max err %g" % float(err.max())


@pytest.mark.parametrize("form", [None, 0, 1, 2], ids=["auto", "64x64", "64x128/2", "64x128/3"])
@pytest.mark.parametrize("B,H,N", [(1, 12, 553), (1, 16, 873), (1, 2, 29), (2, 12, 553)])
def test_qkv_project_lnf(lib, B, H, N, form):
    D = H * 64
    Npad = (N + 63) // 64 * 64
    x = _rows_with_mean_and_outliers(B * N, D, 31)
    g = _rand((D,), 32) * 0.2 + 1.0
    be = _rand((D,), 33) * 0.1
    w = _rand((3 * D, D), 34, 1.0 / math.sqrt(D))
    b = _rand((3 * D,), 35, 0.2)
    qs = 0.18033688011112042
    wf = torch.empty((3 * D, D), dtype=torch.bfloat16, device="cuda")
    bf = torch.empty((3 * D,), device="cuda")
    cs = torch.empty((3 * D,), device="cuda")
    _chk(lib.uvl_fold_ln_linear(_p(w), _p(b), _p(g), _p(be), _p(wf), _p(bf), _p(cs), 3 * D, D, _stream()), lib)
    xb = x.bfloat16()
    st = _partials(x)
    q = torch.full((B, H, Npad, 64), float("nan"), dtype=torch.bfloat16, device="cuda")
    k = torch.full_like(q, float("nan"))
    vt = torch.full((B, H, 64, Npad), float("nan"), dtype=torch.bfloat16, device="cuda")
    from uvltrack_amd import _native
    tune = _native.UvlTuning(lnf_w=form).ref() if form is not None else None
    _chk(lib.uvl_qkv_project_lnf(_p(xb), _p(st), _p(wf), _p(bf), _p(cs), C.c_float(1e-6), _p(q), _p(k), _p(vt), B, N, Npad, D, C.c_float(qs), tune, _stream()), lib)
    torch.cuda.synchronize()
    qkv = (torch.nn.functional.layer_norm(x, (D,), g, be, 1e-6) @ w.t() + b).reshape(B, N, 3, H, 64).permute(2, 0, 3, 1, 4)
    for got, ref in ((q[:, :, :N], qkv[0] * qs), (k[:, :, :N], qkv[1]), (vt[:, :, :, :N].transpose(2, 3), qkv[2])):
        err = (got.float() - ref).abs()
        assert bool((err <= 2e-2 * ref.abs() + 8e-2).all()), "qkv scatter max err %g" % float(err.max())
