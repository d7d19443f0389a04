"""GPU parity of the individual HIP kernels against plain PyTorch fp32 references (through the C ABI)."""
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


def _tune(**kw):
    """A uvl_tuning for ONE call (include/uvltrack_hip.h): the library keeps no process-global tuning state, so a forced kernel
    form cannot leak into another test.  UVL_TEST_ATTN_CFG is a development aid: every attention case on one forced variant."""
    import os
    from uvltrack_amd import _native
    if os.environ.get("UVL_TEST_ATTN_CFG") and "attn_cfg" not in kw:
        kw["attn_cfg"] = int(os.environ["UVL_TEST_ATTN_CFG"])
    return _native.UvlTuning(**kw) if kw else None


def _tref(t):
    return t.ref() if t is not None else None


def _stream():
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


def _p(t):
    return C.c_void_p(t.data_ptr())


def _rand(shape, seed, scale=1.0):
    g = torch.Generator(device="cpu").manual_seed(seed)
    return (torch.randn(shape, generator=g) * scale).cuda()


def _chk(rc, lib):
    assert rc == 0, lib.uvl_last_error().decode()


@pytest.mark.parametrize("M,N,K", [(553, 2304, 768), (513, 768, 768), (40, 3072, 768), (321, 768, 3072),
                                   (1, 64, 64), (65, 128, 128), (873, 1024, 4096), (4424, 3072, 768), (256, 32, 576)])
@pytest.mark.parametrize("mode", ["bf16", "gelu", "relu", "f32", "f32_acc"])
def test_linear(lib, M, N, K, mode):
    x = _rand((M, K), 1).bfloat16()
    # asymmetric weights: a transposed / permuted fragment layout cannot pass
    w = (_rand((N, K), 2, 1.0 / math.sqrt(K)) + torch.linspace(-0.02, 0.03, N).cuda()[:, None]).bfloat16()
    b = _rand((N,), 3, 0.5)
    ref = x.float() @ w.float().t() + b
    if mode in ("bf16", "gelu", "relu"):
        act = {"bf16": 0, "gelu": 1, "relu": 2}[mode]
        if act == 1:
            ref = torch.nn.functional.gelu(ref)
        if act == 2:
            ref = torch.relu(ref)
        y = torch.full((M, N), float("nan"), dtype=torch.bfloat16, device="cuda")
        _chk(lib.uvl_linear(_p(x), _p(w), _p(b), _p(y), M, N, K, act, 0, 0, None, _stream()), lib)
        torch.cuda.synchronize()
        err = (y.float() - ref).abs()
        tol = 1e-2 * ref.abs() + 2e-2
        assert bool((err <= tol).all()), "max err %g" % float(err.max())
    else:
        acc = mode == "f32_acc"
        y0 = _rand((M, N), 4) if acc else torch.full((M, N), float("nan"), device="cuda")
        y = y0.clone()
        _chk(lib.uvl_linear(_p(x), _p(w), _p(b), _p(y), M, N, K, 0, 1, int(acc), None, _stream()), lib)
        torch.cuda.synchronize()
        if acc:
            ref = ref + y0
        err = (y - ref).abs().max().item()
        assert err < 2e-3, err


@pytest.mark.parametrize("M,D", [(553, 768), (40, 768), (7, 128), (873, 1024), (1, 256)])
def test_layernorm(lib, M, D):
    x = _rand((M, D), 5, 3.0) + 1.5
    g = _rand((D,), 6) * 0.2 + 1.0
    b = _rand((D,), 7) * 0.1
    for eps in (1e-6, 1e-12):
        ref = torch.nn.functional.layer_norm(x, (D,), g, b, eps)
        yb = torch.empty((M, D), dtype=torch.bfloat16, device="cuda")
        yf = torch.empty((M, D), device="cuda")
        _chk(lib.uvl_layernorm(_p(x), _p(g), _p(b), C.c_float(eps), _p(yb), _p(yf), M, D, _stream()), lib)
        torch.cuda.synchronize()
        assert (yf - ref).abs().max().item() < 2e-5
        assert (yb.float() - ref).abs().max().item() < 3e-2


QSCALE = 0.18033688011112042          # UVL_ATTN_QSCALE = log2(e) / sqrt(64)


def _attention_case(lib, B, H, N, mode, seed, prescaled=True, tune=None):
    tune = tune if tune is not None else _tune()
    D = H * 64
    Npad = (N + 63) // 64 * 64
    x = _rand((B * N, D), seed).bfloat16()
    w = _rand((3 * D, D), seed + 1, 1.0 / math.sqrt(D)).bfloat16()
    bias = _rand((3 * D,), seed + 2, 0.2)
    # poison the padded workspace: nothing beyond N may leak into the result
    q = torch.full((B, H, Npad, 64), float("nan"), dtype=torch.bfloat16, device="cuda")
    k = torch.full_like(q, float("nan"))
    vt = torch.full((B, H, 64, Npad), float("nan"), dtype=torch.bfloat16, device="cuda")
    _chk(lib.uvl_qkv_project(_p(x), _p(w), _p(bias), _p(q), _p(k), _p(vt), B, N, Npad, D, C.c_float(QSCALE if prescaled else 1.0), _tref(tune), _stream()), lib)
    qkv = (x.float() @ w.float().t() + bias).reshape(B, N, 3, H, 64).permute(2, 0, 3, 1, 4)
    torch.cuda.synchronize()
    qs = QSCALE if prescaled else 1.0
    for got, ref in ((q[:, :, :N], qkv[0] * qs), (k[:, :, :N], qkv[1]), (vt[:, :, :, :N].transpose(2, 3), qkv[2])):
        err = (got.float() - ref).abs()
        assert bool((err <= 1e-2 * ref.abs() + 2e-2).all()), "qkv scatter max err %g" % float(err.max())
    g = torch.Generator(device="cpu").manual_seed(seed + 3)
    masked = (torch.rand((B, N), generator=g) < 0.3).cuda()
    masked[:, N // 2] = False
    add = torch.zeros((B, Npad), device="cuda")
    if mode == "fill":
        add[:, :N] = masked.float() * -1e10
    elif mode == "bert":
        add[:, :N] = masked.float() * -10000.0
    elif mode == "bert_all":            # every key masked: BERT's additive mask keeps the score differences
        add[:, :N] = -10000.0
    add[:, N:] = float("nan")           # must be ignored
    o = torch.full((B * N, D), float("nan"), dtype=torch.bfloat16, device="cuda")
    _chk(lib.uvl_attention(_p(q), _p(k), _p(vt), _p(add), _p(o), B, H, N, Npad, int(prescaled), _tref(tune), _stream()), lib)
    torch.cuda.synchronize()
    qf, kf, vf = q[:, :, :N].float(), k[:, :, :N].float(), vt[:, :, :, :N].transpose(2, 3).float()
    s = (qf @ kf.transpose(-1, -2)) * (math.log(2.0) if prescaled else 0.125)      # pre-scaled q: q k^T is the score in log2 units
    if mode == "fill":
        s = s.masked_fill(masked[:, None, None, :], -1e10)
    else:
        s = s + add[:, None, None, :N]
    ref = (s.softmax(-1) @ vf).transpose(1, 2).reshape(B * N, D)
    err = (o.float() - ref).abs().max().item()
    assert err < 3e-2, "attention max err %g (B=%d H=%d N=%d %s)" % (err, B, H, N, mode)


@pytest.mark.parametrize("N", [1, 31, 32, 33, 40, 64, 65, 321, 361, 513, 553, 681, 873])
def test_attention_token_counts(lib, N):
    _attention_case(lib, 1, 2, N, "fill", 10 + N)


@pytest.mark.parametrize("cfg", [6, 11, 16, 17, 20, 21, 30, 31, 32, 33])
@pytest.mark.parametrize("M,N,K,act", [(777, 512, 192, 0), (300, 256, 64, 1), (6200, 3072, 768, 1), (1000, 256, 128, 0), (513, 768, 448, 2)])
def test_linear_tile_forms_forced(lib, cfg, M, N, K, act):
    """The tile / ring / wave-role forms of the batched GEMM on shapes the heuristic would not give them: 128x128 and 256x256 tiles,
    32-wide K stages with 4 / 5 ring stages (cfg 16 / 17), the producer-wave form (cfg 20 / 21: 2 / 4 extra waves issue every
    LDS-DMA instruction, the four consumer waves only read fragments and issue MFMAs) and the phase-pipelined 256-wide tiles (cfg 30 /
    31: gemm_pipe_body, two wave groups half a phase apart; K = 64 falls back to the plain loop) -- ragged M, one / two /
    three / seven K steps, a long K loop, GELU / ReLU / no activation."""
    x = _rand((M, K), 1).bfloat16()
    w = (_rand((N, K), 2, 1.0 / math.sqrt(K)) + torch.linspace(-0.02, 0.03, N).cuda()[:, None]).bfloat16()
    b = _rand((N,), 3, 0.5)
    ref = x.float() @ w.float().t() + b
    if act == 1:
        ref = torch.nn.functional.gelu(ref)
    elif act == 2:
        ref = torch.relu(ref)
    y = torch.full((M, N), float("nan"), dtype=torch.bfloat16, device="cuda")
    _chk(lib.uvl_linear(_p(x), _p(w), _p(b), _p(y), M, N, K, act, 0, 0, _tune(gemm_cfg=cfg).ref(), _stream()), lib)
    torch.cuda.synchronize()
    err = (y.float() - ref).abs()
    assert bool((err <= 1e-2 * ref.abs() + 2e-2).all()), "cfg %d max err %g" % (cfg, float(err.max()))


@pytest.mark.parametrize("M,N,K,mode", [(777, 512, 192, "bf16"), (300, 256, 64, "gelu"), (6200, 3072, 768, "gelu"), (1000, 256, 128, "relu"),
                                         (513, 768, 448, "f32"), (6984, 1024, 4096, "f32_acc"), (130, 256, 1024, "f32_acc"), (6664, 4096, 1024, "bf16")])
def test_linear_direct_to_register_form(lib, M, N, K, mode):
    """gemm_dr_kernel (cfg 36): 128 x 256 tiles on four waves, two workgroups per CU, A through a four-stage LDS ring, W fragments loaded
    straight into registers from the fragment-native weight image of uvl_pack_weight, K loop in generated assembly.  Ragged M, 1 / 2 / 3 /
    7 / 12 / 16 / 64 K tiles (the loop is unrolled four times with clamped tile indices), every epilogue."""
    x = _rand((M, K), 1).bfloat16()
    w = (_rand((N, K), 2, 1.0 / math.sqrt(K)) + torch.linspace(-0.02, 0.03, N).cuda()[:, None]).bfloat16()
    b = _rand((N,), 3, 0.5)
    wp = torch.empty_like(w)
    _chk(lib.uvl_pack_weight(_p(w), _p(wp), N, K, _stream()), lib)
    ref = x.float() @ w.float().t() + b
    t = _tune(gemm_cfg=36)
    if mode in ("bf16", "gelu", "relu"):
        act = {"bf16": 0, "gelu": 1, "relu": 2}[mode]
        y = torch.full((M, N), float("nan"), dtype=torch.bfloat16, device="cuda")
        _chk(lib.uvl_linear_pk(_p(x), _p(w), _p(wp), _p(b), _p(y), M, N, K, act, 0, 0, t.ref(), _stream()), lib)
        r = torch.nn.functional.gelu(ref) if act == 1 else torch.relu(ref) if act == 2 else ref
        torch.cuda.synchronize()
        err = (y.float() - r).abs()
        assert bool((err <= 1e-2 * r.abs() + 2e-2).all()), "max err %g" % float(err.max())
    else:
        acc = mode == "f32_acc"
        y0 = _rand((M, N), 4) if acc else torch.full((M, N), float("nan"), device="cuda")
        y = y0.clone()
        _chk(lib.uvl_linear_pk(_p(x), _p(w), _p(wp), _p(b), _p(y), M, N, K, 0, 1, int(acc), t.ref(), _stream()), lib)
        torch.cuda.synchronize()
        err = (y - (ref + y0 if acc else ref)).abs().max().item()
        assert err < 3e-3, err


@pytest.mark.parametrize("M,N,K", [(6984, 1024, 1024), (6664, 1024, 4096), (2100, 256, 768), (2049, 512, 832), (5000, 768, 3072), (2128, 256, 1024), (2241, 256, 768)])
def test_residual_rows_requested_inside_the_k_loop(lib, M, N, K):
    """The in-place f32 residual epilogue of the 128 x 256 pipelined GEMM (cfg 31; proj / fc2 of many-sequence frames, block.py:29-32,57-60) with its
    residual rows requested INSIDE the K loop (gemm_pipe128_body PRE: one 16-byte load per lane and phase over eight K tiles, counted into the
    loop's vmcnt waits): 12 (the minimum), 13, 16, 48 and 64 K tiles, ragged last row tile.  Against torch, and BIT-IDENTICAL to the form that
    loads them in the epilogue (uvl_tuning.res_pre = 0) -- a mis-counted wait would show up as stale tile data in one of the two."""
    x = _rand((M, K), 21).bfloat16()
    w = (_rand((N, K), 22, 1.0 / math.sqrt(K)) + torch.linspace(-0.02, 0.03, N).cuda()[:, None]).bfloat16()
    b = _rand((N,), 23, 0.5)
    y0 = _rand((M, N), 24)
    ref = x.float() @ w.float().t() + b + y0
    out = {}
    for pre in (1, 0):                             # res_pre = 2 forces the window (the default rule stops at K = 2048), 0 = the epilogue's own loads
        for rep in range(3):                       # repeated: the window's loads race nothing
            y = y0.clone()
            _chk(lib.uvl_linear(_p(x), _p(w), _p(b), _p(y), M, N, K, 0, 1, 1, _tune(gemm_cfg=31, res_pre=2 * pre).ref(), _stream()), lib)
            torch.cuda.synchronize()
            if rep == 0:
                out[pre] = y
            else:
                assert torch.equal(y, out[pre])
    assert (out[1] - ref).abs().max().item() < 3e-3
    assert torch.equal(out[1], out[0])


@pytest.mark.parametrize("dr", [0, -1])
def test_linear_heuristic_with_and_without_the_direct_to_register_form(lib, dr):
    """The un-forced choice at >= 2048 rows with a packed weight at hand: cfg 36 by default, the eight-wave tile grids with
    uvl_tuning.gemm_dr = 0 -- same function, both against torch; a ragged last tile, K = 576 (nine K tiles), GELU."""
    M, N, K = 8200, 512, 576
    x = _rand((M, K), 11).bfloat16()
    w = _rand((N, K), 12, 1.0 / math.sqrt(K)).bfloat16()
    b = _rand((N,), 13, 0.5)
    wp = torch.empty_like(w)
    _chk(lib.uvl_pack_weight(_p(w), _p(wp), N, K, _stream()), lib)
    ref = torch.nn.functional.gelu(x.float() @ w.float().t() + b)
    y = torch.full((M, N), float("nan"), dtype=torch.bfloat16, device="cuda")
    _chk(lib.uvl_linear_pk(_p(x), _p(w), _p(wp), _p(b), _p(y), M, N, K, 1, 0, 0, _tune(gemm_dr=dr).ref(), _stream()), lib)
    torch.cuda.synchronize()
    err = (y.float() - ref).abs()
    assert bool((err <= 1e-2 * ref.abs() + 2e-2).all()), "gemm_dr %d max err %g" % (dr, float(err.max()))


def test_qkv_project_direct_to_register_form(lib):
    """The QKV scatter epilogue of gemm_dr_kernel (q scaled, k, V^T token-contiguous) at the configs[4] shape (8 x 873 rows, D = 1024) against
    torch and against the tile-grid kernel's output."""
    B, N, D = 8, 873, 1024
    Npad = 896
    x = _rand((B * N, D), 21).bfloat16()
    w = _rand((3 * D, D), 22, 1.0 / math.sqrt(D)).bfloat16()
    b = _rand((3 * D,), 23, 0.5)
    wp = torch.empty_like(w)
    _chk(lib.uvl_pack_weight(_p(w), _p(wp), 3 * D, D, _stream()), lib)
    ref = (x.float() @ w.float().t() + b).reshape(B, N, 3, D // 64, 64)
    qr = (ref[:, :, 0] * QSCALE).permute(0, 2, 1, 3)
    kr = ref[:, :, 1].permute(0, 2, 1, 3)
    vr = ref[:, :, 2].permute(0, 2, 3, 1)
    for cfg in (31, 36):
        q = torch.zeros((B, D // 64, Npad, 64), dtype=torch.bfloat16, device="cuda")
        k = torch.zeros_like(q)
        vt = torch.zeros((B, D // 64, 64, Npad), dtype=torch.bfloat16, device="cuda")
        _chk(lib.uvl_qkv_project_pk(_p(x), _p(w), _p(wp), _p(b), _p(q), _p(k), _p(vt), B, N, Npad, D, C.c_float(QSCALE), _tune(gemm_cfg=cfg).ref(), _stream()), lib)
        torch.cuda.synchronize()
        for got, want in ((q[:, :, :N], qr), (k[:, :, :N], kr), (vt[:, :, :, :N], vr)):
            err = (got.float() - want).abs()
            assert bool((err <= 1e-2 * want.abs() + 2e-2).all()), (cfg, float(err.max()))


@pytest.mark.parametrize("cfg", [8, 10, 11])
@pytest.mark.parametrize("B,H,N,mode", [(1, 2, 1, "fill"), (1, 2, 33, "fill"), (2, 3, 64, "none"), (2, 3, 65, "bert"), (1, 2, 257, "fill"),
                                         (3, 2, 321, "bert_all"), (2, 4, 553, "fill"), (1, 2, 1100, "fill")])
def test_attention_batched_kernels_forced(lib, cfg, B, H, N, mode):
    """The kernels of the batched regime -- attn_stream_kernel (cfg 8), attn_w64_kernel (cfg 10: 64 queries per wave, pass 1
    without a running maximum, exact pass 2 on demand) and its hand-scheduled persistent form attn_p64_kernel (cfg 11; one key tile
    falls back to cfg 10) -- on shapes the heuristic would not give them: one key tile, ragged tails,
    waves without queries, every key masked (pass 2 of the w64 kernel), more than 16 key tiles."""
    _attention_case(lib, B, H, N, mode, 300 + N, tune=_tune(attn_cfg=cfg))


@pytest.mark.parametrize("wgs", [8, 64, 512])
@pytest.mark.parametrize("B,H,N,mode", [(8, 4, 321, "fill"), (6, 3, 553, "bert"), (5, 2, 200, "bert_all"), (16, 4, 130, "none")])
def test_attention_p64_persistent_walk(lib, wgs, B, H, N, mode):
    """attn_p64_kernel (cfg 11, generated assembly) with few persistent workgroups: every workgroup walks several items (flags of item i
    are read at the start of item i + 1, the ring is reused across items, waves without queries and workgroups without an item take
    their own paths); `bert_all` flags EVERY item for the exact pass, so the flagged-item mask is exercised beyond its first bit."""
    _attention_case(lib, B, H, N, mode, 500 + N + wgs, tune=_tune(attn_cfg=11, attn_wgs=wgs))


def test_attention_p64_flagged_item_late_in_the_walk(lib):
    """One key that overflows exp2 in an item that is NOT the first one its workgroup walks: pass 1 of attn_p64_kernel has no running
    maximum, the item's row sums leave [2^-100, 2^100], its bit in the mask sends exactly that item through the exact pass."""
    B, H, N = 6, 4, 300
    Npad = (N + 63) // 64 * 64
    g = torch.Generator(device="cpu").manual_seed(9)
    q = torch.zeros((B, H, Npad, 64), dtype=torch.bfloat16, device="cuda")
    k = torch.zeros_like(q)
    vt = torch.zeros((B, H, 64, Npad), dtype=torch.bfloat16, device="cuda")
    q[:, :, :N] = (torch.randn((B, H, N, 64), generator=g) * QSCALE).cuda().bfloat16()
    k[:, :, :N] = torch.randn((B, H, N, 64), generator=g).cuda().bfloat16()
    vt[:, :, :, :N] = torch.randn((B, H, 64, N), generator=g).cuda().bfloat16()
    k[4, 2, 133] = q[4, 2, 17] * (60.0 / QSCALE)             # scores of ~ +-500 log2 units against query 17 of (4, 2): exp2 overflows
    add = torch.zeros((B, Npad), device="cuda")
    o = torch.empty((B * N, H * 64), dtype=torch.bfloat16, device="cuda")
    _chk(lib.uvl_attention(_p(q), _p(k), _p(vt), _p(add), _p(o), B, H, N, Npad, 1, _tref(_tune(attn_cfg=11, attn_wgs=8)), _stream()), lib)
    torch.cuda.synchronize()
    s = (q[:, :, :N].float() @ k[:, :, :N].float().transpose(-1, -2)) * math.log(2.0)
    ref = (s.softmax(-1) @ vt[:, :, :, :N].transpose(2, 3).float()).transpose(1, 2).reshape(B * N, H * 64)
    assert torch.isfinite(o.float()).all()
    assert (o.float() - ref).abs().max().item() < 3e-2


@pytest.mark.parametrize("mode", ["none", "fill", "bert", "bert_all"])
@pytest.mark.parametrize("B,H,N", [(1, 12, 553), (3, 12, 40), (8, 16, 681), (2, 12, 321), (32, 12, 553), (24, 16, 873)])
def test_attention_modes(lib, B, H, N, mode):
    _attention_case(lib, B, H, N, mode, 77)


@pytest.mark.parametrize("B,H,N", [(1, 12, 553), (32, 12, 361), (24, 16, 681)])
def test_attention_raw_q(lib, B, H, N):
    """q_prescaled = 0: the kernel applies log2(e)/8 itself (single-sequence and streaming kernels)."""
    _attention_case(lib, B, H, N, "fill", 78, prescaled=False)


@pytest.mark.parametrize("B,H,N,tile", [(1, 1, 200, 2), (40, 8, 520, 5), (40, 8, 520, 8)])
def test_attention_spike_forces_rescale(lib, B, H, N, tile):
    """A late key with a huge score forces the online-softmax rescale (attn_body) / the redo of a speculative tile
    (attn_stream_kernel: B*H large enough for the batched heuristic) -- rare on random data, so it is forced here.
    q is pre-scaled as in the frame (one rounding of q; with scores of +-500 a second rounding of q alone moves them by 0.1)."""
    Npad = (N + 63) // 64 * 64
    g = torch.Generator(device="cpu").manual_seed(5)
    q = torch.zeros((B, H, Npad, 64), dtype=torch.bfloat16, device="cuda")
    k = torch.zeros_like(q)
    vt = torch.zeros((B, H, 64, Npad), dtype=torch.bfloat16, device="cuda")
    q[:, :, :N] = (torch.randn((B, H, N, 64), generator=g) * QSCALE).cuda().bfloat16()
    k[:, :, :N] = torch.randn((B, H, N, 64), generator=g).cuda().bfloat16()
    vt[:, :, :, :N] = torch.randn((B, H, 64, N), generator=g).cuda().bfloat16()
    key = 64 * tile + 5
    for bb in range(0, B, 7):
        k[bb, bb % H, key] = q[bb, bb % H, 7 + bb] * (4.0 / QSCALE)          # this key dominates query 7 + bb of head bb % H
    k[B - 1, H - 1, key + 1] = q[B - 1, H - 1, 3] * (40.0 / QSCALE)          # and one that overflows exp2 against the stale maximum
    add = torch.zeros((B, Npad), device="cuda")
    o = torch.empty((B * N, H * 64), dtype=torch.bfloat16, device="cuda")
    _chk(lib.uvl_attention(_p(q), _p(k), _p(vt), _p(add), _p(o), B, H, N, Npad, 1, _tref(_tune()), _stream()), lib)
    torch.cuda.synchronize()
    s = (q[:, :, :N].float() @ k[:, :, :N].float().transpose(-1, -2)) * math.log(2.0)
    ref = (s.softmax(-1) @ vt[:, :, :, :N].transpose(2, 3).float()).transpose(1, 2).reshape(B * N, H * 64)
    assert torch.isfinite(o.float()).all()
    assert (o.float() - ref).abs().max().item() < 3e-2


@pytest.mark.parametrize("B,F,cin,cout,slabs", [(1, 16, 768, 256, True), (1, 16, 768, 256, False), (16, 16, 256, 128, True), (2, 24, 1024, 256, True),
                                                (16, 24, 128, 64, False), (1, 16, 64, 32, False), (16, 16, 64, 32, True),
                                                (8, 24, 1024, 256, True), (7, 24, 1024, 256, False),
                                                # round 6: the layers of a one-sequence frame that run WITHOUT slabs, K quarters / halves on the wave groups of one workgroup
                                                # (conv_fin_kernel<32>: K = 2304, <64>: K = 1152 -- 18 K tiles do not divide by four --, and UVLTrack-L's 24 x 24 map)
                                                (1, 16, 256, 128, True), (1, 16, 128, 64, True), (1, 24, 256, 128, True), (1, 24, 128, 64, False), (2, 16, 256, 128, True)])
def test_conv_tower_layer(lib, B, F, cin, cout, slabs):
    """One layer of the four conv towers (implicit GEMM over NHWC tokens, zero-page padding, towers as groups, BatchNorm folded)
    against F.conv2d -> BatchNorm2d(eval) -> ReLU in fp32 (heads/utils.py:126-131) at batch 1 and 16: the one-sequence split-K
    variant, the 64x64 and the 128x128 tile variants, the 32-channel last layer, and the first layer of 8 / 7 UVLTrack-L sequences (M = 4608 / 4032:
    288 tiles of 128 x 128 on two workgroups per CU / the 64x64 grid)."""
    S = F * F
    first = cin in (768, 1024)                     # first layer: the four towers read the same channels
    x_ld = cin if first else 4 * cin
    goff = (C.c_int32 * 4)(*([0, 0, 0, 0] if first else [g * cin for g in range(4)]))
    x = (_rand((B, S, x_ld), 31, 1.0)).bfloat16()
    wpk = torch.empty((4, cout, 9 * cin), dtype=torch.bfloat16, device="cuda")
    bpk = torch.empty((4 * cout,), device="cuda")
    ref = []
    for g in range(4):
        w = _rand((cout, cin, 3, 3), 40 + g, 1.0 / math.sqrt(9 * cin))
        b = _rand((cout,), 50 + g, 0.1)
        bn_w = _rand((cout,), 60 + g, 0.2) + 1.0
        bn_b = _rand((cout,), 70 + g, 0.1)
        mu = _rand((cout,), 80 + g, 0.1)
        var = _rand((cout,), 90 + g, 0.1).abs() + 0.5
        _chk(lib.uvl_fold_conv_bn(_p(w), _p(b), _p(bn_w), _p(bn_b), _p(mu), _p(var), C.c_void_p(wpk[g].data_ptr()),
                                  C.c_void_p(bpk[g * cout:].data_ptr()), cout, cin, _stream()), lib)
        xg = x[:, :, goff[g]:goff[g] + cin].float().reshape(B, F, F, cin).permute(0, 3, 1, 2)
        y = torch.nn.functional.conv2d(xg, w, b, padding=1)
        y = torch.nn.functional.batch_norm(y, mu, var, bn_w, bn_b, training=False, eps=1e-5)
        ref.append(torch.relu(y).permute(0, 2, 3, 1).reshape(B * S, cout))
    ref = torch.cat(ref, dim=1)
    y = torch.full((B * S, 4 * cout), float("nan"), dtype=torch.bfloat16, device="cuda")
    scratch = torch.empty((8 * B * S * 4 * cout,), device="cuda") if slabs else None
    _chk(lib.uvl_conv_tower_layer(_p(x), B, F, x_ld, goff, cin, cout, _p(wpk), _p(bpk), _p(y),
                                  C.c_void_p(scratch.data_ptr() if slabs else 0), None, _stream()), lib)
    torch.cuda.synchronize()
    err = (y.float() - ref).abs()
    assert bool(torch.isfinite(y.float()).all())
    assert bool((err <= 2e-2 * ref.abs() + 2e-2).all()), "conv tower max err %g" % float(err.max())


@pytest.mark.parametrize("B,F,cin,form,cont_ch,offset_sigmoid,joint_cls", [
    (1, 16, 64, 1, 3, 1, 0), (1, 16, 64, 0, 3, 1, 0), (3, 16, 64, 1, 2, 0, 1), (3, 16, 64, 0, 2, 0, 1), (40, 16, 64, 1, 3, 1, 1),
    (2, 24, 64, 0, 3, 1, 0), (2, 8, 128, 0, 2, 1, 0)])
def test_head_end(lib, B, F, cin, form, cont_ch, offset_sigmoid, joint_cls):
    """The end of the box head from the towers' third layer onwards (last conv3x3 + BN + ReLU, the 1x1 convs, sigmoids, size select by flag, convert2bbox + argmax:
    modality_adaptive_box_head.py:71-94,108-119) against fp32 torch, as ONE launch (form 1: head_fin_kernel, one workgroup per sample) and as the conv launch +
    head_tail_kernel (form 0), for flags 0 / 1 / 2, two and three cont_score channels, both offset modes."""
    S, cout = F * F, cin // 2
    x = torch.relu(_rand((B, S, 4 * cin), 131, 1.0)).bfloat16()
    wpk = torch.empty((4, cout, 9 * cin), dtype=torch.bfloat16, device="cuda")
    bpk = torch.empty((4 * cout,), device="cuda")
    rows = [1, 2, 2, 2]
    w1 = torch.cat([_rand((rows[g], cout), 140 + g, 2.0 / math.sqrt(cout)) for g in range(4)]).contiguous()
    b1 = torch.cat([_rand((7,), 150, 0.3), torch.zeros(1, device="cuda")])
    maps = []
    for g in range(4):
        w = _rand((cout, cin, 3, 3), 40 + g, 1.5 / math.sqrt(9 * cin))
        b = _rand((cout,), 50 + g, 0.1)
        bn_w = _rand((cout,), 60 + g, 0.2) + 1.0
        bn_b = _rand((cout,), 70 + g, 0.1)
        mu = _rand((cout,), 80 + g, 0.1)
        var = _rand((cout,), 90 + g, 0.1).abs() + 0.5
        _chk(lib.uvl_fold_conv_bn(_p(w), _p(b), _p(bn_w), _p(bn_b), _p(mu), _p(var), C.c_void_p(wpk[g].data_ptr()), C.c_void_p(bpk[g * cout:].data_ptr()), cout, cin, _stream()), lib)
        xg = x[:, :, g * cin:(g + 1) * cin].float().reshape(B, F, F, cin).permute(0, 3, 1, 2)
        y = torch.nn.functional.conv2d(xg, w, b, padding=1)
        y = torch.relu(torch.nn.functional.batch_norm(y, mu, var, bn_w, bn_b, training=False, eps=1e-5))
        k0 = sum(rows[:g])
        maps.append(torch.nn.functional.conv2d(y, w1[k0:k0 + rows[g], :, None, None], b1[k0:k0 + rows[g]]).reshape(B, rows[g], S))
    cont = _rand((B, S, cont_ch), 160, 1.0).contiguous()
    flag = (torch.arange(B, device="cuda") % 3).to(torch.int64)
    jj, ii = torch.meshgrid(torch.arange(F, device="cuda").float(), torch.arange(F, device="cuda").float(), indexing="xy")
    coord = torch.stack([jj.reshape(-1), ii.reshape(-1)]) + (0.0 if offset_sigmoid else 0.5)
    cls_ref = torch.sigmoid(maps[0][:, 0])
    off = torch.sigmoid(maps[1]) if offset_sigmoid else maps[1]
    size = torch.where((flag == 1)[:, None, None], torch.sigmoid(maps[3]), torch.sigmoid(maps[2]))
    box_ref = torch.cat([(coord[None] + off) / F, size], 1).transpose(1, 2)
    score_ref = cls_ref * cont.softmax(-1)[..., 0]
    scratch = torch.empty((max(B * S * 4 * cout, 4 * 32 * 9 * 64),), dtype=torch.bfloat16, device="cuda")
    cls = torch.full((B, S), float("nan"), device="cuda")
    cls_test = torch.full((B, S), float("nan"), device="cuda")
    bbox = torch.full((B, S, 4), float("nan"), device="cuda")
    pred = torch.full((B, 4), float("nan"), device="cuda")
    arg = torch.full((B,), -1, dtype=torch.int64, device="cuda")
    _chk(lib.uvl_head_end(_p(x), B, F, cin, _p(wpk), _p(bpk), _p(w1), _p(b1), _p(cont), cont_ch, _p(flag), _p(coord.contiguous()), offset_sigmoid, joint_cls, form,
                          _p(scratch), _p(cls), _p(cls_test), _p(bbox), _p(pred), _p(arg), _stream()), lib)
    torch.cuda.synchronize()
    assert float((cls_test - cls_ref).abs().max()) < 6e-3
    assert float((cls - (score_ref if joint_cls else cls_ref)).abs().max()) < 6e-3
    assert float((bbox - box_ref).abs().max()) < (6e-3 if offset_sigmoid else 3e-2 / F * 4)
    bi = torch.arange(B, device="cuda")
    assert bool((arg >= 0).all()) and bool((arg < S).all())
    assert bool((score_ref[bi, arg] >= score_ref.max(-1).values - 6e-3).all())          # the winner, up to positions that tie within the precision
    assert torch.equal(pred, bbox[bi, arg])


def test_head_end_forms_agree_and_a_nan_map_selects_position_zero(lib):
    """Both forms on the same inputs: the same maps within bf16 accumulation-order noise and the same winner; an all-NaN score map (NaN cont_score) gives argmax 0
    and pred_boxes = bbox_map[0] in both (torch.argmax would return the first NaN: position 0 as well)."""
    B, F, cin, cout = 5, 16, 64, 32
    S = F * F
    x = torch.relu(_rand((B, S, 4 * cin), 231, 1.0)).bfloat16()
    wpk = _rand((4, cout, 9 * cin), 232, 1.5 / math.sqrt(9 * cin)).bfloat16()
    bpk = _rand((4 * cout,), 233, 0.1)
    w1 = _rand((7, cout), 234, 2.0 / math.sqrt(cout))
    b1 = _rand((8,), 235, 0.3)
    cont = _rand((B, S, 3), 236, 1.0)
    cont[2] = float("nan")
    flag = (torch.arange(B, device="cuda") % 3).to(torch.int64)
    coord = _rand((2, S), 237, 1.0).abs()
    outs = []
    for form in (0, 1):
        scratch = torch.empty((B * S * 4 * cout,), dtype=torch.bfloat16, device="cuda")
        cls_test = torch.empty((B, S), device="cuda")
        bbox = torch.empty((B, S, 4), device="cuda")
        pred = torch.empty((B, 4), device="cuda")
        arg = torch.full((B,), -1, dtype=torch.int64, device="cuda")
        _chk(lib.uvl_head_end(_p(x), B, F, cin, _p(wpk), _p(bpk), _p(w1), _p(b1), _p(cont), 3, _p(flag), _p(coord), 1, 1, form, _p(scratch), None, _p(cls_test), _p(bbox), _p(pred),
                              _p(arg), _stream()), lib)
        torch.cuda.synchronize()
        outs.append((cls_test, bbox, pred, arg))
    assert float((outs[0][0] - outs[1][0]).abs().max()) < 2e-3 and float((outs[0][1] - outs[1][1]).abs().max()) < 2e-3
    for cls_test, bbox, pred, arg in outs:
        assert int(arg[2]) == 0 and torch.equal(pred[2], bbox[2, 0])
    keep = [0, 1, 3, 4]
    score = outs[0][0] * cont.softmax(-1)[..., 0]
    gap = score[keep].topk(2, -1).values
    clear = (gap[:, 0] - gap[:, 1]) > 4e-3                       # samples whose winner is not a near-tie
    assert bool((outs[0][3][keep][clear] == outs[1][3][keep][clear]).all())
    assert float((outs[0][2][keep] - outs[1][2][keep])[clear].abs().max() if bool(clear.any()) else 0.0) < 2e-3


@pytest.mark.parametrize("B,D,nz,nx,T,mode,form,skip", [(3, 768, 64, 256, 40, "cls", 0, 0), (3, 768, 64, 256, 40, "cls", 1, 0), (5, 1024, 64, 576, 40, "cls", 1, 0),
                                                     (4, 768, 64, 256, 40, "mean", 0, 0), (2, 256, 4, 16, 8, "cls", 1, 1), (2, 256, 4, 16, 8, "mean", 0, 1), (7, 64, 1, 7, 5, "cls", 1, 0)])
def test_contrast_logits_match_the_oracle(lib, B, D, nz, nx, T, mode, form, skip):
    """The contrastive logits of a layer (extractor.py:79-93) at kernel level, both implementations (contrast_kernel; the job of the LayerNorm-free frames'
    QKV launches) against the numpy oracle: flags 0 / 1 / 2, 'cls' and 'mean' text tokens incl. a sentence of one token, text skipped, UVLTrack-B and -L widths."""
    import numpy as np
    from oracle import uvl_oracle as O
    rows = 1 + nz + nx + T
    x = _rand((B, rows, D), 300 + D, 1.0)
    x[:, 1 + nz:1 + nz + nx] += 0.5 * x[:, :1]                   # some alignment with the vis token: logits not all near zero
    mask = torch.zeros((B, T), dtype=torch.uint8, device="cuda")
    for b in range(B):
        mask[b, :max(1, (b * 7) % T)] = 1
    flag = (torch.arange(B, device="cuda") % 3).to(torch.int64) * (0 if skip else 1)       # (the frame skips the text branch only when every flag is 0)
    scale = torch.tensor([math.log(14.3)], device="cuda")
    logits = torch.full((B, 2, nx), float("nan"), device="cuda")
    _chk(lib.uvl_contrast_logits(_p(x), B, rows, D, nz, nx, 1 + nz + nx, T, _p(mask), 1 if mode == "mean" else 0, skip, _p(flag), _p(scale), _p(logits), 1, 2, form, _stream()), lib)
    torch.cuda.synchronize()
    xn = x.cpu().numpy()
    img, txt = xn[:, :1 + nz + nx], xn[:, 1 + nz + nx:]
    fl = flag.cpu().numpy()
    ref = O.backbone_contrast({"backbone.logit_scale": scale.cpu().numpy()}, img, txt, mask.cpu().numpy().astype(bool), fl, nz, mode)[..., 0]
    got = logits.cpu().numpy()
    assert np.isnan(got[:, 0]).all()                               # the other slot is not touched
    np.testing.assert_allclose(got[:, 1], ref, rtol=1e-4, atol=1e-4)


def test_anno2mask_matches_oracle(lib):
    """uvl_anno2mask against the numpy restatement of the tracker's anno2mask (tracker:183-194), bit-exact, incl. boxes whose
    centre cell is the only one set and boxes touching the border."""
    import numpy as np
    from oracle.tracker_oracle import anno2mask
    rng = np.random.RandomState(0)
    for size in (8, 16, 24):
        xy = rng.uniform(0.0, 0.8, (64, 2))
        wh = np.minimum(rng.uniform(0.0, 0.6, (64, 2)), 0.999 - xy)
        wh[:8] = 1e-3                                # degenerate boxes: only the centre cell
        boxes = np.concatenate([xy, wh], 1).astype(np.float32)
        d_boxes = torch.from_numpy(boxes).cuda()
        mask = torch.empty((64, size * size), dtype=torch.uint8, device="cuda")
        _chk(lib.uvl_anno2mask(_p(d_boxes), 64, size, _p(mask), _stream()), lib)
        torch.cuda.synchronize()
        np.testing.assert_array_equal(mask.cpu().numpy().astype(bool), anno2mask(boxes, size))
