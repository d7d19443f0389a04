"""GPU parity of the whole HIP forward pass (through the C ABI) against the committed reference outputs
and against the CPU oracle on the same seeded inputs."""
import numpy as np
import pytest
import torch

from tests.golden_util import list_cases, load_case, rebuild_inputs, rebuild_weights
from tests.parity_util import compare_outputs, fmt_report

pytestmark = pytest.mark.gpu

_engines = {}


def _engine(meta, spec):
    from uvltrack_amd.engine import HipEngine
    key = (meta["name"].split("_")[0], tuple(sorted(spec.to_dict().items(), key=str).__repr__()), meta["weight_seed"], max(8, meta["batch"]))
    if key not in _engines:
        _engines.clear()                       # one big model resident at a time
        eng = HipEngine(spec, torch.device("cuda:0"), max_batch=max(8, meta["batch"]))
        eng.load_state_dict(rebuild_weights(meta, spec, include_unused=True))
        _engines[key] = eng
    return _engines[key]


def _run(eng, inp, **kw):
    t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).cuda()
    out = eng.forward(t(inp["template"]), t(inp["search"]), t(inp["ids"]), t(inp["mask"]), t(inp["prompt"]), t(inp["flag"]), **kw)
    torch.cuda.synchronize()
    return {k: v.cpu().numpy() for k, v in out.items() if torch.is_tensor(v)}


@pytest.mark.parametrize("name", list_cases())
def test_forward_matches_reference_fixture(name):
    meta, spec, ref = load_case(name)
    inp = rebuild_inputs(meta, spec)
    got = _run(_engine(meta, spec), inp)
    ok, rep = compare_outputs(got, ref, depth=spec.depth)
    assert ok, "\n" + fmt_report(rep)


@pytest.mark.parametrize("key,value", [("attn_cfg", 10), ("attn_cfg", 8), ("gemm_cfg", 21), ("gemm_cfg", 11), ("gemm_cfg", 16), ("gemm_cfg", 30), ("gemm_cfg", 31), ("gemm_dr", 1), ("gemm_dr", 0)])
@pytest.mark.parametrize("name", ["b_z256_x256_b8", "l_z256_x384"])
def test_batched_kernel_forms_match_reference_fixture(name, key, value):
    """The kernels of the many-sequence regime pinned to the reference's own outputs: the fixtures' batches are too small for the
    heuristics to pick attn_w64_kernel (64 queries per wave, pass 1 without a running maximum), the producer-wave GEMM, the 256x256 tiles
    (plain and phase-pipelined) or the 32-wide K stages, so they are forced through the tuning hook and the whole frame is compared with the
    reference fixture at the gates of the default path; gemm_dr = 1 puts the direct-to-register GEMM (cfg 36) on the residual epilogue too,
    gemm_dr = 0 takes it off everything (the batch-8 fixture has the packed weights; the two-sequence one runs its default either way) (mixed flags, padded text keys masked, one all-padding text in the batch-8 case)."""
    meta, spec, ref = load_case(name)
    inp = rebuild_inputs(meta, spec)
    eng = _engine(meta, spec)
    with eng.tuned(**{key: value}):            # the override lives in this engine's handle and is reset on exit
        got = _run(eng, inp)
    ok, rep = compare_outputs(got, ref, depth=spec.depth)
    assert ok, "\n" + fmt_report(rep)


def test_default_path_of_a_32_sequence_frame_matches_the_reference():
    """b_z256_x256_b32 (round 6): 17,696 rows, the size from which the DEFAULT path runs the text branch as riders of the large-tile kernels
    (gemm_dr_pair_kernel, gemm_pipe_pair_kernel<256,1>, attn_p64_rider_kernel on the 1152-item persistent walk).  Until this fixture the path was pinned only to
    this library's own one-sequence runs; here the reference's outputs for the 32 samples pin it directly (batch independence is the reference's property,
    modality_unified_feature_extractor.py:43-77), un-forced, and the profile must show that the rider kernels ran."""
    meta, spec, ref = load_case("b_z256_x256_b32")
    inp = rebuild_inputs(meta, spec)
    eng = _engine(meta, spec)
    got = _run(eng, inp)
    ok, rep = compare_outputs(got, ref, depth=spec.depth)
    assert ok, "\n" + fmt_report(rep)
    t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).cuda()
    eng.forward(t(inp["template"]), t(inp["search"]), t(inp["ids"]), t(inp["mask"]), t(inp["prompt"]), t(inp["flag"]), profile=True)
    torch.cuda.synchronize()
    kernels = {e["kernel"] for e in eng.profile_entries()}
    assert {"gemm_dr_pair_kernel<2>", "gemm_dr_pair_kernel<0>", "attn_p64_rider_kernel"} <= kernels, sorted(kernels)
    assert any(k.startswith("gemm_pipe_pair_kernel<256,1") for k in kernels), sorted(kernels)
    assert any(k.startswith("ln_pair_kernel") for k in kernels), sorted(kernels)


# BASELINE.json's configs -> (model, template, search, batch, mode, skip_text), the kernels the DEFAULT path must run (prefixes), the ones it must not, launches
_DISPATCH_CASES = {
    "configs[1] B x1 BBOX, text skipped": (("B", 256, 256, 1, 0, True),
        ["gemm_fin_kernel<32>", "gemm_lnf_kernel<2,64,4>", "gemm_lnf_kernel<0,64,4>", "attn_kernel<1,9,1>", "conv_fin_kernel<32>", "conv_fin_kernel<64>", "head_fin_kernel"],
        ["ln_", "gemm_fin_pair", "gemm_lnf_pair", "attn_pair", "text_join", "gemm_dr", "gemm_pipe", "head_tail"], 68),
    "configs[2] B x1 NLBBOX (the headline; configs[0] is the same frame on the reference's CPU path)": (("B", 256, 256, 1, 2, False),
        ["gemm_fin_kernel<32>", "gemm_fin_pair_kernel<32,32>", "gemm_lnf_kernel<2,64,4>", "gemm_lnf_pair_kernel<2,64,4>", "gemm_lnf_pair_kernel<0,64,4>", "attn_pair_kernel<1,9,1>", "attn_kernel<1,9,1>",
         "text_join_kernel", "conv_fin_kernel<32>", "conv_fin_kernel<64>", "head_fin_kernel"],
        ["ln_", "gemm_dr", "gemm_pipe", "contrast", "head_tail"], 69),
    "configs[3] L x1 NLBBOX": (("L", 256, 384, 1, 2, False),
        ["gemm_fin_kernel<64>", "gemm_fin_pair_kernel<64,32>", "gemm_lnf_pair_kernel<2,128,2>", "gemm_lnf_kernel<0,128,2>", "attn_pair_kernel<2,4,2>", "text_join_kernel", "conv_fin_kernel", "head_tail"],
        ["ln_", "gemm_dr", "gemm_pipe", "head_fin"], 130),
    "configs[4] L x8 per GPU NLBBOX": (("L", 256, 384, 8, 2, False),
        ["gemm_dr_pair_kernel<2>", "gemm_dr_pair_kernel<0>", "gemm_dr_kernel<2>", "gemm_dr_kernel<0>", "gemm_pipe_pair_kernel<128,1", "gemm_pipe128_kernel<1,1", "attn_p64_rider_kernel",
         "attn_p64_kernel", "ln_pair_kernel", "ln_kernel"],
        ["gemm_fin", "gemm_lnf", "text_join", "conv_fin"], None),
}


@pytest.mark.parametrize("case", list(_DISPATCH_CASES), ids=lambda c: c.split(" ")[0])
def test_default_dispatch_per_baseline_config(case):
    """One row per BASELINE.json config: which kernels the default path runs there (uvl_api.hip::kDispatch and the kernel-level choosers it points to; DESIGN.md
    "Dispatch").  A heuristic lost in a clean-up once ran a whole round's benchmarks on the old kernels while every forced-form test stayed green; this pins the
    kernel SET (and, for the one-sequence frames, the launch count) per workload the bench reports."""
    from uvltrack_amd import weightgen as wg
    from uvltrack_amd.engine import HipEngine
    from uvltrack_amd.spec import spec_b, spec_l
    (model, tz, xs, B, flag, skip), must, must_not, launches = _DISPATCH_CASES[case]
    spec = spec_b(tz, xs) if model == "B" else spec_l(tz, xs)
    _engines.clear()
    eng = HipEngine(spec, torch.device("cuda:0"), max_batch=B)
    try:
        eng.load_state_dict(wg.make_state_dict(spec, 0, include_unused=False))
        inp = wg.make_inputs(spec, batch=B, seed=5, flags=[flag] * B)
        t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).cuda()
        out = eng.forward(t(inp["template"]), t(inp["search"]), t(inp["ids"]), t(inp["mask"]), t(inp["prompt"]), t(inp["flag"]), skip_text=skip, profile=True)
        torch.cuda.synchronize()
        assert bool(torch.isfinite(out["bbox_map"]).all())
        prof = eng.profile_entries()
        kernels = {e["kernel"] for e in prof}
        n = sum(e["launches"] for e in prof)
        for k in must:
            assert any(x.startswith(k) for x in kernels), (k, sorted(kernels))
        for k in must_not:
            assert not any(x.startswith(k) for x in kernels), (k, sorted(kernels))
        if launches is not None:
            assert n == launches, (n, launches, sorted(kernels))
    finally:
        eng.close()


def test_default_kernel_choice_of_a_many_sequence_frame():
    """Which kernels a frame of >= 2048 rows runs BY DEFAULT (a heuristic lost in a clean-up once ran the whole round's benchmarks on the
    old kernels while every forced-form test stayed green): QKV and fc1 on gemm_dr_kernel (cfg 36, packed weights made at
    finalize), the residual GEMMs (f32 read-modify-write epilogue) on the tile-grid kernels."""
    meta, spec, ref = load_case("b_z256_x256_b8")
    inp = rebuild_inputs(meta, spec)
    eng = _engine(meta, spec)
    t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).cuda()
    eng.forward(t(inp["template"]), t(inp["search"]), t(inp["ids"]), t(inp["mask"]), t(inp["prompt"]), t(inp["flag"]), profile=True)
    torch.cuda.synchronize()
    by_site = {}
    for e in eng.profile_entries():
        by_site.setdefault(e["site"], set()).add(e["kernel"])
    assert by_site["gemm.qkv"] == {"gemm_dr_kernel<2>"}, by_site["gemm.qkv"]
    assert by_site["gemm.fc1"] == {"gemm_dr_kernel<0>"}, by_site["gemm.fc1"]
    assert not any(k.startswith("gemm_dr") for k in by_site["gemm.proj"] | by_site["gemm.fc2"]), (by_site["gemm.proj"], by_site["gemm.fc2"])


@pytest.mark.parametrize("forms", [{}, {"gemm_cfg": 31}, {"gemm_cfg": 30}, {"attn_cfg": 11}, {"gemm_cfg": 31, "attn_cfg": 11}], ids=lambda f: "+".join("%s%d" % kv for kv in f.items()) or "default")
def test_text_riders_of_a_many_sequence_frame_match_reference_fixture(forms):
    """uvl_debug_set("pair_text", 2): in a frame of >= 2048 visual rows the text branch has no stream and no launches of its own -- its GEMMs ride
    behind the visual tiles (gemm_dr_pair_kernel for QKV / intermediate, gemm_pipe_pair_kernel for the residual GEMMs where the visual problem
    takes cfg 30 / 31), its attention items behind the persistent walk (attn_p64_rider_kernel where the visual attention takes cfg 11), its
    LayerNorm rows in ln_pair_kernel.  The batch-8 fixture's own heuristics pick the 64 x 128 grid for the residual GEMMs and the streaming
    attention kernel, so those pair forms are forced through the tuning hook; every variant is compared with the reference's outputs at the
    gates of the default path, and the profile must show that the pair kernels ran."""
    meta, spec, ref = load_case("b_z256_x256_b8")
    inp = rebuild_inputs(meta, spec)
    eng = _engine(meta, spec)
    eng.debug_set("pair_text", 2)
    try:
        with eng.tuned(**forms):
            got = _run(eng, inp)
            t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).cuda()
            eng.forward(t(inp["template"]), t(inp["search"]), t(inp["ids"]), t(inp["mask"]), t(inp["prompt"]), t(inp["flag"]), profile=True)
            torch.cuda.synchronize()
            by_site = {}
            for e in eng.profile_entries():
                by_site.setdefault(e["site"], set()).add(e["kernel"])
            if not forms:                      # the frame is a single-stream one now: ONE graph, bit-identical to the eager launches
                st, outs = eng.capture(t(inp["template"]), t(inp["search"]), t(inp["ids"]), t(inp["mask"]), t(inp["prompt"]), t(inp["flag"]))
                eng.replay()
                eng.replay()
                torch.cuda.synchronize()
                for k in ("bbox_map", "cls_score_test", "cont_score", "logits", "search", "text"):
                    np.testing.assert_array_equal(outs[k].cpu().numpy(), got[k])
    finally:
        eng.debug_set("pair_text", 1)
    ok, rep = compare_outputs(got, ref, depth=spec.depth)
    assert ok, "\n" + fmt_report(rep)
    if "gemm_cfg" not in forms:
        assert "gemm_dr_pair_kernel<2>" in by_site["gemm.qkv"] and "gemm_dr_pair_kernel<0>" in by_site["gemm.fc1"], by_site
    else:
        # (cfg 31's visual problem requests its residual rows inside the K loop from 12 K tiles on: the third template argument)
        # (third template argument: cfg 31's visual problem requests its residual rows inside the K loop -- proj, K = D, always; fc2, K = 4 D, where the launch is a
        # single round of tiles, which this frame's is: the same code under the symbol with a 2)
        want = ("gemm_pipe_pair_kernel<128,1,1>", "gemm_pipe_pair_kernel<128,1,2>") if forms["gemm_cfg"] == 31 else ("gemm_pipe_pair_kernel<256,1,0>",) * 2
        assert any(k.startswith(want[0]) for k in by_site["gemm.proj"]) and any(k.startswith(want[1]) for k in by_site["gemm.fc2"]), by_site
    if "attn_cfg" in forms:
        assert "attn_p64_rider_kernel" in by_site["attention"], by_site["attention"]
    assert any(k.startswith("ln_pair_kernel") for k in by_site["layernorm"]), by_site["layernorm"]


_oracle_cache = {}


def _oracle_runs(name):
    """(fp32 oracle outputs + per-layer taps, bf16-emulating oracle outputs + taps) of a fixture, computed once per session."""
    from oracle import uvl_oracle as O
    if name not in _oracle_cache:
        _oracle_cache.clear()                                   # taps of one big case at a time
        meta, spec, _ = load_case(name)
        inp = rebuild_inputs(meta, spec)
        sd = rebuild_weights(meta, spec, include_unused=False)
        a = (sd, spec, inp["template"], inp["search"], inp["ids"], inp["mask"], inp["prompt"], inp["flag"])
        t32, temu = {}, {}
        o32 = O.forward_test(*a, t32)
        oemu = O.forward_test(*a, temu, emulate_bf16_mode=True)
        _oracle_cache[name] = (o32, t32, oemu, temu)
    return _oracle_cache[name]


@pytest.mark.parametrize("name", ["tiny_mixed", "tiny_switches", "tiny_allmasked_text", "b_z128_x256", "b_z256_x256_b8", "l_z256_x384"])
def test_error_is_explained_by_bf16_quantisation(name):
    """Separates quantisation from defects.  Three tensors per output: the reference (fp32), the oracle in its bf16-emulating mode
    (a plain numpy forward with the HIP path's roundings at the same places, no HIP involved) and the HIP result.  bf16 rounding
    is chaotic -- two correct bf16 implementations with different summation orders differ from each other by about as much as
    each differs from fp32 -- so the gate is relative: the HIP error against the reference may not exceed 1.5x the emulation's own
    error (+5e-4), and HIP and emulation may not be further apart than twice that error.  A kernel defect of a few 1e-3 on the box
    maps would break the first bound; the absolute gates of parity_util (1e-2) could hide it."""
    meta, spec, ref = load_case(name)
    inp = rebuild_inputs(meta, spec)
    got = _run(_engine(meta, spec), inp)
    _, _, emu, _ = _oracle_runs(name)
    rep, ok = {}, True
    for k in ("bbox_map", "cls_score_test", "cont_score", "logits"):
        e_emu = float(np.abs(emu[k] - ref[k]).max())
        e_hip = float(np.abs(got[k] - ref[k]).max())
        e_he = float(np.abs(got[k] - emu[k]).max())
        slack = 5e-4 if k in ("bbox_map", "cls_score_test") else 5e-3
        rep[k] = "emulation-vs-fp32 %.2e   HIP-vs-fp32 %.2e   HIP-vs-emulation %.2e" % (e_emu, e_hip, e_he)
        ok &= np.isfinite(got[k]).all() and e_hip <= 1.5 * e_emu + slack and e_he <= 2.0 * e_emu + slack
    # the predicted box in tracker terms (reported; see parity_util on why IoU is not a gate on these synthetic heads)
    from tests.parity_util import pred_box_iou
    rep["pred_boxes IoU"] = "emulation-vs-fp32 %.4f   HIP-vs-fp32 %.4f" % (pred_box_iou(emu, ref), pred_box_iou(got, ref))
    print(name, rep)
    assert ok, "\n" + "\n".join("%-16s %s" % kv for kv in rep.items())


@pytest.mark.parametrize("name", ["b_z256_x256_b8", "l_z256_x384"])
def test_layer_localised_error_is_explained_by_bf16_quantisation(name):
    """The same relative gate on the RESIDUAL STREAM of a full-size frame, cut at the first fusion layer and at the last layer
    (uvl_debug_set "stop_layer"): end to end, the emulation error of UVLTrack-B / -L is 6e-3 / 1.3e-2 on the box maps, wide enough
    to hide a defect of a few 1e-3 confined to code only full-size frames run (more than 8 key tiles, 16 heads, the joint rows).
    Per cut: HIP-vs-fp32 on the visual rows and on the text rows may not exceed 1.5x the emulation's own error at that layer
    (+ 1e-3 of the tensor's abs-max).  The fp32 side is the numpy oracle's taps (pinned to the reference within 6e-5)."""
    meta, spec, _ = load_case(name)
    inp = rebuild_inputs(meta, spec)
    eng = _engine(meta, spec)
    _, t32, _, temu = _oracle_runs(name)
    rep, ok = {}, True
    try:
        for k in (spec.n_bert, spec.depth - 1):
            _native_check(eng.lib.uvl_debug_set(eng.handle, b"stop_layer", k))
            o = _run(eng, inp)
            img = np.concatenate([o["vis_token"], o["template"], o["search"]], axis=1)
            for what, hip, f32, emu in (("img", img, t32["img_%d" % k], temu["img_%d" % k]), ("txt", o["text"], t32["txt_%d" % k], temu["txt_%d" % k])):
                e_emu = float(np.abs(emu - f32).max())
                e_hip = float(np.abs(hip - f32).max())
                amax = float(np.abs(f32).max())
                rep["layer %d %s" % (k, what)] = "emulation-vs-fp32 %.3e   HIP-vs-fp32 %.3e   abs-max %.2f" % (e_emu, e_hip, amax)
                ok &= bool(np.isfinite(hip).all()) and e_hip <= 1.5 * e_emu + 1e-3 * amax
    finally:
        eng.lib.uvl_debug_set(eng.handle, b"stop_layer", -1)
    print(name, rep)
    assert ok, "\n" + "\n".join("%-16s %s" % kv for kv in rep.items())


@pytest.mark.parametrize("name", ["tiny_mixed", "b_z256_x256", "l_z256_x384"])
def test_input_side_alone_matches_the_oracle_taps(name):
    """Patch embedding (im2row + GEMM + position table, mae_vit.py:92-100,203-215), the [cls] row and the BERT embedding (bert_backbone.py:260-274) localised:
    uvl_debug_set("stop_layer", -2) runs no layer, so the head's output copies are the residual stream as the input-side kernels left it (prologue_kernel, gemm.patch);
    compared with the oracle's `embed_img` / `embed_txt` taps (fp32, pinned to the reference) at bf16-operand tolerance for the patch GEMM and f32 tolerance for the embedding.
    Batched (setup + bert_embed + im2row kernels) and one sequence (the single prologue launch)."""
    from oracle import uvl_oracle as O
    meta, spec, _ = load_case(name)
    inp = rebuild_inputs(meta, spec)
    sd = rebuild_weights(meta, spec, include_unused=False)
    taps = {}
    O.forward_test(sd, spec, inp["template"], inp["search"], inp["ids"], inp["mask"], inp["prompt"], inp["flag"], taps)
    eng = _engine(meta, spec)
    _native_check(eng.lib.uvl_debug_set(eng.handle, b"stop_layer", -2))
    try:
        for sl in (slice(None), slice(0, 1)):
            o = _run(eng, {k: v[sl] for k, v in inp.items()})
            img = np.concatenate([o["vis_token"], o["template"], o["search"]], axis=1)
            want_img, want_txt = taps["embed_img"][sl], taps["embed_txt"][sl]
            assert img.shape == want_img.shape and o["text"].shape == want_txt.shape
            amax = float(np.abs(want_img).max())
            assert np.isfinite(img).all() and float(np.abs(img - want_img).max()) <= 1e-2 * amax, (float(np.abs(img - want_img).max()), amax)
            np.testing.assert_array_equal(img[:, 0], want_img[:, 0])                      # the [cls] row is a copy
            assert float(np.abs(o["text"] - want_txt).max()) <= 2e-5 * max(1.0, float(np.abs(want_txt).max()))
    finally:
        eng.lib.uvl_debug_set(eng.handle, b"stop_layer", -1)


_fold_emu_cache = {}


def _fold_emulation(name):
    """Outputs of the oracle in its fold-emulating mode (oracle/uvl_oracle.py: the LayerNorm-free frame's rounding points, no HIP involved), whole fixture batch."""
    from oracle import uvl_oracle as O
    if name not in _fold_emu_cache:
        _fold_emu_cache.clear()
        meta, spec, _ = load_case(name)
        inp = rebuild_inputs(meta, spec)
        sd = rebuild_weights(meta, spec, include_unused=False)
        _fold_emu_cache[name] = O.forward_test(sd, spec, inp["template"], inp["search"], inp["ids"], inp["mask"], inp["prompt"], inp["flag"], None, emulate_bf16_mode="fold")
    return _fold_emu_cache[name]


@pytest.mark.parametrize("name", ["tiny_mixed", "tiny_allmasked_text", "b_z256_x256", "l_z256_x384"])
def test_layernorm_free_frame_error_is_explained_by_bf16_quantisation(name):
    """test_error_is_explained_by_bf16_quantisation for the DEFAULT one-sequence frame (round 6: LayerNorm folded into the QKV / fc1 GEMMs, which read the
    un-normalised rows rounded to bf16): sample by sample, the HIP error against the reference's outputs may not exceed 1.5x the error of the oracle's
    fold-emulating mode (the same rounding points in plain numpy) + the slack of that test, and HIP and emulation may not be further apart than twice it."""
    meta, spec, ref = load_case(name)
    inp = rebuild_inputs(meta, spec)
    eng = _engine(meta, spec)
    emu = _fold_emulation(name)
    rep, ok = {}, True
    for b in range(meta["batch"]):
        got = _run(eng, {k: v[b:b + 1] for k, v in inp.items()})
        for k in ("bbox_map", "cls_score_test", "cont_score", "logits"):
            e_emu = float(np.abs(emu[k][b:b + 1] - ref[k][b:b + 1]).max())
            e_hip = float(np.abs(got[k] - ref[k][b:b + 1]).max())
            e_he = float(np.abs(got[k] - emu[k][b:b + 1]).max())
            slack = 5e-4 if k in ("bbox_map", "cls_score_test") else 5e-3
            rep["sample %d %s" % (b, k)] = "emulation-vs-fp32 %.2e   HIP-vs-fp32 %.2e   HIP-vs-emulation %.2e" % (e_emu, e_hip, e_he)
            ok &= bool(np.isfinite(got[k]).all()) and e_hip <= 1.5 * e_emu + slack and e_he <= 2.0 * e_emu + slack
    print(name, rep)
    assert ok, "\n" + "\n".join("%-28s %s" % kv for kv in rep.items())


def _native_check(rc):
    from uvltrack_amd import _native
    _native.check(rc, "uvl_debug_set")


@pytest.mark.parametrize("name", list_cases())
def test_prompter_matches_reference_fixture(name):
    """forward_prompt_init = backbone + prompter against the reference's output for the same box masks."""
    from oracle import uvl_oracle as O
    meta, spec, ref = load_case(name)
    inp = rebuild_inputs(meta, spec)
    eng = _engine(meta, spec)
    tem_mask, ctx_mask = O.box_masks(spec, meta["batch"], seed=meta["input_seed"])
    t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).cuda()
    out = eng.forward(t(inp["template"]), t(inp["search"]), t(inp["ids"]), t(inp["mask"]), torch.zeros_like(t(inp["prompt"])), t(inp["flag"]))
    prompt = eng.forward_prompt(out, t(tem_mask), t(ctx_mask))
    torch.cuda.synchronize()
    got = prompt.cpu().numpy()
    want = ref["prompt_init"]
    assert got.shape == want.shape
    err = np.abs(got - want).max()
    tol = 0.03 * max(1.0, spec.depth / 12.0) * np.abs(want).max()
    assert np.isfinite(got).all() and err <= tol, "prompt err %g > %g (abs-max %g)" % (err, tol, np.abs(want).max())


@pytest.mark.parametrize("name", list_cases())
def test_forward_no_prompt_branch_matches_reference_fixture(name):
    """UVLTrack.forward in eval mode (the grounding call) against the reference's outputs for the same masks:
    inline prompter on the batch-rolled context, two-channel cont_score, then the usual head."""
    from oracle import uvl_oracle as O
    from tests.parity_util import ATOL, softmax_np
    meta, spec, ref = load_case(name)
    inp = rebuild_inputs(meta, spec)
    eng = _engine(meta, spec)
    tem_mask, ctx_mask = O.box_masks(spec, meta["batch"], seed=meta["input_seed"])
    t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).cuda()
    out = eng.forward_full(t(inp["template"]), t(inp["search"]), t(inp["ids"]), t(inp["mask"]), t(tem_mask), t(ctx_mask), t(inp["flag"]))
    torch.cuda.synchronize()
    B = meta["batch"]
    scale = max(1.0, spec.depth / 12.0)
    assert tuple(out["cont_score"].shape) == (B, spec.nx, 2)
    for k in ("cont_score", "bbox_map", "cls_score", "cls_score_test"):
        got, want = out[k].cpu().numpy(), ref["fwd." + k]
        err = np.abs(got - want).max()
        assert np.isfinite(got).all() and err <= ATOL[k] * scale, "%s err %g" % (k, err)
    got, want = out["prompts"].cpu().numpy(), ref["fwd.prompts"]
    assert np.abs(got - want).max() <= 0.03 * scale * np.abs(want).max()
    # pred_boxes, tie-aware: the reference score at our argmax is within the gate of the reference maximum
    score = ref["fwd.cls_score_test"].reshape(B, -1) * softmax_np(ref["fwd.cont_score"])[:, :, 0]
    idx = out["argmax"].cpu().numpy().reshape(-1)
    assert (score.max(-1) - score[np.arange(B), idx]).max() <= 1e-2 * scale
    assert np.abs(out["pred_boxes"].cpu().numpy()[:, 0] - ref["fwd.bbox_map"][np.arange(B), idx]).max() <= 1e-2 * scale


def test_decode_matches_oracle():
    """On-device tracker decode against the numpy restatement of tracker:116-125 on the same forward outputs."""
    from oracle import uvl_oracle as O
    meta, spec, ref = load_case("b_z128_x256")
    inp = rebuild_inputs(meta, spec)
    eng = _engine(meta, spec)
    t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).cuda()
    out = eng.forward(t(inp["template"]), t(inp["search"]), t(inp["ids"]), t(inp["mask"]), t(inp["prompt"]), t(inp["flag"]))
    B = meta["batch"]
    window = O.hann_window(spec.feat_sz)
    state = np.array([[100.0, 80.0, 60.0, 40.0], [5.0, 5.0, 30.0, 30.0], [600.0, 400.0, 80.0, 120.0]], np.float32)[:B]
    resize = np.array([1.3, 0.9, 2.1], np.float32)[:B]
    hw = np.array([[480.0, 640.0], [720.0, 1280.0], [480.0, 640.0]], np.float32)[:B]
    new_state, score, net, idx = eng.decode(out, t(window), t(state), t(resize), t(hw))
    torch.cuda.synchronize()
    got = {k: v.cpu().numpy() for k, v in out.items() if torch.is_tensor(v)}
    e_state, e_score, e_net, e_idx = O.tracker_decode(got["cls_score_test"], got["cont_score"], got["bbox_map"], window, state, resize, hw,
                                                      spec.search_size)
    np.testing.assert_array_equal(idx.cpu().numpy(), e_idx)
    np.testing.assert_allclose(new_state.cpu().numpy(), e_state, atol=2e-3, rtol=0)
    np.testing.assert_allclose(score.cpu().numpy(), e_score, atol=1e-5, rtol=0)
    np.testing.assert_allclose(net.cpu().numpy(), e_net, atol=0, rtol=0)


@pytest.mark.parametrize("name", ["tiny_mixed", "tiny_switches", "b_z256_x256", "l_z128_x384", "l_z256_x384"])
def test_forward_matches_oracle_per_sample_batch1(name):
    """Batch-1 calls (the tracker's shape: single-stream frame, text-branch kernels riding in the visual launches, logits
    riding on the LayerNorm launches) agree with the reference outputs of the batched fixture, sample by sample.  l_z256_x384 is
    BASELINE configs[3] at the north-star template size: its one-sequence launch form is pinned to the reference, not to a self-comparison."""
    meta, spec, ref = load_case(name)
    inp = rebuild_inputs(meta, spec)
    eng = _engine(meta, spec)
    for b in range(meta["batch"]):
        one = {k: v[b:b + 1] for k, v in inp.items()}
        got = _run(eng, one)
        refb = {k: v[b:b + 1] for k, v in ref.items()}
        ok, rep = compare_outputs(got, refb, depth=spec.depth)
        assert ok, "%s sample %d\n%s" % (name, b, fmt_report(rep))


@pytest.mark.parametrize("batch", [None, 1])
def test_graph_replay_equals_eager(batch):
    meta, spec, ref = load_case("tiny_mixed")
    inp = rebuild_inputs(meta, spec)
    if batch is not None:                      # one sequence: single-stream frame, one graph
        inp = {k: v[:batch].copy() for k, v in inp.items()}
    eng = _engine(meta, spec)
    eager = _run(eng, inp)
    t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).cuda()
    st, outs = eng.capture(t(inp["template"]), t(inp["search"]), t(inp["ids"]), t(inp["mask"]), t(inp["prompt"]), t(inp["flag"]))
    for _ in range(3):
        eng.replay()
    torch.cuda.synchronize()
    for k in ("bbox_map", "cls_score_test", "cont_score", "logits", "search", "text"):
        np.testing.assert_array_equal(outs[k].cpu().numpy(), eager[k])
    # new inputs through the static buffers
    st["search"].mul_(0.5)
    eng.replay()
    torch.cuda.synchronize()
    assert np.abs(outs["bbox_map"].cpu().numpy() - eager["bbox_map"]).max() > 1e-4


@pytest.mark.parametrize("name,batch", [("tiny_mixed", None), ("tiny_mixed", 1), ("tiny_switches", 1), ("b_z128_x256", 1), ("b_z256_x256", 1)])
def test_skip_text_is_exact_for_box_outputs(name, batch):
    """BBOX-only mode may drop the text branch: every box output is bit-identical -- also for one sequence,
    where the full frame runs the paired kernels and the text-less frame the plain ones."""
    meta, spec, _ = load_case(name)
    inp = rebuild_inputs(meta, spec)
    if batch is not None:
        inp = {k: v[:batch].copy() for k, v in inp.items()}
    inp["flag"][:] = 0
    eng = _engine(meta, spec)
    full = _run(eng, inp)
    skip = _run(eng, inp, skip_text=True)
    for k in ("bbox_map", "cls_score_test", "cont_score", "logits", "search", "template", "vis_token", "pred_boxes"):
        np.testing.assert_array_equal(full[k], skip[k], err_msg=k)


@pytest.mark.parametrize("name,batch", [("tiny_mixed", None), ("tiny_switches", 1), ("tiny_allmasked_text", None), ("b_z256_x256", 1), ("b_z128_x256", None)])
def test_reused_text_branch_is_bit_identical(name, batch):
    """uvl_inputs.reuse_text: a later frame (other search crop, same sentence) that takes the text branch from the workspace
    equals the frame that recomputes it, bit for bit, on every output; a changed sentence (in place or a new tensor) is noticed."""
    meta, spec, _ = load_case(name)
    inp = rebuild_inputs(meta, spec)
    if batch is not None:
        inp = {k: v[:batch].copy() for k, v in inp.items()}
    eng = _engine(meta, spec)
    t = {k: torch.from_numpy(np.ascontiguousarray(v)).cuda() for k, v in inp.items()}
    g = torch.Generator().manual_seed(7)
    search2 = torch.randn(t["search"].shape, generator=g).cuda()
    keys = ("bbox_map", "cls_score_test", "cont_score", "logits", "search", "template", "text", "vis_token", "txt_token", "pred_boxes")
    run = lambda srch, **kw: {k: v.clone() for k, v in eng.forward(t["template"], srch, t["ids"], t["mask"], t["prompt"], t["flag"], **kw).items() if k in keys}
    want = run(search2)                                       # full frame on the second crop
    run(t["search"])                                          # frame 1 (full): leaves its text rows in the workspace
    assert eng._text_matches(eng._text_key(t["search"].shape[0], t["ids"], t["mask"]))
    for _ in range(3):                                        # frames 2..4 reuse them
        got = run(search2, reuse_text=True)
        for k in keys:
            assert torch.equal(got[k], want[k]), k
    # another sentence written into the same tensor: the version counter changes, the branch is recomputed
    old = t["ids"].clone()
    t["ids"].copy_((old + 3) % spec.vocab)
    assert not eng._text_matches(eng._text_key(t["search"].shape[0], t["ids"], t["mask"]))
    got2 = run(search2, reuse_text=True)
    want2 = run(search2)
    for k in keys:
        assert torch.equal(got2[k], want2[k]), k
    if not name.startswith("tiny_allmasked") and bool((t["flag"] != 0).any()):
        assert not torch.equal(got2["text"], want["text"])    # the sentence did matter


@pytest.mark.parametrize("name,batch", [("tiny_mixed", None), ("b_z256_x256", 1)])
def test_prevalidated_step_reprimes_its_text_branch(name, batch):
    """make_eager_step(reuse_text=True): the step recomputes the text branch on its first call and whenever anything else has run
    on the engine's workspace in between (another sentence, another batch size, the no-prompt forward); otherwise it reuses it.
    Every call must equal the plain forward on the same inputs, bit for bit."""
    meta, spec, _ = load_case(name)
    inp = rebuild_inputs(meta, spec)
    if batch is not None:
        inp = {k: v[:batch].copy() for k, v in inp.items()}
    eng = _engine(meta, spec)
    t = {k: torch.from_numpy(np.ascontiguousarray(v)).cuda() for k, v in inp.items()}
    B = t["search"].shape[0]
    keys = ("bbox_map", "cls_score_test", "cont_score", "logits", "search", "text", "txt_token", "pred_boxes")
    full = lambda: {k: v.clone() for k, v in eng.forward(t["template"], t["search"], t["ids"], t["mask"], t["prompt"], t["flag"]).items() if k in keys}
    want = full()
    step = eng.make_eager_step(t["template"], t["search"], t["ids"], t["mask"], t["prompt"], t["flag"], reuse_text=True)
    same = lambda got: all(torch.equal(got[k], want[k]) for k in keys)
    assert same(step())                                           # first call: full frame
    assert eng._txt_owner is not None
    assert same(step()) and same(step())                          # reused
    other_ids = (t["ids"] + 5) % spec.vocab
    eng.forward(t["template"], t["search"], other_ids, t["mask"], t["prompt"], t["flag"])     # someone else's sentence lands in the workspace
    assert eng._txt_owner is None
    assert same(step()) and same(step())                          # noticed, recomputed, then reused again
    if B > 1:                                                     # another batch size carves the workspace differently
        eng.forward(t["template"][:1], t["search"][:1], t["ids"][:1], t["mask"][:1], t["prompt"][:1], t["flag"][:1])
        assert same(step()) and same(step())
    g = torch.Generator().manual_seed(3)
    t["search"].copy_(torch.randn(t["search"].shape, generator=g).cuda())     # next frame: the buffers the step is bound to are rewritten
    want = full()
    assert same(step()) and same(step())


def test_cpu_tensors_fail_loudly():
    from uvltrack_amd import _native
    meta, spec, _ = load_case("tiny_mixed")
    inp = rebuild_inputs(meta, spec)
    eng = _engine(meta, spec)
    t = lambda a: torch.from_numpy(np.ascontiguousarray(a))
    with pytest.raises(_native.NativeLibraryError):
        eng.forward(t(inp["template"]), t(inp["search"]), t(inp["ids"]), t(inp["mask"]), t(inp["prompt"]), t(inp["flag"]))


@pytest.mark.parametrize("name,batch", [("tiny_mixed", None), ("b_z256_x256", 1), ("b_z128_x256", None)])
def test_repeated_frames_are_bit_identical(name, batch):
    """No race in the frame: 40 repeats of the same frame (single-stream paired schedule for one sequence, two streams with
    events for several) give bit-identical outputs -- split-K slabs are folded in a fixed order, no atomics anywhere."""
    meta, spec, _ = load_case(name)
    inp = rebuild_inputs(meta, spec)
    if batch is not None:
        inp = {k: v[:batch] for k, v in inp.items()}
    eng = _engine(meta, spec)
    t = {k: torch.from_numpy(np.ascontiguousarray(v)).cuda() for k, v in inp.items()}
    keys = ("bbox_map", "cls_score_test", "cont_score", "logits", "search", "text", "pred_boxes")
    ref = None
    for _ in range(40):
        out = eng.forward(t["template"], t["search"], t["ids"], t["mask"], t["prompt"], t["flag"])
        cur = {k: out[k].clone() for k in keys}
        if ref is None:
            ref = cur
            continue
        for k in keys:
            assert torch.equal(cur[k], ref[k]), k


@pytest.mark.parametrize("name", ["b_z256_x256", "l_z256_x384"])
def test_off_default_forms_of_the_one_sequence_frame_change_no_bit_or_only_the_head(name):
    """The tools' switches of round 6 on a one-sequence frame: "rider_pf" (the riders request the next rider's weight, XCD-matched: requests only -- every output bit
    for bit) and "head_fin" 0 (tower layer 3 + head_tail as two launches: the backbone outputs bit for bit, the head maps within the accumulation-order noise
    of the last layer).  Restores the defaults."""
    meta, spec, _ = load_case(name)
    inp = {k: v[:1] for k, v in rebuild_inputs(meta, spec).items()}
    eng = _engine(meta, spec)
    t = {k: torch.from_numpy(np.ascontiguousarray(v)).cuda() for k, v in inp.items()}
    keys = ("bbox_map", "cls_score_test", "cont_score", "logits", "search", "text", "pred_boxes")
    run = lambda: {k: v.clone() for k, v in eng.forward(t["template"], t["search"], t["ids"], t["mask"], t["prompt"], t["flag"]).items() if k in keys}
    ref = run()
    try:
        eng.debug_set("rider_pf", 64)
        got = run()
        for k in keys:
            assert torch.equal(got[k], ref[k]), k
        eng.debug_set("rider_pf", 0)
        eng.debug_set("head_fin", 0)
        got = run()
        for k in ("cont_score", "logits", "search", "text"):
            assert torch.equal(got[k], ref[k]), k
        for k in ("bbox_map", "cls_score_test"):
            assert float((got[k] - ref[k]).abs().max()) < 2e-3, k
    finally:
        eng.debug_set("rider_pf", 0)
        eng.debug_set("head_fin", 1)


@pytest.mark.parametrize("name", ["b_z256_x256", "l_z256_x384", "tiny_mixed"])
def test_fused_layernorm_gemm_launch_is_bit_identical(name):
    """uvl_debug_set("fuse_ln", 1): one-sequence frames run LayerNorm + the GEMM that consumes it (LN-1 -> QKV, LN-2 -> fc1, with the
    text branch's riders) as ONE launch -- LayerNorm rows, a hierarchical grid barrier with one L2 invalidate per XCD, then the GEMM tiles.
    Same device functions, so every output must equal the two-launch frame bit for bit, frame after frame (the barrier's counters
    are monotonic over launches of different grid sizes).  Off by default: it measures 3-4 % slower (profiles/r03_summary.md).
    Both are forms of the LayerNorm-KERNEL schedule of rounds 1-5 ("fold_ln" 0); the default one-sequence frame has no LayerNorm launches since round 6."""
    meta, spec, _ = load_case(name)
    inp = {k: v[:1] for k, v in rebuild_inputs(meta, spec).items()}
    eng = _engine(meta, spec)
    t = {k: torch.from_numpy(np.ascontiguousarray(v)).cuda() for k, v in inp.items()}
    keys = ("bbox_map", "cls_score_test", "cont_score", "logits", "search", "text", "pred_boxes")
    eng.debug_set("fold_ln", 0)
    try:
        ref = {k: v.clone() for k, v in eng.forward(t["template"], t["search"], t["ids"], t["mask"], t["prompt"], t["flag"]).items() if k in keys}
        eng.debug_set("fuse_ln", 1)
        for _ in range(25):
            out = eng.forward(t["template"], t["search"], t["ids"], t["mask"], t["prompt"], t["flag"])
            for k in keys:
                assert torch.equal(out[k], ref[k]), k
    finally:
        eng.debug_set("fuse_ln", 0)
        eng.debug_set("fold_ln", 1)


@pytest.mark.parametrize("name", ["tiny_mixed", "b_z256_x256", "l_z256_x384"])
def test_layernorm_free_frame_agrees_with_the_layernorm_kernel_schedule(name):
    """The default one-sequence frame (round 6: no LayerNorm launch, no split-K slab -- residual GEMMs that finish x in the launch, LayerNorm folded into the
    QKV / fc1 weights, logits riding on the QKV launches, text rows joined by one small kernel) against the LayerNorm-kernel schedule of rounds 1-5
    (uvl_debug_set "fold_ln" 0) on the same inputs: two bf16 paths with different rounding points, so they agree to the level each agrees with the fp32
    reference (the fixture gates, which test_forward_matches_oracle_per_sample_batch1 applies to the default frame), not bit for bit; the launch count says
    which schedule ran (reference: block.py:29-32, bert_backbone.py:335-339,376-380)."""
    meta, spec, ref = load_case(name)
    inp = {k: v[:1] for k, v in rebuild_inputs(meta, spec).items()}
    eng = _engine(meta, spec)
    t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).cuda()
    args = (t(inp["template"]), t(inp["search"]), t(inp["ids"]), t(inp["mask"]), t(inp["prompt"]), t(inp["flag"]))

    def run():
        out = {k: v.cpu().numpy() for k, v in eng.forward(*args).items() if torch.is_tensor(v)}
        eng.forward(*args, profile=True)
        torch.cuda.synchronize()
        prof = eng.profile_entries()
        return out, sum(e["launches"] for e in prof), {e["kernel"] for e in prof}

    new, n_new, k_new = run()
    eng.debug_set("fold_ln", 0)
    try:
        old, n_old, k_old = run()
    finally:
        eng.debug_set("fold_ln", 1)
    # 7 -> 5 launches per ViT block, one LayerNorm-like launch left (the text join)
    assert n_new == n_old - 2 * spec.depth, (n_new, n_old)
    assert any(k.startswith("gemm_fin") for k in k_new) and any(k.startswith("gemm_lnf") for k in k_new) and not any(k.startswith("ln_") for k in k_new), k_new
    assert any(k.startswith("ln_") for k in k_old) and not any(k.startswith("gemm_fin") for k in k_old), k_old
    refb = {k: v[:1] for k, v in ref.items()}
    for got in (new, old):
        ok, rep = compare_outputs(got, refb, depth=spec.depth)
        assert ok, "\n" + fmt_report(rep)


@pytest.mark.parametrize("name,batch", [("b_z128_x256", 16), ("b_z256_x256", 8), ("l_z256_x384", 8), ("b_z256_x256", 32)])
def test_large_batch_matches_single_sequence_runs(name, batch):
    """The batched regime takes other kernels than the fixtures' 2-3 samples (grouped tile order, 64x128 / 128x128 tiles,
    128x128 implicit-GEMM conv tiles, 128-query attention workgroups with 2 or 3 ring stages): a batch of 8-32 sequences with
    mixed flags must reproduce the one-sequence runs (themselves checked against the reference) sample by sample.  The 8-sequence
    UVLTrack-L frame and the 32-sequence UVLTrack-B frame take the text riders by default (uvl_api.hip::text_rides): the first with one
    attention item per workgroup (the rider's items go to the workgroups of the short last query block), the second with 1152 items on
    512 workgroups (rider items round-robin from the end of the grid) and the residual GEMMs on gemm_pipe_pair_kernel<256,1>."""
    from uvltrack_amd import weightgen as wg
    from uvltrack_amd.engine import HipEngine
    meta, spec, _ = load_case(name)
    _engines.clear()
    eng = HipEngine(spec, torch.device("cuda:0"), max_batch=batch)
    eng.load_state_dict(rebuild_weights(meta, spec, include_unused=True))
    inp = wg.make_inputs(spec, batch=batch, seed=77, flags=[(i * 7) % 3 for i in range(batch)])
    inp["mask"][1, :] = False                      # one all-padding text
    t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).cuda()
    big = eng.forward(t(inp["template"]), t(inp["search"]), t(inp["ids"]), t(inp["mask"]), t(inp["prompt"]), t(inp["flag"]))
    big = {k: v.cpu().numpy() for k, v in big.items() if torch.is_tensor(v)}
    # which text placement the frame took BY DEFAULT: riders for UVLTrack-L from 5000 visual rows and for any model from 16000, else the second stream
    eng.forward(t(inp["template"]), t(inp["search"]), t(inp["ids"]), t(inp["mask"]), t(inp["prompt"]), t(inp["flag"]), profile=True)
    torch.cuda.synchronize()
    kernels = {e["kernel"] for e in eng.profile_entries()}
    rows = batch * (1 + (spec.template_size // 16) ** 2 + (spec.search_size // 16) ** 2)
    want_riders = rows >= 16000 or (spec.dim >= 1024 and rows >= 5000)
    has_riders = any(k.startswith(("gemm_dr_pair_kernel", "gemm_pipe_pair_kernel", "attn_p64_rider_kernel")) for k in kernels)
    assert has_riders == want_riders, (rows, sorted(kernels))
    if want_riders:
        assert {"gemm_dr_pair_kernel<2>", "gemm_dr_pair_kernel<0>", "attn_p64_rider_kernel"} <= kernels, sorted(kernels)
        assert any(k.startswith("gemm_pipe_pair_kernel") for k in kernels), sorted(kernels)
    scale = max(1.0, spec.depth / 12.0)
    # the one-sequence runs on the LayerNorm-KERNEL schedule ("fold_ln" 0): the same precision plan as the batched frame, so this stays a check of the batching
    # (the default one-sequence frame of round 6 rounds at other places: two bf16 plans differ by about what each differs from fp32, and both are pinned to the
    # reference by their own fixtures -- b_z256_x256_b32 for this path)
    eng.debug_set("fold_ln", 0)
    for b in range(batch):
        one = eng.forward(*[t(inp[k][b:b + 1]) for k in ("template", "search", "ids", "mask", "prompt", "flag")])
        one = {k: v.cpu().numpy() for k, v in one.items() if torch.is_tensor(v)}
        for k, tol in (("bbox_map", 1e-2), ("cls_score_test", 1e-2), ("cont_score", 5e-2), ("logits", 0.15)):
            err = np.abs(big[k][b:b + 1] - one[k]).max()
            assert np.isfinite(big[k]).all() and err <= tol * scale, "%s sample %d: %g" % (k, b, err)
        for k in ("search", "template", "text"):
            err = np.abs(big[k][b:b + 1] - one[k]).max()
            assert err <= 0.03 * scale * np.abs(one[k]).max(), "%s sample %d: %g" % (k, b, err)
    eng.close()
