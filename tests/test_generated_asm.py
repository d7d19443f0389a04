"""The hand-scheduled attention kernel's assembly (uvltrack_amd/csrc/attn_p64_asm.inc) is a generated file: it must be what
tools/gen/attn_p64_gen.py produces today, and the generator's hazard lint (wait states behind MFMA results that the assembler does not
insert) must find nothing in it.  CPU-only: no hipcc, no GPU."""
import importlib.util
import os

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _gen():
    spec = importlib.util.spec_from_file_location("attn_p64_gen", os.path.join(ROOT, "tools", "gen", "attn_p64_gen.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def test_committed_asm_is_the_generators_output():
    g = _gen()
    lines = [ln for ln in g.generate() if not ln.startswith(";")]
    committed = [ln for ln in open(os.path.join(ROOT, "uvltrack_amd", "csrc", "attn_p64_asm.inc")).read().splitlines() if ln.startswith('"')]
    assert len(committed) == len(lines)
    for want, got in zip(lines, committed):
        assert got == '"%s\\n\\t"' % want


def test_hazard_lint_is_clean_and_catches_a_planted_hazard():
    g = _gen()
    lines = g.generate()
    assert g.lint(lines) == []
    # an exponential that reads a score register straight behind the MFMA that writes it must be reported
    bad = ["v_mfma_f32_32x32x16_bf16 v[96:111], v[144:147], v[64:67], 0", "v_exp_f32 v96, v96"]
    assert g.lint(bad)
    # and a pack feeding the next MFMA's B operand with no instruction in between
    bad2 = ["v_cvt_pk_bf16_f32 v128, v96, v97", "v_mfma_f32_32x32x16_bf16 v[0:15], v[176:179], v[128:131], v[0:15]"]
    assert g.lint(bad2)


def test_register_map_stays_inside_the_clobber_list():
    g = _gen()
    import re
    used_v, used_s = set(), set()
    for ln in g.generate(trace=True) + g.generate():
        for a, b in re.findall(r"v\[(\d+):(\d+)\]", ln):
            used_v.update(range(int(a), int(b) + 1))
        used_v.update(int(x) for x in re.findall(r"\bv(\d+)\b", ln))
        for a, b in re.findall(r"s\[(\d+):(\d+)\]", ln):
            used_s.update(range(int(a), int(b) + 1))
        used_s.update(int(x) for x in re.findall(r"\bs(\d+)\b", ln))
    assert max(used_v) <= 255
    assert min(used_s) >= 38 and max(used_s) <= 101          # attention.hip: ATTN_P64_SGPRS = s38..s101


# ---- the K loop of gemm_dr_kernel (uvltrack_amd/csrc/gemm_dr_asm.inc, tools/gen/gemm_dr_gen.py)
def _gen_dr():
    spec = importlib.util.spec_from_file_location("gemm_dr_gen", os.path.join(ROOT, "tools", "gen", "gemm_dr_gen.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def test_committed_gemm_loop_is_the_generators_output():
    g = _gen_dr()
    assert open(os.path.join(ROOT, "uvltrack_amd", "csrc", "gemm_dr_asm.inc")).read() == g.render()


def test_gemm_loop_register_map_and_counts():
    """Every register the block names is in gemm_dr.hip's clobber list (v0..v113, s40..s59) or an accumulator operand (a0..a127, each
    written by exactly the MFMAs of its block); per K tile and wave: 64 MFMAs, 16 fragment reads, 8 W loads, 4 LDS-DMA instructions, one
    barrier; the wave index is NOT read back from a VALU-written SGPR; an M0 write and the LDS-DMA that uses it are never adjacent; the loop's
    vector-memory waits leave exactly the four youngest LDS-DMA pieces in flight."""
    import re
    g = _gen_dr()
    lines = g.Gen().generate()
    used_v, used_s, used_a = set(), set(), set()
    for ln in lines:
        for a, b in re.findall(r"\bv\[(\d+):(\d+)\]", ln):
            used_v.update(range(int(a), int(b) + 1))
        used_v.update(int(x) for x in re.findall(r"\bv(\d+)\b", ln))
        for a, b in re.findall(r"\bs\[(\d+):(\d+)\]", ln):
            used_s.update(range(int(a), int(b) + 1))
        used_s.update(int(x) for x in re.findall(r"\bs(\d+)\b", ln))
        for a, b in re.findall(r"\ba\[(\d+):(\d+)\]", ln):
            used_a.update(range(int(a), int(b) + 1))
    assert max(used_v) < g.NV == 114
    assert min(used_s) >= 40 and max(used_s) <= 59
    assert used_a == set(range(128))
    src = open(os.path.join(ROOT, "uvltrack_amd", "csrc", "gemm_dr.hip")).read()
    assert '"v113"' in src and '"v114"' not in src                      # the clobber list ends where the register map ends
    top = lines.index("top_%=:")
    loop = lines[top:]
    assert sum("v_mfma_f32_16x16x32_bf16" in ln for ln in loop) == 4 * 64
    assert sum(ln.startswith("ds_read_b128") for ln in loop) == 4 * 16
    assert sum(ln.startswith("global_load_dwordx4") for ln in loop) == 4 * 8
    assert sum(ln.startswith("global_load_lds_dwordx4") for ln in loop) == 4 * 4
    assert sum(ln == "s_barrier" for ln in loop) == 4
    assert [ln for ln in loop if ln.startswith("s_waitcnt vmcnt")][:4] == ["s_waitcnt vmcnt(4)"] * 4
    assert not any("v_readfirstlane" in ln for ln in lines)
    for a, b in zip(lines, lines[1:]):
        assert not (a.startswith("s_add_u32 m0") and b.startswith("global_load_lds")), (a, b)
