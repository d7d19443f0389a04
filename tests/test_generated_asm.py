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
