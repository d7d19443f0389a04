"""Generator of the phase-alternating ("ping-pong") fused-attention item walk: attn_pp_kernel (uvltrack_amd/csrc/attention.hip).

    python tools/gen/attn_pp_gen.py [--lint]          # rewrites uvltrack_amd/csrc/attn_pp_asm.inc

Why.  The whole-chip timeline of attn_p64_kernel (tools/attn_wgtrace.py, profiles/r03_attention.md) shows that two co-resident waves
that both interleave MFMAs and softmax VALU do not share a SIMD: the older one runs at ~70 cycles per MFMA, the younger at ~210, the
matrix pipe is ~60 % busy.  At head_dim 64 a wave needs the VALU port about as long as the matrix pipe (2 exp2 + 2 adds + 1 pack per
MFMA), so the two resources only fill when one wave of the SIMD is in an MFMA-only phase while the other is in a VALU-only phase.

Structure.  One workgroup = 8 waves = two HALVES of four waves; waves w and w + 4 share a SIMD.  Each half walks its own 256-query
items (64 queries per wave) through its own DMA ring -- the halves share nothing but the barriers.  A wave alternates
    M(t, jb): 16 MFMAs -- scores of key block jb of tile t for both query blocks (8) + P V of the previous key block (8)
    V(t, jb): the softmax of those scores (80 VALU) + the fragment reads of the next M (8 ds_read_b128)
with an s_barrier between phases; half 1 runs one barrier behind half 0, so on every SIMD one wave is in M while the other is in V.
A wave's own stream is strictly serial (no intra-wave interleave to schedule), only 32 score + 16 P + 32 fragment registers are live.

Same math, LDS tile image and operand trick as attn_w64_kernel / attn_p64_kernel (reference: lib/models/backbones/block.py:47-61).
"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import attn_p64_gen as g1  # noqa: E402  (shared helpers: lint, swizzle conventions)

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
OUT = os.path.join(ROOT, "uvltrack_amd", "csrc", "attn_pp_asm.inc")

STAGE = 16384
NS = 4
KADD0 = NS * STAGE                  # per half: [stage][wave][64] f32 key_add rows
FLAG0 = KADD0 + NS * 4 * 256        # per half: 4 x int
HALF = FLAG0 + 64                   # bytes of LDS per half
LDS_BYTES = 2 * HALF

# ---------------------------------------------------------------- VGPR map
def O(x, db, r): return 32 * x + 16 * db + r
def Q(x, kk): return 64 + 16 * x + 4 * kk
def S(x, r): return 96 + 16 * x + r
def P(x, t2): return 128 + 8 * x + 4 * t2
def KF(k): return 144 + 4 * k
def VF(f): return 160 + 4 * f                      # f = 2 * t2 + db
def KA(r): return 176 + r
def PS(x, i): return 192 + 2 * x + i
def KOFF(kk): return 196 + kk
def VOFF(jb, t2): return 200 + 2 * jb + t2
def DK(i): return 204 + i
def DV(i): return 206 + i
V_DKA, V_KAREAD, V_KAADDR, V_LIM, V_NEGINF, V_KACUR, V_HALF8 = 208, 209, 210, 211, 212, 213, 214
def QOFF(x): return 216 + x
T = [218, 219, 220, 221, 222, 223, 224]            # T[2..5] = v220..v223, an aligned quad
BUF = 96                                           # epilogue: packed rows in the dead score / P / fragment registers
NV = 226                                           # registers v0..v225

SG = dict(q=40, k=42, vt=44, ka=46, o=48, N=50, Npad=51, H=52, total=53, cnt=54, idx=55, istr=56, kas=58, wave=59, lds=60,
          nqb=61, mq=62, mh=63, nt=64, ntail=65, tailf=66, qb=67, h=68, b=69, Kp=70, Vp=72, Ap=74, Qp=76, Op=78, t=80, masked=81,
          mnext=82, active=83, r=84, it=85, bad=86, x0=88, x1=89, x2=90, x3=91, x4=92, x5=93, wl=94, wa=95, ibad=96, q0=97,
          have=98, half=99, xcd=57, trace=39, pmask=38)


def s(n): return "s%d" % SG[n]
def s2(n): return "s[%d:%d]" % (SG[n], SG[n] + 1)
def shi(n): return "s%d" % (SG[n] + 1)
def v(n): return "v%d" % n
def vr(n, c): return "v[%d:%d]" % (n, n + c - 1)


OPT = dict(prio=1, dma_in_m=1)
# timing-only ablations for tools/attn_pp_trace.py (WRONG results): PP_ABL=noexp,novalu,nodma,nolds,nobar,prio0,dmav
ABL = set(x for x in os.environ.get("PP_ABL", "").split(",") if x)
if "prio0" in ABL:
    OPT["prio"] = 0
if "dmav" in ABL:
    OPT["dma_in_m"] = 0


def mfma(dst, a, b, c):
    return "v_mfma_f32_32x32x16_bf16 %s, %s, %s, %s" % (vr(dst, 16), vr(a, 4), vr(b, 4), "0" if c is None else vr(c, 16))


def dma_round(stage):
    out = []
    for i in range(2):
        out.append(["s_add_u32 m0, %s, %d" % (s("wl"), stage * STAGE + i * 4096), "s_nop 0", "global_load_lds_dwordx4 %s, %s" % (v(DK(i)), s2("Kp"))])
    for i in range(2):
        out.append(["s_add_u32 m0, %s, %d" % (s("wl"), stage * STAGE + 8192 + i * 4096), "s_nop 0", "global_load_lds_dwordx4 %s, %s" % (v(DV(i)), s2("Vp"))])
    out.append(["s_add_u32 m0, %s, %d" % (s("wa"), stage * 1024), "s_nop 0", "global_load_lds_dword %s, %s" % (v(V_DKA), s2("Ap"))])
    return out


def advance_round():
    return ["s_add_u32 %s, %s, 1" % (s("r"), s("r")),
            "s_cmp_lt_u32 %s, %s" % (s("r"), s("nt")),
            "s_cselect_b32 %s, 8192, 0" % s("x0"),
            "s_add_u32 %s, %s, %s" % (s("Kp"), s("Kp"), s("x0")),
            "s_addc_u32 %s, %s, 0" % (shi("Kp"), shi("Kp")),
            "s_lshr_b32 %s, %s, 6" % (s("x1"), s("x0")),
            "s_add_u32 %s, %s, %s" % (s("Vp"), s("Vp"), s("x1")),
            "s_addc_u32 %s, %s, 0" % (shi("Vp"), shi("Vp")),
            "s_lshr_b32 %s, %s, 5" % (s("x1"), s("x0")),
            "s_add_u32 %s, %s, %s" % (s("Ap"), s("Ap"), s("x1")),
            "s_addc_u32 %s, %s, 0" % (shi("Ap"), shi("Ap"))]


def dma_guarded(a, pairs, tag):
    """a half without an item (odd item count) keeps the barrier skeleton and issues nothing"""
    if "nodma" in ABL and not tag.startswith("L"):
        return
    skip = "nod%s" % tag
    a.e("s_cmp_eq_u32 %s, 0" % s("have"))
    a.e("s_cbranch_scc1 %s" % a.ref(skip))
    for pair in pairs:
        for t in pair:
            a.e(t)
    a.label(skip)


def tail_fix(a, stage, tag):
    """zero the V^T columns of keys >= N in this wave's own two pieces of the tail tile (see attn_p64_gen.tail_fix)"""
    skip = "tfx%s" % tag
    a.e("s_add_u32 %s, %s, 2" % (s("x0"), s("t")))
    a.e("s_cmp_eq_u32 %s, %s" % (s("x0"), s("nt")))
    a.e("s_cselect_b32 %s, %s, 0" % (s("x0"), s("tailf")))
    a.e("s_and_b32 %s, %s, %s" % (s("x0"), s("x0"), s("have")))
    a.e("s_cmp_eq_u32 %s, 0" % s("x0"))
    a.e("s_cbranch_scc1 %s" % a.ref(skip))
    t0, t1, m1, m2 = T[0], T[1], T[6], V_KACUR
    d = T[2]
    a.e("v_mbcnt_lo_u32_b32 %s, -1, 0" % v(t1))
    a.e("v_mbcnt_hi_u32_b32 %s, -1, %s" % (v(t1), v(t1)))
    a.e("v_and_b32 %s, 7, %s" % (v(t0), v(t1)))
    a.e("v_lshrrev_b32 %s, 4, %s" % (v(m1), v(t1)))
    a.e("s_and_b32 %s, %s, 1" % (s("x0"), s("wave")))
    a.e("s_lshl_b32 %s, %s, 2" % (s("x0"), s("x0")))
    a.e("v_add_u32 %s, %s, %s" % (v(m1), s("x0"), v(m1)))
    a.e("v_xor_b32 %s, %s, %s" % (v(t0), v(t0), v(m1)))
    a.e("v_lshlrev_b32 %s, 3, %s" % (v(t0), v(t0)))
    a.e("v_sub_u32 %s, %s, %s" % (v(t0), s("ntail"), v(t0)))
    a.e("v_lshlrev_b32 %s, 4, %s" % (v(t1), v(t1)))
    a.e("v_add_u32 %s, %s, %s" % (v(t1), s("wl"), v(t1)))
    for i in range(2):
        off = stage * STAGE + 8192 + i * 4096
        a.e("ds_read_b128 %s, %s offset:%d" % (vr(d, 4), v(t1), off))
        a.e("s_waitcnt lgkmcnt(0)")
        for e in range(4):
            a.e("v_cmp_lt_i32 vcc, %d, %s" % (2 * e, v(t0)))
            a.e("v_cndmask_b32_e64 %s, 0, -1, vcc" % v(m1))
            a.e("v_cmp_lt_i32 vcc, %d, %s" % (2 * e + 1, v(t0)))
            a.e("v_cndmask_b32_e64 %s, 0, -1, vcc" % v(m2))
            a.e("v_and_b32 %s, 0xffff, %s" % (v(m1), v(m1)))
            a.e("v_and_b32 %s, 0xffff0000, %s" % (v(m2), v(m2)))
            a.e("v_or_b32 %s, %s, %s" % (v(m1), v(m1), v(m2)))
            a.e("v_and_b32 %s, %s, %s" % (v(d + e), v(d + e), v(m1)))
        a.e("ds_write_b128 %s, %s offset:%d" % (v(t1), vr(d, 4), off))
    a.e("s_waitcnt lgkmcnt(0)")
    a.label(skip)


def ka_load(stage, jb):
    out = []
    for gq in range(4):
        off = stage * 1024 + jb * 128 + (gq >> 1) * 64 + (gq & 1) * 16
        out.append("ds_read_b128 %s, %s offset:%d" % (vr(KA(4 * gq), 4), v(V_KAADDR), off))
    return out


def ka_scale():
    return ["v_mul_f32 %s, 0x3fb8aa3b, %s" % (v(KA(r)), v(KA(r))) for r in range(16)]


def mask_apply(x, jb):
    out = []
    for r in range(16):
        key = 32 * jb + 16 * (r >> 3) + 4 * ((r >> 2) & 1) + (r & 3)
        out.append("v_add_f32 %s, %s, %s" % (v(S(x, r)), v(S(x, r)), v(KA(r))))
        out.append("v_cmp_lt_i32 vcc, %d, %s" % (key, v(V_LIM)))
        out.append("v_cndmask_b32 %s, %s, %s, vcc" % (v(S(x, r)), v(V_NEGINF), v(S(x, r))))
    return out


def softmax_gaps(x):
    """The 40 VALU of one query block's softmax in eight groups: exp2 pairs lead, their row-sum adds and the pack trail one group,
    so nothing waits for the transcendental pipe"""
    ex = lambda r: "v_exp_f32 %s, %s" % (v(S(x, r)), v(S(x, r)))
    ad = lambda r: "v_add_f32 %s, %s, %s" % (v(PS(x, r & 1)), v(PS(x, r & 1)), v(S(x, r)))
    cv = lambda i: "v_cvt_pk_bf16_f32 %s, %s, %s" % (v(P(x, i >> 2) + (i & 3)), v(S(x, 2 * i)), v(S(x, 2 * i + 1)))
    gaps = [[] for _ in range(8)]
    for i in range(8):
        gaps[i] += [ex(2 * i), ex(2 * i + 1)]
        if i >= 1:
            gaps[i] += [ad(2 * i - 2), ad(2 * i - 1), cv(i - 1)]
    gaps[7] += [ad(14), ad(15), cv(7)]
    if "novalu" in ABL:
        return [[] for _ in range(8)]
    if "noexp" in ABL:
        gaps = [[t.replace("v_exp_f32", "v_mov_b32") for t in grp] for grp in gaps]
    return gaps


TRACE = False
SG["tr"] = 100


def stamp(a, k):
    """trace build (tools/attn_pp_trace.py): waves 0 and 4 of workgroup 0 store the shader clock at point k of the current tile of
    their first item: record [half][tile][k], 8 bytes each"""
    if not TRACE:
        return
    skip = "st%d" % a.nlabel
    a.nlabel += 1
    a.e("s_cmp_eq_u32 %s, 0" % s("trace"))
    a.e("s_cbranch_scc1 %s" % a.ref(skip))
    a.e("s_memtime %s" % s2("x2"))
    a.e("s_lshl_b32 %s, %s, 9" % (s("x4"), s("half")))
    a.e("s_lshl_b32 %s, %s, 3" % (s("x5"), s("t")))
    a.e("s_add_u32 %s, %s, %s" % (s("x4"), s("x4"), s("x5")))
    a.e("s_add_u32 %s, %s, %d" % (s("x4"), s("x4"), k))
    a.e("s_lshl_b32 %s, %s, 3" % (s("x4"), s("x4")))
    a.e("s_waitcnt lgkmcnt(0)")
    a.e("s_store_dwordx2 %s, s[100:101], %s" % (s2("x2"), s("x4")))
    a.e("s_waitcnt lgkmcnt(0)")
    a.label(skip)


def barrier(a):
    a.e("s_barrier")


def phase_M(a, st, jb, masked, qk=True, body=True, tag=""):
    """The M phase of key block jb of tile t: for query block A  4 score MFMAs + 4 P V MFMAs (of the previous key block), with the
    softmax of query block B's PREVIOUS scores in their gaps (five VALU per gap: this wave waits for the matrix pipe there anyway);
    then the same 8 MFMAs for block B, bare -- pointer arithmetic and LDS-DMA issue ride in those gaps."""
    stamp(a, 0 if jb == 0 else 4)
    if OPT["prio"]:
        a.e("s_setprio 1")
    a.e("s_waitcnt lgkmcnt(0)")
    mf = []
    for x in range(2):
        for k in range(4):
            if qk:
                mf.append(mfma(S(x, 0), KF(k), Q(x, k), None if k == 0 else S(x, 0)))
            mf.append(mfma(O(x, k & 1, 0), VF(k), P(x, k >> 1), O(x, k & 1, 0)))            # fragment k = 2 t2 + db
    gaps = softmax_gaps(1)
    nA = 8 if qk else 4                             # drain: P V only, block B's softmax in the four gaps of block A's MFMAs
    fill = {i: [] for i in range(len(mf))}
    for i in range(8):
        fill[i * nA // 8] += gaps[i]
    pre = []
    if body and jb == 0:
        # block B's pending scores belong to tile t - 1: its mask term (if that tile carried one) is still in KA / V_LIM
        skip = "pm%d%s" % (st, "m" if masked else "p")
        pre.append("s_cmp_eq_u32 %s, 0" % s("pmask"))
        pre.append("s_cbranch_scc1 %s" % a.ref(skip))
        pre += mask_apply(1, 1)
        pre.append("LABEL " + skip)
    elif body and masked:
        pre += mask_apply(1, 0)
    late = {i: [] for i in range(len(mf))}
    rd = dma_round((st + 3) & 3)
    if body:
        if OPT["dma_in_m"]:
            if jb == 0:
                late[9] += ["DMA0"]
                late[11] += ["DMA1"]
                late[13] += ["DMA4"]
            else:
                late[9] += ["DMA2"]
                late[11] += ["DMA3"]
        if jb == 1:
            adv = advance_round()
            late[8] += adv[0:3]
            late[10] += adv[3:5]
            late[12] += adv[5:8]
            late[14] += adv[8:11]
        if masked:
            # this key block's mask term for the V phase that follows (and for block B in the next M phase)
            if jb == 0:
                late[8] += ["s_add_u32 %s, %s, 1" % (s("x0"), s("t")), "s_cmp_eq_u32 %s, %s" % (s("x0"), s("nt")),
                            "s_cselect_b32 %s, %s, 0" % (s("x0"), s("tailf")), "s_cmp_lg_u32 %s, 0" % s("x0"),
                            "s_cselect_b32 %s, %s, 64" % (s("x0"), s("ntail")), "v_sub_u32 %s, %s, %s" % (v(V_LIM), s("x0"), v(V_HALF8))]
            late[8] += ka_load(st, jb)
            late[13] += ["s_waitcnt lgkmcnt(0)"] + ka_scale()[:8]
            late[15] += ka_scale()[8:]
    for i, m in enumerate(mf):
        a.e(m)
        if i == 0:
            for t in pre:
                if t.startswith("LABEL "):
                    a.label(t[6:])
                else:
                    a.e(t)
        for t in fill[i]:
            a.e(t)
        for t in late[i]:
            if t.startswith("DMA"):
                dma_guarded(a, [rd[int(t[3:])]], "%s%d%d%d%s" % (tag, st, jb, i, "m" if masked else "p"))
            else:
                a.e(t)
    if body and jb == 1:
        a.e("s_waitcnt vmcnt(10)")                     # round t + 1 has landed (own pieces); rounds t + 2, t + 3 fly
        tail_fix(a, (st + 1) & 3, "%s%d%s" % (tag, st, "m" if masked else "p"))
    if OPT["prio"]:
        a.e("s_setprio 0")
    stamp(a, 1 if jb == 0 else 5)
    barrier(a)


def phase_V(a, st, jb, masked, body=True):
    """The V phase: the softmax of query block A's scores of key block jb (40 VALU) + the fragments of the next M phase (K of the next
    key block, V^T of this one)."""
    sn = (st + 1) & 3
    kst, kjb = (st, 1) if jb == 0 else (sn, 0)
    stamp(a, 2 if jb == 0 else 6)
    for k in range(4):
        if "nolds" not in ABL:
            a.e("ds_read_b128 %s, %s offset:%d" % (vr(KF(k), 4), v(KOFF(k)), kst * STAGE + kjb * 4096))
    for f in range(4):
        if "nolds" not in ABL:
            a.e("ds_read_b128 %s, %s offset:%d" % (vr(VF(f), 4), v(VOFF(jb, f >> 1)), st * STAGE + (f & 1) * 4096))
    if body and jb == 1:
        a.e("ds_read_b32 %s, %s offset:%d" % (v(V_KACUR), v(V_KAREAD), sn * 1024))
    if masked:
        for t in mask_apply(0, jb):
            a.e(t)
    for grp in softmax_gaps(0):
        for t in grp:
            a.e(t)
    if body and not OPT["dma_in_m"]:
        rd = dma_round((st + 3) & 3)
        dma_guarded(a, [rd[0], rd[1], rd[4]] if jb == 0 else [rd[2], rd[3]], "V%d%d%s" % (st, jb, "m" if masked else "p"))
    if body and jb == 1:
        a.e("s_waitcnt lgkmcnt(0)")
        a.e("v_cmp_neq_f32 vcc, 0, %s" % v(V_KACUR))
        a.e("s_cmp_lg_u64 vcc, 0")
        a.e("s_cselect_b32 %s, 1, 0" % s("mnext"))
    stamp(a, 3 if jb == 0 else 7)
    barrier(a)


def tile_body(a, st, masked):
    for jb in range(2):
        n0 = len(a.lines)
        phase_M(a, st, jb, masked)
        phase_V(a, st, jb, masked)
        if "nobar" in ABL:
            a.lines[n0:] = [t for t in a.lines[n0:] if t != "s_barrier"]
    a.e("s_mov_b32 %s, %d" % (s("pmask"), 1 if masked else 0))          # block B's scores of key block 1 are still pending
    a.e("s_add_u32 %s, %s, 1" % (s("t"), s("t")))
    a.e("s_add_u32 %s, %s, 1" % (s("x0"), s("t")))
    a.e("s_cmp_eq_u32 %s, %s" % (s("x0"), s("nt")))
    a.e("s_cselect_b32 %s, %s, 0" % (s("x0"), s("tailf")))
    a.e("s_or_b32 %s, %s, %s" % (s("masked"), s("mnext"), s("x0")))


def light_tile(a, st):
    """a wave without queries (or a half without an item): the same four barriers, its DMA pieces and its share of the tail fix"""
    n0 = len(a.lines)
    rd = dma_round((st + 3) & 3)
    dma_guarded(a, [rd[0], rd[1], rd[4]], "L%da" % st)
    barrier(a)
    barrier(a)
    dma_guarded(a, [rd[2], rd[3]], "L%db" % st)
    for t in advance_round():
        a.e(t)
    a.e("s_waitcnt vmcnt(10)")
    tail_fix(a, (st + 1) & 3, "L%d" % st)
    barrier(a)
    barrier(a)
    a.e("s_add_u32 %s, %s, 1" % (s("t"), s("t")))
    if "nobar" in ABL:
        a.lines[n0:] = [t for t in a.lines[n0:] if t != "s_barrier"]


def item_decode(a):
    """idx -> L = xcd * cnt + idx -> (qb, h, b); have = this half has an item (idx < cnt and L < total)"""
    a.e("s_mov_b32 %s, 1" % s("have"))
    a.e("s_cmp_ge_u32 %s, %s" % (s("idx"), s("cnt")))
    a.e("s_cselect_b32 %s, 0, %s" % (s("have"), s("have")))
    a.e("s_mul_i32 %s, %s, %s" % (s("x0"), s("xcd"), s("cnt")))
    a.e("s_add_u32 %s, %s, %s" % (s("x0"), s("x0"), s("idx")))                         # L
    a.e("s_cmp_ge_u32 %s, %s" % (s("x0"), s("total")))
    a.e("s_cselect_b32 %s, 0, %s" % (s("have"), s("have")))
    a.e("s_cmp_eq_u32 %s, 0" % s("have"))
    a.e("s_cselect_b32 %s, 0, %s" % (s("x0"), s("x0")))                                # no item: decode item 0 (addresses stay legal, nothing is issued)
    a.e("s_mul_hi_u32 %s, %s, %s" % (s("x1"), s("x0"), s("mq")))
    a.e("s_cmp_eq_u32 %s, 1" % s("nqb"))
    a.e("s_cselect_b32 %s, %s, %s" % (s("x1"), s("x0"), s("x1")))
    a.e("s_mul_i32 %s, %s, %s" % (s("x2"), s("x1"), s("nqb")))
    a.e("s_sub_u32 %s, %s, %s" % (s("qb"), s("x0"), s("x2")))
    a.e("s_mul_hi_u32 %s, %s, %s" % (s("b"), s("x1"), s("mh")))
    a.e("s_cmp_eq_u32 %s, 1" % s("H"))
    a.e("s_cselect_b32 %s, %s, %s" % (s("b"), s("x1"), s("b")))
    a.e("s_mul_i32 %s, %s, %s" % (s("x2"), s("b"), s("H")))
    a.e("s_sub_u32 %s, %s, %s" % (s("h"), s("x1"), s("x2")))
    a.e("s_lshl_b32 %s, %s, 7" % (s("x2"), s("Npad")))
    a.e("s_mul_hi_u32 %s, %s, %s" % (s("x3"), s("x1"), s("x2")))
    a.e("s_mul_i32 %s, %s, %s" % (s("x2"), s("x1"), s("x2")))
    for n, base in (("Qp", "q"), ("Kp", "k"), ("Vp", "vt")):
        a.e("s_add_u32 %s, %s, %s" % (s(n), s(base), s("x2")))
        a.e("s_addc_u32 %s, %s, %s" % (shi(n), shi(base), s("x3")))
    a.e("s_mul_i32 %s, %s, %s" % (s("x2"), s("b"), s("kas")))
    a.e("s_mul_hi_u32 %s, %s, %s" % (s("x3"), s("b"), s("kas")))
    a.e("s_add_u32 %s, %s, %s" % (s("Ap"), s("ka"), s("x2")))
    a.e("s_addc_u32 %s, %s, %s" % (shi("Ap"), shi("ka"), s("x3")))
    a.e("s_mul_i32 %s, %s, %s" % (s("x2"), s("b"), s("N")))
    a.e("s_lshl_b32 %s, %s, 7" % (s("x4"), s("H")))
    a.e("s_mul_hi_u32 %s, %s, %s" % (s("x3"), s("x2"), s("x4")))
    a.e("s_mul_i32 %s, %s, %s" % (s("x2"), s("x2"), s("x4")))
    a.e("s_lshl_b32 %s, %s, 7" % (s("x5"), s("h")))
    a.e("s_add_u32 %s, %s, %s" % (s("x2"), s("x2"), s("x5")))
    a.e("s_addc_u32 %s, %s, 0" % (s("x3"), s("x3")))
    a.e("s_add_u32 %s, %s, %s" % (s("Op"), s("o"), s("x2")))
    a.e("s_addc_u32 %s, %s, %s" % (shi("Op"), shi("o"), s("x3")))
    a.e("s_lshl_b32 %s, %s, 2" % (s("q0"), s("qb")))
    a.e("s_add_u32 %s, %s, %s" % (s("q0"), s("q0"), s("wave")))
    a.e("s_lshl_b32 %s, %s, 6" % (s("q0"), s("q0")))
    a.e("s_cmp_lt_u32 %s, %s" % (s("q0"), s("N")))
    a.e("s_cselect_b32 %s, %s, 0" % (s("active"), s("have")))


def lane_constants(a):
    lane, m31, half, t0, t1 = T[0], T[1], T[2], T[3], T[4]
    a.e("v_mbcnt_lo_u32_b32 %s, -1, 0" % v(lane))
    a.e("v_mbcnt_hi_u32_b32 %s, -1, %s" % (v(lane), v(lane)))
    a.e("v_and_b32 %s, 31, %s" % (v(m31), v(lane)))
    a.e("v_lshrrev_b32 %s, 5, %s" % (v(half), v(lane)))
    a.e("v_lshlrev_b32 %s, 3, %s" % (v(V_HALF8), v(half)))
    a.e("v_and_b32 %s, 0x13, %s" % (v(t0), v(m31)))
    a.e("v_and_b32 %s, 4, %s" % (v(t1), v(m31)))
    a.e("v_lshl_or_b32 %s, %s, 1, %s" % (v(t0), v(t1), v(t0)))
    a.e("v_and_b32 %s, 8, %s" % (v(t1), v(m31)))
    a.e("v_lshrrev_b32 %s, 1, %s" % (v(t1), v(t1)))
    a.e("v_or_b32 %s, %s, %s" % (v(t0), v(t0), v(t1)))                    # kperm

    def swz(dst, row, chunk_emit):
        a.e("v_lshrrev_b32 %s, 1, %s" % (v(T[5]), v(row)))
        a.e("v_and_b32 %s, 7, %s" % (v(T[5]), v(T[5])))
        chunk_emit(T[6])
        a.e("v_xor_b32 %s, %s, %s" % (v(T[5]), v(T[5]), v(T[6])))
        a.e("v_lshlrev_b32 %s, 4, %s" % (v(T[5]), v(T[5])))
        a.e("v_lshl_add_u32 %s, %s, 7, %s" % (v(dst), v(row), v(T[5])))
    for kk in range(4):
        swz(KOFF(kk), t0, lambda r, kk=kk: a.e("v_add_u32 %s, %d, %s" % (v(r), 2 * kk, v(half))))
        a.e("v_add_u32 %s, %s, %s" % (v(KOFF(kk)), s("lds"), v(KOFF(kk))))
    for jb in range(2):
        for t2 in range(2):
            swz(VOFF(jb, t2), m31, lambda r, jb=jb, t2=t2: a.e("v_add_u32 %s, %d, %s" % (v(r), 4 * jb + 2 * t2, v(half))))
            a.e("v_add_u32 %s, %s, %s" % (v(VOFF(jb, t2)), s("lds"), v(VOFF(jb, t2))))
            a.e("v_add_u32 %s, 0x2000, %s" % (v(VOFF(jb, t2)), v(VOFF(jb, t2))))
    a.e("s_lshl_b32 %s, %s, 3" % (s("x0"), s("wave")))
    a.e("v_lshrrev_b32 %s, 3, %s" % (v(t0), v(lane)))
    a.e("v_add_u32 %s, %s, %s" % (v(t0), s("x0"), v(t0)))
    a.e("s_lshl_b32 %s, %s, 1" % (s("x1"), s("Npad")))
    for i in range(2):
        if i == 1:
            a.e("v_add_u32 %s, 32, %s" % (v(t0), v(t0)))
        a.e("v_lshrrev_b32 %s, 1, %s" % (v(T[5]), v(t0)))
        a.e("v_and_b32 %s, 7, %s" % (v(T[5]), v(T[5])))
        a.e("v_and_b32 %s, 7, %s" % (v(T[6]), v(lane)))
        a.e("v_xor_b32 %s, %s, %s" % (v(T[5]), v(T[5]), v(T[6])))
        a.e("v_lshlrev_b32 %s, 4, %s" % (v(T[5]), v(T[5])))
        a.e("v_lshl_add_u32 %s, %s, 7, %s" % (v(DK(i)), v(t0), v(T[5])))
        a.e("v_mul_lo_u32 %s, %s, %s" % (v(T[6]), v(t0), s("x1")))
        a.e("v_add_u32 %s, %s, %s" % (v(DV(i)), v(T[6]), v(T[5])))
    a.e("v_lshlrev_b32 %s, 2, %s" % (v(V_DKA), v(lane)))
    a.e("s_lshl_b32 %s, %s, 8" % (s("x0"), s("wave")))
    a.e("s_add_u32 %s, %s, %s" % (s("wa"), s("lds"), s("x0")))
    a.e("s_add_u32 %s, %s, %d" % (s("wa"), s("wa"), KADD0))
    a.e("v_add_u32 %s, %s, %s" % (v(V_KAREAD), s("wa"), v(V_DKA)))
    a.e("v_lshlrev_b32 %s, 5, %s" % (v(t1), v(half)))
    a.e("v_add_u32 %s, %s, %s" % (v(V_KAADDR), s("wa"), v(t1)))
    a.e("s_lshl_b32 %s, %s, 10" % (s("x0"), s("wave")))
    a.e("s_add_u32 %s, %s, %s" % (s("wl"), s("lds"), s("x0")))
    a.e("v_mov_b32 %s, 0xff800000" % v(V_NEGINF))


def item_prologue_active(a):
    m31, t0 = T[1], T[3]
    a.e("v_mbcnt_lo_u32_b32 %s, -1, 0" % v(T[0]))
    a.e("v_mbcnt_hi_u32_b32 %s, -1, %s" % (v(T[0]), v(T[0])))
    a.e("v_and_b32 %s, 31, %s" % (v(m31), v(T[0])))
    a.e("s_sub_u32 %s, %s, 1" % (s("x0"), s("N")))
    for x in range(2):
        a.e("v_add_u32 %s, %s, %s" % (v(t0), s("q0"), v(m31)))
        if x:
            a.e("v_add_u32 %s, 32, %s" % (v(t0), v(t0)))
        a.e("v_min_u32 %s, %s, %s" % (v(t0), s("x0"), v(t0)))
        a.e("v_lshlrev_b32 %s, 7, %s" % (v(t0), v(t0)))
        a.e("v_lshl_add_u32 %s, %s, 1, %s" % (v(QOFF(x)), v(V_HALF8), v(t0)))
        for kk in range(4):
            a.e("global_load_dwordx4 %s, %s, %s offset:%d" % (vr(Q(x, kk), 4), v(QOFF(x)), s2("Qp"), 32 * kk))


def emit_rounds_012(a, tag):
    skip = "nor%s" % tag
    a.e("s_cmp_eq_u32 %s, 0" % s("have"))
    a.e("s_cbranch_scc1 %s" % a.ref(skip))
    for r in range(3):
        for pair in dma_round(r):
            for t in pair:
                a.e(t)
        for t in advance_round():
            a.e(t)
    a.label(skip)


def flags_read(a, tag):
    """flags of this half's previous item (written behind its epilogue, at least one barrier ago)"""
    skip = "nofl%s" % tag
    a.e("s_cmp_eq_u32 %s, 0" % s("it"))
    a.e("s_cbranch_scc1 %s" % a.ref(skip))
    a.e("v_mov_b32 %s, %s" % (v(T[0]), s("lds")))
    a.e("v_add_u32 %s, 0x%x, %s" % (v(T[0]), FLAG0, v(T[0])))
    a.e("ds_read_b128 %s, %s" % (vr(T[2], 4), v(T[0])))
    a.e("s_waitcnt lgkmcnt(0)")
    a.e("v_or_b32 %s, %s, %s" % (v(T[2]), v(T[2]), v(T[3])))
    a.e("v_or3_b32 %s, %s, %s, %s" % (v(T[2]), v(T[2]), v(T[4]), v(T[5])))
    a.e("s_nop 3")
    a.e("v_readfirstlane_b32 %s, %s" % (s("x0"), v(T[2])))
    a.e("s_cmp_eq_u32 %s, 0" % s("x0"))
    a.e("s_cbranch_scc1 %s" % a.ref(skip))
    a.e("s_sub_u32 %s, %s, 1" % (s("x0"), s("it")))
    a.e("s_bitset1_b64 %s, %s" % (s2("bad"), s("x0")))
    a.label(skip)


def epilogue(a):
    l0, l1, lane, m31, row = T[0], T[1], T[2], T[3], T[4]
    a.e("s_mov_b32 %s, 0" % s("ibad"))
    a.e("v_mbcnt_lo_u32_b32 %s, -1, 0" % v(lane))
    a.e("v_mbcnt_hi_u32_b32 %s, -1, %s" % (v(lane), v(lane)))
    a.e("v_and_b32 %s, 31, %s" % (v(m31), v(lane)))
    a.e("s_lshl_b32 %s, %s, 7" % (s("x4"), s("H")))
    buf = BUF
    for x in range(2):
        a.e("v_add_f32 %s, %s, %s" % (v(l0), v(PS(x, 0)), v(PS(x, 1))))
        a.e("v_mov_b32 %s, %s" % (v(l1), v(l0)))
        a.e("s_nop 1")
        a.e("v_permlane32_swap_b32 %s, %s" % (v(l0), v(l1)))
        a.e("s_nop 1")
        a.e("v_add_f32 %s, %s, %s" % (v(l0), v(l0), v(l1)))
        a.e("v_cmp_gt_f32 vcc, 0x71800000, %s" % v(l0))
        a.e("s_mov_b64 %s, vcc" % s2("x0"))
        a.e("v_cmp_lt_f32 vcc, 0x0d800000, %s" % v(l0))
        a.e("s_and_b64 vcc, vcc, %s" % s2("x0"))
        a.e("s_andn2_b64 %s, exec, vcc" % s2("x0"))
        a.e("s_cmp_lg_u64 %s, 0" % s2("x0"))
        a.e("s_cselect_b32 %s, 1, 0" % s("x0"))
        a.e("s_or_b32 %s, %s, %s" % (s("ibad"), s("ibad"), s("x0")))
        a.e("v_rcp_f32 %s, %s" % (v(l0), v(l0)))
        a.e("v_add_u32 %s, %s, %s" % (v(row), s("q0"), v(m31)))
        if x:
            a.e("v_add_u32 %s, 32, %s" % (v(row), v(row)))
        a.e("s_nop 0")
        for db in range(2):
            for r in range(16):
                a.e("v_mul_f32 %s, %s, %s" % (v(O(x, db, r)), v(O(x, db, r)), v(l0)))
        a.e("v_cmp_gt_u32 vcc, %s, %s" % (s("N"), v(row)))
        a.e("v_mul_lo_u32 %s, %s, %s" % (v(row), v(row), s("x4")))
        a.e("v_lshl_add_u32 %s, %s, 1, %s" % (v(row), v(V_HALF8), v(row)))
        regs = []
        for db in range(2):
            for gp in range(2):
                base = buf
                buf += 4
                for i in range(4):
                    a.e("v_cvt_pk_bf16_f32 %s, %s, %s" % (v(base + i), v(O(x, db, 8 * gp + 2 * i)), v(O(x, db, 8 * gp + 2 * i + 1))))
                regs.append((base, 64 * db + 32 * gp))
        a.e("s_nop 1")
        for base, off in regs:
            a.e("v_permlane32_swap_b32 %s, %s" % (v(base), v(base + 2)))
            a.e("v_permlane32_swap_b32 %s, %s" % (v(base + 1), v(base + 3)))
        a.e("s_nop 1")
        a.e("s_and_saveexec_b64 %s, vcc" % s2("x2"))
        for base, off in regs:
            a.e("global_store_dwordx4 %s, %s, %s offset:%d" % (v(row), vr(base, 4), s2("Op"), off))
        a.e("s_mov_b64 exec, %s" % s2("x2"))


class Asm(g1.Asm):
    pass


def generate(trace=False):
    global TRACE
    TRACE = trace
    a = Asm()
    if trace:
        a.e("s_mov_b64 s[100:101], %[tr]")
    for n in ("q", "k", "vt", "ka", "o"):
        a.e("s_mov_b64 %s, %%[%s]" % (s2(n), n))
    for n in ("N", "Npad", "H", "total", "cnt", "kas", "nqb", "mq", "mh"):
        a.e("s_mov_b32 %s, %%[%s]" % (s(n), n))
    # wave 0..7 -> half, wave within the half; this half's LDS; its item index walk on XCD xcd: idx = 2 (r Gx + wl) + half
    a.e("s_lshr_b32 %s, %%[wave], 2" % s("half"))
    a.e("s_and_b32 %s, %%[wave], 3" % s("wave"))
    a.e("s_mul_i32 %s, %s, %d" % (s("x0"), s("half"), HALF))
    a.e("s_add_u32 %s, %%[lds], %s" % (s("lds"), s("x0")))
    a.e("s_and_b32 %s, %%[wg], 7" % s("xcd"))
    a.e("s_lshr_b32 %s, %%[wg], 3" % s("x0"))
    a.e("s_lshl_b32 %s, %s, 1" % (s("idx"), s("x0")))
    a.e("s_add_u32 %s, %s, %s" % (s("idx"), s("idx"), s("half")))
    a.e("s_lshr_b32 %s, %%[G], 2" % s("istr"))                            # 2 * (G / 8)
    a.e("s_add_u32 %s, %s, 63" % (s("nt"), s("N")))
    a.e("s_lshr_b32 %s, %s, 6" % (s("nt"), s("nt")))
    a.e("s_sub_u32 %s, %s, 1" % (s("x0"), s("nt")))
    a.e("s_lshl_b32 %s, %s, 6" % (s("x0"), s("x0")))
    a.e("s_sub_u32 %s, %s, %s" % (s("ntail"), s("N"), s("x0")))
    a.e("s_and_b32 %s, %s, 63" % (s("tailf"), s("N")))
    a.e("s_cmp_lg_u32 %s, 0" % s("tailf"))
    a.e("s_cselect_b32 %s, 1, 0" % s("tailf"))
    a.e("s_mov_b64 %s, 0" % s2("bad"))
    a.e("s_mov_b32 %s, 0" % s("it"))
    lane_constants(a)
    # half 1 runs one barrier behind half 0 from here to the end
    a.e("s_cmp_eq_u32 %s, 0" % s("half"))
    a.e("s_cbranch_scc1 %s" % a.ref("item"))
    barrier(a)

    # ================================================================ item loop (both halves take the same number of turns)
    a.label("item")
    a.e("s_sub_u32 %s, %s, %s" % (s("x0"), s("idx"), s("half")))
    a.e("s_cmp_ge_u32 %s, %s" % (s("x0"), s("cnt")))
    a.e("s_cbranch_scc1 %s" % a.ref("done"))
    item_decode(a)
    # ---- phase PRO: flags of the previous item, q, rounds 0..2, zeroes
    flags_read(a, "a")
    if trace:
        a.e("s_and_b32 %s, %%[wave], 3" % s("x0"))
        a.e("s_or_b32 %s, %s, %%[wg]" % (s("x0"), s("x0")))
        a.e("s_or_b32 %s, %s, %s" % (s("x0"), s("x0"), s("it")))
        a.e("s_cmp_eq_u32 %s, 0" % s("x0"))
        a.e("s_cselect_b32 %s, 1, 0" % s("trace"))
    a.e("s_mov_b32 %s, 0" % s("r"))
    a.e("s_mov_b32 %s, 0" % s("t"))
    a.e("s_cmp_eq_u32 %s, 0" % s("active"))
    a.e("s_cbranch_scc1 %s" % a.ref("light"))

    item_prologue_active(a)
    emit_rounds_012(a, "a")
    for x in range(2):
        for db in range(2):
            for r in range(16):
                a.e("v_mov_b32 %s, 0" % v(O(x, db, r)))
        for i in range(2):
            a.e("v_mov_b32 %s, 0" % v(PS(x, i)))
        for t2 in range(2):
            for r in range(4):
                a.e("v_mov_b32 %s, 0" % v(P(x, t2) + r))
    for f in range(4):                                                    # "key block -1": P V against P = 0 needs finite V^T fragments
        for r in range(4):
            a.e("v_mov_b32 %s, 0" % v(VF(f) + r))
    for r in range(16):                                                   # and block B's "pending scores" exponentiate to 0
        a.e("v_mov_b32 %s, 0xff800000" % v(S(1, r)))
    a.e("s_mov_b32 %s, 0" % s("pmask"))
    a.e("s_waitcnt vmcnt(10)")
    barrier(a)
    # ---- phase PRE: K fragments of (tile 0, key block 0), mask flag of tile 0
    for k in range(4):
        a.e("ds_read_b128 %s, %s offset:%d" % (vr(KF(k), 4), v(KOFF(k)), 0))
    a.e("ds_read_b32 %s, %s offset:0" % (v(V_KACUR), v(V_KAREAD)))
    a.e("s_waitcnt lgkmcnt(0)")
    a.e("v_cmp_neq_f32 vcc, 0, %s" % v(V_KACUR))
    a.e("s_cmp_lg_u64 vcc, 0")
    a.e("s_cselect_b32 %s, 1, 0" % s("masked"))
    barrier(a)
    # ---- tile loop
    a.label("tile0")
    for st in range(4):
        if st:
            a.label("tile%d" % st)
        a.e("s_cmp_ge_u32 %s, %s" % (s("t"), s("nt")))
        a.e("s_cbranch_scc1 %s" % a.ref("drain"))
        a.e("s_cmp_lg_u32 %s, 0" % s("masked"))
        a.e("s_cbranch_scc1 %s" % a.ref("mtile%d" % st))
        tile_body(a, st, False)
        if st == 3:
            a.e("s_branch %s" % a.ref("tile0"))
    for st in range(4):
        a.label("mtile%d" % st)
        tile_body(a, st, True)
        a.e("s_branch %s" % a.ref("tile%d" % ((st + 1) & 3)))
    # ---- drain: P V of the last key block (an M phase), then the epilogue (a V-like phase)
    a.label("drain")
    # block B's scores of the last key block are still pending: their mask term (if any) first
    a.e("s_cmp_eq_u32 %s, 0" % s("pmask"))
    a.e("s_cbranch_scc1 %s" % a.ref("dnm"))
    for t in mask_apply(1, 1):
        a.e(t)
    a.label("dnm")
    phase_M(a, 0, 0, False, qk=False, body=False, tag="D")
    a.e("s_nop 7")
    epilogue(a)
    a.e("s_branch %s" % a.ref("itemend"))

    # ---------------------------------------------------------------- wave without queries / half without an item
    a.label("light")
    emit_rounds_012(a, "l")
    a.e("s_waitcnt vmcnt(10)")
    barrier(a)
    barrier(a)
    a.e("s_mov_b32 %s, 0" % s("ibad"))
    a.label("ltile0")
    for st in range(4):
        if st:
            a.label("ltile%d" % st)
        a.e("s_cmp_ge_u32 %s, %s" % (s("t"), s("nt")))
        a.e("s_cbranch_scc1 %s" % a.ref("ldrain"))
        light_tile(a, st)
        if st == 3:
            a.e("s_branch %s" % a.ref("ltile0"))
    a.label("ldrain")
    barrier(a)

    # ---------------------------------------------------------------- item end (inside the epilogue phase): flag, drain, barrier
    a.label("itemend")
    a.e("v_mov_b32 %s, %s" % (v(T[0]), s("wave")))
    a.e("v_lshl_add_u32 %s, %s, 2, %s" % (v(T[0]), v(T[0]), s("lds")))
    a.e("v_add_u32 %s, 0x%x, %s" % (v(T[0]), FLAG0, v(T[0])))
    a.e("v_mov_b32 %s, %s" % (v(T[1]), s("ibad")))
    a.e("ds_write_b32 %s, %s" % (v(T[0]), v(T[1])))
    a.e("s_waitcnt vmcnt(0) lgkmcnt(0)")
    barrier(a)
    a.e("s_add_u32 %s, %s, %s" % (s("it"), s("it"), s("have")))
    a.e("s_add_u32 %s, %s, %s" % (s("idx"), s("idx"), s("istr")))
    a.e("s_branch %s" % a.ref("item"))

    # ================================================================ the last item's flags; half 0 pays its barrier back
    a.label("done")
    flags_read(a, "z")
    a.e("s_cmp_lg_u32 %s, 0" % s("half"))
    a.e("s_cbranch_scc1 %s" % a.ref("out"))
    barrier(a)
    a.label("out")
    if trace:
        a.e("s_dcache_wb")
    a.e("s_mov_b64 %[bad], " + s2("bad"))
    return a.lines


def main():
    trace = "--trace" in sys.argv
    out = sys.argv[sys.argv.index("--trace") + 1] if trace else OUT
    lines = generate(trace)
    with open(out, "w") as f:
        f.write("// GENERATED by tools/gen/attn_pp_gen.py -- do not edit.  %d instructions / labels.\n" % len(lines))
        f.write("// LDS per half: %d stages x %d B, key_add rows at %d, flags at %d; %d bytes per half, %d per workgroup.\n" % (NS, STAGE, KADD0, FLAG0, HALF, LDS_BYTES))
        for ln in lines:
            f.write('"%s\\n\\t"\n' % ln)
    print("wrote", out, len(lines), "lines")
    if "--lint" in sys.argv:
        probs = g1.lint(lines)
        for n, ln, why in probs[:40]:
            print("line %d: %s  <- %s" % (n, ln, why))
        print("%d hazard findings" % len(probs))


if __name__ == "__main__":
    main()
