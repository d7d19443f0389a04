"""Generator of the hand-scheduled fused-attention item walk (attn_p64_kernel, uvltrack_amd/csrc/attention.hip).

    python tools/gen/attn_p64_gen.py          # rewrites uvltrack_amd/csrc/attn_p64_asm.inc

The output is ONE inline-asm statement body (a C string literal): the whole pass-1 walk of a persistent workgroup over its items --
item decode, q loads, DMA ring, key tiles, drain, normalise + store -- with every register owned by this file.  hipcc sees only SGPR
operands and the clobber list, so its register allocator (the 256-register wall of attn_w64_kernel) is out of the picture.

Same math, LDS image and MFMA operand trick as attn_w64_kernel (reference: lib/models/backbones/block.py:47-61 -- q k^T * scale,
masked_fill, softmax, @ v): S^T = K Q^T per 32 keys x 32 queries, p = exp2(s) with NO running maximum (checked once per item, failing
items are redone by the compiler-scheduled exact pass), P^T feeds V^T P^T as the B operand without data movement.

Schedule (profiles/r03_attention.md has the measurements behind every choice).  A wave owns 64 queries = two 32-query blocks A, B;
a 64-key tile is two key blocks jb = 0, 1.  Per key block a wave runs
    M phase (s_setprio 1): block A's 4 score MFMAs and 4 P V MFMAs (of the previous key block) interleaved, with block B's PREVIOUS
            softmax in their gaps (five VALU per gap: the wave waits for the matrix pipe there anyway), then block B's 8 MFMAs bare with
            the pointer arithmetic and the LDS-DMA issue of round t + 2 in their gaps
    V phase (priority 0): block A's softmax (40 VALU) and the 8 fragment reads of the next M phase.
The two co-resident workgroups of a CU are NOT synchronised: a wave in its M phase simply outranks the other workgroup's wave in its V
phase, so matrix and VALU phases of the two waves of a SIMD alternate by themselves (a first version interleaved MFMAs and VALU finely
in both waves: the older wave then ran at ~70 cycles per MFMA and the younger at ~210; an 8-wave version with a barrier per phase was
fair but paid for the lock-step: both in the history of this file).  One barrier per tile (the DMA round's publication), 32 live score
registers, every K / V^T fragment read once per tile, the ring runs two rounds ahead.  Options P64_OPT / P64_ABL: measured alternatives
(LDS-DMA or pointer arithmetic in the V phase, block B's softmax over ten gaps: all slower) and timing-only ablations for the trace tools.
"""
import os

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
OUT = os.path.join(ROOT, "uvltrack_amd", "csrc", "attn_p64_asm.inc")

# ---------------------------------------------------------------- LDS image (bytes from the workgroup's LDS base)
STAGE = 16384            # K tile 8 KB + V^T tile 8 KB
NS = 4
KADD0 = NS * STAGE       # [stage][wave][64] f32 key_add rows
FLAG0 = KADD0 + NS * 4 * 256
LDS_BYTES = FLAG0 + 16

# ---------------------------------------------------------------- VGPR map
def O(x, db, r): return 32 * x + 16 * db + r
def Q(x, kk): return 64 + 16 * x + 4 * kk
def S(par, r): return 96 + 16 * par + r
def P(par, t2): return 128 + 8 * par + 4 * t2
def KF(buf, k): return 144 + 16 * buf + 4 * k
def VF(buf, f): return 176 + 16 * buf + 4 * f          # f = 2 * t2 + db
def KA(r): return 208 + r                          # mask term of key block 0 of a masked tile, in score-register order: the C operand of its score MFMAs
def KB(r): return 192 + r                          # the same for key block 1 (loaded while key block 0's MFMAs still read KA)
def PS(x, i): return 224 + 2 * x + i               # (P64_OPT rsum=0 only: row sums as f32 adds)
def PSQ(x): return 164 + 4 * x                     # row-sum accumulator quad of query block x (all four registers hold the same sum)
V_ONES = 160                                       # v160:161 = four bf16 ones: the A operand of the row-sum MFMA
def KOFF(kk): return 228 + kk
def VOFF(jb, t2): return 232 + 2 * jb + t2
def DK(i): return 236 + i
def DV(i): return 238 + i
V_DKA, V_KAREAD, V_KAADDR, V_LIM, V_NEGINF, V_KACUR = 240, 241, 242, 243, 244, 245
def QOFF(x): return 246 + x
V_HALF8 = 255
T = [248, 249, 250, 251, 252, 253, 254]          # T[2..5] is an aligned quad

# ---------------------------------------------------------------- SGPR map (fixed; the operands are copied here first)
SG = dict(q=40, k=42, vt=44, ka=46, o=48, N=50, Npad=51, H=52, total=53, cnt=54, v=55, G=56, vend=57, kas=58, wave=59, lds=60,
          nqb=61, mq=62, mh=63, nt=64, ntail=65, tailf=66, qb=67, h=68, b=69, Kp=70, Vp=72, Ap=74, Qp=76, Op=78, t=80, masked=81,
          mnext=82, active=83, r=84, it=85, bad=86, x0=88, x1=89, x2=90, x3=91, x4=92, x5=93, wl=94, wa=95, ibad=96, q0=97, vrow=98, x6=99)


def s(n): return "s%d" % SG[n]
def s2(n): return "s[%d:%d]" % (SG[n], SG[n] + 1)
def shi(n): return "s%d" % (SG[n] + 1)
def v(n): return "v%d" % n
def vr(n, c): return "v[%d:%d]" % (n, n + c - 1)


class Asm:
    def __init__(self):
        self.lines = []
        self.nlabel = 0
        self.in_loop = False

    def e(self, text):
        # timing-only ablations (P64_ABL; results WRONG, build with -DATTN_P64_NOFALLBACK): instruction classes of the TILE LOOP dropped or replaced
        if self.in_loop:
            op = text.split(" ", 1)[0]
            if "nomfma" in ABL and op.startswith("v_mfma"):
                return
            if "nosoftmax" in ABL and op in ("v_exp_f32", "v_cvt_pk_bf16_f32") or ("nosoftmax" in ABL and op == "v_add_f32"):
                return
            if "noexp" in ABL and op == "v_exp_f32":
                text = "v_mov_b32" + text[len(op):]
            if "nolds" in ABL and op == "ds_read_b128":
                return
            if "nodma" in ABL and op.startswith("global_load_lds"):
                return
        self.lines.append(text)

    def label(self, name):
        self.lines.append(name + "_%=:")

    def ref(self, name):
        return name + "_%="

    def comment(self, text):
        self.lines.append("; " + text)


# ---------------------------------------------------------------- pieces
def mfma(dst, a, b, c):
    if "mfma16" in ABL:        # timing probe (results wrong; build attention.hip with -DATTN_P64_NOFALLBACK): the same flops as two 16x16x32 MFMAs
        return "\\n\\t".join("v_mfma_f32_16x16x32_bf16 %s, %s, %s, %s" % (vr(dst + 4 * h, 4), vr(a, 4), vr(b, 4), "0" if c is None else vr(c + 4 * h, 4))
                              for h in range(2))
    return "v_mfma_f32_32x32x16_bf16 %s, %s, %s, %s" % (vr(dst, 16), vr(a, 4), vr(b, 4), "0" if c is None else vr(c, 16))


def dma_round(stage):
    """The five LDS-DMA instructions of one round into `stage`, as (m0 set, load) pairs; the pointer update comes last."""
    out = []
    for i in range(2):
        out.append(["s_add_u32 m0, %s, %d" % (s("wl"), stage * STAGE + i * 4096), "s_nop 0", "global_load_lds_dwordx4 %s, %s" % (v(DK(i)), s2("Kp"))])
    for i in range(2):
        out.append(["s_add_u32 m0, %s, %d" % (s("wl"), stage * STAGE + 8192 + i * 4096), "s_nop 0", "global_load_lds_dwordx4 %s, %s" % (v(DV(i)), s2("Vp"))])
    out.append(["s_add_u32 m0, %s, %d" % (s("wa"), stage * 1024), "s_nop 0", "global_load_lds_dword %s, %s" % (v(V_DKA), s2("Ap"))])
    return out


def advance_round():
    """Round pointers walk to the next key tile; past the item's last tile they stay (the surplus rounds reload the last tile into
    stages nobody reads: the vmcnt arithmetic stays uniform and no address leaves the allocation)."""
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


def tail_fix(a, stage, tag):
    """Zero the V^T columns AND the K rows of keys >= N in this wave's own pieces of the tail tile (NaN-proof: 0 x garbage would poison
    P V, and a garbage K row would poison the score the -inf mask term is ADDED to -- the mask term is the C operand of the score MFMA since
    round 5); executed once per item, behind the wave's own vmcnt wait and in front of the barrier."""
    skip = "tfx%s" % tag
    a.e("s_add_u32 %s, %s, 2" % (s("x0"), s("t")))                       # this barrier publishes tile t + 1; the tail tile is nt - 1
    a.e("s_cmp_eq_u32 %s, %s" % (s("x0"), s("nt")))
    a.e("s_cselect_b32 %s, %s, 0" % (s("x0"), s("tailf")))
    a.e("s_cmp_eq_u32 %s, 0" % s("x0"))
    a.e("s_cbranch_scc1 %s" % a.ref(skip))
    t0, t1, m1, m2 = T[0], T[1], T[6], V_KACUR
    d = T[2]                                                              # T[2..5] = v250..v253, an aligned quad
    a.e("v_mbcnt_lo_u32_b32 %s, -1, 0" % v(t1))
    a.e("v_mbcnt_hi_u32_b32 %s, -1, %s" % (v(t1), v(t1)))                 # lane
    a.e("v_and_b32 %s, 7, %s" % (v(t0), v(t1)))
    a.e("v_lshrrev_b32 %s, 4, %s" % (v(m1), v(t1)))
    a.e("s_and_b32 %s, %s, 1" % (s("x0"), s("wave")))
    a.e("s_lshl_b32 %s, %s, 2" % (s("x0"), s("x0")))
    a.e("v_add_u32 %s, %s, %s" % (v(m1), s("x0"), v(m1)))
    a.e("v_xor_b32 %s, %s, %s" % (v(t0), v(t0), v(m1)))
    a.e("v_lshlrev_b32 %s, 3, %s" % (v(t0), v(t0)))                       # 8 * chunk
    a.e("v_sub_u32 %s, %s, %s" % (v(t0), s("ntail"), v(t0)))              # valid keys of this lane's 8
    a.e("v_lshlrev_b32 %s, 4, %s" % (v(t1), v(t1)))
    a.e("v_add_u32 %s, %s, %s" % (v(t1), s("wl"), v(t1)))                 # own piece 0 of stage 0's K half
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
    # K rows: piece i of this wave holds rows 8 (wave + 4 i) + (lane >> 3), 16 bytes per lane at t1 + i * 4096
    a.e("v_mbcnt_lo_u32_b32 %s, -1, 0" % v(m1))
    a.e("v_mbcnt_hi_u32_b32 %s, -1, %s" % (v(m1), v(m1)))
    a.e("v_lshrrev_b32 %s, 3, %s" % (v(m1), v(m1)))
    a.e("s_lshl_b32 %s, %s, 3" % (s("x0"), s("wave")))
    a.e("v_add_u32 %s, %s, %s" % (v(m1), s("x0"), v(m1)))                 # row of piece 0
    for e in range(4):
        a.e("v_mov_b32 %s, 0" % v(d + e))
    for i in range(2):
        if i:
            a.e("v_add_u32 %s, 32, %s" % (v(m1), v(m1)))
        a.e("v_cmp_le_i32 vcc, %s, %s" % (s("ntail"), v(m1)))              # ntail <= row: a key >= N
        a.e("s_and_saveexec_b64 %s, vcc" % s2("x2"))
        a.e("ds_write_b128 %s, %s offset:%d" % (v(t1), vr(d, 4), stage * STAGE + i * 4096))
        a.e("s_mov_b64 exec, %s" % s2("x2"))
    a.e("s_waitcnt lgkmcnt(0)")
    a.label(skip)


def ka_load(stage, jb, base):
    """this wave's copy of the tile's key_add row, the 16 values of key block jb in score-register order, into base .. base + 15"""
    out = []
    for gq in range(4):
        off = stage * 1024 + jb * 128 + (gq >> 1) * 64 + (gq & 1) * 16
        out.append("ds_read_b128 %s, %s offset:%d" % (vr(base + 4 * gq, 4), v(V_KAADDR), off))
    return out


def ka_scale(base):
    return ["v_mul_f32 %s, 0x3fb8aa3b, %s" % (v(base + r), v(base + r)) for r in range(16)]       # log2 domain


def ka_limit(base, jb):
    """term = key < limit ? term : -inf   (limit = valid keys of the tile - 8 * half; 64 - 8 * half unless it is the tail).  Once per key
    block and wave -- the term then enters BOTH query blocks' scores as the C operand of their first score MFMA; until round 4 it was added
    and limited per query block behind the MFMAs (48 VALU per block and key block, plus a 48-VALU block for the pending scores)."""
    out = []
    for r in range(16):
        key = 32 * jb + 16 * (r >> 3) + 4 * ((r >> 2) & 1) + (r & 3)
        out.append("v_cmp_lt_i32 vcc, %d, %s" % (key, v(V_LIM)))
        out.append("v_cndmask_b32 %s, %s, %s, vcc" % (v(base + r), v(V_NEGINF), v(base + r)))
    return out


def masked_tile_entry(a, st):
    """in front of a masked tile's first M phase: the tile's key limit and the mask term of key block 0"""
    a.e("s_add_u32 %s, %s, 1" % (s("x0"), s("t")))
    a.e("s_cmp_eq_u32 %s, %s" % (s("x0"), s("nt")))
    a.e("s_cselect_b32 %s, %s, 0" % (s("x0"), s("tailf")))
    a.e("s_cmp_lg_u32 %s, 0" % s("x0"))
    a.e("s_cselect_b32 %s, %s, 64" % (s("x0"), s("ntail")))
    a.e("v_sub_u32 %s, %s, %s" % (v(V_LIM), s("x0"), v(V_HALF8)))
    for t in ka_load(st, 0, KA(0)):
        a.e(t)
    a.e("s_waitcnt lgkmcnt(0)")
    for t in ka_scale(KA(0)) + ka_limit(KA(0), 0):
        a.e(t)


OPT = dict(prio=1, dma_v=0, spread=0, adv_v=0, rsum=0, pure=0, rdprio=0, osc1=0)      # osc1: the output rows stored write-through (sc1): -1.6 % on the configs[4] frame (round 5)
for _kv in os.environ.get("P64_OPT", "").split(","):
    if "=" in _kv:
        OPT[_kv.split("=")[0]] = int(_kv.split("=")[1])
ABL = set(x for x in os.environ.get("P64_ABL", "").split(",") if x)      # timing-only ablations (WRONG results): prio0, nobar
if "prio0" in ABL:
    OPT["prio"] = 0
SG["pmask"] = 39
SG["ptr"] = 38


def rowsum_mfma(x, j):
    """Row sums on the matrix pipe (round 5): v_mfma_f32_4x4x4_16b_bf16 computes, for each of 16 blocks of four lanes, D = A B + C with 4 x 4
    operands; with A = ones, lane l's four result registers all become C + the sum of the four bf16 values lane l itself holds in its B
    registers.  One such instruction on an aligned pair of packed P registers replaces four v_add_f32: 8 cycles of the matrix pipe and one
    issue slot instead of four slots of the VALU port, which is what bounds this loop at head_dim 64 (2 exponentials + 2 adds + 1 pack per
    32-cycle MFMA; profiles/r05_attention.md).  The sum is that of the bf16-rounded P the P V MFMA multiplies -- numerator and denominator of
    the softmax now see the same numbers."""
    q = PSQ(x)
    return "v_mfma_f32_4x4x4_16b_bf16 %s, %s, %s, %s" % (vr(q, 4), vr(V_ONES, 2), vr(P(x, j >> 1) + 2 * (j & 1), 2), vr(q, 4))


def softmax_gaps(x):
    return _softmax_gaps(x, OPT["rsum"] == 1 or (OPT["rsum"] == 2 and x == 0) or (OPT["rsum"] == 3 and x == 1))


def _softmax_gaps(x, by_mfma):
    """One query block's softmax in groups (eight, or nine with the MFMA row sums): exp2 pairs lead, the pack (and the row-sum adds of the
    rsum=0 form) trail one group so nothing waits for the transcendental pipe; a row-sum MFMA follows the second pack of its register pair
    by at least two issue slots (VALU write -> MFMA read)"""
    ex = lambda r: "v_exp_f32 %s, %s" % (v(S(x, r)), v(S(x, r)))
    ad = lambda r: "v_add_f32 %s, %s, %s" % (v(PS(x, r & 1)), v(PS(x, r & 1)), v(S(x, r)))
    cv = lambda i: "v_cvt_pk_bf16_f32 %s, %s, %s" % (v(P(x, i >> 2) + (i & 3)), v(S(x, 2 * i)), v(S(x, 2 * i + 1)))
    gaps = [[] for _ in range(8)]
    if by_mfma:
        for i in range(8):
            gaps[i] += [ex(2 * i), ex(2 * i + 1)]
            if i >= 1:
                gaps[i] += [cv(i - 1)]
            if i in (3, 5, 7) and "norsum" not in ABL:      # packs 2 j, 2 j + 1 are in groups 2 j + 1, 2 j + 2: the MFMA rides in group 2 j + 3, behind two exponentials
                gaps[i].insert(2, rowsum_mfma(x, (i - 3) // 2))
        gaps[7] += [cv(7)]
        gaps.append([rowsum_mfma(x, 3)] if "norsum" not in ABL else [])            # (its caller keeps two issue slots between the last pack and this one)
        return gaps
    for i in range(8):
        gaps[i] += [ex(2 * i), ex(2 * i + 1)]
        if i >= 1:
            gaps[i] += [ad(2 * i - 2), ad(2 * i - 1), cv(i - 1)]
    gaps[7] += [ad(14), ad(15), cv(7)]
    return gaps


def pstamp(a, k):
    """trace build: wave 0 of workgroups 0 and 256 stores the shader clock at phase boundary k of the current tile of its first item
    (records behind the per-item ones: byte 300000 + [wg != 0][tile][k] * 8)"""
    if not TRACE or ("onestamp" in ABL and k != 0):
        return
    skip = "ps%d" % a.nlabel
    a.nlabel += 1
    a.e("s_cmp_eq_u32 %s, 0" % s("ptr"))
    a.e("s_cbranch_scc1 %s" % a.ref(skip))
    a.e("s_memtime s[98:99]")
    a.e("s_lshl_b32 %s, %s, 6" % (s("x5"), s("t")))
    a.e("s_add_u32 %s, %s, %s" % (s("x5"), s("x5"), s("ptr")))
    a.e("s_add_u32 %s, %s, %d" % (s("x5"), s("x5"), 8 * k))
    a.e("s_waitcnt lgkmcnt(0)")
    a.e("s_store_dwordx2 s[98:99], s[100:101], %s" % s("x5"))
    a.e("s_waitcnt lgkmcnt(0)")
    a.label(skip)


def phase_M(a, st, jb, masked, qk=True, body=True):
    """The M phase of key block jb of tile t (s_setprio 1: a wave in its matrix phase outranks the co-resident workgroup's wave in
    its VALU phase, so the two alternate without a barrier between them): for query block A  4 score MFMAs + 4 P V MFMAs of the
    previous key block, with the softmax of query block B's PREVIOUS scores in their gaps (five VALU per gap -- this wave waits for
    the matrix pipe there anyway); then the same 8 MFMAs for block B, bare: pointer arithmetic and the LDS-DMA issue of round t + 2
    ride in those gaps.  Key block 1 ends with the tile's one barrier (round t + 1 published)."""
    if body:
        pstamp(a, 0 if jb == 0 else 4)
    if OPT["prio"]:
        a.e("s_setprio 1")
    a.e("s_waitcnt lgkmcnt(0)")
    mf = []
    for x in range(2):
        for k in range(4):
            if qk:
                c0 = (KA(0) if jb == 0 else KB(0)) if masked else None      # the mask term rides in as the accumulator's initial value
                mf.append(mfma(S(x, 0), KF(0, k), Q(x, k), c0 if k == 0 else S(x, 0)))
            mf.append(mfma(O(x, k & 1, 0), VF(0, k), P(x, k >> 1), O(x, k & 1, 0)))         # fragment k = 2 t2 + db
    gaps = softmax_gaps(1)
    nA = 8 if qk else 4                             # drain: P V only, block B's softmax in the four gaps of block A's MFMAs
    fill = {i: [] for i in range(len(mf))}
    if OPT["pure"]:
        # measured alternative: BOTH query blocks' softmax in the V phase; the M phase is 16 bare MFMAs (+ the scalar / DMA fillers).  pure = 3: the
        # two blocks' MFMAs interleaved k step by k step (four independent accumulation chains instead of two)
        if OPT["pure"] == 3 and qk:
            mf = [mf[8 * x + 2 * k + h] for k in range(4) for h in range(2) for x in range(2)]
    elif qk and OPT["spread"]:
        # block B's MFMAs reordered so that its softmax may use ten gaps: P V (t2 = 0) first, its scores (which overwrite the registers
        # the softmax reads) behind the last VALU group
        pvb = lambda k: mfma(O(1, k & 1, 0), VF(0, k), P(1, k >> 1), O(1, k & 1, 0))
        qkb = lambda k: mfma(S(1, 0), KF(0, k), Q(1, k), None if k == 0 else S(1, 0))
        mf = mf[:8] + [pvb(0), pvb(1), qkb(0), pvb(2), qkb(1), pvb(3), qkb(2), qkb(3)]
        flat = [t for grp in gaps for t in grp]
        # groups 0..4 (P fragment t2 = 0 complete) within gaps 0..7, the rest within gaps 8, 9
        cvfirst = lambda grp: [t for t in grp if t.startswith("v_cvt")] + [t for t in grp if not t.startswith("v_cvt")]
        first = [t for grp in gaps[:5] for t in cvfirst(grp)]          # a pack is never the last VALU in front of the MFMA that reads it
        rest = [t for grp in gaps[5:7] for t in cvfirst(grp)] + cvfirst(gaps[7][:5]) + cvfirst(gaps[7][5:])
        for n, t in enumerate(first):
            fill[n * 8 // len(first)].append(t)
        for n, t in enumerate(rest):
            fill[8 + n * 2 // len(rest)].append(t)
    else:
        for i in range(8):
            fill[i * nA // 8] += gaps[i]
        if len(gaps) > 8:                           # the last row-sum MFMA: two issue slots behind the last pack (block B's first two MFMAs)
            fill[nA + 1] += gaps[8]
    pre = []
    late = {i: [] for i in range(len(mf))}
    if body:
        rd = dma_round((st + 2) & 3)
        if not OPT["dma_v"]:
            if jb == 0:
                late[11] += rd[0]
                late[13] += rd[1]
                late[14] += rd[4]
            else:
                late[11] += rd[2]
                late[13] += rd[3]
        if jb == 1 and not OPT["adv_v"]:
            adv = advance_round()
            late[14] += adv[0:5]
            late[15] += adv[5:11]
        if masked and jb == 0:
            # key block 1's mask term, into its own registers (block B's first score MFMA of THIS phase still reads KA), in block B's bare gaps
            late[8] += ka_load(st, 1, KB(0))
            sc, lm = ka_scale(KB(0)), ka_limit(KB(0), 1)
            late[9] += ["s_waitcnt lgkmcnt(0)"] + sc[:8]
            late[10] += sc[8:] + lm[:8]
            late[12] += lm[8:20]
            late[15] += lm[20:]
    for i, m in enumerate(mf):
        a.e(m)
        if i == 0:
            for t in pre:
                if t.startswith("LABEL "):
                    a.label(t[6:])
                else:
                    a.e(t)
        for t in fill[i] + late[i]:
            a.e(t)
    if OPT["prio"] and not (OPT["rdprio"] and body):
        a.e("s_setprio 0")
    if body:
        pstamp(a, 1 if jb == 0 else 5)
    if body and jb == 1:
        a.e("s_waitcnt vmcnt(5)")                      # round t + 1 has landed (own pieces); round t + 2 flies
        tail_fix(a, (st + 1) & 3, "%d%s" % (st, "m" if masked else "p"))
        if "nobar" not in ABL:
            a.e("s_barrier")
        pstamp(a, 6)


def phase_V(a, st, jb, masked, body=True):
    """The V phase: the softmax of query block A's scores of key block jb (40 VALU) + the fragments of the next M phase (K of the next
    key block, V^T of this one)."""
    sn = (st + 1) & 3
    kst, kjb = (st, 1) if jb == 0 else (sn, 0)
    if jb == 0:
        pstamp(a, 2)
    for k in range(4):
        a.e("ds_read_b128 %s, %s offset:%d" % (vr(KF(0, k), 4), v(KOFF(k)), kst * STAGE + kjb * 4096))
    for f in range(4):
        a.e("ds_read_b128 %s, %s offset:%d" % (vr(VF(0, f), 4), v(VOFF(jb, f >> 1)), st * STAGE + (f & 1) * 4096))
    if body and jb == 1:
        a.e("ds_read_b32 %s, %s offset:%d" % (v(V_KACUR), v(V_KAREAD), sn * 1024))
    if OPT["prio"] and OPT["rdprio"] and body:
        a.e("s_setprio 0")                          # measured alternative: the next M phase's fragment reads are still issued at the M phase's priority
    extra = []
    if body and OPT["dma_v"]:
        rd = dma_round((st + 2) & 3)
        extra += (rd[0] + rd[1] + rd[4]) if jb == 0 else (rd[2] + rd[3])
    if body and jb == 1 and OPT["adv_v"]:
        extra += advance_round()
    grps = softmax_gaps(0)
    if OPT["pure"]:
        gb = softmax_gaps(1)
        grps = (gb + grps) if OPT["pure"] == 2 else (grps + gb)
    for i, grp in enumerate(grps):
        if grp and grp[0].startswith("v_mfma_f32_4x4x4") and len(grp) == 1:
            a.e("s_nop 1")                          # VALU write -> MFMA read: two wait states behind the last pack
        for t in grp:
            a.e(t)
        # the scalar / DMA extras between the VALU groups
        if i < 8:
            for t in extra[i * len(extra) // 8:(i + 1) * len(extra) // 8]:
                a.e(t)
    if body and jb == 1:
        a.e("s_waitcnt lgkmcnt(0)")
        a.e("v_cmp_neq_f32 vcc, 0, %s" % v(V_KACUR))
        a.e("s_cmp_lg_u64 vcc, 0")
        a.e("s_cselect_b32 %s, 1, 0" % s("mnext"))
    pstamp(a, 3 if jb == 0 else 7)


def tile_body(a, st, masked):
    if masked:
        masked_tile_entry(a, st)
    for jb in range(2):
        phase_M(a, st, jb, masked)
        phase_V(a, st, jb, masked)
    a.e("s_add_u32 %s, %s, 1" % (s("t"), s("t")))
    a.e("s_add_u32 %s, %s, 1" % (s("x0"), s("t")))
    a.e("s_cmp_eq_u32 %s, %s" % (s("x0"), s("nt")))
    a.e("s_cselect_b32 %s, %s, 0" % (s("x0"), s("tailf")))
    a.e("s_or_b32 %s, %s, %s" % (s("masked"), s("mnext"), s("x0")))


def item_decode(a, done_label, skip_label):
    """virtual block v -> (qb, h, b) with attn_decode_block's XCD map: L = (v & 7) * cnt + (v >> 3)"""
    a.e("s_cmp_ge_u32 %s, %s" % (s("v"), s("vend")))
    a.e("s_cbranch_scc1 %s" % a.ref(done_label))
    a.e("s_and_b32 %s, %s, 7" % (s("x0"), s("v")))
    a.e("s_lshr_b32 %s, %s, 3" % (s("x1"), s("v")))
    a.e("s_mul_i32 %s, %s, %s" % (s("x0"), s("x0"), s("cnt")))
    a.e("s_add_u32 %s, %s, %s" % (s("x0"), s("x0"), s("x1")))                         # L
    a.e("s_cmp_ge_u32 %s, %s" % (s("x0"), s("total")))
    a.e("s_cbranch_scc1 %s" % a.ref(skip_label))
    a.e("s_mul_hi_u32 %s, %s, %s" % (s("x1"), s("x0"), s("mq")))                      # L / nqb (magic multiply; a divisor of 1 has none)
    a.e("s_cmp_eq_u32 %s, 1" % s("nqb"))
    a.e("s_cselect_b32 %s, %s, %s" % (s("x1"), s("x0"), s("x1")))
    a.e("s_mul_i32 %s, %s, %s" % (s("x2"), s("x1"), s("nqb")))
    a.e("s_sub_u32 %s, %s, %s" % (s("qb"), s("x0"), s("x2")))
    a.e("s_mul_hi_u32 %s, %s, %s" % (s("b"), s("x1"), s("mh")))                       # (L / nqb) / H
    a.e("s_cmp_eq_u32 %s, 1" % s("H"))
    a.e("s_cselect_b32 %s, %s, %s" % (s("b"), s("x1"), s("b")))
    a.e("s_mul_i32 %s, %s, %s" % (s("x2"), s("b"), s("H")))
    a.e("s_sub_u32 %s, %s, %s" % (s("h"), s("x1"), s("x2")))
    # byte offset of head (b, h) in q / k / v^T: bh * Npad * 128
    a.e("s_lshl_b32 %s, %s, 7" % (s("x2"), s("Npad")))
    a.e("s_mul_hi_u32 %s, %s, %s" % (s("x3"), s("x1"), s("x2")))
    a.e("s_mul_i32 %s, %s, %s" % (s("x2"), s("x1"), s("x2")))
    for n, base in (("Qp", "q"), ("Kp", "k"), ("Vp", "vt")):
        a.e("s_add_u32 %s, %s, %s" % (s(n), s(base), s("x2")))
        a.e("s_addc_u32 %s, %s, %s" % (shi(n), shi(base), s("x3")))
    a.e("s_mul_i32 %s, %s, %s" % (s("x2"), s("b"), s("kas")))                         # key_add row of sample b (kas in bytes)
    a.e("s_mul_hi_u32 %s, %s, %s" % (s("x3"), s("b"), s("kas")))
    a.e("s_add_u32 %s, %s, %s" % (s("Ap"), s("ka"), s("x2")))
    a.e("s_addc_u32 %s, %s, %s" % (shi("Ap"), shi("ka"), s("x3")))
    # o + ((b * N) * H * 64 + h * 64) * 2
    a.e("s_mul_i32 %s, %s, %s" % (s("x2"), s("b"), s("N")))
    a.e("s_lshl_b32 %s, %s, 7" % (s("x4"), s("H")))                                   # bytes of one output row
    a.e("s_mul_hi_u32 %s, %s, %s" % (s("x3"), s("x2"), s("x4")))
    a.e("s_mul_i32 %s, %s, %s" % (s("x2"), s("x2"), s("x4")))
    a.e("s_lshl_b32 %s, %s, 7" % (s("x5"), s("h")))
    a.e("s_add_u32 %s, %s, %s" % (s("x2"), s("x2"), s("x5")))
    a.e("s_addc_u32 %s, %s, 0" % (s("x3"), s("x3")))
    a.e("s_add_u32 %s, %s, %s" % (s("Op"), s("o"), s("x2")))
    a.e("s_addc_u32 %s, %s, %s" % (shi("Op"), shi("o"), s("x3")))
    # q0 = (4 * qb + wave) * 64; the wave is active when q0 < N
    a.e("s_lshl_b32 %s, %s, 2" % (s("q0"), s("qb")))
    a.e("s_add_u32 %s, %s, %s" % (s("q0"), s("q0"), s("wave")))
    a.e("s_lshl_b32 %s, %s, 6" % (s("q0"), s("q0")))
    a.e("s_cmp_lt_u32 %s, %s" % (s("q0"), s("N")))
    a.e("s_cselect_b32 %s, 1, 0" % s("active"))


def lane_constants(a):
    lane, m31, half, t0, t1 = T[0], T[1], T[2], T[3], T[4]
    a.e("v_mbcnt_lo_u32_b32 %s, -1, 0" % v(lane))
    a.e("v_mbcnt_hi_u32_b32 %s, -1, %s" % (v(lane), v(lane)))
    a.e("v_and_b32 %s, 31, %s" % (v(m31), v(lane)))
    a.e("v_lshrrev_b32 %s, 5, %s" % (v(half), v(lane)))
    a.e("v_lshlrev_b32 %s, 3, %s" % (v(V_HALF8), v(half)))
    # kperm = (m31 & 0x13) | ((m31 & 4) << 1) | ((m31 & 8) >> 1)
    a.e("v_and_b32 %s, 0x13, %s" % (v(t0), v(m31)))
    a.e("v_and_b32 %s, 4, %s" % (v(t1), v(m31)))
    a.e("v_lshl_or_b32 %s, %s, 1, %s" % (v(t0), v(t1), v(t0)))
    a.e("v_and_b32 %s, 8, %s" % (v(t1), v(m31)))
    a.e("v_lshrrev_b32 %s, 1, %s" % (v(t1), v(t1)))
    a.e("v_or_b32 %s, %s, %s" % (v(t0), v(t0), v(t1)))                    # kperm
    # swz128(row, chunk) = row * 128 + ((chunk ^ ((row >> 1) & 7)) << 4)
    def swz(dst, row, chunk_expr_emit):
        a.e("v_lshrrev_b32 %s, 1, %s" % (v(T[5]), v(row)))
        a.e("v_and_b32 %s, 7, %s" % (v(T[5]), v(T[5])))
        chunk_expr_emit(T[6])                                             # chunk -> T[6]
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
    # DMA source offsets: piece = wave + 4 i, row = 8 piece + (lane >> 3), chunk = (lane & 7) ^ ((row >> 1) & 7)
    a.e("s_lshl_b32 %s, %s, 3" % (s("x0"), s("wave")))
    a.e("v_lshrrev_b32 %s, 3, %s" % (v(t0), v(lane)))
    a.e("v_add_u32 %s, %s, %s" % (v(t0), s("x0"), v(t0)))                 # row of piece 0
    a.e("s_lshl_b32 %s, %s, 1" % (s("x1"), s("Npad")))                    # bytes of a V^T row
    for i in range(2):
        if i == 1:
            a.e("v_add_u32 %s, 32, %s" % (v(t0), v(t0)))
        a.e("v_lshrrev_b32 %s, 1, %s" % (v(T[5]), v(t0)))
        a.e("v_and_b32 %s, 7, %s" % (v(T[5]), v(T[5])))
        a.e("v_and_b32 %s, 7, %s" % (v(T[6]), v(lane)))
        a.e("v_xor_b32 %s, %s, %s" % (v(T[5]), v(T[5]), v(T[6])))
        a.e("v_lshlrev_b32 %s, 4, %s" % (v(T[5]), v(T[5])))               # chunk * 16
        a.e("v_lshl_add_u32 %s, %s, 7, %s" % (v(DK(i)), v(t0), v(T[5])))
        a.e("v_mul_lo_u32 %s, %s, %s" % (v(T[6]), v(t0), s("x1")))
        a.e("v_add_u32 %s, %s, %s" % (v(DV(i)), v(T[6]), v(T[5])))
    a.e("v_lshlrev_b32 %s, 2, %s" % (v(V_DKA), v(lane)))
    a.e("s_lshl_b32 %s, %s, 8" % (s("x0"), s("wave")))
    a.e("s_add_u32 %s, %s, %s" % (s("wa"), s("lds"), s("x0")))
    a.e("s_add_u32 %s, %s, %d" % (s("wa"), s("wa"), KADD0))               # this wave's key_add row of stage 0
    a.e("v_add_u32 %s, %s, %s" % (v(V_KAREAD), s("wa"), v(V_DKA)))
    a.e("v_lshlrev_b32 %s, 5, %s" % (v(t1), v(half)))
    a.e("v_add_u32 %s, %s, %s" % (v(V_KAADDR), s("wa"), v(t1)))
    a.e("s_lshl_b32 %s, %s, 10" % (s("x0"), s("wave")))
    a.e("s_add_u32 %s, %s, %s" % (s("wl"), s("lds"), s("x0")))            # this wave's piece 0 of stage 0
    a.e("v_mov_b32 %s, 0xff800000" % v(V_NEGINF))
    for i in range(2):
        a.e("v_mov_b32 %s, 0x3f803f80" % v(V_ONES + i))                   # bf16 (1, 1)
    for f in range(4):                                                    # "tile -1" runs P V against P = 0: its V^T fragments must be finite
        for r in range(4):
            a.e("v_mov_b32 %s, 0" % v(VF(1, f) + r))


def item_prologue_active(a):
    m31, t0 = T[1], T[3]
    # q rows (clamped to N - 1), 16 bytes per lane and k step
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
        a.e("v_lshl_add_u32 %s, %s, 1, %s" % (v(QOFF(x)), v(V_HALF8), v(t0)))          # + half * 16
        for kk in range(4):
            a.e("global_load_dwordx4 %s, %s, %s offset:%d" % (vr(Q(x, kk), 4), v(QOFF(x)), s2("Qp"), 32 * kk))


def emit_rounds_01(a):
    for r in range(2):
        for pair in dma_round(r):
            for t in pair:
                a.e(t)
        for t in advance_round():
            a.e(t)


def epilogue(a):
    """normalise, range check (2^-100 < row sum < 2^100 else the item is redone exactly), 16-byte stores of rows < N"""
    l0, l1, lane, m31, row = T[0], T[1], T[2], T[3], T[4]
    a.e("s_mov_b32 %s, 0" % s("ibad"))
    a.e("v_mbcnt_lo_u32_b32 %s, -1, 0" % v(lane))
    a.e("v_mbcnt_hi_u32_b32 %s, -1, %s" % (v(lane), v(lane)))
    a.e("v_and_b32 %s, 31, %s" % (v(m31), v(lane)))
    a.e("s_lshl_b32 %s, %s, 7" % (s("x4"), s("H")))
    buf = 96                                                              # v96.. (scores, P, fragments: all dead) hold the packed rows
    for x in range(2):
        a.e("v_add_f32 %s, %s, %s" % (v(l0), v(PS(x, 0)), v(PS(x, 1))))          # (one of the two forms stays zero)
        a.e("v_add_f32 %s, %s, %s" % (v(l0), v(l0), v(PSQ(x))))
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
        a.e("v_lshl_add_u32 %s, %s, 1, %s" % (v(row), v(V_HALF8), v(row)))            # + half * 16 bytes
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
            a.e("global_store_dwordx4 %s, %s, %s offset:%d%s" % (v(row), vr(base, 4), s2("Op"), off, " sc1" if OPT["osc1"] else ""))
        a.e("s_mov_b64 exec, %s" % s2("x2"))


TRACE = False


def stamp(a, k):
    """trace build only (tools/attn_wgtrace.py): wave 0 stores the constant-clock time into slot k of its item's record"""
    if not TRACE:
        return
    skip = "ts%d_%d" % (k, a.nlabel)
    a.nlabel += 1
    a.e("s_cmp_lg_u32 %s, 0" % s("wave"))
    a.e("s_cbranch_scc1 %s" % a.ref(skip))
    a.e("s_memrealtime s[98:99]")
    a.e("s_mul_i32 %s, %s, 48" % (s("x5"), s("v")))
    a.e("s_add_u32 %s, %s, %d" % (s("x5"), s("x5"), 8 * k))
    a.e("s_waitcnt lgkmcnt(0)")
    a.e("s_store_dwordx2 s[98:99], s[100:101], %s" % s("x5"))
    if k == 3:
        a.e("s_getreg_b32 s98, hwreg(HW_REG_HW_ID)")
        a.e("s_getreg_b32 s99, hwreg(HW_REG_XCC_ID)")
        a.e("s_add_u32 %s, %s, 8" % (s("x5"), s("x5")))
        a.e("s_store_dwordx2 s[98:99], s[100:101], %s" % s("x5"))
    a.e("s_waitcnt lgkmcnt(0)")
    a.label(skip)


def generate(trace=False):
    global TRACE
    TRACE = trace
    a = Asm()
    if trace:
        a.e("s_mov_b64 s[100:101], %[tr]")
    else:
        # the block writes M0 for its LDS-DMA; hipcc refuses M0 in a clobber list ("reserved register"), so the block hands back the value it
        # found -- the compiler-scheduled code behind it (attn_w64_item, the exact pass) issues LDS-DMA through the builtin
        a.e("s_mov_b32 s38, m0")
    # ---- operands -> fixed registers
    for n in ("q", "k", "vt", "ka", "o"):
        a.e("s_mov_b64 %s, %%[%s]" % (s2(n), n))
    for n in ("N", "Npad", "H", "total", "cnt", "v", "G", "kas", "wave", "lds", "nqb", "mq", "mh"):
        a.e("s_mov_b32 %s, %%[%s]" % (s(n), n))
    a.e("s_lshl_b32 %s, %s, 3" % (s("vend"), s("cnt")))
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

    # ================================================================ item loop
    a.label("item")
    item_decode(a, "done", "next")
    # flags of the previous item (every wave wrote its own behind its epilogue), behind the barrier that also frees the ring
    a.e("s_barrier")
    a.e("s_cmp_eq_u32 %s, 0" % s("it"))
    a.e("s_cbranch_scc1 %s" % a.ref("noflag"))
    a.e("v_mov_b32 %s, %s" % (v(T[0]), s("lds")))
    a.e("v_add_u32 %s, 0x%x, %s" % (v(T[0]), FLAG0, v(T[0])))
    a.e("ds_read_b128 %s, %s" % (vr(T[2], 4), v(T[0])))
    a.e("s_waitcnt lgkmcnt(0)")
    a.e("v_or_b32 %s, %s, %s" % (v(T[2]), v(T[2]), v(T[3])))
    a.e("v_or3_b32 %s, %s, %s, %s" % (v(T[2]), v(T[2]), v(T[4]), v(T[5])))
    a.e("v_readfirstlane_b32 %s, %s" % (s("x0"), v(T[2])))
    a.e("s_cmp_eq_u32 %s, 0" % s("x0"))
    a.e("s_cbranch_scc1 %s" % a.ref("noflag"))
    a.e("s_sub_u32 %s, %s, 1" % (s("x0"), s("it")))
    a.e("s_bitset1_b64 %s, %s" % (s2("bad"), s("x0")))
    a.label("noflag")
    a.e("s_barrier")                                                      # every wave has read the flags before anyone rewrites them
    stamp(a, 0)
    if trace:
        a.e("s_mov_b32 %s, 0" % s("ptr"))
        a.e("s_or_b32 %s, %s, %s" % (s("x0"), s("wave"), s("it")))
        a.e("s_cmp_lg_u32 %s, 0" % s("x0"))
        a.e("s_cbranch_scc1 %s" % a.ref("noptr"))
        a.e("s_cmp_eq_u32 %s, 0" % s("v"))
        a.e("s_cselect_b32 %s, 300000, 0" % s("ptr"))
        a.e("s_cmp_eq_u32 %s, 256" % s("v"))
        a.e("s_cselect_b32 %s, 304096, %s" % (s("ptr"), s("ptr")))
        a.label("noptr")
    a.e("s_mov_b32 %s, 0" % s("r"))
    a.e("s_mov_b32 %s, 0" % s("t"))
    a.e("s_cmp_eq_u32 %s, 0" % s("active"))
    a.e("s_cbranch_scc1 %s" % a.ref("light"))

    # ---------------------------------------------------------------- active wave
    item_prologue_active(a)
    emit_rounds_01(a)
    for x in range(2):
        for db in range(2):
            for r in range(16):
                a.e("v_mov_b32 %s, 0" % v(O(x, db, r)))
        for i in range(4):
            a.e("v_mov_b32 %s, 0" % v(PSQ(x) + i))
        for i in range(2):
            a.e("v_mov_b32 %s, 0" % v(PS(x, i)))
        for t2 in range(2):
            for r in range(4):
                a.e("v_mov_b32 %s, 0" % v(P(x, t2) + r))
    for f in range(4):                                                    # "key block -1": P V against P = 0 needs finite V^T fragments
        for r in range(4):
            a.e("v_mov_b32 %s, 0" % v(VF(0, f) + r))
    for r in range(16):                                                   # and block B's "pending scores" exponentiate to 0
        a.e("v_mov_b32 %s, 0xff800000" % v(S(1, r)))
    a.e("s_waitcnt vmcnt(5)")                                             # q and round 0 have landed; round 1 flies
    a.e("s_barrier")
    for k in range(4):                                                    # K fragments of (tile 0, key block 0); mask flag of tile 0
        a.e("ds_read_b128 %s, %s offset:%d" % (vr(KF(0, k), 4), v(KOFF(k)), 0))
    a.e("ds_read_b32 %s, %s offset:0" % (v(V_KACUR), v(V_KAREAD)))
    a.e("s_waitcnt lgkmcnt(0)")
    a.e("v_cmp_neq_f32 vcc, 0, %s" % v(V_KACUR))
    a.e("s_cmp_lg_u64 vcc, 0")
    a.e("s_cselect_b32 %s, 1, 0" % s("masked"))
    stamp(a, 1)
    # ---- tile loop, unrolled by the ring depth (a stage is an immediate offset)
    a.in_loop = True
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
    # ---- drain: block B's last scores are still pending, P V of the last key block
    a.in_loop = False
    a.label("drain")
    stamp(a, 2)
    phase_M(a, 0, 0, False, qk=False, body=False)
    a.e("s_nop 15")
    a.e("s_nop 3")
    epilogue(a)
    stamp(a, 5)
    a.e("s_branch %s" % a.ref("itemend"))

    # ---------------------------------------------------------------- wave without queries: DMA, tail fix and barriers only
    a.label("light")
    emit_rounds_01(a)
    a.e("s_waitcnt vmcnt(5)")
    a.e("s_barrier")
    a.e("s_mov_b32 %s, 0" % s("ibad"))
    a.label("ltile0")
    for st in range(4):
        if st:
            a.label("ltile%d" % st)
        a.e("s_cmp_ge_u32 %s, %s" % (s("t"), s("nt")))
        a.e("s_cbranch_scc1 %s" % a.ref("itemend"))
        for pair in dma_round((st + 2) & 3):
            for t in pair:
                a.e(t)
        for t in advance_round():
            a.e(t)
        a.e("s_waitcnt vmcnt(5)")
        tail_fix(a, (st + 1) & 3, "%dL" % st)
        if "nobar" not in ABL:
            a.e("s_barrier")
        a.e("s_add_u32 %s, %s, 1" % (s("t"), s("t")))
        if st == 3:
            a.e("s_branch %s" % a.ref("ltile0"))

    # ---------------------------------------------------------------- item end: flag, drain the surplus rounds and the stores
    a.label("itemend")
    a.e("v_mov_b32 %s, %s" % (v(T[0]), s("wave")))
    a.e("v_lshl_add_u32 %s, %s, 2, %s" % (v(T[0]), v(T[0]), s("lds")))
    a.e("v_add_u32 %s, 0x%x, %s" % (v(T[0]), FLAG0, v(T[0])))
    a.e("v_mov_b32 %s, %s" % (v(T[1]), s("ibad")))
    a.e("ds_write_b32 %s, %s" % (v(T[0]), v(T[1])))
    a.e("s_waitcnt vmcnt(0) lgkmcnt(0)")
    stamp(a, 3)
    a.e("s_add_u32 %s, %s, 1" % (s("it"), s("it")))
    a.label("next")
    a.e("s_add_u32 %s, %s, %s" % (s("v"), s("v"), s("G")))
    a.e("s_branch %s" % a.ref("item"))

    # ================================================================ after the last item: its flags
    a.label("done")
    a.e("s_barrier")
    a.e("s_cmp_eq_u32 %s, 0" % s("it"))
    a.e("s_cbranch_scc1 %s" % a.ref("out"))
    a.e("v_mov_b32 %s, %s" % (v(T[0]), s("lds")))
    a.e("v_add_u32 %s, 0x%x, %s" % (v(T[0]), FLAG0, v(T[0])))
    a.e("ds_read_b128 %s, %s" % (vr(T[2], 4), v(T[0])))
    a.e("s_waitcnt lgkmcnt(0)")
    a.e("v_or_b32 %s, %s, %s" % (v(T[2]), v(T[2]), v(T[3])))
    a.e("v_or3_b32 %s, %s, %s, %s" % (v(T[2]), v(T[2]), v(T[4]), v(T[5])))
    a.e("v_readfirstlane_b32 %s, %s" % (s("x0"), v(T[2])))
    a.e("s_cmp_eq_u32 %s, 0" % s("x0"))
    a.e("s_cbranch_scc1 %s" % a.ref("out"))
    a.e("s_sub_u32 %s, %s, 1" % (s("x0"), s("it")))
    a.e("s_bitset1_b64 %s, %s" % (s2("bad"), s("x0")))
    a.label("out")
    a.e("s_barrier")
    if trace:
        a.e("s_dcache_wb")
    else:
        a.e("s_mov_b32 m0, s38")
    a.e("s_mov_b64 %[bad], " + s2("bad"))
    return a.lines


def main(out=OUT, trace=False):
    lines = generate(trace)
    with open(out, "w") as f:
        f.write("// GENERATED by tools/gen/attn_p64_gen.py -- do not edit.  %d instructions / labels.\n" % len(lines))
        f.write("// LDS image: %d stages x %d B, key_add rows at %d, flags at %d, %d bytes in all.\n" % (NS, STAGE, KADD0, FLAG0, LDS_BYTES))
        for ln in lines:
            if ln.startswith(";"):
                continue
            f.write('"%s\\n\\t"\n' % ln)
    print("wrote", out, len(lines), "lines")


# ---------------------------------------------------------------- hazard lint (the assembler inserts no wait states)
def _regs(tok):
    import re
    m = re.fullmatch(r"v\[(\d+):(\d+)\]", tok)
    if m:
        return set(range(int(m.group(1)), int(m.group(2)) + 1))
    m = re.fullmatch(r"v(\d+)", tok)
    return {int(m.group(1))} if m else set()


def lint(lines):
    """MFMA result -> any other access of those registers needs 12 wait states (measured on hipcc's own output for gfx950:
    s_nop 11 between v_mfma_f32_32x32x16_bf16 and a VALU / memory read of its result, or a later MFMA reading it as A / B; back-to-back
    accumulation into the same registers needs none); VALU write -> MFMA A / B read: 2.  Straight-line approximation: labels and
    branches are ignored, which is conservative enough for this stream (every block starts and ends far from an MFMA's result use)."""
    import re
    recent = []          # (age in wait states, kind, written regs)
    problems = []
    for n, ln in enumerate(lines):
        if ln.endswith(":") or ln.startswith(";"):
            continue
        op, _, rest = ln.partition(" ")
        toks = [t.strip() for t in re.split(r",|\s+offset:\d+", rest) if t and t.strip()]
        states = 1
        if op == "s_nop":
            states = int(toks[0]) + 1
        if op.startswith("v_mfma"):
            dst, a_, b_, c_ = _regs(toks[0]), _regs(toks[1]), _regs(toks[2]), _regs(toks[3])
            for age, kind, wr in recent:
                if kind == "mfma" and age < 12 and (wr & (a_ | b_)):
                    problems.append((n, ln, "MFMA result read as A/B after %d states" % age))
                if kind == "mfma" and age < 12 and (wr & (dst | c_)) and not (wr == dst and (c_ == dst or not c_)):
                    problems.append((n, ln, "MFMA overlapping accumulate after %d states" % age))
                if kind == "valu" and age < 2 and (wr & (a_ | b_ | c_)):
                    problems.append((n, ln, "VALU result read by MFMA after %d states" % age))
            recent = [(a + states, k, w) for a, k, w in recent if a + states < 24] + [(0, "mfma", dst)]
            continue
        used = set()
        for t in toks:
            used |= _regs(t)
        if op.startswith(("v_", "ds_", "global_")):
            for age, kind, wr in recent:
                if kind == "mfma" and age < 12 and (wr & used):
                    problems.append((n, ln, "MFMA result touched after %d states" % age))
        wrote = set()
        if op.startswith("v_") and toks and not op.startswith("v_cmp"):
            wrote = _regs(toks[0])
            if op.startswith("v_permlane32_swap"):
                wrote |= _regs(toks[1])
        recent = [(a + states, k, w) for a, k, w in recent if a + states < 24]
        if wrote:
            recent.append((0, "valu", wrote))
    return problems


if __name__ == "__main__":
    import sys
    if "--trace" in sys.argv:                 # the stamped variant for tools/attn_wgtrace.py (not committed)
        main(sys.argv[sys.argv.index("--trace") + 1], True)
    else:
        main()
    if "--lint" in sys.argv:
        probs = lint(generate())
        for n, ln, why in probs[:40]:
            print("line %d: %s  <- %s" % (n, ln, why))
        print("%d hazard findings" % len(probs))
