#!/usr/bin/env python3
"""Generator of uvltrack_amd/csrc/gemm_dr_asm.inc: the K loop of gemm_dr_kernel -- a 128 x 256 output tile on FOUR waves, TWO such
workgroups per CU -- as ONE inline-asm statement that owns v0..v(NV-1), s40..s59 and names the accumulators a0..a127 directly (the C++
side pins its eight f32x16 operands to a[0:15] ... a[112:127]).

Why this shape (profiles/r04_gemm_streamk.md, r04_gemm_dr.md): at K = 1024 a third of a batched GEMM is what a tile pays OUTSIDE its K
loop (cold prologue, LDS staging, the output burst, GELU / read-modify-write arithmetic: ~10 us per round of tiles, 15-30 with the
frame's epilogues), and with one workgroup per CU nothing runs under it.  Two workgroups per CU drift apart and hide each other's
prologue and epilogue -- but only a wave tile of 128 x 64 keeps the fragment traffic per MFMA of the one-workgroup kernels, and two
workgroups of such waves need 2 x (128 + 256) rows of LDS per K tile.  So only the operand the four waves SHARE goes through LDS:
  * wave w owns all 128 rows x columns [64 w, 64 w + 64) of the tile: 8 x 4 blocks of v_mfma_f32_16x16x32_bf16 = 128 AGPRs;
  * A (activations, 128 rows, shared by the four waves): HBM -> LDS by LDS-DMA, 16 KB per 64-wide K tile, four stages;
  * W (weights): the wave's 64 rows are its own -- each lane loads its MFMA fragment straight from global memory into VGPRs
    (global_load_dwordx4: 16 rows x 64 contiguous bytes per instruction), one K tile ahead.  No LDS write, no ds_read, no LDS-DMA issue
    (60-185 cycles apiece) for two thirds of the operand bytes.

Per K tile and wave: 64 MFMAs (2 k steps of 32), 16 fragment reads (A), 8 global loads (W of tile t + 1), 4 LDS-DMA (A of tile t + 3),
ONE barrier.  Schedule of tile t (W set p = t & 1, A stage t % 4):

    k step 0   for A block ib = 0..7: 4 MFMAs on W(p, 0, 0..3); behind them the read of A(t, k step 1, ib) into the slot just used;
               one W load of tile t + 1 (set 1 - p) per block
    k step 1   s_barrier B(t + 1) -- every wave's A pieces of tile t + 1 have landed (each wave waited for ITS pieces at the end of tile
               t - 1: they are older than the W loads it waited for) and nobody reads stage (t - 1) % 4 any more --
               then per block: 4 MFMAs on W(p, 1, .), the read of A(t + 1, k step 0, ib); the 4 LDS-DMA pieces of tile t + 3 into
               stage (t + 3) % 4 = (t - 1) % 4, M0 write and DMA behind different MFMAs
    end        s_waitcnt vmcnt(4): W(t + 1) has arrived (the 4 pieces of A(t + 3) stay in flight); addresses of the next tile

  * tile indices beyond the last are clamped (the last tile is fetched again into registers / a stage nobody reads): no tail form; the
    loop is unrolled over the four stages with an exit test per tile;
  * lgkmcnt / vmcnt are counted by the generator from its own issue order.

LDS image of an A stage: 128 rows of 128 bytes, 16-byte chunk c of row r at position c ^ ((r >> 1) & 7) -- applied to the per-lane SOURCE
address of the DMA (its LDS side is lane-linear) and again on the fragment read.  A fragment of 16-row block ib for k step s: lane l
reads row 16 ib + (l & 15), chunk 2 (l >> 4) + s (the K index inside a tile is permuted, the same way for both operands).
W comes from a FRAGMENT-NATIVE copy of the weight (gemm_dr.hip::pack_w_dr_kernel, made once per weight): the 16 bytes lane l feeds to the
MFMA for (16-row block nb, K tile kt, k step s) sit at ((nb K/64 + kt) 2 + s) 1024 + 16 l, so a load instruction of a wave reads 1 KB of
consecutive addresses.  From the nn.Linear layout [N, K] the same load is 64 separate 16-byte requests (lanes l & 15 are sixteen different
rows): first build, 64.2 us on the QKV shape of 8 UVLTrack-L sequences against 37.9 us with the W loads removed -- and 66 us with every K
tile loading the SAME bytes, i.e. request count, not bandwidth or misses.
Accumulator block (ib, jb) = a[4 (4 ib + jb) .. + 3] = mfma(W fragment jb, A fragment ib): lane l
holds output row 16 ib + (l & 15), columns 64 w + 16 jb + 4 (l >> 4) + r.

Registers: W fragments v0..v63 (set p, k step s, block jb: 32 p + 16 s + 4 jb), A fragment slots v64..v95 (block ib: 64 + 4 ib), W load
offsets v96..v99, A DMA source offsets v100..v103, A fragment addresses v104..v105 (k step), temporaries v106..v113.
s[40:41] / s[42:43] A / W address of the tile being requested, s44 tiles left, s45 A row pitch in bytes, s46 bytes of a 16-row block of the packed W (K / 64 x 2048), s47 last valid A row of
the tile, s48 LDS base, s49 index of the A tile requested next, s50 nk - 1, s51 scratch, s52 index of the W tile requested next,
s53 LDS base + 1024 * wave, s54 the M0 the block found (restored at its end), s[56:57] / s[58:59] A / W address of tile 0.

Usage: python tools/gen/gemm_dr_gen.py [--check]            (writes / compares uvltrack_amd/csrc/gemm_dr_asm.inc)
       python tools/gen/gemm_dr_gen.py --out PATH [--abl nodma,noread,nobar,nowload]   (timing variants, results wrong)"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
OUT = os.path.join(ROOT, "uvltrack_amd", "csrc", "gemm_dr_asm.inc")

NSTG = 4                           # A stages
STAGE = 16384                      # one A stage: 128 rows of 128 bytes


def WF(p, ks, jb):
    return 32 * p + 16 * ks + 4 * jb


def AF(ib):
    return 64 + 4 * ib


WOFF = lambda jb: 96 + jb          # per-lane byte offset of W block jb's row (+ 16 (lane >> 4))
AOFF = lambda q: 100 + q           # per-lane source byte offset of A piece q
RA = lambda ks: 104 + ks           # LDS address of this lane's A fragment bytes for k step ks (stage 0, block 0)
T = lambda k: 106 + k
NV = 114                           # first VGPR the block does not touch


def v4(r):
    return "v[%d:%d]" % (r, r + 3)


class Gen:
    def __init__(self, abl=()):
        self.abl = set(abl)
        self.out = []
        self.ldsq = []             # tags of the LDS reads in flight, oldest first
        self.vmq = []              # tags of the vector-memory loads in flight (W loads and LDS-DMA retire in order), oldest first
        self.in_loop = False

    def e(self, s):
        self.out.append(s)

    # ---- LDS reads of A fragments
    def a_read(self, ib, ks, stage, tag):
        if "noread" in self.abl and self.in_loop:
            return
        self.e("ds_read_b128 %s, v%d offset:%d" % (v4(AF(ib)), RA(ks), stage * STAGE + ib * 2048))
        self.ldsq.append(tag)

    def need_lds(self, tag):
        if tag not in self.ldsq:
            return
        k = max(i for i, t in enumerate(self.ldsq) if t == tag)
        self.e("s_waitcnt lgkmcnt(%d)" % min(len(self.ldsq) - 1 - k, 15))
        self.ldsq = self.ldsq[k + 1:]

    # ---- vector memory
    def a_m0(self, stage, q):
        if "nodma" in self.abl and self.in_loop:
            return
        self.e("s_add_u32 m0, s53, %d" % (stage * STAGE + q * 4096))

    def a_go(self, q, tag, nop=False):
        if "nodma" in self.abl and self.in_loop:
            return
        if nop:
            self.e("s_nop 0")
        self.e("global_load_lds_dwordx4 v%d, s[40:41]" % AOFF(q))
        self.vmq.append(tag)

    def w_load(self, p, ks, jb, tag):
        if "nowload" in self.abl and self.in_loop:
            return
        if "whalf" in self.abl and self.in_loop and ks == 1:       # timing probe: half of the W loads
            return
        self.e("global_load_dwordx4 %s, v%d, s[42:43]%s" % (v4(WF(p, ks, jb)), WOFF(jb), " offset:1024" if ks else ""))
        self.vmq.append(tag)

    def need_vm(self, tag):
        """every load tagged `tag` (and everything older) has arrived"""
        if tag not in self.vmq:
            return
        k = max(i for i, t in enumerate(self.vmq) if t == tag)
        self.e("s_waitcnt vmcnt(%d)" % (len(self.vmq) - 1 - k))
        self.vmq = self.vmq[k + 1:]

    def mfma(self, p, ks, ib, jb):
        a0 = 4 * (4 * ib + jb)
        self.e("v_mfma_f32_16x16x32_bf16 a[%d:%d], %s, %s, a[%d:%d]" % (a0, a0 + 3, v4(WF(p, ks, jb)), v4(AF(ib)), a0, a0 + 3))

    def next_a_address(self):
        # s49 = min(s49 + 1, nk - 1); s[40:41] = A tile 0 + 128 s49
        self.e("s_add_u32 s49, s49, 1")
        self.e("s_min_u32 s49, s49, s50")
        self.e("s_lshl_b32 s51, s49, 7")
        self.e("s_add_u32 s40, s56, s51")
        self.e("s_addc_u32 s41, s57, 0")

    def next_w_address(self):
        if "wsame" in self.abl and self.in_loop:       # timing probe: every K tile loads the SAME W bytes (L1 / L2 hot)
            return
        self.e("s_add_u32 s52, s52, 1")
        self.e("s_min_u32 s52, s52, s50")
        self.e("s_lshl_b32 s51, s52, 11")
        self.e("s_add_u32 s42, s58, s51")
        self.e("s_addc_u32 s43, s59, 0")

    def iteration(self, u):
        """tile t with t % 4 == u: W set u & 1, A stage u.  On entry: W(t) in set p, A(t, k step 0) requested into the slots, s[42:43]
        at W tile t + 1, s[40:41] at A tile t + 3."""
        p, st = u & 1, u
        # ---- k step 0
        for ib in range(8):
            self.need_lds(("a", 0, ib))
            for jb in range(4):
                self.mfma(p, 0, ib, jb)
                if "wearly" in self.abl:                                   # variant: the eight W loads behind the first eight MFMAs
                    if ib < 2:
                        self.w_load(1 - p, (4 * ib + jb) // 4, (4 * ib + jb) % 4, "w")
                elif jb == 1 and ib % 2 == 0:                              # W(t + 1), block ib // 2, both k steps (2 x 1 KB of consecutive addresses)
                    self.w_load(1 - p, 0, ib // 2, "w")
                    self.w_load(1 - p, 1, ib // 2, "w")
                if jb == 3:
                    self.a_read(ib, 1, st, ("a", 1, ib))                   # A(t, k step 1, ib) into the slot the four MFMAs above have read
        self.next_w_address()
        # ---- k step 1
        if "nobar" not in self.abl:
            self.e("s_barrier")                                            # B(t + 1)
        for ib in range(8):
            self.need_lds(("a", 1, ib))
            for jb in range(4):
                self.mfma(p, 1, ib, jb)
                if ib < 4 and jb == 0:
                    self.a_m0((st + 3) % NSTG, ib)                         # A(t + 3), piece ib
                if ib < 4 and jb == 2:
                    self.a_go(ib, "a")
                if jb == 3:
                    self.a_read(ib, 0, (st + 1) % NSTG, ("a", 0, ib))      # A(t + 1, k step 0, ib)
        self.next_a_address()
        self.need_vm("w")                                                  # W(t + 1) has arrived; A(t + 3) stays in flight

    def prologue(self):
        e = self.e
        e("s_mov_b32 s54, m0")                                             # handed back at the end: hipcc refuses M0 in a clobber list
        e("s_mov_b64 s[56:57], %[ab]")
        e("s_mov_b64 s[58:59], %[wb]")
        e("s_mov_b32 s45, %[lda2]")
        e("s_mov_b32 s46, %[ldw2]")
        e("s_mov_b32 s47, %[rmax]")
        e("s_mov_b32 s48, %[lds]")
        e("s_mov_b32 s44, %[nk]")
        e("s_sub_u32 s50, s44, 1")
        e("s_mov_b64 s[40:41], s[56:57]")
        e("s_mov_b64 s[42:43], s[58:59]")
        e("s_lshl_b32 s53, %[wave], 10")
        e("s_add_u32 s53, s53, s48")                                       # LDS base + 1024 wave: the wave's 8 rows of every group of 32
        e("v_and_b32 v%d, 63, %%[tid]" % T(0))                             # lane
        e("v_lshrrev_b32 v%d, 6, %%[tid]" % T(1))                          # wave
        # A DMA source offsets: piece q row = 32 q + 8 wave + (lane >> 3); LDS position lane & 7 holds source chunk (lane & 7) ^ ((row >> 1) & 7)
        e("v_lshrrev_b32 v%d, 3, v%d" % (T(2), T(0)))                      # lane >> 3
        e("v_lshl_add_u32 v%d, v%d, 3, v%d" % (T(2), T(1), T(2)))          # 8 wave + (lane >> 3): row within a group of 32
        e("v_lshrrev_b32 v%d, 1, v%d" % (T(4), T(2)))
        e("v_and_b32 v%d, 7, v%d" % (T(4), T(4)))                          # (row >> 1) & 7 (32 q does not change it)
        e("v_and_b32 v%d, 7, v%d" % (T(3), T(0)))
        e("v_xor_b32 v%d, v%d, v%d" % (T(3), T(3), T(4)))                  # source chunk
        for q in range(4):
            e("v_add_u32 v%d, %d, v%d" % (T(4), 32 * q, T(2)))             # row of the tile
            e("v_min_u32 v%d, s47, v%d" % (T(5), T(4)))
            e("v_mul_lo_u32 v%d, v%d, s45" % (T(5), T(5)))
            e("v_lshl_add_u32 v%d, v%d, 4, v%d" % (AOFF(q), T(3), T(5)))
        # W load offsets in the fragment-native weight image: 16-row block 4 wave + jb (s46 bytes each = K / 64 tiles x 2 KB), + 16 lane
        e("v_and_b32 v%d, 15, v%d" % (T(2), T(0)))                         # l15
        e("v_lshrrev_b32 v%d, 4, v%d" % (T(3), T(0)))                      # g
        e("v_lshlrev_b32 v%d, 2, v%d" % (T(4), T(1)))                      # 4 wave
        for jb in range(4):
            e("v_add_u32 v%d, %d, v%d" % (T(5), jb, T(4)))
            e("v_mul_lo_u32 v%d, v%d, s46" % (T(5), T(5)))
            e("v_lshl_add_u32 v%d, v%d, 4, v%d" % (WOFF(jb), T(0), T(5)))
        # A tiles 0, 1, 2 -> stages 0, 1, 2 (indices clamped to nk - 1), W tile 0 -> set 0
        e("s_mov_b32 s49, 0")
        e("s_mov_b32 s52, 0")
        # issue order A(0), A(1), W(0), A(2): the wait for W(0) below then covers A(0) AND A(1) -- in the loop a wave's pieces of
        # A(t + 1) are always older than the W loads it waits for before it reaches barrier B(t + 1); A(2) stays in flight
        for t in (0, 1):
            if t:
                self.next_a_address()
            for q in range(4):
                self.a_m0(t, q)
                self.a_go(q, "a", nop=True)
        for ks in range(2):
            for jb in range(4):
                self.w_load(0, ks, jb, "w")
        self.next_a_address()
        for q in range(4):
            self.a_m0(2, q)
            self.a_go(q, "a", nop=True)
        self.next_a_address()                                              # s[40:41]: A tile 3
        self.next_w_address()                                              # s[42:43]: W tile 1
        # A fragment addresses: LDS base + (lane & 15) * 128 + ((4 ks + (lane >> 4)) ^ ((lane & 15) >> 1)) * 16
        e("v_lshrrev_b32 v%d, 1, v%d" % (T(4), T(2)))                      # (row >> 1) & 7 with row = 16 ib + l15
        e("v_lshlrev_b32 v%d, 7, v%d" % (T(5), T(2)))
        e("v_add_u32 v%d, s48, v%d" % (T(5), T(5)))
        for ks in range(2):
            e("v_lshl_add_u32 v%d, v%d, 1, %d" % (T(6), T(3), ks))               # chunk 2 g + ks
            e("v_xor_b32 v%d, v%d, v%d" % (T(6), T(6), T(4)))
            e("v_lshl_add_u32 v%d, v%d, 4, v%d" % (RA(ks), T(6), T(5)))
        self.need_vm("w")                                                  # A(0), A(1) and W(0) have arrived (A(2) stays in flight)
        e("s_barrier")
        for ib in range(8):
            self.a_read(ib, 0, 0, ("a", 0, ib))

    def generate(self):
        self.prologue()
        self.in_loop = True
        if "prio" in self.abl:
            self.e("s_setprio 1")
        if "aprio" in self.abl:            # variant: the two workgroups of a CU in different priority classes (by wave slot parity), to take them out of phase
            self.e("s_getreg_b32 s55, hwreg(HW_REG_HW_ID, 0, 4)")
            self.e("s_and_b32 s55, s55, 1")
            self.e("s_cmp_eq_u32 s55, 0")
            self.e("s_cbranch_scc1 lo_%=")
            self.e("s_setprio 2")
            self.e("lo_%=:")
        self.e("top_%=:")
        start, vstart = list(self.ldsq), list(self.vmq)
        for u in range(NSTG):
            self.iteration(u)
            self.e("s_sub_u32 s44, s44, 1")
            self.e("s_cmp_eq_u32 s44, 0")
            self.e("s_cbranch_scc1 done_%=" if u < NSTG - 1 else "s_cbranch_scc0 top_%=")
            assert self.abl or (start == self.ldsq and vstart == self.vmq), (u, start, self.ldsq, vstart, self.vmq)
        self.e("done_%=:")
        if "prio" in self.abl or "aprio" in self.abl:
            self.e("s_setprio 0")
        # nothing of the block may be in flight when the compiler's code resumes: LDS-DMA, fragment reads, MFMAs
        self.e("s_waitcnt vmcnt(0) lgkmcnt(0)")
        self.e("s_nop 15")
        self.e("s_nop 15")
        self.e("s_mov_b32 m0, s54")
        return self.out


def render(abl=()):
    g = Gen(abl)
    lines = g.generate()
    n = sum(1 for l in lines if not l.endswith(":"))
    head = ["// GENERATED by tools/gen/gemm_dr_gen.py -- do not edit.  %d instructions." % n,
            "// K loop of gemm_dr_kernel: W fragments v0..v63, A fragment slots v64..v95, load offsets v96..v103, fragment addresses v104..v105, temporaries v106..v113; s40..s59; a0..a127."]
    return "\n".join(head + ['"%s\\n\\t"' % l for l in lines]) + "\n"


def main():
    if "--out" in sys.argv:        # development variants
        abl = sys.argv[sys.argv.index("--abl") + 1].split(",") if "--abl" in sys.argv else ()
        with open(sys.argv[sys.argv.index("--out") + 1], "w") as f:
            f.write(render([a for a in abl if a]))
        return
    text = render()
    if "--check" in sys.argv:
        ok = os.path.exists(OUT) and open(OUT).read() == text
        print("gemm_dr_asm.inc %s" % ("up to date" if ok else "STALE"))
        sys.exit(0 if ok else 1)
    with open(OUT, "w") as f:
        f.write(text)
    print("wrote %s (%d lines)" % (OUT, text.count("\n")))


if __name__ == "__main__":
    main()
