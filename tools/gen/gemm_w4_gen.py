#!/usr/bin/env python3
"""Generator of uvltrack_amd/csrc/gemm_w4_asm.inc: the K loop of gemm_w4_kernel (256 x 256 output tile on four waves, one per SIMD; a wave
owns 128 x 128 = 8 x 8 blocks of v_mfma_f32_16x16x32_bf16 = all 256 AGPRs) as ONE inline-asm statement that owns v0..v161, s40..s59 and
names the accumulators a0..a255 directly (the C++ side pins its sixteen f32x16 operands to a[0:15] ... a[240:255]).

Why generated, and why this shape (measurements: tools/probes/gemm_w4_probe.hip, profiles/r03_gemm_w4.md):
  * with one wave per SIMD nothing hides a badly placed instruction, and hipcc places them badly (its version of this loop: 1.1 PFLOP/s);
  * a hand-placed loop keeps the matrix pipe busy 97-98 % of its cycles -- and from there on throughput is the shader CLOCK the power
    management grants under the load (1.5-1.8 GHz, not 2.4): every byte moved costs clock.  So the tile travels global -> LDS by LDS-DMA
    (no VGPR round trip, no ds_write: +4 %), and the MFMA is the 16x16x32 form, which moves half the accumulator bytes per flop of
    32x32x16 (+10 % clock at the same pipe occupancy).

The schedule.  A K tile is 64 wide: 256 A rows + 256 W rows of 128 bytes = 64 KB, two buffers.  Per K tile and wave: two k steps of 32,
each 64 MFMAs of 16 cycles on 8 A + 8 W fragments; 32 fragment reads, 16 LDS-DMA instructions (1 KB each: 8 rows), one barrier.

    k step 0   64 MFMAs on fragment set 0 | the 16 reads of k step 1 -> set 1 (one per 4 MFMAs) | the last 2 LDS-DMAs of tile t + 1
    k step 1    8 MFMAs on set 1, vmcnt (tile t + 1 has landed), lgkmcnt(0), s_barrier, 56 MFMAs
               | the 16 reads of tile t + 1's k step 0 -> set 0 (one per 2 MFMAs) | the first 14 LDS-DMAs of tile t + 2 (one per 4 MFMAs)

  * tile t + 2 goes into the buffer tile t is read from: requested from the barrier on, landed by the next barrier (>= 1000 cycles
    for the youngest piece);
  * tile indices beyond the last are clamped (the last tile is fetched again into a buffer nobody reads): no tail form; the loop is
    unrolled over the two buffers with an exit test per tile;
  * `lgkmcnt` / `vmcnt` are counted by the generator from its own issue order; an M0 write and the LDS-DMA that uses it sit behind
    different MFMAs.

LDS image of a tile: rows of 128 bytes, 16-byte chunk c of row r at position c ^ ((r >> 1) & 7) -- applied to the per-lane SOURCE address
of the DMA (its LDS side is lane-linear) and again on the fragment read.  Fragment of a 16-row block for k step s: lane l reads row
l & 15, chunk 4 s + (l >> 4).  Accumulator block (ib, jb) = a[4 (8 ib + jb) .. + 3] = mfma(W fragment jb, A fragment ib): lane l holds
output row 16 ib + (l & 15), columns 16 jb + 4 (l >> 4) + r.

Registers: fragments v0..v127 (set p, A block ib: 64 p + 4 ib; W block jb: 64 p + 32 + 4 jb), DMA source offsets v128..v143 (A pieces 0..7,
W pieces 0..7), fragment addresses v144..v151 ([A|W][buffer][k step]), temporaries v152..v161.  s[40:41] / s[42:43] A / W address of the
tile being requested, s44 tiles left, s45 / s46 row pitches in bytes, s47 last valid A row of the tile, s48 LDS base, s49 index of the
tile requested next, s50 nk - 1, s51 scratch, s53 LDS base + 1024 * wave, s[56:57] / s[58:59] A / W address of tile 0; the trace form
also uses s52, s54 and s60..s63.

gfx950 note found the hard way: `v_readfirstlane_b32 sN, v` followed directly by an SALU read of sN returned a stale value here (no
interlock); the wave index therefore comes in as an SGPR operand.

Usage: python tools/gen/gemm_w4_gen.py [--check]            (writes / compares uvltrack_amd/csrc/gemm_w4_asm.inc)
       python tools/gen/gemm_w4_gen.py --out PATH [--abl nodma,noread,nobar,prologue_only] [--trace]   (variants for tools/probes/)"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
OUT = os.path.join(ROOT, "uvltrack_amd", "csrc", "gemm_w4_asm.inc")

STAGE = 65536                      # one K tile (64 wide): 256 A rows + 256 W rows, 128 bytes each
W_OFF = 32768
NDMA3 = 14                         # LDS-DMA instructions of the next-but-one tile issued in k step 1 (behind the barrier); the rest in the next k step 0
BAR_AT = 8                         # the barrier sits in front of this MFMA of k step 1


def FA(p, i):
    return 64 * p + 4 * i


def FB(p, j):
    return 64 * p + 32 + 4 * j


OFF = lambda n: 128 + n            # DMA source offset of piece n (0..7 A, 8..15 W)
RA = lambda buf, ks: 144 + 2 * buf + ks
RB = lambda buf, ks: 148 + 2 * buf + ks
T = lambda k: 152 + k
NV = 162                           # first VGPR the block does not touch


def v4(r):
    return "v[%d:%d]" % (r, r + 3)


READ_ORDER = [("b", 0), ("a", 0), ("a", 1), ("b", 1), ("a", 2), ("a", 3), ("b", 2), ("b", 3), ("a", 4), ("a", 5), ("b", 4), ("b", 5), ("a", 6), ("a", 7), ("b", 6), ("b", 7)]
_ARR = {x: n for n, x in enumerate(READ_ORDER)}
# MFMAs in the order their fragments arrive
MFMA_ORDER = sorted(((i, j) for i in range(8) for j in range(8)), key=lambda ij: (max(_ARR[("a", ij[0])], _ARR[("b", ij[1])]), ij))


class Gen:
    def __init__(self, abl=(), trace=False):
        self.abl = set(abl)        # timing-only ablations (results wrong): nodma, noread, nobar; prologue_only (debugging aid)
        self.trace = trace         # development build: shader-clock / constant-clock stamps around the loop, stored through %[dbg]
        self.out = []
        self.ldsq = []             # tags of the LDS reads in flight, oldest first (positions from the END are exact)
        self.vmq = []              # tile numbers of the LDS-DMA instructions in flight, oldest first
        self.gen = 0               # number of the tile the next iteration multiplies
        self.in_loop = False

    def e(self, s):
        self.out.append(s)

    def ds_read(self, dst, addr, off, tag):
        if "noread" in self.abl and self.in_loop:
            return
        self.e("ds_read_b128 %s, v%d offset:%d" % (v4(dst), addr, off))
        self.ldsq.append(tag)

    def need(self, tag):
        if tag not in self.ldsq:
            return
        k = max(i for i, t in enumerate(self.ldsq) if t == tag)
        self.e("s_waitcnt lgkmcnt(%d)" % min(len(self.ldsq) - 1 - k, 15))
        self.ldsq = self.ldsq[k + 1:]

    def lds_drain(self):
        self.e("s_waitcnt lgkmcnt(0)")
        self.ldsq = []

    def dma_m0(self, buf, n):
        """M0 for piece n (0..7 A, 8..15 W: 8 rows of 128 bytes) of buffer buf"""
        if "nodma" in self.abl and self.in_loop:
            return
        self.e("s_add_u32 m0, s53, %d" % (buf * STAGE + (0 if n < 8 else W_OFF) + (n & 7) * 4096))

    def dma_go(self, n, tag, nop=False):
        if "nodma" in self.abl and self.in_loop:
            return
        if nop:
            self.e("s_nop 0")
        self.e("global_load_lds_dwordx4 v%d, s[%d:%d]" % (OFF(n), 40 if n < 8 else 42, 41 if n < 8 else 43))
        self.vmq.append(tag)

    def need_tile(self, tag):
        """every LDS-DMA instruction of tile `tag` has landed (they retire in order)"""
        if tag not in self.vmq:
            assert "nodma" in self.abl
            return
        k = max(i for i, t in enumerate(self.vmq) if t == tag)
        self.e("s_waitcnt vmcnt(%d)" % (len(self.vmq) - 1 - k))
        self.vmq = self.vmq[k + 1:]

    def frag_reads(self, buf, ks, p):
        """the sixteen reads of k step ks of buffer buf into set p, in READ_ORDER"""
        return [(lambda x=x, i=i: self.ds_read(FA(p, i) if x == "a" else FB(p, i), RA(buf, ks) if x == "a" else RB(buf, ks), i * 2048, (x, p, i))) for x, i in READ_ORDER]

    def mfma(self, p, i, j):
        self.need(("a", p, i))
        self.need(("b", p, j))
        a0 = 4 * (8 * i + j)
        self.e("v_mfma_f32_16x16x32_bf16 a[%d:%d], %s, %s, a[%d:%d]" % (a0, a0 + 3, v4(FB(p, j)), v4(FA(p, i)), a0, a0 + 3))

    def next_tile_address(self):
        # s49 = min(s49 + 1, nk - 1); s[40:41] / s[42:43] = tile 0 address + 128 s49
        self.e("s_add_u32 s49, s49, 1")
        self.e("s_min_u32 s49, s49, s50")
        self.e("s_lshl_b32 s51, s49, 7")
        self.e("s_add_u32 s40, s56, s51")
        self.e("s_addc_u32 s41, s57, 0")
        self.e("s_add_u32 s42, s58, s51")
        self.e("s_addc_u32 s43, s59, 0")

    def iteration(self, rb):
        """K tile t in buffer rb: finish requesting tile t + 1 (buffer 1 - rb), hand over to it, start requesting tile t + 2 into rb"""
        wb = 1 - rb
        # ---- k step 0: set 0; reads of k step 1 behind MFMAs 1, 5, 9, ...; the remaining pieces of tile t + 1: M0 behind MFMA 4 d, DMA behind 4 d + 2
        rd = self.frag_reads(rb, 1, 1)
        late = {}
        for d, n in enumerate(range(NDMA3, 16)):
            late[4 * d] = lambda n=n: self.dma_m0(wb, n)
            late[4 * d + 2] = lambda n=n: self.dma_go(n, self.gen + 1)
        for m, (i, j) in enumerate(MFMA_ORDER):
            self.mfma(0, i, j)
            if m % 4 == 1:
                rd[m // 4]()
            if m in late:
                late[m]()
        # ---- k step 1: set 1; barrier in front of MFMA BAR_AT; then reads of tile t + 1's k step 0 behind odd MFMAs, pieces of tile t + 2 behind even ones
        rd = self.frag_reads(wb, 0, 0)
        early = {}
        for d in range(NDMA3):
            early[BAR_AT + 4 * d] = lambda d=d: self.dma_m0(rb, d)
            early[BAR_AT + 2 + 4 * d] = lambda d=d: self.dma_go(d, self.gen + 2)
        assert max(early) < 64
        for m, (i, j) in enumerate(MFMA_ORDER):
            if m == BAR_AT:
                self.need_tile(self.gen + 1)           # this wave's pieces of tile t + 1 have landed ...
                self.lds_drain()                       # ... and its reads of buffer rb are complete
                if "nobar" not in self.abl:
                    self.e("s_barrier")
                self.next_tile_address()               # tile t + 2
            self.mfma(1, i, j)
            if m >= BAR_AT + 1 and (m - BAR_AT - 1) % 2 == 0 and (m - BAR_AT - 1) // 2 < 16:
                rd[(m - BAR_AT - 1) // 2]()
            if m in early:
                early[m]()
        self.gen += 1

    def prologue(self):
        e = self.e
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
        e("s_add_u32 s53, s53, s48")                                   # LDS base + 1024 wave: the wave's 8 rows of every group of 32
        e("v_and_b32 v%d, 63, %%[tid]" % T(0))                         # lane
        e("v_lshrrev_b32 v%d, 6, %%[tid]" % T(1))                      # wave
        # DMA source offsets: piece q row = 32 q + 8 wave + (lane >> 3); LDS position lane & 7 holds source chunk (lane & 7) ^ ((row >> 1) & 7)
        e("v_lshrrev_b32 v%d, 3, v%d" % (T(2), T(0)))                  # lane >> 3
        e("v_lshl_add_u32 v%d, v%d, 3, v%d" % (T(2), T(1), T(2)))      # 8 wave + (lane >> 3): row within a group of 32
        e("v_lshrrev_b32 v%d, 1, v%d" % (T(4), T(2)))
        e("v_and_b32 v%d, 7, v%d" % (T(4), T(4)))                      # (row >> 1) & 7 (32 q does not change it)
        e("v_and_b32 v%d, 7, v%d" % (T(3), T(0)))
        e("v_xor_b32 v%d, v%d, v%d" % (T(3), T(3), T(4)))              # source chunk
        for q in range(8):
            e("v_add_u32 v%d, %d, v%d" % (T(4), 32 * q, T(2)))         # row of the tile
            e("v_min_u32 v%d, s47, v%d" % (T(5), T(4)))
            e("v_mul_lo_u32 v%d, v%d, s45" % (T(5), T(5)))
            e("v_lshl_add_u32 v%d, v%d, 4, v%d" % (OFF(q), T(3), T(5)))
            e("v_mul_lo_u32 v%d, v%d, s46" % (T(5), T(4)))
            e("v_lshl_add_u32 v%d, v%d, 4, v%d" % (OFF(8 + q), T(3), T(5)))
        # tile 0 into buffer 0, the part of tile 1 the loop's schedule has issued by the top of an iteration into buffer 1 (nk >= 2)
        e("s_mov_b32 s49, 0")
        for n in range(16):
            self.dma_m0(0, n)
            self.dma_go(n, 0, nop=True)
        self.next_tile_address()
        for n in range(NDMA3):
            self.dma_m0(1, n)
            self.dma_go(n, 1, nop=True)
        # fragment addresses: l15 = lane & 15, g = lane >> 4, sw = l15 >> 1, wm = wave >> 1, wn = wave & 1
        e("v_and_b32 v%d, 15, v%d" % (T(2), T(0)))                     # l15
        e("v_lshrrev_b32 v%d, 4, v%d" % (T(3), T(0)))                  # g
        e("v_lshrrev_b32 v%d, 1, v%d" % (T(4), T(2)))                  # sw = ((row >> 1) & 7), row = 16 block + l15
        e("v_lshrrev_b32 v%d, 1, v%d" % (T(5), T(1)))                  # wm
        e("v_and_b32 v%d, 1, v%d" % (T(6), T(1)))                      # wn
        e("v_lshlrev_b32 v%d, 14, v%d" % (T(5), T(5)))
        e("v_lshl_add_u32 v%d, v%d, 7, v%d" % (T(5), T(2), T(5)))      # A rows: (128 wm + l15) * 128
        e("v_add_u32 v%d, s48, v%d" % (T(5), T(5)))
        e("v_lshlrev_b32 v%d, 14, v%d" % (T(6), T(6)))
        e("v_lshl_add_u32 v%d, v%d, 7, v%d" % (T(6), T(2), T(6)))      # W rows: 32768 + (128 wn + l15) * 128
        e("v_add_u32 v%d, s48, v%d" % (T(6), T(6)))
        e("v_add_u32 v%d, 0x%x, v%d" % (T(6), W_OFF, T(6)))
        for ks in range(2):
            e("v_add_u32 v%d, %d, v%d" % (T(7), 4 * ks, T(3)))
            e("v_xor_b32 v%d, v%d, v%d" % (T(7), T(7), T(4)))
            e("v_lshlrev_b32 v%d, 4, v%d" % (T(7), T(7)))
            e("v_add_u32 v%d, v%d, v%d" % (RA(0, ks), T(5), T(7)))
            e("v_add_u32 v%d, v%d, v%d" % (RB(0, ks), T(6), T(7)))
            e("v_add_u32 v%d, 0x%x, v%d" % (RA(1, ks), STAGE, RA(0, ks)))
            e("v_add_u32 v%d, 0x%x, v%d" % (RB(1, ks), STAGE, RB(0, ks)))
        self.need_tile(0)
        e("s_barrier")
        for f in self.frag_reads(0, 0, 0):
            f()

    def generate(self):
        self.prologue()
        if self.trace:
            self.e("s_memtime s[60:61]")
            self.e("s_memrealtime s[62:63]")
            self.e("s_waitcnt lgkmcnt(0)")
            self.e("s_mov_b32 s52, s60")
            self.e("s_mov_b32 s54, s62")
        if "prologue_only" in self.abl:            # debugging aid (tools/probes/w4_lds_dump.hip): stop behind the prologue, LDS image intact
            self.e("s_branch done_%=")
        self.in_loop = True
        self.e("top_%=:")
        start, vstart = list(self.ldsq), [t - self.gen for t in self.vmq]
        for b in range(2):
            self.iteration(b)
            self.e("s_sub_u32 s44, s44, 1")
            self.e("s_cmp_eq_u32 s44, 0")
            self.e("s_cbranch_scc1 done_%=" if b == 0 else "s_cbranch_scc0 top_%=")
            assert self.abl or (start == self.ldsq and vstart == [t - self.gen for t in self.vmq]), (start, self.ldsq, vstart, self.vmq)
        self.e("done_%=:")
        # nothing of the block may be in flight when the compiler's code resumes: LDS-DMA, fragment reads, MFMAs
        self.e("s_waitcnt vmcnt(0) lgkmcnt(0)")
        self.e("s_nop 15")
        self.e("s_nop 15")
        if self.trace:             # loop cycles (shader clock) and loop time (constant 100 MHz clock) of every wave; the last workgroup's survive
            self.e("s_memtime s[60:61]")
            self.e("s_memrealtime s[62:63]")
            self.e("s_waitcnt lgkmcnt(0)")
            self.e("s_sub_u32 s60, s60, s52")
            self.e("s_sub_u32 s62, s62, s54")
            self.e("v_lshrrev_b32 v%d, 6, %%[tid]" % T(0))
            self.e("v_lshlrev_b32 v%d, 3, v%d" % (T(0), T(0)))
            self.e("v_mov_b32 v%d, s60" % T(2))
            self.e("v_mov_b32 v%d, s62" % T(3))
            self.e("global_store_dwordx2 v%d, v[%d:%d], %%[dbg]" % (T(0), T(2), T(3)))
            self.e("s_waitcnt vmcnt(0)")
        return self.out


def render(abl=(), trace=False):
    g = Gen(abl, trace)
    lines = g.generate()
    n = sum(1 for l in lines if not l.endswith(":"))
    head = ["// GENERATED by tools/gen/gemm_w4_gen.py -- do not edit.  %d instructions." % n,
            "// K loop of gemm_w4_kernel: fragments v0..v127, DMA offsets v128..v143, fragment addresses v144..v151, temporaries v152..v161; s40..s59; a0..a255."]
    return "\n".join(head + ['"%s\\n\\t"' % l for l in lines]) + "\n"


def main():
    if "--out" in sys.argv:        # development variants
        abl = sys.argv[sys.argv.index("--abl") + 1].split(",") if "--abl" in sys.argv else ()
        with open(sys.argv[sys.argv.index("--out") + 1], "w") as f:
            f.write(render([a for a in abl if a], "--trace" in sys.argv))
        return
    text = render()
    if "--check" in sys.argv:
        ok = os.path.exists(OUT) and open(OUT).read() == text
        print("gemm_w4_asm.inc %s" % ("up to date" if ok else "STALE"))
        sys.exit(0 if ok else 1)
    with open(OUT, "w") as f:
        f.write(text)
    print("wrote %s (%d lines)" % (OUT, text.count("\n")))


if __name__ == "__main__":
    main()
