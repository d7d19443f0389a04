// bf16 MFMA GEMM  C = epilogue(A[M,K] * W[N,K]^T + bias)  for gfx950.
//
// One kernel template serves every dense contraction of the frame:
//   * nn.Linear of the ViT blocks and BERT layers        (reference block.py:42,44; backbones/utils.py:58,61;
//                                                          bert_backbone.py:289-291,338,366,379)
//   * the patch-embed conv as a GEMM over im2row patches   (mae_vit.py:92,99)
//   * the 3x3 conv towers of the box head as implicit GEMM over NHWC tokens, grouped by tower
//                                                          (heads/utils.py:126-131, head:28-50)
// Both operands are K-contiguous (nn.Linear stores W as [out,in]), so A and B fragments of
// v_mfma_f32_32x32x16_bf16 are single 16-byte LDS reads.  Tiles go HBM -> LDS by LDS-DMA (global_load_lds_dwordx4, no
// VGPR round trip) into an NS-deep ring with counted s_waitcnt vmcnt and a raw s_barrier; LDS rows are 128 bytes and
// XOR-swizzled (on the per-lane DMA source address and again on the fragment read) so ds_read_b128 is conflict-free.
// The workgroup -> tile map is XCD-aware (see gemm_glds_body).
#include <cstdio>
#include <type_traits>
#include "common.h"
#include "kernels.h"
#include "ln_body.h"
#include "gemm_epi.h"
#include "fold.h"

namespace uvl {

// ------------------------------------------------------------------------------------------------
// The kernel: tiles go HBM -> LDS directly (global_load_lds,
// 16 B per lane, no VGPR round trip) into an NS-deep ring, NS-1 tiles in flight per workgroup, counted
// s_waitcnt vmcnt(N) so loads stay in flight across the (raw) barrier.  The LDS image of a DMA is
// lane-linear, so the XOR swizzle is applied to the per-lane SOURCE address (same 128-byte line, no
// extra fetch) and again on the fragment read.  At batch 1 a workgroup streams its weight panel with
// exposed HBM latency per K-step; the ring hides it.
// ------------------------------------------------------------------------------------------------
// One k step (16 wide) of a 32 x 32 output block in the pipelined kernels.  MI16 = false: one v_mfma_f32_32x32x16_bf16 on fragment ks of
// both operands.  MI16 = true (the product form): the same flops as two v_mfma_f32_16x16x32_bf16 -- the fragments of a block are then
// indexed t = 2 * (16-row half) + (32-wide k step), "k step" ks stands for (k step ks >> 1, row half ks & 1) of the A operand and both
// column halves of the W operand, and registers 4 (2 hi + hj) .. + 3 of the block are its 16 x 16 sub-block (hi, hj) (gemm_epilogue_lds,
// L16).  Half the accumulator traffic per flop: the loop runs at a higher clock at the same pipe occupancy (profiles/r03_gemm_w4.md,
// sections 2-3 and 8: +2.5..7 % on the pipelined kernels).
template <bool MI16>
__device__ __forceinline__ void pipe_mfma(f32x16& c, const bf16x8 (&wq)[4], const bf16x8 (&aq)[4], const int ks) {
    if constexpr (!MI16) {
        c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wq[ks], aq[ks], c, 0, 0, 0);
    } else {
        const int s = ks >> 1, hi = ks & 1;
#pragma unroll
        for (int hj = 0; hj < 2; ++hj) {
            const int r0 = 4 * (2 * hi + hj);
            f32x4 sub = {c[r0], c[r0 + 1], c[r0 + 2], c[r0 + 3]};
            sub = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wq[2 * hj + s], aq[2 * hi + s], sub, 0, 0, 0);
            c[r0] = sub[0]; c[r0 + 1] = sub[1]; c[r0 + 2] = sub[2]; c[r0 + 3] = sub[3];
        }
    }
}
// fragment t of a 32-row block whose first row is `row0` (lane-independent part): LDS byte offset of this lane's 16 bytes
template <bool MI16>
__device__ __forceinline__ int pipe_frag_off(int row0, int t, int lane) {
    if constexpr (!MI16) return swz128(row0 + (lane & 31), t * 2 + (lane >> 5));
    else return swz128(row0 + 16 * (t >> 1) + (lane & 15), 4 * (t & 1) + (lane >> 4));
}
template <int N_> __device__ __forceinline__ void wait_vmcnt() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N_) : "memory"); }


// NTW: the weight tiles are requested non-temporal (aux = 2).  For the text-branch riders of a one-sequence frame (M = 40 rows: every
// weight byte is read once, by one CU) -- see the note at gemm_glds_pair_kernel.
// BK: K extent of a stage, 64 or 32.  At BK = 32 a 128x128 stage is 16 KB, so four or five stages fit twice per CU: three or four
// K tiles in flight per workgroup instead of one.  With two stages a K step costs a whole DMA round trip (~1.1 us issue -> landed)
// plus its MFMAs, which is where the 31-33 % MFMA utilisation of the batched frame's GEMMs comes from.
// PROD: 0 = every wave stages its share of a tile and computes; 2 / 4 = that many extra PRODUCER waves issue all LDS-DMA instructions (and
// wait for them) while the WGM x WGN consumer waves only read fragments and issue MFMAs -- an LDS-DMA instruction costs the wave that
// issues it ~55 cycles, half of what a consumer of the 128x128 tile issues per K step (profiles/r02_gemm_structure.md, probe 5).
// PRE (the fused LayerNorm + GEMM launch): 1 = request the WEIGHT halves of the first NS - 1 K tiles and return (they do not depend on
// the LayerNorm and fly under it and the grid barrier); 2 = the rest of the tile: the activation halves, then the usual loop.
// Development builds only (tools/glds_trace.py; the product library defines none of these): GLDS_TRACE = wave 0 of every workgroup stamps the 100 MHz clock at
// the seams of gemm_glds_body; GLDS_ABL = timing ablations of its K loop (results are wrong): 1 no MFMA, 2 no fragment reads, 4 no LDS-DMA inside the loop,
// 8 no barrier; GLDS_ORDER=0 = the round-1..4 order of the 64 x 64 loop (LDS-DMA requested before the fragment reads).
#ifndef GLDS_ABL
#define GLDS_ABL 0
#endif
#ifndef GLDS_ORDER
#define GLDS_ORDER 1
#endif
#ifdef GLDS_TRACE
__device__ unsigned long long g_glds_trace[2048 * 8];
__device__ unsigned long long g_glds_acc[8 * 8];     // [4 * NTW + EPI][phase 0..5 ticks summed over every workgroup of every launch, 6: workgroups, 7: K tiles]: the loop inside a whole frame
#define GLDS_STAMP(k) do { gl_t[k] = __builtin_amdgcn_s_memrealtime(); } while (0)
#define GLDS_TRACE_FLUSH() do { if (threadIdx.x == 0) { if (nk < 2) gl_t[4] = gl_t[3]; if (blockIdx.y == 0 && blockIdx.z == 0 && blockIdx.x < 2048) for (int k = 0; k < 7; ++k) g_glds_trace[blockIdx.x * 8 + k] = gl_t[k]; unsigned long long* a = &g_glds_acc[(4 * (int)NTW + EPI) * 8]; for (int k = 0; k < 6; ++k) atomicAdd(a + k, gl_t[k + 1] - gl_t[k]); atomicAdd(a + 6, 1ull); atomicAdd(a + 7, (unsigned long long)nk); } } while (0)
#else
#define GLDS_STAMP(k) do { } while (0)
#define GLDS_TRACE_FLUSH() do { } while (0)
#endif
// LNF (round 6, LayerNorm-free one-sequence frames): A holds the UN-normalised rows rounded to bf16, W / bias are the LayerNorm-folded weight (W gamma, b + W beta),
// p.st_in the rows' partial statistics, p.colsum the folded weight's row sums: the epilogue turns the product into  rstd (acc - mean colsum) + bias  = LayerNorm(a~) W'^T + b'
// (fold.h).  The partials of a lane's row (the transposed tile: lane = row) are requested with the bias, before the first LDS-DMA; nothing is added to the K loop.
template <int BM, int BN, int WGM, int WGN, int EPI, int NS, bool CONV, bool NTW = false, int BK = 64, int PROD = 0, int PRE = 0, bool LNF = false>
__device__ __forceinline__ void gemm_glds_body(const GemmParams& p, const int bx, const int sk_in, const int g, char* smem) {
#ifdef GLDS_TRACE
    unsigned long long gl_t[7] = {};      // (scalar registers: the stamps cost the loop nothing but the s_memrealtime itself; everything is written out behind the last one)
#endif
    GLDS_STAMP(0);
    int sk = sk_in;
    static_assert(PRE == 0 || (!CONV && !PROD && NS == 4), "split prologue: plain four-stage tiles only");
    static_assert(BK == 64 || (BK == 32 && !CONV), "K extent of a stage");
    static_assert(PROD == 0 || (!CONV && !NTW), "producer waves: plain GEMMs only");
    constexpr int RB = BK * 2;                       // bytes of a stage row
    constexpr int LPR_ = RB / 16;                    // lanes (16-byte chunks) per row
    constexpr int RPD = 1024 / RB;                   // rows per DMA instruction (1 KB)
    constexpr int KS = BK / 16;                      // MFMA k steps per stage
    constexpr int WM = BM / WGM, WN = BN / WGN;
    constexpr int TM = WM / 32, TN = WN / 32;
    constexpr int ROWS = BM + BN;                    // stage = A rows then W rows, RB bytes each
    constexpr int STAGE = ROWS * RB;
    constexpr int NW = WGM * WGN;                    // (consumer) waves per workgroup: 4 (tiles up to 128x128) or 8 (256-wide tiles)
    constexpr int NL = PROD ? PROD : NW;             // waves that issue LDS-DMA
    constexpr int LPT = ROWS / (RPD * NL);           // DMA instructions per loading wave per tile (each fills RPD rows)
    constexpr int LPT_A = BM / (RPD * NL);           // the first LPT_A instructions of a wave fill A rows, the rest W rows
    static_assert((NW == 4 || NW == 8) && BM % (RPD * NL) == 0 && BN % (RPD * NL) == 0 && LPT * (NS - 2) <= 63, "geometry");
    // XOR swizzle of the 16-byte chunk index by row: 128-byte rows (row >> 1) & 7 (swz128), 64-byte rows (row >> 2) & 3 -- the 16
    // lanes of a ds_read_b128 service group then hit 16 distinct 16-byte slots of the 256-byte bank row either way
    auto swz = [](int row, int chunk) __attribute__((always_inline)) {
        return BK == 64 ? swz128(row, chunk) : row * 64 + ((chunk ^ ((row >> 2) & 3)) << 4);
    };

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave_all = __builtin_amdgcn_readfirstlane(tid >> 6);   // wave-uniform on purpose: LDS-DMA destinations (M0) stay scalar
    const bool producer = PROD && wave_all >= NW;
    const int wave = PROD ? (producer ? wave_all - NW : wave_all) : wave_all;     // index among the loading waves / among the consumers
    const int wm = wave / WGN, wn = wave % WGN;
    const int MT = (p.M + BM - 1) / BM, NT = p.N / BN;
    // Workgroup b runs on XCD b % 8 (dispatch order; affects speed only), and every XCD has its own L2.
    const int xcd = bx & 7, idx = bx >> 3;
    int nt, mt;
    // (every division below is by a value the launcher knows: gemm_derive leaves the reciprocals in the argument block -- common.h::FastDiv)
    if (p.group_m < 0) {
        // K-SLICE map of a split-K GEMM with few M tiles (one or two sequences): the grid is 1-D over (tile, slice) and XCD x owns ONE
        // K slice (x / nparts) and one contiguous share of the N panels (x % nparts, nparts = 8 / splitk): its L2 then holds
        // A[:, slice] and its share of W[:, slice] -- with the tile map below every XCD pulls ALL of A through its L2 for every slice
        // (fc2 of UVLTrack-B at one sequence: 8 x 3.4 MB of A per launch instead of 8 x 0.85 MB)
        const int NTp = p.ks_ntp;                            // nparts = 1 << ks_log2 = 8 / splitk
        sk = xcd >> p.ks_log2;
        if (idx >= MT * NTp) return;
        const int q = (int)fd_div((uint32_t)idx, p.fd_mt);
        nt = (xcd & ((1 << p.ks_log2) - 1)) * NTp + q;
        mt = idx - q * MT;
    } else if (p.group_m == 0) {
        // panel map (UVL_GEMM_GM=0 only; the default of the first builds): an XCD owns whole N panels, so every weight byte enters
        // exactly one L2 -- but XCDs get unequal shares unless N / BN is a multiple of 8, see launch_glds
        const int q = (int)fd_div((uint32_t)idx, p.fd_mt);
        nt = q * 8 + xcd;
        mt = idx - q * MT;
        if (nt >= NT) return;
    } else {
        // Tiles are ordered in groups of group_m M-tiles x all N-tiles (M fastest) and the order is cut into 8 contiguous runs,
        // one per XCD.  Large M (group_m = 8): the ~64 tiles resident on an XCD form a near-square patch, so its L2 holds 8 A
        // tiles + a few W panels instead of re-streaming all of A once per N panel.  Few M tiles (group_m = MT): N-major order.
        const int T = MT * NT, base = T >> 3, rem = T & 7;
        const int cnt = base + (xcd < rem ? 1 : 0);
        if (idx >= cnt) return;
        const int L = xcd * base + (xcd < rem ? xcd : rem) + idx;
        const int gsz = p.group_m * NT, gi = (int)fd_div((uint32_t)L, p.fd_gsz), within = L - gi * gsz;
        const int gm = min(p.group_m, MT - gi * p.group_m);
        nt = (int)fd_div((uint32_t)within, gm == p.group_m ? p.fd_gm : p.fd_gml);
        mt = gi * p.group_m + (within - nt * gm);
    }
    const int m0 = mt * BM, n0 = nt * BN;

    // split-K: slice sk owns the K range [sk*K/splitk, (sk+1)*K/splitk) and writes its own f32 partial slab
    const int kspan = p.kspan;
    const int kbase = sk * kspan;
    // per-lane DMA sources: instruction i of this wave fills stage rows [8*(wave + NW*i), +8)
    const bf16_t* src[CONV ? LPT : 1];                       // conv: full per-lane pointers (the tap moves them per K tile)
    uint32_t loff[CONV ? 1 : LPT];                           // plain: 32-bit byte offsets from two wave-uniform bases, so a DMA
                                                             // costs no VALU (global_load_lds v_off, s[base:base+1])
    int a_i[LPT_A], a_j[LPT_A];                              // conv: pixel of the A row this lane fetches
    const int convF = p.conv_F, cin_g = p.cin_g, lda = p.lda;
    const char* a_base = reinterpret_cast<const char*>(p.A + (size_t)m0 * lda + kbase);
    const char* w_base = reinterpret_cast<const char*>(p.W + ((size_t)g * p.N + n0) * p.ldw + kbase);
#pragma unroll
    for (int i = 0; i < LPT; ++i) {
        const int r = RPD * (wave + NL * i) + lane / LPR_;
        const int chunk = BK == 64 ? ((lane & 7) ^ ((r >> 1) & 7)) : ((lane & 3) ^ ((r >> 2) & 3));   // logical 16-byte chunk that belongs at this physical slot
        if (i < LPT_A) {
            int gm = m0 + r;
            gm = gm < p.M ? gm : p.M - 1;
            if constexpr (CONV) {
                const int S = convF * convF;
                const int b = gm / S, pix = gm - b * S;
                a_i[i] = pix / convF;
                a_j[i] = pix - a_i[i] * convF;
                const int goff = g == 0 ? p.a_goff[0] : g == 1 ? p.a_goff[1] : g == 2 ? p.a_goff[2] : p.a_goff[3];
                src[i] = p.A + (size_t)b * S * lda + goff + chunk * 8;       // + (pixel row) * lda + channel, per K tile
            } else {
                loff[i] = (uint32_t)(gm - m0) * (uint32_t)lda * 2u + (uint32_t)chunk * 16u;
            }
        } else {
            if constexpr (CONV) src[i] = p.W + ((size_t)g * p.N + n0 + r - BM) * p.ldw + kbase + chunk * 8;
            else loff[i] = (uint32_t)(r - BM) * (uint32_t)p.ldw * 2u + (uint32_t)chunk * 16u;
        }
    }
    // (named OUTSIDE the lambda: a lambda is implicitly host-device, and a static device variable it names is "used by host code" -- hipcc then externalises it
    // and every kernel of this file reaches it through a GOT load, a dependent scalar round trip in front of the bias loads and the first LDS-DMA)
    const bf16_t* const zero_page = reinterpret_cast<const bf16_t*>(g_zero_page);
    auto issue = [&](int kt, bool do_a = true, bool do_w = true) __attribute__((always_inline)) {
        char* st = smem + (kt % NS) * STAGE;
        if constexpr (CONV) {
            // K index = tap * cin_g + channel; a 64-wide tile never straddles taps
            const int k0 = kbase + kt * BK;
            const int tap = k0 / cin_g;
            const int c0 = k0 - tap * cin_g;
            const int tap_di = tap / 3 - 1, tap_dj = tap % 3 - 1;
#pragma unroll
            for (int i = 0; i < LPT; ++i) {
                const bf16_t* gp;
                if (i < LPT_A) {
                    const int ii = a_i[i] + tap_di, jj = a_j[i] + tap_dj;
                    const bool ok = (unsigned)ii < (unsigned)convF && (unsigned)jj < (unsigned)convF;
                    gp = ok ? src[i] + (size_t)(ii * convF + jj) * lda + c0 : zero_page;
                } else {
                    gp = src[i] + kt * BK;
                }
                __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)gp,
                                                 (__attribute__((address_space(3))) void*)(st + (wave + NL * i) * 1024), 16, 0, 0);
            }
        } else {
            // wave-uniform bases; the readfirstlane pair pins them in SGPRs (otherwise LLVM folds them back into per-lane
            // 64-bit pointers and pays a 64-bit VALU add per DMA)
            auto pin = [](const char* q) __attribute__((always_inline)) {
                const uint64_t u = reinterpret_cast<uint64_t>(q);
                const uint32_t lo = __builtin_amdgcn_readfirstlane((uint32_t)u), hi = __builtin_amdgcn_readfirstlane((uint32_t)(u >> 32));
                return reinterpret_cast<const char*>(((uint64_t)hi << 32) | lo);
            };
            const char* ab = pin(a_base + (size_t)kt * (BK * 2));
            const char* wb = pin(w_base + (size_t)kt * (BK * 2));
#pragma unroll
            for (int i = 0; i < LPT; ++i) {
                if ((i < LPT_A && !do_a) || (i >= LPT_A && !do_w)) continue;
                const char* gp = (i < LPT_A ? ab : wb) + loff[i];
                if (NTW && i >= LPT_A)
                    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)gp,
                                                     (__attribute__((address_space(3))) void*)(st + (wave + NL * i) * 1024), 16, 0, 2);
                else
                    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)gp,
                                                     (__attribute__((address_space(3))) void*)(st + (wave + NL * i) * 1024), 16, 0, 0);
            }
        }
    };

    const int nk_ = kspan / BK;
    if constexpr (PRE == 1) {            // weight halves of the first tiles only
#pragma unroll
        for (int t = 0; t < NS - 1; ++t)
            if (t < nk_) issue(t, false, true);
        return;
    }
    GLDS_STAMP(1);
    f32x4 bias_v[TN][4];
    gemm_bias_preload<TN>(p, n0 + wn * WN, lane, g, sk, bias_v);     // older than every DMA: retired by the first tile wait
    f32x4 lnf_cs[LNF ? TN : 1][4];
    f32x4 lnf_st[LNF ? 8 : 1];
    const float* lnf_sp[LNF ? 8 : 1];
    if constexpr (LNF) {
        static_assert(TM == 1 && !CONV && !PROD, "LayerNorm-folded form: one 32-row block per wave (64-row tiles)");
        const float* cp = p.colsum + n0 + wn * WN;
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int q = 0; q < 4; ++q)
#ifdef LNF_ABL_NOCS
                lnf_cs[j][q] = f32x4{0.f, 0.f, 0.f, 0.f};
#else
                lnf_cs[j][q] = *reinterpret_cast<const f32x4*>(cp + j * 32 + 8 * q + 4 * (lane >> 5));
#endif
        // lane l owns row l & 31 of the wave's block and sums its half (l >> 5) of the row's np partial pairs: np / 4 loads of 16 bytes = two pairs each (fold.h::st_off:
        // plane pairs), the 32 lanes of a half reading 32 consecutive rows = 512 contiguous bytes per instruction (np = K / 32 <= 32, np % 4 == 0).  They are REQUESTED
        // behind the prologue's LDS-DMA (below): in front of it they delayed the first tiles of every workgroup (+1..2.7 us per launch on UVLTrack-L's 672-896 workgroups)
        const int np = p.K >> 5, nq = np >> 2;
        int row = m0 + wm * WM + (lane & 31);
        row = row < p.M ? row : p.M - 1;
        const float* sp = p.st_in + st_off((lane >> 5) * (np >> 1), (size_t)row, (size_t)p.M);
#pragma unroll
        for (int i = 0; i < 8; ++i) lnf_sp[i] = i < nq ? sp + (size_t)i * p.M * 4 : reinterpret_cast<const float*>(g_zero_page);
    }

    f32x16 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    const int nk = kspan / BK;
    if (!PROD || producer) {
#pragma unroll
        for (int t = 0; t < NS - 1; ++t)
            if (t < nk) issue(t, true, PRE != 2);          // PRE == 2: the weight halves are in LDS already (the grid barrier drained them)
    }
    if constexpr (LNF) {
        // the rows' partial statistics: eight 16-byte loads per lane, younger than the prologue's tiles and older than every tile requested inside the loop -- the loop's
        // counted waits for tiles 0 .. NS - 2 allow for them (inline asm: hipcc neither moves them across the LDS-DMA nor waits for them; the last tile's vmcnt(0) does)
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int i = 0; i < 8; ++i)
#ifdef LNF_ABL_NOSTATS
            lnf_st[i] = f32x4{1.0f, 2.0f, 1.0f, 2.0f};
#else
            asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(lnf_st[i]) : "v"(lnf_sp[i]) : "memory");
#endif
        __builtin_amdgcn_sched_barrier(0);
    }
    if (PROD && producer) {
        // producer waves: wait for the own pieces of tile kt, meet the consumers, request tile kt + NS - 1 into the stage they left
        for (int kt = 0; kt < nk; ++kt) {
            const int ahead = nk - 1 - kt;
            if (ahead >= NS - 2) wait_vmcnt<LPT * (NS - 2)>();
            else if (NS > 5 && ahead == 3) wait_vmcnt<LPT * 3>();
            else if (NS > 4 && ahead == 2) wait_vmcnt<LPT * 2>();
            else if (NS > 3 && ahead == 1) wait_vmcnt<LPT>();
            else wait_vmcnt<0>();
            __builtin_amdgcn_s_barrier();
            if (kt + NS - 1 < nk) issue(kt + NS - 1);
        }
        __builtin_amdgcn_s_barrier();                        // the epilogue's "every wave has finished reading the ring"
        return;
    }
    GLDS_STAMP(2);
    for (int kt = 0; kt < nk; ++kt) {
        if (!PROD) {
            // tile kt must have landed; tiles kt+1 .. kt+NS-2 may stay in flight
            const int ahead = nk - 1 - kt;
            // split prologue: the first tiles in flight are activation halves only (LPT_A instructions each)
            if (PRE == 2 && kt == 0 && ahead >= NS - 2) wait_vmcnt<LPT_A * (NS - 2)>();
            else if (PRE == 2 && kt == 1 && ahead >= NS - 2) wait_vmcnt<LPT_A * (NS - 3) + LPT>();
            else if (PRE == 2 && kt < NS - 2) wait_vmcnt<0>();
            else if (LNF && kt <= NS - 2 && ahead >= NS - 2) wait_vmcnt<LPT * (NS - 2) + 8>();        // (+ the eight statistics loads behind the prologue's tiles)
            else if (LNF && kt <= NS - 2 && NS > 3 && ahead == 1) wait_vmcnt<LPT + 8>();
            else if (ahead >= NS - 2) wait_vmcnt<LPT * (NS - 2)>();
            else if (NS > 5 && ahead == 3) wait_vmcnt<LPT * 3>();
            else if (NS > 4 && ahead == 2) wait_vmcnt<LPT * 2>();
            else if (NS > 3 && ahead == 1) wait_vmcnt<LPT>();
            else wait_vmcnt<0>();
        }
        if (!(GLDS_ABL & 8)) __builtin_amdgcn_s_barrier();
        if (kt == 0) GLDS_STAMP(3);
        if (kt == 1) GLDS_STAMP(4);
        const char* sA = smem + (kt % NS) * STAGE;
        const char* sB = sA + BM * RB;
        // One 32 x 32 block per wave (the 64 x 64 tile of one-sequence frames: one wave per SIMD, nothing else to run under a stall): the stage's fragment reads
        // go out FIRST, the LDS-DMA of tile kt + NS - 1 is requested under their round trip (an LDS-DMA instruction holds the issuing wave ~55 cycles), then the
        // MFMAs.  tools/glds_trace.py, QKV GEMM of a UVLTrack-B sequence (360 x 2304 x 768, 216 workgroups): K loop 3.39 -> 2.96 us per workgroup, launch
        // 7.17 -> 6.63 us back to back; fc2 slabs (K = 3072 in 4 slices) 8.9 -> 8.5 us; proj slabs 5.45 -> 5.30 us.  (Measured beside it and not kept: the second
        // half of the LDS-DMA behind the MFMAs, or LDS-DMA instructions between the MFMAs, +0.2..0.4 us (an LDS-DMA issued while the wave's MFMAs are in flight is slow); the four waves splitting K with the partial tiles summed through the ring -- a stage is then read from
        // LDS once instead of twice: loop -0.4 us, accumulator set-up and the sum +0.45 us; eight waves, the upper four on k steps 2-3: loop -0.15 us,
        // prologue and hand-over +0.3 us; four PRODUCER waves beside the four consumers (PROD = 4, three or four stages): -0.5 us per launch back to back in
        // isolation, nothing in the frame -- rocprofv3 averages of the frame's GEMM launches 144.7 ms / 145.5 ms per 321 frames, frames/s equal within the
        // repeats.  profiles/r05_glds_loop.md)
        // (Not for the text-branch tiles, NTW: 40 rows against weights nobody has read this frame -- their loop waits for HBM, and a request issued later lands later:
        // tools/glds_trace.py --frame, rider workgroups of one UVLTrack-B sequence 6.4 -> 7.3 us with the reads first.  A deeper ring for the rider's tiles alone,
        // 5 / 6 stages beside the visual tiles' 4: 0 / -3 % in the frame.)
        if constexpr (GLDS_ORDER && !CONV && !PROD && !NTW && TM * TN == 1 && GLDS_ABL == 0) {
            bf16x8 af[KS], bfr[KS];
#pragma unroll
            for (int ks = 0; ks < KS; ++ks) {
                const int chunk = ks * 2 + (lane >> 5);
                af[ks] = *reinterpret_cast<const bf16x8*>(sA + swz(wm * WM + (lane & 31), chunk));
                bfr[ks] = *reinterpret_cast<const bf16x8*>(sB + swz(wn * WN + (lane & 31), chunk));
            }
            __builtin_amdgcn_sched_barrier(0);
            if (kt + NS - 1 < nk) issue(kt + NS - 1);                 // its stage was last read in iteration kt-1
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int ks = 0; ks < KS; ++ks) acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(bfr[ks], af[ks], acc[0][0], 0, 0, 0);
            continue;
        }
        if (!(GLDS_ABL & 4) && !PROD && kt + NS - 1 < nk) issue(kt + NS - 1);            // its stage was last read in iteration kt-1
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) {
            bf16x8 af[TM], bfr[TN];
            const int chunk = ks * 2 + (lane >> 5);
#pragma unroll
            for (int i = 0; i < TM; ++i)
                if (GLDS_ABL & 2) asm volatile("; frag" : "=v"(af[i]));
                else af[i] = *reinterpret_cast<const bf16x8*>(sA + swz(wm * WM + i * 32 + (lane & 31), chunk));
#pragma unroll
            for (int j = 0; j < TN; ++j)
                if (GLDS_ABL & 2) asm volatile("; frag" : "=v"(bfr[j]));
                else bfr[j] = *reinterpret_cast<const bf16x8*>(sB + swz(wn * WN + j * 32 + (lane & 31), chunk));
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int j = 0; j < TN; ++j)
                    if (GLDS_ABL & 1) asm volatile("; use" :: "v"(bfr[j]), "v"(af[i]));
                    else acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(bfr[j], af[i], acc[i][j], 0, 0, 0);   // transposed tile, see gemm_epilogue_lds
        }
    }
    GLDS_STAMP(5);
    static_assert(32 * (WN * 4 + 16) * NW <= NS * STAGE, "epilogue staging fits in the ring");
    // (Round 5, measured and not kept: the split-K slabs of the one-sequence tile stored straight from the accumulator registers -- a lane's register quad is four
    // consecutive columns = a 16-byte store, lanes l / l + 32 the halves of a 32-byte piece -- instead of through this staging: epilogue 0.96 -> 0.43-0.59 us per workgroup in
    // isolation (tools/glds_trace.py), and the one-sequence frame 1422-1449 -> 1351-1377 frames/s, UVLTrack-L x 1 466 -> 456: four 32-byte write-through pieces per line
    // instead of one whole line, and the text rider's K loop beside them went from 4.2 to 6.7 us.  Whole lines per store instruction matter to the OTHER workgroups.)
    if constexpr (LNF) {
        float s1 = 0.f, s2 = 0.f;
#pragma unroll
        for (int i = 0; i < 8; ++i) { s1 += lnf_st[i][0] + lnf_st[i][2]; s2 += lnf_st[i][1] + lnf_st[i][3]; }
        s1 = half_sum(s1);
        s2 = half_sum(s2);
        float mean, rstd;
        st_finish(s1, s2, p.K, p.ln_eps, mean, rstd);
        const float nrm = -rstd * mean;
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int q = 0; q < 4; ++q)
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    acc[0][j][4 * q + e] = rstd * acc[0][j][4 * q + e] + (nrm * lnf_cs[j][q][e] + bias_v[j][q][e]);
                    bias_v[j][q][e] = 0.f;
                }
    }
    gemm_epilogue_lds<TM, TN, WM, WN, EPI, NW>(p, acc, smem, m0, n0, wm, wn, lane, wave, g, sk, bias_v);
    GLDS_STAMP(6);
    GLDS_TRACE_FLUSH();
}

template <int BM, int BN, int WGM, int WGN, int EPI, int NS, bool CONV, bool NTW = false, int BK = 64, int PROD = 0>
__global__ __launch_bounds__(64 * (WGM * WGN + PROD), PROD ? (2 * (WGM * WGN + PROD)) / 4 : 1) void gemm_glds_kernel(const GemmParams p) {
    kernarg_warm<sizeof(GemmParams) + 64>();
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const uint32_t pfs = prefetch_issue<64 * (WGM * WGN + PROD)>(p.pf, p.pf_bytes, (blockIdx.z * gridDim.y + blockIdx.y) * gridDim.x + blockIdx.x, gridDim.x * gridDim.y * gridDim.z);
    gemm_glds_body<BM, BN, WGM, WGN, EPI, NS, CONV, NTW, BK, PROD>(p, blockIdx.x, blockIdx.y, CONV ? blockIdx.z : 0, smem);
    prefetch_retire(pfs);
}

// ------------------------------------------------------------------------------------------------
// Phase-pipelined 256-wide GEMM for frames of many sequences (M >= ~4k rows): BM x 256 tile, 8 waves as 2 (M) x 4 (N), a wave owns
// (BM/2) x 64 of the output (128 x 64 at BM = 256: 128 accumulator registers, 0.75 fragment reads and 0.25 LDS-DMA instructions per
// MFMA -- the loop of gemm_glds_body pays 1.0 and 0.5 at 128 x 128).  What bounds gemm_glds_body at these sizes is not issue alone
// but its shape (profiles/r03_gemm_pipe.md, ISA of the 256 x 256 instantiation): two ring stages mean `s_waitcnt vmcnt(0)` in front
// of EVERY K step -- a K step costs a whole L2 -> LDS round trip -- and hipcc emits the fragment reads four at a time, each group
// followed by `lgkmcnt(0)` and four MFMAs, so neither memory level overlaps the matrix pipe inside a wave.  Here:
//   * a K tile (64 wide) is FOUR phases of 8 MFMAs (one quadrant of the wave's block: A rows lo/hi x B columns lo/hi, visited
//     lo-lo, lo-hi, hi-hi, hi-lo so that one operand's fragments stay in registers between phases);
//   * the two wave groups of a workgroup (wm = 0 / 1: waves w and w + 4 share a SIMD) run HALF A PHASE APART -- group 1 executes one
//     extra s_barrier up front -- so on every SIMD one wave issues its phase's MFMAs (s_setprio 1) while the other issues the next
//     phase's fragment reads and LDS-DMA, two raw s_barriers per phase keeping them in that relation;
//   * a K tile is staged as four GROUPS of rows -- exactly what each phase reads (A-lo, B-lo | B-hi | A-hi | -) -- into two 64-KB
//     buffers; every phase issues one group (2 LDS-DMA instructions per wave) SIX groups ahead of the one read next and waits with a
//     COUNTED vmcnt that leaves the four youngest groups (64 KB per CU) in flight: a group has >= 4 phases to land, and no wait in
//     the loop ever drains the queue.  Write-after-read: a group's slot is re-filled two phases after its last read (the reads of
//     phase p by BOTH wave groups are complete at the second barrier after p); read-after-write: a wave waits for its own pieces in
//     phase p, all waves have done so by the end of p, the group is read in p + 1.
// Epilogue, tile order and the split of rows over lanes are those of gemm_glds_body (same gemm_epilogue_lds).
// ------------------------------------------------------------------------------------------------
#ifdef GEMM_TRACE
// development build only (tools/gemm_trace.py, -DGEMM_TRACE into a separate .so): shader-clock stamps of wave 0 (wave group 0) and wave 4
// (group 1) of workgroup 0, four per phase -- phase start | counted wait done | barrier + fragment reads done | MFMAs issued -- taken with
// s_memtime into SGPRs and only READ behind the next lgkmcnt(0) the loop has anyway, so the stamps add no wait of their own.
__device__ unsigned int g_gemm_trace[2 * 64 * 4 * 4 + 16];
#define GT_STAMP(v) asm volatile("s_memtime %0" : "=s"(v)::"memory")
#else
#define GT_STAMP(v) do { } while (0)
#endif
// VAR 1 (the product form): the phase's two LDS-DMA instructions are issued BETWEEN its MFMAs (behind the 2nd and the 5th) instead of
// in front of the barrier -- an LDS-DMA instruction costs the issuing wave 60-120 cycles, which the other wave group's 256-cycle MFMA
// block cannot hide together with up to 12 fragment reads; between MFMAs it rides on the matrix pipe's own latency (+2..5 % at
// 256 x 256, +5..9 % at 128 x 256; VAR 0 is still instantiable but has no cfg number any more: 32 / 33 are now the 32x32x16 forms).
// position L of the grouped tile order (groups of group_m M-tiles x all N-tiles, M fastest inside a group) -> (mt, nt)
__device__ __forceinline__ void pipe_tile_of(const GemmParams& p, const int L, const int MT, const int NT, int& mt, int& nt) {
    const int group_m = p.group_m;
    const int gsz = group_m * NT, gi = (int)fd_div((uint32_t)L, p.fd_gsz), within = L - gi * gsz;
    const int gm = min(group_m, MT - gi * group_m);
    nt = (int)fd_div((uint32_t)within, gm == group_m ? p.fd_gm : p.fd_gml);
    mt = gi * group_m + (within - nt * gm);
}

// the phase-pipelined K loop over all nk K tiles of output tile (mt, nt), then the epilogue
template <int BM, int EPI, int VAR = 0, bool MI16 = true>
__device__ __forceinline__ void gemm_pipe_tile(const GemmParams& p, const int mt, const int nt, const int nk, char* smem) {
    constexpr int BN = 256, NW = 8, WM = BM / 2, WN = 64, TM = WM / 32, TN = 2, HB = TM / 2;   // HB: A blocks of a half (lo / hi)
    constexpr int STAGE = (BM + BN) * 128;                   // one K tile: A rows then W rows, 128 bytes each
    constexpr int NA = BM / 128, NB = 2;                     // LDS-DMA instructions per wave for an A group / a B group (8 rows each)
    constexpr int LPT = 2 * NA + 2 * NB;                     // ... per K tile
    static_assert(BM == 256 || BM == 128, "tile height");
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 2, wn = wave & 3;
    const int m0 = mt * BM, n0 = nt * BN;
    const char* a_base = reinterpret_cast<const char*>(p.A + (size_t)m0 * p.lda);
    const char* w_base = reinterpret_cast<const char*>(p.W + (size_t)n0 * p.ldw);

    // ---- LDS-DMA plan.  Group g of a K tile: 0 = A-lo (rows wm' * WM + [0, WM/2) of both wave groups), 1 = B-lo (rows wn' * 64 + [0, 32)
    // of the four wave columns), 2 = B-hi, 3 = A-hi.  Instruction n of a group, issued by this wave, fills 8 consecutive tile rows.
    uint32_t loff[4][2];                                     // per-lane source byte offset (row * ld * 2 + swizzled 16-byte chunk)
    int dst[4][2];                                           // wave-uniform LDS byte offset of the 8-row piece inside a K-tile buffer
#pragma unroll
    for (int g = 0; g < 4; ++g) {
        const bool isA = (g == 0 || g == 3), hi = (g >= 2);
#pragma unroll
        for (int n = 0; n < (isA ? NA : NB); ++n) {
            const int gr0 = 8 * (wave + NW * n);             // first row of the piece within the group
            const int R0 = isA ? (gr0 / (WM / 2)) * WM + (hi ? WM / 2 : 0) + gr0 % (WM / 2) : (gr0 / 32) * 64 + (hi ? 32 : 0) + gr0 % 32;
            const int R = R0 + (lane >> 3);
            const int chunk = (lane & 7) ^ ((R >> 1) & 7);
            dst[g][n] = (isA ? 0 : BM * 128) + R0 * 128;
            if (isA) {
                int gmr = m0 + R;
                gmr = gmr < p.M ? gmr : p.M - 1;
                loff[g][n] = (uint32_t)(gmr - m0) * (uint32_t)p.lda * 2u + (uint32_t)chunk * 16u;
            } else {
                loff[g][n] = (uint32_t)R * (uint32_t)p.ldw * 2u + (uint32_t)chunk * 16u;
            }
        }
    }
    auto pin = [](const char* q) __attribute__((always_inline)) {
        const uint64_t u = reinterpret_cast<uint64_t>(q);
        const uint32_t lo = __builtin_amdgcn_readfirstlane((uint32_t)u), hi = __builtin_amdgcn_readfirstlane((uint32_t)(u >> 32));
        return reinterpret_cast<const char*>(((uint64_t)hi << 32) | lo);
    };
    auto issue_part = [&](auto G, int kt, int n0_, int n1_) __attribute__((always_inline)) {      // instructions [n0_, n1_) of group G of K tile kt
        constexpr int g = decltype(G)::value;
        constexpr bool isA = (g == 0 || g == 3);
        const char* gb = pin((isA ? a_base : w_base) + (size_t)kt * 128);
        char* st = smem + (kt & 1) * STAGE;
#pragma unroll
        for (int n = 0; n < (isA ? NA : NB); ++n)
            if (n >= n0_ && n < n1_)
                __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(gb + loff[g][n]),
                                                 (__attribute__((address_space(3))) void*)(st + dst[g][n]), 16, 0, 0);
    };
    auto issue = [&](auto G, int kt) __attribute__((always_inline)) { issue_part(G, kt, 0, 2); };      // group G of K tile kt into buffer kt & 1
    using I0 = std::integral_constant<int, 0>; using I1 = std::integral_constant<int, 1>;
    using I2 = std::integral_constant<int, 2>; using I3 = std::integral_constant<int, 3>;

    // ---- fragment addresses: row = block row + (lane & 31), 16-byte chunk 2 ks + (lane >> 5), XOR-swizzled by the row
    int a_off[4], b_off[4];
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) {
        a_off[ks] = pipe_frag_off<MI16>(wm * WM, ks, lane);
        b_off[ks] = BM * 128 + pipe_frag_off<MI16>(wn * WN, ks, lane);
    }

    f32x16 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
    bf16x8 af[HB][4], bl[4], bh[4];

    // nk >= 2 (launcher)
    // the tile's 256 bias values: requested before the first DMA (so the first tile wait retires them), parked in LDS behind the
    // ring -- 32 registers per lane cannot be held across the loop, and fetched behind it they cost a dependent L2 round trip per tile
    float* sbias = reinterpret_cast<float*>(smem + 2 * STAGE);
    // (load and LDS store are inline asm: hipcc would drain the whole DMA queue -- s_waitcnt vmcnt(0) -- in front of the first use of
    // an ordinary load's result and in front of an LDS store while LDS-DMA is in flight; the counted wait below covers the load)
    f32x4 bias_in;
    if (wave == 0) {
        const float* bsrc = (p.bias ? p.bias + n0 : reinterpret_cast<const float*>(g_zero_page)) + lane * 4;
        asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(bias_in) : "v"(bsrc) : "memory");
    }
    // prologue: K tile 0 and the first two groups of K tile 1; A-lo / B-lo of tile 0 must have landed before phase 0 reads them
    issue(I0{}, 0); issue(I1{}, 0); issue(I2{}, 0); issue(I3{}, 0); issue(I0{}, 1); issue(I1{}, 1);
    wait_vmcnt<LPT>();
    if (wave == 0) {
        const uint32_t sb_addr = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) char*)smem + 2 * STAGE + lane * 16;
        asm volatile("ds_write_b128 %0, %1" ::"v"(sb_addr), "v"(bias_in) : "memory");
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    if (wm == 1) __builtin_amdgcn_s_barrier();               // wave group 1 runs half a phase behind group 0 from here on

#ifdef GEMM_TRACE
    unsigned long long gt[4][4] = {};                        // [quadrant][stamp]: every quadrant has its own SGPRs, nothing is copied
    int gt_n = 0;                                            // phases completed
    const bool gt_on = blockIdx.x == 0 && (wave == 0 || wave == 4);
    unsigned int* gt_lds = reinterpret_cast<unsigned int*>(smem + 2 * STAGE + 1024) + (wave >> 2) * (64 * 4 * 4);
#endif
    // one phase: fragment reads of quadrant Q | the group due this phase | counted wait | barrier | 8 MFMAs | barrier
    auto phase = [&](auto Q, auto ISS, auto VM, int t) __attribute__((always_inline)) {
        constexpr int q = decltype(Q)::value;
        constexpr bool iss = decltype(ISS)::value != 0;
        constexpr int vm = decltype(VM)::value;
        const char* sb = smem + (t & 1) * STAGE;
        GT_STAMP(gt[q][0]);
        if constexpr (q == 0) {
#pragma unroll
            for (int ks = 0; ks < 4; ++ks) bl[ks] = *reinterpret_cast<const bf16x8*>(sb + b_off[ks]);
        }
        __builtin_amdgcn_sched_barrier(0);
        if constexpr (q == 0 || q == 2) {
#pragma unroll
            for (int i = 0; i < HB; ++i)
#pragma unroll
                for (int ks = 0; ks < 4; ++ks) af[i][ks] = *reinterpret_cast<const bf16x8*>(sb + a_off[ks] + ((q == 2 ? HB : 0) + i) * 4096);
        }
        if constexpr (q == 1) {
#pragma unroll
            for (int ks = 0; ks < 4; ++ks) bh[ks] = *reinterpret_cast<const bf16x8*>(sb + b_off[ks] + 4096);
        }
        __builtin_amdgcn_sched_barrier(0);
        auto issue_now = [&](int n0_, int n1_) __attribute__((always_inline)) {   // group (q + 2) % 4 of K tile t + 1 (q < 2) or t + 2
            if constexpr (q == 0) issue_part(I2{}, t + 1, n0_, n1_);
            if constexpr (q == 1) issue_part(I3{}, t + 1, n0_, n1_);
            if constexpr (q == 2) issue_part(I0{}, t + 2, n0_, n1_);
            if constexpr (q == 3) issue_part(I1{}, t + 2, n0_, n1_);
        };
        constexpr int cnt_g = (q == 1 || q == 2) ? NA : NB;  // instructions of the group this phase issues
        if constexpr (VAR == 0 && iss) issue_now(0, 2);
        // VAR 1 waits BEFORE its issue: one group fewer may be outstanding
        constexpr int vm_eff = (VAR == 1 && iss) ? vm - cnt_g : vm;
        if constexpr (vm >= 0) wait_vmcnt<(vm_eff >= 0 ? vm_eff : 0)>();
        GT_STAMP(gt[q][1]);
        __builtin_amdgcn_sched_barrier(0);
        __builtin_amdgcn_s_barrier();
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_sched_barrier(0);
#ifdef GEMM_TRACE
        if (gt_on && gt_n > 0 && gt_n <= 256 && lane == 0) {      // the PREVIOUS phase's four stamps are valid behind the lgkmcnt(0) above
            unsigned int* e = gt_lds + (gt_n - 1) * 4;
            constexpr int pq = (q + 3) & 3;
            e[0] = (unsigned int)gt[pq][0]; e[1] = (unsigned int)gt[pq][1]; e[2] = (unsigned int)gt[pq][2]; e[3] = (unsigned int)gt[pq][3];
        }
#endif
        GT_STAMP(gt[q][2]);
        __builtin_amdgcn_s_setprio(1);
        constexpr int i0 = (q >= 2) ? HB : 0, j = (q == 1 || q == 2) ? 1 : 0;
#pragma unroll
        for (int ks = 0; ks < 4; ++ks)
#pragma unroll
            for (int i = 0; i < HB; ++i) {
                pipe_mfma<MI16>(acc[i0 + i][j], j ? bh : bl, af[i], ks);
                if constexpr (VAR == 1 && iss) {
                    constexpr int NM = 4 * HB;               // MFMAs of the phase
                    const int mi = ks * HB + i;
                    if (mi == NM / 4) { __builtin_amdgcn_sched_barrier(0); issue_now(0, 1); __builtin_amdgcn_sched_barrier(0); }
                    if (mi == (5 * NM) / 8) { __builtin_amdgcn_sched_barrier(0); issue_now(1, 2); __builtin_amdgcn_sched_barrier(0); }
                }
            }
        __builtin_amdgcn_s_setprio(0);
        GT_STAMP(gt[q][3]);
        __builtin_amdgcn_sched_barrier(0);
        __builtin_amdgcn_s_barrier();
#ifdef GEMM_TRACE
        ++gt_n;
#endif
    };
    using Y = std::integral_constant<int, 1>; using N_ = std::integral_constant<int, 0>;
    using VF = std::integral_constant<int, LPT>;             // steady state: the four youngest groups stay in flight
    for (int t = 0; t < nk - 2; ++t) {
        phase(I0{}, Y{}, VF{}, t); phase(I1{}, Y{}, VF{}, t); phase(I2{}, Y{}, VF{}, t); phase(I3{}, Y{}, VF{}, t);
    }
    {   // last two K tiles: nothing left to issue after (nk - 2, phase 1); the allowed in-flight count shrinks group by group
        const int t = nk - 2;
        phase(I0{}, Y{}, VF{}, t); phase(I1{}, Y{}, VF{}, t);
        phase(I2{}, N_{}, std::integral_constant<int, NB + NB + NA>{}, t);
        phase(I3{}, N_{}, std::integral_constant<int, NB + NA>{}, t);
        phase(I0{}, N_{}, std::integral_constant<int, NA>{}, t + 1);
        phase(I1{}, N_{}, std::integral_constant<int, 0>{}, t + 1);
        phase(I2{}, N_{}, std::integral_constant<int, -1>{}, t + 1);
        phase(I3{}, N_{}, std::integral_constant<int, -1>{}, t + 1);
    }
    if (wm == 0) __builtin_amdgcn_s_barrier();               // group 0 waits for group 1's last phase
#ifdef GEMM_TRACE
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    if (gt_on && lane == 0) {
        unsigned int* e = gt_lds + (gt_n - 1) * 4;           // the last phase is quadrant 3
        e[0] = (unsigned int)gt[3][0]; e[1] = (unsigned int)gt[3][1]; e[2] = (unsigned int)gt[3][2]; e[3] = (unsigned int)gt[3][3];
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        const int n = gt_n < 256 ? gt_n : 256;
        for (int i = 0; i < n * 4; ++i) g_gemm_trace[(wave >> 2) * (64 * 4 * 4) + i] = gt_lds[i];
        g_gemm_trace[2 * 64 * 4 * 4 + (wave >> 2)] = (unsigned int)n;
    }
#endif

    f32x4 bias_v[TN][4];
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
        for (int q = 0; q < 4; ++q) bias_v[j][q] = *reinterpret_cast<const f32x4*>(sbias + wn * WN + j * 32 + (MI16 ? 16 * (q & 1) + 4 * (lane >> 4) : 8 * q + 4 * (lane >> 5)));
    static_assert(32 * (WN * 4 + 16) * NW <= 2 * STAGE, "epilogue staging fits in the two buffers");
    gemm_epilogue_lds<TM, TN, WM, WN, EPI, NW, MI16>(p, acc, smem, m0, n0, wm, wn, lane, wave, 0, 0, bias_v);
}

template <int BM, int EPI, int VAR = 0, bool MI16 = true>
__device__ __forceinline__ void gemm_pipe_body(const GemmParams& p, const int bx, char* smem) {
    const int MT = (p.M + BM - 1) / BM, NT = p.N / 256;
    int nt, mt;
    {   // grouped tile order cut into 8 contiguous runs, one per XCD (see gemm_glds_body)
        const int xcd = bx & 7, idx = bx >> 3;
        const int T = MT * NT, base = T >> 3, rem = T & 7;
        const int cnt = base + (xcd < rem ? 1 : 0);
        if (idx >= cnt) return;
        pipe_tile_of(p, xcd * base + (xcd < rem ? xcd : rem) + idx, MT, NT, mt, nt);
    }
    gemm_pipe_tile<BM, EPI, VAR, MI16>(p, mt, nt, p.K / 64, smem);
}

// ------------------------------------------------------------------------------------------------
// 128 x 256 tile of the same family with TWO phases of 8 MFMAs per K tile and THREE 48-KB buffers (cfg 31).  gemm_pipe_body at BM = 128
// would have four phases of FOUR MFMAs (measured first: 1-3 % slower than this form on every shape of the frames; a hand-over between
// the two wave groups costs the same ~110 cycles whether the MFMA block is 128 or 256 cycles long).  Here a wave (64 x 64 of the
// output) reads both A blocks and B-lo in phase 0 (12 fragment reads, 8 MFMAs into acc[.][0]) and B-hi in phase 1 (4 reads, 8 MFMAs
// into acc[.][1]); a K tile is staged as three row groups -- A (128 rows), B-lo, B-hi (128 rows each: the lo / hi 32 columns of the
// four wave columns) -- TWO K tiles ahead: phase 0 of tile t issues A and B-lo of tile t + 2 (4 LDS-DMA instructions per wave, between
// the MFMAs), phase 1 issues B-hi of t + 2.  Three buffers because a group read in phase p may be refilled in p + 2 at the earliest and
// must be waited for in the phase before its read: with two buffers a group would have ONE phase to land.  Counted wait: 6 instructions
// may stay outstanding in every steady-state phase (derived as in gemm_pipe_body: wait in p, read in p + 1, wait placed before the
// phase's own issue).  Epilogue, tile order, bias row in LDS: as gemm_pipe_body.
// ------------------------------------------------------------------------------------------------
// PRE (round 5; the in-place f32 residual epilogue, K >= 12 tiles): the read half of the read-modify-write leaves the epilogue.  These GEMMs are
// one-round launches (8 UVLTrack-L sequences: 220 tiles on 256 CUs, one workgroup per CU), so every tile reaches its epilogue at the same moment
// and 2 x 28 MB of residual traffic ran with no MFMA beside it: +7 us (proj) / +8 us (fc2) over a bf16 store (profiles/r04_gemm_streamk.md).  A
// wave's share of the residual tile is 64 x 64 f32 = sixteen 16-byte loads per lane, and the kernel has the registers for them (186 of 256):
// they are issued ONE PER PHASE over eight K tiles, behind the phase's LDS-DMA, and stay in the counted vmcnt queue like the tile loads -- a
// load has three phases (~1200 cycles) to land before a wait covers it.  Counted waits of the window: the steady state may leave the six
// youngest LDS-DMA outstanding; a residual load issued in one of the last three phases is younger than the group a wait is for, so the count
// grows by the number of such loads: 6, 7, 8, then 9 until the window ends, 9, 8, 7, 6 behind it (derivation at the phase list below).
// Same operands, same order of additions as the epilogue's own load: bit-identical results.
template <int EPI, bool MI16 = true, bool PRE = false>
__device__ __forceinline__ void gemm_pipe128_body(const GemmParams& p, const int bx, char* smem) {
    static_assert(!PRE || EPI == EPI_F32, "the residual window belongs to the f32 epilogue");
    constexpr int BM = 128, BN = 256, NW = 8, WM = 64, WN = 64, TM = 2, TN = 2, NBUF = 3;
    constexpr int STAGE = (BM + BN) * 128;                   // 48 KB: A rows then W rows
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 2, wn = wave & 3;
    const int MT = (p.M + BM - 1) / BM, NT = p.N / BN;
    // split-K (round 5: the text rider of the residual pair launch; f32 slabs, folded by the LayerNorm that follows): slice sk of tile (mt, nt) is
    // workgroup sk * MT * NT + its tile index -- with one slice this is the plain map, bit for bit
    const int nsk = p.splitk > 1 ? p.splitk : 1;
    int nt, mt, sk;
    {
        const int xcd = bx & 7, idx = bx >> 3;
        const int Tt = MT * NT, T = Tt * nsk, base = T >> 3, rem = T & 7;
        const int cnt = base + (xcd < rem ? 1 : 0);
        if (idx >= cnt) return;
        const int L0 = xcd * base + (xcd < rem ? xcd : rem) + idx;
        sk = L0 >= Tt ? (nsk == 2 ? 1 : L0 / Tt) : 0;      // (one slice, or the rider's two: no division on the product path)
        pipe_tile_of(p, L0 - sk * Tt, MT, NT, mt, nt);
    }
    const int m0 = mt * BM, n0 = nt * BN;
    const int nk = p.kspan >> 6;                            // K tiles of this slice: >= 2 (launcher); PRE: >= 12
    const char* a_base = reinterpret_cast<const char*>(p.A + (size_t)m0 * p.lda + (size_t)sk * nk * 64);
    const char* w_base = reinterpret_cast<const char*>(p.W + (size_t)n0 * p.ldw + (size_t)sk * nk * 64);

    // group 0 = A (tile rows 0..127), 1 = B-lo (rows wn' * 64 + [0, 32)), 2 = B-hi; instruction n (0, 1) of a group fills 8 rows
    uint32_t loff[3][2];
    int dst[3][2];
#pragma unroll
    for (int g = 0; g < 3; ++g)
#pragma unroll
        for (int n = 0; n < 2; ++n) {
            const int gr0 = 8 * (wave + NW * n);
            const int R0 = g == 0 ? gr0 : (gr0 / 32) * 64 + (g == 2 ? 32 : 0) + gr0 % 32;
            const int R = R0 + (lane >> 3);
            const int chunk = (lane & 7) ^ ((R >> 1) & 7);
            dst[g][n] = (g == 0 ? 0 : BM * 128) + R0 * 128;
            if (g == 0) {
                int gmr = m0 + R;
                gmr = gmr < p.M ? gmr : p.M - 1;
                loff[g][n] = (uint32_t)(gmr - m0) * (uint32_t)p.lda * 2u + (uint32_t)chunk * 16u;
            } else {
                loff[g][n] = (uint32_t)R * (uint32_t)p.ldw * 2u + (uint32_t)chunk * 16u;
            }
        }
    auto pin = [](const char* q) __attribute__((always_inline)) {
        const uint64_t u = reinterpret_cast<uint64_t>(q);
        const uint32_t lo = __builtin_amdgcn_readfirstlane((uint32_t)u), hi = __builtin_amdgcn_readfirstlane((uint32_t)(u >> 32));
        return reinterpret_cast<const char*>(((uint64_t)hi << 32) | lo);
    };
    auto issue1 = [&](auto G, int kt, int buf, int n) __attribute__((always_inline)) {       // instruction n of group G of K tile kt
        constexpr int g = decltype(G)::value;
        const char* gb = pin((g == 0 ? a_base : w_base) + (size_t)kt * 128);
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(gb + loff[g][n]),
                                         (__attribute__((address_space(3))) void*)(smem + buf * STAGE + dst[g][n]), 16, 0, 0);
    };
    using I0 = std::integral_constant<int, 0>; using I1 = std::integral_constant<int, 1>; using I2 = std::integral_constant<int, 2>;
    auto issue_tile = [&](int kt, int buf) __attribute__((always_inline)) {
        issue1(I0{}, kt, buf, 0); issue1(I0{}, kt, buf, 1); issue1(I1{}, kt, buf, 0); issue1(I1{}, kt, buf, 1); issue1(I2{}, kt, buf, 0); issue1(I2{}, kt, buf, 1);
    };

    int a_off[4], b_off[4];
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) {
        a_off[ks] = pipe_frag_off<MI16>(wm * WM, ks, lane);
        b_off[ks] = BM * 128 + pipe_frag_off<MI16>(wn * WN, ks, lane);
    }
    f32x16 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
    bf16x8 af[2][4], bl[4], bh[4];

    // residual rows of this wave's 64 x 64 block in the epilogue's write-out order: 16 lanes per row, 4 rows per instruction, 8 per 32-row block
    constexpr int R_NIT = 8;
    f32x4 res[PRE ? TM * R_NIT : 1];
    int r_first[TM];
    RowMap r_map[TM];
    if constexpr (PRE) {
#pragma unroll
        for (int i = 0; i < TM; ++i) {
            r_first[i] = min(m0 + wm * WM + i * 32, p.M - 1);        // a block past M reads (and never stores) the last valid row
            r_map[i] = rowmap_of(r_first[i], p.rpb, p.fd_rpb);
        }
    }
    auto res_load = [&](auto RI) __attribute__((always_inline)) {
        constexpr int ri = decltype(RI)::value;
        if constexpr (PRE && ri >= 0) {
            constexpr int i = ri / R_NIT, it = ri % R_NIT;
            const int row = m0 + wm * WM + i * 32 + it * 4 + (lane >> 4);
            int b, rem;
            rowmap_at(r_map[i], r_first[i], (row < p.M ? row : p.M - 1) - r_first[i], b, rem);
            res[ri] = *reinterpret_cast<const f32x4*>(reinterpret_cast<const float*>(p.C) + ((size_t)b * p.obs + p.oro + rem) * p.ldc + n0 + wn * WN + (lane & 15) * 4);
        }
    };

    f32x4 bias_in;
    if (wave == 0) {
        const float* bsrc = ((p.bias && sk == 0) ? p.bias + n0 : reinterpret_cast<const float*>(g_zero_page)) + lane * 4;      // the bias goes into slab 0
        asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(bias_in) : "v"(bsrc) : "memory");
    }
    issue_tile(0, 0);
    issue_tile(1, 1);
    wait_vmcnt<8>();                                         // A and B-lo of tile 0 (and the bias row in front of them) have landed
    if (wave == 0) {
        const uint32_t sb_addr = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) char*)smem + NBUF * STAGE + lane * 16;
        asm volatile("ds_write_b128 %0, %1" ::"v"(sb_addr), "v"(bias_in) : "memory");
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    if (wm == 1) __builtin_amdgcn_s_barrier();               // wave group 1 runs half a phase behind group 0

    // H = 0: A + B-lo | H = 1: B-hi.  ISS: issue tile t + 2's share (H = 0: A, B-lo; H = 1: B-hi) into buffer nb between the MFMAs.
    auto phase = [&](auto H, auto ISS, auto VM, int t, int cb, int nb, auto RI) __attribute__((always_inline)) {
        constexpr int h = decltype(H)::value;
        constexpr bool iss = decltype(ISS)::value != 0;
        constexpr int vm = decltype(VM)::value;
        const char* sb = smem + cb * STAGE;
        if constexpr (h == 0) {
#pragma unroll
            for (int ks = 0; ks < 4; ++ks) bl[ks] = *reinterpret_cast<const bf16x8*>(sb + b_off[ks]);
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int ks = 0; ks < 4; ++ks) af[i][ks] = *reinterpret_cast<const bf16x8*>(sb + a_off[ks] + i * 4096);
        } else {
#pragma unroll
            for (int ks = 0; ks < 4; ++ks) bh[ks] = *reinterpret_cast<const bf16x8*>(sb + b_off[ks] + 4096);
        }
        __builtin_amdgcn_sched_barrier(0);
        if constexpr (vm >= 0) wait_vmcnt<(vm >= 0 ? vm : 0)>();
        __builtin_amdgcn_sched_barrier(0);
        __builtin_amdgcn_s_barrier();
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_sched_barrier(0);
        __builtin_amdgcn_s_setprio(1);
#pragma unroll
        for (int ks = 0; ks < 4; ++ks)
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                pipe_mfma<MI16>(acc[i][h], h ? bh : bl, af[i], ks);
                if constexpr (iss) {
                    const int mi = ks * 2 + i;
                    __builtin_amdgcn_sched_barrier(0);
                    if constexpr (h == 0) {
                        if (mi == 0) issue1(I0{}, t + 2, nb, 0);
                        if (mi == 2) issue1(I0{}, t + 2, nb, 1);
                        if (mi == 4) issue1(I1{}, t + 2, nb, 0);
                        if (mi == 6) issue1(I1{}, t + 2, nb, 1);
                    } else {
                        if (mi == 1) issue1(I2{}, t + 2, nb, 0);
                        if (mi == 5) issue1(I2{}, t + 2, nb, 1);
                    }
                    if (mi == 7) res_load(RI);               // behind the phase's LDS-DMA: the youngest entry of the queue
                    __builtin_amdgcn_sched_barrier(0);
                }
            }
        __builtin_amdgcn_s_setprio(0);
        __builtin_amdgcn_sched_barrier(0);
        __builtin_amdgcn_s_barrier();
    };
    using Y = std::integral_constant<int, 1>; using N_ = std::integral_constant<int, 0>;
    using V6 = std::integral_constant<int, 6>;
    using NR = std::integral_constant<int, -1>;
    int cb = 0;                                              // buffer of K tile t
    int t = 0;
    auto tile = [&](auto VM0, auto VM1, auto R0, auto R1) __attribute__((always_inline)) {
        const int nb = cb == 0 ? 2 : cb - 1;                 // buffer of K tile t + 2 = (t + 2) % 3 = (t - 1) % 3
        phase(I0{}, Y{}, VM0, t, cb, nb, R0);
        phase(I1{}, Y{}, VM1, t, cb, nb, R1);
        cb = cb == 2 ? 0 : cb + 1;
        ++t;
    };
    for (; t < nk - (PRE ? 12 : 2);) tile(V6{}, V6{}, NR{}, NR{});
    if constexpr (PRE) {
        // The wait of phase 0 of tile t is for B-hi of tile t (issued in phase 1 of t - 2): younger than it are the 6 LDS-DMA of tile t + 1 and the
        // residual loads of phases (1, t - 2), (0, t - 1), (1, t - 1); the wait of phase 1 is for A / B-lo of t + 1 (phase 0 of t - 1): younger are
        // B-hi of t + 1, A / B-lo of t + 2 and the residual loads of phases (0, t - 1), (1, t - 1), (0, t).  Window = tiles w0 .. w0 + 7, one load
        // per phase: the counts below are 6 + the number of those phases inside the window.
        using V7 = std::integral_constant<int, 7>; using V8 = std::integral_constant<int, 8>; using V9 = std::integral_constant<int, 9>;
#define RI_(k) std::integral_constant<int, (k)>{}
        tile(V6{}, V7{}, RI_(0), RI_(1));
        tile(V8{}, V9{}, RI_(2), RI_(3));
        tile(V9{}, V9{}, RI_(4), RI_(5));
        tile(V9{}, V9{}, RI_(6), RI_(7));
        tile(V9{}, V9{}, RI_(8), RI_(9));
        tile(V9{}, V9{}, RI_(10), RI_(11));
        tile(V9{}, V9{}, RI_(12), RI_(13));
        tile(V9{}, V9{}, RI_(14), RI_(15));
#undef RI_
        tile(V9{}, V8{}, NR{}, NR{});
        tile(V7{}, V6{}, NR{}, NR{});
    }
    {   // last two K tiles: nothing left to issue
        phase(I0{}, N_{}, V6{}, nk - 2, cb, 0, NR{});
        phase(I1{}, N_{}, std::integral_constant<int, 2>{}, nk - 2, cb, 0, NR{});
        cb = cb == 2 ? 0 : cb + 1;
        phase(I0{}, N_{}, std::integral_constant<int, 0>{}, nk - 1, cb, 0, NR{});
        phase(I1{}, N_{}, std::integral_constant<int, -1>{}, nk - 1, cb, 0, NR{});
    }
    if (wm == 0) __builtin_amdgcn_s_barrier();               // group 0 waits for group 1's last phase

    const float* sbias = reinterpret_cast<const float*>(smem + NBUF * STAGE);
    f32x4 bias_v[TN][4];
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
        for (int q = 0; q < 4; ++q) bias_v[j][q] = *reinterpret_cast<const f32x4*>(sbias + wn * WN + j * 32 + (MI16 ? 16 * (q & 1) + 4 * (lane >> 4) : 8 * q + 4 * (lane >> 5)));
    static_assert(32 * (WN * 4 + 16) * NW <= NBUF * STAGE, "epilogue staging fits in the buffers");
    gemm_epilogue_lds<TM, TN, WM, WN, EPI, NW, MI16, PRE>(p, acc, smem, m0, n0, wm, wn, lane, wave, 0, sk, bias_v, res);
}

// the residual window needs the in-place form of the f32 epilogue and 12 K tiles (8 window + 2 behind it + the 2 that issue nothing).  Taken by default
// up to K = 2048 (tools/res_epilogue_probe.py, 8 UVLTrack-L sequences, us: proj K = 1024 bf16 store 19.4 / f32 store 22.2 / rows loaded in the
// epilogue 25.9 / in the loop 24.6, with rotating operands 28.6 -> 26.4; fc2 K = 4096 53.2 / 57.7 / 59.8 / 61.7: there the epilogue's read is 2 us of
// a 60-us kernel and 28 MB of extra requests inside eight K tiles of a loop that already streams 2.5 TB/s cost more than that); res_pre = 2 forces it
// (returns the PRE template value of the kernel WRAPPERS: 1, or 2 for K > 2048 -- the same code under a second symbol, so that rocprofv3 and bench.py keep telling
// proj from fc2)
static int pipe128_pre_ok(const GemmParams& p) {
    const int want = tune_get(p.tune, &uvl_tuning::res_pre, 1);
    // ... and at every K where the launch is a single round of tiles (fc2 of 8 UVLTrack-L sequences: 220 tiles): in the FRAME its operand arrives cold, the loop is slower than
    // in the probe above and the window pays -- same box, interleaved tools/ab_tune.py res_pre -1 2: 1333.7 -> 1338.2 and 1266.3 -> 1273.1 frames/s (+0.3 / +0.5 %) on two
    // boxes; many-round launches stay with K <= 2048 (32 UVLTrack-L sequences 1489.1 -> 1488.2, 32 UVLTrack-B sequences 6691 -> 6679).
    const long tiles = (long)((p.M + 127) / 128) * (p.N / 256);
    const bool ok = p.epi == EPI_F32 && p.accumulate && p.splitk <= 1 && p.K >= 12 * 64 && want != 0 && (p.K <= 2048 || want == 2 || tiles <= 256);
    return ok ? (p.K > 2048 ? 2 : 1) : 0;
}

template <int EPI, bool MI16, int PRE = 0>
__global__ __launch_bounds__(512) void gemm_pipe128_kernel(const GemmParams p) {
    kernarg_warm<sizeof(GemmParams)>();
    extern __shared__ __attribute__((aligned(16))) char smem[];
    gemm_pipe128_body<EPI, MI16, PRE != 0>(p, blockIdx.x, smem);
}

template <int EPI, bool MI16 = true, int PRE = 0>
static hipError_t launch_pipe128(const GemmParams& p_in, hipStream_t s) {
    if constexpr (EPI == EPI_F32 && MI16 && !PRE) {
        const int pre = pipe128_pre_ok(p_in);
        if (pre == 1) return launch_pipe128<EPI, MI16, 1>(p_in, s);
        if (pre == 2) return launch_pipe128<EPI, MI16, 2>(p_in, s);
    }
    GemmParams p = p_in;
    if (p.N % 256 != 0 || p.K < 128 || p.splitk > 1 || p.conv_F != 0 || p.groups > 1) return hipErrorInvalidValue;
    const int MT = (p.M + 127) / 128, NT = p.N / 256;
    p.group_m = MT >= 16 ? 8 : MT;
    const int forced_gm = tune_get(p.tune, &uvl_tuning::gemm_gm, -1);
    if (forced_gm > 0) p.group_m = forced_gm;
    gemm_derive(p, 128, 256);
    const int nblk = 8 * ((MT * NT + 7) / 8);
    constexpr size_t lds = 3 * (size_t)(128 + 256) * 128 + 1024;      // three K-tile buffers + the tile's bias row
    auto kern = gemm_pipe128_kernel<EPI, MI16, PRE>;
    static bool attr_done = false;
    if (!attr_done) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) return e;
        attr_done = true;
    }
    static char name[48];
    if (!name[0]) snprintf(name, sizeof(name), "gemm_pipe128_kernel<%d,%d,%d>", EPI, (int)MI16, (int)PRE);     // bools as 0 / 1, the way tools/make_profiles.py writes rocprofv3's names
    g_last_kernel = name;
    hipLaunchKernelGGL(kern, dim3(nblk), dim3(512), lds, s, p);
    return hipGetLastError();
}

template <int BM, int EPI, int VAR, bool MI16>
__global__ __launch_bounds__(512) void gemm_pipe_kernel(const GemmParams p) {
    kernarg_warm<sizeof(GemmParams)>();
    extern __shared__ __attribute__((aligned(16))) char smem[];
    gemm_pipe_body<BM, EPI, VAR, MI16>(p, blockIdx.x, smem);
}

template <int BM, int EPI, int VAR = 1, bool MI16 = true>
static hipError_t launch_pipe(const GemmParams& p_in, hipStream_t s) {
    GemmParams p = p_in;
    if (p.N % 256 != 0 || p.K < 128 || p.splitk > 1 || p.conv_F != 0 || p.groups > 1) return hipErrorInvalidValue;
    const int MT = (p.M + BM - 1) / BM, NT = p.N / 256;
    p.group_m = MT >= 16 ? 8 : MT;
    const int forced_gm = tune_get(p.tune, &uvl_tuning::gemm_gm, -1);
    if (forced_gm > 0) p.group_m = forced_gm;
    gemm_derive(p, BM, 256);
    const int nblk = 8 * ((MT * NT + 7) / 8);
#ifdef GEMM_TRACE
    constexpr size_t lds = 2 * (size_t)(BM + 256) * 128 + 1024 + 2 * 64 * 4 * 4 * 4;
#else
    constexpr size_t lds = 2 * (size_t)(BM + 256) * 128 + 1024;      // two K-tile buffers + the tile's bias row
#endif
    auto kern = gemm_pipe_kernel<BM, EPI, VAR, MI16>;
    static bool attr_done = false;
    if (!attr_done) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) return e;
        attr_done = true;
    }
    static char name[48];
    if (!name[0]) snprintf(name, sizeof(name), "gemm_pipe_kernel<%d,%d,%d,%d>", BM, EPI, VAR, (int)MI16);
    g_last_kernel = name;
    hipLaunchKernelGGL(kern, dim3(nblk), dim3(512), lds, s, p);
    return hipGetLastError();
}


// Two independent plain GEMMs of the same instantiation in one launch (batch-1 frames: a text-branch GEMM rides with the
// visual GEMM of the same kind).  1-D grid: problem A owns [0, blocks_a) = tiles_a x splitk_a, problem B the rest; tile
// counts are multiples of 8, so the workgroup -> XCD relation of both tile maps is preserved.
// Problem B -- the rider -- loads its WEIGHT tiles non-temporal: the frame's weights (ViT 170 MB + BERT 85 MB + head 17 MB) do not fit
// the 256 MB Infinity Cache together, and cycling through 273 MB evicts everything before its reuse in the next frame; the rider's
// weights are read once, by one CU each, so taking them out of the allocation stream leaves the rest resident across frames
// (measured on one box: 1207 -> 1282-1290 frames/s; non-temporal on ALL weights: no gain, and 1402 -> 1273 when the text branch
// is reused, because the nine M tiles of a visual GEMM share each weight tile through L2).
// (Round 5, measured and not kept for the rider of THIS kernel: its whole weight panel requested up front before the first LDS-DMA -- non-temporal, one dword per
// line: -5 % on the one-sequence frame; plain, both halves of every line: -5..7 % (the BERT weights then displace the ViT weights from the memory-side cache) -- and
// the rider on 128 x 32 tiles -- half the panel per CU, twice the workgroups: -2.5..3 %.  profiles/r05_summary.md)
template <int BM, int BN, int WGM, int WGN, int EPI, int NS>
__global__ __launch_bounds__(64 * WGM * WGN) void gemm_glds_pair_kernel(const GemmParams pa, const GemmParams pb, int blocks_a, int tiles_a, int tiles_b,
                                                                         const FastDiv fa, const FastDiv fb) {      // fa / fb = fastdiv_of(tiles_a / tiles_b)
    kernarg_warm<2 * sizeof(GemmParams) + 16 + 16 + 64>();
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const uint32_t pfs = prefetch_issue<64 * WGM * WGN>(pa.pf, pa.pf_bytes, blockIdx.x, gridDim.x);     // (the visual problem's next weight; the rider's gain nothing from it)
    // two calls, not a selected reference: selecting between the two by-value argument blocks would copy one into scratch
    // rider_first (round 5): the rider's workgroups take the FIRST block indices.  Its tiles stream BERT weights nobody has read this frame (HBM latency per
    // K tile) while the visual tiles' weights sit in the memory-side cache: dispatched behind the visual tiles they were the tail of every pair launch
    // (rocprofv3, one UVLTrack-B sequence: pair launches 9.3 / 10.8 / 10.5 us against 8.1 / 9.2 / 8.9 us for the same visual problem alone).
    const int blocks_b = (int)gridDim.x - blocks_a;
    const int bid = pb.rider_first ? ((int)blockIdx.x >= blocks_b ? (int)blockIdx.x - blocks_b : (int)blockIdx.x + blocks_a) : (int)blockIdx.x;      // -> [A | B] index
    if (bid < blocks_a) {
        const int id = bid, sk = (int)fd_div((uint32_t)id, fa);
        gemm_glds_body<BM, BN, WGM, WGN, EPI, NS, false>(pa, id - sk * tiles_a, sk, 0, smem);
    } else {
        const int id = bid - blocks_a, sk = (int)fd_div((uint32_t)id, fb);
        gemm_glds_body<BM, BN, WGM, WGN, EPI, NS, false, true>(pb, id - sk * tiles_b, sk, 0, smem);
    }
    prefetch_retire(pfs);
}

// The same for the eight-wave tiles of many-sequence frames (residual GEMMs: f32 read-modify-write epilogue): problem A on 256 x 256
// (BMA = 256, cfg 30) or 128 x 256 tiles (cfg 31), the rider -- a few hundred text rows -- on 128 x 256 tiles behind them.  A's grid is a
// single round of at most 256 workgroups on the shapes that take these kernels, and the rider's 9-30 tiles fit beside it.
template <int BMA, int EPI, int PRE = 0>               // PRE: the visual problem's residual rows are requested inside its K loop (gemm_pipe128_body; BMA = 128 only; 2 = the K > 2048 symbol)
__global__ __launch_bounds__(512) void gemm_pipe_pair_kernel(const GemmParams pa, const GemmParams pb, const int blocks_b) {
    kernarg_warm<2 * sizeof(GemmParams) + 8>();
    extern __shared__ __attribute__((aligned(16))) char smem[];
    if ((int)blockIdx.x >= blocks_b) {              // the rider's workgroups first (see gemm_dr_pair_kernel)
        if constexpr (BMA == 256) gemm_pipe_body<256, EPI, 1, true>(pa, (int)blockIdx.x - blocks_b, smem);
        else gemm_pipe128_body<EPI, true, PRE != 0>(pa, (int)blockIdx.x - blocks_b, smem);
    } else {
        gemm_pipe128_body<EPI, true>(pb, blockIdx.x, smem);
    }
}

static bool pipe_ok(const GemmParams& p) { return p.M > 0 && p.N % 256 == 0 && p.K >= 128 && p.K % 64 == 0 && p.splitk <= 1 && p.conv_F == 0 && p.groups <= 1; }
// the rider of a pair launch may be cut into K slices (f32 slabs): gemm_pipe128_body maps slice and tile from the workgroup index
static bool pipe_ok_rider(const GemmParams& p) {
    if (p.splitk <= 1) return pipe_ok(p);
    return p.M > 0 && p.N % 256 == 0 && p.K % (64 * p.splitk) == 0 && p.K / p.splitk >= 128 && p.conv_F == 0 && p.groups <= 1 && p.epi == EPI_F32 && !p.accumulate;
}

template <int BMA, int EPI, int PRE = 0>
static hipError_t launch_pipe_pair(const GemmParams& a_in, const GemmParams& b_in, hipStream_t s) {
    if constexpr (BMA == 128 && EPI == EPI_F32 && !PRE) {
        const int pre = pipe128_pre_ok(a_in);
        if (pre == 1) return launch_pipe_pair<BMA, EPI, 1>(a_in, b_in, s);
        if (pre == 2) return launch_pipe_pair<BMA, EPI, 2>(a_in, b_in, s);
    }
    GemmParams a = a_in, b = b_in;
    if (!pipe_ok(a) || !pipe_ok_rider(b)) return hipErrorInvalidValue;
    auto grid = [](GemmParams& p, int BM) {
        const int MT = (p.M + BM - 1) / BM, NT = p.N / 256;
        p.group_m = MT >= 16 ? 8 : MT;
        const int forced_gm = tune_get(p.tune, &uvl_tuning::gemm_gm, -1);
        if (forced_gm > 0) p.group_m = forced_gm;
        gemm_derive(p, BM, 256);
        return 8 * ((MT * NT * (p.splitk > 1 ? p.splitk : 1) + 7) / 8);
    };
    const int ba = grid(a, BMA), bb = grid(b, 128);
    constexpr size_t lds = 3 * (size_t)(128 + 256) * 128 + 1024;      // the larger of the two bodies' needs (cfg 31's three buffers)
    static_assert(lds >= 2 * (size_t)(256 + 256) * 128 + 1024, "cfg 30's two buffers fit");
    auto kern = gemm_pipe_pair_kernel<BMA, EPI, PRE>;
    static bool attr_done = false;
    if (!attr_done) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) return e;
        attr_done = true;
    }
    static char name[48];
    if (!name[0]) snprintf(name, sizeof(name), "gemm_pipe_pair_kernel<%d,%d,%d>", BMA, EPI, (int)PRE);
    g_last_kernel = name;
    hipLaunchKernelGGL(kern, dim3(ba + bb), dim3(512), lds, s, a, b, bb);
    return hipGetLastError();
}

// K-slice map (see gemm_glds_body): a split-K GEMM with few M tiles whose slices and N panels divide over the 8 XCDs
static bool gemm_kxcd_ok(const GemmParams& p, int MT, int NT) {
    return tune_get(p.tune, &uvl_tuning::gemm_kxcd, 1) && p.splitk > 1 && 8 % p.splitk == 0 && NT % (8 / p.splitk) == 0 && MT < 16 && p.conv_F == 0 && p.groups <= 1;
}

template <int BM, int BN, int WGM, int WGN, int EPI, int NS, bool CONV = false, bool NTW = false, int BK = 64, int PROD = 0>
static hipError_t launch_glds(const GemmParams& p_in, hipStream_t s) {
    GemmParams p = p_in;
    const int MT = (p.M + BM - 1) / BM, NT = p.N / BN;
    // Tile order: contiguous runs of a (group of M tiles) x (all N tiles) order per XCD.  Many M tiles: groups of 8 (near-square
    // patch per L2).  Few M tiles (one or two sequences): one group = N-major order, so the M tiles of a weight panel still share
    // an L2 AND every XCD gets the same number of tiles -- "XCD x owns N panels x, x+8, ..." left half the XCDs with 2 panels and
    // half with 1 when N = 768 (12 panels): 1378-1405 -> 1468-1499 frames/s in the frame (text branch reused, same box).
    p.group_m = MT >= 16 ? 8 : MT;
    const int forced_gm = tune_get(p.tune, &uvl_tuning::gemm_gm, -1);
    if (forced_gm >= 0) p.group_m = forced_gm;
    const bool kxcd = !CONV && forced_gm < 0 && gemm_kxcd_ok(p, MT, NT);
    if (kxcd) p.group_m = -1;
    gemm_derive(p, BM, BN);
    const int nblk = kxcd ? 8 * MT * (NT / (8 / p.splitk)) : p.group_m ? 8 * ((MT * NT + 7) / 8) : 8 * ((NT + 7) / 8) * MT;
    const size_t lds = (size_t)NS * (BM + BN) * (BK * 2);
    auto kern = gemm_glds_kernel<BM, BN, WGM, WGN, EPI, NS, CONV, NTW, BK, PROD>;
    static bool attr_done = false;
    if (!attr_done && lds > 48 * 1024) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) return e;
        attr_done = true;
    }
    static char name[64];
    if (!name[0]) {
        if (PROD) snprintf(name, sizeof(name), "gemm_glds_kernel<%d,%d,%d,%d,%d,%d,%d,0,%d,%d>", BM, BN, WGM, WGN, EPI, NS, (int)CONV, BK, PROD);   // producer count in the name
        else snprintf(name, sizeof(name), NTW ? "gemm_glds_kernel<%d,%d,%d,%d,%d,%d,%d,nt>" : BK == 64 ? "gemm_glds_kernel<%d,%d,%d,%d,%d,%d,%d>" : "gemm_glds_kernel<%d,%d,%d,%d,%d,%d,%d,0,32>", BM, BN, WGM, WGN, EPI, NS, (int)CONV);
    }
    g_last_kernel = name;
    hipLaunchKernelGGL(kern, dim3(nblk, (p.splitk > 1 && !kxcd) ? p.splitk : 1, CONV ? (p.groups > 0 ? p.groups : 1) : 1), dim3(64 * (WGM * WGN + PROD)), lds, s, p);
    return hipGetLastError();
}

// Ring depth of the one-sequence tile (64 x 64, 16 KB per stage; two workgroups per CU up to 5 stages = 80 KB).  Measured in the
// one-sequence frame, same box, three interleaved repeats (tools/ab_tune.py ring1 ...): 3 stages 1347-1353 frames/s, 4 stages
// 1354-1375 (+1 %), 5 stages 1314-1322 (-2.5 %); 2 stages lost 5 % in round 1.  uvl_tuning.ring1 overrides it.
static int ring1_depth(const GemmParams& p) {
    const int r = tune_get(p.tune, &uvl_tuning::ring1, 4);
    return (r >= 3 && r <= 5) ? r : 4;
}

template <int EPI>
static hipError_t launch_plain_cfg(int cfg, const GemmParams& p, hipStream_t s) {
    if (cfg == 4 && ring1_depth(p) != 3) {
        const bool ns5 = ring1_depth(p) == 5;
        if (p.w_stream) return ns5 ? launch_glds<64, 64, 2, 2, EPI, 5, false, true>(p, s) : launch_glds<64, 64, 2, 2, EPI, 4, false, true>(p, s);
        return ns5 ? launch_glds<64, 64, 2, 2, EPI, 5>(p, s) : launch_glds<64, 64, 2, 2, EPI, 4>(p, s);
    }
    if (p.w_stream) {                      // the configurations the text branch resolves to (M = 40 rows per sequence)
        switch (cfg) {
            case 4: return launch_glds<64, 64, 2, 2, EPI, 3, false, true>(p, s);
            case 9: return launch_glds<128, 64, 2, 2, EPI, 2, false, true>(p, s);
            case 10: return launch_glds<64, 128, 2, 2, EPI, 2, false, true>(p, s);
            default: break;
        }
    }
    switch (cfg) {
        case 0: return launch_glds<64, 64, 2, 2, EPI, 4>(p, s);
        case 1: return launch_glds<128, 64, 2, 2, EPI, 4>(p, s);
        case 2: return launch_glds<128, 128, 2, 2, EPI, 3>(p, s);
        case 3: return launch_glds<64, 128, 2, 2, EPI, 4>(p, s);
        case 4: return launch_glds<64, 64, 2, 2, EPI, 3>(p, s);
        case 5: return launch_glds<64, 64, 2, 2, EPI, 6>(p, s);
        case 6: return launch_glds<128, 128, 2, 2, EPI, 2>(p, s);
        case 7: return launch_glds<64, 64, 2, 2, EPI, 2>(p, s);
        case 8: return launch_glds<128, 64, 2, 2, EPI, 3>(p, s);
        case 9: return launch_glds<128, 64, 2, 2, EPI, 2>(p, s);
        case 10: return launch_glds<64, 128, 2, 2, EPI, 2>(p, s);
        // 8-wave workgroups, one per CU: the L2->LDS fill rate (not MFMA) bounds the 128-wide tiles at large M
        case 11: return launch_glds<256, 256, 2, 4, EPI, 2>(p, s);
        case 12: return launch_glds<256, 128, 2, 4, EPI, 2>(p, s);
        case 13: return launch_glds<256, 128, 4, 2, EPI, 3>(p, s);
        case 14: return launch_glds<128, 256, 2, 4, EPI, 3>(p, s);
        case 15: return launch_glds<256, 128, 4, 2, EPI, 2>(p, s);
        // 32-wide K stages: 16 KB per 128x128 stage, so a deeper ring fits two or three times per CU
        case 16: return launch_glds<128, 128, 2, 2, EPI, 4, false, false, 32>(p, s);     // 64 KB: 2 workgroups per CU, 3 K tiles in flight each
        case 17: return launch_glds<128, 128, 2, 2, EPI, 5, false, false, 32>(p, s);     // 80 KB: 2 per CU, 4 in flight
        case 18: return launch_glds<128, 128, 2, 2, EPI, 3, false, false, 32>(p, s);     // 48 KB: 3 per CU, 2 in flight
        case 19: return launch_glds<128, 128, 2, 2, EPI, 6, false, false, 32>(p, s);     // 96 KB: 1 per CU, 5 in flight
        // producer waves: the consumers issue no LDS-DMA
        case 20: return launch_glds<128, 128, 2, 2, EPI, 2, false, false, 64, 2>(p, s);  // 4 consumers + 2 producers, 2 workgroups per CU (3 waves / SIMD)
        case 21: return launch_glds<128, 128, 2, 2, EPI, 2, false, false, 64, 4>(p, s);  // 4 + 4, 2 per CU (4 waves / SIMD: 128 registers)
        // phase-pipelined 256-wide tiles (gemm_pipe_body)
        case 30: return launch_pipe<256, EPI, 1, true>(p, s);    // 256 x 256, product form: LDS-DMA issued between the MFMAs, 16x16x32 MFMAs
        case 31: return launch_pipe128<EPI, true>(p, s);         // 128 x 256: two phases per K tile, three buffers, 16x16x32 MFMAs
        case 32: return launch_pipe<256, EPI, 1, false>(p, s);   // cfg 30 with 32x32x16 MFMAs (the form of the first half of round 3; A/B)
        case 33: return launch_pipe128<EPI, false>(p, s);        // cfg 31 with 32x32x16 MFMAs
        case 36: return launch_gemm_dr(p, EPI, s);                // 128 x 256 on four waves, two workgroups per CU, W fragments straight from global memory
    }
    return hipErrorInvalidValue;
}

static int pick_plain_cfg(const GemmParams& p) {
    const int forced = tune_get(p.tune, &uvl_tuning::gemm_cfg, -1);      // tools / tests: index into the table above
    if (forced >= 0) return forced;
    // measured on MI355X (tools/gemm_bench.py, tools/lib_compare.py, profiles/): co-resident workgroups matter more than
    // ring depth, so the 2-stage ring wins once there are enough tiles; few tiles keep 64x64 for parallelism
    const int mt64 = (p.M + 63) / 64;
    const long t64 = (long)mt64 * (p.N / 64);
    const bool n128 = p.N % 128 == 0;
    if (mt64 < 16) {                                // one or two sequences: weight panels stream from HBM
        if (t64 < 1024) return 4;                   // 64x64, 3 stages: one more tile in flight pays
        if (t64 < 2048) return 7;                   // 64x64, 2 stages
        return n128 ? 10 : 9;
    }
    if (t64 < 768) return 4;
    if (p.M >= 2048 && tune_get(p.tune, &uvl_tuning::gemm_pipe, 1) && p.N % 256 == 0 && p.K >= 128 && p.splitk <= 1 && !(p.epi == EPI_F32 && p.K < 512)) {
        // phase-pipelined 256-wide tiles (gemm_pipe_body / gemm_pipe128_body), one workgroup per CU: 256 x 256 (7.8 bytes of LDS fill
        // per KFLOP, 0.75 fragment reads per MFMA) where its tiles fill >= 80 % of whole rounds of the 256 CUs, else 128 x 256 under
        // the same rule, else the better-filling of the two from 60 % on.  One workgroup per CU exposes the epilogue; the
        // read-modify-write f32 epilogue took them only behind K >= 2048 until the loops moved to 16x16x32 MFMAs -- since then from
        // K = 512 (same box, new / old rule: 32 UVLTrack-B sequences 5601-5610 / 5517-5519 frames/s, 16: 4854-4858 / 4817-4820,
        // UVLTrack-L x 8 1124-1125 / 1118-1119, x 32 1352-1355 / 1340-1343).  Measured against the round-2 kernels in
        // isolation (tools/gemm_pipe_ab.py, tools/lib_compare.py, profiles/r03_gemm_pipe.md): +5..+35 % from 8 UVLTrack-B sequences
        // (M = 4424: fc1 605 -> 811 TFLOP/s) to 32 UVLTrack-L sequences.
        auto fill = [](long t) { const long rounds = (t + 255) / 256; return (double)t / (double)(rounds * 256); };
        const long nt256 = p.N / 256;
        const double f256 = fill((long)((p.M + 255) / 256) * nt256), f128 = fill((long)((p.M + 127) / 128) * nt256);
        // The direct-to-register form (cfg 36, gemm_dr.hip: 128 x 256 tiles on four waves, TWO workgroups per CU, W fragments loaded straight
        // into registers from the fragment-native weight image): what a tile pays outside its K loop runs under the other workgroup's loop.
        // Measured beside cfg 30 / 31 and hipBLASLt (profiles/r04_gemm_dr.md): ahead of both with the bias / GELU / QKV epilogues on every
        // shape of 8 sequences.  uvl_tuning.gemm_dr: 0 = never, 1 = wherever it applies (the f32 epilogue too).
        {
            const int want = tune_get(p.tune, &uvl_tuning::gemm_dr, -1);
            // The f32 read-modify-write epilogue stays with the eight-wave kernels: in isolation cfg 36 wins it from ~400 tiles on (proj of 16 /
            // 32 UVLTrack-L sequences 41.6 / 64.2 against 44.9 / 71.0 us; 8 sequences = 220 tiles = a single round: 28.3 against 25.0), but in
            // the frames it loses (interleaved tools/ab_tune.py gemm_dr 2 -1: UVLTrack-L x 8 1156-1157 against 1136-1138 frames/s, UVLTrack-B x 32
            // 6150-6172 against 6111-6131): 2 x 28 MB of residual traffic per launch, and two workgroups per CU issue it at the same moment.
            if (want != 0 && p.Wp && p.K % 64 == 0 && (want == 1 || p.epi != EPI_F32)) return 36;
        }
        const int c256 = 30;
        if (f256 >= 0.8) return c256;
        if (f128 >= 0.8) return 31;
        if (f256 >= 0.6 || f128 >= 0.6) return f256 >= f128 ? c256 : 31;
    }
    if (p.M < 6144) return n128 ? 10 : 9;           // 64x128 (128x64), 2 stages
    if (tune_get(p.tune, &uvl_tuning::gemm_prod, 1) && p.epi != EPI_F32 && p.N >= 3072 && n128 && p.splitk <= 1) return 21;   // 4 consumers + 4 producers (probe 5)
    if (tune_get(p.tune, &uvl_tuning::gemm_big, 1) && p.N % 256 == 0 && p.K >= 1024 && p.splitk <= 1) {
        // 256x256 tiles, 8 waves, one workgroup per CU: half the LDS-DMA instructions per MFMA of the 128x128 tile (an LDS-DMA
        // instruction costs its wave ~55 cycles of issue).  Pays where the K loop is long enough to amortise the tile's prologue /
        // epilogue and the tiles fill whole rounds of the 256 CUs: +20 % on fc1 at M = 6984, K = 1024 (448 tiles), +18 % on fc2 at
        // M = 17696, K = 3072 (210 tiles); level or worse elsewhere (tools/gemm_ring_ab.py)
        const long t256 = (long)((p.M + 255) / 256) * (p.N / 256);
        const long rounds = (t256 + 255) / 256;
        if (t256 * 10 >= rounds * 256 * 8) return 11;
    }
    return n128 ? 6 : 9;                            // 128x128, 2 stages
}

template <int EPI>
static hipError_t launch_epi(const GemmParams& p, hipStream_t s) {
    if (p.N % 64 != 0) {                        // N % 32 == 0 (launch_gemm checked): one 32-column tile shape
        return launch_glds<128, 32, 4, 1, EPI, 3>(p, s);
    }
    int cfg = pick_plain_cfg(p);
    if ((cfg == 2 || cfg == 3 || cfg == 6 || cfg == 10 || cfg == 12 || cfg == 13 || cfg == 15 || (cfg >= 16 && cfg <= 21)) && p.N % 128 != 0) cfg = 0;
    if (cfg >= 16 && cfg <= 21 && p.splitk > 1) cfg = 6;
    if ((cfg == 11 || cfg == 14 || (cfg >= 30 && cfg <= 33) || cfg == 36) && (p.N % 256 != 0 || p.K < 128 || p.splitk > 1)) cfg = 0;
    return launch_plain_cfg<EPI>(cfg, p, s);
}

// Both problems in one launch when they resolve to the same 64x64 / 3-stage instantiation (the batch-1 configuration);
// otherwise two launches on the same stream -- same results either way.
template <int EPI, int NS>
static hipError_t launch_pair_epi(const GemmParams& a_in, const GemmParams& b_in, hipStream_t s) {
    GemmParams a = a_in, b = b_in;
    const int mta = (a.M + 63) / 64, mtb = (b.M + 63) / 64;
    a.group_m = mta;                       // N-major runs per XCD, as launch_glds does for few M tiles
    b.group_m = mtb;
    int ta = 8 * ((mta * (a.N / 64) + 7) / 8), tb = 8 * ((mtb * (b.N / 64) + 7) / 8);
    int ba = ta * a.splitk, bb = tb * b.splitk;
    // K-slice map: the grid of a problem is 1-D over (tile, slice); "tiles" = all its blocks makes the kernel pass the block id through
    if (gemm_kxcd_ok(a, mta, a.N / 64)) { a.group_m = -1; ba = ta = 8 * mta * ((a.N / 64) / (8 / a.splitk)); }
    if (gemm_kxcd_ok(b, mtb, b.N / 64)) { b.group_m = -1; bb = tb = 8 * mtb * ((b.N / 64) / (8 / b.splitk)); }
    gemm_derive(a, 64, 64);
    gemm_derive(b, 64, 64);
    constexpr size_t lds = NS * (size_t)(64 + 64) * 128;
    auto kern = gemm_glds_pair_kernel<64, 64, 2, 2, EPI, NS>;
    static bool attr_done = false;
    if (!attr_done && lds > 48 * 1024) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) return e;
        attr_done = true;
    }
    static char name[64];
    if (!name[0]) snprintf(name, sizeof(name), "gemm_glds_pair_kernel<64,64,2,2,%d,%d>", EPI, NS);
    g_last_kernel = name;
    hipLaunchKernelGGL(kern, dim3(ba + bb), dim3(256), lds, s, a, b, ba, ta, tb, fastdiv_of((uint32_t)ta), fastdiv_of((uint32_t)tb));
    return hipGetLastError();
}

hipError_t launch_gemm_pair(const GemmParams& a, const GemmParams& b, hipStream_t s) {
    auto plain = [](const GemmParams& p) {
        return p.conv_F == 0 && (p.groups <= 1) && p.N % 64 == 0 && p.K % 64 == 0 && p.M > 0 && p.splitk >= 1 &&
               (p.splitk == 1 || (p.epi == EPI_F32 && !p.accumulate && (p.K / 64) % p.splitk == 0));
    };
    const bool pairable = plain(a) && plain(b) && a.epi == b.epi && pick_plain_cfg(a) == 4 && pick_plain_cfg(b) == 4 &&
                          tune_get(a.tune, &uvl_tuning::gemm_gm, -1) < 0 && (a.M + 63) / 64 < 16 && (b.M + 63) / 64 < 16;
    if (!pairable) {
        // many-sequence frames: the visual problem on one of the large-tile kernels, the rider on the same kernel's 128 x 256 tiles
        if (plain(a) && plain(b) && a.epi == b.epi && a.splitk == 1 && a.N % 256 == 0 && a.K >= 128 && tune_get(a.tune, &uvl_tuning::gemm_gm, -1) < 0) {
            const int cfg = pick_plain_cfg(a);
            if (cfg == 36 && b.splitk == 1 && gemm_dr_pairable(a, b)) return launch_gemm_dr_pair(a, b, s);
            if (a.epi == EPI_F32 && pipe_ok(a) && pipe_ok_rider(b)) {
                if (cfg == 30) return launch_pipe_pair<256, EPI_F32>(a, b, s);
                if (cfg == 31) return launch_pipe_pair<128, EPI_F32>(a, b, s);
            }
        }
        const hipError_t e = launch_gemm(a, s);
        return e != hipSuccess ? e : launch_gemm(b, s);
    }
    const int ns = ring1_depth(a);
    switch (a.epi * 8 + ns) {
        case EPI_BF16 * 8 + 3: return launch_pair_epi<EPI_BF16, 3>(a, b, s);
        case EPI_F32 * 8 + 3: return launch_pair_epi<EPI_F32, 3>(a, b, s);
        case EPI_QKV * 8 + 3: return launch_pair_epi<EPI_QKV, 3>(a, b, s);
        case EPI_BF16 * 8 + 4: return launch_pair_epi<EPI_BF16, 4>(a, b, s);
        case EPI_F32 * 8 + 4: return launch_pair_epi<EPI_F32, 4>(a, b, s);
        case EPI_QKV * 8 + 4: return launch_pair_epi<EPI_QKV, 4>(a, b, s);
        case EPI_BF16 * 8 + 5: return launch_pair_epi<EPI_BF16, 5>(a, b, s);
        case EPI_F32 * 8 + 5: return launch_pair_epi<EPI_F32, 5>(a, b, s);
        case EPI_QKV * 8 + 5: return launch_pair_epi<EPI_QKV, 5>(a, b, s);
    }
    return hipErrorInvalidValue;
}

// ------------------------------------------------------------------------------------------------
// LayerNorm-folded consumer GEMMs of the LayerNorm-free one- / two-sequence frame (round 6): QKV and fc1 (and the text branch's QKV / intermediate as the
// rider) on the 64 x 64 tiles, A = the un-normalised bf16 rows a finishing GEMM (gemm_fin.hip), the patch embedding or the prologue left, with their partial
// statistics.  The launch may also carry the contrastive logits of the previous layer as its FIRST workgroups (one wave per search row, fold.h::ct_job_block): the
// job used to ride on LayerNorm launches that no longer exist, its rows are complete when this launch starts and nothing of this launch depends on it.
// Block order: [rider tiles | visual tiles | logits job]; the rider's count is a multiple of 8, so every tile keeps its workgroup -> XCD relation.  The job's 64
// workgroups come LAST: dispatched first they held one of the two workgroup slots of 64 CUs while the 324 GEMM tiles were being placed, so more CUs than necessary
// ended up with two tiles (measured: QKV launch 8.9 us with the job first).
// ------------------------------------------------------------------------------------------------
// BN / NS: the visual problem's tile width and ring depth -- 64 x 64 with four stages (two workgroups per CU), or 64 x 128 (a wave = 32 x 64) for grids of more than two 64 x 64 tiles per CU
// (one UVLTrack-L sequence: 672 / 896 tiles): half the workgroups, every A stage feeds twice the MFMAs.  The rider keeps 64 x 64.
template <int EPI, int BN, int NS>
__global__ __launch_bounds__(256) void gemm_lnf_kernel(const GemmParams p, const CtJob ct, const int blocks_ct) {
    kernarg_warm<sizeof(GemmParams) + sizeof(CtJob) + 8 + 64>();
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int nb = (int)gridDim.x - blocks_ct;           // GEMM tiles first, the job's short workgroups behind them (see the note above)
    if ((int)blockIdx.x >= nb) { ct_job_block(ct, (int)blockIdx.x - nb, g_zero_row); return; }
    const uint32_t pfs = prefetch_issue<256>(p.pf, p.pf_bytes, blockIdx.x, gridDim.x);
    gemm_glds_body<64, BN, 2, 2, EPI, NS, false, false, 64, 0, 0, true>(p, (int)blockIdx.x, 0, 0, smem);
    prefetch_retire(pfs);
}
template <int EPI, int BN, int NS>
__global__ __launch_bounds__(256) void gemm_lnf_pair_kernel(const GemmParams pa, const GemmParams pb, const CtJob ct, const int blocks_ct, const int blocks_b) {
    kernarg_warm<2 * sizeof(GemmParams) + sizeof(CtJob) + 8 + 64>();
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int bid = (int)blockIdx.x, nb = (int)gridDim.x - blocks_ct;
    if (bid >= nb) { ct_job_block(ct, bid - nb, g_zero_row); return; }
    uint32_t pfs = prefetch_issue<256>(pa.pf, pa.pf_bytes, blockIdx.x, gridDim.x);
    pfs = prefetch_issue_xcd<256>(pfs, pb.pf2, pb.pf2_tiles, pb.pf2_lp, blockIdx.x, nb);
    if (bid < blocks_b) gemm_glds_body<64, 64, 2, 2, EPI, 4, false, true, 64, 0, 0, true>(pb, bid, 0, 0, smem);       // the rider first: its weight tiles come from HBM (non-temporal)
    else gemm_glds_body<64, BN, 2, 2, EPI, NS, false, false, 64, 0, 0, true>(pa, bid - blocks_b, 0, 0, smem);
    prefetch_retire(pfs);
}

static bool lnf_ok(const GemmParams& p) {
    return p.M > 0 && p.conv_F == 0 && p.groups <= 1 && p.N % 64 == 0 && p.K % 128 == 0 && p.K <= 1024 && p.splitk <= 1 && p.st_in && p.colsum &&
           (p.epi == EPI_BF16 || p.epi == EPI_QKV) && (p.M + 63) / 64 < 64;
}
// 0: 64 x 64 tiles, four stages; 1: 64 x 128, two stages (48 KB: three workgroups per CU); 2: 64 x 128, three stages (72 KB: two per CU).  uvl_tuning.lnf_w overrides.
static int lnf_form(const GemmParams& p) {
    const int forced = tune_get(p.tune, &uvl_tuning::lnf_w, -1);
    if (p.N % 128 != 0) return 0;
    if (forced >= 0 && forced <= 2) return forced;
    const long tiles = (long)((p.M + 63) / 64) * (p.N / 64);
    if (forced >= 100) return tiles > forced ? 1 : 0;          // (tools: the tile-count threshold itself)
    return tiles > 512 ? 1 : 0;
}
template <int EPI, int BN, int NS>
static hipError_t launch_lnf_epi(const GemmParams& a_in, const GemmParams* b_in, const CtJob* ct, hipStream_t s) {
    auto prep = [](GemmParams& p, int bn) {
        const int MT = (p.M + 63) / 64;
        p.group_m = MT;                       // N-major runs per XCD (few M tiles: the M tiles of a weight panel share an L2)
        gemm_derive(p, 64, bn);
        return 8 * ((MT * (p.N / bn) + 7) / 8);
    };
    GemmParams a = a_in;
    const int ba = prep(a, BN);
    CtJob job = ct ? *ct : CtJob();
    const int bc = ct ? 8 * ((((job.B * job.nx + 3) / 4) + 7) / 8) : 0;
    constexpr size_t lds_a = (size_t)NS * (64 + BN) * 128, lds_r = 4 * (size_t)(64 + 64) * 128;
    if (!b_in) {
        auto kern = gemm_lnf_kernel<EPI, BN, NS>;
        static bool attr_done = false;
        if (!attr_done) {
            hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_a);
            if (e != hipSuccess) return e;
            attr_done = true;
        }
        static char name[48];
        if (!name[0]) snprintf(name, sizeof(name), "gemm_lnf_kernel<%d,%d,%d>", EPI, BN, NS);
        g_last_kernel = name;
        hipLaunchKernelGGL(kern, dim3(ba + bc), dim3(256), lds_a, s, a, job, bc);
        return hipGetLastError();
    }
    GemmParams b = *b_in;
    const int bb = prep(b, 64);
    constexpr size_t lds = lds_a > lds_r ? lds_a : lds_r;
    auto kern = gemm_lnf_pair_kernel<EPI, BN, NS>;
    static bool attr_done2 = false;
    if (!attr_done2) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) return e;
        attr_done2 = true;
    }
    static char name2[48];
    if (!name2[0]) snprintf(name2, sizeof(name2), "gemm_lnf_pair_kernel<%d,%d,%d>", EPI, BN, NS);
    g_last_kernel = name2;
    hipLaunchKernelGGL(kern, dim3(bb + ba + bc), dim3(256), lds, s, a, b, job, bc, bb);
    return hipGetLastError();
}
hipError_t launch_gemm_lnf(const GemmParams& a, const GemmParams* b, const CtJob* ct, hipStream_t s) {
    if (!lnf_ok(a) || (b && (!lnf_ok(*b) || b->epi != a.epi))) return hipErrorInvalidValue;
    if (ct && (!ct->x || !ct->logits || !ct->flag || !ct->logit_scale || ct->D % 4 != 0 || ct->D > 1024 || ct->nx <= 0)) return hipErrorInvalidValue;
    const int form = lnf_form(a);
    if (a.epi == EPI_QKV) {
        if (form == 1) return launch_lnf_epi<EPI_QKV, 128, 2>(a, b, ct, s);
        if (form == 2) return launch_lnf_epi<EPI_QKV, 128, 3>(a, b, ct, s);
        return launch_lnf_epi<EPI_QKV, 64, 4>(a, b, ct, s);
    }
    if (form == 1) return launch_lnf_epi<EPI_BF16, 128, 2>(a, b, ct, s);
    if (form == 2) return launch_lnf_epi<EPI_BF16, 128, 3>(a, b, ct, s);
    return launch_lnf_epi<EPI_BF16, 64, 4>(a, b, ct, s);
}

#ifdef GLDS_TRACE
extern "C" int uvl_debug_glds_acc(unsigned long long* dst, int reset) {
    hipError_t e = hipMemcpyFromSymbol(dst, HIP_SYMBOL(uvl::g_glds_acc), sizeof(unsigned long long) * 64);
    if (e == hipSuccess && reset) { static unsigned long long z[64]; e = hipMemcpyToSymbol(HIP_SYMBOL(uvl::g_glds_acc), z, sizeof(z)); }
    return (int)e;
}
extern "C" int uvl_debug_glds_trace(unsigned long long* dst, int n) { return (int)hipMemcpyFromSymbol(dst, HIP_SYMBOL(uvl::g_glds_trace), (size_t)n * 8 * sizeof(unsigned long long)); }
#endif
#ifdef GEMM_TRACE
extern "C" int uvl_debug_gemm_trace(unsigned int* dst) { return (int)hipMemcpyFromSymbol(dst, HIP_SYMBOL(g_gemm_trace), (2 * 64 * 4 * 4 + 16) * sizeof(unsigned int)); }
#endif

hipError_t launch_gemm(const GemmParams& p, hipStream_t s) {
    if (p.K % 64 != 0 || p.M <= 0 || p.N <= 0 || p.N % 32 != 0 || p.splitk < 1) return hipErrorInvalidValue;
    if (p.splitk > 1 && (p.epi != EPI_F32 || p.accumulate || p.N % 64 != 0 || (p.K / 64) % p.splitk != 0))
        return hipErrorInvalidValue;      // partial slabs: f32 store epilogue only
    if (p.conv_F > 0) {                   // implicit GEMM over NHWC tokens, towers as groups (optionally split-K into f32 slabs)
        if (p.cin_g % 64 != 0) return hipErrorInvalidValue;
        if (p.N % 64 == 0) {
            const long ngr = p.groups > 0 ? p.groups : 1;
            // (round 5: 160 x 128 tiles on four waves for the first tower layer of 8 UVLTrack-L sequences -- 232 tiles instead of 288, one per CU -- measured
            // 166 us against 145 by event pairs, frame +0.1 %: four waves alone on a CU lose what the tile count gains; not kept, profiles/r05_summary.md)
            if (p.epi == EPI_BF16 && p.N % 128 == 0 && (long)(p.M / 128) * (p.N / 128) * ngr >= 256)
                return launch_glds<128, 128, 2, 2, EPI_BF16, 2, true>(p, s);   // batched frames: at least one 128x128 tile per CU
            if (p.epi == EPI_BF16) return launch_glds<64, 64, 2, 2, EPI_BF16, 3, true>(p, s);
            if (p.epi == EPI_F32) return launch_glds<64, 64, 2, 2, EPI_F32, 3, true>(p, s);
            return hipErrorInvalidValue;
        }
        if (p.epi != EPI_BF16) return hipErrorInvalidValue;
        return launch_glds<128, 32, 4, 1, EPI_BF16, 3, true>(p, s);            // last tower layer: 32 channels
    }
    if (p.groups > 1) return hipErrorInvalidValue;
    switch (p.epi) {
        case EPI_BF16: return launch_epi<EPI_BF16>(p, s);
        case EPI_F32: return launch_epi<EPI_F32>(p, s);
        case EPI_QKV: return launch_epi<EPI_QKV>(p, s);
    }
    return hipErrorInvalidValue;
}


// ------------------------------------------------------------------------------------------------
// LayerNorm + consumer GEMM of a one-sequence frame in ONE launch (ln_gemm_pair_kernel).  The frame is ~96 dependent launches of
// 5-10 us; a LayerNorm launch is 5.6 us of which ~1 us is work.  Here the workgroups of the GEMM grid first normalise the rows (ln_body,
// the row kernels' own code; bf16 output stored write-through), meet at a HIERARCHICAL grid barrier, and then run their GEMM tiles:
//   * workgroup b arrives at the counter of group b % 8 (the XCD it runs on); the arrival that completes a group arrives at the global
//     counter, waits for all eight groups, invalidates ITS XCD's L2 (one buffer_inv sc1 per XCD: the normalised rows were written by
//     other XCDs) and releases its group; the others spin on the group's release word.  Measured (tools/probes/grid_barrier2_probe.hip):
//     3.1-3.3 us per store + barrier + read with 360-432 workgroups, against 8.5 us for one flat counter and 20+ us with an
//     invalidate per workgroup;
//   * counters are monotonic: the host passes the generation of this launch and the arrivals of all earlier launches (uvl_model keeps
//     both; grids differ from launch to launch), nothing is ever reset -- which is
//     why graph capture keeps the two-launch form (a replay would repeat the generation);
//   * every workgroup must be resident at once: the launcher fuses only when the grid fits what the occupancy query says this device
//     holds of the kernel (512 on a whole MI355X), and only on an 8-XCD, 256-CU device (the barrier's groups are the dispatch order).
// Both problems of a paired launch (visual rows + the text branch's rider) go through the same barrier.  Same arithmetic, bit for
// bit, as layernorm(_pair) followed by gemm(_pair): the device functions are the same.
// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ unsigned gb_load(const unsigned* p) {
    unsigned v;
    asm volatile("global_load_dword %0, %1, off sc0 sc1\n\ts_waitcnt vmcnt(0)" : "=v"(v) : "v"(p) : "memory");
    return v;
}
// `base`: arrivals the group counters have seen in ALL earlier launches (grids differ from launch to launch, so the host keeps the sum:
// one counter value, the same for the eight groups, because every grid is a multiple of 8 workgroups)
__device__ __forceinline__ void grid_barrier(unsigned* ctr, const unsigned gen, const unsigned base, const int nwg) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");            // this wave's write-through stores have reached memory
    __syncthreads();
    if (threadIdx.x == 0) {
        const int grp = (int)blockIdx.x & 7;
        const unsigned gsize = (unsigned)nwg >> 3;               // the grid is a multiple of 8 workgroups
        unsigned* gcnt = ctr + grp * 16;                         // 64-byte spacing
        unsigned* grel = ctr + 128 + grp * 16;
        unsigned* glob = ctr + 256;
        const unsigned prev = __hip_atomic_fetch_add(gcnt, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (prev + 1 == base + gsize) {
            __hip_atomic_fetch_add(glob, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            while ((int)(gb_load(glob) - gen * 8u) < 0) __builtin_amdgcn_s_sleep(1);     // wrap-safe: the counters run for ever
            asm volatile("buffer_inv sc1" ::: "memory");
            asm volatile("global_store_dword %0, %1, off sc0 sc1\n\ts_waitcnt vmcnt(0)" ::"v"(grel), "v"(gen) : "memory");
        } else {
            while ((int)(gb_load(grel) - gen) < 0) __builtin_amdgcn_s_sleep(1);
        }
    }
    __syncthreads();
}

template <int EPI, int NS, int NV>
__global__ __launch_bounds__(256, 2) void ln_gemm_pair_kernel(const LnParams la, const LnParams lb, const int vba, const int vbb, const GemmParams pa, const GemmParams pb,
                                                              const int blocks_a, const int tiles_a, const int tiles_b, unsigned* bar, const unsigned gen, const unsigned base) {
    kernarg_warm<2 * sizeof(LnParams) + 2 * sizeof(GemmParams) + 32>();
    extern __shared__ __attribute__((aligned(16))) char smem[];
    // phase 0: the weight halves of this workgroup's first three K tiles go out (cold, from HBM): they fly under phases 1 and 2
    if ((int)blockIdx.x < blocks_a) {
        const int id = (int)blockIdx.x, sk = id / tiles_a;
        gemm_glds_body<64, 64, 2, 2, EPI, NS, false, false, 64, 0, 1>(pa, id - sk * tiles_a, sk, 0, smem);
    } else {
        const int id = (int)blockIdx.x - blocks_a, sk = id / tiles_b;
        gemm_glds_body<64, 64, 2, 2, EPI, NS, false, true, 64, 0, 1>(pb, id - sk * tiles_b, sk, 0, smem);
    }
    // phase 1: LayerNorm rows, four (one per wave) per virtual block, spread over the whole grid
    for (int vb = (int)blockIdx.x; vb < vba + vbb; vb += (int)gridDim.x) {
        if (vb < vba) ln_body<NV, true, true, true>(la, vb);
        else ln_body<NV, true, true, false>(lb, vb - vba);
    }
    grid_barrier(bar, gen, base, (int)gridDim.x);
    // phase 2: the GEMM tiles, as gemm_glds_pair_kernel
    if ((int)blockIdx.x < blocks_a) {
        const int id = (int)blockIdx.x, sk = id / tiles_a;
        gemm_glds_body<64, 64, 2, 2, EPI, NS, false, false, 64, 0, 2>(pa, id - sk * tiles_a, sk, 0, smem);
    } else {
        const int id = (int)blockIdx.x - blocks_a, sk = id / tiles_b;
        gemm_glds_body<64, 64, 2, 2, EPI, NS, false, true, 64, 0, 2>(pb, id - sk * tiles_b, sk, 0, smem);
    }
}

template <int EPI, int NV>
static hipError_t launch_ln_gemm_epi(const LnParams& la, const LnParams* lb, const GemmParams& a_in, const GemmParams* b_in, unsigned* bar, unsigned gen, unsigned* base, hipStream_t s) {
    constexpr int NS = 4;
    GemmParams a = a_in, b = b_in ? *b_in : a_in;
    const int mta = (a.M + 63) / 64, mtb = (b.M + 63) / 64;
    a.group_m = mta;
    b.group_m = mtb;
    gemm_derive(a, 64, 64);
    gemm_derive(b, 64, 64);
    const int ta = 8 * ((mta * (a.N / 64) + 7) / 8), tb = b_in ? 8 * ((mtb * (b.N / 64) + 7) / 8) : 0;
    const int ba = ta, bb = tb;                                   // no split-K here (QKV / fc1)
    constexpr size_t lds = NS * (size_t)(64 + 64) * 128;
    auto kern = ln_gemm_pair_kernel<EPI, NS, NV>;
    static bool attr_done = false;
    if (!attr_done) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) return e;
        attr_done = true;
    }
    static char name[48];
    if (!name[0]) snprintf(name, sizeof(name), "ln_gemm_pair_kernel<%d,%d,%d>", EPI, NS, NV);
    g_last_kernel = name;
    LnParams l0 = la, l1 = lb ? *lb : la;
    l0.y_wt = 1;
    l1.y_wt = 1;
    l0.fd_rpb = fastdiv_of((uint32_t)l0.rpb);
    l1.fd_rpb = fastdiv_of((uint32_t)l1.rpb);
    const int vba = (l0.M + 3) / 4, vbb = lb ? (l1.M + 3) / 4 : 0;
    hipLaunchKernelGGL(kern, dim3(ba + bb), dim3(256), lds, s, l0, l1, vba, vbb, a, b, ba, ta, tb ? tb : 1, bar, gen, *base);
    const hipError_t e = hipGetLastError();
    if (e == hipSuccess) *base += (unsigned)(ba + bb) >> 3;      // every group counter moves by grid / 8
    return e;
}

// Fused form where it applies, else LayerNorm(_pair) then GEMM(_pair).  `gen` is consumed (the caller increments it) only when the
// function returns with *fused = true.
hipError_t launch_ln_gemm_pair(const LnParams& la, const LnParams* lb, const GemmParams& a, const GemmParams* b, unsigned* bar, unsigned gen, unsigned* base, bool* fused, hipStream_t s) {
    auto plain = [](const GemmParams& p) {
        return p.conv_F == 0 && p.groups <= 1 && p.N % 64 == 0 && p.K % 64 == 0 && p.M > 0 && p.splitk == 1 && !p.accumulate && (p.epi == EPI_BF16 || p.epi == EPI_QKV);
    };
    const int blocks = 8 * ((((a.M + 63) / 64) * (a.N / 64) + 7) / 8) + (b ? 8 * ((((b->M + 63) / 64) * (b->N / 64) + 7) / 8) : 0);
    // every workgroup of the grid must be resident at once (they spin on the barrier): the limit is what THIS device holds of this kernel
    // (two workgroups of 64 KB LDS per CU x its CU count: 512 on a whole MI355X, less on a partitioned or CU-masked one), and the
    // barrier's per-XCD groups assume the 8-XCD dispatch order -- any other device keeps the two-launch form
    static int resident_limit = -1;
    if (resident_limit < 0) {
        int dev = 0, cus = 0, per_cu = 0;
        hipDeviceProp_t prop;
        if (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&prop, dev) == hipSuccess &&
            hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, reinterpret_cast<const void*>(ln_gemm_pair_kernel<EPI_BF16, 4, 4>), 256, 4 * (64 + 64) * 128) == hipSuccess) {
            cus = prop.multiProcessorCount;
            resident_limit = (cus == 256) ? per_cu * cus : 0;
        } else {
            (void)hipGetLastError();
            resident_limit = 0;
        }
    }
    const bool ok = bar && plain(a) && (!b || (plain(*b) && b->epi == a.epi)) && pick_plain_cfg(a) == 4 && (!b || pick_plain_cfg(*b) == 4) && ring1_depth(a) == 4 &&
                    tune_get(a.tune, &uvl_tuning::gemm_gm, -1) < 0 && (a.M + 63) / 64 < 16 && (!b || (b->M + 63) / 64 < 16) && blocks <= resident_limit &&
                    (la.D == 768 || la.D == 1024) && (!lb || lb->D == la.D) && la.nsplit <= LN_MAX_SLABS && !la.ct_self &&      // (ct_self: the direct contrast job needs ln_body's CT = 2 form; the fused kernel instantiates CT = 1)
                    (!lb || (lb->nsplit <= LN_MAX_SLABS && !lb->ct_x)) &&
                    la.y_bf16 == a.A && (!lb || !b || lb->y_bf16 == b->A) && (lb != nullptr) == (b != nullptr);
    *fused = ok;
    if (!ok) {
        hipError_t e = lb ? launch_layernorm_pair(la, *lb, s) : launch_layernorm(la, s);
        if (e != hipSuccess) return e;
        return b ? launch_gemm_pair(a, *b, s) : launch_gemm(a, s);
    }
    if (la.D == 768) return a.epi == EPI_QKV ? launch_ln_gemm_epi<EPI_QKV, 3>(la, lb, a, b, bar, gen, base, s) : launch_ln_gemm_epi<EPI_BF16, 3>(la, lb, a, b, bar, gen, base, s);
    return a.epi == EPI_QKV ? launch_ln_gemm_epi<EPI_QKV, 4>(la, lb, a, b, bar, gen, base, s) : launch_ln_gemm_epi<EPI_BF16, 4>(la, lb, a, b, bar, gen, base, s);
}

}  // namespace uvl
