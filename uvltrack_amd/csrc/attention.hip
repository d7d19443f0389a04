// Fused multi-head self-attention core for gfx950: softmax(q k^T / 8 + key_add) v, head_dim 64.
//
// Replaces the materialised [B,H,N,N] score tensor of Attention.forward (reference block.py:50-58:
// q@k^T * scale -> masked_fill(-1e10) -> softmax -> @v) and BertSelfAttention.forward
// (bert_backbone.py:311-324: scores/sqrt(64) + (1-mask)*-10000 -> softmax -> @v).
//
// Layout/algorithm (wave64, v_mfma_f32_32x32x16_bf16):
//   * one wave owns 32 query rows; QW query-waves x KS key-split-waves per workgroup.  KS > 1 is the batch-1
//     shape: wave (qw, ks) walks key tiles ks, ks+KS, ... and the KS partial (m, l, O) triples are merged
//     through LDS at the end (4x more resident waves, 4x shorter serial chain)
//   * K and V^T 64-key tiles go HBM/L2 -> LDS by LDS-DMA (global_load_lds_dwordx4) into an NS-stage ring with
//     counted s_waitcnt vmcnt + raw s_barrier; no staging VGPRs, no ds_write pass.  LDS rows are 128 B with the
//     16-byte chunk index XORed by (row>>1)&7 -- applied to the per-lane SOURCE address (a DMA image is lane-linear)
//     and again on the fragment reads
//   * scores are computed TRANSPOSED, S^T = K Q^T, so lane (q = lane&31) holds 16 of the 32 keys of its query
//     column per 32-key block: row max / row sum are lane-local plus ONE cross-half exchange (__shfl_xor 32)
//   * P^T feeds the second MFMA as the B operand with no data movement at all: the K rows are fed to the first MFMA in a
//     permuted order (bits 2,3 of the row swapped) so that the 8 score registers a lane-half owns per 16-key step ARE
//     8 contiguous keys; the matching V^T A-operand is then one conflict-free ds_read_b128
//   * O^T = V^T P^T accumulates [d][q]; the rescale factor and 1/l are per-lane scalars
//   * softmax runs in the log2 domain: p = exp2(s * (log2e/8) + add - m), one v_fma + one v_exp per score when the
//     tile carries no mask (a per-tile flag is set while the tile lands); the running max is only raised, and the
//     accumulators rescaled, when some row's maximum grew by more than 2^8 ("defer-max"), so the common tile
//     touches neither l nor O
//   * key_add is a per-key additive f32 term (natural-log domain, as the reference applies it) staged with the
//     tile; keys >= N get -inf; K rows / V^T columns beyond N are zeroed IN LDS on the tail tile only, so garbage
//     (even NaN) in the padded workspace can never reach an accumulator
#include <cstdio>
#include "common.h"
#include "kernels.h"

namespace uvl {

template <int N_> __device__ __forceinline__ void attn_wait_vmcnt() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N_) : "memory"); }

#define ATTN_LOG2E 1.4426950408889634f
#define ATTN_DEFER 8.0f

template <int QW, int KS, int NS>
__device__ __forceinline__ void attn_body(const AttnParams& p, const int bx, const int h, const int b, char* smem) {
    constexpr int NWAVES = QW * KS;
    constexpr int K_BYTES = 8192, V_BYTES = 8192, ADD_BYTES = 256, FLAG_BYTES = 16;
    constexpr int SLOT = K_BYTES + V_BYTES + ADD_BYTES + FLAG_BYTES;
    constexpr int STAGE = KS * SLOT;
    constexpr int KV_PER_WAVE = (16 * KS) / NWAVES;      // K/V DMA instructions per wave per round (16 per slot)
    constexpr int LPW = KV_PER_WAVE + 1;                 // + one key_add DMA, issued only by the slot's owner wave (qw == 0)
    constexpr int NSM2 = NS >= 2 ? NS - 2 : 0;           // NS == 1: "single shot" -- every key tile of the head is resident at once
    static_assert((16 * KS) % NWAVES == 0 && LPW * NSM2 <= 63, "geometry");
    // single shot with one query block: a wave stages the 16 pieces of ITS OWN key tile (piece index = compile-time constant, no
    // K / V^T selection per instruction) and consumes nothing else before the merge -- no barrier in front of the tile either
    constexpr bool OWN = (NS == 1 && QW == 1);
    // otherwise instruction i of wave w covers piece (w + NWAVES * i) & 15; when NWAVES divides 8, whether that is a K piece (< 8)
    // or a V^T piece depends on i alone -- a compile-time choice per instruction instead of a per-wave select
    constexpr bool P2 = (8 % NWAVES == 0);
#define ATTN_IS_K(i, piece) (OWN ? ((i) < 8) : (P2 ? (((i) % (16 / NWAVES)) < (8 / NWAVES)) : ((piece) < 8)))

    const int tid = threadIdx.x, lane = tid & 63, half = lane >> 5;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int qw = wave / KS, ks = wave % KS;
    const bool add_owner = (qw == 0);                    // exactly one wave per slot stages (and rewrites) its key_add row
    const int N = p.N, Npad = p.Npad;
    const size_t bh = (size_t)b * p.H + h;
    const bf16_t* __restrict__ Q = p.q + bh * Npad * 64;
    const bf16_t* __restrict__ K = p.k + bh * Npad * 64;
    const bf16_t* __restrict__ Vt = p.vt + bh * 64 * Npad;
    const float* __restrict__ kadd = p.key_add + (size_t)b * p.key_add_stride;

    const int nt = (N + 63) >> 6;                 // key tiles
    const int rounds = (nt + KS - 1) / KS;
    const int q0 = (bx * QW + qw) * 32;
    const int qrow = q0 + (lane & 31);
    const int qld = qrow < N ? qrow : N - 1;
    bf16x8 qf[4];
#pragma unroll
    for (int kk = 0; kk < 4; ++kk) qf[kk] = *reinterpret_cast<const bf16x8*>(Q + (size_t)qld * 64 + (2 * kk + half) * 8);

    // DMA plan of this wave: instruction i covers global K/V piece g = wave + NWAVES*i: slot g/16, piece g%16
    // (pieces 0-7 = K rows 8j..8j+7, pieces 8-15 = V^T rows), plus key_add of the slot this wave owns.
    // Per-lane source pointers are formed once; a round only adds the (wave-uniform) tile offset.
    const bf16_t* src[KV_PER_WAVE];
#pragma unroll
    for (int i = 0; i < KV_PER_WAVE; ++i) {
        const int g = OWN ? 16 * ks + i : wave + NWAVES * i, piece = OWN ? i : (g & 15);
        const int row = 8 * (piece & 7) + (lane >> 3);
        const int chunk = (lane & 7) ^ ((row >> 1) & 7);               // logical chunk that belongs at this physical slot
        src[i] = ATTN_IS_K(i, piece) ? K + (size_t)row * 64 + chunk * 8 : Vt + (size_t)row * Npad + chunk * 8;
    }
    auto issue = [&](int r) __attribute__((always_inline)) {
        char* st = smem + (r % NS) * STAGE;
#pragma unroll
        for (int i = 0; i < KV_PER_WAVE; ++i) {
            const int g = OWN ? 16 * ks + i : wave + NWAVES * i, slot = OWN ? ks : (g >> 4), piece = OWN ? i : (g & 15);
            int kt = r * KS + slot;
            kt = kt < nt ? kt : nt - 1;                                // tile does not exist: harmless re-read, never consumed
            const size_t off = ATTN_IS_K(i, piece) ? (size_t)kt * 64 * 64 : (size_t)kt * 64;
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(src[i] + off),
                                             (__attribute__((address_space(3))) void*)(st + slot * SLOT + piece * 1024), 16, 0, 0);
        }
        if (add_owner) {
            int kt = r * KS + ks;
            kt = kt < nt ? kt : nt - 1;
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(kadd + kt * 64 + lane),
                                             (__attribute__((address_space(3))) void*)(st + ks * SLOT + K_BYTES + V_BYTES), 4, 0, 0);
        }
    };
    // after this wave's own DMAs of round r have landed: finish the tile image (own-lane rewrites only)
    auto fixup = [&](int r) __attribute__((always_inline)) {
        char* st = smem + (r % NS) * STAGE;
        if (add_owner) {   // key_add -> log2 domain, -inf beyond N, and the per-tile "carries a mask" flag
            const int slot = ks;
            const int k0 = (r * KS + slot) * 64;
            float* sA = reinterpret_cast<float*>(st + slot * SLOT + K_BYTES + V_BYTES);
            float a = sA[lane] * ATTN_LOG2E;
            if (k0 + lane >= N) a = -INFINITY;
            sA[lane] = a;
            const bool any = __any(a != 0.f);
            if (lane == 0) *reinterpret_cast<int*>(st + slot * SLOT + K_BYTES + V_BYTES + ADD_BYTES) = any ? 1 : 0;
        }
#pragma unroll
        for (int i = 0; i < KV_PER_WAVE; ++i) {
            const int g = OWN ? 16 * ks + i : wave + NWAVES * i, slot = OWN ? ks : (g >> 4), piece = OWN ? i : (g & 15);
            const int k0 = (r * KS + slot) * 64;
            if (k0 < N && k0 + 64 > N) {                               // wave-uniform: only the tail tile
                const int row = 8 * (piece & 7) + (lane >> 3);
                char* at = st + slot * SLOT + piece * 1024 + lane * 16;
                if (ATTN_IS_K(i, piece)) {
                    if (k0 + row >= N) *reinterpret_cast<u32x4*>(at) = u32x4{0u, 0u, 0u, 0u};
                } else {
                    const int chunk = (lane & 7) ^ ((row >> 1) & 7);
                    const int kb = k0 + chunk * 8;
                    if (kb + 8 > N) {
                        u32x4 v = *reinterpret_cast<u32x4*>(at);
#pragma unroll
                        for (int e = 0; e < 4; ++e) {
                            uint32_t wv = v[e];
                            if (kb + 2 * e >= N) wv &= 0xffff0000u;
                            if (kb + 2 * e + 1 >= N) wv &= 0x0000ffffu;
                            v[e] = wv;
                        }
                        *reinterpret_cast<u32x4*>(at) = v;
                    }
                }
            }
        }
    };

    // lane-dependent LDS offsets of the fragments (tile-independent): K chunk 2kk+half of row lane&31 (+4096 for the
    // second 32-key block), V^T chunks 0..7 of row lane&31 (+4096 for the second 32 d-rows), 8-byte half = lane half
    // The QK^T A-operand of lane m reads K row perm(m) = m with bits 2 and 3 swapped.  S^T register r of lane-half g then
    // holds key 16(r>>3) + 8g + (r&7) of its 32-key block, i.e. every P fragment (registers 8t..8t+7) is 8 CONTIGUOUS
    // keys and the matching V^T A-fragment is ONE conflict-free ds_read_b128 (chunk 4jb + 2t + g of row d).
    const int m31 = lane & 31;
    const int kperm = (m31 & 0x13) | ((m31 & 4) << 1) | ((m31 & 8) >> 1);
    int koff[4], voff2[2][2];
#pragma unroll
    for (int kk = 0; kk < 4; ++kk) koff[kk] = swz128(kperm, 2 * kk + half);
#pragma unroll
    for (int jb = 0; jb < 2; ++jb)
#pragma unroll
        for (int t = 0; t < 2; ++t) voff2[jb][t] = swz128(m31, 4 * jb + 2 * t + half);

    f32x16 o[2];
#pragma unroll
    for (int r = 0; r < 16; ++r) { o[0][r] = 0.f; o[1][r] = 0.f; }
    float m_run = -INFINITY, l_run = 0.f;                 // log2-domain running max, per-lane partial row sum
    const float CS = p.q_prescaled ? 1.0f : 0.125f * ATTN_LOG2E;   // score scale folded with log2(e) (already in q when pre-scaled)

    if (NS == 1) issue(0);                                // single shot: rounds == 1, all tiles requested at once
#pragma unroll
    for (int t = 0; t < NS - 1; ++t)
        if (t < rounds) issue(t);
    for (int rd = 0; rd < rounds; ++rd) {
        const int ahead = rounds - 1 - rd;
        if (NS >= 2 && ahead >= NS - 2) {                 // leave NS-2 rounds of this wave's own DMAs in flight
            if (add_owner) attn_wait_vmcnt<LPW * NSM2>();
            else attn_wait_vmcnt<KV_PER_WAVE * NSM2>();
        } else if (NS > 3 && ahead == 1) {
            if (add_owner) attn_wait_vmcnt<LPW>();
            else attn_wait_vmcnt<KV_PER_WAVE>();
        } else {
            attn_wait_vmcnt<0>();
        }
        fixup(rd);
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // the fix-up's LDS writes are done before the barrier
        if (!OWN) __builtin_amdgcn_s_barrier();              // OWN: the tile was staged and fixed up by this wave alone
        if (NS >= 2 && rd + NS - 1 < rounds) issue(rd + NS - 1);     // its stage was last read in round rd-1
        if (rd * KS + ks < nt) {
            const char* sK = smem + (rd % NS) * STAGE + ks * SLOT;
            const char* sV = sK + K_BYTES;
            const float* sA = reinterpret_cast<const float*>(sV + V_BYTES);
            const int masked = __builtin_amdgcn_readfirstlane(*reinterpret_cast<const int*>(sV + V_BYTES + ADD_BYTES));

            // ---- S^T = K Q^T for two 32-key blocks ----
            f32x16 s[2];
            const f32x16 zero16 = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int jb = 0; jb < 2; ++jb) {
#pragma unroll
                for (int kk = 0; kk < 4; ++kk) {
                    const bf16x8 kf = *reinterpret_cast<const bf16x8*>(sK + jb * 4096 + koff[kk]);
                    s[jb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kf, qf[kk], kk == 0 ? zero16 : s[jb], 0, 0, 0);
                }
            }
            // ---- log2-domain scores, tile max ----
            float tmax = -INFINITY;
            if (masked) {                                 // wave-uniform: t = s*c + add
#pragma unroll
                for (int jb = 0; jb < 2; ++jb)
#pragma unroll
                    for (int gq = 0; gq < 4; ++gq) {               // registers 4gq..4gq+3 = keys 16(gq>>1) + 8g + 4(gq&1) + 0..3
                        const float4 a4 = *reinterpret_cast<const float4*>(sA + 32 * jb + 16 * (gq >> 1) + 8 * half + 4 * (gq & 1));
                        const float av[4] = {a4.x, a4.y, a4.z, a4.w};
#pragma unroll
                        for (int e = 0; e < 4; ++e) {
                            const float v = fmaf(s[jb][4 * gq + e], CS, av[e]);
                            s[jb][4 * gq + e] = v;
                            tmax = fmaxf(tmax, v);
                        }
                    }
            } else {                                      // max of the raw scores, scaled once (c > 0)
#pragma unroll
                for (int jb = 0; jb < 2; ++jb)
#pragma unroll
                    for (int r = 0; r < 16; ++r) tmax = fmaxf(tmax, s[jb][r]);
                tmax *= CS;
            }
            tmax = fmaxf(tmax, __shfl_xor(tmax, 32, 64));
            if (__any(tmax > m_run + ATTN_DEFER)) {       // rare after the first tiles: raise the max, rescale l and O
                const float m_new = fmaxf(m_run, tmax);
                const float alpha = __builtin_amdgcn_exp2f(m_run - m_new);
                m_run = m_new;
                l_run *= alpha;
#pragma unroll
                for (int r = 0; r < 16; ++r) { o[0][r] *= alpha; o[1][r] *= alpha; }
            }
            // exponent arguments and the row-sum partials are formed two at a time (v_pk_fma_f32 / v_pk_add_f32): the VALU,
            // not the MFMA pipe, bounds this kernel (32 quarter-rate v_exp_f32 per lane and tile cost as much as the 16 MFMAs)
            f32x2 psum2 = {0.f, 0.f};
            if (masked) {
                const f32x2 nm2 = {-m_run, -m_run};
#pragma unroll
                for (int jb = 0; jb < 2; ++jb)
#pragma unroll
                    for (int r = 0; r < 16; r += 2) {
                        const f32x2 a = f32x2{s[jb][r], s[jb][r + 1]} + nm2;
                        const f32x2 pv = {__builtin_amdgcn_exp2f(a[0]), __builtin_amdgcn_exp2f(a[1])};
                        s[jb][r] = pv[0];
                        s[jb][r + 1] = pv[1];
                        psum2 += pv;
                    }
            } else {
                const f32x2 nm2 = {-m_run, -m_run}, cs2 = {CS, CS};
#pragma unroll
                for (int jb = 0; jb < 2; ++jb)
#pragma unroll
                    for (int r = 0; r < 16; r += 2) {
                        const f32x2 a = __builtin_elementwise_fma(f32x2{s[jb][r], s[jb][r + 1]}, cs2, nm2);
                        const f32x2 pv = {__builtin_amdgcn_exp2f(a[0]), __builtin_amdgcn_exp2f(a[1])};
                        s[jb][r] = pv[0];
                        s[jb][r + 1] = pv[1];
                        psum2 += pv;
                    }
            }
            l_run += psum2[0] + psum2[1];

            // ---- O^T += V^T P^T ----
#pragma unroll
            for (int jb = 0; jb < 2; ++jb)
#pragma unroll
                for (int t = 0; t < 2; ++t) {
                    union { uint32_t u[4]; bf16x8 v; } pf;
#pragma unroll
                    for (int e = 0; e < 4; ++e) pf.u[e] = pack_bf16x2(s[jb][8 * t + 2 * e], s[jb][8 * t + 2 * e + 1]);
#pragma unroll
                    for (int db = 0; db < 2; ++db) {                   // V^T rows 32db + m, keys 32jb + 16t + 8g .. +7: one chunk
                        const bf16x8 vf = *reinterpret_cast<const bf16x8*>(sV + db * 4096 + voff2[jb][t]);
                        o[db] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vf, pf.v, o[db], 0, 0, 0);
                    }
                }
        }
    }

    bf16_t* dst = p.o + ((size_t)b * N + (qrow < N ? qrow : 0)) * (p.H * 64) + h * 64;
    if (KS == 1) {
        const float l_tot = l_run + __shfl_xor(l_run, 32, 64);
        const float inv = 1.0f / l_tot;
        if (qrow < N) {
#pragma unroll
            for (int db = 0; db < 2; ++db)
#pragma unroll
                for (int gq = 0; gq < 4; ++gq) {
                    const int d0 = 32 * db + 8 * gq + 4 * half;
                    uint2 w;
                    w.x = pack_bf16x2(o[db][4 * gq + 0] * inv, o[db][4 * gq + 1] * inv);
                    w.y = pack_bf16x2(o[db][4 * gq + 2] * inv, o[db][4 * gq + 3] * inv);
                    *reinterpret_cast<uint2*>(dst + d0) = w;
                }
        }
    } else {
        // ---- merge the KS key-split partials: exchange [wave][34][64] floats through LDS (the ring is dead now) ----
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        float* xch = reinterpret_cast<float*>(smem);
        float* mine = xch + (size_t)wave * 34 * 64;
#pragma unroll
        for (int r = 0; r < 16; ++r) { mine[r * 64 + lane] = o[0][r]; mine[(16 + r) * 64 + lane] = o[1][r]; }
        mine[32 * 64 + lane] = m_run;
        mine[33 * 64 + lane] = l_run;
        __syncthreads();
        const float* grp = xch + (size_t)(qw * KS) * 34 * 64;    // the KS partials of this query block
        float mstar = -INFINITY;
#pragma unroll
        for (int w = 0; w < KS; ++w) mstar = fmaxf(mstar, grp[(w * 34 + 32) * 64 + lane]);
        float sc[KS], lsum = 0.f;
#pragma unroll
        for (int w = 0; w < KS; ++w) {
            sc[w] = __builtin_amdgcn_exp2f(grp[(w * 34 + 32) * 64 + lane] - mstar);   // exp2(-inf) = 0 for a wave that saw no tile
            lsum += grp[(w * 34 + 33) * 64 + lane] * sc[w];
        }
        lsum += __shfl_xor(lsum, 32, 64);
        const float inv = 1.0f / lsum;
        // 8 groups of 4 accumulator registers (4 consecutive d of one row group); wave ks merges groups ks, ks+KS, ...
        for (int g4 = ks; g4 < 8; g4 += KS) {
            const int r0 = 4 * g4;
            float v[4];
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                float a = 0.f;
#pragma unroll
                for (int w = 0; w < KS; ++w) a += grp[(w * 34 + r0 + e) * 64 + lane] * sc[w];
                v[e] = a * inv;
            }
            if (qrow < N) {
                const int db = r0 >> 4, gq = (r0 & 15) >> 2;
                const int d0 = 32 * db + 8 * gq + 4 * half;
                uint2 w2;
                w2.x = pack_bf16x2(v[0], v[1]);
                w2.y = pack_bf16x2(v[2], v[3]);
                *reinterpret_cast<uint2*>(dst + d0) = w2;
            }
        }
    }
}

// Workgroup -> (query block, head, sample).  Workgroup b runs on XCD b % 8 (dispatch order; speed only) and every XCD has its own L2.
// mapped: the query blocks of one head are handed to ONE XCD (contiguous runs of the head-major order per XCD), so its K / V^T
// tiles enter one L2 instead of up to eight -- +1.8 % / +0.7 % on the frame at 8 / 32 sequences.  For one sequence (a single
// round of workgroups, all starting together) the plain order is faster (-1.3 % with the map: the 17 workgroups of a head then
// hammer the same L2 lines at the same moment), so the launcher maps only grids of several workgroups per CU.
__device__ __forceinline__ bool attn_decode_block(int blk, int total, int nqb, int H, bool mapped, int& qb, int& h, int& b) {
    const int xcd = blk & 7, idx = blk >> 3;
    const int cnt = (total + 7) >> 3;
    const int L = mapped ? xcd * cnt + idx : blk;
    if ((mapped && idx >= cnt) || L >= total) return false;
    qb = L % nqb;
    const int r = L / nqb;
    h = r % H;
    b = r / H;
    return true;
}

template <int QW, int KS, int NS>
__global__ __launch_bounds__(64 * QW * KS, (KS == 1 ? 3 : 1)) void attn_kernel(const AttnParams p) {
    extern __shared__ __attribute__((aligned(16))) char smem[];      // NS * STAGE (>= the merge exchange area)
    const int nqb = (p.N + 32 * QW - 1) / (32 * QW);
    int qb, h, b;
    if (!attn_decode_block((int)blockIdx.x, nqb * p.H * p.B, nqb, p.H, p.xcd_map != 0, qb, h, b)) return;
    attn_body<QW, KS, NS>(p, qb, h, b, smem);
}

// Two independent attention problems in one launch (batch-1 frames: the 40-token text-branch attention rides on the visual
// one's configuration).  1-D grid, problem A owns [0, blocks_a); (query block, head, sample) are decoded per problem.
template <int QW, int KS, int NS>
__global__ __launch_bounds__(64 * QW * KS, (KS == 1 ? 3 : 1)) void attn_pair_kernel(const AttnParams pa, const AttnParams pb, int blocks_a) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    if ((int)blockIdx.x < blocks_a) {                 // blocks_a is a multiple of 8: the XCD relation of the map is preserved
        const int nqb = (pa.N + 32 * QW - 1) / (32 * QW);
        int qb, h, b;
        if (!attn_decode_block((int)blockIdx.x, nqb * pa.H * pa.B, nqb, pa.H, pa.xcd_map != 0, qb, h, b)) return;
        attn_body<QW, KS, NS>(pa, qb, h, b, smem);
    } else {
        int id = (int)blockIdx.x - blocks_a;
        const int nqb = (pb.N + 32 * QW - 1) / (32 * QW), qb = id % nqb;
        id /= nqb;
        attn_body<QW, KS, NS>(pb, qb, id % pb.H, id / pb.H, smem);
    }
}

// ------------------------------------------------------------------------------------------------
// attn_stream_kernel<NS>: the batched shape (at least one 128-query workgroup per CU).  Same data layout and MFMA operand
// trick as attn_body (S^T = K Q^T, P^T is the B operand of V^T P^T with no data movement), but the steady-state tile is
// stripped to what the MFMA and the exponentials need:
//   * Q arrives PRE-SCALED by log2(e)/8 (the QKV GEMM epilogue multiplies its q columns before rounding to bf16), and the
//     running maximum enters the score MFMA as its C operand: a persistent register block holds -m broadcast, so the MFMA
//     chain delivers s - m directly.  No scale, no subtract, no max in the common tile: exp2, row-sum add, bf16 pack.
//   * the common tile is SPECULATIVE: it assumes the running max is still good enough.  If some lane's partial row sum of
//     the tile exceeds 2^8 (or is inf / NaN) nothing has been committed yet -- the tile is redone on the exact path (scores
//     with C = 0, true row max, rescale of l and O, new -m block).  Tiles that carry a mask term, the first tile and the
//     tail tile take the exact path directly.  Same deferral bound as attn_body's ATTN_DEFER.
//   * LDS addressing is lane-constant: the loop is unrolled by the ring depth, so a stage is an immediate offset of every
//     ds_read and a DMA costs one SGPR base update per tile (K / V^T / key_add bases walk by 8192 / 128 / 256 bytes)
//   * every wave stages its own copy of the tile's key_add row by DMA (4 bytes per lane, all-DMA loop: no ordinary load for
//     hipcc to drain the ring for) and derives the tile's "carries a mask" flag from it; waves whose 32 queries lie beyond
//     N serve DMA and barriers only
// ------------------------------------------------------------------------------------------------
template <int V_> struct AttnIC { static constexpr int value = V_; };

template <int NS>
__global__ __launch_bounds__(256, 3) void attn_stream_kernel(const AttnParams p) {
    constexpr int STAGE = 16384;                          // K tile 8 KB + V^T tile 8 KB
    constexpr int KADD0 = NS * STAGE;                     // [NS][4 waves][64] f32 key_add rows
    constexpr int VM = 5;                                 // VMEM operations per wave and tile: 2 K + 2 V^T pieces + the key_add row
    static_assert(NS == 2 || NS == 3, "ring depth");
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int nqb = (p.N + 127) / 128;
    int qb, h, b;
    if (!attn_decode_block((int)blockIdx.x, nqb * p.H * p.B, nqb, p.H, p.xcd_map != 0, qb, h, b)) return;

    const int tid = threadIdx.x, lane = tid & 63, half = lane >> 5;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int N = p.N, Npad = p.Npad;
    const size_t bh = (size_t)b * p.H + h;
    const bf16_t* __restrict__ Q = p.q + bh * Npad * 64;
    const char* Kb = reinterpret_cast<const char*>(p.k + bh * Npad * 64);
    const char* Vb = reinterpret_cast<const char*>(p.vt + bh * 64 * Npad);
    const char* Ab = reinterpret_cast<const char*>(p.key_add + (size_t)b * p.key_add_stride);
    const int nt = (N + 63) >> 6;
    const int q0 = (qb * 4 + wave) * 32;
    const bool active = q0 < N;                           // wave-uniform
    const int qrow = q0 + (lane & 31);
    const int qld = qrow < N ? qrow : N - 1;
    bf16x8 qf[4];
#pragma unroll
    for (int kk = 0; kk < 4; ++kk) qf[kk] = *reinterpret_cast<const bf16x8*>(Q + (size_t)qld * 64 + (2 * kk + half) * 8);
    if (!p.q_prescaled) {                                 // test entry point: raw q, scaled (and rounded once more) here
#pragma unroll
        for (int kk = 0; kk < 4; ++kk)
#pragma unroll
            for (int e = 0; e < 8; ++e) qf[kk][e] = f2bf(bf2f(qf[kk][e]) * (0.125f * ATTN_LOG2E));
    }

    // DMA plan: instruction i of wave w fills piece w + 4 (i & 1) (8 rows) of the K tile (i < 2) or of the V^T tile
    uint32_t voff[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int piece = wave + 4 * (i & 1);
        const int row = 8 * piece + (lane >> 3);
        const int chunk = (lane & 7) ^ ((row >> 1) & 7);
        voff[i] = i < 2 ? (uint32_t)(row * 128 + chunk * 16) : (uint32_t)row * (uint32_t)(Npad * 2) + (uint32_t)(chunk * 16);
    }
    auto pin = [](const char* q) __attribute__((always_inline)) {
        const uint64_t u = reinterpret_cast<uint64_t>(q);
        const uint32_t lo = __builtin_amdgcn_readfirstlane((uint32_t)u), hi = __builtin_amdgcn_readfirstlane((uint32_t)(u >> 32));
        return reinterpret_cast<const char*>(((uint64_t)hi << 32) | lo);
    };
#define ATTN_GLDS(src, dst, bytes) __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(src), \
                                                                    (__attribute__((address_space(3))) void*)(dst), bytes, 0, 0)
    auto issue = [&](int t, auto stc) __attribute__((always_inline)) {
        constexpr int ST = decltype(stc)::value;
        const char* kt = pin(Kb + (size_t)t * 8192);
        const char* vt = pin(Vb + (size_t)t * 128);
        const char* at = pin(Ab + (size_t)t * 256);
        char* st = smem + ST * STAGE;
        ATTN_GLDS(kt + voff[0], st + wave * 1024, 16);
        ATTN_GLDS(kt + voff[1], st + (wave + 4) * 1024, 16);
        ATTN_GLDS(vt + voff[2], st + 8192 + wave * 1024, 16);
        ATTN_GLDS(vt + voff[3], st + 8192 + (wave + 4) * 1024, 16);
        ATTN_GLDS(at + lane * 4, smem + KADD0 + (ST * 4 + wave) * 256, 4);
    };

    // lane-constant fragment offsets (see attn_body): K chunk 2kk+half of row perm(lane & 31), V^T chunk 4jb+2t+half of row lane & 31
    const int m31 = lane & 31;
    const int kperm = (m31 & 0x13) | ((m31 & 4) << 1) | ((m31 & 8) >> 1);
    int koff[4], voff2[2][2];
#pragma unroll
    for (int kk = 0; kk < 4; ++kk) koff[kk] = swz128(kperm, 2 * kk + half);
#pragma unroll
    for (int jb = 0; jb < 2; ++jb)
#pragma unroll
        for (int t = 0; t < 2; ++t) voff2[jb][t] = swz128(m31, 4 * jb + 2 * t + half);

    f32x16 o[2], negm;
#pragma unroll
    for (int r = 0; r < 16; ++r) { o[0][r] = 0.f; o[1][r] = 0.f; negm[r] = 0.f; }
    float m_run = -INFINITY, l_run = 0.f;

    auto tile = [&](const int t, auto stc) __attribute__((always_inline)) {
        constexpr int ST = decltype(stc)::value;
        char* sK = smem + ST * STAGE;
        char* sV = sK + 8192;
        float* sA = reinterpret_cast<float*>(smem + KADD0 + (ST * 4 + wave) * 256);
        // this wave's DMAs of tile t have landed; the next tile's may stay in flight
        if (NS == 3 && t + 1 < nt) attn_wait_vmcnt<VM>();
        else attn_wait_vmcnt<0>();
        const int k0 = t * 64;
        if (t == nt - 1 && (N & 63)) {                    // tail tile: zero K rows / V^T columns beyond N in LDS (own pieces)
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int piece = wave + 4 * (i & 1);
                const int row = 8 * piece + (lane >> 3);
                char* at = sK + (i < 2 ? 0 : 8192) + piece * 1024 + lane * 16;
                if (i < 2) {
                    if (k0 + row >= N) *reinterpret_cast<u32x4*>(at) = u32x4{0u, 0u, 0u, 0u};
                } else {
                    const int chunk = (lane & 7) ^ ((row >> 1) & 7);
                    const int kb = k0 + chunk * 8;
                    if (kb + 8 > N) {
                        u32x4 v = *reinterpret_cast<u32x4*>(at);
#pragma unroll
                        for (int e = 0; e < 4; ++e) {
                            uint32_t wv = v[e];
                            if (kb + 2 * e >= N) wv &= 0xffff0000u;
                            if (kb + 2 * e + 1 >= N) wv &= 0x0000ffffu;
                            v[e] = wv;
                        }
                        *reinterpret_cast<u32x4*>(at) = v;
                    }
                }
            }
        }
        float ka = 0.f;
        if (active) ka = sA[lane];                        // own DMA, own wait: no barrier needed for this row
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        if (t + NS - 1 < nt) issue(t + NS - 1, AttnIC<(ST + NS - 1) % NS>{});   // its stage was last read in iteration t-1
        if (!active) return;

        ka = (k0 + lane < N) ? ka * ATTN_LOG2E : -INFINITY;                     // log2 domain; keys beyond N never count
        const bool masked = __any(ka != 0.f);
        const f32x16 zero16 = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
        f32x16 s[2];
        float psum = 0.f;
        bool redo = masked || t == 0;
        if (!redo) {
            // ---- speculative tile: s - m straight from the MFMA, exp2, row-sum partials ----
#pragma unroll
            for (int jb = 0; jb < 2; ++jb)
#pragma unroll
                for (int kk = 0; kk < 4; ++kk) {
                    const bf16x8 kf = *reinterpret_cast<const bf16x8*>(sK + jb * 4096 + koff[kk]);
                    s[jb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kf, qf[kk], kk == 0 ? negm : s[jb], 0, 0, 0);
                }
            float ps[4] = {0.f, 0.f, 0.f, 0.f};           // four short add chains instead of one long one
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                s[0][r] = __builtin_amdgcn_exp2f(s[0][r]);
                s[1][r] = __builtin_amdgcn_exp2f(s[1][r]);
                ps[r & 1] += s[0][r];
                ps[2 + (r & 1)] += s[1][r];
            }
            psum = (ps[0] + ps[1]) + (ps[2] + ps[3]);
            redo = !__all(psum <= 256.0f);                // also catches inf / NaN
        }
        if (redo) {
            // ---- exact tile: scores with C = 0, mask term, true row max, rescale ----
#pragma unroll
            for (int jb = 0; jb < 2; ++jb)
#pragma unroll
                for (int kk = 0; kk < 4; ++kk) {
                    const bf16x8 kf = *reinterpret_cast<const bf16x8*>(sK + jb * 4096 + koff[kk]);
                    s[jb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kf, qf[kk], kk == 0 ? zero16 : s[jb], 0, 0, 0);
                }
            if (masked) {                                 // + key_add * log2(e), read from the staged row (no LDS write: that would drain the ring)
#pragma unroll
                for (int jb = 0; jb < 2; ++jb)
#pragma unroll
                    for (int gq = 0; gq < 4; ++gq) {      // registers 4gq..4gq+3 = keys 16(gq>>1) + 8g + 4(gq&1) + 0..3
                        const int kq = 32 * jb + 16 * (gq >> 1) + 8 * half + 4 * (gq & 1);
                        const float4 a4 = *reinterpret_cast<const float4*>(sA + kq);
                        const float av[4] = {a4.x, a4.y, a4.z, a4.w};
#pragma unroll
                        for (int e = 0; e < 4; ++e) s[jb][4 * gq + e] = fmaf(av[e], ATTN_LOG2E, s[jb][4 * gq + e]);
                    }
                if (k0 + 64 > N) {                        // tail tile: keys beyond N never count (their key_add may be anything)
#pragma unroll
                    for (int jb = 0; jb < 2; ++jb)
#pragma unroll
                        for (int r = 0; r < 16; ++r) {
                            const int key = k0 + 32 * jb + 16 * (r >> 3) + 8 * half + (r & 7);
                            s[jb][r] = key < N ? s[jb][r] : -INFINITY;
                        }
                }
            }
            float tmax = -INFINITY;
#pragma unroll
            for (int r = 0; r < 16; ++r) tmax = fmaxf(tmax, fmaxf(s[0][r], s[1][r]));
            tmax = fmaxf(tmax, __shfl_xor(tmax, 32, 64));
            const float m_new = fmaxf(m_run, tmax);
            const float alpha = __builtin_amdgcn_exp2f(m_run - m_new);          // first tile: exp2(-inf) = 0
            m_run = m_new;
            l_run *= alpha;
#pragma unroll
            for (int r = 0; r < 16; ++r) { o[0][r] *= alpha; o[1][r] *= alpha; negm[r] = -m_new; }
            float ps[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                s[0][r] = __builtin_amdgcn_exp2f(s[0][r] - m_new);
                s[1][r] = __builtin_amdgcn_exp2f(s[1][r] - m_new);
                ps[r & 1] += s[0][r];
                ps[2 + (r & 1)] += s[1][r];
            }
            psum = (ps[0] + ps[1]) + (ps[2] + ps[3]);
        }
        l_run += psum;
        // ---- O^T += V^T P^T ----
#pragma unroll
        for (int jb = 0; jb < 2; ++jb)
#pragma unroll
            for (int t2 = 0; t2 < 2; ++t2) {
                union { uint32_t u[4]; bf16x8 v; } pf;
#pragma unroll
                for (int e = 0; e < 4; ++e) pf.u[e] = pack_bf16x2(s[jb][8 * t2 + 2 * e], s[jb][8 * t2 + 2 * e + 1]);
#pragma unroll
                for (int db = 0; db < 2; ++db) {
                    const bf16x8 vf = *reinterpret_cast<const bf16x8*>(sV + db * 4096 + voff2[jb][t2]);
                    o[db] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vf, pf.v, o[db], 0, 0, 0);
                }
            }
    };

    issue(0, AttnIC<0>{});
    if (NS == 3 && nt > 1) issue(1, AttnIC<1>{});
    for (int t0 = 0; t0 < nt; t0 += NS) {
        tile(t0, AttnIC<0>{});
        if (t0 + 1 < nt) tile(t0 + 1, AttnIC<1>{});
        if (NS == 3 && t0 + 2 < nt) tile(t0 + 2, AttnIC<2 % NS>{});
    }
#undef ATTN_GLDS

    if (qrow < N) {
        const float l_tot = l_run + __shfl_xor(l_run, 32, 64);
        const float inv = 1.0f / l_tot;
        bf16_t* dst = p.o + ((size_t)b * N + qrow) * (p.H * 64) + h * 64;
#pragma unroll
        for (int db = 0; db < 2; ++db)
#pragma unroll
            for (int gq = 0; gq < 4; ++gq) {
                const int d0 = 32 * db + 8 * gq + 4 * half;
                uint2 w;
                w.x = pack_bf16x2(o[db][4 * gq + 0] * inv, o[db][4 * gq + 1] * inv);
                w.y = pack_bf16x2(o[db][4 * gq + 2] * inv, o[db][4 * gq + 3] * inv);
                *reinterpret_cast<uint2*>(dst + d0) = w;
            }
    }
}

// ------------------------------------------------------------------------------------------------
// attn_pipe_kernel: attn_stream_kernel's tile with the LDS round trips taken off the critical path.  PMC of the version above
// (profiles/r02_attention_pmc.md): a wave spends 35 % of its cycles in s_waitcnt -- hipcc places every fragment read directly
// in front of the MFMA that consumes it, eight exposed LDS latencies per tile.  Here ONE 8-fragment register block is
// time-shared between the K and the V^T fragments of a tile (K fragments are dead once the score MFMAs have issued, V^T
// fragments once the output MFMAs have), and every read is issued a phase ahead of its use:
//     top of tile t : the K fragments of tile t are already in registers (requested during the previous tile's P V phase)
//     A  8 score MFMAs, C operand = -m (+ the key_add term on a tile that carries a mask)
//     B  request the 8 V^T fragments of tile t into the same registers
//     C  exp2 / row sums (speculative, see attn_stream_kernel), 64 VALU instructions that cover B's latency
//     D  own DMAs of tile t+1 landed (vmcnt), barrier, request tile t+2 by DMA into the stage tile t-1 has left
//     E  bf16 packing + 8 output MFMAs
//     F  request the K fragments of tile t+1
// The barrier sits between C and E: a wave that reaches it has finished P V of tile t-1, so that stage is free, and everyone's
// share of tile t+1 is in LDS before anyone reads it in F.  LDS-DMA is issued from inline asm with an SGPR base and a 32-bit lane
// offset (the builtin form makes hipcc rebuild a 64-bit VGPR address per instruction inside the loop).
// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ void attn_dma16(uint32_t voff, const char* sbase, uint32_t lds_addr) {
    asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1" ::"v"(voff), "s"(sbase), "s"(lds_addr) : "memory");
}
__device__ __forceinline__ void attn_dma4(uint32_t voff, const char* sbase, uint32_t lds_addr) {
    asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dword %0, %1" ::"v"(voff), "s"(sbase), "s"(lds_addr) : "memory");
}
typedef __attribute__((address_space(3))) const char* lds_cptr;
__device__ __forceinline__ bf16x8 lds_read16(uint32_t addr) { return *reinterpret_cast<__attribute__((address_space(3))) const bf16x8*>((lds_cptr)(uintptr_t)addr); }

template <int OCC>
__global__ __launch_bounds__(256, OCC) void attn_pipe_kernel(const AttnParams p) {
    constexpr int NS = 3;
    constexpr int STAGE = 16384;                          // K tile 8 KB + V^T tile 8 KB
    constexpr int KADD0 = NS * STAGE;                     // [NS][4 waves][64] f32 key_add rows
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int nqb = (p.N + 127) / 128;
    int qb, h, b;
    if (!attn_decode_block((int)blockIdx.x, nqb * p.H * p.B, nqb, p.H, p.xcd_map != 0, qb, h, b)) return;

    const int tid = threadIdx.x, lane = tid & 63, half = lane >> 5;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int N = p.N, Npad = p.Npad;
    const size_t bh = (size_t)b * p.H + h;
    const bf16_t* __restrict__ Q = p.q + bh * Npad * 64;
    const int nt = (N + 63) >> 6;
    const int q0 = (qb * 4 + wave) * 32;
    const bool active = q0 < N && !(p.ablate & 2);        // wave-uniform
    const int qrow = q0 + (lane & 31);
    const int qld = qrow < N ? qrow : N - 1;
    bf16x8 qf[4];
#pragma unroll
    for (int kk = 0; kk < 4; ++kk) qf[kk] = *reinterpret_cast<const bf16x8*>(Q + (size_t)qld * 64 + (2 * kk + half) * 8);
    if (!p.q_prescaled) {                                 // test entry point: raw q, scaled (and rounded once more) here
#pragma unroll
        for (int kk = 0; kk < 4; ++kk)
#pragma unroll
            for (int e = 0; e < 8; ++e) qf[kk][e] = f2bf(bf2f(qf[kk][e]) * (0.125f * ATTN_LOG2E));
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // q is in registers before the first DMA: the loop's vmcnt counts only DMAs

    // DMA plan: instruction i of wave w fills piece w + 4 (i & 1) (8 rows) of the K tile (i < 2) or of the V^T tile
    uint32_t voff[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int piece = wave + 4 * (i & 1);
        const int row = 8 * piece + (lane >> 3);
        const int chunk = (lane & 7) ^ ((row >> 1) & 7);
        voff[i] = i < 2 ? (uint32_t)(row * 128 + chunk * 16) : (uint32_t)row * (uint32_t)(Npad * 2) + (uint32_t)(chunk * 16);
    }
    auto pin = [](const char* q) __attribute__((always_inline)) {
        const uint64_t u = reinterpret_cast<uint64_t>(q);
        const uint32_t lo = __builtin_amdgcn_readfirstlane((uint32_t)u), hi = __builtin_amdgcn_readfirstlane((uint32_t)(u >> 32));
        return reinterpret_cast<const char*>(((uint64_t)hi << 32) | lo);
    };
    const char* Kb = pin(reinterpret_cast<const char*>(p.k + bh * Npad * 64));
    const char* Vb = pin(reinterpret_cast<const char*>(p.vt + bh * 64 * Npad));
    const char* Ab = pin(reinterpret_cast<const char*>(p.key_add + (size_t)b * p.key_add_stride));
    const uint32_t lds0 = (uint32_t)(uintptr_t)(lds_cptr)smem;      // LDS byte address of the ring (0 unless something static precedes it)
    const uint32_t lds_w = lds0 + wave * 1024;
    const uint32_t lds_a = lds0 + KADD0 + wave * 256;
    const uint32_t lane4 = lane * 4;
    auto issue = [&](int t, auto stc) __attribute__((always_inline)) {
        constexpr int ST = decltype(stc)::value;
        const char* kt = Kb + (size_t)t * 8192;
        const char* vt = Vb + (size_t)t * 128;
        const char* at = Ab + (size_t)t * 256;
        attn_dma16(voff[0], kt, lds_w + ST * STAGE);
        attn_dma16(voff[1], kt, lds_w + ST * STAGE + 4096);
        attn_dma16(voff[2], vt, lds_w + ST * STAGE + 8192);
        attn_dma16(voff[3], vt, lds_w + ST * STAGE + 8192 + 4096);
        attn_dma4(lane4, at, lds_a + ST * 1024);
    };

    // lane-constant LDS addresses of the fragments: K chunk 2kk+half of row perm(lane & 31), V^T chunk 4jb+2t+half of row lane & 31
    const int m31 = lane & 31;
    const int kperm = (m31 & 0x13) | ((m31 & 4) << 1) | ((m31 & 8) >> 1);
    uint32_t kaddr[4], vaddr[2][2];
#pragma unroll
    for (int kk = 0; kk < 4; ++kk) kaddr[kk] = lds0 + swz128(kperm, 2 * kk + half);
#pragma unroll
    for (int jb = 0; jb < 2; ++jb)
#pragma unroll
        for (int t = 0; t < 2; ++t) vaddr[jb][t] = lds0 + 8192 + swz128(m31, 4 * jb + 2 * t + half);

    f32x16 o[2], negm;                                     // negm: -m broadcast, the C operand of a plain tile's score MFMAs
#pragma unroll
    for (int r = 0; r < 16; ++r) { o[0][r] = 0.f; o[1][r] = 0.f; negm[r] = 0.f; }
    float m_run = -INFINITY, l_run = 0.f;
    bf16x8 fr[8];                                          // the shared K / V^T fragment block

    auto read_k = [&](auto stc) __attribute__((always_inline)) {       // fr[4 jb + kk] = K fragment (key block jb, d step kk)
        constexpr int ST = decltype(stc)::value;
#pragma unroll
        for (int kk = 0; kk < 4; ++kk)
#pragma unroll
            for (int jb = 0; jb < 2; ++jb) fr[4 * jb + kk] = lds_read16(kaddr[kk] + ST * STAGE + jb * 4096);
    };
    auto read_v = [&](auto stc) __attribute__((always_inline)) {       // fr[4 jb + 2 t2 + db] = V^T fragment (d block db, keys 32 jb + 16 t2 ..)
        constexpr int ST = decltype(stc)::value;
#pragma unroll
        for (int jb = 0; jb < 2; ++jb)
#pragma unroll
            for (int t2 = 0; t2 < 2; ++t2)
#pragma unroll
                for (int db = 0; db < 2; ++db) fr[4 * jb + 2 * t2 + db] = lds_read16(vaddr[jb][t2] + ST * STAGE + db * 4096);
    };

    auto tile = [&](const int t, auto stc) __attribute__((always_inline)) {
        constexpr int ST = decltype(stc)::value;
        const int k0 = t * 64;
        f32x16 s[2];
        float psum = 0.f;
        if (active) {
            const float* sA = reinterpret_cast<const float*>(smem + KADD0 + ST * 1024 + wave * 256);
            const float ka_raw = sA[lane];
            const bool tail = k0 + 64 > N;                                          // wave-uniform
            const bool masked = __any((k0 + lane < N) ? (ka_raw != 0.f) : true);    // a mask term, or keys beyond N
            // C operand of the score MFMAs: (key_add * log2 e, -inf beyond N) - m; all registers equal -m on a plain tile
            auto c_operand = [&](float base, f32x16 (&c)[2]) __attribute__((always_inline)) {
#pragma unroll
                for (int jb = 0; jb < 2; ++jb)
#pragma unroll
                    for (int gq = 0; gq < 4; ++gq) {      // registers 4gq..4gq+3 = keys 32jb + 16(gq>>1) + 8 half + 4(gq&1) + 0..3
                        const int kq = 32 * jb + 16 * (gq >> 1) + 8 * half + 4 * (gq & 1);
                        const float4 a4 = *reinterpret_cast<const float4*>(sA + kq);
                        const float av[4] = {a4.x, a4.y, a4.z, a4.w};
#pragma unroll
                        for (int e = 0; e < 4; ++e) {
                            float v = fmaf(av[e], ATTN_LOG2E, base);
                            if (tail) v = (k0 + kq + e < N) ? v : -INFINITY;
                            c[jb][4 * gq + e] = v;
                        }
                    }
            };
            auto scores = [&](const f32x16& c0, const f32x16& c1) __attribute__((always_inline)) {
#pragma unroll
                for (int kk = 0; kk < 4; ++kk) {
                    s[0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fr[kk], qf[kk], kk == 0 ? c0 : s[0], 0, 0, 0);
                    s[1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fr[4 + kk], qf[kk], kk == 0 ? c1 : s[1], 0, 0, 0);
                }
            };
            auto exp_sum = [&]() __attribute__((always_inline)) {
                float ps[4] = {0.f, 0.f, 0.f, 0.f};       // four short add chains instead of one long one
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    s[0][r] = __builtin_amdgcn_exp2f(s[0][r]);
                    s[1][r] = __builtin_amdgcn_exp2f(s[1][r]);
                    ps[r & 1] += s[0][r];
                    ps[2 + (r & 1)] += s[1][r];
                }
                psum = (ps[0] + ps[1]) + (ps[2] + ps[3]);
            };
            bool redo = (t == 0);
            if (!redo) {
                // ---- A..C, speculative: s - m straight from the MFMA ----
                if (masked) {
                    c_operand(-m_run, s);
                    scores(s[0], s[1]);
                } else {
                    scores(negm, negm);
                }
                __builtin_amdgcn_sched_barrier(0);
                read_v(stc);
                __builtin_amdgcn_sched_barrier(0);
                exp_sum();
                redo = !__all(psum <= 256.0f);            // also catches inf / NaN
                if (redo) read_k(stc);                    // the exact path needs the K fragments again (LDS still holds the tile)
            }
            if (redo) {
                // ---- exact tile: scores with C = mask term only, true row max, rescale of l and O ----
                const f32x16 zero16 = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
                if (masked) {
                    c_operand(0.f, s);
                    scores(s[0], s[1]);
                } else {
                    scores(zero16, zero16);
                }
                __builtin_amdgcn_sched_barrier(0);
                read_v(stc);
                __builtin_amdgcn_sched_barrier(0);
                float tmax = -INFINITY;
#pragma unroll
                for (int r = 0; r < 16; ++r) tmax = fmaxf(tmax, fmaxf(s[0][r], s[1][r]));
                tmax = fmaxf(tmax, __shfl_xor(tmax, 32, 64));
                const float m_new = fmaxf(m_run, tmax);
                if (t != 0) {                             // first tile: l = 0, O = 0
                    const float alpha = __builtin_amdgcn_exp2f(m_run - m_new);
                    l_run *= alpha;
#pragma unroll
                    for (int r = 0; r < 16; ++r) { o[0][r] *= alpha; o[1][r] *= alpha; }
                }
                m_run = m_new;
#pragma unroll
                for (int r = 0; r < 16; ++r) { s[0][r] -= m_new; s[1][r] -= m_new; negm[r] = -m_new; }
                exp_sum();
            }
            l_run += psum;
        }
        // ---- D: tile t+1 is complete in LDS for everyone, the stage of tile t-1 is free ----
        attn_wait_vmcnt<0>();
        if (t + 1 == nt - 1 && (N & 63)) {                // tile t+1 is the tail tile: zero its K rows / V^T columns beyond N (own pieces)
            constexpr int SN = (ST + 1) % NS;
            const int k1 = (t + 1) * 64;
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int piece = wave + 4 * (i & 1);
                const int row = 8 * piece + (lane >> 3);
                char* at = smem + SN * STAGE + (i < 2 ? 0 : 8192) + piece * 1024 + lane * 16;
                if (i < 2) {
                    if (k1 + row >= N) *reinterpret_cast<u32x4*>(at) = u32x4{0u, 0u, 0u, 0u};
                } else {
                    const int chunk = (lane & 7) ^ ((row >> 1) & 7);
                    const int kb = k1 + chunk * 8;
                    if (kb + 8 > N) {
                        u32x4 v = *reinterpret_cast<u32x4*>(at);
#pragma unroll
                        for (int e = 0; e < 4; ++e) {
                            uint32_t wv = v[e];
                            if (kb + 2 * e >= N) wv &= 0xffff0000u;
                            if (kb + 2 * e + 1 >= N) wv &= 0x0000ffffu;
                            v[e] = wv;
                        }
                        *reinterpret_cast<u32x4*>(at) = v;
                    }
                }
            }
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        }
        __builtin_amdgcn_s_barrier();
        if (t + 2 < nt && !(p.ablate & 1)) issue(t + 2, AttnIC<(ST + 2) % NS>{});
        if (!active) return;
        // ---- E: O^T += V^T P^T ----
#pragma unroll
        for (int jb = 0; jb < 2; ++jb)
#pragma unroll
            for (int t2 = 0; t2 < 2; ++t2) {
                union { uint32_t u[4]; bf16x8 v; } pf;
#pragma unroll
                for (int e = 0; e < 4; ++e) pf.u[e] = pack_bf16x2(s[jb][8 * t2 + 2 * e], s[jb][8 * t2 + 2 * e + 1]);
#pragma unroll
                for (int db = 0; db < 2; ++db) o[db] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fr[4 * jb + 2 * t2 + db], pf.v, o[db], 0, 0, 0);
            }
        __builtin_amdgcn_sched_barrier(0);
        // ---- F: K fragments of the next tile ----
        if (t + 1 < nt) read_k(AttnIC<(ST + 1) % NS>{});
    };

    // prologue: tiles 0 and 1 requested; tile 0 complete (tail fix-up if it is the only tile) before its K fragments are read
    issue(0, AttnIC<0>{});
    if (nt > 1) { issue(1, AttnIC<1>{}); attn_wait_vmcnt<5>(); } else { attn_wait_vmcnt<0>(); }
    if (nt == 1 && (N & 63)) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int piece = wave + 4 * (i & 1);
            const int row = 8 * piece + (lane >> 3);
            char* at = smem + (i < 2 ? 0 : 8192) + piece * 1024 + lane * 16;
            if (i < 2) {
                if (row >= N) *reinterpret_cast<u32x4*>(at) = u32x4{0u, 0u, 0u, 0u};
            } else {
                const int chunk = (lane & 7) ^ ((row >> 1) & 7);
                const int kb = chunk * 8;
                if (kb + 8 > N) {
                    u32x4 v = *reinterpret_cast<u32x4*>(at);
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        uint32_t wv = v[e];
                        if (kb + 2 * e >= N) wv &= 0xffff0000u;
                        if (kb + 2 * e + 1 >= N) wv &= 0x0000ffffu;
                        v[e] = wv;
                    }
                    *reinterpret_cast<u32x4*>(at) = v;
                }
            }
        }
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    if (active) read_k(AttnIC<0>{});
    for (int t0 = 0; t0 < nt; t0 += NS) {
        tile(t0, AttnIC<0>{});
        if (t0 + 1 < nt) tile(t0 + 1, AttnIC<1>{});
        if (t0 + 2 < nt) tile(t0 + 2, AttnIC<2>{});
    }

    if (qrow < N) {
        const float l_tot = l_run + __shfl_xor(l_run, 32, 64);
        const float inv = 1.0f / l_tot;
        bf16_t* dst = p.o + ((size_t)b * N + qrow) * (p.H * 64) + h * 64;
#pragma unroll
        for (int db = 0; db < 2; ++db)
#pragma unroll
            for (int gq = 0; gq < 4; ++gq) {
                const int d0 = 32 * db + 8 * gq + 4 * half;
                uint2 w;
                w.x = pack_bf16x2(o[db][4 * gq + 0] * inv, o[db][4 * gq + 1] * inv);
                w.y = pack_bf16x2(o[db][4 * gq + 2] * inv, o[db][4 * gq + 3] * inv);
                *reinterpret_cast<uint2*>(dst + d0) = w;
            }
    }
}

template <int OCC>
static hipError_t launch_attn_pipe(const AttnParams& p_in, hipStream_t s) {
    constexpr size_t lds = (size_t)3 * 16384 + (size_t)3 * 4 * 256;
    auto kern = attn_pipe_kernel<OCC>;
    static bool attr_done = false;
    if (!attr_done) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) return e;
        attr_done = true;
    }
    g_last_kernel = OCC == 3 ? "attn_pipe_kernel<3>" : "attn_pipe_kernel<2>";
    const int total = ((p_in.N + 127) / 128) * p_in.H * p_in.B;
    AttnParams p = p_in;
    p.xcd_map = total >= 400 ? 1 : 0;
    p.ablate = g_tune_attn_abl;
    hipLaunchKernelGGL(kern, dim3(8 * ((total + 7) / 8)), dim3(256), lds, s, p);
    return hipGetLastError();
}

// ------------------------------------------------------------------------------------------------
// attn_persist_kernel<NW, NS>: attn_pipe_kernel's tile inside PERSISTENT workgroups.  Ablation of the one-item-per-workgroup
// kernels at B = 32, H = 16, N = 681 (profiles/r02_attention_pmc.md): skeleton without DMA or arithmetic (q load, barriers,
// output stores) 34 us, DMA + barriers alone 53 us, arithmetic alone ~58 us, everything 109 us -- the phases ADD, because
// every workgroup of a round loads q at the same moment, computes at the same moment and stores at the same moment.  Here a
// workgroup walks a list of (sample, head, query block) items and its K / V^T tile stream never stops: the DMA cursor runs
// NS-1 tiles ahead of the arithmetic ACROSS item boundaries, the next item's q fragments are requested during the current
// item's last tile, and the output stores of an item drain while the next item's first tiles are computed.
// NW waves x 32 queries share a tile (NW = 4: two workgroups per CU; NW = 8: one, half the L2 -> LDS traffic per query).
// ------------------------------------------------------------------------------------------------
template <int NW, int NS>
__global__ __launch_bounds__(64 * NW, NW == 4 ? 2 : 1) void attn_persist_kernel(const AttnParams p) {
    constexpr int STAGE = 16384;                          // K tile 8 KB + V^T tile 8 KB
    constexpr int KADD0 = NS * STAGE;                     // [NS][NW waves][64] f32 key_add rows
    constexpr int QB = 32 * NW;                           // queries per item
    constexpr int NP = 8 / NW;                            // K pieces (and V^T pieces) per wave and tile: 2 or 1
    constexpr int VM = 2 * NP + 1;                        // VMEM operations per wave and tile
    static_assert((NW == 4 || NW == 8) && NS >= 3 && NS <= 6, "geometry");
    extern __shared__ __attribute__((aligned(16))) char smem[];

    const int tid = threadIdx.x, lane = tid & 63, half = lane >> 5;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int N = p.N, Npad = p.Npad, H = p.H;
    const int nt = (N + 63) >> 6;                         // >= 2 (the launcher sends shorter sequences elsewhere)
    const int nqb = (N + QB - 1) / QB;
    const int total = nqb * H * p.B;
    // item list of this workgroup: workgroup w runs on XCD w % 8 (speed only); an XCD owns a contiguous run of the head-major
    // item order and its workgroups take the run's items round-robin, so the items in flight on an XCD are neighbours
    const int xcd = blockIdx.x & 7, widx = blockIdx.x >> 3, wpx = gridDim.x >> 3;
    const int per = (total + 7) >> 3;
    const int run0 = xcd * per, run_n = min(per, total - run0);
    const int n_my = widx < run_n ? (run_n - widx + wpx - 1) / wpx : 0;
    if (n_my <= 0) return;

    auto pin = [](const char* q) __attribute__((always_inline)) {
        const uint64_t u = reinterpret_cast<uint64_t>(q);
        const uint32_t lo = __builtin_amdgcn_readfirstlane((uint32_t)u), hi = __builtin_amdgcn_readfirstlane((uint32_t)(u >> 32));
        return reinterpret_cast<const char*>(((uint64_t)hi << 32) | lo);
    };
    struct Item { int qb, h, b; };
    auto decode = [&](int k) __attribute__((always_inline)) {
        const int L = run0 + widx + k * wpx;
        Item it;
        it.qb = L % nqb;
        const int r = L / nqb;
        it.h = r % H;
        it.b = r / H;
        return it;
    };

    // DMA plan: wave w fills K pieces w (+ 4) and V^T pieces w (+ 4) of a tile and its own copy of the key_add row
    uint32_t voff_k[NP], voff_v[NP];
#pragma unroll
    for (int i = 0; i < NP; ++i) {
        const int piece = wave + NW * i;
        const int row = 8 * piece + (lane >> 3);
        const int chunk = (lane & 7) ^ ((row >> 1) & 7);
        voff_k[i] = (uint32_t)(row * 128 + chunk * 16);
        voff_v[i] = (uint32_t)row * (uint32_t)(Npad * 2) + (uint32_t)(chunk * 16);
    }
    const uint32_t lds0 = (uint32_t)(uintptr_t)(lds_cptr)smem;
    const uint32_t lds_w = lds0 + wave * 1024;
    const uint32_t lds_a = lds0 + KADD0 + wave * 256;
    const uint32_t lane4 = lane * 4;
    // DMA cursor: next tile of the stream to request
    int dk = 0, dt = 0;
    const char *dKb, *dVb, *dAb;
    auto dma_item = [&](int k) __attribute__((always_inline)) {
        const Item it = decode(k);
        const size_t bh = (size_t)it.b * H + it.h;
        dKb = pin(reinterpret_cast<const char*>(p.k + bh * Npad * 64));
        dVb = pin(reinterpret_cast<const char*>(p.vt + bh * 64 * Npad));
        dAb = pin(reinterpret_cast<const char*>(p.key_add + (size_t)it.b * p.key_add_stride));
    };
    auto issue = [&](auto stc) __attribute__((always_inline)) {       // request stream tile (dk, dt) into stage ST, advance the cursor
        constexpr int ST = decltype(stc)::value;
        if (dk >= n_my) return;
        const char* kt = dKb + (size_t)dt * 8192;
        const char* vt = dVb + (size_t)dt * 128;
        const char* at = dAb + (size_t)dt * 256;
#pragma unroll
        for (int i = 0; i < NP; ++i) attn_dma16(voff_k[i], kt, lds_w + ST * STAGE + i * (NW * 1024));
#pragma unroll
        for (int i = 0; i < NP; ++i) attn_dma16(voff_v[i], vt, lds_w + ST * STAGE + 8192 + i * (NW * 1024));
        attn_dma4(lane4, at, lds_a + ST * (NW * 256));
        if (++dt == nt) { dt = 0; ++dk; if (dk < n_my) dma_item(dk); }
    };

    // lane-constant LDS addresses of the fragments: K chunk 2kk+half of row perm(lane & 31), V^T chunk 4jb+2t+half of row lane & 31
    const int m31 = lane & 31;
    const int kperm = (m31 & 0x13) | ((m31 & 4) << 1) | ((m31 & 8) >> 1);
    uint32_t kaddr[4], vaddr[2][2];
#pragma unroll
    for (int kk = 0; kk < 4; ++kk) kaddr[kk] = lds0 + swz128(kperm, 2 * kk + half);
#pragma unroll
    for (int jb = 0; jb < 2; ++jb)
#pragma unroll
        for (int t = 0; t < 2; ++t) vaddr[jb][t] = lds0 + 8192 + swz128(m31, 4 * jb + 2 * t + half);

    // compute cursor: item ck, tile ct; per-item state
    int ck = 0, ct = 0;
    bool active = false;
    int qrow = 0;
    size_t obase = 0;                                      // element offset of this lane's output row
    bf16x8 qf[4], qn[4];
    const float qs = 0.125f * ATTN_LOG2E;
    auto load_q = [&](int k, bf16x8 (&dst)[4]) __attribute__((always_inline)) {
        const Item it = decode(k);
        const bf16_t* Q = p.q + ((size_t)it.b * H + it.h) * Npad * 64;
        const int qr = (it.qb * NW + wave) * 32 + (lane & 31);
        const int qld = qr < N ? qr : N - 1;
#pragma unroll
        for (int kk = 0; kk < 4; ++kk) dst[kk] = *reinterpret_cast<const bf16x8*>(Q + (size_t)qld * 64 + (2 * kk + half) * 8);
    };
    auto begin_item = [&](int k) __attribute__((always_inline)) {
        const Item it = decode(k);
        const int q0 = (it.qb * NW + wave) * 32;
        active = q0 < N && !(p.ablate & 2);
        qrow = q0 + (lane & 31);
        obase = ((size_t)it.b * N + (qrow < N ? qrow : 0)) * (size_t)(H * 64) + (size_t)it.h * 64;
    };
    auto scale_q = [&](bf16x8 (&q)[4]) __attribute__((always_inline)) {
        if (!p.q_prescaled) {                             // test entry point: raw q, scaled (and rounded once more) here
#pragma unroll
            for (int kk = 0; kk < 4; ++kk)
#pragma unroll
                for (int e = 0; e < 8; ++e) q[kk][e] = f2bf(bf2f(q[kk][e]) * qs);
        }
    };

    f32x16 o[2], negm;
#pragma unroll
    for (int r = 0; r < 16; ++r) { o[0][r] = 0.f; o[1][r] = 0.f; negm[r] = 0.f; }
    float m_run = -INFINITY, l_run = 0.f;
    bf16x8 fr[8];                                          // the shared K / V^T fragment block
#ifdef UVL_ATTN_TRACE
    // tools/probes/attn_trace.hip: shader-clock stamps of one wave at the phase boundaries, 64 per register (one per lane)
    int tr_v[3] = {0, 0, 0}, tr_n = 0;
    const bool tr_on = p.trace && blockIdx.x == (unsigned)p.trace_block && wave == p.trace_wave;
#define ATTN_WL(dst, val, idx) dst = (lane == (idx)) ? (val) : dst
#define ATTN_STAMP()                                                                                          \
    if (tr_on && tr_n < 192) {                                                                               \
        const int tt_ = __builtin_amdgcn_readfirstlane((int)(uint32_t)__builtin_amdgcn_s_memtime());         \
        const int ix_ = __builtin_amdgcn_readfirstlane(tr_n & 63);                                           \
        if (tr_n < 64) { ATTN_WL(tr_v[0], tt_, ix_); }                                                       \
        else if (tr_n < 128) { ATTN_WL(tr_v[1], tt_, ix_); }                                                 \
        else { ATTN_WL(tr_v[2], tt_, ix_); }                                                                 \
        ++tr_n;                                                                                              \
    }
#else
#define ATTN_STAMP()
#endif

    auto read_k = [&](auto stc) __attribute__((always_inline)) {       // fr[4 jb + kk] = K fragment (key block jb, d step kk)
        constexpr int ST = decltype(stc)::value;
#pragma unroll
        for (int kk = 0; kk < 4; ++kk)
#pragma unroll
            for (int jb = 0; jb < 2; ++jb) fr[4 * jb + kk] = lds_read16(kaddr[kk] + ST * STAGE + jb * 4096);
    };
    auto read_v = [&](auto stc) __attribute__((always_inline)) {       // fr[4 jb + 2 t2 + db] = V^T fragment (d block db, keys 32 jb + 16 t2 ..)
        constexpr int ST = decltype(stc)::value;
#pragma unroll
        for (int jb = 0; jb < 2; ++jb)
#pragma unroll
            for (int t2 = 0; t2 < 2; ++t2)
#pragma unroll
                for (int db = 0; db < 2; ++db) fr[4 * jb + 2 * t2 + db] = lds_read16(vaddr[jb][t2] + ST * STAGE + db * 4096);
    };
    // zero the K rows / V^T columns beyond N of the tile in stage SN (this wave's own pieces, after its own vmcnt wait)
    auto tail_fix = [&](auto snc, int k1) __attribute__((always_inline)) {
        constexpr int SN = decltype(snc)::value;
#pragma unroll
        for (int i = 0; i < 2 * NP; ++i) {
            const bool isk = i < NP;
            const int piece = wave + NW * (i % NP);
            const int row = 8 * piece + (lane >> 3);
            char* at = smem + SN * STAGE + (isk ? 0 : 8192) + piece * 1024 + lane * 16;
            if (isk) {
                if (k1 + row >= N) *reinterpret_cast<u32x4*>(at) = u32x4{0u, 0u, 0u, 0u};
            } else {
                const int chunk = (lane & 7) ^ ((row >> 1) & 7);
                const int kb = k1 + chunk * 8;
                if (kb + 8 > N) {
                    u32x4 v = *reinterpret_cast<u32x4*>(at);
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        uint32_t wv = v[e];
                        if (kb + 2 * e >= N) wv &= 0xffff0000u;
                        if (kb + 2 * e + 1 >= N) wv &= 0x0000ffffu;
                        v[e] = wv;
                    }
                    *reinterpret_cast<u32x4*>(at) = v;
                }
            }
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    };

    auto tile = [&](auto stc) __attribute__((always_inline)) {
        constexpr int ST = decltype(stc)::value;
        const int t = ct;
        const int k0 = t * 64;
        f32x16 s[2];
        float psum = 0.f;
        ATTN_STAMP();                                      // 0: top of tile
        if (active) {
            const float* sA = reinterpret_cast<const float*>(smem + KADD0 + ST * (NW * 256) + wave * 256);
            const float ka_raw = sA[lane];
            const bool tail = k0 + 64 > N;                                          // wave-uniform
            const bool masked = __any((k0 + lane < N) ? (ka_raw != 0.f) : true);    // a mask term, or keys beyond N
            // C operand of the score MFMAs: (key_add * log2 e, -inf beyond N) - m; all registers equal -m on a plain tile
            auto c_operand = [&](float base, f32x16 (&c)[2]) __attribute__((always_inline)) {
#pragma unroll
                for (int jb = 0; jb < 2; ++jb)
#pragma unroll
                    for (int gq = 0; gq < 4; ++gq) {      // registers 4gq..4gq+3 = keys 32jb + 16(gq>>1) + 8 half + 4(gq&1) + 0..3
                        const int kq = 32 * jb + 16 * (gq >> 1) + 8 * half + 4 * (gq & 1);
                        const float4 a4 = *reinterpret_cast<const float4*>(sA + kq);
                        const float av[4] = {a4.x, a4.y, a4.z, a4.w};
#pragma unroll
                        for (int e = 0; e < 4; ++e) {
                            float v = fmaf(av[e], ATTN_LOG2E, base);
                            if (tail) v = (k0 + kq + e < N) ? v : -INFINITY;
                            c[jb][4 * gq + e] = v;
                        }
                    }
            };
            auto scores = [&](const f32x16& c0, const f32x16& c1) __attribute__((always_inline)) {
#pragma unroll
                for (int kk = 0; kk < 4; ++kk) {
                    s[0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fr[kk], qf[kk], kk == 0 ? c0 : s[0], 0, 0, 0);
                    s[1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fr[4 + kk], qf[kk], kk == 0 ? c1 : s[1], 0, 0, 0);
                }
            };
            auto exp_sum = [&]() __attribute__((always_inline)) {
                float ps[4] = {0.f, 0.f, 0.f, 0.f};       // four short add chains instead of one long one
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    s[0][r] = __builtin_amdgcn_exp2f(s[0][r]);
                    s[1][r] = __builtin_amdgcn_exp2f(s[1][r]);
                    ps[r & 1] += s[0][r];
                    ps[2 + (r & 1)] += s[1][r];
                }
                psum = (ps[0] + ps[1]) + (ps[2] + ps[3]);
            };
            bool redo = (t == 0);
            if (!redo) {
                // ---- A..C, speculative: s - m straight from the MFMA ----
                if (masked) {
                    c_operand(-m_run, s);
                    scores(s[0], s[1]);
                } else {
                    scores(negm, negm);
                }
                __builtin_amdgcn_sched_barrier(0);
                ATTN_STAMP();                             // 1: score MFMAs issued
                read_v(stc);
                __builtin_amdgcn_sched_barrier(0);
                exp_sum();
                redo = !__all(psum <= 256.0f);            // also catches inf / NaN
                ATTN_STAMP();                             // 2: exp / sums done
                if (redo) read_k(stc);                    // the exact path needs the K fragments again (LDS still holds the tile)
            }
            if (redo) {
                // ---- exact tile: scores with C = mask term only, true row max, rescale of l and O ----
                const f32x16 zero16 = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
                if (masked) {
                    c_operand(0.f, s);
                    scores(s[0], s[1]);
                } else {
                    scores(zero16, zero16);
                }
                __builtin_amdgcn_sched_barrier(0);
                read_v(stc);
                __builtin_amdgcn_sched_barrier(0);
                float tmax = -INFINITY;
#pragma unroll
                for (int r = 0; r < 16; ++r) tmax = fmaxf(tmax, fmaxf(s[0][r], s[1][r]));
                tmax = fmaxf(tmax, __shfl_xor(tmax, 32, 64));
                const float m_new = fmaxf(m_run, tmax);
                if (t != 0) {                             // first tile: l = 0, O = 0
                    const float alpha = __builtin_amdgcn_exp2f(m_run - m_new);
                    l_run *= alpha;
#pragma unroll
                    for (int r = 0; r < 16; ++r) { o[0][r] *= alpha; o[1][r] *= alpha; }
                }
                m_run = m_new;
#pragma unroll
                for (int r = 0; r < 16; ++r) { s[0][r] -= m_new; s[1][r] -= m_new; negm[r] = -m_new; }
                exp_sum();
            }
            l_run += psum;
        }
        // ---- D: the next tile of the stream is complete in LDS for everyone, the stage of the previous one is free ----
        // (t == 0: the previous item's output stores are in the queue behind the DMAs; stores and loads do not retire in order
        //  with each other, so no counted wait there)
        // (the counted form also needs the full complement of younger tiles in the queue: not once the stream's last tile has been
        //  requested, and not behind the q prefetch below, which is younger than the tiles)
        ATTN_STAMP();                                      // 3 (1 on an exact tile): before the DMA wait
        if (NS > 3 && t != 0 && t != nt - 1 && dk < n_my) attn_wait_vmcnt<(NS - 3) * VM>(); else attn_wait_vmcnt<0>();
        ATTN_STAMP();                                      // 4: DMA landed
        if (t + 1 == nt - 1 && (N & 63)) tail_fix(AttnIC<(ST + 1) % NS>{}, (t + 1) * 64);
        __builtin_amdgcn_s_barrier();
        ATTN_STAMP();                                      // 5: past the barrier
        if (!(p.ablate & 1)) issue(AttnIC<(ST + NS - 1) % NS>{});
        if (t == nt - 2 && ck + 1 < n_my) load_q(ck + 1, qn);          // next item's q: a whole tile to land before the next wait
        ATTN_STAMP();                                      // 6: DMA issued
        if (active) {
            // ---- E: O^T += V^T P^T ----
#pragma unroll
            for (int jb = 0; jb < 2; ++jb)
#pragma unroll
                for (int t2 = 0; t2 < 2; ++t2) {
                    union { uint32_t u[4]; bf16x8 v; } pf;
#pragma unroll
                    for (int e = 0; e < 4; ++e) pf.u[e] = pack_bf16x2(s[jb][8 * t2 + 2 * e], s[jb][8 * t2 + 2 * e + 1]);
#pragma unroll
                    for (int db = 0; db < 2; ++db) o[db] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fr[4 * jb + 2 * t2 + db], pf.v, o[db], 0, 0, 0);
                }
            __builtin_amdgcn_sched_barrier(0);
        }
        ATTN_STAMP();                                      // 7: output MFMAs issued
        // ---- end of an item: normalise, store, switch to the next item (its q fragments were requested at the top of this tile) ----
        const bool last_tile = (t == nt - 1);
        if (last_tile) {
            if (active && qrow < N) {
                const float l_tot = l_run + __shfl_xor(l_run, 32, 64);
                const float inv = 1.0f / l_tot;
                bf16_t* dst = p.o + obase;
#pragma unroll
                for (int db = 0; db < 2; ++db)
#pragma unroll
                    for (int gq = 0; gq < 4; ++gq) {
                        const int d0 = 32 * db + 8 * gq + 4 * half;
                        uint2 w;
                        w.x = pack_bf16x2(o[db][4 * gq + 0] * inv, o[db][4 * gq + 1] * inv);
                        w.y = pack_bf16x2(o[db][4 * gq + 2] * inv, o[db][4 * gq + 3] * inv);
                        *reinterpret_cast<uint2*>(dst + d0) = w;
                    }
            }
            ++ck;
            ct = 0;
            if (ck < n_my) {
                begin_item(ck);
#pragma unroll
                for (int kk = 0; kk < 4; ++kk) qf[kk] = qn[kk];
                scale_q(qf);
#pragma unroll
                for (int r = 0; r < 16; ++r) { o[0][r] = 0.f; o[1][r] = 0.f; }
                m_run = -INFINITY;
                l_run = 0.f;
            } else {
                active = false;
            }
        } else {
            ct = t + 1;
        }
        // ---- F: K fragments of the next tile of the stream ----
        if (active) read_k(AttnIC<(ST + 1) % NS>{});
    };

    // prologue: q of the first item, the first NS-1 tiles of the stream requested, tile 0 complete for everyone
    begin_item(0);
    load_q(0, qf);
    scale_q(qf);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    dma_item(0);
    issue(AttnIC<0>{});
    issue(AttnIC<1>{});
    if (NS > 3) issue(AttnIC<2 % NS>{});
    if (NS > 4) issue(AttnIC<3 % NS>{});
    if (NS > 5) issue(AttnIC<4 % NS>{});
    if (dk < n_my) attn_wait_vmcnt<(NS - 2) * VM>(); else attn_wait_vmcnt<0>();      // short stream: everything requested already
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    if (active) read_k(AttnIC<0>{});
    const int total_tiles = n_my * nt;
    for (int g = 0; g < total_tiles; g += NS) {
        tile(AttnIC<0>{});
        if (g + 1 < total_tiles) tile(AttnIC<1>{});
        if (g + 2 < total_tiles) tile(AttnIC<2>{});
        if (NS > 3 && g + 3 < total_tiles) tile(AttnIC<3 % NS>{});
        if (NS > 4 && g + 4 < total_tiles) tile(AttnIC<4 % NS>{});
        if (NS > 5 && g + 5 < total_tiles) tile(AttnIC<5 % NS>{});
    }
#ifdef UVL_ATTN_TRACE
    if (tr_on) { p.trace[lane] = tr_v[0]; p.trace[64 + lane] = tr_v[1]; p.trace[128 + lane] = tr_v[2]; p.trace[192] = tr_n; }
#endif
#undef ATTN_STAMP
}

template <int NW, int NS>
static hipError_t launch_attn_persist(const AttnParams& p_in, hipStream_t s) {
    constexpr size_t lds = (size_t)NS * 16384 + (size_t)NS * NW * 256;
    auto kern = attn_persist_kernel<NW, NS>;
    static bool attr_done = false;
    if (!attr_done) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) return e;
        attr_done = true;
    }
    static char name[40];
    if (!name[0]) snprintf(name, sizeof(name), "attn_persist_kernel<%d,%d>", NW, NS);
    g_last_kernel = name;
    const int total = ((p_in.N + 32 * NW - 1) / (32 * NW)) * p_in.H * p_in.B;
    AttnParams p = p_in;
    p.ablate = g_tune_attn_abl;
    const int slots = 256 * (NW == 4 ? 2 : 1);            // resident workgroups on the chip
    const int grid = total < slots ? 8 * ((total + 7) / 8) : slots;
    hipLaunchKernelGGL(kern, dim3(grid), dim3(64 * NW), lds, s, p);
    return hipGetLastError();
}

// ------------------------------------------------------------------------------------------------
// attn_pp_kernel<NS>: two persistent 4-wave groups per workgroup in PING-PONG.  Shader-clock trace of attn_persist_kernel
// (tools/probes/attn_trace.hip, profiles/r02_attention_pmc.md): per 64-key tile a wave needs ~512 cycles of the MFMA pipe and
// ~870 cycles of the VALU (32 quarter-rate v_exp_f32 = 512, the adds / packs / addressing the rest), and two co-resident waves
// in the same phase fight for the same pipe.  Here the 8 waves of a workgroup form two groups (waves 0-3 / 4-7: wave w and
// w + 4 share a SIMD); each group owns its own item list and K / V^T ring, and the WORKGROUP barrier is the phase clock:
//
//        slot 2i   :  group 0  VALU slot of its tile i       |  group 1  MFMA slot of its tile i-1
//        slot 2i+1 :  group 0  MFMA slot of its tile i       |  group 1  VALU slot of its tile i
//
//   VALU slot (tile t): request the V^T fragments of tile t and the K fragments of tile t+1 from LDS, turn the scores S(t)
//                       (they already hold s - m: the running maximum entered the score MFMA as its C operand) into bf16
//                       numerators with exp2, row sums, speculation check (redo on S itself: the scores are not overwritten)
//   MFMA slot (tile t): O += V^T P^T (8 MFMAs), item switch after an item's last tile (normalise, store, next q), S(t+1) =
//                       K Q^T - m (8 MFMAs), wait for this wave's share of tile t+2, request tile t+NS by DMA
// so on every SIMD one wave issues exponentials while its partner issues MFMAs.  The tile stream of a group never stops at an
// item boundary (DMA cursor NS-2 tiles ahead, next item's q requested one tile early, stores drain under the next tiles).
// A row whose keys are ALL masked with -1e10 is not supported (its sum is 0; cat_mask never masks the search tokens,
// extractor.py:43-50); rows masked entirely with BERT's -10000 are.
// ------------------------------------------------------------------------------------------------
// The lane index as a value hipcc cannot hoist: per-lane addresses and predicates of the RARE paths (tail fix-up, mask term, q request,
// output rows) are recomputed where they are used instead of living in registers across the whole tile loop (the kernel is at its
// 256-register budget; a spill reload is a scratch load, and hipcc's wait for it would drain the DMA ring).
__device__ __forceinline__ float attn_max3(float a, float b, float c) {      // one v_max3_f32 (fmaxf chains get a canonicalising v_max per MFMA output)
    float r;
    asm("v_max3_f32 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "v"(c));
    return r;
}
__device__ __forceinline__ int attn_lane_now() {
    int l = (int)(threadIdx.x & 63);
    asm volatile("" : "+v"(l));
    return l;
}

template <int NS>
__global__ __launch_bounds__(512, 1) void attn_pp_kernel(const AttnParams p) {
    constexpr int STAGE = 16384;                          // K tile 8 KB + V^T tile 8 KB
    constexpr int QOFF = NS * STAGE + NS * 4 * 256;       // per group: ring, [NS][4 waves][64] f32 key_add rows, [4 waves][32][64] bf16 q rows
    constexpr int WOFF = QOFF + 4 * 4096;                 // [4 waves][64] f32 landing pad of the L2 warm-up requests
    constexpr int GLDS = WOFF + 4 * 256;
    constexpr int VM = 5;                                 // VMEM operations per wave and tile: 2 K pieces, 2 V^T pieces, key_add row
    static_assert(NS == 3, "ring depth (2 x (NS x 17 KB + 16 KB) of LDS)");
    extern __shared__ __attribute__((aligned(16))) char smem[];

    const int tid = threadIdx.x, lane = tid & 63, half = lane >> 5;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int grp = wave >> 2, gw = wave & 3;
    const int N = p.N, Npad = p.Npad, H = p.H;
    const int nt = (N + 63) >> 6;                         // >= 2 (the launcher sends shorter sequences elsewhere)
    const int nqb = (N + 127) / 128;
    const int total = nqb * H * p.B;
    // items: an XCD owns a contiguous run of the head-major (sample, head, query block) order (workgroup w runs on XCD w % 8,
    // speed only); the run's items go round-robin to the 2 x (workgroups per XCD) groups, so items in flight are neighbours
    const int xcd = blockIdx.x & 7, widx = blockIdx.x >> 3, nworkers = 2 * (gridDim.x >> 3);
    const int per = (total + 7) >> 3;
    const int run0 = xcd * per, run_n = min(per, total - run0);
    const int wk = 2 * widx + grp;
    const int n_my = wk < run_n ? (run_n - wk + nworkers - 1) / nworkers : 0;
    const int n_g0 = 2 * widx < run_n ? (run_n - 2 * widx + nworkers - 1) / nworkers : 0;    // group 0 never has fewer items
    if (n_g0 <= 0) return;
    const int iters = n_g0 * nt;                          // slot pairs of this workgroup (both groups run the same barriers)

    auto pin = [](const char* q) __attribute__((always_inline)) {
        const uint64_t u = reinterpret_cast<uint64_t>(q);
        const uint32_t lo = __builtin_amdgcn_readfirstlane((uint32_t)u), hi = __builtin_amdgcn_readfirstlane((uint32_t)(u >> 32));
        return reinterpret_cast<const char*>(((uint64_t)hi << 32) | lo);
    };
    struct Item { int qb, h, b; };
    auto decode = [&](int k) __attribute__((always_inline)) {
        const int L = run0 + wk + k * nworkers;
        Item it;
        it.qb = L % nqb;
        const int r = L / nqb;
        it.h = r % H;
        it.b = r / H;
        return it;
    };

    char* gsm = smem + grp * GLDS;
    const uint32_t lds0 = (uint32_t)(uintptr_t)(lds_cptr)smem + grp * GLDS;
    // DMA plan: wave gw of the group fills K pieces gw, gw+4 and V^T pieces gw, gw+4 of a tile and its own copy of the key_add row
    uint32_t voff_k[2], voff_v[2];
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int piece = gw + 4 * i;
        const int row = 8 * piece + (lane >> 3);
        const int chunk = (lane & 7) ^ ((row >> 1) & 7);
        voff_k[i] = (uint32_t)(row * 128 + chunk * 16);
        voff_v[i] = (uint32_t)row * (uint32_t)(Npad * 2) + (uint32_t)(chunk * 16);
    }
    const uint32_t lds_w = lds0 + gw * 1024;
    const uint32_t lds_a = lds0 + NS * STAGE + gw * 256;
    const uint32_t lane4 = lane * 4;
    int dk = 0, dt = 0;                                    // DMA cursor: next tile of the group's stream to request
    const char *dKb = nullptr, *dVb = nullptr, *dAb = nullptr;
    auto dma_item = [&](int k) __attribute__((always_inline)) {
        const Item it = decode(k);
        const size_t bh = (size_t)it.b * H + it.h;
        dKb = pin(reinterpret_cast<const char*>(p.k + bh * Npad * 64));
        dVb = pin(reinterpret_cast<const char*>(p.vt + bh * 64 * Npad));
        dAb = pin(reinterpret_cast<const char*>(p.key_add + (size_t)it.b * p.key_add_stride));
    };
    // request stream tile (dk, dt) into stage ST in five parts (so that the MFMA slot can put one between its MFMAs: a DMA
    // instruction holds the wave's issue for ~75 cycles, two MFMAs keep the matrix pipe busy for 64), then advance the cursor
    auto issue_part = [&](auto stc, auto partc) __attribute__((always_inline)) {
        constexpr int ST = decltype(stc)::value;
        constexpr int PART = decltype(partc)::value;
        if (dk >= n_my || (p.ablate & 1)) return;
        if (PART == 0) attn_dma16(voff_k[0], dKb + (size_t)dt * 8192, lds_w + ST * STAGE);
        if (PART == 1) attn_dma16(voff_k[1], dKb + (size_t)dt * 8192, lds_w + ST * STAGE + 4096);
        if (PART == 2) attn_dma16(voff_v[0], dVb + (size_t)dt * 128, lds_w + ST * STAGE + 8192);
        if (PART == 3) attn_dma16(voff_v[1], dVb + (size_t)dt * 128, lds_w + ST * STAGE + 8192 + 4096);
        if (PART == 4) {
            attn_dma4(lane4, dAb + (size_t)dt * 256, lds_a + ST * 1024);
            if (++dt == nt) { dt = 0; ++dk; if (dk < n_my) dma_item(dk); }
        }
    };
    auto issue = [&](auto stc) __attribute__((always_inline)) {
        issue_part(stc, AttnIC<0>{});
        issue_part(stc, AttnIC<1>{});
        issue_part(stc, AttnIC<2>{});
        issue_part(stc, AttnIC<3>{});
        issue_part(stc, AttnIC<4>{});
    };

    // lane-constant LDS addresses of the fragments: K chunk 2kk+half of row perm(lane & 31), V^T chunk 4jb+2t+half of row lane & 31
    const int m31 = lane & 31;
    const int kperm = (m31 & 0x13) | ((m31 & 4) << 1) | ((m31 & 8) >> 1);
    uint32_t kaddr[4], vaddr[2][2];
#pragma unroll
    for (int kk = 0; kk < 4; ++kk) kaddr[kk] = lds0 + swz128(kperm, 2 * kk + half);
#pragma unroll
    for (int jb = 0; jb < 2; ++jb)
#pragma unroll
        for (int t = 0; t < 2; ++t) vaddr[jb][t] = lds0 + 8192 + swz128(m31, 4 * jb + 2 * t + half);

    // compute cursor and per-item state
    int ck = 0, ct = 0;
    bool active = false, active_n = false;                 // this wave has queries in the current / the next item
    int qrow = 0;
    uint32_t obase = 0;                                    // byte offset of this lane's output row (the output is < 4 GB)
    bf16x8 qf[4];
    const float qs = 0.125f * ATTN_LOG2E;
    auto wave_active = [&](int k) __attribute__((always_inline)) {
        const Item it = decode(k);
        return (it.qb * 4 + gw) * 32 < N && !(p.ablate & 2);
    };
    // q goes HBM -> LDS by DMA as well (this wave's own 32 rows, swizzled like a K tile) and is read back at the item switch:
    // an ordinary load inside the loop would make hipcc wait on vmcnt -- and with it on every DMA in flight -- at each use of qf
    const uint32_t lds_q = lds0 + QOFF + gw * 4096;
    auto request_q = [&](int k) __attribute__((always_inline)) {
        const Item it = decode(k);
        const char* qb_ = pin(reinterpret_cast<const char*>(p.q + ((size_t)it.b * H + it.h) * Npad * 64));
        const int q0 = (it.qb * 4 + gw) * 32;
        const int ln = attn_lane_now();
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int row = 8 * i + (ln >> 3);
            const int chunk = (ln & 7) ^ ((row >> 1) & 7);
            const int qr = q0 + row < N ? q0 + row : N - 1;          // rows beyond N repeat the last one (finite data, results discarded)
            attn_dma16((uint32_t)(qr * 128 + chunk * 16), qb_, lds_q + i * 1024);
        }
    };
    // Every group of the chip reaches its item boundary at about the same moment, and the first tiles of a new head come from
    // HBM: with a one-tile DMA lead the slots around the boundary ran 1.5-3.5x long (trace, profiles/r02_attention_pmc.md).  So a
    // whole item ahead, the K and V^T rows of the next item's head are pulled into L2 by sparse 4-byte DMA requests (one lane per
    // 128-byte line, 64 lines per instruction) that land on a scratch row nobody reads.
    auto warm_l2 = [&](int k) __attribute__((always_inline)) {
        const Item it = decode(k);
        const size_t bh = (size_t)it.b * H + it.h;
        const char* kb_ = pin(reinterpret_cast<const char*>(p.k + bh * Npad * 64));
        const char* vb_ = pin(reinterpret_cast<const char*>(p.vt + bh * 64 * Npad));
        const uint32_t bytes = (uint32_t)Npad * 128u;                 // both blocks are Npad * 128 bytes, contiguous
        const uint32_t ln = (uint32_t)attn_lane_now();
        for (uint32_t c = (uint32_t)gw * 8192u; c < bytes; c += 4u * 8192u) {
            const uint32_t off = c + ln * 128u < bytes ? c + ln * 128u : bytes - 128u;
            attn_dma4(off, kb_, lds0 + WOFF + gw * 256);
            attn_dma4(off, vb_, lds0 + WOFF + gw * 256);
        }
    };
    auto fetch_q = [&]() __attribute__((always_inline)) {            // after this wave's own vmcnt wait
        const int ln = attn_lane_now();
#pragma unroll
        for (int kk = 0; kk < 4; ++kk) qf[kk] = lds_read16(lds_q + swz128(ln & 31, 2 * kk + (ln >> 5)));
    };
    auto begin_item = [&](int k) __attribute__((always_inline)) {
        const Item it = decode(k);
        qrow = (it.qb * 4 + gw) * 32 + (attn_lane_now() & 31);
        obase = (((uint32_t)it.b * (uint32_t)N + (uint32_t)(qrow < N ? qrow : 0)) * (uint32_t)(H * 64) + (uint32_t)it.h * 64u) * 2u;
    };
    auto scale_q = [&](bf16x8 (&q)[4]) __attribute__((always_inline)) {
        if (!p.q_prescaled) {                             // test entry point: raw q, scaled (and rounded once more) here
#pragma unroll
            for (int kk = 0; kk < 4; ++kk)
#pragma unroll
                for (int e = 0; e < 8; ++e) q[kk][e] = f2bf(bf2f(q[kk][e]) * qs);
        }
    };

    f32x16 o[2], s[2];
#pragma unroll
    for (int r = 0; r < 16; ++r) { o[0][r] = 0.f; o[1][r] = 0.f; s[0][r] = 0.f; s[1][r] = 0.f; }
    float m_run = 0.f, l_run = 0.f;
    bf16x8 frK[8], frV[8];                                 // K fragments of the next tile, V^T fragments of the current one

    auto read_k = [&](auto stc) __attribute__((always_inline)) {       // frK[4 jb + kk] = K fragment (key block jb, d step kk)
        constexpr int ST = decltype(stc)::value;
#pragma unroll
        for (int kk = 0; kk < 4; ++kk)
#pragma unroll
            for (int jb = 0; jb < 2; ++jb) frK[4 * jb + kk] = lds_read16(kaddr[kk] + ST * STAGE + jb * 4096);
    };
    auto read_v = [&](auto stc) __attribute__((always_inline)) {       // frV[4 jb + 2 t2 + db] = V^T fragment (d block db, keys 32 jb + 16 t2 ..)
        constexpr int ST = decltype(stc)::value;
#pragma unroll
        for (int jb = 0; jb < 2; ++jb)
#pragma unroll
            for (int t2 = 0; t2 < 2; ++t2)
#pragma unroll
                for (int db = 0; db < 2; ++db) frV[4 * jb + 2 * t2 + db] = lds_read16(vaddr[jb][t2] + ST * STAGE + db * 4096);
    };
    // zero the K rows / V^T columns beyond N of the tile in stage SN (this wave's own pieces, after its own vmcnt wait) and put
    // -inf on the keys beyond N in this wave's copy of the tile's key_add row (whatever the caller's buffer holds there)
    auto tail_fix = [&](auto snc, int k1) __attribute__((always_inline)) {
        constexpr int SN = decltype(snc)::value;
        const int ln = attn_lane_now();
        if (k1 + ln >= N) *reinterpret_cast<float*>(gsm + NS * STAGE + SN * 1024 + gw * 256 + ln * 4) = -INFINITY;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const bool isk = i < 2;
            const int piece = gw + 4 * (i & 1);
            const int row = 8 * piece + (ln >> 3);
            char* at = gsm + SN * STAGE + (isk ? 0 : 8192) + piece * 1024 + ln * 16;
            if (isk) {
                if (k1 + row >= N) *reinterpret_cast<u32x4*>(at) = u32x4{0u, 0u, 0u, 0u};
            } else {
                const int chunk = (ln & 7) ^ ((row >> 1) & 7);
                const int kb = k1 + chunk * 8;
                if (kb + 8 > N) {
                    u32x4 v = *reinterpret_cast<u32x4*>(at);
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        uint32_t wv = v[e];
                        if (kb + 2 * e >= N) wv &= 0xffff0000u;
                        if (kb + 2 * e + 1 >= N) wv &= 0x0000ffffu;
                        v[e] = wv;
                    }
                    *reinterpret_cast<u32x4*>(at) = v;
                }
            }
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    };
    // S(tile tn in stage SN) = K Q^T + C with C[key] = key_add[key] * log2 e + base (the staged row holds -inf beyond N, see
    // tail_fix); base = -m, or 0 on the first tile of an item.  The C block is built here, in the MFMA slot, where the VALU is otherwise idle: one code path for plain and
    // masked tiles, no per-tile flag, and no 16-register -m block kept across the VALU slot.
    auto scores = [&](auto snc, float base) __attribute__((always_inline)) {
        constexpr int SN = decltype(snc)::value;
        const float* sA = reinterpret_cast<const float*>(gsm + NS * STAGE + SN * 1024 + gw * 256);
        const int hf = attn_lane_now() >> 5;
#pragma unroll
        for (int jb = 0; jb < 2; ++jb)
#pragma unroll
            for (int gq = 0; gq < 4; ++gq) {              // registers 4gq..4gq+3 = keys 32jb + 16(gq>>1) + 8 half + 4(gq&1) + 0..3
                const int kq = 32 * jb + 16 * (gq >> 1) + 8 * hf + 4 * (gq & 1);
                const float4 a4 = *reinterpret_cast<const float4*>(sA + kq);
                const float av[4] = {a4.x, a4.y, a4.z, a4.w};
#pragma unroll
                for (int e = 0; e < 4; ++e) s[jb][4 * gq + e] = fmaf(av[e], ATTN_LOG2E, base);
            }
#pragma unroll
        for (int kk = 0; kk < 4; ++kk) {
            s[0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(frK[kk], qf[kk], s[0], 0, 0, 0);
            s[1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(frK[4 + kk], qf[kk], s[1], 0, 0, 0);
        }
    };

#ifdef UVL_ATTN_TRACE
    int tr_v[3] = {0, 0, 0}, tr_n = 0;
    const bool tr_on = p.trace && blockIdx.x == (unsigned)p.trace_block && wave == p.trace_wave;
#define ATTN_WL(dst, val, idx) dst = (lane == (idx)) ? (val) : dst
#define ATTN_STAMP()                                                                                          \
    if (tr_on && tr_n < 192) {                                                                               \
        const int tt_ = __builtin_amdgcn_readfirstlane((int)(uint32_t)__builtin_amdgcn_s_memtime());         \
        const int ix_ = __builtin_amdgcn_readfirstlane(tr_n & 63);                                           \
        if (tr_n < 64) { ATTN_WL(tr_v[0], tt_, ix_); }                                                       \
        else if (tr_n < 128) { ATTN_WL(tr_v[1], tt_, ix_); }                                                 \
        else { ATTN_WL(tr_v[2], tt_, ix_); }                                                                 \
        ++tr_n;                                                                                              \
    }
#ifdef UVL_ATTN_TRACE_FINE
#define ATTN_STAMP2() ATTN_STAMP()
#else
#define ATTN_STAMP2()
#endif
#else
#define ATTN_STAMP()
#define ATTN_STAMP2()
#endif

    // ---- VALU slot of the group's current tile (stage ST) ----
    auto valu_slot = [&](auto stc) __attribute__((always_inline)) {
        constexpr int ST = decltype(stc)::value;
        if (ck >= n_my) return;
        const int t = ct;
        const bool has_next = !(t == nt - 1 && ck + 1 >= n_my);
        const bool act_next = has_next && (t == nt - 1 ? active_n : active);
        if (active) read_v(stc);
        if (!active && act_next) read_k(AttnIC<(ST + 1) % NS>{});
        ATTN_STAMP2();                                     // V1: fragment reads issued
        if (active) {
            // s holds score - m (m = 0 on the first tile of an item).  The maximum is only raised when some row of the wave grew by
            // more than 2^8 ("defer-max", decided BEFORE the exponentials: they are computed in place) -- and always on the first
            // tile.  Two complete versions of the slot body, chosen by one wave-uniform branch: a conditional patch-up of s and O
            // inside one body makes hipcc copy both register blocks at the join.
            float tmax = attn_max3(s[0][0], s[1][0], s[0][1]);
#pragma unroll
            for (int r = 1; r < 15; r += 2) tmax = attn_max3(tmax, s[0][r + 1], attn_max3(s[1][r], s[1][r + 1], s[0][r + 2 < 16 ? r + 2 : 15]));
            tmax = attn_max3(tmax, s[1][15], s[0][15]);
            // (a wave keeps at most ~8 ds_read_b128 in flight: the K fragments of the next tile are requested here, behind the
            //  maximum, when the V^T reads above have retired)
            if (act_next) read_k(AttnIC<(ST + 1) % NS>{});
            float ps[4] = {0.f, 0.f, 0.f, 0.f};           // row-sum partials in four short chains
            if (t == 0 || __any(tmax > ATTN_DEFER)) {
                tmax = fmaxf(tmax, __shfl_xor(tmax, 32, 64));
                // first tile: m := row maximum (0 for a row whose keys are all masked with -1e10: its numerators are exactly 0)
                const float delta = t == 0 ? (tmax < -1e9f ? 0.f : tmax) : fmaxf(tmax, 0.f);
                const float alpha = t == 0 ? 1.0f : __builtin_amdgcn_exp2f(-delta);         // l = 0, O = 0 on the first tile
                m_run += delta;
                l_run *= alpha;
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    o[0][r] *= alpha;
                    o[1][r] *= alpha;
                    s[0][r] = __builtin_amdgcn_exp2f(s[0][r] - delta);
                    s[1][r] = __builtin_amdgcn_exp2f(s[1][r] - delta);
                    ps[r & 1] += s[0][r];
                    ps[2 + (r & 1)] += s[1][r];
                }
            } else {
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    s[0][r] = __builtin_amdgcn_exp2f(s[0][r]);
                    s[1][r] = __builtin_amdgcn_exp2f(s[1][r]);
                    ps[r & 1] += s[0][r];
                    ps[2 + (r & 1)] += s[1][r];
                }
            }
            l_run += (ps[0] + ps[1]) + (ps[2] + ps[3]);
        }
        ATTN_STAMP2();                                     // V2: numerators done
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // every LDS read of this slot has landed: after the barrier the stage of tile t may be refilled
    };

    // ---- MFMA slot of the group's current tile (stage ST) ----
    auto mfma_slot = [&](auto stc) __attribute__((always_inline)) {
        constexpr int ST = decltype(stc)::value;
        if (ck >= n_my) return;
        const int t = ct;
        const bool last = (t == nt - 1);
        const bool has_next = !(last && ck + 1 >= n_my);
        // this wave's share of stream tile +2 has landed -- it was requested one tile ago, in the previous MFMA slot -- and so has
        // everything older: the next item's q rows (requested a whole item ago) and the previous item's output stores (issued at the
        // END of their slot, so that this wait does not sit right behind them)
        attn_wait_vmcnt<0>();
        {
            const int t2i = t + 2 < nt ? t + 2 : t + 2 - nt;          // is that tile the tail tile of its item?
            if (t2i == nt - 1 && (N & 63)) tail_fix(AttnIC<(ST + 2) % NS>{}, t2i * 64);
        }
        ATTN_STAMP2();                                     // M1: DMA wait done
        // O^T += V^T P^T, and between the MFMA pairs the request of stream tile +3 into the stage of tile t (free: its K and V^T
        // fragments are in registers everywhere)
        auto pv_step = [&](auto jbc, auto t2c) __attribute__((always_inline)) {
            constexpr int jb = decltype(jbc)::value, t2 = decltype(t2c)::value;
            if (active) {
                union { uint32_t u[4]; bf16x8 v; } pf;      // bf16 packing here, where the VALU is idle
#pragma unroll
                for (int e = 0; e < 4; ++e) pf.u[e] = pack_bf16x2(s[jb][8 * t2 + 2 * e], s[jb][8 * t2 + 2 * e + 1]);
#pragma unroll
                for (int db = 0; db < 2; ++db)
                    o[db] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(frV[4 * jb + 2 * t2 + db], pf.v, o[db], 0, 0, 0);
            }
        };
        pv_step(AttnIC<0>{}, AttnIC<0>{});
        issue_part(stc, AttnIC<0>{});
        pv_step(AttnIC<0>{}, AttnIC<1>{});
        issue_part(stc, AttnIC<1>{});
        pv_step(AttnIC<1>{}, AttnIC<0>{});
        issue_part(stc, AttnIC<2>{});
        pv_step(AttnIC<1>{}, AttnIC<1>{});
        issue_part(stc, AttnIC<3>{});
        issue_part(stc, AttnIC<4>{});
        ATTN_STAMP2();                                     // M2/M3: P V issued, DMA requested
        const bool store_now = last && active;
        int tn = t + 1;
        if (last) {
            // ---- item switch, part 1: the next item's q (its rows are in LDS), so that its first scores can be issued at once ----
            ++ck;
            ct = 0;
            tn = 0;
            if (ck < n_my) {
                active = active_n;
                fetch_q();
                scale_q(qf);
            } else {
                active = false;
            }
        } else {
            ct = t + 1;
        }
        if (has_next && active) scores(AttnIC<(ST + 1) % NS>{}, tn == 0 ? 0.f : -m_run);
        ATTN_STAMP2();                                     // M4: scores issued
        if (last) {
            // ---- item switch, part 2: normalise and store the finished item, reset the accumulators, look one item ahead ----
            if (store_now && qrow < N) {
                const float l_tot = l_run + __shfl_xor(l_run, 32, 64);
                const float inv = 1.0f / l_tot;
                bf16_t* dst = reinterpret_cast<bf16_t*>(reinterpret_cast<char*>(p.o) + obase);
#pragma unroll
                for (int db = 0; db < 2; ++db)
#pragma unroll
                    for (int gq = 0; gq < 4; ++gq) {
                        const int d0 = 32 * db + 8 * gq + 4 * (attn_lane_now() >> 5);
                        uint2 w;
                        w.x = pack_bf16x2(o[db][4 * gq + 0] * inv, o[db][4 * gq + 1] * inv);
                        w.y = pack_bf16x2(o[db][4 * gq + 2] * inv, o[db][4 * gq + 3] * inv);
                        *reinterpret_cast<uint2*>(dst + d0) = w;
                    }
            }
#pragma unroll
            for (int r = 0; r < 16; ++r) { o[0][r] = 0.f; o[1][r] = 0.f; }
            m_run = 0.f;
            l_run = 0.f;
            if (ck < n_my) begin_item(ck);                 // output rows of the item that starts now
            if (ck + 1 < n_my) {
                request_q(ck + 1);                         // a whole item to land (this wave's q rows in LDS have just been consumed)
                active_n = wave_active(ck + 1);
                warm_l2(ck + 1);
            }
        }
    };

    // ---- prologue: q of the first item, the ring filled (NS tiles), tiles 0 and 1 complete, S(0) computed ----
    if (n_my > 0) {
        begin_item(0);
        active = wave_active(0);
        request_q(0);
        dma_item(0);
    }
    issue(AttnIC<0>{});
    issue(AttnIC<1>{});
    issue(AttnIC<2>{});
    if (NS > 3) issue(AttnIC<3 % NS>{});
    attn_wait_vmcnt<0>();
    if (n_my > 0) { fetch_q(); scale_q(qf); }
    if (n_my > 0 && nt == 2 && (N & 63)) tail_fix(AttnIC<1>{}, 64);
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    if (n_my > 1) { request_q(1); active_n = wave_active(1); warm_l2(1); }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    if (n_my > 0 && active) {
        read_k(AttnIC<0>{});
        scores(AttnIC<0>{}, 0.f);
    }
    // ---- the slots: group 0 opens with its VALU slot, group 1 one slot later ----
    if (grp == 1) __builtin_amdgcn_s_barrier();
    for (int g = 0; g < iters; g += NS) {
        valu_slot(AttnIC<0>{});
        __builtin_amdgcn_s_barrier();
        ATTN_STAMP();
        mfma_slot(AttnIC<0>{});
        __builtin_amdgcn_s_barrier();
        ATTN_STAMP();
        if (g + 1 < iters) {
            valu_slot(AttnIC<1>{});
            __builtin_amdgcn_s_barrier();
            ATTN_STAMP();
            mfma_slot(AttnIC<1>{});
            __builtin_amdgcn_s_barrier();
            ATTN_STAMP();
        }
        if (g + 2 < iters) {
            valu_slot(AttnIC<2>{});
            __builtin_amdgcn_s_barrier();
            ATTN_STAMP();
            mfma_slot(AttnIC<2>{});
            __builtin_amdgcn_s_barrier();
            ATTN_STAMP();
        }
        if (NS > 3 && g + 3 < iters) {
            valu_slot(AttnIC<3 % NS>{});
            __builtin_amdgcn_s_barrier();
            ATTN_STAMP();
            mfma_slot(AttnIC<3 % NS>{});
            __builtin_amdgcn_s_barrier();
            ATTN_STAMP();
        }
    }
    if (grp == 0) __builtin_amdgcn_s_barrier();
#ifdef UVL_ATTN_TRACE
    if (tr_on) { p.trace[lane] = tr_v[0]; p.trace[64 + lane] = tr_v[1]; p.trace[128 + lane] = tr_v[2]; p.trace[192] = tr_n; }
#endif
#undef ATTN_STAMP
#undef ATTN_STAMP2
}

template <int NS>
static hipError_t launch_attn_pp(const AttnParams& p_in, hipStream_t s) {
    constexpr size_t lds = 2 * ((size_t)NS * 16384 + (size_t)NS * 4 * 256 + 4 * 4096 + 4 * 256);
    auto kern = attn_pp_kernel<NS>;
    static bool attr_done = false;
    if (!attr_done) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) return e;
        attr_done = true;
    }
    static char name[40];
    if (!name[0]) snprintf(name, sizeof(name), "attn_pp_kernel<%d>", NS);
    g_last_kernel = name;
    const int total = ((p_in.N + 127) / 128) * p_in.H * p_in.B;
    AttnParams p = p_in;
    p.ablate = g_tune_attn_abl;
    const int pairs = (total + 1) / 2;                    // two groups per workgroup
    const int grid = pairs < 256 ? 8 * ((pairs + 7) / 8) : 256;
    hipLaunchKernelGGL(kern, dim3(grid), dim3(512), lds, s, p);
    return hipGetLastError();
}

// ------------------------------------------------------------------------------------------------
// attn_swp_kernel: software-pipelined across key tiles INSIDE a wave.  Measured on the variants above (trace probe,
// profiles/r02_attention_pmc.md): per 64-key tile a wave needs ~512 cycles of the MFMA pipe and ~870 cycles of VALU issue
// (32 quarter-rate v_exp_f32), and two waves of a SIMD do NOT hide each other's phases -- an 8-MFMA group takes 420-520 cycles
// beside a partner's exponentials, a tile costs VALU + MFMA.  What does overlap is ONE wave's own instruction stream: a VALU
// instruction issued between two MFMAs executes while the matrix pipe works.  So the steady-state iteration t of a wave issues
//     8 MFMAs  O += V^T(t-1) P^T(t-1)        (numerators of the PREVIOUS tile, packed)
//     8 MFMAs  S(t+1) = K(t+1) Q^T - m        (scores of the NEXT tile, second score block)
// interleaved one by one with the softmax arithmetic of tile t (exp2 in place, row sums, bf16 packing).  The score blocks and the
// K / V^T fragment blocks are double-buffered in registers across iterations (K fragments of tile t+2 and V^T fragments of tile t
// are requested at the end of iteration t, behind the tile barrier), the ring has 4 stages so that the loop unrolls by 4 with
// compile-time stage offsets and score-block names.  Iterations that cannot run the interleaved form -- first and last tile, a next
// tile that carries a mask term, a row maximum that grew by more than 2^8 -- run the same work sequentially (exact rescale there).
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256, 2) void attn_swp_kernel(const AttnParams p) {
    constexpr int NS = 4;
    constexpr int STAGE = 16384;                          // K tile 8 KB + V^T tile 8 KB
    constexpr int KADD0 = NS * STAGE;                     // [NS][4 waves][64] f32 key_add rows
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int nqb = (p.N + 127) / 128;
    int qb, h, b;
    if (!attn_decode_block((int)blockIdx.x, nqb * p.H * p.B, nqb, p.H, p.xcd_map != 0, qb, h, b)) return;

    const int tid = threadIdx.x, lane = tid & 63, half = lane >> 5;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int N = p.N, Npad = p.Npad;
    const size_t bh = (size_t)b * p.H + h;
    const bf16_t* __restrict__ Q = p.q + bh * Npad * 64;
    const int nt = (N + 63) >> 6;
    const int q0 = (qb * 4 + wave) * 32;
    const bool active = q0 < N;                           // wave-uniform
    const int qrow = q0 + (lane & 31);
    const int qld = qrow < N ? qrow : N - 1;
    bf16x8 qf[4];
#pragma unroll
    for (int kk = 0; kk < 4; ++kk) qf[kk] = *reinterpret_cast<const bf16x8*>(Q + (size_t)qld * 64 + (2 * kk + half) * 8);
    if (!p.q_prescaled) {                                 // test entry point: raw q, scaled (and rounded once more) here
#pragma unroll
        for (int kk = 0; kk < 4; ++kk)
#pragma unroll
            for (int e = 0; e < 8; ++e) qf[kk][e] = f2bf(bf2f(qf[kk][e]) * (0.125f * ATTN_LOG2E));
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // q is in registers before the first DMA: the loop's vmcnt counts only DMAs

    uint32_t voff[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int piece = wave + 4 * (i & 1);
        const int row = 8 * piece + (lane >> 3);
        const int chunk = (lane & 7) ^ ((row >> 1) & 7);
        voff[i] = i < 2 ? (uint32_t)(row * 128 + chunk * 16) : (uint32_t)row * (uint32_t)(Npad * 2) + (uint32_t)(chunk * 16);
    }
    auto pin = [](const char* q) __attribute__((always_inline)) {
        const uint64_t u = reinterpret_cast<uint64_t>(q);
        const uint32_t lo = __builtin_amdgcn_readfirstlane((uint32_t)u), hi = __builtin_amdgcn_readfirstlane((uint32_t)(u >> 32));
        return reinterpret_cast<const char*>(((uint64_t)hi << 32) | lo);
    };
    const char* Kb = pin(reinterpret_cast<const char*>(p.k + bh * Npad * 64));
    const char* Vb = pin(reinterpret_cast<const char*>(p.vt + bh * 64 * Npad));
    const char* Ab = pin(reinterpret_cast<const char*>(p.key_add + (size_t)b * p.key_add_stride));
    const uint32_t lds0 = (uint32_t)(uintptr_t)(lds_cptr)smem;
    const uint32_t lds_w = lds0 + wave * 1024;
    const uint32_t lds_a = lds0 + KADD0 + wave * 256;
    const uint32_t lane4 = lane * 4;
    auto issue = [&](int t, auto stc) __attribute__((always_inline)) {
        constexpr int ST = decltype(stc)::value;
        const char* kt = Kb + (size_t)t * 8192;
        const char* vt = Vb + (size_t)t * 128;
        const char* at = Ab + (size_t)t * 256;
        attn_dma16(voff[0], kt, lds_w + ST * STAGE);
        attn_dma16(voff[1], kt, lds_w + ST * STAGE + 4096);
        attn_dma16(voff[2], vt, lds_w + ST * STAGE + 8192);
        attn_dma16(voff[3], vt, lds_w + ST * STAGE + 8192 + 4096);
        attn_dma4(lane4, at, lds_a + ST * 1024);
    };
    const int m31 = lane & 31;
    const int kperm = (m31 & 0x13) | ((m31 & 4) << 1) | ((m31 & 8) >> 1);
    uint32_t kaddr[4], vaddr[2][2];
#pragma unroll
    for (int kk = 0; kk < 4; ++kk) kaddr[kk] = lds0 + swz128(kperm, 2 * kk + half);
#pragma unroll
    for (int jb = 0; jb < 2; ++jb)
#pragma unroll
        for (int t = 0; t < 2; ++t) vaddr[jb][t] = lds0 + 8192 + swz128(m31, 4 * jb + 2 * t + half);

    f32x16 o[2], sc[2][2];                                 // sc[tile & 1][key block]
#pragma unroll
    for (int r = 0; r < 16; ++r) { o[0][r] = 0.f; o[1][r] = 0.f; }
    float m_run = 0.f, l_run = 0.f;
    bf16x8 frK[8], frV[8];                                 // K fragments of tile t+1, V^T fragments of tile t-1 (at the top of iteration t)
    union PF { uint32_t u[4]; bf16x8 v; } pf[4];           // packed numerators of tile t-1: fragment (jb, t2) = pf[2 jb + t2]

    auto read_k = [&](auto stc) __attribute__((always_inline)) {
        constexpr int ST = decltype(stc)::value;
#pragma unroll
        for (int kk = 0; kk < 4; ++kk)
#pragma unroll
            for (int jb = 0; jb < 2; ++jb) frK[4 * jb + kk] = lds_read16(kaddr[kk] + ST * STAGE + jb * 4096);
    };
    auto read_v = [&](auto stc) __attribute__((always_inline)) {       // frV[2 f + db], f = 2 jb + t2
        constexpr int ST = decltype(stc)::value;
#pragma unroll
        for (int jb = 0; jb < 2; ++jb)
#pragma unroll
            for (int t2 = 0; t2 < 2; ++t2)
#pragma unroll
                for (int db = 0; db < 2; ++db) frV[4 * jb + 2 * t2 + db] = lds_read16(vaddr[jb][t2] + ST * STAGE + db * 4096);
    };
    auto tail_fix = [&](auto snc, int k1) __attribute__((always_inline)) {
        constexpr int SN = decltype(snc)::value;
        const int ln = attn_lane_now();
        if (k1 + ln >= N) *reinterpret_cast<float*>(smem + KADD0 + SN * 1024 + wave * 256 + ln * 4) = -INFINITY;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const bool isk = i < 2;
            const int piece = wave + 4 * (i & 1);
            const int row = 8 * piece + (ln >> 3);
            char* at = smem + SN * STAGE + (isk ? 0 : 8192) + piece * 1024 + ln * 16;
            if (isk) {
                if (k1 + row >= N) *reinterpret_cast<u32x4*>(at) = u32x4{0u, 0u, 0u, 0u};
            } else {
                const int chunk = (ln & 7) ^ ((row >> 1) & 7);
                const int kb = k1 + chunk * 8;
                if (kb + 8 > N) {
                    u32x4 v = *reinterpret_cast<u32x4*>(at);
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        uint32_t wv = v[e];
                        if (kb + 2 * e >= N) wv &= 0xffff0000u;
                        if (kb + 2 * e + 1 >= N) wv &= 0x0000ffffu;
                        v[e] = wv;
                    }
                    *reinterpret_cast<u32x4*>(at) = v;
                }
            }
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    };
    auto mask_flag = [&](auto snc, int tn) __attribute__((always_inline)) {
        constexpr int SN = decltype(snc)::value;
        const int ln = attn_lane_now();
        const float ka = *reinterpret_cast<const float*>(smem + KADD0 + SN * 1024 + wave * 256 + ln * 4);
        return (bool)__any((tn * 64 + ln < N) ? (ka != 0.f) : true);
    };
    // scores of the tile in stage SN into sc[bi]: K fragments in frK; C = base (+ key_add * log2 e on a masked tile, -inf beyond N)
    auto scores_seq = [&](auto snc, auto bic, bool masked, float base) __attribute__((always_inline)) {
        constexpr int SN = decltype(snc)::value, BI = decltype(bic)::value;
        if (masked) {
            const float* sA = reinterpret_cast<const float*>(smem + KADD0 + SN * 1024 + wave * 256);
            const int hf = attn_lane_now() >> 5;
#pragma unroll
            for (int jb = 0; jb < 2; ++jb)
#pragma unroll
                for (int gq = 0; gq < 4; ++gq) {
                    const int kq = 32 * jb + 16 * (gq >> 1) + 8 * hf + 4 * (gq & 1);
                    const float4 a4 = *reinterpret_cast<const float4*>(sA + kq);
                    const float av[4] = {a4.x, a4.y, a4.z, a4.w};
#pragma unroll
                    for (int e = 0; e < 4; ++e) sc[BI][jb][4 * gq + e] = fmaf(av[e], ATTN_LOG2E, base);
                }
        } else {
#pragma unroll
            for (int r = 0; r < 16; ++r) { sc[BI][0][r] = base; sc[BI][1][r] = base; }
        }
#pragma unroll
        for (int kk = 0; kk < 4; ++kk) {
            sc[BI][0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(frK[kk], qf[kk], sc[BI][0], 0, 0, 0);
            sc[BI][1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(frK[4 + kk], qf[kk], sc[BI][1], 0, 0, 0);
        }
    };
    auto pv_seq = [&]() __attribute__((always_inline)) {
#pragma unroll
        for (int f = 0; f < 4; ++f)
#pragma unroll
            for (int db = 0; db < 2; ++db) o[db] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(frV[2 * f + db], pf[f].v, o[db], 0, 0, 0);
    };

    // ---- one iteration: softmax of tile t (stage ST), O += V^T P^T of tile t-1, scores of tile t+1 ----
    auto iter = [&](const int t, auto stc) __attribute__((always_inline)) {
        constexpr int ST = decltype(stc)::value;
        constexpr int CUR = ST & 1, NXT = CUR ^ 1;
        const bool has_next = t + 1 < nt;
        // ---- tile t+1 complete in LDS for everyone (requested one iteration ago); the stage of tile t-2 is free: request tile t+2 ----
        attn_wait_vmcnt<0>();
        if (t + 1 == nt - 1 && t + 1 >= 2 && (N & 63)) tail_fix(AttnIC<(ST + 1) % NS>{}, (t + 1) * 64);
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        if (t + 2 < nt && !(p.ablate & 1)) issue(t + 2, AttnIC<(ST + 2) % NS>{});
        if (active) {
            // the fragments of this iteration (they live in registers only inside it): V^T(t-1) for the output MFMAs, K(t+1) for the scores
            if (t >= 1) read_v(AttnIC<(ST + NS - 1) % NS>{});
            if (has_next) read_k(AttnIC<(ST + 1) % NS>{});
            const bool masked_next = has_next && mask_flag(AttnIC<(ST + 1) % NS>{}, t + 1);
            float tmax = attn_max3(sc[CUR][0][0], sc[CUR][1][0], sc[CUR][0][1]);
#pragma unroll
            for (int r = 1; r < 15; r += 2) tmax = attn_max3(tmax, sc[CUR][0][r + 1], attn_max3(sc[CUR][1][r], sc[CUR][1][r + 1], sc[CUR][0][r + 2 < 16 ? r + 2 : 15]));
            tmax = attn_max3(tmax, sc[CUR][1][15], sc[CUR][0][15]);
            const bool fast = t >= 1 && has_next && !masked_next && !__any(tmax > ATTN_DEFER);
            float ps[4] = {0.f, 0.f, 0.f, 0.f};
            if (fast) {
                // ---- interleaved: MFMA slot i, then the exp2 / add (/ pack) of one score pair ----
                auto slot = [&](auto ic) __attribute__((always_inline)) {
                    constexpr int I = decltype(ic)::value;
                    if constexpr (I < 8) {                 // O += V^T(t-1) P^T(t-1): fragment f = I >> 1, d block I & 1
                        o[I & 1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(frV[I], pf[I >> 1].v, o[I & 1], 0, 0, 0);
                    } else {                               // S(t+1): d step (I - 8) >> 1, key block (I - 8) & 1
                        // the C operand -m is written into the first block only (16 v_mov, in slot 7); the second key block's chain
                        // opens first and reads it from there
                        constexpr int kk = (I - 8) >> 1, jb = ((I - 8) & 1) ^ 1;
                        if constexpr (kk == 0 && jb == 1) sc[NXT][1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(frK[4 + kk], qf[kk], sc[NXT][0], 0, 0, 0);
                        else sc[NXT][jb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(frK[4 * jb + kk], qf[kk], sc[NXT][jb], 0, 0, 0);
                    }
                    if constexpr (I == 7) {
#pragma unroll
                        for (int r = 0; r < 16; ++r) sc[NXT][0][r] = -m_run;
                    }
                    constexpr int jbe = I >> 3, r0 = 2 * (I & 7);   // the score pair of this slot
                    sc[CUR][jbe][r0] = __builtin_amdgcn_exp2f(sc[CUR][jbe][r0]);
                    sc[CUR][jbe][r0 + 1] = __builtin_amdgcn_exp2f(sc[CUR][jbe][r0 + 1]);
                    ps[2 * jbe] += sc[CUR][jbe][r0];
                    ps[2 * jbe + 1] += sc[CUR][jbe][r0 + 1];
                    if constexpr (I >= 8 && I < 15) {      // packing trails the output MFMAs (slots 0-7 read the previous tile's pf): pairs 2(I-8), 2(I-8)+1
#pragma unroll
                        for (int q = 2 * (I - 8); q < 2 * (I - 8) + 2; ++q) {
                            const int jq = q >> 3, rq = 2 * (q & 7);
                            pf[2 * jq + (rq >> 3)].u[(rq & 7) >> 1] = pack_bf16x2(sc[CUR][jq][rq], sc[CUR][jq][rq + 1]);
                        }
                    }
                };
                slot(AttnIC<0>{}); slot(AttnIC<1>{}); slot(AttnIC<2>{}); slot(AttnIC<3>{});
                slot(AttnIC<4>{}); slot(AttnIC<5>{}); slot(AttnIC<6>{}); slot(AttnIC<7>{});
                slot(AttnIC<8>{}); slot(AttnIC<9>{}); slot(AttnIC<10>{}); slot(AttnIC<11>{});
                slot(AttnIC<12>{}); slot(AttnIC<13>{}); slot(AttnIC<14>{}); slot(AttnIC<15>{});
#pragma unroll
                for (int q = 14; q < 16; ++q) {            // the last two pairs (exponentiated in slots 14, 15)
                    const int jq = q >> 3, rq = 2 * (q & 7);
                    pf[2 * jq + (rq >> 3)].u[(rq & 7) >> 1] = pack_bf16x2(sc[CUR][jq][rq], sc[CUR][jq][rq + 1]);
                }
            } else {
                // ---- sequential form: previous tile's P V first (O complete before a rescale), exact maximum, then the next scores ----
                if (t >= 1) pv_seq();
                tmax = fmaxf(tmax, __shfl_xor(tmax, 32, 64));
                const float delta = t == 0 ? (tmax < -1e9f ? 0.f : tmax) : fmaxf(tmax, 0.f);    // first tile: m := row maximum
                const float alpha = t == 0 ? 1.0f : __builtin_amdgcn_exp2f(-delta);
                m_run += delta;
                l_run *= alpha;
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    o[0][r] *= alpha;
                    o[1][r] *= alpha;
                    sc[CUR][0][r] = __builtin_amdgcn_exp2f(sc[CUR][0][r] - delta);
                    sc[CUR][1][r] = __builtin_amdgcn_exp2f(sc[CUR][1][r] - delta);
                    ps[r & 1] += sc[CUR][0][r];
                    ps[2 + (r & 1)] += sc[CUR][1][r];
                }
#pragma unroll
                for (int q = 0; q < 16; ++q) {
                    const int jq = q >> 3, rq = 2 * (q & 7);
                    pf[2 * jq + (rq >> 3)].u[(rq & 7) >> 1] = pack_bf16x2(sc[CUR][jq][rq], sc[CUR][jq][rq + 1]);
                }
                if (has_next) scores_seq(AttnIC<(ST + 1) % NS>{}, AttnIC<NXT>{}, masked_next, -m_run);
            }
            l_run += (ps[0] + ps[1]) + (ps[2] + ps[3]);
        }
    };

    // ---- prologue: tiles 0 and 1 requested and complete, S(0) (raw scores) computed ----
    issue(0, AttnIC<0>{});
    if (nt > 1) issue(1, AttnIC<1>{});
    attn_wait_vmcnt<0>();
    if (nt == 1 && (N & 63)) tail_fix(AttnIC<0>{}, 0);
    if (nt == 2 && (N & 63)) tail_fix(AttnIC<1>{}, 64);
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    if (active) {
        read_k(AttnIC<0>{});
        scores_seq(AttnIC<0>{}, AttnIC<0>{}, mask_flag(AttnIC<0>{}, 0), 0.f);
    }
    for (int t0 = 0; t0 < nt; t0 += NS) {
        iter(t0, AttnIC<0>{});
        if (t0 + 1 < nt) iter(t0 + 1, AttnIC<1>{});
        if (t0 + 2 < nt) iter(t0 + 2, AttnIC<2>{});
        if (t0 + 3 < nt) iter(t0 + 3, AttnIC<3>{});
    }
    if (active) {                                          // the last tile's numerators (its stage is still intact)
        switch ((nt - 1) % NS) {
            case 0: read_v(AttnIC<0>{}); break;
            case 1: read_v(AttnIC<1>{}); break;
            case 2: read_v(AttnIC<2>{}); break;
            default: read_v(AttnIC<3>{}); break;
        }
        pv_seq();
    }

    if (qrow < N) {
        const float l_tot = l_run + __shfl_xor(l_run, 32, 64);
        const float inv = 1.0f / l_tot;
        bf16_t* dst = p.o + ((size_t)b * N + qrow) * (p.H * 64) + h * 64;
#pragma unroll
        for (int db = 0; db < 2; ++db)
#pragma unroll
            for (int gq = 0; gq < 4; ++gq) {
                const int d0 = 32 * db + 8 * gq + 4 * half;
                uint2 w;
                w.x = pack_bf16x2(o[db][4 * gq + 0] * inv, o[db][4 * gq + 1] * inv);
                w.y = pack_bf16x2(o[db][4 * gq + 2] * inv, o[db][4 * gq + 3] * inv);
                *reinterpret_cast<uint2*>(dst + d0) = w;
            }
    }
}

static hipError_t launch_attn_swp(const AttnParams& p_in, hipStream_t s) {
    constexpr size_t lds = (size_t)4 * 16384 + (size_t)4 * 4 * 256;
    auto kern = attn_swp_kernel;
    static bool attr_done = false;
    if (!attr_done) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) return e;
        attr_done = true;
    }
    g_last_kernel = "attn_swp_kernel";
    const int total = ((p_in.N + 127) / 128) * p_in.H * p_in.B;
    AttnParams p = p_in;
    p.xcd_map = total >= 400 ? 1 : 0;
    p.ablate = g_tune_attn_abl;
    hipLaunchKernelGGL(kern, dim3(8 * ((total + 7) / 8)), dim3(256), lds, s, p);
    return hipGetLastError();
}

template <int NS>
static hipError_t launch_attn_stream(const AttnParams& p_in, hipStream_t s) {
    constexpr size_t lds = (size_t)NS * 16384 + (size_t)NS * 4 * 256;
    auto kern = attn_stream_kernel<NS>;
    static bool attr_done = false;
    if (!attr_done && lds > 48 * 1024) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) return e;
        attr_done = true;
    }
    static char name[40];
    if (!name[0]) snprintf(name, sizeof(name), "attn_stream_kernel<%d>", NS);
    g_last_kernel = name;
    const int total = ((p_in.N + 127) / 128) * p_in.H * p_in.B;
    AttnParams p = p_in;
    p.xcd_map = total >= 400 ? 1 : 0;
    hipLaunchKernelGGL(kern, dim3(8 * ((total + 7) / 8)), dim3(256), lds, s, p);
    return hipGetLastError();
}

template <int QW, int KS, int NS>
static hipError_t launch_attn_pair_cfg(const AttnParams& a, const AttnParams& b, hipStream_t s) {
    constexpr size_t ring = (size_t)NS * KS * (8192 + 8192 + 256 + 16);
    constexpr size_t xch = (size_t)QW * KS * 34 * 64 * 4;
    constexpr size_t lds = ring > xch ? ring : xch;
    auto kern = attn_pair_kernel<QW, KS, NS>;
    static bool attr_done = false;
    if (!attr_done && lds > 48 * 1024) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) return e;
        attr_done = true;
    }
    static char name[48];
    if (!name[0]) snprintf(name, sizeof(name), "attn_pair_kernel<%d,%d,%d>", QW, KS, NS);
    g_last_kernel = name;
    const int ta = ((a.N + 32 * QW - 1) / (32 * QW)) * a.H * a.B, ba = 8 * ((ta + 7) / 8);
    const int bb = ((b.N + 32 * QW - 1) / (32 * QW)) * b.H * b.B;
    hipLaunchKernelGGL(kern, dim3(ba + bb), dim3(64 * QW * KS), lds, s, a, b, ba);
    return hipGetLastError();
}

template <int QW, int KS, int NS>
static hipError_t launch_attn_cfg(const AttnParams& p_in, hipStream_t s) {
    constexpr size_t ring = (size_t)NS * KS * (8192 + 8192 + 256 + 16);
    constexpr size_t xch = (size_t)QW * KS * 34 * 64 * 4;
    constexpr size_t lds = ring > xch ? ring : xch;
    auto kern = attn_kernel<QW, KS, NS>;
    static bool attr_done = false;
    if (!attr_done && lds > 48 * 1024) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) return e;
        attr_done = true;
    }
    static char name[40];
    if (!name[0]) snprintf(name, sizeof(name), "attn_kernel<%d,%d,%d>", QW, KS, NS);
    g_last_kernel = name;
    const int total = ((p_in.N + 32 * QW - 1) / (32 * QW)) * p_in.H * p_in.B;
    AttnParams p = p_in;
    p.xcd_map = total >= 400 ? 1 : 0;
    hipLaunchKernelGGL(kern, dim3(8 * ((total + 7) / 8)), dim3(64 * QW * KS), lds, s, p);
    return hipGetLastError();
}

int g_tune_attn_cfg = -1;      // tools/attn_bench.py override
int g_tune_attn_abl = 0;       // tools/attn_bench.py: ablation bits of attn_pipe_kernel (1 = no DMA in the loop, 2 = no compute); results are garbage

static int pick_attn_cfg(const AttnParams& p) {
    int cfg = g_tune_attn_cfg;
    if (cfg < 0) {
        // measured on MI355X (tools/attn_bench.py sweeps): 128-query workgroups share the K/V tiles once they fill the chip;
        // a single sequence of 5..9 key tiles runs "single shot" (one wave per key tile, every tile in flight) as long as
        // its 32-query workgroups fit the chip in one round; everything in between takes 64 queries x 2 key halves
        const long wg4 = (long)((p.N + 127) / 128) * p.H * p.B;
        const long wg1 = (long)((p.N + 31) / 32) * p.H * p.B;
        const long wg2 = (long)((p.N + 63) / 64) * p.H * p.B;
        const int nt = (p.N + 63) / 64;
        if (wg4 >= 256 && nt >= 2) cfg = 8;          // batched: the streaming kernel, 3-stage ring, 3 workgroups per CU
        else if (nt < 2) cfg = 0;
        else if (wg1 <= 288 && nt >= 5 && nt <= 6) cfg = 5;
        else if (wg1 <= 288 && nt >= 7 && nt <= 9) cfg = 6;
        else if (wg2 <= 256 && nt >= 10) cfg = 7;    // one round of 64-query workgroups, 10+ key tiles (UVLTrack-L, one sequence): four key
                                                     // quarters -- 14.8 -> 13.4 us at N = 873, 13.2 -> 10.9 at N = 681 in isolation, 0..+1.7 % on
                                                     // the frame; two UVLTrack-B sequences (9 tiles) lose 3.6 % with it in the two-stream frame
        else cfg = 1;
    }
    return cfg;
}

hipError_t launch_attention_pair(const AttnParams& a, const AttnParams& b, hipStream_t s) {
    auto ok = [](const AttnParams& p) { return p.N > 0 && p.Npad % 64 == 0 && p.Npad >= ((p.N + 63) / 64) * 64; };
    if (!ok(a) || !ok(b)) return hipErrorInvalidValue;
    const int cfg = pick_attn_cfg(a);
    const int ntb = (b.N + 63) / 64;
    // the rider must fit the configuration's key capacity: single-shot variants hold 6 / 9 key tiles, the ring variants any number
    switch (cfg) {
        case 1: return launch_attn_pair_cfg<2, 2, 2>(a, b, s);
        case 7: return launch_attn_pair_cfg<2, 4, 2>(a, b, s);
        case 5: if (ntb <= 6) return launch_attn_pair_cfg<1, 6, 1>(a, b, s); break;
        case 6: if (ntb <= 9) return launch_attn_pair_cfg<1, 9, 1>(a, b, s); break;
        default: break;
    }
    const hipError_t e = launch_attention(a, s);
    return e != hipSuccess ? e : launch_attention(b, s);
}

hipError_t launch_attention(const AttnParams& p, hipStream_t s) {
    if (p.N <= 0 || p.Npad % 64 != 0 || p.Npad < ((p.N + 63) / 64) * 64) return hipErrorInvalidValue;
    const int cfg = pick_attn_cfg(p);
    switch (cfg) {
        case 0: return launch_attn_cfg<4, 1, 2>(p, s);
        case 1: return launch_attn_cfg<2, 2, 2>(p, s);
        case 2: return launch_attn_cfg<1, 4, 2>(p, s);
        case 3: return launch_attn_cfg<4, 1, 3>(p, s);
        case 4: return launch_attn_cfg<1, 4, 3>(p, s);
        case 5: return launch_attn_cfg<1, 6, 1>(p, s);     // single shot, up to 6 key tiles (N <= 384)
        case 6: return launch_attn_cfg<1, 9, 1>(p, s);     // single shot, up to 9 key tiles (N <= 576): 150 KB of LDS
        case 7: return launch_attn_cfg<2, 4, 2>(p, s);     // 64 queries x 4 key quarters (8 waves, 133 KB of LDS)
        case 8: return launch_attn_stream<3>(p, s);        // batched: 128 queries per workgroup, speculative tiles
        case 9: return launch_attn_stream<2>(p, s);
        case 10: return launch_attn_pipe<3>(p, s);         // + fragment reads a phase ahead of their use
        case 11: return launch_attn_pipe<2>(p, s);         // the same with 2 waves per SIMD (256 registers: nothing spills)
        case 12: return launch_attn_persist<4, 3>(p, s);   // persistent workgroups, 128 queries per item
        case 13: return launch_attn_persist<4, 4>(p, s);
        case 14: return launch_attn_persist<8, 4>(p, s);   // 256 queries per item, one workgroup per CU
        case 15: return launch_attn_persist<8, 6>(p, s);
        case 18: return launch_attn_swp(p, s);             // one wave's MFMAs interleaved with its own softmax arithmetic across tiles
        case 16: return launch_attn_pp<3>(p, s);           // two 4-wave groups per workgroup in ping-pong (VALU slot / MFMA slot)
    }
    return hipErrorInvalidValue;
}

}  // namespace uvl
