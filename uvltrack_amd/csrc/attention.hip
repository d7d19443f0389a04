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
//     column per 32-key block: row max / row sum are lane-local plus ONE cross-half exchange (v_permlane32_swap: common.h::half_max / half_sum)
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
__device__ __forceinline__ void attn_body(const AttnParams& p, const int bx, const int h, const int b, char* smem, const int tid_in = -1) {
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

    const int tid = tid_in >= 0 ? tid_in : (int)threadIdx.x, lane = tid & 63, half = lane >> 5;      // (tid_in: a caller behind an asm block that owns every VGPR rebuilds it from the hardware)
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
            tmax = half_max(tmax);
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
        const float l_tot = half_sum(l_run);
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
        lsum = half_sum(lsum);
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
// (fq / fh: fastdiv_of(nqb) / fastdiv_of(H) from the launcher -- AttnParams::fd_nqb / fd_h: no integer division in front of the first load)
__device__ __forceinline__ bool attn_decode_block(int blk, int total, int nqb, int H, bool mapped, const FastDiv& fq, const FastDiv& fh, int& qb, int& h, int& b) {
    const int xcd = blk & 7, idx = blk >> 3;
    const int cnt = (total + 7) >> 3;
    const int L = mapped ? xcd * cnt + idx : blk;
    if ((mapped && idx >= cnt) || L >= total) return false;
    const int r = (int)fd_div((uint32_t)L, fq);
    qb = L - r * nqb;
    b = (int)fd_div((uint32_t)r, fh);
    h = r - b * H;
    return true;
}

template <int QW, int KS, int NS>
__global__ __launch_bounds__(64 * QW * KS, (KS == 1 ? 3 : 1)) void attn_kernel(const AttnParams p) {
    kernarg_warm<sizeof(AttnParams)>();
    extern __shared__ __attribute__((aligned(16))) char smem[];      // NS * STAGE (>= the merge exchange area)
    const int nqb = (p.N + 32 * QW - 1) / (32 * QW);
    int qb, h, b;
    if (!attn_decode_block((int)blockIdx.x, nqb * p.H * p.B, nqb, p.H, p.xcd_map != 0, p.fd_nqb, p.fd_h, qb, h, b)) return;
    attn_body<QW, KS, NS>(p, qb, h, b, smem);
}

// Two independent attention problems in one launch (batch-1 frames: the 40-token text-branch attention rides on the visual
// one's configuration).  1-D grid, problem A owns [0, blocks_a); (query block, head, sample) are decoded per problem.
template <int QW, int KS, int NS>
__global__ __launch_bounds__(64 * QW * KS, (KS == 1 ? 3 : 1)) void attn_pair_kernel(const AttnParams pa, const AttnParams pb, int blocks_a) {
    kernarg_warm<2 * sizeof(AttnParams) + 8>();
    extern __shared__ __attribute__((aligned(16))) char smem[];
    if ((int)blockIdx.x < blocks_a) {                 // blocks_a is a multiple of 8: the XCD relation of the map is preserved
        const int nqb = (pa.N + 32 * QW - 1) / (32 * QW);
        int qb, h, b;
        if (!attn_decode_block((int)blockIdx.x, nqb * pa.H * pa.B, nqb, pa.H, pa.xcd_map != 0, pa.fd_nqb, pa.fd_h, qb, h, b)) return;
        attn_body<QW, KS, NS>(pa, qb, h, b, smem);
    } else {
        const int id = (int)blockIdx.x - blocks_a;
        const int nqb = (pb.N + 32 * QW - 1) / (32 * QW);
        const int r = (int)fd_div((uint32_t)id, pb.fd_nqb), bb = (int)fd_div((uint32_t)r, pb.fd_h);
        attn_body<QW, KS, NS>(pb, id - r * nqb, r - bb * pb.H, bb, smem);
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
    kernarg_warm<sizeof(AttnParams)>();
    constexpr int STAGE = 16384;                          // K tile 8 KB + V^T tile 8 KB
    constexpr int KADD0 = NS * STAGE;                     // [NS][4 waves][64] f32 key_add rows
    constexpr int VM = 5;                                 // VMEM operations per wave and tile: 2 K + 2 V^T pieces + the key_add row
    static_assert(NS == 2 || NS == 3, "ring depth");
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int nqb = (p.N + 127) / 128;
    int qb, h, b;
    if (!attn_decode_block((int)blockIdx.x, nqb * p.H * p.B, nqb, p.H, p.xcd_map != 0, p.fd_nqb, p.fd_h, qb, h, b)) return;

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
            // the two key blocks' chains alternate: a lone wave with one accumulator chain in flight runs the matrix pipe at well
            // under half rate (tools/probes/exp_mfma_probe.hip)
#pragma unroll
            for (int kk = 0; kk < 4; ++kk)
#pragma unroll
                for (int jb = 0; jb < 2; ++jb) {
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
            for (int kk = 0; kk < 4; ++kk)
#pragma unroll
                for (int jb = 0; jb < 2; ++jb) {
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
            tmax = half_max(tmax);
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
        const float l_tot = half_sum(l_run);
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
    p.fd_nqb = fastdiv_of((uint32_t)((p.N + 127) / 128)); p.fd_h = fastdiv_of((uint32_t)p.H);
    hipLaunchKernelGGL(kern, dim3(8 * ((total + 7) / 8)), dim3(256), lds, s, p);
    return hipGetLastError();
}


// ------------------------------------------------------------------------------------------------
// attn_w64_kernel<NS>: 64 queries per wave (two 32-query blocks), four waves = 256 queries per workgroup, one wave per SIMD
// with the whole register file (the guide's one-wave-per-SIMD shape).  Same LDS image, DMA ring and MFMA operand trick as
// attn_stream_kernel; what changes is the work per wave between two barriers and who schedules it:
//   * every K / V^T fragment read from LDS feeds two MFMAs (one per query block): half the LDS reads, DMA issues and barriers
//     per MFMA
//   * pass 1 uses NO running maximum at all: p = exp2(s) straight from the score MFMA (q carries log2(e)/8).  In bf16 / f32 the
//     absolute scale of P is irrelevant (relative precision), so this is exact softmax arithmetic as long as exp2 neither
//     overflows nor flushes a whole row: checked ONCE per item after the last tile (2^-100 < row sum < 2^100, NaN fails); a
//     workgroup that fails redoes its item in pass 2, the textbook online softmax with true row maxima (compiler-scheduled).
//     Masked keys need no special case in pass 1 (exp2(-1.4e10) = 0); a row whose keys are ALL masked has row sum 0 -> pass 2
//   * the pass-1 tile is one straight-line block in five phases pinned with sched_barrier: score MFMAs of block 0 | score MFMAs
//     of block 1 beside the exponentials / row sums / bf16 packing of block 0 (four exponentials per MFMA) and the V^T fragment
//     reads | P V MFMAs of block 0 beside the exponentials of block 1 | P V MFMAs of block 1 beside the next tile's DMA issue.
//     hipcc's own order is all MFMAs, then all exponentials (measured 537 TFLOP/s against the stream kernel's 654 at B = 32,
//     H = 16, N = 681)
//   * pass 1 issues a DMA round only while one is due (do_issue = t + 2 < nt): the tile exists in two phase-1 variants, with
//     and without the round, and the counted vmcnt waits differ accordingly.  The steady-state LDS reads are inline asm with
//     hand-counted lgkmcnt waits that hipcc cannot see: their correctness rests on the sched_barrier placement and on hipcc not
//     inserting an LDS / SMEM operation between them -- re-run the forced attn_cfg = 10 parity cases (masked, tail, pass 2) after
//     any toolchain change (tests/test_kernels_gpu.py::test_attention_batched_kernels_forced)
// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ float attn_vadd(float a, float b) {   // one v_add_f32, never SLP-packed into v_pk_add_f32 (guide: anti-lever beside MFMAs)
    float d;
    asm("v_add_f32 %0, %1, %2" : "=v"(d) : "v"(a), "v"(b));
    return d;
}
#define ATTN_SB() __builtin_amdgcn_sched_barrier(0)
// the lane index, recomputed where it is used: lane-constant addresses of RARE paths (tail fix-up, mask term) are otherwise hoisted
// out of the tile loop and held (or spilled) across it
__device__ __forceinline__ int attn_opaque_lane() {
    int l;
    asm volatile("v_mbcnt_lo_u32_b32 %0, -1, 0\n\tv_mbcnt_hi_u32_b32 %0, -1, %0" : "=v"(l));
    return l;
}

#ifdef ATTN_WGTRACE
// development build only (tools/attn_wgtrace.py): constant-clock (100 MHz) stamps of wave 0 of EVERY workgroup -- entry, loop start,
// loop end, exit -- plus the hardware slot it ran on: the whole-chip timeline of a launch
__device__ unsigned long long g_attn_wg[8192 * 6];
#define ATTN_WGSTAMP(i) do { ATTN_SB(); wg_t[i] = __builtin_amdgcn_s_memrealtime(); asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); ATTN_SB(); } while (0)
#else
#define ATTN_WGSTAMP(i) do { } while (0)
#endif
#ifdef ATTN_TRACE
// development build only (tools/attn_trace.py): shader-clock stamps of wave 0 of workgroup 0, summed per phase over the tiles
__device__ unsigned long long g_attn_trace[16 + 8 * 32];
#define ATTN_TRACE_OFF (4 * 16384 + 4 * 4 * 256 + 16)       // behind the kernel's own LDS (two workgroups per CU still fit)
#define ATTN_STAMP(i) do { ATTN_SB(); const unsigned long long now_ = __builtin_amdgcn_s_memtime(); asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); \
                           tr_acc[i] += now_ - tr_last; \
                           if (blockIdx.x == 0 && threadIdx.x == 0 && tr_tile < 32) reinterpret_cast<unsigned int*>(smem + ATTN_TRACE_OFF)[tr_tile * 8 + (i)] = (unsigned int)(now_ - tr_last); \
                           tr_last = now_; ATTN_SB(); } while (0)
#else
#define ATTN_STAMP(i) do { } while (0)
#endif

// One 256-query item (query block qb of head h of sample b).  PASS1 = false: the exact pass only (attn_p64_kernel's flagged items).
template <int NS, bool PASS1>
__device__ __forceinline__ void attn_w64_item(const AttnParams& p, const int qb, const int h, const int b, char* smem, const int tid) {
    constexpr int STAGE = 16384;                          // K tile 8 KB + V^T tile 8 KB
    constexpr int KADD0 = NS * STAGE;                     // [NS][4 waves][64] f32 key_add rows
    constexpr int FLAG0 = KADD0 + NS * 4 * 256;           // 4 x int: "this wave wants the exact pass"
    constexpr int VM = 5;                                 // VMEM operations per wave and tile: 2 K + 2 V^T pieces + the key_add row
    static_assert(NS == 4, "ring depth");
#ifdef ATTN_WGTRACE
    unsigned long long wg_t[4];
    ATTN_WGSTAMP(0);
#endif
#ifdef ATTN_TRACE
    unsigned long long tr_acc[8] = {0, 0, 0, 0, 0, 0, 0, 0}, tr_last = __builtin_amdgcn_s_memtime();
    const unsigned long long tr_start = tr_last;
    int tr_slot0 = 0, tr_tile = 0;
#endif

    const int lane = tid & 63, half = lane >> 5;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int N = p.N, Npad = p.Npad;
    const size_t bh = (size_t)b * p.H + h;
    const bf16_t* __restrict__ Q = p.q + bh * Npad * 64;
    const char* Kb = reinterpret_cast<const char*>(p.k + bh * Npad * 64);
    const char* Vb = reinterpret_cast<const char*>(p.vt + bh * 64 * Npad);
    const char* Ab = reinterpret_cast<const char*>(p.key_add + (size_t)b * p.key_add_stride);
    const int nt = (N + 63) >> 6;
    const int q0 = (qb * 4 + wave) * 64;
    const bool active = q0 < N;                           // wave-uniform
    bf16x8 qf[2][4];
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
    // one of the five DMA instructions of a round (i = 0, 1: K pieces, 2, 3: V^T pieces, 4: this wave's key_add row)
    auto issue1 = [&](int t, int stage, auto ic) __attribute__((always_inline)) {      // stage: a constant wherever the ring is live
        constexpr int I = decltype(ic)::value;
        const int ST = stage;
        char* st = smem + ST * STAGE;
        if (I < 2) ATTN_GLDS(pin(Kb + (size_t)t * 8192) + voff[I], st + (wave + 4 * I) * 1024, 16);
        else if (I < 4) ATTN_GLDS(pin(Vb + (size_t)t * 128) + voff[I], st + 8192 + (wave + 4 * (I - 2)) * 1024, 16);
        else ATTN_GLDS(pin(Ab + (size_t)t * 256) + lane * 4, smem + KADD0 + (ST * 4 + wave) * 256, 4);
    };
    auto issue = [&](int t, int stc) __attribute__((always_inline)) {
        issue1(t, stc, AttnIC<0>{}); issue1(t, stc, AttnIC<1>{}); issue1(t, stc, AttnIC<2>{}); issue1(t, stc, AttnIC<3>{}); issue1(t, stc, AttnIC<4>{});
    };
    // tail tile: zero K rows / V^T columns beyond N in LDS (own pieces, after the own DMA wait, before the barrier)
    auto tail_fix = [&](char* sK, int k0) __attribute__((always_inline)) {
        const int lane = attn_opaque_lane();
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
    };

    const int m31 = lane & 31;
    const int kperm = (m31 & 0x13) | ((m31 & 4) << 1) | ((m31 & 8) >> 1);
    int koff[4], voff2[2][2];
#pragma unroll
    for (int kk = 0; kk < 4; ++kk) koff[kk] = swz128(kperm, 2 * kk + half);
#pragma unroll
    for (int jb = 0; jb < 2; ++jb)
#pragma unroll
        for (int t = 0; t < 2; ++t) voff2[jb][t] = swz128(m31, 4 * jb + 2 * t + half);

    f32x16 o[2][2];
    float l_run[2] = {0.f, 0.f};
#pragma unroll
    for (int x = 0; x < 2; ++x)
#pragma unroll
        for (int r = 0; r < 16; ++r) { o[x][0][r] = 0.f; o[x][1][r] = 0.f; }
    const f32x16 zero16 = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};

    auto load_q = [&]() __attribute__((always_inline)) {
#pragma unroll
        for (int x = 0; x < 2; ++x) {
            const int qrow = q0 + 32 * x + (lane & 31);
            const int qld = qrow < N ? qrow : N - 1;
#pragma unroll
            for (int kk = 0; kk < 4; ++kk) qf[x][kk] = *reinterpret_cast<const bf16x8*>(Q + (size_t)qld * 64 + (2 * kk + half) * 8);
            if (!p.q_prescaled) {
#pragma unroll
                for (int kk = 0; kk < 4; ++kk)
#pragma unroll
                    for (int e = 0; e < 8; ++e) qf[x][kk][e] = f2bf(bf2f(qf[x][kk][e]) * (0.125f * ATTN_LOG2E));
            }
        }
    };

    // =============================== pass 1: p = exp2(s), no maximum ===============================
    // Software pipeline across tiles, four phases of 8 MFMAs, each beside ONE half of a block's softmax (16 exponentials, 16 row-sum
    // adds, 8 packs: five single-issue fillers per MFMA, the guide's budget for one wave):
    //   phase 1: scores of block 0, tile t      | softmax of block 1, tile t-1, keys 32..63   + this tile's DMA round (t + 2)
    //   phase 2: scores of block 1, tile t      | softmax of block 0, tile t, keys 0..31      + V^T fragments of tile t-1
    //   phase 3: P V of block 1, tile t-1       | softmax of block 0, tile t, keys 32..63     + V^T fragments of tile t
    //   phase 4: P V of block 0, tile t         | softmax of block 1, tile t, keys 0..31
    // Block 1 trails block 0 by half a tile, so no phase is MFMA-only or VALU-only.  Tile t-1's V^T stage is read in tile t: the
    // ring has four stages (t-1, t, and rounds t+1, t+2 in flight) and round t+2 is issued behind tile t's barrier.
    if constexpr (PASS1) {
        static_assert(NS == 4, "the pipelined pass needs four stages");
        float ps[2][2] = {{0.f, 0.f}, {0.f, 0.f}};
        union PF { uint32_t u[4]; bf16x8 v; };
        f32x16 s1c;                                         // scores of block 1, keys 32..63 of the previous tile (not yet exponentiated)
        PF pf1a[2];                                         // P of block 1, keys 0..31 of the previous tile
#pragma unroll
        for (int r = 0; r < 16; ++r) s1c[r] = -INFINITY;    // "tile -1": p = 0
#pragma unroll
        for (int t2 = 0; t2 < 2; ++t2)
#pragma unroll
            for (int e = 0; e < 4; ++e) pf1a[t2].u[e] = 0u;

        // The ring stage is a RUN-TIME LDS offset (one copy of the tile: 15 KB of code; unrolled by the ring depth it is 59 KB and
        // spills).  hipcc cannot tell a run-time-staged ds_read from the round in flight and would drain the ring (s_waitcnt
        // vmcnt(0)) in front of every fragment read, so the steady-state LDS reads are inline asm with hand-placed lgkmcnt waits
        // (the phases are pinned with sched_barrier anyway); only the rare paths (mask term, tail fix-up) use ordinary LDS accesses.
        const uint32_t lds0 = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) char*)smem;
        auto tile = [&](const int t, const int ST) __attribute__((always_inline)) {
            const int STN = (ST + 2) & 3;                   // the stage of the round issued in this tile (last read in tile t-1: V^T of t-2)
            const int STP = t == 0 ? ST : ((ST + 3) & 3);   // the previous tile's stage ("tile -1" reads this tile's V^T against p = 0)
            char* sK = smem + ST * STAGE;
            const uint32_t aK = lds0 + ST * STAGE, aV = aK + 8192, aVp = lds0 + STP * STAGE + 8192;
            float* sA = reinterpret_cast<float*>(smem + KADD0 + (ST * 4 + wave) * 256);
            ATTN_STAMP(0);                                  // [0] = between tiles (loop control)
            if (t + 1 < nt) attn_wait_vmcnt<VM>();          // round t+1 may stay in flight
            else attn_wait_vmcnt<0>();
            const int k0 = t * 64;
            if (t == nt - 1 && (N & 63)) {
                tail_fix(sK, k0);
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            }
            __builtin_amdgcn_s_barrier();
            ATTN_STAMP(1);                                  // [1] = DMA wait + barrier
            const int tn = t + 2;
            const bool do_issue = tn < nt;
            if (!active) { if (do_issue) issue(tn, STN); return; }

            bf16x8 kf[4][2];
#pragma unroll
            for (int kk = 0; kk < 4; ++kk) {
                const uint32_t ak = aK + koff[kk];
                asm volatile("ds_read_b128 %0, %1" : "=v"(kf[kk][0]) : "v"(ak) : "memory");
                asm volatile("ds_read_b128 %0, %1 offset:4096" : "=v"(kf[kk][1]) : "v"(ak) : "memory");
            }
            float ka;                                         // this wave's own copy of the tile's key_add row: its latency rides on the K reads
            {
                const uint32_t aa = lds0 + KADD0 + (ST * 4 + wave) * 256 + lane * 4;
                asm volatile("ds_read_b32 %0, %1" : "=v"(ka) : "v"(aa) : "memory");
            }
            f32x16 s0[2], s1[2];
            bf16x8 vf[2][2][2], vfp[2][2][2];
            PF pf0[2][2], pf1b[2];
            // the mask term of this tile (log2 domain, -inf beyond N), added in the score registers' key order; re-read from the
            // staged row for each block
            auto add_mask = [&](f32x16 (&sx)[2]) __attribute__((always_inline)) {
                const int half = attn_opaque_lane() >> 5;
#pragma unroll
                for (int jb = 0; jb < 2; ++jb)
#pragma unroll
                    for (int gq = 0; gq < 4; ++gq) {          // registers 4gq..4gq+3 = keys 16(gq>>1) + 8 half + 4(gq&1) + 0..3
                        const int kq = 32 * jb + 16 * (gq >> 1) + 8 * half + 4 * (gq & 1);
                        const float4 a4 = *reinterpret_cast<const float4*>(sA + kq);
                        const float a[4] = {a4.x, a4.y, a4.z, a4.w};
#pragma unroll
                        for (int e = 0; e < 4; ++e) sx[jb][4 * gq + e] += (k0 + kq + e < N) ? a[e] * ATTN_LOG2E : -INFINITY;
                    }
            };
            // half a block's softmax, slice i of 8: two exponentials, two row-sum adds, one pack (word i & 3 of P fragment i >> 2)
#define ATTN_SM_SLICE(sblk, psx, pfarr, i)                                                                 \
            do {                                                                                           \
                sblk[2 * (i)] = __builtin_amdgcn_exp2f(sblk[2 * (i)]);                                     \
                sblk[2 * (i) + 1] = __builtin_amdgcn_exp2f(sblk[2 * (i) + 1]);                             \
                psx[0] = attn_vadd(psx[0], sblk[2 * (i)]);                                                 \
                psx[1] = attn_vadd(psx[1], sblk[2 * (i) + 1]);                                             \
                pfarr[(i) >> 2].u[(i) & 3] = pack_bf16x2(sblk[2 * (i)], sblk[2 * (i) + 1]);                \
            } while (0)
#define ATTN_LGKM(n) do { asm volatile("s_waitcnt lgkmcnt(%0)" ::"n"(n) : "memory"); ATTN_SB(); } while (0)
            ATTN_SB();
            // ---- phase 1: scores of block 0 | softmax of block 1, previous tile, keys 32..63 | this tile's DMA round ----
            if (do_issue) {
#pragma unroll
                for (int i = 0; i < 8; ++i) {
                    const int kk = i >> 1, jb = i & 1;
                    ATTN_LGKM(8 - i);                         // K fragment i of 8 (+ the key_add word behind them)
                    s0[jb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kf[kk][jb], qf[0][kk], kk == 0 ? zero16 : s0[jb], 0, 0, 0);
                    ATTN_SM_SLICE(s1c, ps[1], pf1b, i);
                    if (i == 1) issue1(tn, STN, AttnIC<0>{});
                    if (i == 2) issue1(tn, STN, AttnIC<1>{});
                    if (i == 3) issue1(tn, STN, AttnIC<2>{});
                    if (i == 4) issue1(tn, STN, AttnIC<3>{});
                    if (i == 5) issue1(tn, STN, AttnIC<4>{});
                    ATTN_SB();
                }
            } else {
#pragma unroll
                for (int i = 0; i < 8; ++i) {
                    const int kk = i >> 1, jb = i & 1;
                    ATTN_LGKM(8 - i);                         // K fragment i of 8 (+ the key_add word behind them)
                    s0[jb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kf[kk][jb], qf[0][kk], kk == 0 ? zero16 : s0[jb], 0, 0, 0);
                    ATTN_SM_SLICE(s1c, ps[1], pf1b, i);
                    ATTN_SB();
                }
            }
            ATTN_LGKM(0);
            ka = (k0 + lane < N) ? ka * ATTN_LOG2E : -INFINITY;                     // log2 domain; keys beyond N never count
            const bool masked = __any(ka != 0.f);
            ATTN_STAMP(2);                                  // [2] = K fragment reads + phase 1
            if (masked) add_mask(s0);                         // wave-uniform
            ATTN_SB();
            // ---- phase 2: scores of block 1 | softmax of block 0, keys 0..31 | V^T fragments of the previous tile ----
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                const int kk = i >> 1, jb = i & 1;
                s1[jb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kf[kk][jb], qf[1][kk], kk == 0 ? zero16 : s1[jb], 0, 0, 0);
                ATTN_SM_SLICE(s0[0], ps[0], pf0[0], i);
                if (i >= 4) {                                 // in the order P V consumes them (behind the K fragments that die here)
                    const int vjb = (i - 4) >> 1, vt2 = (i - 4) & 1;
                    const uint32_t av = aVp + voff2[vjb][vt2];
                    asm volatile("ds_read_b128 %0, %1" : "=v"(vfp[0][vjb][vt2]) : "v"(av) : "memory");
                    asm volatile("ds_read_b128 %0, %1 offset:4096" : "=v"(vfp[1][vjb][vt2]) : "v"(av) : "memory");
                }
                ATTN_SB();
            }
            ATTN_STAMP(3);                                  // [3] = (mask) + phase 2
            if (masked) add_mask(s1);
            ATTN_SB();
            // ---- phase 3: P V of block 1, previous tile | softmax of block 0, keys 32..63 | V^T fragments of this tile ----
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const int jb = j >> 2, t2 = (j >> 1) & 1, db = j & 1;
                if (j == 0) ATTN_LGKM(6);                     // the previous tile's V^T fragments, pair by pair (in-order returns); from
                if (j == 2) ATTN_LGKM(4);                     // group 4 on this tile's fragment reads queue up behind them
                if (j == 4) ATTN_LGKM(2);
                if (j == 6) ATTN_LGKM(4);
                o[1][db] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vfp[db][jb][t2], jb == 0 ? pf1a[t2].v : pf1b[t2].v, o[1][db], 0, 0, 0);
                ATTN_SM_SLICE(s0[1], ps[0], pf0[1], j);
                if (j >= 4) {
                    const int vjb = (j - 4) >> 1, vt2 = (j - 4) & 1;
                    const uint32_t av = aV + voff2[vjb][vt2];
                    asm volatile("ds_read_b128 %0, %1" : "=v"(vf[0][vjb][vt2]) : "v"(av) : "memory");
                    asm volatile("ds_read_b128 %0, %1 offset:4096" : "=v"(vf[1][vjb][vt2]) : "v"(av) : "memory");
                }
                ATTN_SB();
            }
            ATTN_STAMP(4);                                  // [4] = (mask) + phase 3
            ATTN_SB();
            // ---- phase 4: P V of block 0 | softmax of block 1, keys 0..31 ----
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const int jb = j >> 2, t2 = (j >> 1) & 1, db = j & 1;
                if (j == 0) ATTN_LGKM(6);                     // this tile's V^T fragments
                if (j == 2) ATTN_LGKM(4);
                if (j == 4) ATTN_LGKM(2);
                if (j == 6) ATTN_LGKM(0);
                o[0][db] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vf[db][jb][t2], pf0[jb][t2].v, o[0][db], 0, 0, 0);
                ATTN_SM_SLICE(s1[0], ps[1], pf1a, j);
                ATTN_SB();
            }
            s1c = s1[1];
            ATTN_STAMP(5);                                  // [5] = phase 4
#ifdef ATTN_TRACE
            ++tr_tile;
#endif
        };

        // the first two rounds go out before anything else; q shares their flight
        issue(0, 0);
        if (1 < nt) issue(1, 1);
        load_q();


        // q has landed (and rounds 0 / 1 with it), and hipcc KNOWS it (the builtin, not inline asm): otherwise every MFMA that reads a
        // q fragment inside the loop gets a compiler-inserted s_waitcnt vmcnt(0), which drains the DMA ring once per phase
        __builtin_amdgcn_s_waitcnt(0);
        ATTN_STAMP(6);                                      // [6] = prologue (q loads, addresses)
        ATTN_WGSTAMP(1);
        for (int t = 0; t < nt; ++t) tile(t, t & 3);
        ATTN_WGSTAMP(2);
        // drain: block 1 of the last tile
        if (active) {
            PF pf1b[2];
#pragma unroll
            for (int i = 0; i < 8; ++i) ATTN_SM_SLICE(s1c, ps[1], pf1b, i);
            const char* sVp = smem + ((nt - 1) & 3) * STAGE + 8192;
#pragma unroll
            for (int jb = 0; jb < 2; ++jb)
#pragma unroll
                for (int t2 = 0; t2 < 2; ++t2)
#pragma unroll
                    for (int db = 0; db < 2; ++db) {
                        const bf16x8 vfr = *reinterpret_cast<const bf16x8*>(sVp + db * 4096 + voff2[jb][t2]);
                        o[1][db] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vfr, jb == 0 ? pf1a[t2].v : pf1b[t2].v, o[1][db], 0, 0, 0);
                    }
        }
#undef ATTN_SM_SLICE
#undef ATTN_LGKM
#pragma unroll
        for (int x = 0; x < 2; ++x) l_run[x] = ps[x][0] + ps[x][1];
    }

    // one check per item: did exp2 overflow, or flush a whole row?  Workgroup-wide decision (pass 2 needs every wave for its DMA
    // ring and barriers)
    bool bad = !PASS1;
    if (PASS1 && active) {
#pragma unroll
        for (int x = 0; x < 2; ++x) {
            const float l_tot = half_sum(l_run[x]);
            bad = bad || !(l_tot < 1.2676506e30f && l_tot > 7.888609e-31f);      // 2^100, 2^-100; NaN fails both
        }
        bad = __any(bad);
    }
    if constexpr (PASS1) {
        int* flags = reinterpret_cast<int*>(smem + FLAG0);
        if (lane == 0) flags[wave] = bad ? 1 : 0;
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        const int any_bad = flags[0] | flags[1] | flags[2] | flags[3];
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        bad = __builtin_amdgcn_readfirstlane(any_bad) != 0;
    }

    // =============================== pass 2 (rare): online softmax with true row maxima ===============================
    if (bad) {
        __builtin_amdgcn_s_barrier();                       // every wave has read the flags; the ring may be refilled
        if constexpr (!PASS1) { load_q(); __builtin_amdgcn_s_waitcnt(0); }
        float m_run[2] = {-INFINITY, -INFINITY};
#pragma unroll
        for (int x = 0; x < 2; ++x) {
            l_run[x] = 0.f;
#pragma unroll
            for (int r = 0; r < 16; ++r) { o[x][0][r] = 0.f; o[x][1][r] = 0.f; }
        }
        auto tile2 = [&](const int t, const int ST) __attribute__((always_inline)) {
            char* sK = smem + ST * STAGE;
            char* sV = sK + 8192;
            float* sA = reinterpret_cast<float*>(smem + KADD0 + (ST * 4 + wave) * 256);
            if (t + NS - 2 < nt) attn_wait_vmcnt<VM*(NS - 2)>();    // rounds t+1 .. t+NS-2 may stay in flight
            else if (NS >= 4 && t + 1 < nt) attn_wait_vmcnt<VM>();
            else attn_wait_vmcnt<0>();
            const int k0 = t * 64;
            if (t == nt - 1 && (N & 63)) tail_fix(sK, k0);
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();
            if (t + NS - 1 < nt) issue(t + NS - 1, (ST + NS - 1) % NS);
            if (!active) return;
            f32x16 s[2][2];
#pragma unroll
            for (int x = 0; x < 2; ++x)
#pragma unroll
                for (int kk = 0; kk < 4; ++kk)
#pragma unroll
                    for (int jb = 0; jb < 2; ++jb) {
                        const bf16x8 kf = *reinterpret_cast<const bf16x8*>(sK + jb * 4096 + koff[kk]);
                        s[x][jb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kf, qf[x][kk], kk == 0 ? zero16 : s[x][jb], 0, 0, 0);
                    }
#pragma unroll
            for (int jb = 0; jb < 2; ++jb)
#pragma unroll
                for (int gq = 0; gq < 4; ++gq) {
                    const int kq = 32 * jb + 16 * (gq >> 1) + 8 * half + 4 * (gq & 1);
                    const float4 a4 = *reinterpret_cast<const float4*>(sA + kq);
                    const float a[4] = {a4.x, a4.y, a4.z, a4.w};
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        const float ad = (k0 + kq + e < N) ? a[e] * ATTN_LOG2E : -INFINITY;
                        s[0][jb][4 * gq + e] += ad;
                        s[1][jb][4 * gq + e] += ad;
                    }
                }
#pragma unroll
            for (int x = 0; x < 2; ++x) {
                float tmax = -INFINITY;
#pragma unroll
                for (int r = 0; r < 16; ++r) tmax = fmaxf(tmax, fmaxf(s[x][0][r], s[x][1][r]));
                tmax = half_max(tmax);
                const float m_new = fmaxf(m_run[x], tmax);
                const float alpha = __builtin_amdgcn_exp2f(m_run[x] - m_new);      // first tile: exp2(-inf) = 0
                m_run[x] = m_new;
#pragma unroll
                for (int r = 0; r < 16; ++r) { o[x][0][r] *= alpha; o[x][1][r] *= alpha; }
                float ps[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    s[x][0][r] = __builtin_amdgcn_exp2f(s[x][0][r] - m_new);
                    s[x][1][r] = __builtin_amdgcn_exp2f(s[x][1][r] - m_new);
                    ps[r & 1] += s[x][0][r];
                    ps[2 + (r & 1)] += s[x][1][r];
                }
                l_run[x] = l_run[x] * alpha + ((ps[0] + ps[1]) + (ps[2] + ps[3]));
            }
#pragma unroll
            for (int x = 0; x < 2; ++x)
#pragma unroll
                for (int jb = 0; jb < 2; ++jb)
#pragma unroll
                    for (int t2 = 0; t2 < 2; ++t2) {
                        union { uint32_t u[4]; bf16x8 v; } pf;
#pragma unroll
                        for (int e = 0; e < 4; ++e) pf.u[e] = pack_bf16x2(s[x][jb][8 * t2 + 2 * e], s[x][jb][8 * t2 + 2 * e + 1]);
#pragma unroll
                        for (int db = 0; db < 2; ++db) {
                            const bf16x8 vfr = *reinterpret_cast<const bf16x8*>(sV + db * 4096 + voff2[jb][t2]);
                            o[x][db] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vfr, pf.v, o[x][db], 0, 0, 0);
                        }
                    }
        };
        issue(0, 0);
        if (1 < nt) issue(1, 1);
        if (NS == 4 && 2 < nt) issue(2, 2);
        for (int t = 0; t < nt; ++t) tile2(t, t % NS);
    }
#undef ATTN_GLDS
    ATTN_STAMP(0);
#pragma unroll
    for (int x = 0; x < 2; ++x) {
        const int qrow = q0 + 32 * x + (lane & 31);
        if (qrow < N) {
            const float l_tot = half_sum(l_run[x]);
            const float inv = 1.0f / l_tot;
            bf16_t* dst = p.o + ((size_t)b * N + qrow) * (p.H * 64) + h * 64;
            // 16-byte stores (guide T21): the two half-waves hold columns 8g..8g+3 / 8g+4..8g+7 of the same row; one
            // v_permlane32_swap per dword turns a pair of column groups into 16 contiguous bytes per lane
#pragma unroll
            for (int db = 0; db < 2; ++db)
#pragma unroll
                for (int gp = 0; gp < 2; ++gp) {
                    uint32_t a0 = pack_bf16x2(o[x][db][8 * gp + 0] * inv, o[x][db][8 * gp + 1] * inv);
                    uint32_t a1 = pack_bf16x2(o[x][db][8 * gp + 2] * inv, o[x][db][8 * gp + 3] * inv);
                    uint32_t b0 = pack_bf16x2(o[x][db][8 * gp + 4] * inv, o[x][db][8 * gp + 5] * inv);
                    uint32_t b1 = pack_bf16x2(o[x][db][8 * gp + 6] * inv, o[x][db][8 * gp + 7] * inv);
                    const auto r0 = __builtin_amdgcn_permlane32_swap(a0, b0, false, false);
                    const auto r1 = __builtin_amdgcn_permlane32_swap(a1, b1, false, false);
                    const u32x4 w = {(uint32_t)r0[0], (uint32_t)r1[0], (uint32_t)r0[1], (uint32_t)r1[1]};
                    *reinterpret_cast<u32x4*>(dst + 32 * db + 16 * gp + 8 * half) = w;
                }
        }
    }
#ifdef ATTN_WGTRACE
    ATTN_WGSTAMP(3);
    if (threadIdx.x == 0 && blockIdx.x < 8192) {
        unsigned long long* w = g_attn_wg + (size_t)blockIdx.x * 6;
        w[0] = wg_t[0]; w[1] = wg_t[1]; w[2] = wg_t[2]; w[3] = wg_t[3];
        w[4] = (unsigned long long)__builtin_amdgcn_s_getreg(4 | (31 << 11)) | ((unsigned long long)__builtin_amdgcn_s_getreg(20 | (31 << 11)) << 32);
        w[5] = (unsigned long long)((qb << 20) | (h << 12) | b);
    }
#endif
#ifdef ATTN_TRACE
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    ATTN_STAMP(7);                                          // [7] = epilogue (normalise + stores)
    if (blockIdx.x == 0 && threadIdx.x == 0) {
        for (int i = 0; i < 8; ++i) g_attn_trace[i] = tr_acc[i];
        g_attn_trace[8] = tr_last - tr_start;
        g_attn_trace[9] = (unsigned long long)nt;
        for (int i = 0; i < 8 * 32; ++i) g_attn_trace[16 + i] = reinterpret_cast<unsigned int*>(smem + ATTN_TRACE_OFF)[i];
    }
#endif
}

template <int NS>
__global__ __launch_bounds__(256, 2) void attn_w64_kernel(const AttnParams p) {
    kernarg_warm<sizeof(AttnParams)>();
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int nqb = (((p.N + 63) >> 6) + 3) >> 2;         // 256-query workgroups per head
    int qb, h, b;
    if (!attn_decode_block((int)blockIdx.x, nqb * p.H * p.B, nqb, p.H, p.xcd_map != 0, p.fd_nqb, p.fd_h, qb, h, b)) return;
    attn_w64_item<NS, true>(p, qb, h, b, smem, (int)threadIdx.x);
}

template <int NS>
static hipError_t launch_attn_w64(const AttnParams& p_in, hipStream_t s) {
#ifdef ATTN_TRACE
    constexpr size_t lds = (size_t)ATTN_TRACE_OFF + 1024;
#else
    constexpr size_t lds = (size_t)NS * 16384 + (size_t)NS * 4 * 256 + 16;
#endif
    auto kern = attn_w64_kernel<NS>;
    static bool attr_done = false;
    if (!attr_done && lds > 48 * 1024) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) return e;
        attr_done = true;
    }
    static char name[40];
    if (!name[0]) snprintf(name, sizeof(name), "attn_w64_kernel<%d>", NS);
    g_last_kernel = name;
    const int total = ((((p_in.N + 63) / 64) + 3) / 4) * p_in.H * p_in.B;
    AttnParams p = p_in;
    p.xcd_map = total >= 400 ? 1 : 0;
    p.fd_nqb = fastdiv_of((uint32_t)((((p.N + 63) / 64) + 3) / 4)); p.fd_h = fastdiv_of((uint32_t)p.H);
    hipLaunchKernelGGL(kern, dim3(8 * ((total + 7) / 8)), dim3(256), lds, s, p);
    return hipGetLastError();
}


// ------------------------------------------------------------------------------------------------
// attn_p64_kernel: the hand-scheduled form of attn_w64_kernel (same item shape -- four waves x 64 queries --, same LDS image, same
// arithmetic).  Persistent workgroups walk the items; pass 1 of the whole walk is ONE inline-asm statement generated by
// tools/gen/attn_p64_gen.py (attn_p64_asm.inc) that owns every vector register: units of 4 score MFMAs + 40 VALU + 4 P V MFMAs in a
// three-deep software pipeline, 32 live score registers instead of 64, every K / V^T fragment read once per tile.  Items whose row
// sums leave [2^-100, 2^100] come back as bits of `bad` and are redone by attn_w64_item's exact pass (compiler-scheduled, rare).
// Requires pre-scaled q and at least two key tiles (the launcher falls back to attn_w64_kernel otherwise).
// ------------------------------------------------------------------------------------------------
// every register attn_p64_asm.inc names: s38..s101 (its fixed scalar map; hipcc keeps its operands below) and all 256 VGPRs
#define ATTN_R4(p, n) p #n "0", p #n "1", p #n "2", p #n "3"
#define ATTN_R10(p, n) p #n "0", p #n "1", p #n "2", p #n "3", p #n "4", p #n "5", p #n "6", p #n "7", p #n "8", p #n "9"
#define ATTN_P64_SGPRS "s38", "s39", ATTN_R10("s", 4), ATTN_R10("s", 5), ATTN_R10("s", 6), ATTN_R10("s", 7), ATTN_R10("s", 8), ATTN_R10("s", 9)
#define ATTN_P64_VGPRS "v0", "v1", "v2", "v3", "v4", "v5", "v6", "v7", "v8", "v9", ATTN_R10("v", 1), ATTN_R10("v", 2), ATTN_R10("v", 3), ATTN_R10("v", 4), \
    ATTN_R10("v", 5), ATTN_R10("v", 6), ATTN_R10("v", 7), ATTN_R10("v", 8), ATTN_R10("v", 9), ATTN_R10("v", 10), ATTN_R10("v", 11), ATTN_R10("v", 12), \
    ATTN_R10("v", 13), ATTN_R10("v", 14), ATTN_R10("v", 15), ATTN_R10("v", 16), ATTN_R10("v", 17), ATTN_R10("v", 18), ATTN_R10("v", 19), ATTN_R10("v", 20), \
    ATTN_R10("v", 21), ATTN_R10("v", 22), ATTN_R10("v", 23), ATTN_R10("v", 24), "v250", "v251", "v252", "v253", "v254", "v255"
__device__ __forceinline__ void attn_p64_walk(const AttnParams& p, const int total, const int cnt, const int nqb, const uint32_t mq, const uint32_t mh, char* smem, const int wave) {
    const uint32_t lds0 = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) char*)smem;
    const int v0 = (int)blockIdx.x, G = (int)gridDim.x;
    const int kas = p.key_add_stride * 4;
    uint64_t bad = 0;
#if __HIP_DEVICE_COMPILE__              // the host pass of hipcc parses kernel bodies too and knows no gfx950 register names
    asm volatile(
#ifdef ATTN_WGTRACE
#include "attn_p64_asm_trace.inc"
#else
#include "attn_p64_asm.inc"
#endif
        : [bad] "=s"(bad)
        :
#ifdef ATTN_WGTRACE
          [tr] "s"(g_attn_wg),
#endif
          [q] "s"(p.q), [k] "s"(p.k), [vt] "s"(p.vt), [ka] "s"(p.key_add), [o] "s"(p.o), [N] "s"(p.N), [Npad] "s"(p.Npad), [H] "s"(p.H),
          [total] "s"(total), [cnt] "s"(cnt), [v] "s"(v0), [G] "s"(G), [kas] "s"(kas), [wave] "s"(wave), [lds] "s"(lds0), [nqb] "s"(nqb),
          [mq] "s"(mq), [mh] "s"(mh)
        : "memory", "vcc", "scc", ATTN_P64_SGPRS, ATTN_P64_VGPRS
#ifdef ATTN_WGTRACE
          , "s100", "s101"
#endif
        );
#endif
#ifdef ATTN_P64_NOFALLBACK      // timing probes whose pass 1 is wrong by construction: never take the exact pass
    return;
#endif
    if (bad == 0) return;                   // (wave-uniform: the flag word is an SGPR pair)
    int it = 0;
    for (int v = v0; v < 8 * cnt; v += G) {
        int qb, h, b;
        if (!attn_decode_block(v, total, nqb, p.H, true, p.fd_nqb, p.fd_h, qb, h, b)) continue;
        if ((bad >> it) & 1) attn_w64_item<4, false>(p, qb, h, b, smem, wave * 64 + attn_opaque_lane());
        ++it;
    }
}

__global__ __launch_bounds__(256, 2) void attn_p64_kernel(const AttnParams p, const int total, const int cnt, const int nqb, const uint32_t mq, const uint32_t mh) {
    kernarg_warm<sizeof(AttnParams) + 24 + 64>();
    extern __shared__ __attribute__((aligned(16))) char smem[];
    attn_p64_walk(p, total, cnt, nqb, mq, mh, smem, __builtin_amdgcn_readfirstlane((int)threadIdx.x >> 6));
}

// The persistent walk with a RIDER: a second attention problem of few keys (the text branch of a many-sequence frame: B x H items of 40 x 40)
// whose items are taken by workgroups once their own walk is done -- no launch and no second queue for them.  `tail_only`: every workgroup
// walks exactly one 256-query item and the items of the last query block are short (UVLTrack-L, N = 873: 105 of 256 queries), so those
// B x H workgroups take one rider item each and the kernel ends when the full-length items end; otherwise the rider's items go round-robin,
// last workgroup first.
__global__ __launch_bounds__(256, 2) void attn_p64_rider_kernel(const AttnParams p, const int total, const int cnt, const int nqb, const uint32_t mq, const uint32_t mh,
                                                                const AttnParams pb, const int tail_only) {
    kernarg_warm<2 * sizeof(AttnParams) + 32 + 64>();
    extern __shared__ __attribute__((aligned(16))) char smem[];
    // the walk's asm block owns all 256 VGPRs: nothing per-lane may live across it (threadIdx.x kept for the rider was a spill = a scratch
    // segment for every wave of the launch); the wave index is an SGPR and the lane comes from the hardware again
    const int wave = __builtin_amdgcn_readfirstlane((int)threadIdx.x >> 6);
    attn_p64_walk(p, total, cnt, nqb, mq, mh, smem, wave);
    const int tid_r = (wave * 64 + attn_opaque_lane()) & 255;      // (& 255: provably >= 0, so attn_body's `tid_in >= 0 ? tid_in : threadIdx.x` folds)
    const int nqb_b = (pb.N + 127) / 128, total_b = nqb_b * pb.H * pb.B;
    int first = (int)gridDim.x - 1 - (int)blockIdx.x, step = (int)gridDim.x;      // round-robin from the END of the grid: the walk gives the low indices one item more
    if (tail_only) {
        int qb, h, b;
        if (!attn_decode_block((int)blockIdx.x, total, nqb, p.H, true, p.fd_nqb, p.fd_h, qb, h, b) || qb != nqb - 1) return;
        first = h + p.H * b;
        step = p.H * p.B;
    }
    for (int t = first; t < total_b; t += step) {
        __syncthreads();                    // the walk's (or the previous item's) last LDS reads are done before the ring is refilled
        const int qb = t % nqb_b, r = t / nqb_b;
        attn_body<4, 1, 2>(pb, qb, r % pb.H, r / pb.H, smem, tid_r);
    }
}

// rider = nullptr: the plain kernel
static hipError_t launch_attn_p64(const AttnParams& p_in, hipStream_t s, const AttnParams* rider = nullptr) {
    constexpr size_t lds = (size_t)4 * 16384 + (size_t)4 * 4 * 256 + 16;
    static_assert(lds >= 4 * 34 * 64 * 4 && lds >= 2 * (8192 + 8192 + 256 + 16), "the rider's attn_body<4,1,2> fits");
    const int nt = (p_in.N + 63) / 64;
    if (!p_in.q_prescaled || nt < 2) {
        const hipError_t e = launch_attn_w64<4>(p_in, s);
        return (e != hipSuccess || !rider) ? e : launch_attention(*rider, s);
    }
    static bool attr_done[2] = {false, false};
    if (!attr_done[rider ? 1 : 0]) {
        hipError_t e = hipFuncSetAttribute(rider ? reinterpret_cast<const void*>(attn_p64_rider_kernel) : reinterpret_cast<const void*>(attn_p64_kernel),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) return e;
        attr_done[rider ? 1 : 0] = true;
    }
    g_last_kernel = rider ? "attn_p64_rider_kernel" : "attn_p64_kernel";
    const int nqb = (nt + 3) / 4;
    const int total = nqb * p_in.H * p_in.B, cnt = (total + 7) / 8;
    int wgs = tune_get(p_in.tune, &uvl_tuning::attn_wgs, 512);          // persistent workgroups (two per CU)
    wgs = wgs < 8 ? 8 : (wgs + 7) / 8 * 8;
    int grid = 8 * cnt < wgs ? 8 * cnt : wgs;
    while ((8 * cnt + grid - 1) / grid > 64) grid += 8;                   // the flagged-item mask of a workgroup has 64 bits
    auto magic = [](int d) { return d <= 1 ? 0u : (uint32_t)(((1ull << 32) + (uint64_t)d - 1) / (uint64_t)d); };
    AttnParams p = p_in;
    p.xcd_map = 1;
    p.fd_nqb = fastdiv_of((uint32_t)(nqb)); p.fd_h = fastdiv_of((uint32_t)p.H);
    if (rider) {
        // one item per workgroup and a short last query block: the workgroups of those items have the time for the rider
        const int tail_only = (grid == 8 * cnt && nqb >= 2 && p.N - (nqb - 1) * 256 <= 160) ? 1 : 0;
        hipLaunchKernelGGL(attn_p64_rider_kernel, dim3(grid), dim3(256), lds, s, p, total, cnt, nqb, magic(nqb), magic(p_in.H), *rider, tail_only);
    } else {
        hipLaunchKernelGGL(attn_p64_kernel, dim3(grid), dim3(256), lds, s, p, total, cnt, nqb, magic(nqb), magic(p_in.H));
    }
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
    AttnParams pa = a, pb = b;
    pa.fd_nqb = fastdiv_of((uint32_t)((a.N + 32 * QW - 1) / (32 * QW))); pa.fd_h = fastdiv_of((uint32_t)a.H);
    pb.fd_nqb = fastdiv_of((uint32_t)((b.N + 32 * QW - 1) / (32 * QW))); pb.fd_h = fastdiv_of((uint32_t)b.H);
    hipLaunchKernelGGL(kern, dim3(ba + bb), dim3(64 * QW * KS), lds, s, pa, pb, ba);
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
    p.fd_nqb = fastdiv_of((uint32_t)((p.N + 32 * QW - 1) / (32 * QW))); p.fd_h = fastdiv_of((uint32_t)p.H);
    hipLaunchKernelGGL(kern, dim3(8 * ((total + 7) / 8)), dim3(64 * QW * KS), lds, s, p);
    return hipGetLastError();
}

#ifdef ATTN_WGTRACE
extern "C" int uvl_debug_attn_wgtrace(unsigned long long* dst, int n) { return (int)hipMemcpyFromSymbol(dst, HIP_SYMBOL(g_attn_wg), (size_t)n * 6 * sizeof(unsigned long long)); }
#endif
#ifdef ATTN_TRACE
extern "C" int uvl_debug_attn_trace(unsigned long long* dst) { return (int)hipMemcpyFromSymbol(dst, HIP_SYMBOL(g_attn_trace), (16 + 8 * 32) * sizeof(unsigned long long)); }
#endif
static int pick_attn_cfg(const AttnParams& p) {
    int cfg = tune_get(p.tune, &uvl_tuning::attn_cfg, -1);      // tools / tests: index into launch_attention's table
    if (cfg < 0) {
        // measured on MI355X (tools/attn_bench.py sweeps): 128-query workgroups share the K/V tiles once they fill the chip;
        // a single sequence of 5..9 key tiles runs "single shot" (one wave per key tile, every tile in flight) as long as
        // its 32-query workgroups fit the chip in one round; everything in between takes 64 queries x 2 key halves
        const long wg4 = (long)((p.N + 127) / 128) * p.H * p.B;
        const long wg1 = (long)((p.N + 31) / 32) * p.H * p.B;
        const long wg2 = (long)((p.N + 63) / 64) * p.H * p.B;
        const int nt = (p.N + 63) / 64;
        const long wg64 = (long)((nt + 3) / 4) * p.H * p.B;     // 256-query workgroups of attn_w64_kernel (two per CU)
        if (wg64 > 256 && wg64 < 384 && wg4 <= 512 && nt >= 2) cfg = 8;   // 8 UVLTrack-B sequences (288 items of 256 queries = 0.56 of a round of cfg 11's
                                                     // 512 slots against 480 workgroups of the streaming kernel): 17.9 vs 18.8 us, interleaved rounds
        else if (wg4 >= 256 && nt >= 2 && p.q_prescaled) cfg = 11;   // the batched regime: the hand-scheduled persistent walk (attn_p64_kernel), ahead of
                                                     // both kernels below on every measured shape from 4 sequences up (profiles/r03_attention.md:
                                                     // +7..+23 % over attn_w64_kernel with interleaved rounds; 8 x N = 553 is 5 % behind the streaming kernel, everything else ahead)
        else if (wg64 >= 384 && nt >= 2) cfg = 10;   // 1.5+ workgroups per CU of the 64-queries-per-wave kernel: +6..+29 % over the streaming
                                                     // kernel on every measured shape from 384 workgroups up (profiles/r02_attention_w64.md),
                                                     // except 576 workgroups of N = 553 (-4 %: a second round at 75 % wave occupancy)
        else if (wg4 >= 256 && nt >= 2) cfg = 8;     // batched, fewer workgroups: the streaming kernel, 3-stage ring, 3 workgroups per CU
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
        case 11: return launch_attn_p64(a, s, &b);         // many sequences: the rider's items behind the persistent walk
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
        case 10: return launch_attn_w64<4>(p, s);          // 64 queries per wave, two workgroups per CU
        case 11: return launch_attn_p64(p, s);             // the same item shape, hand-scheduled, persistent workgroups
    }
    return hipErrorInvalidValue;
}

}  // namespace uvl
