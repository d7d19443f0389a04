// Fused multi-head self-attention core for gfx950: softmax(q k^T / 8 + key_add) v, head_dim 64.
//
// Replaces the materialised [B,H,N,N] score tensor of Attention.forward (reference block.py:50-58:
// q@k^T * scale -> masked_fill(-1e10) -> softmax -> @v) and BertSelfAttention.forward
// (bert_backbone.py:311-324: scores/sqrt(64) + (1-mask)*-10000 -> softmax -> @v).
//
// Layout/algorithm (wave64, v_mfma_f32_32x32x16_bf16):
//   * one wave owns 32 query rows; NW waves per workgroup share LDS-staged 64-key K and V^T tiles
//     (HBM -> VGPR -> LDS, next tile's loads issued before the current tile's math)
//   * scores are computed TRANSPOSED, S^T = K Q^T, so lane (q = lane&31) holds 16 of the 32 keys of its
//     query column per 32-key block: the online-softmax row max / row sum are lane-local plus ONE
//     cross-half exchange (__shfl_xor 32)
//   * P^T feeds the second MFMA as the B operand with no data movement at all: the contraction slot of
//     lane-half g, element e is bound to key (16t + 8(e>>2) + 4g + (e&3)), and the V^T A-operand is read
//     from LDS in that same order (two ds_read_b64 per fragment)
//   * O^T = V^T P^T accumulates [d][q]; the rescale factor and 1/l are per-lane scalars
//   * key_add is a per-key additive f32 term staged with the tile; keys >= N get -inf
//   * K rows / V^T columns beyond N are zero-filled when staged, so garbage in the padded workspace
//     can never reach an accumulator
#include <cstdio>
#include "common.h"
#include "kernels.h"

namespace uvl {

template <int QW, int KS>
__global__ __launch_bounds__(64 * QW * KS) void attn_kernel(const AttnParams p) {
    // QW waves along queries (32 rows each) x KS waves along keys: wave (qw, ks) walks key tiles ks, ks+KS, ...
    // of its 32 queries; the KS partial (m, l, O) triples are merged through LDS at the end.  KS > 1 is the
    // batch-1 shape: it multiplies the number of resident waves and divides the serial tile chain by KS.
    constexpr int NT = 64 * QW * KS;
    constexpr int CPT = (KS * 512) / NT;          // 16-byte chunks per thread per round for each of K and V^T
    constexpr int K_BYTES = 8192, V_BYTES = 8192, ADD_BYTES = 256, SLOT = K_BYTES + V_BYTES + ADD_BYTES;
    constexpr int BUF = KS * SLOT;
    extern __shared__ __attribute__((aligned(16))) char smem[];      // 2 * BUF

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, half = lane >> 5;
    const int qw = wave / KS, ks = wave % KS;
    const int h = blockIdx.y, b = blockIdx.z;
    const int N = p.N, Npad = p.Npad;
    const size_t bh = (size_t)b * p.H + h;
    const bf16_t* __restrict__ Q = p.q + bh * Npad * 64;
    const bf16_t* __restrict__ K = p.k + bh * Npad * 64;
    const bf16_t* __restrict__ Vt = p.vt + bh * 64 * Npad;
    const float* __restrict__ kadd = p.key_add + (size_t)b * p.key_add_stride;

    const int q0 = (blockIdx.x * QW + qw) * 32;
    const int qrow = q0 + (lane & 31);
    const int qld = qrow < N ? qrow : N - 1;
    bf16x8 qf[4];
#pragma unroll
    for (int kk = 0; kk < 4; ++kk) qf[kk] = *reinterpret_cast<const bf16x8*>(Q + (size_t)qld * 64 + (2 * kk + half) * 8);

    const int nt = (N + 63) >> 6;                 // key tiles
    const int rounds = (nt + KS - 1) / KS;
    u32x4 rk[CPT], rv[CPT];
    float radd = 0.f;
    auto load_round = [&](int r) __attribute__((always_inline)) {
#pragma unroll
        for (int it = 0; it < CPT; ++it) {
            const int c = tid + it * NT, slot = c >> 9, cc = c & 511, row = cc >> 3, ch = cc & 7;
            const int k0 = (r * KS + slot) * 64;
            const u32x4 zero = {0u, 0u, 0u, 0u};
            if (k0 < N) {
                const u32x4 kv = *reinterpret_cast<const u32x4*>(K + (size_t)(k0 + row) * 64 + ch * 8);
                rk[it] = (k0 + row < N) ? kv : zero;
                u32x4 vv = *reinterpret_cast<const u32x4*>(Vt + (size_t)row * Npad + k0 + ch * 8);
                const int kb = k0 + ch * 8;
                if (kb + 8 > N) {                 // tail chunk: zero the key columns >= N (bf16 pairs per dword)
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        uint32_t w = vv[e];
                        if (kb + 2 * e >= N) w &= 0xffff0000u;
                        if (kb + 2 * e + 1 >= N) w &= 0x0000ffffu;
                        vv[e] = w;
                    }
                }
                rv[it] = vv;
            } else {
                rk[it] = zero;
                rv[it] = zero;
            }
        }
        if (tid < 64 * KS) {
            const int key = r * KS * 64 + tid;
            radd = (key < N) ? kadd[key] : -INFINITY;
        }
    };
    auto store_round = [&](int buf) __attribute__((always_inline)) {
        char* base = smem + buf * BUF;
#pragma unroll
        for (int it = 0; it < CPT; ++it) {
            const int c = tid + it * NT, slot = c >> 9, cc = c & 511, row = cc >> 3, ch = cc & 7;
            char* sK = base + slot * SLOT;
            char* sV = sK + K_BYTES;
            *reinterpret_cast<u32x4*>(sK + swz128(row, ch)) = rk[it];
            *reinterpret_cast<uint2*>(sV + swz64(row, 2 * ch)) = make_uint2(rv[it][0], rv[it][1]);
            *reinterpret_cast<uint2*>(sV + swz64(row, 2 * ch + 1)) = make_uint2(rv[it][2], rv[it][3]);
        }
        if (tid < 64 * KS) {
            float* sA = reinterpret_cast<float*>(base + (tid >> 6) * SLOT + K_BYTES + V_BYTES);
            sA[tid & 63] = radd;
        }
    };

    f32x16 o[2];
#pragma unroll
    for (int r = 0; r < 16; ++r) { o[0][r] = 0.f; o[1][r] = 0.f; }
    float m_run = -INFINITY, l_run = 0.f;

    load_round(0);
    store_round(0);
    __syncthreads();
    for (int rd = 0; rd < rounds; ++rd) {
        const int buf = rd & 1;
        if (rd + 1 < rounds) load_round(rd + 1);
        if (rd * KS + ks < nt) {
            const char* sK = smem + buf * BUF + ks * SLOT;
            const char* sV = sK + K_BYTES;
            const float* sA = reinterpret_cast<const float*>(sV + V_BYTES);

            // ---- S^T = K Q^T for two 32-key blocks ----
            f32x16 s[2];
#pragma unroll
            for (int jb = 0; jb < 2; ++jb) {
#pragma unroll
                for (int r = 0; r < 16; ++r) s[jb][r] = 0.f;
#pragma unroll
                for (int kk = 0; kk < 4; ++kk) {
                    const bf16x8 kf = *reinterpret_cast<const bf16x8*>(sK + swz128(32 * jb + (lane & 31), 2 * kk + half));
                    s[jb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kf, qf[kk], s[jb], 0, 0, 0);
                }
            }
            // ---- scale, per-key additive term, tile max ----
            float tmax = -INFINITY;
#pragma unroll
            for (int jb = 0; jb < 2; ++jb)
#pragma unroll
                for (int gq = 0; gq < 4; ++gq) {
                    const float4 a4 = *reinterpret_cast<const float4*>(sA + 32 * jb + 8 * gq + 4 * half);
                    const float av[4] = {a4.x, a4.y, a4.z, a4.w};
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        const float v = fmaf(s[jb][4 * gq + e], 0.125f, av[e]);
                        s[jb][4 * gq + e] = v;
                        tmax = fmaxf(tmax, v);
                    }
                }
            tmax = fmaxf(tmax, __shfl_xor(tmax, 32, 64));
            const float m_new = fmaxf(m_run, tmax);
            const float alpha = __expf(m_run - m_new);
            m_run = m_new;
            float psum = 0.f;
#pragma unroll
            for (int jb = 0; jb < 2; ++jb)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const float pv = __expf(s[jb][r] - m_new);
                    s[jb][r] = pv;
                    psum += pv;
                }
            l_run = l_run * alpha + psum;
#pragma unroll
            for (int r = 0; r < 16; ++r) { o[0][r] *= alpha; o[1][r] *= alpha; }

            // ---- O^T += V^T P^T ----
#pragma unroll
            for (int jb = 0; jb < 2; ++jb)
#pragma unroll
                for (int t = 0; t < 2; ++t) {
                    union { uint32_t u[4]; bf16x8 v; } pf;
#pragma unroll
                    for (int e = 0; e < 4; ++e) pf.u[e] = pack_bf16x2(s[jb][8 * t + 2 * e], s[jb][8 * t + 2 * e + 1]);
                    const int base = 32 * jb + 16 * t;
                    const int c8a = (base + 4 * half) >> 2, c8b = (base + 8 + 4 * half) >> 2;
#pragma unroll
                    for (int db = 0; db < 2; ++db) {
                        const int d = 32 * db + (lane & 31);
                        union { uint2 u[2]; bf16x8 v; } vf;
                        vf.u[0] = *reinterpret_cast<const uint2*>(sV + swz64(d, c8a));
                        vf.u[1] = *reinterpret_cast<const uint2*>(sV + swz64(d, c8b));
                        o[db] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vf.v, pf.v, o[db], 0, 0, 0);
                    }
                }
        }
        if (rd + 1 < rounds) store_round(buf ^ 1);
        __syncthreads();
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
        // ---- merge the KS key-split partials: exchange [wave][34][64] floats through LDS (staging is dead now) ----
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
            sc[w] = __expf(grp[(w * 34 + 32) * 64 + lane] - mstar);       // exp(-inf) = 0 for a wave that saw no tile
            lsum += grp[(w * 34 + 33) * 64 + lane] * sc[w];
        }
        lsum += __shfl_xor(lsum, 32, 64);
        const float inv = 1.0f / lsum;
        constexpr int RPW = 32 / KS;                                   // accumulator registers merged by this wave
#pragma unroll
        for (int g4 = 0; g4 < RPW / 4; ++g4) {
            const int r0 = ks * RPW + 4 * g4;                          // r0..r0+3: four consecutive d of one row group
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

template <int QW, int KS>
static hipError_t launch_attn_cfg(const AttnParams& p, hipStream_t s) {
    constexpr size_t lds = 2 * KS * (8192 + 8192 + 256);
    auto kern = attn_kernel<QW, KS>;
    static bool attr_done = false;
    if (!attr_done && lds > 48 * 1024) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) return e;
        attr_done = true;
    }
    static char name[32];
    if (!name[0]) snprintf(name, sizeof(name), "attn_kernel<%d,%d>", QW, KS);
    g_last_kernel = name;
    hipLaunchKernelGGL(kern, dim3((p.N + 32 * QW - 1) / (32 * QW), p.H, p.B), dim3(64 * QW * KS), lds, s, p);
    return hipGetLastError();
}

hipError_t launch_attention(const AttnParams& p, hipStream_t s) {
    if (p.N <= 0 || p.Npad % 64 != 0 || p.Npad < ((p.N + 63) / 64) * 64) return hipErrorInvalidValue;
    // enough (batch x heads x query blocks) to fill the chip with 128-query workgroups -> share K/V tiles across
    // 4 query waves; otherwise split the KEYS across the 4 waves (batch-1 latency shape)
    const long wg4 = (long)((p.N + 127) / 128) * p.H * p.B;
    const int nt = (p.N + 63) / 64;
    if (wg4 >= 512) return launch_attn_cfg<4, 1>(p, s);
    if (nt >= 4) return launch_attn_cfg<1, 4>(p, s);
    if (nt >= 2) return launch_attn_cfg<2, 2>(p, s);
    return launch_attn_cfg<4, 1>(p, s);
}

}  // namespace uvl
