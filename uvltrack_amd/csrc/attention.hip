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
#include "common.h"
#include "kernels.h"

namespace uvl {

template <int NW>
__global__ __launch_bounds__(64 * NW) void attn_kernel(const AttnParams p) {
    constexpr int NT = 64 * NW;
    constexpr int CPT = 512 / NT;                 // 16-byte chunks per thread for each of the K and V^T tiles
    constexpr int K_BYTES = 8192, V_BYTES = 8192, ADD_BYTES = 256, BUF = K_BYTES + V_BYTES + ADD_BYTES;
    __shared__ __attribute__((aligned(16))) char smem[2 * BUF];

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, half = lane >> 5;
    const int h = blockIdx.y, b = blockIdx.z;
    const int N = p.N, Npad = p.Npad;
    const size_t bh = (size_t)b * p.H + h;
    const bf16_t* __restrict__ Q = p.q + bh * Npad * 64;
    const bf16_t* __restrict__ K = p.k + bh * Npad * 64;
    const bf16_t* __restrict__ Vt = p.vt + bh * 64 * Npad;
    const float* __restrict__ kadd = p.key_add + (size_t)b * p.key_add_stride;

    const int q0 = (blockIdx.x * NW + wave) * 32;
    const int qrow = q0 + (lane & 31);
    const int qld = qrow < N ? qrow : N - 1;
    bf16x8 qf[4];
#pragma unroll
    for (int kk = 0; kk < 4; ++kk) qf[kk] = *reinterpret_cast<const bf16x8*>(Q + (size_t)qld * 64 + (2 * kk + half) * 8);

    uint4 rk[CPT], rv[CPT];
    float radd = 0.f;
    auto load_tile = [&](int j) __attribute__((always_inline)) {
        const int k0 = j * 64;
#pragma unroll
        for (int it = 0; it < CPT; ++it) {
            const int c = tid + it * NT, row = c >> 3, ch = c & 7;
            uint4 v = *reinterpret_cast<const uint4*>(K + (size_t)(k0 + row) * 64 + ch * 8);
            rk[it] = (k0 + row < N) ? v : make_uint4(0, 0, 0, 0);
            union { uint4 u; uint16_t e[8]; } w;
            w.u = *reinterpret_cast<const uint4*>(Vt + (size_t)row * Npad + k0 + ch * 8);
            const int kb = k0 + ch * 8;
            if (kb + 8 > N) {
#pragma unroll
                for (int e = 0; e < 8; ++e)
                    if (kb + e >= N) w.e[e] = 0;
            }
            rv[it] = w.u;
        }
        if (tid < 64) radd = (k0 + tid < N) ? kadd[k0 + tid] : -INFINITY;
    };
    auto store_tile = [&](int buf) __attribute__((always_inline)) {
        char* sK = smem + buf * BUF;
        char* sV = sK + K_BYTES;
        float* sA = reinterpret_cast<float*>(sV + V_BYTES);
#pragma unroll
        for (int it = 0; it < CPT; ++it) {
            const int c = tid + it * NT, row = c >> 3, ch = c & 7;
            *reinterpret_cast<uint4*>(sK + swz128(row, ch)) = rk[it];
            *reinterpret_cast<uint2*>(sV + swz64(row, 2 * ch)) = make_uint2(rv[it].x, rv[it].y);
            *reinterpret_cast<uint2*>(sV + swz64(row, 2 * ch + 1)) = make_uint2(rv[it].z, rv[it].w);
        }
        if (tid < 64) sA[tid] = radd;
    };

    f32x16 o[2];
#pragma unroll
    for (int r = 0; r < 16; ++r) { o[0][r] = 0.f; o[1][r] = 0.f; }
    float m_run = -INFINITY, l_run = 0.f;

    const int nt = (N + 63) >> 6;
    load_tile(0);
    store_tile(0);
    __syncthreads();
    for (int j = 0; j < nt; ++j) {
        const int buf = j & 1;
        if (j + 1 < nt) load_tile(j + 1);
        const char* sK = smem + buf * BUF;
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
        if (j + 1 < nt) store_tile(buf ^ 1);
        __syncthreads();
    }

    const float l_tot = l_run + __shfl_xor(l_run, 32, 64);
    const float inv = 1.0f / l_tot;
    if (qrow < N) {
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

hipError_t launch_attention(const AttnParams& p, hipStream_t s) {
    if (p.N <= 0 || p.Npad % 64 != 0 || p.Npad < ((p.N + 63) / 64) * 64) return hipErrorInvalidValue;
    // waves per workgroup: share K/V tiles across 4 waves once there is enough work to fill the chip
    const long wg4 = (long)((p.N + 127) / 128) * p.H * p.B;
    if (wg4 >= 512) {
        g_last_kernel = "attn_kernel<4>";
        hipLaunchKernelGGL(attn_kernel<4>, dim3((p.N + 127) / 128, p.H, p.B), dim3(256), 0, s, p);
    } else {
        g_last_kernel = "attn_kernel<1>";
        hipLaunchKernelGGL(attn_kernel<1>, dim3((p.N + 31) / 32, p.H, p.B), dim3(64), 0, s, p);
    }
    return hipGetLastError();
}

}  // namespace uvl
