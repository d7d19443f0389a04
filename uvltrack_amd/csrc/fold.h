// Device helpers of the LayerNorm-free one-sequence frame (round 6): row statistics as per-32-column partials of the f32 rows, and the
// contrastive-logits job as extra workgroups of a GEMM launch.
//
// The consumer GEMM computes  rstd * (sum_k a~_k W'_nk - mean * colsum_n) + b'_n  with a~ = bf16(a), W' = bf16(W gamma), colsum_n = sum_k W'_nk, b' = b + W beta
// = LayerNorm(a) W'^T + b' + rstd (a~ - a) W'^T  (block.py:30-31,42; bert_backbone.py:231-244): against the LayerNorm-kernel path, which rounds the NORMALISED row,
// the operand error is the bf16 rounding of a (relative 2^-9 per element) -- the same size.  mean / rstd are those of the f32 row (the partials are taken before
// the rounding), so the finishing GEMMs can also rebuild BERT's post-LayerNorm residual from them to f32 accuracy; that the centring uses the mean of a and not of
// a~ costs |mean(a~) - mean(a)| * colsum * rstd ~ 1e-4 even on rows with 20x outlier channels (tests/test_fold_gpu.py).
#pragma once
#include "common.h"
#include "kernels.h"

namespace uvl {

// Offset (floats) of partial pair j of row `row` in an array of `rows` rows: plane PAIRS -- [j / 2][row][j % 2][sum, sum of squares] -- so that a consumer lane reads the
// partials (2 i, 2 i + 1) of its row as ONE 16-byte load and the 32 lanes of a half wave (consecutive rows) read 512 contiguous bytes
__host__ __device__ __forceinline__ size_t st_off(int j, size_t row, size_t rows) { return ((size_t)(j >> 1) * rows + row) * 4 + (size_t)(j & 1) * 2; }

// sum over the 8 lanes of an aligned octet (lanes 8j .. 8j+7), every lane of the octet gets it: xor 1, xor 2 as quad permutes, then the mirror of the half row
__device__ __forceinline__ float oct_sum(float v) {
#if __HIP_DEVICE_COMPILE__
    v += __uint_as_float(__builtin_amdgcn_update_dpp(0u, __float_as_uint(v), 0xB1, 0xf, 0xf, false));      // quad_perm [1,0,3,2]
    v += __uint_as_float(__builtin_amdgcn_update_dpp(0u, __float_as_uint(v), 0x4E, 0xf, 0xf, false));      // quad_perm [2,3,0,1]
    v += __uint_as_float(__builtin_amdgcn_update_dpp(0u, __float_as_uint(v), 0x141, 0xf, 0xf, false));     // row_half_mirror: lane i <-> 7 - i of the octet
#endif
    return v;
}

// (sum, sum of squares) of the four f32 values of a lane (before they are rounded to bf16)
__device__ __forceinline__ void st_of4(float r0, float r1, float r2, float r3, float& s1, float& s2) {
    s1 = (r0 + r1) + (r2 + r3);
    s2 = (r0 * r0 + r1 * r1) + (r2 * r2 + r3 * r3);
}

// mean / rstd from the summed partials of a row of D values
__device__ __forceinline__ void st_finish(float s1, float s2, int D, float eps, float& mean, float& rstd) {
    const float inv = 1.0f / (float)D;
    mean = s1 * inv;
    const float var = fmaxf(s2 * inv - mean * mean, 0.f);
    rstd = 1.0f / sqrtf(var + eps);
}

// One wave = one search row of the logits job (extractor.py:85-93): tau * normalize(x_s) . normalize(token), select [vis, txt, mean][flag] -- the arithmetic of
// contrast_kernel / ln_body's second job.  blk = index of the job's workgroup (4 rows each).
__device__ __forceinline__ void ct_job_block(const CtJob& j, const int blk, const float* zero_row) {
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const int r = blk * 4 + wave;
    if (r >= j.B * j.nx || wave >= 4) return;
    const int b = r / j.nx, s = r - b * j.nx;
    const int D = j.D;
    const float* xb = j.x + (size_t)b * j.xbs * D;
    const float* xs = xb + (size_t)(1 + j.nz + s) * D;
    const float* tk = j.skip_text ? xb : (j.txt ? j.txt + (size_t)b * j.txt_bs * D : xb + (size_t)j.nv * D);
    const float* sv = j.sub_vis ? j.sub_vis : zero_row;
    const float* sq = (j.sub_txt && !j.txt) ? j.sub_txt : zero_row;
    const float ls = j.logit_scale[0];
    const int fl = (int)j.flag[b];
    // the text token of a pre-fusion layer is still PRE-norm in a LayerNorm-free frame: normalise it here, with the statistics the finishing GEMMs use for the same row
    // (the row's partials, summed in the same order: gemm_fin_body) -- the value equals, bit for bit, the snapshot the next attention.output GEMM leaves
    // of this row, which is what a frame that reuses the text branch reads instead
    float tmean = 0.f, trstd = 1.f;
    const bool tln = j.txt_g != nullptr && !j.skip_text;
    if (tln) {
        const int np = D >> 5, c16 = lane & 7;
        const size_t srow = (size_t)b * j.txt_st_bs;
        float2 rs[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const int jj = c16 + 8 * k;
            rs[k] = jj < np ? *reinterpret_cast<const float2*>(j.txt_st + st_off(jj, srow, (size_t)j.txt_st_rows)) : make_float2(0.f, 0.f);
        }
        float s1 = (rs[0].x + rs[1].x) + (rs[2].x + rs[3].x), s2 = (rs[0].y + rs[1].y) + (rs[2].y + rs[3].y);
        s1 = oct_sum(s1);
        s2 = oct_sum(s2);
        st_finish(s1, s2, D, j.txt_eps, tmean, trstd);
    }
    const float* tg = tln ? j.txt_g : zero_row;
    const float* tb = tln ? j.txt_b : zero_row;
    float xx = 0.f, xv = 0.f, vv = 0.f, xt = 0.f, tt = 0.f;
    for (int c = lane * 4; c < D; c += 256) {
        float4 a = *reinterpret_cast<const float4*>(xs + c), v = *reinterpret_cast<const float4*>(xb + c), q = *reinterpret_cast<const float4*>(tk + c);
        const float4 m0 = *reinterpret_cast<const float4*>(sv + (j.sub_vis ? c : 0)), m1 = *reinterpret_cast<const float4*>(sq + ((j.sub_txt && !j.txt) ? c : 0));
        if (tln) {
            const float4 g = *reinterpret_cast<const float4*>(tg + c), be = *reinterpret_cast<const float4*>(tb + c);
            q.x = (q.x - tmean) * trstd * g.x + be.x; q.y = (q.y - tmean) * trstd * g.y + be.y;
            q.z = (q.z - tmean) * trstd * g.z + be.z; q.w = (q.w - tmean) * trstd * g.w + be.w;
        }
        a.x -= m0.x; a.y -= m0.y; a.z -= m0.z; a.w -= m0.w;
        v.x -= m0.x; v.y -= m0.y; v.z -= m0.z; v.w -= m0.w;
        q.x -= m1.x; q.y -= m1.y; q.z -= m1.z; q.w -= m1.w;
        xx += a.x * a.x + a.y * a.y + a.z * a.z + a.w * a.w;
        xv += a.x * v.x + a.y * v.y + a.z * v.z + a.w * v.w;
        vv += v.x * v.x + v.y * v.y + v.z * v.z + v.w * v.w;
        if (!j.skip_text) {
            xt += a.x * q.x + a.y * q.y + a.z * q.z + a.w * q.w;
            tt += q.x * q.x + q.y * q.y + q.z * q.z + q.w * q.w;
        }
    }
    const float tau = __expf(ls);
    xx = fmaxf(sqrtf(wave_sum(xx)), 1e-12f);
    vv = fmaxf(sqrtf(wave_sum(vv)), 1e-12f);
    const float lv = tau * wave_sum(xv) / (xx * vv);
    float lt = 0.f;
    if (!j.skip_text) {
        tt = fmaxf(sqrtf(wave_sum(tt)), 1e-12f);
        lt = tau * wave_sum(xt) / (xx * tt);
    }
    const float out = fl == 0 ? lv : (fl == 1 ? lt : 0.5f * (lv + lt));
    if (lane == 0) j.logits[((size_t)b * j.ncont + j.slot) * j.nx + s] = out;
}

}  // namespace uvl
