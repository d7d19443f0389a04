// The prompter: DistributionBasedCrossAttention (reference lib/models/heads/utils.py:23-99), reached through
// UVLTrack.forward_prompt_init / forward_prompt (uvltrack.py:26-38, head:96-106).  It runs at sequence init and every
// UPDATE_INTERVAL frames, not per frame ("next" row 1 of SURVEY.md section 8f).
//
// One workgroup per sample does everything up to the MLP: cosine-similarity logits of the flag-selected token against
// the L = nz + S template/context tokens, the target / background softmaxes, the sort + cumulative-mass split of the
// background into "pure background" (lowest 25 % of the mass) and distractors (utils.py:45-56), the three masked
// softmaxes and the three probability-weighted token sums.  The MLP (768 -> 3072 -> 768 + residual) reuses the MFMA GEMM.
#include "common.h"
#include "kernels.h"

namespace uvl {

#define PR_MAXL 1024
#define PR_NEG (-1e20f)

__device__ __forceinline__ float block_reduce(float v, float* red, bool is_max) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    v = is_max ? wave_max(v) : wave_sum(v);
    __syncthreads();
    if (lane == 0) red[wave] = v;
    __syncthreads();
    float r = red[0];
    for (int w = 1; w < 4; ++w) r = is_max ? fmaxf(r, red[w]) : r + red[w];
    return r;
}

// softmax weights of logits a[l] over l < L (all threads cooperate); w[l] = exp(a - max) / sum
__device__ __forceinline__ void block_softmax(const float* a, float* w, int L, float* red) {
    float mx = -INFINITY;
    for (int l = threadIdx.x; l < L; l += 256) mx = fmaxf(mx, a[l]);
    mx = block_reduce(mx, red, true);
    float sum = 0.f;
    for (int l = threadIdx.x; l < L; l += 256) {
        const float e = __expf(a[l] - mx);
        w[l] = e;
        sum += e;
    }
    sum = block_reduce(sum, red, false);
    const float inv = 1.0f / sum;
    for (int l = threadIdx.x; l < L; l += 256) w[l] *= inv;
    __syncthreads();
}

__global__ __launch_bounds__(256) void prompter_tokens_kernel(const PrompterParams p) {
    __shared__ float token[1024];
    __shared__ float sim[PR_MAXL], wt[PR_MAXL], wd[PR_MAXL], wb[PR_MAXL], tmp[PR_MAXL], srt[PR_MAXL];
    __shared__ uint8_t tmk[PR_MAXL];
    __shared__ float red[4];
    __shared__ float thr_s;
    const int b = blockIdx.x, D = p.D, L = p.nz + p.S;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int fl = (int)p.flag[b];
    const float* tem = p.tem + (size_t)b * p.nz * D;
    const float* ctx = p.ctx + (size_t)((b + p.ctx_roll) % p.B) * p.S * D;

    // token = [vis, txt, (vis+txt)/2][flag]  (head:97-101)
    float tn = 0.f;
    for (int c = threadIdx.x; c < D; c += 256) {
        const float v = p.vis[(size_t)b * D + c], t = p.txt[(size_t)b * D + c];
        const float k = fl == 0 ? v : (fl == 1 ? t : (v + t) / 2.0f);
        token[c] = k;
        tn += k * k;
    }
    tn = fmaxf(sqrtf(block_reduce(tn, red, false)), 1e-12f);
    for (int l = threadIdx.x; l < L; l += 256) tmk[l] = l < p.nz ? p.tem_mask[(size_t)b * p.nz + l] : p.ctx_mask[(size_t)b * p.S + (l - p.nz)];
    __syncthreads();
    // similarity logits (utils.py:90): tau * normalize(token) . normalize(tgt_l)
    const float tau = __expf(p.logit_scale[0]);
    for (int l = wave; l < L; l += 4) {
        const float* row = l < p.nz ? tem + (size_t)l * D : ctx + (size_t)(l - p.nz) * D;
        float dot = 0.f, nn = 0.f;
        for (int c = lane * 4; c < D; c += 256) {
            const float4 a = *reinterpret_cast<const float4*>(row + c);
            const float4 t = *reinterpret_cast<const float4*>(token + c);
            dot += a.x * t.x + a.y * t.y + a.z * t.z + a.w * t.w;
            nn += a.x * a.x + a.y * a.y + a.z * a.z + a.w * a.w;
        }
        dot = wave_sum(dot);
        nn = fmaxf(sqrtf(wave_sum(nn)), 1e-12f);
        if (lane == 0) sim[l] = tau * dot / (nn * tn);
    }
    __syncthreads();
    // target softmax and background softmax (utils.py:60-66)
    for (int l = threadIdx.x; l < L; l += 256) tmp[l] = tmk[l] ? sim[l] : PR_NEG;
    __syncthreads();
    block_softmax(tmp, wt, L, red);
    for (int l = threadIdx.x; l < L; l += 256) tmp[l] = tmk[l] ? PR_NEG : sim[l];
    __syncthreads();
    block_softmax(tmp, wb, L, red);              // wb = bgd_score for now
    // divide_background (utils.py:45-56): ascending sort, threshold = first value whose running mass reaches 0.25
    for (int l = threadIdx.x; l < PR_MAXL; l += 256) srt[l] = l < L ? wb[l] : INFINITY;
    __syncthreads();
    for (int k = 2; k <= PR_MAXL; k <<= 1)
        for (int j = k >> 1; j > 0; j >>= 1) {
            for (int i = threadIdx.x; i < PR_MAXL; i += 256) {
                const int ixj = i ^ j;
                if (ixj > i) {
                    const float a = srt[i], c = srt[ixj];
                    const bool up = (i & k) == 0;
                    if ((a > c) == up) { srt[i] = c; srt[ixj] = a; }
                }
            }
            __syncthreads();
        }
    if (threadIdx.x == 0) {
        float cum = 0.f, thr = 1.0f;
        for (int l = 0; l < L; ++l) {
            cum += srt[l];
            if (!(cum < 0.25f)) { thr = srt[l]; break; }
        }
        thr_s = thr;
    }
    __syncthreads();
    const float thr = thr_s;
    // distractor / pure-background softmaxes (utils.py:72-73); tmp still holds bgd_logit
    for (int l = threadIdx.x; l < L; l += 256) {
        const bool dis = wb[l] >= thr;
        const float bl = tmp[l];
        srt[l] = dis ? bl : PR_NEG;              // dis logits
        sim[l] = dis ? PR_NEG : bl;              // pure background logits
    }
    __syncthreads();
    block_softmax(srt, wd, L, red);
    block_softmax(sim, wb, L, red);
    // probability-weighted sums over the L tokens: columns across threads, rows in the loop
    for (int c = threadIdx.x; c < D; c += 256) {
        float at = 0.f, ad = 0.f, ab = 0.f;
#pragma unroll 4
        for (int l = 0; l < L; ++l) {
            const float x = l < p.nz ? tem[(size_t)l * D + c] : ctx[(size_t)(l - p.nz) * D + c];
            at += wt[l] * x;
            ad += wd[l] * x;
            ab += wb[l] * x;
        }
        // src = [tgt, dis, bgd] + src_,  src_ = query_embed (+ token on row 0)   (utils.py:83-84,93)
        const float s0 = p.query_embed[c] + token[c], s1 = p.query_embed[D + c], s2 = p.query_embed[2 * D + c];
        float* so = p.src0 + (size_t)b * 3 * D;
        so[c] = s0; so[D + c] = s1; so[2 * D + c] = s2;
        float* sr = p.src + (size_t)b * 3 * D;
        const float v0 = at + s0, v1 = ad + s1, v2 = ab + s2;
        sr[c] = v0; sr[D + c] = v1; sr[2 * D + c] = v2;
        bf16_t* sb = p.src_bf16 + (size_t)b * 3 * D;
        sb[c] = f2bf(v0); sb[D + c] = f2bf(v1); sb[2 * D + c] = f2bf(v2);
    }
}

hipError_t launch_prompter_tokens(const PrompterParams& p, hipStream_t s) {
    if (p.nz + p.S > PR_MAXL || p.D > 1024 || p.D % 4) return hipErrorInvalidValue;
    hipLaunchKernelGGL(prompter_tokens_kernel, dim3(p.B), dim3(256), 0, s, p);
    return hipGetLastError();
}

// the "switcher" (utils.py:95-97): [src, src_, src][flag] -- grounding (flag 1) keeps the un-updated queries
__global__ __launch_bounds__(256) void prompter_select_kernel(const float* __restrict__ src, const float* __restrict__ src0,
                                                              const int64_t* __restrict__ flag, float* __restrict__ out, int n_per_sample) {
    const int b = blockIdx.y;
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= n_per_sample) return;
    const size_t at = (size_t)b * n_per_sample + i;
    out[at] = ((int)flag[b] == 1) ? src0[at] : src[at];
}

hipError_t launch_prompter_select(const float* src, const float* src0, const int64_t* flag, float* out, int B, int n_per_sample, hipStream_t s) {
    hipLaunchKernelGGL(prompter_select_kernel, dim3((n_per_sample + 255) / 256, B), dim3(256), 0, s, src, src0, flag, out, n_per_sample);
    return hipGetLastError();
}

// Target-cell masks of the prompter (tracker anno2mask, lib/test/tracker/uvltrack.py:183-194): cell (i, j) of a size x size grid is
// set when its centre (j + 0.5, i + 0.5) lies strictly inside the box (x, y, w, h normalised, scaled by size in f32 as the reference
// does), and the cell that holds the box centre is always set.  One thread per cell; boxes stay on the device (the tracker's
// updated box comes out of the decode kernel).
__global__ __launch_bounds__(256) void anno2mask_kernel(const float* __restrict__ boxes, int B, int size, uint8_t* __restrict__ mask) {
    const int idx = blockIdx.x * 256 + threadIdx.x;
    const int S = size * size;
    if (idx >= B * S) return;
    const int b = idx / S, c = idx - b * S, i = c / size, j = c - i * size;
    const float x = boxes[4 * b], y = boxes[4 * b + 1], w = boxes[4 * b + 2], h = boxes[4 * b + 3];
    const float fs = (float)size;
    const float x1 = x * fs, y1 = y * fs, x2 = (x + w) * fs, y2 = (y + h) * fs;
    const float cj = (float)j + 0.5f, ci = (float)i + 0.5f;
    bool m = cj > x1 && cj < x2 && ci > y1 && ci < y2;
    const long long cx = (long long)((x1 + x2) / 2.0f), cy = (long long)((y1 + y2) / 2.0f);      // .long(): truncation
    m = m || (cx == j && cy == i);
    mask[idx] = m ? 1 : 0;
}
hipError_t launch_anno2mask(const float* boxes, int B, int size, uint8_t* mask, hipStream_t s) {
    const int n = B * size * size;
    hipLaunchKernelGGL(anno2mask_kernel, dim3((n + 255) / 256), dim3(256), 0, s, boxes, B, size, mask);
    return hipGetLastError();
}

}  // namespace uvl
