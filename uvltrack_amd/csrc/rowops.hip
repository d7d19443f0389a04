// Row-wise (HBM/L2-bound) kernels of the frame: LayerNorm, patch gather, BERT embedding, mask setup,
// contrastive logits, head prologue/epilogue and the weight packers.  One wave64 per token row,
// 16-byte vector loads, wave shuffles for the reductions.
#include <cstdio>
#include <cstdlib>
#include "common.h"
#include "kernels.h"
#include "ln_body.h"
#include "fold.h"

namespace uvl {

// ------------------------------------------------------------------------------------------------
// LayerNorm (block.py:30-31 with eps 1e-6; BertLayerNorm bert_backbone.py:231-244 with eps 1e-12).
// Optionally adds a per-row-type vector first and writes it back: that is the permanent
// `img_feat + modal_embed[0]`, `txt_feat + modal_embed[1]` of forward_joint (mae_vit.py:196).
// ------------------------------------------------------------------------------------------------
// (LN_MAX_SLABS, g_zero_row, sel4 and ln_body live in ln_body.h: the fused LayerNorm + GEMM kernel in gemm.hip shares them)

template <int NV, bool FULL, bool SLABS, int CT>
__global__ __launch_bounds__(256) void ln_kernel(const LnParams p) {
    kernarg_warm<sizeof(LnParams) + 64>(); ln_body<NV, FULL, SLABS, CT>(p, blockIdx.x); }

// Two independent LayerNorm problems in one launch (batch-1 frames pair every text-branch kernel with the visual kernel
// of the same kind, see uvl_api.hip): workgroups [0, split) work on pa, the rest on pb.
// SLABS_B: the RIDER alone folds split-K slabs (round 5: the text branch's output GEMM of a many-sequence frame runs as two K halves, the visual rows
// stay in place -- the HBM-bound visual LayerNorm must not pay the slab operand slots); SLABS covers both problems otherwise.
template <int NV, bool FULL, bool SLABS, int CT, bool SLABS_B = SLABS>
__global__ __launch_bounds__(256) void ln_pair_kernel(const LnParams pa, const LnParams pb, int split) {
    kernarg_warm<2 * sizeof(LnParams) + 8 + 64>();      // (+ 64: the kernel reads gridDim -- the first line of the hidden arguments)
    // two calls, not a selected reference: selecting between the two by-value argument blocks would copy one into scratch
    if constexpr (SLABS_B && !SLABS) {
        // the rider's rows fold slabs (more loads per row than the visual rows): its workgroups come FIRST, or they are the launch's tail
        // (rocprofv3, 8 UVLTrack-L sequences: 14.1 us with the rider behind the visual rows against 10.2 us for the launch without slabs)
        const int nb = (int)gridDim.x - split;
        if ((int)blockIdx.x < nb) ln_body<NV, FULL, true, 0>(pb, (int)blockIdx.x);
        else ln_body<NV, FULL, false, CT>(pa, (int)blockIdx.x - nb);
    } else {
        if ((int)blockIdx.x < split) ln_body<NV, FULL, SLABS, CT>(pa, (int)blockIdx.x);
        else ln_body<NV, FULL, SLABS_B, 0>(pb, (int)blockIdx.x - split);      // the rider is a text-branch LayerNorm: never a contrast job
    }
}

// Rows (= waves) per workgroup.  With one memory round trip per row, one sequence of UVLTrack-B (553 rows) is 1.5-2.6 % faster in
// the frame with one row per workgroup (553 workgroups over the 256 CUs instead of 139: 1231-1243 vs 1212-1214 frames/s);
// from 873 rows on (UVLTrack-L, two or more sequences) 1 and 4 measure the same.
static int ln_waves_per_block(int M) { return M <= 768 ? 1 : 4; }

// name of the instantiation as rocprofv3 prints it (bench.py keys its per-kernel rooflines and the committed PMC traffic on it)
static const char* ln_name(bool pair, int nv, bool full, bool slabs, int ct) {
    static char names[2][5][2][2][3][40];              // interned: the profiler keeps the pointer
    char* name = names[pair][nv & 7 ? (nv > 4 ? 4 : nv) : 0][full][slabs][ct];
    if (!name[0]) {
        if (pair) snprintf(name, 40, "ln_pair_kernel<%d,%d,%d,%d,%d>", nv, (int)full, (int)slabs, (int)ct, (int)slabs);      // (SLABS_B defaults to SLABS)
        else snprintf(name, 40, "ln_kernel<%d,%d,%d,%d>", nv, (int)full, (int)slabs, (int)ct);
    }
    return name;
}

template <int NV, bool FULL>
static void launch_ln_variant(const LnParams& p, int grid, int wpb, hipStream_t s) {
    const bool slabs = p.nsplit > 0, ct = p.ct_x != nullptr;
    const bool self = ct && p.ct_self && !slabs;         // (a self job never comes with slabs: the launch must leave its input rows alone)
    g_last_kernel = ln_name(false, NV, FULL, slabs, self ? 2 : ct);
    if (self) hipLaunchKernelGGL((ln_kernel<NV, FULL, false, 2>), dim3(grid), dim3(64 * wpb), 0, s, p);
    else if (slabs && ct) hipLaunchKernelGGL((ln_kernel<NV, FULL, true, true>), dim3(grid), dim3(64 * wpb), 0, s, p);
    else if (slabs) hipLaunchKernelGGL((ln_kernel<NV, FULL, true, false>), dim3(grid), dim3(64 * wpb), 0, s, p);
    else if (ct) hipLaunchKernelGGL((ln_kernel<NV, FULL, false, true>), dim3(grid), dim3(64 * wpb), 0, s, p);
    else hipLaunchKernelGGL((ln_kernel<NV, FULL, false, false>), dim3(grid), dim3(64 * wpb), 0, s, p);
}

hipError_t launch_layernorm(const LnParams& p_in, hipStream_t s) {
    LnParams p = p_in;
    p.fd_rpb = fastdiv_of((uint32_t)p.rpb);
    if (p.D % 4 != 0 || p.D > 1024 || p.M <= 0 || p.nsplit > LN_MAX_SLABS || (p.ct_x && p.ct_self && (p.nsplit > 0 || p.ct_x != p.x))) return hipErrorInvalidValue;
    const int wpb = ln_waves_per_block(p.M);
    const int grid = (p.M + wpb - 1) / wpb;
    if (p.D == 768) launch_ln_variant<3, true>(p, grid, wpb, s);
    else if (p.D == 1024) launch_ln_variant<4, true>(p, grid, wpb, s);
    else if (p.D <= 256) launch_ln_variant<1, false>(p, grid, wpb, s);
    else if (p.D <= 768) launch_ln_variant<3, false>(p, grid, wpb, s);
    else launch_ln_variant<4, false>(p, grid, wpb, s);
    return hipGetLastError();
}

template <int NV, bool FULL>
static void launch_ln_pair_variant(const LnParams& a, const LnParams& b, int ga, int gb, int wpb, hipStream_t s) {
    const bool slabs = a.nsplit > 0 || b.nsplit > 0, ct = a.ct_x != nullptr;
    const bool self = ct && a.ct_self && !slabs;
    if (NV >= 3 && FULL && a.nsplit == 0 && b.nsplit > 0 && !(ct && !a.ct_self)) {      // only the rider has slabs (names as rocprofv3 prints them: "...,0,<ct>,1>")
        static char nm[2][40];
        const int c2 = (ct && a.ct_self) ? 1 : 0;
        if (!nm[c2][0]) snprintf(nm[c2], 40, "ln_pair_kernel<%d,1,0,%d,1>", NV, c2 ? 2 : 0);
        g_last_kernel = nm[c2];
        if (c2) hipLaunchKernelGGL((ln_pair_kernel<NV, FULL, false, 2, true>), dim3(ga + gb), dim3(64 * wpb), 0, s, a, b, ga);
        else hipLaunchKernelGGL((ln_pair_kernel<NV, FULL, false, 0, true>), dim3(ga + gb), dim3(64 * wpb), 0, s, a, b, ga);
        return;
    }
    g_last_kernel = ln_name(true, NV, FULL, slabs, self ? 2 : ct);
    if (self) hipLaunchKernelGGL((ln_pair_kernel<NV, FULL, false, 2>), dim3(ga + gb), dim3(64 * wpb), 0, s, a, b, ga);
    else if (slabs && ct) hipLaunchKernelGGL((ln_pair_kernel<NV, FULL, true, true>), dim3(ga + gb), dim3(64 * wpb), 0, s, a, b, ga);
    else if (slabs) hipLaunchKernelGGL((ln_pair_kernel<NV, FULL, true, false>), dim3(ga + gb), dim3(64 * wpb), 0, s, a, b, ga);
    else if (ct) hipLaunchKernelGGL((ln_pair_kernel<NV, FULL, false, true>), dim3(ga + gb), dim3(64 * wpb), 0, s, a, b, ga);
    else hipLaunchKernelGGL((ln_pair_kernel<NV, FULL, false, false>), dim3(ga + gb), dim3(64 * wpb), 0, s, a, b, ga);
}

hipError_t launch_layernorm_pair(const LnParams& a_in, const LnParams& b_in, hipStream_t s) {
    LnParams a = a_in, b = b_in;
    a.fd_rpb = fastdiv_of((uint32_t)a.rpb);
    b.fd_rpb = fastdiv_of((uint32_t)b.rpb);
    if (a.D != b.D || a.D % 4 != 0 || a.D > 1024 || a.M <= 0 || b.M <= 0 || a.nsplit > LN_MAX_SLABS || b.nsplit > LN_MAX_SLABS || b.ct_x ||
        (a.ct_x && a.ct_self && (a.nsplit > 0 || a.ct_x != a.x)))
        return hipErrorInvalidValue;
    // a direct logits job (ct_self: ln_body's CT = 2 form) beside a rider that folds slabs exists only as the rider-only-slabs kernel of the D = 768 / 1024
    // instantiations (launch_ln_pair_variant's first branch); any other width would fall through to ln_pair_kernel<.., true, CT = 1> on a ct_self job: wrong logits
    if (a.ct_x && a.ct_self && b.nsplit > 0 && !(a.D == 768 || a.D == 1024)) return hipErrorInvalidValue;
    const int wpb = ln_waves_per_block(a.M);
    const int ga = (a.M + wpb - 1) / wpb, gb = (b.M + wpb - 1) / wpb;
    if (a.D == 768) launch_ln_pair_variant<3, true>(a, b, ga, gb, wpb, s);
    else if (a.D == 1024) launch_ln_pair_variant<4, true>(a, b, ga, gb, wpb, s);
    else if (a.D <= 256) launch_ln_pair_variant<1, false>(a, b, ga, gb, wpb, s);
    else if (a.D <= 768) launch_ln_pair_variant<3, false>(a, b, ga, gb, wpb, s);
    else launch_ln_pair_variant<4, false>(a, b, ga, gb, wpb, s);
    return hipGetLastError();
}

// ------------------------------------------------------------------------------------------------
// im2row for PatchEmbed (mae_vit.py:94-100): Conv2d(3, D, k16, s16) == GEMM over (c,kh,kw) patch vectors.
// One thread moves one 16-pixel patch line: 64 B of f32 in, 32 B of bf16 out.
// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ void im2row_body(const float* __restrict__ z, const float* __restrict__ x,
                                            bf16_t* __restrict__ out, int B, int Hz, int Hx, int bx) {
    const int gz = Hz / 16, gx = Hx / 16, nz = gz * gz, nx = gx * gx, per = nz + nx;
    const long gid = (long)bx * 256 + threadIdx.x;
    const long total = (long)B * per * 48;
    if (gid >= total) return;
    const int line = gid % 48;                  // c*16 + kh
    const long row = gid / 48;
    const int c = line >> 4, kh = line & 15;
    const int b = row / per, t = row - (long)b * per;
    const float* src;
    if (t < nz) {
        const int i = t / gz, j = t - i * gz;
        src = z + (((size_t)b * 3 + c) * Hz + i * 16 + kh) * Hz + j * 16;
    } else {
        const int tt = t - nz, i = tt / gx, j = tt - i * gx;
        src = x + (((size_t)b * 3 + c) * Hx + i * 16 + kh) * Hx + j * 16;
    }
    uint32_t w[8];
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const float4 f = *reinterpret_cast<const float4*>(src + 4 * q);
        w[2 * q] = pack_bf16x2(f.x, f.y);
        w[2 * q + 1] = pack_bf16x2(f.z, f.w);
    }
    uint4* dst = reinterpret_cast<uint4*>(out + (size_t)row * 768 + line * 16);
    dst[0] = make_uint4(w[0], w[1], w[2], w[3]);
    dst[1] = make_uint4(w[4], w[5], w[6], w[7]);
}

__global__ __launch_bounds__(256) void im2row_kernel(const float* __restrict__ z, const float* __restrict__ x,
                                                     bf16_t* __restrict__ out, int B, int Hz, int Hx) {
    im2row_body(z, x, out, B, Hz, Hx, blockIdx.x);
}

hipError_t launch_im2row(const float* z, const float* x, bf16_t* out, int B, int Hz, int Hx, hipStream_t s) {
    const long total = (long)B * ((Hz / 16) * (Hz / 16) + (Hx / 16) * (Hx / 16)) * 48;
    hipLaunchKernelGGL(im2row_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, s, z, x, out, B, Hz, Hx);
    return hipGetLastError();
}

// ------------------------------------------------------------------------------------------------
// BERT embedding: word[ids] + position[t] + token_type[0] -> LayerNorm(eps 1e-12)
// (bert_backbone.py:260-274).  Writes the f32 residual row and the bf16 GEMM operand.
// ------------------------------------------------------------------------------------------------
template <int NV>
__device__ __forceinline__ void bert_embed_body(const int64_t* __restrict__ ids, const float* __restrict__ word,
                                                const float* __restrict__ pos, const float* __restrict__ type0,
                                                const float* __restrict__ gamma, const float* __restrict__ beta,
                                                float* __restrict__ x, int xbs, int xro, bf16_t* __restrict__ y,
                                                int B, int T, int D, int vocab, int bx, float* __restrict__ raw_st = nullptr) {
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int m = bx * 4 + wave;
    if (m >= B * T) return;
    const int b = m / T, t = m - b * T;
    // the token id through the scalar cache (ids are < 2^31: the low word of the int64), then every row operand in one round trip
    long id = (long)(int)sload_u32(ids + m);
    id = id < 0 ? 0 : (id >= vocab ? vocab - 1 : id);
    const float* wr = word + (size_t)id * D;
    const float* pr = pos + (size_t)t * D;
    float4 v[NV], qa[NV], ty[NV], g[NV], be[NV];
    bool ok[NV];
#pragma unroll
    for (int i = 0; i < NV; ++i) {
        const int c0 = (lane + 64 * i) * 4;
        ok[i] = c0 < D;
        const int c = ok[i] ? c0 : 0;
        v[i] = *reinterpret_cast<const float4*>(wr + c);
        qa[i] = *reinterpret_cast<const float4*>(pr + c);
        ty[i] = *reinterpret_cast<const float4*>(type0 + c);
        g[i] = *reinterpret_cast<const float4*>(gamma + c);
        be[i] = *reinterpret_cast<const float4*>(beta + c);
    }
    __builtin_amdgcn_sched_barrier(0);
    float sum = 0.f;
#pragma unroll
    for (int i = 0; i < NV; ++i) {
        if (ok[i]) {
            v[i] = make_float4(v[i].x + qa[i].x + ty[i].x, v[i].y + qa[i].y + ty[i].y, v[i].z + qa[i].z + ty[i].z, v[i].w + qa[i].w + ty[i].w);
            sum += v[i].x + v[i].y + v[i].z + v[i].w;
        } else {
            v[i] = make_float4(0.f, 0.f, 0.f, 0.f);
        }
    }
    float* xr = x + ((size_t)b * xbs + xro + t) * D;
    if (raw_st) {
        // LayerNorm-free frame: the row stays pre-norm (f32 row, bf16 copy, per-32-column partials: lanes 8 j .. 8 j + 7 of chunk i hold block j + 8 i)
#pragma unroll
        for (int i = 0; i < NV; ++i) {
            const int c = (lane + 64 * i) * 4;
            const uint32_t lo = pack_bf16x2(v[i].x, v[i].y), hi = pack_bf16x2(v[i].z, v[i].w);
            float s1, s2;
            st_of4(v[i].x, v[i].y, v[i].z, v[i].w, s1, s2);
            s1 = oct_sum(s1);
            s2 = oct_sum(s2);
            if (ok[i]) {
                *reinterpret_cast<float4*>(xr + c) = v[i];
                *reinterpret_cast<uint2*>(y + (size_t)m * D + c) = uint2{lo, hi};
                if ((lane & 7) == 0) *reinterpret_cast<float2*>(raw_st + st_off(c >> 5, (size_t)m, (size_t)B * T)) = float2{s1, s2};
            }
        }
        return;
    }
    const float mean = wave_sum(sum) / (float)D;
    float sq = 0.f;
#pragma unroll
    for (int i = 0; i < NV; ++i) {
        if (ok[i]) {
            const float dx = v[i].x - mean, dy = v[i].y - mean, dz = v[i].z - mean, dw = v[i].w - mean;
            sq += dx * dx + dy * dy + dz * dz + dw * dw;
        }
    }
    const float rstd = 1.0f / sqrtf(wave_sum(sq) / (float)D + 1e-12f);
#pragma unroll
    for (int i = 0; i < NV; ++i) {
        const int c = (lane + 64 * i) * 4;
        if (ok[i]) {
            float4 o;
            o.x = (v[i].x - mean) * rstd * g[i].x + be[i].x;
            o.y = (v[i].y - mean) * rstd * g[i].y + be[i].y;
            o.z = (v[i].z - mean) * rstd * g[i].z + be[i].z;
            o.w = (v[i].w - mean) * rstd * g[i].w + be[i].w;
            *reinterpret_cast<float4*>(xr + c) = o;
            uint2 w;
            w.x = pack_bf16x2(o.x, o.y);
            w.y = pack_bf16x2(o.z, o.w);
            *reinterpret_cast<uint2*>(y + (size_t)m * D + c) = w;
        }
    }
}

template <int NV>
__global__ __launch_bounds__(256) void bert_embed_kernel(const int64_t* __restrict__ ids, const float* __restrict__ word,
                                                         const float* __restrict__ pos, const float* __restrict__ type0,
                                                         const float* __restrict__ gamma, const float* __restrict__ beta,
                                                         float* __restrict__ x, int xbs, int xro, bf16_t* __restrict__ y,
                                                         int B, int T, int D, int vocab) {
    bert_embed_body<NV>(ids, word, pos, type0, gamma, beta, x, xbs, xro, y, B, T, D, vocab, blockIdx.x);
}

hipError_t launch_bert_embed(const int64_t* ids, const float* word, const float* pos, const float* type0,
                             const float* gamma, const float* beta, float* x, int xbs, int xro, bf16_t* y_bf16,
                             int B, int T, int D, int vocab, hipStream_t s) {
    if (D % 4 != 0 || D > 1024) return hipErrorInvalidValue;
    const int grid = (B * T + 3) / 4;
    if (D <= 256) hipLaunchKernelGGL(bert_embed_kernel<1>, dim3(grid), dim3(256), 0, s, ids, word, pos, type0, gamma, beta, x, xbs, xro, y_bf16, B, T, D, vocab);
    else if (D <= 768) hipLaunchKernelGGL(bert_embed_kernel<3>, dim3(grid), dim3(256), 0, s, ids, word, pos, type0, gamma, beta, x, xbs, xro, y_bf16, B, T, D, vocab);
    else hipLaunchKernelGGL(bert_embed_kernel<4>, dim3(grid), dim3(256), 0, s, ids, word, pos, type0, gamma, beta, x, xbs, xro, y_bf16, B, T, D, vocab);
    return hipGetLastError();
}

// ------------------------------------------------------------------------------------------------
// cat_mask (extractor.py:43-50) as additive per-key terms, BERT's extended mask (bert_backbone.py:745-747),
// and the [cls] row of the residual stream (mae_vit.py:212-214).
//   key i: 0 = cls, [1,1+nz) template, [1+nz,nv) search, [nv,nj) text
//   flag==1 masks cls+template; search never; text where mask==0 or flag==0
// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ void setup_body(const uint8_t* __restrict__ tmask, const int64_t* __restrict__ flag,
                                           const float* __restrict__ cls_token, float* __restrict__ x,
                                           float* __restrict__ key_add, float* __restrict__ bert_add,
                                           int nz, int nv, int nj, int npad, int T, int D, int skip_text, int what, int b,
                                           bf16_t* __restrict__ cls_xn = nullptr, int cls_xn_bs = 0, float* __restrict__ cls_st = nullptr, int cls_st_rows = 0) {
    const int fl = (int)sload_u32(flag + b);                          // low word of the int64 flag
    // the text mask is read unconditionally at a clamped index (no branch around the load, no wait per key group)
    const uint8_t* tm = skip_text ? reinterpret_cast<const uint8_t*>(g_zero_row) : tmask + (size_t)b * T;
    if (what & 2) {
        for (int t = threadIdx.x; t < 64; t += 256) {
            const uint8_t mk = tm[t < T ? t : 0];
            bert_add[(size_t)b * 64 + t] = (t < T) ? (mk ? 0.f : -10000.f) : -INFINITY;
        }
    }
    if (!(what & 1)) return;
    for (int i = threadIdx.x; i < npad; i += 256) {
        const int ti = i - nv;
        const uint8_t mk = tm[ti < 0 ? 0 : (ti < T ? ti : T - 1)];
        float a;
        if (i < 1 + nz) a = (fl == 1) ? -1e10f : 0.f;                 // cls + template keys
        else if (i < nv) a = 0.f;                                     // search keys are never masked
        else if (i < nj) a = (skip_text || fl == 0 || mk == 0) ? -1e10f : 0.f;
        else a = -INFINITY;
        key_add[(size_t)b * npad + i] = a;
    }
    for (int c = threadIdx.x; c < D; c += 256) x[(size_t)b * nj * D + c] = cls_token[c];
    if (cls_xn) {                          // LayerNorm-free frame: the row as the first QKV GEMM reads it (bf16, un-normalised) + its partials (fold.h); D <= 1024
        const int c = threadIdx.x * 4;
        const bool ok = c < D;
        const float4 v = ok ? *reinterpret_cast<const float4*>(cls_token + c) : make_float4(0.f, 0.f, 0.f, 0.f);
        const uint32_t lo = pack_bf16x2(v.x, v.y), hi = pack_bf16x2(v.z, v.w);
        float s1, s2;
        st_of4(v.x, v.y, v.z, v.w, s1, s2);
        s1 = oct_sum(s1);
        s2 = oct_sum(s2);
        if (ok) {
            *reinterpret_cast<uint2*>(cls_xn + (size_t)b * cls_xn_bs * D + c) = uint2{lo, hi};
            if ((threadIdx.x & 7) == 0) *reinterpret_cast<float2*>(cls_st + st_off(c >> 5, (size_t)b * cls_xn_bs, (size_t)cls_st_rows)) = float2{s1, s2};
        }
    }
}

__global__ __launch_bounds__(256) void setup_kernel(const uint8_t* __restrict__ tmask, const int64_t* __restrict__ flag,
                                                    const float* __restrict__ cls_token, float* __restrict__ x,
                                                    float* __restrict__ key_add, float* __restrict__ bert_add,
                                                    int nz, int nv, int nj, int npad, int T, int D, int skip_text, int what) {
    setup_body(tmask, flag, cls_token, x, key_add, bert_add, nz, nv, nj, npad, T, D, skip_text, what, blockIdx.x);
}

// The three input-side kernels of a frame in one launch (single-stream frames): workgroups [0, n_setup) build the masks and
// the cls rows, the next n_embed do the BERT embedding + LayerNorm, the rest gather the image patches.
template <int NV>
__global__ __launch_bounds__(256) void prologue_kernel(const PrologueParams p) {
    kernarg_warm<sizeof(PrologueParams)>();
    const int bx = blockIdx.x;
    if (bx < p.n_setup) {
        setup_body(p.text_mask, p.flag, p.cls_token, p.x, p.key_add, p.bert_add, p.nz, p.nv, p.nj, p.npad, p.T, p.D, p.skip_text, p.setup_what, bx, p.cls_xn, p.cls_xn_bs, p.cls_st, p.cls_st_rows);
    } else if (bx < p.n_setup + p.n_embed) {
        bert_embed_body<NV>(p.ids, p.word, p.pos, p.type0, p.emb_g, p.emb_b, p.x, p.nj, p.nv, p.tn, p.B, p.T, p.D, p.vocab, bx - p.n_setup, p.embed_raw ? p.embed_st : nullptr);
    } else {
        im2row_body(p.z, p.ximg, p.patches, p.B, p.Hz, p.Hx, bx - p.n_setup - p.n_embed);
    }
}

hipError_t launch_prologue(const PrologueParams& q, hipStream_t s) {
    PrologueParams p = q;
    if (p.D % 4 != 0 || p.D > 1024) return hipErrorInvalidValue;
    const long total = (long)p.B * ((p.Hz / 16) * (p.Hz / 16) + (p.Hx / 16) * (p.Hx / 16)) * 48;
    p.n_setup = p.B;
    p.n_embed = p.ids ? (p.B * p.T + 3) / 4 : 0;
    const int n_im = (int)((total + 255) / 256);
    const dim3 grid(p.n_setup + p.n_embed + n_im);
    if (p.D <= 256) hipLaunchKernelGGL(prologue_kernel<1>, grid, dim3(256), 0, s, p);
    else if (p.D <= 768) hipLaunchKernelGGL(prologue_kernel<3>, grid, dim3(256), 0, s, p);
    else hipLaunchKernelGGL(prologue_kernel<4>, grid, dim3(256), 0, s, p);
    return hipGetLastError();
}

hipError_t launch_setup(const uint8_t* text_mask, const int64_t* flag, const float* cls_token, float* x,
                        float* key_add, float* bert_add, int B, int nz, int nv, int nj, int npad, int T, int D,
                        int skip_text, int what, hipStream_t s) {
    hipLaunchKernelGGL(setup_kernel, dim3(B), dim3(256), 0, s, text_mask, flag, cls_token, x, key_add, bert_add, nz, nv, nj, npad, T, D, skip_text, what);
    return hipGetLastError();
}

// ------------------------------------------------------------------------------------------------
// helpers: per-wave dot products over a D-vector
// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ float wave_dot(const float* a, const float* b, int D, int lane) {
    float s = 0.f;
    for (int c = lane * 4; c < D; c += 256) {
        const float4 x = *reinterpret_cast<const float4*>(a + c);
        const float4 y = *reinterpret_cast<const float4*>(b + c);
        s += x.x * y.x + x.y * y.y + x.z * y.z + x.w * y.w;
    }
    return wave_sum(s);
}

// txt token (generate_txt_token, extractor.py:79-83) into LDS: 'cls' = first text row, 'mean' = masked mean
__device__ __forceinline__ void block_txt_token(float* dst, const float* xtext, const uint8_t* tm, int T, int D, int mean_mode) {
    if (!mean_mode) {
        for (int c = threadIdx.x; c < D; c += blockDim.x) dst[c] = xtext[c];
    } else {
        float cnt = 0.f;
        for (int t = 0; t < T; ++t) cnt += tm[t] ? 1.f : 0.f;
        for (int c = threadIdx.x; c < D; c += blockDim.x) {
            float s = 0.f;
            for (int t = 0; t < T; ++t) s += tm[t] ? xtext[(size_t)t * D + c] : 0.f;
            dst[c] = s / cnt;
        }
    }
    __syncthreads();
}

// ------------------------------------------------------------------------------------------------
// Backbone contrastive logits for one layer (extractor.py:85-93):
//   tau * normalize(x) . normalize(tok) for tok in {vis_token, txt_token}; select [vis, txt, mean][flag].
// grid (ceil(nx/4), B), one wave per search token.  The residual stream may still carry pending split-K
// slabs of the layer's last GEMM (x_eff = x + sum_s part[s]); rows t < part_rows of each sample have them.
// ------------------------------------------------------------------------------------------------
struct RowView {
    const float* xb;          // sample base in the residual stream
    const float* part;        // slabs [nsplit][B*part_rows, D] or null
    int nsplit, part_rows, D;
    size_t part_stride, mbase;   // mbase = b * part_rows
    __device__ __forceinline__ float4 load(int t, int c) const {
        float4 v = *reinterpret_cast<const float4*>(xb + (size_t)t * D + c);
        const int nsp = (t < part_rows) ? nsplit : 0;
        float4 sl[LN_MAX_SLABS];
#pragma unroll
        for (int sp = 0; sp < LN_MAX_SLABS; ++sp)
            sl[sp] = (sp < nsp) ? *reinterpret_cast<const float4*>(part + (size_t)sp * part_stride + (mbase + t) * D + c)
                                : make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
        for (int sp = 0; sp < LN_MAX_SLABS; ++sp)
            if (sp < nsp) { v.x += sl[sp].x; v.y += sl[sp].y; v.z += sl[sp].z; v.w += sl[sp].w; }
        return v;
    }
};

__global__ __launch_bounds__(256) void contrast_kernel(const ContrastParams p) {
    kernarg_warm<sizeof(ContrastParams)>();
    extern __shared__ float sh[];              // [D] txt token, [D] vis token
    const int b = blockIdx.y, D = p.D;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    RowView rv{p.x + (size_t)b * p.nj * D, p.part, p.nsplit, p.part_rows, D, p.part_stride, (size_t)b * p.part_rows};
    const int fl = (int)p.flag[b];
    float* txt = sh;
    float* vis = sh + D;
    // stage the two tokens (generate_txt_token, extractor.py:79-83: 'cls' = first text row, 'mean' = masked mean)
    for (int c = threadIdx.x * 4; c < D; c += 1024) {
        *reinterpret_cast<float4*>(vis + c) = rv.load(0, c);
        if (!p.skip_text) {
            float4 tk;
            if (p.txt_snap) {                  // pre-fusion layer: text rows of THIS layer, snapshotted by the BERT LayerNorm
                const float* ts = p.txt_snap + (size_t)b * p.T * D;
                if (!p.mean_mode) {
                    tk = *reinterpret_cast<const float4*>(ts + c);
                } else {
                    tk = make_float4(0.f, 0.f, 0.f, 0.f);
                    float cnt = 0.f;
                    const uint8_t* tm = p.text_mask + (size_t)b * p.T;
                    for (int t = 0; t < p.T; ++t)
                        if (tm[t]) {
                            const float4 a = *reinterpret_cast<const float4*>(ts + (size_t)t * D + c);
                            tk.x += a.x; tk.y += a.y; tk.z += a.z; tk.w += a.w;
                            cnt += 1.f;
                        }
                    tk.x /= cnt; tk.y /= cnt; tk.z /= cnt; tk.w /= cnt;
                }
            } else if (!p.mean_mode) {
                tk = rv.load(p.nv, c);
            } else {
                tk = make_float4(0.f, 0.f, 0.f, 0.f);
                float cnt = 0.f;
                const uint8_t* tm = p.text_mask + (size_t)b * p.T;
                for (int t = 0; t < p.T; ++t)
                    if (tm[t]) {
                        const float4 a = rv.load(p.nv + t, c);
                        tk.x += a.x; tk.y += a.y; tk.z += a.z; tk.w += a.w;
                        cnt += 1.f;
                    }
                tk.x /= cnt; tk.y /= cnt; tk.z /= cnt; tk.w /= cnt;
            }
            *reinterpret_cast<float4*>(txt + c) = tk;
        }
    }
    __syncthreads();
    const int s = blockIdx.x * 4 + wave;
    if (s >= p.nx) return;
    float xx = 0.f, xv = 0.f, vv = 0.f, xt = 0.f, tt = 0.f;
    for (int c = lane * 4; c < D; c += 256) {
        const float4 a = rv.load(1 + p.nz + s, c);
        const float4 v = *reinterpret_cast<const float4*>(vis + c);
        xx += a.x * a.x + a.y * a.y + a.z * a.z + a.w * a.w;
        xv += a.x * v.x + a.y * v.y + a.z * v.z + a.w * v.w;
        vv += v.x * v.x + v.y * v.y + v.z * v.z + v.w * v.w;
        if (!p.skip_text) {
            const float4 q = *reinterpret_cast<const float4*>(txt + c);
            xt += a.x * q.x + a.y * q.y + a.z * q.z + a.w * q.w;
            tt += q.x * q.x + q.y * q.y + q.z * q.z + q.w * q.w;
        }
    }
    const float tau = __expf(p.logit_scale[0]);
    xx = fmaxf(sqrtf(wave_sum(xx)), 1e-12f);
    vv = fmaxf(sqrtf(wave_sum(vv)), 1e-12f);
    const float lv = tau * wave_sum(xv) / (xx * vv);
    float lt = 0.f;
    if (!p.skip_text) {
        tt = fmaxf(sqrtf(wave_sum(tt)), 1e-12f);
        lt = tau * wave_sum(xt) / (xx * tt);
    }
    const float out = fl == 0 ? lv : (fl == 1 ? lt : 0.5f * (lv + lt));
    if (lane == 0) p.logits[((size_t)b * p.n_cont + p.slot) * p.nx + s] = out;
}

hipError_t launch_contrast(const ContrastParams& p, hipStream_t s) {
    hipLaunchKernelGGL(contrast_kernel, dim3((p.nx + 3) / 4, p.B), dim3(256), 2 * p.D * sizeof(float), s, p);
    return hipGetLastError();
}

// The logits job of a LayerNorm-free frame (fold.h::ct_job_block, normally the last workgroups of a QKV launch) as a launch of its own: the parity tests' handle on it
__global__ __launch_bounds__(256) void ct_job_kernel(const CtJob j) { ct_job_block(j, blockIdx.x, g_zero_row); }
hipError_t launch_ct_job(const CtJob& j, hipStream_t s) {
    if (j.D % 4 != 0 || j.D > 1024 || j.B <= 0 || j.nx <= 0) return hipErrorInvalidValue;
    hipLaunchKernelGGL(ct_job_kernel, dim3((j.B * j.nx + 3) / 4), dim3(256), 0, s, j);
    return hipGetLastError();
}

// ------------------------------------------------------------------------------------------------
// Head prologue.  grid (ceil(nj/4), B): one wave per residual row.
//   * copies the row into the output dict entry it belongs to (extractor.py:66-76)
//   * search rows: bf16 NHWC conv input (the transpose/reshape of head:73 is free: tokens ARE NHWC),
//     optionally a second copy scaled by the flag-selected token (CLS_TOKENIZE, head:64-69,74)
//   * search rows: cont_score = tau * normalize(x) . normalize(prompt_k) with the SOFTMAX_ONE fold (head:140-148)
// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ void write_cont(const HeadPrepParams& p, int b, int s, float c0, float c1, float c2) {
    if (p.train_cont) {                          // no-prompt branch (head:134-138): two channels
        float* o = p.o_cont + ((size_t)b * p.nx + s) * 2;
        o[0] = c0;
        o[1] = p.softmax_one ? fmaxf(fmaxf(c1, c2), 0.f) : fmaxf(c1, c2);
    } else if (p.softmax_one) {                  // test branch (head:140-148)
        float* o = p.o_cont + ((size_t)b * p.nx + s) * 3;
        o[0] = c0;
        o[1] = fmaxf(fmaxf(c1, c2), 0.f);
        o[2] = 0.f;
    } else {
        float* o = p.o_cont + ((size_t)b * p.nx + s) * 2;
        o[0] = c0;
        o[1] = fmaxf(c1, c2);
    }
}

// All global loads of a wave's row and of the operands it is compared against are issued before the first use (one memory
// round trip, see ln_body); FULL: D == NV * 256.
template <int NV, bool FULL>
__global__ __launch_bounds__(256) void head_prep_kernel(const HeadPrepParams p) {
    kernarg_warm<sizeof(HeadPrepParams)>();
    extern __shared__ float sh[];              // [D] txt token, [D] cls-tokenize token
    const int b = blockIdx.y;
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int D = p.D;
    bool ok[NV];
    int cc[NV];
#pragma unroll
    for (int i = 0; i < NV; ++i) {
        const int c0 = (lane + 64 * i) * 4;
        ok[i] = FULL || c0 < D;
        cc[i] = ok[i] ? c0 : 0;
    }
    if (p.cont_only) {                         // cont_score of the search rows against a prompt that was computed after the first pass
        const int s = blockIdx.x * 4 + wave;
        if (s >= p.nx) return;
        const float* xr = p.o_search + ((size_t)b * p.nx + s) * D;
        const float* pr = p.prompt + (size_t)b * 3 * D;
        float4 a[NV], p0[NV], p1[NV], p2[NV];
#pragma unroll
        for (int i = 0; i < NV; ++i) {
            a[i] = *reinterpret_cast<const float4*>(xr + cc[i]);
            p0[i] = *reinterpret_cast<const float4*>(pr + cc[i]);
            p1[i] = *reinterpret_cast<const float4*>(pr + D + cc[i]);
            p2[i] = *reinterpret_cast<const float4*>(pr + 2 * D + cc[i]);
        }
        const float ls = sload_f32(p.logit_scale);
        __builtin_amdgcn_sched_barrier(0);
        float xx = 0.f, d0 = 0.f, d1 = 0.f, d2 = 0.f, n0 = 0.f, n1 = 0.f, n2 = 0.f;
#pragma unroll
        for (int i = 0; i < NV; ++i) {
            if (!ok[i]) continue;
            xx += a[i].x * a[i].x + a[i].y * a[i].y + a[i].z * a[i].z + a[i].w * a[i].w;
            d0 += a[i].x * p0[i].x + a[i].y * p0[i].y + a[i].z * p0[i].z + a[i].w * p0[i].w;
            d1 += a[i].x * p1[i].x + a[i].y * p1[i].y + a[i].z * p1[i].z + a[i].w * p1[i].w;
            d2 += a[i].x * p2[i].x + a[i].y * p2[i].y + a[i].z * p2[i].z + a[i].w * p2[i].w;
            n0 += p0[i].x * p0[i].x + p0[i].y * p0[i].y + p0[i].z * p0[i].z + p0[i].w * p0[i].w;
            n1 += p1[i].x * p1[i].x + p1[i].y * p1[i].y + p1[i].z * p1[i].z + p1[i].w * p1[i].w;
            n2 += p2[i].x * p2[i].x + p2[i].y * p2[i].y + p2[i].z * p2[i].z + p2[i].w * p2[i].w;
        }
        xx = fmaxf(sqrtf(wave_sum(xx)), 1e-12f);
        n0 = fmaxf(sqrtf(wave_sum(n0)), 1e-12f);
        n1 = fmaxf(sqrtf(wave_sum(n1)), 1e-12f);
        n2 = fmaxf(sqrtf(wave_sum(n2)), 1e-12f);
        const float tau = __expf(ls);
        const float c0 = tau * wave_sum(d0) / (xx * n0), c1 = tau * wave_sum(d1) / (xx * n1), c2 = tau * wave_sum(d2) / (xx * n2);
        if (lane == 0) write_cont(p, b, s, c0, c1, c2);
        return;
    }
    const float* xb = p.x + (size_t)b * p.nj * D;
    const int fl = (int)sload_u32(p.flag + b);       // low word of the int64 flag (0 / 1 / 2)
    const bool with_cont = p.prompt != nullptr && p.o_cont != nullptr;
    const float ls = with_cont ? sload_f32(p.logit_scale) : 0.f;
    const float cls = p.ct_logits ? sload_f32(p.ct_logit_scale) : 0.f;
    float* txt_tok = sh;
    float* tok = sh + D;
    const bool have_text = !p.skip_text;
    const bool use_lds = have_text && (p.mean_mode || p.cls_tokenize);   // 'cls' text token without CLS_TOKENIZE: straight from its row
    const int rows = have_text ? p.nj : p.nv;
    const int r = blockIdx.x * 4 + wave;
    const bool active = r < rows;
    const bool is_search = active && r >= 1 + p.nz && r < p.nv;
    const int s = r - 1 - p.nz;
    const bool txt_out = r == 0 && have_text && p.o_txt != nullptr;     // the wave of the cls row also writes txt_token
    const float* xr = xb + (size_t)(active ? r : 0) * D;
    // ---- loads: the row; for search rows the vis token, the text token and the three prompts (valid addresses even when unused) ----
    float4 a[NV], v[NV], q[NV], p0[NV], p1[NV], p2[NV];
#pragma unroll
    for (int i = 0; i < NV; ++i) a[i] = *reinterpret_cast<const float4*>(xr + cc[i]);
    if (is_search || txt_out) {
        const float* tq = have_text ? xb + (size_t)p.nv * D : xb;
        const float* pq = with_cont ? p.prompt + (size_t)b * 3 * D : xb;
#pragma unroll
        for (int i = 0; i < NV; ++i) {
            v[i] = *reinterpret_cast<const float4*>(xb + cc[i]);
            q[i] = *reinterpret_cast<const float4*>(tq + cc[i]);
            p0[i] = *reinterpret_cast<const float4*>(pq + cc[i]);
            p1[i] = *reinterpret_cast<const float4*>(pq + D + cc[i]);
            p2[i] = *reinterpret_cast<const float4*>(pq + 2 * D + cc[i]);
        }
    }
    __builtin_amdgcn_sched_barrier(0);
    if (use_lds) {                             // masked-mean text token / CLS_TOKENIZE: the token goes through LDS
        block_txt_token(txt_tok, xb + (size_t)p.nv * D, p.text_mask + (size_t)b * p.T, p.T, D, p.mean_mode);
        if (p.cls_tokenize) {
            for (int c = threadIdx.x; c < D; c += 256) {
                const float vv = xb[c], t = txt_tok[c];
                tok[c] = fl == 0 ? vv : (fl == 1 ? t : 0.5f * (vv + t));
            }
            __syncthreads();
        }
        if (is_search || txt_out) {
#pragma unroll
            for (int i = 0; i < NV; ++i) q[i] = *reinterpret_cast<const float4*>(txt_tok + cc[i]);
        }
    } else if (p.cls_tokenize) {               // no text at all: the token is the vis token
        for (int c = threadIdx.x; c < D; c += 256) tok[c] = fl == 1 ? 0.f : (fl == 0 ? xb[c] : 0.5f * xb[c]);
        __syncthreads();
    }
    if (txt_out) {
#pragma unroll
        for (int i = 0; i < NV; ++i)
            if (ok[i]) *reinterpret_cast<float4*>(p.o_txt + (size_t)b * D + cc[i]) = q[i];
    }
    if (!active) return;
    float* dst = nullptr;
    if (r == 0) dst = p.o_vis ? p.o_vis + (size_t)b * D : nullptr;
    else if (r < 1 + p.nz) dst = p.o_template ? p.o_template + ((size_t)b * p.nz + (r - 1)) * D : nullptr;
    else if (r < p.nv) dst = p.o_search ? p.o_search + ((size_t)b * p.nx + (r - 1 - p.nz)) * D : nullptr;
    else dst = p.o_text ? p.o_text + ((size_t)b * p.T + (r - p.nv)) * D : nullptr;
    float xx = 0.f, d0 = 0.f, d1 = 0.f, d2 = 0.f, n0 = 0.f, n1 = 0.f, n2 = 0.f;
#pragma unroll
    for (int i = 0; i < NV; ++i) {
        if (!ok[i]) continue;
        const int c = cc[i];
        if (dst) *reinterpret_cast<float4*>(dst + c) = a[i];
        if (is_search) {
            bf16_t* g = p.g0 + ((size_t)b * p.nx + s) * p.g0_ld;
            uint2 w;
            w.x = pack_bf16x2(a[i].x, a[i].y);
            w.y = pack_bf16x2(a[i].z, a[i].w);
            *reinterpret_cast<uint2*>(g + c) = w;
            if (p.cls_tokenize) {
                const float4 t = *reinterpret_cast<const float4*>(tok + c);
                w.x = pack_bf16x2(a[i].x * t.x, a[i].y * t.y);
                w.y = pack_bf16x2(a[i].z * t.z, a[i].w * t.w);
                *reinterpret_cast<uint2*>(g + D + c) = w;
            }
            if (with_cont) {
                xx += a[i].x * a[i].x + a[i].y * a[i].y + a[i].z * a[i].z + a[i].w * a[i].w;
                d0 += a[i].x * p0[i].x + a[i].y * p0[i].y + a[i].z * p0[i].z + a[i].w * p0[i].w;
                d1 += a[i].x * p1[i].x + a[i].y * p1[i].y + a[i].z * p1[i].z + a[i].w * p1[i].w;
                d2 += a[i].x * p2[i].x + a[i].y * p2[i].y + a[i].z * p2[i].z + a[i].w * p2[i].w;
                n0 += p0[i].x * p0[i].x + p0[i].y * p0[i].y + p0[i].z * p0[i].z + p0[i].w * p0[i].w;
                n1 += p1[i].x * p1[i].x + p1[i].y * p1[i].y + p1[i].z * p1[i].z + p1[i].w * p1[i].w;
                n2 += p2[i].x * p2[i].x + p2[i].y * p2[i].y + p2[i].z * p2[i].z + p2[i].w * p2[i].w;
            }
        }
    }
    if (is_search && with_cont) {
        xx = fmaxf(sqrtf(wave_sum(xx)), 1e-12f);
        n0 = fmaxf(sqrtf(wave_sum(n0)), 1e-12f);
        n1 = fmaxf(sqrtf(wave_sum(n1)), 1e-12f);
        n2 = fmaxf(sqrtf(wave_sum(n2)), 1e-12f);
        const float tau = __expf(ls);
        const float c0 = tau * wave_sum(d0) / (xx * n0);
        const float c1 = tau * wave_sum(d1) / (xx * n1);
        const float c2 = tau * wave_sum(d2) / (xx * n2);
        if (lane == 0) write_cont(p, b, s, c0, c1, c2);
    }
    if (is_search && p.ct_logits) {              // same arithmetic and order as contrast_kernel
        float sx = 0.f, xv = 0.f, vv = 0.f, xt = 0.f, tt = 0.f;
#pragma unroll
        for (int i = 0; i < NV; ++i) {
            if (!ok[i]) continue;
            sx += a[i].x * a[i].x + a[i].y * a[i].y + a[i].z * a[i].z + a[i].w * a[i].w;
            xv += a[i].x * v[i].x + a[i].y * v[i].y + a[i].z * v[i].z + a[i].w * v[i].w;
            vv += v[i].x * v[i].x + v[i].y * v[i].y + v[i].z * v[i].z + v[i].w * v[i].w;
            if (have_text) {
                xt += a[i].x * q[i].x + a[i].y * q[i].y + a[i].z * q[i].z + a[i].w * q[i].w;
                tt += q[i].x * q[i].x + q[i].y * q[i].y + q[i].z * q[i].z + q[i].w * q[i].w;
            }
        }
        const float tau = __expf(cls);
        sx = fmaxf(sqrtf(wave_sum(sx)), 1e-12f);
        vv = fmaxf(sqrtf(wave_sum(vv)), 1e-12f);
        const float lv = tau * wave_sum(xv) / (sx * vv);
        float lt = 0.f;
        if (have_text) {
            tt = fmaxf(sqrtf(wave_sum(tt)), 1e-12f);
            lt = tau * wave_sum(xt) / (sx * tt);
        }
        const float o = fl == 0 ? lv : (fl == 1 ? lt : 0.5f * (lv + lt));
        if (lane == 0) p.ct_logits[((size_t)b * p.ct_ncont + p.ct_slot) * p.nx + s] = o;
    }
}

hipError_t launch_head_prep(const HeadPrepParams& p, hipStream_t s) {
    const int rows = p.cont_only ? p.nx : (p.skip_text ? p.nv : p.nj);
    if (p.cont_only && (!p.o_search || !p.prompt || !p.o_cont)) return hipErrorInvalidValue;
    if (p.D % 4 != 0 || p.D > 1024) return hipErrorInvalidValue;
    const dim3 grid((rows + 3) / 4, p.B), block(256);
    const size_t lds = 2 * p.D * sizeof(float);
    if (p.D == 768) hipLaunchKernelGGL((head_prep_kernel<3, true>), grid, block, lds, s, p);
    else if (p.D == 1024) hipLaunchKernelGGL((head_prep_kernel<4, true>), grid, block, lds, s, p);
    else if (p.D <= 256) hipLaunchKernelGGL((head_prep_kernel<1, false>), grid, block, lds, s, p);
    else if (p.D <= 768) hipLaunchKernelGGL((head_prep_kernel<3, false>), grid, block, lds, s, p);
    else hipLaunchKernelGGL((head_prep_kernel<4, false>), grid, block, lds, s, p);
    return hipGetLastError();
}

// ------------------------------------------------------------------------------------------------
// Head tail.  One workgroup per sample: the four 1x1 convs on the 32-channel tower outputs, sigmoids,
// size-map select by flag (head:80-82), convert2bbox (head:108-119) and the argmax over S.
// ------------------------------------------------------------------------------------------------
// C8 == 32 (HEAD_DIM 256): a position's 4 x 32 tower outputs, its cont_score row and grid coordinates are fetched with 16-byte
// loads before anything else and the 1x1 weights are staged meanwhile -- one memory round trip for the sample's single
// workgroup instead of one per channel group; C8 == 0: any width, plain loops.
template <int C8>
__global__ __launch_bounds__(256) void head_tail_kernel(const HeadTailParams p) {
    kernarg_warm<sizeof(HeadTailParams)>();
    extern __shared__ float shw[];              // [7*c8] weights, [8] bias
    __shared__ float red_v[256];
    __shared__ int red_i[256];
    const int b = blockIdx.x, c8 = C8 ? C8 : p.c8;
    constexpr int NG = C8 ? C8 / 2 : 1;         // 16-byte groups of a position's 4 * C8 bf16 values
    uint4 gv[NG];
    float cs[3], gx = 0.f, gy = 0.f;
    auto fetch = [&](int s) __attribute__((always_inline)) {
        const float* cr = p.cont + ((size_t)b * p.S + s) * p.cont_ch;
#pragma unroll
        for (int k = 0; k < 3; ++k) cs[k] = cr[k < p.cont_ch ? k : 0];
        gx = p.coord[s];
        gy = p.coord[p.S + s];
        if constexpr (C8 != 0) {
            const uint4* g = reinterpret_cast<const uint4*>(p.g4 + ((size_t)b * p.S + s) * p.ld);
#pragma unroll
            for (int q = 0; q < NG; ++q) gv[q] = g[q];
        }
    };
    if ((int)threadIdx.x < p.S) fetch(threadIdx.x);
    const int fl = (int)p.flag[b];
    for (int i = threadIdx.x; i < 7 * c8; i += 256) shw[i] = p.w1[i];
    if (threadIdx.x < 7) shw[7 * c8 + threadIdx.x] = p.b1[threadIdx.x];
    __syncthreads();
    const float* bias = shw + 7 * c8;
    float best = -INFINITY;
    int best_i = 0x7fffffff;
    float4 best_bb = make_float4(0.f, 0.f, 0.f, 0.f);
    for (int s = threadIdx.x; s < p.S; s += 256) {
        if (s != (int)threadIdx.x) fetch(s);
        float o[7];
        // tower t reads channels [t*c8, (t+1)*c8); outputs: 0 cls | 1,2 offset | 3,4 bbox | 5,6 bbox_grounding
        constexpr int tower_of[7] = {0, 1, 1, 2, 2, 3, 3};
#pragma unroll
        for (int k = 0; k < 7; ++k) {
            float acc = bias[k];
            if constexpr (C8 != 0) {
#pragma unroll
                for (int c = 0; c < C8; ++c) {
                    const int e = tower_of[k] * C8 + c;
                    const uint4 u = gv[e / 8];
                    const uint32_t w = (e % 8) / 2 == 0 ? u.x : ((e % 8) / 2 == 1 ? u.y : ((e % 8) / 2 == 2 ? u.z : u.w));
                    const float gval = __uint_as_float((e & 1) ? (w & 0xffff0000u) : (w << 16));
                    acc += gval * shw[k * C8 + c];
                }
            } else {
                const bf16_t* gt = p.g4 + ((size_t)b * p.S + s) * p.ld + tower_of[k] * c8;
                for (int c = 0; c < c8; ++c) acc += bf2f(gt[c]) * shw[k * c8 + c];
            }
            o[k] = acc;
        }
        const float cls = sigmoidf_(o[0]);
        const float ox = p.offset_sigmoid ? sigmoidf_(o[1]) : o[1];
        const float oy = p.offset_sigmoid ? sigmoidf_(o[2]) : o[2];
        const float w = fl == 1 ? sigmoidf_(o[5]) : sigmoidf_(o[3]);
        const float h = fl == 1 ? sigmoidf_(o[6]) : sigmoidf_(o[4]);
        float mx = cs[0];
        for (int k = 1; k < p.cont_ch; ++k) mx = fmaxf(mx, cs[k < 3 ? k : 0]);
        float den = 0.f;
        for (int k = 0; k < p.cont_ch; ++k) den += __expf(cs[k < 3 ? k : 0] - mx);
        const float p0 = __expf(cs[0] - mx) / den;
        const float score = cls * p0;
        const size_t bs = (size_t)b * p.S + s;
        if (p.o_cls_test) p.o_cls_test[bs] = cls;
        if (p.o_cls) p.o_cls[bs] = p.joint_cls ? score : cls;
        float4 bb;
        bb.x = (gx + ox) / (float)p.F;
        bb.y = (gy + oy) / (float)p.F;
        bb.z = w;
        bb.w = h;
        if (p.o_bbox_map) *reinterpret_cast<float4*>(p.o_bbox_map + bs * 4) = bb;
        if (score > best) { best = score; best_i = s; best_bb = bb; }
    }
    red_v[threadIdx.x] = best;
    red_i[threadIdx.x] = best_i;
    __syncthreads();
    for (int st = 128; st > 0; st >>= 1) {
        if (threadIdx.x < st) {
            const float ov = red_v[threadIdx.x + st];
            const int oi = red_i[threadIdx.x + st];
            if (ov > red_v[threadIdx.x] || (ov == red_v[threadIdx.x] && oi < red_i[threadIdx.x])) {
                red_v[threadIdx.x] = ov;
                red_i[threadIdx.x] = oi;
            }
        }
        __syncthreads();
    }
    const bool found = red_i[0] < p.S;           // false only when no score compared greater than -inf (NaN everywhere): position 0
    const int win = found ? red_i[0] : 0;
    if (p.o_pred && p.o_bbox_map) {
        // the winner is some thread's own best (strict >, lowest index first): that thread still holds its box
        if (found && best_i == win) *reinterpret_cast<float4*>(p.o_pred + (size_t)b * 4) = best_bb;
        if (!found && threadIdx.x == 0) {
            // thread 0 computed position 0 itself in its first iteration; its store is visible to itself
            *reinterpret_cast<float4*>(p.o_pred + (size_t)b * 4) = *reinterpret_cast<const float4*>(p.o_bbox_map + (size_t)b * p.S * 4);
        }
    }
    if (threadIdx.x == 0 && p.o_argmax) p.o_argmax[b] = win;
}

hipError_t launch_head_tail(const HeadTailParams& p, hipStream_t s) {
    const size_t lds = (7 * p.c8 + 8) * sizeof(float);
    if (p.cont_ch < 1 || p.cont_ch > 3) return hipErrorInvalidValue;
    if (p.c8 == 32 && p.ld % 8 == 0) hipLaunchKernelGGL(head_tail_kernel<32>, dim3(p.B), dim3(256), lds, s, p);
    else hipLaunchKernelGGL(head_tail_kernel<0>, dim3(p.B), dim3(256), lds, s, p);
    return hipGetLastError();
}

// ------------------------------------------------------------------------------------------------
// Fold the split-K f32 slabs of a conv layer: out = relu(sum_s slab[s]) as bf16 (bias was added into slab 0).
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void slab_relu_kernel(const float* __restrict__ slabs, int nsplit, size_t stride,
                                                        bf16_t* __restrict__ out, size_t n) {
    const size_t i = ((size_t)blockIdx.x * 256 + threadIdx.x) * 4;
    if (i >= n) return;
    float4 sl[8];
#pragma unroll
    for (int sp = 0; sp < 8; ++sp)
        sl[sp] = (sp < nsplit) ? *reinterpret_cast<const float4*>(slabs + (size_t)sp * stride + i) : make_float4(0.f, 0.f, 0.f, 0.f);
    float4 a = sl[0];
#pragma unroll
    for (int sp = 1; sp < 8; ++sp) { a.x += sl[sp].x; a.y += sl[sp].y; a.z += sl[sp].z; a.w += sl[sp].w; }
    uint2 w;
    w.x = pack_bf16x2(fmaxf(a.x, 0.f), fmaxf(a.y, 0.f));
    w.y = pack_bf16x2(fmaxf(a.z, 0.f), fmaxf(a.w, 0.f));
    *reinterpret_cast<uint2*>(out + i) = w;
}
hipError_t launch_slab_relu(const float* slabs, int nsplit, size_t stride, bf16_t* out, size_t n, hipStream_t s) {
    if (nsplit < 1 || nsplit > 8 || n % 4) return hipErrorInvalidValue;
    hipLaunchKernelGGL(slab_relu_kernel, dim3((unsigned)((n / 4 + 255) / 256)), dim3(256), 0, s, slabs, nsplit, stride, out, n);
    return hipGetLastError();
}

// ------------------------------------------------------------------------------------------------
// Tracker decode on the device (lib/test/tracker/uvltrack.py:116-125,167-173 + box_ops.clip_box :117-126): replaces the
// three .cpu() round trips per frame with one tiny kernel.  One workgroup per sample.
// ------------------------------------------------------------------------------------------------
// Every operand (the position's scores and box, the sample's state / scale / frame size) is requested before the first use; the
// thread that owns the winning position finishes the box from its own registers.
__global__ __launch_bounds__(256) void decode_kernel(const DecodeParams p) {
    kernarg_warm<sizeof(DecodeParams)>();
    __shared__ float red_v[256];
    __shared__ int red_i[256];
    const int b = blockIdx.x;
    const bool has_cont = p.cont != nullptr;
    const float* st = p.state + (size_t)b * 4;
    const float s0 = st[0], s1 = st[1], s2 = st[2], s3 = st[3];
    const float rf = p.resize_factor[b];
    const float H = p.image_hw[2 * b], W = p.image_hw[2 * b + 1];
    float best = -INFINITY, best_pc = 1.f, best_cls = 0.f;
    int best_i = 0x7fffffff;
    float4 best_net = make_float4(0.f, 0.f, 0.f, 0.f);
    for (int s = threadIdx.x; s < p.S; s += 256) {
        const size_t bs = (size_t)b * p.S + s;
        const float* cr = has_cont ? p.cont + bs * p.cont_ch : p.cls + bs;      // valid address either way
        float cs[3];
#pragma unroll
        for (int k = 0; k < 3; ++k) cs[k] = cr[(has_cont && k < p.cont_ch) ? k : 0];
        const float cl = p.cls[bs], wn = p.window[s];
        const float4 net = *reinterpret_cast<const float4*>(p.bbox_map + bs * 4);
        float pc = 1.0f;
        if (has_cont) {
            float mx = cs[0];
            for (int k = 1; k < p.cont_ch; ++k) mx = fmaxf(mx, cs[k < 3 ? k : 0]);
            float den = 0.f;
            for (int k = 0; k < p.cont_ch; ++k) den += __expf(cs[k < 3 ? k : 0] - mx);
            pc = __expf(cs[0] - mx) / den;
        }
        const float m = cl * wn * pc;
        if (m > best) { best = m; best_i = s; best_pc = pc; best_cls = cl; best_net = net; }
    }
    red_v[threadIdx.x] = best;
    red_i[threadIdx.x] = best_i;
    __syncthreads();
    for (int stp = 128; stp > 0; stp >>= 1) {
        if (threadIdx.x < stp) {
            const float ov = red_v[threadIdx.x + stp];
            const int oi = red_i[threadIdx.x + stp];
            if (ov > red_v[threadIdx.x] || (ov == red_v[threadIdx.x] && oi < red_i[threadIdx.x])) { red_v[threadIdx.x] = ov; red_i[threadIdx.x] = oi; }
        }
        __syncthreads();
    }
    const bool found = red_i[0] < p.S;           // false only when nothing compared greater than -inf (NaN everywhere): position 0
    const int i = found ? red_i[0] : 0;
    if (found ? (best_i != i) : (threadIdx.x != 0)) return;
    if (!found) {                                // thread 0 owns position 0: recompute its operands
        const size_t bs = (size_t)b * p.S;
        best_net = *reinterpret_cast<const float4*>(p.bbox_map + bs * 4);
        best_cls = p.cls[bs];
        best_pc = 1.0f;
        if (has_cont) {
            const float* cs = p.cont + bs * p.cont_ch;
            float mx = cs[0];
            for (int k = 1; k < p.cont_ch; ++k) mx = fmaxf(mx, cs[k]);
            float den = 0.f;
            for (int k = 0; k < p.cont_ch; ++k) den += __expf(cs[k] - mx);
            best_pc = __expf(cs[0] - mx) / den;
        }
    }
    const float net[4] = {best_net.x, best_net.y, best_net.z, best_net.w};
    const float pc = best_pc;
    const float sc = p.search_size / rf;
    const float cx = net[0] * sc, cy = net[1] * sc, w = net[2] * sc, h = net[3] * sc;
    const float half_side = 0.5f * p.search_size / rf;
    float x1 = cx + (s0 + 0.5f * s2 - half_side) - 0.5f * w;       // map_box_back
    float y1 = cy + (s1 + 0.5f * s3 - half_side) - 0.5f * h;
    const float mg = p.margin;
    float x2 = x1 + w, y2 = y1 + h;                                       // clip_box
    x1 = fminf(fmaxf(0.f, x1), W - mg);
    x2 = fminf(fmaxf(mg, x2), W);
    y1 = fminf(fmaxf(0.f, y1), H - mg);
    y2 = fminf(fmaxf(mg, y2), H);
    float* o = p.new_state + (size_t)b * 4;
    o[0] = x1; o[1] = y1; o[2] = fmaxf(mg, x2 - x1); o[3] = fmaxf(mg, y2 - y1);
    if (p.score) p.score[b] = best_cls * pc;
    if (p.box_net) for (int k = 0; k < 4; ++k) p.box_net[(size_t)b * 4 + k] = net[k];
    if (p.index) p.index[b] = i;
}
hipError_t launch_decode(const DecodeParams& p, hipStream_t s) {
    hipLaunchKernelGGL(decode_kernel, dim3(p.B), dim3(256), 0, s, p);
    return hipGetLastError();
}

// ------------------------------------------------------------------------------------------------

// ------------------------------------------------------------------------------------------------
// weight packers
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void f32_to_bf16_kernel(const float* __restrict__ in, bf16_t* __restrict__ out, size_t n) {
    size_t i = ((size_t)blockIdx.x * 256 + threadIdx.x) * 4;
    const size_t stride = (size_t)gridDim.x * 256 * 4;
    for (; i + 3 < n; i += stride) {
        const float4 f = *reinterpret_cast<const float4*>(in + i);
        uint2 w;
        w.x = pack_bf16x2(f.x, f.y);
        w.y = pack_bf16x2(f.z, f.w);
        *reinterpret_cast<uint2*>(out + i) = w;
    }
    // tail (n % 4) handled by the first threads of block 0
    if (blockIdx.x == 0 && threadIdx.x < (n & 3)) out[(n & ~(size_t)3) + threadIdx.x] = f2bf(in[(n & ~(size_t)3) + threadIdx.x]);
}
hipError_t launch_f32_to_bf16(const float* in, bf16_t* out, size_t n, hipStream_t s) {
    size_t blocks = (n / 4 + 255) / 256;
    if (blocks < 1) blocks = 1;
    if (blocks > 4096) blocks = 4096;
    hipLaunchKernelGGL(f32_to_bf16_kernel, dim3((unsigned)blocks), dim3(256), 0, s, in, out, n);
    return hipGetLastError();
}

__global__ __launch_bounds__(256) void copy_f32_kernel(const float* __restrict__ in, float* __restrict__ out, size_t n) {
    size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    const size_t stride = (size_t)gridDim.x * 256;
    for (; i < n; i += stride) out[i] = in[i];
}
hipError_t launch_copy_f32(const float* in, float* out, size_t n, hipStream_t s) {
    size_t blocks = (n + 255) / 256;
    if (blocks < 1) blocks = 1;
    if (blocks > 4096) blocks = 4096;
    hipLaunchKernelGGL(copy_f32_kernel, dim3((unsigned)blocks), dim3(256), 0, s, in, out, n);
    return hipGetLastError();
}

// conv3x3 [Co,Ci,3,3] + BatchNorm2d(eval, eps 1e-5) -> bf16 [Co][tap][Ci], bias' = (b - mean) * scale + beta
// (heads/utils.py:126-131; eval semantics = running statistics).
__global__ __launch_bounds__(256) void fold_conv_bn_kernel(const float* __restrict__ w, const float* __restrict__ b,
                                                           const float* __restrict__ bn_w, const float* __restrict__ bn_b,
                                                           const float* __restrict__ bn_mean, const float* __restrict__ bn_var,
                                                           bf16_t* __restrict__ w_out, float* __restrict__ b_out, int Co, int Ci) {
    const size_t total = (size_t)Co * 9 * Ci;
    size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    const size_t stride = (size_t)gridDim.x * 256;
    for (; i < total; i += stride) {
        const int ci = i % Ci;
        const int tap = (i / Ci) % 9;
        const int co = i / ((size_t)9 * Ci);
        const float scale = bn_w[co] / sqrtf(bn_var[co] + 1e-5f);
        w_out[i] = f2bf(w[((size_t)co * Ci + ci) * 9 + tap] * scale);
        if (ci == 0 && tap == 0) b_out[co] = (b[co] - bn_mean[co]) * scale + bn_b[co];
    }
}
hipError_t launch_fold_conv_bn(const float* w, const float* b, const float* bn_w, const float* bn_b, const float* bn_mean,
                               const float* bn_var, bf16_t* w_out, float* b_out, int Co, int Ci, hipStream_t s) {
    size_t blocks = ((size_t)Co * 9 * Ci + 255) / 256;
    if (blocks > 4096) blocks = 4096;
    hipLaunchKernelGGL(fold_conv_bn_kernel, dim3((unsigned)blocks), dim3(256), 0, s, w, b, bn_w, bn_b, bn_mean, bn_var, w_out, b_out, Co, Ci);
    return hipGetLastError();
}


// ------------------------------------------------------------------------------------------------
// LayerNorm-free frames (round 6, fold.h).
// fold_ln_linear: one nn.Linear that reads a LayerNorm (block.py:30-31 -> attn.qkv / mlp.fc1; bert_backbone.py:376-380 -> the next layer's query/key/value,
// :335-339 -> intermediate.dense; the embedding LayerNorm :260-274 -> layer 0's query/key/value) with the LayerNorm's affine part folded in:
//   W'[n, k] = bf16(W[n, k] gamma[k]),  b'[n] = b[n] + sum_k W[n, k] beta[k]  (f32 weights),  colsum[n] = sum_k W'[n, k]  (of the ROUNDED values: the
//   consumer's  acc - mean colsum  then cancels the row mean exactly).  One wave per output row.
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void fold_ln_linear_kernel(const float* __restrict__ W, const float* __restrict__ bias, const float* __restrict__ gamma,
                                                             const float* __restrict__ beta, bf16_t* __restrict__ Wf, float* __restrict__ bf, float* __restrict__ colsum, int N, int K) {
    const int n = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (n >= N) return;
    float cs = 0.f, bs = 0.f;
    for (int k = lane * 4; k < K; k += 256) {
        const float4 w = *reinterpret_cast<const float4*>(W + (size_t)n * K + k);
        const float4 g = *reinterpret_cast<const float4*>(gamma + k), be = *reinterpret_cast<const float4*>(beta + k);
        const uint32_t lo = pack_bf16x2(w.x * g.x, w.y * g.y), hi = pack_bf16x2(w.z * g.z, w.w * g.w);
        *reinterpret_cast<uint2*>(Wf + (size_t)n * K + k) = uint2{lo, hi};
        cs += (__uint_as_float(lo << 16) + __uint_as_float(lo & 0xffff0000u)) + (__uint_as_float(hi << 16) + __uint_as_float(hi & 0xffff0000u));
        bs += (w.x * be.x + w.y * be.y) + (w.z * be.z + w.w * be.w);
    }
    cs = wave_sum(cs);
    bs = wave_sum(bs);
    if (lane == 0) { colsum[n] = cs; bf[n] = (bias ? bias[n] : 0.f) + bs; }
}
hipError_t launch_fold_ln_linear(const float* W, const float* bias, const float* gamma, const float* beta, bf16_t* Wf, float* bf, float* colsum, int N, int K, hipStream_t s) {
    if (!W || !gamma || !beta || !Wf || !bf || !colsum || N <= 0 || K <= 0 || K % 4 != 0) return hipErrorInvalidValue;
    hipLaunchKernelGGL(fold_ln_linear_kernel, dim3((N + 3) / 4), dim3(256), 0, s, W, bias, gamma, beta, Wf, bf, colsum, N, K);
    return hipGetLastError();
}

// Text rows entering the first fusion layer (kernels.h::TextJoinParams): one wave per row, the row in registers (D <= 1024).
__global__ __launch_bounds__(256) void text_join_kernel(const TextJoinParams p) {
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const int m = blockIdx.x * 4 + wave;
    if (m >= p.B * p.T) return;
    const int b = m / p.T, t = m - b * p.T, D = p.D;
    float* xr = p.x + ((size_t)b * p.xbs + p.xro + t) * D;
    const float* src = p.alt ? p.alt + (size_t)m * D : xr;
    const float* ad = p.add ? p.add : g_zero_row;
    float4 v[4], g[4], be[4], a4[4];
    bool ok[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int c0 = (lane + 64 * i) * 4;
        ok[i] = c0 < D;
        const int c = ok[i] ? c0 : 0;
        v[i] = *reinterpret_cast<const float4*>(src + c);
        g[i] = *reinterpret_cast<const float4*>(p.gamma + c);
        be[i] = *reinterpret_cast<const float4*>(p.beta + c);
        a4[i] = *reinterpret_cast<const float4*>(ad + c);
    }
    if (!p.alt) {
        float sum = 0.f;
#pragma unroll
        for (int i = 0; i < 4; ++i) { v[i] = sel4(ok[i], v[i]); sum += v[i].x + v[i].y + v[i].z + v[i].w; }
        const float mean = wave_sum(sum) / (float)D;
        float sq = 0.f;
#pragma unroll
        for (int i = 0; i < 4; ++i)
            if (ok[i]) {
                const float dx = v[i].x - mean, dy = v[i].y - mean, dz = v[i].z - mean, dw = v[i].w - mean;
                sq += dx * dx + dy * dy + dz * dz + dw * dw;
            }
        const float rstd = 1.0f / sqrtf(wave_sum(sq) / (float)D + p.eps);
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            v[i].x = (v[i].x - mean) * rstd * g[i].x + be[i].x; v[i].y = (v[i].y - mean) * rstd * g[i].y + be[i].y;
            v[i].z = (v[i].z - mean) * rstd * g[i].z + be[i].z; v[i].w = (v[i].w - mean) * rstd * g[i].w + be[i].w;
        }
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int c = (lane + 64 * i) * 4;
        if (ok[i] && p.snap) *reinterpret_cast<float4*>(p.snap + (size_t)m * D + c) = v[i];
        if (p.add) { v[i].x += a4[i].x; v[i].y += a4[i].y; v[i].z += a4[i].z; v[i].w += a4[i].w; }
        const uint32_t lo = pack_bf16x2(v[i].x, v[i].y), hi = pack_bf16x2(v[i].z, v[i].w);
        float s1, s2;
        st_of4(v[i].x, v[i].y, v[i].z, v[i].w, s1, s2);
        s1 = oct_sum(s1);
        s2 = oct_sum(s2);
        if (ok[i]) {
            *reinterpret_cast<float4*>(xr + c) = v[i];
            const size_t nrow = (size_t)b * p.xn_bs + p.xn_ro + t;
            if (p.xn) *reinterpret_cast<uint2*>(p.xn + nrow * D + c) = uint2{lo, hi};
            if (p.st && (lane & 7) == 0) *reinterpret_cast<float2*>(p.st + st_off(c >> 5, nrow, (size_t)p.st_rows)) = float2{s1, s2};
        }
    }
}
hipError_t launch_text_join(const TextJoinParams& p, hipStream_t s) {
    if (!p.x || !p.gamma || !p.beta || p.D % 32 != 0 || p.D > 1024 || p.B <= 0 || p.T <= 0) return hipErrorInvalidValue;
    g_last_kernel = "text_join_kernel";
    hipLaunchKernelGGL(text_join_kernel, dim3((p.B * p.T + 3) / 4), dim3(256), 0, s, p);
    return hipGetLastError();
}

}  // namespace uvl
