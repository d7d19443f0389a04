// Shared device helpers for the gfx950 kernels (wave64, MFMA 32x32x16 bf16).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

typedef __bf16 bf16_t;
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) __bf16 bf16x4;
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(4))) uint32_t u32x4;   // 16-byte staging register (native vector: stays in VGPRs)

#define WAVE 64

__device__ __forceinline__ bf16_t f2bf(float f) { return (bf16_t)f; }          // RNE (v_cvt_pk_bf16_f32 on gfx950)
__device__ __forceinline__ float bf2f(bf16_t h) { return (float)h; }

typedef __attribute__((ext_vector_type(2))) float f32x2;
typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2;
// two f32 -> packed bf16 pair (RNE), ONE v_cvt_pk_bf16_f32 (the scalar-cast form costs 2 cvt + shift + or)
__device__ __forceinline__ uint32_t pack_bf16x2(float lo, float hi) {
    const f32x2 v = {lo, hi};
    union { bf16x2 h; uint32_t u; } r;
    r.h = __builtin_convertvector(v, bf16x2);
    return r.u;
}

// A 32-bit word at a wave-uniform address through the scalar cache (read-only inputs: flags, logit scales).  hipcc sometimes
// fetches such a word with a vector load and then waits on vmcnt for it in front of the row loads; this keeps it off that counter.
__device__ __forceinline__ uint32_t sload_u32(const void* uniform_ptr) {
    uint32_t v;
    asm volatile("s_load_dword %0, %1, 0x0\n\ts_waitcnt lgkmcnt(0)" : "=s"(v) : "s"(uniform_ptr) : "memory");
    return v;
}
__device__ __forceinline__ float sload_f32(const float* uniform_ptr) { return __uint_as_float(sload_u32(uniform_ptr)); }

// Wave-wide sum / maximum, every lane gets the result.  The xor butterfly 32, 16, 8, 4, 2, 1 -- the pairs (and so the bits) of the `__shfl_xor` loop this replaces
// (round 5) -- without the LDS crossbar: v_permlane32_swap / v_permlane16_swap exchange the halves / the odd and even rows of 16, and inside a row a rotation by
// 8, 4, 2, 1 IS the xor exchange once the row has that period (v_add_f32_dpp row_ror).  Eight VALU instead of six ds_bpermute round trips (~60-100 cycles each,
// each behind its own lgkmcnt wait): a LayerNorm row does two of these, the contrastive job riding on it five more.
__device__ __forceinline__ float wave_sum(float v) {
#if __HIP_DEVICE_COMPILE__
    {
        const auto r = __builtin_amdgcn_permlane32_swap(__float_as_uint(v), __float_as_uint(v), false, false);
        v = __uint_as_float(r[0]) + __uint_as_float(r[1]);
    }
    {
        const auto r = __builtin_amdgcn_permlane16_swap(__float_as_uint(v), __float_as_uint(v), false, false);
        v = __uint_as_float(r[0]) + __uint_as_float(r[1]);
    }
    v += __uint_as_float(__builtin_amdgcn_update_dpp(0u, __float_as_uint(v), 0x128, 0xf, 0xf, false));      // row_ror:8
    v += __uint_as_float(__builtin_amdgcn_update_dpp(0u, __float_as_uint(v), 0x124, 0xf, 0xf, false));
    v += __uint_as_float(__builtin_amdgcn_update_dpp(0u, __float_as_uint(v), 0x122, 0xf, 0xf, false));
    v += __uint_as_float(__builtin_amdgcn_update_dpp(0u, __float_as_uint(v), 0x121, 0xf, 0xf, false));
#endif
    return v;
}
// v(lane) (+ | max) v(lane ^ 32): the cross-half exchange of the attention kernels (one per key tile for the running maximum), one v_permlane32_swap instead of a
// ds_bpermute round trip; the same two values are combined, so the same bits
__device__ __forceinline__ float half_sum(float v) {
#if __HIP_DEVICE_COMPILE__
    const auto r = __builtin_amdgcn_permlane32_swap(__float_as_uint(v), __float_as_uint(v), false, false);
    v = __uint_as_float(r[0]) + __uint_as_float(r[1]);
#endif
    return v;
}
__device__ __forceinline__ float half_max(float v) {
#if __HIP_DEVICE_COMPILE__
    const auto r = __builtin_amdgcn_permlane32_swap(__float_as_uint(v), __float_as_uint(v), false, false);
    v = fmaxf(__uint_as_float(r[0]), __uint_as_float(r[1]));
#endif
    return v;
}
__device__ __forceinline__ float wave_max(float v) {
#if __HIP_DEVICE_COMPILE__
    {
        const auto r = __builtin_amdgcn_permlane32_swap(__float_as_uint(v), __float_as_uint(v), false, false);
        v = fmaxf(__uint_as_float(r[0]), __uint_as_float(r[1]));
    }
    {
        const auto r = __builtin_amdgcn_permlane16_swap(__float_as_uint(v), __float_as_uint(v), false, false);
        v = fmaxf(__uint_as_float(r[0]), __uint_as_float(r[1]));
    }
    v = fmaxf(v, __uint_as_float(__builtin_amdgcn_update_dpp(0u, __float_as_uint(v), 0x128, 0xf, 0xf, false)));
    v = fmaxf(v, __uint_as_float(__builtin_amdgcn_update_dpp(0u, __float_as_uint(v), 0x124, 0xf, 0xf, false)));
    v = fmaxf(v, __uint_as_float(__builtin_amdgcn_update_dpp(0u, __float_as_uint(v), 0x122, 0xf, 0xf, false)));
    v = fmaxf(v, __uint_as_float(__builtin_amdgcn_update_dpp(0u, __float_as_uint(v), 0x121, 0xf, 0xf, false)));
#endif
    return v;
}

// exact erf GELU: 0.5 x (1 + erf(x / sqrt(2)))  (nn.GELU default; bert_backbone.py:118-124)
__device__ __forceinline__ float gelu_erf(float x) { return 0.5f * x * (1.0f + erff(x * 0.70710678118654752440f)); }
// Same function for the GEMM epilogues, where it runs once per output element: erf by Abramowitz & Stegun 7.1.26
// (|error| <= 1.5e-7, far below the bf16 rounding of the result), one v_rcp + one v_exp instead of libm's branchy erff.
__device__ __forceinline__ float gelu_erf_fast(float x) {
    const float z = fabsf(x) * 0.70710678118654752440f;
    const float t = __builtin_amdgcn_rcpf(fmaf(0.3275911f, z, 1.0f));
    float poly = fmaf(1.061405429f, t, -1.453152027f);
    poly = fmaf(poly, t, 1.421413741f);
    poly = fmaf(poly, t, -0.284496736f);
    poly = fmaf(poly, t, 0.254829592f);
    const float e = __builtin_amdgcn_exp2f(-1.44269504088896340736f * z * z);
    const float erf_abs = 1.0f - poly * t * e;                 // erf(|x| / sqrt 2)
    return 0.5f * x + 0.5f * fabsf(x) * erf_abs;               // 0.5 x (1 + sign(x) erf(|x|/sqrt 2))
}
// Two elements at a time, one transcendental per element: erf by Abramowitz & Stegun 7.1.28,
//   erf(z) = 1 - 1 / (1 + a1 z + ... + a6 z^6)^16,  |error| <= 3e-7 for z >= 0,
// i.e. six packed FMAs, four packed squarings and one v_rcp_f32 (the 7.1.26 form above needs v_rcp AND v_exp; 64 GELUs per
// lane and 128x128 tile otherwise cost as much as the tile's MFMAs).  P^16 overflows to +inf for |x| > ~25: rcp(inf) = 0, erf = 1.
__device__ __forceinline__ f32x2 gelu_erf_fast2(f32x2 x) {
    const f32x2 ax = {fabsf(x[0]), fabsf(x[1])};
    const f32x2 z = ax * 0.70710678118654752440f;
    f32x2 pz = __builtin_elementwise_fma(z, f32x2{0.0000430638f, 0.0000430638f}, f32x2{0.0002765672f, 0.0002765672f});
    pz = __builtin_elementwise_fma(pz, z, f32x2{0.0001520143f, 0.0001520143f});
    pz = __builtin_elementwise_fma(pz, z, f32x2{0.0092705272f, 0.0092705272f});
    pz = __builtin_elementwise_fma(pz, z, f32x2{0.0422820123f, 0.0422820123f});
    pz = __builtin_elementwise_fma(pz, z, f32x2{0.0705230784f, 0.0705230784f});
    pz = __builtin_elementwise_fma(pz, z, f32x2{1.0f, 1.0f});
    pz = pz * pz;
    pz = pz * pz;
    pz = pz * pz;
    pz = pz * pz;
    const f32x2 r = {__builtin_amdgcn_rcpf(pz[0]), __builtin_amdgcn_rcpf(pz[1])};
    const f32x2 hax = ax * 0.5f;
    // 0.5 x + 0.5 |x| erf(|x| / sqrt 2) = 0.5 x + 0.5 |x| - 0.5 |x| r
    return __builtin_elementwise_fma(-hax, r, __builtin_elementwise_fma(x, f32x2{0.5f, 0.5f}, hax));
}
// The GEMM epilogues' GELU since round 4: erf(x / sqrt 2) ~ xc Q(xc^2), xc = clamp(x, -c, c), Q of degree 6 (minimax fit on [0, 3.8]:
// |error| <= 1.3e-4).  The clamp point c = 3.8022366 is the float at which the f32 Horner evaluation of xc Q(xc^2) is EXACTLY 1.0f (round 5;
// round 4 clamped at 3.8, where it is 0.9999827: a residual slope of -8.6e-6 x below the clamp, -2.9e-4 at x = -30): beyond +-c the result
// is exactly x / exactly 0.  |error| of 0.5 x (1 + erf) <= 2.8e-4 everywhere (largest at x = -c, where the true value is -2.73e-4 and this
// one is 0) and <= 1.5e-4 RELATIVE for x > 0, a thirteenth of the bf16 rounding the result gets (oracle/uvl_oracle.py keeps the exact erf;
// the fixtures' gates do not move).  tests/test_gelu_poly.py restates these coefficients in numpy float32 and pins both bounds and the
// saturation; a GPU test compares uvl_linear(act = 1) with that restatement.  Ten packed operations and two clamps per PAIR, no
// transcendental: the 7.1.28 form above costs sixteen packed operations and two quarter-rate v_rcp_f32, ~10 us of VALU per fc1 launch of
// 8 UVLTrack-L sequences that no other wave's MFMAs were hiding (profiles/r04_gemm_dr.md).
#define GELU_POLY_CLAMP 3.8022366f
__device__ __forceinline__ f32x2 gelu_erf_poly2(f32x2 x) {
    const f32x2 xc = {__builtin_amdgcn_fmed3f(x[0], -GELU_POLY_CLAMP, GELU_POLY_CLAMP), __builtin_amdgcn_fmed3f(x[1], -GELU_POLY_CLAMP, GELU_POLY_CLAMP)};
    const f32x2 t = xc * xc;
    f32x2 q = __builtin_elementwise_fma(t, f32x2{7.331552609e-08f, 7.331552609e-08f}, f32x2{-4.544922376e-06f, -4.544922376e-06f});
    q = __builtin_elementwise_fma(q, t, f32x2{1.213695723e-04f, 1.213695723e-04f});
    q = __builtin_elementwise_fma(q, t, f32x2{-1.863094978e-03f, -1.863094978e-03f});
    q = __builtin_elementwise_fma(q, t, f32x2{1.863326877e-02f, 1.863326877e-02f});
    q = __builtin_elementwise_fma(q, t, f32x2{-1.314395666e-01f, -1.314395666e-01f});
    q = __builtin_elementwise_fma(q, t, f32x2{7.973535061e-01f, 7.973535061e-01f});
    const f32x2 e = xc * q;                                    // erf(x / sqrt 2)
    const f32x2 hx = x * 0.5f;
    return __builtin_elementwise_fma(hx, e, hx);
}
// Two pairs at once, the two Horner chains written step by step side by side: a chain alone is eight DEPENDENT packed operations (hipcc even
// puts an s_nop between consecutive ones), and the epilogue that runs it holds a workgroup slot (profiles/r04_gemm_dr.md, last section).  Same
// operations on every element as gelu_erf_poly2: same bits.
__device__ __forceinline__ f32x4 gelu_erf_poly4(f32x4 x) {
    const f32x2 xa = {x[0], x[1]}, xb = {x[2], x[3]};
    const f32x2 ca = {__builtin_amdgcn_fmed3f(x[0], -GELU_POLY_CLAMP, GELU_POLY_CLAMP), __builtin_amdgcn_fmed3f(x[1], -GELU_POLY_CLAMP, GELU_POLY_CLAMP)};
    const f32x2 cb = {__builtin_amdgcn_fmed3f(x[2], -GELU_POLY_CLAMP, GELU_POLY_CLAMP), __builtin_amdgcn_fmed3f(x[3], -GELU_POLY_CLAMP, GELU_POLY_CLAMP)};
    const f32x2 ta = ca * ca, tb = cb * cb;
    f32x2 qa = __builtin_elementwise_fma(ta, f32x2{7.331552609e-08f, 7.331552609e-08f}, f32x2{-4.544922376e-06f, -4.544922376e-06f});
    f32x2 qb = __builtin_elementwise_fma(tb, f32x2{7.331552609e-08f, 7.331552609e-08f}, f32x2{-4.544922376e-06f, -4.544922376e-06f});
    qa = __builtin_elementwise_fma(qa, ta, f32x2{1.213695723e-04f, 1.213695723e-04f});
    qb = __builtin_elementwise_fma(qb, tb, f32x2{1.213695723e-04f, 1.213695723e-04f});
    qa = __builtin_elementwise_fma(qa, ta, f32x2{-1.863094978e-03f, -1.863094978e-03f});
    qb = __builtin_elementwise_fma(qb, tb, f32x2{-1.863094978e-03f, -1.863094978e-03f});
    qa = __builtin_elementwise_fma(qa, ta, f32x2{1.863326877e-02f, 1.863326877e-02f});
    qb = __builtin_elementwise_fma(qb, tb, f32x2{1.863326877e-02f, 1.863326877e-02f});
    qa = __builtin_elementwise_fma(qa, ta, f32x2{-1.314395666e-01f, -1.314395666e-01f});
    qb = __builtin_elementwise_fma(qb, tb, f32x2{-1.314395666e-01f, -1.314395666e-01f});
    qa = __builtin_elementwise_fma(qa, ta, f32x2{7.973535061e-01f, 7.973535061e-01f});
    qb = __builtin_elementwise_fma(qb, tb, f32x2{7.973535061e-01f, 7.973535061e-01f});
    const f32x2 ea = ca * qa, eb = cb * qb;
    const f32x2 ha = xa * 0.5f, hb = xb * 0.5f;
    const f32x2 ra = __builtin_elementwise_fma(ha, ea, ha), rb = __builtin_elementwise_fma(hb, eb, hb);
    return f32x4{ra[0], ra[1], rb[0], rb[1]};
}
__device__ __forceinline__ float sigmoidf_(float x) { return 1.0f / (1.0f + __expf(-x)); }

// Division by a run-time value that the HOST knows at launch (tiles per row of the grid, rows per sample, heads, ...).  hipcc compiles `x / d` on wave-uniform
// operands as float reciprocal + v_readfirstlane + a chain of scalar fix-ups (~25 instructions, several VALU -> SALU hand-overs): the tile decode of the
// one-sequence GEMM kernels ran up to eight of them before its first LDS-DMA could be addressed -- a few hundred cycles at the head of an 8-us kernel, ~96 times
// per frame.  FastDiv carries m = floor(2^32 / d) from the host: q = mulhi(x, m) is floor(x / d) or one less (for every x < 2^32, d >= 1), one compare fixes it.
struct FastDiv {
    uint32_t d = 1u << 30, m = 4;           // (the default matches GemmParams::rpb's "no row map": 2^30 rows per sample)
};
static inline FastDiv fastdiv_of(uint32_t d) {
    FastDiv f;
    f.d = d ? d : 1u;
    f.m = f.d == 1u ? 0xffffffffu : (uint32_t)((1ull << 32) / f.d);
    return f;
}
__device__ __forceinline__ uint32_t fd_div(uint32_t x, const FastDiv& f) {
    const uint32_t q = __umulhi(x, f.m);
    return (x - q * f.d >= f.d) ? q + 1 : q;
}

// Output row m -> (sample b = m / rpb, row inside the sample m % rpb) for the rows of ONE block of at most 32 consecutive rows, without a division per
// row and lane: the block's first row is divided once (wave-uniform), a row of the block is that plus at most one wrap.  (An integer division by a
// run-time value is ~40 VALU; the QKV-scatter and residual epilogues did one per stored row group: 4-8 us of a 45-us GEMM, tools/probes/qkv_epi_cost.py.)
struct RowMap { int b0, rem0, rpb; };
__device__ __forceinline__ RowMap rowmap_of(int first_row, int rpb, const FastDiv& fd) {      // fd = fastdiv_of(rpb), from the launcher
    const int rf = __builtin_amdgcn_readfirstlane(first_row);
    const int b0 = (int)fd_div((uint32_t)rf, fd);
    return RowMap{b0, rf - b0 * rpb, rpb};
}
// r = row - first_row, 0 <= r <= 32; exact for rpb > 32, and for any rpb through the division
__device__ __forceinline__ void rowmap_at(const RowMap& m, int first_row, int r, int& b, int& rem) {
    if (m.rpb > 32) {
        const int rr = m.rem0 + r;
        const bool wrap = rr >= m.rpb;
        b = m.b0 + (wrap ? 1 : 0);
        rem = rr - (wrap ? m.rpb : 0);
    } else {
        const int row = first_row + r;
        b = row / m.rpb;
        rem = row - b * m.rpb;
    }
}

// MFMA 32x32x16 bf16 C/D fragment: register r of lane l holds
//   row = (r & 3) + 8 * (r >> 2) + 4 * (l >> 5),  col = l & 31
__device__ __forceinline__ int frag_row(int r, int lane) { return (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5); }

// LDS tile of [rows][64] bf16 (128-byte rows) read by ds_read_b128 with lane = row: XOR the 16-byte
// chunk index with (row>>1)&7 so the 16 lanes of one ds_read_b128 service group hit 16 distinct slots.
__device__ __forceinline__ int swz128(int row, int chunk) { return row * 128 + ((chunk ^ ((row >> 1) & 7)) << 4); }

// Same tile read by ds_read_b64 with lane = row (32-lane service groups, 64 banks): XOR the 8-byte
// chunk index (0..15) with (row>>1)&15.
__device__ __forceinline__ int swz64(int row, int chunk8) { return row * 128 + ((chunk8 ^ ((row >> 1) & 15)) << 3); }

// All cache lines of the kernel-argument segment requested at once, at kernel entry.  hipcc loads the fields of a by-value argument block where
// they are first used, in dependency order (a condition, then the pointers behind it, ...): each first touch of a 64-byte line of the freshly
// written segment is a miss, and a launch-bound kernel pays them one after the other before its first global load.  Measured on the
// one-sequence frame (rocprofv3 --stats, two runs each): GEMM launches -0.3..0.7 us, attention -0.2..0.5, LayerNorm -0.1..0.4; the frame's kernel
// time 760-775 -> 739-740 us.
// (kernels that read gridDim / blockDim add 64 bytes: the first line of the hidden arguments behind the explicit ones.  Only those: a kernel that
// uses no hidden argument has none in its segment, and the request must stay inside the segment.)
template <int BYTES>
__device__ __forceinline__ void kernarg_warm() {
#ifndef UVL_NO_KERNARG_WARM            // (development A/B: tools/probes/bench_lib.py on a variant build)
    typedef const uint32_t __attribute__((address_space(4))) cu32;
    cu32* k = (cu32*)__builtin_amdgcn_kernarg_segment_ptr();
    uint32_t a = 0;
#pragma unroll
    for (int off = 0; off < BYTES; off += 64) a |= k[off / 4];
    asm volatile("" ::"s"(a));
#endif
}

// Prefetch of the NEXT launch's weight into the memory-side cache (one-sequence frames: a launch-bound GEMM whose weight was read by the kernel
// before it is 0.7-1.2 us shorter, tools/probes/wprefetch_probe.py).  Every thread requests up to four 128-byte lines of [ptr, ptr + bytes) at
// kernel entry -- BEFORE the first LDS-DMA, so the loop's counted s_waitcnt see them as the oldest entries of the queue -- into one register
// that stays live until prefetch_retire at the end of the kernel (hipcc does not know that the asm's result arrives later: a dead register
// would be handed to another value and overwritten when the load lands).
template <int THREADS>
__device__ __forceinline__ uint32_t prefetch_issue(const void* ptr, uint32_t bytes, uint32_t wg, uint32_t nwg) {
    uint32_t sink = 0;
    if (ptr) {
        const uint32_t total = nwg * THREADS, id = wg * THREADS + threadIdx.x, lines = bytes >> 7;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const uint32_t l = id + k * total;
            if (l < lines) asm volatile("global_load_dword %0, %1, off" : "+v"(sink) : "v"(reinterpret_cast<const char*>(ptr) + (size_t)l * 128) : "memory");
        }
    }
    return sink;
}
// The same for a RIDER's next weight (the text branch's GEMM that rides in the next launch: its bytes come from HBM, nobody has read them this frame), XCD-matched:
// the rider's `tiles` one-row-of-tiles panels (`lp` 128-byte lines each, contiguous in the weight) are consumed in 8 contiguous runs, run x by the workgroups with
// blockIdx % 8 == x of THAT launch (gemm_glds_body / gemm_fin_body tile order) -- so the workgroups with blockIdx % 8 == x of THIS launch request run x: the lines
// arrive in the L2 of the XCD that will read them (every XCD has its own L2; a request from another XCD only reaches the memory-side cache).
template <int THREADS>
__device__ __forceinline__ uint32_t prefetch_issue_xcd(uint32_t sink, const void* ptr, uint32_t tiles, uint32_t lp, uint32_t wg, uint32_t nwg) {
    if (ptr) {
        const uint32_t x = wg & 7, j = wg >> 3, nx = (nwg - x + 7) >> 3;
        const uint32_t base = tiles >> 3, rem = tiles & 7;
        const uint32_t first = (x * base + (x < rem ? x : rem)) * lp, count = (base + (x < rem ? 1u : 0u)) * lp;
#pragma unroll
        for (int k = 0; k < 2; ++k) {
            const uint32_t q = (j + k * nx) * THREADS + threadIdx.x;
            if (q < count) asm volatile("global_load_dword %0, %1, off" : "+v"(sink) : "v"(reinterpret_cast<const char*>(ptr) + (size_t)(first + q) * 128) : "memory");
        }
    }
    return sink;
}
__device__ __forceinline__ void prefetch_retire(uint32_t sink) { asm volatile("s_waitcnt vmcnt(0)" ::"v"(sink) : "memory"); }
