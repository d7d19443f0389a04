// bf16 MFMA GEMM, 128 x 256 tile on four waves, TWO workgroups per CU, the weight operand loaded straight into registers (cfg 36 of
// gemm.hip's table).  Own translation unit: the other kernels are compiled with -amdgpu-mfma-vgpr-form=1 (accumulators in VGPRs); this
// one keeps the 128 accumulators of a wave in AGPRs so that the K loop can own the 128 architectural VGPRs.
#include <cstdio>
#include <type_traits>
#include "common.h"
#include "kernels.h"
#include "gemm_epi.h"

namespace uvl {

// ------------------------------------------------------------------------------------------------
// What a batched GEMM of the frame pays OUTSIDE its K loop is a third of its time at K = 1024 (profiles/r04_gemm_streamk.md: ~10 us per
// round of 256 x 256 tiles for the cold prologue, the LDS staging and the output burst; 15-30 us with the GELU / read-modify-write
// epilogues), and the one-workgroup-per-CU kernels (cfg 30 / 31) run nothing under it.  Here two workgroups share a CU -- they drift
// apart, and one's prologue / epilogue runs under the other's K loop -- without giving up the 128 x 64 wave tile:
//   * 128 x 256 tile, wave w owns all 128 rows x columns [64 w, + 64): 8 x 4 blocks of v_mfma_f32_16x16x32_bf16, 128 AGPRs;
//   * only A (shared by the four waves) goes through LDS: LDS-DMA, four stages of 16 KB; W fragments are loaded from global memory
//     straight into the lane that feeds them to the MFMA, one K tile ahead (tools/gen/gemm_dr_gen.py: schedule, LDS image, register map);
//   * 73 KB of LDS and 256 registers per wave: two workgroups per CU.
// The K loop is ONE generated inline-asm statement (gemm_dr_asm.inc).  Transposed accumulation (W fragment = MFMA row operand) and the
// epilogue arithmetic are those of gemm_pipe_tile / gemm_epilogue_lds.
// ------------------------------------------------------------------------------------------------
#define DR_R10(p, n) p #n "0", p #n "1", p #n "2", p #n "3", p #n "4", p #n "5", p #n "6", p #n "7", p #n "8", p #n "9"
#define GEMM_DR_SGPRS DR_R10("s", 4), DR_R10("s", 5)        // (m0 is written too; hipcc refuses it in a clobber list and sets it again before its own uses)
#define GEMM_DR_VGPRS "v0", "v1", "v2", "v3", "v4", "v5", "v6", "v7", "v8", "v9", DR_R10("v", 1), DR_R10("v", 2), DR_R10("v", 3), DR_R10("v", 4), \
    DR_R10("v", 5), DR_R10("v", 6), DR_R10("v", 7), DR_R10("v", 8), DR_R10("v", 9), DR_R10("v", 10), "v110", "v111", "v112", "v113"

// Epilogue for the accumulator layout of the 16x16x32 MFMA: block (ib, jb) of the wave's 128 x 64 = accv[ib], registers 4 jb .. + 3: lane l
// holds output row 16 ib + (l & 15), the four consecutive columns 16 jb + 4 (l >> 4) + r.  Same arithmetic, in the same order, as
// gemm_epilogue_lds (bias, q scale, activation, rounding) and the same row-major write-out through LDS, 32 rows at a time, so that every
// store instruction covers whole 128-byte lines; block i + 1 is converted and written to LDS before the rows of block i are read back.
template <int EPI, int ACT>
__device__ __forceinline__ void gemm_epilogue_dr(const GemmParams& p, f32x16 (&accv)[8], char* smem, const float* sbias, int m0, int n0,
                                                 int lane, int wave) {
    constexpr int WN = 64;
    constexpr int ES = (EPI == EPI_F32) ? 4 : 2;               // output element size
    constexpr int RS = WN * ES + 16;                            // padded LDS row stride
    constexpr int LPR = WN * ES / 16;                           // lanes per row on the way out
    constexpr int RPI = 64 / LPR;                               // rows per store instruction
    constexpr int NSL = (EPI == EPI_F32) ? 2 : 4;               // 32-row slices per wave (4 waves x NSL x 32 rows x RS <= 72 KB)
    const int colw = n0 + wave * WN;                            // first column of this wave's sub-tile
    const bool vpart = EPI == EPI_QKV && colw >= 2 * p.D;       // wave-uniform: D % 64 == 0
    const float qs = (EPI == EPI_QKV && colw < p.D) ? p.q_scale : 1.0f;
    const int l15 = lane & 15, g = lane >> 4;
    f32x4 bv[4];
#pragma unroll
    for (int jb = 0; jb < 4; ++jb) bv[jb] = *reinterpret_cast<const f32x4*>(sbias + wave * WN + 16 * jb + 4 * g);
    __builtin_amdgcn_s_barrier();                               // every wave has finished reading the ring
    if (vpart) {                                                // V^T is token-contiguous: stored straight from registers
#pragma unroll
        for (int ib = 0; ib < 8; ++ib) {
            __builtin_amdgcn_sched_barrier(0);
            const int rowl = m0 + ib * 16 + l15;
            if (rowl < p.M) {
                int b, rem;
                rowmap_at(rowmap_of(m0 + ib * 16, p.rpb, p.fd_rpb), m0 + ib * 16, l15, b, rem);
#pragma unroll
                for (int jb = 0; jb < 4; ++jb) {
                    const int col = colw + 16 * jb + 4 * g;
                    const int cc = col - 2 * p.D, hh = cc >> 6, dd = cc & 63;
#pragma unroll
                    for (int e = 0; e < 4; ++e)
                        p.vt[(((size_t)b * p.H + hh) * 64 + dd + e) * p.Npad + rem] = f2bf(accv[ib][4 * jb + e] + bv[jb][e]);
                }
            }
        }
        return;
    }
    char* cw0 = smem + wave * (NSL * 32 * RS);
    auto to_lds = [&](int i) __attribute__((always_inline)) {       // rows [32 i, 32 i + 32) of the wave's block: registers -> slice i % NSL
        char* cw = cw0 + (i % NSL) * (32 * RS);
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            const int ib = 2 * i + h;
#pragma unroll
            for (int jb = 0; jb < 4; ++jb) {
                const int cl = 16 * jb + 4 * g;
                const f32x16& a = accv[ib];
                f32x4 v = {a[4 * jb], a[4 * jb + 1], a[4 * jb + 2], a[4 * jb + 3]};
                v += bv[jb];                                   // (an absent bias is a row of zeros in LDS: no select per element -- 128 v_cndmask per wave and tile)
                if (EPI == EPI_QKV) v *= qs;
                if (EPI == EPI_F32) {
                    *reinterpret_cast<f32x4*>(cw + (16 * h + l15) * RS + cl * 4) = v;
                } else {
                    if (EPI == EPI_BF16 && ACT == 1) {
                        v = gelu_erf_poly4(v);
                    } else if (EPI == EPI_BF16 && ACT == 2) {
#pragma unroll
                        for (int e = 0; e < 4; ++e) v[e] = fmaxf(v[e], 0.f);
                    }
                    uint2 o = {pack_bf16x2(v[0], v[1]), pack_bf16x2(v[2], v[3])};
                    *reinterpret_cast<uint2*>(cw + (16 * h + l15) * RS + cl * 2) = o;
                }
            }
        }
    };
    // f32 outputs that accumulate (the residual stream) or add a table: the operands of a 32-row block are requested before the NEXT
    // block is converted and staged, so their round trip runs under that work (one buffer: 128 architectural VGPRs per wave)
    constexpr int NIT = 32 / RPI;
    f32x4 ov[1][EPI == EPI_F32 ? NIT : 1], tv[1][EPI == EPI_F32 ? NIT : 1];
    auto res_dst = [&](int i, int it, bool& inb) __attribute__((always_inline)) -> float* {
        const int r = it * RPI + lane / LPR;
        const int rfirst = min(m0 + i * 32, p.M - 1);
        const int row = m0 + i * 32 + r;
        inb = row < p.M;
        int b, rem;
        rowmap_at(rowmap_of(rfirst, p.rpb, p.fd_rpb), rfirst, inb ? row - rfirst : p.M - 1 - rfirst, b, rem);      // rows past M read the last valid one
        return reinterpret_cast<float*>(p.C) + ((size_t)b * p.obs + p.oro + rem) * p.ldc + colw + (lane % LPR) * 4;
    };
    auto load_res = [&](int i) __attribute__((always_inline)) {
        if constexpr (EPI == EPI_F32) {
            if (p.accumulate) {
#pragma unroll
                for (int it = 0; it < NIT; ++it) {
                    bool inb;
                    ov[0][it] = *reinterpret_cast<const f32x4*>(res_dst(i, it, inb));
                }
            }
            if (p.addtab) {
#pragma unroll
                for (int it = 0; it < NIT; ++it) {
                    const int r = it * RPI + lane / LPR, row = m0 + i * 32 + r;
                    const int rfirst = min(m0 + i * 32, p.M - 1);
                    int bt, rem;
                    rowmap_at(rowmap_of(rfirst, p.rpb, p.fd_rpb), rfirst, (row < p.M ? row : p.M - 1) - rfirst, bt, rem);
                    tv[0][it] = *reinterpret_cast<const f32x4*>(p.addtab + (size_t)(p.addtab_split ? (rem >= p.addtab_split ? 1 : 0) : rem) * p.N + colw + (lane % LPR) * 4);
                }
            }
        }
    };
    auto store_rows = [&](int i) __attribute__((always_inline)) {   // slice i % NSL -> global, row-major
        const char* cw = cw0 + (i % NSL) * (32 * RS);
        if constexpr (EPI == EPI_F32) {
            const int c16 = lane % LPR;
#pragma unroll
            for (int it = 0; it < NIT; ++it) {
                bool inb;
                float* dst = res_dst(i, it, inb);
                f32x4 v = *reinterpret_cast<const f32x4*>(cw + (it * RPI + lane / LPR) * RS + c16 * 16);
                if (p.addtab) v += tv[0][it];
                if (p.accumulate) v += ov[0][it];
                if (inb) {
                    if (p.c_store == 1) __builtin_nontemporal_store(v, reinterpret_cast<f32x4*>(dst));
                    else if (p.c_store == 2) asm volatile("global_store_dwordx4 %0, %1, off sc1\n\ts_nop 1" ::"v"(dst), "v"(v) : "memory");
                    else *reinterpret_cast<f32x4*>(dst) = v;
                }
            }
        } else {
#pragma unroll
            for (int it = 0; it < 32 / RPI; ++it) {
                const int r = it * RPI + lane / LPR, c16 = lane % LPR;
                const int row = m0 + i * 32 + r;
                const int col = colw + c16 * (16 / ES);
                const u32x4 v = *reinterpret_cast<const u32x4*>(cw + r * RS + c16 * 16);
                if (row < p.M) {
                    if (EPI == EPI_BF16) {
                        u32x4* dst = reinterpret_cast<u32x4*>(reinterpret_cast<bf16_t*>(p.C) + (size_t)row * p.ldc + col);
                        if (p.c_store == 1) __builtin_nontemporal_store(v, dst);
                        else if (p.c_store == 2) asm volatile("global_store_dwordx4 %0, %1, off sc1\n\ts_nop 1" ::"v"(dst), "v"(v) : "memory");
                        else *dst = v;
                    } else {
                        int b, rem;
                        rowmap_at(rowmap_of(m0 + i * 32, p.rpb, p.fd_rpb), m0 + i * 32, r, b, rem);
                        const int which = col >= p.D ? 1 : 0, cc = col - which * p.D;
                        const int hh = cc >> 6, dd = cc & 63;
                        u32x4* dst = reinterpret_cast<u32x4*>((which ? p.k : p.q) + (((size_t)b * p.H + hh) * p.Npad + rem) * 64 + dd);
                        if (p.c_store == 2) asm volatile("global_store_dwordx4 %0, %1, off sc1\n\ts_nop 1" ::"v"(dst), "v"(v) : "memory");
                        else *dst = v;
                    }
                }
            }
        }
    };
    to_lds(0);
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        __builtin_amdgcn_sched_barrier(0);
        load_res(i);
        if (i < 3) to_lds(i + 1);
        __builtin_amdgcn_sched_barrier(0);
        store_rows(i);
    }
}

// nn.Linear weight [N, K] bf16 -> the image the K loop loads from: for (16-row block nb, K tile kt, k step s) the 64 lanes' 16-byte fragments
// in lane order; lane l = 16 g + r holds row 16 nb + r, bytes [128 kt + 16 (2 g + s), + 16) of that row (tools/gen/gemm_dr_gen.py)
__global__ __launch_bounds__(256) void pack_w_dr_kernel(const bf16_t* __restrict__ W, bf16_t* __restrict__ Wp, int N, int K) {
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;        // one 16-byte chunk of the source per thread
    const int cpr = K / 8;                                          // chunks per row
    if (i >= (size_t)N * cpr) return;
    const int n = (int)(i / cpr), cc = (int)(i - (size_t)n * cpr);
    const int kt = cc >> 3, c = cc & 7, g = c >> 1, s = c & 1, nb = n >> 4, r = n & 15, nk = K / 64;
    const u32x4 v = *reinterpret_cast<const u32x4*>(W + (size_t)n * K + (size_t)cc * 8);
    *reinterpret_cast<u32x4*>(Wp + ((((size_t)nb * nk + kt) * 2 + s) * 64 + 16 * g + r) * 8) = v;
}
hipError_t launch_pack_w_dr(const bf16_t* W, bf16_t* Wp, int N, int K, hipStream_t s) {
    if (N % 16 != 0 || K % 64 != 0) return hipErrorInvalidValue;
    const size_t chunks = (size_t)N * (K / 8);
    hipLaunchKernelGGL(pack_w_dr_kernel, dim3((unsigned)((chunks + 255) / 256)), dim3(256), 0, s, W, Wp, N, K);
    return hipGetLastError();
}

#ifdef GEMM_DR_TRACE          // development (tools/dr_wgtrace.py on a variant build): per workgroup {start, K loop done, end} in 100 MHz ticks + where it ran
__device__ unsigned long long g_dr_trace[4096 * 4];
#endif

constexpr int DR_LDS_BIAS = 73728, DR_LDS_BYTES = DR_LDS_BIAS + 1024;       // A ring (4 x 16 KB) / epilogue slices (<= 72 KB), then the tile's bias row

template <int EPI>
__device__ __forceinline__ void gemm_dr_body(const GemmParams& p, const int bx, char* smem, const int wave) {
    constexpr int BM = 128, BN = 256;
    // `wave` arrives as an SGPR and the lane comes from the hardware: threadIdx.x itself must not live across the asm block (the pair kernel's
    // two bodies share code, and a thread index kept for the second one was a spill = a scratch segment for every wave of the launch)
    const int lane = (int)__builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, 0u));
    const int tid = wave * 64 + lane;
    const int MT = (p.M + BM - 1) / BM, NT = p.N / BN;
    int nt, mt;
    {   // grouped tile order cut into 8 contiguous runs, one per XCD (see gemm_glds_body)
        const int xcd = bx & 7, idx = bx >> 3;
        const int T = MT * NT, base = T >> 3, rem = T & 7;
        const int cnt = base + (xcd < rem ? 1 : 0);
        if (idx >= cnt) return;
        const int L = xcd * base + (xcd < rem ? xcd : rem) + idx;
        const int gsz = p.group_m * NT, gi = (int)fd_div((uint32_t)L, p.fd_gsz), within = L - gi * gsz;      // (reciprocals from the launcher: gemm_derive)
        const int gm = min(p.group_m, MT - gi * p.group_m);
        nt = (int)fd_div((uint32_t)within, gm == p.group_m ? p.fd_gm : p.fd_gml);
        mt = gi * p.group_m + (within - nt * gm);
    }
#ifdef GEMM_DR_TRACE
    const unsigned long long tr0 = __builtin_amdgcn_s_memrealtime();
    const unsigned tr_hw = __builtin_amdgcn_s_getreg(((32 - 1) << 11) | (0 << 6) | 4);       // HW_REG_HW_ID, 32 bits (simm16 = (size - 1) << 11 | offset << 6 | id)
    const unsigned tr_xcc = __builtin_amdgcn_s_getreg(((4 - 1) << 11) | (0 << 6) | 20);                              // HW_REG_XCC_ID, bits 0..3
#endif
    const int m0 = mt * BM, n0 = nt * BN;
    const char* a_base = reinterpret_cast<const char*>(p.A + (size_t)m0 * p.lda);
    const int nk = p.K / 64;
    const char* w_base = reinterpret_cast<const char*>(p.Wp + (size_t)(n0 >> 4) * nk * 1024);     // fragment-native image: 2 KB per (16-row block, K tile)

    f32x16 accv[8];                                          // pinned to a[0:15] ... a[112:127] by the asm statement's constraints
#pragma unroll
    for (int t = 0; t < 8; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) accv[t][r] = 0.f;

    float* sbias = reinterpret_cast<float*>(smem + DR_LDS_BIAS);
    if (wave == 0) *reinterpret_cast<f32x4*>(sbias + lane * 4) = *reinterpret_cast<const f32x4*>((p.bias ? p.bias + n0 : reinterpret_cast<const float*>(g_zero_page)) + lane * 4);
    // every "s" operand of the block is made wave-uniform EXPLICITLY: with enough scalar values live across the statement hipcc hands an "s"
    // constraint a VGPR (seen when an epilogue experiment grew: `s_mov_b64 s[56:57], v[118:119]`, which -S prints and only the assembler refuses)
    auto pin64 = [](const char* q) __attribute__((always_inline)) {
        const uint64_t u = reinterpret_cast<uint64_t>(q);
        const uint32_t lo = __builtin_amdgcn_readfirstlane((uint32_t)u), hi = __builtin_amdgcn_readfirstlane((uint32_t)(u >> 32));
        return reinterpret_cast<const char*>(((uint64_t)hi << 32) | lo);
    };
    a_base = pin64(a_base);
    w_base = pin64(w_base);
    const uint32_t lds0 = __builtin_amdgcn_readfirstlane((uint32_t)(uintptr_t)(__attribute__((address_space(3))) char*)smem);
    const int lda2 = __builtin_amdgcn_readfirstlane(p.lda * 2), ldw2 = __builtin_amdgcn_readfirstlane(nk * 2048);
    const int rmax = __builtin_amdgcn_readfirstlane(min(BM, p.M - m0) - 1);      // rows past M re-read the last valid one (never stored)
    const int nk_s = __builtin_amdgcn_readfirstlane(nk);
    (void)a_base; (void)w_base; (void)lds0; (void)lda2; (void)ldw2; (void)rmax; (void)nk_s;      // (the host pass does not see the asm statement that reads them)
#if __HIP_DEVICE_COMPILE__              // the host pass of hipcc parses kernel bodies too and knows no gfx950 register names
    asm volatile(
#include "gemm_dr_asm.inc"
        : "+{a[0:15]}"(accv[0]), "+{a[16:31]}"(accv[1]), "+{a[32:47]}"(accv[2]), "+{a[48:63]}"(accv[3]),
          "+{a[64:79]}"(accv[4]), "+{a[80:95]}"(accv[5]), "+{a[96:111]}"(accv[6]), "+{a[112:127]}"(accv[7])
        : [tid] "v"(tid), [ab] "s"(a_base), [wb] "s"(w_base), [lda2] "s"(lda2), [ldw2] "s"(ldw2), [rmax] "s"(rmax), [nk] "s"(nk_s), [lds] "s"(lds0), [wave] "s"(wave)
        : "memory", "scc", GEMM_DR_SGPRS, GEMM_DR_VGPRS);
#endif
    // the epilogue takes its lane index from the hardware again: the block clobbers v0..v113, and every per-lane value the compiler keeps
    // across it has to live in the 14 registers above (one more was a spill = scratch for the whole kernel)
#ifdef GEMM_DR_TRACE
    const unsigned long long tr1 = __builtin_amdgcn_s_memrealtime();
#endif
    const int lane_e = (int)__builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, 0u));
#ifdef DR_NOEPI          // timing probe (tools/probes/dr_epi_probe.py): the tile ends behind its K loop, nothing is stored
    if (p.M > 0) { asm volatile("" :: "a"(accv[0]), "a"(accv[7])); return; }
#endif
    if (EPI == EPI_BF16 && p.act == 1) gemm_epilogue_dr<EPI, 1>(p, accv, smem, sbias, m0, n0, lane_e, wave);
    else if (EPI == EPI_BF16 && p.act == 2) gemm_epilogue_dr<EPI, 2>(p, accv, smem, sbias, m0, n0, lane_e, wave);
    else gemm_epilogue_dr<EPI, 0>(p, accv, smem, sbias, m0, n0, lane_e, wave);
#ifdef GEMM_DR_TRACE
    if (threadIdx.x == 0 && bx < 4096) {                     // (no wait for the output stores: the wave ends with them in flight, as in the product)
        g_dr_trace[bx * 4 + 0] = tr0; g_dr_trace[bx * 4 + 1] = tr1; g_dr_trace[bx * 4 + 2] = __builtin_amdgcn_s_memrealtime();
        g_dr_trace[bx * 4 + 3] = ((unsigned long long)tr_xcc << 32) | tr_hw;
    }
#endif
}

template <int EPI>
__global__ __launch_bounds__(256, 2) void gemm_dr_kernel(const GemmParams p) {
    kernarg_warm<sizeof(GemmParams)>();
    extern __shared__ __attribute__((aligned(16))) char smem[];
    gemm_dr_body<EPI>(p, blockIdx.x, smem, __builtin_amdgcn_readfirstlane((int)threadIdx.x >> 6));
}

template <int EPI>
static hipError_t launch_dr(const GemmParams& p_in, hipStream_t s) {
    GemmParams p = p_in;
    if (!p.Wp || p.N % 256 != 0 || p.K < 64 || p.K % 64 != 0 || p.splitk > 1 || p.conv_F != 0 || p.groups > 1) return hipErrorInvalidValue;
    const int MT = (p.M + 127) / 128, NT = p.N / 256;
    p.group_m = MT >= 16 ? 8 : MT;
    const int forced_gm = tune_get(p.tune, &uvl_tuning::gemm_gm, -1);
    if (forced_gm > 0) p.group_m = forced_gm;
    gemm_derive(p, 128, 256);
    const int nblk = 8 * ((MT * NT + 7) / 8);
    constexpr size_t lds = DR_LDS_BYTES;
    auto kern = gemm_dr_kernel<EPI>;
    static bool attr_done = false;
    if (!attr_done) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) return e;
        attr_done = true;
    }
    static char name[48];
    if (!name[0]) snprintf(name, sizeof(name), "gemm_dr_kernel<%d>", EPI);
    g_last_kernel = name;
    hipLaunchKernelGGL(kern, dim3(nblk), dim3(256), lds, s, p);
    return hipGetLastError();
}

// Two GEMMs of the same epilogue in one launch (many-sequence frames: the text branch's QKV / intermediate GEMM rides behind the visual
// one's).  The RIDER's workgroups come first, [0, blocks_b): its weights are read once per frame by three row tiles, so its K loop runs at
// HBM latency and a rider tile lives longer than a visual one -- dispatched last it was the launch's tail (+6 us on the QKV launch of
// 8 UVLTrack-L sequences), dispatched first it ends under the visual rounds.  blocks_b is a multiple of 8: both tile maps keep their
// workgroup -> XCD relation.  Two calls, not a selected reference (see gemm_glds_pair_kernel).
template <int EPI>
__global__ __launch_bounds__(256, 2) void gemm_dr_pair_kernel(const GemmParams pa, const GemmParams pb, const int blocks_b) {
    kernarg_warm<2 * sizeof(GemmParams) + 8>();
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int wave = __builtin_amdgcn_readfirstlane((int)threadIdx.x >> 6);
    if ((int)blockIdx.x >= blocks_b) gemm_dr_body<EPI>(pa, (int)blockIdx.x - blocks_b, smem, wave);
    else gemm_dr_body<EPI>(pb, blockIdx.x, smem, wave);
}

static bool dr_ok(const GemmParams& p) {
    return p.Wp && p.M > 0 && p.N % 256 == 0 && p.K >= 64 && p.K % 64 == 0 && p.splitk <= 1 && p.conv_F == 0 && p.groups <= 1;
}

static int dr_grid(GemmParams& p) {
    const int MT = (p.M + 127) / 128, NT = p.N / 256;
    p.group_m = MT >= 16 ? 8 : MT;
    const int forced_gm = tune_get(p.tune, &uvl_tuning::gemm_gm, -1);
    if (forced_gm > 0) p.group_m = forced_gm;
    gemm_derive(p, 128, 256);
    return 8 * ((MT * NT + 7) / 8);
}

template <int EPI>
static hipError_t launch_dr_pair(const GemmParams& a_in, const GemmParams& b_in, hipStream_t s) {
    GemmParams a = a_in, b = b_in;
    if (!dr_ok(a) || !dr_ok(b)) return hipErrorInvalidValue;
    const int ba = dr_grid(a), bb = dr_grid(b);
    auto kern = gemm_dr_pair_kernel<EPI>;
    static bool attr_done = false;
    if (!attr_done) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)DR_LDS_BYTES);
        if (e != hipSuccess) return e;
        attr_done = true;
    }
    static char name[48];
    if (!name[0]) snprintf(name, sizeof(name), "gemm_dr_pair_kernel<%d>", EPI);
    g_last_kernel = name;
    hipLaunchKernelGGL(kern, dim3(ba + bb), dim3(256), DR_LDS_BYTES, s, a, b, bb);
    return hipGetLastError();
}

bool gemm_dr_pairable(const GemmParams& a, const GemmParams& b) { return dr_ok(a) && dr_ok(b) && a.epi == b.epi; }

hipError_t launch_gemm_dr_pair(const GemmParams& a, const GemmParams& b, hipStream_t s) {
    if (a.epi != b.epi) return hipErrorInvalidValue;
    switch (a.epi) {
        case EPI_BF16: return launch_dr_pair<EPI_BF16>(a, b, s);
        case EPI_F32: return launch_dr_pair<EPI_F32>(a, b, s);
        case EPI_QKV: return launch_dr_pair<EPI_QKV>(a, b, s);
    }
    return hipErrorInvalidValue;
}

hipError_t launch_gemm_dr(const GemmParams& p, int epi, hipStream_t s) {
    switch (epi) {
        case EPI_BF16: return launch_dr<EPI_BF16>(p, s);
        case EPI_F32: return launch_dr<EPI_F32>(p, s);
        case EPI_QKV: return launch_dr<EPI_QKV>(p, s);
    }
    return hipErrorInvalidValue;
}

}  // namespace uvl

#ifdef GEMM_DR_TRACE
extern "C" int uvl_debug_dr_trace(unsigned long long* dst, int n) { return (int)hipMemcpyFromSymbol(dst, HIP_SYMBOL(uvl::g_dr_trace), (size_t)n * 4 * sizeof(unsigned long long)); }
#endif
