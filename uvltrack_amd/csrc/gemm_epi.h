// Epilogue shared by the GEMM kernels of gemm.hip and gemm_w4.hip (two translation units: they are built with different register-form
// options, see uvltrack_amd/build.py).
#pragma once
#include "common.h"
#include "kernels.h"

namespace uvl {

enum { EPI_BF16 = 0, EPI_F32 = 1, EPI_QKV = 2 };

// Epilogue of the pipelined kernel, which computes the tile TRANSPOSED (W fragments feed the MFMA row operand, A
// fragments the column operand): lane l holds output row m = l & 31 and, per register quad q = r >> 2, the four
// consecutive columns n = 8 q + 4 (l >> 5) + (r & 3).  Bias / activation are applied in registers, then each wave
// transposes 32 rows at a time through its own slice of the (now idle) LDS ring and writes them back row-major, 16 bytes
// per lane: every store instruction covers whole 128-byte lines (measured on the fc1 shape at M = 17.7k,
// tools/probes/gemm_probe.hip: 112 us, against 124 us for element-per-lane stores and 141 us for 8 bytes per lane on
// 32 different rows).  V^T of the QKV projection is token-contiguous and is stored straight from registers.
//
// The bias of the lane's columns is fetched by gemm_bias_preload at kernel start (it does not depend on the product): as
// conditional loads inside the epilogue they were four dependent L2 round trips per launch, each behind its own s_waitcnt.
static __device__ uint32_t g_zero_page[256];     // 1 KB of zeros: DMA source of out-of-image conv taps (zero padding), absent bias

template <int TN>
__device__ __forceinline__ void gemm_bias_preload(const GemmParams& p, int colw, int lane, int g, int sk, f32x4 (&bv)[TN][4]) {
    const float* bp = (p.bias && sk == 0) ? p.bias + (size_t)g * p.N + colw : reinterpret_cast<const float*>(g_zero_page);
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
        for (int q = 0; q < 4; ++q) bv[j][q] = *reinterpret_cast<const f32x4*>(bp + j * 32 + 8 * q + 4 * (lane >> 5));
}

// L16: the accumulators of a 32 x 32 block were produced by 16x16x32 MFMAs -- registers 4 (2 hi + hj) .. + 3 of the block hold its 16 x 16
// sub-block (hi, hj): lane l has row 16 hi + (l & 15) and the four consecutive columns 16 hj + 4 (l >> 4) + r; bv[j][hj] is the bias of those
// columns.  Everything behind the staging store (row write-out, residual / table operands, V^T) is the same.
// PRE (f32 in-place residual form only): the residual rows were requested by the K loop (gemm_pipe128_body: sixteen 16-byte loads per lane spread
// over eight K tiles) and arrive in `res` -- block i, row group it at res[i * (32 / RPI) + it], exactly the element the epilogue would load here.
template <int TM, int TN, int WM, int WN, int EPI, int NW, bool L16 = false, bool PRE = false>
__device__ __forceinline__ void gemm_epilogue_lds(const GemmParams& p, f32x16 (&acc)[TM][TN], char* smem, int m0, int n0, int wm, int wn,
                                                  int lane, int wave, int g, int sk, const f32x4 (&bv)[TN][4], const f32x4* res = nullptr) {
    constexpr int ES = (EPI == EPI_F32) ? 4 : 2;               // output element size
    constexpr int RS = WN * ES + 16;                            // padded LDS row stride
    constexpr int LPR = WN * ES / 16;                           // lanes per row on the way out
    constexpr int RPI = 64 / LPR;                               // rows per store instruction
    const int colw = n0 + wn * WN;                              // first column of this wave's sub-tile
    const bool vpart = EPI == EPI_QKV && colw >= 2 * p.D;       // wave-uniform: D % 64 == 0 and WN divides 64
    const float qs = (EPI == EPI_QKV && colw < p.D) ? p.q_scale : 1.0f;   // q columns carry the attention's log2(e)/8 (wave-uniform)
    __builtin_amdgcn_s_barrier();                               // every wave has finished reading the ring
    char* cw = smem + wave * (32 * RS);
#pragma unroll
    for (int i = 0; i < TM; ++i) {
        if (vpart) {
#pragma unroll
            for (int hi = 0; hi < (L16 ? 2 : 1); ++hi) {
                const int rfirst = m0 + wm * WM + i * 32, rl = L16 ? 16 * hi + (lane & 15) : (lane & 31);
                const int rowl = rfirst + rl;
                if (rowl < p.M) {
                    int b, rem;
                    rowmap_at(rowmap_of(rfirst, p.rpb, p.fd_rpb), rfirst, rl, b, rem);
#pragma unroll
                    for (int j = 0; j < TN; ++j)
#pragma unroll
                        for (int q = 0; q < (L16 ? 2 : 4); ++q) {
                            const int col = colw + j * 32 + (L16 ? 16 * q + 4 * (lane >> 4) : 8 * q + 4 * (lane >> 5));
                            const int cc = col - 2 * p.D, hh = cc >> 6, dd = cc & 63;
                            const int r0 = L16 ? 4 * (2 * hi + q) : 4 * q;
#pragma unroll
                            for (int e = 0; e < 4; ++e)
                                p.vt[(((size_t)b * p.H + hh) * 64 + dd + e) * p.Npad + rem] = f2bf(acc[i][j][r0 + e] + bv[j][q][e]);
                        }
                }
            }
            continue;
        }
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                // 32x32x16 blocks: register quad q = columns 8 q + 4 (l >> 5) of row l & 31; 16x16x32: q = (hi, hj)
                const int hi = q >> 1, hj = q & 1;
                const int rl = L16 ? 16 * hi + (lane & 15) : (lane & 31);
                const int cl = j * 32 + (L16 ? 16 * hj + 4 * (lane >> 4) : 8 * q + 4 * (lane >> 5));
                f32x4 v = {acc[i][j][4 * q], acc[i][j][4 * q + 1], acc[i][j][4 * q + 2], acc[i][j][4 * q + 3]};
                v += bv[j][L16 ? hj : q];                       // (bv is zero where there is no bias to add -- absent, or a split-K slice other than 0 -- so no select per element)
                if (EPI == EPI_QKV) v *= qs;
                if (EPI == EPI_F32) {
                    *reinterpret_cast<f32x4*>(cw + rl * RS + cl * 4) = v;
                } else {
                    if (EPI == EPI_BF16 && p.act == 1) {
                        v = gelu_erf_poly4(v);
                    } else if (EPI == EPI_BF16 && p.act == 2) {
#pragma unroll
                        for (int e = 0; e < 4; ++e) v[e] = fmaxf(v[e], 0.f);
                    }
                    uint2 o = {pack_bf16x2(v[0], v[1]), pack_bf16x2(v[2], v[3])};
                    *reinterpret_cast<uint2*>(cw + rl * RS + cl * 2) = o;
                }
            }
        if constexpr (EPI == EPI_F32) {
            // the table / residual operands of a chunk of row groups are requested together (one wait), then added in the same order;
            // the chunk is all 32 rows except where the accumulators already fill half the register file (128 x 64 per wave)
            constexpr int NIT = 32 / RPI;
            constexpr int CH = (TM * TN >= 8 && NIT > 4) ? 4 : NIT;
            const int c16 = lane % LPR, col = colw + c16 * (16 / ES);
#pragma unroll
            for (int h0 = 0; h0 < NIT; h0 += CH) {
                f32x4 v[CH], tv[CH], ov[CH];
                float* dst[CH];
                bool inb[CH];
                const int rfirst = min(m0 + wm * WM + i * 32, p.M - 1);      // a block past M reads (and never stores) the last valid row
                const RowMap rmap = rowmap_of(rfirst, p.rpb, p.fd_rpb);
#pragma unroll
                for (int it = 0; it < CH; ++it) {
                    const int r = (h0 + it) * RPI + lane / LPR;
                    const int row = m0 + wm * WM + i * 32 + r;
                    inb[it] = row < p.M;
                    int b, rem;
                    rowmap_at(rmap, rfirst, (inb[it] ? row : p.M - 1) - rfirst, b, rem);
                    v[it] = *reinterpret_cast<const f32x4*>(cw + r * RS + c16 * 16);
                    dst[it] = reinterpret_cast<float*>(p.C) + (size_t)sk * p.part_stride + ((size_t)b * p.obs + p.oro + rem) * p.ldc + (size_t)g * p.N + col;
                    if (p.addtab) tv[it] = *reinterpret_cast<const f32x4*>(p.addtab + (size_t)(p.addtab_split ? (rem >= p.addtab_split ? 1 : 0) : rem) * p.N + col);
                }
                if constexpr (PRE) {
#pragma unroll
                    for (int it = 0; it < CH; ++it) ov[it] = res[i * NIT + h0 + it];
                } else if (p.accumulate) {
#pragma unroll
                    for (int it = 0; it < CH; ++it) ov[it] = *reinterpret_cast<const f32x4*>(dst[it]);
                }
#pragma unroll
                for (int it = 0; it < CH; ++it) {
                    if (p.addtab) v[it] += tv[it];
                    if (PRE || p.accumulate) v[it] += ov[it];
                    if (inb[it]) {
                        if (p.c_store == 1) __builtin_nontemporal_store(v[it], reinterpret_cast<f32x4*>(dst[it]));
                        else if (p.c_store == 2) asm volatile("global_store_dwordx4 %0, %1, off sc1\n\ts_nop 1" ::"v"(dst[it]), "v"(v[it]) : "memory");
                        else *reinterpret_cast<f32x4*>(dst[it]) = v[it];
                    }
                }
            }
            continue;
        }
#pragma unroll
        for (int it = 0; it < 32 / RPI; ++it) {
            const int r = it * RPI + lane / LPR, c16 = lane % LPR;
            const int row = m0 + wm * WM + i * 32 + r;
            const int col = colw + c16 * (16 / ES);
            if (EPI == EPI_F32) {
            } else {
                const u32x4 v = *reinterpret_cast<const u32x4*>(cw + r * RS + c16 * 16);
                if (row < p.M) {
                    if (EPI == EPI_BF16) {
                        *reinterpret_cast<u32x4*>(reinterpret_cast<bf16_t*>(p.C) + (size_t)row * p.ldc + (size_t)g * p.N + col) = v;
                    } else {
                        int b, rem;
                        rowmap_at(rowmap_of(m0 + wm * WM + i * 32, p.rpb, p.fd_rpb), m0 + wm * WM + i * 32, r, b, rem);
                        const int which = col >= p.D ? 1 : 0, cc = col - which * p.D;
                        const int hh = cc >> 6, dd = cc & 63;
                        *reinterpret_cast<u32x4*>((which ? p.k : p.q) + (((size_t)b * p.H + hh) * p.Npad + rem) * 64 + dd) = v;
                    }
                }
            }
        }
    }
}

}  // namespace uvl
