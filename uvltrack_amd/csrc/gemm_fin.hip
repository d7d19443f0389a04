// Residual GEMM of a one- / two-sequence frame that FINISHES its rows inside the launch (round 6):   x (+)= A[M,K] W[N,K]^T + b   with N = D,
// i.e. attn.proj / mlp.fc2 (block.py:29-32, 44; backbones/utils.py:61), BERT attention.output / output with their post-LayerNorm residual
// (bert_backbone.py:335-339, 376-380) and the patch embedding (mae_vit.py:92-100, 203-215).
//
// Rounds 1-5 cut K of these GEMMs into slices over WORKGROUPS (108 tiles of 64 x 64 leave 148 CUs idle) and left f32 slabs that the next LayerNorm
// launch folded -- which is why 25 LayerNorm launches of 5.4 us sat in the dependency chain of the 706-us frame.  Finishing a tile needs every K slice in one
// place; tools/probes/combine2_probe.hip measured the ways of getting them there on MI355X (fc2 of one UVLTrack-B sequence, us per launch; slab GEMM alone 12.2):
//   slabs + arrival counter per tile / per wave, last arriver adds (write-through stores, sc1 loads)     15.5 / 15.4      (proj: 9.6 against 4.6)
//   returning 64-bit atomic swaps on a sentinel-armed slab (one round trip)                              24.2
//   ONE workgroup of eight waves per tile, the two K halves on its two wave groups, combined in LDS      14.5             (proj: 6.4; UVLTrack-L fc2: 17.5 against 18.7 for slabs)
// Anything that crosses workgroups pays three dependent device-scope round trips (store acknowledged -> counter -> partner's slab); the eight-wave workgroup
// pays none: both halves of a tile's K loop run on the same CU (two waves per SIMD, each group with its own four-stage LDS-DMA ring, 128 KB of LDS), group 1
// parks its 32 x 32 blocks in the idle ring, and the eight waves share the epilogue -- every wave finishes 16 rows of its quadrant: residual (optionally
// LayerNorm of the stored pre-norm row), table term, f32 row, bf16 row and the row's partial statistics for the LayerNorm-folded GEMM that reads it next
// (fold.h).  Fixed summation order (group 0 + group 1, + bias, + table, + residual): bit-reproducible, no atomics, no slabs, no consumer-side fold.
#include "common.h"
#include "kernels.h"
#include "gemm_epi.h"
#include "fold.h"

namespace uvl {

template <int N_> __device__ __forceinline__ void fin_wait_vmcnt() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N_) : "memory"); }

// One tile = 64 rows x BN columns, eight waves as G wave groups of 8 / G waves; group g runs the K range [g K / G, (g + 1) K / G) through its own NS-stage ring.
//   BN = 64, G = 2, NS = 4 (128 KB of LDS): four waves per group, the 2 x 2 quadrants of the tile            -- tiles = ceil(M / 64) x N / 64
//   BN = 32, G = 4, NS = 3 (144 KB):        two waves per group, the tile's two 32 x 32 blocks               -- twice the workgroups, half the K loop per group
// The second form is for grids that would leave most of the chip idle (one UVLTrack-B sequence: 108 tiles of 64 x 64) and for the text rider (12 -> 24 workgroups that
// stream BERT weights from HBM: its K loop is a latency chain per workgroup, so the number of workgroups is its bandwidth).
template <bool NTW, int BN, int G, int NS>
__device__ __forceinline__ void gemm_fin_body(const GemmParams& p, const int bx, char* smem_all) {
    constexpr int WPG = 8 / G;                       // waves per group = 32 x 32 blocks of the tile
    static_assert(WPG == 2 * (BN / 32) && (BN == 64 || BN == 32), "tile geometry");
    constexpr int STAGE = (64 + BN) * 128, LPT = (64 + BN) / 8 / WPG, LPT_A = 8 / WPG, RS = 32 * 4 + 16;
    constexpr int ITS = 32 / G / 8;                  // row groups of 8 a wave finishes: its share of a block's 32 rows
    static_assert(G * NS * STAGE <= 160 * 1024 && 8 * 32 * RS <= G * NS * STAGE, "LDS");
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave_all = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int grp = wave_all / WPG, wave = wave_all % WPG;
    const int wm = wave / (BN / 32), wn = wave % (BN / 32);
    char* smem = smem_all + grp * (NS * STAGE);
    const int MT = (p.M + 63) >> 6, NT = p.N / BN;
    // tile order: N-major (the M tiles of a weight panel adjacent), cut into 8 contiguous runs, one per XCD (workgroup b runs on XCD b % 8; speed only)
    const int xcd = bx & 7, idx = bx >> 3;
    const int T = MT * NT, base = T >> 3, rem8 = T & 7;
    if (idx >= base + (xcd < rem8 ? 1 : 0)) return;
    const int L = xcd * base + (xcd < rem8 ? xcd : rem8) + idx;
    const int nt = (int)fd_div((uint32_t)L, p.fd_mt), mt = L - nt * MT;
    const int m0 = mt * 64, n0 = nt * BN;
    const int kspan = p.K / G, kbase = grp * kspan, nk = kspan >> 6;

    // ---- what the finishing step of this wave needs and the K loop does not produce (addresses now; the loads go out behind the prologue's tiles, see below) ----
    // wave (grp, wave) finishes rows grp * (32 / G) + it * 8 + (lane >> 3) of block `wave`: 16 bytes (4 columns) per lane, 8 lanes = one 128-byte line
    const int c16 = lane & 7;
    const int col = n0 + wn * 32 + c16 * 4;
    const int np = p.N >> 5;
    const float* zero = reinterpret_cast<const float*>(g_zero_page);
    bool inb[ITS];
    size_t xrow[ITS], nrow[ITS];
    int grow[ITS];
    f32x4 ov[ITS], tv[ITS];
    f32x2 rs[ITS][4];
    const float *ov_p[ITS], *tv_p[ITS], *rs_p[ITS][4];
    const bool recon = p.res_st != nullptr;
#pragma unroll
    for (int it = 0; it < ITS; ++it) {
        const int row = m0 + wm * 32 + grp * (32 / G) + it * 8 + (lane >> 3);
        inb[it] = row < p.M;
        const int rc = inb[it] ? row : p.M - 1;
        const int b = (int)fd_div((uint32_t)rc, p.fd_rpb), rem = rc - b * p.rpb;
        grow[it] = rc;
        xrow[it] = (size_t)b * p.obs + p.oro + rem;
        nrow[it] = (size_t)b * p.xn_bs + p.xn_ro + rem;
        ov_p[it] = p.accumulate ? reinterpret_cast<const float*>(p.C) + xrow[it] * p.ldc + col : zero;
        tv_p[it] = p.addtab ? p.addtab + (size_t)(p.addtab_split ? (rem >= p.addtab_split ? 1 : 0) : rem) * p.N + col : zero;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const int j = c16 + 8 * k;
            rs_p[it][k] = (recon && j < np) ? p.res_st + st_off(j, nrow[it], (size_t)p.st_rows) : zero;
        }
    }
    const float* bias_p = p.bias ? p.bias + col : zero;
    const float* rg_p = recon ? p.res_g + col : zero;
    const float* rb_p = recon ? p.res_b + col : zero;
    f32x4 bias4, rg4, rb4;
    constexpr int NLATE = ITS * 6 + 3;               // loads per lane requested behind the prologue's LDS-DMA (see below)

    // ---- K loop of this wave group: gemm_glds_body's loop for one 32 x 32 block per wave (fragment reads first, the LDS-DMA of the next free stage under their
    //      round trip, four MFMAs); instruction i of wave w fills stage rows [8 (w + WPG i), + 8): A rows first, then W rows ----
    uint32_t loff[LPT];
    const char* a_base = reinterpret_cast<const char*>(p.A + (size_t)m0 * p.lda + kbase);
    const char* w_base = reinterpret_cast<const char*>(p.W + (size_t)n0 * p.ldw + kbase);
#pragma unroll
    for (int i = 0; i < LPT; ++i) {
        const int r = 8 * (wave + WPG * i) + (lane >> 3);
        const int chunk = (lane & 7) ^ ((r >> 1) & 7);
        if (i < LPT_A) {
            int gm = m0 + r;
            gm = gm < p.M ? gm : p.M - 1;
            loff[i] = (uint32_t)(gm - m0) * (uint32_t)p.lda * 2u + (uint32_t)chunk * 16u;
        } else {
            loff[i] = (uint32_t)(r - 64) * (uint32_t)p.ldw * 2u + (uint32_t)chunk * 16u;
        }
    }
    auto issue = [&](int kt) __attribute__((always_inline)) {
        char* st = smem + (kt % NS) * STAGE;
        auto pin = [](const char* q) __attribute__((always_inline)) {
            const uint64_t u = reinterpret_cast<uint64_t>(q);
            const uint32_t lo = __builtin_amdgcn_readfirstlane((uint32_t)u), hi = __builtin_amdgcn_readfirstlane((uint32_t)(u >> 32));
            return reinterpret_cast<const char*>(((uint64_t)hi << 32) | lo);
        };
        const char* ab = pin(a_base + (size_t)kt * 128);
        const char* wb = pin(w_base + (size_t)kt * 128);
#pragma unroll
        for (int i = 0; i < LPT; ++i) {
            const char* gp = (i < LPT_A ? ab : wb) + loff[i];
            if (NTW && i >= LPT_A)
                __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)gp, (__attribute__((address_space(3))) void*)(st + (wave + WPG * i) * 1024), 16, 0, 2);
            else
                __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)gp, (__attribute__((address_space(3))) void*)(st + (wave + WPG * i) * 1024), 16, 0, 0);
        }
    };
    f32x16 acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;
#pragma unroll
    for (int t = 0; t < NS - 1; ++t)
        if (t < nk) issue(t);
    // what the finishing step needs and the K loop does not produce (residual rows, table term, the residual's partials, bias, the residual LayerNorm's gamma / beta):
    // requested BEHIND the prologue's tiles -- in front of them they delayed the first tile of every workgroup -- and older than every tile requested inside the loop; the
    // counted waits for tiles 0 .. NS - 2 allow for these NLATE loads (inline asm: hipcc neither moves them across the LDS-DMA nor waits for them; the last tile's vmcnt(0) does)
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int it = 0; it < ITS; ++it) {
        asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(ov[it]) : "v"(ov_p[it]) : "memory");
        asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(tv[it]) : "v"(tv_p[it]) : "memory");
#pragma unroll
        for (int k = 0; k < 4; ++k) asm volatile("global_load_dwordx2 %0, %1, off" : "=v"(rs[it][k]) : "v"(rs_p[it][k]) : "memory");
    }
    asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(bias4) : "v"(bias_p) : "memory");
    asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(rg4) : "v"(rg_p) : "memory");
    asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(rb4) : "v"(rb_p) : "memory");
    __builtin_amdgcn_sched_barrier(0);
    for (int kt = 0; kt < nk; ++kt) {
        const int ahead = nk - 1 - kt;
        if (kt <= NS - 2 && ahead >= NS - 2) fin_wait_vmcnt<LPT * (NS - 2) + NLATE>();
        else if (kt <= NS - 2 && NS > 3 && ahead == 1) fin_wait_vmcnt<LPT + NLATE>();
        else if (ahead >= NS - 2) fin_wait_vmcnt<LPT * (NS - 2)>();
        else if (NS > 3 && ahead == 1) fin_wait_vmcnt<LPT>();
        else fin_wait_vmcnt<0>();
        __builtin_amdgcn_s_barrier();
        const char* sA = smem + (kt % NS) * STAGE;
        const char* sB = sA + 64 * 128;
        bf16x8 af[4], bfr[4];
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {
            const int chunk = ks * 2 + (lane >> 5);
            af[ks] = *reinterpret_cast<const bf16x8*>(sA + swz128(wm * 32 + (lane & 31), chunk));
            bfr[ks] = *reinterpret_cast<const bf16x8*>(sB + swz128(wn * 32 + (lane & 31), chunk));
        }
        __builtin_amdgcn_sched_barrier(0);
        if (kt + NS - 1 < nk) issue(kt + NS - 1);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(bfr[ks], af[ks], acc, 0, 0, 0);      // transposed block: lane = row, register quads = columns
    }

    // ---- every group parks its 32 x 32 blocks row-major in the (now idle) rings: block (grp, wave) at smem_all + (grp * WPG + wave) * 32 * RS ----
    __builtin_amdgcn_s_barrier();                           // every wave has finished reading the rings
    {
        char* cw = smem_all + (grp * WPG + wave) * (32 * RS);
        const int rl = lane & 31;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const f32x4 v = {acc[4 * q], acc[4 * q + 1], acc[4 * q + 2], acc[4 * q + 3]};
            *reinterpret_cast<f32x4*>(cw + rl * RS + (8 * q + 4 * (lane >> 5)) * 4) = v;
        }
    }
    __syncthreads();
    // ---- finish: this wave's 32 / G rows of block `wave`, the groups' partial sums added in group order ----
#pragma unroll
    for (int it = 0; it < ITS; ++it) {
        const int r = grp * (32 / G) + it * 8 + (lane >> 3);
        f32x4 v = *reinterpret_cast<const f32x4*>(smem_all + wave * (32 * RS) + r * RS + c16 * 16);
#pragma unroll
        for (int gg = 1; gg < G; ++gg) v += *reinterpret_cast<const f32x4*>(smem_all + (gg * WPG + wave) * (32 * RS) + r * RS + c16 * 16);
        v += bias4;
        v += tv[it];
        f32x4 res = ov[it];
        if (recon) {
            // the stored row is pre-norm: the residual is its LayerNorm (statistics from the row's partials, all np of them across the octet)
            float s1 = (rs[it][0][0] + rs[it][1][0]) + (rs[it][2][0] + rs[it][3][0]), s2 = (rs[it][0][1] + rs[it][1][1]) + (rs[it][2][1] + rs[it][3][1]);
            s1 = oct_sum(s1);
            s2 = oct_sum(s2);
            float mean, rstd;
            st_finish(s1, s2, p.N, p.res_eps, mean, rstd);
#pragma unroll
            for (int e = 0; e < 4; ++e) res[e] = (res[e] - mean) * rstd * rg4[e] + rb4[e];
            if (p.res_copy && inb[it]) *reinterpret_cast<f32x4*>(p.res_copy + (size_t)grow[it] * p.N + col) = res;
        }
        v += res;
        const uint32_t lo = pack_bf16x2(v[0], v[1]), hi = pack_bf16x2(v[2], v[3]);
        float s1, s2;
        st_of4(v[0], v[1], v[2], v[3], s1, s2);
        s1 = oct_sum(s1);
        s2 = oct_sum(s2);
        if (inb[it]) {
            float* xp = reinterpret_cast<float*>(p.C) + xrow[it] * p.ldc + col;
            if (p.c_store == 2) {          // write-through (sc1): the next launch reads these rows on other XCDs; nothing of them waits dirty in this L2 for the end-of-kernel write-back
                asm volatile("global_store_dwordx4 %0, %1, off sc1" ::"v"(xp), "v"(v) : "memory");
                if (p.xn) {
                    const uint2 pk = uint2{lo, hi};
                    asm volatile("global_store_dwordx2 %0, %1, off sc1" ::"v"(p.xn + nrow[it] * p.N + col), "v"(pk) : "memory");
                }
            } else {
                *reinterpret_cast<f32x4*>(xp) = v;
                if (p.xn) *reinterpret_cast<uint2*>(p.xn + nrow[it] * p.N + col) = uint2{lo, hi};
            }
            if (p.st_out && c16 == 0) *reinterpret_cast<float2*>(p.st_out + st_off((n0 >> 5) + wn, nrow[it], (size_t)p.st_rows)) = float2{s1, s2};
        }
    }
}

// BN = 32: the 64 x 32 tile on four wave groups; 64: 64 x 64 on two (the template arguments are the tile widths, so that profilers print the names the library reports)
template <bool NTW, int BN>
__device__ __forceinline__ void gemm_fin_tile(const GemmParams& p, const int bx, char* smem) {
    static_assert(BN == 32 || BN == 64, "tile width");
    if constexpr (BN == 32) gemm_fin_body<NTW, 32, 4, 3>(p, bx, smem);
    else gemm_fin_body<NTW, 64, 2, 4>(p, bx, smem);
}

template <int BN>
__global__ __launch_bounds__(512) void gemm_fin_kernel(const GemmParams p) {
    kernarg_warm<sizeof(GemmParams) + 64>();
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const uint32_t pfs = prefetch_issue<512>(p.pf, p.pf_bytes, blockIdx.x, gridDim.x);
    gemm_fin_tile<false, BN>(p, blockIdx.x, smem);
    prefetch_retire(pfs);
}

// The same with the text branch's GEMM of the same kind as a second problem (the rider: its workgroups first, its weight tiles non-temporal -- see gemm_glds_pair_kernel)
template <int BNA, int BNB>
__global__ __launch_bounds__(512) void gemm_fin_pair_kernel(const GemmParams pa, const GemmParams pb, const int blocks_b) {
    kernarg_warm<2 * sizeof(GemmParams) + 8 + 64>();
    extern __shared__ __attribute__((aligned(16))) char smem[];
    uint32_t pfs = prefetch_issue<512>(pa.pf, pa.pf_bytes, blockIdx.x, gridDim.x);
    pfs = prefetch_issue_xcd<512>(pfs, pb.pf2, pb.pf2_tiles, pb.pf2_lp, blockIdx.x, gridDim.x);
    if ((int)blockIdx.x < blocks_b) gemm_fin_tile<true, BNB>(pb, blockIdx.x, smem);
    else gemm_fin_tile<false, BNA>(pa, (int)blockIdx.x - blocks_b, smem);
    prefetch_retire(pfs);
}

// ------------------------------------------------------------------------------------------------
// The same workgroup shape for a layer of the box head's 3x3 conv towers at one sequence (heads/utils.py:126-131 with BatchNorm folded,
// modality_adaptive_box_head.py:28-50): implicit GEMM over NHWC tokens, towers as groups, y = relu(conv + b) as bf16.  Rounds 1-5 cut K = 9 Cin of these layers into
// slices over workgroups (f32 slabs) and folded them with a second launch (slab_relu_kernel, 4.6 us); with the K quarters / halves on the wave groups of ONE
// workgroup the layer is one launch and needs no slabs.  For the layers whose tiles x K fit that shape (layers 1 and 2 of the towers: K = 2304 / 1152); the first
// layer (K = 6912: 108 K tiles) keeps its slices over 384 workgroups.
// ------------------------------------------------------------------------------------------------
template <int BN, int G, int NS>
__device__ __forceinline__ void conv_fin_body(const GemmParams& p, const int bx, char* smem_all) {
    constexpr int WPG = 8 / G;
    static_assert(WPG == 2 * (BN / 32) && (BN == 64 || BN == 32), "tile geometry");
    constexpr int STAGE = (64 + BN) * 128, LPT = (64 + BN) / 8 / WPG, LPT_A = 8 / WPG, RS = 32 * 4 + 16;
    constexpr int ITS = 32 / G / 8;
    static_assert(G * NS * STAGE <= 160 * 1024 && 8 * 32 * RS <= G * NS * STAGE, "LDS");
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave_all = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int grp = wave_all / WPG, wave = wave_all % WPG;
    const int wm = wave / (BN / 32), wn = wave % (BN / 32);
    char* smem = smem_all + grp * (NS * STAGE);
    const int MT = (p.M + 63) >> 6, NT = p.N / BN, TG = MT * NT;
    if (bx >= TG * p.groups) return;
    const int g = (int)fd_div((uint32_t)bx, p.fd_gsz);                 // tower (fd_gsz = tiles per tower)
    const int L = bx - g * TG;
    const int nt = (int)fd_div((uint32_t)L, p.fd_mt), mt = L - nt * MT;
    const int m0 = mt * 64, n0 = nt * BN;
    const int kspan = p.K / G, kbase = grp * kspan, nk = kspan >> 6;
    const int convF = p.conv_F, cin_g = p.cin_g, lda = p.lda, S = convF * convF;
    const int goff = g == 0 ? p.a_goff[0] : g == 1 ? p.a_goff[1] : g == 2 ? p.a_goff[2] : p.a_goff[3];

    const int c16 = lane & 7;
    const int col = n0 + wn * 32 + c16 * 4;
    const f32x4 bias4 = *reinterpret_cast<const f32x4*>(p.bias ? p.bias + (size_t)g * p.N + col : reinterpret_cast<const float*>(g_zero_page));

    // DMA plan: instruction i of wave w fills stage rows [8 (w + WPG i), + 8): the A rows (pixels of the tile, gathered per 3x3 tap; out-of-image taps read the zero page)
    // first, then the W rows of this tower
    const bf16_t* src[LPT];
    int a_i[LPT_A], a_j[LPT_A];
#pragma unroll
    for (int i = 0; i < LPT; ++i) {
        const int r = 8 * (wave + WPG * i) + (lane >> 3);
        const int chunk = (lane & 7) ^ ((r >> 1) & 7);
        if (i < LPT_A) {
            int gm = m0 + r;
            gm = gm < p.M ? gm : p.M - 1;
            const int b = gm / S, pix = gm - b * S;
            a_i[i] = pix / convF;
            a_j[i] = pix - a_i[i] * convF;
            src[i] = p.A + (size_t)b * S * lda + goff + chunk * 8;
        } else {
            src[i] = p.W + ((size_t)g * p.N + n0 + r - 64) * p.ldw + kbase + chunk * 8;
        }
    }
    const bf16_t* const zero_page = reinterpret_cast<const bf16_t*>(g_zero_page);
    auto issue = [&](int kt) __attribute__((always_inline)) {
        char* st = smem + (kt % NS) * STAGE;
        const int k0 = kbase + kt * 64;                  // K index = tap * cin_g + channel; a 64-wide tile never straddles taps (cin_g % 64 == 0)
        const int tap = k0 / cin_g, c0 = k0 - tap * cin_g;
        const int tap_di = tap / 3 - 1, tap_dj = tap % 3 - 1;
#pragma unroll
        for (int i = 0; i < LPT; ++i) {
            const bf16_t* gp;
            if (i < LPT_A) {
                const int ii = a_i[i] + tap_di, jj = a_j[i] + tap_dj;
                const bool ok = (unsigned)ii < (unsigned)convF && (unsigned)jj < (unsigned)convF;
                gp = ok ? src[i] + (size_t)(ii * convF + jj) * lda + c0 : zero_page;
            } else {
                gp = src[i] + kt * 64;
            }
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)gp, (__attribute__((address_space(3))) void*)(st + (wave + WPG * i) * 1024), 16, 0, 0);
        }
    };
    f32x16 acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;
#pragma unroll
    for (int t = 0; t < NS - 1; ++t)
        if (t < nk) issue(t);
    for (int kt = 0; kt < nk; ++kt) {
        const int ahead = nk - 1 - kt;
        if (ahead >= NS - 2) fin_wait_vmcnt<LPT * (NS - 2)>();
        else if (NS > 3 && ahead == 1) fin_wait_vmcnt<LPT>();
        else fin_wait_vmcnt<0>();
        __builtin_amdgcn_s_barrier();
        const char* sA = smem + (kt % NS) * STAGE;
        const char* sB = sA + 64 * 128;
        bf16x8 af[4], bfr[4];
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {
            const int chunk = ks * 2 + (lane >> 5);
            af[ks] = *reinterpret_cast<const bf16x8*>(sA + swz128(wm * 32 + (lane & 31), chunk));
            bfr[ks] = *reinterpret_cast<const bf16x8*>(sB + swz128(wn * 32 + (lane & 31), chunk));
        }
        __builtin_amdgcn_sched_barrier(0);
        if (kt + NS - 1 < nk) issue(kt + NS - 1);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(bfr[ks], af[ks], acc, 0, 0, 0);
    }
    __builtin_amdgcn_s_barrier();
    {
        char* cw = smem_all + (grp * WPG + wave) * (32 * RS);
        const int rl = lane & 31;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const f32x4 v = {acc[4 * q], acc[4 * q + 1], acc[4 * q + 2], acc[4 * q + 3]};
            *reinterpret_cast<f32x4*>(cw + rl * RS + (8 * q + 4 * (lane >> 5)) * 4) = v;
        }
    }
    __syncthreads();
#pragma unroll
    for (int it = 0; it < ITS; ++it) {
        const int r = grp * (32 / G) + it * 8 + (lane >> 3);
        const int row = m0 + wm * 32 + r;
        f32x4 v = *reinterpret_cast<const f32x4*>(smem_all + wave * (32 * RS) + r * RS + c16 * 16);
#pragma unroll
        for (int gg = 1; gg < G; ++gg) v += *reinterpret_cast<const f32x4*>(smem_all + (gg * WPG + wave) * (32 * RS) + r * RS + c16 * 16);
        v += bias4;
#pragma unroll
        for (int e = 0; e < 4; ++e) v[e] = fmaxf(v[e], 0.f);
        if (row < p.M)
            *reinterpret_cast<uint2*>(reinterpret_cast<bf16_t*>(p.C) + (size_t)row * p.ldc + (size_t)g * p.N + col) = uint2{pack_bf16x2(v[0], v[1]), pack_bf16x2(v[2], v[3])};
    }
}

template <int BN>
__global__ __launch_bounds__(512) void conv_fin_kernel(const GemmParams p) {
    kernarg_warm<sizeof(GemmParams) + 64>();
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const uint32_t pfs = prefetch_issue<512>(p.pf, p.pf_bytes, blockIdx.x, gridDim.x);
    if constexpr (BN == 32) conv_fin_body<32, 4, 3>(p, blockIdx.x, smem);
    else conv_fin_body<64, 2, 4>(p, blockIdx.x, smem);
    prefetch_retire(pfs);
}

// 0 = this layer does not fit the shape; 1 = 64 x 64 tiles on two wave groups; 2 = 64 x 32 on four
int conv_fin_form(const GemmParams& p) {
    if (p.conv_F <= 0 || p.epi != EPI_BF16 || p.act != 2 || p.cin_g % 64 != 0 || p.K != 9 * p.cin_g || p.M <= 0 || p.groups < 1 || p.groups > 4 || !p.C) return 0;
    if (tune_get(p.tune, &uvl_tuning::fin_w, -1) == 2) return 0;                 // uvl_tune_set("fin_w", 2): the slab form everywhere (A/B)
    const int nk = p.K / 64, MT = (p.M + 63) / 64;
    if (nk > 48 && tune_get(p.tune, &uvl_tuning::fin_w, -1) != 3) return 0;      // a long K (the first layer) is better cut over many workgroups (fin_w 3: A/B of that)
    if (p.N % 32 == 0 && nk % 4 == 0 && (long)MT * (p.N / 32) * p.groups <= 256) return 2;
    if (p.N % 64 == 0 && nk % 2 == 0 && (long)MT * (p.N / 64) * p.groups <= 256) return 1;
    return 0;
}
hipError_t launch_conv_fin(const GemmParams& p_in, hipStream_t s) {
    const int form = conv_fin_form(p_in);
    if (!form) return hipErrorInvalidValue;
    GemmParams p = p_in;
    const int MT = (p.M + 63) / 64, NT = p.N / (form == 2 ? 32 : 64);
    p.fd_mt = fastdiv_of((uint32_t)MT);
    p.fd_gsz = fastdiv_of((uint32_t)(MT * NT));
    const int blocks = MT * NT * p.groups;
    if (form == 2) {
        constexpr size_t lds = 4 * 3 * 96 * 128;
        static bool attr_done = false;
        if (!attr_done) {
            hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(conv_fin_kernel<32>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
            if (e != hipSuccess) return e;
            attr_done = true;
        }
        g_last_kernel = "conv_fin_kernel<32>";
        hipLaunchKernelGGL(conv_fin_kernel<32>, dim3(blocks), dim3(512), lds, s, p);
    } else {
        constexpr size_t lds = 2 * 4 * 128 * 128;
        static bool attr_done = false;
        if (!attr_done) {
            hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(conv_fin_kernel<64>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
            if (e != hipSuccess) return e;
            attr_done = true;
        }
        g_last_kernel = "conv_fin_kernel<64>";
        hipLaunchKernelGGL(conv_fin_kernel<64>, dim3(blocks), dim3(512), lds, s, p);
    }
    return hipGetLastError();
}

bool gemm_fin_ok(const GemmParams& p) {
    return p.M > 0 && p.epi == EPI_F32 && p.N > 0 && p.N % 64 == 0 && p.K >= 128 && p.K % 128 == 0 && p.conv_F == 0 && p.groups <= 1 && p.splitk <= 1 && p.C != nullptr &&
           (!p.res_st || (p.res_g && p.res_b && p.N <= 1024 && p.accumulate)) && (!(p.xn || p.st_out) || p.N % 32 == 0) && (!(p.st_out || p.res_st) || p.st_rows > 0);
}

// Tile width: 64 x 32 tiles on four wave groups while they are at most one workgroup per CU (uvl_tuning.fin_w: 0 = always 64 x 64, 1 = always 64 x 32)
static bool fin_w32(const GemmParams& p) {
    const int forced = tune_get(p.tune, &uvl_tuning::fin_w, -1);
    if (forced == 0 || forced == 1) return forced != 0 && (p.K / 64) % 4 == 0;
    return (p.K / 64) % 4 == 0 && (long)((p.M + 63) / 64) * (p.N / 32) <= 256;
}
template <bool W32>
static hipError_t launch_fin1(const GemmParams& a, int ba, hipStream_t s) {
    constexpr size_t lds = W32 ? 4 * 3 * 96 * 128 : 2 * 4 * 128 * 128;
    auto kern = gemm_fin_kernel<W32 ? 32 : 64>;
    static bool attr_done = false;
    if (!attr_done) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) return e;
        attr_done = true;
    }
    g_last_kernel = W32 ? "gemm_fin_kernel<32>" : "gemm_fin_kernel<64>";
    hipLaunchKernelGGL(kern, dim3(ba), dim3(512), lds, s, a);
    return hipGetLastError();
}
template <bool W32A, bool W32B>
static hipError_t launch_fin2(const GemmParams& a, const GemmParams& b, int ba, int bb, hipStream_t s) {
    constexpr size_t lds = (W32A || W32B) ? 4 * 3 * 96 * 128 : 2 * 4 * 128 * 128;
    auto kern = gemm_fin_pair_kernel<W32A ? 32 : 64, W32B ? 32 : 64>;
    static bool attr_done = false;
    if (!attr_done) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) return e;
        attr_done = true;
    }
    g_last_kernel = W32A ? (W32B ? "gemm_fin_pair_kernel<32,32>" : "gemm_fin_pair_kernel<32,64>") : (W32B ? "gemm_fin_pair_kernel<64,32>" : "gemm_fin_pair_kernel<64,64>");
    hipLaunchKernelGGL(kern, dim3(ba + bb), dim3(512), lds, s, a, b, bb);
    return hipGetLastError();
}

hipError_t launch_gemm_fin(const GemmParams& a_in, const GemmParams* b_in, hipStream_t s) {
    if (!gemm_fin_ok(a_in) || (b_in && !gemm_fin_ok(*b_in))) return hipErrorInvalidValue;
    auto prep = [](GemmParams& p, bool w32) {
        const int MT = (p.M + 63) / 64, NT = p.N / (w32 ? 32 : 64);
        p.fd_mt = fastdiv_of((uint32_t)MT);
        p.fd_rpb = fastdiv_of((uint32_t)p.rpb);
        return 8 * ((MT * NT + 7) / 8);
    };
    GemmParams a = a_in;
    const bool wa = fin_w32(a);
    const int ba = prep(a, wa);
    if (!b_in) return wa ? launch_fin1<true>(a, ba, s) : launch_fin1<false>(a, ba, s);
    GemmParams b = *b_in;
    const bool wb = fin_w32(b);
    const int bb = prep(b, wb);
    if (wa) return wb ? launch_fin2<true, true>(a, b, ba, bb, s) : launch_fin2<true, false>(a, b, ba, bb, s);
    return wb ? launch_fin2<false, true>(a, b, ba, bb, s) : launch_fin2<false, false>(a, b, ba, bb, s);
}

}  // namespace uvl
