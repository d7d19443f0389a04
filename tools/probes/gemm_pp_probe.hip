// Probe: "ping-pong" GEMM main loop -- one 8-wave workgroup per CU, two groups of four waves (one wave of each group per SIMD)
// a barrier apart: while one group runs the 16 MFMAs of a K tile, the other reads the fragments of its next K tile from LDS
// (whole K tile into registers) and issues DMA; then they swap.  Tile 128 x 256 (group g owns columns [128 g, +128)), A rows shared.
// Beside it: the product's structure (128 x 128, 4 waves, 2 workgroups per CU, one barrier per K tile) in the same binary.
// Build: hipcc --offload-arch=gfx950 -O3 -mllvm -amdgpu-mfma-vgpr-form=1 -I uvltrack_amd/csrc tools/probes/gemm_pp_probe.hip -o tools/probes/gemm_pp_probe
#include "common.h"
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
template <int N_> __device__ __forceinline__ void wait_vm() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N_) : "memory"); }
__device__ __forceinline__ void wait_lgkm0() { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); }

__device__ __forceinline__ void tile_of(int bid, int MT, int NT, int group_m, int& mt, int& nt, bool& live) {
    const int xcd = bid & 7, idx = bid >> 3;
    const int T = MT * NT, base = T >> 3, rem = T & 7;
    const int cnt = base + (xcd < rem ? 1 : 0);
    live = idx < cnt;
    const int L = xcd * base + (xcd < rem ? xcd : rem) + idx;
    const int gsz = group_m * NT, gi = L / gsz, within = L - gi * gsz;
    const int gm = min(group_m, MT - gi * group_m);
    nt = within / gm;
    mt = gi * group_m + (within - nt * gm);
}

// C rows of a wave's 64 x 64 sub-tile through LDS, 16 bytes per lane, whole 128-byte lines (the product's store)
template <int NW>
__device__ __forceinline__ void store_tile(f32x16 (&acc)[2][2], char* smem, bf16_t* C, int M, int N, int row0, int col0, int lane, int wave) {
    constexpr int RS = 64 * 2 + 16, LPR = 8, RPI = 8;
    char* cw = smem + wave * (32 * RS);
#pragma unroll
    for (int i = 0; i < 2; ++i) {
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                uint2 o = {pack_bf16x2(acc[i][j][4 * q], acc[i][j][4 * q + 1]), pack_bf16x2(acc[i][j][4 * q + 2], acc[i][j][4 * q + 3])};
                *reinterpret_cast<uint2*>(cw + (lane & 31) * RS + (j * 32 + 8 * q + 4 * (lane >> 5)) * 2) = o;
            }
#pragma unroll
        for (int it = 0; it < 32 / RPI; ++it) {
            const int r = it * RPI + lane / LPR, c16 = lane % LPR;
            const u32x4 v = *reinterpret_cast<const u32x4*>(cw + r * RS + c16 * 16);
            const int row = row0 + i * 32 + r;
            if (row < M) *reinterpret_cast<u32x4*>(C + (size_t)row * N + col0 + c16 * 8) = v;
        }
    }
}

// ---- baseline: the product's main loop (128 x 128, 2 x 2 waves, NS = 2) ----
__global__ __launch_bounds__(256) void base_kernel(const bf16_t* __restrict__ A, const bf16_t* __restrict__ W, bf16_t* __restrict__ C, int M, int N, int K, int group_m) {
    constexpr int BM = 128, BN = 128, BK = 64, NW = 4, NS = 2, ROWS = BM + BN, STAGE = ROWS * 128, LPT = ROWS / (8 * NW);
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 1, wn = wave & 1;
    int mt, nt; bool live;
    tile_of(blockIdx.x, (M + BM - 1) / BM, N / BN, group_m, mt, nt, live);
    if (!live) return;
    const int m0 = mt * BM, n0 = nt * BN;
    const bf16_t* src[LPT];
#pragma unroll
    for (int i = 0; i < LPT; ++i) {
        const int r = 8 * (wave + NW * i) + (lane >> 3);
        const int chunk = (lane & 7) ^ ((r >> 1) & 7);
        if (r < BM) { int gmr = m0 + r; gmr = gmr < M ? gmr : M - 1; src[i] = A + (size_t)gmr * K + chunk * 8; }
        else src[i] = W + (size_t)(n0 + r - BM) * K + chunk * 8;
    }
    auto issue = [&](int kt) __attribute__((always_inline)) {
        char* st = smem + (kt % NS) * STAGE;
#pragma unroll
        for (int i = 0; i < LPT; ++i)
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(src[i] + kt * BK),
                                             (__attribute__((address_space(3))) void*)(st + (wave + NW * i) * 1024), 16, 0, 0);
    };
    f32x16 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
    const int nk = K / BK;
    issue(0);
    for (int kt = 0; kt < nk; ++kt) {
        wait_vm<0>();
        __builtin_amdgcn_s_barrier();
        if (kt + 1 < nk) issue(kt + 1);
        const char* sA = smem + (kt % NS) * STAGE;
        const char* sB = sA + BM * 128;
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {
            bf16x8 af[2], bfr[2];
            const int chunk = ks * 2 + (lane >> 5);
#pragma unroll
            for (int i = 0; i < 2; ++i) af[i] = *reinterpret_cast<const bf16x8*>(sA + swz128(wm * 64 + i * 32 + (lane & 31), chunk));
#pragma unroll
            for (int j = 0; j < 2; ++j) bfr[j] = *reinterpret_cast<const bf16x8*>(sB + swz128(wn * 64 + j * 32 + (lane & 31), chunk));
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(bfr[j], af[i], acc[i][j], 0, 0, 0);
        }
    }
    __builtin_amdgcn_s_barrier();
    store_tile<4>(acc, smem, C, M, N, m0 + wm * 64, n0 + wn * 64, lane, wave);
}

// ---- ping-pong: 128 x 256, 8 waves = 2 groups x (2 x 2) waves, NS = 3 stages of 48 KB ----
// MODE: 0 full | 1 no MFMA (reads + DMA + barriers only) | 2 no LDS reads (fragments stay what the prologue read)
template <int MODE>
__global__ __launch_bounds__(512) void pp_kernel(const bf16_t* __restrict__ A, const bf16_t* __restrict__ W, bf16_t* __restrict__ C, int M, int N, int K, int group_m) {
    constexpr int BM = 128, BN = 256, BK = 64, NW = 8, NS = 3, ROWS = BM + BN, STAGE = ROWS * 128, LPT = ROWS / (8 * NW);   // 6 DMAs per wave per K tile
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int g = wave >> 2, wm = (wave >> 1) & 1, wn = wave & 1;
    int mt, nt; bool live;
    tile_of(blockIdx.x, (M + BM - 1) / BM, N / BN, group_m, mt, nt, live);
    if (!live) return;
    const int m0 = mt * BM, n0 = nt * BN;
    const bf16_t* src[LPT];
#pragma unroll
    for (int i = 0; i < LPT; ++i) {
        const int r = 8 * (wave + NW * i) + (lane >> 3);
        const int chunk = (lane & 7) ^ ((r >> 1) & 7);
        if (r < BM) { int gmr = m0 + r; gmr = gmr < M ? gmr : M - 1; src[i] = A + (size_t)gmr * K + chunk * 8; }
        else src[i] = W + (size_t)(n0 + r - BM) * K + chunk * 8;
    }
    auto issue = [&](int kt) __attribute__((always_inline)) {
        char* st = smem + (kt % NS) * STAGE;
#pragma unroll
        for (int i = 0; i < LPT; ++i)
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(src[i] + kt * BK),
                                             (__attribute__((address_space(3))) void*)(st + (wave + NW * i) * 1024), 16, 0, 0);
    };
    f32x16 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
    bf16x8 af[4][2], bfr[4][2];                    // the fragments of one whole K tile: 64 registers
    auto read = [&](int kt) __attribute__((always_inline)) {
        const char* sA = smem + (kt % NS) * STAGE;
        const char* sB = sA + (BM + g * 128) * 128;
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {
            const int chunk = ks * 2 + (lane >> 5);
#pragma unroll
            for (int i = 0; i < 2; ++i) af[ks][i] = *reinterpret_cast<const bf16x8*>(sA + swz128(wm * 64 + i * 32 + (lane & 31), chunk));
#pragma unroll
            for (int j = 0; j < 2; ++j) bfr[ks][j] = *reinterpret_cast<const bf16x8*>(sB + swz128(wn * 64 + j * 32 + (lane & 31), chunk));
        }
        wait_lgkm0();                               // complete before the barrier that lets a DMA refill this stage
    };
    auto mfma = [&]() __attribute__((always_inline)) {
        if (MODE == 1) return;
        __builtin_amdgcn_s_setprio(1);
#pragma unroll
        for (int ks = 0; ks < 4; ++ks)
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(bfr[ks][j], af[ks][i], acc[i][j], 0, 0, 0);
        __builtin_amdgcn_s_setprio(0);
    };
    const int nk = K / BK;
#pragma unroll
    for (int t = 0; t < NS; ++t) if (t < nk) issue(t);
    if (nk >= 3) wait_vm<2 * LPT>(); else if (nk == 2) wait_vm<LPT>(); else wait_vm<0>();
    __builtin_amdgcn_s_barrier();
    if (g == 0 || MODE == 2) read(0);
    __builtin_amdgcn_sched_barrier(0);
    for (int kt = 0; kt < nk; ++kt) {
        // ---- interval A: group 0 multiplies K tile kt, group 1 reads it ----
        if (g == 0) mfma(); else if (MODE != 2) read(kt);
        __builtin_amdgcn_sched_barrier(0);
        if (kt + 2 <= nk - 1) wait_vm<LPT>(); else wait_vm<0>();        // this wave's pieces of K tile kt+1 have landed
        __builtin_amdgcn_s_barrier();
        // ---- interval B: stage kt % 3 is free (both groups have read it): refill; group 0 reads K tile kt+1, group 1 multiplies kt ----
        if (kt + 3 < nk) issue(kt + 3);
        if (g == 0) { if (kt + 1 < nk && MODE != 2) read(kt + 1); } else mfma();
        __builtin_amdgcn_sched_barrier(0);
        __builtin_amdgcn_s_barrier();
    }
    if (MODE == 1) { if (acc[0][0][0] == 1234.5f) C[0] = f2bf(1.f); return; }
    store_tile<8>(acc, smem, C, M, N, m0 + wm * 64, n0 + g * 128 + wn * 64, lane, wave);
}

static float time_it(void (*launch)(void*), void* ctx) {
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    for (int i = 0; i < 3; ++i) launch(ctx);
    hipEventRecord(a);
    const int it = 20;
    for (int i = 0; i < it; ++i) launch(ctx);
    hipEventRecord(b); hipEventSynchronize(b);
    float ms; hipEventElapsedTime(&ms, a, b);
    return ms / it;
}
struct Ctx { const bf16_t* A; const bf16_t* W; bf16_t* C; int M, N, K, gm; };

template <int MODE> static void launch_pp(void* c) {
    auto* x = (Ctx*)c;
    const int MT = (x->M + 127) / 128, NT = x->N / 256;
    hipLaunchKernelGGL(pp_kernel<MODE>, dim3(8 * ((MT * NT + 7) / 8)), dim3(512), 3 * 384 * 128, 0, x->A, x->W, x->C, x->M, x->N, x->K, x->gm);
}
static void launch_base(void* c) {
    auto* x = (Ctx*)c;
    const int MT = (x->M + 127) / 128, NT = x->N / 128;
    hipLaunchKernelGGL(base_kernel, dim3(8 * ((MT * NT + 7) / 8)), dim3(256), 2 * 256 * 128, 0, x->A, x->W, x->C, x->M, x->N, x->K, x->gm);
}

int main() {
    const int Mmax = 21792, Nmax = 4096, Kmax = 4096;
    bf16_t *A, *W, *C;
    hipMalloc(&A, (size_t)Mmax * Kmax * 2); hipMalloc(&W, (size_t)Nmax * Kmax * 2); hipMalloc(&C, (size_t)Mmax * Nmax * 2);
    {
        const size_t na = (size_t)Mmax * Kmax, nw = (size_t)Nmax * Kmax;
        std::vector<uint16_t> h(na > nw ? na : nw);
        uint64_t st = 0x9E3779B97F4A7C15ull;
        auto fill = [&](size_t n) {
            for (size_t i = 0; i < n; ++i) {
                st = st * 6364136223846793005ull + 1442695040888963407ull;
                const float f = (float)((st >> 40) & 0xFFFF) / 32768.0f - 1.0f;
                uint32_t u; memcpy(&u, &f, 4);
                h[i] = (uint16_t)(u >> 16);
            }
        };
        fill(na); hipMemcpy(A, h.data(), na * 2, hipMemcpyHostToDevice);
        fill(nw); hipMemcpy(W, h.data(), nw * 2, hipMemcpyHostToDevice);
    }
    hipFuncSetAttribute((const void*)pp_kernel<0>, hipFuncAttributeMaxDynamicSharedMemorySize, 3 * 384 * 128);
    hipFuncSetAttribute((const void*)pp_kernel<1>, hipFuncAttributeMaxDynamicSharedMemorySize, 3 * 384 * 128);
    hipFuncSetAttribute((const void*)pp_kernel<2>, hipFuncAttributeMaxDynamicSharedMemorySize, 3 * 384 * 128);
    hipFuncSetAttribute((const void*)base_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 2 * 256 * 128);
    // correctness first: both kernels against a host reference on samples
    for (int which = 0; which < 2; ++which) {
        const int M = 4424, N = 3072, K = 768;
        hipMemset(C, 0, (size_t)M * N * 2);
        Ctx c{A, W, C, M, N, K, 8};
        if (which) launch_pp<0>(&c); else launch_base(&c);
        hipDeviceSynchronize();
        std::vector<uint16_t> h2((size_t)M * N), hA((size_t)M * K), hW((size_t)N * K);
        hipMemcpy(h2.data(), C, h2.size() * 2, hipMemcpyDeviceToHost);
        hipMemcpy(hA.data(), A, hA.size() * 2, hipMemcpyDeviceToHost); hipMemcpy(hW.data(), W, hW.size() * 2, hipMemcpyDeviceToHost);
        auto f = [](uint16_t h) { uint32_t u = (uint32_t)h << 16; float x; memcpy(&x, &u, 4); return x; };
        double maxerr = 0; int bad = 0;
        for (int t = 0; t < 20000; ++t) {
            const int m = (int)(((long)t * 7919) % M), n = (int)(((long)t * 104729) % N);
            double acc = 0;
            for (int k2 = 0; k2 < K; ++k2) acc += (double)f(hA[(size_t)m * K + k2]) * f(hW[(size_t)n * K + k2]);
            const double err = fabs((double)f(h2[(size_t)m * N + n]) - acc);
            if (err > 0.02 * fabs(acc) + 0.05) ++bad;
            if (err > maxerr) maxerr = err;
        }
        printf("%s check: max abs err %.4f over 20000 samples, %d outside bf16 tolerance\n", which ? "ping-pong" : "baseline", maxerr, bad);
    }
    const int shapes[][3] = {{17696, 3072, 768}, {17696, 2304, 768}, {17696, 768, 768}, {17696, 768, 3072}, {5448, 4096, 1024}, {5448, 3072, 1024}, {5448, 1024, 4096},
                             {7304, 4096, 1024}, {7304, 1024, 1024}, {21792, 4096, 1024}, {16384, 4096, 4096}};
    for (auto& s : shapes) {
        Ctx c{A, W, C, s[0], s[1], s[2], 8};
        const double fl = 2.0 * s[0] * s[1] * s[2];
        float tb = 1e9f, tp = 1e9f, t1 = 1e9f, t2 = 1e9f;
        for (int rep = 0; rep < 2; ++rep) {
            tb = fminf(tb, time_it(launch_base, &c));
            tp = fminf(tp, time_it(launch_pp<0>, &c));
            t1 = fminf(t1, time_it(launch_pp<1>, &c));
            t2 = fminf(t2, time_it(launch_pp<2>, &c));
        }
        printf("M=%5d N=%4d K=%4d  product loop %7.1f us %6.1f TF | ping-pong %7.1f us %6.1f TF | no MFMA %7.1f us | no LDS reads %7.1f us (%6.1f TF-eq)\n", s[0], s[1], s[2],
               tb * 1e3, fl / (tb * 1e-3) / 1e12, tp * 1e3, fl / (tp * 1e-3) / 1e12, t1 * 1e3, t2 * 1e3, fl / (t2 * 1e-3) / 1e12);
    }
    return 0;
}
