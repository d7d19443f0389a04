// Probe: what does an in-launch split-K combine cost on the one-sequence fc2 shape (x[553,768] += a[553,3072] W[768,3072]^T, 64 x 64
// tiles, 4 K-slices)?  DESIGN.md section 9, step 1: the LayerNorm-free frame needs the residual GEMMs to leave a COMPLETE x.
//   variant 0  slices write f32 slabs (the product today; the next LayerNorm folds them)
//   variant 1  slabs + per-tile arrival counter; the last-arriving slice adds the slabs in slice order to x (agent-scope release /
//              acquire as cdna_hip_programming.md prescribes; the counter is reset by the last arriver, zero-initialised once)
//   variant 2  as 1 with the slices of a tile dispatched to one XCD
//   variant 3  as 2 with write-through (sc1) slab stores instead of the release fence
//   variant 4  as 3 with sc1 slab loads instead of the acquire fence
// Build: hipcc --offload-arch=gfx950 -O3 -mllvm -amdgpu-mfma-vgpr-form=1 -I uvltrack_amd/csrc tools/probes/splitk_combine_probe.hip -o tools/probes/splitk_combine_probe
#include "common.h"
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
template <int N_> __device__ __forceinline__ void wait_vm() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N_) : "memory"); }

template <int VAR>
__global__ __launch_bounds__(256) void sk_kernel(const bf16_t* __restrict__ A, const bf16_t* __restrict__ W, float* __restrict__ slabs, float* __restrict__ X,
                                                 unsigned* __restrict__ cnt, int M, int N, int K, int S) {
    constexpr int BM = 64, BN = 64, BK = 64, NW = 4, NS = 3, ROWS = 128, STAGE = ROWS * 128, LPT = 4;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 1, wn = wave & 1;
    const int MT = (M + BM - 1) / BM, NT = N / BN, T = MT * NT;
    int tile, sk;
    if (VAR >= 2) {            // the S slices of a tile sit next to each other in one XCD's run
        const int xcd = blockIdx.x & 7, idx = blockIdx.x >> 3;
        const int base = T >> 3, rem = T & 7, cntx = base + (xcd < rem ? 1 : 0);
        if (idx >= cntx * S) return;
        tile = xcd * base + (xcd < rem ? xcd : rem) + idx / S;
        sk = idx % S;
    } else {                   // the product's order: slice-major (grid.y = slice)
        tile = blockIdx.x % T; sk = blockIdx.x / T;
        if (sk >= S) return;
    }
    const int nt = tile / MT, mt = tile % MT;      // N-major runs: the M tiles of a weight panel adjacent
    const int m0 = mt * BM, n0 = nt * BN;
    const int kspan = K / S, kbase = sk * kspan, nk = kspan / BK;
    const bf16_t* src[LPT];
#pragma unroll
    for (int i = 0; i < LPT; ++i) {
        const int r = 8 * (wave + NW * i) + (lane >> 3);
        const int chunk = (lane & 7) ^ ((r >> 1) & 7);
        if (r < BM) { int gmr = m0 + r; gmr = gmr < M ? gmr : M - 1; src[i] = A + (size_t)gmr * K + kbase + chunk * 8; }
        else src[i] = W + (size_t)(n0 + r - BM) * K + kbase + chunk * 8;
    }
    auto issue = [&](int kt) __attribute__((always_inline)) {
        char* st = smem + (kt % NS) * STAGE;
#pragma unroll
        for (int i = 0; i < LPT; ++i)
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(src[i] + kt * BK),
                                             (__attribute__((address_space(3))) void*)(st + (wave + NW * i) * 1024), 16, 0, 0);
    };
    f32x16 acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;
    issue(0);
    if (nk > 1) issue(1);
    for (int kt = 0; kt < nk; ++kt) {
        if (nk - 1 - kt >= 1) wait_vm<LPT>(); else wait_vm<0>();
        __builtin_amdgcn_s_barrier();
        if (kt + 2 < nk) issue(kt + 2);
        const char* sA = smem + (kt % NS) * STAGE;
        const char* sB = sA + BM * 128;
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {
            const int chunk = ks * 2 + (lane >> 5);
            const bf16x8 af = *reinterpret_cast<const bf16x8*>(sA + swz128(wm * 32 + (lane & 31), chunk));
            const bf16x8 bfr = *reinterpret_cast<const bf16x8*>(sB + swz128(wn * 32 + (lane & 31), chunk));
            acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(bfr, af, acc, 0, 0, 0);
        }
    }
    // lane holds row m0 + wm*32 + (lane & 31), columns n0 + wn*32 + 8 q + 4 (lane >> 5) + 0..3
    const int row = m0 + wm * 32 + (lane & 31);
    float* slab = slabs + (size_t)sk * M * N;
    if (row < M) {
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int col = n0 + wn * 32 + 8 * q + 4 * (lane >> 5);
            const f32x4 pv = {acc[4 * q], acc[4 * q + 1], acc[4 * q + 2], acc[4 * q + 3]};
            float* dp = slab + (size_t)row * N + col;
            if (VAR >= 3) asm volatile("global_store_dwordx4 %0, %1, off sc1\n s_nop 1" ::"v"(dp), "v"(pv) : "memory");   // write-through: no release fence needed
            else *reinterpret_cast<f32x4*>(dp) = pv;
        }
    }
    if (VAR == 0) return;
    // ---- publish the slab, draw a ticket; the last slice of the tile combines ----
    volatile unsigned* s_lastp = reinterpret_cast<volatile unsigned*>(smem);     // the ring is idle; ONE shared object only (a second one de-pipelines the DMA loop)
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (tid == 0) {
        if (VAR < 3) {
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        }
        const unsigned old = __hip_atomic_fetch_add(cnt + tile, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        const bool last = (old == (unsigned)(S - 1));
        *s_lastp = last ? 1u : 0u;
        if (last) {
            __hip_atomic_store(cnt + tile, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);     // ready for the next launch
            if (VAR != 4) __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
        }
    }
    __syncthreads();
    if (!*s_lastp) return;
    if (row < M) {
        f32x4 part[4][4], xv[4];                      // every load first (one round trip), then the adds in slice order
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int col = n0 + wn * 32 + 8 * q + 4 * (lane >> 5);
            xv[q] = *reinterpret_cast<const f32x4*>(X + (size_t)row * N + col);
#pragma unroll
            for (int s = 0; s < 4; ++s) {
                const float* sp = slabs + ((size_t)(s < S ? s : 0) * M + row) * N + col;
                if (VAR == 4) asm volatile("global_load_dwordx4 %0, %1, off sc1" : "=v"(part[s][q]) : "v"(sp) : "memory");
                else part[s][q] = *reinterpret_cast<const f32x4*>(sp);
            }
        }
        if (VAR == 4) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        {
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int col = n0 + wn * 32 + 8 * q + 4 * (lane >> 5);
            f32x4 v = xv[q];
#pragma unroll
            for (int s = 0; s < 4; ++s) if (s < S) v += part[s][q];
            *reinterpret_cast<f32x4*>(X + (size_t)row * N + col) = v;
        }
        }
    }
}

// the consumer of variant 0: x += sum of slabs (what the next LayerNorm does on its way), one wave per row
__global__ __launch_bounds__(256) void fold_kernel(const float* __restrict__ slabs, float* __restrict__ X, int M, int N, int S) {
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (row >= M) return;
    for (int c = lane * 4; c < N; c += 256) {
        f32x4 v = *reinterpret_cast<const f32x4*>(X + (size_t)row * N + c);
        for (int s = 0; s < S; ++s) v += *reinterpret_cast<const f32x4*>(slabs + ((size_t)s * M + row) * N + c);
        *reinterpret_cast<f32x4*>(X + (size_t)row * N + c) = v;
    }
}

int main() {
    const int M = 553, N = 768, K = 3072, S = 4;
    const int MT = (M + 63) / 64, NT = N / 64, T = MT * NT;
    bf16_t *A, *W; float *slabs, *X, *X0; unsigned* cnt;
    (void)hipMalloc(&A, (size_t)M * K * 2); (void)hipMalloc(&W, (size_t)N * K * 2);
    (void)hipMalloc(&slabs, (size_t)S * M * N * 4); (void)hipMalloc(&X, (size_t)M * N * 4); (void)hipMalloc(&X0, (size_t)M * N * 4);
    (void)hipMalloc(&cnt, T * 4); (void)hipMemset(cnt, 0, T * 4);
    std::vector<uint16_t> hA((size_t)M * K), hW((size_t)N * K); std::vector<float> hX((size_t)M * N);
    uint64_t st = 0x9E3779B97F4A7C15ull;
    auto rnd = [&]() { st = st * 6364136223846793005ull + 1442695040888963407ull; return (float)((st >> 40) & 0xFFFF) / 32768.0f - 1.0f; };
    auto bf = [](float f) { uint32_t u; memcpy(&u, &f, 4); return (uint16_t)(u >> 16); };
    auto fb = [](uint16_t h) { uint32_t u = (uint32_t)h << 16; float x; memcpy(&x, &u, 4); return x; };
    for (auto& v : hA) v = bf(rnd()); for (auto& v : hW) v = bf(rnd() * 0.05f); for (auto& v : hX) v = rnd();
    (void)hipMemcpy(A, hA.data(), hA.size() * 2, hipMemcpyHostToDevice); (void)hipMemcpy(W, hW.data(), hW.size() * 2, hipMemcpyHostToDevice);
    (void)hipMemcpy(X0, hX.data(), hX.size() * 4, hipMemcpyHostToDevice);
    const int lds = 3 * 128 * 128;
    (void)hipFuncSetAttribute((const void*)sk_kernel<0>, hipFuncAttributeMaxDynamicSharedMemorySize, lds);
    (void)hipFuncSetAttribute((const void*)sk_kernel<1>, hipFuncAttributeMaxDynamicSharedMemorySize, lds);
    (void)hipFuncSetAttribute((const void*)sk_kernel<2>, hipFuncAttributeMaxDynamicSharedMemorySize, lds);
    (void)hipFuncSetAttribute((const void*)sk_kernel<3>, hipFuncAttributeMaxDynamicSharedMemorySize, lds);
    (void)hipFuncSetAttribute((const void*)sk_kernel<4>, hipFuncAttributeMaxDynamicSharedMemorySize, lds);
    const int grid01 = T * S, grid2 = 8 * ((T + 7) / 8) * S;
    auto run = [&](int var) {
        if (var == 0) { hipLaunchKernelGGL(sk_kernel<0>, dim3(grid01), dim3(256), lds, 0, A, W, slabs, X, cnt, M, N, K, S);
                        hipLaunchKernelGGL(fold_kernel, dim3((M + 3) / 4), dim3(256), 0, 0, slabs, X, M, N, S); }
        else if (var == 1) hipLaunchKernelGGL(sk_kernel<1>, dim3(grid01), dim3(256), lds, 0, A, W, slabs, X, cnt, M, N, K, S);
        else if (var == 2) hipLaunchKernelGGL(sk_kernel<2>, dim3(grid2), dim3(256), lds, 0, A, W, slabs, X, cnt, M, N, K, S);
        else if (var == 3) hipLaunchKernelGGL(sk_kernel<3>, dim3(grid2), dim3(256), lds, 0, A, W, slabs, X, cnt, M, N, K, S);
        else hipLaunchKernelGGL(sk_kernel<4>, dim3(grid2), dim3(256), lds, 0, A, W, slabs, X, cnt, M, N, K, S);
    };
    // correctness: every variant against the host, 30 repeats of the in-launch forms (stale reads would show up as misses)
    std::vector<float> ref((size_t)M * N), got((size_t)M * N);
    for (int m = 0; m < M; m += 7) for (int n = 0; n < N; ++n) {
        double a = 0; for (int k = 0; k < K; ++k) a += (double)fb(hA[(size_t)m * K + k]) * fb(hW[(size_t)n * K + k]);
        ref[(size_t)m * N + n] = (float)(hX[(size_t)m * N + n] + a);
    }
    for (int var = 0; var < 5; ++var) {
        int bad = 0; double maxerr = 0; int mism = 0;
        std::vector<float> first;
        for (int rep = 0; rep < (var ? 30 : 1); ++rep) {
            (void)hipMemcpy(X, X0, (size_t)M * N * 4, hipMemcpyDeviceToDevice);
            (void)hipMemset(slabs, 0xFF, (size_t)S * M * N * 4);           // poison: a stale read is a NaN
            run(var); (void)hipDeviceSynchronize();
            (void)hipMemcpy(got.data(), X, got.size() * 4, hipMemcpyDeviceToHost);
            for (int m = 0; m < M; m += 7) for (int n = 0; n < N; ++n) {
                const double e = fabs((double)got[(size_t)m * N + n] - ref[(size_t)m * N + n]);
                if (!(e < 2e-3)) ++bad;
                if (e > maxerr) maxerr = e;
            }
            if (rep == 0) first = got; else for (size_t i = 0; i < got.size(); ++i) mism += memcmp(&got[i], &first[i], 4) != 0;
        }
        printf("variant %d: max err %.2e, %d outside 2e-3, %d values differ between repeats\n", var, maxerr, bad, mism);
    }
    // timing: chains of dependent launches on one stream, as in the frame
    hipEvent_t a, b; (void)hipEventCreate(&a); (void)hipEventCreate(&b);
    for (int round = 0; round < 2; ++round)
        for (int var = 0; var < 5; ++var) {
            for (int i = 0; i < 5; ++i) run(var);
            (void)hipEventRecord(a);
            const int it = 200;
            for (int i = 0; i < it; ++i) run(var);
            (void)hipEventRecord(b); (void)hipEventSynchronize(b);
            float ms; (void)hipEventElapsedTime(&ms, a, b);
            printf("variant %d: %.2f us per step (%s)\n", var, ms * 1e3 / it, var == 0 ? "split-K GEMM + separate fold launch" : var == 1 ? "in-launch combine" : var == 2 ? "in-launch combine, slices of a tile on one XCD" : var == 3 ? "sc1 slab stores, acquire fence" : "sc1 slab stores, sc1 slab loads");
        }
    // the GEMM of variant 0 alone (the fold rides on the LayerNorm launch in the product)
    {
        for (int i = 0; i < 5; ++i) hipLaunchKernelGGL(sk_kernel<0>, dim3(grid01), dim3(256), lds, 0, A, W, slabs, X, cnt, M, N, K, S);
        (void)hipEventRecord(a);
        for (int i = 0; i < 200; ++i) hipLaunchKernelGGL(sk_kernel<0>, dim3(grid01), dim3(256), lds, 0, A, W, slabs, X, cnt, M, N, K, S);
        (void)hipEventRecord(b); (void)hipEventSynchronize(b);
        float ms; (void)hipEventElapsedTime(&ms, a, b);
        printf("split-K GEMM alone: %.2f us per launch\n", ms * 1e3 / 200);
    }
    return 0;
}
