// Probe: does hiding the epilogue of output tile t inside the K loop of tile t+1 pay?  (profiles/r02_gemm_structure.md: with the MFMA
// pipe a third busy a tile's time is the SUM of its phases, and the GELU epilogue is ~20 % of a K = 768-1024 tile.)
// 128 x 128 tile, 4 waves, NS = 2, two workgroups per CU, fc1-like epilogue (bias-free GELU, bf16 store).  Variants:
//   0  one tile per workgroup, epilogue stores straight from registers (8 B per lane)
//   1  persistent tile walk (next tile's first K tile requested during the last K iteration), serial epilogue
//   2  persistent, the previous tile's accumulators are kept and its epilogue runs in eight slices behind the MFMAs of the first
//      eight K iterations of the next tile
// Build: hipcc --offload-arch=gfx950 -O3 -mllvm -amdgpu-mfma-vgpr-form=1 -I uvltrack_amd/csrc tools/probes/gemm_ov_probe.hip -o tools/probes/gemm_ov_probe
#include "common.h"
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
__device__ __forceinline__ void wait_vm0() { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }

struct Walk {
    int MT, NT, group_m, first, cnt;
    __device__ __forceinline__ void tile(int seq, int& m0, int& n0) const {
        const int L = first + seq;
        const int gsz = group_m * NT, gi = L / gsz, within = L - gi * gsz;
        const int gm = min(group_m, MT - gi * group_m);
        const int nt = within / gm, mt = gi * group_m + (within - nt * gm);
        m0 = mt * 128; n0 = nt * 128;
    }
};

// one slice of the epilogue of a wave's 64 x 64 sub-tile: two register quads (i, j, q0 / q0 + 1): GELU, round, 8-byte stores
template <int S, bool GELU>
__device__ __forceinline__ void epi_slice(const f32x16 (&acc)[2][2], bf16_t* C, int M, int N, int row0, int col0, int lane) {
    constexpr int i = S >> 2, j = (S >> 1) & 1, q0 = (S & 1) * 2;
    const int row = row0 + i * 32 + (lane & 31);
#pragma unroll
    for (int q = q0; q < q0 + 2; ++q) {
        f32x4 v = {acc[i][j][4 * q], acc[i][j][4 * q + 1], acc[i][j][4 * q + 2], acc[i][j][4 * q + 3]};
        if (GELU) {
            const f32x2 g0 = gelu_erf_fast2(f32x2{v[0], v[1]}), g1 = gelu_erf_fast2(f32x2{v[2], v[3]});
            v = f32x4{g0[0], g0[1], g1[0], g1[1]};
        }
        const uint2 o = {pack_bf16x2(v[0], v[1]), pack_bf16x2(v[2], v[3])};
        if (row < M) *reinterpret_cast<uint2*>(C + (size_t)row * N + col0 + j * 32 + 8 * q + 4 * (lane >> 5)) = o;
    }
}
template <bool GELU>
__device__ __forceinline__ void epi_all(const f32x16 (&acc)[2][2], bf16_t* C, int M, int N, int row0, int col0, int lane) {
    epi_slice<0, GELU>(acc, C, M, N, row0, col0, lane); epi_slice<1, GELU>(acc, C, M, N, row0, col0, lane);
    epi_slice<2, GELU>(acc, C, M, N, row0, col0, lane); epi_slice<3, GELU>(acc, C, M, N, row0, col0, lane);
    epi_slice<4, GELU>(acc, C, M, N, row0, col0, lane); epi_slice<5, GELU>(acc, C, M, N, row0, col0, lane);
    epi_slice<6, GELU>(acc, C, M, N, row0, col0, lane); epi_slice<7, GELU>(acc, C, M, N, row0, col0, lane);
}

template <int VAR, bool GELU>
__global__ __launch_bounds__(256) void ov_kernel(const bf16_t* __restrict__ A, const bf16_t* __restrict__ W, bf16_t* __restrict__ C, int M, int N, int K,
                                                 int group_m, int wgs_per_xcd) {
    constexpr int BM = 128, BN = 128, NW = 4, STAGE = 256 * 128, LPT = 8, LPT_A = 4;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 1, wn = wave & 1;
    Walk wk;
    wk.MT = (M + BM - 1) / BM; wk.NT = N / BN; wk.group_m = group_m;
    const int xcd = blockIdx.x & 7, jb = blockIdx.x >> 3;
    const int T = wk.MT * wk.NT, base = T >> 3, rem = T & 7;
    wk.cnt = base + (xcd < rem ? 1 : 0);
    wk.first = xcd * base + (xcd < rem ? xcd : rem);
    const int nk = K / 64;
    const int stride = VAR == 0 ? (1 << 30) : wgs_per_xcd;
    uint32_t loffA[LPT_A], loffW[LPT - LPT_A];
#pragma unroll
    for (int i = LPT_A; i < LPT; ++i) {
        const int r = 8 * (wave + NW * i) + (lane >> 3);
        const int chunk = (lane & 7) ^ ((r >> 1) & 7);
        loffW[i - LPT_A] = (uint32_t)(r - BM) * (uint32_t)K * 2u + chunk * 16u;
    }
    auto prep_A = [&](int m0) __attribute__((always_inline)) {
#pragma unroll
        for (int i = 0; i < LPT_A; ++i) {
            const int r = 8 * (wave + NW * i) + (lane >> 3);
            const int chunk = (lane & 7) ^ ((r >> 1) & 7);
            int gmr = m0 + r; gmr = gmr < M ? gmr : M - 1;
            loffA[i] = (uint32_t)(gmr - m0) * (uint32_t)K * 2u + chunk * 16u;
        }
    };
    auto issue = [&](int stage, const char* ab, const char* wb) __attribute__((always_inline)) {
        char* st = smem + stage * STAGE;
#pragma unroll
        for (int i = 0; i < LPT; ++i) {
            const char* gp = i < LPT_A ? ab + loffA[i] : wb + loffW[i - LPT_A];
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)gp,
                                             (__attribute__((address_space(3))) void*)(st + (wave + NW * i) * 1024), 16, 0, 0);
        }
    };
    int seq = jb;
    if (seq >= wk.cnt) return;
    int m0, n0;
    wk.tile(seq, m0, n0);
    prep_A(m0);
    const char* ab = reinterpret_cast<const char*>(A + (size_t)m0 * K);
    const char* wb = reinterpret_cast<const char*>(W + (size_t)n0 * K);
    int g = 0;
    issue(0, ab, wb);
    f32x16 prev[2][2];
    bool have_prev = false;
    int pm0 = 0, pn0 = 0;
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) prev[i][j][r] = 0.f;
    while (true) {
        f32x16 acc[2][2];
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
        const int nseq = seq + stride;
        const bool has_next = nseq < wk.cnt;
        int nm0 = 0, nn0 = 0;
        if (has_next) wk.tile(nseq, nm0, nn0);
        for (int kt = 0; kt < nk; ++kt, ++g) {
            wait_vm0();
            __builtin_amdgcn_s_barrier();
            if (kt + 1 < nk) {
                issue((g + 1) & 1, ab + (size_t)(kt + 1) * 128, wb + (size_t)(kt + 1) * 128);
            } else if (has_next) {
                prep_A(nm0);
                issue((g + 1) & 1, reinterpret_cast<const char*>(A + (size_t)nm0 * K), reinterpret_cast<const char*>(W + (size_t)nn0 * K));
            }
            const char* sA = smem + (g & 1) * STAGE;
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
            if (VAR == 2 && have_prev && kt < 8) {
                // one eighth of the previous tile's epilogue: independent of this iteration's MFMAs, which are still in the pipe
                const int r0 = pm0 + wm * 64, c0 = pn0 + wn * 64;
                switch (kt) {
                    case 0: epi_slice<0, GELU>(prev, C, M, N, r0, c0, lane); break;
                    case 1: epi_slice<1, GELU>(prev, C, M, N, r0, c0, lane); break;
                    case 2: epi_slice<2, GELU>(prev, C, M, N, r0, c0, lane); break;
                    case 3: epi_slice<3, GELU>(prev, C, M, N, r0, c0, lane); break;
                    case 4: epi_slice<4, GELU>(prev, C, M, N, r0, c0, lane); break;
                    case 5: epi_slice<5, GELU>(prev, C, M, N, r0, c0, lane); break;
                    case 6: epi_slice<6, GELU>(prev, C, M, N, r0, c0, lane); break;
                    default: epi_slice<7, GELU>(prev, C, M, N, r0, c0, lane); break;
                }
            }
        }
        if (VAR == 2) {
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j) prev[i][j] = acc[i][j];
            pm0 = m0; pn0 = n0; have_prev = true;
            if (!has_next) { epi_all<GELU>(prev, C, M, N, pm0 + wm * 64, pn0 + wn * 64, lane); break; }
        } else {
            epi_all<GELU>(acc, C, M, N, m0 + wm * 64, n0 + wn * 64, lane);
            if (!has_next) break;
        }
        seq = nseq; m0 = nm0; n0 = nn0;
        ab = reinterpret_cast<const char*>(A + (size_t)m0 * K);
        wb = reinterpret_cast<const char*>(W + (size_t)n0 * K);
    }
}

struct Ctx { const bf16_t* A; const bf16_t* W; bf16_t* C; int M, N, K; };
template <int VAR, bool GELU> static void launch(void* c) {
    auto* x = (Ctx*)c;
    const int MT = (x->M + 127) / 128, NT = x->N / 128;
    const int per_xcd = VAR == 0 ? (MT * NT + 7) / 8 : 64;            // 2 workgroups per CU x 32 CUs per XCD
    hipLaunchKernelGGL((ov_kernel<VAR, GELU>), dim3(8 * per_xcd), dim3(256), 2 * 256 * 128, 0, x->A, x->W, x->C, x->M, x->N, x->K, 8, per_xcd);
}
static float time_it(void (*fn)(void*), void* ctx) {
    hipEvent_t a, b; (void)hipEventCreate(&a); (void)hipEventCreate(&b);
    for (int i = 0; i < 3; ++i) fn(ctx);
    (void)hipEventRecord(a);
    const int it = 20;
    for (int i = 0; i < it; ++i) fn(ctx);
    (void)hipEventRecord(b); (void)hipEventSynchronize(b);
    float ms; (void)hipEventElapsedTime(&ms, a, b);
    return ms / it;
}

int main() {
    const int Mmax = 21792, Nmax = 4096, Kmax = 4096;
    bf16_t *A, *W, *C, *C2;
    (void)hipMalloc(&A, (size_t)Mmax * Kmax * 2); (void)hipMalloc(&W, (size_t)Nmax * Kmax * 2);
    (void)hipMalloc(&C, (size_t)Mmax * Nmax * 2); (void)hipMalloc(&C2, (size_t)Mmax * Nmax * 2);
    {
        const size_t na = (size_t)Mmax * Kmax, nw = (size_t)Nmax * Kmax;
        std::vector<uint16_t> h(na > nw ? na : nw);
        uint64_t st = 0x9E3779B97F4A7C15ull;
        auto fill = [&](size_t n, float scale) {
            for (size_t i = 0; i < n; ++i) {
                st = st * 6364136223846793005ull + 1442695040888963407ull;
                const float f = ((float)((st >> 40) & 0xFFFF) / 32768.0f - 1.0f) * scale;
                uint32_t u; memcpy(&u, &f, 4);
                h[i] = (uint16_t)(u >> 16);
            }
        };
        fill(na, 1.0f); (void)hipMemcpy(A, h.data(), na * 2, hipMemcpyHostToDevice);
        fill(nw, 0.06f); (void)hipMemcpy(W, h.data(), nw * 2, hipMemcpyHostToDevice);       // pre-activations of O(1): GELU in its working range
    }
#define ATTR(V, G) (void)hipFuncSetAttribute((const void*)ov_kernel<V, G>, hipFuncAttributeMaxDynamicSharedMemorySize, 2 * 256 * 128)
    ATTR(0, true); ATTR(1, true); ATTR(2, true); ATTR(0, false); ATTR(1, false); ATTR(2, false);
    {   // the three variants must agree bit for bit (same arithmetic, same order)
        Ctx c{A, W, C, 4424, 3072, 768}, c2{A, W, C2, 4424, 3072, 768};
        const size_t n = (size_t)4424 * 3072;
        std::vector<uint16_t> h0(n), h1(n);
        (void)hipMemset(C, 0, n * 2); launch<0, true>(&c); (void)hipDeviceSynchronize(); (void)hipMemcpy(h0.data(), C, n * 2, hipMemcpyDeviceToHost);
        for (int v = 1; v <= 2; ++v) {
            (void)hipMemset(C2, 0, n * 2);
            if (v == 1) launch<1, true>(&c2); else launch<2, true>(&c2);
            (void)hipDeviceSynchronize(); (void)hipMemcpy(h1.data(), C2, n * 2, hipMemcpyDeviceToHost);
            size_t bad = 0; for (size_t i = 0; i < n; ++i) bad += h0[i] != h1[i];
            printf("variant %d vs variant 0: %zu of %zu outputs differ\n", v, bad, n);
        }
    }
    const int shapes[][3] = {{17696, 3072, 768}, {17696, 2304, 768}, {5448, 4096, 1024}, {7304, 4096, 1024}, {7304, 3072, 1024}, {21792, 4096, 1024}};
    for (auto& s : shapes) {
        Ctx c{A, W, C, s[0], s[1], s[2]};
        const double fl = 2.0 * s[0] * s[1] * s[2];
        float t[6] = {1e9f, 1e9f, 1e9f, 1e9f, 1e9f, 1e9f};
        for (int rep = 0; rep < 2; ++rep) {
            t[0] = fminf(t[0], time_it(launch<0, true>, &c)); t[1] = fminf(t[1], time_it(launch<1, true>, &c)); t[2] = fminf(t[2], time_it(launch<2, true>, &c));
            t[3] = fminf(t[3], time_it(launch<0, false>, &c)); t[4] = fminf(t[4], time_it(launch<1, false>, &c)); t[5] = fminf(t[5], time_it(launch<2, false>, &c));
        }
        printf("M=%5d N=%4d K=%4d  GELU: one-shot %6.1f us %5.0f TF | persistent %6.1f us %5.0f TF | overlapped %6.1f us %5.0f TF || plain store: %6.1f / %6.1f / %6.1f us\n",
               s[0], s[1], s[2], t[0] * 1e3, fl / t[0] / 1e9, t[1] * 1e3, fl / t[1] / 1e9, t[2] * 1e3, fl / t[2] / 1e9, t[3] * 1e3, t[4] * 1e3, t[5] * 1e3);
    }
    return 0;
}
