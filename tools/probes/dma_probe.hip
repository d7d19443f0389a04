// Probe: LDS-DMA (global_load_lds_dwordx4) streaming rate per workgroup vs ring depth and waves per workgroup.
// Each workgroup streams its own 2 MB slab (L2/MALL resident after warm-up) into an NS-deep LDS ring of 16 KB stages.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
template <int N_> __device__ __forceinline__ void wait_vm() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N_) : "memory"); }

template <int NW, int NS>
__global__ __launch_bounds__(64 * NW) void stream_kernel(const char* __restrict__ src, size_t slab, int tiles, float* out) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    constexpr int STAGE = 16384;
    constexpr int LPT = STAGE / (1024 * NW);           // DMA instructions per wave per tile
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const char* base = src + (size_t)blockIdx.x * slab;
    auto issue = [&](int t) __attribute__((always_inline)) {
#pragma unroll
        for (int i = 0; i < LPT; ++i) {
            const int piece = wave + NW * i;
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(base + (size_t)t * STAGE + piece * 1024 + lane * 16),
                                             (__attribute__((address_space(3))) void*)(smem + (t % NS) * STAGE + piece * 1024), 16, 0, 0);
        }
    };
    float acc = 0.f;
    for (int t = 0; t < NS - 1 && t < tiles; ++t) issue(t);
    for (int t = 0; t < tiles; ++t) {
        const int ahead = tiles - 1 - t;
        if (ahead >= NS - 2) wait_vm<LPT * (NS - 2)>(); else wait_vm<0>();
        __builtin_amdgcn_s_barrier();
        if (t + NS - 1 < tiles) issue(t + NS - 1);
        acc += *reinterpret_cast<const float*>(smem + (t % NS) * STAGE + threadIdx.x * 16);     // touch the tile
    }
    if (acc == 12345.f) out[0] = acc;
}

template <int NW, int NS>
void run(const char* d, size_t slab, int blocks, float* dout) {
    const int tiles = (int)(slab / 16384);
    auto k = stream_kernel<NW, NS>;
    hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, NS * 16384);
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    for (int i = 0; i < 3; ++i) hipLaunchKernelGGL(k, dim3(blocks), dim3(64 * NW), NS * 16384, 0, d, slab, tiles, dout);
    hipEventRecord(a);
    const int it = 10;
    for (int i = 0; i < it; ++i) hipLaunchKernelGGL(k, dim3(blocks), dim3(64 * NW), NS * 16384, 0, d, slab, tiles, dout);
    hipEventRecord(b); hipEventSynchronize(b);
    float ms; hipEventElapsedTime(&ms, a, b); ms /= it;
    const double gbs = (double)slab * blocks / (ms * 1e-3) / 1e9;
    printf("waves/WG %d  ring %d  blocks %4d : %8.1f us  %8.1f GB/s total  %6.1f GB/s per block  (%.0f cycles per 16 KB tile)\n", NW, NS, blocks, ms * 1e3, gbs, gbs / blocks,
           ms * 1e-3 / tiles * 2.1e9);
}

int main() {
    const size_t slab = 1 << 20;          // 1 MB per block
    const int maxb = 1024;
    char* d; float* dout;
    hipMalloc(&d, slab * maxb); hipMemset(d, 1, slab * maxb); hipMalloc(&dout, 64);
    for (int blocks : {64, 256, 512, 1024}) {
        run<4, 2>(d, slab, blocks, dout); run<4, 3>(d, slab, blocks, dout); run<4, 4>(d, slab, blocks, dout); run<4, 8>(d, slab, blocks, dout);
        run<8, 2>(d, slab, blocks, dout); run<8, 4>(d, slab, blocks, dout); run<16, 2>(d, slab, blocks, dout); run<16, 4>(d, slab, blocks, dout);
    }
    return 0;
}
