// Probe: what does an in-kernel dependency between ALL workgroups of a launch cost on MI355X (8 XCDs, non-coherent L2s)?
// Every iteration: each thread writes a word write-through (sc0 sc1), waits for it, the workgroup arrives at a monotonic device-scope
// counter, thread 0 spins on it (sc1 loads + s_sleep), then the workgroup invalidates its XCD's L2 view (buffer_inv sc1) and reads a word
// another workgroup wrote.  Prints us per iteration for a few grid sizes, with / without the invalidate, and checks the values.
//   hipcc --offload-arch=gfx950 -O3 -o grid_barrier_probe grid_barrier_probe.hip && ./grid_barrier_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

template <int INV>
__global__ __launch_bounds__(256) void gb_kernel(unsigned* counter, unsigned* buf, unsigned* bad, int iters, int nwg) {
    extern __shared__ char smem[];
    (void)smem;
    const int wg = blockIdx.x, tid = threadIdx.x;
    unsigned errs = 0;
    for (int it = 0; it < iters; ++it) {
        unsigned* mine = buf + ((size_t)(it & 1) * nwg + wg) * 256 + tid;
        const unsigned val = (unsigned)(it * 1000003 + wg * 257 + tid);
        asm volatile("global_store_dword %0, %1, off sc0 sc1\n\ts_waitcnt vmcnt(0)" ::"v"(mine), "v"(val) : "memory");
        __syncthreads();
        if (tid == 0) {
            const unsigned target = (unsigned)(it + 1) * (unsigned)nwg;
            __hip_atomic_fetch_add(counter, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            unsigned seen;
            do {
                asm volatile("global_load_dword %0, %1, off sc0 sc1\n\ts_waitcnt vmcnt(0)" : "=v"(seen) : "v"(counter) : "memory");
                if (seen < target) __builtin_amdgcn_s_sleep(2);
            } while (seen < target);
        }
        __syncthreads();
        if (INV) asm volatile("buffer_inv sc1" ::: "memory");
        const int other = (wg + 37) % nwg;
        const unsigned* theirs = buf + ((size_t)(it & 1) * nwg + other) * 256 + tid;
        unsigned got;
        if (INV) got = *(volatile const unsigned*)theirs;
        else asm volatile("global_load_dword %0, %1, off sc0 sc1\n\ts_waitcnt vmcnt(0)" : "=v"(got) : "v"(theirs) : "memory");
        errs += got != (unsigned)(it * 1000003 + other * 257 + tid);
    }
    if (errs) atomicAdd(bad, errs);
}

int main() {
    unsigned *counter, *buf, *bad;
    const int maxwg = 512, iters = 200;
    hipMalloc(&counter, 4); hipMalloc(&bad, 4); hipMalloc(&buf, (size_t)2 * maxwg * 256 * 4);
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    hipFuncSetAttribute(reinterpret_cast<const void*>(gb_kernel<1>), hipFuncAttributeMaxDynamicSharedMemorySize, 65536);
    hipFuncSetAttribute(reinterpret_cast<const void*>(gb_kernel<0>), hipFuncAttributeMaxDynamicSharedMemorySize, 65536);
    for (int inv = 1; inv >= 0; --inv)
        for (int nwg : {64, 256, 360, 432, 512}) {
            float best = 1e30f; unsigned hbad = 0;
            for (int rep = 0; rep < 3; ++rep) {
                hipMemset(counter, 0, 4); hipMemset(bad, 0, 4);
                hipEventRecord(a);
                if (inv) hipLaunchKernelGGL(gb_kernel<1>, dim3(nwg), dim3(256), 65536, 0, counter, buf, bad, iters, nwg);
                else hipLaunchKernelGGL(gb_kernel<0>, dim3(nwg), dim3(256), 65536, 0, counter, buf, bad, iters, nwg);
                hipEventRecord(b); hipEventSynchronize(b);
                float ms; hipEventElapsedTime(&ms, a, b);
                best = ms < best ? ms : best;
                hipMemcpy(&hbad, bad, 4, hipMemcpyDeviceToHost);
            }
            printf("%3d workgroups, %s: %.2f us per write + grid barrier + read   (wrong values: %u)\n", nwg, inv ? "buffer_inv sc1 + plain load" : "sc1 load, no invalidate", best * 1e3f / iters, hbad);
        }
    return 0;
}
