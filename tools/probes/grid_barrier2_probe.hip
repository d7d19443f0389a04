// Probe 2: a HIERARCHICAL grid barrier.  Workgroup b belongs to group b % 8 (the XCD the dispatcher puts it on); every iteration
//   all:     write-through store of a word, wait for it, arrive at the group's counter (device-scope atomic, 8 addresses in parallel)
//   leader:  the workgroup whose arrival completes its group arrives at the global counter, spins until all 8 groups are there,
//            invalidates ITS XCD's L2 (buffer_inv sc1: one per XCD instead of one per workgroup) and releases its group
//   others:  spin on the group's release word
// then every thread reads, THROUGH the L2 (plain load), a word a workgroup of another group wrote.  us per iteration + wrong values.
//   hipcc --offload-arch=gfx950 -O3 -o grid_barrier2_probe grid_barrier2_probe.hip && ./grid_barrier2_probe
#include <hip/hip_runtime.h>
#include <cstdio>

__device__ __forceinline__ unsigned ld_sc1(const unsigned* p) {
    unsigned v;
    asm volatile("global_load_dword %0, %1, off sc0 sc1\n\ts_waitcnt vmcnt(0)" : "=v"(v) : "v"(p) : "memory");
    return v;
}

template <int INV>
__global__ __launch_bounds__(256) void gb2_kernel(unsigned* ctr /* [8] group, [8] release, [1] global; 64-byte spaced */, unsigned* buf, unsigned* bad, int iters, int nwg) {
    extern __shared__ char smem[];
    (void)smem;
    const int wg = blockIdx.x, tid = threadIdx.x, grp = wg & 7;
    const unsigned gsize = (unsigned)((nwg - grp + 7) / 8);
    unsigned* gcnt = ctr + grp * 16;
    unsigned* grel = ctr + 128 + grp * 16;
    unsigned* glob = ctr + 256;
    unsigned errs = 0;
    for (int it = 0; it < iters; ++it) {
        unsigned* mine = buf + ((size_t)(it & 1) * nwg + wg) * 256 + tid;
        const unsigned val = (unsigned)(it * 1000003 + wg * 257 + tid);
        asm volatile("global_store_dword %0, %1, off sc0 sc1\n\ts_waitcnt vmcnt(0)" ::"v"(mine), "v"(val) : "memory");
        __syncthreads();
        if (tid == 0) {
            const unsigned prev = __hip_atomic_fetch_add(gcnt, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (prev + 1 == (unsigned)(it + 1) * gsize) {                      // this arrival completes the group
                __hip_atomic_fetch_add(glob, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                while (ld_sc1(glob) < (unsigned)(it + 1) * 8u) __builtin_amdgcn_s_sleep(1);
                if (INV) asm volatile("buffer_inv sc1" ::: "memory");
                const unsigned rv = (unsigned)(it + 1);
                asm volatile("global_store_dword %0, %1, off sc0 sc1\n\ts_waitcnt vmcnt(0)" ::"v"(grel), "v"(rv) : "memory");
            } else {
                while (ld_sc1(grel) < (unsigned)(it + 1)) __builtin_amdgcn_s_sleep(1);
            }
        }
        __syncthreads();
        const int other = (wg + 37) % nwg;
        const unsigned* theirs = buf + ((size_t)(it & 1) * nwg + other) * 256 + tid;
        unsigned got;
        if (INV) got = *(volatile const unsigned*)theirs;
        else got = ld_sc1(theirs);
        errs += got != (unsigned)(it * 1000003 + other * 257 + tid);
    }
    if (errs) atomicAdd(bad, errs);
}

int main() {
    unsigned *ctr, *buf, *bad;
    const int maxwg = 512, iters = 200;
    if (hipMalloc(&ctr, 4096) != hipSuccess || hipMalloc(&bad, 4) != hipSuccess || hipMalloc(&buf, (size_t)2 * maxwg * 256 * 4) != hipSuccess) return 1;
    hipEvent_t a, b;
    (void)hipEventCreate(&a); (void)hipEventCreate(&b);
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(gb2_kernel<1>), hipFuncAttributeMaxDynamicSharedMemorySize, 65536);
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(gb2_kernel<0>), hipFuncAttributeMaxDynamicSharedMemorySize, 65536);
    for (int inv = 1; inv >= 0; --inv)
        for (int nwg : {64, 256, 360, 432, 512}) {
            float best = 1e30f; unsigned hbad = 0;
            for (int rep = 0; rep < 3; ++rep) {
                (void)hipMemset(ctr, 0, 4096); (void)hipMemset(bad, 0, 4);
                (void)hipEventRecord(a);
                if (inv) hipLaunchKernelGGL(gb2_kernel<1>, dim3(nwg), dim3(256), 65536, 0, ctr, buf, bad, iters, nwg);
                else hipLaunchKernelGGL(gb2_kernel<0>, dim3(nwg), dim3(256), 65536, 0, ctr, buf, bad, iters, nwg);
                (void)hipEventRecord(b); (void)hipEventSynchronize(b);
                float ms; (void)hipEventElapsedTime(&ms, a, b);
                best = ms < best ? ms : best;
                (void)hipMemcpy(&hbad, bad, 4, hipMemcpyDeviceToHost);
            }
            printf("%3d workgroups, hierarchical, %s: %.2f us per write + grid barrier + read   (wrong values: %u)\n", nwg,
                   inv ? "one buffer_inv per group + plain load" : "sc1 load, no invalidate", best * 1e3f / iters, hbad);
        }
    return 0;
}
