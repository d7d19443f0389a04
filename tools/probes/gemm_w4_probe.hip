// Where the K loop of gemm_w4_kernel spends its time: the generated loop (tools/gen/gemm_w4_gen.py) and timing-only ablations of it
// (no LDS-DMA / no fragment reads / no barrier: results wrong by construction) on 256 or 1024 workgroups, with the
// loop's shader-clock cycles and constant-clock time per K tile (-> sustained shader clock) of one workgroup's waves.
//   build:  python tools/gen/gemm_w4_gen.py --out tools/probes/w4/<v>.inc --abl <...> --trace   for v in full noload(nodma) noread nobar mfma(all three)
//           hipcc --offload-arch=gfx950 -O3 -std=c++17 -I tools/probes/w4 tools/probes/gemm_w4_probe.hip -o tools/probes/gemm_w4_probe
//   run (GPU box):  tools/probes/gemm_w4_probe [K] [tiles]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
typedef float f32x16 __attribute__((ext_vector_type(16)));
#define W4_R10(p, n) p #n "0", p #n "1", p #n "2", p #n "3", p #n "4", p #n "5", p #n "6", p #n "7", p #n "8", p #n "9"
#define GEMM_W4_SGPRS W4_R10("s", 4), W4_R10("s", 5), W4_R10("s", 6)
#define GEMM_W4_VGPRS "v0", "v1", "v2", "v3", "v4", "v5", "v6", "v7", "v8", "v9", W4_R10("v", 1), W4_R10("v", 2), W4_R10("v", 3), W4_R10("v", 4), \
    W4_R10("v", 5), W4_R10("v", 6), W4_R10("v", 7), W4_R10("v", 8), W4_R10("v", 9), W4_R10("v", 10), W4_R10("v", 11), W4_R10("v", 12), \
    W4_R10("v", 13), W4_R10("v", 14), W4_R10("v", 15), "v160", "v161"

// one kernel per variant: w4_probe_body.h is included with KNAME / KINC set (an #include cannot sit inside a macro argument)
#if __HIP_DEVICE_COMPILE__
#define KNAME k_full
#define KINC "full.inc"
#include "w4_probe_body.h"
#define KNAME k_noload
#define KINC "noload.inc"
#include "w4_probe_body.h"
#define KNAME k_noread
#define KINC "noread.inc"
#include "w4_probe_body.h"
#define KNAME k_nobar
#define KINC "nobar.inc"
#include "w4_probe_body.h"
#define KNAME k_mfma
#define KINC "mfma.inc"
#include "w4_probe_body.h"
#else
#define STUB(NAME) __global__ void NAME(const char* A, const char* W, float* out, unsigned* dbg, int K, int nt, int lda, int ldw) {}
STUB(k_full) STUB(k_noload) STUB(k_noread) STUB(k_nobar) STUB(k_mfma)
#endif

int main(int argc, char** argv) {
    const int K = argc > 1 ? atoi(argv[1]) : 4096, tiles = argc > 2 ? atoi(argv[2]) : 256, naive = argc > 3 ? atoi(argv[3]) : 0;
    const int nt = 16, mtiles = tiles / nt, M = mtiles * 256, N = nt * 256;
    char *A, *W; float* out; unsigned* dbg;
    hipMalloc(&A, (size_t)M * K * 2); hipMalloc(&W, (size_t)N * K * 2); hipMalloc(&out, (size_t)tiles * 256 * 4); hipMalloc(&dbg, 64);
    {   // bf16 values in [-1, 1)
        std::vector<unsigned short> h((size_t)(M > N ? M : N) * K);
        for (size_t i = 0; i < h.size(); ++i) h[i] = (unsigned short)(0x3c00 + (rand() & 0x3ff) + ((rand() & 1) << 15));
        hipMemcpy(A, h.data(), (size_t)M * K * 2, hipMemcpyHostToDevice);
        hipMemcpy(W, h.data(), (size_t)N * K * 2, hipMemcpyHostToDevice);
    }
    struct { const char* name; void (*fn)(const char*, const char*, float*, unsigned*, int, int, int, int); } ks[] = {
        {"full", k_full}, {"no barrier", k_nobar}, {"no LDS-DMA", k_noload}, {"no fragment reads", k_noread}, {"MFMAs only", k_mfma}};
    const size_t lds = 2 * 65536 + 1024;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    printf("K = %d, %d workgroups (M = %d, N = %d), %s tile map\n", K, tiles, M, N, naive ? "naive" : "XCD-grouped");
    for (int rep = 0; rep < 2; ++rep)
        for (auto& k : ks) {
            hipFuncSetAttribute(reinterpret_cast<const void*>(k.fn), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
            for (int i = 0; i < 3; ++i) hipLaunchKernelGGL(k.fn, dim3(tiles), dim3(256), lds, 0, A, W, out, dbg, K, nt, naive ? -K : K, K);
            hipEventRecord(e0);
            const int it = 10;
            for (int i = 0; i < it; ++i) hipLaunchKernelGGL(k.fn, dim3(tiles), dim3(256), lds, 0, A, W, out, dbg, K, nt, naive ? -K : K, K);
            hipEventRecord(e1); hipEventSynchronize(e1);
            float ms; hipEventElapsedTime(&ms, e0, e1);
            unsigned h[8]; hipMemcpy(h, dbg, 32, hipMemcpyDeviceToHost);
            const double us = ms * 1e3 / it, nk = K / 64.0;      // per 64 of K = two stages
            if (rep) printf("%-18s %8.1f us/launch %7.1f TFLOP/s | per K tile: %6.0f shader cycles, %5.3f us -> %4.2f GHz  (hipError %d)\n", k.name, us,
                            2.0 * M * N * K / us / 1e6, h[0] / nk, h[1] / 100.0 / nk, h[0] / (h[1] / 100.0) / 1e3, (int)hipGetLastError());
        }
    return 0;
}
