// Debugging aid of the generated K loop of gemm_w4_kernel: runs only its prologue (tools/gen/gemm_w4_gen.py --abl prologue_only) on one
// workgroup and dumps the LDS image, to check the LDS-DMA placement against the image the fragment reads expect.
//   python tools/gen/gemm_w4_gen.py --out tools/probes/w4/prologue.inc --abl prologue_only
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -I tools/probes/w4 tools/probes/w4_lds_dump.hip -o tools/probes/w4_lds_dump
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
__global__ __launch_bounds__(256) void k_dump(const char* A, const char* W, unsigned* out, int K) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    for (int i = tid; i < 131072 / 4; i += 256) reinterpret_cast<unsigned*>(smem)[i] = 0xdeadbeefu;
    __syncthreads();
    f32x16 accv[16];
    for (int t = 0; t < 16; ++t) for (int r = 0; r < 16; ++r) accv[t][r] = 0.f;
    const unsigned lds0 = (unsigned)(uintptr_t)(__attribute__((address_space(3))) char*)smem;
    const int lda2 = K * 2, ldw2 = K * 2, rmax = 255, nk = K / 64;
#if __HIP_DEVICE_COMPILE__
    asm volatile(
#include "prologue.inc"
        : "+{a[0:15]}"(accv[0]), "+{a[16:31]}"(accv[1]), "+{a[32:47]}"(accv[2]), "+{a[48:63]}"(accv[3]),
          "+{a[64:79]}"(accv[4]), "+{a[80:95]}"(accv[5]), "+{a[96:111]}"(accv[6]), "+{a[112:127]}"(accv[7]),
          "+{a[128:143]}"(accv[8]), "+{a[144:159]}"(accv[9]), "+{a[160:175]}"(accv[10]), "+{a[176:191]}"(accv[11]),
          "+{a[192:207]}"(accv[12]), "+{a[208:223]}"(accv[13]), "+{a[224:239]}"(accv[14]), "+{a[240:255]}"(accv[15])
        : [tid] "v"(tid), [ab] "s"(A), [wb] "s"(W), [lda2] "s"(lda2), [ldw2] "s"(ldw2), [rmax] "s"(rmax), [nk] "s"(nk), [lds] "s"(lds0), [wave] "s"(wave)
        : "memory", "scc", GEMM_W4_SGPRS, GEMM_W4_VGPRS);
#endif
    __syncthreads();
    for (int i = tid; i < 131072 / 4; i += 256) out[i] = reinterpret_cast<unsigned*>(smem)[i];
    if (accv[0][0] == 1234.5f) out[0] = 0;
}
int main() {
    const int K = 256;
    // element (row, k) of A = row * 256 + k as a 16-bit code; W = 0x8000 | the same
    std::vector<unsigned short> ha(256 * K), hw(256 * K);
    for (int r = 0; r < 256; ++r) for (int k = 0; k < K; ++k) { ha[r * K + k] = (unsigned short)((r << 7) | (k >> 1)); hw[r * K + k] = (unsigned short)(0x8000 | (r << 7) | (k >> 1)); }
    char *A, *W; unsigned* out;
    hipMalloc(&A, ha.size() * 2); hipMalloc(&W, hw.size() * 2); hipMalloc(&out, 131072);
    hipMemcpy(A, ha.data(), ha.size() * 2, hipMemcpyHostToDevice); hipMemcpy(W, hw.data(), hw.size() * 2, hipMemcpyHostToDevice);
    hipFuncSetAttribute(reinterpret_cast<const void*>(k_dump), hipFuncAttributeMaxDynamicSharedMemorySize, 132096);
    hipLaunchKernelGGL(k_dump, dim3(1), dim3(256), 132096, 0, A, W, out, K);
    hipDeviceSynchronize();
    std::vector<unsigned> h(131072 / 4);
    hipMemcpy(h.data(), out, 131072, hipMemcpyDeviceToHost);
    printf("hipError %d\n", (int)hipGetLastError());
    for (int blk = 0; blk < 128; ++blk) {      // what each 1-KB block (8 rows) holds
        const unsigned g = h[blk * 256];
        if (g == 0xdeadbeefu) continue;
        printf("block %3d (byte %6d): %c row %3d k %3d\n", blk, blk * 1024, (g & 0x8000) ? 'W' : 'A', (g >> 7) & 0xff, (g & 0x7f) * 2);
    }
    // expected: buffer b (tile b), region (A at 0, W at 32768), row r, position p holds chunk c = p ^ ((r >> 1) & 7): 8 elements (r, 64 b + 8 c + e)
    for (int b = 0; b < 2; ++b)
        for (int reg = 0; reg < 2; ++reg) {
            int good = 0, dead = 0, other = 0;
            for (int r = 0; r < 256; ++r)
                for (int p = 0; p < 8; ++p)
                    for (int d = 0; d < 4; ++d) {
                        const unsigned got = h[(b * 65536 + reg * 32768 + r * 128 + p * 16 + d * 4) / 4];
                        const int c = p ^ ((r >> 1) & 7), k = 64 * b + 8 * c + 2 * d;
                        const unsigned e0 = (reg ? 0x8000 : 0) | (r << 7) | (k >> 1), e1 = (reg ? 0x8000 : 0) | (r << 7) | ((k + 1) >> 1);
                        if (got == (e0 | (e1 << 16))) ++good; else if (got == 0xdeadbeefu) ++dead; else ++other;
                    }
            printf("buffer %d %s: %d dwords as expected, %d untouched, %d other\n", b, reg ? "W" : "A", good, dead, other);
            if (other) {
                int shown = 0;
                for (int r = 0; r < 256 && shown < 6; ++r) for (int p = 0; p < 8 && shown < 6; ++p) {
                    const unsigned got = h[(b * 65536 + reg * 32768 + r * 128 + p * 16) / 4];
                    const int c = p ^ ((r >> 1) & 7), k = 64 * b + 8 * c;
                    const unsigned e0 = (reg ? 0x8000 : 0) | (r << 7) | (k >> 1);
                    if ((got & 0xffff) != e0 && got != 0xdeadbeefu) { printf("  row %d pos %d: got %08x (row %d k %d), expected row %d k %d\n", r, p, got, (got >> 7) & 0xff, (got & 0x7f) * 2, r, k); ++shown; }
                }
            }
        }
    return 0;
}
