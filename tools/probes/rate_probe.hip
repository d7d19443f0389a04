// Probe: issue / pipe rates on one SIMD of gfx950 by WALL time (hipEvents) and s_memtime, at 1 and 2 waves per SIMD:
//   v_exp_f32, v_add_f32, v_mfma_f32_32x32x16_bf16 alone, and the attention tile's mix (1 MFMA : 2 exp : 3 plain VALU).
// Every CU gets workgroups of 256 x WPS threads; cycles = wall x clock, clock calibrated by the MFMA-only run at 2 waves / SIMD
// (pipe-bound: 32 cycles per MFMA).   hipcc --offload-arch=gfx950 -O3 tools/probes/rate_probe.hip -o tools/probes/rate_probe
#include <hip/hip_runtime.h>
#include <cstdio>
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(16))) float f32x16;
#define SB() __builtin_amdgcn_sched_barrier(0)
__device__ __forceinline__ float vadd(float a, float b) { float d; asm volatile("v_add_f32 %0, %1, %2" : "=v"(d) : "v"(a), "v"(b)); return d; }
__device__ __forceinline__ float vexp(float a) { float d; asm volatile("v_exp_f32 %0, %1" : "=v"(d) : "v"(a)); return d; }

template <int MODE>
__global__ void probe(float* sink, unsigned long long* ticks, int iters) {
    const int lane = threadIdx.x & 63;
    f32x16 a0, a1, a2, a3;
    for (int r = 0; r < 16; ++r) { a0[r] = lane * 0.001f; a1[r] = lane * 0.002f; a2[r] = 0.f; a3[r] = 1.f; }
    bf16x8 x, y;
    for (int e = 0; e < 8; ++e) { x[e] = (__bf16)(0.01f * (lane + e)); y[e] = (__bf16)(0.02f * (lane - e)); }
    float v[8], w[8];
    for (int e = 0; e < 8; ++e) { v[e] = -0.001f * (lane + e); w[e] = 0.5f * e; }
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int i = 0; i < iters; ++i) {
        SB();
        if (MODE == 0) {
#pragma unroll
            for (int k = 0; k < 4; ++k)
#pragma unroll
                for (int e = 0; e < 8; ++e) v[e] = vexp(v[e]);
        } else if (MODE == 1) {
#pragma unroll
            for (int k = 0; k < 4; ++k)
#pragma unroll
                for (int e = 0; e < 8; ++e) v[e] = vadd(v[e], w[e]);
        } else if (MODE == 2) {
#pragma unroll
            for (int k = 0; k < 2; ++k) {
                a0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(x, y, a0, 0, 0, 0);
                a1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(x, y, a1, 0, 0, 0);
                a2 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(x, y, a2, 0, 0, 0);
                a3 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(x, y, a3, 0, 0, 0);
            }
        } else if (MODE == 3 || MODE == 4 || MODE == 5) {   // 8 MFMAs, each followed by 2 exp (+ 3 plain VALU in mode 3, + 2 exp more in mode 5)
#pragma unroll
            for (int k = 0; k < 8; ++k) {
                if ((k & 3) == 0) a0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(x, y, a0, 0, 0, 0);
                if ((k & 3) == 1) a1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(x, y, a1, 0, 0, 0);
                if ((k & 3) == 2) a2 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(x, y, a2, 0, 0, 0);
                if ((k & 3) == 3) a3 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(x, y, a3, 0, 0, 0);
                v[k] = vexp(v[k]);
                v[(k + 4) & 7] = vexp(v[(k + 4) & 7]);
                if (MODE == 5) { w[k] = vexp(w[k]); w[(k + 4) & 7] = vexp(w[(k + 4) & 7]); }
                if (MODE == 3) {
                    w[k] = vadd(w[k], v[(k + 1) & 7]);
                    w[(k + 2) & 7] = vadd(w[(k + 2) & 7], v[(k + 3) & 7]);
                    w[(k + 5) & 7] = vadd(w[(k + 5) & 7], v[(k + 6) & 7]);
                }
                SB();
            }
        } else if (MODE == 6) {   // 8 MFMAs, each followed by 5 plain VALU
#pragma unroll
            for (int k = 0; k < 8; ++k) {
                if ((k & 3) == 0) a0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(x, y, a0, 0, 0, 0);
                if ((k & 3) == 1) a1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(x, y, a1, 0, 0, 0);
                if ((k & 3) == 2) a2 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(x, y, a2, 0, 0, 0);
                if ((k & 3) == 3) a3 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(x, y, a3, 0, 0, 0);
#pragma unroll
                for (int e = 0; e < 5; ++e) w[(k + e) & 7] = vadd(w[(k + e) & 7], v[e]);
                SB();
            }
        }
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    float acc = 0.f;
    for (int r = 0; r < 16; ++r) acc += a0[r] + a1[r] + a2[r] + a3[r];
    for (int e = 0; e < 8; ++e) acc += v[e] + w[e];
    if (acc == 12345.678f) sink[0] = acc;
    if (blockIdx.x == 0 && threadIdx.x == 0) ticks[0] = t1 - t0;
}

template <int MODE>
static void run(const char* name, int wps, int iters, double ops_per_iter, float* sink, unsigned long long* ticks) {
    hipEvent_t a, b;
    hipEventCreate(&a); hipEventCreate(&b);
    probe<MODE><<<256, 256 * wps>>>(sink, ticks, 16);
    hipDeviceSynchronize();
    hipEventRecord(a);
    probe<MODE><<<256, 256 * wps>>>(sink, ticks, iters);
    hipEventRecord(b);
    hipDeviceSynchronize();
    float ms; hipEventElapsedTime(&ms, a, b);
    unsigned long long t; hipMemcpy(&t, ticks, 8, hipMemcpyDeviceToHost);
    const double ns_per_group = ms * 1e6 / iters;     // per wave: one loop iteration
    printf("%-44s wps %d  %8.1f ns / iteration / wave   %8.2f ns per op-unit   ticks/iter %.1f  (tick = %.3f ns)\n", name, wps, ns_per_group,
           ns_per_group / ops_per_iter, (double)t / iters, ms * 1e6 / (double)t);
}

int main() {
    float* sink; unsigned long long* ticks;
    hipMalloc(&sink, 4); hipMalloc(&ticks, 8);
    const int it = 20000;
    for (int wps = 1; wps <= 2; ++wps) {
        run<2>("mfma x8 / iter", wps, it, 8, sink, ticks);
        run<0>("v_exp x32 / iter", wps, it, 32, sink, ticks);
        run<1>("v_add x32 / iter", wps, it, 32, sink, ticks);
        run<4>("8 x (mfma + 2 exp) / iter", wps, it, 8, sink, ticks);
        run<5>("8 x (mfma + 4 exp) / iter", wps, it, 8, sink, ticks);
        run<3>("8 x (mfma + 2 exp + 3 add) / iter", wps, it, 8, sink, ticks);
        run<6>("8 x (mfma + 5 add) / iter", wps, it, 8, sink, ticks);
    }
    return 0;
}
