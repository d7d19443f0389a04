// Probe: do v_exp_f32 (quarter-rate transcendental) and v_mfma_f32_32x32x16_bf16 overlap on one SIMD of gfx950?
//   (a) one wave: 64 MFMAs | 256 exps | both interleaved 1 : 4 | MFMAs interleaved with 256 v_fma instead
//   (b) two waves of one SIMD (waves w and w + 4 of a 512-thread workgroup): one issues MFMAs, the partner exps (or fmas)
// Shader-clock cycles (s_memtime) per wave.  hipcc --offload-arch=gfx950 -O3 tools/probes/exp_mfma_probe.hip -o tools/probes/exp_mfma_probe
#include <hip/hip_runtime.h>
#include <cstdio>
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(16))) float f32x16;

template <int MODE>   // 0 mfma only, 1 exp only, 2 mfma+exp interleaved, 3 mfma+fma interleaved, 4 fma only
__device__ __forceinline__ unsigned long long body(float* sink, int lane) {
    f32x16 a0, a1;
    for (int r = 0; r < 16; ++r) { a0[r] = lane * 0.001f; a1[r] = lane * 0.002f; }
    bf16x8 x, y;
    for (int e = 0; e < 8; ++e) { x[e] = (__bf16)(0.01f * (lane + e)); y[e] = (__bf16)(0.02f * (lane - e)); }
    float v[8];
    for (int e = 0; e < 8; ++e) v[e] = 0.001f * (lane + e);
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_sched_barrier(0);
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int i = 0; i < 32; ++i) {
        if (MODE == 0 || MODE == 2 || MODE == 3) {
            a0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(x, y, a0, 0, 0, 0);
            a1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(x, y, a1, 0, 0, 0);
        }
        if (MODE == 1 || MODE == 2) {
#pragma unroll
            for (int e = 0; e < 8; ++e) v[e] = __builtin_amdgcn_exp2f(v[e]);
        }
        if (MODE == 3 || MODE == 4) {
#pragma unroll
            for (int e = 0; e < 8; ++e) v[e] = __builtin_fmaf(v[e], 0.999f, 0.001f);
        }
    }
    float acc = 0.f;
    for (int r = 0; r < 16; ++r) acc += a0[r] + a1[r];
    for (int e = 0; e < 8; ++e) acc += v[e];
    asm volatile("" ::"v"(acc));
    __builtin_amdgcn_sched_barrier(0);
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_sched_barrier(0);
    if (acc == 12345.678f) sink[0] = acc;
    return t1 - t0;
}

__global__ __launch_bounds__(512) void probe(unsigned long long* out, float* sink, int mode_a, int mode_b, int partner) {
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    unsigned long long dt = 0;
    const int mode = wave == 0 ? mode_a : (wave == partner ? mode_b : -1);
    __syncthreads();
    switch (mode) {
        case 0: dt = body<0>(sink, lane); break;
        case 1: dt = body<1>(sink, lane); break;
        case 2: dt = body<2>(sink, lane); break;
        case 3: dt = body<3>(sink, lane); break;
        case 4: dt = body<4>(sink, lane); break;
        default: break;
    }
    if (lane == 0 && (wave == 0 || wave == partner)) out[blockIdx.x * 2 + (wave != 0)] = dt;
}

int main() {
    unsigned long long* out; float* sink;
    hipMalloc(&out, 16 * sizeof(unsigned long long)); hipMalloc(&sink, 64);
    const char* names[5] = {"64 MFMA", "256 v_exp_f32", "64 MFMA + 256 v_exp interleaved", "64 MFMA + 256 v_fma interleaved", "256 v_fma"};
    auto run = [&](int a, int b, int partner) {
        unsigned long long h[2] = {0, 0};
        for (int rep = 0; rep < 3; ++rep) {
            hipMemset(out, 0, 16 * 8);
            hipLaunchKernelGGL(probe, dim3(1), dim3(512), 0, 0, out, sink, a, b, partner);
            hipDeviceSynchronize();
            hipMemcpy(h, out, 16, hipMemcpyDeviceToHost);
        }
        printf("wave 0: %-36s %6llu cycles", names[a], h[0]);
        if (b >= 0) printf("   | wave %d: %-36s %6llu cycles", partner, names[b], h[1]);
        printf("\n");
    };
    for (int m : {0, 1, 4, 2, 3}) run(m, -1, 4);
    for (int k = 1; k < 8; ++k) run(0, 0, k);            // which wave shares wave 0's SIMD (matrix pipe)?
    for (int k = 1; k < 8; ++k) run(0, 1, k);
    for (int k : {1, 2, 3, 4}) { run(0, 4, k); run(1, 1, k); run(2, 2, k); }
    return 0;
}
