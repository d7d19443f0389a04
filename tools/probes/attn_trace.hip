// Probe: shader-clock timeline of ONE wave of the persistent attention kernel (phase stamps per key tile), plus ablation timings.
// Built with -DUVL_ATTN_TRACE from the library's own source:  hipcc --offload-arch=gfx950 -O3 -std=c++17 -mllvm -amdgpu-mfma-vgpr-form=1
//   -ffp-contract=on -DUVL_ATTN_TRACE -I include -I uvltrack_amd/csrc tools/probes/attn_trace.hip -o tools/probes/attn_trace
// usage: attn_trace [B H N cfg block wave]
#include "../../uvltrack_amd/csrc/attention.hip"
#include <cstdlib>
#include <cstring>
#include <vector>
namespace uvl { thread_local const char* g_last_kernel = "?"; }
using namespace uvl;

static uint16_t f2b(float f) { uint32_t u; memcpy(&u, &f, 4); u += 0x7fff + ((u >> 16) & 1); return (uint16_t)(u >> 16); }

int main(int argc, char** argv) {
    const int B = argc > 1 ? atoi(argv[1]) : 32, H = argc > 2 ? atoi(argv[2]) : 16, N = argc > 3 ? atoi(argv[3]) : 681;
    const int cfg = argc > 4 ? atoi(argv[4]) : 12, blk = argc > 5 ? atoi(argv[5]) : 8, wv = argc > 6 ? atoi(argv[6]) : 1;
    const int Npad = (N + 63) / 64 * 64;
    const size_t nq = (size_t)B * H * Npad * 64;
    std::vector<uint16_t> hq(nq), hk(nq), hv(nq);
    srand(1);
    auto rnd = []() { float s = 0; for (int i = 0; i < 12; ++i) s += rand() / (float)RAND_MAX; return s - 6.f; };
    for (size_t i = 0; i < nq; ++i) { hq[i] = f2b(rnd() * 0.18033688f); hk[i] = f2b(rnd()); hv[i] = f2b(rnd()); }
    bf16_t *q, *k, *vt, *o; float* add; int* trace;
    hipMalloc(&q, nq * 2); hipMalloc(&k, nq * 2); hipMalloc(&vt, nq * 2); hipMalloc(&o, (size_t)B * N * H * 64 * 2);
    hipMalloc(&add, (size_t)B * Npad * 4); hipMemset(add, 0, (size_t)B * Npad * 4); hipMalloc(&trace, 256 * 4); hipMemset(trace, 0, 1024);
    hipMemcpy(q, hq.data(), nq * 2, hipMemcpyHostToDevice); hipMemcpy(k, hk.data(), nq * 2, hipMemcpyHostToDevice); hipMemcpy(vt, hv.data(), nq * 2, hipMemcpyHostToDevice);
    AttnParams p;
    p.q = q; p.k = k; p.vt = vt; p.key_add = add; p.key_add_stride = Npad; p.o = o; p.B = B; p.H = H; p.N = N; p.Npad = Npad; p.q_prescaled = 1;
    g_tune_attn_cfg = cfg;
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    for (int abl : {0, 1, 2, 3}) {
        g_tune_attn_abl = abl;
        for (int i = 0; i < 5; ++i) launch_attention(p, 0);
        hipEventRecord(a);
        for (int i = 0; i < 20; ++i) launch_attention(p, 0);
        hipEventRecord(b); hipEventSynchronize(b);
        float ms; hipEventElapsedTime(&ms, a, b);
        const double us = ms * 1e3 / 20, fl = 4.0 * N * N * H * 64.0 * B;
        printf("cfg %d abl %d: %8.1f us  %7.1f TFLOP/s  (%s)\n", cfg, abl, us, fl / us / 1e6, g_last_kernel);
    }
    g_tune_attn_abl = 0;
    p.trace = trace; p.trace_block = blk; p.trace_wave = wv;
    launch_attention(p, 0);
    hipDeviceSynchronize();
    std::vector<int> ht(256);
    hipMemcpy(ht.data(), trace, 1024, hipMemcpyDeviceToHost);
    const int n = ht[192];
    printf("trace of block %d wave %d: %d stamps\n", blk, wv, n);
    for (int i = 0; i + 1 < n && i < 191; ++i) {
        printf("%6u%s", (unsigned)(ht[i + 1] - ht[i]), ((i + 1) % 8 == 0) ? "\n" : " ");
    }
    printf("\n");
    return 0;
}
