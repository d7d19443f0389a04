// body of one variant of tools/probes/gemm_w4_probe.hip (included once per variant with KNAME / KINC defined)
__global__ __launch_bounds__(256) void KNAME(const char* A, const char* W, float* out, unsigned* dbg, int K, int nt, int lda, int ldw) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, bx = blockIdx.x;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    int mt, n_;
    if (lda < 0) {                     // naive map: consecutive workgroups (= different XCDs) walk along N
        lda = -lda;
        mt = bx / nt; n_ = bx - mt * nt;
    } else {                           // the library's map: contiguous runs of an (8 M tiles x all N tiles) order, one run per XCD
        const int xcd = bx & 7, idx = bx >> 3, per = (int)gridDim.x >> 3;
        const int L = xcd * per + idx, gi = L / (8 * nt), within = L - gi * 8 * nt;
        n_ = within / 8; mt = gi * 8 + (within - n_ * 8);
    }
    const char* a_base = A + (size_t)mt * 256 * lda * 2;
    const char* w_base = W + (size_t)n_ * 256 * ldw * 2;
    f32x16 accv[16];
    for (int t = 0; t < 16; ++t) for (int r = 0; r < 16; ++r) accv[t][r] = 0.f;
    const unsigned lds0 = (unsigned)(uintptr_t)(__attribute__((address_space(3))) char*)smem;
    const int lda2 = lda * 2, ldw2 = ldw * 2, rmax = 255, nk = K / 64;
    asm volatile(
#include KINC
        : "+{a[0:15]}"(accv[0]), "+{a[16:31]}"(accv[1]), "+{a[32:47]}"(accv[2]), "+{a[48:63]}"(accv[3]),
          "+{a[64:79]}"(accv[4]), "+{a[80:95]}"(accv[5]), "+{a[96:111]}"(accv[6]), "+{a[112:127]}"(accv[7]),
          "+{a[128:143]}"(accv[8]), "+{a[144:159]}"(accv[9]), "+{a[160:175]}"(accv[10]), "+{a[176:191]}"(accv[11]),
          "+{a[192:207]}"(accv[12]), "+{a[208:223]}"(accv[13]), "+{a[224:239]}"(accv[14]), "+{a[240:255]}"(accv[15])
        : [tid] "v"(tid), [ab] "s"(a_base), [wb] "s"(w_base), [lda2] "s"(lda2), [ldw2] "s"(ldw2), [rmax] "s"(rmax), [nk] "s"(nk),
          [lds] "s"(lds0), [dbg] "s"(dbg), [wave] "s"(wave)
        : "memory", "scc", GEMM_W4_SGPRS, GEMM_W4_VGPRS);
    float sum = 0.f;
    for (int t = 0; t < 16; ++t) sum += accv[t][0] + accv[t][7];
    if (sum == 12345.678f) out[bx * 256 + tid] = sum;
}
#undef KNAME
#undef KINC
