// The end of the box head as ONE launch (round 6): the last 3x3 layer of the conv towers (C/4 -> C/8 channels, heads/utils.py:126-131 with BatchNorm folded,
// modality_adaptive_box_head.py:28-50), their 1x1 convs, the sigmoids, the size-map select by flag, convert2bbox and the argmax (head:62-94,108-119).  Rounds 1-5 ran
// the layer as a 16-tile GEMM launch (8.4 us at one sequence: K = 576 is 9 K tiles on 16 workgroups) and head_tail_kernel (7.6 us) behind it; the tail needs every
// position of the sample (the argmax) and a tower of the layer is 9.4 MFLOP, 1 us of ONE CU's matrix pipe -- less than the launch boundary + the bf16 round trip
// through HBM + the second kernel's own load latency.
//
// Geometry (the shipped heads at 256-px search crops: HEAD_DIM 256, 16 x 16 search features): a tower's input is 256 positions x 64 channels = 32 KB of bf16 and sits
// in LDS whole (LDS-DMA, 16-byte chunks XOR-swizzled by position so that the 16 lanes of an operand read hit 16 different bank groups); the weights stream from L2
// into registers in MFMA-operand order (packed once at uvl_finalize_weights: head_fin_pack_kernel), double-buffered per (column tap, 32-channel half).
// v_mfma_f32_16x16x32_bf16 with the 16 positions of ONE image row as the operand's columns: a row's operand for column tap dj serves the three row taps di (output
// rows r - di) and both 16-channel halves of the tower's 32 outputs -- six MFMAs per 1 KB LDS read.
// Other geometries (24 x 24 search features of UVLTrack-L at 384 px) keep the two launches.
// Measured (cycle stamps per wave, one sequence, profiles/r06_head_fin.md): 13.3k cycles = 1.2-1.6k to the last request + 1.9k until everything has landed + 6.0k for
// the 288 MFMAs of a wave (21 cycles each; 17 is the pipe's pace) + 2.4k for the 1x1 conv + 1.3k for the tail; 7.1 us in the frame against 8.4 + 7.6 us.
#include "common.h"
#include "kernels.h"

namespace uvl {

namespace {
constexpr int HF_S = 256, HF_F = 16, HF_CIN = 64, HF_COUT = 32, HF_K = 9 * HF_CIN;
constexpr int HF_FRAGS = 4 * 6 * 3 * 2;            // [tower][column tap x channel half][row tap][16-channel output half] fragments of 64 lanes x 16 bytes

// A tower's input in LDS: position p = 128 bytes (64 channels) as eight 16-byte chunks, chunk c stored at c ^ hf_swz(p): the 16 consecutive positions of an operand
// read (one image row, shifted by the column tap) then hit 16 different (128-byte half, chunk) bank groups
__device__ __forceinline__ int hf_swz(int p) { return (p >> 1) & 7; }
constexpr int HF_ROW = 16 * 128;                   // an image row of a tower in LDS
constexpr int HF_SLOT = 18 * HF_ROW;               // a tower: image rows -1 .. 16 (the two outside ones zero)
constexpr int HF_LDS = 2 * HF_SLOT + 5 * 4096;     // two towers + the zeros a lane outside the image reads (10 row pitches of offsets)

}  // namespace

// w: [4 towers][32 outputs][K = tap * 64 + channel] bf16 (the frame's conv layout) -> wf: fragment (g, it = 2 dj + ks, di, nb), lane l = the operand of output
// nb * 16 + (l & 15), channels ks * 32 + (l >> 4) * 8 .. + 8 of tap (di, dj)
__global__ __launch_bounds__(256) void head_fin_pack_kernel(const bf16_t* __restrict__ w, bf16_t* __restrict__ wf) {
    const int idx = blockIdx.x * 256 + threadIdx.x;
    if (idx >= HF_FRAGS * 64) return;
    const int lane = idx & 63;
    int r = idx >> 6;
    const int nb = r & 1; r >>= 1;
    const int di = r % 3; r /= 3;
    const int it = r % 6, g = r / 6;
    const int dj = it >> 1, ks = it & 1;
    const int n = nb * 16 + (lane & 15), c = ks * 32 + (lane >> 4) * 8, tap = di * 3 + dj;
    reinterpret_cast<uint4*>(wf)[idx] = *reinterpret_cast<const uint4*>(w + (size_t)(g * HF_COUT + n) * HF_K + tap * HF_CIN + c);
}
hipError_t launch_head_fin_pack(const bf16_t* w, bf16_t* wf, hipStream_t s) {
    hipLaunchKernelGGL(head_fin_pack_kernel, dim3((HF_FRAGS * 64 + 255) / 256), dim3(256), 0, s, w, wf);
    return hipGetLastError();
}

// Two workgroups per sample, neither waits for the other: both run the cls tower (the argmax needs every position's score; the two copies are the same instructions on the
// same data, hence the same bits), workgroup 0 adds the offset tower and writes the scores, the box centres and the argmax, workgroup 1 adds the ONE size tower the
// sample's flag selects (head:80-82: bbox_grounding for flag 1, bbox otherwise -- the other one is never read) and writes the box sizes.  Four waves, one per SIMD:
// wave = (tower slot, image rows [8 h, + 8)) with all 16 accumulators of its 8 rows x 2 channel halves in registers, so that a row operand read from LDS feeds six
// MFMAs and a weight fragment fetched from L2 feeds eight (49 B/clk of LDS reads and 29 B/clk of L2 reads per CU at the matrix pipe's pace; the eight-wave forms with
// a quarter of the rows or one channel half per wave need 98 or 59 + 59).
__global__ __launch_bounds__(256) void head_fin_kernel(const HeadFinParams q) {
    kernarg_warm<sizeof(HeadFinParams)>();
    extern __shared__ __attribute__((aligned(16))) char smem[];
    __shared__ float red_v[4], red_i[4];
    const HeadTailParams& p = q.t;
    const int b = blockIdx.x >> 1, half = blockIdx.x & 1, tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int ts = wave >> 1, h = wave & 1;
    // the sample's flag (workgroup 1: which size tower), requested first and waited for behind the cls tower's requests
    uint32_t fl_raw;
    asm volatile("s_load_dword %0, %1, 0x0" : "=s"(fl_raw) : "s"(p.flag + b) : "memory");
    const char* src = reinterpret_cast<const char*>(q.g3 + (size_t)b * HF_S * 4 * HF_CIN);

    // the input of one tower: 32 instructions of 1 KB (8 positions x 128 bytes), 8 per wave, into image rows 1 .. 16 of the slot
    auto stage = [&](int slot, int tower) __attribute__((always_inline)) {
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const int n = wave * 8 + i;
            const int pp = 8 * n + (lane >> 3);
            const int c = (lane & 7) ^ hf_swz(pp);
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(src + pp * 512 + tower * 128 + c * 16),
                                             (__attribute__((address_space(3))) void*)(smem + slot * HF_SLOT + HF_ROW + n * 1024), 16, 0, 0);
        }
    };
    stage(0, 0);                                                 // the cls tower does not wait for the flag
    asm volatile("s_waitcnt lgkmcnt(0)" : "+s"(fl_raw) : : "memory");
    const int t1 = half == 0 ? 1 : ((int)fl_raw == 1 ? 3 : 2);
    stage(1, t1);
    const int tower = ts == 0 ? 0 : t1;
    // behind it: the first weight fragments, the layer's bias and the 1x1 weights of this wave's tower, the tail's operands of this thread's position
    const u32x4* wfp = reinterpret_cast<const u32x4*>(q.wf) + (size_t)tower * 36 * 64 + lane;
    u32x4 wc[6], wn[6];
#pragma unroll
    for (int f = 0; f < 6; ++f) wc[f] = wfp[f * 64];
    const int quad = lane >> 4;
    f32x4 bias[2], w1r[2];
    const int k0 = tower == 0 ? 0 : 2 * tower - 1, k1 = tower == 0 ? 0 : 2 * tower;      // the outputs of a tower: 0 cls | 1,2 offset | 3,4 bbox | 5,6 bbox_grounding
    const int m16 = lane & 15;
#pragma unroll
    for (int nb = 0; nb < 2; ++nb) {
        bias[nb] = *reinterpret_cast<const f32x4*>(q.bias3 + tower * HF_COUT + nb * 16 + 4 * quad);
        w1r[nb] = *reinterpret_cast<const f32x4*>(p.w1 + ((m16 & 1) ? k1 : k0) * HF_COUT + nb * 16 + 4 * quad);
    }
    float cs[3];
    {
        const float* cr = p.cont + ((size_t)b * HF_S + tid) * p.cont_ch;
#pragma unroll
        for (int k = 0; k < 3; ++k) cs[k] = cr[k < p.cont_ch ? k : 0];
    }
    const float gx = p.coord[tid], gy = p.coord[HF_S + tid];
    const float b_cls = p.b1[0], b_o1 = p.b1[2 * t1 - 1], b_o2 = p.b1[2 * t1];
    // zeros: the image rows -1 and 16 of both slots and the region the lanes of a column outside the image read (the same row offsets on top of its base)
    {
        const u32x4 z = {0u, 0u, 0u, 0u};
        const int r = tid >> 7, o = (tid & 127) * 16;            // 2 KB per halo row = 128 threads x 16 bytes
        *reinterpret_cast<u32x4*>(smem + r * HF_SLOT + o) = z;
        *reinterpret_cast<u32x4*>(smem + r * HF_SLOT + 17 * HF_ROW + o) = z;
#pragma unroll
        for (int k = 0; k < 5; ++k) *reinterpret_cast<u32x4*>(smem + 2 * HF_SLOT + k * 4096 + tid * 16) = z;
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();

    f32x4 acc[8][2];
#pragma unroll
    for (int o = 0; o < 8; ++o)
#pragma unroll
        for (int nb = 0; nb < 2; ++nb) acc[o][nb] = f32x4{0.f, 0.f, 0.f, 0.f};
    const int j = lane & 15;
    // The ten row operands of step `it` (column tap it / 2, channel half it % 2): slot rows 8 h .. 8 h + 9 (= image rows 8 h - 1 .. 8 h + 8) at ONE lane address + t
    // row pitches (the swizzle depends on the column only); a lane whose column is outside the image reads zeros at the same offsets
    auto operands = [&](int it, u32x4 (&a)[10]) __attribute__((always_inline)) {
        const int dj = it >> 1, ks = it & 1;
        const int jj = j + dj - 1;
        const int m = -(int)((unsigned)jj < (unsigned)HF_F);     // (mask arithmetic: a select of the address became a branch around the reads)
        const int base = ((ts * HF_SLOT + 8 * h * HF_ROW + jj * 128 + (((ks * 4 + quad) ^ hf_swz(jj)) << 4)) & m) | (2 * HF_SLOT & ~m);
#pragma unroll
        for (int t = 0; t < 10; ++t) a[t] = *reinterpret_cast<const u32x4*>(smem + base + t * HF_ROW);
    };
    u32x4 ac[10], an[10];
    operands(0, ac);
    // One wave per SIMD: nothing else hides a latency, so the next step's operands (LDS) and weight fragments (L2) are requested before this step's 48 MFMAs and the
    // scheduler is kept from sinking them to their first use
#pragma unroll
    for (int it = 0; it < 6; ++it) {
        if (it < 5) {
#pragma unroll
            for (int f = 0; f < 6; ++f) wn[f] = wfp[((it + 1) * 6 + f) * 64];
            operands(it + 1, an);
        }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int t = 0; t < 10; ++t) {
#pragma unroll
            for (int di = 0; di < 3; ++di) {
                const int o = t - di;                            // output row 8 h + o reads input row 8 h + o + (di - 1)
                if (o < 0 || o >= 8) continue;
#pragma unroll
                for (int nb = 0; nb < 2; ++nb)
                    acc[o][nb] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, wc[di * 2 + nb]), __builtin_bit_cast(bf16x8, ac[t]), acc[o][nb], 0, 0, 0);
            }
        }
        __builtin_amdgcn_sched_barrier(0);
        if (it < 5) {
#pragma unroll
            for (int f = 0; f < 6; ++f) wc[f] = wn[f];
#pragma unroll
            for (int t = 0; t < 10; ++t) ac[t] = an[t];
        }
    }
    __syncthreads();                                             // every wave is done with the input: its LDS now takes the 1x1 partials

    // y = bf16(relu(conv + b)) (the rounding point of the two-launch path: G4 was bf16) and the tower's 1x1 conv as ONE more MFMA per image row: the accumulator layout
    // (lane = position j, channels 4 quad + e of both 16-channel halves) IS an operand layout once the K index is read as k = 8 quad + e' <-> channel 4 quad + e' (e' < 4),
    // 16 + 4 quad + e' - 4 (e' >= 4), and the 1x1 weights are laid out the same way: rows 0, 1 = bf16(w) of the tower's two outputs, rows 2, 3 = bf16(w - bf16(w)) --
    // f32-grade weights (2^-17 relative), sums over all 32 channels in one lane, no partials across lanes
    u32x4 w1op = {0u, 0u, 0u, 0u};
    if (m16 < 4) {
        uint32_t wds[4];
#pragma unroll
        for (int nb = 0; nb < 2; ++nb)
#pragma unroll
            for (int e = 0; e < 4; e += 2) {
                float a0 = w1r[nb][e], a1 = w1r[nb][e + 1];
                const float h0 = bf2f(f2bf(a0)), h1 = bf2f(f2bf(a1));
                if (m16 >= 2) { a0 -= h0; a1 -= h1; }
                wds[nb * 2 + e / 2] = pack_bf16x2(a0, a1);
            }
        w1op = u32x4{wds[0], wds[1], wds[2], wds[3]};
    }
    float* outs = reinterpret_cast<float*>(smem);               // [cls | second tower's two outputs][256 positions]
#pragma unroll
    for (int o = 0; o < 8; ++o) {
        uint32_t yd[4];
#pragma unroll
        for (int nb = 0; nb < 2; ++nb)
#pragma unroll
            for (int e = 0; e < 4; e += 2)
                yd[nb * 2 + e / 2] = pack_bf16x2(fmaxf(acc[o][nb][e] + bias[nb][e], 0.f), fmaxf(acc[o][nb][e + 1] + bias[nb][e + 1], 0.f));
        const u32x4 yv = {yd[0], yd[1], yd[2], yd[3]};
        const f32x4 d = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, w1op), __builtin_bit_cast(bf16x8, yv), f32x4{0.f, 0.f, 0.f, 0.f}, 0, 0, 0);
        if (quad == 0) {
            const int pos = (8 * h + o) * HF_F + j;
            outs[(ts == 0 ? 0 : 1) * HF_S + pos] = d[0] + d[2];
            if (ts != 0) outs[2 * HF_S + pos] = d[1] + d[3];
        }
    }
    __syncthreads();

    const float o3[3] = {outs[tid], outs[HF_S + tid], outs[2 * HF_S + tid]};
    const float cls = sigmoidf_(b_cls + o3[0]);
    float mx = cs[0];
    for (int k = 1; k < p.cont_ch; ++k) mx = fmaxf(mx, cs[k < 3 ? k : 0]);
    float den = 0.f;
    for (int k = 0; k < p.cont_ch; ++k) den += __expf(cs[k < 3 ? k : 0] - mx);
    const float p0 = __expf(cs[0] - mx) / den;
    const float score = cls * p0;
    // the sample's winner: highest score, lowest position among equals; a NaN score never wins (head_tail_kernel's rule) and no winner at all = position 0
    const float cand = score > -INFINITY ? score : -INFINITY;
    const float wmax = wave_max(cand);
    const float wpos = -wave_max(cand == wmax ? -(float)tid : -INFINITY);
    if (lane == 0) { red_v[wave] = wmax; red_i[wave] = wpos; }
    const size_t bs = (size_t)b * HF_S + tid;
    float2 bb;
    if (half == 0) {
        if (p.o_cls_test) p.o_cls_test[bs] = cls;
        if (p.o_cls) p.o_cls[bs] = p.joint_cls ? score : cls;
        const float ox = p.offset_sigmoid ? sigmoidf_(b_o1 + o3[1]) : b_o1 + o3[1];
        const float oy = p.offset_sigmoid ? sigmoidf_(b_o2 + o3[2]) : b_o2 + o3[2];
        bb.x = (gx + ox) / (float)HF_F;
        bb.y = (gy + oy) / (float)HF_F;
    } else {
        bb.x = sigmoidf_(b_o1 + o3[1]);
        bb.y = sigmoidf_(b_o2 + o3[2]);
    }
    if (p.o_bbox_map) *reinterpret_cast<float2*>(p.o_bbox_map + bs * 4 + 2 * half) = bb;
    __syncthreads();
    float bv = red_v[0], bi = red_i[0];
#pragma unroll
    for (int k = 1; k < 4; ++k)
        if (red_v[k] > bv) { bv = red_v[k]; bi = red_i[k]; }    // (waves in position order: the first wave holding the maximum has its lowest position)
    const int win = bv > -INFINITY ? (int)bi : 0;
    if (tid == win) {
        if (p.o_pred && p.o_bbox_map) *reinterpret_cast<float2*>(p.o_pred + (size_t)b * 4 + 2 * half) = bb;
        if (p.o_argmax && half == 0) p.o_argmax[b] = win;
    }
}

bool head_fin_ok(const HeadFinParams& q) {
    const HeadTailParams& p = q.t;
    return p.F == HF_F && p.S == HF_S && p.c8 == HF_COUT && q.g3_ld == 4 * HF_CIN && q.g3 && q.wf && q.bias3 && p.w1 && p.b1 && p.cont && p.coord && p.flag &&
           p.cont_ch >= 1 && p.cont_ch <= 3 && p.B > 0;
}
hipError_t launch_head_fin(const HeadFinParams& q, hipStream_t s) {
    if (!head_fin_ok(q)) return hipErrorInvalidValue;
    constexpr size_t lds = HF_LDS;
    static bool attr_done = false;
    if (!attr_done) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(head_fin_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) return e;
        attr_done = true;
    }
    g_last_kernel = "head_fin_kernel";
    hipLaunchKernelGGL(head_fin_kernel, dim3(2 * q.t.B), dim3(256), lds, s, q);
    return hipGetLastError();
}

}  // namespace uvl
