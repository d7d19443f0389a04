// LayerNorm row body shared by the row-op kernels (rowops.hip) and the fused LayerNorm + GEMM kernel of one-sequence frames (gemm.hip).
// (block.py:30-31 with eps 1e-6; BertLayerNorm bert_backbone.py:231-244 with eps 1e-12.)
#pragma once
#include "common.h"
#include "kernels.h"

namespace uvl {

#define LN_MAX_SLABS 4

// 4 KB of zeros: the source of loads whose value is not wanted (absent slab / absent pre-add vector), so that every load of a
// row is issued unconditionally up front -- a load under a run-time condition is compiled as branch + load + s_waitcnt vmcnt(0),
// which turned the row into nine dependent memory round trips (7 us per launch for 1.7 MB of rows).
static __device__ float g_zero_row[1024];   // one copy per translation unit that includes this header

__device__ __forceinline__ float4 sel4(bool c, const float4& a) { return c ? a : make_float4(0.f, 0.f, 0.f, 0.f); }

// FULL: D == NV * 256 (the real models): no column guard at all.  The row index is made wave-uniform explicitly, so the row's
// addresses, the slab count and the pre-add selection live in scalar registers.
// SLABS / CT: the launch has split-K slabs to fold / a contrastive job riding on it (host-known); without them their operand
// slots cost neither registers nor load issue (batched frames: no slabs, LayerNorm is HBM-bound there).
// CT: 0 = no second job, 1 = contrastive job from a snapshot (operands requested with the row), 2 = from the launch's own input rows (ct_self:
// the job's operands are requested AFTER the LayerNorm outputs are stored -- L2-hot rows -- so the load phase holds 48 registers less.
// Measured in the UVLTrack-L x 8 frame (7304 rows): plain 8.9-9.5 us, with the self job 14.6 us of which 2.9 are its four L2-hot rows and 2.2 its five reductions)
template <int NV, bool FULL, bool SLABS, int CT>
__device__ __forceinline__ void ln_body(const LnParams& p, int bx) {
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int m = bx * (int)(blockDim.x >> 6) + wave;     // one row per wave, blockDim.x / 64 rows per workgroup
    if (m >= p.M) return;
    const int b = (int)fd_div((uint32_t)m, p.fd_rpb), t = m - b * p.rpb;      // (the reciprocal of rpb comes from the launcher: no division in front of the row's loads)
    const size_t xrow = (size_t)b * p.xbs + p.xro + t;
    float* xr = const_cast<float*>(p.x) + xrow * p.D;
    const float* xin = (p.x_alt && t >= p.split) ? p.x_alt + ((size_t)b * p.x_alt_rows + (t - p.split)) * p.D : xr;
    const float* padd = (t < p.split) ? p.pre_add0 : p.pre_add1;
    const int nsp = (SLABS && t < p.part_rows) ? p.nsplit : 0;       // rows beyond part_rows were not produced by that GEMM
    const size_t pm = (size_t)b * p.part_rows + t;           // compact row index inside a slab
    // ---- phase 1: every load of the row, no use in between ----
    constexpr int NSL = SLABS ? LN_MAX_SLABS : 0;
    const float* slab[LN_MAX_SLABS];
#pragma unroll
    for (int sp = 0; sp < NSL; ++sp)
        slab[sp] = (sp < nsp) ? p.part + (size_t)sp * p.part_stride + pm * p.D : g_zero_row;
    const float* pa = padd ? padd : g_zero_row;
    float4 v[NV], sl[NV][LN_MAX_SLABS], ad[NV], g[NV], be[NV];
    bool ok[NV];
#pragma unroll
    for (int i = 0; i < NV; ++i) {
        const int c0 = (lane + 64 * i) * 4;
        ok[i] = FULL || c0 < p.D;
        const int c = ok[i] ? c0 : 0;
        v[i] = *reinterpret_cast<const float4*>(xin + c);
#pragma unroll
        for (int sp = 0; sp < NSL; ++sp) sl[i][sp] = *reinterpret_cast<const float4*>(slab[sp] + c);
        ad[i] = *reinterpret_cast<const float4*>(pa + c);
        g[i] = *reinterpret_cast<const float4*>(p.gamma + c);
        be[i] = *reinterpret_cast<const float4*>(p.beta + c);
    }
    // the second job's operands (contrastive logits of the previous layer, see below) are independent of the row: same round trip
    const bool do_ct = CT && p.ct_x && t >= 1 + p.ct_nz && t < p.ct_nv;
    float4 ca[NV], cv[NV], cq[NV];
    float ct_ls = 0.f;
    int ct_fl = 0;
    if (CT == 1 && do_ct) {
        const float* xb = p.ct_x + (size_t)b * p.xbs * p.D;
        const float* xs = xb + (size_t)t * p.D;
        const float* tk = p.ct_skip_text ? xb : (p.ct_txt ? p.ct_txt + (size_t)b * p.ct_T * p.D : xb + (size_t)p.ct_nv * p.D);
#pragma unroll
        for (int i = 0; i < NV; ++i) {
            const int c = ok[i] ? (lane + 64 * i) * 4 : 0;
            ca[i] = *reinterpret_cast<const float4*>(xs + c);
            cv[i] = *reinterpret_cast<const float4*>(xb + c);
            cq[i] = *reinterpret_cast<const float4*>(tk + c);
        }
        ct_ls = p.ct_logit_scale[0];
        ct_fl = (int)p.ct_flag[b];
    }
    __builtin_amdgcn_sched_barrier(0);       // keep the scheduler from pulling the first adds up between the loads (it would
                                             // wait for the first few loads, reuse their registers and issue the rest afterwards)
    // ---- phase 2: fold the slabs in slab order, snapshot, pre-add, write back ----
    float sum = 0.f;
#pragma unroll
    for (int i = 0; i < NV; ++i) {
        const int c = (lane + 64 * i) * 4;
        v[i] = sel4(ok[i], v[i]);
#pragma unroll
        for (int sp = 0; sp < NSL; ++sp) {
            const float4 a = sl[i][sp];
            const bool on = ok[i] && sp < nsp;
            v[i].x = on ? v[i].x + a.x : v[i].x; v[i].y = on ? v[i].y + a.y : v[i].y;
            v[i].z = on ? v[i].z + a.z : v[i].z; v[i].w = on ? v[i].w + a.w : v[i].w;
        }
        if (p.x_snap && ok[i]) *reinterpret_cast<float4*>(p.x_snap + xrow * p.D + c) = v[i];
        {
            const bool on = ok[i] && padd != nullptr;
            const float4 a = ad[i];
            v[i].x = on ? v[i].x + a.x : v[i].x; v[i].y = on ? v[i].y + a.y : v[i].y;
            v[i].z = on ? v[i].z + a.z : v[i].z; v[i].w = on ? v[i].w + a.w : v[i].w;
        }
        if ((padd || nsp > 0 || xin != xr) && ok[i]) *reinterpret_cast<float4*>(xr + c) = v[i];
        sum += v[i].x + v[i].y + v[i].z + v[i].w;
    }
    const float mean = wave_sum(sum) / (float)p.D;
    float sq = 0.f;
#pragma unroll
    for (int i = 0; i < NV; ++i) {
        if (ok[i]) {
            const float dx = v[i].x - mean, dy = v[i].y - mean, dz = v[i].z - mean, dw = v[i].w - mean;
            sq += dx * dx + dy * dy + dz * dz + dw * dw;
        }
    }
    const float rstd = 1.0f / sqrtf(wave_sum(sq) / (float)p.D + p.eps);
    const size_t yrow = p.y_remap ? xrow : (size_t)m;
#pragma unroll
    for (int i = 0; i < NV; ++i) {
        const int c = (lane + 64 * i) * 4;
        if (ok[i]) {
            float4 y;
            y.x = (v[i].x - mean) * rstd * g[i].x + be[i].x;
            y.y = (v[i].y - mean) * rstd * g[i].y + be[i].y;
            y.z = (v[i].z - mean) * rstd * g[i].z + be[i].z;
            y.w = (v[i].w - mean) * rstd * g[i].w + be[i].w;
            if (p.y_f32) *reinterpret_cast<float4*>(p.y_f32 + yrow * p.D + c) = y;
            if (p.y_copy) *reinterpret_cast<float4*>(p.y_copy + (size_t)m * p.D + c) = y;
            if (p.y_bf16) {
                uint2 w;
                w.x = pack_bf16x2(y.x, y.y);
                w.y = pack_bf16x2(y.z, y.w);
                // y_wt: write-through (the fused LayerNorm + GEMM kernel reads these rows from other XCDs behind a grid barrier)
                if (p.y_wt) asm volatile("global_store_dwordx2 %0, %1, off sc0 sc1" ::"v"(p.y_bf16 + (size_t)m * p.D + c), "v"(w) : "memory");
                else *reinterpret_cast<uint2*>(p.y_bf16 + (size_t)m * p.D + c) = w;
            }
        }
    }
    // ---- contrastive logits of the previous layer for this wave's search row (same arithmetic, in the same order, as
    //      contrast_kernel: tau * normalize(x) . normalize(token), select [vis, txt, mean][flag]) ----
    if (do_ct) {
        const int s = t - 1 - p.ct_nz;
        float xx = 0.f, xv = 0.f, vv = 0.f, xt = 0.f, tt = 0.f;
        if (CT == 2) {
            // the search row is this wave's own row (v[]: no slabs, no pre-add touched it); the vis / text token rows of the sample are requested
            // now; the NEXT layer's modal embedding, which the fc2 epilogue has already added to every row, comes off again
            const float* xb = p.ct_x + (size_t)b * p.xbs * p.D;
            const float* tk = p.ct_skip_text ? xb : (p.ct_txt ? p.ct_txt + (size_t)b * p.ct_T * p.D : xb + (size_t)p.ct_nv * p.D);
            const float* sv = p.ct_sub_vis ? p.ct_sub_vis : g_zero_row;
            const float* sq = (p.ct_sub_txt && !p.ct_txt) ? p.ct_sub_txt : g_zero_row;
            ct_ls = p.ct_logit_scale[0];
            ct_fl = (int)p.ct_flag[b];
#pragma unroll
            for (int i = 0; i < NV; ++i) {
                const int c = ok[i] ? (lane + 64 * i) * 4 : 0;
                cv[i] = *reinterpret_cast<const float4*>(xb + c);
                cq[i] = *reinterpret_cast<const float4*>(tk + c);
                const float4 m0 = *reinterpret_cast<const float4*>(sv + c), m1 = *reinterpret_cast<const float4*>(sq + c);
                ca[i].x = v[i].x - m0.x; ca[i].y = v[i].y - m0.y; ca[i].z = v[i].z - m0.z; ca[i].w = v[i].w - m0.w;
                cv[i].x -= m0.x; cv[i].y -= m0.y; cv[i].z -= m0.z; cv[i].w -= m0.w;
                cq[i].x -= m1.x; cq[i].y -= m1.y; cq[i].z -= m1.z; cq[i].w -= m1.w;
            }
        }
#pragma unroll
        for (int i = 0; i < NV; ++i) {
            if (ok[i]) {
                const float4 a = ca[i], vq = cv[i];
                xx += a.x * a.x + a.y * a.y + a.z * a.z + a.w * a.w;
                xv += a.x * vq.x + a.y * vq.y + a.z * vq.z + a.w * vq.w;
                vv += vq.x * vq.x + vq.y * vq.y + vq.z * vq.z + vq.w * vq.w;
                if (!p.ct_skip_text) {
                    const float4 q = cq[i];
                    xt += a.x * q.x + a.y * q.y + a.z * q.z + a.w * q.w;
                    tt += q.x * q.x + q.y * q.y + q.z * q.z + q.w * q.w;
                }
            }
        }
        const float tau = __expf(ct_ls);
        xx = fmaxf(sqrtf(wave_sum(xx)), 1e-12f);
        vv = fmaxf(sqrtf(wave_sum(vv)), 1e-12f);
        const float lv = tau * wave_sum(xv) / (xx * vv);
        float lt = 0.f;
        if (!p.ct_skip_text) {
            tt = fmaxf(sqrtf(wave_sum(tt)), 1e-12f);
            lt = tau * wave_sum(xt) / (xx * tt);
        }
        const float out = ct_fl == 0 ? lv : (ct_fl == 1 ? lt : 0.5f * (lv + lt));
        if (lane == 0) p.ct_logits[((size_t)b * p.ct_ncont + p.ct_slot) * p.ct_nx + s] = out;
    }
}


}  // namespace uvl
