// Per-frame pre-processing on the device ("next" row 3 of SURVEY.md section 8f): the tracker's
//   sample_target          (reference lib/train/data/processing_utils.py:159-243; cv2 crop + zero border + cv2.resize)
//   Preprocessor_wo_mask   (reference lib/test/tracker/tracker_utils.py:20-29; (x/255 - mean)/std, HWC -> NCHW)
// as ONE kernel reading the uint8 frame: the padded crop is never materialised (a tap outside the kept image window
// reads 0, the BORDER_CONSTANT value), the resize follows OpenCV's 8-bit INTER_LINEAR arithmetic (11-bit fixed-point
// weights, int horizontal pass, `(((b0*(S0>>4))>>16) + ((b1*(S1>>4))>>16) + 2) >> 2` vertical pass, 2x2 averaging for an
// exact 2x decimation) so the uint8 patch is bit-identical to the CPU restatement in oracle/preprocess_oracle.py, and the
// normalised float image is written in the NCHW layout forward_test consumes.  The host uploads 1 byte per pixel
// instead of a 4-byte float crop.
#include "common.h"
#include "kernels.h"

namespace uvl {

struct Tap { int s0, s1; int w0, w1; float f; };

// OpenCV resize.cpp, linear: destination index d -> source index and 11-bit weights.  clamp_f: the horizontal pass zeroes
// the fraction at the borders, the vertical pass clips the rows instead.
__device__ __forceinline__ Tap axis_tap(int d, double scale, int src, bool clamp_f) {
    float f = (float)(((double)d + 0.5) * scale - 0.5);
    int s = (int)floorf(f);
    f -= (float)s;
    Tap t;
    if (clamp_f) {
        if (s < 0) { f = 0.f; s = 0; }
        if (s >= src - 1) { f = 0.f; s = src - 1; }
        t.s0 = s;
        t.s1 = min(s + 1, src - 1);
    } else {
        t.s0 = min(max(s, 0), src - 1);
        t.s1 = min(max(s + 1, 0), src - 1);
    }
    t.f = f;
    t.w1 = (int)rintf(f * 2048.0f);
    t.w0 = (int)rintf((1.0f - f) * 2048.0f);
    return t;
}

__global__ __launch_bounds__(256) void preprocess_kernel(const PreprocParams p) {
    const int dx = blockIdx.x * 16 + (threadIdx.x & 15), dy = blockIdx.y * 16 + (threadIdx.x >> 4);
    if (dx >= p.out || dy >= p.out) return;
    const int cs = p.crop_sz;
    // pixel (py, px) of the padded crop: inside the kept window -> the frame, else the zero border
    auto pad = [&](int py, int px) __attribute__((always_inline)) {
        return px < p.x1_pad || px >= cs - p.x2_pad || py < p.y1_pad || py >= cs - p.y2_pad;
    };
    auto px3 = [&](int py, int px, int* v) __attribute__((always_inline)) {
        if (pad(py, px)) { v[0] = v[1] = v[2] = 0; return; }
        const uint8_t* s = p.img + (size_t)(p.y1 + py - p.oy) * p.stride + (size_t)(p.x1 + px - p.ox) * 3;
        v[0] = s[0]; v[1] = s[1]; v[2] = s[2];
    };
    int o[3];
    bool att;
    if (cs == p.out) {
        px3(dy, dx, o);
        att = pad(dy, dx);
    } else if (cs == 2 * p.out) {
        int a[3], b[3], c[3], d[3];
        px3(2 * dy, 2 * dx, a); px3(2 * dy, 2 * dx + 1, b); px3(2 * dy + 1, 2 * dx, c); px3(2 * dy + 1, 2 * dx + 1, d);
#pragma unroll
        for (int k = 0; k < 3; ++k) o[k] = (a[k] + b[k] + c[k] + d[k] + 2) >> 2;
        att = pad(2 * dy, 2 * dx) || pad(2 * dy, 2 * dx + 1) || pad(2 * dy + 1, 2 * dx) || pad(2 * dy + 1, 2 * dx + 1);
    } else {
        const double scale = (double)cs / (double)p.out;
        const Tap tx = axis_tap(dx, scale, cs, true), ty = axis_tap(dy, scale, cs, false);
        int v00[3], v01[3], v10[3], v11[3];
        px3(ty.s0, tx.s0, v00); px3(ty.s0, tx.s1, v01); px3(ty.s1, tx.s0, v10); px3(ty.s1, tx.s1, v11);
#pragma unroll
        for (int k = 0; k < 3; ++k) {
            const int h0 = v00[k] * tx.w0 + v01[k] * tx.w1, h1 = v10[k] * tx.w0 + v11[k] * tx.w1;
            const int r = (((ty.w0 * (h0 >> 4)) >> 16) + ((ty.w1 * (h1 >> 4)) >> 16) + 2) >> 2;
            o[k] = min(max(r, 0), 255);
        }
        // cv2.resize(att_mask).astype(bool): any tap with a non-zero weight lies in the border
        const bool fx = tx.f != 0.f, fy = ty.f != 0.f;
        att = pad(ty.s0, tx.s0) || (fx && pad(ty.s0, tx.s1)) || (fy && (pad(ty.s1, tx.s0) || (fx && pad(ty.s1, tx.s1))));
    }
    const size_t at = (size_t)dy * p.out + dx;
    if (p.patch) { p.patch[at * 3] = (uint8_t)o[0]; p.patch[at * 3 + 1] = (uint8_t)o[1]; p.patch[at * 3 + 2] = (uint8_t)o[2]; }
    if (p.att) p.att[at] = att ? 1 : 0;
    if (p.norm) {
        const float mean[3] = {0.485f, 0.456f, 0.406f}, stdv[3] = {0.229f, 0.224f, 0.225f};
        const size_t plane = (size_t)p.out * p.out;
#pragma unroll
        for (int k = 0; k < 3; ++k) p.norm[k * plane + at] = (((float)o[k] / 255.0f) - mean[k]) / stdv[k];
    }
}

hipError_t launch_preprocess(const PreprocParams& p, hipStream_t s) {
    if (p.out <= 0 || p.crop_sz <= 0) return hipErrorInvalidValue;
    hipLaunchKernelGGL(preprocess_kernel, dim3((p.out + 15) / 16, (p.out + 15) / 16), dim3(256), 0, s, p);
    return hipGetLastError();
}

// grounding_resize (processing_utils.py:60-141): the whole frame resized to new_w x new_h (aspect kept, OpenCV 8-bit
// INTER_LINEAR, separate x / y scales), centred in an out x out canvas of zeros; attention mask = 1 on the padding.
__global__ __launch_bounds__(256) void grounding_resize_kernel(const GroundingParams p) {
    const int dx = blockIdx.x * 16 + (threadIdx.x & 15), dy = blockIdx.y * 16 + (threadIdx.x >> 4);
    if (dx >= p.out || dy >= p.out) return;
    const int rx = dx - p.x1_pad, ry = dy - p.y1_pad;
    const bool inside = rx >= 0 && rx < p.new_w && ry >= 0 && ry < p.new_h;
    int o[3] = {0, 0, 0};
    if (inside) {
        auto px3 = [&](int y, int x, int* v) __attribute__((always_inline)) {
            const uint8_t* s = p.img + (size_t)y * p.stride + (size_t)x * 3;
            v[0] = s[0]; v[1] = s[1]; v[2] = s[2];
        };
        if (p.W == p.new_w && p.H == p.new_h) {
            px3(ry, rx, o);
        } else if (p.W == 2 * p.new_w && p.H == 2 * p.new_h) {
            int a[3], b[3], c[3], d[3];
            px3(2 * ry, 2 * rx, a); px3(2 * ry, 2 * rx + 1, b); px3(2 * ry + 1, 2 * rx, c); px3(2 * ry + 1, 2 * rx + 1, d);
#pragma unroll
            for (int k = 0; k < 3; ++k) o[k] = (a[k] + b[k] + c[k] + d[k] + 2) >> 2;
        } else {
            const Tap tx = axis_tap(rx, (double)p.W / (double)p.new_w, p.W, true), ty = axis_tap(ry, (double)p.H / (double)p.new_h, p.H, false);
            int v00[3], v01[3], v10[3], v11[3];
            px3(ty.s0, tx.s0, v00); px3(ty.s0, tx.s1, v01); px3(ty.s1, tx.s0, v10); px3(ty.s1, tx.s1, v11);
#pragma unroll
            for (int k = 0; k < 3; ++k) {
                const int h0 = v00[k] * tx.w0 + v01[k] * tx.w1, h1 = v10[k] * tx.w0 + v11[k] * tx.w1;
                const int r = (((ty.w0 * (h0 >> 4)) >> 16) + ((ty.w1 * (h1 >> 4)) >> 16) + 2) >> 2;
                o[k] = min(max(r, 0), 255);
            }
        }
    }
    const size_t at = (size_t)dy * p.out + dx;
    if (p.patch) { p.patch[at * 3] = (uint8_t)o[0]; p.patch[at * 3 + 1] = (uint8_t)o[1]; p.patch[at * 3 + 2] = (uint8_t)o[2]; }
    if (p.att) p.att[at] = inside ? 0 : 1;
    if (p.norm) {
        const float mean[3] = {0.485f, 0.456f, 0.406f}, stdv[3] = {0.229f, 0.224f, 0.225f};
        const size_t plane = (size_t)p.out * p.out;
#pragma unroll
        for (int k = 0; k < 3; ++k) p.norm[k * plane + at] = (((float)o[k] / 255.0f) - mean[k]) / stdv[k];
    }
}

hipError_t launch_grounding_resize(const GroundingParams& p, hipStream_t s) {
    if (p.out <= 0 || p.new_w <= 0 || p.new_h <= 0) return hipErrorInvalidValue;
    hipLaunchKernelGGL(grounding_resize_kernel, dim3((p.out + 15) / 16, (p.out + 15) / 16), dim3(256), 0, s, p);
    return hipGetLastError();
}

// Preprocessor_wo_mask.process on an already resized patch: uint8 HWC -> float32 CHW, ((x/255) - mean) / std
__global__ __launch_bounds__(256) void normalize_u8_kernel(const uint8_t* __restrict__ src, float* __restrict__ dst, int n_pix) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= n_pix) return;
    const float mean[3] = {0.485f, 0.456f, 0.406f}, stdv[3] = {0.229f, 0.224f, 0.225f};
#pragma unroll
    for (int k = 0; k < 3; ++k) dst[(size_t)k * n_pix + i] = (((float)src[(size_t)i * 3 + k] / 255.0f) - mean[k]) / stdv[k];
}

hipError_t launch_normalize_u8(const uint8_t* src, float* dst, int n_pix, hipStream_t s) {
    hipLaunchKernelGGL(normalize_u8_kernel, dim3((n_pix + 255) / 256), dim3(256), 0, s, src, dst, n_pix);
    return hipGetLastError();
}

}  // namespace uvl
