// preprocess_kernels.h — image_prepare_resnet's resize on the device.
//
// `img.resize_exact(224, 224, FilterType::CatmullRom)` (server/src/clip.rs:154) is the `image`
// crate's separable resampler (image 0.25.8, Cargo.lock:5008-5009, src/imageops/sample.rs:
// resize -> vertical_sample -> horizontal_sample, kernel bc_cubic_spline(x, 0, 0.5), support 2).
// Two kernels, the same two passes, every fp32 operation in the order the crate writes it
// (weights, their sum, the division, the multiply-adds over the window in ascending source
// index; -ffp-contract=off; IEEE division), so the bytes equal oracle/oracle.c's restatement:
//   resize_v_kernel   u8 [h][w][3]      -> f32 [nh][w][3]   one workgroup = one output row slice
//   resize_h_kernel   f32 [nh][w][3]    -> u8 [nh][nw][3]   (clamp to [0,255], round half away)
//                                        or, fused, straight to (u8/255 - mean) / std in planar CHW
// Both are HBM/L2 streaming kernels: the source is read about 4/min(1,1/ratio) times through L2
// (each output row needs 4*ratio source rows), the intermediate stays in L2 for any photo size.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace mi {

constexpr int RESIZE_MAX_TAPS = 1024;   // 4 * ratio + 2 taps: ratio (source / output extent) up to 255
constexpr uint32_t RESIZE_MAX_DIM = 32768;

__device__ __forceinline__ float catmullrom_weight(float x) {
    const float a = fabsf(x);
    float k;
    if (a < 1.0f) k = (9.0f * ((a * a) * a) + -15.0f * (a * a)) + 6.0f;
    else if (a < 2.0f) k = ((-3.0f * ((a * a) * a) + 15.0f * (a * a)) + -24.0f * a) + 12.0f;
    else k = 0.0f;
    return k / 6.0f;
}

struct ResizeWindow { int left, right; float centre, sratio; };
// source window [left, right) of output index o (n_out outputs over n_in inputs)
__device__ __forceinline__ ResizeWindow resize_window(uint32_t o, uint32_t n_in, uint32_t n_out) {
    ResizeWindow r;
    const float ratio = (float)n_in / (float)n_out;
    r.sratio = ratio < 1.0f ? 1.0f : ratio;
    const float support = 2.0f * r.sratio;
    const float in = ((float)o + 0.5f) * ratio;
    int l = (int)floorf(in - support);
    l = l < 0 ? 0 : (l > (int)n_in - 1 ? (int)n_in - 1 : l);
    int rt = (int)ceilf(in + support);
    rt = rt < l + 1 ? l + 1 : (rt > (int)n_in ? (int)n_in : rt);
    r.left = l; r.right = rt; r.centre = in - 0.5f;
    return r;
}

// normalised weights of one output index into LDS (all threads of the workgroup take part)
__device__ __forceinline__ int resize_weights(float* ws, const ResizeWindow& wd) {
    const int nt = wd.right - wd.left;
    for (int i = threadIdx.x; i < nt; i += blockDim.x)
        ws[i] = catmullrom_weight(((float)(wd.left + i) - wd.centre) / wd.sratio);
    __syncthreads();
    float sum = 0.0f;
    for (int i = 0; i < nt; ++i) sum += ws[i];  // ascending order, as the crate accumulates it
    __syncthreads();
    for (int i = threadIdx.x; i < nt; i += blockDim.x) ws[i] = ws[i] / sum;
    __syncthreads();
    return nt;
}

// grid (ceil(3w / (256 * V)), nh), 256 threads; a thread owns V consecutive bytes of the row
// (V = 4 when the row pitch 3w is a multiple of 4: one dword load per tap instead of four byte loads)
template <int V>
__global__ __launch_bounds__(256) void resize_v_kernel(const uint8_t* __restrict__ src, uint32_t w, uint32_t h,
                                                       uint32_t nh, float* __restrict__ tmp) {
    __shared__ float ws[RESIZE_MAX_TAPS];
    const uint32_t oy = blockIdx.y;
    const ResizeWindow wd = resize_window(oy, h, nh);
    const int nt = resize_weights(ws, wd);
    const size_t row = (size_t)w * 3, e = ((size_t)blockIdx.x * 256 + threadIdx.x) * V;
    if (e >= row) return;
    const uint8_t* p = src + (size_t)wd.left * row + e;
    float t[V];
#pragma unroll
    for (int v = 0; v < V; ++v) t[v] = 0.0f;
    for (int i = 0; i < nt; ++i) {
        const float wi = ws[i];
        if constexpr (V == 4) {
            const uint32_t q = *reinterpret_cast<const uint32_t*>(p + (size_t)i * row);
#pragma unroll
            for (int v = 0; v < 4; ++v) t[v] += (float)((q >> (8 * v)) & 0xffu) * wi;
        } else {
            t[0] += (float)p[(size_t)i * row] * wi;
        }
    }
    float* o = tmp + (size_t)oy * row + e;
    if constexpr (V == 4) *reinterpret_cast<float4*>(o) = make_float4(t[0], t[1], t[2], t[3]);
    else o[0] = t[0];
}

// grid (nw), 256 threads; thread y handles output rows y, y + 256, ...
template <bool TO_CHW>
__global__ __launch_bounds__(256) void resize_h_kernel(const float* __restrict__ tmp, uint32_t w, uint32_t nw,
                                                       uint32_t nh, uint8_t* __restrict__ dst_u8,
                                                       float* __restrict__ dst_chw) {
    __shared__ float ws[RESIZE_MAX_TAPS];
    const uint32_t ox = blockIdx.x;
    const ResizeWindow wd = resize_window(ox, w, nw);
    const int nt = resize_weights(ws, wd);
    const float mean[3] = {0.485f, 0.456f, 0.406f};
    const float sd[3] = {0.229f, 0.224f, 0.225f};
    for (uint32_t y = threadIdx.x; y < nh; y += 256) {
        const float* p = tmp + ((size_t)y * w + (size_t)wd.left) * 3;
        float t[3] = {0.0f, 0.0f, 0.0f};
        for (int i = 0; i < nt; ++i) {
#pragma unroll
            for (int c = 0; c < 3; ++c) t[c] += p[(size_t)i * 3 + c] * ws[i];
        }
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            float v = t[c] < 0.0f ? 0.0f : (t[c] > 255.0f ? 255.0f : t[c]);
            const uint8_t q = (uint8_t)roundf(v);
            if constexpr (TO_CHW) {
                const float f = (float)q / 255.0f;  // server/src/clip.rs:165-171
                dst_chw[((size_t)c * nh + y) * nw + ox] = (f - mean[c]) / sd[c];
            } else {
                dst_u8[((size_t)y * nw + ox) * 3 + c] = q;
            }
        }
    }
}

// same-size images are copied, not resampled (imageops::resize's early return)
template <bool TO_CHW>
__global__ void resize_copy_kernel(const uint8_t* __restrict__ src, size_t n_px, uint8_t* __restrict__ dst_u8,
                                   float* __restrict__ dst_chw) {
    const float mean[3] = {0.485f, 0.456f, 0.406f};
    const float sd[3] = {0.229f, 0.224f, 0.225f};
    for (size_t p = (size_t)blockIdx.x * blockDim.x + threadIdx.x; p < n_px; p += (size_t)gridDim.x * blockDim.x) {
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            const uint8_t q = src[p * 3 + c];
            if constexpr (TO_CHW) dst_chw[(size_t)c * n_px + p] = ((float)q / 255.0f - mean[c]) / sd[c];
            else dst_u8[p * 3 + c] = q;
        }
    }
}

// d_src [h][w][3] u8 (device) -> d_u8 [nh][nw][3] or d_chw [3][nh][nw]; d_tmp holds nh*w*3 floats
template <bool TO_CHW>
inline void resize_catmullrom_launch(const uint8_t* d_src, uint32_t w, uint32_t h, uint32_t nw, uint32_t nh,
                                     float* d_tmp, uint8_t* d_u8, float* d_chw, hipStream_t s) {
    if (w == nw && h == nh) {
        hipLaunchKernelGGL((resize_copy_kernel<TO_CHW>), dim3(256), dim3(256), 0, s, d_src, (size_t)w * h, d_u8, d_chw);
        return;
    }
    const size_t row = (size_t)w * 3;
    if (row % 4 == 0 && reinterpret_cast<uintptr_t>(d_src) % 4 == 0)
        hipLaunchKernelGGL((resize_v_kernel<4>), dim3((unsigned)((row / 4 + 255) / 256), nh), dim3(256), 0, s, d_src, w, h, nh, d_tmp);
    else
        hipLaunchKernelGGL((resize_v_kernel<1>), dim3((unsigned)((row + 255) / 256), nh), dim3(256), 0, s, d_src, w, h, nh, d_tmp);
    hipLaunchKernelGGL((resize_h_kernel<TO_CHW>), dim3(nw), dim3(256), 0, s, d_tmp, w, nw, nh, d_u8, d_chw);
}

// sizes this implementation accepts; message in *why otherwise
inline bool resize_supported(uint32_t w, uint32_t h, uint32_t nw, uint32_t nh, const char** why) {
    if (w == 0 || h == 0 || nw == 0 || nh == 0) { *why = "zero image extent"; return false; }
    if (w > RESIZE_MAX_DIM || h > RESIZE_MAX_DIM || nw > RESIZE_MAX_DIM || nh > RESIZE_MAX_DIM) { *why = "image extent above 32768"; return false; }
    const double rw = (double)w / nw, rh = (double)h / nh;
    if (4.0 * (rw < 1 ? 1 : rw) + 3 > RESIZE_MAX_TAPS || 4.0 * (rh < 1 ? 1 : rh) + 3 > RESIZE_MAX_TAPS) { *why = "reduction ratio above 255"; return false; }
    return true;
}

}  // namespace mi
