// vit_kernels.h — device code of the CLIP vision tower (Seam A).
//
// Replaces `clip::clip_vit_large_patch14::Model::forward` (call site
// server/src/clip.rs:118; graph = Xenova/clip-vit-large-patch14 vision_model.onnx,
// clip/build.rs:10-11): conv patch-embed -> +CLS -> +pos -> pre-LN ->
// L x [LN -> MHA -> +res -> LN -> fc1 -> QuickGELU -> fc2 -> +res] -> CLS ->
// post-LN -> projection.  Bound: MFMA (96 % of the 162 GFLOP/image are the six
// linear GEMMs per layer).
//
// Data layout in HBM (M = n*S token rows, padded to a multiple of 256):
//   x    [M][D]   f32   residual stream (fp32 in both precisions)
//   y    [M][D]   T     LayerNorm output / attention context
//   qkv  [M][3D]  T     fused q|k|v projection (head h = columns h*64..h*64+63)
//   hbuf [M][FF]  T     fc1 output after QuickGELU
//   weights [N][K] T    exactly PyTorch's [out,in] = K-contiguous, so both MFMA
//                       operands are read K-major with no transposition
//   T = f32 (MI_PRECISION_F32, exact-f32 MFMA 32x32x2) or bf16 (MFMA 16x16x32).
//
// All GEMMs compute C^T tiles: the MFMA A operand is the WEIGHT tile (rows n) and
// the B operand the ACTIVATION tile (rows m), so a lane ends up with 4 consecutive
// output columns n of one token row m — an 8-byte (bf16) or 16-byte (f32)
// contiguous store, and a float4 bias / residual access.
#pragma once
#include <hip/hip_runtime.h>

#include <cstdint>

namespace mi {

typedef unsigned short bf16_t;
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
typedef short s16x4 __attribute__((ext_vector_type(4)));
typedef float v4f __attribute__((ext_vector_type(4)));
typedef float v16f __attribute__((ext_vector_type(16)));
typedef unsigned int v4u __attribute__((ext_vector_type(4)));
typedef unsigned int v2u __attribute__((ext_vector_type(2)));

enum { EPI_STORE_F32 = 0, EPI_BIAS = 1, EPI_BIAS_QGELU = 2, EPI_BIAS_RESID = 3,
       // the tower WITHOUT LayerNorm kernels (option "ln_fold", persistent bf16 GEMM only):
       EPI_LNF = 4,        // out = rstd[m] * acc + (-mean[m] * rstd[m]) * c[n] + b'[n]: the LayerNorm in front of this linear, folded
       EPI_LNF_QGELU = 5,  // ... followed by QuickGELU (fc1)
       EPI_RESID24 = 6 };  // x[m][n] += bf16(acc + bias) on the 24-bit residual planes in place, + per-row partial sums of x and x^2

// extra operands of the EPI_LNF* / EPI_RESID24 epilogues of gemm_bf16_pp_kernel
struct PpFold {
    const float* cvec = nullptr;   // EPI_LNF*: c[n] = sum_k float(W'[n][k]), W' = bf16(W * diag(gamma))
    const float* stats = nullptr;  // EPI_LNF*: [M][2] = {rstd, -mean * rstd} of the rows of X (ln_stats_kernel / embed_ln_kernel)
    uint8_t* xlo = nullptr;        // EPI_RESID24: lo plane [M][ldo] of the residual stream (its hi plane is `out`)
    float* part = nullptr;         // EPI_RESID24: [M][N / 32][2] = {sum, sum of squares} of the new x over each 32-column block
    // Elements between the starts of consecutive 64-column groups of one row of `out`.  64: token rows [M][ldo].
    // M * 64 with ldo = 64: head-major planes [N / 64][M][64] — a wave's 64 columns are one plane and the 8 rows x 128 bytes of
    // one of its stores are ONE contiguous KiB; the layout attn32_bf16_kernel streams (option "qkv_layout"; not with EPI_RESID24)
    uint32_t col_stride = 64;
};

__device__ __forceinline__ bf16_t f2bf(float f) { return __builtin_bit_cast(bf16_t, (__bf16)f); }
__device__ __forceinline__ float bf2f(bf16_t v) { return __uint_as_float((uint32_t)v << 16); }
// two floats -> one dword of bf16 (a in the low half): ONE v_cvt_pk_bf16_f32 (the element-wise form cost
// two converts + and + or per pair in every epilogue)
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
typedef float v2f __attribute__((ext_vector_type(2)));
__device__ __forceinline__ uint32_t pack2bf(float a, float b) {
    return __builtin_bit_cast(uint32_t, __builtin_convertvector((v2f){a, b}, bf16x2));
}

__device__ __forceinline__ float wave_sum(float v) {
    v += __shfl_xor(v, 1, 64); v += __shfl_xor(v, 2, 64); v += __shfl_xor(v, 4, 64);
    v += __shfl_xor(v, 8, 64); v += __shfl_xor(v, 16, 64); v += __shfl_xor(v, 32, 64);
    return v;
}

// x * sigmoid(1.702 x)  (transformers/activations.py:117-123)
// FAST (bf16 path): v_exp_f32 + v_rcp_f32 (1 ulp each, far below bf16 resolution) instead of the
// correctly rounded expf and IEEE division of the fp32 parity path.
template <bool FAST>
__device__ __forceinline__ float quick_gelu(float x) {
    if constexpr (FAST) {
        const float e = __builtin_amdgcn_exp2f(x * (-1.702f * 1.4426950408889634f));
        return x * __builtin_amdgcn_rcpf(1.0f + e);
    } else {
        return x / (1.0f + expf(-1.702f * x));
    }
}

// ------------------------------------------------------------------ preprocessing
// image_prepare_resnet's arithmetic (server/src/clip.rs:158-172) on the device:
// rgb8 [n][H][W][3] -> chw f32 [n][3][H][W]; IEEE division, same bits as the host.
__global__ void preprocess_rgb8_kernel(const uint8_t* __restrict__ rgb, float* __restrict__ chw, size_t n_px,
                                       size_t plane) {
    const float mean[3] = {0.485f, 0.456f, 0.406f};
    const float sd[3] = {0.229f, 0.224f, 0.225f};
    for (size_t p = (size_t)blockIdx.x * blockDim.x + threadIdx.x; p < n_px; p += (size_t)gridDim.x * blockDim.x) {
        const size_t im = p / plane, i = p % plane;
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            const float v = (float)rgb[p * 3 + c] / 255.0f;
            chw[(im * 3 + c) * plane + i] = (v - mean[c]) / sd[c];
        }
    }
}

// ------------------------------------------------------------------ patch gather
// conv(stride = kernel = P, no bias) as a GEMM: col[b*G*G + gy*G + gx][k],
// k = c*P*P + py*P + px (the flatten order of weight [D,3,P,P]); k >= 3*P*P is zero.
template <typename T>
__global__ void im2col_kernel(const float* __restrict__ img, T* __restrict__ col, int n, int G, int P, int HW,
                              int Kp) {
    // one thread = one patch-row segment: P contiguous pixels of (image b, channel c, row gy*P+py) ->
    // P contiguous columns k = c*P*P + py*P .. of col row (b, gy, gx).  Columns >= 3*P*P are never
    // written: the workspace is zero-filled at allocation and stays so.
    const size_t total = (size_t)n * G * G * 3 * P;
    for (size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total;
         idx += (size_t)gridDim.x * blockDim.x) {
        const int gx = (int)(idx % G);  // fastest: neighbouring threads read neighbouring 4P-byte runs of one image row
        size_t r = idx / G;
        const int py = (int)(r % P); r /= P;
        const int c = (int)(r % 3); r /= 3;
        const int gy = (int)(r % G);
        const size_t b = r / G;
        const float* src = img + ((b * 3 + c) * HW + (size_t)(gy * P + py)) * HW + (size_t)gx * P;
        T* dst = col + ((b * G + gy) * G + gx) * (size_t)Kp + (size_t)(c * P + py) * P;
        if ((P & 1) == 0) {
            for (int px = 0; px < P; px += 2) {
                const v2f v = *reinterpret_cast<const v2f*>(src + px);
                if constexpr (sizeof(T) == 4) { dst[px] = v.x; dst[px + 1] = v.y; }
                else *reinterpret_cast<uint32_t*>(dst + px) = pack2bf(v.x, v.y);
            }
        } else {
            for (int px = 0; px < P; ++px) {
                if constexpr (sizeof(T) == 4) dst[px] = src[px]; else dst[px] = f2bf(src[px]);
            }
        }
    }
}

// The same gather for the bf16 tower, through LDS: one workgroup per (image, patch row).  The 3 P image rows of that
// patch row are read as whole 4 HW-byte rows (float4, coalesced) into LDS; every patch's 3 P P columns leave as ONE
// contiguous run of the col row (1 176 bytes at P = 14), 8 bytes per thread.  im2col_kernel moves 4 P-byte runs on both
// sides and reaches 1.8 TB/s (130 us per 256 images); this one is bound by its 238 MB of traffic.
// Needs HW % 4 == 0, (3 P P) % 4 == 0, Kp % 4 == 0; LDS = 3 P HW floats.
__global__ __launch_bounds__(256) void im2col_rows_kernel(const float* __restrict__ img, bf16_t* __restrict__ col, int G, int P, int HW,
                                                          int Kp) {
    extern __shared__ __attribute__((aligned(16))) float im2col_lds[];
    const int b = blockIdx.x / G, gy = blockIdx.x - b * G;
    const int rows = 3 * P, q4 = HW >> 2;
    for (int i = threadIdx.x; i < rows * q4; i += 256) {
        const int rw = i / q4, x4 = i - rw * q4;  // rw = c * P + py
        const int c = rw / P, py = rw - c * P;
        const v4f v = *reinterpret_cast<const v4f*>(img + (((size_t)b * 3 + c) * HW + (size_t)(gy * P + py)) * HW + 4 * x4);
        *reinterpret_cast<v4f*>(im2col_lds + (size_t)rw * HW + 4 * x4) = v;
    }
    __syncthreads();
    const int per = 3 * P * P, g4 = per >> 2;
    for (int i = threadIdx.x; i < G * g4; i += 256) {
        const int gx = i / g4, j = i - gx * g4;
        float v[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const int k = 4 * j + e, rw = k / P, px = k - rw * P;  // rw = c * P + py again: k = (c P + py) P + px
            v[e] = im2col_lds[(size_t)rw * HW + gx * P + px];
        }
        bf16_t* dst = col + (((size_t)b * G + gy) * G + gx) * (size_t)Kp + 4 * j;
        *reinterpret_cast<v2u*>(dst) = (v2u){pack2bf(v[0], v[1]), pack2bf(v[2], v[3])};
    }
}

// ------------------------------------------------------------------ LayerNorm
// One wave per row, row held in registers, two-pass mean / biased variance,
// y = d / sqrt(var + eps) * w + b  (the decomposed opset-16 form the reference
// graph keeps: clip/scripts/upgrade_opset.py:9,23).  D = 64*VEC*NT.
typedef float v2f __attribute__((ext_vector_type(2)));

template <int VEC>
__device__ __forceinline__ void ld_vec(float* dst, const float* __restrict__ p) {
    if constexpr (VEC == 4) {
        const v4f t = *reinterpret_cast<const v4f*>(p);
        dst[0] = t.x; dst[1] = t.y; dst[2] = t.z; dst[3] = t.w;
    } else if constexpr (VEC == 2) {
        const v2f t = *reinterpret_cast<const v2f*>(p);
        dst[0] = t.x; dst[1] = t.y;
    } else {
        dst[0] = *p;
    }
}
template <int VEC, typename T>
__device__ __forceinline__ void st_vec(T* __restrict__ p, const float* src) {
    if constexpr (sizeof(T) == 4) {
        if constexpr (VEC == 4) *reinterpret_cast<v4f*>(p) = (v4f){src[0], src[1], src[2], src[3]};
        else if constexpr (VEC == 2) *reinterpret_cast<v2f*>(p) = (v2f){src[0], src[1]};
        else *p = src[0];
    } else {
        if constexpr (VEC == 4) *reinterpret_cast<v2u*>(p) = (v2u){pack2bf(src[0], src[1]), pack2bf(src[2], src[3])};
        else if constexpr (VEC == 2) *reinterpret_cast<uint32_t*>(p) = pack2bf(src[0], src[1]);
        else *p = f2bf(src[0]);
    }
}

template <int VEC, int NT>
struct LnRow {
    float v[VEC * NT];
    __device__ __forceinline__ void load(const float* __restrict__ p, int lane) {
#pragma unroll
        for (int t = 0; t < NT; ++t) ld_vec<VEC>(&v[t * VEC], p + (t * 64 + lane) * VEC);
    }
    // The residual stream as 24-BIT floats in two planes (option "x24", bf16 tower): hi = the top 16 bits of the fp32
    // pattern (sign, exponent, 7 mantissa bits), lo = the next 8 mantissa bits — 16 significant bits, 3 bytes per element
    // instead of 4.  The LayerNorms are pure HBM traffic and cost the forward their full stand-alone time:
    // the fp32 residual is 12 of their 22 bytes per element and layer, this makes it 9.  Rounded to nearest when stored.
    // bias: 0 for the truncated planes written by store_x24; X24B_BIAS for the planes of the LayerNorm-free tower, whose hi
    // plane is bf16(x) rounded to nearest (resid24_step, store_x24 with round = 0x8080)
    __device__ __forceinline__ void load_x24(const uint16_t* __restrict__ hi, const uint8_t* __restrict__ lo, int lane, uint32_t bias = 0) {
        static_assert(VEC == 4, "four elements per lane and chunk");
#pragma unroll
        for (int t = 0; t < NT; ++t) {
            const v2u h = *reinterpret_cast<const v2u*>(hi + (t * 64 + lane) * 4);
            const uint32_t l = *reinterpret_cast<const uint32_t*>(lo + (t * 64 + lane) * 4);
            v[t * 4 + 0] = __uint_as_float(((h.x << 16) | ((l & 0xFFu) << 8)) - bias);
            v[t * 4 + 1] = __uint_as_float(((h.x & 0xFFFF0000u) | (l & 0xFF00u)) - bias);
            v[t * 4 + 2] = __uint_as_float(((h.y << 16) | ((l >> 8) & 0xFF00u)) - bias);
            v[t * 4 + 3] = __uint_as_float(((h.y & 0xFFFF0000u) | ((l >> 16) & 0xFF00u)) - bias);
        }
    }
    // round: 0x80 = to 24 bits, nearest (ties up); 0x8080 = the same plus half a bf16 ulp (load_x24 with bias X24B_BIAS).
    // Inf and NaN patterns are stored as they are (an increment could carry a NaN's payload into the sign bit).
    __device__ __forceinline__ void store_x24(uint16_t* __restrict__ hi, uint8_t* __restrict__ lo, int lane, uint32_t round = 0x80u) const {
        static_assert(VEC == 4, "four elements per lane and chunk");
#pragma unroll
        for (int t = 0; t < NT; ++t) {
            uint32_t r[4];
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                const uint32_t u = __float_as_uint(v[t * 4 + c]);
                r[c] = (u & 0x7F800000u) == 0x7F800000u ? u + (round & 0x8000u) : u + round;
            }
            v2u h;
            h.x = (r[0] >> 16) | (r[1] & 0xFFFF0000u);
            h.y = (r[2] >> 16) | (r[3] & 0xFFFF0000u);
            *reinterpret_cast<v2u*>(hi + (t * 64 + lane) * 4) = h;
            *reinterpret_cast<uint32_t*>(lo + (t * 64 + lane) * 4) =
                ((r[0] >> 8) & 0xFFu) | (r[1] & 0xFF00u) | ((r[2] << 8) & 0xFF0000u) | ((r[3] << 16) & 0xFF000000u);
        }
    }
    // last-use loads (LN1: the residual stream and the two deltas are not read again): non-temporal hint
    __device__ __forceinline__ void load_nt(const float* __restrict__ p, int lane) {
#pragma unroll
        for (int t = 0; t < NT; ++t) {
            if constexpr (VEC == 4) {
                const v4f u = __builtin_nontemporal_load(reinterpret_cast<const v4f*>(p + (t * 64 + lane) * VEC));
                v[t * 4 + 0] = u.x; v[t * 4 + 1] = u.y; v[t * 4 + 2] = u.z; v[t * 4 + 3] = u.w;
            } else {
                ld_vec<VEC>(&v[t * VEC], p + (t * 64 + lane) * VEC);
            }
        }
    }
    __device__ __forceinline__ void add_bf16_nt(const bf16_t* __restrict__ p, int lane) {
#pragma unroll
        for (int t = 0; t < NT; ++t) {
            const bf16_t* q = p + (t * 64 + lane) * VEC;
            if constexpr (VEC == 4) {
                const v2u u = __builtin_nontemporal_load(reinterpret_cast<const v2u*>(q));
                v[t * 4 + 0] += __uint_as_float(u.x << 16); v[t * 4 + 1] += __uint_as_float(u.x & 0xffff0000u);
                v[t * 4 + 2] += __uint_as_float(u.y << 16); v[t * 4 + 3] += __uint_as_float(u.y & 0xffff0000u);
            } else if constexpr (VEC == 2) {
                const uint32_t u = *reinterpret_cast<const uint32_t*>(q);
                v[t * 2 + 0] += __uint_as_float(u << 16); v[t * 2 + 1] += __uint_as_float(u & 0xffff0000u);
            } else {
                v[t] += bf2f(*q);
            }
        }
    }
    __device__ __forceinline__ void add(const float* __restrict__ p, int lane) {
#pragma unroll
        for (int t = 0; t < NT; ++t) {
            float u[VEC];
            ld_vec<VEC>(u, p + (t * 64 + lane) * VEC);
#pragma unroll
            for (int c = 0; c < VEC; ++c) v[t * VEC + c] += u[c];
        }
    }
    // v += bf16 row (the deferred residual of the previous GEMM, see vit.hip forward())
    __device__ __forceinline__ void add_bf16(const bf16_t* __restrict__ p, int lane) {
#pragma unroll
        for (int t = 0; t < NT; ++t) {
            const bf16_t* q = p + (t * 64 + lane) * VEC;
            if constexpr (VEC == 4) {
                const v2u u = *reinterpret_cast<const v2u*>(q);
                v[t * 4 + 0] += __uint_as_float(u.x << 16); v[t * 4 + 1] += __uint_as_float(u.x & 0xffff0000u);
                v[t * 4 + 2] += __uint_as_float(u.y << 16); v[t * 4 + 3] += __uint_as_float(u.y & 0xffff0000u);
            } else if constexpr (VEC == 2) {
                const uint32_t u = *reinterpret_cast<const uint32_t*>(q);
                v[t * 2 + 0] += __uint_as_float(u << 16); v[t * 2 + 1] += __uint_as_float(u & 0xffff0000u);
            } else {
                v[t] += bf2f(*q);
            }
        }
    }
    __device__ __forceinline__ void normalize(const float* __restrict__ w, const float* __restrict__ b, float eps,
                                              int lane) {
        constexpr float inv = 1.0f / (64 * VEC * NT);
        float s = 0.0f;
#pragma unroll
        for (int j = 0; j < VEC * NT; ++j) s += v[j];
        const float mean = wave_sum(s) * inv;
        float q = 0.0f;
#pragma unroll
        for (int j = 0; j < VEC * NT; ++j) { v[j] -= mean; q = __builtin_fmaf(v[j], v[j], q); }
        const float sd = sqrtf(wave_sum(q) * inv + eps);
#pragma unroll
        for (int t = 0; t < NT; ++t) {
            float ww[VEC], bb[VEC];
            ld_vec<VEC>(ww, w + (t * 64 + lane) * VEC);
            ld_vec<VEC>(bb, b + (t * 64 + lane) * VEC);
#pragma unroll
            for (int c = 0; c < VEC; ++c) v[t * VEC + c] = v[t * VEC + c] / sd * ww[c] + bb[c];
        }
    }
    template <typename T>
    __device__ __forceinline__ void store(T* __restrict__ p, int lane) const {
#pragma unroll
        for (int t = 0; t < NT; ++t) st_vec<VEC, T>(p + (t * 64 + lane) * VEC, &v[t * VEC]);
    }
    // fp32 row with the non-temporal hint: the residual stream is not read again before the next LayerNorm
    __device__ __forceinline__ void store_nt(float* __restrict__ p, int lane) const {
#pragma unroll
        for (int t = 0; t < NT; ++t)
#pragma unroll
            for (int c = 0; c < VEC; ++c) __builtin_nontemporal_store(v[t * VEC + c], p + (t * 64 + lane) * VEC + c);
    }
};

// y[row] = LN(x[row] + d1[row] + d2[row]) for row < rows; 4 rows per 256-thread block (bf16 path:
// the residual adds of the preceding out_proj / fc2, whose GEMM epilogues are pure bf16 stores).
//   d1, d2 nullable.  WRITE_BACK: x[row] <- the sum (once per layer, in LN1); LN2 only reads
//   x + d1 — it is added again, in the same order, by the next LN1, which saves one fp32 write of
//   the residual stream per layer.
//   y_ld: row pitch of y in elements.  split (bf16 only, MI_PRECISION_BF16_SPLIT): y[row][D + c] receives the part of
//   the normalised value that bf16 rounding dropped, lo = bf16(v - float(bf16(v))): the GEMM that follows runs
//   over K = 2D against [W | W] and sees the activations to ~16 significant bits.
template <typename T, int VEC, int NT, bool WRITE_BACK>
__device__ __forceinline__ void ln_body(float* __restrict__ x, const bf16_t* __restrict__ d1,
                                                 const bf16_t* __restrict__ d2, T* __restrict__ y,
                                                 const float* __restrict__ w, const float* __restrict__ b, int rows,
                                                 float eps, int y_ld, int split, int nt_x, size_t x_lo_off = 0, uint32_t x_bias = 0) {
    constexpr int D = 64 * VEC * NT;
    const int lane = threadIdx.x & 63;
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= rows) return;
    LnRow<VEC, NT> r;
    // x_lo_off != 0: x is the 24-bit residual in two planes (hi plane at x, lo plane x_lo_off bytes behind it)
    uint16_t* xhi = reinterpret_cast<uint16_t*>(x) + (size_t)row * D;
    uint8_t* xlo = reinterpret_cast<uint8_t*>(x) + x_lo_off + (size_t)row * D;
    if constexpr (VEC == 4 && sizeof(T) == 2) {
        if (x_lo_off) {
            r.load_x24(xhi, xlo, lane, x_bias);
            if (d1) r.add_bf16(d1 + (size_t)row * D, lane);
            if (d2) r.add_bf16(d2 + (size_t)row * D, lane);
            if (WRITE_BACK && (d1 || d2)) r.store_x24(xhi, xlo, lane, 0x80u + x_bias);
            r.normalize(w, b, eps, lane);
            r.store(y + (size_t)row * y_ld, lane);
            return;
        }
    }
    if (WRITE_BACK && (nt_x & 2)) {  // LN1, A/B: every operand is a last use
        r.load_nt(x + (size_t)row * D, lane);
        if (d1) r.add_bf16_nt(d1 + (size_t)row * D, lane);
        if (d2) r.add_bf16_nt(d2 + (size_t)row * D, lane);
    } else {
        r.load(x + (size_t)row * D, lane);
        if (d1) r.add_bf16(d1 + (size_t)row * D, lane);
        if (d2) r.add_bf16(d2 + (size_t)row * D, lane);
    }
    if (WRITE_BACK && (d1 || d2)) {
        if (nt_x & 1) r.store_nt(x + (size_t)row * D, lane);
        else r.store(x + (size_t)row * D, lane);
    }
    r.normalize(w, b, eps, lane);
    r.store(y + (size_t)row * y_ld, lane);
    if constexpr (sizeof(T) == 2) {
        if (split) {
#pragma unroll
            for (int j = 0; j < VEC * NT; ++j) r.v[j] -= bf2f(f2bf(r.v[j]));
            r.store(y + (size_t)row * y_ld + D, lane);
        }
    }
}

template <typename T, int VEC, int NT, bool WRITE_BACK>
__global__ __launch_bounds__(256) void ln_kernel(float* __restrict__ x, const bf16_t* __restrict__ d1,
                                                 const bf16_t* __restrict__ d2, T* __restrict__ y,
                                                 const float* __restrict__ w, const float* __restrict__ b, int rows,
                                                 float eps, int y_ld, int split, int nt_x = 0, size_t x_lo_off = 0,
                                                 uint32_t x_bias = 0) {
    ln_body<T, VEC, NT, WRITE_BACK>(x, d1, d2, y, w, b, rows, eps, y_ld, split, nt_x, x_lo_off, x_bias);
}
// token assembly + pre-LN (modeling_clip.py:198-218, :641-651):
// x[b*S+s] = LN_pre((s == 0 ? cls : patch[b*(S-1)+s-1]) + pos[s])
// ln_fold feeds q/k/v and fc1 the bf16 rounding of the UN-normalised residual row: with |mean| = r sigma every element carries
// an error of up to 2^-9 r sigma, i.e. about r times what rounding the normalised value costs.  Measured on ViT-L/14
// (tools/bf16_acceptance.py; max error / rms against fp32, bound 3e-2): r = 0, 1: 1.5e-2 (= the LayerNorm
// tower); r = 4: 2.3e-2; r = 16: 8.8e-2; r = 64: 0.30.  Rows with mean^2 > 16 var (r > 4) are counted
// (mi_clip_ln_fold_stats).  The tower removes the common mode of everything it writes to the stream (center_writer), so on
// handles loaded that way the count stays 0 whatever the checkpoint's biases are; it is the check, not the cure.
constexpr float LN_FOLD_OFFSET_LIMIT = 16.0f;

template <int VEC, int NT>
__global__ __launch_bounds__(256) void embed_ln_kernel(const float* __restrict__ patch, const float* __restrict__ cls,
                                                       const float* __restrict__ pos, float* __restrict__ x,
                                                       const float* __restrict__ w, const float* __restrict__ b,
                                                       int rows, int S, float eps, size_t x_lo_off = 0,
                                                       float* __restrict__ stats = nullptr, int rows_pad = 0,
                                                       unsigned long long* __restrict__ offset_rows = nullptr, int center = 0) {
    constexpr int D = 64 * VEC * NT;
    const int lane = threadIdx.x & 63;
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
    LnRow<VEC, NT> r;
    if constexpr (VEC == 4) {
        // stats != nullptr: the LayerNorm-free tower.  x goes out as the planes whose hi plane is bf16(x) (the first q/k/v
        // GEMM's operand), with {rstd, -mean * rstd} of the row for that GEMM's epilogue; the padding rows up to rows_pad
        // become zeros in that format (all-zero planes are not: they would read as NaN).
        if (stats) {
            if (row >= rows_pad) return;
            float st_a = 0.0f, st_b = 0.0f;
            if (row < rows) {
                const int bimg = row / S, s = row % S;
                r.load(s == 0 ? cls : patch + ((size_t)bimg * (S - 1) + (s - 1)) * D, lane);
                r.add(pos + (size_t)s * D, lane);
                r.normalize(w, b, eps, lane);
                constexpr float inv = 1.0f / D;
                if (center) {   // the stream's readers are LayerNorms: the row's common mode may go (vit.hip: center_writer)
                    float s0 = 0.0f;
#pragma unroll
                    for (int j = 0; j < VEC * NT; ++j) s0 += r.v[j];
                    const float m0 = wave_sum(s0) * inv;
#pragma unroll
                    for (int j = 0; j < VEC * NT; ++j) r.v[j] -= m0;
                }
                float sm = 0.0f;
#pragma unroll
                for (int j = 0; j < VEC * NT; ++j) sm += r.v[j];
                const float mean = wave_sum(sm) * inv;
                float q = 0.0f;
#pragma unroll
                for (int j = 0; j < VEC * NT; ++j) { const float d = r.v[j] - mean; q = __builtin_fmaf(d, d, q); }
                const float var = wave_sum(q) * inv;
                st_a = 1.0f / sqrtf(var + eps);
                st_b = -mean * st_a;
                // ln_fold's precondition (LN_FOLD_OFFSET_LIMIT): a row this far off zero has lost its bits in the hi plane
                if (offset_rows && lane == 0 && mean * mean > LN_FOLD_OFFSET_LIMIT * var) atomicAdd(offset_rows, 1ull);
            } else {
#pragma unroll
                for (int j = 0; j < VEC * NT; ++j) r.v[j] = 0.0f;
            }
            r.store_x24(reinterpret_cast<uint16_t*>(x) + (size_t)row * D, reinterpret_cast<uint8_t*>(x) + x_lo_off + (size_t)row * D, lane, 0x8080u);
            if (lane == 0) *reinterpret_cast<v2f*>(stats + (size_t)row * 2) = (v2f){st_a, st_b};
            return;
        }
    }
    if (row >= rows) return;
    const int bimg = row / S, s = row % S;
    r.load(s == 0 ? cls : patch + ((size_t)bimg * (S - 1) + (s - 1)) * D, lane);
    r.add(pos + (size_t)s * D, lane);
    r.normalize(w, b, eps, lane);
    if constexpr (VEC == 4) {
        if (x_lo_off) {   // the residual stream as 24-bit floats in two planes (LnRow::store_x24)
            r.store_x24(reinterpret_cast<uint16_t*>(x) + (size_t)row * D, reinterpret_cast<uint8_t*>(x) + x_lo_off + (size_t)row * D, lane);
            return;
        }
    }
    r.store(x + (size_t)row * D, lane);
}

// ------------------------------------------------------------------ row gather (last layer: CLS rows only)
// dst[i][:] = src[i * stride_rows][:], rows of D elements of T (16-byte pieces)
template <typename T>
__global__ void gather_rows_kernel(const T* __restrict__ src, T* __restrict__ dst, int n, size_t stride_rows, int D) {
    const int per = D * (int)sizeof(T) / 16;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < (size_t)n * per; i += (size_t)gridDim.x * blockDim.x) {
        const size_t r = i / per, c = i % per;
        reinterpret_cast<v4u*>(dst + r * D)[c] = reinterpret_cast<const v4u*>(src + r * stride_rows * D)[c];
    }
}

// dst[i * stride_rows][0 .. D) = src[i][0 .. D) with dst row pitch ld elements (the CLS queries of the last layer)
// dst[i][:] (fp32) = row i * stride_rows of the 24-bit residual stream (two planes, LnRow::load_x24); D a multiple of 4
__global__ void gather_rows_x24_kernel(const uint16_t* __restrict__ hi, const uint8_t* __restrict__ lo, float* __restrict__ dst, int n,
                                       size_t stride_rows, int D, uint32_t bias = 0) {
    const size_t total = (size_t)n * (D / 4);
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        const size_t r = i / (D / 4), c = (i % (D / 4)) * 4, src = r * stride_rows * D + c;
        const v2u h = *reinterpret_cast<const v2u*>(hi + src);
        const uint32_t l = *reinterpret_cast<const uint32_t*>(lo + src);
        v4f o;
        o.x = __uint_as_float(((h.x << 16) | ((l & 0xFFu) << 8)) - bias);
        o.y = __uint_as_float(((h.x & 0xFFFF0000u) | (l & 0xFF00u)) - bias);
        o.z = __uint_as_float(((h.y << 16) | ((l >> 8) & 0xFF00u)) - bias);
        o.w = __uint_as_float(((h.y & 0xFFFF0000u) | ((l >> 16) & 0xFF00u)) - bias);
        *reinterpret_cast<v4f*>(dst + r * D + c) = o;
    }
}
// dst[i] = stats[i * stride_rows] ({rstd, -mean rstd} of the CLS rows: the last layer's query GEMM runs on those rows only)
__global__ void gather_stats_kernel(const float* __restrict__ stats, float* __restrict__ dst, int n, size_t stride_rows) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) *reinterpret_cast<v2f*>(dst + (size_t)i * 2) = *reinterpret_cast<const v2f*>(stats + (size_t)i * stride_rows * 2);
}
// cs: elements between consecutive 64-column groups of a dst row (64: plain rows; PpFold::col_stride for head-major planes)
template <typename T>
__global__ void scatter_rows_kernel(const T* __restrict__ src, T* __restrict__ dst, int n, size_t stride_rows, int D, size_t ld,
                                    size_t cs = 64) {
    const int per = D * (int)sizeof(T) / 16;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < (size_t)n * per; i += (size_t)gridDim.x * blockDim.x) {
        const size_t r = i / per, c = i % per, col = c * (16 / sizeof(T));
        *reinterpret_cast<v4u*>(dst + r * stride_rows * ld + (col >> 6) * cs + (col & 63)) = reinterpret_cast<const v4u*>(src + r * D)[c];
    }
}

// ------------------------------------------------------------------ diagnostic: the shader clock right now
// MI355X_MICROARCH.md (6): in-kernel clock = delta s_memtime (shader cycles) / delta s_memrealtime (100 MHz) x 100 MHz,
// around a fixed dependent-FMA loop (~0.1 ms).  out = {shader cycles, reference ticks, fma result (keeps the loop alive)}
__global__ void clock_probe_kernel(unsigned long long* __restrict__ out) {
    float x = (float)threadIdx.x * 1e-3f;
    const unsigned long long r0 = __builtin_amdgcn_s_memrealtime();
    const unsigned long long c0 = __builtin_amdgcn_s_memtime();
#pragma unroll 1
    for (int i = 0; i < 40000; ++i) x = __builtin_fmaf(x, 0.999f, 1e-3f);
    const unsigned long long c1 = __builtin_amdgcn_s_memtime();
    const unsigned long long r1 = __builtin_amdgcn_s_memrealtime();
    if (threadIdx.x == 0) {
        out[0] = c1 - c0; out[1] = r1 - r0;
        out[2] = (unsigned long long)__float_as_uint(x);
    }
}

// ------------------------------------------------------------------ one text query: skinny GEMMs
// A text query is 77 token rows (server/src/clip.rs:19-23 sits in front of every search).  On those shapes a tiled GEMM
// is a latency chain: three to a dozen workgroups, each walking K in steps that pay an HBM round trip apiece (~10 us
// per kernel, ~90 kernels per query).  Here every workgroup owns 16 output columns and ONE K chunk of 768, all of whose
// operand loads (6 weight + 30 activation fragments per wave, 16 bytes per lane each) are in flight before the first
// MFMA: 48 - 192 workgroups stream the layer's weights side by side and a kernel lasts about one memory round trip.
//   C^T = W X^T on MFMA 16x16x32 (A = weights [n][k], B = activations [m][k], both K-contiguous, straight from global
//   memory in fragment order): a lane ends up with 4 consecutive output columns of one token row.  The four waves of a
//   workgroup split the K chunk (192 each) and meet in LDS; the sum runs wave 0..3, fixed order.
//   grid = (N / 16, K / 768): blockIdx.y > 0 only with EPI_SLAB (fc2, K = 3072), whose fp32 partial slabs
//   [K/768][SKINNY_ROWS][N] are summed — again in fixed order — by ln_slab_kernel.
enum { SKINNY_BIAS = 0, SKINNY_BIAS_QGELU = 1, SKINNY_SLAB = 2 };
constexpr int SKINNY_MT = 5, SKINNY_ROWS = 16 * SKINNY_MT, SKINNY_KC = 768;  // 80 token rows (77 live), K per workgroup
template <int EPI>
__global__ __launch_bounds__(256) void skinny_gemm_kernel(const bf16_t* __restrict__ X, int ldx, const bf16_t* __restrict__ W, int ldw,
                                                          const float* __restrict__ bias, void* __restrict__ out, int ldo) {
    __shared__ v4f red[4][SKINNY_MT][64];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int n0 = blockIdx.x * 16;
    const int k0 = blockIdx.y * SKINNY_KC + wave * (SKINNY_KC / 4) + 8 * (lane >> 4);
    const bf16_t* wp = W + (size_t)(n0 + (lane & 15)) * ldw + k0;
    const bf16_t* xp = X + (size_t)(lane & 15) * ldx + k0;
    bf16x8 a[6], b[SKINNY_MT][6];
    // this thread's epilogue items (threadIdx.x and threadIdx.x + 256) share one column group: its bias is fetched now,
    // beside the operands, not behind the reduction (one more memory round trip in a kernel that lasts about three)
    v4f bias4 = (v4f){0.0f, 0.0f, 0.0f, 0.0f};
    if constexpr (EPI != SKINNY_SLAB) bias4 = *reinterpret_cast<const v4f*>(bias + n0 + 4 * (lane >> 4));
#pragma unroll
    for (int ks = 0; ks < 6; ++ks) a[ks] = *reinterpret_cast<const bf16x8*>(wp + 32 * ks);
#pragma unroll
    for (int mt = 0; mt < SKINNY_MT; ++mt)
#pragma unroll
        for (int ks = 0; ks < 6; ++ks) b[mt][ks] = *reinterpret_cast<const bf16x8*>(xp + (size_t)(16 * mt) * ldx + 32 * ks);
    v4f acc[SKINNY_MT];
#pragma unroll
    for (int mt = 0; mt < SKINNY_MT; ++mt) acc[mt] = (v4f){0.0f, 0.0f, 0.0f, 0.0f};
#pragma unroll
    for (int ks = 0; ks < 6; ++ks)
#pragma unroll
        for (int mt = 0; mt < SKINNY_MT; ++mt) acc[mt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[ks], b[mt][ks], acc[mt], 0, 0, 0);
#pragma unroll
    for (int mt = 0; mt < SKINNY_MT; ++mt) red[wave][mt][lane] = acc[mt];
    __syncthreads();
    for (int item = threadIdx.x; item < SKINNY_MT * 64; item += 256) {
        const int mt = item >> 6, l = item & 63;
        v4f v = red[0][mt][l];
        v += red[1][mt][l]; v += red[2][mt][l]; v += red[3][mt][l];
        const int row = 16 * mt + (l & 15), col = n0 + 4 * (l >> 4);
        if constexpr (EPI == SKINNY_SLAB) {
            *reinterpret_cast<v4f*>(static_cast<float*>(out) + ((size_t)blockIdx.y * SKINNY_ROWS + row) * ldo + col) = v;
        } else {
            v += bias4;
            if constexpr (EPI == SKINNY_BIAS_QGELU) {
                v.x = quick_gelu<true>(v.x); v.y = quick_gelu<true>(v.y); v.z = quick_gelu<true>(v.z); v.w = quick_gelu<true>(v.w);
            }
            *reinterpret_cast<v2u*>(static_cast<bf16_t*>(out) + (size_t)row * ldo + col) = (v2u){pack2bf(v.x, v.y), pack2bf(v.z, v.w)};
        }
    }
}

// LayerNorm behind the skinny GEMMs: x[row] (+ d1[row], the out_proj output, bf16) (+ the n_slabs fc2 partial slabs, in
// order, + fc2's bias) -> optionally written back -> y[row] = LN(that).  One wave per row, as ln_kernel and with its
// arithmetic (LnRow), but the affine parameters are requested before the reductions: in a kernel this short a dependent
// load is a fifth of its duration.  ids != null (first layer): x[row] = token_embedding[ids[row]] + position_embedding[row].
template <int VEC, int NT>
struct LnSlabRow {
    LnRow<VEC, NT> r;
    __device__ __forceinline__ void gather(const float* __restrict__ x, const bf16_t* __restrict__ d1, const float* __restrict__ slabs,
                                           int n_slabs, const float* __restrict__ bias2, const int* __restrict__ ids,
                                           const float* __restrict__ tok, const float* __restrict__ pos, int row, int lane) {
        constexpr int D = 64 * VEC * NT;
        if (ids) {
            r.load(tok + (size_t)ids[row] * D, lane);
            r.add(pos + (size_t)row * D, lane);
        } else {
            r.load(x + (size_t)row * D, lane);
        }
        if (d1) r.add_bf16(d1 + (size_t)row * D, lane);
        int s = 0;
        for (; s + 4 <= n_slabs; s += 4) {   // four slabs' loads in flight at a time, added in slab order
            float u[4][VEC * NT];
#pragma unroll
            for (int j = 0; j < 4; ++j)
#pragma unroll
                for (int t = 0; t < NT; ++t) ld_vec<VEC>(&u[j][t * VEC], slabs + ((size_t)(s + j) * SKINNY_ROWS + row) * D + (t * 64 + lane) * VEC);
#pragma unroll
            for (int j = 0; j < 4; ++j)
#pragma unroll
                for (int i = 0; i < VEC * NT; ++i) r.v[i] += u[j][i];
        }
        for (; s < n_slabs; ++s) r.add(slabs + ((size_t)s * SKINNY_ROWS + row) * D, lane);
        if (n_slabs) r.add(bias2, lane);
    }
    // LnRow::normalize with w and b already in registers
    __device__ __forceinline__ void normalize(const float (&ww)[VEC * NT], const float (&bb)[VEC * NT], float eps) {
        constexpr float inv = 1.0f / (64 * VEC * NT);
        float s = 0.0f;
#pragma unroll
        for (int j = 0; j < VEC * NT; ++j) s += r.v[j];
        const float mean = wave_sum(s) * inv;
        float q = 0.0f;
#pragma unroll
        for (int j = 0; j < VEC * NT; ++j) { r.v[j] -= mean; q = __builtin_fmaf(r.v[j], r.v[j], q); }
        const float sd = sqrtf(wave_sum(q) * inv + eps);
#pragma unroll
        for (int j = 0; j < VEC * NT; ++j) r.v[j] = r.v[j] / sd * ww[j] + bb[j];
    }
};
template <int VEC, int NT>
__global__ __launch_bounds__(256) void ln_slab_kernel(float* __restrict__ x, const bf16_t* __restrict__ d1, const float* __restrict__ slabs,
                                                      int n_slabs, const float* __restrict__ bias2, bf16_t* __restrict__ y,
                                                      const float* __restrict__ w, const float* __restrict__ b, int rows, float eps,
                                                      int write_back, const int* __restrict__ ids, const float* __restrict__ tok,
                                                      const float* __restrict__ pos) {
    constexpr int D = 64 * VEC * NT;
    const int lane = threadIdx.x & 63;
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= rows) return;
    float ww[VEC * NT], bb[VEC * NT];
#pragma unroll
    for (int t = 0; t < NT; ++t) {
        ld_vec<VEC>(&ww[t * VEC], w + (t * 64 + lane) * VEC);
        ld_vec<VEC>(&bb[t * VEC], b + (t * 64 + lane) * VEC);
    }
    LnSlabRow<VEC, NT> q;
    q.gather(x, d1, slabs, n_slabs, bias2, ids, tok, pos, row, lane);
    if (write_back) q.r.store(x + (size_t)row * D, lane);
    q.normalize(ww, bb, eps);
    q.r.store(y + (size_t)row * D, lane);
}

// The end of one text query in one launch: the EOS row (first maximum of the ids: the EOS token is the largest id) with
// the last layer's residual adds, final_layer_norm, and the bias-free projection — out[e] = proj[e] . LN(x_eos), fp32
// throughout (the stored vector is never rounded to bf16 behind the last residual add).  One wave per output element;
// every wave redoes the row's LayerNorm (3 KB from L2) rather than wait for another kernel to do it once.
template <int VEC, int NT>
__global__ __launch_bounds__(256) void text_head_one_kernel(const float* __restrict__ x, const bf16_t* __restrict__ d1,
                                                            const float* __restrict__ slabs, int n_slabs, const float* __restrict__ bias2,
                                                            const float* __restrict__ w, const float* __restrict__ b,
                                                            const float* __restrict__ proj, float* __restrict__ out,
                                                            const int* __restrict__ ids, int S, int E, float eps) {
    constexpr int D = 64 * VEC * NT;
    const int lane = threadIdx.x & 63;
    const int e = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (e >= E) return;
    float pw[VEC * NT], ww[VEC * NT], bb[VEC * NT];
#pragma unroll
    for (int t = 0; t < NT; ++t) {
        ld_vec<VEC>(&pw[t * VEC], proj + (size_t)e * D + (t * 64 + lane) * VEC);
        ld_vec<VEC>(&ww[t * VEC], w + (t * 64 + lane) * VEC);
        ld_vec<VEC>(&bb[t * VEC], b + (t * 64 + lane) * VEC);
    }
    // argmax over the ids, first maximum: key = (id << 8) | (255 - position), S <= 128
    long long best = -1;
    for (int j = lane; j < S; j += 64) best = max(best, ((long long)ids[j] << 8) | (long long)(255 - j));
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        const int lo = __shfl_xor((int)best, o, 64), hi = __shfl_xor((int)(best >> 32), o, 64);
        best = max(best, ((long long)hi << 32) | (unsigned int)lo);
    }
    const int row = 255 - (int)(best & 0xff);
    LnSlabRow<VEC, NT> q;
    q.gather(x, d1, slabs, n_slabs, bias2, nullptr, nullptr, nullptr, row, lane);
    q.normalize(ww, bb, eps);
    float acc = 0.0f;
#pragma unroll
    for (int j = 0; j < VEC * NT; ++j) acc = __builtin_fmaf(q.r.v[j], pw[j], acc);
    acc = wave_sum(acc);
    if (lane == 0) out[e] = acc;
}

// ------------------------------------------------------------------ text tower front / pooling index
// x[r][:] = token_embedding[ids[r]] + position_embedding[r % S]   (CLIPTextEmbeddings.forward)
__global__ void text_embed_kernel(const int* __restrict__ ids, const float* __restrict__ tok,
                                  const float* __restrict__ pos, float* __restrict__ x, size_t rows, int S, int D) {
    const size_t total = rows * (size_t)(D / 4);
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        const size_t r = i / (D / 4);
        const int c = (int)(i % (D / 4)) * 4;
        const v4f a = *reinterpret_cast<const v4f*>(tok + (size_t)ids[r] * D + c);
        const v4f b = *reinterpret_cast<const v4f*>(pos + (size_t)(r % S) * D + c);
        *reinterpret_cast<v4f*>(x + r * D + c) = a + b;
    }
}
// row_of[i] = i*S + argmax_j ids[i][j] (first maximum): the EOS token is the largest id
__global__ void text_eos_rows_kernel(const int* __restrict__ ids, int n, int S, int* __restrict__ row_of) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    int best = 0, bv = ids[(size_t)i * S];
    for (int j = 1; j < S; ++j) {
        const int v = ids[(size_t)i * S + j];
        if (v > bv) { bv = v; best = j; }
    }
    row_of[i] = i * S + best;
}

// Pooled row + post-LN + bias-free projection (vision: modeling_clip.py:641-651, :944-950; text: the
// EOS row through final_layer_norm and text_projection):
// out[b][e] = sum_d proj[e][d] * LN_post(x[b*S] (+ d1 + d2))[d], all fp32.  One block per 8 images:
// their pooled rows sit in LDS and every projection row is fetched once per block, not per image.
template <int VEC, int NT>
__global__ __launch_bounds__(256) void head_kernel(const float* __restrict__ x, const bf16_t* __restrict__ delta,
                                                   const bf16_t* __restrict__ delta2, const float* __restrict__ w,
                                                   const float* __restrict__ b, const float* __restrict__ proj,
                                                   float* __restrict__ out, int n, int S, int E, float eps,
                                                   const int* __restrict__ row_of, size_t x_lo_off = 0, uint32_t x_bias = 0) {
    constexpr int D = 64 * VEC * NT, IMG = 8;
    __shared__ float pooled[IMG][D];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int img0 = blockIdx.x * IMG;
    for (int i = wave; i < IMG; i += 4) {
        const int img = img0 + i;
        LnRow<VEC, NT> r;
        if (img < n) {
            const size_t row = row_of ? (size_t)row_of[img] : (size_t)img * S;  // vision: the CLS row; text: the EOS row
            bool packed = false;
            if constexpr (VEC == 4) {
                if (x_lo_off) {   // 24-bit residual in two planes
                    r.load_x24(reinterpret_cast<const uint16_t*>(x) + row * D, reinterpret_cast<const uint8_t*>(x) + x_lo_off + row * D, lane, x_bias);
                    packed = true;
                }
            }
            if (!packed) r.load(x + row * D, lane);
            if (delta) r.add_bf16(delta + row * D, lane);
            if (delta2) r.add_bf16(delta2 + row * D, lane);
            r.normalize(w, b, eps, lane);
        } else {
#pragma unroll
            for (int j = 0; j < VEC * NT; ++j) r.v[j] = 0.0f;
        }
        r.store(&pooled[i][0], lane);
    }
    __syncthreads();
    // blockIdx.y picks a slice of the E outputs (each slice redoes the 8 LayerNorms: they are cheap,
    // the projection rows are what needs the parallelism)
    const int ec = (E + gridDim.y - 1) / gridDim.y;
    const int e_end = min(E, (int)(blockIdx.y + 1) * ec);
    for (int e = blockIdx.y * ec + wave; e < e_end; e += 4) {
        const float* pr = proj + (size_t)e * D;
        float acc[IMG];
#pragma unroll
        for (int i = 0; i < IMG; ++i) acc[i] = 0.0f;
#pragma unroll
        for (int t = 0; t < NT; ++t) {
            float pw[VEC];
            ld_vec<VEC>(pw, pr + (t * 64 + lane) * VEC);
#pragma unroll
            for (int i = 0; i < IMG; ++i)
#pragma unroll
                for (int c = 0; c < VEC; ++c) acc[i] = __builtin_fmaf(pw[c], pooled[i][(t * 64 + lane) * VEC + c], acc[i]);
        }
#pragma unroll
        for (int i = 0; i < IMG; ++i) {
            const float v = wave_sum(acc[i]);
            if (lane == 0 && img0 + i < n) out[(size_t)(img0 + i) * E + e] = v;
        }
    }
}

// ------------------------------------------------------------------ epilogues
// `v` = 4 consecutive output columns n..n+3 of token row m.
template <int EPI, typename TO, bool FAST>
__device__ __forceinline__ void epilogue4(v4f v, const float* __restrict__ bias, void* __restrict__ out, size_t m,
                                          int n, int ldo) {
    if constexpr (EPI != EPI_STORE_F32) {
        const v4f bv = *reinterpret_cast<const v4f*>(bias + n);
        v += bv;
    }
    if constexpr (EPI == EPI_BIAS_QGELU) {
#pragma unroll
        for (int c = 0; c < 4; ++c) v[c] = quick_gelu<FAST>(v[c]);
    }
    if constexpr (EPI == EPI_STORE_F32) {
        *reinterpret_cast<v4f*>(static_cast<float*>(out) + m * ldo + n) = v;
    } else if constexpr (EPI == EPI_BIAS_RESID) {
        v4f* p = reinterpret_cast<v4f*>(static_cast<float*>(out) + m * ldo + n);
        *p = *p + v;
    } else if constexpr (sizeof(TO) == 4) {
        *reinterpret_cast<v4f*>(static_cast<float*>(out) + m * ldo + n) = v;
    } else {
        v2u pk;
        pk.x = pack2bf(v[0], v[1]);
        pk.y = pack2bf(v[2], v[3]);
        *reinterpret_cast<v2u*>(static_cast<bf16_t*>(out) + m * ldo + n) = pk;
    }
}

// ------------------------------------------------------------------ fp32 GEMM (parity path)
// out[m][n] (+)= sum_k X[m][k] * W[n][k]; exact-f32 MFMA 32x32x2 (a k-ordered fmaf
// chain, bit for bit).  128x128x16 tiles, 4 waves (2x2), each 64x64 = 2x2 MFMA tiles.
// M % 128 == 0 (buffers are padded), N % 128 == 0, K % 16 == 0.
template <int EPI>
__global__ __launch_bounds__(256) void gemm_f32_kernel(const float* __restrict__ X, const float* __restrict__ W,
                                                       const float* __restrict__ bias, float* __restrict__ out, int N,
                                                       int K, int ldo) {
    constexpr int LDT = 132;
    __shared__ float Xs[2][16][LDT];
    __shared__ float Ws[2][16][LDT];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1, h = lane >> 5, l31 = lane & 31;
    const int nt = N / 128;
    const int tm = blockIdx.x / nt, tn = blockIdx.x % nt;
    const size_t m0 = (size_t)tm * 128;
    const int n0 = tn * 128;
    const int r = tid >> 2, kq = tid & 3;
    const float* xp = X + (m0 + r) * K + 4 * kq;
    const float* wp = W + ((size_t)n0 + r) * K + 4 * kq;
    v4f rx0, rx1, rw0, rw1;
    auto gload = [&](int kt) {
        rx0 = *reinterpret_cast<const v4f*>(xp + kt * 16);
        rx1 = *reinterpret_cast<const v4f*>(xp + (size_t)64 * K + kt * 16);
        rw0 = *reinterpret_cast<const v4f*>(wp + kt * 16);
        rw1 = *reinterpret_cast<const v4f*>(wp + (size_t)64 * K + kt * 16);
    };
    auto lstore = [&](int buf) {
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            Xs[buf][4 * kq + c][r] = rx0[c]; Xs[buf][4 * kq + c][r + 64] = rx1[c];
            Ws[buf][4 * kq + c][r] = rw0[c]; Ws[buf][4 * kq + c][r + 64] = rw1[c];
        }
    };
    v16f acc[2][2];
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int c = 0; c < 2; ++c)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[a][c][e] = 0.0f;

    const int nk = K / 16;
    gload(0);
    lstore(0);
    __syncthreads();
    int cur = 0;
    for (int kt = 0; kt < nk; ++kt) {
        if (kt + 1 < nk) gload(kt + 1);
#pragma unroll
        for (int kk = 0; kk < 8; ++kk) {
            float a[2], b[2];
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                a[i] = Ws[cur][2 * kk + h][wn * 64 + i * 32 + l31];
                b[i] = Xs[cur][2 * kk + h][wm * 64 + i * 32 + l31];
            }
#pragma unroll
            for (int ni = 0; ni < 2; ++ni)
#pragma unroll
                for (int mi = 0; mi < 2; ++mi)
                    acc[ni][mi] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[ni], b[mi], acc[ni][mi], 0, 0, 0);
        }
        if (kt + 1 < nk) lstore(cur ^ 1);
        __syncthreads();
        cur ^= 1;
    }
    // C[i = n][j = m]: lane holds column j = lane&31, rows i = (e&3) + 8*(e>>2) + 4*(lane>>5)
#pragma unroll
    for (int ni = 0; ni < 2; ++ni)
#pragma unroll
        for (int mi = 0; mi < 2; ++mi)
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                v4f v = {acc[ni][mi][4 * q], acc[ni][mi][4 * q + 1], acc[ni][mi][4 * q + 2], acc[ni][mi][4 * q + 3]};
                const size_t m = m0 + wm * 64 + mi * 32 + l31;
                const int n = n0 + wn * 64 + ni * 32 + 8 * q + 4 * h;
                epilogue4<EPI, float, false>(v, bias, out, m, n, ldo);
            }
}

// fp32 GEMM for a handful of rows (the text tower: one query = 77 token rows): the same arithmetic on
// 128 x 32 tiles, so that N/32 instead of N/128 workgroups share the work (4 waves = 4 row blocks of 32).
template <int EPI>
__global__ __launch_bounds__(256) void gemm_f32_n32_kernel(const float* __restrict__ X, const float* __restrict__ W,
                                                           const float* __restrict__ bias, float* __restrict__ out,
                                                           int N, int K, int ldo) {
    constexpr int LDT = 132;
    __shared__ float Xs[2][16][LDT];
    __shared__ float Ws[2][16][36];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int h = lane >> 5, l31 = lane & 31;
    const int nt = N / 32;
    const int tm = blockIdx.x / nt, tn = blockIdx.x % nt;
    const size_t m0 = (size_t)tm * 128;
    const int n0 = tn * 32;
    const int r = tid >> 2, kq = tid & 3;
    const float* xp = X + (m0 + r) * K + 4 * kq;
    const float* wp = W + ((size_t)n0 + (r & 31)) * K + 4 * kq;
    v4f rx0, rx1, rw0;
    auto gload = [&](int kt) {
        rx0 = *reinterpret_cast<const v4f*>(xp + kt * 16);
        rx1 = *reinterpret_cast<const v4f*>(xp + (size_t)64 * K + kt * 16);
        if (r < 32) rw0 = *reinterpret_cast<const v4f*>(wp + kt * 16);
    };
    auto lstore = [&](int buf) {
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            Xs[buf][4 * kq + c][r] = rx0[c]; Xs[buf][4 * kq + c][r + 64] = rx1[c];
            if (r < 32) Ws[buf][4 * kq + c][r] = rw0[c];
        }
    };
    v16f acc;
#pragma unroll
    for (int e = 0; e < 16; ++e) acc[e] = 0.0f;
    const int nk = K / 16;
    gload(0);
    lstore(0);
    __syncthreads();
    int cur = 0;
    for (int kt = 0; kt < nk; ++kt) {
        if (kt + 1 < nk) gload(kt + 1);
#pragma unroll
        for (int kk = 0; kk < 8; ++kk)
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(Ws[cur][2 * kk + h][l31], Xs[cur][2 * kk + h][wave * 32 + l31], acc, 0, 0, 0);
        if (kt + 1 < nk) lstore(cur ^ 1);
        __syncthreads();
        cur ^= 1;
    }
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        v4f v = {acc[4 * q], acc[4 * q + 1], acc[4 * q + 2], acc[4 * q + 3]};
        epilogue4<EPI, float, false>(v, bias, out, m0 + wave * 32 + l31, n0 + 8 * q + 4 * h, ldo);
    }
}

// ------------------------------------------------------------------ bf16 GEMM (throughput path)
// 128x128x64 tiles, 4 waves (2x2), each wave 64x64 = 4x4 MFMA 16x16x32 tiles.
// Both operand tiles go HBM -> LDS with global_load_lds_dwordx4 (1 KiB = 8 rows of
// 128 B per wave-instruction, LDS image linear) into two 32 KiB buffers; the
// ds_read_b128 bank conflicts of 128-byte rows are removed by an XOR swizzle of the
// 16-byte chunk index with (row & 7), applied to the per-lane SOURCE address and
// to the read address (the same involution on both sides).
// M % 128 == 0, N % 128 == 0, K % 64 == 0.
__device__ __forceinline__ void glds16(const void* gsrc, void* lds_wave_base) {
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)gsrc,
                                     (__attribute__((address_space(3))) void*)lds_wave_base, 16, 0, 0);
}

// XCD-aware block id: blocks b and b+8 share an XCD (round-robin dispatch), so give
// each XCD one contiguous chunk of the tile order (bijective for any grid size).
__device__ __forceinline__ uint32_t xcd_remap(uint32_t bid, uint32_t nb) {
    const uint32_t q = nb >> 3, r = nb & 7, xcd = bid & 7;
    return (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (bid >> 3);
}

template <int EPI, typename TO>
__global__ __launch_bounds__(256, 2) void gemm_bf16_kernel(const bf16_t* __restrict__ X, const bf16_t* __restrict__ W,
                                                           const float* __restrict__ bias, void* __restrict__ out,
                                                           int N, int K, int ldo) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];  // 2 x (X 16 KiB | W 16 KiB)
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 1, wn = wave & 1, g = lane >> 4, l15 = lane & 15;
    const int nt = N / 128;
    const uint32_t wg = xcd_remap(blockIdx.x, gridDim.x);
    const int tm = wg / nt, tn = wg % nt;
    const size_t m0 = (size_t)tm * 128;
    const int n0 = tn * 128;

    // staging: wave w fills rows [32w, 32w+32) of both tiles, 4 wave-instructions each
    const int rr = lane >> 3, p = lane & 7;
    const unsigned char* xsrc =
        reinterpret_cast<const unsigned char*>(X) + ((m0 + 32 * wave + rr) * K + 8 * (p ^ rr)) * 2;
    const unsigned char* wsrc =
        reinterpret_cast<const unsigned char*>(W) + (((size_t)n0 + 32 * wave + rr) * K + 8 * (p ^ rr)) * 2;
    const size_t row8 = (size_t)8 * K * 2;
    auto stage = [&](int buf, int kt) {
        unsigned char* xb = smem + buf * 32768 + wave * 4096;
        unsigned char* wb = xb + 16384;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            glds16(xsrc + j * row8 + (size_t)kt * 128, xb + j * 1024);
            glds16(wsrc + j * row8 + (size_t)kt * 128, wb + j * 1024);
        }
    };

    v4f acc[4][4];
#pragma unroll
    for (int a = 0; a < 4; ++a)
#pragma unroll
        for (int c = 0; c < 4; ++c) acc[a][c] = (v4f){0.0f, 0.0f, 0.0f, 0.0f};

    // fragment addresses: row = (tile row) + l15 -> row & 7 == lane & 7
    const int sw = lane & 7;
    const int a_off = (wn * 64 + l15) * 128;  // W tile (MFMA A operand)
    const int b_off = (wm * 64 + l15) * 128;  // X tile (MFMA B operand)
    auto compute = [&](int buf) {
        const unsigned char* xb = smem + buf * 32768;
        const unsigned char* wb = xb + 16384;
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            const int ch = ((4 * ks + g) ^ sw) << 4;
            bf16x8 a[4], b[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                a[i] = *reinterpret_cast<const bf16x8*>(wb + a_off + i * 2048 + ch);
                b[i] = *reinterpret_cast<const bf16x8*>(xb + b_off + i * 2048 + ch);
            }
#pragma unroll
            for (int ni = 0; ni < 4; ++ni)
#pragma unroll
                for (int mi = 0; mi < 4; ++mi)
                    acc[ni][mi] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[ni], b[mi], acc[ni][mi], 0, 0, 0);
        }
    };

    const int nk = K / 64;
    stage(0, 0);
    __syncthreads();
    int cur = 0;
    for (int kt = 0; kt + 1 < nk; ++kt) {
        stage(cur ^ 1, kt + 1);
        compute(cur);
        __syncthreads();
        cur ^= 1;
    }
    compute(cur);

    // C[i = n][j = m]: lane holds column j = lane&15, rows i = 4*(lane>>4) + e
#pragma unroll
    for (int ni = 0; ni < 4; ++ni)
#pragma unroll
        for (int mi = 0; mi < 4; ++mi) {
            const size_t m = m0 + wm * 64 + mi * 16 + l15;
            const int n = n0 + wn * 64 + ni * 16 + 4 * g;
            epilogue4<EPI, TO, true>(acc[ni][mi], bias, out, m, n, ldo);
        }
}

// ------------------------------------------------------------------ bf16 GEMM, persistent form
// The large-M GEMM of the tower.  256x256x64 tiles, 8 waves (2 along m x 4 along n), each wave a
// 128(m) x 64(n) slab = 8 x 4 MFMA 16x16x32 tiles (128 accumulator registers).
// LDS: 2 buffers x [XH0 | XH1 | WH0 | WH1] x 16 KiB = 128 KiB (+ 18 KiB epilogue patches + 2 KiB bias):
//   XH[h] LDS row 64*wm + r  = tile row 128*wm + 64*h + r   (r < 64)   — the X rows of quadrant-row h
//   WH[h] LDS row 32*wn + r  = tile row  64*wn + 32*h + r   (r < 32)   — the W rows of quadrant-column h
// The HBM/L2 -> LDS stream never drains: each K tile is staged as those four 16-KiB half-tiles by
// `buffer_load ... lds` (1 KiB = 8 rows of 128 B per wave-instruction, LDS image linear, the 16-byte
// chunk index XOR-swizzled with row&7 on the SOURCE address and on the ds_read_b128 address), one
// half-tile per phase, four phases ahead of its first use.  A phase computes one 64 x 32 quadrant of
// every wave's slab (16 MFMA) in the snake order (0,0) (0,1) (1,1) (1,0), so the fragments of the
// shared half stay in registers.  Waits are counted (`s_waitcnt vmcnt(4)` leaves two half-tiles in
// flight) and barriers are raw `s_barrier` (3 per K tile) — `__syncthreads()` would drain the LDS-DMA
// queue (cdna_hip_programming.md §5 "Pipelining across barriers").
// Persistent: gridDim.x <= #CUs workgroups walk tiles lb, lb+G, ...; the half-tile stream simply
// continues across the tile boundary (the first K tile of the next tile is staged during the last K
// tile of the current one) and the epilogue's 16 stores per lane are left in flight, counted, while
// the next tile computes (vmcnt counts loads, stores and LDS-DMA together in issue order: the first
// K tile after an epilogue waits vmcnt(4+16+1 bias DMA)) — instead of every CU draining its 128 KiB
// of output at the same moment.  All addressing is SGPR descriptor + 32-bit offsets: one VGPR per
// operand for the per-lane part, everything per-tile is scalar.  Operands and output < 4 GiB each.
// Measured ladder on the ViT-L/14 b=256 shapes (M = 65 792; qkv / out / fc1 / fc2, TFLOP/s):
//   128x128 two-barrier kernel            620 / 830 / 700 / 880
//   256x256, stage-all-then-compute       820 / 810 / 860 / 990
//   + half-tile pipeline, counted vmcnt   900 / 850 / 950 / 1100
//   + persistent, async row-wise epilogue 1060 / 960 / 1070 / 1100
//   + split last round                    1060 / 1040 / 1090 / 1200

typedef __amdgpu_buffer_rsrc_t rsrc_t;
__device__ __forceinline__ rsrc_t make_rsrc(const void* p, uint32_t bytes) {
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p), 0, bytes, 0x00020000);
}
// A 16-byte buffer_store with an SGPR offset reads its data registers AFTER it has issued (gfx950): a VALU
// write of them in the next instruction reaches memory, and LLVM's hazard recogniser pads only the form without an SGPR
// offset.  Every such store in this file goes through here: the wait states are explicit, and the asm's input operand keeps
// the data registers allocated up to it, so no VALU write of them can be scheduled in between.
__device__ __forceinline__ void buffer_store_b128(v4u d, rsrc_t r, uint32_t voff, uint32_t soff) {
    __builtin_amdgcn_raw_buffer_store_b128(d, r, voff, soff, 0);
    asm volatile("s_nop 1" :: "v"(d) : "memory");
}
// ... with a cache-policy operand (2 = nt).  The persistent GEMM's outputs are written once and read by a LATER kernel, while
// the operands it streams (an XCD's 6 MB of X panels and W tiles per round of tiles) live in a 4 MB L2: output lines
// allocated there push operands out.  Which epilogue stores carry nt is set per kind below; probe builds override the
// macros (tools/tower_ab.py --lib, profiles/r06_store_policy_ab.txt).
#ifndef MI_PP_STORE_AUX_QKV
#define MI_PP_STORE_AUX_QKV 2   // EPI_BIAS / EPI_LNF outputs: q|k|v (read by attention), the LayerNorm tower's deltas
#endif
#ifndef MI_PP_STORE_AUX_H
#define MI_PP_STORE_AUX_H 2     // EPI_*_QGELU outputs: h (read by fc2)
#endif
#ifndef MI_PP_RES_LOAD_AUX
#define MI_PP_RES_LOAD_AUX 0    // EPI_RESID24: the loads of the old planes (read once, then rewritten in place)
#endif
#ifndef MI_PP_X_AUX
#define MI_PP_X_AUX 0           // the X operand's LDS-DMA (probe: would evict-first X lines leave the W tiles in L2 across rounds?)
#endif
#ifndef MI_PP_W_AUX
#define MI_PP_W_AUX 0           // the W operand's LDS-DMA
#endif
#ifndef MI_PP_STORE_AUX_RES
#define MI_PP_STORE_AUX_RES 0   // EPI_RESID24: the residual planes, rewritten in place
#endif
template <int AUX>
__device__ __forceinline__ void buffer_store_b128_aux(v4u d, rsrc_t r, uint32_t voff, uint32_t soff) {
    __builtin_amdgcn_raw_buffer_store_b128(d, r, voff, soff, AUX);
    asm volatile("s_nop 1" :: "v"(d) : "memory");
}
__device__ __forceinline__ void glds16_buf(rsrc_t r, uint32_t voff, uint32_t soff, void* lds_wave_base) {
    __builtin_amdgcn_raw_ptr_buffer_load_lds(r, (__attribute__((address_space(3))) void*)lds_wave_base, 16, voff, soff,
                                             0, 0);
}
// the same with a cache-policy operand (AUX = 2: nt, for bytes ONE CU reads once)
template <int AUX>
__device__ __forceinline__ void glds16_buf_aux(rsrc_t r, uint32_t voff, uint32_t soff, void* lds_wave_base) {
    __builtin_amdgcn_raw_ptr_buffer_load_lds(r, (__attribute__((address_space(3))) void*)lds_wave_base, 16, voff, soff,
                                             0, AUX);
}


// ------------------------------------------------------------------ the residual stream of the LayerNorm-free tower
// 24-bit floats in two planes like LnRow::load_x24, but rounded so that the hi plane IS bf16(x), the operand of the GEMM
// that follows: b' = bits(x) + 0x8080 (to 24 bits, nearest, plus half a bf16 ulp); hi = b' >> 16 (bf16(x), nearest, ties
// away from zero), lo = (b' >> 8) & 0xFF; x to 24 bits = ((hi << 16) | (lo << 8)) - 0x8000.  (0.0 is hi = 0, lo = 0x80.)
constexpr uint32_t X24B_BIAS = 0x8000u;
template <uint32_t CTRL>
__device__ __forceinline__ float dpp_movf(float v) {
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, 0xF, 0xF, true));
}
// One step of the EPI_RESID24 epilogue: 8 consecutive columns of one row.  h, l: the old planes (8 x u16, 8 x u8);
// d: 8 x bf16 of acc + bias.  x_new = x_old + d in fp32; returns the new planes and the lane's sum / sum of squares of
// x_new.  The ORDER of those sums is part of the contract (whole tiles and quadrant tasks must agree to the bit, and a row
// must not depend on where it lies): even and odd columns accumulate separately in column order, then even + odd; the
// caller adds the four lanes of a 32-column block as (l ^ 1) then (l ^ 2).
__device__ __forceinline__ void resid24_step(const v4u h, const v2u l, const v4u d, v4u& hn, v2u& ln, float& S, float& Q) {
    float x[8];
#pragma unroll
    for (int i = 0; i < 4; ++i) {   // dword i: columns 2i, 2i + 1
        const uint32_t lw = i < 2 ? l.x : l.y;
        const uint32_t a = __builtin_amdgcn_perm(h[i], lw, (i & 1) ? 0x0504020Cu : 0x0504000Cu) - X24B_BIAS;
        const uint32_t b = __builtin_amdgcn_perm(h[i], lw, (i & 1) ? 0x0706030Cu : 0x0706010Cu) - X24B_BIAS;
        x[2 * i] = __uint_as_float(a) + __uint_as_float(d[i] << 16);
        x[2 * i + 1] = __uint_as_float(b) + __uint_as_float(d[i] & 0xffff0000u);
    }
    v2f s2 = {x[0], x[1]}, q2 = {x[0] * x[0], x[1] * x[1]};
#pragma unroll
    for (int i = 1; i < 4; ++i) {
        const v2f e = {x[2 * i], x[2 * i + 1]};
        s2 += e;
        q2 = __builtin_elementwise_fma(e, e, q2);
    }
    S = s2.x + s2.y;
    Q = q2.x + q2.y;
    uint32_t r[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) r[i] = __float_as_uint(x[i]) + 0x8080u;
#pragma unroll
    for (int i = 0; i < 4; ++i) hn[i] = __builtin_amdgcn_perm(r[2 * i + 1], r[2 * i], 0x07060302u);
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const uint32_t t01 = __builtin_amdgcn_perm(r[4 * i + 1], r[4 * i], 0x0C0C0501u);
        const uint32_t t23 = __builtin_amdgcn_perm(r[4 * i + 3], r[4 * i + 2], 0x0C0C0501u);
        ln[i] = __builtin_amdgcn_perm(t23, t01, 0x05040100u);
    }
}

// partial sums -> {rstd, -mean * rstd} per row.  16 lanes per row: lane j adds blocks 2j and 2j + 1 of each group of 32
// blocks (the row's nb * 8 bytes are read as contiguous 16-byte pieces), groups accumulate in order, then the 16 lanes add
// as a fixed butterfly (l ^ 1, l ^ 2, mirror in 8, mirror in 16) — one order, whatever the launch.
// var = E[x^2] - mean^2 in fp32 on fp32 partial sums: fine while |mean| is not orders of magnitude above the deviation
// (a pre-LN residual stream's never is).
// rows_live / offset_rows: rows < rows_live with mean^2 > LN_FOLD_OFFSET_LIMIT * var are counted (see embed_ln_kernel).
__global__ __launch_bounds__(256) void ln_stats_kernel(const float* __restrict__ part, float* __restrict__ stats, int rows, int nb,
                                                       float inv_d, float eps, int rows_live = 0,
                                                       unsigned long long* __restrict__ offset_rows = nullptr) {
    const int row = blockIdx.x * 16 + (threadIdx.x >> 4), j = threadIdx.x & 15;
    if (row >= rows) return;   // whole 16-lane groups leave together: the DPP adds below stay inside a group
    const v4f* p = reinterpret_cast<const v4f*>(part + (size_t)row * nb * 2);
    float s = 0.0f, q = 0.0f;
    for (int i = j; i < nb / 2; i += 16) {   // nb is even (D a multiple of 256)
        const v4f v = p[i];
        s += v.x + v.z;
        q += v.y + v.w;
    }
    s += dpp_movf<0xB1>(s); q += dpp_movf<0xB1>(q);
    s += dpp_movf<0x4E>(s); q += dpp_movf<0x4E>(q);
    s += dpp_movf<0x141>(s); q += dpp_movf<0x141>(q);   // row_half_mirror
    s += dpp_movf<0x140>(s); q += dpp_movf<0x140>(q);   // row_mirror
    const float mean = s * inv_d;
    const float var = fmaxf(q * inv_d - mean * mean, 0.0f);
    const float rstd = 1.0f / sqrtf(var + eps);
    if (j == 0) {
        *reinterpret_cast<v2f*>(stats + (size_t)row * 2) = (v2f){rstd, -mean * rstd};
        if (row < rows_live && mean * mean > LN_FOLD_OFFSET_LIMIT * var) atomicAdd(offset_rows, 1ull);
    }
}

// ------------------------------------------------------------------ bf16 GEMM, persistent, two staggered wave groups
// Tile (256 x 256 x 64), LDS image and epilogue as described above; the schedule:
// the two wave rows (waves 0-3 / 4-7: one wave of each per SIMD) run ONE BARRIER APART, so that in
// every barrier interval one group issues MFMAs while the other fetches its next fragments from LDS
// and issues the next half-tiles' LDS-DMA -- the LDS latency and bandwidth that the one-barrier form
// pays in front of every MFMA cluster are hidden behind the partner group.
//   a phase is a HALF K tile (two quadrants, 32 MFMAs):
//     ds_read(fragments) ; lgkmcnt(0) ; stage half-tiles ; s_waitcnt vmcnt ; BARRIER ; 32 MFMA ; BARRIER
//   phase A of K tile t: reads XH0 WH0 WH1 (16 ds_read_b128) -> quadrants (0,0) (0,1); stages XH1(t+1)
//   phase B of K tile t: reads XH1 (8)                        -> quadrants (1,1) (1,0); stages XH0 WH0 WH1(t+2)
//   (group 1 enters one barrier late; W fragments of both n-halves stay in registers: 24 reads per
//   K tile, 214 VGPRs.)  The fragment reads are retired BEFORE the phase's first barrier, so a slot
//   may be refilled one phase after its last read; every half-tile is read 3 phases after it was
//   issued and one phase after the counted wait (vmcnt(8): the issues of this and the previous
//   phase) that retires it -- the wait of phase q is in front of q's first barrier, the read behind
//   q's second one, which both groups have passed only after both have waited.
//   epilogue of tile T: inside phase A of tile T+1, for both groups in the same barrier interval
//   (group 1 in front of its barrier, group 0 behind its own = the same instance); its 16 stores and
//   the bias DMA stay in the queue for the next two waits (vmcnt(25)).
//   A four-phase form (16 MFMAs per barrier interval, 8 barriers per K tile) measured 2-3 % slower.
// PP_CLOCK_BEGIN / PP_CLOCK_END: hooks of the diagnostic build of tools/probe/gemm_pp_sweep.hip (-DPP_CLOCK: the shader
// clock the chip holds under this kernel, MI355X_MICROARCH.md 'DVFS give-back' item 6); empty here.
#ifndef PP_CLOCK_BEGIN
#define PP_CLOCK_BEGIN
#define PP_CLOCK_END
#endif
// STORE_NT: the plain output stores (every epilogue but EPI_RESID24) carry the nt cache policy (MI_PP_STORE_AUX_* above);
// 0 = default policy, the A/B partner of option "store_nt".
template <int EPI, typename TO, bool STORE_NT = true>
__global__ __launch_bounds__(512, 2) void gemm_bf16_pp_kernel(const bf16_t* __restrict__ X,
                                                              const bf16_t* __restrict__ W,
                                                              const float* __restrict__ bias,
                                                              void* __restrict__ out, int M, int N, int K, int ldo,
                                                              int n_tiles, int n_full, int order, const PpFold fold = PpFold()) {
    static_assert(EPI == EPI_BIAS || EPI == EPI_BIAS_QGELU || EPI == EPI_LNF || EPI == EPI_LNF_QGELU || EPI == EPI_RESID24,
                  "the persistent form stores bf16 with bias");
    static_assert(sizeof(TO) == 2, "bf16 output");
    constexpr bool LNF = EPI == EPI_LNF || EPI == EPI_LNF_QGELU;   // LayerNorm folded into this linear (PpFold::cvec, stats)
    constexpr bool RES = EPI == EPI_RESID24;                        // residual add + row sums in the epilogue (PpFold::xlo, part)
    constexpr bool GELU = EPI == EPI_BIAS_QGELU || EPI == EPI_LNF_QGELU;
    // per wave behind the patches: 256 B bias [+ 256 B c + 1 KiB {rstd, -mean rstd} of the wave's 128 rows]
    constexpr int AUX = LNF ? 1536 : 256;
    constexpr int AUX_OPS = LNF ? 3 : 1;   // LDS-DMA instructions of stage_aux
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];  // 128 KiB staging + 18 KiB patches + 8 x AUX
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 2, wn = wave & 3, g = lane >> 4, l15 = lane & 15;
    const int nt = N / 256;
    const int G = gridDim.x;
    const int lb = (int)xcd_remap(blockIdx.x, G);
    PP_CLOCK_BEGIN

    const uint32_t Kb = (uint32_t)K * 2;
    const rsrc_t xr = make_rsrc(X, (uint32_t)M * Kb);
    const rsrc_t wr = make_rsrc(W, (uint32_t)N * Kb);
    const uint32_t cs = fold.col_stride;
    const rsrc_t orr = make_rsrc(out, cs == 64u ? (uint32_t)M * (uint32_t)ldo * (uint32_t)sizeof(TO) : (uint32_t)(N / 64) * cs * (uint32_t)sizeof(TO));
    const rsrc_t br = make_rsrc(bias, (uint32_t)N * 4u);
    const int rr = lane >> 3, p = lane & 7;
    const uint32_t x_lane = (uint32_t)rr * Kb + 16 * (p ^ rr);
    const uint32_t x_wave = (uint32_t)(128 * (wave >> 2) + 16 * (wave & 3)) * Kb;
    const uint32_t w_wave = (uint32_t)(64 * (wave >> 1) + 16 * (wave & 1)) * Kb;
    // piece j of a K tile: 0 = XH0, 1 = WH0, 2 = WH1, 3 = XH1
    auto stage_half = [&](int buf, uint32_t xs, uint32_t ws, int kt, int j) {
        const bool is_x = (j == 0 || j == 3);
        const int h = (j >= 2) ? 1 : 0;
        unsigned char* dst = smem + buf * 65536 + (is_x ? 0 : 32768) + h * 16384 + wave * 2048;
        const uint32_t so = (is_x ? xs + x_wave + 64u * h * Kb : ws + w_wave + 32u * h * Kb) + (uint32_t)kt * 128u;
        if (is_x) {
            glds16_buf_aux<MI_PP_X_AUX>(xr, x_lane, so, dst);
            glds16_buf_aux<MI_PP_X_AUX>(xr, x_lane, so + 8u * Kb, dst + 1024);
        } else {
            glds16_buf_aux<MI_PP_W_AUX>(wr, x_lane, so, dst);
            glds16_buf_aux<MI_PP_W_AUX>(wr, x_lane, so + 8u * Kb, dst + 1024);
        }
    };
    unsigned char* bias_lds = smem + 131072 + 18432 + wave * AUX;
    const rsrc_t cr = make_rsrc(LNF ? fold.cvec : bias, (uint32_t)N * 4u);
    const rsrc_t sr = make_rsrc(LNF ? fold.stats : bias, LNF ? (uint32_t)M * 8u : 0u);
    // wave-private: bias (and c) of the wave's 64 columns, and for LNF the row statistics of its 128 rows
    auto stage_aux = [&](int tm, int tn) {
        __builtin_amdgcn_raw_ptr_buffer_load_lds(br, (__attribute__((address_space(3))) void*)bias_lds, 4,
                                                 (uint32_t)lane * 4u, (uint32_t)(tn * 256 + wn * 64) * 4u, 0, 0);
        if constexpr (LNF) {
            __builtin_amdgcn_raw_ptr_buffer_load_lds(cr, (__attribute__((address_space(3))) void*)(bias_lds + 256), 4,
                                                     (uint32_t)lane * 4u, (uint32_t)(tn * 256 + wn * 64) * 4u, 0, 0);
            __builtin_amdgcn_raw_ptr_buffer_load_lds(sr, (__attribute__((address_space(3))) void*)(bias_lds + 512), 16,
                                                     (uint32_t)lane * 16u, (uint32_t)(tm * 256 + wm * 128) * 8u, 0, 0);
        }
    };

    const int sw = lane & 7;
    const int x_off = (wm * 64 + l15) * 128;
    const int w_off = 32768 + (wn * 32 + l15) * 128;
    bf16x8 xf[2][4], w0f[2][2], w1f[2][2];
    v4f acc[4][8];
    auto load_x = [&](const unsigned char* base, int mh) {
#pragma unroll
        for (int ks = 0; ks < 2; ++ks)
#pragma unroll
            for (int i = 0; i < 4; ++i)
                xf[ks][i] = *reinterpret_cast<const bf16x8*>(base + x_off + mh * 16384 + i * 2048 + (((4 * ks + g) ^ sw) << 4));
    };
    auto load_w = [&](bf16x8 (&wf)[2][2], const unsigned char* base, int nh) {
#pragma unroll
        for (int ks = 0; ks < 2; ++ks)
#pragma unroll
            for (int i = 0; i < 2; ++i)
                wf[ks][i] = *reinterpret_cast<const bf16x8*>(base + w_off + nh * 16384 + i * 2048 + (((4 * ks + g) ^ sw) << 4));
    };
#define PP_BAR { asm volatile("" ::: "memory"); __builtin_amdgcn_sched_barrier(0); __builtin_amdgcn_s_barrier(); __builtin_amdgcn_sched_barrier(0); asm volatile("" ::: "memory"); }
#define PP_QUADRANT(MH, NH, WF)                                                                            \
    __builtin_amdgcn_s_setprio(1);                                                                        \
    _Pragma("unroll") for (int ks = 0; ks < 2; ++ks)                                                      \
    _Pragma("unroll") for (int im = 0; im < 4; ++im)                                                      \
    _Pragma("unroll") for (int in = 0; in < 2; ++in)                                                      \
        acc[2 * NH + in][4 * MH + im] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(                          \
            WF[ks][in], xf[ks][im], acc[2 * NH + in][4 * MH + im], 0, 0, 0);                               \
    __builtin_amdgcn_s_setprio(0);

    // position in the flattened (tile, K tile) stream of this workgroup
    struct Pos { uint32_t xs, ws; int kt, tile, tm, tn; bool ok; };
    const int nk = K / 64;
    // visit index -> tile.  order == 0: row-major (an XCD's 32 concurrent tiles = 32 / nt row panels x ALL nt weight tiles:
    // every XCD streams the whole W every round).  order = np > 0 (nt % np == 0): the complete rounds of an XCD are made
    // contiguous (its visits r * G + lb become c * R * per + r * per + lb % per), and the sequence walks column groups of np
    // weight tiles top to bottom — an XCD's 32 concurrent tiles are a (32 / np) x np patch and it stays in one or two
    // column groups for the whole launch (np x 256 x K weights stay in its L2; an X panel is read by the nt / np XCDs that
    // walk the same rows at about the same time).  A bijection on [0, n_tiles): the tail tasks use the same map.
    const int mt = M / 256;
    auto tile_of = [&](int t, int& tm, int& tn) {
        if (order <= 0) { tm = t / nt; tn = t - tm * nt; return; }
        int sq = t;
        const int R = n_full / G;
        if (t < R * G && (G & 7) == 0) {
            const int per = G >> 3, r = t / G, lbv = t - r * G, c = lbv / per;
            sq = c * (R * per) + r * per + (lbv - c * per);
        }
        const int span = mt * order, grp = sq / span, rem = sq - grp * span;
        tm = rem / order; tn = grp * order + (rem - tm * order);
    };
    auto pos_of_tile = [&](int t) {
        Pos q;
        q.tile = t; q.kt = 0; q.ok = t < n_full;
        tile_of(t, q.tm, q.tn);
        q.xs = (uint32_t)q.tm * 256u * Kb; q.ws = (uint32_t)q.tn * 256u * Kb;
        return q;
    };
    auto advance = [&](const Pos& a) {
        if (!a.ok) return a;
        if (a.kt + 1 < nk) { Pos q = a; q.kt = a.kt + 1; return q; }
        return pos_of_tile(a.tile + G);
    };
    const uint32_t o_lane = ((uint32_t)(wm * 128 + (lane >> 3)) * (uint32_t)ldo + (uint32_t)wn * cs + (uint32_t)(8 * (lane & 7))) * 2u;
    unsigned char* patch = smem + 131072 + wave * 2304;
    // EPI_RESID24: the residual planes are updated in place (hi = `out`, bf16 pitch ldo; lo = fold.xlo, byte pitch ldo)
    const rsrc_t lor = make_rsrc(RES ? (const void*)fold.xlo : (const void*)out, RES ? (uint32_t)M * (uint32_t)ldo : 0u);
    const int nslot = N / 32;   // 32-column blocks per row of `part`
    const rsrc_t pr = make_rsrc(RES ? (const void*)fold.part : (const void*)out, RES ? (uint32_t)M * (uint32_t)nslot * 8u : 0u);
    const int t4 = lane & 3;
    const uint32_t p_lane = ((uint32_t)(wm * 128 + (t4 >> 1) * 16 + (t4 & 1) * 8 + (lane >> 3)) * (uint32_t)nslot + (uint32_t)(wn * 2 + ((lane >> 2) & 1))) * 8u;
    // RES_P: steps (8 rows x 64 columns of the wave) whose old planes are in flight ahead of the step being added
#ifndef MI_RES_P
#define MI_RES_P 8   // probe builds sweep it (tools/probe/gemm_fold_sweep.hip -DMI_RES_P=n): 4 / 8 / 12 read within noise of one another
#endif
    constexpr int RES_P = MI_RES_P;
    // vector-memory instructions an epilogue leaves in the queue behind the operand DMA of the phase it runs in
    constexpr int EPI_OPS = RES ? 2 * RES_P + 4 : 16;
    // bias (or the folded LayerNorm) + activation + bf16 + 128-byte row segments through the wave's LDS patch; clears acc
    auto epilogue = [&](int tm, int tn) {
        __builtin_amdgcn_sched_barrier(0);
        const uint32_t o_tile = ((uint32_t)tm * 256u * (uint32_t)ldo + (uint32_t)tn * 4u * cs) * 2u;
        v4f bv[4];
#pragma unroll
        for (int ni = 0; ni < 4; ++ni) bv[ni] = *reinterpret_cast<const v4f*>(bias_lds + (ni * 16 + 4 * g) * 4);
        v4f cv[4];
        if constexpr (LNF) {
#pragma unroll
            for (int ni = 0; ni < 4; ++ni) cv[ni] = *reinterpret_cast<const v4f*>(bias_lds + 256 + (ni * 16 + 4 * g) * 4);
        }
        v4u hq[RES_P];
        v2u lq[RES_P];
        float ks[4] = {0.0f, 0.0f, 0.0f, 0.0f}, kq[4] = {0.0f, 0.0f, 0.0f, 0.0f};
        auto fetch = [&](int st) {   // old planes of step st = (mi, j): rows mi * 16 + j * 8 + (lane >> 3)
            const uint32_t so = o_tile + (uint32_t)((st >> 1) * 16 + (st & 1) * 8) * (uint32_t)ldo * 2u;
            hq[st % RES_P] = __builtin_amdgcn_raw_buffer_load_b128(orr, o_lane, so, MI_PP_RES_LOAD_AUX);
            lq[st % RES_P] = __builtin_amdgcn_raw_buffer_load_b64(lor, o_lane >> 1, so >> 1, MI_PP_RES_LOAD_AUX);
        };
        if constexpr (RES) {   // in step order: the first wait must not stand behind the whole burst
#pragma unroll
            for (int st = 0; st < RES_P; ++st) { fetch(st); __builtin_amdgcn_sched_barrier(0); }
        }
#pragma unroll
        for (int mi = 0; mi < 8; ++mi) {
            v2f st2;
            if constexpr (LNF) st2 = *reinterpret_cast<const v2f*>(bias_lds + 512 + (mi * 16 + l15) * 8);
#pragma unroll
            for (int ni = 0; ni < 4; ++ni) {
                v4f v;
                if constexpr (LNF) {
#pragma unroll
                    for (int c = 0; c < 4; ++c)
                        v[c] = __builtin_fmaf(st2.x, acc[ni][mi][c], __builtin_fmaf(st2.y, cv[ni][c], bv[ni][c]));
                } else {
                    v = acc[ni][mi] + bv[ni];
                }
                acc[ni][mi] = (v4f){0.0f, 0.0f, 0.0f, 0.0f};
                if constexpr (GELU) {
#pragma unroll
                    for (int c = 0; c < 4; ++c) v[c] = quick_gelu<true>(v[c]);
                }
                v2u pk;
                pk.x = pack2bf(v[0], v[1]);
                pk.y = pack2bf(v[2], v[3]);
                *reinterpret_cast<v2u*>(patch + l15 * 144 + (ni * 16 + 4 * g) * 2) = pk;
            }
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                const int row = j * 8 + (lane >> 3);
                const v4u d = *reinterpret_cast<const v4u*>(patch + row * 144 + (lane & 7) * 16);
                const uint32_t so = o_tile + (uint32_t)(mi * 16 + j * 8) * (uint32_t)ldo * 2u;
                if constexpr (RES) {
                    const int st = 2 * mi + j;
                    v4u hn; v2u ln; float S, Q;
                    resid24_step(hq[st % RES_P], lq[st % RES_P], d, hn, ln, S, Q);
                    __builtin_amdgcn_sched_barrier(0);
                    buffer_store_b128_aux<MI_PP_STORE_AUX_RES>(hn, orr, o_lane, so);
                    __builtin_amdgcn_raw_buffer_store_b64(ln, lor, o_lane >> 1, so >> 1, MI_PP_STORE_AUX_RES);
                    if (st + RES_P < 16) fetch(st + RES_P);
                    S += dpp_movf<0xB1>(S); Q += dpp_movf<0xB1>(Q);   // the four lanes of a 32-column block
                    S += dpp_movf<0x4E>(S); Q += dpp_movf<0x4E>(Q);
                    if (t4 == (st & 3)) { ks[st >> 2] = S; kq[st >> 2] = Q; }   // lane t4 of the block keeps steps t4, t4 + 4, ...
                    // The 16-byte store reads its data registers AFTER it has issued: a VALU write of them in the very next
                    // instruction (hipcc reused them for a v_pk_mul_f32) reached memory in the last lanes of every row of 16
                    // (the hazard recogniser pads only stores without an SGPR offset).  Keep them allocated past
                    // the row sums above.
                    __builtin_amdgcn_sched_barrier(0);
                    asm volatile("" :: "v"(hn), "v"(ln));
                } else {
                    buffer_store_b128_aux<STORE_NT ? (GELU ? MI_PP_STORE_AUX_H : MI_PP_STORE_AUX_QKV) : 0>(d, orr, o_lane, so);
                }
            }
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_sched_barrier(0);
        }
        if constexpr (RES) {
            // lane t4 holds steps 4u + t4: rows (2u + (t4 >> 1)) * 16 + (t4 & 1) * 8 + (lane >> 3) of the wave's 128
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const v2u pv = {__float_as_uint(ks[u]), __float_as_uint(kq[u])};
                __builtin_amdgcn_raw_buffer_store_b64(pv, pr, p_lane, (uint32_t)((tm * 256 + 32 * u) * nslot + tn * 8) * 8u, 0);
            }
        }
    };

#pragma unroll
    for (int a = 0; a < 4; ++a)
#pragma unroll
        for (int c = 0; c < 8; ++c) acc[a][c] = (v4f){0.0f, 0.0f, 0.0f, 0.0f};

    Pos C = pos_of_tile(lb);
    if (C.ok) {
        // prologue = the staging of phases -3 .. -1: K tile 0 whole, XH0 WH0 WH1 of K tile 1
        Pos A = advance(C);  // nk >= 2: same tile
        stage_aux(C.tm, C.tn);
#pragma unroll
        for (int j = 0; j < 4; ++j) stage_half(0, C.xs, C.ws, 0, j);
#pragma unroll
        for (int j = 0; j < 3; ++j) stage_half(1, A.xs, A.ws, A.kt, j);
        asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
        PP_BAR
        if (wm == 1) PP_BAR // group 1 runs one barrier behind from here on
        int b = 0, relax = 0;
        bool epi = false;
        int e_tm = 0, e_tn = 0;
// one phase.  READS: this phase's fragment loads; (PK, PB, PJ): the half-tile it stages
#define PP_WAIT(PK)                                                                   \
    if (!(PK).ok) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                    \
    else if (relax > 0) { asm volatile("s_waitcnt vmcnt(%0)" :: "n"(8 + EPI_OPS + AUX_OPS) : "memory"); --relax; } \
    else asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
#define PP_STAGE(PK, PB, PJ) if ((PK).ok) stage_half(PB, (PK).xs, (PK).ws, (PK).kt, PJ);
        for (;;) {
            const Pos B = advance(A);
            const unsigned char* base = smem + b * 65536;
            // ---- phase A: X half 0, both W halves -> quadrants (0,0) (0,1); stages XH1 of the next K tile
            if (!epi) {
                load_w(w0f, base, 0); load_w(w1f, base, 1); __builtin_amdgcn_sched_barrier(0); load_x(base, 0);
                PP_STAGE(A, b ^ 1, 3) PP_WAIT(A)
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                PP_BAR
            } else {
                // both groups run the previous tile's epilogue in the SAME barrier interval: group 1
                // in front of its barrier, group 0 behind its own (which is the same instance)
                PP_STAGE(A, b ^ 1, 3) PP_WAIT(A)
                if (wm == 0) PP_BAR
                epilogue(e_tm, e_tn);
                stage_aux(C.tm, C.tn);
                relax = 2;
                load_w(w0f, base, 0); load_w(w1f, base, 1); __builtin_amdgcn_sched_barrier(0); load_x(base, 0);
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                if (wm == 1) PP_BAR
            }
            epi = false;
            PP_QUADRANT(0, 0, w0f)
            PP_QUADRANT(0, 1, w1f)
            PP_BAR
            // ---- phase B: X half 1 -> quadrants (1,1) (1,0); stages XH0 WH0 WH1 of the K tile after next
            load_x(base, 1);
            PP_STAGE(B, b, 0) PP_STAGE(B, b, 1) PP_STAGE(B, b, 2) PP_WAIT(B)
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            PP_BAR
            PP_QUADRANT(1, 1, w1f)
            PP_QUADRANT(1, 0, w0f)
            PP_BAR
            b ^= 1;
            if (C.kt == nk - 1) { epi = true; e_tm = C.tm; e_tn = C.tn; }
            if (!A.ok) break;
            C = A;
            A = B;
        }
        // last tile of this workgroup: group 0 first (its extra barrier is group 1's last one)
        epilogue(e_tm, e_tn);
        if (wm == 0) PP_BAR
#undef PP_WAIT
#undef PP_STAGE
    }

    // ---- the last, partial round: tiles [n_full, n_tiles) are cut into four quadrant tasks each
    // (64 rows per wave-row x 32 columns per wave-column = quadrant (mh, nh) of the tile), so that up to
    // 4x more CUs share it.  A task only needs the X / W half-tiles of its quadrant: four LDS slots
    // (the four half-tile positions of the two buffers) hold four K tiles, staged 3 ahead, ONE barrier
    // per K tile: slot (kq+3)&3 is refilled behind the barrier of iteration kq, after every wave has
    // retired its reads of it in iteration kq-1.
    const int n_tasks = (n_tiles - n_full) * 4;
    auto stage_task = [&](int slot, uint32_t xs, uint32_t ws, int kt) {
        unsigned char* dx = smem + (slot & 1) * 65536 + (slot >> 1) * 16384 + wave * 2048;
        unsigned char* dw = dx + 32768;
        const uint32_t sx = xs + x_wave + (uint32_t)kt * 128u, sw = ws + w_wave + (uint32_t)kt * 128u;
        glds16_buf_aux<MI_PP_X_AUX>(xr, x_lane, sx, dx);
        glds16_buf_aux<MI_PP_X_AUX>(xr, x_lane, sx + 8u * Kb, dx + 1024);
        glds16_buf_aux<MI_PP_W_AUX>(wr, x_lane, sw, dw);
        glds16_buf_aux<MI_PP_W_AUX>(wr, x_lane, sw + 8u * Kb, dw + 1024);
    };
    for (int task = lb; task < n_tasks; task += G) {
        const int t = n_full + (task >> 2), mh = (task >> 1) & 1, nh = task & 1;
        int ttm, ttn;
        tile_of(t, ttm, ttn);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        PP_BAR
        const uint32_t txs = (uint32_t)ttm * 256u * Kb + 64u * mh * Kb, tws = (uint32_t)ttn * 256u * Kb + 32u * nh * Kb;
        stage_aux(ttm, ttn);
#pragma unroll
        for (int k = 0; k < 3; ++k)
            if (k < nk) stage_task(k, txs, tws, k);
        for (int kq = 0; kq < nk; ++kq) {
            const int rem = nk - 1 - kq;  // K tiles issued behind this one and still allowed in flight: min(rem, 2)
            if (rem >= 2) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
            else if (rem == 1) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
            else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            PP_BAR
            if (kq + 3 < nk) stage_task((kq + 3) & 3, txs, tws, kq + 3);
            const unsigned char* base = smem + (kq & 1) * 65536;
            load_x(base, (kq >> 1) & 1); load_w(w0f, base, (kq >> 1) & 1);
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            PP_QUADRANT(0, 0, w0f)
        }
        __builtin_amdgcn_sched_barrier(0);
        v4f bq[2], cq[2];
#pragma unroll
        for (int ni = 0; ni < 2; ++ni) bq[ni] = *reinterpret_cast<const v4f*>(bias_lds + (32 * nh + ni * 16 + 4 * g) * 4);
        if constexpr (LNF) {
#pragma unroll
            for (int ni = 0; ni < 2; ++ni) cq[ni] = *reinterpret_cast<const v4f*>(bias_lds + 256 + (32 * nh + ni * 16 + 4 * g) * 4);
        }
        const uint32_t q_tile = ((uint32_t)ttm * 256u * (uint32_t)ldo + (uint32_t)ttn * 4u * cs) * 2u;
        const uint32_t q_lane = ((uint32_t)(wm * 128 + 64 * mh + (lane >> 2)) * (uint32_t)ldo + (uint32_t)wn * cs + (uint32_t)(32 * nh + 8 * (lane & 3))) * 2u;
        v4u hq[4];
        v2u lq[4];
        if constexpr (RES) {   // the old planes of the task's four 16-row steps
#pragma unroll
            for (int mi = 0; mi < 4; ++mi) {
                const uint32_t so = q_tile + (uint32_t)(mi * 16) * (uint32_t)ldo * 2u;
                hq[mi] = __builtin_amdgcn_raw_buffer_load_b128(orr, q_lane, so, MI_PP_RES_LOAD_AUX);
                lq[mi] = __builtin_amdgcn_raw_buffer_load_b64(lor, q_lane >> 1, so >> 1, MI_PP_RES_LOAD_AUX);
            }
        }
#pragma unroll
        for (int mi = 0; mi < 4; ++mi) {
            v2f st2;
            if constexpr (LNF) st2 = *reinterpret_cast<const v2f*>(bias_lds + 512 + (64 * mh + mi * 16 + l15) * 8);
#pragma unroll
            for (int ni = 0; ni < 2; ++ni) {
                v4f v;
                if constexpr (LNF) {
#pragma unroll
                    for (int c = 0; c < 4; ++c)
                        v[c] = __builtin_fmaf(st2.x, acc[ni][mi][c], __builtin_fmaf(st2.y, cq[ni][c], bq[ni][c]));
                } else {
                    v = acc[ni][mi] + bq[ni];
                }
                acc[ni][mi] = (v4f){0.0f, 0.0f, 0.0f, 0.0f};
                if constexpr (GELU) {
#pragma unroll
                    for (int c = 0; c < 4; ++c) v[c] = quick_gelu<true>(v[c]);
                }
                v2u pk;
                pk.x = pack2bf(v[0], v[1]);
                pk.y = pack2bf(v[2], v[3]);
                *reinterpret_cast<v2u*>(patch + l15 * 144 + (ni * 16 + 4 * g) * 2) = pk;
            }
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_sched_barrier(0);
            const v4u d = *reinterpret_cast<const v4u*>(patch + (lane >> 2) * 144 + (lane & 3) * 16);
            const uint32_t so = q_tile + (uint32_t)(mi * 16) * (uint32_t)ldo * 2u;
            if constexpr (RES) {
                v4u hn; v2u ln; float S, Q;
                resid24_step(hq[mi], lq[mi], d, hn, ln, S, Q);
                __builtin_amdgcn_sched_barrier(0);
                buffer_store_b128_aux<MI_PP_STORE_AUX_RES>(hn, orr, q_lane, so);
                __builtin_amdgcn_raw_buffer_store_b64(ln, lor, q_lane >> 1, so >> 1, MI_PP_STORE_AUX_RES);
                S += dpp_movf<0xB1>(S); Q += dpp_movf<0xB1>(Q);
                S += dpp_movf<0x4E>(S); Q += dpp_movf<0x4E>(Q);
                __builtin_amdgcn_sched_barrier(0);
                asm volatile("" :: "v"(hn), "v"(ln));   // the store's data registers stay allocated past the sums (see the whole-tile epilogue)
                // the quad's four lanes hold the block's sums: lane 0 of the quad stores them
                const uint32_t prow = (uint32_t)(ttm * 256 + wm * 128 + 64 * mh + mi * 16 + (lane >> 2));
                const v2u pv = {__float_as_uint(S), __float_as_uint(Q)};
                if (t4 == 0) __builtin_amdgcn_raw_buffer_store_b64(pv, pr, (prow * (uint32_t)nslot + (uint32_t)(ttn * 8 + wn * 2 + nh)) * 8u, 0, 0);
            } else {
                buffer_store_b128_aux<STORE_NT ? (GELU ? MI_PP_STORE_AUX_H : MI_PP_STORE_AUX_QKV) : 0>(d, orr, q_lane, so);
            }
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_sched_barrier(0);
        }
    }
    PP_CLOCK_END
#undef PP_QUADRANT
#undef PP_BAR
}

// ------------------------------------------------------------------ attention, fp32 (parity path)
// softmax(q k^T / 8) v per (image, head), one thread per query row, keys streamed
// through LDS in chunks of 64, running max / sum (modeling_clip.py:259-277).
// causal != 0: key j is visible to query i iff j <= i (the text tower's mask).
// q_blocks > 0: only the first q_blocks blocks of 64 queries of every (image, head) are computed (the
// last layer needs the CLS row only).
__global__ __launch_bounds__(64) void attn_f32_kernel(const float* __restrict__ qkv, float* __restrict__ ctx, int S,
                                                      int D, int H, int causal, int q_blocks) {
    __shared__ __attribute__((aligned(16))) float Ks[64][64];
    __shared__ __attribute__((aligned(16))) float Vs[64][64];
    const int tid = threadIdx.x;
    const int nqt = q_blocks > 0 ? q_blocks : (S + 63) / 64;
    const int qt = blockIdx.x % nqt, bh = blockIdx.x / nqt;
    const int b = bh / H, hh = bh % H;
    const size_t ld = (size_t)3 * D;
    const float* base = qkv + (size_t)b * S * ld + hh * 64;
    const int qi = qt * 64 + tid;
    const int qc = qi < S ? qi : S - 1;
    float q[64], o[64];
#pragma unroll
    for (int d = 0; d < 64; d += 4) {
        const v4f t = *reinterpret_cast<const v4f*>(base + (size_t)qc * ld + d);
        q[d] = t.x * 0.125f; q[d + 1] = t.y * 0.125f; q[d + 2] = t.z * 0.125f; q[d + 3] = t.w * 0.125f;
    }
#pragma unroll
    for (int d = 0; d < 64; ++d) o[d] = 0.0f;
    float mx = -INFINITY, l = 0.0f;
    for (int k0 = 0; k0 < S; k0 += 64) {
        __syncthreads();
#pragma unroll 4
        for (int i = 0; i < 16; ++i) {
            const int idx = tid + 64 * i, row = idx >> 4, c4 = (idx & 15) * 4;
            v4f kv = {0, 0, 0, 0}, vv = {0, 0, 0, 0};
            if (k0 + row < S) {
                kv = *reinterpret_cast<const v4f*>(base + (size_t)(k0 + row) * ld + D + c4);
                vv = *reinterpret_cast<const v4f*>(base + (size_t)(k0 + row) * ld + 2 * D + c4);
            }
            *reinterpret_cast<v4f*>(&Ks[row][c4]) = kv;
            *reinterpret_cast<v4f*>(&Vs[row][c4]) = vv;
        }
        __syncthreads();
        int nk = min(64, S - k0);
        if (causal) nk = min(nk, qc - k0 + 1);  // keys up to the query's own position
        for (int j = 0; j < nk; ++j) {
            float s = 0.0f;
#pragma unroll
            for (int d = 0; d < 64; ++d) s = __builtin_fmaf(q[d], Ks[j][d], s);
            if (s > mx) {
                const float c = expf(mx - s);
                l *= c;
#pragma unroll
                for (int d = 0; d < 64; ++d) o[d] *= c;
                mx = s;
            }
            const float pw = expf(s - mx);
            l += pw;
#pragma unroll
            for (int d = 0; d < 64; ++d) o[d] = __builtin_fmaf(pw, Vs[j][d], o[d]);
        }
    }
    if (qi < S) {
        float* dst = ctx + ((size_t)b * S + qi) * D + hh * 64;
#pragma unroll
        for (int d = 0; d < 64; d += 4) {
            v4f t = {o[d] / l, o[d + 1] / l, o[d + 2] / l, o[d + 3] / l};
            *reinterpret_cast<v4f*>(dst + d) = t;
        }
    }
}

// max over the four 16-lane rows of a wave (lanes l, l^16, l^32, l^48), VALU only: v_permlane16_swap /
// v_permlane32_swap exchange rows between two copies of the value (swap16: rows 1<->0 and 3<->2 of
// (a, b); swap32: rows 2,3 <-> 0,1), so max(a', b') holds max(own, partner) in every lane -- no LDS
// crossbar round trip (ds_bpermute) on the softmax's critical path.
__device__ __forceinline__ float xmax_rows(float v) {
    const unsigned u = __float_as_uint(v);
    const auto a = __builtin_amdgcn_permlane16_swap(u, u, false, false);
    v = fmaxf(__uint_as_float(a[0]), __uint_as_float(a[1]));
    const unsigned w = __float_as_uint(v);
    const auto b = __builtin_amdgcn_permlane32_swap(w, w, false, false);
    return fmaxf(__uint_as_float(b[0]), __uint_as_float(b[1]));
}

// ------------------------------------------------------------------ attention, fp32 on the matrix pipe (parity path, S <= 272)
// The one-thread-per-query kernel above was 21 % of the fp32 forward at b = 256 for 4 % of its FLOPs (3.6 ms per layer).
// Same softmax, exact-f32 MFMA 16x16x4: one workgroup of 8 waves per (image, head), K and V of the head as fp32 rows in LDS
// (pitch 68 floats: 16 consecutive rows start in 16 different bank quads), a wave per 16-query tile.
//   S^T = K Q^T:  A = K[key 16T + l15][16g + kk], B = q[query l15][16g + kk] (q already times 1/8) over kk = 0..15 — the
//                 MFMA's k index is a dummy, so lane group g takes the CONTIGUOUS dims 16g .. 16g+15 (four ds_read_b128
//                 per key tile) instead of the strided ones;  sc[T][e] = s[key 16T + 4g + e][query l15]
//   softmax:      max / sum over a lane's 4 * NKT scores, then over the four lane groups (xmax_rows / xsum_rows)
//   O^T = V^T P^T: for the same reason key slot (e, g) of an MFMA means key 16T + 4g + e, so the B operand is the lane's OWN
//                 p[T][e] — P never moves — and A = V[16T + 4g + e][16dt + l15];  o[dt][e'] = O[query l15][16dt + 4g + e']
// causal / q_tiles as attn_f32_kernel (q_tiles counts 16-query tiles here).
__device__ __forceinline__ float xsum_rows(float v) {
    const unsigned u = __float_as_uint(v);
    const auto a = __builtin_amdgcn_permlane16_swap(u, u, false, false);
    v = __uint_as_float(a[0]) + __uint_as_float(a[1]);
    const unsigned w = __float_as_uint(v);
    const auto b = __builtin_amdgcn_permlane32_swap(w, w, false, false);
    return __uint_as_float(b[0]) + __uint_as_float(b[1]);
}
constexpr int ATTNF_PITCH = 68;
constexpr int attnf_lds_bytes(int s_pad) { return 2 * s_pad * ATTNF_PITCH * 4; }
template <int S_PAD>
__global__ __launch_bounds__(512) void attn_f32_mfma_kernel(const float* __restrict__ qkv, float* __restrict__ ctx, int S, int D, int H,
                                                            int causal, int q_tiles) {
    constexpr int NKT = S_PAD / 16;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    float* Ks = reinterpret_cast<float*>(smem);
    float* Vs = Ks + S_PAD * ATTNF_PITCH;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int g = lane >> 4, l15 = lane & 15;
    const int b = blockIdx.x / H, hh = blockIdx.x % H;
    const size_t ld = (size_t)3 * D;
    const float* base = qkv + (size_t)b * S * ld + hh * 64;
    for (int idx = tid; idx < S_PAD * 16; idx += 512) {
        const int row = idx >> 4, c4 = (idx & 15) * 4;
        v4f kv = {0.0f, 0.0f, 0.0f, 0.0f}, vv = {0.0f, 0.0f, 0.0f, 0.0f};
        if (row < S) {
            kv = *reinterpret_cast<const v4f*>(base + (size_t)row * ld + D + c4);
            vv = *reinterpret_cast<const v4f*>(base + (size_t)row * ld + 2 * D + c4);
        }
        *reinterpret_cast<v4f*>(Ks + row * ATTNF_PITCH + c4) = kv;
        *reinterpret_cast<v4f*>(Vs + row * ATTNF_PITCH + c4) = vv;
    }
    __syncthreads();
    const int nkt_all = (S + 15) / 16;
    const int nqt = q_tiles > 0 ? min(q_tiles, nkt_all) : nkt_all;
    for (int qt = wave; qt < nqt; qt += 8) {
        const int qi = qt * 16 + l15;
        const int qc = qi < S ? qi : S - 1;
        const int nkt = causal ? min(nkt_all, qt + 1) : nkt_all;   // causal: key tiles beyond the diagonal one are all masked
        float qf[16];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const v4f t = *reinterpret_cast<const v4f*>(base + (size_t)qc * ld + 16 * g + 4 * i);
            qf[4 * i] = t.x * 0.125f; qf[4 * i + 1] = t.y * 0.125f; qf[4 * i + 2] = t.z * 0.125f; qf[4 * i + 3] = t.w * 0.125f;
        }
        v4f sc[NKT];
#pragma unroll
        for (int T = 0; T < NKT; ++T) {
            sc[T] = (v4f){0.0f, 0.0f, 0.0f, 0.0f};
            if (T < nkt) {
                float kf[16];
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const v4f t = *reinterpret_cast<const v4f*>(Ks + (16 * T + l15) * ATTNF_PITCH + 16 * g + 4 * i);
                    kf[4 * i] = t.x; kf[4 * i + 1] = t.y; kf[4 * i + 2] = t.z; kf[4 * i + 3] = t.w;
                }
#pragma unroll
                for (int kk = 0; kk < 16; ++kk) sc[T] = __builtin_amdgcn_mfma_f32_16x16x4f32(kf[kk], qf[kk], sc[T], 0, 0, 0);
            }
        }
        float mx = -INFINITY;
#pragma unroll
        for (int T = 0; T < NKT; ++T) {
            if (T < nkt) {
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const int key = 16 * T + 4 * g + e;
                    if (key >= S || (causal && key > qc)) sc[T][e] = -INFINITY;
                    mx = fmaxf(mx, sc[T][e]);
                }
            }
        }
        mx = xmax_rows(mx);
        float l = 0.0f;
#pragma unroll
        for (int T = 0; T < NKT; ++T) {
            if (T < nkt) {
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const float pw = expf(sc[T][e] - mx);   // exp(-inf) = 0 for the masked keys
                    sc[T][e] = pw;
                    l += pw;
                }
            }
        }
        l = xsum_rows(l);
        v4f o[4];
#pragma unroll
        for (int dt = 0; dt < 4; ++dt) o[dt] = (v4f){0.0f, 0.0f, 0.0f, 0.0f};
#pragma unroll
        for (int T = 0; T < NKT; ++T) {
            if (T < nkt) {
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const float* vrow = Vs + (16 * T + 4 * g + e) * ATTNF_PITCH + l15;
#pragma unroll
                    for (int dt = 0; dt < 4; ++dt) o[dt] = __builtin_amdgcn_mfma_f32_16x16x4f32(vrow[16 * dt], sc[T][e], o[dt], 0, 0, 0);
                }
            }
        }
        if (qi < S) {
            float* dst = ctx + ((size_t)b * S + qi) * D + hh * 64 + 4 * g;
#pragma unroll
            for (int dt = 0; dt < 4; ++dt) {
                const v4f t = {o[dt][0] / l, o[dt][1] / l, o[dt][2] / l, o[dt][3] / l};
                *reinterpret_cast<v4f*>(dst + 16 * dt) = t;
            }
        }
    }
}


// ------------------------------------------------------------------ attention, bf16 MFMA
// One workgroup per (image, head); K and V of the head ([S_PAD][64] bf16, 128-byte
// rows, 16-byte chunks XOR-swizzled with row&7) stay in LDS; each wave walks query
// tiles of 16 rows.  Scores are computed TRANSPOSED (S^T = K Q^T) so that a lane
// owns one query column: softmax is an in-lane reduction plus two cross-lane steps,
// and the exponentiated tile is already the A operand of P V (k-slot (g,e) of
// step s = key 32s + 16(e>>2) + 4g + (e&3)); V is consumed through
// ds_read_b64_tr_b16 so the same row-major image serves as the B operand.
// Softmax runs in the exp2 domain with the 1/8 scale folded in (one fma + one v_exp
// per score); only key tiles that hold a key >= S are masked, tiles entirely beyond S
// are never computed.  P V is computed transposed too, with the V columns permuted across the
// MFMA rows, so each lane stores 2 x 16 contiguous bytes of its own query's context row; the
// row sums come from a sixth MFMA against a fragment of ones.
// S <= S_PAD, S_PAD % 32 == 0; the whole score row of a query lives in registers.
// (hipcc pitfall: assembling the bf16x8 B fragment element by element from the transposed
//  read's v4i16 result miscompiled to a splat of element 0 — use the v4bf16 builtin and
//  __builtin_shufflevector.)
// S_CT > 0: the token count is a compile-time constant (launcher checks S == S_CT), so every
// "does this key tile exist / straddle S" test folds away; S_CT == 0 keeps them at run time.
//
// NQ query tiles (of 16 rows) are processed TOGETHER by a wave: with two workgroups per CU there
// are only two waves per SIMD and the chain MFMA -> max -> exp -> pack -> MFMA is latency-bound
// (measured: no overlap between the two resident waves), so the second tile supplies the
// independent instructions; the K and V fragments read from LDS serve both tiles.
// emit(u, query row, d0, d1): the context row of tile u's query l15 as bf16 — d0 = head dims 8g .. 8g + 7, d1 = 32 + 8g ..
// (which is also the B-operand fragment pair of an MFMA 16x16x32 over the head's 64 dims: text_attn_out_kernel)
template <int S_PAD, int S_CT, int NQ, class Emit>
__device__ __forceinline__ void attn_tiles_epi(const unsigned char* __restrict__ Ks, const unsigned char* __restrict__ Vs,
                                               const bf16_t* __restrict__ base, size_t ld, int S_rt, const int (&qt)[NQ], int lane,
                                               int causal, Emit&& emit) {
    constexpr int NKT = S_PAD / 16, NPV = S_PAD / 32;
    const int S = S_CT > 0 ? S_CT : S_rt;
    const int g = lane >> 4, l15 = lane & 15;
    const int nkt = (S + 15) / 16;
    constexpr float C2 = 0.125f * 1.4426950408889634f;  // scale * log2(e)

    bf16x8 qf[NQ][2];
#pragma unroll
    for (int u = 0; u < NQ; ++u) {
        const int qi = qt[u] * 16 + l15;
        const int qc = qi < S ? qi : S - 1;
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) qf[u][ks] = *reinterpret_cast<const bf16x8*>(base + (size_t)qc * ld + 32 * ks + 8 * g);
    }

    // S^T = K Q^T; K fragments run KPF key tiles ahead of their MFMAs
    constexpr int KPF = 3;
    v4f sc[NQ][NKT];
    bf16x8 kfr[KPF][2];
    auto load_k = [&](bf16x8 (&dst)[2], int T) {
        const int row = 16 * T + l15;
#pragma unroll
        for (int ks = 0; ks < 2; ++ks)
            dst[ks] = *reinterpret_cast<const bf16x8*>(Ks + row * 128 + (((4 * ks + g) ^ (row & 7)) << 4));
    };
#pragma unroll
    for (int T = 0; T < KPF; ++T)
        if (T < NKT && T < nkt) load_k(kfr[T], T);
#pragma unroll
    for (int T = 0; T < NKT; ++T) {
#pragma unroll
        for (int u = 0; u < NQ; ++u) sc[u][T] = (v4f){0.0f, 0.0f, 0.0f, 0.0f};
        if (T < nkt) {
#pragma unroll
            for (int ks = 0; ks < 2; ++ks)
#pragma unroll
                for (int u = 0; u < NQ; ++u)
                    sc[u][T] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(kfr[T % KPF][ks], qf[u][ks], sc[u][T], 0, 0, 0);
            if (T + KPF < NKT && T + KPF < nkt) load_k(kfr[T % KPF], T + KPF);
        }
    }
    // sc[u][T][e] = (K Q^T)[key 16T + 4g + e][query l15 of tile u]
    bf16x8 pa[NQ][NPV];
#pragma unroll
    for (int u = 0; u < NQ; ++u) {
        float mx = -INFINITY;
#pragma unroll
        for (int T = 0; T < NKT; ++T) {
            if (T < nkt) {
                if (16 * T + 16 > S) {
#pragma unroll
                    for (int e = 0; e < 4; ++e)
                        if (16 * T + 4 * g + e >= S) sc[u][T][e] = -INFINITY;
                }
                if (causal) {  // the text tower's mask: key j is visible to query i iff j <= i
#pragma unroll
                    for (int e = 0; e < 4; ++e)
                        if (16 * T + 4 * g + e > qt[u] * 16 + l15) sc[u][T][e] = -INFINITY;
                }
                mx = fmaxf(mx, fmaxf(fmaxf(sc[u][T][0], sc[u][T][1]), fmaxf(sc[u][T][2], sc[u][T][3])));
            }
        }
        mx = xmax_rows(mx);
        const float mc = -mx * C2;
#pragma unroll
        for (int s = 0; s < NPV; ++s)
#pragma unroll
            for (int half = 0; half < 2; ++half) {
                const int T = 2 * s + half;
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    float pw = 0.0f;
                    if (T < nkt) pw = __builtin_amdgcn_exp2f(__builtin_fmaf(sc[u][T][e], C2, mc));
                    pa[u][s][4 * half + e] = (__bf16)pw;
                }
            }
    }
    // O^T = V^T P^T with permuted V columns (see the kernel header); row sums from a fragment of ones
    bf16x8 ones;
#pragma unroll
    for (int e = 0; e < 8; ++e) ones[e] = (__bf16)1.0f;
    v4f o[NQ][4], osum[NQ];
#pragma unroll
    for (int u = 0; u < NQ; ++u) {
        osum[u] = (v4f){0.0f, 0.0f, 0.0f, 0.0f};
#pragma unroll
        for (int dt = 0; dt < 4; ++dt) o[u][dt] = (v4f){0.0f, 0.0f, 0.0f, 0.0f};
    }
    const int tq = l15 >> 2, tp = l15 & 3;
    bf16x8 vfr[2][4];
    auto load_v = [&](bf16x8 (&dst)[4], int s) {
        const int key0 = 32 * s + 4 * g + tq, key1 = key0 + 16;
#pragma unroll
        for (int dt = 0; dt < 4; ++dt) {
            const int c = 4 * (dt >> 1) + tp, sub = (dt & 1) * 8;
            const bf16x4 v0 = __builtin_amdgcn_ds_read_tr16_b64_v4bf16(
                (__attribute__((address_space(3))) bf16x4*)(Vs + key0 * 128 + ((c ^ (key0 & 7)) << 4) + sub));
            const bf16x4 v1 = __builtin_amdgcn_ds_read_tr16_b64_v4bf16(
                (__attribute__((address_space(3))) bf16x4*)(Vs + key1 * 128 + ((c ^ (key1 & 7)) << 4) + sub));
            dst[dt] = __builtin_shufflevector(v0, v1, 0, 1, 2, 3, 4, 5, 6, 7);
        }
    };
    load_v(vfr[0], 0);
#pragma unroll
    for (int s = 0; s < NPV; ++s) {
        if (2 * s < nkt) {
            if (s + 1 < NPV && 2 * (s + 1) < nkt) load_v(vfr[(s + 1) & 1], s + 1);
#pragma unroll
            for (int dt = 0; dt < 4; ++dt)
#pragma unroll
                for (int u = 0; u < NQ; ++u)
                    o[u][dt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(vfr[s & 1][dt], pa[u][s], o[u][dt], 0, 0, 0);
#pragma unroll
            for (int u = 0; u < NQ; ++u)
                osum[u] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ones, pa[u][s], osum[u], 0, 0, 0);
        }
    }
    // o[u][dt][e] = O[query l15][d = 32*(dt>>1) + 8g + 4*(dt&1) + e]; osum[u][*] = sum_key P[query l15][key]
#pragma unroll
    for (int u = 0; u < NQ; ++u) {
        const int qi = qt[u] * 16 + l15;
        const float ie = __builtin_amdgcn_rcpf(osum[u][0]);
        v4u d[2];
#pragma unroll
        for (int w = 0; w < 2; ++w) {
            d[w].x = pack2bf(o[u][2 * w][0] * ie, o[u][2 * w][1] * ie);
            d[w].y = pack2bf(o[u][2 * w][2] * ie, o[u][2 * w][3] * ie);
            d[w].z = pack2bf(o[u][2 * w + 1][0] * ie, o[u][2 * w + 1][1] * ie);
            d[w].w = pack2bf(o[u][2 * w + 1][2] * ie, o[u][2 * w + 1][3] * ie);
        }
        emit(u, qi, d[0], d[1]);
    }
}
template <int S_PAD, int S_CT, int NQ>
__device__ __forceinline__ void attn_tiles(const unsigned char* __restrict__ Ks, const unsigned char* __restrict__ Vs,
                                           const bf16_t* __restrict__ base, bf16_t* __restrict__ ctx_b, size_t ld,
                                           int S_rt, int D, const int (&qt)[NQ], int lane, int causal = 0) {
    const int S = S_CT > 0 ? S_CT : S_rt;
    const int g = lane >> 4;
    attn_tiles_epi<S_PAD, S_CT, NQ>(Ks, Vs, base, ld, S_rt, qt, lane, causal, [&](int, int qi, v4u d0, v4u d1) {
        if (qi < S) {
            bf16_t* dst = ctx_b + (size_t)qi * D + 8 * g;
            *reinterpret_cast<v4u*>(dst) = d0;
            *reinterpret_cast<v4u*>(dst + 32) = d1;
        }
    });
}

// Two query tiles A, B software-pipelined inside one wave, written in the order the wave should
// issue it (in-order issue: the VALU work placed between MFMAs runs while the matrix pipe works):
//   QK(A)  |  QK(B) tile by tile  with  exp/pack(A) tile by tile  |  PV(A) step by step  with
//   exp/pack(B) two tiles per step  |  PV(B)
// K and V fragments come from LDS once per use (no sharing between A and B in this order, but
// the live registers stay near 200 instead of spilling at 256).
template <int S_PAD, int S_CT>
__device__ __forceinline__ void attn_pair(const unsigned char* __restrict__ Ks, const unsigned char* __restrict__ Vs,
                                          const bf16_t* __restrict__ base, bf16_t* __restrict__ ctx_b, size_t ld,
                                          int S_rt, int D, int qtA, int qtB, int lane, int causal = 0) {
    constexpr int NKT = S_PAD / 16, NPV = S_PAD / 32;
    const int S = S_CT > 0 ? S_CT : S_rt;
    const int g = lane >> 4, l15 = lane & 15;
    const int nkt = (S + 15) / 16;
    constexpr float C2 = 0.125f * 1.4426950408889634f;
    const int tq = l15 >> 2, tp = l15 & 3;

    auto load_q = [&](bf16x8 (&q)[2], int qt) {
        const int qi = qt * 16 + l15;
        const int qc = qi < S ? qi : S - 1;
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) q[ks] = *reinterpret_cast<const bf16x8*>(base + (size_t)qc * ld + 32 * ks + 8 * g);
    };
    auto load_k = [&](bf16x8 (&dst)[2], int T) {
        const int row = 16 * T + l15;
#pragma unroll
        for (int ks = 0; ks < 2; ++ks)
            dst[ks] = *reinterpret_cast<const bf16x8*>(Ks + row * 128 + (((4 * ks + g) ^ (row & 7)) << 4));
    };
    auto load_v = [&](bf16x8 (&dst)[4], int s) {
        const int key0 = 32 * s + 4 * g + tq, key1 = key0 + 16;
#pragma unroll
        for (int dt = 0; dt < 4; ++dt) {
            const int c = 4 * (dt >> 1) + tp, sub = (dt & 1) * 8;
            const bf16x4 v0 = __builtin_amdgcn_ds_read_tr16_b64_v4bf16(
                (__attribute__((address_space(3))) bf16x4*)(Vs + key0 * 128 + ((c ^ (key0 & 7)) << 4) + sub));
            const bf16x4 v1 = __builtin_amdgcn_ds_read_tr16_b64_v4bf16(
                (__attribute__((address_space(3))) bf16x4*)(Vs + key1 * 128 + ((c ^ (key1 & 7)) << 4) + sub));
            dst[dt] = __builtin_shufflevector(v0, v1, 0, 1, 2, 3, 4, 5, 6, 7);
        }
    };
    // masked row maximum of a finished score tile set (in-lane + the two cross-lane steps)
    auto row_max = [&](v4f (&sc)[NKT], int qt) {
        float mx = -INFINITY;
#pragma unroll
        for (int T = 0; T < NKT; ++T) {
            if (T < nkt) {
                if (16 * T + 16 > S) {
#pragma unroll
                    for (int e = 0; e < 4; ++e)
                        if (16 * T + 4 * g + e >= S) sc[T][e] = -INFINITY;
                }
                if (causal) {
#pragma unroll
                    for (int e = 0; e < 4; ++e)
                        if (16 * T + 4 * g + e > qt * 16 + l15) sc[T][e] = -INFINITY;
                }
                mx = fmaxf(mx, fmaxf(fmaxf(sc[T][0], sc[T][1]), fmaxf(sc[T][2], sc[T][3])));
            }
        }
        mx = xmax_rows(mx);
        return -mx * C2;
    };
    auto exp_pack4 = [&](const v4f& sc, float mc, bool live, bf16x8& dst, int half) {
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            float pw = 0.0f;
            if (live) pw = __builtin_amdgcn_exp2f(__builtin_fmaf(sc[e], C2, mc));
            dst[4 * half + e] = (__bf16)pw;
        }
    };
    auto store_o = [&](const v4f (&o)[4], const v4f& osum, int qt) {
        const int qi = qt * 16 + l15;
        const float ie = __builtin_amdgcn_rcpf(osum[0]);
        if (qi < S) {
            bf16_t* dst = ctx_b + (size_t)qi * D + 8 * g;
#pragma unroll
            for (int w = 0; w < 2; ++w) {
                v4u d;
                d.x = pack2bf(o[2 * w][0] * ie, o[2 * w][1] * ie);
                d.y = pack2bf(o[2 * w][2] * ie, o[2 * w][3] * ie);
                d.z = pack2bf(o[2 * w + 1][0] * ie, o[2 * w + 1][1] * ie);
                d.w = pack2bf(o[2 * w + 1][2] * ie, o[2 * w + 1][3] * ie);
                *reinterpret_cast<v4u*>(dst + 32 * w) = d;
            }
        }
    };

    bf16x8 qa[2], qb[2];
    load_q(qa, qtA);
    load_q(qb, qtB);
    constexpr int KPF = 3;
    bf16x8 kfr[KPF][2];
    v4f scA[NKT], scB[NKT];
    // ---- QK(A): key tiles in groups of KPF; within a group all k-step-0 MFMAs, then all k-step-1
    // (a tile's two MFMAs chain through its accumulator: issued back to back they run at half rate)
#pragma unroll
    for (int T = 0; T < KPF; ++T)
        if (T < NKT && T < nkt) load_k(kfr[T], T);
#pragma unroll
    for (int T0 = 0; T0 < NKT; T0 += KPF) {
#pragma unroll
        for (int ks = 0; ks < 2; ++ks)
#pragma unroll
            for (int t = 0; t < KPF; ++t) {
                const int T = T0 + t;
                if (T < NKT) {
                    if (ks == 0) scA[T] = (v4f){0.0f, 0.0f, 0.0f, 0.0f};
                    if (T < nkt) scA[T] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(kfr[t][ks], qa[ks], scA[T], 0, 0, 0);
                }
            }
#pragma unroll
        for (int t = 0; t < KPF; ++t)
            if (T0 + KPF + t < NKT && T0 + KPF + t < nkt) load_k(kfr[t], T0 + KPF + t);
    }
    const float mcA = row_max(scA, qtA);
    __builtin_amdgcn_sched_barrier(0);
    // ---- QK(B) tile by tile, exp/pack(A) in between
    bf16x8 paA[NPV], paB[NPV];
#pragma unroll
    for (int T = 0; T < KPF; ++T)
        if (T < NKT && T < nkt) load_k(kfr[T], T);
#pragma unroll
    for (int T0 = 0; T0 < NKT; T0 += KPF) {
#pragma unroll
        for (int ks = 0; ks < 2; ++ks)
#pragma unroll
            for (int t = 0; t < KPF; ++t) {
                const int T = T0 + t;
                if (T < NKT) {
                    if (ks == 0) scB[T] = (v4f){0.0f, 0.0f, 0.0f, 0.0f};
                    if (T < nkt) scB[T] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(kfr[t][ks], qb[ks], scB[T], 0, 0, 0);
                }
                // exp/pack of A's matching tile rides between the MFMAs
                if (ks == 1 && T < NKT) exp_pack4(scA[T], mcA, T < nkt, paA[T >> 1], T & 1);
            }
#pragma unroll
        for (int t = 0; t < KPF; ++t)
            if (T0 + KPF + t < NKT && T0 + KPF + t < nkt) load_k(kfr[t], T0 + KPF + t);
        __builtin_amdgcn_sched_barrier(0);  // keep the written interleave: hipcc otherwise regroups and spills
    }
    const float mcB = row_max(scB, qtB);
    __builtin_amdgcn_sched_barrier(0);
    // ---- PV(A) step by step, exp/pack(B) in between
    bf16x8 ones;
#pragma unroll
    for (int e = 0; e < 8; ++e) ones[e] = (__bf16)1.0f;
    bf16x8 vfr[2][4];
    v4f oA[4], oB[4], sumA = {0.0f, 0.0f, 0.0f, 0.0f}, sumB = {0.0f, 0.0f, 0.0f, 0.0f};
#pragma unroll
    for (int dt = 0; dt < 4; ++dt) { oA[dt] = (v4f){0.0f, 0.0f, 0.0f, 0.0f}; oB[dt] = (v4f){0.0f, 0.0f, 0.0f, 0.0f}; }
    load_v(vfr[0], 0);
#pragma unroll
    for (int s = 0; s < NPV; ++s) {
        if (2 * s < nkt) {
            if (s + 1 < NPV && 2 * (s + 1) < nkt) load_v(vfr[(s + 1) & 1], s + 1);
#pragma unroll
            for (int dt = 0; dt < 4; ++dt)
                oA[dt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(vfr[s & 1][dt], paA[s], oA[dt], 0, 0, 0);
            sumA = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ones, paA[s], sumA, 0, 0, 0);
        }
        exp_pack4(scB[2 * s], mcB, 2 * s < nkt, paB[s], 0);
        exp_pack4(scB[2 * s + 1], mcB, 2 * s + 1 < nkt, paB[s], 1);
        __builtin_amdgcn_sched_barrier(0);
    }
    store_o(oA, sumA, qtA);
    __builtin_amdgcn_sched_barrier(0);
    // ---- PV(B)
    load_v(vfr[0], 0);
#pragma unroll
    for (int s = 0; s < NPV; ++s) {
        if (2 * s < nkt) {
            if (s + 1 < NPV && 2 * (s + 1) < nkt) load_v(vfr[(s + 1) & 1], s + 1);
#pragma unroll
            for (int dt = 0; dt < 4; ++dt)
                oB[dt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(vfr[s & 1][dt], paB[s], oB[dt], 0, 0, 0);
            sumB = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ones, paB[s], sumB, 0, 0, 0);
        }
    }
    store_o(oB, sumB, qtB);
}

template <int S_PAD, int S_CT>
__global__ __launch_bounds__(256, 2) __attribute__((amdgpu_waves_per_eu(2, 2))) void attn_bf16_kernel(
    const bf16_t* __restrict__ qkv, bf16_t* __restrict__ ctx, int S_rt, int D, int H, int q_tiles, int causal) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    unsigned char* Ks = smem;
    unsigned char* Vs = smem + S_PAD * 128;
    const int S = S_CT > 0 ? S_CT : S_rt;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int b = blockIdx.x / H, hh = blockIdx.x % H;
    const size_t ld = (size_t)3 * D;
    const bf16_t* base = qkv + (size_t)b * S * ld + hh * 64;

    // K and V go HBM -> LDS by LDS-DMA, 8 rows (1 KiB) per wave-instruction; the image is linear,
    // the XOR swizzle is applied to the per-lane source chunk; rows >= S lie beyond the
    // descriptor's num_records, so the hardware range check fills them with zeros.
    {
        const rsrc_t kvr = make_rsrc(base, (uint32_t)((size_t)S * ld * 2 - (size_t)hh * 128));
        const int rr = lane >> 3, p = lane & 7;
        const uint32_t voff = (uint32_t)rr * (uint32_t)(ld * 2) + 16u * (uint32_t)(p ^ rr);
        for (int j = __builtin_amdgcn_readfirstlane(wave); j < S_PAD / 8; j += 4) {
            const uint32_t so = (uint32_t)(8 * j) * (uint32_t)(ld * 2);
            glds16_buf(kvr, voff, so + (uint32_t)D * 2u, Ks + j * 1024);
            glds16_buf(kvr, voff, so + (uint32_t)D * 4u, Vs + j * 1024);
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    __syncthreads();

    // query tiles of this wave: w', w'+4, ... with the start rotated between co-resident workgroups;
    // they are taken two at a time, a last odd one alone
    // q_tiles > 0: only that many leading query tiles (the last layer needs the CLS row = row 0 only)
    const int nqt = q_tiles > 0 ? min(q_tiles, (S + 15) / 16) : (S + 15) / 16;
    bf16_t* ctx_b = ctx + (size_t)b * S * D + hh * 64;
    int q0 = (wave + blockIdx.x) & 3;
    for (; q0 + 4 < nqt; q0 += 8) {
        attn_pair<S_PAD, S_CT>(Ks, Vs, base, ctx_b, ld, S_rt, D, q0, q0 + 4, lane, causal);
    }
    if (q0 < nqt) {
        const int one[1] = {q0};
        attn_tiles<S_PAD, S_CT, 1>(Ks, Vs, base, ctx_b, ld, S_rt, D, one, lane, causal);
    }
}

// ------------------------------------------------------------------ one text query: attention + out_proj in one launch
// forward_text_one's attention followed by its out_proj: the context rows of a head come out of attn_tiles_epi as the very
// B-operand fragments an MFMA over the head's 64 dims wants, so out_proj's contribution of that head,
//   slab[h][row][n] = sum_{d < 64} ctx_h[row][d] * Wo[n][64 h + d]      (fp32, SKINNY_ROWS rows per slab)
// needs no memory round trip; ln_slab_kernel adds the H slabs in head order (+ bias) into the residual stream.
// grid = (H, N / (16 * TAO_NT)): every workgroup redoes its head's (tiny: 77 x 77 x 64) attention and owns TAO_NT
// 16-column tiles of the output, whose weight fragments are requested before anything else.
constexpr int TAO_NT = 12, TAO_WAVES = 5;   // a wave per 16-query tile: 80 positions in one sweep (four waves left one of them two tiles)
template <int S_PAD>
__global__ __launch_bounds__(64 * TAO_WAVES) void text_attn_out_kernel(const bf16_t* __restrict__ qkv, const bf16_t* __restrict__ Wo,
                                                            float* __restrict__ slabs, int S, int D, int causal) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    unsigned char* Ks = smem;
    unsigned char* Vs = smem + S_PAD * 128;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int hh = blockIdx.x, n0 = blockIdx.y * 16 * TAO_NT;
    const int g = lane >> 4, l15 = lane & 15;
    const size_t ld = (size_t)3 * D;
    const bf16_t* base = qkv + hh * 64;
    bf16x8 afr[TAO_NT][2];
#pragma unroll
    for (int nt = 0; nt < TAO_NT; ++nt)
#pragma unroll
        for (int ks = 0; ks < 2; ++ks)
            afr[nt][ks] = *reinterpret_cast<const bf16x8*>(Wo + (size_t)(n0 + 16 * nt + l15) * D + hh * 64 + 32 * ks + 8 * g);
    {
        const rsrc_t kvr = make_rsrc(base, (uint32_t)((size_t)S * ld * 2 - (size_t)hh * 128));
        const int rr = lane >> 3, p = lane & 7;
        const uint32_t voff = (uint32_t)rr * (uint32_t)(ld * 2) + 16u * (uint32_t)(p ^ rr);
        for (int j = __builtin_amdgcn_readfirstlane(wave); j < S_PAD / 8; j += TAO_WAVES) {
            const uint32_t so = (uint32_t)(8 * j) * (uint32_t)(ld * 2);
            glds16_buf(kvr, voff, so + (uint32_t)D * 2u, Ks + j * 1024);
            glds16_buf(kvr, voff, so + (uint32_t)D * 4u, Vs + j * 1024);
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    __syncthreads();
    float* slab = slabs + (size_t)hh * SKINNY_ROWS * D + n0;
    for (int q0 = wave; q0 < (S + 15) / 16; q0 += TAO_WAVES) {
        const int one[1] = {q0};
        attn_tiles_epi<S_PAD, 0, 1>(Ks, Vs, base, ld, S, one, lane, causal, [&](int, int, v4u d0, v4u d1) {
            const bf16x8 b0 = __builtin_bit_cast(bf16x8, d0), b1 = __builtin_bit_cast(bf16x8, d1);
#pragma unroll
            for (int nt = 0; nt < TAO_NT; ++nt) {
                v4f acc = (v4f){0.0f, 0.0f, 0.0f, 0.0f};
                acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(afr[nt][0], b0, acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(afr[nt][1], b1, acc, 0, 0, 0);
                // acc[e] = slab[row q0 * 16 + l15][n0 + 16 nt + 4 g + e]; rows S .. SKINNY_ROWS - 1 hold the padding queries' values, never read
                *reinterpret_cast<v4f*>(slab + (size_t)(q0 * 16 + l15) * D + 16 * nt + 4 * g) = acc;
            }
        });
    }
}

}  // namespace mi
