// Dev harness: times the bf16 GEMM variants on the ViT-L/14 b=256 shapes and checks them
// against each other.  Build: hipcc -O3 --offload-arch=gfx950 -std=c++17 tools/probe/gemm_bench.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <algorithm>
#include <cmath>
#include "../../include/mi355clip.h"
#include "../../image_search_amd/csrc/vit_kernels.h"
using namespace mi;
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e), __FILE__, __LINE__); exit(1); } } while (0)

__global__ void fill_bf16(bf16_t* p, size_t n, uint64_t seed, float scale) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        uint64_t z = (i + seed) * 0x9E3779B97F4A7C15ull; z ^= z >> 29; z *= 0xBF58476D1CE4E5B9ull; z ^= z >> 32;
        float v = ((int)(z & 0xffff) - 32768) / 32768.0f * scale;
        p[i] = f2bf(v);
    }
}
__global__ void mismatch(const bf16_t* a, const bf16_t* b, size_t n, int N, unsigned long long* cnt, unsigned long long* first) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x)
        if (a[i] != b[i]) { unsigned long long k = atomicAdd(cnt, 1ull); atomicMin(first, (unsigned long long)i); if (k < 4096) first[1 + k] = i; }
}
__global__ void maxdiff(const bf16_t* a, const bf16_t* b, size_t n, float* out) {
    float m = 0;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x)
        m = fmaxf(m, fabsf(bf2f(a[i]) - bf2f(b[i])));
    atomicMax((int*)out, __float_as_int(m));
}


// PROBE: the 128x128 two-barrier kernel on MFMA 32x32x16 instead of 16x16x32 (same tile, staging, barriers and bytes of
// LDS traffic; half the MFMA instructions for the same matrix-pipe time) — does the instruction shape matter by itself?
// LDS image: chunk c of row r at c ^ swz32(r) (conflict-free for 32-row b128 reads), swizzle on the DMA source address.
__device__ __forceinline__ int swz32p(int row) { return (((row >> 1) & 1) << 2) | ((row >> 2) & 3); }
template <int EPI, typename TO>
__global__ __launch_bounds__(256, 2) void gemm32_probe_kernel(const bf16_t* __restrict__ X, const bf16_t* __restrict__ W,
                                                              const float* __restrict__ bias, void* __restrict__ out,
                                                              int N, int K, int ldo) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 1, wn = wave & 1, h = lane >> 5, l31 = lane & 31;
    const int nt = N / 128;
    const uint32_t wg = xcd_remap(blockIdx.x, gridDim.x);
    const int tm = wg / nt, tn = wg % nt;
    const size_t m0 = (size_t)tm * 128;
    const int n0 = tn * 128;
    const int rr = lane >> 3, p = lane & 7;
    const size_t row8 = (size_t)8 * K * 2;
    // rows 32 wave + 8 j + rr: swz32 of that row depends on rr and on j & 1
    const unsigned char* xs[2];
    const unsigned char* ws[2];
    for (int par = 0; par < 2; ++par) {
        const int sw = swz32p(8 * par + rr);
        xs[par] = reinterpret_cast<const unsigned char*>(X) + ((m0 + 32 * wave + rr) * K + 8 * (p ^ sw)) * 2;
        ws[par] = reinterpret_cast<const unsigned char*>(W) + (((size_t)n0 + 32 * wave + rr) * K + 8 * (p ^ sw)) * 2;
    }
    auto stage = [&](int buf, int kt) {
        unsigned char* xb = smem + buf * 32768 + wave * 4096;
        unsigned char* wb = xb + 16384;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            glds16(xs[j & 1] + j * row8 + (size_t)kt * 128, xb + j * 1024);
            glds16(ws[j & 1] + j * row8 + (size_t)kt * 128, wb + j * 1024);
        }
    };
    v16f acc[2][2];
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int c = 0; c < 2; ++c)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[a][c][e] = 0.0f;
    const int sw = swz32p(l31);  // fragment rows are (multiple of 32) + l31
    const int a_off = (wn * 64 + l31) * 128, b_off = (wm * 64 + l31) * 128;
    auto compute = [&](int buf) {
        const unsigned char* xb = smem + buf * 32768;
        const unsigned char* wb = xb + 16384;
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {
            const int ch = ((2 * ks + h) ^ sw) << 4;
            bf16x8 a[2], b[2];
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                a[i] = *reinterpret_cast<const bf16x8*>(wb + a_off + i * 4096 + ch);
                b[i] = *reinterpret_cast<const bf16x8*>(xb + b_off + i * 4096 + ch);
            }
#pragma unroll
            for (int ni = 0; ni < 2; ++ni)
#pragma unroll
                for (int mi = 0; mi < 2; ++mi)
                    acc[ni][mi] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[ni], b[mi], acc[ni][mi], 0, 0, 0);
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
    // C[i = n][j = m]: lane holds token column j = l31, weight rows i = (e & 3) + 8 (e >> 2) + 4 h
#pragma unroll
    for (int ni = 0; ni < 2; ++ni)
#pragma unroll
        for (int mi = 0; mi < 2; ++mi)
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const size_t m = m0 + wm * 64 + mi * 32 + l31;
                const int n = n0 + wn * 64 + ni * 32 + 8 * q + 4 * h;
                v4f v = {acc[ni][mi][4 * q], acc[ni][mi][4 * q + 1], acc[ni][mi][4 * q + 2], acc[ni][mi][4 * q + 3]};
                epilogue4<EPI, TO, true>(v, bias, out, m, n, ldo);
            }
}

template <class F> float time_ms(F f, int iters) {
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (int i = 0; i < 3; ++i) f();
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(e0));
    for (int i = 0; i < iters; ++i) f();
    CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    return ms / iters;
}

int main(int argc, char** argv) {
    const size_t M = 65792;
    struct Shape { int N, K; const char* name; } shapes[] = {{3072, 1024, "qkv"}, {1024, 1024, "out"}, {4096, 1024, "fc1"}, {1024, 4096, "fc2"}};
    bf16_t *X, *W, *O1, *O2; float *bias, *d;
    CK(hipMalloc(&X, M * 4096 * 2)); CK(hipMalloc(&W, (size_t)4096 * 4096 * 2));
    CK(hipMalloc(&O1, M * 4096 * 2)); CK(hipMalloc(&O2, M * 4096 * 2));
    CK(hipMalloc(&bias, 4096 * 4)); CK(hipMalloc(&d, 4)); CK(hipMemset(bias, 0, 4096 * 4));
    hipLaunchKernelGGL(fill_bf16, 2048, 256, 0, 0, X, M * 4096, 1, 1.0f);
    hipLaunchKernelGGL(fill_bf16, 2048, 256, 0, 0, W, (size_t)4096 * 4096, 7, 0.05f);
    CK(hipFuncSetAttribute((const void*)gemm_bf16_kernel<EPI_BIAS, bf16_t>, hipFuncAttributeMaxDynamicSharedMemorySize, 65536));
    CK(hipFuncSetAttribute((const void*)gemm32_probe_kernel<EPI_BIAS, bf16_t>, hipFuncAttributeMaxDynamicSharedMemorySize, 65536));
    CK(hipFuncSetAttribute((const void*)gemm_bf16_pp_kernel<EPI_BIAS, bf16_t>, hipFuncAttributeMaxDynamicSharedMemorySize, 151552));
    for (auto& s : shapes) {
        const double flop = 2.0 * M * s.N * s.K;
        auto k128 = [&] { hipLaunchKernelGGL((gemm_bf16_kernel<EPI_BIAS, bf16_t>), dim3((M / 128) * (s.N / 128)), dim3(256), 65536, 0, X, W, bias, (void*)O1, s.N, s.K, s.N); };
        const int n_tiles = (int)((M / 256) * (s.N / 256));
        auto kper = [&] { hipLaunchKernelGGL((gemm32_probe_kernel<EPI_BIAS, bf16_t>), dim3((M / 128) * (s.N / 128)), dim3(256), 65536, 0, X, W, bias, (void*)O2, s.N, s.K, s.N); };
        const int n_full_ = n_tiles - ((n_tiles % 256) * 4 <= 256 ? n_tiles % 256 : 0);
        auto kper2 = [&] { hipLaunchKernelGGL((gemm_bf16_pp_kernel<EPI_BIAS, bf16_t>), dim3(256), dim3(512), 151552, 0, X, W, bias, (void*)O2, (int)M, s.N, s.K, s.N, n_tiles, n_full_, 0); };
        float t1 = time_ms(k128, 10);
        CK(hipMemset(O2, 0, M * s.N * 2));
        float t4 = time_ms(kper, 10);
        CK(hipMemset(d, 0, 4));
        hipLaunchKernelGGL(maxdiff, 1024, 256, 0, 0, O1, O2, M * s.N, d);
        float md32; CK(hipMemcpy(&md32, d, 4, hipMemcpyDeviceToHost));
        CK(hipMemset(O2, 0, M * s.N * 2));
        float t5 = time_ms(kper2, 10);
        // race screen: argv[1] repetitions of {clear, run once, compare with the 128x128 kernel's output}
        for (int rep = 0; rep < (argc > 1 ? atoi(argv[1]) : 0); ++rep) {
            CK(hipMemset(O2, 0xff, M * s.N * 2));
            kper2();
            CK(hipMemset(d, 0, 4));
            hipLaunchKernelGGL(maxdiff, 1024, 256, 0, 0, O1, O2, M * s.N, d);
            float mdr; CK(hipMemcpy(&mdr, d, 4, hipMemcpyDeviceToHost));
            if (mdr != 0.0f) printf("   RACE SCREEN: repetition %d of %s differs from the reference kernel (max diff %g)\n", rep, s.name, mdr);
        }
        CK(hipMemset(d, 0, 4));
        hipLaunchKernelGGL(maxdiff, 1024, 256, 0, 0, O1, O2, M * s.N, d);
        float md; CK(hipMemcpy(&md, d, 4, hipMemcpyDeviceToHost));
        {
            unsigned long long* c; CK(hipMalloc(&c, 16 + 4096 * 8)); unsigned long long init[2] = {0, ~0ull};
            CK(hipMemcpy(c, init, 16, hipMemcpyHostToDevice));
            hipLaunchKernelGGL(mismatch, 1024, 256, 0, 0, O1, O2, M * s.N, s.N, c, c + 1);
            unsigned long long h[2]; CK(hipMemcpy(h, c, 16, hipMemcpyDeviceToHost));
            if (h[0]) {
                std::vector<unsigned long long> idx(std::min<unsigned long long>(h[0], 4096));
                CK(hipMemcpy(idx.data(), c + 2, idx.size() * 8, hipMemcpyDeviceToHost));
                std::sort(idx.begin(), idx.end());
                unsigned long long prev = ~0ull; int run = 0;
                for (size_t q = 0; q <= idx.size(); ++q) {
                    if (q < idx.size() && prev != ~0ull && idx[q] == prev + 1) { ++run; prev = idx[q]; continue; }
                    if (prev != ~0ull) printf("     run of %d ending row %llu col %llu\n", run + 1, prev / s.N, prev % s.N);
                    if (q < idx.size()) { prev = idx[q]; run = 0; }
                }
            }
            if (h[0]) printf("   mismatches %llu first at row %llu col %llu (tile m %llu n %llu)\n", h[0], h[1] / s.N, h[1] % s.N, h[1] / s.N / 256, (h[1] % s.N) / 256);
            CK(hipFree(c));
        }
        printf("%s N=%d K=%d: 128^2 on 16x16x32 %.3f ms %.0f TF | 128^2 on 32x32x16 %.3f ms %.0f TF (max diff %.3g) | pp %.3f ms %.0f TF | maxdiff(pp vs 128) %.4g\n", s.name, s.N, s.K, t1, flop / t1 / 1e9, t4, flop / t4 / 1e9, md32, t5, flop / t5 / 1e9, md);
    }
    return 0;
}
