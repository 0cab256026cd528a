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
    CK(hipFuncSetAttribute((const void*)gemm_bf16_persist_kernel<EPI_BIAS, bf16_t>, hipFuncAttributeMaxDynamicSharedMemorySize, 151552));
    CK(hipFuncSetAttribute((const void*)gemm_bf16_pp_kernel<EPI_BIAS, bf16_t>, hipFuncAttributeMaxDynamicSharedMemorySize, 151552));
    for (auto& s : shapes) {
        const double flop = 2.0 * M * s.N * s.K;
        auto k128 = [&] { hipLaunchKernelGGL((gemm_bf16_kernel<EPI_BIAS, bf16_t>), dim3((M / 128) * (s.N / 128)), dim3(256), 65536, 0, X, W, bias, (void*)O1, s.N, s.K, s.N); };
        const int n_tiles = (int)((M / 256) * (s.N / 256));
        auto kper = [&] { hipLaunchKernelGGL((gemm_bf16_persist_kernel<EPI_BIAS, bf16_t>), dim3(256), dim3(512), 151552, 0, X, W, bias, (void*)O2, (int)M, s.N, s.K, s.N, n_tiles, n_tiles - ((n_tiles % 256) * 4 <= 256 ? n_tiles % 256 : 0), 0); };
        const int n_full_ = n_tiles - ((n_tiles % 256) * 4 <= 256 ? n_tiles % 256 : 0);
        auto kper2 = [&] { hipLaunchKernelGGL((gemm_bf16_pp_kernel<EPI_BIAS, bf16_t>), dim3(256), dim3(512), 151552, 0, X, W, bias, (void*)O2, (int)M, s.N, s.K, s.N, n_tiles, n_full_, 0); };
        float t1 = time_ms(k128, 10);
        CK(hipMemset(O2, 0, M * s.N * 2));
        float t4 = time_ms(kper, 10);
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
        printf("%s N=%d K=%d: 128^2 %.3f ms %.0f TF | persist %.3f ms %.0f TF | pp %.3f ms %.0f TF | maxdiff(pp vs 128) %.4g\n", s.name, s.N, s.K, t1, flop / t1 / 1e9, t4, flop / t4 / 1e9, t5, flop / t5 / 1e9, md);
    }
    return 0;
}
