// Dev harness: times the bf16 attention kernels at ViT-L/14 b = 256 (n*H = 4096 workgroups) on random data:
//   attn32_bf16_kernel<288,257,true> (32-query tiles, attn32_kernels.h) and attn_bf16_kernel<288,257> (16-query tiles).
// Build: hipcc -O3 --offload-arch=gfx950 -std=c++17 -ffp-contract=off tools/probe/attn_bench.hip -o tools/probe/attn_bench
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include "../../include/mi355clip.h"
#include "../../image_search_amd/csrc/vit_kernels.h"
#ifdef ATTN32_STAMPS
#include "attn_stamps.h"   // defines the kernel's timing hooks; the library build leaves them empty
#endif
#include "../../image_search_amd/csrc/attn32_kernels.h"
using namespace mi;
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e), __FILE__, __LINE__); exit(1); } } while (0)
__global__ void fill_bf16(bf16_t* p, size_t n, uint64_t seed, float scale) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        uint64_t z = (i + seed) * 0x9E3779B97F4A7C15ull; z ^= z >> 29; z *= 0xBF58476D1CE4E5B9ull; z ^= z >> 32;
        p[i] = f2bf(((int)(z & 0xffff) - 32768) / 32768.0f * scale);
    }
}
template <class F>
float time_us(F launch, int reps) {
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (int it = 0; it < 3; ++it) launch();
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(e0));
    for (int it = 0; it < reps; ++it) launch();
    CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    return ms / reps * 1000;
}
int main(int argc, char** argv) {
    const int n = argc > 1 ? atoi(argv[1]) : 256, S = 257, H = argc > 2 ? atoi(argv[2]) : 16, D = 64 * H;
    const size_t M = (size_t)n * S;
    bf16_t *qkv, *ctx;
    CK(hipMalloc(&qkv, (M + 256) * 3 * D * 2)); CK(hipMalloc(&ctx, (M + 256) * D * 2));
    hipLaunchKernelGGL(fill_bf16, 2048, 256, 0, 0, qkv, M * 3 * D, 1, 1.0f);
#ifdef ATTN32_STAMPS   // the stamped kernel writes its buffer on EVERY launch: it must exist before the first one
    unsigned long long* d_st; CK(hipMalloc(&d_st, 256 * 8 * 8 * 8)); CK(hipMemset(d_st, 0, 256 * 8 * 8 * 8));
    CK(hipMemcpyToSymbol(HIP_SYMBOL(attn32_stamp_buf), &d_st, sizeof d_st));
#endif
    {
        auto kern = attn32_bf16_kernel<288, 257, true>;
        constexpr int LDS = attn32_lds_bytes(288);
        const int pairs = n * H, grid = pairs < 256 ? pairs : 256;
        CK(hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, LDS));
        for (int shift = 0; shift < 2; ++shift) {
            const float us = time_us([&] { hipLaunchKernelGGL(kern, dim3(grid), dim3(512), LDS, 0, qkv, ctx, S, D, H, pairs, 0, shift, 1, 3 * D, D, 64u, (uint32_t)D); }, 20);
            printf("attn32<288,257> shift=%d n=%d: %.1f us per launch  (%.0f TFLOP/s, %.2f TB/s of q,k,v,ctx)\n", shift, n, us,
                   4.0 * S * S * 64 * H * n / us * 1e-6, (double)M * 4 * D * 2 / us * 1e-6);
        }
#ifdef ATTN32_STAMPS
        {
            CK(hipMemset(d_st, 0, 256 * 8 * 8 * 8));
            hipLaunchKernelGGL(kern, dim3(grid), dim3(512), LDS, 0, qkv, ctx, S, D, H, pairs, 0, 0, 1, 3 * D, D, 64u, (uint32_t)D);
            CK(hipDeviceSynchronize());
            static unsigned long long h[256 * 8 * 8];
            CK(hipMemcpy(h, d_st, sizeof h, hipMemcpyDeviceToHost));
            const char* names[5] = {"own loads landed", "barrier", "issue next", "whole tile", "split tile"};
            for (int w : {0, 3, 7}) {
                printf("wave %d, mean shader cycles per pair over %d workgroups:", w, grid);
                for (int j = 0; j < 5; ++j) {
                    double sum = 0; for (int b = 0; b < grid; ++b) sum += (double)h[((size_t)b * 8 + w) * 8 + j];
                    printf("  %s %.0f", names[j], sum / grid / (pairs / grid));
                }
                printf("\n");
            }
        }
#endif
        const float us1 = time_us([&] { hipLaunchKernelGGL(kern, dim3(grid), dim3(512), LDS, 0, qkv, ctx, S, D, H, pairs, 1, 0, 1, 3 * D, D, 64u, (uint32_t)D); }, 20);
        printf("attn32<288,257> first tile only (last layer): %.1f us\n", us1);
    }
    {
        auto kern = attn_bf16_kernel<288, 257>;
        CK(hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, 288 * 256));
        const float us = time_us([&] { hipLaunchKernelGGL(kern, dim3(n * H), dim3(256), 288 * 256, 0, qkv, ctx, S, D, H, 0, 0); }, 20);
        printf("attn_bf16<288,257> (16-query tiles) n=%d: %.1f us per launch\n", n, us);
    }
    return 0;
}
