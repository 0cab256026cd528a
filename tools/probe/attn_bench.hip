// Dev harness: times attn_bf16_kernel<288,257> at b=256 and ablations of it (compile-time -DABL=n).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include "../../include/mi355clip.h"
#include "../../image_search_amd/csrc/vit_kernels.h"
using namespace mi;
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e), __FILE__, __LINE__); exit(1); } } while (0)
__global__ void fill_bf16(bf16_t* p, size_t n, uint64_t seed, float scale) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        uint64_t z = (i + seed) * 0x9E3779B97F4A7C15ull; z ^= z >> 29; z *= 0xBF58476D1CE4E5B9ull; z ^= z >> 32;
        p[i] = f2bf(((int)(z & 0xffff) - 32768) / 32768.0f * scale);
    }
}
int main() {
    const int n = 256, S = 257, D = 1024, H = 16;
    const size_t M = (size_t)n * S;
    bf16_t *qkv, *ctx;
    CK(hipMalloc(&qkv, (M + 256) * 3 * D * 2)); CK(hipMalloc(&ctx, (M + 256) * D * 2));
    hipLaunchKernelGGL(fill_bf16, 2048, 256, 0, 0, qkv, M * 3 * D, 1, 1.0f);
    auto kern = attn_bf16_kernel<288, 257>;
    CK(hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, 288 * 256));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (int it = 0; it < 3; ++it) hipLaunchKernelGGL(kern, dim3(n * H), dim3(256), 288 * 256, 0, qkv, ctx, S, D, H, 0);
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(e0));
    for (int it = 0; it < 20; ++it) hipLaunchKernelGGL(kern, dim3(n * H), dim3(256), 288 * 256, 0, qkv, ctx, S, D, H, 0);
    CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    printf("ABL=%d: %.1f us per launch\n", ABL, ms / 20 * 1000);
    return 0;
}
