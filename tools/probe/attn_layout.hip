// Dev harness: what the LAYOUT of q|k|v costs attn32_bf16_kernel<288,257,true> at ViT-L/14 (random data; the arithmetic is
// the same in all three):
//   rows      token rows [M][3 D]           (6 144-byte pitch: the layout of rounds 1-4)
//   rows+pad  token rows [M][3 D + 128]     (the tower's default since round 5, option "qkv_pad")
//   planes    head-major [3][H][Mp][64]     (option "qkv_layout" = 1: a head's K / V / q of one image is one contiguous block)
// Each form: back-to-back rate (20 launches) and isolated launches (a sync in front of every one), three rounds alternating.
// Build: hipcc -O3 --offload-arch=gfx950 -std=c++17 -ffp-contract=off tools/probe/attn_layout.hip -o tools/probe/attn_layout
// Run:   tools/probe/attn_layout [images = 256] [heads = 16]
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include "../../include/mi355clip.h"
#include "../../image_search_amd/csrc/vit_kernels.h"
#include "../../image_search_amd/csrc/attn32_kernels.h"
using namespace mi;
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e), __FILE__, __LINE__); exit(1); } } while (0)
__global__ void fill_bf16(bf16_t* p, size_t n, uint64_t seed, float scale) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        uint64_t z = (i + seed) * 0x9E3779B97F4A7C15ull; z ^= z >> 29; z *= 0xBF58476D1CE4E5B9ull; z ^= z >> 32;
        p[i] = f2bf(((int)(z & 0xffff) - 32768) / 32768.0f * scale);
    }
}
int main(int argc, char** argv) {
    const int n = argc > 1 ? atoi(argv[1]) : 256, S = 257, H = argc > 2 ? atoi(argv[2]) : 16, D = 64 * H;
    const size_t M = (size_t)n * S, Mp = (M + 255) / 256 * 256;
    bf16_t *qkv, *ctx;
    const size_t elems = Mp * (size_t)(3 * D + 128);
    CK(hipMalloc(&qkv, elems * 2)); CK(hipMalloc(&ctx, Mp * D * 2));
    hipLaunchKernelGGL(fill_bf16, 2048, 256, 0, 0, qkv, elems, 1, 1.0f);
    auto kern = attn32_bf16_kernel<288, 257, true>;
    constexpr int LDS = attn32_lds_bytes(288);
    const int pairs = n * H, grid = pairs < 256 ? pairs : 256;
    CK(hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, LDS));
    struct Form { const char* name; int ld; uint32_t hs, ss; };
    const Form forms[3] = {{"rows    ", 3 * D, 64u, (uint32_t)D},
                           {"rows+pad", 3 * D + 128, 64u, (uint32_t)D},
                           {"planes  ", 64, (uint32_t)(Mp * 64), (uint32_t)((size_t)H * Mp * 64)}};
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    auto launch = [&](const Form& f) { hipLaunchKernelGGL(kern, dim3(grid), dim3(512), LDS, 0, qkv, ctx, S, D, H, pairs, 0, 0, 1, f.ld, D, f.hs, f.ss); };
    printf("attn32<288,257> images=%d heads=%d pairs=%d grid=%d  (69.3 GFLOP, 539 MB per launch at 256 x 16)\n", n, H, pairs, grid);
    for (int round = 0; round < 3; ++round)
        for (const Form& f : forms) {
            for (int it = 0; it < 3; ++it) launch(f);
            CK(hipDeviceSynchronize());
            CK(hipEventRecord(e0));
            for (int it = 0; it < 20; ++it) launch(f);
            CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
            float ms; CK(hipEventElapsedTime(&ms, e0, e1));
            const float rate = ms / 20 * 1000;
            std::vector<float> iso;
            for (int it = 0; it < 15; ++it) {
                CK(hipDeviceSynchronize());
                CK(hipEventRecord(e0)); launch(f); CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
                CK(hipEventElapsedTime(&ms, e0, e1));
                iso.push_back(ms * 1000);
            }
            std::sort(iso.begin(), iso.end());
            printf("round %d  %s  back to back %.1f us   isolated median %.1f us (min %.1f)   %.2f TB/s of q,k,v,ctx at the rate\n", round, f.name,
                   rate, iso[iso.size() / 2], iso[0], (double)M * 4 * D * 2 / rate * 1e-6);
        }
    return 0;
}
