// Why does attn32 read 162 us in the tower and 146-152 us alone?  The kernel is timed (HIP events around the
// attention launch only) and, in the -DATTN32_STAMPS build, stamped, in the states of the memory system it can meet:
//   warm     back to back on the same qkv / ctx buffers (what attn_bench.hip measures)
//   written  right behind a kernel that has just (re)written all of qkv with 16-byte stores, as the QKV GEMM's epilogue
//            does in the tower: the lines are dirty in the L2s / on their way to the Infinity Cache when attention starts
//   cold     behind a kernel that streamed 2 GB of other memory through the caches: qkv comes from HBM
//   written+ctx  `written`, and the ctx buffer was last touched by a reader (the LayerNorm output the QKV GEMM consumed)
// Build: hipcc -O3 --offload-arch=gfx950 -std=c++17 -ffp-contract=off [-DATTN32_STAMPS] tools/probe/attn_context.hip -o tools/probe/attn_context
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
__global__ void copy16(const v4u* __restrict__ src, v4u* __restrict__ dst, size_t n16) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n16; i += (size_t)gridDim.x * blockDim.x) dst[i] = src[i];
}
__global__ void read16(const v4u* __restrict__ src, size_t n16, unsigned* __restrict__ sink) {
    unsigned acc = 0;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n16; i += (size_t)gridDim.x * blockDim.x) acc ^= src[i].x;
    if (acc == 0x12345678u) *sink = acc;
}
int main(int argc, char** argv) {
    const int n = argc > 1 ? atoi(argv[1]) : 256, S = 257, H = 16, D = 64 * H;
    const size_t M = (size_t)n * S, qkv_bytes = (M + 256) * 3 * D * 2, ctx_bytes = (M + 256) * D * 2, big = (size_t)2 << 30;
    bf16_t *qkv, *qkv_src, *ctx; char *other, *other2; unsigned* sink;
    CK(hipMalloc(&qkv, qkv_bytes)); CK(hipMalloc(&qkv_src, qkv_bytes)); CK(hipMalloc(&ctx, ctx_bytes));
    CK(hipMalloc(&other, big)); CK(hipMalloc(&other2, big)); CK(hipMalloc(&sink, 4));
    hipLaunchKernelGGL(fill_bf16, 2048, 256, 0, 0, qkv_src, M * 3 * D, 1, 1.0f);
    hipLaunchKernelGGL(copy16, 2048, 256, 0, 0, (const v4u*)qkv_src, (v4u*)qkv, qkv_bytes / 16);
    CK(hipMemset(other, 1, big));
#ifdef ATTN32_STAMPS
    unsigned long long* d_st; CK(hipMalloc(&d_st, 256 * 8 * 8 * 8)); CK(hipMemset(d_st, 0, 256 * 8 * 8 * 8));
    CK(hipMemcpyToSymbol(HIP_SYMBOL(attn32_stamp_buf), &d_st, sizeof d_st));
    static unsigned long long h[256 * 8 * 8];
#endif
    auto kern = attn32_bf16_kernel<288, 257, true>;
    constexpr int LDS = attn32_lds_bytes(288);
    const int pairs = n * H, grid = pairs < 256 ? pairs : 256;
    CK(hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, LDS));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    const char* cond[4] = {"warm", "written", "cold", "written+ctx"};
    for (int c = 0; c < 4; ++c) {
        double us = 0;
        const int reps = 12;
#ifdef ATTN32_STAMPS
        double seg[3][5] = {};
#endif
        for (int it = 0; it < reps + 2; ++it) {
            if (c == 1 || c == 3) hipLaunchKernelGGL(copy16, 2048, 256, 0, 0, (const v4u*)qkv_src, (v4u*)qkv, qkv_bytes / 16);
            if (c == 2) hipLaunchKernelGGL(copy16, 2048, 256, 0, 0, (const v4u*)other, (v4u*)other2, big / 16);
            if (c == 3) hipLaunchKernelGGL(read16, 2048, 256, 0, 0, (const v4u*)ctx, ctx_bytes / 16, sink);
            CK(hipEventRecord(e0));
            hipLaunchKernelGGL(kern, dim3(grid), dim3(512), LDS, 0, qkv, ctx, S, D, H, pairs, 0, 0, 1, 3 * D, D, 64u, (uint32_t)D);
            CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
            float ms; CK(hipEventElapsedTime(&ms, e0, e1));
            if (it >= 2) us += ms * 1000;
#ifdef ATTN32_STAMPS
            if (it >= 2) {
                CK(hipMemcpy(h, d_st, sizeof h, hipMemcpyDeviceToHost));
                const int ws[3] = {0, 3, 7};
                for (int wi = 0; wi < 3; ++wi)
                    for (int j = 0; j < 5; ++j) {
                        double sum = 0; for (int b = 0; b < grid; ++b) sum += (double)h[((size_t)b * 8 + ws[wi]) * 8 + j];
                        seg[wi][j] += sum / grid / (pairs / grid);
                    }
            }
#endif
        }
        printf("%-12s attn32<288,257> n=%d: %.1f us per launch\n", cond[c], n, us / reps);
#ifdef ATTN32_STAMPS
        const char* names[5] = {"own loads landed", "barrier", "issue next", "whole tile", "split tile"};
        const int ws[3] = {0, 3, 7};
        for (int wi = 0; wi < 3; ++wi) {
            printf("   wave %d, shader cycles per pair:", ws[wi]);
            for (int j = 0; j < 5; ++j) printf("  %s %.0f", names[j], seg[wi][j] / reps);
            printf("\n");
        }
#endif
    }
    return 0;
}
