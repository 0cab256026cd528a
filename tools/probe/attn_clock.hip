// Dev harness: the shader clock the chip holds INSIDE attn32_bf16_kernel, and a workgroup's busy time, at ViT-L/14 b = 256
// (MI355X_MICROARCH.md 'DVFS give-back' item 6: d(s_memtime) / d(s_memrealtime) x 100 MHz around the whole kernel, after
// >= 2 s of back-to-back launches on random data).  The kernel's three timing hooks (empty in the library) are defined here.
//   ./attn_clock [n=256] [soak seconds=2] [pattern: any third argument] [pair order=1] [qkv row padding=0] [ctx row padding=0]
//   hipcc -O3 --offload-arch=gfx950 -std=c++17 -ffp-contract=off tools/probe/attn_clock.hip -o tools/probe/attn_clock
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <algorithm>
#include <vector>
#include "../../include/mi355clip.h"
#include "../../image_search_amd/csrc/vit_kernels.h"
__device__ unsigned long long attn_clk[256 * 2];
#define ATTN32_STAMP
#ifdef CLK_SEGMENTS   // also the five segments of an iteration, per wave (as attn_stamps.h): 0 own loads landed, 1 barrier, 2 issue next, 3 whole tile, 4 split tile
__device__ unsigned long long attn_seg[256 * 8 * 8];
#define ATTN32_STAMP_BEGIN const unsigned long long c0_ = __builtin_amdgcn_s_memtime(), r0_ = __builtin_amdgcn_s_memrealtime(); unsigned long long acc_[5] = {0, 0, 0, 0, 0}, last_ = c0_;
#define ATTN32_STAMP(SLOT) { const unsigned long long now_ = __builtin_amdgcn_s_memtime(); acc_[SLOT] += now_ - last_; last_ = __builtin_amdgcn_s_memtime(); }
#define ATTN32_STAMP_END if (threadIdx.x == 0) { attn_clk[blockIdx.x * 2] = __builtin_amdgcn_s_memtime() - c0_; attn_clk[blockIdx.x * 2 + 1] = __builtin_amdgcn_s_memrealtime() - r0_; } \
    if (lane == 0) for (int j = 0; j < 5; ++j) attn_seg[((size_t)blockIdx.x * 8 + wave) * 8 + j] = acc_[j];
#else
#define ATTN32_STAMP_BEGIN const unsigned long long c0_ = __builtin_amdgcn_s_memtime(), r0_ = __builtin_amdgcn_s_memrealtime();
#define ATTN32_STAMP(SLOT)
#define ATTN32_STAMP_END if (threadIdx.x == 0) { attn_clk[blockIdx.x * 2] = __builtin_amdgcn_s_memtime() - c0_; attn_clk[blockIdx.x * 2 + 1] = __builtin_amdgcn_s_memrealtime() - r0_; }
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
int main(int argc, char** argv) {
    const int n = argc > 1 ? atoi(argv[1]) : 256, S = 257, H = 16, D = 64 * H;
    const double soak_s = argc > 2 ? atof(argv[2]) : 2.0;
    const int order = argc > 4 ? atoi(argv[4]) : 1;
    const int ATTN32_LD_PAD = argc > 5 ? atoi(argv[5]) : 0, CTX_PAD = argc > 6 ? atoi(argv[6]) : 0;   // row padding of qkv / ctx in elements
    const size_t M = (size_t)n * S;
    bf16_t *qkv, *ctx;
    CK(hipMalloc(&qkv, (M + 256) * (3 * D + ATTN32_LD_PAD) * 2)); CK(hipMalloc(&ctx, (M + 256) * (D + 512) * 2));
    hipLaunchKernelGGL(fill_bf16, 2048, 256, 0, 0, qkv, M * (3 * D + ATTN32_LD_PAD), 1, 1.0f);
    auto kern = attn32_bf16_kernel<288, 257, true>;
    constexpr int LDS = attn32_lds_bytes(288);
    const int pairs = n * H, grid = pairs < 256 ? pairs : 256;
    CK(hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, LDS));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    auto launch = [&] { hipLaunchKernelGGL(kern, dim3(grid), dim3(512), LDS, 0, qkv, ctx, S, D, H, pairs, 0, 0, order, 3 * D + ATTN32_LD_PAD, D + CTX_PAD, 64u, (uint32_t)D); };
    for (int i = 0; i < 3; ++i) launch();
    CK(hipDeviceSynchronize());
    const int reps = (int)(soak_s / 150e-6);
    CK(hipEventRecord(e0));
    for (int i = 0; i < reps; ++i) launch();
    CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    std::vector<unsigned long long> c(512);
    CK(hipMemcpyFromSymbol(c.data(), HIP_SYMBOL(attn_clk), 512 * 8));
    if (argc > 3) {   // the pattern: busy us by XCD (blockIdx % 8) and the raw list
        for (int x = 0; x < 8; ++x) { double a = 0, cy = 0; int k = 0; for (int b = x; b < grid; b += 8) { a += c[2 * b + 1] / 100.0; cy += (double)c[2 * b]; ++k; } printf("xcd-slot %d: mean busy %.1f us, %.0f shader cycles, clock %.3f GHz\n", x, a / k, cy / k, cy / a / 1e3); }
        if (order == 1) { printf("by head (slot %% 16):"); for (int hh = 0; hh < 16; ++hh) { double a = 0; int k = 0; for (int b = 0; b < grid; ++b) if ((b >> 3) % 16 == hh) { a += c[2 * b + 1] / 100.0; ++k; } printf(" %.0f", a / k); } printf("\n"); }
        for (int b = 0; b < grid; ++b) printf("%d:%.0f%s", b, c[2 * b + 1] / 100.0, (b % 16 == 15) ? "\n" : " ");
    }
#ifdef CLK_SEGMENTS
    {
        std::vector<unsigned long long> sg(256 * 8 * 8);
        CK(hipMemcpyFromSymbol(sg.data(), HIP_SYMBOL(attn_seg), sg.size() * 8));
        const char* nm[5] = {"own loads landed", "barrier", "issue next", "whole tile", "split tile"};
        for (int w = 0; w < 8; ++w) {
            printf("wave %d, mean shader cycles per pair:", w);
            double tot = 0;
            for (int j = 0; j < 5; ++j) { double a = 0; for (int b = 0; b < grid; ++b) a += (double)sg[((size_t)b * 8 + w) * 8 + j]; a /= grid * ((double)pairs / grid); tot += a; printf("  %s %.0f", nm[j], a); }
            printf("  | sum %.0f\n", tot);
        }
    }
#endif
    std::vector<double> ghz, busy, cyc;
    for (int b = 0; b < grid; ++b) { ghz.push_back(c[2 * b] / (double)c[2 * b + 1] * 0.1); busy.push_back(c[2 * b + 1] / 100.0); cyc.push_back((double)c[2 * b]); }
    std::sort(ghz.begin(), ghz.end()); std::sort(busy.begin(), busy.end()); std::sort(cyc.begin(), cyc.end());
    printf("attn32<288,257> qkv pad %d ctx pad %d order %d n=%d: %.1f us per launch over %d back-to-back launches; in-kernel clock median %.3f GHz (min %.3f max %.3f); "
           "workgroup busy median %.1f us (min %.1f max %.1f) = %.0f shader cycles = %.0f per pair\n",
           ATTN32_LD_PAD, CTX_PAD, order, n, ms / reps * 1e3, reps, ghz[grid / 2], ghz[0], ghz[grid - 1], busy[grid / 2], busy[0], busy[grid - 1], cyc[grid / 2],
           cyc[grid / 2] / ((double)pairs / grid));
    return 0;
}
