// Dev harness: gemm_bf16_pp_kernel alone on the tower's four shapes at the HALF-chunk and the full-chunk row count, as a
// back-to-back rate (20 launches between two events) and as ISOLATED launches (one launch between two events, the stream
// drained around it, a 512 MB copy in between so that the operands come from HBM as they do in the tower).
//   hipcc -O3 --offload-arch=gfx950 -std=c++17 -ffp-contract=off tools/probe/gemm_pp_sweep.hip -o tools/probe/gemm_pp_sweep
//   ... -DPP_CLOCK -o tools/probe/gemm_pp_clock : the shader clock the chip holds under the kernel (random and all-zero operands)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <algorithm>
#include "../../include/mi355clip.h"
#ifdef PP_CLOCK   // diagnostic build: the shader clock held under the kernel = d(s_memtime) / d(s_memrealtime) x 100 MHz per workgroup
__device__ unsigned long long* pp_clock_buf;
#define PP_CLOCK_BEGIN const unsigned long long c0_ = __builtin_amdgcn_s_memtime(), r0_ = __builtin_amdgcn_s_memrealtime();
#define PP_CLOCK_END if (threadIdx.x == 0) { pp_clock_buf[blockIdx.x * 2] = __builtin_amdgcn_s_memtime() - c0_; pp_clock_buf[blockIdx.x * 2 + 1] = __builtin_amdgcn_s_memrealtime() - r0_; }
#endif
#include "../../image_search_amd/csrc/vit_kernels.h"
using namespace mi;
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e), __FILE__, __LINE__); exit(1); } } while (0)
__global__ void fill_bf16(bf16_t* p, size_t n, uint64_t seed, float scale) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        uint64_t z = (i + seed) * 0x9E3779B97F4A7C15ull; z ^= z >> 29; z *= 0xBF58476D1CE4E5B9ull; z ^= z >> 32;
        p[i] = f2bf(((int)(z & 0xffff) - 32768) / 32768.0f * scale);
    }
}
__global__ void checksum16(const unsigned short* p, size_t n, unsigned long long* out) {   // order-free: sum of (value * (index % 65521 + 1))
    unsigned long long a = 0;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) a += (unsigned long long)p[i] * (i % 65521 + 1);
    atomicAdd(out, a);
}
__global__ void copy16(const v4u* a, v4u* b, size_t n) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) b[i] = a[i];
}
int main(int argc, char** argv) {
    const int split = argc > 1 ? atoi(argv[1]) : 1;
    const int order_arg = argc > 2 ? atoi(argv[2]) : 0;   // n-tiles per column group (0 = row-major order)
    struct Shape { int N, K; const char* name; } shapes[] = {{3072, 1024, "qkv"}, {1024, 1024, "out"}, {4096, 1024, "fc1"}, {1024, 4096, "fc2"}};
    const size_t Mmax = 65792;
    bf16_t *X, *W, *O; float* bias; char *junk, *junk2;
    CK(hipMalloc(&X, Mmax * 4096 * 2)); CK(hipMalloc(&W, (size_t)4096 * 4096 * 2)); CK(hipMalloc(&O, Mmax * 4096 * 2));
    CK(hipMalloc(&bias, 4096 * 4)); CK(hipMemset(bias, 0, 4096 * 4));
    const size_t big = (size_t)512 << 20;
    CK(hipMalloc(&junk, big)); CK(hipMalloc(&junk2, big)); CK(hipMemset(junk, 1, big));
    hipLaunchKernelGGL(fill_bf16, 2048, 256, 0, 0, X, Mmax * 4096, 1, 1.0f);
    hipLaunchKernelGGL(fill_bf16, 2048, 256, 0, 0, W, (size_t)4096 * 4096, 7, 0.05f);
    auto k1 = gemm_bf16_pp_kernel<EPI_BIAS, bf16_t>;
    auto k2 = gemm_bf16_pp_kernel<EPI_BIAS_QGELU, bf16_t>;
    CK(hipFuncSetAttribute((const void*)k1, hipFuncAttributeMaxDynamicSharedMemorySize, 151552));
    CK(hipFuncSetAttribute((const void*)k2, hipFuncAttributeMaxDynamicSharedMemorySize, 151552));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
#ifdef PP_CLOCK
    // per shape: >= 2 s of back-to-back launches, then the clock of the LAST launch (median over workgroups), on the random
    // operands and on all-zero ones (the clock the chip would hold if the MFMAs cost no energy)
    unsigned long long* cbuf; CK(hipMalloc(&cbuf, 256 * 2 * 8));
    CK(hipMemcpyToSymbol(HIP_SYMBOL(pp_clock_buf), &cbuf, sizeof(cbuf)));
    for (int zeros = 0; zeros < 2; ++zeros) {
        if (zeros) { CK(hipMemset(X, 0, Mmax * 4096 * 2)); CK(hipMemset(W, 0, (size_t)4096 * 4096 * 2)); }
        for (size_t Mrows : {(size_t)32896, (size_t)65792}) {
            const size_t Mp = (Mrows + 255) / 256 * 256;
            for (auto& s : shapes) {
                const int n_tiles = (int)((Mp / 256) * (s.N / 256)), grid = std::min(n_tiles * 4, 256);
                const int left = n_tiles % grid;
                const int n_full = (split && left > 0 && left * 4 <= grid) ? n_tiles - left : n_tiles;
                auto launch = [&] {
                    if (s.N == 4096) hipLaunchKernelGGL(k2, dim3(grid), dim3(512), 151552, 0, X, W, bias, (void*)O, (int)Mp, s.N, s.K, s.N, n_tiles, n_full, (order_arg > 0 && (s.N / 256) % order_arg == 0) ? order_arg : 0);
                    else hipLaunchKernelGGL(k1, dim3(grid), dim3(512), 151552, 0, X, W, bias, (void*)O, (int)Mp, s.N, s.K, s.N, n_tiles, n_full, (order_arg > 0 && (s.N / 256) % order_arg == 0) ? order_arg : 0);
                };
                float total = 0, ms = 0; int launches = 0;
                while (total < 2000.0f) {
                    CK(hipEventRecord(e0));
                    for (int i = 0; i < 200; ++i) launch();
                    CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
                    CK(hipEventElapsedTime(&ms, e0, e1));
                    total += ms; launches += 200;
                }
                unsigned long long h[512]; CK(hipMemcpy(h, cbuf, sizeof(h), hipMemcpyDeviceToHost));
                double ghz[256]; for (int i = 0; i < grid; ++i) ghz[i] = (double)h[2 * i] / (double)h[2 * i + 1] * 0.1;
                std::sort(ghz, ghz + grid);
                const double us = ms / 200 * 1e3;
                printf("%s M=%zu %s: %.1f us = %.0f TFLOP/s, shader clock %.3f GHz (median of %d workgroups; min %.3f max %.3f) -> %.1f %% of the MFMA rate at THAT clock\n",
                       zeros ? "zeros " : "random", Mrows, s.name, us, 2.0 * Mrows * s.N * s.K / us / 1e6, ghz[grid / 2], grid, ghz[0], ghz[grid - 1],
                       100.0 * (2.0 * Mrows * s.N * s.K / us / 1e6) / (256 * 4 * 1024 * ghz[grid / 2] / 1e3));
            }
        }
    }
    return 0;
#endif
    for (size_t Mrows : {(size_t)32768, (size_t)32896, (size_t)65536, (size_t)65792}) {   // whole rounds of tiles (no tail) beside the tower's row counts
        const size_t Mp = (Mrows + 255) / 256 * 256;
        for (auto& s : shapes) {
            const double flop = 2.0 * Mrows * s.N * s.K;
            const int n_tiles = (int)((Mp / 256) * (s.N / 256)), grid = std::min(n_tiles * 4, 256);
            const int left = n_tiles % grid;
            const int n_full = (split && left > 0 && left * 4 <= grid) ? n_tiles - left : n_tiles;
            auto launch = [&] {
                if (s.N == 4096) hipLaunchKernelGGL(k2, dim3(grid), dim3(512), 151552, 0, X, W, bias, (void*)O, (int)Mp, s.N, s.K, s.N, n_tiles, n_full, (order_arg > 0 && (s.N / 256) % order_arg == 0) ? order_arg : 0);
                else hipLaunchKernelGGL(k1, dim3(grid), dim3(512), 151552, 0, X, W, bias, (void*)O, (int)Mp, s.N, s.K, s.N, n_tiles, n_full, (order_arg > 0 && (s.N / 256) % order_arg == 0) ? order_arg : 0);
            };
            for (int i = 0; i < 3; ++i) launch();
            CK(hipDeviceSynchronize());
            unsigned long long* dsum; unsigned long long hsum = 0; CK(hipMalloc(&dsum, 8)); CK(hipMemset(dsum, 0, 8));
            hipLaunchKernelGGL(checksum16, 1024, 256, 0, 0, (const unsigned short*)O, Mrows * (size_t)s.N, dsum);
            CK(hipMemcpy(&hsum, dsum, 8, hipMemcpyDeviceToHost)); CK(hipFree(dsum));
            CK(hipEventRecord(e0));
            for (int i = 0; i < 20; ++i) launch();
            CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
            float ms; CK(hipEventElapsedTime(&ms, e0, e1));
            const double rate_us = ms / 20 * 1e3;
            double iso_warm = 0, iso_cold = 0;
            for (int c = 0; c < 2; ++c) {
                double acc = 0;
                for (int i = 0; i < 8; ++i) {
                    if (c == 1) hipLaunchKernelGGL(copy16, 2048, 256, 0, 0, (const v4u*)junk, (v4u*)junk2, big / 16);
                    CK(hipDeviceSynchronize());
                    CK(hipEventRecord(e0)); launch(); CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
                    CK(hipEventElapsedTime(&ms, e0, e1));
                    if (i >= 2) acc += ms * 1e3 / 6;
                }
                (c ? iso_cold : iso_warm) = acc;
            }
            printf("[out checksum %016llx] ", hsum);
            printf("M=%zu %s N=%d K=%d tiles=%d (%.3f rounds, tail tasks %d): back-to-back %.1f us = %.0f TF | isolated warm %.1f us = %.0f TF | isolated behind 512 MB of other traffic %.1f us = %.0f TF\n",
                   Mrows, s.name, s.N, s.K, n_tiles, n_tiles / 256.0, (n_tiles - n_full) * 4, rate_us, flop / rate_us / 1e6, iso_warm, flop / iso_warm / 1e6,
                   iso_cold, flop / iso_cold / 1e6);
        }
    }
    return 0;
}
