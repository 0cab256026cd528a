// Dev harness: what the LayerNorm-free tower's epilogues cost the persistent GEMM.  The tower's four shapes at the
// half-chunk and full-chunk row counts, each with the epilogue it has today and the one "ln_fold" gives it:
//   qkv  EPI_BIAS        -> EPI_LNF          (rstd * acc - mean rstd * c + b')
//   out  EPI_BIAS        -> EPI_RESID24      (x += bf16(acc + b) on the 24-bit planes in place, + row sums)
//   fc1  EPI_BIAS_QGELU  -> EPI_LNF_QGELU
//   fc2  EPI_BIAS        -> EPI_RESID24
// beside the LayerNorm kernels the fold removes (ln_kernel LN1 / LN2 forms on the x24 planes) and ln_stats_kernel.
//   hipcc -O3 --offload-arch=gfx950 -std=c++17 -ffp-contract=off tools/probe/gemm_fold_sweep.hip -o tools/probe/gemm_fold_sweep
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <algorithm>
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
__global__ void fill_u8(uint8_t* p, size_t n) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) p[i] = (uint8_t)(i * 131u);
}
__global__ void fill_stats(float* p, size_t rows) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < rows; i += (size_t)gridDim.x * blockDim.x) { p[2 * i] = 1.0f + (i % 7) * 0.01f; p[2 * i + 1] = 0.01f * (float)(i % 5); }
}
template <int EPI>
static void launch(const bf16_t* X, const bf16_t* W, const float* bias, void* O, size_t Mp, int N, int K, const PpFold& f, int order_arg) {
    constexpr bool LNF = EPI == EPI_LNF || EPI == EPI_LNF_QGELU;
    constexpr int LDS = 131072 + 18432 + 8 * (LNF ? 1536 : 256);
    auto kern = gemm_bf16_pp_kernel<EPI, bf16_t>;
    static bool once = false;
    if (!once) { CK(hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, LDS)); once = true; }
    const int n_tiles = (int)((Mp / 256) * (N / 256)), grid = std::min(n_tiles * 4, 256);
    const int left = n_tiles % grid;
    const int n_full = (left > 0 && left * 4 <= grid) ? n_tiles - left : n_tiles;
    const int nt = N / 256;
    const int order = (order_arg > 0 && nt > order_arg && nt % order_arg == 0) ? order_arg : 0;
    hipLaunchKernelGGL(kern, dim3(grid), dim3(512), LDS, 0, X, W, bias, O, (int)Mp, N, K, N, n_tiles, n_full, order, f);
}
int main(int argc, char** argv) {
    const int order_arg = argc > 1 ? atoi(argv[1]) : 4;
    const size_t Mmax = 65792;
    bf16_t *X, *W, *O, *d1, *d2, *y; float *bias, *cvec, *stats, *part, *lnw; uint8_t* xlo; bf16_t* xhi;
    CK(hipMalloc(&X, Mmax * 4096 * 2)); CK(hipMalloc(&W, (size_t)4096 * 4096 * 2)); CK(hipMalloc(&O, Mmax * 4096 * 2));
    CK(hipMalloc(&bias, 4096 * 4)); CK(hipMemset(bias, 0, 4096 * 4));
    CK(hipMalloc(&cvec, 4096 * 4)); CK(hipMemset(cvec, 0, 4096 * 4));
    CK(hipMalloc(&lnw, 4096 * 4)); CK(hipMemset(lnw, 0, 4096 * 4));
    CK(hipMalloc(&stats, Mmax * 8)); CK(hipMalloc(&part, Mmax * 32 * 8));
    CK(hipMalloc(&xhi, Mmax * 1024 * 4)); xlo = (uint8_t*)xhi + Mmax * 1024 * 2;   // both planes in one allocation, as the library keeps them
    CK(hipMalloc(&d1, Mmax * 1024 * 2)); CK(hipMalloc(&d2, Mmax * 1024 * 2)); CK(hipMalloc(&y, Mmax * 1024 * 2));
    hipLaunchKernelGGL(fill_bf16, 2048, 256, 0, 0, X, Mmax * 4096, 1, 1.0f);
    hipLaunchKernelGGL(fill_bf16, 2048, 256, 0, 0, W, (size_t)4096 * 4096, 7, 0.05f);
    hipLaunchKernelGGL(fill_bf16, 2048, 256, 0, 0, xhi, Mmax * 1024, 3, 1.0f);
    hipLaunchKernelGGL(fill_bf16, 2048, 256, 0, 0, d1, Mmax * 1024, 5, 0.1f);
    hipLaunchKernelGGL(fill_bf16, 2048, 256, 0, 0, d2, Mmax * 1024, 9, 0.1f);
    hipLaunchKernelGGL(fill_u8, 2048, 256, 0, 0, xlo, Mmax * 1024);
    hipLaunchKernelGGL(fill_stats, 256, 256, 0, 0, stats, Mmax);
    CK(hipDeviceSynchronize());
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    auto time_us = [&](auto&& fn, int reps) {
        for (int i = 0; i < 3; ++i) fn();
        CK(hipDeviceSynchronize());
        CK(hipEventRecord(e0));
        for (int i = 0; i < reps; ++i) fn();
        CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
        CK(hipGetLastError());
        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
        return (double)ms / reps * 1e3;
    };
    PpFold lnf; lnf.cvec = cvec; lnf.stats = stats;
    PpFold res; res.xlo = xlo; res.part = part;
    for (size_t Mrows : {(size_t)32896, (size_t)65792}) {
        const size_t Mp = (Mrows + 255) / 256 * 256;
        const double a = time_us([&] { launch<EPI_BIAS>(X, W, bias, O, Mp, 3072, 1024, PpFold(), order_arg); }, 20);
        const double b = time_us([&] { launch<EPI_LNF>(X, W, bias, O, Mp, 3072, 1024, lnf, order_arg); }, 20);
        const double c = time_us([&] { launch<EPI_BIAS>(X, W, bias, O, Mp, 1024, 1024, PpFold(), order_arg); }, 20);
        const double d = time_us([&] { launch<EPI_RESID24>(X, W, bias, xhi, Mp, 1024, 1024, res, order_arg); }, 20);
        const double e = time_us([&] { launch<EPI_BIAS_QGELU>(X, W, bias, O, Mp, 4096, 1024, PpFold(), order_arg); }, 20);
        const double f = time_us([&] { launch<EPI_LNF_QGELU>(X, W, bias, O, Mp, 4096, 1024, lnf, order_arg); }, 20);
        const double g = time_us([&] { launch<EPI_BIAS>(X, W, bias, O, Mp, 1024, 4096, PpFold(), order_arg); }, 20);
        const double h = time_us([&] { launch<EPI_RESID24>(X, W, bias, xhi, Mp, 1024, 4096, res, order_arg); }, 20);
        const unsigned lb = (unsigned)((Mrows + 3) / 4);
        const double l1 = time_us([&] { hipLaunchKernelGGL((ln_kernel<bf16_t, 4, 4, true>), dim3(lb), dim3(256), 0, 0, (float*)xhi, d1, d2, y, lnw, lnw, (int)Mrows, 1e-5f, 1024, 0, 0, Mmax * 1024 * 2, 0u); }, 20);
        const double l2 = time_us([&] { hipLaunchKernelGGL((ln_kernel<bf16_t, 4, 4, false>), dim3(lb), dim3(256), 0, 0, (float*)xhi, d1, (const bf16_t*)nullptr, y, lnw, lnw, (int)Mrows, 1e-5f, 1024, 0, 0, Mmax * 1024 * 2, 0u); }, 20);
        const double st = time_us([&] { hipLaunchKernelGGL(ln_stats_kernel, dim3((unsigned)(Mp / 16)), dim3(256), 0, 0, part, stats, (int)Mp, 32, 1.0f / 1024, 1e-5f); }, 20);
        printf("M=%zu back-to-back us:  qkv %.1f -> LNF %.1f | out %.1f -> RESID24 %.1f | fc1 %.1f -> LNF_QGELU %.1f | fc2 %.1f -> RESID24 %.1f | LN1 %.1f LN2 %.1f stats %.1f\n",
               Mrows, a, b, c, d, e, f, g, h, l1, l2, st);
        printf("M=%zu per layer: GEMMs + LayerNorms today %.1f us; ln_fold %.1f us\n", Mrows, a + c + e + g + l1 + l2, b + d + f + h + 2 * st);
    }
    return 0;
}
