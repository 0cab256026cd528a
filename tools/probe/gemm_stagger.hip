// Dev harness: are the persistent GEMM's epilogues slow because every CU runs its epilogue at the same moment?
// All 256 workgroups of a launch start together and every tile takes the same time, so the chip alternates between
// K loops (operands from L2 / the Infinity Cache, HBM nearly idle) and epilogues (every CU storing — and, for
// EPI_RESID24, loading — its 128 / 392 KB at once).  This probe delays a part of the workgroups at the start
// (the kernel's PP_CLOCK_BEGIN hook; the library's build leaves it empty) and reports the launch's duration and the
// longest workgroup's own busy time: if the busy time falls by more than the delay costs, a desynchronised schedule pays.
//   hipcc -O3 --offload-arch=gfx950 -std=c++17 -ffp-contract=off tools/probe/gemm_stagger.hip -o tools/probe/gemm_stagger
//   ./gemm_stagger [phases=2] [delay_us=14]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <algorithm>
#include <vector>
#include "../../include/mi355clip.h"
__device__ int pp_phases = 1;          // workgroup slot s (= blockIdx.x >> 3: the s-th workgroup of its XCD) waits (s % phases) * delay
__device__ int pp_delay_ticks = 0;     // in s_memrealtime ticks (100 MHz)
__device__ unsigned long long pp_busy[512];
#define PP_CLOCK_BEGIN                                                                                         \
    {                                                                                                          \
        const int ph_ = (int)((blockIdx.x >> 3) % (unsigned)pp_phases);                                        \
        const unsigned long long until_ = __builtin_amdgcn_s_memrealtime() + (unsigned long long)(ph_ * pp_delay_ticks); \
        while (__builtin_amdgcn_s_memrealtime() < until_) __builtin_amdgcn_s_sleep(32);                        \
    }                                                                                                          \
    const unsigned long long r0_ = __builtin_amdgcn_s_memrealtime();
#define PP_CLOCK_END if (threadIdx.x == 0) pp_busy[blockIdx.x] = __builtin_amdgcn_s_memrealtime() - r0_;
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
    const int phases = argc > 1 ? atoi(argv[1]) : 2;
    const double delay_us = argc > 2 ? atof(argv[2]) : 14.0;
    const int order_arg = 4;
    const size_t Mmax = 65792;
    bf16_t *X, *W, *O; float *bias, *cvec, *stats, *part; uint8_t* xlo; bf16_t* xhi;
    CK(hipMalloc(&X, Mmax * 4096 * 2)); CK(hipMalloc(&W, (size_t)4096 * 4096 * 2)); CK(hipMalloc(&O, Mmax * 4096 * 2));
    CK(hipMalloc(&bias, 4096 * 4)); CK(hipMemset(bias, 0, 4096 * 4));
    CK(hipMalloc(&cvec, 4096 * 4)); CK(hipMemset(cvec, 0, 4096 * 4));
    CK(hipMalloc(&stats, Mmax * 8)); CK(hipMalloc(&part, Mmax * 32 * 8));
    CK(hipMalloc(&xhi, Mmax * 1024 * 4)); xlo = (uint8_t*)xhi + Mmax * 1024 * 2;
    hipLaunchKernelGGL(fill_bf16, 2048, 256, 0, 0, X, Mmax * 4096, 1, 1.0f);
    hipLaunchKernelGGL(fill_bf16, 2048, 256, 0, 0, W, (size_t)4096 * 4096, 7, 0.05f);
    hipLaunchKernelGGL(fill_bf16, 2048, 256, 0, 0, xhi, Mmax * 1024, 3, 1.0f);
    hipLaunchKernelGGL(fill_u8, 2048, 256, 0, 0, xlo, Mmax * 1024);
    hipLaunchKernelGGL(fill_stats, 256, 256, 0, 0, stats, Mmax);
    CK(hipDeviceSynchronize());
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    auto set = [&](int ph, double us) {
        const int ticks = (int)(us * 100.0);
        CK(hipMemcpyToSymbol(HIP_SYMBOL(pp_phases), &ph, sizeof(int)));
        CK(hipMemcpyToSymbol(HIP_SYMBOL(pp_delay_ticks), &ticks, sizeof(int)));
    };
    // returns {launch period us (back to back), median / max workgroup busy us of the last launch}
    auto run = [&](auto&& fn, int reps, double* busy_med, double* busy_max) {
        for (int i = 0; i < 3; ++i) fn();
        CK(hipDeviceSynchronize());
        CK(hipEventRecord(e0));
        for (int i = 0; i < reps; ++i) fn();
        CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
        CK(hipGetLastError());
        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
        std::vector<unsigned long long> b(256);
        CK(hipMemcpyFromSymbol(b.data(), HIP_SYMBOL(pp_busy), 256 * 8));
        if (getenv("PP_XCD")) { printf("   busy us by blockIdx %% 8:"); for (int x = 0; x < 8; ++x) { double a = 0; for (int i = x; i < 256; i += 8) a += b[i] / 100.0; printf(" %.1f", a / 32); } printf("\n"); }
        std::sort(b.begin(), b.end());
        *busy_med = b[128] / 100.0; *busy_max = b[255] / 100.0;
        return (double)ms / reps * 1e3;
    };
    PpFold lnf; lnf.cvec = cvec; lnf.stats = stats;
    PpFold res; res.xlo = xlo; res.part = part;
    struct Case { const char* name; int epi, N, K; };
    const Case cases[] = {{"qkv BIAS", 0, 3072, 1024}, {"qkv LNF", 1, 3072, 1024}, {"out BIAS", 0, 1024, 1024}, {"out RESID24", 2, 1024, 1024},
                          {"fc1 LNF_QGELU", 3, 4096, 1024}, {"fc2 BIAS", 0, 1024, 4096}, {"fc2 RESID24", 2, 1024, 4096}};
    for (size_t Mrows : {(size_t)32896, (size_t)65792}) {
        const size_t Mp = (Mrows + 255) / 256 * 256;
        for (const Case& c : cases) {
            auto fn = [&] {
                if (c.epi == 0) launch<EPI_BIAS>(X, W, bias, c.N == 1024 ? (void*)O : (void*)O, Mp, c.N, c.K, PpFold(), order_arg);
                else if (c.epi == 1) launch<EPI_LNF>(X, W, bias, O, Mp, c.N, c.K, lnf, order_arg);
                else if (c.epi == 2) launch<EPI_RESID24>(X, W, bias, xhi, Mp, c.N, c.K, res, order_arg);
                else launch<EPI_LNF_QGELU>(X, W, bias, O, Mp, c.N, c.K, lnf, order_arg);
            };
            double m0, x0, m1, x1;
            set(1, 0.0);
            const double t0 = run(fn, 20, &m0, &x0);
            set(phases, delay_us);
            const double t1 = run(fn, 20, &m1, &x1);
            printf("M=%zu %-14s in step: %.1f us (workgroup busy median %.1f max %.1f) | %d phases x %.1f us: %.1f us (busy median %.1f max %.1f)\n",
                   Mrows, c.name, t0, m0, x0, phases, delay_us, t1, m1, x1);
        }
    }
    return 0;
}
