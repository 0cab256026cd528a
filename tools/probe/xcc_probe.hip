// Which XCD does workgroup b of a launch land on?  xcd_remap (vit_kernels.h) and the persistent GEMM's tile order assume the
// round-robin b -> b mod 8.  Prints the XCC id (HW_REG_XCC_ID) of the first 32 workgroups of a 256- and a 2048-workgroup launch
// and how many of the 256 satisfy xcc(b) == xcc(b mod 8).
//   hipcc -O3 --offload-arch=gfx950 tools/probe/xcc_probe.hip -o tools/probe/xcc_probe
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void xcc_kernel(int* out) {
    if (threadIdx.x == 0) out[blockIdx.x] = __builtin_amdgcn_s_getreg((3 << 11) | 20) & 15;   // HW_REG_XCC_ID[3:0]
}
int main() {
    int* d; hipMalloc(&d, 4096 * 4);
    for (int grid : {256, 2048}) {
        for (int rep = 0; rep < 3; ++rep) {
            hipMemset(d, 0xff, 4096 * 4);
            hipLaunchKernelGGL(xcc_kernel, dim3(grid), dim3(512), 0, 0, d);
            int h[4096]; hipMemcpy(h, d, grid * 4, hipMemcpyDeviceToHost);
            int ok = 0; for (int b = 0; b < grid; ++b) ok += h[b] == h[b & 7];
            int distinct = 0; for (int b = 0; b < 8; ++b) { bool seen = false; for (int c = 0; c < b; ++c) seen |= h[c] == h[b]; distinct += !seen; }
            printf("grid %4d rep %d: first 16 xcc ids:", grid, rep);
            for (int b = 0; b < 16; ++b) printf(" %d", h[b]);
            printf(" | xcc(b) == xcc(b mod 8) for %d of %d, %d distinct among the first 8\n", ok, grid, distinct);
        }
    }
    return 0;
}
