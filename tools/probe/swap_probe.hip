// prints what v_permlane16_swap / v_permlane32_swap do to (vdst, src) = (lane, 100 + lane)
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void k(unsigned* o) {
    const unsigned l = threadIdx.x;
    auto r = __builtin_amdgcn_permlane16_swap(l, 100 + l, false, false);
    o[l] = r[0]; o[64 + l] = r[1];
    auto q = __builtin_amdgcn_permlane32_swap(l, 100 + l, false, false);
    o[128 + l] = q[0]; o[192 + l] = q[1];
}
int main() {
    unsigned* d; hipMalloc(&d, 1024); k<<<1, 64>>>(d);
    unsigned h[256]; hipMemcpy(h, d, 1024, hipMemcpyDeviceToHost);
    const char* names[4] = {"swap16 r0", "swap16 r1", "swap32 r0", "swap32 r1"};
    for (int a = 0; a < 4; ++a) { printf("%s:", names[a]); for (int i = 0; i < 64; i += 4) printf(" %u", h[a * 64 + i]); printf("\n"); }
    return 0;
}
