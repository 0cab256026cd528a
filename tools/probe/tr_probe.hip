// Probe: what does ds_read_b64_tr_b16 deliver?  LDS holds element index e at short e.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef short s16x4 __attribute__((ext_vector_type(4)));
__global__ void k(short* out) {
  __shared__ __attribute__((aligned(16))) short lds[64 * 4];
  for (int i = threadIdx.x; i < 256; i += 64) lds[i] = (short)i;
  __syncthreads();
  // each lane supplies address lane*8 bytes: 16-lane group g covers shorts [64g, 64g+64): 4 rows x 16 cols
  s16x4 v = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)((char*)lds + threadIdx.x * 8));
  for (int e = 0; e < 4; ++e) out[threadIdx.x * 4 + e] = v[e];
}
int main() {
  short* d; hipMalloc(&d, 512); hipLaunchKernelGGL(k, 1, 64, 0, 0, d);
  short h[256]; hipMemcpy(h, d, 512, hipMemcpyDeviceToHost);
  for (int l = 0; l < 64; ++l) { printf("lane %2d:", l); for (int e = 0; e < 4; ++e) printf(" %3d", h[l*4+e]); printf("\n"); }
  return 0;
}
