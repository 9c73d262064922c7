// Probe: lane/element mapping of ds_read_b64_tr_b16 on gfx950.  LDS holds lds[i] = i (16-bit); lane l points at
// element 4*l (linear 8-byte pieces).  Prints which source element each (lane, j) receives.
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef short s16x4 __attribute__((ext_vector_type(4)));
__global__ void k(short* out) {
  __shared__ short lds[1024];
  for (int i = threadIdx.x; i < 1024; i += 64) lds[i] = (short)i;
  __syncthreads();
  typedef __attribute__((address_space(3))) s16x4 lds_v;
  s16x4 v = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_v*)(lds + threadIdx.x * 4));
  for (int j = 0; j < 4; ++j) out[threadIdx.x * 4 + j] = v[j];
}
int main() {
  short* d; short h[256];
  hipMalloc(&d, sizeof(h));
  hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d);
  hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
  int ok = 1;
  for (int l = 0; l < 64; ++l) {
    printf("lane %2d:", l);
    for (int j = 0; j < 4; ++j) {
      printf(" %4d", h[l * 4 + j]);
      int expect = (l & 15) + 16 * j + 64 * (l >> 4);   // = source lane 16*(l>>4) + 4*j + (l&15)/4, element l&3
      if (h[l * 4 + j] != expect) ok = 0;
    }
    printf("\n");
  }
  printf("MAPPING_AS_ASSUMED=%d\n", ok);
  return 0;
}
