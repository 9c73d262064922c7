// Microbenchmark: what does a "filler" instruction between MFMAs cost on gfx950?
//   hipcc --offload-arch=gfx950 -O3 tools/mfma_filler_probe.hip -o tools/mfma_filler_probe.bin && tools/mfma_filler_probe.bin
// One block = 256 threads (1 wave per SIMD) or 512 (2 per SIMD); grid = 256 CUs.  Every wave runs ITER x 16 MFMAs on 16
// independent accumulators, with F filler instructions after every MFMA.  Everything is inline asm, so the order is exact.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

typedef __attribute__((__vector_size__(4 * sizeof(float)))) float f32x4;
typedef __attribute__((__vector_size__(16 * sizeof(float)))) float f32x16;
typedef __attribute__((__vector_size__(4 * sizeof(int)))) int i32x4;

template <int KIND, int F>
__device__ __forceinline__ void fillers(i32x4 (&r)[4], int& addr, int& sacc) {
#pragma unroll
  for (int f = 0; f < F; ++f) {
    if (KIND == 0) asm volatile("ds_read_b128 %0, %1" : "=v"(r[f & 3]) : "v"(addr));
    if (KIND == 1) asm volatile("v_add_u32 %0, 1, %0" : "+v"(addr));
  }
}

template <int MF, int KIND, int F>
__global__ void __launch_bounds__(512) probe(float* out, int iters) {
  __shared__ __attribute__((aligned(16))) char lds[16384];
  int addr = (threadIdx.x & 63) * 16;
  int sacc = 0;
  i32x4 r[4] = {};
  i32x4 a = {0x3f803f80, 0x3f803f80, 0x3f803f80, 0x3f803f80}, b = a;
  if (threadIdx.x < 1024) reinterpret_cast<i32x4*>(lds)[threadIdx.x] = a;
  __syncthreads();
  if (MF == 0) {
    f32x4 acc[16];
#pragma unroll
    for (int i = 0; i < 16; ++i) acc[i] = f32x4{0, 0, 0, 0};
    for (int it = 0; it < iters; ++it) {
#pragma unroll
      for (int i = 0; i < 16; ++i) {
        asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+v"(acc[i]) : "v"(a), "v"(b));
        fillers<KIND, F>(r, addr, sacc);
      }
      if (KIND == 0) asm volatile("s_waitcnt lgkmcnt(0)");
    }
    float s = 0;
#pragma unroll
    for (int i = 0; i < 16; ++i) s += acc[i][0];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s + r[0][0] + r[1][0] + r[2][0] + r[3][0] + addr + sacc;
  } else {
    f32x16 acc[4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int j = 0; j < 16; ++j) acc[i][j] = 0;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
      for (int rep = 0; rep < 2; ++rep)
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+v"(acc[i]) : "v"(a), "v"(b));
          fillers<KIND, 2 * F>(r, addr, sacc);   // same fillers per FLOP as the 16x16x32 case
        }
      if (KIND == 0) asm volatile("s_waitcnt lgkmcnt(0)");
    }
    float s = 0;
#pragma unroll
    for (int i = 0; i < 4; ++i) s += acc[i][0];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s + r[0][0] + r[1][0] + r[2][0] + r[3][0] + addr + sacc;
  }
}

template <int MF, int KIND, int F>
void run(float* out, int threads, const char* name) {
  const int iters = 4000;
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  probe<MF, KIND, F><<<256, threads>>>(out, 10);
  hipDeviceSynchronize();
  hipEventRecord(e0);
  probe<MF, KIND, F><<<256, threads>>>(out, iters);
  hipEventRecord(e1);
  hipDeviceSynchronize();
  float ms = 0;
  hipEventElapsedTime(&ms, e0, e1);
  const double flops = 2.0 * 16 * 16 * 32 * 16 * (double)iters * (threads / 64) * 256;
  printf("%-10s %-8s F=%d waves/SIMD=%d  %7.3f ms  %7.1f TFLOP/s  %6.1f ns per 16x16x32-equivalent MFMA per SIMD\n", MF ? "32x32x16" : "16x16x32", name, F,
         threads / 256, ms, flops / ms * 1e-9, ms * 1e6 / (16.0 * iters * (threads / 256)));
}

int main() {
  float* out;
  hipMalloc(&out, 512 * 256 * sizeof(float));
#define ALL(MF, KIND, NAME)                                                    \
  for (int t = 256; t <= 512; t += 256) {                                      \
    run<MF, KIND, 0>(out, t, NAME);                                            \
    run<MF, KIND, 1>(out, t, NAME);                                            \
    run<MF, KIND, 2>(out, t, NAME);                                            \
    run<MF, KIND, 3>(out, t, NAME);                                            \
  }
  ALL(0, 0, "ds_read")
  ALL(1, 0, "ds_read")
  ALL(0, 1, "valu")
  ALL(1, 1, "valu")
  return 0;
}
