// Microbenchmark: how far ahead of its consumer must a ds_read_b128 be issued on gfx950 when every wave of the CU does the same?
//   hipcc --offload-arch=gfx950 -O3 tools/lds_latency_probe.hip -o tools/lds_latency_probe.bin
// Per iteration: 16 MFMAs (16 accumulators), one ds_read_b128 per two MFMAs; the A operand of MFMA pair n is the register
// filled by the read issued DR pairs earlier, guarded by s_waitcnt lgkmcnt(DR).  All inline asm: the order is exact.
#include <hip/hip_runtime.h>
#include <cstdio>

typedef __attribute__((__vector_size__(4 * sizeof(float)))) float f32x4;
typedef __attribute__((__vector_size__(4 * sizeof(int)))) int i32x4;

template <int DR, int WITH_WRITES>
__global__ void __launch_bounds__(512) probe(float* out, int iters) {
  __shared__ __attribute__((aligned(16))) char lds[32768];
  const int addr = (threadIdx.x & 63) * 16 + (threadIdx.x >> 6) * 1024;
  i32x4 r[8];
  i32x4 b = {0x3f803f80, 0x3f803f80, 0x3f803f80, 0x3f803f80};
#pragma unroll
  for (int i = 0; i < 8; ++i) r[i] = b;
  reinterpret_cast<i32x4*>(lds)[threadIdx.x] = b;
  reinterpret_cast<i32x4*>(lds)[threadIdx.x + 512] = b;
  __syncthreads();
  f32x4 acc[16];
#pragma unroll
  for (int i = 0; i < 16; ++i) acc[i] = f32x4{0, 0, 0, 0};
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int n = 0; n < 8; ++n) {
      asm volatile("ds_read_b128 %0, %1" : "=v"(r[n]) : "v"(addr));
      if (WITH_WRITES && n == 3) asm volatile("ds_write_b128 %0, %1 offset:16384" : : "v"(addr), "v"(b));
      asm volatile("s_waitcnt lgkmcnt(%0)" : : "n"(DR + (WITH_WRITES && n >= 3 && n - 3 < DR ? 1 : 0)));
      asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+v"(acc[2 * n]) : "v"(r[(n + 8 - DR) & 7]), "v"(b));
      asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+v"(acc[2 * n + 1]) : "v"(r[(n + 8 - DR) & 7]), "v"(b));
    }
  }
  asm volatile("s_waitcnt lgkmcnt(0)");
  float s = 0;
#pragma unroll
  for (int i = 0; i < 16; ++i) s += acc[i][0];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template <int DR, int WW>
void run(float* out, int threads) {
  const int iters = 4000;
  hipEvent_t e0, e1;
  (void)hipEventCreate(&e0);
  (void)hipEventCreate(&e1);
  probe<DR, WW><<<256, threads>>>(out, 10);
  (void)hipDeviceSynchronize();
  (void)hipEventRecord(e0);
  probe<DR, WW><<<256, threads>>>(out, iters);
  (void)hipEventRecord(e1);
  (void)hipDeviceSynchronize();
  float ms = 0;
  (void)hipEventElapsedTime(&ms, e0, e1);
  const double flops = 2.0 * 16 * 16 * 32 * 16 * (double)iters * (threads / 64) * 256;
  printf("distance %d reads (%2d MFMAs) writes=%d waves/SIMD=%d  %7.3f ms  %7.1f TFLOP/s\n", DR, 2 * DR, WW, threads / 256, ms, flops / ms * 1e-9);
}

int main() {
  float* out;
  (void)hipMalloc(&out, 512 * 256 * sizeof(float));
#define ROW(WW, T) run<0, WW>(out, T); run<1, WW>(out, T); run<2, WW>(out, T); run<3, WW>(out, T); run<4, WW>(out, T); run<5, WW>(out, T); run<6, WW>(out, T); run<7, WW>(out, T);
  ROW(0, 256) ROW(0, 512) ROW(1, 256) ROW(1, 512)
  return 0;
}
