// Microbenchmark: cost of ONE memory instruction every P MFMAs (v_mfma_f32_16x16x32_bf16 stream, 16 accumulators) on gfx950.
//   kinds: global_load_dwordx4 (L2-resident 1 KiB per wave-instruction), ds_write_b128, ds_read_b128
//   hipcc --offload-arch=gfx950 -O3 tools/mfma_mem_probe.hip -o tools/mfma_mem_probe.bin
#include <hip/hip_runtime.h>
#include <cstdio>

typedef __attribute__((__vector_size__(4 * sizeof(float)))) float f32x4;
typedef __attribute__((__vector_size__(4 * sizeof(int)))) int i32x4;

template <int KIND, int P>
__global__ void __launch_bounds__(512) probe(float* out, const i32x4* src, int iters) {
  __shared__ __attribute__((aligned(16))) char lds[65536];
  const int addr = (threadIdx.x & 63) * 16 + (threadIdx.x >> 6) * 1024;
  const i32x4* gp = src + (blockIdx.x * 512 + threadIdx.x);
  i32x4 r[4];
  i32x4 b = {0x3f803f80, 0x3f803f80, 0x3f803f80, 0x3f803f80};
#pragma unroll
  for (int i = 0; i < 4; ++i) r[i] = b;
  f32x4 acc[16];
#pragma unroll
  for (int i = 0; i < 16; ++i) acc[i] = f32x4{0, 0, 0, 0};
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int i = 0; i < 16; ++i) {
      asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+v"(acc[i]) : "v"(b), "v"(b));
      if (P > 0 && (i % P) == P - 1) {
        if (KIND == 0) asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(r[(i / P) & 3]) : "v"(gp));
        if (KIND == 1) asm volatile("ds_write_b128 %0, %1" : : "v"(addr), "v"(b));
        if (KIND == 2) asm volatile("ds_read_b128 %0, %1" : "=v"(r[(i / P) & 3]) : "v"(addr));
      }
    }
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)");
  }
  float s = 0;
#pragma unroll
  for (int i = 0; i < 16; ++i) s += acc[i][0];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s + r[0][0] + r[1][0] + r[2][0] + r[3][0] + lds[threadIdx.x];
}

template <int KIND, int P>
void run(float* out, const i32x4* src, int threads, const char* name) {
  const int iters = 4000;
  hipEvent_t e0, e1;
  (void)hipEventCreate(&e0);
  (void)hipEventCreate(&e1);
  probe<KIND, P><<<256, threads>>>(out, src, 10);
  (void)hipDeviceSynchronize();
  (void)hipEventRecord(e0);
  probe<KIND, P><<<256, threads>>>(out, src, iters);
  (void)hipEventRecord(e1);
  (void)hipDeviceSynchronize();
  float ms = 0;
  (void)hipEventElapsedTime(&ms, e0, e1);
  const double flops = 2.0 * 16 * 16 * 32 * 16 * (double)iters * (threads / 64) * 256;
  const double ns_per_mfma = ms * 1e6 / (16.0 * iters * (threads / 256));
  printf("%-12s one per %2d MFMAs  waves/SIMD=%d  %7.3f ms  %7.1f TFLOP/s  %5.2f ns per MFMA per SIMD\n", name, P, threads / 256, ms, flops / ms * 1e-9, ns_per_mfma);
}

int main() {
  float* out;
  i32x4* src;
  (void)hipMalloc(&out, 512 * 256 * sizeof(float));
  (void)hipMalloc(&src, 512 * 256 * 16);
  (void)hipMemset(src, 0, 512 * 256 * 16);
#define ROW(K, NAME, T) run<K, 0>(out, src, T, "none"); run<K, 16>(out, src, T, NAME); run<K, 8>(out, src, T, NAME); run<K, 4>(out, src, T, NAME); run<K, 2>(out, src, T, NAME);
  ROW(0, "global_load", 256) ROW(0, "global_load", 512) ROW(1, "ds_write", 256) ROW(1, "ds_write", 512) ROW(2, "ds_read", 256) ROW(2, "ds_read", 512)
  return 0;
}
