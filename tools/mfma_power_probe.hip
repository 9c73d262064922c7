// Microbenchmark: sustained v_mfma_f32_16x16x32_bf16 rate on gfx950 with constant vs random operands (data-dependent power).
//   hipcc --offload-arch=gfx950 -O3 tools/mfma_power_probe.hip -o tools/mfma_power_probe.bin
// Reports TFLOP/s and the shader clock seen by s_memtime (ticks per second of wall time).
#include <hip/hip_runtime.h>
#include <cstdio>

typedef __attribute__((__vector_size__(4 * sizeof(float)))) float f32x4;
typedef __attribute__((__vector_size__(4 * sizeof(int)))) int i32x4;

__device__ __forceinline__ unsigned hash(unsigned x) {
  x ^= x >> 16; x *= 0x7feb352du; x ^= x >> 15; x *= 0x846ca68bu; x ^= x >> 16;
  return x;
}

template <int RANDOM, int NOPER>
__global__ void __launch_bounds__(512) probe(float* out, long long* ticks, int iters) {
  i32x4 a[NOPER], b[NOPER];
#pragma unroll
  for (int k = 0; k < NOPER; ++k)
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      // bf16 pairs: sign and mantissa random, exponent 126..127 -> |v| in [0.5, 2)
      const unsigned h1 = hash(threadIdx.x * 131 + k * 17 + j * 3 + 1), h2 = hash(h1 + blockIdx.x);
      a[k][j] = RANDOM ? (int)((h1 & 0x807f807fu) | 0x3f003f00u | ((h1 >> 3) & 0x00800080u)) : 0x3f803f80;
      b[k][j] = RANDOM ? (int)((h2 & 0x807f807fu) | 0x3f003f00u | ((h2 >> 3) & 0x00800080u)) : 0x3f803f80;
    }
  f32x4 acc[16];
#pragma unroll
  for (int i = 0; i < 16; ++i) acc[i] = f32x4{0, 0, 0, 0};
  const long long t0 = __builtin_readcyclecounter();
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int i = 0; i < 16; ++i)
      asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+v"(acc[i]) : "v"(a[i % NOPER]), "v"(b[(i / 4) % NOPER]));
  }
  const long long t1 = __builtin_readcyclecounter();
  float s = 0;
#pragma unroll
  for (int i = 0; i < 16; ++i) s += acc[i][0];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
  if (threadIdx.x == 0 && blockIdx.x == 0) ticks[0] = t1 - t0;
}

typedef __attribute__((__vector_size__(16 * sizeof(float)))) float f32x16;

// same operands, v_mfma_f32_32x32x16_bf16: half the operand registers read per flop
template <int RANDOM, int NOPER>
__global__ void __launch_bounds__(512) probe32(float* out, long long* ticks, int iters) {
  i32x4 a[NOPER], b[NOPER];
#pragma unroll
  for (int k = 0; k < NOPER; ++k)
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const unsigned h1 = hash(threadIdx.x * 131 + k * 17 + j * 3 + 1), h2 = hash(h1 + blockIdx.x);
      a[k][j] = RANDOM ? (int)((h1 & 0x807f807fu) | 0x3f003f00u | ((h1 >> 3) & 0x00800080u)) : 0x3f803f80;
      b[k][j] = RANDOM ? (int)((h2 & 0x807f807fu) | 0x3f003f00u | ((h2 >> 3) & 0x00800080u)) : 0x3f803f80;
    }
  f32x16 acc[4];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 16; ++j) acc[i][j] = 0.f;
  const long long t0 = __builtin_readcyclecounter();
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int i = 0; i < 8; ++i)
      asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+v"(acc[i & 3]) : "v"(a[i % NOPER]), "v"(b[(i / 2) % NOPER]));
  }
  const long long t1 = __builtin_readcyclecounter();
  float s = 0;
#pragma unroll
  for (int i = 0; i < 4; ++i) s += acc[i][0];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
  if (threadIdx.x == 0 && blockIdx.x == 0) ticks[0] = t1 - t0;
}

template <int RANDOM, int NOPER>
void run32(float* out, long long* ticks, int threads) {
  const int iters = 20000;
  hipEvent_t e0, e1;
  (void)hipEventCreate(&e0);
  (void)hipEventCreate(&e1);
  probe32<RANDOM, NOPER><<<256, threads>>>(out, ticks, 10);
  (void)hipDeviceSynchronize();
  (void)hipEventRecord(e0);
  probe32<RANDOM, NOPER><<<256, threads>>>(out, ticks, iters);
  (void)hipEventRecord(e1);
  (void)hipDeviceSynchronize();
  float ms = 0;
  (void)hipEventElapsedTime(&ms, e0, e1);
  long long t = 0;
  (void)hipMemcpy(&t, ticks, 8, hipMemcpyDeviceToHost);
  const double flops = 2.0 * 32 * 32 * 16 * 8 * (double)iters * (threads / 64) * 256;
  printf("32x32x16 %-8s operands=%d waves/SIMD=%d  %7.3f ms  %7.1f TFLOP/s  s_memtime %.2f GHz\n", RANDOM ? "random" : "constant", NOPER, threads / 256, ms,
         flops / ms * 1e-9, t / (ms * 1e6));
}

template <int RANDOM, int NOPER>
void run(float* out, long long* ticks, int threads) {
  const int iters = 20000;
  hipEvent_t e0, e1;
  (void)hipEventCreate(&e0);
  (void)hipEventCreate(&e1);
  probe<RANDOM, NOPER><<<256, threads>>>(out, ticks, 10);
  (void)hipDeviceSynchronize();
  (void)hipEventRecord(e0);
  probe<RANDOM, NOPER><<<256, threads>>>(out, ticks, iters);
  (void)hipEventRecord(e1);
  (void)hipDeviceSynchronize();
  float ms = 0;
  (void)hipEventElapsedTime(&ms, e0, e1);
  long long t = 0;
  (void)hipMemcpy(&t, ticks, 8, hipMemcpyDeviceToHost);
  const double flops = 2.0 * 16 * 16 * 32 * 16 * (double)iters * (threads / 64) * 256;
  printf("%-8s operands=%d waves/SIMD=%d  %7.3f ms  %7.1f TFLOP/s  s_memtime %.2f GHz  %.1f ticks per MFMA per SIMD\n", RANDOM ? "random" : "constant", NOPER,
         threads / 256, ms, flops / ms * 1e-9, t / (ms * 1e6), (double)t / (16.0 * iters * (threads / 256)));
}

int main() {
  float* out;
  long long* ticks;
  (void)hipMalloc(&out, 512 * 256 * sizeof(float));
  (void)hipMalloc(&ticks, 8);
  for (int rep = 0; rep < 2; ++rep) {
    run<0, 1>(out, ticks, 256);
    run<0, 1>(out, ticks, 512);
    run<1, 1>(out, ticks, 256);
    run<1, 1>(out, ticks, 512);
    run<1, 4>(out, ticks, 256);
    run<1, 4>(out, ticks, 512);
    run32<0, 1>(out, ticks, 256);
    run32<1, 1>(out, ticks, 256);
    run32<1, 4>(out, ticks, 256);
    run32<1, 4>(out, ticks, 512);
  }
  return 0;
}
