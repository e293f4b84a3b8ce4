// Sustained fp32 MFMA rate on gfx950 (no memory traffic): what "peak" really is under load.
// hipcc -O3 --offload-arch=gfx950 tools/micro/mfma_peak.hip -o tools/micro/mfma_peak
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

template <int NACC>
__global__ __launch_bounds__(256) void k32(float* out, int iters, long long* clk) {
  f32x16 acc[NACC];
  for (int i = 0; i < NACC; ++i)
    for (int j = 0; j < 16; ++j) acc[i][j] = 0.f;
  float a = threadIdx.x * 1e-3f, b = 1.0f + blockIdx.x * 1e-6f;
  long long t0 = __builtin_readcyclecounter();
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int u = 0; u < 8; ++u)
#pragma unroll
      for (int i = 0; i < NACC; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[i], 0, 0, 0);
  }
  long long t1 = __builtin_readcyclecounter();
  float s = 0;
  for (int i = 0; i < NACC; ++i)
    for (int j = 0; j < 16; ++j) s += acc[i][j];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
  if (threadIdx.x == 0 && blockIdx.x == 0) clk[0] = t1 - t0;
}

template <int NACC>
__global__ __launch_bounds__(256) void k16(float* out, int iters) {
  f32x4 acc[NACC];
  for (int i = 0; i < NACC; ++i)
    for (int j = 0; j < 4; ++j) acc[i][j] = 0.f;
  float a = threadIdx.x * 1e-3f, b = 1.0f + blockIdx.x * 1e-6f;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int u = 0; u < 8; ++u)
#pragma unroll
      for (int i = 0; i < NACC; ++i) acc[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc[i], 0, 0, 0);
  }
  float s = 0;
  for (int i = 0; i < NACC; ++i)
    for (int j = 0; j < 4; ++j) s += acc[i][j];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template <typename F>
static float run(F launch, int reps) {
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  launch(); hipDeviceSynchronize();
  hipEventRecord(e0);
  for (int r = 0; r < reps; ++r) launch();
  hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  return ms / reps;
}

int main() {
  float* out; hipMalloc(&out, 4096 * 256 * 4);
  long long* clk; hipMalloc(&clk, 8);
  const int iters = 2000;
  for (int blocks_per_cu : {1, 2, 4}) {
    const int nb = 256 * blocks_per_cu;
    {
      float ms = run([&] { hipLaunchKernelGGL(k32<2>, dim3(nb), dim3(256), 0, 0, out, iters, clk); }, 5);
      double fl = (double)nb * 4 * iters * 8 * 2 * 4096.0;
      long long c; hipMemcpy(&c, clk, 8, hipMemcpyDeviceToHost);
      printf("32x32x2 acc=2 blocks/CU=%d waves/SIMD=%d: %.3f ms  %.1f TFLOP/s   (s_memtime ticks %lld -> %.1f MHz ticks)\n", blocks_per_cu,
             blocks_per_cu, ms, fl / ms / 1e9, c, c / (ms * 1e3));
    }
    {
      float ms = run([&] { hipLaunchKernelGGL(k32<4>, dim3(nb), dim3(256), 0, 0, out, iters, clk); }, 5);
      double fl = (double)nb * 4 * iters * 8 * 4 * 4096.0;
      printf("32x32x2 acc=4 blocks/CU=%d: %.3f ms  %.1f TFLOP/s\n", blocks_per_cu, ms, fl / ms / 1e9);
    }
    {
      float ms = run([&] { hipLaunchKernelGGL(k16<8>, dim3(nb), dim3(256), 0, 0, out, iters); }, 5);
      double fl = (double)nb * 4 * iters * 8 * 8 * 2048.0;
      printf("16x16x4 acc=8 blocks/CU=%d: %.3f ms  %.1f TFLOP/s\n", blocks_per_cu, ms, fl / ms / 1e9);
    }
  }
  // long run: does the rate sag (power / clock) when sustained for ~1 s?
  {
    const int nb = 1024;
    for (int rep = 0; rep < 3; ++rep) {
      float ms = run([&] { hipLaunchKernelGGL(k32<2>, dim3(nb), dim3(256), 0, 0, out, iters * 4, clk); }, 40);
      double fl = (double)nb * 4 * iters * 4 * 8 * 2 * 4096.0;
      printf("sustained 32x32x2 (40 launches): %.3f ms/launch  %.1f TFLOP/s\n", ms, fl / ms / 1e9);
    }
  }
  return 0;
}
