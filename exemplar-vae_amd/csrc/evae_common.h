// Shared helpers for the libevae_hip.so translation units (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <math.h>

#include "../../include/evae_hip.h"

namespace evae {

void set_error(const char* fmt, ...);

inline int check_launch(const char* what) {
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) {
    set_error("%s: %s", what, hipGetErrorString(e));
    return EVAE_ELAUNCH;
  }
  return EVAE_OK;
}

__host__ __device__ inline size_t align_up(size_t x, size_t a) { return (x + a - 1) / a * a; }
inline int cdiv(int a, int b) { return (a + b - 1) / b; }

constexpr float kLog2Pi = 1.8378770664093453f;
constexpr float kLog2e = 1.4426950408889634f;
constexpr float kLn2 = 0.6931471805599453f;

__device__ __forceinline__ float fast_exp2(float x) { return __builtin_amdgcn_exp2f(x); }
__device__ __forceinline__ float sigmoidf(float x) { return 1.0f / (1.0f + __expf(-x)); }

// full-wave (64 lanes) reductions through DPP/shuffles
__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}
__device__ __forceinline__ double wave_sum(double v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
  return v;
}

}  // namespace evae

#define EVAE_REQUIRE(cond, ...)        \
  do {                                 \
    if (!(cond)) {                     \
      evae::set_error(__VA_ARGS__);    \
      return EVAE_EINVAL;              \
    }                                  \
  } while (0)
