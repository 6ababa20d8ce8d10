// Shared device helpers for the gfx950 kernels (wave = 64 lanes everywhere).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "rf_hip.h"

#define RF_WAVE 64

extern thread_local const char* rf_g_last_error;

#define RF_CHECK_LAUNCH()                                  \
  do {                                                     \
    hipError_t e__ = hipGetLastError();                    \
    if (e__ != hipSuccess) {                               \
      rf_g_last_error = hipGetErrorString(e__);            \
      return RF_ELAUNCH;                                   \
    }                                                      \
  } while (0)

#define RF_REQUIRE(cond)                                   \
  do {                                                     \
    if (!(cond)) {                                         \
      rf_g_last_error = "invalid argument: " #cond;        \
      return RF_EINVAL;                                    \
    }                                                      \
  } while (0)

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
  return v;
}

__device__ __forceinline__ float gelu_erf(float x) { return 0.5f * x * (1.0f + erff(x * 0.70710678118654752440f)); }
__device__ __forceinline__ float gelu_erf_grad(float x) {
  const float cdf = 0.5f * (1.0f + erff(x * 0.70710678118654752440f));
  const float pdf = 0.39894228040143267794f * expf(-0.5f * x * x);
  return cdf + x * pdf;
}
__device__ __forceinline__ float apply_act(float v, int act) {
  switch (act) {
    case RF_ACT_RELU: return v > 0.f ? v : 0.f;
    case RF_ACT_GELU: return gelu_erf(v);
    case RF_ACT_ELU: return v > 0.f ? v : expm1f(v);
    default: return v;
  }
}
__device__ __forceinline__ float act_grad(float src, int mode) {
  switch (mode) {
    case RF_ACT_RELU: return src > 0.f ? 1.f : 0.f;  // src may be the activation output
    case RF_ACT_GELU: return gelu_erf_grad(src);     // src = pre-activation
    case RF_ACT_ELU: return src > 0.f ? 1.f : expf(src);
    default: return 1.f;
  }
}
