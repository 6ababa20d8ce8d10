// Shared device helpers for the gfx950 kernels (wave = 64 lanes everywhere).
#pragma once
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>
#include <stdint.h>
#include <cstdlib>

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

// Every kernel launch of the library goes through RF_LAUNCH.  While the calling thread has armed the kernel timer
// (rf_kernel_timer_arm), the launch carries a start / stop event pair that brackets exactly the dispatch -- the
// execution time of the kernel itself, the quantity rocprofv3's kernel trace reports -- instead of whatever else a
// pair of stream events around the call would include (dispatch turnaround, host launch path).
extern thread_local hipEvent_t rf_g_timer_start, rf_g_timer_stop;
extern thread_local bool rf_g_timer_armed;
#define RF_LAUNCH(kernel, grid, block, lds, stream, ...)                                                      \
  do {                                                                                                        \
    if (rf_g_timer_armed) {                                                                                   \
      rf_g_timer_armed = false; /* one kernel per arm: the first one of the call */                           \
      hipExtLaunchKernelGGL(kernel, grid, block, lds, stream, rf_g_timer_start, rf_g_timer_stop, 0, __VA_ARGS__); \
    } else {                                                                                                  \
      hipLaunchKernelGGL(kernel, grid, block, lds, stream, __VA_ARGS__);                                      \
    }                                                                                                         \
  } while (0)

#define RF_REQUIRE(cond)                                   \
  do {                                                     \
    if (!(cond)) {                                         \
      rf_g_last_error = "invalid argument: " #cond;        \
      return RF_EINVAL;                                    \
    }                                                      \
  } while (0)

// Activation storage of the frozen conv trunk: fp32, or bf16 in the bf16 matrix-core mode (the convolutions round
// their inputs to bf16 anyway, so keeping the maps in bf16 halves the HBM traffic of an HBM-bound stack).  These
// helpers move 4 consecutive channels (16 B / 8 B); arithmetic between load and store is always fp32.
__device__ __forceinline__ float4 act_ld4(const float* p) { return *reinterpret_cast<const float4*>(p); }
__device__ __forceinline__ float4 act_ld4(const __bf16* p) {
  const uint2 r = *reinterpret_cast<const uint2*>(p);  // bf16 -> fp32 is a 16-bit shift
  return make_float4(__uint_as_float(r.x << 16), __uint_as_float(r.x & 0xffff0000u), __uint_as_float(r.y << 16),
                     __uint_as_float(r.y & 0xffff0000u));
}
__device__ __forceinline__ void act_st4(float* p, const float4& v) { *reinterpret_cast<float4*>(p) = v; }
__device__ __forceinline__ void act_st4(__bf16* p, const float4& v) {
  typedef __bf16 bf16x4_ __attribute__((ext_vector_type(4)));
  const bf16x4_ o = {(__bf16)v.x, (__bf16)v.y, (__bf16)v.z, (__bf16)v.w};  // round to nearest even
  *reinterpret_cast<bf16x4_*>(p) = o;
}
__device__ __forceinline__ float act_ld(const float* p) { return *p; }
__device__ __forceinline__ float act_ld(const __bf16* p) { return (float)*p; }
__device__ __forceinline__ void act_st(float* p, float v) { *p = v; }
__device__ __forceinline__ void act_st(__bf16* p, float v) { *p = (__bf16)v; }

// Cross-lane reductions on the DPP path (data-parallel primitives: the operand of a VALU instruction is
// taken from another lane of the same 16-lane row, no LDS round trip) -- __shfl_xor lowers to
// ds_bpermute_b32, ~100+ cycles per step, which made the 6-step butterflies the slowest part of the
// softmax / LayerNorm rows.  quad_perm [1,0,3,2], quad_perm [2,3,0,1], row_half_mirror, row_mirror leave the
// row total in all 16 lanes; the four row totals are then combined through v_readlane.
template <int CTRL>
__device__ __forceinline__ float dpp_move(float v) {
  return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), CTRL, 0xF, 0xF, true));
}
__device__ __forceinline__ float row16_sum(float v) {  // all-reduce over each aligned group of 16 lanes
  v += dpp_move<0xB1>(v);
  v += dpp_move<0x4E>(v);
  v += dpp_move<0x141>(v);
  v += dpp_move<0x140>(v);
  return v;
}
__device__ __forceinline__ float row16_max(float v) {
  v = fmaxf(v, dpp_move<0xB1>(v));
  v = fmaxf(v, dpp_move<0x4E>(v));
  v = fmaxf(v, dpp_move<0x141>(v));
  v = fmaxf(v, dpp_move<0x140>(v));
  return v;
}
__device__ __forceinline__ float lane_value(float v, int lane) {
  return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), lane));
}
__device__ __forceinline__ float wave_sum(float v) {  // total in every lane
  v = row16_sum(v);
  return (lane_value(v, 0) + lane_value(v, 16)) + (lane_value(v, 32) + lane_value(v, 48));
}
__device__ __forceinline__ float wave_max(float v) {
  v = row16_max(v);
  return fmaxf(fmaxf(lane_value(v, 0), lane_value(v, 16)), fmaxf(lane_value(v, 32), lane_value(v, 48)));
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

// Launch cap for the frozen conv trunk's kernels (they run as a side branch of the train step, underneath a chain of small
// latency-bound launches that needs free workgroup slots on every CU): RF_TRUNK_MAX_WG > 0 bounds their grids, the kernels
// walk their tiles in a grid-stride loop.  0 / unset: one workgroup per tile.
inline int trunk_grid(int blocks) {
  static const int cap = [] { const char* e = getenv("RF_TRUNK_MAX_WG"); return e ? atoi(e) : 0; }();
  return (cap > 0 && blocks > cap) ? cap : blocks;
}
