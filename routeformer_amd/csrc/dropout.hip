// nn.Dropout as a streaming kernel (the sites no fused epilogue covers) + the device-side generator state.
// y = x * keep / (1 - p); the same launch serves backward (dx = dy * keep / (1 - p)): the mask is regenerated
// from (seed, step, site, element index), see philox.h.
#include "common.h"
#include "philox.h"

namespace {

__global__ void rng_advance_kernel(RfRngState* s) { s->step += 1; }

__global__ void rng_seed_kernel(RfRngState* s, unsigned long long seed, unsigned long long step) {
  s->seed = seed;
  s->step = step;
}

// one aligned quad (16 B) per thread and trip; the tail (n % 4) is handled element-wise by the last threads
__global__ __launch_bounds__(256) void dropout_kernel(const float* __restrict__ x, float* __restrict__ y, long n,
                                                      DropCfg cfg, uint8_t* __restrict__ mask_out) {
  const DropGen gen(cfg);
  const long quads = n >> 2;
  for (long q = (long)blockIdx.x * blockDim.x + threadIdx.x; q < quads; q += (long)gridDim.x * blockDim.x) {
    const float4 f = gen.factor4((unsigned long long)q << 2);
    if (x) {
      const float4 v = *reinterpret_cast<const float4*>(x + (q << 2));
      *reinterpret_cast<float4*>(y + (q << 2)) = make_float4(v.x * f.x, v.y * f.y, v.z * f.z, v.w * f.w);
    }
    if (mask_out) {
      *reinterpret_cast<uchar4*>(mask_out + (q << 2)) =
          make_uchar4(f.x != 0.f, f.y != 0.f, f.z != 0.f, f.w != 0.f);
    }
  }
  if (blockIdx.x == 0 && threadIdx.x < (n & 3)) {
    const long e = (quads << 2) + threadIdx.x;
    const float f = gen.factor((unsigned long long)e);
    if (x) y[e] = x[e] * f;
    if (mask_out) mask_out[e] = f != 0.f;
  }
}

inline bool al16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; }

}  // namespace

extern "C" int rf_rng_seed(void* state, int64_t seed, int64_t step, void* stream) {
  RF_REQUIRE(state);
  RF_LAUNCH(rng_seed_kernel, dim3(1), dim3(1), 0, static_cast<hipStream_t>(stream), static_cast<RfRngState*>(state),
            (unsigned long long)seed, (unsigned long long)step);
  RF_CHECK_LAUNCH();
  return RF_OK;
}

extern "C" int rf_rng_advance(void* state, void* stream) {
  RF_REQUIRE(state);
  RF_LAUNCH(rng_advance_kernel, dim3(1), dim3(1), 0, static_cast<hipStream_t>(stream), static_cast<RfRngState*>(state));
  RF_CHECK_LAUNCH();
  return RF_OK;
}

extern "C" int rf_dropout(const float* x, float* y, int64_t n, float p, const void* rng_state, int site,
                          const uint8_t* mask_in, uint8_t* mask_out, void* stream) {
  RF_REQUIRE(n > 0 && p > 0.f && p < 1.f && (rng_state || mask_in) && (x ? y != nullptr : mask_out != nullptr));
  RF_REQUIRE((!x || (al16(x) && al16(y))) && (!mask_in || (reinterpret_cast<uintptr_t>(mask_in) & 3) == 0) &&
             (!mask_out || (reinterpret_cast<uintptr_t>(mask_out) & 3) == 0));
  const DropCfg cfg = make_drop_cfg(rng_state, mask_in, (uint32_t)site, p);
  const long quads = (n + 3) >> 2;
  const int blocks = (int)((quads + 255) / 256 > 4096 ? 4096 : (quads + 255) / 256);
  RF_LAUNCH(dropout_kernel, dim3(blocks), dim3(256), 0, static_cast<hipStream_t>(stream), x, y, (long)n, cfg, mask_out);
  RF_CHECK_LAUNCH();
  return RF_OK;
}
