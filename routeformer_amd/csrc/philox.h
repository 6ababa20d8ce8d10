// Counter-based dropout masks (Philox4x32-10, Salmon et al. 2011) for nn.Dropout on the trainable path
// (cross_modal_transformer.py:49,63,220-231,285-299; gps_backbone/layers/Embedding.py:122-126).
//
// A mask is never stored: the keep-bit of element `e` of dropout site `site` in training step `step` is a pure
// function of (seed, step, site, e), so the backward kernel regenerates exactly the mask its forward drew.
//   counter = (e / 4 low, e / 4 high, site, step)      key = (seed low, seed high)
//   the four 32-bit outputs serve elements 4*(e/4) .. 4*(e/4)+3;  keep  <=>  output >= p * 2^32.
// (seed, step) live in DEVICE memory (RfRngState, two uint64): a HIP graph that replays the step reads the
// current values, rf_rng_advance bumps `step` once per step from inside the graph.
#pragma once
#include <stdint.h>

struct RfRngState {
  unsigned long long seed;
  unsigned long long step;
};

struct DropCfg {
  const RfRngState* state;  // null = dropout off
  const uint8_t* mask_in;   // test hook: explicit keep-mask (one byte per element) instead of Philox
  uint32_t site;
  uint32_t thresh;          // p * 2^32
  float scale;              // 1 / (1 - p)
};

__device__ __forceinline__ uint4 philox4x32_10(uint4 ctr, uint2 key) {
  constexpr uint32_t M0 = 0xD2511F53u, M1 = 0xCD9E8D57u, W0 = 0x9E3779B9u, W1 = 0xBB67AE85u;
#pragma unroll
  for (int r = 0; r < 10; ++r) {
    const uint32_t hi0 = __umulhi(M0, ctr.x), lo0 = M0 * ctr.x;
    const uint32_t hi1 = __umulhi(M1, ctr.z), lo1 = M1 * ctr.z;
    ctr = make_uint4(hi1 ^ ctr.y ^ key.x, lo1, hi0 ^ ctr.w ^ key.y, lo0);
    key.x += W0;
    key.y += W1;
  }
  return ctr;
}

// keep-bits (bit i = element 4*quad + i) of one aligned group of four elements
__device__ __forceinline__ uint32_t drop_keep4(const DropCfg& d, uint2 key, uint32_t step, unsigned long long quad) {
  const uint4 r = philox4x32_10(make_uint4((uint32_t)quad, (uint32_t)(quad >> 32), d.site, step), key);
  return (r.x >= d.thresh ? 1u : 0u) | (r.y >= d.thresh ? 2u : 0u) | (r.z >= d.thresh ? 4u : 0u) |
         (r.w >= d.thresh ? 8u : 0u);
}

// Per-kernel view of the generator state: read once, then masks for any element index.
struct DropGen {
  DropCfg d;
  uint2 key;
  uint32_t step;
  __device__ __forceinline__ explicit DropGen(const DropCfg& cfg) : d(cfg), key(make_uint2(0, 0)), step(0) {
    if (d.state && !d.mask_in) {
      const unsigned long long s = d.state->seed;
      key = make_uint2((uint32_t)s, (uint32_t)(s >> 32));
      step = (uint32_t)d.state->step;
    }
  }
  __device__ __forceinline__ bool on() const { return d.state != nullptr || d.mask_in != nullptr; }
  // multiplier of element e: 0 (dropped) or 1 / (1 - p)
  __device__ __forceinline__ float factor(unsigned long long e) const {
    if (d.mask_in) return d.mask_in[e] ? d.scale : 0.f;
    const uint32_t bits = drop_keep4(d, key, step, e >> 2);
    return (bits >> (e & 3)) & 1u ? d.scale : 0.f;
  }
  // multipliers of the aligned quad starting at e (e % 4 == 0)
  __device__ __forceinline__ float4 factor4(unsigned long long e) const {
    if (d.mask_in) {
      const uchar4 m = *reinterpret_cast<const uchar4*>(d.mask_in + e);
      return make_float4(m.x ? d.scale : 0.f, m.y ? d.scale : 0.f, m.z ? d.scale : 0.f, m.w ? d.scale : 0.f);
    }
    const uint32_t bits = drop_keep4(d, key, step, e >> 2);
    return make_float4(bits & 1u ? d.scale : 0.f, bits & 2u ? d.scale : 0.f, bits & 4u ? d.scale : 0.f,
                       bits & 8u ? d.scale : 0.f);
  }
};

inline DropCfg make_drop_cfg(const void* state, const void* mask_in, uint32_t site, float p) {
  DropCfg d{};
  if ((state || mask_in) && p > 0.f) {
    d.state = static_cast<const RfRngState*>(state);
    d.mask_in = static_cast<const uint8_t*>(mask_in);
    d.site = site;
    const double t = (double)p * 4294967296.0;
    d.thresh = t >= 4294967295.0 ? 0xFFFFFFFFu : (uint32_t)t;
    d.scale = 1.0f / (1.0f - p);
  }
  return d;
}
