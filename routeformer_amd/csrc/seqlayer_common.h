// Shared pieces of the fused per-sequence encoder stack kernels (seqlayer.hip: forward, seqlayer_bwd.hip: backward).
#pragma once
#include "common.h"
#include "philox.h"

namespace {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

constexpr int SL_D = 128, SL_H = 8, SL_E = 16, SL_NW = 8, SL_NT = 64 * SL_NW;
constexpr int SL_XP = SL_D + 8;  // bf16 pitch of the 128-column A images (272 B rows: conflict-light b128 reads)

__device__ __forceinline__ float sl_gelu(float x) {  // erf-GELU, Abramowitz-Stegun 7.1.26 (|erf err| <= 1.5e-7)
  const float u = fabsf(x) * 0.70710678118654752440f;
  const float t = __builtin_amdgcn_rcpf(fmaf(0.3275911f, u, 1.0f));
  float poly = fmaf(1.061405429f, t, -1.453152027f);
  poly = fmaf(poly, t, 1.421413741f);
  poly = fmaf(poly, t, -0.284496736f);
  poly = fmaf(poly, t, 0.254829592f);
  const float erf_abs = 1.0f - poly * t * __expf(-u * u);
  return 0.5f * x * (1.0f + copysignf(erf_abs, x));
}

__device__ __forceinline__ bf16x8 ld_frag(const __bf16* p) { return *reinterpret_cast<const bf16x8*>(p); }
__device__ __forceinline__ bf16x8 zero_frag() {
  bf16x8 z;
#pragma unroll
  for (int i = 0; i < 8; ++i) z[i] = (__bf16)0.f;
  return z;
}
// B fragment number `f` of a fragment-ordered weight: 64 lanes x 16 B contiguous
__device__ __forceinline__ bf16x8 ld_wfrag(const __bf16* base, int f, int lane) {
  return *reinterpret_cast<const bf16x8*>(base + ((long)f * 64 + lane) * 8);
}

__device__ __forceinline__ void wave_sync_lds() {
  __builtin_amdgcn_wave_barrier();
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
}

// One 16 x 16 fp32 tile in the MFMA accumulator layout (acc[r] = row 4 (lane >> 4) + r, column lane & 15) -> global rows
// g[row * ld + 0..15]: transposed through a wave-private LDS patch so that every lane issues ONE 16-B store (16 rows x
// 64 B per wave-instruction instead of four 4-B stores per lane).  tb: wave-private, 16 x 20 floats.
__device__ __forceinline__ void tile_store(const f32x4& acc, float* __restrict__ tb, float* __restrict__ g, int ld,
                                           int rows_valid, int lane) {
  const int fr = lane & 15, fq = lane >> 4;
#pragma unroll
  for (int r = 0; r < 4; ++r) tb[(fq * 4 + r) * 20 + fr] = acc[r];
  wave_sync_lds();
  const int rr = lane >> 2, c4 = (lane & 3) * 4;
  const float4 o = *reinterpret_cast<const float4*>(tb + rr * 20 + c4);
  if (rr < rows_valid) *reinterpret_cast<float4*>(g + rr * ld + c4) = o;
  wave_sync_lds();
}

// Dropout factors (0 or 1 / (1 - p)) of the four accumulator elements of this lane -- rows row0 + 4 (lane >> 4) + r,
// column `col` of a [rows, ncols] dropout site (element index row * ncols + col, the numbering rf_dropout uses, so the
// layer-by-layer backward regenerates the same mask).  One Philox call per lane and tile: the four lanes of a quad
// hold four neighbouring columns, i.e. the same Philox quad of every row; lane j of the quad evaluates row j and the
// keep-bits travel by DPP quad broadcasts.
__device__ __forceinline__ f32x4 drop_factors(const DropCfg& d, uint2 key, uint32_t step, uint32_t site, long row0, int ncols,
                                              int col, int lane) {
  const int fq = lane >> 4, j = lane & 3;
  const unsigned long long e = (unsigned long long)((row0 + fq * 4 + j) * ncols + (col & ~3));
  DropCfg c = d;
  c.site = site;
  const int bits = (int)drop_keep4(c, key, step, e >> 2);
  const int b0 = __builtin_amdgcn_update_dpp(0, bits, 0x00, 0xF, 0xF, true);
  const int b1 = __builtin_amdgcn_update_dpp(0, bits, 0x55, 0xF, 0xF, true);
  const int b2 = __builtin_amdgcn_update_dpp(0, bits, 0xAA, 0xF, 0xF, true);
  const int b3 = __builtin_amdgcn_update_dpp(0, bits, 0xFF, 0xF, 0xF, true);
  f32x4 f;
  f[0] = ((b0 >> j) & 1) ? d.scale : 0.f;
  f[1] = ((b1 >> j) & 1) ? d.scale : 0.f;
  f[2] = ((b2 >> j) & 1) ? d.scale : 0.f;
  f[3] = ((b3 >> j) & 1) ? d.scale : 0.f;
  return f;
}


// q | k | v save slab (fp32, or bf16: what the backward's matrix cores consume either way): four consecutive elements
__device__ __forceinline__ void store_qkv4(void* base, long off, const float4& v, int bf) {
  if (bf) {
    const bf16x4 o = {(__bf16)v.x, (__bf16)v.y, (__bf16)v.z, (__bf16)v.w};
    *reinterpret_cast<bf16x4*>(reinterpret_cast<__bf16*>(base) + off) = o;
  } else {
    *reinterpret_cast<float4*>(reinterpret_cast<float*>(base) + off) = v;
  }
}

__device__ __forceinline__ bf16x8 ld_qkv8(const void* base, long off, int bf) {  // eight consecutive elements as bf16
  if (bf) return *reinterpret_cast<const bf16x8*>(reinterpret_cast<const __bf16*>(base) + off);
  const float* f = reinterpret_cast<const float*>(base) + off;
  const float4 a = *reinterpret_cast<const float4*>(f), b = *reinterpret_cast<const float4*>(f + 4);
  bf16x8 o = {(__bf16)a.x, (__bf16)a.y, (__bf16)a.z, (__bf16)a.w, (__bf16)b.x, (__bf16)b.y, (__bf16)b.z, (__bf16)b.w};
  return o;
}
__device__ __forceinline__ float ld_save1(const void* base, long off, int bf) {  // one element of an fp32 / bf16 save slab
  return bf ? (float)reinterpret_cast<const __bf16*>(base)[off] : reinterpret_cast<const float*>(base)[off];
}
__device__ __forceinline__ __bf16 ld_qkv1(const void* base, long off, int bf) {
  return bf ? reinterpret_cast<const __bf16*>(base)[off] : (__bf16)reinterpret_cast<const float*>(base)[off];
}

// byte offsets inside a layer's packed blob
struct PackOff { long wqkv, wo, w1, w2, vec, lo, total; };
__host__ __device__ inline PackOff pack_offsets(int F) {
  PackOff o;
  o.wqkv = 0;
  o.wo = o.wqkv + 24L * 4 * 1024;          // 24 column tiles x 4 k-steps x 1 KB
  o.w1 = o.wo + 8L * 4 * 1024;
  o.w2 = o.w1 + (long)(F / 16) * 4 * 1024;
  o.vec = o.w2 + 8L * (F / 32) * 1024;     // fp32: bqkv[384] bo[128] b1[F] b2[128] g1 be1 g2 be2 [128 each]
  o.lo = (o.vec + (1152L + F) * 4 + 255) & ~255L;  // bf16 residuals W - bf16(W) of Wq | Wk: 16 column tiles x 4 k-steps
  o.total = o.lo + 16L * 4 * 1024;
  return o;
}

// x = hi + lo with hi = bf16(x), lo = bf16(x - hi): two bf16 numbers carry ~16 mantissa bits of x (split-bf16).  A product
// of two such operands is the sum of three bf16 MFMAs (hi hi + lo hi + hi lo; lo lo is below fp32 rounding).
__device__ __forceinline__ __bf16 bf16_lo(float x, __bf16 hi) { return (__bf16)(x - (float)hi); }

// LayerNorm over the 128 columns of every row, the columns of a row being spread over the 8 waves (16 each, MFMA
// accumulator layout: v[rt][r] = row 16 rt + 4 (lane >> 4) + r, column 16 wave + (lane & 15)).  Per-wave partial
// (sum, sum of squares) meet in part[row][wave]; one thread per row folds them into stat[row] = (mean, 1/sigma)
// (biased variance: nn.LayerNorm); two workgroup barriers.  v becomes x-hat.
template <int RT>
__device__ __forceinline__ void stack_layer_norm(f32x4 (&v)[RT], float* __restrict__ rstd_g, int L, float2* __restrict__ part,
                                                 float2* __restrict__ stat, int wave, int lane, float eps) {
  const int fr = lane & 15, fq = lane >> 4;
#pragma unroll
  for (int rt = 0; rt < RT; ++rt)
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const float s1 = row16_sum(v[rt][r]), s2 = row16_sum(v[rt][r] * v[rt][r]);
      if (fr == 0) part[(rt * 16 + fq * 4 + r) * SL_NW + wave] = make_float2(s1, s2);
    }
  __syncthreads();
  {
    const int row = wave * 64 + lane;
    if (row < 16 * RT) {
      float s1 = 0.f, s2 = 0.f;
#pragma unroll
      for (int w = 0; w < SL_NW; w += 2) {
        const float4 a = *reinterpret_cast<const float4*>(part + row * SL_NW + w);
        s1 += a.x + a.z;
        s2 += a.y + a.w;
      }
      const float mean = s1 * (1.f / 128.f);
      const float rs = __builtin_amdgcn_rsqf(fmaxf(s2 * (1.f / 128.f) - mean * mean, 0.f) + eps);
      stat[row] = make_float2(mean, rs);
      if (rstd_g && row < L) rstd_g[row] = rs;
    }
  }
  __syncthreads();
#pragma unroll
  for (int rt = 0; rt < RT; ++rt)
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const float2 st = stat[rt * 16 + fq * 4 + r];
      v[rt][r] = (v[rt][r] - st.x) * st.y;
    }
}


// ---- shared by the backward kernels (seqlayer_bwd.hip, enclayer.hip) ----
struct BwdPackOff { long w2t, w1t, wot, wqkvt, vec, total; };
__host__ __device__ inline BwdPackOff bwd_pack_offsets(int F) {
  BwdPackOff o;
  o.w2t = 0;                                   // B[k = d][n = f] = W2[d][f]:   F/16 column tiles x 4 k-steps
  o.w1t = o.w2t + (long)(F / 16) * 4 * 1024;   // B[k = f][n = d] = W1[f][d]:   8 column tiles x F/32 k-steps
  o.wot = o.w1t + 8L * (F / 32) * 1024;        // B[k = o][n = i] = Wo[o][i]:   8 x 4
  o.wqkvt = o.wot + 8L * 4 * 1024;             // B[k = j][n = d] = Wqkv[j][d]: 8 x 12
  o.vec = o.wqkvt + 8L * 12 * 1024;            // fp32: gamma1[128] gamma2[128]
  o.total = (o.vec + 256L * 4 + 255) & ~255L;
  return o;
}

__device__ __forceinline__ bf16x8 pack8(const float4& a, const float4& b) {
  bf16x8 o;
  o[0] = (__bf16)a.x; o[1] = (__bf16)a.y; o[2] = (__bf16)a.z; o[3] = (__bf16)a.w;
  o[4] = (__bf16)b.x; o[5] = (__bf16)b.y; o[6] = (__bf16)b.z; o[7] = (__bf16)b.w;
  return o;
}

__device__ __forceinline__ float sl_gelu_grad(float x) {  // Phi(x) + x phi(x), the forward's erf approximation
  const float u = fabsf(x) * 0.70710678118654752440f;
  const float t = __builtin_amdgcn_rcpf(fmaf(0.3275911f, u, 1.0f));
  float poly = fmaf(1.061405429f, t, -1.453152027f);
  poly = fmaf(poly, t, 1.421413741f);
  poly = fmaf(poly, t, -0.284496736f);
  poly = fmaf(poly, t, 0.254829592f);
  const float e = __expf(-u * u);
  const float cdf = 0.5f * (1.0f + copysignf(1.0f - poly * t * e, x));
  return fmaf(x * 0.39894228040143267794f, e, cdf);
}

// LayerNorm backward over the 128 columns of every row (columns spread over the 8 waves as in stack_layer_norm):
//   g = dy * gamma;  dx = rstd * (g - mean(g) - xhat * mean(g * xhat));  dgamma += sum_rows dy * xhat;  dbeta += sum_rows dy.
// In: g = dy (rows >= L hold zeros).  Out: g = dx (rows >= L stay zero).  Two workgroup barriers.
template <int RT>
__device__ __forceinline__ void stack_ln_bwd(f32x4 (&g)[RT], const float* __restrict__ xhat_base, long xhat_off, int xhat_bf16,
                                             const float* __restrict__ rstd_g,
                                             float gamma, float* __restrict__ dgam, float* __restrict__ dbet, int L,
                                             float2* __restrict__ part, float4* __restrict__ stat, int wave, int lane) {
  const int fr = lane & 15, fq = lane >> 4;
  f32x4 xh[RT];
#pragma unroll
  for (int rt = 0; rt < RT; ++rt)
#pragma unroll
    for (int r = 0; r < 4; ++r) xh[rt][r] = ld_save1(xhat_base, xhat_off + (long)min(rt * 16 + fq * 4 + r, L - 1) * SL_D, xhat_bf16);
  float dg = 0.f, db = 0.f;
#pragma unroll
  for (int rt = 0; rt < RT; ++rt)
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      dg = fmaf(g[rt][r], xh[rt][r], dg);
      db += g[rt][r];
    }
  dg += __shfl_xor(dg, 16); db += __shfl_xor(db, 16);
  dg += __shfl_xor(dg, 32); db += __shfl_xor(db, 32);
  if (fq == 0) {
    atomicAdd(dgam, dg);
    atomicAdd(dbet, db);
  }
#pragma unroll
  for (int rt = 0; rt < RT; ++rt)
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const float gm = g[rt][r] * gamma;
      g[rt][r] = gm;
      const float s1 = row16_sum(gm), s2 = row16_sum(gm * xh[rt][r]);
      if (fr == 0) part[(rt * 16 + fq * 4 + r) * SL_NW + wave] = make_float2(s1, s2);
    }
  __syncthreads();
  {
    const int row = wave * 64 + lane;
    if (row < 16 * RT) {
      float s1 = 0.f, s2 = 0.f;
#pragma unroll
      for (int w = 0; w < SL_NW; w += 2) {
        const float4 a = *reinterpret_cast<const float4*>(part + row * SL_NW + w);
        s1 += a.x + a.z;
        s2 += a.y + a.w;
      }
      stat[row] = make_float4(s1 * (1.f / 128.f), s2 * (1.f / 128.f), row < L ? rstd_g[row] : 0.f, 0.f);
    }
  }
  __syncthreads();
#pragma unroll
  for (int rt = 0; rt < RT; ++rt)
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const float4 st = stat[rt * 16 + fq * 4 + r];
      g[rt][r] = st.z * (g[rt][r] - st.x - xh[rt][r] * st.y);
    }
}

// rows 0..L-1 of a bf16 LDS image -> fp32 global rows of `cols` floats (16-B stores, the whole workgroup)
__device__ __forceinline__ void save_image(const __bf16* __restrict__ img, int pitch, int cols, float* __restrict__ dst, int L,
                                           int tid) {
  const int c4n = cols >> 2;
  for (int i = tid; i < L * c4n; i += SL_NT) {
    const int row = i / c4n, c4 = (i - row * c4n) * 4;
    const bf16x4 c = *reinterpret_cast<const bf16x4*>(img + row * pitch + c4);
    *reinterpret_cast<float4*>(dst + (long)row * cols + c4) = make_float4((float)c[0], (float)c[1], (float)c[2], (float)c[3]);
  }
}

// the same rows kept as bf16 (16-B copies; cols % 8 == 0, 16-B aligned rows on both sides)
__device__ __forceinline__ void save_image_bf16(const __bf16* __restrict__ img, int pitch, int cols, __bf16* __restrict__ dst, int L,
                                                int tid) {
  const int c8n = cols >> 3;
  for (int i = tid; i < L * c8n; i += SL_NT) {
    const int row = i / c8n, c8 = (i - row * c8n) * 8;
    *reinterpret_cast<uint4*>(dst + (long)row * cols + c8) = *reinterpret_cast<const uint4*>(img + row * pitch + c8);
  }
}
// dst is [rows][cols] fp32 or (as_bf16) bf16, `off` its element offset
__device__ __forceinline__ void save_image_as(const __bf16* __restrict__ img, int pitch, int cols, float* __restrict__ dst, long off,
                                              int L, int tid, bool as_bf16) {
  if (as_bf16) save_image_bf16(img, pitch, cols, reinterpret_cast<__bf16*>(dst) + off, L, tid);
  else save_image(img, pitch, cols, dst + off, L, tid);
}

}  // namespace
