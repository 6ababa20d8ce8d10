// Row-local chains of a d_model = 64 decoder layer as ONE launch each way (the gaze-video PerceiveDecoder,
// cross_modal_transformer.py:304-365,436-476: masked-ProbSparse self attention -> norm1 -> full cross attention -> norm2 ->
// conv FFN -> norm3).  Everything between two attention launches is ROW-LOCAL:
//
//   a (attention output) -> out-projection + residual x -> LayerNorm            => x1
//                        [-> conv1 -> act -> conv2 + residual x1 -> LayerNorm   => y ]       (the FFN block, optional)
//                        [-> projection (the next attention's q, or q | k | v)  => proj]     (optional)
//
// so a workgroup (4 waves: wave w owns columns 16 w .. 16 w + 15 of a 64-wide row) takes 16 or 32 rows of the flattened
// (B L, 64) activations through the whole chain: 13 launches of a decoder layer become 5 (self attention, chain, k | v
// projection of the memory, cross attention, chain + FFN + the next layer's q | k | v).  At d = 64 every weight is a few
// KB: the bf16 MFMA B fragments are read straight from the fp32 masters in L2 (k-contiguous: two 16-B loads per lane and
// fragment; transposed for the backward: eight 4-B loads) -- no packed copies, no LDS staging of weights.  bf16 operands,
// fp32 accumulation, fp32 residual stream in registers: the arithmetic contract of the layer-by-layer bf16 path.
#include "seqlayer_common.h"

namespace {

constexpr int RC_D = 64, RC_NW = RC_D / 16, RC_NT = 64 * RC_NW, RC_XP = RC_D + 8, RC_FMAX = 256, RC_HP = RC_FMAX + 8;

struct RowChainFwdP {
  RfRowChain c;
  int M;
  DropCfg drop;  // nn.Dropout of the layer in train mode (state == null: off); sites c.drop_site + {0: out-projection output,
                 // 1: hidden activation, 2: conv2 output}, element index row * cols + col (the numbering of rf_dropout)
};
struct RowChainBwdP {
  RfRowChainBwd c;
  int M;
  DropCfg drop;
};

// B fragment of a k-contiguous fp32 weight w[n][k] (pitch ld): row n, columns k .. k + 7 (this lane's share of a 32-wide k-step)
__device__ __forceinline__ bf16x8 rc_wfrag(const float* __restrict__ w, int ld, int n, int k) {
  const float4* q = reinterpret_cast<const float4*>(w + (long)n * ld + k);
  return pack8(q[0], q[1]);
}
// the transposed use of the same weight (dX = dY W: contraction over w's rows): rows k .. k + 7 of column n
__device__ __forceinline__ bf16x8 rc_wfrag_t(const float* __restrict__ w, int ld, int k, int n) {
  bf16x8 o;
#pragma unroll
  for (int j = 0; j < 8; ++j) o[j] = (__bf16)w[(long)(k + j) * ld + n];
  return o;
}

// LayerNorm over the 64 columns of every row (columns spread over the 4 waves, MFMA accumulator layout); v becomes x-hat
template <int RT>
__device__ __forceinline__ void rc_layer_norm(f32x4 (&v)[RT], float* __restrict__ rstd_g, int L, float2* __restrict__ part,
                                              float2* __restrict__ stat, int wave, int lane, float eps) {
  const int fr = lane & 15, fq = lane >> 4;
#pragma unroll
  for (int rt = 0; rt < RT; ++rt)
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const float s1 = row16_sum(v[rt][r]), s2 = row16_sum(v[rt][r] * v[rt][r]);
      if (fr == 0) part[(rt * 16 + fq * 4 + r) * RC_NW + wave] = make_float2(s1, s2);
    }
  __syncthreads();
  if (threadIdx.x < 16 * RT) {
    const int row = threadIdx.x;
    float s1 = 0.f, s2 = 0.f;
#pragma unroll
    for (int w = 0; w < RC_NW; ++w) { s1 += part[row * RC_NW + w].x; s2 += part[row * RC_NW + w].y; }
    const float mean = s1 * (1.f / RC_D);
    const float rs = __builtin_amdgcn_rsqf(fmaxf(s2 * (1.f / RC_D) - mean * mean, 0.f) + eps);
    stat[row] = make_float2(mean, rs);
    if (rstd_g && row < L) rstd_g[row] = rs;
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

// LayerNorm backward (see stack_ln_bwd): in g = dy (rows >= L zero), out g = d pre-norm; dgamma / dbeta by atomics
template <int RT>
__device__ __forceinline__ void rc_ln_bwd(f32x4 (&g)[RT], const f32x4 (&xh)[RT], const float* __restrict__ rstd_g,
                                          float gamma, float* __restrict__ dgam, float* __restrict__ dbet, int L,
                                          float2* __restrict__ part, float4* __restrict__ stat, int wave, int lane) {
  const int fr = lane & 15, fq = lane >> 4;
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
      if (fr == 0) part[(rt * 16 + fq * 4 + r) * RC_NW + wave] = make_float2(s1, s2);
    }
  __syncthreads();
  if (threadIdx.x < 16 * RT) {
    const int row = threadIdx.x;
    float s1 = 0.f, s2 = 0.f;
#pragma unroll
    for (int w = 0; w < RC_NW; ++w) { s1 += part[row * RC_NW + w].x; s2 += part[row * RC_NW + w].y; }
    stat[row] = make_float4(s1 * (1.f / RC_D), s2 * (1.f / RC_D), row < L ? rstd_g[row] : 0.f, 0.f);
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

template <int RT, bool DROP>
__global__ __launch_bounds__(RC_NT) void rowchain_fwd_kernel(const RowChainFwdP pp) {
  constexpr int LP = 16 * RT;
  const RfRowChain& p = pp.c;
  uint2 dkey = make_uint2(0, 0);
  uint32_t dstep = 0;
  if constexpr (DROP) {
    const unsigned long long sd = pp.drop.state->seed;
    dkey = make_uint2((uint32_t)sd, (uint32_t)(sd >> 32));
    dstep = (uint32_t)pp.drop.state->step;
  }
  __shared__ __attribute__((aligned(16))) __bf16 xb[LP * RC_XP];   // A image of the 64-wide operand (a, then x1, then y)
  __shared__ __attribute__((aligned(16))) __bf16 hb[LP * RC_HP];   // hidden activation image
  __shared__ __attribute__((aligned(16))) float patch[RC_NW * 320];
  __shared__ float2 part[LP * RC_NW];
  __shared__ float2 stat[LP];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, fr = lane & 15, fq = lane >> 4;
  const long row0 = (long)blockIdx.x * LP;
  const int L = (int)min((long)LP, (long)pp.M - row0);
  const int F = p.d_ff, col = wave * 16 + fr;
  float* tb = patch + wave * 320;

  auto store_acc = [&](const f32x4 (&v)[RT], float* g, int ld, int c0) {
#pragma unroll
    for (int rt = 0; rt < RT; ++rt) tile_store(v[rt], tb, g + (row0 + rt * 16) * ld + c0, ld, L - rt * 16, lane);
  };

  // ---- first block: out-projection + residual + LayerNorm ----
  f32x4 xres[RT];
#pragma unroll
  for (int rt = 0; rt < RT; ++rt)
#pragma unroll
    for (int r = 0; r < 4; ++r) xres[rt][r] = p.x[(row0 + min(rt * 16 + fq * 4 + r, L - 1)) * RC_D + col];
  bf16x8 wfo[2];
#pragma unroll
  for (int kk = 0; kk < 2; ++kk) wfo[kk] = rc_wfrag(p.wo, RC_D, col, kk * 32 + fq * 8);
  const float bo = p.bo[col], g1 = p.g1[col], be1 = p.be1[col];
  // conv1's fragments (this wave's column tiles wave, wave + 4, ...: at most 4) are requested now, long before their use
  const int nct = p.w1 ? F / 16 : 0;
  bf16x8 wf1[4][2];
  float b1v[4];
#pragma unroll
  for (int it = 0; it < 4; ++it) {
    const int n = min(wave + it * RC_NW, max(nct - 1, 0)) * 16 + fr;
#pragma unroll
    for (int kk = 0; kk < 2; ++kk) wf1[it][kk] = p.w1 ? rc_wfrag(p.w1, RC_D, n, kk * 32 + fq * 8) : zero_frag();
    b1v[it] = p.w1 ? p.b1[n] : 0.f;
  }
  for (int i = tid; i < LP * (RC_D / 4); i += RC_NT) {
    const int row = i >> 4, c4 = (i & 15) * 4;
    const float4 v = *reinterpret_cast<const float4*>(p.a + (row0 + min(row, L - 1)) * RC_D + c4);
    const bf16x4 o = {(__bf16)v.x, (__bf16)v.y, (__bf16)v.z, (__bf16)v.w};
    *reinterpret_cast<bf16x4*>(xb + row * RC_XP + c4) = o;
  }
  __syncthreads();
  f32x4 v[RT];
#pragma unroll
  for (int rt = 0; rt < RT; ++rt) {
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int kk = 0; kk < 2; ++kk)
      acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ld_frag(xb + (rt * 16 + fr) * RC_XP + kk * 32 + fq * 8), wfo[kk], acc, 0, 0, 0);
    if constexpr (DROP) {
      const f32x4 f = drop_factors(pp.drop, dkey, dstep, (uint32_t)p.drop_site, row0 + rt * 16, RC_D, col, lane);
#pragma unroll
      for (int r = 0; r < 4; ++r) v[rt][r] = (acc[r] + bo) * f[r] + xres[rt][r];
    } else {
#pragma unroll
      for (int r = 0; r < 4; ++r) v[rt][r] = acc[r] + bo + xres[rt][r];
    }
  }
  rc_layer_norm<RT>(v, p.rstd1 ? p.rstd1 + row0 : nullptr, L, part, stat, wave, lane, p.eps);  // (barriers fence the a-image reads)
  if (p.xhat1) store_acc(v, p.xhat1, RC_D, wave * 16);
#pragma unroll
  for (int rt = 0; rt < RT; ++rt)
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const float y1 = v[rt][r] * g1 + be1;
      xres[rt][r] = y1;
      xb[(rt * 16 + fq * 4 + r) * RC_XP + col] = (__bf16)y1;
    }
  store_acc(xres, p.x1, RC_D, wave * 16);
  __syncthreads();  // x1 image complete

  // ---- FFN block: conv1 -> act -> conv2 + residual + LayerNorm ----
  if (p.w1) {
#pragma unroll
    for (int it = 0; it < 4; ++it) {
      const int ct = wave + it * RC_NW;
      if (ct >= nct) break;
      const int n = ct * 16 + fr;
      const float b1 = b1v[it];
      f32x4 zz[RT], hh[RT];
#pragma unroll
      for (int rt = 0; rt < RT; ++rt) {
        zz[rt] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int kk = 0; kk < 2; ++kk)
          zz[rt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ld_frag(xb + (rt * 16 + fr) * RC_XP + kk * 32 + fq * 8), wf1[it][kk], zz[rt], 0, 0, 0);
        f32x4 f = {1.f, 1.f, 1.f, 1.f};
        if constexpr (DROP) f = drop_factors(pp.drop, dkey, dstep, (uint32_t)(p.drop_site + 1), row0 + rt * 16, F, n, lane);
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          zz[rt][r] += b1;
          hh[rt][r] = p.act == RF_ACT_GELU ? sl_gelu(zz[rt][r]) : (p.act == RF_ACT_RELU ? fmaxf(zz[rt][r], 0.f) : zz[rt][r]);
          hh[rt][r] *= f[r];  // conv2 consumes (and the backward needs) the dropped activation
          hb[(rt * 16 + fq * 4 + r) * RC_HP + n] = (__bf16)hh[rt][r];
        }
      }
      if (p.z) store_acc(zz, p.z, F, ct * 16);
      if (p.h) store_acc(hh, p.h, F, ct * 16);
    }
    // conv2's fragments are requested before the barrier: one memory round trip, not one per k-step
    const int nk = F / 32;  // <= 8
    bf16x8 wf2[8];
#pragma unroll
    for (int kk = 0; kk < 8; ++kk) wf2[kk] = rc_wfrag(p.w2, F, col, min(kk, nk - 1) * 32 + fq * 8);
    const float b2 = p.b2[col], g2 = p.g2[col], be2 = p.be2[col];
    __syncthreads();  // hidden activation image complete
#pragma unroll
    for (int rt = 0; rt < RT; ++rt) v[rt] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int kk = 0; kk < 8; ++kk) {
      if (kk < nk) {
#pragma unroll
        for (int rt = 0; rt < RT; ++rt)
          v[rt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ld_frag(hb + (rt * 16 + fr) * RC_HP + kk * 32 + fq * 8), wf2[kk], v[rt], 0, 0, 0);
      }
    }
#pragma unroll
    for (int rt = 0; rt < RT; ++rt) {
      f32x4 f = {1.f, 1.f, 1.f, 1.f};
      if constexpr (DROP) f = drop_factors(pp.drop, dkey, dstep, (uint32_t)(p.drop_site + 2), row0 + rt * 16, RC_D, col, lane);
#pragma unroll
      for (int r = 0; r < 4; ++r) v[rt][r] = (v[rt][r] + b2) * f[r] + xres[rt][r];
    }
    rc_layer_norm<RT>(v, p.rstd2 ? p.rstd2 + row0 : nullptr, L, part, stat, wave, lane, p.eps);  // (barriers fence the x1 / h reads)
    if (p.xhat2) store_acc(v, p.xhat2, RC_D, wave * 16);
#pragma unroll
    for (int rt = 0; rt < RT; ++rt)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const float y2 = v[rt][r] * g2 + be2;
        xres[rt][r] = y2;
        xb[(rt * 16 + fq * 4 + r) * RC_XP + col] = (__bf16)y2;
      }
    store_acc(xres, p.y, RC_D, wave * 16);
    __syncthreads();  // y image complete
  }

  // ---- projection for the next attention launch ----
  if (p.wp) {
#pragma unroll 1
    for (int ct = wave; ct < p.n_proj / 16; ct += RC_NW) {
      const int n = ct * 16 + fr;
      bf16x8 wf[2];
#pragma unroll
      for (int kk = 0; kk < 2; ++kk) wf[kk] = rc_wfrag(p.wp, RC_D, n, kk * 32 + fq * 8);
      const float bp = p.bp ? p.bp[n] : 0.f;
      f32x4 acc[RT];
#pragma unroll
      for (int rt = 0; rt < RT; ++rt) {
        acc[rt] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int kk = 0; kk < 2; ++kk)
          acc[rt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ld_frag(xb + (rt * 16 + fr) * RC_XP + kk * 32 + fq * 8), wf[kk], acc[rt], 0, 0, 0);
#pragma unroll
        for (int r = 0; r < 4; ++r) acc[rt][r] += bp;
      }
      store_acc(acc, p.proj, p.n_proj, ct * 16);
    }
  }
}

template <int RT, bool DROP>
__global__ __launch_bounds__(RC_NT) void rowchain_bwd_kernel(const RowChainBwdP pp) {
  constexpr int LP = 16 * RT;
  const RfRowChainBwd& p = pp.c;
  uint2 dkey = make_uint2(0, 0);
  uint32_t dstep = 0;
  if constexpr (DROP) {
    const unsigned long long sd = pp.drop.state->seed;
    dkey = make_uint2((uint32_t)sd, (uint32_t)(sd >> 32));
    dstep = (uint32_t)pp.drop.state->step;
  }
  __shared__ __attribute__((aligned(16))) __bf16 xb[LP * RC_XP];   // 64-wide gradient images (A operands)
  __shared__ __attribute__((aligned(16))) __bf16 hb[LP * RC_HP];   // d proj image, then the dz image
  __shared__ __attribute__((aligned(16))) float patch[RC_NW * 320];
  __shared__ float2 part[LP * RC_NW];
  __shared__ float4 stat[LP];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, fr = lane & 15, fq = lane >> 4;
  const long row0 = (long)blockIdx.x * LP;
  const int L = (int)min((long)LP, (long)pp.M - row0);
  const int F = p.d_ff, col = wave * 16 + fr;
  float* tb = patch + wave * 320;

  auto store_acc = [&](const f32x4 (&v)[RT], float* g, int ld, int c0) {
#pragma unroll
    for (int rt = 0; rt < RT; ++rt) tile_store(v[rt], tb, g + (row0 + rt * 16) * ld + c0, ld, L - rt * 16, lane);
  };
  auto put_image = [&](const f32x4 (&v)[RT]) {
#pragma unroll
    for (int rt = 0; rt < RT; ++rt)
#pragma unroll
      for (int r = 0; r < 4; ++r) xb[(rt * 16 + fq * 4 + r) * RC_XP + col] = (__bf16)v[rt][r];
  };

  // everything the later phases read from global memory is requested now (one round trip instead of one per phase):
  // x-hat of both norms, conv2^T's fragments and the activation-gradient source of this wave's column tiles
  const int nct = p.w1 ? F / 16 : 0;
  f32x4 xh1[RT], xh2[RT];
#pragma unroll
  for (int rt = 0; rt < RT; ++rt)
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const long at = (row0 + min(rt * 16 + fq * 4 + r, L - 1)) * RC_D + col;
      xh1[rt][r] = p.xhat1[at];
      xh2[rt][r] = p.w1 ? p.xhat2[at] : 0.f;
    }
  bf16x8 wf2t[4][2];
  f32x4 zsv[4][RT];
#pragma unroll
  for (int it = 0; it < 4; ++it) {
    const int n = min(wave + it * RC_NW, max(nct - 1, 0)) * 16 + fr;
#pragma unroll
    for (int kk = 0; kk < 2; ++kk) wf2t[it][kk] = p.w1 ? rc_wfrag_t(p.w2, F, kk * 32 + fq * 8, n) : zero_frag();
#pragma unroll
    for (int rt = 0; rt < RT; ++rt)
#pragma unroll
      for (int r = 0; r < 4; ++r) zsv[it][rt][r] = p.w1 ? p.zsrc[(row0 + min(rt * 16 + fq * 4 + r, L - 1)) * F + n] : 0.f;
  }

  // ---- gradient of the chain's output: what arrives from elsewhere + the projection's share ----
  f32x4 g[RT];
#pragma unroll
  for (int rt = 0; rt < RT; ++rt)
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int row = rt * 16 + fq * 4 + r;
      const float x = p.dyin ? p.dyin[(row0 + min(row, L - 1)) * RC_D + col] : 0.f;
      g[rt][r] = row < L ? x : 0.f;
    }
  if (p.dproj) {
    const int NP = p.n_proj, v4 = NP / 4;
    for (int i = tid; i < LP * v4; i += RC_NT) {
      const int row = i / v4, c4 = (i - row * v4) * 4;
      float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
      if (row < L) v = *reinterpret_cast<const float4*>(p.dproj + (row0 + row) * NP + c4);
      const bf16x4 o = {(__bf16)v.x, (__bf16)v.y, (__bf16)v.z, (__bf16)v.w};
      *reinterpret_cast<bf16x4*>(hb + row * RC_HP + c4) = o;
    }
    const int nkp = NP / 32;  // <= 6
    bf16x8 wfp[6];
#pragma unroll
    for (int kk = 0; kk < 6; ++kk) wfp[kk] = rc_wfrag_t(p.wp, RC_D, min(kk, nkp - 1) * 32 + fq * 8, col);  // (one round trip)
    __syncthreads();
#pragma unroll
    for (int kk = 0; kk < 6; ++kk) {
      if (kk < nkp) {
#pragma unroll
        for (int rt = 0; rt < RT; ++rt)
          g[rt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ld_frag(hb + (rt * 16 + fr) * RC_HP + kk * 32 + fq * 8), wfp[kk], g[rt], 0, 0, 0);
      }
    }
#pragma unroll
    for (int rt = 0; rt < RT; ++rt)
#pragma unroll
      for (int r = 0; r < 4; ++r) g[rt][r] = rt * 16 + fq * 4 + r < L ? g[rt][r] : 0.f;
  }

  // ---- FFN block backward ----
  if (p.w1) {
    f32x4 res[RT];
    rc_ln_bwd<RT>(g, xh2, p.rstd2 + row0, p.g2[col], p.dg2 + col, p.db2 + col, L, part, stat, wave,
                  lane);  // (its first barrier also fences the d proj image reads)
#pragma unroll
    for (int rt = 0; rt < RT; ++rt) {
      res[rt] = g[rt];  // the skip carries the unmasked gradient
      if constexpr (DROP) {  // the conv pair sees it through the conv2-output dropout (and so does conv2's weight gradient)
        const f32x4 f = drop_factors(pp.drop, dkey, dstep, (uint32_t)(p.drop_site + 2), row0 + rt * 16, RC_D, col, lane);
#pragma unroll
        for (int r = 0; r < 4; ++r) g[rt][r] *= f[r];
      }
    }
    put_image(g);
    store_acc(g, p.dpre2, RC_D, wave * 16);
    __syncthreads();  // d pre-norm-2 image complete
    // conv2^T + activation': dz
#pragma unroll
    for (int it = 0; it < 4; ++it) {
      const int ct = wave + it * RC_NW;
      if (ct >= nct) break;
      const int n = ct * 16 + fr;
      f32x4 acc[RT];
#pragma unroll
      for (int rt = 0; rt < RT; ++rt) {
        acc[rt] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int kk = 0; kk < 2; ++kk)
          acc[rt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ld_frag(xb + (rt * 16 + fr) * RC_XP + kk * 32 + fq * 8), wf2t[it][kk], acc[rt], 0, 0, 0);
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const float zs = zsv[it][rt][r];
          const float d = p.act == RF_ACT_GELU ? sl_gelu_grad(zs) : (p.act == RF_ACT_RELU ? (zs > 0.f ? 1.f : 0.f) : 1.f);
          acc[rt][r] *= d;
        }
        if constexpr (DROP) {  // hidden-activation dropout
          const f32x4 f = drop_factors(pp.drop, dkey, dstep, (uint32_t)(p.drop_site + 1), row0 + rt * 16, F, n, lane);
#pragma unroll
          for (int r = 0; r < 4; ++r) acc[rt][r] *= f[r];
        }
#pragma unroll
        for (int r = 0; r < 4; ++r) hb[(rt * 16 + fq * 4 + r) * RC_HP + n] = (__bf16)acc[rt][r];
      }
      store_acc(acc, p.dz, F, ct * 16);
    }
    const int nk = F / 32;  // <= 8
    bf16x8 wf1t[8];
#pragma unroll
    for (int kk = 0; kk < 8; ++kk) wf1t[kk] = rc_wfrag_t(p.w1, RC_D, min(kk, nk - 1) * 32 + fq * 8, col);  // (one round trip)
    __syncthreads();  // dz image complete
    // conv1^T + skip
#pragma unroll
    for (int rt = 0; rt < RT; ++rt) g[rt] = res[rt];
#pragma unroll
    for (int kk = 0; kk < 8; ++kk) {
      if (kk < nk) {
#pragma unroll
        for (int rt = 0; rt < RT; ++rt)
          g[rt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ld_frag(hb + (rt * 16 + fr) * RC_HP + kk * 32 + fq * 8), wf1t[kk], g[rt], 0, 0, 0);
      }
    }
#pragma unroll
    for (int rt = 0; rt < RT; ++rt)
#pragma unroll
      for (int r = 0; r < 4; ++r) g[rt][r] = rt * 16 + fq * 4 + r < L ? g[rt][r] : 0.f;
  }

  // ---- first block backward: LayerNorm, out-projection^T ----
  rc_ln_bwd<RT>(g, xh1, p.rstd1 + row0, p.g1[col], p.dg1 + col, p.db1 + col, L, part, stat, wave, lane);
  if constexpr (DROP) {
    store_acc(g, p.dx, RC_D, wave * 16);  // the residual input sees the unmasked gradient
#pragma unroll
    for (int rt = 0; rt < RT; ++rt) {
      const f32x4 f = drop_factors(pp.drop, dkey, dstep, (uint32_t)p.drop_site, row0 + rt * 16, RC_D, col, lane);
#pragma unroll
      for (int r = 0; r < 4; ++r) g[rt][r] *= f[r];
    }
  }
  put_image(g);
  // without dropout: the gradient of the residual input AND the out-projection's weight-gradient operand
  store_acc(g, p.dpre1, RC_D, wave * 16);
  __syncthreads();
  {
    bf16x8 wf[2];
#pragma unroll
    for (int kk = 0; kk < 2; ++kk) wf[kk] = rc_wfrag_t(p.wo, RC_D, kk * 32 + fq * 8, col);
    f32x4 da[RT];
#pragma unroll
    for (int rt = 0; rt < RT; ++rt) {
      da[rt] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int kk = 0; kk < 2; ++kk)
        da[rt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ld_frag(xb + (rt * 16 + fr) * RC_XP + kk * 32 + fq * 8), wf[kk], da[rt], 0, 0, 0);
    }
    store_acc(da, p.da, RC_D, wave * 16);
  }
}

inline bool rc_al16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; }

}  // namespace

extern "C" int rf_rowchain_supported(int d_model, int d_ff, int n_proj) {
  return d_model == RC_D && (d_ff == 0 || (d_ff >= 32 && d_ff <= RC_FMAX && d_ff % 32 == 0)) && n_proj >= 0 && n_proj <= 192 &&
         n_proj % 32 == 0;
}

extern "C" int rf_rowchain_fwd(const RfRowChain* chain, int M, float drop_p, const void* rng_state, void* stream) {
  RF_REQUIRE(chain && M > 0);
  const RfRowChain& c = *chain;
  RF_REQUIRE(rf_rowchain_supported(c.d_model, c.w1 ? c.d_ff : 0, c.wp ? c.n_proj : 0));
  RF_REQUIRE(c.a && c.x && c.wo && c.bo && c.g1 && c.be1 && c.x1 && rc_al16(c.a) && rc_al16(c.wo) && rc_al16(c.x1));
  RF_REQUIRE(!c.w1 || (c.b1 && c.w2 && c.b2 && c.g2 && c.be2 && c.y && rc_al16(c.w1) && rc_al16(c.w2) && rc_al16(c.y)));
  RF_REQUIRE(!c.wp || (c.proj && rc_al16(c.wp) && rc_al16(c.proj)));
  RF_REQUIRE((!c.xhat1 || (c.rstd1 && rc_al16(c.xhat1))) && (!c.xhat2 || (c.rstd2 && rc_al16(c.xhat2))) &&
             (!c.z || rc_al16(c.z)) && (!c.h || rc_al16(c.h)));
  RF_REQUIRE(drop_p >= 0.f && drop_p < 1.f && (drop_p == 0.f || rng_state));
  RowChainFwdP p{c, M, make_drop_cfg(rng_state, nullptr, 0, drop_p)};
  const hipStream_t st = static_cast<hipStream_t>(stream);
  const bool drop = p.drop.state != nullptr;
  if (M <= 1024) {
    if (drop) RF_LAUNCH((rowchain_fwd_kernel<1, true>), dim3((M + 15) / 16), dim3(RC_NT), 0, st, p);
    else RF_LAUNCH((rowchain_fwd_kernel<1, false>), dim3((M + 15) / 16), dim3(RC_NT), 0, st, p);
  } else {
    if (drop) RF_LAUNCH((rowchain_fwd_kernel<2, true>), dim3((M + 31) / 32), dim3(RC_NT), 0, st, p);
    else RF_LAUNCH((rowchain_fwd_kernel<2, false>), dim3((M + 31) / 32), dim3(RC_NT), 0, st, p);
  }
  RF_CHECK_LAUNCH();
  return RF_OK;
}

extern "C" int rf_rowchain_bwd(const RfRowChainBwd* chain, int M, float drop_p, const void* rng_state, void* stream) {
  RF_REQUIRE(chain && M > 0);
  const RfRowChainBwd& c = *chain;
  RF_REQUIRE(rf_rowchain_supported(c.d_model, c.w1 ? c.d_ff : 0, c.dproj ? c.n_proj : 0));
  RF_REQUIRE((c.dproj || c.dyin) && (!c.dproj || (c.wp && rc_al16(c.dproj))));
  RF_REQUIRE(c.wo && c.g1 && c.xhat1 && c.rstd1 && c.dpre1 && c.da && c.dg1 && c.db1 && rc_al16(c.dpre1) && rc_al16(c.da));
  RF_REQUIRE(!c.w1 || (c.w2 && c.g2 && c.xhat2 && c.rstd2 && c.zsrc && c.dpre2 && c.dz && c.dg2 && c.db2 && rc_al16(c.dpre2) &&
                       rc_al16(c.dz)));
  RF_REQUIRE(drop_p >= 0.f && drop_p < 1.f && (drop_p == 0.f || rng_state));
  RowChainBwdP p{c, M, make_drop_cfg(rng_state, nullptr, 0, drop_p)};
  const hipStream_t st = static_cast<hipStream_t>(stream);
  const bool drop = p.drop.state != nullptr;
  RF_REQUIRE(!drop || (c.dx && rc_al16(c.dx)));
  if (M <= 1024) {
    if (drop) RF_LAUNCH((rowchain_bwd_kernel<1, true>), dim3((M + 15) / 16), dim3(RC_NT), 0, st, p);
    else RF_LAUNCH((rowchain_bwd_kernel<1, false>), dim3((M + 15) / 16), dim3(RC_NT), 0, st, p);
  } else {
    if (drop) RF_LAUNCH((rowchain_bwd_kernel<2, true>), dim3((M + 31) / 32), dim3(RC_NT), 0, st, p);
    else RF_LAUNCH((rowchain_bwd_kernel<2, false>), dim3((M + 31) / 32), dim3(RC_NT), 0, st, p);
  }
  RF_CHECK_LAUNCH();
  return RF_OK;
}
