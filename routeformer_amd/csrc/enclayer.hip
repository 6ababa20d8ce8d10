// Row-tile half of a ProbSparse encoder layer for sequences too long for the one-workgroup-per-sequence stack
// (csrc/seqlayer.hip stops at L = 80; the fusion `video_encoder` of routeformer.py:85-92,346 runs L = 160 / 320):
//
//   attention output of layer l  ->  out-projection + residual + LayerNorm1 -> conv1 -> act -> conv2 + residual + LayerNorm2
//                                ->  packed q | k | v projection of layer l + 1              (cross_modal_transformer.py:288-301)
//
// is ROW-LOCAL, so a workgroup (8 waves) owns 16 RT rows of the flattened (B L, 128) activations, whatever sequence they
// belong to; the only step that needs a whole sequence -- ProbSparse attention -- stays its own launch (rf_attn_fwd) between
// two of these.  A layer is 2 launches instead of 4 (QKV row-block, attention, out-projection + LN row-block, FFN + LN
// row-block), for any L and any batch, and the activations between out-projection and the next layer's q | k | v never leave
// the chip.  Same arithmetic contract and the same packed weight blobs as the fused stack: bf16 matrix-core operands
// (fragment-ordered weights read straight from global / L2), fp32 accumulation, fp32 residual stream in registers (wave w
// holds columns 16 w .. 16 w + 15 of every row), q and k in split-bf16 (x = hi + lo, W = hi + lo: three MFMAs per product)
// because they feed the discontinuous top-u selection.  Training mode stores what the layer-by-layer backward consumes
// (x-hat / 1/sigma of both norms, x1, z, h, the layer output) with the meaning of the unfused forward's saved tensors.
#include "seqlayer_common.h"

namespace {

struct EncTileP {
  const float* ctx;            // (M, 128) attention output of this layer; null: projection-only launch (first layer's q | k | v)
  const float* x;              // (M, 128) input of this layer (residual stream)
  const unsigned char* wl;     // this layer's packed blob (pack_offsets)
  const unsigned char* wnext;  // next layer's blob (its [Wq; Wk; Wv], low halves, bias) or null after the last layer
  float* y;                    // (M, 128) layer output
  float* qkv_next;             // (M, 384) packed q | k | v for the next attention launch
  float *xhat1, *rstd1, *x1, *z, *h, *xhat2, *rstd2;  // training saves (M, width) or null
  int M, F, act;
  float eps;
  DropCfg drop;    // nn.Dropout of the layer in train mode (cross_modal_transformer.py:295,298,299); state == null: off
  int drop_site;   // sites drop_site + {0: attention output, 1: hidden activation, 2: conv2 output}, element = row * cols + col
};

template <int RT, bool SAVE, bool DROP>
__global__ __launch_bounds__(SL_NT) void enc_tile_fwd_kernel(const EncTileP p) {
  constexpr int LP = 16 * RT, TB = 320;  // TB: floats of one staged 16 x 16 tile (pitch 20)
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  __bf16* xb = reinterpret_cast<__bf16*>(smem);                 // [LP][SL_XP]  ctx / x1 / y as MFMA A operand
  __bf16* hb = xb + LP * SL_XP;                                 // [LP][F + 8]  hidden activation; later the low half of y
  const int F = p.F, HP = F + 8;
  float* stage_base = reinterpret_cast<float*>(hb + LP * HP);   // per wave RT staged tiles
  float2* part = reinterpret_cast<float2*>(stage_base + SL_NW * RT * TB);
  float2* stat = part + LP * SL_NW;
  __bf16* xlo = hb;  // (pitch SL_XP <= HP)

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, fr = lane & 15, fq = lane >> 4;
  const int srow = lane >> 2, sc4 = (lane & 3) * 4;
  const long row0 = (long)blockIdx.x * LP;
  const int L = (int)min((long)LP, (long)p.M - row0);  // valid rows of this tile
  float* sc_f = stage_base + wave * RT * TB;
  const PackOff po = pack_offsets(F);
  uint2 dkey = make_uint2(0, 0);
  uint32_t dstep = 0;
  if constexpr (DROP) {
    const unsigned long long sd = p.drop.state->seed;
    dkey = make_uint2((uint32_t)sd, (uint32_t)(sd >> 32));
    dstep = (uint32_t)p.drop.state->step;
  }

  // one 16 x 16 fp32 tile set (RT tiles, accumulator layout) -> global rows, 16-B stores through the wave's staging patch
  auto store_tiles = [&](const f32x4 (&v)[RT], float* g, int ld) {
#pragma unroll
    for (int rt = 0; rt < RT; ++rt)
#pragma unroll
      for (int r = 0; r < 4; ++r) sc_f[rt * TB + (fq * 4 + r) * 20 + fr] = v[rt][r];
    wave_sync_lds();
#pragma unroll
    for (int rt = 0; rt < RT; ++rt)
      if (rt * 16 + srow < L)
        *reinterpret_cast<float4*>(g + (long)(rt * 16 + srow) * ld + sc4) = *reinterpret_cast<const float4*>(sc_f + rt * TB + srow * 20 + sc4);
    wave_sync_lds();
  };

  // ---- residual stream slice of this wave ----
  f32x4 xres[RT];
  {
    const float* xg = p.x + row0 * SL_D + wave * 16 + fr;
#pragma unroll
    for (int rt = 0; rt < RT; ++rt)
#pragma unroll
      for (int r = 0; r < 4; ++r) xres[rt][r] = xg[(long)min(rt * 16 + fq * 4 + r, L - 1) * SL_D];
  }

  if (p.ctx) {
    const unsigned char* wl = p.wl;
    const __bf16* w_o = reinterpret_cast<const __bf16*>(wl + po.wo);
    const __bf16* w_1 = reinterpret_cast<const __bf16*>(wl + po.w1);
    const __bf16* w_2 = reinterpret_cast<const __bf16*>(wl + po.w2);
    const float* vec = reinterpret_cast<const float*>(wl + po.vec);
    bf16x8 wfo[4];
#pragma unroll
    for (int kk = 0; kk < 4; ++kk) wfo[kk] = ld_wfrag(w_o, wave * 4 + kk, lane);
    // ---- attention output of the tile -> bf16 A image (coalesced 16-B loads) ----
    for (int i = tid; i < LP * (SL_D / 4); i += SL_NT) {
      const int row = i >> 5, c4 = (i & 31) * 4;
      const float4 v = *reinterpret_cast<const float4*>(p.ctx + (row0 + min(row, L - 1)) * SL_D + c4);
      const bf16x4 o = {(__bf16)v.x, (__bf16)v.y, (__bf16)v.z, (__bf16)v.w};
      *reinterpret_cast<bf16x4*>(xb + row * SL_XP + c4) = o;
    }
    __syncthreads();

    // ================= out-projection + residual + LayerNorm 1 (wave = 16 output columns) =================
    {
      const int col = wave * 16 + fr;
      const float bo = vec[384 + col], g1 = vec[640 + F + col], be1 = vec[768 + F + col];
      f32x4 v[RT];
#pragma unroll
      for (int rt = 0; rt < RT; ++rt) {
        f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int kk = 0; kk < 4; ++kk)
          acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ld_frag(xb + (rt * 16 + fr) * SL_XP + kk * 32 + fq * 8), wfo[kk], acc, 0, 0, 0);
        if constexpr (DROP) {
          const f32x4 f = drop_factors(p.drop, dkey, dstep, (uint32_t)p.drop_site, row0 + rt * 16, SL_D, col, lane);
#pragma unroll
          for (int r = 0; r < 4; ++r) v[rt][r] = (acc[r] + bo) * f[r] + xres[rt][r];
        } else {
#pragma unroll
          for (int r = 0; r < 4; ++r) v[rt][r] = acc[r] + bo + xres[rt][r];
        }
      }
      stack_layer_norm<RT>(v, SAVE ? p.rstd1 + row0 : nullptr, L, part, stat, wave, lane, p.eps);  // (barriers fence the ctx reads)
      if (SAVE) store_tiles(v, p.xhat1 + row0 * SL_D + wave * 16, SL_D);
#pragma unroll
      for (int rt = 0; rt < RT; ++rt)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const float y1 = v[rt][r] * g1 + be1;
          xres[rt][r] = y1;
          xb[(rt * 16 + fq * 4 + r) * SL_XP + col] = (__bf16)y1;
        }
      if (SAVE) store_tiles(xres, p.x1 + row0 * SL_D + wave * 16, SL_D);
    }
    __syncthreads();  // x1 image complete

    // ================= conv1 + activation (wave = column tiles wave, wave + 8, ...) =================
    {
      float* z_g = (SAVE && p.z) ? p.z + row0 * F : nullptr;
      float* h_g = SAVE ? p.h + row0 * F : nullptr;
#pragma unroll 1
      for (int ct = wave; ct < F / 16; ct += SL_NW) {
        bf16x8 wf1[4];
#pragma unroll
        for (int kk = 0; kk < 4; ++kk) wf1[kk] = ld_wfrag(w_1, ct * 4 + kk, lane);
        const int col = ct * 16 + fr;
        const float b1 = vec[512 + col];
        f32x4 zz[RT], hh[RT];
#pragma unroll
        for (int rt = 0; rt < RT; ++rt) {
          zz[rt] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
          for (int kk = 0; kk < 4; ++kk)
            zz[rt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ld_frag(xb + (rt * 16 + fr) * SL_XP + kk * 32 + fq * 8), wf1[kk], zz[rt], 0, 0, 0);
        }
#pragma unroll
        for (int rt = 0; rt < RT; ++rt)
#pragma unroll
          for (int r = 0; r < 4; ++r) zz[rt][r] += b1;
        if (p.act == RF_ACT_GELU) {
#pragma unroll
          for (int rt = 0; rt < RT; ++rt)
#pragma unroll
            for (int r = 0; r < 4; ++r) hh[rt][r] = sl_gelu(zz[rt][r]);
        } else {
#pragma unroll
          for (int rt = 0; rt < RT; ++rt)
#pragma unroll
            for (int r = 0; r < 4; ++r) hh[rt][r] = p.act == RF_ACT_RELU ? fmaxf(zz[rt][r], 0.f) : zz[rt][r];
        }
        if constexpr (DROP) {  // conv2 consumes (and the backward needs) the dropped activation
#pragma unroll
          for (int rt = 0; rt < RT; ++rt) {
            const f32x4 f = drop_factors(p.drop, dkey, dstep, (uint32_t)(p.drop_site + 1), row0 + rt * 16, F, col, lane);
#pragma unroll
            for (int r = 0; r < 4; ++r) hh[rt][r] *= f[r];
          }
        }
#pragma unroll
        for (int rt = 0; rt < RT; ++rt)
#pragma unroll
          for (int r = 0; r < 4; ++r) hb[(rt * 16 + fq * 4 + r) * HP + col] = (__bf16)hh[rt][r];
        if (z_g) store_tiles(zz, z_g + ct * 16, F);
        if (h_g) store_tiles(hh, h_g + ct * 16, F);
      }
    }
    __syncthreads();  // hidden activation image complete

    // ================= conv2 + residual + LayerNorm 2 =================
    {
      const int col = wave * 16 + fr;
      const float b2 = vec[512 + F + col], g2 = vec[896 + F + col], be2 = vec[1024 + F + col];
      f32x4 v[RT];
#pragma unroll
      for (int rt = 0; rt < RT; ++rt) v[rt] = f32x4{0.f, 0.f, 0.f, 0.f};
      const int nk = F / 32;  // <= 8
      bf16x8 wf2[8];
#pragma unroll
      for (int kk = 0; kk < 8; ++kk) wf2[kk] = ld_wfrag(w_2, wave * nk + min(kk, nk - 1), lane);
#pragma unroll
      for (int kk = 0; kk < 8; ++kk) {
        if (kk < nk) {
#pragma unroll
          for (int rt = 0; rt < RT; ++rt)
            v[rt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ld_frag(hb + (rt * 16 + fr) * HP + kk * 32 + fq * 8), wf2[kk], v[rt], 0, 0, 0);
        }
      }
      if constexpr (DROP) {
#pragma unroll
        for (int rt = 0; rt < RT; ++rt) {
          const f32x4 f = drop_factors(p.drop, dkey, dstep, (uint32_t)(p.drop_site + 2), row0 + rt * 16, SL_D, col, lane);
#pragma unroll
          for (int r = 0; r < 4; ++r) v[rt][r] = (v[rt][r] + b2) * f[r] + xres[rt][r];
        }
      } else {
#pragma unroll
        for (int rt = 0; rt < RT; ++rt)
#pragma unroll
          for (int r = 0; r < 4; ++r) v[rt][r] += b2 + xres[rt][r];
      }
      stack_layer_norm<RT>(v, SAVE ? p.rstd2 + row0 : nullptr, L, part, stat, wave, lane, p.eps);  // (barriers fence the hb / xb reads)
      if (SAVE) store_tiles(v, p.xhat2 + row0 * SL_D + wave * 16, SL_D);
#pragma unroll
      for (int rt = 0; rt < RT; ++rt)
#pragma unroll
        for (int r = 0; r < 4; ++r) xres[rt][r] = v[rt][r] * g2 + be2;
      store_tiles(xres, p.y + row0 * SL_D + wave * 16, SL_D);
    }
  }
  if (!p.wnext) return;

  // ================= packed q | k | v projection of the next layer on this tile's rows =================
  // (hi / lo images of the residual stream: q and k feed the ProbSparse selection -> split-bf16, v plain bf16)
  {
    const int col = wave * 16 + fr;
#pragma unroll
    for (int rt = 0; rt < RT; ++rt)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const __bf16 hi = (__bf16)xres[rt][r];
        xb[(rt * 16 + fq * 4 + r) * SL_XP + col] = hi;
        xlo[(rt * 16 + fq * 4 + r) * SL_XP + col] = bf16_lo(xres[rt][r], hi);
      }
  }
  const unsigned char* wn = p.wnext;
  const __bf16* w_qkv = reinterpret_cast<const __bf16*>(wn + po.wqkv);
  const __bf16* w_lo = reinterpret_cast<const __bf16*>(wn + po.lo);
  const float* vecn = reinterpret_cast<const float*>(wn + po.vec);
  bf16x8 wf[3][4], wlo[2][4];
  float bias[3];
#pragma unroll
  for (int pt = 0; pt < 3; ++pt) {
#pragma unroll
    for (int kk = 0; kk < 4; ++kk) wf[pt][kk] = ld_wfrag(w_qkv, (pt * 8 + wave) * 4 + kk, lane);
    bias[pt] = vecn[pt * SL_D + wave * 16 + fr];
  }
#pragma unroll
  for (int pt = 0; pt < 2; ++pt)
#pragma unroll
    for (int kk = 0; kk < 4; ++kk) wlo[pt][kk] = ld_wfrag(w_lo, (pt * 8 + wave) * 4 + kk, lane);
  __syncthreads();  // hi / lo images complete
  f32x4 acc[3][RT];
#pragma unroll
  for (int rt = 0; rt < RT; ++rt) {
    bf16x8 a[4], al[4];
#pragma unroll
    for (int kk = 0; kk < 4; ++kk) {
      a[kk] = ld_frag(xb + (rt * 16 + fr) * SL_XP + kk * 32 + fq * 8);
      al[kk] = ld_frag(xlo + (rt * 16 + fr) * SL_XP + kk * 32 + fq * 8);
    }
#pragma unroll
    for (int pt = 0; pt < 3; ++pt) {
      f32x4 c = {0.f, 0.f, 0.f, 0.f};
      if (pt < 2) {
#pragma unroll
        for (int kk = 0; kk < 4; ++kk) {
          c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(al[kk], wf[pt][kk], c, 0, 0, 0);
          c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[kk], wlo[pt][kk], c, 0, 0, 0);
        }
      }
#pragma unroll
      for (int kk = 0; kk < 4; ++kk) c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[kk], wf[pt][kk], c, 0, 0, 0);
#pragma unroll
      for (int r = 0; r < 4; ++r) c[r] += bias[pt];
      acc[pt][rt] = c;
    }
  }
#pragma unroll
  for (int pt = 0; pt < 3; ++pt) store_tiles(acc[pt], p.qkv_next + row0 * (3 * SL_D) + pt * SL_D + wave * 16, 3 * SL_D);
}

// ---------------------------------------------------------------------------------------------------------------------
// Backward counterpart: between two attention-backward launches ONE row-tile launch does
//     packed q | k | v projection^T of layer l + 1 (+ its skip gradient)  =  gradient of layer l's output
//     -> LayerNorm-2 backward -> conv2^T -> activation' -> conv1^T + skip -> LayerNorm-1 backward -> out-projection^T
// and hands rf_attn_bwd the gradient of layer l's attention output: 2 launches per layer instead of 5.  Weights: the
// TRANSPOSED fragment blobs of the fused stack's backward (rf_seqlayer_pack, transpose = 1).  What leaves the launch per
// layer: d pre-norm-2 and dz (bf16-rounded: weight-gradient operands only), d pre-norm-1 in full fp32 (it is also the skip
// gradient the next launch adds), the LayerNorm parameter gradients (atomics), d ctx.
// ---------------------------------------------------------------------------------------------------------------------
struct EncTileBwdP {
  const float* dy;             // (M, 128) gradient of this layer's output (last layer), or null with dqkv
  const float* dqkv;           // (M, 384) gradient of the NEXT layer's packed q | k | v, or null
  const float* skip;           // (M, 128) d pre-norm-1 of the next layer (with dqkv)
  const unsigned char* wl;     // this layer's transposed blob; null: projection^T only (the stack's input gradient)
  const unsigned char* wnext;  // next layer's transposed blob (with dqkv)
  const float *xhat1, *rstd1, *zsrc, *xhat2, *rstd2;
  float *dpre2, *dz, *dpre1, *dctx, *dx;
  float *dg1, *db1, *dg2, *db2;
  int M, F, act;
  float* dskip;    // DROP: (M, 128) the UNMASKED d pre-norm-1 (the next launch's skip gradient; dpre1 then holds the gradient
                   // behind the attention-output dropout, what the out-projection's weight gradient consumes)
  DropCfg drop;    // the forward's masks are regenerated from (seed, step, site, element); state == null: off
  int drop_site;
};

template <int RT, bool DROP>
__global__ __launch_bounds__(SL_NT) void enc_tile_bwd_kernel(const EncTileBwdP p) {
  constexpr int LP = 16 * RT, QP = 3 * SL_D + 8, TB = 320;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  __bf16* xb = reinterpret_cast<__bf16*>(smem);
  __bf16* hb = xb + LP * SL_XP;
  __bf16* dqi = xb;  // projection phase alias of xb + hb
  const int F = p.F, HP = F + 8;
  const int region = max(LP * SL_XP + LP * HP, LP * QP) * 2;
  float* stage_base = reinterpret_cast<float*>(smem + ((region + 15) & ~15));
  float2* part = reinterpret_cast<float2*>(stage_base + SL_NW * RT * TB);
  float4* stat = reinterpret_cast<float4*>(part + LP * SL_NW);

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, fr = lane & 15, fq = lane >> 4;
  const int srow = lane >> 2, sc4 = (lane & 3) * 4, col = wave * 16 + fr;
  const long row0 = (long)blockIdx.x * LP;
  const int L = (int)min((long)LP, (long)p.M - row0);
  float* sc_f = stage_base + wave * RT * TB;
  const BwdPackOff po = bwd_pack_offsets(F);
  uint2 dkey = make_uint2(0, 0);
  uint32_t dstep = 0;
  if constexpr (DROP) {
    if (p.drop.state) {  // (the projection-only launch of a dropout stack carries no state)
      const unsigned long long sd = p.drop.state->seed;
      dkey = make_uint2((uint32_t)sd, (uint32_t)(sd >> 32));
      dstep = (uint32_t)p.drop.state->step;
    }
  }

  auto store_tiles = [&](const f32x4 (&v)[RT], float* g, int ld) {
#pragma unroll
    for (int rt = 0; rt < RT; ++rt)
#pragma unroll
      for (int r = 0; r < 4; ++r) sc_f[rt * TB + (fq * 4 + r) * 20 + fr] = v[rt][r];
    wave_sync_lds();
#pragma unroll
    for (int rt = 0; rt < RT; ++rt)
      if (rt * 16 + srow < L)
        *reinterpret_cast<float4*>(g + (long)(rt * 16 + srow) * ld + sc4) = *reinterpret_cast<const float4*>(sc_f + rt * TB + srow * 20 + sc4);
    wave_sync_lds();
  };
  auto load_tile = [&](const float* g, f32x4 (&v)[RT]) {  // (M, 128) rows of this tile, accumulator layout, rows >= L -> 0
#pragma unroll
    for (int rt = 0; rt < RT; ++rt)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int row = rt * 16 + fq * 4 + r;
        const float x = g[(row0 + min(row, L - 1)) * SL_D + col];
        v[rt][r] = row < L ? x : 0.f;
      }
  };

  f32x4 dyres[RT];
  if (p.dqkv) {
    // ================= packed q | k | v projection^T of the next layer + its skip gradient =================
    const __bf16* w_qkvt = reinterpret_cast<const __bf16*>(p.wnext + po.wqkvt);
    bf16x8 wfp[12];
#pragma unroll
    for (int kk = 0; kk < 12; ++kk) wfp[kk] = ld_wfrag(w_qkvt, wave * 12 + kk, lane);
    for (int i = tid; i < LP * (3 * SL_D / 4); i += SL_NT) {
      const int row = i / 96, c4 = (i - row * 96) * 4;
      float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
      if (row < L) v = *reinterpret_cast<const float4*>(p.dqkv + (row0 + row) * (3 * SL_D) + c4);
      const bf16x4 o = {(__bf16)v.x, (__bf16)v.y, (__bf16)v.z, (__bf16)v.w};
      *reinterpret_cast<bf16x4*>(dqi + row * QP + c4) = o;
    }
    load_tile(p.skip, dyres);
    __syncthreads();
#pragma unroll
    for (int rt = 0; rt < RT; ++rt) {
      f32x4 acc = dyres[rt];
#pragma unroll
      for (int kk = 0; kk < 12; ++kk)
        acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ld_frag(dqi + (rt * 16 + fr) * QP + kk * 32 + fq * 8), wfp[kk], acc, 0, 0, 0);
#pragma unroll
      for (int r = 0; r < 4; ++r) dyres[rt][r] = rt * 16 + fq * 4 + r < L ? acc[r] : 0.f;
    }
  } else {
    load_tile(p.dy, dyres);
  }
  if (!p.wl) {  // the stack's input gradient
    store_tiles(dyres, p.dx + row0 * SL_D + wave * 16, SL_D);
    return;
  }
  const unsigned char* wl = p.wl;
  const __bf16* w_2t = reinterpret_cast<const __bf16*>(wl + po.w2t);
  const __bf16* w_1t = reinterpret_cast<const __bf16*>(wl + po.w1t);
  const __bf16* w_ot = reinterpret_cast<const __bf16*>(wl + po.wot);
  const float* vec = reinterpret_cast<const float*>(wl + po.vec);
  f32x4 res[RT];

  // ================= norm2 backward (its first barrier also fences the dq image reads above) =================
  stack_ln_bwd<RT>(dyres, p.xhat2, (long)row0 * SL_D + col, 0, p.rstd2 + row0, vec[SL_D + col], p.dg2 + col, p.db2 + col, L, part, stat,
                   wave, lane);
#pragma unroll
  for (int rt = 0; rt < RT; ++rt) {
    res[rt] = dyres[rt];
    f32x4 m = dyres[rt];
    if constexpr (DROP) {  // the conv pair sees the gradient through the conv2-output dropout; the skip does not
      const f32x4 f = drop_factors(p.drop, dkey, dstep, (uint32_t)(p.drop_site + 2), row0 + rt * 16, SL_D, col, lane);
#pragma unroll
      for (int r = 0; r < 4; ++r) m[r] *= f[r];
    }
#pragma unroll
    for (int r = 0; r < 4; ++r) xb[(rt * 16 + fq * 4 + r) * SL_XP + col] = (__bf16)m[r];
  }
  __syncthreads();  // d pre-norm-2 image complete
  save_image(xb, SL_XP, SL_D, p.dpre2 + row0 * SL_D, L, tid);

  // ================= conv2^T + activation' : dz =================
  {
    const float* zs = p.zsrc + row0 * F;
#pragma unroll 1
    for (int ct = wave; ct < F / 16; ct += SL_NW) {
      bf16x8 wf[4];
#pragma unroll
      for (int kk = 0; kk < 4; ++kk) wf[kk] = ld_wfrag(w_2t, ct * 4 + kk, lane);
      f32x4 zz[RT], acc[RT];
#pragma unroll
      for (int rt = 0; rt < RT; ++rt)
#pragma unroll
        for (int r = 0; r < 4; ++r) zz[rt][r] = zs[(long)min(rt * 16 + fq * 4 + r, L - 1) * F + ct * 16 + fr];
#pragma unroll
      for (int rt = 0; rt < RT; ++rt) {
        acc[rt] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int kk = 0; kk < 4; ++kk)
          acc[rt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ld_frag(xb + (rt * 16 + fr) * SL_XP + kk * 32 + fq * 8), wf[kk], acc[rt], 0, 0, 0);
      }
      if (p.act == RF_ACT_GELU) {
#pragma unroll
        for (int rt = 0; rt < RT; ++rt)
#pragma unroll
          for (int r = 0; r < 4; ++r) acc[rt][r] *= sl_gelu_grad(zz[rt][r]);
      } else if (p.act == RF_ACT_RELU) {
#pragma unroll
        for (int rt = 0; rt < RT; ++rt)
#pragma unroll
          for (int r = 0; r < 4; ++r) acc[rt][r] = zz[rt][r] > 0.f ? acc[rt][r] : 0.f;
      }
      if constexpr (DROP) {  // hidden-activation dropout
#pragma unroll
        for (int rt = 0; rt < RT; ++rt) {
          const f32x4 f = drop_factors(p.drop, dkey, dstep, (uint32_t)(p.drop_site + 1), row0 + rt * 16, F, ct * 16 + fr, lane);
#pragma unroll
          for (int r = 0; r < 4; ++r) acc[rt][r] *= f[r];
        }
      }
#pragma unroll
      for (int rt = 0; rt < RT; ++rt)
#pragma unroll
        for (int r = 0; r < 4; ++r) hb[(rt * 16 + fq * 4 + r) * HP + ct * 16 + fr] = (__bf16)acc[rt][r];
    }
  }
  __syncthreads();  // dz image complete
  save_image(hb, HP, F, p.dz + row0 * F, L, tid);

  // ================= conv1^T + skip, norm1 backward =================
  {
    const int nk = F / 32;  // <= 8
    bf16x8 wf[8];
#pragma unroll
    for (int kk = 0; kk < 8; ++kk) wf[kk] = ld_wfrag(w_1t, wave * nk + min(kk, nk - 1), lane);
#pragma unroll
    for (int rt = 0; rt < RT; ++rt) dyres[rt] = res[rt];
#pragma unroll
    for (int kk = 0; kk < 8; ++kk) {
      if (kk < nk) {
#pragma unroll
        for (int rt = 0; rt < RT; ++rt)
          dyres[rt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ld_frag(hb + (rt * 16 + fr) * HP + kk * 32 + fq * 8), wf[kk], dyres[rt], 0, 0, 0);
      }
    }
  }
  stack_ln_bwd<RT>(dyres, p.xhat1, (long)row0 * SL_D + col, 0, p.rstd1 + row0, vec[col], p.dg1 + col, p.db1 + col, L, part, stat, wave,
                   lane);  // (its barriers fence the xb reads of the dz phase)
  if constexpr (DROP) {
    store_tiles(dyres, p.dskip + row0 * SL_D + wave * 16, SL_D);  // the skip carries the unmasked gradient
#pragma unroll
    for (int rt = 0; rt < RT; ++rt) {  // attention-output dropout
      const f32x4 f = drop_factors(p.drop, dkey, dstep, (uint32_t)p.drop_site, row0 + rt * 16, SL_D, col, lane);
#pragma unroll
      for (int r = 0; r < 4; ++r) dyres[rt][r] *= f[r];
    }
  }
#pragma unroll
  for (int rt = 0; rt < RT; ++rt)
#pragma unroll
    for (int r = 0; r < 4; ++r) xb[(rt * 16 + fq * 4 + r) * SL_XP + col] = (__bf16)dyres[rt][r];
  store_tiles(dyres, p.dpre1 + row0 * SL_D + wave * 16, SL_D);  // fp32 (without dropout: also the next launch's skip gradient)
  __syncthreads();  // d pre-norm-1 image complete

  // ================= out-projection^T: gradient of the attention output =================
  {
    bf16x8 wf[4];
#pragma unroll
    for (int kk = 0; kk < 4; ++kk) wf[kk] = ld_wfrag(w_ot, wave * 4 + kk, lane);
    f32x4 dc[RT];
#pragma unroll
    for (int rt = 0; rt < RT; ++rt) {
      dc[rt] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int kk = 0; kk < 4; ++kk)
        dc[rt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ld_frag(xb + (rt * 16 + fr) * SL_XP + kk * 32 + fq * 8), wf[kk], dc[rt], 0, 0, 0);
    }
    store_tiles(dc, p.dctx + row0 * SL_D + wave * 16, SL_D);
  }
}

template <int RT>
size_t tile_bwd_lds_bytes(int F) {
  const int LP = 16 * RT;
  const int a = LP * SL_XP + LP * (F + 8), q = LP * (3 * SL_D + 8);
  const int region = (((a > q ? a : q) * 2) + 15) & ~15;
  return (size_t)region + (size_t)SL_NW * RT * 320 * 4 + (size_t)LP * SL_NW * 8 + (size_t)LP * 16;
}

template <int RT>
size_t tile_lds_bytes(int F) {
  const int LP = 16 * RT;
  return (size_t)LP * SL_XP * 2 + (size_t)LP * (F + 8) * 2 + (size_t)SL_NW * RT * 320 * 4 + (size_t)LP * SL_NW * 8 + (size_t)LP * 8;
}

inline bool al16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; }

// 16-row tiles while they are what fills the chip (the launch is latency-bound: a workgroup's time is its weight stream
// and its dependent phases, not its MFMA count)
inline int tile16_max_rows() {
  static const int v = [] { const char* e = getenv("RF_ENC_TILE16_ROWS"); return e ? atoi(e) : 2048; }();
  return v;
}

}  // namespace

extern "C" int rf_enclayer_tile_supported(int d_model, int n_heads, int d_ff) {
  return d_model == SL_D && n_heads == SL_H && d_ff >= 128 && d_ff <= 256 && d_ff % 32 == 0;
}

// One row-tile launch (see the head of this file).  `wpack` / `wpack_next`: per-layer blobs of rf_seqlayer_pack
// (rf_seqlayer_pack_bytes(d_ff) each).  ctx == NULL: projection only (q | k | v of the first layer from x);
// wpack_next == NULL: no projection (last layer).  save: 1 = write the training saves (all seven pointers required; z only
// for GELU).  Rows per workgroup: 32 (M <= 4096) or 48.
extern "C" int rf_enclayer_tile_fwd(const float* ctx, const float* x, const void* wpack, const void* wpack_next, float* y,
                                    float* qkv_next, float* xhat1, float* rstd1, float* x1, float* z, float* h, float* xhat2,
                                    float* rstd2, int M, int d_model, int n_heads, int d_ff, int act, int save, float eps,
                                    float drop_p, const void* rng_state, int drop_site, void* stream) {
  RF_REQUIRE(x && M > 0 && (ctx || wpack_next) && rf_enclayer_tile_supported(d_model, n_heads, d_ff));
  RF_REQUIRE(!ctx || (wpack && y && al16(ctx) && al16(y) && al16(wpack)));
  RF_REQUIRE(!wpack_next || (qkv_next && al16(qkv_next) && al16(wpack_next)));
  RF_REQUIRE(!save || !ctx || (xhat1 && rstd1 && x1 && h && xhat2 && rstd2 && (z || act != RF_ACT_GELU)));
  RF_REQUIRE(al16(x));
  EncTileP p{};
  p.ctx = ctx; p.x = x; p.wl = static_cast<const unsigned char*>(wpack); p.wnext = static_cast<const unsigned char*>(wpack_next);
  p.y = y; p.qkv_next = qkv_next; p.xhat1 = xhat1; p.rstd1 = rstd1; p.x1 = x1; p.z = z; p.h = h; p.xhat2 = xhat2; p.rstd2 = rstd2;
  p.M = M; p.F = d_ff; p.act = act; p.eps = eps;
  RF_REQUIRE(drop_p >= 0.f && drop_p < 1.f && (drop_p == 0.f || rng_state));
  p.drop = make_drop_cfg(ctx ? rng_state : nullptr, nullptr, 0, drop_p);  // (the projection-only launch has no dropout site)
  p.drop_site = drop_site;
  const hipStream_t st = static_cast<hipStream_t>(stream);
  const bool sv = save && ctx, drop = p.drop.state != nullptr;
#define RF_ET_GO(RT_, SAVE_, DROP_)                                                                                   \
  do {                                                                                                                \
    const size_t lds = tile_lds_bytes<RT_>(d_ff);                                                                     \
    static bool attr = false;                                                                                         \
    if (!attr) {                                                                                                      \
      (void)hipFuncSetAttribute(reinterpret_cast<const void*>(enc_tile_fwd_kernel<RT_, SAVE_, DROP_>),                \
                                hipFuncAttributeMaxDynamicSharedMemorySize, 96 * 1024);                               \
      attr = true;                                                                                                    \
    }                                                                                                                 \
    RF_LAUNCH((enc_tile_fwd_kernel<RT_, SAVE_, DROP_>), dim3((M + 16 * RT_ - 1) / (16 * RT_)), dim3(SL_NT), lds, st, p); \
  } while (0)
#define RF_ET_PICK(RT_)                                                                 \
  do {                                                                                  \
    if (drop && sv) RF_ET_GO(RT_, true, true);                                          \
    else if (drop) RF_ET_GO(RT_, false, true);  /* train-mode forward under no_grad (the target-side pass) */ \
    else if (sv) RF_ET_GO(RT_, true, false);                                            \
    else RF_ET_GO(RT_, false, false);                                                   \
  } while (0)
  if (M <= tile16_max_rows()) RF_ET_PICK(1);
  else if (M <= 4096) RF_ET_PICK(2);
  else RF_ET_PICK(3);
#undef RF_ET_PICK
#undef RF_ET_GO
  RF_CHECK_LAUNCH();
  return RF_OK;
}

// Backward row-tile launch (see above).  dqkv / skip / wpack_next_t: projection^T part (null for the stack's last layer, whose
// output gradient comes in `dy`); wpack_t == NULL: projection^T only -> dx (the stack's input gradient, after the first layer).
extern "C" int rf_enclayer_tile_bwd(const float* dy, const float* dqkv, const float* skip, const void* wpack_t,
                                    const void* wpack_next_t, const float* xhat1, const float* rstd1, const float* zsrc,
                                    const float* xhat2, const float* rstd2, float* dpre2, float* dz, float* dpre1, float* dctx,
                                    float* dx, float* dgamma1, float* dbeta1, float* dgamma2, float* dbeta2, int M, int d_model,
                                    int n_heads, int d_ff, int act, float* dskip, float drop_p, const void* rng_state,
                                    int drop_site, void* stream) {
  RF_REQUIRE(M > 0 && rf_enclayer_tile_supported(d_model, n_heads, d_ff));
  RF_REQUIRE((dqkv != nullptr) != (dy != nullptr));
  RF_REQUIRE(!dqkv || (skip && wpack_next_t && al16(dqkv) && al16(skip) && al16(wpack_next_t)));
  RF_REQUIRE(wpack_t || (dx && al16(dx)));
  RF_REQUIRE(!wpack_t || (xhat1 && rstd1 && zsrc && xhat2 && rstd2 && dpre2 && dz && dpre1 && dctx && dgamma1 && dbeta1 && dgamma2 &&
                          dbeta2 && al16(wpack_t) && al16(dpre2) && al16(dz) && al16(dpre1) && al16(dctx)));
  EncTileBwdP p{};
  p.dy = dy; p.dqkv = dqkv; p.skip = skip; p.wl = static_cast<const unsigned char*>(wpack_t);
  p.wnext = static_cast<const unsigned char*>(wpack_next_t);
  p.xhat1 = xhat1; p.rstd1 = rstd1; p.zsrc = zsrc; p.xhat2 = xhat2; p.rstd2 = rstd2;
  p.dpre2 = dpre2; p.dz = dz; p.dpre1 = dpre1; p.dctx = dctx; p.dx = dx;
  p.dg1 = dgamma1; p.db1 = dbeta1; p.dg2 = dgamma2; p.db2 = dbeta2;
  p.M = M; p.F = d_ff; p.act = act;
  RF_REQUIRE(drop_p >= 0.f && drop_p < 1.f && (drop_p == 0.f || rng_state));
  p.drop = make_drop_cfg(wpack_t ? rng_state : nullptr, nullptr, 0, drop_p);
  p.drop_site = drop_site; p.dskip = dskip;
  const bool drop = p.drop.state != nullptr;
  RF_REQUIRE(!drop || (dskip && al16(dskip)));
  const hipStream_t st = static_cast<hipStream_t>(stream);
#define RF_ETB_GO(RT_, DROP_)                                                                                        \
  do {                                                                                                               \
    static bool attr = false;                                                                                        \
    if (!attr) {                                                                                                     \
      (void)hipFuncSetAttribute(reinterpret_cast<const void*>(enc_tile_bwd_kernel<RT_, DROP_>),                      \
                                hipFuncAttributeMaxDynamicSharedMemorySize, 96 * 1024);                              \
      attr = true;                                                                                                   \
    }                                                                                                                \
    RF_LAUNCH((enc_tile_bwd_kernel<RT_, DROP_>), dim3((M + 16 * RT_ - 1) / (16 * RT_)), dim3(SL_NT), tile_bwd_lds_bytes<RT_>(d_ff), st, p); \
  } while (0)
#define RF_ETB_PICK(RT_) do { if (drop) RF_ETB_GO(RT_, true); else RF_ETB_GO(RT_, false); } while (0)
  if (M <= tile16_max_rows()) RF_ETB_PICK(1);
  else if (M <= 4096) RF_ETB_PICK(2);
  else RF_ETB_PICK(3);
#undef RF_ETB_PICK
#undef RF_ETB_GO
  RF_CHECK_LAUNCH();
  return RF_OK;
}
