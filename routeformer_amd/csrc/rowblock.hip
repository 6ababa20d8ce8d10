// Row-block kernels for the d_model = 128 transformer stacks (frame encoder, gaze encoder, cross-modal
// decoder, fusion encoder: cross_modal_transformer.py:279-365 EncoderLayer / DecoderLayer).
//
// There the activations are tall (12 480 x 128 at B = 8) and the weights tiny (<= 384 x 128), so a classic
// tiled GEMM spends its time in prologue / epilogue and re-reads the activation panel once per column tile.
// These kernels turn the loop nest around: a workgroup owns 64 complete rows, stages the WHOLE weight
// matrix (bf16) in LDS next to its row block, and -- because it holds complete rows -- finishes the
// residual add and the LayerNorm in the epilogue:
//   rb_linear_kernel : y = x W^T + b [+ res] [-> LayerNorm]        (projections; out-projection + norm1)
//   rb_ffn_ln_kernel : y = LayerNorm(x + W2 act(W1 x + b1) + b2)   (conv1 -> act -> conv2 -> norm2, one launch)
// bf16-input MFMA (v_mfma_f32_16x16x32_bf16), fp32 accumulate, fp32 in HBM -- this is the `prec = bf16`
// path only; the exact-fp32 mode keeps using rf_gemm + rf_layernorm_fwd.
#include "common.h"

namespace {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

constexpr int NT = 256;   // 4 waves; wave w owns rows 16w .. 16w+15 of the 64-row block
constexpr int RB = 64;    // rows per workgroup
constexpr int SP = 132;   // fp32 pitch of the output staging tile (conflict-free for the MFMA C layout)

__device__ __forceinline__ void st_bf16x4(__bf16* s, const float4& v) {
  bf16x4 o = {(__bf16)v.x, (__bf16)v.y, (__bf16)v.z, (__bf16)v.w};
  *reinterpret_cast<bf16x4*>(s) = o;
}

// ROWS x KC fp32 (k contiguous, row pitch ld) -> registers; rows >= nrows read as zero
template <int ROWS, int KC>
struct Tile {
  static constexpr int VPR = KC / 4, NV = ROWS * VPR / NT;
  float4 r[NV];
  __device__ __forceinline__ void load(const float* __restrict__ g, long ld, int row0, int nrows, int tid) {
#pragma unroll
    for (int s = 0; s < NV; ++s) {
      const int i = tid + s * NT, rr = i / VPR, kv = (i % VPR) * 4;
      r[s] = (row0 + rr < nrows) ? *reinterpret_cast<const float4*>(g + (long)(row0 + rr) * ld + kv)
                                 : make_float4(0.f, 0.f, 0.f, 0.f);
    }
  }
  __device__ __forceinline__ void store(__bf16* __restrict__ lds, int tid) const {  // [row][k], pitch KC + 8
#pragma unroll
    for (int s = 0; s < NV; ++s) {
      const int i = tid + s * NT;
      st_bf16x4(lds + (i / VPR) * (KC + 8) + (i % VPR) * 4, r[s]);
    }
  }
};

// acc[j] += A(16 rows of this wave, KC) * B(16 j .. 16 j + 15, KC)^T for j < NJ
template <int KC, int NJ>
__device__ __forceinline__ void mma_rows(f32x4 (&acc)[NJ], const __bf16* __restrict__ a_rows,
                                         const __bf16* __restrict__ b, int lane) {
  constexpr int LD = KC + 8;
  const int fr = lane & 15, fq = lane >> 4;
#pragma unroll
  for (int ks = 0; ks < KC / 32; ++ks) {
    const bf16x8 a = *reinterpret_cast<const bf16x8*>(a_rows + fr * LD + ks * 32 + fq * 8);
#pragma unroll
    for (int j = 0; j < NJ; ++j) {
      const bf16x8 bb = *reinterpret_cast<const bf16x8*>(b + (j * 16 + fr) * LD + ks * 32 + fq * 8);
      acc[j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, bb, acc[j], 0, 0, 0);
    }
  }
}

// wave-private 16 x 128 tile (MFMA C layout) -> LDS staging -> row-contiguous float4 stores
__device__ __forceinline__ void store_rows(const f32x4 (&v)[8], float* __restrict__ stage_w, float* __restrict__ g,
                                           long ld, int row0, int nrows, int col0, int ncols, int lane) {
  const int fr = lane & 15, fq = lane >> 4;
#pragma unroll
  for (int j = 0; j < 8; ++j)
#pragma unroll
    for (int r = 0; r < 4; ++r) stage_w[(fq * 4 + r) * SP + j * 16 + fr] = v[j][r];
  __builtin_amdgcn_wave_barrier();
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
  for (int it = 0; it < 8; ++it) {
    const int i = it * 64 + lane, rr = i >> 5, c4 = (i & 31) * 4;
    if (row0 + rr < nrows && col0 + c4 < ncols) {
      const float4 o = *reinterpret_cast<const float4*>(stage_w + rr * SP + c4);
      float* p = g + (long)(row0 + rr) * ld + col0 + c4;
      if (col0 + c4 + 3 < ncols) *reinterpret_cast<float4*>(p) = o;
      else { p[0] = o.x; if (col0 + c4 + 1 < ncols) p[1] = o.y; if (col0 + c4 + 2 < ncols) p[2] = o.z; }
    }
  }
  __builtin_amdgcn_wave_barrier();
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
}

// erf-GELU with Abramowitz-Stegun 7.1.26 (|erf error| <= 1.5e-7, ~15 VALU instructions instead of libm's
// ~60): the activation is the VALU-bound part of the fused FFN kernel at one wave per SIMD.
__device__ __forceinline__ float gelu_fast(float x) {
  const float u = fabsf(x) * 0.70710678118654752440f;
  const float t = __builtin_amdgcn_rcpf(fmaf(0.3275911f, u, 1.0f));
  float poly = fmaf(1.061405429f, t, -1.453152027f);
  poly = fmaf(poly, t, 1.421413741f);
  poly = fmaf(poly, t, -0.284496736f);
  poly = fmaf(poly, t, 0.254829592f);
  const float erf_abs = 1.0f - poly * t * __expf(-u * u);
  return 0.5f * x * (1.0f + copysignf(erf_abs, x));
}

// LayerNorm over the 128 columns a wave holds for each of its rows (two-pass, biased variance, as nn.LayerNorm)
__device__ __forceinline__ void layer_norm_rows(f32x4 (&v)[8], f32x4 (&y)[8], const float* __restrict__ gamma,
                                                const float* __restrict__ beta, float eps, float (&rstd)[4], int lane) {
  const int fr = lane & 15;
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    float s = 0.f;
#pragma unroll
    for (int j = 0; j < 8; ++j) s += v[j][r];
    const float mean = row16_sum(s) * (1.f / 128.f);
    float q = 0.f;
#pragma unroll
    for (int j = 0; j < 8; ++j) { const float d = v[j][r] - mean; q += d * d; }
    rstd[r] = 1.0f / sqrtf(row16_sum(q) * (1.f / 128.f) + eps);
#pragma unroll
    for (int j = 0; j < 8; ++j) v[j][r] = (v[j][r] - mean) * rstd[r];  // v becomes x-hat
  }
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    const float g = gamma[j * 16 + fr], b = beta[j * 16 + fr];
#pragma unroll
    for (int r = 0; r < 4; ++r) y[j][r] = v[j][r] * g + b;
  }
}

struct LinP {
  const float* x; long ldx;
  const float* w;           // (N, KC) row-major, contiguous
  const float* bias;        // (N) or null
  const float* res; long ldr;  // (M, N) or null
  float* y; long ldy;
  int M, N;
  const float* gamma; const float* beta; float* xhat; float* rstd; float eps;  // LN = true only (N == 128)
};

template <int KC, bool LN>
__global__ __launch_bounds__(NT) void rb_linear_kernel(LinP p) {
  constexpr int LD = KC + 8;
  constexpr int W_EL = 128 * LD, STAGE_EL = RB * SP * 2;  // staging (fp32) expressed in bf16 elements
  __shared__ __attribute__((aligned(16))) __bf16 smem[(W_EL > STAGE_EL ? W_EL : STAGE_EL) + RB * LD];
  __bf16* ws = smem;
  __bf16* xs = smem + (W_EL > STAGE_EL ? W_EL : STAGE_EL);
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, fr = lane & 15, fq = lane >> 4;
  const int m0 = blockIdx.y * RB, n0 = blockIdx.x * 128;

  Tile<RB, KC> tx;
  Tile<128, KC> tw;
  tx.load(p.x, p.ldx, m0, p.M, tid);
  tw.load(p.w, KC, n0, p.N, tid);
  tx.store(xs, tid);
  tw.store(ws, tid);
  __syncthreads();

  f32x4 acc[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) acc[j] = f32x4{0.f, 0.f, 0.f, 0.f};
  mma_rows<KC, 8>(acc, xs + wave * 16 * LD, ws, lane);

#pragma unroll
  for (int j = 0; j < 8; ++j) {
    const int n = n0 + j * 16 + fr;
    const float b = (p.bias && n < p.N) ? p.bias[n] : 0.f;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int m = m0 + wave * 16 + fq * 4 + r;
      float v = acc[j][r] + b;
      if (p.res && m < p.M && n < p.N) v += p.res[(long)m * p.ldr + n];
      acc[j][r] = v;
    }
  }
  __syncthreads();  // every wave is done with the weights: their LDS becomes the output staging tile
  float* stage_w = reinterpret_cast<float*>(smem) + wave * 16 * SP;
  if constexpr (LN) {
    f32x4 y[8];
    float rstd[4];
    layer_norm_rows(acc, y, p.gamma, p.beta, p.eps, rstd, lane);
    store_rows(y, stage_w, p.y, p.ldy, m0 + wave * 16, p.M, 0, 128, lane);
    if (p.xhat) {
      store_rows(acc, stage_w, p.xhat, 128, m0 + wave * 16, p.M, 0, 128, lane);
      if (fr == 0) {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int m = m0 + wave * 16 + fq * 4 + r;
          if (m < p.M) p.rstd[m] = rstd[r];
        }
      }
    }
  } else {
    store_rows(acc, stage_w, p.y, p.ldy, m0 + wave * 16, p.M, n0, p.N, lane);
  }
}

struct FfnP {
  const float* x;   // (M, 128) contiguous: FFN input AND residual
  const float* w1; const float* b1;  // (256, 128), (256)
  const float* w2; const float* b2;  // (128, 256), (128)
  float* h;         // (M, 256) activation output, or null (no backward)
  float* z;         // (M, 256) pre-activation, or null
  float* y;         // (M, 128)
  int M, act;
  const float* gamma; const float* beta; float* xhat; float* rstd; float eps;
};

__global__ __launch_bounds__(NT) void rb_ffn_ln_kernel(FfnP p) {
  constexpr int D = 128, F = 256, LDX = D + 8, LDH = F + 8;
  constexpr int WB = (F * LDX > D * LDH) ? F * LDX : D * LDH;  // W1 (256 x 136) then W2 (128 x 264), bf16
  __shared__ __attribute__((aligned(16))) __bf16 smem[WB + RB * LDX + RB * LDH + RB * 68 * 2];
  __bf16* wb = smem;
  __bf16* xs = smem + WB;
  __bf16* hs = xs + RB * LDX;
  float* zs = reinterpret_cast<float*>(hs + RB * LDH);
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, fr = lane & 15, fq = lane >> 4;
  const int m0 = blockIdx.x * RB;

  {
    Tile<RB, D> tx;
    Tile<F, D> t1;
    tx.load(p.x, D, m0, p.M, tid);
    t1.load(p.w1, D, 0, F, tid);
    tx.store(xs, tid);
    t1.store(wb, tid);
  }
  __syncthreads();
  Tile<D, F> t2;  // W2 travels while the first contraction runs
  t2.load(p.w2, F, 0, D, tid);

  // first contraction in four passes of 64 hidden columns; each pass's 16 x 64 block goes through a
  // wave-private fp32 LDS tile so that bias + activation exist once in a rolled loop (code stays small: the
  // instruction cache is 64 KB) and z / h leave as row-contiguous 256-B stores
  float* zt = zs + wave * 16 * 68;
#pragma unroll 1
  for (int c = 0; c < 4; ++c) {
    f32x4 a1[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) a1[j] = f32x4{0.f, 0.f, 0.f, 0.f};
    mma_rows<D, 4>(a1, xs + wave * 16 * LDX, wb + c * 64 * LDX, lane);
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
      for (int r = 0; r < 4; ++r) zt[(fq * 4 + r) * 68 + j * 16 + fr] = a1[j][r];
    __builtin_amdgcn_wave_barrier();
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    const int f = c * 64 + lane;
    const float b = p.b1[f];
#pragma unroll 1
    for (int r0 = 0; r0 < 16; r0 += 4) {
      float zz[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) zz[u] = zt[(r0 + u) * 68 + lane] + b;
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const int rl = wave * 16 + r0 + u, m = m0 + rl;
        const float hh = p.act == RF_ACT_GELU ? gelu_fast(zz[u]) : apply_act(zz[u], p.act);
        if (m < p.M) {
          if (p.z) p.z[(long)m * F + f] = zz[u];
          if (p.h) p.h[(long)m * F + f] = hh;
        }
        hs[rl * LDH + f] = (__bf16)hh;
      }
    }
    __builtin_amdgcn_wave_barrier();
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  }
  __syncthreads();  // W1 dead everywhere, hs complete
  t2.store(wb, tid);
  __syncthreads();

  f32x4 acc[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) acc[j] = f32x4{0.f, 0.f, 0.f, 0.f};
  mma_rows<F, 8>(acc, hs + wave * 16 * LDH, wb, lane);
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    const int n = j * 16 + fr;
    const float b = p.b2[n];
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int m = m0 + wave * 16 + fq * 4 + r;
      acc[j][r] += b + (m < p.M ? p.x[(long)m * D + n] : 0.f);  // exact fp32 residual
    }
  }
  __syncthreads();
  float* stage_w = reinterpret_cast<float*>(smem) + wave * 16 * SP;
  f32x4 y[8];
  float rstd[4];
  layer_norm_rows(acc, y, p.gamma, p.beta, p.eps, rstd, lane);
  store_rows(y, stage_w, p.y, D, m0 + wave * 16, p.M, 0, D, lane);
  if (p.xhat) {
    store_rows(acc, stage_w, p.xhat, D, m0 + wave * 16, p.M, 0, D, lane);
    if (fr == 0) {
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int m = m0 + wave * 16 + fq * 4 + r;
        if (m < p.M) p.rstd[m] = rstd[r];
      }
    }
  }
}

inline bool al16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; }

}  // namespace

extern "C" int rf_rowblock_linear_supported(int N, int K, int with_ln) {
  return (K == 128 || K == 256) && N >= 1 && (!with_ln || N == 128);
}

extern "C" int rf_rowblock_linear(const float* x, int64_t ldx, const float* w, const float* bias, const float* residual,
                                  int64_t ldr, float* y, int64_t ldy, int M, int N, int K, const float* ln_gamma,
                                  const float* ln_beta, float* xhat, float* rstd, float eps, void* stream) {
  const bool ln = ln_gamma != nullptr;
  RF_REQUIRE(x && w && y && M > 0 && rf_rowblock_linear_supported(N, K, ln));
  RF_REQUIRE(al16(x) && al16(w) && al16(y) && ldx % 4 == 0 && ldy % 4 == 0);
  RF_REQUIRE(!ln || (ln_beta && (!xhat || (rstd && al16(xhat)))));
  LinP p{x, ldx, w, bias, residual, ldr, y, ldy, M, N, ln_gamma, ln_beta, xhat, rstd, eps};
  dim3 grid((N + 127) / 128, (M + RB - 1) / RB);
  hipStream_t st = static_cast<hipStream_t>(stream);
  if (K == 128) {
    if (ln) hipLaunchKernelGGL((rb_linear_kernel<128, true>), grid, dim3(NT), 0, st, p);
    else hipLaunchKernelGGL((rb_linear_kernel<128, false>), grid, dim3(NT), 0, st, p);
  } else {
    if (ln) hipLaunchKernelGGL((rb_linear_kernel<256, true>), grid, dim3(NT), 0, st, p);
    else hipLaunchKernelGGL((rb_linear_kernel<256, false>), grid, dim3(NT), 0, st, p);
  }
  RF_CHECK_LAUNCH();
  return RF_OK;
}

extern "C" int rf_rowblock_ffn_ln(const float* x, const float* w1, const float* b1, const float* w2, const float* b2,
                                  float* h, float* z, float* y, int M, int d_model, int d_ff, int act,
                                  const float* ln_gamma, const float* ln_beta, float* xhat, float* rstd, float eps,
                                  void* stream) {
  RF_REQUIRE(x && w1 && b1 && w2 && b2 && y && ln_gamma && ln_beta && M > 0 && d_model == 128 && d_ff == 256);
  RF_REQUIRE(al16(x) && al16(w1) && al16(w2) && al16(y) && (!xhat || (rstd && al16(xhat))));
  FfnP p{x, w1, b1, w2, b2, h, z, y, M, act, ln_gamma, ln_beta, xhat, rstd, eps};
  hipLaunchKernelGGL(rb_ffn_ln_kernel, dim3((M + RB - 1) / RB), dim3(NT), 0, static_cast<hipStream_t>(stream), p);
  RF_CHECK_LAUNCH();
  return RF_OK;
}
