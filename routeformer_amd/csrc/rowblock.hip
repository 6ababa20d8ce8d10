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

__device__ const float4 rb_zero16 = {0.f, 0.f, 0.f, 0.f};  // what rows beyond the matrix read (branch-free loads)

// ROWS x KC fp32 (k contiguous, row pitch ld) -> registers; rows >= nrows read as zero
template <int ROWS, int KC, int NTH = NT>
struct Tile {
  static constexpr int VPR = KC / 4, NV = ROWS * VPR / NTH;
  float4 r[NV];
  __device__ __forceinline__ void load(const float* __restrict__ g, long ld, int row0, int nrows, int tid) {
#pragma unroll
    for (int s = 0; s < NV; ++s) {
      const int i = tid + s * NTH, rr = i / VPR, kv = (i % VPR) * 4;
      // unconditional 16-B load (a predicated one costs an exec-mask branch per load): out-of-range rows
      // read a block of zeros
      const float* src = (row0 + rr < nrows) ? g + (long)(row0 + rr) * ld + kv : reinterpret_cast<const float*>(&rb_zero16);
      r[s] = *reinterpret_cast<const float4*>(src);
    }
  }
  __device__ __forceinline__ void store(__bf16* __restrict__ lds, int tid) const {  // [row][k], pitch KC + 8
#pragma unroll
    for (int s = 0; s < NV; ++s) {
      const int i = tid + s * NTH;
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

// d/dx of erf-GELU = Phi(x) + x phi(x), same erf approximation (abs error <= ~2e-7), exp through v_exp_f32
__device__ __forceinline__ float gelu_grad_fast(float x) {
  const float u = fabsf(x) * 0.70710678118654752440f;
  const float t = __builtin_amdgcn_rcpf(fmaf(0.3275911f, u, 1.0f));
  float poly = fmaf(1.061405429f, t, -1.453152027f);
  poly = fmaf(poly, t, 1.421413741f);
  poly = fmaf(poly, t, -0.284496736f);
  poly = fmaf(poly, t, 0.254829592f);
  const float e = __expf(-u * u);  // = exp(-x^2 / 2)
  const float cdf = 0.5f * (1.0f + copysignf(1.0f - poly * t * e, x));
  return fmaf(x * 0.39894228040143267794f, e, cdf);
}
__device__ __forceinline__ float act_grad_fast(float src, int mode) {
  return mode == RF_ACT_GELU ? gelu_grad_fast(src) : act_grad(src, mode);
}

// LayerNorm over the 128 columns a wave holds for each of its rows (two-pass, biased variance, as nn.LayerNorm)
__device__ __forceinline__ void layer_norm_rows(f32x4 (&v)[8], f32x4 (&y)[8], const float (&gam)[8],
                                                const float (&bet)[8], float eps, float (&rstd)[4]) {
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
  for (int j = 0; j < 8; ++j)
#pragma unroll
    for (int r = 0; r < 4; ++r) y[j][r] = v[j][r] * gam[j] + bet[j];
}

// 8-wave variants: a wave holds 16 rows x (16 NJ) columns; wave = RG * column-group + row-group, where a
// workgroup owns RBT = 64 rows (RG = 4 row groups, 2 column groups) or, for the launches with few row blocks
// (M <= 2048: per-workgroup latency is what counts there), RBT = 32 rows (RG = 2, 4 column groups).
// 16 x 16 NJ tile (MFMA C layout) -> wave-private LDS staging -> row-contiguous float4 stores
template <int NJ>
__device__ __forceinline__ void store_rows_w(const f32x4 (&v)[NJ], float* __restrict__ stage_w, float* __restrict__ g,
                                             long ld, int row0, int nrows, int col0, int ncols, int lane) {
  constexpr int PW = 16 * NJ + 4, V4 = 4 * NJ;  // staging pitch, float4 per row
  const int fr = lane & 15, fq = lane >> 4;
#pragma unroll
  for (int j = 0; j < NJ; ++j)
#pragma unroll
    for (int r = 0; r < 4; ++r) stage_w[(fq * 4 + r) * PW + j * 16 + fr] = v[j][r];
  __builtin_amdgcn_wave_barrier();
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
  for (int it = 0; it < NJ; ++it) {  // 16 * V4 float4 / 64 lanes = NJ trips
    const int i = it * 64 + lane, rr = i / V4, c4 = (i % V4) * 4;
    if (row0 + rr < nrows && col0 + c4 < ncols) {
      const float4 o = *reinterpret_cast<const float4*>(stage_w + rr * PW + c4);
      float* p = g + (long)(row0 + rr) * ld + col0 + c4;
      if (col0 + c4 + 3 < ncols) *reinterpret_cast<float4*>(p) = o;
      else { p[0] = o.x; if (col0 + c4 + 1 < ncols) p[1] = o.y; if (col0 + c4 + 2 < ncols) p[2] = o.z; }
    }
  }
  __builtin_amdgcn_wave_barrier();
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
}

// LayerNorm over 128 columns held by CH waves (16 NJ columns each): per-row partial sums meet in LDS
// (lnred[CH][RBT]), block barriers between the passes (mean, then centred variance: the same two-pass
// arithmetic as the one-wave form).  row = this wave's first row inside the block.
template <int NJ, int CH, int RBT>
__device__ __forceinline__ void layer_norm_rows_split(f32x4 (&v)[NJ], f32x4 (&y)[NJ], const float (&gam)[NJ],
                                                      const float (&bet)[NJ], float eps, float (&rstd)[4],
                                                      float* __restrict__ lnred, int row, int cg, int lane) {
  const int fr = lane & 15, fq = lane >> 4;
  float mean[4];
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    float s = 0.f;
#pragma unroll
    for (int j = 0; j < NJ; ++j) s += v[j][r];
    s = row16_sum(s);
    if (fr == 0) lnred[cg * RBT + row + fq * 4 + r] = s;
  }
  __syncthreads();
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    float t = 0.f;
#pragma unroll
    for (int k = 0; k < CH; ++k) t += lnred[k * RBT + row + fq * 4 + r];
    mean[r] = t * (1.f / 128.f);
  }
  __syncthreads();
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    float q = 0.f;
#pragma unroll
    for (int j = 0; j < NJ; ++j) { const float d = v[j][r] - mean[r]; q += d * d; }
    q = row16_sum(q);
    if (fr == 0) lnred[cg * RBT + row + fq * 4 + r] = q;
  }
  __syncthreads();
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    float t = 0.f;
#pragma unroll
    for (int k = 0; k < CH; ++k) t += lnred[k * RBT + row + fq * 4 + r];
    rstd[r] = 1.0f / sqrtf(t * (1.f / 128.f) + eps);
#pragma unroll
    for (int j = 0; j < NJ; ++j) v[j][r] = (v[j][r] - mean[r]) * rstd[r];  // v becomes x-hat
  }
#pragma unroll
  for (int j = 0; j < NJ; ++j)
#pragma unroll
    for (int r = 0; r < 4; ++r) y[j][r] = v[j][r] * gam[j] + bet[j];
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

constexpr int NT8 = 512;  // 8 waves: two per SIMD (these kernels are instruction-issue bound at one)

template <int KC, bool LN, int RBT>
__global__ __launch_bounds__(NT8) void rb_linear_kernel(LinP p) {
  constexpr int RG = RBT / 16, CH = 8 / RG, NCOL = 128 / CH, NJ = NCOL / 16, PW = NCOL + 4;
  constexpr int LD = KC + 8;
  constexpr int W_EL = 128 * LD, STAGE_EL = 8 * 16 * PW * 2;  // staging (fp32) expressed in bf16 elements
  __shared__ __attribute__((aligned(16))) __bf16 smem[(W_EL > STAGE_EL ? W_EL : STAGE_EL) + RBT * LD];
  __shared__ float lnred[CH * RBT];
  __bf16* ws = smem;
  __bf16* xs = smem + (W_EL > STAGE_EL ? W_EL : STAGE_EL);
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, fr = lane & 15, fq = lane >> 4;
  const int rw = wave % RG, ch = wave / RG;
  const int m0 = blockIdx.y * RBT, n0 = blockIdx.x * 128;

  Tile<RBT, KC, NT8> tx;
  Tile<128, KC, NT8> tw;
  tx.load(p.x, p.ldx, m0, p.M, tid);
  tw.load(p.w, KC, n0, p.N, tid);
  tx.store(xs, tid);
  tw.store(ws, tid);
  __syncthreads();

  // epilogue operands are requested before the matrix work: their latency hides behind it
  float bv[NJ], gam[NJ], bet[NJ];
  f32x4 rv[NJ];
#pragma unroll
  for (int j = 0; j < NJ; ++j) {
    const int cl = ch * NCOL + j * 16 + fr, n = n0 + cl;
    // unconditional loads at clamped indices (a predicated load is an exec-mask branch each; rows / columns
    // outside the matrix are never stored, so what they read does not matter)
    const int nc = min(n, p.N - 1);
    bv[j] = p.bias ? p.bias[nc] : 0.f;
    if constexpr (LN) { gam[j] = p.gamma[cl]; bet[j] = p.beta[cl]; }
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int mc = min(m0 + rw * 16 + fq * 4 + r, p.M - 1);
      rv[j][r] = p.res ? p.res[(long)mc * p.ldr + nc] : 0.f;
    }
  }

  f32x4 acc[NJ];
#pragma unroll
  for (int j = 0; j < NJ; ++j) acc[j] = f32x4{0.f, 0.f, 0.f, 0.f};
  mma_rows<KC, NJ>(acc, xs + rw * 16 * LD, ws + ch * NCOL * LD, lane);

#pragma unroll
  for (int j = 0; j < NJ; ++j)
#pragma unroll
    for (int r = 0; r < 4; ++r) acc[j][r] += bv[j] + rv[j][r];
  __syncthreads();  // every wave is done with the weights: their LDS becomes the output staging tile
  float* stage_w = reinterpret_cast<float*>(smem) + wave * 16 * PW;
  if constexpr (LN) {
    f32x4 y[NJ];
    float rstd[4];
    layer_norm_rows_split<NJ, CH, RBT>(acc, y, gam, bet, p.eps, rstd, lnred, rw * 16, ch, lane);
    store_rows_w<NJ>(y, stage_w, p.y, p.ldy, m0 + rw * 16, p.M, ch * NCOL, 128, lane);
    if (p.xhat) {
      store_rows_w<NJ>(acc, stage_w, p.xhat, 128, m0 + rw * 16, p.M, ch * NCOL, 128, lane);
      if (fr == 0 && ch == 0) {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int m = m0 + rw * 16 + fq * 4 + r;
          if (m < p.M) p.rstd[m] = rstd[r];
        }
      }
    }
  } else {
    store_rows_w<NJ>(acc, stage_w, p.y, p.ldy, m0 + rw * 16, p.M, n0 + ch * NCOL, p.N, lane);
  }
}

// Phase timing aid (tools/rb_phase_probe.py builds a private copy with -DRF_RB_TIMING)
#ifdef RF_RB_TIMING
__device__ unsigned long long rf_rb_timing[16 * 1024];
#define RB_MARK(k) do { if (threadIdx.x == 0 && blockIdx.x < 1024) rf_rb_timing[blockIdx.x * 16 + (k)] = __builtin_readcyclecounter(); } while (0)
extern "C" void* rf_rb_timing_address() {
  void* a = nullptr;
  (void)hipGetSymbolAddress(&a, HIP_SYMBOL(rf_rb_timing));
  return a;
}
#else
#define RB_MARK(k) do {} while (0)
#endif

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

template <int RBT>
__global__ __launch_bounds__(NT8) void rb_ffn_ln_kernel(FfnP p) {
  // 8 waves: wave = RG * column-group + row-group (16 rows); RBT = 64 rows (2 column groups) or 32 (4).
  constexpr int RG = RBT / 16, CH = 8 / RG;
  constexpr int D = 128, F = 256, LDX = D + 8, LDH = F + 8;
  constexpr int ZC = 64 / CH, ZJ = ZC / 16, ZP = ZC + 4;   // phase 1: columns / tiles per wave and pass, z-tile pitch
  constexpr int NC2 = D / CH, NJ2 = NC2 / 16, PW = NC2 + 4;  // phase 2: output columns / tiles per wave
  constexpr int WB = (F * LDX > D * LDH) ? F * LDX : D * LDH;  // W1 (256 x 136) then W2 (128 x 264), bf16
  __shared__ __attribute__((aligned(16))) __bf16 smem[WB + RBT * LDX + RBT * LDH + 8 * 16 * ZP * 2];
  __shared__ float lnred[CH * RBT];
  __bf16* wb = smem;
  __bf16* xs = smem + WB;
  __bf16* hs = xs + RBT * LDX;
  float* zs = reinterpret_cast<float*>(hs + RBT * LDH);
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, fr = lane & 15, fq = lane >> 4;
  const int rw = wave % RG, ch = wave / RG;
  const int m0 = blockIdx.x * RBT;
  RB_MARK(0);

  {
    Tile<RBT, D, NT8> tx;
    Tile<F, D, NT8> t1;
    tx.load(p.x, D, m0, p.M, tid);
    t1.load(p.w1, D, 0, F, tid);
    tx.store(xs, tid);
    t1.store(wb, tid);
  }
  __syncthreads();
  RB_MARK(1);
  Tile<D, F, NT8> t2;  // W2 travels while the first contraction runs
  t2.load(p.w2, F, 0, D, tid);

  // First contraction in four passes of 64 hidden columns (ZC per wave).  Each wave's 16 x ZC block goes
  // through a wave-private fp32 LDS tile, then four consecutive columns per lane: one float4 tile read, four
  // independent GELU chains, one 8-byte bf16 LDS write and two float4 global stores per four elements.  (The
  // loop is instruction-issue bound -- tools/rb_phase_probe.py -- hence float4 work and two waves per SIMD.)
  float* zt = zs + wave * 16 * ZP;
  constexpr int V4 = ZC / 4, RPT = 64 / V4, TRIPS = 16 / RPT;  // float4 per row, rows per trip, trips
  const int c4 = (lane % V4) * 4, rsub = lane / V4;
  float4 b1v[4];
#pragma unroll
  for (int c = 0; c < 4; ++c) b1v[c] = *reinterpret_cast<const float4*>(p.b1 + c * 64 + ch * ZC + c4);
#pragma unroll 1
  for (int c = 0; c < 4; ++c) {
    f32x4 a1[ZJ];
#pragma unroll
    for (int j = 0; j < ZJ; ++j) a1[j] = f32x4{0.f, 0.f, 0.f, 0.f};
    if (c == 0) RB_MARK(6);
    mma_rows<D, ZJ>(a1, xs + rw * 16 * LDX, wb + (c * 64 + ch * ZC) * LDX, lane);
#pragma unroll
    for (int j = 0; j < ZJ; ++j)
#pragma unroll
      for (int r = 0; r < 4; ++r) zt[(fq * 4 + r) * ZP + j * 16 + fr] = a1[j][r];
    __builtin_amdgcn_wave_barrier();
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    if (c == 0) RB_MARK(7);
    const float4 b = c == 0 ? b1v[0] : (c == 1 ? b1v[1] : (c == 2 ? b1v[2] : b1v[3]));
    const int f = c * 64 + ch * ZC + c4;
#pragma unroll
    for (int it = 0; it < TRIPS; ++it) {
      const int rl = rw * 16 + it * RPT + rsub, m = m0 + rl;
      float4 zz = *reinterpret_cast<const float4*>(zt + (it * RPT + rsub) * ZP + c4);
      zz.x += b.x; zz.y += b.y; zz.z += b.z; zz.w += b.w;
      float4 hh;
      if (p.act == RF_ACT_GELU) {
        hh.x = gelu_fast(zz.x); hh.y = gelu_fast(zz.y); hh.z = gelu_fast(zz.z); hh.w = gelu_fast(zz.w);
      } else if (p.act == RF_ACT_RELU) {
        hh.x = fmaxf(zz.x, 0.f); hh.y = fmaxf(zz.y, 0.f); hh.z = fmaxf(zz.z, 0.f); hh.w = fmaxf(zz.w, 0.f);
      } else {
        hh.x = apply_act(zz.x, p.act); hh.y = apply_act(zz.y, p.act); hh.z = apply_act(zz.z, p.act);
        hh.w = apply_act(zz.w, p.act);
      }
      st_bf16x4(hs + rl * LDH + f, hh);
      if (m < p.M) {
        if (p.z) *reinterpret_cast<float4*>(p.z + (long)m * F + f) = zz;
        if (p.h) *reinterpret_cast<float4*>(p.h + (long)m * F + f) = hh;
      }
    }
    __builtin_amdgcn_wave_barrier();
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    if (c == 0) RB_MARK(8);
  }
  __syncthreads();  // W1 dead everywhere, hs complete
  RB_MARK(2);
  t2.store(wb, tid);
  // epilogue operands (bias, exact fp32 residual, norm parameters) are requested now, used after the MFMAs
  float b2v[NJ2], gam[NJ2], bet[NJ2];
  f32x4 xres[NJ2];
#pragma unroll
  for (int j = 0; j < NJ2; ++j) {
    const int n = ch * NC2 + j * 16 + fr;
    b2v[j] = p.b2[n]; gam[j] = p.gamma[n]; bet[j] = p.beta[n];
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int mc = min(m0 + rw * 16 + fq * 4 + r, p.M - 1);  // clamped, unconditional (never stored beyond M)
      xres[j][r] = p.x[(long)mc * D + n];
    }
  }
  __syncthreads();
  RB_MARK(3);

  f32x4 acc[NJ2];
#pragma unroll
  for (int j = 0; j < NJ2; ++j) acc[j] = f32x4{0.f, 0.f, 0.f, 0.f};
  mma_rows<F, NJ2>(acc, hs + rw * 16 * LDH, wb + ch * NC2 * LDH, lane);
#pragma unroll
  for (int j = 0; j < NJ2; ++j)
#pragma unroll
    for (int r = 0; r < 4; ++r) acc[j][r] += b2v[j] + xres[j][r];
  __syncthreads();
  RB_MARK(4);
  float* stage_w = reinterpret_cast<float*>(smem) + wave * 16 * PW;
  f32x4 y[NJ2];
  float rstd[4];
  layer_norm_rows_split<NJ2, CH, RBT>(acc, y, gam, bet, p.eps, rstd, lnred, rw * 16, ch, lane);
  store_rows_w<NJ2>(y, stage_w, p.y, D, m0 + rw * 16, p.M, ch * NC2, D, lane);
  if (p.xhat) {
    store_rows_w<NJ2>(acc, stage_w, p.xhat, D, m0 + rw * 16, p.M, ch * NC2, D, lane);
    if (fr == 0 && ch == 0) {
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int m = m0 + rw * 16 + fq * 4 + r;
        if (m < p.M) p.rstd[m] = rstd[r];
      }
    }
  }
  RB_MARK(5);
}

// ---------------------------------------------------------------------------------------------------
// Backward counterpart: y[M, NOUT] = A[M, KC] W[KC, NOUT] (+ epilogue), W row-major as the forward weight
// (out_features = KC rows, in_features = NOUT columns) -- the dX product of nn.Linear / Conv1d(k=1).  The
// weight is staged TRANSPOSED into LDS ([n][k], bf16) so the B fragments are contiguous.
//   LNBWD: A is not read but produced: the LayerNorm backward of the layer's norm (dy, x-hat, 1/sigma, gamma
//          -> d pre-norm input), written out in fp32 (it is also the skip-branch gradient and the dW operand)
//          and fed to the MFMAs as bf16; d gamma / d beta go to the gradient slots by fp32 atomics.
//   epilogue: multiply by act'(dsrc) (FFN backward), add a residual (skip-branch gradient), coalesced stores.
// One launch replaces LayerNorm-backward + dX GEMM (and its split-K reduce) of a d_model = 128 layer.
// ---------------------------------------------------------------------------------------------------
struct NnP {
  const float* a; long lda;
  const float* dy; const float* xhat; const float* rstd; const float* gamma;
  float* dpre; float* dgamma; float* dbeta;
  const float* w;
  const float* res; long ldr;
  const float* dsrc; long ldd; int dact;
  float* y; long ldy;
  int M;
};

__device__ __forceinline__ float quad_sumf(float v) {
  v += dpp_move<0xB1>(v);
  v += dpp_move<0x4E>(v);
  return v;
}

// WAVES = 8: two waves per SIMD share a row block (wave = 4 * column-half + row-group).  These kernels are
// instruction-issue bound at one wave per SIMD (tools/rb_phase_probe.py: the GELU' epilogue alone was 15k of
// 44k cycles), so the second wave per SIMD is close to a 2x on every phase; LDS use is unchanged.
template <int KC, int NOUT, bool LNBWD, int RBT>
__global__ __launch_bounds__(NT8) void rb_nn_kernel(NnP p) {
  static_assert(!LNBWD || KC == 128, "the LayerNorm-backward prologue works on 128-wide rows");
  // 8 waves, wave = RG * column-group + row-group; RBT = 64 rows per workgroup (2 column groups) or 32 (4)
  constexpr int WAVES = 8, NTH = 64 * WAVES, RG = RBT / 16, CH = WAVES / RG, NCOL = NOUT / CH;
  constexpr int LD = KC + 8, NJ = NCOL / 16, SPW = NCOL + 4;
  constexpr int W_EL = NOUT * LD, STAGE_EL = WAVES * 16 * SPW * 2;
  __shared__ __attribute__((aligned(16))) __bf16 smem[(W_EL > STAGE_EL ? W_EL : STAGE_EL) + RBT * LD];
  __shared__ float red[2 * (NTH / 128)][128];
  __bf16* ws = smem;
  __bf16* as = smem + (W_EL > STAGE_EL ? W_EL : STAGE_EL);
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, fr = lane & 15, fq = lane >> 4;
  const int rw = wave % RG, ch = wave / RG;
  const int m0 = blockIdx.x * RBT;
  RB_MARK(0);

  // Transposed weight staging, a wave covers a 16 (k) x 32 (n) patch per trip: lane = (k pair) + 8 * (n / 4)
  // reads two 128-B row segments (rows k, k + 1) and writes four packed bf16x2 words [n + e][k, k + 1].
  // Two register batches of four trips alternate, so loads are in flight while the other batch is written.
  constexpr int TRIPS = KC * NOUT / (512 * WAVES), KT = KC / 16;
  static_assert(TRIPS % 4 == 0, "weight staging works in batches of four trips");
  const int kp2 = (lane & 7) * 2, g4 = (lane >> 3) * 4;
  float4 ra[8], rb[8];
  auto wload = [&](float4 (&r)[8], int s0) {
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int T = wave + WAVES * (s0 + u), k = (T % KT) * 16 + kp2, n = (T / KT) * 32 + g4;
      r[2 * u] = *reinterpret_cast<const float4*>(p.w + (long)k * NOUT + n);
      r[2 * u + 1] = *reinterpret_cast<const float4*>(p.w + (long)(k + 1) * NOUT + n);
    }
  };
  auto wstore = [&](const float4 (&r)[8], int s0) {
    typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int T = wave + WAVES * (s0 + u), k = (T % KT) * 16 + kp2, n = (T / KT) * 32 + g4;
      const float4 lo = r[2 * u], hi = r[2 * u + 1];
      *reinterpret_cast<bf16x2*>(ws + (n + 0) * LD + k) = bf16x2{(__bf16)lo.x, (__bf16)hi.x};
      *reinterpret_cast<bf16x2*>(ws + (n + 1) * LD + k) = bf16x2{(__bf16)lo.y, (__bf16)hi.y};
      *reinterpret_cast<bf16x2*>(ws + (n + 2) * LD + k) = bf16x2{(__bf16)lo.z, (__bf16)hi.z};
      *reinterpret_cast<bf16x2*>(ws + (n + 3) * LD + k) = bf16x2{(__bf16)lo.w, (__bf16)hi.w};
    }
  };
  wload(ra, 0);

  // ---- A operand ----
  if constexpr (LNBWD) {
    constexpr int LPR = NTH / RBT, CPL = 128 / LPR, V4 = CPL / 4;  // lanes per row, columns / float4 per lane
    const int r = tid / LPR, c0 = (tid % LPR) * CPL, m = m0 + r;
    float g[CPL], xh[CPL];
    float s1 = 0.f, s2 = 0.f;
    const float rs = p.rstd[min(m, p.M - 1)];
#pragma unroll
    for (int v4 = 0; v4 < V4; ++v4) {
      // rows beyond M: unconditional loads of a zero block (no exec-mask branch per load)
      const float* dsrc_ = m < p.M ? p.dy + (long)m * 128 + c0 + v4 * 4 : reinterpret_cast<const float*>(&rb_zero16);
      const float* xsrc_ = m < p.M ? p.xhat + (long)m * 128 + c0 + v4 * 4 : reinterpret_cast<const float*>(&rb_zero16);
      const float4 d = *reinterpret_cast<const float4*>(dsrc_), x = *reinterpret_cast<const float4*>(xsrc_);
      const float4 gm = *reinterpret_cast<const float4*>(p.gamma + c0 + v4 * 4);
      g[v4 * 4 + 0] = d.x * gm.x; g[v4 * 4 + 1] = d.y * gm.y; g[v4 * 4 + 2] = d.z * gm.z; g[v4 * 4 + 3] = d.w * gm.w;
      xh[v4 * 4 + 0] = x.x; xh[v4 * 4 + 1] = x.y; xh[v4 * 4 + 2] = x.z; xh[v4 * 4 + 3] = x.w;
    }
#pragma unroll
    for (int i = 0; i < CPL; ++i) { s1 += g[i]; s2 += g[i] * xh[i]; }
    s1 = quad_sumf(s1); s2 = quad_sumf(s2);
    if constexpr (LPR >= 8) { s1 += dpp_move<0x141>(s1); s2 += dpp_move<0x141>(s2); }   // the other quad
    if constexpr (LPR == 16) { s1 += dpp_move<0x140>(s1); s2 += dpp_move<0x140>(s2); }  // the other half row
    const float m1 = s1 * (1.f / 128.f), m2 = s2 * (1.f / 128.f);
#pragma unroll
    for (int v4 = 0; v4 < V4; ++v4) {
      float4 o;
      o.x = rs * (g[v4 * 4 + 0] - m1 - xh[v4 * 4 + 0] * m2);
      o.y = rs * (g[v4 * 4 + 1] - m1 - xh[v4 * 4 + 1] * m2);
      o.z = rs * (g[v4 * 4 + 2] - m1 - xh[v4 * 4 + 2] * m2);
      o.w = rs * (g[v4 * 4 + 3] - m1 - xh[v4 * 4 + 3] * m2);
      if (m < p.M) *reinterpret_cast<float4*>(p.dpre + (long)m * 128 + c0 + v4 * 4) = o;
      st_bf16x4(as + r * LD + c0 + v4 * 4, o);
    }
  } else {
    Tile<RBT, KC, NTH> ta;
    ta.load(p.a, p.lda, m0, p.M, tid);
    ta.store(as, tid);
  }

  RB_MARK(1);
  // ---- W (KC x NOUT, row-major) -> LDS [n][k]  (first register batch requested before the A operand) ----
#pragma unroll 1
  for (int s0 = 0; s0 < TRIPS; s0 += 8) {
    if (s0 + 4 < TRIPS) wload(rb, s0 + 4);
    wstore(ra, s0);
    if (s0 + 8 < TRIPS) wload(ra, s0 + 8);
    if (s0 + 4 < TRIPS) wstore(rb, s0 + 4);
  }
  __syncthreads();
  RB_MARK(2);

  if constexpr (LNBWD) {
    // d gamma / d beta: column sums over this block's rows (dy, x-hat re-read from L2, column-major work split)
    constexpr int GR = NTH / 128, RPG = RBT / GR;  // row groups, rows per group
    const int c = tid & 127, grp = tid >> 7;
    float sb = 0.f, sg = 0.f;
#pragma unroll 1
    for (int r0 = grp * RPG; r0 < grp * RPG + RPG; r0 += 8) {  // 16 loads in flight per trip
      float d[8], x[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        const int m = m0 + r0 + u;
        const float keep = m < p.M ? 1.f : 0.f;
        const long mc = min(m, p.M - 1);
        d[u] = p.dy[mc * 128 + c] * keep;
        x[u] = p.xhat[mc * 128 + c];
      }
#pragma unroll
      for (int u = 0; u < 8; ++u) { sb += d[u]; sg += d[u] * x[u]; }
    }
    red[grp][c] = sb;
    red[GR + grp][c] = sg;
  }

  RB_MARK(3);
  f32x4 acc[NJ];
#pragma unroll
  for (int j = 0; j < NJ; ++j) acc[j] = f32x4{0.f, 0.f, 0.f, 0.f};
  mma_rows<KC, NJ>(acc, as + rw * 16 * LD, ws + ch * NCOL * LD, lane);
  __syncthreads();  // weights dead: their LDS becomes the output staging tile; `red` complete

  if constexpr (LNBWD) {
    if (tid < 128) {
      constexpr int GR = NTH / 128;
      float sb = 0.f, sg = 0.f;
#pragma unroll
      for (int k = 0; k < GR; ++k) { sb += red[k][tid]; sg += red[GR + k][tid]; }
      atomicAdd(p.dbeta + tid, sb);
      atomicAdd(p.dgamma + tid, sg);
    }
  }

  RB_MARK(4);
  // ---- epilogue through the staging tile: row-contiguous float4 work ----
  float* st = reinterpret_cast<float*>(smem) + wave * 16 * SPW;
#pragma unroll
  for (int j = 0; j < NJ; ++j)
#pragma unroll
    for (int r = 0; r < 4; ++r) st[(fq * 4 + r) * SPW + j * 16 + fr] = acc[j][r];
  __builtin_amdgcn_wave_barrier();
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  constexpr int V4R = NCOL / 4;        // float4 per row of this wave's column group
  constexpr int NIT = 16 * V4R / 64;   // float4 per lane
  static_assert(NIT % 2 == 0, "two float4 per trip");
  // two float4 per trip; the residual / activation-source loads of trip t + 1 are issued before trip t is computed
  auto fetch = [&](int it, float4 (&rz)[2], float4 (&dz)[2]) {
#pragma unroll
    for (int u = 0; u < 2; ++u) {
      const int i = (it + u) * 64 + lane, rr = i / V4R, col = ch * NCOL + (i % V4R) * 4, m = m0 + rw * 16 + rr;
      const long mc = min(m, p.M - 1);  // clamped rows: loaded unconditionally, never stored
      rz[u] = dz[u] = make_float4(0.f, 0.f, 0.f, 0.f);
      if (p.res) rz[u] = *reinterpret_cast<const float4*>(p.res + mc * p.ldr + col);
      if (p.dact) dz[u] = *reinterpret_cast<const float4*>(p.dsrc + mc * p.ldd + col);
    }
  };
  float4 rz[2], dz[2], nrz[2], ndz[2];
  fetch(0, rz, dz);
#pragma unroll 1
  for (int it = 0; it < NIT; it += 2) {
    if (it + 2 < NIT) fetch(it + 2, nrz, ndz);
    else { nrz[0] = nrz[1] = ndz[0] = ndz[1] = make_float4(0.f, 0.f, 0.f, 0.f); }
#pragma unroll
    for (int u = 0; u < 2; ++u) {
      const int i = (it + u) * 64 + lane, rr = i / V4R, cl = (i % V4R) * 4, m = m0 + rw * 16 + rr;
      float4 o = *reinterpret_cast<const float4*>(st + rr * SPW + cl);
      if (p.dact == RF_ACT_GELU) {  // four independent straight-line chains
        o.x *= gelu_grad_fast(dz[u].x); o.y *= gelu_grad_fast(dz[u].y);
        o.z *= gelu_grad_fast(dz[u].z); o.w *= gelu_grad_fast(dz[u].w);
      } else if (p.dact == RF_ACT_RELU) {
        o.x = dz[u].x > 0.f ? o.x : 0.f; o.y = dz[u].y > 0.f ? o.y : 0.f;
        o.z = dz[u].z > 0.f ? o.z : 0.f; o.w = dz[u].w > 0.f ? o.w : 0.f;
      } else if (p.dact) {
        o.x *= act_grad(dz[u].x, p.dact); o.y *= act_grad(dz[u].y, p.dact);
        o.z *= act_grad(dz[u].z, p.dact); o.w *= act_grad(dz[u].w, p.dact);
      }
      o.x += rz[u].x; o.y += rz[u].y; o.z += rz[u].z; o.w += rz[u].w;
      if (m < p.M) *reinterpret_cast<float4*>(p.y + (long)m * p.ldy + ch * NCOL + cl) = o;
    }
#pragma unroll
    for (int u = 0; u < 2; ++u) { rz[u] = nrz[u]; dz[u] = ndz[u]; }
  }
  RB_MARK(5);
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
  hipStream_t st = static_cast<hipStream_t>(stream);
  const bool small = M <= 2048;  // few row blocks: 32-row blocks (shorter per-workgroup chain, 2x the workgroups)
  const int rbt = small ? 32 : RB;
  dim3 grid((N + 127) / 128, (M + rbt - 1) / rbt);
#define RF_RB_LIN(KC_, LN_) \
  do { \
    if (small) RF_LAUNCH((rb_linear_kernel<KC_, LN_, 32>), grid, dim3(NT8), 0, st, p); \
    else RF_LAUNCH((rb_linear_kernel<KC_, LN_, RB>), grid, dim3(NT8), 0, st, p); \
  } while (0)
  if (K == 128) { if (ln) RF_RB_LIN(128, true); else RF_RB_LIN(128, false); }
  else { if (ln) RF_RB_LIN(256, true); else RF_RB_LIN(256, false); }
#undef RF_RB_LIN
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
  if (M <= 2048)  // few row blocks: 32-row blocks, 2x the workgroups
    RF_LAUNCH(rb_ffn_ln_kernel<32>, dim3((M + 31) / 32), dim3(NT8), 0, static_cast<hipStream_t>(stream), p);
  else
    RF_LAUNCH(rb_ffn_ln_kernel<RB>, dim3((M + RB - 1) / RB), dim3(NT8), 0, static_cast<hipStream_t>(stream), p);
  RF_CHECK_LAUNCH();
  return RF_OK;
}

extern "C" int rf_rowblock_linear_nn_supported(int KC, int NOUT, int ln_bwd) {
  if (ln_bwd) return KC == 128 && (NOUT == 128 || NOUT == 256);
  return (KC == 128 || KC == 256 || KC == 384) && NOUT == 128;
}

extern "C" int rf_rowblock_linear_nn(const float* a, int64_t lda, const float* ln_dy, const float* ln_xhat,
                                     const float* ln_rstd, const float* ln_gamma, float* dpre, float* dgamma,
                                     float* dbeta, const float* w, const float* residual, int64_t ldr,
                                     const float* dact_src, int64_t ldd, int dact_mode, float* y, int64_t ldy, int M,
                                     int KC, int NOUT, void* stream) {
  const bool ln = ln_dy != nullptr;
  RF_REQUIRE(w && y && M > 0 && rf_rowblock_linear_nn_supported(KC, NOUT, ln));
  RF_REQUIRE(ln ? (ln_xhat && ln_rstd && ln_gamma && dpre && dgamma && dbeta && al16(ln_dy) && al16(ln_xhat) &&
                   al16(ln_gamma) && al16(dpre))
                : (a && al16(a) && lda % 4 == 0));
  RF_REQUIRE(al16(w) && al16(y) && ldy % 4 == 0 && (!residual || (al16(residual) && ldr % 4 == 0)));
  RF_REQUIRE(!dact_mode || (dact_src && al16(dact_src) && ldd % 4 == 0));
  NnP p{a, lda, ln_dy, ln_xhat, ln_rstd, ln_gamma, dpre, dgamma, dbeta, w, residual, ldr,
        dact_mode ? dact_src : nullptr, ldd, dact_mode, y, ldy, M};
  hipStream_t st = static_cast<hipStream_t>(stream);
  const bool small = M <= 2048;  // few row blocks: 32-row blocks, 2x the workgroups
  dim3 grid(small ? (M + 31) / 32 : (M + RB - 1) / RB);
#define RF_RB_NN(KC_, NOUT_, LN_) \
  do { \
    if (small) RF_LAUNCH((rb_nn_kernel<KC_, NOUT_, LN_, 32>), grid, dim3(NT8), 0, st, p); \
    else RF_LAUNCH((rb_nn_kernel<KC_, NOUT_, LN_, RB>), grid, dim3(NT8), 0, st, p); \
  } while (0)
  if (ln && NOUT == 128) RF_RB_NN(128, 128, true);
  else if (ln) RF_RB_NN(128, 256, true);
  else if (KC == 128) RF_RB_NN(128, 128, false);
  else if (KC == 256) RF_RB_NN(256, 128, false);
  else RF_RB_NN(384, 128, false);
#undef RF_RB_NN
  RF_CHECK_LAUNCH();
  return RF_OK;
}
