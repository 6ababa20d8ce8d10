// Row-wise sequence ops of the transformer blocks: LayerNorm (+residual), circular unfold/fold for
// the k=3 token/distil convolutions, and the Informer distilling tail BatchNorm1d -> ELU -> MaxPool1d.
// All are HBM/latency-bound streaming kernels: channels innermost, one wave per row where a row
// reduction is needed (wave-shuffle reductions), coalesced 256-B accesses per wave-instruction.
#include <cstdlib>

#include "common.h"

namespace {

constexpr int LN_MAXV = 16;  // cols <= 64 * 16
constexpr int LN_WAVES = 4;

// NV = number of 64-column groups a lane walks (cols <= 64 NV), a template parameter so that the unrolled loads carry
// no branches: with a wave-uniform `if (group < cols)` around each of them (the previous form) the compiler waited for
// every load before the next branch -- 13 dependent memory round trips per row at cols = 832 (10-12 us for a 20-row
// launch; 4.5 us at cols = 128, where there are two).  Loads at a clamped column, masked by a select.
template <int NV>
__global__ __launch_bounds__(256) void layernorm_fwd_kernel(const float* __restrict__ x, const float* __restrict__ res,
                                                            const float* __restrict__ gamma,
                                                            const float* __restrict__ beta, float* __restrict__ y,
                                                            float* __restrict__ xhat, float* __restrict__ rstd_out,
                                                            int rows, int cols, float eps, int seg_rows, long seg_stride,
                                                            long row_stride) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int row = blockIdx.x * LN_WAVES + wave;
  if (row >= rows) return;
  const long off = (long)row * cols;
  // x may be a strided view (the consumed tail of a (B, L, C) tensor, a row-pitched 2-D view): row -> (segment, row in it)
  const int seg = row / seg_rows;
  const long xoff = (long)seg * seg_stride + (long)(row - seg * seg_rows) * row_stride;
  float v[NV], gm[NV], bt[NV];
#pragma unroll
  for (int i = 0; i < NV; ++i) v[i] = x[xoff + min(i * 64 + lane, cols - 1)];
  if (res) {
#pragma unroll
    for (int i = 0; i < NV; ++i) v[i] += res[off + min(i * 64 + lane, cols - 1)];
  }
#pragma unroll
  for (int i = 0; i < NV; ++i) {
    const int cc = min(i * 64 + lane, cols - 1);
    gm[i] = gamma[cc];
    bt[i] = beta[cc];
  }
  float s = 0.f;
#pragma unroll
  for (int i = 0; i < NV; ++i) {
    v[i] = (i * 64 + lane < cols) ? v[i] : 0.f;
    s += v[i];
  }
  const float mean = wave_sum(s) / (float)cols;
  float q = 0.f;
#pragma unroll
  for (int i = 0; i < NV; ++i) {
    const float d = (i * 64 + lane < cols) ? v[i] - mean : 0.f;
    q += d * d;
  }
  const float var = wave_sum(q) / (float)cols;
  const float rstd = 1.0f / sqrtf(var + eps);
#pragma unroll
  for (int i = 0; i < NV; ++i) {
    const int c = i * 64 + lane;
    if (c < cols) {
      const float h = (v[i] - mean) * rstd;
      if (xhat) xhat[off + c] = h;
      y[off + c] = h * gm[i] + bt[i];
    }
  }
  if (rstd_out && lane == 0) rstd_out[row] = rstd;
}

// Wide rows (cols > 256: the GPS backbone's d_model = 832): ONE ROW PER WORKGROUP, the four waves take the 64-column
// groups round-robin (wave w: groups w, w + 4, ...; NVW of them), the two row reductions meet in LDS.  A wave per row
// walks 13 groups per lane, and M = 32 .. 560 rows are 8 .. 140 workgroups: pure load latency.  The input may be the
// split-K slabs [splits][rows][cols] of the product that feeds the norm (rf_gemm_partials): they are summed here in slab
// order, then the bias -- the arithmetic of the slab-sum launch this replaces, one launch less per site.
template <int NVW, int SC, bool HAS_BIAS, bool HAS_RES>
__global__ __launch_bounds__(256) void layernorm_row_kernel(const float* __restrict__ x, int splits,
                                                            const float* __restrict__ bias, const float* __restrict__ res,
                                                            const float* __restrict__ gamma,
                                                            const float* __restrict__ beta, float* __restrict__ y,
                                                            float* __restrict__ xhat, float* __restrict__ rstd_out,
                                                            int rows, int cols, float eps, int seg_rows, long seg_stride,
                                                            long row_stride, int unf_L) {
  __shared__ float red[2][LN_WAVES];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int row = blockIdx.x;
  const long off = (long)row * cols, slab = (long)rows * cols;
  const int seg = row / seg_rows;  // strided x (splits == 1 only): see layernorm_fwd_kernel
  const long xoff = (long)seg * seg_stride + (long)(row - seg * seg_rows) * row_stride;
  int cc[NVW];
  float v[NVW], gm[NVW], bt[NVW], bs[NVW], rs[NVW];
  // every operand that does not depend on the slabs is requested first; nothing below branches around a load (a
  // wave-uniform branch makes the compiler drain the memory counter at the join: one round trip per branch)
#pragma unroll
  for (int i = 0; i < NVW; ++i) {
    cc[i] = min((wave + LN_WAVES * i) * 64 + lane, cols - 1);  // clamped, unconditional loads; masked below
    gm[i] = gamma[cc[i]];
    bt[i] = beta[cc[i]];
    if constexpr (HAS_BIAS) bs[i] = bias[cc[i]];
    if constexpr (HAS_RES) rs[i] = res[off + cc[i]];
    v[i] = 0.f;
  }
  // SC slabs in flight per trip (clamped slab index, masked add: v + 0 = v), summed in ascending slab order
  for (int s0 = 0; s0 < splits; s0 += SC) {
    float t[SC][NVW];
#pragma unroll
    for (int u = 0; u < SC; ++u) {
      const long so = (long)min(s0 + u, splits - 1) * slab + xoff;
#pragma unroll
      for (int i = 0; i < NVW; ++i) t[u][i] = x[so + cc[i]];
    }
#pragma unroll
    for (int u = 0; u < SC; ++u) {
      const bool in = s0 + u < splits;
#pragma unroll
      for (int i = 0; i < NVW; ++i) v[i] += in ? t[u][i] : 0.f;
    }
  }
  float s1 = 0.f;
#pragma unroll
  for (int i = 0; i < NVW; ++i) {
    if constexpr (HAS_BIAS) v[i] += bs[i];
    if constexpr (HAS_RES) v[i] += rs[i];
    v[i] = ((wave + LN_WAVES * i) * 64 + lane < cols) ? v[i] : 0.f;
    s1 += v[i];
  }
  s1 = wave_sum(s1);
  if (lane == 0) red[0][wave] = s1;
  __syncthreads();
  const float mean = ((red[0][0] + red[0][1]) + (red[0][2] + red[0][3])) / (float)cols;
  float q = 0.f;
#pragma unroll
  for (int i = 0; i < NVW; ++i) {
    const float d = ((wave + LN_WAVES * i) * 64 + lane < cols) ? v[i] - mean : 0.f;
    q += d * d;
  }
  q = wave_sum(q);
  if (lane == 0) red[1][wave] = q;
  __syncthreads();
  const float var = ((red[1][0] + red[1][1]) + (red[1][2] + red[1][3])) / (float)cols;
  const float rstd = 1.0f / sqrtf(var + eps);
  if (unf_L > 0) {
    // The consumer is the distilling convolution (Conv1d k = 3, circular padding 2: layers/TransformerEncoderDecoder.py:12-18):
    // y is written straight in its im2col layout  cols[b][r][3 c + t] = y[b][(r + t - 2) mod L][c],  r in [0, L + 2)
    // (the layout rf_unfold3_circular(pad = 2) produces: the unfold launch between the norm and the product is gone).
    // Source row l feeds r = l + 2 - t, plus the wrapped copies r -+ L that fall inside [0, L + 2).
    const int L = unf_L, b = row / L, l = row - b * L, Lout = L + 2;
    float* cb = y + (long)b * Lout * 3 * cols;
#pragma unroll
    for (int i = 0; i < NVW; ++i) {
      const int c = (wave + LN_WAVES * i) * 64 + lane;
      if (c < cols) {
        const float h = (v[i] - mean) * rstd;
        if (xhat) xhat[off + c] = h;
        const float o = h * gm[i] + bt[i];
#pragma unroll
        for (int t = 0; t < 3; ++t) {
          const int r = l + 2 - t;
          cb[(long)r * 3 * cols + 3 * c + t] = o;
          if (r - L >= 0) cb[(long)(r - L) * 3 * cols + 3 * c + t] = o;
          if (r + L < Lout) cb[(long)(r + L) * 3 * cols + 3 * c + t] = o;
        }
      }
    }
    if (rstd_out && threadIdx.x == 0) rstd_out[row] = rstd;
    return;
  }
#pragma unroll
  for (int i = 0; i < NVW; ++i) {
    const int c = (wave + LN_WAVES * i) * 64 + lane;
    if (c < cols) {
      const float h = (v[i] - mean) * rstd;
      if (xhat) xhat[off + c] = h;
      y[off + c] = h * gm[i] + bt[i];
    }
  }
  if (rstd_out && threadIdx.x == 0) rstd_out[row] = rstd;
}

constexpr int LN_ROW_MIN_COLS = 257;  // rows this wide (and every slab input) take layernorm_row_kernel

template <int NVW, int SC>
void launch_ln_row_nvw(const float* x, int splits, const float* bias, const float* res, const float* gamma,
                       const float* beta, float* y, float* xhat, float* rstd, int rows, int cols, float eps, int seg_rows,
                       long seg_stride, long row_stride, hipStream_t st, int unf_L = 0) {
#define RF_LN_ROW_GO(B_, R_) \
  RF_LAUNCH((layernorm_row_kernel<NVW, SC, B_, R_>), dim3(rows), dim3(256), 0, st, x, splits, bias, res, gamma, beta, y, xhat, \
            rstd, rows, cols, eps, seg_rows, seg_stride, row_stride, unf_L)
  if (bias && res) RF_LN_ROW_GO(true, true);
  else if (bias) RF_LN_ROW_GO(true, false);
  else if (res) RF_LN_ROW_GO(false, true);
  else RF_LN_ROW_GO(false, false);
#undef RF_LN_ROW_GO
}

template <int SC>
void launch_ln_row(const float* x, int splits, const float* bias, const float* res, const float* gamma,
                   const float* beta, float* y, float* xhat, float* rstd, int rows, int cols, float eps, int seg_rows,
                   long seg_stride, long row_stride, hipStream_t st, int unf_L = 0) {
  const int groups = (cols + 63) / 64;
#define RF_LN_ROW_ARGS x, splits, bias, res, gamma, beta, y, xhat, rstd, rows, cols, eps, seg_rows, seg_stride, row_stride, st, unf_L
  if (groups <= LN_WAVES) launch_ln_row_nvw<1, SC>(RF_LN_ROW_ARGS);
  else if (groups <= 2 * LN_WAVES) launch_ln_row_nvw<2, SC>(RF_LN_ROW_ARGS);
  else launch_ln_row_nvw<4, SC>(RF_LN_ROW_ARGS);
#undef RF_LN_ROW_ARGS
}

template <int NV, bool SLABS>
__device__ __forceinline__ void layernorm_bwd_body(const float* __restrict__ dy, const float* __restrict__ xhat,
                                                            const float* __restrict__ rstd,
                                                            const float* __restrict__ gamma, float* __restrict__ dx,
                                                            float* __restrict__ ws, int rows, int cols,
                                                            float* __restrict__ agamma, float* __restrict__ abeta, int fold_L,
                                                            int slabs, const float* __restrict__ slab_res) {
  __shared__ float red[2][LN_WAVES][NV * 64];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  float dg[NV], db[NV], gm[NV];
#pragma unroll
  for (int i = 0; i < NV; ++i) {
    dg[i] = 0.f; db[i] = 0.f;
    gm[i] = gamma[min(i * 64 + lane, cols - 1)];
  }
  for (int row = blockIdx.x * LN_WAVES + wave; row < rows; row += gridDim.x * LN_WAVES) {
    const long off = (long)row * cols;
    float g[NV], h[NV];
    if (fold_L > 0) {
      // dy arrives as the gradient of the im2col image the forward wrote (layernorm_row_kernel, unf_L): the fold --
      // rf_fold3_circular(pad = 2) -- happens on load: dy[b][l][c] = sum over (r, t) with (r + t - 2) mod L == l
      const int L = fold_L, b = row / L, l = row - b * L, Lout = L + 2;
      const float* cb = dy + (long)b * Lout * 3 * cols;
      // per tap: the row r0 = l + 2 - t always contributes; at most ONE wrapped copy (r0 - L or r0 + L) lies inside
      // [0, L + 2) -- loads are unconditional (a missing copy re-reads r0 with weight 0: no branch around a load)
      long ro[3], rw[3];
      float ww[3];
#pragma unroll
      for (int t = 0; t < 3; ++t) {
        const int r0 = l + 2 - t;
        const int rwrap = r0 - L >= 0 ? r0 - L : (r0 + L < Lout ? r0 + L : -1);
        ro[t] = (long)r0 * 3 * cols + t;
        rw[t] = (long)(rwrap >= 0 ? rwrap : r0) * 3 * cols + t;
        ww[t] = rwrap >= 0 ? 1.f : 0.f;
      }
#pragma unroll
      for (int i = 0; i < NV; ++i) {
        const int cc = min(i * 64 + lane, cols - 1);
        float a = 0.f;
#pragma unroll
        for (int t = 0; t < 3; ++t) a += cb[ro[t] + 3 * cc] + ww[t] * cb[rw[t] + 3 * cc];
        g[i] = a;
        h[i] = xhat[off + cc];
      }
    } else if (SLABS) {
      // dy arrives as the split-K slabs of the product in front of this backward (the FFN's last dX: rf_gemm_partials)
      // plus the skip gradient that product's epilogue would have added: summed on load, slab 0 first (the order of the
      // slab-sum launch this replaces)
      // (four slabs in flight per trip, clamped slab index + masked add: a runtime-count loop of load -> add was a chain of
      //  `slabs` dependent round trips per 64-column group)
      const long slab = (long)rows * cols;
      const float* dummy = dy;
#pragma unroll
      for (int i = 0; i < NV; ++i) {
        const int cc = min(i * 64 + lane, cols - 1);
        g[i] = *(slab_res ? slab_res + off + cc : dummy);
        h[i] = xhat[off + cc];
      }
      if (!slab_res) {
#pragma unroll
        for (int i = 0; i < NV; ++i) g[i] = 0.f;
      }
      float acc_s[NV];
#pragma unroll
      for (int i = 0; i < NV; ++i) acc_s[i] = 0.f;
      for (int s0 = 0; s0 < slabs; s0 += 4) {
        float t[4][NV];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
          const float* sp = dy + (long)min(s0 + u, slabs - 1) * slab + off;
#pragma unroll
          for (int i = 0; i < NV; ++i) t[u][i] = sp[min(i * 64 + lane, cols - 1)];
        }
#pragma unroll
        for (int u = 0; u < 4; ++u)
#pragma unroll
          for (int i = 0; i < NV; ++i) acc_s[i] += (s0 + u < slabs) ? t[u][i] : 0.f;
      }
#pragma unroll
      for (int i = 0; i < NV; ++i) g[i] = acc_s[i] + g[i];  // (slab sum first, then the skip gradient: the slab-sum launch's order)
    } else {
#pragma unroll
      for (int i = 0; i < NV; ++i) {
        const int cc = min(i * 64 + lane, cols - 1);  // clamped, unconditional; masked below
        g[i] = dy[off + cc];
        h[i] = xhat[off + cc];
      }
    }
    const float r = rstd[row];
    float s1 = 0.f, s2 = 0.f;
#pragma unroll
    for (int i = 0; i < NV; ++i) {
      const bool in = i * 64 + lane < cols;
      const float d = in ? g[i] : 0.f;
      h[i] = in ? h[i] : 0.f;
      dg[i] += d * h[i];
      db[i] += d;
      g[i] = d * gm[i];
      s1 += g[i];
      s2 += g[i] * h[i];
    }
    const float m1 = wave_sum(s1) / (float)cols, m2 = wave_sum(s2) / (float)cols;
#pragma unroll
    for (int i = 0; i < NV; ++i) {
      const int c = i * 64 + lane;
      if (c < cols) dx[off + c] = r * (g[i] - m1 - h[i] * m2);
    }
  }
  // block partials -> ws[block][0][cols] (dgamma), ws[block][1][cols] (dbeta), or atomics into the running sums: the four
  // waves' sums meet in LDS behind ONE barrier (was four barriers per 64-column group)
#pragma unroll
  for (int i = 0; i < NV; ++i) {
    red[0][wave][i * 64 + lane] = dg[i];
    red[1][wave][i * 64 + lane] = db[i];
  }
  __syncthreads();
  float* wg = agamma ? nullptr : ws + (long)blockIdx.x * 2 * cols;
  for (int c = threadIdx.x; c < cols; c += 256) {
    const float tg = (red[0][0][c] + red[0][1][c]) + (red[0][2][c] + red[0][3][c]);
    const float tb = (red[1][0][c] + red[1][1][c]) + (red[1][2][c] + red[1][3][c]);
    if (agamma) { atomicAdd(&agamma[c], tg); atomicAdd(&abeta[c], tb); }
    else { wg[c] = tg; wg[cols + c] = tb; }
  }
}

// (two kernels: the slab-summing form keeps four slabs x NV loads in flight and needs twice the registers of the plain one)
template <int NV>
__global__ __launch_bounds__(256) void layernorm_bwd_kernel(const float* __restrict__ dy, const float* __restrict__ xhat,
                                                            const float* __restrict__ rstd, const float* __restrict__ gamma,
                                                            float* __restrict__ dx, float* __restrict__ ws, int rows, int cols,
                                                            float* __restrict__ agamma, float* __restrict__ abeta, int fold_L,
                                                            int slabs, const float* __restrict__ slab_res) {
  layernorm_bwd_body<NV, false>(dy, xhat, rstd, gamma, dx, ws, rows, cols, agamma, abeta, fold_L, slabs, slab_res);
}
template <int NV>
__global__ __launch_bounds__(256) void layernorm_bwd_slabs_kernel(const float* __restrict__ dy, const float* __restrict__ xhat,
                                                                  const float* __restrict__ rstd, const float* __restrict__ gamma,
                                                                  float* __restrict__ dx, float* __restrict__ ws, int rows,
                                                                  int cols, float* __restrict__ agamma, float* __restrict__ abeta,
                                                                  int fold_L, int slabs, const float* __restrict__ slab_res) {
  layernorm_bwd_body<NV, true>(dy, xhat, rstd, gamma, dx, ws, rows, cols, agamma, abeta, fold_L, slabs, slab_res);
}

// smallest instantiated NV >= the 64-column groups of a row
#define RF_LN_DISPATCH(KERNEL, cols, ...)                                                     \
  do {                                                                                        \
    const int nv__ = ((cols) + 63) / 64;                                                      \
    if (nv__ <= 1) RF_LAUNCH(KERNEL<1>, __VA_ARGS__);                                         \
    else if (nv__ <= 2) RF_LAUNCH(KERNEL<2>, __VA_ARGS__);                                    \
    else if (nv__ <= 4) RF_LAUNCH(KERNEL<4>, __VA_ARGS__);                                    \
    else if (nv__ <= 8) RF_LAUNCH(KERNEL<8>, __VA_ARGS__);                                    \
    else if (nv__ <= 13) RF_LAUNCH(KERNEL<13>, __VA_ARGS__);                                  \
    else RF_LAUNCH(KERNEL<16>, __VA_ARGS__);                                                  \
  } while (0)

// block = 64 columns x 4 part-lanes: sums the per-block partials of layernorm_bwd_kernel
__global__ __launch_bounds__(256) void ln_param_reduce_kernel(const float* __restrict__ ws, int parts, int cols,
                                                              float* __restrict__ dgamma, float* __restrict__ dbeta,
                                                              int accumulate) {
  __shared__ float red[2][4][64];
  const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
  const int c = blockIdx.x * 64 + tx;
  float a = 0.f, b = 0.f;
  if (c < cols)
    for (int i = ty; i < parts; i += 4) {
      a += ws[(long)i * 2 * cols + c];
      b += ws[(long)i * 2 * cols + cols + c];
    }
  red[0][ty][tx] = a;
  red[1][ty][tx] = b;
  __syncthreads();
  if (ty == 0 && c < cols) {
    const float g = red[0][0][tx] + red[0][1][tx] + red[0][2][tx] + red[0][3][tx];
    const float bt = red[1][0][tx] + red[1][1][tx] + red[1][2][tx] + red[1][3][tx];
    dgamma[c] = accumulate ? dgamma[c] + g : g;
    dbeta[c] = accumulate ? dbeta[c] + bt : bt;
  }
}

// `ld` = row pitch of cols (>= 3 C): columns 3 C .. ld - 1 are written as zeros (a K padded to a multiple of 4 keeps the
// consuming GEMM on its 16-B vector path: c_in = 69 -> 207 columns would otherwise take the scalar kernel)
// One workgroup iteration per output row, 32-bit index arithmetic (the element-indexed form spent its time in 64-bit
// divisions by run-time values: 66 us for the 12 480 x 720 camera-token unfold, now HBM time).
__global__ __launch_bounds__(256) void unfold3_kernel(const float* __restrict__ x, float* __restrict__ cols, int B, int L,
                                                      int C, int pad, int Lout, int ld) {
  const int rows = B * Lout, kc = 3 * C;
  for (int r = blockIdx.x; r < rows; r += gridDim.x) {
    const int l = r % Lout, b = r / Lout;
    int s0 = (l - pad) % L;
    if (s0 < 0) s0 += L;
    const int s1 = s0 + 1 == L ? 0 : s0 + 1, s2 = s1 + 1 == L ? 0 : s1 + 1;
    const float* base = x + (long)b * L * C;
    const float* x0 = base + (long)s0 * C;
    const float* x1 = base + (long)s1 * C;
    const float* x2 = base + (long)s2 * C;
    float* out = cols + (long)r * ld;
    for (int k = threadIdx.x; k < ld; k += 256) {
      float v = 0.f;
      if (k < kc) {
        const int c = k / 3, t = k - 3 * c;
        v = (t == 0 ? x0 : (t == 1 ? x1 : x2))[c];
      }
      out[k] = v;
    }
  }
}

__global__ void fold3_kernel(const float* __restrict__ dcols, float* __restrict__ dx, int B, int L, int C, int pad,
                             int Lout, int ld) {
  const int total = B * L * C;  // (< 2^31, checked by the caller: 32-bit index arithmetic, see unfold3_kernel)
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < total; i += gridDim.x * blockDim.x) {
    const int c = i % C;
    const int r = i / C;
    const int l = r % L;
    const int b = r / L;
    float s = 0.f;
#pragma unroll
    for (int t = 0; t < 3; ++t) {
      int lo = (l - t + pad) % L;
      if (lo < 0) lo += L;
      for (; lo < Lout; lo += L) s += dcols[((long)b * Lout + lo) * ld + c * 3 + t];
    }
    dx[i] = s;
  }
}

// ---- BatchNorm1d statistics over rows (biased variance), block = 64 channels x 4 row lanes ----
__global__ __launch_bounds__(256) void bn_stats_kernel(const float* __restrict__ x, float* __restrict__ mean,
                                                       float* __restrict__ var, int rows, int C,
                                                       float* __restrict__ run_mean, float* __restrict__ run_var,
                                                       long long* __restrict__ batches, float momentum) {
  __shared__ float red[4][64];
  const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
  const int c = blockIdx.x * 64 + tx;
  float s = 0.f;
  if (c < C) for (int r = ty; r < rows; r += 4) s += x[(long)r * C + c];
  red[ty][tx] = s;
  __syncthreads();
  const float m = (red[0][tx] + red[1][tx] + red[2][tx] + red[3][tx]) / (float)rows;
  __syncthreads();
  float q = 0.f;
  if (c < C) for (int r = ty; r < rows; r += 4) { const float d = x[(long)r * C + c] - m; q += d * d; }
  red[ty][tx] = q;
  __syncthreads();
  if (ty == 0 && c < C) {
    const float v = (red[0][tx] + red[1][tx] + red[2][tx] + red[3][tx]) / (float)rows;
    mean[c] = m;
    var[c] = v;
    if (run_mean) {  // nn.BatchNorm1d running statistics (unbiased variance), same launch
      const float keep = 1.0f - momentum;
      const float unbias = momentum * ((float)rows / (float)max(rows - 1, 1));
      run_mean[c] = run_mean[c] * keep + momentum * m;
      run_var[c] = run_var[c] * keep + unbias * v;
    }
  }
  if (batches && blockIdx.x == 0 && threadIdx.x == 0) *batches += 1;
}

__device__ __forceinline__ float elu(float z) { return z > 0.f ? z : expm1f(z); }

__global__ void bn_elu_pool_fwd_kernel(const float* __restrict__ x, const float* __restrict__ mean,
                                       const float* __restrict__ var, const float* __restrict__ gamma,
                                       const float* __restrict__ beta, float* __restrict__ y,
                                       int32_t* __restrict__ argmax, int B, int L, int C, int Lout, float eps) {
  const long total = (long)B * Lout * C;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    const int c = (int)(i % C);
    const long r = i / C;
    const int lo = (int)(r % Lout);
    const int b = (int)(r / Lout);
    const float sc = gamma[c] / sqrtf(var[c] + eps), sh = beta[c] - mean[c] * sc;
    float best = -INFINITY;
    int arg = -1;
#pragma unroll
    for (int t = 0; t < 3; ++t) {
      const int li = 2 * lo - 1 + t;
      if (li < 0 || li >= L) continue;
      const float z = elu(fmaf(x[((long)b * L + li) * C + c], sc, sh));
      if (z > best || arg < 0) { best = z; arg = li; }
    }
    y[i] = best;
    if (argmax) argmax[i] = arg;
  }
}

// One block per 64 channels: (A) dgamma/dbeta sums over all rows, (B) dx.  rows = B*L is small here.
__global__ __launch_bounds__(256) void bn_elu_pool_bwd_kernel(const float* __restrict__ dy,
                                                              const int32_t* __restrict__ argmax,
                                                              const float* __restrict__ x,
                                                              const float* __restrict__ mean,
                                                              const float* __restrict__ var,
                                                              const float* __restrict__ gamma,
                                                              const float* __restrict__ beta, float* __restrict__ dx,
                                                              float* __restrict__ dgamma, float* __restrict__ dbeta,
                                                              int B, int L, int C, int Lout, float eps, int training,
                                                              int accumulate) {
  // block = 16 channels x 16 row lanes: C / 16 workgroups (52 at d_model 832) instead of C / 64, and a
  // 16x shorter serial row loop per thread -- this launch sits alone on the backward critical path
  constexpr int CH = 16, RL = 16;
  __shared__ float red[2][RL][CH];
  const int tx = threadIdx.x % CH, ty = threadIdx.x / CH;
  const int c = blockIdx.x * CH + tx;
  const int rows = B * L;
  float g = 0.f, istd = 0.f, mu = 0.f, bt = 0.f;
  if (c < C) { g = gamma[c]; istd = 1.0f / sqrtf(var[c] + eps); mu = mean[c]; bt = beta[c]; }
  auto dpre = [&](int r, float& xh) -> float {
    // gradient w.r.t. the BN output at row r (through ELU and the max-pool routing)
    const int b = r / L, li = r - b * L;
    xh = (x[(long)r * C + c] - mu) * istd;
    const float z = fmaf(xh, g, bt);
    float dz = 0.f;
    const int lo_min = li / 2, lo_max = (li + 1) / 2;  // windows [2lo-1, 2lo+1] containing li
    for (int lo = lo_min; lo <= lo_max; ++lo)
      if (lo < Lout && argmax[((long)b * Lout + lo) * C + c] == li) dz += dy[((long)b * Lout + lo) * C + c];
    return dz * (z > 0.f ? 1.f : expf(z));
  };
  float s1 = 0.f, s2 = 0.f;
  if (c < C)
    for (int r = ty; r < rows; r += RL) {
      float xh;
      const float d = dpre(r, xh);
      s1 += d;
      s2 += d * xh;
    }
  red[0][ty][tx] = s1;
  red[1][ty][tx] = s2;
  __syncthreads();
  float S1 = 0.f, S2 = 0.f;
#pragma unroll
  for (int k = 0; k < RL; ++k) { S1 += red[0][k][tx]; S2 += red[1][k][tx]; }
  if (c >= C) return;
  if (ty == 0) {
    if (accumulate) { dbeta[c] += S1; dgamma[c] += S2; }
    else { dbeta[c] = S1; dgamma[c] = S2; }
  }
  const float m1 = S1 / (float)rows, m2 = S2 / (float)rows;
  for (int r = ty; r < rows; r += RL) {
    float xh;
    const float d = dpre(r, xh);
    dx[(long)r * C + c] = training ? g * istd * (d - m1 - xh * m2) : g * istd * d;
  }
}

// ---- distilling tail with the channel slab in LDS (layers/TransformerEncoderDecoder.py:9-29) -----------------------
// The Informer's ConvLayer tail works on (B * L) <= a few hundred rows x d_model channels: a workgroup takes a slab of
// BNS_CH channels x ALL rows into LDS with one coalesced pass and does everything else from there -- train-mode batch
// statistics (two-pass variance), the running-statistics update, BatchNorm -> ELU -> MaxPool1d(3, 2, 1); backward: the
// pre-BN gradient through the pool routing and ELU', the two batch sums of BatchNorm's backward and dx.  One launch each
// way instead of statistics + apply launches whose threads walked the rows with strided global loads.
// Channels per workgroup (template parameter BNS_CH, x 256 / BNS_CH row lanes): 8 from 96 rows on, else 32.  At 32 channels
// the d = 832 layers are 26 workgroups whose threads walk up to 42 rows each -- 8-32 us per launch at B L = 168 .. 336, ten
// launches on the chain; 8 channels = 104 workgroups and a quarter of the rows per thread: step 5.306 / 5.306 -> 5.249 / 5.260
// ms (16 channels: -0.01, 4 channels: +0.04 against 8: 16-byte row segments).  Short maps (a few dozen rows: nothing to gain)
// keep the 32-channel form and with it the order in which the batch statistics are summed.
constexpr int bns_ch(int rows) { return rows >= 96 ? 8 : 32; }

template <int BNS_CH>
__global__ __launch_bounds__(256) void bn_train_elu_pool_fwd_kernel(
    const float* __restrict__ x, const float* __restrict__ gamma, const float* __restrict__ beta, float* __restrict__ mean_out,
    float* __restrict__ var_out, float* __restrict__ run_mean, float* __restrict__ run_var, long long* __restrict__ batches,
    float momentum, float* __restrict__ y, int32_t* __restrict__ argmax, int B, int L, int C, int Lout, float eps,
    int x_slabs, const float* __restrict__ x_bias, float* __restrict__ x_out) {
  // x_slabs > 0: x is still the split-K slabs [x_slabs][B * L][C] of the convolution's product; they are summed on load
  // (+ x_bias) and the finished map is written to x_out for the backward pass (rf_bn_train_elu_pool_fwd_slabs)
  constexpr int BNS_RL = 256 / BNS_CH;
  extern __shared__ float slab[];  // [rows][BNS_CH]
  __shared__ float red[BNS_RL][BNS_CH];
  const int tx = threadIdx.x % BNS_CH, ty = threadIdx.x / BNS_CH;
  const int c = blockIdx.x * BNS_CH + tx, cc = min(c, C - 1);
  const int rows = B * L;
  float s = 0.f;
  const float xb = (x_slabs > 0 && x_bias) ? x_bias[cc] : 0.f;
  // eight rows in flight per thread (clamped, unconditional loads): a load -> LDS store -> next load loop is one memory
  // round trip per row, 40 of them at B L = 320 (27 us for a 1-MB tensor)
  for (int r0 = ty; r0 < rows; r0 += BNS_RL * 8) {
    float v[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) v[j] = x[(long)min(r0 + j * BNS_RL, rows - 1) * C + cc];
    if (x_slabs > 0) {
      const long sl = (long)rows * C;
      for (int s0 = 1; s0 < x_slabs; s0 += 2) {  // two more slabs per trip (v holds slab 0): clamped index, masked add
        float t[2][8];
#pragma unroll
        for (int u = 0; u < 2; ++u)
#pragma unroll
          for (int j = 0; j < 8; ++j) t[u][j] = x[(long)min(s0 + u, x_slabs - 1) * sl + (long)min(r0 + j * BNS_RL, rows - 1) * C + cc];
#pragma unroll
        for (int u = 0; u < 2; ++u)
#pragma unroll
          for (int j = 0; j < 8; ++j) v[j] += (s0 + u < x_slabs) ? t[u][j] : 0.f;
      }
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        v[j] += xb;
        const int r = r0 + j * BNS_RL;
        if (r < rows && c < C) x_out[(long)r * C + c] = v[j];
      }
    }
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const int r = r0 + j * BNS_RL;
      if (r < rows) { slab[r * BNS_CH + tx] = v[j]; s += v[j]; }
    }
  }
  red[ty][tx] = s;
  __syncthreads();
  float m = 0.f;
#pragma unroll
  for (int k = 0; k < BNS_RL; ++k) m += red[k][tx];
  m /= (float)rows;
  __syncthreads();
  float q = 0.f;
  for (int r = ty; r < rows; r += BNS_RL) { const float d = slab[r * BNS_CH + tx] - m; q += d * d; }
  red[ty][tx] = q;
  __syncthreads();
  float v = 0.f;
#pragma unroll
  for (int k = 0; k < BNS_RL; ++k) v += red[k][tx];
  v /= (float)rows;
  if (ty == 0 && c < C) {
    mean_out[c] = m;
    var_out[c] = v;
    if (run_mean) {  // nn.BatchNorm1d running statistics (unbiased variance)
      const float keep = 1.0f - momentum;
      const float unbias = momentum * ((float)rows / (float)max(rows - 1, 1));
      run_mean[c] = run_mean[c] * keep + momentum * m;
      run_var[c] = run_var[c] * keep + unbias * v;
    }
  }
  if (batches && blockIdx.x == 0 && threadIdx.x == 0) *batches += 1;
  if (c >= C) return;
  const float sc = gamma[c] / sqrtf(v + eps), sh = beta[c] - m * sc;
  for (int o = ty; o < B * Lout; o += BNS_RL) {
    const int b = o / Lout, lo = o - b * Lout;
    float best = -INFINITY;
    int arg = -1;
#pragma unroll
    for (int t = 0; t < 3; ++t) {
      const int li = 2 * lo - 1 + t;
      if (li < 0 || li >= L) continue;
      const float z = elu(fmaf(slab[(b * L + li) * BNS_CH + tx], sc, sh));
      if (z > best || arg < 0) { best = z; arg = li; }
    }
    y[(long)o * C + c] = best;
    argmax[(long)o * C + c] = arg;
  }
}

template <int BNS_CH>
__global__ __launch_bounds__(256) void bn_elu_pool_bwd_slab_kernel(
    const float* __restrict__ dy, const int32_t* __restrict__ argmax, const float* __restrict__ x, const float* __restrict__ mean,
    const float* __restrict__ var, const float* __restrict__ gamma, const float* __restrict__ beta, float* __restrict__ dx,
    float* __restrict__ dgamma, float* __restrict__ dbeta, int B, int L, int C, int Lout, float eps, int training, int accumulate,
    int dy_slabs, const float* __restrict__ dy_res) {
  // dy_slabs > 0: dy is still the split-K slabs [dy_slabs][B * Lout][C] of the dX product in front of this backward plus the
  // skip gradient dy_res (or null): summed on load (rf_bn_elu_pool_bwd_slabs)
  constexpr int BNS_RL = 256 / BNS_CH;
  extern __shared__ float slab[];  // xh [rows][BNS_CH], then d [rows][BNS_CH]
  __shared__ float red[2][BNS_RL][BNS_CH];
  const int tx = threadIdx.x % BNS_CH, ty = threadIdx.x / BNS_CH;
  const int c = blockIdx.x * BNS_CH + tx, cc = min(c, C - 1);
  const int rows = B * L;
  float* xh = slab;
  float* dd = slab + (long)rows * BNS_CH;
  const float g = gamma[cc], istd = 1.0f / sqrtf(var[cc] + eps), mu = mean[cc], bt = beta[cc];
  for (int r0 = ty; r0 < rows; r0 += BNS_RL * 8) {  // eight rows in flight per thread (see the forward kernel)
    float v[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) v[j] = x[(long)min(r0 + j * BNS_RL, rows - 1) * C + cc];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const int r = r0 + j * BNS_RL;
      if (r < rows) { xh[r * BNS_CH + tx] = (v[j] - mu) * istd; dd[r * BNS_CH + tx] = 0.f; }
    }
  }
  __syncthreads();
  // pool routing: every pooled output sends its gradient to its arg-max row (windows overlap in one row: two outputs of a
  // channel may name the same row -- the (b, lo) loop of one thread column runs over lo with stride BNS_RL, so the adds
  // of a column go through LDS atomics)
  const int outs = B * Lout;
  for (int o0 = ty; o0 < outs; o0 += BNS_RL * 4) {
    int li[4];
    float gy[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const long at = (long)min(o0 + j * BNS_RL, outs - 1) * C + cc;
      li[j] = argmax[at];
      gy[j] = dy[at];
    }
    if (dy_slabs > 0) {  // (gy holds slab 0; the others four at a time, clamped index + masked add, then the skip gradient)
      const long sl = (long)outs * C;
      for (int s0 = 1; s0 < dy_slabs; s0 += 4) {
        float t[4][4];
#pragma unroll
        for (int u = 0; u < 4; ++u)
#pragma unroll
          for (int j = 0; j < 4; ++j) t[u][j] = dy[(long)min(s0 + u, dy_slabs - 1) * sl + (long)min(o0 + j * BNS_RL, outs - 1) * C + cc];
#pragma unroll
        for (int u = 0; u < 4; ++u)
#pragma unroll
          for (int j = 0; j < 4; ++j) gy[j] += (s0 + u < dy_slabs) ? t[u][j] : 0.f;
      }
      if (dy_res) {
#pragma unroll
        for (int j = 0; j < 4; ++j) gy[j] += dy_res[(long)min(o0 + j * BNS_RL, outs - 1) * C + cc];
      }
    }
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int o = o0 + j * BNS_RL;
      if (o < outs) atomicAdd(&dd[((o / Lout) * L + li[j]) * BNS_CH + tx], gy[j]);
    }
  }
  __syncthreads();
  float s1 = 0.f, s2 = 0.f;
  for (int r = ty; r < rows; r += BNS_RL) {
    const float h = xh[r * BNS_CH + tx];
    const float z = fmaf(h, g, bt);
    const float d = dd[r * BNS_CH + tx] * (z > 0.f ? 1.f : expf(z));  // through ELU'
    dd[r * BNS_CH + tx] = d;
    s1 += d;
    s2 += d * h;
  }
  red[0][ty][tx] = s1;
  red[1][ty][tx] = s2;
  __syncthreads();
  float S1 = 0.f, S2 = 0.f;
#pragma unroll
  for (int k = 0; k < BNS_RL; ++k) { S1 += red[0][k][tx]; S2 += red[1][k][tx]; }
  if (c >= C) return;
  if (ty == 0) {
    if (accumulate) { dbeta[c] += S1; dgamma[c] += S2; }
    else { dbeta[c] = S1; dgamma[c] = S2; }
  }
  const float m1 = S1 / (float)rows, m2 = S2 / (float)rows;
  for (int r = ty; r < rows; r += BNS_RL) {
    const float d = dd[r * BNS_CH + tx], h = xh[r * BNS_CH + tx];
    dx[(long)r * C + c] = training ? g * istd * (d - m1 - h * m2) : g * istd * d;
  }
}

constexpr int BNS_MAX_LDS = 96 * 1024;

inline int grid_for(long total, int block = 256, int cap = 4096) {
  long g = (total + block - 1) / block;
  return (int)(g > cap ? cap : (g < 1 ? 1 : g));
}

}  // namespace

// x row r lives at x + (r / seg_rows) * seg_stride + (r % seg_rows) * row_stride (elements); everything else is contiguous
extern "C" int rf_layernorm_fwd_strided(const float* x, int seg_rows, int64_t seg_stride, int64_t row_stride,
                                        const float* residual, const float* gamma, const float* beta, float* y, float* xhat,
                                        float* rstd, int rows, int cols, float eps, void* stream) {
  RF_REQUIRE(x && gamma && beta && y && rows > 0 && cols > 0 && cols <= 64 * LN_MAXV && seg_rows > 0);
  if (cols >= LN_ROW_MIN_COLS) {
    launch_ln_row<1>(x, 1, nullptr, residual, gamma, beta, y, xhat, rstd, rows, cols, eps, seg_rows, (long)seg_stride,
                     (long)row_stride, static_cast<hipStream_t>(stream));
    RF_CHECK_LAUNCH();
    return RF_OK;
  }
  RF_LN_DISPATCH(layernorm_fwd_kernel, cols, dim3((rows + LN_WAVES - 1) / LN_WAVES), dim3(256), 0,
                 static_cast<hipStream_t>(stream), x, residual, gamma, beta, y, xhat, rstd, rows, cols, eps, seg_rows,
                 (long)seg_stride, (long)row_stride);
  RF_CHECK_LAUNCH();
  return RF_OK;
}

extern "C" int rf_layernorm_fwd(const float* x, const float* residual, const float* gamma, const float* beta,
                                float* y, float* xhat, float* rstd, int rows, int cols, float eps, void* stream) {
  return rf_layernorm_fwd_strided(x, rows > 0 ? rows : 1, 0, cols, residual, gamma, beta, y, xhat, rstd, rows, cols, eps, stream);
}

extern "C" int rf_layernorm_fwd_slabs(const float* slabs, int splits, const float* bias, const float* residual,
                                      const float* gamma, const float* beta, float* y, float* xhat, float* rstd, int rows,
                                      int cols, float eps, void* stream) {
  RF_REQUIRE(slabs && splits >= 1 && gamma && beta && y && rows > 0 && cols > 0 && cols <= 64 * LN_MAXV);
  // one slab is the plain norm's arithmetic: the same instantiation, so that the two entry points agree to the bit
  if (splits == 1)
    launch_ln_row<1>(slabs, 1, bias, residual, gamma, beta, y, xhat, rstd, rows, cols, eps, rows, 0, cols,
                     static_cast<hipStream_t>(stream));
  else
    launch_ln_row<4>(slabs, splits, bias, residual, gamma, beta, y, xhat, rstd, rows, cols, eps, rows, 0, cols,
                     static_cast<hipStream_t>(stream));
  RF_CHECK_LAUNCH();
  return RF_OK;
}

extern "C" int rf_layernorm_fwd_slabs_unfold(const float* slabs, int splits, const float* bias, const float* residual,
                                             const float* gamma, const float* beta, float* cols_out, float* xhat, float* rstd,
                                             int rows, int cols, int L, float eps, void* stream) {
  RF_REQUIRE(slabs && splits >= 1 && gamma && beta && cols_out && rows > 0 && cols >= LN_ROW_MIN_COLS && cols <= 64 * LN_MAXV);
  RF_REQUIRE(L >= 2 && rows % L == 0);
  if (splits == 1)
    launch_ln_row<1>(slabs, 1, bias, residual, gamma, beta, cols_out, xhat, rstd, rows, cols, eps, rows, 0, cols,
                     static_cast<hipStream_t>(stream), L);
  else
    launch_ln_row<4>(slabs, splits, bias, residual, gamma, beta, cols_out, xhat, rstd, rows, cols, eps, rows, 0, cols,
                     static_cast<hipStream_t>(stream), L);
  RF_CHECK_LAUNCH();
  return RF_OK;
}

extern "C" int rf_layernorm_bwd_parts(int rows) {
  // enough workgroups to cover all 256 CUs a few times over; each writes one (dgamma, dbeta) partial row
  const int blocks = (rows + 4 * LN_WAVES - 1) / (4 * LN_WAVES);
  return blocks > 768 ? 768 : (blocks < 1 ? 1 : blocks);
}

static int layernorm_bwd_run(const float* dy, const float* xhat, const float* rstd, const float* gamma, float* dx,
                             float* dgamma, float* dbeta, int accumulate, float* workspace, int rows, int cols, int fold_L,
                             void* stream, int slabs = 0, const float* slab_res = nullptr);

extern "C" int rf_layernorm_bwd(const float* dy, const float* xhat, const float* rstd, const float* gamma, float* dx,
                                float* dgamma, float* dbeta, int accumulate, float* workspace, int rows, int cols,
                                void* stream) {
  return layernorm_bwd_run(dy, xhat, rstd, gamma, dx, dgamma, dbeta, accumulate, workspace, rows, cols, 0, stream);
}

extern "C" int rf_layernorm_bwd_fold(const float* dcols, const float* xhat, const float* rstd, const float* gamma, float* dx,
                                     float* dgamma, float* dbeta, int accumulate, float* workspace, int rows, int cols, int L,
                                     void* stream) {
  RF_REQUIRE(L >= 2 && rows % L == 0);
  return layernorm_bwd_run(dcols, xhat, rstd, gamma, dx, dgamma, dbeta, accumulate, workspace, rows, cols, L, stream);
}

extern "C" int rf_layernorm_bwd_slabs(const float* slabs, int splits, const float* residual, const float* xhat, const float* rstd,
                                      const float* gamma, float* dx, float* dgamma, float* dbeta, int accumulate, float* workspace,
                                      int rows, int cols, void* stream) {
  RF_REQUIRE(splits >= 1 && splits <= 64);
  return layernorm_bwd_run(slabs, xhat, rstd, gamma, dx, dgamma, dbeta, accumulate, workspace, rows, cols, 0, stream, splits,
                           residual);
}

static int layernorm_bwd_run(const float* dy, const float* xhat, const float* rstd, const float* gamma, float* dx,
                             float* dgamma, float* dbeta, int accumulate, float* workspace, int rows, int cols, int fold_L,
                             void* stream, int slabs, const float* slab_res) {
  RF_REQUIRE(dy && xhat && rstd && gamma && dx && dgamma && dbeta && (workspace || accumulate == 2));
  RF_REQUIRE(rows > 0 && cols > 0 && cols <= 64 * LN_MAXV);
  hipStream_t st = static_cast<hipStream_t>(stream);
  if (accumulate == 2) {  // atomics straight into the running sums; few workgroups -> few same-address adds
    // one row per wave until 512 workgroups (M <= 560 layers: 80-140 workgroups instead of 20-35 -- these
    // launches are latency-bound), then grid-stride
    int blocks = (rows + LN_WAVES - 1) / LN_WAVES;
    blocks = blocks > 512 ? 512 : (blocks < 1 ? 1 : blocks);  // (128 .. 1 024 measured: no difference)
    if (slabs > 0)
      RF_LN_DISPATCH(layernorm_bwd_slabs_kernel, cols, dim3(blocks), dim3(256), 0, st, dy, xhat, rstd, gamma, dx,
                     static_cast<float*>(nullptr), rows, cols, dgamma, dbeta, 0, slabs, slab_res);
    else
      RF_LN_DISPATCH(layernorm_bwd_kernel, cols, dim3(blocks), dim3(256), 0, st, dy, xhat, rstd, gamma, dx,
                     static_cast<float*>(nullptr), rows, cols, dgamma, dbeta, fold_L, slabs, slab_res);
    RF_CHECK_LAUNCH();
    return RF_OK;
  }
  const int parts = rf_layernorm_bwd_parts(rows);
  if (slabs > 0)
    RF_LN_DISPATCH(layernorm_bwd_slabs_kernel, cols, dim3(parts), dim3(256), 0, st, dy, xhat, rstd, gamma, dx, workspace, rows,
                   cols, static_cast<float*>(nullptr), static_cast<float*>(nullptr), 0, slabs, slab_res);
  else
    RF_LN_DISPATCH(layernorm_bwd_kernel, cols, dim3(parts), dim3(256), 0, st, dy, xhat, rstd, gamma, dx, workspace, rows,
                   cols, static_cast<float*>(nullptr), static_cast<float*>(nullptr), fold_L, slabs, slab_res);
  RF_CHECK_LAUNCH();
  RF_LAUNCH(ln_param_reduce_kernel, dim3((cols + 63) / 64), dim3(256), 0, st, workspace, parts, cols,
                     dgamma, dbeta, accumulate);
  RF_CHECK_LAUNCH();
  return RF_OK;
}

extern "C" int rf_unfold3_circular_ld(const float* x, float* cols, int B, int L, int C, int pad, int ld, void* stream) {
  RF_REQUIRE(x && cols && B > 0 && L > 0 && C > 0 && pad >= 1 && pad <= 2 && ld >= 3 * C);
  const int Lout = L + 2 * pad - 2;
  RF_REQUIRE((long)B * Lout < (1L << 31));
  const long rows = (long)B * Lout;
  RF_LAUNCH(unfold3_kernel, dim3((unsigned)(rows > 16384 ? 16384 : rows)), dim3(256), 0,
                     static_cast<hipStream_t>(stream), x, cols, B, L, C, pad, Lout, ld);
  RF_CHECK_LAUNCH();
  return RF_OK;
}

extern "C" int rf_unfold3_circular(const float* x, float* cols, int B, int L, int C, int pad, void* stream) {
  return rf_unfold3_circular_ld(x, cols, B, L, C, pad, 3 * C, stream);
}

extern "C" int rf_fold3_circular_ld(const float* dcols, float* dx, int B, int L, int C, int pad, int ld, void* stream) {
  RF_REQUIRE(dcols && dx && B > 0 && L > 0 && C > 0 && pad >= 1 && pad <= 2 && ld >= 3 * C);
  RF_REQUIRE((long)B * L * C < (1L << 31) - (4096L * 256));
  const int Lout = L + 2 * pad - 2;
  RF_LAUNCH(fold3_kernel, dim3(grid_for((long)B * L * C)), dim3(256), 0, static_cast<hipStream_t>(stream),
                     dcols, dx, B, L, C, pad, Lout, ld);
  RF_CHECK_LAUNCH();
  return RF_OK;
}

extern "C" int rf_fold3_circular(const float* dcols, float* dx, int B, int L, int C, int pad, void* stream) {
  return rf_fold3_circular_ld(dcols, dx, B, L, C, pad, 3 * C, stream);
}

extern "C" int rf_bn_stats(const float* x, float* mean, float* var, int rows, int C, float* running_mean,
                           float* running_var, int64_t* num_batches_tracked, float momentum, void* stream) {
  RF_REQUIRE(x && mean && var && rows > 0 && C > 0 && (!running_mean == !running_var));
  RF_LAUNCH(bn_stats_kernel, dim3((C + 63) / 64), dim3(256), 0, static_cast<hipStream_t>(stream), x, mean,
                     var, rows, C, running_mean, running_var, reinterpret_cast<long long*>(num_batches_tracked),
                     momentum);
  RF_CHECK_LAUNCH();
  return RF_OK;
}

extern "C" int rf_bn_elu_pool_fwd(const float* x, const float* mean, const float* var, const float* gamma,
                                  const float* beta, float* y, int32_t* argmax, int B, int L, int C, float eps,
                                  void* stream) {
  RF_REQUIRE(x && mean && var && gamma && beta && y && B > 0 && L > 0 && C > 0);
  const int Lout = (L - 1) / 2 + 1;
  RF_LAUNCH(bn_elu_pool_fwd_kernel, dim3(grid_for((long)B * Lout * C)), dim3(256), 0,
                     static_cast<hipStream_t>(stream), x, mean, var, gamma, beta, y, argmax, B, L, C, Lout, eps);
  RF_CHECK_LAUNCH();
  return RF_OK;
}

static int bn_elu_pool_bwd_run(const float* dy, const int32_t* argmax, const float* x, const float* mean,
                               const float* var, const float* gamma, const float* beta, float* dx, float* dgamma,
                               float* dbeta, int accumulate, int B, int L, int C, float eps, int training,
                               void* stream, int g_bn_dy_slabs, const float* g_bn_dy_res) {
  RF_REQUIRE(dy && argmax && x && mean && var && gamma && beta && dx && dgamma && dbeta && B > 0 && L > 0 && C > 0);
  const int Lout = (L - 1) / 2 + 1;
  const int ch = bns_ch(B * L);
  const size_t lds = (size_t)2 * B * L * ch * sizeof(float);
  if (lds <= (size_t)BNS_MAX_LDS) {  // channel slab in LDS (the Informer's distilling layers: B * L <= a few hundred rows)
    static bool attr = false;
    if (!attr) {
      (void)hipFuncSetAttribute(reinterpret_cast<const void*>(bn_elu_pool_bwd_slab_kernel<8>),
                                hipFuncAttributeMaxDynamicSharedMemorySize, BNS_MAX_LDS);
      (void)hipFuncSetAttribute(reinterpret_cast<const void*>(bn_elu_pool_bwd_slab_kernel<32>),
                                hipFuncAttributeMaxDynamicSharedMemorySize, BNS_MAX_LDS);
      attr = true;
    }
    if (ch == 8)
      RF_LAUNCH(bn_elu_pool_bwd_slab_kernel<8>, dim3((C + 7) / 8), dim3(256), lds, static_cast<hipStream_t>(stream), dy,
                argmax, x, mean, var, gamma, beta, dx, dgamma, dbeta, B, L, C, Lout, eps, training, accumulate, g_bn_dy_slabs,
                g_bn_dy_res);
    else
      RF_LAUNCH(bn_elu_pool_bwd_slab_kernel<32>, dim3((C + 31) / 32), dim3(256), lds, static_cast<hipStream_t>(stream), dy,
                argmax, x, mean, var, gamma, beta, dx, dgamma, dbeta, B, L, C, Lout, eps, training, accumulate, g_bn_dy_slabs,
                g_bn_dy_res);
    RF_CHECK_LAUNCH();
    return RF_OK;
  }
  if (g_bn_dy_slabs > 0) {
    rf_g_last_error = "rf_bn_elu_pool_bwd_slabs: B * L rows do not fit the LDS slab";
    return RF_EUNSUPPORTED;
  }
  RF_LAUNCH(bn_elu_pool_bwd_kernel, dim3((C + 15) / 16), dim3(256), 0, static_cast<hipStream_t>(stream), dy,
                     argmax, x, mean, var, gamma, beta, dx, dgamma, dbeta, B, L, C, Lout, eps, training, accumulate);
  RF_CHECK_LAUNCH();
  return RF_OK;
}

extern "C" int rf_bn_elu_pool_bwd_slab_ok(int B, int L) {
  return (size_t)2 * B * L * bns_ch(B * L) * sizeof(float) <= (size_t)BNS_MAX_LDS ? 1 : 0;
}

// rf_bn_elu_pool_bwd whose incoming gradient is still `splits` split-K slabs [splits][B * Lout][C] (+ an optional skip gradient
// `residual` [B * Lout][C]): summed on load.  Only with the LDS-slab kernel (rf_bn_elu_pool_bwd_slab_ok).
extern "C" int rf_bn_elu_pool_bwd_slabs(const float* slabs, int splits, const float* residual, const int32_t* argmax, const float* x,
                                        const float* mean, const float* var, const float* gamma, const float* beta, float* dx,
                                        float* dgamma, float* dbeta, int accumulate, int B, int L, int C, float eps, int training,
                                        void* stream) {
  RF_REQUIRE(splits >= 1 && splits <= 64 && rf_bn_elu_pool_bwd_slab_ok(B, L));
  return bn_elu_pool_bwd_run(slabs, argmax, x, mean, var, gamma, beta, dx, dgamma, dbeta, accumulate, B, L, C, eps, training, stream,
                             splits, residual);
}

extern "C" int rf_bn_elu_pool_bwd(const float* dy, const int32_t* argmax, const float* x, const float* mean,
                                  const float* var, const float* gamma, const float* beta, float* dx, float* dgamma,
                                  float* dbeta, int accumulate, int B, int L, int C, float eps, int training,
                                  void* stream) {
  return bn_elu_pool_bwd_run(dy, argmax, x, mean, var, gamma, beta, dx, dgamma, dbeta, accumulate, B, L, C, eps, training, stream, 0,
                             nullptr);
}

// Train-mode forward in ONE launch: batch statistics (+ running-statistics update) -> BatchNorm -> ELU -> MaxPool; mean / var
// (biased) are written for the backward pass.  Returns RF_EUNSUPPORTED when the row slab does not fit LDS (the caller then
// uses rf_bn_stats + rf_bn_elu_pool_fwd).
static int bn_train_fwd_run(const float* x, const float* gamma, const float* beta, float* mean, float* var,
                            float* running_mean, float* running_var, int64_t* num_batches_tracked, float momentum,
                            float* y, int32_t* argmax, int B, int L, int C, float eps, void* stream, int x_slabs,
                            const float* x_bias, float* x_out);

extern "C" int rf_bn_train_elu_pool_fwd(const float* x, const float* gamma, const float* beta, float* mean, float* var,
                                        float* running_mean, float* running_var, int64_t* num_batches_tracked, float momentum,
                                        float* y, int32_t* argmax, int B, int L, int C, float eps, void* stream) {
  return bn_train_fwd_run(x, gamma, beta, mean, var, running_mean, running_var, num_batches_tracked, momentum, y, argmax, B, L, C,
                          eps, stream, 0, nullptr, nullptr);
}

// The same launch fed with the `splits` split-K slabs [splits][B * L][C] of the convolution's product (+ its bias): they are
// summed on load and the finished pre-normalisation map is written to x_out ([B * L][C]: what the backward pass reads).
extern "C" int rf_bn_train_elu_pool_fwd_slabs(const float* slabs, int splits, const float* bias, float* x_out, const float* gamma,
                                              const float* beta, float* mean, float* var, float* running_mean, float* running_var,
                                              int64_t* num_batches_tracked, float momentum, float* y, int32_t* argmax, int B, int L,
                                              int C, float eps, void* stream) {
  RF_REQUIRE(splits >= 1 && splits <= 64 && x_out);
  return bn_train_fwd_run(slabs, gamma, beta, mean, var, running_mean, running_var, num_batches_tracked, momentum, y, argmax, B, L, C,
                          eps, stream, splits, bias, x_out);
}

static int bn_train_fwd_run(const float* x, const float* gamma, const float* beta, float* mean, float* var,
                            float* running_mean, float* running_var, int64_t* num_batches_tracked, float momentum,
                            float* y, int32_t* argmax, int B, int L, int C, float eps, void* stream, int x_slabs,
                            const float* x_bias, float* x_out) {
  RF_REQUIRE(x && gamma && beta && mean && var && y && argmax && B > 0 && L > 0 && C > 0 && (!running_mean == !running_var));
  const int Lout = (L - 1) / 2 + 1;
  const int ch = bns_ch(B * L);
  const size_t lds = (size_t)B * L * ch * sizeof(float);
  if (lds > (size_t)BNS_MAX_LDS) {
    rf_g_last_error = "rf_bn_train_elu_pool_fwd: B * L rows do not fit the LDS slab";
    return RF_EUNSUPPORTED;
  }
  static bool attr = false;
  if (!attr) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(bn_train_elu_pool_fwd_kernel<8>),
                              hipFuncAttributeMaxDynamicSharedMemorySize, BNS_MAX_LDS);
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(bn_train_elu_pool_fwd_kernel<32>),
                              hipFuncAttributeMaxDynamicSharedMemorySize, BNS_MAX_LDS);
    attr = true;
  }
  if (ch == 8)
    RF_LAUNCH(bn_train_elu_pool_fwd_kernel<8>, dim3((C + 7) / 8), dim3(256), lds, static_cast<hipStream_t>(stream), x,
              gamma, beta, mean, var, running_mean, running_var, reinterpret_cast<long long*>(num_batches_tracked), momentum, y,
              argmax, B, L, C, Lout, eps, x_slabs, x_bias, x_out);
  else
    RF_LAUNCH(bn_train_elu_pool_fwd_kernel<32>, dim3((C + 31) / 32), dim3(256), lds, static_cast<hipStream_t>(stream), x,
              gamma, beta, mean, var, running_mean, running_var, reinterpret_cast<long long*>(num_batches_tracked), momentum, y,
              argmax, B, L, C, Lout, eps, x_slabs, x_bias, x_out);
  RF_CHECK_LAUNCH();
  return RF_OK;
}

// ---------------------------------------------------------------------------------------------------
// Trajectory head: postprocess_batch (routeformer.py:367-374: de-normalise, cumsum to positions) + the
// train-step loss recipe (full_comparison.py:490-521: gamma^t-discounted SmoothL1 on positions and on the
// dense visual head, dense weight = ratio * traj / max(dense, 1e-6) once enabled, ADE, FDE) in ONE
// single-workgroup launch each way -- these are ~90 tiny dependent torch kernels sitting exactly between the
// forward and the backward pass, where nothing else can overlap them.
// scal[0..4] = {traj, dense, ade, fde, loss};  gpos = gamma^t * sl1'(pos - tgt) (un-normalised).
// ---------------------------------------------------------------------------------------------------
namespace {

__device__ __forceinline__ float sl1(float d) { const float a = fabsf(d); return a < 1.f ? 0.5f * d * d : a - 0.5f; }
__device__ __forceinline__ float sl1g(float d) { return fabsf(d) < 1.f ? d : (d > 0.f ? 1.f : -1.f); }

constexpr int TH_NT = 1024;  // threads of the single workgroup (16 waves: the element loops are pure load latency)

__device__ float block_sum(float v, float* red) {
  v = wave_sum(v);
  __syncthreads();
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
  __syncthreads();
  float t = 0.f;
#pragma unroll
  for (int w = 0; w < TH_NT / 64; ++w) t += red[w];  // fixed order
  return t;
}

// Both kernels are ONE workgroup on a few thousand elements -- pure latency, on the critical path between the forward and
// the backward pass.  The motion channels and the discount table gamma^t go through LDS first (one coalesced round trip),
// so the horizon-long cumulative sums run out of LDS instead of as P dependent global loads, and no thread calls powf
// in a loop.
constexpr int TH_MAXP = 256, TH_MAXROWS = 2048;  // horizon / B * P bounds of the LDS staging (else the direct path)

__global__ __launch_bounds__(TH_NT) void traj_head_fwd_kernel(const float* __restrict__ out, const float* __restrict__ last,
                                                            const float* __restrict__ tgt, const float* __restrict__ tvis,
                                                            float* __restrict__ pos, float* __restrict__ gpos,
                                                            float* __restrict__ scal, int B, int P, int C, int E,
                                                            float gamma, float ratio, int dense_on, float mstd,
                                                            float mmean, long out_bs, long last_bs, long tvis_bs) {
  __shared__ float red[TH_NT / 64];
  __shared__ float disc[TH_MAXP];
  __shared__ float mot[TH_MAXROWS * 2];
  const int tid = threadIdx.x;
  const bool staged = P <= TH_MAXP && B * P <= TH_MAXROWS;
  if (staged) {
    for (int t = tid; t < P; t += TH_NT) disc[t] = powf(gamma, (float)t);
    for (int i = tid; i < B * P * 2; i += TH_NT) mot[i] = out[(long)((i >> 1) / P) * out_bs + ((i >> 1) % P) * C + (i & 1)] * mstd + mmean;
    __syncthreads();
  }
  // positions: one thread per (b, coordinate), sequential cumsum over the horizon (P <= a few dozen)
  for (int i = tid; i < B * 2; i += TH_NT) {
    const int b = i >> 1, c = i & 1;
    float run = last[(long)b * last_bs + c];
    for (int t = 0; t < P; ++t) {
      run += staged ? mot[(b * P + t) * 2 + c] : out[(long)b * out_bs + t * C + c] * mstd + mmean;
      pos[((long)b * P + t) * 2 + c] = run;
      if (staged) mot[(b * P + t) * 2 + c] = run;  // keep the position: the loss loop below reads it from LDS
    }
  }
  __syncthreads();
  float s_traj = 0.f, s_ade = 0.f, s_fde = 0.f;
  for (int i = tid; i < B * P; i += TH_NT) {
    const int b = i / P, t = i - b * P;
    const float w = staged ? disc[t] : powf(gamma, (float)t);
    const float px = staged ? mot[i * 2] : pos[i * 2], py = staged ? mot[i * 2 + 1] : pos[i * 2 + 1];
    const float dx = px - tgt[i * 2], dy = py - tgt[i * 2 + 1];
    s_traj += w * (sl1(dx) + sl1(dy));
    gpos[i * 2] = w * sl1g(dx);
    gpos[i * 2 + 1] = w * sl1g(dy);
    const float d2 = dx * dx + dy * dy;
    s_ade += sqrtf(d2);
    if (b == B - 1) s_fde += d2;
  }
  float s_dense = 0.f;
  if (tvis) {
    // one (b, t) row of E channels per 16-lane group pass: the discount is looked up once per row, no div / mod per element
    // flat over the (b, t, e) elements, eight loads of each operand in flight per thread (this single workgroup sits
    // between the forward and the backward pass: a dependent load per row pass cost 15 us here)
    const int total = B * P * E;
    for (int i0 = tid; i0 < total; i0 += TH_NT * 4) {
      float o[4], tv[4];
      int bt[4];
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const int i = min(i0 + TH_NT * j, total - 1);
        bt[j] = i / E;
        o[j] = out[(long)(bt[j] / P) * out_bs + (bt[j] % P) * C + 2 + (i - bt[j] * E)];
        tv[j] = tvis[(long)(bt[j] / P) * tvis_bs + (bt[j] % P) * E + (i - bt[j] * E)];
      }
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        if (i0 + TH_NT * j < total) {
          const int t = bt[j] % P;
          s_dense += (staged ? disc[t] : powf(gamma, (float)t)) * sl1(o[j] - tv[j]);
        }
      }
    }
  }
  const float traj = block_sum(s_traj, red) / (float)(B * P * 2);
  const float ade = block_sum(s_ade, red) / (float)(B * P);
  const float fde = sqrtf(block_sum(s_fde, red));
  const float dense = tvis ? block_sum(s_dense, red) / (float)((long)B * P * E) : 0.f;
  if (tid == 0) {
    const float w = (tvis && dense_on) ? ratio * traj / fmaxf(dense, 1e-6f) : 0.f;
    scal[0] = traj; scal[1] = dense; scal[2] = ade; scal[3] = fde; scal[4] = traj + w * dense; scal[5] = w;
  }
}

__global__ __launch_bounds__(TH_NT) void traj_head_bwd_kernel(const float* __restrict__ out, const float* __restrict__ tvis,
                                                            const float* __restrict__ gpos, const float* __restrict__ scal,
                                                            const float* __restrict__ gloss, float* __restrict__ dout,
                                                            int B, int P, int C, int E, float gamma, float mstd, long out_bs,
                                                            long tvis_bs) {
  __shared__ float disc[TH_MAXP];
  __shared__ float gp[TH_MAXROWS * 2];
  const int tid = threadIdx.x;
  const float g = gloss ? gloss[0] : 1.f;
  const float w = scal[5];
  const bool staged = P <= TH_MAXP && B * P <= TH_MAXROWS;
  if (staged) {
    for (int t = tid; t < P; t += TH_NT) disc[t] = powf(gamma, (float)t);
    for (int i = tid; i < B * P * 2; i += TH_NT) gp[i] = gpos[i];
    __syncthreads();
  }
  // d traj / d motion[b,s,c] = mstd * sum_{t>=s} gpos[b,t,c] / (B*P*2)
  for (int i = tid; i < B * 2; i += TH_NT) {
    const int b = i >> 1, c = i & 1;
    float run = 0.f;
    for (int t = P - 1; t >= 0; --t) {
      run += staged ? gp[(b * P + t) * 2 + c] : gpos[((long)b * P + t) * 2 + c];
      dout[((long)b * P + t) * C + c] = g * mstd * run / (float)(B * P * 2);
    }
  }
  if (tvis) {
    const float k = g * w / (float)((long)B * P * E);
    const int total = B * P * E;
    for (int i0 = tid; i0 < total; i0 += TH_NT * 4) {  // (see the forward kernel)
      float o[4], tv[4];
      int bt[4];
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const int i = min(i0 + TH_NT * j, total - 1);
        bt[j] = i / E;
        o[j] = out[(long)(bt[j] / P) * out_bs + (bt[j] % P) * C + 2 + (i - bt[j] * E)];
        tv[j] = tvis[(long)(bt[j] / P) * tvis_bs + (bt[j] % P) * E + (i - bt[j] * E)];
      }
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const int i = i0 + TH_NT * j;
        if (i < total) {
          const int t = bt[j] % P;
          dout[(long)bt[j] * C + 2 + (i - bt[j] * E)] = k * (staged ? disc[t] : powf(gamma, (float)t)) * sl1g(o[j] - tv[j]);
        }
      }
    }
  }
}

}  // namespace

extern "C" int rf_traj_head_fwd(const float* out, const float* last_gps, const float* target_gps,
                                const float* target_vis, float* positions, float* gpos, float* scalars, int B, int P,
                                int C, int E, float gamma, float dense_ratio, int dense_on, float motion_std,
                                float motion_mean, int64_t out_batch_stride, int64_t last_batch_stride,
                                int64_t vis_batch_stride, void* stream) {
  RF_REQUIRE(out && last_gps && target_gps && positions && gpos && scalars && B > 0 && P > 0 && C >= 2);
  RF_REQUIRE(!target_vis || C >= 2 + E);
  // batch strides (elements; 0 = packed): the head reads the LAST P rows of the decoder output, the last input position and
  // the first P rows of the target features where they lie -- no contiguous copies in front of this launch
  const long obs = out_batch_stride > 0 ? out_batch_stride : (long)P * C, lbs = last_batch_stride > 0 ? last_batch_stride : 2;
  const long vbs = vis_batch_stride > 0 ? vis_batch_stride : (long)P * E;
  RF_REQUIRE(obs >= (long)P * C && lbs >= 2 && vbs >= (long)P * E);
  RF_LAUNCH(traj_head_fwd_kernel, dim3(1), dim3(TH_NT), 0, static_cast<hipStream_t>(stream), out, last_gps,
                     target_gps, target_vis, positions, gpos, scalars, B, P, C, E, gamma, dense_ratio, dense_on,
                     motion_std, motion_mean, obs, lbs, vbs);
  RF_CHECK_LAUNCH();
  return RF_OK;
}

extern "C" int rf_traj_head_bwd(const float* out, const float* target_vis, const float* gpos, const float* scalars,
                                const float* grad_loss, float* dout, int B, int P, int C, int E, float gamma,
                                float motion_std, int64_t out_batch_stride, int64_t vis_batch_stride, void* stream) {
  RF_REQUIRE(out && gpos && scalars && dout && B > 0 && P > 0 && C >= 2);
  const long obs = out_batch_stride > 0 ? out_batch_stride : (long)P * C;
  const long vbs = vis_batch_stride > 0 ? vis_batch_stride : (long)P * E;
  RF_REQUIRE(obs >= (long)P * C && vbs >= (long)P * E);
  RF_LAUNCH(traj_head_bwd_kernel, dim3(1), dim3(TH_NT), 0, static_cast<hipStream_t>(stream), out, target_vis,
                     gpos, scalars, grad_loss, dout, B, P, C, E, gamma, motion_std, obs, vbs);
  RF_CHECK_LAUNCH();
  return RF_OK;
}

// ---------------------------------------------------------------------------------------------------
// Fusion-encoder input: out[b, s*T + t, :] = stream_s[b, t, :] + emb_s   (stream NULL = zeros: the learned
// "video output" query tokens) -- the per-stream embedding adds, the zeros_like and the cat of
// routeformer.py:331-345 in one launch; backward: emb gradients = sums over (b, t) of the matching slice.
// ---------------------------------------------------------------------------------------------------
namespace {

struct AsmP {
  const float* stream[4];
  const float* emb[4];
  float* demb[4];
  float* out;
  const float* dout;
  int B, T, E, S;
};

__global__ __launch_bounds__(256) void assemble_fwd_kernel(AsmP p) {
  const int E4 = p.E >> 2;
  const long total = (long)p.B * p.S * p.T * E4;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    const int e = (int)(i % E4) * 4;
    long r = i / E4;
    const int t = (int)(r % p.T); r /= p.T;
    const int s = (int)(r % p.S);
    const int b = (int)(r / p.S);
    float4 v = *reinterpret_cast<const float4*>(p.emb[s] + e);
    if (p.stream[s]) {
      const float4 x = *reinterpret_cast<const float4*>(p.stream[s] + ((long)b * p.T + t) * p.E + e);
      v.x += x.x; v.y += x.y; v.z += x.z; v.w += x.w;
    }
    *reinterpret_cast<float4*>(p.out + (((long)b * p.S + s) * p.T + t) * p.E + e) = v;
  }
}

// one workgroup per stream: columns x 4 row lanes (E <= 64), += into the embedding's gradient slot
__global__ __launch_bounds__(256) void assemble_bwd_kernel(AsmP p) {
  __shared__ float red[4][64];
  const int s = blockIdx.x, c = threadIdx.x & 63, rl = threadIdx.x >> 6;
  if (!p.demb[s]) return;
  float a = 0.f;
  const int BT = p.B * p.T, cc = min(c, p.E - 1);
  for (int r0 = rl; r0 < BT; r0 += 32) {  // eight rows in flight per thread
    float v[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const int r = min(r0 + 4 * j, BT - 1), b = r / p.T, t = r - b * p.T;
      v[j] = p.dout[(((long)b * p.S + s) * p.T + t) * p.E + cc];
    }
#pragma unroll
    for (int j = 0; j < 8; ++j) a += (r0 + 4 * j < BT) ? v[j] : 0.f;
  }
  red[rl][c] = a;
  __syncthreads();
  if (rl == 0 && c < p.E) p.demb[s][c] += red[0][c] + red[1][c] + red[2][c] + red[3][c];
}

}  // namespace

extern "C" int rf_assemble_streams_fwd(const float* const* streams, const float* const* embeddings, float* out, int B,
                                       int T, int E, int S, void* stream) {
  RF_REQUIRE(streams && embeddings && out && B > 0 && T > 0 && S >= 1 && S <= 4 && E % 4 == 0);
  AsmP p{};
  for (int s = 0; s < S; ++s) { p.stream[s] = streams[s]; p.emb[s] = embeddings[s]; RF_REQUIRE(embeddings[s]); }
  p.out = out; p.B = B; p.T = T; p.E = E; p.S = S;
  RF_LAUNCH(assemble_fwd_kernel, dim3(grid_for((long)B * S * T * (E / 4))), dim3(256), 0,
                     static_cast<hipStream_t>(stream), p);
  RF_CHECK_LAUNCH();
  return RF_OK;
}

extern "C" int rf_assemble_streams_bwd(const float* dout, float* const* demb, int B, int T, int E, int S, void* stream) {
  RF_REQUIRE(dout && demb && B > 0 && T > 0 && S >= 1 && S <= 4 && E <= 64);
  AsmP p{};
  for (int s = 0; s < S; ++s) p.demb[s] = demb[s];
  p.dout = dout; p.B = B; p.T = T; p.E = E; p.S = S;
  RF_LAUNCH(assemble_bwd_kernel, dim3(S), dim3(256), 0, static_cast<hipStream_t>(stream), p);
  RF_CHECK_LAUNCH();
  return RF_OK;
}

// ---- backbone input / output head (routeformer.py:279-292, 210-233) ---------------------
// x[b,t,:] = [rotate(motion, -origin) | (angle - origin)/pi | |motion| | d|motion|/dt | visual]: ~20 elementwise
// ATen launches (atan2, norm, slices, pad, cos, sin, stack, two cats) in one pass; the motion inputs carry no
// gradient, so the backward of this op is just the slice of dX that belongs to `visual`.
namespace {
__global__ void motion_input_kernel(const float* __restrict__ motion, const float* __restrict__ visual,
                                    float* __restrict__ x, float* __restrict__ origin_out, int B, int T, int E,
                                    int rotate_motion, int zero_visual) {
  const int C = 5 + E;
  const long total = (long)B * T * C;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    const int c = (int)(i % C);
    const long r = i / C;
    const int t = (int)(r % T), b = (int)(r / T);
    if (c >= 5) {
      x[i] = zero_visual ? 0.f : visual[r * E + (c - 5)];
      continue;
    }
    const float* mb = motion + (long)b * T * 2;
    const int to = rotate_motion ? T - 1 : 0;
    const float origin = atan2f(mb[2 * to + 1], mb[2 * to]);
    if (t == 0 && c == 0) origin_out[b] = origin;
    const float mx = mb[2 * t], my = mb[2 * t + 1];
    float v;
    if (c < 2) {
      if (rotate_motion) {  // R(-origin) [mx, my]
        const float cs = cosf(-origin), sn = sinf(-origin);
        v = c == 0 ? cs * mx - sn * my : sn * mx + cs * my;
      } else {
        v = c == 0 ? mx : my;
      }
    } else if (c == 2) {
      v = (atan2f(my, mx) - origin) / 3.14159265358979323846f;
    } else {
      const float n = sqrtf(mx * mx + my * my);
      if (c == 3) v = n;
      else v = t > 0 ? n - sqrtf(mb[2 * t - 2] * mb[2 * t - 2] + mb[2 * t - 1] * mb[2 * t - 1]) : 0.f;
    }
    x[i] = v;
  }
}

// y = out with channels 0,1 rotated by sign * origin[b] (R = [[c,-s],[s,c]]); sign = +1 forward, -1 backward
__global__ void rotate_head_kernel(const float* __restrict__ in, const float* __restrict__ origin,
                                   float* __restrict__ out, int B, int P, int C, float sign) {
  const long total = (long)B * P * C;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    const int c = (int)(i % C);
    const long r = i / C;
    if (c >= 2) { out[i] = in[i]; continue; }
    const int b = (int)(r / P);
    const float a = sign * origin[b];
    const float cs = cosf(a), sn = sinf(a);
    const float vx = in[r * C], vy = in[r * C + 1];
    out[i] = c == 0 ? cs * vx - sn * vy : sn * vx + cs * vy;
  }
}
}  // namespace

extern "C" int rf_motion_input(const float* motion, const float* visual, float* x, float* origin, int B, int T, int E,
                               int rotate_motion, int zero_visual, void* stream) {
  RF_REQUIRE(motion && x && origin && B > 0 && T > 0 && E >= 0 && (E == 0 || visual || zero_visual));
  RF_LAUNCH(motion_input_kernel, dim3(grid_for((long)B * T * (5 + E))), dim3(256), 0, static_cast<hipStream_t>(stream),
            motion, visual, x, origin, B, T, E, rotate_motion, zero_visual);
  RF_CHECK_LAUNCH();
  return RF_OK;
}

extern "C" int rf_rotate_head(const float* in, const float* origin, float* out, int B, int P, int C, float sign,
                              void* stream) {
  RF_REQUIRE(in && origin && out && B > 0 && P > 0 && C >= 2 && (sign == 1.f || sign == -1.f));
  RF_LAUNCH(rotate_head_kernel, dim3(grid_for((long)B * P * C)), dim3(256), 0, static_cast<hipStream_t>(stream), in,
            origin, out, B, P, C, sign);
  RF_CHECK_LAUNCH();
  return RF_OK;
}
