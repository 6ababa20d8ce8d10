// Small tensor plumbing of the hot path as single launches (each replaces 2-6 element-wise / copy / sort launches of the
// framework that a replayed step would otherwise carry):
//   rf_median_windows   `median_downsampler` (utils/filter.py:5-43): lower median of consecutive windows (torch.median
//                       semantics incl. NaN propagation) -- was a sort + gather chain per gaze track;
//   rf_motion_diff      motion = pad(normalise(gps[t] - gps[t-1]), one zero row in front) (routeformer.py:284-292);
//   rf_time_table       DataEmbedding's rank-1 time feature + positional table: out[l, c] = l * w[c] + pe[l, c]
//                       (layers/Embedding.py:99-126 with the position index as the one "timeF" feature), and its backward
//                       dw[c] (+)= sum_l l * dout[l, c];
//   rf_timeline_scatter / rf_timeline_gather   features of the sub-sampled frames into a zero timeline and back
//                       (routeformer.py:443-459: `full[:, idx] = feats`);
//   rf_smart_tail_fwd / _bwd   the "smart" decoder input cat([x, last row of x repeated pred_len times]) (Informer.py:125-136)
//                       and its gradient (the repeated row collects the tail's gradients).
// All latency-bound; one thread per output element, channels innermost.
#include <math.h>

#include "common.h"

namespace {

inline int blocks_for(long total, int block = 256, int cap = 4096) {
  long g = (total + block - 1) / block;
  return (int)(g > cap ? cap : (g < 1 ? 1 : g));
}

// One 64-thread workgroup per (b, window, c): the window in LDS, each thread ranks its elements against the window
// (rank = #smaller + #equal-with-lower-index); the element of rank (w - 1) / 2 is torch.median's lower median.
constexpr int MED_MAXW = 1024;
__global__ __launch_bounds__(64) void median_windows_kernel(const float* __restrict__ x, float* __restrict__ y, int B, int T,
                                                            int C, int target, int w) {
  __shared__ float win[MED_MAXW];
  __shared__ int has_nan;
  const long o = blockIdx.x;  // (b * target + i) * C + c
  const int c = (int)(o % C);
  const long r = o / C;
  const int i = (int)(r % target), b = (int)(r / target);
  if (threadIdx.x == 0) has_nan = 0;
  __syncthreads();
  const float* src = x + ((long)b * T + (long)i * w) * C + c;
  for (int j = threadIdx.x; j < w; j += 64) {
    const float v = src[(long)j * C];
    win[j] = v;
    if (v != v) has_nan = 1;
  }
  __syncthreads();
  if (has_nan) {  // torch.median: NaN is the largest value and wins whenever the window holds one
    if (threadIdx.x == 0) y[o] = NAN;
    return;
  }
  const int k = (w - 1) / 2;
  for (int j = threadIdx.x; j < w; j += 64) {
    const float v = win[j];
    int rank = 0;
    for (int q = 0; q < w; ++q) {
      const float u = win[q];
      rank += (u < v) || (u == v && q < j);
    }
    if (rank == k) y[o] = v;  // exactly one element has rank k
  }
}

__global__ void motion_diff_kernel(const float* __restrict__ gps, float* __restrict__ motion, int B, int T, int normalize,
                                   float mean, float inv_std) {
  const long total = (long)B * T * 2;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    const int t = (int)((i >> 1) % T);
    float v = 0.f;
    if (t > 0) {
      v = gps[i] - gps[i - 2];
      if (normalize) v = (v - mean) * inv_std;
    }
    motion[i] = v;
  }
}

__global__ void time_table_kernel(const float* __restrict__ w, const float* __restrict__ pe, float* __restrict__ out, int L,
                                  int d) {
  const long total = (long)L * d;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    const int c = (int)(i % d);
    const int l = (int)(i / d);
    out[i] = (float)l * w[c] + pe[i];
  }
}

// dw[c] (+)= sum_l l * dout[l, c]: 64 columns x 4 row lanes per workgroup, eight rows in flight per thread, fixed
// summation order (replicas stay bit-identical)
__global__ __launch_bounds__(256) void time_table_bwd_kernel(const float* __restrict__ dout, float* __restrict__ dw, int L, int d,
                                                             int accumulate) {
  __shared__ float red[4][64];
  const int tx = threadIdx.x & 63, rl = threadIdx.x >> 6;
  const int c = blockIdx.x * 64 + tx, cc = min(c, d - 1);
  float s = 0.f;
  for (int l0 = rl; l0 < L; l0 += 32) {
    float v[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) v[j] = dout[(long)min(l0 + 4 * j, L - 1) * d + cc];
#pragma unroll
    for (int j = 0; j < 8; ++j) s = (l0 + 4 * j < L) ? fmaf((float)(l0 + 4 * j), v[j], s) : s;
  }
  red[rl][tx] = s;
  __syncthreads();
  if (rl == 0 && c < d) {
    const float t = (red[0][tx] + red[1][tx]) + (red[2][tx] + red[3][tx]);
    dw[c] = accumulate ? dw[c] + t : t;
  }
}

// out[n, t, :] = feats[n, f, :] if t == idx[f] else 0      (n = stream * batch + b)
__global__ void timeline_scatter_kernel(const float* __restrict__ feats, const int64_t* __restrict__ idx, float* __restrict__ out,
                                        long N, int T, int F, int E) {
  const long total = N * T * E;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    const int e = (int)(i % E);
    const long r = i / E;
    const int t = (int)(r % T);
    const long n = r / T;
    float v = 0.f;
    for (int f = 0; f < F; ++f)
      if ((int)idx[f] == t) v = feats[(n * F + f) * E + e];  // (duplicate indices: the last one wins, as index_put_ does)
    out[i] = v;
  }
}

// dfeats[n, f, :] = dout[n, idx[f], :]
__global__ void timeline_gather_kernel(const float* __restrict__ dout, const int64_t* __restrict__ idx, float* __restrict__ dfeats,
                                       long N, int T, int F, int E) {
  const long total = N * F * E;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    const int e = (int)(i % E);
    const long r = i / E;
    const int f = (int)(r % F);
    const long n = r / F;
    dfeats[i] = dout[(n * T + (int)idx[f]) * E + e];
  }
}

// y (B, L + P, C): rows 0..L-1 = x, rows L.. = x[:, L-1] (smart) or 0
__global__ void smart_tail_fwd_kernel(const float* __restrict__ x, float* __restrict__ y, int B, int L, int P, int C, int smart) {
  const long total = (long)B * (L + P) * C;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    const int c = (int)(i % C);
    const long r = i / C;
    const int t = (int)(r % (L + P));
    const long b = r / (L + P);
    float v = 0.f;
    if (t < L) v = x[(b * L + t) * C + c];
    else if (smart) v = x[(b * L + L - 1) * C + c];
    y[i] = v;
  }
}

// dx (B, L, C) = dy[:, :L] (+ extra, another gradient of x), the last row also collecting the tail's gradients (smart)
__global__ void smart_tail_bwd_kernel(const float* __restrict__ dy, const float* __restrict__ extra, float* __restrict__ dx, int B,
                                      int L, int P, int C, int smart) {
  const long total = (long)B * L * C;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    const int c = (int)(i % C);
    const long r = i / C;
    const int t = (int)(r % L);
    const long b = r / L;
    float v = dy[(b * (L + P) + t) * C + c];
    if (smart && t == L - 1) {
      const float* tail = dy + (b * (L + P) + L) * C + c;
      for (int p0 = 0; p0 < P; p0 += 8) {  // eight tail rows in flight (was one dependent load per row: 15 us at P = 30)
        float u[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) u[j] = tail[(long)min(p0 + j, P - 1) * C];
#pragma unroll
        for (int j = 0; j < 8; ++j) v += (p0 + j < P) ? u[j] : 0.f;
      }
    }
    if (extra) v += extra[i];
    dx[i] = v;
  }
}

// dst (rows, ld) = [src (rows, cols) | zeros]: the K-padded copy of a weight whose row length is not a multiple of 4 (the GPS
// token embedding's 3 x 69 = 207 -> 208: kernels.circular_conv3), and the way back for its gradient
__global__ __launch_bounds__(256) void pad_cols_kernel(const float* __restrict__ src, float* __restrict__ dst, int rows, int cols,
                                                       int ld) {
  const unsigned total = (unsigned)rows * (unsigned)ld;
  for (unsigned i = blockIdx.x * blockDim.x + threadIdx.x; i < total; i += gridDim.x * blockDim.x) {
    const unsigned r = i / (unsigned)ld, c = i - r * (unsigned)ld;
    const float v = src[(size_t)r * cols + min(c, (unsigned)cols - 1)];  // clamped, unconditional load
    dst[i] = c < (unsigned)cols ? v : 0.f;
  }
}
__global__ __launch_bounds__(256) void unpad_cols_kernel(const float* __restrict__ src, float* __restrict__ dst, int rows, int cols,
                                                         int ld, int accumulate) {
  const unsigned total = (unsigned)rows * (unsigned)cols;
  for (unsigned i = blockIdx.x * blockDim.x + threadIdx.x; i < total; i += gridDim.x * blockDim.x) {
    const unsigned r = i / (unsigned)cols, c = i - r * (unsigned)cols;
    const float v = src[(size_t)r * ld + c];
    dst[i] = accumulate ? dst[i] + v : v;
  }
}

}  // namespace

extern "C" int rf_median_windows(const float* x, float* y, int B, int T, int C, int target, void* stream) {
  RF_REQUIRE(x && y && B > 0 && T > 0 && C > 0 && target > 0 && target < T);
  const int w = T / target;
  RF_REQUIRE(w >= 1 && w <= MED_MAXW);
  const long outs = (long)B * target * C;
  RF_REQUIRE(outs < (1L << 31));
  RF_LAUNCH(median_windows_kernel, dim3((unsigned)outs), dim3(64), 0, static_cast<hipStream_t>(stream), x, y, B, T, C, target, w);
  RF_CHECK_LAUNCH();
  return RF_OK;
}

extern "C" int rf_motion_diff(const float* gps, float* motion, int B, int T, int normalize, float mean, float std_, void* stream) {
  RF_REQUIRE(gps && motion && B > 0 && T > 0 && (!normalize || std_ != 0.f));
  RF_LAUNCH(motion_diff_kernel, dim3(blocks_for((long)B * T * 2)), dim3(256), 0, static_cast<hipStream_t>(stream), gps, motion, B,
            T, normalize, mean, normalize ? 1.0f / std_ : 1.0f);
  RF_CHECK_LAUNCH();
  return RF_OK;
}

extern "C" int rf_time_table(const float* w, const float* pe, float* out, int L, int d, void* stream) {
  RF_REQUIRE(w && pe && out && L > 0 && d > 0);
  RF_LAUNCH(time_table_kernel, dim3(blocks_for((long)L * d)), dim3(256), 0, static_cast<hipStream_t>(stream), w, pe, out, L, d);
  RF_CHECK_LAUNCH();
  return RF_OK;
}

extern "C" int rf_time_table_bwd(const float* dout, float* dw, int L, int d, int accumulate, void* stream) {
  RF_REQUIRE(dout && dw && L > 0 && d > 0);
  RF_LAUNCH(time_table_bwd_kernel, dim3((d + 63) / 64), dim3(256), 0, static_cast<hipStream_t>(stream), dout, dw, L, d,
            accumulate);
  RF_CHECK_LAUNCH();
  return RF_OK;
}

extern "C" int rf_timeline_scatter(const float* feats, const int64_t* idx, float* out, int64_t N, int T, int F, int E,
                                   void* stream) {
  RF_REQUIRE(feats && idx && out && N > 0 && T > 0 && F > 0 && F <= T && E > 0);
  RF_LAUNCH(timeline_scatter_kernel, dim3(blocks_for(N * T * E)), dim3(256), 0, static_cast<hipStream_t>(stream), feats, idx, out,
            (long)N, T, F, E);
  RF_CHECK_LAUNCH();
  return RF_OK;
}

extern "C" int rf_timeline_gather(const float* dout, const int64_t* idx, float* dfeats, int64_t N, int T, int F, int E,
                                  void* stream) {
  RF_REQUIRE(dout && idx && dfeats && N > 0 && T > 0 && F > 0 && F <= T && E > 0);
  RF_LAUNCH(timeline_gather_kernel, dim3(blocks_for(N * F * E)), dim3(256), 0, static_cast<hipStream_t>(stream), dout, idx, dfeats,
            (long)N, T, F, E);
  RF_CHECK_LAUNCH();
  return RF_OK;
}

extern "C" int rf_smart_tail_fwd(const float* x, float* y, int B, int L, int P, int C, int smart, void* stream) {
  RF_REQUIRE(x && y && B > 0 && L > 0 && P >= 0 && C > 0);
  RF_LAUNCH(smart_tail_fwd_kernel, dim3(blocks_for((long)B * (L + P) * C)), dim3(256), 0, static_cast<hipStream_t>(stream), x, y,
            B, L, P, C, smart);
  RF_CHECK_LAUNCH();
  return RF_OK;
}

extern "C" int rf_smart_tail_bwd(const float* dy, const float* extra, float* dx, int B, int L, int P, int C, int smart,
                                 void* stream) {
  RF_REQUIRE(dy && dx && B > 0 && L > 0 && P >= 0 && C > 0);
  RF_LAUNCH(smart_tail_bwd_kernel, dim3(blocks_for((long)B * L * C)), dim3(256), 0, static_cast<hipStream_t>(stream), dy, extra,
            dx, B, L, P, C, smart);
  RF_CHECK_LAUNCH();
  return RF_OK;
}


extern "C" int rf_pad_cols(const float* src, float* dst, int rows, int cols, int ld, void* stream) {
  RF_REQUIRE(src && dst && rows > 0 && cols > 0 && ld >= cols && (long)rows * ld < (1L << 31));
  RF_LAUNCH(pad_cols_kernel, dim3(blocks_for((long)rows * ld)), dim3(256), 0, static_cast<hipStream_t>(stream), src, dst, rows,
            cols, ld);
  RF_CHECK_LAUNCH();
  return RF_OK;
}

extern "C" int rf_unpad_cols(const float* src, float* dst, int rows, int cols, int ld, int accumulate, void* stream) {
  RF_REQUIRE(src && dst && rows > 0 && cols > 0 && ld >= cols && (long)rows * ld < (1L << 31));
  RF_LAUNCH(unpad_cols_kernel, dim3(blocks_for((long)rows * cols)), dim3(256), 0, static_cast<hipStream_t>(stream), src, dst,
            rows, cols, ld, accumulate);
  RF_CHECK_LAUNCH();
  return RF_OK;
}
