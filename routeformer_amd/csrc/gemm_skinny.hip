// Skinny GEMM for the GPS backbone's linear layers at training batch sizes (SURVEY 8(b) `informer_linear`):
//     C[M, N] = epilogue(A[M, K] W^T)          M = B x L_layer = 32 .. 560 rows,  N, K = 832 .. 3328
// (Informer encoder with distilling at B = 8: 320, 168, 96, 56, 40, 32 rows; layers/SelfAttention_Family.py:146-148,
// layers/TransformerEncoderDecoder.py:36-37,50-51,98-99).  At these shapes the product is a STREAM OF THE WEIGHTS
// (2.8 - 11 MB fp32 per launch, read once) against an activation that fits in L2 -- HBM-bound, and in practice
// latency-bound: the tiled kernel (gemm.hip) walks K in 64-wide steps with two LDS stages per workgroup, i.e. a chain of
// 3 - 13 dependent HBM round trips, then a second launch sums its split-K slabs.  Here instead:
//   * one workgroup (8 waves) owns a 16 RTM x 16 TN output tile over a K-slice of <= 1024; the 8 waves INTERLEAVE the
//     32-wide k-steps of the slice (wave w: steps w, w + 8, ...; at most 4 each) and every wave requests ALL of its
//     operands before the first MFMA -- one HBM round trip per launch, >= 400 waves in flight for N = 832;
//   * operands go global -> registers -> MFMA: the 16x16x32 bf16 fragment of a k-contiguous matrix is 32 contiguous
//     bytes per lane (two 16-B loads, fp32 -> bf16 in registers), no LDS staging, no barrier in the product;
//   * the 8 partial tiles meet in LDS, the epilogue (bias / residual / activation / activation') runs in the launch:
//     no split-K slabs and no second launch up to K = 1024; longer K uses blockIdx.z slices + a slab sum.
// Weight orientations: BMODE 0 = k contiguous (forward: y = x W^T), BMODE 1 = n contiguous (dX = dY W: the contraction
// index is W's row, 8 strided 4-B loads per fragment, each instruction still 4 x 64-B segments).
// bf16-input MFMA with fp32 accumulation: the same rounding as gemm.hip's PREC = 1 path (sum order differs).
#include <cstdlib>

#include "common.h"

namespace {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

constexpr int SK_NW = 8, SK_NT = 64 * SK_NW, SK_MAXS = 4, SK_KSLICE = 32 * SK_NW * SK_MAXS;  // 1024

struct SkinnyP {
  const float* A; long lda;   // A[m * lda + k]
  const float* B; long ldb;   // BMODE 0: B[n * ldb + k];  BMODE 1: B[k * ldb + n]
  float* C; long ldc;
  int M, N, K;
  const float* bias;
  const float* res; long ldr; int res_rows, res_before_act;
  int act;
  float* preact; long ldp;
  const float* dsrc; long ldd; int dact;
  int kslice;  // K range per blockIdx.z (multiple of 32)
  float* ws;   // gridDim.z > 1: partial sums [z][M][N]
};

__device__ __forceinline__ bf16x8 sk_pack(const float4& a, const float4& b) {
  bf16x8 o;
  o[0] = (__bf16)a.x; o[1] = (__bf16)a.y; o[2] = (__bf16)a.z; o[3] = (__bf16)a.w;
  o[4] = (__bf16)b.x; o[5] = (__bf16)b.y; o[6] = (__bf16)b.z; o[7] = (__bf16)b.w;
  return o;
}

__device__ __forceinline__ void sk_epilogue(const SkinnyP& p, int m, int n, float v) {
  if (p.bias) v += p.bias[n];
  if (p.res && p.res_before_act) v += p.res[(long)(m % p.res_rows) * p.ldr + n];
  if (p.preact) p.preact[(long)m * p.ldp + n] = v;
  v = apply_act(v, p.act);
  if (p.dact) v *= act_grad(p.dsrc[(long)m * p.ldd + n], p.dact);
  if (p.res && !p.res_before_act) v += p.res[(long)(m % p.res_rows) * p.ldr + n];
  p.C[(long)m * p.ldc + n] = v;
}

template <int RTM, int TN, int BMODE>
__global__ __launch_bounds__(SK_NT) void gemm_skinny_kernel(const SkinnyP p) {
  __shared__ float red[SK_NW][RTM * TN][256];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, fr = lane & 15, fq = lane >> 4;
  const int n0 = blockIdx.x * 16 * TN, m0 = blockIdx.y * 16 * RTM;
  const int kbeg = blockIdx.z * p.kslice, kend = min(p.K, kbeg + p.kslice);

  // ---- request every operand of this wave's k-steps (K % 8 == 0: a lane's 8-wide k group is in range or not) ----
  const float4 zero4 = make_float4(0.f, 0.f, 0.f, 0.f);
  float4 araw[SK_MAXS][RTM][2];
  float4 braw[SK_MAXS][TN][2];
  const float* arow[RTM];
#pragma unroll
  for (int rt = 0; rt < RTM; ++rt) arow[rt] = p.A + (long)min(m0 + rt * 16 + fr, p.M - 1) * p.lda;
#pragma unroll
  for (int i = 0; i < SK_MAXS; ++i) {
    const int k = kbeg + (wave + SK_NW * i) * 32 + fq * 8;
    const bool live = k < kend;
#pragma unroll
    for (int tn = 0; tn < TN; ++tn) {
      braw[i][tn][0] = braw[i][tn][1] = zero4;
      if (live) {
        const int n = min(n0 + tn * 16 + fr, p.N - 1);
        if constexpr (BMODE == 0) {
          const float* bp = p.B + (long)n * p.ldb + k;
          braw[i][tn][0] = *reinterpret_cast<const float4*>(bp);
          braw[i][tn][1] = *reinterpret_cast<const float4*>(bp + 4);
        } else {
          const float* bp = p.B + (long)k * p.ldb + n;
          braw[i][tn][0] = make_float4(bp[0], bp[p.ldb], bp[2 * p.ldb], bp[3 * p.ldb]);
          braw[i][tn][1] = make_float4(bp[4 * p.ldb], bp[5 * p.ldb], bp[6 * p.ldb], bp[7 * p.ldb]);
        }
      }
    }
#pragma unroll
    for (int rt = 0; rt < RTM; ++rt) {
      araw[i][rt][0] = araw[i][rt][1] = zero4;
      if (live) {
        araw[i][rt][0] = *reinterpret_cast<const float4*>(arow[rt] + k);
        araw[i][rt][1] = *reinterpret_cast<const float4*>(arow[rt] + k + 4);
      }
    }
  }
  // ---- products ----
  f32x4 acc[RTM][TN];
#pragma unroll
  for (int rt = 0; rt < RTM; ++rt)
#pragma unroll
    for (int tn = 0; tn < TN; ++tn) acc[rt][tn] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int i = 0; i < SK_MAXS; ++i) {
    bf16x8 bf[TN];
#pragma unroll
    for (int tn = 0; tn < TN; ++tn) bf[tn] = sk_pack(braw[i][tn][0], braw[i][tn][1]);
#pragma unroll
    for (int rt = 0; rt < RTM; ++rt) {
      const bf16x8 af = sk_pack(araw[i][rt][0], araw[i][rt][1]);
#pragma unroll
      for (int tn = 0; tn < TN; ++tn) acc[rt][tn] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af, bf[tn], acc[rt][tn], 0, 0, 0);
    }
  }
  // ---- the 8 partial tiles meet in LDS ----
#pragma unroll
  for (int rt = 0; rt < RTM; ++rt)
#pragma unroll
    for (int tn = 0; tn < TN; ++tn)
#pragma unroll
      for (int r = 0; r < 4; ++r) red[wave][rt * TN + tn][(fq * 4 + r) * 16 + fr] = acc[rt][tn][r];
  __syncthreads();
  for (int o = tid; o < RTM * TN * 256; o += SK_NT) {
    const int t = o >> 8, e = o & 255;
    float v = 0.f;
#pragma unroll
    for (int w = 0; w < SK_NW; ++w) v += red[w][t][e];
    const int m = m0 + (t / TN) * 16 + (e >> 4), n = n0 + (t % TN) * 16 + (e & 15);
    if (m < p.M && n < p.N) {
      if (p.ws) p.ws[((long)blockIdx.z * p.M + m) * p.N + n] = v;
      else sk_epilogue(p, m, n, v);
    }
  }
}

// Slab sum + epilogue (K > 1024): as gemm.hip's splitk_reduce_kernel -- four slabs in flight per trip, summed in ascending
// order, the epilogue's operands requested up front through branch-free selects.  M <= 640, N <= a few thousand: 32-bit.
__global__ __launch_bounds__(256) void skinny_reduce_kernel(SkinnyP p, int splits) {
  const unsigned total = (unsigned)p.M * (unsigned)p.N;
  const float* dummy = p.ws;
  for (unsigned i = blockIdx.x * blockDim.x + threadIdx.x; i < total; i += gridDim.x * blockDim.x) {
    const unsigned m = i / (unsigned)p.N, n = i - m * (unsigned)p.N;
    const float bv = *(p.bias ? p.bias + n : dummy);
    const float rv = *(p.res ? p.res + (long)(m % (unsigned)p.res_rows) * p.ldr + n : dummy);
    const float dv = *(p.dact ? p.dsrc + (long)m * p.ldd + n : dummy);
    float v = 0.f;
    for (int s0 = 0; s0 < splits; s0 += 4) {
      float t[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) t[u] = p.ws[(size_t)min(s0 + u, splits - 1) * total + i];
#pragma unroll
      for (int u = 0; u < 4; ++u) v += (s0 + u < splits) ? t[u] : 0.f;
    }
    if (p.bias) v += bv;
    if (p.res && p.res_before_act) v += rv;
    if (p.preact) p.preact[(long)m * p.ldp + n] = v;
    v = apply_act(v, p.act);
    if (p.dact) v *= act_grad(dv, p.dact);
    if (p.res && !p.res_before_act) v += rv;
    p.C[(long)m * p.ldc + n] = v;
  }
}

inline bool sk_al16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; }

// Row tiles per workgroup / column tiles per wave: small M keeps RTM at the row count; more rows widen the tile in n
// so that the activation is re-read by fewer workgroups.
struct SkShape { int rtm, tn; };
inline SkShape sk_shape(int M, int N) {
  SkShape s;
  s.rtm = M <= 16 ? 1 : (M <= 32 ? 2 : 4);
  // RF_SKINNY_TN2_MIN_N: output width from which a workgroup takes two 16-column tiles (measurement switch)
  static const int tn2_min_n = [] { const char* e = getenv("RF_SKINNY_TN2_MIN_N"); return e ? atoi(e) : 1024; }();
  s.tn = (M > 64 && N >= tn2_min_n) ? 2 : 1;
  return s;
}

template <int BMODE>
void sk_launch(const SkinnyP& p, int z, hipStream_t st) {
  const SkShape s = sk_shape(p.M, p.N);
  const dim3 grid((p.N + 16 * s.tn - 1) / (16 * s.tn), (p.M + 16 * s.rtm - 1) / (16 * s.rtm), z);
#define RF_SK_GO(RTM_, TN_) RF_LAUNCH((gemm_skinny_kernel<RTM_, TN_, BMODE>), grid, dim3(SK_NT), 0, st, p)
  if (s.rtm == 1) RF_SK_GO(1, 1);
  else if (s.rtm == 2) RF_SK_GO(2, 1);
  else if (s.tn == 1) RF_SK_GO(4, 1);
  else RF_SK_GO(4, 2);
#undef RF_SK_GO
}

}  // namespace

extern "C" int rf_gemm_skinny_split(const float* A, int64_t lda_m, int64_t lda_k, const float* B, int64_t ldb_k, int64_t ldb_n,
                                    int M, int N, int K) {
  if (!A || !B || M <= 0 || N < 16 || K < 64 || K % 8 != 0 || M > 640) return 0;
  if (lda_k != 1 || lda_m % 4 != 0 || !sk_al16(A)) return 0;
  const bool b0 = ldb_k == 1 && ldb_n % 4 == 0 && sk_al16(B), b1 = ldb_n == 1;
  if (!b0 && !b1) return 0;
  return (K + SK_KSLICE - 1) / SK_KSLICE;
}

static int skinny_run(const float* A, int64_t lda_m, int64_t lda_k, const float* B, int64_t ldb_k, int64_t ldb_n,
                      float* C, int64_t ldc, int M, int N, int K, const float* bias, const float* residual,
                      int64_t ldr, int res_rows, int res_before_act, int act, float* preact, int64_t ldp,
                      const float* dact_src, int64_t ldd, int dact_mode, float* workspace, bool partials_only,
                      void* stream) {
  RF_REQUIRE(C);
  const int z = rf_gemm_skinny_split(A, lda_m, lda_k, B, ldb_k, ldb_n, M, N, K);
  if (z == 0) {
    rf_g_last_error = "rf_gemm_skinny: shape / layout outside the skinny kernel's range (see rf_gemm_skinny_split)";
    return RF_EUNSUPPORTED;
  }
  RF_REQUIRE(z == 1 || workspace);
  RF_REQUIRE(!residual || res_rows > 0);
  RF_REQUIRE(!dact_mode || dact_src);
  SkinnyP p{};
  p.A = A; p.lda = lda_m; p.B = B;
  const bool b0 = ldb_k == 1 && ldb_n % 4 == 0 && sk_al16(B);
  p.ldb = b0 ? ldb_n : ldb_k;
  p.C = C; p.ldc = ldc; p.M = M; p.N = N; p.K = K; p.bias = bias; p.res = residual; p.ldr = ldr;
  p.res_rows = residual ? res_rows : 1; p.res_before_act = res_before_act; p.act = act;
  p.preact = preact; p.ldp = ldp; p.dsrc = dact_src; p.ldd = ldd; p.dact = dact_mode;
  p.kslice = z == 1 ? ((K + 31) / 32) * 32 : SK_KSLICE;
  p.ws = (z > 1 || partials_only) ? workspace : nullptr;
  const hipStream_t st = static_cast<hipStream_t>(stream);
  if (b0) sk_launch<0>(p, z, st);
  else sk_launch<1>(p, z, st);
  RF_CHECK_LAUNCH();
  if (z > 1 && !partials_only) {
    const long total = (long)M * N;
    int blocks = (int)((total + 255) / 256);
    if (blocks > 2048) blocks = 2048;
    RF_LAUNCH(skinny_reduce_kernel, dim3(blocks), dim3(256), 0, st, p, z);
    RF_CHECK_LAUNCH();
  }
  return RF_OK;
}

extern "C" int rf_gemm_skinny(const float* A, int64_t lda_m, int64_t lda_k, const float* B, int64_t ldb_k, int64_t ldb_n,
                              float* C, int64_t ldc, int M, int N, int K, const float* bias, const float* residual,
                              int64_t ldr, int res_rows, int res_before_act, int act, float* preact, int64_t ldp,
                              const float* dact_src, int64_t ldd, int dact_mode, float* workspace, void* stream) {
  return skinny_run(A, lda_m, lda_k, B, ldb_k, ldb_n, C, ldc, M, N, K, bias, residual, ldr, res_rows, res_before_act, act,
                    preact, ldp, dact_src, ldd, dact_mode, workspace, false, stream);
}

// The K slices' raw products, workspace[slice][M][N] with rf_gemm_skinny_split(...) slices: no epilogue, no slab sum.
extern "C" int rf_gemm_skinny_partials(const float* A, int64_t lda_m, int64_t lda_k, const float* B, int64_t ldb_k,
                                       int64_t ldb_n, int M, int N, int K, float* workspace, void* stream) {
  RF_REQUIRE(workspace);
  return skinny_run(A, lda_m, lda_k, B, ldb_k, ldb_n, workspace, N, M, N, K, nullptr, nullptr, 0, 0, 0, 0, nullptr, 0,
                    nullptr, 0, 0, workspace, true, stream);
}
