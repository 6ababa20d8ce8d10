// Dense contraction on the CDNA4 matrix cores, with the fused epilogues the Routeformer hot path
// needs, and the implicit-GEMM (im2col-on-the-fly) A-loader used for the frozen NHWC conv trunk.
//
//   C[M,N] = epi( A[M,K] * B[K,N] )
//
// One workgroup = 4 waves (256 threads).  Operands are staged global -> registers -> LDS with K
// innermost ([row][k], padded), so both MFMA operands are read as K-contiguous fragments:
//   prec 0: v_mfma_f32_16x16x4_f32   (exact fp32; lane l holds A[l&15][l>>4], B[l>>4][l&15])
//   prec 1: v_mfma_f32_16x16x32_bf16 (lane l holds A[l&15][8*(l>>4)+j], j<8; fp32 -> bf16 on the
//           way into LDS, fp32 accumulate)
// C/D fragment: col = lane&15, row = 4*(lane>>4)+reg.
// Tile configs (BM x BN, BK = 32): 0 = 64x64 (2x2 waves, 2x2 tiles), 1 = 128x16 (4x1 waves, 2x1),
// 2 = 128x32 (4x1 waves, 2x2) -- the narrow ones serve the 16/32-channel HRNet branches.
#include "common.h"
#include <cstdlib>

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

namespace {

constexpr int BK = 32;
constexpr int NT = 256;

struct GemmP {
  const float* A; long lda_m, lda_k;
  const float* B; long ldb_k, ldb_n;
  float* C; long ldc;
  int M, N, K;
  const float* bias;
  const float* res; long ldr; int res_rows; int res_before_act;
  int act;
  float* preact; long ldp;
  const float* dsrc; long ldd; int dact;
  int kchunk;  // K range per split-K slice (multiple of BK)
  float* ws;   // split-K partials [splits][M][N] (null: single pass with epilogue)
  int atomic;  // 1: C += partial via fp32 atomics (no other epilogue); 2: C = partial, plain stores (exclusive tile)
  unsigned* tile_cnt;  // split-K arrival counters (one per output tile, zero between launches) or null
  int gx, gy, share;   // tile grid and XCD grouping of the launch (gemm2_kernel): 0 = plain, 1 = column tiles, 2 = row tiles
  float* a_rowsum;  // optional: a_rowsum[m] += sum_k A[m,k] (A row-contiguous)
  // implicit-GEMM conv (A mode 3): A is the NHWC input, row m = output pixel
  int cH, cW, cCin, cKs, cStride, cPad, cHo, cWo;
  int act_bf16;  // implicit-GEMM conv only: A, res and C are bf16 maps (rf_conv2d_nhwc, RF_ACT_BF16)
};

template <int CFG> struct Cfg;
template <> struct Cfg<0> { static constexpr int WM = 2, WN = 2, TM = 2, TN = 2; };
template <> struct Cfg<1> { static constexpr int WM = 4, WN = 1, TM = 2, TN = 1; };
template <> struct Cfg<2> { static constexpr int WM = 4, WN = 1, TM = 2, TN = 2; };

template <int PREC> struct Lds;
template <> struct Lds<0> { using T = float; static constexpr int LD = BK + 2; };   // 34: conflict-free b32 reads
template <> struct Lds<1> { using T = __bf16; static constexpr int LD = BK + 8; };  // 80-B rows, 16-B aligned

template <typename T> __device__ __forceinline__ T cvt(float v);
template <> __device__ __forceinline__ float cvt<float>(float v) { return v; }
template <> __device__ __forceinline__ __bf16 cvt<__bf16>(float v) { return (__bf16)v; }

// ---- global -> LDS tile loaders.  Tile is ROWS x BK, stored [row][k]. ----
// MODE 0: k contiguous in memory (ld_k == 1), 16-B vectorizable.
// MODE 1: row index contiguous in memory (ld_row == 1), 16-B vectorizable (transposing store).
// MODE 2: arbitrary strides / alignment, scalar.
template <int ROWS, int MODE, typename T, int LD>
__device__ __forceinline__ void load_tile(T* __restrict__ S, const float* __restrict__ G, long ld_row,
                                          long ld_k, int row0, int nrows, int k0, int kend, int tid) {
  if constexpr (MODE == 0) {
    constexpr int VPR = BK / 4;
#pragma unroll
    for (int i = tid; i < ROWS * VPR; i += NT) {
      const int r = i / VPR, kv = (i % VPR) * 4;
      const int gr = row0 + r, gk = k0 + kv;
      float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
      if (gr < nrows) {
        const float* p = G + (long)gr * ld_row + gk;
        if (gk + 3 < kend) {
          v = *reinterpret_cast<const float4*>(p);
        } else {
          if (gk + 0 < kend) v.x = p[0];
          if (gk + 1 < kend) v.y = p[1];
          if (gk + 2 < kend) v.z = p[2];
        }
      }
      T* s = S + r * LD + kv;
      s[0] = cvt<T>(v.x); s[1] = cvt<T>(v.y); s[2] = cvt<T>(v.z); s[3] = cvt<T>(v.w);
    }
  } else if constexpr (MODE == 1) {
    constexpr int VPK = ROWS / 4;  // row-vectors per k
#pragma unroll
    for (int i = tid; i < BK * VPK; i += NT) {
      const int k = i / VPK, rv = (i % VPK) * 4;
      const int gr = row0 + rv, gk = k0 + k;
      float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
      if (gk < kend) {
        const float* p = G + (long)gk * ld_k + gr;
        if (gr + 3 < nrows) {
          v = *reinterpret_cast<const float4*>(p);
        } else {
          if (gr + 0 < nrows) v.x = p[0];
          if (gr + 1 < nrows) v.y = p[1];
          if (gr + 2 < nrows) v.z = p[2];
        }
      }
      T* s = S + rv * LD + k;
      s[0] = cvt<T>(v.x); s[LD] = cvt<T>(v.y); s[2 * LD] = cvt<T>(v.z); s[3 * LD] = cvt<T>(v.w);
    }
  } else {
#pragma unroll 4
    for (int i = tid; i < ROWS * BK; i += NT) {
      const int r = i / BK, k = i % BK;
      const int gr = row0 + r, gk = k0 + k;
      float v = 0.f;
      if (gr < nrows && gk < kend) v = G[(long)gr * ld_row + (long)gk * ld_k];
      S[r * LD + k] = cvt<T>(v);
    }
  }
}

template <int PREC, int AM, int BMODE, int CFG>
__global__ __launch_bounds__(NT) void gemm_kernel(GemmP p) {
  using C_ = Cfg<CFG>;
  using L_ = Lds<PREC>;
  using T = typename L_::T;
  constexpr int LD = L_::LD;
  constexpr int BM = C_::WM * C_::TM * 16, BN = C_::WN * C_::TN * 16;
  __shared__ __attribute__((aligned(16))) T smem[(BM + BN) * LD];
  T* As = smem;
  T* Bs = smem + BM * LD;

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave / C_::WN, wn = wave % C_::WN;
  const int m0 = blockIdx.x * BM, n0 = blockIdx.y * BN;
  const int kbeg = blockIdx.z * p.kchunk;
  const int kend = min(p.K, kbeg + p.kchunk);

  f32x4 acc[C_::TM][C_::TN];
#pragma unroll
  for (int i = 0; i < C_::TM; ++i)
#pragma unroll
    for (int j = 0; j < C_::TN; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

  // implicit-GEMM row decode (thread's rows are the same for every k-tile)
  constexpr int VPR = BK / 4;
  constexpr int RPT = (BM * VPR + NT - 1) / NT;  // rows per thread in the vector loader
  long c_img[RPT]; int c_h[RPT], c_w[RPT];
  if constexpr (AM == 3) {
#pragma unroll
    for (int s = 0; s < RPT; ++s) {
      const int r = (tid + s * NT) / VPR;
      const int gm = m0 + r;
      if (gm < p.M) {
        const int wo = gm % p.cWo, t = gm / p.cWo, ho = t % p.cHo, n = t / p.cHo;
        c_img[s] = (long)n * p.cH * p.cW * p.cCin;
        c_h[s] = ho * p.cStride - p.cPad;
        c_w[s] = wo * p.cStride - p.cPad;
      } else {
        c_img[s] = -1; c_h[s] = 0; c_w[s] = 0;
      }
    }
  }

  for (int k0 = kbeg; k0 < kend; k0 += BK) {
    if constexpr (AM == 3) {
#pragma unroll
      for (int s = 0; s < RPT; ++s) {
        const int i = tid + s * NT;
        if (i < BM * VPR) {
          const int r = i / VPR, kv = (i % VPR) * 4, gk = k0 + kv;
          float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
          if (c_img[s] >= 0 && gk < kend) {
            const int tap = gk / p.cCin, c = gk - tap * p.cCin;
            const int kh = tap / p.cKs, kw = tap - kh * p.cKs;
            const int hi = c_h[s] + kh, wi = c_w[s] + kw;
            if (hi >= 0 && hi < p.cH && wi >= 0 && wi < p.cW)
              v = *reinterpret_cast<const float4*>(p.A + c_img[s] + ((long)hi * p.cW + wi) * p.cCin + c);
          }
          T* sp = As + r * LD + kv;
          sp[0] = cvt<T>(v.x); sp[1] = cvt<T>(v.y); sp[2] = cvt<T>(v.z); sp[3] = cvt<T>(v.w);
        }
      }
    } else {
      load_tile<BM, AM, T, LD>(As, p.A, p.lda_m, p.lda_k, m0, p.M, k0, kend, tid);
    }
    load_tile<BN, BMODE, T, LD>(Bs, p.B, p.ldb_n, p.ldb_k, n0, p.N, k0, kend, tid);
    __syncthreads();

    const int fr = lane & 15, fq = lane >> 4;
    if constexpr (PREC == 0) {
#pragma unroll
      for (int kk = 0; kk < BK / 4; ++kk) {
        float a[C_::TM], b[C_::TN];
#pragma unroll
        for (int i = 0; i < C_::TM; ++i) a[i] = As[((wm * C_::TM + i) * 16 + fr) * LD + kk * 4 + fq];
#pragma unroll
        for (int j = 0; j < C_::TN; ++j) b[j] = Bs[((wn * C_::TN + j) * 16 + fr) * LD + kk * 4 + fq];
#pragma unroll
        for (int i = 0; i < C_::TM; ++i)
#pragma unroll
          for (int j = 0; j < C_::TN; ++j)
            acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[i], b[j], acc[i][j], 0, 0, 0);
      }
    } else {
      bf16x8 a[C_::TM], b[C_::TN];
#pragma unroll
      for (int i = 0; i < C_::TM; ++i)
        a[i] = *reinterpret_cast<const bf16x8*>(As + ((wm * C_::TM + i) * 16 + fr) * LD + fq * 8);
#pragma unroll
      for (int j = 0; j < C_::TN; ++j)
        b[j] = *reinterpret_cast<const bf16x8*>(Bs + ((wn * C_::TN + j) * 16 + fr) * LD + fq * 8);
#pragma unroll
      for (int i = 0; i < C_::TM; ++i)
#pragma unroll
        for (int j = 0; j < C_::TN; ++j)
          acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[i], b[j], acc[i][j], 0, 0, 0);
    }
    __syncthreads();
  }

  // ---- epilogue ----
  const int fr = lane & 15, fq = lane >> 4;
#pragma unroll
  for (int i = 0; i < C_::TM; ++i) {
#pragma unroll
    for (int j = 0; j < C_::TN; ++j) {
      const int n = n0 + (wn * C_::TN + j) * 16 + fr;
      if (n >= p.N) continue;
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int m = m0 + (wm * C_::TM + i) * 16 + fq * 4 + r;
        if (m >= p.M) continue;
        float v = acc[i][j][r];
        if (p.ws) {
          p.ws[((long)blockIdx.z * p.M + m) * p.N + n] = v;
          continue;
        }
        if (p.bias) v += p.bias[n];
        if (p.res && p.res_before_act) v += p.res[(long)(m % p.res_rows) * p.ldr + n];
        if (p.preact) p.preact[(long)m * p.ldp + n] = v;
        v = apply_act(v, p.act);
        if (p.dact) v *= act_grad(p.dsrc[(long)m * p.ldd + n], p.dact);
        if (p.res && !p.res_before_act) v += p.res[(long)(m % p.res_rows) * p.ldr + n];
        p.C[(long)m * p.ldc + n] = v;
      }
    }
  }
}

// Sum of the split-K slabs + the epilogue.  One element per thread trip; four slabs in flight per trip (clamped slab
// index, masked add -- the slabs are still summed in ascending order) and the epilogue's operands requested before the
// slab loop through branch-free selects: the plain form (a runtime-count loop of load -> add, then one uniform branch
// per optional operand) was a chain of splits + 3 dependent memory round trips for a 1-MB tensor.
__global__ __launch_bounds__(256) void splitk_reduce_kernel(GemmP p, int splits) {
  const unsigned total = (unsigned)p.M * (unsigned)p.N;  // rf_gemm bounds M * N * splits below 2^31 for this path
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

// =============================================================================================
// v2: software-pipelined variant.  Same tiles / fragments / epilogue, but
//   * BK = 64 (bf16) / 32 (f32) per stage, two LDS stages;
//   * the global loads of stage t+1 are issued into registers BEFORE the MFMAs of stage t and
//     written to the other LDS stage after them -> one barrier per K-step, HBM/L2 latency hidden
//     behind the matrix work of the same workgroup (these GEMMs run at ~1-2 workgroups per CU, so
//     there is little thread-level parallelism to hide it otherwise);
//   * extra tile config 3 = 128x64 (4x1 waves, 2x4 tiles) for the tall activations (M >= 4096).
// Used whenever both operands are 16-B vectorizable (modes 0/1) and for the implicit-GEMM convs.
// =============================================================================================
template <> struct Cfg<3> { static constexpr int WM = 4, WN = 1, TM = 2, TN = 4; };

template <int PREC> struct Lds2;
template <> struct Lds2<0> { using T = float; static constexpr int BKV = 32, LD = 34; };
template <> struct Lds2<1> { using T = __bf16; static constexpr int BKV = 64, LD = 72; };

template <typename T> __device__ __forceinline__ void st4(T* s, const float4& v);
template <> __device__ __forceinline__ void st4<float>(float* s, const float4& v) {
  *reinterpret_cast<float2*>(s) = make_float2(v.x, v.y);
  *reinterpret_cast<float2*>(s + 2) = make_float2(v.z, v.w);
}
template <> __device__ __forceinline__ void st4<__bf16>(__bf16* s, const float4& v) {
  typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
  bf16x4 o = {(__bf16)v.x, (__bf16)v.y, (__bf16)v.z, (__bf16)v.w};
  *reinterpret_cast<bf16x4*>(s) = o;
}

// global -> registers (NV float4 per thread) for a ROWS x BKV tile; MODE 0: k contiguous, 1: row contiguous
template <int ROWS, int BKV, int MODE, int NV>
__device__ __forceinline__ void gload(float4 (&r)[NV], const float* __restrict__ G, long ld_row, long ld_k,
                                      int row0, int nrows, int k0, int kend, int tid) {
#pragma unroll
  for (int s = 0; s < NV; ++s) {
    const int i = tid + s * NT;
    float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
    if constexpr (MODE == 0) {
      constexpr int VPR = BKV / 4;
      const int rr = i / VPR, kv = (i % VPR) * 4;
      const int gr = row0 + rr, gk = k0 + kv;
      if (i < ROWS * VPR && gr < nrows && gk < kend) {
        const float* p = G + (long)gr * ld_row + gk;
        if (gk + 3 < kend) {
          v = *reinterpret_cast<const float4*>(p);
        } else {
          v.x = p[0];
          if (gk + 1 < kend) v.y = p[1];
          if (gk + 2 < kend) v.z = p[2];
        }
      }
    } else {
      constexpr int VPK = ROWS / 4;
      const int k = i / VPK, rv = (i % VPK) * 4;
      const int gr = row0 + rv, gk = k0 + k;
      if (i < BKV * VPK && gk < kend && gr < nrows) {
        const float* p = G + (long)gk * ld_k + gr;
        if (gr + 3 < nrows) {
          v = *reinterpret_cast<const float4*>(p);
        } else {
          v.x = p[0];
          if (gr + 1 < nrows) v.y = p[1];
          if (gr + 2 < nrows) v.z = p[2];
        }
      }
    }
    r[s] = v;
  }
}

// The same tile fetch with everything loop-invariant hoisted: per slot the address of its float4 at k = 0
// (NULL when the slot lies outside the tile / the matrix) and its k offset are computed once, a K-step then
// costs one add and one compare per load instead of the index arithmetic (the 64x64 kernels run one wave per
// SIMD and are instruction-issue bound: ~250 instructions per wave and K-step before this).
__device__ const float4 rf_zero16 = {0.f, 0.f, 0.f, 0.f};  // what out-of-range tile slots read

template <int ROWS, int BKV, int MODE, int NV>
struct GLoader {
  // Branch-free: every slot issues its 16-B load unconditionally; slots outside the tile / matrix / K range
  // read a 16-B block of zeros instead (no select on the loaded value, so nothing waits for the data).  The predicated form cost ~8 exec-mask
  // branches per load -- 1.5k of the 2.5k cycles of a K-step (tools/gemm_phase_probe.py).  Requires whole
  // float4s: K % 4 == 0 (MODE 0) / rows % 4 == 0 (MODE 1), which rf_gemm checks before choosing this kernel.
  const float* base[NV];
  int kofs[NV];
  bool ok[NV];
  long kstep;  // elements per unit of k
  __device__ __forceinline__ void init(const float* __restrict__ G, long ld_row, long ld_k, int row0, int nrows, int tid) {
    kstep = MODE == 0 ? 1 : ld_k;
#pragma unroll
    for (int s = 0; s < NV; ++s) {
      const int i = tid + s * NT;
      if constexpr (MODE == 0) {
        constexpr int VPR = BKV / 4;
        const int rr = i / VPR, kv = (i % VPR) * 4, gr = row0 + rr;
        kofs[s] = kv;
        ok[s] = i < ROWS * VPR && gr < nrows;
        base[s] = G + (ok[s] ? (long)gr * ld_row + kv : 0);
      } else if constexpr (NV % 2 == 0) {
        // slots come in pairs (2p, 2p+1) holding the SAME four rows at two adjacent k: the transposing LDS store
        // then writes (k, k+1) pairs -- half the ds_write instructions, which were 58 % of a weight-gradient K trip
        constexpr int VPK = ROWS / 4;
        const int ip = tid + (s >> 1) * NT;
        const int k = 2 * (ip / VPK) + (s & 1), rv = (ip % VPK) * 4, gr = row0 + rv;
        kofs[s] = k;
        ok[s] = ip < BKV * VPK / 2 && gr < nrows;
        base[s] = G + (ok[s] ? (long)k * ld_k + gr : 0);
      } else {
        constexpr int VPK = ROWS / 4;
        const int k = i / VPK, rv = (i % VPK) * 4, gr = row0 + rv;
        kofs[s] = k;
        ok[s] = i < BKV * VPK && gr < nrows;
        base[s] = G + (ok[s] ? (long)k * ld_k + gr : 0);
      }
    }
  }
  __device__ __forceinline__ void load(float4 (&r)[NV], int k0, int kend) const {
    const long adv = (long)k0 * kstep;
#pragma unroll
    for (int s = 0; s < NV; ++s) {
      const bool in = ok[s] && (k0 + kofs[s] < kend);
      const float* p = in ? base[s] + adv : reinterpret_cast<const float*>(&rf_zero16);
      r[s] = *reinterpret_cast<const float4*>(p);
    }
  }
};

#ifdef RF_GEMM_NOSWZ  // (A/B build: tools/probes, RF_HIP_LIB)
constexpr bool GEMM_SWZ = false;
#else
constexpr bool GEMM_SWZ = true;
#endif
// registers -> LDS stage ([row][k], pitch LD)
template <typename T> __device__ __forceinline__ void st2(T* s, float a, float b);
template <> __device__ __forceinline__ void st2<float>(float* s, float a, float b) {
  *reinterpret_cast<float2*>(s) = make_float2(a, b);
}
template <> __device__ __forceinline__ void st2<__bf16>(__bf16* s, float a, float b) {
  typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
  const bf16x2 o = {(__bf16)a, (__bf16)b};
  *reinterpret_cast<bf16x2*>(s) = o;
}

template <int ROWS, int BKV, int MODE, int NV, typename T, int LD>
__device__ __forceinline__ void lstore(T* __restrict__ S, const float4 (&r)[NV], int tid) {
  if constexpr (MODE == 1 && NV % 2 == 0) {  // paired slots (see GLoader::init): (k, k+1) pairs of four rows
    constexpr int VPK = ROWS / 4;
#pragma unroll
    for (int p = 0; p < NV / 2; ++p) {
      const int ip = tid + p * NT;
      if (ip < BKV * VPK / 2) {
        // bf16 stage: the 16-B chunk a (k, k+1) pair lands in is XOR-ed with bits 4-5 of its row (swz_chunk; the fragment
        // reads apply the same permutation).  Unswizzled, the 16 row groups of a wave instruction sit 4 x 144 B apart, i.e.
        // on 4 banks x 4 k-pairs = 16 of the 64 banks: a 4-way conflict on every transposing store (PMC: 0.59 of the LDS
        // cycles of the dX products).  With it the four row groups that shared a bank take four different chunks.
        const int g = ip % VPK, kp = ip / VPK;
        T* sp = S + (g * 4) * LD + ((GEMM_SWZ && sizeof(T) == 2) ? (((kp >> 2) ^ ((g >> 2) & 3)) << 3) + ((kp & 3) << 1) : 2 * kp);
        st2<T>(sp, r[2 * p].x, r[2 * p + 1].x);
        st2<T>(sp + LD, r[2 * p].y, r[2 * p + 1].y);
        st2<T>(sp + 2 * LD, r[2 * p].z, r[2 * p + 1].z);
        st2<T>(sp + 3 * LD, r[2 * p].w, r[2 * p + 1].w);
      }
    }
    return;
  }
#pragma unroll
  for (int s = 0; s < NV; ++s) {
    const int i = tid + s * NT;
    if constexpr (MODE == 0) {
      constexpr int VPR = BKV / 4;
      if (i < ROWS * VPR) st4<T>(S + (i / VPR) * LD + (i % VPR) * 4, r[s]);
    } else {
      constexpr int VPK = ROWS / 4;
      if (i < BKV * VPK) {
        T* sp = S + ((i % VPK) * 4) * LD + i / VPK;
        sp[0] = cvt<T>(r[s].x); sp[LD] = cvt<T>(r[s].y); sp[2 * LD] = cvt<T>(r[s].z); sp[3 * LD] = cvt<T>(r[s].w);
      }
    }
  }
}

// Phase timing aid (tools/gemm_phase_probe.py builds a private copy with -DRF_GEMM_TIMING)
#ifdef RF_GEMM_TIMING
__device__ unsigned long long rf_gemm_timing[16 * 1024];
#define GM_MARK(k) do { if (threadIdx.x == 0 && blockIdx.x < 1024 && blockIdx.z == 0) rf_gemm_timing[blockIdx.x * 16 + (k)] = __builtin_readcyclecounter(); } while (0)
#else
#define GM_MARK(k) do {} while (0)
#endif

struct BlockId { int x, y, z, gx, gz; };  // position of this workgroup inside ITS problem's grid

template <int PREC, int AM, int BMODE, int CFG>
__device__ __forceinline__ void gemm2_body(const GemmP& p, const BlockId blk) {
  using C_ = Cfg<CFG>;
  using L_ = Lds2<PREC>;
  using T = typename L_::T;
  constexpr int LD = L_::LD, BKV = L_::BKV;
  constexpr int BM = C_::WM * C_::TM * 16, BN = C_::WN * C_::TN * 16;
  constexpr int NA = (BM * BKV / 4 + NT - 1) / NT, NB = (BN * BKV / 4 + NT - 1) / NT;
  // operands staged by the paired transposing store (lstore, MODE 1) carry the chunk swizzle; the fragment reads undo it
  constexpr bool ASWZ = GEMM_SWZ && PREC == 1 && AM == 1 && NA % 2 == 0, BSWZ = GEMM_SWZ && PREC == 1 && BMODE == 1 && NB % 2 == 0;
  __shared__ __attribute__((aligned(16))) T smem[2 * (BM + BN) * LD];
  T* As0 = smem;
  T* Bs0 = smem + 2 * BM * LD;

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave / C_::WN, wn = wave % C_::WN;
  // blockIdx.x walks the N tiles: the column tiles that share one A row-panel are dispatched together,
  // so the panel is fetched from HBM once and re-read from L2
  const int m0 = blk.y * BM, n0 = blk.x * BN;
  const int kbeg = blk.z * p.kchunk;
  const int kend = min(p.K, kbeg + p.kchunk);
  const int fr = lane & 15, fq = lane >> 4;
  // the epilogue column of this thread and its bias: requested now, needed after the K loop
  const int ecol = n0 + tid % BN;
  const float bias_v = (p.bias && ecol < p.N) ? p.bias[ecol] : 0.f;

  f32x4 acc[C_::TM][C_::TN];
#pragma unroll
  for (int i = 0; i < C_::TM; ++i)
#pragma unroll
    for (int j = 0; j < C_::TN; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

  // implicit-GEMM (AM == 3): decode this thread's output pixels once.  bf16 maps with C_in % 8 == 0 ("wide"): a
  // slot is 8 channels = one 16-B load that goes to the bf16 LDS stage as it is (half the load instructions of
  // the 4-channel slots, no convert); the first NA/2 slots of a thread then cover the tile.
  constexpr int VPR = BKV / 4;
  bool wide = false;
  if constexpr (AM == 3 && PREC == 1) wide = p.act_bf16 && (p.cCin & 7) == 0;
  constexpr int SH4 = VPR == 16 ? 4 : (VPR == 8 ? 3 : 2);  // log2(slots per row)
  static_assert((1 << SH4) == VPR, "slots per row must be a power of two");
  const int vshift = wide ? SH4 - 1 : SH4;
  long c_img[NA]; int c_h[NA], c_w[NA];
  if constexpr (AM == 3) {
#pragma unroll
    for (int s = 0; s < NA; ++s) {
      const int gm = m0 + ((tid + s * NT) >> vshift);
      if (gm < p.M && (tid + s * NT) < (BM << vshift)) {
        const int wo = gm % p.cWo, t = gm / p.cWo, ho = t % p.cHo, n = t / p.cHo;
        c_img[s] = (long)n * p.cH * p.cW * p.cCin;
        c_h[s] = ho * p.cStride - p.cPad;
        c_w[s] = wo * p.cStride - p.cPad;
      } else {
        c_img[s] = -1; c_h[s] = 0; c_w[s] = 0;
      }
    }
  }

  GLoader<BM, BKV, (AM == 3 ? 0 : AM), NA> la;
  GLoader<BN, BKV, BMODE, NB> lb;
  if constexpr (AM != 3) la.init(p.A, p.lda_m, p.lda_k, m0, p.M, tid);
  lb.init(p.B, p.ldb_n, p.ldb_k, n0, p.N, tid);
  float4 ra[NA], rb[NB];
  float4 rsum[NA];  // per-thread partial row sums of A (bias-gradient side product, AM == 1 only)
  if constexpr (AM == 1) {
#pragma unroll
    for (int s = 0; s < NA; ++s) rsum[s] = make_float4(0.f, 0.f, 0.f, 0.f);
  }
  auto load_a = [&](int k0) {
    if constexpr (AM == 3) {
      if constexpr (PREC == 1) {
        if (wide) {
#pragma unroll
          for (int s = 0; s < (NA + 1) / 2; ++s) {
            const int gk = k0 + ((tid + s * NT) % (BKV / 8)) * 8;
            float4 v = make_float4(0.f, 0.f, 0.f, 0.f);  // (eight bf16 zeros)
            if (c_img[s] >= 0 && gk < kend) {
              const int tap = gk / p.cCin, c = gk - tap * p.cCin;
              const int kh = tap / p.cKs, kw = tap - kh * p.cKs;
              const int hi = c_h[s] + kh, wi = c_w[s] + kw;
              if (hi >= 0 && hi < p.cH && wi >= 0 && wi < p.cW)
                v = *reinterpret_cast<const float4*>(reinterpret_cast<const __bf16*>(p.A) + c_img[s] +
                                                     ((long)hi * p.cW + wi) * p.cCin + c);
            }
            ra[s] = v;
          }
          return;
        }
      }
#pragma unroll
      for (int s = 0; s < NA; ++s) {
        const int gk = k0 + ((tid + s * NT) % VPR) * 4;
        float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
        if (c_img[s] >= 0 && gk < kend) {
          const int tap = gk / p.cCin, c = gk - tap * p.cCin;
          const int kh = tap / p.cKs, kw = tap - kh * p.cKs;
          const int hi = c_h[s] + kh, wi = c_w[s] + kw;
          if (hi >= 0 && hi < p.cH && wi >= 0 && wi < p.cW) {
            const long at = c_img[s] + ((long)hi * p.cW + wi) * p.cCin + c;
            v = p.act_bf16 ? act_ld4(reinterpret_cast<const __bf16*>(p.A) + at) : act_ld4(p.A + at);
          }
        }
        ra[s] = v;
      }
    } else {
      la.load(ra, k0, kend);
      if constexpr (AM == 1) {
        if (p.a_rowsum) {
#pragma unroll
          for (int s = 0; s < NA; ++s) {
            rsum[s].x += ra[s].x; rsum[s].y += ra[s].y; rsum[s].z += ra[s].z; rsum[s].w += ra[s].w;
          }
        }
      }
    }
  };
  auto store_a = [&](T* As) {
    if constexpr (AM == 3 && PREC == 1) {
      if (wide) {
#pragma unroll
        for (int s = 0; s < (NA + 1) / 2; ++s) {
          const int i = tid + s * NT;
          if (i < BM * (BKV / 8)) *reinterpret_cast<float4*>(As + (i / (BKV / 8)) * LD + (i % (BKV / 8)) * 8) = ra[s];
        }
        return;
      }
    }
    if constexpr (AM == 3) lstore<BM, BKV, 0, NA, T, LD>(As, ra, tid);
    else lstore<BM, BKV, AM, NA, T, LD>(As, ra, tid);
  };

  GM_MARK(0);
  load_a(kbeg);
  lb.load(rb, kbeg, kend);
  store_a(As0);
  lstore<BN, BKV, BMODE, NB, T, LD>(Bs0, rb, tid);
  __syncthreads();
  GM_MARK(1);

  int cur = 0;
  for (int k0 = kbeg; k0 < kend; k0 += BKV) {
    const bool more = k0 + BKV < kend;
    const bool probe = k0 == kbeg + BKV;  // second trip
    if (probe) GM_MARK(2);
    if (more) {  // next stage's global loads go out before this stage's matrix work
      load_a(k0 + BKV);
      lb.load(rb, k0 + BKV, kend);
    }
    if (probe) GM_MARK(3);
    const T* As = As0 + cur * BM * LD;
    const T* Bs = Bs0 + cur * BN * LD;
    if constexpr (PREC == 0) {
#pragma unroll
      for (int kk = 0; kk < BKV / 4; ++kk) {
        float a[C_::TM], b[C_::TN];
#pragma unroll
        for (int i = 0; i < C_::TM; ++i) a[i] = As[((wm * C_::TM + i) * 16 + fr) * LD + kk * 4 + fq];
#pragma unroll
        for (int j = 0; j < C_::TN; ++j) b[j] = Bs[((wn * C_::TN + j) * 16 + fr) * LD + kk * 4 + fq];
#pragma unroll
        for (int i = 0; i < C_::TM; ++i)
#pragma unroll
          for (int j = 0; j < C_::TN; ++j)
            acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[i], b[j], acc[i][j], 0, 0, 0);
      }
    } else {
#pragma unroll
      for (int ks = 0; ks < BKV / 32; ++ks) {
        bf16x8 a[C_::TM], b[C_::TN];
#pragma unroll
        for (int i = 0; i < C_::TM; ++i)
          a[i] = *reinterpret_cast<const bf16x8*>(As + ((wm * C_::TM + i) * 16 + fr) * LD +
                                                  (ASWZ ? ((ks * 4 + fq) ^ ((wm * C_::TM + i) & 3)) : ks * 4 + fq) * 8);
#pragma unroll
        for (int j = 0; j < C_::TN; ++j)
          b[j] = *reinterpret_cast<const bf16x8*>(Bs + ((wn * C_::TN + j) * 16 + fr) * LD +
                                                  (BSWZ ? ((ks * 4 + fq) ^ ((wn * C_::TN + j) & 3)) : ks * 4 + fq) * 8);
#pragma unroll
        for (int i = 0; i < C_::TM; ++i)
#pragma unroll
          for (int j = 0; j < C_::TN; ++j)
            acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[i], b[j], acc[i][j], 0, 0, 0);
      }
    }
    if (probe) GM_MARK(4);
    if (more) {
      store_a(As0 + (cur ^ 1) * BM * LD);
      lstore<BN, BKV, BMODE, NB, T, LD>(Bs0 + (cur ^ 1) * BN * LD, rb, tid);
    }
    if (probe) GM_MARK(5);
    __syncthreads();
    if (probe) GM_MARK(6);
    cur ^= 1;
  }
  GM_MARK(7);

  if constexpr (AM == 1) {
    // bias-gradient side product: only the first column tile of each row panel contributes (the other
    // column tiles stage the same A panel); rows of a thread: 4*(tid % (BM/4)) .. +3 in every slot
    if (p.a_rowsum && blk.x == 0) {
      constexpr int VPK = BM / 4;
      float4 t = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
      for (int s = 0; s < NA; ++s) {
        if (tid + s * NT < BKV * VPK) { t.x += rsum[s].x; t.y += rsum[s].y; t.z += rsum[s].z; t.w += rsum[s].w; }
      }
      float* red = reinterpret_cast<float*>(smem);  // all stages are dead after the last barrier
      if (tid < BM) red[tid] = 0.f;
      __syncthreads();
      const int rv = (tid % VPK) * 4;
      atomicAdd(&red[rv + 0], t.x); atomicAdd(&red[rv + 1], t.y);
      atomicAdd(&red[rv + 2], t.z); atomicAdd(&red[rv + 3], t.w);
      __syncthreads();
      if (tid < BM && m0 + tid < p.M) atomicAdd(&p.a_rowsum[m0 + tid], red[tid]);
    }
  }

  // ---- epilogue ----
  // The accumulators go through an fp32 LDS tile and ONE rolled loop finishes them, a wave per output row
  // segment: (a) the bias / activation / residual code exists once instead of once per accumulator
  // register -- the unrolled form made this kernel 70-140 KB of instructions, more than the 64 KB
  // instruction cache, so every launch streamed its own code from L2; (b) stores are row-contiguous
  // (256 B per wave instruction) instead of 64-B column fragments of the MFMA layout.
  constexpr int CP = BN + 4;  // pitch: conflict-free for the C-fragment writes
  static_assert(BM * CP * 4 <= 2 * (BM + BN) * LD * (int)sizeof(T), "C staging tile must fit in the K-loop stages");
  const int mode = p.atomic ? 1 : (p.ws ? 2 : 0);
  if (mode != 0) {
    // split-K partials (fp32 atomics into the gradient slot, or a slab of the workspace): nothing to
    // compute, so they leave straight from the accumulator registers
#pragma unroll
    for (int i = 0; i < C_::TM; ++i) {
#pragma unroll
      for (int j = 0; j < C_::TN; ++j) {
        const int n = n0 + (wn * C_::TN + j) * 16 + fr;
        if (n >= p.N) continue;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int m = m0 + (wm * C_::TM + i) * 16 + fq * 4 + r;
          if (m >= p.M) continue;
          if (mode == 1) {
            if (p.atomic == 2) p.C[(long)m * p.ldc + n] = acc[i][j][r];
            else atomicAdd(&p.C[(long)m * p.ldc + n], acc[i][j][r]);
          } else {
            p.ws[((long)blk.z * p.M + m) * p.N + n] = acc[i][j][r];
          }
        }
      }
    }
  }
  float* ct = reinterpret_cast<float*>(smem);
  if (mode == 0) {
    __syncthreads();  // K-loop stages (and the row-sum scratch) are dead
#pragma unroll
    for (int i = 0; i < C_::TM; ++i)
#pragma unroll
      for (int j = 0; j < C_::TN; ++j)
#pragma unroll
        for (int r = 0; r < 4; ++r)
          ct[((wm * C_::TM + i) * 16 + fq * 4 + r) * CP + (wn * C_::TN + j) * 16 + fr] = acc[i][j][r];
    __syncthreads();
  }

  // Every element a thread finishes lies in ONE column (NT % BN == 0) at rows er0, er0 + RSTEP, ...: its bias
  // was loaded at the top of the kernel, and the residual / activation-source loads of trip t + 1 are issued
  // before trip t is computed -- the epilogue no longer pays one L2 latency per four elements.
  constexpr int RSTEP = NT / BN, EPT = BM * BN / NT;
  static_assert(NT % BN == 0 && EPT % 4 == 0, "tile must split into 4-element single-column trips");
  const int er0 = tid / BN;
  auto epi_inputs = [&](int k0, float (&rs)[4], float (&ds)[4]) {
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int m = m0 + er0 + (k0 + u) * RSTEP;
      const bool ok = m < p.M && ecol < p.N;
      if constexpr (AM == 3) {
        const long at = (long)(m % p.res_rows) * p.ldr + ecol;
        rs[u] = (ok && p.res) ? (p.act_bf16 ? act_ld(reinterpret_cast<const __bf16*>(p.res) + at) : p.res[at]) : 0.f;
      } else {
        rs[u] = (ok && p.res) ? p.res[(long)(m % p.res_rows) * p.ldr + ecol] : 0.f;
      }
      ds[u] = (ok && p.dact) ? p.dsrc[(long)m * p.ldd + ecol] : 0.f;
    }
  };
  const bool plain = !p.act && !p.dact && !p.preact;  // bias (+ residual) only: the common case
  auto finish_trip = [&](int k0, const float (&v)[4], const float (&rs)[4], const float (&ds)[4]) {
    if (plain) {  // four independent, branch-free chains (one wave per SIMD: dependent issue is ~8 cycles)
      float t[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) t[u] = v[u] + bias_v + rs[u];
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const int m = m0 + er0 + (k0 + u) * RSTEP;
        if (m < p.M && ecol < p.N) {
          if (AM == 3 && p.act_bf16) act_st(reinterpret_cast<__bf16*>(p.C) + (long)m * p.ldc + ecol, t[u]);
          else p.C[(long)m * p.ldc + ecol] = t[u];
        }
      }
      return;
    }
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int m = m0 + er0 + (k0 + u) * RSTEP;
      if (m >= p.M || ecol >= p.N) continue;
      float t = v[u] + bias_v;
      if (p.res_before_act) t += rs[u];
      if (p.preact) p.preact[(long)m * p.ldp + ecol] = t;
      t = apply_act(t, p.act);
      if (p.dact) t *= act_grad(ds[u], p.dact);
      if (!p.res_before_act) t += rs[u];
      if (AM == 3 && p.act_bf16) act_st(reinterpret_cast<__bf16*>(p.C) + (long)m * p.ldc + ecol, t);
      else p.C[(long)m * p.ldc + ecol] = t;
    }
  };
  const bool simple = !p.preact && !p.res_before_act && (p.act == 0 || p.act == RF_ACT_RELU) &&
                      (p.dact == 0 || p.dact == RF_ACT_RELU) && (!p.res || p.res_rows >= p.M);
  bool maps_done = false;
  if constexpr (AM == 3) {
    if (p.act_bf16) {
      // conv epilogue on bf16 maps: y = [relu](acc + bias [+ residual]); four channels per thread and trip --
      // one 16-B LDS read, one 8-B residual load, one 8-B store (rf_conv2d_nhwc checks N, ldc, ldr % 4 == 0)
      constexpr int TPR = BN / 4, ROWS_PER_TRIP = NT / TPR;
      const int c4 = (tid % TPR) * 4, col = n0 + c4;
      if (col < p.N) {
        float4 b4 = make_float4(0.f, 0.f, 0.f, 0.f);
        if (p.bias) b4 = make_float4(p.bias[col], p.bias[col + 1], p.bias[col + 2], p.bias[col + 3]);
        __bf16* cb = reinterpret_cast<__bf16*>(p.C);
        const __bf16* rb = reinterpret_cast<const __bf16*>(p.res);
        const bool relu = p.act == RF_ACT_RELU;
#pragma unroll 2
        for (int r = tid / TPR; r < BM; r += ROWS_PER_TRIP) {
          const int m = m0 + r;
          if (m >= p.M) break;
          float4 v = *reinterpret_cast<const float4*>(ct + r * CP + c4);
          v.x += b4.x; v.y += b4.y; v.z += b4.z; v.w += b4.w;
          if (rb) {
            const float4 q = act_ld4(rb + (long)m * p.ldr + col);
            v.x += q.x; v.y += q.y; v.z += q.z; v.w += q.w;
          }
          if (relu) { v.x = fmaxf(v.x, 0.f); v.y = fmaxf(v.y, 0.f); v.z = fmaxf(v.z, 0.f); v.w = fmaxf(v.w, 0.f); }
          act_st4(cb + (long)m * p.ldc + col, v);
        }
      }
      maps_done = true;
    }
  }
  if (maps_done) {
  } else if (mode == 0 && simple) {
    // y = [relu](acc + bias) [* (src > 0)] [+ residual]: most forward and dX launches.  One column predicate
    // around the whole epilogue, a row count instead of per-element bounds tests, pointers advanced by constant
    // strides, flags applied by selects -- the per-element guarded form spent 8.4k cycles here
    // (tools/gemm_phase_probe.py), more than 4 K-steps
    if (ecol < p.N) {
      const int kmax = min(EPT, (p.M - m0 - er0 + RSTEP - 1) / RSTEP);
      const long row = m0 + er0;
      float* cp = p.C + row * p.ldc + ecol;
      const float* rp = p.res ? p.res + row * p.ldr + ecol : nullptr;
      const float* dp = p.dact ? p.dsrc + row * p.ldd + ecol : nullptr;
      const float* lp = ct + er0 * CP + tid % BN;
      const bool relu = p.act == RF_ACT_RELU;
#pragma unroll 1
      for (int k0 = 0; k0 < kmax; k0 += 4) {
        float v[4], rs[4] = {0.f, 0.f, 0.f, 0.f}, ds[4] = {1.f, 1.f, 1.f, 1.f};
        if (rp) {
#pragma unroll
          for (int u = 0; u < 4; ++u)
            if (k0 + u < kmax) rs[u] = rp[(long)(k0 + u) * RSTEP * p.ldr];
        }
        if (dp) {
#pragma unroll
          for (int u = 0; u < 4; ++u)
            if (k0 + u < kmax) ds[u] = dp[(long)(k0 + u) * RSTEP * p.ldd];
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) {
          float t = lp[(k0 + u) * RSTEP * CP] + bias_v;  // (reads stay inside the tile)
          t = relu ? fmaxf(t, 0.f) : t;
          t = ds[u] > 0.f ? t : 0.f;
          v[u] = t + rs[u];
        }
#pragma unroll
        for (int u = 0; u < 4; ++u)
          if (k0 + u < kmax) cp[(long)(k0 + u) * RSTEP * p.ldc] = v[u];
      }
    }
  } else if (mode == 0) {
    float rs[4], ds[4], nrs[4] = {0.f, 0.f, 0.f, 0.f}, nds[4] = {0.f, 0.f, 0.f, 0.f};
    epi_inputs(0, rs, ds);
#pragma unroll 1
    for (int k0 = 0; k0 < EPT; k0 += 4) {
      if (k0 + 4 < EPT) epi_inputs(k0 + 4, nrs, nds);
      float v[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) v[u] = ct[(er0 + (k0 + u) * RSTEP) * CP + tid % BN];
      finish_trip(k0, v, rs, ds);
#pragma unroll
      for (int u = 0; u < 4; ++u) { rs[u] = nrs[u]; ds[u] = nds[u]; }
    }
  }

  // ---- in-launch split-K reduction: the LAST K-slice workgroup of a tile to arrive sums all slices ----
  // (placement-independent agent-scope release / acquire hand-off, cdna_hip_programming.md "In-launch
  //  split-K reduction"; slices are summed in ascending order whichever workgroup does it -> deterministic)
  if (p.ws && p.tile_cnt && !p.atomic) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // every storing wave drains its partial-slab stores
    __syncthreads();
    int* last_flag = reinterpret_cast<int*>(smem);    // the staging tile is dead too
    unsigned* cnt = p.tile_cnt + (blk.y * blk.gx + blk.x);
    if (tid == 0) {
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      const unsigned ticket = __hip_atomic_fetch_add(cnt, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      const int last = ticket == blk.gz - 1;
      if (last) {
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __hip_atomic_store(cnt, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);  // ready for the next launch
      }
      *last_flag = last;
    }
    __syncthreads();
    if (*last_flag) {
      const int splits = blk.gz;
      const long slab = (long)p.M * p.N;
#pragma unroll 1
      for (int k0 = 0; k0 < EPT; k0 += 4) {
        float v[4], rs[4], ds[4];
        epi_inputs(k0, rs, ds);
#pragma unroll
        for (int u = 0; u < 4; ++u) {
          const int m = m0 + er0 + (k0 + u) * RSTEP;
          v[u] = 0.f;
          if (m < p.M && ecol < p.N)
            for (int z = 0; z < splits; ++z) v[u] += p.ws[z * slab + (long)m * p.N + ecol];
        }
        finish_trip(k0, v, rs, ds);
      }
    }
  }
  GM_MARK(8);
}

// XCD-aware tile order.  Workgroups are dealt to the 8 XCDs round-robin in dispatch order and each XCD has its
// own L2.  share == 1 (few row tiles, big weight matrix -- the M <= 560 layers): all row tiles of one COLUMN
// tile go to the same XCD, so a weight tile is fetched from HBM once instead of once per XCD that happens to
// hold one of its row tiles.  share == 2 (tall activations / implicit-GEMM convs, few column tiles): all
// column tiles of one ROW tile share an XCD, so the activation panel is fetched once.  The 1-D launch is padded
// to a multiple of 8 groups; workgroups that fall outside the tile grid exit at once.
template <int PREC, int AM, int BMODE, int CFG>
__global__ __launch_bounds__(NT) void gemm2_kernel(GemmP p) {
  int x, y;
  const int id = blockIdx.x, xcd = id & 7, j = id >> 3;
  if (p.share == 1) {
    y = j % p.gy;
    x = (j / p.gy) * 8 + xcd;
  } else if (p.share == 2) {
    x = j % p.gx;
    y = (j / p.gx) * 8 + xcd;
  } else {
    x = id % p.gx;
    y = id / p.gx;
  }
  if (x >= p.gx || y >= p.gy) return;
  gemm2_body<PREC, AM, BMODE, CFG>(p, BlockId{x, y, (int)blockIdx.z, p.gx, (int)gridDim.z});
}

// ---------------------------------------------------------------------------------------------
// Grouped weight gradients: dW_e[N_e, K_e] += dY_e[M_e, N_e]^T X_e[M_e, K_e] (+ db_e += colsum dY_e) for up
// to RF_WGRAD_MAX_GROUP layers in ONE launch.  Weight gradients feed only the optimizer, so the backward
// pass queues them and flushes the queue in a few launches that fill all 256 CUs, instead of ~150 small
// split-K launches (13 us each, a handful of workgroups) interleaved with the dX chain.  The table rides in
// the kernel arguments (no device-side descriptor upload; safe under hipGraph capture).
// ---------------------------------------------------------------------------------------------
struct WgradTable {
  int count, pad;
  RfWgradEntry e[RF_WGRAD_MAX_GROUP];
  int first_block[RF_WGRAD_MAX_GROUP + 1];
};

template <int PREC>
__global__ __launch_bounds__(NT) void wgrad_grouped_kernel(const WgradTable t) {
  // which problem does this workgroup belong to?  (uniform binary search over the prefix of block counts)
  const int b = blockIdx.x;
  int lo = 0, hi = t.count - 1;
  while (lo < hi) {
    const int mid = (lo + hi + 1) >> 1;
    if (t.first_block[mid] <= b) lo = mid; else hi = mid - 1;
  }
  const RfWgradEntry& e = t.e[lo];
  GemmP p{};
  p.A = static_cast<const float*>(e.dy); p.lda_m = 1; p.lda_k = e.ld_dy;     // A = dY^T: "row" m = output feature, contiguous across m
  p.B = static_cast<const float*>(e.x); p.ldb_k = e.ld_x; p.ldb_n = 1;
  p.C = e.dw; p.ldc = e.K;
  p.M = e.N; p.N = e.K; p.K = e.M;
  p.res_rows = 1; p.atomic = (e.exclusive && e.splits == 1) ? 2 : 1; p.a_rowsum = e.db;
  p.kchunk = e.kchunk;
  const int gx = (e.K + 63) / 64, gy = (e.N + 63) / 64;
  const int local = b - t.first_block[lo];
  const int bz = local / (gx * gy), rem = local - bz * gx * gy;
  gemm2_body<PREC, 1, 1, 0>(p, BlockId{rem % gx, rem / gx, bz, gx, e.splits});
}

template <int PREC, int AM, int BMODE, int CFG>
void launch2(const GemmP& p, int splits, hipStream_t st) {
  using C_ = Cfg<CFG>;
  constexpr int BM = C_::WM * C_::TM * 16, BN = C_::WN * C_::TN * 16;
  GemmP q = p;
  q.gx = (p.N + BN - 1) / BN;
  q.gy = (p.M + BM - 1) / BM;
  // which operand is worth keeping on one XCD?  the one whose panel is re-read by the other dimension's tiles
  q.share = (q.gx >= 2 && q.gy >= 2) ? (q.gy <= q.gx ? 1 : 2) : 0;
  const int blocks = q.share == 1 ? 8 * q.gy * ((q.gx + 7) / 8) : (q.share == 2 ? 8 * q.gx * ((q.gy + 7) / 8) : q.gx * q.gy);
  RF_LAUNCH((gemm2_kernel<PREC, AM, BMODE, CFG>), dim3(blocks, 1, splits), dim3(NT), 0, st, q);
}

template <int PREC>
void dispatch2(const GemmP& p, int am, int bm, bool tall, int splits, hipStream_t st) {
  if (tall) {
    if (am == 0 && bm == 0) launch2<PREC, 0, 0, 3>(p, splits, st);
    else if (am == 0) launch2<PREC, 0, 1, 3>(p, splits, st);
    else if (bm == 0) launch2<PREC, 1, 0, 3>(p, splits, st);
    else launch2<PREC, 1, 1, 3>(p, splits, st);
  } else {
    if (am == 0 && bm == 0) launch2<PREC, 0, 0, 0>(p, splits, st);
    else if (am == 0) launch2<PREC, 0, 1, 0>(p, splits, st);
    else if (bm == 0) launch2<PREC, 1, 0, 0>(p, splits, st);
    else launch2<PREC, 1, 1, 0>(p, splits, st);
  }
}

template <int PREC, int AM, int BMODE, int CFG>
void launch(const GemmP& p, int splits, hipStream_t st) {
  using C_ = Cfg<CFG>;
  constexpr int BM = C_::WM * C_::TM * 16, BN = C_::WN * C_::TN * 16;
  dim3 grid((p.M + BM - 1) / BM, (p.N + BN - 1) / BN, splits);
  RF_LAUNCH((gemm_kernel<PREC, AM, BMODE, CFG>), grid, dim3(NT), 0, st, p);
}

template <int PREC, int AM>
void dispatch_b(const GemmP& p, int bmode, int splits, hipStream_t st) {
  switch (bmode) {
    case 0: launch<PREC, AM, 0, 0>(p, splits, st); break;
    case 1: launch<PREC, AM, 1, 0>(p, splits, st); break;
    default: launch<PREC, AM, 2, 0>(p, splits, st); break;
  }
}

inline bool aligned16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; }

}  // namespace

namespace {
// K range per split-K slice (a multiple of 64, which covers the k-step of both kernel generations) and the number of
// slices that are not empty
constexpr int GEMM_KQ = 64;
inline int split_chunk(int K, int splitk, int* eff) {
  const int ktiles = (K + GEMM_KQ - 1) / GEMM_KQ;
  if (splitk > ktiles) splitk = ktiles;
  if (splitk < 1) splitk = 1;
  const int kchunk = ((ktiles + splitk - 1) / splitk) * GEMM_KQ;
  *eff = (K + kchunk - 1) / kchunk;
  return kchunk;
}
}  // namespace

// `partials_only`: every slice (also a single one) leaves its raw product in workspace[slice][M][N]; no epilogue, no
// slab sum -- the consumer sums the slabs (rf_layernorm_fwd_slabs)
static int gemm_run(const float* A, int64_t lda_m, int64_t lda_k, const float* B, int64_t ldb_k,
                    int64_t ldb_n, float* C, int64_t ldc, int M, int N, int K, const float* bias,
                    const float* residual, int64_t ldr, int res_rows, int res_before_act, int act,
                    float* preact, int64_t ldp, const float* dact_src, int64_t ldd, int dact_mode,
                    int prec, int splitk, float* workspace, int atomic_accumulate, float* a_rowsum,
                    uint32_t* tile_counters, bool partials_only, void* stream) {
  RF_REQUIRE(A && B && C && M > 0 && N > 0 && K > 0);
  RF_REQUIRE(prec == 0 || prec == 1);
  RF_REQUIRE(splitk >= 1 && (splitk == 1 || workspace != nullptr || atomic_accumulate));
  RF_REQUIRE(!atomic_accumulate || (!bias && !residual && !act && !preact && !dact_mode));
  RF_REQUIRE(!a_rowsum || (atomic_accumulate && lda_m == 1));
  RF_REQUIRE(!residual || res_rows > 0);
  RF_REQUIRE(!dact_mode || dact_src);
  hipStream_t st = static_cast<hipStream_t>(stream);
  GemmP p{};
  p.A = A; p.lda_m = lda_m; p.lda_k = lda_k; p.B = B; p.ldb_k = ldb_k; p.ldb_n = ldb_n;
  p.C = C; p.ldc = ldc; p.M = M; p.N = N; p.K = K; p.bias = bias; p.res = residual; p.ldr = ldr;
  p.res_rows = residual ? res_rows : 1; p.res_before_act = res_before_act; p.act = act;
  p.preact = preact; p.ldp = ldp; p.dsrc = dact_src; p.ldd = ldd; p.dact = dact_mode;
  p.kchunk = split_chunk(K, splitk, &splitk);
  p.ws = ((splitk > 1 && !atomic_accumulate) || partials_only) ? workspace : nullptr;
  p.atomic = atomic_accumulate; p.a_rowsum = a_rowsum;
  p.tile_cnt = nullptr;

  // vector layouts (pipelined kernel): 16-B addressable rows made of whole float4s, else the scalar kernel
  int am = 2, bm = 2;
  if (lda_k == 1 && (lda_m % 4) == 0 && (K % 4) == 0 && aligned16(A)) am = 0;
  else if (lda_m == 1 && (lda_k % 4) == 0 && (M % 4) == 0 && aligned16(A)) am = 1;
  if (ldb_k == 1 && (ldb_n % 4) == 0 && (K % 4) == 0 && aligned16(B)) bm = 0;
  else if (ldb_n == 1 && (ldb_k % 4) == 0 && (N % 4) == 0 && aligned16(B)) bm = 1;

  if (atomic_accumulate && !(am <= 1 && bm <= 1)) {
    rf_g_last_error = "atomic_accumulate needs 16-B vectorizable operands";
    return RF_EUNSUPPORTED;
  }
  if (a_rowsum && am != 1) { rf_g_last_error = "a_rowsum needs a row-contiguous, aligned A"; return RF_EUNSUPPORTED; }
  bool in_kernel_reduce = false;
  if (am <= 1 && bm <= 1) {  // both operands vectorizable: pipelined kernel
    // 128-row tiles only while they still fill the chip: M = 12 480, N = 128 (the camera-token embedding) is 196 of them --
    // one four-wave workgroup on 3/4 of the CUs, 12 dependent K-steps each (67 us); 64 x 64 tiles give 390
    static const long tall_min = [] { const char* e = getenv("RF_GEMM_TALL_MIN"); return e ? atol(e) : 0L; }();
    const bool tall = M >= 4096 && N >= 64 && (long)((M + 127) / 128) * ((N + 63) / 64) * splitk >= tall_min;
    if (splitk > 1 && !atomic_accumulate && tile_counters) {
      const long tiles = (long)((M + (tall ? 127 : 63)) / (tall ? 128 : 64)) * ((N + 63) / 64);
      if (tiles <= 4096) { p.tile_cnt = tile_counters; in_kernel_reduce = true; }  // counter buffer: 4096 tiles
    }
    if (prec == 0) dispatch2<0>(p, am, bm, tall, splitk, st);
    else dispatch2<1>(p, am, bm, tall, splitk, st);
  } else if (prec == 0) {
    if (am == 0) dispatch_b<0, 0>(p, bm, splitk, st);
    else if (am == 1) dispatch_b<0, 1>(p, bm, splitk, st);
    else dispatch_b<0, 2>(p, bm, splitk, st);
  } else {
    if (am == 0) dispatch_b<1, 0>(p, bm, splitk, st);
    else if (am == 1) dispatch_b<1, 1>(p, bm, splitk, st);
    else dispatch_b<1, 2>(p, bm, splitk, st);
  }
  RF_CHECK_LAUNCH();
  if (splitk > 1 && !atomic_accumulate && !in_kernel_reduce && !partials_only) {
    const long total = (long)M * N;
    RF_REQUIRE(total < (1L << 31));  // splitk_reduce_kernel indexes an element with 32 bits
    int blocks = (int)((total + 255) / 256);
    if (blocks > 2048) blocks = 2048;
    RF_LAUNCH(splitk_reduce_kernel, dim3(blocks), dim3(256), 0, st, p, splitk);
    RF_CHECK_LAUNCH();
  }
  return RF_OK;
}

extern "C" int rf_gemm(const float* A, int64_t lda_m, int64_t lda_k, const float* B, int64_t ldb_k,
                       int64_t ldb_n, float* C, int64_t ldc, int M, int N, int K, const float* bias,
                       const float* residual, int64_t ldr, int res_rows, int res_before_act, int act,
                       float* preact, int64_t ldp, const float* dact_src, int64_t ldd, int dact_mode,
                       int prec, int splitk, float* workspace, int atomic_accumulate, float* a_rowsum,
                       uint32_t* tile_counters, void* stream) {
  return gemm_run(A, lda_m, lda_k, B, ldb_k, ldb_n, C, ldc, M, N, K, bias, residual, ldr, res_rows, res_before_act, act,
                  preact, ldp, dact_src, ldd, dact_mode, prec, splitk, workspace, atomic_accumulate, a_rowsum,
                  tile_counters, false, stream);
}

extern "C" int rf_gemm_split_count(int K, int splitk) {
  int eff = 1;
  if (K > 0) (void)split_chunk(K, splitk, &eff);
  return eff;
}

extern "C" int rf_gemm_partials(const float* A, int64_t lda_m, int64_t lda_k, const float* B, int64_t ldb_k,
                                int64_t ldb_n, int M, int N, int K, int prec, int splitk, float* workspace,
                                void* stream) {
  RF_REQUIRE(workspace && splitk >= 1);
  return gemm_run(A, lda_m, lda_k, B, ldb_k, ldb_n, workspace, N, M, N, K, nullptr, nullptr, 0, 0, 0, 0, nullptr, 0,
                  nullptr, 0, 0, prec, splitk, workspace, 0, nullptr, nullptr, true, stream);
}

extern "C" int rf_conv2d_nhwc(const void* x_, const float* w, const float* bias, const void* residual_,
                              void* y_, int act_dtype, int N, int H, int W, int cin, int cout, int ksize, int stride,
                              int pad, int Ho, int Wo, int64_t ldy, int64_t ldres, int relu, int prec,
                              void* stream) {
  // bf16 maps travel through the fp32-typed GemmP fields; the implicit-GEMM loader / epilogue reinterpret them
  const float* x = static_cast<const float*>(x_);
  const float* residual = static_cast<const float*>(residual_);
  float* y = static_cast<float*>(y_);
  RF_REQUIRE(x && w && y && N > 0 && cin > 0 && cout > 0);
  RF_REQUIRE(act_dtype == RF_ACT_F32 || act_dtype == RF_ACT_BF16);
  RF_REQUIRE(act_dtype == RF_ACT_F32 || prec == 1);  // bf16 maps only with the bf16 matrix-core path
  RF_REQUIRE(cin % 4 == 0 && aligned16(w) &&
             (reinterpret_cast<uintptr_t>(x) & ((act_dtype == RF_ACT_BF16 && cin % 8 != 0) ? 7 : 15)) == 0);
  RF_REQUIRE(prec == 0 || prec == 1);
  RF_REQUIRE(Ho == (H + 2 * pad - ksize) / stride + 1 && Wo == (W + 2 * pad - ksize) / stride + 1);
  hipStream_t st = static_cast<hipStream_t>(stream);
  GemmP p{};
  const int K = ksize * ksize * cin;
  p.A = x; p.B = w; p.ldb_k = 1; p.ldb_n = K;  // w: [cout][kh][kw][cin]
  p.C = y; p.ldc = ldy; p.M = N * Ho * Wo; p.N = cout; p.K = K; p.bias = bias;
  p.res = residual; p.ldr = ldres; p.res_rows = p.M; p.res_before_act = 1;
  p.act = relu ? RF_ACT_RELU : RF_ACT_NONE;
  p.kchunk = ((K + 63) / 64) * 64; p.ws = nullptr;
  p.cH = H; p.cW = W; p.cCin = cin; p.cKs = ksize; p.cStride = stride; p.cPad = pad; p.cHo = Ho; p.cWo = Wo;
  p.act_bf16 = act_dtype == RF_ACT_BF16;
  if (p.act_bf16)  // the bf16-map epilogue moves four channels at a time
    RF_REQUIRE(cout % 4 == 0 && ldy % 4 == 0 && (!residual || ldres % 4 == 0) &&
               (reinterpret_cast<uintptr_t>(y) & 7) == 0 && (reinterpret_cast<uintptr_t>(residual) & 7) == 0);
  const int cfg = cout <= 16 ? 1 : (cout <= 32 ? 2 : 0);
  if (prec == 0) {
    if (cfg == 1) launch2<0, 3, 0, 1>(p, 1, st);
    else if (cfg == 2) launch2<0, 3, 0, 2>(p, 1, st);
    else launch2<0, 3, 0, 0>(p, 1, st);
  } else {
    if (cfg == 1) launch2<1, 3, 0, 1>(p, 1, st);
    else if (cfg == 2) launch2<1, 3, 0, 2>(p, 1, st);
    else launch2<1, 3, 0, 0>(p, 1, st);
  }
  RF_CHECK_LAUNCH();
  return RF_OK;
}

// ---- column sums (bias gradients): two deterministic passes --------------------------------
namespace {
constexpr int CS_ROWS = 256;  // rows per first-pass block
// `atomic_out`: the chunk sums are added straight into `part` = the output vector (fp32 atomics: one launch, no second pass)
__global__ __launch_bounds__(256) void colsum_part_kernel(const float* __restrict__ X, long ldx, int M, int N,
                                                           float* __restrict__ part, int atomic_out) {
  // block = 64 columns x 4 row-lanes; blockIdx.y = row chunk
  __shared__ float red[4][64];
  const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
  const int n = blockIdx.x * 64 + tx;
  const int r0 = blockIdx.y * CS_ROWS, r1 = min(M, r0 + CS_ROWS);
  float s = 0.f;
  if (n < N)
    for (int r = r0 + ty; r < r1; r += 4) s += X[(long)r * ldx + n];
  red[ty][tx] = s;
  __syncthreads();
  if (ty == 0 && n < N) {
    const float v = (red[0][tx] + red[1][tx]) + (red[2][tx] + red[3][tx]);
    if (atomic_out) atomicAdd(part + n, v);
    else part[(long)blockIdx.y * N + n] = v;
  }
}
__global__ void colsum_final_kernel(const float* __restrict__ part, int parts, int N, float* __restrict__ out,
                                    int accumulate) {
  const int n = blockIdx.x * blockDim.x + threadIdx.x;
  if (n >= N) return;
  float s = 0.f;
  for (int i = 0; i < parts; ++i) s += part[(long)i * N + n];
  out[n] = accumulate ? out[n] + s : s;
}
}  // namespace

#ifdef RF_GEMM_TIMING
extern "C" void* rf_gemm_timing_address() {
  void* a = nullptr;
  (void)hipGetSymbolAddress(&a, HIP_SYMBOL(rf_gemm_timing));
  return a;
}
#endif

extern "C" int rf_wgrad_tr(const RfWgradEntry* entries, int count, void* stream);

extern "C" int rf_wgrad_grouped(const RfWgradEntry* entries, int count, int prec, void* stream) {
  RF_REQUIRE(entries && count >= 1 && count <= RF_WGRAD_MAX_GROUP && (prec == 0 || prec == 1));
  // bf16 matrix-core mode: the transposed-read kernel (wgrad_tr.hip); RF_WGRAD_TR=0 keeps the tiled one below
  static const bool tr_on = [] { const char* v = getenv("RF_WGRAD_TR"); return !(v && v[0] == '0'); }();
  if (prec == 1 && tr_on) return rf_wgrad_tr(entries, count, stream);
  WgradTable t{};
  t.count = count;
  int blocks = 0;
  constexpr int KQ = 64;
  for (int i = 0; i < count; ++i) {
    RfWgradEntry e = entries[i];
    RF_REQUIRE(e.dy && e.x && e.dw && e.M > 0 && e.N > 0 && e.K > 0 && e.splits >= 1);
    RF_REQUIRE(aligned16(e.dy) && aligned16(e.x) && e.ld_dy % 4 == 0 && e.ld_x % 4 == 0 && e.N % 4 == 0 && e.K % 4 == 0);
    RF_REQUIRE(!e.dy_bf16 && !e.x_bf16);  // bf16 operands: the transposed-read kernel only
    const int ktiles = (e.M + KQ - 1) / KQ;
    int splits = e.splits > ktiles ? ktiles : e.splits;
    e.kchunk = ((ktiles + splits - 1) / splits) * KQ;
    e.splits = (e.M + e.kchunk - 1) / e.kchunk;
    t.e[i] = e;
    t.first_block[i] = blocks;
    blocks += ((e.K + 63) / 64) * ((e.N + 63) / 64) * e.splits;
  }
  t.first_block[count] = blocks;
  hipStream_t st = static_cast<hipStream_t>(stream);
  if (prec == 1) RF_LAUNCH(wgrad_grouped_kernel<1>, dim3(blocks), dim3(NT), 0, st, t);
  else RF_LAUNCH(wgrad_grouped_kernel<0>, dim3(blocks), dim3(NT), 0, st, t);
  RF_CHECK_LAUNCH();
  return RF_OK;
}

extern "C" int rf_colsum_parts(int M, int N) { (void)N; return (M + CS_ROWS - 1) / CS_ROWS; }

extern "C" int rf_colsum(const float* X, int64_t ldx, int M, int N, float* out, int accumulate, float* workspace,
                         void* stream) {
  RF_REQUIRE(X && out && (workspace || accumulate == 2) && M > 0 && N > 0);
  hipStream_t st = static_cast<hipStream_t>(stream);
  const int parts = rf_colsum_parts(M, N);
  if (accumulate == 2) {  // out += column sums by fp32 atomics (order of the additions not fixed): ONE launch
    RF_LAUNCH(colsum_part_kernel, dim3((N + 63) / 64, parts), dim3(256), 0, st, X, (long)ldx, M, N, out, 1);
    RF_CHECK_LAUNCH();
    return RF_OK;
  }
  RF_LAUNCH(colsum_part_kernel, dim3((N + 63) / 64, parts), dim3(256), 0, st, X, (long)ldx, M, N,
                     workspace, 0);
  RF_CHECK_LAUNCH();
  RF_LAUNCH(colsum_final_kernel, dim3((N + 255) / 256), dim3(256), 0, st, workspace, parts, N, out, accumulate);
  RF_CHECK_LAUNCH();
  return RF_OK;
}
