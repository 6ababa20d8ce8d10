// Grouped weight gradients, bf16 matrix-core path (autograd's grad_weight = grad_out^T @ input for every nn.Linear /
// Conv1d(k=1) of the step; SURVEY 8(b) `linear_wgrad`):   dW[N, K] (+)= dY[M, N]^T X[M, K],   db[N] += colsum dY.
// Both operands have the REDUCTION index m as their slow dimension, so an MFMA fragment (8 consecutive m of one
// column) is a column of the operand as it lies in memory.  The tiled kernel in gemm.hip builds [column][m] LDS images
// with transposing 2-B stores and 64 x 64 tiles: paced by those stores, and every operand panel is re-read N/64 or K/64
// times (PMC: 758 MB moved per launch for 281 MB of algorithmic bytes).  Here:
//   * operands are copied ROW-MAJOR into LDS (coalesced 16-B global loads, fp32 -> bf16, 8-B LDS stores; the swizzled
//     256-B-row image (b) of the CDNA4 guide's T10) and read back with gfx950's hardware transpose read
//     ds_read_b64_tr_b16: two reads deliver the 8 consecutive m of a lane's column -- no transposing stores at all;
//   * a workgroup (8 waves, 4 x 2) owns a 256 (n) x 128 (k) block of dW, 64 x 64 per wave in 16 accumulator tiles:
//     a 32-row step stages 32 x 384 operand elements for 32 768 outputs (64 x 64 tiles: 32 x 128 for 4 096) -- 2.7 x
//     less operand traffic per output;
//   * one workgroup per CU (168+ VGPRs), so nothing but the workgroup itself hides the ~1.5 us of a global load under
//     load: a ring of THREE register sets keeps the loads of steps s+1 .. s+3 in flight while step s computes (with one
//     set a 32-row step cost its load latency: measured no faster than the tiled kernel; a 128 x 128 / 4-wave variant
//     squeezed to 128 VGPRs for 4 workgroups per CU spilled its accumulators and was 2.5 x slower); two LDS stages,
//     one barrier per step;
//   * the reduction over M is cut into chunks of >= 1 024 rows only (one chunk for M <= 1 024): a single-chunk
//     exclusive problem writes dW with plain stores, the others add with fp32 atomics; the bias gradient is summed in
//     fp32 from the staged registers (no second pass over dY).
#include "common.h"

namespace {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
typedef short s16x4 __attribute__((ext_vector_type(4)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

constexpr int WT_BN = 256, WT_BK = 128, WT_MS = 32, WT_NT = 512, WT_RING = 3;
constexpr int WT_HALF = WT_MS * 256;  // bytes of one [32 rows][128 columns] bf16 image

struct TrTable {
  int count, run;  // run: blocks an XCD takes in a row (0: plain block order), see wgrad_tr_kernel
  RfWgradEntry e[RF_WGRAD_MAX_GROUP];  // splits = number of M chunks, kchunk = rows per chunk (multiple of 32)
  int first_block[RF_WGRAD_MAX_GROUP + 1];
};

// byte offset of 16-byte chunk `ch` (0..15) of row `row` (0..31) inside a [32][128] bf16 image (guide T10, image (b))
__device__ __forceinline__ int img_off(int row, int ch) {
  return 256 * row + 16 * (ch ^ (((row & 3) << 2) | ((row >> 2) & 3)));
}

// fragment of 8 consecutive rows (8 fq .. 8 fq + 7) of column 16 t + (lane & 15) of an image: two transposed reads
__device__ __forceinline__ bf16x8 tr_frag(const unsigned char* img, int t, int lane) {
  const int l = lane & 15, fq = lane >> 4, q = l >> 2, p = l & 3;
  const int c = 2 * t + (p >> 1), hb = 8 * (p & 1);
  typedef __attribute__((address_space(3))) s16x4 lds_s16x4;
  const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(img + img_off(8 * fq + q, c) + hb));
  const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(img + img_off(8 * fq + 4 + q, c) + hb));
  union { s16x4 s[2]; bf16x8 b; } u;
  u.s[0] = lo;
  u.s[1] = hi;
  return u.b;
}

// Phase timing aid (tools/wgrad_probe.py: private -DRF_WT_TIMING build): lane 0 of every wave of the first 64 workgroups
// stamps the shader clock inside step 4 of its loop.
#ifdef RF_WT_TIMING
__device__ unsigned long long rf_wt_timing[64 * 8 * 8];
#define WT_MARK(k) do { if (s == 4 && (threadIdx.x & 63) == 0 && blockIdx.x < 64) \
  rf_wt_timing[(blockIdx.x * 8 + (threadIdx.x >> 6)) * 8 + (k)] = __builtin_readcyclecounter(); } while (0)
#else
#define WT_MARK(k) do {} while (0)
#endif

// one thread's share of a 32-row step: dY 32 x 256 (4 x 4 columns), X 32 x 128 (2 x 4 columns); an operand that already
// lies in memory as bf16 (YBF / XBF: the fused encoder stacks write their `dy` slabs and activation saves that way -- the
// values were bf16-rounded MFMA operands all along) travels as 8-B groups and reaches LDS without a conversion
template <bool BF> struct Quad;
template <> struct Quad<false> { typedef float4 T; };
template <> struct Quad<true> { typedef uint2 T; };
__device__ __forceinline__ float4 quad_f32(const float4& v) { return v; }
__device__ __forceinline__ float4 quad_f32(const uint2& v) {
  return make_float4(__uint_as_float(v.x << 16), __uint_as_float(v.x & 0xffff0000u), __uint_as_float(v.y << 16),
                     __uint_as_float(v.y & 0xffff0000u));
}
__device__ __forceinline__ bf16x4 quad_bf16(const float4& v) { return bf16x4{(__bf16)v.x, (__bf16)v.y, (__bf16)v.z, (__bf16)v.w}; }
__device__ __forceinline__ bf16x4 quad_bf16(const uint2& v) {
  union { uint2 u; bf16x4 b; } c;
  c.u = v;
  return c.b;
}
__device__ __forceinline__ void quad_zero(float4& v) { v = make_float4(0.f, 0.f, 0.f, 0.f); }
__device__ __forceinline__ void quad_zero(uint2& v) { v = make_uint2(0u, 0u); }

template <bool YBF, bool XBF>
struct Staged { typename Quad<YBF>::T y[4]; typename Quad<XBF>::T x[2]; };

// The block of one (problem, chunk, tile): instantiated per operand-type combination; the kernel below picks the
// instantiation with ONE workgroup-uniform branch at its top (straight-line code inside: the compiler keeps counting the
// outstanding loads of the ring), so fp32-operand and bf16-operand problems share a launch.
template <bool YBF, bool XBF>
__device__ __forceinline__ void wgrad_tr_block(const TrTable& t, const int lo, unsigned char* lds, const int b) {
  typedef typename Quad<YBF>::T YQ;
  typedef typename Quad<XBF>::T XQ;
  typedef Staged<YBF, XBF> StagedT;
  const RfWgradEntry& e = t.e[lo];
  const int M = e.M, N = e.N, K = e.K;
  const int gn = (N + WT_BN - 1) / WT_BN, gk = (K + WT_BK - 1) / WT_BK;
  int local = b - t.first_block[lo];
  const int chunk = local / (gn * gk);
  local -= chunk * gn * gk;
  const int bn = local / gk, bk = local - bn * gk;
  const int n0 = bn * WT_BN, k0 = bk * WT_BK;
  const int m_lo = chunk * e.kchunk, m_hi = min(M, m_lo + e.kchunk);
  const int nsteps = (m_hi - m_lo + WT_MS - 1) / WT_MS;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, fr = lane & 15, fq = lane >> 4;
  const int wn = wave >> 1, wk = wave & 1;  // this wave's 64 x 64 block: rows n0 + 64 wn, columns k0 + 64 wk

  // staging map: dY tile = 32 rows x 64 float4 (4 per thread), X tile = 32 rows x 32 float4 (2 per thread)
  const int yc4 = tid & 63, yr = tid >> 6;         // float4 column, first row (rows yr + 8 u)
  const int xc4 = tid & 31, xr = tid >> 5;         // rows xr + 16 u
  const bool ycol_ok = n0 + 4 * yc4 < N, xcol_ok = k0 + 4 * xc4 < K;
  // (columns beyond N / K: a valid address, the value is dropped); pointers / pitches in units of 4-column groups' bytes
  const unsigned char* yg = reinterpret_cast<const unsigned char*>(e.dy) + (long)(ycol_ok ? n0 + 4 * yc4 : 0) * (YBF ? 2 : 4);
  const unsigned char* xg = reinterpret_cast<const unsigned char*>(e.x) + (long)(xcol_ok ? k0 + 4 * xc4 : 0) * (XBF ? 2 : 4);
  const long ldy = (long)e.ld_dy * (YBF ? 2 : 4), ldx = (long)e.ld_x * (XBF ? 2 : 4);
  // LDS store offsets (row-dependent part added per u): chunk = column / 8, 8-byte half = (column / 4) & 1
  const int y_half = (yc4 >> 5), y_ch = (yc4 & 31) >> 1, y_hb = 8 * (yc4 & 1);
  const int x_ch = xc4 >> 1, x_hb = 8 * (xc4 & 1);

  float4 bsum = make_float4(0.f, 0.f, 0.f, 0.f);
  const bool bias = e.db != nullptr && bk == 0;

  // UNCONDITIONAL loads at clamped addresses, zeroed by a select afterwards: a predicated load is a branch, and behind
  // control flow the compiler can no longer count outstanding loads -- every use then waits with vmcnt(0), i.e. for
  // the loads issued a moment ago as well, and the ring hides nothing (measured: 1.5 us per step either way).
  YQ zero_y; quad_zero(zero_y);
  XQ zero_x; quad_zero(zero_x);
  auto gload = [&](StagedT& r, int step) {
    const int m0 = m_lo + step * WT_MS;
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int m = m0 + yr + 8 * u;
      const YQ v = *reinterpret_cast<const YQ*>(yg + (long)min(m, m_hi - 1) * ldy);
      r.y[u] = (ycol_ok && m < m_hi) ? v : zero_y;
    }
#pragma unroll
    for (int u = 0; u < 2; ++u) {
      const int m = m0 + xr + 16 * u;
      const XQ v = *reinterpret_cast<const XQ*>(xg + (long)min(m, m_hi - 1) * ldx);
      r.x[u] = (xcol_ok && m < m_hi) ? v : zero_x;
    }
  };
  auto lstore = [&](const StagedT& r, int stage) {
    unsigned char* base = lds + stage * 3 * WT_HALF;
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      *reinterpret_cast<bf16x4*>(base + y_half * WT_HALF + img_off(yr + 8 * u, y_ch) + y_hb) = quad_bf16(r.y[u]);
      if (bias) {
        const float4 f = quad_f32(r.y[u]);
        bsum.x += f.x; bsum.y += f.y; bsum.z += f.z; bsum.w += f.w;
      }
    }
#pragma unroll
    for (int u = 0; u < 2; ++u)
      *reinterpret_cast<bf16x4*>(base + 2 * WT_HALF + img_off(xr + 16 * u, x_ch) + x_hb) = quad_bf16(r.x[u]);
  };

  f32x4 acc[4][4];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
  const bool active = n0 + 64 * wn < N && k0 + 64 * wk < K;  // (wave-uniform; every wave still stages and syncs)

  // ring of register sets: set s % 3 holds the operands of step s from the moment they are requested (three steps
  // ahead) until they are written to LDS stage s & 1 at the bottom of step s - 1
  StagedT ring[WT_RING];
#pragma unroll
  for (int r = 0; r < WT_RING; ++r) gload(ring[r], r);
  lstore(ring[0], 0);
  __syncthreads();
  for (int s0 = 0; s0 < nsteps; s0 += WT_RING) {
#pragma unroll
    for (int r = 0; r < WT_RING; ++r) {
      const int s = s0 + r;
      if (s < nsteps) {  // (workgroup-uniform)
        WT_MARK(0);
        gload(ring[r], s + WT_RING);  // set r's previous content (step s) went to LDS at the bottom of step s - 1
        WT_MARK(1);
        {
          const unsigned char* base = lds + (s & 1) * 3 * WT_HALF;
          const unsigned char* yimg = base + (wn >> 1) * WT_HALF;  // this wave's 64 dY columns: tiles 4 (wn & 1) .. of half wn / 2
          const unsigned char* ximg = base + 2 * WT_HALF;
          bf16x8 bf[4];
#pragma unroll
          for (int j = 0; j < 4; ++j) bf[j] = tr_frag(ximg, 4 * wk + j, lane);
#pragma unroll
          for (int i = 0; i < 4; ++i) {
            const bf16x8 af = tr_frag(yimg, 4 * (wn & 1) + i, lane);  // (all lanes of all waves read: EXEC all ones)
            if (active) {
#pragma unroll
              for (int j = 0; j < 4; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af, bf[j], acc[i][j], 0, 0, 0);
            }
          }
        }
        WT_MARK(2);
        if (s + 1 < nsteps) lstore(ring[(r + 1) % WT_RING], (s + 1) & 1);
        WT_MARK(3);
        __syncthreads();
        WT_MARK(4);
      }
    }
  }
  // ---- dW block ----
  const bool plain = e.exclusive && e.splits == 1;
  if (plain) {
    // plain stores: whole 256-B rows.  A 4-B store from the accumulator layout covers 64 B of a row; the other half of
    // the 128-B line arrives with another instruction and the L2 has to merge (or fill) partial lines -- measured: the
    // 288 MB of the GPS backbone's gradients took the same ~230 us with three different product kernels.  Each wave
    // turns 16 rows x 64 columns at a time through a private 4-KB LDS patch (the staging images are dead: the loop
    // ended on a barrier) and stores 16 B per lane, four full rows per instruction.
    float* patch = reinterpret_cast<float*>(lds) + wave * 1024;
    if (active) {
#pragma unroll
      for (int i = 0; i < 4; ++i) {
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
          for (int r = 0; r < 4; ++r) patch[(4 * fq + r) * 64 + 16 * j + fr] = acc[i][j][r];
        __builtin_amdgcn_wave_barrier();
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
        for (int u = 0; u < 4; ++u) {
          const int row = (lane >> 4) + 4 * u, c4 = (lane & 15) * 4;
          const int n = n0 + 64 * wn + 16 * i + row, k = k0 + 64 * wk + c4;
          if (n < N && k < K) *reinterpret_cast<float4*>(e.dw + (long)n * K + k) = *reinterpret_cast<const float4*>(patch + row * 64 + c4);
        }
        __builtin_amdgcn_wave_barrier();
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      }
    }
  } else if (active) {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int nb = n0 + 64 * wn + 16 * i + 4 * fq;
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const int k = k0 + 64 * wk + 16 * j + fr;
        if (k < K) {
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            if (nb + r < N) {
              atomicAdd(e.dw + (long)(nb + r) * K + k, acc[i][j][r]);
            }
          }
        }
      }
    }
  }
  // ---- bias gradient: the 8 threads that staged the same 4 columns (one per wave) meet in LDS ----
  if (bias) {
    float* red = reinterpret_cast<float*>(lds + 2 * 3 * WT_HALF);
    *reinterpret_cast<float4*>(red + (wave * 64 + yc4) * 4) = bsum;
    __syncthreads();
    if (tid < 256) {
      const int c4 = tid >> 2, comp = tid & 3;
      float s = 0.f;
#pragma unroll
      for (int w = 0; w < 8; ++w) s += red[(w * 64 + c4) * 4 + comp];
      const int n = n0 + tid;
      if (n < N) atomicAdd(e.db + n, s);
    }
  }
}

__global__ __launch_bounds__(WT_NT) void wgrad_tr_kernel(const TrTable t) {
  // buffers: [2 stages][dY half 0 | dY half 1 | X] images of 8 KB each; then the bias reduction scratch
  __shared__ __attribute__((aligned(16))) unsigned char lds[2 * 3 * WT_HALF + 8 * 256 * 4];
  // XCD-aware block order (measurement switch, off: see rf_wgrad_tr).  Workgroups are dealt to the 8 XCDs round-robin by
  // blockIdx, each XCD with its own L2; in plain order the gk blocks that share a 256-column dY panel (and the gn blocks that share an X panel) are spread over all eight,
  // and every one of them fetches the panel from HBM: PMC showed 452 MB fetched by the GPS backbone's group for ~90 MB of
  // operands.  With t.run > 0 an XCD takes RUNS of t.run consecutive blocks (one after the other in its own dispatch order):
  // the blocks of a run share their dY panel, consecutive runs of an XCD mostly belong to the same weight, whose X panels
  // (K / 128 x 287 KB at M = 560) stay in its 4-MB L2.  Runs -- not one contiguous eighth per XCD -- keep the XCDs balanced:
  // the problems differ 6 x in reduction depth.  (The grid is padded to a multiple of 8 runs; blocks past the end exit.)
  int b = blockIdx.x;
  if (t.run > 0) {
    const int x = b & 7, j = b >> 3;  // j-th block of XCD x
    b = (j / t.run) * (8 * t.run) + x * t.run + j % t.run;
    if (b >= t.first_block[t.count]) return;
  }
  int lo = 0, hi = t.count - 1;
  while (lo < hi) {
    const int mid = (lo + hi + 1) >> 1;
    if (t.first_block[mid] <= b) lo = mid; else hi = mid - 1;
  }
  if (t.e[lo].dy_bf16) wgrad_tr_block<true, true>(t, lo, lds, b);  // (rf_wgrad_tr: both operands bf16, or both fp32)
  else wgrad_tr_block<false, false>(t, lo, lds, b);
}

}  // namespace

#ifdef RF_WT_TIMING
extern "C" void* rf_wt_timing_address() {
  void* a = nullptr;
  (void)hipGetSymbolAddress(&a, HIP_SYMBOL(rf_wt_timing));
  return a;
}
#endif

// dW / db of up to RF_WGRAD_MAX_GROUP problems in one launch (bf16 matrix-core path of rf_wgrad_grouped).
extern "C" int rf_wgrad_tr(const RfWgradEntry* entries, int count, void* stream) {
  RF_REQUIRE(entries && count >= 1 && count <= RF_WGRAD_MAX_GROUP);
  TrTable t{};
  t.count = count;
  int blocks = 0;
  for (int i = 0; i < count; ++i) {
    RfWgradEntry e = entries[i];
    RF_REQUIRE(e.dy && e.x && e.dw && e.M > 0 && e.N > 0 && e.K > 0 && e.splits >= 1);
    RF_REQUIRE((reinterpret_cast<uintptr_t>(e.dy) & 15) == 0 && (reinterpret_cast<uintptr_t>(e.x) & 15) == 0 &&
               e.ld_dy % 4 == 0 && e.ld_x % 4 == 0 && e.N % 4 == 0 && e.K % 4 == 0);
    RF_REQUIRE((e.dy_bf16 != 0) == (e.x_bf16 != 0));  // both operands of a problem in the same storage type
    // chunks of the reduction: at least 1 024 rows each (whole 32-row steps), never more than the caller asked for
    int chunks = (e.M + 1023) / 1024;
    if (chunks > e.splits) chunks = e.splits;
    if (chunks < 1) chunks = 1;
    e.kchunk = (((e.M + chunks - 1) / chunks + WT_MS - 1) / WT_MS) * WT_MS;
    e.splits = (e.M + e.kchunk - 1) / e.kchunk;
    t.e[i] = e;
    t.first_block[i] = blocks;
    blocks += ((e.N + WT_BN - 1) / WT_BN) * ((e.K + WT_BK - 1) / WT_BK) * e.splits;
  }
  t.first_block[count] = blocks;
  // (measured: 95.8 -> 98.8 us per launch with runs of 16, step unchanged -- the re-fetched panels come out of the memory-side
  //  cache, not HBM; the plain order stays.  RF_WGRAD_XCD_RUN = 16 turns the run order on.)
  static const int run = [] { const char* e = getenv("RF_WGRAD_XCD_RUN"); return e ? atoi(e) : 0; }();
  t.run = blocks >= 64 ? run : 0;
  if (t.run > 0) blocks = (blocks + 8 * t.run - 1) / (8 * t.run) * (8 * t.run);
  RF_LAUNCH(wgrad_tr_kernel, dim3(blocks), dim3(WT_NT), 0, static_cast<hipStream_t>(stream), t);
  RF_CHECK_LAUNCH();
  return RF_OK;
}
