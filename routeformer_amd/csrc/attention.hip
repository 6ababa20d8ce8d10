// Attention for the Routeformer hot path: full softmax attention and Informer "ProbSparse"
// attention (unmasked / masked), forward and backward.  One workgroup (256 threads = 4 waves) per
// (batch, head); the whole K, V (and Q) head slice lives in LDS (L <= 320, E <= 128: <= 160 KB).
//
// ProbSparse forward (cross_modal_transformer.py:88-166, restated in SURVEY.md A.3):
//   1. M[q] = max_j Q[q].K[idx[q,j]] - (sum_j Q[q].K[idx[q,j]]) / L_K      (fp32, idx from the host RNG)
//   2. top = the n_top queries with the largest M (ties: lower index first)
//   3. selected rows:  ctx[q] = softmax(scale * Q[q] K^T (masked: keys <= q)) V
//   4. other rows:     ctx[q] = mean_s V[s]  (unmasked)   |   cumsum_{s<=q} V[s]  (masked)
// Sequence lengths here are tiny, so the kernels are latency-bound by design;
// rows are spread over lanes and reductions use wave shuffles.
#include <cstdlib>

#include "common.h"
#include "philox.h"

namespace {

// Threads per workgroup are chosen at launch (threads_for): these kernels are chains of short barrier-
// separated phases, so when a launch has fewer (batch, head) problems than CUs (the Informer / gaze /
// fusion layers: 64 workgroups) each problem gets 8 waves instead of 4 to shorten every phase (16 until round 3).
inline int threads_for(int problems) {
  // RF_ATTN_SMALL_THREADS: threads of a launch with <= 128 problems (measurement switch).  8 waves since round 3: with the
  // whole-score-matrix form limited to 64 KB (below) the C2 step takes 5.40-5.41 ms against 5.44-5.45 with 16 waves and a
  // 160-KB limit, four alternating pairs on one box (gpurun_out/r6m); most of that is the form, 8 vs 16 waves alone is
  // 0.00-0.03 ms (r6j, r6k, r6w2); 4 waves: 5.67 (r6i)
  static const int small = [] { const char* e = getenv("RF_ATTN_SMALL_THREADS"); const int v = e ? atoi(e) : 512;
                                return (v == 256 || v == 512 || v == 1024) ? v : 512; }();
  return problems <= 128 ? small : (problems <= 512 ? 512 : 256);
}

struct AttnP {
  const float *q, *k, *v;
  long q_ld, k_ld, v_ld;
  float* ctx;
  const float* dctx;
  int dctx_slabs;    // > 1: dctx is the first of that many split-K slabs, dctx_slab elements apart, summed on load
  long dctx_slab;    //      (rf_attn_bwd_slabs: the out-projection's input gradient without its slab-sum launch)
  int out_layout;
  const int32_t* idx;
  int32_t* top;
  int force_top;
  float *dq, *dk, *dv;
  long dq_ld, dk_ld, dv_ld;
  int B, H, LQ, LK, E, sample_k, n_top, mode, idx_group;
  long idx_stride;  // elements between the key-sample tables of consecutive groups
  int Qs_rows;  // rows of Q staged by load_qkv (LQ in forward, 0 in backward: only the selected rows are needed)
  float scale;
  DropCfg drop;  // nn.Dropout on the attention probabilities of FullAttention (cross_modal_transformer.py:63)
};

// Dropout on the probabilities of the active rows (row si = query top_list[si]): element index of A[b,h,q,s] in the
// reference's (B,H,L_Q,L_K) probability tensor -- forward and backward regenerate the same keep-bits from it.
// Row pitch (floats) of the score / probability matrices S, P, dS.  Their rows are read as fp32 MFMA A fragments (lane =
// row, 4 k per quad: conflict-free when pitch * row mod 64 walks distinct multiples of 4), written in the accumulator layout
// (rows 4 apart per lane group), walked four rows at a time by the softmax (16 consecutive columns each) and read transposed
// (column = lane, rows one pitch apart per quad).  A pitch of LK itself is the worst case for the sizes that matter -- 160
// and 320 keys put every other / every row on the same banks: 8- and 16-way conflicts on the P V operand reads, 4-way on the
// score stores (PMC: 0.42 / 0.50 of the LDS cycles of attn_fwd / attn_bwd).  pitch mod 64 in {20, 44} serves all four
// patterns (row offsets 0, 20, 40, 60: at most a 12-bank overlap between four 16-column row pieces).
__host__ __device__ inline int score_pitch(int lkp) {
#ifdef RF_ATTN_NOPAD  // (A/B build: RF_HIP_LIB)
  return lkp;
#endif
  int p = lkp;
  while ((p & 63) != 20 && (p & 63) != 44) p += 4;
  return p;
}

// d ctx element(s) at `ptr`, summed over the split-K slabs when it arrives that way (AttnP::dctx_slabs)
__device__ __forceinline__ float ld_dctx(const AttnP& p, const float* ptr) {
  float a = *ptr;
  for (int s = 1; s < p.dctx_slabs; ++s) a += ptr[(long)s * p.dctx_slab];
  return a;
}
__device__ __forceinline__ float4 ld_dctx4(const AttnP& p, const float* ptr) {
  float4 a = *reinterpret_cast<const float4*>(ptr);
  for (int s = 1; s < p.dctx_slabs; ++s) {
    const float4 b = *reinterpret_cast<const float4*>(ptr + (long)s * p.dctx_slab);
    a.x += b.x; a.y += b.y; a.z += b.z; a.w += b.w;
  }
  return a;
}

__device__ __forceinline__ void drop_rows(float* S, int n_rows, int LK, int ld, const int* top_list, const DropGen& gen,
                                          long row_base) {
  for (int i = threadIdx.x; i < n_rows * LK; i += (int)blockDim.x) {
    const int si = i / LK, s_ = i - si * LK;
    S[(long)si * ld + s_] *= gen.factor((unsigned long long)((row_base + top_list[si]) * LK + s_));
  }
}

// XCD-aware problem order.  Workgroups are dealt round-robin to the 8 XCDs (blockIdx % 8), each with its
// own L2, and the H heads of one sequence read interleaved 4E-byte pieces of the same packed rows.  With
// the plain (b, h) = (blk / H, blk % H) order the heads of a sequence land on different XCDs and every
// 128-B line is fetched by two of them; here one XCD walks all heads of a sequence back to back.
__device__ __forceinline__ void problem_of(int blk, int B, int H, int& b, int& h) {
  if ((B & 7) == 0) {
    const int xcd = blk & 7, j = blk >> 3;
    b = (j / H) * 8 + xcd;
    h = j % H;
  } else {
    b = blk / H;
    h = blk % H;
  }
}

__device__ __forceinline__ long ctx_off(const AttnP& p, int b, int h, int l) {
  return p.out_layout == 0 ? (((long)b * p.LQ + l) * p.H + h) * p.E : (((long)b * p.H + h) * p.LQ + l) * p.E;
}

// V4 = every head row is 16-B addressable (E % 4 == 0, aligned pointers / pitches): LDS rows get pitch E+4
// and all row traffic is 128-bit (4x fewer LDS instructions, the limiter of these kernels); else pitch E+1
// and scalar accesses.  Dot products keep the same ascending-e fmaf chain in both forms.
template <bool V4> __device__ __forceinline__ int pitch(int E) { return V4 ? E + 4 : E + 1; }

template <bool V4>
__device__ __forceinline__ float dot_rows(const float* __restrict__ a, const float* __restrict__ b, int E) {
  float d = 0.f;
  if constexpr (V4) {
    for (int e = 0; e < E; e += 4) {
      const float4 x = *reinterpret_cast<const float4*>(a + e), y = *reinterpret_cast<const float4*>(b + e);
      d = fmaf(x.x, y.x, d); d = fmaf(x.y, y.y, d); d = fmaf(x.z, y.z, d); d = fmaf(x.w, y.w, d);
    }
  } else {
    for (int e = 0; e < E; ++e) d = fmaf(a[e], b[e], d);
  }
  return d;
}

// Load an (L x E) head slice into LDS with row pitch EP.
template <bool V4>
__device__ __forceinline__ void load_head(float* S, const float* G, long ld, int b, int h, int L, int E,
                                          int EP, int tid) {
  const float* base = G + (long)b * L * ld + (long)h * E;
  if constexpr (V4) {
    const int E4 = E >> 2;
    for (int i = tid; i < L * E4; i += (int)blockDim.x) {
      const int l = i / E4, e = (i - l * E4) << 2;
      *reinterpret_cast<float4*>(S + l * EP + e) = *reinterpret_cast<const float4*>(base + (long)l * ld + e);
    }
  } else {
    for (int i = tid; i < L * E; i += (int)blockDim.x) {
      const int l = i / E, e = i - l * E;
      S[l * EP + e] = base[(long)l * ld + e];
    }
  }
}

// (row, col) of a flattened index that advances by the workgroup size: two integer divisions when the
// iterator is built instead of one per trip (integer division is ~25 VALU instructions on CDNA and these
// kernels are instruction-bound: ~1.7k VALU instructions per wave measured)
struct RowCol {
  int r, c, dr, dc, w;
  __device__ __forceinline__ RowCol(int start, int step, int width) : w(width) {
    r = start / width; c = start - r * width;
    dr = step / width; dc = step - dr * width;
  }
  __device__ __forceinline__ void next() {
    r += dr; c += dc;
    if (c >= w) { c -= w; ++r; }
  }
};

// Q, K and V head slices in one go: every global load of the three slices is issued before the first LDS
// store (one memory round trip for the whole prologue instead of one per slice and loop trip).  TRIPS =
// loads per thread and slice, a launch-uniform count: the body is branch-free (a branch around a load makes
// the compiler drain the memory counter at the join, which serialises the round trips -- measured 2.3x on this
// prologue), and an all-clamped trip would still cost the CU's address unit 16 cycles per wave and slice.
template <int TRIPS, bool HAS_Q>
__device__ __forceinline__ void load_qkv_trips(float* Qs, float* Ks, float* Vs, const AttnP& p, int b, int h, int EP,
                                               int tid, int nq, int nk) {
  const int nt = blockDim.x, E4 = p.E >> 2;
  const float* qb = p.q + (long)b * p.LQ * p.q_ld + (long)h * p.E;
  const float* kb = p.k + (long)b * p.LK * p.k_ld + (long)h * p.E;
  const float* vb = p.v + (long)b * p.LK * p.v_ld + (long)h * p.E;
  float4 rq[TRIPS], rk[TRIPS], rv[TRIPS];
  int ls[TRIPS], es[TRIPS];
  RowCol it(tid, nt, E4);
#pragma unroll
  for (int u = 0; u < TRIPS; ++u) {
    const int l = it.r, e = it.c << 2;
    ls[u] = l; es[u] = e;
    it.next();
    // unconditional loads at clamped rows (a predicated load is an exec-mask branch); the stores are guarded
    const int lk = min(l, p.LK - 1);
    if constexpr (HAS_Q) rq[u] = *reinterpret_cast<const float4*>(qb + (long)min(l, p.Qs_rows - 1) * p.q_ld + e);
    rk[u] = *reinterpret_cast<const float4*>(kb + (long)lk * p.k_ld + e);
    rv[u] = *reinterpret_cast<const float4*>(vb + (long)lk * p.v_ld + e);
  }
  // Pin every loaded register HERE.  Without it the compiler sinks each load into the guarded store that consumes it
  // (legal, the pointers are restrict) and the prologue becomes a chain of load -> wait -> ds_write blocks, with one
  // slice staged through scratch: five dependent memory round trips at TRIPS = 2 (7.2 k of the 26 k cycles of a
  // GPS-backbone attention launch, tools/attn_phase_probe.py) instead of one.
#pragma unroll
  for (int u = 0; u < TRIPS; ++u) {
    if constexpr (HAS_Q) asm volatile("" : "+v"(rq[u].x), "+v"(rq[u].y), "+v"(rq[u].z), "+v"(rq[u].w));
    asm volatile("" : "+v"(rk[u].x), "+v"(rk[u].y), "+v"(rk[u].z), "+v"(rk[u].w));
    asm volatile("" : "+v"(rv[u].x), "+v"(rv[u].y), "+v"(rv[u].z), "+v"(rv[u].w));
  }
#pragma unroll
  for (int u = 0; u < TRIPS; ++u) {
    const int i = tid + u * nt, l = ls[u], e = es[u];
    if constexpr (HAS_Q)
      if (i < nq) *reinterpret_cast<float4*>(Qs + l * EP + e) = rq[u];
    if (i < nk) {
      *reinterpret_cast<float4*>(Ks + l * EP + e) = rk[u];
      *reinterpret_cast<float4*>(Vs + l * EP + e) = rv[u];
    }
  }
}

template <bool V4, bool HAS_Q>
__device__ __forceinline__ void load_qkv(float* Qs, float* Ks, float* Vs, const AttnP& p, int b, int h, int EP, int tid) {
  const int nt = blockDim.x;
  if constexpr (V4) {
    const int E4 = p.E >> 2, nq = HAS_Q ? p.Qs_rows * E4 : 0, nk = p.LK * E4;
    const int trips = (max(nq, nk) + nt - 1) / nt;
    if (trips == 1) { load_qkv_trips<1, HAS_Q>(Qs, Ks, Vs, p, b, h, EP, tid, nq, nk); return; }
    if (trips == 2) { load_qkv_trips<2, HAS_Q>(Qs, Ks, Vs, p, b, h, EP, tid, nq, nk); return; }
    if (trips == 3) { load_qkv_trips<3, HAS_Q>(Qs, Ks, Vs, p, b, h, EP, tid, nq, nk); return; }
  }
  if constexpr (HAS_Q) load_head<V4>(Qs, p.q, p.q_ld, b, h, p.LQ, p.E, EP, tid);
  load_head<V4>(Ks, p.k, p.k_ld, b, h, p.LK, p.E, EP, tid);
  load_head<V4>(Vs, p.v, p.v_ld, b, h, p.LK, p.E, EP, tid);
}

__device__ __forceinline__ int quad_sum(int v) {  // all-reduce over each aligned group of 4 lanes (DPP)
  v += __builtin_amdgcn_update_dpp(0, v, 0xB1, 0xF, 0xF, true);
  v += __builtin_amdgcn_update_dpp(0, v, 0x4E, 0xF, 0xF, true);
  return v;
}

// Sparsity measure M[q] = max_j s(q,j) - sum_j s(q,j) / LK over the sampled scores of query q; four lanes
// share one query.  gather == nullptr: s(q,j) = S[q*sample_k + j]; else s(q,j) = S[q*ld + gather[q*sample_k + j]].
__device__ __forceinline__ void sparsity_measure(const float* S, int ld, const int* gather, float* Ms, int LQ,
                                                 int sample_k, int LK, int tid) {
  const int quads = blockDim.x >> 2, sub = tid & 3;
  for (int q0 = 0; q0 < LQ; q0 += quads) {
    const int q = q0 + (tid >> 2), qq = min(q, LQ - 1);
    float mx = -INFINITY, sm = 0.f;
    for (int j = sub; j < sample_k; j += 4) {
      const float d = gather ? S[qq * ld + gather[qq * sample_k + j]] : S[qq * sample_k + j];
      mx = fmaxf(mx, d);
      sm += d;
    }
    mx = fmaxf(mx, dpp_move<0xB1>(mx));
    mx = fmaxf(mx, dpp_move<0x4E>(mx));
    sm += dpp_move<0xB1>(sm);
    sm += dpp_move<0x4E>(sm);
    if (q < LQ && sub == 0) Ms[q] = mx - sm / (float)LK;
  }
}

// Select the n_top rows of M (size LQ): sel[q] = position among the selected (ascending q) or -1.
__device__ void select_top(float* Ms, int* sel, int* top_list, int LQ, int n_top, int tid) {
  // four lanes share one query: each counts a quarter of the competitors, a quad DPP add combines them
  const int quads = blockDim.x >> 2, sub = tid & 3;
  for (int q0 = 0; q0 < LQ; q0 += quads) {
    const int q = q0 + (tid >> 2);
    int rank = 0;
    if (q < LQ) {
      const float mq = Ms[q];
      for (int o = sub; o < LQ; o += 4) {
        const float mo = Ms[o];
        rank += (mo > mq) || (mo == mq && o < q);
      }
    }
    rank = quad_sum(rank);
    if (q < LQ && sub == 0) sel[q] = rank < n_top ? 1 : 0;
  }
  __syncthreads();
  // positions go to a scratch array (Ms is dead now) so no thread reads a flag another one rewrites
  int* posbuf = reinterpret_cast<int*>(Ms);
  for (int q0 = 0; q0 < LQ; q0 += quads) {
    const int q = q0 + (tid >> 2);
    int before = 0;
    if (q < LQ)
      for (int o = sub; o < q; o += 4) before += sel[o];
    before = quad_sum(before);
    if (q < LQ && sub == 0) {
      const int pos = sel[q] ? before : -1;
      if (pos >= 0) top_list[pos] = q;
      posbuf[q] = pos;
    }
  }
  __syncthreads();
  for (int q = tid; q < LQ; q += (int)blockDim.x) sel[q] = posbuf[q];
  __syncthreads();
}

// Flattened work decomposition: every phase spreads (row, column) pairs over all 256 threads, so the
// serial depth per thread is ~(rows*cols/256) short dot products instead of whole rows.
//
// LDS carve (floats): Qs[LQ*EP] Ks[LK*EP] Vs[LK*EP] S[max(LQ*sample_k, n_sel*LK)] Ms[LQ] vmean[E]
//            | ints: sel[LQ] top[n_sel]
typedef float f32x4 __attribute__((ext_vector_type(4)));

// C(row, col) = sum_k A(row, k) * B(k, col) on the fp32 matrix cores (v_mfma_f32_16x16x4_f32: an exact,
// k-ordered fmaf chain, so results equal the scalar loops bit for bit).  16x16 output tiles go round-robin
// over the 4 waves.  pa(row) / pb(col) return the LDS address of element k = 0 of that row / column and
// ask / bsk the k strides; every address touched must be readable and finite (callers pad with zeros or
// clamp indices), rows / columns beyond the real extent are dropped by the store functor.
template <class PA, class PB, class FS>
__device__ __forceinline__ void mm_tiles(int TI, int TJ, int KS, int lane, int wave, PA pa, int ask, PB pb, int bsk,
                                         FS fs) {
  const int lr = lane & 15, lq = lane >> 4;
  for (int t = wave; t < TI * TJ; t += (int)(blockDim.x >> 6)) {
    const int ti = t / TJ, tj = t - ti * TJ;
    const float* ap = pa(16 * ti + lr) + lq * ask;
    const float* bp = pb(16 * tj + lr) + lq * bsk;
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
    for (int kk = 0; kk < KS; ++kk)
      acc = __builtin_amdgcn_mfma_f32_16x16x4f32(ap[4 * kk * ask], bp[4 * kk * bsk], acc, 0, 0, 0);
#pragma unroll
    for (int r = 0; r < 4; ++r) fs(16 * ti + 4 * lq + r, 16 * tj + lr, acc[r]);
  }
}

// Row softmax over S (row pitch ld >= LK; columns [kmax, ld) are set to 0).  Sixteen lanes per row (four
// rows per wave trip): the two reductions are 4-step DPP row reductions, and a 16..160-wide row keeps all
// of its lanes busy.  row_of() is the row -> lane-group mapping, shared with the backward's dS pass.
__device__ __forceinline__ int rows_per_trip() { return (blockDim.x >> 6) * 4; }
__device__ __forceinline__ int row_of(int trip_base) { return trip_base + (threadIdx.x >> 6) * 4 + ((threadIdx.x & 63) >> 4); }

// One row by its 16 lanes: row[s] = softmax over s < kmax of (row[s] * scale), zeros up to ldw.  Up to 64 columns the row
// stays in registers for all three passes (one LDS read and one write per element instead of three and two); wider rows
// are read in chunks of four elements per lane (clamped index, masked use).  Same ascending-s order of the sum as the
// plain loops.  Measured with tools/attn_phase_probe.py: 2 371 -> 1 637 cycles at L = 40, 5 236 -> 4 437 at L = 160.
// (These kernels are VALU-issue bound -- 16 waves of a lone workgroup share the CU's four SIMDs at 4 cycles per wave64
//  instruction -- so what pays is FEWER instructions: chunking the MFMA k-loops, the ranking and the sparsity measure the
//  same way issued their LDS reads together but added clamps and selects, and measured no faster or slower; not kept.)
__device__ __forceinline__ void softmax_row16(float* row, int kmax, int ldw, float scale, bool live) {
  const int l16 = threadIdx.x & 15;
  if (ldw <= 64) {
    float r[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) r[u] = row[min(l16 + 16 * u, ldw - 1)];
    float mx = -INFINITY;
#pragma unroll
    for (int u = 0; u < 4; ++u) mx = (l16 + 16 * u < kmax) ? fmaxf(mx, r[u]) : mx;
    mx = row16_max(mx);
    float sum = 0.f;
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const float e_ = __expf((r[u] - mx) * scale);
      r[u] = (l16 + 16 * u < kmax) ? e_ : 0.f;
      if (l16 + 16 * u < kmax) sum += e_;
    }
    sum = row16_sum(sum);
    const float inv = 1.f / sum;
    if (live) {
#pragma unroll
      for (int u = 0; u < 4; ++u)
        if (l16 + 16 * u < ldw) row[l16 + 16 * u] = r[u] * inv;
    }
    return;
  }
  float mx = -INFINITY;
  for (int s0 = l16; s0 < kmax; s0 += 64) {
    float r[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) r[u] = row[min(s0 + 16 * u, ldw - 1)];
#pragma unroll
    for (int u = 0; u < 4; ++u) mx = (s0 + 16 * u < kmax) ? fmaxf(mx, r[u]) : mx;
  }
  mx = row16_max(mx);
  float sum = 0.f;
  for (int s0 = l16; s0 < kmax; s0 += 64) {
    float r[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) r[u] = row[min(s0 + 16 * u, ldw - 1)];
#pragma unroll
    for (int u = 0; u < 4; ++u)
      if (s0 + 16 * u < kmax) {
        const float e_ = __expf((r[u] - mx) * scale);
        row[s0 + 16 * u] = e_;
        sum += e_;
      }
  }
  sum = row16_sum(sum);
  const float inv = 1.f / sum;
  if (live)
    for (int s0 = l16; s0 < ldw; s0 += 64) {
      float r[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) r[u] = row[min(s0 + 16 * u, ldw - 1)];
#pragma unroll
      for (int u = 0; u < 4; ++u)
        if (s0 + 16 * u < ldw) row[s0 + 16 * u] = (s0 + 16 * u < kmax) ? r[u] * inv : 0.f;
    }
}

__device__ __forceinline__ void softmax_rows(float* S, int n_rows, int LK, const int* top_list, int masked, int ld = 0,
                                             int row_pitch = 0) {
  if (ld == 0) ld = LK;
  if (row_pitch == 0) row_pitch = ld;
  for (int base = 0; base < n_rows; base += rows_per_trip()) {
    const int si = row_of(base);
    const bool live = si < n_rows;
    float* row = S + (long)(live ? si : 0) * row_pitch;
    const int kmax = live ? (masked ? top_list[si] + 1 : LK) : 0;
    softmax_row16(row, kmax, ld, 1.0f, live);
  }
}

// Phase timing aid (tools/attn_phase_probe.py builds a private copy with -DRF_ATTN_TIMING): thread 0 of every
// workgroup stamps the shader clock after each phase into rf_attn_timing (16 slots per workgroup).
#ifdef RF_ATTN_TIMING
__device__ unsigned long long rf_attn_timing[16 * 4096];
#define RF_MARK(k) do { if (threadIdx.x == 0 && blockIdx.x < 4096) rf_attn_timing[blockIdx.x * 16 + (k)] = __builtin_readcyclecounter(); } while (0)
#else
#define RF_MARK(k) do {} while (0)
#endif

// FULLS (chosen by rf_attn_fwd_full_scores: the whole score matrix fits next to Q, K, V): Q K^T is formed once
// on the matrix cores; the sampled scores of the sparsity measure are gathers from it and the rows of the selected
// queries are already there for the softmax -- the scalar sampling stage and the second score pass disappear, and
// so do four of the eleven block barriers.
template <bool V4, bool FULLS>
__global__ __launch_bounds__(1024) void attn_fwd_kernel(AttnP p) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  int b, h;
  problem_of(blockIdx.x, p.B, p.H, b, h);
  const int LQ = p.LQ, LK = p.LK, E = p.E, EP = pitch<V4>(E);
  const int n_sel = (p.mode == 0) ? LQ : p.n_top;
  const int LKP = V4 ? ((LK + 3) & ~3) : LK;  // key rows / score columns padded to the MFMA k granule
  const int SLD = V4 ? score_pitch(LKP) : LK; // row pitch of the score matrix (see score_pitch)
  const int s_elems = FULLS ? ((LQ * SLD + 3) & ~3) : ((max(LQ * p.sample_k, n_sel * SLD) + 3) & ~3);
  float* Qs = smem;
  float* Ks = Qs + LQ * EP;
  float* Vs = Ks + LKP * EP;
  float* S = Vs + LKP * EP;
  float* Ms = S + s_elems;
  float* vmean = Ms + LQ;
  int* sel = reinterpret_cast<int*>(vmean + E);
  int* top_list = sel + LQ;

  RF_MARK(0);
  // the key-sample table rides along with the head loads: its entries wait in the score buffer, each
  // later overwritten by the dot product it selects (same thread, same slot)
  const bool sampling = p.mode != 0 && !p.force_top;
  int* Sidx = FULLS ? top_list + n_sel : reinterpret_cast<int*>(S);  // FULLS: own region (S holds Q K^T)
  float* vpart = reinterpret_cast<float*>(Sidx + LQ * p.sample_k);     // FULLS: partial column sums of V
  if (sampling) {
    const int32_t* idx = p.idx + (long)(b / p.idx_group) * p.idx_stride;
    for (int i = tid; i < LQ * p.sample_k; i += (int)blockDim.x) Sidx[i] = idx[i];
  }
  load_qkv<V4, true>(Qs, Ks, Vs, p, b, h, EP, tid);
  for (int i = tid; i < (LKP - LK) * EP; i += (int)blockDim.x) { Ks[LK * EP + i] = 0.f; Vs[LK * EP + i] = 0.f; }
  __syncthreads();
  RF_MARK(1);

  if constexpr (FULLS) {
    // ---- (a) S = Q K^T (all rows), and the partial column sums of V for the lazy rows, one phase ----
    mm_tiles((LQ + 15) >> 4, (LK + 15) >> 4, E >> 2, lane, wave,
             [&](int q) { return Qs + min(q, LQ - 1) * EP; }, 1,
             [&](int s_) { return Ks + min(s_, LK - 1) * EP; }, 1,
             [&](int q, int s_, float v) {
               if (q < LQ && s_ < LKP) S[q * SLD + s_] = s_ < LK ? v : 0.f;
             });
    const int parts = max(1, min((int)blockDim.x / E, 16));
    if (p.mode == 1) {
      for (int i = tid; i < parts * E; i += (int)blockDim.x) {
        const int part = i / E, d = i - part * E;
        float a = 0.f;
        for (int l = part; l < LK; l += parts) a += Vs[l * EP + d];
        vpart[i] = a;
      }
    }
    __syncthreads();
    RF_MARK(2);
    // ---- (b) sparsity measure from the sampled entries of S; column means of V ----
    int32_t* gtop = p.top + ((long)b * p.H + h) * p.n_top;
    if (sampling) {
      sparsity_measure(S, SLD, Sidx, Ms, LQ, p.sample_k, LK, tid);
    } else {
      for (int q = tid; q < LQ; q += (int)blockDim.x) sel[q] = -1;
    }
    if (p.mode == 1) {
      for (int d = tid; d < E; d += (int)blockDim.x) {
        float a = 0.f;
        for (int part = 0; part < parts; ++part) a += vpart[part * E + d];
        vmean[d] = a / (float)LK;
      }
    }
    __syncthreads();
    RF_MARK(3);
    // ---- (c) top-u queries ----
    if (sampling) {
      select_top(Ms, sel, top_list, LQ, n_sel, tid);
      for (int i = tid; i < n_sel; i += (int)blockDim.x) gtop[i] = top_list[i];
    } else {
      for (int i = tid; i < n_sel; i += (int)blockDim.x) { top_list[i] = gtop[i]; sel[gtop[i]] = i; }
      __syncthreads();
    }
    RF_MARK(4);
    // ---- (d) lazy rows out; softmax of the selected rows in place (scale and mask folded in) ----
    if (p.mode == 1) {
      RowCol rc(tid, blockDim.x, E);
      const long row_step = p.out_layout == 0 ? (long)p.H * E : (long)E;
      float* base = p.ctx + ctx_off(p, b, h, 0);
      for (int i = tid; i < LQ * E; i += (int)blockDim.x, rc.next())
        if (sel[rc.r] < 0) base[rc.r * row_step + rc.c] = vmean[rc.c];
    } else {
      for (int d = tid; d < E; d += (int)blockDim.x) {
        float a = 0.f;
        for (int ql = 0; ql < LQ; ++ql) {
          a += Vs[ql * EP + d];
          if (sel[ql] < 0) p.ctx[ctx_off(p, b, h, ql) + d] = a;
        }
      }
    }
    RF_MARK(5);
    RF_MARK(6);
    for (int base_ = 0; base_ < n_sel; base_ += rows_per_trip()) {
      const int si = row_of(base_);
      const bool live = si < n_sel;
      const int q = live ? top_list[si] : 0;
      // scale > 0: the row maximum is the same before and after scaling
      softmax_row16(S + q * SLD, live ? (p.mode == 2 ? q + 1 : LK) : 0, LKP, p.scale, live);
    }
    __syncthreads();
    RF_MARK(7);
    // ---- (e) P V for the selected rows ----
    mm_tiles((n_sel + 15) >> 4, (E + 15) >> 4, LKP >> 2, lane, wave,
             [&](int si) { return S + top_list[min(si, n_sel - 1)] * SLD; }, 1,
             [&](int d) { return Vs + min(d, E - 1); }, EP,
             [&](int si, int d, float v) {
               if (si < n_sel && d < E) p.ctx[ctx_off(p, b, h, top_list[si]) + d] = v;
             });
    RF_MARK(8);
    return;
  }

  if (p.mode == 0) {
    for (int q = tid; q < LQ; q += (int)blockDim.x) { sel[q] = q; top_list[q] = q; }
    __syncthreads();
  } else {
    int32_t* gtop = p.top + ((long)b * p.H + h) * p.n_top;
    if (p.force_top) {
      for (int q = tid; q < LQ; q += (int)blockDim.x) sel[q] = -1;
      __syncthreads();
      for (int i = tid; i < n_sel; i += (int)blockDim.x) { top_list[i] = gtop[i]; sel[gtop[i]] = i; }
      __syncthreads();
    } else {
      // (1) sampled scores Q[q].K[idx[q,j]] (one table per group of `idx_group` consecutive batch rows)
      RowCol qs(tid, blockDim.x, p.sample_k);
      for (int i = tid; i < LQ * p.sample_k; i += (int)blockDim.x, qs.next())
        S[i] = dot_rows<V4>(Qs + qs.r * EP, Ks + Sidx[i] * EP, E);
      __syncthreads();
      RF_MARK(2);
      sparsity_measure(S, 0, nullptr, Ms, LQ, p.sample_k, LK, tid);
      __syncthreads();
      RF_MARK(3);
      // (2) top-u queries
      select_top(Ms, sel, top_list, LQ, n_sel, tid);
      RF_MARK(4);
      for (int i = tid; i < n_sel; i += (int)blockDim.x) gtop[i] = top_list[i];
    }
    // (4) lazy rows: mean(V) or cumsum(V)
    if (p.mode == 1) {
      // column means of V: `parts` threads per column, partial sums through the (currently dead) score buffer
      const int parts = min(min((int)blockDim.x / E, s_elems / E), 16);
      if (parts >= 2) {
        for (int i = tid; i < parts * E; i += (int)blockDim.x) {
          const int part = i / E, d = i - part * E;
          float s = 0.f;
          for (int l = part; l < LK; l += parts) s += Vs[l * EP + d];
          S[i] = s;
        }
        __syncthreads();
        for (int d = tid; d < E; d += (int)blockDim.x) {
          float s = 0.f;
          for (int part = 0; part < parts; ++part) s += S[part * E + d];
          vmean[d] = s / (float)LK;
        }
      } else {
        for (int d = tid; d < E; d += (int)blockDim.x) {
          float s = 0.f;
          for (int l = 0; l < LK; ++l) s += Vs[l * EP + d];
          vmean[d] = s / (float)LK;
        }
      }
      __syncthreads();
      {
        RowCol rc(tid, blockDim.x, E);
        const long row_step = p.out_layout == 0 ? (long)p.H * E : (long)E;
        float* base = p.ctx + ctx_off(p, b, h, 0);
        for (int i = tid; i < LQ * E; i += (int)blockDim.x, rc.next())
          if (sel[rc.r] < 0) base[rc.r * row_step + rc.c] = vmean[rc.c];
      }
    } else {
      for (int d = tid; d < E; d += (int)blockDim.x) {
        float s = 0.f;
        for (int ql = 0; ql < LQ; ++ql) {
          s += Vs[ql * EP + d];
          if (sel[ql] < 0) p.ctx[ctx_off(p, b, h, ql) + d] = s;
        }
      }
    }
    __syncthreads();  // S is re-used below
    RF_MARK(5);
  }

  // (3) active rows.  A: scores = scale * Qsel K^T, B: row softmax, C: P V
  if constexpr (V4) {
    const int TI = (n_sel + 15) >> 4;
    mm_tiles(TI, (LK + 15) >> 4, E >> 2, lane, wave,
             [&](int si) { return Qs + top_list[min(si, n_sel - 1)] * EP; }, 1,
             [&](int s_) { return Ks + min(s_, LK - 1) * EP; }, 1,
             [&](int si, int s_, float v) {
               if (si < n_sel && s_ < LK)
                 S[si * SLD + s_] = (p.mode == 2 && s_ > top_list[si]) ? -INFINITY : v * p.scale;
             });
    __syncthreads();
    RF_MARK(6);
    softmax_rows(S, n_sel, LK, top_list, p.mode == 2, LKP, SLD);
    __syncthreads();
    RF_MARK(7);
    {
      const DropGen gen(p.drop);
      if (gen.on()) {
        drop_rows(S, n_sel, LK, SLD, top_list, gen, ((long)b * p.H + h) * LQ);
        __syncthreads();
      }
    }
    mm_tiles(TI, (E + 15) >> 4, LKP >> 2, lane, wave,
             [&](int si) { return S + min(si, n_sel - 1) * SLD; }, 1,
             [&](int d) { return Vs + min(d, E - 1); }, EP,
             [&](int si, int d, float v) {
               if (si < n_sel && d < E) p.ctx[ctx_off(p, b, h, top_list[si]) + d] = v;
             });
    RF_MARK(8);
  } else {
    for (int i = tid; i < n_sel * LK; i += (int)blockDim.x) {
      const int si = i / LK, s_ = i - si * LK;
      const int q = top_list[si];
      float d = -INFINITY;
      if (p.mode != 2 || s_ <= q) d = dot_rows<V4>(Qs + q * EP, Ks + s_ * EP, E) * p.scale;
      S[i] = d;
    }
    __syncthreads();
    softmax_rows(S, n_sel, LK, top_list, p.mode == 2);
    __syncthreads();
    {
      const DropGen gen(p.drop);
      if (gen.on()) {
        drop_rows(S, n_sel, LK, LK, top_list, gen, ((long)b * p.H + h) * LQ);
        __syncthreads();
      }
    }
    for (int i = tid; i < n_sel * E; i += (int)blockDim.x) {
      const int si = i / E, d = i - si * E;
      const int q = top_list[si];
      const int kmax = (p.mode == 2) ? q + 1 : LK;
      const float* row = S + (long)si * LK;
      float a = 0.f;
      for (int s_ = 0; s_ < kmax; ++s_) a = fmaf(row[s_], Vs[s_ * EP + d], a);
      p.ctx[ctx_off(p, b, h, q) + d] = a;
    }
  }
}

// Backward.  LDS carve (floats): Ks[LK*EP] Vs[LK*EP] Qsel[n*EP] dCsel[n*EP] P[n*LK] dS[n*LK]
//            colsum[E] | ints: top[n] sel[LQ]
template <bool V4>
__global__ __launch_bounds__(1024) void attn_bwd_kernel(AttnP p) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  int b, h;
  problem_of(blockIdx.x, p.B, p.H, b, h);
  const int LQ = p.LQ, LK = p.LK, E = p.E, EP = pitch<V4>(E);
  const int n_sel = (p.mode == 0) ? LQ : p.n_top;
  const int LKP = V4 ? ((LK + 3) & ~3) : LK;      // padded key count (MFMA k granule)
  const int SLD = V4 ? score_pitch(LKP) : LK;     // row pitch of P / dS (see score_pitch)
  const int NSP = V4 ? ((n_sel + 3) & ~3) : n_sel;  // padded active-row count
  const int pl_elems = (NSP * SLD + 3) & ~3;
  float* Ks = smem;
  float* Vs = Ks + LKP * EP;
  float* Qsel = Vs + LKP * EP;
  float* dCsel = Qsel + NSP * EP;
  float* P = dCsel + NSP * EP;
  float* dS = P + pl_elems;
  float* colsum = dS + pl_elems;
  int* top_list = reinterpret_cast<int*>(colsum + ((E + 3) & ~3));
  int* sel = top_list + n_sel;

  RF_MARK(9);
  // the selection list rides along with the K/V loads (one memory round trip instead of two)
  const int32_t* gtop = p.mode != 0 ? p.top + ((long)b * p.H + h) * p.n_top : nullptr;
  const bool top_in_regs = n_sel <= (int)blockDim.x;
  int my_top = tid;
  if (p.mode != 0 && top_in_regs) my_top = gtop[min(tid, n_sel - 1)];
  load_qkv<V4, false>(nullptr, Ks, Vs, p, b, h, EP, tid);
  for (int i = tid; i < (LKP - LK) * EP; i += (int)blockDim.x) { Ks[LK * EP + i] = 0.f; Vs[LK * EP + i] = 0.f; }
  for (int i = tid; i < (NSP - n_sel) * EP; i += (int)blockDim.x) { Qsel[n_sel * EP + i] = 0.f; dCsel[n_sel * EP + i] = 0.f; }
  for (int i = tid; i < (NSP - n_sel) * SLD; i += (int)blockDim.x) { P[n_sel * SLD + i] = 0.f; dS[n_sel * SLD + i] = 0.f; }
  for (int q = tid; q < LQ; q += (int)blockDim.x) sel[q] = (p.mode == 0) ? q : -1;
  __syncthreads();
  if (p.mode == 0) {
    for (int i = tid; i < n_sel; i += (int)blockDim.x) top_list[i] = i;
  } else if (top_in_regs) {
    if (tid < n_sel) { top_list[tid] = my_top; sel[my_top] = tid; }
  } else {
    for (int i = tid; i < n_sel; i += (int)blockDim.x) { top_list[i] = gtop[i]; sel[gtop[i]] = i; }
  }
  __syncthreads();
  const float* qbase = p.q + (long)b * LQ * p.q_ld + (long)h * E;
  {
    const long row_step = p.out_layout == 0 ? (long)p.H * E : (long)E;
    const float* dbase = p.dctx + ctx_off(p, b, h, 0);
    if constexpr (V4) {
      const int E4 = E >> 2;
      RowCol rc(tid, blockDim.x, E4);
      for (int i = tid; i < n_sel * E4; i += (int)blockDim.x, rc.next()) {
        const int q = top_list[rc.r], e = rc.c << 2;
        const float4 a = *reinterpret_cast<const float4*>(qbase + (long)q * p.q_ld + e);
        const float4 d = ld_dctx4(p, dbase + q * row_step + e);
        *reinterpret_cast<float4*>(Qsel + rc.r * EP + e) = a;
        *reinterpret_cast<float4*>(dCsel + rc.r * EP + e) = d;
      }
    } else {
      RowCol rc(tid, blockDim.x, E);
      for (int i = tid; i < n_sel * E; i += (int)blockDim.x, rc.next()) {
        const int q = top_list[rc.r];
        Qsel[rc.r * EP + rc.c] = qbase[(long)q * p.q_ld + rc.c];
        dCsel[rc.r * EP + rc.c] = ld_dctx(p, dbase + q * row_step + rc.c);
      }
    }
  }
  __syncthreads();

  RF_MARK(10);
  // phase 1: recompute P = softmax(scale * Qsel K^T); dP = dC V^T; dS = P * (dP - rowsum(P*dP)) * scale
  if constexpr (V4) {
    const int TI = (n_sel + 15) >> 4, TJ = (LK + 15) >> 4;
    mm_tiles(TI, TJ, E >> 2, lane, wave,
             [&](int si) { return Qsel + min(si, n_sel - 1) * EP; }, 1,
             [&](int s_) { return Ks + min(s_, LK - 1) * EP; }, 1,
             [&](int si, int s_, float v) {
               if (si < n_sel && s_ < LK)
                 P[si * SLD + s_] = (p.mode == 2 && s_ > top_list[si]) ? -INFINITY : v * p.scale;
             });
    mm_tiles(TI, TJ, E >> 2, lane, wave,
             [&](int si) { return dCsel + min(si, n_sel - 1) * EP; }, 1,
             [&](int s_) { return Vs + min(s_, LK - 1) * EP; }, 1,
             [&](int si, int s_, float v) {
               if (si < n_sel && s_ < LK) dS[si * SLD + s_] = (p.mode == 2 && s_ > top_list[si]) ? 0.f : v;
             });
  } else {
    for (int i = tid; i < n_sel * LK; i += (int)blockDim.x) {
      const int si = i / LK, s_ = i - si * LK;
      const int q = top_list[si];
      float d = -INFINITY, dp = 0.f;
      if (p.mode != 2 || s_ <= q) {
        d = dot_rows<V4>(Qsel + si * EP, Ks + s_ * EP, E) * p.scale;
        dp = dot_rows<V4>(dCsel + si * EP, Vs + s_ * EP, E);
      }
      P[i] = d;
      dS[i] = dp;
    }
  }
  __syncthreads();
  RF_MARK(11);
  softmax_rows(P, n_sel, LK, top_list, p.mode == 2, LKP, SLD);
  // the same 16 lanes own the same row in softmax_rows and here: no barrier needed in between
  const DropGen gen(p.drop);
  for (int base = 0; base < n_sel; base += rows_per_trip()) {
    const int si = row_of(base), l16 = tid & 15;
    const bool live = si < n_sel;
    float* Pr = P + (long)(live ? si : 0) * SLD;
    float* dSr = dS + (long)(live ? si : 0) * SLD;
    // attention-probability dropout: ctx = (P * keep / (1-p)) V, so dP arrives masked and dV needs the masked P
    const unsigned long long e0 = (unsigned long long)((((long)b * p.H + h) * LQ + top_list[live ? si : 0]) * LK);
    float dot = 0.f;
    if (!gen.on() && LKP <= 64) {
      // the row pair stays in registers (four elements per lane, clamped reads): each element read once instead of twice;
      // same ascending-s order of the dot product
      float pr[4], ds[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const int s_ = min(l16 + 16 * u, LKP - 1);
        pr[u] = Pr[s_];
        ds[u] = dSr[s_];
      }
#pragma unroll
      for (int u = 0; u < 4; ++u)
        if (live && l16 + 16 * u < LK) dot += pr[u] * ds[u];
      dot = row16_sum(dot);
      if (live) {
#pragma unroll
        for (int u = 0; u < 4; ++u) {
          const int s_ = l16 + 16 * u;
          if (s_ < LKP) dSr[s_] = s_ < LK ? pr[u] * (ds[u] - dot) * p.scale : 0.f;
        }
      }
      continue;
    }
    if (live) {
      if (gen.on())
        for (int s_ = l16; s_ < LK; s_ += 16) dSr[s_] *= gen.factor(e0 + s_);
      for (int s_ = l16; s_ < LK; s_ += 16) dot += Pr[s_] * dSr[s_];
    }
    dot = row16_sum(dot);
    if (live) {
      for (int s_ = l16; s_ < LKP; s_ += 16) dSr[s_] = s_ < LK ? Pr[s_] * (dSr[s_] - dot) * p.scale : 0.f;
      if (gen.on())
        for (int s_ = l16; s_ < LK; s_ += 16) Pr[s_] *= gen.factor(e0 + s_);
    }
  }
  __syncthreads();

  RF_MARK(12);
  // dQ[q] = dS K for the active rows, zero for the others (the sampling stage is not differentiated)
  if constexpr (V4) {
    mm_tiles((n_sel + 15) >> 4, (E + 15) >> 4, LKP >> 2, lane, wave,
             [&](int si) { return dS + min(si, n_sel - 1) * SLD; }, 1,
             [&](int e) { return Ks + min(e, E - 1); }, EP,
             [&](int si, int e, float v) {
               if (si < n_sel && e < E) p.dq[((long)b * LQ + top_list[si]) * p.dq_ld + (long)h * E + e] = v;
             });
  } else {
    for (int i = tid; i < n_sel * E; i += (int)blockDim.x) {
      const int si = i / E, e = i - si * E;
      const int q = top_list[si];
      const int kmax = (p.mode == 2) ? q + 1 : LK;
      const float* dSr = dS + (long)si * SLD;
      float a = 0.f;
      for (int s_ = 0; s_ < kmax; ++s_) a = fmaf(dSr[s_], Ks[s_ * EP + e], a);
      p.dq[((long)b * LQ + q) * p.dq_ld + (long)h * E + e] = a;
    }
  }
  {
    RowCol rc(tid, blockDim.x, E);
    float* qb_ = p.dq + (long)b * LQ * p.dq_ld + (long)h * E;
    for (int i = tid; i < LQ * E; i += (int)blockDim.x, rc.next())
      if (sel[rc.r] < 0) qb_[(long)rc.r * p.dq_ld + rc.c] = 0.f;
  }
  // lazy-row gradient source: column sums of dctx over NON-selected rows (unmasked mode).  All threads take
  // part ((row part, column) work split, partials through the V tile, which is dead after phase 1 in this
  // mode): done by E threads walking all L_Q rows one dependent global load at a time, this was 40 % of the
  // whole backward kernel (tools/attn_phase_probe.py)
  if (p.mode == 1) {
    const int parts = max(1, min(min((int)blockDim.x / E, LK), 16));
    const long row_step = p.out_layout == 0 ? (long)p.H * E : (long)E;
    const float* dbase = p.dctx + ctx_off(p, b, h, 0);
    for (int i = tid; i < parts * E; i += (int)blockDim.x) {
      const int part = i / E, d = i - part * E;
      float s_ = 0.f;
#pragma unroll 4
      for (int ql = part; ql < LQ; ql += parts) {
        const float v = sel[ql] < 0 ? ld_dctx(p, dbase + ql * row_step + d) : 0.f;
        s_ += v;
      }
      Vs[i] = s_;
    }
    __syncthreads();
    for (int d = tid; d < E; d += (int)blockDim.x) {
      float s_ = 0.f;
      for (int part = 0; part < parts; ++part) s_ += Vs[part * E + d];
      colsum[d] = s_ / (float)LK;
    }
  }
  __syncthreads();

  RF_MARK(13);
  // phase 2: dK[s,e] = sum_q dS[q,s] Q[q,e];  dV[s,d] = sum_q P[q,s] dC[q,d] + lazy-row term
  if (p.mode == 2) {
    // masked lazy rows: ctx[q] = sum_{s<=q} V[s]  =>  dV[s] += sum_{q>=s, q not selected} dC[q]
    // (1) all threads fetch the lazy rows of dC into the (no longer needed) V tile, (2) E threads turn it into
    // the reverse running sum in LDS -- not E threads doing L_Q dependent global loads each
    {
      RowCol rc(tid, blockDim.x, E);
      const long row_step = p.out_layout == 0 ? (long)p.H * E : (long)E;
      const float* dbase = p.dctx + ctx_off(p, b, h, 0);
      for (int i = tid; i < LQ * E; i += (int)blockDim.x, rc.next())
        Vs[rc.r * EP + rc.c] = sel[rc.r] < 0 ? ld_dctx(p, dbase + rc.r * row_step + rc.c) : 0.f;
    }
    __syncthreads();
    for (int d = tid; d < E; d += (int)blockDim.x) {
      float run = 0.f;
      for (int ql = LQ - 1; ql >= 0; --ql) {
        run += Vs[ql * EP + d];
        Vs[ql * EP + d] = run;
      }
    }
    __syncthreads();
  }
  if constexpr (V4) {
    const int TJ = (LK + 15) >> 4, TE = (E + 15) >> 4;
    mm_tiles(TJ, TE, NSP >> 2, lane, wave,
             [&](int s_) { return dS + min(s_, LK - 1); }, SLD,
             [&](int e) { return Qsel + min(e, E - 1); }, EP,
             [&](int s_, int e, float v) {
               if (s_ < LK && e < E) p.dk[((long)b * LK + s_) * p.dk_ld + (long)h * E + e] = v;
             });
    mm_tiles(TJ, TE, NSP >> 2, lane, wave,
             [&](int s_) { return P + min(s_, LK - 1); }, SLD,
             [&](int d) { return dCsel + min(d, E - 1); }, EP,
             [&](int s_, int d, float v) {
               if (s_ < LK && d < E) {
                 if (p.mode == 1) v += colsum[d];
                 if (p.mode == 2) v += Vs[s_ * EP + d];
                 p.dv[((long)b * LK + s_) * p.dv_ld + (long)h * E + d] = v;
               }
             });
  } else {
    for (int i = tid; i < LK * E; i += (int)blockDim.x) {
      const int s_ = i / E, e = i - s_ * E;
      float ak = 0.f, av = 0.f;
      for (int si = 0; si < n_sel; ++si) {
        ak = fmaf(dS[(long)si * SLD + s_], Qsel[si * EP + e], ak);
        av = fmaf(P[(long)si * SLD + s_], dCsel[si * EP + e], av);
      }
      if (p.mode == 1) av += colsum[e];
      if (p.mode == 2) av += Vs[s_ * EP + e];
      p.dk[((long)b * LK + s_) * p.dk_ld + (long)h * E + e] = ak;
      p.dv[((long)b * LK + s_) * p.dv_ld + (long)h * E + e] = av;
    }
  }
  RF_MARK(14);
}

size_t fwd_lds(int LQ, int LK, int E, int n_sel, int sample_k, bool v4, bool fulls = false) {
  const size_t EP = v4 ? E + 4 : E + 1;
  const size_t LKP = v4 ? ((LK + 3) & ~3) : LK, SLD = v4 ? (size_t)score_pitch((int)LKP) : (size_t)LK;
  const size_t s_elems = fulls ? (((size_t)LQ * SLD + 3) & ~(size_t)3)
                               : ((max((size_t)LQ * sample_k, (size_t)n_sel * SLD) + 3) & ~(size_t)3);
  const size_t extra = fulls ? sizeof(int) * (size_t)LQ * sample_k + sizeof(float) * 16 * (size_t)E : 0;
  return sizeof(float) * (LQ * EP + 2 * LKP * EP + s_elems + LQ + E) + sizeof(int) * ((size_t)LQ + n_sel) + extra + 16;
}
size_t bwd_lds(int LQ, int LK, int E, int n_sel, bool v4) {
  const size_t EP = v4 ? E + 4 : E + 1;
  const size_t LKP = v4 ? ((LK + 3) & ~3) : LK, NSP = v4 ? ((n_sel + 3) & ~3) : n_sel;
  const size_t SLD = v4 ? (size_t)score_pitch((int)LKP) : (size_t)LK;
  const size_t pl = (NSP * SLD + 3) & ~(size_t)3;
  return sizeof(float) * (2 * LKP * EP + 2 * NSP * EP + 2 * pl + ((E + 3) & ~3)) + sizeof(int) * ((size_t)n_sel + LQ) + 16;
}
inline bool al16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; }

}  // namespace

extern "C" int rf_attn_fwd_full_scores(int B, int H, int LQ, int LK, int E, int sample_k, int n_top, int mode);

extern "C" int rf_attn_fwd_drop(const float* q, const float* k, const float* v, int64_t q_ld, int64_t k_ld,
                                int64_t v_ld, float* ctx, int out_layout, const int32_t* index_sample,
                                int idx_group, int64_t idx_group_stride, int32_t* top_idx, int force_top, int B, int H,
                                int LQ, int LK, int E, int sample_k, int n_top, int mode, float scale, float drop_p,
                                const void* rng_state, int drop_site, const uint8_t* drop_mask, void* stream) {
  RF_REQUIRE(q && k && v && ctx && B > 0 && H > 0 && LQ > 0 && LK > 0 && E > 0);
  RF_REQUIRE(mode >= 0 && mode <= 2);
  RF_REQUIRE(mode == 0 || (top_idx && n_top > 0 && n_top <= LQ));
  RF_REQUIRE(mode == 0 || force_top || (index_sample && sample_k > 0));
  RF_REQUIRE(mode != 2 || LQ == LK);
  const bool v4 = (E % 4 == 0) && al16(q) && al16(k) && al16(v) && al16(ctx) && q_ld % 4 == 0 && k_ld % 4 == 0 &&
                  v_ld % 4 == 0;
  const int threads = threads_for(B * H);
  RF_REQUIRE(drop_p >= 0.f && drop_p < 1.f && (drop_p == 0.f || rng_state || drop_mask));
  // attention-probability dropout exists for FullAttention only (mode 0, or the causal form = mode 2 with every row
  // imposed); ProbAttention defines a Dropout it never applies (cross_modal_transformer.py:86)
  RF_REQUIRE(drop_p == 0.f || mode == 0 || (mode == 2 && force_top && n_top == LQ));
  const bool fulls = v4 && drop_p == 0.f && rf_attn_fwd_full_scores(B, H, LQ, LK, E, sample_k, n_top, mode);
  const size_t lds = fwd_lds(LQ, LK, E, mode == 0 ? LQ : n_top, mode == 0 ? 0 : sample_k, v4, fulls);
  if (lds > 160 * 1024) { rf_g_last_error = "attention head slice exceeds 160 KB LDS"; return RF_EUNSUPPORTED; }
  AttnP p{};
  p.q = q; p.k = k; p.v = v; p.q_ld = q_ld; p.k_ld = k_ld; p.v_ld = v_ld; p.ctx = ctx;
  p.out_layout = out_layout; p.idx = index_sample; p.top = top_idx; p.force_top = force_top;
  p.B = B; p.H = H; p.LQ = LQ; p.LK = LK; p.E = E; p.sample_k = sample_k; p.n_top = n_top;
  p.mode = mode; p.scale = scale; p.Qs_rows = LQ;
  p.idx_group = (idx_group <= 0 || idx_group > B) ? B : idx_group;
  p.idx_stride = idx_group_stride > 0 ? idx_group_stride : (long)LQ * sample_k;
  p.drop = make_drop_cfg(rng_state, drop_mask, (uint32_t)drop_site, drop_p);
  static bool attr_set = false;
  if (!attr_set) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(attn_fwd_kernel<true, true>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(attn_fwd_kernel<true, false>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(attn_fwd_kernel<false, false>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    attr_set = true;
  }
  const hipStream_t st = static_cast<hipStream_t>(stream);
  if (fulls) RF_LAUNCH((attn_fwd_kernel<true, true>), dim3(B * H), dim3(threads), lds, st, p);
  else if (v4) RF_LAUNCH((attn_fwd_kernel<true, false>), dim3(B * H), dim3(threads), lds, st, p);
  else RF_LAUNCH((attn_fwd_kernel<false, false>), dim3(B * H), dim3(threads), lds, st, p);
  RF_CHECK_LAUNCH();
  return RF_OK;
}

extern "C" int rf_attn_fwd(const float* q, const float* k, const float* v, int64_t q_ld, int64_t k_ld,
                           int64_t v_ld, float* ctx, int out_layout, const int32_t* index_sample,
                           int idx_group, int64_t idx_group_stride, int32_t* top_idx, int force_top, int B, int H,
                           int LQ, int LK,
                           int E, int sample_k, int n_top, int mode, float scale, void* stream) {
  return rf_attn_fwd_drop(q, k, v, q_ld, k_ld, v_ld, ctx, out_layout, index_sample, idx_group, idx_group_stride, top_idx,
                          force_top, B, H, LQ, LK, E, sample_k, n_top, mode, scale, 0.f, nullptr, 0, nullptr, stream);
}

extern "C" int rf_attn_fwd_full_scores(int B, int H, int LQ, int LK, int E, int sample_k, int n_top, int mode) {
  if (E % 4 != 0 || mode == 0) return 0;
  const size_t full = fwd_lds(LQ, LK, E, n_top, sample_k, true, true);
  // a workgroup alone on its CU (<= 128 problems) may use the whole LDS; a chip-filling launch must keep four
  // workgroups per CU resident (at 48 KB -- three per CU -- the frame-encoder launch went from 53 to 75 us)
  // RF_ATTN_FULLS_KB: the LDS budget (KB) of the whole-score-matrix form in a launch of <= 128 problems (measurement switch).
  // At 64 KB none of the C2 step's 64-problem shapes takes it any more -- the fusion encoder (L 160, E 16: 25 600 scores formed
  // for 9 600 needed), the GPS decoder (L 70, E 104) and the GPS encoder (L 40, E 104: 68.9 KB) -- smaller heads still do.
  // A launch ALONE is faster in that form (13.4 vs 15.9 us at L 40, 20.1 vs 22.2 at L 160, tools/attn_phase_probe.py); the
  // replayed step, where these launches run next to the side-stream branches, is faster without it (5.40 vs 5.44 ms, four
  // alternating pairs, gpurun_out/r6m; 64 vs 72 KB, i.e. with / without the GPS encoder's: no difference, r6w)
  static const size_t small_kb = [] { const char* e = getenv("RF_ATTN_FULLS_KB"); const int v = e ? atoi(e) : 64;
                                      return (size_t)(v < 0 ? 0 : (v > 160 ? 160 : v)); }();
  return full <= (B * H <= 128 ? small_kb * 1024 : (size_t)32 * 1024);
}

#ifdef RF_ATTN_TIMING
extern "C" void* rf_attn_timing_address() {
  void* a = nullptr;
  (void)hipGetSymbolAddress(&a, HIP_SYMBOL(rf_attn_timing));
  return a;
}
#endif

static int attn_bwd_run(const float* q, const float* k, const float* v, int64_t q_ld, int64_t k_ld,
                        int64_t v_ld, const float* dctx, int out_layout, const int32_t* top_idx,
                        float* dq, float* dk, float* dv, int64_t dq_ld, int64_t dk_ld, int64_t dv_ld,
                        int B, int H, int LQ, int LK, int E, int n_top, int mode, float scale, float drop_p,
                        const void* rng_state, int drop_site, const uint8_t* drop_mask, void* stream, int dctx_slabs,
                        int64_t dctx_slab);

extern "C" int rf_attn_bwd_drop(const float* q, const float* k, const float* v, int64_t q_ld, int64_t k_ld,
                                int64_t v_ld, const float* dctx, int out_layout, const int32_t* top_idx,
                                float* dq, float* dk, float* dv, int64_t dq_ld, int64_t dk_ld, int64_t dv_ld,
                                int B, int H, int LQ, int LK, int E, int n_top, int mode, float scale, float drop_p,
                                const void* rng_state, int drop_site, const uint8_t* drop_mask, void* stream) {
  return attn_bwd_run(q, k, v, q_ld, k_ld, v_ld, dctx, out_layout, top_idx, dq, dk, dv, dq_ld, dk_ld, dv_ld, B, H, LQ, LK, E, n_top,
                      mode, scale, drop_p, rng_state, drop_site, drop_mask, stream, 0, 0);
}

// rf_attn_bwd whose d ctx is still `splits` split-K slabs, `slab_stride` elements apart (the out-projection's input gradient,
// rf_gemm_partials): summed on load.
extern "C" int rf_attn_bwd_slabs(const float* q, const float* k, const float* v, int64_t q_ld, int64_t k_ld,
                                 int64_t v_ld, const float* dctx_slabs, int splits, int64_t slab_stride, int out_layout,
                                 const int32_t* top_idx, float* dq, float* dk, float* dv, int64_t dq_ld, int64_t dk_ld,
                                 int64_t dv_ld, int B, int H, int LQ, int LK, int E, int n_top, int mode, float scale,
                                 void* stream) {
  RF_REQUIRE(splits >= 1 && splits <= 64 && slab_stride >= (int64_t)B * LQ * H * E && slab_stride % 4 == 0);
  return attn_bwd_run(q, k, v, q_ld, k_ld, v_ld, dctx_slabs, out_layout, top_idx, dq, dk, dv, dq_ld, dk_ld, dv_ld, B, H, LQ, LK, E,
                      n_top, mode, scale, 0.f, nullptr, 0, nullptr, stream, splits, slab_stride);
}

static int attn_bwd_run(const float* q, const float* k, const float* v, int64_t q_ld, int64_t k_ld,
                        int64_t v_ld, const float* dctx, int out_layout, const int32_t* top_idx,
                        float* dq, float* dk, float* dv, int64_t dq_ld, int64_t dk_ld, int64_t dv_ld,
                        int B, int H, int LQ, int LK, int E, int n_top, int mode, float scale, float drop_p,
                        const void* rng_state, int drop_site, const uint8_t* drop_mask, void* stream, int dctx_slabs,
                        int64_t dctx_slab) {
  RF_REQUIRE(q && k && v && dctx && dq && dk && dv && B > 0 && H > 0 && LQ > 0 && LK > 0 && E > 0);
  RF_REQUIRE(mode >= 0 && mode <= 2);
  RF_REQUIRE(mode == 0 || (top_idx && n_top > 0 && n_top <= LQ));
  RF_REQUIRE(mode != 2 || LQ == LK);
  const int n_sel = mode == 0 ? LQ : n_top;
  const bool v4 = (E % 4 == 0) && al16(q) && al16(k) && al16(v) && al16(dq) && al16(dk) && al16(dv) && al16(dctx) &&
                  q_ld % 4 == 0 && k_ld % 4 == 0 && v_ld % 4 == 0 && dq_ld % 4 == 0 && dk_ld % 4 == 0 && dv_ld % 4 == 0;
  const size_t lds = bwd_lds(LQ, LK, E, n_sel, v4);
  if (lds > 160 * 1024) { rf_g_last_error = "attention backward exceeds 160 KB LDS"; return RF_EUNSUPPORTED; }
  AttnP p{};
  p.q = q; p.k = k; p.v = v; p.q_ld = q_ld; p.k_ld = k_ld; p.v_ld = v_ld; p.dctx = dctx;
  p.dctx_slabs = dctx_slabs; p.dctx_slab = (long)dctx_slab;
  p.out_layout = out_layout; p.top = const_cast<int32_t*>(top_idx); p.dq = dq; p.dk = dk; p.dv = dv;
  p.dq_ld = dq_ld; p.dk_ld = dk_ld; p.dv_ld = dv_ld; p.B = B; p.H = H; p.LQ = LQ; p.LK = LK;
  p.E = E; p.n_top = n_top; p.mode = mode; p.scale = scale; p.Qs_rows = 0;
  RF_REQUIRE(drop_p >= 0.f && drop_p < 1.f && (drop_p == 0.f || rng_state || drop_mask));
  RF_REQUIRE(drop_p == 0.f || mode == 0 || (mode == 2 && n_top == LQ));
  p.drop = make_drop_cfg(rng_state, drop_mask, (uint32_t)drop_site, drop_p);
  static bool attr_set = false;
  if (!attr_set) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(attn_bwd_kernel<true>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(attn_bwd_kernel<false>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    attr_set = true;
  }
  if (v4) RF_LAUNCH(attn_bwd_kernel<true>, dim3(B * H), dim3(threads_for(B * H)), lds, static_cast<hipStream_t>(stream), p);
  else RF_LAUNCH(attn_bwd_kernel<false>, dim3(B * H), dim3(threads_for(B * H)), lds, static_cast<hipStream_t>(stream), p);
  RF_CHECK_LAUNCH();
  return RF_OK;
}

extern "C" int rf_attn_bwd(const float* q, const float* k, const float* v, int64_t q_ld, int64_t k_ld,
                           int64_t v_ld, const float* dctx, int out_layout, const int32_t* top_idx,
                           float* dq, float* dk, float* dv, int64_t dq_ld, int64_t dk_ld, int64_t dv_ld,
                           int B, int H, int LQ, int LK, int E, int n_top, int mode, float scale,
                           void* stream) {
  return rf_attn_bwd_drop(q, k, v, q_ld, k_ld, v_ld, dctx, out_layout, top_idx, dq, dk, dv, dq_ld, dk_ld, dv_ld, B, H, LQ,
                          LK, E, n_top, mode, scale, 0.f, nullptr, 0, nullptr, stream);
}
