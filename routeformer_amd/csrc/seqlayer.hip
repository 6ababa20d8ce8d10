// Fused per-sequence encoder stack (SURVEY section 7 step 5, 8(b) `encoder_layer_fused`): ONE workgroup owns ONE
// sequence (L <= 80 tokens x d_model 128) and walks every EncoderLayer of a PerceiveEncoder
// (cross_modal_transformer.py:288-301 x layers, called from routeformer.py:488 for 8 B sequences per camera stream):
//
//     QKV projection -> ProbSparse attention (8 heads) -> out-projection + residual + LayerNorm
//     -> conv1 -> GELU -> conv2 + residual + LayerNorm,            for every layer, in one launch.
//
// Layout of the work inside the workgroup (8 waves, two per SIMD):
//   * the fp32 residual stream never leaves REGISTERS: wave w holds columns 16w..16w+15 of every row in the MFMA
//     accumulator layout (the out-projection, conv2 and both LayerNorms produce exactly that slice);
//   * every GEMM operand A (x, ctx, x1, h) is a bf16 image in LDS, every weight B-fragment is read straight from a
//     fragment-ordered bf16 copy of the weights in global memory (rf_seqlayer_pack: the 64 lanes of a fragment read
//     one contiguous 1-KB block; the 262 KB of a layer stay L2-resident across the 192-336 workgroups);
//     all contractions are v_mfma_f32_16x16x32_bf16 with fp32 accumulation;
//   * attention: wave h owns head h from the projection to the context -- q, k (row-major) and v (transposed) of
//     the head stay in LDS as bf16; Q K^T of 16 queries at a time goes through a wave-private score tile from
//     which the sampled scores of the sparsity measure are gathered; the top-u ranking, the softmax of the
//     selected rows and P V need no workgroup barrier at all.
// Training mode writes what the (layer-by-layer) backward kernels consume -- packed q|k|v, ctx, x-hat / 1/sigma of
// both norms, x1, z, h, the selected rows -- with the same meaning as the unfused forward's saved tensors.
// bf16 matrix-core mode only (like rowblock.hip); the exact-fp32 mode keeps the layer-by-layer path.
#include <cstdlib>
#include "seqlayer_common.h"

namespace {


// Everything a layer needs sits at a fixed offset from one per-layer base (weights: one packed blob per layer, saves:
// [layer][B*L][width] slabs), so the kernel carries a handful of base pointers instead of 25 pointers per layer.
struct SeqStackP {
  const float* x;               // (B, L, 128) input of the first layer
  const unsigned char* wpack;   // per layer: fragment-ordered bf16 weights + fp32 vectors (layout: pack_offsets)
  long wpack_stride;            // bytes between layers
  const int32_t* idx[RF_SEQLAYER_MAX_LAYERS];  // (G, L, sample_k) key samples per layer
  long idx_stride;              // elements between consecutive group tables
  int32_t* top;                 // (layers, B, 8, n_top): written (read when force_top); may be null without save
  float* y;                     // (layers, B*L, 128): layer outputs
  float *qkv, *ctx, *xhat1, *rstd1, *x1, *z, *h, *xhat2, *rstd2;  // training saves (layers, B*L, width)
  int bf16_saves;  // RfSeqStack.flags: bit 0 = ctx, x1 and (with z) h are bf16 slabs; bit 1 = qkv; bit 2 = xhat1, xhat2, z
  __bf16* xin;     // optional bf16 (layers, B*L, 128): every layer's input image
  int B, L, F, n_layers, act, sample_k, n_top, idx_group, force_top, save;
  int split;  // 1: q / k projection and sparsity-measure scores in split-bf16 (default); 0: plain bf16 (RF_SEQ_SPLIT=0, A/B only)
  float scale, eps;
  DropCfg drop;   // nn.Dropout of the layers (cross_modal_transformer.py:295,298,299); state == null: off
  int drop_site0; // layer i uses sites drop_site0 + 3 i + {0: attention output, 1: hidden activation, 2: conv2 output}
};

// Phase timing aid (tools/seqlayer_probe.py builds a private copy with -DRF_SL_TIMING): lane 0 of every wave stamps
// the shader clock at the phase boundaries of the FIRST layer into rf_sl_timing[workgroup][wave][16].
#ifdef RF_SL_TIMING
__device__ unsigned long long rf_sl_timing[512 * 8 * 16];
#define SL_MARK(k) do { if (li == 0 && (threadIdx.x & 63) == 0 && blockIdx.x < 512) \
  rf_sl_timing[(blockIdx.x * 8 + (threadIdx.x >> 6)) * 16 + (k)] = __builtin_readcyclecounter(); } while (0)
#else
#define SL_MARK(k) do {} while (0)
#endif

// RT = row tiles of 16 (L <= 16 RT).  LDS (bytes), RT = 5:
//   xb   bf16 [16 RT][136]            21 760   x / ctx / x1 as MFMA A operand (one image, reused phase by phase)
//   qs   bf16 [8][16 RT][16]          20 480 \
//   ks   bf16 [8][16 RT][16]          20 480  } conv-pair phase: hb bf16 [16 RT][F + 8] (<= 42 240) aliases these
//   vt   bf16 [8][16][KS32 + 8]       26 624 /
//   scr  per wave 7 168               57 344   score tile fp32 [16][16 RT + 4] / P bf16 [32][KS32 + 8]; Ms, top, flags
//   part float2 [16 RT][8] + stat float2 [16 RT]   5 760   LayerNorm partial sums / per-row (mean, 1/sigma)
//   cnt  uint8 [16 RT][16 RT]         6 400   cnt[key][query] = how often `key` is among the query's samples (all heads)
template <int RT, bool SAVE, bool DROP>
__global__ __launch_bounds__(SL_NT) void seq_stack_fwd_kernel(const SeqStackP p) {
  constexpr int LP = 16 * RT, KS32 = ((LP + 31) / 32) * 32, KSTEPS = KS32 / 32, VP = KS32 + 8, SP = LP + 4;
  constexpr int SCR_BYTES = 7168, TB = 320;  // TB: floats of one staged 16 x 16 tile (pitch 20)
  static_assert(32 * VP * 2 <= 6656 && 5 * TB * 4 <= 6656 && SP > 0, "wave scratch layout");
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  __bf16* xb = reinterpret_cast<__bf16*>(smem);
  __bf16* qs = xb + LP * SL_XP;
  __bf16* ks = qs + SL_H * LP * SL_E;
  __bf16* vt = ks + SL_H * LP * SL_E;
  unsigned char* scr_base = reinterpret_cast<unsigned char*>(vt + SL_H * SL_E * VP);
  float2* part = reinterpret_cast<float2*>(scr_base + SL_NW * SCR_BYTES);
  float2* stat = part + LP * SL_NW;
  unsigned int* cnt32 = reinterpret_cast<unsigned int*>(stat + LP);  // [key][query / 4]: four queries per word
  __bf16* hb = qs;  // conv-pair phase alias
  // low half of the split-bf16 image of x ([LP][SL_XP], 16 RT x 272 B <= the V^T region): lives in the V^T region from
  // the end of a layer (the conv-pair images are dead) until the q / k projection of the next one has read it
  __bf16* xlo = vt;
  static_assert((size_t)LP * SL_XP <= (size_t)SL_H * SL_E * VP, "x_lo image must fit the V^T region");

  int tid = threadIdx.x, lane = tid & 63, fr = lane & 15, fq = lane >> 4;
  const int wave = tid >> 6;
  // (offsets are functions of the lane coordinates only: passing those through an empty asm at the phase boundaries keeps
  //  the compiler from hoisting hundreds of invariant addresses out of the layer loop and spilling them)
#define SL_LOCAL() asm volatile("" : "+v"(tid), "+v"(lane), "+v"(fr), "+v"(fq), "+v"(srow), "+v"(sc4))
  const int b = blockIdx.x, L = p.L, F = p.F, HP = F + 8;
  unsigned char* scr = scr_base + wave * SCR_BYTES;
  float* sc_f = reinterpret_cast<float*>(scr);           // score tile fp32 [16][SP] / staged output tiles [5][16][20]
  __bf16* sc_p = reinterpret_cast<__bf16*>(scr);         // probabilities bf16 [32][VP]
  float* Ms = reinterpret_cast<float*>(scr + 6656);      // [LP] sparsity measure (320 B)
  int* top_l = reinterpret_cast<int*>(scr + 6656 + 320);  // [32] selected rows, ascending (128 B)
  int srow = lane >> 2, sc4 = (lane & 3) * 4;            // staged-tile read-back: row, first column of this lane

  // ---- residual stream slice of this wave + bf16 image of x ----
  f32x4 xres[RT];
  {
    const float* xg = p.x + (long)b * L * SL_D + wave * 16 + fr;
#pragma unroll
    for (int rt = 0; rt < RT; ++rt)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int row = rt * 16 + fq * 4 + r;
        xres[rt][r] = xg[min(row, L - 1) * SL_D];  // clamped, unconditional (padded rows are never stored)
      }
#pragma unroll
    for (int rt = 0; rt < RT; ++rt)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const __bf16 hi = (__bf16)xres[rt][r];
        xb[(rt * 16 + fq * 4 + r) * SL_XP + wave * 16 + fr] = hi;
        xlo[(rt * 16 + fq * 4 + r) * SL_XP + wave * 16 + fr] = bf16_lo(xres[rt][r], hi);
      }
  }
  const PackOff po = pack_offsets(F);
  uint2 dkey = make_uint2(0, 0);
  uint32_t dstep = 0;
  if constexpr (DROP) {
    const unsigned long long sd = p.drop.state->seed;
    dkey = make_uint2((uint32_t)sd, (uint32_t)(sd >> 32));
    dstep = (uint32_t)p.drop.state->step;
  }

#pragma unroll 1
  for (int li = 0; li < p.n_layers; ++li) {
    const unsigned char* wl = p.wpack + (long)li * p.wpack_stride;
    const __bf16* w_qkv = reinterpret_cast<const __bf16*>(wl + po.wqkv);
    const __bf16* w_o = reinterpret_cast<const __bf16*>(wl + po.wo);
    const __bf16* w_1 = reinterpret_cast<const __bf16*>(wl + po.w1);
    const __bf16* w_2 = reinterpret_cast<const __bf16*>(wl + po.w2);
    const float* vec = reinterpret_cast<const float*>(wl + po.vec);
    const long lrow = ((long)li * p.B + b) * L;  // first row of this sequence in a [layers][B*L][...] slab

    // weight fragments of the projection are requested first: their L2 latency hides behind the staging below
    bf16x8 wf[3][4];
    float bias[3];
#pragma unroll
    for (int pt = 0; pt < 3; ++pt) {
#pragma unroll
      for (int kk = 0; kk < 4; ++kk) wf[pt][kk] = ld_wfrag(w_qkv, (pt * 8 + wave) * 4 + kk, lane);
      bias[pt] = vec[pt * SL_D + wave * 16 + fr];
    }
    // key samples of this layer -> count image cnt[key][query] (uint8, four queries per word): the sparsity measure
    // needs sum_j s(q, idx[q,j]) and max_j s(q, idx[q,j]) -- with the multiplicities in the layout of the MFMA
    // accumulator both come straight from the score registers (no score tile in LDS, no gathers)
    if (!p.force_top) {
      const int32_t* ig = p.idx[li] + (long)(b / p.idx_group) * p.idx_stride;
      const int n = L * p.sample_k;
      int v4[4];
#pragma unroll
      for (int u4 = 0; u4 < 4; ++u4) v4[u4] = ig[min(tid + u4 * SL_NT, n - 1)];
      for (int i = tid; i < LP * LP / 4; i += SL_NT) cnt32[i] = 0u;
      __syncthreads();
#pragma unroll
      for (int u4 = 0; u4 < 4; ++u4) {
        const int i = tid + u4 * SL_NT;
        if (i < n) {
          const int q = i / p.sample_k;
          atomicAdd(&cnt32[(v4[u4] * LP + q) >> 2], 1u << (8 * (q & 3)));
        }
      }
    }
    SL_MARK(0);
    __syncthreads();  // xb / x_lo complete (written by all waves), count image visible
    SL_MARK(1);

    SL_LOCAL();
    if (SAVE && p.xin) save_image_bf16(xb, SL_XP, SL_D, p.xin + lrow * SL_D, L, tid);
    // ================= phase 1: q | k | v of head `wave` =================
    // q and k feed the ProbSparse sparsity measure, whose top-u ranking is DISCONTINUOUS: a bf16-level rounding of the
    // projection or of q / k flips selections that the fp32 reference makes the other way (SURVEY section 7 "ProbSparse
    // parity": measure from ~fp32-accurate scores).  Both are therefore computed in split-bf16 (x = hi + lo, W = hi + lo:
    // three MFMAs per product, ~2^-16 relative) and kept as hi (qs / ks: what the softmax rows and the backward use)
    // + lo (wave-private scratch, read by the measure only); v, P V, the out-projection and the conv pair stay bf16.
    {
      constexpr int LO_BYTES = 2 * LP * SL_E * 2;
      static_assert(LO_BYTES + TB * 4 <= 6656, "q / k low halves + one staged tile must fit the wave scratch");
      // (element offsets from the slab base: the slab is fp32 or, RfSeqStack.flags bit 1, bf16)
      void* qkv_g = nullptr;
      if (SAVE) qkv_g = (p.bf16_saves & 2) ? static_cast<void*>(reinterpret_cast<__bf16*>(p.qkv) + lrow * (3 * SL_D) + wave * 16)
                                           : static_cast<void*>(p.qkv + lrow * (3 * SL_D) + wave * 16);
      __bf16* ql = reinterpret_cast<__bf16*>(scr);
      __bf16* kl = ql + LP * SL_E;
      float* st1 = reinterpret_cast<float*>(scr + LO_BYTES);  // one staged 16 x 16 tile (pitch 20)
      const __bf16* w_lo = reinterpret_cast<const __bf16*>(wl + po.lo);
      bf16x8 wlo[2][4];
#pragma unroll
      for (int pt = 0; pt < 2; ++pt)
#pragma unroll
        for (int kk = 0; kk < 4; ++kk) wlo[pt][kk] = ld_wfrag(w_lo, (pt * 8 + wave) * 4 + kk, lane);
      // ---- 1a: q, k ----
#pragma unroll
      for (int rt = 0; rt < RT; ++rt) {
        bf16x8 a[4], al[4];
#pragma unroll
        for (int kk = 0; kk < 4; ++kk) {
          a[kk] = ld_frag(xb + (rt * 16 + fr) * SL_XP + kk * 32 + fq * 8);
          al[kk] = ld_frag(xlo + (rt * 16 + fr) * SL_XP + kk * 32 + fq * 8);
        }
        f32x4 acc[2];
#pragma unroll
        for (int pt = 0; pt < 2; ++pt) {
          acc[pt] = f32x4{0.f, 0.f, 0.f, 0.f};
          if (p.split) {
#pragma unroll
            for (int kk = 0; kk < 4; ++kk) {
              acc[pt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(al[kk], wf[pt][kk], acc[pt], 0, 0, 0);
              acc[pt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[kk], wlo[pt][kk], acc[pt], 0, 0, 0);
            }
          }
#pragma unroll
          for (int kk = 0; kk < 4; ++kk) acc[pt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[kk], wf[pt][kk], acc[pt], 0, 0, 0);
        }
        const int row0 = rt * 16 + fq * 4;
#pragma unroll
        for (int pt = 0; pt < 2; ++pt)
#pragma unroll
          for (int r = 0; r < 4; ++r)
            acc[pt][r] = row0 + r < L ? acc[pt][r] + bias[pt] : 0.f;  // padded rows: exact zeros (masked keys, unused queries)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const __bf16 qh = (__bf16)acc[0][r], kh = (__bf16)acc[1][r];
          qs[(wave * LP + row0 + r) * SL_E + fr] = qh;
          ks[(wave * LP + row0 + r) * SL_E + fr] = kh;
          ql[(row0 + r) * SL_E + fr] = bf16_lo(acc[0][r], qh);
          kl[(row0 + r) * SL_E + fr] = bf16_lo(acc[1][r], kh);
        }
        if (qkv_g) {  // q | k of these 16 rows: a staged tile each, one 16-B store per lane and tile
#pragma unroll
          for (int pt = 0; pt < 2; ++pt) {
#pragma unroll
            for (int r = 0; r < 4; ++r) st1[(fq * 4 + r) * 20 + fr] = acc[pt][r];
            wave_sync_lds();
            if (rt * 16 + srow < L)
              store_qkv4(qkv_g, (long)(rt * 16 + srow) * (3 * SL_D) + pt * SL_D + sc4,
                         *reinterpret_cast<const float4*>(st1 + srow * 20 + sc4), p.bf16_saves & 2);
            wave_sync_lds();
          }
        }
      }
      __syncthreads();  // every wave has read x_lo: its LDS becomes V^T
      // ---- 1b: v ----
      // zero the key padding of V^T (columns LP..KS32-1 are never written; x_lo / the conv-pair phase overwrote the region)
      if constexpr (KS32 > LP) {
        for (int i = lane; i < SL_E * (KS32 - LP); i += 64)
          vt[(wave * SL_E + i / (KS32 - LP)) * VP + LP + i % (KS32 - LP)] = (__bf16)0.f;
      }
#pragma unroll
      for (int rt = 0; rt < RT; ++rt) {
        f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int kk = 0; kk < 4; ++kk)
          acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ld_frag(xb + (rt * 16 + fr) * SL_XP + kk * 32 + fq * 8), wf[2][kk], acc, 0, 0, 0);
        const int row0 = rt * 16 + fq * 4;
#pragma unroll
        for (int r = 0; r < 4; ++r) acc[r] = row0 + r < L ? acc[r] + bias[2] : 0.f;
        const bf16x4 v4 = {(__bf16)acc[0], (__bf16)acc[1], (__bf16)acc[2], (__bf16)acc[3]};
        *reinterpret_cast<bf16x4*>(vt + (wave * SL_E + fr) * VP + row0) = v4;  // V^T: 4 consecutive keys of channel fr
        if (qkv_g) {
#pragma unroll
          for (int r = 0; r < 4; ++r) st1[(fq * 4 + r) * 20 + fr] = acc[r];
          wave_sync_lds();
          if (rt * 16 + srow < L)
            store_qkv4(qkv_g, (long)(rt * 16 + srow) * (3 * SL_D) + 2 * SL_D + sc4,
                       *reinterpret_cast<const float4*>(st1 + srow * 20 + sc4), p.bf16_saves & 2);
          wave_sync_lds();
        }
      }
    }
    // out-projection fragments travel during the attention phase
    bf16x8 wfo[4];
#pragma unroll
    for (int kk = 0; kk < 4; ++kk) wfo[kk] = ld_wfrag(w_o, wave * 4 + kk, lane);
    SL_MARK(2);
    __syncthreads();  // every wave is done with xb: it becomes the ctx image
    SL_MARK(3);

    SL_LOCAL();
    // ================= phase 2: ProbSparse attention of head `wave` (no workgroup barrier inside) =================
    {
      const __bf16* Q = qs + wave * LP * SL_E;
      const __bf16* Kh = ks + wave * LP * SL_E;
      const __bf16* Vt = vt + wave * SL_E * VP;
      const __bf16* Ql = reinterpret_cast<const __bf16*>(scr);  // low halves of q / k (phase 1a), dead after the measure
      const __bf16* Kl = Ql + LP * SL_E;
      int32_t* top_g = p.top ? p.top + (((long)li * p.B + b) * SL_H + wave) * p.n_top : nullptr;
      const int u = p.n_top;
      bf16x8 kb[RT];
#pragma unroll
      for (int ct = 0; ct < RT; ++ct) kb[ct] = fq < 2 ? ld_frag(Kh + (ct * 16 + fr) * SL_E + fq * 8) : zero_frag();

      const float inv_L = 1.0f / (float)L;
      if (!p.force_top) {
        // (a) sparsity measure M[q] = max_j s(q, idx[q,j]) - sum_j s(q, idx[q,j]) / L, 16 queries per MFMA pass: each lane
        //     weighs its 4 x RT score registers with the sample multiplicities (one word = the 4 rows of a column)
#pragma unroll
        for (int rt = 0; rt < RT; ++rt) {
          const bf16x8 qa = fq < 2 ? ld_frag(Q + (rt * 16 + fr) * SL_E + fq * 8) : zero_frag();
          const bf16x8 qal = fq < 2 ? ld_frag(Ql + (rt * 16 + fr) * SL_E + fq * 8) : zero_frag();
          float mx[4] = {-INFINITY, -INFINITY, -INFINITY, -INFINITY}, sm[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
          for (int ct = 0; ct < RT; ++ct) {
            const unsigned int c4 = cnt32[((ct * 16 + fr) * LP + rt * 16 + fq * 4) >> 2];
            const bf16x8 kbl = fq < 2 ? ld_frag(Kl + (ct * 16 + fr) * SL_E + fq * 8) : zero_frag();
            f32x4 sv = {0.f, 0.f, 0.f, 0.f};
            if (p.split) {
              sv = __builtin_amdgcn_mfma_f32_16x16x32_bf16(qal, kb[ct], sv, 0, 0, 0);
              sv = __builtin_amdgcn_mfma_f32_16x16x32_bf16(qa, kbl, sv, 0, 0, 0);
            }
            sv = __builtin_amdgcn_mfma_f32_16x16x32_bf16(qa, kb[ct], sv, 0, 0, 0);  // ~fp32-accurate q . k
#pragma unroll
            for (int r = 0; r < 4; ++r) {
              const unsigned int c = (c4 >> (8 * r)) & 0xffu;
              sm[r] = fmaf((float)c, sv[r], sm[r]);
              mx[r] = fmaxf(mx[r], c ? sv[r] : -INFINITY);
            }
          }
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const float m_ = row16_max(mx[r]) - row16_sum(sm[r]) * inv_L;
            const int q = rt * 16 + fq * 4 + r;
            if (fr == 0) Ms[q] = q < L ? m_ : -INFINITY;
          }
        }
        wave_sync_lds();
        SL_MARK(4);
        // (b) rank: lane l ranks rows l and l + 64 against all measures (broadcast 16-B LDS reads, compare + carry-add in
        //     the vector unit only).  selected <=> fewer than u rows have a larger measure; ties: lower index first.
        //     Fast pass with strict comparisons; it selects exactly u rows unless a tie straddles the cut -- only then the
        //     exact tie-breaking pass runs.
        {
          const int q1 = lane, q2 = lane + 64;
          const float m1 = Ms[min(q1, LP - 1)], m2 = Ms[min(q2, LP - 1)];
          int r1 = 0, r2 = 0;
#pragma unroll
          for (int c = 0; c < LP / 4; ++c) {
            const float4 mo = *reinterpret_cast<const float4*>(Ms + 4 * c);  // rows >= L hold -inf: never larger
            r1 += (int)(mo.x > m1) + (int)(mo.y > m1) + (int)(mo.z > m1) + (int)(mo.w > m1);
            r2 += (int)(mo.x > m2) + (int)(mo.y > m2) + (int)(mo.z > m2) + (int)(mo.w > m2);
          }
          unsigned long long sel_lo = __ballot(q1 < L && r1 < u), sel_hi = __ballot(q2 < L && r2 < u);
          if (__popcll(sel_lo) + __popcll(sel_hi) != u) {  // (wave-uniform) a tie at the cut: exact ranks
#ifdef RF_SL_TIMING
            if (lane == 0 && blockIdx.x < 512) atomicAdd(&rf_sl_timing[(blockIdx.x * 8 + wave) * 16 + 15], 1ull);
#endif
            r1 = r2 = 0;
            for (int o = 0; o < L; ++o) {
              const float mo = Ms[o];
              r1 += (int)((mo > m1) | ((mo == m1) & (o < q1)));
              r2 += (int)((mo > m2) | ((mo == m2) & (o < q2)));
            }
            sel_lo = __ballot(q1 < L && r1 < u);
            sel_hi = __ballot(q2 < L && r2 < u);
          }
          const unsigned long long below = lane == 0 ? 0ull : (~0ull >> (64 - lane));
          const int p1 = __popcll(sel_lo & below), p2 = __popcll(sel_lo) + __popcll(sel_hi & below);
          if (((sel_lo >> lane) & 1ull) && p1 < 32) top_l[p1] = q1;  // (p < 32 always holds for finite measures)
          if (((sel_hi >> lane) & 1ull) && p2 < 32) top_l[p2] = q2;
        }
        wave_sync_lds();
        if (top_g && lane < u) top_g[lane] = top_l[lane];
        SL_MARK(5);
      } else {
        if (lane < u) top_l[lane] = top_g[lane];
        wave_sync_lds();
      }

      // (c) lazy rows: ctx[q] = mean_s V[s]  (cross_modal_transformer.py:113-116).  Every row is filled with the mean
      //     first (full 32-B row slices, no selection test); the selected rows are overwritten by (d) -- LDS
      //     operations of one wave execute in order.
      {
        float a = 0.f;
#pragma unroll
        for (int c = 0; c < (KS32 / 8 + 3) / 4; ++c) {  // 16-B chunks of channel fr's key row: chunk fq + 4 c
          if (fq + 4 * c < KS32 / 8) {
            const bf16x8 vv = ld_frag(Vt + fr * VP + (fq + 4 * c) * 8);  // (keys >= L hold exact zeros)
#pragma unroll
            for (int j = 0; j < 8; ++j) a += (float)vv[j];
          }
        }
        a += __shfl_xor(a, 16);
        a += __shfl_xor(a, 32);
        const float vm = a * inv_L;  // lane (fr, *) holds the mean of channel fr
        typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
        u32x4 lo, hi;  // the 16 means as packed bf16 pairs, wave-uniform
#pragma unroll
        for (int e2 = 0; e2 < 8; ++e2) {
          const float m0 = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(vm), 2 * e2));
          const float m1 = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(vm), 2 * e2 + 1));
          typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
          const bf16x2 pk = {(__bf16)m0, (__bf16)m1};
          const unsigned int w = *reinterpret_cast<const unsigned int*>(&pk);
          if (e2 < 4) lo[e2] = w; else hi[e2 - 4] = w;
        }
        for (int q = lane; q < L; q += 64) {
          *reinterpret_cast<u32x4*>(xb + q * SL_XP + wave * 16) = lo;
          *reinterpret_cast<u32x4*>(xb + q * SL_XP + wave * 16 + 8) = hi;
        }
      }
      SL_MARK(6);

      // (d) active rows: P = softmax(scale * Q_sel K^T), ctx[top] = P V
      if constexpr (KS32 > LP) {  // zero the key padding of the probability image (columns LP..KS32-1 of the 32 rows)
        for (int i = lane; i < 32 * (KS32 - LP); i += 64) sc_p[(i / (KS32 - LP)) * VP + LP + i % (KS32 - LP)] = (__bf16)0.f;
      }
#pragma unroll
      for (int t2 = 0; t2 < 2; ++t2) {
        if (t2 * 16 < u) {
          const int qrow = top_l[min(t2 * 16 + fr, u - 1)];
          const bf16x8 qa = fq < 2 ? ld_frag(Q + qrow * SL_E + fq * 8) : zero_frag();
          f32x4 sv[RT];
          float mx[4] = {-INFINITY, -INFINITY, -INFINITY, -INFINITY};
#pragma unroll
          for (int ct = 0; ct < RT; ++ct) {
            sv[ct] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(qa, kb[ct], f32x4{0.f, 0.f, 0.f, 0.f}, 0, 0, 0);
            const bool live = ct * 16 + fr < L;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
              sv[ct][r] = live ? sv[ct][r] * p.scale : -INFINITY;
              mx[r] = fmaxf(mx[r], sv[ct][r]);
            }
          }
          float sum[4];
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            mx[r] = row16_max(mx[r]);
            sum[r] = 0.f;
          }
#pragma unroll
          for (int ct = 0; ct < RT; ++ct)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
              sv[ct][r] = __expf(sv[ct][r] - mx[r]);
              sum[r] += sv[ct][r];
            }
#pragma unroll
          for (int r = 0; r < 4; ++r) sum[r] = __builtin_amdgcn_rcpf(row16_sum(sum[r]));
#pragma unroll
          for (int ct = 0; ct < RT; ++ct)
#pragma unroll
            for (int r = 0; r < 4; ++r) sc_p[(t2 * 16 + fq * 4 + r) * VP + ct * 16 + fr] = (__bf16)(sv[ct][r] * sum[r]);
        }
      }
      wave_sync_lds();
#pragma unroll
      for (int t2 = 0; t2 < 2; ++t2) {
        if (t2 * 16 < u) {
          f32x4 o = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
          for (int kk = 0; kk < KSTEPS; ++kk) {
            const bf16x8 pa = ld_frag(sc_p + (t2 * 16 + fr) * VP + kk * 32 + fq * 8);
            const bf16x8 vb = ld_frag(Vt + fr * VP + kk * 32 + fq * 8);
            o = __builtin_amdgcn_mfma_f32_16x16x32_bf16(pa, vb, o, 0, 0, 0);
          }
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const int i = t2 * 16 + fq * 4 + r;
            if (i < u) xb[top_l[i] * SL_XP + wave * 16 + fr] = (__bf16)o[r];
          }
        }
      }
    }
    SL_MARK(7);
    __syncthreads();  // ctx image complete
    SL_MARK(8);

    SL_LOCAL();
    // ================= phase 3: out-projection + residual + LayerNorm 1 (wave = 16 output columns) =================
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
          const f32x4 f = drop_factors(p.drop, dkey, dstep, (uint32_t)(p.drop_site0 + 3 * li), (long)b * L + rt * 16, SL_D, col, lane);
#pragma unroll
          for (int r = 0; r < 4; ++r) v[rt][r] = (acc[r] + bo) * f[r] + xres[rt][r];
        } else {
#pragma unroll
          for (int r = 0; r < 4; ++r) v[rt][r] = acc[r] + bo + xres[rt][r];
        }
      }
      if (SAVE)  // the context as the out-projection (and its weight gradient) consumes it: the bf16 image (widened, or as it is)
        save_image_as(xb, SL_XP, SL_D, p.ctx, lrow * SL_D, L, tid, (p.bf16_saves & 1) != 0);
      stack_layer_norm<RT>(v, SAVE ? p.rstd1 + lrow : nullptr, L, part, stat, wave, lane, p.eps);  // (barriers fence the ctx reads)
      if (SAVE) {  // x-hat of norm1: RT staged tiles, one sync pair
        const long xh_o = lrow * SL_D + wave * 16;  // (element offset: the slab is fp32 or, flags bit 2, bf16)
#pragma unroll
        for (int rt = 0; rt < RT; ++rt)
#pragma unroll
          for (int r = 0; r < 4; ++r) sc_f[rt * TB + (fq * 4 + r) * 20 + fr] = v[rt][r];
        wave_sync_lds();
#pragma unroll
        for (int rt = 0; rt < RT; ++rt)
          if (rt * 16 + srow < L)
            store_qkv4(p.xhat1, xh_o + (rt * 16 + srow) * SL_D + sc4, *reinterpret_cast<const float4*>(sc_f + rt * TB + srow * 20 + sc4),
                       p.bf16_saves & 4);
        wave_sync_lds();
      }
#pragma unroll
      for (int rt = 0; rt < RT; ++rt)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const float y1 = v[rt][r] * g1 + be1;
          xres[rt][r] = y1;
          xb[(rt * 16 + fq * 4 + r) * SL_XP + col] = (__bf16)y1;
        }
      if (SAVE && !(p.bf16_saves & 1)) {  // (bf16 saves: the x1 IMAGE is copied out once it is complete, below)
        float* x1_g = p.x1 + lrow * SL_D + wave * 16;
#pragma unroll
        for (int rt = 0; rt < RT; ++rt)
#pragma unroll
          for (int r = 0; r < 4; ++r) sc_f[rt * TB + (fq * 4 + r) * 20 + fr] = xres[rt][r];
        wave_sync_lds();
#pragma unroll
        for (int rt = 0; rt < RT; ++rt)
          if (rt * 16 + srow < L)
            *reinterpret_cast<float4*>(x1_g + (rt * 16 + srow) * SL_D + sc4) = *reinterpret_cast<const float4*>(sc_f + rt * TB + srow * 20 + sc4);
        wave_sync_lds();
      }
    }
    SL_MARK(9);
    __syncthreads();  // x1 image complete; q / k / v^T are dead: their LDS becomes the hidden activation image
    SL_MARK(10);

    SL_LOCAL();
    if (SAVE && (p.bf16_saves & 1)) save_image_bf16(xb, SL_XP, SL_D, reinterpret_cast<__bf16*>(p.x1) + lrow * SL_D, L, tid);
    // ================= phase 4: conv1 + activation (wave = column tiles wave, wave + 8) =================
    const bool h_img = SAVE && (p.bf16_saves & 1) && p.z;  // h leaves as the bf16 image it is for conv2 (z stays fp32: GELU' reads it)
    {
      float* z_g = (SAVE && p.z) ? p.z : nullptr;  // (+ lrow * F elements: fp32 or, flags bit 2, bf16)
      float* h_g = (SAVE && !h_img) ? p.h : nullptr;
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
        if (p.act == RF_ACT_GELU) {  // (one uniform branch per tile column, not one per element)
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
            const f32x4 f = drop_factors(p.drop, dkey, dstep, (uint32_t)(p.drop_site0 + 3 * li + 1), (long)b * L + rt * 16, F, col, lane);
#pragma unroll
            for (int r = 0; r < 4; ++r) hh[rt][r] *= f[r];
          }
        }
#pragma unroll
        for (int rt = 0; rt < RT; ++rt)
#pragma unroll
          for (int r = 0; r < 4; ++r) hb[(rt * 16 + fq * 4 + r) * HP + col] = (__bf16)hh[rt][r];
        if (h_g || z_g) {
#pragma unroll
          for (int which = 0; which < 2; ++which) {
            float* dst = which == 0 ? z_g : h_g;
            if (dst) {
#pragma unroll
              for (int rt = 0; rt < RT; ++rt)
#pragma unroll
                for (int r = 0; r < 4; ++r) sc_f[rt * TB + (fq * 4 + r) * 20 + fr] = which == 0 ? zz[rt][r] : hh[rt][r];
              wave_sync_lds();
#pragma unroll
              for (int rt = 0; rt < RT; ++rt)
                if (rt * 16 + srow < L)
                  store_qkv4(dst, (lrow + rt * 16 + srow) * F + ct * 16 + sc4, *reinterpret_cast<const float4*>(sc_f + rt * TB + srow * 20 + sc4),
                             which == 0 ? (p.bf16_saves & 4) : 0);
              wave_sync_lds();
            }
          }
        }
      }
    }
    SL_MARK(11);
    __syncthreads();  // hidden activation image complete
    SL_MARK(12);

    SL_LOCAL();
    if (h_img) save_image_bf16(hb, HP, F, reinterpret_cast<__bf16*>(p.h) + lrow * F, L, tid);
    // ================= phase 5: conv2 + residual + LayerNorm 2 =================
    {
      const int col = wave * 16 + fr;
      const float b2 = vec[512 + F + col], g2 = vec[896 + F + col], be2 = vec[1024 + F + col];
      f32x4 v[RT];
#pragma unroll
      for (int rt = 0; rt < RT; ++rt) v[rt] = f32x4{0.f, 0.f, 0.f, 0.f};
      const int nk = F / 32;  // <= 8
      bf16x8 wf2[8];
#pragma unroll
      for (int kk = 0; kk < 8; ++kk) wf2[kk] = ld_wfrag(w_2, wave * nk + min(kk, nk - 1), lane);  // one round trip, not nk
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
          const f32x4 f = drop_factors(p.drop, dkey, dstep, (uint32_t)(p.drop_site0 + 3 * li + 2), (long)b * L + rt * 16, SL_D, col, lane);
#pragma unroll
          for (int r = 0; r < 4; ++r) v[rt][r] = (v[rt][r] + b2) * f[r] + xres[rt][r];
        }
      } else {
#pragma unroll
        for (int rt = 0; rt < RT; ++rt)
#pragma unroll
          for (int r = 0; r < 4; ++r) v[rt][r] += b2 + xres[rt][r];
      }
      stack_layer_norm<RT>(v, SAVE ? p.rstd2 + lrow : nullptr, L, part, stat, wave, lane, p.eps);  // (barriers fence the hb / xb reads)
      // without saves only the last layer's output is needed: it goes to slab 0.  With saves AND the bf16 input images
      // (p.xin: what the weight gradients read) the intermediate outputs have no reader either
      const bool all_y = SAVE && !p.xin;
      const bool store_y = all_y || li == p.n_layers - 1;
      float* y_g = p.y + (all_y ? lrow : (long)b * L) * SL_D + wave * 16;
      if (SAVE) {
        const long xh_o = lrow * SL_D + wave * 16;
#pragma unroll
        for (int rt = 0; rt < RT; ++rt)
#pragma unroll
          for (int r = 0; r < 4; ++r) sc_f[rt * TB + (fq * 4 + r) * 20 + fr] = v[rt][r];
        wave_sync_lds();
#pragma unroll
        for (int rt = 0; rt < RT; ++rt)
          if (rt * 16 + srow < L)
            store_qkv4(p.xhat2, xh_o + (rt * 16 + srow) * SL_D + sc4, *reinterpret_cast<const float4*>(sc_f + rt * TB + srow * 20 + sc4),
                       p.bf16_saves & 4);
        wave_sync_lds();
      }
#pragma unroll
      for (int rt = 0; rt < RT; ++rt)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const float y2 = v[rt][r] * g2 + be2;
          xres[rt][r] = y2;
          const __bf16 hi = (__bf16)y2;
          xb[(rt * 16 + fq * 4 + r) * SL_XP + col] = hi;  // next layer's A operand ...
          xlo[(rt * 16 + fq * 4 + r) * SL_XP + col] = bf16_lo(y2, hi);  // ... and its low half (the conv-pair images are dead)
        }
      if (store_y) {
#pragma unroll
        for (int rt = 0; rt < RT; ++rt)
#pragma unroll
          for (int r = 0; r < 4; ++r) sc_f[rt * TB + (fq * 4 + r) * 20 + fr] = xres[rt][r];
        wave_sync_lds();
#pragma unroll
        for (int rt = 0; rt < RT; ++rt)
          if (rt * 16 + srow < L)
            *reinterpret_cast<float4*>(y_g + (rt * 16 + srow) * SL_D + sc4) = *reinterpret_cast<const float4*>(sc_f + rt * TB + srow * 20 + sc4);
        wave_sync_lds();
      }
    }
    SL_MARK(13);
    // (the barrier at the head of the next layer publishes xb)
  }
}

template <int RT>
constexpr size_t stack_lds_bytes() {
  constexpr int LP = 16 * RT, KS32 = ((LP + 31) / 32) * 32, VP = KS32 + 8;
  return (size_t)LP * SL_XP * 2 + 2 * (size_t)SL_H * LP * SL_E * 2 + (size_t)SL_H * SL_E * VP * 2 + SL_NW * 7168 +
         (size_t)LP * SL_NW * 8 + (size_t)LP * 8 + (size_t)LP * LP;
}

// ---- weight fragments -------------------------------------------------------------------------------------------
struct PackTable {
  int count, pad;
  RfSeqPackEntry e[RF_SEQLAYER_MAX_PACK];
  int first_block[RF_SEQLAYER_MAX_PACK + 1];
};

// one workgroup of 256 threads per 4 fragments (4 KB of output)
__global__ __launch_bounds__(256) void seq_pack_kernel(const PackTable t) {
  int e = 0;
  while (e + 1 < t.count && (int)blockIdx.x >= t.first_block[e + 1]) ++e;
  const RfSeqPackEntry& ent = t.e[e];
  if (ent.K == 0) {  // fp32 vector copy (biases, norm parameters), N floats
    const int i = ((int)blockIdx.x - t.first_block[e]) * 256 + threadIdx.x;
    if (i < ent.N) static_cast<float*>(ent.out)[i] = ent.w[i];
    return;
  }
  const int KS = ent.K / 32, nfrag = (ent.N / 16) * KS;
  const int f = ((int)blockIdx.x - t.first_block[e]) * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
  if (f >= nfrag) return;
  const int ct = f / KS, kk = f % KS, n = ct * 16 + (lane & 15), k0 = kk * 32 + (lane >> 4) * 8;
  bf16x8 o;
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    const float w = ent.transpose ? ent.w[(long)(k0 + j) * ent.ldw + n] : ent.w[(long)n * ent.ldw + k0 + j];
    o[j] = ent.residual ? bf16_lo(w, (__bf16)w) : (__bf16)w;
  }
  *reinterpret_cast<bf16x8*>(static_cast<__bf16*>(ent.out) + ((long)f * 64 + lane) * 8) = o;
}

// the same work from a table in DEVICE memory (any number of entries: every stack of a model in one launch)
__global__ __launch_bounds__(256) void seq_pack_table_kernel(const RfSeqPackEntry* __restrict__ ents,
                                                             const int* __restrict__ first_block, int count) {
  int lo = 0, hi = count - 1;  // last entry whose first block is <= blockIdx.x
  while (lo < hi) {
    const int mid = (lo + hi + 1) >> 1;
    if (first_block[mid] <= (int)blockIdx.x) lo = mid; else hi = mid - 1;
  }
  const RfSeqPackEntry ent = ents[lo];
  const int rel = (int)blockIdx.x - first_block[lo];
  if (ent.K == 0) {
    const int i = rel * 256 + threadIdx.x;
    if (i < ent.N) static_cast<float*>(ent.out)[i] = ent.w[i];
    return;
  }
  const int KS = ent.K / 32, nfrag = (ent.N / 16) * KS;
  const int f = rel * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
  if (f >= nfrag) return;
  const int ct = f / KS, kk = f % KS, n = ct * 16 + (lane & 15), k0 = kk * 32 + (lane >> 4) * 8;
  bf16x8 o;
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    const float w = ent.transpose ? ent.w[(long)(k0 + j) * ent.ldw + n] : ent.w[(long)n * ent.ldw + k0 + j];
    o[j] = ent.residual ? bf16_lo(w, (__bf16)w) : (__bf16)w;
  }
  *reinterpret_cast<bf16x8*>(static_cast<__bf16*>(ent.out) + ((long)f * 64 + lane) * 8) = o;
}

inline bool al16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; }

}  // namespace

extern "C" int rf_seqlayer_pack_blocks(int N, int K) { return K == 0 ? (N + 255) / 256 : ((N / 16) * (K / 32) + 3) / 4; }

extern "C" int rf_seqlayer_pack_table(const RfSeqPackEntry* entries_dev, const int32_t* first_block_dev, int count, int blocks,
                                      void* stream) {
  RF_REQUIRE(entries_dev && first_block_dev && count > 0 && blocks > 0);
  RF_LAUNCH(seq_pack_table_kernel, dim3(blocks), dim3(256), 0, static_cast<hipStream_t>(stream), entries_dev, first_block_dev, count);
  RF_CHECK_LAUNCH();
  return RF_OK;
}

#ifdef RF_SL_TIMING
extern "C" void* rf_sl_timing_address() {
  void* a = nullptr;
  (void)hipGetSymbolAddress(&a, HIP_SYMBOL(rf_sl_timing));
  return a;
}
#endif

extern "C" int rf_seqlayer_pack(const RfSeqPackEntry* entries, int count, void* stream) {
  RF_REQUIRE(entries && count > 0 && count <= RF_SEQLAYER_MAX_PACK);
  PackTable t{};
  t.count = count;
  int blocks = 0;
  for (int i = 0; i < count; ++i) {
    const RfSeqPackEntry& e = entries[i];
    RF_REQUIRE(e.w && e.out && e.N > 0 && e.K >= 0);
    RF_REQUIRE(e.K == 0 || (e.N % 16 == 0 && e.K % 32 == 0 && al16(e.out)));
    t.e[i] = e;
    t.first_block[i] = blocks;
    blocks += e.K == 0 ? (e.N + 255) / 256 : ((e.N / 16) * (e.K / 32) + 3) / 4;
  }
  t.first_block[count] = blocks;
  RF_LAUNCH(seq_pack_kernel, dim3(blocks), dim3(256), 0, static_cast<hipStream_t>(stream), t);
  RF_CHECK_LAUNCH();
  return RF_OK;
}

extern "C" int rf_seqlayer_supported(int L, int d_model, int n_heads, int d_ff, int sample_k, int n_top) {
  return L >= 2 && L <= 80 && d_model == SL_D && n_heads == SL_H && d_ff >= 32 && d_ff <= 256 && d_ff % 32 == 0 &&
         sample_k >= 1 && sample_k <= L && n_top >= 1 && n_top <= 32 && n_top <= L && L * sample_k <= 2048;
}

extern "C" int64_t rf_seqlayer_pack_bytes(int d_ff) { return pack_offsets(d_ff).total; }

extern "C" int rf_seqlayer_fwd(const RfSeqStack* st_, const float* x, int B, int L, int d_model, int n_heads, int d_ff,
                               int act, int sample_k, int n_top, int idx_group, int force_top, int save, float scale,
                               float eps, float drop_p, const void* rng_state, int drop_site0, void* stream) {
  RF_REQUIRE(st_ && x && B > 0 && st_->n_layers > 0 && st_->n_layers <= RF_SEQLAYER_MAX_LAYERS);
  if (!rf_seqlayer_supported(L, d_model, n_heads, d_ff, sample_k, n_top)) {
    rf_g_last_error = "rf_seqlayer_fwd: shape outside the fused kernel's range";
    return RF_EUNSUPPORTED;
  }
  const RfSeqStack& s = *st_;
  RF_REQUIRE(s.wpack && al16(s.wpack) && s.wpack_stride >= rf_seqlayer_pack_bytes(d_ff) && s.wpack_stride % 16 == 0 && s.y);
  RF_REQUIRE(!force_top || s.top);
  RF_REQUIRE(!save || (s.qkv && s.ctx && s.xhat1 && s.rstd1 && s.x1 && s.h && s.xhat2 && s.rstd2 && s.top));
  SeqStackP p{};
  p.x = x; p.wpack = static_cast<const unsigned char*>(s.wpack); p.wpack_stride = s.wpack_stride;
  for (int i = 0; i < s.n_layers; ++i) {
    RF_REQUIRE(force_top || s.idx[i]);
    p.idx[i] = s.idx[i];
  }
  p.idx_stride = s.idx_stride > 0 ? s.idx_stride : (long)L * sample_k;
  p.top = s.top; p.y = s.y;
  p.qkv = s.qkv; p.ctx = s.ctx; p.xhat1 = s.xhat1; p.rstd1 = s.rstd1; p.x1 = s.x1; p.z = s.z; p.h = s.h;
  p.xhat2 = s.xhat2; p.rstd2 = s.rstd2; p.bf16_saves = save ? (s.flags & 7) : 0;
  p.xin = save ? static_cast<__bf16*>(s.xin) : nullptr;
  RF_REQUIRE(!p.xin || al16(p.xin));
  p.B = B; p.L = L; p.F = d_ff; p.n_layers = s.n_layers; p.act = act; p.sample_k = sample_k; p.n_top = n_top;
  p.idx_group = (idx_group <= 0 || idx_group > B) ? B : idx_group;
  p.force_top = force_top; p.save = save; p.scale = scale; p.eps = eps;
  {
    const char* e = getenv("RF_SEQ_SPLIT");
    p.split = !(e && e[0] == '0');
  }
  RF_REQUIRE(drop_p >= 0.f && drop_p < 1.f && (drop_p == 0.f || rng_state));
  p.drop = make_drop_cfg(rng_state, nullptr, 0, drop_p);
  p.drop_site0 = drop_site0;
  const hipStream_t st = static_cast<hipStream_t>(stream);
  const bool drop = p.drop.state != nullptr;
#define RF_SL_GO(RT_, SAVE_, DROP_)                                                                                  \
  do {                                                                                                               \
    static bool attr = false;                                                                                        \
    if (!attr) {                                                                                                     \
      (void)hipFuncSetAttribute(reinterpret_cast<const void*>(seq_stack_fwd_kernel<RT_, SAVE_, DROP_>),              \
                                hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);                             \
      attr = true;                                                                                                   \
    }                                                                                                                \
    RF_LAUNCH((seq_stack_fwd_kernel<RT_, SAVE_, DROP_>), dim3(B), dim3(SL_NT), stack_lds_bytes<RT_>(), st, p);       \
  } while (0)
  if (L <= 48) {
    if (drop && save) RF_SL_GO(3, true, true);
    else if (drop) RF_SL_GO(3, false, true);   // train-mode forward without grad (the target-side pass of a train step)
    else if (save) RF_SL_GO(3, true, false);
    else RF_SL_GO(3, false, false);
  } else {
    if (drop && save) RF_SL_GO(5, true, true);
    else if (drop) RF_SL_GO(5, false, true);
    else if (save) RF_SL_GO(5, true, false);
    else RF_SL_GO(5, false, false);
  }
#undef RF_SL_GO
  RF_CHECK_LAUNCH();
  return RF_OK;
}
