// Backward of the fused per-sequence encoder stack (seqlayer.hip): ONE workgroup owns ONE sequence and walks the
// layers of a PerceiveEncoder (cross_modal_transformer.py:288-301 x layers) from the last to the first, per layer
//
//     LayerNorm-2 backward -> conv2^T -> activation' -> conv1^T + skip -> LayerNorm-1 backward -> out-projection^T
//     -> ProbSparse attention backward (8 heads) -> packed q|k|v projection^T + skip,
//
// i.e. what the layer-by-layer path does with four row-block launches and one attention launch per layer.  The
// gradient of the residual stream stays in REGISTERS between the layers (wave w: columns 16w..16w+15 of every row, MFMA
// accumulator layout, as the forward); every GEMM operand A is a bf16 image in LDS; the weights are read as B
// fragments of a TRANSPOSED fragment-ordered bf16 copy (rf_seqlayer_pack with transpose = 1).
//
// Attention backward of head h on wave h, no workgroup barrier inside:  P = softmax(scale Q_sel K^T) is recomputed from
// the saved q|k|v exactly as the forward computed it (same bf16 operands, same MFMA);
//     dP = dC_sel V^T,  dS = scale P o (dP - rowsum(P o dP)),  dV = P^T dC_sel + (1/L) sum_{q not selected} dC[q],
//     dK = dS^T Q_sel,  dQ[top] = dS K                                   (cross_modal_transformer.py:76-131 reversed).
// q, k, v fragments come straight from global memory (fp32 -> bf16 in registers); the accumulator layout writes
// TRANSPOSED images for free (4 consecutive rows of a column = one 8-B store), which is the A operand P^T / dS^T need.
//
// What leaves the kernel: dx, and per layer the four gradients the weight-gradient GEMMs consume (d pre-norm-2, dz,
// d pre-norm-1, d q|k|v; stored from the bf16 images -- the weight-gradient kernels round their operands to bf16
// anyway) plus the LayerNorm parameter gradients (atomicAdd into the caller's accumulators).
#include "seqlayer_common.h"

namespace {

struct SeqStackBwdP {
  const float* dy;              // (B*L, 128) gradient of the stack output
  float* dx;                    // (B*L, 128) gradient of the stack input
  const unsigned char* wpack;   // per layer: transposed fragment-ordered bf16 weights + gamma1 | gamma2 (bwd_pack_offsets)
  long wpack_stride;
  const float *qkv, *xhat1, *rstd1, *zsrc, *xhat2, *rstd2;  // saves of the fused forward, (layers, B*L, width)
  const int32_t* top;           // (layers, B, 8, n_top)
  float *dpre2, *dz, *dpre1, *dqkv;  // (layers, B*L, 128 | F | 128 | 384)
  float* dg1[RF_SEQLAYER_MAX_LAYERS];
  float* db1[RF_SEQLAYER_MAX_LAYERS];
  float* dg2[RF_SEQLAYER_MAX_LAYERS];
  float* db2[RF_SEQLAYER_MAX_LAYERS];
  int B, L, F, n_layers, act, n_top;
  int bf16_grads;  // dpre2 / dz / dpre1 / dqkv are bf16 slabs (RfSeqStackBwd.flags & 1)
  int qkv_bf16;    // the saved q | k | v slab is bf16 (RfSeqStackBwd.flags & 2)
  int norm_bf16, z_bf16;  // xhat1 / xhat2 and zsrc are bf16 slabs (RfSeqStackBwd.flags & 4 / & 8)
  float scale;
  DropCfg drop;   // the forward's nn.Dropout masks are regenerated from (seed, step, site, element); state == null: off
  int drop_site0; // layer i: sites drop_site0 + 3 i + {0: attention output, 1: hidden activation, 2: conv2 output}
};

// Phase timing aid (tools/seqlayer_probe.py: private -DRF_SL_TIMING build): lane 0 of every wave stamps the shader
// clock at the phase boundaries of the LAST layer (the first one processed) into rf_slb_timing[workgroup][wave][16].
#ifdef RF_SL_TIMING
__device__ unsigned long long rf_slb_timing[512 * 8 * 16];
#define SLB_MARK(k) do { if (li == p.n_layers - 1 && (threadIdx.x & 63) == 0 && blockIdx.x < 512) \
  rf_slb_timing[(blockIdx.x * 8 + (threadIdx.x >> 6)) * 16 + (k)] = __builtin_readcyclecounter(); } while (0)
#else
#define SLB_MARK(k) do {} while (0)
#endif

// RT = row tiles of 16 (L <= 16 RT).  LDS (bytes), RT = 5, F = 256:
//   xb   bf16 [16 RT][136]   21 760 \  d pre-norm images (A operand of conv2^T / out-projection^T)
//   hb   bf16 [16 RT][F + 8] 42 240 /  dz image (A operand of conv1^T);  the attention phase reuses both as
//                                      dq  bf16 [16 RT][392]  62 720   d q|k|v image (A operand of the projection^T)
//   scr  per wave 10 240     81 920    dC^T bf16 [16][KS32 + 8] | P^T / dS^T bf16 [16 RT][40] or dS bf16 [32][KS32 + 8] | top
//   part float2 [16 RT][8] + stat float4 [16 RT]   6 400
template <int RT, bool DROP>
__global__ __launch_bounds__(SL_NT) void seq_stack_bwd_kernel(const SeqStackBwdP p) {
  constexpr int LP = 16 * RT, KS32 = ((LP + 31) / 32) * 32, KSTEPS = KS32 / 32, DTP = KS32 + 8, TP = 40, QP = 3 * SL_D + 8;
  constexpr int SCR_BYTES = 10240, DT_BYTES = 3328, BUF_BYTES = 6656;
  static_assert(SL_E * DTP * 2 <= DT_BYTES && LP * TP * 2 <= BUF_BYTES && 32 * DTP * 2 <= BUF_BYTES, "wave scratch layout");
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  __bf16* xb = reinterpret_cast<__bf16*>(smem);
  __bf16* hb = xb + LP * SL_XP;
  __bf16* dqi = xb;  // attention / projection phase alias of xb + hb
  const int F = p.F, HP = F + 8;
  // region size: the larger of (xb + hb) and the dq image
  const int region = max(LP * SL_XP + LP * HP, LP * QP) * 2;
  unsigned char* scr_base = smem + ((region + 15) & ~15);
  float2* part = reinterpret_cast<float2*>(scr_base + SL_NW * SCR_BYTES);
  float4* stat = reinterpret_cast<float4*>(part + LP * SL_NW);

  int tid = threadIdx.x, lane = tid & 63, fr = lane & 15, fq = lane >> 4;
  const int wave = tid >> 6, b = blockIdx.x, L = p.L, u = p.n_top;
  int col = wave * 16 + fr;
  // Every LDS / global offset below is a function of (tid, lane, fr, fq, col) only, i.e. invariant across the layers and
  // phases: left alone, the compiler hoists hundreds of them out of the layer loop and spills them.  Passing the lane
  // coordinates through an empty asm at every phase boundary keeps the address arithmetic next to its use.
#define SLB_LOCAL() asm volatile("" : "+v"(tid), "+v"(lane), "+v"(fr), "+v"(fq), "+v"(col))
  unsigned char* scr = scr_base + wave * SCR_BYTES;
  __bf16* dT = reinterpret_cast<__bf16*>(scr);                          // dC^T of this head: [16][DTP]
  __bf16* buf = reinterpret_cast<__bf16*>(scr + DT_BYTES);              // P^T / dS^T [LP][TP], then dS [32][DTP]
  int* top_l = reinterpret_cast<int*>(scr + DT_BYTES + BUF_BYTES);      // [32]

  // ---- gradient slice of this wave ----
  f32x4 dyres[RT];
  {
    const float* dg = p.dy + (long)b * L * SL_D + col;
#pragma unroll
    for (int rt = 0; rt < RT; ++rt)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int row = rt * 16 + fq * 4 + r;
        const float v = dg[min(row, L - 1) * SL_D];
        dyres[rt][r] = row < L ? v : 0.f;
      }
  }
  const BwdPackOff po = bwd_pack_offsets(F);
  const float inv_L = 1.0f / (float)L;
  uint2 dkey = make_uint2(0, 0);
  uint32_t dstep = 0;
  if constexpr (DROP) {
    const unsigned long long sd = p.drop.state->seed;
    dkey = make_uint2((uint32_t)sd, (uint32_t)(sd >> 32));
    dstep = (uint32_t)p.drop.state->step;
  }

#pragma unroll 1
  for (int li = p.n_layers - 1; li >= 0; --li) {
    const unsigned char* wl = p.wpack + (long)li * p.wpack_stride;
    const __bf16* w_2t = reinterpret_cast<const __bf16*>(wl + po.w2t);
    const __bf16* w_1t = reinterpret_cast<const __bf16*>(wl + po.w1t);
    const __bf16* w_ot = reinterpret_cast<const __bf16*>(wl + po.wot);
    const __bf16* w_qkvt = reinterpret_cast<const __bf16*>(wl + po.wqkvt);
    const float* vec = reinterpret_cast<const float*>(wl + po.vec);
    const long lrow = ((long)li * p.B + b) * L;
    f32x4 res[RT];  // the skip gradient: d pre-norm-2, later d pre-norm-1

    SLB_MARK(0);
    // ================= norm2 backward =================
    SLB_LOCAL();
    stack_ln_bwd<RT>(dyres, p.xhat2, lrow * SL_D + col, p.norm_bf16, p.rstd2 + lrow, vec[SL_D + col], p.dg2[li] + col, p.db2[li] + col, L,
                     part, stat, wave, lane);
#pragma unroll
    for (int rt = 0; rt < RT; ++rt) {
      res[rt] = dyres[rt];
      f32x4 m = dyres[rt];
      if constexpr (DROP) {  // the conv pair sees the gradient through the conv2-output dropout; the skip does not
        const f32x4 f = drop_factors(p.drop, dkey, dstep, (uint32_t)(p.drop_site0 + 3 * li + 2), (long)b * L + rt * 16, SL_D, col, lane);
#pragma unroll
        for (int r = 0; r < 4; ++r) m[r] *= f[r];
      }
#pragma unroll
      for (int r = 0; r < 4; ++r) xb[(rt * 16 + fq * 4 + r) * SL_XP + col] = (__bf16)m[r];
    }
    SLB_MARK(1);
    __syncthreads();  // d pre-norm-2 image complete
    SLB_MARK(2);
    save_image_as(xb, SL_XP, SL_D, p.dpre2, lrow * SL_D, L, tid, p.bf16_grads);

    // ================= conv2^T + activation' : dz (wave = column tiles wave, wave + 8, ...) =================
    SLB_LOCAL();
    {
      const long zs = lrow * F;  // (element offset into the z / h slab: fp32 or, flags bit 2 with z, bf16)
#pragma unroll 1
      for (int ct = wave; ct < F / 16; ct += SL_NW) {
        bf16x8 wf[4];
#pragma unroll
        for (int kk = 0; kk < 4; ++kk) wf[kk] = ld_wfrag(w_2t, ct * 4 + kk, lane);
        f32x4 zz[RT], acc[RT];
#pragma unroll
        for (int rt = 0; rt < RT; ++rt)
#pragma unroll
          for (int r = 0; r < 4; ++r) zz[rt][r] = ld_save1(p.zsrc, zs + (long)min(rt * 16 + fq * 4 + r, L - 1) * F + ct * 16 + fr, p.z_bf16);
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
            const f32x4 f = drop_factors(p.drop, dkey, dstep, (uint32_t)(p.drop_site0 + 3 * li + 1), (long)b * L + rt * 16, F,
                                         ct * 16 + fr, lane);
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
    SLB_MARK(3);
    __syncthreads();  // dz image complete
    SLB_MARK(4);
    save_image_as(hb, HP, F, p.dz, lrow * F, L, tid, p.bf16_grads);

    // ================= conv1^T + skip, norm1 backward =================
    SLB_LOCAL();
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
    stack_ln_bwd<RT>(dyres, p.xhat1, lrow * SL_D + col, p.norm_bf16, p.rstd1 + lrow, vec[col], p.dg1[li] + col, p.db1[li] + col, L, part,
                     stat, wave, lane);  // (its barriers fence the xb reads of the dz phase)
#pragma unroll
    for (int rt = 0; rt < RT; ++rt) {
      res[rt] = dyres[rt];
      f32x4 m = dyres[rt];
      if constexpr (DROP) {  // attention-output dropout
        const f32x4 f = drop_factors(p.drop, dkey, dstep, (uint32_t)(p.drop_site0 + 3 * li), (long)b * L + rt * 16, SL_D, col, lane);
#pragma unroll
        for (int r = 0; r < 4; ++r) m[r] *= f[r];
      }
#pragma unroll
      for (int r = 0; r < 4; ++r) xb[(rt * 16 + fq * 4 + r) * SL_XP + col] = (__bf16)m[r];
    }
    SLB_MARK(5);
    __syncthreads();  // d pre-norm-1 image complete
    SLB_MARK(6);
    save_image_as(xb, SL_XP, SL_D, p.dpre1, lrow * SL_D, L, tid, p.bf16_grads);

    // ================= out-projection^T: dC of head `wave` =================
    SLB_LOCAL();
    f32x4 dc[RT];
    {
      bf16x8 wf[4];
#pragma unroll
      for (int kk = 0; kk < 4; ++kk) wf[kk] = ld_wfrag(w_ot, wave * 4 + kk, lane);
#pragma unroll
      for (int rt = 0; rt < RT; ++rt) {
        dc[rt] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int kk = 0; kk < 4; ++kk)
          dc[rt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ld_frag(xb + (rt * 16 + fr) * SL_XP + kk * 32 + fq * 8), wf[kk], dc[rt], 0, 0, 0);
      }
    }
    // the attention phase needs only wave-private LDS until it writes the dq image, which aliases xb / hb
    // q | k | v as the forward saved them: an fp32 slab or (RfSeqStackBwd.flags bit 1) a bf16 slab -- the same bf16 operands
    const int qbf = p.qkv_bf16;
    const void* qkv_g = qbf ? static_cast<const void*>(reinterpret_cast<const __bf16*>(p.qkv) + lrow * (3 * SL_D))
                            : static_cast<const void*>(p.qkv + lrow * (3 * SL_D));
    {
      const int32_t* top_g = p.top + (((long)li * p.B + b) * SL_H + wave) * u;
      if (lane < 32) top_l[lane] = top_g[min(lane, u - 1)];
      // dC^T image: 4 consecutive rows of channel fr per store
#pragma unroll
      for (int rt = 0; rt < RT; ++rt) {
        const bf16x4 v4 = {(__bf16)dc[rt][0], (__bf16)dc[rt][1], (__bf16)dc[rt][2], (__bf16)dc[rt][3]};
        *reinterpret_cast<bf16x4*>(dT + fr * DTP + rt * 16 + fq * 4) = v4;
      }
      wave_sync_lds();
    }
    // which rows are selected (bit masks, wave-uniform)
    SLB_LOCAL();
    unsigned long long sel_lo, sel_hi;
    {
      const int q1 = lane, q2 = lane + 64;
      bool s1 = false, s2 = false;
#pragma unroll
      for (int c = 0; c < 8; ++c) {
        const int4 tt = *reinterpret_cast<const int4*>(top_l + 4 * c);
        s1 |= (tt.x == q1) | (tt.y == q1) | (tt.z == q1) | (tt.w == q1);
        s2 |= (tt.x == q2) | (tt.y == q2) | (tt.z == q2) | (tt.w == q2);
      }
      sel_lo = __ballot(s1);
      sel_hi = __ballot(s2);
    }
    // lazy rows: ctx[q] = mean_s V[s]  =>  every key receives (1/L) * sum of the unselected rows' dC
    float lz = 0.f;
#pragma unroll
    for (int rt = 0; rt < RT; ++rt)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int row = rt * 16 + fq * 4 + r;
        const bool sel = rt < 4 ? ((sel_lo >> row) & 1ull) != 0 : ((sel_hi >> (row - 64)) & 1ull) != 0;
        lz += sel ? 0.f : dc[rt][r];  // (rows >= L: dC is zero)
      }
    lz += __shfl_xor(lz, 16);
    lz += __shfl_xor(lz, 32);
    lz *= inv_L;

    SLB_MARK(7);
    // ---- P = softmax(scale Q_sel K^T), recomputed as in the forward ----
    SLB_LOCAL();
    f32x4 P[2][RT], dS[2][RT];
    {
      bf16x8 kb[RT];
#pragma unroll
      for (int ct = 0; ct < RT; ++ct) {
        if (fq < 2) {
          kb[ct] = ld_qkv8(qkv_g, (long)min(ct * 16 + fr, L - 1) * (3 * SL_D) + SL_D + wave * 16 + fq * 8, qbf);
        } else {
          kb[ct] = zero_frag();
        }
      }
#pragma unroll
      for (int t2 = 0; t2 < 2; ++t2) {
        bf16x8 qa = zero_frag();
        if (fq < 2) {
          qa = ld_qkv8(qkv_g, (long)top_l[t2 * 16 + fr] * (3 * SL_D) + wave * 16 + fq * 8, qbf);
        }
        float mx[4] = {-INFINITY, -INFINITY, -INFINITY, -INFINITY};
#pragma unroll
        for (int ct = 0; ct < RT; ++ct) {
          P[t2][ct] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(qa, kb[ct], f32x4{0.f, 0.f, 0.f, 0.f}, 0, 0, 0);
          const bool live = ct * 16 + fr < L;
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            P[t2][ct][r] = live ? P[t2][ct][r] * p.scale : -INFINITY;
            mx[r] = fmaxf(mx[r], P[t2][ct][r]);
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
            P[t2][ct][r] = __expf(P[t2][ct][r] - mx[r]);
            sum[r] += P[t2][ct][r];
          }
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const float inv = __builtin_amdgcn_rcpf(row16_sum(sum[r]));
          sum[r] = t2 * 16 + fq * 4 + r < u ? inv : 0.f;  // rows >= u of the selection tile: no probabilities
        }
#pragma unroll
        for (int ct = 0; ct < RT; ++ct)
#pragma unroll
          for (int r = 0; r < 4; ++r) P[t2][ct][r] *= sum[r];
      }
    }
    SLB_MARK(8);
    // ---- dP = dC_sel V^T, dS ----
    SLB_LOCAL();
    {
      bf16x8 vb[RT];
#pragma unroll
      for (int ct = 0; ct < RT; ++ct) {
        if (fq < 2) {
          vb[ct] = ld_qkv8(qkv_g, (long)min(ct * 16 + fr, L - 1) * (3 * SL_D) + 2 * SL_D + wave * 16 + fq * 8, qbf);
        } else {
          vb[ct] = zero_frag();
        }
      }
#pragma unroll
      for (int t2 = 0; t2 < 2; ++t2) {
        bf16x8 dca = zero_frag();
        if (fq < 2) {
          const int trow = top_l[t2 * 16 + fr];
#pragma unroll
          for (int j = 0; j < 8; ++j) dca[j] = dT[(fq * 8 + j) * DTP + trow];
        }
        float rs[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int ct = 0; ct < RT; ++ct) {
          dS[t2][ct] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(dca, vb[ct], f32x4{0.f, 0.f, 0.f, 0.f}, 0, 0, 0);
#pragma unroll
          for (int r = 0; r < 4; ++r) rs[r] = fmaf(P[t2][ct][r], dS[t2][ct][r], rs[r]);
        }
#pragma unroll
        for (int r = 0; r < 4; ++r) rs[r] = row16_sum(rs[r]);
#pragma unroll
        for (int ct = 0; ct < RT; ++ct)
#pragma unroll
          for (int r = 0; r < 4; ++r) dS[t2][ct][r] = P[t2][ct][r] * (dS[t2][ct][r] - rs[r]) * p.scale;
      }
    }
    SLB_MARK(9);
    // (every wave must have finished reading xb -- out-projection^T -- before any wave writes the dq image)
    __syncthreads();
    SLB_MARK(10);
    // ---- dV = P^T dC_sel + lazy term ----
    SLB_LOCAL();
    {
#pragma unroll
      for (int t2 = 0; t2 < 2; ++t2)
#pragma unroll
        for (int ct = 0; ct < RT; ++ct) {
          const bf16x4 v4 = {(__bf16)P[t2][ct][0], (__bf16)P[t2][ct][1], (__bf16)P[t2][ct][2], (__bf16)P[t2][ct][3]};
          *reinterpret_cast<bf16x4*>(buf + (ct * 16 + fr) * TP + t2 * 16 + fq * 4) = v4;
        }
      const int4 ta = *reinterpret_cast<const int4*>(top_l + fq * 8), tb = *reinterpret_cast<const int4*>(top_l + fq * 8 + 4);
      bf16x8 dcb;  // B[k = selected i][col = e]: dC[top[i]][e]
      dcb[0] = dT[fr * DTP + ta.x]; dcb[1] = dT[fr * DTP + ta.y]; dcb[2] = dT[fr * DTP + ta.z]; dcb[3] = dT[fr * DTP + ta.w];
      dcb[4] = dT[fr * DTP + tb.x]; dcb[5] = dT[fr * DTP + tb.y]; dcb[6] = dT[fr * DTP + tb.z]; dcb[7] = dT[fr * DTP + tb.w];
      wave_sync_lds();
#pragma unroll
      for (int kt = 0; kt < RT; ++kt) {
        const f32x4 dv = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ld_frag(buf + (kt * 16 + fr) * TP + fq * 8), dcb,
                                                                 f32x4{0.f, 0.f, 0.f, 0.f}, 0, 0, 0);
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int key = kt * 16 + fq * 4 + r;
          dqi[key * QP + 2 * SL_D + col] = (__bf16)(key < L ? dv[r] + lz : 0.f);
        }
      }
      wave_sync_lds();
    }
    SLB_MARK(11);
    // ---- dK = dS^T Q_sel ----
    SLB_LOCAL();
    {
#pragma unroll
      for (int t2 = 0; t2 < 2; ++t2)
#pragma unroll
        for (int ct = 0; ct < RT; ++ct) {
          const bf16x4 v4 = {(__bf16)dS[t2][ct][0], (__bf16)dS[t2][ct][1], (__bf16)dS[t2][ct][2], (__bf16)dS[t2][ct][3]};
          *reinterpret_cast<bf16x4*>(buf + (ct * 16 + fr) * TP + t2 * 16 + fq * 4) = v4;
        }
      const int4 ta = *reinterpret_cast<const int4*>(top_l + fq * 8), tb = *reinterpret_cast<const int4*>(top_l + fq * 8 + 4);
      const long qc = wave * 16 + fr;  // B[k = selected i][col = e]: q[top[i]][e]
      bf16x8 qtb;
      qtb[0] = ld_qkv1(qkv_g, qc + (long)ta.x * (3 * SL_D), qbf); qtb[1] = ld_qkv1(qkv_g, qc + (long)ta.y * (3 * SL_D), qbf);
      qtb[2] = ld_qkv1(qkv_g, qc + (long)ta.z * (3 * SL_D), qbf); qtb[3] = ld_qkv1(qkv_g, qc + (long)ta.w * (3 * SL_D), qbf);
      qtb[4] = ld_qkv1(qkv_g, qc + (long)tb.x * (3 * SL_D), qbf); qtb[5] = ld_qkv1(qkv_g, qc + (long)tb.y * (3 * SL_D), qbf);
      qtb[6] = ld_qkv1(qkv_g, qc + (long)tb.z * (3 * SL_D), qbf); qtb[7] = ld_qkv1(qkv_g, qc + (long)tb.w * (3 * SL_D), qbf);
      wave_sync_lds();
#pragma unroll
      for (int kt = 0; kt < RT; ++kt) {
        const f32x4 dk = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ld_frag(buf + (kt * 16 + fr) * TP + fq * 8), qtb,
                                                                 f32x4{0.f, 0.f, 0.f, 0.f}, 0, 0, 0);
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int key = kt * 16 + fq * 4 + r;
          dqi[key * QP + SL_D + col] = (__bf16)(key < L ? dk[r] : 0.f);
        }
      }
      wave_sync_lds();
    }
    SLB_MARK(12);
    // ---- dQ[top] = dS K ----
    SLB_LOCAL();
    {
#pragma unroll
      for (int t2 = 0; t2 < 2; ++t2)
#pragma unroll
        for (int ct = 0; ct < RT; ++ct)
#pragma unroll
          for (int r = 0; r < 4; ++r) buf[(t2 * 16 + fq * 4 + r) * DTP + ct * 16 + fr] = (__bf16)dS[t2][ct][r];
      if constexpr (KS32 > LP) {
        for (int i = lane; i < 32 * (KS32 - LP); i += 64) buf[(i / (KS32 - LP)) * DTP + LP + i % (KS32 - LP)] = (__bf16)0.f;
      }
      const long kc = SL_D + wave * 16 + fr;  // B[k = key][col = e]: k[key][e]
      bf16x8 ktb[KSTEPS];
#pragma unroll
      for (int ks = 0; ks < KSTEPS; ++ks)
#pragma unroll
        for (int j = 0; j < 8; ++j) ktb[ks][j] = ld_qkv1(qkv_g, kc + (long)min(ks * 32 + fq * 8 + j, L - 1) * (3 * SL_D), qbf);
      // q part of the dq image: zeros, then the selected rows
      {
        typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
        const u32x4 z4 = {0u, 0u, 0u, 0u};
        for (int q = lane; q < LP; q += 64) {
          *reinterpret_cast<u32x4*>(dqi + q * QP + wave * 16) = z4;
          *reinterpret_cast<u32x4*>(dqi + q * QP + wave * 16 + 8) = z4;
        }
      }
      wave_sync_lds();
#pragma unroll
      for (int t2 = 0; t2 < 2; ++t2) {
        f32x4 dq = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int ks = 0; ks < KSTEPS; ++ks)
          dq = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ld_frag(buf + (t2 * 16 + fr) * DTP + ks * 32 + fq * 8), ktb[ks], dq, 0, 0, 0);
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int i = t2 * 16 + fq * 4 + r;
          if (i < u) dqi[top_l[i] * QP + col] = (__bf16)dq[r];
        }
      }
    }
    // packed projection fragments travel across the barrier
    bf16x8 wfp[12];
#pragma unroll
    for (int kk = 0; kk < 12; ++kk) wfp[kk] = ld_wfrag(w_qkvt, wave * 12 + kk, lane);
    SLB_MARK(13);
    __syncthreads();  // d q|k|v image complete
    SLB_MARK(14);
    save_image_as(dqi, QP, 3 * SL_D, p.dqkv, lrow * (3 * SL_D), L, tid, p.bf16_grads);

    // ================= packed q|k|v projection^T + skip =================
    SLB_LOCAL();
#pragma unroll
    for (int rt = 0; rt < RT; ++rt) {
      f32x4 acc = res[rt];
#pragma unroll
      for (int kk = 0; kk < 12; ++kk)
        acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ld_frag(dqi + (rt * 16 + fr) * QP + kk * 32 + fq * 8), wfp[kk], acc, 0, 0, 0);
#pragma unroll
      for (int r = 0; r < 4; ++r) dyres[rt][r] = rt * 16 + fq * 4 + r < L ? acc[r] : 0.f;
    }
    SLB_MARK(15);
    // (the next layer writes xb only behind the two barriers of its norm2 backward)
  }

  {
    float* dxg = p.dx + (long)b * L * SL_D + col;
#pragma unroll
    for (int rt = 0; rt < RT; ++rt)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int row = rt * 16 + fq * 4 + r;
        if (row < L) dxg[row * SL_D] = dyres[rt][r];
      }
  }
}

template <int RT>
size_t stack_bwd_lds_bytes(int F) {
  const int LP = 16 * RT;
  const int a = LP * SL_XP + LP * (F + 8), q = LP * (3 * SL_D + 8);
  const int region = (((a > q ? a : q) * 2) + 15) & ~15;
  return (size_t)region + SL_NW * 10240 + (size_t)LP * SL_NW * 8 + (size_t)LP * 16;
}

inline bool al16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; }

}  // namespace

#ifdef RF_SL_TIMING
extern "C" void* rf_slb_timing_address() {
  void* a = nullptr;
  (void)hipGetSymbolAddress(&a, HIP_SYMBOL(rf_slb_timing));
  return a;
}
#endif

extern "C" int64_t rf_seqlayer_bwd_pack_bytes(int d_ff) { return bwd_pack_offsets(d_ff).total; }

extern "C" int rf_seqlayer_bwd(const RfSeqStackBwd* st_, const float* dy, float* dx, int B, int L, int d_model, int n_heads,
                               int d_ff, int act, int n_top, float scale, float drop_p, const void* rng_state, int drop_site0,
                               void* stream) {
  RF_REQUIRE(st_ && dy && dx && B > 0 && st_->n_layers > 0 && st_->n_layers <= RF_SEQLAYER_MAX_LAYERS);
  if (!rf_seqlayer_supported(L, d_model, n_heads, d_ff, 1, n_top)) {
    rf_g_last_error = "rf_seqlayer_bwd: shape outside the fused kernel's range";
    return RF_EUNSUPPORTED;
  }
  const RfSeqStackBwd& s = *st_;
  RF_REQUIRE(s.wpack && al16(s.wpack) && s.wpack_stride >= rf_seqlayer_bwd_pack_bytes(d_ff) && s.wpack_stride % 16 == 0);
  RF_REQUIRE(s.qkv && s.xhat1 && s.rstd1 && s.zsrc && s.xhat2 && s.rstd2 && s.top && s.dpre2 && s.dz && s.dpre1 && s.dqkv);
  RF_REQUIRE(al16(s.qkv) && al16(s.dpre2) && al16(s.dz) && al16(s.dpre1) && al16(s.dqkv));
  SeqStackBwdP p{};
  p.dy = dy; p.dx = dx;
  p.wpack = static_cast<const unsigned char*>(s.wpack); p.wpack_stride = s.wpack_stride;
  p.qkv = s.qkv; p.xhat1 = s.xhat1; p.rstd1 = s.rstd1; p.zsrc = s.zsrc; p.xhat2 = s.xhat2; p.rstd2 = s.rstd2; p.top = s.top;
  p.dpre2 = s.dpre2; p.dz = s.dz; p.dpre1 = s.dpre1; p.dqkv = s.dqkv;
  for (int i = 0; i < s.n_layers; ++i) {
    RF_REQUIRE(s.dgamma1[i] && s.dbeta1[i] && s.dgamma2[i] && s.dbeta2[i]);
    p.dg1[i] = s.dgamma1[i]; p.db1[i] = s.dbeta1[i]; p.dg2[i] = s.dgamma2[i]; p.db2[i] = s.dbeta2[i];
  }
  p.B = B; p.L = L; p.F = d_ff; p.n_layers = s.n_layers; p.act = act; p.n_top = n_top; p.scale = scale;
  p.bf16_grads = s.flags & 1;
  p.qkv_bf16 = (s.flags & 2) ? 1 : 0;
  p.norm_bf16 = (s.flags & 4) ? 1 : 0;
  p.z_bf16 = (s.flags & 8) ? 1 : 0;
  RF_REQUIRE(drop_p >= 0.f && drop_p < 1.f && (drop_p == 0.f || rng_state));
  p.drop = make_drop_cfg(rng_state, nullptr, 0, drop_p);
  p.drop_site0 = drop_site0;
  const bool drop = p.drop.state != nullptr;
  const hipStream_t st = static_cast<hipStream_t>(stream);
#define RF_SLB_GO(RT_, DROP_)                                                                                       \
  do {                                                                                                              \
    static bool attr = false;                                                                                       \
    if (!attr) {                                                                                                    \
      (void)hipFuncSetAttribute(reinterpret_cast<const void*>(seq_stack_bwd_kernel<RT_, DROP_>),                    \
                                hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);                            \
      attr = true;                                                                                                  \
    }                                                                                                               \
    RF_LAUNCH((seq_stack_bwd_kernel<RT_, DROP_>), dim3(B), dim3(SL_NT), stack_bwd_lds_bytes<RT_>(d_ff), st, p);     \
  } while (0)
  if (L <= 48) {
    if (drop) RF_SLB_GO(3, true);
    else RF_SLB_GO(3, false);
  } else {
    if (drop) RF_SLB_GO(5, true);
    else RF_SLB_GO(5, false);
  }
#undef RF_SLB_GO
  RF_CHECK_LAUNCH();
  return RF_OK;
}
