// 3x3 / stride 1 / pad 1 convolution over NHWC (fp32 or bf16) activations on the bf16 matrix cores, for the
// residual blocks of the frozen HRNet-16 trunk (C_in = C_out in {16,32,64,128}; 256->16 transition).
//
// "Raster window" formulation.  NHWC makes the (n,h,w) raster order of pixels contiguous in memory, and a
// 3x3 stencil around raster index m only touches raster indices m + (kh-1)*W + (kw-1).  So one workgroup
// (4 waves) takes 128 consecutive output pixels and stages the CONTIGUOUS input span
// [m0 - W - 1, m0 + 127 + W + 1] x C_in into LDS exactly once (coalesced 16-B loads, fp32 -> bf16 on the
// way in): no im2col, no 9x re-read through the texture path, no index arithmetic per element.  Each MFMA A
// fragment (16 pixels x 8 channels per lane group) is a single ds_read_b128 out of that window at a
// per-tap constant offset; image borders (and tiles that straddle two images) are handled by a per-lane
// validity mask.  Weights are tiny ([C_out][9][C_in] bf16, pre-folded with BatchNorm) and shared by every
// workgroup, so B fragments come straight from global memory (L1/L2 hits) in fragment order, software-prefetched
// a few k-steps ahead.  Epilogue: + bias (+ residual) -> ReLU -> NHWC in the maps' storage type.
//
// v_mfma_f32_16x16x32_bf16: a k-step covers 32 input channels of one tap, or (C_in = 16) both halves of
// two taps (the packed weights then carry a zero 10th tap).
#include "common.h"

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

#ifdef RF_CONV_TIMING
__device__ unsigned long long rf_conv_timing[8 * 4096];
#define CV_MARK(k) do { if (threadIdx.x == 0 && blockIdx.x < 4096) rf_conv_timing[blockIdx.x * 8 + (k)] = __builtin_readcyclecounter(); } while (0)
extern "C" void* rf_conv_timing_address() {
  void* a = nullptr;
  (void)hipGetSymbolAddress(&a, HIP_SYMBOL(rf_conv_timing));
  return a;
}
#else
#define CV_MARK(k) do {} while (0)
#endif

namespace {

constexpr int NT = 256;
#ifndef RF_CONV_U
#define RF_CONV_U 4
#endif
constexpr int CONV_U = RF_CONV_U;  // 16-B loads a thread keeps in flight while it stages a channel slice
constexpr int TILE = 128;  // output pixels per workgroup (4 waves x 2 MFMA row tiles)

// How many k-steps the weight fragments run ahead of the MFMAs (register ring).  A fragment comes from L2, ~1 us away under
// load; with a distance of 4 the 256 -> 16 transition's loop ran at ~400 cycles per k-step (tools/conv_phase_probe.py: 2 MFMAs
// of 32 cycles each per step, the rest is waiting for weights).  Few column tiles = few registers per step: fetch deeper
// (one / two column tiles: up to 24 fragments = 96 VGPRs in flight, whole slices at once when they fit; with four or more
// column tiles a step already holds 8+ MFMAs and the deeper ring only cost occupancy: 99 -> 108 us for layer1's 64 -> 64).
constexpr int pf_depth(int ksteps, int ntl) {
  const int want = ntl <= 2 ? 24 / ntl : 4;
  return ksteps < want ? ksteps : want;
}

extern __shared__ __attribute__((aligned(16))) __bf16 rf_conv_win[];

// One 128-pixel tile (`tile`) of one convolution; all threads of the workgroup take the same path.
// CH < CIN (the 256 -> 16 transition): the window is staged CH input channels at a time and the accumulators carry over --
// the whole 256-channel window of a 56-wide map is 128 KB of LDS (ONE four-wave workgroup per CU staging 124 KB for 144
// MFMAs per wave: 433 us for 405 MB at C5, 1 TB/s); a 64-channel slice is 35 KB, four workgroups per CU overlap each
// other's staging.  The packed weights are the same (k-step = tap x 32-channel group).
template <int CIN, int COUT, typename AT, int CH = CIN>
__device__ __forceinline__ void conv3x3_body(const AT* __restrict__ x, const __bf16* __restrict__ wt,
                                             const float* __restrict__ bias, const AT* __restrict__ residual,
                                             AT* __restrict__ y, int total, int H, int W, int relu, int tile) {
  static_assert(CIN % CH == 0 && (CH == CIN || CH % 32 == 0), "channel slices are whole 32-channel k-groups");
  constexpr int LDC = CH + 8;                        // LDS pixel pitch (bf16): odd multiple of 16 B
  constexpr int NCH = CIN / CH;                      // channel slices
  constexpr int KSTEPS = (CIN == 16) ? 5 : 9 * (CH / 32);  // k-steps per slice
  constexpr int NTL = COUT / 16;                     // MFMA column tiles
  __bf16* win = rf_conv_win;

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int fr = lane & 15, fq = lane >> 4;
  const long m0 = (long)tile * TILE;
  const int span = TILE + 2 * W + 2;                 // staged pixels
  const long s0 = m0 - W - 1;                        // raster index of window pixel 0

  CV_MARK(0);
  // ---- stage (a channel slice of) the input window (contiguous pixels in memory) ----
  // 16-B loads (8 bf16 / 4 fp32 channels), four per thread in flight before the first LDS store: the plain
  // load -> store loop paid one memory round trip per iteration (49 of them for the 256-channel window)
  auto stage = [&](int ch) {
    constexpr int VEC = sizeof(AT) == 2 ? 8 : 4, VPP = CH / VEC, U = NCH > 1 ? CONV_U : 4;  // (slices: fewer, deeper batches)
    const int nvec = span * VPP;
    for (int base = 0; base < nvec; base += NT * U) {
      float4 raw[U];
      int at[U];
#pragma unroll
      for (int u = 0; u < U; ++u) {
        const int i = base + tid + u * NT;
        const int px = i / VPP, c = (i % VPP) * VEC;
        const long g = s0 + px;
        at[u] = i < nvec ? px * LDC + c : -1;
        raw[u] = make_float4(0.f, 0.f, 0.f, 0.f);
        if (i < nvec && g >= 0 && g < total) raw[u] = *reinterpret_cast<const float4*>(x + g * CIN + ch * CH + c);
      }
#pragma unroll
      for (int u = 0; u < U; ++u) {
        if (at[u] < 0) continue;
        if constexpr (sizeof(AT) == 2) {  // bf16 maps: the window is a plain copy
          *reinterpret_cast<float4*>(win + at[u]) = raw[u];
        } else {
          bf16x4 o = {(__bf16)raw[u].x, (__bf16)raw[u].y, (__bf16)raw[u].z, (__bf16)raw[u].w};
          *reinterpret_cast<bf16x4*>(win + at[u]) = o;
        }
      }
    }
  };
  stage(0);

  // ---- this lane's two output pixels (one per MFMA row tile) and their border masks ----
  int pl[2];          // window-relative index of the centre pixel
  unsigned vmask[2];  // bit t set <=> tap t reads inside the image
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    const int p = wave * 32 + i * 16 + fr;
    const long m = m0 + p;
    pl[i] = p + W + 1;
    unsigned msk = 0;
    if (m < total) {
      const int w_ = (int)(m % W), h_ = (int)((m / W) % H);
#pragma unroll
      for (int t = 0; t < 9; ++t) {
        const int hi = h_ + t / 3 - 1, wi = w_ + t % 3 - 1;
        if (hi >= 0 && hi < H && wi >= 0 && wi < W) msk |= 1u << t;
      }
    }
    vmask[i] = msk;
  }

  f32x4 acc[2][NTL];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < NTL; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

  // B fragment of k-step s, column tile j.  The weights arrive in FRAGMENT ORDER (rf_conv3x3_pack_bf16): the 64
  // lanes of a wave read one contiguous 1-KB block.  Read from the [cout][tap][cin] tensor, the same fragment is 16
  // rows 2*9*CIN bytes apart -- 64 cache-line look-ups per load instruction, and with eight such loads per wave and
  // k-step the L1 tag pipe, not the matrix cores, set the pace of the loop (1.9k cycles per step measured).
  auto ldb = [&](int s, int j) -> bf16x8 {
    return *reinterpret_cast<const bf16x8*>(wt + ((long)(s * NTL + j) * 64 + lane) * 8);
  };

  // Weight fragments run PF k-steps ahead of the MFMAs in a register ring: every step's fragments come from L2
  // (~1 us away when little else is in flight) and the small maps (4x4, 7x7: a few dozen workgroups, one per CU)
  // have nothing else to hide that behind -- with a distance of one the loop ran at one memory round trip per step.
  constexpr int PF = pf_depth(KSTEPS, NTL);
  // k-step s of channel slice ch in the packed order (tap-major over ALL 32-channel groups of CIN)
  auto gstep = [&](int s, int ch) -> int {
    if constexpr (NCH == 1) return s;
    else return (s / (CH / 32)) * (CIN / 32) + ch * (CH / 32) + s % (CH / 32);
  };
  bf16x8 bq[PF][NTL];
#pragma unroll 1
  for (int ch = 0; ch < NCH; ++ch) {
    if (ch > 0) {
      __syncthreads();  // every wave is done reading the previous slice
      stage(ch);
    }
#pragma unroll
    for (int d = 0; d < PF; ++d)
#pragma unroll
      for (int j = 0; j < NTL; ++j) bq[d][j] = ldb(gstep(d, ch), j);
    CV_MARK(1);
    __syncthreads();  // window staged
    CV_MARK(2);

#pragma unroll 1
    for (int s0 = 0; s0 < KSTEPS; s0 += PF) {
#pragma unroll
      for (int d = 0; d < PF; ++d) {
        const int s = s0 + d;
        if (s < KSTEPS) {
          int tap, c0;
          if constexpr (CIN == 16) { tap = 2 * s + (fq >> 1); c0 = (fq & 1) * 8; }
          else { tap = s / (CH / 32); c0 = (s % (CH / 32)) * 32 + fq * 8; }
          const int toff = (tap < 9) ? (tap / 3 - 1) * W + (tap % 3 - 1) : 0;
          bf16x8 a[2];
#pragma unroll
          for (int i = 0; i < 2; ++i) {
            const bool ok = tap < 9 && ((vmask[i] >> tap) & 1u);
            bf16x8 v = *reinterpret_cast<const bf16x8*>(win + (pl[i] + (ok ? toff : 0)) * LDC + c0);
            if (!ok) v = bf16x8{0, 0, 0, 0, 0, 0, 0, 0};
            a[i] = v;
          }
#pragma unroll
          for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < NTL; ++j)
              acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[i], bq[d][j], acc[i][j], 0, 0, 0);
          if (s + PF < KSTEPS) {
#pragma unroll
            for (int j = 0; j < NTL; ++j) bq[d][j] = ldb(gstep(s + PF, ch), j);
          }
        }
      }
    }
  }

  CV_MARK(3);
  // ---- epilogue ----
  // The C/D fragment holds (4 pixels x 1 channel) per lane and tile: finishing it from the registers meant one
  // 2- or 4-byte residual load and store per element (and a bias load each) -- a quarter to a third of the kernel
  // (tools/conv_phase_probe.py).  Each wave instead drops one 16-pixel row tile at a time into its own fp32 patch
  // of the (now dead) window and finishes it 8 channels per lane: 16-B residual loads, 16-B (bf16) / 2x16-B stores.
  constexpr int SP = COUT + 4;                       // patch pitch (floats)
  float* patch = reinterpret_cast<float*>(win) + wave * 16 * SP;
  __syncthreads();  // every wave is done reading the window
#pragma unroll
  for (int i = 0; i < 2; ++i) {
#pragma unroll
    for (int j = 0; j < NTL; ++j)
#pragma unroll
      for (int r = 0; r < 4; ++r) patch[(fq * 4 + r) * SP + j * 16 + fr] = acc[i][j][r];
    __syncthreads();
    constexpr int VPP = COUT / 8;                    // 8-channel vectors per pixel
    for (int v = lane; v < 16 * VPP; v += 64) {
      const int px = v / VPP, c = (v % VPP) * 8;
      const long m = m0 + wave * 32 + i * 16 + px;
      if (m < total) {
        float4 lo = *reinterpret_cast<const float4*>(patch + px * SP + c);
        float4 hi = *reinterpret_cast<const float4*>(patch + px * SP + c + 4);
        const float4 b0 = *reinterpret_cast<const float4*>(bias + c), b1 = *reinterpret_cast<const float4*>(bias + c + 4);
        lo.x += b0.x; lo.y += b0.y; lo.z += b0.z; lo.w += b0.w;
        hi.x += b1.x; hi.y += b1.y; hi.z += b1.z; hi.w += b1.w;
        if (residual) {
          const float4 r0 = act_ld4(residual + m * COUT + c), r1 = act_ld4(residual + m * COUT + c + 4);
          lo.x += r0.x; lo.y += r0.y; lo.z += r0.z; lo.w += r0.w;
          hi.x += r1.x; hi.y += r1.y; hi.z += r1.z; hi.w += r1.w;
        }
        if (relu) {
          lo.x = fmaxf(lo.x, 0.f); lo.y = fmaxf(lo.y, 0.f); lo.z = fmaxf(lo.z, 0.f); lo.w = fmaxf(lo.w, 0.f);
          hi.x = fmaxf(hi.x, 0.f); hi.y = fmaxf(hi.y, 0.f); hi.z = fmaxf(hi.z, 0.f); hi.w = fmaxf(hi.w, 0.f);
        }
        act_st4(y + m * COUT + c, lo);
        act_st4(y + m * COUT + c + 4, hi);
      }
    }
    if (i == 0) __syncthreads();  // the patch is rewritten by the second row tile
  }
  CV_MARK(4);
}

// ---------------------------------------------------------------------------------------------------------------
// BasicBlock pair (round 4):  y = relu(conv2(relu(conv1(x) + b1)) + b2 + x)  (hrnetv2.py:45-61) in ONE launch on bf16
// maps, the intermediate map in LDS.  Same raster-window formulation: a workgroup owns 128 consecutive output pixels m0 ..
// m0 + 127; conv2 needs the intermediate on the span [m0 - W - 1, m0 + 127 + W + 1], conv1 therefore runs on that whole span
// (1.2 - 1.9 x the tile's conv1 work, matrix-core time that was idle anyway) from an input window of TILE + 4 W + 4 pixels staged
// once.  The residual comes out of the staged window.  HBM: x read once, y written once -- the two-launch form writes and re-reads
// the intermediate and reads x twice.  Border taps are masked per OUTPUT pixel exactly as in conv3x3_body, so intermediate values at
// raster positions outside the image (or in the neighbouring image) are never read.  The arithmetic is the two-launch path's
// (same fragments, same k order, intermediate rounded to bf16): bit-identical results.
template <int C>
__device__ __forceinline__ void conv3x3_pair_body(const __bf16* __restrict__ x, const __bf16* __restrict__ w1,
                                                  const float* __restrict__ b1, const __bf16* __restrict__ w2,
                                                  const float* __restrict__ b2, __bf16* __restrict__ y, int total, int H, int W,
                                                  int tile) {
  constexpr int LDC = C + 8;
  constexpr int KSTEPS = (C == 16) ? 5 : 9 * (C / 32);
  constexpr int NTL = C / 16;
  constexpr int PF = NTL >= 8 ? 2 : pf_depth(KSTEPS, NTL);  // (128 channels: 64 accumulator registers + 32 per ring stage)
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int fr = lane & 15, fq = lane >> 4;
  const long m0 = (long)tile * TILE;
  const int span_m = TILE + 2 * W + 2, span_x = TILE + 4 * W + 4;
  __bf16* xw = rf_conv_win;                 // input window: raster [m0 - 2W - 2, m0 + 127 + 2W + 2]
  __bf16* mid = rf_conv_win + span_x * LDC; // intermediate:  raster [m0 - W - 1,  m0 + 127 + W + 1]
  const long sx0 = m0 - 2 * W - 2;

  // ---- stage the input window ----
  {
    constexpr int VPP = C / 8, U = 4;
    const int nvec = span_x * VPP;
    for (int base = 0; base < nvec; base += NT * U) {
      float4 raw[U];
      int at[U];
#pragma unroll
      for (int u = 0; u < U; ++u) {
        const int i = base + tid + u * NT;
        const int px = i / VPP, c = (i % VPP) * 8;
        const long g = sx0 + px;
        at[u] = i < nvec ? px * LDC + c : -1;
        raw[u] = make_float4(0.f, 0.f, 0.f, 0.f);
        if (i < nvec && g >= 0 && g < total) raw[u] = *reinterpret_cast<const float4*>(x + g * C + c);
      }
#pragma unroll
      for (int u = 0; u < U; ++u)
        if (at[u] >= 0) *reinterpret_cast<float4*>(xw + at[u]) = raw[u];
    }
  }
  auto tap_mask = [&](long m) -> unsigned {  // bit t set <=> tap t of output pixel m reads inside its image
    unsigned msk = 0;
    if (m >= 0 && m < total) {
      const int w_ = (int)(m % W), h_ = (int)((m / W) % H);
#pragma unroll
      for (int t = 0; t < 9; ++t) {
        const int hi = h_ + t / 3 - 1, wi = w_ + t % 3 - 1;
        if (hi >= 0 && hi < H && wi >= 0 && wi < W) msk |= 1u << t;
      }
    }
    return msk;
  };
  auto ldb = [&](const __bf16* wt, int s, int j) -> bf16x8 {
    return *reinterpret_cast<const bf16x8*>(wt + ((long)(s * NTL + j) * 64 + lane) * 8);
  };
  // one pass of the k loop over two row tiles whose centre pixels sit at window indices pl[0], pl[1] of `win`
  auto kloop = [&](const __bf16* win, const __bf16* wt, const int (&pl)[2], const unsigned (&vmask)[2], f32x4 (&acc)[2][NTL]) {
    bf16x8 bq[PF][NTL];
#pragma unroll
    for (int d = 0; d < PF; ++d)
#pragma unroll
      for (int j = 0; j < NTL; ++j) bq[d][j] = ldb(wt, d, j);
#pragma unroll 1
    for (int s0 = 0; s0 < KSTEPS; s0 += PF) {
#pragma unroll
      for (int d = 0; d < PF; ++d) {
        const int s = s0 + d;
        if (s < KSTEPS) {
          int tap, c0;
          if constexpr (C == 16) { tap = 2 * s + (fq >> 1); c0 = (fq & 1) * 8; }
          else { tap = s / (C / 32); c0 = (s % (C / 32)) * 32 + fq * 8; }
          const int toff = (tap < 9) ? (tap / 3 - 1) * W + (tap % 3 - 1) : 0;
          bf16x8 a[2];
#pragma unroll
          for (int i = 0; i < 2; ++i) {
            const bool ok = tap < 9 && ((vmask[i] >> tap) & 1u);
            bf16x8 v = *reinterpret_cast<const bf16x8*>(win + (pl[i] + (ok ? toff : 0)) * LDC + c0);
            if (!ok) v = bf16x8{0, 0, 0, 0, 0, 0, 0, 0};
            a[i] = v;
          }
#pragma unroll
          for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < NTL; ++j)
              acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[i], bq[d][j], acc[i][j], 0, 0, 0);
          if (s + PF < KSTEPS) {
#pragma unroll
            for (int j = 0; j < NTL; ++j) bq[d][j] = ldb(wt, s + PF, j);
          }
        }
      }
    }
  };
  __syncthreads();  // input window staged

  // ---- conv1 + bias + ReLU on the whole intermediate span, two row tiles per wave and pass ----
  const int n_mt = (span_m + 15) >> 4;
  for (int t0 = wave * 2; t0 < n_mt; t0 += 8) {
    int pl[2];
    unsigned vmask[2];
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const int p = min((t0 + i) * 16 + fr, span_m - 1);  // (clamped: the lanes of a partial last tile recompute its last pixel)
      pl[i] = p + W + 1;                                  // the intermediate pixel's index in the INPUT window
      vmask[i] = tap_mask(m0 - W - 1 + p);
    }
    f32x4 acc[2][NTL];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int j = 0; j < NTL; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
    kloop(xw, w1, pl, vmask, acc);
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int j = 0; j < NTL; ++j) {
        const float bj = b1[j * 16 + fr];
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int p = (t0 + i) * 16 + fq * 4 + r;
          if (p < span_m) mid[p * LDC + j * 16 + fr] = (__bf16)fmaxf(acc[i][j][r] + bj, 0.f);
        }
      }
  }
  __syncthreads();  // intermediate complete

  // ---- conv2 on the tile's 128 pixels ----
  int pl[2];
  unsigned vmask[2];
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    const int p = wave * 32 + i * 16 + fr;
    pl[i] = p + W + 1;
    vmask[i] = tap_mask(m0 + p < total ? m0 + p : -1);
  }
  f32x4 acc[2][NTL];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < NTL; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
  kloop(mid, w2, pl, vmask, acc);

  // ---- epilogue: + bias + x (from the staged window) -> ReLU -> y; fp32 patch per wave over the dead intermediate ----
  constexpr int SP = C + 4;
  float* patch = reinterpret_cast<float*>(mid) + wave * 16 * SP;
  __syncthreads();  // every wave is done reading the intermediate
#pragma unroll
  for (int i = 0; i < 2; ++i) {
#pragma unroll
    for (int j = 0; j < NTL; ++j)
#pragma unroll
      for (int r = 0; r < 4; ++r) patch[(fq * 4 + r) * SP + j * 16 + fr] = acc[i][j][r];
    __syncthreads();
    constexpr int VPP = C / 8;
    for (int v = lane; v < 16 * VPP; v += 64) {
      const int px = v / VPP, c = (v % VPP) * 8;
      const int p = wave * 32 + i * 16 + px;
      const long m = m0 + p;
      if (m < total) {
        float4 lo = *reinterpret_cast<const float4*>(patch + px * SP + c);
        float4 hi = *reinterpret_cast<const float4*>(patch + px * SP + c + 4);
        const float4 c0 = *reinterpret_cast<const float4*>(b2 + c), c1 = *reinterpret_cast<const float4*>(b2 + c + 4);
        const bf16x8 rx = *reinterpret_cast<const bf16x8*>(xw + (p + 2 * W + 2) * LDC + c);
        // (bias first, then the residual: the two-launch path's order of the fp32 additions)
        lo.x += c0.x; lo.y += c0.y; lo.z += c0.z; lo.w += c0.w;
        hi.x += c1.x; hi.y += c1.y; hi.z += c1.z; hi.w += c1.w;
        lo.x += (float)rx[0]; lo.y += (float)rx[1]; lo.z += (float)rx[2]; lo.w += (float)rx[3];
        hi.x += (float)rx[4]; hi.y += (float)rx[5]; hi.z += (float)rx[6]; hi.w += (float)rx[7];
        lo.x = fmaxf(lo.x, 0.f); lo.y = fmaxf(lo.y, 0.f); lo.z = fmaxf(lo.z, 0.f); lo.w = fmaxf(lo.w, 0.f);
        hi.x = fmaxf(hi.x, 0.f); hi.y = fmaxf(hi.y, 0.f); hi.z = fmaxf(hi.z, 0.f); hi.w = fmaxf(hi.w, 0.f);
        act_st4(y + m * C + c, lo);
        act_st4(y + m * C + c + 4, hi);
      }
    }
    if (i == 0) __syncthreads();  // the patch is rewritten by the second row tile
  }
}

inline size_t pair_lds(int c, int W) {
  const size_t win = (size_t)((TILE + 4 * W + 4) + (TILE + 2 * W + 2)) * (c + 8) * sizeof(__bf16);
  return win;  // (the epilogue patches, 4 x 16 x (c + 4) floats, fit inside the intermediate: TILE + 2 W + 2 >= 130 pixels)
}

struct ConvPairGroup {
  int count, nblocks;
  struct Item {
    const __bf16* x; const __bf16* w1; const float* b1; const __bf16* w2; const float* b2; __bf16* y;
    int total, H, W, c, first_block;
  } e[4];
};

__global__ __launch_bounds__(NT) void conv3x3_pair_group_kernel(const ConvPairGroup g) {
  for (int v = blockIdx.x; v < g.nblocks; v += gridDim.x) {
    int k = 0;
    for (int i = 1; i < g.count; ++i)
      if (v >= g.e[i].first_block) k = i;
    const ConvPairGroup::Item& e = g.e[k];
    const int tile = v - e.first_block;
    if (e.c == 16) conv3x3_pair_body<16>(e.x, e.w1, e.b1, e.w2, e.b2, e.y, e.total, e.H, e.W, tile);
    else if (e.c == 32) conv3x3_pair_body<32>(e.x, e.w1, e.b1, e.w2, e.b2, e.y, e.total, e.H, e.W, tile);
    else if (e.c == 64) conv3x3_pair_body<64>(e.x, e.w1, e.b1, e.w2, e.b2, e.y, e.total, e.H, e.W, tile);
    else conv3x3_pair_body<128>(e.x, e.w1, e.b1, e.w2, e.b2, e.y, e.total, e.H, e.W, tile);
    if (v + (int)gridDim.x < g.nblocks) __syncthreads();
  }
}

// (forcing 5 / 6 / 8 waves per SIMD on this kernel spills: 36 / 50 / 56 us against 23 at its natural 132 registers)
// one map, 16 channels (the high-resolution branch: thousands of workgroups, a small register / LDS footprint of its own)
__global__ __launch_bounds__(NT) void conv3x3_pair16_kernel(const __bf16* __restrict__ x, const __bf16* __restrict__ w1,
                                                            const float* __restrict__ b1, const __bf16* __restrict__ w2,
                                                            const float* __restrict__ b2, __bf16* __restrict__ y, int total,
                                                            int H, int W) {
  const int ntiles = (total + TILE - 1) / TILE;
  for (int t = blockIdx.x; t < ntiles; t += gridDim.x) {
    conv3x3_pair_body<16>(x, w1, b1, w2, b2, y, total, H, W, t);
    if (t + (int)gridDim.x < ntiles) __syncthreads();
  }
}

// ---------------------------------------------------------------------------------------------------------------
// 3x3 / STRIDE 2 / pad 1 (round 4): the two stem convolutions of the trunk (hrnetv2.py:292-293,434-440: 4 -> 64 and
// 64 -> 64 channels, 3 % + 14 % of the trunk's FLOPs, 20 % of its time as K = 36-padded-to-64 implicit GEMMs with
// per-element index arithmetic).  Same raster-window idea: output raster index m = (n Ho + ho) Wo + wo has its
// stencil centre at INPUT raster index c(m) = 2 (m / Wo) W + 2 (m % Wo)  (H = 2 Ho, W = 2 Wo: image n's rows follow
// image n - 1's, so the formula holds across images), and the input rows that the 16 * RT * 4 consecutive outputs of a
// workgroup touch -- from the row above the first output's centre row to the row below the last one's -- are ONE
// contiguous span of memory, staged once with 16-B loads.  A fragments are ds_reads at c(m) - s0 + tap offset.
// CIN = 64: k-steps / packed weights exactly as the stride-1 kernel's (rf_conv3x3_pack_bf16).  CIN = 4 (the 3 + 1
// channels rf_stem_conv0 writes): a k-step is 8 taps x 4 channels, a lane's 8 k-values are two taps (two 8-B reads),
// two k-steps (tap 8 + seven zero taps in the second); weights packed by rf_conv3x3s2_pack_bf16.
// CIN >= 64: the window is staged one 32-channel slice at a time (S2Slice), the accumulators carry over -- as the
// stride-1 kernel's 256-channel case: the 64-channel window of the second stem convolution is 80 KB at W = 112 (one
// workgroup per CU: 395 us for 506 MB at C5), a 32-channel slice 45 KB.
template <int CIN> struct S2Slice { static constexpr int DEFAULT = 32; };

// output pixels a stride-2 workgroup of capacity `cap` (64 or 128) takes: whole output rows when a row fits
__host__ __device__ inline int s2_tile_px(int cap, int Wo) { return Wo <= cap ? (cap / Wo) * Wo : cap; }

template <int CIN, int COUT, int RT, int CH>
__device__ __forceinline__ void conv3x3s2_body(const __bf16* __restrict__ x, const __bf16* __restrict__ wt,
                                               const float* __restrict__ bias, const __bf16* __restrict__ residual,
                                               __bf16* __restrict__ y, int total_out,
                                               long total_in, int H, int W, int relu, int tile) {
  constexpr int TILE_ = 64 * RT;                     // output pixels per workgroup
  constexpr int NCH = CIN / CH;
  constexpr int LDC = CIN == 4 ? 4 : CH + 8;         // LDS pixel pitch (bf16)
  constexpr int KSTEPS = CIN == 4 ? 2 : (CIN == 16 ? 5 : 9 * (CH / 32));  // per slice
  constexpr int NTL = COUT / 16;
  __bf16* win = rf_conv_win;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int fr = lane & 15, fq = lane >> 4;
  const int Wo = W >> 1, Ho = H >> 1;
  // Output rows no wider than the tile: a workgroup takes WHOLE rows (tile_px = the largest multiple of Wo that fits) -- its
  // window is then 2 rows + 3 input rows instead of the up to two more an unaligned tile drags in (Wo = 28, 64 outputs:
  // 504 staged pixels unaligned, 280 for 56 aligned outputs), at the price of a few idle rows in the last MFMA tile.
  const int tile_px = s2_tile_px(TILE_, Wo);
  const long m0 = (long)tile * tile_px;
  const long mend = min(m0 + tile_px, (long)total_out);  // one past this workgroup's last output
  const long mlast = mend - 1;
  const long s0 = 2 * (m0 / Wo) * W - W;             // first pixel of the row above the first centre row (may be < 0)
  const int span = (int)(2 * (mlast / Wo) * W + 2 * W - s0);  // ... through the end of the row below the last centre row

  // ---- stage the window: a plain copy of `span` pixels x (a slice of) the channels (16-B loads, four in flight per thread) ----
  auto stage = [&](int ch) {
    constexpr int U = NCH > 1 ? CONV_U : 4;
    const int nvec = span * CH / 8;
    for (int base = 0; base < nvec; base += NT * U) {
      float4 raw[U];
      int at[U];
#pragma unroll
      for (int u = 0; u < U; ++u) {
        const int i = base + tid + u * NT;
        long g;  // first element of this 16-B group in the map
        if constexpr (CIN == 4) {
          g = s0 * CIN + (long)i * 8;                // (two pixels per group; s0 * 4 is a multiple of 8: W is even)
          at[u] = i < nvec ? i * 8 : -1;
        } else {
          const int px = i / (CH / 8), c = (i % (CH / 8)) * 8;
          g = (s0 + px) * CIN + ch * CH + c;
          at[u] = i < nvec ? px * LDC + c : -1;
        }
        raw[u] = make_float4(0.f, 0.f, 0.f, 0.f);
        if (i < nvec && g >= 0 && g + 8 <= total_in * CIN) raw[u] = *reinterpret_cast<const float4*>(x + g);
      }
#pragma unroll
      for (int u = 0; u < U; ++u)
        if (at[u] >= 0) *reinterpret_cast<float4*>(win + at[u]) = raw[u];
    }
  };
  stage(0);

  // ---- this lane's output pixels (one per MFMA row tile), their window-relative centres and border masks ----
  int pl[RT];
  unsigned vmask[RT];
#pragma unroll
  for (int i = 0; i < RT; ++i) {
    const long m = m0 + wave * (16 * RT) + i * 16 + fr;
    unsigned msk = 0;
    int c = W + 1;
    if (m < mend) {
      const int wo = (int)(m % Wo);
      const long r = m / Wo;
      const int ho = (int)(r % Ho);
      c = (int)(2 * r * W + 2 * wo - s0);
#pragma unroll
      for (int t = 0; t < 9; ++t) {
        const int hi = 2 * ho + t / 3 - 1, wi = 2 * wo + t % 3 - 1;
        if (hi >= 0 && hi < H && wi >= 0 && wi < W) msk |= 1u << t;
      }
    }
    pl[i] = c;
    vmask[i] = msk;
  }

  f32x4 acc[RT][NTL];
#pragma unroll
  for (int i = 0; i < RT; ++i)
#pragma unroll
    for (int j = 0; j < NTL; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
  auto ldb = [&](int s, int j) -> bf16x8 {
    return *reinterpret_cast<const bf16x8*>(wt + ((long)(s * NTL + j) * 64 + lane) * 8);
  };
  constexpr int PF = pf_depth(KSTEPS, NTL);
  auto gstep = [&](int s, int ch) -> int {  // k-step s of slice ch in the packed (tap-major) order
    if constexpr (NCH == 1) return s;
    else return (s / (CH / 32)) * (CIN / 32) + ch * (CH / 32) + s % (CH / 32);
  };
  bf16x8 bq[PF][NTL];
#pragma unroll 1
  for (int ch = 0; ch < NCH; ++ch) {
  if (ch > 0) {
    __syncthreads();  // every wave is done reading the previous slice
    stage(ch);
  }
#pragma unroll
  for (int d = 0; d < PF; ++d)
#pragma unroll
    for (int j = 0; j < NTL; ++j) bq[d][j] = ldb(gstep(d, ch), j);
  __syncthreads();  // window staged

#pragma unroll 1
  for (int sb = 0; sb < KSTEPS; sb += PF) {
#pragma unroll
    for (int d = 0; d < PF; ++d) {
      const int s = sb + d;
      if (s < KSTEPS) {
        bf16x8 a[RT];
        if constexpr (CIN == 4) {
          const int t0 = 8 * s + 2 * fq, t1 = t0 + 1;  // this lane's two taps (>= 9: zero)
          const int o0 = t0 < 9 ? (t0 / 3 - 1) * W + (t0 % 3 - 1) : 0, o1 = t1 < 9 ? (t1 / 3 - 1) * W + (t1 % 3 - 1) : 0;
#pragma unroll
          for (int i = 0; i < RT; ++i) {
            const bool k0 = t0 < 9 && ((vmask[i] >> t0) & 1u), k1 = t1 < 9 && ((vmask[i] >> t1) & 1u);
            bf16x4 lo = *reinterpret_cast<const bf16x4*>(win + (pl[i] + (k0 ? o0 : 0)) * LDC);
            bf16x4 hi = *reinterpret_cast<const bf16x4*>(win + (pl[i] + (k1 ? o1 : 0)) * LDC);
            if (!k0) lo = bf16x4{0, 0, 0, 0};
            if (!k1) hi = bf16x4{0, 0, 0, 0};
            a[i] = bf16x8{lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
          }
        } else {
          int tap, c0;  // (CIN = 16: a k-step is both 8-channel halves of two taps, the tenth tap is zero -- as in the stride-1 kernel)
          if constexpr (CIN == 16) { tap = 2 * s + (fq >> 1); c0 = (fq & 1) * 8; }
          else { tap = s / (CH / 32); c0 = (s % (CH / 32)) * 32 + fq * 8; }
          const int toff = tap < 9 ? (tap / 3 - 1) * W + (tap % 3 - 1) : 0;
#pragma unroll
          for (int i = 0; i < RT; ++i) {
            const bool ok = tap < 9 && ((vmask[i] >> tap) & 1u);
            bf16x8 v = *reinterpret_cast<const bf16x8*>(win + (pl[i] + (ok ? toff : 0)) * LDC + c0);
            if (!ok) v = bf16x8{0, 0, 0, 0, 0, 0, 0, 0};
            a[i] = v;
          }
        }
#pragma unroll
        for (int i = 0; i < RT; ++i)
#pragma unroll
          for (int j = 0; j < NTL; ++j)
            acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[i], bq[d][j], acc[i][j], 0, 0, 0);
        if (s + PF < KSTEPS) {
#pragma unroll
          for (int j = 0; j < NTL; ++j) bq[d][j] = ldb(gstep(s + PF, ch), j);
        }
      }
    }
  }
  }  // channel slices

  // ---- epilogue: as the stride-1 kernel's (fp32 patch per wave in the dead window, 8 channels per lane) ----
  constexpr int SP = COUT + 4;
  float* patch = reinterpret_cast<float*>(win) + wave * 16 * SP;
  __syncthreads();
#pragma unroll
  for (int i = 0; i < RT; ++i) {
#pragma unroll
    for (int j = 0; j < NTL; ++j)
#pragma unroll
      for (int r = 0; r < 4; ++r) patch[(fq * 4 + r) * SP + j * 16 + fr] = acc[i][j][r];
    __syncthreads();
    constexpr int VPP = COUT / 8;
    for (int v = lane; v < 16 * VPP; v += 64) {
      const int px = v / VPP, c = (v % VPP) * 8;
      const long m = m0 + wave * (16 * RT) + i * 16 + px;
      if (m < mend) {
        float4 lo = *reinterpret_cast<const float4*>(patch + px * SP + c);
        float4 hi = *reinterpret_cast<const float4*>(patch + px * SP + c + 4);
        const float4 b0 = *reinterpret_cast<const float4*>(bias + c), b1 = *reinterpret_cast<const float4*>(bias + c + 4);
        lo.x += b0.x; lo.y += b0.y; lo.z += b0.z; lo.w += b0.w;
        hi.x += b1.x; hi.y += b1.y; hi.z += b1.z; hi.w += b1.w;
        if (residual) {
          const float4 r0 = act_ld4(residual + m * COUT + c), r1 = act_ld4(residual + m * COUT + c + 4);
          lo.x += r0.x; lo.y += r0.y; lo.z += r0.z; lo.w += r0.w;
          hi.x += r1.x; hi.y += r1.y; hi.z += r1.z; hi.w += r1.w;
        }
        if (relu) {
          lo.x = fmaxf(lo.x, 0.f); lo.y = fmaxf(lo.y, 0.f); lo.z = fmaxf(lo.z, 0.f); lo.w = fmaxf(lo.w, 0.f);
          hi.x = fmaxf(hi.x, 0.f); hi.y = fmaxf(hi.y, 0.f); hi.z = fmaxf(hi.z, 0.f); hi.w = fmaxf(hi.w, 0.f);
        }
        act_st4(y + m * COUT + c, lo);
        act_st4(y + m * COUT + c + 4, hi);
      }
    }
    if (i + 1 < RT) __syncthreads();
  }
}

template <int CIN, int COUT, int RT, int CH>
__global__ __launch_bounds__(NT) void conv3x3s2_kernel(const __bf16* __restrict__ x, const __bf16* __restrict__ wt,
                                                        const float* __restrict__ bias, const __bf16* __restrict__ residual,
                                                        __bf16* __restrict__ y, int total_out, long total_in, int H, int W,
                                                        int relu) {
  const int tile_px = s2_tile_px(64 * RT, W >> 1), ntiles = (total_out + tile_px - 1) / tile_px;
  for (int t = blockIdx.x; t < ntiles; t += gridDim.x) {
    conv3x3s2_body<CIN, COUT, RT, CH>(x, wt, bias, residual, y, total_out, total_in, H, W, relu, t);
    if (t + (int)gridDim.x < ntiles) __syncthreads();
  }
}

// window pixels of a stride-2 workgroup: the row above + the centre rows + the row below of the output rows it touches
static int s2_span(int rt, int W) {
  const int Wo = W / 2, tile = s2_tile_px(64 * rt, Wo);
  const int rows = Wo <= 64 * rt ? tile / Wo : 2;  // (row-aligned, or a piece of two rows)
  return 2 * (rows - 1) * W + 3 * W;
}

// channels staged at a time: 32-channel slices once the whole 64-channel window would be large (one workgroup per CU); the
// small maps (a few hundred workgroups, latency-bound) keep the single pass -- slicing them cost 10 -> 15 us for 64 -> 128 at
// 14 x 14.  RF_CONV_S2_SLICE = 32 | 64 forces one (measurement switch).
static int s2_slice(int cin, int rt, int W) {
  static const int want = [] { const char* e = getenv("RF_CONV_S2_SLICE"); return e ? atoi(e) : 0; }();
  if (cin < 64) return cin;
  if (cin > 64) return want == 64 ? 64 : 32;
  if (want == 32 || want == 64) return want;
  return (size_t)s2_span(rt, W) * (64 + 8) * sizeof(__bf16) <= 40 * 1024 ? 64 : 32;
}

// LDS bytes of a stride-2 workgroup: the window (slice) or the epilogue patches
static size_t s2_lds(int cin, int cout, int rt, int W) {
  const int ldc = cin == 4 ? 4 : s2_slice(cin, rt, W) + 8;
  size_t lds = (size_t)s2_span(rt, W) * ldc * sizeof(__bf16) + 64;
  const size_t patches = (size_t)(NT / 64) * 16 * (cout + 4) * sizeof(float);
  return lds < patches ? patches : lds;
}

template <int CIN, int COUT, int RT, int CH>
int launch_s2_ch(const void* x, const void* wt, const float* bias, const void* residual, void* y, long total_out, long total_in,
                 int H, int W, int relu, hipStream_t st) {
  const size_t lds = s2_lds(CIN, COUT, RT, W);
  static bool attr = false;
  if (!attr) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(conv3x3s2_kernel<CIN, COUT, RT, CH>),
                              hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    attr = true;
  }
  const int tile_px = s2_tile_px(64 * RT, W / 2);
  const int blocks = trunk_grid((int)((total_out + tile_px - 1) / tile_px));
  RF_LAUNCH((conv3x3s2_kernel<CIN, COUT, RT, CH>), dim3(blocks), dim3(NT), lds, st, static_cast<const __bf16*>(x),
            static_cast<const __bf16*>(wt), bias, static_cast<const __bf16*>(residual), static_cast<__bf16*>(y), (int)total_out,
            total_in, H, W, relu);
  RF_CHECK_LAUNCH();
  return RF_OK;
}

template <int CIN, int COUT, int RT>
int launch_s2(const void* x, const void* wt, const float* bias, const void* residual, void* y, long total_out, long total_in,
              int H, int W, int relu, hipStream_t st) {
  if constexpr (CIN >= 64) {
    if (s2_slice(CIN, RT, W) == 64) return launch_s2_ch<CIN, COUT, RT, 64>(x, wt, bias, residual, y, total_out, total_in, H, W, relu, st);
    return launch_s2_ch<CIN, COUT, RT, 32>(x, wt, bias, residual, y, total_out, total_in, H, W, relu, st);
  } else {
    return launch_s2_ch<CIN, COUT, RT, CIN>(x, wt, bias, residual, y, total_out, total_in, H, W, relu, st);
  }
}

// CIN = 4 packing: out[((s*NTL + j)*64 + lane)*8 + e] = w[j*16 + (lane&15)][tap = 8 s + 2 (lane>>4) + (e>>2)][e & 3]
__global__ void pack_weights_c4_kernel(const float* __restrict__ w, __bf16* __restrict__ out, int cout) {
  const int ntl = cout / 16;
  const long total = (long)2 * ntl * 64 * 8;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    const int e = (int)(i & 7), lane = (int)((i >> 3) & 63);
    const int sj = (int)(i >> 9), j = sj % ntl, s = sj / ntl;
    const int fr = lane & 15, fq = lane >> 4;
    const int tap = 8 * s + 2 * fq + (e >> 2), c = e & 3;
    out[i] = (__bf16)(tap < 9 ? w[((long)(j * 16 + fr) * 9 + tap) * 4 + c] : 0.f);
  }
}

// channel slice staged at a time: the whole window, except for the 256-channel transition (see conv3x3_body)
template <int CIN> struct SliceOf { static constexpr int CH = CIN > 128 ? 64 : CIN; };

template <int CIN, int COUT, typename AT, int CH>
__global__ __launch_bounds__(NT) void conv3x3_kernel(const AT* __restrict__ x, const __bf16* __restrict__ wt,
                                                      const float* __restrict__ bias,
                                                      const AT* __restrict__ residual, AT* __restrict__ y,
                                                      int total, int H, int W, int relu) {
  // (one tile per workgroup unless the launch was capped: trunk_grid)
  const int ntiles = (total + TILE - 1) / TILE;
  for (int t = blockIdx.x; t < ntiles; t += gridDim.x) {
    conv3x3_body<CIN, COUT, AT, CH>(x, wt, bias, residual, y, total, H, W, relu, t);
    if (t + (int)gridDim.x < ntiles) __syncthreads();  // the epilogue patches alias the next tile's window
  }
}

// Grouped launch: the same convolution step of up to four INDEPENDENT maps (the branches of an HRNet module:
// 16 ch @ 28x28, 32 @ 14x14, 64 @ 7x7, 128 @ 4x4) in one launch.  The low-resolution branches are a few dozen
// workgroups walking a long k-loop (17 us for 42 workgroups at 4x4): launched one after the other they leave the
// chip idle four times per block, launched together they disappear underneath the 28x28 branch.  Entries are
// ordered by the host so that the longest-running workgroups (most channels) are dispatched first.
struct ConvGroup {
  int count, nblocks;  // nblocks: tiles of all entries (the launch may be capped below that: trunk_grid)
  struct Item {
    const void* x; const void* w; const float* bias; const void* res; void* y;
    int total, H, W, relu, cin, first_block;
  } e[4];
};

template <typename AT>
__global__ __launch_bounds__(NT) void conv3x3_group_kernel(const ConvGroup g) {
  for (int v = blockIdx.x; v < g.nblocks; v += gridDim.x) {
    int k = 0;
    for (int i = 1; i < g.count; ++i)
      if (v >= g.e[i].first_block) k = i;
    const ConvGroup::Item& e = g.e[k];
    const int tile = v - e.first_block;
    const AT* x = static_cast<const AT*>(e.x);
    const AT* res = static_cast<const AT*>(e.res);
    AT* y = static_cast<AT*>(e.y);
    const __bf16* w = static_cast<const __bf16*>(e.w);
    if (e.cin == 16) conv3x3_body<16, 16, AT>(x, w, e.bias, res, y, e.total, e.H, e.W, e.relu, tile);
    else if (e.cin == 32) conv3x3_body<32, 32, AT>(x, w, e.bias, res, y, e.total, e.H, e.W, e.relu, tile);
    else if (e.cin == 64) conv3x3_body<64, 64, AT>(x, w, e.bias, res, y, e.total, e.H, e.W, e.relu, tile);
    else conv3x3_body<128, 128, AT>(x, w, e.bias, res, y, e.total, e.H, e.W, e.relu, tile);
    if (v + (int)gridDim.x < g.nblocks) __syncthreads();
  }
}

template <int CIN, int COUT, typename AT, int CH>
int launch_ch(const void* x, const void* wt, const float* bias, const void* residual, void* y, long total, int H,
              int W, int relu, hipStream_t st) {
  size_t lds = (size_t)(TILE + 2 * W + 2) * (CH + 8) * sizeof(__bf16);
  const size_t patches = (size_t)(NT / 64) * 16 * (COUT + 4) * sizeof(float);  // epilogue staging, one per wave
  if (lds < patches) lds = patches;
  if (lds > 160 * 1024) { rf_g_last_error = "conv3x3 window exceeds LDS"; return RF_EUNSUPPORTED; }
  static bool attr = false;
  if (!attr) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(conv3x3_kernel<CIN, COUT, AT, CH>),
                              hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    attr = true;
  }
  const int blocks = trunk_grid((int)((total + TILE - 1) / TILE));
  RF_LAUNCH((conv3x3_kernel<CIN, COUT, AT, CH>), dim3(blocks), dim3(NT), lds, st, static_cast<const AT*>(x),
                     static_cast<const __bf16*>(wt), bias, static_cast<const AT*>(residual), static_cast<AT*>(y),
                     (int)total, H, W, relu);
  RF_CHECK_LAUNCH();
  return RF_OK;
}

template <int CIN, int COUT, typename AT>
int launch_t(const void* x, const void* wt, const float* bias, const void* residual, void* y, long total, int H,
             int W, int relu, hipStream_t st) {
  if constexpr (CIN > 128) {  // channel slices (conv3x3_body); RF_CONV_SLICE: measurement switch
    static const int ch = [] { const char* e = getenv("RF_CONV_SLICE"); return e ? atoi(e) : SliceOf<CIN>::CH; }();
    if (ch == 32) return launch_ch<CIN, COUT, AT, 32>(x, wt, bias, residual, y, total, H, W, relu, st);
    if (ch == 128) return launch_ch<CIN, COUT, AT, 128>(x, wt, bias, residual, y, total, H, W, relu, st);
    if (ch == CIN) return launch_ch<CIN, COUT, AT, CIN>(x, wt, bias, residual, y, total, H, W, relu, st);
    return launch_ch<CIN, COUT, AT, 64>(x, wt, bias, residual, y, total, H, W, relu, st);
  } else {
    return launch_ch<CIN, COUT, AT, CIN>(x, wt, bias, residual, y, total, H, W, relu, st);
  }
}

template <int CIN, int COUT>
int launch(const void* x, const void* wt, const float* bias, const void* residual, void* y, int act_dtype, long total,
           int H, int W, int relu, hipStream_t st) {
  if (act_dtype == RF_ACT_BF16) return launch_t<CIN, COUT, __bf16>(x, wt, bias, residual, y, total, H, W, relu, st);
  return launch_t<CIN, COUT, float>(x, wt, bias, residual, y, total, H, W, relu, st);
}

// fragment-order packing: out[((s*NTL + j)*64 + lane)*8 + e] = w[j*16 + (lane&15)][tap(s, lane>>4)][c0(s, lane>>4) + e]
__global__ void pack_weights_kernel(const float* __restrict__ w, __bf16* __restrict__ out, int cin, int cout) {
  const int ntl = cout / 16, ksteps = cin == 16 ? 5 : 9 * (cin / 32);
  const long total = (long)ksteps * ntl * 64 * 8;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    const int e = (int)(i & 7), lane = (int)((i >> 3) & 63);
    const int sj = (int)(i >> 9), j = sj % ntl, s = sj / ntl;
    const int fr = lane & 15, fq = lane >> 4;
    int tap, c0;
    if (cin == 16) { tap = 2 * s + (fq >> 1); c0 = (fq & 1) * 8; }
    else { tap = s / (cin / 32); c0 = (s % (cin / 32)) * 32 + fq * 8; }
    const float v = tap < 9 ? w[((long)(j * 16 + fr) * 9 + tap) * cin + c0 + e] : 0.f;
    out[i] = (__bf16)v;
  }
}

}  // namespace

extern "C" int64_t rf_conv3x3_packed_elems(int cin, int cout) {
  if (!rf_conv3x3_bf16_supported(cin, cout)) return 0;
  return (int64_t)(cin == 16 ? 5 : 9 * (cin / 32)) * (cout / 16) * 64 * 8;
}

extern "C" int rf_conv3x3_pack_bf16(const float* w, void* w_packed, int cin, int cout, void* stream) {
  RF_REQUIRE(w && w_packed && rf_conv3x3_bf16_supported(cin, cout));
  const long total = rf_conv3x3_packed_elems(cin, cout);
  RF_LAUNCH(pack_weights_kernel, dim3((int)((total + 255) / 256)), dim3(256), 0, static_cast<hipStream_t>(stream),
                     w, static_cast<__bf16*>(w_packed), cin, cout);
  RF_CHECK_LAUNCH();
  return RF_OK;
}

extern "C" int rf_conv3x3_group_bf16(const RfConvEntry* entries, int count, int act_dtype, void* stream) {
  RF_REQUIRE(entries && count >= 1 && count <= 4 && (act_dtype == RF_ACT_F32 || act_dtype == RF_ACT_BF16));
  ConvGroup g{};
  g.count = count;
  // longest-running workgroups first: more channels = longer k-loop
  int order[4] = {0, 1, 2, 3};
  for (int i = 1; i < count; ++i)
    for (int j = i; j > 0 && entries[order[j]].cin > entries[order[j - 1]].cin; --j) { const int v = order[j]; order[j] = order[j - 1]; order[j - 1] = v; }
  int blocks = 0;
  size_t lds = 0;
  for (int k = 0; k < count; ++k) {
    const RfConvEntry& e = entries[order[k]];
    RF_REQUIRE(e.x && e.w_packed && e.bias && e.y && e.N > 0 && e.H > 0 && e.W > 0);
    RF_REQUIRE(e.cin == e.cout && (e.cin == 16 || e.cin == 32 || e.cin == 64 || e.cin == 128));
    const long total = (long)e.N * e.H * e.W;
    RF_REQUIRE(total < (1L << 31));
    g.e[k].x = e.x; g.e[k].w = e.w_packed; g.e[k].bias = e.bias; g.e[k].res = e.residual; g.e[k].y = e.y;
    g.e[k].total = (int)total; g.e[k].H = e.H; g.e[k].W = e.W; g.e[k].relu = e.relu; g.e[k].cin = e.cin;
    g.e[k].first_block = blocks;
    blocks += (int)((total + TILE - 1) / TILE);
    size_t need = (size_t)(TILE + 2 * e.W + 2) * (e.cin + 8) * sizeof(__bf16);
    const size_t patches = (size_t)(NT / 64) * 16 * (e.cout + 4) * sizeof(float);
    if (need < patches) need = patches;
    if (lds < need) lds = need;
  }
  if (lds > 160 * 1024) { rf_g_last_error = "conv3x3 window exceeds LDS"; return RF_EUNSUPPORTED; }
  static bool attr = false;
  if (!attr) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(conv3x3_group_kernel<float>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(conv3x3_group_kernel<__bf16>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    attr = true;
  }
  hipStream_t st = static_cast<hipStream_t>(stream);
  g.nblocks = blocks;
  blocks = trunk_grid(blocks);
  if (act_dtype == RF_ACT_BF16) RF_LAUNCH(conv3x3_group_kernel<__bf16>, dim3(blocks), dim3(NT), lds, st, g);
  else RF_LAUNCH(conv3x3_group_kernel<float>, dim3(blocks), dim3(NT), lds, st, g);
  RF_CHECK_LAUNCH();
  return RF_OK;
}

extern "C" int rf_conv3x3_pair_supported(int c, int W) {
  return (c == 16 || c == 32 || c == 64 || c == 128) && W >= 1 && pair_lds(c, W) <= 160 * 1024;
}

extern "C" int rf_conv3x3_pair_group_bf16(const RfConvPairEntry* entries, int count, void* stream) {
  RF_REQUIRE(entries && count >= 1 && count <= 4);
  ConvPairGroup g{};
  g.count = count;
  int order[4] = {0, 1, 2, 3};  // longest-running workgroups first: more channels = longer k-loops
  for (int i = 1; i < count; ++i)
    for (int j = i; j > 0 && entries[order[j]].c > entries[order[j - 1]].c; --j) { const int v = order[j]; order[j] = order[j - 1]; order[j - 1] = v; }
  int blocks = 0;
  size_t lds = 0;
  for (int k = 0; k < count; ++k) {
    const RfConvPairEntry& e = entries[order[k]];
    RF_REQUIRE(e.x && e.w1_packed && e.bias1 && e.w2_packed && e.bias2 && e.y && e.N > 0 && e.H > 0 && e.W > 0);
    RF_REQUIRE(rf_conv3x3_pair_supported(e.c, e.W));
    RF_REQUIRE(((reinterpret_cast<uintptr_t>(e.x) | reinterpret_cast<uintptr_t>(e.y)) & 15) == 0 && e.x != e.y);
    const long total = (long)e.N * e.H * e.W;
    RF_REQUIRE(total < (1L << 31));
    ConvPairGroup::Item& it = g.e[k];
    it.x = static_cast<const __bf16*>(e.x); it.w1 = static_cast<const __bf16*>(e.w1_packed); it.b1 = e.bias1;
    it.w2 = static_cast<const __bf16*>(e.w2_packed); it.b2 = e.bias2; it.y = static_cast<__bf16*>(e.y);
    it.total = (int)total; it.H = e.H; it.W = e.W; it.c = e.c; it.first_block = blocks;
    blocks += (int)((total + TILE - 1) / TILE);
    const size_t need = pair_lds(e.c, e.W);
    if (lds < need) lds = need;
  }
  g.nblocks = blocks;
  hipStream_t st = static_cast<hipStream_t>(stream);
  static bool attr = false;
  if (!attr) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(conv3x3_pair_group_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(conv3x3_pair16_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    attr = true;
  }
  if (count == 1 && g.e[0].c == 16) {
    const ConvPairGroup::Item& it = g.e[0];
    RF_LAUNCH(conv3x3_pair16_kernel, dim3(trunk_grid(blocks)), dim3(NT), lds, st, it.x, it.w1, it.b1, it.w2, it.b2, it.y, it.total,
              it.H, it.W);
  } else {
    RF_LAUNCH(conv3x3_pair_group_kernel, dim3(trunk_grid(blocks)), dim3(NT), lds, st, g);
  }
  RF_CHECK_LAUNCH();
  return RF_OK;
}

static bool s2_combo(int cin, int cout) {
  if (cin == 4) return cout == 64;
  if (cin == 256) return cout == 32;  // (the 256 -> 32 transition)
  return (cin == 16 || cin == 32 || cin == 64) && (cout == 16 || cout == 32 || cout == 64 || cout == 128) && cout >= cin;
}

extern "C" int rf_conv3x3s2_bf16_supported(int cin, int cout, int W) {
  if (!s2_combo(cin, cout) || W < 4 || (W & 1)) return 0;
  return s2_lds(cin, cout, 2, W) <= 160 * 1024 || s2_lds(cin, cout, 1, W) <= 160 * 1024;
}

extern "C" int64_t rf_conv3x3s2_packed_elems(int cin, int cout) {
  if (!s2_combo(cin, cout)) return 0;
  return (int64_t)(cin == 4 ? 2 : (cin == 16 ? 5 : 9 * (cin / 32))) * (cout / 16) * 64 * 8;
}

extern "C" int rf_conv3x3s2_pack_bf16(const float* w, void* w_packed, int cin, int cout, void* stream) {
  RF_REQUIRE(w && w_packed && rf_conv3x3s2_packed_elems(cin, cout) > 0);
  const long total = rf_conv3x3s2_packed_elems(cin, cout);
  hipStream_t st = static_cast<hipStream_t>(stream);
  if (cin == 4)
    RF_LAUNCH(pack_weights_c4_kernel, dim3((int)((total + 255) / 256)), dim3(256), 0, st, w, static_cast<__bf16*>(w_packed), cout);
  else  // the stride-1 kernel's fragment order (k-steps depend on cin only)
    RF_LAUNCH(pack_weights_kernel, dim3((int)((total + 255) / 256)), dim3(256), 0, st, w, static_cast<__bf16*>(w_packed), cin, cout);
  RF_CHECK_LAUNCH();
  return RF_OK;
}

template <int CIN, int COUT>
static int launch_s2_rt(bool big, const void* x, const void* w, const float* bias, const void* res, void* y, long to, long ti, int H,
                        int W, int relu, hipStream_t st) {
  return big ? launch_s2<CIN, COUT, 2>(x, w, bias, res, y, to, ti, H, W, relu, st)
             : launch_s2<CIN, COUT, 1>(x, w, bias, res, y, to, ti, H, W, relu, st);
}

extern "C" int rf_conv3x3s2_bf16(const void* x, const void* w_bf16, const float* bias, const void* residual, void* y, int N, int H,
                                 int W, int cin, int cout, int relu, void* stream) {
  RF_REQUIRE(x && w_bf16 && bias && y && N > 0 && H >= 2 && W >= 4 && !(H & 1) && !(W & 1));
  RF_REQUIRE(rf_conv3x3s2_bf16_supported(cin, cout, W));
  const long ti = (long)N * H * W, to = ti / 4;
  RF_REQUIRE(ti < (1L << 31));
  hipStream_t st = static_cast<hipStream_t>(stream);
  // 128-pixel tiles while two workgroups still fit a CU (or when the 64-pixel window does not fit at all)
  // ... and the launch still has a couple of workgroups per CU (the 14 x 14 / 7 x 7 maps are a few hundred 64-pixel tiles:
  // halving their number doubled the time of 64 -> 128 at 14 x 14)
  const bool big = (s2_lds(cin, cout, 2, W) <= 80 * 1024 && to / s2_tile_px(128, W / 2) >= 512) || s2_lds(cin, cout, 1, W) > 160 * 1024;
#define RF_S2_GO(CI, CO) if (cin == CI && cout == CO) return launch_s2_rt<CI, CO>(big, x, w_bf16, bias, residual, y, to, ti, H, W, relu, st)
  RF_S2_GO(4, 64); RF_S2_GO(16, 16); RF_S2_GO(16, 32); RF_S2_GO(16, 64); RF_S2_GO(16, 128); RF_S2_GO(32, 32); RF_S2_GO(32, 64);
  RF_S2_GO(32, 128); RF_S2_GO(64, 64); RF_S2_GO(64, 128); RF_S2_GO(256, 32);
#undef RF_S2_GO
  return RF_EUNSUPPORTED;
}

extern "C" int rf_conv3x3_bf16_supported(int cin, int cout) {
  return (cin == 16 && cout == 16) || (cin == 32 && cout == 32) || (cin == 64 && cout == 64) ||
         (cin == 128 && cout == 128) || (cin == 256 && cout == 16);
}

extern "C" int rf_conv3x3_bf16(const void* x, const void* w_bf16, const float* bias, const void* residual,
                               void* y, int act_dtype, int N, int H, int W, int cin, int cout, int relu,
                               void* stream) {
  RF_REQUIRE(x && w_bf16 && bias && y && N > 0 && H > 0 && W > 0);
  RF_REQUIRE(act_dtype == RF_ACT_F32 || act_dtype == RF_ACT_BF16);
  RF_REQUIRE(rf_conv3x3_bf16_supported(cin, cout));
  const long total = (long)N * H * W;
  RF_REQUIRE(total < (1L << 31));
  hipStream_t st = static_cast<hipStream_t>(stream);
  if (cin == 16) return launch<16, 16>(x, w_bf16, bias, residual, y, act_dtype, total, H, W, relu, st);
  if (cin == 32) return launch<32, 32>(x, w_bf16, bias, residual, y, act_dtype, total, H, W, relu, st);
  if (cin == 64) return launch<64, 64>(x, w_bf16, bias, residual, y, act_dtype, total, H, W, relu, st);
  if (cin == 128) return launch<128, 128>(x, w_bf16, bias, residual, y, act_dtype, total, H, W, relu, st);
  return launch<256, 16>(x, w_bf16, bias, residual, y, act_dtype, total, H, W, relu, st);
}
