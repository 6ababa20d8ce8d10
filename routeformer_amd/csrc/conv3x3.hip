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
// workgroup, so B fragments come straight from global memory (L1/L2 hits), software-prefetched one
// k-step ahead.  Epilogue: + bias (+ residual) -> ReLU -> NHWC in the maps' storage type.
//
// v_mfma_f32_16x16x32_bf16: a k-step covers 32 input channels of one tap, or (C_in = 16) both halves of
// two taps (the weight tensor then carries a zero 10th tap).
#include "common.h"

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

namespace {

constexpr int NT = 256;
constexpr int TILE = 128;  // output pixels per workgroup (4 waves x 2 MFMA row tiles)

template <int CIN, int COUT, typename AT>
__global__ __launch_bounds__(NT) void conv3x3_kernel(const AT* __restrict__ x, const __bf16* __restrict__ wt,
                                                      const float* __restrict__ bias,
                                                      const AT* __restrict__ residual, AT* __restrict__ y,
                                                      int total, int H, int W, int relu) {
  constexpr int LDC = CIN + 8;                       // LDS pixel pitch (bf16): odd multiple of 16 B
  constexpr int TAPS = (CIN == 16) ? 10 : 9;         // taps stored per output channel
  constexpr int KSTEPS = (CIN == 16) ? 5 : 9 * (CIN / 32);
  constexpr int NTL = COUT / 16;                     // MFMA column tiles
  extern __shared__ __attribute__((aligned(16))) __bf16 win[];

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int fr = lane & 15, fq = lane >> 4;
  const long m0 = (long)blockIdx.x * TILE;
  const int span = TILE + 2 * W + 2;                 // staged pixels
  const long s0 = m0 - W - 1;                        // raster index of window pixel 0

  // ---- stage the input window (contiguous in memory) ----
  {
    const long nvec = (long)span * (CIN / 4);
    for (long i = tid; i < nvec; i += NT) {
      const int px = (int)(i / (CIN / 4)), c = (int)(i % (CIN / 4)) * 4;
      const long g = s0 + px;
      if constexpr (sizeof(AT) == 2) {  // bf16 maps: the window is a plain copy
        uint2 raw = make_uint2(0u, 0u);
        if (g >= 0 && g < total) raw = *reinterpret_cast<const uint2*>(x + g * CIN + c);
        *reinterpret_cast<uint2*>(win + px * LDC + c) = raw;
      } else {
        float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
        if (g >= 0 && g < total) v = act_ld4(x + g * CIN + c);
        bf16x4 o = {(__bf16)v.x, (__bf16)v.y, (__bf16)v.z, (__bf16)v.w};
        *reinterpret_cast<bf16x4*>(win + px * LDC + c) = o;
      }
    }
  }

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

  // B fragment of k-step s, column tile j: wt[(j*16+fr)][tap][c0 .. c0+7]
  auto ldb = [&](int s, int j) -> bf16x8 {
    int tap, c0;
    if constexpr (CIN == 16) { tap = 2 * s + (fq >> 1); c0 = (fq & 1) * 8; }
    else { tap = s / (CIN / 32); c0 = (s % (CIN / 32)) * 32 + fq * 8; }
    return *reinterpret_cast<const bf16x8*>(wt + ((long)(j * 16 + fr) * TAPS + tap) * CIN + c0);
  };

  bf16x8 bcur[NTL], bnext[NTL];
#pragma unroll
  for (int j = 0; j < NTL; ++j) bcur[j] = ldb(0, j);
  __syncthreads();  // window staged

#pragma unroll 1
  for (int s = 0; s < KSTEPS; ++s) {
    if (s + 1 < KSTEPS) {
#pragma unroll
      for (int j = 0; j < NTL; ++j) bnext[j] = ldb(s + 1, j);
    }
    int tap, c0;
    if constexpr (CIN == 16) { tap = 2 * s + (fq >> 1); c0 = (fq & 1) * 8; }
    else { tap = s / (CIN / 32); c0 = (s % (CIN / 32)) * 32 + fq * 8; }
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
        acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[i], bcur[j], acc[i][j], 0, 0, 0);
#pragma unroll
    for (int j = 0; j < NTL; ++j) bcur[j] = bnext[j];
  }

  // ---- epilogue: C/D fragment col = lane&15 (channel), row = 4*(lane>>4)+r (pixel) ----
#pragma unroll
  for (int i = 0; i < 2; ++i) {
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const long m = m0 + wave * 32 + i * 16 + fq * 4 + r;
      if (m >= total) continue;
#pragma unroll
      for (int j = 0; j < NTL; ++j) {
        const int n = j * 16 + fr;
        float v = acc[i][j][r] + bias[n];
        if (residual) v += act_ld(residual + m * COUT + n);
        if (relu) v = v > 0.f ? v : 0.f;
        act_st(y + m * COUT + n, v);
      }
    }
  }
}

template <int CIN, int COUT, typename AT>
int launch_t(const void* x, const void* wt, const float* bias, const void* residual, void* y, long total, int H,
             int W, int relu, hipStream_t st) {
  const size_t lds = (size_t)(TILE + 2 * W + 2) * (CIN + 8) * sizeof(__bf16);
  if (lds > 160 * 1024) { rf_g_last_error = "conv3x3 window exceeds LDS"; return RF_EUNSUPPORTED; }
  static bool attr = false;
  if (!attr) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(conv3x3_kernel<CIN, COUT, AT>),
                              hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    attr = true;
  }
  const int blocks = (int)((total + TILE - 1) / TILE);
  hipLaunchKernelGGL((conv3x3_kernel<CIN, COUT, AT>), dim3(blocks), dim3(NT), lds, st, static_cast<const AT*>(x),
                     static_cast<const __bf16*>(wt), bias, static_cast<const AT*>(residual), static_cast<AT*>(y),
                     (int)total, H, W, relu);
  RF_CHECK_LAUNCH();
  return RF_OK;
}

template <int CIN, int COUT>
int launch(const void* x, const void* wt, const float* bias, const void* residual, void* y, int act_dtype, long total,
           int H, int W, int relu, hipStream_t st) {
  if (act_dtype == RF_ACT_BF16) return launch_t<CIN, COUT, __bf16>(x, wt, bias, residual, y, total, H, W, relu, st);
  return launch_t<CIN, COUT, float>(x, wt, bias, residual, y, total, H, W, relu, st);
}

}  // namespace

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
