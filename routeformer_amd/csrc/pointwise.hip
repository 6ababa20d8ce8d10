// 1x1 convolution over bf16 NHWC maps = a streaming GEMM  y[M, COUT] = act(x[M, CIN] W^T + bias (+ residual))
// with a tiny reduction depth (CIN = 64 or 256), for the Bottleneck stage of the frozen HRNet-16 trunk
// (inverse_form_layers/hrnetv2.py:79-99 -- 263k pixels x 256 channels per map at the bench batch).
//
// The tiled implicit-GEMM kernel spends such a launch in prologues and epilogues: with K = 64 a workgroup's whole
// main loop is ONE k-step between two memory round trips.  Here the weights of the layer (<= 32 MFMA B fragments,
// fragment order, rf_pointwise_pack_bf16) live in REGISTERS for the lifetime of a wave, and every wave streams
// 16-pixel row tiles through them: A fragments come straight from global memory (prefetched one tile ahead), the
// residual of the tile is requested before its MFMAs are issued, and the result leaves through a wave-private
// fp32 LDS patch eight channels (16 B) per lane.  Bound: HBM, (CIN + COUT (+ COUT)) * 2 bytes per pixel.
#include "common.h"

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

namespace {

constexpr int NT = 256;

template <int CIN, int COUT>
__global__ __launch_bounds__(NT) void pointwise_kernel(const __bf16* __restrict__ x, const __bf16* __restrict__ wt,
                                                        const float* __restrict__ bias,
                                                        const __bf16* __restrict__ residual, __bf16* __restrict__ y,
                                                        int M, int relu) {
  constexpr int KS = CIN / 32, NTL = COUT / 16, SP = COUT + 4, VPP = COUT / 8;
  constexpr int VPL = 16 * VPP / 64;  // 8-channel vectors per lane in the epilogue
  static_assert(KS * NTL <= 32, "the layer's weights must fit the register budget");
  static_assert((16 * VPP) % 64 == 0, "epilogue vectors must split evenly over the wave");
  extern __shared__ __attribute__((aligned(16))) float patches[];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int fr = lane & 15, fq = lane >> 4;
  float* patch = patches + wave * 16 * SP;

  bf16x8 wfrag[KS][NTL];
#pragma unroll
  for (int s = 0; s < KS; ++s)
#pragma unroll
    for (int j = 0; j < NTL; ++j) wfrag[s][j] = *reinterpret_cast<const bf16x8*>(wt + ((long)(s * NTL + j) * 64 + lane) * 8);

  const int ntiles = (M + 15) >> 4, groups = (ntiles + 3) >> 2;
  auto load_a = [&](int tile, bf16x8 (&a)[KS]) {
    const int m = min(tile * 16 + fr, M - 1);  // clamped: rows past the end are computed and dropped
#pragma unroll
    for (int s = 0; s < KS; ++s) a[s] = *reinterpret_cast<const bf16x8*>(x + (long)m * CIN + s * 32 + fq * 8);
  };
  bf16x8 a_next[KS];
  int g = blockIdx.x;
  if (g < groups) load_a(min(g * 4 + wave, ntiles - 1), a_next);
  for (; g < groups; g += gridDim.x) {  // workgroup-uniform trip count (the barriers below)
    const int tile = g * 4 + wave;
    const bool live = tile < ntiles;
    bf16x8 a[KS];
#pragma unroll
    for (int s = 0; s < KS; ++s) a[s] = a_next[s];
    const int gn = g + gridDim.x;
    if (gn < groups) load_a(min(gn * 4 + wave, ntiles - 1), a_next);
    // the residual of this tile, in the epilogue's (pixel, 8-channel vector) mapping: in flight under the MFMAs
    uint4 rres[VPL];
    if (residual) {
#pragma unroll
      for (int q = 0; q < VPL; ++q) {
        const int v = lane + q * 64, px = v / VPP, c = (v % VPP) * 8;
        const int m = min(tile * 16 + px, M - 1);
        rres[q] = *reinterpret_cast<const uint4*>(residual + (long)(live ? m : 0) * COUT + c);
      }
    }
    f32x4 acc[NTL];
#pragma unroll
    for (int j = 0; j < NTL; ++j) acc[j] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int s = 0; s < KS; ++s)
#pragma unroll
      for (int j = 0; j < NTL; ++j) acc[j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[s], wfrag[s][j], acc[j], 0, 0, 0);
#pragma unroll
    for (int j = 0; j < NTL; ++j)
#pragma unroll
      for (int r = 0; r < 4; ++r) patch[(fq * 4 + r) * SP + j * 16 + fr] = acc[j][r];
    __syncthreads();
#pragma unroll
    for (int q = 0; q < VPL; ++q) {
      const int v = lane + q * 64, px = v / VPP, c = (v % VPP) * 8;
      const int m = tile * 16 + px;
      float4 lo = *reinterpret_cast<const float4*>(patch + px * SP + c);
      float4 hi = *reinterpret_cast<const float4*>(patch + px * SP + c + 4);
      const float4 b0 = *reinterpret_cast<const float4*>(bias + c), b1 = *reinterpret_cast<const float4*>(bias + c + 4);
      lo.x += b0.x; lo.y += b0.y; lo.z += b0.z; lo.w += b0.w;
      hi.x += b1.x; hi.y += b1.y; hi.z += b1.z; hi.w += b1.w;
      if (residual) {
        const uint4 rw = rres[q];  // eight bf16: a 16-bit shift each
        lo.x += __uint_as_float(rw.x << 16); lo.y += __uint_as_float(rw.x & 0xffff0000u);
        lo.z += __uint_as_float(rw.y << 16); lo.w += __uint_as_float(rw.y & 0xffff0000u);
        hi.x += __uint_as_float(rw.z << 16); hi.y += __uint_as_float(rw.z & 0xffff0000u);
        hi.z += __uint_as_float(rw.w << 16); hi.w += __uint_as_float(rw.w & 0xffff0000u);
      }
      if (relu) {
        lo.x = fmaxf(lo.x, 0.f); lo.y = fmaxf(lo.y, 0.f); lo.z = fmaxf(lo.z, 0.f); lo.w = fmaxf(lo.w, 0.f);
        hi.x = fmaxf(hi.x, 0.f); hi.y = fmaxf(hi.y, 0.f); hi.z = fmaxf(hi.z, 0.f); hi.w = fmaxf(hi.w, 0.f);
      }
      if (live && m < M) {
        const bf16x8 o = {(__bf16)lo.x, (__bf16)lo.y, (__bf16)lo.z, (__bf16)lo.w,
                          (__bf16)hi.x, (__bf16)hi.y, (__bf16)hi.z, (__bf16)hi.w};
        *reinterpret_cast<bf16x8*>(y + (long)m * COUT + c) = o;
      }
    }
    __syncthreads();  // the patch is rewritten by the next tile
  }
}

// out[((s*NTL + j)*64 + lane)*8 + e] = w[j*16 + (lane&15)][s*32 + (lane>>4)*8 + e]
__global__ void pointwise_pack_kernel(const float* __restrict__ w, __bf16* __restrict__ out, int cin, int cout) {
  const int ntl = cout / 16;
  const long total = (long)(cin / 32) * ntl * 64 * 8;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    const int e = (int)(i & 7), lane = (int)((i >> 3) & 63);
    const int sj = (int)(i >> 9), j = sj % ntl, s = sj / ntl;
    out[i] = (__bf16)w[(long)(j * 16 + (lane & 15)) * cin + s * 32 + (lane >> 4) * 8 + e];
  }
}

template <int CIN, int COUT>
int launch(const void* x, const void* wt, const float* bias, const void* residual, void* y, int M, int relu,
           hipStream_t st) {
  const size_t lds = (size_t)(NT / 64) * 16 * (COUT + 4) * sizeof(float);
  static bool attr = false;
  if (!attr) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(pointwise_kernel<CIN, COUT>),
                              hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    attr = true;
  }
  const int groups = ((M + 15) / 16 + 3) / 4;
  const int per_cu = (int)(160 * 1024 / lds) < 4 ? (int)(160 * 1024 / lds) : 4;  // co-resident workgroups per CU
  const int grid = trunk_grid(groups < 256 * per_cu ? groups : 256 * per_cu);  // persistent: every wave streams tiles
  RF_LAUNCH((pointwise_kernel<CIN, COUT>), dim3(grid), dim3(NT), lds, st, static_cast<const __bf16*>(x),
                     static_cast<const __bf16*>(wt), bias, static_cast<const __bf16*>(residual),
                     static_cast<__bf16*>(y), M, relu);
  RF_CHECK_LAUNCH();
  return RF_OK;
}

inline bool al16p(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; }

}  // namespace

extern "C" int rf_pointwise_bf16_supported(int cin, int cout) {
  return (cin == 64 && cout == 64) || (cin == 64 && cout == 256) || (cin == 256 && cout == 64);
}

extern "C" int64_t rf_pointwise_packed_elems(int cin, int cout) {
  return rf_pointwise_bf16_supported(cin, cout) ? (int64_t)(cin / 32) * (cout / 16) * 64 * 8 : 0;
}

extern "C" int rf_pointwise_pack_bf16(const float* w, void* w_packed, int cin, int cout, void* stream) {
  RF_REQUIRE(w && w_packed && rf_pointwise_bf16_supported(cin, cout));
  const long total = rf_pointwise_packed_elems(cin, cout);
  RF_LAUNCH(pointwise_pack_kernel, dim3((int)((total + 255) / 256)), dim3(256), 0,
                     static_cast<hipStream_t>(stream), w, static_cast<__bf16*>(w_packed), cin, cout);
  RF_CHECK_LAUNCH();
  return RF_OK;
}

extern "C" int rf_pointwise_bf16(const void* x, const void* w_packed, const float* bias, const void* residual, void* y,
                                 int64_t M, int cin, int cout, int relu, void* stream) {
  RF_REQUIRE(x && w_packed && bias && y && M > 0 && M < (1L << 31) - 64);
  RF_REQUIRE(rf_pointwise_bf16_supported(cin, cout));
  RF_REQUIRE(al16p(x) && al16p(w_packed) && al16p(bias) && al16p(residual) && al16p(y));
  hipStream_t st = static_cast<hipStream_t>(stream);
  if (cin == 64 && cout == 64) return launch<64, 64>(x, w_packed, bias, residual, y, (int)M, relu, st);
  if (cin == 64) return launch<64, 256>(x, w_packed, bias, residual, y, (int)M, relu, st);
  return launch<256, 64>(x, w_packed, bias, residual, y, (int)M, relu, st);
}
