// Cross-resolution fusion of the HRNet trunk without intermediate maps ("fuse into the consuming add /
// concat", "fuse with K2 concat epilogue"):
//
//   rf_fuse_upsample_sum   every output branch of a HighResolutionModule's fuse layer that receives up-sampled terms, in
//                          ONE launch:  out_i = relu(base_i + base2_i + sum_j bilinear(t_ij))   (hrnetv2.py:250-271: the
//                          sum over j in ascending order, ReLU after the last term) -- instead of one read-modify-write
//                          pass over the full-resolution map per (i, j) pair;
//   rf_concat_pool_tokens  up-sample the four branch outputs to the first branch's resolution, concatenate along the
//                          channels (hrnetv2.py:453-498), AdaptiveAvgPool2d((8, 8)) (InverseForm.py:66-67) and emit the
//                          (N, 65, C) token layout with the constant -1 row (routeformer.py:478-487) -- the 240-channel
//                          full-resolution map is never written.
// Bilinear = F.interpolate(mode="bilinear", align_corners=False) (ATen area_pixel_compute_source_index, clamped at 0).
// HBM-bound streaming kernels: channels innermost, four channels per thread.
#include "common.h"

namespace {

struct Tap {
  int o00, o01, o10, o11;  // element offsets of the four source pixels (channel 0)
  float lh, lw;
};

__device__ __forceinline__ Tap bilinear_tap(int ho, int wo, int Hi, int Wi, float sh, float sw, int C) {
  float fh = ((float)ho + 0.5f) * sh - 0.5f; if (fh < 0.f) fh = 0.f;
  float fw = ((float)wo + 0.5f) * sw - 0.5f; if (fw < 0.f) fw = 0.f;
  const int h0 = (int)fh, w0 = (int)fw;
  const int h1 = h0 + (h0 < Hi - 1 ? 1 : 0), w1 = w0 + (w0 < Wi - 1 ? 1 : 0);
  Tap t;
  t.lh = fh - (float)h0; t.lw = fw - (float)w0;
  t.o00 = (h0 * Wi + w0) * C; t.o01 = (h0 * Wi + w1) * C; t.o10 = (h1 * Wi + w0) * C; t.o11 = (h1 * Wi + w1) * C;
  return t;
}

template <typename AT>
__device__ __forceinline__ float4 bilinear4(const AT* __restrict__ img, const Tap& t) {
  const float4 a00 = act_ld4(img + t.o00), a01 = act_ld4(img + t.o01), a10 = act_ld4(img + t.o10), a11 = act_ld4(img + t.o11);
  const float lh = t.lh, lw = t.lw;
  auto lerp = [&](float v00, float v01, float v10, float v11) {
    return (1.f - lh) * ((1.f - lw) * v00 + lw * v01) + lh * ((1.f - lw) * v10 + lw * v11);
  };
  return make_float4(lerp(a00.x, a01.x, a10.x, a11.x), lerp(a00.y, a01.y, a10.y, a11.y), lerp(a00.z, a01.z, a10.z, a11.z),
                     lerp(a00.w, a01.w, a10.w, a11.w));
}

struct FuseTable {
  int count;
  int first_block[RF_FUSE_MAX + 1];
  RfFuseEntry e[RF_FUSE_MAX];
};

template <typename AT>
__global__ __launch_bounds__(256) void fuse_upsample_sum_kernel(const FuseTable t) {
  int p = 0;
  while (p + 1 < t.count && (int)blockIdx.x >= t.first_block[p + 1]) ++p;
  const RfFuseEntry& e = t.e[p];
  const int C4 = e.C >> 2;
  const long total = (long)e.N * e.Ho * e.Wo * C4;
  const int nblk = t.first_block[p + 1] - t.first_block[p];
  const AT* base = static_cast<const AT*>(e.base);
  const AT* base2 = static_cast<const AT*>(e.base2);
  AT* out = static_cast<AT*>(e.out);
  for (long i = (long)((int)blockIdx.x - t.first_block[p]) * 256 + threadIdx.x; i < total; i += (long)nblk * 256) {
    const int c = (int)(i % C4) << 2;
    long r = i / C4;
    const int wo = (int)(r % e.Wo); r /= e.Wo;
    const int ho = (int)(r % e.Ho);
    const int n = (int)(r / e.Ho);
    float4 o = make_float4(0.f, 0.f, 0.f, 0.f);
    if (base) o = act_ld4(base + i * 4);
    if (base2) {
      const float4 b = act_ld4(base2 + i * 4);
      o.x += b.x; o.y += b.y; o.z += b.z; o.w += b.w;
    }
#pragma unroll
    for (int s = 0; s < 3; ++s) {
      if (s < e.n_src) {
        const int Hi = e.Hi[s], Wi = e.Wi[s];
        const Tap tp = bilinear_tap(ho, wo, Hi, Wi, (float)Hi / (float)e.Ho, (float)Wi / (float)e.Wo, e.C);
        const float4 v = bilinear4(static_cast<const AT*>(e.src[s]) + (long)n * Hi * Wi * e.C + c, tp);
        o.x += v.x; o.y += v.y; o.z += v.z; o.w += v.w;
      }
    }
    if (e.relu) { o.x = fmaxf(o.x, 0.f); o.y = fmaxf(o.y, 0.f); o.z = fmaxf(o.z, 0.f); o.w = fmaxf(o.w, 0.f); }
    act_st4(out + i * 4, o);
  }
}

struct PoolBranches {
  const void* x[4];
  int H[4], W[4], C[4], coff[4];  // maps (N, H, W, C) and the first channel of each branch in the concatenation
  int n;
};

template <typename AT>
__global__ __launch_bounds__(256) void concat_pool_tokens_kernel(const PoolBranches br, float* __restrict__ tok, int N,
                                                                 int Hf, int Wf, int Ctot) {
  const int C4 = Ctot >> 2;
  const long total = (long)N * 65 * C4;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    const int c = (int)(i % C4) << 2;
    const long r = i / C4;
    const int t = (int)(r % 65);
    const int n = (int)(r / 65);
    float* tp = tok + r * Ctot + c;
    if (t == 64) { *reinterpret_cast<float4*>(tp) = make_float4(-1.f, -1.f, -1.f, -1.f); continue; }
    int b = 0;
#pragma unroll
    for (int k = 1; k < 4; ++k) b += (k < br.n && c >= br.coff[k]) ? 1 : 0;
    const int Hi = br.H[b], Wi = br.W[b], Cb = br.C[b];
    const AT* img = static_cast<const AT*>(br.x[b]) + (long)n * Hi * Wi * Cb + (c - br.coff[b]);
    const int bh = t / 8, bw = t % 8;
    const int h0 = (bh * Hf) / 8, h1 = ((bh + 1) * Hf + 7) / 8;
    const int w0 = (bw * Wf) / 8, w1 = ((bw + 1) * Wf + 7) / 8;
    const float sh = (float)Hi / (float)Hf, sw = (float)Wi / (float)Wf;
    float4 s = make_float4(0.f, 0.f, 0.f, 0.f);
    for (int h = h0; h < h1; ++h)
      for (int w = w0; w < w1; ++w) {
        float4 v;
        if (Hi == Hf && Wi == Wf) v = act_ld4(img + ((long)h * Wi + w) * Cb);  // identity scale: the pixel itself
        else v = bilinear4(img, bilinear_tap(h, w, Hi, Wi, sh, sw, Cb));
        s.x += v.x; s.y += v.y; s.z += v.z; s.w += v.w;
      }
    const float inv = (float)((h1 - h0) * (w1 - w0));
    *reinterpret_cast<float4*>(tp) = make_float4(s.x / inv, s.y / inv, s.z / inv, s.w / inv);
  }
}

inline bool aligned_for(const void* p, int act_dtype) {
  return (reinterpret_cast<uintptr_t>(p) & (act_dtype == 1 ? 7 : 15)) == 0;
}

}  // namespace

extern "C" int rf_fuse_upsample_sum(const RfFuseEntry* entries, int count, int act_dtype, void* stream) {
  RF_REQUIRE(entries && count > 0 && count <= RF_FUSE_MAX && (act_dtype == 0 || act_dtype == 1));
  FuseTable t{};
  t.count = count;
  int blocks = 0;
  for (int p = 0; p < count; ++p) {
    const RfFuseEntry& e = entries[p];
    RF_REQUIRE(e.out && e.N > 0 && e.Ho > 0 && e.Wo > 0 && e.C > 0 && e.C % 4 == 0 && e.n_src >= 0 && e.n_src <= 3);
    RF_REQUIRE((e.base || e.base2 || e.n_src > 0) && aligned_for(e.out, act_dtype) && aligned_for(e.base, act_dtype) &&
               aligned_for(e.base2, act_dtype));
    RF_REQUIRE((long)e.Ho * e.Wo * e.C < (1L << 30));
    for (int s = 0; s < e.n_src; ++s)
      RF_REQUIRE(e.src[s] && e.Hi[s] > 0 && e.Wi[s] > 0 && aligned_for(e.src[s], act_dtype) &&
                 (long)e.Hi[s] * e.Wi[s] * e.C < (1L << 30));
    t.e[p] = e;
    t.first_block[p] = blocks;
    const long total = (long)e.N * e.Ho * e.Wo * (e.C / 4);
    long g = (total + 255) / 256;
    blocks += trunk_grid((int)(g > 4096 ? 4096 : (g < 1 ? 1 : g)));  // (grid-stride inside a part)
  }
  t.first_block[count] = blocks;
  hipStream_t st = static_cast<hipStream_t>(stream);
  if (act_dtype == 1) RF_LAUNCH(fuse_upsample_sum_kernel<__bf16>, dim3(blocks), dim3(256), 0, st, t);
  else RF_LAUNCH(fuse_upsample_sum_kernel<float>, dim3(blocks), dim3(256), 0, st, t);
  RF_CHECK_LAUNCH();
  return RF_OK;
}

extern "C" int rf_concat_pool_tokens(const void* const* maps, const int32_t* H, const int32_t* W, const int32_t* C, int n_maps,
                                     int act_dtype, float* tokens, int N, void* stream) {
  RF_REQUIRE(maps && H && W && C && tokens && n_maps >= 1 && n_maps <= 4 && N > 0 && (act_dtype == 0 || act_dtype == 1));
  PoolBranches br{};
  br.n = n_maps;
  int off = 0;
  for (int b = 0; b < n_maps; ++b) {
    RF_REQUIRE(maps[b] && H[b] > 0 && W[b] > 0 && C[b] > 0 && C[b] % 4 == 0 && aligned_for(maps[b], act_dtype));
    RF_REQUIRE((long)H[b] * W[b] * C[b] < (1L << 30));
    br.x[b] = maps[b]; br.H[b] = H[b]; br.W[b] = W[b]; br.C[b] = C[b]; br.coff[b] = off;
    off += C[b];
  }
  RF_REQUIRE((reinterpret_cast<uintptr_t>(tokens) & 15) == 0);
  const long total = (long)N * 65 * (off / 4);
  long g = (total + 255) / 256;
  const int grid = trunk_grid((int)(g > 8192 ? 8192 : g));
  hipStream_t st = static_cast<hipStream_t>(stream);
  if (act_dtype == 1)
    RF_LAUNCH(concat_pool_tokens_kernel<__bf16>, dim3(grid), dim3(256), 0, st, br, tokens, N, H[0], W[0], off);
  else
    RF_LAUNCH(concat_pool_tokens_kernel<float>, dim3(grid), dim3(256), 0, st, br, tokens, N, H[0], W[0], off);
  RF_CHECK_LAUNCH();
  return RF_OK;
}
