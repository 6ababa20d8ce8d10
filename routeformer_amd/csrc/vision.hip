// Streaming NHWC helpers around the implicit-GEMM convolutions of the frozen HRNet-16 trunk:
// stem (frame gather + fp16 -> fp32 + 2x2/s2 conv), bilinear upsample(+add), add+ReLU, and the
// AdaptiveAvgPool2d((8,8)) -> token layout.  Channels are innermost so consecutive lanes touch
// consecutive addresses; each kernel is a grid-stride loop (HBM-bound, no reuse to stage).
#include <hip/hip_fp16.h>

#include "common.h"

namespace {

inline int grid_for(long total, int block = 256, int cap = 8192) {
  long g = (total + block - 1) / block;
  return (int)(g > cap ? cap : (g < 1 ? 1 : g));
}

__device__ __forceinline__ float ldf(const __half* p) { return __half2float(*p); }
__device__ __forceinline__ float ldf(const float* p) { return *p; }
// raw camera bytes: the dataset's `frame.astype(np.float16) / 255.0` (io/dataset.py:1506-1523) happens here --
// quotient rounded to fp16 exactly as numpy's half division does (computed in fp32, stored as fp16)
__device__ __forceinline__ float ldf(const uint8_t* p) { return __half2float(__float2half((float)*p / 255.0f)); }

template <typename TIn, typename AT>
__global__ void stem_conv0_kernel(const TIn* __restrict__ video, const int32_t* __restrict__ fidx,
                                  const float* __restrict__ w, AT* __restrict__ y, int B, int T, int F, int H,
                                  int W) {
  __shared__ float ws[36];
  if (threadIdx.x < 36) ws[threadIdx.x] = w[threadIdx.x];  // [co][ci][kh][kw]
  __syncthreads();
  const int Ho = H / 2, Wo = W / 2;
  const long total = (long)B * F * Ho * Wo;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    const int wo = (int)(i % Wo);
    long r = i / Wo;
    const int ho = (int)(r % Ho); r /= Ho;
    const int f = (int)(r % F);
    const int b = (int)(r / F);
    int fi = fidx[f];
    fi = fi < 0 ? 0 : (fi >= T ? T - 1 : fi);  // never read outside the clip, whatever the host passed
    const TIn* img = video + (((long)b * T + fi) * 3) * H * W;
    float acc[3] = {0.f, 0.f, 0.f};
#pragma unroll
    for (int ci = 0; ci < 3; ++ci)
#pragma unroll
      for (int kh = 0; kh < 2; ++kh) {
        const TIn* row = img + ((long)ci * H + 2 * ho + kh) * W + 2 * wo;
        const float x0 = ldf(row), x1 = ldf(row + 1);
#pragma unroll
        for (int co = 0; co < 3; ++co)
          acc[co] = fmaf(x1, ws[(co * 3 + ci) * 4 + kh * 2 + 1], fmaf(x0, ws[(co * 3 + ci) * 4 + kh * 2], acc[co]));
      }
    act_st4(y + i * 4, make_float4(acc[0], acc[1], acc[2], 0.f));
  }
}

// Four channels per thread (C % 4 == 0 everywhere in the trunk): 16-B / 8-B accesses instead of scalar ones.
template <typename AT>
__global__ void upsample_kernel(const AT* __restrict__ x, const AT* __restrict__ addend, AT* __restrict__ y,
                                int N, int Hi, int Wi, int C, int Ho, int Wo, long ldy, int accumulate, int relu) {
  const int C4 = C >> 2;
  const long total = (long)N * Ho * Wo * C4;
  const float sh = (float)Hi / (float)Ho, sw = (float)Wi / (float)Wo;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    const int c = (int)(i % C4) << 2;
    long r = i / C4;
    const int wo = (int)(r % Wo); r /= Wo;
    const int ho = (int)(r % Ho);
    const int n = (int)(r / Ho);
    // align_corners=False source index, clamped at 0 (ATen area_pixel_compute_source_index)
    float fh = ((float)ho + 0.5f) * sh - 0.5f; if (fh < 0.f) fh = 0.f;
    float fw = ((float)wo + 0.5f) * sw - 0.5f; if (fw < 0.f) fw = 0.f;
    const int h0 = (int)fh, w0 = (int)fw;
    const int h1 = h0 + (h0 < Hi - 1 ? 1 : 0), w1 = w0 + (w0 < Wi - 1 ? 1 : 0);
    const float lh = fh - (float)h0, lw = fw - (float)w0;
    const AT* xb = x + (long)n * Hi * Wi * C + c;
    const float4 a00 = act_ld4(xb + ((long)h0 * Wi + w0) * C), a01 = act_ld4(xb + ((long)h0 * Wi + w1) * C);
    const float4 a10 = act_ld4(xb + ((long)h1 * Wi + w0) * C), a11 = act_ld4(xb + ((long)h1 * Wi + w1) * C);
    AT* yp = y + (((long)n * Ho + ho) * Wo + wo) * ldy + c;
    float4 o = accumulate ? act_ld4(yp) : make_float4(0.f, 0.f, 0.f, 0.f);
    if (addend) {
      const float4 ad = act_ld4(addend + i * 4);
      o.x += ad.x; o.y += ad.y; o.z += ad.z; o.w += ad.w;
    }
    auto lerp = [&](float v00, float v01, float v10, float v11) {
      return (1.f - lh) * ((1.f - lw) * v00 + lw * v01) + lh * ((1.f - lw) * v10 + lw * v11);
    };
    o.x += lerp(a00.x, a01.x, a10.x, a11.x); o.y += lerp(a00.y, a01.y, a10.y, a11.y);
    o.z += lerp(a00.z, a01.z, a10.z, a11.z); o.w += lerp(a00.w, a01.w, a10.w, a11.w);
    if (relu) { o.x = fmaxf(o.x, 0.f); o.y = fmaxf(o.y, 0.f); o.z = fmaxf(o.z, 0.f); o.w = fmaxf(o.w, 0.f); }
    act_st4(yp, o);
  }
}

template <typename AT>
__global__ void add_relu_kernel(const AT* __restrict__ a, const AT* __restrict__ b, AT* __restrict__ out, long n,
                                int relu) {
  const long n4 = n >> 2;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (long)gridDim.x * blockDim.x) {
    const float4 u = act_ld4(a + 4 * i), w = act_ld4(b + 4 * i);
    float4 v = make_float4(u.x + w.x, u.y + w.y, u.z + w.z, u.w + w.w);
    if (relu) { v.x = fmaxf(v.x, 0.f); v.y = fmaxf(v.y, 0.f); v.z = fmaxf(v.z, 0.f); v.w = fmaxf(v.w, 0.f); }
    act_st4(out + 4 * i, v);
  }
  for (long i = 4 * n4 + (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
    float v = act_ld(a + i) + act_ld(b + i);
    if (relu) v = v > 0.f ? v : 0.f;
    act_st(out + i, v);
  }
}

template <typename AT>
__global__ void avgpool8_tokens_kernel(const AT* __restrict__ x, float* __restrict__ tok, int N, int H, int W,
                                       int C) {
  const int C4 = C >> 2;
  const long total = (long)N * 65 * C4;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    const int c = (int)(i % C4) << 2;
    const long r = i / C4;
    const int t = (int)(r % 65);
    const int n = (int)(r / 65);
    float* tp = tok + r * C + c;
    if (t == 64) { *reinterpret_cast<float4*>(tp) = make_float4(-1.f, -1.f, -1.f, -1.f); continue; }
    const int bh = t / 8, bw = t % 8;
    const int h0 = (bh * H) / 8, h1 = ((bh + 1) * H + 7) / 8;
    const int w0 = (bw * W) / 8, w1 = ((bw + 1) * W + 7) / 8;
    float4 s = make_float4(0.f, 0.f, 0.f, 0.f);
    for (int h = h0; h < h1; ++h)
      for (int w = w0; w < w1; ++w) {
        const float4 v = act_ld4(x + (((long)n * H + h) * W + w) * C + c);
        s.x += v.x; s.y += v.y; s.z += v.z; s.w += v.w;
      }
    const float inv = (float)((h1 - h0) * (w1 - w0));
    *reinterpret_cast<float4*>(tp) = make_float4(s.x / inv, s.y / inv, s.z / inv, s.w / inv);
  }
}

// ---- optimizer ----
__global__ __launch_bounds__(256) void sumsq_kernel(const float* __restrict__ g, long n, float* __restrict__ out) {
  __shared__ float red[4];
  float s = 0.f;
  // 16-B loads, four independent partial sums per thread (the gradient buffer is 256-B aligned); scalar tail
  const long n4 = ((reinterpret_cast<uintptr_t>(g) & 15) == 0) ? n >> 2 : 0;
  float4 a = make_float4(0.f, 0.f, 0.f, 0.f);
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (long)gridDim.x * blockDim.x) {
    const float4 v = reinterpret_cast<const float4*>(g)[i];
    a.x = fmaf(v.x, v.x, a.x); a.y = fmaf(v.y, v.y, a.y); a.z = fmaf(v.z, v.z, a.z); a.w = fmaf(v.w, v.w, a.w);
  }
  s = (a.x + a.y) + (a.z + a.w);
  for (long i = 4 * n4 + (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) s += g[i] * g[i];
  s = wave_sum(s);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
  __syncthreads();
  // one partial per workgroup, summed in fixed order by the consumer: the global norm (hence the clip factor,
  // hence every parameter) is bit-identical on every rank and every run -- an atomicAdd here let data-parallel
  // replicas drift apart by an ulp per step
  if (threadIdx.x == 0) out[blockIdx.x] = red[0] + red[1] + red[2] + red[3];
}

// HYPER: the scalars come from a device array (rf_adamw_clip_dev) -- a launch replayed from a HIP graph cannot
// take new by-value arguments, and the bias corrections / learning rate change every step.
template <bool HYPER>
__global__ __launch_bounds__(256) void adamw_clip_kernel(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ m,
                                  float* __restrict__ v, long n, const float* __restrict__ sumsq, float max_norm,
                                  float lr, float b1, float b2, float eps, float wd, float bc1, float bc2_sqrt,
                                  float grad_scale, int sumsq_parts, const float* __restrict__ hyper) {
  if constexpr (HYPER) {
    if (hyper[0] == 0.f) return;  // no update pending (first replay after a capture / after a flush)
    max_norm = hyper[1]; lr = hyper[2]; b1 = hyper[3]; b2 = hyper[4]; eps = hyper[5]; wd = hyper[6];
    bc1 = hyper[7]; bc2_sqrt = hyper[8]; grad_scale = hyper[9];
  }
  float coef = grad_scale;
  if (max_norm > 0.f) {
    // the norm from the per-workgroup partials: every workgroup adds them in the SAME fixed order (thread t takes k = t,
    // t + 256, ...; DPP wave sums; four wave totals) -- bit-identical in every workgroup, on every rank and run.  (One
    // thread walking all 1024 partials cost ~50 us of dependent loads at the head of every workgroup: the 3.5-M
    // parameter slice of the encoders took 170 us.)
    __shared__ float red[4];
    float part = 0.f;
    for (int k = threadIdx.x; k < sumsq_parts; k += 256) part += sumsq[k];
    part = wave_sum(part);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = part;
    __syncthreads();
    const float ss = (red[0] + red[1]) + (red[2] + red[3]);
    const float total = grad_scale * sqrtf(ss);
    const float c = max_norm / (total + 1e-6f);
    if (c < 1.f) coef *= c;
  }
  const float step = lr / bc1;
  auto upd = [&](float& pi, float gi, float& mi, float& vi) {
    gi *= coef;
    pi *= (1.f - lr * wd);
    mi = b1 * mi + (1.f - b1) * gi;
    vi = b2 * vi + (1.f - b2) * gi * gi;
    pi -= step * mi / (sqrtf(vi) / bc2_sqrt + eps);
  };
  // 16-B accesses over the 256-B aligned flat buffers (7 streams: 4 read, 3 written); scalar tail
  const bool al = ((reinterpret_cast<uintptr_t>(p) | reinterpret_cast<uintptr_t>(g) | reinterpret_cast<uintptr_t>(m) |
                    reinterpret_cast<uintptr_t>(v)) & 15) == 0;
  const long n4 = al ? n >> 2 : 0;
  // RF_ADAMW_NT: the moments and gradients are touched once per step and are far larger than the caches --
  // nontemporal accesses keep them from evicting what the next forward is about to read
  typedef float f4 __attribute__((ext_vector_type(4)));
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (long)gridDim.x * blockDim.x) {
    float4 pp = reinterpret_cast<float4*>(p)[i];
    const f4 m_ = __builtin_nontemporal_load(reinterpret_cast<f4*>(m) + i), v_ = __builtin_nontemporal_load(reinterpret_cast<f4*>(v) + i);
    const f4 gg = __builtin_nontemporal_load(reinterpret_cast<const f4*>(g) + i);
    float4 mm = make_float4(m_[0], m_[1], m_[2], m_[3]), vv = make_float4(v_[0], v_[1], v_[2], v_[3]);
    upd(pp.x, gg[0], mm.x, vv.x); upd(pp.y, gg[1], mm.y, vv.y); upd(pp.z, gg[2], mm.z, vv.z); upd(pp.w, gg[3], mm.w, vv.w);
    __builtin_nontemporal_store(f4{mm.x, mm.y, mm.z, mm.w}, reinterpret_cast<f4*>(m) + i);
    __builtin_nontemporal_store(f4{vv.x, vv.y, vv.z, vv.w}, reinterpret_cast<f4*>(v) + i);
    reinterpret_cast<float4*>(p)[i] = pp;
  }
  for (long i = 4 * n4 + (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
    float pi = p[i], mi = m[i], vi = v[i];
    upd(pi, g[i], mi, vi);
    m[i] = mi; v[i] = vi; p[i] = pi;
  }
}

}  // namespace

namespace {
template <typename TIn>
void launch_stem(const void* video, const int32_t* frame_idx, const float* w, void* y, int act_dtype, int B, int T,
                 int F, int H, int W, hipStream_t st) {
  const long total = (long)B * F * (H / 2) * (W / 2);
  if (act_dtype == 1)
    RF_LAUNCH((stem_conv0_kernel<TIn, __bf16>), dim3(grid_for(total)), dim3(256), 0, st,
                       static_cast<const TIn*>(video), frame_idx, w, static_cast<__bf16*>(y), B, T, F, H, W);
  else
    RF_LAUNCH((stem_conv0_kernel<TIn, float>), dim3(grid_for(total)), dim3(256), 0, st,
                       static_cast<const TIn*>(video), frame_idx, w, static_cast<float*>(y), B, T, F, H, W);
}
inline bool al8(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 7) == 0; }
inline bool al16v(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; }
// 4 consecutive channels must be one aligned vector access in the given storage type
inline bool act_aligned(const void* p, int act_dtype) { return !p || (act_dtype == 1 ? al8(p) : al16v(p)); }
}  // namespace

extern "C" int rf_stem_conv0(const void* video, int video_dtype, const int32_t* frame_idx, const float* w, void* y,
                             int act_dtype, int B, int T, int F, int H, int W, void* stream) {
  RF_REQUIRE(video && frame_idx && w && y && B > 0 && T > 0 && F > 0 && H > 1 && W > 1);
  RF_REQUIRE(H % 2 == 0 && W % 2 == 0 && video_dtype >= 0 && video_dtype <= 2);
  RF_REQUIRE((act_dtype == 0 || act_dtype == 1) && act_aligned(y, act_dtype));
  hipStream_t st = static_cast<hipStream_t>(stream);
  if (video_dtype == 2) launch_stem<uint8_t>(video, frame_idx, w, y, act_dtype, B, T, F, H, W, st);
  else if (video_dtype == 1) launch_stem<float>(video, frame_idx, w, y, act_dtype, B, T, F, H, W, st);
  else launch_stem<__half>(video, frame_idx, w, y, act_dtype, B, T, F, H, W, st);
  RF_CHECK_LAUNCH();
  return RF_OK;
}

extern "C" int rf_upsample_bilinear_nhwc(const void* x, const void* addend, void* y, int act_dtype, int N, int Hi,
                                         int Wi, int C, int Ho, int Wo, int64_t ldy, int accumulate, int relu,
                                         void* stream) {
  RF_REQUIRE(x && y && N > 0 && Hi > 0 && Wi > 0 && C > 0 && Ho > 0 && Wo > 0 && ldy >= C);
  RF_REQUIRE(act_dtype == 0 || act_dtype == 1);
  RF_REQUIRE(C % 4 == 0 && ldy % 4 == 0 && act_aligned(x, act_dtype) && act_aligned(addend, act_dtype) &&
             act_aligned(y, act_dtype));
  const int grid = grid_for((long)N * Ho * Wo * (C / 4));
  hipStream_t st = static_cast<hipStream_t>(stream);
  if (act_dtype == 1)
    RF_LAUNCH(upsample_kernel<__bf16>, dim3(grid), dim3(256), 0, st, static_cast<const __bf16*>(x),
                       static_cast<const __bf16*>(addend), static_cast<__bf16*>(y), N, Hi, Wi, C, Ho, Wo, (long)ldy,
                       accumulate, relu);
  else
    RF_LAUNCH(upsample_kernel<float>, dim3(grid), dim3(256), 0, st, static_cast<const float*>(x),
                       static_cast<const float*>(addend), static_cast<float*>(y), N, Hi, Wi, C, Ho, Wo, (long)ldy,
                       accumulate, relu);
  RF_CHECK_LAUNCH();
  return RF_OK;
}

extern "C" int rf_add_relu(const void* a, const void* b, void* out, int act_dtype, int64_t n, int relu, void* stream) {
  RF_REQUIRE(a && b && out && n > 0 && (act_dtype == 0 || act_dtype == 1));
  RF_REQUIRE(act_aligned(a, act_dtype) && act_aligned(b, act_dtype) && act_aligned(out, act_dtype));
  hipStream_t st = static_cast<hipStream_t>(stream);
  const int grid = grid_for((n + 3) / 4);
  if (act_dtype == 1)
    RF_LAUNCH(add_relu_kernel<__bf16>, dim3(grid), dim3(256), 0, st, static_cast<const __bf16*>(a),
                       static_cast<const __bf16*>(b), static_cast<__bf16*>(out), (long)n, relu);
  else
    RF_LAUNCH(add_relu_kernel<float>, dim3(grid), dim3(256), 0, st, static_cast<const float*>(a),
                       static_cast<const float*>(b), static_cast<float*>(out), (long)n, relu);
  RF_CHECK_LAUNCH();
  return RF_OK;
}

extern "C" int rf_avgpool8_tokens(const void* x, int act_dtype, float* tokens, int N, int H, int W, int C,
                                  void* stream) {
  RF_REQUIRE(x && tokens && N > 0 && H > 0 && W > 0 && C > 0 && (act_dtype == 0 || act_dtype == 1));
  RF_REQUIRE(C % 4 == 0 && act_aligned(x, act_dtype) && al16v(tokens));
  hipStream_t st = static_cast<hipStream_t>(stream);
  const int grid = grid_for((long)N * 65 * (C / 4));
  if (act_dtype == 1)
    RF_LAUNCH(avgpool8_tokens_kernel<__bf16>, dim3(grid), dim3(256), 0, st, static_cast<const __bf16*>(x),
                       tokens, N, H, W, C);
  else
    RF_LAUNCH(avgpool8_tokens_kernel<float>, dim3(grid), dim3(256), 0, st, static_cast<const float*>(x),
                       tokens, N, H, W, C);
  RF_CHECK_LAUNCH();
  return RF_OK;
}

extern "C" int rf_sumsq_parts(int64_t n) { return grid_for(n, 256, 1024); }

extern "C" int rf_sumsq(const float* g, int64_t n, float* sumsq, void* stream) {
  RF_REQUIRE(g && sumsq && n > 0);
  RF_LAUNCH(sumsq_kernel, dim3(rf_sumsq_parts(n)), dim3(256), 0, static_cast<hipStream_t>(stream), g,
                     (long)n, sumsq);
  RF_CHECK_LAUNCH();
  return RF_OK;
}

extern "C" int rf_adamw_clip(float* p, const float* g, float* m, float* v, int64_t n, const float* sumsq,
                             int sumsq_parts, float max_norm, float lr, float beta1, float beta2, float eps, float wd,
                             int step, float grad_scale, void* stream) {
  RF_REQUIRE(p && g && m && v && n > 0 && step >= 1 && (max_norm <= 0.f || (sumsq && sumsq_parts >= 1)));
  const float bc1 = 1.f - powf(beta1, (float)step);
  const float bc2_sqrt = sqrtf(1.f - powf(beta2, (float)step));
  RF_LAUNCH(adamw_clip_kernel<false>, dim3(grid_for(n, 256, 4096)), dim3(256), 0,
                     static_cast<hipStream_t>(stream), p, g, m, v, (long)n, sumsq, max_norm, lr, beta1, beta2, eps, wd,
                     bc1, bc2_sqrt, grad_scale, sumsq_parts, static_cast<const float*>(nullptr));
  RF_CHECK_LAUNCH();
  return RF_OK;
}

extern "C" int rf_adamw_clip_dev(float* p, const float* g, float* m, float* v, int64_t n, const float* sumsq,
                                 int sumsq_parts, const float* hyper, int max_blocks, void* stream) {
  RF_REQUIRE(p && g && m && v && n > 0 && hyper && sumsq && sumsq_parts >= 1 && max_blocks >= 0);
  RF_LAUNCH(adamw_clip_kernel<true>, dim3(grid_for(n, 256, max_blocks > 0 ? max_blocks : 4096)), dim3(256), 0,
                     static_cast<hipStream_t>(stream), p, g, m, v, (long)n, sumsq, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 1.f, 1.f,
                     1.f, sumsq_parts, hyper);
  RF_CHECK_LAUNCH();
  return RF_OK;
}

thread_local const char* rf_g_last_error = "";
thread_local hipEvent_t rf_g_timer_start = nullptr, rf_g_timer_stop = nullptr;
thread_local bool rf_g_timer_armed = false;
namespace {
constexpr int TIMER_PAIRS = 4096;
thread_local hipEvent_t timer_ev[2 * TIMER_PAIRS];
thread_local int timer_created = 0, timer_used = 0;
}  // namespace

extern "C" int rf_kernel_timer_arm(void) {
  if (timer_used >= TIMER_PAIRS) { rf_g_last_error = "kernel timer: collect before arming more"; return RF_EINVAL; }
  if (timer_used >= timer_created) {
    if (hipEventCreate(&timer_ev[2 * timer_created]) != hipSuccess ||
        hipEventCreate(&timer_ev[2 * timer_created + 1]) != hipSuccess) {
      rf_g_last_error = "hipEventCreate failed";
      return RF_ELAUNCH;
    }
    ++timer_created;
  }
  rf_g_timer_start = timer_ev[2 * timer_used];
  rf_g_timer_stop = timer_ev[2 * timer_used + 1];
  ++timer_used;
  rf_g_timer_armed = true;
  return RF_OK;
}

extern "C" int rf_kernel_timer_collect(float* us, int capacity) {
  const int n = timer_used < capacity ? timer_used : capacity;
  if (rf_g_timer_armed) { rf_g_timer_armed = false; }  // armed but nothing was launched: that slot reads < 0
  for (int i = 0; i < n; ++i) {
    float ms = -1.f;
    if (us) {
      if (hipEventSynchronize(timer_ev[2 * i + 1]) != hipSuccess ||
          hipEventElapsedTime(&ms, timer_ev[2 * i], timer_ev[2 * i + 1]) != hipSuccess)
        ms = -1.f;
      us[i] = ms < 0.f ? -1.f : ms * 1000.f;
    }
  }
  (void)hipGetLastError();  // an unrecorded pair is an expected, reported (< 0) condition
  timer_used = 0;
  return n;
}

extern "C" int rf_version(void) { return 1; }
extern "C" const char* rf_last_error(void) { return rf_g_last_error; }
