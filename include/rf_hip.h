/*
 * rf_hip.h -- C ABI of librf_hip.so: the MI355X (gfx950) kernels under the Routeformer hot path.
 *
 * The reference (meakbiyik/routeformer) is pure Python and has no FFI of its own; every arithmetic
 * op on its hot path is an ATen call.  Each entry point below replaces the ATen work of the cited
 * reference lines (paths relative to /root/reference).  All entry points:
 *   - take raw DEVICE pointers, sizes and a hipStream_t (passed as void*); no torch types;
 *   - never allocate: workspaces are passed in; all launches are asynchronous on `stream`;
 *   - return 0 on success, a negative RF_E* code on bad arguments / launch failure (no exceptions).
 * Activations are fp32 in HBM.  `prec` selects the matrix-core input type of the dense
 * contractions: 0 = exact fp32 (v_mfma_f32_16x16x4_f32), 1 = bf16 inputs / fp32 accumulate
 * (v_mfma_f32_16x16x32_bf16, operands rounded to bf16 while staging into LDS).
 */
#ifndef RF_HIP_H
#define RF_HIP_H
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define RF_OK 0
#define RF_EINVAL (-1)
#define RF_ELAUNCH (-2)
#define RF_EUNSUPPORTED (-3)

#define RF_ACT_NONE 0
#define RF_ACT_RELU 1
#define RF_ACT_GELU 2 /* erf form, F.gelu default (cross_modal_transformer.py:221,286) */
#define RF_ACT_ELU 3

int rf_version(void);
/* Last HIP error string seen by the library on this thread (for diagnostics). */
const char* rf_last_error(void);

/* ---- dense contraction -------------------------------------------------------------------
 * C[M,N] = epi( A[M,K] * B[K,N] ),  A[m,k] = A[m*lda_m + k*lda_k],  B[k,n] = B[k*ldb_k + n*ldb_n].
 * epi(v) : v += bias[n];  if (preact) preact[m,n] = v;  v = act(v);
 *          if (dact_mode) v *= act'(dact_src[m,n]) (1: relu mask src>0, 2: gelu'(src), 3: elu'(src));
 *          if (residual) v += residual[(m % res_rows), n];   (res_before_act: add before act)
 * Replaces nn.Linear / Conv1d(k=1) forward+backward GEMMs: cross_modal_transformer.py:177-180,
 * 189-198,215-216,281-282,423,496; gps_backbone/layers/SelfAttentionFamily.py:176-192;
 * layers/TransformerEncoderDecoder.py:36-37,50-51,98-99; Informer.py:102.
 * splitk > 1 needs workspace of splitk*M*N floats (deterministic two-pass reduction), unless
 * atomic_accumulate = 1: then every K-slice adds its partial product into C with fp32 atomics
 * (C += A*B; C must hold the running sum, e.g. a zeroed slot of the flat gradient buffer; no epilogue
 * other than that is allowed; summation order, hence the last bits, vary run to run).
 * a_rowsum (optional, needs atomic_accumulate and a row-contiguous A, i.e. lda_m == 1):
 * a_rowsum[m] += sum_k A[m,k] -- the bias gradient rides along with the weight-gradient GEMM
 * (dW = dY^T X, db = dY^T 1) instead of a second pass over dY.
 * tile_counters (optional): >= 4096 zero-initialised uint32 owned by the caller and used by ONE stream at
 * a time; with it the split-K slabs are summed inside the launch by the last-arriving workgroup of each
 * output tile (agent-scope release/acquire hand-off; slices summed in ascending order => deterministic)
 * and the counters are left at zero again -- no second reduction launch. */
int rf_gemm(const float* A, int64_t lda_m, int64_t lda_k, const float* B, int64_t ldb_k,
            int64_t ldb_n, float* C, int64_t ldc, int M, int N, int K, const float* bias,
            const float* residual, int64_t ldr, int res_rows, int res_before_act, int act,
            float* preact, int64_t ldp, const float* dact_src, int64_t ldd, int dact_mode,
            int prec, int splitk, float* workspace, int atomic_accumulate, float* a_rowsum,
            uint32_t* tile_counters, void* stream);

/* Skinny variant of rf_gemm for the GPS backbone's linear layers at training batch sizes (M = B x L <= 640 rows against
 * 832..3328-wide weights; same call sites as rf_gemm): the launch streams the weight matrix once with every operand of
 * a workgroup requested up front (one HBM round trip), 8 waves interleave the k-steps of an output tile and meet in LDS,
 * the epilogue runs in the launch.  bf16-input MFMA, fp32 accumulation (rf_gemm's prec = 1 contract).  A must be k
 * contiguous (lda_k == 1), B either k contiguous (ldb_k == 1: y = x W^T) or n contiguous (ldb_n == 1: dX = dY W);
 * K % 8 == 0.  rf_gemm_skinny_split returns the number of K slices the kernel will use for a problem (0: not
 * supported -- call rf_gemm); with more than one slice `workspace` must hold slices * M * N floats. */
int rf_gemm_skinny_split(const float* A, int64_t lda_m, int64_t lda_k, const float* B, int64_t ldb_k, int64_t ldb_n,
                         int M, int N, int K);
int rf_gemm_skinny(const float* A, int64_t lda_m, int64_t lda_k, const float* B, int64_t ldb_k, int64_t ldb_n,
                   float* C, int64_t ldc, int M, int N, int K, const float* bias, const float* residual,
                   int64_t ldr, int res_rows, int res_before_act, int act, float* preact, int64_t ldp,
                   const float* dact_src, int64_t ldd, int dact_mode, float* workspace, void* stream);

/* The split-K slices' RAW products of rf_gemm / rf_gemm_skinny, left as slabs workspace[slice][M][N] (row pitch N): no
 * epilogue and no slab-sum launch -- the consumer sums them (rf_layernorm_fwd_slabs: the out-projection / Conv1d(k=1) pair
 * in front of every LayerNorm of the GPS backbone, layers/TransformerEncoderDecoder.py:44-53,104-118, is a product, a slab
 * sum and a norm; the slab sum rides in the norm's loads).  Same kernels, slice boundaries and per-slice arithmetic as
 * the full calls.  Slab counts: rf_gemm_split_count(K, splitk) for rf_gemm_partials (>= 1, the effective count after
 * rf_gemm's own clamping of `splitk`), rf_gemm_skinny_split(...) for rf_gemm_skinny_partials (0: unsupported). */
int rf_gemm_split_count(int K, int splitk);
int rf_gemm_partials(const float* A, int64_t lda_m, int64_t lda_k, const float* B, int64_t ldb_k, int64_t ldb_n,
                     int M, int N, int K, int prec, int splitk, float* workspace, void* stream);
int rf_gemm_skinny_partials(const float* A, int64_t lda_m, int64_t lda_k, const float* B, int64_t ldb_k, int64_t ldb_n,
                            int M, int N, int K, float* workspace, void* stream);

/* out[n] (+)= sum_m X[m*ldx + n] (bias gradients; accumulate=1 adds into out, e.g. a slot of the flat
 * gradient buffer; accumulate=2 does that with fp32 atomics in ONE launch -- the order of the additions is then not
 * fixed -- and needs no workspace).  workspace: parts*N floats, parts = rf_colsum_parts(M,N). */
int rf_colsum_parts(int M, int N);
int rf_colsum(const float* X, int64_t ldx, int M, int N, float* out, int accumulate, float* workspace,
              void* stream);

/* Activation storage of the trunk's NHWC maps (`act_dtype` below): RF_ACT_F32 = fp32, RF_ACT_BF16 = bf16.  The bf16
 * matrix-core mode rounds every convolution input to bf16 anyway; keeping the maps themselves in bf16 halves the
 * HBM traffic of this bandwidth-bound stack.  Arithmetic between load and store is fp32 in both forms; pointers
 * typed `void*` below are float* or bf16* accordingly, pitches are in elements. */
#define RF_ACT_F32 0
#define RF_ACT_BF16 1

/* ---- conv2d as implicit GEMM over NHWC (frozen HRNet-16 trunk, inference only) -------------
 * y[n,ho,wo,co] = act( sum_{kh,kw,ci} x[n,ho*s-p+kh,wo*s-p+kw,ci] * w[co,kh,kw,ci] + bias[co]
 *                      (+ residual[n,ho,wo,co]) )
 * BatchNorm2d(eval) is folded into w/bias by the host.  x has row pitch `cin` (channels innermost),
 * y is written with channel pitch ldy at channel offset 0 of the given pointer.
 * Replaces Conv2d+BN+ReLU(+add) of inverse_form_layers/hrnetv2.py:45-61,79-99,434-440. */
int rf_conv2d_nhwc(const void* x, const float* w, const float* bias, const void* residual,
                   void* y, int act_dtype, int N, int H, int W, int cin, int cout, int ksize, int stride, int pad,
                   int Ho, int Wo, int64_t ldy, int64_t ldres, int relu, int prec, void* stream);

/* 3x3 / stride 1 / pad 1 fast path on the bf16 matrix cores ("raster window": the contiguous NHWC span a
 * 128-pixel tile needs is staged into LDS once; no im2col).  Supported (cin,cout): (16,16) (32,32) (64,64)
 * (128,128) (256,16) -- the BasicBlock / Bottleneck / transition convs of hrnetv2.py:45-61,79-99,310-330.
 * Weights (BatchNorm folded) are consumed in MFMA fragment order: rf_conv3x3_pack_bf16 turns the fp32 device tensor
 * w[cout][3][3][cin] into that form once (rf_conv3x3_packed_elems bf16 elements).
 * Same arithmetic contract as rf_conv2d_nhwc(prec = 1). */
int rf_conv3x3_bf16_supported(int cin, int cout);
int64_t rf_conv3x3_packed_elems(int cin, int cout);
int rf_conv3x3_pack_bf16(const float* w, void* w_packed, int cin, int cout, void* stream);
/* The same convolution step of up to four independent maps in ONE launch (the branches of an HRNet module, cin ==
 * cout in {16,32,64,128}); entries as for rf_conv3x3_bf16, all maps in the same act_dtype. */
typedef struct RfConvEntry {
  const void* x; const void* w_packed; const float* bias; const void* residual; void* y;
  int N, H, W, cin, cout, relu;
} RfConvEntry;
int rf_conv3x3_group_bf16(const RfConvEntry* entries, int count, int act_dtype, void* stream);
int rf_conv3x3_bf16(const void* x, const void* w_packed, const float* bias, const void* residual, void* y,
                    int act_dtype, int N, int H, int W, int cin, int cout, int relu, void* stream);
/* BasicBlock pair on bf16 NHWC maps (round 4): y = relu(conv2(relu(conv1(x) + bias1)) + bias2 + x) of hrnetv2.py:45-61 in one
 * launch, the intermediate map in LDS; c = cin = cout in {16, 32, 64, 128}; weights in rf_conv3x3_pack_bf16 order (BatchNorm
 * folded); up to four independent maps (the branches of an HRNet module) per launch; y must not alias x.  Results are
 * bit-identical to two rf_conv3x3_bf16 launches with a bf16 intermediate map. */
typedef struct RfConvPairEntry {
  const void* x; const void* w1_packed; const float* bias1; const void* w2_packed; const float* bias2; void* y;
  int N, H, W, c;
} RfConvPairEntry;
int rf_conv3x3_pair_supported(int c, int W);
int rf_conv3x3_pair_group_bf16(const RfConvPairEntry* entries, int count, void* stream);
/* 3x3 / stride 2 / pad 1 on bf16 NHWC maps (round 4): the trunk's two stem convolutions (hrnetv2.py:292-293,434-440: cin 4 -- the
 * 3 + 1 channels rf_stem_conv0 writes -- and 64 -> 64) and the stride-2 chains of the cross-resolution fuse layers and transitions
 * (hrnetv2.py:148-200,337-370: cin in {16, 32, 64} -> cout in {16, 32, 64, 128}, cout >= cin); H, W even.  Weights: [cout][3][3][cin]
 * fp32 (BatchNorm folded) -> rf_conv3x3s2_pack_bf16 (rf_conv3x3s2_packed_elems bf16 elements).  y = relu?(conv + bias [+ residual]),
 * residual = a bf16 map of y's shape or NULL. */
int rf_conv3x3s2_bf16_supported(int cin, int cout, int W);
int64_t rf_conv3x3s2_packed_elems(int cin, int cout);
int rf_conv3x3s2_pack_bf16(const float* w, void* w_packed, int cin, int cout, void* stream);
int rf_conv3x3s2_bf16(const void* x, const void* w_bf16, const float* bias, const void* residual, void* y, int N, int H, int W,
                      int cin, int cout, int relu, void* stream);

/* 1x1 convolution over bf16 maps as a streaming GEMM with the layer's weights held in registers (the Bottleneck
 * stage, hrnetv2.py:79-99): y[M,cout] = relu?(x[M,cin] W^T + bias (+ residual[M,cout])), x / residual / y bf16,
 * dense rows.  Supported (cin,cout): (64,64) (64,256) (256,64).  Weights (BatchNorm folded) in fragment order:
 * rf_pointwise_pack_bf16 converts the fp32 device tensor w[cout][cin] once (rf_pointwise_packed_elems bf16
 * elements).  Same arithmetic contract as rf_conv2d_nhwc(prec = 1, RF_ACT_BF16) with ksize 1. */
int rf_pointwise_bf16_supported(int cin, int cout);
int64_t rf_pointwise_packed_elems(int cin, int cout);
int rf_pointwise_pack_bf16(const float* w, void* w_packed, int cin, int cout, void* stream);
int rf_pointwise_bf16(const void* x, const void* w_packed, const float* bias, const void* residual, void* y,
                      int64_t M, int cin, int cout, int relu, void* stream);

/* Stem: frame gather + cast + conv0 (3->3, k2 s2, no BN; hrnetv2.py:292-293,432-433).
 * video: (B,T,3,H,W), video_dtype 0 = fp16 in [0,1] (what the dataset emits), 1 = fp32, 2 = raw uint8 camera
 * bytes -- the dataset's `astype(float16) / 255` (io/dataset.py:1506-1523) is then applied on the fly, bit for
 * bit, so clips can stay uint8 in HBM; frame_idx[F] picks frames (routeformer.py:418-421);
 * y: (B*F, H/2, W/2, 4) NHWC (act_dtype) with a zero 4th channel. w: (3,3,2,2) as in the state dict. */
int rf_stem_conv0(const void* video, int video_dtype, const int32_t* frame_idx, const float* w,
                  void* y, int act_dtype, int B, int T, int F, int H, int W, void* stream);

/* y[n,ho,wo,c] = (accumulate ? y : 0) + (addend ? addend[n,ho,wo,c] : 0) + bilinear(x)[n,ho,wo,c]
 * (align_corners=False), optional ReLU afterwards; y has channel pitch ldy, addend is dense (pitch C).
 * hrnetv2.py:266-271,453-498.  C % 4 == 0, ldy % 4 == 0 (four channels per access). */
int rf_upsample_bilinear_nhwc(const void* x, const void* addend, void* y, int act_dtype, int N, int Hi, int Wi,
                              int C, int Ho, int Wo, int64_t ldy, int accumulate, int relu,
                              void* stream);

/* out = relu?(a + b) elementwise over n elements (fuse-layer identity terms). */
int rf_add_relu(const void* a, const void* b, void* out, int act_dtype, int64_t n, int relu, void* stream);

/* AdaptiveAvgPool2d((8,8)) on NHWC + token layout + the constant -1 row:
 * x (N,H,W,C) (act_dtype) -> tokens (N,65,C) fp32; InverseForm.py:66-67 + routeformer.py:478-487.  C % 4 == 0. */
int rf_avgpool8_tokens(const void* x, int act_dtype, float* tokens, int N, int H, int W, int C, void* stream);

/* ---- sequence ops ---------------------------------------------------------------------------
 * Circular unfold for Conv1d(k=3, padding_mode="circular"): x (B,L,C) -> cols (B,Lout,3C),
 * Lout = L + 2*pad - 2, cols[b,l,c*3+t] = x[b,(l+t-pad) mod L,c] -- (c,t) order = the memory order of a
 * Conv1d weight (d,c,3), so the weight is used (and its gradient written) in place as a (d,3c) matrix;
 * fold is the adjoint (gradient).
 * cross_modal_transformer.py:352-369; layers/Embedding.py:28-46; TransformerEncoderDecoder.py:12-18. */
int rf_unfold3_circular(const float* x, float* cols, int B, int L, int C, int pad, void* stream);
int rf_fold3_circular(const float* dcols, float* dx, int B, int L, int C, int pad, void* stream);
/* The same with a row pitch `ld` >= 3 C for the unfolded matrix: unfold zero-fills columns 3 C .. ld - 1, fold ignores them.
 * A c_in whose 3 C is not a multiple of 4 (the GPS backbone's 69 input channels) is unfolded with ld = 208 so that the
 * token-embedding GEMM and its backward stay on the 16-B vector path (weights zero-padded to the same K). */
int rf_unfold3_circular_ld(const float* x, float* cols, int B, int L, int C, int pad, int ld, void* stream);
int rf_fold3_circular_ld(const float* dcols, float* dx, int B, int L, int C, int pad, int ld, void* stream);

/* LayerNorm over the last dim (eps 1e-5) of s = x (+ residual); saves xhat and rstd for backward.
 * cross_modal_transformer.py:283-284,297,301,421. cols <= 1024. */
int rf_layernorm_fwd(const float* x, const float* residual, const float* gamma, const float* beta,
                     float* y, float* xhat, float* rstd, int rows, int cols, float eps, void* stream);
/* The same with x a strided view: row r of x at x + (r / seg_rows) * seg_stride + (r % seg_rows) * row_stride (elements) --
 * the consumed tail of a (B, L, C) activation (seg_rows = tail length, seg_stride = L * C, row_stride = C) or a row-pitched
 * 2-D view (seg_rows = rows); residual and the outputs are contiguous (rows, cols). */
int rf_layernorm_fwd_strided(const float* x, int seg_rows, int64_t seg_stride, int64_t row_stride, const float* residual,
                             const float* gamma, const float* beta, float* y, float* xhat, float* rstd, int rows, int cols,
                             float eps, void* stream);
/* The same norm on s = (slabs[0] + slabs[1] + ... + slabs[splits-1]) (+ bias[col]) (+ residual), slabs =
 * [splits][rows][cols] from rf_gemm_partials / rf_gemm_skinny_partials: summed in slab order, then the bias, then the
 * residual -- bit-identical to rf_gemm (bias epilogue) followed by rf_layernorm_fwd(residual, product). */
int rf_layernorm_fwd_slabs(const float* slabs, int splits, const float* bias, const float* residual,
                           const float* gamma, const float* beta, float* y, float* xhat, float* rstd, int rows,
                           int cols, float eps, void* stream);
/* rf_layernorm_fwd_slabs whose output is written straight in the im2col layout of the distilling convolution that
 * consumes it (Conv1d k = 3, circular padding 2, layers/TransformerEncoderDecoder.py:12-18): rows = B * L sequences rows,
 * cols_out[b][r][3 c + t] = y[b][(r + t - 2) mod L][c] for r in [0, L + 2) -- what rf_unfold3_circular(pad = 2) would
 * produce from y; the unfold launch between the norm and the product is gone (round 4).  cols >= 257. */
int rf_layernorm_fwd_slabs_unfold(const float* slabs, int splits, const float* bias, const float* residual,
                                  const float* gamma, const float* beta, float* cols_out, float* xhat, float* rstd,
                                  int rows, int cols, int L, float eps, void* stream);
/* rf_layernorm_bwd whose incoming gradient is that of the im2col image above, dcols[b][L + 2][3 cols]: the fold
 * (rf_fold3_circular, pad = 2) happens on load. */
int rf_layernorm_bwd_fold(const float* dcols, const float* xhat, const float* rstd, const float* gamma, float* dx,
                          float* dgamma, float* dbeta, int accumulate, float* workspace, int rows, int cols, int L,
                          void* stream);
/* LayerNorm backward whose incoming gradient is still the `splits` split-K slabs [splits][rows][cols] of the product in front of
 * it (rf_gemm_partials / rf_gemm_skinny_partials of a dX) plus, optionally, the skip gradient `residual` [rows][cols] that
 * product's epilogue would have added: summed on load -- rf_layernorm_bwd minus the slab-sum launch. */
int rf_layernorm_bwd_slabs(const float* slabs, int splits, const float* residual, const float* xhat, const float* rstd,
                           const float* gamma, float* dx, float* dgamma, float* dbeta, int accumulate, float* workspace,
                           int rows, int cols, void* stream);
/* dx = d(loss)/d(s); dgamma/dbeta reduced deterministically through `workspace`
 * (rf_layernorm_bwd_parts(rows)*2*cols floats); accumulate=1 adds them into dgamma/dbeta. */
int rf_layernorm_bwd_parts(int rows);
int rf_layernorm_bwd(const float* dy, const float* xhat, const float* rstd, const float* gamma,
                     float* dx, float* dgamma, float* dbeta, int accumulate, float* workspace, int rows,
                     int cols, void* stream);
/* accumulate = 2: every workgroup adds its partial (dgamma, dbeta) into the outputs with fp32 atomics
 * (outputs hold the running sum, e.g. slots of the zeroed flat gradient buffer): one launch, no workspace,
 * summation order not reproducible to the last bit. */

/* Informer distilling layer tail: BatchNorm1d (train: batch stats, eval: running stats) -> ELU ->
 * MaxPool1d(3,2,1) on (B,L,C), C innermost.  layers/TransformerEncoderDecoder.py:19-28.
 * stats: mean[C], var[C] (biased) computed by rf_bn_stats in train mode; with running_mean != NULL the same
 * launch applies nn.BatchNorm1d's running-statistics update (momentum, unbiased variance) and bumps
 * num_batches_tracked (int64 scalar on the device, may be NULL). */
int rf_bn_stats(const float* x, float* mean, float* var, int rows, int C, float* running_mean,
                float* running_var, int64_t* num_batches_tracked, float momentum, void* stream);
int rf_bn_elu_pool_fwd(const float* x, const float* mean, const float* var, const float* gamma,
                       const float* beta, float* y, int32_t* argmax, int B, int L, int C, float eps,
                       void* stream);
/* dx for the whole BN(train)->ELU->pool chain; dgamma/dbeta too (accumulate=1: added to what is there).
 * training=0: stats are constants. */
int rf_bn_elu_pool_bwd(const float* dy, const int32_t* argmax, const float* x, const float* mean,
                       const float* var, const float* gamma, const float* beta, float* dx,
                       float* dgamma, float* dbeta, int accumulate, int B, int L, int C, float eps,
                       int training, void* stream);
/* The same backward with the incoming gradient still in `splits` split-K slabs [splits][B * Lout][C] of the dX product in front of
 * it (+ an optional skip gradient `residual`): summed on load.  rf_bn_elu_pool_bwd_slab_ok(B, L): the LDS-slab kernel applies. */
int rf_bn_elu_pool_bwd_slab_ok(int B, int L);
int rf_bn_elu_pool_bwd_slabs(const float* slabs, int splits, const float* residual, const int32_t* argmax, const float* x,
                             const float* mean, const float* var, const float* gamma, const float* beta, float* dx,
                             float* dgamma, float* dbeta, int accumulate, int B, int L, int C, float eps, int training,
                             void* stream);
/* Train-mode forward of the same tail in ONE launch (batch statistics + running-statistics update + BatchNorm -> ELU ->
 * MaxPool; mean / var = the biased batch statistics, kept for the backward pass): a workgroup holds a 32-channel slab of
 * all B * L rows in LDS.  RF_EUNSUPPORTED when the slab does not fit (then: rf_bn_stats + rf_bn_elu_pool_fwd). */
int rf_bn_train_elu_pool_fwd(const float* x, const float* gamma, const float* beta, float* mean, float* var,
                             float* running_mean, float* running_var, int64_t* num_batches_tracked, float momentum, float* y,
                             int32_t* argmax, int B, int L, int C, float eps, void* stream);
/* ... fed with the `splits` split-K slabs [splits][B * L][C] of the convolution's product (rf_gemm_partials) and its bias: summed on
 * load, the finished pre-normalisation map is written to x_out for the backward pass. */
int rf_bn_train_elu_pool_fwd_slabs(const float* slabs, int splits, const float* bias, float* x_out, const float* gamma,
                                   const float* beta, float* mean, float* var, float* running_mean, float* running_var,
                                   int64_t* num_batches_tracked, float momentum, float* y, int32_t* argmax, int B, int L, int C,
                                   float eps, void* stream);

/* Grouped weight gradients (the dW / db GEMMs of nn.Linear / Conv1d(k=1) backward, autograd's
 * `grad_weight = grad_out^T @ input`): for each entry  dw[N,K] += dy[M,N]^T x[M,K]  and, if db != NULL,
 * db[N] += column sums of dy -- fp32 atomics into gradient slots that were zeroed at the start of the step.
 * Up to RF_WGRAD_MAX_GROUP problems per launch; `entries` is a HOST array (copied into the kernel arguments).
 * dy / x: 16-B aligned, unit column stride, row pitches ld_dy / ld_x and N, K multiples of 4; dw contiguous (N,K).
 * `splits` = requested split of the reduction dimension M (clamped); `kchunk` is filled in by the library.
 * `exclusive` != 0: the caller guarantees that nothing else writes dw during this launch and that dw holds
 * zeros; if the problem also ends up with a single K slice the tile is then written with plain stores
 * instead of fp32 atomics (most of the bytes of a backward pass: the large weights have shallow reductions). */
#define RF_WGRAD_MAX_GROUP 48
typedef struct RfWgradEntry {
  const void* dy; const void* x; float* dw; float* db;
  int M, N, K, ld_dy, ld_x, splits, kchunk, exclusive;
  int dy_bf16, x_bf16; /* both != 0: the operands lie in memory as bf16 (rf_wgrad_tr only; ld in elements; both or neither).
                          The fused encoder stacks write their dy slabs / activation saves that way. */
} RfWgradEntry;
int rf_wgrad_grouped(const RfWgradEntry* entries, int count, int prec, void* stream);
/* The bf16 matrix-core path of rf_wgrad_grouped (prec = 1 routes here; RF_WGRAD_TR=0 in the environment keeps the tiled
 * kernel): operands copied row-major into LDS and read back with the hardware transpose read (ds_read_b64_tr_b16),
 * 256 x 128 blocks of dW per workgroup, a three-deep register ring of global loads, reduction chunks of >= 1 024 rows;
 * same entry semantics (plain stores for an exclusive entry that ends up with one chunk, fp32 atomics otherwise). */
int rf_wgrad_tr(const RfWgradEntry* entries, int count, void* stream);

/* ---- row-block kernels for the d_model = 128 stacks (bf16-input MFMA only) ----
 * A workgroup owns 64 complete rows and stages the whole weight matrix in LDS, so the residual add and the
 * LayerNorm of EncoderLayer / DecoderLayer (cross_modal_transformer.py:279-365) finish in the epilogue.
 * rf_rowblock_linear:  y[M,N] = x[M,K] w[N,K]^T + bias [+ residual[M,N]]   (K in {128, 256}; w contiguous)
 *   with ln_gamma != NULL (N == 128): y = LayerNorm(that) and, if xhat != NULL, xhat[M,128] / rstd[M] are
 *   written for rf_layernorm_bwd.
 * rf_rowblock_ffn_ln:  y = LayerNorm(x + conv2(act(conv1(x))))  for d_model 128, d_ff 256; h / z (M,256) =
 *   activation output / pre-activation side outputs for the backward pass (either may be NULL). */
int rf_rowblock_linear_supported(int N, int K, int with_ln);
int rf_rowblock_linear(const float* x, int64_t ldx, const float* w, const float* bias, const float* residual,
                       int64_t ldr, float* y, int64_t ldy, int M, int N, int K, const float* ln_gamma,
                       const float* ln_beta, float* xhat, float* rstd, float eps, void* stream);
int rf_rowblock_ffn_ln(const float* x, const float* w1, const float* b1, const float* w2, const float* b2,
                       float* h, float* z, float* y, int M, int d_model, int d_ff, int act,
                       const float* ln_gamma, const float* ln_beta, float* xhat, float* rstd, float eps,
                       void* stream);

/* Backward counterpart of the row-block kernels:  y[M,NOUT] = A[M,KC] w[KC,NOUT]  (w = the forward weight,
 * out_features = KC rows: the dX product of nn.Linear / Conv1d(k=1)), epilogue: y *= act'(dact_src) when
 * dact_mode != 0, then y += residual (the skip-branch gradient).  bf16-input MFMA only.
 * With ln_dy != NULL (KC == 128) A is not read: it is the LayerNorm backward of (ln_dy, ln_xhat, ln_rstd,
 * ln_gamma), written to dpre[M,128] in fp32 and used as bf16; dgamma / dbeta[128] += by fp32 atomics.
 * Supported: ln: NOUT in {128, 256}; plain: KC in {128, 256, 384}, NOUT == 128. */
int rf_rowblock_linear_nn_supported(int KC, int NOUT, int ln_bwd);
int rf_rowblock_linear_nn(const float* a, int64_t lda, const float* ln_dy, const float* ln_xhat,
                          const float* ln_rstd, const float* ln_gamma, float* dpre, float* dgamma,
                          float* dbeta, const float* w, const float* residual, int64_t ldr,
                          const float* dact_src, int64_t ldd, int dact_mode, float* y, int64_t ldy, int M,
                          int KC, int NOUT, void* stream);

/* Input of the fusion encoder (routeformer.py:331-345): out[b, s*T + t, :] = streams[s][b, t, :] + embeddings[s]
 * for s < S <= 4 (streams[s] == NULL: zeros, i.e. the learned output-query tokens); `streams`, `embeddings`,
 * `demb` are HOST arrays of S device pointers.  Backward: demb[s][e] += sum_{b,t} dout[b, s*T + t, e] (E <= 64;
 * NULL entries skipped); the stream gradients are the slices of dout themselves. */
int rf_assemble_streams_fwd(const float* const* streams, const float* const* embeddings, float* out, int B,
                            int T, int E, int S, void* stream);
int rf_assemble_streams_bwd(const float* dout, float* const* demb, int B, int T, int E, int S, void* stream);

/* Trajectory head = postprocess_batch (routeformer.py:367-374) + the loss recipe of the train step
 * (experiments/full_comparison.py:490-521, losses/future_discounted_mse.py:56-95, score/error.py:29,51):
 *   positions = last_gps + cumsum(out[...,:2] * motion_std + motion_mean)
 *   traj  = mean(gamma^t * SmoothL1(positions - target_gps)),  dense = mean(gamma^t * SmoothL1(out[...,2:2+E] - target_vis))
 *   loss  = traj + w * dense,  w = dense_on ? dense_ratio * traj / max(dense, 1e-6) : 0  (w is a constant in backward)
 *   ade = mean_bt ||positions - target||_2,  fde = ||positions[B-1] - target[B-1]||_F
 * scalars[6] = {traj, dense, ade, fde, loss, w}; gpos (B,P,2) is kept for the backward launch, which writes
 * dout (B,P,C) = d loss / d out * grad_loss[0] (channels beyond 2+E are NOT written: caller zero-fills). */
int rf_traj_head_fwd(const float* out, const float* last_gps, const float* target_gps,
                     const float* target_vis, float* positions, float* gpos, float* scalars, int B, int P,
                     int C, int E, float gamma, float dense_ratio, int dense_on, float motion_std,
                     float motion_mean, int64_t out_batch_stride, int64_t last_batch_stride, int64_t vis_batch_stride,
                     void* stream);
/* *_batch_stride (elements between batches; 0 = packed): `out` may be the last P rows of a longer decoder output, `last_gps`
 * the last row of the input track, `target_vis` the first P rows of a longer feature sequence -- read in place. */
int rf_traj_head_bwd(const float* out, const float* target_vis, const float* gpos, const float* scalars,
                     const float* grad_loss, float* dout, int B, int P, int C, int E, float gamma,
                     float motion_std, int64_t out_batch_stride, int64_t vis_batch_stride, void* stream);

/* ---- attention ------------------------------------------------------------------------------
 * One workgroup per (batch, head).  q[(b*LQ+l)*q_ld + h*E + e] etc. (row pitches in floats).
 * mode 0: full softmax(scale*QK^T)V          (cross_modal_transformer.py:51-69)
 * mode 1: ProbSparse, unmasked               (cross_modal_transformer.py:88-166)
 * mode 2: ProbSparse, masked (ProbMask+cumsum context)
 * index_sample: int32 [G, LQ, sample_k]; batch row b uses table b / idx_group (idx_group <= 0: one
 * table shared by all (b,h), as in the reference's single host-RNG draw, :95).  Batching several
 * encoder calls of the reference (one per video stream) into one launch keeps each call's own draw;
 * idx_group_stride = elements between consecutive tables (<= 0: LQ*sample_k, i.e. packed) -- the tables of a
 * layer then stay where the host laid the draws out, in the reference's call order, without a gather copy.
 * top_idx: int32 [B,H,n_top] selected query rows, ascending; written unless force_top (then read).
 * out_layout 0: ctx[b,l,h,:] (cross-modal variant), 1: ctx[b,h,l,:] (GPS variant,
 * layers/SelfAttentionFamily.py:165 -- the un-transposed "head scramble"). */
int rf_attn_fwd(const float* q, const float* k, const float* v, int64_t q_ld, int64_t k_ld,
                int64_t v_ld, float* ctx, int out_layout, const int32_t* index_sample, int idx_group,
                int64_t idx_group_stride, int32_t* top_idx, int force_top, int B, int H, int LQ, int LK, int E,
                int sample_k, int n_top, int mode, float scale, void* stream);
/* 1 when rf_attn_fwd runs the ProbSparse forward in its whole-score-matrix form for this shape (the full Q K^T
 * fits the LDS budget of the launch): same results, fewer passes; exported so that a profile can name the kernel
 * variant a launch maps to. */
int rf_attn_fwd_full_scores(int B, int H, int LQ, int LK, int E, int sample_k, int n_top, int mode);
int rf_attn_bwd(const float* q, const float* k, const float* v, int64_t q_ld, int64_t k_ld,
                int64_t v_ld, const float* dctx, int out_layout, const int32_t* top_idx, float* dq,
                float* dk, float* dv, int64_t dq_ld, int64_t dk_ld, int64_t dv_ld, int B, int H,
                int LQ, int LK, int E, int n_top, int mode, float scale, void* stream);
/* rf_attn_bwd whose d ctx is still `splits` split-K slabs `slab_stride` elements apart (the out-projection's input gradient left
 * by rf_gemm_partials): summed on load -- one slab-sum launch less. */
int rf_attn_bwd_slabs(const float* q, const float* k, const float* v, int64_t q_ld, int64_t k_ld, int64_t v_ld,
                      const float* dctx_slabs, int splits, int64_t slab_stride, int out_layout, const int32_t* top_idx, float* dq,
                      float* dk, float* dv, int64_t dq_ld, int64_t dk_ld, int64_t dv_ld, int B, int H, int LQ, int LK, int E,
                      int n_top, int mode, float scale, void* stream);

/* ---- fused per-sequence encoder stack (bf16-input MFMA only) ---------------------------------------
 * One workgroup owns one sequence (L <= 80 tokens, d_model 128, 8 heads of 16) and walks EVERY EncoderLayer of a
 * PerceiveEncoder in one launch: QKV projection -> ProbSparse attention -> out-projection + residual + LayerNorm ->
 * conv1 -> act -> conv2 + residual + LayerNorm (cross_modal_transformer.py:288-301, the loop of :322-324, called from
 * routeformer.py:488 for every frame of every camera stream).  The residual stream stays in registers, the GEMM
 * operands are bf16 LDS images, weight fragments come from a fragment-ordered bf16 copy (rf_seqlayer_pack).
 * RfSeqStack describes the stack with a handful of base pointers (every per-layer quantity sits at a fixed stride):
 *   wpack  : per layer one blob of rf_seqlayer_pack_bytes(d_ff) bytes (wpack_stride apart): the four weights in bf16
 *            fragment order [Wq;Wk;Wv] (384x128) | Wo (128x128) | conv1 (d_ff x 128) | conv2 (128 x d_ff), then fp32
 *            vectors bqkv[384] bo[128] b1[d_ff] b2[128] norm1.weight norm1.bias norm2.weight norm2.bias [128 each]
 *            -- written by rf_seqlayer_pack (entries with K == 0 copy an fp32 vector of N floats);
 *   idx[i] : key samples of layer i, int32 [G, L, sample_k]; sequence b uses table b / idx_group, tables idx_stride
 *            elements apart (<= 0: packed);
 *   top    : int32 [layers, B, 8, n_top] selected rows (written; read when force_top);
 *   y      : [layers, B*L, 128] layer outputs (layer i + 1 continues from registers; y[layers-1] is the result);
 *            with save == 0 only the LAST layer's output is written, to y[0] (a [B*L, 128] buffer suffices);
 *   with save != 0 what the layer-by-layer backward kernels consume, each [layers, B*L, width]: qkv (384: q | k | v
 *   packed as the projection GEMM emits them), ctx (128), xhat1 / xhat2 (128) and rstd1 / rstd2 (1) of the two
 *   norms, x1 (128: norm1 output = conv1 input), z (optional) / h (d_ff: conv1 pre-activation / activation).
 * rf_seqlayer_pack: bf16 fragment order of a weight for the 16x16x32 MFMA B operand -- out[((n/16 * K/32 + k/32) * 64
 * + lane) * 8 + j] = W[n = 16 (n/16) + lane % 16][k = 32 (k/32) + 8 (lane / 16) + j], W[n][k] = w[n * ldw + k]
 * (transpose: w[k * ldw + n], the dX orientation); N % 16 == 0, K % 32 == 0; up to RF_SEQLAYER_MAX_PACK per launch. */
#define RF_SEQLAYER_MAX_LAYERS 8
#define RF_SEQLAYER_MAX_PACK 64
typedef struct RfSeqStack {
  const void* wpack; int64_t wpack_stride;
  const int32_t* idx[RF_SEQLAYER_MAX_LAYERS]; int64_t idx_stride;
  int32_t* top;
  float* y;
  float *qkv, *ctx, *xhat1, *rstd1, *x1, *z, *h, *xhat2, *rstd2;
  void* xin; /* optional (save != 0): bf16 [layers, B*L, 128], the INPUT of every layer as the projection consumes it (layer
                0: x, layer i: y[i - 1], rounded to bf16) -- the `x` operand of the packed projection's weight gradient */
  int n_layers;
  int flags; /* bit 0: ctx, x1 and (when z is given) h are bf16 slabs -- they are bf16-rounded MFMA operands in the kernel
                anyway, and only the weight-gradient GEMMs (RfWgradEntry.x_bf16) read them again;
                bit 1: qkv is a bf16 slab (rf_seqlayer_bwd, flags bit 1, rounds q | k | v to bf16 on load either way);
                bit 2: xhat1, xhat2 and z are bf16 slabs (NOT lossless: the backward's fp32 math reads them).
                With `xin` given only the LAST layer's output is written (y = ONE [B*L, 128] slab): nothing reads the others */
} RfSeqStack;
typedef struct RfSeqPackEntry {
  const float* w; void* out; int64_t ldw; int N, K, transpose;
  int residual; /* 1: emit bf16(w - bf16(w)), the low half of a split-bf16 operand (q / k projection of the forward) */
} RfSeqPackEntry;
int rf_seqlayer_pack(const RfSeqPackEntry* entries, int count, void* stream);
/* The same packing driven by a table in DEVICE memory: `entries_dev` (count entries, validated by the caller as for
 * rf_seqlayer_pack) and `first_block_dev` (count int32: prefix sums of rf_seqlayer_pack_blocks(N, K) per entry; blocks =
 * their total).  Lets a training engine re-pack every fused stack of a model with ONE launch per step. */
int rf_seqlayer_pack_blocks(int N, int K);
int rf_seqlayer_pack_table(const RfSeqPackEntry* entries_dev, const int32_t* first_block_dev, int count, int blocks,
                           void* stream);
int64_t rf_seqlayer_pack_bytes(int d_ff);
int rf_seqlayer_supported(int L, int d_model, int n_heads, int d_ff, int sample_k, int n_top);
 /* drop_p > 0: nn.Dropout(p) of the layers in train mode (cross_modal_transformer.py:295,298,299), masks from the
 * counter-based generator below -- layer i draws sites drop_site0 + 3 i + {0: attention output [B*L,128], 1: hidden
 * activation [B*L,d_ff] (saved h = the dropped activation), 2: conv2 output [B*L,128]}, element index row * cols + col:
 * exactly the masks rf_dropout generates for the same (site, element), which the layer-by-layer backward re-applies. */
int rf_seqlayer_fwd(const RfSeqStack* stack, const float* x, int B, int L, int d_model, int n_heads, int d_ff, int act,
                    int sample_k, int n_top, int idx_group, int force_top, int save, float scale, float eps,
                    float drop_p, const void* rng_state, int drop_site0, void* stream);

/* Backward of the same stack, one launch (one workgroup per sequence, the gradient of the residual stream in
 * registers from the last layer to the first): per layer LayerNorm-2 backward -> conv2^T -> act' -> conv1^T + skip ->
 * LayerNorm-1 backward -> out-projection^T -> ProbSparse attention backward -> packed q|k|v projection^T + skip
 * (the reverse of cross_modal_transformer.py:288-301; replaces 4 rf_rowblock_linear_nn + 1 rf_attn_bwd per layer).
 *   wpack  : per layer one blob of rf_seqlayer_bwd_pack_bytes(d_ff) bytes -- rf_seqlayer_pack with transpose = 1 of
 *            conv2 (N = d_ff, K = 128) | conv1 (N = 128, K = d_ff) | Wo (N = 128, K = 128) | [Wq;Wk;Wv] (N = 128,
 *            K = 384), then fp32 norm1.weight[128] norm2.weight[128];
 *   qkv, xhat1, rstd1, xhat2, rstd2, top : the saves of rf_seqlayer_fwd; zsrc = z (GELU) or h (ReLU);
 *   dpre2, dz, dpre1, dqkv : [layers, B*L, 128 | d_ff | 128 | 384] -- written: the gradients of the conv2 output,
 *            conv1 output, out-projection output and packed projection output, i.e. the `dy` operands of the four
 *            weight-gradient GEMMs of each layer (values rounded to bf16, as those GEMMs round them anyway);
 *   dgamma1/dbeta1/dgamma2/dbeta2[i] : [128] accumulators of layer i's LayerNorm parameters (atomicAdd).
 * dy / dx: [B*L, 128] gradient of the stack output / input.
 * drop_p > 0: the masks rf_seqlayer_fwd drew (same rng_state step, same drop_site0) are regenerated in the kernel; dpre2
 * and dpre1 then hold the gradients BEHIND the conv2-output / attention-output dropout (what the weight gradients of
 * conv2 / the out-projection consume), the skip connections carry the unmasked ones. */
typedef struct RfSeqStackBwd {
  const void* wpack; int64_t wpack_stride;
  const float *qkv, *xhat1, *rstd1, *zsrc, *xhat2, *rstd2;
  const int32_t* top;
  float *dpre2, *dz, *dpre1, *dqkv;
  float* dgamma1[RF_SEQLAYER_MAX_LAYERS]; float* dbeta1[RF_SEQLAYER_MAX_LAYERS];
  float* dgamma2[RF_SEQLAYER_MAX_LAYERS]; float* dbeta2[RF_SEQLAYER_MAX_LAYERS];
  int n_layers;
  int flags; /* bit 0: dpre2 / dz / dpre1 / dqkv are bf16 slabs (RfWgradEntry.dy_bf16 for the weight-gradient GEMMs);
                bit 1: qkv points to the bf16 slab rf_seqlayer_fwd wrote with ITS flags bit 1;
                bit 2: xhat1 / xhat2 are bf16 slabs; bit 3: zsrc is a bf16 slab */
} RfSeqStackBwd;
int64_t rf_seqlayer_bwd_pack_bytes(int d_ff);
int rf_seqlayer_bwd(const RfSeqStackBwd* stack, const float* dy, float* dx, int B, int L, int d_model, int n_heads,
                    int d_ff, int act, int n_top, float scale, float drop_p, const void* rng_state, int drop_site0,
                    void* stream);

/* ---- video ingest ---------------------------------------------------------------
 * rf_resize_area: cv2.resize(..., interpolation=cv2.INTER_AREA) of io/dataset.py:1476-1497 (down-scaling, factor < 1)
 *   on uint8 planes [n_planes][H][W] -> [n_planes][h][w]: coverage-weighted mean of the source pixels under each
 *   output pixel, rounded half-to-even and saturated (integer factors: plain s x s block means).
 * rf_frame_hash: keys[f] = 64-bit content hash (never 0) of frame frame_ids[f] (NULL: f) of `frames` (frames of
 *   bytes_per_frame bytes, back to back): the key of the backbone-feature cache that stands in for @torchcache(persistent=True)
 *   (models/video_backbone/__init__.py:14-32).
 * rf_cache_lookup: slots[i] = token slot of keys[i] in the open-addressing table (table_keys[capacity] uint64, 0 =
 *   empty; table_slots[capacity] int32; capacity a power of two) or -1; *misses += number of -1 (caller zeroes it).
 * rf_cache_insert: for every i with slots[i] < 0 claim a table entry and the next free token slot (*next_slot,
 *   atomically; at most n_slots); duplicates inside one call and a full cache stay -1 (look up again afterwards). */
/* Frame sub-sampling into staging buffers (the `video[:, frame_idx]` of routeformer.py:470-480 as one launch for all
 * camera streams of a step): dst[b][f] = src[b][idx[f]], frames of frame_bytes bytes, src (B, T, ...) and dst (B, F, ...)
 * contiguous, idx = F int64 frame numbers in DEVICE memory.  Up to RF_GATHER_MAX clips per launch. */
#define RF_GATHER_MAX 8
typedef struct RfGatherEntry {
  const void* src; void* dst; const int64_t* idx;
  int B, T, F, pad;
  int64_t frame_bytes;
} RfGatherEntry;
int rf_gather_frames(const RfGatherEntry* entries, int count, void* stream);
int rf_resize_area(const uint8_t* src, uint8_t* dst, int64_t n_planes, int H, int W, int h, int w, void* stream);
int rf_frame_hash(const void* frames, const int64_t* frame_ids, int64_t n_frames, int64_t bytes_per_frame, uint64_t* keys,
                  int64_t seed, void* stream);
int rf_cache_lookup(const uint64_t* keys, int n, const uint64_t* table_keys, const int32_t* table_slots, int capacity,
                    int32_t* slots, int32_t* misses, void* stream);
int rf_cache_insert(const uint64_t* keys, int n, uint64_t* table_keys, int32_t* table_slots, int capacity,
                    int32_t* next_slot, int n_slots, int32_t* slots, void* stream);

/* ---- dropout on the trainable path -------------------------------------------------------------
 * nn.Dropout of cross_modal_transformer.py:49,63,220-231,285-299, gps_backbone/layers/Embedding.py:122-126,
 * layers/TransformerEncoderDecoder.py:41-50,102-113.  Masks are never stored: the keep-bit of element e of dropout
 * site `site` in training step `step` is Philox4x32-10(counter = (e/4, site, step), key = seed)[e % 4] >= p * 2^32,
 * so a backward kernel regenerates the mask its forward drew.  rng_state = two uint64 in DEVICE memory
 * {seed, step}: rf_rng_seed sets them, rf_rng_advance does step += 1 (one launch at the head of every training
 * step, also when the step is replayed from a HIP graph).  `mask_in` (one byte per element, non-zero = keep)
 * replaces the generator: the parity tests feed the masks the reference run recorded.
 * rf_dropout: y = x * keep / (1 - p) over n contiguous floats (x == y allowed; the same call is its own backward);
 * x == y == NULL with mask_out: only materialise the keep-mask of that site (tests).
 * rf_attn_fwd_drop / rf_attn_bwd_drop: rf_attn_fwd / rf_attn_bwd with dropout on the softmax probabilities
 * A[b,h,q,s] (element index ((b*H + h)*LQ + q)*LK + s) -- FullAttention only: mode 0, or mode 2 with every query
 * row imposed (the causal form); drop_p == 0 is exactly rf_attn_fwd / rf_attn_bwd. */
int rf_rng_seed(void* rng_state, int64_t seed, int64_t step, void* stream);
int rf_rng_advance(void* rng_state, void* stream);
int rf_dropout(const float* x, float* y, int64_t n, float p, const void* rng_state, int site, const uint8_t* mask_in,
               uint8_t* mask_out, void* stream);
int rf_attn_fwd_drop(const float* q, const float* k, const float* v, int64_t q_ld, int64_t k_ld, int64_t v_ld,
                     float* ctx, int out_layout, const int32_t* index_sample, int idx_group, int64_t idx_group_stride,
                     int32_t* top_idx, int force_top, int B, int H, int LQ, int LK, int E, int sample_k, int n_top,
                     int mode, float scale, float drop_p, const void* rng_state, int drop_site,
                     const uint8_t* drop_mask, void* stream);
int rf_attn_bwd_drop(const float* q, const float* k, const float* v, int64_t q_ld, int64_t k_ld, int64_t v_ld,
                     const float* dctx, int out_layout, const int32_t* top_idx, float* dq, float* dk, float* dv,
                     int64_t dq_ld, int64_t dk_ld, int64_t dv_ld, int B, int H, int LQ, int LK, int E, int n_top,
                     int mode, float scale, float drop_p, const void* rng_state, int drop_site,
                     const uint8_t* drop_mask, void* stream);

/* ---- optimizer (experiments/full_comparison.py:694-702,829-830) -------------------------------
 * sumsq[k] = partial sum of g^2 of workgroup k, k < rf_sumsq_parts(n) (no atomics: the consumer adds the
 * partials in ascending order, so the global norm is reproducible bit for bit across runs and ranks). */
int rf_sumsq_parts(int64_t n);
int rf_sumsq(const float* g, int64_t n, float* sumsq, void* stream);
/* AdamW with global-norm clipping fused: g *= min(1, max_norm/(sqrt(sum_k sumsq[k])+1e-6)); decoupled
 * weight decay; bias-corrected moments (torch.optim.AdamW semantics).  max_norm<=0 disables. */
int rf_adamw_clip(float* p, const float* g, float* m, float* v, int64_t n, const float* sumsq, int sumsq_parts,
                  float max_norm, float lr, float beta1, float beta2, float eps, float wd, int step,
                  float grad_scale, void* stream);
/* Same update with the scalars read from device memory, for a launch replayed from a HIP graph:
 * hyper[10] = {pending (0: leave everything untouched), max_norm, lr, beta1, beta2, eps, wd, 1-beta1^t,
 * sqrt(1-beta2^t), grad_scale}.  `sumsq` may cover a larger buffer than [p, p+n): the clip coefficient is global.
 * max_blocks > 0 caps the grid (0: 4096 workgroups): a slice updated on a side stream underneath other work is
 * throttled this way so that its 28 B/parameter stream does not starve the kernels it runs next to. */
#define RF_ADAMW_HYPER 10
int rf_adamw_clip_dev(float* p, const float* g, float* m, float* v, int64_t n, const float* sumsq, int sumsq_parts,
                      const float* hyper, int max_blocks, void* stream);

/* ---- backbone input / output head (routeformer.py:210-233,279-292) ------------------------------
 * rf_motion_input: x[b,t,:] = [R(-origin_b) motion | (angle - origin_b)/pi | |motion| | |motion|_t - |motion|_{t-1}
 * | visual[b,t,:]] with origin_b = the angle of the last step (rotate_motion) or the first; without rotate_motion the
 * motion channels are copied unrotated.  x: (B,T,5+E), origin: (B) (kept for the output rotation), zero_visual
 * writes zeros instead of `visual` (the `_only_motion` ablation).  Motion carries no gradient; d visual = dx[..., 5:].
 * rf_rotate_head: out = in with channels 0,1 rotated by sign * origin_b, other channels copied (sign = +1: the
 * un-rotation of the predicted motion; -1: its backward). */
int rf_motion_input(const float* motion, const float* visual, float* x, float* origin, int B, int T, int E,
                    int rotate_motion, int zero_visual, void* stream);
int rf_rotate_head(const float* in, const float* origin, float* out, int B, int P, int C, float sign, void* stream);

/* ---- row-tile half of a ProbSparse encoder layer (csrc/enclayer.hip) -- for sequences beyond the fused stack's L <= 80
 * (the fusion `video_encoder`, routeformer.py:85-92,346: L = 160 / 320).  One launch takes the attention output of layer l
 * through out-projection + residual + LayerNorm1 -> conv1 -> act -> conv2 + residual + LayerNorm2 AND the packed q | k | v
 * projection of layer l + 1 (cross_modal_transformer.py:288-301), 32 / 48 rows of the flattened (B L, 128) activations per
 * workgroup; rf_attn_fwd runs between two of them: a layer = 2 launches instead of 4.  Weights: the per-layer blobs of
 * rf_seqlayer_pack.  ctx == NULL: projection only (first layer's q | k | v from x); wpack_next == NULL: no projection
 * (last layer).  save = 1: also write what the layer-by-layer backward consumes (z only for GELU). */
int rf_enclayer_tile_supported(int d_model, int n_heads, int d_ff);
int rf_enclayer_tile_fwd(const float* ctx, const float* x, const void* wpack, const void* wpack_next, float* y,
                         float* qkv_next, float* xhat1, float* rstd1, float* x1, float* z, float* h, float* xhat2,
                         float* rstd2, int M, int d_model, int n_heads, int d_ff, int act, int save, float eps,
                         float drop_p, const void* rng_state, int drop_site, void* stream);
/* drop_p > 0: nn.Dropout(p) of the layer in train mode (cross_modal_transformer.py:295,298,299), Philox masks of
 * sites drop_site + {0: attention output, 1: hidden activation (saved h = the dropped activation), 2: conv2 output}, element
 * index row * cols + col -- the masks rf_dropout / rf_seqlayer_fwd generate for the same (site, element). */

/* Backward counterpart: between two rf_attn_bwd launches one row-tile launch does the packed q | k | v projection^T of
 * layer l + 1 (+ skip = its d pre-norm-1) -> LayerNorm-2 backward -> conv2^T -> act' -> conv1^T + skip -> LayerNorm-1 backward
 * -> out-projection^T of layer l (2 launches per layer instead of 5).  Weights: the transposed blobs of rf_seqlayer_bwd
 * (rf_seqlayer_bwd_pack_bytes each).  dy: the stack's output gradient (last layer; then dqkv / skip / wpack_next_t NULL);
 * wpack_t NULL: projection^T only -> dx (the stack's input gradient).  Outputs per layer: dpre2, dz (weight-gradient
 * operands, bf16-rounded), dpre1 (fp32), dctx (for rf_attn_bwd); dgamma / dbeta of both norms are ACCUMULATED (atomics). */
int rf_enclayer_tile_bwd(const float* dy, const float* dqkv, const float* skip, const void* wpack_t, const void* wpack_next_t,
                         const float* xhat1, const float* rstd1, const float* zsrc, const float* xhat2, const float* rstd2,
                         float* dpre2, float* dz, float* dpre1, float* dctx, float* dx, float* dgamma1, float* dbeta1,
                         float* dgamma2, float* dbeta2, int M, int d_model, int n_heads, int d_ff, int act, float* dskip,
                         float drop_p, const void* rng_state, int drop_site, void* stream);
/* drop_p > 0: the forward's masks are regenerated (same rng_state step, same drop_site); dpre2 / dpre1 then hold the gradients
 * BEHIND the conv2-output / attention-output dropout (the weight-gradient operands) and `dskip` (M, 128) receives the unmasked
 * d pre-norm-1, which is what the next launch takes as `skip`. */

/* ---- row-local chains of a d_model = 64 decoder layer (csrc/rowchain.hip) -- the gaze-video PerceiveDecoder
 * (cross_modal_transformer.py:304-365,436-476).  ONE launch takes an attention output `a` through
 *     out-projection + residual x -> LayerNorm  => x1   [-> conv1 -> act -> conv2 + residual x1 -> LayerNorm => y]
 *     [-> projection of the result (the next attention launch's q, or q | k | v) => proj]
 * on 16 / 32-row tiles of the flattened (M, 64) activations: a decoder layer is self attention, chain, k | v projection of
 * the memory, cross attention, chain + FFN + next projection -- 5 launches instead of 13.  Weights are the fp32 masters
 * (bf16 MFMA fragments are formed from them in registers); saves (xhat / rstd of the norms, z, h) are what rf_rowchain_bwd
 * and the weight-gradient GEMMs consume.  rf_rowchain_bwd: gradient of the chain output (`dyin`, from other consumers of
 * it, and / or `dproj` through the projection) -> [LayerNorm-2 backward -> conv2^T -> act' -> conv1^T + skip] -> LayerNorm-1
 * backward -> out-projection^T: writes da (gradient of `a`), dpre1 (= gradient of the residual input x, and the dy operand
 * of the out-projection's weight gradient), dpre2 / dz (dy operands of conv2 / conv1), ACCUMULATES dgamma / dbeta (atomics). */
typedef struct RfRowChain {
  const float *a, *x;
  const float *wo, *bo, *g1, *be1;
  const float *w1, *b1, *w2, *b2, *g2, *be2; /* w1 == NULL: no FFN block */
  const float *wp, *bp;                      /* wp == NULL: no projection; bp may be NULL */
  float *x1, *y, *proj;
  float *xhat1, *rstd1, *z, *h, *xhat2, *rstd2; /* training saves, each may be NULL */
  int d_model, d_ff, n_proj, act;
  float eps;
  int drop_site; /* drop_p > 0: Philox masks of sites drop_site + {0: out-projection output, 1: hidden activation (saved h = the
                    dropped activation), 2: conv2 output}, element index row * cols + col -- the masks rf_dropout generates */
} RfRowChain;
typedef struct RfRowChainBwd {
  const float *dproj, *dyin, *wp;
  const float *w1, *w2, *g2, *xhat2, *rstd2, *zsrc; /* zsrc = z (GELU) or h (ReLU) */
  float *dpre2, *dz, *dg2, *db2;
  const float *wo, *g1, *xhat1, *rstd1;
  float *dpre1, *da, *dg1, *db1;
  int d_model, d_ff, n_proj, act;
  float* dx;     /* drop_p > 0: (M, 64) the UNMASKED d pre-norm-1 = gradient of the residual input (dpre1 / dpre2 then hold the
                    gradients behind the out-projection / conv2-output dropout: the weight-gradient operands) */
  int drop_site, pad;
} RfRowChainBwd;
int rf_rowchain_supported(int d_model, int d_ff, int n_proj);
int rf_rowchain_fwd(const RfRowChain* chain, int M, float drop_p, const void* rng_state, void* stream);
int rf_rowchain_bwd(const RfRowChainBwd* chain, int M, float drop_p, const void* rng_state, void* stream);

/* ---- small tensor plumbing of the hot path as single launches (csrc/smallops.hip) ---------------------------
 * rf_median_windows: y (B,target,C) = lower median (torch.median: NaN wins) of the consecutive windows of T / target
 *   samples of x (B,T,C) -- `median_downsampler`, routeformer/utils/filter.py:5-43 (gaze 200 Hz -> seq_len).
 * rf_motion_diff: motion (B,T,2): row 0 = 0, row t = normalise(gps[t] - gps[t-1]) (routeformer.py:284-292).
 * rf_time_table / _bwd: out[l,c] = l * w[c] + pe[l,c] (DataEmbedding's time feature + positional table,
 *   gps_backbone/layers/Embedding.py:99-126) and dw[c] (+)= sum_l l * dout[l,c].
 * rf_timeline_scatter / _gather: out (N,T,E) = zeros with out[:, idx[f]] = feats[:, f] (routeformer.py:443-459) and the
 *   gradient gather; idx int64 (F).
 * rf_smart_tail_fwd / _bwd: y (B,L+P,C) = cat(x, x[:, L-1] repeated P times (smart) or zeros) -- the Informer decoder
 *   input (gps_backbone/Informer.py:125-136); dx = dy[:, :L] (+ extra) with the tail's gradients added to the last row. */
int rf_median_windows(const float* x, float* y, int B, int T, int C, int target, void* stream);
int rf_motion_diff(const float* gps, float* motion, int B, int T, int normalize, float mean, float std_, void* stream);
int rf_time_table(const float* w, const float* pe, float* out, int L, int d, void* stream);
int rf_time_table_bwd(const float* dout, float* dw, int L, int d, int accumulate, void* stream);
int rf_timeline_scatter(const float* feats, const int64_t* idx, float* out, int64_t N, int T, int F, int E, void* stream);
int rf_timeline_gather(const float* dout, const int64_t* idx, float* dfeats, int64_t N, int T, int F, int E, void* stream);
int rf_smart_tail_fwd(const float* x, float* y, int B, int L, int P, int C, int smart, void* stream);
int rf_smart_tail_bwd(const float* dy, const float* extra, float* dx, int B, int L, int P, int C, int smart, void* stream);
/* dst (rows, ld) = [src (rows, cols) | zero columns]: the K-padded copy of a Conv1d(k=3) weight whose 3 c_in is not a
 * multiple of 4 (the GPS token embedding, layers/Embedding.py:28-46 with c_in = 69: 207 -> 208 keeps the embedding GEMMs on
 * the 16-B vector path), and the way back for its gradient: dst (rows, cols) (+)= src (rows, ld)[:, :cols]. */
int rf_pad_cols(const float* src, float* dst, int rows, int cols, int ld, void* stream);
int rf_unpad_cols(const float* src, float* dst, int rows, int cols, int ld, int accumulate, void* stream);

/* ---- cross-resolution fusion of the conv trunk without intermediate maps (hrnetv2.py:250-271,453-498;
 * InverseForm.py:66-67; routeformer.py:478-487) ---------------------------------------------------------------
 * rf_fuse_upsample_sum: for each entry  out = [relu](base + base2 + sum_s bilinear_up(src[s]))  (terms added in this
 * order; base / base2 optional full-resolution maps (N,Ho,Wo,C), src[s] low-resolution maps (N,Hi[s],Wi[s],C), bilinear
 * = F.interpolate(align_corners=False); `out` may alias base).  Up to RF_FUSE_MAX entries per launch: all output
 * branches of one HighResolutionModule's fuse layer.  Maps in act_dtype (0 fp32, 1 bf16), channels innermost, C % 4 == 0.
 * rf_concat_pool_tokens: tokens[n, t, :] (fp32, (N,65,sum C)) = AdaptiveAvgPool2d((8,8)) of the channel-concatenation of
 * the maps up-sampled to maps[0]'s resolution, token 64 = -1: the concatenated full-resolution map is never stored. */
#define RF_FUSE_MAX 4
typedef struct RfFuseEntry {
  const void* base; const void* base2; const void* src[3]; void* out;
  int N, Ho, Wo, C, n_src, relu;
  int Hi[3], Wi[3];
} RfFuseEntry;
int rf_fuse_upsample_sum(const RfFuseEntry* entries, int count, int act_dtype, void* stream);
int rf_concat_pool_tokens(const void* const* maps, const int32_t* H, const int32_t* W, const int32_t* C, int n_maps,
                          int act_dtype, float* tokens, int N, void* stream);

/* ---- gradient exchange over RCCL (SURVEY 8(b) "Comm"; replaces Lightning's DDPStrategy(process_group_backend=
 * "nccl"), experiments/full_comparison.py:794,832: bucketed gradient all-reduce overlapped with backward) ------
 * A communicator owns an explicit, highest-priority HIP communication stream and two events.
 *   rf_comm_unique_id(id)            rank 0 only: 128 bytes to hand to every rank (any host channel)
 *   rf_comm_init(&comm, id, rank, world)   collective; uses the calling thread's current device
 *   rf_comm_allreduce_bucket(comm, buf, count, dtype, average, producer_stream)
 *        in-place all-reduce (SUM, or the mean when `average`) of `count` elements (dtype 0 fp32, 1 bf16) on the
 *        communication stream, ordered after everything enqueued so far on `producer_stream` (event record + wait);
 *   rf_comm_broadcast(...)           the one-time parameter broadcast of DDP construction, same discipline
 *   rf_comm_wait(comm, consumer_stream)   `consumer_stream` waits on the DEVICE for every collective launched so far
 * No call blocks the host.  RCCL is resolved at run time (dlsym, then librccl.so): rf_comm_available() says whether
 * it was found; the library itself loads without it. */
int rf_comm_available(void);
int rf_comm_unique_id(void* id_out_128_bytes);
int rf_comm_init(void** comm_out, const void* id_128_bytes, int rank, int world);
int rf_comm_allreduce_bucket(void* comm, void* buf, int64_t count, int dtype, int average, void* producer_stream);
int rf_comm_broadcast(void* comm, void* buf, int64_t count, int dtype, int root, void* producer_stream);
int rf_comm_wait(void* comm, void* consumer_stream);
int rf_comm_destroy(void* comm);

/* ---- measurement -----------------------------------------------------------------------------
 * rf_kernel_timer_arm(): the NEXT kernel this thread launches through the library carries a start / stop event
 * pair bracketing exactly its dispatch (hipExtLaunchKernelGGL) -- the kernel's own execution time, what rocprofv3's
 * kernel trace reports.  Up to 4096 arms may be outstanding (launches keep queueing: no synchronisation, so the
 * timed kernels run in a busy stream at load clocks).  rf_kernel_timer_collect(us, capacity): waits, writes the
 * durations in microseconds in arm order (< 0: nothing was launched for that arm), resets, returns the count.
 * Not for use inside stream capture. */
int rf_kernel_timer_arm(void);
int rf_kernel_timer_collect(float* us, int capacity);

#ifdef __cplusplus
}
#endif
#endif /* RF_HIP_H */
