/*
 * usseg.h - C ABI of libusseg_hip.so: hand-written gfx950 (MI355X / CDNA4) kernels for the
 * ResNeSt/UNet segmentation hot path of silverlight6/Ultrasound_Modeling.
 *
 * The reference (TensorFlow/Keras) has no FFI or operator boundary of its own: every
 * kernel it runs is implicit inside tf.keras.layers.*.  Each entry point below therefore cites the
 * reference call sites whose implicit framework kernel it replaces (paths relative to the reference
 * repo root).  INTEGRATION.md shows the binding a maintainer of the reference would add.
 *
 * Contract (all functions)
 *  - plain pointers and sizes only; every pointer is a DEVICE pointer unless stated otherwise;
 *  - the caller owns every buffer, including workspaces; the library allocates no device memory,
 *    never synchronises and never throws.  Mutable host state is THREAD-LOCAL: the error string; the queue of
 *    deferred finishing reductions of every stream between usseg_defer_begin() and usseg_defer_end() (defer.hip);
 *    the opt-in launch timer of usseg_prof_* (runtime.hip) - so host threads that drive different streams do not
 *    share any of it (a stream's begin / launches / flush / end come from one thread).  The only process-wide state
 *    is immutable after first use: the USSEG_* planner switches, each read from the environment once (DESIGN.md
 *    section 4), and the per-kernel "dynamic LDS size set" latches;
 *  - one producer per destination between two flushes is NOT required: deferred finishes that share a
 *    destination are issued as separate, stream-ordered launches (defer.hip);
 *  - all work is enqueued on the hipStream_t passed in (graph-capture safe);
 *  - return value: 0 on success, negative UssegStatus on error (usseg_last_error() has the text).
 *
 * Data layout
 *  - activations: NHWC, bf16, channel stride `ld` (elements) >= C; C and ld are multiples of 8 and
 *    the base pointer is 16-byte aligned, so a pixel's channels can be read 8 at a time.  Channels
 *    beyond the layer's logical width are kept exactly zero by every kernel ("pad channels").
 *  - packed conv operand ("Wp"): bf16 matrix [Nrows][Kw], Nrows = roundup(N,16), K index =
 *    tap*Cin + c  (Cin = physical input channels of the op), row stride Kw = ntaps*Cin.
 *    Built from the fp32 Keras-layout master weights by usseg_pack_weight().
 *  - per-channel vectors (bias, gamma, beta, ...) are fp32, padded with zeros to the physical width.
 */
#ifndef USSEG_H
#define USSEG_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef void* usseg_stream_t; /* a hipStream_t */

typedef enum UssegStatus {
  USSEG_OK = 0,
  USSEG_ERR_BAD_ARG = -1,
  USSEG_ERR_UNSUPPORTED = -2,
  USSEG_ERR_LAUNCH = -3
} UssegStatus;

typedef enum UssegAct { USSEG_ACT_NONE = 0, USSEG_ACT_LRELU = 1, USSEG_ACT_RELU = 2, USSEG_ACT_ELU = 3, USSEG_ACT_GELU = 4 } UssegAct;

enum {
  USSEG_OUT_F32 = 1,   /* y is float* (ldy counted in floats); default bf16 */
  USSEG_ACCUMULATE = 2 /* y += result (residual pointer == y is also allowed) */
};

/* Geometry of one Conv2D (stride 1, SAME, dilation d) or Conv2DTranspose (stride 2, SAME). */
typedef struct UssegConvDesc {
  int32_t B, H, W;     /* batch and INPUT spatial size of the forward op (tconv output is 2H x 2W) */
  int32_t Cin, Cout;   /* physical channel counts (multiples of 8) of the forward input / output */
  int32_t ldx, ldy;    /* channel strides (elements) of the forward input / output buffers */
  int32_t ksize;       /* conv: 1 or 3; tconv: 3 or 4 */
  int32_t dilation;    /* conv only */
  int32_t act;         /* UssegAct fused into the forward epilogue */
  float alpha;         /* LeakyReLU / ELU alpha */
  int32_t flags;       /* USSEG_OUT_F32 | USSEG_ACCUMULATE */
} UssegConvDesc;

const char* usseg_last_error(void);
int usseg_version(void);

/* ---- convolution: Conv2D 3x3 / dilated 3x3 / 1x1, stride 1, SAME -------------------------------
 * replaces tf.keras.layers.Conv2D at ResNest.py:14,17,21,77,82,122,128,160,166; Decoder.py:11-25,
 * 36-50,103; VisionTransformer.py:106; TBI_ResNest.py:83,85,88,140,143,162,167,189,195.
 * y = act(conv(x, Wp) + bias) (+ residual).  bias may be NULL; residual (bf16, stride ldr) may be NULL. */
int usseg_conv2d_fwd(const UssegConvDesc* d, const void* x, const void* wp_fwd, const float* bias,
                     const void* residual, int32_t ldr, void* y, usseg_stream_t stream);
/* Conv2D followed by an inference-mode BatchNormalization and the activation in ONE pass (ResNest.py:17-23; Decoder.py:67-76):
 * y = act(scale[n] * conv(x, Wp)[n] + shift[n]) with scale = gamma*rsqrt(var+eps), shift = beta - mean*scale + scale*conv_bias
 * (usseg_bn_fold_batched builds both).  The pre-normalisation tensor is never written; the backward recovers what it needs
 * from y (usseg_norm_act_bwd mode 2). */
int usseg_conv2d_fwd_affine(const UssegConvDesc* d, const void* x, const void* wp_fwd, const float* scale, const float* shift,
                            const void* residual, int32_t ldr, void* y, usseg_stream_t stream);
typedef struct UssegBnFoldJob {   /* one BatchNormalization layer (device pointers; C logical, Cp physical channels) */
  const float *gamma, *beta, *mean, *var;
  const float* bias;               /* bias of the conv in front of it, or NULL */
  float *scale, *shift;            /* outputs, Cp floats each (zero past C) */
  int32_t C, Cp;
  float eps;
  int32_t reserved;
} UssegBnFoldJob;
int usseg_bn_fold_batched(const UssegBnFoldJob* jobs_dev, int32_t njobs, usseg_stream_t stream);
/* dx = conv_transpose(dy, W) (+ residual): the backward-data of the op above.  wp_dgrad is the operand
 * packed with in/out swapped (K index = tap*Cout + co).  dx has d->Cin channels, stride d->ldx. */
int usseg_conv2d_dgrad(const UssegConvDesc* d, const void* dy, const void* wp_dgrad, const void* residual,
                       int32_t ldr, void* dx, usseg_stream_t stream);
/* dW[tap][ci][co] += sum_pixels x * dy, accumulated into a [ntaps][Cin][Cout] fp32 buffer (zeroed or holding a
 * running sum).  ws (may be NULL) is a caller-owned fp32 workspace of ws_floats floats for the split-K partial slabs of
 * the 3x3 kernel (any size helps; 9*Cin*Cout*512 is always enough); without it the partials are combined with atomics. */
int usseg_conv2d_wgrad(const UssegConvDesc* d, const void* x, const void* dy, float* dw_scratch, float* ws,
                       int64_t ws_floats, usseg_stream_t stream);

/* Backward-data of several parallel convolutions that read the SAME input (Decoder.py:67-75,79-87: conv*_1 1x1 and the three
 * dilated 3x3 convs) in one pass: dx = sum_b dgrad_b(dy[..., ch_off[b] : ch_off[b]+Cb]).  d: B,H,W, Cin/ldx of dx, ldy = channel
 * stride of the concatenated dy tensor.  wp_cat: bf16 [roundup(Cin,16)][Ktot], Ktot = sum_b ksize_b^2 * Cb, branch b's taps at
 * columns (taps before b)*Cb + tap*Cb + co (tap = kh*k + kw of the forward kernel).  At most 32 taps in total. */
int usseg_conv2d_dgrad_branches(const UssegConvDesc* d, int32_t nbranches, const int32_t* ksize, const int32_t* dilation,
                                const int32_t* ch_off, int32_t Cb, const void* dy, const void* wp_cat, const void* residual,
                                int32_t ldr, void* dx, usseg_stream_t stream);

/* Weight gradient written straight into the framework's variables: up to 4 rectangular blocks of the physical
 * [ntaps][Cin_phys][Cout_phys] gradient go to strided destinations (Keras kernel [k,k,Cin,Cout] / [k,k,Cout,Cin] with the
 * LOGICAL channel counts; one block per cardinal path for the block-diagonal grouped convs, ResNest.py:91-96).  Elements
 * outside every block (channel padding, off-diagonal blocks) are dropped.  dst[t*sT + i*sI + o*sO] += dW[t][i_off+i][o_off+o]. */
typedef struct UssegWgradBlock {
  float* dst;
  int64_t sT, sI, sO;          /* element strides of the destination per tap / input channel / output channel */
  int32_t i_off, o_off, ni, no; /* the block: input channels [i_off, i_off+ni), output channels [o_off, o_off+no) */
} UssegWgradBlock;
typedef struct UssegWgradDst {
  int32_t nblocks;             /* 1..4 */
  int32_t reserved;
  UssegWgradBlock blk[4];
} UssegWgradDst;
int usseg_conv2d_wgrad_mapped(const UssegConvDesc* d, const void* x, const void* dy, const UssegWgradDst* dst, float* ws,
                              int64_t ws_floats, usseg_stream_t stream);
int usseg_tconv2d_wgrad_mapped(const UssegConvDesc* d, const void* x, const void* dy, const UssegWgradDst* dst, float* ws,
                               int64_t ws_floats, usseg_stream_t stream);

/* Several independent 3x3 weight gradients in one launch (the DecoderBlock's dilation branches; the stem's three convs, ResNest.py:39-44): same
 * semantics as usseg_conv2d_wgrad / usseg_conv2d_wgrad_mapped per job (dst NULL: accumulate into dw; dst set: dw ignored).  The jobs share one
 * grid when they have the same (B, H, W), the same number of 128-pixel groups and - channel counts may differ - the same tile shape and tile
 * count of the halo kernel; otherwise they are launched one after the other. */
typedef struct UssegWgradJob {
  UssegConvDesc desc;
  const void* x;
  const void* dy;
  float* dw;
  const UssegWgradDst* dst;
} UssegWgradJob;
int usseg_conv2d_wgrad_multi(int32_t njobs, const UssegWgradJob* jobs, float* ws, int64_t ws_floats, usseg_stream_t stream);

/* Deferred finishing reductions.  Between usseg_defer_begin(stream, ...) and usseg_defer_end(stream) the small kernels
 * that add per-workgroup partial sums into gradient variables (the tail of usseg_norm_act_bwd, usseg_colsum,
 * usseg_conv2d_wgrad[_mapped], usseg_tconv2d_wgrad_mapped on that stream) are queued and run as a few batched launches at
 * usseg_defer_flush / _end (or earlier when a workspace fills).  The gradients those calls produce are therefore complete only
 * after the flush: flush before anything on the stream reads them (the optimiser; training-mode BatchNorm's backward fix).
 * The two workspaces (fp32, caller-owned, must outlive the end call) replace the per-call ws arguments while deferring;
 * either may be NULL to keep that family immediate.  One context per stream. */
int usseg_defer_begin(usseg_stream_t stream, float* reduce_ws, int64_t reduce_floats, float* wgrad_ws, int64_t wgrad_floats);
int usseg_defer_flush(usseg_stream_t stream);
int usseg_defer_end(usseg_stream_t stream);

/* Several independent 3x3 convolutions in ONE launch: the DecoderBlock's parallel dilation branches
 * (Decoder.py:14-25,39-50: the conv2_x / conv3_x / conv4_x layers read the same input and write disjoint channel slices of the
 * concatenated output, Decoder.py:67-75,79-87).  Semantics = usseg_conv2d_fwd / usseg_conv2d_dgrad called once per job
 * in order; the jobs must not depend on each other (disjoint outputs).  1 <= njobs <= 4.  Jobs that cannot share a
 * grid are launched one after the other.  Forward only: a 1x1 job (ksize 1, dilation 1: the block's conv1_x, Decoder.py:11-13) beside at
 * least one 3x3 job on the same input rides in their launch as a centre-tap-only job of the 3x3 geometry (its wp is the 1x1 conv's own
 * [N][Cin] operand). */
typedef struct UssegConvJob {
  UssegConvDesc desc;
  const void* x;        /* fwd: input x;  dgrad: dy */
  const void* wp;       /* packed operand (forward packing for fwd, in/out-swapped packing for dgrad) */
  const float* bias;    /* fwd only, may be NULL */
  const void* residual; /* bf16, may be NULL */
  int32_t ldr;
  int32_t reserved;
  void* y;              /* fwd: y;  dgrad: dx */
  const float* scale;   /* fwd only, may be NULL: y = act(scale[n]*conv + bias[n]) (folded inference BatchNorm, see _affine) */
} UssegConvJob;
int usseg_conv2d_fwd_multi(int32_t njobs, const UssegConvJob* jobs, usseg_stream_t stream);
int usseg_conv2d_dgrad_multi(int32_t njobs, const UssegConvJob* jobs, usseg_stream_t stream);

/* ---- transposed convolution: Conv2DTranspose 3x3 s2 (Decoder.py:57,120) and 4x4 s2 (TBI_ResNest.py:124,210),
 * padding 'same' (k=3: out[2i+k] += x[i] w[k], last row/col cropped; k=4: out[2i+k-1]).  Four parity-class
 * implicit GEMMs, no zero insertion. */
int usseg_tconv2d_fwd(const UssegConvDesc* d, const void* x, const void* wp_fwd, const float* bias, void* y,
                      usseg_stream_t stream);
int usseg_tconv2d_dgrad(const UssegConvDesc* d, const void* dy, const void* wp_dgrad, const void* residual,
                        int32_t ldr, void* dx, usseg_stream_t stream);
int usseg_tconv2d_wgrad(const UssegConvDesc* d, const void* x, const void* dy, float* dw_scratch,
                        usseg_stream_t stream);

/* ---- operand packing -------------------------------------------------------------------------
 * dst[(n_off+n)*Kw + tap*tap_stride + k_off + k] = bf16(src[tap*sT + n*sN + k*sK]),  n<Nn, k<Kk, tap<T.
 * Inverse (gradient gather): dst_f32[tap*sT + n*sN + k*sK] (+)= scale * scratch[tap][k_off+k][n_off+n],
 * scratch being the [T][Mrows][Ncols] fp32 output of a *_wgrad call. */
int usseg_pack_weight(const float* src, int64_t sT, int64_t sN, int64_t sK, int32_t T, int32_t Nn, int32_t Kk,
                      void* dst, int32_t Kw, int32_t tap_stride, int32_t n_off, int32_t k_off,
                      usseg_stream_t stream);
/* All packs of a model in one launch: `jobs_dev` is a DEVICE array of njobs descriptors (same fields as above). */
typedef struct UssegPackJob {
  const float* src;
  void* dst;
  int64_t sT, sN, sK;
  int32_t T, Nn, Kk, Kw, tap_stride, n_off, k_off, reserved;
  const float* nscale; /* optional fp32 multiplier per row n (NULL = none): an inference BatchNormalization folded into the
                          FORWARD operand, W'[n][..] = W[n][..] * gamma[n]*rsqrt(var[n]+eps) (the conv then runs its plain
                          bias + activation epilogue with bias' = beta - mean*scale + scale*bias) */
} UssegPackJob;
int usseg_pack_weights_batched(const UssegPackJob* jobs_dev, int32_t njobs, usseg_stream_t stream);
/* Same packs, flat grid: tilemap = nblocks triples (job index, first 32x32 tile of that job, tile count) that cover every tile of
 * every job exactly once (tiles of a job are numbered tap-major, then 32-row blocks of n, then 32-column blocks of k). */
int usseg_pack_weights_flat(const UssegPackJob* jobs_dev, const int32_t* tilemap_dev, int32_t nblocks, usseg_stream_t stream);
int usseg_unpack_wgrad(const float* scratch, int32_t Mrows, int32_t Ncols, int32_t T, int32_t Nn, int32_t Kk,
                       int32_t n_off, int32_t k_off, float* dst, int64_t sT, int64_t sN, int64_t sK,
                       float scale, int32_t accumulate, usseg_stream_t stream);
/* Several unpacks in one launch (device-resident job table, static pointers): the per-branch blocks of a grouped gradient. */
typedef struct UssegUnpackJob {
  const float* scratch;
  float* dst;
  int64_t sT, sN, sK;
  int32_t Mrows, Ncols, T, Nn, Kk, n_off, k_off, accumulate;
  float scale;
  int32_t reserved;
} UssegUnpackJob;
int usseg_unpack_wgrad_batched(const UssegUnpackJob* jobs_dev, int32_t njobs, int32_t max_elems, usseg_stream_t stream);

/* ---- normalisation + activation (bf16 rows of C channels, G groups of Cg = C/G) ---------------
 * mode 0: LayerNormalization(axis=-1, eps) per pixel and per group (ResNest.py:86,125,132,164; Decoder.py:112)
 * mode 1: BatchNormalization in inference mode = per-channel affine with moving statistics
 *         (ResNest.py:19,23; Decoder.py:32-35,51-54; TBI_ResNest.py:90,144,164,169,190,213)
 * followed by act (LeakyReLU / ReLU / ELU / none).  mult scales the activated output (Arch A dropout = 2*mask). */
typedef struct UssegNormDesc {
  int64_t M;          /* pixels */
  int32_t C, Cphys;   /* logical channels, physical channels (multiple of 8; pad channels written as 0) */
  int32_t ldx, ldy;   /* channel strides of input and output */
  int32_t G;          /* groups (LN only), C % G == 0 */
  int32_t mode;       /* 0 LN, 1 affine, 2 (usseg_norm_act_bwd only) affine whose input x is the ACTIVATED output of usseg_conv2d_fwd_affine */
  float eps;
  int32_t act;
  float alpha;
  int32_t lddx;       /* usseg_norm_act_bwd: channel stride of dx (0 = ldx) */
} UssegNormDesc;
/* mask (may be NULL): bf16 tensor [M][ldm] multiplied into the ACTIVATED output (dropout: 0 or 1/keep; since the mask is
 * non-negative relu(bn(x)*mask) == relu(bn(x))*mask, TBI_ResNest.py:213-218); the backward applies it to dy.
 * gamma / beta / mean / var: fp32, 16-byte aligned, readable up to Cphys floats (only the first C are used). */
int usseg_norm_act_fwd(const UssegNormDesc* d, const void* x, const float* gamma, const float* beta,
                       const float* mean, const float* var, const void* mask, int32_t ldm, void* y, usseg_stream_t stream);
/* tf.nn.dropout mask (TBI_ResNest.py:216): mask = keep ? 1/(1-rate) : 0 from a counter-based hash of (seed, index). */
int usseg_dropout_mask(void* mask, int64_t M, int32_t C, int32_t ld, uint64_t seed, float rate, usseg_stream_t stream);
/* Same, with a device-resident step counter mixed into the seed (the optimiser's step variable): a captured HIP graph of the
 * training step then draws a fresh mask at every replay. */
int usseg_dropout_mask_step(void* mask, int64_t M, int32_t C, int32_t ld, uint64_t seed, const int32_t* step_dev, float rate,
                            usseg_stream_t stream);
/* dx, and dgamma/dbeta accumulated.  x is the SAME pre-normalisation input as in fwd. */
int usseg_norm_act_bwd(const UssegNormDesc* d, const void* x, const void* dy, const float* gamma, const float* beta,
                       const float* mean, const float* var, const void* mask, int32_t ldm, void* dx, float* dgamma,
                       float* dbeta, float* dbias, float* ws, usseg_stream_t stream);
/* LayerNormalization backward (mode 0, G = 1) + the residual branch around a pre-norm transformer block in one pass
 * (VisionTransformer.py:137-146, SwinTransformer.py:224-257: x = x + f(norm(x))): dx = bf16(bf16(LN backward) + dres) - the bits of
 * usseg_norm_act_bwd followed by usseg_copy_channels(accumulate).  dres: bf16 [M][lddres].  dbias (may be NULL) += the column sums
 * of the STORED dx: the bias gradient of the Dense layer whose output (plus a shortcut) this gradient belongs to (attention
 * projection / the previous block's fc2) - no usseg_colsum pass for it. */
int usseg_norm_act_bwd_res(const UssegNormDesc* d, const void* x, const void* dy, const float* gamma, const float* beta,
                           const void* dres, int32_t lddres, void* dx, float* dgamma, float* dbeta, float* dbias, float* ws,
                           usseg_stream_t stream);
/* The split attention of a residual_S stage (ResNest.py:171-199) folded into the norms on either side of it:
 *  - usseg_norm_act_fwd_gap: the norm + activation that produces the cardinal output ALSO writes, per image and workgroup, the
 *    partial sums of its (bf16) output over the pixels: gap_rows [B][nb][Cphys] - the global average pool of ResNest.py:179
 *    without a pass over the tensor (usseg_splitattn_mlp_fwd adds the nb rows);
 *  - usseg_norm_act_bwd_sa: the norm backward whose incoming gradient is the re-weighting's backward
 *    dy = sa_mult * sa_s[b][c] * dout + sa_dg[b][c]  (identical radix branches: R == 1 in the descriptor), formed in registers
 *    (replaces usseg_splitattn_apply_bwd_dy + the tensor it wrote).  d->M must be B * pixels-per-image. */
int usseg_norm_act_fwd_gap(const UssegNormDesc* d, const void* x, const float* gamma, const float* beta, const float* mean,
                           const float* var, void* y, int32_t B, int32_t nb, float* gap_rows, usseg_stream_t stream);
int usseg_norm_act_bwd_sa(const UssegNormDesc* d, const void* x, const void* dout, const float* gamma, const float* beta,
                          const float* mean, const float* var, int32_t B, const float* sa_s, const float* sa_dg, float sa_mult,
                          void* dx, float* dgamma, float* dbeta, float* dbias, float* ws, usseg_stream_t stream);
/* Two INDEPENDENT LayerNormalization + LeakyReLU backward passes as ONE launch: A plain (a residual_S stage's shortcut norm, ResNest.py:100-101),
 * B with the split-attention re-weighting's backward folded in as in usseg_norm_act_bwd_sa (conv2_bn, ResNest.py:143-144 behind :194-197).
 * Same results as the two calls (bitwise: the same tile kernel runs both roles).  USSEG_ERR_UNSUPPORTED, and nothing launched, when the pair
 * has no instantiation (channel widths other than a stage of ResNest.py, tensors below 32k pixels): the caller then makes the two calls. */
int usseg_norm_act_bwd_pair(const UssegNormDesc* da, const void* xa, const void* dya, const float* gamma_a, const float* beta_a, void* dxa,
                            float* dgamma_a, float* dbeta_a, float* dbias_a, const UssegNormDesc* db, const void* xb, const void* doutb,
                            const float* gamma_b, const float* beta_b, int32_t B, const float* sa_s, const float* sa_dg, float sa_mult, void* dxb,
                            float* dgamma_b, float* dbeta_b, float* dbias_b, float* ws, usseg_stream_t stream);
/* The encoder stem as ONE launch (ResNest.py:39-47): Conv2D(1->16) + LeakyReLU -> Conv2D(16->32) + inference BatchNormalization (its scale
 * folded into the packed operand w2, its shift = b2) + LeakyReLU -> Conv2D(32->32) -> BatchNormalization + LeakyReLU -> AveragePooling2D(2,2).
 * x [B,H,W,8 physical channels]; w1 [16][72], w2 [32][144], w3 [32][288]: the packed forward operands of usseg_conv2d_fwd; b1 [16], b2 [32],
 * b3 [32], gamma / beta / mean / var [32] fp32.  Outputs, all kept for the backward pass: y1 [B,H,W,16] and t1 [B,H,W,32] (activated),
 * c2 [B,H,W,32] (pre-norm convtmp_2 output), pooled [B,H/2,W/2,32].  A workgroup owns a 16x16 tile and recomputes the two inner convs on
 * its halo, so no intermediate is re-read from HBM; rounding points are those of the four unfused launches. */
int usseg_stem_fwd(int32_t B, int32_t H, int32_t W, const void* x, int32_t ldx, const void* w1, const float* b1, const void* w2, const float* b2,
                   const void* w3, const float* b3, const float* gamma, const float* beta, const float* mean, const float* var, float eps,
                   float alpha, void* y1, void* t1, void* c2, void* pooled, usseg_stream_t stream);
/* A stem conv's 3x3 backward-data pass + the backward of the layer in front of it, in ONE launch (GradientTape through ResNest.py:41-44 /
 * :39-41): dy [B,H,W,32] -> dx_in = conv3x3_backward_data(dy, wd) (wd = the packed operand of usseg_conv2d_dgrad, [Co][9*32]) -> then
 *   mode 2 (Co = 32): the folded inference BatchNormalization + LeakyReLU backward from the ACTIVATED tensor yact the forward stored
 *                     (= usseg_norm_act_bwd mode 2: dx, dgamma += , dbeta +=, dbias += sum dx);
 *   mode 0 (Co = 16): LeakyReLU backward from the activated tensor + column sums (= usseg_act_bwd_colsum: dx, dbias +=).
 * The intermediate gradient never goes to HBM (it stays fp32 in registers: the unfused pair rounds it to bf16 in between).
 * USSEG_ERR_UNSUPPORTED - nothing launched - for any other (Co, mode).  ws: usseg_reduce_ws_floats() floats. */
int usseg_conv3_dgrad_actbwd(int32_t B, int32_t H, int32_t W, const void* dy, int32_t lddy, const void* wd, int32_t Co, const void* yact, int32_t ldy,
                             int32_t mode, const float* gamma, const float* beta, const float* var, float eps, float alpha, void* dx,
                             int32_t lddx, float* dgamma, float* dbeta, float* dbias, float* ws, usseg_stream_t stream);
/* One residual_S stage's cardinal group AND shortcut as ONE launch (SURVEY.md section 2.2 "K3"; ResNest.py:136-147 per path - the `kpaths`
 * paths share the input, :99-101 shortcut): replaces, in the implicit TensorFlow graph of the reference, Conv2D(1x1) -> LayerNormalization ->
 * LeakyReLU -> Conv2D(3x3) -> LayerNormalization -> LeakyReLU (+ the reduce_mean of :179) per path and Conv2D(1x1) -> LayerNormalization ->
 * LeakyReLU of the shortcut.  One workgroup owns an 8x8-pixel tile (+1 halo) of one image in LDS.
 *   x [B,H,W,Cin] bf16; w1 [roundup(Up,16)][Cin], w2 [roundup(Vp,16)][9*Up] (block diagonal over the paths), wsc [Oc][Cin]: the packed
 *   forward operands of usseg_conv2d_fwd; b1/g1/be1 [Up]/[P*cv11], b2/g2/be2 [Vp]/[P*cvkk], bsc/gsc/besc [Oc] fp32 (per-path vectors back to back);
 *   outputs (all kept for the backward pass): u_raw, u [B,H,W,Up]; v_raw, y [B,H,W,Vp]; sc_raw, sc [B,H,W,Oc];
 *   gap_rows [B][ceil(H/8)*ceil(W/8)][Vp] fp32: per-tile sums of y over its pixels (consumed by usseg_splitattn_mlp_fwd).
 * Values are rounded to bf16 where the unfused launches round them.  usseg_cardinal_supported: 1 if a fused kernel exists for the
 * channel configuration (the four stages of ResNest.py with radix 3 / kpaths 3), else the caller runs the unfused launches. */
typedef struct UssegCardinalDesc {
  int32_t B, H, W;
  int32_t Cin;              /* physical input channels */
  int32_t P, cv11, cvkk;    /* paths; logical channels per path after the 1x1 / the 3x3 (ResNest.py:120-121) */
  int32_t Up, Vp, Oc;       /* physical widths roundup(P*cv11, 8), roundup(P*cvkk, 8); shortcut / stage output channels */
  int32_t ldx, ldu, ldv, ldsc;
  float eps, alpha;         /* LayerNormalization epsilon, LeakyReLU slope */
} UssegCardinalDesc;
int32_t usseg_cardinal_supported(const UssegCardinalDesc* d);
int usseg_cardinal_fwd(const UssegCardinalDesc* d, const void* x, const void* w1, const float* b1, const float* g1, const float* be1,
                       const void* w2, const float* b2, const float* g2, const float* be2, const void* wsc, const float* bsc,
                       const float* gsc, const float* besc, void* u_raw, void* u, void* v_raw, void* y, float* gap_rows, void* sc_raw,
                       void* sc, usseg_stream_t stream);
/* The backward pass of the same stage half as ONE launch (K3 backward; GradientTape through ResNest.py:136-147 and :100-101): replaces the
 * split-attention re-weighting's backward + LayerNormalization/LeakyReLU backward of conv2_bn, the grouped 3x3's backward-data pass, the
 * LayerNormalization/LeakyReLU backward of conv1_bn, and the shortcut norm's backward.
 *   dout [B,H,W,Vp] (stride ldo): gradient w.r.t. the split-attention output (concats_1); sa_s / sa_dg [B][P*cvkk] fp32 from
 *   usseg_splitattn_bwd_fused: the gradient w.r.t. y is sa_mult*sa_s[b][c]*dout + sa_dg[b][c], formed in registers;
 *   dsc [B,H,W,Oc] (stride lddsc): gradient w.r.t. the activated shortcut; v_raw, u_raw, sc_raw: saved by usseg_cardinal_fwd;
 *   w2d [roundup(Up,16)][9*Vp]: the packed backward-data operand of the grouped 3x3 (block diagonal, as usseg_conv2d_dgrad takes it);
 *   outputs: dv [B,H,W,Vp] (stride d->ldv; gradient w.r.t. v_raw - the grouped 3x3's weight gradient reads it) and
 *   dcat [B,H,W,ldc] = [du_raw (Up) | dsc_raw (Oc)]: the gradients w.r.t. both 1x1 outputs, the operand of ONE backward-data GEMM and ONE
 *   weight-gradient launch (both 1x1 convs read the stage input);
 *   dg2/dbe2/db2, dg1/dbe1/db1, dgsc/dbesc/dbsc: gamma / beta gradients of the three norms and the bias gradients of the convs in front of
 *   them, ACCUMULATED (one partial row per workgroup + an ordered finishing reduction: bitwise reproducible);
 *   ws: usseg_reduce_ws_floats() floats.  A workgroup owns an 8x8 tile and recomputes the first norm's backward on its one-pixel halo. */
int usseg_cardinal_bwd(const UssegCardinalDesc* d, const void* dout, int32_t ldo, const void* dsc, int32_t lddsc, const void* v_raw,
                       const void* u_raw, const void* sc_raw, const void* w2d, const float* g2, const float* be2, const float* g1,
                       const float* be1, const float* gsc, const float* besc, const float* sa_s, const float* sa_dg, float sa_mult, void* dv,
                       void* dcat, int32_t ldc, float* dg2, float* dbe2, float* db2, float* dg1, float* dbe1, float* db1, float* dgsc,
                       float* dbesc, float* dbsc, float* ws, usseg_stream_t stream);
/* Inference BatchNormalization + activation + AveragePooling2D(2,2) in one pass - the stem's convtmp_2bn -> LeakyReLU ->
 * conv1_pool (ResNest.py:45-47) and conv2_1_2bn -> ELU -> pool_1 (TBI_ResNest.py:90-92): the activated full-resolution tensor
 * feeds the pool only, so it is never written.  x [B,H,W,Cphys] pre-norm; y / dy [B,H/2,W/2,Cphys]; dx [B,H,W,Cphys] is the
 * gradient w.r.t. x; dgamma / dbeta / dbias (= sum of dx: the producing conv's bias gradient, may be NULL) accumulate.
 * Bit-identical to usseg_norm_act_fwd + usseg_avgpool2_fwd (resp. avgpool2_bwd + norm_act_bwd). */
int usseg_bn_act_pool_fwd(const void* x, int32_t B, int32_t H, int32_t W, int32_t C, int32_t Cphys, int32_t ldx, const float* gamma,
                          const float* beta, const float* mean, const float* var, float eps, int32_t act, float alpha, void* y,
                          int32_t ldy, usseg_stream_t stream);
int usseg_bn_act_pool_bwd(const void* x, const void* dy, int32_t B, int32_t H, int32_t W, int32_t C, int32_t Cphys, int32_t ldx,
                          int32_t lddy, const float* gamma, const float* beta, const float* mean, const float* var, float eps,
                          int32_t act, float alpha, void* dx, int32_t lddx, float* dgamma, float* dbeta, float* dbias, float* ws,
                          usseg_stream_t stream);
/* Every per-channel reduction (norm backward, colsum, channel_stats, split-attention reductions) writes one partial
 * row per workgroup into the caller's fp32 workspace `ws` (at least usseg_reduce_ws_floats() floats) and a finishing
 * kernel ADDS the column sums to the destination: no atomics, bitwise reproducible.
 * usseg_norm_act_bwd: dgamma/dbeta += ...; dbias (may be NULL) += sum over pixels of dx, i.e. the bias gradient of
 * the convolution that produced x. */
int64_t usseg_reduce_ws_floats(void);
/* BatchNormalization training mode statistics: sum[c] += sum_m x, sumsq[c] += sum_m x^2. */
int usseg_channel_stats(const void* x, int64_t M, int32_t C, int32_t ldx, float* sum, float* sumsq, float* ws,
                        usseg_stream_t stream);

/* BatchNormalization TRAINING mode (Keras semantics: batch mean, biased variance, moving statistics *= momentum).
 * forward : usseg_channel_stats -> usseg_bn_finalize_stats -> usseg_norm_act_fwd(mode 1, mean/var = batch statistics);
 * backward: usseg_norm_act_bwd(mode 1) with ZEROED scratch tg/tb as dgamma/dbeta, then usseg_bn_train_bwd_fix
 *           (dx -= gamma*rstd*(tb + xhat*tg)/M), then add tg/tb to the real gradients. */
int usseg_bn_finalize_stats(const float* sum, const float* sumsq, int64_t M, int32_t C, float momentum, float* mean, float* var,
                            float* moving_mean, float* moving_var, usseg_stream_t stream);
int usseg_bn_train_bwd_fix(const void* x, void* dx, int64_t M, int32_t C, int32_t Cphys, int32_t ldx, int32_t lddx, const float* gamma,
                           const float* mean, const float* var, float eps, const float* tg, const float* tb, usseg_stream_t stream);

/* ---- plain activation (conv1 + LeakyReLU, ResNest.py:39-40; TBI_ResNest.py:83-87) ------------- */
int usseg_act_fwd(const void* x, int64_t M, int32_t C, int32_t ldx, int32_t ldy, int32_t act, float alpha, void* y,
                  usseg_stream_t stream);
int usseg_act_bwd(const void* x, const void* dy, int64_t M, int32_t C, int32_t ldx, int32_t lddy, int32_t lddx,
                  int32_t act, float alpha, void* dx, usseg_stream_t stream);
/* act_bwd that also accumulates db[c] += sum_m dx[m][c] (C logical channels): the bias gradient of the conv whose fused
 * activation it undoes (ResNest.py:39-40), in the same pass.  ws: usseg_reduce_ws_floats() floats. */
int usseg_act_bwd_colsum(const void* x, const void* dy, int64_t M, int32_t C, int32_t ldx, int32_t lddy, int32_t lddx, int32_t act,
                         float alpha, void* dx, float* db, float* ws, usseg_stream_t stream);

/* ---- AveragePooling2D(2,2) (ResNest.py:25-28,47-53; TBI_ResNest.py:92-107) -------------------- */
int usseg_avgpool2_fwd(const void* x, int32_t B, int32_t H, int32_t W, int32_t C, int32_t ldx, int32_t ldy, void* y,
                       usseg_stream_t stream);
/* dx = 0.25 * dy[h/2,w/2] (+ add, bf16 stride ldadd, may be NULL) */
int usseg_avgpool2_bwd(const void* dy, int32_t B, int32_t H, int32_t W, int32_t C, int32_t lddy, int32_t lddx,
                       const void* add, int32_t ldadd, void* dx, usseg_stream_t stream);
/* avgpool2_bwd that also accumulates db[c] += sum over the pixels of dx: dx is the gradient w.r.t. the output of the conv in front of
 * the pool (a residual_S stage's concats_2, ResNest.py:98,49-54), so this is that conv's bias gradient.  C <= 512. */
int usseg_avgpool2_bwd_colsum(const void* dy, int32_t B, int32_t H, int32_t W, int32_t C, int32_t lddy, int32_t lddx, const void* add,
                              int32_t ldadd, void* dx, float* db, float* ws, usseg_stream_t stream);

/* ---- channel-slice copy / accumulate: tf.concat(axis=3) and the raw reshape re-injection
 * (ResNest.py:96; Decoder.py:66,75,87,140-141; TBI_ResNest.py:110-122,139) ---------------------- */
int usseg_copy_channels(const void* src, int64_t M, int32_t C, int32_t lds, void* dst, int32_t ldd, int32_t accumulate,
                        usseg_stream_t stream);
/* fp32/fp64 NHWC host-layout input -> bf16 with the channel count padded to Cphys (ResNest.py:39 input cast) */
int usseg_cast_input(const void* src, int32_t src_is_f64, int64_t M, int32_t C, void* dst, int32_t Cphys,
                     usseg_stream_t stream);
int usseg_cast_bf16_to_f32(const void* src, int64_t M, int32_t C, int32_t lds, float* dst, usseg_stream_t stream);

/* ---- split attention (ResNest.py:171-199; TBI_ResNest.py:175-207) ----------------------------
 * Tensors: y [B,HW,P*R*Cg] bf16 (P cardinal paths, R distinct radix branches, Cg=cvkk channels each).
 * gap:    g[b][c] = sum_hw y[b,hw,c]                                  (fp32, zeroed by caller)
 * mlp:    per (b,path): a = act(norm(dense1(mult/HW * sum_r g))) ; z_r = dense2_r(a) ; s_r = softmax_c(z_r)
 * apply:  out[b,hw,p,c] = mult * sum_r y[b,hw,p,r,c] * s[b,p,r,c]
 * In the reference's Arch B the R branches share weights and are identical: R=1, mult=radix. */
typedef struct UssegSplitAttnDesc {
  int32_t B, HW;
  int32_t P, R, Cg, Hd;     /* paths, distinct radix branches, channels per branch, hidden = Cg/2 */
  int32_t ldy, ldo;         /* channel strides of y and out */
  int32_t Cy_phys, Co_phys; /* physical widths of y and out (pads written as zero in out) */
  float mult;               /* radix multiplicity of identical branches (Arch B) or 1 (Arch A) */
  int32_t norm_mode;        /* 0 LN (Arch B), 1 affine/BN-inference (Arch A) */
  float eps;
  int32_t act;              /* LeakyReLU (B) / ELU (A) */
  float alpha;
  int32_t use_sigmoid;      /* radix == 1 (ResNest.py:189-190) */
} UssegSplitAttnDesc;
/* parameter block for the tiny MLP, all fp32, per path p: w1[p][Cg][Hd], b1[p][Hd], gamma/beta/mean/var[p][Hd],
 * w2[p][R][Hd][Cg], b2[p][R][Cg] */
typedef struct UssegSplitAttnParams {
  const float *w1, *b1, *gamma, *beta, *mean, *var, *w2, *b2;
} UssegSplitAttnParams;
typedef struct UssegSplitAttnGrads {
  float *w1, *b1, *gamma, *beta, *w2, *b2;
} UssegSplitAttnGrads;
/* g[b][c] = sum_hw y[b,hw,c] (g is OVERWRITTEN: no pre-zeroing needed) */
int usseg_splitattn_gap(const UssegSplitAttnDesc* d, const void* y, float* g, float* ws, usseg_stream_t stream);
/* ws: fp32 workspace of usseg_splitattn_ws_floats(d) floats holding the saved MLP intermediates */
int64_t usseg_splitattn_ws_floats(const UssegSplitAttnDesc* d);
/* g = pooled sums as [B][g_rows][g_stride] partial rows which the kernel adds up: (1, Cy) for the vector usseg_splitattn_gap
 * wrote, or the rows usseg_norm_act_fwd_gap produced (g_rows = its nb, g_stride = Cphys) - no pooling pass, no finishing launch */
int usseg_splitattn_mlp_fwd(const UssegSplitAttnDesc* d, const float* g, int32_t g_rows, int32_t g_stride,
                            const UssegSplitAttnParams* p, float* s, float* ws, usseg_stream_t stream);
int usseg_splitattn_apply_fwd(const UssegSplitAttnDesc* d, const void* y, const float* s, void* out,
                              usseg_stream_t stream);
/* backward: (1) ds[b][p][r][c] = sum_hw mult*y*dout (ds is OVERWRITTEN) (2) MLP backward -> dg, param grads
 * (3) dy = mult*s*dout + dg*mult/HW */
int usseg_splitattn_apply_bwd_reduce(const UssegSplitAttnDesc* d, const void* y, const void* dout, int32_t lddo,
                                     float* ds, float* ws, usseg_stream_t stream);
/* grad_ws: usseg_splitattn_mlp_bwd_ws_floats(d) floats - one row of parameter-gradient partials per (path, image); the rows
 * are added into the variables in image order by a finishing reduction (no float atomics: bitwise reproducible). */
int64_t usseg_splitattn_mlp_bwd_ws_floats(const UssegSplitAttnDesc* d);
int usseg_splitattn_mlp_bwd(const UssegSplitAttnDesc* d, const float* g, int32_t g_rows, int32_t g_stride,
                            const UssegSplitAttnParams* p, const float* s, const float* ws, const float* ds, float* dg,
                            const UssegSplitAttnGrads* grads, float* grad_ws, usseg_stream_t stream);
/* usseg_splitattn_apply_bwd_reduce + usseg_splitattn_mlp_bwd as one call with no finishing launch between them: the MLP kernel adds the
 * reduce kernel's per-workgroup rows (in `reduce_ws`, usseg_reduce_ws_floats()) itself.  Outputs: dg [B][P*R*Cg] and the parameter gradients. */
int usseg_splitattn_bwd_fused(const UssegSplitAttnDesc* d, const void* y, const void* dout, int32_t lddo, const float* g, int32_t g_rows,
                              int32_t g_stride, const UssegSplitAttnParams* p, const float* s, const float* ws, float* dg,
                              const UssegSplitAttnGrads* grads, float* reduce_ws, float* grad_ws, usseg_stream_t stream);
int usseg_splitattn_apply_bwd_dy(const UssegSplitAttnDesc* d, const void* dout, int32_t lddo, const float* s,
                                 const float* dg, void* dy, int32_t lddy, usseg_stream_t stream);

/* ---- head softmax + loss ---------------------------------------------------------------------
 * logits fp32 [M][ldl] (C classes) -> probs fp32 [M][C] (Decoder.py:121; TBI_ResNest.py:125).
 * loss_kind 0: CategoricalCrossentropy(label_smoothing, reduction NONE) summed and divided by the GLOBAL batch
 *              (VisionTransformer.py:205,225-227); loss[0] receives the scalar.
 * loss_kind 1: my_loss_cat (TBI_ResNest.py:234-248) with precomputed per-pixel-per-class scale[HW][C];
 *              loss_map [HW] is overwritten.
 * dlogits (bf16 [M][lddl], may be NULL) = d(sum loss)/d logits. */
typedef struct UssegLossDesc {
  int64_t M;            /* B*H*W pixels */
  int32_t HW;           /* pixels per image */
  int32_t C, ldl, lddl; /* classes, logits stride (floats), dlogits stride (bf16 elements) */
  int32_t loss_kind;
  float label_smoothing;
  float clip_eps;       /* 1e-7 */
  float inv_global_batch;
  int32_t quad_w;       /* 0: logits/dlogits are [M][ldl]; W (full-resolution width): they are in the space-to-depth layout
                           [B][H/2][W/2][16] that the 2x2-tap form of the stride-2 head produces, pixel (y,x) in slot
                           4*((y&1)*2+(x&1)) (ldl = lddl = 16, C <= 4).  probs and y_true are always [M][C]. */
} UssegLossDesc;
/* Reproducible scalar accumulators (loss of loss_kind 0, usseg_sumsq): the pointer names USSEG_ACC_FLOATS floats -
 * [0] the result (OVERWRITTEN by every call), [1] a ticket counter (zero between calls), [2..] one partial per workgroup; the
 * last workgroup to arrive adds the partials in workgroup order, so no float atomics and the same bits on every run.  Zero the
 * whole buffer once at allocation; no per-call zeroing.  The loss map of loss_kind 1 is [HW] floats, each pixel owned by one
 * thread and OVERWRITTEN. */
#define USSEG_ACC_FLOATS 2050
int usseg_softmax_loss_fwd_bwd(const UssegLossDesc* d, const float* logits, const float* y_true, const float* scale,
                               float* probs, float* loss, void* dlogits, usseg_stream_t stream);
/* The quad-form head conv (bias included), its softmax and the loss of loss_kind 0 in ONE launch: replaces usseg_quad_bias_expand +
 * usseg_conv2d_fwd (fp32 quad logits) + usseg_softmax_loss_fwd_bwd (quad_w != 0) - the reference's
 * Conv2DTranspose(num_classes, 3, strides 2) -> Softmax -> CategoricalCrossentropy (Decoder.py:119-121,142; VisionTransformer.py:205,225-229).
 *   x [B,h,w,Cin_phys] bf16 (stride ldx); wq [16][9*Cin_phys] bf16: the quad operand (rows = parity class*4 + class, K = stencil tap*Cin_phys + channel);
 *   bias [C]; y_true [B,2h,2w,C] fp32 or NULL (probabilities only); probs [B,2h,2w,C] fp32; loss: USSEG_ACC_FLOATS floats ([0] is overwritten);
 *   dlogits [B,h,w,16] bf16 in the quad layout or NULL.  USSEG_ERR_UNSUPPORTED (nothing launched) when ceil(h/16)*ceil(w/16)*B exceeds the
 *   slots of the ordered sum, or Cin_phys is neither 72 (Decoder.py: the last block's 16 channels + the 56 re-injected ones) nor 16: the caller
 *   then runs the three calls. */
int usseg_head_quad_softmax_loss(const void* x, int32_t B, int32_t h, int32_t w, int32_t Cin_phys, int32_t ldx, const void* wq, const float* bias, int32_t C,
                                 const float* y_true, float* probs, float* loss, void* dlogits, float label_smoothing, float clip_eps,
                                 float inv_global_batch, usseg_stream_t stream);
/* The same two losses evaluated on PROBABILITIES, as the reference's public methods take them:
 * loss_kind 0 = VisionTransformer.compute_loss(y_true, y_pred) (VisionTransformer.py:225-227: Keras normalises the
 * probabilities by their class sum, clips to [clip_eps, 1-clip_eps], -sum_c y_smoothed*log p, summed / global batch into *loss);
 * loss_kind 1 = ResNest.my_loss_cat(y_true, y_pred) (TBI_ResNest.py:234-248: loss map [HW] -= y*log(p+1e-7)*scale[hw][c]).
 * probs, y_true fp32 [M][C]; only M, HW, C, loss_kind, label_smoothing, clip_eps, inv_global_batch of the descriptor are read.
 * loss is overwritten (loss_kind 0: a USSEG_ACC_FLOATS accumulator, [0] = the scalar). */
/* Accuracy metric of a step (TBI_ResNest.py:48-51; VisionTransformer.py:256-262): acc[0] = fraction of the M pixels whose
 * argmax(probs) equals argmax(y_true) (first maximum, as tf.argmax); probs, y_true fp32 [M][C].  acc is a USSEG_ACC_FLOATS
 * accumulator (zeroed once by the caller), [0] OVERWRITTEN.  One pass instead of the reference's four framework reductions. */
int usseg_accuracy(const float* probs, const float* y_true, int64_t M, int32_t C, float* acc, usseg_stream_t stream);
int usseg_loss_from_probs(const UssegLossDesc* d, const float* probs, const float* y_true, const float* scale, float* loss,
                          usseg_stream_t stream);
/* The <= 4-class head Conv2DTranspose(k x k, stride 2, 'same') (Decoder.py:120 k=3; TBI_ResNest.py:124 k=4) in "quad" form: a
 * stride-1 convolution at the INPUT resolution whose 16 output channels are (output parity class)*4 + n, run by
 * usseg_conv2d_fwd/dgrad/wgrad as a 3x3 conv (each parity uses <= 2x2 of the stencil taps, the rest are zero).  With
 * pad = (k==4): kernel tap kh feeds output parity a = (kh+pad)&1 from source offset di = (a+pad-kh)/2.  These helpers move
 * the bias and the gradients between the Keras variables and that form: bias16[cls*4+n] = bias[n];
 * dbias[n] += sum_cls d16[cls*4+n] (head, Np = 4);  grad[kh][kw][n][c] += dq[(di+1)*3+(dj+1)][c][(a*2+b)*Np+n], dq = [9][Cin_phys][4*Np]. */
int usseg_quad_bias_expand(const float* bias, int32_t C, int32_t Np, float* out /* [4*Np] */, usseg_stream_t stream);
int usseg_quad_bias_fold(const float* d16, int32_t C, float* dbias, usseg_stream_t stream);
int usseg_tconv_quad_unpack(const float* dq, int32_t Cin_phys, int32_t Cin, int32_t Cout, int32_t Np, int32_t ksize, float* grad,
                            usseg_stream_t stream);
/* usseg_tconv_quad_unpack + usseg_quad_bias_fold (with d16 = [4][Np] column sums) as ONE launch: the tail of the quad-form head's backward. */
int usseg_quad_head_fold(const float* dq, int32_t Cin_phys, int32_t Cin, int32_t Cout, int32_t Np, int32_t ksize, float* grad, const float* d16,
                         float* dbias, usseg_stream_t stream);
/* General quad-form transposed conv (any channel count; Np = physical output channels per parity class, the 16 of the head = 4*4):
 * y4 / dy4 are [B,H,W,4*Np] space-to-depth tensors of the [B,2H,2W,Np] output (usseg_space_to_depth2 converts either way);
 * the descriptor is that of the 3x3 stride-1 conv (ksize 3, Cout = 4*Np).  Each parity class only uses <= 2x2 of the nine stencil
 * taps: the conv kernels skip the others per channel class (tap masks), so no multiply is wasted on the structural zeros.
 * Replaces tf.keras.layers.Conv2DTranspose(k=4, strides 2) of TBI_ResNest.py:124,210 (and k=3 of Decoder.py:120). */
int usseg_space_to_depth2(const void* full, int32_t B, int32_t H, int32_t W, int32_t C, int32_t ld_full, void* quad, int32_t Np,
                          int32_t ld_quad, int32_t to_quad, usseg_stream_t stream);
int usseg_tconv_quad_fwd(const UssegConvDesc* d, int32_t ksize, int32_t Np, const void* x, const void* wq_f, const float* bias_q, void* y4,
                         usseg_stream_t stream);
int usseg_tconv_quad_dgrad(const UssegConvDesc* d, int32_t ksize, int32_t Np, const void* dy4, const void* wq_d, void* dx, usseg_stream_t stream);
int usseg_tconv_quad_wgrad(const UssegConvDesc* d, int32_t ksize, int32_t Np, const void* x, const void* dy4, float* dq, float* ws,
                           int64_t ws_floats, usseg_stream_t stream);

/* my_loss_cat scale[hw][c] = 1/(sum_b y[b,hw,c] + 1)/(H*W)  (TBI_ResNest.py:240-241) */
int usseg_loss_cat_scale(const float* y_true, int32_t B, int32_t HW, int32_t C, float* scale, usseg_stream_t stream);

/* ---- host input pipeline on the device -----------------------------------------------------
 * label2vec (Dataset_2.py:6-20): label [M] fp32 -> soft class maps [M][num_classes] (3: c2 = clip(l-1,0,1) where l >= 1.05,
 * c1 = (l > 0.95) ? 1-c2 : 0, c0 = l <= 0.95; 2: (1-l, l)). */
int usseg_label2vec(const float* label, int64_t M, int32_t num_classes, float* out, usseg_stream_t stream);
/* dataAug (DataAugs.py:82-102) AS EXECUTED for a batch, one fused pass: imageReduc (zeroes the image where the label is 0,
 * DataAugs.py:76-78), up to 2 clip boxes (:27-38), shift (:6-24, last row/column stay zero) and noisy (:41-51, unit Gaussian
 * / 5000: from `noise` [B,H,W,C] if given, else a counter-based generator seeded per sample), then label2vec and the cast
 * to the model's input.  The draws are made on the host in the reference's order and passed per sample. */
typedef struct UssegAugSample {
  int32_t do_reduc, nclip;
  int32_t clip[2][4];        /* r, c, ra, ca per box */
  int32_t do_shift, shift_r, shift_c, shift_dir;
  int32_t do_noise, reserved;
  uint64_t seed;
} UssegAugSample;
typedef struct UssegAugDesc {
  int32_t B, H, W, C;        /* x: [B,H,W,C] fp32 or fp64, y: [B,H,W] fp32 */
  int32_t Cphys;             /* channels of the bf16 output (multiple of 8, zero padded) */
  int32_t num_classes;       /* for y_vec */
} UssegAugDesc;
/* outputs (any may be NULL except one of the x outputs): x_out_bf16 [B,H,W,Cphys], x_out_f32 [B,H,W,C], y_out [B,H,W],
 * y_vec [B,H,W,num_classes] */
int usseg_augment(const UssegAugDesc* d, const UssegAugSample* samples_dev, const void* x, int32_t x_is_f64, const float* y,
                  const float* noise, void* x_out_bf16, float* x_out_f32, float* y_out, float* y_vec, usseg_stream_t stream);

/* ---- Decoder.py:140-141: the hidden state [B][N][hidden] is re-injected at every decoder scale through a RAW row-major
 * reshape to [B][gh*s][gw*s][c0_i] and concatenated behind the block output.  bufs[i] = first element of scale i's channel
 * slice (pixel stride ld[i], c0[i] channels, c0 % 8 == 0), n <= 4 scales in one launch.
 * backward == 0: scatter `hidden` into the n slices; backward == 1: hidden = sum_i slice_i (fp32 sum, one rounding). */
int usseg_reinject_hidden(void* hidden, int64_t numel, int32_t n, void* const* bufs, const int32_t* c0, const int32_t* ld,
                          int32_t backward, usseg_stream_t stream);

/* ---- windowed-attention encoder (SwinTransformer.py: PatchEmbed :341-365, WindowAttention :60-141, SwinTransformerBlock :162-261,
 * PatchMerging :264-290, GlobalAveragePooling1D :451).  Tokens are NHWC bf16 tensors [B][H][W][C] (the reference's [B, H*W, C]).
 *
 * usseg_patchify: x fp32/fp64 [B][H][W][C] -> bf16 [B][H/p][W/p][Cp], channel (ph*p + pw)*C + c (zero pad up to Cp): the
 *   space-to-depth half of Conv2D(kernel = stride = p); the other half is a 1x1 conv whose kernel is the Keras [p][p][C][E] variable
 *   read as [p*p*C][E] (same memory).
 * usseg_patch_merge: merged[b][i][j][(a + 2*bb)*C + c] = full[b][2i+a][2j+bb][c] (:280-284); backward != 0 scatters merged -> full.
 * usseg_ln_wide_*: LayerNormalization over C <= 4096 channels (C % 8 == 0) of M tokens; backward accumulates dgamma / dbeta
 *   (ws: >= 2*C floats per workgroup, ws_floats total - more is faster up to 512 workgroups).
 * usseg_window_attn_*: qkv [B][H][W][3C] (q | k | v, each [heads][C/heads]) -> out [B][H][W][C]; window side ws in {2,4,8}, cyclic
 *   shift `shift` (0 <= shift < ws) with the -100 region mask of :192-217, relative-position bias table [(2ws-1)^2][heads] (:78-99),
 *   q scaled by (C/heads)^-0.5.  Backward writes dqkv (same layout as qkv) and ACCUMULATES the table gradient;
 *   ws_rows: usseg_window_attn_bwd_ws_floats() floats.
 * usseg_token_mean_*: out[b][c] = mean over the L tokens; backward dx[b][t][c] = dy[b][c] / L. */
int usseg_patchify(const void* x, int32_t x_is_f64, int32_t B, int32_t H, int32_t W, int32_t C, int32_t patch, void* out, int32_t Cp,
                   usseg_stream_t stream);
int usseg_patch_merge(void* full, int32_t B, int32_t H, int32_t W, int32_t C, int32_t ldf, void* merged, int32_t ldm, int32_t backward,
                      usseg_stream_t stream);
int usseg_ln_wide_fwd(const void* x, int64_t M, int32_t C, int32_t ldx, const float* gamma, const float* beta, float eps, void* y,
                      int32_t ldy, usseg_stream_t stream);
int usseg_ln_wide_bwd(const void* x, const void* dy, int64_t M, int32_t C, int32_t ldx, int32_t lddy, const float* gamma, float eps,
                      void* dx, int32_t lddx, float* dgamma, float* dbeta, float* ws, int64_t ws_floats, usseg_stream_t stream);
int usseg_window_attn_fwd(const void* qkv, int32_t ldq, const float* table, int32_t B, int32_t H, int32_t W, int32_t C, int32_t heads,
                          int32_t ws, int32_t shift, void* out, int32_t ldo, usseg_stream_t stream);
int64_t usseg_window_attn_bwd_ws_floats(int32_t B, int32_t H, int32_t W, int32_t heads, int32_t ws);
int usseg_window_attn_bwd(const void* qkv, int32_t ldq, const void* dout, int32_t ldo, const float* table, int32_t B, int32_t H, int32_t W,
                          int32_t C, int32_t heads, int32_t ws, int32_t shift, void* dqkv, float* dtable, float* ws_rows,
                          usseg_stream_t stream);
int usseg_token_mean_fwd(const void* x, int32_t B, int32_t L, int32_t C, int32_t ld, float* out, usseg_stream_t stream);
int usseg_token_mean_bwd(const float* dy, int32_t B, int32_t L, int32_t C, int32_t ld, void* dx, usseg_stream_t stream);

/* ---- bias gradient: db[c] += sum_pixels dy[m][c] --------------------------------------------- */
int usseg_colsum(const void* dy, int64_t M, int32_t C, int32_t ld, float* db, float* ws, usseg_stream_t stream);

/* ---- optimiser: tf.clip_by_global_norm(1.0) + Adam (VisionTransformer.py:204,244-245; TBI_ResNest.py:28,46)
 * sumsq: out[0] = sum g^2 over n floats (OVERWRITTEN).  `out` is a USSEG_ACC_FLOATS accumulator (zeroed by the caller once; see below),
 *        so the sum - and with it the clip factor of every update - is bitwise reproducible from run to run.
 * adam:  scale = clip_norm > 0 ? clip_norm / max(sqrt(*sumsq), clip_norm) : 1;  g' = g*scale*grad_scale;
 *        Keras Adam with bias-corrected lr_t (host-computed from the device step counter is avoided: the caller
 *        passes lr_t = lr*sqrt(1-b2^t)/(1-b1^t) through a device scalar so a captured graph can be replayed). */
int usseg_sumsq(const float* g, int64_t n, float* out, usseg_stream_t stream);
/* out[0] = sum of n floats through a USSEG_ACC_FLOATS accumulator (see above): the scalar of TBI_ResNest.py's [H,W] loss map
 * (my_loss_cat, :234-248) that the mirrored step reduces (MainParallel.py:131-134) - reproducible, no framework reduction. */
int usseg_sum_f32(const float* x, int64_t n, float* out, usseg_stream_t stream);
/* the same launch also advances the optimiser's device step counter (usseg_adam_advance) in its last workgroup */
int usseg_sumsq_advance(const float* g, int64_t n, float* out, int32_t* step, float* lr_t_dev, float lr, float beta1, float beta2,
                        usseg_stream_t stream);
int usseg_adam_clip_step(float* p, const float* g, float* m, float* v, int64_t n, const float* sumsq, float clip_norm,
                         const float* lr_t_dev, float beta1, float beta2, float eps, usseg_stream_t stream);
/* device-side step counter: *step += 1; *lr_t = lr*sqrt(1-b2^step)/(1-b1^step) */
int usseg_adam_advance(int32_t* step, float* lr_t_dev, float lr, float beta1, float beta2, usseg_stream_t stream);
int usseg_fill_f32(float* p, int64_t n, float value, usseg_stream_t stream);
int usseg_scale_f32(float* p, int64_t n, const float* sumsq, float clip_norm, usseg_stream_t stream);

/* ---- ViT bottleneck helpers (VisionTransformer.py:9-79,127-174) ------------------------
 * Batched GEMMs on the conv kernels (bf16 operands, fp32 accumulate), 2-level batch (image, head) with element strides:
 *   nt: Y[b1,b2][m][n]  = sum_k X[b1,b2][m][k] * W[b1,b2][n][k]   (M rows, K contiguous in both operands)
 *   tn: O[b1,b2][m][n] += sum_r A[b1,b2][r][m] * B[b1,b2][r][n]   (K = number of rows r; fp32 atomics, zero O first)
 * For tn, ldx and ldw are the row strides of A and B, and xs1, xs2, ws1, ws2 their batch strides. */
typedef struct UssegGemmDesc {
  int32_t M, N, K;
  int32_t ldx, ldw, ldy;
  int32_t nb1, nb2;
  int64_t xs1, xs2, ws1, ws2, ys1, ys2;
  int32_t flags;   /* USSEG_OUT_F32 (nt only) */
} UssegGemmDesc;
int usseg_gemm_nt_batched(const UssegGemmDesc* d, const void* x, const void* w, void* y, usseg_stream_t stream);
int usseg_gemm_tn_batched(const UssegGemmDesc* d, const void* a, const void* b, float* out, usseg_stream_t stream);
/* Row softmax of scale*S (fp32 [rows][ld]) -> P fp32 (the attention weights the reference returns, :44) and P bf16;
 * backward: dS = scale * P * (dP - sum_k dP*P)  (bf16 out). */
int usseg_softmax_rows_fwd(const float* s, int64_t rows, int32_t n, float scale, float* p32, void* pbf, usseg_stream_t stream);
int usseg_softmax_rows_bwd(const float* p32, const float* dp, int64_t rows, int32_t n, float scale, void* ds_bf, usseg_stream_t stream);
/* dst[b][c][r] = src[b][r][c]: bf16, src row stride lds, batch strides ss1/ss2 (2-level), dst dense [nb1*nb2][C][R]. */
int usseg_transpose_batched(const void* src, int32_t R, int32_t C, int32_t lds, int32_t nb1, int32_t nb2, int64_t ss1, int64_t ss2,
                            void* dst, usseg_stream_t stream);
/* dst_bf16[b1,b2][r][c] = src_f32[(b1*nb2+b2)][r][c] with dst row stride ldd and batch strides ds1/ds2 */
int usseg_cast_f32_to_bf16_batched(const float* src, int32_t R, int32_t C, int32_t nb1, int32_t nb2, void* dst, int32_t ldd, int64_t ds1,
                                   int64_t ds2, usseg_stream_t stream);

/* ---- fused multi-head attention (VisionTransformer.py:38-50, TBI_TransUNet.py:44-62; head size 128) -------------------------
 * O = softmax(scale * Q K^T) V per (image, head) without the [B, heads, N, N] score tensors in memory.  q / k / v point at head 0
 * of image 0 inside the fused projection output (bf16, row stride ld_qkv elements, image stride N*ld_qkv, head h at columns
 * h*128); o / d_o are [B][N][ld_o] bf16 with the same head columns.  lse (fp32 [B*H][N], base-2 log-sum-exp of the scaled scores)
 * is written by the forward and read by the backward; delta (fp32 [B*H][N]) is backward workspace.  o32 (optional, may be NULL)
 * is an fp32 copy of O, dense [B][N][H*128]: with it the backward's delta = sum_d dO*O is free of O's bf16 rounding, which keeps
 * the rows of dS summing to zero to rounding noise (the key bias gradient, exactly zero in the reference, stays at noise level).  dq / dk / dv are written
 * (not accumulated) with row stride ld_qkv - normally the three slices of one [B][N][3*hidden] gradient tensor.  The backward
 * recomputes the probabilities (two kernels: by query tile and by key tile), uses no atomics and is bitwise reproducible.
 * The attention weights the reference also returns (:44,56) are not produced: callers that want them use the unfused
 * usseg_gemm_nt_batched + usseg_softmax_rows_fwd path. */
typedef struct UssegFlashDesc {
  int32_t B, N, H, head_dim; /* images, tokens, heads, head size (128) */
  int32_t ld_qkv, ld_o;      /* row strides in elements */
  float scale;               /* 1/sqrt(num_heads) in the reference (VisionTransformer.py:42) */
} UssegFlashDesc;
int usseg_flash_attn_fwd(const UssegFlashDesc* d, const void* q, const void* k, const void* v, void* o, float* o32, float* lse,
                         usseg_stream_t stream);
int usseg_flash_attn_bwd(const UssegFlashDesc* d, const void* q, const void* k, const void* v, const void* o, const float* o32,
                         const void* d_o, const float* lse, float* delta, void* dq, void* dk, void* dv, usseg_stream_t stream);

/* ---- opt-in per-launch timing (bench.py roofline leg) ------------------------------------------
 * When enabled, every launch of the selected kernel family is bracketed by hipEventRecord on ITS stream.
 * kinds (bits): 1 = conv / tconv forward + backward-data launches, 2 = weight-gradient launches, 4 = fused tile kernels (usseg_cardinal_fwd /
 * _bwd, usseg_stem_fwd: convs AND their norms in one launch).  Events are created by
 * usseg_prof_enable (never inside a launch function); usseg_prof_read synchronises the recorded events and
 * returns the summed duration and the launch count.  Not graph-capture safe: disable before capturing. */
int usseg_prof_enable(int32_t kind_mask, int32_t capacity);
int usseg_prof_read(int32_t kind, double* total_ms, int64_t* launches);
int usseg_prof_disable(void);

#ifdef __cplusplus
}
#endif
#endif /* USSEG_H */
