/*
 * tok.h — C ABI of libtok_gfx950.so: the MI355X (gfx950 / CDNA4) hot path behind the
 * TorchOk vision-training step (backbone -> pooling -> head -> loss -> optimizer).
 *
 * The reference (eora-ai/torchok, pure Python) has no native boundary of its own: its hot
 * path bottoms out in ATen ops called from nn.Modules.  Every entry point below replaces one
 * of those ATen call sites; the citation on each declaration is the reference line
 * (/root/reference/...) or the timm-0.6.13 / torch symbol whose arithmetic it takes over.
 *
 * Conventions
 *  - Plain pointers and sizes only; no torch types.  All pointers are DEVICE pointers
 *    (HBM) unless a parameter says "host".  The caller owns every buffer; nothing is
 *    allocated or freed here; kernels borrow pointers for the duration of the enqueue.
 *  - Every call is asynchronous on `stream` (a hipStream_t passed as void*; NULL = the
 *    null stream).  No call synchronises the device.
 *  - Return value: 0 on success, negative tok_status otherwise; tok_last_error() returns
 *    a thread-local message for the last failure on the calling thread.
 *  - Activations: NHWC ("channels-last") bf16, channel count a multiple of 8 (the caller
 *    pads).  Convolution weights: bf16 [K][R][S][C] ("KRSC") for fwd, bf16 [C][R][S][K]
 *    spatially flipped for dgrad (tok_pack_weight_* produce both from the fp32 master).
 *    Statistics, gradients of parameters, optimizer state: fp32.
 *  - No communication entry point: the data-parallel gradient exchange (SURVEY.md section 8(b) sketched an
 *    ncclComm_t + stream + event call) stays on the host side of this boundary, in torchok_amd/dist/ddp.py over
 *    torch.distributed (backend "nccl" = RCCL): buckets are contiguous ranges of the gradient arena this library writes
 *    into, so the exchange needs no kernel of its own beyond tok_cast_f32_bf16 / tok_cast_bf16_f32 / tok_scale_f32.
 *  - Thread-safety: all entry points are re-entrant (PyTorch calls backward from an
 *    autograd worker thread).  Mutable state is thread-local only (the error string; the
 *    event armed by tok_next_launch_event).  Process-wide state is write-once: environment
 *    knobs (TOK_*) and the per-kernel hipFuncSetAttribute calls are function-local
 *    `static const` values, initialised exactly once under the C++11 guarantee for local
 *    statics — a knob changed after its first use has no effect.
 */
#ifndef TOK_H_
#define TOK_H_

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef enum tok_status {
  TOK_OK = 0,
  TOK_ERR_INVALID = -1,     /* bad descriptor / unsupported shape            */
  TOK_ERR_LAUNCH = -2,      /* hipLaunchKernel failed                        */
  TOK_ERR_WORKSPACE = -3    /* workspace too small                           */
} tok_status;

typedef enum tok_dtype { TOK_F32 = 0, TOK_F16 = 1, TOK_BF16 = 2 } tok_dtype;

/* Convolution geometry.  c and k are the PADDED channel counts of the bf16 activation
 * tensors (multiples of 8), except in "c4" stem mode (c == 4: 3 image channels padded to 4,
 * filter taps along S padded to `s_pad`).  groups == 1, dilation == 1 only
 * (resnet18/resnet50/hrnet use nothing else: resnet.py:472-490, [timm] Bottleneck).       */
typedef struct tok_conv_desc {
  int32_t n, h, w, c;   /* input  N, H, W, C                                  */
  int32_t k;            /* output channels                                    */
  int32_t r, s;         /* filter height, width                               */
  int32_t p, q;         /* output height, width                               */
  int32_t stride, pad;  /* same in both spatial dims                          */
  int32_t s_pad;        /* filter width as stored (== s, or 8 in c4 mode)     */
} tok_conv_desc;

const char* tok_last_error(void);
int tok_version(void);

/* ---- layout / dtype plumbing ------------------------------------------------------- */

/* image NCHW (f32|f16|bf16) -> NHWC bf16 with channels zero-padded to c_pad.
 * Replaces the implicit dtype cast + layout handling of torch autocast in front of
 * `backbone(input_data)` (tasks/classification.py:109).                                  */
int tok_nchw_to_nhwc_bf16(const void* src, int src_dtype, int n, int c, int h, int w,
                          void* dst, int c_pad, void* stream);
/* fp32 -> bf16 elementwise cast (refreshing the bf16 weight shadow of the fp32 masters). */
int tok_cast_f32_bf16(const float* src, void* dst, size_t count, void* stream);
/* dst[i] = float(src[i]) * scale: the widening half of a bf16 gradient exchange (dist/ddp.py: torch DDP's
   bf16_compress_hook equivalent, scale = 1 / world for backends without an averaging all-reduce).  16-byte aligned buffers. */
int tok_cast_bf16_f32(const void* src, float* dst, float scale, size_t count, void* stream);
/* master fp32 [k][r][s][c] -> bf16 [k_pad][r][s_pad][c_pad] (zero padded)               */
int tok_pack_weight_fwd(const float* src, int k, int r, int s, int c,
                        void* dst, int k_pad, int s_pad, int c_pad, void* stream);
/* master fp32 [k][r][s][c] -> bf16 [c_pad][r][s][k_pad], taps flipped (dgrad operand)    */
int tok_pack_weight_dgrad(const float* src, int k, int r, int s, int c,
                          void* dst, int k_pad, int c_pad, void* stream);

/* both packs in one launch (what a training forward needs)                                */
int tok_pack_weight_both(const float* src, int k, int r, int s, int c, void* dst_fwd, int k_pad,
                         int s_pad, int c_pad, void* dst_dgrad, void* stream);

/* every weight of an optimizer in ONE launch (called right after the optimizer step: the bf16
 * operands of the next forward are refreshed while the masters are still in cache).  `items_dev`
 * is a DEVICE array sorted by block_start, block_start[i+1] = block_start[i] +
 * tok_pack_item_blocks(&item[i]); dst_dgrad (or dst_fwd) may be NULL.                          */
typedef struct tok_pack_item {
  const float* src;
  void* dst_fwd;
  void* dst_dgrad;
  int32_t k, r, s, c, k_pad, s_pad, c_pad;
  int32_t block_start;
} tok_pack_item;
int tok_pack_item_blocks(const tok_pack_item* item /* host */);
int tok_pack_weights_batched(const tok_pack_item* items_dev, int n_items, int total_blocks, void* stream);

/* ---- convolution (implicit GEMM on MFMA) --------------------------------------------
 * Replace aten::conv2d fwd/bwd reached from resnet.py:488 (stem), [timm] BasicBlock /
 * Bottleneck conv1-3, downsample_conv (resnet.py:383-387), hrnet.py:64-69,122,133,156,
 * convbnact.py:38-44, and aten::linear at linear_head.py:31 / poolings linear.py:18
 * (a linear layer is the 1x1 case with h = w = 1).                                        */

/* Rows of the per-channel partial-statistics buffer tok_conv_fwd fills: the buffer is
 * float[2][rows][k] (sum, then sum of squares, of the bf16-rounded outputs).              */
int tok_conv_fwd_stat_rows(const tok_conv_desc* d);
/* y = conv(x, w) (+ bias[k] if bias != NULL).  stats may be NULL.                         */
int tok_conv_fwd(const tok_conv_desc* d, const void* x, const void* w, const float* bias,
                 void* y, float* stats, void* stream);
/* dx = conv_transpose(dy, w) using the flipped/transposed operand from
 * tok_pack_weight_dgrad;  accumulate != 0 adds into the existing dx (residual fan-in).     */
int tok_conv_dgrad(const tok_conv_desc* d, const void* dy, const void* w_dgrad, void* dx,
                   int accumulate, void* stream);
/* dgrad fused with the BatchNorm-backward reduction of the unit that PRODUCED x (the tensor
 * whose gradient dx this call completes): the epilogue, holding the final dx tile in
 * registers, also loads that unit's raw conv output bn_y (same NHWC shape as dx) and its ReLU
 * bit mask (nullable = no ReLU) and accumulates sum(dz), sum(dz*y), dz = dx*mask, into
 * partial[2][tok_conv_dgrad_stat_rows(d)][c] — replacing a separate tok_bn_bwd_reduce pass over
 * dx and y (feed the result to tok_bn_bwd_finalize with dzy_form = 1).                     */
int tok_conv_dgrad_stat_rows(const tok_conv_desc* d);
int tok_conv_dgrad_bnstats(const tok_conv_desc* d, const void* dy, const void* w_dgrad, void* dx,
                           int accumulate, const void* bn_y, const uint8_t* bn_mask, float* partial,
                           void* stream);
/* Fused BatchNorm finalize.  The conv launches that produce BatchNorm partial rows can also fold them: the
 * LAST workgroup of a channel tile to deliver its row (device-side ticket) does the work of tok_bn_finalize
 * (tok_conv_fwd_bn) or of tok_bn_bwd_finalize in dzy form (tok_conv_dgrad_bn) for that tile's channels — one
 * kernel less on the dependency chain conv -> finalize -> apply.  `counters`: >= 64 device ints, zero on entry,
 * left zero (give concurrent launches distinct counters).  Same arithmetic as the stand-alone finalizes.     */
typedef struct tok_bn_fused {
  int32_t* counters;
  int64_t count;             /* elements per channel (n*p*q of the normalised tensor) */
  int32_t c_real;            /* num_features (<= padded channel count) */
  int32_t param_accumulate;  /* backward: dgamma / dbeta += */
  float momentum, eps;       /* forward */
  const float* gamma;
  const float* beta;         /* forward */
  float* running_mean;       /* forward, may be NULL (with running_var, nbt) */
  float* running_var;
  int64_t* nbt;
  float* mean;               /* forward: out; backward: in */
  float* rstd;
  float* scale;              /* forward: out */
  float* shift;
  float* dgamma;             /* backward: out (may be NULL) */
  float* dbeta;
  float* coef;               /* backward: out [3][c] */
} tok_bn_fused;
int tok_conv_fwd_bn(const tok_conv_desc* d, const void* x, const void* w_fwd, void* y, float* stats,
                    const tok_bn_fused* bn, void* stream);
int tok_conv_dgrad_bn(const tok_conv_desc* d, const void* dy, const void* w_dgrad, void* dx, int accumulate,
                      const void* bn_y, const uint8_t* bn_mask, float* partial, const tok_bn_fused* bn,
                      void* stream);
/* ---- "unit 3" of a bottleneck without its pre-normalisation tensor ------------------------------------------------
 * [timm] Bottleneck: x = conv3(x); x = bn3(x); x += shortcut; x = act3(x)  (via torchok/models/backbones/resnet.py:12-14).
 * The 4P-channel tensor between conv3 and bn3 is the largest of the block; because conv3 is a 1x1 convolution its batch
 * statistics follow from the P x P second moments of conv3's INPUT, so the normalisation can run in the GEMM epilogue and the
 * backward pass can be written on tensors that exist anyway (csrc/unit3.hip has the algebra).
 *
 * tok_bn_gram_finalize: Z = z^T z (P x P, e.g. from tok_conv_wgrad on (z, z)), zsum = column sums of z, w = fp32 master
 *   filter [k][p] (rounded to bf16 inside, the operand the GEMM uses) -> mean / rstd / scale / shift [k] and the running
 *   statistics update of F.batch_norm (momentum, unbiased variance, num_batches_tracked += 1); count = rows of z.
 * tok_conv_fwd_bn_apply: out = act(conv1x1(x) * scale + shift (+ shortcut)) as bf16, act = ReLU or identity (the projection
 *   shortcut conv + BatchNorm of [timm] downsample_conv), mask bit = (out > 0)  (the bits tok_bn_act_fwd writes); replaces
 *   tok_conv_fwd + tok_bn_finalize + tok_bn_act_fwd for such a unit.
 * tok_conv_dgrad_maskstore: tok_conv_dgrad whose epilogue stores dz = mask ? dx : 0 and reduces sum(dz) into
 *   partial[2][tok_conv_dgrad_stat_rows][c] (second half zero) — for the launch that COMPLETES the gradient of the unit's
 *   output; tok_relu_mask_reduce is the stand-alone form (partial[2][tok_bn_bwd_rows(m, c)][c], in place allowed;
 *   mask NULL: plain column sums of dout).
 * tok_bn3_bwd_prepare: G = dz^T z [k][p], w, wz (from tok_bn_gram_finalize), zsum, the sum(dz) partial rows -> dgamma / dbeta (+= if param_accumulate),
 *   coef [3][k], dw [k][p] (+= if dw_accumulate), wa = bf16 diag(c1) W in dgrad-pack layout [p][k], wb = bf16
 *   W^T diag(c2) W [p][p], cvec = c3^T W [p]:   d(input) = dz wa + z wb + cvec  (tok_conv_dgrad + tok_conv_dgrad_bias).
 * tok_conv_dgrad_bias: tok_conv_dgrad / tok_conv_dgrad_bnstats with a per-channel fp32 bias added to the result.          */
int tok_bn_gram_finalize(const float* Z, const float* zsum, const float* w, int64_t count, int p, int k,
                         const float* gamma, const float* beta, float* running_mean, float* running_var,
                         int64_t* num_batches_tracked, float momentum, float eps, float* mean, float* rstd,
                         float* scale, float* shift, float* wz /* out [k][p] = W_bf16 Z, input of tok_bn3_bwd_prepare */,
                         void* stream);
int tok_conv_fwd_bn_apply(const tok_conv_desc* d, const void* x, const void* w, const float* scale, const float* shift,
                          const void* shortcut /* may be NULL */, int relu, void* out, uint8_t* mask /* may be NULL */,
                          void* stream);
int tok_conv_dgrad_maskstore(const tok_conv_desc* d, const void* dy, const void* w_dgrad, void* dx, int accumulate,
                             const uint8_t* mask, float* partial, void* stream);
int tok_relu_mask_reduce(const void* dout, const uint8_t* mask, int64_t m, int c, void* dz, float* partial, void* stream);
int tok_bn3_bwd_prepare(const float* G, const float* w, const float* wz, const float* zsum, const float* partial, int rows,
                        int64_t count, int p, int k, const float* gamma, const float* mean, const float* rstd,
                        float* dgamma, float* dbeta, int param_accumulate, float* coef, float* dw, int dw_accumulate,
                        void* wa, void* wb, float* cvec, float* ws /* scratch, tok_bn3_bwd_prepare_ws_floats(p, k) */,
                        void* stream);
size_t tok_bn3_bwd_prepare_ws_floats(int p, int k);
int tok_conv_dgrad_bias(const tok_conv_desc* d, const void* dy, const void* w_dgrad, const float* bias, void* dx,
                        int accumulate, const void* bn_y, const uint8_t* bn_mask, float* partial, void* stream);
/* Stride-2 projection shortcuts ([timm] downsample_conv with stride 2, resnet.py) as pointwise layers:
 * tok_subsample2_fwd: out[b][p][q][:] = x[b][2p][2q][:], p < ceil(h/2), q < ceil(w/2) — conv1x1/stride2(x) == conv1x1(out).
 * tok_subsample2_bwd: dx[b][hh][ww][:] (+= if accumulate) = (hh, ww even) ? dsub[b][hh/2][ww/2][:] : 0 (stand-alone form).
 * tok_conv_dgrad_subacc: dx = dgrad_1x1(dy) + that scatter of dsub, in the epilogue of the pointwise data gradient that
 *   shares the input tensor (conv1 of a strided bottleneck); optional BatchNorm-backward sums (bn_y, mask, partial: as
 *   tok_conv_dgrad_bnstats) or, with mask_store, the ReLU-masked store of tok_conv_dgrad_maskstore.  Served for the layers
 *   tok_conv_dgrad_subacc_ok(d) returns 1 for; TOK_ERR_INVALID otherwise. */
int tok_subsample2_fwd(const void* x, int n, int h, int w, int c, void* out, void* stream);
int tok_subsample2_bwd(const void* dsub, int n, int h, int w, int c, void* dx, int accumulate, void* stream);
int tok_conv_dgrad_subacc_ok(const tok_conv_desc* d);
int tok_conv_dgrad_subacc(const tok_conv_desc* d, const void* dy, const void* w_dgrad, void* dx, const void* dsub,
                          const void* bn_y /* may be NULL */, const uint8_t* mask /* may be NULL */,
                          float* partial /* may be NULL */, int mask_store, void* stream);
/* Sum of two pointwise data gradients over the same pixels and input width in ONE launch:
 * dx (+= if accumulate) = dgrad(d1; dy1, w1_dgrad) + dgrad(d2; dy2, w2_dgrad) (+ bias), epilogue options of tok_conv_dgrad_bias
 * (partial sized with tok_conv_dgrad_stat_rows(d2)).  The fused residual unit's d(input) = dz Wa + z Wb + c.  Served where
 * tok_conv_dgrad2_ok(d1, d2) returns 1. */
int tok_conv_dgrad2_ok(const tok_conv_desc* d1, const tok_conv_desc* d2);
int tok_conv_dgrad2(const tok_conv_desc* d1, const void* dy1, const void* w1_dgrad, const tok_conv_desc* d2, const void* dy2,
                    const void* w2_dgrad, const float* bias /* may be NULL */, void* dx, int accumulate,
                    const void* bn_y /* may be NULL */, const uint8_t* bn_mask /* may be NULL */, float* partial /* with bn_y */,
                    void* stream);
size_t tok_conv_wgrad_ws_bytes(const tok_conv_desc* d);
/* dw fp32 [k_real][r][s][c_real] (+= if accumulate) from x, dy; ws = scratch of at least
 * tok_conv_wgrad_ws_bytes(d) bytes.  k_real/c_real/s are the unpadded master dims.        */
int tok_conv_wgrad(const tok_conv_desc* d, const void* x, const void* dy, float* dw,
                   int k_real, int c_real, void* ws, size_t ws_bytes, int accumulate,
                   void* stream);
/* tok_conv_wgrad that also produces the bias gradient dbias[k_real] (+= if bias_accumulate) = column sums of dy, from the dy
 * fragments the kernel already holds (one extra MFMA against an all-ones operand per fragment) — replaces the separate
 * tok_colsum_partial + tok_colsum_f32 pass of nn.Linear / conv biases ([timm] Mlp, WindowAttention qkv/proj; swin_v2.py).
 * Served for the layers tok_conv_wgrad_bias_ok(d) returns 1 for (the pointwise layers); ws of at least
 * tok_conv_wgrad_bias_ws_bytes(d) bytes. */
int tok_conv_wgrad_bias_ok(const tok_conv_desc* d);
size_t tok_conv_wgrad_bias_ws_bytes(const tok_conv_desc* d);
int tok_conv_wgrad_bias(const tok_conv_desc* d, const void* x, const void* dy, float* dw, int k_real, int c_real,
                        void* ws, size_t ws_bytes, int accumulate, float* dbias, int bias_accumulate, void* stream);

/* ---- batch norm (training mode, batch statistics) -------------------------------------
 * Replace aten::batch_norm fwd/bwd + relu_ + residual add_ reached from resnet.py:489-490,
 * [timm] blocks (bn1..3, act1..3, `x += shortcut`), hrnet.py:65-69, convbnact.py:45-53.    */

/* Reduce tok_conv_fwd partials -> mean, rstd (biased var), scale = gamma*rstd,
 * shift = beta - mean*scale; running_mean/var momentum update (unbiased var), and
 * num_batches_tracked += 1 (int64) — the semantics of torch.nn.BatchNorm2d in training.
 * running_* / nbt may be NULL (track_running_stats=False).
 * c = padded channel count of the activation (multiple of 8), c_real <= c the module's
 * num_features (HRNet-W18: 18 -> 24): parameter / running-stat arrays hold c_real entries,
 * padding channels get scale = shift = mean = rstd = 0 (their activations stay zero).        */
int tok_bn_finalize(const float* stats, int rows, int64_t count, int c, int c_real,
                    const float* gamma, const float* beta,
                    float* running_mean, float* running_var, int64_t* nbt,
                    float momentum, float eps,
                    float* mean, float* rstd, float* scale, float* shift, void* stream);
/* eval mode: scale/shift from running statistics                                          */
int tok_bn_eval_coeffs(const float* gamma, const float* beta, const float* running_mean,
                       const float* running_var, float eps, int c, int c_real,
                       float* scale, float* shift, void* stream);
/* per-channel sum / sumsq partials of an NHWC bf16 tensor (BN after a non-conv producer)  */
int tok_bn_stats_rows(int64_t m, int c);
int tok_bn_stats(const void* y, int64_t m, int c, float* stats, void* stream);
/* out = act(y*scale[c] + shift[c] (+ shortcut));  act = relu if relu != 0.
 * mask (nullable): uint8 [m][c/8], bit e of byte (row, g) = (out[row][8g+e] > 0) — the ReLU mask
 * the backward kernels read instead of `out` (1/16 of its bytes).                            */
int tok_bn_act_fwd(const void* y, const float* scale, const float* shift,
                   const void* shortcut, int relu, void* out, uint8_t* mask, int64_t m, int c,
                   void* stream);
/* tok_bn_act_fwd that also leaves per-block column sums of `out` (the fused residual unit's colsum(z) without a pass over z):
 * partial fp32 [tok_bn_act_fwd_colsum_rows(m, c)][c], folded by tok_colsum_f32 */
/* Completion events carried by a launch: tok_next_launch_event(ev) makes the calling thread's NEXT tok_bn_bwd_apply signal ev
 * when the kernel finishes (no separate record packet on that stream); tok_stream_wait_event orders another stream behind it. */
void* tok_event_create(void);
int tok_event_destroy(void* ev);
int tok_next_launch_event(void* ev);
int tok_stream_wait_event(void* stream, void* ev);
int tok_bn_act_fwd_colsum_rows(int64_t m, int c);
int tok_bn_act_fwd_colsum(const void* y, const float* scale, const float* shift, const void* shortcut, int relu, void* out,
                          uint8_t* mask, int64_t m, int c, float* partial, void* stream);
int tok_bn_bwd_rows(int64_t m, int c);
/* partial[2][rows][c]: sum(dz), sum(dz * xhat), dz = dout * relu_mask.
 * mask = the bit mask tok_bn_act_fwd wrote (may be NULL when no shortcut was added: the mask
 * is then recomputed from y, scale, shift).                                                */
int tok_bn_bwd_reduce(const void* dout, const void* y, const uint8_t* mask,
                      const float* scale, const float* shift,
                      const float* mean, const float* rstd, int relu,
                      int64_t m, int c, float* partial, void* stream);
/* dgamma, dbeta (+= if accumulate) and the 3 per-channel coefficients of
 * dy = a1*dz + a2*y + a3  ->  coef[3][c].  dzy_form != 0: the second partial is sum(dz * y)
 * (what tok_conv_dgrad_bnstats accumulates) instead of sum(dz * xhat).                     */
int tok_bn_bwd_finalize(const float* partial, int rows, int64_t m, int c, int c_real,
                        const float* gamma, const float* mean, const float* rstd,
                        float* dgamma, float* dbeta, float* coef, int accumulate, int dzy_form,
                        void* stream);
/* dy = a1*dz + a2*y + a3;  if dshortcut != NULL: dshortcut (=|+=) dz                       */
int tok_bn_bwd_apply(const void* dout, const void* y, const uint8_t* mask,
                     const float* scale, const float* shift, const float* coef, int relu,
                     void* dy, void* dshortcut, int dshortcut_accumulate,
                     int64_t m, int c, void* stream);


/* ---- pooling ----------------------------------------------------------------------------
 * aten::max_pool2d(3, stride 2, pad 1) at resnet.py:510; adaptive avg pool + flatten at
 * poolings/classification/pooling.py:7-12 ([timm] SelectAdaptivePool2d).                   */
int tok_maxpool3x3s2_fwd(const void* x, void* y, uint8_t* argmax, int n, int h, int w, int c,
                         void* stream);
int tok_maxpool3x3s2_bwd(const void* dy, const uint8_t* argmax, void* dx, int accumulate,
                         int n, int h, int w, int c, void* stream);
/* AvgPool2d(2, stride 2, ceil_mode=True, count_include_pad=False) of [timm] downsample_avg (avg_down shortcut of the
 * d-variant ResNets, resnet.py:598-603,...): NHWC bf16 [n][h][w][c] -> [n][ceil(h/2)][ceil(w/2)][c], c % 8 == 0.       */
int tok_avgpool2x2_fwd(const void* x, void* y, int n, int h, int w, int c, void* stream);
int tok_avgpool2x2_bwd(const void* dy, void* dx, int accumulate, int n, int h, int w, int c, void* stream);
int tok_gap_fwd(const void* x, void* y, int n, int hw, int c, void* stream);
int tok_gap_bwd(const void* dy, void* dx, int accumulate, int n, int hw, int c, void* stream);
/* SelectAdaptivePool2d(1, pool_type) beyond 'avg' ([timm] adaptive_avgmax_pool; reference
 * poolings/classification/pooling.py:7-12): mode 1 'max', 2 'avgmax' = 0.5 * (avg + max), 3 'catavgmax' = cat(avg, max)
 * (y row = [avg(c) | max(c)], needs ldy >= 2c).  argmax [n][c] int32 = first maximal pixel per (image, channel), the
 * index-exact route of the max gradient (ATen adaptive_max_pool2d_backward).                                            */
int tok_global_pool_fwd(const void* x, void* y, int* argmax, int n, int hw, int c, int ldy, int mode, void* stream);
int tok_global_pool_bwd(const void* dy, const int* argmax, void* dx, int accumulate, int n, int hw, int c, int ldy,
                        int mode, void* stream);

/* column sums of a bf16 [m][n] matrix -> fp32 (bias gradients of Linear / biased conv)     */
int tok_colsum(const void* dy, int64_t m, int n_pad, int n_real, float* out, int accumulate,
               void* stream);

/* ---- loss -------------------------------------------------------------------------------
 * torch.nn.CrossEntropyLoss(mean, ignore_index) registered at losses/__init__.py:26, called
 * from JointLoss.forward (losses/base.py:78-79).  logits bf16 [rows][ld] (classes <= ld).  */
#define TOK_CE_LOSS_FLOATS 2050 /* loss buffer: [0]=mean loss, [1]=n_valid, rest = reduction scratch */
int tok_softmax_ce_fwd(const void* logits, const int64_t* target, int rows, int classes, int ld,
                       int64_t ignore_index, float* lse, float* row_loss,
                       float* loss /* TOK_CE_LOSS_FLOATS floats, 8-byte aligned */, void* stream);
/* dlogits = (softmax - onehot) * gscale[0] / n_valid  (0 for ignored rows / pad columns)   */
int tok_softmax_ce_bwd(const void* logits, const int64_t* target, const float* lse,
                       const float* loss, const float* gscale, int rows, int classes, int ld,
                       int64_t ignore_index, void* dlogits, void* stream);

/* the same with torch's label_smoothing in [0, 1]: row loss = (1 - s)(lse - z_t) + s (lse - mean_c z_c), and
 * dlogits = (softmax - (1 - s) onehot - s / classes) * gscale[0] / n_valid.  s = 0 is tok_softmax_ce_fwd/_bwd bit for bit. */
int tok_softmax_ce_smooth_fwd(const void* logits, const int64_t* target, int rows, int classes, int ld,
                              int64_t ignore_index, float label_smoothing, float* lse, float* row_loss,
                              float* loss, void* stream);
int tok_softmax_ce_smooth_bwd(const void* logits, const int64_t* target, const float* lse, const float* loss,
                              const float* gscale, int rows, int classes, int ld, int64_t ignore_index,
                              float label_smoothing, void* dlogits, void* stream);

/* Pixel-wise cross entropy ON the bilinearly upsampled logits without the upsampled tensor: SegmentationHead.forward ends in
 * F.interpolate(segm_logits, size=input.shape[2:], mode='bilinear') (models/heads/segmentation/base.py:31-41, align_corners
 * None == False) and the segmentation recipes put CrossEntropyLoss on the result (losses/__init__.py:26).  low = the
 * channel-last bf16 logits [n][hs][ws][ld] (ld = classes padded to 8, <= 32), target int64 [n][hd][wd].  Forward: every
 * full-resolution pixel interpolates its logits from its 2x2 source footprint (ATen's source-index formula, the interpolated
 * value rounded to bf16 exactly as tok_bilinear_fwd stores it), lse / row_loss [n*hd*wd] and the mean in loss[] as
 * tok_softmax_ce_fwd.  Backward: d(low) (+)= adjoint of the interpolation applied to bf16(d upsampled logits), in gather form
 * (deterministic) — the values tok_softmax_ce_bwd + tok_bilinear_bwd produce, without their 2 x [n][hd][wd][ld] tensors.   */
int tok_upsample_ce_serves(int classes, int ld);
int tok_upsample_ce_fwd(const void* low, int n, int hs, int ws, int classes, int ld, int hd, int wd, const int64_t* target,
                        int64_t ignore_index, float* lse, float* row_loss, float* loss, void* stream);
int tok_upsample_ce_bwd(const void* low, int n, int hs, int ws, int classes, int ld, int hd, int wd, const int64_t* target,
                        int64_t ignore_index, const float* lse, const float* loss, const float* gscale, void* dlow,
                        int accumulate, void* stream);

/* DiceLoss (losses/segmentation/dice.py:86-188) on bf16 logits rows [rows][ld] (pixels of the channel-last logits).
 * mode 0 'multiclass': softmax + one_hot(target int64 [rows]); mode 1 'binary': sigmoid of column 0, target float32
 * [rows], classes = 1; mode 2 'multilabel': sigmoid per class, target float32 [rows][classes].  dims=(0, 2) statistics per class, `1 - dice` or `-log(dice)`, classes without true pixels
 * masked, mean over the (selected) classes.  partial fp32 [tok_dice_rows(rows)][3][classes] (scratch), loss fp32 [1],
 * coef fp32 [2][classes] (saved: d loss / d p = coef[0][c] * y + coef[1][c]).                                      */
int tok_dice_rows(int64_t rows);
int tok_dice_fwd(const void* logits, const void* target, int64_t rows, int classes, int ld, int mode, float smooth,
                 float eps, int log_loss, const int64_t* class_sel, int n_sel, float* partial, float* loss,
                 float* coef, void* stream);
int tok_dice_bwd(const void* logits, const void* target, const float* coef, const float* gscale, int64_t rows,
                 int classes, int ld, int mode, void* dlogits, void* stream);

/* BCEWithLogitsLoss with an ignore value (losses/classification/binary_cross_entropy.py:13-59; its forward :50-59 drops
 * the elements whose target equals ignore_index, then F.binary_cross_entropy_with_logits over the rest, 'mean' or 'sum';
 * nothing selected -> 0).  logits bf16 [rows][ld], target fp32 [rows][classes] (dense), loss as in tok_softmax_ce_fwd:
 * TOK_CE_LOSS_FLOATS floats, [0] = loss, [1] = number of selected elements.
 * backward: dlogits bf16 [rows][ld] = (sigmoid(x) - t) * gscale[0] / (mean ? n_selected : 1), 0 elsewhere.          */
int tok_bce_logits_fwd(const void* logits, const float* target, int64_t rows, int classes, int ld,
                       float ignore_value, int mean, float* loss, void* stream);
int tok_bce_logits_bwd(const void* logits, const float* target, const float* loss, const float* gscale,
                       int64_t rows, int classes, int ld, float ignore_value, int mean, void* dlogits, void* stream);

/* torch.nn.L1Loss / MSELoss / SmoothL1Loss / HuberLoss as registered by the reference (losses/__init__.py:13,19,23,24):
 * x bf16 [n] (flat), target fp32 [n]; kind 0 L1, 1 MSE, 2 smooth-L1 (knee = beta), 3 Huber (knee = delta); 'mean' or 'sum'.
 * loss: TOK_CE_LOSS_FLOATS floats, [0] = loss.  backward: dx bf16 [n] = d elem / d x * gscale[0] / (mean ? n : 1).        */
int tok_regression_loss_fwd(const void* x, const float* target, int64_t n, int kind, float knee, int mean, float* loss,
                            void* stream);
int tok_regression_loss_bwd(const void* x, const float* target, const float* gscale, int64_t n, int kind, float knee,
                            int mean, void* dx, void* stream);

/* On-device classification statistics behind the Accuracy / F1Score metrics the reference configs log every step
 * (metrics/metrics_manager.py:147-158, classification_cifar10.yaml:134-150; torchmetrics itself is third-party):
 * counts int64 [3][classes] += {true positives, predicted, actual} per class.  Predictions are bf16 logits
 * [rows][ld] (first maximum, as torch.argmax) or int64 labels; rows whose target is ignore_index are skipped.   */
int tok_cls_stats_update(const void* logits, const int64_t* labels, const int64_t* target, int64_t rows,
                         int classes, int ld, int64_t ignore_index, int64_t* counts, void* stream);
/* ConfusionMatrix (metrics/__init__.py:53): confusion int64 [classes][classes], [target][prediction] += 1, same inputs */
int tok_confusion_update(const void* logits, const int64_t* labels, const int64_t* target, int64_t rows,
                         int classes, int ld, int64_t ignore_index, int64_t* confusion, void* stream);

/* ---- metric-learning head and loss -----------------------------------------------------------
 * F.normalize (arcface_head.py:125-126, linear_head.py:33-34): y = x / max(||x||_2, eps) per row;
 * is_f32 selects fp32 rows (class-weight matrix) instead of bf16 activations.               */
int tok_l2norm_fwd(const void* x, void* y, float* inv_norm, int rows, int c, int ld, int is_f32,
                   float eps, void* stream);
int tok_l2norm_bwd(const void* dy, const void* y, const float* inv_norm, void* dx, int accumulate,
                   int rows, int c, int ld, int is_f32, void* stream);
/* ArcFaceHead.__add_margin (arcface_head.py:95-108) on the bf16 cosine matrix [rows][ld]:
 * the target column becomes phi (or its easy/hard fallback), everything is scaled.          */
int tok_arcface_margin_fwd(const void* cosine, const int64_t* target, int rows, int classes, int ld,
                           float cos_m, float sin_m, float th, float mm, int easy_margin, float scale,
                           void* out, void* stream);
int tok_arcface_margin_bwd(const void* cosine, const int64_t* target, const void* dout, int rows,
                           int classes, int ld, float cos_m, float sin_m, float th, float mm,
                           int easy_margin, float scale, void* dcos, void* stream);
/* PairwiseLearnTask.calc_relevance_matrix for 1-D labels (pairwise_task.py:87-107): exact.   */
int tok_relevance_matrix(const int64_t* labels_a, const int64_t* labels_b, int na, int nb, float* R,
                         void* stream);
/* the same for multi-label matrices ya fp32 [na][classes], yb [nb][classes] (:103-105): R_ij = [sum_c ya_ic yb_jc > 0] */
int tok_relevance_matrix_multilabel(const float* ya, const float* yb, int na, int nb, int classes, float* R,
                                    void* stream);
/* ContrastiveLoss (losses/representation/pairwise.py:126-136, mean reduction, no regulariser):
 * S = cdist(e1, e2) [n1][n2] (saved), loss[0] = mean_i sum_j (1-R)relu(mu-S)^2 + R S^2.      */
int tok_contrastive_fwd(const void* e1, const void* e2, const float* R, int n1, int n2, int d, int ld,
                        float margin, float* S, float* row_loss, float* loss, void* stream);
int tok_contrastive_bwd(const void* e1, const void* e2, const float* R, const float* S,
                        const float* gscale, int n1, int n2, int d, int ld, float margin, void* de1,
                        void* de2, int same_tensor, void* stream);

/* BasePairwiseLoss.regularize (pairwise.py:28-46) on the first embedding matrix, bf16 [n][ld]: mode 1 'L1' reg_i = sum_c |e_ic|,
 * mode 2 'L2' reg_i = ||e_i||_2; row_reg fp32 [n] (saved), out[0] = mean_i reg_i.
 * backward: de bf16 [n][ld] = gscale[0] * coeff * (sign(e) | e / reg_i), pad columns zero.                       */
int tok_embed_reg_fwd(const void* e, int n, int d, int ld, int mode, float* row_reg, float* out, void* stream);
int tok_embed_reg_bwd(const void* e, const float* row_reg, const float* gscale, float coeff, int n, int d, int ld,
                      int mode, void* de, void* stream);

/* ---- multi-resolution glue (HRNet) ------------------------------------------------------------
 * [timm 0.6.13] HighResolutionModule.forward: out = relu(sum_j up_j(t_j)), where term j is an NHWC
 * bf16 tensor at (h >> s_j, w >> s_j) and up_j the nearest-neighbour nn.Upsample(scale_factor=2^s_j)
 * of its fuse layer (s_j = 0: same resolution).  Unused terms are NULL.  `mask` (may be NULL)
 * receives the ReLU bits [n*h*w][c/8].  Backward, once per term: dterm (+)= block-sum of the masked
 * output gradient (mask NULL: no ReLU).                                                          */
int tok_fuse_sum_relu_fwd(const void* t0, int s0, const void* t1, int s1, const void* t2, int s2,
                          const void* t3, int s3, int n, int h, int w, int c, int relu, void* out,
                          uint8_t* mask, void* stream);
/* ... with per-term BatchNorm coefficients: a term j given with sc_j / sf_j [c] (fp32) is the RAW convolution output of a unit
 * without activation (the last Conv2d + BatchNorm2d of a fuse path, hrnet.py:122-140) and enters the sum as t * sc + sf — the
 * unit's apply pass (tok_bn_act_fwd: a write and a read of the term) is folded into this kernel; sc_j NULL: as above.       */
int tok_fuse_sum_affine_relu_fwd(const void* t0, int s0, const float* sc0, const float* sf0, const void* t1, int s1,
                                 const float* sc1, const float* sf1, const void* t2, int s2, const float* sc2,
                                 const float* sf2, const void* t3, int s3, const float* sc3, const float* sf3, int n, int h,
                                 int w, int c, int relu, void* out, uint8_t* mask, void* stream);
int tok_fuse_sum_relu_bwd(const void* dout, const uint8_t* mask, int n, int h, int w, int c, int shift,
                          void* dterm, int accumulate, void* stream);
/* F.interpolate(mode='bilinear', align_corners=False) (necks/segmentation/hrnet.py:36-39,
 * heads/segmentation/base.py:37): src [n][hs][ws][ld_src] (first c channels) -> channels
 * [ch_off, ch_off + c) of dst [n][hd][wd][ld_dst] — i.e. torch.cat (hrnet.py:41) costs nothing.
 * hs == hd && ws == wd is an exact strided copy.  Backward = exact transpose, gather form.        */
int tok_bilinear_fwd(const void* src, int n, int hs, int ws, int c, int ld_src, void* dst, int hd,
                     int wd, int ld_dst, int ch_off, void* stream);
int tok_bilinear_bwd(const void* ddst, int n, int hd, int wd, int ld_dst, int ch_off, void* dsrc,
                     int hs, int ws, int c, int ld_src, int accumulate, void* stream);
/* y = y0 + sum_j bilinear(t_j -> h x w) (align_corners=False as above; t_j NULL = absent), every map [..][c] bf16 with row
 * pitch c, summed in fp32 and rounded once; y may be y0.  stats (nullable): float[2][tok_bilinear_sum_stats_rows(n, h, w, c)][c]
 * per-channel partial (sum, sum of squares) of the rounded y, the rows tok_bn_finalize folds.
 * HRNetSegmentationNeck (necks/segmentation/hrnet.py:36-45) is ConvBnRelu1x1(cat_j interpolate(x_j)): a 1x1 convolution
 * commutes with the interpolation, conv(cat_j up(x_j)) = sum_j up(conv_j(x_j)) with conv_j = the filter columns of source j,
 * so the product runs at every source's own resolution and this entry point adds the results up (and leaves the BatchNorm
 * statistics of the sum); the concat tensor is never built.                                              */
/* The transposes of up to three interpolations of ONE map: d_j [n][h_j][w_j][c] = up_j^T ddst, ddst [n][hd][wd][c] (row pitch c
 * everywhere; d_j NULL = absent) — tok_bilinear_bwd (ch_off 0, no accumulate) once per source, as ONE pass over ddst where
 * every factor hd / h_j is 2, 4 or 8 and hd, wd are multiples of 16 (the commuted neck's d(y_j) = up_j^T d(y)).           */
int tok_bilinear_bwd_multi(const void* ddst, int n, int hd, int wd, int c, void* d1, int h1, int w1, void* d2, int h2,
                           int w2, void* d3, int h3, int w3, void* stream);
int tok_bilinear_sum_stats_rows(int n, int h, int w, int c);
int tok_bilinear_sum_stats(const void* y0, const void* t1, int h1, int w1, const void* t2, int h2, int w2,
                           const void* t3, int h3, int w3, int n, int h, int w, int c, void* y, float* stats,
                           void* stream);

/* ---- token-major transformer units (SwinV2) -------------------------------------------------------
 * swin.py:71-256 over [timm 0.6.13] swin_transformer_v2.  Tokens are rows of a bf16 [rows][ld] matrix
 * (rows = B*H*W, first c columns valid).                                                            */

/* out = shortcut + row_scale[row / rows_per_sample] * LayerNorm(x)   (shortcut / row_scale may be NULL):
 * nn.LayerNorm, and the res-post-norm residual `x + drop_path(norm(f(x)))` of SwinTransformerBlock with the
 * per-sample stochastic-depth factor.  mean / rstd [rows] fp32 are saved for the backward.            */
int tok_layernorm_fwd(const void* x, const void* shortcut, const float* row_scale, int rows_per_sample,
                      const float* gamma, const float* beta, void* out, float* mean, float* rstd,
                      int64_t rows, int c, int ld, float eps, void* stream);
int tok_layernorm_bwd_rows(int64_t rows, int c);
/* dx (+)= LN backward of (dout * row_scale); partial [2][tok_layernorm_bwd_rows][c]: dgamma rows then dbeta
 * rows, to be folded by tok_colsum_f32.                                                                          */
int tok_layernorm_bwd(const void* dout, const void* x, const float* mean, const float* rstd,
                      const float* gamma, const float* row_scale, int rows_per_sample, void* dx,
                      int accumulate, float* partial, int64_t rows, int c, int ld, void* stream);
/* column sums of a TALL bf16 matrix (bias gradients over B*H*W token rows): partial fp32
 * [tok_colsum_partial_rows(m, n_pad)][n_pad], one row per contiguous row chunk, folded by tok_colsum_f32  */
int tok_colsum_partial_rows(int64_t m, int n_pad);
int tok_colsum_partial(const void* dy, int64_t m, int n_pad, float* partial, void* stream);
/* dst[col] (+)= sum_r src[r][col]: fixed-order fp64 fold of fp32 partial rows                           */
int tok_colsum_f32(const float* src, int64_t rows, int cols, float* dst, int accumulate, void* stream);
/* two folds of the same geometry in one launch (the d(weight) / d(bias) rows of an nn.LayerNorm backward,
 * torch.nn.functional.layer_norm's grad_weight / grad_bias: swin.py blocks' norm1 / norm2)                */
int tok_colsum_f32_pair(const float* src0, const float* src1, int64_t rows, int cols, float* dst0, int accumulate0,
                        float* dst1, int accumulate1, void* stream);
/* kind 0: ReLU (cpb_mlp), 1: GELU erf (Mlp); count % 8 == 0.  tok_act_bwd also takes kind 2 = identity
 * (dx (+)= dout: the pass-through branch of a residual)                                                  */
int tok_act_fwd(int kind, const void* x, void* out, size_t count, void* stream);
int tok_act_bwd(int kind, const void* dout, const void* x, void* dx, int accumulate, size_t count, void* stream);
/* WindowAttention.forward + the roll / window_partition / window_reverse of SwinTransformerBlock._attn:
 * qkv [B*H*W][ld] (q | k | v, each c = heads*32 wide) -> out [B*H*W][c] in token order.
 * attn = normalize(q) normalize(k)^T * exp(min(logit_scale[h], ln 100)) + bias[h] (+ mask[window]), softmax, @ v.
 * bias fp32 [heads][N][N], mask fp32 [nW][N][N] or NULL, lse fp32 [B*nW*heads][N] (saved).              */
int tok_window_attn_fwd(const void* qkv, int batch, int h, int w, int c, int heads, int ws, int shift, int ld,
                        const float* logit_scale, const float* bias, const float* mask, void* out,
                        float* lse, void* stream);
/* dqkv [B*H*W][ld]; with R = tok_window_attn_bwd_rows(...): ds_scratch fp32 [R][heads][N][N] receives partial
 * sums of d(attn logits) — tok_colsum_f32 over its R rows is d(bias); dscale_part fp32 [R][heads] -> colsum per
 * head = d(logit_scale).  (Windows of <= 64 tokens run on MFMA tiles and fold several images per wave: R < B*nW.) */
int tok_window_attn_bwd_rows(int batch, int h, int w, int heads, int ws);
int tok_window_attn_bwd(const void* qkv, const void* dout, int batch, int h, int w, int c, int heads, int ws,
                        int shift, int ld, const float* logit_scale, const float* bias, const float* mask,
                        const float* lse, void* dqkv, float* ds_scratch, float* dscale_part, void* stream);
/* bias[h][i][j] = 16 * sigmoid(table[index[i][j]][h]); table bf16 [T][ld] = cpb_mlp(relative_coords_table),
 * index int64 [N][N] = relative_position_index (exact gather); backward = its transpose, ordered.        */
int tok_cpb_bias_fwd(const void* table, int ld, const int64_t* index, int heads, int n_tokens, float* bias,
                     void* stream);
int tok_cpb_bias_bwd(const float* dbias, int transposed, const void* table, int ld, const int64_t* index,
                     int heads, int n_tokens, int table_rows, void* dtable, void* stream);
/* PatchMerging gather [B][H][W][c] -> [B][H/2][W/2][4c] (x0, x1, x2, x3 order); inverse != 0: the inverse
 * permutation (its backward)                                                                             */
int tok_patch_merge(const void* src, void* dst, int batch, int h, int w, int c, int inverse, void* stream);

/* ---- self-supervised / triplet losses ---------------------------------------------------
 * NT_XentLoss (losses/representation/unsupervised.py:7-54, emb_m = None): emb = cat(emb1, emb2) bf16 [n][ld]
 * (n = 2B); logits = emb emb^T / T, diagonal -1e9, label(i) = (i + B) mod n, mean CE.  lse [n] is saved.        */
int tok_ntxent_fwd(const void* emb, int n, int d, int ld, float temperature, float* lse, float* row_loss,
                   float* loss, void* stream);
int tok_ntxent_bwd(const void* emb, const float* lse, const float* gscale, int n, int d, int ld, float temperature,
                   void* demb, void* stream);
/* torch.nn.TripletMarginLoss(p=2) registered at losses/__init__.py:39: d(x, y) = ||x - y + eps||, mean over
 * rows of relu(d_ap - d_an + margin) (swap: d_an = min(d_an, d_pn)).  dist fp32 [rows][3] is saved.             */
int tok_triplet_fwd(const void* anchor, const void* positive, const void* negative, int rows, int d, int ld,
                    float margin, float eps, int swap, float* dist, float* row_loss, float* loss, void* stream);
int tok_triplet_bwd(const void* anchor, const void* positive, const void* negative, const float* dist,
                    const float* gscale, int rows, int d, int ld, float margin, float eps, int swap,
                    void* d_anchor, void* d_positive, void* d_negative, void* stream);

/* ---- stem: BatchNorm + ReLU + max-pool in one pass (resnet.py:541-546: conv1 -> bn1 -> act1 -> maxpool) ---------------
 * tok_bn_relu_maxpool_fwd == tok_bn_act_fwd(relu) followed by tok_maxpool3x3s2_fwd without storing the activated map;
 * tok_bn_pool_bwd_reduce / _apply == tok_maxpool3x3s2_bwd followed by tok_bn_bwd_reduce / tok_bn_bwd_apply without
 * storing d(activated map): y fp bf16 [n][h][w][c] raw conv output, dpool / argmax [n][p][q][c] of the pooled map,
 * partial fp32 [2][tok_bn_bwd_rows(n*h*w, c)][c] (xhat form: tok_bn_bwd_finalize with dzy_form = 0).  Bit-identical to
 * the unfused launches.                                                                                          */
int tok_bn_relu_maxpool_fwd(const void* y, const float* scale, const float* shift, int n, int h, int w, int c,
                            void* pooled, uint8_t* argmax, void* ypool, void* stream);
/* ypool (optional, bf16 [n][p][q][c]): the raw conv output at each winning tap.  With it the backward sums can be taken in
 * the pooled domain — partial fp32 [2][tok_bn_bwd_rows(m_pooled, c)][c], 3 pooled-size reads instead of a 4-window gather
 * per input position; equal to tok_bn_pool_bwd_reduce up to the bf16 rounding of positions hit by several windows.    */
int tok_bn_pool_bwd_reduce_pooled(const void* dpool, const void* pooled, const void* ypool, const float* mean,
                                  const float* rstd, int64_t m_pooled, int c, float* partial, void* stream);
int tok_bn_pool_bwd_reduce(const void* dpool, const uint8_t* argmax, const void* y, const float* scale,
                           const float* shift, const float* mean, const float* rstd, int n, int h, int w, int c,
                           float* partial, void* stream);
int tok_bn_pool_bwd_apply(const void* dpool, const uint8_t* argmax, const void* y, const float* scale,
                          const float* shift, const float* coef, int n, int h, int w, int c, void* dy, void* stream);

/* ---- GEMM + activation (Mlp.fc1 -> GELU -> fc2 of the transformer blocks: [timm] Mlp, modules/bricks/mlp.py:37-41) ----
 * tok_conv_fwd_act: y = conv(x, w) + bias AND y_act = act(y) from one launch (the backward needs y, the next layer
 * y_act).  tok_conv_dgrad_act: dx = conv_dgrad(dy) * act'(act_x) — the dgrad of the layer AFTER the activation writes the
 * gradient of the activation's INPUT.  Pointwise layers only (1x1, stride 1, no padding); kind 0 ReLU, 1 GELU (erf).
 * Same arithmetic as tok_conv_fwd + tok_act_fwd resp. tok_conv_dgrad + tok_act_bwd on the bf16-rounded GEMM result.       */
int tok_conv_fwd_act(const tok_conv_desc* d, const void* x, const void* w, const float* bias, void* y, void* y_act,
                     int kind, void* stream);
int tok_conv_dgrad_act(const tok_conv_desc* d, const void* dy, const void* w_dgrad, const void* act_x, int kind,
                       void* dx, void* stream);

/* ---- the whole Mlp with its hidden tile on chip (csrc/mlp_fused.hip) -------------------------------------------------
 * [timm 0.6.13] models/layers/mlp.py: Mlp.forward = fc2(GELU(fc1(x))), drop = 0, as SwinTransformerBlock / DaViT's blocks
 * call it (models/backbones/swin.py:18,238; davit.py:16,196).  x, y: bf16 [rows][c]; w1 = fc1 forward pack [hidden][c],
 * w2 = fc2 forward pack [c][hidden] (tok_pack_weight_fwd), b1 [hidden] / b2 [c] fp32.  The 4c-wide hidden tensor is never
 * read back: pre / act = NULL (inference) it is never stored; with pre / act [rows][hidden] given the bf16
 * pre-activation and activation rows are written for the backward GEMMs (tok_conv_dgrad_act, the two weight gradients)
 * while fc2 consumes them out of registers; with pre given and act = NULL only the pre-activation rows are written
 * (tok_mlp_bwd_dx reads nothing else).  Rounding points
 * (pre-activation and activation to bf16) and results are those of tok_conv_fwd_act + tok_conv_fwd, bit for bit.
 * tok_mlp_serves: 1 when the geometry has a kernel (c in {96, 192, 384}, hidden = 4c), else the caller stays on the two
 * GEMM launches.                                                                                                         */
int tok_mlp_serves(int64_t rows, int c, int hidden);
int tok_mlp_fwd(const void* x, const void* w1, const float* b1, const void* w2, const float* b2, void* y, void* pre,
                void* act, int64_t rows, int c, int hidden, void* stream);
/* Backward of that Mlp to its input, one launch: d(pre) = bf16(bf16(dy W2) * GELU'(pre)), dx (+)= d(pre) W1 with d(pre) in
 * registers between the two products.  w2_dgrad = fc2's dgrad pack [hidden][c], w1_dgrad = fc1's dgrad pack [c][hidden]
 * (tok_pack_weight_both), pre = the rows tok_mlp_fwd saved, accumulate != 0: dx already holds a gradient contribution.
 * dpre [rows][hidden] != NULL: the d(pre) rows are also written (the weight gradients of fc1 read them).  Results are
 * those of tok_conv_dgrad_act + tok_conv_dgrad, bit for bit.                                                              */
int tok_mlp_bwd_dx(const void* dy, const void* w2_dgrad, const void* pre, const void* w1_dgrad, void* dx, int accumulate,
                   void* dpre, int64_t rows, int c, int hidden, void* stream);
/* ---- DaViT (models/backbones/davit.py) --------------------------------------------------------------
 * SpatialBlock's WindowAttention (davit.py:168-207) is tok_window_attn_fwd/_bwd with logit_scale == bias == NULL:
 * softmax(q k^T / sqrt(32)) v on unshifted windows, no cosine normalisation (ds_scratch / dscale_part unused).
 * ChannelAttention (davit.py:131-165): A = softmax_rows((k * scale)^T v) per (image, head) over all tokens of the
 * image, out = q A^T.  tok_chan_gram: out[u] = f(scale * X_u^T Y_u), u = image * heads + head, X / Y the 32-wide head
 * slices of bf16 token matrices (row pitches ldx / ldy, pointers already offset to the q / k / v block), fp32
 * [images*heads][32][32]; mode 0: f = id, 1: row softmax, 2: softmax backward A o (G - rowsum(G o A)) with A = a_in.
 * tok_chan_apply: out[n][h*32+i] = scale * sum_j M[u][i][j] x[n][h*32+j] (transposed != 0: M[u][j][i]).           */
int tok_chan_gram(const void* x, int ldx, const void* y, int ldy, int rows_per_image, int images, int heads,
                  float scale, int mode, const float* a_in, float* out, void* stream);
int tok_chan_apply(const void* x, int ldx, const float* m, int transposed, float scale, int rows_per_image,
                   int images, int heads, void* out, int ldo, void* stream);
/* out (+)= a + row_scale[row / rows_per_sample] * b on bf16 [rows][ld]; a and row_scale may be NULL.  The pre-norm
 * residual x + drop_path(f(norm(x))) of davit.py:262-271,330-366 and its backward (db = row_scale * dout).         */
int tok_scale_rows_add(const void* a, const void* b, const float* row_scale, int rows_per_sample, void* out,
                       int accumulate, int64_t rows, int ld, void* stream);

/* ---- object-contextual representations (heads/segmentation/ocr.py) -------------------------------------------
 * SpatialGather_Module (:23-46): p = softmax over the PIXELS of the auxiliary logits, context[k] = sum_n p[n][k] x[n];
 * ObjectAttentionBlock (:49-104): sim = softmax_k(q key^T / sqrt(C_key)), context[n] = sum_k sim[n][k] value[k].
 * Building blocks (K <= 64 classes; x / out bf16 pixel tensors [images][n][ld], m bf16 class matrices [images][k][ldm],
 * w / p fp32 [images][n][k]):
 *   tok_pix_class_matmul  out[b][n][k]  = scale * sum_c x[b][n][c] m[b][k][c]
 *   tok_class_pix_expand  out[b][n][c] (+)= scale * sum_k w[b][n][k] m[b][k][c]
 *   tok_weighted_pool     out[b][k][c] (+)= scale * sum_n w[b][n][k] x[b][n][c]   (k * c <= 10240; partial fp32
 *                         [images][tok_weighted_pool_chunks(n)][k][c], fixed-order fold)
 *   tok_softmax_rows_f32 / _bwd_f32   softmax over the k entries of a row and its backward p o (dp - <p, dp>)
 *   tok_softmax_cols_fwd / _bwd       softmax over the n pixels of every (image, class) of bf16 logits * scale
 *   tok_channel_scale     out (+)= x * s[b][c]   (Dropout2d of SpatialOCR, :121-124, and its backward)            */
int tok_pix_class_matmul(const void* x, int ldx, const void* m, int ldm, int images, int n, int k, int c, float scale,
                         float* out, void* stream);
int tok_class_pix_expand(const float* w, const void* m, int ldm, int images, int n, int k, int c, float scale, void* out,
                         int ldo, int accumulate, void* stream);
int tok_weighted_pool_chunks(int n);
int tok_weighted_pool(const float* w, const void* x, int ldx, int images, int n, int k, int c, float scale, float* partial,
                      void* out, int ldo, int accumulate, void* stream);
int tok_softmax_rows_f32(const float* x, int64_t rows, int k, float* out, void* stream);
int tok_softmax_rows_bwd_f32(const float* p, const float* dp, int64_t rows, int k, float* dx, void* stream);
int tok_softmax_cols_fwd(const void* logits, int ld, int images, int n, int k, float scale, float* p, void* stream);
int tok_softmax_cols_bwd(const float* p, const float* dp, int images, int n, int k, float scale, void* dlogits, int ld,
                         int accumulate, void* stream);
int tok_channel_scale(const void* x, const float* s, void* out, int accumulate, int images, int n, int c, int ld,
                      void* stream);

/* Depthwise 3x3 / stride 1 / pad 1 convolution with bias (ConvPosEnc.proj, davit.py:101-106; only reached with
 * cpe_act=True).  x / out bf16 NHWC [n][h][w][ld], w fp32 [c][3][3] (the master), bias fp32 [c] or NULL.  flip = 1 applies
 * the tap-flipped filter (the data gradient).  tok_dwconv3x3_wgrad: dw fp32 [c][3][3], db fp32 [c]; partial fp32
 * [tok_dwconv3x3_wgrad_blocks(n, h)][c][10], fixed-order fold.                                                        */
int tok_dwconv3x3(const void* x, const float* w, const float* bias, void* out, int accumulate, int flip, int n, int h, int wd,
                  int c, int ld, void* stream);
int tok_dwconv3x3_wgrad_blocks(int n, int h);
int tok_dwconv3x3_wgrad(const void* x, const void* dout, int n, int h, int wd, int c, int ld, float* partial, float* dw,
                        float* db, int accumulate, void* stream);

/* ---- retrieval meters (validation path) -------------------------------------------------------
 * IndexBasedMeter.compute (metrics/index_base_metric.py:170-270) with exact_index=True: the faiss flat index
 * (:523-545) is an exhaustive search = similarity matrix + k best per row; the ranx metric functions bound by
 * metrics/representation_ranx.py:56-123 are evaluated per query on the device.
 * tok_sim_matrix: out[i][j] = <q_i, g_j> (metric 0, IndexFlatIP) or -|q_i - g_j|^2 (metric 1, IndexFlatL2; negated
 * so that larger is closer), fp32 [nq][ldo] for fp32 rows q [nq][ldq], g [ng][ldg].                        */
int tok_sim_matrix(const float* q, const float* g, int nq, int ng, int d, int ldq, int ldg, int metric,
                   float* out, int64_t ldo, void* stream);
/* k best columns per row, larger value first, lower index on ties; k > cols is padded with (-inf, -1) as
 * faiss does.  vals fp32 [rows][k], idx int64 [rows][k].                                                  */
int tok_topk_rows(const float* s, int rows, int cols, int64_t ld, int k, float* vals, int64_t* idx, void* stream);
/* Number of relevant vectors per query.  labels != NULL (classification data, :379-418): vectors with the label
 * of row q_row[i], the query excluded.  Otherwise scores fp32 [n][n_cols] (representation data, :342-377): rows
 * whose entry in column q_col[i] reaches ranx's relevance level 1.                                        */
int tok_retrieval_nrel(const int64_t* labels, const float* scores, int n, int n_cols, const int64_t* q_row,
                       const int64_t* q_col, int nq, int32_t* n_rel, void* stream);
/* Per-query metric of the kk - 1 results left by clear_faiss_output (:420-444: drop the first of the kk = k + 1
 * found when drop_first[i], else the last).  kind 0 hit_rate, 1 precision, 2 recall, 3 average_precision,
 * 4 ndcg (Jarvelin, linear gains).  idx int64 [nq][kk] are rows of `gallery` (int64 [ng] -> global row; NULL =
 * identity), -1 meaning gallery[ng-1] (:503).  ideal fp32 [nq][kk-1] = descending gains of the relevant set
 * (NULL: all ones).  out fp32 [nq].                                                                        */
int tok_retrieval_eval(int kind, const int64_t* idx, int kk, const uint8_t* drop_first, const int64_t* gallery,
                       int ng, const int64_t* labels, const float* scores, int n_cols, const int64_t* q_row,
                       const int64_t* q_col, const int32_t* n_rel, const float* ideal, int nq, float* out,
                       void* stream);

/* ---- optimizers (flat arenas) -------------------------------------------------------------
 * torch.optim.SGD / Adam / AdamW registered at optim/optimizers/__init__.py:11,13,18 and
 * built by Constructor.create_optimizer (constructor/constructor.py:151-158).  One launch
 * updates `count` contiguous fp32 parameters and rewrites their bf16 shadow.               */
int tok_sgd_step(float* param, const float* grad, float* momentum_buf, void* shadow_bf16,
                 size_t count, float lr, float momentum, float dampening, float weight_decay,
                 int nesterov, int first_step, int maximize, void* stream);
int tok_adam_step(float* param, const float* grad, float* exp_avg, float* exp_avg_sq,
                  void* shadow_bf16, size_t count, float lr, float beta1, float beta2,
                  float eps, float weight_decay, int decoupled /*AdamW*/, int64_t step,
                  int maximize, void* stream);
/* torch.optim.Adam / AdamW with capturable=True: the step count is a device scalar (*step_dev = steps already taken), the bias
 * corrections are formed in the kernel; tok_step_advance increments it after the arenas of a step have been updated.  No
 * launch argument changes between steps: the optimizer step can be recorded into a hipGraph (engine/graph.py). */
int tok_adam_step_capturable(float* param, const float* grad, float* exp_avg, float* exp_avg_sq, void* shadow_bf16,
                             size_t count, float lr, float beta1, float beta2, float eps, float weight_decay,
                             int decoupled /*AdamW*/, const int64_t* step_dev, int maximize, void* stream);
int tok_step_advance(int64_t* step_dev, void* stream);
/* torch.optim.RMSprop (registered at optim/optimizers/__init__.py:16): square_avg / momentum_buf / grad_avg start at 0 */
int tok_rmsprop_step(float* param, const float* grad, float* square_avg, float* momentum_buf, float* grad_avg,
                     size_t count, float lr, float alpha, float eps, float weight_decay, float momentum,
                     int centered, int maximize, void* stream);
int tok_fill_f32(float* dst, float value, size_t count, void* stream);
int tok_scale_f32(float* dst, float factor, size_t count, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* TOK_H_ */
