// Fused softmax cross-entropy (mean reduction, ignore_index), bf16 logits, fp32 math.
// One wavefront per row: lane-strided max / sum-exp with 64-lane butterflies, no LDS.
#include "tok_common.h"
#include <stdlib.h>
#include <math.h>

namespace {

// ONE validity predicate for forward, mean denominator and backward: a row counts iff its label is a class index and is not
// `ignore_index`.  (torch raises on labels outside [0, classes) that are not ignore_index; a device-side raise would need a
// host sync per step, so such rows are dropped — consistently: zero loss, not counted in the mean, zero gradient.  The
// Python wrapper validates the labels on the host under TOK_CHECK_TARGETS=1.)
__device__ __forceinline__ bool ce_row_valid(int64_t t, int64_t ignore_index, int classes) {
  return t != ignore_index && t >= 0 && t < (int64_t)classes;
}

__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) v = fmaxf(v, __shfl_xor(v, off, 64));
  return v;
}

__global__ __launch_bounds__(256) void ce_fwd_kernel(const bf16* __restrict__ logits,
                                                     const int64_t* __restrict__ target, int rows,
                                                     int classes, int ld, int64_t ignore_index, float smooth,
                                                     float* __restrict__ lse, float* __restrict__ row_loss) {
  const int lane = threadIdx.x & 63;
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= rows) return;
  const bf16* z = logits + (size_t)row * ld;
  float mx = -INFINITY;
  for (int c = lane; c < classes; c += 64) mx = fmaxf(mx, bf2f(z[c]));
  mx = wave_max(mx);
  float s = 0.f, zs = 0.f;
  for (int c = lane; c < classes; c += 64) { const float v = bf2f(z[c]); s += expf(v - mx); zs += v; }
  s = wave_sum(s);
  const float l = mx + logf(s);
  if (smooth != 0.f) zs = wave_sum(zs);   // label smoothing: + smooth * mean_c(-log p_c) = smooth * (lse - mean_c z_c)
  if (lane == 0) {
    lse[row] = l;
    const int64_t t = target[row];
    float rl = 0.f;
    if (ce_row_valid(t, ignore_index, classes)) {
      rl = l - bf2f(z[t]);
      if (smooth != 0.f) rl = (1.f - smooth) * rl + smooth * (l - zs / (float)classes);
    }
    row_loss[row] = rl;
  }
}

// Few classes, many rows (segmentation logits: 4 M pixel rows x 19 classes): one THREAD per row, the row in
// registers through 16-byte loads (ld <= 64); a wavefront per row would leave most of its lanes idle.
__global__ __launch_bounds__(256) void ce_fwd_small_kernel(const bf16* __restrict__ logits,
                                                           const int64_t* __restrict__ target, int rows, int classes,
                                                           int ld, int64_t ignore_index, float smooth,
                                                           float* __restrict__ lse, float* __restrict__ row_loss) {
  const int row = blockIdx.x * 256 + threadIdx.x;
  if (row >= rows) return;
  const bf16* z = logits + (size_t)row * ld;
  const int nv = ld >> 3;
  float v[64];
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    if (i < nv) {
      const bf16x8 q = ldg16(z + i * 8);
#pragma unroll
      for (int e = 0; e < 8; ++e) v[i * 8 + e] = (i * 8 + e < classes) ? bf2f(q[e]) : -INFINITY;
    } else {
#pragma unroll
      for (int e = 0; e < 8; ++e) v[i * 8 + e] = -INFINITY;
    }
  }
  float mx = -INFINITY;
#pragma unroll
  for (int c = 0; c < 64; ++c) mx = fmaxf(mx, v[c]);
  float s = 0.f;
#pragma unroll
  for (int c = 0; c < 64; ++c) s += (c < classes) ? expf(v[c] - mx) : 0.f;
  const float l = mx + logf(s);
  lse[row] = l;
  const int64_t t = target[row];
  float rl = 0.f;
  if (ce_row_valid(t, ignore_index, classes)) {
    rl = l - bf2f(z[t]);
    if (smooth != 0.f) {
      float zs = 0.f;
#pragma unroll
      for (int c = 0; c < 64; ++c) zs += (c < classes) ? v[c] : 0.f;
      rl = (1.f - smooth) * rl + smooth * (l - zs / (float)classes);
    }
  }
  row_loss[row] = rl;
}

__global__ __launch_bounds__(256) void ce_bwd_small_kernel(const bf16* __restrict__ logits,
                                                           const int64_t* __restrict__ target,
                                                           const float* __restrict__ lse, const float* __restrict__ loss,
                                                           const float* __restrict__ gscale, int rows, int classes, int ld,
                                                           int64_t ignore_index, float smooth, bf16* __restrict__ dlogits) {
  const int row = blockIdx.x * 256 + threadIdx.x;
  if (row >= rows) return;
  const int64_t t = target[row];
  const bool valid = ce_row_valid(t, ignore_index, classes);
  const float g = valid ? (gscale ? gscale[0] : 1.f) / loss[1] : 0.f;
  const float l = lse[row];
  const bf16* z = logits + (size_t)row * ld;
  bf16* d = dlogits + (size_t)row * ld;
  const int nv = ld >> 3;
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    if (i < nv) {
      const bf16x8 q = ldg16(z + i * 8);
      bf16x8 o;
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        const int c = i * 8 + e;
        float w = 0.f;
        if (valid && c < classes) w = (expf(bf2f(q[e]) - l) - (c == t ? 1.f - smooth : 0.f) - smooth / (float)classes) * g;
        o[e] = f2bf(w);
      }
      stg16(d + i * 8, o);
    }
  }
}

// deterministic mean over valid rows in two fixed-order stages: up to 512 blocks each reduce one contiguous
// chunk into a (sum, count) double pair kept in the caller's loss buffer, then one block folds the pairs
constexpr int CE_PARTS = (TOK_CE_LOSS_FLOATS - 2) / 4;   // (sum, count) doubles

__device__ __forceinline__ void block_reduce2(double& s, double& c) {
  __shared__ double rs[256];
  __shared__ double rc[256];
  rs[threadIdx.x] = s;
  rc[threadIdx.x] = c;
  __syncthreads();
  for (int k = 128; k > 0; k >>= 1) {
    if (threadIdx.x < k) { rs[threadIdx.x] += rs[threadIdx.x + k]; rc[threadIdx.x] += rc[threadIdx.x + k]; }
    __syncthreads();
  }
  s = rs[0];
  c = rc[0];
}

__global__ __launch_bounds__(256) void ce_partial_kernel(const float* __restrict__ row_loss,
                                                         const int64_t* __restrict__ target, int rows,
                                                         int64_t ignore_index, int classes, int chunk,
                                                         double* __restrict__ part) {
  const int r0 = blockIdx.x * chunk;
  const int r1 = min(rows, r0 + chunk);
  double s = 0.0, cnt = 0.0;
  for (int r = r0 + threadIdx.x; r < r1; r += 256) {
    if (ce_row_valid(target[r], ignore_index, classes)) { s += (double)row_loss[r]; cnt += 1.0; }
  }
  block_reduce2(s, cnt);
  if (threadIdx.x == 0) { part[2 * blockIdx.x] = s; part[2 * blockIdx.x + 1] = cnt; }
}

__global__ __launch_bounds__(256) void ce_mean_kernel(const double* __restrict__ part, int nparts, float* loss) {
  double s = 0.0, cnt = 0.0;
  for (int i = threadIdx.x; i < nparts; i += 256) { s += part[2 * i]; cnt += part[2 * i + 1]; }
  block_reduce2(s, cnt);
  if (threadIdx.x == 0) {
    loss[0] = (float)(s / cnt);  // 0/0 = nan, as torch
    loss[1] = (float)cnt;
  }
}

__global__ __launch_bounds__(256) void ce_bwd_kernel(const bf16* __restrict__ logits,
                                                     const int64_t* __restrict__ target,
                                                     const float* __restrict__ lse,
                                                     const float* __restrict__ loss,
                                                     const float* __restrict__ gscale, int rows, int classes,
                                                     int ld, int64_t ignore_index, float smooth,
                                                     bf16* __restrict__ dlogits) {
  const int lane = threadIdx.x & 63;
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= rows) return;
  const int64_t t = target[row];
  const bool valid = ce_row_valid(t, ignore_index, classes);
  const float g = valid ? (gscale ? gscale[0] : 1.f) / loss[1] : 0.f;
  const float l = lse[row];
  const bf16* z = logits + (size_t)row * ld;
  bf16* d = dlogits + (size_t)row * ld;
  for (int c = lane; c < ld; c += 64) {
    float v = 0.f;
    if (valid && c < classes) v = (expf(bf2f(z[c]) - l) - (c == t ? 1.f - smooth : 0.f) - smooth / (float)classes) * g;
    d[c] = f2bf(v);
  }
}

}  // namespace

namespace {

// ---- Dice loss (losses/segmentation/dice.py:86-188) ------------------------------------------------------------------
// mode 0 (multiclass): p = softmax(logits) per pixel row, y = one_hot(target);  mode 1 (binary): p = sigmoid(logit 0),
// y = target (float);  mode 2 (multilabel): p = sigmoid(logit c), y = target[row][c] (float).  Per class: I = sum p*y, P = sum p, Y = sum y.  One wave per pixel row, lane = class; every lane
// keeps its class's three sums in registers over all the rows of its wave; waves -> block partial row (fixed order).
__global__ __launch_bounds__(256) void dice_fwd_kernel(const bf16* __restrict__ logits, const void* __restrict__ target,
                                                       int64_t rows, int classes, int ld, int mode,
                                                       float* __restrict__ partial) {
  __shared__ float red[4][3][64];
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  float si = 0.f, sp = 0.f, sy = 0.f;
  for (int64_t row = (int64_t)blockIdx.x * 4 + wv; row < rows; row += (int64_t)gridDim.x * 4) {
    const bf16* z = logits + row * ld;
    float p = 0.f, y = 0.f;
    if (mode == 0) {
      const float v = lane < classes ? bf2f(z[lane]) : -INFINITY;
      const float mx = wave_max(v);
      const float e = lane < classes ? expf(v - mx) : 0.f;
      const float den = wave_sum(e);
      p = e / den;
      y = (lane < classes && ((const int64_t*)target)[row] == lane) ? 1.f : 0.f;
    } else if (mode == 2) {            // multilabel: a sigmoid per class, dense fp32 targets [rows][classes]
      if (lane < classes) {
        p = 1.f / (1.f + expf(-bf2f(z[lane])));
        y = ((const float*)target)[row * classes + lane];
      }
    } else if (lane == 0) {
      p = 1.f / (1.f + expf(-bf2f(z[0])));
      y = ((const float*)target)[row];
    }
    si = fmaf(p, y, si);
    sp += p;
    sy += y;
  }
  red[wv][0][lane] = si; red[wv][1][lane] = sp; red[wv][2][lane] = sy;
  __syncthreads();
  if (threadIdx.x < 3 * 64) {
    const int which = threadIdx.x / 64, c = threadIdx.x & 63;
    if (c < classes)
      partial[((size_t)blockIdx.x * 3 + which) * classes + c] =
          red[0][which][c] + red[1][which][c] + red[2][which][c] + red[3][which][c];
  }
}

// folds the partial rows, evaluates the loss and the two per-class coefficients of its gradient:
//   d loss / d p[row][c] = coef[0][c] * y[row][c] + coef[1][c]
__global__ __launch_bounds__(64) void dice_finalize_kernel(const float* __restrict__ partial, int nparts, int classes,
                                                           float smooth, float eps, int log_loss,
                                                           const int64_t* __restrict__ sel, int nsel,
                                                           float* __restrict__ loss, float* __restrict__ coef) {
  const int c = threadIdx.x;
  double I = 0.0, P = 0.0, Y = 0.0;
  if (c < classes)
    for (int r = 0; r < nparts; ++r) {
      I += (double)partial[((size_t)r * 3 + 0) * classes + c];
      P += (double)partial[((size_t)r * 3 + 1) * classes + c];
      Y += (double)partial[((size_t)r * 3 + 2) * classes + c];
    }
  float l = 0.f, a = 0.f, b = 0.f;
  bool counted = c < classes;
  if (counted && sel != nullptr) {
    counted = false;
    for (int i = 0; i < nsel; ++i) counted = counted || sel[i] == c;
  }
  const int ncount = sel != nullptr ? nsel : classes;
  if (counted && Y > 0.0) {        // classes without true pixels contribute zero (dice.py:180-181)
    const float card = (float)(P + Y);
    const float cc = fmaxf(card, eps);
    const float num = 2.f * (float)I + smooth, den = cc + smooth;
    const float score = num / den;
    // d score / dI = 2 / den ;  d score / d card = -num / den^2 (zero where the clamp is active)
    const float ds_dI = 2.f / den, ds_dc = card > eps ? -num / (den * den) : 0.f;
    float dl_ds;
    if (log_loss) { l = -logf(fmaxf(score, eps)); dl_ds = score > eps ? -1.f / score : 0.f; }
    else { l = 1.f - score; dl_ds = -1.f; }
    a = dl_ds * ds_dI / (float)ncount;
    b = dl_ds * ds_dc / (float)ncount;
  }
  if (c < classes) { coef[c] = a; coef[classes + c] = b; }
  const float tot = wave_sum(counted ? l : 0.f);
  if (c == 0) loss[0] = tot / (float)ncount;
}

__global__ __launch_bounds__(256) void dice_bwd_kernel(const bf16* __restrict__ logits, const void* __restrict__ target,
                                                       const float* __restrict__ coef, const float* __restrict__ gscale,
                                                       int64_t rows, int classes, int ld, int mode,
                                                       bf16* __restrict__ dlogits) {
  const int lane = threadIdx.x & 63;
  const int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= rows) return;
  const float g = gscale ? gscale[0] : 1.f;
  const bf16* z = logits + row * ld;
  bf16* d = dlogits + row * ld;
  if (mode == 0) {
    const float v = lane < classes ? bf2f(z[lane]) : -INFINITY;
    const float mx = wave_max(v);
    const float e = lane < classes ? expf(v - mx) : 0.f;
    const float p = e / wave_sum(e);
    float dp = 0.f;
    if (lane < classes) dp = (((const int64_t*)target)[row] == lane ? coef[lane] : 0.f) + coef[classes + lane];
    const float dot = wave_sum(p * dp);
    if (lane < ld) d[lane] = f2bf(lane < classes ? p * (dp - dot) * g : 0.f);
  } else if (mode == 2) {
    if (lane < classes) {
      const float p = 1.f / (1.f + expf(-bf2f(z[lane])));
      const float dp = coef[lane] * ((const float*)target)[row * classes + lane] + coef[classes + lane];
      d[lane] = f2bf(p * (1.f - p) * dp * g);
    } else if (lane < ld) {
      d[lane] = f2bf(0.f);
    }
  } else {
    if (lane == 0) {
      const float p = 1.f / (1.f + expf(-bf2f(z[0])));
      const float dp = coef[0] * ((const float*)target)[row] + coef[1];
      d[0] = f2bf(p * (1.f - p) * dp * g);
    } else if (lane < ld) {
      d[lane] = f2bf(0.f);
    }
  }
}

}  // namespace

extern "C" int tok_dice_rows(int64_t rows) {
  const int64_t b = (rows + 3) / 4;
  return (int)(b > 2048 ? 2048 : (b < 1 ? 1 : b));
}

extern "C" int tok_dice_fwd(const void* logits, const void* target, int64_t rows, int classes, int ld, int mode,
                            float smooth, float eps, int log_loss, const int64_t* class_sel, int n_sel, float* partial,
                            float* loss, float* coef, void* stream) {
  TOK_CHECK_ARG(logits && target && partial && loss && coef && rows > 0 && classes > 0 && classes <= 64 && ld >= classes &&
                ld <= 64 && (mode == 0 || mode == 2 || (mode == 1 && classes == 1)), "tok_dice_fwd: bad args (<= 64 classes)");
  TOK_CHECK_ARG(!class_sel || n_sel > 0, "tok_dice_fwd: empty class selection");
  hipStream_t st = tok_stream(stream);
  const int g = tok_dice_rows(rows);
  hipLaunchKernelGGL(dice_fwd_kernel, dim3(g), dim3(256), 0, st, (const bf16*)logits, target, rows, classes, ld, mode,
                     partial);
  TOK_CHECK_LAUNCH("tok_dice_fwd");
  hipLaunchKernelGGL(dice_finalize_kernel, dim3(1), dim3(64), 0, st, partial, g, classes, smooth, eps, log_loss, class_sel,
                     n_sel, loss, coef);
  TOK_CHECK_LAUNCH("tok_dice_fwd(finalize)");
  return TOK_OK;
}

extern "C" int tok_dice_bwd(const void* logits, const void* target, const float* coef, const float* gscale, int64_t rows,
                            int classes, int ld, int mode, void* dlogits, void* stream) {
  TOK_CHECK_ARG(logits && target && coef && dlogits && rows > 0 && classes > 0 && classes <= 64 && ld >= classes && ld <= 64,
                "tok_dice_bwd: bad args");
  hipLaunchKernelGGL(dice_bwd_kernel, dim3((unsigned)((rows + 3) / 4)), dim3(256), 0, tok_stream(stream),
                     (const bf16*)logits, target, coef, gscale, rows, classes, ld, mode, (bf16*)dlogits);
  TOK_CHECK_LAUNCH("tok_dice_bwd");
  return TOK_OK;
}

// thread-per-row kernels: rows of at most 64 bf16 in whole 16-byte vectors, enough rows to fill the chip
static inline bool ce_small(const void* logits, int rows, int ld) {
  return ld <= 64 && (ld & 7) == 0 && rows >= 16384 && (reinterpret_cast<uintptr_t>(logits) & 15) == 0;
}

extern "C" int tok_softmax_ce_smooth_fwd(const void* logits, const int64_t* target, int rows, int classes, int ld,
                                         int64_t ignore_index, float label_smoothing, float* lse, float* row_loss,
                                         float* loss, void* stream) {
  TOK_CHECK_ARG(logits && target && lse && row_loss && loss, "tok_softmax_ce_fwd: null pointer");
  TOK_CHECK_ARG(rows > 0 && classes > 0 && ld >= classes, "tok_softmax_ce_fwd: bad sizes");
  TOK_CHECK_ARG(label_smoothing >= 0.f && label_smoothing <= 1.f, "tok_softmax_ce_fwd: label_smoothing outside [0, 1]");
  hipStream_t st = tok_stream(stream);
  if (ce_small(logits, rows, ld))
    hipLaunchKernelGGL(ce_fwd_small_kernel, dim3((rows + 255) / 256), dim3(256), 0, st, (const bf16*)logits, target, rows,
                       classes, ld, ignore_index, label_smoothing, lse, row_loss);
  else
    hipLaunchKernelGGL(ce_fwd_kernel, dim3((rows + 3) / 4), dim3(256), 0, st, (const bf16*)logits, target, rows,
                       classes, ld, ignore_index, label_smoothing, lse, row_loss);
  TOK_CHECK_LAUNCH("tok_softmax_ce_fwd");
  double* part = reinterpret_cast<double*>(loss + 2);   // loss holds TOK_CE_LOSS_FLOATS floats, 8-byte aligned
  TOK_CHECK_ARG((reinterpret_cast<uintptr_t>(loss) & 7) == 0, "tok_softmax_ce_fwd: loss must be 8-byte aligned");
  const int nparts = rows < 4096 ? 1 : (tok_cdiv(rows, 4096) < CE_PARTS ? tok_cdiv(rows, 4096) : CE_PARTS);
  const int chunk = tok_cdiv(rows, nparts);
  hipLaunchKernelGGL(ce_partial_kernel, dim3(nparts), dim3(256), 0, st, row_loss, target, rows, ignore_index, classes,
                     chunk, part);
  TOK_CHECK_LAUNCH("tok_softmax_ce_fwd(partial)");
  hipLaunchKernelGGL(ce_mean_kernel, dim3(1), dim3(256), 0, st, part, nparts, loss);
  TOK_CHECK_LAUNCH("tok_softmax_ce_fwd(mean)");
  return TOK_OK;
}

extern "C" int tok_softmax_ce_fwd(const void* logits, const int64_t* target, int rows, int classes, int ld,
                                  int64_t ignore_index, float* lse, float* row_loss, float* loss, void* stream) {
  return tok_softmax_ce_smooth_fwd(logits, target, rows, classes, ld, ignore_index, 0.f, lse, row_loss, loss, stream);
}

extern "C" int tok_softmax_ce_smooth_bwd(const void* logits, const int64_t* target, const float* lse,
                                         const float* loss, const float* gscale, int rows, int classes, int ld,
                                         int64_t ignore_index, float label_smoothing, void* dlogits, void* stream) {
  TOK_CHECK_ARG(logits && target && lse && loss && dlogits, "tok_softmax_ce_bwd: null pointer");
  TOK_CHECK_ARG(rows > 0 && classes > 0 && ld >= classes, "tok_softmax_ce_bwd: bad sizes");
  if (ce_small(logits, rows, ld) && (reinterpret_cast<uintptr_t>(dlogits) & 15) == 0)
    hipLaunchKernelGGL(ce_bwd_small_kernel, dim3((rows + 255) / 256), dim3(256), 0, tok_stream(stream),
                       (const bf16*)logits, target, lse, loss, gscale, rows, classes, ld, ignore_index, label_smoothing,
                       (bf16*)dlogits);
  else
    hipLaunchKernelGGL(ce_bwd_kernel, dim3((rows + 3) / 4), dim3(256), 0, tok_stream(stream),
                       (const bf16*)logits, target, lse, loss, gscale, rows, classes, ld, ignore_index, label_smoothing,
                       (bf16*)dlogits);
  TOK_CHECK_LAUNCH("tok_softmax_ce_bwd");
  return TOK_OK;
}

extern "C" int tok_softmax_ce_bwd(const void* logits, const int64_t* target, const float* lse,
                                  const float* loss, const float* gscale, int rows, int classes, int ld,
                                  int64_t ignore_index, void* dlogits, void* stream) {
  return tok_softmax_ce_smooth_bwd(logits, target, lse, loss, gscale, rows, classes, ld, ignore_index, 0.f, dlogits, stream);
}

// ---- pixel-wise cross entropy on bilinearly upsampled logits, without the upsampled tensor -------------------------------------
// SegmentationHead.forward (/root/reference/torchok/models/heads/segmentation/base.py:31-41) ends in
// F.interpolate(segm_logits, size=input.shape[2:], mode='bilinear') and the recipe's loss is CrossEntropyLoss on the result
// (examples/configs/segmentation_*.yaml): at HRNet-W48 / 512x1024 / batch 24 the (B, 19 -> 24, 512, 1024) bf16 tensor is
// 604 MB, written by tok_bilinear_fwd, read by the loss, and its gradient written and read once more by tok_bilinear_bwd.
// Here every full-resolution pixel evaluates its 2x2 footprint of the LOW-resolution logits on the fly (same source-index
// formula and the same bf16 rounding of the interpolated value as tok_bilinear_fwd: the loss is the unfused one), and the
// backward is the adjoint in gather form (tok_bilinear_bwd's walk): a low-resolution pixel visits the full-resolution pixels
// whose footprint contains it, recomputes their softmax from the saved log-sum-exp and sums weight * bf16(d logits) —
// deterministic, no atomics, nothing full-resolution in HBM except lse / row_loss (4 bytes per pixel each).
namespace {
// ATen area_pixel_compute_source_index(scale, dst, align_corners=false, cubic=false) — as in resample.hip
__device__ __forceinline__ void up_src_index(float scale, int dst, int in_size, int& i0, int& i1, float& l1) {
  float s = scale * ((float)dst + 0.5f) - 0.5f;
  s = s < 0.f ? 0.f : s;
  i0 = (int)s;
  i0 = i0 > in_size - 1 ? in_size - 1 : i0;
  i1 = i0 + (i0 < in_size - 1 ? 1 : 0);
  l1 = s - (float)i0;
}

struct UpArgs {
  int n, hs, ws, classes, ld, hd, wd;
  float sh, sw;     // in / out
};

// interpolated, bf16-rounded logits of full-resolution pixel (b, y, x): v[0 .. 8 NV)
template <int NV>
__device__ __forceinline__ void up_logits(const bf16* __restrict__ low, const UpArgs& a, int b, int y, int x, float* v) {
  int y0, y1, x0, x1;
  float ly, lx;
  up_src_index(a.sh, y, a.hs, y0, y1, ly);
  up_src_index(a.sw, x, a.ws, x0, x1, lx);
  const float hy = 1.f - ly, hx = 1.f - lx;
  const bf16* base = low + (size_t)b * a.hs * a.ws * a.ld;
  const bf16* p00 = base + ((size_t)y0 * a.ws + x0) * a.ld;
  const bf16* p01 = base + ((size_t)y0 * a.ws + x1) * a.ld;
  const bf16* p10 = base + ((size_t)y1 * a.ws + x0) * a.ld;
  const bf16* p11 = base + ((size_t)y1 * a.ws + x1) * a.ld;
#pragma unroll
  for (int i = 0; i < NV; ++i) {
    const bf16x8 v00 = ldg16(p00 + i * 8), v01 = ldg16(p01 + i * 8), v10 = ldg16(p10 + i * 8), v11 = ldg16(p11 + i * 8);
#pragma unroll
    for (int e = 0; e < 8; ++e)
      v[i * 8 + e] = bf2f(f2bf(hy * (hx * bf2f(v00[e]) + lx * bf2f(v01[e])) + ly * (hx * bf2f(v10[e]) + lx * bf2f(v11[e]))));
  }
}

template <int NV>
__global__ __launch_bounds__(256) void upce_fwd_kernel(const bf16* __restrict__ low, const int64_t* __restrict__ target, UpArgs a,
                                                       int64_t ignore_index, float* __restrict__ lse,
                                                       float* __restrict__ row_loss) {
  const size_t total = (size_t)a.n * a.hd * a.wd;
  for (size_t pix = (size_t)blockIdx.x * 256 + threadIdx.x; pix < total; pix += (size_t)gridDim.x * 256) {
    const int x = (int)(pix % a.wd);
    const size_t t2 = pix / a.wd;
    const int y = (int)(t2 % a.hd);
    const int b = (int)(t2 / a.hd);
    float v[8 * NV];
    up_logits<NV>(low, a, b, y, x, v);
    float mx = -INFINITY;
#pragma unroll
    for (int c = 0; c < 8 * NV; ++c) mx = fmaxf(mx, c < a.classes ? v[c] : -INFINITY);
    float s = 0.f;
#pragma unroll
    for (int c = 0; c < 8 * NV; ++c) s += (c < a.classes) ? __expf(v[c] - mx) : 0.f;
    const float l = mx + __logf(s);
    lse[pix] = l;
    const int64_t t = target[pix];
    float rl = 0.f;
    if (ce_row_valid(t, ignore_index, a.classes)) {
      float zt = 0.f;
#pragma unroll
      for (int c = 0; c < 8 * NV; ++c) zt = (c == (int)t) ? v[c] : zt;
      rl = l - zt;
    }
    row_loss[pix] = rl;
  }
}

template <int NV>
__global__ __launch_bounds__(256) void upce_bwd_kernel(const bf16* __restrict__ low, const int64_t* __restrict__ target, UpArgs a,
                                                       int64_t ignore_index, const float* __restrict__ lse,
                                                       const float* __restrict__ loss, const float* __restrict__ gscale,
                                                       bf16* dlow, int accumulate) {
  const size_t total = (size_t)a.n * a.hs * a.ws;
  const float rh = 1.f / a.sh, rw = 1.f / a.sw;
  const float g = (gscale ? gscale[0] : 1.f) / loss[1];
  for (size_t pix = (size_t)blockIdx.x * 256 + threadIdx.x; pix < total; pix += (size_t)gridDim.x * 256) {
    const int xs = (int)(pix % a.ws);
    const size_t t2 = pix / a.ws;
    const int ys = (int)(t2 % a.hs);
    const int b = (int)(t2 / a.hs);
    // full-resolution rows / columns whose source coordinate lies in (ys - 1, ys + 1), widened by one for rounding;
    // source row 0 also owns the clamp (tok_bilinear_bwd's window)
    int yd0 = ys == 0 ? 0 : (int)floorf(((float)ys - 0.5f) * rh - 0.5f) - 1;
    int yd1 = (int)ceilf(((float)ys + 1.5f) * rh - 0.5f) + 1;
    int xd0 = xs == 0 ? 0 : (int)floorf(((float)xs - 0.5f) * rw - 0.5f) - 1;
    int xd1 = (int)ceilf(((float)xs + 1.5f) * rw - 0.5f) + 1;
    yd0 = yd0 < 0 ? 0 : yd0;  xd0 = xd0 < 0 ? 0 : xd0;
    yd1 = yd1 > a.hd - 1 ? a.hd - 1 : yd1;  xd1 = xd1 > a.wd - 1 ? a.wd - 1 : xd1;
    float acc[8 * NV];
#pragma unroll
    for (int c = 0; c < 8 * NV; ++c) acc[c] = 0.f;
    for (int yd = yd0; yd <= yd1; ++yd) {
      int y0, y1; float ly;
      up_src_index(a.sh, yd, a.hs, y0, y1, ly);
      const float wy = (y0 == ys ? 1.f - ly : 0.f) + (y1 == ys ? ly : 0.f);
      if (wy == 0.f) continue;
      for (int xd = xd0; xd <= xd1; ++xd) {
        int x0, x1; float lx;
        up_src_index(a.sw, xd, a.ws, x0, x1, lx);
        const float wx = (x0 == xs ? 1.f - lx : 0.f) + (x1 == xs ? lx : 0.f);
        if (wx == 0.f) continue;
        const size_t dp = ((size_t)b * a.hd + yd) * a.wd + xd;
        const int64_t t = target[dp];
        if (!ce_row_valid(t, ignore_index, a.classes)) continue;
        float v[8 * NV];
        up_logits<NV>(low, a, b, yd, xd, v);
        const float l = lse[dp];
        const float wgt = wy * wx;
#pragma unroll
        for (int c = 0; c < 8 * NV; ++c) {
          // d(upsampled logits), rounded to bf16 as the unfused tok_softmax_ce_bwd stores it
          const float w = c < a.classes ? bf2f(f2bf((__expf(v[c] - l) - (c == (int)t ? 1.f : 0.f)) * g)) : 0.f;
          acc[c] = fmaf(wgt, w, acc[c]);
        }
      }
    }
    bf16* d = dlow + pix * a.ld;
#pragma unroll
    for (int i = 0; i < NV; ++i) {
      bf16x8 o;
      if (accumulate) {
        const bf16x8 prev = ldg16(d + i * 8);
#pragma unroll
        for (int e = 0; e < 8; ++e) o[e] = f2bf(acc[i * 8 + e] + bf2f(prev[e]));
      } else {
#pragma unroll
        for (int e = 0; e < 8; ++e) o[e] = f2bf(acc[i * 8 + e]);
      }
      stg16(d + i * 8, o);
    }
  }
}

// The same adjoint with every full-resolution pixel evaluated ONCE per block instead of once per low-resolution pixel it
// touches (4x for a scale of 4: the gather kernel above spends its time in 19 expf per visit): a block owns UT x UT
// low-resolution pixels, stages bf16(d upsampled logits) of the full-resolution window that can touch them in LDS (phase 1),
// then every (low-resolution pixel, 8-channel group) walks its footprint with tok_bilinear_bwd's exact index / weight
// arithmetic, reading LDS (phase 2).  Window = the union of the per-pixel windows of the gather kernel, at most UWIN x UWIN.
constexpr int UT = 8, UWIN = 40, UPP = 12;     // tile edge, window edge, window of ONE source pixel (2 r + 4 at r <= 4)

__device__ __forceinline__ void up_window(float r, int s, int n_dst, int& d0, int& d1) {
  d0 = s == 0 ? 0 : (int)floorf(((float)s - 0.5f) * r - 0.5f) - 1;
  d1 = (int)ceilf(((float)s + 1.5f) * r - 0.5f) + 1;
  d0 = d0 < 0 ? 0 : d0;
  d1 = d1 > n_dst - 1 ? n_dst - 1 : d1;
}

template <int NV>
__global__ __launch_bounds__(256) void upce_bwd_tiled_kernel(const bf16* __restrict__ low, const int64_t* __restrict__ target,
                                                             UpArgs a, int64_t ignore_index, const float* __restrict__ lse,
                                                             const float* __restrict__ loss, const float* __restrict__ gscale,
                                                             bf16* dlow, int accumulate, int tiles_y, int tiles_x) {
  extern __shared__ __attribute__((aligned(16))) char up_smem[];
  bf16* win = reinterpret_cast<bf16*>(up_smem);                 // [wh][ww][8 NV]
  const int tile = blockIdx.x;
  const int tx = tile % tiles_x;
  const int t2 = tile / tiles_x;
  const int ty = t2 % tiles_y;
  const int b = t2 / tiles_y;
  const int ys0 = ty * UT, xs0 = tx * UT;
  const int ys1 = min(a.hs, ys0 + UT) - 1, xs1 = min(a.ws, xs0 + UT) - 1;
  const float rh = 1.f / a.sh, rw = 1.f / a.sw;
  int wy0, wy1, wx0, wx1, tmp0, tmp1;
  up_window(rh, ys0, a.hd, wy0, tmp1);
  up_window(rh, ys1, a.hd, tmp0, wy1);
  up_window(rw, xs0, a.wd, wx0, tmp1);
  up_window(rw, xs1, a.wd, tmp0, wx1);
  const int wh = wy1 - wy0 + 1, ww = wx1 - wx0 + 1;             // <= UWIN (checked by the launcher for the scale)
  const float g = (gscale ? gscale[0] : 1.f) / loss[1];
  // ---- phase 1: bf16(d upsampled logits) of the window ----------------------------------------------------------------------
  for (int i = threadIdx.x; i < wh * ww; i += 256) {
    const int yd = wy0 + i / ww, xd = wx0 + i % ww;
    const size_t dp = ((size_t)b * a.hd + yd) * a.wd + xd;
    const int64_t t = target[dp];
    bf16* o = win + (size_t)i * (8 * NV);
    if (!ce_row_valid(t, ignore_index, a.classes)) {
#pragma unroll
      for (int v8 = 0; v8 < NV; ++v8) *reinterpret_cast<bf16x8*>(o + v8 * 8) = zero8();
      continue;
    }
    float v[8 * NV];
    up_logits<NV>(low, a, b, yd, xd, v);
    const float l = lse[dp];
#pragma unroll
    for (int v8 = 0; v8 < NV; ++v8) {
      bf16x8 q;
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        const int c = v8 * 8 + e;
        q[e] = f2bf(c < a.classes ? (__expf(v[c] - l) - (c == (int)t ? 1.f : 0.f)) * g : 0.f);
      }
      *reinterpret_cast<bf16x8*>(o + v8 * 8) = q;
    }
  }
  __syncthreads();
  // ---- phase 2: the adjoint of the interpolation, one (low-resolution pixel, 8-channel group) per thread ------------------------
  for (int i = threadIdx.x; i < UT * UT * NV; i += 256) {
    const int v8 = i % NV;
    const int p = i / NV;
    const int ys = ys0 + p / UT, xs = xs0 + p % UT;
    if (ys > ys1 || xs > xs1) continue;
    int yd0, yd1, xd0, xd1;
    up_window(rh, ys, a.hd, yd0, yd1);
    up_window(rw, xs, a.wd, xd0, xd1);
    float acc[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) acc[e] = 0.f;
    // the column weights of this pixel once (the window of one source pixel is at most UPP wide at the scales the tiled
    // kernel serves), then rows x columns of LDS reads
    float wxs[UPP];
#pragma unroll
    for (int j = 0; j < UPP; ++j) {
      const int xd = xd0 + j;
      int x0, x1; float lx;
      up_src_index(a.sw, xd <= xd1 ? xd : xd1, a.ws, x0, x1, lx);
      wxs[j] = xd <= xd1 ? (x0 == xs ? 1.f - lx : 0.f) + (x1 == xs ? lx : 0.f) : 0.f;
    }
    for (int yd = yd0; yd <= yd1; ++yd) {
      int y0, y1; float ly;
      up_src_index(a.sh, yd, a.hs, y0, y1, ly);
      const float wy = (y0 == ys ? 1.f - ly : 0.f) + (y1 == ys ? ly : 0.f);
      if (wy == 0.f) continue;
      const bf16* rowp = win + ((size_t)(yd - wy0) * ww + (xd0 - wx0)) * (8 * NV) + v8 * 8;
#pragma unroll
      for (int j = 0; j < UPP; ++j) {
        if (wxs[j] == 0.f) continue;
        const bf16x8 q = *reinterpret_cast<const bf16x8*>(rowp + (size_t)j * (8 * NV));
        const float wgt = wy * wxs[j];
#pragma unroll
        for (int e = 0; e < 8; ++e) acc[e] = fmaf(wgt, bf2f(q[e]), acc[e]);
      }
    }
    bf16* d = dlow + (((size_t)b * a.hs + ys) * a.ws + xs) * a.ld + v8 * 8;
    bf16x8 o;
    if (accumulate) {
      const bf16x8 prev = ldg16(d);
#pragma unroll
      for (int e = 0; e < 8; ++e) o[e] = f2bf(acc[e] + bf2f(prev[e]));
    } else {
#pragma unroll
      for (int e = 0; e < 8; ++e) o[e] = f2bf(acc[e]);
    }
    stg16(d, o);
  }
}

// window of UT source pixels at destination/source ratio r: (UT - 1) r between first and last pixel + (2 r + 4) per pixel
bool up_tiled_ok(const UpArgs& a) {
  static const int flag = [] { const char* e = getenv("TOK_UPCE_TILED"); return (int)(e ? atoi(e) : 1); }();
  const float r = fmaxf(1.f / a.sh, 1.f / a.sw);
  return flag && (int)ceilf((UT + 1) * r) + 4 <= UWIN;
}

bool fill_up(UpArgs& a, int n, int hs, int ws, int classes, int ld, int hd, int wd) {
  if (n <= 0 || hs <= 0 || ws <= 0 || hd <= 0 || wd <= 0 || classes <= 0 || ld < classes || (ld & 7) || ld > 32) return false;
  if ((long long)n * hd * wd >= (1ll << 31)) return false;
  a.n = n; a.hs = hs; a.ws = ws; a.classes = classes; a.ld = ld; a.hd = hd; a.wd = wd;
  a.sh = (float)hs / (float)hd; a.sw = (float)ws / (float)wd;
  return true;
}
int up_blocks(size_t work) {
  const size_t b = (work + 255) / 256;
  return (int)(b > 65536 ? 65536 : b);
}
}  // namespace

extern "C" int tok_upsample_ce_serves(int classes, int ld) { return classes > 0 && ld >= classes && (ld & 7) == 0 && ld <= 32; }

extern "C" int tok_upsample_ce_fwd(const void* low, int n, int hs, int ws, int classes, int ld, int hd, int wd,
                                   const int64_t* target, int64_t ignore_index, float* lse, float* row_loss, float* loss,
                                   void* stream) {
  UpArgs a;
  TOK_CHECK_ARG(low && target && lse && row_loss && loss && fill_up(a, n, hs, ws, classes, ld, hd, wd),
                "tok_upsample_ce_fwd: bad args (row pitch a multiple of 8 up to 32 channels, n*hd*wd < 2^31)");
  TOK_CHECK_ARG((reinterpret_cast<uintptr_t>(low) & 15) == 0 && (reinterpret_cast<uintptr_t>(loss) & 7) == 0,
                "tok_upsample_ce_fwd: low must be 16-byte, loss 8-byte aligned");
  hipStream_t st = tok_stream(stream);
  const int rows = n * hd * wd;
  const int grid = up_blocks((size_t)rows);
  switch (ld >> 3) {
    case 1: hipLaunchKernelGGL(upce_fwd_kernel<1>, dim3(grid), dim3(256), 0, st, (const bf16*)low, target, a, ignore_index, lse, row_loss); break;
    case 2: hipLaunchKernelGGL(upce_fwd_kernel<2>, dim3(grid), dim3(256), 0, st, (const bf16*)low, target, a, ignore_index, lse, row_loss); break;
    case 3: hipLaunchKernelGGL(upce_fwd_kernel<3>, dim3(grid), dim3(256), 0, st, (const bf16*)low, target, a, ignore_index, lse, row_loss); break;
    default: hipLaunchKernelGGL(upce_fwd_kernel<4>, dim3(grid), dim3(256), 0, st, (const bf16*)low, target, a, ignore_index, lse, row_loss); break;
  }
  TOK_CHECK_LAUNCH("tok_upsample_ce_fwd");
  double* part = reinterpret_cast<double*>(loss + 2);
  const int nparts = rows < 4096 ? 1 : (tok_cdiv(rows, 4096) < CE_PARTS ? tok_cdiv(rows, 4096) : CE_PARTS);
  const int chunk = tok_cdiv(rows, nparts);
  hipLaunchKernelGGL(ce_partial_kernel, dim3(nparts), dim3(256), 0, st, row_loss, target, rows, ignore_index, classes, chunk, part);
  TOK_CHECK_LAUNCH("tok_upsample_ce_fwd(partial)");
  hipLaunchKernelGGL(ce_mean_kernel, dim3(1), dim3(256), 0, st, part, nparts, loss);
  TOK_CHECK_LAUNCH("tok_upsample_ce_fwd(mean)");
  return TOK_OK;
}

extern "C" int tok_upsample_ce_bwd(const void* low, int n, int hs, int ws, int classes, int ld, int hd, int wd,
                                   const int64_t* target, int64_t ignore_index, const float* lse, const float* loss,
                                   const float* gscale, void* dlow, int accumulate, void* stream) {
  UpArgs a;
  TOK_CHECK_ARG(low && target && lse && loss && dlow && fill_up(a, n, hs, ws, classes, ld, hd, wd), "tok_upsample_ce_bwd: bad args");
  TOK_CHECK_ARG((reinterpret_cast<uintptr_t>(low) & 15) == 0 && (reinterpret_cast<uintptr_t>(dlow) & 15) == 0,
                "tok_upsample_ce_bwd: low / dlow must be 16-byte aligned");
  hipStream_t st = tok_stream(stream);
  if (up_tiled_ok(a)) {
    const int tiles_y = tok_cdiv(hs, UT), tiles_x = tok_cdiv(ws, UT);
    const int tgrid = n * tiles_y * tiles_x;
    const int smem = UWIN * UWIN * ld * 2;
#define TOK_UPCE_TILED(NV)                                                                                                          \
    {                                                                                                                               \
      static const bool attr_set = [] {                                                                                             \
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&upce_bwd_tiled_kernel<NV>),                                        \
                                  hipFuncAttributeMaxDynamicSharedMemorySize, UWIN * UWIN * NV * 16);                               \
        return true;                                                                                                                \
      }();                                                                                                                          \
      (void)attr_set;                                                                                                               \
      hipLaunchKernelGGL(upce_bwd_tiled_kernel<NV>, dim3(tgrid), dim3(256), smem, st, (const bf16*)low, target, a, ignore_index,    \
                         lse, loss, gscale, (bf16*)dlow, accumulate, tiles_y, tiles_x);                                             \
    }
    switch (ld >> 3) {
      case 1: TOK_UPCE_TILED(1) break;
      case 2: TOK_UPCE_TILED(2) break;
      case 3: TOK_UPCE_TILED(3) break;
      default: TOK_UPCE_TILED(4) break;
    }
#undef TOK_UPCE_TILED
    TOK_CHECK_LAUNCH("tok_upsample_ce_bwd(tiled)");
    return TOK_OK;
  }
  const int grid = up_blocks((size_t)n * hs * ws);
  switch (ld >> 3) {
    case 1: hipLaunchKernelGGL(upce_bwd_kernel<1>, dim3(grid), dim3(256), 0, st, (const bf16*)low, target, a, ignore_index, lse, loss, gscale, (bf16*)dlow, accumulate); break;
    case 2: hipLaunchKernelGGL(upce_bwd_kernel<2>, dim3(grid), dim3(256), 0, st, (const bf16*)low, target, a, ignore_index, lse, loss, gscale, (bf16*)dlow, accumulate); break;
    case 3: hipLaunchKernelGGL(upce_bwd_kernel<3>, dim3(grid), dim3(256), 0, st, (const bf16*)low, target, a, ignore_index, lse, loss, gscale, (bf16*)dlow, accumulate); break;
    default: hipLaunchKernelGGL(upce_bwd_kernel<4>, dim3(grid), dim3(256), 0, st, (const bf16*)low, target, a, ignore_index, lse, loss, gscale, (bf16*)dlow, accumulate); break;
  }
  TOK_CHECK_LAUNCH("tok_upsample_ce_bwd");
  return TOK_OK;
}

namespace {
// BCEWithLogitsLoss with an ignore value (losses/classification/binary_cross_entropy.py:50-59): elements whose target
// equals `ignore` are dropped, the rest take  (1 - t) x - log_sigmoid(x)  in fp32 (ATen's formula), mean or sum.
// Stage 1: one contiguous chunk of the row-major element range per block -> (sum, count) doubles; stage 2 folds them in
// fixed order.  Element e lives at logits[(e / classes) * ld + e % classes].
__device__ __forceinline__ float bce_elem(float x, float t) {
  const float ls = fminf(x, 0.f) - log1pf(expf(-fabsf(x)));   // log_sigmoid(x)
  return (1.f - t) * x - ls;
}

__global__ __launch_bounds__(256) void bce_partial_kernel(const bf16* __restrict__ logits, const float* __restrict__ target,
                                                          int64_t n, int classes, int ld, float ignore, int64_t chunk,
                                                          double* __restrict__ part) {
  const int64_t e0 = (int64_t)blockIdx.x * chunk;
  const int64_t e1 = e0 + chunk < n ? e0 + chunk : n;
  double s = 0.0, cnt = 0.0;
  for (int64_t e = e0 + threadIdx.x; e < e1; e += 256) {
    const float t = target[e];
    if (t != ignore) {
      const int64_t r = e / classes;
      const int c = (int)(e - r * classes);
      s += (double)bce_elem(bf2f(logits[r * ld + c]), t);
      cnt += 1.0;
    }
  }
  block_reduce2(s, cnt);
  if (threadIdx.x == 0) { part[2 * blockIdx.x] = s; part[2 * blockIdx.x + 1] = cnt; }
}

__global__ __launch_bounds__(256) void bce_final_kernel(const double* __restrict__ part, int nparts, int mean, float* loss) {
  double s = 0.0, cnt = 0.0;
  for (int i = threadIdx.x; i < nparts; i += 256) { s += part[2 * i]; cnt += part[2 * i + 1]; }
  block_reduce2(s, cnt);
  if (threadIdx.x == 0) {
    // nothing selected: the reference returns a zero (binary_cross_entropy.py:58-59), not 0/0
    loss[0] = cnt > 0.0 ? (float)(mean ? s / cnt : s) : 0.f;
    loss[1] = (float)cnt;
  }
}

// d loss / d x = (sigmoid(x) - t) * gscale / (mean ? n_valid : 1) on the selected elements, 0 elsewhere and in the pad
__global__ __launch_bounds__(256) void bce_bwd_kernel(const bf16* __restrict__ logits, const float* __restrict__ target,
                                                      const float* __restrict__ loss, const float* __restrict__ gscale,
                                                      int64_t rows, int classes, int ld, float ignore, int mean,
                                                      bf16* __restrict__ dlogits) {
  const float nv = loss[1];
  const float g = gscale[0] * (mean ? (nv > 0.f ? 1.f / nv : 0.f) : 1.f);
  const int64_t total = rows * ld;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
    const int64_t r = i / ld;
    const int c = (int)(i - r * ld);
    float d = 0.f;
    if (c < classes) {
      const float t = target[r * classes + c];
      if (t != ignore) {
        const float x = bf2f(logits[i]);
        d = (1.f / (1.f + expf(-x)) - t) * g;
      }
    }
    dlogits[i] = f2bf(d);
  }
}
}  // namespace

extern "C" int tok_bce_logits_fwd(const void* logits, const float* target, int64_t rows, int classes, int ld,
                                  float ignore_value, int mean, float* loss, void* stream) {
  TOK_CHECK_ARG(logits && target && loss, "tok_bce_logits_fwd: null pointer");
  TOK_CHECK_ARG(rows > 0 && classes > 0 && ld >= classes, "tok_bce_logits_fwd: bad sizes");
  TOK_CHECK_ARG((reinterpret_cast<uintptr_t>(loss) & 7) == 0, "tok_bce_logits_fwd: loss must be 8-byte aligned");
  hipStream_t st = tok_stream(stream);
  double* part = reinterpret_cast<double*>(loss + 2);   // loss holds TOK_CE_LOSS_FLOATS floats
  const int64_t n = rows * classes;
  const int nparts = n < 8192 ? 1 : (int)(tok_cdiv(n, (int64_t)8192) < CE_PARTS ? tok_cdiv(n, (int64_t)8192) : CE_PARTS);
  const int64_t chunk = tok_cdiv(n, (int64_t)nparts);
  hipLaunchKernelGGL(bce_partial_kernel, dim3(nparts), dim3(256), 0, st, (const bf16*)logits, target, n, classes, ld,
                     ignore_value, chunk, part);
  TOK_CHECK_LAUNCH("tok_bce_logits_fwd(partial)");
  hipLaunchKernelGGL(bce_final_kernel, dim3(1), dim3(256), 0, st, part, nparts, mean, loss);
  TOK_CHECK_LAUNCH("tok_bce_logits_fwd(final)");
  return TOK_OK;
}

extern "C" int tok_bce_logits_bwd(const void* logits, const float* target, const float* loss, const float* gscale,
                                  int64_t rows, int classes, int ld, float ignore_value, int mean, void* dlogits,
                                  void* stream) {
  TOK_CHECK_ARG(logits && target && loss && gscale && dlogits, "tok_bce_logits_bwd: null pointer");
  TOK_CHECK_ARG(rows > 0 && classes > 0 && ld >= classes, "tok_bce_logits_bwd: bad sizes");
  const int64_t total = rows * ld;
  const int blocks = (int)(tok_cdiv(total, (int64_t)256) < 4096 ? tok_cdiv(total, (int64_t)256) : 4096);
  hipLaunchKernelGGL(bce_bwd_kernel, dim3(blocks), dim3(256), 0, tok_stream(stream), (const bf16*)logits, target, loss,
                     gscale, rows, classes, ld, ignore_value, mean, (bf16*)dlogits);
  TOK_CHECK_LAUNCH("tok_bce_logits_bwd");
  return TOK_OK;
}

namespace {
// torch.nn.L1Loss / MSELoss / SmoothL1Loss(beta) / HuberLoss(delta) (registered at losses/__init__.py:13-25) on a flat bf16
// prediction and fp32 target: kind 0 |d|, 1 d^2, 2 smooth-L1, 3 Huber; d = x - t.  Same two-stage fp64 fold as above.
__device__ __forceinline__ float reg_elem(int kind, float d, float k) {
  const float a = fabsf(d);
  if (kind == 0) return a;
  if (kind == 1) return d * d;
  if (kind == 2) return a < k ? 0.5f * d * d / k : a - 0.5f * k;
  return a <= k ? 0.5f * d * d : k * (a - 0.5f * k);
}
__device__ __forceinline__ float reg_grad(int kind, float d, float k) {
  const float a = fabsf(d), sg = d > 0.f ? 1.f : (d < 0.f ? -1.f : 0.f);
  if (kind == 0) return sg;
  if (kind == 1) return 2.f * d;
  if (kind == 2) return a < k ? d / k : sg;
  return a <= k ? d : k * sg;
}

__global__ __launch_bounds__(256) void reg_partial_kernel(const bf16* __restrict__ x, const float* __restrict__ t, int64_t n,
                                                          int kind, float k, int64_t chunk, double* __restrict__ part) {
  const int64_t e0 = (int64_t)blockIdx.x * chunk;
  const int64_t e1 = e0 + chunk < n ? e0 + chunk : n;
  double s = 0.0, cnt = 0.0;
  for (int64_t e = e0 + threadIdx.x; e < e1; e += 256) { s += (double)reg_elem(kind, bf2f(x[e]) - t[e], k); cnt += 1.0; }
  block_reduce2(s, cnt);
  if (threadIdx.x == 0) { part[2 * blockIdx.x] = s; part[2 * blockIdx.x + 1] = cnt; }
}

__global__ __launch_bounds__(256) void reg_bwd_kernel(const bf16* __restrict__ x, const float* __restrict__ t,
                                                      const float* __restrict__ gscale, int64_t n, int kind, float k,
                                                      float scale, bf16* __restrict__ dx) {
  const float g = gscale[0] * scale;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256)
    dx[i] = f2bf(reg_grad(kind, bf2f(x[i]) - t[i], k) * g);
}
}  // namespace

extern "C" int tok_regression_loss_fwd(const void* x, const float* target, int64_t n, int kind, float knee, int mean,
                                       float* loss, void* stream) {
  TOK_CHECK_ARG(x && target && loss && n > 0 && kind >= 0 && kind <= 3, "tok_regression_loss_fwd: bad args");
  TOK_CHECK_ARG(kind < 2 || knee > 0.f, "tok_regression_loss_fwd: beta / delta must be positive");
  TOK_CHECK_ARG((reinterpret_cast<uintptr_t>(loss) & 7) == 0, "tok_regression_loss_fwd: loss must be 8-byte aligned");
  hipStream_t st = tok_stream(stream);
  double* part = reinterpret_cast<double*>(loss + 2);   // loss holds TOK_CE_LOSS_FLOATS floats
  const int nparts = n < 8192 ? 1 : (int)(tok_cdiv(n, (int64_t)8192) < CE_PARTS ? tok_cdiv(n, (int64_t)8192) : CE_PARTS);
  const int64_t chunk = (n + nparts - 1) / nparts;
  hipLaunchKernelGGL(reg_partial_kernel, dim3(nparts), dim3(256), 0, st, (const bf16*)x, target, n, kind, knee, chunk, part);
  TOK_CHECK_LAUNCH("tok_regression_loss_fwd(partial)");
  hipLaunchKernelGGL(bce_final_kernel, dim3(1), dim3(256), 0, st, part, nparts, mean, loss);
  TOK_CHECK_LAUNCH("tok_regression_loss_fwd(final)");
  return TOK_OK;
}

extern "C" int tok_regression_loss_bwd(const void* x, const float* target, const float* gscale, int64_t n, int kind,
                                       float knee, int mean, void* dx, void* stream) {
  TOK_CHECK_ARG(x && target && gscale && dx && n > 0 && kind >= 0 && kind <= 3, "tok_regression_loss_bwd: bad args");
  const int blocks = (int)((n + 255) / 256 < 4096 ? (n + 255) / 256 : 4096);
  hipLaunchKernelGGL(reg_bwd_kernel, dim3(blocks), dim3(256), 0, tok_stream(stream), (const bf16*)x, target, gscale, n,
                     kind, knee, mean ? 1.f / (float)n : 1.f, (bf16*)dx);
  TOK_CHECK_LAUNCH("tok_regression_loss_bwd");
  return TOK_OK;
}

namespace {
// classification statistics for the on-device metrics: per class c: counts[0][c] += [argmax == c == target],
// counts[1][c] += [argmax == c], counts[2][c] += [target == c]  (int64 atomics: exact, order-independent)
__global__ __launch_bounds__(256) void cls_stats_kernel(const bf16* __restrict__ logits, const int64_t* __restrict__ labels,
                                                        const int64_t* __restrict__ target, int64_t rows, int classes,
                                                        int ld, int64_t ignore_index, unsigned long long* counts,
                                                        unsigned long long* confusion) {
  const int lane = threadIdx.x & 63;
  for (int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6); row < rows; row += (int64_t)gridDim.x * 4) {
    const int64_t t = target[row];
    if (t == ignore_index || t < 0 || t >= classes) continue;
    int pred;
    if (labels != nullptr) {
      pred = (int)labels[row];
    } else {                       // first maximum, as torch.argmax
      float best = -INFINITY;
      int bi = 0x7fffffff;
      for (int c = lane; c < classes; c += 64) {
        const float v = bf2f(logits[row * ld + c]);
        if (v > best) { best = v; bi = c; }
      }
      for (int off = 32; off > 0; off >>= 1) {
        const float ov = __shfl_xor(best, off, 64);
        const int oi = __shfl_xor(bi, off, 64);
        if (ov > best || (ov == best && oi < bi)) { best = ov; bi = oi; }
      }
      pred = bi;
    }
    if (lane == 0 && confusion != nullptr) {        // confusion[target][prediction] += 1
      if (pred >= 0 && pred < classes) atomicAdd(&confusion[(size_t)t * classes + pred], 1ull);
    } else if (lane == 0) {
      if (pred >= 0 && pred < classes) {
        atomicAdd(&counts[classes + pred], 1ull);
        if (pred == (int)t) atomicAdd(&counts[pred], 1ull);
      }
      atomicAdd(&counts[2 * classes + (int)t], 1ull);
    }
  }
}
}  // namespace

extern "C" int tok_cls_stats_update(const void* logits, const int64_t* labels, const int64_t* target, int64_t rows,
                                    int classes, int ld, int64_t ignore_index, int64_t* counts, void* stream) {
  TOK_CHECK_ARG((logits != nullptr) != (labels != nullptr), "tok_cls_stats_update: give logits OR predicted labels");
  TOK_CHECK_ARG(target && counts && rows > 0 && classes > 0 && (labels || ld >= classes), "tok_cls_stats_update: bad args");
  const int64_t b = (rows + 3) / 4;
  hipLaunchKernelGGL(cls_stats_kernel, dim3((unsigned)(b > 2048 ? 2048 : b)), dim3(256), 0, tok_stream(stream),
                     (const bf16*)logits, labels, target, rows, classes, ld, ignore_index, (unsigned long long*)counts,
                     (unsigned long long*)nullptr);
  TOK_CHECK_LAUNCH("tok_cls_stats_update");
  return TOK_OK;
}

extern "C" int tok_confusion_update(const void* logits, const int64_t* labels, const int64_t* target, int64_t rows,
                                    int classes, int ld, int64_t ignore_index, int64_t* confusion, void* stream) {
  TOK_CHECK_ARG((logits != nullptr) != (labels != nullptr), "tok_confusion_update: give logits OR predicted labels");
  TOK_CHECK_ARG(target && confusion && rows > 0 && classes > 0 && (labels || ld >= classes), "tok_confusion_update: bad args");
  const int64_t b = (rows + 3) / 4;
  hipLaunchKernelGGL(cls_stats_kernel, dim3((unsigned)(b > 2048 ? 2048 : b)), dim3(256), 0, tok_stream(stream),
                     (const bf16*)logits, labels, target, rows, classes, ld, ignore_index, (unsigned long long*)nullptr,
                     (unsigned long long*)confusion);
  TOK_CHECK_LAUNCH("tok_confusion_update");
  return TOK_OK;
}
