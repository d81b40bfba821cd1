// Fused softmax cross-entropy (mean reduction, ignore_index), bf16 logits, fp32 math.
// One wavefront per row: lane-strided max / sum-exp with 64-lane butterflies, no LDS.
#include "tok_common.h"
#include <math.h>

namespace {

__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) v = fmaxf(v, __shfl_xor(v, off, 64));
  return v;
}

__global__ __launch_bounds__(256) void ce_fwd_kernel(const bf16* __restrict__ logits,
                                                     const int64_t* __restrict__ target, int rows,
                                                     int classes, int ld, int64_t ignore_index,
                                                     float* __restrict__ lse, float* __restrict__ row_loss) {
  const int lane = threadIdx.x & 63;
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= rows) return;
  const bf16* z = logits + (size_t)row * ld;
  float mx = -INFINITY;
  for (int c = lane; c < classes; c += 64) mx = fmaxf(mx, bf2f(z[c]));
  mx = wave_max(mx);
  float s = 0.f;
  for (int c = lane; c < classes; c += 64) s += expf(bf2f(z[c]) - mx);
  s = wave_sum(s);
  const float l = mx + logf(s);
  if (lane == 0) {
    lse[row] = l;
    const int64_t t = target[row];
    row_loss[row] = (t == ignore_index || t < 0 || t >= classes) ? 0.f : l - bf2f(z[t]);
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
                                                         int64_t ignore_index, int chunk, double* __restrict__ part) {
  const int r0 = blockIdx.x * chunk;
  const int r1 = min(rows, r0 + chunk);
  double s = 0.0, cnt = 0.0;
  for (int r = r0 + threadIdx.x; r < r1; r += 256) {
    if (target[r] != ignore_index) { s += (double)row_loss[r]; cnt += 1.0; }
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
                                                     int ld, int64_t ignore_index, bf16* __restrict__ dlogits) {
  const int lane = threadIdx.x & 63;
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= rows) return;
  const int64_t t = target[row];
  const bool valid = t != ignore_index;
  const float g = valid ? (gscale ? gscale[0] : 1.f) / loss[1] : 0.f;
  const float l = lse[row];
  const bf16* z = logits + (size_t)row * ld;
  bf16* d = dlogits + (size_t)row * ld;
  for (int c = lane; c < ld; c += 64) {
    float v = 0.f;
    if (valid && c < classes) v = (expf(bf2f(z[c]) - l) - (c == t ? 1.f : 0.f)) * g;
    d[c] = f2bf(v);
  }
}

}  // namespace

extern "C" int tok_softmax_ce_fwd(const void* logits, const int64_t* target, int rows, int classes, int ld,
                                  int64_t ignore_index, float* lse, float* row_loss, float* loss,
                                  void* stream) {
  TOK_CHECK_ARG(logits && target && lse && row_loss && loss, "tok_softmax_ce_fwd: null pointer");
  TOK_CHECK_ARG(rows > 0 && classes > 0 && ld >= classes, "tok_softmax_ce_fwd: bad sizes");
  hipStream_t st = tok_stream(stream);
  hipLaunchKernelGGL(ce_fwd_kernel, dim3((rows + 3) / 4), dim3(256), 0, st, (const bf16*)logits, target, rows,
                     classes, ld, ignore_index, lse, row_loss);
  TOK_CHECK_LAUNCH("tok_softmax_ce_fwd");
  double* part = reinterpret_cast<double*>(loss + 2);   // loss holds TOK_CE_LOSS_FLOATS floats, 8-byte aligned
  TOK_CHECK_ARG((reinterpret_cast<uintptr_t>(loss) & 7) == 0, "tok_softmax_ce_fwd: loss must be 8-byte aligned");
  const int nparts = rows < 4096 ? 1 : (tok_cdiv(rows, 4096) < CE_PARTS ? tok_cdiv(rows, 4096) : CE_PARTS);
  const int chunk = tok_cdiv(rows, nparts);
  hipLaunchKernelGGL(ce_partial_kernel, dim3(nparts), dim3(256), 0, st, row_loss, target, rows, ignore_index, chunk,
                     part);
  TOK_CHECK_LAUNCH("tok_softmax_ce_fwd(partial)");
  hipLaunchKernelGGL(ce_mean_kernel, dim3(1), dim3(256), 0, st, part, nparts, loss);
  TOK_CHECK_LAUNCH("tok_softmax_ce_fwd(mean)");
  return TOK_OK;
}

extern "C" int tok_softmax_ce_bwd(const void* logits, const int64_t* target, const float* lse,
                                  const float* loss, const float* gscale, int rows, int classes, int ld,
                                  int64_t ignore_index, void* dlogits, void* stream) {
  TOK_CHECK_ARG(logits && target && lse && loss && dlogits, "tok_softmax_ce_bwd: null pointer");
  TOK_CHECK_ARG(rows > 0 && classes > 0 && ld >= classes, "tok_softmax_ce_bwd: bad sizes");
  hipLaunchKernelGGL(ce_bwd_kernel, dim3((rows + 3) / 4), dim3(256), 0, tok_stream(stream),
                     (const bf16*)logits, target, lse, loss, gscale, rows, classes, ld, ignore_index,
                     (bf16*)dlogits);
  TOK_CHECK_LAUNCH("tok_softmax_ce_bwd");
  return TOK_OK;
}
