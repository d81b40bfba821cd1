// Pooling kernels, NHWC bf16: 3x3/s2/p1 max-pool (index-exact), global average pool, column sums.
#include "tok_common.h"
#include <math.h>

namespace {

// One thread = one output pixel x 8 channels.  argmax = tap index r*3+s of the FIRST maximum in
// (r, s) scan order over the in-bounds taps; NaN wins (aten max_pool2d: `val > max || isnan(val)`).
__global__ __launch_bounds__(256) void maxpool_fwd_kernel(const bf16* __restrict__ x, bf16* __restrict__ y,
                                                          uint8_t* __restrict__ argmax, int N, int H,
                                                          int W, int C, int P, int Q) {
  const int cg_total = C >> 3;
  const size_t total = (size_t)N * P * Q * cg_total;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total;
       i += (size_t)gridDim.x * blockDim.x) {
    const int cg = (int)(i % cg_total);
    size_t pix = i / cg_total;
    const int q = (int)(pix % Q);
    pix /= Q;
    const int p = (int)(pix % P);
    const int n = (int)(pix / P);
    float best[8];
    int idx[8];
    bool first = true;
#pragma unroll
    for (int r = 0; r < 3; ++r) {
      const int h = 2 * p - 1 + r;
      if ((unsigned)h >= (unsigned)H) continue;
#pragma unroll
      for (int s = 0; s < 3; ++s) {
        const int w = 2 * q - 1 + s;
        if ((unsigned)w >= (unsigned)W) continue;
        const bf16x8 v = ldg16(x + (((size_t)n * H + h) * W + w) * C + cg * 8);
        if (first) {
#pragma unroll
          for (int e = 0; e < 8; ++e) { best[e] = -INFINITY; idx[e] = r * 3 + s; }
          first = false;
        }
#pragma unroll
        for (int e = 0; e < 8; ++e) {
          const float f = bf2f(v[e]);
          if (f > best[e] || f != f) { best[e] = f; idx[e] = r * 3 + s; }
        }
      }
    }
    bf16x8 o;
    uint64_t packed = 0;
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      o[e] = f2bf(best[e]);
      packed |= (uint64_t)(idx[e] & 0xff) << (8 * e);
    }
    const size_t off = (((size_t)n * P + p) * Q + q) * C + cg * 8;
    stg16(y + off, o);
    *reinterpret_cast<uint64_t*>(argmax + off) = packed;
  }
}

// Gather formulation (no atomics): each input pixel visits the <= 4 windows that contain it.
__global__ __launch_bounds__(256) void maxpool_bwd_kernel(const bf16* __restrict__ dy,
                                                          const uint8_t* __restrict__ argmax,
                                                          bf16* dx, int accumulate, int N, int H, int W,
                                                          int C, int P, int Q) {
  const int cg_total = C >> 3;
  const size_t total = (size_t)N * H * W * cg_total;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total;
       i += (size_t)gridDim.x * blockDim.x) {
    const int cg = (int)(i % cg_total);
    size_t pix = i / cg_total;
    const int w = (int)(pix % W);
    pix /= W;
    const int h = (int)(pix % H);
    const int n = (int)(pix / H);
    float g[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) g[e] = 0.f;
    // windows p with 2p-1 <= h <= 2p+1: at most 2 x 2; all four (tap index, gradient) pairs are loaded unconditionally
    // (clamped addresses: the eight loads are in flight together), a window that does not exist contributes nothing
    const int p0 = h >> 1, p1 = (h + 1) >> 1, q0 = w >> 1, q1 = (w + 1) >> 1;
    const bool vp[2] = {p0 < P, p1 != p0 && p1 < P}, vq[2] = {q0 < Q, q1 != q0 && q1 < Q};
    const int pc[2] = {p0 < P ? p0 : P - 1, vp[1] ? p1 : (p0 < P ? p0 : P - 1)};
    const int qc[2] = {q0 < Q ? q0 : Q - 1, vq[1] ? q1 : (q0 < Q ? q0 : Q - 1)};
    uint64_t packed[4];
    bf16x8 d[4];
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
      for (int b = 0; b < 2; ++b) {
        const size_t off = (((size_t)n * P + pc[a]) * Q + qc[b]) * C + cg * 8;
        packed[a * 2 + b] = *reinterpret_cast<const uint64_t*>(argmax + off);
        d[a * 2 + b] = ldg16(dy + off);
      }
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
      for (int b = 0; b < 2; ++b) {
        const bool valid = vp[a] && vq[b];
        const int tap = (h - (2 * pc[a] - 1)) * 3 + (w - (2 * qc[b] - 1));
#pragma unroll
        for (int e = 0; e < 8; ++e)
          g[e] += (valid && (int)((packed[a * 2 + b] >> (8 * e)) & 0xff) == tap) ? bf2f(d[a * 2 + b][e]) : 0.f;
      }
    const size_t xoff = (((size_t)n * H + h) * W + w) * C + cg * 8;
    bf16x8 o;
    if (accumulate) {
      const bf16x8 old = ldg16(dx + xoff);
#pragma unroll
      for (int e = 0; e < 8; ++e) o[e] = f2bf(g[e] + bf2f(old[e]));
    } else {
#pragma unroll
      for (int e = 0; e < 8; ++e) o[e] = f2bf(g[e]);
    }
    stg16(dx + xoff, o);
  }
}

__global__ __launch_bounds__(256) void gap_fwd_kernel(const bf16* __restrict__ x, bf16* __restrict__ y,
                                                      int N, int HW, int C, float inv) {
  const int cg_total = C >> 3;
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= N * cg_total) return;
  const int cg = i % cg_total, n = i / cg_total;
  float acc[8];
#pragma unroll
  for (int e = 0; e < 8; ++e) acc[e] = 0.f;
  const bf16* p = x + (size_t)n * HW * C + cg * 8;
  for (int j = 0; j < HW; ++j) {
    const bf16x8 v = ldg16(p + (size_t)j * C);
#pragma unroll
    for (int e = 0; e < 8; ++e) acc[e] += bf2f(v[e]);
  }
  bf16x8 o;
#pragma unroll
  for (int e = 0; e < 8; ++e) o[e] = f2bf(acc[e] * inv);
  stg16(y + (size_t)n * C + cg * 8, o);
}

__global__ __launch_bounds__(256) void gap_bwd_kernel(const bf16* __restrict__ dy, bf16* dx, int accumulate,
                                                      int N, int HW, int C, float inv) {
  const int cg_total = C >> 3;
  const size_t total = (size_t)N * HW * cg_total;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total;
       i += (size_t)gridDim.x * blockDim.x) {
    const int cg = (int)(i % cg_total);
    const size_t pix = i / cg_total;
    const int n = (int)(pix / HW);
    const bf16x8 d = ldg16(dy + (size_t)n * C + cg * 8);
    const size_t off = pix * C + cg * 8;
    bf16x8 o;
    if (accumulate) {
      const bf16x8 old = ldg16(dx + off);
#pragma unroll
      for (int e = 0; e < 8; ++e) o[e] = f2bf(bf2f(d[e]) * inv + bf2f(old[e]));
    } else {
#pragma unroll
      for (int e = 0; e < 8; ++e) o[e] = f2bf(bf2f(d[e]) * inv);
    }
    stg16(dx + off, o);
  }
}

// one block per 8-column group; 256 threads stride the rows; fixed-order LDS tree (deterministic)
__global__ __launch_bounds__(256) void colsum_kernel(const bf16* __restrict__ dy, int64_t M, int ld,
                                                     int n_real, float* out, int accumulate) {
  __shared__ float red[256][8];
  const int cg = blockIdx.x, tid = threadIdx.x;
  float acc[8];
#pragma unroll
  for (int e = 0; e < 8; ++e) acc[e] = 0.f;
  for (int64_t m = tid; m < M; m += 256) {
    const bf16x8 v = ldg16(dy + (size_t)m * ld + cg * 8);
#pragma unroll
    for (int e = 0; e < 8; ++e) acc[e] += bf2f(v[e]);
  }
#pragma unroll
  for (int e = 0; e < 8; ++e) red[tid][e] = acc[e];
  __syncthreads();
  for (int s = 128; s > 0; s >>= 1) {
    if (tid < s)
#pragma unroll
      for (int e = 0; e < 8; ++e) red[tid][e] += red[tid + s][e];
    __syncthreads();
  }
  if (tid < 8) {
    const int c = cg * 8 + tid;
    if (c < n_real) out[c] = accumulate ? out[c] + red[0][tid] : red[0][tid];
  }
}

// tall matrices (token rows of a transformer: M ~ 10^5..10^6): every block sweeps a contiguous row chunk with
// fully coalesced 16-byte row segments and leaves one fp32 partial row; tok_colsum_f32 folds the rows
__global__ __launch_bounds__(256) void colsum_partial_kernel(const bf16* __restrict__ dy, int64_t M, int ld,
                                                             int chunk, float* __restrict__ partial) {
  extern __shared__ float red[];   // [rpb][ld]
  const int cgs = ld >> 3;
  const int cgt = cgs < 256 ? cgs : 256;      // column groups handled per pass
  const int rpb = 256 / cgt;                  // rows in flight per pass
  const int cgl = threadIdx.x % cgt, rl = threadIdx.x / cgt;
  const int64_t r0 = (int64_t)blockIdx.x * chunk;
  const int64_t r1 = r0 + chunk < M ? r0 + chunk : M;
  for (int cg = cgl; cg < cgs; cg += cgt) {
    float acc[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) acc[e] = 0.f;
    if (rl < rpb && r0 < r1) {
      // 8 independent 16-byte loads in flight per lane (rows past the chunk are clamped and dropped); the
      // additions keep the row order, so the partial sums do not depend on the unroll factor
      const bf16* col = dy + cg * 8;
      for (int64_t m = r0 + rl; m < r1; m += 8 * (int64_t)rpb) {
        bf16x8 v[8];
        bool live[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) {
          const int64_t mu = m + (int64_t)u * rpb;
          live[u] = mu < r1;
          v[u] = ldg16(col + (size_t)(mu < r1 ? mu : r1 - 1) * ld);
        }
#pragma unroll
        for (int u = 0; u < 8; ++u)
#pragma unroll
          for (int e = 0; e < 8; ++e) acc[e] += live[u] ? bf2f(v[u][e]) : 0.f;
      }
    }
    if (rl < rpb)
#pragma unroll
      for (int e = 0; e < 8; ++e) red[(size_t)rl * ld + cg * 8 + e] = acc[e];
  }
  __syncthreads();
  for (int c = threadIdx.x; c < ld; c += 256) {
    float s = 0.f;
    for (int r = 0; r < rpb; ++r) s += red[(size_t)r * ld + c];
    partial[(size_t)blockIdx.x * ld + c] = s;
  }
}

inline int grid_for(size_t total) {
  size_t b = (total + 255) / 256;
  return (int)(b > 4096 ? 4096 : (b < 1 ? 1 : b));
}

}  // namespace

extern "C" int tok_colsum_partial_rows(int64_t m, int n_pad) {
  (void)n_pad;
  const int64_t b = (m + 255) / 256;
  return (int)(b > 1024 ? 1024 : (b < 1 ? 1 : b));
}

extern "C" int tok_colsum_partial(const void* dy, int64_t m, int n_pad, float* partial, void* stream) {
  TOK_CHECK_ARG(dy && partial && m > 0 && n_pad > 0 && n_pad % 8 == 0, "tok_colsum_partial: bad args");
  const int g = tok_colsum_partial_rows(m, n_pad);
  const int chunk = (int)((m + g - 1) / g);
  const int cgs = n_pad >> 3, cgt = cgs < 256 ? cgs : 256, rpb = 256 / cgt;
  const size_t smem = (size_t)rpb * n_pad * sizeof(float);
  TOK_CHECK_ARG(smem <= 64 * 1024, "tok_colsum_partial: n_pad too large (%d)", n_pad);
  hipLaunchKernelGGL(colsum_partial_kernel, dim3(g), dim3(256), smem, tok_stream(stream), (const bf16*)dy, m, n_pad,
                     chunk, partial);
  TOK_CHECK_LAUNCH("tok_colsum_partial");
  return TOK_OK;
}

extern "C" int tok_maxpool3x3s2_fwd(const void* x, void* y, uint8_t* argmax, int n, int h, int w, int c,
                                    void* stream) {
  TOK_CHECK_ARG(x && y && argmax && n > 0 && h > 0 && w > 0 && c > 0 && c % 8 == 0, "tok_maxpool3x3s2_fwd: bad args");
  const int P = (h + 2 - 3) / 2 + 1, Q = (w + 2 - 3) / 2 + 1;
  hipLaunchKernelGGL(maxpool_fwd_kernel, dim3(grid_for((size_t)n * P * Q * (c / 8))), dim3(256), 0,
                     tok_stream(stream), (const bf16*)x, (bf16*)y, argmax, n, h, w, c, P, Q);
  TOK_CHECK_LAUNCH("tok_maxpool3x3s2_fwd");
  return TOK_OK;
}

extern "C" int tok_maxpool3x3s2_bwd(const void* dy, const uint8_t* argmax, void* dx, int accumulate, int n,
                                    int h, int w, int c, void* stream) {
  TOK_CHECK_ARG(dy && dx && argmax && n > 0 && h > 0 && w > 0 && c > 0 && c % 8 == 0, "tok_maxpool3x3s2_bwd: bad args");
  const int P = (h + 2 - 3) / 2 + 1, Q = (w + 2 - 3) / 2 + 1;
  hipLaunchKernelGGL(maxpool_bwd_kernel, dim3(grid_for((size_t)n * h * w * (c / 8))), dim3(256), 0,
                     tok_stream(stream), (const bf16*)dy, argmax, (bf16*)dx, accumulate, n, h, w, c, P, Q);
  TOK_CHECK_LAUNCH("tok_maxpool3x3s2_bwd");
  return TOK_OK;
}

namespace {
// AvgPool2d(2, stride 2, ceil_mode=True, count_include_pad=False) — the 'avg_down' shortcut of the d-variant ResNets
// ([timm] downsample_avg): output ceil(h/2) x ceil(w/2); a window hanging over the edge averages its valid pixels only.
// One thread per (image, output pixel, 8 channels); NHWC bf16, 16-byte accesses.
__global__ __launch_bounds__(256) void avgpool2_fwd_kernel(const bf16* __restrict__ x, bf16* __restrict__ y, int n, int h,
                                                           int w, int c) {
  const int p = (h + 1) >> 1, q = (w + 1) >> 1, c8 = c >> 3;
  const size_t total = (size_t)n * p * q * c8;
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (size_t)gridDim.x * 256) {
    const int cc = (int)(i % c8);
    size_t t = i / c8;
    const int oq = (int)(t % q); t /= q;
    const int op = (int)(t % p);
    const int b = (int)(t / p);
    float acc[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    int cnt = 0;
#pragma unroll
    for (int dh = 0; dh < 2; ++dh)
#pragma unroll
      for (int dw = 0; dw < 2; ++dw) {
        const int hh = 2 * op + dh, ww = 2 * oq + dw;
        if (hh < h && ww < w) {
          const bf16x8 v = ldg16(x + (((size_t)b * h + hh) * w + ww) * c + cc * 8);
#pragma unroll
          for (int e = 0; e < 8; ++e) acc[e] += bf2f(v[e]);
          ++cnt;
        }
      }
    const float inv = 1.f / (float)cnt;
    bf16x8 o;
#pragma unroll
    for (int e = 0; e < 8; ++e) o[e] = f2bf(acc[e] * inv);
    stg16(y + i * 8, o);
  }
}

__global__ __launch_bounds__(256) void avgpool2_bwd_kernel(const bf16* __restrict__ dy, bf16* __restrict__ dx, int accumulate,
                                                           int n, int h, int w, int c) {
  const int p = (h + 1) >> 1, q = (w + 1) >> 1, c8 = c >> 3;
  const size_t total = (size_t)n * h * w * c8;
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (size_t)gridDim.x * 256) {
    const int cc = (int)(i % c8);
    size_t t = i / c8;
    const int ww = (int)(t % w); t /= w;
    const int hh = (int)(t % h);
    const int b = (int)(t / h);
    const int op = hh >> 1, oq = ww >> 1;
    const int cnt = ((2 * op + 1 < h) ? 2 : 1) * ((2 * oq + 1 < w) ? 2 : 1);
    const float inv = 1.f / (float)cnt;
    const bf16x8 g = ldg16(dy + (((size_t)b * p + op) * q + oq) * c + cc * 8);
    bf16x8 o;
    if (accumulate) {
      const bf16x8 old = ldg16(dx + i * 8);
#pragma unroll
      for (int e = 0; e < 8; ++e) o[e] = f2bf(bf2f(old[e]) + bf2f(g[e]) * inv);
    } else {
#pragma unroll
      for (int e = 0; e < 8; ++e) o[e] = f2bf(bf2f(g[e]) * inv);
    }
    stg16(dx + i * 8, o);
  }
}
}  // namespace

extern "C" int tok_avgpool2x2_fwd(const void* x, void* y, int n, int h, int w, int c, void* stream) {
  TOK_CHECK_ARG(x && y && n > 0 && h > 0 && w > 0 && c > 0 && c % 8 == 0, "tok_avgpool2x2_fwd: bad args");
  const size_t total = (size_t)n * ((h + 1) / 2) * ((w + 1) / 2) * (c / 8);
  hipLaunchKernelGGL(avgpool2_fwd_kernel, dim3(grid_for(total)), dim3(256), 0, tok_stream(stream), (const bf16*)x, (bf16*)y,
                     n, h, w, c);
  TOK_CHECK_LAUNCH("tok_avgpool2x2_fwd");
  return TOK_OK;
}

extern "C" int tok_avgpool2x2_bwd(const void* dy, void* dx, int accumulate, int n, int h, int w, int c, void* stream) {
  TOK_CHECK_ARG(dy && dx && n > 0 && h > 0 && w > 0 && c > 0 && c % 8 == 0, "tok_avgpool2x2_bwd: bad args");
  hipLaunchKernelGGL(avgpool2_bwd_kernel, dim3(grid_for((size_t)n * h * w * (c / 8))), dim3(256), 0, tok_stream(stream),
                     (const bf16*)dy, (bf16*)dx, accumulate, n, h, w, c);
  TOK_CHECK_LAUNCH("tok_avgpool2x2_bwd");
  return TOK_OK;
}

extern "C" int tok_gap_fwd(const void* x, void* y, int n, int hw, int c, void* stream) {
  TOK_CHECK_ARG(x && y && n > 0 && hw > 0 && c > 0 && c % 8 == 0, "tok_gap_fwd: bad args");
  hipLaunchKernelGGL(gap_fwd_kernel, dim3((n * (c / 8) + 255) / 256), dim3(256), 0, tok_stream(stream),
                     (const bf16*)x, (bf16*)y, n, hw, c, 1.0f / (float)hw);
  TOK_CHECK_LAUNCH("tok_gap_fwd");
  return TOK_OK;
}

extern "C" int tok_gap_bwd(const void* dy, void* dx, int accumulate, int n, int hw, int c, void* stream) {
  TOK_CHECK_ARG(dy && dx && n > 0 && hw > 0 && c > 0 && c % 8 == 0, "tok_gap_bwd: bad args");
  hipLaunchKernelGGL(gap_bwd_kernel, dim3(grid_for((size_t)n * hw * (c / 8))), dim3(256), 0,
                     tok_stream(stream), (const bf16*)dy, (bf16*)dx, accumulate, n, hw, c, 1.0f / (float)hw);
  TOK_CHECK_LAUNCH("tok_gap_bwd");
  return TOK_OK;
}

namespace {
// SelectAdaptivePool2d(output_size=1) beyond 'avg': max / avgmax / catavgmax ([timm] adaptive_avgmax_pool via the
// reference's poolings/classification/pooling.py:7-12).  One thread owns 8 channels of one image and walks its pixels:
// running fp32 sum, running max with the FIRST maximal pixel as argmax (ATen's adaptive_max_pool2d keeps the earlier
// pixel on ties).  Values follow the bf16 autocast chain of the reference: avg and max are bf16 tensors, avgmax is
// 0.5 * bf16(avg + max).
__global__ __launch_bounds__(256) void global_pool_fwd_kernel(const bf16* __restrict__ x, bf16* __restrict__ y,
                                                              int* __restrict__ argmax, int N, int HW, int C, int ldy,
                                                              int mode, float inv) {
  const int cg_total = C >> 3;
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= N * cg_total) return;
  const int cg = i % cg_total, n = i / cg_total;
  float acc[8], mx[8];
  int am[8];
#pragma unroll
  for (int e = 0; e < 8; ++e) { acc[e] = 0.f; mx[e] = -INFINITY; am[e] = 0; }
  const bf16* p = x + (size_t)n * HW * C + cg * 8;
  for (int j = 0; j < HW; ++j) {
    const bf16x8 v = ldg16(p + (size_t)j * C);
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      const float f = bf2f(v[e]);
      acc[e] += f;
      if (f > mx[e] || f != f) { mx[e] = f; am[e] = j; }
    }
  }
  bf16x8 oa, om;
#pragma unroll
  for (int e = 0; e < 8; ++e) { oa[e] = f2bf(acc[e] * inv); om[e] = f2bf(mx[e]); }
  bf16* yo = y + (size_t)n * ldy + cg * 8;
  if (mode == 1) {
    stg16(yo, om);
  } else if (mode == 2) {
    bf16x8 o;
#pragma unroll
    for (int e = 0; e < 8; ++e) o[e] = f2bf(0.5f * bf2f(f2bf(bf2f(oa[e]) + bf2f(om[e]))));
    stg16(yo, o);
  } else {
    stg16(yo, oa);
    stg16(yo + C, om);
  }
#pragma unroll
  for (int e = 0; e < 8; ++e) argmax[(size_t)n * C + cg * 8 + e] = am[e];
}

__global__ __launch_bounds__(256) void global_pool_bwd_kernel(const bf16* __restrict__ dy, const int* __restrict__ argmax,
                                                              bf16* dx, int accumulate, int N, int HW, int C, int ldy,
                                                              int mode, float inv) {
  const int cg_total = C >> 3;
  const size_t total = (size_t)N * HW * cg_total;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
    const int cg = (int)(i % cg_total);
    const size_t pix = i / cg_total;
    const int n = (int)(pix / HW), j = (int)(pix - (size_t)n * HW);
    const bf16* g = dy + (size_t)n * ldy + cg * 8;
    const bf16x8 ga = ldg16(g);                                // avg part (mode 2 / 3) or the max gradient (mode 1)
    const bf16x8 gm = mode == 3 ? ldg16(g + C) : ga;
    const int* am = argmax + (size_t)n * C + cg * 8;
    const float wa = mode == 1 ? 0.f : (mode == 2 ? 0.5f * inv : inv);
    const float wm = mode == 2 ? 0.5f : 1.f;
    bf16* d = dx + pix * C + cg * 8;
    bf16x8 o;
    bf16x8 old = zero8();
    if (accumulate) old = ldg16(d);
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      float v = bf2f(ga[e]) * wa + (am[e] == j ? bf2f(gm[e]) * wm : 0.f);
      if (accumulate) v += bf2f(old[e]);
      o[e] = f2bf(v);
    }
    stg16(d, o);
  }
}
}  // namespace

extern "C" int tok_global_pool_fwd(const void* x, void* y, int* argmax, int n, int hw, int c, int ldy, int mode,
                                   void* stream) {
  TOK_CHECK_ARG(x && y && argmax && n > 0 && hw > 0 && c > 0 && c % 8 == 0, "tok_global_pool_fwd: bad args");
  TOK_CHECK_ARG(mode >= 1 && mode <= 3 && ldy >= (mode == 3 ? 2 * c : c) && ldy % 8 == 0,
                "tok_global_pool_fwd: mode 1 (max) / 2 (avgmax) / 3 (catavgmax), ldy >= channels written");
  hipLaunchKernelGGL(global_pool_fwd_kernel, dim3((n * (c / 8) + 255) / 256), dim3(256), 0, tok_stream(stream),
                     (const bf16*)x, (bf16*)y, argmax, n, hw, c, ldy, mode, 1.0f / (float)hw);
  TOK_CHECK_LAUNCH("tok_global_pool_fwd");
  return TOK_OK;
}

extern "C" int tok_global_pool_bwd(const void* dy, const int* argmax, void* dx, int accumulate, int n, int hw, int c,
                                   int ldy, int mode, void* stream) {
  TOK_CHECK_ARG(dy && argmax && dx && n > 0 && hw > 0 && c > 0 && c % 8 == 0, "tok_global_pool_bwd: bad args");
  TOK_CHECK_ARG(mode >= 1 && mode <= 3 && ldy >= (mode == 3 ? 2 * c : c) && ldy % 8 == 0, "tok_global_pool_bwd: bad mode / ldy");
  hipLaunchKernelGGL(global_pool_bwd_kernel, dim3(grid_for((size_t)n * hw * (c / 8))), dim3(256), 0, tok_stream(stream),
                     (const bf16*)dy, argmax, (bf16*)dx, accumulate, n, hw, c, ldy, mode, 1.0f / (float)hw);
  TOK_CHECK_LAUNCH("tok_global_pool_bwd");
  return TOK_OK;
}

extern "C" int tok_colsum(const void* dy, int64_t m, int n_pad, int n_real, float* out, int accumulate,
                          void* stream) {
  TOK_CHECK_ARG(dy && out && m > 0 && n_pad > 0 && n_pad % 8 == 0 && n_real <= n_pad, "tok_colsum: bad args");
  hipLaunchKernelGGL(colsum_kernel, dim3(n_pad / 8), dim3(256), 0, tok_stream(stream), (const bf16*)dy, m,
                     n_pad, n_real, out, accumulate);
  TOK_CHECK_LAUNCH("tok_colsum");
  return TOK_OK;
}
