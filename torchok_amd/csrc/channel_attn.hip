// DaViT channel attention (torchok/models/backbones/davit.py:131-165) and the pre-norm residual of its blocks.
//   ChannelAttention.forward:  A = softmax_rows((k * scale)^T v)   [head_dim x head_dim] per (image, head), the sum running
//                              over ALL tokens of the image;  out = q A^T.
// The [N x C] token matrix is only streamed: `chan_gram` reduces two 32-wide head slices over the tokens of one image
// into a 32 x 32 fp32 matrix (one workgroup per (image, head), fixed order -> deterministic) with the softmax (forward)
// or the softmax backward (backward) as its epilogue; `chan_apply` multiplies every token's head slice by such a matrix.
// Backward:  dA = dout^T q;  dS = A o (dA - rowsum(dA o A));  dq = dout A;  dk = scale * v dS^T;  dv = scale * k dS.
// head_dim = 32 (every DaViT variant: 96/3, 192/6, ... 1024/32).
#include "tok_common.h"
#include <math.h>

namespace {

constexpr int HD = 32;

// out[u][i][j] = epilogue( scale * sum_n x[n][h*32+i] * y[n][h*32+j] ),  u = image * heads + h
// mode 0: identity   1: row softmax   2: softmax backward with the saved A: A o (G - rowsum(G o A))
__global__ __launch_bounds__(256) void chan_gram_kernel(const bf16* __restrict__ x, int ldx, const bf16* __restrict__ y,
                                                        int ldy, int rows_per_image, int heads, float scale, int mode,
                                                        const float* __restrict__ a_in, float* __restrict__ out) {
  __shared__ float xs[64][HD + 1];
  __shared__ float ys[64][HD + 4];
  __shared__ float gm[HD][HD + 1];
  __shared__ float rowv[HD];
  const int tid = threadIdx.x;
  const int img = blockIdx.x / heads, h = blockIdx.x - img * heads;
  const int i = tid >> 3, j4 = (tid & 7) * 4;
  const int lt = tid >> 2, lc = (tid & 3) * 8;           // loader: token lt of the chunk, 8 channels from lc
  const bf16* xb = x + (size_t)img * rows_per_image * ldx + h * HD;
  const bf16* yb = y + (size_t)img * rows_per_image * ldy + h * HD;
  float acc[4] = {0.f, 0.f, 0.f, 0.f};
  for (int n0 = 0; n0 < rows_per_image; n0 += 64) {
    const int n = n0 + lt;
    bf16x8 xv = zero8(), yv = zero8();
    if (n < rows_per_image) {
      xv = ldg16(xb + (size_t)n * ldx + lc);
      yv = ldg16(yb + (size_t)n * ldy + lc);
    }
#pragma unroll
    for (int e = 0; e < 8; ++e) { xs[lt][lc + e] = bf2f(xv[e]); ys[lt][lc + e] = bf2f(yv[e]); }
    __syncthreads();
#pragma unroll 16
    for (int t = 0; t < 64; ++t) {
      const float xa = xs[t][i];
#pragma unroll
      for (int e = 0; e < 4; ++e) acc[e] = fmaf(xa, ys[t][j4 + e], acc[e]);
    }
    __syncthreads();
  }
  float* o = out + (size_t)blockIdx.x * HD * HD;
  if (mode == 0) {
#pragma unroll
    for (int e = 0; e < 4; ++e) o[i * HD + j4 + e] = acc[e] * scale;
    return;
  }
#pragma unroll
  for (int e = 0; e < 4; ++e) gm[i][j4 + e] = acc[e] * scale;
  __syncthreads();
  if (mode == 1) {
    if (tid < HD) {
      float mx = -INFINITY;
      for (int j = 0; j < HD; ++j) mx = fmaxf(mx, gm[tid][j]);
      float sum = 0.f;
      for (int j = 0; j < HD; ++j) sum += expf(gm[tid][j] - mx);
      rowv[tid] = mx + logf(sum);
    }
    __syncthreads();
#pragma unroll
    for (int e = 0; e < 4; ++e) o[i * HD + j4 + e] = expf(gm[i][j4 + e] - rowv[i]);
  } else {
    const float* A = a_in + (size_t)blockIdx.x * HD * HD;
    if (tid < HD) {
      float rs = 0.f;
      for (int j = 0; j < HD; ++j) rs = fmaf(gm[tid][j], A[tid * HD + j], rs);
      rowv[tid] = rs;
    }
    __syncthreads();
#pragma unroll
    for (int e = 0; e < 4; ++e) o[i * HD + j4 + e] = A[i * HD + j4 + e] * (gm[i][j4 + e] - rowv[i]);
  }
}

// out[n][h*32+i] = scale * sum_j M[u][i][j] * x[n][h*32+j]      (transposed: M[u][j][i])
// grid (token chunks of 128, image * heads); a thread owns 16 outputs of one token
__global__ __launch_bounds__(256) void chan_apply_kernel(const bf16* __restrict__ x, int ldx, const float* __restrict__ m,
                                                         int transposed, float scale, int rows_per_image, int heads,
                                                         bf16* __restrict__ out, int ldo) {
  __shared__ float ms[HD][HD];            // ms[j][i]: the 16 outputs of a thread read one broadcast row per j
  const int tid = threadIdx.x;
  const int u = blockIdx.y, img = u / heads, h = u - img * heads;
  const float* mu = m + (size_t)u * HD * HD;
  for (int e = tid; e < HD * HD; e += 256) {
    const int r = e >> 5, c = e & 31;     // element M[r][c]
    if (transposed) ms[r][c] = mu[e] * scale;      // out_i = sum_j M[j][i] x_j -> ms[j][i] = M[j][i]
    else ms[c][r] = mu[e] * scale;                 // out_i = sum_j M[i][j] x_j -> ms[j][i] = M[i][j]
  }
  __syncthreads();
  const int n = blockIdx.x * 128 + (tid >> 1), half = tid & 1;
  if (n >= rows_per_image) return;
  const size_t row = (size_t)img * rows_per_image + n;
  const bf16* xr = x + row * ldx + h * HD;
  float xv[HD];
#pragma unroll
  for (int d = 0; d < HD; d += 8) {
    const bf16x8 v = ldg16(xr + d);
#pragma unroll
    for (int e = 0; e < 8; ++e) xv[d + e] = bf2f(v[e]);
  }
  float acc[16];
#pragma unroll
  for (int e = 0; e < 16; ++e) acc[e] = 0.f;
#pragma unroll
  for (int j = 0; j < HD; ++j)
#pragma unroll
    for (int e = 0; e < 16; ++e) acc[e] = fmaf(ms[j][half * 16 + e], xv[j], acc[e]);
  bf16* orow = out + row * ldo + h * HD + half * 16;
  bf16x8 o0, o1;
#pragma unroll
  for (int e = 0; e < 8; ++e) { o0[e] = f2bf(acc[e]); o1[e] = f2bf(acc[8 + e]); }
  stg16(orow, o0);
  stg16(orow + 8, o1);
}

// out (+)= a + row_scale[row / rps] * b        (a, row_scale optional) — x + drop_path(f(norm(x))) and its backward
__global__ __launch_bounds__(256) void scale_rows_add_kernel(const bf16* __restrict__ a, const bf16* __restrict__ b,
                                                             const float* __restrict__ row_scale, int rps, bf16* out,
                                                             int accumulate, size_t n8, int cg) {
  for (size_t v = (size_t)blockIdx.x * 256 + threadIdx.x; v < n8; v += (size_t)gridDim.x * 256) {
    const float s = row_scale ? row_scale[(v / cg) / rps] : 1.f;
    const bf16x8 bv = ldg16(b + v * 8);
    bf16x8 av = zero8(), ov = zero8(), r;
    if (a) av = ldg16(a + v * 8);
    if (accumulate) ov = ldg16(out + v * 8);
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      float t = s * bf2f(bv[e]);
      if (a) t += bf2f(av[e]);
      if (accumulate) t += bf2f(ov[e]);
      r[e] = f2bf(t);
    }
    stg16(out + v * 8, r);
  }
}

}  // namespace

extern "C" int tok_chan_gram(const void* x, int ldx, const void* y, int ldy, int rows_per_image, int images, int heads,
                             float scale, int mode, const float* a_in, float* out, void* stream) {
  TOK_CHECK_ARG(x && y && out && rows_per_image > 0 && images > 0 && heads > 0 && ldx >= heads * HD && ldy >= heads * HD &&
                (ldx & 7) == 0 && (ldy & 7) == 0, "tok_chan_gram: bad args (head_dim is 32)");
  TOK_CHECK_ARG(mode >= 0 && mode <= 2 && (mode != 2 || a_in), "tok_chan_gram: mode 0..2 (2 needs the saved attention)");
  hipLaunchKernelGGL(chan_gram_kernel, dim3(images * heads), dim3(256), 0, tok_stream(stream), (const bf16*)x, ldx,
                     (const bf16*)y, ldy, rows_per_image, heads, scale, mode, a_in, out);
  TOK_CHECK_LAUNCH("tok_chan_gram");
  return TOK_OK;
}

extern "C" int tok_chan_apply(const void* x, int ldx, const float* m, int transposed, float scale, int rows_per_image,
                              int images, int heads, void* out, int ldo, void* stream) {
  TOK_CHECK_ARG(x && m && out && rows_per_image > 0 && images > 0 && heads > 0 && ldx >= heads * HD && ldo >= heads * HD &&
                (ldx & 7) == 0 && (ldo & 7) == 0, "tok_chan_apply: bad args (head_dim is 32)");
  TOK_CHECK_ARG((long long)images * heads <= 65535, "tok_chan_apply: images * heads exceeds 65535");
  hipLaunchKernelGGL(chan_apply_kernel, dim3(tok_cdiv(rows_per_image, 128), images * heads), dim3(256), 0,
                     tok_stream(stream), (const bf16*)x, ldx, m, transposed, scale, rows_per_image, heads, (bf16*)out, ldo);
  TOK_CHECK_LAUNCH("tok_chan_apply");
  return TOK_OK;
}

extern "C" int tok_scale_rows_add(const void* a, const void* b, const float* row_scale, int rows_per_sample, void* out,
                                  int accumulate, int64_t rows, int ld, void* stream) {
  TOK_CHECK_ARG(b && out && rows > 0 && ld > 0 && (ld & 7) == 0 && (!row_scale || rows_per_sample > 0),
                "tok_scale_rows_add: bad args");
  const size_t n8 = (size_t)rows * (ld >> 3);
  size_t blocks = (n8 + 255) / 256;
  if (blocks > 4096) blocks = 4096;
  hipLaunchKernelGGL(scale_rows_add_kernel, dim3((unsigned)blocks), dim3(256), 0, tok_stream(stream), (const bf16*)a,
                     (const bf16*)b, row_scale, rows_per_sample > 0 ? rows_per_sample : 1, (bf16*)out, accumulate, n8, ld >> 3);
  TOK_CHECK_LAUNCH("tok_scale_rows_add");
  return TOK_OK;
}

// ---- depthwise 3x3 / stride 1 / pad 1 convolution with bias (ConvPosEnc.proj, davit.py:101-106) -------------------------------
// NHWC bf16, weights fp32 [C][3][3] (the master itself: 9 values per channel), thread = (pixel, 8 channels).
namespace {

template <bool FLIP>   // FLIP: the data gradient (correlation -> convolution)
__global__ __launch_bounds__(256) void dwconv3x3_kernel(const bf16* __restrict__ x, const float* __restrict__ w,
                                                        const float* __restrict__ bias, bf16* out, int accumulate,
                                                        int N, int H, int W, int C, int ld) {
  const int cg_total = ld >> 3;
  const size_t total = (size_t)N * H * W * cg_total;
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (size_t)gridDim.x * 256) {
    const int cg = (int)(i % cg_total);
    size_t pix = i / cg_total;
    const int wq = (int)(pix % W);
    pix /= W;
    const int hq = (int)(pix % H);
    const int n = (int)(pix / H);
    float acc[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      const int c = cg * 8 + e;
      const float bv = bias != nullptr ? bias[c < C ? c : C - 1] : 0.f;   // unconditional (clamped) load, padding lanes stay 0
      acc[e] = c < C ? bv : 0.f;
    }
    // nine unconditional loads at clamped coordinates (a load inside a branch is fenced with s_waitcnt vmcnt(0) by hipcc: nine
    // dependent round trips per element); a tap outside the image multiplies a zero, which leaves the sum bit-identical
    bf16x8 v[9];
    bool tap_ok[9];
#pragma unroll
    for (int r = 0; r < 3; ++r)
#pragma unroll
      for (int s = 0; s < 3; ++s) {
        const int hh = hq + r - 1, ww = wq + s - 1;
        tap_ok[r * 3 + s] = (unsigned)hh < (unsigned)H && (unsigned)ww < (unsigned)W;
        const int hc = hh < 0 ? 0 : (hh >= H ? H - 1 : hh), wc = ww < 0 ? 0 : (ww >= W ? W - 1 : ww);
        v[r * 3 + s] = ldg16(x + (((size_t)n * H + hc) * W + wc) * ld + cg * 8);
      }
    float wt[8][9];                                // the 8 channels' filters: 24 more loads in the same batch
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      const int c = cg * 8 + e;
      const float* wc_ = w + (size_t)(c < C ? c : C - 1) * 9;
#pragma unroll
      for (int t = 0; t < 9; ++t) wt[e][t] = wc_[t];
    }
    __builtin_amdgcn_sched_barrier(0);             // (keeps the loads in flight together: the scheduler sank each to its use)
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      const int c = cg * 8 + e;
#pragma unroll
      for (int t = 0; t < 9; ++t) {
        const float wv = FLIP ? wt[e][8 - t] : wt[e][t];
        const float xv = tap_ok[t] ? bf2f(v[t][e]) : 0.f;
        if (c < C) acc[e] = fmaf(xv, wv, acc[e]);
      }
    }
    const size_t off = (((size_t)n * H + hq) * W + wq) * ld + cg * 8;
    bf16x8 o;
    if (accumulate) {
      const bf16x8 old = ldg16(out + off);
#pragma unroll
      for (int e = 0; e < 8; ++e) o[e] = f2bf(acc[e] + bf2f(old[e]));
    } else {
#pragma unroll
      for (int e = 0; e < 8; ++e) o[e] = f2bf(acc[e]);
    }
    stg16(out + off, o);
  }
}

// partial[block][c][10]: 9 tap sums of x * dout and the sum of dout (bias gradient); one block per row chunk
__global__ __launch_bounds__(256) void dwconv3x3_wgrad_partial_kernel(const bf16* __restrict__ x, const bf16* __restrict__ dout,
                                                                      int N, int H, int W, int C, int ld, int rows_per_block,
                                                                      float* __restrict__ partial) {
  // thread = (channel c = tid % Cb, row lane); channels beyond 256 handled by the cb loop
  const int total_rows = N * H;
  const int r0 = blockIdx.x * rows_per_block, r1 = min(total_rows, r0 + rows_per_block);
  for (int c = threadIdx.x; c < C; c += 256) {
    float acc[10];
#pragma unroll
    for (int t = 0; t < 10; ++t) acc[t] = 0.f;
    for (int row = r0; row < r1; ++row) {
      const int n = row / H, hq = row - n * H;
      for (int wq = 0; wq < W; ++wq) {
        const float g = bf2f(dout[(((size_t)n * H + hq) * W + wq) * ld + c]);
        acc[9] += g;
#pragma unroll
        for (int r = 0; r < 3; ++r) {
          const int hh = hq + r - 1;
          if ((unsigned)hh >= (unsigned)H) continue;
#pragma unroll
          for (int s = 0; s < 3; ++s) {
            const int ww = wq + s - 1;
            if ((unsigned)ww >= (unsigned)W) continue;
            acc[r * 3 + s] = fmaf(g, bf2f(x[(((size_t)n * H + hh) * W + ww) * ld + c]), acc[r * 3 + s]);
          }
        }
      }
    }
#pragma unroll
    for (int t = 0; t < 10; ++t) partial[((size_t)blockIdx.x * C + c) * 10 + t] = acc[t];
  }
}

__global__ __launch_bounds__(256) void dwconv3x3_wgrad_fold_kernel(const float* __restrict__ partial, int blocks, int C,
                                                                   float* dw, float* db, int accumulate) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= C * 10) return;
  const int c = i / 10, t = i - c * 10;
  double a = 0.0;
  for (int b = 0; b < blocks; ++b) a += (double)partial[((size_t)b * C + c) * 10 + t];
  float* dst = t < 9 ? (dw ? dw + c * 9 + t : nullptr) : (db ? db + c : nullptr);
  if (dst) *dst = (float)a + (accumulate ? *dst : 0.f);
}

}  // namespace

extern "C" int tok_dwconv3x3(const void* x, const float* w, const float* bias, void* out, int accumulate, int flip, int n, int h,
                             int wd, int c, int ld, void* stream) {
  TOK_CHECK_ARG(x && w && out && n > 0 && h > 0 && wd > 0 && c > 0 && ld >= c && (ld & 7) == 0, "tok_dwconv3x3: bad args");
  const size_t total = (size_t)n * h * wd * (ld >> 3);
  size_t blocks = (total + 255) / 256;
  if (blocks > 4096) blocks = 4096;
  if (flip) hipLaunchKernelGGL(dwconv3x3_kernel<true>, dim3((unsigned)blocks), dim3(256), 0, tok_stream(stream), (const bf16*)x, w,
                               bias, (bf16*)out, accumulate, n, h, wd, c, ld);
  else hipLaunchKernelGGL(dwconv3x3_kernel<false>, dim3((unsigned)blocks), dim3(256), 0, tok_stream(stream), (const bf16*)x, w,
                          bias, (bf16*)out, accumulate, n, h, wd, c, ld);
  TOK_CHECK_LAUNCH("tok_dwconv3x3");
  return TOK_OK;
}

extern "C" int tok_dwconv3x3_wgrad_blocks(int n, int h) { return n <= 0 || h <= 0 ? TOK_ERR_INVALID : (n * h < 1024 ? n * h : 1024); }

extern "C" int tok_dwconv3x3_wgrad(const void* x, const void* dout, int n, int h, int wd, int c, int ld, float* partial,
                                   float* dw, float* db, int accumulate, void* stream) {
  TOK_CHECK_ARG(x && dout && partial && (dw || db) && n > 0 && h > 0 && wd > 0 && c > 0 && ld >= c,
                "tok_dwconv3x3_wgrad: bad args");
  const int blocks = tok_dwconv3x3_wgrad_blocks(n, h);
  const int rpb = tok_cdiv(n * h, blocks);
  hipStream_t st = tok_stream(stream);
  hipLaunchKernelGGL(dwconv3x3_wgrad_partial_kernel, dim3(tok_cdiv(n * h, rpb)), dim3(256), 0, st, (const bf16*)x,
                     (const bf16*)dout, n, h, wd, c, ld, rpb, partial);
  TOK_CHECK_LAUNCH("tok_dwconv3x3_wgrad(partial)");
  hipLaunchKernelGGL(dwconv3x3_wgrad_fold_kernel, dim3(tok_cdiv(c * 10, 256)), dim3(256), 0, st, partial,
                     tok_cdiv(n * h, rpb), c, dw, db, accumulate);
  TOK_CHECK_LAUNCH("tok_dwconv3x3_wgrad(fold)");
  return TOK_OK;
}
