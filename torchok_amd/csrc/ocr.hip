// Object-contextual representations (torchok/models/heads/segmentation/ocr.py): the pixel <-> class products of
// SpatialGather_Module (:23-46) and ObjectAttentionBlock (:49-104).  K = number of classes (<= 64) is tiny, every tensor
// with a pixel axis is streamed once per launch, the class-side matrices ([K][C] per image) sit in LDS.
//   pix_class_matmul    out[b][n][k]  = scale * sum_c x[b][n][c] * m[b][k][c]                 (fp32 [B][N][K])
//   class_pix_expand    out[b][n][c] (+)= scale * sum_k w[b][n][k] * m[b][k][c]               (bf16 [B][N][ldo])
//   weighted_pool       out[b][k][c] (+)= scale * sum_n w[b][n][k] * x[b][n][c]               (bf16 [B][K][ldo], fixed order)
//   softmax_rows / _bwd over the K classes of a pixel; softmax_cols / _bwd over the N pixels of an (image, class)
//   channel_scale       out[b][n][c] = x[b][n][c] * s[b][c]                                   (Dropout2d)
#include "tok_common.h"
#include <math.h>

namespace {

constexpr int KMAX = 64;

// one wavefront per pixel: lanes stride the channels, K dot products reduced with butterflies
__global__ __launch_bounds__(256) void pix_class_matmul_kernel(const bf16* __restrict__ x, int ldx,
                                                               const bf16* __restrict__ m, int ldm, int N, int K, int C,
                                                               float scale, float* __restrict__ out) {
  extern __shared__ float ms[];                    // [K][C]
  const int b = blockIdx.y;
  const bf16* mb = m + (size_t)b * K * ldm;
  for (int i = threadIdx.x; i < K * C; i += 256) ms[i] = bf2f(mb[(size_t)(i / C) * ldm + (i % C)]);
  __syncthreads();
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  for (int n = blockIdx.x * 4 + wv; n < N; n += gridDim.x * 4) {
    const bf16* xr = x + ((size_t)b * N + n) * ldx;
    float acc[KMAX];
#pragma unroll
    for (int k = 0; k < KMAX; ++k) acc[k] = 0.f;
    for (int c = lane; c < C; c += 64) {
      const float xv = bf2f(xr[c]);
#pragma unroll
      for (int k = 0; k < KMAX; ++k)
        if (k < K) acc[k] = fmaf(xv, ms[k * C + c], acc[k]);
    }
    float* o = out + ((size_t)b * N + n) * K;
#pragma unroll
    for (int k = 0; k < KMAX; ++k)
      if (k < K) {
        const float v = wave_sum(acc[k]);
        if (lane == 0) o[k] = v * scale;
      }
  }
}

// thread = (pixel, 8 channels)
__global__ __launch_bounds__(256) void class_pix_expand_kernel(const float* __restrict__ w, const bf16* __restrict__ m,
                                                               int ldm, int N, int K, int C, float scale, bf16* out,
                                                               int ldo, int accumulate) {
  extern __shared__ float ms[];                    // [K][Cp8]
  const int b = blockIdx.y;
  const int cg_total = (C + 7) >> 3, cp = cg_total * 8;
  const bf16* mb = m + (size_t)b * K * ldm;
  for (int i = threadIdx.x; i < K * cp; i += 256) {
    const int k = i / cp, c = i % cp;
    ms[i] = c < C ? bf2f(mb[(size_t)k * ldm + c]) * scale : 0.f;
  }
  __syncthreads();
  const size_t total = (size_t)N * cg_total;
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (size_t)gridDim.x * 256) {
    const int cg = (int)(i % cg_total);
    const size_t n = i / cg_total;
    const float* wr = w + ((size_t)b * N + n) * K;
    float acc[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) acc[e] = 0.f;
    for (int k = 0; k < K; ++k) {
      const float wk = wr[k];
#pragma unroll
      for (int e = 0; e < 8; ++e) acc[e] = fmaf(wk, ms[k * cp + cg * 8 + e], acc[e]);
    }
    bf16* o = out + ((size_t)b * N + n) * ldo + cg * 8;
    bf16x8 r;
    if (accumulate) {
      const bf16x8 old = ldg16(o);
#pragma unroll
      for (int e = 0; e < 8; ++e) r[e] = f2bf(acc[e] + bf2f(old[e]));
    } else {
#pragma unroll
      for (int e = 0; e < 8; ++e) r[e] = f2bf(acc[e]);
    }
    stg16(o, r);
  }
}

// partial[b][chunk][k][c]: block (chunk, b) folds its pixel chunk; thread owns the (k, c) pairs tid, tid + 256, ...
__global__ __launch_bounds__(256) void weighted_pool_partial_kernel(const float* __restrict__ w, const bf16* __restrict__ x,
                                                                    int ldx, int N, int K, int C, int chunk, int stage,
                                                                    float* __restrict__ partial) {
  extern __shared__ float sm[];                    // xs[stage][C] then ws[stage][K]
  float* xs = sm;
  float* ws = sm + stage * C;
  const int b = blockIdx.y;
  const int n0 = blockIdx.x * chunk, n1 = min(N, n0 + chunk);
  const int kc = K * C;
  constexpr int MAXP = 40;                         // K * C <= 10240 (19 classes x 512 channels of HRNet-OCR)
  float acc[MAXP];
#pragma unroll
  for (int p = 0; p < MAXP; ++p) acc[p] = 0.f;
  for (int nb = n0; nb < n1; nb += stage) {
    const int rows = min(stage, n1 - nb);
    for (int i = threadIdx.x; i < rows * C; i += 256)
      xs[i] = bf2f(x[((size_t)b * N + nb + i / C) * ldx + (i % C)]);
    for (int i = threadIdx.x; i < rows * K; i += 256) ws[i] = w[((size_t)b * N + nb) * K + i];
    __syncthreads();
#pragma unroll
    for (int p = 0; p < MAXP; ++p) {
      const int idx = threadIdx.x + p * 256;
      if (idx < kc) {
        const int k = idx / C, c = idx - k * C;
        float a = acc[p];
        for (int r = 0; r < rows; ++r) a = fmaf(ws[r * K + k], xs[r * C + c], a);
        acc[p] = a;
      }
    }
    __syncthreads();
  }
  float* o = partial + ((size_t)b * gridDim.x + blockIdx.x) * kc;
#pragma unroll
  for (int p = 0; p < MAXP; ++p) {
    const int idx = threadIdx.x + p * 256;
    if (idx < kc) o[idx] = acc[p];
  }
}

__global__ __launch_bounds__(256) void weighted_pool_fold_kernel(const float* __restrict__ partial, int chunks, int K, int C,
                                                                 float scale, bf16* out, int ldo, int accumulate) {
  const int b = blockIdx.y;
  const int idx = blockIdx.x * 256 + threadIdx.x;
  if (idx >= K * C) return;
  double a = 0.0;
  for (int ch = 0; ch < chunks; ++ch) a += (double)partial[((size_t)b * chunks + ch) * K * C + idx];
  const int k = idx / C, c = idx - k * C;
  bf16* o = out + ((size_t)b * K + k) * ldo + c;
  const float v = (float)a * scale + (accumulate ? bf2f(*o) : 0.f);
  *o = f2bf(v);
}

// softmax over the K entries of a row (thread per row)
__global__ __launch_bounds__(256) void softmax_rows_kernel(const float* __restrict__ x, size_t rows, int K,
                                                           float* __restrict__ out) {
  for (size_t r = (size_t)blockIdx.x * 256 + threadIdx.x; r < rows; r += (size_t)gridDim.x * 256) {
    const float* xr = x + r * K;
    float mx = -INFINITY;
    for (int k = 0; k < K; ++k) mx = fmaxf(mx, xr[k]);
    float s = 0.f;
    for (int k = 0; k < K; ++k) s += expf(xr[k] - mx);
    const float inv = 1.f / s;
    for (int k = 0; k < K; ++k) out[r * K + k] = expf(xr[k] - mx) * inv;
  }
}

__global__ __launch_bounds__(256) void softmax_rows_bwd_kernel(const float* __restrict__ p, const float* __restrict__ dp,
                                                               size_t rows, int K, float* __restrict__ dx) {
  for (size_t r = (size_t)blockIdx.x * 256 + threadIdx.x; r < rows; r += (size_t)gridDim.x * 256) {
    float dot = 0.f;
    for (int k = 0; k < K; ++k) dot = fmaf(p[r * K + k], dp[r * K + k], dot);
    for (int k = 0; k < K; ++k) dx[r * K + k] = p[r * K + k] * (dp[r * K + k] - dot);
  }
}

// softmax over the N pixels of every (image, class): one block per image, classes in the low lane bits
__global__ __launch_bounds__(256) void softmax_cols_kernel(const bf16* __restrict__ logits, int ld, int N, int K,
                                                           float scale, float* __restrict__ p) {
  __shared__ float red[256];
  __shared__ float mxs[KMAX], inv[KMAX];
  const int b = blockIdx.x, tid = threadIdx.x;
  const bf16* lb = logits + (size_t)b * N * ld;
  for (int k = 0; k < K; ++k) {
    float mx = -INFINITY;
    for (int n = tid; n < N; n += 256) mx = fmaxf(mx, scale * bf2f(lb[(size_t)n * ld + k]));
    red[tid] = mx;
    __syncthreads();
    for (int s = 128; s > 0; s >>= 1) { if (tid < s) red[tid] = fmaxf(red[tid], red[tid + s]); __syncthreads(); }
    if (tid == 0) mxs[k] = red[0];
    __syncthreads();
    float sum = 0.f;
    for (int n = tid; n < N; n += 256) sum += expf(scale * bf2f(lb[(size_t)n * ld + k]) - mxs[k]);
    red[tid] = sum;
    __syncthreads();
    for (int s = 128; s > 0; s >>= 1) { if (tid < s) red[tid] += red[tid + s]; __syncthreads(); }
    if (tid == 0) inv[k] = 1.f / red[0];
    __syncthreads();
  }
  for (size_t i = tid; i < (size_t)N * K; i += 256) {
    const int k = (int)(i % K);
    const size_t n = i / K;
    p[((size_t)b * N + n) * K + k] = expf(scale * bf2f(lb[n * ld + k]) - mxs[k]) * inv[k];
  }
}

// dlogits[b][n][k] (+)= scale * p * (dp - sum_n p * dp)
__global__ __launch_bounds__(256) void softmax_cols_bwd_kernel(const float* __restrict__ p, const float* __restrict__ dp,
                                                               int N, int K, float scale, bf16* dlogits, int ld,
                                                               int accumulate) {
  __shared__ float red[256];
  __shared__ float dots[KMAX];
  const int b = blockIdx.x, tid = threadIdx.x;
  const float* pb = p + (size_t)b * N * K;
  const float* db = dp + (size_t)b * N * K;
  for (int k = 0; k < K; ++k) {
    float s = 0.f;
    for (int n = tid; n < N; n += 256) s = fmaf(pb[(size_t)n * K + k], db[(size_t)n * K + k], s);
    red[tid] = s;
    __syncthreads();
    for (int st = 128; st > 0; st >>= 1) { if (tid < st) red[tid] += red[tid + st]; __syncthreads(); }
    if (tid == 0) dots[k] = red[0];
    __syncthreads();
  }
  bf16* dl = dlogits + (size_t)b * N * ld;
  for (size_t i = tid; i < (size_t)N * ld; i += 256) {
    const int k = (int)(i % ld);
    const size_t n = i / ld;
    float v = 0.f;
    if (k < K) v = scale * pb[n * K + k] * (db[n * K + k] - dots[k]);
    if (accumulate) v += bf2f(dl[i]);
    dl[i] = f2bf(v);
  }
}

__global__ __launch_bounds__(256) void channel_scale_kernel(const bf16* __restrict__ x, const float* __restrict__ s,
                                                            bf16* out, int accumulate, int N, int C, int ld) {
  const int b = blockIdx.y;
  const int cg_total = ld >> 3;
  const size_t total = (size_t)N * cg_total;
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (size_t)gridDim.x * 256) {
    const int cg = (int)(i % cg_total);
    const size_t off = ((size_t)b * N + i / cg_total) * ld + cg * 8;
    const bf16x8 v = ldg16(x + off);
    bf16x8 o;
    bf16x8 old = accumulate ? ldg16(out + off) : zero8();
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      const int c = cg * 8 + e;
      o[e] = f2bf(bf2f(v[e]) * (c < C ? s[(size_t)b * C + c] : 0.f) + bf2f(old[e]));
    }
    stg16(out + off, o);
  }
}

inline unsigned blocks_for(size_t n, unsigned cap = 2048) {
  size_t b = (n + 255) / 256;
  return (unsigned)(b > cap ? cap : (b < 1 ? 1 : b));
}

}  // namespace

extern "C" int tok_pix_class_matmul(const void* x, int ldx, const void* m, int ldm, int images, int n, int k, int c, float scale,
                                    float* out, void* stream) {
  TOK_CHECK_ARG(x && m && out && images > 0 && n > 0 && k > 0 && k <= KMAX && c > 0 && ldx >= c && ldm >= c,
                "tok_pix_class_matmul: bad args (at most 64 classes)");
  const size_t smem = (size_t)k * c * sizeof(float);
  TOK_CHECK_ARG(smem <= 64 * 1024 && images <= 65535, "tok_pix_class_matmul: class matrix of %d x %d does not fit LDS", k, c);
  hipLaunchKernelGGL(pix_class_matmul_kernel, dim3(blocks_for((size_t)n * 64, 1024), images), dim3(256), smem,
                     tok_stream(stream), (const bf16*)x, ldx, (const bf16*)m, ldm, n, k, c, scale, out);
  TOK_CHECK_LAUNCH("tok_pix_class_matmul");
  return TOK_OK;
}

extern "C" int tok_class_pix_expand(const float* w, const void* m, int ldm, int images, int n, int k, int c, float scale,
                                    void* out, int ldo, int accumulate, void* stream) {
  TOK_CHECK_ARG(w && m && out && images > 0 && n > 0 && k > 0 && k <= KMAX && c > 0 && ldm >= c && ldo >= ((c + 7) & ~7) &&
                (ldo & 7) == 0, "tok_class_pix_expand: bad args (at most 64 classes)");
  const size_t smem = (size_t)k * ((c + 7) & ~7) * sizeof(float);
  TOK_CHECK_ARG(smem <= 64 * 1024 && images <= 65535, "tok_class_pix_expand: class matrix of %d x %d does not fit LDS", k, c);
  hipLaunchKernelGGL(class_pix_expand_kernel, dim3(blocks_for((size_t)n * ((c + 7) >> 3), 1024), images), dim3(256), smem,
                     tok_stream(stream), w, (const bf16*)m, ldm, n, k, c, scale, (bf16*)out, ldo, accumulate);
  TOK_CHECK_LAUNCH("tok_class_pix_expand");
  return TOK_OK;
}

extern "C" int tok_weighted_pool_chunks(int n) { return n <= 0 ? TOK_ERR_INVALID : (n + 511) / 512 < 1 ? 1 : (n + 511) / 512; }

extern "C" int tok_weighted_pool(const float* w, const void* x, int ldx, int images, int n, int k, int c, float scale,
                                 float* partial, void* out, int ldo, int accumulate, void* stream) {
  TOK_CHECK_ARG(w && x && partial && out && images > 0 && n > 0 && k > 0 && k <= KMAX && c > 0 && ldx >= c && ldo >= c,
                "tok_weighted_pool: bad args (at most 64 classes)");
  TOK_CHECK_ARG(k * c <= 10240 && images <= 65535, "tok_weighted_pool: K * C = %d exceeds 10240", k * c);
  const int chunks = tok_weighted_pool_chunks(n);
  const int stage = c <= 256 ? 32 : 16;           // pixel rows staged in LDS per pass (<= 64 KB)
  const size_t smem = (size_t)stage * (c + k) * sizeof(float);
  TOK_CHECK_ARG(smem <= 64 * 1024, "tok_weighted_pool: %d channels do not fit the LDS stage", c);
  hipStream_t st = tok_stream(stream);
  hipLaunchKernelGGL(weighted_pool_partial_kernel, dim3(chunks, images), dim3(256), smem, st, w, (const bf16*)x, ldx, n, k, c,
                     512, stage, partial);
  TOK_CHECK_LAUNCH("tok_weighted_pool(partial)");
  hipLaunchKernelGGL(weighted_pool_fold_kernel, dim3((k * c + 255) / 256, images), dim3(256), 0, st, partial, chunks, k, c,
                     scale, (bf16*)out, ldo, accumulate);
  TOK_CHECK_LAUNCH("tok_weighted_pool(fold)");
  return TOK_OK;
}

extern "C" int tok_softmax_rows_f32(const float* x, int64_t rows, int k, float* out, void* stream) {
  TOK_CHECK_ARG(x && out && rows > 0 && k > 0, "tok_softmax_rows_f32: bad args");
  hipLaunchKernelGGL(softmax_rows_kernel, dim3(blocks_for((size_t)rows)), dim3(256), 0, tok_stream(stream), x, (size_t)rows, k,
                     out);
  TOK_CHECK_LAUNCH("tok_softmax_rows_f32");
  return TOK_OK;
}

extern "C" int tok_softmax_rows_bwd_f32(const float* p, const float* dp, int64_t rows, int k, float* dx, void* stream) {
  TOK_CHECK_ARG(p && dp && dx && rows > 0 && k > 0, "tok_softmax_rows_bwd_f32: bad args");
  hipLaunchKernelGGL(softmax_rows_bwd_kernel, dim3(blocks_for((size_t)rows)), dim3(256), 0, tok_stream(stream), p, dp,
                     (size_t)rows, k, dx);
  TOK_CHECK_LAUNCH("tok_softmax_rows_bwd_f32");
  return TOK_OK;
}

extern "C" int tok_softmax_cols_fwd(const void* logits, int ld, int images, int n, int k, float scale, float* p,
                                    void* stream) {
  TOK_CHECK_ARG(logits && p && images > 0 && n > 0 && k > 0 && k <= KMAX && ld >= k, "tok_softmax_cols_fwd: bad args");
  hipLaunchKernelGGL(softmax_cols_kernel, dim3(images), dim3(256), 0, tok_stream(stream), (const bf16*)logits, ld, n, k,
                     scale, p);
  TOK_CHECK_LAUNCH("tok_softmax_cols_fwd");
  return TOK_OK;
}

extern "C" int tok_softmax_cols_bwd(const float* p, const float* dp, int images, int n, int k, float scale, void* dlogits,
                                    int ld, int accumulate, void* stream) {
  TOK_CHECK_ARG(p && dp && dlogits && images > 0 && n > 0 && k > 0 && k <= KMAX && ld >= k, "tok_softmax_cols_bwd: bad args");
  hipLaunchKernelGGL(softmax_cols_bwd_kernel, dim3(images), dim3(256), 0, tok_stream(stream), p, dp, n, k, scale,
                     (bf16*)dlogits, ld, accumulate);
  TOK_CHECK_LAUNCH("tok_softmax_cols_bwd");
  return TOK_OK;
}

extern "C" int tok_channel_scale(const void* x, const float* s, void* out, int accumulate, int images, int n, int c, int ld,
                                 void* stream) {
  TOK_CHECK_ARG(x && s && out && images > 0 && images <= 65535 && n > 0 && c > 0 && ld >= c && (ld & 7) == 0,
                "tok_channel_scale: bad args");
  hipLaunchKernelGGL(channel_scale_kernel, dim3(blocks_for((size_t)n * (ld >> 3), 1024), images), dim3(256), 0,
                     tok_stream(stream), (const bf16*)x, s, (bf16*)out, accumulate, n, c, ld);
  TOK_CHECK_LAUNCH("tok_channel_scale");
  return TOK_OK;
}
