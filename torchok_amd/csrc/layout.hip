// Layout / dtype plumbing kernels: image NCHW -> NHWC bf16, fp32 -> bf16 shadows, weight packs.
#include "tok_common.h"
#include <hip/hip_fp16.h>

namespace {

template <typename T> __device__ __forceinline__ float to_f32(T v);
template <> __device__ __forceinline__ float to_f32<float>(float v) { return v; }
template <> __device__ __forceinline__ float to_f32<__half>(__half v) { return __half2float(v); }
template <> __device__ __forceinline__ float to_f32<bf16>(bf16 v) { return (float)v; }

// thread = (pixel, group of up to 8 output channels); reads are coalesced along W per channel plane
template <typename T, int G>
__global__ __launch_bounds__(256) void nchw_to_nhwc_kernel(const T* __restrict__ src, bf16* __restrict__ dst,
                                                           int N, int C, int HW, int c_pad) {
  const int groups = c_pad / G;
  const size_t total = (size_t)N * HW * groups;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total;
       i += (size_t)gridDim.x * blockDim.x) {
    const size_t pix = i % ((size_t)N * HW);   // pixel fastest: coalesced plane reads
    const int gidx = (int)(i / ((size_t)N * HW));
    const int n = (int)(pix / HW);
    const int hw = (int)(pix - (size_t)n * HW);
    bf16 o[G];
#pragma unroll
    for (int e = 0; e < G; ++e) {
      const int c = gidx * G + e;
      o[e] = (c < C) ? f2bf(to_f32<T>(src[((size_t)n * C + c) * HW + hw])) : (bf16)0.f;
    }
    bf16* d = dst + pix * c_pad + gidx * G;
    if (G == 8) {
      bf16x8 v;
#pragma unroll
      for (int e = 0; e < 8; ++e) v[e] = o[e % G];
      stg16(d, v);
    } else {
      bf16x4 v;
#pragma unroll
      for (int e = 0; e < 4; ++e) v[e] = o[e % G];
      *reinterpret_cast<bf16x4*>(d) = v;
    }
  }
}

__global__ __launch_bounds__(256) void cast_kernel(const float* __restrict__ src, bf16* __restrict__ dst, size_t n) {
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x)
    dst[i] = f2bf(src[i]);
}

__global__ __launch_bounds__(256) void pack_fwd_kernel(const float* __restrict__ src, int k, int r, int s, int c,
                                                       bf16* __restrict__ dst, int k_pad, int s_pad, int c_pad) {
  const size_t total = (size_t)k_pad * r * s_pad * c_pad;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total;
       i += (size_t)gridDim.x * blockDim.x) {
    const int cc = (int)(i % c_pad);
    size_t rest = i / c_pad;
    const int ss = (int)(rest % s_pad);
    rest /= s_pad;
    const int rr = (int)(rest % r);
    const int kk = (int)(rest / r);
    float v = 0.f;
    if (kk < k && ss < s && cc < c) v = src[(((size_t)kk * r + rr) * s + ss) * c + cc];
    dst[i] = f2bf(v);
  }
}

// dst[c][r'][s'][k] = src[k][R-1-r'][S-1-s'][c]
__global__ __launch_bounds__(256) void pack_dgrad_kernel(const float* __restrict__ src, int k, int r, int s, int c,
                                                         bf16* __restrict__ dst, int k_pad, int c_pad) {
  const size_t total = (size_t)c_pad * r * s * k_pad;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total;
       i += (size_t)gridDim.x * blockDim.x) {
    const int kk = (int)(i % k_pad);
    size_t rest = i / k_pad;
    const int ss = (int)(rest % s);
    rest /= s;
    const int rr = (int)(rest % r);
    const int cc = (int)(rest / r);
    float v = 0.f;
    if (kk < k && cc < c) v = src[(((size_t)kk * r + (r - 1 - rr)) * s + (s - 1 - ss)) * c + cc];
    dst[i] = f2bf(v);
  }
}

// both MFMA operand packs of one fp32 master in ONE launch (grid covers max of the two index spaces)
__global__ __launch_bounds__(256) void pack_both_kernel(const float* __restrict__ src, int k, int r, int s, int c,
                                                        bf16* __restrict__ fwd, int k_pad, int s_pad, int c_pad,
                                                        bf16* __restrict__ dgr) {
  const size_t total_f = (size_t)k_pad * r * s_pad * c_pad;
  const size_t total_d = (size_t)c_pad * r * s * k_pad;
  const size_t total = total_f > total_d ? total_f : total_d;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total;
       i += (size_t)gridDim.x * blockDim.x) {
    if (i < total_f) {
      const int cc = (int)(i % c_pad);
      size_t rest = i / c_pad;
      const int ss = (int)(rest % s_pad);
      rest /= s_pad;
      const int rr = (int)(rest % r);
      const int kk = (int)(rest / r);
      float v = 0.f;
      if (kk < k && ss < s && cc < c) v = src[(((size_t)kk * r + rr) * s + ss) * c + cc];
      fwd[i] = f2bf(v);
    }
    if (i < total_d) {
      const int kk = (int)(i % k_pad);
      size_t rest = i / k_pad;
      const int ss = (int)(rest % s);
      rest /= s;
      const int rr = (int)(rest % r);
      const int cc = (int)(rest / r);
      float v = 0.f;
      if (kk < k && cc < c) v = src[(((size_t)kk * r + (r - 1 - rr)) * s + (s - 1 - ss)) * c + cc];
      dgr[i] = f2bf(v);
    }
  }
}

// every weight of a model in ONE launch: block b serves item i (largest block_start <= b) and one 32 (k) x 32 (c) tile of
// one filter tap (r, s) of it.  The tile is read once from the fp32 master in 128-byte row segments, goes through LDS, and
// leaves as the forward pack [k][r][s_pad][c_pad] (rows of c) AND the tap-flipped transposed dgrad pack [c][r][s][k_pad]
// (rows of k) in 64-byte segments — the element-wise version gathered the dgrad pack with one cache line per lane.
__device__ __forceinline__ int pack_tiles_k(const tok_pack_item& it) { return (it.k_pad + 31) >> 5; }
__device__ __forceinline__ int pack_tiles_c(const tok_pack_item& it) { return (it.c_pad + 31) >> 5; }

__global__ __launch_bounds__(256) void pack_batched_kernel(const tok_pack_item* __restrict__ items, int n_items) {
  __shared__ float tile[32][33];
  int lo = 0, hi = n_items - 1;
  const int b = blockIdx.x;
  while (lo < hi) {
    const int mid = (lo + hi + 1) >> 1;
    if (items[mid].block_start <= b) lo = mid; else hi = mid - 1;
  }
  const tok_pack_item it = items[lo];
  const float* __restrict__ src = it.src;
  bf16* __restrict__ fwd = (bf16*)it.dst_fwd;
  bf16* __restrict__ dgr = (bf16*)it.dst_dgrad;
  const int k = it.k, r = it.r, s = it.s, c = it.c, k_pad = it.k_pad, s_pad = it.s_pad, c_pad = it.c_pad;
  const int tk = pack_tiles_k(it), tc = pack_tiles_c(it);
  int t = b - it.block_start;                 // ((rr * s_pad + ss) * tk + kt) * tc + ct
  const int ct = t % tc; t /= tc;
  const int kt = t % tk; t /= tk;
  const int ss = t % s_pad, rr = t / s_pad;
  const int k0 = kt * 32, c0 = ct * 32;
  const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;    // 32 x 8
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const int kk = k0 + ty + 8 * j, cc = c0 + tx;
    float v = 0.f;
    if (kk < k && ss < s && cc < c) v = src[(((size_t)kk * r + rr) * s + ss) * c + cc];
    tile[ty + 8 * j][tx] = v;
  }
  __syncthreads();
  if (fwd != nullptr) {
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int kk = k0 + ty + 8 * j, cc = c0 + tx;
      if (kk < k_pad && cc < c_pad) fwd[(((size_t)kk * r + rr) * s_pad + ss) * c_pad + cc] = f2bf(tile[ty + 8 * j][tx]);
    }
  }
  if (dgr != nullptr && ss < s) {
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int cc = c0 + ty + 8 * j, kk = k0 + tx;
      if (cc < c_pad && kk < k_pad)
        dgr[(((size_t)cc * r + (r - 1 - rr)) * s + (s - 1 - ss)) * k_pad + kk] = f2bf(tile[tx][ty + 8 * j]);
    }
  }
}

inline int grid_for(size_t total) {
  size_t b = (total + 255) / 256;
  return (int)(b > 4096 ? 4096 : (b < 1 ? 1 : b));
}

template <typename T>
void launch_nchw(const void* src, void* dst, int n, int c, int hw, int c_pad, hipStream_t st) {
  if (c_pad % 8 == 0) {
    hipLaunchKernelGGL((nchw_to_nhwc_kernel<T, 8>), dim3(grid_for((size_t)n * hw * (c_pad / 8))), dim3(256), 0,
                       st, (const T*)src, (bf16*)dst, n, c, hw, c_pad);
  } else {
    hipLaunchKernelGGL((nchw_to_nhwc_kernel<T, 4>), dim3(grid_for((size_t)n * hw * (c_pad / 4))), dim3(256), 0,
                       st, (const T*)src, (bf16*)dst, n, c, hw, c_pad);
  }
}

}  // namespace

extern "C" int tok_nchw_to_nhwc_bf16(const void* src, int src_dtype, int n, int c, int h, int w, void* dst,
                                     int c_pad, void* stream) {
  TOK_CHECK_ARG(src && dst && n > 0 && c > 0 && h > 0 && w > 0, "tok_nchw_to_nhwc_bf16: bad args");
  TOK_CHECK_ARG(c_pad >= c && c_pad % 4 == 0, "tok_nchw_to_nhwc_bf16: c_pad=%d must be >= c and a multiple of 4", c_pad);
  hipStream_t st = tok_stream(stream);
  switch (src_dtype) {
    case TOK_F32: launch_nchw<float>(src, dst, n, c, h * w, c_pad, st); break;
    case TOK_F16: launch_nchw<__half>(src, dst, n, c, h * w, c_pad, st); break;
    case TOK_BF16: launch_nchw<bf16>(src, dst, n, c, h * w, c_pad, st); break;
    default: tok_set_error("tok_nchw_to_nhwc_bf16: unknown dtype %d", src_dtype); return TOK_ERR_INVALID;
  }
  TOK_CHECK_LAUNCH("tok_nchw_to_nhwc_bf16");
  return TOK_OK;
}

extern "C" int tok_cast_f32_bf16(const float* src, void* dst, size_t count, void* stream) {
  TOK_CHECK_ARG(src && dst && count > 0, "tok_cast_f32_bf16: bad args");
  hipLaunchKernelGGL(cast_kernel, dim3(grid_for(count)), dim3(256), 0, tok_stream(stream), src, (bf16*)dst, count);
  TOK_CHECK_LAUNCH("tok_cast_f32_bf16");
  return TOK_OK;
}

namespace {
__global__ __launch_bounds__(256) void widen_kernel(const bf16* __restrict__ src, float* __restrict__ dst, float scale, size_t n) {
  const size_t n8 = n >> 3;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n8; i += (size_t)gridDim.x * blockDim.x) {
    const bf16x8 v = ldg16(src + i * 8);
    f32x4 lo, hi;
#pragma unroll
    for (int e = 0; e < 4; ++e) { lo[e] = bf2f(v[e]) * scale; hi[e] = bf2f(v[4 + e]) * scale; }
    reinterpret_cast<f32x4*>(dst)[2 * i] = lo;
    reinterpret_cast<f32x4*>(dst)[2 * i + 1] = hi;
  }
  if (blockIdx.x == 0 && threadIdx.x < (n & 7)) dst[n8 * 8 + threadIdx.x] = bf2f(src[n8 * 8 + threadIdx.x]) * scale;
}
}  // namespace

extern "C" int tok_cast_bf16_f32(const void* src, float* dst, float scale, size_t count, void* stream) {
  TOK_CHECK_ARG(src && dst && count > 0, "tok_cast_bf16_f32: bad args");
  TOK_CHECK_ARG((((uintptr_t)src | (uintptr_t)dst) & 15) == 0, "tok_cast_bf16_f32: 16-byte aligned buffers");
  hipLaunchKernelGGL(widen_kernel, dim3(grid_for(count / 8 + 1)), dim3(256), 0, tok_stream(stream), (const bf16*)src, dst,
                     scale, count);
  TOK_CHECK_LAUNCH("tok_cast_bf16_f32");
  return TOK_OK;
}

// ---- stride-2 pixel subsample (the input of a 1x1 / stride-2 projection as a dense tensor) ---------------------------------
// out[b][p][q][:] = x[b][2p][2q][:],  P = ceil(H/2), Q = ceil(W/2): the projection then runs as a plain pointwise GEMM in
// forward, weight- and data-gradient (conv_igemm's parity-class path streams at 1.2-2.4 TB/s on these layers).
namespace {
__global__ __launch_bounds__(256) void subsample2_fwd_kernel(const bf16* __restrict__ x, bf16* __restrict__ out, int h, int w,
                                                            int p, int q, int c8, size_t total) {
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
    const int cc = (int)(i % c8);
    size_t pix = i / c8;
    const int qq = (int)(pix % q);
    pix /= q;
    const int pp = (int)(pix % p);
    const size_t b = pix / p;
    stg16(out + i * 8, ldg16(x + (((b * h + 2 * pp) * w + 2 * qq) * (size_t)c8 + cc) * 8));
  }
}
// dx[b][hh][ww][:] (+)= (hh, ww both even) ? dsub[b][hh/2][ww/2][:] : 0
__global__ __launch_bounds__(256) void subsample2_bwd_kernel(const bf16* __restrict__ dsub, bf16* __restrict__ dx, int h, int w,
                                                            int p, int q, int c8, int accumulate, size_t total) {
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
    const int cc = (int)(i % c8);
    size_t pix = i / c8;
    const int ww = (int)(pix % w);
    pix /= w;
    const int hh = (int)(pix % h);
    const size_t b = pix / h;
    const bool hit = ((hh | ww) & 1) == 0;
    if (!hit) {
      if (!accumulate) stg16(dx + i * 8, zero8());
      continue;
    }
    bf16x8 v = ldg16(dsub + (((b * p + (hh >> 1)) * q + (ww >> 1)) * (size_t)c8 + cc) * 8);
    if (accumulate) {
      const bf16x8 old = ldg16(dx + i * 8);
#pragma unroll
      for (int e = 0; e < 8; ++e) v[e] = f2bf(bf2f(old[e]) + bf2f(v[e]));
    }
    stg16(dx + i * 8, v);
  }
}
}  // namespace

extern "C" int tok_subsample2_fwd(const void* x, int n, int h, int w, int c, void* out, void* stream) {
  TOK_CHECK_ARG(x && out && n > 0 && h > 0 && w > 0 && c > 0 && c % 8 == 0, "tok_subsample2_fwd: bad args (c %% 8 == 0)");
  const int p = (h + 1) / 2, q = (w + 1) / 2;
  const size_t total = (size_t)n * p * q * (c / 8);
  hipLaunchKernelGGL(subsample2_fwd_kernel, dim3(grid_for(total)), dim3(256), 0, tok_stream(stream), (const bf16*)x, (bf16*)out, h,
                     w, p, q, c / 8, total);
  TOK_CHECK_LAUNCH("tok_subsample2_fwd");
  return TOK_OK;
}

extern "C" int tok_subsample2_bwd(const void* dsub, int n, int h, int w, int c, void* dx, int accumulate, void* stream) {
  TOK_CHECK_ARG(dsub && dx && n > 0 && h > 0 && w > 0 && c > 0 && c % 8 == 0, "tok_subsample2_bwd: bad args (c %% 8 == 0)");
  const int p = (h + 1) / 2, q = (w + 1) / 2;
  const size_t total = (size_t)n * h * w * (c / 8);
  hipLaunchKernelGGL(subsample2_bwd_kernel, dim3(grid_for(total)), dim3(256), 0, tok_stream(stream), (const bf16*)dsub, (bf16*)dx,
                     h, w, p, q, c / 8, accumulate, total);
  TOK_CHECK_LAUNCH("tok_subsample2_bwd");
  return TOK_OK;
}

extern "C" int tok_pack_weight_fwd(const float* src, int k, int r, int s, int c, void* dst, int k_pad,
                                   int s_pad, int c_pad, void* stream) {
  TOK_CHECK_ARG(src && dst && k > 0 && r > 0 && s > 0 && c > 0, "tok_pack_weight_fwd: bad args");
  TOK_CHECK_ARG(k_pad >= k && s_pad >= s && c_pad >= c, "tok_pack_weight_fwd: pads smaller than dims");
  hipLaunchKernelGGL(pack_fwd_kernel, dim3(grid_for((size_t)k_pad * r * s_pad * c_pad)), dim3(256), 0,
                     tok_stream(stream), src, k, r, s, c, (bf16*)dst, k_pad, s_pad, c_pad);
  TOK_CHECK_LAUNCH("tok_pack_weight_fwd");
  return TOK_OK;
}

extern "C" int tok_pack_weight_dgrad(const float* src, int k, int r, int s, int c, void* dst, int k_pad,
                                     int c_pad, void* stream) {
  TOK_CHECK_ARG(src && dst && k > 0 && r > 0 && s > 0 && c > 0, "tok_pack_weight_dgrad: bad args");
  TOK_CHECK_ARG(k_pad >= k && c_pad >= c, "tok_pack_weight_dgrad: pads smaller than dims");
  hipLaunchKernelGGL(pack_dgrad_kernel, dim3(grid_for((size_t)c_pad * r * s * k_pad)), dim3(256), 0,
                     tok_stream(stream), src, k, r, s, c, (bf16*)dst, k_pad, c_pad);
  TOK_CHECK_LAUNCH("tok_pack_weight_dgrad");
  return TOK_OK;
}

extern "C" int tok_pack_weight_both(const float* src, int k, int r, int s, int c, void* dst_fwd, int k_pad,
                                    int s_pad, int c_pad, void* dst_dgrad, void* stream) {
  TOK_CHECK_ARG(src && dst_fwd && dst_dgrad && k > 0 && r > 0 && s > 0 && c > 0, "tok_pack_weight_both: bad args");
  TOK_CHECK_ARG(k_pad >= k && s_pad >= s && c_pad >= c, "tok_pack_weight_both: pads smaller than dims");
  const size_t tf = (size_t)k_pad * r * s_pad * c_pad, td = (size_t)c_pad * r * s * k_pad;
  hipLaunchKernelGGL(pack_both_kernel, dim3(grid_for(tf > td ? tf : td)), dim3(256), 0, tok_stream(stream), src, k,
                     r, s, c, (bf16*)dst_fwd, k_pad, s_pad, c_pad, (bf16*)dst_dgrad);
  TOK_CHECK_LAUNCH("tok_pack_weight_both");
  return TOK_OK;
}

extern "C" int tok_pack_item_blocks(const tok_pack_item* item) {
  if (!item) return 0;
  const long long tiles = (long long)((item->k_pad + 31) / 32) * ((item->c_pad + 31) / 32) * item->r * item->s_pad;
  if (tiles <= 0 || tiles >= (1ll << 30)) return TOK_ERR_INVALID;
  return (int)tiles;
}

extern "C" int tok_pack_weights_batched(const tok_pack_item* items_dev, int n_items, int total_blocks, void* stream) {
  TOK_CHECK_ARG(items_dev && n_items > 0 && total_blocks > 0, "tok_pack_weights_batched: bad args");
  hipLaunchKernelGGL(pack_batched_kernel, dim3(total_blocks), dim3(256), 0, tok_stream(stream), items_dev, n_items);
  TOK_CHECK_LAUNCH("tok_pack_weights_batched");
  return TOK_OK;
}
