// Multi-resolution glue of the HRNet rows, NHWC bf16, 16 B per lane:
//   fuse_sum_relu fwd/bwd   [timm] HighResolutionModule.forward: y_i = relu(sum_j fuse_ij(x_j)) with the
//                           nearest-neighbour nn.Upsample(2^(j-i)) of the low-resolution terms folded in
//   bilinear fwd/bwd        F.interpolate(mode='bilinear', align_corners=False): HRNetSegmentationNeck
//                           (necks/segmentation/hrnet.py:36-39, written straight into its slice of the
//                           concat buffer, :41) and SegmentationHead (heads/segmentation/base.py:37)
// All four are HBM streaming kernels; the backward passes are gathers (deterministic, no atomics).
#include "tok_common.h"
#include <stdlib.h>

namespace {

struct FuseArgs {
  const bf16* t[4];
  const float* sc[4];      // per-term BatchNorm scale / shift (NULL: the term is taken as stored).  A term whose unit has no
  const float* sf[4];      // activation is passed as its RAW convolution output: the apply pass (a write + a read of the term) is this fma
  int sh[4];
  int nt, n, h, w, c;
};

__global__ __launch_bounds__(256) void fuse_sum_relu_fwd_kernel(FuseArgs a, int relu, bf16* __restrict__ out,
                                                                uint8_t* __restrict__ mask) {
  const int cgs = a.c >> 3;
  const size_t total = (size_t)a.n * a.h * a.w * cgs;
  for (size_t idx = (size_t)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (size_t)gridDim.x * 256) {
    const int cg = (int)(idx % cgs);
    const size_t pix = idx / cgs;
    const int x = (int)(pix % a.w);
    const size_t t2 = pix / a.w;
    const int y = (int)(t2 % a.h);
    const int b = (int)(t2 / a.h);
    float acc[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) acc[e] = 0.f;
    for (int j = 0; j < a.nt; ++j) {
      const int s = a.sh[j];
      const int hs = a.h >> s, ws = a.w >> s;
      const size_t off = (((size_t)b * hs + (y >> s)) * ws + (x >> s)) * a.c + cg * 8;
      const bf16x8 v = ldg16(a.t[j] + off);
      if (a.sc[j] != nullptr) {
        const float4 s0 = *reinterpret_cast<const float4*>(a.sc[j] + cg * 8), s1 = *reinterpret_cast<const float4*>(a.sc[j] + cg * 8 + 4);
        const float4 f0 = *reinterpret_cast<const float4*>(a.sf[j] + cg * 8), f1 = *reinterpret_cast<const float4*>(a.sf[j] + cg * 8 + 4);
        const float sc[8] = {s0.x, s0.y, s0.z, s0.w, s1.x, s1.y, s1.z, s1.w}, sf[8] = {f0.x, f0.y, f0.z, f0.w, f1.x, f1.y, f1.z, f1.w};
#pragma unroll
        for (int e = 0; e < 8; ++e) acc[e] += fmaf(bf2f(v[e]), sc[e], sf[e]);
      } else {
#pragma unroll
        for (int e = 0; e < 8; ++e) acc[e] += bf2f(v[e]);
      }
    }
    bf16x8 o;
    unsigned bits = 0;
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      const float z = relu ? fmaxf(acc[e], 0.f) : acc[e];
      o[e] = f2bf(z);
      bits |= (bf2f(o[e]) > 0.f ? 1u : 0u) << e;
    }
    stg16(out + pix * a.c + cg * 8, o);
    if (mask != nullptr) mask[pix * cgs + cg] = (uint8_t)bits;
  }
}

// d term[b][ys][xs][:] (+)= sum over the 2^s x 2^s block of (mask ? dout : 0)
__global__ __launch_bounds__(256) void fuse_sum_relu_bwd_kernel(const bf16* __restrict__ dout,
                                                                const uint8_t* __restrict__ mask, int n, int h,
                                                                int w, int c, int s, bf16* dterm, int accumulate) {
  const int cgs = c >> 3;
  const int hs = h >> s, ws = w >> s, f = 1 << s;
  const size_t total = (size_t)n * hs * ws * cgs;
  for (size_t idx = (size_t)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (size_t)gridDim.x * 256) {
    const int cg = (int)(idx % cgs);
    const size_t pix = idx / cgs;
    const int xs = (int)(pix % ws);
    const size_t t2 = pix / ws;
    const int ys = (int)(t2 % hs);
    const int b = (int)(t2 / hs);
    float acc[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) acc[e] = 0.f;
    for (int dy = 0; dy < f; ++dy)
      for (int dx = 0; dx < f; ++dx) {
        const size_t p = ((size_t)b * h + (ys * f + dy)) * w + (xs * f + dx);
        const bf16x8 g = ldg16(dout + p * c + cg * 8);
        const unsigned bits = mask ? mask[p * cgs + cg] : 0xFFu;
#pragma unroll
        for (int e = 0; e < 8; ++e) acc[e] += ((bits >> e) & 1u) ? bf2f(g[e]) : 0.f;
      }
    bf16* d = dterm + pix * c + cg * 8;
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

// ATen area_pixel_compute_source_index(scale, dst, align_corners=false, cubic=false)
__device__ __forceinline__ void src_index(float scale, int dst, int in_size, int& i0, int& i1, float& l1) {
  float s = scale * ((float)dst + 0.5f) - 0.5f;
  s = s < 0.f ? 0.f : s;
  i0 = (int)s;
  i0 = i0 > in_size - 1 ? in_size - 1 : i0;
  i1 = i0 + (i0 < in_size - 1 ? 1 : 0);
  l1 = s - (float)i0;
}

struct BilArgs {
  int n, hs, ws, c, ld_src, hd, wd, ld_dst, ch_off;
  float sh, sw;   // in / out
};

__global__ __launch_bounds__(256) void bilinear_fwd_kernel(const bf16* __restrict__ src, bf16* __restrict__ dst,
                                                           BilArgs a) {
  const int cgs = a.c >> 3;
  const size_t total = (size_t)a.n * a.hd * a.wd * cgs;
  for (size_t idx = (size_t)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (size_t)gridDim.x * 256) {
    const int cg = (int)(idx % cgs);
    const size_t pix = idx / cgs;
    const int x = (int)(pix % a.wd);
    const size_t t2 = pix / a.wd;
    const int y = (int)(t2 % a.hd);
    const int b = (int)(t2 / a.hd);
    int y0, y1, x0, x1;
    float ly, lx;
    src_index(a.sh, y, a.hs, y0, y1, ly);
    src_index(a.sw, x, a.ws, x0, x1, lx);
    const float hy = 1.f - ly, hx = 1.f - lx;
    const bf16* base = src + (size_t)b * a.hs * a.ws * a.ld_src + cg * 8;
    const bf16x8 v00 = ldg16(base + ((size_t)y0 * a.ws + x0) * a.ld_src);
    const bf16x8 v01 = ldg16(base + ((size_t)y0 * a.ws + x1) * a.ld_src);
    const bf16x8 v10 = ldg16(base + ((size_t)y1 * a.ws + x0) * a.ld_src);
    const bf16x8 v11 = ldg16(base + ((size_t)y1 * a.ws + x1) * a.ld_src);
    bf16x8 o;
#pragma unroll
    for (int e = 0; e < 8; ++e)
      o[e] = f2bf(hy * (hx * bf2f(v00[e]) + lx * bf2f(v01[e])) + ly * (hx * bf2f(v10[e]) + lx * bf2f(v11[e])));
    stg16(dst + pix * a.ld_dst + a.ch_off + cg * 8, o);
  }
}

// adjoint as a gather: every source pixel visits the destination pixels whose 2x2 footprint can contain it and
// re-evaluates their forward indices/weights exactly (same src_index), so fwd and bwd are transposes bit for bit
__global__ __launch_bounds__(256) void bilinear_bwd_kernel(const bf16* __restrict__ ddst, bf16* dsrc, BilArgs a,
                                                           int accumulate) {
  const int cgs = a.c >> 3;
  const size_t total = (size_t)a.n * a.hs * a.ws * cgs;
  const float rh = 1.f / a.sh, rw = 1.f / a.sw;
  for (size_t idx = (size_t)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (size_t)gridDim.x * 256) {
    const int cg = (int)(idx % cgs);
    const size_t pix = idx / cgs;
    const int xs = (int)(pix % a.ws);
    const size_t t2 = pix / a.ws;
    const int ys = (int)(t2 % a.hs);
    const int b = (int)(t2 / a.hs);
    // dst rows with src coordinate in (ys - 1, ys + 1), widened by one for rounding; row 0 also owns the clamp
    int yd0 = ys == 0 ? 0 : (int)floorf(((float)ys - 0.5f) * rh - 0.5f) - 1;
    int yd1 = (int)ceilf(((float)ys + 1.5f) * rh - 0.5f) + 1;
    int xd0 = xs == 0 ? 0 : (int)floorf(((float)xs - 0.5f) * rw - 0.5f) - 1;
    int xd1 = (int)ceilf(((float)xs + 1.5f) * rw - 0.5f) + 1;
    yd0 = yd0 < 0 ? 0 : yd0;  xd0 = xd0 < 0 ? 0 : xd0;
    yd1 = yd1 > a.hd - 1 ? a.hd - 1 : yd1;  xd1 = xd1 > a.wd - 1 ? a.wd - 1 : xd1;
    float acc[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) acc[e] = 0.f;
    for (int yd = yd0; yd <= yd1; ++yd) {
      int y0, y1; float ly;
      src_index(a.sh, yd, a.hs, y0, y1, ly);
      const float wy = (y0 == ys ? 1.f - ly : 0.f) + (y1 == ys ? ly : 0.f);
      if (wy == 0.f) continue;
      for (int xd = xd0; xd <= xd1; ++xd) {
        int x0, x1; float lx;
        src_index(a.sw, xd, a.ws, x0, x1, lx);
        const float wx = (x0 == xs ? 1.f - lx : 0.f) + (x1 == xs ? lx : 0.f);
        if (wx == 0.f) continue;
        const bf16x8 g = ldg16(ddst + (((size_t)b * a.hd + yd) * a.wd + xd) * a.ld_dst + a.ch_off + cg * 8);
        const float wgt = wy * wx;
#pragma unroll
        for (int e = 0; e < 8; ++e) acc[e] = fmaf(wgt, bf2f(g[e]), acc[e]);
      }
    }
    bf16* d = dsrc + pix * a.ld_src + cg * 8;
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

// channel counts / slice offsets that are not multiples of 8 (HRNet-W18/W30/W44: 18+36+72+144 = 270 concat
// channels): one element per thread, same index math
__global__ __launch_bounds__(256) void bilinear_fwd_generic_kernel(const bf16* __restrict__ src, bf16* __restrict__ dst,
                                                                   BilArgs a) {
  const size_t total = (size_t)a.n * a.hd * a.wd * a.c;
  for (size_t idx = (size_t)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (size_t)gridDim.x * 256) {
    const int ch = (int)(idx % a.c);
    const size_t pix = idx / a.c;
    const int x = (int)(pix % a.wd);
    const size_t t2 = pix / a.wd;
    const int y = (int)(t2 % a.hd);
    const int b = (int)(t2 / a.hd);
    int y0, y1, x0, x1;
    float ly, lx;
    src_index(a.sh, y, a.hs, y0, y1, ly);
    src_index(a.sw, x, a.ws, x0, x1, lx);
    const float hy = 1.f - ly, hx = 1.f - lx;
    const bf16* base = src + (size_t)b * a.hs * a.ws * a.ld_src + ch;
    const float v00 = bf2f(base[((size_t)y0 * a.ws + x0) * a.ld_src]), v01 = bf2f(base[((size_t)y0 * a.ws + x1) * a.ld_src]);
    const float v10 = bf2f(base[((size_t)y1 * a.ws + x0) * a.ld_src]), v11 = bf2f(base[((size_t)y1 * a.ws + x1) * a.ld_src]);
    dst[pix * a.ld_dst + a.ch_off + ch] = f2bf(hy * (hx * v00 + lx * v01) + ly * (hx * v10 + lx * v11));
  }
}

// The transposes of up to three interpolations of ONE map in one pass (the commuted HRNet neck: d(y_j) = up_j^T d(y) for the
// factor-2 / 4 / 8 sources, all over the same 720-channel d(y)).  The gather kernel above reads every element of d(y) four
// times per scale through L2 (each destination pixel lies in the 2 x 2 footprints of four source pixels): 12 x 1.13 GB at
// 512x1024 B=24, 1.7 ms.  Here a workgroup stages a 16 x 16 destination tile with a 4-pixel halo (24 x 24 pixels x 4
// channel groups = 36 KB of LDS) ONCE and every source pixel whose footprint lies inside it — 8 x 8 at factor 2 (4 x 4
// taps), 4 x 4 at factor 4 (8 x 8 taps), 2 x 2 at factor 8 (16 x 16 taps) — takes its taps from LDS; d(y) crosses L2 -> CU
// 2.25 times instead of 12.  One work unit of 16 taps per thread and scale; the units of one source pixel sit in lanes
// 4 / 8 (/ 16 / 32) apart and are folded with fixed-order shuffles.  Weights through src_index, as forward, once per
// workgroup.  The next chunk's window is in flight (nine 16-byte loads per thread, held in registers) while this one is
// consumed: with the loads, the LDS fill and the taps as three serial phases the kernel ran at 2.4 TB/s of staged bytes.
constexpr int ADJ_T = 16, ADJ_HALO = 4, ADJ_W = ADJ_T + 2 * ADJ_HALO, ADJ_G = 4;
constexpr int ADJ_SMEM = ADJ_W * ADJ_W * ADJ_G * 16;
constexpr int ADJ_LD = ADJ_W * ADJ_W * ADJ_G / 256;     // staged 16-byte items per thread: 9

struct Adj3Args {
  const bf16* dy;
  bf16 *d2, *d4, *d8;    // the factor-2 / 4 / 8 sources (NULL = absent)
  int n, h, w, c;
};

typedef __attribute__((ext_vector_type(2))) float f32x2;

// Weights of one work unit: ROWS footprint rows x 2F columns of source pixel (gsy, gsx).  They depend on the tile and the thread
// only — computed once per workgroup, kept in registers across the channel-group chunks (plain local arrays of the kernel: as
// members of a struct they stayed in scratch memory).
template <int F, int ROWS>
__device__ __forceinline__ void adj_weights(float (&wx)[2 * F], float (&wy)[ROWS], int sy, int sx, int row0, int gsy, int gsx,
                                            int ty, int tx, int hs, int ws) {
  // footprint of source pixel (gsy, gsx): destination rows [F gsy - F/2, F gsy + 3F/2), columns alike
  const float scale = 1.f / (float)F;
#pragma unroll
  for (int j = 0; j < 2 * F; ++j) {
    int x0, x1; float lx;
    src_index(scale, tx * ADJ_T + F * sx - F / 2 + j, ws, x0, x1, lx);
    wx[j] = (x0 == gsx ? 1.f - lx : 0.f) + (x1 == gsx ? lx : 0.f);
  }
#pragma unroll
  for (int i = 0; i < ROWS; ++i) {
    int y0, y1; float ly;
    src_index(scale, ty * ADJ_T + F * sy - F / 2 + row0 + i, hs, y0, y1, ly);
    wy[i] = (y0 == gsy ? 1.f - ly : 0.f) + (y1 == gsy ? ly : 0.f);
  }
}

// acc += sum_i wy[i] * (sum_j wx[j] * window[row i][col j]) — the row sums first (one packed fma per channel pair and tap)
template <int F, int ROWS>
__device__ __forceinline__ void adj_unit(const uint4* __restrict__ win, int cgl, int sy, int sx, int row0,
                                         const float (&wx)[2 * F], const float (&wy)[ROWS], f32x2 (&acc)[4]) {
#pragma unroll
  for (int i = 0; i < ROWS; ++i) {
    const int wr = F * sy + ADJ_HALO - F / 2 + row0 + i;
    const uint4* rowp = win + ((size_t)wr * ADJ_W + (F * sx + ADJ_HALO - F / 2)) * ADJ_G + cgl;
    f32x2 r[4] = {{0.f, 0.f}, {0.f, 0.f}, {0.f, 0.f}, {0.f, 0.f}};
#pragma unroll
    for (int j = 0; j < 2 * F; ++j) {
      const uint4 v = rowp[j * ADJ_G];
      const f32x2 wj = {wx[j], wx[j]};
      r[0] += wj * (f32x2){__uint_as_float(v.x << 16), __uint_as_float(v.x & 0xffff0000u)};
      r[1] += wj * (f32x2){__uint_as_float(v.y << 16), __uint_as_float(v.y & 0xffff0000u)};
      r[2] += wj * (f32x2){__uint_as_float(v.z << 16), __uint_as_float(v.z & 0xffff0000u)};
      r[3] += wj * (f32x2){__uint_as_float(v.w << 16), __uint_as_float(v.w & 0xffff0000u)};
    }
    const f32x2 wi = {wy[i], wy[i]};
#pragma unroll
    for (int q = 0; q < 4; ++q) acc[q] += wi * r[q];
  }
}

__device__ __forceinline__ void adj_store(bf16* dst, size_t pix, int c, int cg, const f32x2 (&acc)[4]) {
  bf16x8 o;
#pragma unroll
  for (int q = 0; q < 4; ++q) { o[2 * q] = f2bf(acc[q].x); o[2 * q + 1] = f2bf(acc[q].y); }
  stg16(dst + pix * c + cg * 8, o);
}

template <int MASKS>
__device__ __forceinline__ void adj_fold(f32x2 (&acc)[4]) {
#pragma unroll
  for (int q = 0; q < 4; ++q) {
#pragma unroll
    for (int m = 0; m < MASKS; ++m) {
      acc[q].x += __shfl_xor(acc[q].x, ADJ_G << m);
      acc[q].y += __shfl_xor(acc[q].y, ADJ_G << m);
    }
  }
}

__global__ __launch_bounds__(256) void bilinear_adj3_kernel(Adj3Args a) {
  extern __shared__ __attribute__((aligned(16))) char adj_smem[];
  uint4* win = reinterpret_cast<uint4*>(adj_smem);
  const int tid = threadIdx.x;
  const int tiles_x = a.w / ADJ_T, tiles_y = a.h / ADJ_T;
  const int tx = blockIdx.x % tiles_x;
  const int ty = (blockIdx.x / tiles_x) % tiles_y;
  const int b = blockIdx.x / (tiles_x * tiles_y);
  const int cg_total = a.c >> 3;
  const int cgl = tid % ADJ_G;
  // factor 2: source pixel tid / 4 of the tile's 8 x 8, 4 x 4 taps
  // factor 4: source pixel tid / 16 of 4 x 4, part (tid / 4) % 4 = two of the eight footprint rows
  // factor 8: source pixel tid / 64 of 2 x 2, part (tid / 4) % 16 = one of the sixteen rows
  const int s2 = tid / ADJ_G, s4 = tid / (ADJ_G * 4), s8 = tid / (ADJ_G * 16);
  const int p4 = (tid / ADJ_G) % 4, p8 = (tid / ADJ_G) % 16;
  float w2x[4], w2y[4], w4x[8], w4y[2], w8x[16], w8y[1];
  adj_weights<2, 4>(w2x, w2y, s2 / 8, s2 % 8, 0, ty * 8 + s2 / 8, tx * 8 + s2 % 8, ty, tx, a.h / 2, a.w / 2);
  adj_weights<4, 2>(w4x, w4y, s4 / 4, s4 % 4, 2 * p4, ty * 4 + s4 / 4, tx * 4 + s4 % 4, ty, tx, a.h / 4, a.w / 4);
  adj_weights<8, 1>(w8x, w8y, s8 / 2, s8 % 2, p8, ty * 2 + s8 / 2, tx * 2 + s8 % 2, ty, tx, a.h / 8, a.w / 8);
  // this thread's nine window items: pixel (tid + 256 i) / 4 of the 24 x 24 window, channel group cgl
  const bf16* src[ADJ_LD];
#pragma unroll
  for (int i = 0; i < ADJ_LD; ++i) {
    const int px = (tid + 256 * i) / ADJ_G;
    const int yd = ty * ADJ_T - ADJ_HALO + px / ADJ_W, xd = tx * ADJ_T - ADJ_HALO + px % ADJ_W;
    src[i] = (yd >= 0 && yd < a.h && xd >= 0 && xd < a.w) ? a.dy + (((size_t)b * a.h + yd) * a.w + xd) * a.c + cgl * 8 : nullptr;
  }
  uint4 nxt[ADJ_LD];
  auto fetch = [&](int cg0) {
#pragma unroll
    for (int i = 0; i < ADJ_LD; ++i) {
      nxt[i] = make_uint4(0u, 0u, 0u, 0u);
      if (src[i] != nullptr && cg0 + cgl < cg_total) nxt[i] = *reinterpret_cast<const uint4*>(src[i] + cg0 * 8);
    }
  };
  fetch(0);
  for (int cg0 = 0; cg0 < cg_total; cg0 += ADJ_G) {
    __syncthreads();
#pragma unroll
    for (int i = 0; i < ADJ_LD; ++i) win[tid + 256 * i] = nxt[i];
    __syncthreads();
    if (cg0 + ADJ_G < cg_total) fetch(cg0 + ADJ_G);
    const bool live = cg0 + cgl < cg_total;
    if (a.d2 != nullptr) {
      const int hs = a.h / 2, ws = a.w / 2;
      f32x2 acc[4] = {{0.f, 0.f}, {0.f, 0.f}, {0.f, 0.f}, {0.f, 0.f}};
      adj_unit<2, 4>(win, cgl, s2 / 8, s2 % 8, 0, w2x, w2y, acc);
      if (live) adj_store(a.d2, ((size_t)b * hs + ty * 8 + s2 / 8) * ws + tx * 8 + s2 % 8, a.c, cg0 + cgl, acc);
    }
    if (a.d4 != nullptr) {
      const int hs = a.h / 4, ws = a.w / 4;
      f32x2 acc[4] = {{0.f, 0.f}, {0.f, 0.f}, {0.f, 0.f}, {0.f, 0.f}};
      adj_unit<4, 2>(win, cgl, s4 / 4, s4 % 4, 2 * p4, w4x, w4y, acc);
      adj_fold<2>(acc);
      if (live && p4 == 0) adj_store(a.d4, ((size_t)b * hs + ty * 4 + s4 / 4) * ws + tx * 4 + s4 % 4, a.c, cg0 + cgl, acc);
    }
    if (a.d8 != nullptr) {
      const int hs = a.h / 8, ws = a.w / 8;
      f32x2 acc[4] = {{0.f, 0.f}, {0.f, 0.f}, {0.f, 0.f}, {0.f, 0.f}};
      adj_unit<8, 1>(win, cgl, s8 / 2, s8 % 2, p8, w8x, w8y, acc);
      adj_fold<4>(acc);
      if (live && p8 == 0) adj_store(a.d8, ((size_t)b * hs + ty * 2 + s8 / 2) * ws + tx * 2 + s8 % 2, a.c, cg0 + cgl, acc);
    }
  }
}

// y = y0 + sum_j bilinear(t_j -> h x w), all maps [..][c] bf16 (row pitch c), rounded once; y may be y0 (in place).
// partial[2][gridDim.x][c] = per-channel (sum, sum of squares) of the ROUNDED y: the rows tok_bn_finalize folds.
// The HRNet segmentation neck is conv1x1(cat_j(bilinear(x_j))) — and a 1x1 convolution (a linear map across channels at one
// pixel) commutes with the interpolation (a linear map across pixels of one channel): conv1x1(cat_j up(x_j)) =
// sum_j up(conv1x1_j(x_j)) with conv1x1_j = the filter's columns of source j.  So the 720 -> 720 product runs at each
// source's OWN resolution (48 + 96/4 + 192/16 + 384/64 = 90 input channels' worth of MACs per output pixel instead of 720)
// and this kernel adds the results up: the 720-channel concat tensor (1.1 GB at 512x1024 B=24) and its gradient are never
// built.  Block geometry of the bn.hip streaming kernels (c / 8 channel groups across the block, 256 / that rows).
struct UpSumArgs {
  const bf16* t[3];
  int hs[3], ws[3];
  float sh[3], sw[3];
  int nt, n, h, w, c;
};

__global__ __launch_bounds__(256) void bilinear_sum_stats_kernel(const bf16* y0, bf16* y, UpSumArgs a, int cge, int rpb,
                                                                 float* __restrict__ partial) {
  __shared__ float red[2][256][8];
  const int tid = threadIdx.x;
  const int cgl = tid % cge, rl = tid / cge;
  const int cg_total = a.c >> 3;
  const int M = a.n * a.h * a.w;
  for (int cg = cgl; cg < cg_total; cg += cge) {
    float s1[8], s2[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) { s1[e] = 0.f; s2[e] = 0.f; }
    if (rl < rpb) {
      for (int m = blockIdx.x * rpb + rl; m < M; m += gridDim.x * rpb) {
        const int x = m % a.w;
        const int t2 = m / a.w;
        const int yy = t2 % a.h;
        const int b = t2 / a.h;
        const bf16x8 v0 = ldg16(y0 + (size_t)m * a.c + cg * 8);
        float acc[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) acc[e] = bf2f(v0[e]);
        for (int j = 0; j < a.nt; ++j) {
          int r0, r1, c0, c1;
          float ly, lx;
          src_index(a.sh[j], yy, a.hs[j], r0, r1, ly);
          src_index(a.sw[j], x, a.ws[j], c0, c1, lx);
          const float hy = 1.f - ly, hx = 1.f - lx;
          const bf16* base = a.t[j] + (size_t)b * a.hs[j] * a.ws[j] * a.c + cg * 8;
          const bf16x8 v00 = ldg16(base + ((size_t)r0 * a.ws[j] + c0) * a.c);
          const bf16x8 v01 = ldg16(base + ((size_t)r0 * a.ws[j] + c1) * a.c);
          const bf16x8 v10 = ldg16(base + ((size_t)r1 * a.ws[j] + c0) * a.c);
          const bf16x8 v11 = ldg16(base + ((size_t)r1 * a.ws[j] + c1) * a.c);
#pragma unroll
          for (int e = 0; e < 8; ++e)
            acc[e] += hy * (hx * bf2f(v00[e]) + lx * bf2f(v01[e])) + ly * (hx * bf2f(v10[e]) + lx * bf2f(v11[e]));
        }
        bf16x8 o;
#pragma unroll
        for (int e = 0; e < 8; ++e) {
          o[e] = f2bf(acc[e]);
          const float f = bf2f(o[e]);
          s1[e] += f;
          s2[e] += f * f;
        }
        stg16(y + (size_t)m * a.c + cg * 8, o);
      }
    }
    if (partial != nullptr) {
#pragma unroll
      for (int e = 0; e < 8; ++e) { red[0][tid][e] = s1[e]; red[1][tid][e] = s2[e]; }
      __syncthreads();
      if (rl == 0) {
        for (int r = 1; r < rpb; ++r)
#pragma unroll
          for (int e = 0; e < 8; ++e) { s1[e] += red[0][r * cge + cgl][e]; s2[e] += red[1][r * cge + cgl][e]; }
#pragma unroll
        for (int e = 0; e < 8; ++e) {
          partial[((size_t)0 * gridDim.x + blockIdx.x) * a.c + cg * 8 + e] = s1[e];
          partial[((size_t)1 * gridDim.x + blockIdx.x) * a.c + cg * 8 + e] = s2[e];
        }
      }
      __syncthreads();
    }
  }
}

__global__ __launch_bounds__(256) void bilinear_bwd_generic_kernel(const bf16* __restrict__ ddst, bf16* dsrc, BilArgs a,
                                                                   int accumulate) {
  const size_t total = (size_t)a.n * a.hs * a.ws * a.c;
  const float rh = 1.f / a.sh, rw = 1.f / a.sw;
  for (size_t idx = (size_t)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (size_t)gridDim.x * 256) {
    const int ch = (int)(idx % a.c);
    const size_t pix = idx / a.c;
    const int xs = (int)(pix % a.ws);
    const size_t t2 = pix / a.ws;
    const int ys = (int)(t2 % a.hs);
    const int b = (int)(t2 / a.hs);
    int yd0 = ys == 0 ? 0 : (int)floorf(((float)ys - 0.5f) * rh - 0.5f) - 1;
    int yd1 = (int)ceilf(((float)ys + 1.5f) * rh - 0.5f) + 1;
    int xd0 = xs == 0 ? 0 : (int)floorf(((float)xs - 0.5f) * rw - 0.5f) - 1;
    int xd1 = (int)ceilf(((float)xs + 1.5f) * rw - 0.5f) + 1;
    yd0 = yd0 < 0 ? 0 : yd0;  xd0 = xd0 < 0 ? 0 : xd0;
    yd1 = yd1 > a.hd - 1 ? a.hd - 1 : yd1;  xd1 = xd1 > a.wd - 1 ? a.wd - 1 : xd1;
    float acc = 0.f;
    for (int yd = yd0; yd <= yd1; ++yd) {
      int y0, y1; float ly;
      src_index(a.sh, yd, a.hs, y0, y1, ly);
      const float wy = (y0 == ys ? 1.f - ly : 0.f) + (y1 == ys ? ly : 0.f);
      if (wy == 0.f) continue;
      for (int xd = xd0; xd <= xd1; ++xd) {
        int x0, x1; float lx;
        src_index(a.sw, xd, a.ws, x0, x1, lx);
        const float wx = (x0 == xs ? 1.f - lx : 0.f) + (x1 == xs ? lx : 0.f);
        if (wx == 0.f) continue;
        acc = fmaf(wy * wx, bf2f(ddst[(((size_t)b * a.hd + yd) * a.wd + xd) * a.ld_dst + a.ch_off + ch]), acc);
      }
    }
    bf16* d = dsrc + pix * a.ld_src + ch;
    *d = f2bf(accumulate ? acc + bf2f(*d) : acc);
  }
}

inline int blocks_for(size_t total) {
  const size_t b = (total + 255) / 256;
  return (int)(b > 65536 ? 65536 : (b < 1 ? 1 : b));
}

bool fill_bil(BilArgs& a, int n, int hs, int ws, int c, int ld_src, int hd, int wd, int ld_dst, int ch_off) {
  if (n <= 0 || hs <= 0 || ws <= 0 || hd <= 0 || wd <= 0 || c <= 0 || ch_off < 0 || ld_src < c || ld_dst < ch_off + c)
    return false;
  a.n = n; a.hs = hs; a.ws = ws; a.c = c; a.ld_src = ld_src; a.hd = hd; a.wd = wd; a.ld_dst = ld_dst; a.ch_off = ch_off;
  a.sh = (float)hs / (float)hd;
  a.sw = (float)ws / (float)wd;
  return true;
}

inline bool vec_ok(const BilArgs& a) { return !((a.c | a.ld_src | a.ld_dst | a.ch_off) & 7); }

// The tiled form of the kernel above for the commuted neck's geometry (factors 2 / 4 / 8, map a multiple of 16): the generic
// kernel re-derives indices and weights and unpacks 12 taps for every (pixel, channel group) — ~330 VALU instructions per 16
// output bytes, 1.5 ms at 512x1024 B=24 where HBM needs 0.45.  Here a workgroup owns a 16 x 16 output tile; per chunk of 4
// channel groups the source windows (10 x 10, 6 x 6, 4 x 4 pixels) sit in LDS and a thread owns one column x FOUR CONSECUTIVE
// rows x one channel group: those rows read 4 / 3 / 2 source rows, so the horizontal interpolation runs 9 times per thread
// instead of 24, rows and weights are per-thread constants (computed once per workgroup through src_index, as everywhere),
// and the next chunk's y0 and windows are in flight while this one is summed.  Statistics: per chunk, lanes of one channel
// group are folded with shuffles inside the wave, the four waves through LDS in wave order; one partial row per workgroup.
constexpr int UPS_T = 16, UPS_G = 4;
constexpr int UPS_N2 = UPS_T / 2 + 2, UPS_N4 = UPS_T / 4 + 2, UPS_N8 = UPS_T / 8 + 2;      // window edge: 10, 6, 4 pixels
constexpr int UPS_O2 = 0, UPS_O4 = UPS_N2 * UPS_N2, UPS_O8 = UPS_O4 + UPS_N4 * UPS_N4, UPS_PX = UPS_O8 + UPS_N8 * UPS_N8;   // 152
constexpr int UPS_WLD = (UPS_PX * UPS_G + 255) / 256;     // window items per thread: 3

struct UpsTArgs {
  const bf16* y0;
  bf16* y;
  const bf16 *t2, *t4, *t8;     // the factor-2 / 4 / 8 sources (NULL = absent)
  float* partial;               // [2][gridDim.x][c] or NULL
  int n, h, w, c;
};

__device__ __forceinline__ void ups_unpack(const uint4 v, f32x2 (&o)[4]) {
  o[0] = (f32x2){__uint_as_float(v.x << 16), __uint_as_float(v.x & 0xffff0000u)};
  o[1] = (f32x2){__uint_as_float(v.y << 16), __uint_as_float(v.y & 0xffff0000u)};
  o[2] = (f32x2){__uint_as_float(v.z << 16), __uint_as_float(v.z & 0xffff0000u)};
  o[3] = (f32x2){__uint_as_float(v.w << 16), __uint_as_float(v.w & 0xffff0000u)};
}

// hoisted constants of one factor: window columns + weights of the thread's column, the window row of its first source row,
// and for each of its four output rows the weights of the two source rows the row pattern assigns to it
template <int F>
struct UpsRowPat;
template <> struct UpsRowPat<2> { static constexpr int NR = 4; static constexpr int A[4] = {0, 1, 1, 2}; };
template <> struct UpsRowPat<4> { static constexpr int NR = 3; static constexpr int A[4] = {0, 0, 1, 1}; };
template <> struct UpsRowPat<8> { static constexpr int NR = 2; static constexpr int A[4] = {0, 0, 0, 0}; };

template <int F>
__device__ __forceinline__ void ups_consts(int ty, int tx, int rb, int col, int hs, int ws, int& cA, int& cB, float& hx, float& lx,
                                           int& wrow, float (&wa)[4], float (&wb)[4]) {
  const float scale = 1.f / (float)F;
  int x0, x1;
  src_index(scale, tx * UPS_T + col, ws, x0, x1, lx);
  hx = 1.f - lx;
  const int oc = tx * (UPS_T / F) - 1;                       // source column of window column 0
  cA = x0 - oc;  cB = x1 - oc;
  const int d0 = ty * UPS_T + 4 * rb;
  // first source row of the thread's four output rows: rows d0 .. d0 + 3 read source rows base .. base + NR - 1
  const int base = F == 8 ? d0 / 8 - 1 + ((d0 % 8) >= 4 ? 1 : 0) : d0 / F - 1;
  wrow = base - (ty * (UPS_T / F) - 1);
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    int y0, y1; float ly;
    src_index(scale, d0 + k, hs, y0, y1, ly);
    const int ra = base + UpsRowPat<F>::A[k], rbb = ra + 1;
    wa[k] = (ra >= 0 && ra < hs) ? (y0 == ra ? 1.f - ly : 0.f) + (y1 == ra ? ly : 0.f) : 0.f;
    wb[k] = (rbb >= 0 && rbb < hs) ? (y0 == rbb ? 1.f - ly : 0.f) + (y1 == rbb ? ly : 0.f) : 0.f;
  }
}

template <int F, int OFF, int N>
__device__ __forceinline__ void ups_add(const uint4* __restrict__ lw, int cgl, int cA, int cB, float hx, float lx, int wrow,
                                        const float (&wa)[4], const float (&wb)[4], f32x2 (&acc)[4][4]) {
  constexpr int NR = UpsRowPat<F>::NR;
  f32x2 X[NR][4];
  const f32x2 hx2 = {hx, hx}, lx2 = {lx, lx};
#pragma unroll
  for (int r = 0; r < NR; ++r) {
    const uint4* rowp = lw + (size_t)(OFF + (wrow + r) * N) * UPS_G + cgl;
    f32x2 a[4], b[4];
    ups_unpack(rowp[cA * UPS_G], a);
    ups_unpack(rowp[cB * UPS_G], b);
#pragma unroll
    for (int q = 0; q < 4; ++q) X[r][q] = hx2 * a[q] + lx2 * b[q];
  }
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    const f32x2 wa2 = {wa[k], wa[k]}, wb2 = {wb[k], wb[k]};
#pragma unroll
    for (int q = 0; q < 4; ++q) acc[k][q] += wa2 * X[UpsRowPat<F>::A[k]][q] + wb2 * X[UpsRowPat<F>::A[k] + 1][q];
  }
}

__global__ __launch_bounds__(256) void bilinear_sum_tiled_kernel(UpsTArgs a) {
  __shared__ uint4 lw[UPS_PX * UPS_G];
  __shared__ float sred[4][UPS_G][16];
  const int tid = threadIdx.x;
  const int tiles_x = a.w / UPS_T, tiles_y = a.h / UPS_T;
  const int tx = blockIdx.x % tiles_x;
  const int ty = (blockIdx.x / tiles_x) % tiles_y;
  const int b = blockIdx.x / (tiles_x * tiles_y);
  const int cg_total = a.c >> 3;
  const int cgl = tid % UPS_G, col = (tid / UPS_G) % UPS_T, rb = tid / (UPS_G * UPS_T);     // rb = wave index
  int cA2, cB2, cA4, cB4, cA8, cB8, wr2, wr4, wr8;
  float hx2, lx2, hx4, lx4, hx8, lx8, wa2[4], wb2[4], wa4[4], wb4[4], wa8[4], wb8[4];
  ups_consts<2>(ty, tx, rb, col, a.h / 2, a.w / 2, cA2, cB2, hx2, lx2, wr2, wa2, wb2);
  ups_consts<4>(ty, tx, rb, col, a.h / 4, a.w / 4, cA4, cB4, hx4, lx4, wr4, wa4, wb4);
  ups_consts<8>(ty, tx, rb, col, a.h / 8, a.w / 8, cA8, cB8, hx8, lx8, wr8, wa8, wb8);
  // the thread's window items (pixel (tid + 256 i) / 4 of the three windows) and its four y0 rows
  const bf16* wsrc[UPS_WLD];
#pragma unroll
  for (int i = 0; i < UPS_WLD; ++i) {
    const int px = (tid + 256 * i) / UPS_G;
    wsrc[i] = nullptr;
    if (px < UPS_PX) {
      const int f = px < UPS_O4 ? 2 : (px < UPS_O8 ? 4 : 8);
      const int nn = f == 2 ? UPS_N2 : (f == 4 ? UPS_N4 : UPS_N8);
      const int p = px - (f == 2 ? UPS_O2 : (f == 4 ? UPS_O4 : UPS_O8));
      const int hs = a.h / f, ws = a.w / f;
      const int r = ty * (UPS_T / f) - 1 + p / nn, cc = tx * (UPS_T / f) - 1 + p % nn;
      const bf16* t = f == 2 ? a.t2 : (f == 4 ? a.t4 : a.t8);
      if (t != nullptr && r >= 0 && r < hs && cc >= 0 && cc < ws) wsrc[i] = t + (((size_t)b * hs + r) * ws + cc) * a.c + cgl * 8;
    }
  }
  const size_t pix0 = ((size_t)b * a.h + ty * UPS_T + 4 * rb) * a.w + tx * UPS_T + col;
  uint4 nw[UPS_WLD], ny[4];
  auto fetch = [&](int cg0) {
    const bool ok = cg0 + cgl < cg_total;
#pragma unroll
    for (int i = 0; i < UPS_WLD; ++i) {
      nw[i] = make_uint4(0u, 0u, 0u, 0u);
      if (ok && wsrc[i] != nullptr) nw[i] = *reinterpret_cast<const uint4*>(wsrc[i] + cg0 * 8);
    }
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      ny[k] = make_uint4(0u, 0u, 0u, 0u);
      if (ok) ny[k] = *reinterpret_cast<const uint4*>(a.y0 + (pix0 + (size_t)k * a.w) * a.c + (cg0 + cgl) * 8);
    }
  };
  fetch(0);
  for (int cg0 = 0; cg0 < cg_total; cg0 += UPS_G) {
    __syncthreads();
#pragma unroll
    for (int i = 0; i < UPS_WLD; ++i)
      if (tid + 256 * i < UPS_PX * UPS_G) lw[tid + 256 * i] = nw[i];
    f32x2 acc[4][4];
#pragma unroll
    for (int k = 0; k < 4; ++k) ups_unpack(ny[k], acc[k]);
    __syncthreads();
    if (cg0 + UPS_G < cg_total) fetch(cg0 + UPS_G);
    if (a.t2 != nullptr) ups_add<2, UPS_O2, UPS_N2>(lw, cgl, cA2, cB2, hx2, lx2, wr2, wa2, wb2, acc);
    if (a.t4 != nullptr) ups_add<4, UPS_O4, UPS_N4>(lw, cgl, cA4, cB4, hx4, lx4, wr4, wa4, wb4, acc);
    if (a.t8 != nullptr) ups_add<8, UPS_O8, UPS_N8>(lw, cgl, cA8, cB8, hx8, lx8, wr8, wa8, wb8, acc);
    const bool live = cg0 + cgl < cg_total;
    f32x2 s1[4] = {{0.f, 0.f}, {0.f, 0.f}, {0.f, 0.f}, {0.f, 0.f}}, s2[4] = {{0.f, 0.f}, {0.f, 0.f}, {0.f, 0.f}, {0.f, 0.f}};
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      bf16x8 o;
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        o[2 * q] = f2bf(acc[k][q].x);
        o[2 * q + 1] = f2bf(acc[k][q].y);
        const f32x2 f = {bf2f(o[2 * q]), bf2f(o[2 * q + 1])};
        s1[q] += f;
        s2[q] += f * f;
      }
      if (live) stg16(a.y + (pix0 + (size_t)k * a.w) * a.c + (cg0 + cgl) * 8, o);
    }
    if (a.partial != nullptr) {
      // lanes cgl, cgl + 4, ... of the wave hold the same channel group (16 columns): fold, then the four waves in order
#pragma unroll
      for (int q = 0; q < 4; ++q) {
#pragma unroll
        for (int m = UPS_G; m < 64; m <<= 1) {
          s1[q].x += __shfl_xor(s1[q].x, m);  s1[q].y += __shfl_xor(s1[q].y, m);
          s2[q].x += __shfl_xor(s2[q].x, m);  s2[q].y += __shfl_xor(s2[q].y, m);
        }
      }
      if ((tid & 63) < UPS_G) {
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          sred[rb][cgl][2 * q] = s1[q].x;      sred[rb][cgl][2 * q + 1] = s1[q].y;
          sred[rb][cgl][8 + 2 * q] = s2[q].x;  sred[rb][cgl][8 + 2 * q + 1] = s2[q].y;
        }
      }
      __syncthreads();
      if (tid < UPS_G * 16) {
        const int g = tid / 16, e = tid % 16;
        if (cg0 + g < cg_total) {
          const float v = ((sred[0][g][e] + sred[1][g][e]) + sred[2][g][e]) + sred[3][g][e];
          a.partial[((size_t)(e >> 3) * gridDim.x + blockIdx.x) * a.c + (cg0 + g) * 8 + (e & 7)] = v;
        }
      }
    }
  }
}

}  // namespace

extern "C" int tok_fuse_sum_affine_relu_fwd(const void* t0, int s0, const float* sc0, const float* sf0, const void* t1, int s1,
                                            const float* sc1, const float* sf1, const void* t2, int s2, const float* sc2,
                                            const float* sf2, const void* t3, int s3, const float* sc3, const float* sf3, int n,
                                            int h, int w, int c, int relu, void* out, uint8_t* mask, void* stream) {
  FuseArgs a;
  const void* ts[4] = {t0, t1, t2, t3};
  const int ss[4] = {s0, s1, s2, s3};
  const float* scs[4] = {sc0, sc1, sc2, sc3};
  const float* sfs[4] = {sf0, sf1, sf2, sf3};
  a.nt = 0;
  for (int j = 0; j < 4; ++j) {
    if (ts[j] == nullptr) continue;
    TOK_CHECK_ARG(ss[j] >= 0 && ss[j] < 8 && (h >> ss[j]) << ss[j] == h && (w >> ss[j]) << ss[j] == w,
                  "tok_fuse_sum_relu_fwd: term %d: %dx%d is not a multiple of 2^%d", j, h, w, ss[j]);
    TOK_CHECK_ARG((scs[j] == nullptr) == (sfs[j] == nullptr), "tok_fuse_sum_relu_fwd: term %d: scale without shift", j);
    a.t[a.nt] = (const bf16*)ts[j];
    a.sh[a.nt] = ss[j];
    a.sc[a.nt] = scs[j];
    a.sf[a.nt] = sfs[j];
    ++a.nt;
  }
  for (int j = a.nt; j < 4; ++j) { a.t[j] = nullptr; a.sh[j] = 0; a.sc[j] = a.sf[j] = nullptr; }
  TOK_CHECK_ARG(a.nt > 0 && out && n > 0 && h > 0 && w > 0 && c > 0 && c % 8 == 0, "tok_fuse_sum_relu_fwd: bad args");
  a.n = n; a.h = h; a.w = w; a.c = c;
  hipLaunchKernelGGL(fuse_sum_relu_fwd_kernel, dim3(blocks_for((size_t)n * h * w * (c >> 3))), dim3(256), 0,
                     tok_stream(stream), a, relu, (bf16*)out, mask);
  TOK_CHECK_LAUNCH("tok_fuse_sum_relu_fwd");
  return TOK_OK;
}

extern "C" int tok_fuse_sum_relu_fwd(const void* t0, int s0, const void* t1, int s1, const void* t2, int s2,
                                     const void* t3, int s3, int n, int h, int w, int c, int relu, void* out,
                                     uint8_t* mask, void* stream) {
  return tok_fuse_sum_affine_relu_fwd(t0, s0, nullptr, nullptr, t1, s1, nullptr, nullptr, t2, s2, nullptr, nullptr, t3, s3, nullptr,
                                      nullptr, n, h, w, c, relu, out, mask, stream);
}

extern "C" int tok_fuse_sum_relu_bwd(const void* dout, const uint8_t* mask, int n, int h, int w, int c, int shift,
                                     void* dterm, int accumulate, void* stream) {
  TOK_CHECK_ARG(dout && dterm && n > 0 && h > 0 && w > 0 && c > 0 && c % 8 == 0 && shift >= 0 && shift < 8 &&
                (h >> shift) << shift == h && (w >> shift) << shift == w, "tok_fuse_sum_relu_bwd: bad args");
  hipLaunchKernelGGL(fuse_sum_relu_bwd_kernel, dim3(blocks_for((size_t)n * (h >> shift) * (w >> shift) * (c >> 3))),
                     dim3(256), 0, tok_stream(stream), (const bf16*)dout, mask, n, h, w, c, shift, (bf16*)dterm,
                     accumulate);
  TOK_CHECK_LAUNCH("tok_fuse_sum_relu_bwd");
  return TOK_OK;
}

extern "C" int tok_bilinear_fwd(const void* src, int n, int hs, int ws, int c, int ld_src, void* dst, int hd, int wd,
                                int ld_dst, int ch_off, void* stream) {
  BilArgs a;
  TOK_CHECK_ARG(src && dst && fill_bil(a, n, hs, ws, c, ld_src, hd, wd, ld_dst, ch_off), "tok_bilinear_fwd: bad args");
  if (vec_ok(a))
    hipLaunchKernelGGL(bilinear_fwd_kernel, dim3(blocks_for((size_t)n * hd * wd * (c >> 3))), dim3(256), 0,
                       tok_stream(stream), (const bf16*)src, (bf16*)dst, a);
  else
    hipLaunchKernelGGL(bilinear_fwd_generic_kernel, dim3(blocks_for((size_t)n * hd * wd * c)), dim3(256), 0,
                       tok_stream(stream), (const bf16*)src, (bf16*)dst, a);
  TOK_CHECK_LAUNCH("tok_bilinear_fwd");
  return TOK_OK;
}

extern "C" int tok_bilinear_bwd(const void* ddst, int n, int hd, int wd, int ld_dst, int ch_off, void* dsrc, int hs,
                                int ws, int c, int ld_src, int accumulate, void* stream) {
  BilArgs a;
  TOK_CHECK_ARG(ddst && dsrc && fill_bil(a, n, hs, ws, c, ld_src, hd, wd, ld_dst, ch_off), "tok_bilinear_bwd: bad args");
  if (vec_ok(a))
    hipLaunchKernelGGL(bilinear_bwd_kernel, dim3(blocks_for((size_t)n * hs * ws * (c >> 3))), dim3(256), 0,
                       tok_stream(stream), (const bf16*)ddst, (bf16*)dsrc, a, accumulate);
  else
    hipLaunchKernelGGL(bilinear_bwd_generic_kernel, dim3(blocks_for((size_t)n * hs * ws * c)), dim3(256), 0,
                       tok_stream(stream), (const bf16*)ddst, (bf16*)dsrc, a, accumulate);
  TOK_CHECK_LAUNCH("tok_bilinear_bwd");
  return TOK_OK;
}

static bool ups_tiled_shape(int n, int h, int w, int c) {
  return c % 8 == 0 && h % UPS_T == 0 && w % UPS_T == 0 && (long long)n * (h / UPS_T) * (w / UPS_T) <= 65536;
}

extern "C" int tok_bilinear_sum_stats_rows(int n, int h, int w, int c) {
  if (n <= 0 || h <= 0 || w <= 0 || c <= 0 || c % 8) return TOK_ERR_INVALID;
  if (ups_tiled_shape(n, h, w, c)) return n * (h / UPS_T) * (w / UPS_T);      // one row per 16 x 16 tile
  return tok_bn_stats_rows((int64_t)n * h * w, c);
}

extern "C" int tok_bilinear_sum_stats(const void* y0, const void* t1, int h1, int w1, const void* t2, int h2, int w2,
                                      const void* t3, int h3, int w3, int n, int h, int w, int c, void* y, float* stats,
                                      void* stream) {
  UpSumArgs a;
  const void* ts[3] = {t1, t2, t3};
  const int hs[3] = {h1, h2, h3}, ws[3] = {w1, w2, w3};
  TOK_CHECK_ARG(y0 && y && n > 0 && h > 0 && w > 0 && c > 0 && c % 8 == 0 && (int64_t)n * h * w < (int64_t)1 << 31,
                "tok_bilinear_sum_stats: bad args");
  a.nt = 0;
  for (int j = 0; j < 3; ++j) {
    a.t[j] = nullptr; a.hs[j] = a.ws[j] = 1; a.sh[j] = a.sw[j] = 1.f;
  }
  UpsTArgs u;
  u.y0 = (const bf16*)y0; u.y = (bf16*)y; u.t2 = u.t4 = u.t8 = nullptr; u.partial = stats;
  u.n = n; u.h = h; u.w = w; u.c = c;
  bool tiled = ups_tiled_shape(n, h, w, c);
  for (int j = 0; j < 3; ++j) {
    if (ts[j] == nullptr) continue;
    TOK_CHECK_ARG(hs[j] > 0 && ws[j] > 0, "tok_bilinear_sum_stats: term %d: bad size", j + 1);
    a.t[a.nt] = (const bf16*)ts[j];
    a.hs[a.nt] = hs[j]; a.ws[a.nt] = ws[j];
    a.sh[a.nt] = (float)hs[j] / (float)h;
    a.sw[a.nt] = (float)ws[j] / (float)w;
    ++a.nt;
    const int f = h / hs[j];
    const bf16** slot = f == 2 ? &u.t2 : (f == 4 ? &u.t4 : (f == 8 ? &u.t8 : nullptr));
    if (hs[j] * f == h && ws[j] * f == w && slot != nullptr && *slot == nullptr) *slot = (const bf16*)ts[j];   // one source per factor
    else tiled = false;
  }
  a.n = n; a.h = h; a.w = w; a.c = c;
  const int rows = tok_bilinear_sum_stats_rows(n, h, w, c);
  static const int off = [] { const char* e = getenv("TOK_BILINEAR_SUM_TILED"); return (int)(e ? atoi(e) == 0 : 0); }();   // A/B switch
  if (tiled && !off) {
    hipLaunchKernelGGL(bilinear_sum_tiled_kernel, dim3(rows), dim3(256), 0, tok_stream(stream), u);
  } else {
    // (any grid works for the generic kernel: blocks beyond the rows write zero partials)
    const int cg_total = c / 8, cge = cg_total < 256 ? cg_total : 256, rpb = 256 / cge;
    hipLaunchKernelGGL(bilinear_sum_stats_kernel, dim3(rows), dim3(256), 0, tok_stream(stream), (const bf16*)y0, (bf16*)y, a,
                       cge, rpb, stats);
  }
  TOK_CHECK_LAUNCH("tok_bilinear_sum_stats");
  return TOK_OK;
}

extern "C" int tok_bilinear_bwd_multi(const void* ddst, int n, int hd, int wd, int c, void* d1, int h1, int w1, void* d2, int h2,
                                      int w2, void* d3, int h3, int w3, void* stream) {
  void* ds[3] = {d1, d2, d3};
  const int hs[3] = {h1, h2, h3}, ws[3] = {w1, w2, w3};
  TOK_CHECK_ARG(ddst && n > 0 && hd > 0 && wd > 0 && c > 0, "tok_bilinear_bwd_multi: bad args");
  Adj3Args a;
  a.dy = (const bf16*)ddst; a.n = n; a.h = hd; a.w = wd; a.c = c;
  bool fast = c % 8 == 0 && hd % ADJ_T == 0 && wd % ADJ_T == 0 && (long long)n * (hd / ADJ_T) * (wd / ADJ_T) < (1ll << 31);
  int present = 0;
  a.d2 = a.d4 = a.d8 = nullptr;
  for (int j = 0; j < 3; ++j) {
    if (ds[j] == nullptr) continue;
    TOK_CHECK_ARG(hs[j] > 0 && ws[j] > 0, "tok_bilinear_bwd_multi: source %d: bad size", j + 1);
    ++present;
    const int f = hd / hs[j];
    bf16** slot = f == 2 ? &a.d2 : (f == 4 ? &a.d4 : (f == 8 ? &a.d8 : nullptr));
    if (hs[j] * f == hd && ws[j] * f == wd && slot != nullptr && *slot == nullptr) *slot = (bf16*)ds[j];   // one source per factor
    else fast = false;
  }
  static const int off = [] { const char* e = getenv("TOK_BILINEAR_ADJ3"); return (int)(e ? atoi(e) == 0 : 0); }();   // A/B switch
  if (present == 0) return TOK_OK;
  if (fast && !off) {
    static const bool attr_set = [] {
      (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&bilinear_adj3_kernel), hipFuncAttributeMaxDynamicSharedMemorySize,
                                ADJ_SMEM);
      return true;
    }();
    (void)attr_set;
    hipLaunchKernelGGL(bilinear_adj3_kernel, dim3(n * (hd / ADJ_T) * (wd / ADJ_T)), dim3(256), ADJ_SMEM, tok_stream(stream), a);
    TOK_CHECK_LAUNCH("tok_bilinear_bwd_multi");
    return TOK_OK;
  }
  for (int j = 0; j < 3; ++j)        // any other geometry: one gather launch per source
    if (ds[j] != nullptr)
      if (int e = tok_bilinear_bwd(ddst, n, hd, wd, c, 0, ds[j], hs[j], ws[j], c, c, 0, stream)) return e;
  return TOK_OK;
}
