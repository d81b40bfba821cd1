// Multi-resolution glue of the HRNet rows, NHWC bf16, 16 B per lane:
//   fuse_sum_relu fwd/bwd   [timm] HighResolutionModule.forward: y_i = relu(sum_j fuse_ij(x_j)) with the
//                           nearest-neighbour nn.Upsample(2^(j-i)) of the low-resolution terms folded in
//   bilinear fwd/bwd        F.interpolate(mode='bilinear', align_corners=False): HRNetSegmentationNeck
//                           (necks/segmentation/hrnet.py:36-39, written straight into its slice of the
//                           concat buffer, :41) and SegmentationHead (heads/segmentation/base.py:37)
// All four are HBM streaming kernels; the backward passes are gathers (deterministic, no atomics).
#include "tok_common.h"

namespace {

struct FuseArgs {
  const bf16* t[4];
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
#pragma unroll
      for (int e = 0; e < 8; ++e) acc[e] += bf2f(v[e]);
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

}  // namespace

extern "C" int tok_fuse_sum_relu_fwd(const void* t0, int s0, const void* t1, int s1, const void* t2, int s2,
                                     const void* t3, int s3, int n, int h, int w, int c, int relu, void* out,
                                     uint8_t* mask, void* stream) {
  FuseArgs a;
  const void* ts[4] = {t0, t1, t2, t3};
  const int ss[4] = {s0, s1, s2, s3};
  a.nt = 0;
  for (int j = 0; j < 4; ++j) {
    if (ts[j] == nullptr) continue;
    TOK_CHECK_ARG(ss[j] >= 0 && ss[j] < 8 && (h >> ss[j]) << ss[j] == h && (w >> ss[j]) << ss[j] == w,
                  "tok_fuse_sum_relu_fwd: term %d: %dx%d is not a multiple of 2^%d", j, h, w, ss[j]);
    a.t[a.nt] = (const bf16*)ts[j];
    a.sh[a.nt] = ss[j];
    ++a.nt;
  }
  for (int j = a.nt; j < 4; ++j) { a.t[j] = nullptr; a.sh[j] = 0; }
  TOK_CHECK_ARG(a.nt > 0 && out && n > 0 && h > 0 && w > 0 && c > 0 && c % 8 == 0, "tok_fuse_sum_relu_fwd: bad args");
  a.n = n; a.h = h; a.w = w; a.c = c;
  hipLaunchKernelGGL(fuse_sum_relu_fwd_kernel, dim3(blocks_for((size_t)n * h * w * (c >> 3))), dim3(256), 0,
                     tok_stream(stream), a, relu, (bf16*)out, mask);
  TOK_CHECK_LAUNCH("tok_fuse_sum_relu_fwd");
  return TOK_OK;
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

extern "C" int tok_bilinear_sum_stats_rows(int64_t m, int c) { return tok_bn_stats_rows(m, c); }

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
  for (int j = 0; j < 3; ++j) {
    if (ts[j] == nullptr) continue;
    TOK_CHECK_ARG(hs[j] > 0 && ws[j] > 0, "tok_bilinear_sum_stats: term %d: bad size", j + 1);
    a.t[a.nt] = (const bf16*)ts[j];
    a.hs[a.nt] = hs[j]; a.ws[a.nt] = ws[j];
    a.sh[a.nt] = (float)hs[j] / (float)h;
    a.sw[a.nt] = (float)ws[j] / (float)w;
    ++a.nt;
  }
  a.n = n; a.h = h; a.w = w; a.c = c;
  const int64_t m = (int64_t)n * h * w;
  const int rows = tok_bn_stats_rows(m, c);
  const int cg_total = c / 8, cge = cg_total < 256 ? cg_total : 256, rpb = 256 / cge;
  hipLaunchKernelGGL(bilinear_sum_stats_kernel, dim3(rows), dim3(256), 0, tok_stream(stream), (const bf16*)y0, (bf16*)y, a,
                     cge, rpb, stats);
  TOK_CHECK_LAUNCH("tok_bilinear_sum_stats");
  return TOK_OK;
}
