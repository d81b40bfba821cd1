// BatchNorm (training, batch statistics) + ReLU + residual add, NHWC bf16, fp32 statistics.
// All passes are HBM-bound streaming kernels: 16-byte (8 x bf16) accesses per lane, a thread
// keeps ONE 8-channel group for its whole life so scale/shift/coefficients live in registers.
//
// Forward :  conv epilogue partial sums -> tok_bn_finalize -> tok_bn_act_fwd
// Backward:  tok_bn_bwd_reduce (partials) -> tok_bn_bwd_finalize -> tok_bn_bwd_apply
//            dz = dout * mask ;  dy = a1*dz + a2*y + a3   (a* per channel)
#include "tok_common.h"
#include <hip/hip_ext.h>
#include <stdlib.h>

static thread_local hipEvent_t t_done_event = nullptr;

namespace {

// Block geometry shared by the streaming kernels: CGE = min(C/8, 256) channel groups across
// the block, RPB = 256 / CGE rows per block iteration.
struct Geo {
  int cg_total, cge, rpb;
};
inline Geo make_geo(int c) {
  Geo g;
  g.cg_total = c / 8;
  g.cge = g.cg_total < 256 ? g.cg_total : 256;
  g.rpb = 256 / g.cge;
  return g;
}
inline int stream_blocks(int64_t m, const Geo& g, int cap) {
  int64_t b = (m + g.rpb - 1) / g.rpb;
  if (b > cap) b = cap;
  if (b < 1) b = 1;
  return (int)b;
}

__device__ __forceinline__ void load8f(const float* p, float (&v)[8]) {
  const float4 a = *reinterpret_cast<const float4*>(p);
  const float4 b = *reinterpret_cast<const float4*>(p + 4);
  v[0] = a.x; v[1] = a.y; v[2] = a.z; v[3] = a.w;
  v[4] = b.x; v[5] = b.y; v[6] = b.z; v[7] = b.w;
}

// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void bn_act_fwd_kernel(const bf16* __restrict__ y,
                                                         const float* __restrict__ scale,
                                                         const float* __restrict__ shift,
                                                         const bf16* __restrict__ shortcut, int relu,
                                                         bf16* __restrict__ out, uint8_t* __restrict__ mask,
                                                         int64_t M, int C, int cge, int rpb,
                                                         float* __restrict__ csum /* may be null: [gridDim.x][C] */) {
  __shared__ float cred[256][8];
  const int tid = threadIdx.x;
  const int cgl = tid % cge, rl = tid / cge;
  if (rl >= rpb && csum == nullptr) return;
  const int cg_total = C >> 3;
  for (int cg = cgl; cg < cg_total; cg += cge) {
    float sc[8], sh[8], cs[8];
    load8f(scale + cg * 8, sc);
    load8f(shift + cg * 8, sh);
#pragma unroll
    for (int e = 0; e < 8; ++e) cs[e] = 0.f;
    if (rl < rpb)
    for (int64_t m = (int64_t)blockIdx.x * rpb + rl; m < M; m += (int64_t)gridDim.x * rpb) {
      const size_t off = (size_t)m * C + cg * 8;
      const bf16x8 v = ldg16(y + off);
      bf16x8 o;
      if (shortcut != nullptr) {
        const bf16x8 s = ldg16(shortcut + off);
#pragma unroll
        for (int e = 0; e < 8; ++e) {
          float z = fmaf(bf2f(v[e]), sc[e], sh[e]) + bf2f(s[e]);
          if (relu) z = fmaxf(z, 0.f);
          o[e] = f2bf(z);
        }
      } else {
#pragma unroll
        for (int e = 0; e < 8; ++e) {
          float z = fmaf(bf2f(v[e]), sc[e], sh[e]);
          if (relu) z = fmaxf(z, 0.f);
          o[e] = f2bf(z);
        }
      }
      stg16(out + off, o);
      if (csum != nullptr) {
#pragma unroll
        for (int e = 0; e < 8; ++e) cs[e] += bf2f(o[e]);
      }
      if (mask != nullptr) {   // bit e = (out[e] > 0): the ReLU mask the backward kernels read (1/16 of `out`)
        unsigned bits = 0;
#pragma unroll
        for (int e = 0; e < 8; ++e) bits |= (bf2f(o[e]) > 0.f ? 1u : 0u) << e;
        mask[(size_t)m * cg_total + cg] = (uint8_t)bits;
      }
    }
    if (csum != nullptr) {
      // column sums of the STORED values over this block's rows: one row of partials per block (the fused residual unit
      // wants colsum(z) of its input; producing it here saves the stand-alone pass over z)
#pragma unroll
      for (int e = 0; e < 8; ++e) cred[tid][e] = cs[e];
      __syncthreads();
      if (rl == 0) {
        for (int r = 1; r < rpb; ++r)
#pragma unroll
          for (int e = 0; e < 8; ++e) cs[e] += cred[r * cge + cgl][e];
#pragma unroll
        for (int e = 0; e < 8; ++e) csum[(size_t)blockIdx.x * C + cg * 8 + e] = cs[e];
      }
      __syncthreads();
    }
  }
}

// generic per-channel (sum, sumsq) partials of an NHWC tensor: partial[2][gridDim.x][C]
__global__ __launch_bounds__(256) void bn_stats_kernel(const bf16* __restrict__ y, int64_t M, int C,
                                                       int cge, int rpb, float* __restrict__ partial) {
  __shared__ float red[2][256][8];
  const int tid = threadIdx.x;
  const int cgl = tid % cge, rl = tid / cge;
  const int cg_total = C >> 3;
  for (int cg = cgl; cg < cg_total; cg += cge) {
    float s1[8], s2[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) { s1[e] = 0.f; s2[e] = 0.f; }
    if (rl < rpb) {
      for (int64_t m = (int64_t)blockIdx.x * rpb + rl; m < M; m += (int64_t)gridDim.x * rpb) {
        const bf16x8 v = ldg16(y + (size_t)m * C + cg * 8);
#pragma unroll
        for (int e = 0; e < 8; ++e) { const float f = bf2f(v[e]); s1[e] += f; s2[e] += f * f; }
      }
    }
#pragma unroll
    for (int e = 0; e < 8; ++e) { red[0][tid][e] = s1[e]; red[1][tid][e] = s2[e]; }
    __syncthreads();
    if (rl == 0) {
      for (int r = 1; r < rpb; ++r)
#pragma unroll
        for (int e = 0; e < 8; ++e) { s1[e] += red[0][r * cge + cgl][e]; s2[e] += red[1][r * cge + cgl][e]; }
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        partial[((size_t)0 * gridDim.x + blockIdx.x) * C + cg * 8 + e] = s1[e];
        partial[((size_t)1 * gridDim.x + blockIdx.x) * C + cg * 8 + e] = s2[e];
      }
    }
    __syncthreads();
  }
}

// Per-lane fold of partial rows rl, rl+64, ... for both sums: 8 loads in flight (the serial version was bound by the
// load latency: 12 dependent round trips on 768 rows), additions in row order so the result does not depend on it.
// RL = row lanes of the block (256 / channels per block).
template <int RL, int U>
__device__ __forceinline__ void fold_rows_u(const float* __restrict__ p1, const float* __restrict__ p2, int rows, int C, int rl,
                                            double& a1, double& a2) {
  for (int r = rl; r < rows; r += U * RL) {
    float v1[U], v2[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int ru = r + RL * u;
      const int rc = ru < rows ? ru : rows - 1;      // unconditional loads (a branch would serialise them)
      v1[u] = p1[(size_t)rc * C];
      v2[u] = p2[(size_t)rc * C];
    }
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const bool live = r + RL * u < rows;
      a1 += live ? (double)v1[u] : 0.0;
      a2 += live ? (double)v2[u] : 0.0;
    }
  }
}
template <int RL>
__device__ __forceinline__ void fold_rows(const float* __restrict__ p, int rows, int C, int c, int rl, double& a1,
                                          double& a2) {
  const float* p1 = p + c;
  const float* p2 = p + (size_t)rows * C + c;
  // The fold is a chain of L2-miss rounds (~1.5 us each): as many rows per lane in flight as the lane has (round 5: with four
  // a 1024-row backward fold took four rounds, profiles/r05_bn_fold_probe.txt) — but no more: the loads are unconditional,
  // a 48-row fold on 16 row lanes must not issue 16 of them.  Additions in row order whatever the unroll.
  const int per = (rows + RL - 1) / RL;     // block-uniform
  if (per <= 4) fold_rows_u<RL, 4>(p1, p2, rows, C, rl, a1, a2);
  else if (per <= 8) fold_rows_u<RL, 8>(p1, p2, rows, C, rl, a1, a2);
  else fold_rows_u<RL, 16>(p1, p2, rows, C, rl, a1, a2);
}

// The per-channel arithmetic of the two finalize steps, shared by the stand-alone kernels and the folded ones, with every
// rounding written out: under -ffp-contract=fast hipcc turns `s2 * inv - mu * mu` into an fma on one product or the other
// depending on the surrounding code, and the stand-alone and the folded launch must give the same bits.
__device__ __forceinline__ void bn_fwd_coeffs(double s1, double s2, double inv_count, float eps, float gamma, float beta,
                                              float& muf, float& rs, float& sc, float& sh, double& var) {
  const double mu = __dmul_rn(s1, inv_count);
  var = fma(-mu, mu, __dmul_rn(s2, inv_count));
  if (var < 0.0) var = 0.0;
  muf = (float)mu;
  rs = (float)(1.0 / sqrt(var + (double)eps));
  sc = __fmul_rn(gamma, rs);
  sh = fmaf(-muf, sc, beta);
}
// s1 = sum(dz), s2 = sum(dz * xhat) (or sum(dz * y) with dzy_form) -> sdz, sdzx, the three apply coefficients
__device__ __forceinline__ void bn_bwd_coeffs(double s1, double s2, int dzy_form, double inv_m, float g, float mu, float rs,
                                              float& sdz, float& sdzx, float& c1, float& c2, float& c3) {
  if (dzy_form) s2 = __dmul_rn((double)rs, fma(-(double)mu, s1, s2));   // sum(dz * xhat) = rstd * (sum(dz*y) - mean * sum(dz))
  sdz = (float)s1;
  sdzx = (float)s2;
  const float m1 = (float)__dmul_rn(s1, inv_m);
  const float m2 = (float)__dmul_rn(s2, inv_m);
  c1 = __fmul_rn(g, rs);
  c2 = __fmul_rn(__fmul_rn(-c1, rs), m2);
  c3 = fmaf(-c2, mu, __fmul_rn(-c1, m1));
}

// CW channels per block, 256 / CW row-lanes each; fp64 accumulation across partial rows.  CW = 4 keeps many blocks
// for narrow layers; CW = 16 (64-byte row segments) for the wide ones, whose 16-byte segments made the fold
// transaction-bound (2048 channels x 768 rows: 78 us).
template <int CW>
__global__ __launch_bounds__(256) void bn_finalize_kernel(const float* __restrict__ stats, int rows,
                                                          double inv_count, double unbias, int C, int Creal,
                                                          const float* __restrict__ gamma,
                                                          const float* __restrict__ beta,
                                                          float* running_mean, float* running_var,
                                                          int64_t* nbt, float momentum, float eps,
                                                          float* mean, float* rstd, float* scale,
                                                          float* shift) {
  __shared__ double red[2][256];
  constexpr int RL = 256 / CW;
  const int tid = threadIdx.x;
  const int cl = tid % CW, rl = tid / CW;
  const int c = blockIdx.x * CW + cl;
  double a1 = 0.0, a2 = 0.0;
  // the per-channel parameters are asked for BEFORE the fold (round 6): behind it they were one more dependent miss on a launch
  // that sits on the step's critical chain 88 (ResNet-50) ... 600 (HRNet-W48) times
  const bool owner = rl == 0 && c < Creal;
  float g_c = 0.f, b_c = 0.f, rm_c = 0.f, rv_c = 0.f;
  if (owner) {
    g_c = gamma[c];
    b_c = beta[c];
    if (running_mean != nullptr) { rm_c = running_mean[c]; rv_c = running_var[c]; }
  }
  if (c < C) {
    fold_rows<RL>(stats, rows, C, c, rl, a1, a2);
  }
  red[0][tid] = a1;
  red[1][tid] = a2;
  __syncthreads();
  for (int s = RL / 2; s > 0; s >>= 1) {
    if (rl < s) {
      red[0][tid] += red[0][tid + s * CW];
      red[1][tid] += red[1][tid + s * CW];
    }
    __syncthreads();
  }
  if (rl == 0 && c >= Creal && c < C) {   // padding channel (num_features not a multiple of 8)
    mean[c] = 0.f; rstd[c] = 0.f; scale[c] = 0.f; shift[c] = 0.f;
  }
  if (owner) {
    float muf, rs, sc, sh;
    double var;
    bn_fwd_coeffs(red[0][tid], red[1][tid], inv_count, eps, g_c, b_c, muf, rs, sc, sh, var);
    mean[c] = muf;
    rstd[c] = rs;
    scale[c] = sc;
    shift[c] = sh;
    if (running_mean != nullptr) {
      running_mean[c] = fmaf(momentum, muf, __fmul_rn(1.f - momentum, rm_c));
      running_var[c] = fmaf(momentum, (float)__dmul_rn(var, unbias), __fmul_rn(1.f - momentum, rv_c));
    }
  }
  if (nbt != nullptr && blockIdx.x == 0 && tid == 0) *nbt += 1;
}

__global__ void bn_eval_coeffs_kernel(const float* gamma, const float* beta, const float* rm,
                                      const float* rv, float eps, int C, int Creal, float* scale, float* shift) {
  const int c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c >= Creal && c < C) { scale[c] = 0.f; shift[c] = 0.f; }
  if (c < Creal) {
    const float rs = 1.0f / sqrtf(rv[c] + eps);
    const float sc = gamma[c] * rs;
    scale[c] = sc;
    shift[c] = fmaf(-rm[c], sc, beta[c]);
  }
}

// partial[2][gridDim.x][C] = (sum dz, sum dz * xhat)
__global__ __launch_bounds__(256) void bn_bwd_reduce_kernel(
    const bf16* __restrict__ dout, const bf16* __restrict__ y, const uint8_t* __restrict__ mask,
    const float* __restrict__ scale, const float* __restrict__ shift, const float* __restrict__ mean,
    const float* __restrict__ rstd, int relu, int64_t M, int C, int cge, int rpb,
    float* __restrict__ partial) {
  __shared__ float red[2][256][8];
  const int tid = threadIdx.x;
  const int cgl = tid % cge, rl = tid / cge;
  const int cg_total = C >> 3;
  for (int cg = cgl; cg < cg_total; cg += cge) {
    float s1[8], s2[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) { s1[e] = 0.f; s2[e] = 0.f; }
    if (rl < rpb) {
      float sc[8], sh[8], mu[8], rs[8];
      load8f(scale + cg * 8, sc);
      load8f(shift + cg * 8, sh);
      load8f(mean + cg * 8, mu);
      load8f(rstd + cg * 8, rs);
      for (int64_t m = (int64_t)blockIdx.x * rpb + rl; m < M; m += (int64_t)gridDim.x * rpb) {
        const size_t off = (size_t)m * C + cg * 8;
        const bf16x8 g = ldg16(dout + off);
        const bf16x8 v = ldg16(y + off);
        if (relu && mask != nullptr) {
          const unsigned bits = mask[(size_t)m * cg_total + cg];
#pragma unroll
          for (int e = 0; e < 8; ++e) {
            const float yf = bf2f(v[e]);
            const float dz = ((bits >> e) & 1u) ? bf2f(g[e]) : 0.f;
            s1[e] += dz;
            s2[e] = fmaf(dz, (yf - mu[e]) * rs[e], s2[e]);
          }
        } else {
#pragma unroll
          for (int e = 0; e < 8; ++e) {
            const float yf = bf2f(v[e]);
            float dz = bf2f(g[e]);
            if (relu && !(fmaf(yf, sc[e], sh[e]) > 0.f)) dz = 0.f;
            s1[e] += dz;
            s2[e] = fmaf(dz, (yf - mu[e]) * rs[e], s2[e]);
          }
        }
      }
    }
#pragma unroll
    for (int e = 0; e < 8; ++e) { red[0][tid][e] = s1[e]; red[1][tid][e] = s2[e]; }
    __syncthreads();
    if (rl == 0) {
      for (int r = 1; r < rpb; ++r)
#pragma unroll
        for (int e = 0; e < 8; ++e) { s1[e] += red[0][r * cge + cgl][e]; s2[e] += red[1][r * cge + cgl][e]; }
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        partial[((size_t)0 * gridDim.x + blockIdx.x) * C + cg * 8 + e] = s1[e];
        partial[((size_t)1 * gridDim.x + blockIdx.x) * C + cg * 8 + e] = s2[e];
      }
    }
    __syncthreads();
  }
}

template <int CW>
__global__ __launch_bounds__(256) void bn_bwd_finalize_kernel(
    const float* __restrict__ partial, int rows, double inv_m, int C, int Creal, const float* __restrict__ gamma,
    const float* __restrict__ mean, const float* __restrict__ rstd, float* dgamma, float* dbeta,
    float* coef, int accumulate, int dzy_form) {
  __shared__ double red[2][256];
  constexpr int RL = 256 / CW;
  const int tid = threadIdx.x;
  const int cl = tid % CW, rl = tid / CW;
  const int c = blockIdx.x * CW + cl;
  double a1 = 0.0, a2 = 0.0;
  const bool owner = rl == 0 && c < Creal;      // parameters asked for before the fold, as in bn_finalize_kernel
  float g_c = 0.f, mu_c = 0.f, rs_c = 0.f, dg_c = 0.f, db_c = 0.f;
  if (owner) {
    g_c = gamma[c]; mu_c = mean[c]; rs_c = rstd[c];
    if (accumulate && dgamma != nullptr) dg_c = dgamma[c];
    if (accumulate && dbeta != nullptr) db_c = dbeta[c];
  }
  if (c < C) {
    fold_rows<RL>(partial, rows, C, c, rl, a1, a2);
  }
  red[0][tid] = a1;
  red[1][tid] = a2;
  __syncthreads();
  for (int s = RL / 2; s > 0; s >>= 1) {
    if (rl < s) {
      red[0][tid] += red[0][tid + s * CW];
      red[1][tid] += red[1][tid + s * CW];
    }
    __syncthreads();
  }
  if (rl == 0 && c >= Creal && c < C) {
    coef[c] = 0.f; coef[C + c] = 0.f; coef[2 * C + c] = 0.f;
  }
  if (owner) {
    float sdz, sdzx, c1, c2, c3;
    bn_bwd_coeffs(red[0][tid], red[1][tid], dzy_form, inv_m, g_c, mu_c, rs_c, sdz, sdzx, c1, c2, c3);
    if (dgamma != nullptr) dgamma[c] = accumulate ? dg_c + sdzx : sdzx;
    if (dbeta != nullptr) dbeta[c] = accumulate ? db_c + sdz : sdz;
    coef[c] = c1;
    coef[C + c] = c2;
    coef[2 * C + c] = c3;
  }
}

// dout / dshortcut may alias (in-place masking of the incoming gradient): no restrict on them.
__global__ __launch_bounds__(256) void bn_bwd_apply_kernel(
    const bf16* dout, const bf16* __restrict__ y, const uint8_t* __restrict__ mask,
    const float* __restrict__ scale, const float* __restrict__ shift, const float* __restrict__ coef,
    int relu, bf16* __restrict__ dy, bf16* dshortcut, int ds_acc, int64_t M, int C, int cge, int rpb) {
  const int tid = threadIdx.x;
  const int cgl = tid % cge, rl = tid / cge;
  if (rl >= rpb) return;
  const int cg_total = C >> 3;
  for (int cg = cgl; cg < cg_total; cg += cge) {
    float sc[8], sh[8], c1[8], c2[8], c3[8];
    load8f(scale + cg * 8, sc);
    load8f(shift + cg * 8, sh);
    load8f(coef + cg * 8, c1);
    load8f(coef + C + cg * 8, c2);
    load8f(coef + 2 * C + cg * 8, c3);
    for (int64_t m = (int64_t)blockIdx.x * rpb + rl; m < M; m += (int64_t)gridDim.x * rpb) {
      const size_t off = (size_t)m * C + cg * 8;
      const bf16x8 g = ldg16(dout + off);
      const bf16x8 v = ldg16(y + off);
      float dz[8];
      if (relu && mask != nullptr) {
        const unsigned bits = mask[(size_t)m * cg_total + cg];
#pragma unroll
        for (int e = 0; e < 8; ++e) dz[e] = ((bits >> e) & 1u) ? bf2f(g[e]) : 0.f;
      } else {
#pragma unroll
        for (int e = 0; e < 8; ++e) {
          dz[e] = bf2f(g[e]);
          if (relu && !(fmaf(bf2f(v[e]), sc[e], sh[e]) > 0.f)) dz[e] = 0.f;
        }
      }
      bf16x8 o;
#pragma unroll
      for (int e = 0; e < 8; ++e) o[e] = f2bf(fmaf(c1[e], dz[e], fmaf(c2[e], bf2f(v[e]), c3[e])));
      stg16(dy + off, o);
      if (dshortcut != nullptr) {
        bf16x8 d;
        if (ds_acc) {
          const bf16x8 old = ldg16(dshortcut + off);
#pragma unroll
          for (int e = 0; e < 8; ++e) d[e] = f2bf(dz[e] + bf2f(old[e]));
        } else {
#pragma unroll
          for (int e = 0; e < 8; ++e) d[e] = f2bf(dz[e]);
        }
        stg16(dshortcut + off, d);
      }
    }
  }
}



// ---- stem: BatchNorm + ReLU + 3x3/s2/p1 max-pool as ONE pass (forward) and pool-gather + BatchNorm backward as the
// reduce / apply pair (backward).  The activated map z = relu(bn(y)) (ResNet stem: 112 x 112 x 64 per image, the largest
// tensor of the network) is never written: forward reads y and writes the pooled map + tap indices, backward rebuilds
// d(z) per position from the <= 4 windows that contain it.  Arithmetic and summation order are those of
// bn_act_fwd -> maxpool_fwd resp. maxpool_bwd -> bn_bwd_reduce -> bn_bwd_apply, so the results are bit-identical.
__global__ __launch_bounds__(256) void bn_relu_maxpool_fwd_kernel(const bf16* __restrict__ y,
                                                                  const float* __restrict__ scale,
                                                                  const float* __restrict__ shift, bf16* __restrict__ out,
                                                                  uint8_t* __restrict__ argmax, bf16* __restrict__ ypool,
                                                                  int N, int H, int W, int C, int P, int Q) {
  const int cg_total = C >> 3;
  const size_t total = (size_t)N * P * Q * cg_total;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
    const int cg = (int)(i % cg_total);
    size_t pix = i / cg_total;
    const int q = (int)(pix % Q);
    pix /= Q;
    const int p = (int)(pix % P);
    const int n = (int)(pix / P);
    float sc[8], sh[8];
    load8f(scale + cg * 8, sc);
    load8f(shift + cg * 8, sh);
    float best[8], ybest[8];
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
        const bf16x8 v = ldg16(y + (((size_t)n * H + h) * W + w) * C + cg * 8);
        if (first) {
#pragma unroll
          for (int e = 0; e < 8; ++e) { best[e] = -INFINITY; ybest[e] = 0.f; idx[e] = r * 3 + s; }
          first = false;
        }
#pragma unroll
        for (int e = 0; e < 8; ++e) {
          const float f = bf2f(f2bf(fmaxf(fmaf(bf2f(v[e]), sc[e], sh[e]), 0.f)));   // the bf16 value bn_act_fwd stores
          if (f > best[e] || f != f) { best[e] = f; ybest[e] = bf2f(v[e]); idx[e] = r * 3 + s; }
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
    stg16(out + off, o);
    *reinterpret_cast<uint64_t*>(argmax + off) = packed;
    if (ypool != nullptr) {      // raw conv output at the winning tap: the backward statistics read it instead of gathering y
      bf16x8 yo;
#pragma unroll
      for (int e = 0; e < 8; ++e) yo[e] = f2bf(ybest[e]);
      stg16(ypool + off, yo);
    }
  }
}

// d(z) of input position m (8 channels), as maxpool_bwd would have stored it (bf16) and the ReLU mask applied
__device__ __forceinline__ void pooled_dz(const bf16* __restrict__ dpool, const uint8_t* __restrict__ argmax,
                                          const bf16x8 yv, const float (&sc)[8], const float (&sh)[8], int n, int h,
                                          int w, int C, int P, int Q, int cg, float (&dz)[8]) {
  float g[8];
#pragma unroll
  for (int e = 0; e < 8; ++e) g[e] = 0.f;
  // the <= 2 x 2 windows that contain (h, w): all four (tap index, gradient) pairs are loaded unconditionally (clamped
  // addresses) so that the eight loads are in flight together; a window that does not exist contributes nothing.
  // Accumulation order (p outer, q inner) is that of maxpool_bwd_kernel.
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
      d[a * 2 + b] = ldg16(dpool + off);
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
#pragma unroll
  for (int e = 0; e < 8; ++e) {
    const bool on = bf2f(f2bf(fmaxf(fmaf(bf2f(yv[e]), sc[e], sh[e]), 0.f))) > 0.f;
    dz[e] = on ? bf2f(f2bf(g[e])) : 0.f;
  }
}

// REDUCE: partial[2][gridDim.x][C] = (sum dz, sum dz * xhat), geometry and order of bn_bwd_reduce_kernel
// else:   dy = c1 * dz + c2 * y + c3, geometry of bn_bwd_apply_kernel
template <bool REDUCE>
__global__ __launch_bounds__(256) void bn_pool_bwd_kernel(const bf16* __restrict__ dpool,
                                                          const uint8_t* __restrict__ argmax, const bf16* __restrict__ y,
                                                          const float* __restrict__ scale, const float* __restrict__ shift,
                                                          const float* __restrict__ mean, const float* __restrict__ rstd,
                                                          const float* __restrict__ coef, int64_t M, int H, int W, int C,
                                                          int P, int Q, int cge, int rpb, float* __restrict__ partial,
                                                          bf16* __restrict__ dy) {
  __shared__ float red[REDUCE ? 2 : 1][REDUCE ? 256 : 1][8];
  const int tid = threadIdx.x;
  const int cgl = tid % cge, rl = tid / cge;
  const int cg_total = C >> 3;
  for (int cg = cgl; cg < cg_total; cg += cge) {
    float s1[8], s2[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) { s1[e] = 0.f; s2[e] = 0.f; }
    if (rl < rpb) {
      float sc[8], sh[8], a[8], b[8], c3[8];
      load8f(scale + cg * 8, sc);
      load8f(shift + cg * 8, sh);
      if (REDUCE) {
        load8f(mean + cg * 8, a);
        load8f(rstd + cg * 8, b);
      } else {
        load8f(coef + cg * 8, a);
        load8f(coef + C + cg * 8, b);
        load8f(coef + 2 * C + cg * 8, c3);
      }
      // (n, h, w) of row m advance with the row stride: no division inside the loop
      const int64_t m0 = (int64_t)blockIdx.x * rpb + rl, step = (int64_t)gridDim.x * rpb;
      int w = (int)(m0 % W), h = (int)((m0 / W) % H), n = (int)(m0 / ((int64_t)W * H));
      const int dw = (int)(step % W), dh = (int)((step / W) % H), dn = (int)(step / ((int64_t)W * H));
      for (int64_t m = m0; m < M; m += step) {
        const size_t off = (size_t)m * C + cg * 8;
        const bf16x8 v = ldg16(y + off);
        float dz[8];
        pooled_dz(dpool, argmax, v, sc, sh, n, h, w, C, P, Q, cg, dz);
        w += dw;
        if (w >= W) { w -= W; ++h; }
        h += dh;
        if (h >= H) { h -= H; ++n; }
        n += dn;
        if (REDUCE) {
#pragma unroll
          for (int e = 0; e < 8; ++e) {
            s1[e] += dz[e];
            s2[e] = fmaf(dz[e], (bf2f(v[e]) - a[e]) * b[e], s2[e]);
          }
        } else {
          bf16x8 o;
#pragma unroll
          for (int e = 0; e < 8; ++e) o[e] = f2bf(fmaf(a[e], dz[e], fmaf(b[e], bf2f(v[e]), c3[e])));
          stg16(dy + off, o);
        }
      }
    }
    if (REDUCE) {
#pragma unroll
      for (int e = 0; e < 8; ++e) { red[0][tid][e] = s1[e]; red[1][tid][e] = s2[e]; }
      __syncthreads();
      if (rl == 0) {
        for (int r = 1; r < rpb; ++r)
#pragma unroll
          for (int e = 0; e < 8; ++e) { s1[e] += red[0][r * cge + cgl][e]; s2[e] += red[1][r * cge + cgl][e]; }
#pragma unroll
        for (int e = 0; e < 8; ++e) {
          partial[((size_t)0 * gridDim.x + blockIdx.x) * C + cg * 8 + e] = s1[e];
          partial[((size_t)1 * gridDim.x + blockIdx.x) * C + cg * 8 + e] = s2[e];
        }
      }
      __syncthreads();
    }
  }
}


// BatchNorm-backward sums of the stem taken in the POOLED domain: every pooled element sends its gradient to exactly one
// position (its winning tap), so sum dz = sum over pooled elements with pooled > 0 of dpool, and sum dz * xhat the same with
// the raw conv output of that tap (ypool, written by the forward).  Reads 3 pooled-size tensors (1/4 of y each) instead
// of gathering four windows per input position.  Differs from the position-domain sums only by the bf16 rounding the
// unfused chain applies when several windows hit one position.  partial[2][gridDim.x][C], geometry of bn_bwd_reduce.
__global__ __launch_bounds__(256) void bn_pool_bwd_reduce_pooled_kernel(const bf16* __restrict__ dpool,
                                                                        const bf16* __restrict__ pooled,
                                                                        const bf16* __restrict__ ypool,
                                                                        const float* __restrict__ mean,
                                                                        const float* __restrict__ rstd, int64_t M, int C,
                                                                        int cge, int rpb, float* __restrict__ partial) {
  __shared__ float red[2][256][8];
  const int tid = threadIdx.x;
  const int cgl = tid % cge, rl = tid / cge;
  const int cg_total = C >> 3;
  for (int cg = cgl; cg < cg_total; cg += cge) {
    float s1[8], s2[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) { s1[e] = 0.f; s2[e] = 0.f; }
    if (rl < rpb) {
      float mu[8], rs[8];
      load8f(mean + cg * 8, mu);
      load8f(rstd + cg * 8, rs);
      for (int64_t m = (int64_t)blockIdx.x * rpb + rl; m < M; m += (int64_t)gridDim.x * rpb) {
        const size_t off = (size_t)m * C + cg * 8;
        const bf16x8 g = ldg16(dpool + off), z = ldg16(pooled + off), yv = ldg16(ypool + off);
#pragma unroll
        for (int e = 0; e < 8; ++e) {
          const float dz = bf2f(z[e]) > 0.f ? bf2f(g[e]) : 0.f;
          s1[e] += dz;
          s2[e] = fmaf(dz, (bf2f(yv[e]) - mu[e]) * rs[e], s2[e]);
        }
      }
    }
#pragma unroll
    for (int e = 0; e < 8; ++e) { red[0][tid][e] = s1[e]; red[1][tid][e] = s2[e]; }
    __syncthreads();
    if (rl == 0) {
      for (int r = 1; r < rpb; ++r)
#pragma unroll
        for (int e = 0; e < 8; ++e) { s1[e] += red[0][r * cge + cgl][e]; s2[e] += red[1][r * cge + cgl][e]; }
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        partial[((size_t)0 * gridDim.x + blockIdx.x) * C + cg * 8 + e] = s1[e];
        partial[((size_t)1 * gridDim.x + blockIdx.x) * C + cg * 8 + e] = s2[e];
      }
    }
    __syncthreads();
  }
}

}  // namespace

static int stream_cap() {   // blocks of the elementwise kernels (TOK_BN_BLOCKS overrides).  1024 = 4 per CU = half the wave slots: the
                            // weight-gradient kernels of the side stream run beside them (2048 measured 1.7 % slower end to end)
  static const int v = [] { const char* e = getenv("TOK_BN_BLOCKS"); return (int)(e ? atoi(e) : 1024); }();
  return v;
}
#define kStreamCap stream_cap()
static const int kReduceCap = 1024;   // partial rows of the reducing kernels

extern "C" int tok_bn_finalize(const float* stats, int rows, int64_t count, int c, int c_real, const float* gamma,
                               const float* beta, float* running_mean, float* running_var,
                               int64_t* nbt, float momentum, float eps, float* mean, float* rstd,
                               float* scale, float* shift, void* stream) {
  TOK_CHECK_ARG(stats && gamma && beta && mean && rstd && scale && shift, "tok_bn_finalize: null pointer");
  if (tok_dbg_skip(1)) return TOK_OK;
  TOK_CHECK_ARG(rows > 0 && count > 0 && c > 0 && c_real > 0 && c_real <= c, "tok_bn_finalize: bad sizes");
  TOK_CHECK_ARG((running_mean == nullptr) == (running_var == nullptr), "tok_bn_finalize: running stats");
  const double unbias = count > 1 ? (double)count / (double)(count - 1) : 1.0;
  if (c >= 512)
    hipLaunchKernelGGL(bn_finalize_kernel<16>, dim3((c + 15) / 16), dim3(256), 0, tok_stream(stream), stats, rows,
                       1.0 / (double)count, unbias, c, c_real, gamma, beta, running_mean, running_var, nbt,
                       momentum, eps, mean, rstd, scale, shift);
  else
    hipLaunchKernelGGL(bn_finalize_kernel<4>, dim3((c + 3) / 4), dim3(256), 0, tok_stream(stream), stats, rows,
                       1.0 / (double)count, unbias, c, c_real, gamma, beta, running_mean, running_var, nbt,
                       momentum, eps, mean, rstd, scale, shift);
  TOK_CHECK_LAUNCH("tok_bn_finalize");
  return TOK_OK;
}

extern "C" int tok_bn_eval_coeffs(const float* gamma, const float* beta, const float* running_mean,
                                  const float* running_var, float eps, int c, int c_real, float* scale,
                                  float* shift, void* stream) {
  TOK_CHECK_ARG(gamma && beta && running_mean && running_var && scale && shift, "tok_bn_eval_coeffs: null");
  TOK_CHECK_ARG(c > 0 && c_real > 0 && c_real <= c, "tok_bn_eval_coeffs: bad sizes");
  hipLaunchKernelGGL(bn_eval_coeffs_kernel, dim3((c + 255) / 256), dim3(256), 0, tok_stream(stream), gamma,
                     beta, running_mean, running_var, eps, c, c_real, scale, shift);
  TOK_CHECK_LAUNCH("tok_bn_eval_coeffs");
  return TOK_OK;
}

extern "C" int tok_bn_stats_rows(int64_t m, int c) {
  if (m <= 0 || c <= 0 || c % 8) return TOK_ERR_INVALID;
  return stream_blocks(m, make_geo(c), kReduceCap);
}

extern "C" int tok_bn_stats(const void* y, int64_t m, int c, float* stats, void* stream) {
  TOK_CHECK_ARG(y && stats && m > 0 && c > 0 && c % 8 == 0, "tok_bn_stats: bad args");
  const Geo g = make_geo(c);
  hipLaunchKernelGGL(bn_stats_kernel, dim3(stream_blocks(m, g, kReduceCap)), dim3(256), 0,
                     tok_stream(stream), (const bf16*)y, m, c, g.cge, g.rpb, stats);
  TOK_CHECK_LAUNCH("tok_bn_stats");
  return TOK_OK;
}

extern "C" int tok_bn_act_fwd(const void* y, const float* scale, const float* shift,
                              const void* shortcut, int relu, void* out, uint8_t* mask, int64_t m, int c,
                              void* stream) {
  TOK_CHECK_ARG(y && scale && shift && out && m > 0 && c > 0 && c % 8 == 0, "tok_bn_act_fwd: bad args");
  if (tok_dbg_skip(8)) return TOK_OK;
  if (tok_dbg_skip(32) && !relu && shortcut == nullptr) return TOK_OK;     // ablation: the apply passes of units WITHOUT activation (HRNet's fuse-path terms, projection shortcuts)
  const Geo g = make_geo(c);
  hipLaunchKernelGGL(bn_act_fwd_kernel, dim3(stream_blocks(m, g, kStreamCap)), dim3(256), 0,
                     tok_stream(stream), (const bf16*)y, scale, shift, (const bf16*)shortcut, relu,
                     (bf16*)out, mask, m, c, g.cge, g.rpb, (float*)nullptr);
  TOK_CHECK_LAUNCH("tok_bn_act_fwd");
  return TOK_OK;
}

// tok_bn_act_fwd that also leaves per-block column sums of `out` (fp32 sums of the stored bf16 values):
// partial[tok_bn_act_fwd_colsum_rows(m, c)][c], to be folded by tok_colsum_f32
extern "C" int tok_bn_act_fwd_colsum_rows(int64_t m, int c) {
  if (m <= 0 || c <= 0 || c % 8 != 0) return 0;
  return stream_blocks(m, make_geo(c), kStreamCap);
}

extern "C" int tok_bn_act_fwd_colsum(const void* y, const float* scale, const float* shift, const void* shortcut, int relu,
                                     void* out, uint8_t* mask, int64_t m, int c, float* partial, void* stream) {
  TOK_CHECK_ARG(y && scale && shift && out && partial && m > 0 && c > 0 && c % 8 == 0, "tok_bn_act_fwd_colsum: bad args");
  TOK_CHECK_ARG(c <= 2048, "tok_bn_act_fwd_colsum: c <= 2048 (one channel-group pass per thread: uniform barriers)");
  if (tok_dbg_skip(8)) return TOK_OK;
  const Geo g = make_geo(c);
  hipLaunchKernelGGL(bn_act_fwd_kernel, dim3(stream_blocks(m, g, kStreamCap)), dim3(256), 0, tok_stream(stream), (const bf16*)y,
                     scale, shift, (const bf16*)shortcut, relu, (bf16*)out, mask, m, c, g.cge, g.rpb, partial);
  TOK_CHECK_LAUNCH("tok_bn_act_fwd_colsum");
  return TOK_OK;
}

extern "C" int tok_bn_bwd_rows(int64_t m, int c) { return tok_bn_stats_rows(m, c); }

extern "C" int tok_bn_bwd_reduce(const void* dout, const void* y, const uint8_t* mask, const float* scale,
                                 const float* shift, const float* mean, const float* rstd, int relu,
                                 int64_t m, int c, float* partial, void* stream) {
  TOK_CHECK_ARG(dout && y && scale && shift && mean && rstd && partial, "tok_bn_bwd_reduce: null pointer");
  TOK_CHECK_ARG(m > 0 && c > 0 && c % 8 == 0, "tok_bn_bwd_reduce: bad sizes");
  const Geo g = make_geo(c);
  hipLaunchKernelGGL(bn_bwd_reduce_kernel, dim3(stream_blocks(m, g, kReduceCap)), dim3(256), 0,
                     tok_stream(stream), (const bf16*)dout, (const bf16*)y, mask, scale, shift,
                     mean, rstd, relu, m, c, g.cge, g.rpb, partial);
  TOK_CHECK_LAUNCH("tok_bn_bwd_reduce");
  return TOK_OK;
}

extern "C" int tok_bn_bwd_finalize(const float* partial, int rows, int64_t m, int c, int c_real, const float* gamma,
                                   const float* mean, const float* rstd, float* dgamma, float* dbeta,
                                   float* coef, int accumulate, int dzy_form, void* stream) {
  TOK_CHECK_ARG(partial && gamma && mean && rstd && coef, "tok_bn_bwd_finalize: null pointer");
  if (tok_dbg_skip(1)) return TOK_OK;
  TOK_CHECK_ARG(c > 0 && c_real > 0 && c_real <= c, "tok_bn_bwd_finalize: bad sizes");
  if (c >= 512)
    hipLaunchKernelGGL(bn_bwd_finalize_kernel<16>, dim3((c + 15) / 16), dim3(256), 0, tok_stream(stream), partial,
                       rows, 1.0 / (double)m, c, c_real, gamma, mean, rstd, dgamma, dbeta, coef, accumulate, dzy_form);
  else
    hipLaunchKernelGGL(bn_bwd_finalize_kernel<4>, dim3((c + 3) / 4), dim3(256), 0, tok_stream(stream), partial,
                       rows, 1.0 / (double)m, c, c_real, gamma, mean, rstd, dgamma, dbeta, coef, accumulate, dzy_form);
  TOK_CHECK_LAUNCH("tok_bn_bwd_finalize");
  return TOK_OK;
}

extern "C" int tok_bn_bwd_apply(const void* dout, const void* y, const uint8_t* mask, const float* scale,
                                const float* shift, const float* coef, int relu, void* dy,
                                void* dshortcut, int dshortcut_accumulate, int64_t m, int c,
                                void* stream) {
  TOK_CHECK_ARG(dout && y && scale && shift && coef && dy, "tok_bn_bwd_apply: null pointer");
  TOK_CHECK_ARG(m > 0 && c > 0 && c % 8 == 0, "tok_bn_bwd_apply: bad sizes");
  if (tok_dbg_skip(8)) { t_done_event = nullptr; return TOK_OK; }
  const Geo g = make_geo(c);
  if (t_done_event != nullptr) {
    // completion event attached to THIS dispatch (hipExtLaunchKernelGGL stop event): a side stream can wait for the kernel
    // without a separate event-record packet on the main queue (a ~7 us bubble per fork)
    hipEvent_t ev = t_done_event;
    t_done_event = nullptr;
    hipExtLaunchKernelGGL(bn_bwd_apply_kernel, dim3(stream_blocks(m, g, kStreamCap)), dim3(256), 0, tok_stream(stream), nullptr, ev,
                          0, (const bf16*)dout, (const bf16*)y, mask, scale, shift, coef, relu, (bf16*)dy, (bf16*)dshortcut,
                          dshortcut_accumulate, m, c, g.cge, g.rpb);
  } else {
    hipLaunchKernelGGL(bn_bwd_apply_kernel, dim3(stream_blocks(m, g, kStreamCap)), dim3(256), 0,
                       tok_stream(stream), (const bf16*)dout, (const bf16*)y, mask, scale, shift,
                       coef, relu, (bf16*)dy, (bf16*)dshortcut, dshortcut_accumulate, m, c, g.cge, g.rpb);
  }
  TOK_CHECK_LAUNCH("tok_bn_bwd_apply");
  return TOK_OK;
}


// ---- completion events carried by a launch (two-stream schedule without record packets) ------------------------------------
extern "C" void* tok_event_create(void) {
  hipEvent_t ev = nullptr;
  if (hipEventCreateWithFlags(&ev, hipEventDisableTiming) != hipSuccess) return nullptr;
  return (void*)ev;
}
extern "C" int tok_event_destroy(void* ev) { return ev && hipEventDestroy((hipEvent_t)ev) == hipSuccess ? TOK_OK : TOK_ERR_INVALID; }
// the NEXT tok_bn_bwd_apply launch of the calling thread signals `ev` on completion
extern "C" int tok_next_launch_event(void* ev) { t_done_event = (hipEvent_t)ev; return TOK_OK; }
extern "C" int tok_stream_wait_event(void* stream, void* ev) {
  TOK_CHECK_ARG(ev != nullptr, "tok_stream_wait_event: null event");
  if (hipStreamWaitEvent(tok_stream(stream), (hipEvent_t)ev, 0) != hipSuccess) { tok_set_error("hipStreamWaitEvent failed"); return TOK_ERR_INVALID; }
  return TOK_OK;
}

// ---- stem: BatchNorm + ReLU + max-pool fused (see the kernels) ---------------------------------------------------------------
extern "C" int tok_bn_relu_maxpool_fwd(const void* y, const float* scale, const float* shift, int n, int h, int w, int c,
                                       void* pooled, uint8_t* argmax, void* ypool, void* stream) {
  TOK_CHECK_ARG(y && scale && shift && pooled && argmax && n > 0 && h > 0 && w > 0 && c > 0 && c % 8 == 0,
                "tok_bn_relu_maxpool_fwd: bad args");
  const int p = (h + 2 - 3) / 2 + 1, q = (w + 2 - 3) / 2 + 1;
  const size_t total = (size_t)n * p * q * (c >> 3);
  size_t blocks = (total + 255) / 256;
  if (blocks > 8192) blocks = 8192;
  hipLaunchKernelGGL(bn_relu_maxpool_fwd_kernel, dim3((unsigned)blocks), dim3(256), 0, tok_stream(stream), (const bf16*)y,
                     scale, shift, (bf16*)pooled, argmax, (bf16*)ypool, n, h, w, c, p, q);
  TOK_CHECK_LAUNCH("tok_bn_relu_maxpool_fwd");
  return TOK_OK;
}

extern "C" int tok_bn_pool_bwd_reduce(const void* dpool, const uint8_t* argmax, const void* y, const float* scale,
                                      const float* shift, const float* mean, const float* rstd, int n, int h, int w, int c,
                                      float* partial, void* stream) {
  TOK_CHECK_ARG(dpool && argmax && y && scale && shift && mean && rstd && partial && n > 0 && h > 0 && w > 0 && c > 0 &&
                c % 8 == 0, "tok_bn_pool_bwd_reduce: bad args");
  const int64_t m = (int64_t)n * h * w;
  const Geo g = make_geo(c);
  hipLaunchKernelGGL(bn_pool_bwd_kernel<true>, dim3(stream_blocks(m, g, kReduceCap)), dim3(256), 0, tok_stream(stream),
                     (const bf16*)dpool, argmax, (const bf16*)y, scale, shift, mean, rstd, (const float*)nullptr, m, h, w, c,
                     (h + 2 - 3) / 2 + 1, (w + 2 - 3) / 2 + 1, g.cge, g.rpb, partial, (bf16*)nullptr);
  TOK_CHECK_LAUNCH("tok_bn_pool_bwd_reduce");
  return TOK_OK;
}

extern "C" int tok_bn_pool_bwd_apply(const void* dpool, const uint8_t* argmax, const void* y, const float* scale,
                                     const float* shift, const float* coef, int n, int h, int w, int c, void* dy,
                                     void* stream) {
  TOK_CHECK_ARG(dpool && argmax && y && scale && shift && coef && dy && n > 0 && h > 0 && w > 0 && c > 0 && c % 8 == 0,
                "tok_bn_pool_bwd_apply: bad args");
  const int64_t m = (int64_t)n * h * w;
  const Geo g = make_geo(c);
  hipLaunchKernelGGL(bn_pool_bwd_kernel<false>, dim3(stream_blocks(m, g, kStreamCap)), dim3(256), 0, tok_stream(stream),
                     (const bf16*)dpool, argmax, (const bf16*)y, scale, shift, (const float*)nullptr, (const float*)nullptr,
                     coef, m, h, w, c, (h + 2 - 3) / 2 + 1, (w + 2 - 3) / 2 + 1, g.cge, g.rpb, (float*)nullptr, (bf16*)dy);
  TOK_CHECK_LAUNCH("tok_bn_pool_bwd_apply");
  return TOK_OK;
}

extern "C" int tok_bn_pool_bwd_reduce_pooled(const void* dpool, const void* pooled, const void* ypool, const float* mean,
                                             const float* rstd, int64_t m_pooled, int c, float* partial, void* stream) {
  TOK_CHECK_ARG(dpool && pooled && ypool && mean && rstd && partial && m_pooled > 0 && c > 0 && c % 8 == 0,
                "tok_bn_pool_bwd_reduce_pooled: bad args");
  const Geo g = make_geo(c);
  hipLaunchKernelGGL(bn_pool_bwd_reduce_pooled_kernel, dim3(stream_blocks(m_pooled, g, kReduceCap)), dim3(256), 0,
                     tok_stream(stream), (const bf16*)dpool, (const bf16*)pooled, (const bf16*)ypool, mean, rstd, m_pooled, c,
                     g.cge, g.rpb, partial);
  TOK_CHECK_LAUNCH("tok_bn_pool_bwd_reduce_pooled");
  return TOK_OK;
}
