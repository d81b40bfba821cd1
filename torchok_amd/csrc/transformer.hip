// Token-major kernels of the SwinV2 row: swin.py:71-256 over [timm 0.6.13]
// swin_transformer_v2 (SwinTransformerBlock / WindowAttention / PatchMerging).
//   layernorm fwd/bwd       res-post-norm  x = shortcut + drop_path(LN(y))  (per-sample scale = stochastic depth)
//   act fwd/bwd             GELU (erf) of the MLP, ReLU of the cpb_mlp
//   window attention        cosine attention with learned logit scale, continuous relative position bias and the
//                           shifted-window mask; roll / window_partition / window_reverse are index arithmetic on
//                           the token grid (exact), nothing is permuted in HBM
//   cpb bias                16 * sigmoid(table)[relative_position_index]  (exact gather) and its transpose
//   patch merge             the 2x2 strided gather of PatchMerging (exact permutation) and its inverse
//   colsum_f32              fixed-order reduction of fp32 partial rows (dgamma/dbeta, dbias, dlogit_scale)
// Tokens are rows of a [B*H*W][C] bf16 matrix; statistics and softmax in fp32.  Deterministic: no atomics.
#include "tok_common.h"
#include <math.h>
#include <stdlib.h>

namespace {

// Reductions over the 16 lanes of a DPP row (the lanes that share lane >> 4: the 16 keys of an accumulator column block), every
// lane gets the result: four rotate-and-combine steps on the VALU (row_ror 8 / 4 / 2 / 1).  A __shfl_xor butterfly costs five
// VALU instructions + a ds_bpermute round trip per step, all on the unit's dependent chain.
__device__ __forceinline__ float row16_ror(float v, int sel) {
  switch (sel) {
    case 8: return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x128, 0xF, 0xF, false));
    case 4: return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x124, 0xF, 0xF, false));
    case 2: return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x122, 0xF, 0xF, false));
    default: return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x121, 0xF, 0xF, false));
  }
}
__device__ __forceinline__ float row16_sum(float v) {
  v += row16_ror(v, 8); v += row16_ror(v, 4); v += row16_ror(v, 2); v += row16_ror(v, 1);
  return v;
}
__device__ __forceinline__ float row16_max(float v) {
  v = fmaxf(v, row16_ror(v, 8)); v = fmaxf(v, row16_ror(v, 4)); v = fmaxf(v, row16_ror(v, 2)); v = fmaxf(v, row16_ror(v, 1));
  return v;
}

// sum over the LPR (16 / 32 / 64) consecutive lanes that hold one LayerNorm row
template <int LPR>
__device__ __forceinline__ float lpr_sum(float v) {
  v = row16_sum(v);
  if constexpr (LPR > 16) v += __shfl_xor(v, 16, 64);
  if constexpr (LPR > 32) v += __shfl_xor(v, 32, 64);
  return v;
}

// ---------------------------------------------------------------------------------------------------------------
// LayerNorm.  16-byte vector lanes: a row of c channels is covered by LPR lanes (16 / 32 / 64) holding VPL vectors
// of 8 channels each, so a wave normalises 64 / LPR rows at once and every global access is a full 16-byte lane
// (c % 8 == 0, c <= 1024).  Other widths (HRNet never, SwinV2 never) use the scalar one-wave-per-row kernels.
template <int LPR, int VPL>
__global__ __launch_bounds__(256) void ln_fwd_vec_kernel(const bf16* __restrict__ x, const bf16* __restrict__ shortcut,
                                                         const float* __restrict__ row_scale, int rows_per_sample,
                                                         const float* __restrict__ gamma, const float* __restrict__ beta,
                                                         bf16* __restrict__ out, float* __restrict__ mean,
                                                         float* __restrict__ rstd, int64_t rows, int c, float eps) {
  constexpr int RPW = 64 / LPR;                 // rows per wave
  const int lane = threadIdx.x & 63, sub = lane % LPR;
  const int64_t row = ((int64_t)blockIdx.x * 4 + (threadIdx.x >> 6)) * RPW + lane / LPR;
  const bool live = row < rows;
  const int cg = c >> 3;
  float v[VPL][8];
  float s = 0.f;
#pragma unroll
  for (int u = 0; u < VPL; ++u) {
    const int g = sub + u * LPR;
    if (live && g < cg) {
      const bf16x8 t = ldg16(x + row * c + g * 8);
#pragma unroll
      for (int e = 0; e < 8; ++e) { v[u][e] = bf2f(t[e]); s += v[u][e]; }
    } else {
#pragma unroll
      for (int e = 0; e < 8; ++e) v[u][e] = 0.f;
    }
  }
  s = lpr_sum<LPR>(s);
  const float mu = s / (float)c;
  float q = 0.f;
#pragma unroll
  for (int u = 0; u < VPL; ++u)
    if (sub + u * LPR < cg)
#pragma unroll
      for (int e = 0; e < 8; ++e) { const float d = v[u][e] - mu; q = fmaf(d, d, q); }
  q = lpr_sum<LPR>(q);
  const float rs = rsqrtf(q / (float)c + eps);
  if (!live) return;
  if (sub == 0) { mean[row] = mu; rstd[row] = rs; }
  const float sc = row_scale ? row_scale[row / rows_per_sample] : 1.f;
#pragma unroll
  for (int u = 0; u < VPL; ++u) {
    const int g = sub + u * LPR;
    if (g >= cg) continue;
    float ga[8], be[8];
    *reinterpret_cast<f32x4*>(ga) = *reinterpret_cast<const f32x4*>(gamma + g * 8);
    *reinterpret_cast<f32x4*>(ga + 4) = *reinterpret_cast<const f32x4*>(gamma + g * 8 + 4);
    *reinterpret_cast<f32x4*>(be) = *reinterpret_cast<const f32x4*>(beta + g * 8);
    *reinterpret_cast<f32x4*>(be + 4) = *reinterpret_cast<const f32x4*>(beta + g * 8 + 4);
    bf16x8 o;
    if (shortcut != nullptr) {
      const bf16x8 sh = ldg16(shortcut + row * c + g * 8);
#pragma unroll
      for (int e = 0; e < 8; ++e) o[e] = f2bf(((v[u][e] - mu) * rs * ga[e] + be[e]) * sc + bf2f(sh[e]));
    } else {
#pragma unroll
      for (int e = 0; e < 8; ++e) o[e] = f2bf(((v[u][e] - mu) * rs * ga[e] + be[e]) * sc);
    }
    stg16(out + row * c + g * 8, o);
  }
}

// backward: dx = rstd * (g - mean(g) - xhat * mean(g * xhat)), g = dout * scale * gamma; each lane keeps the dgamma /
// dbeta contributions of ITS 8*VPL columns in registers across all the rows it visits, then the lanes that own the
// same columns are folded through LDS -> one partial row per block (fixed order: deterministic)
template <int LPR, int VPL>
__global__ __launch_bounds__(256) void ln_bwd_vec_kernel(const bf16* __restrict__ dout, const bf16* __restrict__ x,
                                                         const float* __restrict__ mean, const float* __restrict__ rstd,
                                                         const float* __restrict__ gamma, const float* __restrict__ row_scale,
                                                         int rows_per_sample, bf16* dx, int accumulate,
                                                         float* __restrict__ partial, int64_t rows, int c) {
  constexpr int RPW = 64 / LPR, RPB = 4 * RPW;  // rows per wave / per block pass
  extern __shared__ float sm[];                 // [RPB][2][c]
  const int lane = threadIdx.x & 63, sub = lane % LPR;
  const int rl = (threadIdx.x >> 6) * RPW + lane / LPR;     // row slot inside the block pass
  const int cg = c >> 3;
  float ga[VPL][8], pg[VPL][8], pb[VPL][8];
#pragma unroll
  for (int u = 0; u < VPL; ++u) {
    const int g = sub + u * LPR;
#pragma unroll
    for (int e = 0; e < 8; ++e) { ga[u][e] = g < cg ? gamma[g * 8 + e] : 0.f; pg[u][e] = 0.f; pb[u][e] = 0.f; }
  }
  // two row slots per iteration: all loads of both rows are requested before the first is consumed (one row per iteration
  // left a wave with two 16-byte loads in flight: 2.7 TB/s at 1024 blocks)
  constexpr int UNR = 2;
  for (int64_t row0 = (int64_t)blockIdx.x * RPB; row0 < rows; row0 += (int64_t)gridDim.x * RPB * UNR) {
    int64_t rowv[UNR];
    bool livev[UNR];
    bf16x8 xv[UNR][VPL], gv[UNR][VPL], pv[UNR][VPL];
    float muv[UNR], rsv[UNR], scv[UNR];
#pragma unroll
    for (int r = 0; r < UNR; ++r) {
      rowv[r] = row0 + (int64_t)r * gridDim.x * RPB + rl;
      livev[r] = rowv[r] < rows;
#pragma unroll
      for (int u = 0; u < VPL; ++u) {
        const int g = sub + u * LPR;
        const bool ok = livev[r] && g < cg;
        xv[r][u] = ok ? ldg16(x + rowv[r] * c + g * 8) : zero8();
        gv[r][u] = ok ? ldg16(dout + rowv[r] * c + g * 8) : zero8();
        pv[r][u] = (ok && accumulate) ? ldg16(dx + rowv[r] * c + g * 8) : zero8();
      }
      muv[r] = livev[r] ? mean[rowv[r]] : 0.f;
      rsv[r] = livev[r] ? rstd[rowv[r]] : 0.f;
      scv[r] = (livev[r] && row_scale) ? row_scale[rowv[r] / rows_per_sample] : 1.f;
    }
#pragma unroll
    for (int r = 0; r < UNR; ++r) {
      const int64_t row = rowv[r];
      const bool live = livev[r];
      const float mu = muv[r], rs = rsv[r], sc = scv[r];
      float xh[VPL][8], gg[VPL][8];
      float s1 = 0.f, s2 = 0.f;
#pragma unroll
      for (int u = 0; u < VPL; ++u) {
        const int g = sub + u * LPR;
        if (live && g < cg) {
#pragma unroll
          for (int e = 0; e < 8; ++e) {
            xh[u][e] = (bf2f(xv[r][u][e]) - mu) * rs;
            const float go = bf2f(gv[r][u][e]) * sc;
            gg[u][e] = go * ga[u][e];
            s1 += gg[u][e];
            s2 = fmaf(gg[u][e], xh[u][e], s2);
            pg[u][e] = fmaf(go, xh[u][e], pg[u][e]);
            pb[u][e] += go;
          }
        } else {
#pragma unroll
          for (int e = 0; e < 8; ++e) { xh[u][e] = 0.f; gg[u][e] = 0.f; }
        }
      }
      s1 = lpr_sum<LPR>(s1);
      s2 = lpr_sum<LPR>(s2);
      const float m1 = s1 / (float)c, m2 = s2 / (float)c;
      if (live)
#pragma unroll
        for (int u = 0; u < VPL; ++u) {
          const int g = sub + u * LPR;
          if (g >= cg) continue;
          bf16x8 o;
#pragma unroll
          for (int e = 0; e < 8; ++e) o[e] = f2bf(rs * (gg[u][e] - m1 - xh[u][e] * m2) + bf2f(pv[r][u][e]));
          stg16(dx + row * c + g * 8, o);
        }
    }
  }
#pragma unroll
  for (int u = 0; u < VPL; ++u) {
    const int g = sub + u * LPR;
    if (g < cg)
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        sm[((size_t)rl * 2 + 0) * c + g * 8 + e] = pg[u][e];
        sm[((size_t)rl * 2 + 1) * c + g * 8 + e] = pb[u][e];
      }
  }
  __syncthreads();
  for (int i = threadIdx.x; i < 2 * c; i += 256) {
    const int which = i / c, col = i - which * c;
    float t = 0.f;
    for (int r = 0; r < RPB; ++r) t += sm[((size_t)r * 2 + which) * c + col];
    partial[((size_t)which * gridDim.x + blockIdx.x) * c + col] = t;
  }
}

// scalar fallback: one wave per row.
__global__ __launch_bounds__(256) void ln_fwd_kernel(const bf16* __restrict__ x, const bf16* __restrict__ shortcut,
                                                     const float* __restrict__ row_scale, int rows_per_sample,
                                                     const float* __restrict__ gamma, const float* __restrict__ beta,
                                                     bf16* __restrict__ out, float* __restrict__ mean,
                                                     float* __restrict__ rstd, int64_t rows, int c, int ld, float eps) {
  const int lane = threadIdx.x & 63;
  const int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= rows) return;
  const bf16* xr = x + row * ld;
  float s = 0.f;
  for (int i = lane; i < c; i += 64) s += bf2f(xr[i]);
  const float mu = wave_sum(s) / (float)c;
  float v = 0.f;
  for (int i = lane; i < c; i += 64) { const float d = bf2f(xr[i]) - mu; v = fmaf(d, d, v); }
  const float rs = rsqrtf(wave_sum(v) / (float)c + eps);
  if (lane == 0) { mean[row] = mu; rstd[row] = rs; }
  const float sc = row_scale ? row_scale[row / rows_per_sample] : 1.f;
  bf16* orow = out + row * ld;
  const bf16* srow = shortcut ? shortcut + row * ld : nullptr;
  for (int i = lane; i < ld; i += 64) {
    float o = 0.f;
    if (i < c) {
      o = ((bf2f(xr[i]) - mu) * rs * gamma[i] + beta[i]) * sc;
      if (srow) o += bf2f(srow[i]);
    }
    orow[i] = f2bf(o);
  }
}

// dx = rstd * (g - mean(g) - xhat * mean(g * xhat)),  g = dout * scale * gamma;  partial dgamma/dbeta per block
__global__ __launch_bounds__(256) void ln_bwd_kernel(const bf16* __restrict__ dout, const bf16* __restrict__ x,
                                                     const float* __restrict__ mean, const float* __restrict__ rstd,
                                                     const float* __restrict__ gamma, const float* __restrict__ row_scale,
                                                     int rows_per_sample, bf16* dx, int accumulate,
                                                     float* __restrict__ partial, int64_t rows, int c, int ld) {
  extern __shared__ float sm[];   // [4 waves][2][c]
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  float* pg = sm + (size_t)wv * 2 * c;
  float* pb = pg + c;
  for (int i = lane; i < c; i += 64) { pg[i] = 0.f; pb[i] = 0.f; }
  for (int64_t row = (int64_t)blockIdx.x * 4 + wv; row < rows; row += (int64_t)gridDim.x * 4) {
    const bf16* xr = x + row * ld;
    const bf16* gr = dout + row * ld;
    const float mu = mean[row], rs = rstd[row];
    const float sc = row_scale ? row_scale[row / rows_per_sample] : 1.f;
    float s1 = 0.f, s2 = 0.f;
    for (int i = lane; i < c; i += 64) {
      const float xh = (bf2f(xr[i]) - mu) * rs;
      const float go = bf2f(gr[i]) * sc;
      const float g = go * gamma[i];
      s1 += g;
      s2 = fmaf(g, xh, s2);
      pg[i] = fmaf(go, xh, pg[i]);   // each lane owns its columns: no conflicts inside a wave
      pb[i] += go;
    }
    const float m1 = wave_sum(s1) / (float)c, m2 = wave_sum(s2) / (float)c;
    bf16* dr = dx + row * ld;
    for (int i = lane; i < ld; i += 64) {
      float o = 0.f;
      if (i < c) {
        const float xh = (bf2f(xr[i]) - mu) * rs;
        o = rs * (bf2f(gr[i]) * sc * gamma[i] - m1 - xh * m2);
        if (accumulate) o += bf2f(dr[i]);
      }
      dr[i] = f2bf(o);
    }
  }
  __syncthreads();
  // partial [2][gridDim.x][c]: dgamma rows, then dbeta rows
  for (int i = threadIdx.x; i < 2 * c; i += 256) {
    const float v = sm[i] + sm[2 * c + i] + sm[4 * c + i] + sm[6 * c + i];
    if (i < c) partial[(size_t)blockIdx.x * c + i] = v;
    else partial[((size_t)gridDim.x + blockIdx.x) * c + (i - c)] = v;
  }
}

// dst[col] (+)= sum_r src[r][col], fixed order, fp64 accumulation.  blockIdx.y = 1: the second (src, dst, accumulate) of a pair
// launch (LayerNorm's d(gamma) and d(beta) rows in one launch: 48 small launches less per SwinV2-T step)
template <int U>
__device__ __forceinline__ void colsum_rows(const float* __restrict__ s0, int64_t rows, int cols, int rl, double& a) {
  for (int64_t r = rl; r < rows; r += 16 * U) {
    float v[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int64_t ru = r + 16 * u;
      v[u] = s0[(ru < rows ? ru : rows - 1) * cols];
    }
#pragma unroll
    for (int u = 0; u < U; ++u) a += r + 16 * u < rows ? (double)v[u] : 0.0;
  }
}

__global__ __launch_bounds__(256) void colsum_f32_kernel(const float* __restrict__ src, int64_t rows, int cols,
                                                         float* dst, int accumulate, const float* __restrict__ src1,
                                                         float* dst1, int accumulate1) {
  __shared__ double red[256];
  if (blockIdx.y == 1) { src = src1; dst = dst1; accumulate = accumulate1; }
  const int cl = threadIdx.x & 15, rl = threadIdx.x >> 4;
  const int col = blockIdx.x * 16 + cl;
  double a = 0.0;
  if (col < cols) {
    const float* s0 = src + col;
    // as many loads in flight per lane as it has rows, up to 32; additions in row order (round 5: with 8 in flight the 1024
    // partial rows of an activation pass were eight dependent L2-miss rounds, 6 us on the forward chain of every fused
    // residual unit; the loads are unconditional, so short folds keep the 8-wide form)
    if (rows > 128) colsum_rows<32>(s0, rows, cols, rl, a);
    else colsum_rows<8>(s0, rows, cols, rl, a);
  }
  red[threadIdx.x] = a;
  __syncthreads();
  for (int s = 8; s > 0; s >>= 1) {
    if (rl < s) red[threadIdx.x] += red[threadIdx.x + s * 16];
    __syncthreads();
  }
  if (rl == 0 && col < cols) dst[col] = (float)red[threadIdx.x] + (accumulate ? dst[col] : 0.f);
}

// The same fold for WIDE matrices (window attention's d(bias) partials: 64 ... 1024 rows of 7 203 ... 57 624 columns, 15-30 MB
// per launch, twelve launches per SwinV2-T step).  colsum_f32_kernel gives a block 16 columns: a wave instruction touches four
// rows x 64 bytes, and the fold ran at ~160 GB/s (up to 360 us per launch, 1.2 ms of a step on the position-bias stream).
// Here a block owns 256 columns and a wave instruction reads 256 contiguous bytes of ONE row (lane l: columns l, l + 64,
// l + 128, l + 192 — no alignment condition: 49 x 49 x heads is odd for three heads); the block's sixteen waves take the rows
// round-robin, sixteen loads in flight per lane, and are folded through LDS in wave order; fp64 accumulation in row order per
// wave like the narrow kernel (deterministic; the order of additions differs from the narrow kernel's, which no caller mixes
// on one tensor).
__global__ __launch_bounds__(1024) void colsum_f32_wide_kernel(const float* __restrict__ src, int64_t rows, int cols,
                                                               float* dst, int accumulate) {
  __shared__ double red[16][4][64];
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  const int col0 = blockIdx.x * 256 + lane;
  double a[4] = {0.0, 0.0, 0.0, 0.0};
  constexpr int U = 4;
  for (int64_t r = wv; r < rows; r += 16 * U) {
    float v[U][4];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int64_t ru = r + 16 * u;
      const float* s0 = src + (ru < rows ? ru : rows - 1) * cols;
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const int c = col0 + 64 * e;
        v[u][e] = s0[c < cols ? c : cols - 1];
      }
    }
#pragma unroll
    for (int u = 0; u < U; ++u)
      if (r + 16 * u < rows)
#pragma unroll
        for (int e = 0; e < 4; ++e) a[e] += (double)v[u][e];
  }
#pragma unroll
  for (int e = 0; e < 4; ++e) red[wv][e][lane] = a[e];
  __syncthreads();
  if (wv < 4) {                      // wave e folds column slot e
    const int c = col0 + 64 * wv;
    double t = 0.0;
    for (int w = 0; w < 16; ++w) t += red[w][wv][lane];
    if (c < cols) dst[c] = (float)t + (accumulate ? dst[c] : 0.f);
  }
}

// ---------------------------------------------------------------------------------------------------------------
// activations (kind 0 = ReLU, 1 = GELU erf)
// gelu_f / gelu_d: tok_common.h (shared with the GEMM epilogues of conv_igemm.hip)

__global__ __launch_bounds__(256) void act_fwd_kernel(int kind, const bf16* __restrict__ x, bf16* __restrict__ out,
                                                      size_t n8) {
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n8; i += (size_t)gridDim.x * 256) {
    const bf16x8 v = ldg16(x + i * 8);
    bf16x8 o;
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      const float f = bf2f(v[e]);
      o[e] = f2bf(kind == 0 ? fmaxf(f, 0.f) : gelu_f(f));
    }
    stg16(out + i * 8, o);
  }
}

__global__ __launch_bounds__(256) void act_bwd_kernel(int kind, const bf16* __restrict__ dout,
                                                      const bf16* __restrict__ x, bf16* dx, int accumulate, size_t n8) {
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n8; i += (size_t)gridDim.x * 256) {
    const bf16x8 v = ldg16(x + i * 8), g = ldg16(dout + i * 8);
    bf16x8 o;
    bf16x8 prev = accumulate ? ldg16(dx + i * 8) : zero8();
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      const float f = bf2f(v[e]);
      const float d = kind == 0 ? (f > 0.f ? 1.f : 0.f) : (kind == 1 ? gelu_d(f) : 1.f);
      o[e] = f2bf(bf2f(g[e]) * d + bf2f(prev[e]));
    }
    stg16(dx + i * 8, o);
  }
}

// ---------------------------------------------------------------------------------------------------------------
// window attention.  One wave per (image, window, head); lane = query row (rows loop for N > 64).
struct AttnArgs {
  int B, H, W, C, heads, ws, shift, nWx, nW, N, ld;   // ld: row pitch of qkv (>= 3C); head_dim = 32
  int plain;   // 1: softmax(q k^T / sqrt(head_dim)) v — no cosine normalisation, logit scale, bias (DaViT, davit.py:168-207)
};

constexpr int HD = 32;

__device__ __forceinline__ int64_t token_row(const AttnArgs& a, int b, int win, int t) {
  const int wy = win / a.nWx, wx = win - wy * a.nWx;
  const int iy = t / a.ws, ix = t - iy * a.ws;
  int oy = wy * a.ws + iy + a.shift, ox = wx * a.ws + ix + a.shift;   // roll(-shift): rolled[i] = x[(i + shift) % H]
  oy = oy >= a.H ? oy - a.H : oy;
  ox = ox >= a.W ? ox - a.W : ox;
  return ((int64_t)b * a.H + oy) * a.W + ox;
}

// loads q, k, v of one (b, window, head) into LDS as fp32, q and k L2-normalised (F.normalize eps 1e-12)
__device__ __forceinline__ void load_qkv(const AttnArgs& a, const bf16* __restrict__ qkv, int b, int win, int h,
                                         float* qn, float* kn, float* v, float* qinv, float* kinv) {
  const int lane = threadIdx.x;
  for (int t = lane; t < a.N; t += 64) {
    const bf16* r = qkv + token_row(a, b, win, t) * a.ld + h * HD;
    float sq = 0.f, sk = 0.f;
#pragma unroll
    for (int d = 0; d < HD; d += 8) {
      const bf16x8 q8 = ldg16(r + d), k8 = ldg16(r + a.C + d), v8 = ldg16(r + 2 * a.C + d);
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        const float qf = bf2f(q8[e]), kf = bf2f(k8[e]);
        qn[t * HD + d + e] = qf;
        kn[t * HD + d + e] = kf;
        v[t * HD + d + e] = bf2f(v8[e]);
        sq = fmaf(qf, qf, sq);
        sk = fmaf(kf, kf, sk);
      }
    }
    const float qi = 1.f / fmaxf(sqrtf(sq), 1e-12f), ki = 1.f / fmaxf(sqrtf(sk), 1e-12f);
    if (qinv) { qinv[t] = qi; kinv[t] = ki; }
    for (int d = 0; d < HD; ++d) { qn[t * HD + d] *= qi; kn[t * HD + d] *= ki; }
  }
}

__global__ __launch_bounds__(64) void attn_fwd_kernel(AttnArgs a, const bf16* __restrict__ qkv,
                                                      const float* __restrict__ logit_scale,
                                                      const float* __restrict__ bias, const float* __restrict__ mask,
                                                      bf16* __restrict__ out, float* __restrict__ lse) {
  extern __shared__ float sm[];
  float* qn = sm;
  float* kn = qn + a.N * HD;
  float* v = kn + a.N * HD;
  const int h = blockIdx.x % a.heads;
  const int win = (blockIdx.x / a.heads) % a.nW;
  const int b = blockIdx.x / (a.heads * a.nW);
  load_qkv(a, qkv, b, win, h, qn, kn, v, nullptr, nullptr);
  __syncthreads();
  const float scale = expf(fminf(logit_scale[h], 4.605170185988092f));   // clamp(max = ln 100).exp()
  const float* bh = bias + (size_t)h * a.N * a.N;
  const float* mw = mask ? mask + (size_t)win * a.N * a.N : nullptr;
  for (int i = threadIdx.x; i < a.N; i += 64) {
    float q[HD], o[HD];
#pragma unroll
    for (int d = 0; d < HD; ++d) { q[d] = qn[i * HD + d]; o[d] = 0.f; }
    float mx = -INFINITY, den = 0.f;
    for (int j = 0; j < a.N; ++j) {
      float s = 0.f;
#pragma unroll
      for (int d = 0; d < HD; ++d) s = fmaf(q[d], kn[j * HD + d], s);
      s = s * scale + bh[i * a.N + j] + (mw ? mw[i * a.N + j] : 0.f);
      const float nm = fmaxf(mx, s);
      const float corr = expf(mx - nm), p = expf(s - nm);
      den = den * corr + p;
#pragma unroll
      for (int d = 0; d < HD; ++d) o[d] = fmaf(o[d], corr, p * v[j * HD + d]);
      mx = nm;
    }
    const float inv = 1.f / den;
    bf16* orow = out + token_row(a, b, win, i) * a.C + h * HD;
#pragma unroll
    for (int d = 0; d < HD; d += 8) {
      bf16x8 o8;
#pragma unroll
      for (int e = 0; e < 8; ++e) o8[e] = f2bf(o[d + e] * inv);
      stg16(orow + d, o8);
    }
    lse[((size_t)blockIdx.x) * a.N + i] = mx + logf(den);
  }
}

// ---- MFMA kernels for windows of up to 64 tokens (7x7, 8x8): one 4-wave workgroup per (image, window, head) ------
// S = Qn Kn^T as 16x16 tiles of mfma_f32_16x16x32_bf16 (K = head_dim = 32: one instruction per tile), fp32 softmax on
// the accumulator layout (row = 4*(lane>>4)+reg, col = lane&15), P through LDS (bf16) into the A operand of
// O = P V (V kept transposed in LDS so a lane's 8 k-slots are 8 consecutive keys).  Wave w owns query tile w; the
// four waves stage q / k / v (/ dO) in parallel.  (One wave per unit left < 1 wave per SIMD resident: LDS-bound
// occupancy, every latency exposed.)
constexpr int QPITCH = 40;    // bf16 elements per Q/K row in LDS (32 + 8: spreads 16 rows over the banks)
constexpr int PPITCH = 72;    // P rows / transposed rows: 64 + 8
constexpr int MFMA_FWD_LDS = (3 * 64 * QPITCH + 64 * PPITCH) * 2;   // bytes per workgroup: q, k, v row-major + P^T

// token t of the unit: loads one 32-wide head slice, optionally L2-normalises it (F.normalize, eps 1e-12)
__device__ __forceinline__ float load_head_row(const bf16* __restrict__ src, bool valid, bool normalise, bf16x8 (&o)[4]) {
  float f[HD], ss = 0.f;
  if (valid) {
#pragma unroll
    for (int d = 0; d < HD; d += 8) {
      const bf16x8 t8 = ldg16(src + d);
#pragma unroll
      for (int e = 0; e < 8; ++e) { f[d + e] = bf2f(t8[e]); ss = fmaf(f[d + e], f[d + e], ss); }
    }
  } else {
#pragma unroll
    for (int d = 0; d < HD; ++d) f[d] = 0.f;
  }
  // F.normalize: 1 / max(|x|, 1e-12) = min(rsqrt(|x|^2), 1e12) — one v_rsq_f32 (the forward and the backward's recomputation
  // use the same expression: the recomputed probabilities are normalised by the forward's log-sum-exp)
  const float inv = normalise ? fminf(__builtin_amdgcn_rsqf(ss), 1e12f) : 1.f;
#pragma unroll
  for (int d = 0; d < HD; d += 8)
#pragma unroll
    for (int e = 0; e < 8; ++e) o[d >> 3][e] = f2bf(f[d + e] * inv);
  return inv;
}

// the same in two halves, so that the NEXT image's rows can be in flight while the current one is computed
__device__ __forceinline__ void load_head_raw(const bf16* __restrict__ src, bool valid, bf16x8 (&o)[4]) {
#pragma unroll
  for (int d = 0; d < 4; ++d) o[d] = valid ? ldg16(src + d * 8) : zero8();
}
__device__ __forceinline__ float finish_head_row(bool normalise, bf16x8 (&o)[4]) {
  if (!normalise) return 1.f;
  float f[HD], ss = 0.f;
#pragma unroll
  for (int d = 0; d < HD; d += 8)
#pragma unroll
    for (int e = 0; e < 8; ++e) { f[d + e] = bf2f(o[d >> 3][e]); ss = fmaf(f[d + e], f[d + e], ss); }
  const float inv = fminf(__builtin_amdgcn_rsqf(ss), 1e12f);
#pragma unroll
  for (int d = 0; d < HD; d += 8)
#pragma unroll
    for (int e = 0; e < 8; ++e) o[d >> 3][e] = f2bf(f[d + e] * inv);
  return inv;
}

__device__ __forceinline__ void put_row(bf16* rowmaj, bf16* trans, int t, const bf16x8 (&v)[4]) {
#pragma unroll
  for (int d = 0; d < HD; d += 8) {
    if (rowmaj) *reinterpret_cast<bf16x8*>(rowmaj + t * QPITCH + d) = v[d >> 3];
    if (trans)
#pragma unroll
      for (int e = 0; e < 8; ++e) trans[(d + e) * PPITCH + t] = v[d >> 3][e];
  }
}

typedef __attribute__((address_space(3))) bf16x4 attn_lds_bf16x4;
__device__ __forceinline__ bf16x4 attn_tr4(const bf16* p) { return __builtin_amdgcn_ds_read_tr16_b64_v4bf16((attn_lds_bf16x4*)(p)); }
// fragment of a row-major [token][pitch] tile whose k-slots are tokens 32 ks .. and whose MFMA row / column index is the
// tile column c0 + l15
__device__ __forceinline__ bf16x8 attn_tr_frag(const bf16* tile, int pitch, int ks, int c0, int g, int l15) {
  const bf16* p = tile + (32 * ks + 4 * g + (l15 >> 2)) * pitch + c0 + (l15 & 3) * 4;
  const bf16x4 lo = attn_tr4(p), hi = attn_tr4(p + 16 * pitch);
  bf16x8 r;
#pragma unroll
  for (int e = 0; e < 4; ++e) { r[e] = lo[e]; r[4 + e] = hi[e]; }
  return r;
}
// the same k-slot order read along a ROW of a [row][pitch] tile: tokens 32 ks + 4g .. +3 and 32 ks + 16 + 4g .. +3
__device__ __forceinline__ bf16x8 attn_row_frag(const bf16* tile, int pitch, int row, int ks, int g) {
  const bf16* p = tile + row * pitch + 32 * ks + 4 * g;
  const bf16x4 lo = *reinterpret_cast<const bf16x4*>(p), hi = *reinterpret_cast<const bf16x4*>(p + 16);
  bf16x8 r;
#pragma unroll
  for (int e = 0; e < 4; ++e) { r[e] = lo[e]; r[4 + e] = hi[e]; }
  return r;
}
__device__ __forceinline__ void put_row_major(bf16* rowmaj, int t, const bf16x8 (&v)[4]) {
#pragma unroll
  for (int d = 0; d < HD; d += 8) *reinterpret_cast<bf16x8*>(rowmaj + t * QPITCH + d) = v[d >> 3];
}

// Workgroup -> unit.  A 32-wide head slice of a token row is 64 bytes: the heads of a window share 128-byte lines (and a row's
// q / k / v parts are contiguous), so when consecutive workgroups — which the dispatcher deals round-robin to the eight XCDs,
// each with its own L2 — take consecutive heads, every line is fetched by two L2s: PMC read traffic of the round-3 kernels was
// 2.0x the q, k, v, dO bytes (4.43 GB against 2.2 for the SwinV2-T forward launches of a step).  Unit u = (xcd, slot) with
// xcd = blockIdx % 8 instead: each XCD walks a contiguous range of units, so the heads of one window are neighbours in time on
// ONE L2.  The grid is 8 * ceil(units / 8) workgroups; the surplus ones leave at once.  (Speed only: nothing depends on where
// a workgroup really runs.)
__device__ __forceinline__ int attn_unit(int units) {
  const int per = (units + 7) >> 3;
  return (int)(blockIdx.x & 7) * per + (int)(blockIdx.x >> 3);
}

__global__ __launch_bounds__(256) void attn_fwd_mfma_kernel(AttnArgs a, const bf16* __restrict__ qkv,
                                                            const float* __restrict__ logit_scale,
                                                            const float* __restrict__ bias, const float* __restrict__ mask,
                                                            bf16* __restrict__ out, float* __restrict__ lse, int bpw,
                                                            int units) {
  extern __shared__ char smraw[];
  bf16* qs = reinterpret_cast<bf16*>(smraw);
  bf16* ks = qs + 64 * QPITCH;
  bf16* vs = ks + 64 * QPITCH;          // V row-major [token][QPITCH]
  bf16* pt = vs + 64 * QPITCH;          // P^T [64 keys][PPITCH queries]: a wave owns the 16 columns of its query tile
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  const int unit_id = attn_unit(units);
  if (unit_id >= units) return;
  const int h = unit_id % a.heads;
  const int win = (unit_id / a.heads) % a.nW;
  const int bg = unit_id / (a.heads * a.nW);
  const int N = a.N;
  const int l15 = lane & 15, g = lane >> 4, qi = wv;
  constexpr float LOG2E = 1.4426950408889634f, LN2 = 0.6931471805599453f;
  // logits in log2 units: exp(x) = exp2(x log2 e) on v_exp_f32 (one instruction; libm's expf is ~20)
  const float scale = (a.plain ? 0.17677669529663687f : expf(fminf(logit_scale[h], 4.605170185988092f))) * LOG2E;
  // additive logit terms of this wave's query tile (position bias + shift mask; -inf on padding): loaded once per
  // workgroup, reused for every image it walks.  element (reg, kj): query i = qi*16 + 4g + reg, key j = kj*16 + l15.
  // Straight-line loads on clamped indices + selects (a branch per element costs one exposed L2 round trip each).
  float addt[4][4];
  {
    const float* bh = bias ? bias + (size_t)h * N * N : nullptr;
    const float* mw = mask ? mask + (size_t)win * N * N : nullptr;
    const bool any_add = bh != nullptr || mw != nullptr;
    const float bsel = bh ? 1.f : 0.f, msel = mw ? 1.f : 0.f;
    const float* bp = bh ? bh : mw;
    const float* mp = mw ? mw : bp;
#pragma unroll
    for (int reg = 0; reg < 4; ++reg)
#pragma unroll
      for (int kj = 0; kj < 4; ++kj) {
        const int i = qi * 16 + 4 * g + reg, j = kj * 16 + l15;
        const int idx = (i < N ? i : N - 1) * N + (j < N ? j : N - 1);
        float v = 0.f;
        if (any_add) v = (bsel * bp[idx] + msel * mp[idx]) * LOG2E;
        addt[reg][kj] = (i < N && j < N) ? v : -INFINITY;
      }
  }
  // rows of the image being staged (wave 0: q, 1: k, 2: v) are requested one image ahead, as in the backward
  const int64_t tok_sp = token_row(a, 0, win, lane < N ? lane : 0);
  const int64_t img_rows = (int64_t)a.H * a.W;
  bf16x8 rnext[4];
  auto request = [&](int b) {
    if (wv < 3) load_head_raw(qkv + (tok_sp + (int64_t)b * img_rows) * a.ld + h * HD + wv * a.C, lane < N, rnext);
  };
  if (bg * bpw < a.B) request(bg * bpw);
  for (int bb = 0; bb < bpw; ++bb) {
    const int b = bg * bpw + bb;
    if (b >= a.B) break;                 // uniform for the workgroup
    const size_t unit = ((size_t)b * a.nW + win) * a.heads + h;
    if (wv < 3) {   // wave 0: q (normalised), wave 1: k (normalised), wave 2: v — all row-major
      bf16x8 r8[4];
#pragma unroll
      for (int d = 0; d < 4; ++d) r8[d] = rnext[d];
      finish_head_row(wv < 2 && !a.plain, r8);
      put_row_major(wv == 0 ? qs : (wv == 1 ? ks : vs), lane, r8);
    }
    if (bb + 1 < bpw && b + 1 < a.B) request(b + 1);
    __syncthreads();
    const bf16x8 qf = *reinterpret_cast<const bf16x8*>(qs + (qi * 16 + l15) * QPITCH + g * 8);
    f32x4 sc[4];
#pragma unroll
    for (int kj = 0; kj < 4; ++kj) {
      const bf16x8 kf = *reinterpret_cast<const bf16x8*>(ks + (kj * 16 + l15) * QPITCH + g * 8);
      const f32x4 z = {0.f, 0.f, 0.f, 0.f};
      sc[kj] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(qf, kf, z, 0, 0, 0);
    }
    // the logits leave the accumulator file here, in one go: no v_accvgpr_read among the LDS traffic of the softmax below
    // (see the note at the delta reduction of attn_bwd_mfma_kernel; tools/isa_lint.py)
    asm volatile("" : "+v"(sc[0]), "+v"(sc[1]), "+v"(sc[2]), "+v"(sc[3]));
    float rsum[4], rmax[4];
#pragma unroll
    for (int reg = 0; reg < 4; ++reg) {
      const int i = qi * 16 + 4 * g + reg;
      float mx = -INFINITY;
#pragma unroll
      for (int kj = 0; kj < 4; ++kj) {
        const float v = fmaf(sc[kj][reg], scale, addt[reg][kj]);
        sc[kj][reg] = v;
        mx = fmaxf(mx, v);
      }
      mx = row16_max(mx);
      float sum = 0.f;
#pragma unroll
      for (int kj = 0; kj < 4; ++kj) {
        const float p = (i < N) ? __builtin_amdgcn_exp2f(sc[kj][reg] - mx) : 0.f;
        sc[kj][reg] = p;
        sum += p;
      }
      sum = row16_sum(sum);
      rsum[reg] = sum;
      rmax[reg] = mx;
      // normalised probabilities (v_rcp_f32 once per query row; what the backward recomputes from the log-sum-exp)
      const float rinv = (i < N) ? __builtin_amdgcn_rcpf(sum) : 0.f;
#pragma unroll
      for (int kj = 0; kj < 4; ++kj) sc[kj][reg] *= rinv;
    }
    // P^T [key][query]: this lane holds queries qi*16 + 4g .. +3 of key kj*16 + l15 -> one 8-byte store per key tile (sixteen
    // 2-byte stores in the [query][key] layout of round 2)
#pragma unroll
    for (int kj = 0; kj < 4; ++kj) {
      bf16x4 p4;
#pragma unroll
      for (int reg = 0; reg < 4; ++reg) p4[reg] = f2bf(sc[kj][reg]);
      *reinterpret_cast<bf16x4*>(pt + (kj * 16 + l15) * PPITCH + qi * 16 + 4 * g) = p4;
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_wave_barrier();      // the P^T columns of this query tile are produced and consumed by the same wave
    // ---- O^T[dim][query] = V^T P^T: both operands by transpose reads from the row-major tiles (k-slots = keys), and the
    // accumulator puts four consecutive dims of ONE query in a lane: 8-byte global stores, no staging of the output tile ----
    {
      bf16x8 pb[2];
#pragma unroll
      for (int kt = 0; kt < 2; ++kt) pb[kt] = attn_tr_frag(pt, PPITCH, kt, qi * 16, g, l15);
      const int t = qi * 16 + l15;
      bf16* orow = out + token_row(a, b, win, t < N ? t : 0) * a.C + h * HD + 4 * g;
#pragma unroll
      for (int dj = 0; dj < 2; ++dj) {
        f32x4 o = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int kt = 0; kt < 2; ++kt)
          o = __builtin_amdgcn_mfma_f32_16x16x32_bf16(attn_tr_frag(vs, QPITCH, kt, dj * 16, g, l15), pb[kt], o, 0, 0, 0);
        bf16x4 o4;
#pragma unroll
        for (int reg = 0; reg < 4; ++reg) o4[reg] = f2bf(o[reg]);
        if (t < N) *reinterpret_cast<bf16x4*>(orow + dj * 16) = o4;
      }
    }
#pragma unroll
    for (int reg = 0; reg < 4; ++reg) {
      const int i = qi * 16 + 4 * g + reg;
      if (i < N && l15 == 0) lse[unit * N + i] = (rmax[reg] + __builtin_amdgcn_logf(rsum[reg])) * LN2;   // natural-log units
    }
    __syncthreads();                     // before the next image overwrites q / k / v
  }
}

// (Occupancy: 228 registers = two workgroups per CU.  Forcing three or four with __launch_bounds__ spills: 3.85 vs 2.44 ms per
// SwinV2-T step in isolation, tools/ubench/attn_time.py; the forward at six instead of five: no change.)
// Backward: recomputes P from (Qn, Kn, lse); dP = dO V^T on the same accumulator layout, dS = P (dP - delta) in registers.
// Round 3: the three products whose reduction index is a token — dV^T = dO^T P, dKn^T = Qn^T dS, dQn^T = Kn^T dS^T — take
// their operands with ds_read_b64_tr_b16 from the ROW-MAJOR tiles (conv_wgrad.hip's recipe: k-slot (g, e) <-> token
// 32s + (e < 4 ? 4g + e : 16 + 4g + e - 4) for both operands): no transposed copies of Qn / Kn / dO (96 scalar LDS stores per
// image), P and dS staged once as [key][query] with 8-byte stores (the accumulator layout holds four consecutive queries of a
// key), and the TRANSPOSED results put four consecutive dims of one token in a lane: 8-byte global stores straight from the
// accumulators, F.normalize's backward with two cross-lane steps.  A workgroup walks `bpw` images of its (window, head) and
// sums d(logits) into ONE [N][N] scratch tile, so the d(bias) scratch is bpw times smaller.
constexpr int BW_ROWMAJ = 4 * 64 * QPITCH;            // qn, kn, v, dO (bf16 elements), row-major [token][QPITCH]
constexpr int BW_ST = 64 * PPITCH;                    // dS^T staging: [key][PPITCH queries]
constexpr int BW_PT = 64 * PPITCH;                    // P^T staging:  [key][PPITCH queries]
constexpr int MFMA_BWD_LDS = (BW_ROWMAJ + BW_ST + BW_PT) * 2 + (2 * 64 + 4) * 4;   // + qinv, kinv, 4 partial sums

__global__ __launch_bounds__(256) void attn_bwd_mfma_kernel(AttnArgs a, const bf16* __restrict__ qkv,
                                                            const bf16* __restrict__ dout,
                                                            const float* __restrict__ logit_scale,
                                                            const float* __restrict__ bias, const float* __restrict__ mask,
                                                            const float* __restrict__ lse, bf16* __restrict__ dqkv,
                                                            float* __restrict__ ds_scratch, float* __restrict__ dscale_part,
                                                            int bpw, int units) {
  extern __shared__ char smraw[];
  bf16* qs = reinterpret_cast<bf16*>(smraw);
  bf16* ks = qs + 64 * QPITCH;
  bf16* vs = ks + 64 * QPITCH;
  bf16* gs = vs + 64 * QPITCH;           // dO row-major
  bf16* dst = gs + 64 * QPITCH;          // dS^T * scale [key][query]
  bf16* pt = dst + BW_ST;                // P^T [key][query]
  float* qinv = reinterpret_cast<float*>(pt + BW_PT);
  float* kinv = qinv + 64;
  float* wsum = kinv + 64;
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6, l15 = lane & 15, g = lane >> 4;
  const int N = a.N;
  const int unit_id = attn_unit(units);
  if (unit_id >= units) return;
  const int h = unit_id % a.heads;
  const int win = (unit_id / a.heads) % a.nW;
  const int bg = unit_id / (a.heads * a.nW);
  const float raw_ls = a.plain ? 0.f : logit_scale[h];
  const float scale = a.plain ? 0.17677669529663687f : expf(fminf(raw_ls, 4.605170185988092f));
  const float* bh = bias ? bias + (size_t)h * N * N : nullptr;
  const float* mw = mask ? mask + (size_t)win * N * N : nullptr;
  float* dS = ds_scratch ? ds_scratch + (size_t)unit_id * N * N : nullptr;
  float dsc = 0.f;
  float dsa[4][4];         // d(logits) of this wave's query tile summed over the images the workgroup walks (-> d(bias))
#pragma unroll
  for (int reg = 0; reg < 4; ++reg)
#pragma unroll
    for (int kj = 0; kj < 4; ++kj) dsa[reg][kj] = 0.f;
  // position bias + shift mask of this wave's query tile (-inf on padding), loaded once, in log2 units (P = exp2(..)): straight-
  // line loads on clamped indices + selects — a branch per element puts an s_waitcnt (one L2 round trip) in front of each load
  constexpr float LOG2E = 1.4426950408889634f;
  float addt[4][4];
  {
    const bool any_add = bh != nullptr || mw != nullptr;
    const float bsel = bh ? 1.f : 0.f, msel = mw ? 1.f : 0.f;
    const float* bp = bh ? bh : mw;
    const float* mp = mw ? mw : bp;
#pragma unroll
    for (int reg = 0; reg < 4; ++reg)
#pragma unroll
      for (int kj = 0; kj < 4; ++kj) {
        const int i = wv * 16 + 4 * g + reg, j = kj * 16 + l15;
        const int idx = (i < N ? i : N - 1) * N + (j < N ? j : N - 1);
        float v = 0.f;
        if (any_add) v = (bsel * bp[idx] + msel * mp[idx]) * LOG2E;
        addt[reg][kj] = (i < N && j < N) ? v : -INFINITY;
      }
  }
  const float scale2 = scale * LOG2E;
  // this lane's row of the image being staged (wave 0: q, 1: k, 2: v, 3: dO) and the log-sum-exp of its four queries are
  // requested one image AHEAD: global latency (and the mid-kernel lse round trip) sit under the previous image's arithmetic
  const int64_t tok_sp = token_row(a, 0, win, lane < N ? lane : 0);        // row inside image 0; + b * H * W per image
  const int64_t img_rows = (int64_t)a.H * a.W;
  bf16x8 rnext[4];
  float lse_next[4];
  auto request = [&](int b) {
    const int64_t row = tok_sp + (int64_t)b * img_rows;
    const bf16* src = wv < 3 ? qkv + row * a.ld + h * HD + wv * a.C : dout + row * a.C + h * HD;
    load_head_raw(src, lane < N, rnext);
    const size_t u = ((size_t)b * a.nW + win) * a.heads + h;
#pragma unroll
    for (int reg = 0; reg < 4; ++reg) {
      const int i = wv * 16 + 4 * g + reg;
      lse_next[reg] = lse[u * N + (i < N ? i : 0)];
    }
  };
  if (bg * bpw < a.B) request(bg * bpw);
  for (int bb = 0; bb < bpw; ++bb) {
    const int b = bg * bpw + bb;
    if (b >= a.B) break;                 // uniform for the workgroup
    float lsev[4];
#pragma unroll
    for (int reg = 0; reg < 4; ++reg) lsev[reg] = lse_next[reg] * LOG2E;
    {   // row-major tiles only (q, k normalised, + 1/norm)
      const int t = lane;
      bf16x8 r8[4];
#pragma unroll
      for (int d = 0; d < 4; ++d) r8[d] = rnext[d];
      const float inv = finish_head_row(wv < 2 && !a.plain, r8);
      if (wv == 0) { put_row_major(qs, t, r8); qinv[t] = inv; }
      else if (wv == 1) { put_row_major(ks, t, r8); kinv[t] = inv; }
      else if (wv == 2) put_row_major(vs, t, r8);
      else put_row_major(gs, t, r8);
    }
    if (bb + 1 < bpw && b + 1 < a.B) request(b + 1);
    __syncthreads();
    const int qi = wv;
    const bf16x8 qf = *reinterpret_cast<const bf16x8*>(qs + (qi * 16 + l15) * QPITCH + g * 8);
    const bf16x8 gf = *reinterpret_cast<const bf16x8*>(gs + (qi * 16 + l15) * QPITCH + g * 8);
    f32x4 sc[4], dp[4];
#pragma unroll
    for (int kj = 0; kj < 4; ++kj) {
      const bf16x8 kf = *reinterpret_cast<const bf16x8*>(ks + (kj * 16 + l15) * QPITCH + g * 8);
      const bf16x8 vf = *reinterpret_cast<const bf16x8*>(vs + (kj * 16 + l15) * QPITCH + g * 8);
      const f32x4 z = {0.f, 0.f, 0.f, 0.f};
      sc[kj] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(qf, kf, z, 0, 0, 0);
      dp[kj] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(gf, vf, z, 0, 0, 0);
    }
    // logits and dP leave the accumulator file in one go (no v_accvgpr_read among the ds_bpermute / ds_write below)
    asm volatile("" : "+v"(sc[0]), "+v"(sc[1]), "+v"(sc[2]), "+v"(sc[3]), "+v"(dp[0]), "+v"(dp[1]), "+v"(dp[2]), "+v"(dp[3]));
    // ---- P, dP, dS on the accumulator layout: query i = qi*16 + 4g + reg, key j = kj*16 + l15 ----
    float pv[4][4], dsv[4][4];           // [kj][reg]
#pragma unroll
    for (int reg = 0; reg < 4; ++reg) {
      const int i = qi * 16 + 4 * g + reg;
      const float li = lsev[reg];
      float dl = 0.f;
#pragma unroll
      for (int kj = 0; kj < 4; ++kj) {
        // (padding: addt is -inf there and the staged rows / the clamped log-sum-exp are finite, so exp2 gives exactly 0)
        pv[kj][reg] = __builtin_amdgcn_exp2f(fmaf(sc[kj][reg], scale2, addt[reg][kj]) - li);
        dl = fmaf(pv[kj][reg], dp[kj][reg], dl);
      }
      dl = row16_sum(dl);
      // Value barrier (kept with the DPP reduction; it was found with the __shfl_xor butterfly that stood here): the reduced
      // delta is materialised in a register of its own before its consumers.  Without it hipcc (ROCm 7.2) paired the
      // butterfly's last steps with the d(logits) arithmetic and overwrote the address register of two ds_bpermute in flight
      // with a v_accvgpr_read of the next accumulator ("ds_bpermute v152, v20, v144; ds_bpermute v153, v20, v145;
      // v_accvgpr_read_b32 v20, a6"): under load the last quarter-wave (lanes 48-63) of the second permute then read a stale
      // index and rows 12..15 of a query tile got delta = 0 for one key tile — a few hundred wrong d(q) / d(k) elements per
      // launch, different ones every run (found by tests/test_fullsize_properties_gpu.py's bit-reproducibility check;
      // tests/test_kernels_gpu.py::test_window_attention_is_bit_reproducible pins it at the kernel level).
      asm volatile("" : "+v"(dl));
#pragma unroll
      for (int kj = 0; kj < 4; ++kj) {
        const float ds = pv[kj][reg] * (dp[kj][reg] - dl);
        dsa[reg][kj] += ds;
        dsc = fmaf(ds, sc[kj][reg], dsc);
        dsv[kj][reg] = ds * scale;       // d(qn kn^T) = d(logits) * scale
      }
    }
    // [key][query] staging: this lane holds queries qi*16 + 4g .. +3 of key kj*16 + l15 -> one 8-byte store per matrix and kj
#pragma unroll
    for (int kj = 0; kj < 4; ++kj) {
      bf16x4 p4, d4;
#pragma unroll
      for (int reg = 0; reg < 4; ++reg) { p4[reg] = f2bf(pv[kj][reg]); d4[reg] = f2bf(dsv[kj][reg]); }
      const int off = (kj * 16 + l15) * PPITCH + qi * 16 + 4 * g;
      *reinterpret_cast<bf16x4*>(pt + off) = p4;
      *reinterpret_cast<bf16x4*>(dst + off) = d4;
    }
    __syncthreads();
    // ---- wave mt: dV^T, dKn^T of key tile mt (= dO^T P, Qn^T dS) and dQn^T of query tile mt (= Kn^T dS^T) ----
    {
      const int mt = wv;
      const int t = mt * 16 + l15;        // the token (key for dv / dk, query for dq) of this lane's accumulator column
      f32x4 dv[2], dk[2], dq[2];
#pragma unroll
      for (int dj = 0; dj < 2; ++dj) dv[dj] = dk[dj] = dq[dj] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int kk = 0; kk < 2; ++kk) {
        const bf16x8 pb = attn_row_frag(pt, PPITCH, t, kk, g);                    // P[q][key t], q in k-slot order
        const bf16x8 db = attn_row_frag(dst, PPITCH, t, kk, g);                   // dS[q][key t] * scale
        const bf16x8 dtb = attn_tr_frag(dst, PPITCH, kk, mt * 16, g, l15);        // dS^T[key][query t] * scale, keys in k-slot order
#pragma unroll
        for (int dj = 0; dj < 2; ++dj) {
          const bf16x8 gta = attn_tr_frag(gs, QPITCH, kk, dj * 16, g, l15);       // dO^T[dim][q]
          const bf16x8 qta = attn_tr_frag(qs, QPITCH, kk, dj * 16, g, l15);       // Qn^T[dim][q]
          const bf16x8 kta = attn_tr_frag(ks, QPITCH, kk, dj * 16, g, l15);       // Kn^T[dim][key]
          dv[dj] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(gta, pb, dv[dj], 0, 0, 0);
          dk[dj] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(qta, db, dk[dj], 0, 0, 0);
          dq[dj] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(kta, dtb, dq[dj], 0, 0, 0);
        }
      }
      // accumulator element (dj, reg): dim dj*16 + 4g + reg of token t.  F.normalize backward: d = (dn - n <n, dn>) / |x|
      float qn[2][4], kn[2][4], dotq = 0.f, dotk = 0.f;
#pragma unroll
      for (int dj = 0; dj < 2; ++dj) {
        const bf16x4 q4 = *reinterpret_cast<const bf16x4*>(qs + t * QPITCH + dj * 16 + 4 * g);
        const bf16x4 k4 = *reinterpret_cast<const bf16x4*>(ks + t * QPITCH + dj * 16 + 4 * g);
#pragma unroll
        for (int reg = 0; reg < 4; ++reg) {
          qn[dj][reg] = bf2f(q4[reg]);
          kn[dj][reg] = bf2f(k4[reg]);
          dotq = fmaf(qn[dj][reg], dq[dj][reg], dotq);
          dotk = fmaf(kn[dj][reg], dk[dj][reg], dotk);
        }
      }
      dotq += __shfl_xor(dotq, 16, 64); dotq += __shfl_xor(dotq, 32, 64);
      dotk += __shfl_xor(dotk, 16, 64); dotk += __shfl_xor(dotk, 32, 64);
      if (t < N) {
        const float qi_ = qinv[t], ki_ = kinv[t];
        bf16* dr = dqkv + token_row(a, b, win, t) * a.ld + h * HD + 4 * g;
#pragma unroll
        for (int dj = 0; dj < 2; ++dj) {
          bf16x4 oq, ok, ov;
#pragma unroll
          for (int reg = 0; reg < 4; ++reg) {
            oq[reg] = f2bf(a.plain ? dq[dj][reg] : (dq[dj][reg] - qn[dj][reg] * dotq) * qi_);
            ok[reg] = f2bf(a.plain ? dk[dj][reg] : (dk[dj][reg] - kn[dj][reg] * dotk) * ki_);
            ov[reg] = f2bf(dv[dj][reg]);
          }
          *reinterpret_cast<bf16x4*>(dr + dj * 16) = oq;
          *reinterpret_cast<bf16x4*>(dr + a.C + dj * 16) = ok;
          *reinterpret_cast<bf16x4*>(dr + 2 * a.C + dj * 16) = ov;
        }
      }
    }
    __syncthreads();                     // before the next image overwrites the tiles / the staging
  }
  if (dS != nullptr) {
#pragma unroll
    for (int reg = 0; reg < 4; ++reg)
#pragma unroll
      for (int kj = 0; kj < 4; ++kj) {
        const int i = wv * 16 + 4 * g + reg, j = kj * 16 + l15;
        if (i < N && j < N) dS[(size_t)i * N + j] = dsa[reg][kj];
      }
  }
  dsc = wave_sum(dsc);
  if (lane == 0) wsum[wv] = dsc;
  __syncthreads();
  if (threadIdx.x == 0 && dscale_part != nullptr)
    dscale_part[unit_id] = raw_ls < 4.605170185988092f ? (wsum[0] + wsum[1] + wsum[2] + wsum[3]) * scale : 0.f;
}

// backward, phase A (lane = query i): dS row -> scratch dS[(b,w)][h][i][j] (fp32), dq; partial dscale
// phase B (lane = key j): dv_j, dk_j from the columns of P and dS
__global__ __launch_bounds__(64) void attn_bwd_kernel(AttnArgs a, const bf16* __restrict__ qkv,
                                                      const bf16* __restrict__ dout, const float* __restrict__ logit_scale,
                                                      const float* __restrict__ bias, const float* __restrict__ mask,
                                                      const float* __restrict__ lse, bf16* __restrict__ dqkv,
                                                      float* __restrict__ dST, float* __restrict__ dscale_part) {
  extern __shared__ float sm[];
  const int N = a.N;
  float* qn = sm;
  float* kn = qn + N * HD;
  float* v = kn + N * HD;
  float* dO = v + N * HD;
  float* qinv = dO + N * HD;
  float* kinv = qinv + N;
  float* delta = kinv + N;
  float* lrow = delta + N;
  const int h = blockIdx.x % a.heads;
  const int win = (blockIdx.x / a.heads) % a.nW;
  const int b = blockIdx.x / (a.heads * a.nW);
  load_qkv(a, qkv, b, win, h, qn, kn, v, qinv, kinv);
  const float raw = logit_scale[h];
  const float scale = expf(fminf(raw, 4.605170185988092f));
  const float* bh = bias + (size_t)h * N * N;
  const float* mw = mask ? mask + (size_t)win * N * N : nullptr;
  float* dS = dST + (size_t)blockIdx.x * N * N;
  for (int t = threadIdx.x; t < N; t += 64) {
    const bf16* g = dout + token_row(a, b, win, t) * a.C + h * HD;
#pragma unroll
    for (int d = 0; d < HD; d += 8) {
      const bf16x8 g8 = ldg16(g + d);
#pragma unroll
      for (int e = 0; e < 8; ++e) dO[t * HD + d + e] = bf2f(g8[e]);
    }
    lrow[t] = lse[(size_t)blockIdx.x * N + t];
  }
  __syncthreads();
  float dsc = 0.f;
  // ---- phase A ----
  for (int i = threadIdx.x; i < N; i += 64) {
    float q[HD], go[HD], dq[HD];
#pragma unroll
    for (int d = 0; d < HD; ++d) { q[d] = qn[i * HD + d]; go[d] = dO[i * HD + d]; dq[d] = 0.f; }
    const float li = lrow[i];
    // delta_i = sum_j p_ij dP_ij  (= dO_i . O_i)
    float dl = 0.f;
    for (int j = 0; j < N; ++j) {
      float s = 0.f, dp = 0.f;
#pragma unroll
      for (int d = 0; d < HD; ++d) { s = fmaf(q[d], kn[j * HD + d], s); dp = fmaf(go[d], v[j * HD + d], dp); }
      const float p = expf(s * scale + bh[i * N + j] + (mw ? mw[i * N + j] : 0.f) - li);
      dl = fmaf(p, dp, dl);
    }
    delta[i] = dl;
    for (int j = 0; j < N; ++j) {
      float s = 0.f, dp = 0.f;
#pragma unroll
      for (int d = 0; d < HD; ++d) { s = fmaf(q[d], kn[j * HD + d], s); dp = fmaf(go[d], v[j * HD + d], dp); }
      const float p = expf(s * scale + bh[i * N + j] + (mw ? mw[i * N + j] : 0.f) - li);
      const float ds = p * (dp - dl);
      dS[(size_t)i * N + j] = ds;
      dsc = fmaf(ds, s, dsc);
#pragma unroll
      for (int d = 0; d < HD; ++d) dq[d] = fmaf(ds * scale, kn[j * HD + d], dq[d]);
    }
    // through F.normalize: dq_raw = (dqn - qn (qn . dqn)) / |q|
    float dot = 0.f;
#pragma unroll
    for (int d = 0; d < HD; ++d) dot = fmaf(q[d], dq[d], dot);
    bf16* dr = dqkv + token_row(a, b, win, i) * a.ld + h * HD;
#pragma unroll
    for (int d = 0; d < HD; d += 8) {
      bf16x8 o8;
#pragma unroll
      for (int e = 0; e < 8; ++e) o8[e] = f2bf((dq[d + e] - q[d + e] * dot) * qinv[i]);
      stg16(dr + d, o8);
    }
  }
  dsc = wave_sum(dsc);
  // d logit_scale = d scale * scale (zero where the clamp is active)
  if (threadIdx.x == 0) dscale_part[blockIdx.x] = raw < 4.605170185988092f ? dsc * scale : 0.f;
  __syncthreads();
  // ---- phase B ----
  for (int j = threadIdx.x; j < N; j += 64) {
    float k[HD], dk[HD], dv[HD];
#pragma unroll
    for (int d = 0; d < HD; ++d) { k[d] = kn[j * HD + d]; dk[d] = 0.f; dv[d] = 0.f; }
    for (int i = 0; i < N; ++i) {
      float s = 0.f;
#pragma unroll
      for (int d = 0; d < HD; ++d) s = fmaf(qn[i * HD + d], k[d], s);
      const float p = expf(s * scale + bh[i * N + j] + (mw ? mw[i * N + j] : 0.f) - lrow[i]);
      const float ds = dS[(size_t)i * N + j] * scale;
#pragma unroll
      for (int d = 0; d < HD; ++d) {
        dv[d] = fmaf(p, dO[i * HD + d], dv[d]);
        dk[d] = fmaf(ds, qn[i * HD + d], dk[d]);
      }
    }
    float dot = 0.f;
#pragma unroll
    for (int d = 0; d < HD; ++d) dot = fmaf(k[d], dk[d], dot);
    bf16* dr = dqkv + token_row(a, b, win, j) * a.ld + h * HD;
#pragma unroll
    for (int d = 0; d < HD; d += 8) {
      bf16x8 k8, v8;
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        k8[e] = f2bf((dk[d + e] - k[d + e] * dot) * kinv[j]);
        v8[e] = f2bf(dv[d + e]);
      }
      stg16(dr + a.C + d, k8);
      stg16(dr + 2 * a.C + d, v8);
    }
  }
}

// ---------------------------------------------------------------------------------------------------------------
// continuous position bias: bias[h][i][j] = 16 * sigmoid(table[index[i][j]][h])
__global__ __launch_bounds__(256) void cpb_bias_fwd_kernel(const bf16* __restrict__ table, int ld,
                                                           const int64_t* __restrict__ index, int heads, int nn,
                                                           float* __restrict__ bias) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= heads * nn) return;
  const int h = i / nn, ij = i - h * nn;
  const float t = bf2f(table[index[ij] * ld + h]);
  bias[i] = 16.f / (1.f + expf(-t));
}

// dtable[t][h] = sum_{(i,j): index == t} dbias[h][i][j] * 16 s (1 - s)     (gather over the index: exact, ordered)
__global__ __launch_bounds__(64) void cpb_bias_bwd_kernel(const float* __restrict__ dbias, const bf16* __restrict__ table,
                                                          int ld, const int64_t* __restrict__ index, int heads, int n,
                                                          int transposed, bf16* __restrict__ dtable) {
  const int t = blockIdx.x;
  const int h = blockIdx.y;
  const int nn = n * n;
  if (h >= heads) {   // padding columns of the cpb_mlp output row: defined zeros for the GEMMs downstream
    if (threadIdx.x == 0) dtable[(size_t)t * ld + h] = f2bf(0.f);
    return;
  }
  // eight positions per lane and iteration, every load unconditional (select afterwards): the branchy one-at-a-time form was a
  // chain of 38 dependent L2 round trips per block (68 us per launch on the SwinV2-T step's main queue); the summation order per
  // lane is unchanged (ascending p), so the result keeps its bits
  float acc = 0.f;
  for (int p0 = threadIdx.x; p0 < nn; p0 += 64 * 8) {
    int64_t idx[8];
    float dv[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      const int p = p0 + 64 * u;
      const int pc = p < nn ? p : nn - 1;
      // transposed: dbias is stored [h][j][i] (what tok_window_attn_bwd's scratch reduces to)
      const int ij = transposed ? (pc % n) * n + pc / n : pc;
      idx[u] = index[ij];
      dv[u] = dbias[(size_t)h * nn + pc];
    }
#pragma unroll
    for (int u = 0; u < 8; ++u) acc += (p0 + 64 * u < nn && idx[u] == t) ? dv[u] : 0.f;
  }
  acc = wave_sum(acc);
  if (threadIdx.x == 0) {
    const float s = 1.f / (1.f + expf(-bf2f(table[(size_t)t * ld + h])));
    dtable[(size_t)t * ld + h] = f2bf(acc * 16.f * s * (1.f - s));
  }
}

// ---------------------------------------------------------------------------------------------------------------
// PatchMerging gather: out[b][y][x][q*C + c] = in[b][2y + (q & 1)][2x + (q >> 1)][c]   (x0, x1, x2, x3 order)
__global__ __launch_bounds__(256) void patch_merge_kernel(const bf16* __restrict__ src, bf16* __restrict__ dst, int B,
                                                          int H, int W, int C, int inverse) {
  const int cg = C >> 3, H2 = H >> 1, W2 = W >> 1;
  const size_t total = (size_t)B * H2 * W2 * 4 * cg;
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (size_t)gridDim.x * 256) {
    const int g = (int)(i % cg);
    size_t r = i / cg;
    const int q = (int)(r & 3);
    r >>= 2;
    const int x = (int)(r % W2);
    r /= W2;
    const int y = (int)(r % H2);
    const int b = (int)(r / H2);
    const size_t merged = ((((size_t)b * H2 + y) * W2 + x) * 4 + q) * C + g * 8;
    const size_t plain = (((size_t)b * H + 2 * y + (q & 1)) * W + 2 * x + (q >> 1)) * C + g * 8;
    if (inverse) stg16(dst + plain, ldg16(src + merged));
    else stg16(dst + merged, ldg16(src + plain));
  }
}

inline int blocks_for(size_t total) {
  const size_t b = (total + 255) / 256;
  return (int)(b > 65536 ? 65536 : (b < 1 ? 1 : b));
}

// TOK_ATTN_SCALAR=1 forces the VALU reference kernels (debugging aid)
inline bool tok_attn_scalar() {
  static const int v = [] { const char* e = getenv("TOK_ATTN_SCALAR"); return (int)((e && e[0] == '1') ? 1 : 0); }();
  return v == 1;
}

bool fill_attn(AttnArgs& a, int B, int H, int W, int C, int heads, int ws, int shift, int ld) {
  if (B <= 0 || H <= 0 || W <= 0 || heads <= 0 || ws <= 0 || C != heads * HD || H % ws || W % ws || shift < 0 ||
      shift >= ws || ld < 3 * C || (ld & 7)) return false;
  a.B = B; a.H = H; a.W = W; a.C = C; a.heads = heads; a.ws = ws; a.shift = shift;
  a.nWx = W / ws; a.nW = (H / ws) * a.nWx; a.N = ws * ws; a.ld = ld; a.plain = 0;
  return true;
}

}  // namespace

extern "C" int tok_layernorm_fwd(const void* x, const void* shortcut, const float* row_scale, int rows_per_sample,
                                 const float* gamma, const float* beta, void* out, float* mean, float* rstd,
                                 int64_t rows, int c, int ld, float eps, void* stream) {
  TOK_CHECK_ARG(x && gamma && beta && out && mean && rstd && rows > 0 && c > 0 && ld >= c, "tok_layernorm_fwd: bad args");
  TOK_CHECK_ARG(!row_scale || rows_per_sample > 0, "tok_layernorm_fwd: rows_per_sample");
  hipStream_t st = tok_stream(stream);
  if (c == ld && c % 8 == 0 && c <= 1024) {
#define TOK_LN_FWD(LPR, VPL)                                                                                          \
  hipLaunchKernelGGL((ln_fwd_vec_kernel<LPR, VPL>), dim3((unsigned)tok_cdiv(rows, 4 * (64 / LPR))), dim3(256), 0, st, \
                     (const bf16*)x, (const bf16*)shortcut, row_scale, rows_per_sample, gamma, beta, (bf16*)out, mean,  \
                     rstd, rows, c, eps)
    const int cg = c >> 3;
    if (cg <= 16) TOK_LN_FWD(16, 1);
    else if (cg <= 32) TOK_LN_FWD(32, 1);
    else if (cg <= 64) TOK_LN_FWD(64, 1);
    else TOK_LN_FWD(64, 2);
#undef TOK_LN_FWD
  } else {
    hipLaunchKernelGGL(ln_fwd_kernel, dim3((unsigned)((rows + 3) / 4)), dim3(256), 0, st, (const bf16*)x,
                       (const bf16*)shortcut, row_scale, rows_per_sample, gamma, beta, (bf16*)out, mean, rstd, rows, c, ld,
                       eps);
  }
  TOK_CHECK_LAUNCH("tok_layernorm_fwd");
  return TOK_OK;
}

extern "C" int tok_layernorm_bwd_rows(int64_t rows, int c) {
  const int64_t b = (rows + 3) / 4;
  (void)c;
  // four blocks per CU = what is resident at 102 registers; the partial d(gamma) / d(beta) rows the fold reads scale with the
  // grid (SwinV2-T B=256, ms/step: 512 20.42, 1024 20.37, 2048 20.43, 4096 20.59, 8192 20.97)
  static const int cap = [] { const char* e = getenv("TOK_LN_BWD_BLOCKS"); const int v = e ? atoi(e) : 1024; return v < 1 ? 1 : v; }();
  return (int)(b > cap ? cap : (b < 1 ? 1 : b));
}

extern "C" int tok_layernorm_bwd(const void* dout, const void* x, const float* mean, const float* rstd,
                                 const float* gamma, const float* row_scale, int rows_per_sample, void* dx,
                                 int accumulate, float* partial, int64_t rows, int c, int ld, void* stream) {
  TOK_CHECK_ARG(dout && x && mean && rstd && gamma && dx && partial && rows > 0 && c > 0 && ld >= c,
                "tok_layernorm_bwd: bad args");
  const int g = tok_layernorm_bwd_rows(rows, c);
  hipStream_t st = tok_stream(stream);
  if (c == ld && c % 8 == 0 && c <= 1024) {
#define TOK_LN_BWD(LPR, VPL)                                                                                              \
  do {                                                                                                                    \
    const size_t smem = (size_t)(4 * (64 / LPR)) * 2 * c * sizeof(float);                                                 \
    hipLaunchKernelGGL((ln_bwd_vec_kernel<LPR, VPL>), dim3(g), dim3(256), smem, st, (const bf16*)dout, (const bf16*)x,    \
                       mean, rstd, gamma, row_scale, rows_per_sample, (bf16*)dx, accumulate, partial, rows, c);           \
  } while (0)
    const int cg = c >> 3;
    if (cg <= 16) TOK_LN_BWD(16, 1);
    else if (cg <= 32) TOK_LN_BWD(32, 1);
    else if (cg <= 64) TOK_LN_BWD(64, 1);
    else TOK_LN_BWD(64, 2);
#undef TOK_LN_BWD
  } else {
    TOK_CHECK_ARG((size_t)c * 8 * sizeof(float) <= 64 * 1024, "tok_layernorm_bwd: c too large (%d)", c);
    hipLaunchKernelGGL(ln_bwd_kernel, dim3(g), dim3(256), (size_t)c * 8 * sizeof(float), st, (const bf16*)dout,
                       (const bf16*)x, mean, rstd, gamma, row_scale, rows_per_sample, (bf16*)dx, accumulate, partial,
                       rows, c, ld);
  }
  TOK_CHECK_LAUNCH("tok_layernorm_bwd");
  return TOK_OK;
}

extern "C" int tok_colsum_f32(const float* src, int64_t rows, int cols, float* dst, int accumulate, void* stream) {
  TOK_CHECK_ARG(src && dst && rows > 0 && cols > 0, "tok_colsum_f32: bad args");
  if (tok_dbg_skip(4)) return TOK_OK;
  static const int wide = [] { const char* e = getenv("TOK_COLSUM_WIDE"); return e ? atoi(e) : 1; }();   // 0: the narrow kernel everywhere (A/B switch)
  if (wide && cols >= 2048) {
    hipLaunchKernelGGL(colsum_f32_wide_kernel, dim3((cols + 255) / 256), dim3(1024), 0, tok_stream(stream), src, rows, cols, dst,
                       accumulate);
    TOK_CHECK_LAUNCH("tok_colsum_f32(wide)");
    return TOK_OK;
  }
  hipLaunchKernelGGL(colsum_f32_kernel, dim3((cols + 15) / 16), dim3(256), 0, tok_stream(stream), src, rows, cols, dst,
                     accumulate, (const float*)nullptr, (float*)nullptr, 0);
  TOK_CHECK_LAUNCH("tok_colsum_f32");
  return TOK_OK;
}

extern "C" int tok_colsum_f32_pair(const float* src0, const float* src1, int64_t rows, int cols, float* dst0, int accumulate0,
                                   float* dst1, int accumulate1, void* stream) {
  TOK_CHECK_ARG(src0 && src1 && dst0 && dst1 && rows > 0 && cols > 0, "tok_colsum_f32_pair: bad args");
  hipLaunchKernelGGL(colsum_f32_kernel, dim3((cols + 15) / 16, 2), dim3(256), 0, tok_stream(stream), src0, rows, cols, dst0,
                     accumulate0, src1, dst1, accumulate1);
  TOK_CHECK_LAUNCH("tok_colsum_f32_pair");
  return TOK_OK;
}

extern "C" int tok_act_fwd(int kind, const void* x, void* out, size_t count, void* stream) {
  TOK_CHECK_ARG(x && out && count > 0 && count % 8 == 0 && (kind == 0 || kind == 1), "tok_act_fwd: bad args");
  hipLaunchKernelGGL(act_fwd_kernel, dim3(blocks_for(count / 8)), dim3(256), 0, tok_stream(stream), kind, (const bf16*)x,
                     (bf16*)out, count / 8);
  TOK_CHECK_LAUNCH("tok_act_fwd");
  return TOK_OK;
}

extern "C" int tok_act_bwd(int kind, const void* dout, const void* x, void* dx, int accumulate, size_t count,
                           void* stream) {
  TOK_CHECK_ARG(dout && x && dx && count > 0 && count % 8 == 0 && kind >= 0 && kind <= 2, "tok_act_bwd: bad args");
  hipLaunchKernelGGL(act_bwd_kernel, dim3(blocks_for(count / 8)), dim3(256), 0, tok_stream(stream), kind,
                     (const bf16*)dout, (const bf16*)x, (bf16*)dx, accumulate, count / 8);
  TOK_CHECK_LAUNCH("tok_act_bwd");
  return TOK_OK;
}

namespace {
int attn_bpw(const AttnArgs& a) {     // images per wave on the MFMA path
  const long long units = (long long)a.B * a.nW * a.heads;
  // images a workgroup walks: about 1536 workgroups per launch (three rounds of the backward's 512 resident ones), at most 16
  // images each — the per-workgroup prologue (sixteen bias / mask loads per lane) and the first image's exposed load are paid
  // once per workgroup.  Measured per SwinV2-T step in isolation (tools/ubench/attn_time.py): units / 4096 capped at 8 (rounds
  // 2-3) 1.00 / 2.36 ms forward / backward, / 1536 capped at 16: 0.95 / 1.99 ms; / 1024 cap 8: 0.96 / 2.06; / 8192: 1.15 / 2.84.
  // (parsed once, clamped to >= 1: tok_window_attn_bwd_rows and the launch must agree on this number for the life of the process)
  static const int div = [] { const char* e = getenv("TOK_ATTN_BPW_DIV"); const int v = e ? atoi(e) : 1536; return v < 1 ? 1 : v; }();
  static const int cap = [] { const char* e = getenv("TOK_ATTN_BPW_CAP"); const int v = e ? atoi(e) : 16; return v < 1 ? 1 : v; }();
  long long bpw = units / div;
  bpw = bpw < 1 ? 1 : (bpw > cap ? cap : bpw);
  return (int)(bpw > a.B ? a.B : bpw);
}
}  // namespace

extern "C" int tok_window_attn_fwd(const void* qkv, int batch, int h, int w, int c, int heads, int ws, int shift, int ld,
                                   const float* logit_scale, const float* bias, const float* mask, void* out,
                                   float* lse, void* stream) {
  AttnArgs a;
  TOK_CHECK_ARG(qkv && out && lse && fill_attn(a, batch, h, w, c, heads, ws, shift, ld),
                "tok_window_attn_fwd: bad args (head_dim must be 32, h/w multiples of the window)");
  TOK_CHECK_ARG((logit_scale == nullptr) == (bias == nullptr), "tok_window_attn_fwd: logit_scale and bias go together");
  a.plain = logit_scale == nullptr;
  TOK_CHECK_ARG(!a.plain || (a.N <= 64 && !mask && shift == 0),
                "tok_window_attn_fwd: the plain mode covers unshifted windows of up to 64 tokens");
  if (a.N <= 64 && (a.plain || !tok_attn_scalar())) {
    const int bpw = attn_bpw(a);
    const int groups = tok_cdiv(batch, bpw) * a.nW * heads;
    hipLaunchKernelGGL(attn_fwd_mfma_kernel, dim3(8 * tok_cdiv(groups, 8)), dim3(256), MFMA_FWD_LDS, tok_stream(stream), a,
                       (const bf16*)qkv, logit_scale, bias, mask, (bf16*)out, lse, bpw, groups);
    TOK_CHECK_LAUNCH("tok_window_attn_fwd(mfma)");
    return TOK_OK;
  }
  const size_t smem = (size_t)a.N * HD * 3 * sizeof(float);
  TOK_CHECK_ARG(smem <= 160 * 1024, "tok_window_attn_fwd: window %d too large", ws);
  static const bool attr = [&] {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&attn_fwd_kernel),
                                         hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    return true;
  }();   // once per process (thread-safe function-local static)
  (void)attr;
  hipLaunchKernelGGL(attn_fwd_kernel, dim3(batch * a.nW * heads), dim3(64), smem, tok_stream(stream), a,
                     (const bf16*)qkv, logit_scale, bias, mask, (bf16*)out, lse);
  TOK_CHECK_LAUNCH("tok_window_attn_fwd");
  return TOK_OK;
}


extern "C" int tok_window_attn_bwd_rows(int batch, int h, int w, int heads, int ws) {
  AttnArgs a;
  if (!fill_attn(a, batch, h, w, heads * HD, heads, ws, 0, 3 * heads * HD)) return TOK_ERR_INVALID;
  if (a.N <= 64 && !tok_attn_scalar()) return tok_cdiv(batch, attn_bpw(a)) * a.nW;
  return batch * a.nW;
}

extern "C" int tok_window_attn_bwd(const void* qkv, const void* dout, int batch, int h, int w, int c, int heads, int ws,
                                   int shift, int ld, const float* logit_scale, const float* bias, const float* mask,
                                   const float* lse, void* dqkv, float* ds_scratch, float* dscale_part, void* stream) {
  AttnArgs a;
  TOK_CHECK_ARG(qkv && dout && lse && dqkv && fill_attn(a, batch, h, w, c, heads, ws, shift, ld),
                "tok_window_attn_bwd: bad args");
  a.plain = logit_scale == nullptr;
  TOK_CHECK_ARG(a.plain ? (!bias && !mask && shift == 0 && a.N <= 64) : (bias && ds_scratch && dscale_part),
                "tok_window_attn_bwd: bad args (plain mode: no bias / mask / shift, windows of up to 64 tokens)");
  if (a.N <= 64 && (a.plain || !tok_attn_scalar())) {
    const int bpw = attn_bpw(a);
    const int waves = tok_cdiv(batch, bpw) * a.nW * heads;
    static const bool attr_m = [&] {
      (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&attn_bwd_mfma_kernel),
                                             hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
      return true;
    }();   // once per process (thread-safe function-local static)
    (void)attr_m;
    hipLaunchKernelGGL(attn_bwd_mfma_kernel, dim3(8 * tok_cdiv(waves, 8)), dim3(256), MFMA_BWD_LDS, tok_stream(stream), a,
                       (const bf16*)qkv, (const bf16*)dout, logit_scale, bias, mask, lse, (bf16*)dqkv, ds_scratch, dscale_part, bpw,
                       waves);
    TOK_CHECK_LAUNCH("tok_window_attn_bwd(mfma)");
    return TOK_OK;
  }
  const size_t smem = ((size_t)a.N * HD * 4 + (size_t)a.N * 4) * sizeof(float);
  TOK_CHECK_ARG(smem <= 160 * 1024, "tok_window_attn_bwd: window %d too large", ws);
  static const bool attr = [&] {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&attn_bwd_kernel),
                                         hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    return true;
  }();   // once per process (thread-safe function-local static)
  (void)attr;
  hipLaunchKernelGGL(attn_bwd_kernel, dim3(batch * a.nW * heads), dim3(64), smem, tok_stream(stream), a,
                     (const bf16*)qkv, (const bf16*)dout, logit_scale, bias, mask, lse, (bf16*)dqkv, ds_scratch,
                     dscale_part);
  TOK_CHECK_LAUNCH("tok_window_attn_bwd");
  return TOK_OK;
}

extern "C" int tok_cpb_bias_fwd(const void* table, int ld, const int64_t* index, int heads, int n_tokens, float* bias,
                                void* stream) {
  TOK_CHECK_ARG(table && index && bias && heads > 0 && n_tokens > 0 && ld >= heads, "tok_cpb_bias_fwd: bad args");
  const int nn = n_tokens * n_tokens;
  hipLaunchKernelGGL(cpb_bias_fwd_kernel, dim3((heads * nn + 255) / 256), dim3(256), 0, tok_stream(stream),
                     (const bf16*)table, ld, index, heads, nn, bias);
  TOK_CHECK_LAUNCH("tok_cpb_bias_fwd");
  return TOK_OK;
}

extern "C" int tok_cpb_bias_bwd(const float* dbias, int transposed, const void* table, int ld, const int64_t* index,
                                int heads, int n_tokens, int table_rows, void* dtable, void* stream) {
  TOK_CHECK_ARG(dbias && table && index && dtable && heads > 0 && n_tokens > 0 && table_rows > 0 && ld >= heads,
                "tok_cpb_bias_bwd: bad args");
  hipLaunchKernelGGL(cpb_bias_bwd_kernel, dim3(table_rows, ld), dim3(64), 0, tok_stream(stream), dbias,
                     (const bf16*)table, ld, index, heads, n_tokens, transposed, (bf16*)dtable);
  TOK_CHECK_LAUNCH("tok_cpb_bias_bwd");
  return TOK_OK;
}

extern "C" int tok_patch_merge(const void* src, void* dst, int batch, int h, int w, int c, int inverse, void* stream) {
  TOK_CHECK_ARG(src && dst && batch > 0 && h > 0 && w > 0 && h % 2 == 0 && w % 2 == 0 && c > 0 && c % 8 == 0,
                "tok_patch_merge: bad args");
  hipLaunchKernelGGL(patch_merge_kernel, dim3(blocks_for((size_t)batch * h * w * (c >> 3))), dim3(256), 0,
                     tok_stream(stream), (const bf16*)src, (bf16*)dst, batch, h, w, c, inverse);
  TOK_CHECK_LAUNCH("tok_patch_merge");
  return TOK_OK;
}
