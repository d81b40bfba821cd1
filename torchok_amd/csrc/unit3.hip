// "Unit 3" of a bottleneck — 1x1 conv (P -> K = 4P channels) -> BatchNorm -> + shortcut -> ReLU — WITHOUT the K-channel
// pre-normalisation tensor y = z W^T ever reaching HBM ([timm] Bottleneck conv3 / bn3 / act3; the largest tensors of a
// ResNet step: written once and read three times by the conv -> BatchNorm -> apply chain).
//
// Forward.  y is linear in the unit's input z (M x P, the ReLU output of the 3x3 unit), so its batch statistics follow from
// the P x P second-moment matrix of z:      mean_k = w_k . mu_z        var_k = w_k^T (Z / M - mu_z mu_z^T) w_k
// with Z = z^T z (a weight-gradient-shaped launch on (z, z)), mu_z = colsum(z) / M and w_k the bf16-rounded filter row the
// GEMM multiplies with.  scale / shift are then known BEFORE the GEMM, whose epilogue writes relu(y * scale + shift +
// shortcut) directly (conv_igemm.hip, PWM 4).  tok_bn_gram_finalize is the counterpart of tok_bn_finalize.
//
// Backward.  dz = relu_mask * d(out) (the same tensor the shortcut receives), S1_k = sum_m dz.  With G = dz^T z (the weight-
// gradient launch, on dz instead of dy) the remaining BatchNorm-backward sum needs no y either:
//       S2_k = sum_m dz[m,k] y[m,k] = <G_k, w_k>
// and with the usual coefficients  dy = c1 dz + c2 y + c3  (c1 = gamma rstd, c2 = -c1 rstd dgamma / M, c3 = -c1 S1 / M - c2 mean):
//       dW_k  = c1_k G_k + c2_k (Z w_k) + c3_k colsum(z)
//       dz_in = dz (diag(c1) W) + z (W^T diag(c2) W) + c3^T W           (two GEMMs on tensors that exist + a bias)
// tok_bn3_bwd_prepare produces dgamma / dbeta / dW and the bf16 operands Wa = diag(c1) W, Wb = W^T diag(c2) W, cvec = c3^T W.
#include "tok_common.h"

namespace {

__device__ __forceinline__ float round_bf16(float v) { return bf2f(f2bf(v)); }

// ---- small fp32 GEMM on the vector units (the P x P x 4P products of this file: 1 ... 540 M multiply-adds) -------------------
//   C[i][j] = sum_l a(i, l) * B[l][j]      a(i, l) = TRANS_A ? A[l][i] : A[i][l]
// 64 x 64 output tile per workgroup, 256 threads x (4 x 4) outputs, reduction in steps of 16 through LDS.
// ROUND_A / ROUND_B: operand rounded to bf16 on load (the filter as the MFMA GEMMs see it).
// Epilogue kinds: 0 fp32 C;  1 rows < split as bf16 into Cb[i][j] (ldcb), row == split as fp32 into cvec[j]
template <bool TRANS_A, bool ROUND_A, bool ROUND_B>
__global__ __launch_bounds__(256) void gemm_f32_kernel(const float* __restrict__ A, int lda, const float* __restrict__ B, int ldb,
                                                       int M, int N, int L, float* __restrict__ C, int ldc, int epi, int split,
                                                       bf16* __restrict__ Cb, int ldcb, float* __restrict__ cvec) {
  __shared__ float As[16][64 + 4];     // [l][i]
  __shared__ float Bs[16][64 + 4];     // [l][j]
  const int tid = threadIdx.x;
  const int i0 = blockIdx.y * 64, j0 = blockIdx.x * 64;
  const int ti = tid >> 4, tj = tid & 15;          // thread owns rows ti*4.., cols tj*4..
  // split reduction: slice z covers [z * lchunk, (z + 1) * lchunk) and writes its own partial matrix (summed by
  // gemm_splits_reduce_kernel in slice order: deterministic)
  const int lchunk = ((L + (int)gridDim.z - 1) / (int)gridDim.z + 15) / 16 * 16;
  const int lbeg = blockIdx.z * lchunk;
  const int lend = min(L, lbeg + lchunk);
  if (gridDim.z > 1) C += (size_t)blockIdx.z * M * ldc;
  float acc[4][4];
#pragma unroll
  for (int r = 0; r < 4; ++r)
#pragma unroll
    for (int c = 0; c < 4; ++c) acc[r][c] = 0.f;
  for (int l0 = lbeg; l0 < lend; l0 += 16) {
    // stage A tile (64 x 16) and B tile (16 x 64): 1024 elements each, 4 per thread
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const int idx = tid + e * 256;
      int ii, ll;
      if (TRANS_A) { ii = idx & 63; ll = idx >> 6; }     // consecutive threads walk i: A[l][i] rows are contiguous in i
      else { ll = idx & 15; ii = idx >> 4; }             // consecutive threads walk l: A[i][l] rows are contiguous in l
      const int gi = i0 + ii, gl = l0 + ll;
      float v = 0.f;
      if (gi < M && gl < lend) v = TRANS_A ? A[(size_t)gl * lda + gi] : A[(size_t)gi * lda + gl];
      As[ll][ii] = ROUND_A ? round_bf16(v) : v;
      const int jj = idx & 63, lb = idx >> 6;
      const int gj = j0 + jj, glb = l0 + lb;
      float u = 0.f;
      if (gj < N && glb < lend) u = B[(size_t)glb * ldb + gj];
      Bs[lb][jj] = ROUND_B ? round_bf16(u) : u;
    }
    __syncthreads();
#pragma unroll
    for (int l = 0; l < 16; ++l) {
      const float4 av = *reinterpret_cast<const float4*>(&As[l][ti * 4]);
      const float4 bv = *reinterpret_cast<const float4*>(&Bs[l][tj * 4]);
      const float a4[4] = {av.x, av.y, av.z, av.w}, b4[4] = {bv.x, bv.y, bv.z, bv.w};
#pragma unroll
      for (int r = 0; r < 4; ++r)
#pragma unroll
        for (int c = 0; c < 4; ++c) acc[r][c] = fmaf(a4[r], b4[c], acc[r][c]);
    }
    __syncthreads();
  }
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    const int gi = i0 + ti * 4 + r;
    if (gi >= M) continue;
#pragma unroll
    for (int c = 0; c < 4; ++c) {
      const int gj = j0 + tj * 4 + c;
      if (gj >= N) continue;
      if (epi == 0) C[(size_t)gi * ldc + gj] = acc[r][c];
      else if (gi < split) Cb[(size_t)gi * ldcb + gj] = f2bf(acc[r][c]);
      else cvec[gj] = acc[r][c];
    }
  }
}

// sum of the split-reduction slices, in slice order; output kinds as in gemm_f32_kernel
__global__ __launch_bounds__(256) void gemm_splits_reduce_kernel(const float* __restrict__ parts, int splits, int M, int N,
                                                                 float* __restrict__ C, int ldc, int epi, int split_row,
                                                                 bf16* __restrict__ Cb, int ldcb, float* __restrict__ cvec) {
  const size_t total = (size_t)M * N;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
    float t = 0.f;
    for (int z = 0; z < splits; ++z) t += parts[(size_t)z * total + i];
    const int gi = (int)(i / N), gj = (int)(i - (size_t)gi * N);
    if (epi == 0) C[(size_t)gi * ldc + gj] = t;
    else if (gi < split_row) Cb[(size_t)gi * ldcb + gj] = f2bf(t);
    else cvec[gj] = t;
  }
}

// forward finalize from T = W_bf16 Z (K x P) and colsum(z): one wave per output channel k
//   mean_k = <w_k, mu_z>,   E[y^2]_k = <T_k, w_k> / M,   var = E[y^2] - mean^2  (fp64 folds)
__global__ __launch_bounds__(256) void bn_gram_finalize_kernel(
    const float* __restrict__ T, const float* __restrict__ zsum, const float* __restrict__ w, int64_t count, int P, int K,
    const float* __restrict__ gamma, const float* __restrict__ beta, float* __restrict__ running_mean,
    float* __restrict__ running_var, int64_t* __restrict__ nbt, float momentum, float eps, float* __restrict__ mean,
    float* __restrict__ rstd, float* __restrict__ scale, float* __restrict__ shift) {
  const int lane = threadIdx.x & 63;
  const int k = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (k < K) {
    const double inv = 1.0 / (double)count;
    double e2 = 0.0, m = 0.0;
    for (int p = lane; p < P; p += 64) {
      const double wq = (double)round_bf16(w[(size_t)k * P + p]);
      e2 += (double)T[(size_t)k * P + p] * wq;
      m += wq * (double)zsum[p];
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
      e2 += __shfl_xor(e2, off, 64);
      m += __shfl_xor(m, off, 64);
    }
    if (lane == 0) {
      m *= inv;
      double var = e2 * inv - m * m;
      if (var < 0.0) var = 0.0;
      const float muf = (float)m;
      const float rs = (float)(1.0 / sqrt(var + (double)eps));
      mean[k] = muf;
      rstd[k] = rs;
      const float sc = gamma[k] * rs;
      scale[k] = sc;
      shift[k] = fmaf(-muf, sc, beta[k]);
      if (running_mean != nullptr) {
        const double unbias = count > 1 ? (double)count / (double)(count - 1) : 1.0;
        running_mean[k] = (1.f - momentum) * running_mean[k] + momentum * muf;
        running_var[k] = (1.f - momentum) * running_var[k] + momentum * (float)(var * unbias);
      }
    }
  }
  if (blockIdx.x == 0 && threadIdx.x == 0 && nbt != nullptr) *nbt += 1;
}

// The same finalize for P in {64, 128, 256} WITHOUT the separate K x P x P product launch: one block per output channel forms
// its row T_k = w_k Z itself — thread (slice, p) sums a slice of the reduction for column p (Z rows read coalesced from L2),
// the slices are folded through LDS — stores it to T for the backward pass, and finishes the statistics.  (One wave per
// channel doing the whole row was latency-bound: +0.45 ms/step; the stand-alone product is 4 workgroups and 22 us of latency.)
__global__ __launch_bounds__(256) void bn_gram_finalize_block_kernel(
    float* __restrict__ T, const float* __restrict__ Z, const float* __restrict__ zsum, const float* __restrict__ w, int64_t count,
    int P, int K, const float* __restrict__ gamma, const float* __restrict__ beta, float* __restrict__ running_mean,
    float* __restrict__ running_var, int64_t* __restrict__ nbt, float momentum, float eps, float* __restrict__ mean,
    float* __restrict__ rstd, float* __restrict__ scale, float* __restrict__ shift) {
  __shared__ float red[256];
  __shared__ double cred[256];
  __shared__ double dred[2][4];
  const int k = blockIdx.x;
  const int tid = threadIdx.x;
  const int p = tid % P, sl = tid / P;
  const int slices = 256 / P, qper = P / slices;
  const float* wk = w + (size_t)k * P;
  // T_k = w_k Z in fp32 (an operand of the backward products) and, beside it, the CENTRED row w_k (Z - zsum zsum^T / M) in
  // fp64: var_k = w_k^T (Z / M - mu mu^T) w_k from E[y^2] - mean^2 cancels catastrophically for a channel with |mean| >> std
  // (fp32 Z over ~1e6 rows), and a negative result clamped to 0 would make rstd 1 / sqrt(eps)
  float acc = 0.f;
  double accc = 0.0;
  const double inv_m = 1.0 / (double)count;
  const double zp = (double)zsum[p] * inv_m;
  const int q0 = sl * qper;
  // the slice's loads 16 at a time (round 5: four at a time made P = 128 a chain of sixteen L2 round trips), same order of sums
  for (int qb = q0; qb < q0 + qper; qb += 16) {
    float wv[16], zv[16], sv[16];
#pragma unroll
    for (int u = 0; u < 16; ++u) {
      const int q = qb + u < q0 + qper ? qb + u : q0 + qper - 1;
      wv[u] = wk[q]; zv[u] = Z[(size_t)q * P + p]; sv[u] = zsum[q];
    }
#pragma unroll
    for (int u = 0; u < 16; ++u) {
      if (qb + u < q0 + qper) {
        const float wq = round_bf16(wv[u]);
        acc = fmaf(wq, zv[u], acc);
        accc += (double)wq * ((double)zv[u] - (double)sv[u] * zp);
      }
    }
  }
  red[tid] = acc;
  cred[tid] = accc;
  __syncthreads();
  double e2 = 0.0, m = 0.0;
  if (sl == 0) {
    float t = red[p];
    double tc = cred[p];
    for (int s_ = 1; s_ < slices; ++s_) { t += red[s_ * P + p]; tc += cred[s_ * P + p]; }
    T[(size_t)k * P + p] = t;
    const double wq = (double)round_bf16(wk[p]);
    e2 = tc * wq;                     // centred: sums to M * var_k
    m = wq * (double)zsum[p];
  }
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) {
    e2 += __shfl_xor(e2, off, 64);
    m += __shfl_xor(m, off, 64);
  }
  if ((tid & 63) == 0) { dred[0][tid >> 6] = e2; dred[1][tid >> 6] = m; }
  __syncthreads();
  if (tid == 0) {
    e2 = dred[0][0] + dred[0][1] + dred[0][2] + dred[0][3];
    m = dred[1][0] + dred[1][1] + dred[1][2] + dred[1][3];
    const double inv = 1.0 / (double)count;
    m *= inv;
    double var = e2 * inv;
    if (var < 0.0) var = 0.0;
    const float muf = (float)m;
    const float rs = (float)(1.0 / sqrt(var + (double)eps));
    mean[k] = muf;
    rstd[k] = rs;
    const float sc = gamma[k] * rs;
    scale[k] = sc;
    shift[k] = fmaf(-muf, sc, beta[k]);
    if (running_mean != nullptr) {
      const double unbias = count > 1 ? (double)count / (double)(count - 1) : 1.0;
      running_mean[k] = (1.f - momentum) * running_mean[k] + momentum * muf;
      running_var[k] = (1.f - momentum) * running_var[k] + momentum * (float)(var * unbias);
    }
    if (k == 0 && nbt != nullptr) *nbt += 1;
  }
}

// [Wb ; c] = At^T W_bf16 ((P + 1) x P, reduction over the K output channels) for P in {64, 128, 256}: one block per output ROW
// r — thread (slice, p) sums a slice of the reduction for column p (W rows read coalesced, At[k][r] is a broadcast), slices
// folded through LDS in slice order.  Replaces the tiled product + its split-reduce launch (two dependent launches of
// latency on the backward chain of every fused unit).
// (round 5: 1024 threads — four times the slices, each a quarter as long, 16 loads in flight: the reduction over K = 4P channels
//  was a chain of 8 ... 32 L2 round trips on the backward chain of every fused unit)
__global__ __launch_bounds__(1024) void bn3_wb_block_kernel(const float* __restrict__ At, const float* __restrict__ w, int P, int K,
                                                            bf16* __restrict__ wb, float* __restrict__ cvec) {
  __shared__ float red[1024];
  const int r = blockIdx.x;                 // 0 .. P: row of [Wb ; c]
  const int tid = threadIdx.x;
  const int p = tid % P, sl = tid / P;
  const int slices = 1024 / P;
  const int kper = (K + slices - 1) / slices;
  const int k0 = sl * kper, k1 = min(K, k0 + kper);
  float acc = 0.f;
  for (int kb = k0; kb < k1; kb += 16) {
    float av[16], wv[16];
#pragma unroll
    for (int u = 0; u < 16; ++u) {
      const int k = kb + u < k1 ? kb + u : k1 - 1;
      av[u] = At[(size_t)k * (P + 1) + r];
      wv[u] = w[(size_t)k * P + p];
    }
#pragma unroll
    for (int u = 0; u < 16; ++u)
      if (kb + u < k1) acc = fmaf(av[u], round_bf16(wv[u]), acc);
  }
  red[tid] = acc;
  __syncthreads();
  if (sl == 0) {
    float t = red[p];
    for (int s_ = 1; s_ < slices; ++s_) t += red[s_ * P + p];
    if (r < P) wb[(size_t)r * P + p] = f2bf(t);
    else cvec[p] = t;
  }
}

// dz = mask ? dout : 0 (in place allowed) and per-block partial sums of dz: partial[0][gridDim.x][C] (row 1 zero-filled so
// that the buffer has the layout of the other BatchNorm-backward partials).  Geometry of bn_bwd_reduce_kernel.
__global__ __launch_bounds__(256) void relu_mask_reduce_kernel(const bf16* dout, const uint8_t* __restrict__ mask, int64_t M,
                                                               int C, int cge, int rpb, bf16* dz,
                                                               float* __restrict__ partial) {
  __shared__ float red[256][8];
  const int tid = threadIdx.x;
  const int cgl = tid % cge, rl = tid / cge;
  const int cg_total = C >> 3;
  for (int cg = cgl; cg < cg_total; cg += cge) {
    float s1[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) s1[e] = 0.f;
    if (rl < rpb) {
      for (int64_t m = (int64_t)blockIdx.x * rpb + rl; m < M; m += (int64_t)gridDim.x * rpb) {
        const size_t off = (size_t)m * C + cg * 8;
        bf16x8 g = ldg16(dout + off);
        const unsigned bits = mask != nullptr ? mask[(size_t)m * cg_total + cg] : 0xffu;
#pragma unroll
        for (int e = 0; e < 8; ++e) {
          if (!((bits >> e) & 1u)) g[e] = (bf16)0.f;
          s1[e] += bf2f(g[e]);
        }
        if (mask != nullptr || dz != dout) stg16(dz + off, g);
      }
    }
#pragma unroll
    for (int e = 0; e < 8; ++e) red[tid][e] = s1[e];
    __syncthreads();
    if (rl == 0) {
      for (int r = 1; r < rpb; ++r)
#pragma unroll
        for (int e = 0; e < 8; ++e) s1[e] += red[r * cge + cgl][e];
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        partial[((size_t)0 * gridDim.x + blockIdx.x) * C + cg * 8 + e] = s1[e];
        partial[((size_t)1 * gridDim.x + blockIdx.x) * C + cg * 8 + e] = 0.f;
      }
    }
    __syncthreads();
  }
}

// ---- backward prepare, launch 1: 16 output channels k per workgroup (one wave walks 4 of them) --------------------------------
//   S1_k (fold of the partial rows, fp64), S2_k = <G_k, w_k>, dgamma / dbeta, c1 / c2 / c3,
//   dW_k = c1 G_k + c2 T_k + c3 colsum(z)        (T = W Z, kept from the forward pass)
//   Wa[p][k] = bf16(c1_k w_kp)  (dgrad pack layout [P][K]: transposed through LDS, 32-byte segments)
//   At[k][0..P) = c2_k w_kp,  At[k][P] = c3_k      (left operand of launch 2)
template <int U>
__device__ __forceinline__ void fold_partial(const float* __restrict__ partial, int rows, int K, int k, int lane, double& a1) {
  for (int rb = lane; rb < rows; rb += 64 * U) {
    float pv[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int r = rb + 64 * u;
      pv[u] = partial[(size_t)(r < rows ? r : rows - 1) * K + k];
    }
#pragma unroll
    for (int u = 0; u < U; ++u) a1 += rb + 64 * u < rows ? (double)pv[u] : 0.0;
  }
}

constexpr int PR_CH = 16;
// (round 5: one WAVE per channel — 1024 threads — instead of a wave walking four channels one after the other, and the
//  partial rows / G row requested up front: the kernel is a dependent chain of L2 round trips, 15 us on the backward chain of
//  every fused unit before)
__global__ __launch_bounds__(1024) void bn3_prepare_rows_kernel(
    const float* __restrict__ G, const float* __restrict__ w, const float* __restrict__ T, const float* __restrict__ zsum,
    const float* __restrict__ partial, int rows, int64_t count, int P, int K, const float* __restrict__ gamma,
    const float* __restrict__ mean, const float* __restrict__ rstd, float* dgamma, float* dbeta, int param_accumulate,
    float* __restrict__ coef, float* dw, int dw_accumulate, bf16* __restrict__ wa, float* __restrict__ At) {
  extern __shared__ __attribute__((aligned(16))) char smem_raw[];
  float* wl = reinterpret_cast<float*>(smem_raw);        // [PR_CH][P] c1-scaled bf16-rounded filter rows (for the Wa transpose)
  const int tid = threadIdx.x, lane = tid & 63, c = tid >> 6;
  const int k0 = blockIdx.x * PR_CH;
  const int k = k0 + c;
  if (k < K) {
    double a1 = 0.0, a2 = 0.0;
    // additions in row order; as many rows per lane in flight as the lane has (unconditional loads), up to 16
    if (rows <= 256) fold_partial<4>(partial, rows, K, k, lane, a1);
    else if (rows <= 512) fold_partial<8>(partial, rows, K, k, lane, a1);
    else fold_partial<16>(partial, rows, K, k, lane, a1);
    for (int p = lane; p < P; p += 64) a2 += (double)G[(size_t)k * P + p] * (double)round_bf16(w[(size_t)k * P + p]);
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
      a1 += __shfl_xor(a1, off, 64);
      a2 += __shfl_xor(a2, off, 64);
    }
    const double sx = (double)rstd[k] * (a2 - (double)mean[k] * a1);      // sum(dz * xhat)
    const double inv_m = 1.0 / (double)count;
    const float m1 = (float)(a1 * inv_m), m2 = (float)(sx * inv_m);
    const float g = gamma[k], rs = rstd[k], mu = mean[k];
    const float c1 = g * rs;
    const float c2 = -c1 * rs * m2;
    const float c3 = -c1 * m1 - c2 * mu;
    if (lane == 0) {
      const float sdz = (float)a1, sdzx = (float)sx;
      if (dgamma != nullptr) dgamma[k] = param_accumulate ? dgamma[k] + sdzx : sdzx;
      if (dbeta != nullptr) dbeta[k] = param_accumulate ? dbeta[k] + sdz : sdz;
      coef[k] = c1; coef[K + k] = c2; coef[2 * K + k] = c3;
      At[(size_t)k * (P + 1) + P] = c3;
    }
    for (int p = lane; p < P; p += 64) {
      const size_t o = (size_t)k * P + p;
      const float wq = round_bf16(w[o]);
      const float v = fmaf(c1, G[o], fmaf(c2, T[o], c3 * zsum[p]));
      dw[o] = dw_accumulate ? dw[o] + v : v;
      At[(size_t)k * (P + 1) + p] = c2 * wq;
      wl[c * P + p] = c1 * wq;
    }
  }
  __syncthreads();
  // Wa[p][k0 .. k0 + 16): one 32-byte segment per p
  for (int i = tid; i < P * (PR_CH / 8); i += 1024) {
    const int p = i / (PR_CH / 8), h = i % (PR_CH / 8);
    if (k0 + h * 8 + 8 <= K) {
      bf16x8 o;
#pragma unroll
      for (int e = 0; e < 8; ++e) o[e] = f2bf(wl[(h * 8 + e) * P + p]);
      stg16(wa + (size_t)p * K + k0 + h * 8, o);
    }
  }
}

inline int mr_geo(int c, int& cge, int& rpb) {
  const int cg_total = c / 8;
  cge = cg_total < 256 ? cg_total : 256;
  rpb = 256 / cge;
  return cg_total;
}

}  // namespace

extern "C" int tok_bn_bwd_rows(int64_t m, int c);

namespace {
// slices of the reduction so that ~512 workgroups exist (each with at least 64 reduction steps of work)
inline int gemm_splits(int M, int N, int L) {
  const int tiles = tok_cdiv(N, 64) * tok_cdiv(M, 64);
  int s = tok_cdiv(512, tiles);
  const int maxs = L / 64 > 0 ? L / 64 : 1;
  if (s > maxs) s = maxs;
  return s < 1 ? 1 : s;
}

template <bool TA, bool RA, bool RB>
void launch_gemm(const float* A, int lda, const float* B, int ldb, int M, int N, int L, float* C, int ldc, int epi, int split,
                 bf16* Cb, int ldcb, float* cvec, float* parts, hipStream_t st) {
  const int splits = parts != nullptr ? gemm_splits(M, N, L) : 1;
  if (splits == 1) {
    hipLaunchKernelGGL((gemm_f32_kernel<TA, RA, RB>), dim3(tok_cdiv(N, 64), tok_cdiv(M, 64), 1), dim3(256), 0, st, A, lda, B,
                       ldb, M, N, L, C, ldc, epi, split, Cb, ldcb, cvec);
    return;
  }
  hipLaunchKernelGGL((gemm_f32_kernel<TA, RA, RB>), dim3(tok_cdiv(N, 64), tok_cdiv(M, 64), splits), dim3(256), 0, st, A, lda, B,
                     ldb, M, N, L, parts, N, 0, 0, nullptr, 0, nullptr);
  const size_t total = (size_t)M * N;
  hipLaunchKernelGGL(gemm_splits_reduce_kernel, dim3((int)((total + 255) / 256 < 1024 ? (total + 255) / 256 : 1024)), dim3(256),
                     0, st, parts, splits, M, N, C, ldc, epi, split, Cb, ldcb, cvec);
}
}  // namespace

extern "C" int tok_bn_gram_finalize(const float* Z, const float* zsum, const float* w, int64_t count, int p, int k,
                                    const float* gamma, const float* beta, float* running_mean, float* running_var,
                                    int64_t* num_batches_tracked, float momentum, float eps, float* mean, float* rstd,
                                    float* scale, float* shift, float* wz, void* stream) {
  TOK_CHECK_ARG(Z && zsum && w && gamma && beta && mean && rstd && scale && shift && wz, "tok_bn_gram_finalize: null pointer");
  TOK_CHECK_ARG(count > 0 && p > 0 && k > 0, "tok_bn_gram_finalize: bad sizes");
  TOK_CHECK_ARG((running_mean == nullptr) == (running_var == nullptr), "tok_bn_gram_finalize: running stats go together");
  if (tok_dbg_skip(4)) return TOK_OK;
  hipStream_t st = tok_stream(stream);
  // wz = W_bf16 Z  (k x p): kept by the caller for tok_bn3_bwd_prepare
  if (p == 64 || p == 128 || p == 256) {
    hipLaunchKernelGGL(bn_gram_finalize_block_kernel, dim3(k), dim3(256), 0, st, wz, Z, zsum, w, count, p, k, gamma, beta,
                       running_mean, running_var, num_batches_tracked, momentum, eps, mean, rstd, scale, shift);
    TOK_CHECK_LAUNCH("tok_bn_gram_finalize");
    return TOK_OK;
  }
  launch_gemm<false, true, false>(w, p, Z, p, k, p, p, wz, p, 0, 0, nullptr, 0, nullptr, nullptr, st);
  TOK_CHECK_LAUNCH("tok_bn_gram_finalize(gemm)");
  hipLaunchKernelGGL(bn_gram_finalize_kernel, dim3(tok_cdiv(k, 4)), dim3(256), 0, st, wz, zsum, w, count, p, k, gamma, beta,
                     running_mean, running_var, num_batches_tracked, momentum, eps, mean, rstd, scale, shift);
  TOK_CHECK_LAUNCH("tok_bn_gram_finalize");
  return TOK_OK;
}

extern "C" int tok_relu_mask_reduce(const void* dout, const uint8_t* mask, int64_t m, int c, void* dz, float* partial,
                                    void* stream) {
  TOK_CHECK_ARG(dout && dz && partial && m > 0 && c > 0 && c % 8 == 0, "tok_relu_mask_reduce: bad args");
  int cge, rpb;
  mr_geo(c, cge, rpb);
  const int rows = tok_bn_bwd_rows(m, c);
  hipLaunchKernelGGL(relu_mask_reduce_kernel, dim3(rows), dim3(256), 0, tok_stream(stream), (const bf16*)dout, mask, m, c, cge,
                     rpb, (bf16*)dz, partial);
  TOK_CHECK_LAUNCH("tok_relu_mask_reduce");
  return TOK_OK;
}

extern "C" size_t tok_bn3_bwd_prepare_ws_floats(int p, int k) {
  return (size_t)k * (p + 1) + (size_t)gemm_splits(p + 1, p, k) * (p + 1) * p;
}

extern "C" int tok_bn3_bwd_prepare(const float* G, const float* w, const float* wz, const float* zsum, const float* partial,
                                   int rows, int64_t count, int p, int k, const float* gamma, const float* mean,
                                   const float* rstd, float* dgamma, float* dbeta, int param_accumulate, float* coef,
                                   float* dw, int dw_accumulate, void* wa, void* wb, float* cvec, float* ws, void* stream) {
  TOK_CHECK_ARG(G && w && wz && zsum && partial && gamma && mean && rstd && coef && dw && wa && wb && cvec && ws,
                "tok_bn3_bwd_prepare: null pointer");
  TOK_CHECK_ARG(rows > 0 && count > 0 && p > 0 && p <= 2048 && k > 0 && k % 8 == 0, "tok_bn3_bwd_prepare: bad sizes");
  if (tok_dbg_skip(4)) return TOK_OK;
  hipStream_t st = tok_stream(stream);
  const size_t smem = (size_t)PR_CH * p * sizeof(float);
  static const bool attr_set = [&] {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&bn3_prepare_rows_kernel), hipFuncAttributeMaxDynamicSharedMemorySize,
                              PR_CH * 2048 * 4);
    return true;
  }();   // once per process (thread-safe function-local static)
  (void)attr_set;
  float* At = ws;                                   // [k][p + 1]
  float* parts = ws + (size_t)k * (p + 1);          // split-reduction partials of the (p + 1) x p product
  hipLaunchKernelGGL(bn3_prepare_rows_kernel, dim3(tok_cdiv(k, PR_CH)), dim3(1024), smem, st, G, w, wz, zsum, partial, rows, count,
                     p, k, gamma, mean, rstd, dgamma, dbeta, param_accumulate, coef, dw, dw_accumulate, (bf16*)wa, At);
  TOK_CHECK_LAUNCH("tok_bn3_bwd_prepare(rows)");
  // [wb ; cvec] = At^T W_bf16   ((p + 1) x p, reduction over k)
  if (p == 64 || p == 128 || p == 256) {
    hipLaunchKernelGGL(bn3_wb_block_kernel, dim3(p + 1), dim3(1024), 0, st, (const float*)At, w, p, k, (bf16*)wb, cvec);
    TOK_CHECK_LAUNCH("tok_bn3_bwd_prepare(wb)");
    return TOK_OK;
  }
  launch_gemm<true, false, true>(At, p + 1, w, p, p + 1, p, k, nullptr, 0, 1, p, (bf16*)wb, p, cvec, parts, st);
  TOK_CHECK_LAUNCH("tok_bn3_bwd_prepare(wb)");
  return TOK_OK;
}
