// Metric-learning head / loss kernels (small tensors: rows = batch or classes, C = embedding dim).
//   L2 row normalisation fwd/bwd (bf16 activations, fp32 class weights)      F.normalize, arcface_head.py:125-126,
//                                                                            linear_head.py:33-34
//   ArcFace margin fwd/bwd on the cosine matrix                              arcface_head.py:95-108
//   relevance matrix R[i][j] = (label_i == label_j)                          pairwise_task.py:87-107
//   contrastive loss fwd/bwd over pairwise Euclidean distances               pairwise.py:126-136 (torch.cdist)
// One wavefront per row, 64-lane butterflies, fp32 math.
#include "tok_common.h"
#include <math.h>

namespace {

// ---- row L2 normalisation: y = x / max(||x||, eps) ----------------------------------------------------
template <typename T> __device__ __forceinline__ float ldf(const T* p, int i);
template <> __device__ __forceinline__ float ldf<bf16>(const bf16* p, int i) { return bf2f(p[i]); }
template <> __device__ __forceinline__ float ldf<float>(const float* p, int i) { return p[i]; }
template <typename T> __device__ __forceinline__ void stf(T* p, int i, float v);
template <> __device__ __forceinline__ void stf<bf16>(bf16* p, int i, float v) { p[i] = f2bf(v); }
template <> __device__ __forceinline__ void stf<float>(float* p, int i, float v) { p[i] = v; }

template <typename T>
__global__ __launch_bounds__(256) void l2norm_fwd_kernel(const T* __restrict__ x, T* __restrict__ y,
                                                         float* __restrict__ inv_norm, int rows, int c, int ld,
                                                         float eps) {
  const int lane = threadIdx.x & 63;
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= rows) return;
  const T* xr = x + (size_t)row * ld;
  float ss = 0.f;
  for (int i = lane; i < c; i += 64) { const float v = ldf<T>(xr, i); ss = fmaf(v, v, ss); }
  ss = wave_sum(ss);
  const float inv = 1.0f / fmaxf(sqrtf(ss), eps);
  if (lane == 0) inv_norm[row] = inv;
  T* yr = y + (size_t)row * ld;
  for (int i = lane; i < ld; i += 64) stf<T>(yr, i, i < c ? ldf<T>(xr, i) * inv : 0.f);
}

// dx = (dy - y * <y, dy>) * inv_norm     (the clamp branch ||x|| < eps has zero measure: dx = dy * inv)
template <typename T>
__global__ __launch_bounds__(256) void l2norm_bwd_kernel(const T* __restrict__ dy, const T* __restrict__ y,
                                                         const float* __restrict__ inv_norm, T* dx, int accumulate,
                                                         int rows, int c, int ld) {
  const int lane = threadIdx.x & 63;
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= rows) return;
  const T* gr = dy + (size_t)row * ld;
  const T* yr = y + (size_t)row * ld;
  float dot = 0.f;
  for (int i = lane; i < c; i += 64) dot = fmaf(ldf<T>(gr, i), ldf<T>(yr, i), dot);
  dot = wave_sum(dot);
  const float inv = inv_norm[row];
  T* dr = dx + (size_t)row * ld;
  for (int i = lane; i < ld; i += 64) {
    float v = i < c ? (ldf<T>(gr, i) - ldf<T>(yr, i) * dot) * inv : 0.f;
    if (accumulate) v += ldf<T>(dr, i);
    stf<T>(dr, i, v);
  }
}

// ---- ArcFace margin -------------------------------------------------------------------------------------
struct Arc { float cos_m, sin_m, th, mm, scale; int easy; };

__global__ __launch_bounds__(256) void arcface_fwd_kernel(const bf16* __restrict__ cosine,
                                                          const int64_t* __restrict__ target, int rows, int classes,
                                                          int ld, Arc a, bf16* __restrict__ out) {
  const size_t total = (size_t)rows * ld;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
    const int r = (int)(i / ld), c = (int)(i - (size_t)r * ld);
    float v = 0.f;
    if (c < classes) {
      const float cs = bf2f(cosine[i]);
      v = cs;
      if ((int64_t)c == target[r]) {
        const float sn = sqrtf(fminf(fmaxf(1.0f - cs * cs, 0.f), 1.f));
        // `.type(cosine.dtype)` in the reference: phi is rounded to the activation dtype
        const float phi = bf2f(f2bf(cs * a.cos_m - sn * a.sin_m));
        v = a.easy ? (cs > 0.f ? phi : cs) : (cs > a.th ? phi : cs - a.mm);
      }
      v *= a.scale;
    }
    out[i] = f2bf(v);
  }
}

__global__ __launch_bounds__(256) void arcface_bwd_kernel(const bf16* __restrict__ cosine,
                                                          const int64_t* __restrict__ target,
                                                          const bf16* __restrict__ dout, int rows, int classes,
                                                          int ld, Arc a, bf16* __restrict__ dcos) {
  const size_t total = (size_t)rows * ld;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
    const int r = (int)(i / ld), c = (int)(i - (size_t)r * ld);
    float g = 0.f;
    if (c < classes) {
      float d = 1.f;
      if ((int64_t)c == target[r]) {
        const float cs = bf2f(cosine[i]);
        const float q = 1.0f - cs * cs;
        const bool use_phi = a.easy ? (cs > 0.f) : (cs > a.th);
        if (use_phi) {
          // d/dcos [cos*cos_m - sqrt(clamp(1-cos^2,0,1))*sin_m]; the clamp kills the sine term outside (0,1)
          const float dsn = (q > 0.f && q < 1.f) ? -cs / sqrtf(q) : 0.f;
          d = a.cos_m - dsn * a.sin_m;
        }
      }
      g = bf2f(dout[i]) * a.scale * d;
    }
    dcos[i] = f2bf(g);
  }
}

// ---- relevance matrix (exact) ----------------------------------------------------------------------------
__global__ void relevance_kernel(const int64_t* __restrict__ la, const int64_t* __restrict__ lb, int na, int nb,
                                 float* __restrict__ R) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < na * nb) R[i] = la[i / nb] == lb[i % nb] ? 1.f : 0.f;
}

// multi-label form (pairwise_task.py:103-105): R_ij = [ sum_c ya_ic * yb_jc > 0 ] on fp32 label matrices
__global__ void relevance_ml_kernel(const float* __restrict__ ya, const float* __restrict__ yb, int na, int nb, int classes,
                                    float* __restrict__ R) {
  const int idx = blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= na * nb) return;
  const float* a = ya + (size_t)(idx / nb) * classes;
  const float* b = yb + (size_t)(idx % nb) * classes;
  float s = 0.f;
  for (int c = 0; c < classes; ++c) s = fmaf(a[c], b[c], s);
  R[idx] = s > 0.f ? 1.f : 0.f;
}

// ---- contrastive loss: L_i = sum_j (1-R_ij) relu(mu - S_ij)^2 + R_ij S_ij^2,  S = ||e1_i - e2_j|| --------
// one wave per (i, j) pair strip: wave handles row i, loops j; lanes stride the embedding dim
__global__ __launch_bounds__(256) void contrastive_fwd_kernel(const bf16* __restrict__ e1, const bf16* __restrict__ e2,
                                                              const float* __restrict__ R, int n1, int n2, int d,
                                                              int ld, float mu, float* __restrict__ S,
                                                              float* __restrict__ row_loss) {
  const int lane = threadIdx.x & 63;
  const int i = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (i >= n1) return;
  const bf16* a = e1 + (size_t)i * ld;
  float li = 0.f;
  for (int j = 0; j < n2; ++j) {
    const bf16* b = e2 + (size_t)j * ld;
    float ss = 0.f;
    for (int k = lane; k < d; k += 64) { const float df = bf2f(a[k]) - bf2f(b[k]); ss = fmaf(df, df, ss); }
    ss = wave_sum(ss);
    const float s = sqrtf(ss);
    const float r = R[(size_t)i * n2 + j];
    const float h = fmaxf(mu - s, 0.f);
    li += (1.f - r) * h * h + r * ss;
    if (lane == 0) S[(size_t)i * n2 + j] = s;
  }
  if (lane == 0) row_loss[i] = li;
}

__global__ __launch_bounds__(256) void mean_kernel(const float* __restrict__ v, int n, float* out) {
  __shared__ double red[256];
  double s = 0.0;
  for (int i = threadIdx.x; i < n; i += 256) s += (double)v[i];
  red[threadIdx.x] = s;
  __syncthreads();
  for (int k = 128; k > 0; k >>= 1) {
    if (threadIdx.x < k) red[threadIdx.x] += red[threadIdx.x + k];
    __syncthreads();
  }
  if (threadIdx.x == 0) out[0] = (float)(red[0] / n);
}

// d e1_i = sum_j w_ij (e1_i - e2_j);  d e2_j = -sum_i w_ij (e1_i - e2_j);  w_ij = dL/dS_ij / S_ij * g / n1
// mode 0: rows of e1 (sum over j);  mode 1: rows of e2 (sum over i).  Deterministic gather, no atomics.
__global__ __launch_bounds__(256) void contrastive_bwd_kernel(const bf16* __restrict__ e1, const bf16* __restrict__ e2,
                                                              const float* __restrict__ R, const float* __restrict__ S,
                                                              const float* __restrict__ gscale, int n1, int n2, int d,
                                                              int ld, float mu, int mode, bf16* dx, int accumulate) {
  const int lane = threadIdx.x & 63;
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
  const int nrows = mode == 0 ? n1 : n2, nother = mode == 0 ? n2 : n1;
  if (row >= nrows) return;
  const float g = (gscale ? gscale[0] : 1.f) / (float)n1;
  const bf16* self = (mode == 0 ? e1 : e2) + (size_t)row * ld;
  const bf16* others = mode == 0 ? e2 : e1;
  for (int k = lane; k < ld; k += 64) {
    float acc = 0.f;
    if (k < d) {
      const float me = bf2f(self[k]);
      for (int o = 0; o < nother; ++o) {
        const size_t ij = mode == 0 ? (size_t)row * n2 + o : (size_t)o * n2 + row;
        const float s = S[ij], r = R[ij];
        // dL/dS = -2 (1-r) relu(mu - s) + 2 r s ;  dS/de = (e_self - e_other) / s  (0 at s = 0)
        const float dls = -2.f * (1.f - r) * fmaxf(mu - s, 0.f) + 2.f * r * s;
        const float w = s > 0.f ? dls / s : 0.f;
        acc = fmaf(w, me - bf2f(others[(size_t)o * ld + k]), acc);
      }
      acc *= g;
    }
    if (accumulate) acc += bf2f(dx[(size_t)row * ld + k]);
    dx[(size_t)row * ld + k] = f2bf(acc);
  }
}

// ---- NT-Xent (losses/representation/unsupervised.py:7-54): rows of E = cat(emb1, emb2) [2B][d] ----------------------------
// logits_ij = <e_i, e_j> / T with the diagonal at -1e9, label(i) = (i + B) mod 2B, mean cross-entropy.  One wave per row.
__global__ __launch_bounds__(256) void ntxent_fwd_kernel(const bf16* __restrict__ e, int n, int d, int ld, float inv_t,
                                                         float* __restrict__ lse, float* __restrict__ row_loss) {
  const int lane = threadIdx.x & 63;
  const int i = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (i >= n) return;
  const bf16* a = e + (size_t)i * ld;
  const int lab = (i + n / 2) % n;
  float mx = -INFINITY, den = 0.f, pos = 0.f;
  for (int j = 0; j < n; ++j) {
    const bf16* b = e + (size_t)j * ld;
    float s = 0.f;
    for (int k = lane; k < d; k += 64) s = fmaf(bf2f(a[k]), bf2f(b[k]), s);
    s = j == i ? -1e9f : wave_sum(s) * inv_t;
    if (j == lab) pos = s;
    const float nm = fmaxf(mx, s);
    den = den * expf(mx - nm) + expf(s - nm);
    mx = nm;
  }
  if (lane == 0) { const float l = mx + logf(den); lse[i] = l; row_loss[i] = l - pos; }
}

// d e_i = sum_{j != i} (w_ij + w_ji) e_j * g / (n T),  w_ij = softmax_ij - [j == label(i)]   (logits are symmetric)
__global__ __launch_bounds__(256) void ntxent_bwd_kernel(const bf16* __restrict__ e, const float* __restrict__ lse,
                                                         const float* __restrict__ gscale, int n, int d, int ld,
                                                         float inv_t, bf16* __restrict__ de) {
  const int lane = threadIdx.x & 63;
  const int i = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (i >= n) return;
  const bf16* a = e + (size_t)i * ld;
  const int half = n / 2;
  const float g = (gscale ? gscale[0] : 1.f) * inv_t / (float)n;
  const float li = lse[i];
  float acc[16];     // d <= 1024: 16 columns per lane
#pragma unroll
  for (int u = 0; u < 16; ++u) acc[u] = 0.f;
  for (int j = 0; j < n; ++j) {
    if (j == i) continue;
    const bf16* b = e + (size_t)j * ld;
    float s = 0.f;
    for (int k = lane; k < d; k += 64) s = fmaf(bf2f(a[k]), bf2f(b[k]), s);
    s = wave_sum(s) * inv_t;
    const float w = (expf(s - li) - (j == (i + half) % n ? 1.f : 0.f)) + (expf(s - lse[j]) - (i == (j + half) % n ? 1.f : 0.f));
#pragma unroll
    for (int u = 0; u < 16; ++u) {
      const int k = lane + u * 64;
      if (k < d) acc[u] = fmaf(w, bf2f(b[k]), acc[u]);
    }
  }
#pragma unroll
  for (int u = 0; u < 16; ++u) {
    const int k = lane + u * 64;
    if (k < ld) de[(size_t)i * ld + k] = f2bf(k < d ? acc[u] * g : 0.f);
  }
}

// ---- torch.nn.TripletMarginLoss (p = 2): d(x, y) = || x - y + eps ||_2, loss = mean(relu(d_ap - d_an + margin)) -------------
__global__ __launch_bounds__(256) void triplet_fwd_kernel(const bf16* __restrict__ a, const bf16* __restrict__ p,
                                                          const bf16* __restrict__ ng, int rows, int d, int ld, float margin,
                                                          float eps, int swap, float* __restrict__ dist,
                                                          float* __restrict__ row_loss) {
  const int lane = threadIdx.x & 63;
  const int i = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (i >= rows) return;
  const size_t o = (size_t)i * ld;
  float sap = 0.f, san = 0.f, spn = 0.f;
  for (int k = lane; k < d; k += 64) {
    const float av = bf2f(a[o + k]), pv = bf2f(p[o + k]), nv = bf2f(ng[o + k]);
    const float x = av - pv + eps, y = av - nv + eps, z = pv - nv + eps;
    sap = fmaf(x, x, sap); san = fmaf(y, y, san); spn = fmaf(z, z, spn);
  }
  const float dap = sqrtf(wave_sum(sap)), dan = sqrtf(wave_sum(san)), dpn = sqrtf(wave_sum(spn));
  if (lane == 0) {
    dist[i * 3 + 0] = dap; dist[i * 3 + 1] = dan; dist[i * 3 + 2] = dpn;
    const float dneg = (swap && dpn < dan) ? dpn : dan;
    row_loss[i] = fmaxf(dap - dneg + margin, 0.f);
  }
}

__global__ __launch_bounds__(256) void triplet_bwd_kernel(const bf16* __restrict__ a, const bf16* __restrict__ p,
                                                          const bf16* __restrict__ ng, const float* __restrict__ dist,
                                                          const float* __restrict__ gscale, int rows, int d, int ld,
                                                          float margin, float eps, int swap, bf16* __restrict__ da,
                                                          bf16* __restrict__ dp, bf16* __restrict__ dn) {
  const int lane = threadIdx.x & 63;
  const int i = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (i >= rows) return;
  const size_t o = (size_t)i * ld;
  const float dap = dist[i * 3], dan = dist[i * 3 + 1], dpn = dist[i * 3 + 2];
  const bool sw = swap && dpn < dan;
  const float dneg = sw ? dpn : dan;
  const bool active = dap - dneg + margin > 0.f;
  const float g = active ? (gscale ? gscale[0] : 1.f) / (float)rows : 0.f;
  for (int k = lane; k < ld; k += 64) {
    float ga = 0.f, gp = 0.f, gn = 0.f;
    if (k < d && active) {
      const float av = bf2f(a[o + k]), pv = bf2f(p[o + k]), nv = bf2f(ng[o + k]);
      const float up = dap > 0.f ? (av - pv + eps) / dap : 0.f;      // d d_ap / d a
      ga += up; gp -= up;
      if (sw) { const float un = dpn > 0.f ? (pv - nv + eps) / dpn : 0.f; gp -= un; gn += un; }
      else { const float un = dan > 0.f ? (av - nv + eps) / dan : 0.f; ga -= un; gn += un; }
    }
    da[o + k] = f2bf(ga * g); dp[o + k] = f2bf(gp * g); dn[o + k] = f2bf(gn * g);
  }
}

inline int grid_for(size_t total) {
  size_t b = (total + 255) / 256;
  return (int)(b > 4096 ? 4096 : (b < 1 ? 1 : b));
}

}  // namespace

extern "C" int tok_l2norm_fwd(const void* x, void* y, float* inv_norm, int rows, int c, int ld, int is_f32,
                              float eps, void* stream) {
  TOK_CHECK_ARG(x && y && inv_norm && rows > 0 && c > 0 && ld >= c, "tok_l2norm_fwd: bad args");
  if (is_f32) hipLaunchKernelGGL(l2norm_fwd_kernel<float>, dim3((rows + 3) / 4), dim3(256), 0, tok_stream(stream),
                                 (const float*)x, (float*)y, inv_norm, rows, c, ld, eps);
  else hipLaunchKernelGGL(l2norm_fwd_kernel<bf16>, dim3((rows + 3) / 4), dim3(256), 0, tok_stream(stream),
                          (const bf16*)x, (bf16*)y, inv_norm, rows, c, ld, eps);
  TOK_CHECK_LAUNCH("tok_l2norm_fwd");
  return TOK_OK;
}

extern "C" int tok_l2norm_bwd(const void* dy, const void* y, const float* inv_norm, void* dx, int accumulate,
                              int rows, int c, int ld, int is_f32, void* stream) {
  TOK_CHECK_ARG(dy && y && inv_norm && dx && rows > 0 && c > 0 && ld >= c, "tok_l2norm_bwd: bad args");
  if (is_f32) hipLaunchKernelGGL(l2norm_bwd_kernel<float>, dim3((rows + 3) / 4), dim3(256), 0, tok_stream(stream),
                                 (const float*)dy, (const float*)y, inv_norm, (float*)dx, accumulate, rows, c, ld);
  else hipLaunchKernelGGL(l2norm_bwd_kernel<bf16>, dim3((rows + 3) / 4), dim3(256), 0, tok_stream(stream),
                          (const bf16*)dy, (const bf16*)y, inv_norm, (bf16*)dx, accumulate, rows, c, ld);
  TOK_CHECK_LAUNCH("tok_l2norm_bwd");
  return TOK_OK;
}

extern "C" int tok_arcface_margin_fwd(const void* cosine, const int64_t* target, int rows, int classes, int ld,
                                      float cos_m, float sin_m, float th, float mm, int easy_margin, float scale,
                                      void* out, void* stream) {
  TOK_CHECK_ARG(cosine && target && out && rows > 0 && classes > 0 && ld >= classes, "tok_arcface_margin_fwd: bad args");
  const Arc a = {cos_m, sin_m, th, mm, scale, easy_margin};
  hipLaunchKernelGGL(arcface_fwd_kernel, dim3(grid_for((size_t)rows * ld)), dim3(256), 0, tok_stream(stream),
                     (const bf16*)cosine, target, rows, classes, ld, a, (bf16*)out);
  TOK_CHECK_LAUNCH("tok_arcface_margin_fwd");
  return TOK_OK;
}

extern "C" int tok_arcface_margin_bwd(const void* cosine, const int64_t* target, const void* dout, int rows,
                                      int classes, int ld, float cos_m, float sin_m, float th, float mm,
                                      int easy_margin, float scale, void* dcos, void* stream) {
  TOK_CHECK_ARG(cosine && target && dout && dcos && rows > 0 && classes > 0 && ld >= classes,
                "tok_arcface_margin_bwd: bad args");
  const Arc a = {cos_m, sin_m, th, mm, scale, easy_margin};
  hipLaunchKernelGGL(arcface_bwd_kernel, dim3(grid_for((size_t)rows * ld)), dim3(256), 0, tok_stream(stream),
                     (const bf16*)cosine, target, (const bf16*)dout, rows, classes, ld, a, (bf16*)dcos);
  TOK_CHECK_LAUNCH("tok_arcface_margin_bwd");
  return TOK_OK;
}

namespace {
// BasePairwiseLoss.regularize (pairwise.py:28-46): per embedding row sum|e| ('L1') or ||e||_2 ('L2'); a wavefront per row
__global__ __launch_bounds__(256) void embed_reg_fwd_kernel(const bf16* __restrict__ e, int n, int d, int ld, int mode,
                                                            float* __restrict__ row_reg) {
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
  const int lane = threadIdx.x & 63;
  if (row >= n) return;
  float s = 0.f;
  for (int c = lane; c < d; c += 64) {
    const float v = bf2f(e[(size_t)row * ld + c]);
    s += mode == 1 ? fabsf(v) : v * v;
  }
  s = wave_sum(s);
  if (lane == 0) row_reg[row] = mode == 1 ? s : sqrtf(s);
}

// d reg_i / d e_ic = sign(e_ic) ('L1') or e_ic / ||e_i|| ('L2', 0 for a zero row), times gscale * coeff; pad columns zeroed
__global__ __launch_bounds__(256) void embed_reg_bwd_kernel(const bf16* __restrict__ e, const float* __restrict__ row_reg,
                                                            const float* __restrict__ gscale, float coeff, int n, int d,
                                                            int ld, int mode, bf16* __restrict__ de) {
  const float g = gscale[0] * coeff;
  const size_t total = (size_t)n * ld;
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (size_t)gridDim.x * 256) {
    const int row = (int)(i / ld);
    const int c = (int)(i - (size_t)row * ld);
    float v = 0.f;
    if (c < d) {
      const float x = bf2f(e[i]);
      if (mode == 1) v = x > 0.f ? g : (x < 0.f ? -g : 0.f);
      else { const float nr = row_reg[row]; v = nr > 0.f ? g * x / nr : 0.f; }
    }
    de[i] = f2bf(v);
  }
}
}  // namespace

extern "C" int tok_embed_reg_fwd(const void* e, int n, int d, int ld, int mode, float* row_reg, float* out, void* stream) {
  TOK_CHECK_ARG(e && row_reg && out && n > 0 && d > 0 && ld >= d && (mode == 1 || mode == 2), "tok_embed_reg_fwd: bad args");
  hipStream_t st = tok_stream(stream);
  hipLaunchKernelGGL(embed_reg_fwd_kernel, dim3((n + 3) / 4), dim3(256), 0, st, (const bf16*)e, n, d, ld, mode, row_reg);
  TOK_CHECK_LAUNCH("tok_embed_reg_fwd");
  hipLaunchKernelGGL(mean_kernel, dim3(1), dim3(256), 0, st, row_reg, n, out);
  TOK_CHECK_LAUNCH("tok_embed_reg_fwd(mean)");
  return TOK_OK;
}

extern "C" int tok_embed_reg_bwd(const void* e, const float* row_reg, const float* gscale, float coeff, int n, int d,
                                 int ld, int mode, void* de, void* stream) {
  TOK_CHECK_ARG(e && row_reg && gscale && de && n > 0 && d > 0 && ld >= d && (mode == 1 || mode == 2),
                "tok_embed_reg_bwd: bad args");
  hipLaunchKernelGGL(embed_reg_bwd_kernel, dim3(grid_for((size_t)n * ld)), dim3(256), 0, tok_stream(stream),
                     (const bf16*)e, row_reg, gscale, coeff, n, d, ld, mode, (bf16*)de);
  TOK_CHECK_LAUNCH("tok_embed_reg_bwd");
  return TOK_OK;
}

extern "C" int tok_relevance_matrix(const int64_t* labels_a, const int64_t* labels_b, int na, int nb, float* R,
                                    void* stream) {
  TOK_CHECK_ARG(labels_a && labels_b && R && na > 0 && nb > 0, "tok_relevance_matrix: bad args");
  hipLaunchKernelGGL(relevance_kernel, dim3((na * nb + 255) / 256), dim3(256), 0, tok_stream(stream), labels_a,
                     labels_b, na, nb, R);
  TOK_CHECK_LAUNCH("tok_relevance_matrix");
  return TOK_OK;
}

extern "C" int tok_relevance_matrix_multilabel(const float* ya, const float* yb, int na, int nb, int classes, float* R,
                                               void* stream) {
  TOK_CHECK_ARG(ya && yb && R && na > 0 && nb > 0 && classes > 0, "tok_relevance_matrix_multilabel: bad args");
  hipLaunchKernelGGL(relevance_ml_kernel, dim3((na * nb + 255) / 256), dim3(256), 0, tok_stream(stream), ya, yb, na, nb,
                     classes, R);
  TOK_CHECK_LAUNCH("tok_relevance_matrix_multilabel");
  return TOK_OK;
}

extern "C" int tok_contrastive_fwd(const void* e1, const void* e2, const float* R, int n1, int n2, int d, int ld,
                                   float margin, float* S, float* row_loss, float* loss, void* stream) {
  TOK_CHECK_ARG(e1 && e2 && R && S && row_loss && loss && n1 > 0 && n2 > 0 && d > 0 && ld >= d,
                "tok_contrastive_fwd: bad args");
  hipStream_t st = tok_stream(stream);
  hipLaunchKernelGGL(contrastive_fwd_kernel, dim3((n1 + 3) / 4), dim3(256), 0, st, (const bf16*)e1, (const bf16*)e2,
                     R, n1, n2, d, ld, margin, S, row_loss);
  TOK_CHECK_LAUNCH("tok_contrastive_fwd");
  hipLaunchKernelGGL(mean_kernel, dim3(1), dim3(256), 0, st, row_loss, n1, loss);
  TOK_CHECK_LAUNCH("tok_contrastive_fwd(mean)");
  return TOK_OK;
}

extern "C" int tok_contrastive_bwd(const void* e1, const void* e2, const float* R, const float* S,
                                   const float* gscale, int n1, int n2, int d, int ld, float margin, void* de1,
                                   void* de2, int same_tensor, void* stream) {
  TOK_CHECK_ARG(e1 && e2 && R && S && de1 && n1 > 0 && n2 > 0 && d > 0 && ld >= d, "tok_contrastive_bwd: bad args");
  TOK_CHECK_ARG(same_tensor || de2, "tok_contrastive_bwd: de2 missing");
  hipStream_t st = tok_stream(stream);
  hipLaunchKernelGGL(contrastive_bwd_kernel, dim3((n1 + 3) / 4), dim3(256), 0, st, (const bf16*)e1, (const bf16*)e2,
                     R, S, gscale, n1, n2, d, ld, margin, 0, (bf16*)de1, 0);
  TOK_CHECK_LAUNCH("tok_contrastive_bwd");
  // second operand: into de2, or (emb1 is emb2, pairwise_task.py:79) accumulated onto de1
  hipLaunchKernelGGL(contrastive_bwd_kernel, dim3((n2 + 3) / 4), dim3(256), 0, st, (const bf16*)e1, (const bf16*)e2,
                     R, S, gscale, n1, n2, d, ld, margin, 1, (bf16*)(same_tensor ? de1 : de2), same_tensor ? 1 : 0);
  TOK_CHECK_LAUNCH("tok_contrastive_bwd(2)");
  return TOK_OK;
}

extern "C" int tok_ntxent_fwd(const void* emb, int n, int d, int ld, float temperature, float* lse, float* row_loss,
                              float* loss, void* stream) {
  TOK_CHECK_ARG(emb && lse && row_loss && loss && n > 1 && n % 2 == 0 && d > 0 && ld >= d && temperature > 0.f,
                "tok_ntxent_fwd: bad args");
  hipStream_t st = tok_stream(stream);
  hipLaunchKernelGGL(ntxent_fwd_kernel, dim3((n + 3) / 4), dim3(256), 0, st, (const bf16*)emb, n, d, ld, 1.f / temperature,
                     lse, row_loss);
  TOK_CHECK_LAUNCH("tok_ntxent_fwd");
  hipLaunchKernelGGL(mean_kernel, dim3(1), dim3(256), 0, st, row_loss, n, loss);
  TOK_CHECK_LAUNCH("tok_ntxent_fwd(mean)");
  return TOK_OK;
}

extern "C" int tok_ntxent_bwd(const void* emb, const float* lse, const float* gscale, int n, int d, int ld,
                              float temperature, void* demb, void* stream) {
  TOK_CHECK_ARG(emb && lse && demb && n > 1 && n % 2 == 0 && d > 0 && d <= 1024 && ld >= d && ld <= 1024 && temperature > 0.f,
                "tok_ntxent_bwd: bad args (d <= 1024)");
  hipLaunchKernelGGL(ntxent_bwd_kernel, dim3((n + 3) / 4), dim3(256), 0, tok_stream(stream), (const bf16*)emb, lse, gscale,
                     n, d, ld, 1.f / temperature, (bf16*)demb);
  TOK_CHECK_LAUNCH("tok_ntxent_bwd");
  return TOK_OK;
}

extern "C" int tok_triplet_fwd(const void* anchor, const void* positive, const void* negative, int rows, int d, int ld,
                               float margin, float eps, int swap, float* dist, float* row_loss, float* loss, void* stream) {
  TOK_CHECK_ARG(anchor && positive && negative && dist && row_loss && loss && rows > 0 && d > 0 && ld >= d,
                "tok_triplet_fwd: bad args");
  hipStream_t st = tok_stream(stream);
  hipLaunchKernelGGL(triplet_fwd_kernel, dim3((rows + 3) / 4), dim3(256), 0, st, (const bf16*)anchor, (const bf16*)positive,
                     (const bf16*)negative, rows, d, ld, margin, eps, swap, dist, row_loss);
  TOK_CHECK_LAUNCH("tok_triplet_fwd");
  hipLaunchKernelGGL(mean_kernel, dim3(1), dim3(256), 0, st, row_loss, rows, loss);
  TOK_CHECK_LAUNCH("tok_triplet_fwd(mean)");
  return TOK_OK;
}

extern "C" int tok_triplet_bwd(const void* anchor, const void* positive, const void* negative, const float* dist,
                               const float* gscale, int rows, int d, int ld, float margin, float eps, int swap,
                               void* d_anchor, void* d_positive, void* d_negative, void* stream) {
  TOK_CHECK_ARG(anchor && positive && negative && dist && d_anchor && d_positive && d_negative && rows > 0 && d > 0 &&
                ld >= d, "tok_triplet_bwd: bad args");
  hipLaunchKernelGGL(triplet_bwd_kernel, dim3((rows + 3) / 4), dim3(256), 0, tok_stream(stream), (const bf16*)anchor,
                     (const bf16*)positive, (const bf16*)negative, dist, gscale, rows, d, ld, margin, eps, swap,
                     (bf16*)d_anchor, (bf16*)d_positive, (bf16*)d_negative);
  TOK_CHECK_LAUNCH("tok_triplet_bwd");
  return TOK_OK;
}
