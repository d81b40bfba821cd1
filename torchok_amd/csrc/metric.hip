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

extern "C" int tok_relevance_matrix(const int64_t* labels_a, const int64_t* labels_b, int na, int nb, float* R,
                                    void* stream) {
  TOK_CHECK_ARG(labels_a && labels_b && R && na > 0 && nb > 0, "tok_relevance_matrix: bad args");
  hipLaunchKernelGGL(relevance_kernel, dim3((na * nb + 255) / 256), dim3(256), 0, tok_stream(stream), labels_a,
                     labels_b, na, nb, R);
  TOK_CHECK_LAUNCH("tok_relevance_matrix");
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
