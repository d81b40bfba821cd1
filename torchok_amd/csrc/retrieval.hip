// Retrieval meters of the validation path: exhaustive nearest-neighbour search + ranking metrics.
//   sim_matrix       S = Q G^T (IP) or -(|q - g|^2) (L2), exact fp32     faiss IndexFlatIP / IndexFlatL2 (exact_index=True),
//                                                                        metrics/index_base_metric.py:523-545
//   topk_rows        k best columns of every row, (value desc, index asc); missing -> (-inf, -1) like faiss
//   retrieval_nrel   number of relevant vectors of every query           index_base_metric.py:342-418
//   retrieval_eval   hit rate / precision / recall / average precision / nDCG (Jarvelin) of every query after
//                    clear_faiss_output (:420-444)                      ranx 0.3.8 metrics via representation_ranx.py:29-53
// Embeddings stay fp32 end to end (a bf16 product would reorder near neighbours); the matrix is produced per query
// chunk and never leaves HBM, only nq floats per meter go back to the host.
#include "tok_common.h"
#include <math.h>

namespace {

// 64 x 64 output tile, 256 threads x (4 x 4), K in steps of 16 through LDS (k-major so that the inner product reads are
// broadcasts along one operand and conflict-free float4 along the other).
template <int L2>
__global__ __launch_bounds__(256) void sim_kernel(const float* __restrict__ q, const float* __restrict__ g, int nq, int ng,
                                                  int d, int ldq, int ldg, float* __restrict__ out, int64_t ldo) {
  __shared__ float As[16][68];
  __shared__ float Bs[16][68];
  const int tid = threadIdx.x;
  const int tx = tid & 15, ty = tid >> 4;
  const int q0 = blockIdx.y * 64, g0 = blockIdx.x * 64;
  const int lr = tid >> 2, lk = (tid & 3) * 4;   // loader: row lr of the tile, 4 consecutive k
  float acc[4][4];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[i][j] = 0.f;
  for (int k0 = 0; k0 < d; k0 += 16) {
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const int k = k0 + lk + e;
      As[lk + e][lr] = (q0 + lr < nq && k < d) ? q[(size_t)(q0 + lr) * ldq + k] : 0.f;
      Bs[lk + e][lr] = (g0 + lr < ng && k < d) ? g[(size_t)(g0 + lr) * ldg + k] : 0.f;
    }
    __syncthreads();
#pragma unroll
    for (int k = 0; k < 16; ++k) {
      float a[4], b[4];
#pragma unroll
      for (int i = 0; i < 4; ++i) a[i] = As[k][ty * 4 + i];
#pragma unroll
      for (int j = 0; j < 4; ++j) b[j] = Bs[k][tx * 4 + j];
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          if (L2) { const float t = a[i] - b[j]; acc[i][j] = fmaf(-t, t, acc[i][j]); }
          else acc[i][j] = fmaf(a[i], b[j], acc[i][j]);
        }
    }
    __syncthreads();
  }
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int r = q0 + ty * 4 + i;
    if (r >= nq) continue;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int c = g0 + tx * 4 + j;
      if (c < ng) out[(size_t)r * ldo + c] = acc[i][j];
    }
  }
}

// (v, i) precedes (pv, pi) in the ranking order: larger value first, lower index on ties
__device__ __forceinline__ bool ranks_before(float v, int i, float pv, int pi) { return v > pv || (v == pv && i < pi); }

// one wavefront per row; round r picks the best element that ranks after the pick of round r - 1
__global__ __launch_bounds__(256) void topk_rows_kernel(const float* __restrict__ s, int rows, int cols, int64_t ld, int k,
                                                        float* __restrict__ vals, int64_t* __restrict__ idx) {
  const int lane = threadIdx.x & 63;
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= rows) return;
  const float* sr = s + (size_t)row * ld;
  float pv = INFINITY;
  int pi = -1;
  for (int r = 0; r < k; ++r) {
    float bv = -INFINITY;
    int bi = 0x7fffffff;
    bool have = false;
    if (r < cols) {
      for (int c = lane; c < cols; c += 64) {
        const float v = sr[c];
        if (!(v == v)) continue;                                  // NaN never ranks
        if (r > 0 && !ranks_before(pv, pi, v, c)) continue;       // already taken
        if (!have || ranks_before(v, c, bv, bi)) { bv = v; bi = c; have = true; }
      }
#pragma unroll
      for (int off = 32; off > 0; off >>= 1) {
        const float ov = __shfl_xor(bv, off);
        const int oi = __shfl_xor(bi, off);
        const bool oh = __shfl_xor((int)have, off) != 0;
        if (oh && (!have || ranks_before(ov, oi, bv, bi))) { bv = ov; bi = oi; have = true; }
      }
    }
    if (lane == 0) {
      vals[(size_t)row * k + r] = have ? bv : -INFINITY;
      idx[(size_t)row * k + r] = have ? bi : -1;
    }
    if (!have) { pv = -INFINITY; pi = 0x7fffffff; } else { pv = bv; pi = bi; }
  }
}

__device__ __forceinline__ int wave_sum_i(int v) {
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off);
  return v;
}

// labels != NULL: classification data — relevant = same label, the query itself excluded.
// else: representation data — relevant = rows whose score in the query's column reaches relevance level 1.
__global__ __launch_bounds__(256) void nrel_kernel(const int64_t* __restrict__ labels, const float* __restrict__ scores,
                                                   int n, int n_cols, const int64_t* __restrict__ q_row,
                                                   const int64_t* __restrict__ q_col, int nq, int32_t* __restrict__ n_rel) {
  const int lane = threadIdx.x & 63;
  const int qi = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (qi >= nq) return;
  int cnt = 0;
  if (labels != nullptr) {
    const int64_t row = q_row[qi], lab = labels[row];
    for (int j = lane; j < n; j += 64) cnt += (labels[j] == lab && j != row) ? 1 : 0;
  } else {
    const int64_t col = q_col[qi];
    for (int j = lane; j < n; j += 64) cnt += scores[(size_t)j * n_cols + col] >= 1.f ? 1 : 0;
  }
  cnt = wave_sum_i(cnt);
  if (lane == 0) n_rel[qi] = cnt;
}

// one thread per query: the retrieved list is short (k), the arithmetic is sequential by definition (AP, DCG)
__global__ __launch_bounds__(256) void eval_kernel(int kind, const int64_t* __restrict__ idx, int kk,
                                                   const uint8_t* __restrict__ drop_first,
                                                   const int64_t* __restrict__ gallery, int ng,
                                                   const int64_t* __restrict__ labels, const float* __restrict__ scores,
                                                   int n_cols, const int64_t* __restrict__ q_row,
                                                   const int64_t* __restrict__ q_col, const int32_t* __restrict__ n_rel,
                                                   const float* __restrict__ ideal, int nq, float* __restrict__ out) {
  const int qi = blockIdx.x * 256 + threadIdx.x;
  if (qi >= nq) return;
  const int k = kk - 1;
  const int nr = n_rel[qi];
  if (nr == 0 || k <= 0) { out[qi] = 0.f; return; }     // ranx: no relevant document -> 0
  const int first = drop_first[qi] ? 1 : 0;             // clear_faiss_output: drop the head (the query) or the tail
  const int64_t row = q_row[qi];
  const int64_t lab = labels != nullptr ? labels[row] : 0;
  const int64_t col = labels != nullptr ? 0 : q_col[qi];
  double hits = 0.0, ap = 0.0, dcg = 0.0;
  for (int p = 0; p < k; ++p) {
    int64_t li = idx[(size_t)qi * kk + first + p];
    if (li < 0) li = ng - 1;                            // faiss label -1 indexes faiss_vector_idxs[-1] (:503)
    const int64_t gi = gallery != nullptr ? gallery[li] : li;
    float gain;
    if (labels != nullptr) gain = (labels[gi] == lab && gi != row) ? 1.f : 0.f;
    else { gain = scores[(size_t)gi * n_cols + col]; if (gain < 1.f) gain = 0.f; }
    if (gain > 0.f) {
      hits += 1.0;
      ap += hits / (double)(p + 1);
      dcg += (double)gain / log2((double)(p + 2));
    }
  }
  double v;
  switch (kind) {
    case 0: v = hits > 0.0 ? 1.0 : 0.0; break;
    case 1: v = hits / (double)k; break;
    case 2: v = hits / (double)nr; break;
    case 3: v = ap / (double)nr; break;
    default: {
      const int ki = k < nr ? k : nr;
      double idcg = 0.0;
      for (int p = 0; p < ki; ++p) {
        const double gsc = ideal != nullptr ? (double)ideal[(size_t)qi * k + p] : 1.0;
        idcg += gsc / log2((double)(p + 2));
      }
      v = dcg / idcg;
    }
  }
  out[qi] = (float)v;
}

}  // namespace

extern "C" int tok_sim_matrix(const float* q, const float* g, int nq, int ng, int d, int ldq, int ldg, int metric,
                              float* out, int64_t ldo, void* stream) {
  TOK_CHECK_ARG(q && g && out && nq > 0 && ng > 0 && d > 0 && ldq >= d && ldg >= d && ldo >= ng,
                "tok_sim_matrix: bad args");
  TOK_CHECK_ARG(metric == 0 || metric == 1, "tok_sim_matrix: metric must be 0 (IP) or 1 (L2), got %d", metric);
  const dim3 grid(tok_cdiv(ng, 64), tok_cdiv(nq, 64));
  TOK_CHECK_ARG(grid.y <= 65535, "tok_sim_matrix: more than 65535*64 query rows per call (%d)", nq);
  if (metric == 0)
    hipLaunchKernelGGL(sim_kernel<0>, grid, dim3(256), 0, tok_stream(stream), q, g, nq, ng, d, ldq, ldg, out, ldo);
  else
    hipLaunchKernelGGL(sim_kernel<1>, grid, dim3(256), 0, tok_stream(stream), q, g, nq, ng, d, ldq, ldg, out, ldo);
  TOK_CHECK_LAUNCH("tok_sim_matrix");
  return TOK_OK;
}

extern "C" int tok_topk_rows(const float* s, int rows, int cols, int64_t ld, int k, float* vals, int64_t* idx,
                             void* stream) {
  TOK_CHECK_ARG(s && vals && idx && rows > 0 && cols > 0 && ld >= cols && k > 0, "tok_topk_rows: bad args");
  hipLaunchKernelGGL(topk_rows_kernel, dim3(tok_cdiv(rows, 4)), dim3(256), 0, tok_stream(stream), s, rows, cols, ld, k,
                     vals, idx);
  TOK_CHECK_LAUNCH("tok_topk_rows");
  return TOK_OK;
}

extern "C" int tok_retrieval_nrel(const int64_t* labels, const float* scores, int n, int n_cols, const int64_t* q_row,
                                  const int64_t* q_col, int nq, int32_t* n_rel, void* stream) {
  TOK_CHECK_ARG((labels != nullptr) != (scores != nullptr), "tok_retrieval_nrel: exactly one of labels / scores");
  TOK_CHECK_ARG(n > 0 && nq > 0 && n_rel && (labels ? q_row != nullptr : (q_col != nullptr && n_cols > 0)),
                "tok_retrieval_nrel: bad args");
  hipLaunchKernelGGL(nrel_kernel, dim3(tok_cdiv(nq, 4)), dim3(256), 0, tok_stream(stream), labels, scores, n, n_cols,
                     q_row, q_col, nq, n_rel);
  TOK_CHECK_LAUNCH("tok_retrieval_nrel");
  return TOK_OK;
}

extern "C" int tok_retrieval_eval(int kind, const int64_t* idx, int kk, const uint8_t* drop_first, const int64_t* gallery,
                                  int ng, const int64_t* labels, const float* scores, int n_cols, const int64_t* q_row,
                                  const int64_t* q_col, const int32_t* n_rel, const float* ideal, int nq, float* out,
                                  void* stream) {
  TOK_CHECK_ARG(kind >= 0 && kind <= 4, "tok_retrieval_eval: kind must be 0..4, got %d", kind);
  TOK_CHECK_ARG((labels != nullptr) != (scores != nullptr), "tok_retrieval_eval: exactly one of labels / scores");
  TOK_CHECK_ARG(idx && drop_first && q_row && n_rel && out && kk >= 2 && ng > 0 && nq > 0 &&
                (labels || (q_col != nullptr && n_cols > 0)), "tok_retrieval_eval: bad args");
  hipLaunchKernelGGL(eval_kernel, dim3(tok_cdiv(nq, 256)), dim3(256), 0, tok_stream(stream), kind, idx, kk, drop_first,
                     gallery, ng, labels, scores, n_cols, q_row, q_col, n_rel, ideal, nq, out);
  TOK_CHECK_LAUNCH("tok_retrieval_eval");
  return TOK_OK;
}
