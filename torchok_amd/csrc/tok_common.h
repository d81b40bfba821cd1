// Shared device/host helpers for libtok_gfx950.so (gfx950 / CDNA4 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stddef.h>
#include <stdio.h>
#include <stdarg.h>
#include <math.h>
#include "../../include/tok.h"

typedef __bf16 bf16;
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) __bf16 bf16x4;
typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(4))) uint32_t u32x4;
typedef __attribute__((ext_vector_type(2))) uint32_t u32x2;

#define TOK_WAVE 64

// ---- host-side error plumbing -------------------------------------------------------------
void tok_set_error(const char* fmt, ...);

#define TOK_CHECK_ARG(cond, ...)                 \
  do {                                           \
    if (!(cond)) {                               \
      tok_set_error(__VA_ARGS__);                \
      return TOK_ERR_INVALID;                    \
    }                                            \
  } while (0)

#define TOK_CHECK_LAUNCH(name)                                                   \
  do {                                                                           \
    hipError_t e__ = hipGetLastError();                                          \
    if (e__ != hipSuccess) {                                                     \
      tok_set_error("%s: launch failed: %s", name, hipGetErrorString(e__));      \
      return TOK_ERR_LAUNCH;                                                     \
    }                                                                            \
  } while (0)

static inline hipStream_t tok_stream(void* s) { return reinterpret_cast<hipStream_t>(s); }

// TOK_DBG_SKIP=<bitmask>: ablation probe for step-time budgeting (tools/ubench/runs/*): the named launches are NOT issued, results are
// garbage, only the step time means something (an upper bound on what removing / folding that launch class can buy).
//   1 BatchNorm finalizes (forward + backward)   2 wgrad_reduce_flat   4 fused-unit helpers (Gram finalize, prepare rows, Wb) + colsum_f32
//   8 BatchNorm apply passes (bn_act_fwd, bn_bwd_apply)   16 every weight-gradient launch   32 optimizer + repack
#include <stdlib.h>
// A process that runs with it set says so once on stderr: the launches it names return TOK_OK without running.
inline int tok_dbg_skip(int bit) {   // (inline, not static: ONE instance and one warning per process)
  static const int v = [] {
    const char* e = getenv("TOK_DBG_SKIP");
    const int m = (int)(e ? atoi(e) : 0);
    if (m != 0)
      fprintf(stderr, "libtok_gfx950: TOK_DBG_SKIP=%d — ablation mode, the named launch classes are NOT issued and every result of "
                      "this process is garbage (timing probe only; unset it for real runs)\n", m);
    return m;
  }();
  return v & bit;
}
static inline int tok_cdiv(long long a, long long b) { return (int)((a + b - 1) / b); }

// ---- device helpers -----------------------------------------------------------------------
__device__ __forceinline__ float bf2f(bf16 v) { return (float)v; }
__device__ __forceinline__ bf16 f2bf(float v) { return (bf16)v; }  // RNE (v_cvt_pk_bf16_f32)

__device__ __forceinline__ bf16x8 ldg16(const bf16* p) {
  return *reinterpret_cast<const bf16x8*>(p);
}
__device__ __forceinline__ void stg16(bf16* p, bf16x8 v) { *reinterpret_cast<bf16x8*>(p) = v; }

// erf-GELU and its derivative: one definition for tok_act_fwd/_bwd, the fused GEMM epilogues and the fused MLP, so that all
// paths give the same bits.  GELU(x) = x Phi(x) with the lower tail Phi(-a) = 0.5 erfc(a / sqrt 2) = exp2(P5(a)), a = min(|x|, 5.6):
// a degree-5 polynomial of log2 Phi(-a) (weighted minimax fit, |error in Phi| <= 3e-7 over the whole range; beyond 5.6 the tail is
// < 2e-8) and ONE v_exp_f32, then GELU(x) = max(x, 0) - a Phi(-a) for either sign.  8 VALU + 1 transcendental per element (the
// Abramowitz-Stegun 7.1.26 form of round 3a cost 12 + 2, libm's erff + expf ~55): in the GEMM epilogues and in the fused MLP the
// activation's VALU time is what competes with the MFMA pipe.  All of it is fma / med3, so the two-wide form (v_pk_fma_f32)
// rounds identically.  A NaN input gives 0 (med3 / max drop it); the residual stream next to every GELU here keeps the NaN.
typedef __attribute__((ext_vector_type(2))) float f32x2;
#define TOK_GELU_C0 (-1.0000008344650269f)
#define TOK_GELU_C1 (-1.1510831117630005f)
#define TOK_GELU_C2 (-0.45927679538726807f)
#define TOK_GELU_C3 (-0.05253789573907852f)
#define TOK_GELU_C4 (0.007387148682028055f)
#define TOK_GELU_C5 (-0.0005188509239815176f)
#define TOK_GELU_AMAX 5.6f
__device__ __forceinline__ float tok_gelu_tail(float a) {   // Phi(-a), 0 <= a <= 5.6
  float q = fmaf(a, TOK_GELU_C5, TOK_GELU_C4);
  q = fmaf(q, a, TOK_GELU_C3);
  q = fmaf(q, a, TOK_GELU_C2);
  q = fmaf(q, a, TOK_GELU_C1);
  q = fmaf(q, a, TOK_GELU_C0);
  return __builtin_amdgcn_exp2f(q);
}
#ifdef TOK_GELU_AS726   // A/B: the Abramowitz-Stegun 7.1.26 form (rcp + exp per element)
__device__ __forceinline__ float tok_erf_core(float z, float& gauss) {
  const float az = fabsf(z);
  const float t = __builtin_amdgcn_rcpf(fmaf(0.3275911f, az, 1.0f));
  float p = fmaf(t, 1.061405429f, -1.453152027f);
  p = fmaf(t, p, 1.421413741f);
  p = fmaf(t, p, -0.284496736f);
  p = fmaf(t, p, 0.254829592f);
  p *= t;
  gauss = __builtin_amdgcn_exp2f(az * az * -1.4426950408889634f);
  return fmaf(-p, gauss, 1.0f);
}
__device__ __forceinline__ float gelu_f(float x) {
  float g;
  const float e = copysignf(tok_erf_core(x * 0.70710678118654752f, g), x);
  const float hx = 0.5f * x;
  return fmaf(hx, e, hx);
}
__device__ __forceinline__ float gelu_d(float x) {
  float g;
  const float e = copysignf(tok_erf_core(x * 0.70710678118654752f, g), x);
  return fmaf(x * 0.3989422804014327f, g, fmaf(0.5f, e, 0.5f));
}
#else
__device__ __forceinline__ float gelu_f(float x) {
  const float a = __builtin_amdgcn_fmed3f(fabsf(x), 0.f, TOK_GELU_AMAX);
  const float m = __builtin_amdgcn_fmed3f(x, 0.f, __builtin_inff());
  return fmaf(-a, tok_gelu_tail(a), m);
}
__device__ __forceinline__ float gelu_d(float x) {           // Phi(x) + x phi(x)
  const float a = __builtin_amdgcn_fmed3f(fabsf(x), 0.f, TOK_GELU_AMAX);
  const float t = tok_gelu_tail(a);
  const float cdf = x >= 0.f ? 1.f - t : t;
  const float pdf = __builtin_amdgcn_exp2f(fmaf(x * x, -0.72134752044448170f, -1.3257480647361593f));
  return fmaf(x, pdf, cdf);
}
#endif
// two elements per instruction where the ISA has a packed form
__device__ __forceinline__ f32x2 tok_gelu_tail2(f32x2 a) {
  f32x2 q = __builtin_elementwise_fma(a, (f32x2)(TOK_GELU_C5), (f32x2)(TOK_GELU_C4));
  q = __builtin_elementwise_fma(q, a, (f32x2)(TOK_GELU_C3));
  q = __builtin_elementwise_fma(q, a, (f32x2)(TOK_GELU_C2));
  q = __builtin_elementwise_fma(q, a, (f32x2)(TOK_GELU_C1));
  q = __builtin_elementwise_fma(q, a, (f32x2)(TOK_GELU_C0));
  f32x2 e;
  e.x = __builtin_amdgcn_exp2f(q.x);
  e.y = __builtin_amdgcn_exp2f(q.y);
  return e;
}
__device__ __forceinline__ f32x2 gelu_f2(f32x2 x) {
  f32x2 a, m;
  a.x = __builtin_amdgcn_fmed3f(fabsf(x.x), 0.f, TOK_GELU_AMAX);
  a.y = __builtin_amdgcn_fmed3f(fabsf(x.y), 0.f, TOK_GELU_AMAX);
  m.x = __builtin_amdgcn_fmed3f(x.x, 0.f, __builtin_inff());
  m.y = __builtin_amdgcn_fmed3f(x.y, 0.f, __builtin_inff());
  return __builtin_elementwise_fma(-a, tok_gelu_tail2(a), m);
}

__device__ __forceinline__ f32x2 gelu_d2(f32x2 x) {          // the two-wide gelu_d: same operations, same bits
  f32x2 a;
  a.x = __builtin_amdgcn_fmed3f(fabsf(x.x), 0.f, TOK_GELU_AMAX);
  a.y = __builtin_amdgcn_fmed3f(fabsf(x.y), 0.f, TOK_GELU_AMAX);
  const f32x2 t = tok_gelu_tail2(a);
  f32x2 cdf;
  cdf.x = x.x >= 0.f ? 1.f - t.x : t.x;
  cdf.y = x.y >= 0.f ? 1.f - t.y : t.y;
  const f32x2 e = __builtin_elementwise_fma(x * x, (f32x2)(-0.72134752044448170f), (f32x2)(-1.3257480647361593f));
  f32x2 pdf;
  pdf.x = __builtin_amdgcn_exp2f(e.x);
  pdf.y = __builtin_amdgcn_exp2f(e.y);
  return __builtin_elementwise_fma(x, pdf, cdf);
}

__device__ __forceinline__ bf16x8 zero8() {
  bf16x8 z;
#pragma unroll
  for (int i = 0; i < 8; ++i) z[i] = (bf16)0.0f;
  return z;
}

// XCD-aware bijective block remap (guide T1): block b runs on XCD b % 8; give each XCD a
// contiguous range of logical tile ids so neighbouring tiles share that XCD's L2.
__device__ __forceinline__ int tok_xcd_remap(int b, int nwg) {
  const int xcd = b & 7, slot = b >> 3;
  const int q = nwg >> 3, r = nwg & 7;
  const int base = (xcd < r) ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q;
  return base + slot;
}

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
  return v;
}
