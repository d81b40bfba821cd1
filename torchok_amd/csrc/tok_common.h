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
static inline int tok_cdiv(long long a, long long b) { return (int)((a + b - 1) / b); }

// ---- device helpers -----------------------------------------------------------------------
__device__ __forceinline__ float bf2f(bf16 v) { return (float)v; }
__device__ __forceinline__ bf16 f2bf(float v) { return (bf16)v; }  // RNE (v_cvt_pk_bf16_f32)

__device__ __forceinline__ bf16x8 ldg16(const bf16* p) {
  return *reinterpret_cast<const bf16x8*>(p);
}
__device__ __forceinline__ void stg16(bf16* p, bf16x8 v) { *reinterpret_cast<bf16x8*>(p) = v; }

// erf-GELU and its derivative: one definition for tok_act_fwd/_bwd and the fused GEMM epilogues, so that the fused and the
// unfused paths give the same bits.  erf by Abramowitz-Stegun 7.1.26 (|error| <= 1.5e-7, below fp32 rounding of the
// surrounding arithmetic and 4 decimal orders below the bf16 storage of the result): 1 - (a1 t + .. + a5 t^5) exp(-z^2),
// t = 1 / (1 + p |z|), on v_rcp_f32 / v_exp_f32 — ~14 VALU per element where libm's erff + expf cost ~55, which is what
// made a GELU epilogue as expensive as the elementwise pass it replaces.  exp(-z^2) with z = x / sqrt(2) is exp(-x^2 / 2):
// the derivative's Gaussian term reuses it.
__device__ __forceinline__ float tok_erf_core(float z, float& gauss) {   // returns erf(|z|), gauss = exp(-z^2)
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
