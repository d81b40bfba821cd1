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

// exact (erf) GELU and its derivative: one definition for tok_act_fwd/_bwd and the fused GEMM epilogues, so that the
// fused and the unfused paths give the same bits
__device__ __forceinline__ float gelu_f(float x) { return 0.5f * x * (1.f + erff(x * 0.70710678118654752f)); }
__device__ __forceinline__ float gelu_d(float x) {
  return 0.5f * (1.f + erff(x * 0.70710678118654752f)) + x * 0.3989422804014327f * expf(-0.5f * x * x);
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
