// Transformer MLP with the hidden tile on chip (gfx950): y = W2 gelu(W1 x + b1) + b2 over a token matrix.
//
// Follows timm 0.6.13 models/layers/mlp.py: Mlp.forward (fc1 -> GELU -> fc2, drop = 0) as SwinV2 / DaViT call it
// (/root/reference/torchok/models/backbones/swin.py:18,238; davit.py:16,196).  Unfused, the 4C-wide hidden tensor is written twice
// (pre-activation + activation) and read back by fc2 and by three backward GEMMs: 38 C-wide tensor passes per block out of
// ~90.  Here a workgroup keeps a tile of tokens in registers, walks the hidden dimension in chunks and never stores it:
//
//   * transposed products, so that nothing crosses lanes between the two GEMMs: pre^T[hidden, token] = W1 x^T has the weight
//     rows as the MFMA A operand and x (16 bytes of a token row per lane, straight from global memory) as B; the D registers
//     of two 16-row blocks hold, per lane, 8 hidden units of ONE token — with the rows of W1 dealt to the two blocks as
//     {8q..8q+3} / {8q+4..8q+7} these are 8 CONSECUTIVE hidden units = the B fragment of y^T[C, token] += W2 h^T after
//     bias + GELU + bf16 rounding in place.
//   * weights stream through a three-slot LDS ring of 1-KB fragments, each the lane-linear image of one A operand
//     (buffer_load ... lds with per-lane source addresses: one DMA instruction = one fragment = one conflict-free
//     ds_read_b128), counted vmcnt + one raw barrier per stage, two stages in flight.  Every tile re-streams the 16 C^2 bytes
//     of weights out of L2: 256 tokens per workgroup keep that under the MFMA time.
//   * rounding points are those of the unfused launches (pre-activation and activation rounded to bf16), so the backward's
//     recomputation reproduces the forward bit for bit.
#include "tok_common.h"
#include <stdlib.h>
#include <type_traits>

#ifndef TOK_MLP_PROBE
#define TOK_MLP_PROBE 0   // timing probes (results are garbage): 1 no weight DMA, 2 no GELU, 3 no stage barrier, 4 no LDS fragment reads
#endif

namespace {

typedef __attribute__((address_space(3))) void lds_void;
typedef __attribute__((address_space(3))) char lds_char;

struct MlpArgs {
  const bf16* x;        // (T, C)
  const bf16* w1;       // (H, C)   fc1 forward pack
  const float* b1;      // (H)
  const bf16* w2;       // (C, H)   fc2 forward pack
  const float* b2;      // (C)
  bf16* y;              // (T, C)
  bf16* pre;            // (T, H)   fc1 output (bf16) and its GELU, for the backward GEMMs; SAVE kernels only
  bf16* act;
  uint32_t x_bytes, w_bytes, h_bytes;
  int accumulate;       // backward: dx += (the rows already there)
  int T, H, ntiles;
};

template <int N>
__device__ __forceinline__ void wait_vm_lgkm0() {
  asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)" ::"n"(N) : "memory");
}
// wave-uniform count -> immediate (the counts of a stage depend on where in the tile / walk it sits)
template <int LO, int HI>
__device__ __forceinline__ void wait_vm_dyn(int n) {
  if constexpr (LO == HI) {
    wait_vm_lgkm0<LO>();
  } else {
    constexpr int MID = (LO + HI) / 2;
    if (n <= MID) wait_vm_dyn<LO, MID>(n);
    else wait_vm_dyn<MID + 1, HI>(n);
  }
}
__device__ __forceinline__ u32x4 lds_read16(uint32_t addr) {
  u32x4 v;
  asm volatile("ds_read_b128 %0, %1" : "=v"(v) : "v"(addr));
  return v;
}
__device__ __forceinline__ f32x4 lds_read16f(uint32_t addr) {
  f32x4 v;
  asm volatile("ds_read_b128 %0, %1" : "=v"(v) : "v"(addr));
  return v;
}
// immediate offsets: without them every fragment address is a loop-invariant VGPR of its own
template <int OFF>
__device__ __forceinline__ u32x4 lds_read16o(uint32_t addr) {
  static_assert(OFF >= 0 && OFF < 65536, "ds offset range");
  u32x4 v;
  asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(v) : "v"(addr), "n"(OFF));
  return v;
}
template <int OFF>
__device__ __forceinline__ f32x4 lds_read16fo(uint32_t addr) {
  static_assert(OFF >= 0 && OFF < 65536, "ds offset range");
  f32x4 v;
  asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(v) : "v"(addr), "n"(OFF));
  return v;
}
template <int I, int N, class F>
__device__ __forceinline__ void static_for(F&& f) {
  if constexpr (I < N) {
    f(std::integral_constant<int, I>{});
    static_for<I + 1, N>(f);
  }
}

// C: model width; HC: hidden units per ring stage; NT: 16-token blocks per wave; WAVES per workgroup; PF: next tile's
// tokens are requested three stages before the tile ends (short tiles only: registers for a second copy of the rows).
// MODE 0, forward: in = x, A1 = fc1 pack, elementwise = + b1, bf16, GELU, bf16, A2 = fc2 pack, out = y + b2; SAVE writes the
//   pre-activation and activation rows.
// MODE 1, backward to the input: in = dy, A1 = fc2's dgrad pack (W2^T, [H][C]), elementwise = bf16, * GELU'(pre) with the saved
//   pre-activation rows (requested two stages ahead, 16 bytes per lane = the D-register layout), bf16, A2 = fc1's dgrad pack
//   (W1^T, [C][H]), out = dx (+ the rows already there: ACCUMULATE at run time); SAVE writes d(pre) for the weight gradients.
// Both reproduce the separate launches bit for bit (same rounding points, same accumulation order).
// two workgroups of four waves per CU while the output accumulators + token rows of a lane leave room for it (256 registers
// per lane at two waves per SIMD); the wide 4-wave forms run one wave per SIMD on the whole 512-register file
template <int C, int NT, int WAVES>
constexpr int mlp_min_blocks() { return (WAVES == 4 && (C / 16) * NT * 4 + (C / 32) * NT * 4 <= 160) ? 2 : 1; }

template <int C, int HC, int NT, int WAVES, bool PF, int MODE, bool SAVE, int RD>
__global__ __launch_bounds__(WAVES * 64, (mlp_min_blocks<C, NT, WAVES>())) void mlp_kernel(MlpArgs a) {
  constexpr int KS = C / 32, CB = C / 16, SUB = HC / 32;
  constexpr int FR_SUB = 2 * KS + CB;            // fragments of one 32-hidden sub-chunk: A1 (2 blocks x KS), A2 (CB)
  constexpr int NF = SUB * FR_SUB;
  constexpr int STAGE = NF * 1024;
  static_assert(NF % WAVES == 0, "fragments per stage must divide among the waves");
  constexpr int NI = NF / WAVES;                 // DMA instructions per wave and stage
  constexpr int TILE = WAVES * NT * 16;
  constexpr bool BWD = MODE == 1;

  extern __shared__ __attribute__((aligned(16))) char smem[];
  const uint32_t lds_base = (uint32_t)(size_t)(lds_char*)smem;
  const uint32_t b1_base = lds_base + 3 * STAGE;               // H floats   (forward)
  const uint32_t b2_base = b1_base + (uint32_t)a.H * 4;        // C floats

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int r = lane & 15, g = lane >> 4;
  const int NS = a.H / HC;

  const __amdgpu_buffer_rsrc_t xsrd = __builtin_amdgcn_make_buffer_rsrc((void*)a.x, 0, a.x_bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t ysrd = __builtin_amdgcn_make_buffer_rsrc((void*)a.y, 0, a.x_bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t w1srd = __builtin_amdgcn_make_buffer_rsrc((void*)a.w1, 0, a.w_bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t w2srd = __builtin_amdgcn_make_buffer_rsrc((void*)a.w2, 0, a.w_bytes, 0x00020000);
  const bool hrows = SAVE || BWD;
  const __amdgpu_buffer_rsrc_t psrd = __builtin_amdgcn_make_buffer_rsrc((void*)(hrows ? a.pre : a.y), 0, hrows ? a.h_bytes : 0, 0x00020000);
  // (forward, pre given without act: the activation rows go to a zero-length buffer — out-of-range stores are dropped by the
  //  range check, no traffic — so that one instruction stream serves "save both" and "save the pre-activation only")
  const bool arows = SAVE && a.act != nullptr;
  const __amdgpu_buffer_rsrc_t asrd = __builtin_amdgcn_make_buffer_rsrc((void*)(arows ? a.act : a.y), 0, arows ? a.h_bytes : 0, 0x00020000);

  // ---- biases into LDS (read back as float4 per lane) -------------------------------------------------------------------
  if constexpr (!BWD) {
    float* b1s = reinterpret_cast<float*>(smem + 3 * STAGE);
    for (int i = tid; i < a.H; i += WAVES * 64) b1s[i] = a.b1[i];
    float* b2s = b1s + a.H;
    for (int i = tid; i < C; i += WAVES * 64) b2s[i] = a.b2[i];
  }

  // ---- vector-memory ledger: every load / store / DMA instruction of this wave gets a sequence number; "everything up to
  // mark m has landed" is s_waitcnt vmcnt(issued - m) (the counter retires in issue order).  Stage- and tile-dependent
  // wait counts come out of this arithmetic instead of case analysis.
  int issued = 0;
  auto wait_until = [&](int mark) {
    int n = issued - mark;
    n = n < 0 ? 0 : (n > 63 ? 63 : n);
    wait_vm_dyn<0, 63>(n);
  };

  // ---- weight DMA: fragment f = j * WAVES + wave of a stage ------------------------------------------------------------------
  uint32_t voff[NI];
  int vinc[NI];
  bool isw1[NI];
#pragma unroll
  for (int j = 0; j < NI; ++j) {
    const int f = j * WAVES + wave;
    const int sc = f / FR_SUB, q = f - sc * FR_SUB;
    if (q < 2 * KS) {
      const int b = q / KS, ks = q - b * KS;
      const int row = sc * 32 + 8 * (r >> 2) + 4 * b + (r & 3);
      voff[j] = (uint32_t)((row * C + ks * 32 + 8 * g) * 2);
      vinc[j] = HC * C * 2;
      isw1[j] = true;
    } else {
      const int cb = q - 2 * KS;
      voff[j] = (uint32_t)(((cb * 16 + r) * a.H + sc * 32 + 8 * g) * 2);
      vinc[j] = HC * 2;
      isw1[j] = false;
    }
  }
  int dma_mark[3] = {0, 0, 0};
  auto issue = [&](int s, auto slotc, bool live) {
    constexpr int SLOT = decltype(slotc)::value;
#pragma unroll
    for (int j = 0; j < NI; ++j) {
      uint32_t off = (live && TOK_MLP_PROBE != 1) ? voff[j] + (uint32_t)(s * vinc[j]) : 0xFFFFFFF0u;
      asm volatile("" : "+v"(off));
      lds_void* dst = (lds_void*)(smem + SLOT * STAGE + (j * WAVES + wave) * 1024);
      if (isw1[j]) __builtin_amdgcn_raw_ptr_buffer_load_lds(w1srd, dst, 16, off, 0, 0, 0);
      else __builtin_amdgcn_raw_ptr_buffer_load_lds(w2srd, dst, 16, off, 0, 0, 0);
    }
    issued += NI;
    dma_mark[SLOT] = issued;
  };

  // ---- tokens of a tile: lane (r, g) holds channels ks * 32 + 8 g .. + 8 of token t0 + 16 n + r ------------------------------
  u32x4 xf[NT][KS], xn[PF ? NT : 1][PF ? KS : 1];
  int x_mark = 0;
  auto tok_row = [&](int tile, int n) { return (uint32_t)(tile * TILE + wave * (NT * 16) + n * 16 + r); };
  auto load_x = [&](u32x4 (*dst)[KS], int tile) {
#pragma unroll
    for (int n = 0; n < NT; ++n) {
      const uint32_t o = (tok_row(tile, n) * C + 8 * g) * 2;
#pragma unroll
      for (int ks = 0; ks < KS; ++ks) dst[n][ks] = __builtin_amdgcn_raw_buffer_load_b128(xsrd, o + ks * 64, 0, 0);
    }
    issued += NT * KS;
    x_mark = issued;
  };

  // ---- backward: saved pre-activation rows of a stage, one register set per ring slot --------------------------------------
  u32x4 pp[BWD ? 3 : 1][SUB][NT];
  int pre_mark[3] = {0, 0, 0};
  auto load_pre = [&](auto slotc, int tile, int sk, bool live) {
    constexpr int SLOT = decltype(slotc)::value;
    if constexpr (BWD) {
#pragma unroll
      for (int sc = 0; sc < SUB; ++sc)
#pragma unroll
        for (int n = 0; n < NT; ++n) {
          uint32_t off = live ? (tok_row(tile, n) * (uint32_t)a.H + (uint32_t)(sk * HC + sc * 32 + 8 * g)) * 2 : 0xFFFFFFF0u;
          pp[SLOT][sc][n] = __builtin_amdgcn_raw_buffer_load_b128(psrd, off, 0, 0);
        }
      issued += SUB * NT;
      pre_mark[SLOT] = issued;
    }
  };

  f32x4 acc[CB][NT];
  const uint32_t frag = lds_base + (uint32_t)lane * 16;

  // one stage out of ring slot SLOT; h0 = first hidden unit of the stage
  const uint32_t b1_lane = b1_base + (uint32_t)(8 * g * 4);
  auto compute = [&](auto slotc, int h0, int tile) {
    constexpr int SLOT = decltype(slotc)::value;
    const uint32_t sbase = frag + (uint32_t)(SLOT * STAGE);
    const uint32_t bia = b1_lane + (uint32_t)(h0 * 4);
    // fragment groups of the stage in consumption order — per sub-chunk KS first-product groups (the two row blocks of one
    // 32-channel step), then CB / 2 second-product groups (two output blocks) — on a three-deep register ring: two groups are
    // in flight while one is consumed (one group = 4 NT MFMAs = 64 NT cycles of matrix pipe; an LDS round trip under load
    // is longer than that), and the first two second-product groups of a sub-chunk travel under its elementwise part.
    constexpr int GPS = KS + CB / 2;               // groups per sub-chunk
    constexpr int M = SUB * GPS;
    f32x4 bv[BWD ? 1 : SUB][2];
    if constexpr (!BWD) {
      static_for<0, SUB>([&](auto scc) {
        constexpr int SC = decltype(scc)::value;
        bv[SC][0] = lds_read16fo<SC * 128>(bia);
        bv[SC][1] = lds_read16fo<SC * 128 + 16>(bia);
      });
    }
    u32x4 wf[RD][2];
    auto request = [&](auto idxc) {
      constexpr int idx = decltype(idxc)::value;
      constexpr int SC = idx / GPS, w = idx % GPS;
      constexpr int FB = SC * FR_SUB * 1024;
      if constexpr (TOK_MLP_PROBE == 4) {
        wf[idx % RD][0] = (u32x4){sbase, (uint32_t)idx, sbase, sbase};
        wf[idx % RD][1] = (u32x4){sbase, sbase, (uint32_t)idx, sbase};
      } else if constexpr (w < KS) {
        wf[idx % RD][0] = lds_read16o<FB + w * 1024>(sbase);
        wf[idx % RD][1] = lds_read16o<FB + (KS + w) * 1024>(sbase);
      } else {
        wf[idx % RD][0] = lds_read16o<FB + (2 * KS + 2 * (w - KS)) * 1024>(sbase);
        wf[idx % RD][1] = lds_read16o<FB + (2 * KS + 2 * (w - KS) + 1) * 1024>(sbase);
      }
    };
    static_for<0, RD - 1>([&](auto ic) { request(ic); });
    f32x4 d[2][NT];
    bf16x8 hf[NT];
    static_for<0, M>([&](auto idxc) {
      constexpr int idx = decltype(idxc)::value;
      constexpr int SC = idx / GPS, w = idx % GPS;
      // RD - 1 groups (two reads each) stay in flight behind the one consumed now
      if constexpr (idx + RD - 1 < M) request(std::integral_constant<int, idx + RD - 1>{});
      {
        constexpr int ahead = (M - 1 - idx) < (RD - 1) ? (M - 1 - idx) : (RD - 1);
        asm volatile("s_waitcnt lgkmcnt(%0)" ::"n"(2 * ahead) : "memory");
      }
      __builtin_amdgcn_sched_barrier(0);
      if constexpr (w < KS) {
        if constexpr (w == 0) {
#pragma unroll
          for (int b = 0; b < 2; ++b)
#pragma unroll
            for (int n = 0; n < NT; ++n) d[b][n] = (f32x4){0.f, 0.f, 0.f, 0.f};
        }
#pragma unroll
        for (int b = 0; b < 2; ++b)
#pragma unroll
          for (int n = 0; n < NT; ++n)
            d[b][n] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, wf[idx % RD][b]),
                                                              __builtin_bit_cast(bf16x8, xf[n][w]), d[b][n], 0, 0, 0);
        if constexpr (w == KS - 1) {
          // the D registers become the B fragment of the second product.  Forward: bias, bf16 rounding of the
          // pre-activation, GELU, bf16.  Backward: bf16 rounding of d(act), times GELU'(pre), bf16.
#pragma unroll
          for (int n = 0; n < NT; ++n) {
            u32x4 prw;
#pragma unroll
            for (int b = 0; b < 2; ++b) {
#pragma unroll
              for (int jp = 0; jp < 2; ++jp) {
                bf16x2 pr;
                if constexpr (!BWD) {
                  pr[0] = f2bf(d[b][n][2 * jp] + bv[SC][b][2 * jp]);
                  pr[1] = f2bf(d[b][n][2 * jp + 1] + bv[SC][b][2 * jp + 1]);
                } else {
                  pr[0] = f2bf(d[b][n][2 * jp]);
                  pr[1] = f2bf(d[b][n][2 * jp + 1]);
                }
                const uint32_t u = __builtin_bit_cast(uint32_t, pr);
                const float v0 = __builtin_bit_cast(float, u << 16), v1 = __builtin_bit_cast(float, u & 0xFFFF0000u);
                // the activation two elements per instruction (v_pk_fma_f32 for the polynomial: the elementwise part is
                // what competes with the MFMA pipe here); same operations as gelu_f / gelu_d, same bits
                if constexpr (!BWD) {
                  prw[2 * b + jp] = u;
                  const f32x2 gv = TOK_MLP_PROBE == 2 ? (f32x2){v0, v1} : gelu_f2((f32x2){v0, v1});
                  hf[n][4 * b + 2 * jp] = f2bf(gv.x);
                  hf[n][4 * b + 2 * jp + 1] = f2bf(gv.y);
                } else {
                  const uint32_t q = pp[SLOT][SC][n][2 * b + jp];
                  const float p0 = __builtin_bit_cast(float, q << 16), p1 = __builtin_bit_cast(float, q & 0xFFFF0000u);
                  const f32x2 dv = TOK_MLP_PROBE == 2 ? (f32x2){p0, p1} : gelu_d2((f32x2){p0, p1});
                  hf[n][4 * b + 2 * jp] = f2bf(v0 * dv.x);
                  hf[n][4 * b + 2 * jp + 1] = f2bf(v1 * dv.y);
                }
              }
            }
            if constexpr (SAVE) {
              const uint32_t off = (tok_row(tile, n) * (uint32_t)a.H + (uint32_t)(h0 + SC * 32 + 8 * g)) * 2;
              if constexpr (!BWD) __builtin_amdgcn_raw_buffer_store_b128(prw, psrd, off, 0, 0);
              __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, hf[n]), asrd, off, 0, 0);
            }
          }
          if constexpr (SAVE) issued += (BWD ? 1 : 2) * NT;
        }
      } else {
        constexpr int cg = w - KS;
#pragma unroll
        for (int c2 = 0; c2 < 2; ++c2)
#pragma unroll
          for (int n = 0; n < NT; ++n)
            acc[2 * cg + c2][n] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, wf[idx % RD][c2]), hf[n],
                                                                          acc[2 * cg + c2][n], 0, 0, 0);
      }
    });
  };

  // output tile: lane (r, g) holds channels 16 cb + 4 g .. + 4 of its token; v_permlane16_swap trades the two blocks of a pair
  // between neighbouring lane rows so that every lane owns 8 consecutive channels = one 16-byte store (half as many store
  // instructions: the tail of a tile is store-ISSUE bound)
  auto epilogue = [&](int tile) {
    const uint32_t b2_lane = b2_base + (uint32_t)(4 * g * 4);
    static_for<0, CB / 2>([&](auto cpc) {
      constexpr int cp = decltype(cpc)::value;
      f32x4 bva = (f32x4){0.f, 0.f, 0.f, 0.f}, bvb = bva;
      if constexpr (!BWD) {
        bva = lds_read16fo<(2 * cp) * 64>(b2_lane);
        bvb = lds_read16fo<(2 * cp + 1) * 64>(b2_lane);
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_sched_barrier(0);
      }
#pragma unroll
      for (int n = 0; n < NT; ++n) {
        const uint32_t rowb = tok_row(tile, n) * C * 2;
        f32x4 va = acc[2 * cp][n] + bva, vb = acc[2 * cp + 1][n] + bvb;
        if (BWD && a.accumulate) {
          // the rows already in dx, in this lane's own (pre-swap) channels
          const u32x2 oa = __builtin_amdgcn_raw_buffer_load_b64(ysrd, rowb + ((2 * cp) * 16 + 4 * g) * 2, 0, 0);
          const u32x2 ob = __builtin_amdgcn_raw_buffer_load_b64(ysrd, rowb + ((2 * cp + 1) * 16 + 4 * g) * 2, 0, 0);
          issued += 2;
          const bf16x4 fa = __builtin_bit_cast(bf16x4, oa), fb = __builtin_bit_cast(bf16x4, ob);
#pragma unroll
          for (int j = 0; j < 4; ++j) {
            va[j] += bf2f(fa[j]);
            vb[j] += bf2f(fb[j]);
          }
        }
        bf16x4 oa, ob;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          oa[j] = f2bf(va[j]);
          ob[j] = f2bf(vb[j]);
        }
        const u32x2 ua = __builtin_bit_cast(u32x2, oa), ub = __builtin_bit_cast(u32x2, ob);
        const u32x2 s0 = __builtin_amdgcn_permlane16_swap(ua[0], ub[0], false, false);
        const u32x2 s1 = __builtin_amdgcn_permlane16_swap(ua[1], ub[1], false, false);
        const u32x4 o = {s0[0], s1[0], s0[1], s1[1]};
        __builtin_amdgcn_raw_buffer_store_b128(o, ysrd, rowb + ((2 * cp + (g & 1)) * 16 + 4 * (g & 2)) * 2, 0, 0);
        acc[2 * cp][n] = (f32x4){0.f, 0.f, 0.f, 0.f};
        acc[2 * cp + 1][n] = (f32x4){0.f, 0.f, 0.f, 0.f};
      }
    });
    issued += NT * CB / 2;
  };

  // ---- persistent walk -------------------------------------------------------------------------------------------------
  const int tile0 = blockIdx.x, tstep = gridDim.x;
  const int mine = tile0 < a.ntiles ? (a.ntiles - tile0 + tstep - 1) / tstep : 0;
  const int total = mine * NS;
  if (mine > 0) load_x(xf, tile0);
  issue(0, std::integral_constant<int, 0>{}, total > 0);
  issue(1, std::integral_constant<int, 1>{}, total > 1);
  load_pre(std::integral_constant<int, 0>{}, tile0, 0, total > 0);
  load_pre(std::integral_constant<int, 1>{}, tile0, 1, total > 1);
#pragma unroll
  for (int cb = 0; cb < CB; ++cb)
#pragma unroll
    for (int n = 0; n < NT; ++n) acc[cb][n] = (f32x4){0.f, 0.f, 0.f, 0.f};

  int gi = 0;                                      // stage counter over the whole walk
  for (int it = 0; it < mine; ++it) {
    const int tile = tile0 + it * tstep;
    const bool has_next = it + 1 < mine;
    for (int s = 0; s < NS; s += 3) {
#define TOK_MLP_STAGE(K)                                                                                                        \
      {                                                                                                                         \
        const int sk = s + K;                                                                                                   \
        /* this stage's weights (and, backward, its pre-activation rows; first stage of a tile: its tokens) have landed */     \
        {                                                                                                                       \
          int need = dma_mark[K];                                                                                               \
          if (BWD && pre_mark[K] > need) need = pre_mark[K];                                                                    \
          if (sk == 0 && x_mark > need) need = x_mark;                                                                          \
          wait_until(need);                                                                                                     \
        }                                                                                                                       \
        if (TOK_MLP_PROBE != 3) __builtin_amdgcn_s_barrier();                                                                   \
        {                                                                                                                       \
          int s2 = sk + 2, t2 = tile;                                                                                           \
          if (s2 >= NS) { s2 -= NS; t2 += tstep; }                                                                              \
          issue(s2, std::integral_constant<int, (K + 2) % 3>{}, gi + 2 < total);                                                \
          load_pre(std::integral_constant<int, (K + 2) % 3>{}, t2, s2, gi + 2 < total);                                         \
        }                                                                                                                       \
        if constexpr (PF) {                                                                                                     \
          if (sk == NS - 3 && has_next) load_x(xn, tile + tstep);                                                               \
        }                                                                                                                       \
        compute(std::integral_constant<int, K>{}, sk * HC, tile);                                                               \
        ++gi;                                                                                                                   \
      }
      TOK_MLP_STAGE(0) TOK_MLP_STAGE(1) TOK_MLP_STAGE(2)
#undef TOK_MLP_STAGE
    }
    epilogue(tile);
    if (has_next) {
      if constexpr (PF) {
#pragma unroll
        for (int n = 0; n < NT; ++n)
#pragma unroll
          for (int ks = 0; ks < KS; ++ks) xf[n][ks] = xn[n][ks];
      } else {
        load_x(xf, tile + tstep);
      }
    }
  }
  asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
}

template <int C, int HC, int NT, int WAVES, bool PF, int MODE, bool SAVE, int RD>
void launch_v(const MlpArgs& a, hipStream_t st) {
  constexpr int STAGE = (HC / 32) * (2 * (C / 32) + C / 16) * 1024;
  const int smem = 3 * STAGE + (MODE == 0 ? (a.H + C) * 4 : 0);
  static const bool attr_set = [&] {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&mlp_kernel<C, HC, NT, WAVES, PF, MODE, SAVE, RD>),
                              hipFuncAttributeMaxDynamicSharedMemorySize, smem);
    return true;
  }();   // once per process (thread-safe function-local static)
  (void)attr_set;
  MlpArgs b = a;
  constexpr int TILE = WAVES * NT * 16;
  b.ntiles = (a.T + TILE - 1) / TILE;
  const int cap = 256 * mlp_min_blocks<C, NT, WAVES>();
  const int grid = b.ntiles < cap ? b.ntiles : cap;
  hipLaunchKernelGGL((mlp_kernel<C, HC, NT, WAVES, PF, MODE, SAVE, RD>), dim3(grid), dim3(WAVES * 64), smem, st, b);
}

template <int C, int HC, int NT, int WAVES, bool PF, int MODE, int RD = 3>
void launch_m(const MlpArgs& a, hipStream_t st) {
  static_assert(RD >= 2 && RD <= 8, "lgkmcnt is a 4-bit counter: at most 14 fragment reads in flight");
  const bool save = MODE == 0 ? a.pre != nullptr : a.act != nullptr;
  if (save) launch_v<C, HC, NT, WAVES, PF, MODE, true, RD>(a, st);
  else launch_v<C, HC, NT, WAVES, PF, MODE, false, RD>(a, st);
}

template <int MODE>
void launch_c(const MlpArgs& a, int c, hipStream_t st) {
  // measured and not kept (round 3, per call at the SwinV2-T stage shapes; tools/ubench/exp_mlp.sh): 4-wave workgroups with twice
  // the tokens per wave on the whole 512-register file (C = 384: 241 vs 223 us, C = 192: 323 vs 290 — one wave per SIMD cannot
  // overlap its own MFMA / VALU / LDS phases, and the C = 192 form spills); a deeper fragment register ring (RD 4 .. 8: 198..201
  // vs 199 us — the LDS round trip is not what the waves wait for).  Timing probes (TOK_MLP_PROBE): no weight DMA -20 %, no GELU
  // -8 % (C = 384) / -28 % (C = 192), no stage barrier -7 %: no single bound, the stage's three phases serialise per wave.
  // C = 192 forward on 64 hidden units per stage (half the stage barriers: 259 vs 292 us in save mode; the backward spills at
  // that stage size: 349 vs 278 us; C = 96 with 128 per stage: 406 vs 376 us)
  static const int w4 = [] { const char* e = getenv("TOK_MLP_W4"); return (int)(e ? atoi(e) : 0); }();
  if (c == 96 && w4) launch_m<96, 64, 2, 4, true, MODE>(a, st);
  else if (c == 192 && w4) launch_m<192, 32, 2, 4, false, MODE>(a, st);
  else if (c == 96) launch_m<96, 64, 2, 8, true, MODE>(a, st);
  else if (c == 192) {
    // (if constexpr: the 64-per-stage backward form spills 44-58 VGPRs and must not even be instantiated)
    if constexpr (MODE == 0) launch_m<192, 64, 2, 8, false, MODE>(a, st);
    else launch_m<192, 32, 2, 8, false, MODE>(a, st);
  }
  else launch_m<384, 32, 1, 8, false, MODE>(a, st);
}

int mlp_min_rows() {   // below this the serial chain of hidden chunks of one tile is longer than the two GEMM launches
  static const int v = [] { const char* e = getenv("TOK_MLP_MIN_ROWS"); return (int)(e ? atoi(e) : 32768); }();
  return v;
}
int mlp_max_c() {   // TOK_MLP_MAX_C=<c>: widths above c stay on the two GEMM launches (A/B switch per stage)
  static const int v = [] { const char* e = getenv("TOK_MLP_MAX_C"); return (int)(e ? atoi(e) : 384); }();
  return v;
}
int mlp_flag() {   // TOK_MLP_FUSED=0: the MLP stays on the two GEMM launches (A/B switch)
  static const int v = [] { const char* e = getenv("TOK_MLP_FUSED"); return (int)(e ? atoi(e) : 1); }();
  return v;
}

}  // namespace

extern "C" int tok_mlp_serves(int64_t rows, int c, int hidden) {
  if (!mlp_flag()) return 0;
  if (!(c == 96 || c == 192 || c == 384) || c > mlp_max_c()) return 0;
  if (hidden != 4 * c) return 0;
  if (rows < mlp_min_rows() || rows <= 0 || rows * (long long)hidden * 2 >= (1ll << 32)) return 0;
  return 1;
}

extern "C" int tok_mlp_fwd(const void* x, const void* w1, const float* b1, const void* w2, const float* b2, void* y, void* pre,
                           void* act, int64_t rows, int c, int hidden, void* stream) {
  TOK_CHECK_ARG(x && w1 && b1 && w2 && b2 && y, "tok_mlp_fwd: null pointer");
  TOK_CHECK_ARG(pre != nullptr || act == nullptr, "tok_mlp_fwd: act rows are saved together with the pre rows only");
  TOK_CHECK_ARG(tok_mlp_serves(rows, c, hidden), "tok_mlp_fwd: geometry (%lld, %d, %d) is not served (tok_mlp_serves)", (long long)rows, c,
                hidden);
  MlpArgs a;
  a.x = (const bf16*)x; a.w1 = (const bf16*)w1; a.b1 = b1; a.w2 = (const bf16*)w2; a.b2 = b2; a.y = (bf16*)y;
  a.pre = (bf16*)pre; a.act = (bf16*)act;
  a.h_bytes = (uint32_t)(rows * hidden * 2);
  a.T = (int)rows; a.H = hidden; a.ntiles = 0;
  a.x_bytes = (uint32_t)(rows * c * 2);
  a.w_bytes = (uint32_t)((long long)hidden * c * 2);
  a.accumulate = 0;
  launch_c<0>(a, c, tok_stream(stream));
  TOK_CHECK_LAUNCH("tok_mlp_fwd");
  return TOK_OK;
}

extern "C" int tok_mlp_bwd_dx(const void* dy, const void* w2_dgrad, const void* pre, const void* w1_dgrad, void* dx, int accumulate,
                              void* dpre, int64_t rows, int c, int hidden, void* stream) {
  TOK_CHECK_ARG(dy && w2_dgrad && pre && w1_dgrad && dx, "tok_mlp_bwd_dx: null pointer");
  TOK_CHECK_ARG(tok_mlp_serves(rows, c, hidden), "tok_mlp_bwd_dx: geometry (%lld, %d, %d) is not served (tok_mlp_serves)",
                (long long)rows, c, hidden);
  MlpArgs a;
  a.x = (const bf16*)dy; a.w1 = (const bf16*)w2_dgrad; a.b1 = nullptr; a.w2 = (const bf16*)w1_dgrad; a.b2 = nullptr;
  a.y = (bf16*)dx;
  a.pre = (bf16*)pre; a.act = (bf16*)dpre;
  a.h_bytes = (uint32_t)(rows * hidden * 2);
  a.T = (int)rows; a.H = hidden; a.ntiles = 0;
  a.x_bytes = (uint32_t)(rows * c * 2);
  a.w_bytes = (uint32_t)((long long)hidden * c * 2);
  a.accumulate = accumulate;
  launch_c<1>(a, c, tok_stream(stream));
  TOK_CHECK_LAUNCH("tok_mlp_bwd_dx");
  return TOK_OK;
}
