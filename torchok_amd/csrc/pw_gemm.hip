// Pointwise GEMM  Y[m][n] = sum_k X[m][k] W[n][k]  (1x1 / stride-1 convolutions forward and data-gradient, every nn.Linear)
// with a THREE-stage global -> LDS DMA ring.
//
// Why a second kernel next to conv_igemm.hip: the pointwise layers of a ResNet (K = 64 ... 512: one to eight 64-deep steps per
// output tile) and the token GEMMs of a transformer are streaming problems.  conv_igemm's two-buffer loop issues the loads of
// stage s+1, multiplies stage s and then waits for ALL loads at `__syncthreads()` (hipcc drains vmcnt(0) while an LDS-DMA is
// in flight): one exposed memory latency per stage unless other workgroups of the CU cover it — measured 3.0-4.5 TB/s on
// layers whose element-wise neighbours stream at 5.5.  Here two stages stay in flight across every barrier:
//     iteration s:  s_waitcnt vmcnt(L)  (L = DMA instructions per stage and thread: stage s landed, s+1 may still fly)
//                   s_barrier           (every wave's share of stage s landed; nobody reads stage s-1 any more)
//                   issue stage s+2 into the slot of stage s-1;  fragments + MFMAs (or the epilogue) of stage s
// Every stage issues EXACTLY L DMA instructions per thread (out-of-range offsets return zeros without traffic), so the counted
// wait is exact.  All LDS reads are inline asm: for hipcc a `ds_read` behind an in-flight LDS-DMA means "wait vmcnt(0)".
//
// Everything an epilogue needs from HBM rides the ring as well (ordinary loads in the loop would drain it):
//   * E1 stage: a [128 x BN] tile shaped like the output — the previous value of y (accumulate) or the shortcut of the
//     BatchNorm epilogue.  Accumulate: added to the accumulators in its own iteration.
//   * E2 stage: the raw conv output of the unit that produced the tensor whose gradient is completed here (BatchNorm-
//     backward sums in the dgrad epilogue).
//   * the ReLU mask bytes of the tile (1 / 4 KB) with the LAST stage of the tile.
// A tile is therefore KT K-stages (+ E1) (+ E2); the epilogue runs in the iteration of its last stage.
//
// Tiles, wave layout, swizzles, persistent XCD-aware tile walk, BatchNorm partial-sum butterfly: as in conv_igemm.hip.
#include "pw_gemm.h"
#include <stdlib.h>

namespace {

constexpr int BK = 64;
constexpr int BM = 128;

__device__ __forceinline__ int fw_swz(int n) {
  return ((0x78 >> (((n >> 3) & 3) << 1)) & 3) | (((n >> 1) & 1) << 2);
}

struct PwDiv { uint32_t mul, shift; };
PwDiv make_pwdiv(uint32_t d) {
  PwDiv f = {0, 0};
  if (d <= 1) return f;
  uint32_t l = 0;
  while ((1ull << l) < d) ++l;
  f.mul = (uint32_t)((((1ull << 32) * ((1ull << l) - d)) / d) + 1);
  f.shift = l;
  return f;
}
__device__ __forceinline__ uint32_t pwdiv(uint32_t n, PwDiv f) { return (uint32_t)(((uint64_t)__umulhi(n, f.mul) + n) >> f.shift); }

__device__ __forceinline__ u32x4 lds_read16(uint32_t addr) {
  u32x4 v;
  asm volatile("ds_read_b128 %0, %1" : "=v"(v) : "v"(addr));
  return v;
}
__device__ __forceinline__ uint32_t lds_read_u8(uint32_t addr) {
  uint32_t v;
  asm volatile("ds_read_u8 %0, %1" : "=v"(v) : "v"(addr));
  return v;
}

template <int BN, bool BNEP>
__global__ __launch_bounds__(256, BN == 64 ? 2 : 1) void pw_gemm_ring_kernel(PwArgs a, uint32_t x_bytes, uint32_t w_bytes,
                                                                             uint32_t y_bytes, uint32_t m_bytes, int KT,
                                                                             int SPT, uint32_t e1_bytes, PwDiv fd_hw, PwDiv fd_w,
                                                                             uint32_t x2_bytes, uint32_t w2_bytes, int KT1) {
  constexpr int NT = 256;
  constexpr int WGN = BN / 64;
  constexpr int WGM = (NT / 64) / WGN;
  constexpr int MT = BM / (WGM * 16);
  constexpr int RSTEP = NT / 8;
  constexpr int AROWS = BM / RSTEP;            // 4
  constexpr int WROWS = BN / RSTEP;            // 2 / 4
  constexpr int A_BYTES = BM * BK * 2;         // 16 KB
  constexpr int W_BYTES = BN * BK * 2;         // 8 / 16 KB
  constexpr int MK_BYTES = BN == 64 ? 1024 : 4096;   // mask bytes of the tile (one DMA instruction per thread)
  constexpr int STAGE = A_BYTES + W_BYTES + MK_BYTES;
  constexpr int LOADS = AROWS + WROWS + 1;     // DMA instructions per thread and stage
  constexpr int NST = 3;

  extern __shared__ __attribute__((aligned(16))) char smem[];
  typedef __attribute__((address_space(3))) char lds_char;
  typedef __attribute__((address_space(3))) void lds_void;
  const uint32_t lds_base = (uint32_t)(size_t)(lds_char*)smem;

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wv = tid >> 6;
  const int wn = wv % WGN;
  const int wm = wv / WGN;
  const int kc = tid & 7;
  const int lrow = tid >> 3;
  const int kcA = kc ^ (lrow & 7);
  const int kcW = kc ^ fw_swz(lrow);
  const int wave_u = __builtin_amdgcn_readfirstlane(tid >> 6);

  const __amdgpu_buffer_rsrc_t xsrd = __builtin_amdgcn_make_buffer_rsrc((void*)a.x, 0, x_bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t wsrd = __builtin_amdgcn_make_buffer_rsrc((void*)a.w, 0, w_bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t x2srd = __builtin_amdgcn_make_buffer_rsrc((void*)(a.x2 ? a.x2 : a.x), 0, a.x2 ? x2_bytes : 0, 0x00020000);
  const __amdgpu_buffer_rsrc_t w2srd = __builtin_amdgcn_make_buffer_rsrc((void*)(a.w2 ? a.w2 : a.w), 0, a.w2 ? w2_bytes : 0, 0x00020000);
  const __amdgpu_buffer_rsrc_t e1srd = __builtin_amdgcn_make_buffer_rsrc((void*)(a.e1 ? a.e1 : a.x), 0, a.e1 ? e1_bytes : 0, 0x00020000);
  const __amdgpu_buffer_rsrc_t e2srd = __builtin_amdgcn_make_buffer_rsrc((void*)(a.e2 ? a.e2 : a.x), 0, a.e2 ? y_bytes : 0, 0x00020000);
  const __amdgpu_buffer_rsrc_t msrd =
      __builtin_amdgcn_make_buffer_rsrc((void*)(a.mask_in ? (const void*)a.mask_in : (const void*)a.x), 0, a.mask_in ? m_bytes : 0, 0x00020000);

  // tile ownership (conv_igemm.hip): workgroup b on XCD b % 8, fixed channel tile, m-tiles it, it + sweep, ...
  const int xcd = blockIdx.x & 7, jx = blockIdx.x >> 3;
  const int bn_fixed = jx % a.gridN, jm = jx / a.gridN;
  const int S8 = (gridDim.x >> 3) / a.gridN;
  const int sweep = 8 * S8;
  const int n0 = bn_fixed * BN;
  const int it0 = xcd * S8 + jm;
  const int ntiles = it0 < a.gridM ? (a.gridM - it0 + sweep - 1) / sweep : 0;
  const int total = ntiles * SPT;
  const int nE1 = a.e1 != nullptr ? 1 : 0;
  const int mask_cols = a.N >> 3;              // mask bytes per pixel row

  // ---- loader: stage (itL, ksL) ----------------------------------------------------------------------------------------
  int itL = it0, ksL = 0;
  auto issue = [&](int slot) {
    char* Adst = smem + slot * STAGE + wave_u * 1024;
    char* Wdst = smem + slot * STAGE + A_BYTES + wave_u * 1024;
    char* Mdst = smem + slot * STAGE + A_BYTES + W_BYTES + wave_u * (MK_BYTES / 4);
    const bool live = itL < a.gridM;
    const int m0 = itL * BM;
    const bool kstage = ksL < KT;
    const bool last = ksL == SPT - 1;
    if (kstage) {
      const bool second = ksL >= KT1;            // wave-uniform: K stages of the second (x2, w2) source
      const int kl = second ? ksL - KT1 : ksL;
      const int Cc = second ? a.C2 : a.C;
      const __amdgpu_buffer_rsrc_t xs = second ? x2srd : xsrd;
      const __amdgpu_buffer_rsrc_t ws = second ? w2srd : wsrd;
      const int kcol = kl * BK + kcA * 8;
#pragma unroll
      for (int i = 0; i < AROWS; ++i) {
        const int m = m0 + lrow + RSTEP * i;
        uint32_t off = (live && m < a.M && kcol < Cc) ? (uint32_t)(m * Cc + kcol) * 2u : 0xFFFFFFF0u;
        asm volatile("" : "+v"(off));
        __builtin_amdgcn_raw_ptr_buffer_load_lds(xs, (lds_void*)(Adst + i * (RSTEP * 128)), 16, off, 0, 0, 0);
      }
      const int wcol = kl * BK + kcW * 8;
#pragma unroll
      for (int j = 0; j < WROWS; ++j) {
        const int n = n0 + lrow + RSTEP * j;
        uint32_t off = (live && n < a.N && wcol < Cc) ? (uint32_t)(n * Cc + wcol) * 2u : 0xFFFFFFF0u;
        asm volatile("" : "+v"(off));
        __builtin_amdgcn_raw_ptr_buffer_load_lds(ws, (lds_void*)(Wdst + j * (RSTEP * 128)), 16, off, 0, 0, 0);
      }
    } else {
      // epilogue operand tile: channels n0 .. n0+63 in the A region, n0+64 .. n0+127 (BN = 128) in the W region
      const bool first_e = (ksL - KT) == 0 && nE1;
      const __amdgpu_buffer_rsrc_t esrd = first_e ? e1srd : e2srd;
      const bool sub = first_e && a.e1_sub;      // wave-uniform
      // element offset of row m / channel n in the operand tensor; e1_sub: the row of the half-resolution tensor, even pixels only
      auto erow = [&](int m, int n, bool ok) -> uint32_t {
        if (!ok) return 0xFFFFFFF0u;
        if (sub) {
          const uint32_t b = pwdiv((uint32_t)m, fd_hw);
          const uint32_t rem = (uint32_t)m - b * (uint32_t)(a.sub_H * a.sub_W);
          const uint32_t h = pwdiv(rem, fd_w);
          const uint32_t w = rem - h * (uint32_t)a.sub_W;
          if ((h | w) & 1u) return 0xFFFFFFF0u;
          const uint32_t ph = (uint32_t)(a.sub_H + 1) >> 1, pw = (uint32_t)(a.sub_W + 1) >> 1;
          return (((b * ph + (h >> 1)) * pw + (w >> 1)) * (uint32_t)a.N + (uint32_t)n) * 2u;
        }
        return (uint32_t)(m * a.N + n) * 2u;
      };
#pragma unroll
      for (int i = 0; i < AROWS; ++i) {
        const int m = m0 + lrow + RSTEP * i;
        const int n = n0 + kcA * 8;
        uint32_t off = erow(m, n, live && m < a.M && n < a.N);
        asm volatile("" : "+v"(off));
        __builtin_amdgcn_raw_ptr_buffer_load_lds(esrd, (lds_void*)(Adst + i * (RSTEP * 128)), 16, off, 0, 0, 0);
      }
#pragma unroll
      for (int j = 0; j < WROWS; ++j) {
        const int m = m0 + lrow + RSTEP * j;
        const int n = n0 + 64 + kcA * 8;
        uint32_t off = erow(m, n, BN == 128 && live && m < a.M && n < a.N);
        asm volatile("" : "+v"(off));
        __builtin_amdgcn_raw_ptr_buffer_load_lds(esrd, (lds_void*)(Wdst + j * (RSTEP * 128)), 16, off, 0, 0, 0);
      }
    }
    {
      // ReLU bits of the tile with its last stage: BN/8 bytes per pixel row
      uint32_t off = 0xFFFFFFF0u;
      if (BN == 64) {
        const int row = tid >> 1;
        if (live && last && a.mask_in != nullptr && m0 + row < a.M) off = (uint32_t)((m0 + row) * mask_cols + (n0 >> 3) + (tid & 1) * 4);
        asm volatile("" : "+v"(off));
        __builtin_amdgcn_raw_ptr_buffer_load_lds(msrd, (lds_void*)Mdst, 4, off, 0, 0, 0);
      } else {
        if (live && last && a.mask_in != nullptr && tid < BM && m0 + tid < a.M) off = (uint32_t)((m0 + tid) * mask_cols + (n0 >> 3));
        asm volatile("" : "+v"(off));
        __builtin_amdgcn_raw_ptr_buffer_load_lds(msrd, (lds_void*)Mdst, 16, off, 0, 0, 0);
      }
    }
    if (++ksL == SPT) { ksL = 0; itL += sweep; }
  };

  // ---- fragment addressing --------------------------------------------------------------------------------------------
  const int sl = lane >> 4;
  const int li = lane & 15;
  const int wrow0 = wn * 64 + (li >> 2) * 8 + (li & 3);
  const int wswz = fw_swz(wrow0);
  const int arow0 = wm * (MT * 16) + li;
  const int aswz = li & 7;

  float s1[16], s2[16];
#pragma unroll
  for (int c = 0; c < 16; ++c) { s1[c] = 0.f; s2[c] = 0.f; }
  float esc[16], esh[16];
  if (BNEP) {
#pragma unroll
    for (int c = 0; c < 16; ++c) {
      const int n = n0 + wn * 64 + sl * 8 + (c >> 3) * 32 + (c & 7);
      esc[c] = n < a.N ? a.ep_scale[n] : 0.f;
      esh[c] = n < a.N ? a.ep_shift[n] : 0.f;
    }
  }
  float bia[16];
#pragma unroll
  for (int c = 0; c < 16; ++c) {
    const int n = n0 + wn * 64 + sl * 8 + (c >> 3) * 32 + (c & 7);
    bia[c] = (a.bias != nullptr && n < a.N) ? a.bias[n] : 0.f;
  }

  f32x4 acc[4][MT];
  auto zero_acc = [&]() {
#pragma unroll
    for (int t = 0; t < 4; ++t)
#pragma unroll
      for (int mt = 0; mt < MT; ++mt) acc[t][mt] = (f32x4){0.f, 0.f, 0.f, 0.f};
  };

  auto compute = [&](uint32_t sb) {
    const uint32_t Ab = sb, Wb = sb + A_BYTES;
    u32x4 wf[2][4], af[2][MT];
#pragma unroll
    for (int kk = 0; kk < 2; ++kk) {
      const int s = sl + 4 * kk;
#pragma unroll
      for (int t = 0; t < 4; ++t) wf[kk][t] = lds_read16(Wb + (wrow0 + (t >> 1) * 32 + (t & 1) * 4) * 128 + ((s ^ wswz) << 4));
#pragma unroll
      for (int mt = 0; mt < MT; ++mt) af[kk][mt] = lds_read16(Ab + (arow0 + mt * 16) * 128 + ((s ^ aswz) << 4));
    }
    asm volatile("s_waitcnt lgkmcnt(%0)" ::"n"(4 + MT) : "memory");
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int t = 0; t < 4; ++t)
#pragma unroll
      for (int mt = 0; mt < MT; ++mt)
        acc[t][mt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, wf[0][t]), __builtin_bit_cast(bf16x8, af[0][mt]),
                                                             acc[t][mt], 0, 0, 0);
    __builtin_amdgcn_sched_barrier(0);   // (the wait must not float above the first half's MFMAs)
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int t = 0; t < 4; ++t)
#pragma unroll
      for (int mt = 0; mt < MT; ++mt)
        acc[t][mt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, wf[1][t]), __builtin_bit_cast(bf16x8, af[1][mt]),
                                                             acc[t][mt], 0, 0, 0);
  };

  // this lane's [mt][half] 8-channel vectors of an epilogue operand tile sitting in slot `sb`
  auto etile = [&](uint32_t sb, bf16x8 (&dst)[MT][2]) {
    u32x4 raw[MT][2];
#pragma unroll
    for (int mt = 0; mt < MT; ++mt)
#pragma unroll
      for (int half = 0; half < 2; ++half)
        raw[mt][half] = lds_read16(sb + wn * A_BYTES + (arow0 + mt * 16) * 128 + (((sl + 4 * half) ^ aswz) << 4));
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int mt = 0; mt < MT; ++mt)
#pragma unroll
      for (int half = 0; half < 2; ++half) dst[mt][half] = __builtin_bit_cast(bf16x8, raw[mt][half]);
  };

  // E1 of an accumulating launch: acc += previous value (fp32 add of the bf16 tile, rounded once in the epilogue)
  auto add_e1 = [&](uint32_t sb) {
    bf16x8 ev[MT][2];
    etile(sb, ev);
#pragma unroll
    for (int mt = 0; mt < MT; ++mt)
#pragma unroll
      for (int half = 0; half < 2; ++half)
#pragma unroll
        for (int e = 0; e < 8; ++e) acc[half * 2 + (e >> 2)][mt][e & 3] += bf2f(ev[mt][half][e]);
  };

  auto epilogue = [&](int it, uint32_t sb) {
    const int m0 = it * BM;
    const int nb = n0 + wn * 64 + sl * 8;
    bf16x8 ev[MT][2];
    const bool need_tile = BNEP ? (a.e1 != nullptr) : (a.e2 != nullptr);
    if (need_tile) etile(sb, ev);
    uint32_t mbits[MT][2];
    if (a.mask_in != nullptr) {
#pragma unroll
      for (int mt = 0; mt < MT; ++mt)
#pragma unroll
        for (int half = 0; half < 2; ++half)
          mbits[mt][half] = lds_read_u8(sb + A_BYTES + W_BYTES + (arow0 + mt * 16) * (BN / 8) + wn * 8 + sl + half * 4);
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      __builtin_amdgcn_sched_barrier(0);
    }
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) {
      const int m = m0 + arow0 + mt * 16;
      if (m >= a.M) continue;
      bf16* yp = a.y + (size_t)m * a.N + nb;
#pragma unroll
      for (int half = 0; half < 2; ++half) {
        if (nb + half * 32 + 8 > a.N) continue;
        float v[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) v[e] = acc[half * 2 + (e >> 2)][mt][e & 3] + bia[half * 8 + e];
        bf16x8 o;
#pragma unroll
        for (int e = 0; e < 8; ++e) o[e] = f2bf(v[e]);
        const size_t eoff = (size_t)m * a.N + nb + half * 32;
        if (BNEP) {
          unsigned bits = 0;
#pragma unroll
          for (int e = 0; e < 8; ++e) {
            float z = fmaf(v[e], esc[half * 8 + e], esh[half * 8 + e]);
            if (a.e1 != nullptr) z += bf2f(ev[mt][half][e]);
            if (a.ep_relu) z = fmaxf(z, 0.f);
            o[e] = f2bf(z);
            bits |= (bf2f(o[e]) > 0.f ? 1u : 0u) << e;
          }
          if (a.mask_out != nullptr) a.mask_out[eoff >> 3] = (uint8_t)bits;
        }
        if (a.mask_store) {
          const unsigned bits = mbits[mt][half];
#pragma unroll
          for (int e = 0; e < 8; ++e) {
            if (!((bits >> e) & 1u)) o[e] = (bf16)0.f;
            s1[half * 8 + e] += bf2f(o[e]);
          }
        }
        stg16(yp + half * 32, o);
        if (a.stats != nullptr && !a.mask_store) {
          if (a.e2 != nullptr) {
            const unsigned bits = a.mask_in != nullptr ? mbits[mt][half] : 0xffu;
#pragma unroll
            for (int e = 0; e < 8; ++e) {
              const float dz = ((bits >> e) & 1u) ? bf2f(o[e]) : 0.f;
              s1[half * 8 + e] += dz;
              s2[half * 8 + e] = fmaf(dz, bf2f(ev[mt][half][e]), s2[half * 8 + e]);
            }
          } else {
#pragma unroll
            for (int e = 0; e < 8; ++e) {
              const float r = bf2f(o[e]);
              s1[half * 8 + e] += r;
              s2[half * 8 + e] = fmaf(r, r, s2[half * 8 + e]);
            }
          }
        }
      }
    }
  };

  // ---- the ring ---------------------------------------------------------------------------------------------------------
  zero_acc();
  issue(0);
  issue(1);
  int cur = 0, nxt = 2;
  int itC = it0, ksC = 0;
  for (int s = 0; s < total; ++s) {
    asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)" ::"n"(LOADS) : "memory");
    __builtin_amdgcn_s_barrier();
    issue(nxt);
    const uint32_t sb = lds_base + cur * STAGE;
    if (ksC < KT) {
      compute(sb);
    } else if (!BNEP && a.accumulate && ksC == KT) {
      add_e1(sb);
    }
    if (ksC == SPT - 1) {
      epilogue(itC, sb);
      zero_acc();
      ksC = 0;
      itC += sweep;
    } else {
      ++ksC;
    }
    cur = cur == NST - 1 ? 0 : cur + 1;
    nxt = nxt == NST - 1 ? 0 : nxt + 1;
  }
  asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");

  // ---- BatchNorm partial sums -> one row per workgroup (conv_igemm.hip) --------------------------------------------------
  if (a.stats != nullptr) {
    __syncthreads();
    float* red = reinterpret_cast<float*>(smem);     // [2][WGM][BN]: the ring is drained
#pragma unroll
    for (int step = 0; step < 4; ++step) {
      const int off = 8 >> step;
      const int cnt = 8 >> step;
      const bool up = (li & off) != 0;
#pragma unroll
      for (int j = 0; j < cnt; ++j) {
        const float send1 = up ? s1[j] : s1[j + cnt];
        const float send2 = up ? s2[j] : s2[j + cnt];
        const float keep1 = up ? s1[j + cnt] : s1[j];
        const float keep2 = up ? s2[j + cnt] : s2[j];
        s1[j] = keep1 + __shfl_xor(send1, off, 64);
        s2[j] = keep2 + __shfl_xor(send2, off, 64);
      }
    }
    const int nl = wn * 64 + (li >> 3) * 32 + sl * 8 + (li & 7);
    red[(0 * WGM + wm) * BN + nl] = s1[0];
    red[(1 * WGM + wm) * BN + nl] = s2[0];
    __syncthreads();
    if (tid < 2 * BN) {
      const int which = tid / BN;
      const int c = tid - which * BN;
      float t = 0.f;
#pragma unroll
      for (int w_ = 0; w_ < WGM; ++w_) t += red[(which * WGM + w_) * BN + c];
      const int row = xcd * S8 + jm;
      const int n = n0 + c;
      if (n < a.N) a.stats[((size_t)which * a.stat_rows + row) * a.N + n] = t;
    }
  }
}

int ring_flag() {   // TOK_PW_RING=0: every pointwise launch stays on conv_igemm's two-buffer loop (A/B switch)
  static const int v = [] { const char* e = getenv("TOK_PW_RING"); return (int)(e ? atoi(e) : 1); }();
  return v;
}

template <int BN, bool BNEP>
int launch_ring(const PwArgs& a, hipStream_t st) {
  constexpr int STAGE = BM * BK * 2 + BN * BK * 2 + (BN == 64 ? 1024 : 4096);
  constexpr int smem = 3 * STAGE;
  static const bool attr_set = [&] {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&pw_gemm_ring_kernel<BN, BNEP>),
                              hipFuncAttributeMaxDynamicSharedMemorySize, smem);
    return true;
  }();   // once per process (thread-safe function-local static)
  (void)attr_set;
  const int KT1 = tok_cdiv(a.C, BK);
  const int KT = KT1 + (a.x2 != nullptr ? tok_cdiv(a.C2, BK) : 0);
  const int SPT = KT + (a.e1 ? 1 : 0) + (a.e2 ? 1 : 0);
  const unsigned long long xb = (unsigned long long)a.M * a.C * 2, wb = (unsigned long long)a.N * a.C * 2,
                           yb = (unsigned long long)a.M * a.N * 2, mb = (unsigned long long)a.M * (a.N / 8);
  if (xb >= 0xFFFFFFF0ull || wb >= 0xFFFFFFF0ull || yb >= 0xFFFFFFF0ull) return 1;
  const unsigned long long x2b = a.x2 ? (unsigned long long)a.M * a.C2 * 2 : 0, w2b = a.x2 ? (unsigned long long)a.N * a.C2 * 2 : 0;
  if (x2b >= 0xFFFFFFF0ull) return 1;
  unsigned long long e1b = yb;
  PwDiv fd_hw = {0, 0}, fd_w = {0, 0};
  if (a.e1_sub) {
    const unsigned long long hw = (unsigned long long)a.sub_H * a.sub_W;
    e1b = (unsigned long long)(a.M / hw) * ((a.sub_H + 1) / 2) * ((a.sub_W + 1) / 2) * a.N * 2;
    fd_hw = make_pwdiv((uint32_t)hw);
    fd_w = make_pwdiv((uint32_t)a.sub_W);
  }
  const int grid = pw_ring_grid(BN, a.gridM, a.gridN);
  hipLaunchKernelGGL((pw_gemm_ring_kernel<BN, BNEP>), dim3(grid), dim3(256), smem, st, a, (uint32_t)xb, (uint32_t)wb, (uint32_t)yb,
                     (uint32_t)mb, KT, SPT, (uint32_t)e1b, fd_hw, fd_w, (uint32_t)x2b, (uint32_t)w2b, KT1);
  return 0;
}

}  // namespace

bool pw_ring_enabled() { return ring_flag() != 0; }

int pw_ring_grid(int bn_tile, int gridM, int gridN) {
  const int unit = 8 * gridN;
  int G = 256 * (bn_tile == 64 ? 2 : 1);
  const long long need = (long long)gridM * gridN;
  if (need < G) G = (int)((need + unit - 1) / unit) * unit;
  G = G / unit * unit;
  if (G < unit) G = unit;
  return G;
}

int pw_ring_launch(const PwArgs& a, int bn_tile, hipStream_t st) {
  if (!pw_ring_enabled()) return 1;
  if (a.C % 8 != 0 || a.N % 8 != 0) return 1;
  if (a.mask_in != nullptr && a.N % 64 != 0) return 1;       // mask rows are fetched as aligned 4 / 16-byte pieces
  if (a.accumulate && a.e1 == nullptr) return 1;
  if ((a.x2 == nullptr) != (a.w2 == nullptr) || (a.x2 != nullptr && (a.C2 <= 0 || a.C2 % 8 != 0))) return -1;
  if (a.e1_sub && (!a.accumulate || a.ep_scale != nullptr || a.sub_H <= 0 || a.sub_W <= 0 ||
                   a.M % (a.sub_H * a.sub_W) != 0)) return -1;
  const bool bnep = a.ep_scale != nullptr;
  if (bn_tile == 64) return bnep ? launch_ring<64, true>(a, st) : launch_ring<64, false>(a, st);
  if (bnep) return 1;
  return launch_ring<128, false>(a, st);
}
