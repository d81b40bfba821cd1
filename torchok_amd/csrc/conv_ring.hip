// Implicit-GEMM convolution / token GEMM on 256 x 128 output tiles with a THREE-stage global -> LDS DMA ring (gfx950).
//
//   Y[m][n] = sum_k A[m][k] * Wt[n][k]      m = (b, p, q)   k = (r, s, c)   NHWC bf16, fp32 accumulate
//
// Why a third GEMM kernel next to conv_igemm.hip (128 x {64,128} tiles, two LDS buffers, vmcnt(0) + barrier per 64-deep K
// step) and pw_gemm.hip (the ring for 64-wide pointwise tiles): the deep-K layers — 3x3 convolutions (K = 576 ... 4608), the
// 1x1 layers of ResNet stages 3-4 (K = 256 ... 2048), the token GEMMs of a transformer's deep stages — are bound by what a
// 64 x 64 wave tile costs per MFMA: one 16-byte LDS fragment read per two MFMAs, one DMA row per four, and an exposed
// vmcnt(0) per K step (profiles/r02_*: 590 TFLOP/s = 24 % of the MFMA peak in the step, 29 % of wave cycles in s_waitcnt).
// Here
//   * a wave owns 128 pixels x 64 channels (acc[4][8] = 128 accumulator registers): 12 fragment reads per 32 MFMAs
//     (-25 % LDS bytes per MFMA), 6 DMA instructions per thread per 32 MFMAs (-25 % staged bytes per MFMA);
//   * a K stage is 32 deep: (256 + 128) rows x 64 B = 24 KB, three stages = 72 KB -> TWO workgroups per CU (2 waves per
//     SIMD: one wave's fragment reads / epilogue run under the other's MFMAs), two stages in flight per workgroup across
//     every barrier (s_waitcnt vmcnt(6) + raw s_barrier, exact because every stage issues the same six DMA instructions —
//     out-of-range offsets return zeros without traffic);
//   * 64-byte LDS rows: one ds_read_b128 lane group covers rows {0-3, 12-15} at k-chunk s and rows {4-11} at chunk s+1
//     (MI355X_MICROARCH.md, LDS table); chunk' = chunk ^ F[(row >> 2) & 3] with F = {0, 2, 3, 1} puts the 16 lanes on 16
//     distinct 16-byte slots of the 256-byte bank row (weight rows, fed in channel-interleaved order, use (row >> 3) & 3).
//     The LDS image of a DMA wave instruction is lane-linear, so the swizzle is applied to the SOURCE chunk (guide rule 21).
// Everything else follows conv_igemm.hip: weights are the MFMA A operand (a lane ends up with 8 consecutive channels of a
// pixel: 16-byte NHWC stores), persistent XCD-aware tile walk with a fixed channel tile per workgroup, BatchNorm partial
// sums in registers across tiles, dgrad = the same kernel on the tap-flipped transposed pack, epilogue options accumulate /
// bias / forward statistics / backward statistics of the producing BatchNorm / masked store.
#include "conv_common.h"
#include <stdlib.h>

#ifdef TOK_BUILD_EXPERIMENTS   // measured neutral on the step: not in the default library

namespace {

constexpr int RBM = 256, RBN = 128, RBK = 32, RNST = 3;

__device__ __forceinline__ int ring_f(int q) { return (0x78 >> ((q & 3) << 1)) & 3; }   // {0, 2, 3, 1}

__device__ __forceinline__ u32x4 rlds16(uint32_t addr) {
  u32x4 v;
  asm volatile("ds_read_b128 %0, %1" : "=v"(v) : "v"(addr));
  return v;
}

#ifndef TOK_RING_FAKE_A
#define TOK_RING_FAKE_A 4      // probe builds only (-DTOK_RING_FAKE_A=n): the first n activation DMA rows of a stage are real
#endif
template <bool PW>
__global__ __launch_bounds__(256, 2) void conv_ring_kernel(ConvArgs a) {
  constexpr int NT = 256;
  constexpr int BM = RBM, BN = RBN, BK = RBK;
  constexpr int WGM = 2;
  constexpr int MT = 8;                       // 16-pixel MFMA tiles per wave (128 pixels)
  constexpr int RSTEP = NT / 4;               // 64 rows per staging pass (4 chunks of 16 B per row)
  constexpr int AROWS = BM / RSTEP;           // 4
  constexpr int WROWS = BN / RSTEP;           // 2
  constexpr int A_BYTES = BM * BK * 2;        // 16 KB
  constexpr int W_BYTES = BN * BK * 2;        // 8 KB
  constexpr int STAGE = A_BYTES + W_BYTES;    // 24 KB
  constexpr int LOADS = (TOK_RING_FAKE_A < AROWS ? TOK_RING_FAKE_A : AROWS) + WROWS;        // DMA instructions per thread and stage

  extern __shared__ __attribute__((aligned(16))) char smem[];
  typedef __attribute__((address_space(3))) char lds_char;
  typedef __attribute__((address_space(3))) void lds_void;
  const uint32_t lds_base = (uint32_t)(size_t)(lds_char*)smem;

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wv = tid >> 6;
  const int wn = wv & 1;
  const int wm = wv >> 1;
  const int kc = tid & 3;
  const int lrow = tid >> 2;                  // 0..63
  const int kcA = kc ^ ring_f(lrow >> 2);     // logical 16-byte chunk this thread fetches for its activation rows
  const int kcW = kc ^ ring_f(lrow >> 3);     // ... and for its weight rows
  const int wave_u = __builtin_amdgcn_readfirstlane(tid >> 6);

  const __amdgpu_buffer_rsrc_t xsrd = __builtin_amdgcn_make_buffer_rsrc((void*)a.x, 0, a.x_bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t wsrd = __builtin_amdgcn_make_buffer_rsrc((void*)a.w, 0, a.w_bytes, 0x00020000);

  // tile ownership (conv_igemm.hip): workgroup b on XCD b % 8, fixed channel tile, m-tiles it0, it0 + sweep, ...
  const int xcd = blockIdx.x & 7, jx = blockIdx.x >> 3;
  const int bn_fixed = jx % a.gridN, jm = jx / a.gridN;
  const int S8 = (gridDim.x >> 3) / a.gridN;
  const int sweep = 8 * S8;
  const int n0 = bn_fixed * BN;
  const int it0 = xcd * S8 + jm;
  const int ntiles = it0 < a.gridM ? (a.gridM - it0 + sweep - 1) / sweep : 0;
  const int KT = a.KT;
  const int total = ntiles * KT;

  // ---- loader state: rows of the tile whose stages are being issued, K cursor of this thread's chunk -----------------
  int h0[AROWS], w0[AROWS];
  int pixb[AROWS];                            // BYTE offset of tap (0,0), chunk 0 of the row (may be negative: padding)
  int itL = it0, ktL = 0;
  int kr = 0, ks = 0, kc0 = 0;                // tap / channel of logical chunk kcA at stage ktL
  int wk = 0;                                 // K index of logical chunk kcW at stage ktL

  auto setup_rows = [&](int it) {
    const int m0 = it * BM;
#pragma unroll
    for (int i = 0; i < AROWS; ++i) {
      const int m = m0 + lrow + RSTEP * i;
      if (PW) {
        h0[i] = (it < a.gridM && m < a.M) ? 0 : -1;
        w0[i] = 0;
        pixb[i] = (int)((uint32_t)m * (uint32_t)a.C * 2u);
      } else if (it < a.gridM && m < a.M) {
        const uint32_t b = fdiv(m, a.fd_pq);
        const uint32_t rem = m - b * a.PQ;
        const uint32_t p = fdiv(rem, a.fd_q);
        const uint32_t q = rem - p * a.Q;
        h0[i] = (int)p * a.stride - a.pad;
        w0[i] = (int)q * a.stride - a.pad;
        pixb[i] = (int)((((b * (uint32_t)a.H + (uint32_t)h0[i]) * (uint32_t)a.W + (uint32_t)w0[i]) * (uint32_t)a.C) * 2u);
      } else {
        h0[i] = -0x40000000; w0[i] = 0; pixb[i] = 0;
      }
    }
    if (PW) {
      kc0 = kcA * 8;
    } else {
      const int k0 = kcA * 8;                 // < 32
      const int tap = k0 / a.C;
      kc0 = k0 - tap * a.C;
      kr = tap / a.S;
      ks = tap - kr * a.S;
    }
    wk = kcW * 8;
  };

  auto issue = [&](int slot) {
    char* Adst = smem + slot * STAGE + wave_u * 1024;
    char* Wdst = smem + slot * STAGE + A_BYTES + wave_u * 1024;
    if (PW) {
      const bool kok = kc0 < a.C;
#pragma unroll
      for (int i = 0; i < AROWS; ++i) {
        uint32_t off = (kok && h0[i] == 0) ? (uint32_t)(pixb[i] + kc0 * 2) : 0xFFFFFFF0u;
        asm volatile("" : "+v"(off));
        __builtin_amdgcn_raw_ptr_buffer_load_lds(xsrd, (lds_void*)(Adst + i * (RSTEP * 64)), 16, off, 0, 0, 0);
      }
    } else {
      const bool kok = kr < a.R;
      const int tap_delta = ((kr * a.W + ks) * a.C + kc0) * 2;
#pragma unroll
      for (int i = 0; i < AROWS; ++i) {
        if (i >= TOK_RING_FAKE_A) continue;
        const bool ok = kok && ((unsigned)(h0[i] + kr) < (unsigned)a.H) && ((unsigned)(w0[i] + ks) < (unsigned)a.W);
        uint32_t off = ok ? (uint32_t)(pixb[i] + tap_delta) : 0xFFFFFFF0u;
        asm volatile("" : "+v"(off));
        __builtin_amdgcn_raw_ptr_buffer_load_lds(xsrd, (lds_void*)(Adst + i * (RSTEP * 64)), 16, off, 0, 0, 0);
      }
    }
    const bool wok = itL < a.gridM && wk < a.Ktot;
#pragma unroll
    for (int j = 0; j < WROWS; ++j) {
      const int n = n0 + lrow + RSTEP * j;
      uint32_t woff = (uint32_t)(n * a.Ktot + wk) * 2u;
      asm volatile("" : "+v"(woff));
      uint32_t off = (wok && n < a.K) ? woff : 0xFFFFFFF0u;
      asm volatile("" : "+v"(off));
      __builtin_amdgcn_raw_ptr_buffer_load_lds(wsrd, (lds_void*)(Wdst + j * (RSTEP * 64)), 16, off, 0, 0, 0);
    }
    // advance the cursor by one stage
    if (++ktL == KT) {
      ktL = 0;
      itL += sweep;
      setup_rows(itL);
    } else {
      wk += BK;
      kc0 += BK;
      if (!PW) {
        while (kc0 >= a.C) {
          kc0 -= a.C;
          if (++ks == a.S) { ks = 0; ++kr; }
        }
      }
    }
  };

  // ---- fragment addressing ---------------------------------------------------------------------------------------------
  const int sl = lane >> 4;
  const int li = lane & 15;
  const int fswz = ring_f(li >> 2);           // both operands: (row >> 2) & 3 == li >> 2 (activations), (row >> 3) & 3 == li >> 2 (weights)
  const int wrow0 = wn * 64 + (li >> 2) * 8 + (li & 3);
  const int arow0 = wm * (MT * 16) + li;
  const uint32_t fchunk = (uint32_t)((sl ^ fswz) << 4);

  // BatchNorm partial sums: a tile's 16 per-lane sums live only inside its epilogue, are folded over the 16 pixel lanes there
  // (recursive halving: lane li ends up owning channel li of its group) and added to these two registers — the 128
  // accumulators leave no room for 32 more registers across the main loop
  float s1r = 0.f, s2r = 0.f;

  f32x4 acc[4][MT];
  auto zero_acc = [&]() {
#pragma unroll
    for (int t = 0; t < 4; ++t)
#pragma unroll
      for (int mt = 0; mt < MT; ++mt) acc[t][mt] = (f32x4){0.f, 0.f, 0.f, 0.f};
  };

  auto compute = [&](uint32_t sb) {
    const uint32_t Ab = sb + fchunk, Wb = sb + A_BYTES + fchunk;
    u32x4 wf[4], af[MT];
#pragma unroll
    for (int t = 0; t < 4; ++t) wf[t] = rlds16(Wb + (wrow0 + (t >> 1) * 32 + (t & 1) * 4) * 64);
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) af[mt] = rlds16(Ab + (arow0 + mt * 16) * 64);
    asm volatile("s_waitcnt lgkmcnt(4)" ::: "memory");      // weights + the first four pixel tiles have landed
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int mt = 0; mt < 4; ++mt)
#pragma unroll
      for (int t = 0; t < 4; ++t)
        acc[t][mt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, wf[t]), __builtin_bit_cast(bf16x8, af[mt]),
                                                             acc[t][mt], 0, 0, 0);
    __builtin_amdgcn_sched_barrier(0);
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int mt = 4; mt < MT; ++mt)
#pragma unroll
      for (int t = 0; t < 4; ++t)
        acc[t][mt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, wf[t]), __builtin_bit_cast(bf16x8, af[mt]),
                                                             acc[t][mt], 0, 0, 0);
  };

  // epilogue of one finished tile: lane (sl, li) holds channels nb + {0..7} and nb + 32 + {0..7} of pixels m0 + arow0 + 16 mt
  auto epilogue = [&](int it) {
    const int m0 = it * BM;
    const int nb = n0 + wn * 64 + sl * 8;
    if (a.bias != nullptr) {
#pragma unroll
      for (int c = 0; c < 16; ++c) {
        const int n = nb + (c >> 3) * 32 + (c & 7);
        const float bv = n < a.K ? a.bias[n] : 0.f;
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) acc[c >> 2][mt][c & 3] += bv;
      }
    }
    const bool want_old = a.accumulate != 0;
    const bool want_y = a.stats != nullptr && !a.mask_store && a.bn_y != nullptr;
    const bool want_bits = a.bn_mask != nullptr && (a.mask_store || want_y);
    float s1[16], s2[16];
#pragma unroll
    for (int c = 0; c < 16; ++c) { s1[c] = 0.f; s2[c] = 0.f; }
    // two pixel tiles at a time: every global operand of the pair (previous value, producer's raw output, ReLU bits) is
    // requested before the first is consumed
#pragma unroll
    for (int g = 0; g < MT; g += 2) {
      bf16x8 pre_old[2][2], pre_y[2][2];
      unsigned pre_bits[2][2];
#pragma unroll
      for (int q = 0; q < 2; ++q)
#pragma unroll
        for (int half = 0; half < 2; ++half) {
          const int m = m0 + arow0 + (g + q) * 16;
          const bool ok = m < a.M && nb + half * 32 + 8 <= a.K;
          const size_t eoff = (size_t)m * a.K + nb + half * 32;
          pre_old[q][half] = (ok && want_old) ? ldg16(a.y + eoff) : zero8();
          pre_y[q][half] = (ok && want_y) ? ldg16(a.bn_y + eoff) : zero8();
          pre_bits[q][half] = (ok && want_bits) ? (unsigned)a.bn_mask[eoff >> 3] : 0xffu;
        }
#pragma unroll
      for (int q = 0; q < 2; ++q) {
        const int mt = g + q;
        const int m = m0 + arow0 + mt * 16;
        if (m >= a.M) continue;
        bf16* yp = a.y + (size_t)m * a.K + nb;
#pragma unroll
        for (int half = 0; half < 2; ++half) {
          if (nb + half * 32 + 8 > a.K) continue;
          float v[8];
#pragma unroll
          for (int e = 0; e < 8; ++e) v[e] = acc[half * 2 + (e >> 2)][mt][e & 3];
          if (a.accumulate) {
#pragma unroll
            for (int e = 0; e < 8; ++e) v[e] += bf2f(pre_old[q][half][e]);
          }
          bf16x8 o;
#pragma unroll
          for (int e = 0; e < 8; ++e) o[e] = f2bf(v[e]);
          if (a.mask_store) {
            const unsigned bits = pre_bits[q][half];
#pragma unroll
            for (int e = 0; e < 8; ++e) {
              if (!((bits >> e) & 1u)) o[e] = (bf16)0.f;
              s1[half * 8 + e] += bf2f(o[e]);
            }
          }
          stg16(yp + half * 32, o);
          if (a.stats != nullptr && !a.mask_store) {
            if (a.bn_y != nullptr) {
              const unsigned bits = pre_bits[q][half];
#pragma unroll
              for (int e = 0; e < 8; ++e) {
                const float dz = ((bits >> e) & 1u) ? bf2f(o[e]) : 0.f;
                s1[half * 8 + e] += dz;
                s2[half * 8 + e] = fmaf(dz, bf2f(pre_y[q][half][e]), s2[half * 8 + e]);
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
    }
    if (a.stats != nullptr) {
      // recursive halving over the 16 pixel-lanes: after the 4 steps lane li owns channel li of its group
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
      s1r += s1[0];
      s2r += s2[0];
    }
  };

  // ---- the ring ---------------------------------------------------------------------------------------------------------
  setup_rows(itL);
  zero_acc();
  issue(0);
  issue(1);
  int cur = 0, nxt = 2;
  int itC = it0, ktC = 0;
  for (int s = 0; s < total; ++s) {
    asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)" ::"n"(LOADS) : "memory");   // this thread's share of stage s has landed
    __builtin_amdgcn_s_barrier();                                              // everyone's has; nobody reads stage s-1 any more
    issue(nxt);
    compute(lds_base + cur * STAGE);
    if (ktC == KT - 1) {
      epilogue(itC);
      zero_acc();
      ktC = 0;
      itC += sweep;
    } else {
      ++ktC;
    }
    cur = cur == RNST - 1 ? 0 : cur + 1;
    nxt = nxt == RNST - 1 ? 0 : nxt + 1;
  }
  asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");

  // ---- BatchNorm partial sums -> one row per workgroup (conv_igemm.hip) --------------------------------------------------
  if (a.stats != nullptr) {
    __syncthreads();
    float* red = reinterpret_cast<float*>(smem);     // [2][WGM][BN]: the ring is drained
    const int nl = wn * 64 + (li >> 3) * 32 + sl * 8 + (li & 7);
    red[(0 * WGM + wm) * BN + nl] = s1r;
    red[(1 * WGM + wm) * BN + nl] = s2r;
    __syncthreads();
    if (tid < 2 * BN) {
      const int which = tid / BN;
      const int c = tid - which * BN;
      float t = 0.f;
#pragma unroll
      for (int w_ = 0; w_ < WGM; ++w_) t += red[(which * WGM + w_) * BN + c];
      const int row = xcd * S8 + jm;
      const int n = n0 + c;
      if (n < a.K) a.stats[((size_t)which * a.stat_rows + row) * a.K + n] = t;
    }
  }
}

// TOK_CONV_RING=1 routes the deep-K layers here.  OFF by default: measured on MI355X (round 3, tools/ubench/exp_ring.sh,
// exp_win2.sh) the 256 x 128 ring ties the two-buffer 128 x 128 kernel — per layer 0.85-1.17x (3x3 at 14x14: 81.6 -> 78.7 /
// 80.3 -> 68.7 us, but 28x28: 77.5 -> 88.8 and Swin's 1536 -> 384: 76 -> 107 us, 588 tiles on 512 slots), ResNet-50 step
// 18.29 -> 18.37 ms, SwinV2-T 24.70 -> 25.03 — because both are bound by the L2 -> LDS path, not by the wave tile: with ONE of
// the four activation DMA rows real the ring runs 1.2-1.3x faster.  That probe is what conv_win.hip (shared input window for
// the 3x3 layers) is built on; this kernel stays as the tested basis for larger pointwise tiles.
int ring_flag() {
  static const int v = [] { const char* e = getenv("TOK_CONV_RING"); return (int)(e ? atoi(e) : 0); }();
  return v;
}
int ring_min_k() {  // reduction depth from which the 256 x 128 ring serves a layer
  static const int v = [] { const char* e = getenv("TOK_CONV_RING_MIN_K"); return (int)(e ? atoi(e) : 256); }();
  return v;
}
int ring_min_tiles() {   // fewer 256 x 128 tiles than this cannot fill the chip: the smaller tiles of conv_igemm serve the layer
  static const int v = [] { const char* e = getenv("TOK_CONV_RING_MIN_TILES"); return (int)(e ? atoi(e) : 384); }();
  return v;
}

}  // namespace

bool conv_ring_serves(const ConvArgs& a, bool pointwise) {
  if (!ring_flag()) return false;
  if (a.C % 8 != 0 || a.K % 8 != 0 || a.K < 128) return false;
  if (a.y2 != nullptr || a.act_x != nullptr || a.ep_scale != nullptr || a.sub != nullptr || a.fin_mode != 0) return false;
  if (a.Ktot < ring_min_k()) return false;
  const long long tiles = (long long)tok_cdiv(a.M, RBM) * tok_cdiv(a.K, RBN);
  if (tiles < ring_min_tiles()) return false;
  (void)pointwise;
  return true;
}

int conv_ring_grid(int gridM256, int gridN128) {
  const int unit = 8 * gridN128;
  int G = 512;                                    // two workgroups per CU
  const long long need = (long long)gridM256 * gridN128;
  if (need < G) G = (int)((need + unit - 1) / unit) * unit;
  G = G / unit * unit;
  if (G < unit) G = unit;
  return G;
}

int conv_ring_launch(ConvArgs& a, hipStream_t st) {
  constexpr int smem = RNST * (RBM + RBN) * RBK * 2;
  const bool pw = a.R == 1 && a.S == 1 && a.stride == 1 && a.pad == 0;
  static const bool attr_set = [] {   // once per process (thread-safe function-local static), both instantiations
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&conv_ring_kernel<true>), hipFuncAttributeMaxDynamicSharedMemorySize, smem);
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&conv_ring_kernel<false>), hipFuncAttributeMaxDynamicSharedMemorySize, smem);
    return true;
  }();
  (void)attr_set;
  a.KT = tok_cdiv(a.Ktot, RBK);
  const int grid = conv_ring_grid(a.gridM, a.gridN);
  a.stat_rows = grid / a.gridN;
  if (pw) hipLaunchKernelGGL((conv_ring_kernel<true>), dim3(grid), dim3(256), smem, st, a);
  else hipLaunchKernelGGL((conv_ring_kernel<false>), dim3(grid), dim3(256), smem, st, a);
  return 0;
}

#else   // default build: the kernel is not compiled; nothing is ever served by it

bool conv_ring_serves(const ConvArgs&, bool) { return false; }
int conv_ring_grid(int, int) { return 0; }
int conv_ring_launch(ConvArgs&, hipStream_t) {
  tok_set_error("conv_ring: not built (compile with TOK_BUILD_EXPERIMENTS=1)");
  return TOK_ERR_INVALID;
}

#endif
