// Implicit-GEMM convolution on MFMA for gfx950: forward and data-gradient.
//
//   Y[m][n] = sum_k A[m][k] * Wt[n][k]      m = (b, p, q)   k = (r, s, c)   NHWC bf16
//
// One kernel serves both directions:
//   fwd   : A gathered from x at (p*stride - pad + r, q*stride - pad + s)
//   dgrad : "x" is dY, the weights are the flipped/transposed pack ([C][R][S][K]), the gather
//           runs at stride 1 / pad R-1-pad.  For stride-2 convolutions (IN_DIV = 2) a tap
//           contributes only where (h - pad' + r') is even: output pixels are enumerated per
//           PARITY CLASS (h&1, w&1) so that a whole tile shares the parity and K steps whose tap
//           cannot contribute are skipped outright (no loads, no MFMAs) instead of multiplying
//           zeros — a 1x1/s2 dgrad touches 1 class in 4, a 3x3/s2 one runs 1+2+2+4 of 36 taps.
//
// Tile: BM=128 output pixels x BN in {64,128} channels x BK=64, 256 threads (4 waves),
// mfma_f32_16x16x32_bf16.  The MFMA "A" operand is the WEIGHT tile and the "B" operand the
// activation tile, so D[i=channel][j=pixel].  Weight rows are fed in the order
//   n = (t>>1)*32 + (i>>2)*8 + (t&1)*4 + (i&3)          (t = 16-row MFMA tile, i = MFMA row)
// which leaves lane (g = lane>>4) with channels {h*32 + g*8 + e}: the epilogue is two 16-byte
// stores per pixel and the four g-lanes of a pixel cover 64 contiguous bytes per instruction.
// BatchNorm partial sums: recursive-halving butterfly over the 16 pixel-lanes (15 shuffles per
// quantity; lane li ends up owning channel li of its group).
//
// Persistent workgroups: each workgroup walks tiles b, b+G, ...; the first K-step loads of the
// NEXT tile are issued before the epilogue of the current one, so HBM loads stay in flight while
// the stores drain (the small-K layers of ResNet are a streaming problem, not a GEMM problem).
// Staging: global -> registers -> LDS (two buffers, one barrier per K step), 16-byte
// XOR-swizzled slots so every ds_read_b128 of a fragment is bank-conflict free.
#include "conv_common.h"
#include "pw_gemm.h"
#include <stdlib.h>

namespace {

constexpr int BK = 64;
#ifdef TOK_NO_FRAG_ASM
constexpr bool FRAG_ASM = false;
#else
constexpr bool FRAG_ASM = true;
#endif
#ifdef TOK_NO_DMA
constexpr bool DMA_ENABLED = false;
#else
constexpr bool DMA_ENABLED = true;
#endif

__device__ __forceinline__ int fw_swz(int n) {
  // weight-tile slot swizzle.  One ds_read_b128 lane group reads rows n = c0 + 8q + i (q,i in
  // 0..3): h = [0,2,3,1] separates the q's that share a lane group, bit 2 separates i>>1, and
  // (i&1) lands in the other half of the 256-byte bank row.
  return ((0x78 >> (((n >> 3) & 3) << 1)) & 3) | (((n >> 1) & 1) << 2);
}

template <int BM, int BN, int IN_DIV, bool C4, int PWM>
__global__ __launch_bounds__(256, (BN == 64 && PWM != 4) ? 3 : 2) void conv_igemm_kernel(ConvArgs a) {
  constexpr bool PW = PWM != 0;    // PWM 1: pointwise; 2: pointwise with streaming (non-temporal) stores
  constexpr bool NTS = PWM == 2;
  constexpr bool ACT = PWM == 3;   // pointwise with a fused activation epilogue (forward: second output; dgrad: * act')
  constexpr bool BNEP = PWM == 4;  // pointwise forward with the BatchNorm + shortcut + ReLU epilogue
  // DMA: global -> LDS directly (buffer_load ... lds), no staging registers and no ds_write pass
  // (LDS stores run at ~80 B/clk/CU: the register-staged loop was LDS-write bound on deep-K layers).
  // The LDS image of a wave instruction is lane-linear, so the XOR swizzle is applied to the SOURCE:
  // the lane sitting at slot kc fetches logical chunk kc ^ swz(row).
  constexpr bool DMA = !C4 && DMA_ENABLED;
  // PW: pointwise fast path (1x1, stride 1, no padding, forward or dgrad): a pixel row of the A
  // matrix is simply x[m][:], so the per-tile / per-K-step address arithmetic collapses (these
  // layers are VALU-issue bound, not HBM bound, when done the general way).
  // 4 waves.  128x64 tile (the streaming layers): 4 waves along m, a wave owns 32 pixels x 64
  // channels (8 accumulator tiles), <= 168 registers -> 3 workgroups per CU.  128x128 tile (the
  // deep-K, MFMA-bound layers): 2 x 2 waves, a wave owns 64 x 64 (16 accumulator tiles: 16 MFMAs
  // per 8 fragment reads), 2 workgroups per CU.
  constexpr int NT = 256;
  constexpr int WGN = BN / 64;
  constexpr int WGM = (NT / 64) / WGN;
  constexpr int MT = BM / (WGM * 16);
  constexpr int RSTEP = NT / 8;             // rows covered by one staging pass
  constexpr int AROWS = BM / RSTEP;
  constexpr int WROWS = BN / RSTEP;
  constexpr int TILE_BYTES = (BM + BN) * BK * 2;

  extern __shared__ __attribute__((aligned(16))) char smem[];
  float* red = reinterpret_cast<float*>(smem + 2 * TILE_BYTES);  // [2][WGM][BN]
  // BNEP: the shortcut tile (BM pixels x BN channels, the layout and swizzle of the activation tile) is staged by DMA
  // together with the FIRST K step of its output tile — one whole tile ahead of the epilogue that adds it; two buffers
  constexpr int S_BYTES = BM * BN * 2;
  char* sbase = smem + 2 * TILE_BYTES + 2 * 4 * BN * 4;

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wv = tid >> 6;
  const int wn = wv % WGN;
  const int wm = wv / WGN;
  const int kc = tid & 7;
  const int lrow = tid >> 3;
  // logical 16-byte k-chunk this thread fetches for the activation / weight tile
  const int kcA = DMA ? (kc ^ (lrow & 7)) : kc;
  const int kcW = DMA ? (kc ^ fw_swz(lrow)) : kc;
  const int wave_u = __builtin_amdgcn_readfirstlane(tid >> 6);

  const __amdgpu_buffer_rsrc_t xsrd = __builtin_amdgcn_make_buffer_rsrc((void*)a.x, 0, a.x_bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t wsrd = __builtin_amdgcn_make_buffer_rsrc((void*)a.w, 0, a.w_bytes, 0x00020000);

  // ---- loader state (belongs to the tile whose loads are being issued) ----------------------
  int h0[AROWS], w0[AROWS], pix[AROWS];   // pix: pixel base (C4 / IN_DIV 2) or ELEMENT base of tap (0,0)
  int spix[AROWS];                        // BNEP: element base of the row in the shortcut tensor (m * K)
  int s_ld = 0, s_rd = 0;                 // BNEP: shortcut buffer being filled / being consumed
  int kr, ks, kc0, kt;
  size_t wk;
  int par_h = 0, par_w = 0;  // (ph - pad), (pw - pad) of the loader's tile (IN_DIV == 2)
  bf16x8 ra[AROWS], rw[WROWS];

  struct Ctx { int m0, n0, bm, cls; };

  // Tile ownership: workgroup b sits on XCD b % 8.  Inside an XCD the index j = b / 8 splits into
  // (channel tile bn = j % gridN, m-slot jm = j / gridN); in sweep i the XCD owns the S8 consecutive
  // m-tiles [i*S + xcd*S8, +S8): neighbouring pixel tiles (3x3 halos) share that XCD's L2, all
  // channel tiles of one pixel tile run together, and bn never changes for a workgroup (its
  // BatchNorm partial sums stay in registers across tiles).
  const int xcd = blockIdx.x & 7, jx = blockIdx.x >> 3;
  const int bn_fixed = jx % a.gridN, jm = jx / a.gridN;
  const int S8 = (gridDim.x >> 3) / a.gridN;
  const int sweep = 8 * S8;

  auto setup = [&](int bm) -> Ctx {
    Ctx c;
    c.bm = bm;
    c.n0 = bn_fixed * BN;
    c.cls = 0;
    if (IN_DIV == 2) {
      // class of m-tile bm, rotated per sweep so no workgroup keeps drawing the heavy class
      c.cls = ((bm & 3) + (bm >> 2) / (sweep >> 2)) & 3;
      c.m0 = (bm >> 2) * BM;
      const int ph = c.cls >> 1, pw = c.cls & 1;
      par_h = ph - a.pad;
      par_w = pw - a.pad;
#pragma unroll
      for (int i = 0; i < AROWS; ++i) {
        const int ml = c.m0 + lrow + RSTEP * i;
        if (ml < a.cls_M[c.cls]) {
          const uint32_t b = fdiv(ml, a.cls_fd_hw[c.cls]);
          const uint32_t rem = ml - b * a.cls_hw[c.cls];
          const uint32_t h2 = fdiv(rem, a.cls_fd_w[c.cls]);
          const uint32_t w2 = rem - h2 * a.cls_nw[c.cls];
          h0[i] = 2 * (int)h2 + par_h;
          w0[i] = 2 * (int)w2 + par_w;
          pix[i] = b * a.H * a.W;
        } else {
          h0[i] = -0x40000000; w0[i] = 0; pix[i] = 0;
        }
      }
    } else {
      c.m0 = c.bm * BM;
#pragma unroll
      for (int i = 0; i < AROWS; ++i) {
        const int m = c.m0 + lrow + RSTEP * i;
        if (PW) {
          h0[i] = m < a.M ? 0 : -1;
          w0[i] = 0;
          pix[i] = m * a.C;
          if (BNEP) spix[i] = m * a.K;
        } else if (m < a.M) {
          const uint32_t b = fdiv(m, a.fd_pq);
          const uint32_t rem = m - b * a.PQ;
          const uint32_t p = fdiv(rem, a.fd_q);
          const uint32_t q = rem - p * a.Q;
          h0[i] = (int)p * a.stride - a.pad;
          w0[i] = (int)q * a.stride - a.pad;
          pix[i] = C4 ? (int)(b * a.H * a.W) : (int)(((b * a.H + h0[i]) * a.W + w0[i]) * a.C);
        } else {
          h0[i] = -0x40000000; w0[i] = 0; pix[i] = 0;
        }
      }
    }
    // k-chunk cursor of this thread at K step 0
    if (C4) {
      kr = kc >> 2; ks = (kc & 3) << 1; kc0 = 0;
    } else if (PW) {
      kc0 = kcA * 8; kr = kc0 >= a.C ? 1 : 0; ks = 0;   // C >= 8; a chunk past C is past the K extent
    } else {
      const int k0 = kcA * 8;
      const int tap = k0 / a.C;   // k0 < 64: at most a handful of taps
      kc0 = k0 - tap * a.C;
      kr = tap / a.S;
      ks = tap - kr * a.S;
    }
    wk = (size_t)kcW * 8;
    kt = 0;
    return c;
  };

  auto advance = [&]() {
    wk += BK;
    if (C4) {
      kr += 2;
    } else if (PW) {
      kc0 += BK;
      kr = kc0 >= a.C ? 1 : 0;
    } else {
      kc0 += BK;
      while (kc0 >= a.C) {
        kc0 -= a.C;
        if (++ks == a.S) { ks = 0; ++kr; }
      }
    }
  };

  // stride-2 dgrad: skip K steps whose (uniform) tap has the wrong parity for this tile
  auto seek = [&]() {
    if (IN_DIV == 2) {
      if (a.uniform_taps) {
        while (kt < a.KT) {
          const int tap = (kt * BK) / a.C;
          const int r = tap / a.S, s = tap - r * a.S;
          if ((((par_h + r) | (par_w + s)) & 1) == 0) break;
          advance();
          ++kt;
        }
      }
    }
  };

  bool ld_on = true;   // false: issue the same loads with out-of-range offsets (zeros, no traffic)
  auto load_tile = [&](int dbuf) {
    const bool kvalid = ld_on && kr < a.R;
    typedef __attribute__((address_space(3))) void lds_void;
    char* Adst = smem + dbuf * TILE_BYTES + wave_u * 1024;
    char* Wdst = Adst + BM * BK * 2;
    const int tap_delta = (kr * a.W + ks) * a.C + kc0;
#pragma unroll
    for (int i = 0; i < AROWS; ++i) {
      int hh = h0[i] + kr;
      int ww = w0[i] + ks;
      bool ok = kvalid;
      if (IN_DIV == 2) {
        ok = ok && (((hh | ww) & 1) == 0);
        hh >>= 1;
        ww >>= 1;
      }
      ok = ok && ((unsigned)hh < (unsigned)a.H);
      if (C4) {
        // two 8-byte pixels (s, s+1 of the padded filter row) as unconditional buffer loads: a tap outside the image takes
        // an out-of-range offset and comes back as zeros (the branchy form put an s_waitcnt in front of every load)
        const uint32_t base = (uint32_t)(pix[i] + hh * a.W + ww) * 8u;
        const uint32_t off_lo = (ok && (unsigned)ww < (unsigned)a.W) ? base : 0xFFFFFFF0u;
        const uint32_t off_hi = (ok && (unsigned)(ww + 1) < (unsigned)a.W) ? base + 8u : 0xFFFFFFF0u;
        const u32x2 lo = __builtin_amdgcn_raw_buffer_load_b64(xsrd, off_lo, 0, 0);
        const u32x2 hi = __builtin_amdgcn_raw_buffer_load_b64(xsrd, off_hi, 0, 0);
        const u32x4 both = {lo[0], lo[1], hi[0], hi[1]};
        ra[i] = __builtin_bit_cast(bf16x8, both);
      } else {
        // buffer load: padding taps / rows past M / K tail use an out-of-range offset, which the
        // buffer unit answers with zeros — one unconditional instruction per row, no zero-fill VALU
        // write for hipcc to fence with a vmcnt(0) (that fence used to serialise these loads behind
        // the previous tile's stores)
        uint32_t off;
        if (IN_DIV == 2) {
          ok = ok && ((unsigned)ww < (unsigned)a.W);
          off = ok ? (uint32_t)((pix[i] + hh * a.W + ww) * a.C + kc0) * 2u : 0xFFFFFFF0u;
        } else if (PW) {
          off = (kvalid && h0[i] == 0) ? (uint32_t)(pix[i] + kc0) * 2u : 0xFFFFFFF0u;
        } else {
          // pix[i] = element offset of tap (0,0); the tap displacement is the same for all rows
          ok = ok && ((unsigned)ww < (unsigned)a.W);
          off = ok ? (uint32_t)(pix[i] + tap_delta) * 2u : 0xFFFFFFF0u;
        }
        if (DMA) {
          // (value barrier: without it hipcc turns the select above into a branch around the load — eight branches per K step)
          asm volatile("" : "+v"(off));
          __builtin_amdgcn_raw_ptr_buffer_load_lds(xsrd, (lds_void*)(Adst + i * (RSTEP * 128)), 16, off, 0, 0, 0);
        } else {
          ra[i] = __builtin_bit_cast(bf16x8, __builtin_amdgcn_raw_buffer_load_b128(xsrd, off, 0, 0));
        }
      }
    }
    if (BNEP) {
      if (kt == 0 && a.ep_short != nullptr) {     // wave-uniform: first K step of an output tile
        char* Sdst = sbase + s_ld * S_BYTES + wave_u * 1024;
        const __amdgpu_buffer_rsrc_t ssrd = __builtin_amdgcn_make_buffer_rsrc((void*)a.ep_short, 0, a.s_bytes, 0x00020000);
#pragma unroll
        for (int i = 0; i < AROWS; ++i) {
          const int n = bn_fixed * BN + kcA * 8;
          uint32_t off = (ld_on && h0[i] == 0 && n < a.K) ? (uint32_t)(spix[i] + n) * 2u : 0xFFFFFFF0u;
          asm volatile("" : "+v"(off));
          __builtin_amdgcn_raw_ptr_buffer_load_lds(ssrd, (lds_void*)(Sdst + i * (RSTEP * 128)), 16, off, 0, 0, 0);
        }
        s_ld ^= 1;
      }
    }
#pragma unroll
    for (int j = 0; j < WROWS; ++j) {
      const int n = bn_fixed * BN + lrow + RSTEP * j;
      const bool wvalid = DMA ? (ld_on && (int)wk < a.Ktot) : kvalid;
      uint32_t woff = (uint32_t)(n * a.Ktot + (int)wk) * 2u;
      if (DMA) asm volatile("" : "+v"(woff));     // offset computed unconditionally: the select stays a v_cndmask
      uint32_t off = (wvalid && n < a.K) ? woff : 0xFFFFFFF0u;
      if (DMA) {
        asm volatile("" : "+v"(off));
        __builtin_amdgcn_raw_ptr_buffer_load_lds(wsrd, (lds_void*)(Wdst + j * (RSTEP * 128)), 16, off, 0, 0, 0);
      }
      else rw[j] = __builtin_bit_cast(bf16x8, __builtin_amdgcn_raw_buffer_load_b128(wsrd, off, 0, 0));
    }
  };

  auto store_tile = [&](int buf) {
    if (DMA) return;   // the data is already on its way into LDS; hipcc drains vmcnt before the barrier
    char* Ab = smem + buf * TILE_BYTES;
    char* Wb = Ab + BM * BK * 2;
#pragma unroll
    for (int i = 0; i < AROWS; ++i) {
      const int row = lrow + RSTEP * i;
      *reinterpret_cast<bf16x8*>(Ab + row * 128 + ((kc ^ (row & 7)) << 4)) = ra[i];
    }
#pragma unroll
    for (int j = 0; j < WROWS; ++j) {
      const int row = lrow + RSTEP * j;
      *reinterpret_cast<bf16x8*>(Wb + row * 128 + ((kc ^ fw_swz(row)) << 4)) = rw[j];
    }
  };

  // ---- fragment addressing (lane constants) -------------------------------------------------------
  const int sl = lane >> 4;
  const int li = lane & 15;
  // weight row of MFMA tile t fed by this lane: wn*64 + (t>>1)*32 + (li>>2)*8 + (t&1)*4 + (li&3)
  const int wrow0 = wn * 64 + (li >> 2) * 8 + (li & 3);
  const int wswz = fw_swz(wrow0);
  const int arow0 = wm * (MT * 16) + li;
  const int aswz = li & 7;

  float s1[16], s2[16];   // per-lane BatchNorm partial sums of this workgroup's channel tile
#pragma unroll
  for (int c = 0; c < 16; ++c) { s1[c] = 0.f; s2[c] = 0.f; }
  float esc[16], esh[16];   // BNEP: scale / shift of this lane's 16 channels (the channel tile never changes)
  if (BNEP) {
#pragma unroll
    for (int c = 0; c < 16; ++c) {
      const int n = bn_fixed * BN + wn * 64 + sl * 8 + (c >> 3) * 32 + (c & 7);
      esc[c] = n < a.K ? a.ep_scale[n] : 0.f;
      esh[c] = n < a.K ? a.ep_shift[n] : 0.f;
    }
  }

  f32x4 acc[4][MT];
  auto zero_acc = [&]() {
#pragma unroll
    for (int t = 0; t < 4; ++t)
#pragma unroll
      for (int mt = 0; mt < MT; ++mt) acc[t][mt] = (f32x4){0.f, 0.f, 0.f, 0.f};
  };

  typedef __attribute__((address_space(3))) char lds_char_t;
  const uint32_t lds_base = (uint32_t)(size_t)(lds_char_t*)smem;
  auto lds_read16 = [](uint32_t addr) -> u32x4 {
    u32x4 v;
    asm volatile("ds_read_b128 %0, %1" : "=v"(v) : "v"(addr));
    return v;
  };
  auto compute = [&](int buf) {
    if constexpr (DMA && FRAG_ASM) {   // (the register-staged stem keeps hipcc's own schedule: measured 294 -> 494 us with this form)
      // all 2 x (4 + MT) fragment reads of the K step go out first (inline asm: hipcc interleaves its own ds_reads with the
      // MFMAs one wait at a time, and fences them with vmcnt(0) while the next stage's DMA is in flight); the first half's
      // MFMAs start when ITS fragments have landed, the second half's reads complete underneath
      const uint32_t Ab = lds_base + buf * TILE_BYTES, Wb = Ab + BM * BK * 2;
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
          acc[t][mt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, wf[0][t]),
                                                               __builtin_bit_cast(bf16x8, af[0][mt]), acc[t][mt], 0, 0, 0);
      __builtin_amdgcn_sched_barrier(0);   // (the wait must not float above the first half's MFMAs)
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int t = 0; t < 4; ++t)
#pragma unroll
        for (int mt = 0; mt < MT; ++mt)
          acc[t][mt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, wf[1][t]),
                                                               __builtin_bit_cast(bf16x8, af[1][mt]), acc[t][mt], 0, 0, 0);
      return;
    }
    const char* Ab = smem + buf * TILE_BYTES;
    const char* Wb = Ab + BM * BK * 2;
#pragma unroll
    for (int kk = 0; kk < 2; ++kk) {
      bf16x8 wf[4], af[MT];
      const int s = sl + 4 * kk;
#pragma unroll
      for (int t = 0; t < 4; ++t)
        wf[t] = *reinterpret_cast<const bf16x8*>(Wb + (wrow0 + (t >> 1) * 32 + (t & 1) * 4) * 128 +
                                                 ((s ^ wswz) << 4));
#pragma unroll
      for (int mt = 0; mt < MT; ++mt)
        af[mt] = *reinterpret_cast<const bf16x8*>(Ab + (arow0 + mt * 16) * 128 + ((s ^ aswz) << 4));
#pragma unroll
      for (int t = 0; t < 4; ++t)
#pragma unroll
        for (int mt = 0; mt < MT; ++mt)
          acc[t][mt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[t], af[mt], acc[t][mt], 0, 0, 0);
    }
  };

  // epilogue of one finished tile: lane (sl, li) holds channels nb + {0..7} and nb + 32 + {0..7}
  auto epilogue = [&](const Ctx& ctx) {
    const int nb = ctx.n0 + wn * 64 + sl * 8;
    if (a.bias != nullptr) {
#pragma unroll
      for (int c = 0; c < 16; ++c) {
        const int n = nb + (c >> 3) * 32 + (c & 7);
        const float bv = n < a.K ? a.bias[n] : 0.f;
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) acc[c >> 2][mt][c & 3] += bv;
      }
    }
    // output pixel of each of this lane's MT rows, then EVERY global operand of the epilogue (previous value, producer's raw
    // output, ReLU bits) requested before the first one is consumed: one exposed memory latency per tile instead of one per
    // 16-byte piece (hipcc keeps the load -> use -> store chains in program order)
    size_t opix_[MT];
    bool rowok_[MT];
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) {
      const int ml = ctx.m0 + arow0 + mt * 16;
      if (IN_DIV == 2) {
        rowok_[mt] = ml < a.cls_M[ctx.cls];
        const uint32_t mm = rowok_[mt] ? ml : 0;
        const uint32_t b = fdiv(mm, a.cls_fd_hw[ctx.cls]);
        const uint32_t rem = mm - b * a.cls_hw[ctx.cls];
        const uint32_t h2 = fdiv(rem, a.cls_fd_w[ctx.cls]);
        const uint32_t w2 = rem - h2 * a.cls_nw[ctx.cls];
        opix_[mt] = ((size_t)b * a.P + (2 * h2 + (ctx.cls >> 1))) * a.Q + (2 * w2 + (ctx.cls & 1));
      } else {
        rowok_[mt] = ml < a.M;
        opix_[mt] = (size_t)ml;
      }
    }
    bf16x8 pre_old[MT][2], pre_y[MT][2];
    unsigned pre_bits[MT][2];
    const bool want_old = a.accumulate != 0;
    const bool want_y = a.stats != nullptr && !a.mask_store && a.bn_y != nullptr;
    const bool want_bits = a.bn_mask != nullptr && (a.mask_store || want_y);
    constexpr bool PRE = !BNEP && !ACT && !C4;     // (the stem kernel is forward-only: no epilogue operands)
    if (PRE) {
#pragma unroll
      for (int mt = 0; mt < MT; ++mt)
#pragma unroll
        for (int half = 0; half < 2; ++half) {
          const bool ok = rowok_[mt] && nb + half * 32 + 8 <= a.K;
          const size_t eoff = opix_[mt] * a.K + nb + half * 32;
          pre_old[mt][half] = (ok && want_old) ? ldg16(a.y + eoff) : zero8();
          pre_y[mt][half] = (ok && want_y) ? ldg16(a.bn_y + eoff) : zero8();
          pre_bits[mt][half] = (ok && want_bits) ? (unsigned)a.bn_mask[eoff >> 3] : 0xffu;
        }
    }
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) {
      const size_t opix = opix_[mt];
      const bool rowok = rowok_[mt];
      if (rowok) {
        bf16* yp = a.y + opix * a.K + nb;
#pragma unroll
        for (int half = 0; half < 2; ++half) {
          if (nb + half * 32 + 8 <= a.K) {
            bf16x8 o;
            float v[8];
#pragma unroll
            for (int e = 0; e < 8; ++e) v[e] = acc[half * 2 + (e >> 2)][mt][e & 3];
            if (a.accumulate) {
              const bf16x8 old = PRE ? pre_old[mt][half] : ldg16(yp + half * 32);
#pragma unroll
              for (int e = 0; e < 8; ++e) v[e] += bf2f(old[e]);
            }
#pragma unroll
            for (int e = 0; e < 8; ++e) o[e] = f2bf(v[e]);
            if (ACT) {
              // same arithmetic as tok_act_fwd / tok_act_bwd on the bf16-rounded GEMM result: bit-identical to the
              // unfused pair of launches, minus one read (forward) or one write + one read (backward) of the tensor
              const size_t eoff = opix * a.K + nb + half * 32;
              if (a.act_x != nullptr) {
                const bf16x8 hx = ldg16(a.act_x + eoff);
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                  const float f = bf2f(hx[e]);
                  const float dd = a.act == 0 ? (f > 0.f ? 1.f : 0.f) : gelu_d(f);
                  o[e] = f2bf(bf2f(o[e]) * dd);
                }
              }
              if (a.y2 != nullptr) {
                bf16x8 o2;
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                  const float f = bf2f(o[e]);
                  o2[e] = f2bf(a.act == 0 ? fmaxf(f, 0.f) : gelu_f(f));
                }
                stg16(a.y2 + eoff, o2);
              }
            }
            if (BNEP) {
              // v = acc * scale + shift + shortcut, ReLU, bf16; mask bit e = (out[e] > 0) as tok_bn_act_fwd writes it
              const int n0 = nb + half * 32;
              const size_t eoff = opix * a.K + n0;
              bf16x8 sv = zero8();
              if (a.ep_short != nullptr)
                sv = *reinterpret_cast<const bf16x8*>(sbase + s_rd * S_BYTES + (arow0 + mt * 16) * 128 +
                                                      (((sl + 4 * half) ^ aswz) << 4));
              unsigned bits = 0;
#pragma unroll
              for (int e = 0; e < 8; ++e) {
                float z = fmaf(v[e], esc[half * 8 + e], esh[half * 8 + e]) + bf2f(sv[e]);
                if (a.ep_relu) z = fmaxf(z, 0.f);
                o[e] = f2bf(z);
                bits |= (bf2f(o[e]) > 0.f ? 1u : 0u) << e;
              }
              if (a.ep_mask != nullptr) a.ep_mask[eoff >> 3] = (uint8_t)bits;
            }
            if (a.mask_store) {
              const unsigned bits = PRE ? pre_bits[mt][half] : (unsigned)a.bn_mask[(opix * a.K + nb + half * 32) >> 3];
#pragma unroll
              for (int e = 0; e < 8; ++e) {
                if (!((bits >> e) & 1u)) o[e] = (bf16)0.f;
                s1[half * 8 + e] += bf2f(o[e]);
              }
            }
            if (NTS) __builtin_nontemporal_store(o, reinterpret_cast<bf16x8*>(yp + half * 32));
            else stg16(yp + half * 32, o);
            if (a.stats != nullptr && !a.mask_store) {
              if (a.bn_y != nullptr) {
                // backward statistics of the producing BatchNorm: dz = dx * relu_mask, sum dz, sum dz*y
                const size_t eoff = opix * a.K + nb + half * 32;
                const bf16x8 yv = PRE ? pre_y[mt][half] : ldg16(a.bn_y + eoff);
                const unsigned bits = PRE ? pre_bits[mt][half] : (a.bn_mask != nullptr ? (unsigned)a.bn_mask[eoff >> 3] : 0xffu);
                // (sums over the STORED bf16 values: the statistics of the tensor the next kernels read — and the rounding
                //  point of the reference's bf16 autocast, where BatchNorm's backward reduces a bf16 gradient tensor)
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                  const float dz = ((bits >> e) & 1u) ? bf2f(o[e]) : 0.f;
                  s1[half * 8 + e] += dz;
                  s2[half * 8 + e] = fmaf(dz, bf2f(yv[e]), s2[half * 8 + e]);
                }
              } else {
                // batch statistics of the bf16 conv output as stored (torch autocast: batch_norm reduces the bf16 output
                // of the convolution in fp32): mean / variance describe exactly the tensor bn_act_fwd normalises
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
    }
  };

  // ---- one software pipeline over the flattened (tile, K step) sequence of this workgroup ----------
  // iteration: issue the loads of the NEXT stage (next K step, or the first K step of the next
  // tile), run the MFMAs of the current stage out of LDS, drain a finished tile through the epilogue,
  // then park the landed registers in the other LDS buffer.  The next stage's HBM latency is hidden
  // behind compute + epilogue of the current one, also across tile seams.  (Single call site per
  // phase on purpose: duplicated bodies push hipcc into spilling inside the loop, and a scratch
  // reload's vmcnt(0) would drain the prefetch.)
  int it = xcd * S8 + jm;
  bool any = it < a.gridM;   // a workgroup without tiles still owns a (zero) statistics row
  Ctx cur = {0, bn_fixed * BN, 0, 0};
  bool cur_has = false;
  if (any) {
    cur = setup(it);
    seek();
    cur_has = kt < a.KT;
    if (cur_has) { load_tile(0); store_tile(0); }
  }
  if (DMA) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  zero_acc();
  int buf = 0;
#ifdef TOK_TIMING
  unsigned long long tacc[6] = {0, 0, 0, 0, 0, 0};
  const unsigned long long tk0 = __builtin_amdgcn_s_memtime();
#endif
  while (any) {
    bool tile_done, nxt_has, any_next = true;
    Ctx nxt = cur;
    if (cur_has) { advance(); ++kt; seek(); }
    if (kt < a.KT) {
      tile_done = false;
      nxt_has = true;
    } else {
      tile_done = true;
      it += sweep;
      any_next = it < a.gridM;
      nxt_has = false;
      if (any_next) {
        nxt = setup(it);
        seek();
        nxt_has = kt < a.KT;
      }
    }
#ifdef TOK_TIMING
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
#endif
    // load / LDS-store run unconditionally (an idle last stage loads zeros): with every iteration
    // issuing and retiring the same loads, hipcc's wait-count bookkeeping stays exact across the
    // back edge instead of falling back to "wait for everything" before the next loads
    ld_on = nxt_has;
    load_tile(buf ^ 1);
#ifdef TOK_TIMING
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
#endif
    if (cur_has) compute(buf);
#ifdef TOK_TIMING
    __builtin_amdgcn_sched_barrier(0);
    const unsigned long long t2 = __builtin_amdgcn_s_memtime();
#endif
    if (tile_done) {
      epilogue(cur);
      zero_acc();
      if (BNEP) s_rd ^= 1;
    }
#ifdef TOK_TIMING
    __builtin_amdgcn_sched_barrier(0);
    const unsigned long long t3 = __builtin_amdgcn_s_memtime();
#endif
    store_tile(buf ^ 1);
#ifdef TOK_TIMING
    __builtin_amdgcn_sched_barrier(0);
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    const unsigned long long t4 = __builtin_amdgcn_s_memtime();
#endif
    if (DMA) {
      __builtin_amdgcn_sched_barrier(0);                         // (MFMAs / epilogue stay in front of the wait)
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");           // this thread's share of the next stage has landed in LDS
    }
    __syncthreads();
#ifdef TOK_TIMING
    const unsigned long long t5 = __builtin_amdgcn_s_memtime();
    tacc[0] += t1 - t0;   // issue loads
    tacc[1] += t2 - t1;   // LDS fragment reads + MFMAs
    tacc[2] += t3 - t2;   // epilogue
    tacc[3] += t4 - t3;   // wait for the loads + LDS store
    tacc[4] += t5 - t4;   // barrier
    tacc[5] += 1;
#endif
    buf ^= 1;
    any = !(tile_done && !any_next);
    cur = nxt;
    cur_has = nxt_has;
  }

#ifdef TOK_TIMING
  if (tid == 0 && a.timing != nullptr) {
    for (int i = 0; i < 6; ++i) atomicAdd(&a.timing[i], tacc[i]);
    atomicAdd(&a.timing[6], __builtin_amdgcn_s_memtime() - tk0);
  }
#endif
  // ---- BatchNorm partial sums of everything this workgroup produced -> one row per workgroup ----
  if (a.stats != nullptr) {
    // recursive halving over the 16 pixel-lanes: after the 4 steps lane li owns channel li
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
    if (tid < 2 * BN) {   // NT >= 2*BN
      const int which = tid / BN;
      const int c = tid - which * BN;
      float t = 0.f;
#pragma unroll
      for (int w_ = 0; w_ < WGM; ++w_) t += red[(which * WGM + w_) * BN + c];
      const int row = xcd * S8 + jm;
      const int n = bn_fixed * BN + c;
      if (n < a.K) {
        float* dst = &a.stats[((size_t)which * a.stat_rows + row) * a.K + n];
        // fused finalize: device-coherent (sc1, write-through) store so that the folding workgroup can read the row
        // without anybody flushing or invalidating an L2 — an agent-scope release fence here costs a full L2
        // write-back per workgroup (measured: +40 % step time)
        if (a.fin_mode != 0) __hip_atomic_store(dst, t, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        else *dst = t;
      }
    }
    if (a.fin_mode != 0) {
      __shared__ int s_ticket;
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // this thread's row elements have reached the coherence point
      __syncthreads();
      if (tid == 0) s_ticket = __hip_atomic_fetch_add(&a.fin.counters[bn_fixed], 1, __ATOMIC_RELAXED,
                                                      __HIP_MEMORY_SCOPE_AGENT);
      __syncthreads();
      if (s_ticket == a.stat_rows - 1) {
        constexpr int PARTS = NT / BN;
        double* scr = reinterpret_cast<double*>(smem);   // [2][PARTS][BN]; the tile buffers are dead by now
        const int c = tid % BN, part = tid / BN;
        const int n = bn_fixed * BN + c;
        double a1 = 0.0, a2 = 0.0;
        if (n < a.K)
          for (int r = part; r < a.stat_rows; r += PARTS) {
            a1 += (double)__hip_atomic_load(&a.stats[(size_t)r * a.K + n], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            a2 += (double)__hip_atomic_load(&a.stats[((size_t)a.stat_rows + r) * a.K + n], __ATOMIC_RELAXED,
                                            __HIP_MEMORY_SCOPE_AGENT);
          }
        scr[(0 * PARTS + part) * BN + c] = a1;
        scr[(1 * PARTS + part) * BN + c] = a2;
        __syncthreads();
        if (part == 0 && n < a.K) {
#pragma unroll
          for (int q = 1; q < PARTS; ++q) { a1 += scr[(0 * PARTS + q) * BN + c]; a2 += scr[(1 * PARTS + q) * BN + c]; }
          const tok_bn_fused& f = a.fin;
          const bool real = n < f.c_real;
          if (a.fin_mode == 1) {
            const double inv = 1.0 / (double)f.count;
            const double mu = a1 * inv;
            double var = a2 * inv - mu * mu;
            if (var < 0.0) var = 0.0;
            const float muf = (float)mu;
            const float rs = (float)(1.0 / sqrt(var + (double)f.eps));
            if (real) {
              f.mean[n] = muf;
              f.rstd[n] = rs;
              const float sc = f.gamma[n] * rs;
              f.scale[n] = sc;
              f.shift[n] = fmaf(-muf, sc, f.beta[n]);
              if (f.running_mean != nullptr) {
                const double unbias = f.count > 1 ? (double)f.count / (double)(f.count - 1) : 1.0;
                f.running_mean[n] = (1.f - f.momentum) * f.running_mean[n] + f.momentum * muf;
                f.running_var[n] = (1.f - f.momentum) * f.running_var[n] + f.momentum * (float)(var * unbias);
              }
            } else {
              f.mean[n] = 0.f; f.rstd[n] = 0.f; f.scale[n] = 0.f; f.shift[n] = 0.f;
            }
          } else {
            float* coef = f.coef;
            if (real) {
              // second sum is sum(dz * y): sum(dz * xhat) = rstd * (sum(dz*y) - mean * sum(dz))
              const double sx = (double)f.rstd[n] * (a2 - (double)f.mean[n] * a1);
              const float sdz = (float)a1, sdzx = (float)sx;
              if (f.dgamma != nullptr) f.dgamma[n] = f.param_accumulate ? f.dgamma[n] + sdzx : sdzx;
              if (f.dbeta != nullptr) f.dbeta[n] = f.param_accumulate ? f.dbeta[n] + sdz : sdz;
              const double inv_m = 1.0 / (double)f.count;
              const float m1 = (float)(a1 * inv_m), m2 = (float)(sx * inv_m);
              const float g = f.gamma[n], rs = f.rstd[n], mu = f.mean[n];
              const float c1 = g * rs;
              const float c2 = -c1 * rs * m2;
              coef[n] = c1;
              coef[a.K + n] = c2;
              coef[2 * a.K + n] = -c1 * m1 - c2 * mu;
            } else {
              coef[n] = 0.f; coef[a.K + n] = 0.f; coef[2 * a.K + n] = 0.f;
            }
          }
        }
        if (tid == 0) {
          a.fin.counters[bn_fixed] = 0;     // leave the ticket counters zero for the next launch
          if (a.fin_mode == 1 && bn_fixed == 0 && a.fin.nbt != nullptr) *a.fin.nbt += 1;
        }
      }
    }
  }
}

// Persistent grid: 2 (128x128 tile) or 3 (128x64) workgroups per CU, rounded to a multiple of
// 8 * gridN so every XCD holds whole (channel tile, m-slot) groups; never more than the tiles need.
int plan_grid(int bn_tile, int gridM, int gridN, int per_cu = 0) {
  static const int pc64 = [] { const char* e = getenv("TOK_IGEMM_PER_CU_64"); return (int)(e ? atoi(e) : 0); }();      // TOK_IGEMM_PER_CU_64=<n>: persistent workgroups per CU of the 128 x 64 tile (experiment; default 3)
  if (per_cu == 0 && bn_tile == 64 && pc64 > 0) per_cu = pc64;
  const int unit = 8 * gridN;
  int G = 256 * (per_cu > 0 ? per_cu : (bn_tile == 64 ? 3 : 2));
  const long long need = (long long)gridM * gridN;
  if (need < G) G = (int)((need + unit - 1) / unit) * unit;
  G = G / unit * unit;
  if (G < unit) G = unit;
  return G;
}

template <int BM, int BN, int IN_DIV, bool C4, int PW>
int launch_pw(ConvArgs& a, hipStream_t st) {
  constexpr int smem = 2 * (BM + BN) * BK * 2 + 2 * 4 * BN * 4 + (PW == 4 ? 2 * BM * BN * 2 : 0);
  static const bool attr_set = [&] {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&conv_igemm_kernel<BM, BN, IN_DIV, C4, PW>),
                              hipFuncAttributeMaxDynamicSharedMemorySize, smem);
    return true;
  }();   // once per process (thread-safe function-local static)
  (void)attr_set;
  const int grid = a.force_grid > 0 ? a.force_grid
                                    : plan_grid(BN, a.gridM, a.gridN, PW == 4 ? 2 : 0);   // (PW 4: two shortcut tiles in LDS -> 2 per CU)
  a.stat_rows = grid / a.gridN;

#ifdef TOK_TIMING
  {
    static unsigned long long* tbuf = nullptr;
    if (!tbuf) { (void)hipMalloc(&tbuf, 64); }
    (void)hipMemsetAsync(tbuf, 0, 64, st);
    a.timing = tbuf;
  }
#endif
  hipLaunchKernelGGL((conv_igemm_kernel<BM, BN, IN_DIV, C4, PW>), dim3(grid), dim3(256), smem, st, a);
#ifdef TOK_TIMING
  {
    unsigned long long h[8];
    (void)hipMemcpyAsync(h, a.timing, 64, hipMemcpyDeviceToHost, st);
    (void)hipStreamSynchronize(st);
    const double n = h[5] ? (double)h[5] : 1.0;
    fprintf(stderr, "[timing BN=%d grid=%d M=%d K=%d N=%d] iters/WG %.1f  ticks/iter: issue %.0f  mfma %.0f  epilogue %.0f  wait+lds %.0f  barrier %.0f | loop ticks/WG %.0f\n",
            BN, grid, a.M, a.Ktot, a.K, n / grid, h[0] / n, h[1] / n, h[2] / n, h[3] / n, h[4] / n, (double)h[6] / grid);
  }
#endif
  return 0;
}

// pointwise layers whose reduction and output widths fit the ring kernel's access pattern (decided from the geometry alone,
// so that tok_conv_*_stat_rows can size the statistics rows before any launch)
// (the ring pays on the long-M streaming layers; short-M / deep-K ones are MFMA/LDS-bound and keep the two-buffer kernel's
//  3 workgroups per CU — measured per layer with tools/bench_conv.py)
static long long pw_min_rows() {
  static long long v = -1;
  if (v < 0) { const char* e = getenv("TOK_PW_RING_MIN_ROWS"); v = e ? atoll(e) : 100000; }
  return v;
}
static int pw_ring128() {   // TOK_PW_RING_BN128=<k>: 128-wide tiles with a reduction depth >= k also ride the ring (experiment; 0 = off)
  static const int v = [] { const char* e = getenv("TOK_PW_RING_BN128"); return (int)(e ? atoi(e) : 0); }();
  return v;
}
static bool pw_serves(int bn_tile, long long rows, int c_red, int n_out) {
  // (the 128-wide tile would run 1 workgroup per CU on the ring: SwinV2-T 26.0 -> 30.3 ms/step; 64-wide tiles only)
  if (bn_tile == 128 && pw_ring128() > 0 && c_red >= pw_ring128())
    return pw_ring_enabled() && rows >= pw_min_rows() && c_red % 8 == 0 && n_out % 64 == 0;
  return pw_ring_enabled() && bn_tile == 64 && rows >= pw_min_rows() && c_red % 8 == 0 && n_out % 64 == 0;
}

// 256 x 256 tiles (gemm256.hip) — decided on the GEOMETRY alone, and behind the pointwise ring, so that the statistics-row
// queries (which know the descriptor, not the epilogue mode) and every launch agree on who owns a layer
static bool g256_owns(const ConvArgs& a, int bn_tile) {
  return gemm256_geometry(a) && !pw_serves(bn_tile, a.M, a.Ktot, a.K);
}

template <int BM, int BN, int IN_DIV, bool C4>
int launch(ConvArgs& a, hipStream_t st) {
  if constexpr (IN_DIV == 1 && !C4) {
    if (a.R == 1 && a.S == 1 && a.stride == 1 && a.pad == 0) {
      if (pw_serves(BN, a.M, a.C, a.K)) {
        // pointwise layers run on the three-stage DMA ring (pw_gemm.hip); modes it does not carry (fused activation, "last
        // workgroup finalizes") stay here, on the ring's grid so that the statistics rows agree
        if (a.y2 == nullptr && a.act_x == nullptr && a.fin_mode == 0) {
          PwArgs p = {};
          p.x = a.x; p.w = a.w; p.y = a.y; p.bias = a.bias; p.stats = a.stats;
          p.stat_rows = pw_ring_grid(BN, a.gridM, a.gridN) / a.gridN;
          p.e1 = a.accumulate ? a.y : a.ep_short;
          p.e2 = a.bn_y; p.mask_in = a.bn_mask; p.mask_out = a.ep_mask;
          p.ep_scale = a.ep_scale; p.ep_shift = a.ep_shift; p.ep_relu = a.ep_relu;
          p.accumulate = a.accumulate; p.mask_store = a.mask_store;
          if (a.sub != nullptr) { p.e1 = a.sub; p.accumulate = 1; p.e1_sub = 1; p.sub_H = a.P; p.sub_W = a.Q; }
          p.M = a.M; p.C = a.C; p.N = a.K; p.gridM = a.gridM; p.gridN = a.gridN;
          const int rc = pw_ring_launch(p, BN, st);
          if (rc == 0) return 0;
          if (rc < 0) { tok_set_error("pointwise ring kernel: inconsistent arguments"); return TOK_ERR_INVALID; }
        }
        a.force_grid = pw_ring_grid(BN, a.gridM, a.gridN);
      }
      if (a.sub != nullptr) {
        tok_set_error("tok_conv_dgrad_subacc: layer not served by the pointwise ring kernel (ask tok_conv_dgrad_subacc_ok)");
        return TOK_ERR_INVALID;
      }
      // (PWM 2 = streaming / non-temporal output stores: +5..25 % on the write-heavy layers in
      //  isolation, but the consumer BatchNorm pass then misses the 256 MB Infinity Cache and the
      //  whole step loses 2 % — measured, so it stays off)
      if (a.y2 != nullptr || a.act_x != nullptr) return launch_pw<BM, BN, 1, false, 3>(a, st);
      if (a.ep_scale != nullptr) return launch_pw<BM, BN, 1, false, 4>(a, st);
      return launch_pw<BM, BN, 1, false, 1>(a, st);
    }
  }
  if (a.y2 != nullptr || a.act_x != nullptr || a.ep_scale != nullptr || a.sub != nullptr) {
    tok_set_error("fused activation / BatchNorm epilogue / subsample accumulate: 1x1 / stride 1 / no padding layers only");
    return TOK_ERR_INVALID;
  }
  return launch_pw<BM, BN, IN_DIV, C4, 0>(a, st);
}

// experiment knob: TOK_BN64=1 forces the 128x64 tile for every layer
static int force_bn64() {
  static const int v = [] { const char* e = getenv("TOK_BN64"); return (int)((e && e[0] == '1') ? 1 : 0); }();
  return v;
}

// Channel-tile width.  Short-K layers are HBM-streaming problems: the 128x64 tile (4 waves, 3
// workgroups per CU) keeps more loads/stores in flight; deep-K layers are MFMA-bound and want the
// 128x128 tile's operand reuse.  (Measured on the ResNet-50 shapes, tools/bench_conv.py.)
static int short_k() {   // TOK_SHORT_K=<k>: reduction depths up to k take the 128x64 tile (default 400: ResNet-50 B=256 sweep 100 / 260 / 400 / 520 / 768 -> 20.4 / 19.8 / 19.9 / 20.1 / 20.0 ms)
  static const int v = [] { const char* e = getenv("TOK_SHORT_K"); return (int)(e ? atoi(e) : 400); }();
  return v;
}

static int small_m_tiles() {   // TOK_SMALLM_TILES=<n>: layers with fewer 128x128 tiles than n take the 128x64 tile
  static const int v = [] { const char* e = getenv("TOK_SMALLM_TILES"); return (int)(e ? atoi(e) : 0); }();
  return v;
}

int pick_bn(int n_out, int ktot, int grid_m, bool token_rows = false) {
  if (n_out <= 64 || force_bn64()) return 64;
  // token matrices (h = w = 1: the Linear layers of SwinV2 / DaViT, N = 288 ... 3072 output features): the 128-wide tile
  // at every depth — half the tiles and half the re-reads of the A operand (measured: SwinV2-T 27.7 -> 26.9 ms/step,
  // DaViT-T 27.0 -> 26.2); the short-K rule below was tuned on the ResNet-50 convolutions, where 64 wins
  if (token_rows && getenv("TOK_SHORT_K") == nullptr) return 128;
  if (ktot <= short_k()) return 64;
  // few pixels x deep K (HRNet's low-resolution branches, the 7x7 ResNet stage): 128x128 tiles cannot fill 256 CUs
  if ((long long)grid_m * tok_cdiv(n_out, 128) < small_m_tiles()) return 64;
  return 128;
}

int check_desc(const tok_conv_desc* d, const char* who) {
  TOK_CHECK_ARG(d != nullptr, "%s: null descriptor", who);
  TOK_CHECK_ARG(d->n > 0 && d->h > 0 && d->w > 0 && d->c > 0 && d->k > 0, "%s: bad dims", who);
  TOK_CHECK_ARG(d->r > 0 && d->s > 0 && d->stride > 0 && d->pad >= 0, "%s: bad filter", who);
  TOK_CHECK_ARG(d->k % 8 == 0, "%s: k=%d must be a multiple of 8", who, d->k);
  TOK_CHECK_ARG(d->c % 8 == 0 || d->c == 4, "%s: c=%d must be a multiple of 8 (or 4: stem)", who, d->c);
  TOK_CHECK_ARG(d->p == (d->h + 2 * d->pad - d->r) / d->stride + 1 &&
                    d->q == (d->w + 2 * d->pad - d->s) / d->stride + 1,
                "%s: p/q inconsistent with h/w/r/s/stride/pad", who);
  if (d->c == 4) TOK_CHECK_ARG(d->s_pad == 8 && d->s <= 8, "%s: c4 mode needs s_pad == 8", who);
  else TOK_CHECK_ARG(d->s_pad == d->s, "%s: s_pad must equal s", who);
  TOK_CHECK_ARG((long long)d->n * d->h * d->w < (1ll << 31) && (long long)d->n * d->p * d->q < (1ll << 31),
                "%s: pixel count exceeds int32", who);
  return 0;
}

}  // namespace

extern "C" int tok_conv_fwd_stat_rows(const tok_conv_desc* d) {
  if (check_desc(d, "tok_conv_fwd_stat_rows")) return TOK_ERR_INVALID;
  const int gridM = tok_cdiv((long long)d->n * d->p * d->q, 128);
  const int bn_tile = pick_bn(d->k, d->r * d->s_pad * d->c, gridM, d->h == 1 && d->w == 1);
  const int gridN = tok_cdiv(d->k, bn_tile);
  if (d->c != 4) {
    ConvArgs g = {};
    g.C = d->c; g.K = d->k; g.Ktot = d->r * d->s_pad * d->c; g.M = d->n * d->p * d->q;
    g.H = d->h; g.W = d->w; g.P = d->p; g.Q = d->q; g.R = d->r; g.S = d->s_pad; g.stride = d->stride; g.pad = d->pad;
    const unsigned long long xb = (unsigned long long)d->n * d->h * d->w * d->c * 2;
    g.x_bytes = xb < 0xFFFFFFF0ull ? (uint32_t)xb : 0xFFFFFFF0u;
    if (conv_win_serves(g)) {
      int gm, gn;
      conv_win_tiles(g, &gm, &gn);
      return conv_win_grid(gm, gn) / gn;
    }
    if (g256_owns(g, bn_tile)) return gemm256_rows(g);
  }
  if (d->r == 1 && d->s == 1 && d->stride == 1 && d->pad == 0 && d->c != 4 && pw_serves(bn_tile, (long long)d->n * d->p * d->q, d->c, d->k))
    return pw_ring_grid(bn_tile, gridM, gridN) / gridN;
  return plan_grid(bn_tile, gridM, gridN) / gridN;
}

extern "C" int tok_conv_dgrad_stat_rows(const tok_conv_desc* d);

namespace {
int check_fused(const tok_bn_fused* bn, int k, bool fwd, const char* who) {
  TOK_CHECK_ARG(bn->counters && bn->count > 0 && bn->c_real > 0 && bn->c_real <= k && bn->gamma && bn->mean && bn->rstd,
                "%s: bad tok_bn_fused", who);
  if (fwd) TOK_CHECK_ARG(bn->beta && bn->scale && bn->shift && ((bn->running_mean == nullptr) == (bn->running_var == nullptr)),
                         "%s: bad tok_bn_fused (forward fields)", who);
  else TOK_CHECK_ARG(bn->coef, "%s: bad tok_bn_fused (coef)", who);
  return 0;
}
struct BnEpilogue { const float* scale; const float* shift; const void* shortcut; uint8_t* mask; int relu; };
int conv_fwd_impl(const tok_conv_desc* d, const void* x, const void* w, const float* bias, void* y, float* stats,
                  const tok_bn_fused* bn, void* stream, void* y_act = nullptr, int act = 0, const BnEpilogue* ep = nullptr);
}  // namespace

extern "C" int tok_conv_fwd(const tok_conv_desc* d, const void* x, const void* w,
                            const float* bias, void* y, float* stats, void* stream) {
  return conv_fwd_impl(d, x, w, bias, y, stats, nullptr, stream);
}

extern "C" int tok_conv_fwd_bn(const tok_conv_desc* d, const void* x, const void* w, void* y, float* stats,
                               const tok_bn_fused* bn, void* stream) {
  TOK_CHECK_ARG(stats && bn, "tok_conv_fwd_bn: stats / bn must not be null");
  return conv_fwd_impl(d, x, w, nullptr, y, stats, bn, stream);
}

namespace {
int conv_fwd_impl(const tok_conv_desc* d, const void* x, const void* w, const float* bias, void* y, float* stats,
                  const tok_bn_fused* bn, void* stream, void* y_act, int act, const BnEpilogue* ep) {
  if (int e = check_desc(d, "tok_conv_fwd")) return e;
  TOK_CHECK_ARG(x && w && y, "tok_conv_fwd: null pointer");
  ConvArgs a = {};
  if (bn != nullptr) {
    if (int e = check_fused(bn, d->k, true, "tok_conv_fwd_bn")) return e;
    TOK_CHECK_ARG(tok_cdiv(d->k, 64) <= 64, "tok_conv_fwd_bn: more than 64 channel tiles");
    a.fin_mode = 1;
    a.fin = *bn;
  }
  a.x = (const bf16*)x; a.w = (const bf16*)w; a.y = (bf16*)y; a.bias = bias; a.stats = stats;
  a.y2 = (bf16*)y_act; a.act = act;
  if (ep != nullptr) {
    a.ep_scale = ep->scale; a.ep_shift = ep->shift; a.ep_short = (const bf16*)ep->shortcut; a.ep_mask = ep->mask;
    a.ep_relu = ep->relu;
    const unsigned long long sb = (unsigned long long)d->n * d->p * d->q * d->k * 2;
    TOK_CHECK_ARG(sb < 0xFFFFFFF0ull, "tok_conv_fwd_bn_apply: tensors of 4 GiB or more are not supported");
    a.s_bytes = (uint32_t)sb;
  }
  a.H = d->h; a.W = d->w; a.C = d->c; a.K = d->k; a.R = d->r; a.S = d->s_pad;
  a.P = d->p; a.Q = d->q; a.stride = d->stride; a.pad = d->pad;
  a.M = d->n * d->p * d->q; a.PQ = d->p * d->q;
  a.Ktot = d->r * d->s_pad * d->c; a.KT = tok_cdiv(a.Ktot, BK);
  a.gridM = tok_cdiv(a.M, 128);
  a.fd_pq = make_fastdiv(a.PQ); a.fd_q = make_fastdiv(a.Q);
  const unsigned long long xb = (unsigned long long)d->n * d->h * d->w * d->c * 2, wb = (unsigned long long)d->k * a.Ktot * 2;
  TOK_CHECK_ARG(xb < 0xFFFFFFF0ull && wb < 0xFFFFFFF0ull, "tok_conv_fwd: tensors of 4 GiB or more are not supported");
  a.x_bytes = (uint32_t)xb; a.w_bytes = (uint32_t)wb;
  hipStream_t st = tok_stream(stream);
  const bool c4 = d->c == 4;
  int rc;
  const int bn_pick = pick_bn(d->k, a.Ktot, a.gridM, d->h == 1 && d->w == 1);
  if (c4 && ep == nullptr && stem_win_serves(a)) {
    // the 7x7 / stride 2 stem on a shared input window (stem.hip); statistics rows as tok_conv_fwd_stat_rows sized them
    rc = stem_win_launch(a, tok_conv_fwd_stat_rows(d), st);
    if (rc) return rc;
    TOK_CHECK_LAUNCH("tok_conv_fwd");
    return TOK_OK;
  }
  if (!c4 && ep == nullptr && conv_win_serves(a)) {
    // 3x3 / stride 1 / padding 1: shared input window in LDS (conv_win.hip)
    rc = conv_win_launch(a, st);
    if (rc) return rc;
    TOK_CHECK_LAUNCH("tok_conv_fwd");
    return TOK_OK;
  }
  const bool g256 = !c4 && g256_owns(a, bn_pick);
  if (g256 && gemm256_modes(a)) {
    // deep-K pointwise layers with a mid-sized pixel count: 256 x 256 tiles, eight waves (gemm256.hip)
    rc = gemm256_launch(a, st);
    if (rc) return rc;
    TOK_CHECK_LAUNCH("tok_conv_fwd");
    return TOK_OK;
  }
  // (a mode gemm256 does not carry that still writes statistics rows: this file's kernel on gemm256's row count — a multiple of 8)
  if (g256 && a.stats != nullptr && a.fin_mode == 0) a.force_grid = gemm256_rows(a) * tok_cdiv(d->k, ep != nullptr ? 64 : bn_pick);
  if (a.fin_mode != 0) {
    // "last workgroup finalizes" stays on this file's kernels; the caller sized `stats` with tok_conv_fwd_stat_rows, which may
    // describe the (smaller) grid of conv_win: never write more rows than that
    const int rows_q = tok_conv_fwd_stat_rows(d);
    const int gn = tok_cdiv(d->k, bn_pick);
    if (rows_q > 0 && rows_q * gn < plan_grid(bn_pick, a.gridM, gn)) a.force_grid = rows_q * gn;
  }
  if (ep != nullptr || bn_pick == 64) {   // (BN epilogue: 64-wide tiles)
    a.gridN = tok_cdiv(d->k, 64);
    rc = c4 ? launch<128, 64, 1, true>(a, st) : launch<128, 64, 1, false>(a, st);
  } else {
    a.gridN = tok_cdiv(d->k, 128);
    rc = c4 ? launch<128, 128, 1, true>(a, st) : launch<128, 128, 1, false>(a, st);
  }
  if (rc) return rc;
  TOK_CHECK_LAUNCH("tok_conv_fwd");
  return TOK_OK;
}
}  // namespace

namespace {

struct DgradPlan { int bn_tile, gridM, gridN; };

int dgrad_fill(const tok_conv_desc* d, ConvArgs& a, DgradPlan& pl) {
  TOK_CHECK_ARG(d->c % 8 == 0, "tok_conv_dgrad: c4 (stem) input needs no data gradient");
  TOK_CHECK_ARG(d->stride == 1 || d->stride == 2, "tok_conv_dgrad: stride %d unsupported", d->stride);
  TOK_CHECK_ARG(d->r - 1 - d->pad >= 0, "tok_conv_dgrad: pad > r-1 unsupported");
  TOK_CHECK_ARG(d->r == d->s, "tok_conv_dgrad: square filters only");
  // gathered tensor = dY (P x Q x K), output = dX (H x W x C)
  a.H = d->p; a.W = d->q; a.C = d->k; a.K = d->c; a.R = d->r; a.S = d->s;
  a.P = d->h; a.Q = d->w; a.stride = 1; a.pad = d->r - 1 - d->pad;
  a.M = d->n * d->h * d->w; a.PQ = d->h * d->w;
  a.Ktot = d->r * d->s * d->k; a.KT = tok_cdiv(a.Ktot, BK);
  a.fd_pq = make_fastdiv(a.PQ); a.fd_q = make_fastdiv(a.Q);
  a.uniform_taps = (d->k % BK == 0) ? 1 : 0;
  const unsigned long long xb = (unsigned long long)d->n * d->p * d->q * d->k * 2, wb = (unsigned long long)d->c * a.Ktot * 2;
  TOK_CHECK_ARG(xb < 0xFFFFFFF0ull && wb < 0xFFFFFFF0ull, "tok_conv_dgrad: tensors of 4 GiB or more are not supported");
  a.x_bytes = (uint32_t)xb; a.w_bytes = (uint32_t)wb;
  if (d->stride == 1) {
    a.gridM = tok_cdiv(a.M, 128);
  } else {
    int tmax = 0;
    for (int cls = 0; cls < 4; ++cls) {
      const int ph = cls >> 1, pw = cls & 1;
      const int nh = (d->h - ph + 1) / 2, nw = (d->w - pw + 1) / 2;
      a.cls_M[cls] = d->n * nh * nw;
      a.cls_nw[cls] = nw;
      a.cls_hw[cls] = nh * nw;
      a.cls_fd_hw[cls] = make_fastdiv(nh * nw > 0 ? nh * nw : 1);
      a.cls_fd_w[cls] = make_fastdiv(nw > 0 ? nw : 1);
      const int t = tok_cdiv(a.cls_M[cls], 128);
      if (t > tmax) tmax = t;
    }
    a.gridM = 4 * tmax;   // classes interleaved (m-tile & 3) so heavy and light tiles mix on every XCD
  }
  pl.bn_tile = pick_bn(d->c, a.Ktot, a.gridM, d->h == 1 && d->w == 1);
  a.gridN = tok_cdiv(d->c, pl.bn_tile);
  pl.gridM = a.gridM; pl.gridN = a.gridN;
  return 0;
}

int dgrad_impl(const tok_conv_desc* d, const void* dy, const void* w_dgrad, void* dx, int accumulate,
               const void* bn_y, const uint8_t* bn_mask, float* partial, void* stream, const char* who,
               const tok_bn_fused* bn = nullptr, const void* act_x = nullptr, int act = 0, const float* bias = nullptr,
               int mask_store = 0, const void* sub = nullptr) {
  if (int e = check_desc(d, who)) return e;
  TOK_CHECK_ARG(dy && w_dgrad && dx, "%s: null pointer", who);
  ConvArgs a = {};
  DgradPlan pl;
  if (int e = dgrad_fill(d, a, pl)) return e;
  a.x = (const bf16*)dy; a.w = (const bf16*)w_dgrad; a.y = (bf16*)dx; a.bias = bias;
  a.stats = partial; a.bn_y = (const bf16*)bn_y; a.bn_mask = bn_mask; a.mask_store = mask_store;
  a.accumulate = accumulate;
  a.sub = (const bf16*)sub;
  a.act_x = (const bf16*)act_x; a.act = act;
  if (bn != nullptr) {
    if (int e = check_fused(bn, d->c, false, who)) return e;
    TOK_CHECK_ARG(tok_cdiv(d->c, 64) <= 64, "%s: more than 64 channel tiles", who);
    a.fin_mode = 2;
    a.fin = *bn;
  }
  hipStream_t st = tok_stream(stream);
  int rc;
  if (d->stride == 1 && conv_win_serves(a)) {
    rc = conv_win_launch(a, st);
    if (rc) return rc;
    TOK_CHECK_LAUNCH(who);
    return TOK_OK;
  }
  if (d->stride == 2 && conv_s2d_serves(a, d->stride, d->pad)) {
    rc = conv_s2d_launch(a, st);
    if (rc) return rc;
    TOK_CHECK_LAUNCH(who);
    return TOK_OK;
  }
  const bool g256 = d->stride == 1 && g256_owns(a, pl.bn_tile);
  if (g256 && gemm256_modes(a)) {
    rc = gemm256_launch(a, st);
    if (rc) return rc;
    TOK_CHECK_LAUNCH(who);
    return TOK_OK;
  }
  if (g256 && a.stats != nullptr && a.fin_mode == 0) a.force_grid = gemm256_rows(a) * pl.gridN;   // (see conv_fwd_impl)
  if (a.fin_mode != 0) {
    const int rows_q = tok_conv_dgrad_stat_rows(d);
    if (rows_q > 0 && rows_q * pl.gridN < plan_grid(pl.bn_tile, pl.gridM, pl.gridN)) a.force_grid = rows_q * pl.gridN;
  }
  if (pl.bn_tile == 64) rc = d->stride == 1 ? launch<128, 64, 1, false>(a, st) : launch<128, 64, 2, false>(a, st);
  else rc = d->stride == 1 ? launch<128, 128, 1, false>(a, st) : launch<128, 128, 2, false>(a, st);
  if (rc) return rc;
  TOK_CHECK_LAUNCH(who);
  return TOK_OK;
}

}  // namespace

extern "C" int tok_conv_fwd_act(const tok_conv_desc* d, const void* x, const void* w, const float* bias, void* y,
                                void* y_act, int kind, void* stream) {
  TOK_CHECK_ARG(y_act != nullptr && (kind == 0 || kind == 1), "tok_conv_fwd_act: y_act and kind 0 (ReLU) / 1 (GELU)");
  return conv_fwd_impl(d, x, w, bias, y, nullptr, nullptr, stream, y_act, kind);
}

extern "C" int tok_conv_dgrad_act(const tok_conv_desc* d, const void* dy, const void* w_dgrad, const void* act_x, int kind,
                                  void* dx, void* stream) {
  TOK_CHECK_ARG(act_x != nullptr && (kind == 0 || kind == 1), "tok_conv_dgrad_act: act_x and kind 0 (ReLU) / 1 (GELU)");
  return dgrad_impl(d, dy, w_dgrad, dx, 0, nullptr, nullptr, nullptr, stream, "tok_conv_dgrad_act", nullptr, act_x, kind);
}

extern "C" int tok_conv_dgrad(const tok_conv_desc* d, const void* dy, const void* w_dgrad, void* dx,
                              int accumulate, void* stream) {
  return dgrad_impl(d, dy, w_dgrad, dx, accumulate, nullptr, nullptr, nullptr, stream, "tok_conv_dgrad");
}

extern "C" int tok_conv_dgrad_stat_rows(const tok_conv_desc* d) {
  if (check_desc(d, "tok_conv_dgrad_stat_rows")) return TOK_ERR_INVALID;
  ConvArgs a = {};
  DgradPlan pl;
  if (dgrad_fill(d, a, pl)) return TOK_ERR_INVALID;
  if (d->stride == 1 && conv_win_serves(a)) {
    int gm, gn;
    conv_win_tiles(a, &gm, &gn);
    return conv_win_grid(gm, gn) / gn;
  }
  if (d->stride == 2 && conv_s2d_serves(a, d->stride, d->pad)) {
    int gm, gn;
    conv_s2d_tiles(a, &gm, &gn);
    return conv_s2d_grid(gm, gn) / gn;
  }
  if (d->stride == 1 && g256_owns(a, pl.bn_tile)) return gemm256_rows(a);
  if (d->r == 1 && d->s == 1 && d->stride == 1 && d->pad == 0 && pw_serves(pl.bn_tile, (long long)d->n * d->h * d->w, d->k, d->c))
    return pw_ring_grid(pl.bn_tile, pl.gridM, pl.gridN) / pl.gridN;
  return plan_grid(pl.bn_tile, pl.gridM, pl.gridN) / pl.gridN;
}

extern "C" int tok_conv_dgrad_bnstats(const tok_conv_desc* d, const void* dy, const void* w_dgrad, void* dx,
                                      int accumulate, const void* bn_y, const uint8_t* bn_mask,
                                      float* partial, void* stream) {
  TOK_CHECK_ARG(bn_y && partial, "tok_conv_dgrad_bnstats: bn_y / partial must not be null");
  return dgrad_impl(d, dy, w_dgrad, dx, accumulate, bn_y, bn_mask, partial, stream, "tok_conv_dgrad_bnstats");
}

extern "C" int tok_conv_dgrad_bn(const tok_conv_desc* d, const void* dy, const void* w_dgrad, void* dx, int accumulate,
                                 const void* bn_y, const uint8_t* bn_mask, float* partial, const tok_bn_fused* bn,
                                 void* stream) {
  TOK_CHECK_ARG(bn_y && partial && bn, "tok_conv_dgrad_bn: bn_y / partial / bn must not be null");
  return dgrad_impl(d, dy, w_dgrad, dx, accumulate, bn_y, bn_mask, partial, stream, "tok_conv_dgrad_bn", bn);
}


// ---- "unit 3" of a bottleneck: 1x1 conv -> BatchNorm -> + shortcut -> ReLU without the pre-normalisation tensor -----------

extern "C" int tok_conv_fwd_bn_apply(const tok_conv_desc* d, const void* x, const void* w, const float* scale,
                                     const float* shift, const void* shortcut, int relu, void* out, uint8_t* mask,
                                     void* stream) {
  TOK_CHECK_ARG(d && scale && shift && out, "tok_conv_fwd_bn_apply: null pointer");
  TOK_CHECK_ARG(d->r == 1 && d->s == 1 && d->stride == 1 && d->pad == 0 && d->c % 8 == 0,
                "tok_conv_fwd_bn_apply: 1x1 / stride 1 / no padding layers only");
  const BnEpilogue ep = {scale, shift, shortcut, mask, relu};
  return conv_fwd_impl(d, x, w, nullptr, out, nullptr, nullptr, stream, nullptr, 0, &ep);
}

extern "C" int tok_conv_dgrad_maskstore(const tok_conv_desc* d, const void* dy, const void* w_dgrad, void* dx,
                                        int accumulate, const uint8_t* mask, float* partial, void* stream) {
  TOK_CHECK_ARG(mask && partial, "tok_conv_dgrad_maskstore: mask / partial must not be null");
  return dgrad_impl(d, dy, w_dgrad, dx, accumulate, nullptr, mask, partial, stream, "tok_conv_dgrad_maskstore", nullptr,
                    nullptr, 0, nullptr, 1);
}

extern "C" int tok_conv_dgrad_bias(const tok_conv_desc* d, const void* dy, const void* w_dgrad, const float* bias, void* dx,
                                   int accumulate, const void* bn_y, const uint8_t* bn_mask, float* partial, void* stream) {
  TOK_CHECK_ARG((bn_y == nullptr) == (partial == nullptr), "tok_conv_dgrad_bias: bn_y and partial go together");
  return dgrad_impl(d, dy, w_dgrad, dx, accumulate, bn_y, bn_mask, partial, stream, "tok_conv_dgrad_bias", nullptr, nullptr, 0,
                    bias);
}

// ---- pointwise dgrad + the gradient of the stride-2 pixel subsample of the same tensor ----------------------------------------
// The block input x of a strided bottleneck feeds conv1 (1x1 / stride 1) and, through tok_subsample2_fwd, the projection
// shortcut.  d(x) = dgrad(conv1) + scatter(d(subsample)): the scatter rides the ring kernel's accumulate stage (rows with an
// odd coordinate fetch nothing), so d(x) is written once and the half-resolution gradient is never expanded in HBM.

extern "C" int tok_conv_dgrad_subacc_ok(const tok_conv_desc* d) {
  if (check_desc(d, "tok_conv_dgrad_subacc_ok")) return 0;
  if (!(d->r == 1 && d->s == 1 && d->stride == 1 && d->pad == 0 && d->c % 8 == 0)) return 0;
  ConvArgs a = {};
  DgradPlan pl;
  if (dgrad_fill(d, a, pl)) return 0;
  return pw_serves(pl.bn_tile, (long long)d->n * d->h * d->w, d->k, d->c) ? 1 : 0;
}

extern "C" int tok_conv_dgrad_subacc(const tok_conv_desc* d, const void* dy, const void* w_dgrad, void* dx, const void* dsub,
                                     const void* bn_y, const uint8_t* mask, float* partial, int mask_store, void* stream) {
  TOK_CHECK_ARG(dsub != nullptr, "tok_conv_dgrad_subacc: dsub must not be null");
  TOK_CHECK_ARG(!mask_store || (mask && partial && !bn_y), "tok_conv_dgrad_subacc: mask_store needs mask + partial, no bn_y");
  TOK_CHECK_ARG(mask_store || ((bn_y == nullptr) == (partial == nullptr)), "tok_conv_dgrad_subacc: bn_y and partial go together");
  return dgrad_impl(d, dy, w_dgrad, dx, 0, bn_y, mask, partial, stream, "tok_conv_dgrad_subacc", nullptr, nullptr, 0, nullptr,
                    mask_store, dsub);
}

// ---- sum of two pointwise data gradients in one launch --------------------------------------------------------------------------
// d(x) = dgrad(d1: dy1, w1) + dgrad(d2: dy2, w2) (+ bias) for two 1x1 / stride-1 layers over the same pixels and the same input
// width: the fused residual unit's  d(input) = dz Wa + z Wb + c.  The K stages of the second product follow the first's in the
// ring kernel's tile loop: one store of d(x) instead of a store, a re-read and a second store.  Epilogue options as
// tok_conv_dgrad_bias.  Served where tok_conv_dgrad2_ok(d1, d2) (64-wide tiles on the ring).

namespace {
bool dgrad2_geometry(const tok_conv_desc* d1, const tok_conv_desc* d2) {
  if (d1 == nullptr || d2 == nullptr) return false;
  for (const tok_conv_desc* d : {d1, d2})
    if (!(d->r == 1 && d->s == 1 && d->stride == 1 && d->pad == 0 && d->k % 8 == 0 && d->c % 64 == 0)) return false;
  if (d1->n != d2->n || d1->h != d2->h || d1->w != d2->w || d1->c != d2->c) return false;
  const long long rows = (long long)d1->n * d1->h * d1->w;
  // the second layer alone must sit on the 64-wide ring too: its statistics-row count is the one the caller sizes `partial` with
  const int gridM = tok_cdiv(rows, 128);
  if (pick_bn(d2->c, d2->k, gridM) != 64) return false;
  return pw_serves(64, rows, d1->k, d1->c) && pw_serves(64, rows, d2->k, d2->c);
}
}  // namespace

extern "C" int tok_conv_dgrad2_ok(const tok_conv_desc* d1, const tok_conv_desc* d2) { return dgrad2_geometry(d1, d2) ? 1 : 0; }

extern "C" int tok_conv_dgrad2(const tok_conv_desc* d1, const void* dy1, const void* w1_dgrad, const tok_conv_desc* d2,
                               const void* dy2, const void* w2_dgrad, const float* bias, void* dx, int accumulate,
                               const void* bn_y, const uint8_t* bn_mask, float* partial, void* stream) {
  TOK_CHECK_ARG(dgrad2_geometry(d1, d2), "tok_conv_dgrad2: layers not served (ask tok_conv_dgrad2_ok)");
  TOK_CHECK_ARG(dy1 && w1_dgrad && dy2 && w2_dgrad && dx, "tok_conv_dgrad2: null pointer");
  TOK_CHECK_ARG((bn_y == nullptr) == (partial == nullptr), "tok_conv_dgrad2: bn_y and partial go together");
  const long long rows = (long long)d1->n * d1->h * d1->w;
  PwArgs p = {};
  p.x = (const bf16*)dy1; p.w = (const bf16*)w1_dgrad; p.C = d1->k;
  p.x2 = (const bf16*)dy2; p.w2 = (const bf16*)w2_dgrad; p.C2 = d2->k;
  p.y = (bf16*)dx; p.bias = bias; p.stats = partial;
  p.M = (int)rows; p.N = d1->c;
  p.gridM = tok_cdiv(rows, 128); p.gridN = tok_cdiv(d1->c, 64);
  p.stat_rows = pw_ring_grid(64, p.gridM, p.gridN) / p.gridN;
  p.e1 = accumulate ? (const bf16*)dx : nullptr;
  p.accumulate = accumulate;
  p.e2 = (const bf16*)bn_y; p.mask_in = bn_mask;
  const int rc = pw_ring_launch(p, 64, tok_stream(stream));
  if (rc != 0) { tok_set_error("tok_conv_dgrad2: ring kernel refused the launch (%d)", rc); return TOK_ERR_INVALID; }
  TOK_CHECK_LAUNCH("tok_conv_dgrad2");
  return TOK_OK;
}
