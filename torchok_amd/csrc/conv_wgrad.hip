// Weight-gradient of the NHWC bf16 convolution on MFMA for gfx950.
//
//   dW[n][k] = sum_m dY[m][n] * A[m][k]      m = (b, p, q)   k = (r, s, c)
//
// The reduction index m is the SLOW (strided) index of both operands, so the MFMA fragments
// (8 consecutive reduction elements per lane) are produced by the gfx950 LDS transpose read
// ds_read_b64_tr_b16: tiles are staged row-major [m][n] / [m][k] exactly as they sit in HBM
// (16-byte coalesced loads, 16-byte LDS stores) and each 16-lane group pulls a 4(m) x 16(col)
// block out transposed.  The k-slot -> m mapping inside one MFMA is
//   slot (g, j<4) -> m = 4g + j,  slot (g, j>=4) -> m = 16 + 4g + (j-4)      (g = lane>>4)
// identically for both operands, which is all a reduction needs.
//
// Work split: output tile TN x TK (64 or 128 each), the M range is cut into `splitM` chunks
// that run as independent workgroups and write fp32 partial tiles to a workspace; a second
// kernel sums the partials in a fixed order (deterministic — no atomics) and un-pads into the
// fp32 master-gradient layout [k][r][s][c].
#include "tok_common.h"
#include "conv_common.h"
#include <type_traits>
#include <stdlib.h>

namespace {

struct WgradArgs {
  const bf16* x;
  const bf16* dy;
  float* ws;
  int H, W, C, K, R, S, P, Q, stride, pad;
  int M, PQ, HW, Ktot;
  int tilesN, tilesK, splitM, mchunk;
  uint32_t x_bytes, dy_bytes;
  uint32_t pq_mul, pq_shift, q_mul, q_shift;   // magic-number division by PQ and Q (row cursor of the 3x3 gather)
  float* cs;     // ring kernel, may be null: column sums of dy over this split's rows (bias-gradient partials); split s at
                 // cs + s * cs_stride — with the flat reduce the row sits right behind the split's dW slab (one reduce launch)
  int cs_cols;   // unpadded output channel count
  int adv_q, adv_p;   // pipelined window kernel: 32 % Q and (32 / Q) % P — how a pixel's (row, column) moves per 32-pixel stage
  size_t cs_stride, ws_stride;   // floats between consecutive splits of cs / of the dW partial slabs
};

// n / d for n < 2^31 with (mul, shift) from make_magic(d)
__device__ __forceinline__ uint32_t magic_div(uint32_t n, uint32_t mul, uint32_t shift) {
  return (uint32_t)(((uint64_t)__umulhi(n, mul) + n) >> shift);
}
void make_magic(uint32_t d, uint32_t& mul, uint32_t& shift) {
  mul = 0; shift = 0;
  if (d <= 1) return;
  uint32_t l = 0;
  while ((1ull << l) < d) ++l;
  mul = (uint32_t)((((1ull << 32) * ((1ull << l) - d)) / d) + 1);
  shift = l;
}

typedef __attribute__((address_space(3))) bf16x4 lds_bf16x4;

__device__ __forceinline__ u32x2 tr_read_asm(uint32_t lds_addr) {
  u32x2 v;
  asm volatile("ds_read_b64_tr_b16 %0, %1" : "=v"(v) : "v"(lds_addr));
  return v;
}

__device__ __forceinline__ bf16x4 tr_read(const char* p) {
  return __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_bf16x4*)(p));
}

template <int TN, int TK, bool C4, int MS, bool DMA_T, bool PWK>
__global__ __launch_bounds__(256) void conv_wgrad_kernel(WgradArgs a) {              // reduction rows staged per barrier (2 MFMA k-steps of 32)
  constexpr int CN = TN / 8, CK = TK / 8;      // 16-byte chunks per tile row
  constexpr int RPY = 256 / CN, RPX = 256 / CK;  // rows covered per pass
  constexpr int YP = MS / RPY, XP = MS / RPX;    // passes per step
  // DMA (all but the c4 stem): tiles go global -> LDS directly (buffer_load ... lds): no staging
  // registers, no ds_write pass (LDS stores run at ~80 B/clk/CU and bounded this loop).  The LDS
  // image of a wave instruction is lane-linear, so rows are unpadded and the transpose reads are
  // kept conflict-free by an XOR swizzle of the 16-byte chunk index with the row, applied on the
  // SOURCE side (the lane at slot c fetches logical chunk c ^ swz(row)) and on the reads.
  // Measured per layer (tools/bench_conv.py + whole-step bench): DMA wins on the short-M layers
  // (14x14, 7x7: -18 %), register staging with its longer prefetch distance on the long-M ones.
  constexpr bool DMA = !C4 && DMA_T;
  constexpr int YS = DMA ? TN * 2 : (TN + 16) * 2, XS = DMA ? TK * 2 : (TK + 16) * 2;  // row strides (bytes)
  constexpr int YBYTES = MS * YS, XBYTES = MS * XS;
  constexpr int SWY = CN / 2 - 1, SWX = CK / 2 - 1;   // swz(row) = (row & SW) << 1
  constexpr int NT = TN / 32, KTL = TK / 32;   // 16-wide tiles per wave (2x2 waves)

  extern __shared__ __attribute__((aligned(16))) char smem[];

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wv = tid >> 6;
  const int wn2 = wv & 1, wk2 = wv >> 1;

  const int ntile = a.tilesN * a.tilesK;
  const int id = tok_xcd_remap(blockIdx.x, ntile * a.splitM);
  const int split = id / ntile;
  const int t = id - split * ntile;
  const int tn = t / a.tilesK;
  const int tk = t - tn * a.tilesK;

  const int mstart = split * a.mchunk;
  const int mend = min(a.M, mstart + a.mchunk);
  const int steps = (mend - mstart + MS - 1) / MS;

  // ---- staging assignment ---------------------------------------------------------------
  const int ycol = tid % CN, yrow = tid / CN;
  const int xcol = tid % CK, xrow = tid / CK;
  // logical chunk fetched by this thread (slot ^ swz(row); the row's low bits are pass-invariant)
  const int ylog = DMA ? (ycol ^ ((yrow & SWY) << 1)) : ycol;
  const int xlog = DMA ? (xcol ^ ((xrow & SWX) << 1)) : xcol;
  const int wave_u = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int yn = tn * TN + ylog * 8;
  const bool yn_ok = yn < a.K;

  // this thread's k-chunk never changes: tap (kr, ks), channel offset kc0
  const int k0 = tk * TK + xlog * 8;
  int kr, ks, kc0;
  if (C4) { kr = k0 >> 5; ks = (k0 & 31) >> 2; kc0 = 0; }
  else { const int tap = k0 / a.C; kc0 = k0 - tap * a.C; kr = tap / a.S; ks = tap - kr * a.S; }
  const bool k_ok = kr < a.R;
  // PWK: 1x1, stride 1, no padding (every Linear and most ResNet convs): input pixel == output pixel, no row cursor
  constexpr bool pointwise = PWK;

  // row cursor: output pixel of this thread's X rows; (image, p, q) are re-derived per step by magic-number division —
  // straight-line code for any map size (the incremental "while (q >= Q)" cursor was a divergent loop per row and step)
  int xm[XP];
#pragma unroll
  for (int i = 0; i < XP; ++i) xm[i] = mstart + xrow + i * RPX;
  int ym = mstart + yrow;

  bf16x8 ry[YP], rx[XP];

  // buffer loads: rows past the chunk end, padding taps and the K tail use an out-of-range offset
  // (answered with zeros): unconditional instructions, exact wait counts, no zero-fill writes
  const __amdgpu_buffer_rsrc_t ysrd = __builtin_amdgcn_make_buffer_rsrc((void*)a.dy, 0, a.dy_bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t xsrd = __builtin_amdgcn_make_buffer_rsrc((void*)a.x, 0, a.x_bytes, 0x00020000);
  typedef __attribute__((address_space(3))) void lds_void;
  auto load_tile = [&](int dbuf) {
    char* Ydst = smem + dbuf * (YBYTES + XBYTES) + wave_u * 1024;
    char* Xdst = smem + dbuf * (YBYTES + XBYTES) + YBYTES + wave_u * 1024;
#pragma unroll
    for (int i = 0; i < YP; ++i) {
      const int m = ym + i * RPY;
      uint32_t off = (yn_ok && m < mend) ? (uint32_t)(m * a.K + yn) * 2u : 0xFFFFFFF0u;
      if (DMA) {
        asm volatile("" : "+v"(off));      // one unconditional DMA per row (no branch around the load)
        __builtin_amdgcn_raw_ptr_buffer_load_lds(ysrd, (lds_void*)(Ydst + i * RPY * YS), 16, off, 0, 0, 0);
      }
      else ry[i] = __builtin_bit_cast(bf16x8, __builtin_amdgcn_raw_buffer_load_b128(ysrd, off, 0, 0));
    }
    ym += MS;
#pragma unroll
    for (int i = 0; i < XP; ++i) {
      int hh = 0, ww = 0, xpix_i = 0;
      if (!pointwise) {
        const uint32_t mm = (uint32_t)min(xm[i], a.M - 1);
        const uint32_t b = magic_div(mm, a.pq_mul, a.pq_shift);
        const uint32_t rem = mm - b * (uint32_t)a.PQ;
        const uint32_t pp = magic_div(rem, a.q_mul, a.q_shift);
        hh = (int)pp * a.stride - a.pad + kr;
        ww = (int)(rem - pp * (uint32_t)a.Q) * a.stride - a.pad + ks;
        xpix_i = (int)b * a.HW;
      }
      const bool ok = k_ok && xm[i] < mend && (unsigned)hh < (unsigned)a.H;
      bf16x8 v = zero8();
      if (C4) {
        if (ok) {
          const bf16* ptr = a.x + ((size_t)(xpix_i + hh * a.W + ww)) * 4;
          bf16x4 lo = {(bf16)0.f, (bf16)0.f, (bf16)0.f, (bf16)0.f}, hi = lo;
          if ((unsigned)ww < (unsigned)a.W) lo = *reinterpret_cast<const bf16x4*>(ptr);
          if ((unsigned)(ww + 1) < (unsigned)a.W) hi = *reinterpret_cast<const bf16x4*>(ptr + 4);
#pragma unroll
          for (int e = 0; e < 4; ++e) { v[e] = lo[e]; v[4 + e] = hi[e]; }
        }
      } else {
        // pointwise (1x1, stride 1, no padding: every Linear and most ResNet convs): input pixel == output pixel
        uint32_t goff = pointwise ? (uint32_t)(xm[i] * a.C + kc0) * 2u : (uint32_t)((xpix_i + hh * a.W + ww) * a.C + kc0) * 2u;
        if (DMA) asm volatile("" : "+v"(goff));
        uint32_t off = (pointwise ? (k_ok && xm[i] < mend) : (ok && (unsigned)ww < (unsigned)a.W)) ? goff : 0xFFFFFFF0u;
        if (DMA) {
          asm volatile("" : "+v"(off));
          __builtin_amdgcn_raw_ptr_buffer_load_lds(xsrd, (lds_void*)(Xdst + i * RPX * XS), 16, off, 0, 0, 0);
        }
        else v = __builtin_bit_cast(bf16x8, __builtin_amdgcn_raw_buffer_load_b128(xsrd, off, 0, 0));
      }
      rx[i] = v;
      // advance this row cursor by MS output pixels
      xm[i] += MS;
    }
  };

  auto store_tile = [&](int buf) {
    if (DMA) return;   // already in flight into LDS; hipcc drains vmcnt before the barrier
    char* Yb = smem + buf * (YBYTES + XBYTES);
    char* Xb = Yb + YBYTES;
#pragma unroll
    for (int i = 0; i < YP; ++i)
      *reinterpret_cast<bf16x8*>(Yb + (yrow + i * RPY) * YS + ycol * 16) = ry[i];
#pragma unroll
    for (int i = 0; i < XP; ++i)
      *reinterpret_cast<bf16x8*>(Xb + (xrow + i * RPX) * XS + xcol * 16) = rx[i];
  };

  const int g = lane >> 4, li = lane & 15;
  // transpose-read address of this lane inside a 16-column tile: row 4g + (li>>2), cols (li&3)*4..
  const int rrow = 4 * g + (li >> 2);                       // row inside a 16-row half step
  const int yoff = rrow * YS + (wn2 * (TN / 2) + (li & 3) * 4) * 2;
  const int xoff = rrow * XS + (wk2 * (TK / 2) + (li & 3) * 4) * 2;
  // DMA layout: byte column -> ((chunk ^ swz(row)) << 4) | (byte & 15); rows +16/+32 keep row & 7
  const int yswz = ((rrow & SWY) << 1) << 4, xswz = ((rrow & SWX) << 1) << 4;
  auto ycolb = [&](int i) { const int b = (wn2 * (TN / 2) + i * 16 + (li & 3) * 4) * 2; return DMA ? (b ^ yswz) : b; };
  auto xcolb = [&](int j) { const int b = (wk2 * (TK / 2) + j * 16 + (li & 3) * 4) * 2; return DMA ? (b ^ xswz) : b; };

  f32x4 acc[NT][KTL];
#pragma unroll
  for (int i = 0; i < NT; ++i)
#pragma unroll
    for (int j = 0; j < KTL; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};

  if (steps > 0) {
    load_tile(0);
    store_tile(0);
  }
  if (DMA) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  typedef __attribute__((address_space(3))) char lds_char;
  const uint32_t lds_base = (uint32_t)(size_t)(lds_char*)smem;

  for (int st = 0; st < steps; ++st) {
    // past the last step the cursors point beyond `mend`: the loads return zeros without traffic
    load_tile((st + 1) & 1);
    const char* Yb = smem + (st & 1) * (YBYTES + XBYTES);
    const char* Xb = Yb + YBYTES;
    if constexpr (true) {
      // every transpose read of the stage first (inline asm: no vmcnt(0) fence against the DMA of the next stage, no
      // one-wait-per-MFMA interleaving), then the MFMAs of each 32-row half as its fragments land
      constexpr int HALVES = MS / 32;
      const uint32_t Yl = lds_base + (st & 1) * (YBYTES + XBYTES), Xl = Yl + YBYTES;
      u32x2 ya[HALVES][NT][2], xb[HALVES][KTL][2];
#pragma unroll
      for (int ks = 0; ks < HALVES; ++ks) {
#pragma unroll
        for (int i = 0; i < NT; ++i) {
          ya[ks][i][0] = tr_read_asm(Yl + (rrow + ks * 32) * YS + ycolb(i));
          ya[ks][i][1] = tr_read_asm(Yl + (rrow + ks * 32 + 16) * YS + ycolb(i));
        }
#pragma unroll
        for (int j = 0; j < KTL; ++j) {
          xb[ks][j][0] = tr_read_asm(Xl + (rrow + ks * 32) * XS + xcolb(j));
          xb[ks][j][1] = tr_read_asm(Xl + (rrow + ks * 32 + 16) * XS + xcolb(j));
        }
      }
#pragma unroll
      for (int ks = 0; ks < HALVES; ++ks) {
        __builtin_amdgcn_sched_barrier(0);
        if (ks + 1 < HALVES) asm volatile("s_waitcnt lgkmcnt(%0)" ::"n"(2 * (NT + KTL) > 15 ? 15 : 2 * (NT + KTL)) : "memory");   // (4-bit counter)
        else asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int i = 0; i < NT; ++i) {
          const bf16x8 af = __builtin_bit_cast(bf16x8, (u32x4){ya[ks][i][0][0], ya[ks][i][0][1], ya[ks][i][1][0], ya[ks][i][1][1]});
#pragma unroll
          for (int j = 0; j < KTL; ++j) {
            const bf16x8 bfr = __builtin_bit_cast(bf16x8, (u32x4){xb[ks][j][0][0], xb[ks][j][0][1], xb[ks][j][1][0], xb[ks][j][1][1]});
            acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af, bfr, acc[i][j], 0, 0, 0);
          }
        }
      }
      __builtin_amdgcn_sched_barrier(0);                 // (the MFMAs stay in front of the wait: they cover the loads' latency)
      if (DMA) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // this thread's share of the next stage has landed
      else store_tile((st + 1) & 1);                     // register-staged stem: park the landed rows (ds_write)
      __syncthreads();
      continue;
    }
#pragma unroll
    for (int ks = 0; ks < MS / 32; ++ks) {
      bf16x8 af[NT], bfr[KTL];
#pragma unroll
      for (int i = 0; i < NT; ++i) {
        const bf16x4 lo = tr_read(Yb + (rrow + ks * 32) * YS + ycolb(i));
        const bf16x4 hi = tr_read(Yb + (rrow + ks * 32 + 16) * YS + ycolb(i));
#pragma unroll
        for (int e = 0; e < 4; ++e) { af[i][e] = lo[e]; af[i][4 + e] = hi[e]; }
      }
#pragma unroll
      for (int j = 0; j < KTL; ++j) {
        const bf16x4 lo = tr_read(Xb + (rrow + ks * 32) * XS + xcolb(j));
        const bf16x4 hi = tr_read(Xb + (rrow + ks * 32 + 16) * XS + xcolb(j));
#pragma unroll
        for (int e = 0; e < 4; ++e) { bfr[j][e] = lo[e]; bfr[j][4 + e] = hi[e]; }
      }
#pragma unroll
      for (int i = 0; i < NT; ++i)
#pragma unroll
        for (int j = 0; j < KTL; ++j)
          acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[i], bfr[j], acc[i][j], 0, 0, 0);
    }
    store_tile((st + 1) & 1);
    __syncthreads();
  }

  // D[i = n][j = kcol]: lane holds rows g*4+reg, column li
  float* out = a.ws + (size_t)split * a.K * a.Ktot;
#pragma unroll
  for (int i = 0; i < NT; ++i) {
#pragma unroll
    for (int j = 0; j < KTL; ++j) {
      const int kcol = tk * TK + wk2 * (TK / 2) + j * 16 + li;
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int n = tn * TN + wn2 * (TN / 2) + i * 16 + g * 4 + r;
        if (n < a.K && kcol < a.Ktot) out[(size_t)n * a.Ktot + kcol] = acc[i][j][r];
      }
    }
  }
}


// ---- ring variant ------------------------------------------------------------------------------------------
// Same tiles and fragment reads as conv_wgrad_kernel (DMA layout), but the staging is a THREE-deep ring of LDS stages filled by
// global -> LDS DMA with counted waits: stage st+2 is issued while stage st is consumed, so two stages (2 x 16-20 KB per
// workgroup) are in flight across every barrier.  The long-M layers of a ResNet (56x56 / 28x28 maps at batch 256) are
// HBM-latency problems for this kernel shape — 98 dependent "load a stage, barrier, 16 MFMAs" rounds per workgroup with 2
// workgroups per CU gave 0.9-1.3 TB/s —, and __syncthreads() drains vmcnt(0) while a DMA is in flight, so the ring needs the
// raw barrier + `s_waitcnt vmcnt(L)` form (L = DMA instructions per stage and thread: the NEXT stage may stay in flight).
//   iteration st:  wait vmcnt(L) -> stage st has landed (this wave's share); lgkmcnt(0) -> this wave's reads of stage st-1 retired
//                  s_barrier     -> every wave's share of stage st landed, nobody reads stage st-1 any more
//                  issue stage st+2 into the buffer of stage st-1;  fragment reads + MFMAs of stage st
// Tiles 64/128/256 x 64/128/256 (2 x 2 waves): a 64-channel layer takes a 64 x 256 (or 256 x 64) tile so that every barrier
// still covers 16 MFMAs per wave.
template <int TN, int TK, bool PWK>
__global__ __launch_bounds__(256) void conv_wgrad_ring_kernel(WgradArgs a) {
  constexpr int MS = 32, NST = 3;
  constexpr int CN = TN / 8, CK = TK / 8;
  constexpr int RPY = 256 / CN, RPX = 256 / CK;
  constexpr int YP = MS / RPY, XP = MS / RPX;
  static_assert(YP >= 1 && XP >= 1, "tile too narrow for 256 threads");
  constexpr int LOADS = YP + XP;                 // DMA instructions per thread and stage
  constexpr int YS = TN * 2, XS = TK * 2;
  constexpr int YBYTES = MS * YS, XBYTES = MS * XS, STAGE = YBYTES + XBYTES;
  // swizzle masks: row bits that do not change from one staging pass to the next (a thread keeps ONE logical chunk)
  constexpr int SWY = (CN / 2 - 1) < (RPY - 1) ? (CN / 2 - 1) : (RPY - 1);
  constexpr int SWX = (CK / 2 - 1) < (RPX - 1) ? (CK / 2 - 1) : (RPX - 1);
  constexpr int NT = TN / 32, KTL = TK / 32;

  extern __shared__ __attribute__((aligned(16))) char smem[];

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wv = tid >> 6;
  const int wn2 = wv & 1, wk2 = wv >> 1;

  const int ntile = a.tilesN * a.tilesK;
  const int id = tok_xcd_remap(blockIdx.x, ntile * a.splitM);
  const int split = id / ntile;
  const int t = id - split * ntile;
  const int tn = t / a.tilesK;
  const int tk = t - tn * a.tilesK;

  const int mstart = split * a.mchunk;
  const int mend = min(a.M, mstart + a.mchunk);
  const int steps = (mend - mstart + MS - 1) / MS;

  const int ycol = tid % CN, yrow = tid / CN;
  const int xcol = tid % CK, xrow = tid / CK;
  const int ylog = ycol ^ ((yrow & SWY) << 1);
  const int xlog = xcol ^ ((xrow & SWX) << 1);
  const int wave_u = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int yn = tn * TN + ylog * 8;
  const bool yn_ok = yn < a.K;

  const int k0 = tk * TK + xlog * 8;
  const int tap = k0 / a.C;
  const int kc0 = k0 - tap * a.C;
  const int kr = tap / a.S;
  const int ks = tap - kr * a.S;
  const bool k_ok = kr < a.R;

  // row cursor: output pixel of this thread's X rows; (image, p, q) are re-derived per step by magic-number division —
  // uniform straight-line code (the incremental "while (q >= Q)" cursor was a divergent loop per row and step)
  int xm[XP];
#pragma unroll
  for (int i = 0; i < XP; ++i) xm[i] = mstart + xrow + i * RPX;
  int ym = mstart + yrow;

  const __amdgpu_buffer_rsrc_t ysrd = __builtin_amdgcn_make_buffer_rsrc((void*)a.dy, 0, a.dy_bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t xsrd = __builtin_amdgcn_make_buffer_rsrc((void*)a.x, 0, a.x_bytes, 0x00020000);
  typedef __attribute__((address_space(3))) void lds_void;

  auto issue = [&](int dbuf) {
    char* Ydst = smem + dbuf * STAGE + wave_u * 1024;
    char* Xdst = smem + dbuf * STAGE + YBYTES + wave_u * 1024;
#pragma unroll
    for (int i = 0; i < YP; ++i) {
      const int m = ym + i * RPY;
      uint32_t off = (yn_ok && m < mend) ? (uint32_t)(m * a.K + yn) * 2u : 0xFFFFFFF0u;
      asm volatile("" : "+v"(off));      // one unconditional DMA per row (hipcc otherwise splits it into exec-masked halves)
      __builtin_amdgcn_raw_ptr_buffer_load_lds(ysrd, (lds_void*)(Ydst + i * RPY * YS), 16, off, 0, 0, 0);
    }
    ym += MS;
#pragma unroll
    for (int i = 0; i < XP; ++i) {
      uint32_t off;
      if (PWK) {
        off = (k_ok && xm[i] < mend) ? (uint32_t)(xm[i] * a.C + kc0) * 2u : 0xFFFFFFF0u;
      } else {
        const uint32_t mm = (uint32_t)min(xm[i], a.M - 1);
        const uint32_t b = magic_div(mm, a.pq_mul, a.pq_shift);
        const uint32_t rem = mm - b * (uint32_t)a.PQ;
        const uint32_t pp = magic_div(rem, a.q_mul, a.q_shift);
        const int hh = (int)pp * a.stride - a.pad + kr;
        const int ww = (int)(rem - pp * (uint32_t)a.Q) * a.stride - a.pad + ks;
        const bool ok = k_ok && xm[i] < mend && (unsigned)hh < (unsigned)a.H && (unsigned)ww < (unsigned)a.W;
        uint32_t goff = (uint32_t)(((int)b * a.HW + hh * a.W + ww) * a.C + kc0) * 2u;
        asm volatile("" : "+v"(goff));
        off = ok ? goff : 0xFFFFFFF0u;
      }
      asm volatile("" : "+v"(off));
      __builtin_amdgcn_raw_ptr_buffer_load_lds(xsrd, (lds_void*)(Xdst + i * RPX * XS), 16, off, 0, 0, 0);
      xm[i] += MS;
    }
  };

  const int g = lane >> 4, li = lane & 15;
  const int rrow = 4 * g + (li >> 2);
  const int yswz = ((rrow & SWY) << 1) << 4, xswz = ((rrow & SWX) << 1) << 4;
  auto ycolb = [&](int i) { return ((wn2 * (TN / 2) + i * 16 + (li & 3) * 4) * 2) ^ yswz; };
  auto xcolb = [&](int j) { return ((wk2 * (TK / 2) + j * 16 + (li & 3) * 4) * 2) ^ xswz; };

  f32x4 acc[NT][KTL];
#pragma unroll
  for (int i = 0; i < NT; ++i)
#pragma unroll
    for (int j = 0; j < KTL; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};

  // bias gradient = dy^T 1: the waves that own the first k-tile multiply their dy fragments with an all-ones operand as well
  // (NT extra MFMAs per stage; every column of the result tile holds the column sums) — no separate pass over dy
  const bool do_cs = a.cs != nullptr && tk == 0 && wk2 == 0;     // wave-uniform
  f32x4 csacc[NT];
#pragma unroll
  for (int i = 0; i < NT; ++i) csacc[i] = (f32x4){0.f, 0.f, 0.f, 0.f};
  const bf16 one_b = (bf16)1.0f;
  const bf16x8 ones = {one_b, one_b, one_b, one_b, one_b, one_b, one_b, one_b};

  typedef __attribute__((address_space(3))) char lds_char;
  const uint32_t lds_base = (uint32_t)(size_t)(lds_char*)smem;
  issue(0);
  issue(1);
  int cur = 0, nxt = 2;     // ring slots of stage st and of stage st + 2
  for (int st = 0; st < steps; ++st) {
    // stage st has landed once at most one stage's worth of this wave's DMAs is still outstanding
    asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)" ::"n"(LOADS) : "memory");
    __builtin_amdgcn_s_barrier();
    issue(nxt);
    // fragment reads as inline asm: for hipcc a `ds_read` behind an LDS-DMA in flight means "wait vmcnt(0)" (it cannot
    // prove the stages disjoint), which would drain the ring; the counted wait + barrier above is the real dependency
    const uint32_t Yb = lds_base + cur * STAGE;
    const uint32_t Xb = Yb + YBYTES;
    u32x2 ya[NT][2], xb[KTL][2];
#pragma unroll
    for (int i = 0; i < NT; ++i) {
      ya[i][0] = tr_read_asm(Yb + rrow * YS + ycolb(i));
      ya[i][1] = tr_read_asm(Yb + (rrow + 16) * YS + ycolb(i));
    }
#pragma unroll
    for (int j = 0; j < KTL; ++j) {
      xb[j][0] = tr_read_asm(Xb + rrow * XS + xcolb(j));
      xb[j][1] = tr_read_asm(Xb + (rrow + 16) * XS + xcolb(j));
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_sched_barrier(0);     // keeps the MFMAs below the wait (they only depend on registers)
    bf16x8 af[NT], bfr[KTL];
#pragma unroll
    for (int i = 0; i < NT; ++i) af[i] = __builtin_bit_cast(bf16x8, (u32x4){ya[i][0][0], ya[i][0][1], ya[i][1][0], ya[i][1][1]});
#pragma unroll
    for (int j = 0; j < KTL; ++j) bfr[j] = __builtin_bit_cast(bf16x8, (u32x4){xb[j][0][0], xb[j][0][1], xb[j][1][0], xb[j][1][1]});
#pragma unroll
    for (int i = 0; i < NT; ++i)
#pragma unroll
      for (int j = 0; j < KTL; ++j)
        acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[i], bfr[j], acc[i][j], 0, 0, 0);
    if (do_cs) {
#pragma unroll
      for (int i = 0; i < NT; ++i) csacc[i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[i], ones, csacc[i], 0, 0, 0);
    }
    cur = cur == NST - 1 ? 0 : cur + 1;
    nxt = nxt == NST - 1 ? 0 : nxt + 1;
  }
  // the two stages issued past the end (zeros) must have landed before this workgroup's LDS can be handed to another one
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");

  if (do_cs && li == 0) {      // column 0 of the ones-product: rows g*4 + r
#pragma unroll
    for (int i = 0; i < NT; ++i)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int n = tn * TN + wn2 * (TN / 2) + i * 16 + g * 4 + r;
        if (n < a.cs_cols) a.cs[(size_t)split * a.cs_stride + n] = csacc[i][r];
      }
  }

  float* out = a.ws + (size_t)split * a.ws_stride;
#pragma unroll
  for (int i = 0; i < NT; ++i) {
#pragma unroll
    for (int j = 0; j < KTL; ++j) {
      const int kcol = tk * TK + wk2 * (TK / 2) + j * 16 + li;
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int n = tn * TN + wn2 * (TN / 2) + i * 16 + g * 4 + r;
        if (n < a.K && kcol < a.Ktot) out[(size_t)n * a.Ktot + kcol] = acc[i][j][r];
      }
    }
  }
}


// ---- 256 x 256 tiles, eight waves (pointwise layers with wide outputs AND deep filters) ------------------------------------------
// What bounds conv_wgrad_ring_kernel<128, 128> on the transformer and late-ResNet layers is the L2 -> LDS staging path: a
// 128 x 128 tile stages (128 + 128) x 2 bytes per reduction row for 128 x 128 MACs, and every (n, k) tile of a layer stages ALL
// M rows of its two operand slices again — SwinV2-T stage-3 fc2 (384 x 1536 weights, 50 176 tokens): 36 tiles x 25.7 MB =
// 925 MB through a path that sustains ~10 TB/s chip-wide = the 93 us the launch takes (2.0 TB/s of HBM, MFMA util 0.23:
// "neither bound", profiles/r03_*).  A 256 x 256 tile stages half the bytes per MAC.  Same ring (three 32-row stages, counted
// vmcnt + raw barrier, transpose reads from row-major tiles) on 512 threads: waves 2 (n) x 4 (k), a wave owns 128 x 64 of the
// tile = 32 accumulator blocks, 24 transpose reads per 32 MFMAs.  One workgroup per CU (96 KB of LDS, ~200 registers); the
// split-M partial tiles are 256 KB each, so the plan aims at one workgroup per CU (256 slabs) — twice the partial-sum bytes of
// the 128 x 128 plan, against half the staged bytes.
template <int TN, int TK>
__global__ __launch_bounds__(512, 1) void conv_wgrad_ring8_kernel(WgradArgs a) {
  constexpr int MS = 32, NST = 3, NTHR = 512;
  constexpr int CN = TN / 8, CK = TK / 8;
  constexpr int RPY = NTHR / CN, RPX = NTHR / CK;
  constexpr int YP = MS / RPY, XP = MS / RPX;
  static_assert(YP >= 1 && XP >= 1 && MS % RPY == 0 && MS % RPX == 0, "tile too narrow for 512 threads");
  constexpr int LOADS = YP + XP;
  constexpr int YS = TN * 2, XS = TK * 2;
  constexpr int YBYTES = MS * YS, XBYTES = MS * XS, STAGE = YBYTES + XBYTES;
  constexpr int SWY = (CN / 2 - 1) < (RPY - 1) ? (CN / 2 - 1) : (RPY - 1);
  constexpr int SWX = (CK / 2 - 1) < (RPX - 1) ? (CK / 2 - 1) : (RPX - 1);
  constexpr int NT = TN / 32, KTL = TK / 64;       // 16-wide blocks per wave: TN / 2 rows of n, TK / 4 columns of k

  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wv = tid >> 6;
  const int wn2 = wv & 1, wk4 = wv >> 1;

  const int ntile = a.tilesN * a.tilesK;
  const int id = tok_xcd_remap(blockIdx.x, ntile * a.splitM);
  const int split = id / ntile;
  const int t = id - split * ntile;
  const int tn = t / a.tilesK;
  const int tk = t - tn * a.tilesK;
  const int mstart = split * a.mchunk;
  const int mend = min(a.M, mstart + a.mchunk);
  const int steps = (mend - mstart + MS - 1) / MS;

  const int ycol = tid % CN, yrow = tid / CN;
  const int xcol = tid % CK, xrow = tid / CK;
  const int ylog = ycol ^ ((yrow & SWY) << 1);
  const int xlog = xcol ^ ((xrow & SWX) << 1);
  const int wave_u = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int yn = tn * TN + ylog * 8;
  const bool yn_ok = yn < a.K;
  const int kc0 = tk * TK + xlog * 8;              // pointwise: the k column IS the input channel
  const bool k_ok = kc0 < a.C;
  int xm[XP];
#pragma unroll
  for (int i = 0; i < XP; ++i) xm[i] = mstart + xrow + i * RPX;
  int ym = mstart + yrow;

  const __amdgpu_buffer_rsrc_t ysrd = __builtin_amdgcn_make_buffer_rsrc((void*)a.dy, 0, a.dy_bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t xsrd = __builtin_amdgcn_make_buffer_rsrc((void*)a.x, 0, a.x_bytes, 0x00020000);
  typedef __attribute__((address_space(3))) void lds_void;
  auto issue = [&](int dbuf) {
    char* Ydst = smem + dbuf * STAGE + wave_u * 1024;
    char* Xdst = smem + dbuf * STAGE + YBYTES + wave_u * 1024;
#pragma unroll
    for (int i = 0; i < YP; ++i) {
      const int m = ym + i * RPY;
      uint32_t off = (yn_ok && m < mend) ? (uint32_t)(m * a.K + yn) * 2u : 0xFFFFFFF0u;
      asm volatile("" : "+v"(off));
      __builtin_amdgcn_raw_ptr_buffer_load_lds(ysrd, (lds_void*)(Ydst + i * RPY * YS), 16, off, 0, 0, 0);
    }
    ym += MS;
#pragma unroll
    for (int i = 0; i < XP; ++i) {
      uint32_t off = (k_ok && xm[i] < mend) ? (uint32_t)(xm[i] * a.C + kc0) * 2u : 0xFFFFFFF0u;
      asm volatile("" : "+v"(off));
      __builtin_amdgcn_raw_ptr_buffer_load_lds(xsrd, (lds_void*)(Xdst + i * RPX * XS), 16, off, 0, 0, 0);
      xm[i] += MS;
    }
  };

  const int g = lane >> 4, li = lane & 15;
  const int rrow = 4 * g + (li >> 2);
  const int yswz = ((rrow & SWY) << 1) << 4, xswz = ((rrow & SWX) << 1) << 4;
  auto ycolb = [&](int i) { return ((wn2 * (TN / 2) + i * 16 + (li & 3) * 4) * 2) ^ yswz; };
  auto xcolb = [&](int j) { return ((wk4 * (TK / 4) + j * 16 + (li & 3) * 4) * 2) ^ xswz; };

  f32x4 acc[NT][KTL];
#pragma unroll
  for (int i = 0; i < NT; ++i)
#pragma unroll
    for (int j = 0; j < KTL; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
  const bool do_cs = a.cs != nullptr && tk == 0 && wk4 == 0;     // wave-uniform: bias gradient = dy^T 1 (see the ring kernel)
  f32x4 csacc[NT];
#pragma unroll
  for (int i = 0; i < NT; ++i) csacc[i] = (f32x4){0.f, 0.f, 0.f, 0.f};
  const bf16 one_b = (bf16)1.0f;
  const bf16x8 ones = {one_b, one_b, one_b, one_b, one_b, one_b, one_b, one_b};

  typedef __attribute__((address_space(3))) char lds_char;
  const uint32_t lds_base = (uint32_t)(size_t)(lds_char*)smem;
  issue(0);
  issue(1);
  int cur = 0, nxt = 2;
  for (int st = 0; st < steps; ++st) {
    asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)" ::"n"(LOADS) : "memory");
    __builtin_amdgcn_s_barrier();
    issue(nxt);
    const uint32_t Yb = lds_base + cur * STAGE;
    const uint32_t Xb = Yb + YBYTES;
    u32x2 ya[NT][2], xb[KTL][2];
#pragma unroll
    for (int j = 0; j < KTL; ++j) {
      xb[j][0] = tr_read_asm(Xb + rrow * XS + xcolb(j));
      xb[j][1] = tr_read_asm(Xb + (rrow + 16) * XS + xcolb(j));
    }
#pragma unroll
    for (int i = 0; i < NT; ++i) {
      ya[i][0] = tr_read_asm(Yb + rrow * YS + ycolb(i));
      ya[i][1] = tr_read_asm(Yb + (rrow + 16) * YS + ycolb(i));
    }
    // (24 transpose reads: the 4-bit lgkmcnt counter cannot count them all; the first 8 — every x fragment — are waited for
    //  together with the first half of the dy fragments, then the rest)
    bf16x8 af[NT], bfr[KTL];
    asm volatile("s_waitcnt lgkmcnt(8)" ::: "memory");
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int j = 0; j < KTL; ++j) bfr[j] = __builtin_bit_cast(bf16x8, (u32x4){xb[j][0][0], xb[j][0][1], xb[j][1][0], xb[j][1][1]});
#pragma unroll
    for (int i = 0; i < NT / 2; ++i) {
      af[i] = __builtin_bit_cast(bf16x8, (u32x4){ya[i][0][0], ya[i][0][1], ya[i][1][0], ya[i][1][1]});
#pragma unroll
      for (int j = 0; j < KTL; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[i], bfr[j], acc[i][j], 0, 0, 0);
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int i = NT / 2; i < NT; ++i) {
      af[i] = __builtin_bit_cast(bf16x8, (u32x4){ya[i][0][0], ya[i][0][1], ya[i][1][0], ya[i][1][1]});
#pragma unroll
      for (int j = 0; j < KTL; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[i], bfr[j], acc[i][j], 0, 0, 0);
    }
    if (do_cs) {
#pragma unroll
      for (int i = 0; i < NT; ++i) csacc[i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[i], ones, csacc[i], 0, 0, 0);
    }
    cur = cur == NST - 1 ? 0 : cur + 1;
    nxt = nxt == NST - 1 ? 0 : nxt + 1;
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");

  if (do_cs && li == 0) {
#pragma unroll
    for (int i = 0; i < NT; ++i)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int n = tn * TN + wn2 * (TN / 2) + i * 16 + g * 4 + r;
        if (n < a.cs_cols) a.cs[(size_t)split * a.cs_stride + n] = csacc[i][r];
      }
  }
  float* out = a.ws + (size_t)split * a.ws_stride;
#pragma unroll
  for (int i = 0; i < NT; ++i) {
#pragma unroll
    for (int j = 0; j < KTL; ++j) {
      const int kcol = tk * TK + wk4 * (TK / 4) + j * 16 + li;
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int n = tn * TN + wn2 * (TN / 2) + i * 16 + g * 4 + r;
        if (n < a.K && kcol < a.Ktot) out[(size_t)n * a.Ktot + kcol] = acc[i][j][r];
      }
    }
  }
}


// ---- tap-stationary 3x3 variant --------------------------------------------------------------------------------------------
// The two kernels above treat the nine taps as independent column tiles: every (tap, channel) tile stages the SAME dy rows
// again, decomposes the SAME output pixels into (image, p, q) again, and a barrier round covers 16 MFMAs per wave — on the 3x3
// layers they sit at ~20 % MFMA utilisation whatever the shape (ResNet-50 56x56 ... 7x7: 100 us each), bounded by address
// arithmetic and load latency, not by LDS or the matrix pipe.  Here one workgroup owns 64 output channels x 64 input channels
// x ALL NINE taps:
//   * a stage = 32 output pixels: dy[32][64] once + the nine shifted x[32][64] tiles (40 KB), three stages in a DMA ring
//     (two in flight across every barrier: 80 KB of loads per CU, enough to cover HBM latency with ONE resident workgroup);
//   * thread t owns pixel t/8 and 16-byte chunk t%8 of every tile: one (image, p, q) decomposition per stage, the nine x
//     offsets are base + tap displacement (uniform) under 3 + 3 precomputed bounds flags;
//   * the four waves split the 36 sixteen-column tiles (9 each); every wave reads all four dy fragments (8 transpose reads)
//     + its 18 x half-fragments and issues 36 MFMAs per stage: 2.25x the MFMAs per barrier and per fragment byte.
// Output layout, split-M partials and the reduce kernels are those of the other variants.
template <int DUMMY>
__global__ __launch_bounds__(256, 1) void conv_wgrad_taps_kernel(WgradArgs a) {
  constexpr int MS = 32, NST = 3, TN = 64, TC = 64, TAPS = 9;
  constexpr int TILE = MS * 128;                 // one 32 x 64 bf16 tile: 4 KB, one DMA instruction per thread
  constexpr int STAGE = (1 + TAPS) * TILE;       // 40 KB
  constexpr int LOADS = 1 + TAPS;
  constexpr int NT = 4, KTL = 9;                 // per wave: 64 output channels x 144 (tap, channel) columns

  extern __shared__ __attribute__((aligned(16))) char smem[];

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);

  const int tilesC = a.tilesK;                   // input-channel tiles (all taps inside)
  const int ntile = a.tilesN * tilesC;
  const int id = tok_xcd_remap(blockIdx.x, ntile * a.splitM);
  const int split = id / ntile;
  const int t = id - split * ntile;
  const int tn = t / tilesC;
  const int ct = t - tn * tilesC;

  const int mstart = split * a.mchunk;
  const int mend = min(a.M, mstart + a.mchunk);
  const int steps = (mend - mstart + MS - 1) / MS;

  const int row = tid >> 3, cc = tid & 7;
  const int clog = cc ^ ((row & 3) << 1);        // logical chunk fetched into slot cc (source-side swizzle)
  const int yn = tn * TN + clog * 8;
  const bool yn_ok = yn < a.K;
  const int cx = ct * TC + clog * 8;
  const bool cx_ok = cx < a.C;

  const __amdgpu_buffer_rsrc_t ysrd = __builtin_amdgcn_make_buffer_rsrc((void*)a.dy, 0, a.dy_bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t xsrd = __builtin_amdgcn_make_buffer_rsrc((void*)a.x, 0, a.x_bytes, 0x00020000);
  typedef __attribute__((address_space(3))) void lds_void;

  int mcur = mstart + row;
  auto issue = [&](int slot) {
    char* base = smem + slot * STAGE + wv * 1024;
    const int m = mcur;
    const bool mok = m < mend;
    {
      uint32_t off = (uint32_t)(m * a.K + yn) * 2u;
      asm volatile("" : "+v"(off));
      off = (mok && yn_ok) ? off : 0xFFFFFFF0u;
      asm volatile("" : "+v"(off));
      __builtin_amdgcn_raw_ptr_buffer_load_lds(ysrd, (lds_void*)base, 16, off, 0, 0, 0);
    }
    const uint32_t mm = (uint32_t)min(m, a.M - 1);
    const uint32_t b = magic_div(mm, a.pq_mul, a.pq_shift);
    const uint32_t rem = mm - b * (uint32_t)a.PQ;
    const uint32_t pp = magic_div(rem, a.q_mul, a.q_shift);
    const int h0 = (int)pp * a.stride - a.pad;
    const int w0 = (int)(rem - pp * (uint32_t)a.Q) * a.stride - a.pad;
    const int xbase = (((int)b * a.H + h0) * a.W + w0) * a.C + cx;
    const bool live = mok && cx_ok;
    bool hok[3], wok[3];
#pragma unroll
    for (int k = 0; k < 3; ++k) {
      hok[k] = live && (unsigned)(h0 + k) < (unsigned)a.H;
      wok[k] = (unsigned)(w0 + k) < (unsigned)a.W;
    }
#pragma unroll
    for (int tap = 0; tap < TAPS; ++tap) {
      const int kr = tap / 3, ks = tap - kr * 3;
      uint32_t off = (uint32_t)(xbase + (kr * a.W + ks) * a.C) * 2u;
      asm volatile("" : "+v"(off));
      off = (hok[kr] && wok[ks]) ? off : 0xFFFFFFF0u;
      asm volatile("" : "+v"(off));
      __builtin_amdgcn_raw_ptr_buffer_load_lds(xsrd, (lds_void*)(base + (1 + tap) * TILE), 16, off, 0, 0, 0);
    }
    mcur += MS;
  };

  const int g = lane >> 4, li = lane & 15;
  const int rrow = 4 * g + (li >> 2);
  const int swz = ((rrow & 3) << 1) << 4;
  const uint32_t rbase = (uint32_t)(rrow * 128);
  auto colb = [&](int tile16) { return (uint32_t)(((tile16 * 16 + (li & 3) * 4) * 2) ^ swz); };

  f32x4 acc[NT][KTL];
#pragma unroll
  for (int i = 0; i < NT; ++i)
#pragma unroll
    for (int j = 0; j < KTL; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};

  typedef __attribute__((address_space(3))) char lds_char;
  const uint32_t lds_base = (uint32_t)(size_t)(lds_char*)smem;
  issue(0);
  issue(1);
  int cur = 0, nxt = 2;
  for (int st = 0; st < steps; ++st) {
    asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)" ::"n"(LOADS) : "memory");
    __builtin_amdgcn_s_barrier();
    issue(nxt);
    const uint32_t Yb = lds_base + cur * STAGE + rbase;
    const uint32_t Xb = Yb + TILE;
    u32x2 ya[NT][2], xb[KTL][2];
#pragma unroll
    for (int i = 0; i < NT; ++i) {
      ya[i][0] = tr_read_asm(Yb + colb(i));
      ya[i][1] = tr_read_asm(Yb + 16 * 128 + colb(i));
    }
#pragma unroll
    for (int j = 0; j < KTL; ++j) {
      const int J = wv * KTL + j;                 // global 16-column tile: tap J / 4, channels (J % 4) * 16 ...
      const uint32_t ad = Xb + (uint32_t)((J >> 2) * TILE) + colb(J & 3);
      xb[j][0] = tr_read_asm(ad);
      xb[j][1] = tr_read_asm(ad + 16 * 128);
    }
    bf16x8 af[NT];
    // first five column tiles as soon as their fragments are in (2 * (KTL - 5) reads still in flight), the rest behind them
    asm volatile("s_waitcnt lgkmcnt(%0)" ::"n"(2 * (KTL - 5)) : "memory");
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int i = 0; i < NT; ++i) af[i] = __builtin_bit_cast(bf16x8, (u32x4){ya[i][0][0], ya[i][0][1], ya[i][1][0], ya[i][1][1]});
#pragma unroll
    for (int j = 0; j < 5; ++j) {
      const bf16x8 bfr = __builtin_bit_cast(bf16x8, (u32x4){xb[j][0][0], xb[j][0][1], xb[j][1][0], xb[j][1][1]});
#pragma unroll
      for (int i = 0; i < NT; ++i) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[i], bfr, acc[i][j], 0, 0, 0);
    }
    __builtin_amdgcn_sched_barrier(0);
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int j = 5; j < KTL; ++j) {
      const bf16x8 bfr = __builtin_bit_cast(bf16x8, (u32x4){xb[j][0][0], xb[j][0][1], xb[j][1][0], xb[j][1][1]});
#pragma unroll
      for (int i = 0; i < NT; ++i) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[i], bfr, acc[i][j], 0, 0, 0);
    }
    cur = cur == NST - 1 ? 0 : cur + 1;
    nxt = nxt == NST - 1 ? 0 : nxt + 1;
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");

  float* out = a.ws + (size_t)split * a.K * a.Ktot;
#pragma unroll
  for (int j = 0; j < KTL; ++j) {
    const int J = wv * KTL + j;
    const int cin = ct * TC + (J & 3) * 16 + li;
    const int kcol = (J >> 2) * a.C + cin;
#pragma unroll
    for (int i = 0; i < NT; ++i)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int n = tn * TN + i * 16 + g * 4 + r;
        if (n < a.K && cin < a.C) out[(size_t)n * a.Ktot + kcol] = acc[i][j][r];
      }
  }
}


// ---- shared-window form of the tap-stationary kernel (3x3, stride 1, padding 1) ---------------------------------------------
// Measured on the kernel above: 36 MFMAs per barrier did not move the 3x3 layers (113-124 us, ~500 TF/s) — the nine shifted
// x tiles of a stage are 36 of its 40 KB, and 25 k workgroup-stages x 40 KB in 113 us is 9 TB/s of L2 -> LDS traffic for 51 MB
// of distinct data: the layer is bound by re-staging the same pixels nine times.  With stride 1 / padding 1 the input pixel of
// output pixel m and tap (kr, ks) is m + (kr - 1) W + (ks - 1) in the flat [B H W] pixel index, image borders aside: the
// three taps of a filter row read ONE window of 34 consecutive pixels at row offsets 0, 1, 2.  A stage therefore holds
// dy[32][64] + three 34-pixel windows (102 rows, 13 KB instead of 36) and the nine taps are nine row offsets of the transpose
// reads.  Border taps (p + kr - 1 or q + ks - 1 outside the image) are dropped on the READ side: a lane's transpose read
// covers one reduction row, so a lane whose row is invalid for the tap reads a zeroed LDS line instead.
// CT = 16-channel tiles per side of the workgroup tile: 4 -> 64 x 64 channels (four compute waves, nine column tiles each, taps
// straddle waves), 3 -> 48 x 48 (HRNet's 48 / 96-wide branches: three compute waves, one filter row each; the fourth wave only
// stages).  Rows keep the 128-byte pitch; the 48-wide form leaves the last two 16-byte slots of a row unfetched.
// NST = stages of the ring (NST - 1 in flight ahead of the one being read): 4 x 20 KB = 80 KB, two workgroups per CU.
// KTL = column tiles (tap, 16 channels) per compute wave: 9 (CT = 4: 36 tiles on four waves; CT = 3: 27 tiles on three waves, the
// fourth only stages) or 7 (CT = 3: 27 tiles on FOUR waves, 7 + 7 + 7 + 6: a quarter less MFMA / transpose-read chain per barrier).
// ---- the window kernel, software-pipelined inside the wave (round 6) -------------------------------------------------------------
// Round 5's form of this kernel (conv_wgrad_win_kernel, in the history) ran one wave per SIMD at the step's split target (256
// workgroups), and its stage was a serial chain:
// barrier -> five DMA issues -> border masks -> 26 transpose reads -> lgkmcnt -> 36 MFMAs (576 of the stage's ~1700 cycles;
// profiles/r05_layer_tables.txt: 411-644 TF/s).  Nothing overlaps the MFMA phase because the fragments it multiplies are the
// ones it just waited for.  Here the fragments of stage st + 1 are read WHILE stage st is multiplied from registers: the x
// fragment of column tile j is re-read in place right after the NT MFMAs that consumed it, only the dy fragments (used by every
// MFMA of the stage) need a second register set (the stage body is instantiated twice, roles swapped).  The five DMA issues of
// the stage that refills the just-freed slot ride behind the first five column tiles, the read bases of stage st + 2 (border
// handling) are formed behind the last ones.  And the stage's VALU count — what a single wave per SIMD cannot hide — is cut:
//  * DMA offsets are running sums (one add per piece and stage); rows past the tensor fall to the buffer range check, rows
//    past the split's chunk are fetched and multiplied by the zero line (their reduction rows are invalid for every tap);
//  * the (row, column) of a lane's two reduction rows are advanced by 32 pixels with one wrap each (a.adv_q = 32 % Q,
//    a.adv_p = (32 / Q) % P from the host) instead of two magic-number divisions per stage;
//  * a wave tests only the border conditions of ITS three taps (compile-time tap numbers), not a 9-bit mask.
// Ring, stage layout, border handling, summation order per accumulator (stage order) and hence every output bit equal
// conv_wgrad_win_kernel's.
template <int CT, int NST, int KTLP>
__global__ __launch_bounds__(256, 2) void conv_wgrad_winp_kernel(WgradArgs a) {
  constexpr int MS = 32, TN = 16 * CT, TC = 16 * CT, TAPS = 9;
  constexpr int WIN = MS + 2;
  constexpr int YT = MS * 128;
  constexpr int XT = 128 * 128;
  constexpr int STAGE = YT + XT;
  constexpr int LOADS = 1 + 4;
  constexpr int NT = CT, KTL = KTLP;
  constexpr int NTILES = TAPS * CT;
  constexpr int NCW = (NTILES + KTL - 1) / KTL;
  static_assert(NCW == 4 && KTL >= 7 && NST >= 3, "four compute waves");

  extern __shared__ __attribute__((aligned(16))) char smem[];

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);

  const int tilesC = a.tilesK;
  const int ntile = a.tilesN * tilesC;
  const int id = tok_xcd_remap(blockIdx.x, ntile * a.splitM);
  const int split = id / ntile;
  const int t = id - split * ntile;
  const int tn = t / tilesC;
  const int ct = t - tn * tilesC;

  const int mstart = split * a.mchunk;
  const int mend = min(a.M, mstart + a.mchunk);
  const int steps = (mend - mstart + MS - 1) / MS;

  // (everything lane-dependent lives INSIDE the per-wave instantiation: nothing of one copy stays live across another's stage loop)
  auto run = [&](auto WVC) {
  const int row = tid >> 3, cc = tid & 7;
  const int clog = cc ^ ((row & 3) << 1);
  const int yn = tn * TN + clog * 8;
  const bool yn_ok = yn < a.K && clog * 8 < TN;
  const int cx = ct * TC + clog * 8;
  const bool cx_ok = cx < a.C && clog * 8 < TC;
  // running byte offsets of this thread's five DMA pieces (0: dy row, 1..4: x window rows R = row + 32 j).  A piece the thread
  // never fetches (padding columns, rows 102..127 = the zero lines) starts at 2^31 + 2^30: the host serves this kernel only
  // for tensors below 2^30 bytes, so such an offset stays past the buffer range check while it advances with the others
  // (uniform increments: no per-piece increment registers in a loop that runs at the 256-register limit)
  uint32_t doff[LOADS];
  const uint32_t dinc_y = (uint32_t)(MS * a.K) * 2u;
  const uint32_t dinc_x = (uint32_t)(MS * a.C) * 2u;
  doff[0] = yn_ok ? (uint32_t)((mstart + row) * a.K + yn) * 2u : 0xC0000000u;
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const int R = row + 32 * j;
    const int win = R / WIN;
    const bool live = cx_ok && R < 3 * WIN;
    const int pix = mstart + (win - 1) * a.W - 1 + (R - win * WIN);      // may be negative: wraps past the range check
    doff[1 + j] = live ? (uint32_t)(pix * a.C + cx) * 2u : 0xC0000000u;
  }

  const __amdgpu_buffer_rsrc_t ysrd = __builtin_amdgcn_make_buffer_rsrc((void*)a.dy, 0, a.dy_bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t xsrd = __builtin_amdgcn_make_buffer_rsrc((void*)a.x, 0, a.x_bytes, 0x00020000);
  typedef __attribute__((address_space(3))) void lds_void;

  // piece `pc` of the next stage to fetch into ring slot `slot`
  auto issue_piece = [&](int slot, int pc) {
    char* base = smem + slot * STAGE + wv * 1024;
    if (pc == 0) __builtin_amdgcn_raw_ptr_buffer_load_lds(ysrd, (lds_void*)base, 16, doff[0], 0, 0, 0);
    else __builtin_amdgcn_raw_ptr_buffer_load_lds(xsrd, (lds_void*)(base + YT + (pc - 1) * 4096), 16, doff[pc], 0, 0, 0);
    doff[pc] += pc == 0 ? dinc_y : dinc_x;
  };

  const int rrow = 4 * (lane >> 4) + ((lane & 15) >> 2);
  const uint32_t cq = (uint32_t)((lane & 3) * 8);
  typedef __attribute__((address_space(3))) char lds_char;
  const uint32_t lds_base = (uint32_t)(size_t)(lds_char*)smem;     // 0: the kernel has no static LDS (128-byte rows are XORed below)
  // LDS byte address of this lane's transpose read = (row base | lane's 8-byte column quarter | row swizzle) ^ (32 * tile):
  // the swizzle of a 128-byte row (its 16-byte chunks XOR ((row & 3) << 1)) moves 32-byte tiles as wholes
  const uint32_t yrc = (uint32_t)(rrow * 128) | cq | (uint32_t)(((rrow & 3) << 1) << 4);

  // (row, column) of this lane's two reduction rows and the index of the first, one stage BEHIND the stage whose read bases
  // are formed next (the body advances them first)
  int pp[2], qq[2], mm0 = mstart + rrow;
#pragma unroll
  for (int h = 0; h < 2; ++h) {
    const uint32_t mc = (uint32_t)min(mm0 + 16 * h, a.M - 1);
    const uint32_t b = magic_div(mc, a.pq_mul, a.pq_shift);
    const uint32_t rem = mc - b * (uint32_t)a.PQ;
    const uint32_t p0 = magic_div(rem, a.q_mul, a.q_shift);
    pp[h] = (int)p0; qq[h] = (int)(rem - p0 * (uint32_t)a.Q);
  }
  const int mend16 = mend - 16;

  constexpr int W0 = decltype(WVC)::value;
  constexpr int T0 = (W0 * KTL) / CT;
  // tap index (0..2, relative to T0) of the wave's column tile j, and the last tile that reads tap k's bases
  auto tap_of = [](int j) constexpr { return (W0 * KTL + j) / CT - T0; };
  auto last_j = [&](int k) constexpr { int l = -1; for (int j = 0; j < KTL && W0 * KTL + j < NTILES; ++j) if (tap_of(j) == k) l = j; return l; };
  // per tap: lane part of the read address inside a stage's x region (row of the tap's window | column quarter | swizzle)
  uint32_t xrc[3];
#pragma unroll
  for (int k = 0; k < 3; ++k) {
    const int tap = T0 + k < TAPS ? T0 + k : TAPS - 1;
    const int kr = tap / 3, ks = tap - kr * 3;
    const int R = kr * WIN + ks + rrow;
    xrc[k] = (uint32_t)(R * 128) | cq | (uint32_t)(((R & 3) << 1) << 4);
  }
  f32x4 acc[NT][KTL];
#pragma unroll
  for (int i = 0; i < NT; ++i)
#pragma unroll
    for (int j = 0; j < KTL; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};

  // the two (half) read bases of tap k for the stage at (pp, qq, mm0) in ring slot `slot`: the tap's row of the window, or the
  // slot's zero line where the lane's reduction row is outside the image for that tap / past the chunk
  auto bases_tap = [&](int slot, int k, uint32_t (&tb)[3][2]) {
    const uint32_t Xb = lds_base + slot * STAGE + YT;
    const uint32_t zero_line = Xb + 127 * 128;
    const int tap = T0 + k < TAPS ? T0 + k : TAPS - 1;
    const int kr = tap / 3, ks = tap - kr * 3;
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      bool ok = h == 0 ? mm0 < mend : mm0 < mend16;
      if (kr == 0) ok = ok && pp[h] >= 1;
      if (kr == 2) ok = ok && pp[h] <= a.H - 2;
      if (ks == 0) ok = ok && qq[h] >= 1;
      if (ks == 2) ok = ok && qq[h] <= a.W - 2;
      tb[k][h] = ok ? xrc[k] + (Xb + (uint32_t)(h * 16 * 128)) : zero_line;
    }
  };
  auto advance = [&]() {
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      int q = qq[h] + a.adv_q;
      const bool wq = q >= a.Q;
      q -= wq ? a.Q : 0;
      int p = pp[h] + a.adv_p + (wq ? 1 : 0);
      p -= p >= a.P ? a.P : 0;
      qq[h] = q; pp[h] = p;
    }
    mm0 += MS;
  };
  auto read_y = [&](int slot, int i, u32x2 (&ya)[NT][2]) {
    const uint32_t ad = (yrc + (lds_base + slot * STAGE)) ^ (uint32_t)(i * 32);
    asm volatile("ds_read_b64_tr_b16 %0, %1" : "=v"(ya[i][0]) : "v"(ad));
    asm volatile("ds_read_b64_tr_b16 %0, %1 offset:2048" : "=v"(ya[i][1]) : "v"(ad));
  };
  // (a tile past the wave's share is NOT read: the result of an asm read nobody consumes is a dead register to the compiler,
  //  which hands it to another value while the LDS data is still on its way — seen as sporadically zeroed read bases)
  auto read_x = [&](const uint32_t (&tb)[3][2], int j, u32x2 (&xb)[KTL][2]) {
    if (W0 * KTL + j >= NTILES) return;
    const int J = W0 * KTL + j;
    const int k = J / CT - T0;
    const uint32_t tl = (uint32_t)((J % CT) * 32);
    xb[j][0] = tr_read_asm(tb[k][0] ^ tl);
    xb[j][1] = tr_read_asm(tb[k][1] ^ tl);
  };

  u32x2 yA[NT][2], yB[NT][2], xb[KTL][2];
  uint32_t tb[3][2];

  // prologue: the whole ring in flight, fragments of stage 0 into registers, read bases of stage 1
#pragma unroll
  for (int s_ = 0; s_ < NST; ++s_)
#pragma unroll
    for (int pc = 0; pc < LOADS; ++pc) issue_piece(s_, pc);
  asm volatile("s_waitcnt vmcnt(%0)" ::"n"((NST - 1) * LOADS) : "memory");
  __builtin_amdgcn_s_barrier();
#pragma unroll
  for (int k = 0; k < 3; ++k) bases_tap(0, k, tb);
#pragma unroll
  for (int i = 0; i < NT; ++i) read_y(0, i, yA);
#pragma unroll
  for (int j = 0; j < KTL; ++j) read_x(tb, j, xb);
  advance();
#pragma unroll
  for (int k = 0; k < 3; ++k) bases_tap(1 % NST, k, tb);
  int rd = 1 % NST, wr = 0;                        // slot of stage st + 1 (read in iteration st), slot refilled in iteration st

  // one stage: multiply stage st from (ycur, xb) while (ynxt, xb) are refilled from stage st + 1 (its bases are in tb); tap k's
  // bases are replaced by those of stage st + 2 right behind the last read that uses them
  auto body = [&](u32x2 (&ycur)[NT][2], u32x2 (&ynxt)[NT][2]) {
    // stage st + 1 landed (this wave's pieces), every read of stage st retired -> after the barrier: all pieces, slot `wr` free
    asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)" ::"n"((NST - 2) * LOADS) : "memory");
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_sched_barrier(0);
    const int rd2 = rd == NST - 1 ? 0 : rd + 1;
    bf16x8 af[NT];
#pragma unroll
    for (int i = 0; i < NT; ++i) af[i] = __builtin_bit_cast(bf16x8, (u32x4){ycur[i][0][0], ycur[i][0][1], ycur[i][1][0], ycur[i][1][1]});
#pragma unroll
    for (int j = 0; j < KTL; ++j) {
      if (W0 * KTL + j < NTILES) {
        const bf16x8 bfr = __builtin_bit_cast(bf16x8, (u32x4){xb[j][0][0], xb[j][0][1], xb[j][1][0], xb[j][1][1]});
#pragma unroll
        for (int i = 0; i < NT; ++i) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[i], bfr, acc[i][j], 0, 0, 0);
      }
      if (j < LOADS) issue_piece(wr, j);
      if (j == 0) advance();
      if (j >= 1 && (j - 1) * 2 < NT) { read_y(rd, (j - 1) * 2, ynxt); if ((j - 1) * 2 + 1 < NT) read_y(rd, (j - 1) * 2 + 1, ynxt); }
      read_x(tb, j, xb);
#pragma unroll
      for (int k = 0; k < 3; ++k)
        if (last_j(k) == j) bases_tap(rd2, k, tb);
      __builtin_amdgcn_sched_barrier(0);          // each column tile's MFMAs keep their own fillers
    }
    rd = rd2;
    wr = wr == NST - 1 ? 0 : wr + 1;
  };
  for (int st = 0; st < steps; st += 2) {
    body(yA, yB);
    if (st + 1 < steps) body(yB, yA);
  }
  asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");

  // (the lane id is taken again here: nothing of the store addressing stays live across the stage loop, which runs at the
  //  256-register limit)
  int lane2;
  asm volatile("v_mbcnt_lo_u32_b32 %0, -1, 0\n\tv_mbcnt_hi_u32_b32 %0, -1, %0" : "=v"(lane2));
  const int g = lane2 >> 4, li = lane2 & 15;
  float* out = a.ws + (size_t)split * a.K * a.Ktot;
#pragma unroll
  for (int j = 0; j < KTL; ++j) {
    const int J = W0 * KTL + j;
    if (J >= NTILES) continue;
    const int cin = ct * TC + (J % CT) * 16 + li;
    const int kcol = (J / CT) * a.C + cin;
#pragma unroll
    for (int i = 0; i < NT; ++i)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int n = tn * TN + i * 16 + g * 4 + r;
        if (n < a.K && cin < a.C) out[(size_t)n * a.Ktot + kcol] = acc[i][j][r];
      }
  }
  };   // run
  if (wv == 0) run(std::integral_constant<int, 0>{});
  else if (wv == 1) run(std::integral_constant<int, 1>{});
  else if (wv == 2) run(std::integral_constant<int, 2>{});
  else run(std::integral_constant<int, 3>{});
}

// dw[k][r][s][c] (+)= sum_split ws[split][k][r][s_pad][c_pad]
__global__ void wgrad_reduce_kernel(const float* __restrict__ ws, float* __restrict__ dw, int splitM,
                                    int k_real, int R, int S, int c_real, int K, int S_pad,
                                    int C_pad, int accumulate) {
  const size_t total = (size_t)k_real * R * S * c_real;
  const size_t slab = (size_t)K * R * S_pad * C_pad;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total;
       i += (size_t)gridDim.x * blockDim.x) {
    const int c = (int)(i % c_real);
    size_t rest = i / c_real;
    const int s = (int)(rest % S);
    rest /= S;
    const int r = (int)(rest % R);
    const int k = (int)(rest / R);
    const size_t src = (((size_t)k * R + r) * S_pad + s) * C_pad + c;
    float t = 0.f;
    for (int sp = 0; sp < splitM; sp += 8) {       // 8 loads in flight, additions in split order
      float v[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) v[u] = ws[(size_t)(sp + u < splitM ? sp + u : splitM - 1) * slab + src];
#pragma unroll
      for (int u = 0; u < 8; ++u) t += sp + u < splitM ? v[u] : 0.f;
    }
    dw[i] = accumulate ? dw[i] + t : t;
  }
}

// no padding anywhere: dw and the partial slabs share one flat index space.  A block owns 64
// consecutive floats (16 float4 columns); its 16 thread-rows stride the split dimension so the
// partial slabs are read with independent, coalesced 256-byte segments; fixed-order LDS tree.
// `extra` (may be null): the floats [slab, slab + extra_pad) of every split are a second, short vector (the bias-gradient
// partials of tok_conv_wgrad_bias) reduced by the same launch into extra[0 .. extra_n) (+= if extra_acc); stride = floats
// between splits (slab + extra_pad, or slab).
template <int U>
__device__ __forceinline__ void reduce_slabs(const float* __restrict__ ws, size_t stride, size_t i, int row, int splitM, float4& t) {
  for (int sp = row; sp < splitM; sp += 16 * U) {
    float4 v[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int su = sp + 16 * u;
      v[u] = reinterpret_cast<const float4*>(ws + (size_t)(su < splitM ? su : splitM - 1) * stride)[i];
    }
#pragma unroll
    for (int u = 0; u < U; ++u) {
      if (sp + 16 * u < splitM) { t.x += v[u].x; t.y += v[u].y; t.z += v[u].z; t.w += v[u].w; }
    }
  }
}

__global__ __launch_bounds__(256) void wgrad_reduce_flat_kernel(const float* __restrict__ ws, float* __restrict__ dw,
                                                                int splitM, size_t slab, int accumulate, size_t stride,
                                                                float* __restrict__ extra, int extra_n, int extra_pad,
                                                                int extra_acc) {
  __shared__ float4 red[16][16];
  const int col = threadIdx.x & 15, row = threadIdx.x >> 4;
  const size_t m4 = slab >> 2;
  const size_t n4 = m4 + (extra != nullptr ? (size_t)(extra_pad >> 2) : 0);
  for (size_t base = (size_t)blockIdx.x * 16; base < n4; base += (size_t)gridDim.x * 16) {
    const size_t i = base + col;
    float4 t = {0.f, 0.f, 0.f, 0.f};
    if (i < n4) {
      // up to eight slabs per thread in flight (round 5: a 512-way split was eight dependent rounds of four loads) — but only
      // as many as the thread has: the loads are unconditional.  Additions in slab order whatever the unroll.
      if (splitM > 64) reduce_slabs<8>(ws, stride, i, row, splitM, t);
      else if (splitM > 32) reduce_slabs<4>(ws, stride, i, row, splitM, t);
      else if (splitM > 16) reduce_slabs<2>(ws, stride, i, row, splitM, t);
      else if (row < splitM) {
        const float4 v = reinterpret_cast<const float4*>(ws + (size_t)row * stride)[i];
        t.x += v.x; t.y += v.y; t.z += v.z; t.w += v.w;
      }
    }
    red[row][col] = t;
    __syncthreads();
    for (int s_ = 8; s_ > 0; s_ >>= 1) {
      if (row < s_) {
        const float4 o = red[row + s_][col];
        float4 m = red[row][col];
        m.x += o.x; m.y += o.y; m.z += o.z; m.w += o.w;
        red[row][col] = m;
      }
      __syncthreads();
    }
    if (row == 0 && i < m4) {
      float4 r = red[0][col];
      if (accumulate) {
        const float4 o = reinterpret_cast<const float4*>(dw)[i];
        r.x += o.x; r.y += o.y; r.z += o.z; r.w += o.w;
      }
      reinterpret_cast<float4*>(dw)[i] = r;
    } else if (row == 0 && i < n4) {
      const float4 r = red[0][col];
      const float rv[4] = {r.x, r.y, r.z, r.w};
      const int e0 = (int)(i - m4) * 4;
#pragma unroll
      for (int e = 0; e < 4; ++e)
        if (e0 + e < extra_n) extra[e0 + e] = extra_acc ? extra[e0 + e] + rv[e] : rv[e];
    }
    __syncthreads();
  }
}

struct Plan {
  int TN, TK, tilesN, tilesK, splitM, mchunk, MS;
  bool ring;
  bool taps;     // tap-stationary 3x3 kernel: tilesK counts 64-wide INPUT-CHANNEL tiles
};

static int taps_enabled() {   // TOK_WGRAD_TAPS=0: 3x3 layers stay on the two-buffer kernel (A/B switch)
  static const int v = [] { const char* e = getenv("TOK_WGRAD_TAPS"); return (int)(e ? atoi(e) : 1); }();
  return v;
}
static int taps_target() {    // TOK_WGRAD_TAPS_WGS=<n>: workgroups the split aims at (default 256; 512 is faster in isolation, 256 on the step)
  static const int v = [] { const char* e = getenv("TOK_WGRAD_TAPS_WGS"); return (int)(e ? atoi(e) : 256); }();
  return v;
}

static int ring_enabled() {   // TOK_WGRAD_RING=0: the two-buffer kernels of round 1 (A/B switch)
  static const int v = [] { const char* e = getenv("TOK_WGRAD_RING"); return (int)(e ? atoi(e) : 1); }();   // 2: every layer
  return v;
}
static int ring_target() {    // TOK_WGRAD_WGS=<n>: workgroups the split aims at (default: what is resident at once)
  static const int v = [] { const char* e = getenv("TOK_WGRAD_WGS"); return (int)(e ? atoi(e) : 0); }();
  return v;
}

Plan make_plan(const tok_conv_desc* d) {
  Plan p;
  const int Ktot = d->r * d->s_pad * d->c;
  const long long M = (long long)d->n * d->p * d->q;
  // the ring pays on the streaming (pointwise) layers; 3x3 / strided layers are LDS-read bound and keep the 64-row
  // two-buffer kernel (measured per layer, tools/bench_conv.py: ring 3x3 0.9-2x slower)
  p.taps = false;
  // (64-wide channel tiles; widths that are multiples of 48 but not of 64 — HRNet's 48 / 96 — take the 48-wide form of the
  //  window kernel on stride-1 layers and stay on the two-buffer kernel otherwise)
  const bool same3 = d->r == 3 && d->s == 3 && d->s_pad == 3 && d->stride == 1 && d->pad == 1 && taps_enabled() != 2 &&
                     (unsigned long long)d->n * d->h * d->w * d->c * 2 < 0x40000000ull &&
                     (unsigned long long)d->n * d->p * d->q * d->k * 2 < 0x40000000ull;
  const bool w64 = d->c % 64 == 0 && d->k % 64 == 0;
  const bool w48 = !w64 && d->c % 48 == 0 && d->k % 48 == 0 && same3 && taps_enabled() != 3;   // window kernel only
  if (d->c != 4 && (w64 || w48) && d->r == 3 && d->s == 3 && d->s_pad == 3 && taps_enabled()) {
    p.taps = true; p.ring = false;
    p.TN = w64 ? 64 : 48; p.TK = p.TN; p.MS = 32;
    p.tilesN = tok_cdiv(d->k, p.TN);
    p.tilesK = tok_cdiv(d->c, p.TN);
    const int tiles = p.tilesN * p.tilesK;
    long long split = (taps_target() + tiles - 1) / tiles;
    const long long max_split = (M + 8 * 32 - 1) / (8 * 32);        // at least 8 stages per workgroup
    if (split > max_split) split = max_split;
    if (split > 512) split = 512;
    if (split < 1) split = 1;
    long long chunk = (M + split - 1) / split;
    chunk = ((chunk + 31) / 32) * 32;
    p.mchunk = (int)chunk;
    p.splitM = (int)((M + chunk - 1) / chunk);
    return p;
  }
  p.ring = d->c != 4 && ring_enabled() && ((d->r == 1 && d->s == 1 && d->stride == 1 && d->pad == 0) || ring_enabled() == 2);
  if (p.ring) {
    // ring kernel: 32 reduction rows per stage, three stages.  Narrow layers take a 64 x 256 / 256 x 64 tile.
    p.MS = 32;
    if (d->k <= 64) { p.TN = 64; p.TK = Ktot >= 256 ? 256 : (Ktot >= 128 ? 128 : 64); }
    else if (Ktot <= 64) { p.TK = 64; p.TN = d->k >= 256 ? 256 : 128; }
    else { p.TN = 128; p.TK = 128; }
    // wide outputs AND deep filters over very many pixels: 256 x 256 tiles on eight waves (conv_wgrad_ring8_kernel: half the staged
    // bytes per MAC) where the ragged edge tiles waste at most a third of the work.  Measured per call (tools/ubench/
    // wgrad256_check.py, same dW to fp32 summation order): HRNet-W48's 720 x 720 neck convolution over 786 432 pixels 1829 ->
    // 1479 us; on the 12 544 ... 50 176-pixel layers of ResNet-50 / SwinV2-T the 128 x 128 plan wins (99 vs 152 us at
    // 384 x 1536: twice the partial-slab bytes, one workgroup per CU, a barrier per 32 MFMAs) — hence the pixel threshold.
    // TOK_WGRAD_256=0 keeps the 128 x 128 plan everywhere, =2 lowers the threshold to 8192 pixels (A/B, tests).
    static const int big = [] { const char* e = getenv("TOK_WGRAD_256"); return (int)(e ? atoi(e) : 1); }();
    if (big && d->r == 1 && d->s == 1 && d->stride == 1 && d->pad == 0 && d->k >= 256 && Ktot >= 256 && M >= (big >= 2 ? 8192 : 200000) &&
        (long long)tok_cdiv(d->k, 256) * tok_cdiv(Ktot, 256) * 65536 * 2 <= (long long)d->k * Ktot * 3) {
      p.TN = 256; p.TK = 256;
    }
    p.tilesN = tok_cdiv(d->k, p.TN);
    p.tilesK = tok_cdiv(Ktot, p.TK);
    const int tiles = p.tilesN * p.tilesK;
    const int stage = 32 * (p.TN + p.TK) * 2;
    const int per_cu = (160 * 1024) / (3 * stage);                 // LDS-resident workgroups per CU
    // (split target: what is resident at once was 768 workgroups on the 128 x 128 tile; 512 measured better on the step — the
    //  partial-sum slabs are a third smaller and the side stream leaves more of every CU to the main chain:
    //  ResNet-50 19.93 -> 19.38 ms/step together with the tap kernels' 256, tools/ubench/sweep_r02.sh)
    static const int long_target = [] { const char* e = getenv("TOK_WGRAD_WGS_LONG"); return (int)(e ? atoi(e) : 0); }();   // TOK_WGRAD_WGS_LONG=<n>: split target of the long-M layers (>= 100 k rows: the main-stream launches)
    const int target = (long_target > 0 && M >= 100000) ? long_target
                       : ring_target() > 0 ? ring_target() : 256 * (per_cu > 2 ? 2 : per_cu);
    long long split = (target + tiles - 1) / tiles;
    const long long max_split = (M + 8 * 32 - 1) / (8 * 32);        // at least 8 stages per workgroup
    if (split > max_split) split = max_split;
    if (split > 512) split = 512;
    if (split < 1) split = 1;
    long long chunk = (M + split - 1) / split;
    chunk = ((chunk + 31) / 32) * 32;
    p.mchunk = (int)chunk;
    p.splitM = (int)((M + chunk - 1) / chunk);
    return p;
  }
  p.TN = d->k >= 128 ? 128 : 64;
  p.TK = Ktot >= 128 ? 128 : 64;
  // (a 128 x 256 tile for deep filters was measured: no gain — the loop is LDS-write bound, not barrier bound)
  p.tilesN = tok_cdiv(d->k, p.TN);
  p.tilesK = tok_cdiv(Ktot, p.TK);
  const int tiles = p.tilesN * p.tilesK;
  static const int tb_target = [] { const char* e = getenv("TOK_WGRAD_2BUF_WGS"); return (int)(e ? atoi(e) : 1024); }();     // TOK_WGRAD_2BUF_WGS=<n>: workgroups the split of the two-buffer kernel aims at
  long long split = (tb_target + tiles - 1) / tiles;  // aim at ~4 workgroups per CU
  // reduction rows per barrier: 64 on the long-M layers (twice the MFMAs per barrier), 32 where M is
  // short and occupancy (4 workgroups per CU instead of 2) matters more
  static const int ms64_any = [] { const char* e = getenv("TOK_WGRAD_MS64_ANY"); return (int)(e ? atoi(e) : 0); }();     // TOK_WGRAD_MS64_ANY=1: 64 rows per barrier on every long-M tile shape (experiment)
  p.MS = (M >= 100000 && ((p.TN == 128 && p.TK == 128) || ms64_any)) ? 64 : 32;
  const long long max_split = (M + 8 * p.MS - 1) / (8 * p.MS);   // at least 8 steps per workgroup
  if (split > max_split) split = max_split;
  // (512 only for the stem — 2 tiles over 3.2 M pixels at B = 256: 512 workgroups left every CU with 2 and the launch
  //  latency-bound; every other layer keeps its partition, and with it its summation order)
  const long long cap = (d->c == 4 && tiles <= 2) ? 512 : 256;
  if (split > cap) split = cap;
  if (split < 1) split = 1;
  long long chunk = (M + split - 1) / split;
  chunk = ((chunk + p.MS - 1) / p.MS) * p.MS;
  p.mchunk = (int)chunk;
  p.splitM = (int)((M + chunk - 1) / chunk);
  return p;
}

template <int TN, int TK, bool PWK>
void launch_ring_pw(const WgradArgs& a, hipStream_t st) {
  constexpr int smem = 3 * 32 * (TN + TK) * 2;
  static const bool attr_set = [&] {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&conv_wgrad_ring_kernel<TN, TK, PWK>),
                              hipFuncAttributeMaxDynamicSharedMemorySize, smem);
    return true;
  }();   // once per process (thread-safe function-local static)
  (void)attr_set;
  hipLaunchKernelGGL((conv_wgrad_ring_kernel<TN, TK, PWK>), dim3(a.tilesN * a.tilesK * a.splitM), dim3(256), smem, st, a);
}

template <int TN, int TK>
void launch_ring(const WgradArgs& a, hipStream_t st) {
  if (a.R == 1 && a.S == 1 && a.stride == 1 && a.pad == 0) launch_ring_pw<TN, TK, true>(a, st);
  else launch_ring_pw<TN, TK, false>(a, st);
}

template <int TN, int TK, bool C4, int MS, bool DMA_T, bool PWK>
void launch_wgrad_pw(const WgradArgs& a, hipStream_t st) {
  constexpr int smem = 2 * MS * ((TN + 16) * 2 + (TK + 16) * 2);   // (the unpadded DMA layout needs less)
  static const bool attr_set = [&] {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&conv_wgrad_kernel<TN, TK, C4, MS, DMA_T, PWK>),
                              hipFuncAttributeMaxDynamicSharedMemorySize, smem);
    return true;
  }();   // once per process (thread-safe function-local static)
  (void)attr_set;
  hipLaunchKernelGGL((conv_wgrad_kernel<TN, TK, C4, MS, DMA_T, PWK>), dim3(a.tilesN * a.tilesK * a.splitM), dim3(256),
                     smem, st, a);
}

template <int TN, int TK, bool C4, int MS, bool DMA_T>
void launch_wgrad_dma(const WgradArgs& a, hipStream_t st) {
  if constexpr (!C4) {
    if (a.R == 1 && a.S == 1 && a.stride == 1 && a.pad == 0) { launch_wgrad_pw<TN, TK, C4, MS, DMA_T, true>(a, st); return; }
  }
  launch_wgrad_pw<TN, TK, C4, MS, DMA_T, false>(a, st);
}

template <int TN, int TK, bool C4, int MS>
void launch_wgrad_ms(const WgradArgs& a, hipStream_t st) {
  if constexpr (!C4) {
    static long long dma_rows = -1;      // TOK_WGRAD_DMA_ROWS=<n>: layers with fewer output pixels stage by DMA
    if (dma_rows < 0) { const char* e = getenv("TOK_WGRAD_DMA_ROWS"); dma_rows = e ? atoll(e) : (1ll << 40); }
    if (a.M < dma_rows) { launch_wgrad_dma<TN, TK, C4, MS, true>(a, st); return; }
  }
  launch_wgrad_dma<TN, TK, C4, MS, false>(a, st);
}

template <int TN, int TK, bool C4>
void launch_wgrad(const WgradArgs& a, hipStream_t st, int ms) {
  if (ms == 64) launch_wgrad_ms<TN, TK, C4, 64>(a, st);
  else launch_wgrad_ms<TN, TK, C4, 32>(a, st);
}

}  // namespace

extern "C" size_t tok_conv_wgrad_ws_bytes(const tok_conv_desc* d) {
  if (d == nullptr) return 0;
  const Plan p = make_plan(d);
  return (size_t)p.splitM * d->k * d->r * d->s_pad * d->c * sizeof(float);
}

extern "C" int tok_conv_wgrad_bias_ok(const tok_conv_desc* d) {
  if (d == nullptr || d->c == 4) return 0;
  return make_plan(d).ring ? 1 : 0;
}

extern "C" size_t tok_conv_wgrad_bias_ws_bytes(const tok_conv_desc* d) {
  if (d == nullptr) return 0;
  const Plan p = make_plan(d);
  return tok_conv_wgrad_ws_bytes(d) + (size_t)p.splitM * d->k * sizeof(float);
}

namespace {
int wgrad_impl(const tok_conv_desc* d, const void* x, const void* dy, float* dw, int k_real, int c_real, void* ws, size_t ws_bytes,
               int accumulate, float* dbias, int bias_accumulate, void* stream);
}

extern "C" int tok_conv_wgrad(const tok_conv_desc* d, const void* x, const void* dy, float* dw,
                              int k_real, int c_real, void* ws, size_t ws_bytes, int accumulate,
                              void* stream) {
  return wgrad_impl(d, x, dy, dw, k_real, c_real, ws, ws_bytes, accumulate, nullptr, 0, stream);
}

extern "C" int tok_conv_wgrad_bias(const tok_conv_desc* d, const void* x, const void* dy, float* dw, int k_real, int c_real,
                                   void* ws, size_t ws_bytes, int accumulate, float* dbias, int bias_accumulate, void* stream) {
  TOK_CHECK_ARG(dbias != nullptr, "tok_conv_wgrad_bias: dbias must not be null");
  TOK_CHECK_ARG(tok_conv_wgrad_bias_ok(d), "tok_conv_wgrad_bias: layer not served (ask tok_conv_wgrad_bias_ok)");
  return wgrad_impl(d, x, dy, dw, k_real, c_real, ws, ws_bytes, accumulate, dbias, bias_accumulate, stream);
}

namespace {
int wgrad_impl(const tok_conv_desc* d, const void* x, const void* dy, float* dw, int k_real, int c_real, void* ws, size_t ws_bytes,
               int accumulate, float* dbias, int bias_accumulate, void* stream) {
  TOK_CHECK_ARG(d && x && dy && dw && ws, "tok_conv_wgrad: null pointer");
  if (tok_dbg_skip(16)) return TOK_OK;
  TOK_CHECK_ARG(d->k % 8 == 0 && (d->c % 8 == 0 || d->c == 4), "tok_conv_wgrad: bad channel padding");
  TOK_CHECK_ARG(k_real <= d->k && c_real <= d->c, "tok_conv_wgrad: real dims exceed padded dims");
  TOK_CHECK_ARG(d->c == 4 ? d->s_pad == 8 : d->s_pad == d->s, "tok_conv_wgrad: bad s_pad");
  const Plan p = make_plan(d);
  const size_t need = dbias ? tok_conv_wgrad_bias_ws_bytes(d) : tok_conv_wgrad_ws_bytes(d);
  if (ws_bytes < need) {
    tok_set_error("tok_conv_wgrad: workspace %zu < %zu bytes", ws_bytes, need);
    return TOK_ERR_WORKSPACE;
  }
  const bool flat = (k_real == d->k) && (c_real == d->c) && (d->s_pad == d->s) &&
                    (((size_t)d->k * d->r * d->s * d->c) % 4 == 0) && (((uintptr_t)dw & 15) == 0);
  const bool direct = flat && p.splitM == 1 && !accumulate;   // single chunk: write dW in place
  WgradArgs a;
  a.x = (const bf16*)x; a.dy = (const bf16*)dy; a.ws = direct ? dw : (float*)ws;
  a.H = d->h; a.W = d->w; a.C = d->c; a.K = d->k; a.R = d->r; a.S = d->s_pad; a.P = d->p; a.Q = d->q;
  a.stride = d->stride; a.pad = d->pad;
  a.M = d->n * d->p * d->q; a.PQ = d->p * d->q; a.HW = d->h * d->w;
  make_magic((uint32_t)a.PQ, a.pq_mul, a.pq_shift);
  make_magic((uint32_t)d->q, a.q_mul, a.q_shift);
  a.adv_q = 32 % d->q; a.adv_p = (32 / d->q) % d->p;
  a.Ktot = d->r * d->s_pad * d->c;
  a.tilesN = p.tilesN; a.tilesK = p.tilesK; a.splitM = p.splitM; a.mchunk = p.mchunk;
  // bias partials: with the flat reduce each split's row sits behind its dW slab (padded to 4 floats) and the one reduce
  // launch folds both; otherwise (direct write / strided masters) a separate block behind all slabs + tok_colsum_f32
  const size_t slab_f = (size_t)d->k * d->r * d->s_pad * d->c;
  const int cs_pad = (k_real + 3) / 4 * 4;
  const bool cs_inline = dbias != nullptr && flat && !direct;
  a.ws_stride = cs_inline ? slab_f + cs_pad : slab_f;
  a.cs = dbias ? (cs_inline ? (float*)ws + slab_f : (float*)((char*)ws + tok_conv_wgrad_ws_bytes(d))) : nullptr;
  a.cs_stride = cs_inline ? a.ws_stride : (size_t)k_real;
  a.cs_cols = k_real;
  {
    const unsigned long long xb = (unsigned long long)d->n * d->h * d->w * d->c * 2;
    const unsigned long long yb = (unsigned long long)a.M * d->k * 2;
    TOK_CHECK_ARG(xb < 0xFFFFFFF0ull && yb < 0xFFFFFFF0ull, "tok_conv_wgrad: tensors of 4 GiB or more are not supported");
    a.x_bytes = (uint32_t)xb; a.dy_bytes = (uint32_t)yb;
  }
  hipStream_t st = tok_stream(stream);
  const bool c4 = d->c == 4;
  if (c4 && dbias == nullptr && stem_wgrad_serves(d)) {
    // the 7x7 / stride 2 stem: transpose reads straight from a shared input window (stem.hip); one slab per workgroup
    stem_wgrad_launch(d, x, dy, a.ws, p.splitM, st);
  } else
  if (p.taps) {
    constexpr int smem = 3 * 10 * 32 * 128;
    static const bool attr_set = [&] {
      (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&conv_wgrad_taps_kernel<0>),
                                hipFuncAttributeMaxDynamicSharedMemorySize, smem);
      return true;
    }();   // once per process (thread-safe function-local static)
    (void)attr_set;
    // stride 1 / "same" padding: the shared-window kernel (its idle DMA lanes sit at 2^31 + 2^30: tensors below 1 GiB — make_plan
    // sends 48-wide layers elsewhere above that, 64-wide ones take the per-tap kernel below)
    const bool same = a.stride == 1 && a.pad == 1 && a.P == a.H && a.Q == a.W && a.x_bytes < 0x40000000u && a.dy_bytes < 0x40000000u &&
                      taps_enabled() != 2;
    if (same) {
      // (probe, round 6: asking for 160 KB of LDS — no LDS-using workgroup of another kernel beside this one — costs HRNet-W48 +1.2 ms and
      //  ResNet-50 +0.13 ms per step: the co-residency of the main stream's kernels is worth more than an undisturbed CU;
      //  profiles/r06_wgrad_winp_ab.txt)
      constexpr int smem_p = 3 * (32 * 128 + 128 * 128);
      static const bool attr_p = [&] {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&conv_wgrad_winp_kernel<4, 3, 9>), hipFuncAttributeMaxDynamicSharedMemorySize, smem_p);
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&conv_wgrad_winp_kernel<3, 3, 7>), hipFuncAttributeMaxDynamicSharedMemorySize, smem_p);
        return true;
      }();   // once per process (thread-safe function-local static)
      (void)attr_p;
      const dim3 gw(a.tilesN * a.tilesK * a.splitM);
      if (p.TN == 48) hipLaunchKernelGGL((conv_wgrad_winp_kernel<3, 3, 7>), gw, dim3(256), smem_p, st, a);
      else hipLaunchKernelGGL((conv_wgrad_winp_kernel<4, 3, 9>), gw, dim3(256), smem_p, st, a);
    } else {
      hipLaunchKernelGGL((conv_wgrad_taps_kernel<0>), dim3(a.tilesN * a.tilesK * a.splitM), dim3(256), smem, st, a);
    }
  } else if (p.ring) {
    if (p.TN == 256 && p.TK == 256) {
      constexpr int smem8 = 3 * 32 * (256 + 256) * 2;
      static const bool attr8 = [] {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&conv_wgrad_ring8_kernel<256, 256>),
                                  hipFuncAttributeMaxDynamicSharedMemorySize, smem8);
        return true;
      }();
      (void)attr8;
      hipLaunchKernelGGL((conv_wgrad_ring8_kernel<256, 256>), dim3(a.tilesN * a.tilesK * a.splitM), dim3(512), smem8, st, a);
    }
    else if (p.TN == 64 && p.TK == 64) launch_ring<64, 64>(a, st);
    else if (p.TN == 64 && p.TK == 128) launch_ring<64, 128>(a, st);
    else if (p.TN == 64 && p.TK == 256) launch_ring<64, 256>(a, st);
    else if (p.TN == 128 && p.TK == 64) launch_ring<128, 64>(a, st);
    else if (p.TN == 256 && p.TK == 64) launch_ring<256, 64>(a, st);
    else launch_ring<128, 128>(a, st);
  } else if (p.TN == 128 && p.TK == 128) {
    if (c4) launch_wgrad<128, 128, true>(a, st, p.MS); else launch_wgrad<128, 128, false>(a, st, p.MS);
  } else if (p.TN == 128) {
    if (c4) launch_wgrad<128, 64, true>(a, st, p.MS); else launch_wgrad<128, 64, false>(a, st, p.MS);
  } else if (p.TK == 128) {
    if (c4) launch_wgrad<64, 128, true>(a, st, p.MS); else launch_wgrad<64, 128, false>(a, st, p.MS);
  } else {
    if (c4) launch_wgrad<64, 64, true>(a, st, p.MS); else launch_wgrad<64, 64, false>(a, st, p.MS);
  }
  TOK_CHECK_LAUNCH("tok_conv_wgrad");
  if (dbias != nullptr && !cs_inline) {
    // fold the per-split column sums (fixed order) into the bias gradient; padded channels (>= k_real) are not written
    if (int e = tok_colsum_f32(a.cs, p.splitM, k_real, dbias, bias_accumulate, stream)) return e;
  }
  if (direct) return TOK_OK;
  if (flat) {
    if (tok_dbg_skip(2)) return TOK_OK;
    const size_t slab = (size_t)d->k * d->r * d->s * d->c;
    const size_t nb = (slab / 4 + 15) / 16;
    hipLaunchKernelGGL(wgrad_reduce_flat_kernel, dim3((int)(nb < 4096 ? nb : 4096)), dim3(256), 0, st,
                       (const float*)ws, dw, p.splitM, slab, accumulate, cs_inline ? a.ws_stride : slab,
                       cs_inline ? dbias : (float*)nullptr, k_real, cs_inline ? cs_pad : 0, bias_accumulate);
    TOK_CHECK_LAUNCH("tok_conv_wgrad(reduce)");
    return TOK_OK;
  }
  const size_t total = (size_t)k_real * d->r * d->s * c_real;
  const int blocks = (int)((total + 255) / 256 < 2048 ? (total + 255) / 256 : 2048);
  hipLaunchKernelGGL(wgrad_reduce_kernel, dim3(blocks), dim3(256), 0, st, (const float*)ws, dw,
                     p.splitM, k_real, d->r, d->s, c_real, d->k, d->s_pad, d->c, accumulate);
  TOK_CHECK_LAUNCH("tok_conv_wgrad(reduce)");
  return TOK_OK;
}
}  // namespace
