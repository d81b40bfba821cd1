// Parameter gradients of the transformer MLP from its INPUT and its OUTPUT GRADIENT alone (gfx950): the 4C-wide hidden tensors
// (pre-activation, activation, and their gradients) are recomputed tile-wise in registers and never touch HBM.
//
// Follows timm 0.6.13 models/layers/mlp.py: Mlp (fc1 -> GELU -> fc2, drop = 0) as SwinV2 / DaViT call it
// (/root/reference/torchok/models/backbones/swin.py:18,238; davit.py:16,196), differentiated:
//
//   pre = x W1^T + b1,  act = GELU(pre),  y = act W2^T + b2
//   d(act) = dy W2,  d(pre) = d(act) * GELU'(pre)
//   dW1 = d(pre)^T x,  db1 = colsum d(pre),  dW2 = dy^T act,  db2 = colsum dy
//
// Unfused, the two weight-gradient GEMMs read act and d(pre) (2 x T x 4C bf16) that the forward and the data-gradient launch
// had to write first: per SwinV2-T block 4 of the 6 hidden-sized tensor passes, 30 of the step's 72 GB.  The reference carries
// the same trade as a switch (swin.py:75-78 `grad_checkpointing`: recompute in the backward instead of keeping activations).
//
// Work split: a WAVE owns HB x 16 hidden units for a whole range of tokens.  Its slices of W1 and W2^T are MFMA operands that
// stay in registers (HB x C/32 fragments each), and so do its two accumulators dW1[h][c] and dW2[c][h] (HB x C/16 16x16 blocks
// each) — the reduction index of both weight gradients is the token, which the wave walks 32 at a time:
//
//   phase 1   pre[t][h]  = sum_c x[t][c] W1[h][c]      A = token rows of x  (ds_read_b128 from the LDS tile), B = W1 slice
//             dact[t][h] = sum_c dy[t][c] W2[c][h]     A = token rows of dy,                                   B = W2^T slice
//             the D registers of the two 16-token blocks hold, per lane, hidden unit (lane & 15) of tokens 4g .. 4g+3 and
//             16+4g .. 16+4g+3: after bias / rounding / GELU / GELU' in place they ARE the operand of phase 2 whose k-slot
//             order is that of the LDS transpose read (conv_wgrad.hip's recipe) — nothing crosses lanes.
//   phase 2   dW1[h][c] += sum_t dpre[t][h] x[t][c]    A = dpre (registers), B = x  columns (ds_read_b64_tr_b16, same LDS tile)
//             dW2[c][h] += sum_t dy[t][c] act[t][h]    A = dy columns (ds_read_b64_tr_b16),  B = act (registers)
//
// A workgroup is four such waves (one per SIMD, the whole 512-register file each) sharing the x / dy tile: global -> LDS DMA,
// two buffers, rows padded to a power-of-two number of 16-byte chunks and XOR-swizzled with (row & 7) << 1, which makes both
// read patterns conflict-free (checked by enumeration against the bank rules of MI355X_MICROARCH.md).  Rounding points are those
// of the unfused launches (pre, act, d(act), d(pre) rounded to bf16), so the gradients differ from them only by summation order.
// Token ranges run as independent workgroups writing fp32 partial slabs; a second kernel folds them in a fixed order
// (deterministic, no atomics) into the gradient slots.
#include "tok_common.h"
#include <stdlib.h>
#include <type_traits>

#ifdef TOK_BUILD_EXPERIMENTS   // measured 1.4 ms slower on the SwinV2-T step: not in the default library

namespace {

typedef __attribute__((address_space(3))) void lds_void;
typedef __attribute__((address_space(3))) char lds_char;

struct DwArgs {
  const bf16* x;        // (T, C)
  const bf16* dy;       // (T, C)
  const bf16* w1;       // (H, C)  fc1 forward pack
  const float* b1;      // (H)
  const bf16* w2t;      // (H, C)  fc2 dgrad pack (W2^T)
  float* part;          // [NR][slab]: dW1 [H][C] | dW2 [C][H] | db1 [H] | db2 [C]
  uint32_t x_bytes, w_bytes;
  int T, H, SG, NR, rpr;       // slice groups (4 waves each), token ranges, rows per range (multiple of 32)
  size_t slab;
};

__device__ __forceinline__ u32x4 lds_read16(uint32_t addr) {
  u32x4 v;
  asm volatile("ds_read_b128 %0, %1" : "=v"(v) : "v"(addr));
  return v;
}
__device__ __forceinline__ u32x2 lds_read_tr(uint32_t addr) {
  u32x2 v;
  asm volatile("ds_read_b64_tr_b16 %0, %1" : "=v"(v) : "v"(addr));
  return v;
}
template <int N>
__device__ __forceinline__ void wait_lgkm() {
  asm volatile("s_waitcnt lgkmcnt(%0)" ::"n"(N) : "memory");
}

template <int C>
constexpr int row_chunks() { return C <= 128 ? 16 : (C <= 256 ? 32 : 64); }

template <int C, int HB>
__global__ __launch_bounds__(256, 1) void mlp_dw_kernel(DwArgs a) {
  constexpr int KS = C / 32, CB = C / 16, CH = C / 8;
  constexpr int P = row_chunks<C>(), S = P * 16, TT = 32;
  constexpr int TILE = TT * S, STAGE = 2 * TILE;
  constexpr int NBLK = STAGE / 1024, NI = NBLK / 4;
  static_assert(NBLK % 4 == 0, "DMA blocks of a stage divide among the four waves");

  extern __shared__ __attribute__((aligned(16))) char smem[];
  const uint32_t lds_base = (uint32_t)(size_t)(lds_char*)smem;

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int r = lane & 15, g = lane >> 4;

  const int id = tok_xcd_remap(blockIdx.x, gridDim.x);      // the slice groups of one token range share an XCD's L2
  const int range = id / a.SG, sg = id - range * a.SG;
  const int row0 = range * a.rpr;
  const int rend = min(a.T, row0 + a.rpr);
  const int ntiles = (rend - row0 + TT - 1) / TT;
  const int h0 = (sg * 4 + wave) * (HB * 16);

  const __amdgpu_buffer_rsrc_t xsrd = __builtin_amdgcn_make_buffer_rsrc((void*)a.x, 0, a.x_bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t ysrd = __builtin_amdgcn_make_buffer_rsrc((void*)a.dy, 0, a.x_bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t w1srd = __builtin_amdgcn_make_buffer_rsrc((void*)a.w1, 0, a.w_bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t w2srd = __builtin_amdgcn_make_buffer_rsrc((void*)a.w2t, 0, a.w_bytes, 0x00020000);

  // ---- stationary operands: lane (r, g) holds channels ks*32 + 8g .. +8 of hidden unit h0 + hb*16 + r ---------------------
  u32x4 w1f[HB][KS], w2f[HB][KS];
  float b1v[HB];
#pragma unroll
  for (int hb = 0; hb < HB; ++hb) {
    const uint32_t o = (uint32_t)(((h0 + hb * 16 + r) * C + 8 * g) * 2);
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) {
      w1f[hb][ks] = __builtin_amdgcn_raw_buffer_load_b128(w1srd, o + ks * 64, 0, 0);
      w2f[hb][ks] = __builtin_amdgcn_raw_buffer_load_b128(w2srd, o + ks * 64, 0, 0);
    }
    b1v[hb] = a.b1[h0 + hb * 16 + r];
  }

  // ---- tile DMA: block bid = j*4 + wave of a stage is 1 KB of the LDS image [x tile | dy tile], rows of P chunks -------------
  uint32_t voff[NI];
#pragma unroll
  for (int j = 0; j < NI; ++j) {
    const int bid = j * 4 + wave;
    const int pos = (bid % (NBLK / 2)) * 1024 + lane * 16;
    const int t = pos / S, ph = (pos % S) / 16;
    const int q = ph ^ ((t & 7) << 1);
    voff[j] = q < CH ? (uint32_t)((t * C + q * 8) * 2) : 0xFFFFFFF0u;
  }
  auto issue = [&](int tile, int buf, bool live) {
    const uint32_t tbase = (uint32_t)(row0 + tile * TT) * (uint32_t)(C * 2);
#pragma unroll
    for (int j = 0; j < NI; ++j) {
      const int bid = j * 4 + wave;
      uint32_t off = (live && voff[j] != 0xFFFFFFF0u) ? voff[j] + tbase : 0xFFFFFFF0u;
      asm volatile("" : "+v"(off));
      lds_void* dst = (lds_void*)(smem + buf * STAGE + bid * 1024);
      if (bid < NBLK / 2) __builtin_amdgcn_raw_ptr_buffer_load_lds(xsrd, dst, 16, off, 0, 0, 0);
      else __builtin_amdgcn_raw_ptr_buffer_load_lds(ysrd, dst, 16, off, 0, 0, 0);
    }
  };

  // ---- per-lane LDS read addressing --------------------------------------------------------------------------------------
  // token-row fragments (phase 1): row 16 tb + r, chunk (4 ks + g) ^ ((r & 7) << 1)
  const uint32_t a_row = (uint32_t)(r * S);
  const uint32_t a_sw = (uint32_t)((r & 7) << 1);
  // transpose reads (phase 2): row 16 half + 4g + (r >> 2), chunk (2 cb + ((r & 3) >> 1)) ^ ((row & 7) << 1), byte 8 (r & 1)
  const int trow = 4 * g + (r >> 2);
  const uint32_t t_row = (uint32_t)(trow * S + (r & 1) * 8);
  const uint32_t t_sw = (uint32_t)((trow & 7) << 1);
  const uint32_t t_lo = (uint32_t)((r & 3) >> 1);

  f32x4 dw1[HB][CB], dw2[HB][CB];
#pragma unroll
  for (int hb = 0; hb < HB; ++hb)
#pragma unroll
    for (int cb = 0; cb < CB; ++cb) {
      dw1[hb][cb] = (f32x4){0.f, 0.f, 0.f, 0.f};
      dw2[hb][cb] = (f32x4){0.f, 0.f, 0.f, 0.f};
    }
  float db1a[HB];
#pragma unroll
  for (int hb = 0; hb < HB; ++hb) db1a[hb] = 0.f;
  const bool do_db2 = (sg == 0 && wave == 0);       // one wave per token range sums dy's columns
  float db2a[CB];
#pragma unroll
  for (int cb = 0; cb < CB; ++cb) db2a[cb] = 0.f;

  issue(0, 0, ntiles > 0);

  for (int it = 0; it < ntiles; ++it) {
    const int buf = it & 1;
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");       // this wave's share of tile `it` has landed
    __builtin_amdgcn_s_barrier();                          // everyone's has; nobody reads the other buffer any more
    issue(it + 1, buf ^ 1, it + 1 < ntiles);
    const uint32_t xb = lds_base + (uint32_t)(buf * STAGE), yb = xb + TILE;

    // ---- phase 1: pre and d(act) of the wave's hidden units for 32 tokens -------------------------------------------------
    f32x4 pq[HB][2], dq[HB][2];
#pragma unroll
    for (int tb = 0; tb < 2; ++tb) {
#pragma unroll
      for (int hb = 0; hb < HB; ++hb) {
        pq[hb][tb] = (f32x4){0.f, 0.f, 0.f, 0.f};
        dq[hb][tb] = (f32x4){0.f, 0.f, 0.f, 0.f};
      }
      constexpr int KG = KS > 6 ? 6 : KS;               // fragment reads per wait (lgkmcnt is a 4-bit counter)
#pragma unroll
      for (int k0 = 0; k0 < KS; k0 += KG) {
        u32x4 xa[KG], ya[KG];
#pragma unroll
        for (int kk = 0; kk < KG; ++kk) {
          const uint32_t ch = (((uint32_t)(4 * (k0 + kk) + g)) ^ a_sw) << 4;
          xa[kk] = lds_read16(xb + a_row + (uint32_t)(tb * 16 * S) + ch);
          ya[kk] = lds_read16(yb + a_row + (uint32_t)(tb * 16 * S) + ch);
        }
        wait_lgkm<0>();
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int kk = 0; kk < KG; ++kk)
#pragma unroll
          for (int hb = 0; hb < HB; ++hb) {
            pq[hb][tb] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, xa[kk]),
                                                                __builtin_bit_cast(bf16x8, w1f[hb][k0 + kk]), pq[hb][tb], 0, 0, 0);
            dq[hb][tb] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, ya[kk]),
                                                                __builtin_bit_cast(bf16x8, w2f[hb][k0 + kk]), dq[hb][tb], 0, 0, 0);
          }
      }
    }
    // ---- elementwise, in place: the D registers become the phase-2 operands -----------------------------------------------
    bf16x8 actf[HB], dpf[HB];
#pragma unroll
    for (int hb = 0; hb < HB; ++hb) {
#pragma unroll
      for (int tb = 0; tb < 2; ++tb)
#pragma unroll
        for (int jp = 0; jp < 2; ++jp) {
          bf16x2 pr, dr;
          pr[0] = f2bf(pq[hb][tb][2 * jp] + b1v[hb]);
          pr[1] = f2bf(pq[hb][tb][2 * jp + 1] + b1v[hb]);
          dr[0] = f2bf(dq[hb][tb][2 * jp]);
          dr[1] = f2bf(dq[hb][tb][2 * jp + 1]);
          const f32x2 pv = {bf2f(pr[0]), bf2f(pr[1])};
          const f32x2 gv = gelu_f2(pv), gd = gelu_d2(pv);
          const bf16 a0 = f2bf(gv.x), a1 = f2bf(gv.y);
          const bf16 d0 = f2bf(bf2f(dr[0]) * gd.x), d1 = f2bf(bf2f(dr[1]) * gd.y);
          actf[hb][4 * tb + 2 * jp] = a0;
          actf[hb][4 * tb + 2 * jp + 1] = a1;
          dpf[hb][4 * tb + 2 * jp] = d0;
          dpf[hb][4 * tb + 2 * jp + 1] = d1;
          db1a[hb] += bf2f(d0) + bf2f(d1);
        }
    }
    // ---- phase 2: both weight gradients, 16 columns of x / dy at a time ----------------------------------------------------
    constexpr int CG = CB > 3 ? 3 : CB;                  // column blocks per wait: 4 transpose reads each
    static_assert(CB % CG == 0, "column blocks per group");
#pragma unroll
    for (int c0 = 0; c0 < CB; c0 += CG) {
      u32x2 xt[CG][2], yt[CG][2];
#pragma unroll
      for (int cc = 0; cc < CG; ++cc) {
        const uint32_t ch = (((uint32_t)(2 * (c0 + cc)) + t_lo) ^ t_sw) << 4;
        xt[cc][0] = lds_read_tr(xb + t_row + ch);
        xt[cc][1] = lds_read_tr(xb + t_row + (uint32_t)(16 * S) + ch);
        yt[cc][0] = lds_read_tr(yb + t_row + ch);
        yt[cc][1] = lds_read_tr(yb + t_row + (uint32_t)(16 * S) + ch);
      }
      wait_lgkm<0>();
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int cc = 0; cc < CG; ++cc) {
        const bf16x8 xf = __builtin_bit_cast(bf16x8, (u32x4){xt[cc][0][0], xt[cc][0][1], xt[cc][1][0], xt[cc][1][1]});
        const bf16x8 yf = __builtin_bit_cast(bf16x8, (u32x4){yt[cc][0][0], yt[cc][0][1], yt[cc][1][0], yt[cc][1][1]});
#pragma unroll
        for (int hb = 0; hb < HB; ++hb) {
          dw1[hb][c0 + cc] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(dpf[hb], xf, dw1[hb][c0 + cc], 0, 0, 0);
          dw2[hb][c0 + cc] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(yf, actf[hb], dw2[hb][c0 + cc], 0, 0, 0);
        }
        if (do_db2) {
          float s = 0.f;
#pragma unroll
          for (int e = 0; e < 8; ++e) s += bf2f(yf[e]);
          db2a[c0 + cc] += s;
        }
      }
    }
  }
  asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");

  // ---- partial slab of this token range ---------------------------------------------------------------------------------------
  float* slab = a.part + (size_t)range * a.slab;
  float* p1 = slab;                                  // dW1 [H][C]: rows h = 4g + i, column c = r
  float* p2 = slab + (size_t)a.H * C;                // dW2 [C][H]: rows c = 4g + i, column h = r
#pragma unroll
  for (int hb = 0; hb < HB; ++hb) {
    const int hbase = h0 + hb * 16;
#pragma unroll
    for (int cb = 0; cb < CB; ++cb)
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        p1[(size_t)(hbase + 4 * g + i) * C + cb * 16 + r] = dw1[hb][cb][i];
        p2[(size_t)(cb * 16 + 4 * g + i) * a.H + hbase + r] = dw2[hb][cb][i];
      }
    float s = db1a[hb];
    s += __shfl_xor(s, 16, 64);
    s += __shfl_xor(s, 32, 64);
    if (g == 0) slab[(size_t)2 * a.H * C + hbase + r] = s;
  }
  if (do_db2) {
#pragma unroll
    for (int cb = 0; cb < CB; ++cb) {
      float s = db2a[cb];
      s += __shfl_xor(s, 16, 64);
      s += __shfl_xor(s, 32, 64);
      if (g == 0) slab[(size_t)2 * a.H * C + a.H + cb * 16 + r] = s;
    }
  }
}

// out segment (+)= sum over the token ranges, in range order; four destinations behind one index space
struct FoldArgs {
  const float* part;
  size_t slab;
  int NR;
  long long n1, n2, n3, n4;      // element counts of the four segments (dW1, dW2, db1, db2)
  float* o1; float* o2; float* o3; float* o4;
  int a1, a2, a3, a4;            // accumulate into the destination (1) or overwrite (0)
};

__global__ __launch_bounds__(256) void mlp_dw_fold_kernel(FoldArgs f) {
  const long long total4 = (f.n1 + f.n2 + f.n3 + f.n4) / 4;
  for (long long i = blockIdx.x * 256ll + threadIdx.x; i < total4; i += (long long)gridDim.x * 256) {
    const long long e = i * 4;
    f32x4 s = *reinterpret_cast<const f32x4*>(f.part + e);
    for (int rr = 1; rr < f.NR; ++rr) s += *reinterpret_cast<const f32x4*>(f.part + (size_t)rr * f.slab + e);
    float* dst;
    int acc;
    long long off = e;
    if (off < f.n1) { dst = f.o1; acc = f.a1; }
    else if ((off -= f.n1) < f.n2) { dst = f.o2; acc = f.a2; }
    else if ((off -= f.n2) < f.n3) { dst = f.o3; acc = f.a3; }
    else { off -= f.n3; dst = f.o4; acc = f.a4; }
    if (dst == nullptr) continue;
    f32x4* d = reinterpret_cast<f32x4*>(dst + off);
    if (acc) s += *d;
    *d = s;
  }
}

struct DwPlan { int hb, sg, nr, rpr; size_t slab; };

DwPlan dw_plan(long long rows, int c, int hidden) {
  DwPlan p;
  p.hb = c == 96 ? 3 : (c == 192 ? 2 : 1);
  p.sg = hidden / (p.hb * 16 * 4);
  // one workgroup per CU and round; ranges of whole 32-token tiles, at least 16 tiles each
  static const int target = [] { const char* e = getenv("TOK_MLP_DW_WGS"); const int v = e ? atoi(e) : 256; return v < 1 ? 1 : v; }();
  long long nr = target / p.sg;
  if (nr < 1) nr = 1;
  const long long tiles = (rows + 31) / 32;
  if (nr > tiles / 16) nr = tiles / 16 > 0 ? tiles / 16 : 1;
  const long long tpr = (tiles + nr - 1) / nr;
  p.rpr = (int)(tpr * 32);
  p.nr = (int)((rows + p.rpr - 1) / p.rpr);
  p.slab = (size_t)2 * hidden * c + hidden + c;
  return p;
}

template <int C, int HB>
void launch_dw(const DwArgs& a, hipStream_t st) {
  constexpr int P = row_chunks<C>();
  constexpr int smem = 2 * 2 * 32 * P * 16;
  static const bool attr_set = [] {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&mlp_dw_kernel<C, HB>), hipFuncAttributeMaxDynamicSharedMemorySize, smem);
    return true;
  }();
  (void)attr_set;
  hipLaunchKernelGGL((mlp_dw_kernel<C, HB>), dim3(a.SG * a.NR), dim3(256), smem, st, a);
}

}  // namespace

extern "C" int tok_mlp_serves(int64_t rows, int c, int hidden);

extern "C" size_t tok_mlp_bwd_dw_ws_bytes(int64_t rows, int c, int hidden) {
  if (!tok_mlp_serves(rows, c, hidden)) return 0;
  const DwPlan p = dw_plan(rows, c, hidden);
  return p.slab * (size_t)p.nr * sizeof(float);
}

extern "C" int tok_mlp_bwd_dw(const void* x, const void* dy, const void* w1, const float* b1, const void* w2_dgrad, float* dw1,
                              int acc_w1, float* db1, int acc_b1, float* dw2, int acc_w2, float* db2, int acc_b2, void* ws,
                              size_t ws_bytes, int64_t rows, int c, int hidden, void* stream) {
  TOK_CHECK_ARG(x && dy && w1 && b1 && w2_dgrad && ws, "tok_mlp_bwd_dw: null pointer");
  TOK_CHECK_ARG(tok_mlp_serves(rows, c, hidden), "tok_mlp_bwd_dw: geometry (%lld, %d, %d) is not served (tok_mlp_serves)",
                (long long)rows, c, hidden);
  const DwPlan p = dw_plan(rows, c, hidden);
  TOK_CHECK_ARG(ws_bytes >= p.slab * (size_t)p.nr * sizeof(float), "tok_mlp_bwd_dw: workspace of %zu bytes, %zu needed", ws_bytes,
                p.slab * (size_t)p.nr * sizeof(float));
  DwArgs a;
  a.x = (const bf16*)x; a.dy = (const bf16*)dy; a.w1 = (const bf16*)w1; a.b1 = b1; a.w2t = (const bf16*)w2_dgrad;
  a.part = (float*)ws;
  a.x_bytes = (uint32_t)(rows * c * 2);
  a.w_bytes = (uint32_t)((long long)hidden * c * 2);
  a.T = (int)rows; a.H = hidden; a.SG = p.sg; a.NR = p.nr; a.rpr = p.rpr; a.slab = p.slab;
  hipStream_t st = tok_stream(stream);
  if (c == 96) launch_dw<96, 3>(a, st);
  else if (c == 192) launch_dw<192, 2>(a, st);
  else launch_dw<384, 1>(a, st);
  TOK_CHECK_LAUNCH("tok_mlp_bwd_dw");
  FoldArgs f;
  f.part = (const float*)ws; f.slab = p.slab; f.NR = p.nr;
  f.n1 = (long long)hidden * c; f.n2 = f.n1; f.n3 = hidden; f.n4 = c;
  f.o1 = dw1; f.o2 = dw2; f.o3 = db1; f.o4 = db2;
  f.a1 = acc_w1; f.a2 = acc_w2; f.a3 = acc_b1; f.a4 = acc_b2;
  const long long total4 = (f.n1 + f.n2 + f.n3 + f.n4) / 4;
  int blocks = (int)((total4 + 255) / 256);
  if (blocks > 2048) blocks = 2048;
  hipLaunchKernelGGL(mlp_dw_fold_kernel, dim3(blocks), dim3(256), 0, st, f);
  TOK_CHECK_LAUNCH("tok_mlp_bwd_dw(fold)");
  return TOK_OK;
}

#else   // default build: entry points stay (C ABI), the kernel is not compiled

extern "C" size_t tok_mlp_bwd_dw_ws_bytes(int64_t, int, int) { return 0; }   // 0 = "not served", as for an unsupported geometry
extern "C" int tok_mlp_bwd_dw(const void*, const void*, const void*, const float*, const void*, float*, int, float*, int, float*,
                              int, float*, int, void*, size_t, int64_t, int, int, void*) {
  tok_set_error("tok_mlp_bwd_dw: not built (compile with TOK_BUILD_EXPERIMENTS=1)");
  return TOK_ERR_INVALID;
}

#endif

extern "C" int tok_built_with_experiments(void) {
#ifdef TOK_BUILD_EXPERIMENTS
  return 1;
#else
  return 0;
#endif
}
