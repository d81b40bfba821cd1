// Data gradient of a 3x3 / stride 2 / padding 1 convolution on a SHARED dY WINDOW in LDS (gfx950).
//
// dX[y][x] = sum over the taps (R, S) of the flipped pack with (y + R - 1) and (x + S - 1) even of
// dY[(y + R - 1) / 2][(x + S - 1) / 2] . Wd[R][S]: the pixels of dX fall into four parity classes (py, px) = (y & 1, x & 1)
// with 1 / 2 / 2 / 4 live taps.  With H and W even every class is a P x Q map — the map of dY itself: class pixel (u, v) is
// dX pixel (2u + py, 2v + px) and its taps read dY at (u + dR, v + dS), dR = (R == 2), dS = (S == 2).  The implicit-GEMM
// kernel (conv_igemm.hip, IN_DIV = 2) gathers one activation tile per live tap; here a workgroup owns, as conv_win.hip does
// for stride 1, a patch of TH x TW = 256 class pixels x 128 channels and keeps per 32-channel chunk the (TH + 1) x (TW + 1)
// window of dY in LDS once; the live taps read their MFMA B fragments from it at the four (dR, dS) shifts.
//
//   * tile = (class, TH rows of the flattened (image, row) axis of dY, TW columns).  The tiles of a position come as four
//     consecutive tile numbers; the class a workgroup draws is rotated by its tile ordinal, so every workgroup walks through
//     heavy (four taps) and light (one tap) tiles.
//   * stage = (32-channel chunk, live tap): weight tile on a three-slot ring, issued two stages ahead by a second cursor that
//     walks the same (tile, chunk, tap) sequence; the NEXT chunk's window (six 1-KB pieces per thread, second buffer) is issued
//     whole at the first tap of a chunk, BEFORE that stage's weight rows, so that the counted waits are
//     vmcnt(weight rows) at the first tap of a chunk and vmcnt(weight rows + 6) at the second.
//   * tap, ring slot and class are run-time values (one kernel for the four classes): a fragment address is a lane-constant
//     base + a wave-uniform (dR, dS, slot) term + an immediate.
//   * bottom taps (dR = 1) of the last row of an image are skipped per 16-pixel fragment as in conv_win.hip; columns past the
//     map are zeros from the buffer unit.
//   * epilogue = conv_win.hip's with the class pixel mapping (dX row 2 gy + py on the flattened axis, column 2 gx + px).
#include "conv_common.h"
#include <stdlib.h>
#include <type_traits>


namespace {

constexpr int WBK = 32;
constexpr int NPJ = 6;                            // 1-KB window pieces per thread
constexpr int S2_BUF = NPJ * 4 * 1024;            // 24 KB >= (TH + 1)(TW + 1) pixels x 64 B for every TW
constexpr int s2_wrows(int bn) { return bn <= 64 ? 64 : 128; }              // weight rows of a ring slot (96-channel tiles keep 128-row slots)
constexpr int s2_smem(int bn) { return 3 * s2_wrows(bn) * WBK * 2 + 2 * S2_BUF; }   // 72 KB (128 / 96 channels) / 60 KB (64)

__device__ __forceinline__ int win_f(int q) { return (0x78 >> ((q & 3) << 1)) & 3; }   // {0, 2, 3, 1}

template <int OFF>
__device__ __forceinline__ u32x4 wlds16(uint32_t addr) {
  u32x4 v;
  asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(v) : "v"(addr), "n"(OFF));
  return v;
}

struct S2Geo {
  int BH;            // B * P: rows of the flattened (image, row) axis of dY = of every class map
  int XT;            // x-tiles per row group
  FastDiv fd_xt, fd_h;
};

// tile number `it`, drawn as the k-th tile of its workgroup -> class (py * 2 + px) and position tile
__device__ __forceinline__ int s2_cls(int it, int k) { return (it + k) & 3; }
__device__ __forceinline__ int s2_ntaps(int cls) { return (1 + (cls >> 1)) * (1 + (cls & 1)); }

template <int TW, int WBN>
__global__ __launch_bounds__(256, 2) void conv_s2d_kernel(ConvArgs a, S2Geo geo) {
  constexpr int W_STAGE = s2_wrows(WBN) * WBK * 2;   // one tap's weight slot: 8 / 4 KB
  constexpr int WGN = WBN == 128 ? 2 : 1, WGM = 4 / WGN;   // wave layouts as conv_win.hip: 2 x 2 (128), 4 x 1 (64, 96)
  constexpr int WCH = WBN / WGN;                  // channels of a wave: 64 / 96
  constexpr int NH = WCH / 32;                    // 32-channel halves of a wave
  constexpr bool QT = WCH % 32 != 0;              // + a 16-channel quarter (48-channel tiles, see conv_win.hip)
  constexpr int NTL = WCH / 16;                   // MFMA tiles of a wave along the channels
  static_assert(!QT || (NH == 1 && WGN == 1), "the quarter tile is written for 48-channel waves");
  constexpr int WROWS = W_STAGE / 4096;           // weight DMA instructions per thread and stage
  constexpr int TH = 256 / TW;
  constexpr int WW = TW + 1;                      // window pitch in pixels
  constexpr int WIN_PX = (TH + 1) * WW;
  static_assert(WIN_PX * 64 <= S2_BUF, "window does not fit its buffer");
  constexpr int MT = 256 / (WGM * 16);            // 16-pixel fragments per wave: 8 / 4
  constexpr int SEGS = TW / 16;                   // 16-pixel fragments per tile row
  constexpr int WROWS_PER_WAVE = TH / WGM;        // tile rows of a wave
  static_assert(WROWS_PER_WAVE * SEGS == MT && WROWS_PER_WAVE >= 1, "wave tiling");

  extern __shared__ __attribute__((aligned(16))) char smem[];
  typedef __attribute__((address_space(3))) char lds_char;
  typedef __attribute__((address_space(3))) void lds_void;
  const uint32_t lds_base = (uint32_t)(size_t)(lds_char*)smem;     // [3 weight stages][2 window buffers]
  constexpr int WIN0 = 3 * W_STAGE;

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wv = tid >> 6;
  const int wn = wv % WGN;
  const int wm = wv / WGN;
  const int kc = tid & 3;
  const int lrow = tid >> 2;                      // 0..63
  const int kcW = kc ^ win_f(lrow >> 3);          // logical chunk of this thread's weight rows
  const int wave_u = __builtin_amdgcn_readfirstlane(tid >> 6);

  const __amdgpu_buffer_rsrc_t xsrd = __builtin_amdgcn_make_buffer_rsrc((void*)a.x, 0, a.x_bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t wsrd = __builtin_amdgcn_make_buffer_rsrc((void*)a.w, 0, a.w_bytes, 0x00020000);

  const int xcd = blockIdx.x & 7, jx = blockIdx.x >> 3;
  const int bn_fixed = jx % a.gridN, jm = jx / a.gridN;
  const int S8 = (gridDim.x >> 3) / a.gridN;
  const int sweep = 8 * S8;
  const int n0 = bn_fixed * WBN;
  const int it0 = xcd * S8 + jm;
  const int NC = (a.C + WBK - 1) / WBK;           // 32-channel chunks
  const int C2 = a.C * 2;                         // pixel pitch of dY in bytes

  // ---- window loader: byte offsets (pixel part) of this thread's window pieces, for the tile of the NEXT chunk -----------
  int winoff[NPJ];
  auto setup_window = [&](int it) {
    const int pos = it >> 2;
    const uint32_t rg = fdiv((uint32_t)pos, geo.fd_xt);
    const int xt = pos - (int)rg * geo.XT;
    const int gy0 = (int)rg * TH, gx0 = xt * TW;
#pragma unroll
    for (int j = 0; j < NPJ; ++j) {
      const int wp = 16 * (j * 4 + wave_u) + (lane >> 2);
      const int wr = wp / WW, wx = wp - wr * WW;
      const int gy = gy0 + wr, gx = gx0 + wx;
      const bool ok = it < a.gridM && wp < WIN_PX && gy < geo.BH && gx < a.W;
      winoff[j] = ok ? (int)(((uint32_t)gy * (uint32_t)a.W + (uint32_t)gx) * (uint32_t)C2) + kc * 16 : -1;
    }
  };
  auto issue_window = [&](int j, int cc, int wb) {
    const bool cok = cc * WBK + kc * 8 < a.C;
    uint32_t off = (winoff[j] >= 0 && cok) ? (uint32_t)(winoff[j] + cc * (WBK * 2)) : 0xFFFFFFF0u;
    asm volatile("" : "+v"(off));
    __builtin_amdgcn_raw_ptr_buffer_load_lds(xsrd, (lds_void*)(smem + WIN0 + wb * S2_BUF + (j * 4 + wave_u) * 1024), 16, off, 0, 0,
                                             0);
  };
  // weight tile of (chunk cc, tap) into ring slot `slot`
  // channel (of this tile) that goes to LDS row lrow: the row itself, but for the quarter's rows 32 + 8 g + j <- 32 + 4 g + j
  const int wsrc0 = (QT && lrow >= 32) ? 32 + ((lrow - 32) >> 3) * 4 + (lrow & 3) : lrow;
  const bool wuse0 = !(QT && lrow >= 32 && (lrow & 4));
  const int wrow_off0 = (n0 + wsrc0) * a.Ktot * 2, wrow_off1 = (n0 + lrow + 64) * a.Ktot * 2;
  const bool wrow_ok0 = wuse0 && wsrc0 < WBN && n0 + wsrc0 < a.K, wrow_ok1 = lrow + 64 < WBN && n0 + lrow + 64 < a.K;
  auto issue_weights = [&](int slot, int cc, int tap, bool live) {
    const int kch = cc * WBK + kcW * 8;
    const bool kok = live && kch < a.C;
    const int koff = (tap * a.C + kch) * 2;
    char* Wdst = smem + slot * W_STAGE + wave_u * 1024;
    uint32_t o0 = (kok && wrow_ok0) ? (uint32_t)(wrow_off0 + koff) : 0xFFFFFFF0u;
    uint32_t o1 = (kok && wrow_ok1) ? (uint32_t)(wrow_off1 + koff) : 0xFFFFFFF0u;
    asm volatile("" : "+v"(o0));
    asm volatile("" : "+v"(o1));
    __builtin_amdgcn_raw_ptr_buffer_load_lds(wsrd, (lds_void*)Wdst, 16, o0, 0, 0, 0);
    if (WROWS == 2) __builtin_amdgcn_raw_ptr_buffer_load_lds(wsrd, (lds_void*)(Wdst + 4096), 16, o1, 0, 0, 0);
  };

  // ---- fragment addressing ---------------------------------------------------------------------------------------------
  const int sl = lane >> 4;
  const int li = lane & 15;
  const int wrow0 = wn * WCH + (li >> 2) * 8 + (li & 3);
  const uint32_t wfrag = lds_base + (uint32_t)(wrow0 * 64 + ((sl ^ win_f(li >> 2)) << 4));   // + slot * W_STAGE + row imm
  // window: pixel (wave's first row + row + dR) * WW + seg * 16 + li + dS, chunk sl
  uint32_t afrag = lds_base + WIN0 + (uint32_t)((wm * WROWS_PER_WAVE * WW + li) * 64 + sl * 16);   // window buffer 0

#define TOK_WIN_PIXEL(gy, gx) ((size_t)(2 * (gy) + pyC) * a.Q + 2 * (gx) + pxC)   // dX pixel of class pixel (gy, gx)
#define TOK_WIN_EPI_PART 1
#include "conv_win_epilogue.inc"
#undef TOK_WIN_EPI_PART
  f32x4 acc[NTL][MT];
  auto zero_acc = [&]() {
#pragma unroll
    for (int t = 0; t < NTL; ++t)
#pragma unroll
      for (int mt = 0; mt < MT; ++mt) acc[t][mt] = (f32x4){0.f, 0.f, 0.f, 0.f};
  };

  // rows of this wave whose bottom tap (dR = 1) falls outside their image (bit = row of the wave), tile being computed
  uint32_t bot_mask = 0;
  int gyC = 0, gxC = 0;                           // first row (flattened) / column of the class map of the tile being computed
  int pyC = 0, pxC = 0;
  auto setup_compute = [&](int it, int cls) {
    const int pos = it >> 2;
    const uint32_t rg = fdiv((uint32_t)pos, geo.fd_xt);
    const int xt = pos - (int)rg * geo.XT;
    gyC = (int)rg * TH;
    gxC = xt * TW;
    pyC = cls >> 1;
    pxC = cls & 1;
    uint32_t bm = 0;
#pragma unroll
    for (int rr = 0; rr < WROWS_PER_WAVE; ++rr) {
      const int gy = gyC + wm * WROWS_PER_WAVE + rr;
      const uint32_t b = fdiv((uint32_t)gy, geo.fd_h);
      const int y = gy - (int)b * a.H;
      bm |= (y == a.H - 1 ? 1u : 0u) << rr;
    }
    bot_mask = __builtin_amdgcn_readfirstlane(bm);
  };

  // one stage: the tap with window shift (dR, dS) of the current chunk out of weight slot `slot`
  auto compute = [&](int dR, int dS, int slot) {
    u32x4 wf[NTL], af[MT];
    const uint32_t wa = wfrag + (uint32_t)(slot * W_STAGE);
    const uint32_t aa = afrag + (uint32_t)((dR * WW + dS) * 64);
    wf[0] = wlds16<0 * 64>(wa);
    wf[1] = wlds16<4 * 64>(wa);
    wf[2] = wlds16<32 * 64>(wa);
    if constexpr (NTL >= 4) wf[3] = wlds16<36 * 64>(wa);
    if constexpr (NH == 3) {
      wf[4] = wlds16<64 * 64>(wa);
      wf[5] = wlds16<68 * 64>(wa);
    }
#define TOK_AF(mt) af[mt] = wlds16<(((mt) / SEGS) * WW + ((mt) % SEGS) * 16) * 64>(aa)
    TOK_AF(0); TOK_AF(1); TOK_AF(2); TOK_AF(3);
    if constexpr (MT == 8) { TOK_AF(4); TOK_AF(5); TOK_AF(6); TOK_AF(7); }
#undef TOK_AF
    const uint32_t skip = dR ? bot_mask : 0u;
    asm volatile("s_waitcnt lgkmcnt(%0)" ::"n"(MT / 2) : "memory");      // weights + the first half of the pixel tiles
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int mt = 0; mt < MT / 2; ++mt) {
      if ((skip >> (mt / SEGS)) & 1u) continue;
#pragma unroll
      for (int t = 0; t < NTL; ++t)
        acc[t][mt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, wf[t]), __builtin_bit_cast(bf16x8, af[mt]),
                                                             acc[t][mt], 0, 0, 0);
    }
    __builtin_amdgcn_sched_barrier(0);
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int mt = MT / 2; mt < MT; ++mt) {
      if ((skip >> (mt / SEGS)) & 1u) continue;
#pragma unroll
      for (int t = 0; t < NTL; ++t)
        acc[t][mt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, wf[t]), __builtin_bit_cast(bf16x8, af[mt]),
                                                             acc[t][mt], 0, 0, 0);
    }
  };

  // epilogue of the tile being computed: lane (sl, li) holds channels nb + {0..7}, nb + 32 + {0..7} of pixel
  // (gyC + wave row + mt / SEGS, gxC + (mt % SEGS) * 16 + li) of the class map
#define TOK_WIN_EPI_PART 2
#include "conv_win_epilogue.inc"
#undef TOK_WIN_EPI_PART

  // ---- cursors over the (tile, chunk, live tap) sequence of this workgroup ------------------------------------------------
  // C: the stage being computed; Wc: the stage whose weights are issued (two ahead)
  int itC = it0, kC = 0, ccC = 0, tC = 0, clsC = s2_cls(it0, 0), ntC = s2_ntaps(clsC);
  int itW = it0, kW = 0, ccW = 0, tW = 0, clsW = clsC, ntW = ntC;
  int slotC = 0, slotW = 0;
  auto issue_next_weights = [&]() {
    // live tap tW of class clsW: R = py ? 2 * (t >> px) : 1, S = px ? 2 * (t & 1) : 1
    const int py = clsW >> 1, px = clsW & 1;
    const int R = py ? 2 * (tW >> px) : 1, S = px ? 2 * (tW & 1) : 1;
    issue_weights(slotW, ccW, R * 3 + S, itW < a.gridM);
    slotW = slotW == 2 ? 0 : slotW + 1;
    if (++tW == ntW) {
      tW = 0;
      if (++ccW == NC) {
        ccW = 0;
        itW += sweep;
        ++kW;
        clsW = s2_cls(itW, kW);
        ntW = s2_ntaps(clsW);
      }
    }
  };

  // ---- prologue: window of the first chunk (buffer 0), weight stages 0 and 1 ------------------------------------------------
  setup_window(it0);
#pragma unroll
  for (int j = 0; j < NPJ; ++j) issue_window(j, 0, 0);
  issue_next_weights();
  issue_next_weights();
  zero_acc();
  if (it0 < a.gridM) setup_compute(it0, clsC);

  int wb = 0;                                      // window buffer of the chunk being computed
  while (itC < a.gridM) {
    // newer than this stage's data: the weight rows of the previous stage and, at the second tap of a chunk, the window
    // pieces the first tap issued (they went out BEFORE its weight rows)
    if (tC == 1) asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)" ::"n"(WROWS + NPJ) : "memory");
    else asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)" ::"n"(WROWS) : "memory");
    __builtin_amdgcn_s_barrier();
    if (tC == 0) {
      // the next chunk's window into the other buffer (its last reader finished before this barrier)
      int ccN = ccC + 1;
      if (ccN == NC) { ccN = 0; setup_window(itC + sweep); }
#pragma unroll
      for (int j = 0; j < NPJ; ++j) issue_window(j, ccN, wb ^ 1);
    }
    issue_next_weights();
    {
      const int py = clsC >> 1, px = clsC & 1;
      compute(py ? (tC >> px) : 0, px ? (tC & 1) : 0, slotC);
    }
    slotC = slotC == 2 ? 0 : slotC + 1;
    if (++tC == ntC) {
      tC = 0;
      wb ^= 1;
      afrag = wb ? afrag + S2_BUF : afrag - S2_BUF;
      if (++ccC == NC) {
        ccC = 0;
        epilogue();
        zero_acc();
        itC += sweep;
        ++kC;
        clsC = s2_cls(itC, kC);
        ntC = s2_ntaps(clsC);
        if (itC < a.gridM) setup_compute(itC, clsC);
      }
    }
  }
  asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");

#define TOK_WIN_EPI_PART 3
#include "conv_win_epilogue.inc"
#undef TOK_WIN_EPI_PART
#undef TOK_WIN_PIXEL
}

int s2d_flag() {   // TOK_CONV_S2D=0: stride-2 data gradients stay on the implicit-GEMM kernel (A/B switch)
  static const int v = [] { const char* e = getenv("TOK_CONV_S2D"); return (int)(e ? atoi(e) : 1); }();
  return v;
}
int s2d_min_tiles() {
  static const int v = [] { const char* e = getenv("TOK_CONV_S2D_MIN_TILES"); return (int)(e ? atoi(e) : 256); }();
  return v;
}

int pick_tw(int W) { return W <= 16 ? 16 : (W <= 32 ? 32 : 64); }
int s2d_96() {      // TOK_CONV_WIN_96=0: layers of 96 / 192 channels stay on 128-wide tiles (A/B switch shared with conv_win.hip)
  static const int v = [] { const char* e = getenv("TOK_CONV_WIN_96"); return (int)(e ? atoi(e) : 1); }();
  return v;
}
// channel tile of a layer of K output channels whose map gives `ptiles` pixel tiles: the widest form that does not pad (48 / 96 /
// 192 -> 48 / 96), and 96 instead of 128 where both divide K but 128-wide tiles would leave CUs without a workgroup (HRNet-W48's
// 384-channel branch at 16 x 32: 48 pixel tiles x 3 = 144 workgroups on 256 CUs; x 4 = 192 shorter ones)
int pick_wbn(int K, long long ptiles) {
  if (!s2d_96()) return K <= 64 ? 64 : 128;
  if (K == 48) return 48;
  if (K <= 64) return 64;
  if (K % 96 == 0 && K % 128 != 0) return 96;
  if (K % 96 == 0 && ptiles * (K / 128) <= 256 && ptiles * (K / 96) <= 512) return 96;
  return 128;
}

template <int TW, int BN>
void launch_variant(const ConvArgs& a, const S2Geo& g, int grid, hipStream_t st) {
  static const bool attr_set = [&] {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&conv_s2d_kernel<TW, BN>), hipFuncAttributeMaxDynamicSharedMemorySize,
                              s2_smem(BN));
    return true;
  }();   // once per process (thread-safe function-local static)
  (void)attr_set;
  hipLaunchKernelGGL((conv_s2d_kernel<TW, BN>), dim3(grid), dim3(256), s2_smem(BN), st, a, g);
}

}  // namespace

// `a` as dgrad_fill (conv_igemm.hip) leaves it: gathered tensor = dY (B, P, Q, K) as (a.H, a.W, a.C), output = dX (a.P, a.Q, a.K)
bool conv_s2d_serves(const ConvArgs& a, int stride, int pad) {
  if (!s2d_flag()) return false;
  if (!(a.R == 3 && a.S == 3 && stride == 2 && pad == 1)) return false;
  if ((a.P & 1) || (a.Q & 1) || a.P != 2 * a.H || a.Q != 2 * a.W) return false;
  if (a.C % 8 != 0 || a.K % 8 != 0 || a.K < 32 || a.C < 32) return false;
  if (a.K > 64 && a.K < 96) return false;
  if (a.W < 12) return false;
  if (a.x_bytes >= 0x7FFFFFF0u) return false;
  if (a.y2 != nullptr || a.act_x != nullptr || a.ep_scale != nullptr || a.sub != nullptr || a.fin_mode != 0) return false;
  int gm, gn;
  conv_s2d_tiles(a, &gm, &gn);
  return (long long)gm * gn >= s2d_min_tiles();
}

int conv_s2d_grid(int gridM, int gridN) {
  const int unit = 8 * gridN;
  int G = 512;                                    // two workgroups per CU
  const long long need = (long long)gridM * gridN;
  if (need < G) G = (int)((need + unit - 1) / unit) * unit;
  G = G / unit * unit;
  if (G < unit) G = unit;
  return G;
}

// gridM = 4 classes x row groups x x-tiles (class fastest), gridN = channel tiles
void conv_s2d_tiles(const ConvArgs& a, int* gridM, int* gridN) {
  const int tw = pick_tw(a.W), th = 256 / tw;
  const long long bh = (long long)(a.M / (a.P * a.Q)) * a.H;
  *gridM = (int)(4 * ((bh + th - 1) / th) * ((a.W + tw - 1) / tw));
  const int bn = pick_wbn(a.K, *gridM);
  *gridN = (a.K + bn - 1) / bn;
}

int conv_s2d_launch(ConvArgs& a, hipStream_t st) {
  const int tw = pick_tw(a.W);
  S2Geo g;
  g.BH = (a.M / (a.P * a.Q)) * a.H;
  g.XT = (a.W + tw - 1) / tw;
  g.fd_xt = make_fastdiv(g.XT);
  g.fd_h = make_fastdiv(a.H);
  conv_s2d_tiles(a, &a.gridM, &a.gridN);
  const int bn = pick_wbn(a.K, a.gridM);
  const int grid = conv_s2d_grid(a.gridM, a.gridN);
  a.stat_rows = grid / a.gridN;
  if (bn == 128) {
    if (tw == 16) launch_variant<16, 128>(a, g, grid, st);
    else if (tw == 32) launch_variant<32, 128>(a, g, grid, st);
    else launch_variant<64, 128>(a, g, grid, st);
  } else if (bn == 96) {
    if (tw == 16) launch_variant<16, 96>(a, g, grid, st);
    else if (tw == 32) launch_variant<32, 96>(a, g, grid, st);
    else launch_variant<64, 96>(a, g, grid, st);
  } else if (bn == 48) {
    if (tw == 16) launch_variant<16, 48>(a, g, grid, st);
    else if (tw == 32) launch_variant<32, 48>(a, g, grid, st);
    else launch_variant<64, 48>(a, g, grid, st);
  } else {
    if (tw == 16) launch_variant<16, 64>(a, g, grid, st);
    else if (tw == 32) launch_variant<32, 64>(a, g, grid, st);
    else launch_variant<64, 64>(a, g, grid, st);
  }
  return 0;
}
