// 3x3 / stride 1 / padding 1 convolution (forward, and data gradient = the same kernel on the tap-flipped transposed pack)
// on a SHARED INPUT WINDOW in LDS (gfx950).
//
// What bounds the implicit-GEMM kernels on these layers is not the MFMA pipe and not HBM but the path from L2 into the CU:
// the vector-memory front end delivers ~40 B/clk/CU (tools/ubench/dma_rate.hip: 36-43), and a 128 x 128 x 64 K step stages
// 32 KB for 512 MFMA cycles = 64 B/clk at the matrix peak (256 x 128: 48).  conv_igemm.hip (and round 3's ring kernel) stage the
// activation tile once PER TAP — nine copies of (almost) the same pixels (probe: with one of four activation DMA rows real,
// the ring kernel ran 1.2-1.3x faster; profiles/r03_ring_probe.txt).  Here a workgroup owns a 2-D patch of TH x TW = 256 output
// pixels x 128 output channels and keeps, per 32-channel chunk, the (TH + 2) x (TW + 2) input window in LDS ONCE; the nine
// taps read their MFMA B fragments from it at shifted pixel offsets.  Staged bytes per 32-channel chunk: window ~25 KB +
// weights 9 x 8 KB = 97 KB instead of 9 x 24 KB = 216 KB (21 B/clk at the matrix peak), DMA instructions per thread and
// stage 3 instead of 6.
//
//   * tile = TH consecutive rows of the FLATTENED (image, row) axis x TW columns (TW in {16, 32, 64}, TH = 256 / TW): maps
//     narrower than TW waste the tail columns (14 of 16, 28 of 32, 56 of 64 = 87.5 % on ResNet's maps), wide maps (HRNet) are
//     walked in x-tiles whose halo columns are real neighbours.  A tile may span several images: a tap row outside the image
//     (top tap of row 0, bottom tap of row H-1) is SKIPPED per 16-pixel fragment (all pixels of a fragment share a row: the
//     branch is wave-uniform), the window itself holds the neighbouring image's rows for the fragments that want them.
//   * stage = (32-channel chunk, tap): one 128 x 32 weight tile (8 KB) on a three-slot ring + one 1-KB piece of the NEXT
//     chunk's window (two window buffers).  A stage issues two weight DMA instructions per thread plus, at taps 0..6, one
//     window piece (out-of-range offsets return zeros without traffic): the counted `s_waitcnt vmcnt(2 | 3)` + raw s_barrier
//     per unrolled tap is exact; two stages stay in flight.
//     The nine taps are unrolled: ring slot, tap offsets and the window-piece schedule are compile-time constants and a
//     fragment address is ONE lane-constant base register plus an immediate.
//   * the window image is lane-linear (pixel p at p * 64 B, no swizzle): a ds_read_b128 of 16 consecutive pixels is 2-way
//     bank-conflicted (8 LDS cycles instead of 4; 640 of the 1024 MFMA cycles of a stage pair per CU) — cheaper than the
//     nine VALU instructions per read a swizzle that survives arbitrary pixel shifts would cost.  Weight rows keep
//     a conflict-free XOR swizzle (applied to the DMA source chunk).
//   * 4 waves = 2 (pixel rows) x 2 (64 channels): a wave owns 128 pixels x 64 channels (128 accumulator registers), weights
//     are the MFMA A operand (16-byte NHWC stores), persistent XCD-aware tile walk, BatchNorm partial sums folded per tile
//     into two registers, epilogue options as conv_igemm.hip.  80 KB of LDS: two workgroups per CU.
//     (s_setprio(1) around the MFMA clusters: measured neutral to -3 % — the two waves of a SIMD belong to different
//     workgroups at unrelated phases; not kept.)
#include "conv_common.h"
#include <stdlib.h>
#include <type_traits>

// TOK_WIN_PROBE=<mask> (probe builds only, tools/ubench/win_probe.sh; results invalid, timing only):
//   1 no window-fragment LDS reads after the first tap of a chunk   2 no weight-fragment LDS reads after the first tap
//   4 no block barrier   8 no DMA (weights + window)   16 no MFMAs   32 no epilogue
#ifndef TOK_WIN_PROBE
#define TOK_WIN_PROBE 0
#endif

namespace {

constexpr int WBK = 32;
constexpr int WIN_PIECES = 28;                    // 1-KB DMA pieces per window buffer (4 waves x 7 slots)
constexpr int WIN_BUF = WIN_PIECES * 1024;        // 28 KB >= (TH + 2)(TW + 2) pixels x 64 B for every TW
constexpr int win_wrows(int bn) { return bn <= 64 ? 64 : 128; }             // weight rows of a ring slot (96-channel tiles keep 128-row slots)
constexpr int win_smem(int bn) { return 3 * win_wrows(bn) * WBK * 2 + 2 * WIN_BUF; }   // 80 KB (128 / 96 channels) / 68 KB (64)

__device__ __forceinline__ int win_f(int q) { return (0x78 >> ((q & 3) << 1)) & 3; }   // {0, 2, 3, 1}

template <int OFF>
__device__ __forceinline__ u32x4 wlds16(uint32_t addr) {
  u32x4 v;
  asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(v) : "v"(addr), "n"(OFF));
  return v;
}

struct WinGeo {
  int BH;            // B * H: rows of the flattened (image, row) axis
  int XT;            // x-tiles per row group
  FastDiv fd_xt, fd_h;
};

// BN = 128: 2 (pixel rows) x 2 (64 channels) waves of 128 pixels x 64 channels; BN = 64 (layers of <= 64 output channels: the
// high-resolution HRNet branches, ResNet's first stage): 4 x 1 waves of 64 pixels x 64 channels; BN = 96 (layers of 96 / 192
// channels, HRNet-W48's second and third branch: 128-wide tiles would run a quarter of their MFMAs on padding): 4 x 1 waves of
// 64 pixels x 96 channels = three 32-channel halves per wave; BN = 48 (HRNet-W48's first branch): 4 x 1 waves of 64 pixels x
// 48 channels = one half + one QUARTER (a single 16-row MFMA tile whose LDS rows 32 + 8 g + j hold channels 32 + 4 g + j, so
// that a lane's four accumulator rows are four consecutive channels: 8-byte stores, half a mask byte).
template <int TW, int WBN>
__global__ __launch_bounds__(256, 2) void conv_win_kernel(ConvArgs a, WinGeo geo) {
  constexpr int W_STAGE = win_wrows(WBN) * WBK * 2;   // one tap's weight slot: 8 / 4 KB
  constexpr int WGN = WBN == 128 ? 2 : 1, WGM = 4 / WGN;
  constexpr int WCH = WBN / WGN;                  // channels of a wave: 64 / 96
  constexpr int NH = WCH / 32;                    // 32-channel halves of a wave (two 16-row MFMA tiles each)
  constexpr bool QT = WCH % 32 != 0;              // + a 16-channel quarter
  constexpr int NTL = WCH / 16;                   // MFMA tiles of a wave along the channels
  static_assert(!QT || (NH == 1 && WGN == 1), "the quarter tile is written for 48-channel waves");
  constexpr int WROWS = W_STAGE / 4096;           // weight DMA instructions per thread and stage
  constexpr int TH = 256 / TW;
  constexpr int WW = TW + 2;                      // window pitch in pixels
  constexpr int WIN_PX = (TH + 2) * WW;
  static_assert(WIN_PX * 64 <= WIN_BUF, "window does not fit its buffer");
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
  const int ntiles = it0 < a.gridM ? (a.gridM - it0 + sweep - 1) / sweep : 0;
  const int NC = (a.C + WBK - 1) / WBK;           // 32-channel chunks
  const int nchunks = ntiles * NC;
  const int C2 = a.C * 2;                         // pixel pitch of the gathered tensor in bytes

  // ---- window loader: byte offsets (pixel part) of this thread's seven window pieces, for the tile of the NEXT chunk -----
  int winoff[7];
  auto setup_window = [&](int it) {
    // tile it -> (row group, x-tile)
    const uint32_t rg = fdiv((uint32_t)it, geo.fd_xt);
    const int xt = it - (int)rg * geo.XT;
    const int gy0 = (int)rg * TH - 1, gx0 = xt * TW - 1;
#pragma unroll
    for (int j = 0; j < 7; ++j) {
      const int wp = 16 * (j * 4 + wave_u) + (lane >> 2);
      const int wr = wp / WW, wx = wp - wr * WW;
      const int gy = gy0 + wr, gx = gx0 + wx;
      const bool ok = it < a.gridM && wp < WIN_PX && (unsigned)gy < (unsigned)geo.BH && (unsigned)gx < (unsigned)a.W;
      winoff[j] = ok ? (int)(((uint32_t)gy * (uint32_t)a.W + (uint32_t)gx) * (uint32_t)C2) + kc * 16 : -1;
    }
  };
  // piece j of the window of chunk cc into buffer `wb`
  auto issue_window = [&](int j, int cc, int wb) {
    const bool cok = cc * WBK + kc * 8 < a.C;
    uint32_t off = (winoff[j] >= 0 && cok) ? (uint32_t)(winoff[j] + cc * (WBK * 2)) : 0xFFFFFFF0u;
    asm volatile("" : "+v"(off));
    __builtin_amdgcn_raw_ptr_buffer_load_lds(xsrd, (lds_void*)(smem + WIN0 + wb * WIN_BUF + (j * 4 + wave_u) * 1024), 16, off, 0, 0,
                                             0);
  };
  // weight tile of (chunk cc, tap) into ring slot `slot`
  // channel (of this tile) that goes to LDS row lrow: the row itself, but for the quarter's rows (see above)
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
  // window: pixel (wave's first row + row + r) * WW + seg * 16 + li + s, chunk sl; everything but the lane part is an immediate
  uint32_t afrag = lds_base + WIN0 + (uint32_t)((wm * WROWS_PER_WAVE * WW + li) * 64 + sl * 16);   // window buffer 0

#define TOK_WIN_PIXEL(gy, gx) ((size_t)(gy) * a.W + (gx))
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

  // rows of this wave whose top / bottom tap falls outside their image (bit = row of the wave), for the tile being computed
  uint32_t top_mask = 0, bot_mask = 0;
  int gyC = 0, gxC = 0;                           // first output row (flattened) / column of the tile being computed
  auto setup_compute = [&](int it) {
    const uint32_t rg = fdiv((uint32_t)it, geo.fd_xt);
    const int xt = it - (int)rg * geo.XT;
    gyC = (int)rg * TH;
    gxC = xt * TW;
    uint32_t tm = 0, bm = 0;
#pragma unroll
    for (int rr = 0; rr < WROWS_PER_WAVE; ++rr) {
      const int gy = gyC + wm * WROWS_PER_WAVE + rr;
      const uint32_t b = fdiv((uint32_t)gy, geo.fd_h);
      const int y = gy - (int)b * a.H;
      tm |= (y == 0 ? 1u : 0u) << rr;
      bm |= (y == a.H - 1 ? 1u : 0u) << rr;
    }
    top_mask = __builtin_amdgcn_readfirstlane(tm);
    bot_mask = __builtin_amdgcn_readfirstlane(bm);
  };

#if TOK_WIN_PROBE & 3
  u32x4 pwf[NTL] = {}, paf[MT] = {};
#endif
  // one stage: tap (R, S) of the current chunk out of weight slot SLOT
  auto compute = [&](auto tapc, auto slotc) {
    constexpr int TAP = decltype(tapc)::value, SLOT = decltype(slotc)::value;
    constexpr int R = TAP / 3, S = TAP % 3;
#if TOK_WIN_PROBE & 3
    u32x4 (&wf)[NTL] = pwf;
    u32x4 (&af)[MT] = paf;
#else
    u32x4 wf[NTL], af[MT];
#endif
    if (!(TOK_WIN_PROBE & 2) || TAP == 0) {
    wf[0] = wlds16<SLOT * W_STAGE + 0 * 64>(wfrag);
    wf[1] = wlds16<SLOT * W_STAGE + 4 * 64>(wfrag);
    wf[2] = wlds16<SLOT * W_STAGE + 32 * 64>(wfrag);
    if constexpr (NTL >= 4) wf[3] = wlds16<SLOT * W_STAGE + 36 * 64>(wfrag);
    if constexpr (NH == 3) {
      wf[4] = wlds16<SLOT * W_STAGE + 64 * 64>(wfrag);
      wf[5] = wlds16<SLOT * W_STAGE + 68 * 64>(wfrag);
    }
    }
#define TOK_AF(mt) af[mt] = wlds16<(((mt) / SEGS + R) * WW + ((mt) % SEGS) * 16 + S) * 64>(afrag)
    if (!(TOK_WIN_PROBE & 1) || TAP == 0) {
    TOK_AF(0); TOK_AF(1); TOK_AF(2); TOK_AF(3);
    if constexpr (MT == 8) { TOK_AF(4); TOK_AF(5); TOK_AF(6); TOK_AF(7); }
    }
#undef TOK_AF
    const uint32_t skip = R == 0 ? top_mask : (R == 2 ? bot_mask : 0u);
    asm volatile("s_waitcnt lgkmcnt(%0)" ::"n"(MT / 2) : "memory");      // weights + the first half of the pixel tiles
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int mt = 0; mt < MT / 2; ++mt) {
      if (R != 1 && ((skip >> (mt / SEGS)) & 1u)) continue;
      if (TOK_WIN_PROBE & 16) { if (mt == 0) acc[0][0][0] += __builtin_bit_cast(f32x4, af[0])[0] + __builtin_bit_cast(f32x4, wf[0])[0]; continue; }
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
      if (R != 1 && ((skip >> (mt / SEGS)) & 1u)) continue;
      if (TOK_WIN_PROBE & 16) { if (mt == MT / 2) acc[0][0][1] += __builtin_bit_cast(f32x4, af[MT - 1])[0] + __builtin_bit_cast(f32x4, wf[NTL - 1])[0]; continue; }
#pragma unroll
      for (int t = 0; t < NTL; ++t)
        acc[t][mt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, wf[t]), __builtin_bit_cast(bf16x8, af[mt]),
                                                             acc[t][mt], 0, 0, 0);
    }
  };

  // epilogue of the tile being computed: lane (sl, li) holds channels nb + {0..7}, nb + 32 + {0..7} of pixel
  // (gyC + wave row + mt / SEGS, gxC + (mt % SEGS) * 16 + li)
#define TOK_WIN_EPI_PART 2
#include "conv_win_epilogue.inc"
#undef TOK_WIN_EPI_PART

  // ---- prologue: window of the first chunk (buffer 0), weight stages of its taps 0 and 1 ------------------------------------
  setup_window(it0);
#pragma unroll
  for (int j = 0; j < 7; ++j) issue_window(j, 0, 0);
  issue_weights(0, 0, 0, ntiles > 0);
  issue_weights(1, 0, 1, ntiles > 0);
  zero_acc();
  if (ntiles > 0) setup_compute(it0);

  // ---- chunk loop: nine unrolled stages ------------------------------------------------------------------------------------
  int itC = it0, ccC = 0;                          // chunk being computed
  int itN = it0, ccN = 0;                          // the chunk after it (its window is loaded during this iteration)
  int wb = 0;                                      // window buffer of the chunk being computed
  for (int ch = 0; ch < nchunks; ++ch) {
    // next chunk
    ccN = ccC + 1;
    itN = itC;
    const bool new_tile = ccN == NC;
    if (new_tile) { ccN = 0; itN = itC + sweep; setup_window(itN); }
    const bool liveN = ch + 1 < nchunks;

#define TOK_STAGE(TAP)                                                                                                          \
    {                                                                                                                           \
      /* newer than this stage's data: what the previous stage issued (the weight rows + a window piece at taps 0..6) */          \
      asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)" ::"n"((TAP == 0 || TAP == 8) ? WROWS : WROWS + 1) : "memory");                  \
      if (!(TOK_WIN_PROBE & 4)) __builtin_amdgcn_s_barrier();                                                                   \
      /* weights of stage + 2: taps 2..8 of this chunk, taps 0, 1 of the next one */                                            \
      if (TAP + 2 < 9) issue_weights((TAP + 2) % 3, ccC, TAP + 2, !(TOK_WIN_PROBE & 8));                                        \
      else issue_weights((TAP + 2) % 3, ccN, TAP + 2 - 9, liveN && !(TOK_WIN_PROBE & 8));                                       \
      /* one piece of the next chunk's window into the other buffer (its last reader finished before this barrier) */          \
      if (TAP < 7) issue_window(TAP, (TOK_WIN_PROBE & 8) ? (1 << 20) : ccN, wb ^ 1);                                            \
      compute(std::integral_constant<int, TAP>{}, std::integral_constant<int, TAP % 3>{});                                      \
    }
    TOK_STAGE(0) TOK_STAGE(1) TOK_STAGE(2) TOK_STAGE(3) TOK_STAGE(4) TOK_STAGE(5) TOK_STAGE(6) TOK_STAGE(7) TOK_STAGE(8)
#undef TOK_STAGE

    if (new_tile) {
      if (!(TOK_WIN_PROBE & 32) || ch + 1 == nchunks) epilogue();
      zero_acc();
      if (liveN) setup_compute(itN);
    }
    itC = itN;
    ccC = ccN;
    wb ^= 1;
    afrag = wb ? afrag + WIN_BUF : afrag - WIN_BUF;
  }
  asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");

#define TOK_WIN_EPI_PART 3
#include "conv_win_epilogue.inc"
#undef TOK_WIN_EPI_PART
#undef TOK_WIN_PIXEL
}

int win_flag() {   // TOK_CONV_WIN=0: 3x3 layers stay on the implicit-GEMM kernels (A/B switch)
  static const int v = [] { const char* e = getenv("TOK_CONV_WIN"); return (int)(e ? atoi(e) : 1); }();
  return v;
}
int win_min_tiles() {
  static const int v = [] { const char* e = getenv("TOK_CONV_WIN_MIN_TILES"); return (int)(e ? atoi(e) : 128); }();
  return v;
}

int pick_tw(int W) { return W <= 16 ? 16 : (W <= 32 ? 32 : 64); }
int win_96() {      // TOK_CONV_WIN_96=0: layers of 48 / 96 / 192 channels stay on 64- / 128-wide tiles (A/B switch)
  static const int v = [] { const char* e = getenv("TOK_CONV_WIN_96"); return (int)(e ? atoi(e) : 1); }();
  return v;
}
// channel tile of a layer of K output channels whose map gives `ptiles` pixel tiles: the widest form that does not pad (48 / 96 /
// 192 -> 48 / 96), and 96 instead of 128 where both divide K but 128-wide tiles would leave CUs without a workgroup (HRNet-W48's
// 384-channel branch at 16 x 32: 48 pixel tiles x 3 = 144 workgroups on 256 CUs; x 4 = 192 shorter ones)
int pick_wbn(int K, long long ptiles) {
  if (!win_96()) return K <= 64 ? 64 : 128;
  if (K == 48) return 48;
  if (K <= 64) return 64;
  if (K % 96 == 0 && K % 128 != 0) return 96;
  if (K % 96 == 0 && ptiles * (K / 128) <= 256 && ptiles * (K / 96) <= 512) return 96;
  return 128;
}

template <int TW, int BN>
void launch_variant(const ConvArgs& a, const WinGeo& g, int grid, hipStream_t st) {
  static const bool attr_set = [&] {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&conv_win_kernel<TW, BN>), hipFuncAttributeMaxDynamicSharedMemorySize,
                              win_smem(BN));
    return true;
  }();   // once per process (thread-safe function-local static)
  (void)attr_set;
  hipLaunchKernelGGL((conv_win_kernel<TW, BN>), dim3(grid), dim3(256), win_smem(BN), st, a, g);
}

}  // namespace

// geometry of the 3x3 window kernel for an (H x W, C -> K) layer over B images; gathered tensor = (B, H, W, C)
bool conv_win_serves(const ConvArgs& a) {
  if (!win_flag()) return false;
  if (!(a.R == 3 && a.S == 3 && a.stride == 1 && a.pad == 1)) return false;
  if (a.H != a.P || a.W != a.Q) return false;
  if (a.C % 8 != 0 || a.K % 8 != 0 || a.K < 32 || a.C < 32) return false;
  if (a.K > 64 && a.K < 96) return false;          // 72 / 80 / 88 channels would leave 128-wide tiles a third empty
  if (a.W < 12) return false;                       // 7 x 7 maps would use 7 of 16 columns
  if (a.x_bytes >= 0x7FFFFFF0u) return false;       // window offsets are kept as non-negative ints
  if (a.y2 != nullptr || a.act_x != nullptr || a.ep_scale != nullptr || a.sub != nullptr || a.fin_mode != 0) return false;
  int gm, gn;
  conv_win_tiles(a, &gm, &gn);
  return (long long)gm * gn >= win_min_tiles();
}

int conv_win_grid(int gridM, int gridN) {
  const int unit = 8 * gridN;
  int G = 512;                                    // two workgroups per CU
  const long long need = (long long)gridM * gridN;
  if (need < G) G = (int)((need + unit - 1) / unit) * unit;
  G = G / unit * unit;
  if (G < unit) G = unit;
  return G;
}

// tile counts of a layer: gridM = row groups x x-tiles, gridN = channel tiles (128 wide; 64 for layers of <= 64 channels)
void conv_win_tiles(const ConvArgs& a, int* gridM, int* gridN) {
  const int tw = pick_tw(a.W), th = 256 / tw;
  const long long bh = (long long)(a.M / (a.H * a.W)) * a.H;
  *gridM = (int)(((bh + th - 1) / th) * ((a.W + tw - 1) / tw));
  const int bn = pick_wbn(a.K, *gridM);
  *gridN = (a.K + bn - 1) / bn;
}

int conv_win_launch(ConvArgs& a, hipStream_t st) {
  const int tw = pick_tw(a.W);
  WinGeo g;
  g.BH = (a.M / (a.H * a.W)) * a.H;
  g.XT = (a.W + tw - 1) / tw;
  g.fd_xt = make_fastdiv(g.XT);
  g.fd_h = make_fastdiv(a.H);
  conv_win_tiles(a, &a.gridM, &a.gridN);
  const int bn = pick_wbn(a.K, a.gridM);
  const int grid = conv_win_grid(a.gridM, a.gridN);
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
