// The ResNet stem — 7x7 / stride 2 / padding 3 convolution of a 3-channel image (NHWC, channels padded to 4) into <= 64 channels
// (reference: torchok/models/backbones/resnet.py:471-476 `conv1`, run by ResNet.forward at :541-543) — on a SHARED INPUT WINDOW.
//
// conv_igemm.hip's C4 path gathers every output pixel's 7 x 8 x 4 patch through registers: 260 us for the 256 x 224 x 224 batch
// against ~100 us for its bytes (103 MB in, 411 MB out) and its MFMA work.  Here a workgroup owns a 16 x 16 patch of output
// pixels of one image and stages the (2 * 16 + 5) x (2 * 16 + 8)-pixel input window ONCE (12 KB by LDS-DMA, two buffers: the
// next tile's window is in flight while this one is multiplied).  One filter ROW is one 32-deep MFMA K step: 8 taps x 4
// channels = 64 contiguous bytes of a window row.  The window starts at an EVEN input column (2 * ox0 - 4) so that every DMA
// piece (two pixels) is entirely inside or entirely outside the image and every fragment read is 16-byte aligned; the price is
// that the padding tap of the packed filter ([k][7][8][4], tap 7 zero) has to sit in FRONT: the weight fragments are loaded
// shifted by one tap (slot 0 = zero, slots 1..7 = taps 0..6) — once per kernel, into registers (7 rows x 4 blocks x 4 VGPRs),
// nothing of the weights ever touches LDS.  4 waves x 4 output rows; a lane ends up with 8 (+8) consecutive channels of one pixel:
// 16-byte NHWC stores.  BatchNorm partial sums of the stored values per tile into two registers, one row per workgroup.
#include "conv_common.h"
#include <stdlib.h>

namespace {

constexpr int ST_T = 16;                      // output tile edge
constexpr int ST_ROWS = 2 * ST_T + 5;         // 37 window rows
constexpr int ST_WP = 40;                     // window pitch in pixels (20 sixteen-byte pieces per row; 38 used)
constexpr int ST_PIECES = ST_ROWS * (ST_WP / 2);   // 740
constexpr int ST_BUF = 3 * 256 * 16;          // three DMA rounds of 256 lanes x 16 B = 12 KB >= 740 pieces

__device__ __forceinline__ u32x4 st_lds16(uint32_t addr) {
  u32x4 v;
  asm volatile("ds_read_b128 %0, %1" : "=v"(v) : "v"(addr));
  return v;
}

struct StemGeo {
  int TX, TY;            // tiles per image row / column
  FastDiv fd_tpi, fd_tx;
  int tiles;             // n * TY * TX
  int rows;              // statistics rows the caller sized (>= gridDim.x: the surplus is zero-filled)
};

__global__ __launch_bounds__(256, 2) void stem_win_kernel(ConvArgs a, StemGeo geo) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  typedef __attribute__((address_space(3))) char lds_char;
  typedef __attribute__((address_space(3))) void lds_void;
  const uint32_t lds_base = (uint32_t)(size_t)(lds_char*)smem;
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wm = tid >> 6;
  const int wave_u = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int sl = lane >> 4, li = lane & 15;

  const __amdgpu_buffer_rsrc_t xsrd = __builtin_amdgcn_make_buffer_rsrc((void*)a.x, 0, a.x_bytes, 0x00020000);

  // ---- weights: register-resident fragments, shifted by one tap (see the header) ------------------------------------------------
  // fragment (r, t), lane (li, sl): filter row n = (li >> 2) * 8 + (li & 3) + (t & 1) * 4 + (t >> 1) * 32, tap slots 2 sl, 2 sl + 1
  // = taps 2 sl - 1, 2 sl of row r (tap -1 = the zero slot)
  u32x4 wf[7][4];
#pragma unroll
  for (int t = 0; t < 4; ++t) {
    const int n = (li >> 2) * 8 + (li & 3) + (t & 1) * 4 + (t >> 1) * 32;
    const bool nok = n < a.K;
#pragma unroll
    for (int r = 0; r < 7; ++r) {
      const u32x2* row = reinterpret_cast<const u32x2*>(a.w + ((size_t)n * 7 + r) * 32);     // 8 taps x 4 channels = 8 u32x2
      u32x2 lo = {0u, 0u}, hi = {0u, 0u};
      if (nok && sl > 0) lo = row[2 * sl - 1];
      if (nok) hi = row[2 * sl];
      wf[r][t] = (u32x4){lo.x, lo.y, hi.x, hi.y};
    }
  }

  // ---- window loader ------------------------------------------------------------------------------------------------------
  auto issue_window = [&](int tile, int buf) {
    const bool live = tile < geo.tiles;
    const uint32_t img = fdiv((uint32_t)tile, geo.fd_tpi);
    const int rem = tile - (int)img * (geo.TX * geo.TY);
    const int ty = (int)fdiv((uint32_t)rem, geo.fd_tx);
    const int tx = rem - ty * geo.TX;
    const int iy0 = 2 * ty * ST_T - 3, ix0 = 2 * tx * ST_T - 4;
#pragma unroll
    for (int j = 0; j < 3; ++j) {
      const int pi = (j * 4 + wave_u) * 64 + lane;
      const int wr = pi / (ST_WP / 2), cp = pi - wr * (ST_WP / 2);
      const int iy = iy0 + wr, ix = ix0 + 2 * cp;
      const bool ok = live && pi < ST_PIECES && (unsigned)iy < (unsigned)a.H && (unsigned)ix < (unsigned)a.W;
      uint32_t off = ok ? (uint32_t)((((int)img * a.H + iy) * a.W + ix) * 8) : 0xFFFFFFF0u;
      asm volatile("" : "+v"(off));
      __builtin_amdgcn_raw_ptr_buffer_load_lds(xsrd, (lds_void*)(smem + buf * ST_BUF + (j * 4 + wave_u) * 1024), 16, off, 0, 0, 0);
    }
  };

  const int G = gridDim.x;
  float s1r = 0.f, s2r = 0.f;
  // fragment base of this lane: window row 2 * (wm * 4 + f) + r, pixel 2 li + 2 sl  ->  ((2 * (4 wm + f) + r) * WP + 2 li + 2 sl) * 8
  const uint32_t fbase = lds_base + (uint32_t)(((2 * 4 * wm) * ST_WP + 2 * li + 2 * sl) * 8);

  int tile = blockIdx.x;
  issue_window(tile, 0);
  int buf = 0;
  for (; tile < geo.tiles; tile += G) {
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");     // this tile's window has landed (own pieces) ...
    __builtin_amdgcn_s_barrier();                                    // ... everybody's; and everybody has left the other buffer
    issue_window(tile + G, buf ^ 1);

    f32x4 acc[4][4];
#pragma unroll
    for (int t = 0; t < 4; ++t)
#pragma unroll
      for (int f = 0; f < 4; ++f) acc[t][f] = (f32x4){0.f, 0.f, 0.f, 0.f};
    const uint32_t fb = fbase + (uint32_t)(buf * ST_BUF);
#pragma unroll
    for (int r = 0; r < 7; ++r) {
      u32x4 bf[4];
#pragma unroll
      for (int f = 0; f < 4; ++f) bf[f] = st_lds16(fb + (uint32_t)(((2 * f + r) * ST_WP) * 8));
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int f = 0; f < 4; ++f)
#pragma unroll
        for (int t = 0; t < 4; ++t)
          acc[t][f] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, wf[r][t]), __builtin_bit_cast(bf16x8, bf[f]),
                                                               acc[t][f], 0, 0, 0);
    }

    // ---- epilogue: lane (li, sl) holds channels sl * 8 + {0..7} and 32 + sl * 8 + {0..7} of pixel (oy, ox0 + li) ----------------
    const uint32_t img = fdiv((uint32_t)tile, geo.fd_tpi);
    const int rem = tile - (int)img * (geo.TX * geo.TY);
    const int ty = (int)fdiv((uint32_t)rem, geo.fd_tx);
    const int tx = rem - ty * geo.TX;
    const int ox = tx * ST_T + li;
    float s1[16], s2[16];
#pragma unroll
    for (int c = 0; c < 16; ++c) { s1[c] = 0.f; s2[c] = 0.f; }
#pragma unroll
    for (int f = 0; f < 4; ++f) {
      const int oy = ty * ST_T + wm * 4 + f;
      if (oy >= a.P || ox >= a.Q) continue;
      bf16* yp = a.y + ((size_t)((int)img * a.P + oy) * a.Q + ox) * a.K + sl * 8;
#pragma unroll
      for (int half = 0; half < 2; ++half) {
        if (sl * 8 + half * 32 + 8 > a.K) continue;
        bf16x8 o;
#pragma unroll
        for (int e = 0; e < 8; ++e) o[e] = f2bf(acc[half * 2 + (e >> 2)][f][e & 3]);
        stg16(yp + half * 32, o);
        if (a.stats != nullptr) {
#pragma unroll
          for (int e = 0; e < 8; ++e) {
            const float rv = bf2f(o[e]);
            s1[half * 8 + e] += rv;
            s2[half * 8 + e] = fmaf(rv, rv, s2[half * 8 + e]);
          }
        }
      }
    }
    if (a.stats != nullptr) {      // fold the 16 pixels of a row group into one lane per channel (the other kernels' butterfly)
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
    buf ^= 1;
  }
  asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");

  if (a.stats != nullptr) {
    __syncthreads();
    float* red = reinterpret_cast<float*>(smem);     // [2][4 waves][64]
    const int nl = (li >> 3) * 32 + sl * 8 + (li & 7);
    red[(0 * 4 + wm) * 64 + nl] = s1r;
    red[(1 * 4 + wm) * 64 + nl] = s2r;
    __syncthreads();
    if (tid < 128) {
      const int which = tid >> 6, c = tid & 63;
      const float t = red[(which * 4 + 0) * 64 + c] + red[(which * 4 + 1) * 64 + c] + red[(which * 4 + 2) * 64 + c] +
                      red[(which * 4 + 3) * 64 + c];
      if (c < a.K) {
        a.stats[((size_t)which * geo.rows + blockIdx.x) * a.K + c] = t;
        // the caller sized the rows for the persistent grid of the implicit-GEMM kernel: zero the ones this grid does not own
        for (int rr = blockIdx.x + G; rr < geo.rows; rr += G) a.stats[((size_t)which * geo.rows + rr) * a.K + c] = 0.f;
      }
    }
  }
}

// ---- the stem's weight gradient on the same window ----------------------------------------------------------------------------
// dW[k][r][s][c] = sum over output pixels of dy[pix][k] * x[2 oy + r - 3][2 ox + s - 3][c]: both MFMA operands have the reduction
// (pixels) along their rows, so both are read with ds_read_b64_tr_b16 — dy from a row-major [128 pixels][64 channels] LDS tile
// (LDS-DMA, XOR-swizzled 16-byte chunks like conv_wgrad.hip's), and the patch matrix [pixel][r][slot][c] is never built: a
// transpose read takes one address per lane, and element (pixel (oy, ox), r, slot, c) is the window pixel (2 oy + r, 2 ox + slot)
// (even-aligned window, padding tap in front: slot = tap + 1, as in the forward).  conv_wgrad_kernel's C4 path gathers the same
// patches with two 8-byte global loads per 16-byte chunk and an address computation per row: 241 us for the 256 x 224 x 224 batch,
// the last launch of the backward pass with the main stream idle behind it.  Tile = 8 x 16 output pixels (four 32-pixel MFMA
// steps), window 21 x 40 pixels (6.6 KB) + dy tile 16 KB, two buffers; 2 x 2 waves: 32 of the 64 filters x 7 of the 14
// sixteen-column blocks (7 rows x 2 halves of a filter row) each, 56 accumulator registers.  One [k][224] slab per workgroup
// (column = r * 32 + tap * 4 + c: the layout wgrad_reduce_kernel folds), as many workgroups as the plan has slabs.
constexpr int SG_TH = 8, SG_TW = 16;
constexpr int SG_WROWS = 2 * SG_TH + 5;                 // 21
constexpr int SG_WPIECES = SG_WROWS * (ST_WP / 2);      // 420
constexpr int SG_WBUF = 2 * 256 * 16;                   // 8 KB
constexpr int SG_YBUF = 128 * 128;                      // 16 KB
constexpr int SG_STAGE = SG_WBUF + SG_YBUF;

__device__ __forceinline__ u32x2 st_tr_read(uint32_t lds_addr) {
  u32x2 v;
  asm volatile("ds_read_b64_tr_b16 %0, %1" : "=v"(v) : "v"(lds_addr));
  return v;
}

struct StemWg {
  const bf16* x;
  const bf16* dy;
  float* ws;
  int H, W, P, Q, K;
  int TX, TY;
  FastDiv fd_tpi, fd_tx;
  int tiles;
  uint32_t x_bytes, dy_bytes;
};

__global__ __launch_bounds__(256, 2) void stem_wgrad_kernel(StemWg a) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  typedef __attribute__((address_space(3))) char lds_char;
  typedef __attribute__((address_space(3))) void lds_void;
  const uint32_t lds_base = (uint32_t)(size_t)(lds_char*)smem;
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wv = tid >> 6;
  const int wave_u = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wn2 = wv & 1, wk2 = wv >> 1;
  const int g = lane >> 4, li = lane & 15;
  const int rrow = 4 * g + (li >> 2);

  const __amdgpu_buffer_rsrc_t xsrd = __builtin_amdgcn_make_buffer_rsrc((void*)a.x, 0, a.x_bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t ysrd = __builtin_amdgcn_make_buffer_rsrc((void*)a.dy, 0, a.dy_bytes, 0x00020000);

  auto issue = [&](int tile, int buf) {
    const bool live = tile < a.tiles;
    const uint32_t img = fdiv((uint32_t)tile, a.fd_tpi);
    const int rem = tile - (int)img * (a.TX * a.TY);
    const int ty = (int)fdiv((uint32_t)rem, a.fd_tx);
    const int tx = rem - ty * a.TX;
    const int oy0 = ty * SG_TH, ox0 = tx * SG_TW;
    const int iy0 = 2 * oy0 - 3, ix0 = 2 * ox0 - 4;
    char* base = smem + buf * SG_STAGE;
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const int pi = (j * 4 + wave_u) * 64 + lane;
      const int wr = pi / (ST_WP / 2), cp = pi - wr * (ST_WP / 2);
      const int iy = iy0 + wr, ix = ix0 + 2 * cp;
      const bool ok = live && pi < SG_WPIECES && (unsigned)iy < (unsigned)a.H && (unsigned)ix < (unsigned)a.W;
      uint32_t off = ok ? (uint32_t)((((int)img * a.H + iy) * a.W + ix) * 8) : 0xFFFFFFF0u;
      asm volatile("" : "+v"(off));
      __builtin_amdgcn_raw_ptr_buffer_load_lds(xsrd, (lds_void*)(base + (j * 4 + wave_u) * 1024), 16, off, 0, 0, 0);
    }
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int q = (j * 4 + wave_u) * 64 + lane;           // LDS piece: pixel row y = q / 8, slot q % 8
      const int y = q >> 3, slot = q & 7;
      const int chunk = slot ^ ((y & 3) << 1);              // logical 8-channel chunk kept in that slot (conflict-free transpose reads)
      const int oy = oy0 + (y >> 4), ox = ox0 + (y & 15);
      const bool ok = live && oy < a.P && ox < a.Q && chunk * 8 < a.K;
      uint32_t off = ok ? (uint32_t)(((((int)img * a.P + oy) * a.Q + ox) * a.K + chunk * 8) * 2) : 0xFFFFFFF0u;
      asm volatile("" : "+v"(off));
      __builtin_amdgcn_raw_ptr_buffer_load_lds(ysrd, (lds_void*)(base + SG_WBUF + (j * 4 + wave_u) * 1024), 16, off, 0, 0, 0);
    }
  };

  f32x4 acc[2][7];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 7; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};

  // lane-constant parts of the fragment addresses
  const uint32_t yswz = (uint32_t)(((rrow & 3) << 1) << 4);
  uint32_t ycol[2];
#pragma unroll
  for (int i = 0; i < 2; ++i) ycol[i] = (uint32_t)((((wn2 * 2 + i) * 16 + (li & 3) * 4) * 2)) ^ yswz;
  // window: pixel (2 oy_l + r, 2 ox_l + slot), ox_l = rrow, slot = (J & 1) * 4 + (li & 3), J = wk2 * 7 + j, r = J >> 1
  uint32_t wcol[7];
#pragma unroll
  for (int j = 0; j < 7; ++j) {
    const int J = wk2 * 7 + j;
    wcol[j] = (uint32_t)((((J >> 1) * ST_WP) + 2 * rrow + (J & 1) * 4 + (li & 3)) * 8);
  }

  const int G = gridDim.x;
  int tile = blockIdx.x;
  issue(tile, 0);
  int buf = 0;
  for (; tile < a.tiles; tile += G) {
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    issue(tile + G, buf ^ 1);
    const uint32_t Wl = lds_base + (uint32_t)(buf * SG_STAGE), Yl = Wl + SG_WBUF;
#pragma unroll
    for (int st = 0; st < 4; ++st) {
      u32x2 ya[2][2], xb[7][2];
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        ya[i][0] = st_tr_read(Yl + (uint32_t)((st * 32 + rrow) * 128) + ycol[i]);
        ya[i][1] = st_tr_read(Yl + (uint32_t)((st * 32 + rrow + 16) * 128) + ycol[i]);
      }
#pragma unroll
      for (int j = 0; j < 7; ++j) {
        xb[j][0] = st_tr_read(Wl + (uint32_t)((2 * (2 * st) * ST_WP) * 8) + wcol[j]);
        xb[j][1] = st_tr_read(Wl + (uint32_t)((2 * (2 * st + 1) * ST_WP) * 8) + wcol[j]);
      }
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        const bf16x8 af = __builtin_bit_cast(bf16x8, (u32x4){ya[i][0][0], ya[i][0][1], ya[i][1][0], ya[i][1][1]});
#pragma unroll
        for (int j = 0; j < 7; ++j) {
          const bf16x8 bfr = __builtin_bit_cast(bf16x8, (u32x4){xb[j][0][0], xb[j][0][1], xb[j][1][0], xb[j][1][1]});
          acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af, bfr, acc[i][j], 0, 0, 0);
        }
      }
    }
    buf ^= 1;
  }
  asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");

  // D[n][col]: lane holds filters (wn2 * 2 + i) * 16 + g * 4 + reg, column li of block J = (r, slot = (J & 1) * 4 + (li >> 2), c = li & 3)
  float* out = a.ws + (size_t)blockIdx.x * a.K * 224;
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 7; ++j) {
      const int J = wk2 * 7 + j;
      const int slot = (J & 1) * 4 + (li >> 2);
      const int kcol = (J >> 1) * 32 + (slot - 1) * 4 + (li & 3);      // tap = slot - 1 (slot 0 is the padding tap in front)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int n = (wn2 * 2 + i) * 16 + g * 4 + r;
        if (n < a.K && slot >= 1) out[(size_t)n * 224 + kcol] = acc[i][j][r];
      }
    }
}

int stem_flag() {   // TOK_STEM_WIN=0: the stem stays on conv_igemm's C4 path (A/B switch)
  static const int v = [] { const char* e = getenv("TOK_STEM_WIN"); return (int)(e ? atoi(e) : 1); }();
  return v;
}

}  // namespace

bool stem_win_serves(const ConvArgs& a) {
  if (!stem_flag()) return false;
  if (!(a.C == 4 && a.R == 7 && a.S == 8 && a.stride == 2 && a.pad == 3)) return false;
  if (a.K % 8 != 0 || a.K > 64 || a.W % 2 != 0) return false;
  if (a.bias != nullptr || a.y2 != nullptr || a.act_x != nullptr || a.ep_scale != nullptr || a.fin_mode != 0 || a.accumulate) return false;
  return (long long)(a.M / (a.P * a.Q)) * tok_cdiv(a.P, ST_T) * tok_cdiv(a.Q, ST_T) >= 16;      // (tiny inputs: not worth a second kernel)
}

int stem_win_launch(ConvArgs& a, int stat_rows, hipStream_t st) {
  StemGeo g;
  g.TX = tok_cdiv(a.Q, ST_T);
  g.TY = tok_cdiv(a.P, ST_T);
  g.fd_tpi = make_fastdiv((uint32_t)(g.TX * g.TY));
  g.fd_tx = make_fastdiv((uint32_t)g.TX);
  g.tiles = (a.M / (a.P * a.Q)) * g.TX * g.TY;
  g.rows = stat_rows;
  int grid = 512;                                  // two workgroups per CU
  if (grid > g.tiles) grid = g.tiles;
  if (a.stats != nullptr && grid > stat_rows) grid = stat_rows;
  constexpr int smem = 2 * ST_BUF;
  hipLaunchKernelGGL(stem_win_kernel, dim3(grid), dim3(256), smem, st, a, g);
  return 0;
}

bool stem_wgrad_serves(const tok_conv_desc* d) {
  static const int on = [] { const char* e = getenv("TOK_STEM_WGRAD"); return (int)(e ? atoi(e) : 1); }();   // TOK_STEM_WGRAD=0: conv_wgrad_kernel's C4 path (A/B switch)
  if (!on || !stem_flag()) return false;
  if (!(d->c == 4 && d->r == 7 && d->s == 7 && d->s_pad == 8 && d->stride == 2 && d->pad == 3)) return false;
  if (d->k % 8 != 0 || d->k > 64 || d->w % 2 != 0) return false;
  return (long long)d->n * tok_cdiv(d->p, SG_TH) * tok_cdiv(d->q, SG_TW) >= 16;
}

// one [k][224] fp32 slab per workgroup into ws (slab stride k * 224), `slabs` workgroups
int stem_wgrad_launch(const tok_conv_desc* d, const void* x, const void* dy, float* ws, int slabs, hipStream_t st) {
  StemWg a;
  a.x = (const bf16*)x; a.dy = (const bf16*)dy; a.ws = ws;
  a.H = d->h; a.W = d->w; a.P = d->p; a.Q = d->q; a.K = d->k;
  a.TX = tok_cdiv(d->q, SG_TW); a.TY = tok_cdiv(d->p, SG_TH);
  a.fd_tpi = make_fastdiv((uint32_t)(a.TX * a.TY));
  a.fd_tx = make_fastdiv((uint32_t)a.TX);
  a.tiles = d->n * a.TX * a.TY;
  a.x_bytes = (uint32_t)((unsigned long long)d->n * d->h * d->w * 8);
  a.dy_bytes = (uint32_t)((unsigned long long)d->n * d->p * d->q * d->k * 2);
  hipLaunchKernelGGL(stem_wgrad_kernel, dim3(slabs), dim3(256), 2 * SG_STAGE, st, a);
  return 0;
}
