// Pointwise convolution / Linear as a 256 x 256-tile GEMM for gfx950: the deep-K, mid-M layers that are neither HBM- nor
// MFMA-bound on the 128 x 128 two-buffer kernel (conv_igemm.hip).
//
//   y[m][n] = sum_k x[m][k] w[n][k]            x: (M, K) activations, w: (N, K) operand pack, both K-contiguous bf16
//
// covers nn.Conv2d 1x1 / stride 1 forward and data gradient ([timm] Bottleneck conv1 / conv3 / downsample of ResNet layers 3-4,
// /root/reference/torchok/models/backbones/resnet.py:12-14,363-405) and nn.Linear forward / data gradient of the SwinV2 / DaViT
// blocks' qkv, proj, fc1, fc2 on the later stages (/root/reference/torchok/models/backbones/swin.py:18-20, davit.py:133-207).
//
// Why another tile: on those layers (M = 12 544 ... 50 176 pixels, K and N = 256 ... 3072) conv_igemm's 64 x 64 wave tile reads
// one 1-KB LDS fragment per two MFMAs — 250 B/clk/CU with two resident workgroups, the whole LDS bandwidth — and stages
// (128 + 128) x 64 x 2 B per 512 MFMA cycles from L2, above the ~40 B/clk/CU that path sustains: 500-650 TF/s, "neither bound"
// (profiles/r03_*: MFMA util 0.18-0.19).  Here a workgroup of EIGHT waves (2 along the channels x 4 along the pixels) owns a
// 256 x 256 tile, a wave 128 channels x 64 pixels = 32 accumulator blocks: 12 fragment reads per 32 MFMAs (190 B/clk/CU) and
// 64 KB staged per 2048 MFMA cycles (32 B/clk/CU) — the geometry of the programming guide's 256^2 template.
//   * weights are the MFMA A operand (rows = output channels), pixels the B operand, so D registers hold 4 consecutive rows =
//     channels of ONE pixel; the rows of a 32-channel pair of blocks are dealt {8q .. 8q+3} / {8q+4 .. 8q+7}, which makes a
//     lane's two blocks 8 consecutive channels: one 16-byte NHWC store per pixel and block pair.
//   * global -> LDS DMA (buffer_load ... lds), two 64-KB stages, rows of 64 k = 128 bytes, 16-byte chunks XOR-swizzled with
//     (row & 3) | ((row >> 3) & 1) << 2 on the SOURCE side: conflict-free ds_read_b128 for both the dealt weight rows and the
//     contiguous pixel rows (enumerated against the bank rules of MI355X_MICROARCH.md).
//   * epilogues of the call sites it serves: bias, accumulate onto an existing gradient, per-channel sum / sum of squares of the
//     stored bf16 output (BatchNorm batch statistics, one partial row per pixel tile), BatchNorm-backward sums of the produced
//     gradient (sum dz, sum dz*y with the ReLU bits of the producing unit).  Everything else stays on conv_igemm.hip.
#include "conv_common.h"
#include <stdlib.h>

namespace {

typedef __attribute__((address_space(3))) void lds_void;
typedef __attribute__((address_space(3))) char lds_char;

constexpr int GT = 256;                 // tile edge, both ways
constexpr int GK = 64;                  // reduction elements per stage
constexpr int TILE_B = GT * GK * 2;     // one operand tile: 256 rows x 128 bytes
constexpr int STAGE_B = 2 * TILE_B;     // [weights | pixels]
constexpr int NBLK = STAGE_B / 1024;    // 1-KB DMA blocks per stage
constexpr int NI = NBLK / 8;            // per wave

__device__ __forceinline__ u32x4 lds_read16(uint32_t addr) {
  u32x4 v;
  asm volatile("ds_read_b128 %0, %1" : "=v"(v) : "v"(addr));
  return v;
}
__device__ __forceinline__ float row16_sum_f(float v) {
  v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x128, 0xF, 0xF, false));
  v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x124, 0xF, 0xF, false));
  v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x122, 0xF, 0xF, false));
  v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x121, 0xF, 0xF, false));
  return v;
}

// MODE 0: plain (bias, accumulate); 1: + batch statistics of the stored output; 2: + BatchNorm-backward sums (bn_y, bn_mask);
// 3: dz = relu_mask ? dx : 0 is what gets stored, and sum(dz) goes to the statistics rows (tok_conv_dgrad_maskstore)
template <int MODE>
__global__ __launch_bounds__(512, 1) void gemm256_kernel(ConvArgs a) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const uint32_t lds_base = (uint32_t)(size_t)(lds_char*)smem;
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int r = lane & 15, g = lane >> 4;
  const int wn = wave & 1, wm = wave >> 1;

  const int ntiles = a.gridM * a.gridN;
  const int KD = a.Ktot;                                   // reduction length = row pitch of both operands (elements)
  const int KT = (KD + GK - 1) / GK;               // (KD % 8 == 0: a 16-byte chunk is inside or outside the row)

  const __amdgpu_buffer_rsrc_t xsrd = __builtin_amdgcn_make_buffer_rsrc((void*)a.x, 0, a.x_bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t wsrd = __builtin_amdgcn_make_buffer_rsrc((void*)a.w, 0, a.w_bytes, 0x00020000);

  // ---- stage DMA: block bid = j * 8 + wave; blocks 0 .. 31 weight rows, 32 .. 63 pixel rows, 8 rows of 128 bytes each ----------
  uint32_t voff[NI];
  int kq[NI];              // first reduction element of this lane's 16-byte chunk inside a stage (K tail: chunks past K are zeros)
#pragma unroll
  for (int j = 0; j < NI; ++j) {
    const int row = ((j * 8 + wave) & 31) * 8 + (lane >> 3);
    kq[j] = ((lane & 7) ^ ((row & 3) | (((row >> 3) & 1) << 2))) * 8;
  }
  // a workgroup walks tiles id, id + grid, ...; consecutive ids share a pixel tile (the channel tiles of one pixel tile on one XCD's L2)
  auto setup = [&](int id, int& m0, int& n0) {
    const int mt = id / a.gridN, nt = id - mt * a.gridN;
    m0 = mt * GT;
    n0 = nt * GT;
#pragma unroll
    for (int j = 0; j < NI; ++j) {
      const int bid = j * 8 + wave;
      const int row = (bid & 31) * 8 + (lane >> 3);
      const bool isw = bid < 32;
      const int grow = (isw ? n0 : m0) + row;
      const bool ok = grow < (isw ? a.K : a.M);
      voff[j] = ok ? (uint32_t)(((long long)grow * KD + kq[j]) * 2) : 0xFFFFFFF0u;
    }
  };
  auto issue = [&](int kt, int buf, bool live) {
#pragma unroll
    for (int j = 0; j < NI; ++j) {
      const int bid = j * 8 + wave;
      uint32_t off = (live && voff[j] != 0xFFFFFFF0u && kt * GK + kq[j] < KD) ? voff[j] + (uint32_t)(kt * (GK * 2)) : 0xFFFFFFF0u;
      asm volatile("" : "+v"(off));
      lds_void* dst = (lds_void*)(smem + buf * STAGE_B + bid * 1024);
      if (bid < 32) __builtin_amdgcn_raw_ptr_buffer_load_lds(wsrd, dst, 16, off, 0, 0, 0);
      else __builtin_amdgcn_raw_ptr_buffer_load_lds(xsrd, dst, 16, off, 0, 0, 0);
    }
  };

  // ---- fragment addressing: the swizzle term is lane-constant for both operands ------------------------------------------------
  const uint32_t swz = (uint32_t)((r & 3) | (((r >> 2) & 1) << 2));            // weights: dealt row 8 (r >> 2) + 4 b + (r & 3)
  const uint32_t swx = (uint32_t)((r & 3) | (((r >> 3) & 1) << 2));            // pixels: row 16 mb + r
  // weight fragment (pair p, block b): row wn*128 + 32 p + 8 (r >> 2) + 4 b + (r & 3)
  const uint32_t a_lane = (uint32_t)((wn * 128 + 8 * (r >> 2) + (r & 3)) * 128);
  // pixel fragment (block mb): row wm*64 + 16 mb + r
  const uint32_t b_lane = (uint32_t)(TILE_B + (wm * 64 + r) * 128);
  uint32_t a_ch[2], b_ch[2];
#pragma unroll
  for (int kk = 0; kk < 2; ++kk) {
    a_ch[kk] = (((uint32_t)(4 * kk + g)) ^ swz) << 4;
    b_ch[kk] = (((uint32_t)(4 * kk + g)) ^ swx) << 4;
  }

  // Persistent walk over the tiles.  Per tile, one barrier per K stage: wait for this wave's share of stage kt, barrier (everyone's
  // share landed, nobody reads the other buffer any more), request stage kt + 1 into the other buffer, then two 32-deep steps of
  // 12 fragment reads + 32 MFMAs; the two waves of a SIMD interleave their read and MFMA phases.  The first stage of the NEXT tile is
  // requested before the epilogue of this one (its buffer is free since the barrier of the last stage: KT >= 2), so a tile's
  // ~2 us of first-stage latency hide behind the previous tile's stores.  (Measured and not kept: fragment reads one step ahead
  // of the MFMAs on a second register set with the DMA two stages ahead — 255 registers, 3-7 % slower on every shape.)
  int tile = tok_xcd_remap(blockIdx.x, gridDim.x);
  int m0, n0;
  setup(tile, m0, n0);
  int base = 0;
  issue(0, base, tile < ntiles);
  while (tile < ntiles) {
  f32x4 acc[4][2][4];
#pragma unroll
  for (int p = 0; p < 4; ++p)
#pragma unroll
    for (int b = 0; b < 2; ++b)
#pragma unroll
      for (int mb = 0; mb < 4; ++mb) acc[p][b][mb] = (f32x4){0.f, 0.f, 0.f, 0.f};
  for (int kt = 0; kt < KT; ++kt) {
    const int buf = (base + kt) & 1;
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    issue(kt + 1, buf ^ 1, kt + 1 < KT);
    const uint32_t sb = lds_base + (uint32_t)(buf * STAGE_B);
#pragma unroll
    for (int kk = 0; kk < 2; ++kk) {
      u32x4 af[4][2], bfr[4];
#pragma unroll
      for (int mb = 0; mb < 4; ++mb) bfr[mb] = lds_read16(sb + b_lane + (uint32_t)(mb * 16 * 128) + b_ch[kk]);
#pragma unroll
      for (int p = 0; p < 4; ++p)
#pragma unroll
        for (int b = 0; b < 2; ++b) af[p][b] = lds_read16(sb + a_lane + (uint32_t)((32 * p + 4 * b) * 128) + a_ch[kk]);
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int p = 0; p < 4; ++p)
#pragma unroll
        for (int b = 0; b < 2; ++b)
#pragma unroll
          for (int mb = 0; mb < 4; ++mb)
            acc[p][b][mb] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, af[p][b]),
                                                                    __builtin_bit_cast(bf16x8, bfr[mb]), acc[p][b][mb], 0, 0, 0);
    }
  }
  // next tile's first stage (the last stage's barrier freed the other buffer), then this tile's epilogue
  const int cm0 = m0, cn0 = n0, cmt = m0 / GT;
  const int next = tile + gridDim.x;
  base = (base + KT) & 1;
  if (next < ntiles) {
    setup(next, m0, n0);
    issue(0, base, true);
  }

  // ---- epilogue: lane (r, g) holds channels n0 + wn*128 + 32 p + 8 g .. + 8 of pixel m0 + wm*64 + 16 mb + r ---------------------
  float s1[4][8], s2[4][8];
  if constexpr (MODE >= 1 && MODE <= 3) {
#pragma unroll
    for (int p = 0; p < 4; ++p)
#pragma unroll
      for (int e = 0; e < 8; ++e) { s1[p][e] = 0.f; s2[p][e] = 0.f; }
  }
#pragma unroll
  for (int p = 0; p < 4; ++p) {
    const int n8 = cn0 + wn * 128 + 32 * p + 8 * g;
    const bool nok = n8 + 8 <= a.K;
    float bv[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) bv[e] = (a.bias != nullptr && nok) ? a.bias[n8 + e] : 0.f;
#pragma unroll
    for (int mb = 0; mb < 4; ++mb) {
      const int pix = cm0 + wm * 64 + 16 * mb + r;
      if (pix < a.M && nok) {
        const size_t off = (size_t)pix * a.K + n8;
        float v[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) v[e] = acc[p][e >> 2][mb][e & 3] + bv[e];
        if (a.accumulate) {
          const bf16x8 old = ldg16(a.y + off);
#pragma unroll
          for (int e = 0; e < 8; ++e) v[e] += bf2f(old[e]);
        }
        bf16x8 o;
#pragma unroll
        for (int e = 0; e < 8; ++e) o[e] = f2bf(v[e]);
        if constexpr (MODE == 3) {
          const unsigned bits = (unsigned)a.bn_mask[off >> 3];
#pragma unroll
          for (int e = 0; e < 8; ++e) {
            if (!((bits >> e) & 1u)) o[e] = (bf16)0.f;
            s1[p][e] += bf2f(o[e]);
          }
        }
        stg16(a.y + off, o);
        if constexpr (MODE == 1) {
          // batch statistics of the bf16 output as stored (what bn_act_fwd normalises; conv_igemm.hip's rounding point)
#pragma unroll
          for (int e = 0; e < 8; ++e) {
            const float q = bf2f(o[e]);
            s1[p][e] += q;
            s2[p][e] = fmaf(q, q, s2[p][e]);
          }
        }
        if constexpr (MODE == 2) {
          const bf16x8 yv = ldg16(a.bn_y + off);
          const unsigned bits = a.bn_mask != nullptr ? (unsigned)a.bn_mask[off >> 3] : 0xffu;
#pragma unroll
          for (int e = 0; e < 8; ++e) {
            const float dz = ((bits >> e) & 1u) ? bf2f(o[e]) : 0.f;
            s1[p][e] += dz;
            s2[p][e] = fmaf(dz, bf2f(yv[e]), s2[p][e]);
          }
        }
      }
    }
  }
  if constexpr (MODE >= 1 && MODE <= 3) {
    // one partial row per pixel tile: 16 pixel-lanes folded by DPP, the four pixel-waves of a channel range through LDS (a region
    // behind the two stages: the next tile's first stage may be landing in them)
    float* red = reinterpret_cast<float*>(smem + 2 * STAGE_B);       // [2][4 wm][256 n]
#pragma unroll
    for (int p = 0; p < 4; ++p)
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        const float t1 = row16_sum_f(s1[p][e]), t2 = row16_sum_f(s2[p][e]);
        if (r == 15) {                                  // (row_ror sums land in every lane of the row; one lane writes)
          const int nl = wn * 128 + 32 * p + 8 * g + e;
          red[(0 * 4 + wm) * GT + nl] = t1;
          red[(1 * 4 + wm) * GT + nl] = t2;
        }
      }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    const int which = tid >> 8, c = tid & 255;
    const float t = red[(which * 4 + 0) * GT + c] + red[(which * 4 + 1) * GT + c] + red[(which * 4 + 2) * GT + c] +
                    red[(which * 4 + 3) * GT + c];
    const int n = cn0 + c;
    if (n < a.K) a.stats[((size_t)which * a.stat_rows + cmt) * a.K + n] = t;
    if (cmt == a.gridM - 1 && n < a.K)                  // the pad rows behind the last pixel tile
      for (int rr = a.gridM; rr < a.stat_rows; ++rr) a.stats[((size_t)which * a.stat_rows + rr) * a.K + n] = 0.f;
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();                        // red is free again before the next tile's epilogue writes it
  }
  tile = next;
  }   // tiles
  asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
}

int g256_flag() {
  static const int v = [] { const char* e = getenv("TOK_GEMM256"); return (int)(e ? atoi(e) : 1); }();
  return v;
}
int g256_min_k() {
  static const int v = [] { const char* e = getenv("TOK_GEMM256_MIN_K"); return (int)(e ? atoi(e) : 256); }();
  return v;
}
int g256_min_tiles() {
  static const int v = [] { const char* e = getenv("TOK_GEMM256_MIN_TILES"); return (int)(e ? atoi(e) : 128); }();
  return v;
}

}  // namespace

// Two tests.  GEOMETRY (a pure function of the layer) is what the statistics-row queries see — they know the descriptor, not the
// epilogue mode — so every mode that writes statistics rows must be served whenever the geometry is (plain, BatchNorm forward /
// backward sums, mask-store: all here; fused activation has no rows); the modes below carry no rows (or, fused finalize, fold their own) and fall back freely.
bool gemm256_modes(const ConvArgs& a) {
  if (a.ep_scale != nullptr || a.sub != nullptr || a.fin_mode != 0) return false;
  if (a.mask_store && (a.bn_mask == nullptr || a.stats == nullptr)) return false;
  // (fused activation — round 4's gemm256_kernel<4> — was bit-identical and 0.06 ms slower on the step: DESIGN.md "measured and rejected")
  if (a.y2 != nullptr || a.act_x != nullptr) return false;
  return true;
}
bool gemm256_serves(const ConvArgs& a) { return gemm256_geometry(a) && gemm256_modes(a); }

bool gemm256_geometry(const ConvArgs& a) {
  const int flag = g256_flag();
  if (!flag) return false;
  if (!(a.R == 1 && a.S == 1 && a.stride == 1 && a.pad == 0 && a.C != 4)) return false;
  if (a.Ktot % 8 != 0 || a.K % 8 != 0 || a.Ktot <= GK) return false;      // (two stages at least: the next-tile prefetch relies on it)
  if (flag >= 3) return true;                   // TOK_GEMM256=3: every layer the kernel can run at all (stress runs of the test suite)
  if (a.K < 192 || a.M < 4096) return false;
  const long long tiles = (long long)tok_cdiv(a.M, GT) * tok_cdiv(a.K, GT);
  // a ragged last channel tile wastes its empty part: at most a quarter of the work
  const int nt = tok_cdiv(a.K, GT);
  if ((long long)nt * GT * 3 > (long long)a.K * 4) return false;
  if (tiles < g256_min_tiles()) return false;
  if (flag >= 2) return a.Ktot >= g256_min_k();      // TOK_GEMM256=2: every layer the kernel can run (tests, A/B)
  // Default rule, decided on the STEP (same-box A/B, round 4): every pointwise layer with a reduction of 384 or more and at least
  // 128 tiles (half a tile per CU: the 98-tile 2048 -> 512 layer of ResNet-50's last stage takes 55 us here against 35 us).  In isolation the kernel only wins on the deepest / widest layers (see g256_flag); inside a step, where every kernel
  // runs 1.3-1.8x slower than alone, its lower LDS and L2 -> LDS traffic per MFMA pays more widely: SwinV2-T 20.71 -> 20.33 ms,
  // DaViT-T 22.07 -> 21.84, HRNet-W48 76.54 -> 76.13, ResNet-50 18.36 -> 18.22 (reduction >= 256: 18.24; >= 1024: 18.28).
  return a.Ktot >= 384;
}

// (a multiple of 8: conv_igemm.hip's kernels, which take over for the modes this one does not carry, run grids of 8 * gridN)
int gemm256_rows(const ConvArgs& a) { return (tok_cdiv(a.M, GT) + 7) / 8 * 8; }

int gemm256_launch(ConvArgs& a, hipStream_t st) {
  constexpr int smem = 2 * STAGE_B + 2 * 4 * GT * 4;
  static const bool attr_set = [] {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm256_kernel<0>), hipFuncAttributeMaxDynamicSharedMemorySize, smem);
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm256_kernel<1>), hipFuncAttributeMaxDynamicSharedMemorySize, smem);
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm256_kernel<2>), hipFuncAttributeMaxDynamicSharedMemorySize, smem);
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm256_kernel<3>), hipFuncAttributeMaxDynamicSharedMemorySize, smem);
    return true;
  }();
  (void)attr_set;
  a.gridM = tok_cdiv(a.M, GT);
  a.gridN = tok_cdiv(a.K, GT);
  a.stat_rows = gemm256_rows(a);
  const int tiles = a.gridM * a.gridN;
  const int grid = tiles < 256 ? tiles : 256;          // one workgroup per CU, walking tiles
  if (a.mask_store) hipLaunchKernelGGL(gemm256_kernel<3>, dim3(grid), dim3(512), smem, st, a);
  else if (a.stats == nullptr) hipLaunchKernelGGL(gemm256_kernel<0>, dim3(grid), dim3(512), smem, st, a);
  else if (a.bn_y == nullptr) hipLaunchKernelGGL(gemm256_kernel<1>, dim3(grid), dim3(512), smem, st, a);
  else hipLaunchKernelGGL(gemm256_kernel<2>, dim3(grid), dim3(512), smem, st, a);
  return 0;
}
