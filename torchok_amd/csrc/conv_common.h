// Argument block and fast division shared by the convolution kernels of libtok_gfx950.so (conv_igemm.hip: two-buffer
// implicit GEMM; conv_win.hip / conv_s2d.hip / stem.hip: shared-window kernels; gemm256.hip).  Not part of the C ABI.
#pragma once
#include "tok_common.h"

struct FastDiv {
  uint32_t mul, shift;
};
static inline FastDiv make_fastdiv(uint32_t d) {
  FastDiv f;
  if (d <= 1) { f.mul = 0; f.shift = 0; return f; }
  uint32_t l = 0;
  while ((1ull << l) < d) ++l;
  f.mul = (uint32_t)((((1ull << 32) * ((1ull << l) - d)) / d) + 1);
  f.shift = l;
  return f;
}
__device__ __forceinline__ uint32_t fdiv(uint32_t n, FastDiv f) {
  return (uint32_t)(((uint64_t)__umulhi(n, f.mul) + n) >> f.shift);
}

struct ConvArgs {
  const bf16* x;
  const bf16* w;
  bf16* y;
  const float* bias;
  float* stats;  // [2][stat_rows][K] or null
  bf16* y2;                // PWM 3 forward: act(y) is stored here beside the pre-activation y (fc1 -> GELU)
  const bf16* act_x;       // PWM 3 dgrad: pre-activation of the tensor whose gradient is produced: dx = dy * act'(act_x)
  int act;                 // 0 ReLU, 1 GELU (erf)
  const bf16* bn_y;        // dgrad + BatchNorm-backward statistics: raw conv output of the unit that
  const uint8_t* bn_mask;  // produced the tensor whose gradient this launch completes, and its ReLU bits
  // PWM 4 forward ("unit 3": 1x1 conv whose BatchNorm scale / shift are known BEFORE the GEMM): the epilogue applies
  // out = relu(acc * scale + shift + shortcut) and emits the ReLU bits; the pre-normalisation tensor is never stored
  const float* ep_scale;
  const float* ep_shift;
  const bf16* ep_short;    // may be null: no residual term
  uint8_t* ep_mask;
  int ep_relu;
  // dgrad completing the gradient of such a unit's output: the epilogue stores dz = relu_mask ? dx : 0 (what the unit's
  // backward and its shortcut both want) and reduces sum(dz) into the statistics rows; bn_y is not needed
  int mask_store;
  int force_grid;          // > 0: persistent grid size decided by the caller (statistics rows sized for the ring kernel)
  // pointwise dgrad on the ring whose result also takes the gradient of the stride-2 pixel subsample of the same tensor
  // (tok_conv_dgrad_subacc): dx[b][h][w] = acc + (h, w even ? sub[b][h/2][w/2] : 0)
  const bf16* sub;
  int H, W, C;   // gathered tensor
  int K;         // output channels (padded count of y)
  int R, S;      // S = stored filter width (s_pad)
  int P, Q;      // output spatial
  int stride, pad;
  int M, PQ, Ktot, KT;
  int gridM, gridN;
  int accumulate, uniform_taps;
  uint32_t x_bytes, w_bytes;   // extents of x / w for the buffer descriptors (< 4 GiB)
  uint32_t s_bytes;            // BNEP: extent of the shortcut tensor
  FastDiv fd_pq, fd_q;
  unsigned long long* timing;   // TOK_TIMING builds only: per-phase cycle totals of wave 0
  int stat_rows;   // workgroups per channel tile = rows of the partial-statistics buffer
  // fused BatchNorm finalize: the LAST workgroup of a channel tile to deliver its statistics row (device
  // ticket counter) folds the rows of that tile — saves the separate finalize launch between two dependent
  // kernels.  fin_mode 0: off, 1: forward (mean/rstd/scale/shift/running stats), 2: backward (dgamma/dbeta/coef)
  int fin_mode;
  tok_bn_fused fin;
  // IN_DIV == 2: per parity class (ph*2 + pw); m-tile index = 4 * (tile inside class) + class
  int cls_M[4], cls_nw[4], cls_hw[4];
  FastDiv cls_fd_hw[4], cls_fd_w[4];
};


// conv_win.hip: 3x3 / stride 1 / padding 1 layers on a shared input window
bool conv_win_serves(const ConvArgs& a);
int conv_win_grid(int gridM, int gridN);
void conv_win_tiles(const ConvArgs& a, int* gridM, int* gridN);
int conv_win_launch(ConvArgs& a, hipStream_t st);            // fills a.gridM / a.gridN / a.stat_rows itself
// conv_s2d.hip: data gradient of 3x3 / stride 2 / padding 1 layers on a shared dY window (`a` as dgrad_fill leaves it)
bool conv_s2d_serves(const ConvArgs& a, int stride, int pad);
int conv_s2d_grid(int gridM, int gridN);
void conv_s2d_tiles(const ConvArgs& a, int* gridM, int* gridN);
int conv_s2d_launch(ConvArgs& a, hipStream_t st);            // fills a.gridM / a.gridN / a.stat_rows itself

// gemm256.hip: pointwise layers with a deep reduction and a mid-sized pixel count on 256 x 256 tiles (8 waves)
bool gemm256_geometry(const ConvArgs& a);                    // does the kernel own this layer (pure function of the geometry)
bool gemm256_modes(const ConvArgs& a);                       // ... and carry this epilogue mode
bool gemm256_serves(const ConvArgs& a);                      // both
int gemm256_rows(const ConvArgs& a);                         // statistics rows = pixel tiles, rounded up to a multiple of 8 (pad rows are zero)
int gemm256_launch(ConvArgs& a, hipStream_t st);             // fills a.gridM / a.gridN / a.stat_rows itself

// stem.hip: the 7x7 / stride 2 stem convolution of a 4-channel-padded image on a shared input window
bool stem_win_serves(const ConvArgs& a);
int stem_win_launch(ConvArgs& a, int stat_rows, hipStream_t st);   // stat_rows: rows of a.stats as sized by tok_conv_fwd_stat_rows
bool stem_wgrad_serves(const tok_conv_desc* d);
int stem_wgrad_launch(const tok_conv_desc* d, const void* x, const void* dy, float* ws, int slabs, hipStream_t st);
