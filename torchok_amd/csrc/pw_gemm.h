// Internal interface of the pointwise (1x1 / stride 1 / token-matrix) GEMM with the three-stage DMA ring (pw_gemm.hip).
// conv_igemm.hip routes its pointwise launches here; not part of the C ABI.
#pragma once
#include "tok_common.h"

struct PwArgs {
  const bf16* x;        // [M][C]  activations (forward) or output gradients (dgrad)
  const bf16* w;        // [N][C]  forward pack / dgrad pack of a 1x1 filter
  bf16* y;              // [M][N]
  const float* bias;    // [N] or null
  float* stats;         // [2][stat_rows][N] partial sums or null
  int stat_rows;
  const bf16* e1;       // epilogue operand tile, same shape as y: previous value (accumulate) or shortcut (BatchNorm epilogue)
  const bf16* e2;       // raw conv output of the producer (BatchNorm-backward sums of the completed gradient), or null
  const uint8_t* mask_in;   // ReLU bits of the producer: e2 statistics / mask-store
  uint8_t* mask_out;        // BatchNorm epilogue: ReLU bits of the result
  const float* ep_scale;    // BatchNorm epilogue when non-null: out = act(acc * scale + shift (+ e1))
  const float* ep_shift;
  int ep_relu;
  int accumulate;       // y = e1 + acc  (e1 == y's previous contents)
  // e1_sub: e1 is the gradient of the stride-2 pixel subsample of y's tensor ([B][ceil(H/2)][ceil(W/2)][N], tok_subsample2_fwd):
  // row m = (b, h, w) of y receives e1[b][h/2][w/2] when h and w are even and nothing otherwise (accumulate must be set)
  int e1_sub, sub_H, sub_W;
  int mask_store;       // store relu_mask * y, sum it into stats row 0
  // second reduction source (may be null): Y += X2 W2^T with X2 [M][C2], W2 [N][C2] — its K stages follow those of (x, w) in
  // the same tile, so a sum of two pointwise products is one launch and one store of Y (the fused residual unit's
  // d(input) = dz Wa + z Wb)
  const bf16* x2;
  const bf16* w2;
  int C2;
  int M, C, N;
  int gridM, gridN;
};

// 0 = launched; > 0: this configuration is not served by the ring kernel (caller falls back); < 0: error
int pw_ring_launch(const PwArgs& a, int bn_tile, hipStream_t st);
int pw_ring_grid(int bn_tile, int gridM, int gridN);   // persistent grid size (for the statistics-row count)
bool pw_ring_enabled();
