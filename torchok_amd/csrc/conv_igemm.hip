// Implicit-GEMM convolution on MFMA for gfx950: forward and data-gradient.
//
//   Y[m][n] = sum_k A[m][k] * Wt[n][k]      m = (b, p, q)   k = (r, s, c)   NHWC bf16
//
// One kernel serves both directions:
//   fwd   : A gathered from x at (p*stride - pad + r, q*stride - pad + s)
//   dgrad : "x" is dY, the weights are the flipped/transposed pack ([C][R][S][K]), the gather
//           runs at stride 1 / pad R-1-pad with an input divisor IN_DIV = conv stride: a tap
//           contributes only where (h - pad' + r') is divisible by the stride.
//
// Tile: BM=128 output pixels x BN in {64,128} channels x BK=64, 256 threads (4 waves),
// mfma_f32_16x16x32_bf16.  The MFMA "A" operand is the WEIGHT tile and the "B" operand the
// activation tile, so D[i=channel][j=pixel]: each lane ends up holding 16 CONSECUTIVE output
// channels of one pixel (rows of the weight tile are fed in the order
// n = (i>>2)*16 + t*4 + (i&3)), i.e. the epilogue is two 16-byte NHWC stores per pixel and the
// per-channel BatchNorm partial sums are a 16-lane butterfly.
//
// Staging: global -> registers (next tile in flight during the MFMAs of the current one)
// -> LDS (two buffers, one barrier per K step), 16-byte XOR-swizzled slots so every
// ds_read_b128 of a fragment is bank-conflict free (see swizzle notes at the reads).
#include "tok_common.h"

namespace {

struct ConvArgs {
  const bf16* x;
  const bf16* w;
  bf16* y;
  const float* bias;
  float* stats;  // [2][gridM][K] or null
  int H, W, C;   // gathered tensor
  int K;         // output channels (padded count of y)
  int R, S;      // S = stored filter width (s_pad)
  int P, Q;      // output spatial
  int stride, pad;
  int M, PQ, Ktot, KT;
  int gridM, gridN;
  int accumulate;
};

constexpr int BK = 64;

__device__ __forceinline__ int fw_swz(int n) {
  // weight-tile slot swizzle: rows read together by one ds_read_b128 lane group are
  // n = q*16 + t*4 + i (q,i in 0..3).  h = [0,2,3,1] separates the q's that share a lane
  // group, bit 2 separates i>>1; (i&1) already lands in the other half of the 256-B row.
  return ((0x78 >> (((n >> 4) & 3) << 1)) & 3) | (((n >> 1) & 1) << 2);
}

template <int BM, int BN, int IN_DIV, bool C4>
__global__ __launch_bounds__(256) void conv_igemm_kernel(ConvArgs a) {
  constexpr int WGN = BN / 64;
  constexpr int WGM = 4 / WGN;
  constexpr int MT = BM / (WGM * 16);
  constexpr int AROWS = BM / 32;  // A rows staged per thread
  constexpr int WROWS = BN / 32;  // W rows staged per thread
  constexpr int TILE_BYTES = (BM + BN) * BK * 2;

  extern __shared__ __attribute__((aligned(16))) char smem[];

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wv = tid >> 6;
  const int wn = wv % WGN;
  const int wm = wv / WGN;

  const int tile = tok_xcd_remap(blockIdx.x, a.gridM * a.gridN);
  const int bm = tile / a.gridN;
  const int bn = tile - bm * a.gridN;
  const int m0 = bm * BM;
  const int n0 = bn * BN;

  // ---- staging assignment: thread -> (16-byte k-chunk kc, rows lrow + 32*i) -----------------
  const int kc = tid & 7;
  const int lrow = tid >> 3;

  int h0[AROWS], w0[AROWS], pix[AROWS];
#pragma unroll
  for (int i = 0; i < AROWS; ++i) {
    const int m = m0 + lrow + 32 * i;
    if (m < a.M) {
      const int b = m / a.PQ;
      const int rem = m - b * a.PQ;
      const int p = rem / a.Q;
      const int q = rem - p * a.Q;
      h0[i] = p * a.stride - a.pad;
      w0[i] = q * a.stride - a.pad;
      pix[i] = b * a.H * a.W;
    } else {
      h0[i] = -0x40000000;
      w0[i] = 0;
      pix[i] = 0;
    }
  }
  // k-chunk cursor (tap r,s and channel offset c0 of this thread's 8 elements)
  int kr, ks, kc0;
  if (C4) {
    kr = kc >> 2;            // k0 = kc*8 ; r = k0 / 32
    ks = (kc & 3) << 1;      // s = (k0 % 32) / 4
    kc0 = 0;
  } else {
    const int k0 = kc * 8;
    const int tap = k0 / a.C;
    kc0 = k0 - tap * a.C;
    kr = tap / a.S;
    ks = tap - kr * a.S;
  }
  size_t wk = (size_t)kc * 8;  // offset of this thread's chunk inside a weight row

  bf16x8 ra[AROWS], rw[WROWS];

  auto load_tile = [&]() {
    const bool kvalid = kr < a.R;
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
        const bf16* ptr = a.x + ((size_t)(pix[i] + hh * a.W + ww)) * 4;
        const bool ok0 = ok && ((unsigned)ww < (unsigned)a.W);
        const bool ok1 = ok && ((unsigned)(ww + 1) < (unsigned)a.W);
        bf16x4 lo = {(bf16)0.f, (bf16)0.f, (bf16)0.f, (bf16)0.f}, hi = lo;
        if (ok0) lo = *reinterpret_cast<const bf16x4*>(ptr);
        if (ok1) hi = *reinterpret_cast<const bf16x4*>(ptr + 4);
        bf16x8 v;
#pragma unroll
        for (int e = 0; e < 4; ++e) { v[e] = lo[e]; v[4 + e] = hi[e]; }
        ra[i] = v;
      } else {
        ok = ok && ((unsigned)ww < (unsigned)a.W);
        bf16x8 v = zero8();
        if (ok) v = ldg16(a.x + ((size_t)(pix[i] + hh * a.W + ww)) * a.C + kc0);
        ra[i] = v;
      }
    }
#pragma unroll
    for (int j = 0; j < WROWS; ++j) {
      const int n = n0 + lrow + 32 * j;
      bf16x8 v = zero8();
      if (kvalid && n < a.K) v = ldg16(a.w + (size_t)n * a.Ktot + wk);
      rw[j] = v;
    }
    // advance the cursor by BK
    wk += BK;
    if (C4) {
      kr += 2;
    } else {
      kc0 += BK;
      while (kc0 >= a.C) {
        kc0 -= a.C;
        if (++ks == a.S) { ks = 0; ++kr; }
      }
    }
  };

  auto store_tile = [&](int buf) {
    char* Ab = smem + buf * TILE_BYTES;
    char* Wb = Ab + BM * BK * 2;
#pragma unroll
    for (int i = 0; i < AROWS; ++i) {
      const int row = lrow + 32 * i;
      *reinterpret_cast<bf16x8*>(Ab + row * 128 + ((kc ^ (row & 7)) << 4)) = ra[i];
    }
#pragma unroll
    for (int j = 0; j < WROWS; ++j) {
      const int row = lrow + 32 * j;
      *reinterpret_cast<bf16x8*>(Wb + row * 128 + ((kc ^ fw_swz(row)) << 4)) = rw[j];
    }
  };

  // ---- fragment addressing ------------------------------------------------------------------
  const int sl = lane >> 4;
  const int li = lane & 15;
  // weight rows fed to MFMA row i (= li): n = (li>>2)*16 + t*4 + (li&3)
  const int wrow0 = wn * 64 + (li >> 2) * 16 + (li & 3);
  const int wswz = fw_swz(wrow0);
  const int arow0 = wm * (MT * 16) + li;
  const int aswz = li & 7;

  f32x4 acc[4][MT];
#pragma unroll
  for (int t = 0; t < 4; ++t)
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) acc[t][mt] = (f32x4){0.f, 0.f, 0.f, 0.f};

  load_tile();
  store_tile(0);
  __syncthreads();

  for (int kt = 0; kt < a.KT; ++kt) {
    const bool more = (kt + 1) < a.KT;
    if (more) load_tile();
    const char* Ab = smem + (kt & 1) * TILE_BYTES;
    const char* Wb = Ab + BM * BK * 2;
#pragma unroll
    for (int kk = 0; kk < 2; ++kk) {
      bf16x8 wf[4], af[MT];
      const int s = sl + 4 * kk;
#pragma unroll
      for (int t = 0; t < 4; ++t)
        wf[t] = *reinterpret_cast<const bf16x8*>(Wb + (wrow0 + t * 4) * 128 + ((s ^ wswz) << 4));
#pragma unroll
      for (int mt = 0; mt < MT; ++mt)
        af[mt] = *reinterpret_cast<const bf16x8*>(Ab + (arow0 + mt * 16) * 128 + ((s ^ aswz) << 4));
#pragma unroll
      for (int t = 0; t < 4; ++t)
#pragma unroll
        for (int mt = 0; mt < MT; ++mt)
          acc[t][mt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[t], af[mt], acc[t][mt], 0, 0, 0);
    }
    if (more) store_tile((kt + 1) & 1);
    __syncthreads();
  }

  // ---- epilogue: lane (sl, li) holds channels nb..nb+15 of pixels m0 + arow0 + mt*16 ----------
  const int nb = n0 + wn * 64 + sl * 16;
  float bias_v[16];
#pragma unroll
  for (int c = 0; c < 16; ++c) bias_v[c] = (a.bias != nullptr && nb + c < a.K) ? a.bias[nb + c] : 0.f;

  float s1[16], s2[16];
#pragma unroll
  for (int c = 0; c < 16; ++c) { s1[c] = 0.f; s2[c] = 0.f; }

#pragma unroll
  for (int mt = 0; mt < MT; ++mt) {
    const int m = m0 + arow0 + mt * 16;
    if (m < a.M) {
      bf16* yp = a.y + (size_t)m * a.K + nb;
      float v[16];
#pragma unroll
      for (int t = 0; t < 4; ++t)
#pragma unroll
        for (int r = 0; r < 4; ++r) v[t * 4 + r] = acc[t][mt][r] + bias_v[t * 4 + r];
#pragma unroll
      for (int half = 0; half < 2; ++half) {
        if (nb + half * 8 + 8 <= a.K) {
          bf16x8 o;
          if (a.accumulate) {
            const bf16x8 old = ldg16(yp + half * 8);
#pragma unroll
            for (int e = 0; e < 8; ++e) o[e] = f2bf(v[half * 8 + e] + bf2f(old[e]));
          } else {
#pragma unroll
            for (int e = 0; e < 8; ++e) o[e] = f2bf(v[half * 8 + e]);
          }
          stg16(yp + half * 8, o);
#pragma unroll
          for (int e = 0; e < 8; ++e) {
            const float f = bf2f(o[e]);
            s1[half * 8 + e] += f;
            s2[half * 8 + e] += f * f;
          }
        }
      }
    }
  }

  if (a.stats != nullptr) {
    // butterfly over the 16 pixel-lanes that share this lane's channel group
#pragma unroll
    for (int c = 0; c < 16; ++c) {
#pragma unroll
      for (int off = 1; off < 16; off <<= 1) {
        s1[c] += __shfl_xor(s1[c], off, 64);
        s2[c] += __shfl_xor(s2[c], off, 64);
      }
    }
    float* red = reinterpret_cast<float*>(smem);  // [2][WGM][BN]; tiles are dead (barrier above)
    if (li == 0) {
#pragma unroll
      for (int c = 0; c < 16; ++c) {
        red[(0 * WGM + wm) * BN + wn * 64 + sl * 16 + c] = s1[c];
        red[(1 * WGM + wm) * BN + wn * 64 + sl * 16 + c] = s2[c];
      }
    }
    __syncthreads();
    if (tid < 2 * BN) {
      const int which = tid / BN;
      const int c = tid - which * BN;
      float t = 0.f;
#pragma unroll
      for (int w_ = 0; w_ < WGM; ++w_) t += red[(which * WGM + w_) * BN + c];
      if (n0 + c < a.K) a.stats[((size_t)which * a.gridM + bm) * a.K + n0 + c] = t;
    }
  }
}

template <int BM, int BN, int IN_DIV, bool C4>
int launch(const ConvArgs& a, hipStream_t st) {
  constexpr int smem = 2 * (BM + BN) * BK * 2;
  static bool attr_set = false;  // benign race: idempotent
  if (!attr_set) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&conv_igemm_kernel<BM, BN, IN_DIV, C4>),
                        hipFuncAttributeMaxDynamicSharedMemorySize, smem);
    attr_set = true;
  }
  hipLaunchKernelGGL((conv_igemm_kernel<BM, BN, IN_DIV, C4>), dim3(a.gridM * a.gridN), dim3(256),
                     smem, st, a);
  return 0;
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
  return tok_cdiv((long long)d->n * d->p * d->q, 128);
}

extern "C" int tok_conv_fwd(const tok_conv_desc* d, const void* x, const void* w,
                            const float* bias, void* y, float* stats, void* stream) {
  if (int e = check_desc(d, "tok_conv_fwd")) return e;
  TOK_CHECK_ARG(x && w && y, "tok_conv_fwd: null pointer");
  ConvArgs a;
  a.x = (const bf16*)x; a.w = (const bf16*)w; a.y = (bf16*)y; a.bias = bias; a.stats = stats;
  a.H = d->h; a.W = d->w; a.C = d->c; a.K = d->k; a.R = d->r; a.S = d->s_pad;
  a.P = d->p; a.Q = d->q; a.stride = d->stride; a.pad = d->pad;
  a.M = d->n * d->p * d->q; a.PQ = d->p * d->q;
  a.Ktot = d->r * d->s_pad * d->c; a.KT = tok_cdiv(a.Ktot, BK);
  a.gridM = tok_cdiv(a.M, 128);
  a.accumulate = 0;
  hipStream_t st = tok_stream(stream);
  const bool c4 = d->c == 4;
  if (d->k <= 64) {
    a.gridN = 1;
    if (c4) launch<128, 64, 1, true>(a, st); else launch<128, 64, 1, false>(a, st);
  } else {
    a.gridN = tok_cdiv(d->k, 128);
    if (c4) launch<128, 128, 1, true>(a, st); else launch<128, 128, 1, false>(a, st);
  }
  TOK_CHECK_LAUNCH("tok_conv_fwd");
  return TOK_OK;
}

extern "C" int tok_conv_dgrad(const tok_conv_desc* d, const void* dy, const void* w_dgrad, void* dx,
                              int accumulate, void* stream) {
  if (int e = check_desc(d, "tok_conv_dgrad")) return e;
  TOK_CHECK_ARG(dy && w_dgrad && dx, "tok_conv_dgrad: null pointer");
  TOK_CHECK_ARG(d->c % 8 == 0, "tok_conv_dgrad: c4 (stem) input needs no data gradient");
  TOK_CHECK_ARG(d->stride == 1 || d->stride == 2, "tok_conv_dgrad: stride %d unsupported", d->stride);
  TOK_CHECK_ARG(d->r - 1 - d->pad >= 0, "tok_conv_dgrad: pad > r-1 unsupported");
  ConvArgs a;
  a.x = (const bf16*)dy; a.w = (const bf16*)w_dgrad; a.y = (bf16*)dx; a.bias = nullptr; a.stats = nullptr;
  // gathered tensor = dY (P x Q x K), output = dX (H x W x C)
  a.H = d->p; a.W = d->q; a.C = d->k; a.K = d->c; a.R = d->r; a.S = d->s;
  a.P = d->h; a.Q = d->w; a.stride = 1; a.pad = d->r - 1 - d->pad;
  TOK_CHECK_ARG(d->s - 1 - d->pad == a.pad, "tok_conv_dgrad: square filters only");
  a.M = d->n * d->h * d->w; a.PQ = d->h * d->w;
  a.Ktot = d->r * d->s * d->k; a.KT = tok_cdiv(a.Ktot, BK);
  a.gridM = tok_cdiv(a.M, 128);
  a.accumulate = accumulate;
  hipStream_t st = tok_stream(stream);
  if (d->c <= 64) {
    a.gridN = 1;
    if (d->stride == 1) launch<128, 64, 1, false>(a, st); else launch<128, 64, 2, false>(a, st);
  } else {
    a.gridN = tok_cdiv(d->c, 128);
    if (d->stride == 1) launch<128, 128, 1, false>(a, st); else launch<128, 128, 2, false>(a, st);
  }
  TOK_CHECK_LAUNCH("tok_conv_dgrad");
  return TOK_OK;
}
