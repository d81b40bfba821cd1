// Micro-benchmark: does the STORE PATTERN of a GEMM epilogue cost HBM write bandwidth?  Writes an [M x K] bf16 tensor (no loads, no
// MFMA) tile by tile, 128 pixels x 64 channels per tile like pw_gemm_ring_kernel<64>, with three ways of laying 16-byte lane stores:
//   gemm : a wave instruction = 16 pixels x 64 B (lane&15 = pixel, lane>>4 = 16-byte group), two instructions ("halves") per 128-B row —
//          the MFMA D-layout epilogue of the conv / GEMM kernels
//   line : a wave instruction = 8 pixels x 128 B (lane&7 = chunk, lane>>3 = pixel): whole cache lines per instruction (what an epilogue
//          staged through LDS would issue)
//   flat : a wave instruction = 1 KB contiguous (tile = 16 pixels x 256 channels... any elementwise kernel)
//   hipcc --offload-arch=gfx950 -O3 -o store_rate store_rate.hip && ./store_rate
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>

typedef __attribute__((ext_vector_type(4))) uint32_t u32x4;

template <int PAT>
__global__ __launch_bounds__(256) void store_kernel(char* y, long long M, int K, int tilesN, long long tiles) {
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const u32x4 v = {(uint32_t)tid, 1u, 2u, 3u};
  for (long long t = blockIdx.x; t < tiles; t += gridDim.x) {
    const long long mt = t / tilesN;
    const int nt = (int)(t - mt * tilesN);
    const long long m0 = mt * 128;
    char* base = y + (m0 * K + (long long)nt * 64) * 2;
    const long long pitch = (long long)K * 2;
    if (PAT == 0) {
      // wave owns pixels 32*wave .. +31: 2 blocks of 16 pixels x 2 halves
#pragma unroll
      for (int mb = 0; mb < 2; ++mb)
#pragma unroll
        for (int half = 0; half < 2; ++half) {
          const int pix = 32 * wave + 16 * mb + (lane & 15);
          if (m0 + pix < M) *reinterpret_cast<u32x4*>(base + pix * pitch + half * 64 + (lane >> 4) * 16) = v;
        }
    } else if (PAT == 1) {
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const int pix = 32 * wave + 8 * i + (lane >> 3);
        if (m0 + pix < M) *reinterpret_cast<u32x4*>(base + pix * pitch + (lane & 7) * 16) = v;
      }
    } else {
      // flat: the same bytes per tile (16 KB) as one contiguous run: tile t covers bytes [t * 16 KB, +16 KB)
      char* fb = y + t * 16384ll;
#pragma unroll
      for (int i = 0; i < 4; ++i) *reinterpret_cast<u32x4*>(fb + (i * 256 + tid) * 16) = v;
    }
  }
}

int main() {
  const long long Ms[] = {802816, 802816, 200704, 200704};
  const int Ks[] = {256, 64, 512, 128};
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  for (int c = 0; c < 4; ++c) {
    const long long M = Ms[c];
    const int K = Ks[c];
    const size_t bytes = (size_t)M * K * 2;
    char* y;
    hipMalloc(&y, bytes + 65536);
    const int tilesN = K / 64;
    const long long tiles = (M + 127) / 128 * tilesN;
    for (int grid : {768, 1536, 3072}) {
      for (int pat = 0; pat < 3; ++pat) {
        auto launch = [&]() {
          if (pat == 0) hipLaunchKernelGGL(store_kernel<0>, dim3(grid), dim3(256), 0, 0, y, M, K, tilesN, tiles);
          else if (pat == 1) hipLaunchKernelGGL(store_kernel<1>, dim3(grid), dim3(256), 0, 0, y, M, K, tilesN, tiles);
          else hipLaunchKernelGGL(store_kernel<2>, dim3(grid), dim3(256), 0, 0, y, M, K, tilesN, tiles);
        };
        for (int i = 0; i < 3; ++i) launch();
        hipEventRecord(e0, 0);
        const int N = 20;
        for (int i = 0; i < N; ++i) launch();
        hipEventRecord(e1, 0);
        hipEventSynchronize(e1);
        float ms;
        hipEventElapsedTime(&ms, e0, e1);
        const double us = ms * 1e3 / N;
        printf("M=%lld K=%d (%.0f MB) grid %4d pattern %s: %7.1f us  %.2f TB/s\n", M, K, bytes / 1e6, grid,
               pat == 0 ? "gemm" : pat == 1 ? "line" : "flat", us, bytes / us / 1e6);
      }
    }
    hipFree(y);
  }
  return 0;
}
