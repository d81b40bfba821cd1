// Micro-benchmark: HBM write bandwidth of the conv epilogue's store pattern as a function of
// (a) lanes per row of one store instruction, (b) resident workgroups per CU (LDS-limited, like the
// conv kernels), (c) pixel stride between consecutive tile rows (2 = stride-2 dgrad parity classes).
//   hipcc --offload-arch=gfx950 -O3 -o store_pattern store_pattern.hip && ./store_pattern
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>

typedef __attribute__((ext_vector_type(4))) uint32_t u32x4;

template <int LPR, int TW, int NT>
__global__ __launch_bounds__(NT) void store_kernel(char* out, int RS, int ntiles, int tiles_per_row, int pstride,
                                                   int spin) {
  extern __shared__ char lds[];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  constexpr int CPR = TW / 16, BPR = CPR / LPR, RPB = 64 / LPR, NBLK = 128 * CPR / 64, NW = NT / 64;
  float x = tid;
  for (int t = blockIdx.x; t < ntiles; t += gridDim.x) {
    const int tm = t / tiles_per_row, tn = t % tiles_per_row;
    // rows of a tile are pstride pixels apart; the pstride interleaved tiles cover the gaps
    const int grp = tm / pstride, ph = tm % pstride;
    char* base = out + ((size_t)grp * 128 * pstride + ph) * RS + (size_t)tn * TW;
    for (int i = 0; i < spin; ++i) x = x * 1.0001f + 0.5f;     // stand-in for the tile's compute
    u32x4 v = {(uint32_t)t, (uint32_t)x, 1u, 2u};
#pragma unroll
    for (int b = wave; b < NBLK; b += NW) {
      const int brow = b / BPR, bcol = b % BPR;
      const int row = brow * RPB + lane / LPR;
      const int col = bcol * LPR + lane % LPR;
      *reinterpret_cast<u32x4*>(base + (size_t)row * pstride * RS + col * 16) = v;
    }
  }
  if (x == 12345.f) lds[0] = 1;
}

template <int LPR, int TW, int NT>
float run(char* buf, int M, int RS, int grid, int lds_bytes, int pstride, int spin) {
  const int tiles_per_row = RS / TW;
  const int ntiles = (M / 128) * tiles_per_row;
  hipFuncSetAttribute((const void*)&store_kernel<LPR, TW, NT>, hipFuncAttributeMaxDynamicSharedMemorySize, lds_bytes);
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  for (int i = 0; i < 2; ++i)
    hipLaunchKernelGGL((store_kernel<LPR, TW, NT>), dim3(grid), dim3(NT), lds_bytes, 0, buf, RS, ntiles, tiles_per_row, pstride, spin);
  hipEventRecord(e0);
  for (int i = 0; i < 5; ++i)
    hipLaunchKernelGGL((store_kernel<LPR, TW, NT>), dim3(grid), dim3(NT), lds_bytes, 0, buf, RS, ntiles, tiles_per_row, pstride, spin);
  hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  return ms / 5;
}

int main() {
  const int M = 802816, RS = 512;
  char* buf; hipMalloc(&buf, (size_t)M * RS);
  const double gb = (double)M * RS / 1e9;
  printf("M=%d rows of %d B; tile 128 rows x 256 B, 4 lanes (64 B) per row per instruction\n", M, RS);
  for (int pstride : {1, 2}) {
    for (int spin : {0, 2000}) {
      printf("pixel stride %d, spin %4d:", pstride, spin);
      // LDS bytes per WG chosen to pin residency: 160 KB / x
      printf("  8WG/CU(256thr) %5.0f", gb / run<4, 256, 256>(buf, M, RS, 2048, 16 << 10, pstride, spin) * 1e3);
      printf("  3WG/CU(256thr) %5.0f", gb / run<4, 256, 256>(buf, M, RS, 768, 50 << 10, pstride, spin) * 1e3);
      printf("  2WG/CU(512thr) %5.0f", gb / run<4, 256, 512>(buf, M, RS, 512, 66 << 10, pstride, spin) * 1e3);
      printf("  1WG/CU(512thr) %5.0f", gb / run<4, 256, 512>(buf, M, RS, 256, 100 << 10, pstride, spin) * 1e3);
      printf("  GB/s\n");
    }
  }
  return 0;
}
