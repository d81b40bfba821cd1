// Micro-benchmark: how fast can a CU fill LDS with `buffer_load_dwordx4 ... lds` (the staging path of every GEMM / conv kernel
// here) from an L2-resident source?  Persistent workgroups, no MFMA, a counted wait keeps DEPTH stages in flight.
//   hipcc --offload-arch=gfx950 -O3 -o dma_rate dma_rate.hip && ./dma_rate
// Prints bytes / clock / CU for: workgroups per CU (1, 2, 3), loads per thread and stage (4, 8, 12), source footprint
// (1 MB: L2; 64 MB: Infinity Cache; 1 GB: HBM), row length of the gathered tile (128 B like the igemm tiles, 1 KB contiguous).
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
#include <stdlib.h>

template <int L>
__global__ __launch_bounds__(256) void dma_kernel(const char* src, uint32_t bytes, int iters, int row_stride, unsigned* sink) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  typedef __attribute__((address_space(3))) void lds_void;
  const int tid = threadIdx.x;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const __amdgpu_buffer_rsrc_t srd = __builtin_amdgcn_make_buffer_rsrc((void*)src, 0, bytes, 0x00020000);
  // a "tile" of L passes x 32 rows x 128 B: thread = (row tid>>3, 16-byte chunk tid&7) of each pass
  const uint32_t lane_off = (uint32_t)((tid >> 3) * row_stride + (tid & 7) * 16);
  uint32_t tile = (uint32_t)blockIdx.x * 7919u;
  auto issue = [&](int slot) {
    const uint32_t base = (uint32_t)(((uint64_t)tile * 4096u * L) % (bytes - (uint32_t)(32 * L) * row_stride - 4096));
#pragma unroll
    for (int i = 0; i < L; ++i) {
      uint32_t off = (base & ~127u) + lane_off + (uint32_t)(i * 32) * row_stride;
      asm volatile("" : "+v"(off));
      __builtin_amdgcn_raw_ptr_buffer_load_lds(srd, (lds_void*)(smem + slot * (L * 4096) + i * 4096 + wave * 1024), 16, off, 0, 0, 0);
    }
    tile += gridDim.x;
  };
  issue(0);
  issue(1);
  int nxt = 2;
  for (int s = 0; s < iters; ++s) {
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(L) : "memory");
    __builtin_amdgcn_s_barrier();
    issue(nxt);
    nxt = nxt == 2 ? 0 : nxt + 1;
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  if (tid == 0 && smem[17] == 123) sink[0] = 1;
}

template <int L>
void run(const char* src, size_t bytes, int per_cu, int row_stride, unsigned* sink, const char* what) {
  const int smem = 3 * L * 4096;
  hipFuncSetAttribute((const void*)&dma_kernel<L>, hipFuncAttributeMaxDynamicSharedMemorySize, smem);
  const int grid = 256 * per_cu, iters = 2000;
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  for (int rep = 0; rep < 2; ++rep) {
    hipEventRecord(e0);
    hipLaunchKernelGGL(dma_kernel<L>, dim3(grid), dim3(256), smem, 0, src, (uint32_t)bytes, iters, row_stride, sink);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
  }
  float ms;
  hipEventElapsedTime(&ms, e0, e1);
  const double total = (double)grid * iters * L * 4096.0;
  printf("%-26s L=%2d wg/cu=%d row_stride=%5d : %7.2f TB/s  = %5.1f B/clk/CU (2.4 GHz)\n", what, L, per_cu, row_stride, total / ms / 1e9,
         total / (ms * 1e-3) / 256.0 / 2.4e9);
}

int main() {
  char* buf;
  const size_t big = 1ull << 30;
  hipMalloc(&buf, big);
  hipMemset(buf, 1, big);
  unsigned* sink;
  hipMalloc(&sink, 4);
  const size_t sizes[3] = {1ull << 20, 64ull << 20, 1ull << 30};
  const char* names[3] = {"1 MB (L2)", "64 MB (Infinity Cache)", "1 GB (HBM)"};
  for (int s = 0; s < 3; ++s)
    for (int rs : {128, 256, 1024})
      for (int wg = 1; wg <= 3; ++wg) {
        run<4>(buf, sizes[s], wg, rs, sink, names[s]);
        run<8>(buf, sizes[s], wg, rs, sink, names[s]);
        if (wg <= 2) run<12>(buf, sizes[s], wg, rs, sink, names[s]);
      }
  return 0;
}
