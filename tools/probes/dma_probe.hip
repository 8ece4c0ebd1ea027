// LDS-DMA (buffer_load_dwordx4 ... lds) throughput per CU as a function of the contiguous segment a piece
// reads per row: SEG bytes per row -> 1024/SEG rows per wave-instruction.  Table small enough for one XCD's L2.
//   hipcc --offload-arch=gfx950 -O3 -o dma_probe dma_probe.hip && ./dma_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

template <int SEG, int WAVES>
__global__ __launch_bounds__(WAVES * 64) void probe(const char *tab, int rows, int row_bytes, int iters, int span) {
  extern __shared__ float lds[];
  const int lane = threadIdx.x & 63;
  const int w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  constexpr int LPR = SEG / 16;        // lanes per row
  constexpr int RPI = 64 / LPR;        // rows per instruction
  // each workgroup walks `span` consecutive rows starting at a block-dependent row; wave w takes rows w*RPI.. of every step
  const int row0 = (blockIdx.x * 37) % (rows - span);
  const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc((void *)tab, 0, rows * row_bytes, 0x00020000);
  int voff = (row0 + w * RPI + lane / LPR) * row_bytes + (lane % LPR) * 16;
  float *base = lds + w * 4 * 256;  // 4 KiB per wave: 4 pieces in rotation
  int koff = 0, r = 0;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int p = 0; p < 4; ++p) {
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, (__attribute__((address_space(3))) void *)(base + p * 256), 16,
                                               voff + r * row_bytes, koff, 0, 0);
      r += WAVES * RPI;
      if (r + WAVES * RPI > span) { r = 0; koff += SEG; if (koff + SEG > row_bytes) koff = 0; }
    }
    asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  if (lds[threadIdx.x] == 123.456f) printf("x");
}

template <int SEG, int WAVES>
void run(const char *tab, int rows, int row_bytes, int blocks_per_cu) {
  int iters = 4000, span = 512;
  hipEvent_t a, b;
  hipEventCreate(&a); hipEventCreate(&b);
  const int grid = 256 * blocks_per_cu;
  probe<SEG, WAVES><<<grid, WAVES * 64, WAVES * 4096>>>(tab, rows, row_bytes, 10, span);
  hipEventRecord(a);
  probe<SEG, WAVES><<<grid, WAVES * 64, WAVES * 4096>>>(tab, rows, row_bytes, iters, span);
  hipEventRecord(b);
  hipEventSynchronize(b);
  float ms; hipEventElapsedTime(&ms, a, b);
  double bytes = (double)grid * WAVES * iters * 4 * 1024;
  printf("SEG %4d B  waves/wg %d  wg/CU %d : %.2f TB/s  (%.1f GB/s per CU)\n", SEG, WAVES, blocks_per_cu,
         bytes / ms / 1e9, bytes / ms / 1e6 / 256);
}

int main() {
  const int rows = 4096, row_bytes = 1536;   // 6 MiB table; a workgroup touches 512 rows
  char *tab; hipMalloc(&tab, (size_t)rows * row_bytes); hipMemset(tab, 1, (size_t)rows * row_bytes);
  run<32, 8>(tab, rows, row_bytes, 1);
  run<64, 8>(tab, rows, row_bytes, 1);
  run<128, 8>(tab, rows, row_bytes, 1);
  run<256, 8>(tab, rows, row_bytes, 1);
  run<512, 8>(tab, rows, row_bytes, 1);
  run<1024, 8>(tab, rows, row_bytes, 1);
  run<64, 4>(tab, rows, row_bytes, 2);
  run<128, 4>(tab, rows, row_bytes, 2);
  run<64, 4>(tab, rows, row_bytes, 1);
  run<64, 8>(tab, rows, row_bytes, 2);
  run<128, 8>(tab, rows, row_bytes, 2);
  return 0;
}
