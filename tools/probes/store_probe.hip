// Round 5: what the split GEMM's epilogue pays per store / load INSTRUCTION shape.  One 512-thread workgroup per CU writes (or
// reads) 256 x 256 f32 tiles of a [rows][768] matrix, 16 bytes per lane, with the rows x bytes-per-row footprint of one wave
// instruction varied: 16 rows x 64 B (the MFMA 16x16 C layout: what gemm_split16_kernel's epilogue issues), 8 x 128 B, 4 x 256 B,
// 2 x 512 B, 1 x 1024 B.  Same bytes, same number of instructions; only the number of cache lines an instruction touches changes.
//     hipcc --offload-arch=gfx950 -O3 tools/probes/store_probe.hip -o tools/probes/store_probe
#include <hip/hip_runtime.h>
#include <cstdio>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s -> %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

typedef float f32x4 __attribute__((ext_vector_type(4)));

// SEG = contiguous bytes per row and instruction (64 .. 1024); a wave instruction covers 1024 / SEG rows.
// Tile = 256 rows x 1024 B; wave w of 8 owns rows 32 w .. 32 w + 31 (32 instructions of 1 KiB).
template <int SEG, int MODE>   // MODE 0: store, 1: load (+ a checksum store at the end), 2: load then store (residual epilogue)
__global__ __launch_bounds__(512) void tile_kernel(float *__restrict__ C, const float *__restrict__ R, int n_mtiles, int ldc, float *__restrict__ sink) {
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  constexpr int LPR = SEG / 16;          // lanes per row
  constexpr int RPI = 64 / LPR;          // rows per instruction
  const int r_in = lane / LPR, c_in = (lane % LPR) * 4;   // row within the instruction's footprint, first float
  f32x4 acc = {0.f, 0.f, 0.f, 0.f};
  for (int tile = blockIdx.x; tile < n_mtiles * 3; tile += gridDim.x) {
    const int mt = tile / 3, nt = tile % 3;
    float *base = C + (size_t)(mt * 256 + 32 * w) * ldc + nt * 256;
    const float *rbase = R + (size_t)(mt * 256 + 32 * w) * ldc + nt * 256;
    // the wave's 32 rows x 1024 B as 32 instructions: row group g (RPI rows), column segment s (SEG bytes)
    constexpr int NSEG = 1024 / SEG, NGRP = 32 / RPI;
#pragma unroll 4
    for (int i = 0; i < 32; ++i) {
      const int g = i / NSEG, s = i % NSEG;
      const size_t off = (size_t)(g * RPI + r_in) * ldc + s * (SEG / 4) + c_in;
      (void)NGRP;
      if (MODE == 0) {
        f32x4 v = {(float)i, (float)lane, (float)tile, 1.f};
        *reinterpret_cast<f32x4 *>(base + off) = v;
      } else if (MODE == 1) {
        const f32x4 v = *reinterpret_cast<const f32x4 *>(rbase + off);
        acc += v;
      } else {
        f32x4 v = *reinterpret_cast<const f32x4 *>(rbase + off);
        v += f32x4{1.f, 1.f, 1.f, 1.f};
        *reinterpret_cast<f32x4 *>(base + off) = v;
      }
    }
  }
  if (MODE == 1 && acc[0] + acc[1] + acc[2] + acc[3] == 12345.678f) sink[threadIdx.x] = acc[0];
}

template <int SEG, int MODE>
static int run(float *C, float *R, int n_mtiles, float *sink) {
  hipEvent_t a, b;
  CK(hipEventCreate(&a));
  CK(hipEventCreate(&b));
  float best = 1e30f;
  for (int rep = 0; rep < 6; ++rep) {
    CK(hipEventRecord(a));
    tile_kernel<SEG, MODE><<<256, 512>>>(C, R, n_mtiles, 768, sink);
    CK(hipEventRecord(b));
    CK(hipEventSynchronize(b));
    float ms;
    CK(hipEventElapsedTime(&ms, a, b));
    if (rep && ms < best) best = ms;
  }
  const double bytes = (double)n_mtiles * 256 * 768 * 4 * (MODE == 2 ? 2 : 1);
  printf("%-12s %2d rows x %4d B per instruction: %7.1f us  (%.2f TB/s, %.1f us per round of 256 tiles)\n",
         MODE == 0 ? "store" : (MODE == 1 ? "load" : "load+store"), 1024 / SEG, SEG, best * 1e3, bytes / best / 1e9,
         best * 1e3 / (n_mtiles * 3 / 256.0));
  return 0;
}

int main() {
  const int n_mtiles = 273;   // 69 800 rows: 819 tiles = 3.2 rounds on 256 CUs, as the N = 768 projections of a 6980-query pass
  float *C, *R, *sink;
  CK(hipMalloc(&C, (size_t)n_mtiles * 256 * 768 * 4));
  CK(hipMalloc(&R, (size_t)n_mtiles * 256 * 768 * 4));
  CK(hipMalloc(&sink, 4096));
  CK(hipMemset(R, 0, (size_t)n_mtiles * 256 * 768 * 4));
#define ALL(MODE) \
  if (run<64, MODE>(C, R, n_mtiles, sink) || run<128, MODE>(C, R, n_mtiles, sink) || run<256, MODE>(C, R, n_mtiles, sink) || \
      run<512, MODE>(C, R, n_mtiles, sink) || run<1024, MODE>(C, R, n_mtiles, sink)) return 1;
  ALL(0) ALL(1) ALL(2)
  return 0;
}
