// Where does the time of the persistent f16 tile stream (mfma_pp_f16.h) go?  The GEMM-shaped loop of gemm_split.hip
// without an epilogue, timed with parts of the loop removed (ABL bits: 1 no s_barrier, 2 no DMA, 4 no LDS reads,
// 8 no MFMA).  Build + run on the GPU box:
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -I mevi_amd/csrc tools/probes/stream_probe.hip mevi_amd/csrc/abi.hip -o /tmp/stream_probe && /tmp/stream_probe
#include <hip/hip_runtime.h>

#include <cstdio>
#include <vector>

#include "mfma_split_stream.h"

using namespace mevi;

struct BlockedUnits {   // unit-major image: a 256-row tile's unit u is one contiguous 16 KiB block (rows 64 B apart)
  __device__ __forceinline__ int operator()(int u) const { return u * 16384; }
};

template <int ABL, int NBUF = 4, bool BLOCKED = false>
__global__ __launch_bounds__(PP_THREADS, 2) void probe_kernel(const _Float16 *A, int M, const _Float16 *W, int N, int kp,
                                                              float *sink, int n_mtiles, int n_ntiles) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  const int nwg = n_mtiles * n_ntiles;
  const int xcd = blockIdx.x & 7, per_xcd = gridDim.x >> 3;
  const int q8 = nwg >> 3, r8 = nwg & 7;
  const int range_base = (xcd < r8) ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8;
  const int range_len = q8 + (xcd < r8 ? 1 : 0);
  int item = blockIdx.x >> 3;
  const int w8 = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int row_bytes = kp * 2;
  float keep = 0.f;
  auto next = [&](H1Src &s) -> bool {
    if (item >= range_len) return false;
    int mt, nt;
    supertile_order<4, 8>(range_base + item, n_ntiles, n_mtiles, nt, mt);
    item += per_xcd;
    if (w8 < 4) {
      s.src = reinterpret_cast<const char *>(W) + (size_t)nt * 256 * (size_t)row_bytes;
      s.bytes = (unsigned)(256 * row_bytes);
    } else {
      s.src = reinterpret_cast<const char *>(A) + (size_t)mt * 256 * (size_t)row_bytes;
      s.bytes = (unsigned)(256 * row_bytes);
    }
    return true;
  };
  auto begin = [&]() {};
  auto emit = [&](f32x16 (&acc)[2][4]) {
#pragma unroll
    for (int mi = 0; mi < 2; ++mi)
#pragma unroll
      for (int ni = 0; ni < 4; ++ni) keep += acc[mi][ni][0] + acc[mi][ni][7];
  };
  if constexpr (BLOCKED)
    h1_tile_stream<decltype(next), decltype(begin), decltype(emit), BlockedUnits, ABL, NBUF>(64, kp / 32, lds, next, begin, emit);
  else
    h1_tile_stream<decltype(next), decltype(begin), decltype(emit), H1PlainUnits, ABL, NBUF>(row_bytes, kp / 32, lds, next, begin, emit);
  if (keep == 12345.678f) sink[threadIdx.x] = keep;
}

// the split GEMM's stream (mfma_split_stream.h): image rows [hi | lo] of kp halves each
template <int ABL, bool BLOCKED = false>
__global__ __launch_bounds__(PP_THREADS, 2) void probe_split_kernel(const _Float16 *A, int M, const _Float16 *W, int N, int kp,
                                                                    float *sink, int n_mtiles, int n_ntiles) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  const int nwg = n_mtiles * n_ntiles;
  const int xcd = blockIdx.x & 7, per_xcd = gridDim.x >> 3;
  const int q8 = nwg >> 3, r8 = nwg & 7;
  const int range_base = (xcd < r8) ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8;
  const int range_len = q8 + (xcd < r8 ? 1 : 0);
  int item = blockIdx.x >> 3;
  const int w8 = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int row_bytes = kp * 4;
  float keep = 0.f;
  auto next = [&](H1Src &s) -> bool {
    if (item >= range_len) return false;
    int mt, nt;
    supertile_order<4, 8>(range_base + item, n_ntiles, n_mtiles, nt, mt);
    item += per_xcd;
    s.src = w8 < 4 ? reinterpret_cast<const char *>(W) + (size_t)nt * 256 * (size_t)row_bytes
                   : reinterpret_cast<const char *>(A) + (size_t)mt * 256 * (size_t)row_bytes;
    s.bytes = (unsigned)(256 * row_bytes);
    return true;
  };
  auto begin = [&]() {};
  auto emit = [&](f32x16 (&acc)[2][4]) {
#pragma unroll
    for (int mi = 0; mi < 2; ++mi)
#pragma unroll
      for (int ni = 0; ni < 4; ++ni) keep += acc[mi][ni][0] + acc[mi][ni][7];
  };
  if constexpr (BLOCKED)   // unit-major images: [tile][unit][hi | lo][256 rows][32 halves] -- every DMA piece 1 KiB contiguous
    split_tile_stream<decltype(next), decltype(begin), decltype(emit), ABL>(64, 16384, kp / 32, lds, next, begin, emit, 32768);
  else
    split_tile_stream<decltype(next), decltype(begin), decltype(emit), ABL>(row_bytes, kp * 2, kp / 32, lds, next, begin, emit);
  if (keep == 12345.678f) sink[threadIdx.x] = keep;
}

template <int ABL, bool BLOCKED = false>
float run_split(const _Float16 *A, int M, const _Float16 *W, int N, int kp, float *sink) {
  const int n_mtiles = M / 256, n_ntiles = N / 256;
  hipFuncSetAttribute(reinterpret_cast<const void *>(probe_split_kernel<ABL, BLOCKED>), hipFuncAttributeMaxDynamicSharedMemorySize,
                      (int)ss_lds_bytes());
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  float best = 1e30f;
  for (int rep = 0; rep < 4; ++rep) {
    hipEventRecord(e0);
    hipLaunchKernelGGL((probe_split_kernel<ABL, BLOCKED>), dim3(256), dim3(PP_THREADS), ss_lds_bytes(), 0, A, M, W, N, kp, sink, n_mtiles, n_ntiles);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    if (rep && ms < best) best = ms;
  }
  return best;
}

template <int ABL, int NBUF = 4, bool BLOCKED = false>
float run(const _Float16 *A, int M, const _Float16 *W, int N, int kp, float *sink) {
  const int n_mtiles = M / 256, n_ntiles = N / 256;
  const size_t lds_b = (size_t)NBUF * H1_UNIT * sizeof(float);
  hipFuncSetAttribute(reinterpret_cast<const void *>(probe_kernel<ABL, NBUF, BLOCKED>), hipFuncAttributeMaxDynamicSharedMemorySize,
                      (int)lds_b);
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  float best = 1e30f;
  for (int rep = 0; rep < 4; ++rep) {
    hipEventRecord(e0);
    hipLaunchKernelGGL((probe_kernel<ABL, NBUF, BLOCKED>), dim3(256), dim3(PP_THREADS), lds_b, 0, A, M, W, N, kp, sink, n_mtiles, n_ntiles);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    if (rep && ms < best) best = ms;
  }
  return best;
}

int main() {
  const int M = 76800, N = 2304, kp = 2304;   // the q|k|v GEMM of a tower pass, K' = 3 x 768 (one plain row per operand)
  std::vector<_Float16> h((size_t)M * kp);
  unsigned s = 1u;
  for (auto &v : h) {
    s = s * 1664525u + 1013904223u;
    v = (_Float16)(((int)(s >> 16) % 2001 - 1000) * 0.01f);
  }
  _Float16 *A, *W;
  float *sink;
  hipMalloc(&A, h.size() * 2);
  hipMalloc(&W, (size_t)N * kp * 2);
  hipMalloc(&sink, 4096);
  hipMemcpy(A, h.data(), h.size() * 2, hipMemcpyHostToDevice);
  hipMemcpy(W, h.data(), (size_t)N * kp * 2, hipMemcpyHostToDevice);
  const double flop = 2.0 * M * N * kp;
  struct { const char *name; float ms; } r[] = {
      {"full loop", run<0>(A, M, W, N, kp, sink)},      {"no s_barrier", run<1>(A, M, W, N, kp, sink)},
      {"no DMA", run<2>(A, M, W, N, kp, sink)},          {"no LDS reads", run<4>(A, M, W, N, kp, sink)},
      {"no DMA, no LDS reads", run<6>(A, M, W, N, kp, sink)}, {"no DMA, no LDS reads, no barrier (MFMA only)", run<7>(A, M, W, N, kp, sink)},
      {"no MFMA (data movement + barriers only)", run<8>(A, M, W, N, kp, sink)}, {"no MFMA, no barrier", run<9>(A, M, W, N, kp, sink)},
      {"no MFMA, no LDS reads (DMA + barriers)", run<12>(A, M, W, N, kp, sink)}};
  for (auto &x : r) printf("%-48s %8.3f ms  %7.1f TFLOP/s (f16)\n", x.name, x.ms, flop / x.ms / 1e9);
  printf("five unit buffers (160 KiB, four units in flight): full loop %8.3f ms, DMA + barriers only %8.3f ms\n",
         run<0, 5>(A, M, W, N, kp, sink), run<12, 5>(A, M, W, N, kp, sink));
  printf("unit-major (blocked) images, every DMA piece 1 KiB contiguous: full loop %8.3f ms, DMA + barriers only %8.3f ms\n",
         run<0, 4, true>(A, M, W, N, kp, sink), run<12, 4, true>(A, M, W, N, kp, sink));
  // the same problem through the split stream: rows [hi | lo] of 768 halves each (the buffers above hold 2304 halves per
  // row, of which 1536 are read), 3 x 768 / 16 MFMA k-steps per tile as before
  printf("split stream (each operand slab fetched once per 32 k):\n");
  const int kps = 768;
  struct { const char *name; float ms; } q[] = {
      {"full loop", run_split<0>(A, M, W, N, kps, sink)},        {"no s_barrier", run_split<1>(A, M, W, N, kps, sink)},
      {"no DMA", run_split<2>(A, M, W, N, kps, sink)},            {"no LDS reads", run_split<4>(A, M, W, N, kps, sink)},
      {"no DMA, no LDS reads", run_split<6>(A, M, W, N, kps, sink)}, {"MFMA only", run_split<7>(A, M, W, N, kps, sink)},
      {"no MFMA (data movement + barriers only)", run_split<8>(A, M, W, N, kps, sink)},
      {"no MFMA, no LDS reads (DMA + barriers)", run_split<12>(A, M, W, N, kps, sink)}};
  for (auto &x : q) printf("%-48s %8.3f ms  %7.1f TFLOP/s (f16)\n", x.name, x.ms, flop / x.ms / 1e9);
  printf("split stream on unit-major images: full loop %8.3f ms, DMA + barriers only %8.3f ms\n",
         run_split<0, true>(A, M, W, N, kps, sink), run_split<12, true>(A, M, W, N, kps, sink));
  return 0;
}
