// 16x16x32 vs 32x32x16 f16 MFMA in the filter's persistent tile stream (mfma_pp_f16x16.h vs mfma_pp_f16.h): same tiles, same
// unit-major images, no epilogue; plus a correctness check of the 16x16 stream's C layout against a host reference.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -I mevi_amd/csrc tools/probes/mfma16_probe.hip mevi_amd/csrc/abi.hip -o /tmp/mfma16_probe && /tmp/mfma16_probe
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdio>
#include <vector>

#include "mfma_pp_f16x16.h"

using namespace mevi;

// HOT: every workgroup of an XCD walks the SAME few operand tiles (4 A tiles x 8 W tiles, re-visited forever): the L2-hot
// ceiling of the loop -- what perfect super-tile locality would give
template <int SHAPE, bool STORE, bool HOT = false, int ABL = 0>
__global__ __launch_bounds__(PP_THREADS, 2) void probe_kernel(const _Float16 *A, const _Float16 *W, int kp, float *C, int ldc,
                                                              float *sink, int n_mtiles, int n_ntiles) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  const int nwg = n_mtiles * n_ntiles;
  const int xcd = blockIdx.x & 7, per_xcd = gridDim.x >> 3;
  const int q8 = nwg >> 3, r8 = nwg & 7;
  const int range_base = (xcd < r8) ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8;
  const int range_len = q8 + (xcd < r8 ? 1 : 0);
  int item = blockIdx.x >> 3;
  const int w8 = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int lane = threadIdx.x & 63;
  const int grp = w8 >> 2, wm = (w8 >> 1) & 1, wn = w8 & 1;
  const size_t block_bytes = (size_t)256 * kp * 2;
  float keep = 0.f;
  int hm = 0, hn = 0, tm = 0, tn = 0, np = 0;
  auto next = [&](H1Src &s) -> bool {
    if (item >= range_len) return false;
    int mt, nt;
    supertile_order<4, 8>(range_base + item, n_mtiles, n_ntiles, mt, nt);
    item += per_xcd;
    if constexpr (HOT) mt = (blockIdx.x >> 3) & 3, nt = ((blockIdx.x >> 3) >> 2) & 7;
    if (np == 0) hm = mt, hn = nt; else tm = mt, tn = nt;
    ++np;
    s.src = w8 < 4 ? reinterpret_cast<const char *>(A) + (size_t)mt * block_bytes
                   : reinterpret_cast<const char *>(W) + (size_t)nt * block_bytes;
    s.bytes = (unsigned)block_bytes;
    return true;
  };
  auto begin = [&]() {};
  if constexpr (SHAPE == 16) {
    auto emit = [&](f32x4 (&acc)[4][8]) {
      const int mt = hm, nt = hn;
      hm = tm, hn = tn, --np;
      if constexpr (STORE) {
#pragma unroll
        for (int mi = 0; mi < 4; ++mi)
#pragma unroll
          for (int ni = 0; ni < 8; ++ni)
#pragma unroll
            for (int j = 0; j < 4; ++j) {
              const int row = mt * 256 + 128 * grp + 64 * wm + 16 * mi + 4 * (lane >> 4) + j;   // A (corpus) row
              const int col = nt * 256 + 128 * wn + 16 * ni + (lane & 15);                       // B (query) row
              C[(size_t)row * ldc + col] = acc[mi][ni][j];
            }
      } else {
#pragma unroll
        for (int mi = 0; mi < 4; ++mi)
#pragma unroll
          for (int ni = 0; ni < 8; ++ni) keep += acc[mi][ni][0] + acc[mi][ni][3];
      }
    };
    h16_tile_stream<decltype(next), decltype(begin), decltype(emit), H1BlockedUnits, ABL>(64, kp / 32, lds, next, begin, emit);
  } else {
    auto emit = [&](f32x16 (&acc)[2][4]) {
      hm = tm, hn = tn, --np;
#pragma unroll
      for (int mi = 0; mi < 2; ++mi)
#pragma unroll
        for (int ni = 0; ni < 4; ++ni) keep += acc[mi][ni][0] + acc[mi][ni][7];
    };
    h1_tile_stream<decltype(next), decltype(begin), decltype(emit), H1BlockedUnits>(64, kp / 32, lds, next, begin, emit);
  }
  if (keep == 12345.678f) sink[threadIdx.x] = keep;
}

static size_t image_at(long long r, int k, int dimp) {
  return ((size_t)((r >> 8) * (dimp >> 5) + (k >> 5)) * 256 + (size_t)(r & 255)) * 32 + (k & 31);
}

// sustained rate: `launches` back-to-back launches timed as one batch (the chip's clock under load settles over seconds)
template <int SHAPE>
float run_batch(const _Float16 *A, int M, const _Float16 *W, int N, int kp, float *sink, int launches) {
  hipFuncSetAttribute(reinterpret_cast<const void *>(probe_kernel<SHAPE, false>), hipFuncAttributeMaxDynamicSharedMemorySize,
                      (int)h1_lds_bytes());
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  hipEventRecord(e0);
  for (int i = 0; i < launches; ++i)
    hipLaunchKernelGGL((probe_kernel<SHAPE, false>), dim3(256), dim3(PP_THREADS), h1_lds_bytes(), 0, A, W, kp, nullptr, 0, sink,
                       M / 256, N / 256);
  hipEventRecord(e1);
  hipEventSynchronize(e1);
  float ms;
  hipEventElapsedTime(&ms, e0, e1);
  return ms / launches;
}

template <int ABL>
float run_abl(const _Float16 *A, int M, const _Float16 *W, int N, int kp, float *sink, int launches) {
  hipFuncSetAttribute(reinterpret_cast<const void *>(probe_kernel<16, false, false, ABL>), hipFuncAttributeMaxDynamicSharedMemorySize,
                      (int)h1_lds_bytes());
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  hipEventRecord(e0);
  for (int i = 0; i < launches; ++i)
    hipLaunchKernelGGL((probe_kernel<16, false, false, ABL>), dim3(256), dim3(PP_THREADS), h1_lds_bytes(), 0, A, W, kp, nullptr, 0, sink,
                       M / 256, N / 256);
  hipEventRecord(e1);
  hipEventSynchronize(e1);
  float ms;
  hipEventElapsedTime(&ms, e0, e1);
  return ms / launches;
}

template <int SHAPE>
float run_hot(const _Float16 *A, int M, const _Float16 *W, int N, int kp, float *sink, int launches) {
  hipFuncSetAttribute(reinterpret_cast<const void *>(probe_kernel<SHAPE, false, true>), hipFuncAttributeMaxDynamicSharedMemorySize,
                      (int)h1_lds_bytes());
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  hipEventRecord(e0);
  for (int i = 0; i < launches; ++i)
    hipLaunchKernelGGL((probe_kernel<SHAPE, false, true>), dim3(256), dim3(PP_THREADS), h1_lds_bytes(), 0, A, W, kp, nullptr, 0, sink,
                       M / 256, N / 256);
  hipEventRecord(e1);
  hipEventSynchronize(e1);
  float ms;
  hipEventElapsedTime(&ms, e0, e1);
  return ms / launches;
}

template <int SHAPE>
float run(const _Float16 *A, int M, const _Float16 *W, int N, int kp, float *sink, int reps = 6) {
  hipFuncSetAttribute(reinterpret_cast<const void *>(probe_kernel<SHAPE, false>), hipFuncAttributeMaxDynamicSharedMemorySize,
                      (int)h1_lds_bytes());
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  float best = 1e30f;
  for (int rep = 0; rep < reps; ++rep) {
    hipEventRecord(e0);
    hipLaunchKernelGGL((probe_kernel<SHAPE, false>), dim3(256), dim3(PP_THREADS), h1_lds_bytes(), 0, A, W, kp, nullptr, 0, sink,
                       M / 256, N / 256);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    if (rep && ms < best) best = ms;
  }
  return best;
}

int main() {
  // ---- correctness of the 16x16 stream: 512 x 512 x 768, unit-major images --------------------------------------------
  {
    const int M = 512, N = 512, kp = 768;
    std::vector<float> a((size_t)M * kp), w((size_t)N * kp);
    unsigned s = 7u;
    for (auto &v : a) { s = s * 1664525u + 1013904223u; v = (float)((int)(s >> 16) % 17 - 8); }
    for (auto &v : w) { s = s * 1664525u + 1013904223u; v = (float)((int)(s >> 16) % 13 - 6); }
    std::vector<_Float16> ai((size_t)M * kp), wi((size_t)N * kp);
    for (int r = 0; r < M; ++r) for (int k = 0; k < kp; ++k) ai[image_at(r, k, kp)] = (_Float16)a[(size_t)r * kp + k];
    for (int r = 0; r < N; ++r) for (int k = 0; k < kp; ++k) wi[image_at(r, k, kp)] = (_Float16)w[(size_t)r * kp + k];
    _Float16 *A, *W; float *C, *sink;
    hipMalloc(&A, ai.size() * 2); hipMalloc(&W, wi.size() * 2); hipMalloc(&C, (size_t)M * N * 4); hipMalloc(&sink, 4096);
    hipMemcpy(A, ai.data(), ai.size() * 2, hipMemcpyHostToDevice);
    hipMemcpy(W, wi.data(), wi.size() * 2, hipMemcpyHostToDevice);
    hipMemset(C, 0xff, (size_t)M * N * 4);
    hipFuncSetAttribute(reinterpret_cast<const void *>(probe_kernel<16, true>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)h1_lds_bytes());
    hipLaunchKernelGGL((probe_kernel<16, true>), dim3(8), dim3(PP_THREADS), h1_lds_bytes(), 0, A, W, kp, C, N, sink, M / 256, N / 256);
    std::vector<float> c((size_t)M * N);
    hipMemcpy(c.data(), C, c.size() * 4, hipMemcpyDeviceToHost);
    long long bad = 0;
    for (int r = 0; r < M; ++r)
      for (int q = 0; q < N; ++q) {
        double ref = 0;
        for (int k = 0; k < kp; ++k) ref += (double)a[(size_t)r * kp + k] * w[(size_t)q * kp + k];
        if (c[(size_t)r * N + q] != (float)ref) { if (bad < 5) printf("  mismatch (%d,%d): %g vs %g\n", r, q, c[(size_t)r * N + q], ref); ++bad; }
      }
    printf("16x16x32 stream, 512 x 512 x 768 integers: %lld mismatches (exact expected)\n", bad);
    hipFree(A); hipFree(W); hipFree(C); hipFree(sink);
  }
  // ---- rate: 76800 x 2304 x 768-k tiles (24 units per tile, as the filter at dim 768) and 2304-k ----------------------
  for (int kp : {768, 2304}) {
    const int M = 76800, N = 2304;
    std::vector<_Float16> h((size_t)M * kp);
    unsigned s = 1u;
    for (auto &v : h) { s = s * 1664525u + 1013904223u; v = (_Float16)(((int)(s >> 16) % 2001 - 1000) * 0.01f); }
    _Float16 *A, *W; float *sink;
    hipMalloc(&A, h.size() * 2); hipMalloc(&W, (size_t)N * kp * 2); hipMalloc(&sink, 4096);
    hipMemcpy(A, h.data(), h.size() * 2, hipMemcpyHostToDevice);
    hipMemcpy(W, h.data(), (size_t)N * kp * 2, hipMemcpyHostToDevice);
    const double flop = 2.0 * M * N * kp;
    for (int round = 0; round < 3; ++round) {
      const float m32 = run<32>(A, M, W, N, kp, sink), m16 = run<16>(A, M, W, N, kp, sink);
      printf("k %4d round %d: 32x32x16 %7.3f ms %7.1f TFLOP/s | 16x16x32 %7.3f ms %7.1f TFLOP/s | ratio %.3f\n", kp, round, m32,
             flop / m32 / 1e9, m16, flop / m16 / 1e9, m32 / m16);
    }
    // sustained: ~1.5 s batches, alternating shapes
    const int launches = (int)(1.5 / (flop / 1.1e15)) + 1;   // ~1.5 s at 1.1 PFLOP/s
    for (int round = 0; round < 3; ++round) {
      const float m32 = run_batch<32>(A, M, W, N, kp, sink, launches), m16 = run_batch<16>(A, M, W, N, kp, sink, launches);
      printf("k %4d sustained (%d launches) round %d: 32x32x16 %7.3f ms %7.1f TFLOP/s | 16x16x32 %7.3f ms %7.1f TFLOP/s | ratio %.3f\n",
             kp, launches, round, m32, flop / m32 / 1e9, m16, flop / m16 / 1e9, m32 / m16);
    }
    {
      const float h32 = run_hot<32>(A, M, W, N, kp, sink, launches), h16 = run_hot<16>(A, M, W, N, kp, sink, launches);
      printf("k %4d L2-hot (every workgroup of an XCD on the same 4 x 8 operand tiles): 32x32x16 %7.1f TFLOP/s | 16x16x32 %7.1f TFLOP/s\n",
             kp, flop / h32 / 1e9, flop / h16 / 1e9);
    }
    {
      const int l2 = launches / 3 + 1;
      printf("k %4d 16x16x32 ablations (TFLOP/s equivalents): full %7.1f | no barrier %7.1f | no DMA %7.1f | no LDS reads %7.1f | no DMA + no reads %7.1f | MFMA only %7.1f | no MFMA %7.1f\n",
             kp, flop / run_abl<0>(A, M, W, N, kp, sink, l2) / 1e9, flop / run_abl<1>(A, M, W, N, kp, sink, l2) / 1e9,
             flop / run_abl<2>(A, M, W, N, kp, sink, l2) / 1e9, flop / run_abl<4>(A, M, W, N, kp, sink, l2) / 1e9,
             flop / run_abl<6>(A, M, W, N, kp, sink, l2) / 1e9, flop / run_abl<7>(A, M, W, N, kp, sink, l2) / 1e9,
             flop / run_abl<8>(A, M, W, N, kp, sink, l2) / 1e9);
    }
    hipFree(A); hipFree(W); hipFree(sink);
  }
  return 0;
}
