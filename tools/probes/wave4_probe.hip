// Four waves with 4 x 4 accumulators each (one wave per SIMD, 512 registers) instead of eight with 2 x 4: the same 256 x 256
// tile and the same LDS-DMA staging of BOTH operands, but a third fewer fragment reads (8 ds_read_b128 per 16 MFMAs instead
// of 6 per 8) and no bytes moved elsewhere -- the variant the direct-operand probe (same bytes, other path: +-0) did not test.
// Build + run on the GPU box:
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -I mevi_amd/csrc tools/probes/wave4_probe.hip mevi_amd/csrc/abi.hip -o /tmp/wave4_probe && /tmp/wave4_probe
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdio>
#include <vector>

#include "mfma_pp_f16.h"

using namespace mevi;

constexpr int WU = 512 * 16;   // floats per LDS unit buffer: (256 + 256) rows x 64 B

// Ab, Wb: unit-major images [rows/256 tiles][kp/32 units][256 rows][32 halves]
template <int NB, int ABL, bool CHECK = false>
__global__ __launch_bounds__(256, 1) void probe_wave4_kernel(const _Float16 *__restrict__ Wb, const _Float16 *__restrict__ Ab, int kp,
                                                             float *sink, float *dbg, int n_mtiles, int n_ntiles) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  const int nwg = n_mtiles * n_ntiles;
  const int xcd = blockIdx.x & 7, per_xcd = gridDim.x >> 3;
  const int q8 = nwg >> 3, r8 = nwg & 7;
  const int range_base = (xcd < r8) ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8;
  const int range_len = q8 + (xcd < r8 ? 1 : 0);
  int item = blockIdx.x >> 3;
  const int t = threadIdx.x, lane = t & 63;
  const int w4 = __builtin_amdgcn_readfirstlane(t >> 6);
  const int wm = w4 >> 1, wn = w4 & 1;
  const int lrow = lane & 31, half = lane >> 5;
  const int U = kp / 32;
  const size_t tile_bytes = (size_t)256 * kp * 2;
  struct Tile {
    const char *src;      // waves 0-1 stage the W tile (LDS rows [0, 256)), waves 2-3 the A tile (rows [256, 512)): 128 rows each
    unsigned bytes;
  };
  auto next = [&](Tile &s) -> bool {
    if (item >= range_len) return false;
    int mt, nt;
    supertile_order<4, 8>(range_base + item, n_ntiles, n_mtiles, nt, mt);
    item += per_xcd;
    s.src = w4 < 2 ? reinterpret_cast<const char *>(Wb) + (size_t)nt * tile_bytes : reinterpret_cast<const char *>(Ab) + (size_t)mt * tile_bytes;
    s.bytes = (unsigned)tile_bytes;
    return true;
  };
  Tile cur, nxt;
  if (!next(cur)) return;
  bool have_nxt = next(nxt);
  if (!have_nxt) nxt = cur, nxt.bytes = 0u;
  const int cpiece = (lane & 3) ^ ((lane >> 4) & 3);
  int voff[8];
#pragma unroll
  for (int i = 0; i < 8; ++i) voff[i] = (128 * (w4 & 1) + 16 * i + (lane >> 2)) * 64 + cpiece * 16;
  auto dma = [&](int uu, int buf) {
    if constexpr (ABL & 2) return;
    const bool spill = uu >= U;
    const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<char *>(spill ? nxt.src : cur.src), 0, (int)(spill ? nxt.bytes : cur.bytes), 0x00020000);
    const int soff = (spill ? uu - U : uu) * 16384;
    float *base = lds + buf * WU + (128 * w4) * 16;
#pragma unroll
    for (int i = 0; i < 8; ++i)
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, (__attribute__((address_space(3))) void *)(base + 16 * i * 16), 16, voff[i],
                                               soff, 0, 0);
  };
  const int sw = (lrow >> 2) & 3;
  int cj[2];
#pragma unroll
  for (int j = 0; j < 2; ++j) cj[j] = ((2 * j + half) ^ sw) * 4;
  const int offw = (128 * wm + lrow) * 16, offa = (256 + 128 * wn + lrow) * 16;
  struct Frag {
    f16x8 w[4], a[4];
  };
  auto read = [&](int buf, int j, Frag &f) {
    if constexpr (ABL & 4) return;
    const float *p = lds + buf * WU + cj[j];
#pragma unroll
    for (int i = 0; i < 4; ++i) f.w[i] = *reinterpret_cast<const f16x8 *>(p + offw + 32 * i * 16);
#pragma unroll
    for (int i = 0; i < 4; ++i) f.a[i] = *reinterpret_cast<const f16x8 *>(p + offa + 32 * i * 16);
  };
  f32x16 acc[4][4];
  auto mma = [&](const Frag &f) {
    if constexpr (ABL & 8) return;
#pragma unroll
    for (int ni = 0; ni < 4; ++ni)
#pragma unroll
      for (int mi = 0; mi < 4; ++mi) acc[mi][ni] = __builtin_amdgcn_mfma_f32_32x32x16_f16(f.w[mi], f.a[ni], acc[mi][ni], 0, 0, 0);
  };
  auto zero = [&]() {
#pragma unroll
    for (int mi = 0; mi < 4; ++mi)
#pragma unroll
      for (int ni = 0; ni < 4; ++ni)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[mi][ni][r] = 0.f;
  };
  float keep = 0.f;
  bool first_tile = true;
  auto emit = [&]() {
    if (CHECK && first_tile && blockIdx.x == 0) {
#pragma unroll
      for (int mi = 0; mi < 4; ++mi)
#pragma unroll
        for (int ni = 0; ni < 4; ++ni)
#pragma unroll
          for (int r = 0; r < 16; ++r) {
            const int wrow = 128 * wm + 32 * mi + (r & 3) + 8 * (r >> 2) + 4 * half;
            const int arow = 128 * wn + 32 * ni + lrow;
            dbg[wrow * 256 + arow] = acc[mi][ni][r];
          }
    }
    first_tile = false;
#pragma unroll
    for (int mi = 0; mi < 4; ++mi)
#pragma unroll
      for (int ni = 0; ni < 4; ++ni) keep += acc[mi][ni][0] + acc[mi][ni][7];
  };
  Frag F0, F1;
  {
    const f16x8 one = {1, 1, 1, 1, 1, 1, 1, 1};
#pragma unroll
    for (int i = 0; i < 4; ++i) F0.w[i] = F0.a[i] = F1.w[i] = F1.a[i] = one;
  }
  for (int g = 0; g < NB - 1; ++g) dma(g, g);
  asm volatile("s_waitcnt vmcnt(0)\n\ts_barrier" ::: "memory");
  zero();
  int buf = 0;
  auto ring = [](int b) { return b >= NB ? b - NB : b; };
  bool pending = false;
  while (true) {
    for (int u = 0; u < U; ++u) {
      // unit u of this tile is in LDS (waited for at the end of the previous unit); buffer of unit u - 1 is free
      dma(u + NB - 1, ring(buf + NB - 1));
      read(buf, 0, F0);
      if (pending) mma(F1);
      if (u == 0 && pending) {
        emit();
        zero();
      }
      __builtin_amdgcn_sched_barrier(0);
      read(buf, 1, F1);
      mma(F0);
      __builtin_amdgcn_sched_barrier(0);
      // unit u + 1 landed when only the NB - 2 youngest units' pieces (8 each) are outstanding; own reads of u done
      constexpr int W = 8 * (NB - 2);
      if constexpr (ABL & 1) asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)" ::"n"(W) : "memory");
      else asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)\n\ts_barrier" ::"n"(W) : "memory");
      __builtin_amdgcn_sched_barrier(0);
      pending = true;
      buf = ring(buf + 1);
    }
    if (!have_nxt) break;
    cur = nxt;
    have_nxt = next(nxt);
    if (!have_nxt) nxt.bytes = 0u;
  }
  mma(F1);
  emit();
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  if (keep == 12345.678f) sink[t] = keep;
}

template <int NB, int ABL, bool CHECK = false>
float run(const _Float16 *Wb, const _Float16 *Ab, int M, int N, int kp, float *sink, float *dbg) {
  const int n_mtiles = M / 256, n_ntiles = N / 256;
  const size_t lds_b = (size_t)NB * WU * sizeof(float);
  (void)hipFuncSetAttribute(reinterpret_cast<const void *>(probe_wave4_kernel<NB, ABL, CHECK>), hipFuncAttributeMaxDynamicSharedMemorySize,
                            (int)lds_b);
  hipEvent_t e0, e1;
  (void)hipEventCreate(&e0);
  (void)hipEventCreate(&e1);
  float best = 1e30f;
  for (int rep = 0; rep < 4; ++rep) {
    (void)hipEventRecord(e0);
    hipLaunchKernelGGL((probe_wave4_kernel<NB, ABL, CHECK>), dim3(256), dim3(256), lds_b, 0, Wb, Ab, kp, sink, dbg, n_mtiles, n_ntiles);
    (void)hipEventRecord(e1);
    (void)hipEventSynchronize(e1);
    float ms;
    (void)hipEventElapsedTime(&ms, e0, e1);
    if (rep && ms < best) best = ms;
  }
  if (hipGetLastError() != hipSuccess) printf("launch error\n");
  return best;
}

int main() {
  const int M = 76800, N = 2304, kp = 2304;   // the problem of stream_probe.hip
  const int U = kp / 32;
  std::vector<_Float16> a((size_t)M * kp), w((size_t)N * kp);
  unsigned s = 1u;
  for (auto &v : a) {
    s = s * 1664525u + 1013904223u;
    v = (_Float16)(((int)(s >> 16) % 2001 - 1000) * 0.001f);
  }
  for (auto &v : w) {
    s = s * 1664525u + 1013904223u;
    v = (_Float16)(((int)(s >> 16) % 2001 - 1000) * 0.001f);
  }
  std::vector<_Float16> ab(a.size()), wb(w.size());
  for (int r = 0; r < M; ++r)
    for (int k = 0; k < kp; ++k) ab[((size_t)(r / 256) * U + k / 32) * 8192 + (size_t)(r % 256) * 32 + k % 32] = a[(size_t)r * kp + k];
  for (int r = 0; r < N; ++r)
    for (int k = 0; k < kp; ++k) wb[((size_t)(r / 256) * U + k / 32) * 8192 + (size_t)(r % 256) * 32 + k % 32] = w[(size_t)r * kp + k];
  _Float16 *Ab, *Wb;
  float *sink, *dbg;
  (void)hipMalloc(&Ab, ab.size() * 2);
  (void)hipMalloc(&Wb, wb.size() * 2);
  (void)hipMalloc(&sink, 4096);
  (void)hipMalloc(&dbg, 256 * 256 * 4);
  (void)hipMemcpy(Ab, ab.data(), ab.size() * 2, hipMemcpyHostToDevice);
  (void)hipMemcpy(Wb, wb.data(), wb.size() * 2, hipMemcpyHostToDevice);
  (void)hipMemset(dbg, 0, 256 * 256 * 4);
  run<4, 0, true>(Wb, Ab, M, N, kp, sink, dbg);
  std::vector<float> got(256 * 256);
  (void)hipMemcpy(got.data(), dbg, got.size() * 4, hipMemcpyDeviceToHost);
  double worst = 0;
  for (int i = 0; i < 256; i += 7)
    for (int j = 0; j < 256; j += 5) {
      double ref = 0;
      for (int k = 0; k < kp; ++k) ref += (double)(float)w[(size_t)i * kp + k] * (double)(float)a[(size_t)j * kp + k];
      worst = std::fmax(worst, std::fabs(ref - got[i * 256 + j]));
    }
  printf("first tile against float64: max |diff| %.3g (values ~ %.3g)\n", worst, std::fabs((double)got[3 * 256 + 5]));
  const double flop = 2.0 * M * N * kp;
  struct { const char *name; float ms; } r[] = {
      {"4 waves x (4 x 4), 4 unit buffers", run<4, 0>(Wb, Ab, M, N, kp, sink, nullptr)},
      {"4 waves x (4 x 4), 5 unit buffers", run<5, 0>(Wb, Ab, M, N, kp, sink, nullptr)},
      {"  no barrier", run<4, 1>(Wb, Ab, M, N, kp, sink, nullptr)},
      {"  no DMA", run<4, 2>(Wb, Ab, M, N, kp, sink, nullptr)},
      {"  no LDS reads", run<4, 4>(Wb, Ab, M, N, kp, sink, nullptr)},
      {"  no DMA, no LDS reads (MFMA + barriers)", run<4, 6>(Wb, Ab, M, N, kp, sink, nullptr)},
      {"  no MFMA", run<4, 8>(Wb, Ab, M, N, kp, sink, nullptr)}};
  for (auto &x : r) printf("%-48s %8.3f ms  %7.1f TFLOP/s (f16)\n", x.name, x.ms, flop / x.ms / 1e9);
  return 0;
}
