// Would the f16 tile loop run faster with ONE operand bypassing LDS?
//
// tools/probes/stream_probe.hip shows the persistent tile stream (mfma_pp_f16.h) co-limited by LDS bandwidth: per 32-k unit
// and CU, 32 KiB of LDS-DMA writes + 96 KiB of fragment reads = 128 KiB = 1024 cycles at 128 B/clk -- exactly the 1024
// cycles its 128 MFMAs occupy the four matrix pipes.  This probe keeps the geometry (8 waves, 256 x 256 tile, wave tile
// 64 x 128, acc[2][4]) but fetches the operand a wave needs TWO fragments of per k-step (the filter's documents, the GEMM's
// weights) straight from global memory into registers, from a FRAGMENT-MAJOR image (one wave-load = one 1 KiB contiguous
// fragment: rows 32 b .. 32 b + 31, k 16 s .. 16 s + 15, lane (row, half) = 16 bytes), PD units ahead in a register ring;
// only the four-fragment operand goes through LDS (16 KiB per unit, a ring of NB units).  LDS traffic per unit: 16 + 64 =
// 80 KiB instead of 128.
// Build + run on the GPU box:
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -I mevi_amd/csrc tools/probes/direct_probe.hip mevi_amd/csrc/abi.hip -o /tmp/direct_probe && /tmp/direct_probe
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdio>
#include <type_traits>
#include <vector>

#include "mfma_pp_f16.h"

using namespace mevi;

typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
constexpr int DU = 256 * 16;   // floats per LDS unit buffer: 256 rows x 64 B

// Df: fragment-major image of the direct operand: [N/32 blocks][kp/16 k-steps][64 lanes][8 halves]
// Qb: unit-major image of the LDS operand: [M/256 tiles][kp/32 units][256 rows][32 halves]
template <int NB, int PD, int ABL, bool CHECK = false>
__global__ __launch_bounds__(PP_THREADS, 2) void probe_direct_kernel(const _Float16 *__restrict__ Df, const _Float16 *__restrict__ Qb,
                                                                     int kp, float *sink, float *dbg, int n_mtiles, int n_ntiles) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  const int nwg = n_mtiles * n_ntiles;
  const int xcd = blockIdx.x & 7, per_xcd = gridDim.x >> 3;
  const int q8 = nwg >> 3, r8 = nwg & 7;
  const int range_base = (xcd < r8) ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8;
  const int range_len = q8 + (xcd < r8 ? 1 : 0);
  int item = blockIdx.x >> 3;
  const int t = threadIdx.x, lane = t & 63;
  const int w8 = __builtin_amdgcn_readfirstlane(t >> 6);
  const int grp = w8 >> 2, wm = (w8 >> 1) & 1, wn = w8 & 1;
  const int lrow = lane & 31, half = lane >> 5;
  const int U = kp / 32, nks = kp / 16;
  const size_t q_tile_bytes = (size_t)256 * kp * 2;

  struct Tile {
    const char *q;        // the LDS operand's 256-row block
    const _Float16 *d;    // the wave's first fragment of the direct operand: block 8 nt + 4 grp + 2 wm, k-step 0, this lane
    unsigned q_bytes;
  };
  auto next = [&](Tile &s) -> bool {
    if (item >= range_len) return false;
    int mt, nt;
    supertile_order<4, 8>(range_base + item, n_ntiles, n_mtiles, nt, mt);
    item += per_xcd;
    s.q = reinterpret_cast<const char *>(Qb) + (size_t)mt * q_tile_bytes;
    s.q_bytes = (unsigned)q_tile_bytes;
    s.d = Df + ((size_t)(8 * nt + 4 * grp + 2 * wm) * nks) * 512;
    return true;
  };
  Tile cur, nxt;
  if (!next(cur)) return;
  bool have_nxt = next(nxt);
  if (!have_nxt) nxt = cur, nxt.q_bytes = 0u;

  // DMA duty: rows 32 w8 .. 32 w8 + 31 of a unit = two pieces of 16 rows x 64 B; slot lane & 3 of row r holds piece
  // (lane & 3) ^ ((r >> 2) & 3)  (source-side swizzle, as mfma_pp_f16.h)
  const int cpiece = (lane & 3) ^ ((lane >> 4) & 3);
  int voff[2];
#pragma unroll
  for (int i = 0; i < 2; ++i) voff[i] = (32 * w8 + 16 * i + (lane >> 2)) * 64 + cpiece * 16;
  auto dma = [&](int uu, int buf) {      // stream unit: uu < U -> cur, else nxt
    if constexpr (ABL & 2) return;
    const bool spill = uu >= U;
    const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<char *>(spill ? nxt.q : cur.q), 0, (int)(spill ? nxt.q_bytes : cur.q_bytes), 0x00020000);
    const int soff = (spill ? uu - U : uu) * 16384;
    float *base = lds + buf * DU + (32 * w8) * 16;
#pragma unroll
    for (int i = 0; i < 2; ++i)
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, (__attribute__((address_space(3))) void *)(base + 16 * i * 16), 16, voff[i],
                                               soff, 0, 0);
  };
  const int sw = (lrow >> 2) & 3;
  int cj[2];
#pragma unroll
  for (int j = 0; j < 2; ++j) cj[j] = ((2 * j + half) ^ sw) * 4;
  const int offq = (128 * wn + lrow) * 16;
  struct QF {
    f16x8 b[4];
  };
  auto readq = [&](int buf, int j, QF &f) {
    if constexpr (ABL & 4) return;
    const float *p = lds + buf * DU + cj[j] + offq;
#pragma unroll
    for (int ni = 0; ni < 4; ++ni) f.b[ni] = *reinterpret_cast<const f16x8 *>(p + 32 * ni * 16);
  };
  struct DF {
    f16x8 a[2];
  };
  static_assert(PD >= 2 && PD <= 4, "register ring written out for two to four units");
  DF d00, d01, d10, d11, d20, d21, d30, d31;   // [ring slot][k-step]: named, so that the ring stays in registers
  auto loadd = [&](int uu, int j, DF &dst) {   // fragments (mi = 0, 1) of stream unit uu, k-step j
    if constexpr (ABL & 16) return;
    const bool spill = uu >= U;
    const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<_Float16 *>(spill ? nxt.d : cur.d), 0, 0x7ffffff0, 0x00020000);
    const int soff = (2 * (spill ? uu - U : uu) + j) * 1024;
#pragma unroll
    for (int mi = 0; mi < 2; ++mi) {
      const u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(rsrc, lane * 16, soff + mi * nks * 1024, 0);
      dst.a[mi] = __builtin_bit_cast(f16x8, v);
    }
  };
  f32x16 acc[2][4];
  auto mma = [&](const DF &d, const QF &q) {
    if constexpr (ABL & 8) return;
#pragma unroll
    for (int ni = 0; ni < 4; ++ni)
#pragma unroll
      for (int mi = 0; mi < 2; ++mi) acc[mi][ni] = __builtin_amdgcn_mfma_f32_32x32x16_f16(d.a[mi], q.b[ni], acc[mi][ni], 0, 0, 0);
  };
  auto zero = [&]() {
#pragma unroll
    for (int mi = 0; mi < 2; ++mi)
#pragma unroll
      for (int ni = 0; ni < 4; ++ni)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[mi][ni][r] = 0.f;
  };
  float keep = 0.f;
  bool first_tile = true;
  auto emit = [&]() {
    if (CHECK && first_tile && blockIdx.x == 0) {   // the first tile of workgroup 0, whole: checked on the host
#pragma unroll
      for (int mi = 0; mi < 2; ++mi)
#pragma unroll
        for (int ni = 0; ni < 4; ++ni)
#pragma unroll
          for (int r = 0; r < 16; ++r) {
            const int drow = 128 * grp + 64 * wm + 32 * mi + (r & 3) + 8 * (r >> 2) + 4 * half;
            const int qrow = 128 * wn + 32 * ni + lrow;
            dbg[drow * 256 + qrow] = acc[mi][ni][r];
          }
    }
    first_tile = false;
#pragma unroll
    for (int mi = 0; mi < 2; ++mi)
#pragma unroll
      for (int ni = 0; ni < 4; ++ni) keep += acc[mi][ni][0] + acc[mi][ni][7];
  };
  QF q0, q1;
  {
    const f16x8 one = {1, 1, 1, 1, 1, 1, 1, 1};
#pragma unroll
    for (int i = 0; i < 4; ++i) q0.b[i] = q1.b[i] = one;
    d00.a[0] = d00.a[1] = d01.a[0] = d01.a[1] = d10.a[0] = d10.a[1] = d11.a[0] = d11.a[1] = one;
    d20.a[0] = d20.a[1] = d21.a[0] = d21.a[1] = d30.a[0] = d30.a[1] = d31.a[0] = d31.a[1] = one;
  }
  // prologue: LDS units 0 .. NB-2; direct fragments (., 0) of units 0 .. PD-1 and (., 1) of units 0 .. PD-2 -- the state the
  // steady loop has at the start of a unit.  Everything lands before the loop starts (once per workgroup); from then on
  // every unit issues exactly six vector-memory operations [2 DMA pieces, 2 loads (u - 1 + PD, 1), 2 loads (u + PD, 0)],
  // which is what the counted wait relies on.
  for (int g = 0; g < NB - 1; ++g) dma(g, g);
  loadd(0, 0, d00);
  loadd(0, 1, d01);
  loadd(1, 0, d10);
  if constexpr (PD >= 3) {
    loadd(1, 1, d11);
    loadd(2, 0, d20);
  }
  if constexpr (PD >= 4) {
    loadd(2, 1, d21);
    loadd(3, 0, d30);
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  zero();
  int buf = 0;   // LDS ring position of the unit being computed
  auto ring = [](int b) { return b >= NB ? b - NB : b; };
  bool pending = false;   // the previous unit's second k-step is still to be multiplied (it runs under this unit's first reads)
  // one unit: `c0` = this unit's ring slot, k-step 0; `p1` = the previous unit's slot, k-step 1
  auto unit = [&](int u, DF &c0, DF &p1) {
    constexpr int W = 6 * (PD - 1);
    // unit u landed (own DMA pieces: ancient; direct fragments of (u, 0) and (u - 1, 1): everything but the last PD - 1
    // units' issues), then everybody's pieces; everybody is done reading the previous unit's buffer
    if constexpr (ABL & 1) asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)" ::"n"(W) : "memory");
    else asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)\n\ts_barrier" ::"n"(W) : "memory");
    __builtin_amdgcn_sched_barrier(0);
    dma(u + NB - 1, ring(buf + NB - 1));
    readq(buf, 0, q0);
    if (pending) mma(p1, q1);
    if (u == 0 && pending) {   // that was the last k-step of the previous tile
      emit();
      zero();
    }
    __builtin_amdgcn_sched_barrier(0);
    loadd(u - 1 + PD, 1, p1);
    readq(buf, 1, q1);
    mma(c0, q0);
    __builtin_amdgcn_sched_barrier(0);
    loadd(u + PD, 0, c0);
    pending = true;
    buf = ring(buf + 1);
  };
  while (true) {
    for (int u0 = 0; u0 < U; u0 += PD) {
      if constexpr (PD == 4) {
        unit(u0, d00, d31);
        unit(u0 + 1, d10, d01);
        unit(u0 + 2, d20, d11);
        unit(u0 + 3, d30, d21);
      } else if constexpr (PD == 3) {
        unit(u0, d00, d21);
        unit(u0 + 1, d10, d01);
        unit(u0 + 2, d20, d11);
      } else {
        unit(u0, d00, d11);
        unit(u0 + 1, d10, d01);
      }
    }
    if (!have_nxt) break;
    cur = nxt;                 // the addressing moves on before the next tile's first unit issues anything
    have_nxt = next(nxt);
    if (!have_nxt) nxt.q_bytes = 0u;
  }
  if constexpr (PD == 4) mma((U - 1) % 4 == 0 ? d01 : ((U - 1) % 4 == 1 ? d11 : ((U - 1) % 4 == 2 ? d21 : d31)), q1);
  else if constexpr (PD == 3) mma((U - 1) % 3 == 0 ? d01 : ((U - 1) % 3 == 1 ? d11 : d21), q1);
  else mma((U - 1) % 2 == 0 ? d01 : d11, q1);
  emit();
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  if (keep == 12345.678f) sink[t] = keep;
}

template <int NB, int PD, int ABL, bool CHECK = false>
float run(const _Float16 *Df, const _Float16 *Qb, int M, int N, int kp, float *sink, float *dbg) {
  const int n_mtiles = M / 256, n_ntiles = N / 256;
  const size_t lds_b = (size_t)NB * DU * sizeof(float);
  hipFuncSetAttribute(reinterpret_cast<const void *>(probe_direct_kernel<NB, PD, ABL, CHECK>), hipFuncAttributeMaxDynamicSharedMemorySize,
                      (int)lds_b);
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  float best = 1e30f;
  for (int rep = 0; rep < 4; ++rep) {
    hipEventRecord(e0);
    hipLaunchKernelGGL((probe_direct_kernel<NB, PD, ABL, CHECK>), dim3(256), dim3(PP_THREADS), lds_b, 0, Df, Qb, kp, sink, dbg, n_mtiles,
                       n_ntiles);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    if (rep && ms < best) best = ms;
  }
  if (hipGetLastError() != hipSuccess) printf("launch error\n");
  return best;
}

int main() {
  const int M = 76800, N = 2304, kp = 2304;   // the problem of stream_probe.hip: the LDS operand has M rows, the direct one N
  const int U = kp / 32, nks = kp / 16;
  std::vector<_Float16> q((size_t)M * kp), d((size_t)N * kp);   // plain row-major values
  unsigned s = 1u;
  for (auto &v : q) {
    s = s * 1664525u + 1013904223u;
    v = (_Float16)(((int)(s >> 16) % 2001 - 1000) * 0.001f);
  }
  for (auto &v : d) {
    s = s * 1664525u + 1013904223u;
    v = (_Float16)(((int)(s >> 16) % 2001 - 1000) * 0.001f);
  }
  std::vector<_Float16> qb(q.size()), df(d.size());
  for (int r = 0; r < M; ++r)
    for (int k = 0; k < kp; ++k)
      qb[((size_t)(r / 256) * U + k / 32) * 8192 + (size_t)(r % 256) * 32 + k % 32] = q[(size_t)r * kp + k];
  for (int r = 0; r < N; ++r)
    for (int k = 0; k < kp; ++k) {
      const int lane = (r % 32) + 32 * ((k % 16) / 8);
      df[((size_t)(r / 32) * nks + k / 16) * 512 + lane * 8 + k % 8] = d[(size_t)r * kp + k];
    }
  _Float16 *Df, *Qb;
  float *sink, *dbg;
  hipMalloc(&Df, df.size() * 2);
  hipMalloc(&Qb, qb.size() * 2);
  hipMalloc(&sink, 4096);
  hipMalloc(&dbg, 256 * 256 * 4);
  hipMemcpy(Df, df.data(), df.size() * 2, hipMemcpyHostToDevice);
  hipMemcpy(Qb, qb.data(), qb.size() * 2, hipMemcpyHostToDevice);
  // correctness of the first tile of workgroup 0 (supertile item 0 = tiles nt 0, mt 0)
  hipMemset(dbg, 0, 256 * 256 * 4);
  run<8, 3, 0, true>(Df, Qb, M, N, kp, sink, dbg);
  std::vector<float> got(256 * 256);
  hipMemcpy(got.data(), dbg, got.size() * 4, hipMemcpyDeviceToHost);
  double worst = 0;
  for (int i = 0; i < 256; i += 7)
    for (int j = 0; j < 256; j += 5) {
      double ref = 0;
      for (int k = 0; k < kp; ++k) ref += (double)(float)d[(size_t)i * kp + k] * (double)(float)q[(size_t)j * kp + k];
      worst = std::fmax(worst, std::fabs(ref - got[i * 256 + j]));
    }
  printf("first tile against float64: max |diff| %.3g (values ~ %.3g)\n", worst, std::fabs((double)got[3 * 256 + 5]));
  const double flop = 2.0 * M * N * kp;
  struct { const char *name; float ms; } r[] = {
      {"direct operand, 8 LDS units, 3 units ahead", run<8, 3, 0>(Df, Qb, M, N, kp, sink, nullptr)},
      {"direct operand, 8 LDS units, 4 units ahead", run<8, 4, 0>(Df, Qb, M, N, kp, sink, nullptr)},
      {"direct operand, 9 LDS units, 3 units ahead", run<9, 3, 0>(Df, Qb, M, N, kp, sink, nullptr)},
      {"direct operand, 10 LDS units, 4 units ahead", run<10, 4, 0>(Df, Qb, M, N, kp, sink, nullptr)},
      {"direct operand, 6 LDS units, 3 units ahead", run<6, 3, 0>(Df, Qb, M, N, kp, sink, nullptr)},
      {"direct operand, 8 LDS units, 2 units ahead", run<8, 2, 0>(Df, Qb, M, N, kp, sink, nullptr)},
      {"  no DMA", run<8, 3, 2>(Df, Qb, M, N, kp, sink, nullptr)},
      {"  no LDS reads", run<8, 3, 4>(Df, Qb, M, N, kp, sink, nullptr)},
      {"  no direct loads", run<8, 3, 16>(Df, Qb, M, N, kp, sink, nullptr)},
      {"  no DMA, no LDS reads, no direct loads (MFMA + barriers)", run<8, 3, 22>(Df, Qb, M, N, kp, sink, nullptr)},
      {"  no MFMA", run<8, 3, 8>(Df, Qb, M, N, kp, sink, nullptr)}};
  for (auto &x : r) printf("%-60s %8.3f ms  %7.1f TFLOP/s (f16)\n", x.name, x.ms, flop / x.ms / 1e9);
  return 0;
}
