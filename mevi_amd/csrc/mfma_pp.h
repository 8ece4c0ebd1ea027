// Shared core of the f32-MFMA "ping-pong" tile kernels (dense filter, GEMM).  gfx950 only.
//
// One 512-thread workgroup per CU, two 4-wave groups.  Group g owns a 128-row tile of
// operand A; both share one tile of QT = 64*NI rows of operand B (both operands are
// row-major with K contiguous, i.e. C = A . B^T).  Inside a group, wave (wm, wn) owns
// 64 x 32*NI outputs = 2 x NI accumulators of v_mfma_f32_32x32x2_f32.
// The two waves that share a SIMD belong to different groups and alternate roles:
//   MFMA role    -- 16 k-steps x (2*NI) MFMAs on slab s.  A wave issues in order and the
//                   matrix pipe holds ONE MFMA (64 cycles), so every other instruction sits
//                   in the shadow of an MFMA: the stream is pinned (sched_group_barrier) to
//                   {MFMA, ds_read} x(2+NI), {MFMA, global_load}, MFMA... per pair of
//                   k-steps; fragment reads run two pairs ahead; the global loads of the
//                   slab this group stages next are spread one per pair.
//   staging role -- convert the staged registers into the LDS image of the next slab
//                   (raised priority: a handful of ds_write2_b32).
// Hand-over is early: the MFMA role executes its barrier once its last LDS read has been
// issued (after pair 5 of 8), so the other group warms up underneath the remaining MFMAs
// and the pipe does not idle at the switch.
//   G0: [compute(s) + loads(s+1)] [lstore(s+1)] barrier ...
//   G1: [lstore(s+1)] barrier [compute(s) + loads(s+2)] ...
// LDS image: row-major [row][k], natural k order, odd row stride (33 floats) so that both
// the b32 fragment reads (32 rows, same k) and the staging stores are bank-conflict free.
// MFMA lane half h consumes k = 2j + h, which makes every output the sequential f32 fmaf
// chain over k = 0..K-1 (bit-exact contract with the CPU oracle).
// Measured: 85 % matrix-pipe utilisation at 2.38 GHz on the C2 dense workload;
// development used an s_memtime-stamped build of this loop (profiles/README.md).
#pragma once

#include "common.h"

namespace mevi {

typedef float f32x16 __attribute__((ext_vector_type(16)));

constexpr int BM = 128;          // A rows per wave-group tile
constexpr int BK = 32;           // K slab
constexpr int PP_THREADS = 512;
constexpr int PP_LD = 33;        // floats per LDS row

template <int NI>
constexpr size_t pp_lds_bytes() {
  return (size_t)2 * (2 * BM + 64 * NI) * PP_LD * sizeof(float);
}

// XCD-aware bijective block remap: blocks that share `bid % 8` share an XCD (L2), so give
// each XCD a contiguous range of work items -- neighbouring tiles then hit the same L2.
__device__ inline int xcd_remap(int bid, int nwg) {
  const int q = nwg >> 3, r = nwg & 7, xcd = bid & 7;
  const int base = (xcd < r) ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q;
  return base + (bid >> 3);
}

// Work-item order inside an XCD's contiguous range: super-tiles of SD A-tiles x SQ B-tiles, so that the
// ~32 workgroups resident on one XCD share BOTH operands through its 4 MiB L2 (a plain row-major order
// shares the A tile only and re-reads every B tile from the Infinity Cache).  Bijective for any shape.
template <int SD, int SQ>
__device__ inline void supertile_order(int wg, int n_a, int n_b, int &a_tile, int &b_tile) {
  const int band_size = SD * n_b;
  const int band = wg / band_size;
  const int r = wg - band * band_size;
  const int sd = min(SD, n_a - band * SD);
  const int n_groups = (n_b + SQ - 1) / SQ;
  int g = r / (sd * SQ);
  if (g > n_groups - 1) g = n_groups - 1;
  const int r2 = r - g * sd * SQ;
  const int sq = (g == n_groups - 1) ? n_b - g * SQ : SQ;
  const int d_in = r2 / sq;
  a_tile = band * SD + d_in;
  b_tile = g * SQ + (r2 - d_in * sq);
}

// aptr[i]: this thread's staging pointers into its group's A tile (row srow+32i, + skq),
// bptr[i]: into the shared B tile (row (QT/2)*grp + srow + 32i, + skq); rows pre-clamped.
// acc[mi][ni]: 32x32 accumulators of wave (wm, wn): A rows 64*wm + 32*mi + ..., B rows
// 32*NI*wn + 32*ni + ...  (C/D map: col = lane&31 -> B row, row = (r&3)+8*(r>>2)+4*half -> A row).
template <int NI, bool KTAIL>
__device__ __forceinline__ void pp_mainloop(const float *const (&aptr)[4], const float *const (&bptr)[NI],
                                            int kdim, float *lds, f32x16 (&acc)[2][NI]) {
  constexpr int QT = 64 * NI;
  constexpr int ROWS = 2 * BM + QT;  // LDS rows per buffer: A0[128] | A1[128] | B[QT]
  constexpr int NLOAD = 4 + NI;      // float4 global loads per thread per slab

  const int t = threadIdx.x;
  const int grp = __builtin_amdgcn_readfirstlane(t >> 8);  // wave-uniform
  const int tg = t & 255;
  const int lane = t & 63;
  const int wave = tg >> 6;
  const int wm = wave >> 1;
  const int wn = wave & 1;
  const int lrow = lane & 31;
  const int half = lane >> 5;
  const int srow = tg >> 3;
  const int skq = (tg & 7) * 4;

  float4 ra[4], rb[NI];
  const int nslab = (kdim + BK - 1) / BK;

  auto gload_one = [&](int s, int i) {
    int kk = s * BK;
    if (KTAIL && kk + skq + 4 > kdim) kk = kdim - 4 - skq;  // stay in bounds; zeroed in lstore
    if (i < 4) ra[i] = *reinterpret_cast<const float4 *>(aptr[i] + kk);
    else rb[i - 4] = *reinterpret_cast<const float4 *>(bptr[i - 4] + kk);
  };
  auto gload = [&](int s) {
#pragma unroll
    for (int i = 0; i < NLOAD; ++i) gload_one(s, i);
  };
  auto lstore = [&](int s) {
    float *base = lds + (s & 1) * ROWS * PP_LD;
    float *sA = base + (grp * BM + srow) * PP_LD + skq;
    float *sB = base + (2 * BM + (QT / 2) * grp + srow) * PP_LD + skq;
    const bool zero = KTAIL && (s * BK + skq >= kdim);
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      float4 v = ra[i];
      if (zero) v = make_float4(0.f, 0.f, 0.f, 0.f);
      float *p = sA + 32 * i * PP_LD;
      p[0] = v.x; p[1] = v.y; p[2] = v.z; p[3] = v.w;
    }
#pragma unroll
    for (int i = 0; i < NI; ++i) {
      float4 v = rb[i];
      if (zero) v = make_float4(0.f, 0.f, 0.f, 0.f);
      float *p = sB + 32 * i * PP_LD;
      p[0] = v.x; p[1] = v.y; p[2] = v.z; p[3] = v.w;
    }
  };

  auto compute = [&](int s, int gs, bool with_barrier) {
    const float *base = lds + (s & 1) * ROWS * PP_LD;
    const float *pa = base + (grp * BM + 64 * wm + lrow) * PP_LD + half;
    const float *pb = base + (2 * BM + 32 * NI * wn + lrow) * PP_LD + half;
    float av[2][16], bv[NI][16];
    auto ld = [&](int pr) {
#pragma unroll
      for (int j = 2 * pr; j < 2 * pr + 2; ++j) {
#pragma unroll
        for (int mi = 0; mi < 2; ++mi) av[mi][j] = pa[32 * mi * PP_LD + 2 * j];
#pragma unroll
        for (int ni = 0; ni < NI; ++ni) bv[ni][j] = pb[32 * ni * PP_LD + 2 * j];
      }
    };
    ld(0);
    ld(1);
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int pr = 0; pr < 8; ++pr) {
      if (pr + 2 < 8) ld(pr + 2);
      if (gs >= 0 && pr < NLOAD) gload_one(gs, pr);
#pragma unroll
      for (int j = 2 * pr; j < 2 * pr + 2; ++j)
#pragma unroll
        for (int mi = 0; mi < 2; ++mi)
#pragma unroll
          for (int ni = 0; ni < NI; ++ni)
            acc[mi][ni] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[mi][j], bv[ni][j], acc[mi][ni], 0, 0, 0);
      // sched_group_barrier(mask, count, sync id): 0x008 MFMA, 0x100 DS read, 0x020 VMEM read
      constexpr int NM = 4 * NI;  // MFMAs in this pair of k-steps
      const bool rd = (pr + 2 < 8), gl = (gs >= 0 && pr < NLOAD);
      if (rd) {
#pragma unroll
        for (int i = 0; i < 2 + NI; ++i) {
          __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
          __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
        }
      }
      if (gl) {
        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
        __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);
      }
      if (rd && gl) __builtin_amdgcn_sched_group_barrier(0x008, NM - (2 + NI) - 1, 0);
      else if (rd) __builtin_amdgcn_sched_group_barrier(0x008, NM - (2 + NI), 0);
      else if (gl) __builtin_amdgcn_sched_group_barrier(0x008, NM - 1, 0);
      else __builtin_amdgcn_sched_group_barrier(0x008, NM, 0);
      // Early hand-over: after pair 5 every LDS read of this slab has been issued (the
      // barrier's lgkmcnt(0) retires them), so the other group may start its phase now.
      if (pr == 5) {
        __builtin_amdgcn_sched_barrier(0);
        if (with_barrier) __syncthreads();
      }
    }
    __builtin_amdgcn_sched_barrier(0);
  };

#pragma unroll
  for (int mi = 0; mi < 2; ++mi)
#pragma unroll
    for (int ni = 0; ni < NI; ++ni)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[mi][ni][r] = 0.f;

  // ---- prologue: slab 0 staged by everyone; G1 already has slab 1 in flight
  gload(0);
  lstore(0);
  if (grp == 1 && nslab > 1) gload(1);
  __syncthreads();

  // Two role-specialised loops (wave-uniform branch; both execute 2*(nslab-1) barriers,
  // one inside compute() and one after lstore()).
  if (grp == 0) {
    for (int s = 0; s + 1 < nslab; ++s) {
      compute(s, s + 1, true);
      __builtin_amdgcn_s_setprio(3);
      lstore(s + 1);
      __builtin_amdgcn_s_setprio(0);
      __syncthreads();
    }
  } else {
    for (int s = 0; s + 2 < nslab; ++s) {
      __builtin_amdgcn_s_setprio(3);
      lstore(s + 1);
      __builtin_amdgcn_s_setprio(0);
      __syncthreads();
      compute(s, s + 2, true);
    }
    if (nslab >= 2) {
      lstore(nslab - 1);
      __syncthreads();
      compute(nslab - 2, -1, true);
    }
  }
  compute(nslab - 1, -1, false);  // last slab: nothing left to stage, both groups run together
}

}  // namespace mevi
