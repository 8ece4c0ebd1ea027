// Tile loop of the split-precision GEMM (gemm_split.hip): the persistent LDS-DMA stream of mfma_pp_f16.h re-cut so that
// every operand slab crosses global -> LDS ONCE per 32 k.
//
// A product a.w is three f16 MFMAs (a_lo w_hi + a_hi w_hi + a_hi w_lo).  Walking K three times over plain (hi | lo) rows
// -- the first version of the kernel -- fetches a_hi and w_hi twice: 96 KiB of DMA per 32 k and CU.  The probe
// (tools/probes/stream_probe.hip, profiles/r02_stream_ablation.txt) shows that loop bound by exactly that DMA: 49 GB/s
// per CU = three 32 KiB units in flight at ~2 us latency, matrix pipes 49 % idle.  Here one SUPER-UNIT = 32 k of all four
// slabs (a_lo, w_hi, a_hi, w_lo; 16 KiB each: 256 rows x 64 B) = 64 KiB of DMA for the same 48 MFMAs per wave, fragments of
// a_hi and w_hi are read from LDS once for two product sets (24 ds_read_b128 per 48 MFMAs instead of 36), and there
// is one barrier per 48 MFMAs instead of three.
//
// LDS = a ring of NSLAB = 10 slabs (160 KiB, the whole CU): slab 4 s + t holds type t of super-unit s,
//   t = 0: a_lo   1: w_hi   2: a_hi   3: w_lo          (waves 4-7 stage the a slabs, waves 0-3 the w slabs: 4 pieces of
//                                                        16 rows x 64 B per wave and slab, source swizzle as mfma_pp_f16.h)
// Window s (the stream runs ACROSS tiles) computes super-unit s and issues the DMA of slabs 4s+6 .. 4s+9: the second half
// of super-unit s+1, then the first half of s+2 -- six slabs (96 KiB) in flight, the same depth as before for 2/3 of the
// bytes.  At its end `s_waitcnt vmcnt(4)` (the four pieces of this window's second half may still fly) + barrier: super-
// unit s+1 has landed, every wave has finished reading s, whose ring positions the NEXT window's DMA overwrites.
// Per wave and window (F0 / F1 = fragment sets of k-steps j = 0 / 1; a fragment set = w_hi[2] w_lo[2] a_lo[4] a_hi[4]):
//   PA: ds_read F0 <- (s, j=0) | 24 MFMA on F1 = (s-1, j=1) | 4 DMA pieces
//   PB: ds_read F1 <- (s, j=1) | 24 MFMA on F0             | 4 DMA pieces
// Order of accumulation (the skinny kernel reproduces it): per 16 k, a_lo w_hi, then a_hi w_hi, then a_hi w_lo.
// W is the MFMA's first operand, the activations its second (see gemm_split.hip: four consecutive columns per lane).
#pragma once

#include "mfma_pp_f16.h"

namespace mevi {

constexpr int SS_NSLAB = 10;
constexpr int SS_SLAB = 256 * H1_LD;  // floats per slab (16 KiB)
constexpr size_t ss_lds_bytes() { return (size_t)SS_NSLAB * SS_SLAB * sizeof(float); }

// next(H1Src &) / begin() / emit(acc) as h1_tile_stream.  `U` super-units per tile (>= 2), `lo_bytes` = byte offset of the
// lo half inside an image row; `unit_bytes` = distance of consecutive 32-k units (64 for plain rows; the probe's unit-major
// variant passes row_bytes 64, lo_bytes 16384, unit_bytes 32768).
// ABL: ablation switches for tools/probes/stream_probe.hip, as h1_tile_stream (the product instantiates 0).
template <class Next, class Begin, class Emit, int ABL = 0>
__device__ __forceinline__ void split_tile_stream(int row_bytes, int lo_bytes, int U, float *lds, Next next, Begin begin,
                                                  Emit emit, int unit_bytes = 64) {
  const int t = threadIdx.x;
  const int lane = t & 63;
  const int w8 = __builtin_amdgcn_readfirstlane(t >> 6);
  const int grp = w8 >> 2, wm = (w8 >> 1) & 1, wn = w8 & 1;
  const int lrow = lane & 31;
  const int half = lane >> 5;
  const bool is_w = w8 < 4;
  const int cpiece = (lane & 3) ^ ((lane >> 4) & 3);
  int voff[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) voff[i] = (64 * (w8 & 3) + 16 * i + (lane >> 2)) * row_bytes + cpiece * 16;

  H1Src cur, nxt;
  if (!next(cur)) return;
  bool have_nxt = next(nxt);
  if (!have_nxt) nxt.bytes = 0u, nxt.src = cur.src;

  // the wave's four pieces of the slab at ring position `pos`: unit `uu` of the tile stream (uu >= U: next tile), lo / hi
  auto dma_slab = [&](int pos, int uu, bool lo) {
    if constexpr (ABL & 2) return;
    const bool spill = uu >= U;
    const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<void *>(spill ? nxt.src : cur.src), 0, (int)(spill ? nxt.bytes : cur.bytes), 0x00020000);
    const int soff = (spill ? uu - U : uu) * unit_bytes + (lo ? lo_bytes : 0);
    float *base = lds + pos * SS_SLAB + (64 * (w8 & 3)) * H1_LD;
#pragma unroll
    for (int i = 0; i < 4; ++i)
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, (__attribute__((address_space(3))) void *)(base + 16 * i * H1_LD), 16,
                                               voff[i], soff, 0, 0);
  };
  auto ring = [](int p) { return p >= SS_NSLAB ? p - SS_NSLAB : p; };

  const int sw = (lrow >> 2) & 3;
  int cj[2];
#pragma unroll
  for (int j = 0; j < 2; ++j) cj[j] = ((2 * j + half) ^ sw) * 4;
  const int offw = (grp * BM + 64 * wm + lrow) * H1_LD;  // w slabs: the wave's 64 W rows (2 fragments)
  const int offa = (128 * wn + lrow) * H1_LD;            // a slabs: the wave's 128 activation rows (4 fragments)
  struct Frag {
    f16x8 wh[2], wl[2], al[4], ah[4];
  };
  auto read = [&](int base, int j, Frag &f) {  // base = ring position of the super-unit's slab 0
    if constexpr (ABL & 4) return;
    const float *pal = lds + base * SS_SLAB + cj[j] + offa;
    const float *pwh = lds + ring(base + 1) * SS_SLAB + cj[j] + offw;
    const float *pah = lds + ring(base + 2) * SS_SLAB + cj[j] + offa;
    const float *pwl = lds + ring(base + 3) * SS_SLAB + cj[j] + offw;
#pragma unroll
    for (int mi = 0; mi < 2; ++mi) f.wh[mi] = *reinterpret_cast<const f16x8 *>(pwh + 32 * mi * H1_LD);
#pragma unroll
    for (int ni = 0; ni < 4; ++ni) f.al[ni] = *reinterpret_cast<const f16x8 *>(pal + 32 * ni * H1_LD);
#pragma unroll
    for (int ni = 0; ni < 4; ++ni) f.ah[ni] = *reinterpret_cast<const f16x8 *>(pah + 32 * ni * H1_LD);
#pragma unroll
    for (int mi = 0; mi < 2; ++mi) f.wl[mi] = *reinterpret_cast<const f16x8 *>(pwl + 32 * mi * H1_LD);
  };
  f32x16 acc[2][4];
  auto mma = [&](const Frag &f) {
    if constexpr (ABL & 8) return;
#pragma unroll
    for (int ni = 0; ni < 4; ++ni)
#pragma unroll
      for (int mi = 0; mi < 2; ++mi)
        acc[mi][ni] = __builtin_amdgcn_mfma_f32_32x32x16_f16(f.wh[mi], f.al[ni], acc[mi][ni], 0, 0, 0);
#pragma unroll
    for (int ni = 0; ni < 4; ++ni)
#pragma unroll
      for (int mi = 0; mi < 2; ++mi)
        acc[mi][ni] = __builtin_amdgcn_mfma_f32_32x32x16_f16(f.wh[mi], f.ah[ni], acc[mi][ni], 0, 0, 0);
#pragma unroll
    for (int ni = 0; ni < 4; ++ni)
#pragma unroll
      for (int mi = 0; mi < 2; ++mi)
        acc[mi][ni] = __builtin_amdgcn_mfma_f32_32x32x16_f16(f.wl[mi], f.ah[ni], acc[mi][ni], 0, 0, 0);
  };

  Frag F0, F1;
  if constexpr (ABL & 4) {
    const f16x8 one = {1, 1, 1, 1, 1, 1, 1, 1};
#pragma unroll
    for (int i = 0; i < 2; ++i) F0.wh[i] = F0.wl[i] = F1.wh[i] = F1.wl[i] = one;
#pragma unroll
    for (int i = 0; i < 4; ++i) F0.al[i] = F0.ah[i] = F1.al[i] = F1.ah[i] = one;
  }
  int base = 0;  // ring position of slab 0 of the super-unit being computed

  auto window = [&](int u, bool first) {
    // this window's DMA: slabs 4s+6 (a_hi, u+1) / 4s+7 (w_lo, u+1), then 4s+8 (a_lo, u+2) / 4s+9 (w_hi, u+2).  All eight
    // pieces are issued at the START of the window (their ring positions, super-unit s-1's, are free since the barrier):
    // the first four must land within this window, every cycle of head start counts
    dma_slab(ring(base + (is_w ? 7 : 6)), u + 1, is_w);
    dma_slab(ring(base + (is_w ? 9 : 8)), u + 2, !is_w);
    read(base, 0, F0);
    if (!first) mma(F1);
    __builtin_amdgcn_sched_group_barrier(0x020, 8, 0);
    if (!first) {
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        __builtin_amdgcn_sched_group_barrier(0x100, 3, 0);
        __builtin_amdgcn_sched_group_barrier(0x008, 6, 0);
      }
    }
    __builtin_amdgcn_sched_barrier(0);
    read(base, 1, F1);
    mma(F0);
    __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);  // first MFMA ahead of the reads: it waits for F0, not for these
#pragma unroll
    for (int i = 0; i < 3; ++i) {
      __builtin_amdgcn_sched_group_barrier(0x100, 3, 0);
      __builtin_amdgcn_sched_group_barrier(0x008, 6, 0);
    }
    __builtin_amdgcn_sched_group_barrier(0x100, 3, 0);
    __builtin_amdgcn_sched_group_barrier(0x008, 5, 0);
    __builtin_amdgcn_sched_barrier(0);
    // super-unit s+1 landed once only this window's last four pieces are outstanding; own reads of s done
    if constexpr (ABL & 1) asm volatile("s_waitcnt vmcnt(4) lgkmcnt(0)" ::: "memory");
    else asm volatile("s_waitcnt vmcnt(4) lgkmcnt(0)\n\ts_barrier" ::: "memory");
    __builtin_amdgcn_sched_barrier(0);
    base = ring(base + 4);
  };

  // prologue: super-unit 0 (slabs 0-3) and the first half of super-unit 1 (slabs 4, 5)
  dma_slab(is_w ? 1 : 0, 0, !is_w);  // w_hi | a_lo
  dma_slab(is_w ? 3 : 2, 0, is_w);   // w_lo | a_hi
  dma_slab(is_w ? 5 : 4, 1, !is_w);  // w_hi | a_lo of unit 1
  asm volatile("s_waitcnt vmcnt(4)\n\ts_barrier" ::: "memory");
  __builtin_amdgcn_sched_barrier(0);
  while (true) {
#pragma unroll
    for (int mi = 0; mi < 2; ++mi)
#pragma unroll
      for (int ni = 0; ni < 4; ++ni)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[mi][ni][r] = 0.f;
    begin();
    window(0, true);
    for (int u = 1; u < U; ++u) window(u, false);
    mma(F1);
    emit(acc);
    if (!have_nxt) break;
    cur = nxt;
    have_nxt = next(nxt);
    if (!have_nxt) nxt.bytes = 0u;
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // the empty tail pieces: nothing may target LDS past the loop
}

}  // namespace mevi
