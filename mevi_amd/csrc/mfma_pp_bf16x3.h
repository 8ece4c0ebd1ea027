// bf16x3 split variant of the ping-pong tile loop (see mfma_pp.h for the structure).
//
// Operands are pre-split f32 -> (hi, lo) bf16 pairs, hi = bf16(x), lo = bf16(x - hi), stored
// "slab-interleaved": per row and per 32-wide k slab 64 B of hi followed by 64 B of lo (the row
// stays 4*dim bytes), which is exactly the 128-byte LDS row image.
//
//   acc += A_hi.B_hi + A_hi.B_lo + A_lo.B_hi       (v_mfma_f32_32x32x16_bf16, f32 accumulate)
//
// Three bf16 MFMAs per product at 16x the f32-MFMA rate = 5.3x the throughput of the exact-f32 loop.
// The result is an APPROXIMATION of the f32 dot product with a rigorous bound
//   |approx - chain| <= C_ERR * ||a|| * ||b||,  C_ERR = 2.5e-4
// (dropped lo.lo and split residuals <= 3*2^-16, f32 accumulation over 3*dim terms <= 2304*2^-24,
// the exact chain's own distance from the real sum <= 768*2^-24).  It is only ever used to SELECT
// candidates that are then re-scored exactly (ip_topk.hip).
//
// Geometry: 512 threads = 8 waves, all computing (no role split: staging is DMA); block tile 256 A rows
// x 256 B rows; wave (grp, wm, wn) owns 64 x 128 outputs = 2 x 4 accumulators; per 16-wide k unit 24 MFMAs
// per wave; four LDS unit buffers, one barrier per unit, fragments double-buffered in registers.
#pragma once

#include "mfma_pp.h"

namespace mevi {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

constexpr int X3_QT = 256;       // B rows per workgroup
constexpr int X3_ROWS = 2 * BM + X3_QT;
constexpr int X3_LD = 16;        // floats (64 B) per LDS row of one 16-wide k unit: 32 B hi | 32 B lo, swizzled
constexpr int X3_NBUF = 4;       // units resident in LDS: one being read, three landing
constexpr int X3_UNIT = X3_ROWS * X3_LD;  // floats per unit buffer (32 KiB)
constexpr size_t x3_lds_bytes() { return (size_t)X3_NBUF * X3_UNIT * sizeof(float); }

// The K loop advances in UNITS of 16 k (half a slab of the split image): one v_mfma_f32_32x32x16_bf16 step.
// Per unit an LDS row holds four 16-byte pieces: c = 0,1 -> hi, k-halves 0,1; c = 2,3 -> lo, k-halves 0,1.
//
// Staging is LDS-DMA in its MUBUF form (buffer_load_dwordx4 ... lds): no staging registers, no ds_write, and --
// unlike global_load_lds, which the compiler books on lgkmcnt as an out-of-order FLAT event, degrading every
// `s_waitcnt lgkmcnt(N)` of the fragment pipeline to lgkmcnt(0) -- it is vmcnt only.  One wave-instruction moves
// 1 KiB = 16 rows x 64 B; the LDS destination is linear (M0 base + lane*16), so the bank swizzle is applied on
// the SOURCE address and again on the read:  logical piece c of row r lives in slot c ^ ((r >> 2) & 3).
// ds_read_b128 is serviced in lane groups {0-3,12-15,20-27} {4-11,16-19,28-31} (+32): with 64-byte rows the 16
// rows of a group then cover all 16 slots of the 256-byte bank line -> conflict free.
//
// A wave stages 64 rows of ONE operand per unit (4 pieces), so its buffer descriptor is wave-uniform: `src` =
// first row of the wave's operand tile (docs: waves 0-3, queries: waves 4-7), `src_bytes` = bytes from there to
// the end of the operand (rows past it read as 0 and are masked in the epilogue), rows `row_bytes` apart.
// LDS rows are A0[0,128) | A1[128,256) | B[256,512); all 8 waves compute: wave w8 = 4*grp + 2*wm + wn owns
// A rows 128*grp + 64*wm + [0,64) x B rows 128*wn + [0,128) = 2 x 4 accumulators.
//
// Pipeline (per wave, unit u, window W_u = barrier u-1 .. barrier u):
//   PA: ds_read A_u, B_u[nh=0]        | 12 MFMA of group (u-1, nh=1) | 2 DMA pieces of unit u+3
//   PB: ds_read B_u[nh=1]             | 12 MFMA of group (u,   nh=0) | 2 DMA pieces of unit u+3
//   s_waitcnt lgkmcnt(0) vmcnt(8); s_barrier      -- unit u+1 landed (units u+2, u+3 = 8 pieces stay in flight),
//                                                    every wave has finished READING unit u
// so an LDS round trip is covered by 12 MFMAs of the same wave, a DMA piece has two full windows to land, and
// the memory queue is never drained.  Buffer (u+3)&3 = (u-1)&3 is free from barrier u-1 on.
__device__ __forceinline__ void pp_mainloop_bf16x3(const float *src, unsigned int src_bytes, int row_bytes, int nslab,
                                                   float *lds, f32x16 (&acc)[2][4]) {
  const int t = threadIdx.x;
  const int lane = t & 63;
  const int w8 = __builtin_amdgcn_readfirstlane(t >> 6);
  const int grp = w8 >> 2, wm = (w8 >> 1) & 1, wn = w8 & 1;
  const int lrow = lane & 31;
  const int half = lane >> 5;
  const int nunits = 2 * nslab;
  // DMA piece i of this wave fills LDS rows 64*w8 + 16*i + (lane>>2), slot lane&3, which holds logical piece
  // c = (lane&3) ^ ((row>>2)&3), (row>>2)&3 = (lane>>4)&3;  source byte offset in the row: (c>>1)*64 + (c&1)*16
  const int cpiece = (lane & 3) ^ ((lane >> 4) & 3);
  int voff[4];
#pragma unroll
  for (int i = 0; i < 4; ++i)
    voff[i] = (64 * (w8 & 3) + 16 * i + (lane >> 2)) * row_bytes + (cpiece >> 1) * 64 + (cpiece & 1) * 16;

  // pieces [p0, p0+2) of unit u; units past the end get an empty descriptor (no fetch), so the number of
  // pieces in flight -- what the counted vmcnt relies on -- is the same in every window
  auto dma2 = [&](int u, int p0) {
    const __amdgpu_buffer_rsrc_t rsrc =
        __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(src), 0, u < nunits ? (int)src_bytes : 0, 0x00020000);
    float *base = lds + (u & (X3_NBUF - 1)) * X3_UNIT + (64 * w8) * X3_LD;
    const int soff = (u >> 1) * 128 + (u & 1) * 32;
#pragma unroll
    for (int i = p0; i < p0 + 2; ++i)
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, (__attribute__((address_space(3))) void *)(base + 16 * i * X3_LD), 16,
                                               voff[i], soff, 0, 0);
  };

  const int sw = (lrow >> 2) & 3;  // fragment rows are lrow + multiples of 32
  const int chi = ((half) ^ sw) * 4, clo = ((2 + half) ^ sw) * 4;  // float offsets of this lane's hi / lo piece
  const int offa = (grp * BM + 64 * wm + lrow) * X3_LD;
  const int offb = (2 * BM + 128 * wn + lrow) * X3_LD;
  auto readA = [&](int u, bf16x8 (&A)[4]) {  // ah0 ah1 al0 al1
    const float *pa = lds + (u & (X3_NBUF - 1)) * X3_UNIT + offa;
    A[0] = *reinterpret_cast<const bf16x8 *>(pa + chi);
    A[1] = *reinterpret_cast<const bf16x8 *>(pa + 32 * X3_LD + chi);
    A[2] = *reinterpret_cast<const bf16x8 *>(pa + clo);
    A[3] = *reinterpret_cast<const bf16x8 *>(pa + 32 * X3_LD + clo);
  };
  auto readB = [&](int u, int nh, bf16x8 (&B)[4]) {  // bh0 bh1 bl0 bl1 of B tiles 2nh, 2nh+1
    const float *pb = lds + (u & (X3_NBUF - 1)) * X3_UNIT + offb + 64 * nh * X3_LD;
    B[0] = *reinterpret_cast<const bf16x8 *>(pb + chi);
    B[1] = *reinterpret_cast<const bf16x8 *>(pb + 32 * X3_LD + chi);
    B[2] = *reinterpret_cast<const bf16x8 *>(pb + clo);
    B[3] = *reinterpret_cast<const bf16x8 *>(pb + 32 * X3_LD + clo);
  };
  auto mma = [&](const bf16x8 (&A)[4], const bf16x8 (&B)[4], int nh) {
#pragma unroll
    for (int nn = 0; nn < 2; ++nn)
#pragma unroll
      for (int mi = 0; mi < 2; ++mi) {
        f32x16 &c = acc[mi][2 * nh + nn];
        c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(A[2 + mi], B[nn], c, 0, 0, 0);  // lo . hi
        c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(A[mi], B[2 + nn], c, 0, 0, 0);  // hi . lo
        c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(A[mi], B[nn], c, 0, 0, 0);      // hi . hi
      }
  };

#pragma unroll
  for (int mi = 0; mi < 2; ++mi)
#pragma unroll
    for (int ni = 0; ni < 4; ++ni)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[mi][ni][r] = 0.f;

  bf16x8 A0[4], A1[4], B0[4], B1[4];

  // one window; FIRST: there is no group (u-1, 1) yet
  auto window = [&](int u, bf16x8 (&Acur)[4], const bf16x8 (&Aprev)[4], bool first) {
    readA(u, Acur);
    readB(u, 0, B0);
    if (!first) mma(Aprev, B1, 1);
    dma2(u + 3, 0);
    if (!first) {
      __builtin_amdgcn_sched_group_barrier(0x100, 8, 0);
      __builtin_amdgcn_sched_group_barrier(0x008, 3, 0);
      __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);
      __builtin_amdgcn_sched_group_barrier(0x008, 6, 0);
      __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);
      __builtin_amdgcn_sched_group_barrier(0x008, 3, 0);
    }
    __builtin_amdgcn_sched_barrier(0);
    readB(u, 1, B1);
    mma(Acur, B0, 0);
    dma2(u + 3, 2);
    // first MFMA ahead of the reads: its wait covers the fragments issued a group ago, not these
    __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
    __builtin_amdgcn_sched_group_barrier(0x100, 4, 0);
    __builtin_amdgcn_sched_group_barrier(0x008, 2, 0);
    __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);
    __builtin_amdgcn_sched_group_barrier(0x008, 6, 0);
    __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);
    __builtin_amdgcn_sched_group_barrier(0x008, 3, 0);
    __builtin_amdgcn_sched_barrier(0);
    // unit u+1 landed once at most the 8 pieces of units u+2, u+3 are outstanding; own reads of unit u done
    asm volatile("s_waitcnt vmcnt(8) lgkmcnt(0)\n\ts_barrier" ::: "memory");
    __builtin_amdgcn_sched_barrier(0);
  };

  dma2(0, 0);
  dma2(0, 2);
  dma2(1, 0);
  dma2(1, 2);
  dma2(2, 0);
  dma2(2, 2);
  asm volatile("s_waitcnt vmcnt(8)\n\ts_barrier" ::: "memory");  // unit 0 landed
  __builtin_amdgcn_sched_barrier(0);
  window(0, A0, A1, true);
  int u = 1;
  for (; u + 1 < nunits; u += 2) {  // nunits is even: units 1 .. nunits-2 in pairs
    window(u, A1, A0, false);
    window(u + 1, A0, A1, false);
  }
  window(u, A1, A0, false);  // unit nunits-1
  mma(A1, B1, 1);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // the empty tail pieces: nothing may target LDS past the loop
}

}  // namespace mevi
