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
// x 256 B rows; wave (grp, wm, wn) owns 64 x 128 outputs = 2 x 4 accumulators; per 32-wide slab 48 MFMAs
// per wave; two LDS buffers, one barrier per slab.
#pragma once

#include "mfma_pp.h"

namespace mevi {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

constexpr int X3_QT = 256;       // B rows per workgroup
constexpr int X3_ROWS = 2 * BM + X3_QT;
constexpr int X3_LD = 32;        // floats (128 B) per LDS row: 64 B hi | 64 B lo, XOR-swizzled 16-byte slots
constexpr size_t x3_lds_bytes() { return (size_t)2 * X3_ROWS * X3_LD * sizeof(float); }

// Staging is LDS-DMA (global_load_lds_dwordx4): no staging registers, no ds_write.  One wave-instruction
// moves 1 KiB = 8 rows x 128 B; the LDS destination is linear (wave base + lane*16), so the bank-conflict
// swizzle is applied on the SOURCE address and again on the read (both sides or neither):
//   logical 16-byte piece c of row r lives in slot c ^ ((r >> 1) & 7).
// With 128-byte rows the 16 lanes of a ds_read_b128 group (16 consecutive rows, same logical piece) then
// cover all 16 slots of the 256-byte bank row -> conflict free.
//
// rowptr[i]: global base (float*) of LDS row 64*wave8 + 8*i + (lane>>3), i = 0..7, where LDS rows are
// A0[0,128) | A1[128,256) | B[256,512); rows pre-clamped.  All 8 waves compute: wave w8 = 4*grp + 2*wm + wn
// owns A rows 128*grp + 64*wm + [0,64) x B rows 128*wn + [0,128).
__device__ __forceinline__ void pp_mainloop_bf16x3(const float *const (&rowptr)[8], int nslab, float *lds,
                                                   f32x16 (&acc)[2][4]) {
  const int t = threadIdx.x;
  const int lane = t & 63;
  const int w8 = __builtin_amdgcn_readfirstlane(t >> 6);
  const int grp = w8 >> 2, wm = (w8 >> 1) & 1, wn = w8 & 1;
  const int lrow = lane & 31;
  const int half = lane >> 5;
  // source piece of this lane for DMA instruction i: LDS slot lane&7 of row 64*w8 + 8*i + (lane>>3)
  // holds logical piece (lane&7) ^ ((row>>1)&7); row>>1 & 7 = ((8*i + (lane>>3)) >> 1) & 7 = (4*i + (lane>>4)) & 7
  int srcpiece[8];
#pragma unroll
  for (int i = 0; i < 8; ++i) srcpiece[i] = ((lane & 7) ^ ((4 * i + (lane >> 4)) & 7)) * 4;  // float offset

  auto dma = [&](int s) {
    float *base = lds + (s & 1) * X3_ROWS * X3_LD + (64 * w8) * X3_LD;
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      __builtin_amdgcn_global_load_lds(rowptr[i] + s * 32 + srcpiece[i],
                                       (__attribute__((address_space(3))) void *)(base + 8 * i * X3_LD), 16, 0, 0);
    }
  };

  const int sw = (lrow >> 1) & 7;  // swizzle of this lane's fragment rows (row offsets are multiples of 32)
  auto compute = [&](int s) {
    const float *base = lds + (s & 1) * X3_ROWS * X3_LD;
    const float *pa = base + (grp * BM + 64 * wm + lrow) * X3_LD;
    const float *pb = base + (2 * BM + 128 * wn + lrow) * X3_LD;
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const int chi = ((2 * j + half) ^ sw) * 4, clo = ((4 + 2 * j + half) ^ sw) * 4;
      bf16x8 ah[2], al[2], bh[4], bl[4];
#pragma unroll
      for (int mi = 0; mi < 2; ++mi) {
        ah[mi] = *reinterpret_cast<const bf16x8 *>(pa + 32 * mi * X3_LD + chi);
        al[mi] = *reinterpret_cast<const bf16x8 *>(pa + 32 * mi * X3_LD + clo);
      }
#pragma unroll
      for (int ni = 0; ni < 4; ++ni) {
        bh[ni] = *reinterpret_cast<const bf16x8 *>(pb + 32 * ni * X3_LD + chi);
        bl[ni] = *reinterpret_cast<const bf16x8 *>(pb + 32 * ni * X3_LD + clo);
      }
#pragma unroll
      for (int ni = 0; ni < 4; ++ni)
#pragma unroll
        for (int mi = 0; mi < 2; ++mi) {
          acc[mi][ni] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al[mi], bh[ni], acc[mi][ni], 0, 0, 0);
          acc[mi][ni] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[mi], bl[ni], acc[mi][ni], 0, 0, 0);
          acc[mi][ni] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[mi], bh[ni], acc[mi][ni], 0, 0, 0);
        }
    }
  };

#pragma unroll
  for (int mi = 0; mi < 2; ++mi)
#pragma unroll
    for (int ni = 0; ni < 4; ++ni)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[mi][ni][r] = 0.f;

  dma(0);
  __syncthreads();  // drains the DMA (vmcnt(0)) and publishes slab 0
  for (int s = 0; s < nslab; ++s) {
    if (s + 1 < nslab) dma(s + 1);  // buffer (s+1)&1 was last read in iteration s-1 (barrier since)
    compute(s);
    __syncthreads();                // vmcnt(0) + barrier: slab s+1 landed, everyone done with slab s
  }
}

}  // namespace mevi
