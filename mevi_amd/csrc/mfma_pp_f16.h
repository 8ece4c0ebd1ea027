// f16 tile loop of the pre-filter (ip_topk.hip, ip_filter_h1_kernel).
//
// Operands are f16 images of the f32 rows, row-major, row = dimp halves (dimp = dim padded to 32):
// (dimp >= 96: a tile spans at least three 32-wide units, see h1_tile_stream)
//   docs    : f16((d - mu) * S_d)      mu = column mean of the shard, S_d one power of two for the shard
//   queries : f16(q * S_q)             S_q a power of two per query
//   acc = sum_k a_k * b_k              ONE v_mfma_f32_32x32x16_f16 per 16 k, f32 accumulate
//
// acc / (S_q S_d) APPROXIMATES the f32 dot product q.(d - mu) with the rigorous bound
//   |acc/(S_q S_d) - sum_k q_k (d_k - mu_k)| <= C1(dim) * ||q|| * ||d - mu||,
//   C1 = (2u + u^2) + 4 * dimp * 2^-24 + 2e-7,   u = 2^-11 (f16 unit roundoff)
// (both operands rounded once: (1+u)^2 - 1 per product, Cauchy-Schwarz over the row; products of two
// f16 are exact in f32; 4 * 2^-24 per accumulation step allows the matrix core to truncate instead of
// rounding; 2e-7 covers elements that fall below the f16 normal range -- with a row maximum scaled to
// [2^14, 2^15) those are < 2^-29 of it -- and the rounding of d - mu).  The result is only ever used to
// SELECT candidates that are then re-scored exactly (ip_topk.hip: rescore_kernel holds the proof).
//
// Geometry: 512 threads = 8 waves, all computing (staging is DMA); block tile 256 A rows x 256 B rows;
// wave w8 = 4*grp + 2*wm + wn owns A rows 128*grp + 64*wm + [0,64) x B rows 128*wn + [0,128) = 2 x 4
// accumulators of 32x32.  LDS rows are A0[0,128) | A1[128,256) | B[256,512).
//
// The K loop advances in UNITS of 32 k = 64 bytes per row = four 16-byte pieces (piece c = k 8c..8c+7); a
// unit is two MFMA k-steps j = 0,1, the lane (row, half) of a fragment reads piece 2j + half.
// Staging is LDS-DMA in its MUBUF form (buffer_load_dwordx4 ... lds): no staging registers, no ds_write,
// and -- unlike global_load_lds, which the compiler books on lgkmcnt as an out-of-order FLAT event,
// degrading every `s_waitcnt lgkmcnt(N)` of the fragment pipeline to lgkmcnt(0) -- it counts on vmcnt
// only.  One wave-instruction moves 1 KiB = 16 rows x 64 B; the LDS destination is linear (M0 base +
// lane*16), so the bank swizzle is applied on the SOURCE address and again on the read:
//   logical piece c of row r lives in slot c ^ ((r >> 2) & 3).
// ds_read_b128 is serviced in lane groups {0-3,12-15,20-27} {4-11,16-19,28-31} (+32): with 64-byte rows
// the 16 rows of a group then cover all 16 slots of the 256-byte bank line -> conflict free.
// A wave stages 64 rows of ONE operand per unit (4 pieces), so its buffer descriptor is wave-uniform:
// `src` = first row of the wave's operand tile (docs: waves 0-3, queries: waves 4-7), `src_bytes` = bytes
// from there to the end of the operand (rows past it read as 0 and are masked in the epilogue).
//
// Pipeline (per wave; the unit stream runs ACROSS tiles, g = stream position, u = unit within the tile;
// window W_g = barrier g-1 .. barrier g; F0/F1 = fragment register sets):
//   PA: ds_read F0 <- (g, j=0)   | 8 MFMA on F1 = (g-1, j=1) | 2 DMA pieces of stream unit g+3
//   PB: ds_read F1 <- (g, j=1)   | 8 MFMA on F0 = (g,   j=0) | 2 DMA pieces of stream unit g+3
//   s_waitcnt lgkmcnt(0) vmcnt(8); s_barrier   -- unit g+1 landed (units g+2, g+3 = 8 pieces stay in flight),
//                                                 every wave has finished READING unit g
// so an LDS round trip is covered by 8 MFMAs of the same wave, a DMA piece has two full windows to land
// and the memory queue is never drained.  Buffer (g+3)&3 = (g-1)&3 is free from barrier g-1 on.
// The workgroup is PERSISTENT: it walks a list of tiles, and the last three windows of a tile already stage
// the first three units of the next one, so the fill latency of a tile (a cold 4 us round trip through
// L2/HBM) is paid once per workgroup, not once per tile, and the staging of tile t+1 proceeds under the
// epilogue of tile t.  (Other vector-memory operations of the epilogue only ever make the counted wait
// stricter: vmcnt completes in order.)
#pragma once

#include "mfma_pp.h"

namespace mevi {

typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));

constexpr int H1_QT = 256;  // B rows per workgroup
constexpr int H1_ROWS = 2 * BM + H1_QT;
constexpr int H1_LD = 16;   // floats (64 B) per LDS row of one unit
constexpr int H1_NBUF = 4;  // units resident in LDS: one being read, three landing
constexpr int H1_UNIT = H1_ROWS * H1_LD;  // floats per unit buffer (32 KiB)
constexpr size_t h1_lds_bytes() { return (size_t)H1_NBUF * H1_UNIT * sizeof(float); }

// What one wave stages for a tile: the 64 rows [0, 64) from `src` on, rows `row_bytes` apart, valid up to
// `bytes` (wave-uniform).
struct H1Src {
  const void *src;
  unsigned int bytes;
};

// next(H1Src &) -> bool : fetch this wave's source of the workgroup's next tile (false: no more tiles); it is
//                         called one tile AHEAD of the tile being computed
// begin()               : the oldest fetched tile starts computing (load what its epilogue will need)
// emit(acc)             : epilogue of that tile (tiles complete in the order next() returned them)
// uoff(u)               : byte offset of unit u inside this wave's operand rows (wave-uniform; the filter's images
//                         are plain rows, u * 64; the split-precision GEMM walks (hi | lo) row halves, gemm_split.hip)
// nunits >= 3 (the images are padded to at least 96 k).
struct H1PlainUnits {
  __device__ __forceinline__ int operator()(int u) const { return u * 64; }
};
// unit-major images (ip_topk.hip: image_at): unit u of a 256-row block is one contiguous 16 KiB; rows 64 B apart (row_bytes = 64)
struct H1BlockedUnits {
  __device__ __forceinline__ int operator()(int u) const { return u * 16384; }
};
// ABL: ablation switches of tools/probes/stream_probe.hip (the product instantiates 0): 1 = no s_barrier, 2 = no DMA,
// 4 = no LDS fragment reads, 8 = no MFMA -- wrong results, used to attribute the loop's time.
// NBUF: unit buffers in LDS (NBUF - 1 units in flight while one is read); 4 = 128 KiB (the filter, which keeps its
// candidate stash behind them), 5 = 160 KiB.  nunits >= NBUF - 1.
template <class Next, class Begin, class Emit, class UOff = H1PlainUnits, int ABL = 0, int NBUF = H1_NBUF>
__device__ __forceinline__ void h1_tile_stream(int row_bytes, int nunits, float *lds, Next next, Begin begin,
                                               Emit emit, UOff uoff = UOff()) {
  static_assert(NBUF == 4 || NBUF == 5, "vmcnt immediates below are written for 4 and 5 buffers");
  constexpr int DEPTH = NBUF - 1;
  const int t = threadIdx.x;
  const int lane = t & 63;
  const int w8 = __builtin_amdgcn_readfirstlane(t >> 6);
  const int grp = w8 >> 2, wm = (w8 >> 1) & 1, wn = w8 & 1;
  const int lrow = lane & 31;
  const int half = lane >> 5;
  // DMA piece i of this wave fills LDS rows 64*w8 + 16*i + (lane>>2), slot lane&3, which holds logical piece
  // c = (lane&3) ^ ((row>>2)&3), (row>>2)&3 = (lane>>4)&3
  const int cpiece = (lane & 3) ^ ((lane >> 4) & 3);
  int voff[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) voff[i] = (64 * (w8 & 3) + 16 * i + (lane >> 2)) * row_bytes + cpiece * 16;

  H1Src cur, nxt;
  if (!next(cur)) return;  // no tile for this workgroup (uniform)
  bool have_nxt = next(nxt);
  if (!have_nxt) nxt.bytes = 0u, nxt.src = cur.src;

  // pieces [p0, p0+2) of unit u of tile `s`, into stream buffer gb; an exhausted stream has bytes = 0 (empty
  // descriptor, no fetch), so the number of pieces in flight -- what the counted vmcnt relies on -- is the
  // same in every window
  auto dma2 = [&](const H1Src &s, int u, int gb, int p0) {
    if constexpr (ABL & 2) return;
    const __amdgpu_buffer_rsrc_t rsrc =
        __builtin_amdgcn_make_buffer_rsrc(const_cast<void *>(s.src), 0, (int)s.bytes, 0x00020000);
    float *base = lds + gb * H1_UNIT + (64 * w8) * H1_LD;   // gb = buffer index
#pragma unroll
    for (int i = p0; i < p0 + 2; ++i)
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, (__attribute__((address_space(3))) void *)(base + 16 * i * H1_LD), 16,
                                               voff[i], uoff(u), 0, 0);
  };

  const int sw = (lrow >> 2) & 3;  // fragment rows are lrow + multiples of 32
  int cj[2];
#pragma unroll
  for (int j = 0; j < 2; ++j) cj[j] = ((2 * j + half) ^ sw) * 4;  // float offset of this lane's piece, k-step j
  const int offa = (grp * BM + 64 * wm + lrow) * H1_LD;
  const int offb = (2 * BM + 128 * wn + lrow) * H1_LD;
  struct Frag {
    f16x8 a[2], b[4];
  };
  auto read = [&](int gb, int j, Frag &f) {
    if constexpr (ABL & 4) return;
    const float *p = lds + gb * H1_UNIT + cj[j];
#pragma unroll
    for (int mi = 0; mi < 2; ++mi) f.a[mi] = *reinterpret_cast<const f16x8 *>(p + offa + 32 * mi * H1_LD);
#pragma unroll
    for (int ni = 0; ni < 4; ++ni) f.b[ni] = *reinterpret_cast<const f16x8 *>(p + offb + 32 * ni * H1_LD);
  };
  f32x16 acc[2][4];
  auto mma = [&](const Frag &f) {
    if constexpr (ABL & 8) return;
#pragma unroll
    for (int ni = 0; ni < 4; ++ni)
#pragma unroll
      for (int mi = 0; mi < 2; ++mi)
        acc[mi][ni] = __builtin_amdgcn_mfma_f32_32x32x16_f16(f.a[mi], f.b[ni], acc[mi][ni], 0, 0, 0);
  };

  Frag F0, F1;
  if constexpr (ABL & 4) {   // fragments never loaded: give them defined contents
#pragma unroll
    for (int i = 0; i < 2; ++i) F0.a[i] = F1.a[i] = f16x8{1, 1, 1, 1, 1, 1, 1, 1};
#pragma unroll
    for (int i = 0; i < 4; ++i) F0.b[i] = F1.b[i] = f16x8{1, 1, 1, 1, 1, 1, 1, 1};
  }
  int rb = 0;  // buffer of the unit being computed (stream position mod NBUF)

  auto window = [&](int u, bool first) {
    // stream unit g+DEPTH: unit u+DEPTH of this tile, or unit u+DEPTH-nunits of the next one; its buffer is the one
    // before rb in the ring (free since the last barrier)
    const bool spill = u + DEPTH >= nunits;
    H1Src tgt;
    tgt.src = spill ? nxt.src : cur.src;
    tgt.bytes = spill ? nxt.bytes : cur.bytes;
    const int tu = spill ? u + DEPTH - nunits : u + DEPTH;
    const int wb = rb == 0 ? NBUF - 1 : rb - 1;
    read(rb, 0, F0);
    if (!first) mma(F1);
    dma2(tgt, tu, wb, 0);
    if (!first) {
      __builtin_amdgcn_sched_group_barrier(0x100, 6, 0);
      __builtin_amdgcn_sched_group_barrier(0x008, 2, 0);
      __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);
      __builtin_amdgcn_sched_group_barrier(0x008, 4, 0);
      __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);
      __builtin_amdgcn_sched_group_barrier(0x008, 2, 0);
    }
    __builtin_amdgcn_sched_barrier(0);
    read(rb, 1, F1);
    mma(F0);
    dma2(tgt, tu, wb, 2);
    // first MFMA ahead of the reads: its wait covers the fragments issued a group ago, not these
    __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
    __builtin_amdgcn_sched_group_barrier(0x100, 6, 0);
    __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
    __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);
    __builtin_amdgcn_sched_group_barrier(0x008, 4, 0);
    __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);
    __builtin_amdgcn_sched_group_barrier(0x008, 2, 0);
    __builtin_amdgcn_sched_barrier(0);
    // unit g+1 landed once at most the 8 pieces of units g+2, g+3 are outstanding; own reads of unit g done
    if constexpr (ABL & 1) asm volatile("s_waitcnt vmcnt(8) lgkmcnt(0)" ::: "memory");
    else if constexpr (NBUF == 5) asm volatile("s_waitcnt vmcnt(12) lgkmcnt(0)\n\ts_barrier" ::: "memory");
    else asm volatile("s_waitcnt vmcnt(8) lgkmcnt(0)\n\ts_barrier" ::: "memory");
    __builtin_amdgcn_sched_barrier(0);
    rb = rb == NBUF - 1 ? 0 : rb + 1;
  };

#pragma unroll
  for (int u = 0; u < DEPTH; ++u) {
    dma2(cur, u, u, 0);
    dma2(cur, u, u, 2);
  }
  if constexpr (NBUF == 5) asm volatile("s_waitcnt vmcnt(12)\n\ts_barrier" ::: "memory");
  else asm volatile("s_waitcnt vmcnt(8)\n\ts_barrier" ::: "memory");  // unit 0 landed
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
    for (int u = 1; u < nunits; ++u) window(u, false);
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
