// f16 tile loop of the pre-filter on v_mfma_f32_16x16x32_f16 (ip_topk.hip, ip_filter_h16_kernel): the persistent LDS-DMA
// unit stream of mfma_pp_f16.h -- same tiles, same unit-major images, same ring of 32 KiB units, same barrier per unit --
// with the 16 x 16 x 32 shape of the matrix instruction.
//
// Why: on this part a bf16 / f16 MFMA loop on random data runs at the clock the chip holds under load, and that clock depends
// on the shape: the 16x16x32 loop delivers ~1.12-1.15x the FLOP/s of the 32x32x16 loop at equal cycles per FLOP
// (/opt/skills/guides/MI355X_MICROARCH.md, "DVFS give-back" item 7).  One instruction covers the whole 32 k of a unit,
// so a unit is ONE fragment set: 4 fragments of 16 corpus rows, 8 fragments of 16 queries, 32 MFMAs of 16 cycles
// (= the 16 MFMAs of 32 cycles of the 32x32x16 form).  LDS bytes read per unit are the same (the wave tile is the same
// 64 x 128), the accumulators are the same 128 registers.
//
// Fragment layout (A and B alike): lane l holds row (l & 15) of the fragment's 16 rows, k = 8 (l >> 4) .. + 7 = the 16-byte
// piece l >> 4 of the row's 64-byte unit.  A ds_read_b128 is serviced in lane groups {0-3, 12-15, 20-27}, {4-11, 16-19,
// 28-31} (+ 32): rows 0-3 and 12-15 with piece p, rows 4-11 with piece p + 1.  Logical piece c of row r therefore lives in
// slot c ^ f((r >> 2) & 3), f = (0, 0, 3, 3): the four row quads of a group land in the four different slots, the four
// rows of a quad in the four 64-byte bank sets -> conflict free.  (The 32x32x16 loop's swizzle is f = identity; the DMA
// applies f on the SOURCE address, the images in memory are the same.)
//
// C/D layout of one 16 x 16 block: lane l holds column (l & 15) = the query, rows 4 (l >> 4) + j, j = 0..3 = four
// consecutive corpus rows in the four registers of the block.
//
// Pipeline per wave and window (unit g; A0 / A1 = the two register sets of the corpus fragments, alternating per unit;
// BL / BH = query fragments 0-3 / 4-7):
//   PA: ds_read A[g & 1], BL <- unit g (8 reads) | 16 MFMA  A[(g-1) & 1] x BH (unit g-1) | 2 DMA pieces of unit g+3
//   PB: ds_read BH <- unit g (4 reads)           | 16 MFMA  A[g & 1] x BL                 | 2 DMA pieces of unit g+3
//   s_waitcnt lgkmcnt(0) vmcnt(8); s_barrier
// Fragment registers: 2 x 16 + 16 + 16 = 64.  The unit count of a tile must be EVEN (the A sets alternate at compile
// time): images are padded to a multiple of 64 k.
#pragma once

#include <type_traits>

#include "mfma_pp_f16.h"

namespace mevi {

typedef float f32x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ int h16_swz(int quad) { return (quad & 2) ? 3 : 0; }

// next / begin / emit / uoff as h1_tile_stream (begin() is called two windows before the tile's emit, not at its start); emit receives f32x4 acc[4][NI]: block (mi, ni) = corpus rows 16 mi + [0, 16)
// of the wave's 64, queries 16 ni + [0, 16) of the wave's 16 NI.  nunits even, >= 4, and >= DEPTH (= NBUF - 1: 3 / 4 / 5 at
// NI = 8 / 4 / 2): a window's look-ahead reaches into the NEXT tile only, never past it -- the host picks NI accordingly
// (ip_topk.hip::run_pass; a 128-k image has four units, so NI = 2 is not offered there).
//
// NI = 8: the 256-query tile above.  NI = 4 / 2 (round 5): query tiles of 128 / 64 for searches of 33 .. 128 queries
// (faiss_search.profile's larger batches): such a search is bound by streaming the corpus image, and in the 256-query tile
// three quarters of its matrix work, fragment reads and query staging multiply rows that do not exist.  Same ring, same unit
// images; a wave owns 64 corpus rows x 16 NI queries and a window is ONE product group:
//   window g:  ds_read A[g & 1], B[g & 1] <- unit g (4 + NI reads) | 4 NI MFMA  A[(g-1) & 1] x B[(g-1) & 1] | 4 DMA pieces of unit g+3
// Of the query-side waves only those whose 64 rows exist in the tile stage anything.  An accumulator sums its 32-k products in
// the same order from a zero-C first product, so the raw scores -- the keys -- are the 256-query tile's, bit for bit.
template <int NI = 8, class Next, class Begin, class Emit, class UOff = H1PlainUnits, int ABL = 0>
__device__ __forceinline__ void h16_tile_stream(int row_bytes, int nunits, float *lds, Next next, Begin begin, Emit emit,
                                                UOff uoff = UOff()) {
  static_assert(NI == 8 || NI == 4 || NI == 2, "query blocks per wave");
  constexpr int NA = NI == 8 ? 4 : NI;   // query fragments per read / product group
  // NI < 8: a unit holds the 256 corpus rows + the tile's 32 NI query rows only (20 / 24 KiB instead of 32), so the same 128 KiB
  // take six / five units: five / four in flight instead of three -- what a search bound by the corpus stream is short of
  constexpr int UNIT = NI == 8 ? H1_UNIT : (256 + 32 * NI) * H1_LD;
  constexpr int NBUF = NI == 8 ? H1_NBUF : (NI == 4 ? 5 : 6), DEPTH = NBUF - 1;
  static_assert((size_t)NBUF * UNIT * 4 <= h1_lds_bytes(), "ring inside the kernel's LDS");
  auto wait_units = [](auto barrier) {       // all but the last DEPTH - 1 units' pieces (four per unit and wave) have landed
    constexpr bool B = decltype(barrier)::value;
    if constexpr (DEPTH == 3) { if constexpr (B) asm volatile("s_waitcnt vmcnt(8) lgkmcnt(0)\n\ts_barrier" ::: "memory"); else asm volatile("s_waitcnt vmcnt(8)\n\ts_barrier" ::: "memory"); }
    else if constexpr (DEPTH == 4) { if constexpr (B) asm volatile("s_waitcnt vmcnt(12) lgkmcnt(0)\n\ts_barrier" ::: "memory"); else asm volatile("s_waitcnt vmcnt(12)\n\ts_barrier" ::: "memory"); }
    else { if constexpr (B) asm volatile("s_waitcnt vmcnt(16) lgkmcnt(0)\n\ts_barrier" ::: "memory"); else asm volatile("s_waitcnt vmcnt(16)\n\ts_barrier" ::: "memory"); }
  };
  const int t = threadIdx.x;
  const int lane = t & 63;
  const int w8 = __builtin_amdgcn_readfirstlane(t >> 6);
  const int grp = w8 >> 2, wm = (w8 >> 1) & 1, wn = w8 & 1;
  const int r16 = lane & 15, kq = lane >> 4;
  // DMA piece i of this wave fills LDS rows 64*w8 + 16*i + (lane>>2), slot lane&3, which holds logical piece
  // c = (lane&3) ^ f((row>>2)&3), (row>>2)&3 = (lane>>4)&3
  const int cpiece = (lane & 3) ^ h16_swz((lane >> 4) & 3);
  int voff[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) voff[i] = (64 * (w8 & 3) + 16 * i + (lane >> 2)) * row_bytes + cpiece * 16;

  H1Src cur, nxt;
  if (!next(cur)) return;
  bool have_nxt = next(nxt);
  if (!have_nxt) nxt.bytes = 0u, nxt.src = cur.src;

  const bool stages = w8 < 4 || 64 * (w8 & 3) < 32 * NI;   // query-side waves beyond the tile's 32 NI rows stage nothing
  auto dma2 = [&](const H1Src &s, int u, int gb, int p0) {
    if constexpr (ABL & 2) return;
    if (NI != 8 && !stages) return;
    const __amdgpu_buffer_rsrc_t rsrc =
        __builtin_amdgcn_make_buffer_rsrc(const_cast<void *>(s.src), 0, (int)s.bytes, 0x00020000);
    float *base = lds + gb * UNIT + (64 * w8) * H1_LD;
#pragma unroll
    for (int i = p0; i < p0 + 2; ++i)
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, (__attribute__((address_space(3))) void *)(base + 16 * i * H1_LD), 16,
                                               voff[i], uoff(u), 0, 0);
  };

  const int cj = (kq ^ h16_swz((r16 >> 2) & 3)) * 4;  // float offset of this lane's piece inside its row
  const int offa = (grp * BM + 64 * wm + r16) * H1_LD + cj;
  const int offb = (2 * BM + 16 * NI * wn + r16) * H1_LD + cj;
  struct FragA {
    f16x8 a[4];
  };
  struct FragB {
    f16x8 b[4];
  };
  auto read_a = [&](int gb, FragA &f) {
    if constexpr (ABL & 4) return;
    const float *p = lds + gb * UNIT + offa;
#pragma unroll
    for (int mi = 0; mi < 4; ++mi) f.a[mi] = *reinterpret_cast<const f16x8 *>(p + 16 * mi * H1_LD);
  };
  auto read_b = [&](int gb, int hi, FragB &f) {
    if constexpr (ABL & 4) return;
    const float *p = lds + gb * UNIT + offb + 64 * hi * H1_LD;
#pragma unroll
    for (int ni = 0; ni < NA; ++ni) f.b[ni] = *reinterpret_cast<const f16x8 *>(p + 16 * ni * H1_LD);
  };
  f32x4 acc[4][NI];
  // zero: the first product of a tile into these sixteen blocks -- the instruction's C operand is the constant 0, so the 128
  // accumulator registers are never cleared by vector moves (128 v_mov per wave and tile otherwise)
  auto mma = [&](const FragA &fa, const FragB &fb, int hi, auto zero) {
    if constexpr (ABL & 8) return;
    const f32x4 z = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int ni = 0; ni < NA; ++ni)
#pragma unroll
      for (int mi = 0; mi < 4; ++mi)
        acc[mi][4 * hi + ni] = __builtin_amdgcn_mfma_f32_16x16x32_f16(fa.a[mi], fb.b[ni], decltype(zero)::value ? z : acc[mi][4 * hi + ni], 0, 0, 0);
  };
  using No = std::integral_constant<bool, false>;

  FragA A0, A1;
  FragB BL, BH;
  if constexpr (ABL & 4) {
    const f16x8 one = {1, 1, 1, 1, 1, 1, 1, 1};
#pragma unroll
    for (int i = 0; i < 4; ++i) A0.a[i] = A1.a[i] = BL.b[i] = BH.b[i] = one;
  }
  int rb = 0;

  // one window: unit u of the tile, corpus fragments into `An` (the previous unit's are in `Ap`).  kind 0: the tile's first
  // window (no product pending; its PB is the first into blocks 0-3), 1: the second (its PA is the first into blocks 4-7), 2: the rest
  auto window = [&](int u, auto kind, FragA &An, const FragA &Ap) {
    constexpr int KIND = decltype(kind)::value;
    constexpr bool first = KIND == 0;
    const bool spill = u + DEPTH >= nunits;
    H1Src tgt;
    tgt.src = spill ? nxt.src : cur.src;
    tgt.bytes = spill ? nxt.bytes : cur.bytes;
    const int tu = spill ? u + DEPTH - nunits : u + DEPTH;
    const int wb = rb == 0 ? NBUF - 1 : rb - 1;
    read_a(rb, An);
    read_b(rb, 0, BL);
    if constexpr (!first) mma(Ap, BH, 1, std::integral_constant<bool, KIND == 1>());
    dma2(tgt, tu, wb, 0);
    if constexpr (!first) {
      // 8 reads spread over the first half of the 16 MFMAs, the DMA pieces after them
      __builtin_amdgcn_sched_group_barrier(0x100, 4, 0);
      __builtin_amdgcn_sched_group_barrier(0x008, 3, 0);
      __builtin_amdgcn_sched_group_barrier(0x100, 4, 0);
      __builtin_amdgcn_sched_group_barrier(0x008, 3, 0);
      __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);
      __builtin_amdgcn_sched_group_barrier(0x008, 5, 0);
      __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);
      __builtin_amdgcn_sched_group_barrier(0x008, 5, 0);
    }
    __builtin_amdgcn_sched_barrier(0);
    read_b(rb, 1, BH);
    mma(An, BL, 0, std::integral_constant<bool, KIND == 0>());
    dma2(tgt, tu, wb, 2);
    // the first MFMAs ahead of the reads: their wait covers the fragments issued a phase ago, not these
    __builtin_amdgcn_sched_group_barrier(0x008, 2, 0);
    __builtin_amdgcn_sched_group_barrier(0x100, 4, 0);
    __builtin_amdgcn_sched_group_barrier(0x008, 4, 0);
    __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);
    __builtin_amdgcn_sched_group_barrier(0x008, 5, 0);
    __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);
    __builtin_amdgcn_sched_group_barrier(0x008, 5, 0);
    __builtin_amdgcn_sched_barrier(0);
    if constexpr (ABL & 1) asm volatile("s_waitcnt vmcnt(8) lgkmcnt(0)" ::: "memory");
    else asm volatile("s_waitcnt vmcnt(8) lgkmcnt(0)\n\ts_barrier" ::: "memory");
    __builtin_amdgcn_sched_barrier(0);
    rb = rb == NBUF - 1 ? 0 : rb + 1;
  };

  // NI < 8: one product group per window -- unit u's fragments into (An, Bn) while the previous unit's (Ap, Bp) multiply
  auto window_h = [&](int u, auto first_, auto zero_, FragA &An, const FragA &Ap, FragB &Bn, const FragB &Bp) {
    constexpr bool first = decltype(first_)::value;
    const bool spill = u + DEPTH >= nunits;
    H1Src tgt;
    tgt.src = spill ? nxt.src : cur.src;
    tgt.bytes = spill ? nxt.bytes : cur.bytes;
    const int tu = spill ? u + DEPTH - nunits : u + DEPTH;
    const int wb = rb == 0 ? NBUF - 1 : rb - 1;
    read_a(rb, An);
    read_b(rb, 0, Bn);
    if constexpr (!first) mma(Ap, Bp, 0, zero_);
    dma2(tgt, tu, wb, 0);
    dma2(tgt, tu, wb, 2);
    if constexpr (!first) {   // the reads in fours behind the first MFMAs, the DMA pieces after them
      __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
      __builtin_amdgcn_sched_group_barrier(0x100, 4, 0);
      __builtin_amdgcn_sched_group_barrier(0x008, 2, 0);
      __builtin_amdgcn_sched_group_barrier(0x100, NA, 0);
      __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
      __builtin_amdgcn_sched_group_barrier(0x020, 4, 0);
      __builtin_amdgcn_sched_group_barrier(0x008, 4 * NA - 4, 0);
    }
    __builtin_amdgcn_sched_barrier(0);
    wait_units(std::integral_constant<bool, true>());
    __builtin_amdgcn_sched_barrier(0);
    rb = rb == NBUF - 1 ? 0 : rb + 1;
  };

#pragma unroll
  for (int u = 0; u < DEPTH; ++u) {
    dma2(cur, u, u, 0);
    dma2(cur, u, u, 2);
  }
  wait_units(std::integral_constant<bool, false>());  // unit 0 landed
  __builtin_amdgcn_sched_barrier(0);
  while (true) {
    if constexpr (ABL & 8) {
#pragma unroll
      for (int mi = 0; mi < 4; ++mi)
#pragma unroll
        for (int ni = 0; ni < NI; ++ni) acc[mi][ni] = f32x4{0.f, 0.f, 0.f, 0.f};
    }
    if constexpr (NI != 8) {
      using Yes = std::integral_constant<bool, true>;
      window_h(0, Yes(), No(), A0, A1, BL, BH);           // (BL, BH serve as the two alternating query fragment sets)
      window_h(1, No(), Yes(), A1, A0, BH, BL);           // unit 0's product: the first into the blocks
      for (int u = 2; u < nunits; u += 2) {
        if (u + 2 >= nunits) begin();
        window_h(u, No(), No(), A0, A1, BL, BH);
        window_h(u + 1, No(), No(), A1, A0, BH, BL);
      }
      mma(A1, BH, 0, No());
    } else {
    window(0, std::integral_constant<int, 0>(), A0, A1);
    window(1, std::integral_constant<int, 1>(), A1, A0);
    for (int u = 2; u < nunits; u += 2) {
      if (u + 2 >= nunits) begin();   // two windows (~2 us) ahead of the epilogue: what begin() loads is not held in registers through the tile
      window(u, std::integral_constant<int, 2>(), A0, A1);
      window(u + 1, std::integral_constant<int, 2>(), A1, A0);
    }
    mma(A1, BH, 1, No());
    }
    emit(acc);
    if (!have_nxt) break;
    cur = nxt;
    have_nxt = next(nxt);
    if (!have_nxt) nxt.bytes = 0u;
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
}

}  // namespace mevi
