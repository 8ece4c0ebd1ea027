// Tile loop of the split-precision GEMM (gemm_split.hip) on v_mfma_f32_16x16x32_f16: the slab ring, the LDS-DMA schedule and
// the one barrier per 32 k of mfma_split_stream.h, with the 16 x 16 x 32 shape of the matrix instruction (on this part the chip
// holds a higher clock on it: the same tile loop delivers ~10 % more FLOP/s, tools/probes/mfma16_probe.hip and
// /opt/skills/guides/MI355X_MICROARCH.md "DVFS give-back" item 7).
//
// One instruction covers the 32 k of a super-unit, so per super-unit and wave there are SIX groups of 16 MFMAs (4 W fragments
// of 16 rows x 4 activation fragments of 16 rows; the wave's 128 activation rows in two halves h = 0, 1):
//   M1: a_lo(h0) x w_hi   M2: a_hi(h0) x w_hi   M3: a_hi(h0) x w_lo   M4: a_lo(h1) x w_hi   M5: a_hi(h1) x w_hi   M6: a_hi(h1) x w_lo
// Every output block therefore accumulates, per 32 k, a_lo w_hi, then a_hi w_hi, then a_hi w_lo -- the order
// gemm_split_skinny_kernel reproduces instruction for instruction (a row keeps its bits whatever batch it travels in).
//
// Registers: four fragment sets of 16 registers (al, ah: activations lo / hi of the current half; wh, wl), each loaded one
// group ahead of its first use and overwritten once its last use has been issued:
//   window s    P1: read al(h0), wh      | M6 of super-unit s-1      P4: read al(h1)  | M3
//               P2: read ah(h0)          | M1                        P5: read ah(h1)  | M4
//               P3: read wl              | M2                        P6: --           | M5
//   then `s_waitcnt vmcnt(4) lgkmcnt(0)` + barrier as in mfma_split_stream.h: all LDS reads of super-unit s happen inside
//   window s, its last group (M6) runs from registers under the next window's first reads.
// Fragment layout and bank swizzle as mfma_pp_f16x16.h: lane l holds row (l & 15), the 16-byte piece (l >> 4) of the row's
// 64-byte unit; logical piece c of row r lives in slot c ^ f((r >> 2) & 3), f = (0, 0, 3, 3).
// C/D layout of a block: lane l holds column (l & 15) = the activation row m, rows 4 (l >> 4) + j = four consecutive W rows =
// four consecutive output columns n: the epilogue moves 16 bytes per lane and block.
#pragma once

#include <type_traits>

#include "mfma_pp_f16x16.h"
#include "mfma_split_stream.h"

namespace mevi {

// next(H1Src &) / begin() / emit(acc) as split_tile_stream; emit receives f32x4 acc[4][NI]: block (mi, ni) = W rows 16 mi + [0, 16)
// of the wave's 64, activation rows 16 ni + [0, 16) of the wave's 16 NI.  `U` super-units per tile (>= 2).
//
// NI = 8 is the 256 x 256 tile described above.  NI = 4 / 2 (round 5): tiles of 128 / 64 ACTIVATION rows x 256 W rows for GEMMs
// whose 256-row tiles would leave most of the 256 CUs idle (a few thousand rows: the 873 queries a rank of the 8-GPU ensemble
// gets, the reference's own batches of 128) -- the same ring, DMA schedule and barrier; a wave owns 64 W rows x 16 NI activation
// rows and runs ONE half per window:
//   window s    P1: read al, wh | M3 of super-unit s-1 (a_hi x w_lo)      P2: read ah | M1 (a_lo x w_hi)      P3: read wl | M2 (a_hi x w_hi)
// An output block still accumulates, per 32 k, a_lo w_hi, then a_hi w_hi, then a_hi w_lo from a zero-C first product: a row has
// the bits it has in the 256-row tile (and in the latency kernel).  The a slabs keep their 256-row LDS slots; rows beyond the
// tile's 32 NI are outside the source descriptor (no traffic; zeros land in LDS and are never read).
template <int NI = 8, class Next, class Begin, class Emit>
__device__ __forceinline__ void split_tile_stream16(int row_bytes, int lo_bytes, int U, float *lds, Next next, Begin begin,
                                                    Emit emit, int unit_bytes = 64) {
  static_assert(NI == 8 || NI == 4 || NI == 2, "activation blocks per wave");
  constexpr int NA = NI == 8 ? 4 : NI;   // activation fragments per read / product group
  const int t = threadIdx.x;
  const int lane = t & 63;
  const int w8 = __builtin_amdgcn_readfirstlane(t >> 6);
  const int grp = w8 >> 2, wm = (w8 >> 1) & 1, wn = w8 & 1;
  const int r16 = lane & 15, kq = lane >> 4;
  const bool is_w = w8 < 4;
  const int cpiece = (lane & 3) ^ h16_swz((lane >> 4) & 3);
  int voff[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) voff[i] = (64 * (w8 & 3) + 16 * i + (lane >> 2)) * row_bytes + cpiece * 16;

  H1Src cur, nxt;
  if (!next(cur)) return;
  bool have_nxt = next(nxt);
  if (!have_nxt) nxt.bytes = 0u, nxt.src = cur.src;

  auto dma_slab = [&](int pos, int uu, bool lo) {
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

  const int cj = (kq ^ h16_swz((r16 >> 2) & 3)) * 4;
  const int offw = (grp * BM + 64 * wm + r16) * H1_LD + cj;  // w slabs: the wave's 64 W rows (4 fragments)
  const int offa = (16 * NI * wn + r16) * H1_LD + cj;        // a slabs: the wave's 16 NI activation rows (NI = 8: 2 x 4 fragments)
  struct Frag4 {
    f16x8 f[4];
  };
  // slab t of the super-unit whose slab 0 sits at ring position `base`: 0 a_lo, 1 w_hi, 2 a_hi, 3 w_lo
  auto read_w = [&](int base, int t_, Frag4 &f) {
    const float *p = lds + ring(base + t_) * SS_SLAB + offw;
#pragma unroll
    for (int i = 0; i < 4; ++i) f.f[i] = *reinterpret_cast<const f16x8 *>(p + 16 * i * H1_LD);
  };
  auto read_a = [&](int base, int t_, int h, Frag4 &f) {
    const float *p = lds + ring(base + t_) * SS_SLAB + offa + 64 * h * H1_LD;
#pragma unroll
    for (int i = 0; i < NA; ++i) f.f[i] = *reinterpret_cast<const f16x8 *>(p + 16 * i * H1_LD);
  };
  f32x4 acc[4][NI];
  auto mma = [&](const Frag4 &w, const Frag4 &a, int h, auto zero) {
    const f32x4 z = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int ni = 0; ni < NA; ++ni)
#pragma unroll
      for (int mi = 0; mi < 4; ++mi)
        acc[mi][4 * h + ni] =
            __builtin_amdgcn_mfma_f32_16x16x32_f16(w.f[mi], a.f[ni], decltype(zero)::value ? z : acc[mi][4 * h + ni], 0, 0, 0);
  };
  using Yes = std::integral_constant<bool, true>;
  using No = std::integral_constant<bool, false>;

  Frag4 AL, AH, WH, WL;
  int base = 0;  // ring position of slab 0 of the super-unit being read

  // one group: `reads` ds_read_b128 spread over the first MFMAs of the group's 16
  auto sched = [](auto reads) {
    constexpr int R = decltype(reads)::value;
    if constexpr (R == 8) {
      __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
      __builtin_amdgcn_sched_group_barrier(0x100, 4, 0);
      __builtin_amdgcn_sched_group_barrier(0x008, 3, 0);
      __builtin_amdgcn_sched_group_barrier(0x100, 4, 0);
      __builtin_amdgcn_sched_group_barrier(0x008, 12, 0);
    } else if constexpr (R == 4) {
      __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
      __builtin_amdgcn_sched_group_barrier(0x100, 4, 0);
      __builtin_amdgcn_sched_group_barrier(0x008, 15, 0);
    }
    __builtin_amdgcn_sched_barrier(0);
  };

  // a group of the half windows: `reads` ds_read_b128 (in fours) behind the first of the group's 4 NA MFMAs
  auto sched_h = [](auto reads) {
    constexpr int R = decltype(reads)::value, MM = 4 * NA;
    __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
    if constexpr (R > 4) {
      __builtin_amdgcn_sched_group_barrier(0x100, 4, 0);
      __builtin_amdgcn_sched_group_barrier(0x008, 3, 0);
      __builtin_amdgcn_sched_group_barrier(0x100, R - 4, 0);
      __builtin_amdgcn_sched_group_barrier(0x008, MM - 4, 0);
    } else {
      __builtin_amdgcn_sched_group_barrier(0x100, R, 0);
      __builtin_amdgcn_sched_group_barrier(0x008, MM - 1, 0);
    }
    __builtin_amdgcn_sched_barrier(0);
  };

  // The six groups of a window, written out (every fragment set has exactly one reader group after its load):
  auto run_window = [&](int u, auto first_) {
    constexpr bool first = decltype(first_)::value;
    dma_slab(ring(base + (is_w ? 7 : 6)), u + 1, is_w);
    dma_slab(ring(base + (is_w ? 9 : 8)), u + 2, !is_w);
    if constexpr (NI != 8) {
      // P1: read al, wh | M3 (previous super-unit): a_hi x w_lo
      read_a(base, 0, 0, AL);
      read_w(base, 1, WH);
      if constexpr (!first) {
        mma(WL, AH, 0, No());
        __builtin_amdgcn_sched_group_barrier(0x020, 8, 0);
        sched_h(std::integral_constant<int, 4 + NA>());
      } else {
        __builtin_amdgcn_sched_barrier(0);
      }
      // P2: read ah | M1: a_lo x w_hi
      read_a(base, 2, 0, AH);
      mma(WH, AL, 0, first_);
      sched_h(std::integral_constant<int, NA>());
      // P3: read wl | M2: a_hi x w_hi
      read_w(base, 3, WL);
      mma(WH, AH, 0, No());
      sched_h(std::integral_constant<int, 4>());
    } else {
    // P1: read al(h0), wh | M6 (previous super-unit): a_hi(h1) x w_lo
    read_a(base, 0, 0, AL);
    read_w(base, 1, WH);
    if constexpr (!first) {
      mma(WL, AH, 1, No());
      __builtin_amdgcn_sched_group_barrier(0x020, 8, 0);
      sched(std::integral_constant<int, 8>());
    } else {
      __builtin_amdgcn_sched_barrier(0);
    }
    // P2: read ah(h0) | M1: a_lo(h0) x w_hi
    read_a(base, 2, 0, AH);
    mma(WH, AL, 0, first_);
    sched(std::integral_constant<int, 4>());
    // P3: read wl | M2: a_hi(h0) x w_hi
    read_w(base, 3, WL);
    mma(WH, AH, 0, No());
    sched(std::integral_constant<int, 4>());
    // P4: read al(h1) | M3: a_hi(h0) x w_lo
    read_a(base, 0, 1, AL);
    mma(WL, AH, 0, No());
    sched(std::integral_constant<int, 4>());
    // P5: read ah(h1) | M4: a_lo(h1) x w_hi
    read_a(base, 2, 1, AH);
    mma(WH, AL, 1, first_);
    sched(std::integral_constant<int, 4>());
    // P6: -- | M5: a_hi(h1) x w_hi
    mma(WH, AH, 1, No());
    __builtin_amdgcn_sched_barrier(0);
    }
    asm volatile("s_waitcnt vmcnt(4) lgkmcnt(0)\n\ts_barrier" ::: "memory");
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
    begin();
    run_window(0, Yes());
    for (int u = 1; u < U; ++u) run_window(u, No());
    mma(WL, AH, NI == 8 ? 1 : 0, No());
    emit(acc);
    if (!have_nxt) break;
    cur = nxt;
    have_nxt = next(nxt);
    if (!have_nxt) nxt.bytes = 0u;
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // the empty tail pieces: nothing may target LDS past the loop
}

}  // namespace mevi
