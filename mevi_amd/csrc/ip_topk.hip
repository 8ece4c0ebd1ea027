// Dense arm: exact inner-product top-k over a row-major f32 corpus shard.
//
// Replaces faiss IndexFlat(IP).add/search at MEVI/faiss_search.py:13-21 and the
// in-cluster matmul+sort at MEVI/main_models.py:3967-3968,4012.
//
// Structure:
//   * ip_filter_kernel     -- exact path: f32 MFMA (v_mfma_f32_32x32x2_f32) tile GEMM, 2 x 128 docs x 128 queries
//     per workgroup (mfma_pp.h).  The score matrix is never written: the epilogue (emit_tile) compares every
//     accumulator with the query's running threshold tau[q] (its current k-th best score) and appends the
//     survivors to the query's candidate slots (count pass, one slot-allocating atomic per column, plain stores).
//   * ip_filter_h1_kernel  -- indexed path: the same filter on APPROXIMATE scores, one f16 MFMA per product on
//     centred, scaled f16 images (mfma_pp_f16.h); persistent workgroups, per-wave candidate stash.
//   * compact_kernel       -- per query: radix-selects the k largest of (current list + new candidates) in LDS
//     (64-bit score|id keys) and raises tau[q]; the lists are sorted once (sort_lists_kernel / after re-scoring).
//   * rescore_rows_kernel, rescore_finish_kernel -- exact f32 chains of the survivors, exact top-k and the
//     per-query proof that nothing outside the survivors can belong to it; unproven queries get a second, wider
//     f16 pass and then the exact path.
//   The corpus is walked in geometrically growing chunks so the expected number of candidates per chunk is a small
//   multiple of k per query; a query whose candidate slots overflow (adversarial row order) is flagged and
//   recomputed by the guaranteed path (chunk <= capacity).
//   * finalize_kernel      -- keys -> (f32 score, i64 id) with faiss-style padding; merge_kernel -- shard lists.
//
// Numerics: every returned score is the f32 fmaf chain over k = 0..dim-1 in order (MFMA lane half h supplies
// k = 2j + h); oracle/mevi_oracle.c computes the same chain on the CPU, so parity is bit-exact on both paths.

#include "mfma_pp_f16x16.h"

#include <float.h>
#include <math.h>
#include <string.h>
#include <vector>

namespace mevi {
namespace {

constexpr int MAX_SORT = 16384;  // largest LDS sort (128 KiB of keys)

struct TopkGeom {
  int k;    // results per query
  int S;    // per-query slots in `buf` (power of two): [0,k) top list, [k,S) candidates
  int cap;  // S - k
};

static inline int next_pow2(int x) {
  int p = 1;
  while (p < x) p <<= 1;
  return p;
}

// tuning hook (tools/bench_shard_sim.py sweeps it; unset in production): smallest per-query slot count
static inline int min_slots() {
  const char *e = getenv("MEVI_IP_TOPK_MIN_SLOTS");
  const int v = e ? atoi(e) : 0;
  return (v >= 256 && v <= MAX_SORT && (v & (v - 1)) == 0) ? v : 2048;
}

static inline TopkGeom make_geom(int k, int64_t nq = (int64_t)1 << 40) {
  TopkGeom g;
  g.k = k;
  int S = next_pow2(k) * 4;
  // a handful of queries (the streaming filter kernel): measured on a single query over 8.8 M rows (k = 100 / 1000), slots
  // 2048: 2.61 / -, 4096: 2.56 / 2.78 ms, 8192: 2.58 / 2.80, 16384: 2.69 / 2.87 -- few large chunks save launches but their
  // first chunk (as many rows as slots, every row a candidate) and the compaction of large areas cost more than they save
  if (nq <= 32) {
    S = 2 * next_pow2(k) > 4096 ? 2 * next_pow2(k) : 4096;
    // round 6: room for the sampled last launch (run_pass: ~3 k expected keys + 8 deviations need a candidate area of >= 2.8 k):
    // one query at top-1000 2.87 -> 2.74 ms, 32 queries 3.13 -> 2.94 (profiles/r06_sampled_threshold.txt)
    if ((double)(S - k) < 2.8 * (double)k) S *= 2;
    if (S > MAX_SORT) S = MAX_SORT;
    if (const char *e = getenv("MEVI_IP_TOPK_SMALL_SLOTS")) {  // tuning hook
      const int v = atoi(e);
      if (v >= 1024 && v <= MAX_SORT && (v & (v - 1)) == 0 && v >= 2 * next_pow2(k)) S = v;
    }
  }
  // short lists (the first round of a sharded search keeps ~k/W entries) still get 2048 slots: the chunk schedule grows
  // with cap / k, and every chunk costs a compaction launch whose time does not shrink with the shard
  if (S < min_slots()) S = min_slots();
  if (S > MAX_SORT) S = MAX_SORT;
  g.S = S;
  g.cap = S - k;
  return g;
}

// ---------------------------------------------------------------------------
// Per-wave stash of candidates (LDS) for the persistent f16 filter.  In the large chunks a wave tile holds a
// couple of candidates at most, yet handing out their slots costs a returning atomic -- a memory round trip
// with the matrix pipes idle -- on nearly every tile.  Sparse tiles therefore only APPEND (key, query) to the
// stash; it is flushed 64 entries at a time (one atomic round trip per 64 candidates instead of per tile),
// when it fills up and at the end of the kernel.
constexpr int STASH_N = 256;        // entries per wave
constexpr int STASH_TILE_MAX = 64;  // tiles with more candidates than this take the direct path
constexpr size_t STASH_BYTES_PER_WAVE = (size_t)STASH_N * 12;
struct Stash {
  unsigned long long *keys;  // [STASH_N]
  unsigned int *qs;          // [STASH_N]
  int n;                     // wave-uniform
};
__device__ __forceinline__ void stash_flush(Stash &st, unsigned long long *__restrict__ buf,
                                            unsigned int *__restrict__ count, int S, int k, int cap) {
  const int lane = threadIdx.x & 63;
  __builtin_amdgcn_wave_barrier();
  for (int base = 0; base < st.n; base += 64) {
    const int e = base + lane;
    if (e < st.n) {
      const unsigned int q = st.qs[e];
      const unsigned long long key = st.keys[e];
      const unsigned int slot = atomicAdd(&count[q], 1u);
      if (slot < (unsigned int)cap) buf[(size_t)q * S + k + slot] = key;
    }
  }
  __builtin_amdgcn_wave_barrier();
  st.n = 0;
}

// Threshold-filter epilogue of one wave tile (64 corpus rows x 32*NI queries, acc[2][NI]).
// C/D map of the 32x32 MFMA: col = lane&31 (query), row = (r&3) + 8*(r>>2) + 4*half (doc).
// Every score above the query's current threshold tq[ni] (= tau of column q0 + 32*ni + lane&31, +inf past
// nq; loaded by the caller, early) becomes a candidate key in the query's buffer.
// Slots are handed out by ONE atomic per query column and wave tile (the two lanes of a column pool
// their counts), all NI of them in flight together, instead of one returning atomic per candidate:
//   pass 1  per accumulator: maxima of its register quads (v_max3) -> quads no lane beats its threshold in are
//           skipped wave-wide; in the others count the passing elements per lane
//   atomics base slot per column
//   pass 2  store the keys of the quads that had any (plain stores)
template <int NI, bool STASH>
__device__ __forceinline__ void emit_tile_impl(f32x16 (&acc)[2][NI], const float (&tq)[NI], int q0, long long d0,
                                               long long doc_end, unsigned long long *__restrict__ buf,
                                               unsigned int *__restrict__ count, int S, int k, int cap,
                                               unsigned int id_base, Stash &stash_ref) {
  Stash *const stash = &stash_ref;
  const int lane = threadIdx.x & 63;
  const int lrow = lane & 31, half = lane >> 5;
  if (d0 + 64 > doc_end) {  // ragged last tile (wave-uniform): rows past the shard never pass
#pragma unroll
    for (int mi = 0; mi < 2; ++mi)
#pragma unroll
      for (int r = 0; r < 16; ++r)
        if (d0 + 32 * mi + (r & 3) + 8 * (r >> 2) + 4 * half >= doc_end) {
#pragma unroll
          for (int ni = 0; ni < NI; ++ni) acc[mi][ni][r] = -INFINITY;
        }
  }
  // pass 1: per accumulator the maxima of its four register quads (rows 8i .. 8i+3 of the lane's half) and,
  // where some lane beats its threshold, the number of passing elements per lane.  qmask: bit 4*(mi*NI+ni)+i
  // set when some lane of the wave passes in quad i of accumulator (mi, ni) -- a wave-uniform work list for
  // pass 2, which is then a handful of scalar tests instead of a vector compare per element.
  unsigned int n[NI], base[NI];
  unsigned int qmask = 0u;
  static_assert(2 * NI * 4 <= 32, "quad mask must fit one SGPR");
#pragma unroll
  for (int ni = 0; ni < NI; ++ni) {
    n[ni] = 0u;
#pragma unroll
    for (int mi = 0; mi < 2; ++mi) {
      const f32x16 &a = acc[mi][ni];
      float qm[4];
#pragma unroll
      for (int i = 0; i < 4; ++i) qm[i] = fmaxf(fmaxf(fmaxf(a[4 * i], a[4 * i + 1]), a[4 * i + 2]), a[4 * i + 3]);
      const float m = fmaxf(fmaxf(fmaxf(qm[0], qm[1]), qm[2]), qm[3]);
      if (__any(m > tq[ni])) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          if (__any(qm[i] > tq[ni])) {
            qmask |= 1u << (4 * (mi * NI + ni) + i);
#pragma unroll
            for (int r = 4 * i; r < 4 * i + 4; ++r) n[ni] += a[r] > tq[ni] ? 1u : 0u;
          }
        }
      }
    }
  }
  if (qmask == 0u) return;  // nothing passes anywhere in the wave tile
  // key = ord(score) << 32 | ~id;  ~(id0 + c) = ~id0 - c
  const unsigned int nid0 = 0xFFFFFFFFu - (id_base + (unsigned int)(d0 + 4 * half));
  if constexpr (STASH) {
    unsigned int tot = 0u;
#pragma unroll
    for (int ni = 0; ni < NI; ++ni) tot += n[ni];
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) tot += __shfl_xor(tot, off);
    if (tot <= (unsigned int)STASH_TILE_MAX) {  // sparse tile (wave-uniform): append, no atomics, no global stores
      if (stash->n + (int)tot > STASH_N) stash_flush(*stash, buf, count, S, k, cap);
#pragma unroll
      for (int ni = 0; ni < NI; ++ni) {
#pragma unroll
        for (int mi = 0; mi < 2; ++mi) {
          if (((qmask >> (4 * (mi * NI + ni))) & 15u) == 0u) continue;
#pragma unroll
          for (int i = 0; i < 4; ++i) {
            if (((qmask >> (4 * (mi * NI + ni) + i)) & 1u) == 0u) continue;
#pragma unroll
            for (int r = 4 * i; r < 4 * i + 4; ++r) {
              const float v = acc[mi][ni][r];
              const bool pass = v > tq[ni];
              const unsigned long long mask = __ballot(pass);
              if (mask != 0ull) {
                if (pass) {
                  const int pos = stash->n + (int)__builtin_amdgcn_mbcnt_hi((unsigned int)(mask >> 32),
                                                                             __builtin_amdgcn_mbcnt_lo((unsigned int)mask, 0u));
                  stash->keys[pos] = ((unsigned long long)f32_to_ord(v) << 32) |
                                     (unsigned long long)(nid0 - (unsigned int)(32 * mi + (r & 3) + 8 * (r >> 2)));
                  stash->qs[pos] = (unsigned int)(q0 + 32 * ni + lrow);
                }
                stash->n += __popcll(mask);
              }
            }
          }
        }
      }
      return;
    }
  }
#pragma unroll
  for (int ni = 0; ni < NI; ++ni) {
    const unsigned int other = __shfl_xor(n[ni], 32);
    base[ni] = 0u;
    if (half == 0 && n[ni] + other != 0u) base[ni] = atomicAdd(&count[q0 + 32 * ni + lrow], n[ni] + other);
    n[ni] = other;  // kept for the upper half's offset
  }
#pragma unroll
  for (int ni = 0; ni < NI; ++ni) {
    const unsigned int b = __shfl(base[ni], lrow);
    unsigned int slot = half == 0 ? b : b + n[ni];  // lower half's candidates first, then the upper half's
    unsigned long long *dst = buf + (size_t)(q0 + 32 * ni + lrow) * S + k;
#pragma unroll
    for (int mi = 0; mi < 2; ++mi) {
      if (((qmask >> (4 * (mi * NI + ni))) & 15u) == 0u) continue;
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        if (((qmask >> (4 * (mi * NI + ni) + i)) & 1u) == 0u) continue;
#pragma unroll
        for (int r = 4 * i; r < 4 * i + 4; ++r) {
          const float v = acc[mi][ni][r];
          if (v > tq[ni]) {
            if (slot < (unsigned int)cap)
              dst[slot] = ((unsigned long long)f32_to_ord(v) << 32) |
                          (unsigned long long)(nid0 - (unsigned int)(32 * mi + (r & 3) + 8 * (r >> 2)));
            ++slot;
          }
        }
      }
    }
  }
}

// the stash is taken BY REFERENCE (a pointer parameter kept the struct in scratch in the streaming kernel: 32 bytes per lane,
// a scratch load per use)
template <int NI>
__device__ __forceinline__ void emit_tile(f32x16 (&acc)[2][NI], const float (&tq)[NI], int q0, long long d0, long long doc_end,
                                          unsigned long long *__restrict__ buf, unsigned int *__restrict__ count, int S, int k,
                                          int cap, unsigned int id_base) {
  Stash none = {nullptr, nullptr, 0};
  emit_tile_impl<NI, false>(acc, tq, q0, d0, doc_end, buf, count, S, k, cap, id_base, none);
}
template <int NI>
__device__ __forceinline__ void emit_tile(f32x16 (&acc)[2][NI], const float (&tq)[NI], int q0, long long d0, long long doc_end,
                                          unsigned long long *__restrict__ buf, unsigned int *__restrict__ count, int S, int k,
                                          int cap, unsigned int id_base, Stash &stash) {
  emit_tile_impl<NI, true>(acc, tq, q0, d0, doc_end, buf, count, S, k, cap, id_base, stash);
}

// ---------------------------------------------------------------------------
// Epilogue of the 16 x 16 x 32 tile stream (mfma_pp_f16x16.h).  A wave tile is 4 x 8 blocks of 16 corpus rows x 16 queries; a
// lane holds, per block, FOUR CONSECUTIVE corpus rows (4 (lane >> 4) + j) of ONE query (lane & 15).
//
// What the 32x32x16 epilogue (emit_tile) costs is not its arithmetic but its SIZE: every element has its own copy of the
// append code (128 copies, ~50 KB of instructions), a candidate lands in a different copy each time, and the instruction
// cache misses (kernel trace by chunk: +600 cycles per candidate and wave between the 3-per-wave-tile and the 9-per-wave-tile
// chunk).  Here the per-block code is 5 instructions when nothing passes and ~12 when something does, whatever passes:
//   per block   m = max of the lane's four scores (v_max3 + v_max); vote m > tau[query]; no lane -> next block.
//               Lanes that pass write ONE 32-byte RECORD to the wave's LDS stash: their four raw scores + (tau, query, ~id
//               of the first row) -- which of the four pass is decided when the stash is flushed.
//   flush       (one copy of the code) a lane per record: count its passing scores, ONE returning atomic on the query's
//               slot counter for all of them, plain stores of the keys.  All eight waves flush on the same tiles (every
//               H16_FLUSH_EVERY-th, or when a stash is two thirds full), so their atomic round trips overlap instead of
//               stalling the workgroup's barrier eight times.
//   dense tiles (the first chunks, where most rows still pass): the stash overflows -> the tile's records are dropped, and
//               the tile takes the direct path: per query column one slot-allocating atomic for the four lanes that share
//               it, then plain stores -- as emit_tile.
// Keys and thresholds are the RAW accumulators, as in emit_tile.  An accumulator's low bits may differ between the two
// instruction shapes (the sums are taken in another order), so the candidate sets can differ in near-ties at tau -- every
// returned score comes from the exact re-scoring chains and every list is proven complete per query, never from here.
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
constexpr int H16_REC_N = 128;           // records per wave: 128 x (16 + 8) B = 3 KiB (STASH_BYTES_PER_WAVE)
constexpr int H16_FLUSH_AT = 80;         // flush before a tile when this many records are waiting
constexpr int H16_FLUSH_EVERY = 8;       // ... and on every 8th (4th, 2nd: the host sizes the cadence to the chunk's expected
                                         // candidate density, ip_search_pass) tile of the workgroup -- all waves together
static_assert((size_t)H16_REC_N * 24 <= STASH_BYTES_PER_WAVE, "record stash must fit the wave's stash area");
typedef unsigned int u32x2v __attribute__((ext_vector_type(2)));
typedef __attribute__((address_space(3))) f32x4 lds_f32x4;
typedef __attribute__((address_space(3))) u32x2v lds_u32x2;
struct RecStash {
  lds_f32x4 *vals;   // [H16_REC_N] the lane's four raw scores          (LDS pointers: 32 bits, ds_ instructions)
  lds_u32x2 *meta;   // [H16_REC_N] {query, ~id of row 0 of the four}   (the query's tau is re-read at the flush: it does not
                     //                                                   change during a launch)
  int n;             // wave-uniform
};

template <int NI>
__device__ __forceinline__ void load_tq16(float (&tq)[NI], const float *__restrict__ tau, int q0, int nq) {
  const int r16 = threadIdx.x & 15;
#pragma unroll
  for (int ni = 0; ni < NI; ++ni) {
    const int qi = q0 + 16 * ni + r16;
    tq[ni] = tau[qi < nq ? qi : nq - 1];     // branch-free (the caller masks columns >= nq to +inf)
  }
}

__device__ __forceinline__ void rec_flush(RecStash &st, const float *__restrict__ tau, unsigned long long *__restrict__ buf,
                                          unsigned int *__restrict__ count, int S, int k, int cap) {
  const int lane = threadIdx.x & 63;
  __builtin_amdgcn_wave_barrier();
  for (int base = 0; base < st.n; base += 64) {
    const int e = base + lane;
    if (e < st.n) {
      const f32x4 a = st.vals[e];
      const u32x2v mm = st.meta[e];
      const unsigned int q = mm[0];
      const unsigned int m[3] = {0u, q, mm[1]};
      const float t = tau[q];
      unsigned int c = 0u;
#pragma unroll
      for (int j = 0; j < 4; ++j) c += a[j] > t ? 1u : 0u;
      unsigned int slot = atomicAdd(&count[q], c);       // c >= 1: the lane wrote the record because one of the four passed
      unsigned long long *dst = buf + (size_t)q * S + k;
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        if (a[j] > t) {
          if (slot < (unsigned int)cap) dst[slot] = ((unsigned long long)f32_to_ord(a[j]) << 32) | (unsigned long long)(m[2] - (unsigned int)j);
          ++slot;
        }
      }
    }
  }
  __builtin_amdgcn_wave_barrier();
  st.n = 0;
}

template <int NI>
__device__ __forceinline__ void emit_tile16(f32x4 (&acc)[4][NI], const float (&tq)[NI], int q0, long long d0, long long doc_end,
                                            const float *__restrict__ tau, unsigned long long *__restrict__ buf,
                                            unsigned int *__restrict__ count, int S, int k, int cap, unsigned int id_base,
                                            RecStash &st, bool flush_now) {
  int lane = threadIdx.x & 63;
  asm volatile("" : "+v"(lane));   // opaque per tile: keeps the compiler from hoisting per-block invariants out of the tile loop (spills)
  const int r16 = lane & 15, kq = lane >> 4;
  if (d0 + 64 > doc_end) {  // ragged last tile (wave-uniform): rows past the shard never pass
#pragma unroll
    for (int mi = 0; mi < 4; ++mi)
#pragma unroll
      for (int j = 0; j < 4; ++j)
        if (d0 + 16 * mi + 4 * kq + j >= doc_end) {
#pragma unroll
          for (int ni = 0; ni < NI; ++ni) acc[mi][ni][j] = -INFINITY;
        }
  }
  if (st.n >= H16_FLUSH_AT || (flush_now && st.n > 0)) rec_flush(st, tau, buf, count, S, k, cap);
  // ~id of the lane's first row of block row 0: key = ord(score) << 32 | ~id;  ~(id0 + c) = ~id0 - c
  const unsigned int nid0 = 0xFFFFFFFFu - (id_base + (unsigned int)(d0 + 4 * kq));
  const int n0 = st.n;
  int n = n0;
#pragma unroll
  for (int mi = 0; mi < 4; ++mi)
#pragma unroll
    for (int ni = 0; ni < NI; ++ni) {
      const f32x4 a = acc[mi][ni];
      // v_max3 + v_max through asm: fmaxf() makes the compiler canonicalise every input first (five vector instructions per
      // block instead of two); a NaN accumulator cannot pass either way
      float m;
      asm("v_max3_f32 %0, %1, %2, %3\n\tv_max_f32 %0, %0, %4" : "=&v"(m) : "v"(a[0]), "v"(a[1]), "v"(a[2]), "v"(a[3]));
      const bool pass = m > tq[ni];
      const unsigned long long vote = __ballot(pass);
      if (vote != 0ull) {
        if (pass) {
          const int pos = n + (int)__builtin_amdgcn_mbcnt_hi((unsigned int)(vote >> 32),
                                                             __builtin_amdgcn_mbcnt_lo((unsigned int)vote, 0u));
          if (pos < H16_REC_N) {
            st.vals[pos] = a;
            st.meta[pos] = u32x2v{(unsigned int)(q0 + 16 * ni + r16), nid0 - (unsigned int)(16 * mi)};
          }
        }
        n += __popcll(vote);
      }
    }
  st.n = n;
  if (n <= H16_REC_N) return;
  // dense tile: forget its records, empty the stash, then per query column (16 ni + r16) the four lanes kq = 0..3 pool their
  // counts and lane kq = 0 takes the slots -- the eight columns' atomics are issued together (one round trip, not eight)
  st.n = n0;
  rec_flush(st, tau, buf, count, S, k, cap);
  unsigned int before[NI], base[NI];
#pragma unroll
  for (int ni = 0; ni < NI; ++ni) {
    unsigned int c = 0u;
#pragma unroll
    for (int mi = 0; mi < 4; ++mi)
#pragma unroll
      for (int j = 0; j < 4; ++j) c += acc[mi][ni][j] > tq[ni] ? 1u : 0u;
    const unsigned int c1 = __shfl_xor(c, 16), s01 = c + c1;       // pair (kq, kq ^ 1)
    const unsigned int s23 = __shfl_xor(s01, 32);                  // the other pair's total
    const unsigned int total = s01 + s23;
    base[ni] = 0u;
    if (kq == 0 && total != 0u) base[ni] = atomicAdd(&count[q0 + 16 * ni + r16], total);
    before[ni] = ((kq & 1) ? c1 : 0u) + ((kq & 2) ? s23 : 0u);    // lanes in kq order: the lower lane of the own pair, the whole lower pair
  }
#pragma unroll
  for (int ni = 0; ni < NI; ++ni) {
    unsigned int slot = __shfl(base[ni], r16) + before[ni];
    unsigned long long *dst = buf + (size_t)(q0 + 16 * ni + r16) * S + k;
#pragma unroll
    for (int mi = 0; mi < 4; ++mi)
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const float v = acc[mi][ni][j];
        if (v > tq[ni]) {
          if (slot < (unsigned int)cap)
            dst[slot] = ((unsigned long long)f32_to_ord(v) << 32) | (unsigned long long)(nid0 - (unsigned int)(16 * mi + j));
          ++slot;
        }
      }
  }
}

// thresholds of the NI query columns of this lane
template <int NI>
__device__ __forceinline__ void load_tq(float (&tq)[NI], const float *__restrict__ tau, int q0, int nq) {
  const int lrow = threadIdx.x & 31;
#pragma unroll
  for (int ni = 0; ni < NI; ++ni) {
    const int qi = q0 + 32 * ni + lrow;
    tq[ni] = qi < nq ? tau[qi] : INFINITY;
  }
}

// ip_filter_kernel: the shared f32-MFMA ping-pong tile loop (mfma_pp.h) with a threshold-
// filter epilogue.  A = corpus rows (two 128-row tiles per workgroup), B = 128 queries.
template <int NI, bool KTAIL>
__global__ __launch_bounds__(PP_THREADS, 2) void ip_filter_kernel(
    const float *__restrict__ Q, int nq, const float *__restrict__ D, long long doc_begin,
    long long doc_end, int dim, const float *__restrict__ tau, unsigned long long *__restrict__ buf,
    unsigned int *__restrict__ count, int S, int k, int cap, unsigned int id_base, int n_qtiles,
    int n_dpairs) {
  constexpr int QT = 64 * NI;
  extern __shared__ __attribute__((aligned(16))) float lds[];

  const int nwg = n_qtiles * n_dpairs;
  const int wg = xcd_remap(blockIdx.x, nwg);
  int dpair, qtile;
  supertile_order<4, 8>(wg, n_dpairs, n_qtiles, dpair, qtile);

  const int t = threadIdx.x;
  const int grp = __builtin_amdgcn_readfirstlane(t >> 8);
  const int tg = t & 255;
  const int wave = tg >> 6;
  const int wm = wave >> 1, wn = wave & 1;
  const int srow = tg >> 3, skq = (tg & 7) * 4;
  const long long drow0 = doc_begin + ((long long)dpair * 2 + grp) * BM;
  const int qrow0 = qtile * QT;

  const float *dptr[4];
  const float *qptr[NI];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    long long dr = drow0 + srow + 32 * i;
    if (dr > doc_end - 1) dr = doc_end - 1;  // clamp: masked in the epilogue
    dptr[i] = D + (size_t)dr * (size_t)dim + skq;
  }
#pragma unroll
  for (int i = 0; i < NI; ++i) {
    int qr = qrow0 + (QT / 2) * grp + srow + 32 * i;
    if (qr > nq - 1) qr = nq - 1;
    qptr[i] = Q + (size_t)qr * (size_t)dim + skq;
  }

  f32x16 acc[2][NI];
  pp_mainloop<NI, KTAIL>(dptr, qptr, dim, lds, acc);

  float tq[NI];
  load_tq<NI>(tq, tau, qrow0 + 32 * NI * wn, nq);
  emit_tile<NI>(acc, tq, qrow0 + 32 * NI * wn, drow0 + 64 * wm, doc_end, buf, count, S, k, cap, id_base);
}

__global__ __launch_bounds__(256) void init_state_kernel(unsigned long long *buf, unsigned int *count,
                                                        float *tau, unsigned int *failed, long long nq,
                                                        int S, int k) {
  const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  const long long total = nq * (long long)k;
  if (i < total) {
    const long long q = i / k;
    const int j = (int)(i - q * k);
    buf[q * S + j] = 0ull;
  }
  if (i < nq) {
    count[i] = 0u;
    tau[i] = -INFINITY;
    failed[i] = 0u;
  }
}

// One query: keep the k largest of its (k survivors + c new candidates) keys, refresh tau, reset the
// candidate count.  This is a SELECTION, not a sort: an MSB-first radix select (8-bit digits, LDS histogram)
// finds the k-th largest key T, the keys above it are written back in arbitrary order and T itself goes to
// slot k-1 (consumers read the k-th key there; nothing else relies on order -- the lists are sorted once, by
// sort_lists_kernel or after the exact re-scoring).  A bitonic sort of the 4096-key case cost 78 LDS-bound
// stages (1.3 ms per chunk for 6980 queries); the select reads the keys about five times.
// Keys are unique (score, id) pairs except the empty key 0, so "above T" are exactly k-1 keys when T != 0.
__device__ __forceinline__ void compact_query(unsigned long long *skeys, unsigned int *hist,
                                              unsigned long long *__restrict__ row, unsigned int *__restrict__ count,
                                              float *__restrict__ tau, unsigned int *__restrict__ failed, int q, int k,
                                              int cap) {
  const int t = threadIdx.x;
  const unsigned int c_raw = count[q];
  if (c_raw == 0u) return;  // nothing new for this query (uniform per workgroup)
  const int c = c_raw > (unsigned int)cap ? cap : (int)c_raw;
  const int n = k + c;
  for (int i = t; i < n; i += 256) skeys[i] = row[i];
  // hist[256..259]: prefix lo, prefix hi, rank still needed inside the prefix bucket, bucket size
  unsigned long long prefix = 0ull;
  unsigned int need = (unsigned int)k;
  int shift = 56;
  for (; shift >= 0; shift -= 8) {
    hist[t] = 0u;
    __syncthreads();
    for (int i = t; i < n; i += 256) {
      const unsigned long long key = skeys[i];
      if (shift == 56 || (key >> (shift + 8)) == prefix) atomicAdd(&hist[(unsigned int)(key >> shift) & 255u], 1u);
    }
    __syncthreads();
    if (t < 64) {  // wave 0: the digit d where the count of larger digits first reaches `need`
      const unsigned int h0 = hist[4 * t], h1 = hist[4 * t + 1], h2 = hist[4 * t + 2], h3 = hist[4 * t + 3];
      const unsigned int g = h0 + h1 + h2 + h3;
      unsigned int above = g;  // inclusive suffix sum over lanes >= t
#pragma unroll
      for (int off = 1; off < 64; off <<= 1) {
        const unsigned int o = __shfl_down(above, off);
        if (t + off < 64) above += o;
      }
      const unsigned int excl = above - g;  // keys in groups of larger digits
      if (excl < need && need <= above) {   // exactly one lane
        unsigned int cum = excl;
        int d = 4 * t + 3;
        unsigned int hd = h3;
        if (cum + h3 < need) { cum += h3; d = 4 * t + 2; hd = h2;
          if (cum + h2 < need) { cum += h2; d = 4 * t + 1; hd = h1;
            if (cum + h1 < need) { cum += h1; d = 4 * t; hd = h0; } } }
        hist[256] = (unsigned int)d;
        hist[257] = need - cum;  // rank inside bucket d
        hist[258] = hd;
      }
    }
    __syncthreads();
    prefix = (prefix << 8) | (unsigned long long)hist[256];
    need = hist[257];
    const unsigned int bucket = hist[258];
    __syncthreads();
    if (bucket == 1u) break;  // T is the only key with this prefix (uniform)
  }
  // T: all 64 bits fixed (shift < 0 after the last pass), or the unique key whose top bits equal `prefix`
  unsigned long long T = prefix;
  if (shift >= 0) {
    if (shift > 0) {
      for (int i = t; i < n; i += 256)
        if ((skeys[i] >> shift) == prefix) {
          hist[256] = (unsigned int)skeys[i];
          hist[257] = (unsigned int)(skeys[i] >> 32);
        }
      __syncthreads();
      T = ((unsigned long long)hist[257] << 32) | (unsigned long long)hist[256];
      __syncthreads();
    }
  }
  if (t == 0) hist[259] = 0u;
  __syncthreads();
  for (int i = t; i < n; i += 256) {
    const unsigned long long key = skeys[i];
    if (key > T) row[atomicAdd(&hist[259], 1u)] = key;
  }
  __syncthreads();
  const int m = (int)hist[259];  // k-1 if T != 0; the number of non-empty keys (< k) if T == 0
  for (int i = m + t; i < k; i += 256) row[i] = (i == k - 1) ? T : 0ull;
  if (t == 0) {
    tau[q] = (T != 0ull) ? key_score(T) : -INFINITY;
    count[q] = 0u;
    if (c_raw > (unsigned int)cap) failed[q] = 1u;
  }
  __syncthreads();  // skeys / hist are reused by the workgroup's next query
}

// Workgroups walk the queries with a grid stride (S * 8 + 2 KiB of LDS each).
__global__ __launch_bounds__(256) void compact_kernel(unsigned long long *__restrict__ buf,
                                                     unsigned int *__restrict__ count,
                                                     float *__restrict__ tau,
                                                     unsigned int *__restrict__ failed, int nq, int S, int k,
                                                     int cap) {
  extern __shared__ __attribute__((aligned(16))) unsigned long long skeys[];
  unsigned int *hist = reinterpret_cast<unsigned int *>(skeys + S);
  for (int q = blockIdx.x; q < nq; q += gridDim.x)
    compact_query(skeys, hist, buf + (size_t)q * S, count, tau, failed, q, k, cap);
}

// The same selection by ONE WAVE per query with the keys in REGISTERS (n = k + c <= 64 KPL keys, KPL per lane): no LDS, no
// workgroup barrier, no atomics.  compact_kernel spends ~23 us per query -- twenty barriers, LDS atomic chains and its global
// loads behind two to four resident workgroups per CU -- for a few thousand instructions of work.  Here the k-th largest key
// T is built bit by bit from the top (T |= bit whenever at least k keys are >= T | bit; the count is a ballot + popcount per
// register, no cross-lane reduction), stopping as soon as exactly k keys remain above the candidate (T is then their minimum).
// Same contract as compact_query: the keys above T in slots [0, m) in arbitrary order, T in slot k - 1, zeros between, tau,
// count reset, overflow flag.  Queries with more keys than fit are left untouched (count != 0) for compact_kernel, which is
// launched right after and skips every query whose count is already 0.
template <int KPL>
__global__ __launch_bounds__(256) void compact_wave_kernel(unsigned long long *__restrict__ buf, unsigned int *__restrict__ count,
                                                          float *__restrict__ tau, unsigned int *__restrict__ failed, int nq,
                                                          int S, int k, int cap) {
  const int lane = threadIdx.x & 63;
  const int q = blockIdx.x * 4 + (int)(threadIdx.x >> 6);
  if (q >= nq) return;
  const unsigned int c_raw = count[q];
  if (c_raw == 0u) return;  // nothing new for this query (wave-uniform)
  const int c = c_raw > (unsigned int)cap ? cap : (int)c_raw;
  const int n = k + c;
  if (n > 64 * KPL) return;  // compact_kernel takes it
  unsigned long long *row = buf + (size_t)q * S;
  unsigned long long key[KPL];
#pragma unroll
  for (int j = 0; j < KPL; ++j) key[j] = (64 * j + lane < n) ? row[64 * j + lane] : 0ull;
  const int nj = (n + 63) >> 6;   // registers that hold keys (wave-uniform); the rest are 0 and never count
  // the loops over the registers run in blocks of eight under a wave-uniform test: a `break` inside an unrolled loop sent the
  // key array to scratch (runtime-indexed), 528 bytes per lane
  unsigned long long T = 0ull;
  bool exact = false;              // exactly k keys >= T: T is not the k-th key yet, their minimum is
  for (int b = 63; b >= 0; --b) {
    const unsigned long long cand = T | (1ull << b);
    int cnt = 0;
#pragma unroll
    for (int jb = 0; jb < KPL / 8; ++jb)
      if (8 * jb < nj) {
#pragma unroll
        for (int jj = 0; jj < 8; ++jj) cnt += __popcll(__ballot(key[8 * jb + jj] >= cand));
      }
    if (cnt >= k) {
      T = cand;
      if (cnt == k) {
        exact = true;
        break;
      }
    }
  }
  if (exact) {
    unsigned long long mn = ~0ull;
#pragma unroll
    for (int jb = 0; jb < KPL / 8; ++jb)
      if (8 * jb < nj) {
#pragma unroll
        for (int jj = 0; jj < 8; ++jj) {
          const unsigned long long kk = key[8 * jb + jj];
          if (kk >= T && kk < mn) mn = kk;
        }
      }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
      const unsigned long long o = ((unsigned long long)(unsigned int)__shfl_xor((int)(mn >> 32), off) << 32) |
                                   (unsigned int)__shfl_xor((int)mn, off);
      mn = o < mn ? o : mn;
    }
    T = mn;
  }
  int base = 0;
#pragma unroll
  for (int jb = 0; jb < KPL / 8; ++jb)
    if (8 * jb < nj) {
#pragma unroll
      for (int jj = 0; jj < 8; ++jj) {
        const unsigned long long kk = key[8 * jb + jj];
        const bool pass = kk > T;
        const unsigned long long mask = __ballot(pass);
        if (pass)
          row[base + (int)__builtin_amdgcn_mbcnt_hi((unsigned int)(mask >> 32), __builtin_amdgcn_mbcnt_lo((unsigned int)mask, 0u))] = kk;
        base += __popcll(mask);
      }
    }
  for (int i = base + lane; i < k; i += 64) row[i] = (i == k - 1) ? T : 0ull;
  if (lane == 0) {
    tau[q] = (T != 0ull) ? key_score(T) : -INFINITY;
    count[q] = 0u;
    if (c_raw > (unsigned int)cap) failed[q] = 1u;
  }
}

// ---- sampled threshold (round 6) ------------------------------------------------------------------------------------------------
// The chunk schedule raises tau to the k-th best score SEEN SO FAR after every launch, so a pass appends ~k (g - 1) keys per
// query and launch -- 17 k over eight launches -- and the early launches are bound by those appends, not by their rows.  With
// exchangeable row order (the premise of the schedule's growth already) the rows seen so far are a SAMPLE of the corpus: the
// rank-r best score of `seen` rows estimates the rank r nd / seen score of all nd.  So once enough rows are in (r >= 96 ranks:
// +-10 %), rank_tau_kernel sets tau to the score of rank r = c k seen / nd (c ~ 3) and ONE launch takes all remaining rows,
// appending ~c k keys per query instead of ~k (g - 1) per launch of the rest of the schedule.  Exactness does not rest on the
// estimate: the launch drops only rows with score <= tau_s, so the list is the exact top-k iff the k-th best score afterwards is
// STRICTLY above tau_s (then at least k rows beat tau_s and every row that could belong was kept) -- sample_check_kernel flags
// every other query (and compact_* flags overflow as always); flagged queries take the guaranteed path like any overflow.
template <int KPL>   // tau == nullptr: only tau_s is written (the re-scoring's exclusion threshold: rescore_rows_kernel)
__global__ __launch_bounds__(256) void rank_tau_kernel(const unsigned long long *__restrict__ buf, int nq, int S, int k, int rank,
                                                      float *__restrict__ tau, float *__restrict__ tau_s) {
  const int lane = threadIdx.x & 63;
  const int q = blockIdx.x * 4 + (int)(threadIdx.x >> 6);
  if (q >= nq) return;
  const unsigned long long *row = buf + (size_t)q * S;
  unsigned long long key[KPL];
#pragma unroll
  for (int j = 0; j < KPL; ++j) key[j] = (64 * j + lane < k) ? row[64 * j + lane] : 0ull;
  const int nj = (k + 63) >> 6;
  unsigned long long T = 0ull;     // the largest T with at least `rank` keys >= T = the rank-th largest key
  for (int b = 63; b >= 0; --b) {
    const unsigned long long cand = T | (1ull << b);
    int cnt = 0;
#pragma unroll
    for (int jb = 0; jb < KPL / 8; ++jb)
      if (8 * jb < nj) {
#pragma unroll
        for (int jj = 0; jj < 8; ++jj) cnt += __popcll(__ballot(key[8 * jb + jj] >= cand));
      }
    if (cnt >= rank) T = cand;
  }
  if (lane == 0) {
    const float t = (T != 0ull) ? key_score(T) : -INFINITY;   // fewer than `rank` rows so far: no threshold, nothing is dropped
    tau_s[q] = t;
    if (tau) tau[q] = t;
  }
}

__global__ __launch_bounds__(256) void sample_check_kernel(const float *__restrict__ tau, const float *__restrict__ tau_s,
                                                          unsigned int *__restrict__ failed, int nq) {
  const int q = blockIdx.x * 256 + threadIdx.x;
  if (q < nq && !(tau[q] > tau_s[q]) && tau_s[q] > -INFINITY) failed[q] = 1u;   // (tau_s = -inf: nothing was dropped)
}

// Sort every query's list (row[0, k)) descending: the last step of a pass whose lists are read as ranked output.
__global__ __launch_bounds__(256) void sort_lists_kernel(unsigned long long *__restrict__ buf, int S, int k) {
  extern __shared__ __attribute__((aligned(16))) unsigned long long skeys[];
  const int t = threadIdx.x;
  unsigned long long *row = buf + (size_t)blockIdx.x * S;
  int P = 64;
  while (P < k) P <<= 1;
  for (int i = t; i < P; i += 256) skeys[i] = (i < k) ? row[i] : 0ull;
  __syncthreads();
  bitonic_sort_desc<256>(skeys, P, t);
  for (int i = t; i < k; i += 256) row[i] = skeys[i];
}

__global__ __launch_bounds__(256) void finalize_kernel(const unsigned long long *__restrict__ buf, int S,
                                                      int k, long long nq, float *__restrict__ out_score,
                                                      long long *__restrict__ out_id) {
  const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= nq * (long long)k) return;
  const long long q = i / k;
  const int j = (int)(i - q * k);
  const unsigned long long key = buf[q * S + j];
  if (key == 0ull) {
    out_score[i] = -FLT_MAX;
    out_id[i] = -1;
  } else {
    out_score[i] = key_score(key);
    out_id[i] = (long long)key_id(key);
  }
}

// rows of src selected by idx -> dst (float4 granularity)
__global__ __launch_bounds__(256) void gather_rows_kernel(const float *__restrict__ src,
                                                         const int *__restrict__ idx, int nrows, int dim,
                                                         float *__restrict__ dst) {
  const int r = blockIdx.x;
  if (r >= nrows) return;
  const float4 *s = reinterpret_cast<const float4 *>(src + (size_t)idx[r] * dim);
  float4 *d = reinterpret_cast<float4 *>(dst + (size_t)r * dim);
  for (int i = threadIdx.x; i < dim / 4; i += blockDim.x) d[i] = s[i];
}

// top-k rows of the fallback state -> rows idx[r] of the main state
__global__ __launch_bounds__(256) void scatter_top_kernel(const unsigned long long *__restrict__ src,
                                                         const int *__restrict__ idx, int nrows, int S,
                                                         int k, unsigned long long *__restrict__ dst, int dst_ld) {
  const int r = blockIdx.x;
  if (r >= nrows) return;
  const unsigned long long *s = src + (size_t)r * S;
  unsigned long long *d = dst + (size_t)idx[r] * dst_ld;
  for (int i = threadIdx.x; i < k; i += blockDim.x) d[i] = s[i];
}

// shard lists -> keys, one workgroup per query, sort, keep k_out
__global__ __launch_bounds__(256) void merge_kernel(const float *__restrict__ scores,
                                                   const long long *__restrict__ ids, int nlists,
                                                   long long nq, int k_in, int k_out,
                                                   float *__restrict__ out_score,
                                                   long long *__restrict__ out_id) {
  extern __shared__ __attribute__((aligned(16))) unsigned long long skeys[];
  const long long q = blockIdx.x;
  const int t = threadIdx.x;
  const int n = nlists * k_in;
  int P = 512;
  while (P < n) P <<= 1;
  for (int i = t; i < P; i += 256) {
    unsigned long long key = 0ull;
    if (i < n) {
      const int l = i / k_in, j = i - l * k_in;
      const size_t off = ((size_t)l * nq + q) * k_in + j;
      const long long id = ids[off];
      if (id >= 0) key = make_key(scores[off], (unsigned int)id);
    }
    skeys[i] = key;
  }
  __syncthreads();
  bitonic_sort_desc<256>(skeys, P, t);
  for (int i = t; i < k_out; i += 256) {
    const unsigned long long key = (i < P) ? skeys[i] : 0ull;
    const size_t off = (size_t)q * k_out + i;
    if (key == 0ull) {
      out_score[off] = -FLT_MAX;
      out_id[off] = -1;
    } else {
      out_score[off] = key_score(key);
      out_id[off] = (long long)key_id(key);
    }
  }
}

// (score, id) pairs <-> the 8-byte entries of the sharded search's all-gather: score bits << 32 | id as u32 (id -1 = padding
// travels as 0xFFFFFFFF)
__global__ __launch_bounds__(256) void pack_lists_kernel(const float *__restrict__ scores, const long long *__restrict__ ids,
                                                        long long n, unsigned long long *__restrict__ out) {
  const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
  if (i < n) out[i] = ((unsigned long long)__float_as_uint(scores[i]) << 32) | (unsigned long long)(unsigned int)ids[i];
}

// Merge of the all-gathered, PACKED shard lists with the proof of dense.merge_truncated, one workgroup per query.
// Every shard list arrives sorted (score desc, id asc), so the lists are loaded into LDS with alternating directions
// and only the LAST stages of the bitonic network run (sizes 2 Lp .. P: 30 compare-exchange stages for 8 lists of <= 256
// instead of the 66 of a full sort of 2048 keys).  unproven[q] = 1 when some shard's last entry reached the merged top-k
// (then deeper rows of that shard could belong to it: the caller repeats the query with full lists).
__global__ __launch_bounds__(256) void merge_packed_kernel(const unsigned long long *__restrict__ packed, int nlists, long long nq,
                                                          int k_in, int Lp, int k_out, int truncated,
                                                          float *__restrict__ out_score, long long *__restrict__ out_id,
                                                          unsigned char *__restrict__ unproven) {
  extern __shared__ __attribute__((aligned(16))) unsigned long long skeys[];
  __shared__ unsigned long long last_key[64];
  const long long q = blockIdx.x;
  const int t = threadIdx.x;
  int P = 2 * Lp;
  while (P < nlists * Lp) P <<= 1;
  for (int i = t; i < P; i += 256) {
    const int l = i / Lp, j0 = i - l * Lp;
    const int j = (l & 1) ? Lp - 1 - j0 : j0;  // odd lists ascending: every 2 Lp block is bitonic
    unsigned long long key = 0ull;
    if (l < nlists && j < k_in) {
      const unsigned long long e = packed[((size_t)l * nq + q) * k_in + j];
      const unsigned int id = (unsigned int)e;
      if (id != 0xFFFFFFFFu) key = make_key(__uint_as_float((unsigned int)(e >> 32)), id);
      if (j == k_in - 1 && l < 64) last_key[l] = key;
    }
    skeys[i] = key;
  }
  __syncthreads();
  for (int size = 2 * Lp; size <= P; size <<= 1)
    for (int stride = size >> 1; stride > 0; stride >>= 1) bitonic_stage<256, false>(skeys, P, size, stride, t);
  for (int i = t; i < k_out; i += 256) {
    const unsigned long long key = (i < P) ? skeys[i] : 0ull;
    const size_t off = (size_t)q * k_out + i;
    if (key == 0ull) {
      out_score[off] = -FLT_MAX;
      out_id[off] = -1;
    } else {
      out_score[off] = key_score(key);
      out_id[off] = (long long)key_id(key);
    }
  }
  if (t == 0) {
    // kth == 0: fewer than k rows exist in total -> every returned row is already in the list
    const unsigned long long kth = (k_out - 1 < P) ? skeys[k_out - 1] : 0ull;
    unsigned char u = 0;
    if (truncated && kth != 0ull)
      for (int l = 0; l < nlists && l < 64; ++l)
        if (last_key[l] != 0ull && last_key[l] >= kth) u = 1;
    unproven[q] = u;
  }
}

// ---------------------------------------------------------------------------
// f16 pre-filter (see mfma_pp_f16.h): centred + scaled f16 images, approximate filter, exact re-score + proof.

// Rounds 1-5 bounded |acc/(S_q S_d) - q.(d - mu)| by h1_c1 ||q|| ||d - mu||, h1_c1 = (2u + u^2) + 4 dimp 2^-24 + 2e-7 (derivation:
// mfma_pp_f16.h); round 6 keeps its accumulation part and MEASURES the rounding part (h1_err_bound below).
// the accumulation part of it alone (h1_err_bound: the rounding part is measured per query and per shard)
static inline float h1_cacc(int64_t dimp) { return (float)((4.0 * (double)dimp / 16777216.0 + 2e-7) * 1.001); }
// |chain_f32(q, d) - q.d| <= h1_c2 * ||q|| * ||d||   (n sequential fmaf: gamma_n = n u / (1 - n u), u = 2^-24)
static inline float h1_c2(int64_t dim) { return (float)((double)dim / 16777216.0 * 1.01); }

// power of two S with m * S in [2^14, 2^15) (m > 0), exponent clamped so S and 1/S stay finite normal floats
__device__ __forceinline__ float pow2_scale(float m) {
  if (!(m > 0.f)) return 1.f;
  int e;
  (void)frexpf(m, &e);  // m = f * 2^e, f in [0.5, 1)
  int s = 15 - e;
  s = s > 100 ? 100 : (s < -100 ? -100 : s);
  return ldexpf(1.f, s);
}

// column sums of the shard in f64 (mu = sum / n); 256 threads = 256 columns per pass, 512 rows per workgroup
__global__ __launch_bounds__(256) void colsum_kernel(const float *__restrict__ x, long long n, int dim,
                                                    double *__restrict__ sum) {
  const long long r0 = (long long)blockIdx.x * 512;
  const long long r1 = r0 + 512 < n ? r0 + 512 : n;
  for (int c = threadIdx.x; c < dim; c += 256) {
    double a = 0.0;
    for (long long r = r0; r < r1; ++r) a += (double)x[(size_t)r * dim + c];
    atomicAdd(&sum[c], a);
  }
}

__global__ __launch_bounds__(256) void mean_kernel(const double *__restrict__ sum, long long n, int dim, int dimp,
                                                  float *__restrict__ mu) {
  const int c = blockIdx.x * 256 + threadIdx.x;
  if (c < dimp) mu[c] = (c < dim && n > 0) ? (float)(sum[c] / (double)n) : 0.f;
}

// per row: ||d - mu|| (stored, rounded up), and shard maxima bits[0] = max ||d - mu||, bits[1] = max ||d||,
// bits[2] = max |d_k - mu_k|.  One wave per row, 16 rows per wave, one atomic per workgroup and maximum
// (every row hitting the same three words serialises the whole pass).
__global__ __launch_bounds__(256) void doc_stats_kernel(const float *__restrict__ x, long long n, int dim,
                                                       const float *__restrict__ mu, float *__restrict__ norms_c,
                                                       unsigned int *__restrict__ bits) {
  __shared__ float red[3][4];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const long long r0 = ((long long)blockIdx.x * 4 + wave) * 16;
  float mc = 0.f, mr = 0.f, mx = 0.f;
  for (long long r = r0; r < r0 + 16 && r < n; ++r) {
    const float *xr = x + (size_t)r * dim;
    float sc = 0.f, sr = 0.f;
    for (int k = lane * 4; k < dim; k += 256) {
      const float4 v = *reinterpret_cast<const float4 *>(xr + k);
      const float4 m = *reinterpret_cast<const float4 *>(mu + k);
      const float c0 = v.x - m.x, c1 = v.y - m.y, c2 = v.z - m.z, c3 = v.w - m.w;
      sc = fmaf(c0, c0, fmaf(c1, c1, fmaf(c2, c2, fmaf(c3, c3, sc))));
      sr = fmaf(v.x, v.x, fmaf(v.y, v.y, fmaf(v.z, v.z, fmaf(v.w, v.w, sr))));
      mx = fmaxf(fmaxf(mx, fmaxf(fabsf(c0), fabsf(c1))), fmaxf(fabsf(c2), fabsf(c3)));
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
      sc += __shfl_xor(sc, off);
      sr += __shfl_xor(sr, off);
    }
    const float nc = sqrtf(sc) * 1.00001f, nr = sqrtf(sr) * 1.00001f;  // round up: bounds must hold for the real norms
    if (lane == 0) norms_c[r] = nc;
    mc = fmaxf(mc, nc);
    mr = fmaxf(mr, nr);
  }
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) mx = fmaxf(mx, __shfl_xor(mx, off));
  if (lane == 0) red[0][wave] = mc, red[1][wave] = mr, red[2][wave] = mx;
  __syncthreads();
  if (threadIdx.x < 3) {
    const float v = fmaxf(fmaxf(red[threadIdx.x][0], red[threadIdx.x][1]), fmaxf(red[threadIdx.x][2], red[threadIdx.x][3]));
    atomicMax(&bits[threadIdx.x], __float_as_uint(v));  // non-negative floats order as uints
  }
}

__global__ void doc_scale_kernel(const unsigned int *__restrict__ bits, float *__restrict__ scal) {
  if (threadIdx.x == 0 && blockIdx.x == 0) {
    const float s = pow2_scale(__uint_as_float(bits[2]));
    scal[0] = s;
    scal[1] = 1.f / s;
  }
}

// The f16 images are UNIT-MAJOR: rows in blocks of 256 (one operand tile of the filter), and inside a block the 32 k of
// unit u of all 256 rows together -- element (r, k) at halves ((r / 256 * U + k / 32) * 256 + r % 256) * 32 + k % 32,
// U = dimp / 32.  A tile's unit is then one contiguous 16 KiB block and every LDS-DMA piece (16 rows x 64 B) 1 KiB of
// consecutive memory instead of 16 segments dimp * 2 bytes apart: the DMA, which bounds the loop, moves 49 -> 62 GB/s per
// CU (tools/probes/stream_probe.hip, profiles/r02_stream_ablation.txt).  Rows up to the end of the last block exist and are zero.
__device__ __forceinline__ size_t image_at(long long r, int k, int dimp) {
  return ((size_t)((r >> 8) * (dimp >> 5) + (k >> 5)) * 256 + (size_t)(r & 255)) * 32 + (k & 31);
}
inline int64_t image_rows(int64_t n) { return (n + 255) / 256 * 256; }

// docs -> f16((d - mu) * S_d), unit-major, zero padded (columns to dimp, rows to the end of the block).  One wave per row.
// Also MEASURES what the rounding did (round 6): bits[3] = max over rows of ||(d - mu) - image row / S_d|| -- the real distance
// between a centred row and its f16 image (f64 per element: the subtraction, the f32 rounding of d - mu, the f16 rounding and
// any underflow are all inside), rounded up.  The proof's bound uses it in place of the worst case u ||d - mu||.
__global__ __launch_bounds__(256) void split_docs_f16_kernel(const float *__restrict__ x, long long n, int dim, int dimp,
                                                            const float *__restrict__ mu,
                                                            const float *__restrict__ scal,
                                                            _Float16 *__restrict__ out, unsigned int *__restrict__ bits) {
  const long long r = (long long)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (r >= ((n + 255) >> 8 << 8)) return;
  const int lane = threadIdx.x & 63;
  const float *xr = x + (size_t)r * dim;
  const float s = scal[0];
  const double inv_s = (double)scal[1];
  typedef _Float16 h4 __attribute__((ext_vector_type(4)));
  double dd = 0.0;
  for (int k = lane * 4; k < dimp; k += 256) {
    _Float16 *o = out + image_at(r, k, dimp) - k;
    h4 h = {0, 0, 0, 0};
    if (k < dim && r < n) {
      const float4 v = *reinterpret_cast<const float4 *>(xr + k);
      const float4 m = *reinterpret_cast<const float4 *>(mu + k);
      h[0] = (_Float16)((v.x - m.x) * s);
      h[1] = (_Float16)((v.y - m.y) * s);
      h[2] = (_Float16)((v.z - m.z) * s);
      h[3] = (_Float16)((v.w - m.w) * s);
      const double e0 = ((double)v.x - (double)m.x) - (double)(float)h[0] * inv_s, e1 = ((double)v.y - (double)m.y) - (double)(float)h[1] * inv_s;
      const double e2 = ((double)v.z - (double)m.z) - (double)(float)h[2] * inv_s, e3 = ((double)v.w - (double)m.w) - (double)(float)h[3] * inv_s;
      dd += e0 * e0 + e1 * e1 + e2 * e2 + e3 * e3;
    }
    *reinterpret_cast<h4 *>(o + k) = h;
  }
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) dd += __shfl_xor(dd, off);
  if (lane == 0 && r < n) {
    float e = (float)sqrt(dd) * 1.0001f;
    if (!(e >= 0.f) || e > 3.0e38f) e = 3.0e38f;   // inf / NaN rows (f16 overflow cannot happen: the scale is set from the maximum)
    const unsigned int eb = __float_as_uint(e);
    // the running maximum is read first: after the first rows almost no row raises it (8.8 M atomics on one word otherwise)
    if (eb > __atomic_load_n(&bits[3], __ATOMIC_RELAXED)) atomicMax(&bits[3], eb);
  }
}

// queries -> f16(q * S_q) with a power of two S_q per row; qnorm = ||q|| (rounded up), qinv = 1 / (S_q S_d),
// qshift = q.mu in f64 (approx + qshift estimates q.d).  One wave per row.
__global__ __launch_bounds__(256) void split_queries_f16_kernel(const float *__restrict__ x, long long n, int dim,
                                                               int dimp, const float *__restrict__ mu,
                                                               const float *__restrict__ doc_scal,
                                                               _Float16 *__restrict__ out, float *__restrict__ qnorm,
                                                               float *__restrict__ qinv, double *__restrict__ qshift,
                                                               float *__restrict__ qdelta, float *__restrict__ qn16) {
  const long long r = (long long)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (r >= ((n + 255) >> 8 << 8)) return;
  const int lane = threadIdx.x & 63;
  typedef _Float16 h4 __attribute__((ext_vector_type(4)));
  if (r >= n) {  // rows up to the end of the last 256-row block: zero (the tile stream reads whole blocks)
    for (int k = lane * 4; k < dimp; k += 256) *reinterpret_cast<h4 *>(out + image_at(r, k, dimp)) = h4{0, 0, 0, 0};
    return;
  }
  const float *xr = x + (size_t)r * dim;
  float ss = 0.f, mx = 0.f;
  double sh = 0.0;
  for (int k = lane * 4; k < dim; k += 256) {
    const float4 v = *reinterpret_cast<const float4 *>(xr + k);
    const float4 m = *reinterpret_cast<const float4 *>(mu + k);
    ss = fmaf(v.x, v.x, fmaf(v.y, v.y, fmaf(v.z, v.z, fmaf(v.w, v.w, ss))));
    mx = fmaxf(fmaxf(mx, fmaxf(fabsf(v.x), fabsf(v.y))), fmaxf(fabsf(v.z), fabsf(v.w)));
    sh += (double)v.x * m.x + (double)v.y * m.y + (double)v.z * m.z + (double)v.w * m.w;
  }
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) {
    ss += __shfl_xor(ss, off);
    mx = fmaxf(mx, __shfl_xor(mx, off));
    sh += __shfl_xor(sh, off);
  }
  const float s = pow2_scale(mx);
  const double inv_s = 1.0 / (double)s;
  double dq = 0.0, n16 = 0.0;   // ||q - image / S_q||^2 and ||image / S_q||^2, measured (f64 per element)
  for (int k = lane * 4; k < dimp; k += 256) {
    h4 h = {0, 0, 0, 0};
    if (k < dim) {
      const float4 v = *reinterpret_cast<const float4 *>(xr + k);
      h[0] = (_Float16)(v.x * s);
      h[1] = (_Float16)(v.y * s);
      h[2] = (_Float16)(v.z * s);
      h[3] = (_Float16)(v.w * s);
      const double g0 = (double)(float)h[0] * inv_s, g1 = (double)(float)h[1] * inv_s, g2 = (double)(float)h[2] * inv_s, g3 = (double)(float)h[3] * inv_s;
      const double e0 = (double)v.x - g0, e1 = (double)v.y - g1, e2 = (double)v.z - g2, e3 = (double)v.w - g3;
      dq += e0 * e0 + e1 * e1 + e2 * e2 + e3 * e3;
      n16 += g0 * g0 + g1 * g1 + g2 * g2 + g3 * g3;
    }
    *reinterpret_cast<h4 *>(out + image_at(r, k, dimp)) = h;
  }
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) {
    dq += __shfl_xor(dq, off);
    n16 += __shfl_xor(n16, off);
  }
  if (lane == 0) {
    qnorm[r] = sqrtf(ss) * 1.00001f;
    qinv[r] = (1.f / s) * doc_scal[1];
    qshift[r] = sh;
    qdelta[r] = (float)sqrt(dq) * 1.0001f;
    qn16[r] = (float)sqrt(n16) * 1.0001f;
  }
}

// The bound of the f16 approximation with MEASURED roundings (round 6).  With q = q16 + dq and d - mu = d16 + dd (images
// divided by their scales), q.(d - mu) - q16.d16 = dq.(d - mu) + q16.dd, and the matrix core's f32 accumulation of the exact
// f16 products adds at most c_acc ||q16|| ||d16||, c_acc = 4 dimp 2^-24 (four ulps per step: it may truncate):
//   |acc / (S_q S_d) - q.(d - mu)| <= ||dq|| ||d - mu|| + ||q16|| ||dd|| + c_acc ||q16|| (||d - mu|| + ||dd||)
// -- the worst case u ||q|| ||d - mu|| (2 + u) of rounds 1-5 (h1_c1) with each factor replaced by what the rounding really
// did (~0.4 u ||.|| on real-valued rows: about half the bound; exactly representable rows: zero).  `dn` = ||d - mu|| of the
// document (or the shard's maximum), `ddmax` = the shard's max ||dd||; the f32 chain's own c2 ||q|| max ||d|| is added by
// the callers.
__device__ __forceinline__ double h1_err_bound(float qdelta, float qn16, double dn, double ddmax, float c_acc) {
  return (double)qdelta * dn + (double)qn16 * ddmax + (double)c_acc * (double)qn16 * (dn + ddmax);
}

// Approximate scores (f16) + threshold filter: same epilogue as ip_filter_kernel, A = f16 corpus rows (two
// 128-row tiles), B = 256 f16 queries.  Keys carry the RAW accumulator (= S_q S_d x the centred approximate
// score): per query that is a monotone image of the approximate score, which is all tau and the ranking need.
// Persistent: gridDim.x (a multiple of 8) workgroups; the workgroups of one XCD (blockIdx & 7) walk that XCD's
// contiguous range of work items side by side, so at any time they sit in one super-tile and share both
// operands through the XCD's L2.
__global__ __launch_bounds__(PP_THREADS, 2) void ip_filter_h1_kernel(
    const float *__restrict__ Qh, int nq, const float *__restrict__ Dh, long long doc_begin, long long doc_end,
    int dimp, const float *__restrict__ tau, unsigned long long *__restrict__ buf,
    unsigned int *__restrict__ count, int S, int k, int cap, unsigned int id_base, int n_qtiles, int n_dpairs) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  const int nwg = n_qtiles * n_dpairs;
  const int xcd = blockIdx.x & 7, per_xcd = gridDim.x >> 3;
  const int q8 = nwg >> 3, r8 = nwg & 7;
  const int range_base = (xcd < r8) ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8;  // as xcd_remap
  const int range_len = q8 + (xcd < r8 ? 1 : 0);
  int item = blockIdx.x >> 3;  // next work item of this workgroup inside the XCD's range
  const int t = threadIdx.x;
  const int w8 = __builtin_amdgcn_readfirstlane(t >> 6);
  const int grp = w8 >> 2, wm = (w8 >> 1) & 1, wn = w8 & 1;
  const size_t block_bytes = (size_t)256 * dimp * 2;  // one 256-row block of a unit-major image
  // tiles fetched ahead of their epilogue: (dpair, qtile) FIFO of at most two entries, head and tail in
  // scalars (an indexed array would live in scratch and drain the DMA queue on every access)
  int head_d = 0, head_q = 0, tail_d = 0, tail_q = 0, n_pend = 0;

  auto next = [&](H1Src &s) -> bool {
    if (item >= range_len) return false;
    int dpair, qtile;
    supertile_order<4, 8>(range_base + item, n_dpairs, n_qtiles, dpair, qtile);
    item += per_xcd;
    if (n_pend == 0) head_d = dpair, head_q = qtile;
    else tail_d = dpair, tail_q = qtile;
    ++n_pend;
    // DMA duty of this wave: waves 0-3 stage the two corpus tiles (LDS rows [0,256)), waves 4-7 the query tile.  A tile
    // is one 256-row block of its unit-major image (doc_begin is a multiple of 256; rows past the operand are zero)
    if (w8 < 4) {
      const long long first = doc_begin + (long long)dpair * 2 * BM;
      s.src = reinterpret_cast<const char *>(Dh) + (size_t)(first >> 8) * block_bytes;
    } else {
      s.src = reinterpret_cast<const char *>(Qh) + (size_t)qtile * block_bytes;
    }
    s.bytes = (unsigned int)block_bytes;
    return true;
  };
  float tq[4];
  Stash stash;  // behind the unit buffers of the tile stream
  {
    char *sb = reinterpret_cast<char *>(lds) + h1_lds_bytes() + (size_t)w8 * STASH_BYTES_PER_WAVE;
    stash.keys = reinterpret_cast<unsigned long long *>(sb);
    stash.qs = reinterpret_cast<unsigned int *>(sb + (size_t)STASH_N * 8);
    stash.n = 0;
  }
  auto begin = [&]() {  // start of the tile at the head of the FIFO: fetch its thresholds under the main loop
    load_tq<4>(tq, tau, head_q * H1_QT + 128 * wn, nq);
  };
  auto emit = [&](f32x16 (&acc)[2][4]) {
    const int dpair = head_d, qtile = head_q;
    head_d = tail_d, head_q = tail_q;
    --n_pend;
    const long long drow0 = doc_begin + ((long long)dpair * 2 + grp) * BM;
    emit_tile<4>(acc, tq, qtile * H1_QT + 128 * wn, drow0 + 64 * wm, doc_end, buf, count, S, k, cap, id_base, stash);
  };
  h1_tile_stream(64, dimp / 32, lds, next, begin, emit, H1BlockedUnits());
  stash_flush(stash, buf, count, S, k, cap);
}

// The same kernel on v_mfma_f32_16x16x32_f16 (mfma_pp_f16x16.h; the default: +10 % sustained rate of the tile loop on this
// part, tools/probes/mfma16_probe.hip) with the 16 x 16 epilogue.  dimp must be a multiple of 64 (pad_k).
template <int NI>   // query blocks per wave: query tiles of QT = 32 NI (h16_tile_stream<NI>)
__global__ __launch_bounds__(PP_THREADS, 2) void ip_filter_h16_kernel(
    const float *__restrict__ Qh, int nq, const float *__restrict__ Dh, long long doc_begin, long long doc_end,
    int dimp, const float *__restrict__ tau, unsigned long long *__restrict__ buf,
    unsigned int *__restrict__ count, int S, int k, int cap, unsigned int id_base, int n_qtiles, int n_dpairs,
    int flush_mask) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  const int nwg = n_qtiles * n_dpairs;
  const int xcd = blockIdx.x & 7, per_xcd = gridDim.x >> 3;
  const int q8 = nwg >> 3, r8 = nwg & 7;
  const int range_base = (xcd < r8) ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8;  // as xcd_remap
  const int range_len = q8 + (xcd < r8 ? 1 : 0);
  int item = blockIdx.x >> 3;
  const int t = threadIdx.x;
  const int w8 = __builtin_amdgcn_readfirstlane(t >> 6);
  const int grp = w8 >> 2, wm = (w8 >> 1) & 1, wn = w8 & 1;
  const size_t block_bytes = (size_t)256 * dimp * 2;
  constexpr int QT = 32 * NI;
  int head_d = 0, head_q = 0, tail_d = 0, tail_q = 0, n_pend = 0;

  auto next = [&](H1Src &s) -> bool {
    if (item >= range_len) return false;
    int dpair, qtile;
    supertile_order<4, 8>(range_base + item, n_dpairs, n_qtiles, dpair, qtile);
    item += per_xcd;
    if (n_pend == 0) head_d = dpair, head_q = qtile;
    else tail_d = dpair, tail_q = qtile;
    ++n_pend;
    if (w8 < 4) {
      const long long first = doc_begin + (long long)dpair * 2 * BM;
      s.src = reinterpret_cast<const char *>(Dh) + (size_t)(first >> 8) * block_bytes;
    } else {
      // (the query image is cut in blocks of 256 rows: query tile `qtile` of 32 NI rows starts at row qtile * QT of its block)
      s.src = reinterpret_cast<const char *>(Qh) + (size_t)((qtile * QT) >> 8) * block_bytes + (size_t)((qtile * QT) & 255) * 64;
    }
    s.bytes = (unsigned int)block_bytes;
    return true;
  };
  float tq[NI];
  RecStash stash;
  {
    char *sb = reinterpret_cast<char *>(lds) + h1_lds_bytes() + (size_t)w8 * STASH_BYTES_PER_WAVE;
    stash.vals = (lds_f32x4 *)(sb);
    stash.meta = (lds_u32x2 *)(sb + (size_t)H16_REC_N * 16);
    stash.n = 0;
  }
  int tiles_done = 0;
  auto begin = [&]() {
    load_tq16<NI>(tq, tau, head_q * QT + 16 * NI * wn, nq);
  };
  auto emit = [&](f32x4 (&acc)[4][NI]) {
    const int dpair = head_d, qtile = head_q;
    head_d = tail_d, head_q = tail_q;
    --n_pend;
    const long long drow0 = doc_begin + ((long long)dpair * 2 + grp) * BM;
    const int qb = qtile * QT + 16 * NI * wn;
    if (qb + 16 * NI > nq) {   // the last query tile (wave-uniform): columns past nq never pass
#pragma unroll
      for (int ni = 0; ni < NI; ++ni)
        if (qb + 16 * ni + (t & 15) >= nq) tq[ni] = INFINITY;
    }
    ++tiles_done;
    emit_tile16<NI>(acc, tq, qb, drow0 + 64 * wm, doc_end, tau, buf, count, S, k, cap, id_base, stash, (tiles_done & flush_mask) == 0);
  };
  h16_tile_stream<NI>(64, dimp / 32, lds, next, begin, emit, H1BlockedUnits());
  rec_flush(stash, tau, buf, count, S, k, cap);
}

// ---------------------------------------------------------------------------
// Few queries (nq <= 32: faiss_search.profile's batch sizes, MEVI/faiss_search.py:32-68).  With one MFMA tile of query
// columns the filter is bound by streaming the corpus image from HBM, and the tile stream above keeps only three 16 KiB
// units of it in flight per CU (the other half of its LDS ring stages the query tile again and again).  Here the query
// tile is STATIONARY in LDS (32 rows x dimp halves, 48 KiB at dim 768, loaded once per workgroup) and every wave streams
// its own 32 corpus rows of a 256-row block through a private ring of SM_NB units (2 KiB each; no workgroup barrier in
// the loop): 8 waves x 4 units = 64 KiB of the image in flight per CU (the stash and the query tile take the rest of LDS).  Same keys, same epilogue (emit_tile), same
// compaction as the large-batch kernel -- the returned bits cannot differ.
constexpr int SM_NB = 5;                       // units in a wave's ring (one being read, four landing)
constexpr size_t sm_lds_bytes(int dimp) { return (size_t)dimp * 64 + (size_t)8 * SM_NB * 2048; }

__global__ __launch_bounds__(512, 2) void ip_filter_h1_small_kernel(
    const float *__restrict__ Qh, int nq, const float *__restrict__ Dh, long long doc_begin, long long doc_end, int dimp,
    const float *__restrict__ tau, unsigned long long *__restrict__ buf, unsigned int *__restrict__ count, int S, int k,
    int cap, unsigned int id_base) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  const int t = threadIdx.x, lane = t & 63;
  const int w8 = __builtin_amdgcn_readfirstlane(t >> 6);
  const int lrow = lane & 31, half = lane >> 5;
  const int U = dimp >> 5;
  float *qst = lds;                                       // [U][32 rows][16 floats]: the query tile, piece-swizzled
  float *ring = lds + (size_t)U * 512 + (size_t)w8 * SM_NB * 512;  // this wave's units: [SM_NB][32 rows][16 floats]
  // lane (r16 = lane >> 2, slot = lane & 3) of a 16-row piece fetches logical piece slot ^ ((row >> 2) & 3)
  const int voff = (lane >> 2) * 64 + (((lane & 3) ^ ((lane >> 4) & 3)) << 4);
  {  // query tile: unit u of block 0 of the query image, rows 0..31 = its first 2 KiB; units spread over the waves
    const __amdgpu_buffer_rsrc_t rq = __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(Qh), 0, (int)((size_t)U * 16384), 0x00020000);
    for (int u = w8; u < U; u += 8) {
#pragma unroll
      for (int i = 0; i < 2; ++i)
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rq, (__attribute__((address_space(3))) void *)(qst + u * 512 + i * 256), 16,
                                                 voff + i * 1024, u * 16384, 0, 0);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
  }
  const long long n_blocks = (doc_end - doc_begin + 255) >> 8;
  long long blk = blockIdx.x;
  if (blk >= n_blocks) return;
  const size_t block_bytes = (size_t)256 * dimp * 2;
  auto src_of = [&](long long b) { return reinterpret_cast<const char *>(Dh) + (size_t)((doc_begin >> 8) + b) * block_bytes + (size_t)w8 * 2048; };
  const char *cur = src_of(blk);
  long long blk_n = blk + gridDim.x;
  bool have_nxt = blk_n < n_blocks;
  const char *nxt = have_nxt ? src_of(blk_n) : cur;
  // unit u of a block: this wave's 32 rows = 2 KiB at byte u * 16 KiB of the block (rows 32 w8 .. 32 w8 + 31)
  auto dma = [&](const char *base, bool live, int u, int slot) {
    const __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc(const_cast<char *>(base), 0, live ? (int)(block_bytes - (size_t)w8 * 2048) : 0, 0x00020000);
#pragma unroll
    for (int i = 0; i < 2; ++i)
      __builtin_amdgcn_raw_ptr_buffer_load_lds(r, (__attribute__((address_space(3))) void *)(ring + slot * 512 + i * 256), 16,
                                               voff + i * 1024, u * 16384, 0, 0);
  };
  const int sw = (lrow >> 2) & 3;
  const int off0 = lrow * 16 + (((0 + half) ^ sw) << 2), off1 = lrow * 16 + (((2 + half) ^ sw) << 2);  // k-steps j = 0, 1
  float tq[1];
  Stash stash;
  {
    char *sb = reinterpret_cast<char *>(lds) + sm_lds_bytes(dimp) + (size_t)w8 * STASH_BYTES_PER_WAVE;
    stash.keys = reinterpret_cast<unsigned long long *>(sb);
    stash.qs = reinterpret_cast<unsigned int *>(sb + (size_t)STASH_N * 8);
    stash.n = 0;
  }
  load_tq<1>(tq, tau, 0, nq);
  int rs = 0;  // ring slot of the unit being computed
#pragma unroll
  for (int u = 0; u < SM_NB - 1; ++u) dma(cur, true, u, u);   // U >= SM_NB - 1 is guaranteed by the host (dimp >= 160)
  while (true) {
    f32x16 acc[2][1];
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[0][0][r] = 0.f, acc[1][0][r] = -INFINITY;   // rows 32..63 of emit_tile's tile: none
    for (int u = 0; u < U; ++u) {
      // request stream unit g + SM_NB - 1 (slot before rs in the ring: its reads finished an iteration ago), then wait
      // for unit g: at most the 2 * (SM_NB - 1) pieces behind it stay outstanding
      const int un = u + SM_NB - 1;
      const bool spill = un >= U;
      dma(spill ? nxt : cur, spill ? have_nxt : true, spill ? un - U : un, rs == 0 ? SM_NB - 1 : rs - 1);
      asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
      static_assert(SM_NB == 5, "vmcnt immediate above = 2 * (SM_NB - 1)");
      const float *ub = ring + rs * 512;
      const f16x8 a0 = *reinterpret_cast<const f16x8 *>(ub + off0), a1 = *reinterpret_cast<const f16x8 *>(ub + off1);
      const f16x8 b0 = *reinterpret_cast<const f16x8 *>(qst + u * 512 + off0), b1 = *reinterpret_cast<const f16x8 *>(qst + u * 512 + off1);
      acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a0, b0, acc[0][0], 0, 0, 0);
      acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a1, b1, acc[0][0], 0, 0, 0);
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // the slot is re-filled by the next iteration's request
      rs = rs == SM_NB - 1 ? 0 : rs + 1;
    }
    emit_tile<1>(acc, tq, 0, doc_begin + blk * 256 + 32 * w8, doc_end, buf, count, S, k, cap, id_base, stash);
    if (!have_nxt) break;
    blk = blk_n;
    cur = nxt;
    blk_n += gridDim.x;
    have_nxt = blk_n < n_blocks;
    if (have_nxt) nxt = src_of(blk_n);
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  stash_flush(stash, buf, count, S, k, cap);
}

// ---------------------------------------------------------------------------
// Round 6 (late): an 8-BIT corpus image for searches of <= 32 queries.  The streaming kernel above is bound by reading the
// f16 image once per launch set (13.6 GB at MS MARCO size: 2.5 ms at the 5.4 TB/s it reaches); an int8 image is half of
// that.  What makes 8 bits usable for an EXACT search is that the integer matrix cores accumulate without rounding, so the
// whole error of the approximate score is the two quantisations, and both are MEASURED:
//   y_k = (d_k - mu_k) / c_k        c = per-column scale (the column's standard deviation: outlier dimensions of real encoders
//                                   do not eat the 8 bits of the others); folded into the query: w_k = q_k c_k
//   y^_k = s_r I_k                  I = int8 row, s_r = max_k |y_k| / 127 per ROW (rows of different length use all levels)
//   w^_k = t_q Q_k                  Q = 15-bit integer in two int8 digits, Q = 128 hi + lo, lo in [-64, 63]: two MFMAs per
//                                   product (the kernel is HBM-bound at 1/6 of the integer pipes)
//   q.(d - mu) - t_q s_r (128 N_hi + N_lo) = w.(y - y^) + (w - w^).y^      N = exact int32 sums
//   |.| <= ||w|| max_r ||y - y^|| + ||w - w^|| max_r ||y^||                 (Cauchy-Schwarz; both maxima measured at build,
//                                                                            the query norms per search, in f64)
// The first term is ~0.2 of the score deviation on Gaussian rows (f16: 0.002) and its MAXIMUM over rows half as much again
// (a row with one 5.5-sigma coordinate has a coarser step) -- measured on the first build: with the shard's maximum in the
// proof, 3 k survivors prove top-10 but not top-100 / top-1000 of 8.8 M rows.  So the term is carried PER ROW, inside the key:
// ||y - y^||_r <= rho s_r with rho = max_r ||y - y^||_r / s_r (the quantisation noise of a row is uniform in its own step: 8.0
// +- 0.7 at dim 768; measured at build), and the filter ranks rows by an UPPER BOUND of their centred score,
//   key_r = fl((fl(128 N_hi + N_lo) + G_q) s_r),   G_q = (rho ||w|| + eta ||w - w^||) / t_q      (t_q key_r >= w.y_r - 3 ulp)
// with eta = max_r ||I_r|| (the query's quantisation against row r is at most ||w - w^|| s_r ||I_r||: per row as well, +2.5 %
// on G_q, and rows of extreme length no longer widen every other row's bound through max ||y^||).
// A row outside the K' survivors has key <= the K'-th key, so the proof needs only what is left: 3 ulp for the float form of
// the key (|N_hi|, |N_lo| < 2^24 for dimp <= 1024).  That is
// h1_err_bound with (qdelta, qn16, dn, ddmax, c_acc) = (0, ||w|| + ||w - w^||, ||y^||, 2^-20 rho max s_r, 2^-22): the
// re-scoring, the proof and the observed / bound check (one-sided here: exact - key) are the SAME kernels as the f16
// search's, with qinv = t_q; the exclusion of survivors is not used (its lower bound would need the two-sided error).
// K' = 3 k + 64 (h8_kprime) instead of 1.25 k: re-scoring 3 k rows per query is nothing for <= 32 queries, and would be everything for 7 k.
// A query the proof leaves open sends the whole (small) batch through the f16 search.
typedef int i32x4 __attribute__((ext_vector_type(4)));
typedef int i32x16 __attribute__((ext_vector_type(16)));
constexpr int I8_QMAX = 127 * 128;

// element (r, k) of the 8-bit image: blocks of 256 rows, inside a block the 64 k of unit u of all rows together
// (the f16 image's layout with 64-k units: a unit is 16 KiB, a wave's 32 rows of it 2 KiB, a row of it 64 B)
__device__ __forceinline__ size_t image8_at(long long r, int k, int dimp) {
  return ((size_t)((r >> 8) * (dimp >> 6) + (k >> 6)) * 256 + (size_t)(r & 255)) * 64 + (k & 63);
}

// sum over rows of (d_k - mu_k)^2 per column, f64 (the column scale is its root mean; any positive scale is valid)
__global__ __launch_bounds__(256) void colsq_kernel(const float *__restrict__ x, long long n, int dim, const float *__restrict__ mu,
                                                   double *__restrict__ sum) {
  const long long r0 = (long long)blockIdx.x * 512;
  const long long r1 = r0 + 512 < n ? r0 + 512 : n;
  for (int c = threadIdx.x; c < dim; c += 256) {
    double a = 0.0;
    const double m = (double)mu[c];
    for (long long r = r0; r < r1; ++r) {
      const double v = (double)x[(size_t)r * dim + c] - m;
      a += v * v;
    }
    atomicAdd(&sum[c], a);
  }
}
__global__ __launch_bounds__(256) void colscale_kernel(const double *__restrict__ sum, long long n, int dim, int dimp, float *__restrict__ cs) {
  const int c = blockIdx.x * 256 + threadIdx.x;
  if (c >= dimp) return;
  float v = 1.f;
  if (c < dim && n > 0) {
    const float s = (float)sqrt(sum[c] / (double)n);
    if (s > 1e-18f && s < 1e18f) v = s;   // (with the row limits below: keys (N + G) s_r stay far inside the float range)
  }
  cs[c] = v;
}

// docs -> int8 image + per-row scale + per-row ||y^|| (rounded up); bits[0] = max ||y^||, bits[2] = rho = max ||y - y^|| / s_r,
// bits[4] = max ||y - y^|| (f64 per element, rounded up).  One wave per row; rows up to the end of the last block exist and are zero (scale 0).
__global__ __launch_bounds__(256) void split_docs_i8_kernel(const float *__restrict__ x, long long n, int dim, int dimp,
                                                           const float *__restrict__ mu, const float *__restrict__ cs,
                                                           signed char *__restrict__ out, float *__restrict__ scale,
                                                           float *__restrict__ ynorm, unsigned int *__restrict__ bits) {
  const long long r = (long long)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (r >= ((n + 255) >> 8 << 8)) return;
  const int lane = threadIdx.x & 63;
  if (r >= n) {
    for (int k = lane * 4; k < dimp; k += 256) *reinterpret_cast<unsigned int *>(out + image8_at(r, k, dimp)) = 0u;
    if (lane == 0) scale[r] = 0.f, ynorm[r] = 0.f;
    return;
  }
  const float *xr = x + (size_t)r * dim;
  float m = 0.f;
  bool bad = false;
  for (int k = lane * 4; k < dim; k += 256) {
    const float4 v = *reinterpret_cast<const float4 *>(xr + k);
    const float4 u = *reinterpret_cast<const float4 *>(mu + k);
    const float4 c = *reinterpret_cast<const float4 *>(cs + k);
    const float y0 = fabsf((v.x - u.x) / c.x), y1 = fabsf((v.y - u.y) / c.y), y2 = fabsf((v.z - u.z) / c.z), y3 = fabsf((v.w - u.w) / c.w);
    bad = bad || !(y0 <= 1.0e25f) || !(y1 <= 1.0e25f) || !(y2 <= 1.0e25f) || !(y3 <= 1.0e25f);   // (inf, NaN, or out of the keys' range)
    m = fmaxf(fmaxf(m, fmaxf(y0, y1)), fmaxf(y2, y3));
  }
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) m = fmaxf(m, __shfl_xor(m, off));
  bad = __any(bad || (m > 0.f && m < 1.0e-25f)) != 0;
  const float s = bad ? 0.f : m / 127.f;
  const float inv = (!bad && m > 0.f) ? 127.f / m : 0.f;
  double ee = 0.0, yy = 0.0;
  for (int k = lane * 4; k < dimp; k += 256) {
    unsigned int word = 0u;
    if (k < dim && !bad) {
      const float4 v = *reinterpret_cast<const float4 *>(xr + k);
      const float4 u = *reinterpret_cast<const float4 *>(mu + k);
      const float4 c = *reinterpret_cast<const float4 *>(cs + k);
      const float vv[4] = {v.x, v.y, v.z, v.w}, uu[4] = {u.x, u.y, u.z, u.w}, cc[4] = {c.x, c.y, c.z, c.w};
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        int q = (int)rintf(((vv[j] - uu[j]) / cc[j]) * inv);
        q = q > 127 ? 127 : (q < -127 ? -127 : q);
        const double yh = (double)s * (double)q;
        const double e = ((double)vv[j] - (double)uu[j]) / (double)cc[j] - yh;
        ee += e * e;
        yy += yh * yh;
        word |= ((unsigned int)q & 0xFFu) << (8 * j);
      }
    }
    *reinterpret_cast<unsigned int *>(out + image8_at(r, k, dimp)) = word;
  }
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) {
    ee += __shfl_xor(ee, off);
    yy += __shfl_xor(yy, off);
  }
  if (lane == 0) {
    float e = (float)sqrt(ee) * 1.0001f, y = (float)sqrt(yy) * 1.0001f;
    if (bad || !(e >= 0.f) || e > 3.0e38f) e = 3.0e38f;   // inf / NaN rows: no bound -> every query takes the f16 search
    if (!(y >= 0.f) || y > 3.0e38f) y = 3.0e38f;
    scale[r] = s;
    ynorm[r] = y;
    float rho = s > 0.f ? (float)(sqrt(ee) / (double)s) * 1.0001f : (e > 0.f ? 3.0e38f : 0.f);
    if (!(rho >= 0.f) || rho > 3.0e38f) rho = 3.0e38f;
    const unsigned int eb = __float_as_uint(e), yb = __float_as_uint(y), rb = __float_as_uint(rho);
    if (eb > __atomic_load_n(&bits[4], __ATOMIC_RELAXED)) atomicMax(&bits[4], eb);
    if (yb > __atomic_load_n(&bits[0], __ATOMIC_RELAXED)) atomicMax(&bits[0], yb);
    if (rb > __atomic_load_n(&bits[2], __ATOMIC_RELAXED)) atomicMax(&bits[2], rb);
    const unsigned int sb = __float_as_uint(s);
    if (sb > __atomic_load_n(&bits[5], __ATOMIC_RELAXED)) atomicMax(&bits[5], sb);
    const float eta = s > 0.f ? (float)(sqrt(yy) / (double)s) * 1.0001f : 0.f;   // ||I_r||, the length of the int8 row (<= 127 sqrt(dim))
    const unsigned int hb = __float_as_uint(eta <= 3.0e38f ? eta : 3.0e38f);
    if (hb > __atomic_load_n(&bits[6], __ATOMIC_RELAXED)) atomicMax(&bits[6], hb);
  }
}
// bits[1] = max ||d|| of the f16 index (the f32 chain's own term of the bound); bits[3] = 2^-20 rho max s_r (what the proof's
// formula still needs of the rows' quantisation: 3 ulp of the key's G_q s_r part, G_q s_r t_q = rho s_r ||w|| <= rho max s ||w||)
__global__ void finish_bits8_kernel(const unsigned int *__restrict__ src, unsigned int *__restrict__ dst) {
  if (threadIdx.x == 0 && blockIdx.x == 0) {
    dst[1] = src[1];
    const double e = (double)__uint_as_float(dst[2]) * (double)__uint_as_float(dst[5]) * 1.0001 / 1048576.0;
    dst[3] = __float_as_uint(e > 3.0e38 ? 3.0e38f : (float)e);
  }
}

// queries (<= 32) -> two int8 digit images [digit][unit][32 rows][64 B] of round(q_k c_k / t_q); qnorm = ||q|| (up), qinv = t_q,
// qshift = q.mu (f64), qdelta = 0 (the query's quantisation is inside G_q), qn16 = ||w|| + ||w - w^|| (up), qub = G_q (the key's
// per-row error terms in raw units).  One wave per row, rows >= n zero.
__global__ __launch_bounds__(256) void split_queries_i8_kernel(const float *__restrict__ x, int n, int dim, int dimp,
                                                              const float *__restrict__ mu, const float *__restrict__ cs,
                                                              signed char *__restrict__ out, float *__restrict__ qnorm,
                                                              float *__restrict__ qinv, double *__restrict__ qshift,
                                                              float *__restrict__ qdelta, float *__restrict__ qn16,
                                                              const unsigned int *__restrict__ bits8, float *__restrict__ qub,
                                                              unsigned int *__restrict__ qbad) {
  const int r = blockIdx.x * 4 + (threadIdx.x >> 6);   // 8 workgroups: rows 0..31
  const int lane = threadIdx.x & 63;
  const int U = dimp >> 6;
  auto at = [&](int digit, int k) { return ((size_t)(digit * U + (k >> 6)) * 32 + r) * 64 + (k & 63); };
  if (r >= n) {
    for (int k = lane * 4; k < dimp; k += 256) {
      *reinterpret_cast<unsigned int *>(out + at(0, k)) = 0u;
      *reinterpret_cast<unsigned int *>(out + at(1, k)) = 0u;
    }
    if (lane == 0) qub[r] = 0.f, qbad[r] = 0u;
    return;
  }
  const float *xr = x + (size_t)r * dim;
  float ss = 0.f, mw = 0.f;
  double sh = 0.0;
  for (int k = lane * 4; k < dim; k += 256) {
    const float4 v = *reinterpret_cast<const float4 *>(xr + k);
    const float4 m = *reinterpret_cast<const float4 *>(mu + k);
    const float4 c = *reinterpret_cast<const float4 *>(cs + k);
    ss = fmaf(v.x, v.x, fmaf(v.y, v.y, fmaf(v.z, v.z, fmaf(v.w, v.w, ss))));
    mw = fmaxf(fmaxf(mw, fmaxf(fabsf(v.x * c.x), fabsf(v.y * c.y))), fmaxf(fabsf(v.z * c.z), fabsf(v.w * c.w)));
    sh += (double)v.x * m.x + (double)v.y * m.y + (double)v.z * m.z + (double)v.w * m.w;
  }
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) {
    ss += __shfl_xor(ss, off);
    mw = fmaxf(mw, __shfl_xor(mw, off));
    sh += __shfl_xor(sh, off);
  }
  float t = mw / (float)I8_QMAX;
  if (!(t > 1e-37f) || !(t < 1e37f)) t = 1.f;
  const double inv_t = 1.0 / (double)t;
  double dq = 0.0, nw = 0.0;
  for (int k = lane * 4; k < dimp; k += 256) {
    unsigned int whi = 0u, wlo = 0u;
    if (k < dim) {
      const float4 v = *reinterpret_cast<const float4 *>(xr + k);
      const float4 c = *reinterpret_cast<const float4 *>(cs + k);
      const float vv[4] = {v.x, v.y, v.z, v.w}, cc[4] = {c.x, c.y, c.z, c.w};
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const double w = (double)vv[j] * (double)cc[j];
        double qd = rint(w * inv_t);
        qd = qd > (double)I8_QMAX ? (double)I8_QMAX : (qd < -(double)I8_QMAX ? -(double)I8_QMAX : qd);
        if (!(qd == qd)) qd = 0.0;
        const int Q = (int)qd;
        const int lo = ((Q + 64) & 127) - 64, hi = (Q - lo) >> 7;
        const double e = w - (double)t * (double)Q;
        dq += e * e;
        nw += w * w;
        whi |= ((unsigned int)hi & 0xFFu) << (8 * j);
        wlo |= ((unsigned int)lo & 0xFFu) << (8 * j);
      }
    }
    *reinterpret_cast<unsigned int *>(out + at(0, k)) = whi;
    *reinterpret_cast<unsigned int *>(out + at(1, k)) = wlo;
  }
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) {
    dq += __shfl_xor(dq, off);
    nw += __shfl_xor(nw, off);
  }
  if (lane == 0) {
    const float qn = sqrtf(ss) * 1.00001f, qd = (float)sqrt(dq) * 1.0001f, qw = (float)(sqrt(nw) + sqrt(dq)) * 1.0001f;
    qnorm[r] = qn;
    qinv[r] = t;
    qshift[r] = sh;
    qdelta[r] = 0.f;     // the query's own quantisation is inside G_q too (below): nothing of it is left for the proof's formula
    qn16[r] = qw;
    // G_q = (rho ||w|| + eta ||w - w^||) / t_q, rounded up: |w.(y - y^)_r| <= ||w|| rho s_r and |(w - w^).y^_r| <= ||w - w^|| s_r ||I_r||
    // <= ||w - w^|| s_r eta -- both per ROW through s_r, so a few rows of extreme length do not widen every other row's bound
    const float g = (float)(((double)__uint_as_float(bits8[2]) * sqrt(nw) + (double)__uint_as_float(bits8[6]) * sqrt(dq)) * inv_t * 1.0001);
    qub[r] = g;
    // a non-finite query, or rows the image could not hold (rho = 3e38): keys would be inf / NaN and NaN keys are silently
    // dropped by the filter's comparisons -- the driver sends such a query's batch through the f16 search
    const float chk = g + qd + qw + qn + (float)sh;
    qbad[r] = (chk - chk == 0.f && g < 1e30f) ? 0u : 1u;
  }
}

// ip_filter_h1_small_kernel on the 8-bit image: same ring, same DMA pieces, same LDS addresses (a unit row is 64 B in both:
// 32 halves there, 64 bytes here), U = dimp / 64 units of four integer MFMAs (two k-steps x two query digits).  The int32
// sums leave as the float keys (fl(128 N_hi + N_lo) + G_q) * s_row through the same epilogue (emit_tile).
__global__ __launch_bounds__(512, 2) void ip_filter_i8_small_kernel(
    const float *__restrict__ Qh, int nq, const float *__restrict__ Dh, long long doc_begin, long long doc_end, int dimp,
    const float *__restrict__ tau, unsigned long long *__restrict__ buf, unsigned int *__restrict__ count, int S, int k,
    int cap, unsigned int id_base, const float *__restrict__ row_scale, const float *__restrict__ qub) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  const int t = threadIdx.x, lane = t & 63;
  const int w8 = __builtin_amdgcn_readfirstlane(t >> 6);
  const int lrow = lane & 31, half = lane >> 5;
  const int U = dimp >> 6;
  float *qst = lds;                                                    // [2 digits][U][32 rows][16 floats], piece-swizzled
  float *ring = lds + (size_t)U * 1024 + (size_t)w8 * SM_NB * 512;     // this wave's units: [SM_NB][32 rows][16 floats]
  const int voff = (lane >> 2) * 64 + (((lane & 3) ^ ((lane >> 4) & 3)) << 4);
  {
    const __amdgpu_buffer_rsrc_t rq = __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(Qh), 0, (int)((size_t)U * 4096), 0x00020000);
    for (int u = w8; u < 2 * U; u += 8) {
#pragma unroll
      for (int i = 0; i < 2; ++i)
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rq, (__attribute__((address_space(3))) void *)(qst + u * 512 + i * 256), 16,
                                                 voff + i * 1024, u * 2048, 0, 0);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
  }
  const long long n_blocks = (doc_end - doc_begin + 255) >> 8;
  long long blk = blockIdx.x;
  if (blk >= n_blocks) return;
  const size_t block_bytes = (size_t)256 * dimp;
  auto src_of = [&](long long b) { return reinterpret_cast<const char *>(Dh) + (size_t)((doc_begin >> 8) + b) * block_bytes + (size_t)w8 * 2048; };
  const char *cur = src_of(blk);
  long long blk_n = blk + gridDim.x;
  bool have_nxt = blk_n < n_blocks;
  const char *nxt = have_nxt ? src_of(blk_n) : cur;
  auto dma = [&](const char *base, bool live, int u, int slot) {
    const __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc(const_cast<char *>(base), 0, live ? (int)(block_bytes - (size_t)w8 * 2048) : 0, 0x00020000);
#pragma unroll
    for (int i = 0; i < 2; ++i)
      __builtin_amdgcn_raw_ptr_buffer_load_lds(r, (__attribute__((address_space(3))) void *)(ring + slot * 512 + i * 256), 16,
                                               voff + i * 1024, u * 16384, 0, 0);
  };
  const int sw = (lrow >> 2) & 3;
  const int off0 = lrow * 16 + (((0 + half) ^ sw) << 2), off1 = lrow * 16 + (((2 + half) ^ sw) << 2);  // k-steps j = 0, 1
  float tq[1];
  Stash stash;
  {
    char *sb = reinterpret_cast<char *>(lds) + sm_lds_bytes(dimp) + (size_t)w8 * STASH_BYTES_PER_WAVE;
    stash.keys = reinterpret_cast<unsigned long long *>(sb);
    stash.qs = reinterpret_cast<unsigned int *>(sb + (size_t)STASH_N * 8);
    stash.n = 0;
  }
  load_tq<1>(tq, tau, 0, nq);
  const float gq = qub[lrow];   // (32 entries, rows >= nq: 0)
  int rs = 0;
#pragma unroll
  for (int u = 0; u < SM_NB - 1; ++u) dma(cur, true, u, u);   // U >= SM_NB - 1 is guaranteed by the host (dimp >= 256)
  while (true) {
    // the scales of the wave's 32 rows: a wave-uniform address, i.e. SCALAR loads (they count in lgkmcnt; a vector load here
    // makes the compiler drain vmcnt -- the ring's look-ahead -- before the epilogue reads them)
    const float *sp = row_scale + (doc_begin + blk * 256 + 32 * w8);
    float sr[32];
#pragma unroll
    for (int j = 0; j < 32; ++j) sr[j] = sp[j];
    i32x16 nh, nl;
#pragma unroll
    for (int r = 0; r < 16; ++r) nh[r] = 0, nl[r] = 0;
    for (int u = 0; u < U; ++u) {
      const int un = u + SM_NB - 1;
      const bool spill = un >= U;
      dma(spill ? nxt : cur, spill ? have_nxt : true, spill ? un - U : un, rs == 0 ? SM_NB - 1 : rs - 1);
      asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
      static_assert(SM_NB == 5, "vmcnt immediate above = 2 * (SM_NB - 1)");
      const float *ub = ring + rs * 512;
      const i32x4 a0 = *reinterpret_cast<const i32x4 *>(ub + off0), a1 = *reinterpret_cast<const i32x4 *>(ub + off1);
      const float *qh = qst + u * 512, *ql = qst + (U + u) * 512;
      const i32x4 h0 = *reinterpret_cast<const i32x4 *>(qh + off0), h1 = *reinterpret_cast<const i32x4 *>(qh + off1);
      const i32x4 l0 = *reinterpret_cast<const i32x4 *>(ql + off0), l1 = *reinterpret_cast<const i32x4 *>(ql + off1);
      nh = __builtin_amdgcn_mfma_i32_32x32x32_i8(a0, h0, nh, 0, 0, 0);
      nl = __builtin_amdgcn_mfma_i32_32x32x32_i8(a0, l0, nl, 0, 0, 0);
      nh = __builtin_amdgcn_mfma_i32_32x32x32_i8(a1, h1, nh, 0, 0, 0);
      nl = __builtin_amdgcn_mfma_i32_32x32x32_i8(a1, l1, nl, 0, 0, 0);
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      rs = rs == SM_NB - 1 ? 0 : rs + 1;
    }
    f32x16 acc[2][1];
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const float sc = half ? sr[8 * (r >> 2) + 4 + (r & 3)] : sr[8 * (r >> 2) + (r & 3)];   // row (r & 3) + 8 (r >> 2) + 4 half
      acc[0][0][r] = (fmaf((float)nh[r], 128.f, (float)nl[r]) + gq) * sc;
      acc[1][0][r] = -INFINITY;
    }
    emit_tile<1>(acc, tq, 0, doc_begin + blk * 256 + 32 * w8, doc_end, buf, count, S, k, cap, id_base, stash);
    if (!have_nxt) break;
    blk = blk_n;
    cur = nxt;
    blk_n += gridDim.x;
    have_nxt = blk_n < n_blocks;
    if (have_nxt) nxt = src_of(blk_n);
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  stash_flush(stash, buf, count, S, k, cap);
}

// Exact re-scoring of the kp approximate survivors of every query, exact top-k, and the proof that nothing
// outside the survivors can belong to it.  With a = acc * qinv (the centred approximate score),
//   chain(q, d) <= a + q.mu + eps_q,   eps_q = h1_err_bound(q; max||d - mu||, max||dd||) + c2 ||q|| max||d||
//   (c1 below is the accumulation constant h1_cacc; the rounding terms are measured: h1_err_bound)
// for every document of the shard; every non-survivor has acc <= acc_last (the kp-th raw accumulator = the
// final tau of the approximate pass), so if
//   acc_last * qinv + q.mu + eps_q < e_k   (the exact k-th score; compared in f64)
// the list is exact.
//
// rescore_rows_kernel: one WAVE per 64 survivors of a query, one lane per survivor, sequential fmaf chain over k
// (the oracle's chain).  Rows are gathered 64 x 128 B at a time (whole lines, the next slab in flight in
// registers while this one is consumed) through a 9 KiB per-wave LDS tile; waves are independent, so a CU holds
// 16 of them and the gather runs near the random-row HBM rate.  The keys are re-scored in place.
constexpr int RS_LD = 36;  // floats per staged row: 32 + 4 pad (16-byte aligned, conflict-light float4 reads)
__global__ __launch_bounds__(256) void rescore_rows_kernel(const float *__restrict__ Q, const float *__restrict__ D,
                                                          int dim, unsigned long long *__restrict__ buf, int S,
                                                          int kp, int nq, unsigned int id_base,
                                                          const float *__restrict__ qnorm,
                                                          const float *__restrict__ qinv,
                                                          const double *__restrict__ qshift, float c1, float c2,
                                                          const unsigned int *__restrict__ dmax_bits,
                                                          const float *__restrict__ dnorm_c,
                                                          unsigned int *__restrict__ err_ratio_bits,
                                                          const float *__restrict__ qdelta, const float *__restrict__ qn16,
                                                          const float *__restrict__ acc_k, int upper_keys = 0) {
  __shared__ __attribute__((aligned(16))) float tiles[4][64 * RS_LD];
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int wpq = (kp + 63) / 64;
  const long long gw = (long long)blockIdx.x * 4 + wave;
  const int q = (int)(gw / wpq);
  if (q >= nq) return;  // wave-uniform; no workgroup barrier in this kernel
  const int c = (int)(gw - (long long)q * wpq) * 64 + lane;
  unsigned long long *row = buf + (size_t)q * S;
  unsigned long long key = (c < kp) ? row[c] : 0ull;
  // Round 6: a survivor that PROVABLY cannot be among the exact top-k is not re-scored.  acc_k[q] = the raw accumulator of the
  // k-th best survivor by approximate score: each of those k has exact >= acc_k qinv + q.mu - E (E = the bound of 4.1b' at the
  // shard's maxima), this one has exact <= acc qinv + q.mu + E_c (its own ||d - mu||) -- if the second is below the first, k
  // survivors beat it.  Its key is emptied (the finish kernel sorts what is left; the proof about NON-survivors is unchanged);
  // its lane gathers row 0 (cached).  ~15 % of the K' = 1280 survivors of an MS MARCO-sized search: 27.4 -> ~23 GB gathered.
  if (acc_k && key != 0ull) {
    const unsigned int idc = key_id(key) - id_base;
    const double e_max = h1_err_bound(qdelta[q], qn16[q], (double)__uint_as_float(dmax_bits[0]), (double)__uint_as_float(dmax_bits[3]), c1) +
                         (double)qnorm[q] * (double)c2 * __uint_as_float(dmax_bits[1]);
    const double e_c = h1_err_bound(qdelta[q], qn16[q], (double)dnorm_c[idc], (double)__uint_as_float(dmax_bits[3]), c1) +
                       (double)qnorm[q] * (double)c2 * __uint_as_float(dmax_bits[1]);
    if ((double)key_score(key) * (double)qinv[q] + e_c < (double)acc_k[q] * (double)qinv[q] - e_max) {
      key = 0ull;
      if (c < kp) row[c] = 0ull;
    }
  }
  const bool valid = key != 0ull;
  const unsigned int my_id = valid ? key_id(key) - id_base : 0u;
  float *sd = tiles[wave];
  // staging duty: float4 number lane&7 of the rows (lane>>3) + 8i
  const float *src[8];
#pragma unroll
  for (int i = 0; i < 8; ++i) src[i] = D + (size_t)__shfl(my_id, (lane >> 3) + 8 * i) * dim + (lane & 7) * 4;
  const float *qr = Q + (size_t)q * dim;
  const int nslab = (dim + 31) / 32;
  float4 v[8];
  auto fetch = [&](int s) {
    const bool in = s * 32 + (lane & 7) * 4 < dim;
#pragma unroll
    for (int i = 0; i < 8; ++i) v[i] = in ? *reinterpret_cast<const float4 *>(src[i] + s * 32) : make_float4(0.f, 0.f, 0.f, 0.f);
  };
  fetch(0);
  float acc = 0.f;
  for (int s = 0; s < nslab; ++s) {
#pragma unroll
    for (int i = 0; i < 8; ++i) *reinterpret_cast<float4 *>(sd + ((lane >> 3) + 8 * i) * RS_LD + (lane & 7) * 4) = v[i];
    if (s + 1 < nslab) fetch(s + 1);
    __builtin_amdgcn_wave_barrier();  // LDS is in order within a wave: the reads below see the stores above
    const float *dr = sd + lane * RS_LD;
#pragma unroll
    for (int k4 = 0; k4 < 32; k4 += 4) {
      const float4 x = *reinterpret_cast<const float4 *>(dr + k4);
      const int kk = s * 32 + k4;  // dim % 4 == 0: a float4 of q is inside the row or entirely past it
      const float4 qq = kk < dim ? *reinterpret_cast<const float4 *>(qr + kk) : make_float4(0.f, 0.f, 0.f, 0.f);
      acc = fmaf(qq.x, x.x, acc);
      acc = fmaf(qq.y, x.y, acc);
      acc = fmaf(qq.z, x.z, acc);
      acc = fmaf(qq.w, x.w, acc);
    }
    __builtin_amdgcn_wave_barrier();  // ... and the next slab's stores come after these reads
  }
  if (c < kp) row[c] = valid ? make_key(acc, key_id(key)) : 0ull;
  if (err_ratio_bits) {  // observed |approx - exact| / its bound: must stay far below 1
    float ratio = 0.f;
    if (valid) {
      const double den = h1_err_bound(qdelta[q], qn16[q], (double)dnorm_c[my_id], (double)__uint_as_float(dmax_bits[3]), c1) +
                         (double)qnorm[q] * (double)c2 * __uint_as_float(dmax_bits[1]);
      const double est = (double)key_score(key) * (double)qinv[q] + qshift[q];
      // (upper_keys: the 8-bit search's keys are upper bounds -- only exact - key is held against the bound)
      if (den > 0.0) ratio = (float)((upper_keys ? fmax((double)acc - est, 0.0) : fabs(est - (double)acc)) / den);
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) ratio = fmaxf(ratio, __shfl_xor(ratio, off));
    if (lane == 0 && ratio > 0.f) atomicMax(err_ratio_bits, __float_as_uint(ratio));
  }
}

// rescore_finish_kernel: per query, sort the kp re-scored keys, emit the exact top-k, run the proof.
// tau[q] is the raw accumulator of the kp-th survivor (-inf: fewer than kp rows exist, nothing is outside).
__global__ __launch_bounds__(256) void rescore_finish_kernel(const unsigned long long *__restrict__ buf, int S, int k,
                                                            int kp, const float *__restrict__ tau,
                                                            const float *__restrict__ qnorm,
                                                            const float *__restrict__ qinv,
                                                            const double *__restrict__ qshift, float c1, float c2,
                                                            const unsigned int *__restrict__ dmax_bits,
                                                            unsigned int *__restrict__ failed,
                                                            unsigned long long *__restrict__ out_top, int out_ld,
                                                            const float *__restrict__ qdelta, const float *__restrict__ qn16) {
  extern __shared__ __attribute__((aligned(16))) unsigned long long skeys[];
  const int q = blockIdx.x, t = threadIdx.x;
  int P = 64;
  while (P < kp) P <<= 1;
  const unsigned long long *row = buf + (size_t)q * S;
  for (int i = t; i < P; i += 256) skeys[i] = (i < kp) ? row[i] : 0ull;
  __syncthreads();
  bitonic_sort_desc<256>(skeys, P, t);
  for (int i = t; i < k; i += 256) out_top[(size_t)q * out_ld + i] = skeys[i];
  if (t == 0) {
    bool ok = true;
    const float a_last = tau[q];
    const double eps = h1_err_bound(qdelta[q], qn16[q], (double)__uint_as_float(dmax_bits[0]), (double)__uint_as_float(dmax_bits[3]), c1) +
                       (double)qnorm[q] * (double)c2 * __uint_as_float(dmax_bits[1]);
    if (a_last > -INFINITY) {  // the survivor list is full: there are documents outside it
      const unsigned long long kth = skeys[k - 1];
      ok = (kth != 0ull) && ((double)a_last * (double)qinv[q] + qshift[q] + eps < (double)key_score(kth));
    }
    // a bound that is not a finite number (inf / NaN rows or queries: the images then hold NaN accumulators, which every
    // comparison of the filter drops) proves nothing, full list or not
    if (!(eps < 1.0e38)) ok = false;
    if (!ok) failed[q] = 1u;  // keeps an overflow flag set by compact_kernel during the approximate pass
  }
}

// Diagnostics (mevi_ip_topk_set_profiling(2) only): how many keys the filter launch just finished appended per query --
// what its epilogue and the compaction pay for, and what the data decides (score density around the running threshold).
// out[0] += sum over queries, out[1] = max over queries and launches, out[2] += queries over the list's capacity.
__global__ __launch_bounds__(256) void count_candidates_kernel(const unsigned int *__restrict__ count, int nq, int cap,
                                                              unsigned long long *__restrict__ out) {
  const int q = blockIdx.x * 256 + threadIdx.x;
  const unsigned int c = q < nq ? count[q] : 0u;
  unsigned long long sum = c;
  unsigned int mx = c, over = c > (unsigned int)cap ? 1u : 0u;
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) {
    sum += ((unsigned long long)(unsigned int)__shfl_xor((int)(sum >> 32), off) << 32) | (unsigned int)__shfl_xor((int)sum, off);
    mx = max(mx, (unsigned int)__shfl_xor((int)mx, off));
    over += (unsigned int)__shfl_xor((int)over, off);
  }
  if ((threadIdx.x & 63) == 0 && sum != 0ull) {
    atomicAdd(out, sum);
    atomicMax(out + 1, (unsigned long long)mx);
    if (over) atomicAdd(out + 2, (unsigned long long)over);
  }
}

// ---------------------------------------------------------------------------
struct SearchState {
  unsigned long long *buf;  // [nq, S]
  unsigned int *count;      // [nq]
  float *tau;               // [nq]
  unsigned int *failed;     // [nq]
  float *tau_s;             // [nq]  the sampled threshold of the pass's last launch (run_pass), kept for its check
};

static size_t state_bytes(int64_t nq, const TopkGeom &g) {
  return align_up((size_t)nq * g.S * 8, 256) + 4 * align_up((size_t)nq * 4, 256);
}

static SearchState carve_state(char *&p, int64_t nq, const TopkGeom &g) {
  SearchState st;
  st.buf = reinterpret_cast<unsigned long long *>(p);
  p += align_up((size_t)nq * g.S * 8, 256);
  st.count = reinterpret_cast<unsigned int *>(p);
  p += align_up((size_t)nq * 4, 256);
  st.tau = reinterpret_cast<float *>(p);
  p += align_up((size_t)nq * 4, 256);
  st.failed = reinterpret_cast<unsigned int *>(p);
  p += align_up((size_t)nq * 4, 256);
  st.tau_s = reinterpret_cast<float *>(p);
  p += align_up((size_t)nq * 4, 256);
  return st;
}

thread_local double g_growth = 0.0;
thread_local int g_profile = 0;
thread_local mevi_ip_topk_stats g_stats = {0, 0, 0, 0.0, 0.0, 0.0, 0.0, 0.0, 0, 0, 0, 0};
thread_local std::vector<hipEvent_t> g_events;  // triples: before filter, after filter, after compact
thread_local std::vector<long long> g_chunk_rows;  // rows of the launch each triple brackets (profiling on)

static void profile_mark(hipStream_t stream) {
  if (!g_profile) return;
  hipEvent_t e;
  if (hipEventCreate(&e) != hipSuccess) return;
  (void)hipEventRecord(e, stream);
  g_events.push_back(e);
}

// call after the stream has been synchronised
static void profile_collect() {
  for (size_t i = 0; i + 3 <= g_events.size(); i += 3) {
    float f = 0.f, c = 0.f;
    (void)hipEventElapsedTime(&f, g_events[i], g_events[i + 1]);
    (void)hipEventElapsedTime(&c, g_events[i + 1], g_events[i + 2]);
    g_stats.filter_ms += f;
    g_stats.compact_ms += c;
    if (getenv("MEVI_IP_TOPK_TRACE") && i / 3 < g_chunk_rows.size())  // per-launch breakdown on stderr
      fprintf(stderr, "ip_topk launch %zu: %lld rows  filter %.3f ms  compact %.3f ms\n", i / 3, g_chunk_rows[i / 3], f, c);
  }
  g_chunk_rows.clear();
  for (hipEvent_t e : g_events) (void)hipEventDestroy(e);
  g_events.clear();
}

// Walk docs [0, nd) in chunks; returns number of filter launches, <0 on error.
static int64_t run_pass(const float *Q, int64_t nq, const float *D, int64_t nd, int dim,
                        const TopkGeom &g, uint32_t id_base, const SearchState &st, bool guaranteed,
                        hipStream_t stream, bool h1 = false, unsigned long long *cand = nullptr,
                        const float *row_scale = nullptr /* non-null: Q, D = the 8-bit images (<= 32 queries), per-row scales */,
                        const float *qub = nullptr /* ... and the queries' G_q */) {
  const long long total = nq * (long long)g.k;
  const long long init_n = total > nq ? total : nq;
  hipLaunchKernelGGL(init_state_kernel, dim3((unsigned)((init_n + 255) / 256)), dim3(256), 0, stream,
                     st.buf, st.count, st.tau, st.failed, (long long)nq, g.S, g.k);
  const size_t compact_lds = (size_t)g.S * 8 + 2048;  // keys + histogram
  if (compact_lds > 65536) {  // dynamic LDS beyond 64 KiB must be opted into
    if (hipFuncSetAttribute(reinterpret_cast<const void *>(compact_kernel),
                            hipFuncAttributeMaxDynamicSharedMemorySize, (int)compact_lds) != hipSuccess) {
      set_error("ip_topk: cannot raise dynamic LDS to %zu bytes", compact_lds);
      return -1;
    }
  }
  // Query-tile width: NI = 2 (128 query rows per workgroup).  NI = 4 (256 rows, 128
  // MFMAs per phase) was measured at the same MFMA-pipe utilisation (85.6 % vs 85.4 %)
  // and pads nq further, so only NI = 2 is instantiated.
  const int ni = 2;
  // the 16x16x32 pre-filter's query tile: 256 queries, or 128 / 64 when the whole search is that small (33 .. 128 queries:
  // faiss_search.profile's larger batches -- HBM-bound, and a 256-query tile multiplies mostly rows that do not exist).
  // MEVI_IP_FILTER_QT=256 pins the large tile (A/B; same lists)
  static const bool shape32_f = [] { const char *e = getenv("MEVI_IP_FILTER_MFMA"); return e && atoi(e) == 32; }();
  static const int qt_pin = [] { const char *e = getenv("MEVI_IP_FILTER_QT"); return e ? atoi(e) : 0; }();
  const bool k16shape = h1 && !shape32_f && dim % 64 == 0;
  int ni16 = 8;
  if (k16shape) {
    ni16 = nq <= 64 ? 2 : (nq <= 128 ? 4 : 8);
    if (qt_pin == 256 || qt_pin == 128 || qt_pin == 64) ni16 = qt_pin / 32;
    // a tile must hold at least as many 32-k units as the ring keeps in flight (h16_tile_stream: DEPTH = 5 / 4 / 3 at
    // NI = 2 / 4 / 8): the look-ahead reaches into the NEXT tile only.  dim <= 128 pads to four units, so the 64-query
    // tile (five in flight) is not available there -- it would request a unit past the tile's end
    const int nunits = dim / 32;
    if (ni16 == 2 && nunits < 5) ni16 = 4;
    if (ni16 == 4 && nunits < 4) ni16 = 8;
  }
  if (k16shape && dim / 32 < (ni16 == 2 ? 5 : ni16 == 4 ? 4 : 3)) {
    set_error("ip_topk: %d k units per tile are fewer than the filter ring's look-ahead", dim / 32);
    return -1;
  }
  const int qt = h1 ? (k16shape ? 32 * ni16 : H1_QT) : 64 * ni;
  const int n_qtiles = (int)((nq + qt - 1) / qt);
  // expected survivors per chunk = k * growth = cap / growth_div (default 3: 3x head-room over the mean, tens of standard
  // deviations for exchangeable row order; a query that still overflows takes the guaranteed path)
  double growth_div = 3.0;
  if (const char *e = getenv("MEVI_IP_TOPK_GROWTH_DIV")) {  // tuning hook, unset in production
    const double v = atof(e);
    if (v >= 1.05 && v <= 64.0) growth_div = v;
  }
  double growth = g_growth > 0.0 ? g_growth : (double)g.cap / (growth_div * g.k);
  if (growth < 1.0) growth = 1.0;
  // Round 6: the growth the candidate area ALLOWS is not always the growth that is cheapest.  A launch whose chunk is g x the rows
  // seen appends ~k g keys per query, and the pass needs ln(nd / first) / ln(1 + g) launches, so
  //     cost(g) ~ (nq k t_key g + t_launch) / ln(1 + g),   t_key ~ 60 ns per appended key, t_launch ~ 0.2 ms (launch + compaction)
  // (measured on the C2 search, profiles/r06_filter_launches.txt: growth 1.8 -> 1.0 = 8 -> 12 launches, 22.8 k -> 20.0 k keys per
  // query, 88.6 -> 87.0 and 87.9 -> 87.0 ms on two boxes).  Large batches with long lists want g ~ 1, everything else the
  // largest g its area allows (shards of a W-way search, small batches: launches dominate).  Not under the tuning hooks.
  if (g_growth <= 0.0 && !getenv("MEVI_IP_TOPK_GROWTH_DIV") && !guaranteed) {
    const double keys = (double)nq * (double)g.k * 60e-9 * 1e3, t_launch = 0.2;       // ms per unit g, ms per launch
    double best = growth, best_cost = (keys * growth + t_launch) / log(1.0 + growth);
    for (double cand = 1.0; cand < growth; cand += 0.25) {
      const double c = (keys * cand + t_launch) / log(1.0 + cand);
      if (c < best_cost * 0.99) best = cand, best_cost = c;                            // (ties go to fewer launches)
    }
    growth = best;
  }
  int64_t cap_docs = (g.cap / (2 * BM)) * (2 * BM);  // chunk that can never overflow, tile aligned
  if (const char *e = getenv("MEVI_IP_TOPK_FIRST_CHUNK")) {  // tuning hook (rows of the first chunk, <= the candidate area), unset in production
    const int64_t v = atoll(e) / (2 * BM) * (2 * BM);
    if (v >= 2 * BM && v < cap_docs) cap_docs = v;
  }
  size_t pp_lds = h1 ? h1_lds_bytes() + 8 * STASH_BYTES_PER_WAVE : pp_lds_bytes<2>();
  const bool ktail = (dim % BK) != 0;
  const void *fn = nullptr;
#define MEVI_PICK(NI_, T_) fn = reinterpret_cast<const void *>(ip_filter_kernel<NI_, T_>)
  // stationary-query streaming kernel: few queries, and the query tile + rings + stash fit the 160 KiB of LDS
  const bool small = row_scale != nullptr ||
                     (h1 && nq <= 32 && dim >= 160 && sm_lds_bytes(dim) + 8 * STASH_BYTES_PER_WAVE <= 160 * 1024 &&
                      !getenv("MEVI_IP_TOPK_NO_SMALL"));
  if (row_scale) fn = reinterpret_cast<const void *>(ip_filter_i8_small_kernel);   // (the caller checked the shape: i8_eligible)
  else if (small) fn = reinterpret_cast<const void *>(ip_filter_h1_small_kernel);
  else if (h1) {  // Q, D = f16 images, dim = padded dim.  MEVI_IP_FILTER_MFMA=32: the 32x32x16 form (A/B; same lists)
    fn = !k16shape ? reinterpret_cast<const void *>(ip_filter_h1_kernel)
         : ni16 == 8 ? reinterpret_cast<const void *>(ip_filter_h16_kernel<8>)
         : ni16 == 4 ? reinterpret_cast<const void *>(ip_filter_h16_kernel<4>)
                     : reinterpret_cast<const void *>(ip_filter_h16_kernel<2>);
  }
  else if (ktail) MEVI_PICK(2, true);
  else MEVI_PICK(2, false);
#undef MEVI_PICK
  if (small) pp_lds = sm_lds_bytes(dim) + 8 * STASH_BYTES_PER_WAVE;
  // opt in to > 64 KiB dynamic LDS (per device; cheap, so done on every call)
  if (hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)pp_lds) != hipSuccess) {
    set_error("ip_topk: cannot raise dynamic LDS to %zu bytes", pp_lds);
    return -1;
  }
  int n_cu = 256;
  {
    int dev = 0, v = 0;
    if (hipGetDevice(&dev) == hipSuccess &&
        hipDeviceGetAttribute(&v, hipDeviceAttributeMultiprocessorCount, dev) == hipSuccess && v >= 8)
      n_cu = v;
  }
  // MEVI_IP_FILTER_CUS=<n> (probe, profiles/r06_dense_overlap_probe.txt): the persistent filter takes n CUs instead of all,
  // leaving the rest to whatever else is in flight (a second search's re-scoring on another stream)
  static const int cu_cap = [] { const char *e = getenv("MEVI_IP_FILTER_CUS"); return e ? atoi(e) : 0; }();
  if (cu_cap >= 8 && cu_cap < n_cu) n_cu = cu_cap / 8 * 8;
  // sampled threshold for the pass's LAST launch (rank_tau_kernel above).  r ranks give the estimate +-1/sqrt(r); the launch is
  // sized for c k expected keys per query with (1 - 1/c) sqrt(r) >= 4.5 deviations between that and the k it needs, and
  // c k (1 + 8 / sqrt(r)) inside the candidate area.  MEVI_IP_SAMPLE_TAU=0: the geometric schedule to the end (A/B; same lists).
  // Used for searches of up to 1024 queries (MEVI_IP_SAMPLE_TAU=1: any size): measured on one box (profiles/r06_sampled_threshold.txt)
  // 64 / 128 / 255 queries at top-1000 3.54 / 3.56 / 4.46 -> 3.23 / 3.32 / 4.24 ms; at 6980 queries the launches it merges are
  // matrix-bound already (-0.8 ms of 88) and on a corpus whose clusters sit in runs of adjacent rows the estimate's variance is
  // that of ~r / 8 ranks: ~10 of 6980 queries are flagged, and their second pass (+2.3 ms) costs more than the merge returns.
  static const int sample_mode = [] { const char *e = getenv("MEVI_IP_SAMPLE_TAU"); return e ? atoi(e) : -1; }();
  const bool sample_on = sample_mode == 1 || (sample_mode != 0 && nq <= 1024);
  double s_c = 0.0;
  int64_t n_sample = 0;
  bool sampled = false;
  if (sample_on && !guaranteed && g.k <= 4096 && g.k >= 32) {
    const double r = g.k / 2 < 16 ? 16.0 : (g.k / 2 > 96 ? 96.0 : (double)(g.k / 2));
    double c = r >= 96.0 ? 3.0 : 8.0;
    const double c_fit = (double)g.cap / ((double)g.k * (1.0 + 8.0 / sqrt(r)));
    if (c_fit < c) c = c_fit;
    if (c >= 1.5 && (1.0 - 1.0 / c) * sqrt(r) >= 4.5) {
      s_c = c;
      n_sample = (int64_t)ceil(r * (double)nd / (c * (double)g.k));
    }
  }
  int64_t seen = 0, launches = 0;
  while (seen < nd) {
    int64_t chunk = cap_docs;
    if (!guaranteed && seen >= g.k) {
      int64_t grown = (int64_t)((double)seen * growth);
      grown = grown / (2 * BM) * (2 * BM);
      if (grown > chunk) chunk = grown;
    }
    if (chunk > nd - seen) chunk = nd - seen;
    double expect_per_q = seen >= g.k ? (double)g.k * (double)chunk / (double)seen : (double)chunk;
    if (s_c > 0.0 && !sampled && seen >= n_sample && seen >= g.k && chunk < nd - seen) {
      // enough rows are in and the schedule would need more than one further launch: estimate, then everything that is left
      const int64_t rank = (int64_t)ceil(s_c * (double)g.k * (double)seen / (double)nd);
      if (rank >= 16 && rank <= g.k) {
        hipLaunchKernelGGL(rank_tau_kernel<64>, dim3((unsigned)((nq + 3) / 4)), dim3(256), 0, stream, st.buf, (int)nq, g.S, g.k, (int)rank,
                           st.tau, st.tau_s);
        sampled = true;
        chunk = nd - seen;
        expect_per_q = s_c * (double)g.k;
      }
    }
    const int64_t n_dtiles = (chunk + 2 * BM - 1) / (2 * BM);  // doc-tile pairs (ping-pong kernel)
    const int64_t nwg = n_dtiles * n_qtiles;
    if (nwg > 0x7fffffffLL) {
      set_error("ip_topk: grid too large (%lld workgroups)", (long long)nwg);
      return -1;
    }
    profile_mark(stream);
{
      int nq_i = (int)nq, n_dp = (int)n_dtiles, n_qt = n_qtiles;
      long long d0 = (long long)seen, d1 = (long long)(seen + chunk);
      const float *tau_c = st.tau;
      void *args[] = {(void *)&Q, &nq_i, (void *)&D, &d0, &d1, &dim, (void *)&tau_c, (void *)&st.buf,
                      (void *)&st.count, (void *)&g.S, (void *)&g.k, (void *)&g.cap, &id_base, &n_qt, &n_dp};
      unsigned grid = (unsigned)nwg;
      if (h1) {  // persistent: one workgroup per CU (128 KiB of LDS each), a multiple of 8 so every XCD gets its share
        const int64_t per_xcd = (nwg + 7) / 8 < n_cu / 8 ? (nwg + 7) / 8 : n_cu / 8;
        grid = (unsigned)(8 * per_xcd);
      }
      // flush cadence of the 16x16 kernel's record stashes: with `seen` rows behind tau a chunk is expected to yield
      // k * chunk / seen candidates per query (exchangeable row order; every row in the first chunk), i.e. r per 64 x 128
      // wave tile; flush every T tiles with T r <= ~56 records (a stash holds 128)
      int flush_mask = H16_FLUSH_EVERY - 1;
      {
        const double per_q = expect_per_q;
        const double r = per_q / (double)chunk * 64.0 * 16.0 * (double)ni16;     // records per wave tile (64 rows x 16 NI queries)
        int T = H16_FLUSH_EVERY;
        while (T > 1 && T * r > 56.0) T >>= 1;
        flush_mask = T - 1;
      }
      void *args16[] = {(void *)&Q, &nq_i, (void *)&D, &d0, &d1, &dim, (void *)&tau_c, (void *)&st.buf,
                        (void *)&st.count, (void *)&g.S, (void *)&g.k, (void *)&g.cap, &id_base, &n_qt, &n_dp, &flush_mask};
      void *args_small[] = {(void *)&Q, &nq_i, (void *)&D, &d0, &d1, &dim, (void *)&tau_c, (void *)&st.buf,
                            (void *)&st.count, (void *)&g.S, (void *)&g.k, (void *)&g.cap, &id_base, (void *)&row_scale, (void *)&qub};
      if (small) {  // one workgroup per CU, each walking 256-row blocks of the chunk
        const int64_t nb = (chunk + 255) / 256;
        grid = (unsigned)(nb < n_cu ? nb : n_cu);
      }
      const bool k16 = k16shape && !small;
      if (hipLaunchKernel(fn, dim3(grid), dim3(PP_THREADS), small ? args_small : (k16 ? args16 : args), pp_lds, stream) != hipSuccess) {
        set_error("ip_topk: filter kernel launch failed");
        return -1;
      }
    }
    profile_mark(stream);
    if (cand)
      hipLaunchKernelGGL(count_candidates_kernel, dim3((unsigned)((nq + 255) / 256)), dim3(256), 0, stream, st.count, (int)nq, g.cap, cand);
    {
      const int nq_i = (int)nq;
      // one wave per query with the keys in registers first (MEVI_IP_TOPK_COMPACT=lds: the LDS kernel alone); what it leaves
      // -- queries with more than 4096 keys -- is the LDS kernel's, which returns at once for every query already done
      static const bool wave_compact = [] { const char *e = getenv("MEVI_IP_TOPK_COMPACT"); return !(e && strcmp(e, "lds") == 0); }();
      if (wave_compact)
        hipLaunchKernelGGL(compact_wave_kernel<64>, dim3((unsigned)((nq + 3) / 4)), dim3(256), 0, stream, st.buf, st.count, st.tau,
                           st.failed, nq_i, g.S, g.k, g.cap);
      hipLaunchKernelGGL(compact_kernel, dim3((unsigned)(nq < 8 * n_cu ? nq : 8 * n_cu)), dim3(256), compact_lds, stream,
                         st.buf, st.count, st.tau, st.failed, nq_i, g.S, g.k, g.cap);
    }
    if (sampled && seen + chunk >= nd)   // the sampled launch was the last: did k rows beat its threshold?
      hipLaunchKernelGGL(sample_check_kernel, dim3((unsigned)((nq + 255) / 256)), dim3(256), 0, stream, st.tau, st.tau_s, st.failed, (int)nq);
    profile_mark(stream);
    g_stats.filter_flops += 2.0 * (double)nq * (double)chunk * (double)dim;
    if (g_profile) g_chunk_rows.push_back((long long)chunk);
    seen += chunk;
    ++launches;
  }
  if (!h1)  // exact passes hand their lists out as ranked results (the f16 pass is re-scored and sorted later)
    hipLaunchKernelGGL(sort_lists_kernel, dim3((unsigned)nq), dim3(256), (size_t)next_pow2(g.k < 64 ? 64 : g.k) * 8, stream,
                       st.buf, g.S, g.k);
  if (hipGetLastError() != hipSuccess) {
    set_error("ip_topk: kernel launch failed");
    return -1;
  }
  return launches;
}

}  // namespace
}  // namespace mevi

using namespace mevi;

extern "C" size_t mevi_ip_topk_workspace_bytes(int64_t nq, int64_t dim, int64_t k) {
  if (nq <= 0 || k <= 0 || k > 4096 || dim <= 0) return 0;
  const TopkGeom g = make_geom((int)k);
  // main state + fallback state + gathered fallback queries + index list
  return 2 * state_bytes(nq, g) + align_up((size_t)nq * dim * 4, 256) + align_up((size_t)nq * 4, 256) + 256;
}

extern "C" void mevi_ip_topk_set_growth(double growth) { g_growth = growth; }
extern "C" void mevi_ip_topk_set_profiling(int enable) { g_profile = enable; }

extern "C" void mevi_ip_topk_get_stats(mevi_ip_topk_stats *out) {
  if (out) *out = g_stats;
}

extern "C" int mevi_ip_topk_f32(const float *q, int64_t nq, const float *docs, int64_t nd, int64_t dim,
                                int64_t k, int64_t id_offset, float *out_score, int64_t *out_id,
                                void *workspace, size_t workspace_bytes, void *stream_) {
  hipStream_t stream = reinterpret_cast<hipStream_t>(stream_);
  g_stats = {0, 0, 0, 0.0, 0.0, 0.0, 0.0, 0.0, 0, 0, 0, 0};
  for (hipEvent_t e : g_events) (void)hipEventDestroy(e);
  g_events.clear();
  MEVI_REQUIRE(nq >= 0 && nd >= 0 && dim > 0 && k > 0, MEVI_ERR_INVALID_ARG,
               "ip_topk: bad shape nq=%lld nd=%lld dim=%lld k=%lld", (long long)nq, (long long)nd,
               (long long)dim, (long long)k);
  if (nq == 0) return MEVI_OK;
  MEVI_REQUIRE(q && out_score && out_id && (docs || nd == 0), MEVI_ERR_INVALID_ARG, "ip_topk: null pointer");
  MEVI_REQUIRE(dim % 4 == 0, MEVI_ERR_UNSUPPORTED, "ip_topk: dim=%lld must be a multiple of 4", (long long)dim);
  MEVI_REQUIRE(k <= 4096, MEVI_ERR_UNSUPPORTED, "ip_topk: k=%lld > 4096 not supported", (long long)k);
  MEVI_REQUIRE(((uintptr_t)q % 16) == 0 && ((uintptr_t)docs % 16) == 0, MEVI_ERR_INVALID_ARG,
               "ip_topk: q/docs must be 16-byte aligned");
  MEVI_REQUIRE(id_offset >= 0 && id_offset + nd < 0xFFFFFFFFLL, MEVI_ERR_UNSUPPORTED,
               "ip_topk: id_offset + nd must be < 2^32-1");
  MEVI_REQUIRE(nq < (1LL << 31) && dim < (1LL << 24), MEVI_ERR_UNSUPPORTED, "ip_topk: nq/dim too large");
  const size_t need = mevi_ip_topk_workspace_bytes(nq, dim, k);
  MEVI_REQUIRE(workspace && workspace_bytes >= need, MEVI_ERR_WORKSPACE,
               "ip_topk: workspace %zu bytes < required %zu", workspace_bytes, need);
  MEVI_REQUIRE(((uintptr_t)workspace % 256) == 0, MEVI_ERR_INVALID_ARG, "ip_topk: workspace must be 256-byte aligned");

  const TopkGeom g = make_geom((int)k);
  char *p = reinterpret_cast<char *>(workspace);
  SearchState st = carve_state(p, nq, g);
  SearchState fb = carve_state(p, nq, g);
  float *qsub = reinterpret_cast<float *>(p);
  p += align_up((size_t)nq * dim * 4, 256);
  int *fidx = reinterpret_cast<int *>(p);

  int64_t launches = run_pass(q, nq, docs, nd, (int)dim, g, (uint32_t)id_offset, st, false, stream);
  if (launches < 0) return MEVI_ERR_HIP;
  g_stats.n_chunks = launches;

  // One sync: did any query overflow its candidate list?
  std::vector<unsigned int> failed((size_t)nq);
  MEVI_HIP_CHECK(hipMemcpyAsync(failed.data(), st.failed, (size_t)nq * 4, hipMemcpyDeviceToHost, stream));
  MEVI_HIP_CHECK(hipStreamSynchronize(stream));
  profile_collect();
  std::vector<int> idx;
  for (int64_t i = 0; i < nq; ++i)
    if (failed[(size_t)i]) idx.push_back((int)i);
  if (!idx.empty()) {
    const int64_t nf = (int64_t)idx.size();
    g_stats.n_failed_queries = nf;
    MEVI_HIP_CHECK(hipMemcpyAsync(fidx, idx.data(), (size_t)nf * 4, hipMemcpyHostToDevice, stream));
    hipLaunchKernelGGL(gather_rows_kernel, dim3((unsigned)nf), dim3(256), 0, stream, q, fidx, (int)nf, (int)dim, qsub);
    int64_t fl = run_pass(qsub, nf, docs, nd, (int)dim, g, (uint32_t)id_offset, fb, true, stream);
    if (fl < 0) return MEVI_ERR_HIP;
    g_stats.n_fallback_chunks = fl;
    hipLaunchKernelGGL(scatter_top_kernel, dim3((unsigned)nf), dim3(256), 0, stream, fb.buf, fidx, (int)nf, g.S, g.k, st.buf,
                       g.S);
    // idx (host) must outlive the async H2D copy
    MEVI_HIP_CHECK(hipStreamSynchronize(stream));
    profile_collect();
  }
  const long long total = nq * (long long)k;
  hipLaunchKernelGGL(finalize_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, stream, st.buf, g.S,
                     g.k, (long long)nq, out_score, reinterpret_cast<long long *>(out_id));
  MEVI_HIP_CHECK(hipGetLastError());
  return MEVI_OK;
}

// ---------------------------------------------------------------------------
// Indexed (f16 pre-filtered) search.  Index = centred f16 corpus image + centred row norms + column mean +
// shard scalars.
namespace {
struct IndexView {
  const float *image;        // f16 image of the shard, unit-major, whole 256-row blocks (held as raw bytes)
  const float *norms_c;      // [nd]  ||d - mu||
  const float *mu;           // [dimp]
  const unsigned int *bits;  // [0] max ||d - mu||, [1] max ||d||, [2] max |d_k - mu_k|, [3] max ||(d - mu) - f16 image row / S_d||  (float bits)
  const float *scal;         // [0] S_d, [1] 1 / S_d
  double *colsum;            // [dimp] build scratch
};
inline int64_t pad32(int64_t d) { return (d + 31) / 32 * 32; }
// k extent of the f16 images: whole 32-wide units, at least three (h1_tile_stream prefetches three units ahead)
// images are padded to whole PAIRS of 32-k units, at least four (the 16x16x32 tile stream alternates two fragment sets per unit
// pair and keeps three units in flight)
inline int64_t pad_k(int64_t d) { return (d + 63) / 64 * 64 < 128 ? 128 : (d + 63) / 64 * 64; }
inline size_t index_image_bytes(int64_t nd, int64_t dim) { return align_up((size_t)image_rows(nd) * pad_k(dim) * 2, 256); }
inline IndexView view_index(const void *index, int64_t nd, int64_t dim) {
  const char *p = reinterpret_cast<const char *>(index);
  IndexView v;
  v.image = reinterpret_cast<const float *>(p);
  p += index_image_bytes(nd, dim);
  v.norms_c = reinterpret_cast<const float *>(p);
  p += align_up((size_t)nd * 4, 256);
  v.mu = reinterpret_cast<const float *>(p);
  p += align_up((size_t)pad_k(dim) * 4, 256);
  v.bits = reinterpret_cast<const unsigned int *>(p);
  v.scal = reinterpret_cast<const float *>(p + 16);
  p += 256;
  v.colsum = reinterpret_cast<double *>(const_cast<char *>(p));
  return v;
}
// survivors kept per query: k plus a margin for the approximation error.  The margin has to cover the documents whose
// approximate score lies within the error bound of the k-th one; their number grows with the density of scores at rank
// k, i.e. with k -- a fixed floor of 128 made the re-scoring of a sharded search's short first-round lists (k/W + slack)
// the largest per-rank cost that does not shrink with the shard.  Unproven queries get the second pass (2 K').
// Round 6 re-measured the margin with the tighter bound (MEVI_IP_KPRIME_DIV, 6980 queries x 8.84 M rows, k = 1000; profiles/
// r06_filter_launches.txt): k / 8 (K' = 1152) is no faster on the i.i.d. corpus (86.7 vs 86.4 ms) and sends 1004 queries of the
// CLUSTERED corpus to the second pass (98.3 ms); k / 16 sends ~3000 queries of three corpora to the exact fallback (700 ms).  k / 4 stays.
inline int h1_kprime(int k) {
  static const int div = [] { const char *e = getenv("MEVI_IP_KPRIME_DIV"); const int v = e ? atoi(e) : 0; return v >= 2 && v <= 64 ? v : 4; }();  // tuning hook
  int extra = k / div < 48 ? 48 : k / div;
  return (k + extra + 63) / 64 * 64;
}
}  // namespace

extern "C" size_t mevi_ip_index_bytes(int64_t nd, int64_t dim) {
  if (nd < 0 || dim <= 0) return 0;
  return index_image_bytes(nd, dim) + align_up((size_t)nd * 4, 256) + align_up((size_t)pad_k(dim) * 4, 256) + 256 +
         align_up((size_t)pad_k(dim) * 8, 256);
}

extern "C" int mevi_ip_index_build_f32(const float *docs, int64_t nd, int64_t dim, void *index, size_t index_bytes,
                                       void *stream_) {
  hipStream_t stream = reinterpret_cast<hipStream_t>(stream_);
  MEVI_REQUIRE(nd >= 0 && dim > 0, MEVI_ERR_INVALID_ARG, "ip_index_build: bad shape");
  MEVI_REQUIRE(dim % 4 == 0, MEVI_ERR_UNSUPPORTED, "ip_index_build: dim %% 4 != 0");
  MEVI_REQUIRE(index && index_bytes >= mevi_ip_index_bytes(nd, dim), MEVI_ERR_WORKSPACE, "ip_index_build: index buffer too small");
  MEVI_REQUIRE(((uintptr_t)index % 256) == 0 && ((uintptr_t)docs % 16) == 0, MEVI_ERR_INVALID_ARG,
               "ip_index_build: index must be 256-byte, docs 16-byte aligned");
  IndexView v = view_index(index, nd, dim);
  const int dimp = (int)pad_k(dim);
  // mean, maxima, scale and the scratch sums all start from zero (an empty shard keeps mu = 0, S_d = 1)
  MEVI_HIP_CHECK(hipMemsetAsync(const_cast<float *>(v.mu), 0,
                                align_up((size_t)dimp * 4, 256) + 256 + align_up((size_t)dimp * 8, 256), stream));
  if (nd > 0) {
    MEVI_REQUIRE(docs, MEVI_ERR_INVALID_ARG, "ip_index_build: null docs");
    hipLaunchKernelGGL(colsum_kernel, dim3((unsigned)((nd + 511) / 512)), dim3(256), 0, stream, docs, (long long)nd,
                       (int)dim, v.colsum);
    hipLaunchKernelGGL(mean_kernel, dim3((unsigned)((dimp + 255) / 256)), dim3(256), 0, stream, v.colsum, (long long)nd,
                       (int)dim, dimp, const_cast<float *>(v.mu));
    hipLaunchKernelGGL(doc_stats_kernel, dim3((unsigned)((nd + 63) / 64)), dim3(256), 0, stream, docs, (long long)nd, (int)dim, v.mu,
                       const_cast<float *>(v.norms_c), const_cast<unsigned int *>(v.bits));
  }
  hipLaunchKernelGGL(doc_scale_kernel, dim3(1), dim3(64), 0, stream, v.bits, const_cast<float *>(v.scal));
  if (nd > 0)
    hipLaunchKernelGGL(split_docs_f16_kernel, dim3((unsigned)(image_rows(nd) / 4)), dim3(256), 0, stream, docs, (long long)nd,
                       (int)dim, dimp, v.mu, v.scal, reinterpret_cast<_Float16 *>(const_cast<float *>(v.image)),
                       const_cast<unsigned int *>(v.bits));
  MEVI_HIP_CHECK(hipGetLastError());
  return MEVI_OK;
}

namespace {
// Second f16 pass for the queries the first one could not prove: twice the survivors (0: not available for
// this k -- the sort + proof kernel holds at most 8192 keys in LDS), for at most a quarter of the batch.
inline int h1_kprime2(int k) {
  const int kp2 = 2 * h1_kprime(k);
  return kp2 <= 8192 ? kp2 : 0;
}
inline int64_t h1_second_pass_max(int64_t nq) { return (nq + 3) / 4; }
inline size_t h1_second_pass_bytes(int64_t nq, int64_t dim, int64_t k) {
  const int kp2 = h1_kprime2((int)k);
  if (kp2 == 0) return 0;
  const int64_t n2 = h1_second_pass_max(nq);
  // state + exact lists + f32 and f16 query rows + norm / scale / shift + row indices
  return state_bytes(n2, make_geom(kp2)) + align_up((size_t)n2 * k * 8, 256) + align_up((size_t)n2 * dim * 4, 256) +
         align_up((size_t)image_rows(n2) * pad_k(dim) * 2, 256) + 4 * align_up((size_t)(n2 + 1) * 4, 256) +
         align_up((size_t)n2 * 8, 256) + align_up((size_t)n2 * 4, 256);
}
}  // namespace

extern "C" size_t mevi_ip_topk_indexed_workspace_bytes(int64_t nq, int64_t dim, int64_t k) {
  if (nq <= 0 || k <= 0 || k > 4096 || dim <= 0) return 0;
  const TopkGeom gp = make_geom(h1_kprime((int)k), nq);
  // approx state (K' geometry) + exact top lists + f16 queries + per-query norm / scale / shift, the second
  // pass, then the exact-path workspace for the fallback
  return state_bytes(nq, gp) + align_up((size_t)nq * k * 8, 256) + align_up((size_t)image_rows(nq) * pad_k(dim) * 2, 256) +
         4 * align_up((size_t)(nq + 1) * 4, 256) + align_up((size_t)nq * 8, 256) + h1_second_pass_bytes(nq, dim, k) +
         mevi_ip_topk_workspace_bytes(nq, dim, k) + 256;
}

extern "C" int mevi_ip_topk_indexed_f32(const float *q, int64_t nq, const float *docs, const void *index, int64_t nd,
                                        int64_t dim, int64_t k, int64_t id_offset, float *out_score,
                                        int64_t *out_id, void *workspace, size_t workspace_bytes, void *stream_) {
  hipStream_t stream = reinterpret_cast<hipStream_t>(stream_);
  g_stats = {0, 0, 0, 0.0, 0.0, 0.0, 0.0, 0.0, 0, 0, 0, 0};
  for (hipEvent_t e : g_events) (void)hipEventDestroy(e);
  g_events.clear();
  MEVI_REQUIRE(nq >= 0 && nd >= 0 && dim > 0 && k > 0, MEVI_ERR_INVALID_ARG, "ip_topk_indexed: bad shape");
  if (nq == 0) return MEVI_OK;
  MEVI_REQUIRE(q && out_score && out_id && index && (docs || nd == 0), MEVI_ERR_INVALID_ARG, "ip_topk_indexed: null pointer");
  MEVI_REQUIRE(dim % 4 == 0 && k <= 4096, MEVI_ERR_UNSUPPORTED, "ip_topk_indexed: dim %% 4 != 0 or k > 4096");
  MEVI_REQUIRE(((uintptr_t)q % 16) == 0 && ((uintptr_t)docs % 16) == 0 && ((uintptr_t)index % 256) == 0,
               MEVI_ERR_INVALID_ARG, "ip_topk_indexed: misaligned pointer");
  MEVI_REQUIRE(id_offset >= 0 && id_offset + nd < 0xFFFFFFFFLL && nq < (1LL << 31), MEVI_ERR_UNSUPPORTED,
               "ip_topk_indexed: ids / nq out of range");
  const size_t need = mevi_ip_topk_indexed_workspace_bytes(nq, dim, k);
  MEVI_REQUIRE(workspace && workspace_bytes >= need && ((uintptr_t)workspace % 256) == 0, MEVI_ERR_WORKSPACE,
               "ip_topk_indexed: workspace %zu bytes < required %zu (or misaligned)", workspace_bytes, need);

  const int kp = h1_kprime((int)k);
  const TopkGeom gp = make_geom(kp, nq), g = make_geom((int)k);
  const int64_t dimp = pad_k(dim);
  IndexView iv = view_index(index, nd, dim);
  char *p = reinterpret_cast<char *>(workspace);
  SearchState st = carve_state(p, nq, gp);
  unsigned long long *top = reinterpret_cast<unsigned long long *>(p);  // [nq, k] exact keys
  p += align_up((size_t)nq * k * 8, 256);
  float *qimage = reinterpret_cast<float *>(p);  // f16 image of the queries, unit-major, whole 256-row blocks
  p += align_up((size_t)image_rows(nq) * dimp * 2, 256);
  float *qnorm = reinterpret_cast<float *>(p);  // [nq] + 1 slot for the observed error ratio
  p += align_up((size_t)(nq + 1) * 4, 256);
  float *qinv = reinterpret_cast<float *>(p);   // [nq] 1 / (S_q S_d)
  p += align_up((size_t)(nq + 1) * 4, 256);
  float *qdelta = reinterpret_cast<float *>(p);  // [nq] ||q - f16 image / S_q||  (measured)
  p += align_up((size_t)(nq + 1) * 4, 256);
  float *qn16 = reinterpret_cast<float *>(p);    // [nq] ||f16 image / S_q||
  p += align_up((size_t)(nq + 1) * 4, 256);
  double *qshift = reinterpret_cast<double *>(p);  // [nq] q.mu
  p += align_up((size_t)nq * 8, 256);
  char *second_ws = p;
  p += h1_second_pass_bytes(nq, dim, k);
  void *exact_ws = p;
  const size_t exact_ws_bytes = mevi_ip_topk_workspace_bytes(nq, dim, k);
  const float c1 = h1_cacc(dimp), c2 = h1_c2(dim);   // c1: the accumulation constant of h1_err_bound (the roundings are measured)
  // candidate counters of the main pass (profiling level 2): the spare 256 bytes behind the exact path's workspace
  unsigned long long *cand = g_profile >= 2 ? reinterpret_cast<unsigned long long *>(p + exact_ws_bytes) : nullptr;
  if (cand) MEVI_HIP_CHECK(hipMemsetAsync(cand, 0, 32, stream));

  hipLaunchKernelGGL(split_queries_f16_kernel, dim3((unsigned)(image_rows(nq) / 4)), dim3(256), 0, stream, q, (long long)nq,
                     (int)dim, (int)dimp, iv.mu, iv.scal, reinterpret_cast<_Float16 *>(qimage), qnorm, qinv, qshift, qdelta, qn16);
  int64_t launches = run_pass(qimage, nq, iv.image, nd, (int)dimp, gp, (uint32_t)id_offset, st, false, stream, true, cand);
  if (launches < 0) return MEVI_ERR_HIP;
  g_stats.n_chunks = launches;
  g_stats.filter_flops *= (double)dim / (double)dimp;  // algorithmic flops count dim, not the padding
  // exact re-scoring + verification
  int P = 64;
  while (P < kp) P <<= 1;
  unsigned int *err_bits = reinterpret_cast<unsigned int *>(qnorm + nq);  // spare slot behind the norms (256-byte padded)
  MEVI_HIP_CHECK(hipMemsetAsync(err_bits, 0, 4, stream));
  {
    const long long waves = nq * (long long)((kp + 63) / 64);
    // the k-th best approximate score per query -> the exclusion threshold of the re-scoring (st.tau_s is free after the pass).
    // MEVI_IP_RESCORE_ALL=1: re-score every survivor (A/B; same lists)
    static const bool rescore_all = [] { const char *e = getenv("MEVI_IP_RESCORE_ALL"); return e && atoi(e) == 1; }();
    const float *acc_k = nullptr;
    if (!rescore_all && kp <= 4096 && k < kp) {
      hipLaunchKernelGGL(rank_tau_kernel<64>, dim3((unsigned)((nq + 3) / 4)), dim3(256), 0, stream, st.buf, (int)nq, gp.S, kp, (int)k,
                         (float *)nullptr, st.tau_s);
      acc_k = st.tau_s;
    }
    hipLaunchKernelGGL(rescore_rows_kernel, dim3((unsigned)((waves + 3) / 4)), dim3(256), 0, stream, q, docs, (int)dim, st.buf,
                       gp.S, kp, (int)nq, (unsigned int)id_offset, qnorm, qinv, qshift, c1, c2, iv.bits, iv.norms_c, err_bits, qdelta, qn16,
                       acc_k);
    hipLaunchKernelGGL(rescore_finish_kernel, dim3((unsigned)nq), dim3(256), (size_t)P * 8, stream, st.buf, gp.S, (int)k, kp,
                       st.tau, qnorm, qinv, qshift, c1, c2, iv.bits, st.failed, top, (int)k, qdelta, qn16);
  }
  // the results are finalised BEFORE the host looks at the proof flags (the common case: every query proven); a repaired
  // list is finalised again below
  const long long total = nq * (long long)k;
  hipLaunchKernelGGL(finalize_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, stream, top, (int)k, (int)k,
                     (long long)nq, out_score, reinterpret_cast<long long *>(out_id));
  unsigned int err_host = 0;
  MEVI_HIP_CHECK(hipMemcpyAsync(&err_host, err_bits, 4, hipMemcpyDeviceToHost, stream));
  unsigned long long cand_host[3] = {0ull, 0ull, 0ull};
  if (cand) MEVI_HIP_CHECK(hipMemcpyAsync(cand_host, cand, 24, hipMemcpyDeviceToHost, stream));
  MEVI_HIP_CHECK(hipGetLastError());
  std::vector<unsigned int> failed((size_t)nq);
  MEVI_HIP_CHECK(hipMemcpyAsync(failed.data(), st.failed, (size_t)nq * 4, hipMemcpyDeviceToHost, stream));
  MEVI_HIP_CHECK(hipStreamSynchronize(stream));
  profile_collect();
  {
    float r;
    memcpy(&r, &err_host, 4);
    g_stats.max_err_ratio = r;
    g_stats.err_bound = 1.0;  // the ratio is observed error / proven bound
    g_stats.n_filter_candidates = (int64_t)cand_host[0];
    g_stats.max_launch_candidates = (int64_t)cand_host[1];
    g_stats.n_list_overflows = (int64_t)cand_host[2];
  }
  // approx-state overflow (adversarial order) is flagged by compact_kernel in the same array (bitwise or: both set 1)
  // Safety net for the proof itself: the bound must dominate every error actually observed on the re-scored
  // survivors (typically by 5-15x).  If an observation ever EXCEEDS it, the premise of the proofs is false and every query
  // takes the exact path.
  const bool bound_suspect = g_stats.max_err_ratio > 1.0;   // (0.5 while the bound was the worst case 2u ||q|| ||d - mu||: the measured
                                                            // bound is Cauchy-Schwarz on what the rounding did -- rows with few
                                                            // large coordinates legitimately come close to it)
  std::vector<int> idx;
  for (int64_t i = 0; i < nq; ++i)
    if (failed[(size_t)i] || bound_suspect) idx.push_back((int)i);
  const bool repaired = !idx.empty();
  const int kp2 = h1_kprime2((int)k);
  if (!idx.empty() && !bound_suspect && kp2 != 0 && (int64_t)idx.size() <= h1_second_pass_max(nq)) {
    // Second chance before the 7x slower exact path: the same f16 search for the unproven queries only, with twice
    // the survivors -- what a cluster of near-identical scores around the k-th needs (duplicated passages).
    const int64_t n2 = (int64_t)idx.size();
    g_stats.n_second_pass_queries = n2;
    const TopkGeom g2 = make_geom(kp2);
    char *w = second_ws;
    SearchState s2 = carve_state(w, h1_second_pass_max(nq), g2);
    unsigned long long *top2 = reinterpret_cast<unsigned long long *>(w);
    w += align_up((size_t)h1_second_pass_max(nq) * k * 8, 256);
    float *q2 = reinterpret_cast<float *>(w);
    w += align_up((size_t)h1_second_pass_max(nq) * dim * 4, 256);
    float *qimage2 = reinterpret_cast<float *>(w);
    w += align_up((size_t)image_rows(h1_second_pass_max(nq)) * dimp * 2, 256);
    float *qnorm2 = reinterpret_cast<float *>(w);
    w += align_up((size_t)(h1_second_pass_max(nq) + 1) * 4, 256);
    float *qinv2 = reinterpret_cast<float *>(w);
    w += align_up((size_t)(h1_second_pass_max(nq) + 1) * 4, 256);
    float *qdelta2 = reinterpret_cast<float *>(w);
    w += align_up((size_t)(h1_second_pass_max(nq) + 1) * 4, 256);
    float *qn162 = reinterpret_cast<float *>(w);
    w += align_up((size_t)(h1_second_pass_max(nq) + 1) * 4, 256);
    double *qshift2 = reinterpret_cast<double *>(w);
    w += align_up((size_t)h1_second_pass_max(nq) * 8, 256);
    int *idx2 = reinterpret_cast<int *>(w);
    MEVI_HIP_CHECK(hipMemcpyAsync(idx2, idx.data(), (size_t)n2 * 4, hipMemcpyHostToDevice, stream));
    hipLaunchKernelGGL(gather_rows_kernel, dim3((unsigned)n2), dim3(256), 0, stream, q, idx2, (int)n2, (int)dim, q2);
    hipLaunchKernelGGL(split_queries_f16_kernel, dim3((unsigned)(image_rows(n2) / 4)), dim3(256), 0, stream, q2, (long long)n2,
                       (int)dim, (int)dimp, iv.mu, iv.scal, reinterpret_cast<_Float16 *>(qimage2), qnorm2, qinv2, qshift2, qdelta2, qn162);
    const double keep_flops = g_stats.filter_flops;
    const int64_t l2 = run_pass(qimage2, n2, iv.image, nd, (int)dimp, g2, (uint32_t)id_offset, s2, false, stream, true);
    g_stats.filter_flops = keep_flops;
    if (l2 < 0) return MEVI_ERR_HIP;
    int P2 = 64;
    while (P2 < kp2) P2 <<= 1;
    const long long waves2 = n2 * (long long)((kp2 + 63) / 64);
    hipLaunchKernelGGL(rescore_rows_kernel, dim3((unsigned)((waves2 + 3) / 4)), dim3(256), 0, stream, q2, docs, (int)dim, s2.buf,
                       g2.S, kp2, (int)n2, (unsigned int)id_offset, qnorm2, qinv2, qshift2, c1, c2, iv.bits, iv.norms_c,
                       (unsigned int *)nullptr, qdelta2, qn162, (const float *)nullptr);
    hipLaunchKernelGGL(rescore_finish_kernel, dim3((unsigned)n2), dim3(256), (size_t)P2 * 8, stream, s2.buf, g2.S, (int)k, kp2,
                       s2.tau, qnorm2, qinv2, qshift2, c1, c2, iv.bits, s2.failed, top2, (int)k, qdelta2, qn162);
    // proven rows go to their place in `top` (unproven ones are overwritten by the exact path below)
    hipLaunchKernelGGL(scatter_top_kernel, dim3((unsigned)n2), dim3(256), 0, stream, top2, idx2, (int)n2, (int)k, (int)k, top,
                       (int)k);
    std::vector<unsigned int> failed2((size_t)n2);
    MEVI_HIP_CHECK(hipMemcpyAsync(failed2.data(), s2.failed, (size_t)n2 * 4, hipMemcpyDeviceToHost, stream));
    MEVI_HIP_CHECK(hipStreamSynchronize(stream));
    profile_collect();
    std::vector<int> still;
    for (int64_t i = 0; i < n2; ++i)
      if (failed2[(size_t)i]) still.push_back(idx[(size_t)i]);
    idx.swap(still);
  }
  if (!idx.empty()) {  // unproven or overflowed queries: exact f32 search of just those, scattered into `top`
    const int64_t nf = (int64_t)idx.size();
    g_stats.n_failed_queries = nf;
    char *e = reinterpret_cast<char *>(exact_ws);
    SearchState fb = carve_state(e, nq, g);
    e += state_bytes(nq, g);  // skip the second state of the exact workspace layout
    float *qsub = reinterpret_cast<float *>(e);
    e += align_up((size_t)nq * dim * 4, 256);
    int *fidx = reinterpret_cast<int *>(e);
    (void)exact_ws_bytes;
    MEVI_HIP_CHECK(hipMemcpyAsync(fidx, idx.data(), (size_t)nf * 4, hipMemcpyHostToDevice, stream));
    hipLaunchKernelGGL(gather_rows_kernel, dim3((unsigned)nf), dim3(256), 0, stream, q, fidx, (int)nf, (int)dim, qsub);
    const double keep_flops = g_stats.filter_flops;
    int64_t fl = run_pass(qsub, nf, docs, nd, (int)dim, g, (uint32_t)id_offset, fb, true, stream);
    g_stats.filter_flops = keep_flops;
    if (fl < 0) return MEVI_ERR_HIP;
    g_stats.n_fallback_chunks = fl;
    hipLaunchKernelGGL(scatter_top_kernel, dim3((unsigned)nf), dim3(256), 0, stream, fb.buf, fidx, (int)nf, g.S, g.k, top,
                       (int)k);
    MEVI_HIP_CHECK(hipStreamSynchronize(stream));
    profile_collect();
  }
  if (repaired)
    hipLaunchKernelGGL(finalize_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, stream, top, (int)k, (int)k,
                       (long long)nq, out_score, reinterpret_cast<long long *>(out_id));
  MEVI_HIP_CHECK(hipGetLastError());
  return MEVI_OK;
}

// ---------------------------------------------------------------------------
// The 8-bit image (searches of <= 32 queries; kernels and bound above ip_filter_i8_small_kernel).
namespace {
struct Index8View {
  const float *image;        // int8 image, unit-major (64-k units), whole 256-row blocks
  const float *scale;        // [image_rows(nd)] s_r
  const float *ynorm;        // [image_rows(nd)] ||y^_r|| (rounded up)
  const float *cscale;       // [dimp] c_k
  const unsigned int *bits;  // [0] max ||y^||, [1] max ||d|| (from the f16 index), [2] rho, [3] 2^-20 rho max s_r, [4] max ||y - y^||, [5] max s_r, [6] eta = max ||I_r||  (float bits)
  double *colsq;             // [dimp] build scratch
};
inline size_t index8_image_bytes(int64_t nd, int64_t dim) { return align_up((size_t)image_rows(nd) * pad_k(dim), 256); }
inline Index8View view_index8(const void *index8, int64_t nd, int64_t dim) {
  const char *p = reinterpret_cast<const char *>(index8);
  Index8View v;
  v.image = reinterpret_cast<const float *>(p);
  p += index8_image_bytes(nd, dim);
  v.scale = reinterpret_cast<const float *>(p);
  p += align_up((size_t)image_rows(nd) * 4, 256);
  v.ynorm = reinterpret_cast<const float *>(p);
  p += align_up((size_t)image_rows(nd) * 4, 256);
  v.cscale = reinterpret_cast<const float *>(p);
  p += align_up((size_t)pad_k(dim) * 4, 256);
  v.bits = reinterpret_cast<const unsigned int *>(p);
  p += 256;
  v.colsq = reinterpret_cast<double *>(const_cast<char *>(p));
  return v;
}
// survivors of the 8-bit pass.  A row's error term is ~0.2 of the score deviation (Gaussian rows): the K'-th key has to lie that
// far below the k-th exact score -- 2.3 k rows at k = 1000 of 8.8 M i.i.d. rows: K' = 3 k + 64.  Measured at MS MARCO size
// (profiles/r06_i8_small.txt): every list proven on the i.i.d., ANCE-scale and duplicates corpora; on the CLUSTERED corpus a
// third of the queries stay open in batches of 8 / 32 (the within-cluster spread of the scores is about the size of the term:
// the list would have to hold the query's whole cluster).  A floor of 2048 survivors proves those too but costs more than it
// returns: the chunk schedule grows with cap / K', 12 launches instead of 5-9 (one query top-10: 1.40 -> 1.88 ms, 32 queries
// 1.53 -> 2.66 ms against 2.5-2.6 through the f16 image) -- so the list stays short and a corpus that keeps failing is
// searched through the f16 image (DenseIndex keeps a running share of repeated searches).  MEVI_IP_I8_KPRIME_MUL / _MIN: hooks.
inline int h8_kprime(int k) {
  static const double mul = [] { const char *e = getenv("MEVI_IP_I8_KPRIME_MUL"); const double v = e ? atof(e) : 0.0; return v >= 1.25 && v <= 16.0 ? v : 3.0; }();
  static const int kmin = [] { const char *e = getenv("MEVI_IP_I8_KPRIME_MIN"); const int v = e ? atoi(e) : 0; return v >= 64 && v <= 4096 ? v / 64 * 64 : 64; }();
  int kp = ((int)(mul * (double)k) + 64 + 63) / 64 * 64;
  if (kp < kmin) kp = kmin;
  return kp;
}
// shapes the 8-bit pass takes: one query tile, four 64-k units at least (the ring's look-ahead), int32 digit sums below 2^24,
// the stationary tile + rings + stash inside the LDS, a survivor list the proof kernel can sort and the exclusion can rank
inline bool i8_eligible(int64_t nq, int64_t dim, int64_t k) {
  const int64_t dimp = pad_k(dim);
  return nq >= 1 && nq <= 32 && k >= 1 && k <= 4096 && dim % 4 == 0 && dimp >= 256 && dimp <= 1024 &&
         sm_lds_bytes((int)dimp) + 8 * STASH_BYTES_PER_WAVE <= 160 * 1024 && h8_kprime((int)k) <= 4096;
}
inline size_t i8_own_workspace_bytes(int64_t nq, int64_t dim, int64_t k) {
  const TopkGeom gp = make_geom(h8_kprime((int)k), nq);
  return state_bytes(nq, gp) + align_up((size_t)nq * k * 8, 256) + align_up((size_t)pad_k(dim) * 64, 256) +
         4 * align_up((size_t)(nq + 1) * 4, 256) + align_up((size_t)nq * 8, 256) + 512;
}
}  // namespace

extern "C" size_t mevi_ip_index8_bytes(int64_t nd, int64_t dim) {
  if (nd < 0 || dim <= 0) return 0;
  return index8_image_bytes(nd, dim) + 2 * align_up((size_t)image_rows(nd) * 4, 256) + align_up((size_t)pad_k(dim) * 4, 256) + 256 +
         align_up((size_t)pad_k(dim) * 8, 256);
}

extern "C" int mevi_ip_index8_build_f32(const float *docs, const void *index, int64_t nd, int64_t dim, void *index8,
                                        size_t index8_bytes, void *stream_) {
  hipStream_t stream = reinterpret_cast<hipStream_t>(stream_);
  MEVI_REQUIRE(nd >= 0 && dim > 0 && dim % 4 == 0, MEVI_ERR_INVALID_ARG, "ip_index8_build: bad shape");
  MEVI_REQUIRE(index && index8 && index8_bytes >= mevi_ip_index8_bytes(nd, dim), MEVI_ERR_WORKSPACE, "ip_index8_build: index buffer too small");
  MEVI_REQUIRE(((uintptr_t)index8 % 256) == 0 && ((uintptr_t)index % 256) == 0 && ((uintptr_t)docs % 16) == 0, MEVI_ERR_INVALID_ARG,
               "ip_index8_build: indexes must be 256-byte, docs 16-byte aligned");
  const IndexView iv = view_index(index, nd, dim);
  Index8View v = view_index8(index8, nd, dim);
  const int dimp = (int)pad_k(dim);
  // scales, maxima and the scratch sums start from zero
  MEVI_HIP_CHECK(hipMemsetAsync(const_cast<float *>(v.cscale), 0,
                                align_up((size_t)dimp * 4, 256) + 256 + align_up((size_t)dimp * 8, 256), stream));
  if (nd > 0) {
    MEVI_REQUIRE(docs, MEVI_ERR_INVALID_ARG, "ip_index8_build: null docs");
    hipLaunchKernelGGL(colsq_kernel, dim3((unsigned)((nd + 511) / 512)), dim3(256), 0, stream, docs, (long long)nd, (int)dim, iv.mu, v.colsq);
  }
  hipLaunchKernelGGL(colscale_kernel, dim3((unsigned)((dimp + 255) / 256)), dim3(256), 0, stream, v.colsq, (long long)nd, (int)dim, dimp,
                     const_cast<float *>(v.cscale));
  if (nd > 0)
    hipLaunchKernelGGL(split_docs_i8_kernel, dim3((unsigned)(image_rows(nd) / 4)), dim3(256), 0, stream, docs, (long long)nd, (int)dim, dimp,
                       iv.mu, v.cscale, reinterpret_cast<signed char *>(const_cast<float *>(v.image)), const_cast<float *>(v.scale),
                       const_cast<float *>(v.ynorm), const_cast<unsigned int *>(v.bits));
  hipLaunchKernelGGL(finish_bits8_kernel, dim3(1), dim3(64), 0, stream, iv.bits, const_cast<unsigned int *>(v.bits));
  MEVI_HIP_CHECK(hipGetLastError());
  return MEVI_OK;
}

extern "C" size_t mevi_ip_topk_indexed8_workspace_bytes(int64_t nq, int64_t dim, int64_t k) {
  const size_t inner = mevi_ip_topk_indexed_workspace_bytes(nq, dim, k);
  if (inner == 0) return 0;
  return (i8_eligible(nq, dim, k) ? i8_own_workspace_bytes(nq, dim, k) : 0) + inner;
}

extern "C" int mevi_ip_topk_indexed8_f32(const float *q, int64_t nq, const float *docs, const void *index, const void *index8,
                                         int64_t nd, int64_t dim, int64_t k, int64_t id_offset, float *out_score, int64_t *out_id,
                                         void *workspace, size_t workspace_bytes, void *stream_) {
  hipStream_t stream = reinterpret_cast<hipStream_t>(stream_);
  MEVI_REQUIRE(nq >= 0 && nd >= 0 && dim > 0 && k > 0, MEVI_ERR_INVALID_ARG, "ip_topk_indexed8: bad shape");
  if (nq == 0) return MEVI_OK;
  const size_t need = mevi_ip_topk_indexed8_workspace_bytes(nq, dim, k);
  MEVI_REQUIRE(need != 0, MEVI_ERR_UNSUPPORTED, "ip_topk_indexed8: unsupported shape");
  MEVI_REQUIRE(workspace && workspace_bytes >= need && ((uintptr_t)workspace % 256) == 0, MEVI_ERR_WORKSPACE,
               "ip_topk_indexed8: workspace %zu bytes < required %zu (or misaligned)", workspace_bytes, need);
  static const bool off = [] { const char *e = getenv("MEVI_IP_I8"); return e && atoi(e) == 0; }();
  const bool use8 = !off && index8 != nullptr && i8_eligible(nq, dim, k) && nd > 0;
  const size_t own = i8_eligible(nq, dim, k) ? i8_own_workspace_bytes(nq, dim, k) : 0;
  char *inner_ws = reinterpret_cast<char *>(workspace) + own;
  const size_t inner_bytes = workspace_bytes - own;
  if (!use8)
    return mevi_ip_topk_indexed_f32(q, nq, docs, index, nd, dim, k, id_offset, out_score, out_id, inner_ws, inner_bytes, stream_);

  MEVI_REQUIRE(q && out_score && out_id && index && docs, MEVI_ERR_INVALID_ARG, "ip_topk_indexed8: null pointer");
  MEVI_REQUIRE(((uintptr_t)q % 16) == 0 && ((uintptr_t)docs % 16) == 0 && ((uintptr_t)index % 256) == 0 && ((uintptr_t)index8 % 256) == 0,
               MEVI_ERR_INVALID_ARG, "ip_topk_indexed8: misaligned pointer");
  MEVI_REQUIRE(id_offset >= 0 && id_offset + nd < 0xFFFFFFFFLL, MEVI_ERR_UNSUPPORTED, "ip_topk_indexed8: ids out of range");
  g_stats = {0, 0, 0, 0.0, 0.0, 0.0, 0.0, 0.0, 0, 0, 0, 0};
  for (hipEvent_t e : g_events) (void)hipEventDestroy(e);
  g_events.clear();
  const int kp = h8_kprime((int)k);
  const TopkGeom gp = make_geom(kp, nq);
  const int64_t dimp = pad_k(dim);
  const IndexView iv = view_index(index, nd, dim);
  const Index8View v8 = view_index8(index8, nd, dim);
  char *p = reinterpret_cast<char *>(workspace);
  SearchState st = carve_state(p, nq, gp);
  unsigned long long *top = reinterpret_cast<unsigned long long *>(p);
  p += align_up((size_t)nq * k * 8, 256);
  float *qimage = reinterpret_cast<float *>(p);  // two int8 digit images of the 32-row query tile
  p += align_up((size_t)dimp * 64, 256);
  float *qnorm = reinterpret_cast<float *>(p);
  p += align_up((size_t)(nq + 1) * 4, 256);
  float *qinv = reinterpret_cast<float *>(p);
  p += align_up((size_t)(nq + 1) * 4, 256);
  float *qdelta = reinterpret_cast<float *>(p);
  p += align_up((size_t)(nq + 1) * 4, 256);
  float *qn16 = reinterpret_cast<float *>(p);
  p += align_up((size_t)(nq + 1) * 4, 256);
  double *qshift = reinterpret_cast<double *>(p);
  p += align_up((size_t)nq * 8, 256);
  float *qub = reinterpret_cast<float *>(p);       // [32] G_q
  p += 256;
  unsigned int *qbad = reinterpret_cast<unsigned int *>(p);   // [32] queries the 8-bit pass must not answer
  const float c1 = (float)(1.001 / 4194304.0), c2 = h1_c2(dim);   // c1: three roundings of the key's float form (< 2^-22)

  hipLaunchKernelGGL(split_queries_i8_kernel, dim3(8), dim3(256), 0, stream, q, (int)nq, (int)dim, (int)dimp, iv.mu, v8.cscale,
                     reinterpret_cast<signed char *>(qimage), qnorm, qinv, qshift, qdelta, qn16, v8.bits, qub, qbad);
  const int64_t launches = run_pass(qimage, nq, v8.image, nd, (int)dimp, gp, (uint32_t)id_offset, st, false, stream, true, nullptr, v8.scale, qub);
  if (launches < 0) return MEVI_ERR_HIP;
  int P = 64;
  while (P < kp) P <<= 1;
  unsigned int *err_bits = reinterpret_cast<unsigned int *>(qnorm + nq);
  MEVI_HIP_CHECK(hipMemsetAsync(err_bits, 0, 4, stream));
  {
    const long long waves = nq * (long long)((kp + 63) / 64);
    // every survivor is re-scored (the keys are upper bounds: no lower bound to exclude against)
    hipLaunchKernelGGL(rescore_rows_kernel, dim3((unsigned)((waves + 3) / 4)), dim3(256), 0, stream, q, docs, (int)dim, st.buf,
                       gp.S, kp, (int)nq, (unsigned int)id_offset, qnorm, qinv, qshift, c1, c2, v8.bits, v8.ynorm, err_bits, qdelta, qn16,
                       (const float *)nullptr, 1);
    hipLaunchKernelGGL(rescore_finish_kernel, dim3((unsigned)nq), dim3(256), (size_t)P * 8, stream, st.buf, gp.S, (int)k, kp,
                       st.tau, qnorm, qinv, qshift, c1, c2, v8.bits, st.failed, top, (int)k, qdelta, qn16);
  }
  const long long total = nq * (long long)k;
  hipLaunchKernelGGL(finalize_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, stream, top, (int)k, (int)k,
                     (long long)nq, out_score, reinterpret_cast<long long *>(out_id));
  unsigned int err_host = 0;
  MEVI_HIP_CHECK(hipMemcpyAsync(&err_host, err_bits, 4, hipMemcpyDeviceToHost, stream));
  MEVI_HIP_CHECK(hipGetLastError());
  unsigned int failed[32], bad[32];
  MEVI_HIP_CHECK(hipMemcpyAsync(failed, st.failed, (size_t)nq * 4, hipMemcpyDeviceToHost, stream));
  MEVI_HIP_CHECK(hipMemcpyAsync(bad, qbad, (size_t)nq * 4, hipMemcpyDeviceToHost, stream));
  MEVI_HIP_CHECK(hipStreamSynchronize(stream));
  profile_collect();
  float ratio;
  memcpy(&ratio, &err_host, 4);
  int64_t open = 0;
  for (int64_t i = 0; i < nq; ++i) open += (failed[i] || bad[i]) ? 1 : 0;
  if (!(ratio <= 1.0f)) open = nq;   // an observation above the bound: its premise is false (as the f16 search's safety net)
  g_stats.n_chunks = launches;
  g_stats.max_err_ratio = ratio;
  g_stats.err_bound = 1.0;
  g_stats.n_i8_queries = nq;
  g_stats.n_i8_unproven = open;
  if (open == 0) return MEVI_OK;
  // some list is not proven: the whole (small) batch through the f16 search, which has its own second pass and fallback
  const mevi_ip_topk_stats first = g_stats;
  const int rc = mevi_ip_topk_indexed_f32(q, nq, docs, index, nd, dim, k, id_offset, out_score, out_id, inner_ws, inner_bytes, stream_);
  g_stats.n_i8_queries = first.n_i8_queries;
  g_stats.n_i8_unproven = first.n_i8_unproven;
  g_stats.n_chunks += first.n_chunks;
  g_stats.filter_ms += first.filter_ms;
  g_stats.compact_ms += first.compact_ms;
  g_stats.filter_flops += first.filter_flops;
  return rc;
}

extern "C" size_t mevi_topk_merge_workspace_bytes(int64_t, int64_t, int64_t, int64_t) { return 0; }

extern "C" int mevi_topk_merge_f32(const float *scores, const int64_t *ids, int64_t nlists, int64_t nq,
                                   int64_t k_in, int64_t k_out, float *out_score, int64_t *out_id,
                                   void *, size_t, void *stream_) {
  hipStream_t stream = reinterpret_cast<hipStream_t>(stream_);
  MEVI_REQUIRE(nlists > 0 && nq >= 0 && k_in > 0 && k_out > 0, MEVI_ERR_INVALID_ARG, "topk_merge: bad shape");
  if (nq == 0) return MEVI_OK;
  MEVI_REQUIRE(scores && ids && out_score && out_id, MEVI_ERR_INVALID_ARG, "topk_merge: null pointer");
  MEVI_REQUIRE(nlists * k_in <= MAX_SORT, MEVI_ERR_UNSUPPORTED,
               "topk_merge: nlists*k_in=%lld > %d (merge hierarchically)", (long long)(nlists * k_in), MAX_SORT);
  MEVI_REQUIRE(k_out <= MAX_SORT, MEVI_ERR_UNSUPPORTED, "topk_merge: k_out too large");
  int P = 512;
  while (P < nlists * k_in) P <<= 1;
  if ((size_t)P * 8 > 65536) {
    MEVI_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void *>(merge_kernel),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, P * 8));
  }
  hipLaunchKernelGGL(merge_kernel, dim3((unsigned)nq), dim3(256), (size_t)P * 8, stream, scores,
                     reinterpret_cast<const long long *>(ids), (int)nlists, (long long)nq, (int)k_in, (int)k_out,
                     out_score, reinterpret_cast<long long *>(out_id));
  MEVI_HIP_CHECK(hipGetLastError());
  return MEVI_OK;
}

extern "C" int mevi_pack_lists_i64(const float *scores, const int64_t *ids, int64_t n, int64_t *packed, void *stream_) {
  MEVI_REQUIRE(n >= 0, MEVI_ERR_INVALID_ARG, "pack_lists: bad size");
  if (n == 0) return MEVI_OK;
  MEVI_REQUIRE(scores && ids && packed, MEVI_ERR_INVALID_ARG, "pack_lists: null pointer");
  hipLaunchKernelGGL(pack_lists_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, reinterpret_cast<hipStream_t>(stream_), scores,
                     reinterpret_cast<const long long *>(ids), (long long)n, reinterpret_cast<unsigned long long *>(packed));
  MEVI_HIP_CHECK(hipGetLastError());
  return MEVI_OK;
}

extern "C" int mevi_topk_merge_packed_f32(const int64_t *packed, int64_t nlists, int64_t nq, int64_t k_in, int64_t k_out,
                                          int truncated, float *out_score, int64_t *out_id, uint8_t *unproven, void *stream_) {
  hipStream_t stream = reinterpret_cast<hipStream_t>(stream_);
  MEVI_REQUIRE(nlists > 0 && nlists <= 64 && nq >= 0 && k_in > 0 && k_out > 0, MEVI_ERR_INVALID_ARG, "topk_merge_packed: bad shape");
  if (nq == 0) return MEVI_OK;
  MEVI_REQUIRE(packed && out_score && out_id && unproven, MEVI_ERR_INVALID_ARG, "topk_merge_packed: null pointer");
  int Lp = 32;
  while (Lp < k_in) Lp <<= 1;
  int P = 2 * Lp;
  while (P < nlists * Lp) P <<= 1;
  MEVI_REQUIRE(P <= MAX_SORT && k_out <= MAX_SORT, MEVI_ERR_UNSUPPORTED, "topk_merge_packed: %lld lists of %lld entries exceed the LDS sort",
               (long long)nlists, (long long)k_in);
  if ((size_t)P * 8 > 65536)
    MEVI_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void *>(merge_packed_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, P * 8));
  hipLaunchKernelGGL(merge_packed_kernel, dim3((unsigned)nq), dim3(256), (size_t)P * 8, stream,
                     reinterpret_cast<const unsigned long long *>(packed), (int)nlists, (long long)nq, (int)k_in, Lp, (int)k_out,
                     truncated, out_score, reinterpret_cast<long long *>(out_id), unproven);
  MEVI_HIP_CHECK(hipGetLastError());
  return MEVI_OK;
}
