// Shared host/device helpers for libmevi_hip.so (gfx950 only).
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdarg.h>

#include "../../include/mevi_hip.h"

namespace mevi {

// ---- error plumbing --------------------------------------------------------
void set_error(const char *fmt, ...);

#define MEVI_HIP_CHECK(expr)                                                     \
  do {                                                                           \
    hipError_t _e = (expr);                                                      \
    if (_e != hipSuccess) {                                                      \
      mevi::set_error("%s failed: %s (%s:%d)", #expr, hipGetErrorString(_e),     \
                      __FILE__, __LINE__);                                       \
      return MEVI_ERR_HIP;                                                       \
    }                                                                            \
  } while (0)

#define MEVI_REQUIRE(cond, code, ...)                                            \
  do {                                                                           \
    if (!(cond)) {                                                               \
      mevi::set_error(__VA_ARGS__);                                              \
      return (code);                                                             \
    }                                                                            \
  } while (0)

static inline size_t align_up(size_t x, size_t a) { return (x + a - 1) / a * a; }

// ---- order-preserving float <-> uint32 map ---------------------------------
// ord(a) < ord(b)  <=>  a < b for all non-NaN floats (-0.0 is folded onto +0.0).
__host__ __device__ static inline uint32_t f32_to_ord(float f) {
  f += 0.0f;  // -0.0 -> +0.0 so that equal scores get equal keys
  union { float f; uint32_t u; } c;
  c.f = f;
  return (c.u & 0x80000000u) ? ~c.u : (c.u | 0x80000000u);
}
__host__ __device__ static inline float ord_to_f32(uint32_t o) {
  union { float f; uint32_t u; } c;
  c.u = (o & 0x80000000u) ? (o & 0x7fffffffu) : ~o;
  return c.f;
}

// 64-bit composite key: sorting keys DESCENDING gives (score desc, id asc).
// Key 0 is the "empty slot" sentinel (only a negative NaN would map there and
// NaNs never pass the strict `score > tau` filter).
__host__ __device__ static inline uint64_t make_key(float score, uint32_t id) {
  return ((uint64_t)f32_to_ord(score) << 32) | (uint64_t)(0xFFFFFFFFu - id);
}
__host__ __device__ static inline float key_score(uint64_t key) {
  return ord_to_f32((uint32_t)(key >> 32));
}
__host__ __device__ static inline uint32_t key_id(uint64_t key) {
  return 0xFFFFFFFFu - (uint32_t)(key & 0xFFFFFFFFu);
}

// One compare-exchange stage of an in-LDS bitonic network over P 64-bit keys, NT threads.  A thread loads
// the keys of four pairs before it compares and stores them: the LDS round trips of a stage overlap instead
// of chaining (a plain loop re-reads the array it has just written, so every iteration waits for the last).
// MERGE: every pair orders descending (the merge stages of a descending sort); otherwise the direction
// follows the bitonic sort's block parity, (lo & size) == 0 -> descending.
template <int NT, bool MERGE>
__device__ __forceinline__ void bitonic_stage(unsigned long long *s, int P, int size, int stride, int t) {
  const int npairs = P >> 1;
  for (int i0 = t; i0 < npairs; i0 += 4 * NT) {
    unsigned long long a[4], b[4];
    int lo[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int i = i0 + u * NT;
      lo[u] = i < npairs ? 2 * i - (i & (stride - 1)) : -1;
      if (lo[u] >= 0) {
        a[u] = s[lo[u]];
        b[u] = s[lo[u] + stride];
      }
    }
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      if (lo[u] >= 0) {
        const bool desc = MERGE || ((lo[u] & size) == 0);
        if ((a[u] < b[u]) == desc) {
          s[lo[u]] = b[u];
          s[lo[u] + stride] = a[u];
        }
      }
    }
  }
  __syncthreads();
}

// In-LDS bitonic sort (descending) of P 64-bit keys by NT threads.
template <int NT>
__device__ inline void bitonic_sort_desc(unsigned long long *s, int P, int t) {
  for (int size = 2; size <= P; size <<= 1)
    for (int stride = size >> 1; stride > 0; stride >>= 1) bitonic_stage<NT, false>(s, P, size, stride, t);
}

// Merge stages only: s[0, P) bitonic (descending then ascending, or any rotation) -> sorted descending.
template <int NT>
__device__ inline void bitonic_merge_desc(unsigned long long *s, int P, int t) {
  for (int stride = P >> 1; stride > 0; stride >>= 1) bitonic_stage<NT, true>(s, P, P, stride, t);
}

}  // namespace mevi
