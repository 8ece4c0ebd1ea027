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

// In-LDS bitonic sort (descending) of P 64-bit keys by NT threads.
template <int NT>
__device__ inline void bitonic_sort_desc(unsigned long long *s, int P, int t) {
  for (int size = 2; size <= P; size <<= 1) {
    for (int stride = size >> 1; stride > 0; stride >>= 1) {
      for (int i = t; i < (P >> 1); i += NT) {
        const int lo = 2 * i - (i & (stride - 1));
        const int hi = lo + stride;
        const bool desc = ((lo & size) == 0);
        const unsigned long long a = s[lo], b = s[hi];
        if ((a < b) == desc) {
          s[lo] = b;
          s[hi] = a;
        }
      }
      __syncthreads();
    }
  }
}


}  // namespace mevi
