// Residual-quantisation encode (K13): codes[n, M] = per level argmin_c sum_k (r_k - C[j][c][k])^2,
// r <- r - C[j][code_j].
//
// Replaces pq.get_rq_document_cluster / forward_rq with dist_mode 'l2'
// (MEVI/pq.py:281-305, 337-369; compute_scores :124-131), which the reference runs on
// the CPU in batches of 128 (main_models.py:3207-3212).
//
// Direct-difference form on the VALU (NOT the |c|^2 - 2 x.c GEMM form): every distance is
// the sequential f32 fmaf chain  d = fma(r_k - c_k, r_k - c_k, d), k = 0..dim-1, exactly what
// oracle/mevi_oracle.c computes, so codes are bit-identical to the oracle; ties go to the
// lowest centroid index.
//
// One launch per level.  A 256-thread workgroup owns 128 rows; each wave a 32-row x 32-
// centroid tile with a 4x4 register tile per lane (16 independent chains), K-slabs of 32
// staged through LDS (double buffered).  The residual is never stored: the staging step
// re-derives it from X and the row's previous codes with the reference's operation order
// ((x - c0) - c1) - ..., so a level costs one read of X (HBM) and the codebook stays in L2.
// Bound: VALU (2 lane-ops per (row, centroid, k)); LDS traffic is halved four times over by
// the register tile (8 b128 reads per 128 VALU instructions).

#include "common.h"

#include <math.h>

namespace mevi {
namespace {

constexpr int RQ_ROWS = 128;  // rows per workgroup
constexpr int RQ_CENTS = 32;  // centroids per chunk
constexpr int RQ_KS = 32;     // k slab
constexpr int RQ_LD = 36;     // floats per LDS row (16-byte aligned, conflict-free b128 reads)
constexpr int RQ_MAXM = 8;    // levels supported by the in-register code history

// LEVEL is a template parameter so the code-history loops unroll and stay in registers
// (a runtime-indexed history array lands in scratch: 20x slower).
// STORE = true: write -distance to neg_dist[row, c] instead of taking the argmin (pq.beam_search needs
// the whole score row: compute_scores, pq.py:124-131); X is then an explicit residual matrix (LEVEL 0).
// `rows` (optional): the kernel encodes X[rows[i]] for i < *nrows instead of rows 0..n-1 (the exact re-encode of the rows the
// matrix-core encoder could not decide, rq_fast.hip); codes are read and written at the rows' own positions.
template <int LEVEL, bool STORE>
__global__ __launch_bounds__(256, 2) void rq_level_kernel(const float *__restrict__ X, long long n, int dim,
                                                         const float *__restrict__ C, int M, int K,
                                                         int *__restrict__ codes, float *__restrict__ neg_dist,
                                                         const long long *__restrict__ rows = nullptr,
                                                         const unsigned int *__restrict__ nrows = nullptr) {
  constexpr int level = LEVEL;
  if (rows) {
    n = (long long)*nrows;
    if ((long long)blockIdx.x * RQ_ROWS >= n) return;
  }
  __shared__ __attribute__((aligned(16))) float xs[2][RQ_ROWS * RQ_LD];
  __shared__ __attribute__((aligned(16))) float cs[2][RQ_CENTS * RQ_LD];

  const int t = threadIdx.x;
  const int lane = t & 63;
  const int wave = t >> 6;
  const int ld = lane >> 3;  // row group 0..7   -> rows  32*wave + ld + 8*i
  const int lc = lane & 7;   // cent group 0..7  -> cents lc + 8*j
  const long long row0 = (long long)blockIdx.x * RQ_ROWS;

  // staging duty: 4 float4 of the row slab (rows srow + 32*i), 1 float4 of the centroid slab
  const int srow = t >> 3;
  const int skq = (t & 7) * 4;
  const float *xptr[4];
  int prev[4][LEVEL > 0 ? LEVEL : 1];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    long long r = row0 + srow + 32 * i;
    if (r > n - 1) r = n - 1;
    if (rows) r = rows[r];
    xptr[i] = X + (size_t)r * dim + skq;
#pragma unroll
    for (int j = 0; j < LEVEL; ++j) prev[i][j] = codes[(size_t)r * M + j];
  }
  const size_t level_stride = (size_t)K * dim;
  const int nslab = (dim + RQ_KS - 1) / RQ_KS;
  const int nchunk = (K + RQ_CENTS - 1) / RQ_CENTS;

  float4 rx[4], rc;
  auto gload = [&](int chunk, int s) {
    const int kk = s * RQ_KS + skq;
    const bool in = kk < dim;  // dim % 4 == 0
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
      if (in) {
        v = *reinterpret_cast<const float4 *>(xptr[i] + s * RQ_KS);
#pragma unroll
        for (int j = 0; j < LEVEL; ++j) {  // residual with the reference's operation order
          const float4 c = *reinterpret_cast<const float4 *>(C + j * level_stride + (size_t)prev[i][j] * dim + kk);
          v.x -= c.x; v.y -= c.y; v.z -= c.z; v.w -= c.w;
        }
      }
      rx[i] = v;
    }
    const int cent = chunk * RQ_CENTS + srow;
    rc = make_float4(0.f, 0.f, 0.f, 0.f);
    if (in && cent < K)
      rc = *reinterpret_cast<const float4 *>(C + level * level_stride + (size_t)cent * dim + kk);
  };
  auto lstore = [&](int b) {
#pragma unroll
    for (int i = 0; i < 4; ++i) *reinterpret_cast<float4 *>(&xs[b][(srow + 32 * i) * RQ_LD + skq]) = rx[i];
    *reinterpret_cast<float4 *>(&cs[b][srow * RQ_LD + skq]) = rc;
  };

  float best_d[4];
  int best_c[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    best_d[i] = INFINITY;
    best_c[i] = 0;
  }

  for (int chunk = 0; chunk < nchunk; ++chunk) {
    float acc[4][4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int j = 0; j < 4; ++j) acc[i][j] = 0.f;

    __syncthreads();  // previous chunk's readers are done with both buffers
    gload(chunk, 0);
    lstore(0);
    __syncthreads();
    for (int s = 0; s < nslab; ++s) {
      if (s + 1 < nslab) gload(chunk, s + 1);
      const float *px = &xs[s & 1][(32 * wave + ld) * RQ_LD];
      const float *pc = &cs[s & 1][lc * RQ_LD];
#pragma unroll 2
      for (int k4 = 0; k4 < RQ_KS; k4 += 4) {  // limited unroll: a full unroll hoists 64 float4 reads and spills
        float4 xv[4], cv[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) xv[i] = *reinterpret_cast<const float4 *>(px + 8 * i * RQ_LD + k4);
#pragma unroll
        for (int j = 0; j < 4; ++j) cv[j] = *reinterpret_cast<const float4 *>(pc + 8 * j * RQ_LD + k4);
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
          for (int j = 0; j < 4; ++j) {
            float d;
            d = xv[i].x - cv[j].x; acc[i][j] = fmaf(d, d, acc[i][j]);
            d = xv[i].y - cv[j].y; acc[i][j] = fmaf(d, d, acc[i][j]);
            d = xv[i].z - cv[j].z; acc[i][j] = fmaf(d, d, acc[i][j]);
            d = xv[i].w - cv[j].w; acc[i][j] = fmaf(d, d, acc[i][j]);
          }
      }
      if (s + 1 < nslab) lstore((s + 1) & 1);
      __syncthreads();
    }
    if (STORE) {
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const long long r = row0 + 32 * wave + ld + 8 * i;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          const int c = chunk * RQ_CENTS + lc + 8 * j;
          if (r < n && c < K) neg_dist[(size_t)r * K + c] = -acc[i][j];
        }
      }
      continue;
    }
    // running argmin: (distance, index) lexicographic, lowest index wins ties
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const int c = chunk * RQ_CENTS + lc + 8 * j;
        const float d = acc[i][j];
        if (c < K && (d < best_d[i] || (d == best_d[i] && c < best_c[i]))) {
          best_d[i] = d;
          best_c[i] = c;
        }
      }
  }
  if (STORE) return;
  // reduce over the 8 lanes (lc) that share a row group
#pragma unroll
  for (int i = 0; i < 4; ++i) {
#pragma unroll
    for (int off = 1; off < 8; off <<= 1) {
      const float od = __shfl_xor(best_d[i], off);
      const int oc = __shfl_xor(best_c[i], off);
      if (od < best_d[i] || (od == best_d[i] && oc < best_c[i])) {
        best_d[i] = od;
        best_c[i] = oc;
      }
    }
    const long long r = row0 + 32 * wave + ld + 8 * i;
    if (lc == 0 && r < n) codes[(size_t)(rows ? rows[r] : r) * M + level] = best_c[i];
  }
}

// The same level for a LIST of rows (the exact re-encode of the rows the matrix-core encoder could not decide, rq_fast.hip) --
// typically ten thousand of 8.8 M.  rq_level_kernel's 128-row workgroups left most of the device idle there (79 workgroups of
// ~0.45 ms each per level, behind 69 000 empty ones launched to cover a count only the device knows): here a workgroup takes
// 32 rows and its four waves split the centroid chunks (wave w: centroids 128 g + 32 w .. + 31 of group g), staged together
// (one float4 of the residual slab and four of the centroid slabs per thread), the waves' (distance, index) minima combined
// through LDS; the grid is fixed and walks the list.  Every distance is the same sequential fmaf chain, ties go to the lowest
// centroid: same codes.
template <int LEVEL>
__global__ __launch_bounds__(256, 2) void rq_rows32_kernel(const float *__restrict__ X, int dim, const float *__restrict__ C, int M, int K,
                                                          int *__restrict__ codes, const long long *__restrict__ rows,
                                                          const unsigned int *__restrict__ nrows) {
  constexpr int level = LEVEL;
  __shared__ __attribute__((aligned(16))) float xs[2][32 * RQ_LD];
  __shared__ __attribute__((aligned(16))) float cs[2][128 * RQ_LD];
  __shared__ float sbd[4][32];
  __shared__ int sbc[4][32];
  const long long n = (long long)*nrows;
  const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
  const int ld = lane >> 3, lc = lane & 7;
  const int srow = t >> 3, skq = (t & 7) * 4;
  const size_t level_stride = (size_t)K * dim;
  const int nslab = (dim + RQ_KS - 1) / RQ_KS;
  const int ngroup = (K + 127) / 128;
  for (long long blk = blockIdx.x; blk * 32 < n; blk += gridDim.x) {
    const long long row0 = blk * 32;
    long long r = row0 + srow;
    if (r > n - 1) r = n - 1;
    r = rows[r];
    const float *xptr = X + (size_t)r * dim + skq;
    int prev[LEVEL > 0 ? LEVEL : 1];
#pragma unroll
    for (int j = 0; j < LEVEL; ++j) prev[j] = codes[(size_t)r * M + j];
    float4 rx, rc[4];
    auto gload = [&](int g, int s) {
      const int kk = s * RQ_KS + skq;
      const bool in = kk < dim;
      float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
      if (in) {
        v = *reinterpret_cast<const float4 *>(xptr + s * RQ_KS);
#pragma unroll
        for (int j = 0; j < LEVEL; ++j) {  // residual with the reference's operation order
          const float4 c = *reinterpret_cast<const float4 *>(C + j * level_stride + (size_t)prev[j] * dim + kk);
          v.x -= c.x; v.y -= c.y; v.z -= c.z; v.w -= c.w;
        }
      }
      rx = v;
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const int cent = g * 128 + srow + 32 * i;
        rc[i] = make_float4(0.f, 0.f, 0.f, 0.f);
        if (in && cent < K) rc[i] = *reinterpret_cast<const float4 *>(C + level * level_stride + (size_t)cent * dim + kk);
      }
    };
    auto lstore = [&](int b) {
      *reinterpret_cast<float4 *>(&xs[b][srow * RQ_LD + skq]) = rx;
#pragma unroll
      for (int i = 0; i < 4; ++i) *reinterpret_cast<float4 *>(&cs[b][(srow + 32 * i) * RQ_LD + skq]) = rc[i];
    };
    float best_d[4];
    int best_c[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) best_d[i] = INFINITY, best_c[i] = 0;
    for (int g = 0; g < ngroup; ++g) {
      float acc[4][4];
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = 0.f;
      __syncthreads();  // the previous group's (and block's) readers are done with both buffers
      gload(g, 0);
      lstore(0);
      __syncthreads();
      for (int s = 0; s < nslab; ++s) {
        if (s + 1 < nslab) gload(g, s + 1);
        const float *px = &xs[s & 1][ld * RQ_LD];
        const float *pc = &cs[s & 1][(32 * wave + lc) * RQ_LD];
#pragma unroll 2
        for (int k4 = 0; k4 < RQ_KS; k4 += 4) {
          float4 xv[4], cv[4];
#pragma unroll
          for (int i = 0; i < 4; ++i) xv[i] = *reinterpret_cast<const float4 *>(px + 8 * i * RQ_LD + k4);
#pragma unroll
          for (int j = 0; j < 4; ++j) cv[j] = *reinterpret_cast<const float4 *>(pc + 8 * j * RQ_LD + k4);
#pragma unroll
          for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j) {
              float d;
              d = xv[i].x - cv[j].x; acc[i][j] = fmaf(d, d, acc[i][j]);
              d = xv[i].y - cv[j].y; acc[i][j] = fmaf(d, d, acc[i][j]);
              d = xv[i].z - cv[j].z; acc[i][j] = fmaf(d, d, acc[i][j]);
              d = xv[i].w - cv[j].w; acc[i][j] = fmaf(d, d, acc[i][j]);
            }
        }
        if (s + 1 < nslab) lstore((s + 1) & 1);
        __syncthreads();
      }
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          const int c = g * 128 + 32 * wave + lc + 8 * j;
          const float d = acc[i][j];
          if (c < K && (d < best_d[i] || (d == best_d[i] && c < best_c[i]))) best_d[i] = d, best_c[i] = c;
        }
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) {
#pragma unroll
      for (int off = 1; off < 8; off <<= 1) {
        const float od = __shfl_xor(best_d[i], off);
        const int oc = __shfl_xor(best_c[i], off);
        if (od < best_d[i] || (od == best_d[i] && oc < best_c[i])) best_d[i] = od, best_c[i] = oc;
      }
      if (lc == 0) sbd[wave][ld + 8 * i] = best_d[i], sbc[wave][ld + 8 * i] = best_c[i];
    }
    __syncthreads();
    if (t < 32) {
      float d = sbd[0][t];
      int c = sbc[0][t];
#pragma unroll
      for (int w = 1; w < 4; ++w) {
        const float od = sbd[w][t];
        const int oc = sbc[w][t];
        if (od < d || (od == d && oc < c)) d = od, c = oc;
      }
      if (row0 + t < n) codes[(size_t)rows[row0 + t] * M + level] = c;
    }
    // (the next block's first __syncthreads orders these reads before its stores)
  }
}

// out[r] = X[src[r]] - C[code[r]]: the residual hand-down of pq.beam_search (pq.py:690-693). One wave per row.
__global__ __launch_bounds__(256) void gather_sub_kernel(const float *__restrict__ X, const long long *__restrict__ src,
                                                        const float *__restrict__ C, const int *__restrict__ code,
                                                        long long n, int dim, float *__restrict__ out) {
  const long long r = (long long)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (r >= n) return;
  const int lane = threadIdx.x & 63;
  const float4 *x = reinterpret_cast<const float4 *>(X + (size_t)src[r] * dim);
  const float4 *c = reinterpret_cast<const float4 *>(C + (size_t)code[r] * dim);
  float4 *o = reinterpret_cast<float4 *>(out + (size_t)r * dim);
  for (int i = lane; i < dim / 4; i += 64) {
    const float4 a = x[i], b = c[i];
    o[i] = make_float4(a.x - b.x, a.y - b.y, a.z - b.z, a.w - b.w);
  }
}

}  // namespace

// exact encode of the rows listed in `rows[0, *nrows)` (device-side count; the grid covers max_rows): rq_fast.hip's fallback
int rq_encode_exact_rows(const float *x, int64_t dim, const float *codebook, int64_t M, int64_t K, int32_t *codes,
                         const long long *rows, const unsigned int *nrows, int64_t max_rows, hipStream_t stream) {
  MEVI_REQUIRE(M <= RQ_MAXM && dim % 4 == 0, MEVI_ERR_UNSUPPORTED, "rq_encode_exact_rows: shape");
  // a fixed grid walks the list (its length lives on the device): enough workgroups for ~64 k rows in one round
  int64_t grid = (max_rows + 31) / 32;
  if (grid > 2048) grid = 2048;
  for (int level = 0; level < (int)M; ++level) {
#define MEVI_RQ_LEVEL(L)                                                                                             \
  case L:                                                                                                            \
    hipLaunchKernelGGL((rq_rows32_kernel<L>), dim3((unsigned)grid), dim3(256), 0, stream, x, (int)dim, codebook, (int)M, \
                       (int)K, codes, rows, nrows);                                                                  \
    break;
    switch (level) {
      MEVI_RQ_LEVEL(0) MEVI_RQ_LEVEL(1) MEVI_RQ_LEVEL(2) MEVI_RQ_LEVEL(3)
      MEVI_RQ_LEVEL(4) MEVI_RQ_LEVEL(5) MEVI_RQ_LEVEL(6) MEVI_RQ_LEVEL(7)
    }
#undef MEVI_RQ_LEVEL
  }
  MEVI_HIP_CHECK(hipGetLastError());
  return MEVI_OK;
}

}  // namespace mevi

using namespace mevi;

extern "C" int mevi_rq_encode_f32(const float *x, int64_t n, int64_t dim, const float *codebook, int64_t M,
                                  int64_t K, int32_t *codes, void *stream_) {
  hipStream_t stream = reinterpret_cast<hipStream_t>(stream_);
  MEVI_REQUIRE(n >= 0 && dim > 0 && M > 0 && K > 0, MEVI_ERR_INVALID_ARG, "rq_encode: bad shape");
  if (n == 0) return MEVI_OK;
  MEVI_REQUIRE(x && codebook && codes, MEVI_ERR_INVALID_ARG, "rq_encode: null pointer");
  MEVI_REQUIRE(dim % 4 == 0, MEVI_ERR_UNSUPPORTED, "rq_encode: dim=%lld must be a multiple of 4", (long long)dim);
  MEVI_REQUIRE(M <= RQ_MAXM, MEVI_ERR_UNSUPPORTED, "rq_encode: M=%lld > %d levels", (long long)M, RQ_MAXM);
  MEVI_REQUIRE(((uintptr_t)x % 16) == 0 && ((uintptr_t)codebook % 16) == 0, MEVI_ERR_INVALID_ARG,
               "rq_encode: x/codebook must be 16-byte aligned");
  const int64_t nblk = (n + RQ_ROWS - 1) / RQ_ROWS;
  MEVI_REQUIRE(nblk <= 0x7fffffffLL, MEVI_ERR_UNSUPPORTED, "rq_encode: too many rows");
  for (int level = 0; level < (int)M; ++level) {
#define MEVI_RQ_LEVEL(L)                                                                                   \
  case L:                                                                                                  \
    hipLaunchKernelGGL((rq_level_kernel<L, false>), dim3((unsigned)nblk), dim3(256), 0, stream, x,         \
                       (long long)n, (int)dim, codebook, (int)M, (int)K, codes, (float *)nullptr);         \
    break;
    switch (level) {
      MEVI_RQ_LEVEL(0) MEVI_RQ_LEVEL(1) MEVI_RQ_LEVEL(2) MEVI_RQ_LEVEL(3)
      MEVI_RQ_LEVEL(4) MEVI_RQ_LEVEL(5) MEVI_RQ_LEVEL(6) MEVI_RQ_LEVEL(7)
    }
#undef MEVI_RQ_LEVEL
  }
  MEVI_HIP_CHECK(hipGetLastError());
  return MEVI_OK;
}

extern "C" int mevi_rq_neg_dist_f32(const float *x, int64_t n, int64_t dim, const float *centroids, int64_t K,
                                    float *neg_dist, void *stream_) {
  hipStream_t stream = reinterpret_cast<hipStream_t>(stream_);
  MEVI_REQUIRE(n >= 0 && dim > 0 && K > 0 && dim % 4 == 0, MEVI_ERR_INVALID_ARG, "rq_neg_dist: bad shape");
  if (n == 0) return MEVI_OK;
  MEVI_REQUIRE(x && centroids && neg_dist, MEVI_ERR_INVALID_ARG, "rq_neg_dist: null pointer");
  const int64_t nblk = (n + RQ_ROWS - 1) / RQ_ROWS;
  hipLaunchKernelGGL((rq_level_kernel<0, true>), dim3((unsigned)nblk), dim3(256), 0, stream, x, (long long)n, (int)dim,
                     centroids, 1, (int)K, (int *)nullptr, neg_dist);
  MEVI_HIP_CHECK(hipGetLastError());
  return MEVI_OK;
}

extern "C" int mevi_gather_sub_f32(const float *x, const int64_t *src, const float *centroids, const int32_t *code,
                                   int64_t n, int64_t dim, float *out, void *stream_) {
  MEVI_REQUIRE(n >= 0 && dim > 0 && dim % 4 == 0, MEVI_ERR_INVALID_ARG, "gather_sub: bad shape");
  if (n == 0) return MEVI_OK;
  MEVI_REQUIRE(x && src && centroids && code && out, MEVI_ERR_INVALID_ARG, "gather_sub: null pointer");
  hipLaunchKernelGGL(gather_sub_kernel, dim3((unsigned)((n + 3) / 4)), dim3(256), 0, reinterpret_cast<hipStream_t>(stream_), x,
                     reinterpret_cast<const long long *>(src), centroids, code, (long long)n, (int)dim, out);
  MEVI_HIP_CHECK(hipGetLastError());
  return MEVI_OK;
}
