// Fine stage (K15): in-cluster re-ranking.
//
// Replaces the per-cluster Python loop of infer(): numpy fancy-indexing of the CPU memmap,
// .cuda() per <= 1024 rows, torch.matmul(q, P^T), torch.cat + torch.sort
// (MEVI/main_models.py:3915-4014; DocumentEncoder.generate, document_encoder.py:128-132,213-226).
//
//   pair_dot      out[i] = <A[ia[i]], B[ib[i]]>: one lane per pair, sequential f32 fmaf chain over
//                 k -- the same chain as the dense arm, so a (query, doc) pair gets the same score
//                 in the dense list and in the fine list.  Rows are staged through LDS in 32-wide
//                 k slabs (coalesced 128-byte row segments in, conflict-free per-lane rows out).
//                 Bound: HBM on the gathered 3 KB rows (random rows; algorithmic bytes 4*dim per pair).
//   segment_sort  per query: candidates -> (score desc, id asc) with the LDS bitonic sort on 64-bit keys.
#include "common.h"

namespace mevi {
namespace {

constexpr int PD_LD = 36;  // floats per staged row (16-byte aligned, conflict-free b128 reads)

__global__ __launch_bounds__(256) void pair_dot_kernel(const float *__restrict__ A, long long lda,
                                                      const long long *__restrict__ ia,
                                                      const float *__restrict__ B, long long ldb,
                                                      const long long *__restrict__ ib, long long n, int dim,
                                                      float *__restrict__ out) {
  __shared__ __attribute__((aligned(16))) float sa[4][64 * PD_LD];
  __shared__ __attribute__((aligned(16))) float sb[4][64 * PD_LD];
  const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
  const long long base = ((long long)blockIdx.x * 4 + wave) * 64;
  // staging duty of a lane: rows (lane>>3) + 8*i of the wave's 64 pairs, 4 floats at (lane&7)*4
  const int srow = lane >> 3, skq = (lane & 7) * 4;
  const float *ap[8];
  const float *bp[8];
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    long long p = base + srow + 8 * i;
    if (p > n - 1) p = n - 1;
    ap[i] = A + ia[p] * lda + skq;
    bp[i] = B + ib[p] * ldb + skq;
  }
  float acc = 0.f;
  const int nslab = (dim + 31) / 32;
  for (int s = 0; s < nslab; ++s) {
    const int kk = s * 32;
    const bool in = kk + skq < dim;  // dim % 4 == 0
    __syncthreads();
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      float4 va = make_float4(0.f, 0.f, 0.f, 0.f), vb = va;
      if (in) {
        va = *reinterpret_cast<const float4 *>(ap[i] + kk);
        vb = *reinterpret_cast<const float4 *>(bp[i] + kk);
      }
      *reinterpret_cast<float4 *>(&sa[wave][(srow + 8 * i) * PD_LD + skq]) = va;
      *reinterpret_cast<float4 *>(&sb[wave][(srow + 8 * i) * PD_LD + skq]) = vb;
    }
    __syncthreads();
    const float *ra = &sa[wave][lane * PD_LD];
    const float *rb = &sb[wave][lane * PD_LD];
#pragma unroll
    for (int k4 = 0; k4 < 32; k4 += 4) {
      const float4 x = *reinterpret_cast<const float4 *>(ra + k4);
      const float4 y = *reinterpret_cast<const float4 *>(rb + k4);
      acc = fmaf(x.x, y.x, acc);
      acc = fmaf(x.y, y.y, acc);
      acc = fmaf(x.z, y.z, acc);
      acc = fmaf(x.w, y.w, acc);
    }
  }
  const long long p = base + lane;
  if (p < n) out[p] = acc + 0.0f;
}

__global__ __launch_bounds__(256) void segment_sort_kernel(const float *__restrict__ scores,
                                                          const long long *__restrict__ ids,
                                                          const long long *__restrict__ seg, float *__restrict__ out_s,
                                                          long long *__restrict__ out_i) {
  extern __shared__ __attribute__((aligned(16))) unsigned long long skeys[];
  const long long s0 = seg[blockIdx.x], s1 = seg[blockIdx.x + 1];
  const int n = (int)(s1 - s0);
  if (n <= 0) return;
  const int t = threadIdx.x;
  int P = 64;
  while (P < n) P <<= 1;
  for (int i = t; i < P; i += 256)
    skeys[i] = (i < n) ? make_key(scores[s0 + i], (unsigned int)ids[s0 + i]) : 0ull;
  __syncthreads();
  bitonic_sort_desc<256>(skeys, P, t);
  for (int i = t; i < n; i += 256) {
    out_s[s0 + i] = key_score(skeys[i]);
    out_i[s0 + i] = (long long)key_id(skeys[i]);
  }
}

// --doc_multiclus > 1 (MEVI/main_models.py:3997-4011): a document reached through several beam clusters is listed once, its
// per-cluster scores aggregated -- 'add': uscores = 0, then += s in candidate order (sequential f32 adds); 'max': the maximum
// (from -inf) -- and the unique list is sorted by (score desc, id asc).  One workgroup per segment, everything in LDS:
// sort by (id, position), run heads aggregate their run in position order, second sort by the aggregated score.
__global__ __launch_bounds__(256) void segment_aggregate_sort_kernel(const float *__restrict__ scores,
                                                                    const long long *__restrict__ ids,
                                                                    const long long *__restrict__ seg, int mode,
                                                                    float *__restrict__ out_s, long long *__restrict__ out_i,
                                                                    int *__restrict__ out_n) {
  extern __shared__ __attribute__((aligned(16))) unsigned long long skeys[];
  const long long s0 = seg[blockIdx.x], s1 = seg[blockIdx.x + 1];
  const int n = (int)(s1 - s0);
  const int t = threadIdx.x;
  if (n <= 0) {
    if (t == 0) out_n[blockIdx.x] = 0;
    return;
  }
  int P = 64;
  while (P < n) P <<= 1;
  // descending sort of ~(id, pos): ascending (id, pos) order; padding = 0 sorts last
  for (int i = t; i < P; i += 256)
    skeys[i] = (i < n) ? ~(((unsigned long long)(unsigned int)ids[s0 + i] << 32) | (unsigned long long)(unsigned int)i) : 0ull;
  __syncthreads();
  bitonic_sort_desc<256>(skeys, P, t);
  __shared__ int n_unique;
  if (t == 0) n_unique = 0;
  __syncthreads();
  for (int i = t; i < n; i += 256) {
    const unsigned long long key = ~skeys[i];
    const unsigned int id = (unsigned int)(key >> 32);
    const bool head = i == 0 || (unsigned int)((~skeys[i - 1]) >> 32) != id;
    float agg = 0.f;
    long long oid = -1;
    if (head) {
      agg = mode == 0 ? 0.f : -INFINITY;
      for (int r = i; r < n; ++r) {
        const unsigned long long kr = ~skeys[r];
        if ((unsigned int)(kr >> 32) != id) break;
        const float sc = scores[s0 + (unsigned int)kr];
        agg = mode == 0 ? agg + sc : fmaxf(agg, sc);
      }
      oid = (long long)id;
      atomicAdd(&n_unique, 1);
    }
    out_s[s0 + i] = agg;      // scratch use of the output range: re-read below
    out_i[s0 + i] = oid;
  }
  __syncthreads();
  for (int i = t; i < P; i += 256) {
    unsigned long long key = 0ull;
    if (i < n && out_i[s0 + i] >= 0) key = make_key(out_s[s0 + i], (unsigned int)out_i[s0 + i]);
    skeys[i] = key;
  }
  __syncthreads();
  bitonic_sort_desc<256>(skeys, P, t);
  const int nu = n_unique;
  for (int i = t; i < nu; i += 256) {
    out_s[s0 + i] = key_score(skeys[i]);
    out_i[s0 + i] = (long long)key_id(skeys[i]);
  }
  if (t == 0) out_n[blockIdx.x] = nu;
}

}  // namespace
}  // namespace mevi

using namespace mevi;

extern "C" int mevi_segment_aggregate_sort_f32(const float *scores, const int64_t *ids, const int64_t *seg_offsets, int64_t nseg,
                                               int64_t max_seg_len, int mode, float *out_scores, int64_t *out_ids,
                                               int32_t *out_counts, void *stream) {
  MEVI_REQUIRE(nseg >= 0 && max_seg_len >= 0 && (mode == 0 || mode == 1), MEVI_ERR_INVALID_ARG, "segment_aggregate_sort: bad arguments");
  if (nseg == 0) return MEVI_OK;
  MEVI_REQUIRE(scores && ids && seg_offsets && out_scores && out_ids && out_counts, MEVI_ERR_INVALID_ARG, "segment_aggregate_sort: null pointer");
  MEVI_REQUIRE(max_seg_len <= 16384, MEVI_ERR_UNSUPPORTED, "segment_aggregate_sort: segment of %lld > 16384 entries", (long long)max_seg_len);
  int P = 64;
  while (P < max_seg_len) P <<= 1;
  if ((size_t)P * 8 > 65536)
    MEVI_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void *>(segment_aggregate_sort_kernel),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, P * 8));
  hipLaunchKernelGGL(segment_aggregate_sort_kernel, dim3((unsigned)nseg), dim3(256), (size_t)P * 8, (hipStream_t)stream, scores,
                     reinterpret_cast<const long long *>(ids), reinterpret_cast<const long long *>(seg_offsets), mode, out_scores,
                     reinterpret_cast<long long *>(out_ids), out_counts);
  MEVI_HIP_CHECK(hipGetLastError());
  return MEVI_OK;
}

extern "C" int mevi_pair_dot_f32(const float *a, int64_t lda, const int64_t *ia, const float *b, int64_t ldb,
                                 const int64_t *ib, int64_t n, int64_t dim, float *out, void *stream) {
  MEVI_REQUIRE(n >= 0 && dim > 0 && dim % 4 == 0 && lda % 4 == 0 && ldb % 4 == 0, MEVI_ERR_INVALID_ARG,
               "pair_dot: dim/lda/ldb must be multiples of 4");
  if (n == 0) return MEVI_OK;
  MEVI_REQUIRE(a && ia && b && ib && out, MEVI_ERR_INVALID_ARG, "pair_dot: null pointer");
  MEVI_REQUIRE(((uintptr_t)a % 16) == 0 && ((uintptr_t)b % 16) == 0, MEVI_ERR_INVALID_ARG,
               "pair_dot: a/b must be 16-byte aligned");
  hipLaunchKernelGGL(pair_dot_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, a,
                     (long long)lda, reinterpret_cast<const long long *>(ia), b, (long long)ldb,
                     reinterpret_cast<const long long *>(ib), (long long)n, (int)dim, out);
  MEVI_HIP_CHECK(hipGetLastError());
  return MEVI_OK;
}

extern "C" int mevi_segment_sort_desc_f32(const float *scores, const int64_t *ids, const int64_t *seg_offsets,
                                          int64_t nseg, int64_t max_seg_len, float *out_scores, int64_t *out_ids,
                                          void *stream) {
  MEVI_REQUIRE(nseg >= 0 && max_seg_len >= 0, MEVI_ERR_INVALID_ARG, "segment_sort: bad shape");
  if (nseg == 0 || max_seg_len == 0) return MEVI_OK;
  MEVI_REQUIRE(scores && ids && seg_offsets && out_scores && out_ids, MEVI_ERR_INVALID_ARG, "segment_sort: null pointer");
  MEVI_REQUIRE(max_seg_len <= 16384, MEVI_ERR_UNSUPPORTED,
               "segment_sort: segment of %lld entries > 16384 (split the cluster list)", (long long)max_seg_len);
  int P = 64;
  while (P < max_seg_len) P <<= 1;
  if ((size_t)P * 8 > 65536)
    MEVI_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void *>(segment_sort_kernel),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, P * 8));
  hipLaunchKernelGGL(segment_sort_kernel, dim3((unsigned)nseg), dim3(256), (size_t)P * 8, (hipStream_t)stream, scores,
                     reinterpret_cast<const long long *>(ids), reinterpret_cast<const long long *>(seg_offsets),
                     out_scores, reinterpret_cast<long long *>(out_ids));
  MEVI_HIP_CHECK(hipGetLastError());
  return MEVI_OK;
}
