// The consumers' arithmetic on the device: ensemble_marco.py's cluster ranks + score combination + ranking
// (MEVI/ensemble_marco.py:174-238) and the first-hit ranks evaluate() turns into Recall / MRR (MEVI/evaluate.py:27-42,
// ensemble_marco.py:28-43).  The reference walks Python dicts: at MS MARCO size that is 6980 queries x (1000 dense +
// fine) entries = 14 M dict operations per (alpha, beta, gamma) point -- seconds, against 0.1 s for the search that
// produced the lists.  Here one workgroup takes one query and everything stays in LDS.
//
// All score arithmetic is IEEE float64 in the reference's order.  The only non-trivial operations, alpha / (beta * crank + 1)
// and (1 - gamma * alpha), depend on the cluster rank alone: the caller evaluates them in Python floats (a table of
// n_clusters + 1 doubles and one factor), so the device does one add and one multiply per entry -- no division, nothing
// a compiler could contract.
#include "common.h"

namespace mevi {
namespace {

constexpr int ENS_MAX = 8192;       // entries per query the LDS sort holds (a dense list of 1000 + a fine list fit 4x over)
constexpr int ENS_POS_BITS = 13;    // log2(ENS_MAX)
constexpr long long ENS_DOC_BIAS = 1LL << 46;

// ---- cluster ranks ---------------------------------------------------------------------------------------------------
// out[e] = LAST index r with beam[q][r] == codes[docs[e]] (the reference's dict keeps the last of a repeated cluster),
// n_clusters when none matches or the id is the -1 padding of a short dense list.  An id without a code row makes the
// reference raise KeyError: the smallest such entry index is reported through `bad`.
__global__ __launch_bounds__(256) void cluster_ranks_kernel(const int *__restrict__ codes, long long n_docs, int M,
                                                            const long long *__restrict__ docs,
                                                            const long long *__restrict__ seg, const int *__restrict__ beam,
                                                            int R, int n_clusters, int *__restrict__ out,
                                                            unsigned long long *__restrict__ bad) {
  extern __shared__ int sbeam[];
  const int q = blockIdx.x;
  for (int i = threadIdx.x; i < R * M; i += blockDim.x) sbeam[i] = beam[(size_t)q * R * M + i];
  __syncthreads();
  for (long long e = seg[q] + threadIdx.x; e < seg[q + 1]; e += blockDim.x) {
    const long long d = docs[e];
    int rank = n_clusters;
    if (d != -1) {
      if (d < 0 || d >= n_docs || codes[d * M] < 0) {
        atomicMin(bad, (unsigned long long)e);
      } else {
        for (int r = 0; r < R; ++r) {
          bool same = true;
          for (int j = 0; j < M; ++j) same = same && codes[d * M + j] == sbeam[r * M + j];
          if (same) rank = r;
        }
      }
    }
    out[e] = rank;
  }
}

// ---- ensemble -------------------------------------------------------------------------------------------------------
__device__ __forceinline__ unsigned long long f64_desc_key(double v) {   // ascending key <=> descending value; -0.0 == 0.0
  v += 0.0;
  unsigned long long u = (unsigned long long)__double_as_longlong(v);
  u = (u >> 63) ? ~u : (u | 0x8000000000000000ull);
  return ~u;
}

// bitonic sort of n_pad (power of two) keys in LDS, ascending; `pos` (optional) breaks ties ascending and moves along
template <bool WITH_POS>
__device__ __forceinline__ void lds_sort(unsigned long long *key, unsigned short *pos, int n_pad) {
  for (int k = 2; k <= n_pad; k <<= 1) {
    for (int j = k >> 1; j > 0; j >>= 1) {
      for (int i = threadIdx.x; i < n_pad; i += blockDim.x) {
        const int l = i ^ j;
        if (l > i) {
          const unsigned long long a = key[i], b = key[l];
          bool gt = a > b;
          if constexpr (WITH_POS) gt = gt || (a == b && pos[i] > pos[l]);
          if (gt == ((i & k) == 0)) {
            key[i] = b, key[l] = a;
            if constexpr (WITH_POS) {
              const unsigned short t = pos[i];
              pos[i] = pos[l], pos[l] = t;
            }
          }
        }
      }
      __syncthreads();
    }
  }
}

// One query per workgroup.  The combined list is the dense list followed by the fine list, cut to
// n = min(n_dense + n_fine, 2 n_dense) entries (the reference zips it with chain(cranks, cranks): ensemble_marco.py:226-232);
// entry j takes the cluster rank of position j (j < n_dense) or j - n_dense -- the fine list re-uses the DENSE list's ranks
// position by position (:200-208).  A document seen twice keeps its first position and its last score (a dict), the ranking
// is by descending score, equal scores in first-seen order (sorted() is stable).
__global__ __launch_bounds__(256) void ensemble_kernel(const long long *__restrict__ seg_d, const long long *__restrict__ docs_d,
                                                       const double *__restrict__ sc_d, const int *__restrict__ cr_d,
                                                       const long long *__restrict__ fine_row,
                                                       const long long *__restrict__ seg_f, const long long *__restrict__ docs_f,
                                                       const double *__restrict__ sc_f, int n_clusters,
                                                       const double *__restrict__ term, double punish,
                                                       const long long *__restrict__ out_seg, long long *__restrict__ out_docs,
                                                       int *__restrict__ out_n, int n_pad_max, int *__restrict__ err) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  unsigned long long *keyA = reinterpret_cast<unsigned long long *>(smem);
  unsigned long long *keyB = keyA + n_pad_max;
  unsigned short *posB = reinterpret_cast<unsigned short *>(keyB + n_pad_max);
  __shared__ int counts[257];
  const int q = blockIdx.x, t = threadIdx.x;
  const long long bd = seg_d[q];
  const int nd = (int)(seg_d[q + 1] - bd);
  long long bf = 0;
  int nf = 0;
  if (seg_f) {
    const long long fr = fine_row[q];
    bf = seg_f[fr];
    nf = (int)(seg_f[fr + 1] - bf);
  }
  const long long n_ll = seg_f ? min((long long)nd + nf, 2LL * nd) : (long long)nd;
  if (n_ll > n_pad_max) {      // the caller sized n_pad_max from the same lengths: cannot happen unless they changed
    if (t == 0) *err = 1, out_n[q] = 0;
    return;
  }
  const int n = (int)n_ll;
  int n_pad = 2;
  while (n_pad < n) n_pad <<= 1;
  auto doc_at = [&](int j) { return j < nd ? docs_d[bd + j] : docs_f[bf + j - nd]; };
  // phase A: (doc, position) ascending -> every document's occurrences side by side, in list order
  bool range_bad = false;
  for (int j = t; j < n_pad; j += blockDim.x) {
    unsigned long long k = ~0ull;
    if (j < n) {
      const long long d = doc_at(j);
      if (d < -ENS_DOC_BIAS || d >= ENS_DOC_BIAS) range_bad = true;
      k = ((unsigned long long)(d + ENS_DOC_BIAS) << ENS_POS_BITS) | (unsigned long long)j;
    }
    keyA[j] = k;
  }
  if (range_bad) *err = 2;
  __syncthreads();
  lds_sort<false>(keyA, nullptr, n_pad);
  // one entry per document: (key of the LAST occurrence's score, FIRST position); contiguous chunk per thread
  const int chunk = (n + blockDim.x - 1) / blockDim.x;
  const int lo = min(n, t * chunk), hi = min(n, lo + chunk);
  auto is_tail = [&](int i) { return i == n - 1 || (keyA[i + 1] >> ENS_POS_BITS) != (keyA[i] >> ENS_POS_BITS); };
  int mine = 0;
  for (int i = lo; i < hi; ++i) mine += is_tail(i) ? 1 : 0;
  counts[t + 1] = mine;
  if (t == 0) counts[0] = 0;
  __syncthreads();
  if (t == 0)
    for (int i = 1; i <= (int)blockDim.x; ++i) counts[i] += counts[i - 1];
  __syncthreads();
  const int nu = counts[blockDim.x];
  int c = counts[t];
  for (int i = lo; i < hi; ++i) {
    if (!is_tail(i)) continue;
    const unsigned long long doc_bits = keyA[i] >> ENS_POS_BITS;
    const int last = (int)(keyA[i] & (ENS_MAX - 1));
    int h = i;
    while (h > 0 && (keyA[h - 1] >> ENS_POS_BITS) == doc_bits) --h;
    const int first = (int)(keyA[h] & (ENS_MAX - 1));
    const double s = last < nd ? sc_d[bd + last] : sc_f[bf + last - nd];
    const int cr = cr_d[bd + (last < nd ? last : last - nd)];
    double v = s + term[cr];
    if (cr == n_clusters) v = v * punish;
    keyB[c] = f64_desc_key(v);
    posB[c] = (unsigned short)first;
    ++c;
  }
  int nu_pad = 2;
  while (nu_pad < nu) nu_pad <<= 1;
  __syncthreads();
  for (int i = nu + t; i < nu_pad; i += blockDim.x) keyB[i] = ~0ull, posB[i] = 0xffff;
  __syncthreads();
  lds_sort<true>(keyB, posB, nu_pad);
  const long long ob = out_seg[q];
  for (int i = t; i < nu; i += blockDim.x) out_docs[ob + i] = doc_at(posB[i]);
  if (t == 0) out_n[q] = nu;
}

// ---- first hits -----------------------------------------------------------------------------------------------------
// out[p] = first index of pair_doc[p] in list pair_row[p] (its first n entries; n = list_n[row] or the whole segment),
// -1 when absent or pair_row[p] < 0.  One wave per pair.
__global__ __launch_bounds__(256) void first_hits_kernel(const long long *__restrict__ lists, const long long *__restrict__ seg,
                                                         const int *__restrict__ list_n, const long long *__restrict__ pair_row,
                                                         const long long *__restrict__ pair_doc, long long n_pairs,
                                                         int *__restrict__ out) {
  const long long p = (long long)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (p >= n_pairs) return;
  const int lane = threadIdx.x & 63;
  const long long row = pair_row[p];
  int best = 0x7fffffff;
  if (row >= 0) {
    const long long b = seg[row];
    const long long n = list_n ? (long long)list_n[row] : seg[row + 1] - b;
    const long long d = pair_doc[p];
    for (long long i0 = 0; i0 < n && best == 0x7fffffff; i0 += 64) {
      const long long i = i0 + lane;
      int hit = (i < n && lists[b + i] == d) ? (int)i : 0x7fffffff;
      for (int o = 32; o > 0; o >>= 1) hit = min(hit, __shfl_xor(hit, o));
      best = hit;
    }
  }
  if (lane == 0) out[p] = best == 0x7fffffff ? -1 : best;
}

}  // namespace
}  // namespace mevi

using namespace mevi;

extern "C" int mevi_cluster_ranks_i32(const int32_t *codes, int64_t n_docs, int64_t M, const int64_t *docs,
                                      const int64_t *seg, int64_t nq, const int32_t *beam, int64_t R, int32_t n_clusters,
                                      int32_t *out_ranks, uint64_t *first_bad, void *stream) {
  MEVI_REQUIRE(nq >= 0 && M > 0 && R > 0 && n_docs >= 0 && R * M <= 8192, MEVI_ERR_INVALID_ARG, "cluster_ranks: bad shape");
  MEVI_REQUIRE(first_bad, MEVI_ERR_INVALID_ARG, "cluster_ranks: null pointer");
  MEVI_HIP_CHECK(hipMemsetAsync(first_bad, 0xff, sizeof(uint64_t), (hipStream_t)stream));
  if (nq == 0) return MEVI_OK;
  MEVI_REQUIRE(codes && docs && seg && beam && out_ranks, MEVI_ERR_INVALID_ARG, "cluster_ranks: null pointer");
  hipLaunchKernelGGL(cluster_ranks_kernel, dim3((unsigned)nq), dim3(256), (size_t)(R * M) * sizeof(int), (hipStream_t)stream,
                     codes, (long long)n_docs, (int)M, reinterpret_cast<const long long *>(docs),
                     reinterpret_cast<const long long *>(seg), beam, (int)R, (int)n_clusters, out_ranks,
                     reinterpret_cast<unsigned long long *>(first_bad));
  MEVI_HIP_CHECK(hipGetLastError());
  return MEVI_OK;
}

extern "C" int mevi_ensemble_rank_f64(const int64_t *seg_dense, const int64_t *docs_dense, const double *scores_dense,
                                      const int32_t *cranks_dense, const int64_t *fine_row, const int64_t *seg_fine,
                                      const int64_t *docs_fine, const double *scores_fine, int64_t nq, int64_t max_entries,
                                      int32_t n_clusters, const double *term, double punish, const int64_t *out_seg,
                                      int64_t *out_docs, int32_t *out_n, int32_t *err, void *stream) {
  MEVI_REQUIRE(nq >= 0 && max_entries >= 0 && n_clusters >= 0, MEVI_ERR_INVALID_ARG, "ensemble_rank: bad shape");
  MEVI_REQUIRE(max_entries <= ENS_MAX, MEVI_ERR_UNSUPPORTED, "ensemble_rank: %lld entries for one query > %d",
               (long long)max_entries, ENS_MAX);
  MEVI_REQUIRE(err, MEVI_ERR_INVALID_ARG, "ensemble_rank: null pointer");
  MEVI_HIP_CHECK(hipMemsetAsync(err, 0, sizeof(int32_t), (hipStream_t)stream));
  if (nq == 0) return MEVI_OK;
  MEVI_REQUIRE(seg_dense && docs_dense && scores_dense && cranks_dense && term && out_seg && out_docs && out_n,
               MEVI_ERR_INVALID_ARG, "ensemble_rank: null pointer");
  MEVI_REQUIRE(!seg_fine || (fine_row && docs_fine && scores_fine), MEVI_ERR_INVALID_ARG, "ensemble_rank: fine lists incomplete");
  int n_pad = 2;
  while (n_pad < max_entries) n_pad <<= 1;
  const size_t lds = (size_t)n_pad * (8 + 8 + 2);
  if (lds > 65536)
    MEVI_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void *>(ensemble_kernel),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
  hipLaunchKernelGGL(ensemble_kernel, dim3((unsigned)nq), dim3(256), lds, (hipStream_t)stream,
                     reinterpret_cast<const long long *>(seg_dense), reinterpret_cast<const long long *>(docs_dense),
                     scores_dense, cranks_dense, reinterpret_cast<const long long *>(fine_row),
                     reinterpret_cast<const long long *>(seg_fine), reinterpret_cast<const long long *>(docs_fine), scores_fine,
                     (int)n_clusters, term, punish, reinterpret_cast<const long long *>(out_seg),
                     reinterpret_cast<long long *>(out_docs), out_n, n_pad, err);
  MEVI_HIP_CHECK(hipGetLastError());
  return MEVI_OK;
}

extern "C" int mevi_first_hits_i64(const int64_t *lists, const int64_t *seg, const int32_t *list_n, const int64_t *pair_row,
                                   const int64_t *pair_doc, int64_t n_pairs, int32_t *out_rank, void *stream) {
  MEVI_REQUIRE(n_pairs >= 0, MEVI_ERR_INVALID_ARG, "first_hits: bad shape");
  if (n_pairs == 0) return MEVI_OK;
  MEVI_REQUIRE(seg && pair_row && pair_doc && out_rank, MEVI_ERR_INVALID_ARG, "first_hits: null pointer");
  hipLaunchKernelGGL(first_hits_kernel, dim3((unsigned)((n_pairs + 3) / 4)), dim3(256), 0, (hipStream_t)stream,
                     reinterpret_cast<const long long *>(lists), reinterpret_cast<const long long *>(seg), list_n,
                     reinterpret_cast<const long long *>(pair_row), reinterpret_cast<const long long *>(pair_doc),
                     (long long)n_pairs, out_rank);
  MEVI_HIP_CHECK(hipGetLastError());
  return MEVI_OK;
}
