// Centroid update of k-means / residual-quantisation training (the offline index build):
// per-cluster means of the rows assigned to each centroid, plus the sum of squared norms, DETERMINISTIC
// (no float atomics: fixed-order two-stage reduction), so a seed reproduces a codebook bit for bit.
//
// The reference trains its codebook with scikit-learn (MEVI/pq.py:550-598: MiniBatchKMeans per level on the
// running residual); here the assignment step is mevi_rq_encode_f32 with a one-level codebook and this file
// is the update step.  Stage 1: every workgroup owns a contiguous range of rows and accumulates, per 256-column
// chunk, a [K][chunk] table in LDS -- one thread per column, rows in order; stage 2: the per-workgroup tables
// are summed in f64 in workgroup order.
#include "common.h"

namespace mevi {
namespace {

constexpr int KM_MAX_BLOCKS = 1024;

__global__ __launch_bounds__(256) void cluster_partial_kernel(const float *__restrict__ x, long long n, int dim,
                                                             const int *__restrict__ codes, long long code_stride,
                                                             int K, int dch, float *__restrict__ partial,
                                                             int *__restrict__ pcount, double *__restrict__ psq) {
  extern __shared__ __attribute__((aligned(16))) float acc[];  // [K][dch]
  __shared__ double sq_red[256];
  const int t = threadIdx.x, b = blockIdx.x, nb = gridDim.x;
  const long long per = (n + nb - 1) / nb;
  const long long r0 = (long long)b * per, r1 = r0 + per < n ? r0 + per : n;
  double sq = 0.0;
  for (int c0 = 0; c0 < dim; c0 += dch) {
    for (int i = t; i < K * dch; i += 256) acc[i] = 0.f;
    __syncthreads();
    if (t < dch && c0 + t < dim) {
      for (long long r = r0; r < r1; ++r) {
        const float v = x[(size_t)r * dim + c0 + t];
        acc[codes[r * code_stride] * dch + t] += v;
        sq += (double)v * (double)v;
      }
    }
    __syncthreads();
    for (int i = t; i < K * dch; i += 256) {
      const int k = i / dch, c = i - k * dch;
      if (c0 + c < dim) partial[((size_t)b * K + k) * dim + c0 + c] = acc[i];
    }
    __syncthreads();
  }
  // counts (integers: any order) and the block's sum of squares (fixed tree)
  for (int k = t; k < K; k += 256) pcount[(size_t)b * K + k] = 0;
  __syncthreads();
  for (long long r = r0 + t; r < r1; r += 256) atomicAdd(&pcount[(size_t)b * K + codes[r * code_stride]], 1);
  sq_red[t] = sq;
  __syncthreads();
  for (int off = 128; off > 0; off >>= 1) {
    if (t < off) sq_red[t] += sq_red[t + off];
    __syncthreads();
  }
  if (t == 0) psq[b] = sq_red[0];
}

__global__ __launch_bounds__(256) void cluster_finish_kernel(const float *__restrict__ partial,
                                                            const int *__restrict__ pcount,
                                                            const double *__restrict__ psq, int nb, int K, int dim,
                                                            const float *__restrict__ old,
                                                            float *__restrict__ centroids, int *__restrict__ counts,
                                                            double *__restrict__ stats) {
  const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
  if (i < (long long)K * dim) {
    const int k = (int)(i / dim);
    long long cnt = 0;
    double s = 0.0;
    for (int b = 0; b < nb; ++b) {
      cnt += pcount[(size_t)b * K + k];
      s += (double)partial[(size_t)b * K * dim + i];
    }
    centroids[i] = cnt > 0 ? (float)(s / (double)cnt) : (old ? old[i] : 0.f);  // an empty cluster keeps its centre
    if (i % dim == 0) counts[k] = (int)cnt;
  }
  if (i == 0 && stats) {
    double s = 0.0;
    for (int b = 0; b < nb; ++b) s += psq[b];
    stats[0] = s;  // sum of squared norms of the rows
  }
}

inline int km_blocks(int64_t n) {
  int64_t b = (n + 63) / 64;
  return (int)(b < 1 ? 1 : (b > KM_MAX_BLOCKS ? KM_MAX_BLOCKS : b));
}
inline int km_chunk(int64_t K) {  // columns per LDS table: K * chunk * 4 <= 64 KiB, at most 256 (one thread each)
  int64_t c = 16384 / K;
  return (int)(c > 256 ? 256 : c);
}

}  // namespace
}  // namespace mevi

using namespace mevi;

extern "C" size_t mevi_cluster_means_workspace_bytes(int64_t n, int64_t dim, int64_t K) {
  if (n < 0 || dim <= 0 || K <= 0) return 0;
  const size_t nb = (size_t)km_blocks(n);
  return align_up(nb * K * dim * 4, 256) + align_up(nb * K * 4, 256) + align_up(nb * 8, 256);
}

extern "C" int mevi_cluster_means_f32(const float *x, int64_t n, int64_t dim, const int32_t *codes,
                                      int64_t code_stride, int64_t K, const float *old_centroids, float *centroids,
                                      int32_t *counts, double *sum_sq, void *workspace, size_t workspace_bytes,
                                      void *stream_) {
  hipStream_t stream = reinterpret_cast<hipStream_t>(stream_);
  MEVI_REQUIRE(n >= 0 && dim > 0 && K > 0 && K <= 4096 && code_stride > 0, MEVI_ERR_INVALID_ARG, "cluster_means: bad shape");
  MEVI_REQUIRE(centroids && counts && (n == 0 || (x && codes)), MEVI_ERR_INVALID_ARG, "cluster_means: null pointer");
  const size_t need = mevi_cluster_means_workspace_bytes(n, dim, K);
  MEVI_REQUIRE(workspace && workspace_bytes >= need && ((uintptr_t)workspace % 256) == 0, MEVI_ERR_WORKSPACE,
               "cluster_means: workspace %zu bytes < required %zu (or misaligned)", workspace_bytes, need);
  const int nb = km_blocks(n), dch = km_chunk(K);
  MEVI_REQUIRE(dch >= 1, MEVI_ERR_UNSUPPORTED, "cluster_means: K too large");
  char *p = reinterpret_cast<char *>(workspace);
  float *partial = reinterpret_cast<float *>(p);
  p += align_up((size_t)nb * K * dim * 4, 256);
  int *pcount = reinterpret_cast<int *>(p);
  p += align_up((size_t)nb * K * 4, 256);
  double *psq = reinterpret_cast<double *>(p);
  hipLaunchKernelGGL(cluster_partial_kernel, dim3((unsigned)nb), dim3(256), (size_t)K * dch * 4, stream, x, (long long)n,
                     (int)dim, codes, (long long)code_stride, (int)K, dch, partial, pcount, psq);
  hipLaunchKernelGGL(cluster_finish_kernel, dim3((unsigned)((K * dim + 255) / 256)), dim3(256), 0, stream, partial, pcount,
                     psq, nb, (int)K, (int)dim, old_centroids, centroids, counts, sum_sq);
  MEVI_HIP_CHECK(hipGetLastError());
  return MEVI_OK;
}
