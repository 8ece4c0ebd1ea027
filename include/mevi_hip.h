/*
 * mevi_hip.h -- C ABI of libmevi_hip.so, the MI355X (gfx950) kernels behind the
 * MEVI inference hot path.
 *
 * The reference (HugoZHL/MEVI) is pure Python and has no FFI layer of its own;
 * each entry point below replaces one third-party / torch call site of the
 * reference hot path (cited per function, paths relative to the reference
 * checkout).  The reference-side binding (ctypes) is shown in INTEGRATION.md.
 *
 * Conventions
 *  - every function returns 0 on success, a negative MEVI_ERR_* code otherwise;
 *    mevi_last_error() returns a thread-local message for the last failure.
 *  - all data pointers are caller-owned DEVICE pointers unless a parameter is
 *    documented as host memory.  No torch types cross this boundary.
 *  - `stream` is a hipStream_t passed as void* (NULL = default stream).  Work is
 *    enqueued on that stream; functions documented as "synchronising" wait for
 *    the stream once before returning.
 *  - kernels never allocate: scratch comes from the caller-provided workspace,
 *    sized by the matching *_workspace_bytes() query (pure host arithmetic).
 */
#ifndef MEVI_HIP_H
#define MEVI_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define MEVI_OK 0
#define MEVI_ERR_INVALID_ARG (-1)   /* bad shape / null pointer / misaligned pointer */
#define MEVI_ERR_UNSUPPORTED (-2)   /* valid request outside the implemented envelope */
#define MEVI_ERR_WORKSPACE (-3)     /* workspace too small */
#define MEVI_ERR_HIP (-4)           /* a HIP runtime call failed */

/* ABI version (bumped when a signature changes). */
int mevi_abi_version(void);

/* Thread-local description of the last error returned on this thread. */
const char *mevi_last_error(void);

/* ------------------------------------------------------------------------
 * Dense arm: exact inner-product top-k ("Flat" index).
 * Replaces faiss.index_factory(dim, "Flat", METRIC_INNER_PRODUCT) + index.add
 * + index.search at MEVI/faiss_search.py:13-21, and the in-cluster
 * torch.matmul + torch.sort at MEVI/main_models.py:3967-3968,4012.
 *
 *   q      f32 [nq, dim]  row-major queries
 *   docs   f32 [nd, dim]  row-major corpus shard
 *   out_score f32 [nq, k] descending; out_id i64 [nq, k] = id_offset + row
 *   Missing results (nd < k) are padded like faiss: id -1, score -FLT_MAX.
 *   Order: score descending, ties by ascending id (deterministic).
 *   Scores are bit-exact sequential fmaf chains over dim (k = 0..dim-1).
 *   Requirements: dim % 4 == 0, 1 <= k <= 4096, 16-byte aligned q/docs,
 *   id_offset + nd < 2^32 - 1.
 * Synchronising: waits on `stream` once (to check the rare overflow fallback).
 * ---------------------------------------------------------------------- */
size_t mevi_ip_topk_workspace_bytes(int64_t nq, int64_t dim, int64_t k);

int mevi_ip_topk_f32(const float *q, int64_t nq, const float *docs, int64_t nd,
                     int64_t dim, int64_t k, int64_t id_offset,
                     float *out_score, int64_t *out_id, void *workspace,
                     size_t workspace_bytes, void *stream);

/* Merge `nlists` per-shard top-k lists into one (the step after the RCCL
 * all-gather of the row-sharded dense arm; new in this build).
 *   scores f32 [nlists, nq, k_in], ids i64 [nlists, nq, k_in] (id -1 = padding)
 *   out    f32/i64 [nq, k_out], same ordering rule as mevi_ip_topk_f32.
 *   Requirement: nlists * k_in <= 16384.  Fully stream-ordered. */
size_t mevi_topk_merge_workspace_bytes(int64_t nlists, int64_t nq, int64_t k_in, int64_t k_out);

int mevi_topk_merge_f32(const float *scores, const int64_t *ids, int64_t nlists,
                        int64_t nq, int64_t k_in, int64_t k_out,
                        float *out_score, int64_t *out_id, void *workspace,
                        size_t workspace_bytes, void *stream);

/* ------------------------------------------------------------------------
 * Residual-quantisation encode: codes[n, M] (int32, values in [0, K)).
 * Replaces pq.get_rq_document_cluster / the index path of forward_rq with
 * dist_mode 'l2' (MEVI/pq.py:281-305, 337-369, 124-131), which the reference
 * evaluates on the CPU 128 rows at a time (MEVI/main_models.py:3207-3212).
 *   x f32 [n, dim], codebook f32 [M, K, dim] (rqcodebook*.pt layout, pq.py:470)
 *   per level: code = argmin_c sum_k (r_k - C[level][c][k])^2 (sequential fmaf
 *   chain, lowest index on ties), then r -= C[level][code].
 *   Requirements: dim % 4 == 0, M <= 8, 16-byte aligned x/codebook.
 * Fully stream-ordered; no workspace.
 * ---------------------------------------------------------------------- */
int mevi_rq_encode_f32(const float *x, int64_t n, int64_t dim, const float *codebook, int64_t M,
                       int64_t K, int32_t *codes, void *stream);

/* Test / tuning hooks for the dense arm (not part of the drop-in surface):
 * force the chunk growth factor (0 = default) and read back statistics of the
 * last mevi_ip_topk_f32 call on this thread. */
typedef struct mevi_ip_topk_stats {
  int64_t n_chunks;          /* filter launches of the main pass */
  int64_t n_failed_queries;  /* queries re-run through the guaranteed path */
  int64_t n_fallback_chunks; /* filter launches of the guaranteed path */
  double filter_ms;          /* sum of ip_filter_kernel durations (HIP events on the call's stream; profiling on) */
  double compact_ms;         /* sum of compact_kernel durations (profiling on) */
  double filter_flops;       /* algorithmic flops of those filter launches: 2 * nq * rows * dim */
} mevi_ip_topk_stats;
void mevi_ip_topk_set_growth(double growth);
void mevi_ip_topk_set_profiling(int enable); /* record HIP events around every filter/compact launch */
void mevi_ip_topk_get_stats(mevi_ip_topk_stats *out);

#ifdef __cplusplus
}
#endif
#endif /* MEVI_HIP_H */
