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

/* Indexed form of the same search (same results, bit for bit): the corpus shard is converted once into
 * a centred, scaled float16 image f16((d - mu) * S) (the analogue of faiss `index.add`,
 * MEVI/faiss_search.py:19; mu = column mean of the shard); a search then selects k + margin candidates
 * per query with ONE f16 MFMA per product (16x the f32 MFMA rate), re-scores them with the exact f32
 * fmaf chain, and PROVES per query that no other row can enter the top-k:
 *   |approx + q.mu - exact| <= ||dq|| max||d - mu|| + ||q16|| max||dd|| + c_acc ||q16|| (max||d - mu|| + max||dd||) + c2 ||q|| max||d||,
 *   dq = q - its f16 image / S_q and dd = (d - mu) - the row's f16 image / S (both MEASURED while the images are written:
 *   per query, and the maximum over the shard kept in the index), c_acc = 4 dim 2^-24 (f32 accumulation of the exact f16
 *   products), c2 = dim 2^-24 (the exact chain) -- Cauchy-Schwarz on what the roundings did, about half the worst case
 *   (2^-10 + 2^-22) ||q|| ||d - mu|| of rounds 1-5 (round 6; DESIGN 4.1b');
 * survivors that provably cannot reach the top-k are not re-scored; searches of up to 1024 queries take the last launch of a
 * pass under a threshold estimated from the rows already seen, checked exactly afterwards (DESIGN 4.1b');
 * unproven queries are retried once with twice as many survivors (when they are at most a quarter of the
 * batch), what is still unproven is re-run through the exact f32 path.  `docs` (f32) is still needed for the
 * exact re-scoring.
 *   index buffer: mevi_ip_index_bytes(nd, dim) bytes, 256-byte aligned, filled by mevi_ip_index_build_f32.
 *   dim % 4 == 0.  Synchronising like mevi_ip_topk_f32. */
size_t mevi_ip_index_bytes(int64_t nd, int64_t dim);
int mevi_ip_index_build_f32(const float *docs, int64_t nd, int64_t dim, void *index, size_t index_bytes,
                            void *stream);
size_t mevi_ip_topk_indexed_workspace_bytes(int64_t nq, int64_t dim, int64_t k);
int mevi_ip_topk_indexed_f32(const float *q, int64_t nq, const float *docs, const void *index, int64_t nd,
                             int64_t dim, int64_t k, int64_t id_offset, float *out_score, int64_t *out_id,
                             void *workspace, size_t workspace_bytes, void *stream);

/* The same search for FEW queries (nq <= 32: faiss_search.profile's batches, MEVI/faiss_search.py:32-68, and a resident
 * service's single queries) through an optional 8-BIT image of the shard (round 6).  With one tile of query columns the
 * filter is bound by reading the corpus image from HBM; an int8 image is half the f16 one.  The integer matrix cores
 * accumulate exactly, so the approximate score's whole error is the two quantisations, both measured (per-column scales c
 * folded into the query, per-row scales s_r; queries in two int8 digits):
 *   exact <= t_q s_r (N + (rho ||w|| + eta ||w - w^||) / t_q) + q.mu + 3 ulp + c2 ||q|| max||d||,
 *   w = q * c, y = (d - mu) / c, y^ = s_r * int8 row I_r, w^ = t_q * 15-bit integer query, N = the exact int32 sum,
 *   rho = max over rows of ||y - y^|| / s_r (a row's quantisation error in units of its own step), eta = max ||I_r|| (both
 *   measured at build):
 * rows are ranked by that UPPER BOUND of their score (both quantisation terms are inside the key, per row through s_r), so a
 * row outside the survivors is bounded by the last survivor's key plus rounding only.
 * The bound is ~100x the f16 one, so 3 k + 64 survivors are re-scored (exact f32 chains) instead of 1.25 k; lists are proven
 * complete per query exactly as in the f16 search, and when any list of the batch stays open the call runs
 * mevi_ip_topk_indexed_f32 for the batch (same results either way, bit for bit).  Shapes outside the 8-bit pass (nq > 32,
 * padded dim < 256 or > 896, 3 k + 128 > 4096), index8 == NULL or MEVI_IP_I8=0: mevi_ip_topk_indexed_f32 directly.
 *   index8 buffer: mevi_ip_index8_bytes(nd, dim) bytes (~ nd * dim + 8 nd), 256-byte aligned, built from the docs and the
 *   f16 index of the same shard (its column mean).  Synchronising like mevi_ip_topk_f32. */
size_t mevi_ip_index8_bytes(int64_t nd, int64_t dim);
int mevi_ip_index8_build_f32(const float *docs, const void *index, int64_t nd, int64_t dim, void *index8, size_t index8_bytes,
                             void *stream);
size_t mevi_ip_topk_indexed8_workspace_bytes(int64_t nq, int64_t dim, int64_t k);
int mevi_ip_topk_indexed8_f32(const float *q, int64_t nq, const float *docs, const void *index, const void *index8, int64_t nd,
                              int64_t dim, int64_t k, int64_t id_offset, float *out_score, int64_t *out_id,
                              void *workspace, size_t workspace_bytes, void *stream);

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

/* The exchange of dense.sharded_ip_topk in its wire format.  mevi_pack_lists_i64: (score, id) pairs -> one i64 per entry,
 * score bits << 32 | id as u32 (id -1 = padding travels as 0xFFFFFFFF): 8 bytes per entry on xGMI instead of 12.
 * mevi_topk_merge_packed_f32: the all-gathered buffer packed i64 [nlists, nq, k_in] -- every list sorted (score desc,
 * id asc) as mevi_ip_topk_*_f32 returns them -- merged to out f32 / i64 [nq, k_out] with the ordering rule above (only
 * the merge stages of the sorting network run), and, with `truncated` != 0 (k_in < k: first round of the two-round
 * exchange), unproven[q] = 1 when some list's LAST entry ranks inside the merged top-k_out (deeper rows of that shard
 * may belong to it).  nlists <= 64, nlists * next_pow2(k_in) <= 16384. */
int mevi_pack_lists_i64(const float *scores, const int64_t *ids, int64_t n, int64_t *packed, void *stream);
int mevi_topk_merge_packed_f32(const int64_t *packed, int64_t nlists, int64_t nq, int64_t k_in, int64_t k_out,
                               int truncated, float *out_score, int64_t *out_id, uint8_t *unproven, void *stream);

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

/* The same codes, bit for bit, at the HBM rate (csrc/rq_fast.hip): one f16-MFMA product of the rows against ALL
 * M*K centroids (x centred on the level-0 mean, converted on the fly) + inter-centroid tables approximate every level's
 * distances; a rigorous error bound leaves, per row and level, either ONE possible argmin (taken) or a short candidate
 * list whose exact reference-order chains (the arithmetic above) decide; rows the speculation got wrong, rows with
 * more than 8 candidates or f16 overflow are re-encoded by the exact kernel.  Same call site as mevi_rq_encode_f32
 * (MEVI/pq.py:124-131, 281-305; MEVI/main_models.py:3182-3220).
 *   Requirements: dim % 32 == 0, 96 <= dim <= ~4000 (the kernel's LDS holds dim floats beside its rings), M <= 8,
 *   K <= 256 (else MEVI_ERR_UNSUPPORTED: call mevi_rq_encode_f32); workspace of mevi_rq_encode_fast_workspace_bytes (256-byte aligned; 0 = shape unsupported).
 * Stream-ordered, no host synchronisation.  mevi_rq_encode_fast_stats (tests / bench only; synchronises) reads
 * {ambiguity records, rows re-encoded exactly, ambiguous row-levels} of the last call that used `workspace`. */
size_t mevi_rq_encode_fast_workspace_bytes(int64_t n, int64_t dim, int64_t M, int64_t K);
int mevi_rq_encode_fast_f32(const float *x, int64_t n, int64_t dim, const float *codebook, int64_t M, int64_t K,
                            int32_t *codes, void *workspace, size_t workspace_bytes, void *stream);
int mevi_rq_encode_fast_stats(const void *workspace, int64_t n, int64_t dim, int64_t M, int64_t K, int64_t *out3,
                              void *stream);

/* ------------------------------------------------------------------------
 * Linear layer:  C[M,N] = act(A[M,K] . W[N,K]^T + bias[N]) + residual[M,N]
 * Replaces every torch.nn.Linear / torch.matmul of the T5 stacks on the hot path:
 * q/k/v/o and wi/wo (MEVI/transformers/modeling_t5.py:181-186, 217-220, 350-358, 412),
 * the adaptor's nn.TransformerDecoderLayer projections and adaptor_linear
 * (modeling_t5.py:1252-1255, 1677-1682).  W is the [out, in] weight of nn.Linear.
 *   lda/ldw/ldc/ldr: row strides in floats; bias, residual may be NULL;
 *   act: 0 = identity, 1 = ReLU, 2 = erf GELU (BERT towers, modeling_bert.py intermediate) -- applied before the residual add.
 *   Each output is the sequential f32 fmaf chain over k.
 *   Requirements: K, lda, ldw multiples of 4; A, W 16-byte aligned.
 * Fully stream-ordered; no workspace.
 * ---------------------------------------------------------------------- */
int mevi_gemm_nt_f32(const float *a, int64_t lda, const float *w, int64_t ldw, float *c, int64_t ldc,
                     int64_t m, int64_t n, int64_t k, const float *bias, const float *residual,
                     int64_t ldr, int act, void *stream);

/* ------------------------------------------------------------------------
 * Split-precision linear layer (the same reference call sites as mevi_gemm_nt_f32, for the T5 / BERT / adaptor
 * weights; csrc/gemm_split.hip).  Operands are images of f32 rows made by mevi_split_rows_f16:
 *   row r of x f32[m, k]  ->  img[r] = [hi (kp halves) | lo (kp halves)], kp = mevi_split_kp(k) = k rounded up to 32 (at least 64),
 *   x = 2^-exps[r] * (hi + lo) (exps int8, |e| <= 100), hi = f16(x 2^e), lo = f16(x 2^e - hi): 22 significant bits per element.
 * mevi_gemm_nt_split_f32 computes C = act(A.W^T + bias) + residual with three f16 MFMAs per product
 * (a_lo w_hi + a_hi w_lo + a_hi w_hi, f32 accumulate): relative error ~3 * 2^-22 per product instead of the exact
 * chain's 2^-24; every output depends only on its own two rows (same bits in any batch).  c f32 [m, n] (row stride
 * ldc); bias / residual (row stride ldr) may be NULL; act as mevi_gemm_nt_f32.
 *   Requirements: k, ldx multiples of 4; x and the images 16-byte aligned; img holds m * 2 * kp halves.
 * ---------------------------------------------------------------------- */
int64_t mevi_split_kp(int64_t k);
int mevi_split_rows_f16(const float *x, int64_t ldx, int64_t m, int64_t k, void *img, int8_t *exps, float *norms,
                        void *stream);   /* norms (may be NULL): l2 norm of every row, rounded up */
/* T5LayerNorm (mevi_rmsnorm_f32, modeling_t5.py:155-171) written straight into a split image: the normed states only
 * ever feed linear layers. */
int mevi_rmsnorm_split_f16(const float *x, int64_t ldx, const float *w, float eps, int64_t rows, int64_t dim,
                           void *img, int8_t *exps, float *norms, void *stream);
/* The same GEMM writing act(A.W^T + bias) as a split image [m, 2 * kp(n)] for a following GEMM (the FFN's
 * relu(x Wi^T), T5DenseReluDense modeling_t5.py:181-186 / BertIntermediate).  A row's exponent is fixed before its
 * columns exist, from |out| <= a_norm[r] * w_norm_max + bias_abs_max (Cauchy-Schwarz; w_norm_max = largest l2 norm of
 * a W row).  out_norm (may be NULL) receives a bound on the output rows' l2 norms.  When n % 32 != 0 the caller zeroes
 * out_img first (columns n .. kp(n) are not written). */
int mevi_gemm_nt_split_to_split(const void *a_img, const int8_t *a_exp, const float *a_norm, const void *w_img,
                                const int8_t *w_exp, float w_norm_max, int64_t m, int64_t n, int64_t k,
                                const float *bias, float bias_abs_max, int act, void *out_img, int8_t *out_exp,
                                float *out_norm, void *stream);
int mevi_gemm_nt_split_f32(const void *a_img, const int8_t *a_exp, const void *w_img, const int8_t *w_exp,
                           float *c, int64_t ldc, int64_t m, int64_t n, int64_t k, const float *bias,
                           const float *residual, int64_t ldr, int act, void *stream);

/* ------------------------------------------------------------------------
 * T5LayerNorm folded into the linear layers around it (round 5).  The reference normalises the residual stream before every
 * projection (T5LayerSelfAttention / T5LayerCrossAttention / T5LayerFF: `self.layer_norm(hidden_states)` then q|k|v / q / wi,
 * MEVI/transformers/modeling_t5.py:155-171, 189-201, 421-445, 452-480); a T5LayerNorm is a per-row scale times a per-column
 * weight, so   rmsnorm(x) W^T = rsqrt(mean(x^2) + eps) * (x (W (.) w_ln)^T):
 *   - the weights are held as W (.) w_ln (the caller folds them at load),
 *   - the residual stream travels as f32 rows + their split image + a bound on max |row| + the sums of squares of every row's
 *     16-column blocks (ssq f32 [m, n / 16]), written by the GEMM that produces it (mevi_gemm_nt_split_residual_stream: the
 *     o / wo projections' `hidden_states + dropout(y)`, :200, :444, :479) or, where it starts, by mevi_split_rows_ssq_f16,
 *   - the consuming projection multiplies the image and scales each row of the product (mevi_gemm_nt_split_normed_*): the
 *     scale comes from the block sums, blocks added in index order (mevi_row_rscale_f32 states the formula; GEMMs on the
 *     latency kernels compute it themselves, the tile stream takes it from `rscale_ws`, filled by one small launch).
 * No pass over the rows exists any more just to normalise them (mevi_rmsnorm_split_f16 read and wrote 4 B per element).
 * The product is the same value up to f32 rounding (the scale is applied after the product instead of before) and a row's
 * result has the same bits in any batch and on either kernel family -- but the IMAGE of the residual stream is coarser than
 * the one mevi_rmsnorm_split_f16 writes: its exponent comes from the carried bound
 *     out_bound = x_bound + ||a|| * w_norm_max (x 1.0001),
 * which is never re-tightened and, after the 24-36 residual adds of a stack, sits tens to hundreds of times above the rows'
 * real maximum, so the (hi, lo) pair keeps several bits fewer than its 22 in the deep layers.  The goldens hold at 5e-5 either
 * way; an A/B against the default path compares unequal operand precision.  OFF by default (MEVI_FOLD_NORM=1 turns it on); it
 * also measured slower at scale.  n of the stream: multiple of 16, <= 1024 (mevi_gemm_norm_fold_supported).
 * ---------------------------------------------------------------------- */
int mevi_gemm_norm_fold_supported(int64_t n_stream);
/* x f32 [m, k] -> image + exponents, bound[r] >= max |x_r| (the l2 norm, rounded up), ssq [m, k / 16] */
int mevi_split_rows_ssq_f16(const float *x, int64_t ldx, int64_t m, int64_t k, void *img, int8_t *exps, float *bound,
                            float *ssq, void *stream);
/* out[r] = 1 / sqrt(sum(parts[r, :]) / dim + eps) */
int mevi_row_rscale_f32(const float *parts, int64_t m, int64_t nparts, float dim, float eps, float *out, void *stream);
/* C = act(rs (.) (A.W^T) + bias) (+ residual), rs[r] from parts [m, nparts]; rscale_ws f32 [m] (scratch of the call) */
int mevi_gemm_nt_split_normed_f32(const void *a_img, const int8_t *a_exp, const float *parts, int64_t nparts, float rs_dim,
                                  float rs_eps, float *rscale_ws, const void *w_img, const int8_t *w_exp, float *c,
                                  int64_t ldc, int64_t m, int64_t n, int64_t k, const float *bias, const float *residual,
                                  int64_t ldr, int act, void *stream);
/* ... written as a split image (mevi_gemm_nt_split_to_split); a_norm_bound >= ||rs_r A_r|| for every row (sqrt(d)) */
int mevi_gemm_nt_split_normed_to_split(const void *a_img, const int8_t *a_exp, const float *parts, int64_t nparts,
                                       float rs_dim, float rs_eps, float *rscale_ws, float a_norm_bound, const void *w_img,
                                       const int8_t *w_exp, float w_norm_max, int64_t m, int64_t n, int64_t k,
                                       const float *bias, float bias_abs_max, int act, void *out_img, int8_t *out_exp,
                                       float *out_norm, void *stream);
/* c = residual + A.W^T (f32 [m, n]) AND its image / exponents / bound / block sums.  a_norm [m] (or NULL: a_norm_const for
 * every row) bounds ||A_r||; the image's exponent comes from x_bound[r] + a_norm * w_norm_max.  parts != NULL: A is itself a
 * normed stream (c = residual + rs (.) (A.W^T): the one-position decoder's o(v(norm x)) projection), a_norm_const = sqrt(d). */
int mevi_gemm_nt_split_residual_stream(const void *a_img, const int8_t *a_exp, const float *a_norm, float a_norm_const,
                                       const float *parts, int64_t nparts, float rs_dim, float rs_eps, float *rscale_ws,
                                       const void *w_img, const int8_t *w_exp, float w_norm_max, float *c, int64_t ldc,
                                       int64_t m, int64_t n, int64_t k, const float *residual, int64_t ldr,
                                       const float *x_bound, void *out_img, int8_t *out_exp, float *out_bound,
                                       float *out_ssq, void *stream);

/* ------------------------------------------------------------------------
 * Small-shape T5 / adaptor operators (wave-per-row kernels).  Stream-ordered, no workspace.
 * ---------------------------------------------------------------------- */
/* T5LayerNorm: out = w * (x / sqrt(mean(x^2) + eps))  (MEVI/transformers/modeling_t5.py:155-171) */
int mevi_rmsnorm_f32(const float *x, int64_t ldx, const float *w, float eps, int64_t rows, int64_t dim,
                     float *out, int64_t ldo, void *stream);
/* torch LayerNorm(x + y + cvec) * w + b, y / cvec optional: the post-LN residual blocks of
 * nn.TransformerDecoderLayer (adaptor, modeling_t5.py:1252-1255); cvec carries the adaptor's
 * constant cross-attention output (its memory is one learned vector, modeling_t5.py:1663). */
int mevi_add_layernorm_f32(const float *x, int64_t ldx, const float *y, int64_t ldy, const float *cvec,
                           const float *w, const float *b, float eps, int64_t rows, int64_t dim,
                           float *out, int64_t ldo, void *stream);
/* out[i] = table[idx[i]]: nn.Embedding (modeling_t5.py:718) and beam reorder (generation_utils.py:927-934) */
int mevi_gather_rows_f32(const float *table, int64_t ldt, const int64_t *idx, int64_t n, int64_t dim,
                         float *out, int64_t ldo, void *stream);
/* out[idx[r], :] = src[r, :] for r < n (rows must be distinct): the inverse of mevi_gather_rows_f32.  Used to run the
 * linear layers of the T5 / BERT encoders on the REAL tokens of a batch only (the reference pads every query to 32 and
 * every passage to 128 tokens, modeling_t5.py:969-1069 runs them all) and to put the rows back into the padded
 * layout the attention kernels read. */
int mevi_scatter_rows_f32(const float *src, int64_t lds, const int64_t *idx, int64_t n, int64_t dim, float *out,
                          int64_t ldo, void *stream);
int mevi_scale_f32(const float *x, float alpha, int64_t n, float *out, void *stream);
/* softmax(scale*q.k^T + bias[h, q_pos0+t, j] + key mask + causal mask) . v for <= 256 keys
 * (T5Attention.forward, modeling_t5.py:374-410: no 1/sqrt(d) scaling, fp32 softmax; also
 * nn.MultiheadAttention of the adaptor with scale = head_dim^-0.5).
 *   q[b, t, h*dh + d] via (q_bs, q_ts); k/v[b / kv_div, j, h*dh + d]; out like q.
 *   key_mask i64 [nb / kv_div, tk] (1 = attend) or NULL; causal: key j allowed iff j <= q_pos0 + t.
 *   kv_off (i64 [nb / kv_div + 1], device) or NULL: K|V are PACKED -- kv batch c owns rows kv_off[c] .. kv_off[c+1]-1
 *   (k_ts / v_ts apart; k_bs / v_bs and key_mask unused), all keys real, tk = the longest sequence: the cross-attention
 *   of the decoders over the real encoder positions only (same bits as the padded, masked form). */
int mevi_attention_f32(const float *q, int64_t q_bs, int64_t q_ts, const float *k, int64_t k_bs, int64_t k_ts,
                       const float *v, int64_t v_bs, int64_t v_ts, float *out, int64_t o_bs, int64_t o_ts,
                       int64_t nb, int64_t tq, int64_t tk, int64_t heads, int64_t dh, int64_t kv_div,
                       const float *bias, int64_t bias_rows, int64_t bias_ld, int64_t q_pos0,
                       const int64_t *key_mask, int causal, float scale, const int64_t *kv_off, void *stream);
/* One decode step of self-attention over beam-shared caches: row b attends to tk <= 8 cached positions, position j of
 * row b living in cache row key_rows[b * tk + j] (k / v[row, j, h*dh + d] via (k_bs, k_ts) / (v_bs, v_ts)).  The reference
 * re-orders every layer's K|V cache after each beam step (generation_utils.py:927-934 `_reorder_cache`: index_select of the
 * whole cache by the surviving beams' parents); here the caches stay where each step wrote them and the rows carry their
 * ancestors' indices -- same arithmetic as mevi_attention_f32 (tq = 1) on the re-ordered copy, bit for bit. */
int mevi_attention_cached_f32(const float *q, int64_t q_bs, const float *k, int64_t k_bs, int64_t k_ts, const float *v,
                              int64_t v_bs, int64_t v_ts, float *out, int64_t o_bs, int64_t nb, int64_t tk, int64_t heads,
                              int64_t dh, const int32_t *key_rows, const float *bias, int64_t bias_rows, int64_t bias_ld,
                              int64_t q_pos0, int causal, float scale, void *stream);
/* The same attention over PACKED sequences (padding-free encoders): sequence b owns rows seq_off[b] .. seq_off[b+1]-1
 * (i64 [nseq + 1], device) of q / k / v / out, whose rows are `*_ts` floats apart; every key is real, bias row / column =
 * position inside the sequence, max_len = longest sequence (<= 256).  Bit-identical to mevi_attention_f32 on the padded
 * layout with a key mask (masked keys contribute exact zeros there); no reference counterpart -- the reference
 * attends over the padded [B, S] layout (modeling_t5.py:374-410). */
int mevi_attention_varlen_f32(const float *q, int64_t q_ts, const float *k, int64_t k_ts, const float *v, int64_t v_ts,
                              float *out, int64_t o_ts, const int64_t *seq_off, int64_t nseq, int64_t max_len,
                              int64_t heads, int64_t dh, const float *bias, int64_t bias_rows, int64_t bias_ld,
                              int causal, float scale, void *stream);
/* The three attention forms with the context written as the SPLIT IMAGE the o-projection reads (mevi_gemm_nt_split_f32's A
 * operand: out_img [rows, 2 * img_np] f16, hi at column c, lo at img_np + c, both scaled by 2^out_exp) instead of f32 --
 * the reference's `o(context)` (modeling_t5.py:407-410) is the only consumer of the context.  One exponent for all rows,
 * because the heads of a row are written by different waves: the caller derives it from a bound on |V| (the context is a
 * convex combination of V rows; mevi_amd/ops.py: split_bound) so that |context| 2^out_exp < 2^15, and fills the rows'
 * exponent array with it.  img_bs / img_ts: batch / token strides in f16 elements (img_ts = 2 * img_np when rows are
 * contiguous).  Same arithmetic as the f32 forms up to the final rounding into (hi, lo). */
int mevi_attention_split_f16(const float *q, int64_t q_bs, int64_t q_ts, const float *k, int64_t k_bs, int64_t k_ts,
                             const float *v, int64_t v_bs, int64_t v_ts, void *out_img, int64_t img_np, int out_exp,
                             int64_t img_bs, int64_t img_ts, int64_t nb, int64_t tq, int64_t tk, int64_t heads, int64_t dh,
                             int64_t kv_div, const float *bias, int64_t bias_rows, int64_t bias_ld, int64_t q_pos0,
                             const int64_t *key_mask, int causal, float scale, const int64_t *kv_off, void *stream);
int mevi_attention_cached_split_f16(const float *q, int64_t q_bs, const float *k, int64_t k_bs, int64_t k_ts, const float *v,
                                    int64_t v_bs, int64_t v_ts, void *out_img, int64_t img_np, int out_exp, int64_t img_bs,
                                    int64_t nb, int64_t tk, int64_t heads, int64_t dh, const int32_t *key_rows,
                                    const float *bias, int64_t bias_rows, int64_t bias_ld, int64_t q_pos0, int causal,
                                    float scale, void *stream);
int mevi_attention_varlen_split_f16(const float *q, int64_t q_ts, const float *k, int64_t k_ts, const float *v, int64_t v_ts,
                                    void *out_img, int64_t img_np, int out_exp, int64_t img_ts, const int64_t *seq_off,
                                    int64_t nseq, int64_t max_len, int64_t heads, int64_t dh, const float *bias,
                                    int64_t bias_rows, int64_t bias_ld, int causal, float scale, void *stream);
/* PAWA adaptive head on the valid columns only (modeling_t5.py:1607, 1677-1689):
 * out[row, c] = sum_d s[row, d] * (t[trow, c*dim + d] + e[c, d]),  t = adaptor_linear slice, e = lm_head rows;
 * trow = row when t_index is NULL, else t_index[row] (i64, device): the adaptor sees only the code prefix of a beam,
 * so t can be a table with one row per prefix shared by every beam that carries it (mevi_amd/nci.py PrefixTables). */
int mevi_adaptive_logits_f32(const float *s, int64_t lds, const float *t, int64_t ldt, const int64_t *t_index,
                             const float *e, int64_t rows, int64_t ncol, int64_t dim, float *out, void *stream);

/* The same head with lm_head's rows already inside the matrices and the d_model^-0.5 of modeling_t5.py:1607 applied here:
 * out[row, c] = sum_d (s[row, d] * alpha) * te[trow, c*dim + d],  te[., c*dim + d] = t[., c*dim + d] + e[c, d]  (the bias of the
 * GEMM that wrote it).  The products of mevi_scale_f32 + mevi_adaptive_logits_f32 (bit-identical, except that at dim 768 the sums
 * follow mevi_gemm_nt_split_head_f32's order); the row's hidden state is read once per 64 columns instead of once per column.
 * dim <= 1024. */
int mevi_adaptive_logits_rows_f32(const float *s, int64_t lds, float alpha, const float *te, int64_t ldt, const int64_t *t_index,
                                  int64_t rows, int64_t ncol, int64_t dim, float *out, void *stream);

/* The PAWA head of rows that have no table (the final position of the beam search) in one pass: the head GEMM
 * te[m, c*dim + d] = a[m, :] . w[c*dim + d, :] + bias[c*dim + d] (split images as mevi_gemm_nt_split_f32) is multiplied with the rows'
 * hidden states inside its epilogue -- part f32 [n / 256][m][4] holds one partial per (tile, row, wave) -- and mevi_logits_finish_f32
 * adds the twelve partials of a (row, column): out[m, c] = sum_d (s[m, d] * alpha) * te[m, c*dim + d], in the summation order
 * mevi_adaptive_logits_rows_f32 uses at dim = 768, so table rows and per-beam rows give the same bits; the [m, n] head matrices
 * (55 GB for 70 k beams at K = 256) are never written.  mevi_gemm_nt_split_head_supported: dim == 768, n % 768 == 0, a shape of
 * the tile-stream kernel (not the latency kernels'); elsewhere use mevi_gemm_nt_split_f32 + mevi_adaptive_logits_rows_f32. */
int mevi_gemm_nt_split_head_supported(int64_t m, int64_t n, int64_t k, int64_t dim);
int mevi_gemm_nt_split_head_f32(const void *a_img, const int8_t *a_exp, const void *w_img, const int8_t *w_exp, int64_t m, int64_t n,
                                int64_t k, const float *bias, const float *s, int64_t lds, float alpha, int64_t dim, float *part,
                                void *stream);
int mevi_logits_finish_f32(const float *part, int64_t rows, int64_t ncol, float *out, void *stream);

/* T5LayerNorm + the ONE projection it feeds (q|k|v, the cross-attention q, wi: MEVI/transformers/modeling_t5.py:155-171 followed by
 * :181, :350-352) in one launch, for the few rows of the latency path (the reference's --timing_infer_step regime): every workgroup
 * normalises its rows itself, with mevi_rmsnorm_split_f16's arithmetic, and multiplies with mevi_gemm_nt_split_*'s accumulation
 * chain -- results equal the two-call form bit for bit.  mevi_gemm_rmsnorm_supported(m, n, k) says whether a shape is on this path
 * (k == 768, few output tiles); elsewhere use the two calls. */
int mevi_gemm_rmsnorm_supported(int64_t m, int64_t n, int64_t k);
int mevi_gemm_nt_rmsnorm_split_f32(const float *x, int64_t ldx, const float *ln_w, float eps, const void *w_img,
                                   const int8_t *w_exp, float *c, int64_t ldc, int64_t m, int64_t n, int64_t k,
                                   const float *bias, const float *residual, int64_t ldr, int act, void *stream);
int mevi_gemm_nt_rmsnorm_split_to_split(const float *x, int64_t ldx, const float *ln_w, float eps, const void *w_img,
                                        const int8_t *w_exp, float w_norm_max, int64_t m, int64_t n, int64_t k,
                                        const float *bias, float bias_abs_max, int act, void *out_img, int8_t *out_exp,
                                        float *out_norm, void *stream);

/* ------------------------------------------------------------------------
 * Constrained beam step over the shared-layer RQ tree (one decoding step).
 * Replaces select_valid_embedding + log_softmax + prefix-tree mask + top-k + the Python
 * candidate loop (modeling_t5.py:1578-1603,1689; generation_utils.py:783, 803-818, 851-945).
 *   logits f32 [nq*nb, K+1]: col 0 = eos, cols 1..K = the K codes of this level
 *   beam_scores f32 [nq, nb]
 *   final_step == 0: out_scores/out_parent/out_code [nq, R] = the R best of
 *       beam_score[r] + log_softmax(logits[r])[1 + c], descending, ties by lower r*K + c
 *   final_step == 1: out_scores [nq, nb] = beam_score[r] + log_softmax(logits[r])[0] (eos closes)
 *   final_step == 2: pq.beam_search step -- logits f32 [nq*nb, K] (no eos column), candidates
 *       beam_score[r] * softmax(logits[r])[c], outputs as for 0
 * ---------------------------------------------------------------------- */
int mevi_beam_step_f32(const float *logits, const float *beam_scores, int64_t nq, int64_t nb, int64_t K,
                       int64_t R, int final_step, float *out_scores, int32_t *out_parent,
                       int32_t *out_code, void *stream);

/* The same step under a GENERIC prefix tree -- TreeBuilder(share_sons=False).add(path) per existing code path
 * (MEVI/main_models.py:50-63, 1707-1728) and the per-beam trie walk of MEVI/transformers/generation_utils.py:803-818.
 * The trie arrives level by level: for the level being decoded, node i32 [nq, nb] = every beam's node, tree_mask
 * u32 [n_nodes, ceil(K/32)] = which codes its children carry, tree_base i32 [n_nodes] = index (in the NEXT level) of its
 * first child, children contiguous in code order.  Candidates are the children only; the log-softmax normaliser spans eos
 * and all K codes as in the reference (its tree mask is added after log_softmax).  All R beams run from the first step
 * (nb >= R; the reference seeds beams 1..R-1 with -1e9), so R candidates always exist.  out_node i32 [nq, R] = the kept
 * candidates' nodes in the next level; the other outputs as mevi_beam_step_f32 (final_step 0).  The eos step after the last
 * level is mevi_beam_step_f32 with final_step = 1 (a leaf's only child is eos). */
int mevi_beam_step_tree_f32(const float *logits, const float *beam_scores, int64_t nq, int64_t nb, int64_t K,
                            int64_t R, const int32_t *node, const uint32_t *tree_mask, const int32_t *tree_base,
                            int64_t n_nodes, float *out_scores, int32_t *out_parent, int32_t *out_code,
                            int32_t *out_node, void *stream);

/* Row-wise (log-)softmax with the beam step's arithmetic (max, sum of expf(x - max), logf), for the branches that keep
 * EVERY candidate instead of a top-R: mode 0 = log_softmax of x f32 [rows, cols] (the all-paths walk `_generate_all`,
 * MEVI/transformers/generation_utils.py:1013-1136); mode 1 = scale[row] * softmax(x) (pq.beam_search while beams * K <= R,
 * MEVI/pq.py:660-676; scale may be null = 1).  out f32 [rows, cols] (may alias x). */
int mevi_row_softmax_f32(const float *x, int64_t rows, int64_t cols, int mode, const float *scale, float *out,
                         void *stream);

/* ------------------------------------------------------------------------
 * Fine stage: in-cluster re-ranking (MEVI/main_models.py:3915-4014).
 *   pair_dot: out[i] = <a[ia[i]], b[ib[i]]>, sequential f32 fmaf chain (same chain as
 *             mevi_ip_topk_f32, so dense and fine scores of one pair are identical)
 *   segment_sort_desc: every segment [seg_offsets[s], seg_offsets[s+1]) sorted by
 *             (score desc, id asc); segments of at most 16384 entries.
 * ---------------------------------------------------------------------- */
int mevi_pair_dot_f32(const float *a, int64_t lda, const int64_t *ia, const float *b, int64_t ldb,
                      const int64_t *ib, int64_t n, int64_t dim, float *out, void *stream);
int mevi_segment_sort_desc_f32(const float *scores, const int64_t *ids, const int64_t *seg_offsets,
                               int64_t nseg, int64_t max_seg_len, float *out_scores, int64_t *out_ids,
                               void *stream);
/* --doc_multiclus > 1 (MEVI/main_models.py:3997-4011: np.unique + the `uscores[ui] += s` / torch.max loop + torch.sort):
 * per segment the entries of one id are merged -- mode 0 'add': 0 + s + s ... in list order (sequential f32 adds),
 * mode 1 'max' -- and the unique entries sorted by (score desc, id asc) into out[seg_offsets[s] ...], out_counts[s] of
 * them.  ids in [0, 2^32), segments of at most 16384 entries. */
int mevi_segment_aggregate_sort_f32(const float *scores, const int64_t *ids, const int64_t *seg_offsets, int64_t nseg,
                                    int64_t max_seg_len, int mode, float *out_scores, int64_t *out_ids,
                                    int32_t *out_counts, void *stream);

/* Pieces of pq.beam_search (MEVI/pq.py:613-713; only reached with doc_multiclus > 1): the score row
 * -sum_k (x_k - c_k)^2 of every row against the K centroids of one level, and the residual hand-down
 * out[r] = x[src[r]] - centroids[code[r]].  The top-R step is mevi_beam_step_f32 with final_step = 2
 * (logits = the K score columns, candidate = beam_prob * softmax). */
int mevi_rq_neg_dist_f32(const float *x, int64_t n, int64_t dim, const float *centroids, int64_t K,
                         float *neg_dist, void *stream);
int mevi_gather_sub_f32(const float *x, const int64_t *src, const float *centroids, const int32_t *code,
                        int64_t n, int64_t dim, float *out, void *stream);

/* ------------------------------------------------------------------------
 * Centroid update of k-means / RQ codebook training (the offline index build the reference does with
 * scikit-learn, MEVI/pq.py:550-598; assignment = mevi_rq_encode_f32 with a one-level codebook):
 *   centroids[k] = mean of the rows x[i] with codes[i * code_stride] == k  (an empty cluster keeps
 *   old_centroids[k], or 0 when old_centroids is null), counts[k] = their number, *sum_sq = sum ||x_i||^2.
 * Deterministic (fixed-order two-stage reduction, f64 second stage).  Stream-ordered.
 *   x f32 [n, dim], codes i32, centroids f32 [K, dim], counts i32 [K], sum_sq f64 [1] (may be null). */
size_t mevi_cluster_means_workspace_bytes(int64_t n, int64_t dim, int64_t K);
int mevi_cluster_means_f32(const float *x, int64_t n, int64_t dim, const int32_t *codes, int64_t code_stride,
                           int64_t K, const float *old_centroids, float *centroids, int32_t *counts, double *sum_sq,
                           void *workspace, size_t workspace_bytes, void *stream);

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
  double max_err_ratio;      /* indexed search: largest observed |f16 approx - exact| / proven bound among survivors */
  double err_bound;          /* indexed search: 1.0 (max_err_ratio is relative to the bound the proof uses) */
  int64_t n_second_pass_queries; /* indexed search: unproven queries retried with a 2x wider survivor list */
  /* indexed search, mevi_ip_topk_set_profiling(2) only (one small counting launch per filter launch; 0 otherwise): what the
   * DATA decides about the pre-filter's cost -- keys appended by the filter launches of the main pass (sum over queries and
   * launches), the largest per-query count of one launch, and (query, launch) pairs that filled their list (those queries
   * take the guaranteed path) */
  int64_t n_filter_candidates;
  int64_t max_launch_candidates;
  int64_t n_list_overflows;
  /* mevi_ip_topk_indexed8_f32: queries that went through the 8-bit pass, and how many of them it left unproven (> 0: the batch
   * was searched again through the f16 image; the other fields then describe both searches) */
  int64_t n_i8_queries;
  int64_t n_i8_unproven;
} mevi_ip_topk_stats;
void mevi_ip_topk_set_growth(double growth);
void mevi_ip_topk_set_profiling(int enable); /* 1: record HIP events around every filter/compact launch; 2: also count candidates */
void mevi_ip_topk_get_stats(mevi_ip_topk_stats *out);

/* ---- host-side text output (no device work) --------------------------------------------------------------------
 * The ranked TSVs carry Python `str(int)` / `str(float)` renderings (faiss_search.to_file, MEVI/faiss_search.py:71-77:
 * `','.join(str(x) for x in row.tolist())`, f32 widened to double) -- 14 M numbers per dense file.  These write the same
 * bytes: comma-joined values into `out` (cap >= 26 n / 21 n bytes), returning the byte count or a negative status. */
int64_t mevi_format_f32_list(const float *v, int64_t n, char *out, int64_t cap);
int64_t mevi_format_i64_list(const int64_t *v, int64_t n, char *out, int64_t cap);
/* both list columns of a whole ranked TSV in one call (faiss_search.to_file writes 6980 x 1000 ids and scores per dense file,
 * MEVI/faiss_search.py:71-77): row r -> `id,id,...<TAB>score,score,...` at out + r * row_cap (row_cap >= 47 k + 1), its byte
 * count in lens[r]; `threads` host threads share the rows.  Status. */
int mevi_format_ranked_rows(const int64_t *ids, const float *scores, int64_t rows, int64_t k, char *out, int64_t row_cap,
                            int64_t *lens, int32_t threads);
/* and back: one comma-separated field of such a file -> numbers (the consumers `eval()` every field,
 * evaluate.py:91-110, ensemble_marco.py:85-110).  Count, or a negative status when a token is not a plain number of
 * that kind (the caller falls back to Python's parser) or cap is short. */
int64_t mevi_parse_i64_list(const char *s, int64_t len, int64_t *out, int64_t cap);
int64_t mevi_parse_f64_list(const char *s, int64_t len, double *out, int64_t cap);

/* a whole ranked TSV in one call: every line's query field as (start, length) spans of `buf`, the comma lists of column
 * col_i (integers) and col_f (floats; either -1 = absent) flat with per-line offsets seg_*[lines + 1] -- what
 * ensemble_marco.parse / evaluate.parse build line by line with eval() (MEVI/ensemble_marco.py:92-111,
 * MEVI/evaluate.py:91-110).  Returns the line count, or a negative status as soon as the file is not of the plain shape
 * (the caller then parses it the reference's way). */
int64_t mevi_parse_tsv_columns(const char *buf, int64_t len, int32_t col_q, int32_t col_i, int32_t col_f, int64_t *q_span,
                               int64_t *seg_i, int64_t *vals_i, int64_t cap_i, int64_t *seg_f, double *vals_f,
                               int64_t cap_f, int64_t cap_lines);

/* ------------------------------------------------------------------------
 * The consumers on the device (ensemble_marco.combine_main, MEVI/ensemble_marco.py:150-238; evaluate(), :28-73 and
 * MEVI/evaluate.py:27-72).  Lists are flat arrays with per-query offsets seg[nq + 1].
 *   cluster_ranks: out[e] = LAST index r with beam[q][r][:] == codes[docs[e]][:] (the dict `cr[tuple(clus)] = i` keeps the
 *             last of a repeated cluster, :183-189), n_clusters when none matches or docs[e] == -1.  *first_bad = smallest
 *             entry whose id has no code row (the reference raises KeyError there), ~0 when none.
 *   ensemble_rank: per query the list dense ++ fine cut to min(n_d + n_f, 2 n_d) entries (zip with chain(cranks, cranks),
 *             :226-232; the fine list re-uses the DENSE ranks position by position), value
 *             v = score + term[crank], times `punish` when crank == n_clusters -- term[c] = alpha / (beta * c + 1) and
 *             punish = 1 - gamma * alpha evaluated by the caller in host doubles, so the device does one f64 add and one
 *             f64 multiply per entry, exactly the reference's roundings (:233-235); a document listed twice keeps its first
 *             position and its last value (dict), ranking by descending value, ties in first-seen order (sorted() is
 *             stable, :38-39).  out_docs[out_seg[q] ..] receives out_n[q] ids.  At most 8192 entries per query; *err != 0
 *             when an id is outside +-2^46 or a list changed size (the caller then takes the host path).
 *   first_hits: out[p] = first index of pair_doc[p] in list pair_row[p] (its first list_n[row] entries, or the whole
 *             segment when list_n is NULL), -1 when absent or pair_row[p] < 0 -- the rank evaluate() looks up per gt.
 * ---------------------------------------------------------------------- */
int mevi_cluster_ranks_i32(const int32_t *codes, int64_t n_docs, int64_t M, const int64_t *docs, const int64_t *seg,
                           int64_t nq, const int32_t *beam, int64_t R, int32_t n_clusters, int32_t *out_ranks,
                           uint64_t *first_bad, void *stream);
int mevi_ensemble_rank_f64(const int64_t *seg_dense, const int64_t *docs_dense, const double *scores_dense,
                           const int32_t *cranks_dense, const int64_t *fine_row, const int64_t *seg_fine,
                           const int64_t *docs_fine, const double *scores_fine, int64_t nq, int64_t max_entries,
                           int32_t n_clusters, const double *term, double punish, const int64_t *out_seg,
                           int64_t *out_docs, int32_t *out_n, int32_t *err, void *stream);
int mevi_first_hits_i64(const int64_t *lists, const int64_t *seg, const int32_t *list_n, const int64_t *pair_row,
                        const int64_t *pair_doc, int64_t n_pairs, int32_t *out_rank, void *stream);

#ifdef __cplusplus
}
#endif
#endif /* MEVI_HIP_H */
