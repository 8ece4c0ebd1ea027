// Constrained beam step over the shared-layer RQ tree (K10 / K11).
//
// Replaces, per decoding step, the reference's position mask + log_softmax + prefix-tree
// mask + top-2R + Python candidate loop (MEVI/transformers/modeling_t5.py:1578-1603,1689;
// generation_utils.py:783, 803-818, 851-945).  For the shared-sons tree every level-p node
// has the same K children, so the step collapses to (SURVEY 8(a')):
//     lsm      = log_softmax over {eos} U {K level-p codes}     (all other columns are exp(-1e9) = 0)
//     cand[r,c] = beam_score[r] + lsm[r, c]                      (eos is cut by the tree)
//     keep the R best of the nb*K candidates (descending, ties -> lower r*K + c)
// Input logits hold only the valid columns: col 0 = eos, cols 1..K = codes of the level.
// One workgroup per query: wave-per-beam logsumexp, 64-bit (score|index) keys, LDS bitonic sort.
#include "common.h"

#include <math.h>

namespace mevi {
namespace {

// mode 0: NCI step (K+1 columns, col 0 = eos, log-domain);  mode 1: NCI final step;
// mode 2: pq.beam_search step (K columns of -distance): cand = beam_prob[r] * softmax(row)[c]  (pq.py:660-676)
// Generic prefix trees (TreeBuilder(share_sons=False), MEVI/main_models.py:50-63; the mask walk of
// generation_utils.py:803-818): `node` i32 [nq, nb] is every beam's trie node at this level, `tmask` u32 [nodes, W] the
// set of codes its children carry (bit c of word c / 32), `tbase` i32 [nodes] the next level's index of its first child
// (children in code order, contiguous).  A candidate (r, c) exists iff bit c of tmask[node[r]] is set; the log-softmax
// normaliser still spans eos and all K codes of the level (the tree mask is ADDED to the log-probabilities).  out_node =
// the child's index in the next level.  node == nullptr: the shared-sons tree (every code allowed).
__global__ __launch_bounds__(256) void beam_step_kernel(const float *__restrict__ logits,
                                                       const float *__restrict__ beam_scores, int nb, int K,
                                                       int R, int final_step, float *__restrict__ out_scores,
                                                       int *__restrict__ out_parent, int *__restrict__ out_code,
                                                       const int *__restrict__ node, const unsigned int *__restrict__ tmask,
                                                       const int *__restrict__ tbase, int W, int *__restrict__ out_node) {
  extern __shared__ __attribute__((aligned(16))) unsigned long long skeys[];  // P keys, then nb floats x2
  const int q = blockIdx.x;
  const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
  const int ncol = (final_step == 2) ? K : K + 1;
  const int ncand = nb * K;
  int P = 64;
  while (P < ncand) P <<= 1;
  float *smax = reinterpret_cast<float *>(skeys + P);
  float *slog = smax + nb;
  const float *lq = logits + (size_t)q * nb * ncol;

  for (int r = wave; r < nb; r += 4) {  // log-sum-exp of beam r over its K+1 valid columns
    const float *row = lq + (size_t)r * ncol;
    float m = -INFINITY;
    for (int c = lane; c < ncol; c += 64) m = fmaxf(m, row[c]);
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) m = fmaxf(m, __shfl_xor(m, off));
    float s = 0.f;
    for (int c = lane; c < ncol; c += 64) s += expf(row[c] - m);
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) s += __shfl_xor(s, off);
    if (lane == 0) {
      smax[r] = m;
      slog[r] = logf(s);
    }
  }
  __syncthreads();
  if (final_step == 1) {  // hypothesis closes with eos: score + log_softmax[eos]
    for (int r = t; r < nb; r += 256)
      out_scores[(size_t)q * nb + r] = beam_scores[(size_t)q * nb + r] + ((lq[(size_t)r * ncol] - smax[r]) - slog[r]);
    return;
  }
  for (int i = t; i < P; i += 256) {
    unsigned long long key = 0ull;
    if (i < ncand) {
      const int r = i / K, c = i - r * K;
      if (final_step == 2) {
        const float p = expf(lq[(size_t)r * ncol + c] - smax[r]) / expf(slog[r]);
        key = make_key(beam_scores[(size_t)q * nb + r] * p, (unsigned int)i);
      } else if (!node || ((tmask[(size_t)node[(size_t)q * nb + r] * W + (c >> 5)] >> (c & 31)) & 1u)) {
        const float lsm = (lq[(size_t)r * ncol + 1 + c] - smax[r]) - slog[r];
        key = make_key(beam_scores[(size_t)q * nb + r] + lsm, (unsigned int)i);
      }
    }
    skeys[i] = key;
  }
  __syncthreads();
  bitonic_sort_desc<256>(skeys, P, t);
  for (int i = t; i < R; i += 256) {
    const unsigned long long key = skeys[i];
    const int flat = (int)key_id(key);
    const int r = flat / K, c = flat % K;
    out_scores[(size_t)q * R + i] = key == 0ull ? -INFINITY : key_score(key);   // key 0: fewer candidates than R (host refuses)
    out_parent[(size_t)q * R + i] = key == 0ull ? 0 : r;
    out_code[(size_t)q * R + i] = key == 0ull ? -1 : c;
    if (node) {
      int nn = -1;
      if (key != 0ull) {
        const int n0 = node[(size_t)q * nb + r];
        const unsigned int *m = tmask + (size_t)n0 * W;
        int below = 0;
        for (int w = 0; w < (c >> 5); ++w) below += __popc(m[w]);
        below += __popc(m[c >> 5] & ((1u << (c & 31)) - 1u));
        nn = tbase[n0] + below;
      }
      out_node[(size_t)q * R + i] = nn;
    }
  }
}

// Row-wise (log-)softmax with the arithmetic of the beam step above (max, sum of expf(x - max), logf): one wave per row.
// mode 0: out = (x - max) - log(sum)                       (the all-paths walk of _generate_all, generation_utils.py:1013-1136)
// mode 1: out = scale[row] * expf(x - max) / expf(log sum)  (pq.beam_search keeping every candidate, pq.py:660-676)
__global__ __launch_bounds__(256) void row_softmax_kernel(const float *__restrict__ x, long long rows, int cols, int mode,
                                                         const float *__restrict__ scale, float *__restrict__ out) {
  const long long r = (long long)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (r >= rows) return;
  const int lane = threadIdx.x & 63;
  const float *row = x + (size_t)r * cols;
  float m = -INFINITY;
  for (int c = lane; c < cols; c += 64) m = fmaxf(m, row[c]);
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) m = fmaxf(m, __shfl_xor(m, off));
  float s = 0.f;
  for (int c = lane; c < cols; c += 64) s += expf(row[c] - m);
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) s += __shfl_xor(s, off);
  const float ls = logf(s);
  float *o = out + (size_t)r * cols;
  if (mode == 0) {
    for (int c = lane; c < cols; c += 64) o[c] = (row[c] - m) - ls;
  } else {
    const float w = scale ? scale[r] : 1.f, den = expf(ls);
    for (int c = lane; c < cols; c += 64) o[c] = w * (expf(row[c] - m) / den);
  }
}

}  // namespace
}  // namespace mevi

using namespace mevi;

extern "C" int mevi_row_softmax_f32(const float *x, int64_t rows, int64_t cols, int mode, const float *scale, float *out,
                                    void *stream) {
  MEVI_REQUIRE(rows >= 0 && cols > 0 && cols < (1LL << 30) && (mode == 0 || mode == 1), MEVI_ERR_INVALID_ARG, "row_softmax: bad arguments");
  if (rows == 0) return MEVI_OK;
  MEVI_REQUIRE(x && out, MEVI_ERR_INVALID_ARG, "row_softmax: null pointer");
  hipLaunchKernelGGL(row_softmax_kernel, dim3((unsigned)((rows + 3) / 4)), dim3(256), 0, (hipStream_t)stream, x, (long long)rows,
                     (int)cols, mode, scale, out);
  MEVI_HIP_CHECK(hipGetLastError());
  return MEVI_OK;
}

extern "C" int mevi_beam_step_f32(const float *logits, const float *beam_scores, int64_t nq, int64_t nb, int64_t K,
                                  int64_t R, int final_step, float *out_scores, int32_t *out_parent,
                                  int32_t *out_code, void *stream) {
  MEVI_REQUIRE(nq >= 0 && nb > 0 && K > 0 && R > 0, MEVI_ERR_INVALID_ARG, "beam_step: bad shape");
  if (nq == 0) return MEVI_OK;
  MEVI_REQUIRE(final_step >= 0 && final_step <= 2, MEVI_ERR_INVALID_ARG, "beam_step: mode must be 0, 1 or 2");
  MEVI_REQUIRE(logits && beam_scores && out_scores && (final_step == 1 || (out_parent && out_code)),
               MEVI_ERR_INVALID_ARG, "beam_step: null pointer");
  MEVI_REQUIRE(nb * K <= 16384, MEVI_ERR_UNSUPPORTED, "beam_step: nb*K=%lld > 16384", (long long)(nb * K));
  MEVI_REQUIRE(final_step == 1 || nb * K >= R, MEVI_ERR_UNSUPPORTED,
               "beam_step: fewer candidates (%lld) than beams (%lld)", (long long)(nb * K), (long long)R);
  int P = 64;
  while (P < nb * K) P <<= 1;
  const size_t lds = (size_t)P * 8 + (size_t)nb * 8;
  if (lds > 65536)
    MEVI_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void *>(beam_step_kernel),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
  hipLaunchKernelGGL(beam_step_kernel, dim3((unsigned)nq), dim3(256), lds, (hipStream_t)stream, logits, beam_scores,
                     (int)nb, (int)K, (int)R, final_step, out_scores, out_parent, out_code, (const int *)nullptr,
                     (const unsigned int *)nullptr, (const int *)nullptr, 0, (int *)nullptr);
  MEVI_HIP_CHECK(hipGetLastError());
  return MEVI_OK;
}

extern "C" int mevi_beam_step_tree_f32(const float *logits, const float *beam_scores, int64_t nq, int64_t nb, int64_t K,
                                       int64_t R, const int32_t *node, const uint32_t *tree_mask, const int32_t *tree_base,
                                       int64_t n_nodes, float *out_scores, int32_t *out_parent, int32_t *out_code,
                                       int32_t *out_node, void *stream) {
  MEVI_REQUIRE(nq >= 0 && nb > 0 && K > 0 && R > 0 && n_nodes > 0, MEVI_ERR_INVALID_ARG, "beam_step_tree: bad shape");
  if (nq == 0) return MEVI_OK;
  MEVI_REQUIRE(logits && beam_scores && node && tree_mask && tree_base && out_scores && out_parent && out_code && out_node,
               MEVI_ERR_INVALID_ARG, "beam_step_tree: null pointer");
  MEVI_REQUIRE(nb * K <= 16384, MEVI_ERR_UNSUPPORTED, "beam_step_tree: nb*K=%lld > 16384", (long long)(nb * K));
  MEVI_REQUIRE(nb >= R, MEVI_ERR_UNSUPPORTED,
               "beam_step_tree: needs nb >= R beams (every beam sits on a trie node with at least one child, so nb >= R "
               "guarantees R candidates; the reference runs all R beams from the first step)");
  int P = 64;
  while (P < nb * K) P <<= 1;
  const size_t lds = (size_t)P * 8 + (size_t)nb * 8;
  if (lds > 65536)
    MEVI_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void *>(beam_step_kernel),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
  hipLaunchKernelGGL(beam_step_kernel, dim3((unsigned)nq), dim3(256), lds, (hipStream_t)stream, logits, beam_scores,
                     (int)nb, (int)K, (int)R, 0, out_scores, out_parent, out_code, node, tree_mask, tree_base,
                     (int)((K + 31) / 32), out_node);
  MEVI_HIP_CHECK(hipGetLastError());
  return MEVI_OK;
}
