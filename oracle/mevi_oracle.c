/*
 * mevi_oracle.c -- CPU restatement of the MEVI inference hot path.
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing under oracle/ is part of the product: only
 * tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load this
 * library, and only as the checker.  The product path (mevi_amd/) never falls
 * back to it.
 *
 * Every function cites the reference site it restates (paths relative to the
 * reference checkout, HugoZHL/MEVI @ v2).
 */
#include <float.h>
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#ifdef _OPENMP
#include <omp.h>
#endif

/* ----------------------------------------------------------------------------
 * Dense arm: exact inner-product top-k.
 * Restates MEVI/faiss_search.py:13-21 with param "Flat":
 *   index = faiss.index_factory(dim, "Flat", METRIC_INNER_PRODUCT); index.add(doc)
 *   dists, indices = index.search(query, topk)
 * faiss-cpu==1.7.4 (environment.yml:117) is a third-party dependency absent from
 * the reference tree; its published behaviour for IndexFlatIP is restated here:
 * scores = <q, d> in f32, the k largest per query in descending order, unfilled
 * slots id = -1 / score = -FLT_MAX (CMin<float>::neutral()).  faiss leaves the
 * summation order (BLAS sgemm) and the order of exact ties unspecified; this
 * oracle pins them: score = fmaf chain over k = 0..dim-1, ties by ascending id.
 * PARITY UNPINNED against faiss itself (not installable here; no reference test
 * holds vectors for it) -- see DESIGN.md.
 * -------------------------------------------------------------------------- */
typedef struct {
  float s;
  int64_t id;
} hit_t;

static int hit_before(const hit_t *a, const hit_t *b) {
  /* 1 if a ranks strictly before b: score desc, id asc */
  if (a->s > b->s) return 1;
  if (a->s < b->s) return 0;
  return a->id < b->id;
}

static int hit_cmp(const void *pa, const void *pb) {
  const hit_t *a = (const hit_t *)pa, *b = (const hit_t *)pb;
  if (hit_before(a, b)) return -1;
  if (hit_before(b, a)) return 1;
  return 0;
}

/* min-heap on "rank": root = worst retained hit */
static void heap_sift_down(hit_t *h, int64_t n, int64_t i) {
  for (;;) {
    int64_t l = 2 * i + 1, r = l + 1, w = i;
    if (l < n && hit_before(&h[w], &h[l])) w = l;
    if (r < n && hit_before(&h[w], &h[r])) w = r;
    if (w == i) return;
    hit_t t = h[i];
    h[i] = h[w];
    h[w] = t;
    i = w;
  }
}

float oracle_dot_f32(const float *a, const float *b, int64_t dim) {
  float acc = 0.0f;
  for (int64_t i = 0; i < dim; ++i) acc = fmaf(a[i], b[i], acc);
  return acc;
}

int oracle_ip_topk_f32(const float *q, int64_t nq, const float *docs, int64_t nd, int64_t dim,
                       int64_t k, int64_t id_offset, float *out_score, int64_t *out_id) {
  if (nq < 0 || nd < 0 || dim <= 0 || k <= 0) return -1;
#pragma omp parallel
  {
    hit_t *heap = (hit_t *)malloc(sizeof(hit_t) * (size_t)k);
#pragma omp for schedule(dynamic, 1)
    for (int64_t qi = 0; qi < nq; ++qi) {
      const float *qv = q + qi * dim;
      int64_t n = 0;
      for (int64_t d = 0; d < nd; ++d) {
        hit_t h;
        h.s = oracle_dot_f32(qv, docs + d * dim, dim) + 0.0f; /* -0.0 -> +0.0 */
        h.id = id_offset + d;
        if (!(h.s > -INFINITY)) continue; /* NaN / -inf never rank (HIP path: strict s > tau) */
        if (n < k) {
          heap[n++] = h;
          if (n == k)
            for (int64_t i = k / 2 - 1; i >= 0; --i) heap_sift_down(heap, k, i);
        } else if (hit_before(&h, &heap[0])) {
          heap[0] = h;
          heap_sift_down(heap, k, 0);
        }
      }
      qsort(heap, (size_t)n, sizeof(hit_t), hit_cmp);
      for (int64_t i = 0; i < k; ++i) {
        out_score[qi * k + i] = i < n ? heap[i].s : -FLT_MAX;
        out_id[qi * k + i] = i < n ? heap[i].id : -1;
      }
    }
    free(heap);
  }
  return 0;
}

/* faiss-style evaluation of Flat-IP: the caller computes a block of scores with BLAS sgemm
 * (scores[nq, nb] = Q . D[b0:b0+nb]^T); this routine folds the block into per-query min-heaps of the k
 * best (faiss HeapResultHandler), OpenMP over queries.  heap_n[q] = filled slots.  Finish with
 * oracle_heap_finalize_f32 (sorts each heap: score desc, id asc; pads -FLT_MAX / -1). */
int oracle_heap_update_f32(const float *scores, int64_t nq, int64_t nb, int64_t id0, int64_t k,
                           float *heap_s, int64_t *heap_i, int64_t *heap_n) {
#pragma omp parallel for schedule(static)
  for (int64_t q = 0; q < nq; ++q) {
    hit_t *h = (hit_t *)malloc(sizeof(hit_t) * (size_t)k);
    int64_t n = heap_n[q];
    for (int64_t j = 0; j < n; ++j) {
      h[j].s = heap_s[q * k + j];
      h[j].id = heap_i[q * k + j];
    }
    const float *row = scores + q * nb;
    for (int64_t d = 0; d < nb; ++d) {
      hit_t c;
      c.s = row[d] + 0.0f;
      c.id = id0 + d;
      if (!(c.s > -INFINITY)) continue;
      if (n < k) {
        h[n++] = c;
        if (n == k)
          for (int64_t i = k / 2 - 1; i >= 0; --i) heap_sift_down(h, k, i);
      } else if (hit_before(&c, &h[0])) {
        h[0] = c;
        heap_sift_down(h, k, 0);
      }
    }
    for (int64_t j = 0; j < n; ++j) {
      heap_s[q * k + j] = h[j].s;
      heap_i[q * k + j] = h[j].id;
    }
    heap_n[q] = n;
    free(h);
  }
  return 0;
}

int oracle_heap_finalize_f32(int64_t nq, int64_t k, float *heap_s, int64_t *heap_i, const int64_t *heap_n) {
#pragma omp parallel for schedule(static)
  for (int64_t q = 0; q < nq; ++q) {
    hit_t *h = (hit_t *)malloc(sizeof(hit_t) * (size_t)k);
    int64_t n = heap_n[q];
    for (int64_t j = 0; j < n; ++j) {
      h[j].s = heap_s[q * k + j];
      h[j].id = heap_i[q * k + j];
    }
    qsort(h, (size_t)n, sizeof(hit_t), hit_cmp);
    for (int64_t j = 0; j < k; ++j) {
      heap_s[q * k + j] = j < n ? h[j].s : -FLT_MAX;
      heap_i[q * k + j] = j < n ? h[j].id : -1;
    }
    free(h);
  }
  return 0;
}

/* Merge nlists per-shard lists (scores [nlists,nq,k_in], ids, id<0 = padding) into
 * [nq,k_out]; same ordering rule.  New in the build: what a single
 * un-sharded search would have returned. */
int oracle_topk_merge_f32(const float *scores, const int64_t *ids, int64_t nlists, int64_t nq,
                          int64_t k_in, int64_t k_out, float *out_score, int64_t *out_id) {
  hit_t *all = (hit_t *)malloc(sizeof(hit_t) * (size_t)(nlists * k_in + 1));
  for (int64_t qi = 0; qi < nq; ++qi) {
    int64_t n = 0;
    for (int64_t l = 0; l < nlists; ++l)
      for (int64_t j = 0; j < k_in; ++j) {
        int64_t off = (l * nq + qi) * k_in + j;
        if (ids[off] < 0) continue;
        all[n].s = scores[off] + 0.0f;
        all[n].id = ids[off];
        ++n;
      }
    qsort(all, (size_t)n, sizeof(hit_t), hit_cmp);
    for (int64_t i = 0; i < k_out; ++i) {
      out_score[qi * k_out + i] = i < n ? all[i].s : -FLT_MAX;
      out_id[qi * k_out + i] = i < n ? all[i].id : -1;
    }
  }
  free(all);
  return 0;
}

/* ----------------------------------------------------------------------------
 * Residual quantisation encode.
 * Restates pq.get_rq_document_cluster (MEVI/pq.py:281-305) == the index path of
 * forward_rq (MEVI/pq.py:337-369) with dist_mode 'l2': compute_scores (pq.py:124-131)
 * is -sum((a - b)**2, -1); index = argmax of it; the residual loses the chosen
 * centroid after every level (rq_minus_centroids, pq.py:121-122).
 * torch leaves the summation order of .sum(-1) and the tie order of .max unspecified;
 * pinned here: d = fmaf chain over k = 0..dim-1 of (r_k - c_k)^2, lowest index on ties.
 * neg_dist (optional, [n, M, K]) receives -d, i.e. forward_rq's `proba`.
 * -------------------------------------------------------------------------- */
int oracle_rq_encode_f32(const float *x, int64_t n, int64_t dim, const float *codebook, int64_t M,
                         int64_t K, int32_t *codes, float *neg_dist) {
  if (n < 0 || dim <= 0 || M <= 0 || K <= 0) return -1;
#pragma omp parallel
  {
    float *r = (float *)malloc(sizeof(float) * (size_t)dim);
#pragma omp for schedule(static)
    for (int64_t i = 0; i < n; ++i) {
      memcpy(r, x + i * dim, sizeof(float) * (size_t)dim);
      for (int64_t j = 0; j < M; ++j) {
        const float *cb = codebook + j * K * dim;
        float best = INFINITY;
        int32_t arg = 0;
        for (int64_t c = 0; c < K; ++c) {
          float d = 0.0f;
          for (int64_t k = 0; k < dim; ++k) {
            float diff = r[k] - cb[c * dim + k];
            d = fmaf(diff, diff, d);
          }
          if (neg_dist) neg_dist[(i * M + j) * K + c] = -d;
          if (d < best) {
            best = d;
            arg = (int32_t)c;
          }
        }
        codes[i * M + j] = arg;
        for (int64_t k = 0; k < dim; ++k) r[k] = r[k] - cb[(int64_t)arg * dim + k];
      }
    }
    free(r);
  }
  return 0;
}

int oracle_num_threads(void) {
#ifdef _OPENMP
  return omp_get_max_threads();
#else
  return 1;
#endif
}
