"""Oracle for the dense arm (TEST INFRASTRUCTURE ONLY -- see oracle/__init__.py).

Two restatements of `faiss_search.search(..., param="Flat")`
(MEVI/faiss_search.py:13-21; faiss-cpu==1.7.4 IndexFlatIP, third-party, absent):

* `ip_topk_exact`  -- oracle/mevi_oracle.c: sequential fmaf chain per score,
  (score desc, id asc) order.  Bit-exact contract of the HIP kernel.
* `ip_topk_blas`   -- numpy sgemm (BLAS, all host cores) + partial sort: how
  faiss itself evaluates Flat-IP (blocked sgemm + heap).  Summation order is
  BLAS's, so it agrees with the exact chain only to f32 rounding; used as the
  CPU baseline in bench.py and as a tolerance cross-check in tests.
"""
import ctypes
import os
import subprocess
from ctypes import c_int, c_int64, c_void_p

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
LIB = os.path.join(HERE, "libmevi_oracle.so")
_lib = None

FLT_MAX = np.finfo(np.float32).max


def build(force=False):
    src = os.path.join(HERE, "mevi_oracle.c")
    if force or not os.path.exists(LIB) or os.path.getmtime(LIB) < os.path.getmtime(src):
        subprocess.run(["make", "-C", HERE, "-B" if force else "-s"], check=True,
                       stdout=subprocess.DEVNULL)
    return LIB


def lib():
    global _lib
    if _lib is None:
        build()
        L = ctypes.CDLL(LIB)
        L.oracle_ip_topk_f32.restype = c_int
        L.oracle_ip_topk_f32.argtypes = [c_void_p, c_int64, c_void_p, c_int64, c_int64, c_int64, c_int64,
                                         c_void_p, c_void_p]
        L.oracle_topk_merge_f32.restype = c_int
        L.oracle_topk_merge_f32.argtypes = [c_void_p, c_void_p, c_int64, c_int64, c_int64, c_int64,
                                            c_void_p, c_void_p]
        L.oracle_num_threads.restype = c_int
        L.oracle_heap_update_f32.restype = c_int
        L.oracle_heap_update_f32.argtypes = [c_void_p, c_int64, c_int64, c_int64, c_int64, c_void_p, c_void_p, c_void_p]
        L.oracle_heap_finalize_f32.restype = c_int
        L.oracle_heap_finalize_f32.argtypes = [c_int64, c_int64, c_void_p, c_void_p, c_void_p]
        _lib = L
    return _lib


def _p(a):
    return a.ctypes.data_as(c_void_p)


def ip_topk_exact(query, docs, k, id_offset=0):
    q = np.ascontiguousarray(query, dtype=np.float32)
    d = np.ascontiguousarray(docs, dtype=np.float32)
    nq, dim = q.shape
    out_s = np.empty((nq, k), np.float32)
    out_i = np.empty((nq, k), np.int64)
    rc = lib().oracle_ip_topk_f32(_p(q), nq, _p(d), d.shape[0], dim, k, id_offset, _p(out_s), _p(out_i))
    assert rc == 0
    return out_s, out_i


def topk_merge(scores, ids, k_out):
    s = np.ascontiguousarray(scores, dtype=np.float32)
    i = np.ascontiguousarray(ids, dtype=np.int64)
    nlists, nq, k_in = s.shape
    out_s = np.empty((nq, k_out), np.float32)
    out_i = np.empty((nq, k_out), np.int64)
    rc = lib().oracle_topk_merge_f32(_p(s), _p(i), nlists, nq, k_in, k_out, _p(out_s), _p(out_i))
    assert rc == 0
    return out_s, out_i


def ip_topk_blas(query, docs, k, id_offset=0, block=16384, timing=None):
    """faiss-style Flat-IP on the CPU, the way faiss evaluates it: blocked sgemm (numpy -> BLAS, all host
    threads) and per-query heaps of the k best updated block by block (oracle_heap_update_f32, OpenMP over
    queries).  Ties by ascending id.  Summation order is BLAS's.  `timing` (a dict) receives the seconds spent in
    the sgemm blocks and in the heap updates separately."""
    import time

    q = np.ascontiguousarray(query, dtype=np.float32)
    d = np.asarray(docs, dtype=np.float32)
    nq = q.shape[0]
    L = lib()
    heap_s = np.empty((nq, k), np.float32)
    heap_i = np.empty((nq, k), np.int64)
    heap_n = np.zeros(nq, np.int64)
    t_gemm = t_heap = 0.0
    for b0 in range(0, d.shape[0], block):
        t0 = time.perf_counter()
        sc = np.ascontiguousarray(q @ d[b0:b0 + block].T)
        t1 = time.perf_counter()
        L.oracle_heap_update_f32(_p(sc), nq, sc.shape[1], id_offset + b0, k, _p(heap_s), _p(heap_i), _p(heap_n))
        t_gemm, t_heap = t_gemm + (t1 - t0), t_heap + (time.perf_counter() - t1)
    t1 = time.perf_counter()
    L.oracle_heap_finalize_f32(nq, k, _p(heap_s), _p(heap_i), _p(heap_n))
    if timing is not None:
        timing["sgemm_s"] = timing.get("sgemm_s", 0.0) + t_gemm
        timing["heap_s"] = timing.get("heap_s", 0.0) + t_heap + (time.perf_counter() - t1)
    return heap_s, heap_i


def num_threads():
    return lib().oracle_num_threads()


def pair_dot(query, docs):
    """<query, docs[j]> for every row j with the sequential fmaf chain (oracle_dot_f32): the scoring call of the
    fine stage, DocumentEncoder.generate(q, p_reps=rows).scores = torch.matmul (MEVI/document_encoder.py:128-132,
    213-226), with the summation order pinned like the dense arm's."""
    from ctypes import c_float

    L = lib()
    L.oracle_dot_f32.restype = c_float
    L.oracle_dot_f32.argtypes = [c_void_p, c_void_p, c_int64]
    q = np.ascontiguousarray(query, dtype=np.float32)
    d = np.ascontiguousarray(docs, dtype=np.float32).reshape(-1, q.shape[0])
    return np.array([L.oracle_dot_f32(_p(q), d[j].ctypes.data, q.shape[0]) for j in range(d.shape[0])], np.float32)


def fine_stage(query, emb, doc_cluster, beam_codes):
    """Fine stage of infer() for ONE query (MEVI/main_models.py:3921-4013, single-cluster documents): the documents
    of the beam clusters in beam order (a repeated cluster is listed -- and scored -- again, an absent one skipped),
    scored by q.d, sorted descending.  Pinned here: fmaf-chain scores, ties by ascending doc id (torch.sort leaves
    the order of equal scores unspecified).  Returns (doc ids i64, scores f32, ndoc)."""
    docs = []
    for code in beam_codes:
        cur = doc_cluster.get(tuple(int(x) for x in code))
        if cur is not None:
            docs += list(cur)
    docs = np.array(docs, dtype=np.int64)
    if docs.size == 0:
        return docs, np.zeros(0, np.float32), 0
    sc = pair_dot(query, np.asarray(emb, dtype=np.float32)[docs])
    order = np.lexsort((docs, -sc))
    return docs[order], sc[order], int(docs.size)


def ivf_flat_search(query, docs, centroids, k, nprobe):
    """IVF-Flat given its centroids (faiss IndexIVFFlat with an IndexFlatIP quantiser, METRIC_INNER_PRODUCT; the
    factory string 'IVF<n>,Flat' of MEVI/faiss_search.py:13-21,89): a document lives in the list of its best centroid
    (largest inner product, lowest list on ties), a query scans its `nprobe` best lists and returns the exact top-k among
    their documents (score desc, id asc; -FLT_MAX / -1 padding).  Scores are the pinned fmaf chains.  Returns
    (scores, ids, list_of)."""
    q = np.ascontiguousarray(query, np.float32)
    d = np.ascontiguousarray(docs, np.float32)
    c = np.ascontiguousarray(centroids, np.float32)
    _, best = ip_topk_exact(d, c, 1)
    list_of = best[:, 0]
    _, probe = ip_topk_exact(q, c, nprobe)
    out_s = np.full((len(q), k), -FLT_MAX, np.float32)
    out_i = np.full((len(q), k), -1, np.int64)
    for i in range(len(q)):
        rows = np.flatnonzero(np.isin(list_of, probe[i]))
        if rows.size == 0:
            continue
        s = pair_dot(q[i], d[rows])
        order = np.lexsort((rows, -s))[:k]
        out_s[i, :len(order)], out_i[i, :len(order)] = s[order], rows[order]
    return out_s, out_i, list_of
