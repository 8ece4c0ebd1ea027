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


def ip_topk_blas(query, docs, k, id_offset=0, block=65536):
    """faiss-style Flat-IP: blocked sgemm + running top-k (ties by ascending id)."""
    q = np.ascontiguousarray(query, dtype=np.float32)
    d = np.asarray(docs, dtype=np.float32)
    nq = q.shape[0]
    best_s = np.full((nq, 0), 0, np.float32)
    best_i = np.full((nq, 0), 0, np.int64)
    for b0 in range(0, d.shape[0], block):
        sc = q @ d[b0:b0 + block].T
        ids = np.broadcast_to(np.arange(b0, b0 + sc.shape[1], dtype=np.int64) + id_offset, sc.shape)
        cs = np.concatenate([best_s, sc], axis=1)
        ci = np.concatenate([best_i, ids], axis=1)
        if cs.shape[1] > k:
            part = np.argpartition(-cs, k - 1, axis=1)[:, :k]
            cs = np.take_along_axis(cs, part, 1)
            ci = np.take_along_axis(ci, part, 1)
        best_s, best_i = cs, ci
    order = np.lexsort((best_i, -best_s), axis=1)
    best_s = np.take_along_axis(best_s, order, 1)
    best_i = np.take_along_axis(best_i, order, 1)
    if best_s.shape[1] < k:
        pad = k - best_s.shape[1]
        best_s = np.concatenate([best_s, np.full((nq, pad), -FLT_MAX, np.float32)], 1)
        best_i = np.concatenate([best_i, np.full((nq, pad), -1, np.int64)], 1)
    return best_s, best_i


def num_threads():
    return lib().oracle_num_threads()
