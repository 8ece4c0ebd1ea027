"""Oracle for residual quantisation (TEST INFRASTRUCTURE ONLY -- see oracle/__init__.py).

* `rq_encode`     -- oracle/mevi_oracle.c:oracle_rq_encode_f32 (restates MEVI/pq.py:281-305, 337-369)
* `rq_beam_search`-- numpy restatement of pq.beam_search with rq_topk_score='prod' (MEVI/pq.py:613-713)
* `reconstruct`   -- pq.get_reconstruct_vector for 'rq' (MEVI/pq.py:768-784): sum of the chosen centroids
* `cluster_dict`  -- the dict[tuple -> list[int]] / dict[int -> tuple] of get_document_cluster (pq.py:217-247)
"""
from ctypes import c_int, c_int64, c_void_p

import numpy as np

from . import dense as _d


def _lib():
    L = _d.lib()
    L.oracle_rq_encode_f32.restype = c_int
    L.oracle_rq_encode_f32.argtypes = [c_void_p, c_int64, c_int64, c_void_p, c_int64, c_int64, c_void_p, c_void_p]
    return L


def rq_encode(x, codebook, return_neg_dist=False):
    x = np.ascontiguousarray(x, dtype=np.float32)
    cb = np.ascontiguousarray(codebook, dtype=np.float32)
    M, K, dim = cb.shape
    n = x.shape[0]
    codes = np.empty((n, M), np.int32)
    nd = np.empty((n, M, K), np.float32) if return_neg_dist else None
    rc = _lib().oracle_rq_encode_f32(_d._p(x), n, dim, _d._p(cb), M, K, _d._p(codes),
                                     _d._p(nd) if nd is not None else None)
    assert rc == 0
    return (codes, nd) if return_neg_dist else codes


def reconstruct(codes, codebook):
    cb = np.asarray(codebook, dtype=np.float32)
    codes = np.asarray(codes)
    out = np.zeros(codes.shape[:-1] + (cb.shape[-1],), np.float32)
    for j in range(cb.shape[0]):     # torch.sum over the level axis adds level 0 first
        out = out + cb[j][codes[..., j]]
    return out


def cluster_dict(codes, start=0):
    cluster, mapping = {}, {}
    for i, c in enumerate(np.asarray(codes).tolist()):
        key = tuple(c)
        cluster.setdefault(key, []).append(i + start)
        mapping[i + start] = key
    return cluster, mapping


def rq_beam_search(x, codebook, num_return_sequences):
    """Top-R code paths per row: per level softmax(-dist) times the running beam probability,
    top-R over beams x K (pq.py:640-700).  Returns (labels i32[n,R,M], scores f32[n,R])."""
    x = np.asarray(x, dtype=np.float32)
    cb = np.asarray(codebook, dtype=np.float32)
    M, K, dim = cb.shape
    n, R = x.shape[0], num_return_sequences
    beam_scores = np.ones((n, 1), np.float32)
    resid = x[:, None, :].copy()
    labels = np.zeros((n, 1, 0), np.int32)
    for j in range(M):
        diff = resid[:, :, None, :] - cb[j][None, None, :, :]
        neg = -(diff * diff).sum(-1, dtype=np.float32)
        neg = neg - neg.max(-1, keepdims=True)
        p = np.exp(neg)
        p = (p / p.sum(-1, keepdims=True)).astype(np.float32)
        p = beam_scores[:, :, None] * p
        nb = p.shape[1]
        flat = p.reshape(n, nb * K)
        if R < nb * K:
            order = np.argsort(-flat, axis=1, kind="stable")[:, :R]
            prev, code = order // K, order % K
            beam_scores = np.take_along_axis(flat, order, 1)
            labels = np.concatenate([np.take_along_axis(labels, prev[:, :, None], 1), code[:, :, None].astype(np.int32)], -1)
            if j != M - 1:
                resid = np.take_along_axis(resid, prev[:, :, None], 1) - cb[j][code]
        else:
            beam_scores = flat
            code = np.tile(np.arange(K), nb)
            labels = np.concatenate([np.repeat(labels, K, axis=1),
                                     np.broadcast_to(code[None, :, None], (n, nb * K, 1)).astype(np.int32)], -1)
            if j != M - 1:
                resid = np.repeat(resid, K, axis=1) - cb[j][code][None]
    return labels, beam_scores
