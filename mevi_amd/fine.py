"""Coarse + fine stages of `infer()` after the beam search (MEVI/main_models.py:3736-4098).

coarse  -- log (text, clusters, gt codes, scores); rank of every gt cluster in the beam list;
           ndoc = sum of the beam clusters' sizes                              (:3758-3780)
fine    -- twin-tower query embedding . embeddings of the documents inside the beam clusters,
           concatenated in beam order, sorted descending; gt-document scores; hard-negative log
           line truncated to save_hard_neg                                      (:3781-4089)

The reference walks Python dicts and a CPU memmap per cluster with one H2D copy and one matmul per
<= 1024 rows; here the cluster -> doc list is a CSR index, all (query, doc) pairs of a batch are
scored by ONE pair_dot launch on the HBM-resident embedding matrix and ordered by ONE segmented
sort.  Order: score descending, ties by ascending doc id (torch.sort leaves ties unspecified).
"""
import numpy as np
import torch

from . import ops
from .rq import ClusterIndex

MAX_SEGMENT = 16384


def f32_repr(values):
    """','.join(str(x.item()) for x in f32 tensor): the reference's way of printing scores."""
    from .io import join_f32

    return join_f32(values)


class FineStage:
    def __init__(self, doc_embeddings, cluster_index: ClusterIndex):
        assert doc_embeddings.is_cuda and doc_embeddings.dtype == torch.float32 and doc_embeddings.dim() == 2
        self.emb = doc_embeddings
        self.index = cluster_index
        self.dev = doc_embeddings.device
        self._dev_index = None

    def candidates(self, beam_codes):
        """beam_codes i64[B, R, M] -> (cand_doc_ids i64[total], cand_query i64[total], seg i64[B+1], ndoc i64[B])
        with each query's candidates in beam order (cluster after cluster, ids ascending inside)."""
        idx = self.index
        codes = np.asarray(beam_codes, dtype=np.int64)
        B, R, _ = codes.shape
        keys = ClusterIndex.code_keys(codes, idx.K).reshape(-1)
        pos = np.searchsorted(idx.keys, keys)
        pos_c = np.minimum(pos, max(len(idx.keys) - 1, 0))
        found = (pos < len(idx.keys)) & (idx.keys[pos_c] == keys) if len(idx.keys) else np.zeros_like(keys, bool)
        start = np.where(found, idx.offsets[pos_c], 0)
        size = np.where(found, idx.offsets[pos_c + 1] - idx.offsets[pos_c], 0)
        total = int(size.sum())
        first = np.cumsum(size) - size
        flat = np.arange(total, dtype=np.int64) - np.repeat(first, size) + np.repeat(start, size)
        cand = idx.doc_ids[flat]
        cand_q = np.repeat(np.repeat(np.arange(B, dtype=np.int64), R), size)
        ndoc = size.reshape(B, R).sum(1)
        seg = np.concatenate([[0], np.cumsum(ndoc)]).astype(np.int64)
        return cand, cand_q, seg, ndoc

    def candidates_device(self, beam_codes, with_beam=False):
        """candidates() with the expansion done on the GPU (the CSR arrays are uploaded once): the candidate list of
        a batch is tens of millions of ids -- building it with numpy and shipping two i64 arrays over PCIe
        dominated the fine stage.  Returns (cand i64[total] CUDA, cand_q i64[total] CUDA, seg, ndoc) with seg / ndoc
        as host arrays."""
        idx = self.index
        if self._dev_index is None:
            self._dev_index = tuple(torch.from_numpy(np.ascontiguousarray(a, dtype=np.int64)).to(self.dev)
                                    for a in (idx.keys, idx.offsets, idx.doc_ids))
        keys_t, offs_t, docs_t = self._dev_index
        codes = torch.as_tensor(np.asarray(beam_codes, dtype=np.int64), device=self.dev)
        B, R, M = codes.shape
        w = torch.tensor([idx.K ** (M - 1 - j) for j in range(M)], dtype=torch.int64, device=self.dev)
        keys = (codes * w).sum(-1).reshape(-1)
        n = keys_t.numel()
        if n == 0:
            empty = (torch.zeros(0, dtype=torch.int64, device=self.dev),) * 2 + (np.zeros(B + 1, np.int64), np.zeros(B, np.int64))
            return empty + (torch.zeros(0, dtype=torch.int64, device=self.dev),) if with_beam else empty
        pos = torch.searchsorted(keys_t, keys)
        pos_c = pos.clamp(max=n - 1)
        found = (pos < n) & (keys_t[pos_c] == keys)
        zero = torch.zeros_like(keys)
        start = torch.where(found, offs_t[pos_c], zero)
        size = torch.where(found, offs_t[pos_c + 1] - offs_t[pos_c], zero)
        first = torch.cumsum(size, 0) - size
        total = int(size.sum().item())
        flat = torch.arange(total, dtype=torch.int64, device=self.dev) + torch.repeat_interleave(start - first, size)
        cand = docs_t[flat]
        cand_q = torch.repeat_interleave(torch.arange(B, dtype=torch.int64, device=self.dev).repeat_interleave(R), size)
        ndoc = size.view(B, R).sum(1).cpu().numpy()
        seg = np.concatenate([[0], np.cumsum(ndoc)]).astype(np.int64)
        if with_beam:      # which entry of the query's beam list each candidate came from
            return cand, cand_q, seg, ndoc, torch.repeat_interleave(torch.arange(R, dtype=torch.int64, device=self.dev).repeat(B), size)
        return cand, cand_q, seg, ndoc

    def rerank(self, query_emb, beam_codes, aggregate=None, beam_weights=None, doc_proba=None, ratio=0.0, beam_recon=None):
        """query_emb f32[B, dim] (CUDA).  Returns per query: (doc ids i64 ndarray, scores f32 ndarray) -- views of
        one host copy of the sorted batch --, and ndoc (candidates incl. repeats).

        aggregate 'add' | 'max' (--doc_multiclus > 1, main_models.py:3997-4011): a document reached through several
        beam clusters is listed once, with the sum (sequential f32 adds, as the reference accumulates) or the maximum
        of its per-cluster scores -- which are all the same q.d.

        beam_weights f32[B, R] (--use_topic_model 1, main_models.py:3539-3552,3952: get_inference_scores with
        topic_score_ratio 0): every document's score is its cluster's beam score times q.d (one f32 multiply); with
        `doc_proba` f32[N] and ratio > 0 the reference's full form w * (ratio * doc_proba[d] + (1 - ratio) * q.d), each
        operation an f32 tensor operation in that order.  With multi-cluster documents (--doc_multiclus > 1) the document
        probability belongs to the (document, cluster) pair -- all_doc_proba[d].gather(inclus_index), main_models.py:3944-
        3985,3311-3372: <reconstruct vector of THAT cluster's code path, emb[d]> -- and is computed per candidate from
        `beam_recon` f32 [B * R, dim], the reconstruct vectors of the beam code paths (exact fmaf chains)."""
        cand_t, cand_q, seg, ndoc, cand_beam = self.candidates_device(beam_codes, with_beam=True)
        B = len(ndoc)
        if cand_t.numel() == 0:
            return [(np.zeros(0, np.int64), np.zeros(0, np.float32)) for _ in range(B)], ndoc
        sc = ops.pair_dot(query_emb, cand_q, self.emb, cand_t)
        if beam_weights is not None:
            w = torch.as_tensor(beam_weights, dtype=torch.float32, device=self.dev)
            if beam_recon is not None and ratio:
                R = w.shape[1]
                dp = ops.pair_dot(beam_recon, cand_q * R + cand_beam, self.emb, cand_t)
                sc = ratio * dp + (1 - ratio) * sc
            elif doc_proba is not None and ratio:
                sc = ratio * doc_proba[cand_t] + (1 - ratio) * sc
            sc = w[cand_q, cand_beam] * sc
        seg_len = ndoc
        if aggregate is not None:
            if aggregate not in ("add", "max"):
                raise ValueError(aggregate)
            if int(ndoc.max()) <= MAX_SEGMENT:          # unique + aggregate + final sort in one kernel (csrc/rerank.hip)
                s_u, i_u, cnt = ops.segment_aggregate_sort(sc, cand_t, torch.from_numpy(seg), int(ndoc.max()), aggregate)
                s_u, i_u, cnt = s_u.cpu().numpy(), i_u.cpu().numpy(), cnt.cpu().numpy()
                return [(i_u[a:a + c], s_u[a:a + c]) for a, c in zip(seg[:-1], cnt)], ndoc
            # a query with more candidates than the LDS sort holds (rare): the same merge with device-wide sorts
            n = self.emb.shape[0]
            key, inverse, count = torch.unique(cand_q * n + cand_t, sorted=True, return_inverse=True, return_counts=True)
            one = torch.empty(key.shape, dtype=torch.float32, device=self.dev)
            one[inverse] = sc                                    # duplicates carry the same bits
            if aggregate == "add":
                acc = one.clone()
                for t in range(1, int(count.max().item())):
                    acc = torch.where(count > t, acc + one, acc)
                one = acc
            sc, cand_t = one, key % n
            seg_len = torch.bincount(key // n, minlength=B).cpu().numpy()
            seg = np.concatenate([[0], np.cumsum(seg_len)]).astype(np.int64)
        longest = int(seg_len.max())
        if longest <= MAX_SEGMENT:
            s_sorted, i_sorted = ops.segment_sort_desc(sc, cand_t, torch.from_numpy(seg), longest)
        else:
            # a beam cluster list above the LDS sort capacity: order the flat list by
            # (segment, -score, id) with one device sort (rare: > 16384 candidates for one query)
            segid = torch.from_numpy(np.repeat(np.arange(B, dtype=np.int64), seg_len)).to(self.dev)
            o = torch.argsort(cand_t, stable=True)
            o = o[torch.argsort(-sc[o], stable=True)]
            o = o[torch.argsort(segid[o], stable=True)]
            s_sorted, i_sorted = sc[o], cand_t[o]
        s_sorted, i_sorted = s_sorted.cpu().numpy(), i_sorted.cpu().numpy()
        out = [(i_sorted[a:b], s_sorted[a:b]) for a, b in zip(seg[:-1], seg[1:])]
        return out, ndoc

    def gt_scores(self, query_emb, gt_doc_ids):
        """q . emb[gt] for the hard-negative log line (main_models.py:4024-4045): list of f32 arrays."""
        lens = [len(g) for g in gt_doc_ids]
        if sum(lens) == 0:
            return [np.zeros(0, np.float32) for _ in lens]
        ib = torch.tensor([d for g in gt_doc_ids for d in g], dtype=torch.int64, device=self.dev)
        ia = torch.from_numpy(np.repeat(np.arange(len(lens), dtype=np.int64), lens)).to(self.dev)
        sc = ops.pair_dot(query_emb, ia, self.emb, ib).cpu().numpy()
        return np.split(sc, np.cumsum(lens)[:-1])


def coarse_ranks(beam_codes, gt_codes):
    """index of every gt cluster in the query's beam list, None when absent (main_models.py:3773-3774)."""
    d = [list(map(int, c)) for c in beam_codes]
    return tuple(d.index(list(map(int, g))) if list(map(int, g)) in d else None for g in gt_codes)


def fine_ranks(sorted_docs, gt_doc_ids):
    """first position of every gt doc in the ranked list, None when absent (main_models.py:4016-4019)."""
    docs = np.asarray(sorted_docs, dtype=np.int64)
    out = []
    for g in gt_doc_ids:
        hit = np.flatnonzero(docs == int(g))
        out.append(int(hit[0]) if hit.size else None)
    return tuple(out)
