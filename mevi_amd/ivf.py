"""IVF-Flat behind `faiss_search.py --param IVF<n>,Flat` (the script's DEFAULT factory string, MEVI/faiss_search.py:89;
SURVEY 8(f).4).

faiss builds `IndexIVFFlat(IndexFlatIP, dim, nlist, METRIC_INNER_PRODUCT)`: a coarse quantiser of `nlist` centroids
trained by k-means on at most 256 * nlist sampled rows (25 iterations, assignment by the quantiser = largest inner
product, centroid = mean of its rows), every document stored in the inverted list of its best centroid, and a search
that scans only the `nprobe` (default 1) best lists of the query.  faiss-cpu 1.7.4 is absent from the reference tree and
its k-means is seeded from its own RNG, so list contents cannot be matched bit for bit; what IS defined -- and tested
against the oracle -- given the centroids:
  * assignment: argmax_c <d, centroid_c> (exact f32 chains, lowest list on ties),
  * search: the exact top-k (score desc, id asc, faiss padding) among the documents of the query's `nprobe` best lists.
Because the result is approximate by construction, `search` also runs the exact search and reports recall against it
(stderr, so the script's stdout stays the reference's).  `nprobe` comes from MEVI_IVF_NPROBE (faiss default: 1).
"""
import os
import re
import sys

import numpy as np
import torch

from . import dense, ops, rq

MAX_POINTS_PER_CENTROID = 256     # faiss ClusteringParameters defaults
NITER = 25


def parse_factory(param):
    """nlist of an 'IVF<n>,Flat' factory string, else None (every other index type is served by exact search)."""
    m = re.fullmatch(r"\s*IVF(\d+)\s*,\s*Flat\s*", str(param))
    return int(m.group(1)) if m else None


def assign(x, centroids, chunk=1 << 18):
    """argmax_c <x, centroid_c> per row (exact f32 chains; lowest c on ties) -> i32 [n]."""
    out = torch.empty(x.shape[0], dtype=torch.int32, device=x.device)
    for a in range(0, x.shape[0], chunk):
        out[a:a + chunk] = torch.argmax(ops.linear(x[a:a + chunk].contiguous(), centroids), dim=1).to(torch.int32)
    return out


def train_centroids(docs, nlist, seed=1234):
    """k-means as faiss's Clustering runs it for an inner-product IVF: sample, random distinct rows as the first
    centroids, NITER x (assign by inner product, mean); an empty list takes a row of the largest one."""
    n, dim = docs.shape
    gen = torch.Generator(device=docs.device).manual_seed(int(seed))
    perm = torch.randperm(n, generator=gen, device=docs.device)
    xs = docs[perm[:min(n, MAX_POINTS_PER_CENTROID * nlist)]].contiguous()
    cent = xs[:nlist].clone() if xs.shape[0] >= nlist else torch.cat(
        [xs, torch.zeros((nlist - xs.shape[0], dim), dtype=torch.float32, device=docs.device)])
    for _ in range(NITER):
        lab = assign(xs, cent)
        new, counts, _ = rq.cluster_means(xs, lab, nlist, old=cent)
        empty = torch.nonzero(counts == 0).flatten()
        if empty.numel():
            big = int(torch.argmax(counts).item())
            rows = torch.nonzero(lab == big).flatten()
            pick = rows[torch.randperm(rows.numel(), generator=gen, device=docs.device)[:empty.numel()]]
            new[empty[:pick.numel()]] = xs[pick]
        cent = new
    return cent.contiguous()


class IVFFlatIndex:
    """`index.train(doc); index.add(doc)` of the IVF factory string: centroids, documents stored list by list."""

    def __init__(self, docs, nlist, centroids=None, seed=1234):
        assert docs.is_cuda and docs.dtype == torch.float32 and docs.dim() == 2
        self.nlist = nlist
        self.centroids = train_centroids(docs, nlist, seed) if centroids is None else centroids.to(docs.device).contiguous()
        self.list_of = assign(docs, self.centroids)
        order = torch.argsort(self.list_of.long(), stable=True)           # ids ascending inside a list
        self.ids = order
        self.docs = docs[order].contiguous()
        self.offsets = torch.zeros(nlist + 1, dtype=torch.int64)
        self.offsets[1:] = torch.cumsum(torch.bincount(self.list_of.long(), minlength=nlist), 0).cpu()

    def search(self, query, k, nprobe=1):
        """(scores f32 [nq, k] desc, ids i64 [nq, k]; -FLT_MAX / -1 padding when the probed lists hold fewer than k)."""
        nq = query.shape[0]
        nprobe = max(1, min(int(nprobe), self.nlist))
        dev = query.device
        probe = dense.ip_topk(query, self.centroids, nprobe)[1]            # [nq, nprobe] best lists (score desc, list asc)
        neg = -torch.finfo(torch.float32).max
        cand_s = torch.full((nprobe, nq, k), neg, dtype=torch.float32, device=dev)
        cand_i = torch.full((nprobe, nq, k), -1, dtype=torch.int64, device=dev)
        probe_h = probe.cpu().numpy()
        off = self.offsets.numpy()
        for l in np.unique(probe_h):
            a, b = int(off[l]), int(off[l + 1])
            if a == b:
                continue
            slot, qsel = np.nonzero(probe_h.T == l)                        # which probe slot of which query
            qs = torch.from_numpy(qsel).to(dev)
            s, i = dense.ip_topk(query[qs].contiguous(), self.docs[a:b], k)
            gid = torch.where(i >= 0, self.ids[a:b][i.clamp_min(0)], i)    # list-local row -> document id
            cand_s[torch.from_numpy(slot).to(dev), qs] = s
            cand_i[torch.from_numpy(slot).to(dev), qs] = gid
        if nprobe == 1:
            out_s, out_i = cand_s[0], cand_i[0]
            # inside one list the local order is (score desc, LOCAL row asc) = (score desc, id asc): ids ascend in a list
            return out_s, out_i
        return dense.topk_merge(cand_s, cand_i, k)


def recall_report(approx_ids, exact_ids, cutoffs=(1, 10, 100, 1000)):
    """|approx top-c  ∩  exact top-c| / c averaged over queries, for every cutoff <= k."""
    out = {}
    k = exact_ids.shape[1]
    for c in cutoffs:
        if c > k:
            continue
        hit = (approx_ids[:, :c, None] == exact_ids[:, None, :c]) & (exact_ids[:, None, :c] >= 0)
        denom = (exact_ids[:, :c] >= 0).sum(1).clamp_min(1)
        out[c] = float((hit.any(1).sum(1).double() / denom.double()).mean().item())
    return out


def search(query, docs, k, nlist, nprobe=None, report=True):
    """faiss_search.search for 'IVF<n>,Flat' on device tensors: train + add + search, recall vs exact on stderr."""
    if nprobe is None:
        nprobe = int(os.environ.get("MEVI_IVF_NPROBE", "1"))
    index = IVFFlatIndex(docs, nlist)
    s, i = index.search(query, k, nprobe)
    if report:
        es, ei = dense.DenseIndex(docs).search(query, min(k, 4096))
        # in chunks of queries: the comparison tensor is [nq, c, c]
        rec = {}
        for a in range(0, query.shape[0], 256):
            part = recall_report(i[a:a + 256, :ei.shape[1]], ei[a:a + 256])
            for c, v in part.items():
                rec[c] = rec.get(c, 0.0) + v * min(256, query.shape[0] - a)
        rec = {c: v / max(1, query.shape[0]) for c, v in rec.items()}
        sizes = (index.offsets[1:] - index.offsets[:-1])
        print(f"[mevi_amd] IVF{nlist},Flat nprobe={nprobe}: recall vs exact search " +
              ", ".join(f"@{c} {v:.4f}" for c, v in rec.items()) +
              f"; lists: min {int(sizes.min())} / mean {float(sizes.float().mean()):.0f} / max {int(sizes.max())} documents",
              file=sys.stderr)
    return s, i
