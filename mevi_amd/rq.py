"""Residual quantisation on MI355X: the 'rq' / 'l2' slice of the reference's
`ProductQuantization` (MEVI/pq.py:15) that the eval scripts use, plus the cluster index
the fine stage consumes.

Reference surface mirrored (same names and argument meaning):
  ProductQuantization(pq_type, subvector_num, subvector_bits, dist_mode, emb_size, ...)
    .get_codebook()                                           pq.py:133-141
    .initialize(index_file, doc_emb, rank, seed, pq_cluster_path, encode_batch_size)
         -- only the "load rqcodebook*.pt" branch (pq.py:459-463) + broadcast (:484)
    .get_document_cluster(doc_embeddings, rank, nrank, batch_size, return_mapping)  pq.py:217-247
    .forward(vecs) -> index   (forward_rq, pq.py:337-369; proba/loss are training outputs)
    .get_reconstruct_vector(index)                            pq.py:768-784
Training (EMA / k-means / gumbel), 'pq' / 'opq' and 'ip' / 'iptol2' are out of scope.
"""
from collections import defaultdict

import numpy as np
import torch

from . import hip


_LAST_ENCODE = {}


def last_encode_stats():
    """What the last rq_encode call did: path taken and, for the matrix-core path, {ambiguity records, rows re-encoded by
    the exact kernel, ambiguous row-levels} (read back with a stream synchronisation: bench / tests only)."""
    st = dict(_LAST_ENCODE)
    ws = st.pop("_ws", None)
    if ws is not None:
        import ctypes

        out = (ctypes.c_int64 * 3)()
        with hip.device_guard(ws.device):
            hip.check(hip.lib().mevi_rq_encode_fast_stats(hip.ptr(ws), st["n"], st["dim"], st["M"], st["K"], out, hip.stream_ptr()),
                      "mevi_rq_encode_fast_stats")
        st.update(records=int(out[0]), rows_reencoded_exactly=int(out[1]), ambiguous_row_levels=int(out[2]))
    return st


KEEP_ENCODE_WORKSPACE = False     # bench / tests set this: last_encode_stats() then reads the fast path's record counters


def rq_encode(x, codebook, mode=None):
    """codes i32[n, M] of x f32[n, dim] against codebook f32[M, K, dim] (CUDA tensors).

    mode (default MEVI_RQ or 'fast'): 'fast' = the matrix-core encoder with exact re-check (csrc/rq_fast.hip) wherever its
    shape constraints hold, 'exact' = the f32 VALU kernel (csrc/rq_encode.hip).  Both return the oracle's codes bit for bit."""
    import os

    hip.require_gpu()
    assert x.is_cuda and codebook.is_cuda and x.dtype == torch.float32 and codebook.dtype == torch.float32
    x = x.contiguous()
    codebook = codebook.contiguous()
    M, K, dim = codebook.shape
    assert x.dim() == 2 and x.shape[1] == dim
    n = x.shape[0]
    mode = mode or os.environ.get("MEVI_RQ", "fast")
    assert mode in ("fast", "exact"), mode
    codes = torch.empty((n, M), dtype=torch.int32, device=x.device)
    L = hip.lib()
    _LAST_ENCODE.clear()
    with hip.device_guard(x.device):
        nbytes = L.mevi_rq_encode_fast_workspace_bytes(n, dim, M, K) if mode == "fast" and n > 0 else 0
        if nbytes:
            ws = torch.empty(nbytes, dtype=torch.uint8, device=x.device)
            st = L.mevi_rq_encode_fast_f32(hip.ptr(x), n, dim, hip.ptr(codebook), M, K, hip.ptr(codes), hip.ptr(ws), nbytes,
                                           hip.stream_ptr())
            hip.check(st, "mevi_rq_encode_fast_f32")
            _LAST_ENCODE.update(path="fast", n=n, dim=dim, M=M, K=K)
            if KEEP_ENCODE_WORKSPACE:       # ~n*M*32 bytes (1.2 GB at MS MARCO size): never pinned in the product path
                _LAST_ENCODE["_ws"] = ws
        else:
            st = L.mevi_rq_encode_f32(hip.ptr(x), n, dim, hip.ptr(codebook), M, K, hip.ptr(codes), hip.stream_ptr())
            hip.check(st, "mevi_rq_encode_f32")
            _LAST_ENCODE.update(path="exact", n=n, dim=dim, M=M, K=K)
    return codes


def cluster_means(x, labels, K, old=None):
    """Per-cluster means of x f32[n, dim] under labels i32[n] (or [n, 1]): (centroids f32[K, dim], counts i32[K],
    sum of squared row norms as a float).  Deterministic; an empty cluster keeps `old[k]` (0 without `old`)."""
    hip.require_gpu()
    assert x.is_cuda and x.dtype == torch.float32 and labels.dtype == torch.int32 and x.is_contiguous()
    n, dim = x.shape
    labels = labels.contiguous().view(-1)
    cent = torch.empty((K, dim), dtype=torch.float32, device=x.device)
    counts = torch.empty(K, dtype=torch.int32, device=x.device)
    sumsq = torch.zeros(1, dtype=torch.float64, device=x.device)
    L = hip.lib()
    nbytes = L.mevi_cluster_means_workspace_bytes(n, dim, K)
    ws = torch.empty(max(nbytes, 256), dtype=torch.uint8, device=x.device)
    with hip.device_guard(x.device):
        st = L.mevi_cluster_means_f32(hip.ptr(x), n, dim, hip.ptr(labels), 1, K,
                                      hip.ptr(old.contiguous()) if old is not None else None, hip.ptr(cent),
                                      hip.ptr(counts), hip.ptr(sumsq), hip.ptr(ws), nbytes, hip.stream_ptr())
    hip.check(st, "mevi_cluster_means_f32")
    return cent, counts, float(sumsq.item())


def _kmeans_pp(xs, K, gen):
    """k-means++ seeding with 2 + log(K) greedy trials per centre (the scheme scikit-learn's `init='k-means++'` uses),
    on a sample that fits a [trials, m] distance matrix; torch ops on the device (seeding is a few thousand rows)."""
    m = xs.shape[0]
    trials = 2 + int(np.log(K))
    centers = torch.empty((K, xs.shape[1]), dtype=torch.float32, device=xs.device)
    first = int(torch.randint(m, (1,), generator=gen, device=xs.device).item())
    centers[0] = xs[first]
    d2 = ((xs - centers[0]) ** 2).sum(1)
    for c in range(1, K):
        # (every remaining point ON a chosen centre -- fewer distinct points than K, or a residual level that is already zero:
        # d2 sums to 0 and torch.multinomial aborts the device on an all-zero distribution -- then any point will do: uniform)
        tot = d2.sum()
        probs = torch.where(tot > 0, (d2 / tot.clamp_min(1e-30)).clamp_min(0), torch.full_like(d2, 1.0 / m))
        probs = torch.nan_to_num(probs, nan=1.0 / m, posinf=1.0 / m)
        cand = torch.multinomial(probs, trials, replacement=True, generator=gen)
        dc = torch.cdist(xs[cand], xs) ** 2                    # [trials, m]
        pot = torch.minimum(dc, d2[None]).sum(1)
        best = int(torch.argmin(pot).item())
        centers[c] = xs[cand[best]]
        d2 = torch.minimum(d2, dc[best])
    return centers


def kmeans(x, K, seed=0, max_iter=100, n_init=10, sample=32768, tol=1e-7):
    """Lloyd's k-means on the GPU: assignment = mevi_rq_encode_f32 against a one-level codebook (lowest index on ties,
    as everywhere), update = mevi_cluster_means_f32.  `n_init` k-means++ seedings are tried on a sample, the one with
    the lowest sample inertia after a few iterations starts the full run.  Returns (centres f32[K, dim],
    labels i32[n], inertia).  Deterministic for a given seed.

    Stands in for the scikit-learn (Mini-batch) KMeans the reference trains its codebooks with (MEVI/pq.py:550-567):
    a randomised algorithm, so parity is statistical -- tests hold the final quantisation error against
    scikit-learn's on the same data."""
    assert x.is_cuda and x.dtype == torch.float32 and x.dim() == 2
    x = x.contiguous()
    n = x.shape[0]
    gen = torch.Generator(device=x.device).manual_seed(int(seed))
    xs = x[torch.randperm(n, generator=gen, device=x.device)[:min(n, sample)]].contiguous()

    def lloyd(data, centers, iters):
        inertia = float("inf")
        labels = None
        for _ in range(iters):
            labels = rq_encode(data, centers[None].contiguous()).view(-1)
            new, counts, sumsq = cluster_means(data, labels, K, old=centers)
            # inertia of the NEW centres w.r.t. this assignment: sum ||x||^2 - sum_k n_k ||m_k||^2
            new_inertia = sumsq - float((counts.double() * (new.double() ** 2).sum(1)).sum().item())
            shift = float(((new - centers) ** 2).sum().item())
            centers = new
            if shift <= tol * max(float((centers ** 2).sum().item()), 1e-30) or abs(inertia - new_inertia) <= tol * abs(new_inertia):
                inertia = new_inertia
                break
            inertia = new_inertia
        return centers, labels, inertia

    best = None
    for _ in range(max(1, n_init)):
        c0 = _kmeans_pp(xs, K, gen)
        c1, _, inert = lloyd(xs, c0, 10)
        if best is None or inert < best[1]:
            best = (c1, inert)
    centers, labels, inertia = lloyd(x, best[0], max_iter)
    labels = rq_encode(x, centers[None].contiguous()).view(-1)   # labels of the returned centres
    return centers, labels, inertia


def train_rq_codebook(x, M, K, seed=0, **kw):
    """Residual quantisation codebook as the reference trains it (MEVI/pq.py:577-592): level i is k-means on the
    residual left by levels < i.  Returns (codebook f32[M, K, dim], codes i32[n, M]); `x` is not modified."""
    res = x.contiguous().clone()
    n, dim = res.shape
    src = torch.arange(n, dtype=torch.int64, device=x.device)
    book, codes = [], []
    L = hip.lib()
    for level in range(M):
        centers, labels, _ = kmeans(res, K, seed=seed + level, **kw)
        book.append(centers)
        codes.append(labels)
        if level != M - 1:
            with hip.device_guard(x.device):   # res -= centers[labels], row by row in place
                st = L.mevi_gather_sub_f32(hip.ptr(res), hip.ptr(src), hip.ptr(centers), hip.ptr(labels), n, dim,
                                           hip.ptr(res), hip.stream_ptr())
            hip.check(st, "mevi_gather_sub_f32")
    return torch.stack(book), torch.stack(codes, 1).contiguous()


class ClusterIndex:
    """CSR form of the reference's `pq_doc_cluster: dict[tuple -> list[int]]` and
    `pq_mapping: dict[int -> tuple]` (MEVI/main_models.py:3200-3220).

    key(code) = sum_j code[j] * K**(M-1-j);  doc ids of a cluster are ascending, exactly the
    order in which get_document_cluster appends them (pq.py:236-241)."""

    def __init__(self, M, K, keys, offsets, doc_ids, codes=None):
        self.M, self.K = M, K
        self.keys = keys          # i64[n_clusters] sorted distinct keys
        self.offsets = offsets    # i64[n_clusters + 1]
        self.doc_ids = doc_ids    # i64[N]
        self.codes = codes        # optional i32[N, M] (the mapping)

    @staticmethod
    def code_keys(codes, K):
        codes = np.asarray(codes, dtype=np.int64)
        M = codes.shape[-1]
        w = K ** np.arange(M - 1, -1, -1, dtype=np.int64)
        return (codes * w).sum(-1)

    @classmethod
    def from_codes(cls, codes, K, start=0):
        codes = np.asarray(codes)
        keys_all = cls.code_keys(codes, K)
        order = np.argsort(keys_all, kind="stable")
        sk = keys_all[order]
        keys, first = np.unique(sk, return_index=True)
        offsets = np.append(first, len(sk)).astype(np.int64)
        return cls(codes.shape[1], K, keys, offsets, (order + start).astype(np.int64), codes.astype(np.int32))

    @classmethod
    def from_topk_labels(cls, labels, K):
        """labels i32[N, C, M] (every document's top-C code paths, pq.get_topk_document_mapping): the multi-cluster
        index of gen_pq_doc_topk(return_cluster=True) (MEVI/main_models.py:3246-3256) -- a document is listed in each of
        its C clusters, lists in (document, path rank) order."""
        labels = np.asarray(labels)
        N, C, M = labels.shape
        idx = cls.from_codes(labels.reshape(N * C, M), K)
        return cls(M, K, idx.keys, idx.offsets, idx.doc_ids // C, None)

    @classmethod
    def from_dict(cls, cluster, M, K):
        """From the reference's pickled dict (rqclus*.pkl): vectorised, the dict has ~1 M keys / 8.8 M ids on MS MARCO."""
        from itertools import chain

        n = len(cluster)
        if n == 0:
            return cls(M, K, np.zeros(0, np.int64), np.zeros(1, np.int64), np.zeros(0, np.int64))
        keys = cls.code_keys(np.array(list(cluster.keys()), dtype=np.int64).reshape(n, M), K)
        sizes = np.fromiter(map(len, cluster.values()), dtype=np.int64, count=n)
        docs = np.fromiter(chain.from_iterable(cluster.values()), dtype=np.int64, count=int(sizes.sum()))
        order = np.argsort(keys, kind="stable")
        start = np.cumsum(sizes) - sizes                   # segment starts in dict order
        ssz = sizes[order]
        offsets = np.concatenate([[0], np.cumsum(ssz)]).astype(np.int64)
        # gather the segments in key order: element j of sorted segment i comes from start[order[i]] + j
        src = np.repeat(start[order] - offsets[:-1], ssz) + np.arange(len(docs), dtype=np.int64)
        return cls(M, K, keys[order], offsets, docs[src])

    def doc_codes(self, n_docs):
        """i32 [n_docs, M]: the code of every document (the inverse map `rqmapping*.pkl` stores as a dict of tuples);
        rows of documents that appear in no cluster are -1."""
        w = self.K ** np.arange(self.M - 1, -1, -1, dtype=np.int64)
        per_cluster = ((self.keys[:, None] // w[None, :]) % self.K).astype(np.int32)
        out = np.full((n_docs, self.M), -1, dtype=np.int32)
        out[self.doc_ids] = np.repeat(per_cluster, np.diff(self.offsets), axis=0)
        return out

    def lookup(self, code):
        """doc ids (ascending) of one cluster, empty array when the cluster owns no document."""
        key = int(self.code_keys(np.asarray(code), self.K))
        i = int(np.searchsorted(self.keys, key))
        if i >= len(self.keys) or self.keys[i] != key:
            return self.doc_ids[:0]
        return self.doc_ids[self.offsets[i]:self.offsets[i + 1]]

    def to_dicts(self):
        """(cluster dict, mapping dict) in the reference's pickle layout."""
        w = self.K ** np.arange(self.M - 1, -1, -1, dtype=np.int64)
        cluster = {}
        for i, key in enumerate(self.keys.tolist()):
            code = tuple(int((key // int(wj)) % self.K) for wj in w)
            cluster[code] = self.doc_ids[self.offsets[i]:self.offsets[i + 1]].tolist()
        mapping = {d: c for c, ds in cluster.items() for d in ds}
        return cluster, mapping


class ProductQuantization:
    def __init__(self, pq_type="rq", subvector_num=4, subvector_bits=5, dist_mode="l2", emb_size=768,
                 pq_init_method="kmeans", pq_update_method="none", device=None, **unused):
        if pq_type != "rq" or dist_mode != "l2":
            raise NotImplementedError("only pq_type='rq', dist_mode='l2' is on the MEVI eval path")
        self.pq_type, self.dist_mode = pq_type, dist_mode
        self.subvector_num, self.subvector_bits = subvector_num, subvector_bits
        self.subvector_cents = 2 ** subvector_bits
        self.emb_size = self.last_dim = emb_size
        self.pq_init_method = pq_init_method
        self.device = torch.device(device if device is not None else "cuda")
        self.codebook = torch.empty((subvector_num, self.subvector_cents, emb_size), dtype=torch.float32,
                                    device=self.device)

    def get_codebook(self):
        return self.codebook

    def fix(self):
        pass  # nothing trains here

    def load_codebook(self, tensor):
        t = torch.as_tensor(np.asarray(tensor) if not torch.is_tensor(tensor) else tensor.detach(), dtype=torch.float32)
        assert tuple(t.shape) == tuple(self.codebook.shape), (t.shape, self.codebook.shape)
        self.codebook.copy_(t.to(self.device))

    def initialize(self, index_file, doc_emb=None, rank=0, seed=0, pq_cluster_path=None, encode_batch_size=None):
        """Rank 0 loads `rqcodebook{M}_{bits}.pt` (torch.save of the f32[M,K,dim] parameter),
        then the codebook is broadcast (RCCL) when a process group exists."""
        import os

        import torch.distributed as dist

        if rank == 0:
            if index_file is not None and os.path.isfile(index_file):
                print("Intializing codebook with torch file...")
                self.load_codebook(torch.load(index_file, map_location="cpu"))
            elif doc_emb is not None:  # pq.py:402-419: no file -> train on the embeddings, then save next to it
                self.unsupervised_update_codebook_manually(doc_emb, seed, self.pq_init_method)
                if index_file is not None:
                    torch.save(self.codebook.detach().cpu(), index_file)
            else:
                raise FileNotFoundError(f"RQ codebook {index_file} not found and no embeddings to train one on")
        if dist.is_available() and dist.is_initialized():
            dist.broadcast(self.codebook, 0)

    def unsupervised_update_codebook_manually(self, doc_emb, seed, kmeans_method="kmeans"):
        """Train the residual codebook on `doc_emb` (MEVI/pq.py:550-598, the scikit-learn path): level i = k-means
        (k = 2**bits) on the residual of levels < i.  Runs on the GPU (`train_rq_codebook`); sets `last_preds`
        (i64 ndarray [n, M]) like the reference."""
        assert kmeans_method == "kmeans", kmeans_method
        print("Updating codebook using KMeans...")
        x = doc_emb if torch.is_tensor(doc_emb) else torch.from_numpy(np.ascontiguousarray(doc_emb, dtype=np.float32))
        x = x.to(self.device, torch.float32)
        book, codes = train_rq_codebook(x, self.subvector_num, self.subvector_cents, seed=int(seed))
        self.codebook.copy_(book)
        self.last_preds = codes.cpu().numpy().astype(np.int64)

    def forward(self, vecs, return_loss=False):
        """index i32[n, M]; the reference's (proba, index, loss) triple minus the training outputs."""
        return None, rq_encode(vecs.to(self.device, torch.float32), self.codebook), None

    def get_document_cluster(self, doc_embeddings, rank, nrank, batch_size=1 << 20, return_mapping=False,
                             as_index=False):
        """Encode this rank's contiguous slice (rows // nrank each, last rank takes the rest:
        pq.py:218-224) and group row ids by code tuple.  `batch_size` rows are resident on the
        GPU at a time (the reference's 128-row CPU batches compute the same thing)."""
        num_docs = doc_embeddings.shape[0]
        per = num_docs // nrank
        start = per * rank
        ending = num_docs if rank + 1 == nrank else start + per
        parts = []
        for b0 in range(start, ending, batch_size):
            b1 = min(b0 + batch_size, ending)
            chunk = doc_embeddings[b0:b1]
            chunk = chunk if torch.is_tensor(chunk) else torch.from_numpy(np.ascontiguousarray(chunk, dtype=np.float32))
            parts.append(rq_encode(chunk.to(self.device, torch.float32), self.codebook).cpu())
        codes = torch.cat(parts).numpy() if parts else np.zeros((0, self.subvector_num), np.int32)
        index = ClusterIndex.from_codes(codes, self.subvector_cents, start=start)
        print("Number of document clusters:", len(index.keys))
        if as_index:
            return index
        cluster = defaultdict(list)
        mapping = {}
        for i, c in enumerate(codes.tolist()):
            key = tuple(c)
            cluster[key].append(i + start)
            if return_mapping:
                mapping[i + start] = key
        return (dict(cluster), mapping) if return_mapping else dict(cluster)

    @torch.no_grad()
    def beam_search(self, doc_emb, num_return_sequences, num_beams=None, do_sample=False, return_proba=False):
        """Top-R code paths per row (pq.beam_search, MEVI/pq.py:613-713, rq_topk_score='prod'): per level
        softmax(-distance) times the running beam probability, top-R over beams x K.  Returns labels
        i32[bs, R, M] (and probabilities f32[bs, R])."""
        assert not do_sample and num_beams in (None, num_return_sequences)
        with hip.device_guard(self.device):          # stream_ptr() and the launches below refer to the codebook's GPU
            return self._beam_search(doc_emb, num_return_sequences, return_proba)

    def _beam_search(self, doc_emb, num_return_sequences, return_proba):
        L = hip.lib()
        R, M, K, dim = num_return_sequences, self.subvector_num, self.subvector_cents, self.emb_size
        x = doc_emb.to(self.device, torch.float32).contiguous()
        bs = x.shape[0]
        resid, nb = x, 1
        scores = torch.ones((bs, 1), dtype=torch.float32, device=self.device)
        labels = torch.zeros((bs, 1, 0), dtype=torch.int32, device=self.device)
        base = torch.arange(bs, device=self.device)[:, None]
        for j in range(M):
            nd = torch.empty((bs * nb, K), dtype=torch.float32, device=self.device)
            hip.check(L.mevi_rq_neg_dist_f32(hip.ptr(resid), bs * nb, dim, hip.ptr(self.codebook[j]), K, hip.ptr(nd),
                                             hip.stream_ptr()), "mevi_rq_neg_dist_f32")
            if R < nb * K:
                sc = torch.empty((bs, R), dtype=torch.float32, device=self.device)
                parent = torch.empty((bs, R), dtype=torch.int32, device=self.device)
                code = torch.empty((bs, R), dtype=torch.int32, device=self.device)
                hip.check(L.mevi_beam_step_f32(hip.ptr(nd), hip.ptr(scores), bs, nb, K, R, 2, hip.ptr(sc),
                                               hip.ptr(parent), hip.ptr(code), hip.stream_ptr()), "mevi_beam_step_f32")
                nb_new = R
            else:  # fewer candidates than beams: keep them all, in (beam, code) order
                from . import ops

                sc = ops.row_softmax(nd, log=False, scale=scores.reshape(-1)).reshape(bs, nb * K)
                parent = torch.arange(nb, device=self.device, dtype=torch.int32).repeat_interleave(K)[None].expand(bs, -1).contiguous()
                code = torch.arange(K, device=self.device, dtype=torch.int32).repeat(nb)[None].expand(bs, -1).contiguous()
                nb_new = nb * K
            labels = torch.cat([torch.gather(labels, 1, parent.long()[:, :, None].expand(-1, -1, labels.shape[2])),
                                code[:, :, None]], dim=2)
            if j != M - 1:
                src = (base * nb + parent.long()).reshape(-1).contiguous()
                nxt = torch.empty((bs * nb_new, dim), dtype=torch.float32, device=self.device)
                hip.check(L.mevi_gather_sub_f32(hip.ptr(resid), hip.ptr(src), hip.ptr(self.codebook[j]),
                                                hip.ptr(code.reshape(-1).contiguous()), bs * nb_new, dim, hip.ptr(nxt),
                                                hip.stream_ptr()), "mevi_gather_sub_f32")
                resid = nxt
            scores, nb = sc, nb_new
        return (labels, scores) if return_proba else labels

    def get_topk_document_mapping(self, doc_embeddings, rank, nrank, num_return_sequences, batch_size=1 << 16):
        """Top-R code paths of this rank's contiguous slice of the corpus (pq.py:715-741: rows // nrank each, the last
        rank takes the rest): i32 [rows, R, M] on the CPU."""
        n = doc_embeddings.shape[0]
        per = n // nrank
        start, end = per * rank, n if rank + 1 == nrank else per * (rank + 1)
        out = torch.empty((end - start, num_return_sequences, self.subvector_num), dtype=torch.int32)
        for s in range(start, end, batch_size):
            e = min(s + batch_size, end)
            x = doc_embeddings[s:e]
            x = x if torch.is_tensor(x) else torch.from_numpy(np.ascontiguousarray(x, dtype=np.float32))
            out[s - start:e - start] = self.beam_search(x, num_return_sequences).cpu()
        return out

    def get_reconstruct_vector(self, index, codebook=None):
        cb = self.codebook if codebook is None else codebook
        index = index.to(cb.device).long()
        out = torch.zeros(index.shape[:-1] + (cb.shape[-1],), dtype=torch.float32, device=cb.device)
        for j in range(cb.shape[0]):
            out = out + cb[j][index[..., j]]
        return out
