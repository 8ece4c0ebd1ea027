"""Dense arm: exact inner-product top-k on MI355X.

Mirrors the reference's `faiss_search.search(query, doc, dim, topk, param)`
(MEVI/faiss_search.py:13-21) and adds the row-sharded multi-GPU form required by
the north star: every rank searches its contiguous shard with
global ids, one RCCL all-gather of the per-shard top-k, then a device-side merge.
"""
import os

import numpy as np
import torch

from . import hip


def _as_device_f32(x, device):
    if isinstance(x, np.ndarray) and x.ndim == 2 and x.dtype == np.float32 and x.nbytes >= (64 << 20):
        from .io import upload_rows

        return upload_rows(x, device)           # the 27 GB corpus of faiss_search.py: pinned, chunked
    if isinstance(x, np.ndarray):
        x = torch.from_numpy(np.ascontiguousarray(x, dtype=np.float32))
    x = x.to(device=device, dtype=torch.float32)
    return x.contiguous()


def ip_topk(query, docs, k, id_offset=0):
    """Exact top-k of query @ docs.T on the current GPU.

    query f32[nq, dim], docs f32[nd, dim] (torch CUDA tensors, row-major).
    Returns (scores f32[nq,k] descending, ids i64[nq,k]); ids = id_offset + row,
    ties by ascending id, missing slots (-FLT_MAX, -1) like faiss.
    """
    hip.require_gpu()
    assert query.is_cuda and docs.is_cuda and query.dtype == torch.float32 and docs.dtype == torch.float32
    assert query.dim() == 2 and docs.dim() == 2 and query.shape[1] == docs.shape[1]
    query = query.contiguous()
    docs = docs.contiguous()
    nq, dim = query.shape
    nd = docs.shape[0]
    L = hip.lib()
    out_s = torch.empty((nq, k), dtype=torch.float32, device=query.device)
    out_i = torch.empty((nq, k), dtype=torch.int64, device=query.device)
    if nq == 0:
        return out_s, out_i
    ws_bytes = L.mevi_ip_topk_workspace_bytes(nq, dim, k)
    if ws_bytes == 0:
        raise hip.MeviHipError(f"ip_topk: unsupported shape nq={nq} dim={dim} k={k}")
    ws = torch.empty(ws_bytes, dtype=torch.uint8, device=query.device)
    with hip.device_guard(query.device):
        st = L.mevi_ip_topk_f32(hip.ptr(query), nq, hip.ptr(docs), nd, dim, k, id_offset,
                                hip.ptr(out_s), hip.ptr(out_i), hip.ptr(ws), ws_bytes, hip.stream_ptr())
    hip.check(st, "mevi_ip_topk_f32")
    return out_s, out_i


class DenseIndex:
    """A corpus shard prepared for searching -- the analogue of faiss `index.add(doc)`
    (MEVI/faiss_search.py:19): keeps the f32 rows and their centred, scaled f16 image (+ norms, mean, scales).
    `search` returns exactly what `ip_topk` returns (bit for bit), ~3x faster: candidates are
    selected with f16 MFMAs, re-scored with the exact f32 chain and proven complete per query.  Searches of up to 32 queries
    select through an int8 image instead (half the bytes of the HBM-bound small search; DESIGN 4.1c'): same lists."""

    def __init__(self, docs):
        hip.require_gpu()
        assert docs.is_cuda and docs.dtype == torch.float32 and docs.dim() == 2
        self.docs = docs.contiguous()
        nd, dim = self.docs.shape
        L = hip.lib()
        nbytes = L.mevi_ip_index_bytes(nd, dim)
        self.index = torch.empty(nbytes, dtype=torch.uint8, device=docs.device)
        with hip.device_guard(docs.device):
            st = L.mevi_ip_index_build_f32(hip.ptr(self.docs), nd, dim, hip.ptr(self.index), nbytes, hip.stream_ptr())
        hip.check(st, "mevi_ip_index_build_f32")
        self.index8 = None          # the 8-bit image for searches of <= 32 queries (prepare_small, or once they keep coming)
        self._i8_open = {}          # per k: running share of small searches the 8-bit pass left unproven (above I8_GIVE_UP: not used)
        self._small_calls = 0       # small searches seen so far (the image is built when they keep coming: SMALL_BUILD_AFTER)

    SMALL_QUERIES = 32              # searches up to this many queries go through the 8-bit image (csrc/ip_topk.hip: i8_eligible)
    SMALL_BUILD_AFTER = 8           # ... once the index has seen this many of them: building the image is two passes over the rows
                                    # (42 ms at 8.8 M x 768 = the saving of ~40 searches) -- an index made for one search never pays it;
                                    # prepare_small() builds it at once (faiss_search.profile does, inside its `add` time)
    I8_GIVE_UP = 0.4                # ... while the running share of searches it had to repeat through the f16 image stays below this
                                    # (a repeated search costs both passes: the 8-bit image pays as long as t8 + p t16 < t16, t8 ~ 0.55 t16)

    def small_image_wanted(self, nq, k):
        nd, dim = self.docs.shape
        dimp = max(128, (dim + 63) // 64 * 64)
        return (0 < nq <= self.SMALL_QUERIES and nd >= int(os.environ.get("MEVI_IP_I8_MIN_ROWS", "65536")) and 256 <= dimp <= 896 and 3 * k + 128 <= 4096
                and os.environ.get("MEVI_IP_I8", "1") != "0" and self._i8_open.get(k, 0) < self.I8_GIVE_UP)

    def prepare_small(self):
        """Build the 8-bit image now (otherwise the search after SMALL_BUILD_AFTER small ones does): nd * dim bytes + 8 per row,
        two passes over the rows (column scales, then the image)."""
        if self.index8 is None:
            nd, dim = self.docs.shape
            L = hip.lib()
            nbytes = L.mevi_ip_index8_bytes(nd, dim)
            index8 = torch.empty(nbytes, dtype=torch.uint8, device=self.docs.device)
            with hip.device_guard(self.docs.device):
                st = L.mevi_ip_index8_build_f32(hip.ptr(self.docs), hip.ptr(self.index), nd, dim, hip.ptr(index8), nbytes, hip.stream_ptr())
            hip.check(st, "mevi_ip_index8_build_f32")
            self.index8 = index8
        return self

    def _search_small(self, query, k, id_offset, out_s, out_i):
        nq, dim = query.shape
        nd = self.docs.shape[0]
        L = hip.lib()
        self.prepare_small()
        ws_bytes = L.mevi_ip_topk_indexed8_workspace_bytes(nq, dim, k)
        if ws_bytes == 0:
            raise hip.MeviHipError(f"ip_topk_indexed8: unsupported shape nq={nq} dim={dim} k={k}")
        ws = torch.empty(ws_bytes, dtype=torch.uint8, device=query.device)
        with hip.device_guard(query.device):
            st = L.mevi_ip_topk_indexed8_f32(hip.ptr(query), nq, hip.ptr(self.docs), hip.ptr(self.index), hip.ptr(self.index8), nd, dim, k,
                                             id_offset, hip.ptr(out_s), hip.ptr(out_i), hip.ptr(ws), ws_bytes, hip.stream_ptr())
        hip.check(st, "mevi_ip_topk_indexed8_f32")
        stats = hip.IpTopkStats()
        L.mevi_ip_topk_get_stats(stats)
        if stats.n_i8_queries:      # a corpus whose scores are too dense for the 8-bit bound pays both searches: stop trying
            self._i8_open[k] = 0.75 * self._i8_open.get(k, 0.0) + (0.25 if stats.n_i8_unproven else 0.0)
        return out_s, out_i

    def search(self, query, k, id_offset=0):
        assert query.is_cuda and query.dtype == torch.float32 and query.dim() == 2 and query.shape[1] == self.docs.shape[1]
        query = query.contiguous()
        nq, dim = query.shape
        nd = self.docs.shape[0]
        L = hip.lib()
        out_s = torch.empty((nq, k), dtype=torch.float32, device=query.device)
        out_i = torch.empty((nq, k), dtype=torch.int64, device=query.device)
        if nq == 0:
            return out_s, out_i
        if self.small_image_wanted(nq, k):
            self._small_calls += 1
            if self.index8 is not None or self._small_calls > self.SMALL_BUILD_AFTER:
                return self._search_small(query, k, id_offset, out_s, out_i)
        ws_bytes = L.mevi_ip_topk_indexed_workspace_bytes(nq, dim, k)
        if ws_bytes == 0:
            raise hip.MeviHipError(f"ip_topk_indexed: unsupported shape nq={nq} dim={dim} k={k}")
        ws = torch.empty(ws_bytes, dtype=torch.uint8, device=query.device)
        with hip.device_guard(query.device):
            st = L.mevi_ip_topk_indexed_f32(hip.ptr(query), nq, hip.ptr(self.docs), hip.ptr(self.index), nd, dim, k,
                                            id_offset, hip.ptr(out_s), hip.ptr(out_i), hip.ptr(ws), ws_bytes,
                                            hip.stream_ptr())
        hip.check(st, "mevi_ip_topk_indexed_f32")
        return out_s, out_i


def topk_merge(scores, ids, k_out):
    """Merge per-shard lists: scores f32[nlists,nq,k_in], ids i64[nlists,nq,k_in] -> [nq,k_out]."""
    hip.require_gpu()
    assert scores.is_cuda and ids.is_cuda and scores.shape == ids.shape and scores.dim() == 3
    scores = scores.contiguous()
    ids = ids.contiguous()
    nlists, nq, k_in = scores.shape
    L = hip.lib()
    # hierarchical merge if the flat list does not fit one LDS sort
    while nlists * k_in > 16384 and nlists > 1:
        half = (nlists + 1) // 2
        parts_s, parts_i = [], []
        for a in range(0, nlists, 2):
            s, i = topk_merge(scores[a:a + 2], ids[a:a + 2], min(k_out, 2 * k_in))
            parts_s.append(s)
            parts_i.append(i)
        k_in = parts_s[0].shape[1]
        scores, ids, nlists = torch.stack(parts_s), torch.stack(parts_i), half
    out_s = torch.empty((nq, k_out), dtype=torch.float32, device=scores.device)
    out_i = torch.empty((nq, k_out), dtype=torch.int64, device=scores.device)
    with hip.device_guard(scores.device):
        st = L.mevi_topk_merge_f32(hip.ptr(scores), hip.ptr(ids), nlists, nq, k_in, k_out,
                                   hip.ptr(out_s), hip.ptr(out_i), None, 0, hip.stream_ptr())
    hip.check(st, "mevi_topk_merge_f32")
    return out_s, out_i


def shard_range(n_rows, rank, world_size):
    """Contiguous row shard [start, end) of rank: ceil(n/world) rows each."""
    per = (n_rows + world_size - 1) // world_size
    start = min(rank * per, n_rows)
    return start, min(start + per, n_rows)


def truncated_list_len(k, world):
    """Entries each shard contributes in the first round of a sharded search.  With rows spread evenly a
    shard holds Binomial(k, 1/world) of the global top-k: mean k/world, deviation < sqrt(k/world); five
    deviations + 8 of head-room (W = 8, k = 1000: 193 entries, 6.5 actual deviations: < 1e-9 per query and
    shard) make a second round rare, and the proof in merge_truncated catches the rest.  The per-rank costs
    that do not shrink with the shard (re-scoring K' = k_local + slack survivors, compaction, the all-gather
    payload, the merge) are all proportional to this length."""
    if world <= 1:
        return k
    share = (k + world - 1) // world
    sig = float(os.environ.get("MEVI_SHARD_HEADROOM_SIGMAS", "5"))
    return min(k, share + int(np.ceil(sig * np.ceil(np.sqrt(share)))) + 8)


def merge_truncated(all_s, all_i, k, merge=None):
    """Merge per-shard lists that may have been TRUNCATED to k_local < k entries and decide, per query,
    whether the merged top-k is provably the un-sharded answer.

    all_s f32[world, nq, k_local], all_i i64[world, nq, k_local] (id -1 = padding: that shard is exhausted).
    A shard's unseen rows all rank after its last returned entry, so the merged list is exact unless some
    shard's LAST entry made it into the merged top-k (then deeper rows of that shard might belong too).
    Returns (scores [nq, k], ids [nq, k], unproven bool[nq])."""
    merge = merge or topk_merge
    world, nq, kl = all_s.shape
    ms, mi = merge(all_s, all_i, k)
    last_s, last_i = all_s[:, :, kl - 1], all_i[:, :, kl - 1]          # [world, nq]
    kth_s, kth_i = ms[:, k - 1], mi[:, k - 1]                           # [nq]
    # kth_i == -1: fewer than k rows exist in total -> every returned row is already in the list
    in_topk = (last_i >= 0) & (kth_i[None] >= 0) & (
        (last_s > kth_s[None]) | ((last_s == kth_s[None]) & (last_i <= kth_i[None])))
    truncated = kl < k
    unproven = in_topk.any(0) if truncated else torch.zeros(nq, dtype=torch.bool, device=ms.device)
    return ms, mi, unproven


def pack_lists(scores, ids):
    """(scores f32 [..], ids i64 [..]) -> i64 [..]: score bits << 32 | id as u32 -- the 8-byte wire format of the exchange."""
    scores, ids = scores.contiguous(), ids.contiguous()
    if not scores.is_cuda:       # the gloo CPU tests inject their own search / merge; same format
        return (scores.view(torch.int32).to(torch.int64) << 32) | (ids & 0xFFFFFFFF)
    out = torch.empty(scores.shape, dtype=torch.int64, device=scores.device)
    with hip.device_guard(scores.device):
        hip.check(hip.lib().mevi_pack_lists_i64(hip.ptr(scores), hip.ptr(ids), scores.numel(), hip.ptr(out), hip.stream_ptr()),
                  "mevi_pack_lists_i64")
    return out


def unpack_lists(packed):
    all_s = (packed >> 32).to(torch.int32).view(torch.float32)
    all_i = packed & 0xFFFFFFFF
    return all_s, torch.where(all_i == 0xFFFFFFFF, torch.full_like(all_i, -1), all_i)


def merge_packed(gathered, k):
    """merge_truncated on the all-gathered wire format, in ONE kernel (mevi_topk_merge_packed_f32): gathered i64
    [world, nq, k_local], every list sorted as the local searches return them.  Returns (scores, ids, unproven bool[nq])."""
    world, nq, kl = gathered.shape
    out_s = torch.empty((nq, k), dtype=torch.float32, device=gathered.device)
    out_i = torch.empty((nq, k), dtype=torch.int64, device=gathered.device)
    unproven = torch.zeros(nq, dtype=torch.uint8, device=gathered.device)
    with hip.device_guard(gathered.device):
        st = hip.lib().mevi_topk_merge_packed_f32(hip.ptr(gathered), world, nq, kl, k, 1 if kl < k else 0, hip.ptr(out_s),
                                                  hip.ptr(out_i), hip.ptr(unproven), hip.stream_ptr())
    hip.check(st, "mevi_topk_merge_packed_f32")
    return out_s, out_i, unproven.bool()


def _packed_merge_fits(world, kl, k):
    lp = 32
    while lp < kl:
        lp <<= 1
    return world <= 64 and world * lp <= 16384 and k <= 16384


class SearchTrace:
    """What bench.py asks a search to record (never set by the CLIs): the kernel counters of EVERY local search of a call
    (mevi_ip_topk_get_stats restarts per library call, so a second round would otherwise overwrite the first) and,
    with `timed`, the wall-clock of each phase of the sharded search -- every phase is then followed by a device
    synchronisation, so a traced call is for diagnosis, not for the timed region."""

    FIELDS = ("n_chunks", "n_failed_queries", "n_fallback_chunks", "filter_ms", "compact_ms", "filter_flops",
              "n_second_pass_queries")

    def __init__(self, timed=False):
        self.timed = timed
        self.stats = {f: 0.0 for f in self.FIELDS}
        self.ms = {"local_search": 0.0, "all_gather": 0.0, "merge": 0.0}
        self.second_round_queries = 0
        self.rounds = 0

    def add_stats(self):
        st = hip.IpTopkStats()
        hip.lib().mevi_ip_topk_get_stats(st)
        for f in self.FIELDS:
            self.stats[f] += getattr(st, f)

    def lap(self, name, t0):
        import time

        if not self.timed:
            return t0
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        self.ms[name] += (t1 - t0) * 1e3
        return t1


def sharded_ip_topk(query, local_docs, k, id_offset, group=None, local_search=None, merge=None, trace=None):
    """Row-sharded search: local top lists with global ids, all-gather, merge.

    Every rank passes the full (replicated) query matrix and its own shard; every rank returns the
    identical merged (scores, ids), equal to the un-sharded search whatever the shard count.
    Round 1 exchanges only k_local = truncated_list_len(k, world) entries per shard -- k/world + 5 sqrt(k/world) + 8
    (193 at k = 1000, world = 8; MEVI_SHARD_HEADROOM_SIGMAS overrides the 5) -- so per-rank re-scoring, compaction
    and the all-gather shrink with the world size; queries whose merged list cannot be proven complete (a shard's
    last entry reached the global top-k) are repeated with full k-entry lists -- every rank takes the same decision
    from the same gathered data.  The binomial head-room assumes the top-k rows are spread evenly over the shards; a
    corpus stored cluster by cluster concentrates them, the second round then runs for most queries (correct, but a
    second search + all-gather): `trace.second_round_queries` reports it, bench.py prints it for N > 1, and
    tools/bench_shard_sim.py --layout sorted measures that case.
    `local_search` / `merge` default to the HIP kernels; they are injection points for the CPU (gloo)
    test of the collective plumbing and are never set by product code.
    """
    import time

    import torch.distributed as dist

    def local(q, kk):
        if local_search is not None:
            return local_search(q, local_docs, kk, id_offset=id_offset)
        if isinstance(local_docs, DenseIndex):
            out = local_docs.search(q, kk, id_offset=id_offset)
        else:
            out = ip_topk(q, local_docs, kk, id_offset=id_offset)
        if trace is not None:
            trace.add_stats()
        return out

    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size(group) == 1:
        return local(query, k)
    world = dist.get_world_size(group)

    def exchange(q, kk):
        # one collective per round: (score bits << 32 | id as u32) in an i64 per entry -- 8 bytes instead of the 12
        # of separate f32 + i64 tensors, and half the launches.  Ids fit 32 bits (the C ABI enforces it); the
        # padding id -1 travels as 0xFFFFFFFF.
        t0 = time.perf_counter() if trace is not None else 0.0
        s, i = local(q, kk)
        if trace is not None:
            t0 = trace.lap("local_search", t0)
            trace.rounds += 1
        n = s.shape[0]
        packed = pack_lists(s, i)
        if packed.is_cuda and dist.get_backend(group) == "gloo":     # rehearsal backend: collectives through host memory
            host = torch.empty((world * n, kk), dtype=torch.int64)
            dist.all_gather_into_tensor(host, packed.cpu().contiguous(), group=group)
            gathered = host.to(s.device)
        else:
            gathered = torch.empty((world * n, kk), dtype=torch.int64, device=s.device)   # rank-major concatenation
            dist.all_gather_into_tensor(gathered, packed.contiguous(), group=group)
        if trace is not None:
            trace.lap("all_gather", t0)
        return gathered.view(world, n, kk)

    def merged(q, kk):
        gathered = exchange(q, kk)
        t0 = time.perf_counter() if trace is not None else 0.0
        if merge is None and gathered.is_cuda and _packed_merge_fits(world, kk, k):
            out = merge_packed(gathered, k)                          # unpack + merge + proof in one kernel
        else:
            all_s, all_i = unpack_lists(gathered)
            out = merge_truncated(all_s, all_i, k, merge=merge)
        if trace is not None:
            trace.lap("merge", t0)
        return out

    kl = truncated_list_len(k, world)
    ms, mi, unproven = merged(query, kl)
    redo = torch.nonzero(unproven).flatten()
    if trace is not None:
        trace.second_round_queries += int(redo.numel())
    if redo.numel() > 0:                       # identical on every rank
        rs, ri, _ = merged(query[redo].contiguous(), k)
        ms[redo], mi[redo] = rs, ri
    return ms, mi


def is_trained_before_train(param):
    """What `faiss.index_factory(dim, param).is_trained` reports BEFORE `index.train` (the value the reference prints,
    MEVI/faiss_search.py:16): False for every component that learns from data -- IVF coarse quantisers, product /
    scalar / residual quantisers, OPQ / PCA / ITQ transforms --, True for Flat, HNSW over Flat storage, LSH, IDMap."""
    learns = ("IVF", "IMI", "PQ", "OPQ", "PCA", "ITQ", "SQ", "RQ", "LSQ", "PRQ", "PLSQ")
    for comp in str(param).split(","):
        c = comp.strip()
        if c.startswith("HNSW") and "_" in c:                  # HNSW32_PQ8 ...: storage given after the underscore
            c = c.split("_", 1)[1]
        if c.startswith(learns):
            return False
    return True


def search(query, doc, dim, topk, param="Flat", device=None):
    """Drop-in for faiss_search.search (MEVI/faiss_search.py:13-21).

    `param` is the faiss factory string the reference forwards: "Flat" -> exact search (result parity defined);
    "IVF<n>,Flat" (the script's default) -> IVF-Flat with faiss's structure and defaults (mevi_amd/ivf.py: inner-
    product k-means, nprobe = MEVI_IVF_NPROBE or 1; recall vs the exact search is reported on stderr); every other
    index type (HNSW*, PQ*, ...) is served by exact search (recall >= the approximate index it names).
    Returns numpy (dists f32[nq,topk], indices i64[nq,topk]).
    """
    hip.require_gpu()
    device = torch.device(device if device is not None else "cuda")
    print(f"Param {param} trained: {is_trained_before_train(param)}.")  # index.is_trained before index.train
    q = _as_device_f32(np.asarray(query).reshape(-1, dim) if isinstance(query, np.ndarray) else query, device)
    d = _as_device_f32(np.asarray(doc).reshape(-1, dim) if isinstance(doc, np.ndarray) else doc, device)
    from . import ivf

    nlist = ivf.parse_factory(param)
    if nlist is not None and topk <= MAX_K and 0 < nlist <= d.shape[0]:
        s, i = ivf.search(q, d, topk, nlist)
    elif topk > MAX_K:
        s, i = _search_large_k(q, d, topk)
    else:
        s, i = DenseIndex(d).search(q, topk)  # index.add(doc); index.search(query, topk)
    return s.cpu().numpy(), i.cpu().numpy()


def profile(query, doc, dim, topk, param, bs=[1, 2, 4, 8], device=None):
    """Drop-in for faiss_search.profile (MEVI/faiss_search.py:32-68), the reference authors' timing hook of the dense arm:
    seconds of `train`, `add` (= upload + index build here) and the MEAN seconds of one `index.search` call at each batch size
    over the first ten batches of `query` (a ragged last batch is filled with random other queries, as the reference does).
    Every search takes the batch from host memory and returns numpy arrays, like faiss's API.  Returns the `all_time` dict."""
    import time

    from . import ivf

    hip.require_gpu()
    device = torch.device(device if device is not None else "cuda")
    all_time = {"train": 0, "add": 0}
    for b in bs:
        all_time[f"search_bs{b}"] = 0
    print(f"Param {param} trained: {is_trained_before_train(param)}.")
    query = np.asarray(query, dtype=np.float32).reshape(-1, dim)
    nlist = ivf.parse_factory(param)
    with hip.device_guard(device):
        t = time.time()
        d = _as_device_f32(np.asarray(doc).reshape(-1, dim) if isinstance(doc, np.ndarray) else doc, device)
        torch.cuda.synchronize()
        upload = time.time() - t
        use_ivf = nlist is not None and 0 < nlist <= d.shape[0] and topk <= MAX_K
        t = time.time()
        cent = ivf.train_centroids(d, nlist) if use_ivf else None      # index.train(doc)
        torch.cuda.synchronize()
        all_time["train"] += time.time() - t
        t = time.time()
        index = ivf.IVFFlatIndex(d, nlist, centroids=cent) if use_ivf else (DenseIndex(d) if topk <= MAX_K else None)
        if isinstance(index, DenseIndex) and any(index.small_image_wanted(b, topk) for b in bs):
            index.prepare_small()                               # the 8-bit image of the small batches is part of index.add
        torch.cuda.synchronize()
        all_time["add"] += upload + time.time() - t

        def one(bq):
            q = torch.from_numpy(np.ascontiguousarray(bq)).to(device)
            if use_ivf:
                s, i = index.search(q, topk, int(os.environ.get("MEVI_IVF_NPROBE", "1")))
            elif index is None:
                s, i = _search_large_k(q, d, topk)
            else:
                s, i = index.search(q, topk)
            return s.cpu().numpy(), i.cpu().numpy()

        nquery = len(query)
        for b in bs:
            print(f"Profile batch size {b}...")
            batched = []
            for start in range(0, nquery, b):
                cur = query[start:start + b]
                if len(cur) < b:
                    rest = np.random.choice(np.arange(nquery), size=b - len(cur), replace=False)
                    cur = np.concatenate((cur, query[rest]), axis=0)
                batched.append(cur)
                if len(batched) >= 10:
                    break
            if not batched:
                continue
            one(batched[0])                                     # first-call set-up (module load, LDS opt-in) is not a search cost
            torch.cuda.synchronize()
            t = time.time()
            for bq in batched:
                one(bq)
            all_time[f"search_bs{b}"] += (time.time() - t) / len(batched)
    return all_time


MAX_K = 4096     # list length the threshold-filter kernels keep per query (include/mevi_hip.h)


def _search_large_k(q, d, k, rows_budget=1 << 28):
    """faiss accepts any k; the filter kernels keep at most MAX_K entries per query.  Beyond that (not a configuration of
    any reference script) the scores of a few queries at a time are materialised with the f32 GEMM -- the same
    sequential fmaf chains -- and ordered by (score desc, id asc) with two stable device sorts; -1 / -FLT_MAX padding
    when k exceeds the corpus, as faiss."""
    from . import ops

    nq, nd = q.shape[0], d.shape[0]
    out_s = torch.full((nq, k), torch.finfo(torch.float32).min, dtype=torch.float32, device=q.device)
    out_i = torch.full((nq, k), -1, dtype=torch.int64, device=q.device)
    kk = min(k, nd)
    step = max(1, rows_budget // max(nd, 1))
    for a in range(0, nq, step):
        sc = ops.linear(q[a:a + step].contiguous(), d)                        # [b, nd], exact chains
        order = torch.argsort(sc, dim=1, descending=True, stable=True)         # ties keep ascending id
        out_i[a:a + step, :kk] = order[:, :kk]
        out_s[a:a + step, :kk] = torch.gather(sc, 1, order[:, :kk])
    return out_s, out_i
