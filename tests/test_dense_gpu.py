"""GPU parity of the dense arm (mevi_ip_topk_f32 / mevi_topk_merge_f32 through the C ABI)
against oracle/mevi_oracle.c.  Bar: BIT-EXACT scores and identical ids -- both sides
compute the same sequential fmaf chain and order by (score desc, id asc)."""
import os

import numpy as np
import pytest
import torch

from mevi_amd import dense, hip
from oracle import dense as odense

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

pytestmark = pytest.mark.gpu


def _run(q, d, k, cuda, id_offset=0):
    s, i = dense.ip_topk(torch.from_numpy(q).to(cuda), torch.from_numpy(d).to(cuda), k, id_offset=id_offset)
    torch.cuda.synchronize()
    return s.cpu().numpy(), i.cpu().numpy()


def _run_indexed(q, d, k, cuda, id_offset=0):
    idx = dense.DenseIndex(torch.from_numpy(d).to(cuda))
    s, i = idx.search(torch.from_numpy(q).to(cuda), k, id_offset=id_offset)
    torch.cuda.synchronize()
    return s.cpu().numpy(), i.cpu().numpy()


def _check(q, d, k, cuda, id_offset=0):
    """Both the exact-f32 path and the indexed (f16 pre-filter + exact re-score) path, bit for bit."""
    s, i = _run(q, d, k, cuda, id_offset)
    es, ei = odense.ip_topk_exact(q, d, k, id_offset)
    np.testing.assert_array_equal(i, ei)
    np.testing.assert_array_equal(s.view(np.uint32), es.view(np.uint32))
    if d.shape[0] > 0:
        s2, i2 = _run_indexed(q, d, k, cuda, id_offset)
        np.testing.assert_array_equal(i2, ei)
        np.testing.assert_array_equal(s2.view(np.uint32), es.view(np.uint32))
    return s, i


def _stats():
    st = hip.IpTopkStats()
    hip.lib().mevi_ip_topk_get_stats(st)
    return st


@pytest.mark.parametrize("nq,nd,dim,k", [
    (200, 20000, 768, 100),    # C1-shaped slice
    (1, 5000, 768, 10),        # single query
    (130, 4099, 768, 1000),    # ragged tiles, k=1000 (the reference's topk)
    (37, 1000, 100, 7),        # dim not a multiple of the 32-wide K slab
    (5, 129, 4, 3),            # tiny dim
    (64, 9000, 768, 4096),     # largest supported k
])
def test_ip_topk_bit_exact(cuda, nq, nd, dim, k):
    rng = np.random.default_rng(nq * 7919 + nd)
    q = rng.standard_normal((nq, dim), dtype=np.float32)
    d = rng.standard_normal((nd, dim), dtype=np.float32)
    _check(q, d, k, cuda)


@pytest.mark.parametrize("nq,nd,dim,k", [
    (1, 70000, 768, 100), (2, 33333, 768, 1000), (8, 50001, 768, 10), (32, 20000, 768, 300), (7, 9000, 160, 50),
    (3, 12345, 200, 64), (31, 257, 768, 100), (4, 300000, 768, 1000),
])
def test_few_queries_take_the_streaming_kernel_with_the_same_bits(cuda, nq, nd, dim, k):
    """nq <= 32 (faiss_search.profile's batch sizes): ip_filter_h1_small_kernel -- stationary query tile, every wave streaming
    its own corpus rows -- must return the oracle's lists bit for bit, like the tile-stream kernel it replaces there
    (MEVI_IP_TOPK_NO_SMALL=1 forces that one: same result)."""
    import os

    rng = np.random.default_rng(nq * 31 + nd + dim)
    q = rng.standard_normal((nq, dim), dtype=np.float32)
    d = rng.standard_normal((nd, dim), dtype=np.float32)
    es, ei = odense.ip_topk_exact(q, d, k)
    s1, i1 = _run_indexed(q, d, k, cuda)
    st = _stats()
    os.environ["MEVI_IP_TOPK_NO_SMALL"] = "1"
    try:
        s2, i2 = _run_indexed(q, d, k, cuda)
    finally:
        del os.environ["MEVI_IP_TOPK_NO_SMALL"]
    for s_, i_ in ((s1, i1), (s2, i2)):
        np.testing.assert_array_equal(i_, ei)
        np.testing.assert_array_equal(s_.view(np.uint32), es.view(np.uint32))
    assert st.n_failed_queries == 0 or nd < 1000


@pytest.mark.parametrize("nq,nd,dim,k", [
    (33, 40000, 768, 100), (64, 70001, 768, 1000), (65, 30000, 768, 300), (100, 25000, 768, 1000), (128, 50000, 768, 100),
    (129, 20000, 768, 100), (255, 30000, 768, 1000), (48, 9000, 128, 50), (90, 12345, 192, 64),
    # persistent workgroups streaming SEVERAL tiles of a four-unit (dim <= 128) image: the 64-query tile's five-unit
    # look-ahead does not fit such a tile -- these take the 128-query tile
    (48, 300000, 128, 50), (64, 400000, 64, 100), (40, 350000, 192, 30),
])
def test_medium_batches_take_the_narrow_query_tiles_with_the_same_bits(cuda, nq, nd, dim, k):
    """33 .. 128 queries (faiss_search.profile's larger batches, MEVI/faiss_search.py:32-68): ip_filter_h16_kernel<2> / <4> --
    query tiles of 64 / 128 instead of 256 -- must return the oracle's lists bit for bit, like the 256-query tile it
    replaces there (MEVI_IP_FILTER_QT=256 pins that one: same result); 129 .. 255 queries keep the wide tile."""
    import os

    rng = np.random.default_rng(nq * 131 + nd + dim)
    q = rng.standard_normal((nq, dim), dtype=np.float32)
    d = rng.standard_normal((nd, dim), dtype=np.float32)
    es, ei = odense.ip_topk_exact(q, d, k)
    s1, i1 = _run_indexed(q, d, k, cuda)
    st = _stats()
    np.testing.assert_array_equal(i1, ei)
    np.testing.assert_array_equal(s1.view(np.uint32), es.view(np.uint32))
    assert st.n_failed_queries == 0
    code = ("import os, sys, numpy as np, torch; sys.path.insert(0, %r); from mevi_amd import dense; "
            "q = np.load(sys.argv[1]); d = np.load(sys.argv[2]); dev = torch.device('cuda:0'); "
            "s, i = dense.DenseIndex(torch.from_numpy(d).to(dev)).search(torch.from_numpy(q).to(dev), int(sys.argv[3])); "
            "np.save(sys.argv[4], i.cpu().numpy()); np.save(sys.argv[5], s.cpu().numpy())" % ROOT)
    import subprocess
    import sys
    import tempfile

    with tempfile.TemporaryDirectory() as tmp:      # the switch is read once per process: the pinned tile runs in a child
        np.save(tmp + "/q.npy", q), np.save(tmp + "/d.npy", d)
        r = subprocess.run([sys.executable, "-c", code, tmp + "/q.npy", tmp + "/d.npy", str(k), tmp + "/i.npy", tmp + "/s.npy"],
                           env=dict(os.environ, MEVI_IP_FILTER_QT="256"), capture_output=True, text=True, timeout=600)
        assert r.returncode == 0, r.stderr[-1500:]
        np.testing.assert_array_equal(np.load(tmp + "/i.npy"), ei)
        np.testing.assert_array_equal(np.load(tmp + "/s.npy").view(np.uint32), es.view(np.uint32))


def test_k_larger_than_corpus_pads_like_faiss(cuda):
    rng = np.random.default_rng(3)
    q = rng.standard_normal((9, 64), dtype=np.float32)
    d = rng.standard_normal((50, 64), dtype=np.float32)
    s, i = _check(q, d, 80, cuda)
    assert (i[:, 50:] == -1).all() and (s[:, 50:] == -np.finfo(np.float32).max).all()
    assert (np.sort(i[:, :50], axis=1) == np.arange(50)).all()


def test_empty_corpus_and_empty_queries(cuda):
    q = np.zeros((3, 8), np.float32)
    s, i = _run(q, np.zeros((0, 8), np.float32), 5, cuda)
    assert (i == -1).all() and (s == -np.finfo(np.float32).max).all()
    s, i = _run(np.zeros((0, 8), np.float32), np.ones((10, 8), np.float32), 5, cuda)
    assert s.shape == (0, 5) and i.shape == (0, 5)


def test_exact_ties_break_by_ascending_id(cuda):
    # MSMARCO holds duplicate passages -> exact score ties (SURVEY "Tie semantics")
    rng = np.random.default_rng(11)
    base = rng.integers(-3, 4, size=(40, 32)).astype(np.float32)
    d = np.concatenate([base] * 30, axis=0)           # every row appears 30 times
    q = rng.integers(-3, 4, size=(17, 32)).astype(np.float32)
    s, i = _check(q, d, 100, cuda)
    for r in range(q.shape[0]):
        for a in range(99):
            if s[r, a] == s[r, a + 1]:
                assert i[r, a] < i[r, a + 1]


def test_id_offset_and_shard_merge_equals_unsharded(cuda):
    rng = np.random.default_rng(5)
    q = rng.standard_normal((70, 128), dtype=np.float32)
    d = rng.standard_normal((10007, 128), dtype=np.float32)
    k = 200
    full_s, full_i = _check(q, d, k, cuda)
    for world in (2, 3, 8):
        ls, li = [], []
        for r in range(world):
            a, b = dense.shard_range(d.shape[0], r, world)
            s, i = _run(q, d[a:b], k, cuda, id_offset=a)
            ls.append(s)
            li.append(i)
        ms, mi = dense.topk_merge(torch.from_numpy(np.stack(ls)).to(cuda), torch.from_numpy(np.stack(li)).to(cuda), k)
        np.testing.assert_array_equal(mi.cpu().numpy(), full_i)
        np.testing.assert_array_equal(ms.cpu().numpy().view(np.uint32), full_s.view(np.uint32))
        os_, oi_ = odense.topk_merge(np.stack(ls), np.stack(li), k)
        np.testing.assert_array_equal(oi_, full_i)


def test_truncated_shard_lists_are_proven_or_redone(cuda):
    """Round 1 of the sharded search exchanges k_local < k entries per shard.  Even shards prove every
    query; a shard that owns most of a query's top-k (skewed corpus) must be detected as unproven."""
    rng = np.random.default_rng(17)
    nq, nd, dim, k, world = 40, 16000, 64, 200, 8
    q = rng.standard_normal((nq, dim), dtype=np.float32)
    d = rng.standard_normal((nd, dim), dtype=np.float32)
    q[1:] -= np.outer(q[1:] @ q[0], q[0]) / (q[0] @ q[0])   # only query 0 sees the skew
    d[:1500] += 0.8 * q[0]                      # query 0's neighbours all live in shard 0
    full_s, full_i = _run(q, d, k, cuda)
    kl = dense.truncated_list_len(k, world)
    assert kl == 25 + 5 * 5 + 8 and dense.truncated_list_len(k, 1) == k and dense.truncated_list_len(10, 8) == 10
    ls, li = [], []
    for r in range(world):
        a, b = dense.shard_range(nd, r, world)
        s, i = _run_indexed(q, d[a:b], kl, cuda, id_offset=a)
        ls.append(s)
        li.append(i)
    ms, mi, unproven = dense.merge_truncated(torch.from_numpy(np.stack(ls)).to(cuda), torch.from_numpy(np.stack(li)).to(cuda), k)
    unproven = unproven.cpu().numpy()
    assert unproven[0] and unproven.sum() <= 3
    ok = ~unproven
    np.testing.assert_array_equal(mi.cpu().numpy()[ok], full_i[ok])            # proven queries are exact
    np.testing.assert_array_equal(ms.cpu().numpy()[ok].view(np.uint32), full_s[ok].view(np.uint32))
    assert not np.array_equal(mi.cpu().numpy()[0], full_i[0])                  # and the flagged one really was wrong
    # redo of the flagged queries with full lists
    redo = np.nonzero(unproven)[0]
    ls, li = [], []
    for r in range(world):
        a, b = dense.shard_range(nd, r, world)
        s, i = _run_indexed(q[redo], d[a:b], k, cuda, id_offset=a)
        ls.append(s)
        li.append(i)
    rs, ri, still = dense.merge_truncated(torch.from_numpy(np.stack(ls)).to(cuda), torch.from_numpy(np.stack(li)).to(cuda), k)
    assert not still.any()
    np.testing.assert_array_equal(ri.cpu().numpy(), full_i[redo])


def test_centred_image_proves_queries_on_mean_shifted_embeddings(cuda):
    """Dense-retriever embeddings share a large common component (cosines ~0.97 between any two rows).  The
    error bound of the f16 pre-filter scales with ||d - mu||, not ||d||, so such a corpus is still proven
    without falling back; the answer is the exact one either way."""
    rng = np.random.default_rng(41)
    nq, nd, dim, k = 64, 60000, 768, 100
    common = np.full(dim, 0.2, dtype=np.float32)
    d = (0.05 * rng.standard_normal((nd, dim), dtype=np.float32) + common).astype(np.float32)
    q = (0.05 * rng.standard_normal((nq, dim), dtype=np.float32) + common).astype(np.float32)
    cos = (d[:100] @ d[100:200].T) / (np.linalg.norm(d[:100], axis=1)[:, None] * np.linalg.norm(d[100:200], axis=1)[None])
    assert cos.mean() > 0.9
    s, i = _run_indexed(q, d, k, cuda)
    st = _stats()
    es, ei = odense.ip_topk_exact(q, d, k)
    np.testing.assert_array_equal(i, ei)
    np.testing.assert_array_equal(s.view(np.uint32), es.view(np.uint32))
    assert st.n_failed_queries == 0, st.n_failed_queries
    assert st.max_err_ratio <= 0.5            # round 6: the bound is the MEASURED rounding (about half the worst case of rounds 1-5)


def test_heavy_tailed_magnitudes_keep_the_f16_bound_honest(cuda):
    """Rows whose elements span 2^-30 .. 2^10 (log-normal magnitudes, many below the f16 normal range after
    scaling, exact zeros, one dominant column): the observed f16 error must stay below the proven bound and the
    answer must be the exact one, proven or not."""
    rng = np.random.default_rng(53)
    nq, nd, dim, k = 48, 30000, 256, 50
    mag_d = np.exp(rng.normal(0.0, 4.0, size=(nd, dim))).astype(np.float32)
    d = (mag_d * rng.choice([-1.0, 1.0], size=(nd, dim))).astype(np.float32)
    d[:, 7] *= 64.0
    d[rng.random((nd, dim)) < 0.05] = 0.0
    mag_q = np.exp(rng.normal(0.0, 3.0, size=(nq, dim))).astype(np.float32)
    q = (mag_q * rng.choice([-1.0, 1.0], size=(nq, dim))).astype(np.float32)
    q[3] *= 1e-6                                   # one query far below the others (per-query scaling)
    s, i = _run_indexed(q, d, k, cuda)
    st = _stats()
    es, ei = odense.ip_topk_exact(q, d, k)
    np.testing.assert_array_equal(i, ei)
    np.testing.assert_array_equal(s.view(np.uint32), es.view(np.uint32))
    assert 0 < st.max_err_ratio <= 1.0, st.max_err_ratio      # a rigorous bound: few large coordinates may come close to it, never past it


def test_adversarial_row_order_takes_guaranteed_path(cuda):
    # rows sorted by ascending score for every query: each chunk floods the
    # candidate list -> overflow -> flagged queries are recomputed exactly.
    rng = np.random.default_rng(9)
    dim, nd, k = 32, 60000, 50
    direction = rng.standard_normal(dim).astype(np.float32)
    scale = np.linspace(-1.0, 1.0, nd, dtype=np.float32)[:, None]
    d = scale * direction[None, :] + 1e-3 * rng.standard_normal((nd, dim)).astype(np.float32)
    q = np.stack([direction * (1 + 0.1 * j) for j in range(6)]).astype(np.float32)
    q = np.concatenate([q, rng.standard_normal((10, dim)).astype(np.float32)])
    _check(q, d, k, cuda)
    st = _stats()
    assert st.n_failed_queries >= 6 and st.n_fallback_chunks > 0


def test_growth_override_changes_chunking_not_results(cuda):
    rng = np.random.default_rng(21)
    q = rng.standard_normal((33, 64), dtype=np.float32)
    d = rng.standard_normal((50000, 64), dtype=np.float32)
    L = hip.lib()
    try:
        L.mevi_ip_topk_set_growth(0.0)
        a = _check(q, d, 20, cuda)
        _run(q, d, 20, cuda)
        n0 = _stats().n_chunks
        L.mevi_ip_topk_set_growth(1.0)
        b = _check(q, d, 20, cuda)
        _run(q, d, 20, cuda)
        n1 = _stats().n_chunks
    finally:
        L.mevi_ip_topk_set_growth(0.0)
    assert n1 > n0
    np.testing.assert_array_equal(a[1], b[1])


def test_blas_restatement_agrees_to_rounding(cuda):
    # the faiss-style BLAS evaluation differs from the fmaf chain only by f32 rounding
    rng = np.random.default_rng(8)
    q = rng.standard_normal((50, 768), dtype=np.float32)
    d = rng.standard_normal((30000, 768), dtype=np.float32)
    s, i = _run(q, d, 100, cuda)
    bs, bi = odense.ip_topk_blas(q, d, 100)
    assert np.abs(s - bs).max() <= 1e-3          # |score| ~ 100, f32 eps * sum|a*b|
    assert (i == bi).mean() > 0.99               # only near-ties may swap


def test_indexed_prefilter_proves_most_queries_and_falls_back_for_the_rest(cuda):
    """Random data: the f16 bound proves (nearly) every query.  Near-duplicate scores beyond the
    margin (many rows within eps of the k-th score) cannot be proven -> exact fallback, same answer."""
    rng = np.random.default_rng(31)
    q = rng.standard_normal((300, 768), dtype=np.float32)
    d = rng.standard_normal((40000, 768), dtype=np.float32)
    s, i = _run_indexed(q, d, 100, cuda)
    es, ei = odense.ip_topk_exact(q, d, 100)
    np.testing.assert_array_equal(i, ei)
    st = _stats()
    assert st.n_failed_queries <= 3
    # the proof's error bound must dominate what is actually observed, with a wide margin
    assert 0 < st.max_err_ratio <= st.err_bound / 4, (st.max_err_ratio, st.err_bound)
    # 5000 copies of one row + noise far below the error bound: the top-50 cannot be proven from 256 survivors
    base = rng.standard_normal((1, 64), dtype=np.float32)
    d2 = np.concatenate([base + 1e-6 * rng.standard_normal((5000, 64)).astype(np.float32),
                         rng.standard_normal((3000, 64), dtype=np.float32)])
    q2 = np.concatenate([base * 2, rng.standard_normal((7, 64), dtype=np.float32)])
    s, i = _run_indexed(q2, d2, 50, cuda)
    es, ei = odense.ip_topk_exact(q2, d2, 50)
    np.testing.assert_array_equal(i, ei)
    np.testing.assert_array_equal(s.view(np.uint32), es.view(np.uint32))
    assert _stats().n_failed_queries >= 1


def test_second_pass_proves_a_cluster_of_duplicates(cuda):
    """250 near-identical rows on top of a query's list: the 192 survivors of the first pass cannot separate
    them from the rest, the second pass (384 survivors) holds them all and proves the list -- no exact-path run."""
    rng = np.random.default_rng(61)
    base = rng.standard_normal((1, 64), dtype=np.float32)
    d = np.concatenate([base + 1e-6 * rng.standard_normal((250, 64)).astype(np.float32),
                        rng.standard_normal((8000, 64), dtype=np.float32)])
    d = d[rng.permutation(len(d))]
    q = np.concatenate([base * 2, rng.standard_normal((7, 64), dtype=np.float32)])
    s, i = _run_indexed(q, d, 50, cuda)
    st = _stats()
    es, ei = odense.ip_topk_exact(q, d, 50)
    np.testing.assert_array_equal(i, ei)
    np.testing.assert_array_equal(s.view(np.uint32), es.view(np.uint32))
    assert st.n_second_pass_queries >= 1 and st.n_failed_queries == 0, (st.n_second_pass_queries, st.n_failed_queries)


def test_rejects_bad_arguments(cuda):
    q = torch.zeros((4, 6), device=cuda)
    d = torch.zeros((9, 6), device=cuda)
    with pytest.raises(hip.MeviHipError):
        dense.ip_topk(q, d, 3)                    # dim % 4 != 0
    with pytest.raises(hip.MeviHipError):
        dense.ip_topk(torch.zeros((4, 8), device=cuda), torch.zeros((9, 8), device=cuda), 5000)  # k > 4096


def test_search_beyond_the_kernel_list_length(cuda):
    """faiss_search.py --topk > 4096: dense.search falls back to materialised exact scores; bit-exact vs the oracle,
    ties by ascending id, -1 padding past the corpus."""
    rng = np.random.default_rng(12)
    q = rng.standard_normal((5, 32)).astype(np.float32)
    d = rng.standard_normal((6000, 32)).astype(np.float32)
    d[100:140] = d[7]                                   # ties
    for k in (5000, 6100):
        s, i = dense.search(q, d, 32, k)
        es, ei = odense.ip_topk_exact(q, d, k)
        assert np.array_equal(i, ei) and np.array_equal(s.view(np.int32), es.view(np.int32))


def test_bench_multi_rank_launch_rehearsal(cuda, tmp_path):
    """The driver's N > 1 launch line (torch.distributed.run, one process per rank) on a reduced corpus, two ranks sharing
    this GPU over MEVI_BENCH_BACKEND=gloo: shard generation, per-shard index, the two-round truncated exchange and the
    merge must return every planted neighbour, and rank 0 prints exactly one JSON line."""
    import json
    import os
    import socket
    import subprocess
    import sys

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr",
                        "127.0.0.1", "--master-port", str(port), os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "1",
                        "--warmup", "1", "--docs", "400000", "--queries", "700"],
                       capture_output=True, text=True, timeout=900, env=dict(os.environ, MEVI_BENCH_BACKEND="gloo", PYTHONPATH=root))
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1
    j = json.loads(lines[0])
    assert j["n_gpus"] == 2 and j["config"]["planted_top1_ok"] == 1.0 and j["scaling"] == "strong"
    assert j["config"]["docs"] == 400000 and "cpu_baseline" not in j


@pytest.mark.parametrize("nlist,nprobe,k", [(16, 1, 50), (16, 4, 100), (7, 7, 30), (40, 3, 1000)])
def test_ivf_flat_equals_the_oracle_given_its_centroids(cuda, nlist, nprobe, k):
    """--param IVF<n>,Flat: with the centroids the index trained, list membership (argmax inner product) and the search
    (exact top-k among the documents of the nprobe best lists) equal the CPU restatement bit for bit; probing every list
    is the Flat search."""
    from mevi_amd import ivf

    rng = np.random.default_rng(nlist * 31 + nprobe)
    centres = rng.standard_normal((12, 64)).astype(np.float32) * 2
    d = (centres[rng.integers(0, 12, size=6000)] + rng.standard_normal((6000, 64))).astype(np.float32)
    q = (centres[rng.integers(0, 12, size=90)] + rng.standard_normal((90, 64))).astype(np.float32)
    dt, qt = torch.from_numpy(d).to(cuda), torch.from_numpy(q).to(cuda)
    index = ivf.IVFFlatIndex(dt, nlist)
    s, i = index.search(qt, k, nprobe)
    es, ei, list_of = odense.ivf_flat_search(q, d, index.centroids.cpu().numpy(), k, nprobe)
    assert np.array_equal(index.list_of.cpu().numpy(), list_of)
    np.testing.assert_array_equal(i.cpu().numpy(), ei)
    np.testing.assert_array_equal(s.cpu().numpy().view(np.uint32), es.view(np.uint32))
    sizes = np.bincount(list_of, minlength=nlist)
    assert sizes.min() > 0 and sizes.max() < 0.6 * len(d)                    # the k-means produced usable lists
    if nprobe == nlist:
        fs, fi = odense.ip_topk_exact(q, d, k)
        np.testing.assert_array_equal(ei, fi)
    rec = ivf.recall_report(i, dense.ip_topk(qt, dt, k)[1])
    assert rec[1] > 0.5 and all(0.0 <= v <= 1.0 for v in rec.values())      # clustered data: the best list holds the top hit
    assert ivf.parse_factory("IVF100,Flat") == 100 and ivf.parse_factory("HNSW256") is None and ivf.parse_factory("IVF8,PQ4") is None


def test_search_api_serves_ivf_factory_strings(cuda, capsys):
    """faiss_search.search(..., param='IVF<n>,Flat'): prints what the reference prints (is_trained False before
    training), returns the IVF result and reports recall vs exact on stderr; MEVI_IVF_NPROBE = nlist gives Flat."""
    import os

    rng = np.random.default_rng(3)
    d = rng.standard_normal((3000, 32)).astype(np.float32)
    q = rng.standard_normal((40, 32)).astype(np.float32)
    s1, i1 = dense.search(q, d, 32, 20, "IVF10,Flat", device=cuda)
    out = capsys.readouterr()
    assert out.out == "Param IVF10,Flat trained: False.\n" and "recall vs exact search" in out.err and "nprobe=1" in out.err
    es, ei = odense.ip_topk_exact(q, d, 20)
    # approximate by construction (one list of ten probed), but never wrong about what it returns: every (id, score)
    # is that document's exact chain score, lists are ordered (score desc, id asc), and the top hit is found for some
    for r in range(len(q)):
        ok = i1[r] >= 0
        assert np.array_equal(s1[r][ok].view(np.uint32), odense.pair_dot(q[r], d[i1[r][ok]]).view(np.uint32))
        assert all((s1[r][j] > s1[r][j + 1]) or (s1[r][j] == s1[r][j + 1] and i1[r][j] < i1[r][j + 1])
                   for j in range(int(ok.sum()) - 1))
    assert 0.0 < (i1[:, 0] == ei[:, 0]).mean() <= 1.0
    old = os.environ.get("MEVI_IVF_NPROBE")
    os.environ["MEVI_IVF_NPROBE"] = "10"
    try:
        s2, i2 = dense.search(q, d, 32, 20, "IVF10,Flat", device=cuda)
    finally:
        if old is None:
            del os.environ["MEVI_IVF_NPROBE"]
        else:
            os.environ["MEVI_IVF_NPROBE"] = old
    np.testing.assert_array_equal(i2, ei)
    np.testing.assert_array_equal(s2.view(np.uint32), es.view(np.uint32))


def test_profile_is_the_reference_timing_hook(cuda, capsys):
    """faiss_search.profile (MEVI/faiss_search.py:32-68): seconds of train / add and the mean seconds of one search call at
    each batch size over the first ten batches; prints what the reference prints."""
    import faiss_search

    rng = np.random.default_rng(5)
    d = rng.standard_normal((4000, 64)).astype(np.float32)
    q = rng.standard_normal((37, 64)).astype(np.float32)
    t = faiss_search.profile(q, d, 64, 50, "Flat", bs=[1, 4, 8])
    out = capsys.readouterr().out
    assert out.splitlines()[0] == "Param Flat trained: True." and out.count("Profile batch size") == 3
    assert set(t) == {"train", "add", "search_bs1", "search_bs4", "search_bs8"}
    assert t["add"] > 0 and all(t[f"search_bs{b}"] > 0 for b in (1, 4, 8)) and t["train"] >= 0
    t2 = faiss_search.profile(q, d, 64, 20, "IVF8,Flat", bs=[2])
    assert capsys.readouterr().out.splitlines()[0] == "Param IVF8,Flat trained: False." and t2["train"] > 0 and t2["search_bs2"] > 0


@pytest.mark.parametrize("world,nq,kl,k", [(8, 50, 193, 1000), (2, 9, 623, 1000), (4, 33, 40, 100), (3, 5, 7, 7), (8, 4, 256, 300)])
def test_packed_merge_equals_merge_truncated(cuda, world, nq, kl, k):
    """mevi_topk_merge_packed_f32 (wire format in, merge stages only, proof fused) against dense.merge_truncated (torch unpack +
    full-sort merge kernel + torch proof) and the oracle's merge: lists with ties across shards, exhausted shards (-1
    padding), negative scores, fewer than k rows in total."""
    rng = np.random.default_rng(world * 1000 + kl)
    ls, li = [], []
    for r in range(world):
        n_real = kl if r % 3 else max(0, kl - 5)                     # some shards are exhausted before k_local
        sc = np.round(rng.standard_normal((nq, kl)) * 3, 1).astype(np.float32)      # coarse grid: ties across shards
        ids = np.stack([rng.choice(10_000, kl, replace=False) for _ in range(nq)]).astype(np.int64) * world + r
        order = np.lexsort((ids, -sc), axis=1)
        sc, ids = np.take_along_axis(sc, order, 1), np.take_along_axis(ids, order, 1)
        sc[:, n_real:], ids[:, n_real:] = -np.finfo(np.float32).max, -1
        ls.append(sc)
        li.append(ids)
    all_s, all_i = torch.from_numpy(np.stack(ls)).to(cuda), torch.from_numpy(np.stack(li)).to(cuda)
    want_s, want_i, want_u = dense.merge_truncated(all_s, all_i, k)
    packed = dense.pack_lists(all_s, all_i)
    us, ui = dense.unpack_lists(packed)
    assert torch.equal(ui, all_i) and torch.equal(us.view(torch.int32), all_s.view(torch.int32))
    got_s, got_i, got_u = dense.merge_packed(packed, k)
    assert torch.equal(got_i, want_i) and torch.equal(got_s.view(torch.int32), want_s.view(torch.int32)) and torch.equal(got_u, want_u)
    os_, oi_ = odense.topk_merge(np.stack(ls), np.stack(li), k)
    np.testing.assert_array_equal(got_i.cpu().numpy(), oi_)
    np.testing.assert_array_equal(got_s.cpu().numpy().view(np.uint32), os_.view(np.uint32))


@pytest.mark.parametrize("kind", ["iid", "clustered", "ance_scale", "duplicates"])
def test_realistic_corpus_distributions_bit_exact(cuda, kind):
    """the corpora of the bench's `data_sensitivity` leg (tools/synth.py: clustered rows in document runs, a
    T5-ANCE-like scale with a common component of norm ~11 and outlier dimensions, 1 % exact + 1 % near duplicates with
    queries planted ON duplicated rows) at a size the oracle finishes: exact-f32 path and f16-pre-filtered path both return
    the oracle's lists bit for bit, whatever the proof sends to the second pass or the fallback."""
    import sys

    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import synth

    nd, nq, k = 70_000, 64, 100
    docs, info = synth.corpus(kind, cuda, nd, 768, block=16384, n_clusters=40)
    q, planted = synth.corpus_queries(kind, docs, nq, info)
    d_h, q_h = docs.cpu().numpy(), q.cpu().numpy()
    _check(q_h, d_h, k, cuda)
    st = _stats()                                            # of the indexed search _check ran last
    es, ei = odense.ip_topk_exact(q_h, d_h, k)
    if kind == "duplicates":                                 # queries on duplicated rows: the copy pair ties, lower id first
        tied = (es[:, 0] == es[:, 1]).sum()
        assert tied >= nq // 4, tied
        assert (ei[es[:, 0] == es[:, 1], 0] < ei[es[:, 0] == es[:, 1], 1]).all()
    elif kind != "ance_scale":                               # (un-normalised rows of norm 8-16: the inner product prefers the long ones)
        assert (ei[:, :10] == planted.cpu().numpy()[:, None]).any(1).mean() > 0.9
    assert st.max_err_ratio <= 1.0, st.max_err_ratio         # the proof's bound dominates what was observed
    # k = 1000 over a cluster-sized neighbourhood (the whole top-k inside one cluster: dense scores at rank k)
    s2, i2 = _run_indexed(q_h[:16], d_h, 1000, cuda)
    es2, ei2 = odense.ip_topk_exact(q_h[:16], d_h, 1000)
    np.testing.assert_array_equal(i2, ei2)
    np.testing.assert_array_equal(s2.view(np.uint32), es2.view(np.uint32))


def test_candidate_counters_of_profiling_level_two(cuda):
    """mevi_ip_topk_set_profiling(2): the indexed search also reports how many keys its filter launches appended (what the
    data decides about its cost); level 0 / 1 leave the counters at zero and the results never change."""
    rng = np.random.default_rng(11)
    q = rng.standard_normal((40, 768), dtype=np.float32)
    d = rng.standard_normal((30000, 768), dtype=np.float32)
    s0, i0 = _run_indexed(q, d, 100, cuda)
    assert _stats().n_filter_candidates == 0
    L = hip.lib()
    L.mevi_ip_topk_set_profiling(2)
    try:
        s1, i1 = _run_indexed(q, d, 100, cuda)
        st = _stats()
    finally:
        L.mevi_ip_topk_set_profiling(0)
    np.testing.assert_array_equal(i0, i1)
    np.testing.assert_array_equal(s0.view(np.uint32), s1.view(np.uint32))
    kp = 100 + 48                                            # h1_kprime: every query keeps at least K' keys of the first launch
    assert st.n_filter_candidates >= 40 * kp and st.max_launch_candidates >= kp and st.n_list_overflows == 0, \
        (st.n_filter_candidates, st.max_launch_candidates, st.n_list_overflows)


def _child_search(q, d, k, env, tmp):
    """One indexed search in a child process (switches of the library are read once per process): ids, scores, filter launches."""
    import subprocess
    import sys

    code = ("import os, sys, ctypes, numpy as np, torch; sys.path.insert(0, %r); from mevi_amd import dense, hip; "
            "q = np.load(sys.argv[1]); d = np.load(sys.argv[2]); dev = torch.device('cuda:0'); "
            "s, i = dense.DenseIndex(torch.from_numpy(d).to(dev)).search(torch.from_numpy(q).to(dev), int(sys.argv[3])); torch.cuda.synchronize(); "
            "st = hip.IpTopkStats(); hip.lib().mevi_ip_topk_get_stats(st); "
            "np.save(sys.argv[4], i.cpu().numpy()); np.save(sys.argv[5], s.cpu().numpy()); print(int(st.n_chunks), int(st.n_failed_queries))" % ROOT)
    r = subprocess.run([sys.executable, "-c", code, tmp + "/q.npy", tmp + "/d.npy", str(k), tmp + "/i.npy", tmp + "/s.npy"],
                       env=dict(os.environ, **env), capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-1500:]
    chunks, failed = (int(x) for x in r.stdout.split()[-2:])
    return np.load(tmp + "/i.npy"), np.load(tmp + "/s.npy"), chunks, failed


@pytest.mark.parametrize("nq,nd,dim,k", [(40, 2_000_000, 64, 1000), (130, 1_500_000, 64, 1000), (300, 3_000_000, 64, 100),
                                          (5, 2_000_000, 768, 1000)])       # (<= 32 queries: the streaming kernel, 8192 slots)
def test_sampled_threshold_for_the_last_launch(cuda, nq, nd, dim, k, tmp_path):
    """Round 6: once enough rows are in, the pass estimates its threshold from the rank-r score of the rows seen (a sample under
    exchangeable row order) and takes ALL remaining rows in one launch (csrc/ip_topk.hip: rank_tau_kernel, sample_check_kernel).
    Fewer launches, the oracle's lists bit for bit, no query flagged on exchangeable data -- and the same lists as the geometric
    schedule to the end (MEVI_IP_SAMPLE_TAU=0)."""
    rng = np.random.default_rng(nq + nd + k)
    q = rng.standard_normal((nq, dim), dtype=np.float32)
    d = rng.standard_normal((nd, dim), dtype=np.float32)
    es, ei = odense.ip_topk_exact(q, d, k)
    tmp = str(tmp_path)
    np.save(tmp + "/q.npy", q), np.save(tmp + "/d.npy", d)
    i1, s1, chunks1, failed1 = _child_search(q, d, k, {}, tmp)
    i0, s0, chunks0, failed0 = _child_search(q, d, k, {"MEVI_IP_SAMPLE_TAU": "0"}, tmp)
    for i_, s_ in ((i1, s1), (i0, s0)):
        np.testing.assert_array_equal(i_, ei)
        np.testing.assert_array_equal(s_.view(np.uint32), es.view(np.uint32))
    assert failed1 == 0 and failed0 == 0
    assert chunks1 < chunks0, (chunks1, chunks0)
    # and through the exact-f32 path (the same schedule on exact keys)
    s, i = _run(q[:8], d, k, cuda)
    np.testing.assert_array_equal(i, ei[:8])
    np.testing.assert_array_equal(s.view(np.uint32), es[:8].view(np.uint32))


@pytest.mark.parametrize("order", ["best rows first", "best rows last", "ties at the threshold"])
def test_sampled_threshold_on_row_orders_that_break_its_premise(cuda, order):
    """The estimate is only a guess; exactness rests on sample_check_kernel (k rows must beat the threshold) and on the overflow
    flag.  Rows sorted so that the sample holds the BEST rows (threshold far too high: too few rows beat it -> flagged), the WORST
    (too low: the last launch floods the candidate area -> flagged), and a corpus of a few distinct rows (scores tie AT the
    threshold: rows equal to it are dropped, the check must notice): the oracle's lists every time, through the guaranteed path
    where needed."""
    rng = np.random.default_rng(17)
    nq, nd, dim, k = 40, 1_500_000, 64, 1000                  # (33 .. 1024 queries: the searches that use the sampled threshold)
    q = rng.standard_normal((nq, dim), dtype=np.float32)
    d = rng.standard_normal((nd, dim), dtype=np.float32)
    if order == "ties at the threshold":
        d = d[:40][rng.integers(0, 40, nd)].copy()             # 40 distinct rows: every score value is shared by ~37 k rows
    else:
        sc = d @ q[0]
        d = d[np.argsort(-sc if order == "best rows first" else sc, kind="stable")].copy()
    es, ei = odense.ip_topk_exact(q, d, k)
    for run in (_run_indexed, _run):
        s, i = run(q, d, k, cuda)
        st = _stats()
        np.testing.assert_array_equal(i, ei)
        np.testing.assert_array_equal(s.view(np.uint32), es.view(np.uint32))
        if order != "ties at the threshold" and run is _run_indexed:    # query 0 at least was flagged: second pass / guaranteed path
            assert st.n_failed_queries + st.n_second_pass_queries >= 1, order    # (the exact path's 3096-slot area at k = 1000 is too
                                                                                 # small for the sampled launch: geometric schedule)


# ---- round 6 (late): the 8-bit image of small searches --------------------------------------------------------------
def _small_index(d, cuda, monkeypatch):
    monkeypatch.setenv("MEVI_IP_I8_MIN_ROWS", "0")
    return dense.DenseIndex(torch.from_numpy(d).to(cuda)).prepare_small()


@pytest.mark.parametrize("nq,nd,dim,k", [
    (1, 70000, 768, 100), (2, 33333, 768, 1000), (8, 50001, 768, 10), (32, 20000, 768, 300), (7, 9000, 256, 50),
    (3, 12345, 200, 64), (31, 257, 768, 100), (4, 300000, 768, 1000), (32, 4097, 320, 1), (5, 1000, 896, 1300),
])
def test_small_searches_through_the_8_bit_image_return_the_oracles_lists(cuda, monkeypatch, nq, nd, dim, k):
    """<= 32 queries: int8 image + integer matrix cores + upper-bound keys + exact re-scoring + proof (mevi_ip_topk_indexed8_f32)
    must return the oracle's lists bit for bit -- padded dims (200 -> 256), ragged last blocks, k = 1, k past the corpus, a
    non-zero id offset; and the 8-bit pass must be what ran and proved them on Gaussian rows."""
    rng = np.random.default_rng(nq * 131 + nd + dim)
    q = rng.standard_normal((nq, dim), dtype=np.float32)
    d = rng.standard_normal((nd, dim), dtype=np.float32)
    d += 0.3                                                     # a common component (the image is centred)
    idx = _small_index(d, cuda, monkeypatch)
    for off in (0, 1_000_003):
        s, i = idx.search(torch.from_numpy(q).to(cuda), k, id_offset=off)
        st = _stats()
        es, ei = odense.ip_topk_exact(q, d, k, off)
        np.testing.assert_array_equal(i.cpu().numpy(), ei)
        np.testing.assert_array_equal(s.cpu().numpy().view(np.uint32), es.view(np.uint32))
        if 3 * k + 128 <= 4096:
            assert st.n_i8_queries == nq and st.max_err_ratio <= 1.0, (st.n_i8_queries, st.max_err_ratio)
            if k * 8 < nd:
                assert st.n_i8_unproven == 0, st.n_i8_unproven
        else:
            assert st.n_i8_queries == 0                          # 3 k + 64 survivors would not fit the proof kernel's sort
    assert idx.index8 is not None or 3 * k + 128 > 4096


@pytest.mark.parametrize("case", ["massive ties", "outlier rows and columns", "zero rows and a zero query", "best rows last", "non-finite row"])
def test_8_bit_image_on_inputs_that_break_its_premises(cuda, monkeypatch, case):
    """What the 8-bit bound cannot prove goes through the f16 image (and what that cannot prove through the exact path): the
    lists are the oracle's in every case, and the share of repeated searches switches the 8-bit pass off for that k."""
    rng = np.random.default_rng(len(case))
    nq, nd, dim, k = 9, 40000, 768, 20
    q = rng.standard_normal((nq, dim), dtype=np.float32)
    d = rng.standard_normal((nd, dim), dtype=np.float32)
    if case == "massive ties":
        d = d[:50][rng.integers(0, 50, nd)].copy()               # 50 distinct rows: every score 800 times
    elif case == "outlier rows and columns":
        d[:, 5] *= 300.0
        d[:, 700] *= 1e-4
        d[rng.integers(0, nd, 40)] *= 50.0                        # long rows (their step is coarse, their scores large)
        d[rng.integers(0, nd, 40), rng.integers(0, dim, 40)] = 900.0   # single huge coordinates
        q[2] *= 1e-8
    elif case == "zero rows and a zero query":
        d[::7] = 0.0
        q[4] = 0.0
    elif case == "best rows last":
        d = d[np.argsort(d @ q[0])].copy()
    elif case == "non-finite row":
        d[123, 45] = np.inf
    idx = _small_index(d, cuda, monkeypatch)
    tq = torch.from_numpy(q).to(cuda)
    unproven = 0
    for rep in range(4):
        s, i = idx.search(tq, k)
        st = _stats()
        unproven += int(st.n_i8_unproven > 0)
        if case == "non-finite row":                              # (the oracle's chain and the kernel's agree on inf / nan placement
            ref_s, ref_i = dense.ip_topk(tq, idx.docs, k)        #  only through the same exact path: compare with that)
            assert torch.equal(i, ref_i) and torch.equal(s.view(torch.int32), ref_s.view(torch.int32))
        else:
            es, ei = odense.ip_topk_exact(q, d, k)
            np.testing.assert_array_equal(i.cpu().numpy(), ei)
            np.testing.assert_array_equal(s.cpu().numpy().view(np.uint32), es.view(np.uint32))
        assert st.max_err_ratio <= 1.0
    if case == "non-finite row":                                  # the same through the f16 image alone (33 queries)
        q33 = torch.cat([tq] * 4)[:33].contiguous()
        s33, i33 = idx.search(q33, k)
        r33s, r33i = dense.ip_topk(q33, idx.docs, k)
        assert torch.equal(i33, r33i) and torch.equal(s33.view(torch.int32), r33s.view(torch.int32))
    if case in ("massive ties", "non-finite row"):
        assert unproven == 2 and idx._i8_open[k] > idx.I8_GIVE_UP   # two repeated searches, then the f16 image directly
        assert not idx.small_image_wanted(nq, k) and idx.small_image_wanted(nq, k + 1)
    # (the other cases may or may not be proven by the short survivor list: a zero query ties every row, a row with one huge
    #  coordinate doubles rho, best-rows-last breaks the sampled threshold's premise -- the lists above are what is asserted)


def test_8_bit_image_through_the_c_abi_and_its_switches(cuda, monkeypatch):
    """mevi_ip_topk_indexed8_f32 with index8 = NULL, with a shape outside the 8-bit pass (33 queries) and under MEVI_IP_I8=0 is
    mevi_ip_topk_indexed_f32; realistic corpora (tools/synth.py) return the exact path's lists through the 8-bit image."""
    import sys

    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import synth

    L = hip.lib()
    for kind in synth.CORPUS_KINDS:
        docs, info = synth.corpus(kind, cuda, 150_000, 768, block=16384, n_clusters=40)
        q, _ = synth.corpus_queries(kind, docs, 33, info)
        monkeypatch.setenv("MEVI_IP_I8_MIN_ROWS", "0")
        idx = dense.DenseIndex(docs).prepare_small()
        for nq, k in ((1, 10), (32, 100), (8, 1000), (33, 100)):
            es, ei = dense.ip_topk(q[:nq].contiguous(), docs, k)
            s, i = idx.search(q[:nq].contiguous(), k)
            st = _stats()
            assert torch.equal(i, ei) and torch.equal(s.view(torch.int32), es.view(torch.int32)), (kind, nq, k)
            assert st.n_i8_queries == (nq if nq <= 32 else 0)
            # the C entry point with no 8-bit image: the f16 search
            ws = torch.empty(L.mevi_ip_topk_indexed8_workspace_bytes(nq, 768, k), dtype=torch.uint8, device=cuda)
            s2, i2 = torch.empty_like(s), torch.empty_like(i)
            rc = L.mevi_ip_topk_indexed8_f32(hip.ptr(q), nq, hip.ptr(docs), hip.ptr(idx.index), None, docs.shape[0], 768, k, 0,
                                             hip.ptr(s2), hip.ptr(i2), hip.ptr(ws), ws.numel(), hip.stream_ptr())
            assert rc == 0 and _stats().n_i8_queries == 0
            assert torch.equal(i2, ei) and torch.equal(s2.view(torch.int32), es.view(torch.int32))
        del idx, docs
        torch.cuda.empty_cache()
    # an index that sees only a few small searches never builds the image; the ninth does
    idx = dense.DenseIndex(torch.randn(70000, 768, device=cuda))
    q4 = torch.randn(4, 768, device=cuda)
    es, ei = dense.ip_topk(q4, idx.docs, 10)
    for n in range(1, idx.SMALL_BUILD_AFTER + 2):
        s, i = idx.search(q4, 10)
        assert torch.equal(i, ei) and torch.equal(s.view(torch.int32), es.view(torch.int32))
        assert (idx.index8 is not None) == (n > idx.SMALL_BUILD_AFTER) and (_stats().n_i8_queries == 4) == (n > idx.SMALL_BUILD_AFTER)
    monkeypatch.setenv("MEVI_IP_I8", "0")
    assert not dense.DenseIndex(torch.randn(70000, 768, device=cuda)).small_image_wanted(4, 10)
