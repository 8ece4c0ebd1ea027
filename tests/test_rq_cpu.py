"""CPU: the RQ oracle against the reference's goldens (tools/capture_goldens.py g4) and the
host-side cluster index."""
import glob
import os

import numpy as np
import pytest

from oracle import rq as orq

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


@pytest.mark.parametrize("path", sorted(glob.glob(os.path.join(GOLD, "g4_rq_*.npz"))))
def test_oracle_pinned_by_reference(path):
    g = np.load(path)
    codes, neg = orq.rq_encode(g["X"], g["C"], return_neg_dist=True)
    assert np.array_equal(codes, g["codes"])                      # pq.get_document_cluster / forward_rq index
    ref = g["forward_proba"]                                       # forward_rq proba = -dist
    assert np.abs(neg - ref).max() <= 4e-6 * np.abs(ref).max()     # summation order only
    assert np.array_equal(orq.reconstruct(g["codes"][:32], g["C"]), g["reconstruct32"])
    cluster, mapping = orq.cluster_dict(codes)
    keys = [tuple(k) for k in g["cluster_keys"].tolist()]
    assert sorted(cluster) == keys and [d for k in keys for d in cluster[k]] == g["cluster_docs"].tolist()
    for R in (5, 10):
        if f"beam{R}_labels" in g:
            lab, sc = orq.rq_beam_search(g["X"][:64], g["C"], R)
            assert np.array_equal(lab, g[f"beam{R}_labels"])      # pq.beam_search labels
            assert np.abs(sc - g[f"beam{R}_scores"]).max() <= 1e-4


def test_cluster_index_round_trip():
    from mevi_amd.rq import ClusterIndex

    rng = np.random.default_rng(1)
    codes = rng.integers(0, 8, size=(500, 3)).astype(np.int32)
    idx = ClusterIndex.from_codes(codes, 8, start=1000)
    cluster, mapping = orq.cluster_dict(codes, start=1000)
    c2, m2 = idx.to_dicts()
    assert c2 == cluster and m2 == mapping
    idx2 = ClusterIndex.from_dict(cluster, 3, 8)
    assert np.array_equal(idx2.keys, idx.keys) and np.array_equal(idx2.offsets, idx.offsets)
    assert np.array_equal(idx2.doc_ids, idx.doc_ids)
    missing = next(k for k in ((a, b, c) for a in range(8) for b in range(8) for c in range(8)) if k not in cluster)
    assert idx.lookup(missing).size == 0
    # a pickled dict arrives in insertion (first-seen) order, not key order; lists keep the reference's append order
    shuffled = dict(sorted(cluster.items(), key=lambda kv: kv[1][0]))
    idx3 = ClusterIndex.from_dict(shuffled, 3, 8)
    assert np.array_equal(idx3.keys, idx.keys) and np.array_equal(idx3.offsets, idx.offsets)
    assert np.array_equal(idx3.doc_ids, idx.doc_ids)
    assert ClusterIndex.from_dict({}, 3, 8).lookup((0, 0, 0)).size == 0
    # the inverse map as an array (what rqmapping*.pkl holds as a dict of tuples)
    idx0 = ClusterIndex.from_codes(codes, 8)
    dc = idx0.doc_codes(len(codes) + 2)
    assert np.array_equal(dc[:len(codes)], codes) and (dc[len(codes):] == -1).all()
    from mevi_amd.evalrun import CodeMap
    cm = CodeMap(dc)
    assert cm[17] == tuple(codes[17].tolist()) and len(cm) == len(codes) + 2
    with pytest.raises(KeyError):
        cm[len(codes)]
