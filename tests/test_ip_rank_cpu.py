"""CPU: the dense / fine-stage oracle against what the reference ITSELF computed (goldens G9,
tools/capture_goldens.py g9_ip_rank): DocumentEncoder.generate (torch.matmul) + torch.sort per query as the fine
stage of infer() runs them (MEVI/main_models.py:3921-4013), and the --eval_all_documents streaming torch.topk loop
(:3818-3876).  'int' data: bit-equal scores, ids equal as sets inside exact-tie runs.  'flt' data: scores within the
f32 summation-order tolerance stated below, ids equal outside near-tie gaps."""
import os

import numpy as np
import pytest

from oracle import dense as odense
from oracle import rq as orq
from rankcheck import same_ranking

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
# |score| <= ~200 on the 'flt' set (dim 128, |x| ~ 1.3): 128 roundings of <= 2^-24 * 200 each, both orders
TOL = {"int": 0.0, "flt": 2e-4}


def load(name):
    return np.load(os.path.join(GOLD, f"g9_ip_rank_{name}.npz"))


@pytest.mark.parametrize("name", ["int", "flt"])
def test_fine_stage_oracle_pinned_by_reference(name):
    g = load(name)
    cluster, _ = orq.cluster_dict(g["codes"])
    seg = g["fine_seg"]
    gt_off = np.concatenate([[0], np.cumsum(g["gt_len"])])
    for b in range(len(g["q"])):
        docs, sc, ndoc = odense.fine_stage(g["q"][b], g["emb"], cluster, g["beams"][b])
        assert ndoc == g["fine_ndoc"][b]
        one = same_ranking(sc, docs, g["fine_scores"][seg[b]:seg[b + 1]], g["fine_docs"][seg[b]:seg[b + 1]], tol=TOL[name])
        assert name == "int" or b == 1 or one > 0.95                  # 'flt': (nearly) every position compared one to one
        gts = odense.pair_dot(g["q"][b], g["emb"][g["gt_flat"][gt_off[b]:gt_off[b + 1]]])
        ref = g["gt_scores"][gt_off[b]:gt_off[b + 1]]
        if TOL[name] == 0.0:
            assert np.array_equal(gts.view(np.uint32), ref.view(np.uint32))
        else:
            assert np.abs(gts - ref).max() <= TOL[name]
    assert g["fine_ndoc"][2] == 0 and seg[3] == seg[2]                 # the all-empty beam list is in the fixture


@pytest.mark.parametrize("name", ["int", "flt"])
def test_exact_topk_oracle_pinned_by_reference(name):
    g = load(name)
    q, emb = g["q"], g["emb"]
    n = emb.shape[0]
    full = np.stack([odense.pair_dot(q[b], emb) for b in range(len(q))])
    for key in [k for k in g.files if k.startswith("all") and k.endswith("_docs")]:
        pool = int(key[3:-5])
        ref_i, ref_s = g[key], g[f"all{pool}_scores"]
        s, i = odense.ip_topk_exact(q, emb, pool)
        kk = min(pool, n)
        assert ref_i.shape == (len(q), kk)                             # torch.topk(k=min(n, pool_size))
        if pool > n:                                                   # faiss-style padding beyond the corpus
            assert np.all(i[:, n:] == -1) and np.all(s[:, n:] == -odense.FLT_MAX)
        for b in range(len(q)):
            same_ranking(s[b, :kk], i[b, :kk], ref_s[b], ref_i[b], tol=TOL[name],
                         full_scores=full[b] if kk < n else None)


def test_fixture_holds_the_hard_cases():
    g = load("int")
    s = g["fine_scores"]
    seg = g["fine_seg"]
    ties = sum(int(np.sum(s[a:b][1:] == s[a:b][:-1])) for a, b in zip(seg[:-1], seg[1:]))
    assert ties > 0                                                    # exact ties between different documents
    d1 = g["fine_docs"][seg[1]:seg[2]]
    assert len(np.unique(d1)) < len(d1)                                # the repeated beam cluster: documents listed twice
    assert np.array_equal(g["emb"][100], g["emb"][5])                  # duplicated passages
