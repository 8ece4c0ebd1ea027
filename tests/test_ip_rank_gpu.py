"""GPU: the HIP dense search and fine stage against what the reference ITSELF computed (goldens G9: torch.matmul +
torch.sort / streaming torch.topk exactly as MEVI/main_models.py:3818-3876,3921-4013 issue them).  Same bar as
tests/test_ip_rank_cpu.py: 'int' data bit-equal scores and identical documents inside exact-tie runs; 'flt' data
within the stated f32 summation-order tolerance, identical documents outside near-tie gaps."""
import os

import numpy as np
import pytest
import torch

from mevi_amd import dense
from mevi_amd.fine import FineStage
from mevi_amd.rq import ClusterIndex
from rankcheck import same_ranking

pytestmark = pytest.mark.gpu
GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
TOL = {"int": 0.0, "flt": 2e-4}


def load(name):
    return np.load(os.path.join(GOLD, f"g9_ip_rank_{name}.npz"))


@pytest.mark.parametrize("name", ["int", "flt"])
def test_fine_stage_equals_reference_run(cuda, name):
    g = load(name)
    K = int(g["K"])
    emb = torch.from_numpy(g["emb"]).to(cuda)
    q = torch.from_numpy(g["q"]).to(cuda)
    fs = FineStage(emb, ClusterIndex.from_codes(g["codes"], K))
    out, ndoc = fs.rerank(q, g["beams"])
    seg = g["fine_seg"]
    assert np.array_equal(np.asarray(ndoc), g["fine_ndoc"])
    for b in range(len(g["q"])):
        same_ranking(out[b][1], out[b][0], g["fine_scores"][seg[b]:seg[b + 1]], g["fine_docs"][seg[b]:seg[b + 1]],
                     tol=TOL[name])
    gt_off = np.concatenate([[0], np.cumsum(g["gt_len"])])
    gts = fs.gt_scores(q, [g["gt_flat"][a:b].tolist() for a, b in zip(gt_off[:-1], gt_off[1:])])
    got = np.concatenate(gts)
    if TOL[name] == 0.0:
        assert np.array_equal(got.view(np.uint32), g["gt_scores"].view(np.uint32))
    else:
        assert np.abs(got - g["gt_scores"]).max() <= TOL[name]


@pytest.mark.parametrize("name", ["int", "flt"])
@pytest.mark.parametrize("path", ["exact", "indexed", "search"])
def test_dense_topk_equals_reference_streaming_topk(cuda, name, path):
    g = load(name)
    emb = torch.from_numpy(g["emb"]).to(cuda)
    q = torch.from_numpy(g["q"]).to(cuda)
    n, dim = g["emb"].shape
    full = (q.double() @ emb.double().T).float().cpu().numpy()     # membership of the last run only ('int': exact)
    for key in [k for k in g.files if k.startswith("all") and k.endswith("_docs")]:
        pool = int(key[3:-5])
        ref_i, ref_s = g[key], g[f"all{pool}_scores"]
        if path == "exact":
            s, i = dense.ip_topk(q, emb, pool)
        elif path == "indexed":
            s, i = dense.DenseIndex(emb).search(q, pool)
        else:
            s, i = dense.search(g["q"], g["emb"], dim, pool, "Flat", device=cuda)   # faiss_search.search drop-in
            s, i = torch.from_numpy(s), torch.from_numpy(i)
        s, i = s.cpu().numpy(), i.cpu().numpy()
        kk = min(pool, n)
        if pool > n:
            assert np.all(i[:, n:] == -1)
        for b in range(len(g["q"])):
            same_ranking(s[b, :kk], i[b, :kk], ref_s[b], ref_i[b], tol=TOL[name], full_scores=full[b] if kk < n else None)
