"""BASELINE.json configs[0] (C1) exactly as SURVEY.md 8(d) defines it: 1000 x 768 queries (seed 1234) x 100 000 docs
(seed 4321), doc g_i = 97 i mod 100000 overwritten by Q_i + 0.1 N(0,1); files query_emb.bin / docemb.bin /
raw_query.tsv; `faiss_search.py --param Flat` (top-1000, the script's default) then `evaluate.py`, both through
their command lines.  Bar: ids identical and scores bit-equal to the oracle through the TSV; MRR@10 / Recall equal
to the values computed from the oracle's lists."""
import os
import subprocess
import sys

import numpy as np
import pytest

from oracle import dense as odense

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_c1_config(cuda, tmp_path):
    nq, nd, dim, k = 1000, 100_000, 768, 1000
    Q = np.random.default_rng(1234).standard_normal((nq, dim), dtype=np.float32)
    rd = np.random.default_rng(4321)
    D = rd.standard_normal((nd, dim), dtype=np.float32)
    g = (np.arange(nq) * 97) % nd
    D[g] = Q + np.float32(0.1) * rd.standard_normal((nq, dim), dtype=np.float32)
    Q.tofile(tmp_path / "query_emb.bin")
    D.tofile(tmp_path / "docemb.bin")
    with open(tmp_path / "raw_query.tsv", "w") as f:
        for i in range(nq):
            f.write(f"q{i}\t{g[i]}\n")
    env = dict(os.environ, PYTHONPATH=ROOT)
    out = str(tmp_path / "dense.tsv")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "faiss_search.py"), "--query_path", str(tmp_path / "query_emb.bin"),
                        "--doc_path", str(tmp_path / "docemb.bin"), "--output_path", out, "--raw_query_path",
                        str(tmp_path / "raw_query.tsv"), "--param", "Flat"], capture_output=True, text=True, env=env)
    assert r.returncode == 0, r.stderr[-1500:]
    assert "Param Flat trained: True." in r.stdout and f"int64 ({nq}, {k}) float32 ({nq}, {k})" in r.stdout
    es, ei = odense.ip_topk_exact(Q, D, k)
    lines = [l.rstrip("\n").split("\t") for l in open(out)]
    assert len(lines) == nq
    for i, l in enumerate(lines):
        assert l[0] == f"q{i}" and l[1] == ""
        assert np.array_equal(np.array(l[2].split(","), dtype=np.int64), ei[i])
        got = np.array([float(x) for x in l[3].split(",")], dtype=np.float64)      # str(float) of the widened f32
        assert np.array_equal(got.astype(np.float32).view(np.uint32), es[i].view(np.uint32)) and \
            np.array_equal(got, es[i].astype(np.float64))
    r = subprocess.run([sys.executable, os.path.join(ROOT, "evaluate.py"), "--dir_path", str(tmp_path), "--gt_file",
                        "raw_query.tsv", "--ance_file", "dense.tsv", "--ofile", str(tmp_path / "metrics.txt")],
                       capture_output=True, text=True, env=env)
    assert r.returncode == 0, r.stderr[-1500:]
    printed = dict(l.split() for l in r.stdout.splitlines() if l.startswith(("Recall", "MRR")))
    rank = np.array([int(np.flatnonzero(ei[i] == g[i])[0]) if g[i] in ei[i] else k for i in range(nq)])
    for c in (1, 5, 10, 20, 50, 100, 1000):
        # evaluate.py accumulates per query in file order: hits/|gt| and 1/(rank+1) (MEVI/evaluate.py:120-150)
        rec = mrr = 0
        for x in rank:
            rec += (1 if x < c else 0) / 1
            if x < c:
                mrr += 1 / (x + 1)
        assert float(printed[f"Recall{c}"]) == rec / nq and float(printed[f"MRR{c}"]) == mrr / nq
    assert float(printed["MRR10"]) > 0.99                                          # the planted documents are found
    assert open(tmp_path / "metrics.txt").read().startswith("Scoring ANCE Pred\nRecall1 ")
