"""The consumers on flat arrays and the device (mevi_amd/consumers.py, csrc/consumers.hip) against the dict path of
mevi_amd/metrics.py -- the line-by-line restatement of ensemble_marco.py / evaluate.py that the reference-run golden G6
pins -- and against G6's bytes through the real scripts."""
import json
import os
import pickle
import shutil
import subprocess
import sys

import numpy as np
import pytest
import torch

from mevi_amd import consumers, metrics

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
G6 = os.path.join(ROOT, "tests", "golden", "g6_consumers")


def _lists(rng, nq, n_docs, k, fine_max, M, K, R, repeats=1):
    """Dense + fine lists with everything the reference's loop is sensitive to: -1 padding, documents repeated inside and
    across the lists, equal combined scores, fine lists longer than the dense one, empty lists."""
    codes = rng.integers(0, K, (n_docs, M)).astype(np.int32)
    dense_p, dense_s, fine_p, fine_s, beams = {}, {}, {}, {}, {}
    for i in range(nq):
        q = f"query {i}"
        nd = int(rng.integers(0, k + 1)) if i % 7 == 0 else k
        d = rng.choice(n_docs, nd, replace=False).astype(np.int64)
        if nd > 4 and i % 3 == 0:
            d[rng.integers(0, nd, 3)] = d[0]                       # repeats inside the dense list
        if nd > 10 and i % 4 == 0:
            d[-5:] = -1                                             # faiss padding
        s = np.sort(rng.normal(size=nd))[::-1].copy()
        if nd > 6 and i % 2 == 0:
            s[3:6] = s[3]                                           # equal scores
        nf = int(rng.integers(0, fine_max + 1))
        if i % 5 == 0:
            nf = nd + 7                                             # longer than the dense list: the zip cuts it
        f = rng.choice(n_docs, nf, replace=False).astype(np.int64)
        if nf and nd:
            f[: min(nf, nd, 4)] = d[: min(nf, nd, 4)]               # documents of both lists
        fs = np.sort(rng.normal(size=nf) + 1)[::-1].copy()
        if nf > 3:
            fs[1] = fs[2]
        b = []                                                      # R - 1 distinct clusters ...
        for c in [tuple(codes[x]) for x in d[d >= 0][:3]] + [tuple(c) for c in rng.integers(0, K, (8 * R, M)).tolist()]:
            if c not in b and len(b) < R - repeats:                 # ... some of them holding dense documents
                b.append(tuple(int(v) for v in c))
        b = np.asarray(b + [b[j % len(b)] for j in range(repeats)])  # ... and repeated ones: the LAST index counts
        dense_p[q], dense_s[q] = d.tolist(), s.tolist()
        fine_p[q], fine_s[q] = f.tolist(), fs.tolist()
        beams[q] = b.tolist()
    return codes, dense_p, dense_s, fine_p, fine_s, beams


def _inputs(codes, dense_p, dense_s, fine_p, fine_s, beams, with_fine=True):
    qs = list(dense_p)
    seg = np.concatenate([[0], np.cumsum([len(dense_p[q]) for q in qs])]).astype(np.int64)
    docs = np.concatenate([np.asarray(dense_p[q], np.int64) for q in qs])
    sc = np.concatenate([np.asarray(dense_s[q], np.float64) for q in qs])
    fine = None
    if with_fine:
        fq = list(reversed(qs))                                     # the fine file in another order
        fseg = np.concatenate([[0], np.cumsum([len(fine_p[q]) for q in fq])]).astype(np.int64)
        fdocs = np.concatenate([np.asarray(fine_p[q], np.int64) for q in fq])
        fsc = np.concatenate([np.asarray(fine_s[q], np.float64) for q in fq])
        row = {q: i for i, q in enumerate(fq)}
        fine = (fq, np.asarray([row[q] for q in qs], np.int64), fseg, fdocs, fsc)
    return consumers.EnsembleInputs(qs, seg, docs, sc, np.asarray([beams[q] for q in qs]), codes, fine)


@pytest.mark.parametrize("with_fine", [True, False])
@pytest.mark.parametrize("nq,k,fine_max", [(60, 100, 40), (9, 1000, 3000), (5, 3, 2)])
def test_ensemble_on_the_device_equals_the_dict_loop(cuda, nq, k, fine_max, with_fine):
    rng = np.random.default_rng(nq + k)
    codes, dense_p, dense_s, fine_p, fine_s, beams = _lists(rng, nq, 5000, k, fine_max, 3, 6, 10)
    mapping = metrics.ArrayMapping(codes)
    cranks, n_clusters = metrics.cluster_ranks(dense_p, beams, mapping)
    inp = _inputs(codes, dense_p, dense_s, fine_p, fine_s, beams, with_fine)
    assert inp.n_clusters == n_clusters
    cr = inp.ranks()
    seg = inp.seg_d.cpu().numpy()
    for i, q in enumerate(inp.queries):
        assert cr[seg[i]:seg[i + 1]].cpu().tolist() == cranks[q]
    oseg = inp.out_seg.cpu().numpy()
    for a, b, g in ((0.6, 0.03, 0.02), (0.0, 0.5, 0.9), (1.7, 0.3, 0.0), (0.4, 0.0, 0.5)):
        out_docs, out_n = inp.ensemble(cr, a, b, g)
        out_docs, out_n = out_docs.cpu().numpy(), out_n.cpu().numpy()
        for i, q in enumerate(inp.queries):
            want = _dict_loop(dense_p[q], dense_s[q], cranks[q], fine_p[q] if with_fine else None,
                              fine_s[q] if with_fine else None, n_clusters, a, b, g)
            assert out_docs[oseg[i]:oseg[i] + out_n[i]].tolist() == want, (q, a, b, g)


def test_several_repeated_beam_clusters_rank_past_the_number_of_distinct_ones(cuda):
    """with two or more repeated clusters in a beam the LAST-index ranks run up to R - 1 > n_clusters; the
    term table must cover them (it used to be n_clusters + 1 long: reads past its end, silently wrong scores)."""
    rng = np.random.default_rng(11)
    R = 10
    codes, dense_p, dense_s, fine_p, fine_s, beams = _lists(rng, 24, 5000, 60, 30, 3, 6, R, repeats=3)
    cranks, n_clusters = metrics.cluster_ranks(dense_p, beams, metrics.ArrayMapping(codes))
    assert n_clusters == R - 3 and max(max(c) for c in cranks.values() if c) > n_clusters
    inp = _inputs(codes, dense_p, dense_s, fine_p, fine_s, beams)
    cr = inp.ranks()
    seg, oseg = inp.seg_d.cpu().numpy(), inp.out_seg.cpu().numpy()
    for i, q in enumerate(inp.queries):
        assert cr[seg[i]:seg[i + 1]].cpu().tolist() == cranks[q]
    for a, b, g in ((0.6, 0.03, 0.02), (1.7, 0.3, 0.4)):
        out_docs, out_n = inp.ensemble(cr, a, b, g)
        out_docs, out_n = out_docs.cpu().numpy(), out_n.cpu().numpy()
        for i, q in enumerate(inp.queries):
            want = _dict_loop(dense_p[q], dense_s[q], cranks[q], fine_p[q], fine_s[q], n_clusters, a, b, g)
            assert out_docs[oseg[i]:oseg[i] + out_n[i]].tolist() == want, (q, a, b, g)
    # a zero denominator raises only when some entry carries that rank (beta = -1 / rank), as the per-entry division does
    present = next(c for c in (8, 4, 2, 1) if bool((cr == c).any()))
    with pytest.raises(ZeroDivisionError):
        inp.ensemble(cr, 0.6, -1.0 / present, 0.0)
    assert inp.ensemble(cr, 0.6, -1.0 / 64, 0.0) is not None        # rank 64 does not occur: no error, as the reference


def _dict_loop(dense_p, dense_s, cranks, fine_p, fine_s, n_clusters, alpha, beta, gamma):
    """ensemble_marco.py:222-238 + evaluate()'s sort (:38-39), literally."""
    from itertools import chain

    docs, scores, ranks = dense_p, dense_s, cranks
    if fine_p is not None:
        docs, scores, ranks = dense_p + fine_p, dense_s + fine_s, chain(cranks, cranks)
    combined = {}
    for p, s, cr in zip(docs, scores, ranks):
        combined[p] = s + alpha / (beta * cr + 1)
        if cr == n_clusters:
            combined[p] *= (1 - gamma * alpha)
    return [p for p, _ in sorted(combined.items(), key=lambda kv: -kv[1])]


def test_a_document_without_code_row_raises_like_the_mapping(cuda):
    rng = np.random.default_rng(3)
    codes, dense_p, dense_s, fine_p, fine_s, beams = _lists(rng, 8, 500, 20, 5, 3, 4, 4)
    victim = next(d for d in dense_p["query 1"] if d >= 0)
    codes[victim] = -1
    inp = _inputs(codes, dense_p, dense_s, fine_p, fine_s, beams)
    with pytest.raises(KeyError) as e:
        inp.ranks()
    with pytest.raises(KeyError) as e2:
        metrics.cluster_ranks(dense_p, beams, metrics.ArrayMapping(codes))
    assert e.value.args[0] == victim == e2.value.args[0]


def test_first_hits_equal_gt_ranks(cuda):
    rng = np.random.default_rng(11)
    lists = [rng.integers(0, 50, int(rng.integers(0, 300))).tolist() for _ in range(40)]
    seg = np.concatenate([[0], np.cumsum([len(l) for l in lists])]).astype(np.int64)
    flat = np.concatenate([np.asarray(l, np.int64) for l in lists])
    rows, docs, want = [], [], []
    for r, l in enumerate(lists):
        gt = rng.integers(0, 60, 3).tolist()
        rows += [r] * 3
        docs += gt
        want += [-1 if x is None else x for x in metrics.gt_ranks(l, gt)]
    rows.append(-1), docs.append(5), want.append(-1)
    t = lambda a: torch.from_numpy(a).to(cuda)       # noqa: E731
    got = consumers.first_hits(t(flat), t(seg), None, rows, docs)
    assert got.tolist() == want
    cut = np.asarray([len(l) // 2 for l in lists], np.int32)
    got = consumers.first_hits(t(flat), t(seg), t(cut), rows[:-1], docs[:-1])
    assert got.tolist() == [-1 if x is None else x for r, l in enumerate(lists)
                            for x in metrics.gt_ranks(l[:cut[r]], docs[3 * r:3 * r + 3])]


@pytest.fixture()
def g6_dir(tmp_path):
    for f in os.listdir(G6):
        if f.endswith((".tsv", ".pkl")):
            shutil.copy(os.path.join(G6, f), tmp_path)
    mp = os.path.join(tmp_path, "rqmapping.pkl")
    mapping = pickle.load(open(mp, "rb"))
    codes = np.full((max(mapping) + 1, len(next(iter(mapping.values())))), -1, np.int32)
    for k, v in mapping.items():
        codes[k] = v
    metrics.write_mapping_sidecar(mp, codes)
    return str(tmp_path)


@pytest.mark.parametrize("name", ["evaluate_default", "evaluate_recall5_20", "ensemble_default", "ensemble_grid",
                                  "ensemble_nofine"])
def test_scripts_on_the_device_print_the_reference_bytes(cuda, name, g6_dir):
    """evaluate.py / ensemble_marco.py with MEVI_CONSUMERS=device (a declined input is an error, so the device path is what
    ran): stdout and ofile byte-identical to the UNMODIFIED reference scripts' (golden G6: -1 ids, duplicated documents,
    fine lists longer than the dense one)."""
    exp = json.load(open(os.path.join(G6, "expected.json")))[name]
    argv = [a.replace("{d}", g6_dir) for a in exp["argv"]]
    r = subprocess.run([sys.executable, os.path.join(ROOT, argv[0])] + argv[1:], capture_output=True, text=True, cwd=g6_dir,
                       env=dict(os.environ, PYTHONPATH=ROOT, MEVI_CONSUMERS="device"))
    assert r.returncode == 0, r.stderr[-2000:]
    assert r.stdout == exp["stdout"]
    if exp["ofile"] is not None:
        out = [f for f in ("eval_out.txt", "ens_out.txt") if os.path.exists(os.path.join(g6_dir, f))]
        assert open(os.path.join(g6_dir, out[0])).read() == exp["ofile"]
    assert not os.path.exists(os.path.join(g6_dir, "dense.pkl"))        # the big lists never became dicts


def test_inputs_the_device_path_declines_fall_back_to_the_dict_path(cuda, tmp_path):
    """More than 8192 combined entries for one query (the LDS sort's capacity), or a bracketed list: in the default mode the
    scripts print what the dict path prints; MEVI_CONSUMERS=device refuses instead of falling back."""
    rng = np.random.default_rng(1)
    n_docs, M, K, R = 30000, 3, 5, 4
    codes = rng.integers(0, K, (n_docs, M)).astype(np.int32)
    lens = {"q0": (5000, 5000), "q1": (40, 10), "q2": (1000, 0)}
    with open(tmp_path / "gt.tsv", "w") as fg, open(tmp_path / "dense.tsv", "w") as fd, open(tmp_path / "fine.tsv", "w") as ff, \
            open(tmp_path / "coarse.tsv", "w") as fc:
        for q, (nd, nf) in lens.items():
            d = rng.choice(n_docs, nd, replace=False)
            ds = np.sort(rng.normal(size=nd))[::-1]
            f = rng.choice(n_docs, max(nf, 1), replace=False)
            fs = np.sort(rng.normal(size=max(nf, 1)))[::-1]
            beam = [tuple(int(v) for v in codes[x]) for x in d[:R - 1]]
            while len(set(beam)) < R - 1:
                beam.append(tuple(int(v) for v in rng.integers(0, K, M)))
            beam = list(dict.fromkeys(beam))[:R - 1]
            fg.write(f"{q}\t{int(d[3])},{int(f[0])}\n")
            fd.write(f"{q}\t\t{','.join(map(str, d.tolist()))}\t{','.join(repr(float(x)) for x in ds)}\n")
            ff.write(f"{q}\t\t{','.join(map(str, f.tolist()))}\t{','.join(repr(float(x)) for x in fs)}\n")
            fc.write(f"{q}\t{[list(b) for b in beam] + [list(beam[0])]}\n")
    import pickle

    with open(tmp_path / "rqmapping.pkl", "wb") as f:
        pickle.dump({i: tuple(int(v) for v in codes[i]) for i in range(n_docs)}, f)
    metrics.write_mapping_sidecar(str(tmp_path / "rqmapping.pkl"), codes)
    argv = [sys.executable, os.path.join(ROOT, "ensemble_marco.py"), "--dir_path", str(tmp_path), "--gt_file", "gt.tsv", "--ance_file",
            "dense.tsv", "--fine_file", "fine.tsv", "--coarse_file", "coarse.tsv", "--mapping_file", str(tmp_path / "rqmapping.pkl"),
            "--recall_num", "10,100"]
    outs = {}
    for mode in ("host", "auto", "device"):
        for f in os.listdir(tmp_path):
            if f.endswith(".pkl") and not f.startswith("rqmapping"):
                os.remove(tmp_path / f)                                   # no parse caches between the runs
        r = subprocess.run(argv, capture_output=True, text=True, cwd=tmp_path, env=dict(os.environ, PYTHONPATH=ROOT, MEVI_CONSUMERS=mode))
        outs[mode] = (r.returncode, r.stdout, r.stderr[-600:])
    assert outs["host"][0] == 0 and outs["auto"][:2] == outs["host"][:2], outs["auto"][2]
    assert outs["device"][0] != 0 and "declined" in outs["device"][2]



@pytest.mark.parametrize("with_fine", [True, False])
def test_files_whose_lists_are_all_empty(cuda, with_fine):
    """Found by tools/stress_consumers.py: when every dense (or fine) list of a file is empty the flat arrays have no device
    pointer and the C ABI refused the call ("null pointer").  The ensemble of such a file is what the dict loop gives: the fine
    list cut to the dense list's length, i.e. nothing."""
    qs = ["q a", "q b"]
    codes = np.zeros((10, 2), np.int32)
    dense_p, dense_s = {q: [] for q in qs}, {q: [] for q in qs}
    fine_p, fine_s = {q: [] for q in qs}, {q: [] for q in qs}
    beams = {q: [[0, 0], [1, 1]] for q in qs}
    cranks, n_clusters = metrics.cluster_ranks(dense_p, beams, metrics.ArrayMapping(codes))
    inp = _inputs(codes, dense_p, dense_s, fine_p, fine_s, beams, with_fine)
    cr = inp.ranks()
    assert cr.numel() == 0
    out_docs, out_n = inp.ensemble(cr, 0.6, 0.03, 0.02)
    assert out_n.cpu().tolist() == [0, 0]
    for q in qs:
        assert _dict_loop(dense_p[q], dense_s[q], cranks[q], fine_p[q] if with_fine else None, fine_s[q] if with_fine else None,
                          n_clusters, 0.6, 0.03, 0.02) == []
