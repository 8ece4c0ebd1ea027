"""Host-side consumers and writers against goldens captured from the UNMODIFIED reference
(tools/capture_goldens.py g6 g7): evaluate.py / ensemble_marco.py stdout + ofile bytes,
faiss_search.to_file bytes, LogTxtFile lines."""
import json
import os
import shutil
import subprocess
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLD = os.path.join(ROOT, "tests", "golden")
G6 = os.path.join(GOLD, "g6_consumers")


@pytest.fixture()
def g6_dir(tmp_path):
    for f in os.listdir(G6):
        if f.endswith((".tsv", ".pkl")):
            shutil.copy(os.path.join(G6, f), tmp_path)
    return str(tmp_path)


@pytest.mark.parametrize("name", ["evaluate_default", "evaluate_recall5_20", "ensemble_default",
                                  "ensemble_grid", "ensemble_nofine"])
def test_cli_output_matches_reference(name, g6_dir):
    exp = json.load(open(os.path.join(G6, "expected.json")))[name]
    argv = [a.replace("{d}", g6_dir) for a in exp["argv"]]
    cmd = [sys.executable, os.path.join(ROOT, argv[0])] + argv[1:]
    r = subprocess.run(cmd, capture_output=True, text=True, cwd=g6_dir,
                       env=dict(os.environ, PYTHONPATH=ROOT))
    assert r.returncode == 0, r.stderr[-2000:]
    assert r.stdout == exp["stdout"]
    if exp["ofile"] is not None:
        out = [f for f in ("eval_out.txt", "ens_out.txt") if os.path.exists(os.path.join(g6_dir, f))]
        assert open(os.path.join(g6_dir, out[0])).read() == exp["ofile"]


G6N = os.path.join(GOLD, "g6_consumers_nq")


@pytest.mark.parametrize("name", ["nq_default", "nq_grid_nofine", "nq_noensemble"])
def test_nq_ensemble_cli_matches_reference(name, tmp_path):
    """ensemble_nqdpr.py (answer-based hits through the inverse index files) vs the unmodified reference script."""
    for f in os.listdir(G6N):
        if f.endswith((".tsv", ".pkl", ".bin")):
            shutil.copy(os.path.join(G6N, f), tmp_path)
    d = str(tmp_path)
    exp = json.load(open(os.path.join(G6N, "expected.json")))[name]
    argv = [a.replace("{d}", d) for a in exp["argv"]]
    r = subprocess.run([sys.executable, os.path.join(ROOT, argv[0])] + argv[1:], capture_output=True, text=True, cwd=d,
                       env=dict(os.environ, PYTHONPATH=ROOT))
    assert r.returncode == 0, r.stderr[-2000:]
    assert r.stdout == exp["stdout"]
    if exp["ofile"] is not None:
        assert open(os.path.join(d, "ens_out.txt")).read() == exp["ofile"]


def test_stale_parse_cache_is_not_served(g6_dir):
    """The reference pickles the parsed TSV next to it and reuses it forever; we ignore a cache
    older than its TSV (documented deviation, SURVEY 'Bug-compatible ensemble')."""
    from mevi_amd import io as mio
    from mevi_amd.metrics import RANKED_TEMPLATE

    path = os.path.join(g6_dir, "dense.tsv")
    a, _, _ = mio.load_parsed(path, RANKED_TEMPLATE)
    assert os.path.exists(os.path.join(g6_dir, "dense.pkl"))
    lines = open(path).read().splitlines()
    q0 = lines[0].split("\t")[0]
    lines[0] = f"{q0}\t\t7,8,9\t3.0,2.0,1.0"
    with open(path, "w") as f:
        f.write("\n".join(lines) + "\n")
    os.utime(path, (os.path.getmtime(path) + 5, os.path.getmtime(path) + 5))
    b, _, _ = mio.load_parsed(path, RANKED_TEMPLATE)
    assert b[q0] == [7, 8, 9] and a[q0] != b[q0]


def test_to_file_bytes(tmp_path):
    from mevi_amd import io as mio

    g = json.load(open(os.path.join(GOLD, "g7_writers.json")))["to_file"]
    qf, of = tmp_path / "q.tsv", tmp_path / "o.tsv"
    qf.write_text(g["raw_query"])
    dists = np.frombuffer(bytes.fromhex(g["dists_hex"]), dtype=np.float32).reshape(g["shape"])
    mio.to_file(str(qf), str(of), dists, np.array(g["indices"], dtype=np.int64))
    assert of.read_text() == g["expected"]


def test_rank_log_merge_bytes(tmp_path):
    from mevi_amd import io as mio

    g = json.load(open(os.path.join(GOLD, "g7_writers.json")))["logtxt"]
    final = str(tmp_path / "res_coarse.tsv")
    logs = [mio.RankLog(final, r, 2, tmpdir=str(tmp_path)) for r in range(2)]
    for ln in g["rank0"]:
        logs[0].add(ln)
    for ln in g["rank1"]:
        logs[1].add(ln)
    logs[1].flush()
    logs[0].merge()
    assert open(final).read() == g["expected"]
    assert not os.path.exists(logs[0].tmp) and not os.path.exists(logs[1].tmp)
    with pytest.raises(FileExistsError):
        open(logs[0].tmp, "w").close()
        mio.RankLog(final, 0, 2, tmpdir=str(tmp_path))


def test_read_raw_f32(tmp_path):
    from mevi_amd import io as mio

    a = np.arange(24, dtype=np.float32).reshape(3, 8)
    p = tmp_path / "e.bin"
    a.tofile(p)
    assert np.array_equal(mio.read(str(p), 8), a)
    with pytest.raises(ValueError):
        mio.read(str(p), 7)


@pytest.mark.parametrize("name", ["ensemble_default", "ensemble_grid", "ensemble_nofine"])
def test_ensemble_with_the_array_form_of_the_mapping(name, g6_dir):
    """main.py writes rqmapping*.pkl AND the same mapping as an array (rqmapping*.npy); with the array present the
    ensemble script skips the 8.8 M-tuple unpickle and ranks clusters with numpy -- stdout and ofile stay byte-identical to
    the UNMODIFIED reference script's (golden G6, which holds -1 ids, duplicated documents, F > len(dense))."""
    import pickle

    from mevi_amd import metrics

    mp = os.path.join(g6_dir, "rqmapping.pkl")
    mapping = pickle.load(open(mp, "rb"))
    codes = np.full((max(mapping) + 1, len(next(iter(mapping.values())))), -1, np.int32)
    for k, v in mapping.items():
        codes[k] = v
    metrics.write_mapping_sidecar(mp, codes)
    assert isinstance(metrics.load_mapping(mp), metrics.ArrayMapping)
    exp = json.load(open(os.path.join(G6, "expected.json")))[name]
    argv = [a.replace("{d}", g6_dir) for a in exp["argv"]]
    r = subprocess.run([sys.executable, os.path.join(ROOT, argv[0])] + argv[1:], capture_output=True, text=True, cwd=g6_dir,
                       env=dict(os.environ, PYTHONPATH=ROOT))
    assert r.returncode == 0, r.stderr[-2000:]
    assert r.stdout == exp["stdout"]
    if exp["ofile"] is not None:
        assert open(os.path.join(g6_dir, "ens_out.txt")).read() == exp["ofile"]
    os.utime(mp, (os.path.getmtime(mp) + 10,) * 2)                           # a pickle newer than the array wins
    assert isinstance(metrics.load_mapping(mp), dict)


def test_numpy_ensemble_equals_the_reference_loop():
    """_ensemble_scores_numpy against the literal dict loop on adversarial lists: duplicates inside and across the lists,
    exact score ties, a fine list longer than the dense one (zip truncation), ranks equal to n_clusters."""
    from itertools import chain

    from mevi_amd import metrics

    rng = np.random.default_rng(0)
    for trial in range(200):
        nd, nf = int(rng.integers(1, 40)), int(rng.integers(0, 90))
        dense_p = rng.integers(-1, 25, size=nd).tolist()
        fine_p = rng.integers(0, 25, size=nf).tolist()
        dense_s = np.round(rng.standard_normal(nd), 1).tolist()
        fine_s = np.round(rng.standard_normal(nf), 1).tolist()
        cr = rng.integers(0, 6, size=nd).tolist()
        a, b, g, ncl = 0.6, 0.03, 0.02, 5
        for fp, fs in ((fine_p, fine_s), (None, None)):
            docs, scores, ranks = (dense_p + fp, dense_s + fs, chain(cr, cr)) if fp is not None else (dense_p, dense_s, cr)
            combined = {}
            for p, s, c in zip(docs, scores, ranks):
                v = s + a / (b * c + 1)
                if c == ncl:
                    v *= (1 - g * a)
                combined[p] = v
            want = [p for p, _ in sorted(combined.items(), key=lambda kv: -kv[1])]
            assert metrics.ensemble_scores(dense_p, dense_s, cr, fp, fs, ncl, a, b, g) == want


def test_whole_file_parser_equals_the_field_parser(tmp_path):
    """consumers.parse_ranked (one native pass: mevi_parse_tsv_columns) gives the lists io.parse_file gives, and declines
    (None) whatever is not the plain shape, so those files keep their Python meaning."""
    from mevi_amd import consumers, io as mio, metrics

    rng = np.random.default_rng(0)
    p = tmp_path / "ranked.tsv"
    lines = []
    for i in range(50):
        n = int(rng.integers(1, 400))
        ids = rng.integers(-1, 10 ** 7, n)
        sc = (rng.normal(size=n) * 10.0 ** rng.integers(-8, 8, n)).astype(np.float32).astype(np.float64)
        if i == 3:
            sc[:4] = [np.inf, -np.inf, -0.0, 1e-05]
        lines.append(f"query number {i} ?\t\t{','.join(map(str, ids.tolist()))}\t{','.join(str(x) for x in sc.tolist())}")
    p.write_text("\n".join(lines))                                   # no newline at the end
    want_p, want_s, _ = mio.parse_file(str(p), metrics.RANKED_TEMPLATE)
    got = consumers.parse_ranked(str(p), metrics.RANKED_TEMPLATE)
    assert got is not None and got.queries == list(want_p)
    for i, q in enumerate(got.queries):
        a, b = got.seg[i], got.seg[i + 1]
        assert got.docs[a:b].tolist() == want_p[q]
        assert [x.hex() for x in got.scores[a:b].tolist()] == [float(x).hex() for x in want_s[q]]
    ids_only = consumers.parse_ranked(str(p), {"query": 0, "pred": 2})
    assert ids_only.scores is None and np.array_equal(ids_only.docs, got.docs)
    p.write_text("\n".join(lines) + "\n")
    assert np.array_equal(consumers.parse_ranked(str(p), metrics.RANKED_TEMPLATE).docs, got.docs)

    def declined(text):
        p.write_text(text)
        return consumers.parse_ranked(str(p), metrics.RANKED_TEMPLATE) is None

    assert declined("q\t\t[1,2]\t0.5,0.25\n")             # bracketed list
    assert declined("q\t\t1,2\t0.5,0.25\nq\t\t3\t1.0\n")  # the same query twice
    assert declined("q\t\t1,2\t0.5\n")                    # ids and scores differ in number
    assert declined("q\t\t1,2\t0.5,0.25\r\n")             # carriage return
    assert declined("q\t\t1,2\n")                         # a column short
    assert declined("q\t\t\t\n")                          # empty fields
    assert declined("q\t\t1,x\t0.5,0.25\n")               # not a number
    assert declined("q\t\t1,2\t0.5,0.25\n\n")             # an empty line
    assert declined("q\t\t1.5,2\t0.5,0.25\n")             # ids that are not integers


def test_array_mapping_raises_keyerror_outside_the_table_and_stale_sidecars_are_ignored(tmp_path):
    """a negative id other than -1 must not wrap to a row from the end, an id >= N raises KeyError like the
    reference's dict (both ArrayMapping paths); a pickle regenerated within the same timestamp is recognised by its
    fingerprint (size + mtime_ns), not by mtime >=."""
    import pickle

    from mevi_amd import metrics

    codes = np.arange(12, dtype=np.int32).reshape(6, 2)
    m = metrics.ArrayMapping(codes)
    assert m[5] == (10, 11)
    for bad in (-2, 6, 1 << 40):
        with pytest.raises(KeyError):
            m[bad]
        with pytest.raises(KeyError) as e:
            metrics.cluster_ranks({"q": [1, bad, 2]}, {"q": [[0, 1], [2, 3]]}, m)
        assert e.value.args[0] == bad
    ranks, n = metrics.cluster_ranks({"q": [1, -1, 0]}, {"q": [[0, 1], [2, 3]]}, m)       # -1 stays the padding id
    assert ranks["q"] == [1, 2, 0] and n == 2
    mp = str(tmp_path / "rqmapping.pkl")
    with open(mp, "wb") as f:
        pickle.dump({i: tuple(int(v) for v in codes[i]) for i in range(6)}, f)
    metrics.write_mapping_sidecar(mp, codes)
    assert isinstance(metrics.load_mapping(mp), metrics.ArrayMapping)
    st = os.stat(mp)
    with open(mp, "wb") as f:                                   # regenerated: other contents, SAME timestamps
        pickle.dump({i: tuple(int(v) for v in codes[i]) for i in range(5)}, f)
    os.utime(mp, ns=(st.st_atime_ns, st.st_mtime_ns))
    assert isinstance(metrics.load_mapping(mp), dict)
