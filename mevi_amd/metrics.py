"""Recall@k / MRR@k and the MEVI ensemble rule (host logic, float64 like the reference).

Mirrors the consumers of the hot path:
  evaluate.py       (MEVI/evaluate.py:7-157)      -- metrics of one ranked list
  ensemble_marco.py (MEVI/ensemble_marco.py:152-240) -- dense + fine lists re-scored with
      score + alpha / (beta * cluster_rank + 1), times (1 - gamma*alpha) when the doc's
      cluster is not among the beam clusters.
Printed lines are byte-identical to the reference's (tests/golden/g6_consumers), which
requires accumulating the per-query terms in the same order with the same operations.
"""
import os
import pickle
from itertools import chain

from . import io as mio

GT_TEMPLATE = {"query": 0, "pred": -1}
RANKED_TEMPLATE = {"query": 0, "pred": 2, "score": 3}
COARSE_TEMPLATE = {"query": 0, "cluster": 1}


def gt_ranks(preds, gt):
    """Rank (first occurrence) of every gt doc in `preds`, None when absent."""
    first = {}
    for i, p in enumerate(preds):
        if p not in first:
            first[p] = i
    return [first.get(g) for g in gt]


def accumulate(ranks, cutoffs, recall, mrr):
    found = [r for r in ranks if r is not None]
    if not found:
        return
    best = min(found)
    for k in cutoffs:
        recall[k] += sum(r < k for r in found) / len(ranks)
        if best < k:
            mrr[k] += 1 / (best + 1)


def report(title, cutoffs, recall, mrr, n, ofile, stdout_prefix=""):
    lines = [f"Recall{k} {recall[k] / n}" for k in cutoffs] + [f"MRR{k} {mrr[k] / n}" for k in cutoffs]
    print(f"{stdout_prefix}{title}")
    print("\n".join(lines))
    print()
    if ofile is not None:
        with open(ofile, "a") as f:
            f.write(f"Scoring {title}\n" + "\n".join(lines) + "\n\n")


def evaluate_ranked(title, cutoffs, gts, ranked, ofile=None, stdout_prefix=""):
    """Metrics of `ranked[q]` (list of doc ids) against `gts[q]`; returns (recall, mrr) dicts."""
    recall = {k: 0 for k in cutoffs}
    mrr = {k: 0 for k in cutoffs}
    for q, gt in gts.items():
        accumulate(gt_ranks(ranked[q], gt), cutoffs, recall, mrr)
    n = len(gts)
    report(title, cutoffs, recall, mrr, n, ofile, stdout_prefix)
    return {k: v / n for k, v in recall.items()}, {k: v / n for k, v in mrr.items()}


def resolve(path, dir_path, must_exist=True):
    """A path may be given relative to --dir_path (evaluate.py:74-81)."""
    if path is not None and not os.path.exists(path) and dir_path is not None:
        path = os.path.join(dir_path, path)
    if must_exist and (path is None or not os.path.exists(path)):
        raise FileNotFoundError(path)
    return path


def evaluate_main(dir_path, gt_file, ance_file, recall_num, ofile=None):
    cutoffs = [int(r) for r in recall_num.split(",")]
    gts, _, _ = mio.load_parsed(resolve(gt_file, dir_path), GT_TEMPLATE)
    from . import consumers

    fast = consumers.evaluate_main(gts, resolve(ance_file, dir_path), cutoffs, ofile)      # flat arrays + device look-up
    if fast is not None:
        return fast
    preds, _, _ = mio.load_parsed(resolve(ance_file, dir_path), RANKED_TEMPLATE)
    if ofile is not None:
        open(ofile, "w").close()
    return evaluate_ranked("ANCE Pred", cutoffs, gts, preds, ofile, stdout_prefix="Scoring ")


class ArrayMapping:
    """`rqmapping*.pkl` (dict doc id -> code tuple, main_models.py:3200-3203) held as an i32 [N, M] array (row of -1 = id
    absent).  The pickle of 8.8 M tuples takes seconds to load; main.py / the index build write the array beside it
    (`<same name>.npy`) and the ensemble scripts prefer it when it is not older than the pickle."""

    def __init__(self, codes):
        self.codes = codes

    def __getitem__(self, doc):
        if not 0 <= doc < len(self.codes):         # a negative id would wrap to a row from the end; the dict raises KeyError
            raise KeyError(doc)
        c = self.codes[doc]
        if c[0] < 0:
            raise KeyError(doc)
        return tuple(int(x) for x in c)

    def __len__(self):
        return len(self.codes)


def mapping_sidecar(mapping_file):
    return mapping_file[:-4] + ".npy" if mapping_file.endswith(".pkl") else mapping_file + ".npy"


def _pickle_fingerprint(mapping_file):
    st = os.stat(mapping_file)
    return {"size": int(st.st_size), "mtime_ns": int(st.st_mtime_ns)}


def write_mapping_sidecar(mapping_file, codes):
    """The mapping as an array beside its pickle, plus the pickle's fingerprint (size, mtime_ns) the array was made for:
    load_mapping trusts the array only while the pickle still carries it (comparing mtimes with >= kept a stale array when
    the pickle was regenerated within the timestamp granularity)."""
    import json

    import numpy as np

    side = mapping_sidecar(mapping_file)
    np.save(side, np.ascontiguousarray(codes, dtype=np.int32))
    with open(side + ".json", "w") as f:
        json.dump(_pickle_fingerprint(mapping_file) if os.path.exists(mapping_file) else {}, f)


def load_mapping(mapping_file):
    import json

    import numpy as np

    side = mapping_sidecar(mapping_file)
    if os.path.exists(side) and os.path.exists(side + ".json"):
        try:
            with open(side + ".json") as f:
                made_for = json.load(f)
        except ValueError:
            made_for = None
        if made_for and made_for == _pickle_fingerprint(mapping_file):
            return ArrayMapping(np.load(side, mmap_mode="r"))
    with open(mapping_file, "rb") as f:
        return pickle.load(f)


def _cluster_ranks_array(dense_preds, coarse_clusters, mapping):
    """cluster_ranks for an ArrayMapping: the same ranks, one numpy comparison per query instead of a dict lookup per
    document (a query's 1000 documents x 10 beam clusters)."""
    import numpy as np

    out, n_clusters = {}, None
    for q, preds in dense_preds.items():
        beam = np.asarray(coarse_clusters[q], dtype=np.int64).reshape(len(coarse_clusters[q]), -1)
        distinct = len({tuple(c) for c in beam.tolist()})
        if n_clusters is not None and n_clusters != distinct:
            raise AssertionError("queries disagree on the number of beam clusters")
        n_clusters = distinct
        docs = np.asarray(preds, dtype=np.int64)
        valid = docs != -1
        inside = (docs >= 0) & (docs < len(mapping.codes))
        dcodes = np.asarray(mapping.codes[np.where(inside, docs, 0)], dtype=np.int64)
        bad = valid & (~inside | (dcodes[:, 0] < 0))       # ids outside the table or without a code row: KeyError, as the dict
        if bad.any():
            raise KeyError(int(docs[bad][0]))
        R = beam.shape[0]
        if R == 0 or beam.shape[1] != dcodes.shape[1]:
            out[q] = [n_clusters] * len(docs)
            continue
        eq = (dcodes[:, None, :] == beam[None, :, :]).all(-1)                  # [n, R]
        last = R - 1 - np.argmax(eq[:, ::-1], axis=1)                          # a repeated cluster keeps its LAST index
        out[q] = np.where(eq.any(1) & valid, last, n_clusters).tolist()
    return out, n_clusters


def cluster_ranks(dense_preds, coarse_clusters, mapping):
    """rank of each dense doc's RQ cluster among the query's beam clusters (else n_clusters)."""
    if isinstance(mapping, ArrayMapping):
        return _cluster_ranks_array(dense_preds, coarse_clusters, mapping)
    out, n_clusters = {}, None
    for q, preds in dense_preds.items():
        pos = {}
        for i, c in enumerate(coarse_clusters[q]):
            pos[tuple(c)] = i   # a repeated cluster keeps its LAST index, as the reference's dict does
        if n_clusters is not None and n_clusters != len(pos):
            raise AssertionError("queries disagree on the number of beam clusters")
        n_clusters = len(pos)
        out[q] = [pos.get(mapping[p], n_clusters) if p != -1 else n_clusters for p in preds]
    return out, n_clusters


def ensemble_scores(dense_p, dense_s, cranks, fine_p, fine_s, n_clusters, alpha, beta, gamma):
    """One query's ensemble: returns doc ids ordered by descending combined score.

    Reference quirks kept on purpose (ensemble_marco.py:200-208,231-238): the fine list
    reuses the DENSE list's cluster ranks position by position, the three sequences are
    zipped (so they truncate to the shortest), a doc seen twice keeps its first position
    but its last score, and ties keep first-seen order (stable sort)."""
    docs, scores = dense_p, dense_s
    ranks = cranks
    if fine_p is not None:
        docs = dense_p + fine_p
        scores = dense_s + fine_s
        ranks = chain(cranks, cranks)
    fast = _ensemble_scores_numpy(docs, scores, cranks, fine_p is not None, n_clusters, alpha, beta, gamma)
    if fast is not None:
        return fast
    combined = {}
    for p, s, cr in zip(docs, scores, ranks):
        v = s + alpha / (beta * cr + 1)
        if cr == n_clusters:
            v *= (1 - gamma * alpha)
        combined[p] = v
    return [p for p, _ in sorted(combined.items(), key=lambda kv: -kv[1])]


def _ensemble_scores_numpy(docs, scores, cranks, doubled, n_clusters, alpha, beta, gamma):
    """The loop of ensemble_scores in numpy float64 -- the same IEEE operations in the same order per element
    (s + alpha / (beta * cr + 1), then * (1 - gamma * alpha)), a document seen twice keeps its FIRST position and its LAST
    score, ties keep first-seen order.  None when the lists are not plain numbers (the generic loop then runs)."""
    import numpy as np

    try:
        p = np.asarray(docs)
        s = np.asarray(scores, dtype=np.float64)
        cr = np.asarray(cranks)
        if p.ndim != 1 or p.dtype.kind not in "iu" or s.ndim != 1 or cr.ndim != 1 or cr.dtype.kind not in "iu":
            return None
    except (ValueError, TypeError):
        return None
    if doubled:
        cr = np.concatenate([cr, cr])
    n = min(len(p), len(s), len(cr))                       # zip() truncates to the shortest
    p, s, cr = p[:n], s[:n], cr[:n]
    if n == 0:
        return []
    v = s + alpha / (beta * cr + 1)
    v = np.where(cr == n_clusters, v * (1 - gamma * alpha), v)
    uniq, first, inv = np.unique(p, return_index=True, return_inverse=True)
    last = np.empty(len(uniq), dtype=np.float64)
    order_in = np.argsort(inv, kind="stable")               # occurrences of each document in list order
    last[inv[order_in]] = v[order_in]                       # ascending position per document: the last one is written last
    order = np.lexsort((first, -last))                      # by -score, then by first position (stable sort of the dict)
    return uniq[order].tolist()


def ensemble_main(dir_path, gt_file, ance_file, fine_file, coarse_file, mapping_file,
                  alphas="0.6", betas="0.03", gammas="0.02", recall_num="10,50,1000", ofile=None):
    if mapping_file is None or not os.path.exists(mapping_file):
        raise AssertionError(f"mapping file {mapping_file} does not exist")
    from .phases import mark

    mark("start-up + imports")
    alphas, betas, gammas = ([float(x) for x in v.split(",")] for v in (alphas, betas, gammas))
    cutoffs = [int(x) for x in recall_num.split(",")]
    gts, _, _ = mio.load_parsed(resolve(gt_file, dir_path), GT_TEMPLATE)
    ance_path = resolve(ance_file, dir_path)
    fine_path = resolve(fine_file, dir_path, must_exist=False)
    have_fine = fine_path is not None and os.path.exists(fine_path)
    _, _, clusters = mio.load_parsed(resolve(coarse_file, dir_path), COARSE_TEMPLATE)
    mapping = load_mapping(mapping_file)
    mark("gt + coarse TSV + doc -> code mapping read")
    from . import consumers

    fast = consumers.ensemble_main(gts, ance_path, fine_path if have_fine else None, clusters, mapping, alphas, betas, gammas,
                                   cutoffs, ofile)            # the big lists as flat arrays, the arithmetic on the device
    if fast is not None:
        mark("dense + fine TSV parsed, ranks / combination / metrics (device path)")
        return fast
    dense_p, dense_s, _ = mio.load_parsed(ance_path, RANKED_TEMPLATE)
    if have_fine:
        fine_p, fine_s, _ = mio.load_parsed(fine_path, RANKED_TEMPLATE)
    cranks, n_clusters = cluster_ranks(dense_p, clusters, mapping)
    if ofile is not None:
        open(ofile, "w").close()
    results = {"ANCE Pred": evaluate_ranked("ANCE Pred", cutoffs, gts, dense_p, ofile)}
    if have_fine:
        results["Fine Pred"] = evaluate_ranked("Fine Pred", cutoffs, gts, fine_p, ofile)
    for a in alphas:
        for b in betas:
            for g in gammas:
                ranked = {q: [] for q in gts}
                for q in dense_p:
                    ranked[q] = ensemble_scores(dense_p[q], dense_s[q], cranks[q],
                                                fine_p[q] if have_fine else None,
                                                fine_s[q] if have_fine else None, n_clusters, a, b, g)
                title = f"score + {a} / ({b} * crank + 1); punishment (1 - {g} * {a})"
                results[title] = evaluate_ranked(title, cutoffs, gts, ranked, ofile)
    return results


# ---- Natural Questions (answer-based hits): ensemble_nqdpr.py -------------------------------------------------
def parse_indexed(path, template, index_of=None):
    """(pred, score, cluster) dicts keyed by LINE INDEX (the dense file's order), as ensemble_nqdpr.py:81-113 does:
    `index_of` maps the query text of the other files to its line in the dense file (None: the file's own lines)."""
    qi = template["query"]
    cols = [template.get(k) for k in ("pred", "score", "cluster")]
    out = ({}, {}, {})
    with open(path, "r") as f:
        for i, line in enumerate(f):
            items = line.rstrip("\n").split("\t")
            key = i if index_of is None else index_of[items[qi]]
            for c, d in zip(cols, out):
                if c is not None:
                    d[key] = mio.parse_list(items[c])
    return out


def nq_first_hit(qind, preds, offsets, array):
    """Rank of the first doc that answers question `qind`: test_inverse_{offsets,array}.bin list, per doc, the
    questions it answers (ensemble_nqdpr.py:27-31; a padded id -1 indexes an empty slice, as there)."""
    for j, res in enumerate(preds):
        if qind in array[offsets[res]:offsets[res + 1]]:
            return j
    return None


def evaluate_nq(title, cutoffs, nq_eval, ranked, ofile=None):
    offsets, array = nq_eval
    mrr = {k: 0 for k in cutoffs}
    hit = {k: 0 for k in cutoffs}
    for qind, preds in ranked.items():
        ind = nq_first_hit(qind, preds, offsets, array)
        for k in cutoffs:
            if ind is not None:
                mrr[k] += 1 / (ind + 1) if ind < k else 0
                hit[k] += ind < k
    n = len(ranked)
    lines = [f"MRR{k} {mrr[k] / n}" for k in cutoffs] + [f"HitRate{k} {hit[k] / n}" for k in cutoffs]
    print(f"{title}")
    print("\n".join(lines))
    print()
    if ofile is not None:
        with open(ofile, "a") as f:
            f.write(f"Scoring {title}\n" + "\n".join(lines) + "\n\n")
    return {k: v / n for k, v in mrr.items()}, {k: v / n for k, v in hit.items()}


def ensemble_nqdpr_main(dir_path, ance_file, fine_file=None, coarse_file=None, mapping_file=None, alphas="0.4",
                        betas="0.03", gammas="0.02", recall_num="5,20,100", ofile=None, noensemble=False):
    """MEVI/ensemble_nqdpr.py:166-252: the marco ensemble with NQ's answer-based hit test, queries keyed by their
    line in the dense file."""
    import numpy as np

    if mapping_file is None or not os.path.exists(mapping_file):
        raise AssertionError(f"mapping file {mapping_file} does not exist")
    alphas, betas, gammas = ([float(x) for x in v.split(",")] for v in (alphas, betas, gammas))
    cutoffs = [int(x) for x in recall_num.split(",")]
    ance_path = resolve(ance_file, dir_path)
    fine_path = resolve(fine_file, dir_path, must_exist=False)
    have_fine = fine_path is not None and os.path.exists(fine_path)
    nq_eval = (np.memmap(os.path.join(dir_path, "test_inverse_offsets.bin"), mode="r", dtype=np.int32),
               np.memmap(os.path.join(dir_path, "test_inverse_array.bin"), mode="r", dtype=np.int32))
    dense_p, dense_s, _ = parse_indexed(ance_path, RANKED_TEMPLATE)
    index_of = {}
    with open(ance_path, "r") as f:
        for i, line in enumerate(f):
            index_of[line.rstrip("\n").split("\t")[0]] = i
    if have_fine:
        fine_p, fine_s, _ = parse_indexed(fine_path, RANKED_TEMPLATE, index_of)
    if ofile is not None:
        open(ofile, "w").close()
    results = {"ANCE Pred": evaluate_nq("ANCE Pred", cutoffs, nq_eval, dense_p, ofile)}
    if have_fine:
        results["Fine Pred"] = evaluate_nq("Fine Pred", cutoffs, nq_eval, fine_p, ofile)
    if noensemble:
        return results
    _, _, clusters = parse_indexed(resolve(coarse_file, dir_path), COARSE_TEMPLATE, index_of)
    mapping = load_mapping(mapping_file)
    cranks, n_clusters = cluster_ranks(dense_p, clusters, mapping)
    for a in alphas:
        for b in betas:
            for g in gammas:
                ranked = {q: ensemble_scores(dense_p[q], dense_s[q], cranks[q], fine_p[q] if have_fine else None,
                                             fine_s[q] if have_fine else None, n_clusters, a, b, g) for q in dense_p}
                title = f"score + {a} / ({b} * crank + 1); punishment (1 - {g} * {a})"
                results[title] = evaluate_nq(title, cutoffs, nq_eval, ranked, ofile)
    return results
