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
    preds, _, _ = mio.load_parsed(resolve(ance_file, dir_path), RANKED_TEMPLATE)
    if ofile is not None:
        open(ofile, "w").close()
    return evaluate_ranked("ANCE Pred", cutoffs, gts, preds, ofile, stdout_prefix="Scoring ")


def cluster_ranks(dense_preds, coarse_clusters, mapping):
    """rank of each dense doc's RQ cluster among the query's beam clusters (else n_clusters)."""
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
    combined = {}
    for p, s, cr in zip(docs, scores, ranks):
        v = s + alpha / (beta * cr + 1)
        if cr == n_clusters:
            v *= (1 - gamma * alpha)
        combined[p] = v
    return [p for p, _ in sorted(combined.items(), key=lambda kv: -kv[1])]


def ensemble_main(dir_path, gt_file, ance_file, fine_file, coarse_file, mapping_file,
                  alphas="0.6", betas="0.03", gammas="0.02", recall_num="10,50,1000", ofile=None):
    if mapping_file is None or not os.path.exists(mapping_file):
        raise AssertionError(f"mapping file {mapping_file} does not exist")
    alphas, betas, gammas = ([float(x) for x in v.split(",")] for v in (alphas, betas, gammas))
    cutoffs = [int(x) for x in recall_num.split(",")]
    gts, _, _ = mio.load_parsed(resolve(gt_file, dir_path), GT_TEMPLATE)
    dense_p, dense_s, _ = mio.load_parsed(resolve(ance_file, dir_path), RANKED_TEMPLATE)
    fine_path = resolve(fine_file, dir_path, must_exist=False)
    have_fine = fine_path is not None and os.path.exists(fine_path)
    if have_fine:
        fine_p, fine_s, _ = mio.load_parsed(fine_path, RANKED_TEMPLATE)
    _, _, clusters = mio.load_parsed(resolve(coarse_file, dir_path), COARSE_TEMPLATE)
    with open(mapping_file, "rb") as f:
        mapping = pickle.load(f)
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
