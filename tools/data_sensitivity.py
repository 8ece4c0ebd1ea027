#!/usr/bin/env python3
"""Data sensitivity of the two DATA-DEPENDENT fast paths at full C2 size.

The headline of bench.py is measured on an i.i.d. Gaussian corpus (bench.gen_block) and the RQ encode on a random codebook.
Both paths are exact by construction -- the f16 pre-filter of the dense search selects, the f32 chains and the per-query proof
decide (csrc/ip_topk.hip); the matrix-core RQ shortlist is re-checked by exact chains (csrc/rq_fast.hip) -- but what they COST
depends on the data: rows passing the running threshold, queries the proof cannot close (second pass / exact fallback),
ambiguous row-levels.  This tool times them on corpora shaped like a dense retriever's output (tools/synth.py: clustered,
T5-ANCE-like scale with a common component, 1 % exact + 1 % near duplicates) and with a TRAINED codebook
(rq.train_rq_codebook = what MEVI/pq.py:550-598 produces), and checks the results bit for bit against the exact paths.

  python tools/data_sensitivity.py [--docs N] [--queries Q] [--kinds iid,clustered,...] [--out file.json] [--no-rq]

Reference sites: MEVI/faiss_search.py:13-21 (search), MEVI/pq.py:281-305 (get_rq_document_cluster), :550-598 (training)."""
import argparse
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))
import synth  # noqa: E402
from mevi_amd import dense, hip, rq  # noqa: E402

DIM, TOPK = 768, 1000


def _timed(fn, reps):
    ms = []
    for _ in range(reps):
        torch.cuda.synchronize()
        t = time.perf_counter()
        out = fn()
        torch.cuda.synchronize()
        ms.append((time.perf_counter() - t) * 1e3)
    return sorted(ms)[len(ms) // 2], ms, out


def dense_leg(kind, docs, info, nq, k=TOPK, reps=3, exact_check=True):
    """DenseIndex.search of nq planted queries on corpus `docs`: ms (median of `reps`), filter candidates per query, queries
    to the second pass / exact fallback, and the lists against the exact f32 path (dense.ip_topk), bit for bit."""
    L = hip.lib()
    q, planted = synth.corpus_queries(kind, docs, nq, info)
    index = dense.DenseIndex(docs)                           # first build: also grows the allocator by the 13.6 GB image
    del index
    t_idx, _, index = _timed(lambda: dense.DenseIndex(docs), 1)
    index.search(q, k)                                       # warm: kernels, allocator
    ms, ms_all, (s, i) = _timed(lambda: index.search(q, k), reps)
    L.mevi_ip_topk_set_profiling(2)                          # one more search with the per-launch events and the candidate counters
    try:
        s2, i2 = index.search(q, k)
        torch.cuda.synchronize()
        st = hip.IpTopkStats()
        L.mevi_ip_topk_get_stats(st)
    finally:
        L.mevi_ip_topk_set_profiling(0)
    rec = {
        "kind": kind, "queries": nq, "rows": int(docs.shape[0]), "row_norm": info.get("row_norm"),
        "ms_per_search": round(ms, 2), "ms_all": [round(x, 2) for x in ms_all], "queries_per_s": round(nq / ms * 1e3, 1),
        "index_build_ms": round(t_idx, 1),
        "filter_launches": int(st.n_chunks), "filter_ms": round(st.filter_ms, 2), "compact_ms": round(st.compact_ms, 2),
        "tail_ms": round(ms - st.filter_ms - st.compact_ms, 2),
        "filter_candidates_per_query": round(st.n_filter_candidates / max(nq, 1), 1),
        "max_candidates_one_query_one_launch": int(st.max_launch_candidates), "list_overflows": int(st.n_list_overflows),
        "second_pass_queries": int(st.n_second_pass_queries), "exact_fallback_queries": int(st.n_failed_queries),
        "max_err_over_bound": round(st.max_err_ratio, 4),
        "planted_in_top10": float((i[:, :10] == planted[:, None]).any(1).float().mean()),
    }
    if exact_check:
        t_ex, _, (es, ei) = _timed(lambda: dense.ip_topk(q, docs, k), 1)
        rec["exact_f32_path_ms"] = round(t_ex, 1)
        rec["lists_identical_to_exact_f32_path"] = bool(torch.equal(i, ei) and torch.equal(s.view(torch.int32), es.view(torch.int32))
                                                        and torch.equal(i2, ei))
        del es, ei
    del index, q, s, i, s2, i2
    torch.cuda.empty_cache()
    return rec


def rq_leg(kind, docs, shapes=((4, 32), (3, 256)), sample_rows=1_000_000, reps=3):
    """rq_encode of the whole corpus against a codebook TRAINED on a strided 1 M-row sample of it: ms, ambiguous row-levels,
    rows handed to the exact kernel, codes against mode='exact'."""
    n = docs.shape[0]
    out = {}
    step = max(1, n // sample_rows)
    sample = docs[::step][:sample_rows].contiguous()
    rq.KEEP_ENCODE_WORKSPACE = True
    try:
        for M, K in shapes:
            t0 = time.perf_counter()
            cb, _ = rq.train_rq_codebook(sample, M, K, seed=1, n_init=3, max_iter=40)
            torch.cuda.synchronize()
            t_train = time.perf_counter() - t0
            rq.rq_encode(docs[:1 << 16], cb)
            rq.rq_encode(docs, cb)                       # sizes the workspace
            ms, ms_all, codes = _timed(lambda: rq.rq_encode(docs, cb), reps)
            stats = {k_: v for k_, v in rq.last_encode_stats().items() if isinstance(v, (int, float, str))}
            t_ex, _, exact = _timed(lambda: rq.rq_encode(docs, cb, mode="exact"), 1)
            byts = 4.0 * n * DIM + 4.0 * n * M
            sizes = torch.unique((codes.long() * (K ** torch.arange(M, device=codes.device))).sum(1), return_counts=True)[1]
            out["%dx%d" % (M, K)] = {
                "ms": round(ms, 2), "ms_all": [round(x, 2) for x in ms_all], "train_s_on_%d_rows" % sample.shape[0]: round(t_train, 2),
                "stats": stats, "ambiguous_row_levels_frac": round(stats.get("ambiguous_row_levels", 0) / (n * M), 4),
                "codes_identical_to_exact_mode": bool(torch.equal(codes, exact)),
                "rows_differing": int((codes != exact).any(1).sum()), "exact_mode_ms": round(t_ex, 1),
                "frac_of_hbm_roofline": round(byts / ms / 1e6 / 8000.0, 4),
                "clusters_used": int(sizes.numel()), "docs_per_cluster_mean": round(float(n / sizes.numel()), 2),
                "docs_per_cluster_max": int(sizes.max())}
            del cb, codes, exact, sizes
    finally:
        rq.KEEP_ENCODE_WORKSPACE = False
        rq._LAST_ENCODE.clear()
    torch.cuda.empty_cache()
    return out


def sweep(device, n_docs, nq, kinds=synth.CORPUS_KINDS, with_rq=True, rq_kinds=("iid", "clustered", "ance_scale"), k=TOPK,
          rq_shapes=((4, 32), (3, 256))):
    per = {}
    for kind in kinds:
        docs, info = synth.corpus(kind, device, n_docs, DIM)
        rec = {"dense": dense_leg(kind, docs, info, nq, k)}
        if with_rq and kind in rq_kinds:
            rec["rq_trained_codebook"] = rq_leg(kind, docs, rq_shapes)
        per[kind] = rec
        del docs, info
        torch.cuda.empty_cache()
    qps = {k_: v["dense"]["queries_per_s"] for k_, v in per.items()}
    base = qps.get("iid") or max(qps.values())
    out = {
        "what": "DenseIndex.search (top-%d) and rq_encode with a trained codebook at %d rows x %d queries on four corpus "
                "distributions (tools/synth.py); every result compared bit for bit with the exact path" % (k, n_docs, nq),
        "kinds": list(per), "dense_min": min(qps.values()), "dense_max": max(qps.values()),
        "dense_worst_vs_iid": round(min(qps.values()) / base, 4),
        "all_lists_identical": all(v["dense"].get("lists_identical_to_exact_f32_path", True) for v in per.values()),
        "second_pass_queries_max": max(v["dense"]["second_pass_queries"] for v in per.values()),
        "exact_fallback_queries_max": max(v["dense"]["exact_fallback_queries"] for v in per.values()),
        "by_kind": {"dense_queries_per_s": qps,
                    "queries_to_second_pass": {k_: v["dense"]["second_pass_queries"] for k_, v in per.items()},
                    "queries_to_exact_fallback": {k_: v["dense"]["exact_fallback_queries"] for k_, v in per.items()},
                    "candidates_per_query": {k_: v["dense"]["filter_candidates_per_query"] for k_, v in per.items()}},
        "per_kind": per}
    if with_rq:
        rqm = {k_: {s_: r["ms"] for s_, r in v["rq_trained_codebook"].items()} for k_, v in per.items() if "rq_trained_codebook" in v}
        out["by_kind"]["rq_ms_trained"] = rqm
        for shape in sorted({s_ for v in rqm.values() for s_ in v}):
            ms = [v[shape] for v in rqm.values() if shape in v]
            out["rq_%s_trained_ms_min_max" % shape] = [min(ms), max(ms)]
        out["rq_all_codes_identical"] = all(r["codes_identical_to_exact_mode"] for v in per.values()
                                            for r in v.get("rq_trained_codebook", {}).values())
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--docs", type=int, default=8_841_823)
    ap.add_argument("--queries", type=int, default=6980)
    ap.add_argument("--kinds", default=",".join(synth.CORPUS_KINDS))
    ap.add_argument("--no-rq", action="store_true")
    ap.add_argument("--out", default=None)
    args = ap.parse_args()
    hip.require_gpu()
    dev = torch.device("cuda", 0)
    res = sweep(dev, args.docs, args.queries, tuple(args.kinds.split(",")), with_rq=not args.no_rq)
    res["device"] = torch.cuda.get_device_name(0)
    txt = json.dumps(res, indent=1)
    if args.out:
        os.makedirs(os.path.dirname(os.path.abspath(args.out)), exist_ok=True)
        with open(args.out, "w") as f:
            f.write(txt)
    print(json.dumps({k_: v for k_, v in res.items() if k_ != "per_kind"}))
    for kind, v in res["per_kind"].items():
        print(kind, json.dumps(v["dense"]))
        for s_, r in v.get("rq_trained_codebook", {}).items():
            print(kind, "rq", s_, json.dumps(r))


if __name__ == "__main__":
    main()
