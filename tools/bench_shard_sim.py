#!/usr/bin/env python3
"""Per-rank cost of the row-sharded dense search, simulated on ONE GPU: rank 0's shard of a world-W job at C2 sizes,
searched with the first-round list length (dense.truncated_list_len) and with full k, plus what follows the local search
on every rank -- packing the (score, id) pairs into the all-gather payload, unpacking W of them and merge_truncated with
its proof (the real payload shape: W x nq x k_local packed i64; rank 0's own list stands in for the peers').

    python tools/bench_shard_sim.py [out.json]            # W in {1, 2, 4, 8}; efficiency = T(1) / (W x (search + post))
    SWEEP=1 python tools/bench_shard_sim.py               # also sweeps the chunk schedule knobs at W = 8 and W = 1
    python tools/bench_shard_sim.py out.json --layout sorted   # + the adversarial layout: the corpus sorted by its score against
                                                          #   the mean query, so that shard 0 holds most of every top-k and the
                                                          #   truncated first round fails (second round: full-k lists)
    python tools/bench_shard_sim.py out.json --predict-c5 # + the seq2seq legs at 6980 / 873 queries: predicted 8-GPU chain

The all-gather itself cannot be measured on a 1-GPU box: `allgather_payload_bytes` / (7 links x ~153 GB/s) is quoted as the
xGMI floor.  VERDICT r2 #1 target: efficiency >= 0.94 at W = 8."""
import json
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from mevi_amd import dense, hip  # noqa: E402

dev = torch.device("cuda", 0)
nq, k = bench.N_QUERIES, bench.TOPK
query = bench.gen_queries(nq, dev, bench.N_DOCS)
REPS = int(os.environ.get("REPS", "5"))


def timed(fn, reps=REPS):
    fn()
    torch.cuda.synchronize()
    t = time.perf_counter()
    for _ in range(reps):
        out = fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t) / reps * 1e3, out


def post_exchange(s, i, world):
    """What a rank does after its local search, minus the wire time: pack, (gather), merge + proof -- the calls of
    dense.sharded_ip_topk (mevi_pack_lists_i64, mevi_topk_merge_packed_f32)."""
    packed = dense.pack_lists(s, i)
    gathered = packed.unsqueeze(0).expand(world, -1, -1).contiguous()           # stands in for all_gather_into_tensor
    return dense.merge_packed(gathered, k)


def post_exchange_unfused(s, i, world):       # round 2's form: torch pack / unpack + full-sort merge + torch proof
    packed = (s.contiguous().view(torch.int32).to(torch.int64) << 32) | (i & 0xFFFFFFFF)
    gathered = packed.unsqueeze(0).expand(world, -1, -1).contiguous()
    return dense.merge_truncated(*dense.unpack_lists(gathered), k)


rows = []
sweep = []
for world in (1, 2, 4, 8):
    a, b = dense.shard_range(bench.N_DOCS, 0, world)
    docs = bench.gen_shard(a, b, dev, bench.N_DOCS)
    index = dense.DenseIndex(docs)
    kl = dense.truncated_list_len(k, world)
    out = {"world": world, "shard_rows": b - a, "k_local": kl}
    out["search_ms"], (s, i) = timed(lambda: index.search(query, kl, id_offset=a))
    st = hip.IpTopkStats()
    hip.lib().mevi_ip_topk_get_stats(st)
    out["filter_launches"] = int(st.n_chunks)
    out["unproven_queries"] = int(st.n_failed_queries) + int(st.n_second_pass_queries)
    if world > 1:
        out["search_full_k_ms"], _ = timed(lambda: index.search(query, k, id_offset=a), 3)
        out["post_exchange_ms"], (ms_, mi_, un_) = timed(lambda: post_exchange(s, i, world))
        out["post_exchange_unfused_ms"], (ms2, mi2, un2) = timed(lambda: post_exchange_unfused(s, i, world))
        out["fused_merge_equals_unfused"] = bool(torch.equal(mi_, mi2) and torch.equal(ms_.view(torch.int32), ms2.view(torch.int32))
                                                 and torch.equal(un_, un2))
        out["allgather_payload_bytes_per_rank"] = nq * kl * 8
        out["allgather_xgmi_floor_ms"] = round((world - 1) * nq * kl * 8 / (7 * 153e9) * 1e3, 4)
    else:
        out["post_exchange_ms"] = 0.0
    out["per_rank_total_ms"] = out["search_ms"] + out["post_exchange_ms"]
    rows.append(out)
    print(json.dumps(out), flush=True)
    if os.environ.get("SWEEP") == "1" and world in (1, 8):
        for slots in ("1024", "2048", "4096", "8192"):
            for div in ("3", "2", "1.5", "1.25"):
                os.environ["MEVI_IP_TOPK_MIN_SLOTS"], os.environ["MEVI_IP_TOPK_GROWTH_DIV"] = slots, div
                ms, _ = timed(lambda: index.search(query, kl, id_offset=a), 3)
                hip.lib().mevi_ip_topk_get_stats(st)
                r = {"world": world, "min_slots": int(slots), "growth_div": float(div), "search_ms": round(ms, 3),
                     "launches": int(st.n_chunks), "unproven": int(st.n_failed_queries) + int(st.n_second_pass_queries)}
                sweep.append(r)
                print(json.dumps(r), flush=True)
        del os.environ["MEVI_IP_TOPK_MIN_SLOTS"], os.environ["MEVI_IP_TOPK_GROWTH_DIV"]
    del index, docs
    torch.cuda.empty_cache()

def all_shards(world, corpus=None, tag="bench"):
    """The W-way search with EVERY shard searched in turn on this device (one resident at a time): the true first-round lists
    of all ranks, the merge's proof, the second round it asks for -- per-rank times of both rounds, max over ranks."""
    per_rank, lists = [], []
    kl = dense.truncated_list_len(k, world)
    for r in range(world):
        a, b = dense.shard_range(bench.N_DOCS, r, world)
        docs = corpus[a:b] if corpus is not None else bench.gen_shard(a, b, dev, bench.N_DOCS)
        index = dense.DenseIndex(docs)
        ms, (s, i) = timed(lambda: index.search(query, kl, id_offset=a), 3)
        lists.append(dense.pack_lists(s, i))
        per_rank.append({"rank": r, "round1_search_ms": round(ms, 3)})
        del index, docs
    gathered = torch.stack(lists)
    post_ms, (ms_, mi_, unproven) = timed(lambda: dense.merge_packed(gathered, k), 3)
    q2 = torch.nonzero(unproven).view(-1)
    out = {"layout": tag, "world": world, "k_local": kl, "round1_search_ms_max": max(p["round1_search_ms"] for p in per_rank),
           "round1_post_ms": round(post_ms, 3), "second_round_queries": int(q2.numel())}
    if q2.numel():
        qq = query[q2].contiguous()
        lists2 = []
        for r in range(world):
            a, b = dense.shard_range(bench.N_DOCS, r, world)
            docs = corpus[a:b] if corpus is not None else bench.gen_shard(a, b, dev, bench.N_DOCS)
            index = dense.DenseIndex(docs)
            ms, (s, i) = timed(lambda: index.search(qq, k, id_offset=a), 3)
            lists2.append(dense.pack_lists(s, i))
            per_rank[r]["round2_search_ms"] = round(ms, 3)
            del index, docs
        g2 = torch.stack(lists2)
        post2, (ms2, mi2, un2) = timed(lambda: dense.merge_packed(g2, k), 3)
        out["round2_search_ms_max"] = max(p["round2_search_ms"] for p in per_rank)
        out["round2_post_ms"] = round(post2, 3)
        out["round2_queries_per_s_per_rank"] = round(q2.numel() / out["round2_search_ms_max"] * 1e3, 1)
        out["unproven_after_round2"] = int(un2.sum().item())
        out["allgather_round2_bytes_per_rank"] = int(q2.numel()) * k * 8
        ms_[q2], mi_[q2] = ms2, mi2
    out["per_rank_total_ms"] = out["round1_search_ms_max"] + out["round1_post_ms"] + out.get("round2_search_ms_max", 0.0) + out.get("round2_post_ms", 0.0)
    out["per_rank"] = per_rank
    return out, (ms_, mi_)


extra = {}
if "--layout" in sys.argv and sys.argv[sys.argv.index("--layout") + 1] == "sorted":
    corpus = bench.gen_shard(0, bench.N_DOCS, dev, bench.N_DOCS)
    ref_s, ref_i = dense.DenseIndex(corpus).search(query, k)                       # the un-sharded answer (ids of the ORIGINAL layout)
    order = torch.argsort(corpus @ query.mean(0), descending=True)                # rows most like the mean query first: shard 0 is hot
    corpus = corpus[order].contiguous()
    inv = torch.empty_like(order)
    inv[order] = torch.arange(order.numel(), device=dev)
    lay = []
    for world in (2, 4, 8):
        o, (ms_, mi_) = all_shards(world, corpus, "sorted by score against the mean query (shard 0 holds most of every top-k)")
        got = torch.sort(order[mi_.clamp_min(0)], 1)[0]                        # ids mapped back to the original layout
        o["same_documents_as_unsharded"] = bool(torch.equal(got, torch.sort(ref_i, 1)[0]))
        o["rows_with_other_documents"] = int((got != torch.sort(ref_i, 1)[0]).any(1).sum().item())
        o["scores_bit_equal_to_unsharded"] = bool(torch.equal(ms_.view(torch.int32), ref_s.view(torch.int32)))
        o["note"] = ("rows_with_other_documents: exact score ties at the k-th position -- a tie goes to the lower id, and ids are "
                     "positions in the layout; the score lists are bit-equal")
        o["efficiency_vs_w1_bench_layout"] = rows[0]["per_rank_total_ms"] / (world * o["per_rank_total_ms"])
        lay.append(o)
        print(json.dumps({k_: v for k_, v in o.items() if k_ != "per_rank"}), flush=True)
    extra["sorted_layout"] = lay
    del corpus, order, inv
    torch.cuda.empty_cache()
if "--all-shards" in sys.argv or "--predict-c5" in sys.argv:
    o, _ = all_shards(8)
    extra["bench_layout_all_8_shards"] = o
    print(json.dumps({k_: v for k_, v in o.items() if k_ != "per_rank"}), flush=True)

t1 = rows[0]["per_rank_total_ms"]
for r in rows:
    r["ideal_ms"] = t1 / r["world"]
    r["efficiency"] = t1 / (r["world"] * r["per_rank_total_ms"])
summary = {"what": "per-rank cost of dense.sharded_ip_topk on one MI355X simulating rank 0 of W (C2: 6980 x 768 queries, "
                   "8,841,823 x 768 docs, top-1000); efficiency = T(W=1) / (W x (local search + pack/unpack/merge/proof)); the "
                   "all-gather's wire time is not included (1-GPU box): see allgather_xgmi_floor_ms",
           "device": torch.cuda.get_device_name(0), "reps": REPS, "rows": rows}
summary.update(extra)
if "--predict-c5" in sys.argv:
    # BASELINE.json configs[4] predicted from one GPU: what a rank of 8 runs is the dense arm on its shard (above) and the seq2seq
    # arm + both tower passes + the fine stage on ITS 873 queries (DistributedSampler, MEVI/main.py:318-322)
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    import numpy as np
    import batch_sweep
    import synth

    torch.cuda.empty_cache()
    model, tower, _, _ = synth.build(dev, 4, 32, None)
    ids, mask = synth.query_ids(nq, dev, np.random.default_rng(0))
    model.generate(ids, mask, num_beams=10)
    # per-rank query counts of the replica layout (8 x 873) and of the ARM-SPLIT layouts: the beam search on `a`
    # ranks (ceil(nq / a) queries each), both tower passes on the other 8 - a, concurrently
    W8 = 8
    per = lambda parts: (nq + parts - 1) // parts       # noqa: E731
    split_sizes = sorted({per(a) for a in range(2, W8 - 1)} | {per(W8 - a) for a in range(2, W8 - 1)})
    sw = {r["queries"]: r for r in batch_sweep.sweep(model, tower, ids, mask, 4, 32, 10, tuple(sorted(set(split_sizes) | {873, nq})),
                                                    bench.seq2seq_flops, bench.tower_flops)}
    fine_ms_1 = float(os.environ.get("FINE_MS", "2.7"))               # fine stage of 6980 queries (chain_c4.stage_ms.fine_stage)
    w8 = next(r for r in rows if r["world"] == 8)
    dense8 = extra.get("bench_layout_all_8_shards", {}).get("per_rank_total_ms", w8["per_rank_total_ms"]) + w8["allgather_xgmi_floor_ms"]
    one = 2 * sw[nq]["tower_ms"] + rows[0]["per_rank_total_ms"] + sw[nq]["nci_ms"] + fine_ms_1
    eight = 2 * sw[873]["tower_ms"] + dense8 + sw[873]["nci_ms"] + fine_ms_1 / 8
    summary["predicted_c5"] = {
        "what": "chain per rank at 8 GPUs = 2 x tower(873 queries) + sharded dense search of all 6980 queries on 1/8 of the corpus "
                "(+ the all-gather's xGMI floor) + beam search(873) + fine stage / 8, against the same sum at 1 GPU; all legs measured "
                "in this process on one MI355X.  To be falsified by the first SCALE run.",
        "one_gpu": {"tower_ms": sw[nq]["tower_ms"], "dense_ms": rows[0]["per_rank_total_ms"], "nci_ms": sw[nq]["nci_ms"], "fine_ms": fine_ms_1,
                    "chain_ms": round(one, 2), "queries_per_s": round(nq / one * 1e3, 1)},
        "eight_gpus_per_rank": {"tower_ms": sw[873]["tower_ms"], "dense_ms": round(dense8, 3), "nci_ms": sw[873]["nci_ms"],
                                "nci_frac": sw[873].get("nci_frac"), "tower_frac": sw[873].get("tower_frac"),
                                "fine_ms": round(fine_ms_1 / 8, 3), "chain_ms": round(eight, 2), "queries_per_s": round(nq / eight * 1e3, 1)},
        "predicted_efficiency": {"dense_arm": round(rows[0]["per_rank_total_ms"] / (8 * dense8), 4),
                                 "target_dense_arm": "within 15 % of linear (north_star): >= 0.85",
                                 "tower": round(sw[nq]["tower_ms"] / (8 * sw[873]["tower_ms"]), 4),
                                 "beam_search": round(sw[nq]["nci_ms"] / (8 * sw[873]["nci_ms"]), 4),
                                 "whole_chain": round(one / (8 * eight), 4)}}
    # ---- arm-split layouts: every rank searches its corpus shard for all queries (as above); the `a` beam-search ranks start the
    # search at t = 0 (it needs no embedding), the 8 - a tower ranks run the first tower pass, all ranks exchange the embeddings
    # (all-gather, 21 MB) and search their shards, the tower ranks run the second pass; beams (nq x R x M ints) and embeddings
    # meet on the tower ranks for the fine stage.  A rank's time = its own arm + the dense search; the layout's = the slowest rank.
    layouts = [{"layout": "replicas 8 x %d queries (MEVI/main.py:318-322: what chain_c5 runs)" % per(W8), "nci_ranks": W8, "tower_ranks": W8,
                "nci_rank_ms": round(eight, 2), "tower_rank_ms": round(eight, 2), "chain_ms": round(eight, 2)}]
    for a in range(2, W8 - 1):
        nn, tt = per(a), per(W8 - a)
        nci_rank = sw[nn]["nci_ms"] + dense8
        tower_rank = 2 * sw[tt]["tower_ms"] + dense8 + fine_ms_1 / (W8 - a)
        layouts.append({"layout": "arm split: beam search on %d ranks x %d queries, both tower passes on %d ranks x %d" % (a, nn, W8 - a, tt),
                        "nci_ranks": a, "tower_ranks": W8 - a, "nci_ms": sw[nn]["nci_ms"], "nci_frac": sw[nn].get("nci_frac"),
                        "tower_ms": sw[tt]["tower_ms"], "tower_frac": sw[tt].get("tower_frac"),
                        "nci_rank_ms": round(nci_rank, 2), "tower_rank_ms": round(tower_rank, 2), "chain_ms": round(max(nci_rank, tower_rank), 2)})
    best = min(layouts, key=lambda l: l["chain_ms"])
    summary["predicted_c5"]["layouts"] = layouts
    summary["predicted_c5"]["best_layout"] = {"layout": best["layout"], "chain_ms": best["chain_ms"],
                                              "queries_per_s": round(nq / best["chain_ms"] * 1e3, 1),
                                              "gain_over_replicas": round(eight / best["chain_ms"], 4),
                                              "whole_chain_efficiency": round(one / (8 * best["chain_ms"]), 4)}
    print(json.dumps(summary["predicted_c5"]), flush=True)
if sweep:
    summary["schedule_sweep"] = sweep
print(json.dumps({"efficiency": {r["world"]: round(r["efficiency"], 4) for r in rows}}), flush=True)
if len(sys.argv) > 1:
    with open(sys.argv[1], "w") as f:
        json.dump(summary, f, indent=1)
