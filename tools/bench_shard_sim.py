#!/usr/bin/env python3
"""Per-rank cost of the row-sharded dense search, simulated on ONE GPU: rank 0's shard of a world-W job at C2 sizes,
searched with the first-round list length (dense.truncated_list_len) and with full k, plus what follows the local search
on every rank -- packing the (score, id) pairs into the all-gather payload, unpacking W of them and merge_truncated with
its proof (the real payload shape: W x nq x k_local packed i64; rank 0's own list stands in for the peers').

    python tools/bench_shard_sim.py [out.json]            # W in {1, 2, 4, 8}; efficiency = T(1) / (W x (search + post))
    SWEEP=1 python tools/bench_shard_sim.py               # also sweeps the chunk schedule knobs at W = 8 and W = 1

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

t1 = rows[0]["per_rank_total_ms"]
for r in rows:
    r["ideal_ms"] = t1 / r["world"]
    r["efficiency"] = t1 / (r["world"] * r["per_rank_total_ms"])
summary = {"what": "per-rank cost of dense.sharded_ip_topk on one MI355X simulating rank 0 of W (C2: 6980 x 768 queries, "
                   "8,841,823 x 768 docs, top-1000); efficiency = T(W=1) / (W x (local search + pack/unpack/merge/proof)); the "
                   "all-gather's wire time is not included (1-GPU box): see allgather_xgmi_floor_ms",
           "device": torch.cuda.get_device_name(0), "reps": REPS, "rows": rows}
if sweep:
    summary["schedule_sweep"] = sweep
print(json.dumps({"efficiency": {r["world"]: round(r["efficiency"], 4) for r in rows}}), flush=True)
if len(sys.argv) > 1:
    with open(sys.argv[1], "w") as f:
        json.dump(summary, f, indent=1)
