"""BASELINE.json full size (config C2: 6980 x 768 queries x 8,841,823 x 768 docs, top-1000) on one MI355X,
checked through size-independent properties (the oracle cannot finish this size in seconds):
  * every list is ordered by (score desc, id asc) and holds distinct in-range ids,
  * the planted neighbour of every query is its rank-1 hit,
  * shard-count invariance: 8 row shards searched with global ids + merge == the un-sharded search, bit for bit,
  * returned scores are the exact fmaf chains (pair_dot re-computation, bit for bit),
  * completeness against an independent implementation: for sampled queries no document outside the
    returned list scores above the list's k-th score (torch / hipBLAS matvec, f32 rounding tolerance).
"""
import numpy as np
import pytest
import torch

import bench
from mevi_amd import dense, ops

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def c2(cuda):
    free, _ = torch.cuda.mem_get_info()
    if free < 40e9:
        pytest.skip("needs ~30 GB of HBM")
    docs = bench.gen_shard(0, bench.N_DOCS, cuda, bench.N_DOCS)
    query = bench.gen_queries(bench.N_QUERIES, cuda, bench.N_DOCS)
    s, i = dense.ip_topk(query, docs, bench.TOPK)
    torch.cuda.synchronize()
    return docs, query, s, i


def test_lists_are_ordered_distinct_and_in_range(c2):
    _, _, s, i = c2
    assert bool((i >= 0).all()) and bool((i < bench.N_DOCS).all())
    ds = s[:, 1:] - s[:, :-1]
    assert bool((ds <= 0).all())
    ties = ds == 0
    assert bool((i[:, 1:][ties] > i[:, :-1][ties]).all())
    srt = torch.sort(i, dim=1).values
    assert bool((srt[:, 1:] != srt[:, :-1]).all())


def test_planted_neighbour_is_rank_one(c2):
    _, _, _, i = c2
    assert np.array_equal(i[:, 0].cpu().numpy(), bench.planted_ids(bench.N_QUERIES, bench.N_DOCS))


def test_shard_count_invariance_at_full_size(c2):
    docs, query, s, i = c2
    parts_s, parts_i = [], []
    for r in range(8):
        a, b = dense.shard_range(bench.N_DOCS, r, 8)
        ps, pi = dense.ip_topk(query, docs[a:b], bench.TOPK, id_offset=a)
        parts_s.append(ps)
        parts_i.append(pi)
    ms, mi = dense.topk_merge(torch.stack(parts_s), torch.stack(parts_i), bench.TOPK)
    assert torch.equal(mi, i) and torch.equal(ms.view(torch.int32), s.view(torch.int32))


def test_indexed_prefilter_equals_exact_path_at_full_size(c2):
    """The f16 pre-filtered search must return the exact-f32 search's lists bit for bit, with (nearly)
    every query proven by the error bound rather than by the fallback."""
    from mevi_amd import hip

    docs, query, s, i = c2
    index = dense.DenseIndex(docs)
    s2, i2 = index.search(query, bench.TOPK)
    st = hip.IpTopkStats()
    hip.lib().mevi_ip_topk_get_stats(st)
    assert torch.equal(i2, i) and torch.equal(s2.view(torch.int32), s.view(torch.int32))
    assert st.n_failed_queries <= 0.01 * bench.N_QUERIES
    assert st.max_err_ratio <= st.err_bound / 4      # the measured-rounding bound of round 6 (observed 0.06 of the old worst-case bound)


def test_scores_are_exact_chains_and_lists_are_complete(c2):
    docs, query, s, i = c2
    rows = torch.tensor([0, 17, 3333, 6979], device=docs.device)
    ia = rows.repeat_interleave(bench.TOPK)
    again = ops.pair_dot(query, ia, docs, i[rows].reshape(-1)).view(len(rows), bench.TOPK)
    assert torch.equal(again.view(torch.int32), s[rows].view(torch.int32))
    for r in rows.tolist():
        full = docs @ query[r]                        # independent implementation (hipBLAS), f32 rounding differs
        kth = s[r, -1]
        full[i[r]] = -float("inf")
        assert float(full.max()) <= float(kth) + 2e-5 * max(1.0, abs(float(kth)))


def _bench_two_ranks(backend_env, extra=("--docs", "700000", "--queries", "600", "--no-cpu-baseline", "--no-seq2seq-legs")):
    import json
    import os
    import socket
    import subprocess
    import sys

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    import tempfile

    detail = os.path.join(tempfile.mkdtemp(), "detail.json")       # the full record (per-rank entries) beside the compact line
    env = dict(os.environ, PYTHONPATH=root, MEVI_BENCH_DETAIL=detail, **backend_env)
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr",
                        "127.0.0.1", "--master-port", str(port), os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "2",
                        "--warmup", "1", *extra],
                       capture_output=True, text=True, env=env, timeout=900)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]                      # rank 0 prints ONE JSON line
    line = json.loads(lines[0])
    assert len(lines[0]) < 8192 and line["n_gpus"] == 2 and "multi_gpu" in line
    with open(detail) as f:
        full = json.load(f)
    assert full["value"] == pytest.approx(line["value"], rel=1e-4)
    return full


def _check_two_rank_line(d, rows=700000):
    assert d["n_gpus"] == 2 and d["config"]["planted_top1_ok"] == 1.0 and d["value"] > 0
    m = d["multi_gpu"]
    assert m["rounds"] >= 1 and len(m["per_rank"]) == 2 and {r["rank"] for r in m["per_rank"]} == {0, 1}
    assert sum(r["shard_rows"] for r in m["per_rank"]) == rows
    assert m["local_search_ms"]["max"] >= m["local_search_ms"]["min"] > 0 and m["all_gather_ms"] >= 0 and m["merge_ms"] > 0
    assert d["roofline"]["launches"] > 0 and d["roofline"]["achieved"] > 0


def test_bench_launch_line_with_two_ranks_sharing_the_device():
    """The driver's multi-GPU launch line (torch.distributed.run ... bench.py --gpus 2) with both ranks on this GPU and the
    collectives over gloo: the sharded search, its per-rank phase breakdown and the single JSON line -- a rehearsal of
    the N > 1 path, not a timing."""
    _check_two_rank_line(_bench_two_ranks({"MEVI_BENCH_BACKEND": "gloo"}))


def test_bench_c5_chain_rehearsal_with_two_ranks_sharing_the_device():
    """N > 1 runs C5, not C2-sharded: after the timed dense steps every rank runs the chain -- dense arm sharded,
    NCI beam search + tower + fine stage as replicas over the rank's DistributedSampler slice, full corpus per rank -- and
    rank 0's line carries `chain_c5`.  Two ranks on this GPU over gloo, reduced corpus, 101 queries (odd: rank 1 pads)."""
    free, _ = torch.cuda.mem_get_info()
    if free < 60e9:
        pytest.skip("two t5-base model replicas + prefix tables: needs ~40 GB of HBM")
    d = _bench_two_ranks({"MEVI_BENCH_BACKEND": "gloo"}, ("--docs", "300000", "--queries", "101", "--no-cpu-baseline"))
    _check_two_rank_line(d, rows=300000)
    assert "chain_c5_error" not in d, d.get("chain_c5_error")
    c = d["chain_c5"]
    assert c["chain_ms"] > 0 and c["queries_per_s"] > 0 and len(c["per_rank"]) == 2
    assert c["dense_lists_identical_on_all_ranks"] and c["doc_codes_identical_on_all_ranks"]
    assert sum(r["queries"] for r in c["per_rank"]) == 101 and [r["queries_with_padding"] for r in c["per_rank"]] == [51, 51]
    assert set(c["stage_ms_max_over_ranks"]) >= {"tower", "query_all_gather", "dense_top1000_sharded", "nci_beam_search",
                                                 "tower_again", "fine_stage"}
    assert set(c["mrr10"]) == {"dense", "fine", "ensemble", "ensemble_alpha20"} and 0.0 <= c["mrr10"]["ensemble"] <= 1.0
    assert "REHEARSAL" in c["workload"]
    f = c["seq2seq_frac_per_rank"]                                  # each rank's share of the seq2seq arm against the matrix peak
    assert len(f["nci"]) == 2 and len(f["tower"]) == 2 and all(0.0 < v < 1.0 for v in f["nci"] + f["tower"])
    assert d["config"]["chain_c5"]["seq2seq_frac_per_rank"] == f


def test_bench_starts_its_own_ranks_from_the_plain_command():
    """`python bench.py --gpus 2` with no launcher around it: the process starts its two ranks itself before
    touching the GPU (as MEVI/main.py:286-298 spawns its workers), relays rank 0's ONE line -- with the C5 chain in it --
    and returns the child's exit status."""
    import json
    import os
    import subprocess
    import sys

    free, _ = torch.cuda.mem_get_info()
    if free < 60e9:
        pytest.skip("two t5-base model replicas + prefix tables: needs ~40 GB of HBM")
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    env.update(PYTHONPATH=root, MEVI_BENCH_BACKEND="gloo")
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1",
                        "--docs", "300000", "--queries", "101", "--no-cpu-baseline"],
                       capture_output=True, text=True, env=env, timeout=900)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]
    assert len(lines[0]) < 8192                                   # the whole line fits the driver's stdout tail
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["config"]["planted_top1_ok"] == 1.0 and d["value"] > 0
    assert "chain_c5_error" not in d and d["chain_c5"]["queries_per_s"] > 0
    assert d["config"]["chain_c5"]["dense_lists_identical_on_all_ranks"] is True
    # a wrong world size is an error, not a silent single-rank run
    bad = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2"], capture_output=True, text=True,
                         env=dict(env, WORLD_SIZE="1", RANK="0", LOCAL_RANK="0"), timeout=300)
    assert bad.returncode != 0 and "WORLD_SIZE" in bad.stderr


def test_bench_over_rccl_when_two_gpus_are_visible():
    """The same line over the real `nccl` (RCCL) backend, one rank per GPU -- runs wherever two devices are visible."""
    if torch.cuda.device_count() < 2:
        pytest.skip("one GPU visible: the RCCL path is exercised by the driver's 8-GPU run")
    d = _bench_two_ranks({})
    _check_two_rank_line(d)
    assert "REHEARSAL" not in d["config"]["parallelism"]
