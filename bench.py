#!/usr/bin/env python3
"""bench.py -- headline benchmark of the MEVI hot path on MI355X.

Workload (BASELINE.json configs[1], "C2"): the dense arm of MEVI, i.e. what
`faiss_search.py --param Flat` computes for MSMARCO dev: 6980 x 768 f32 queries
against the 8,841,823 x 768 f32 passage-embedding matrix (27.16 GB, resident in
HBM), exact inner-product top-1000.  One "step" = one full search of all queries.

  python bench.py [--gpus N] [--steps K] [--warmup W]
  python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N ...

N > 1: the corpus is row-sharded (ceil(N_docs/N) rows per rank), queries are
replicated, every rank searches its shard with global ids, one RCCL all-gather of
the per-shard top-k + a device merge (mevi_amd.dense.sharded_ip_topk).  Total
work is fixed => "scaling": "strong".  Rank 0 prints ONE JSON line.

Synthetic data (there is no network for MSMARCO): corpus rows ~ 0.05*N(0,1) + 0.02
generated on device from per-block seeds, queries = planted neighbours of rows in
block 0 (so rank-1 correctness is checkable without the oracle).
"""
import argparse
import json
import os
import sys
import time

import numpy as np
import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

from mevi_amd import dense, hip  # noqa: E402

N_DOCS = 8_841_823   # MEVI/marco_eval_nci_rq.sh:26 (--save_hard_neg = corpus size)
N_QUERIES = 6980     # MSMARCO dev
DIM = 768
TOPK = 1000          # MEVI/faiss_search.py:88
BLOCK = 65536        # rows per RNG block (seed = 10_000 + block index)
PEAK_F32_MFMA_TFLOPS = 157.3   # /opt/skills/guides/MI355X_MICROARCH.md, "Peak FP32 (matrix)"
PEAK_F16_MFMA_TFLOPS = 2500.0  # same guide, "BF16/F16 ~2.5 PF dense"


def gen_block(b, device, n_docs):
    rows = min(BLOCK, n_docs - b * BLOCK)
    g = torch.Generator(device=device).manual_seed(10_000 + b)
    return 0.05 * torch.randn((rows, DIM), device=device, generator=g) + 0.02


def gen_shard(start, end, device, n_docs):
    out = torch.empty((end - start, DIM), dtype=torch.float32, device=device)
    b = start // BLOCK
    while b * BLOCK < end:
        blk = gen_block(b, device, n_docs)
        lo = max(start, b * BLOCK)
        hi = min(end, b * BLOCK + blk.shape[0])
        out[lo - start:hi - start] = blk[lo - b * BLOCK:hi - b * BLOCK]
        b += 1
    return out


def planted_ids(nq, n_docs):
    return (np.arange(nq, dtype=np.int64) * 9) % min(BLOCK, n_docs)


def gen_queries(nq, device, n_docs):
    blk0 = gen_block(0, device, n_docs)
    gid = torch.from_numpy(planted_ids(nq, n_docs)).to(device)
    g = torch.Generator(device=device).manual_seed(1234)
    return (blk0[gid] + 0.005 * torch.randn((nq, DIM), device=device, generator=g)).contiguous()


def cpu_baseline(n_docs, nq_full, target_s=60.0):
    """faiss-Flat-style CPU evaluation (BLAS sgemm blocks + per-query heaps = oracle.dense.ip_topk_blas)
    on a bounded sample, scaled linearly in rows to the full corpus."""
    from oracle import dense as odense

    nd_s = min(n_docs, 500_000)
    rng = np.random.default_rng(7)
    d = (0.05 * rng.standard_normal((nd_s, DIM), dtype=np.float32) + 0.02).astype(np.float32)
    q_all = (0.05 * rng.standard_normal((nq_full, DIM), dtype=np.float32) + 0.02).astype(np.float32)
    odense.ip_topk_blas(q_all[:32], d[:50_000], TOPK)      # BLAS / OpenMP thread pools up before anything is timed
    t = time.time()
    odense.ip_topk_blas(q_all[:128], d, TOPK)
    cal = time.time() - t
    nq_s = int(min(nq_full, max(128, 128 * target_s / max(cal, 1e-3))))
    t = time.time()
    odense.ip_topk_blas(q_all[:nq_s], d, TOPK)
    dt = time.time() - t
    qps_sample = nq_s / dt
    qps_full = qps_sample * nd_s / n_docs
    try:
        import threadpoolctl
        cores = max([p.get("num_threads", 1) for p in threadpoolctl.threadpool_info()] or [1])
    except Exception:
        cores = os.cpu_count()
    return {
        "value": qps_full, "unit": "queries/s", "cores": int(cores), "kind": "port",
        "sample": f"{nq_s} queries x {nd_s} docs x {DIM} f32, top-{TOPK}, BLAS sgemm blocks + per-query heaps "
                  f"(faiss Flat-IP algorithm) took {dt:.2f}s; scaled x{nd_s}/{n_docs} rows to the full corpus",
        "host_cpus": os.cpu_count(),
    }


def seq2seq_legs(device, nq, search_ms, with_cpu):
    """Untimed extras (N = 1), measured after the timed region and not part of `value`: the stages around the dense
    search at t5-base shapes (synthetic weights, MS MARCO-like query lengths, tools/synth.py) --
      * `generate.py --gen_query`: the T5-ANCE query tower (12 + 12 layers) over the same number of queries,
      * `main.py --mode eval`: NCI generate, beams 10, RQ (4,32) (12 + 6 layers, 4 adaptor layers, adaptive head);
    and, with the CPU baseline on, the oracle's torch-fp32 restatement of both (what §8(c) validated against the
    reference) on the box's host cores for a small sample of the same queries, with the agreement of the two paths on
    that sample."""
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import synth
    from mevi_amd import nci, t5

    M, K, R, gen_batch = 4, 32, 10, 8192
    W, TW, _, _ = synth.weights(device, M, K)
    cpu_w = ({k: v.cpu() for k, v in W.items()}, {k: v.cpu() for k, v in TW.items()}) if with_cpu else None
    model = nci.NCIModel(W, device=device, M=M, K=K, adaptor_layer_num=4, num_layers=12, num_decoder_layers=6)
    tower = t5.TwinTower(TW, device=device, num_layers=12, num_decoder_layers=12)
    del W, TW
    ids, mask = synth.query_ids(nq, device, np.random.default_rng(0))

    def timed(fn, reps):
        fn()
        torch.cuda.synchronize()
        t = time.perf_counter()
        for _ in range(reps):
            out = fn()
        torch.cuda.synchronize()
        return (time.perf_counter() - t) / reps * 1e3, out

    def gen_all():
        return [model.generate(ids[a:a + gen_batch], mask[a:a + gen_batch], num_beams=R) for a in range(0, nq, gen_batch)]

    tower_ms, qemb = timed(lambda: tower.encode_query({"input_ids": ids, "attention_mask": mask}), 3)
    nci_ms, gen = timed(gen_all, 1)
    out = {
        "dense_arm_with_tower": {
            "tower_ms": tower_ms, "tower_queries_per_s": nq / tower_ms * 1e3, "search_ms": search_ms,
            "queries_per_s": nq / (tower_ms + search_ms) * 1e3, "pass_tokens": t5.DEVICE_PASS_TOKENS,
            "note": "generate.py --gen_query (T5-ANCE tower, t5-base shapes, f32, synthetic weights) + faiss_search.py"},
        "seq2seq_arm": {
            "nci_generate_ms": nci_ms, "nci_generate_queries_per_s": nq / nci_ms * 1e3, "beams": R, "rq": [M, K],
            "queries_per_pass": gen_batch,
            "note": "main.py --mode eval beam search (t5-base NCI model, f32, synthetic weights); the fine stage adds "
                    "the tower again + a gather-dot (tools/bench_chain.py, profiles/r01_chain_c4.txt)"},
    }
    chain_ms = tower_ms + search_ms + nci_ms + tower_ms
    out["chain_c4_derived"] = {
        "ms": chain_ms, "queries_per_s": nq / chain_ms * 1e3,
        "queries_per_s_reusing_query_embeddings": nq / (chain_ms - tower_ms) * 1e3,
        "note": "tower + dense search + NCI beam search + tower again (as the reference's fine stage does) from the legs of "
                "this run; the gather-dot of the fine stage adds 3-4 ms on the C4 corpus (tools/bench_chain.py measures the "
                "chain with it: profiles/r01_chain_c4.txt)"}
    if with_cpu:
        from oracle import t5 as ot5

        ncfg, tcfg = synth.oracle_cfgs(M, K)
        n_t, n_g = min(nq, 64), min(nq, 8)
        ci, cm = ids.cpu(), mask.cpu()
        with torch.no_grad():
            t = time.time()
            ref_e = ot5.tower_encode(cpu_w[1], tcfg, ci[:n_t], cm[:n_t])
            t_tower = time.time() - t
            t = time.time()
            ref_dec, ref_sc, _ = ot5.nci_generate(cpu_w[0], ncfg, ci[:n_g], cm[:n_g], R)
            t_gen = time.time() - t
        dec = gen[0][0][:n_g * R].cpu()
        sc = np.asarray(gen[0][1][:n_g * R])
        out["seq2seq_cpu_sample"] = {
            "kind": "port", "threads": torch.get_num_threads(),
            "tower_queries_per_s": n_t / t_tower, "nci_generate_queries_per_s": n_g / t_gen,
            "sample": f"oracle.t5 (torch fp32): tower on {n_t} queries in {t_tower:.2f}s, nci_generate on {n_g} queries "
                      f"(beams {R}, per-query loop like the reference's infer) in {t_gen:.2f}s",
            "agreement_on_sample": {
                "tower_max_abs_diff": float((qemb[:n_t].cpu() - ref_e).abs().max()),
                "beams_identical": bool(torch.equal(dec, ref_dec)),
                "beam_score_max_abs_diff": float(np.abs(sc - ref_sc.numpy()).max())},
        }
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--docs", type=int, default=N_DOCS, help="corpus rows (default: MSMARCO 8,841,823)")
    ap.add_argument("--queries", type=int, default=N_QUERIES)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-seq2seq-legs", action="store_true", help="skip the untimed tower / beam-search extras")
    ap.add_argument("--exact-f32-path", action="store_true",
                    help="search with the f32-MFMA kernel only (no f16 pre-filter); same results")
    args = ap.parse_args()

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            raise SystemExit("launch multi-GPU runs with torch.distributed.run (one process per GPU)")
    hip.require_gpu()
    # MEVI_BENCH_BACKEND=gloo: rehearsal of the N > 1 path on a box with fewer GPUs than ranks (ranks share devices, the
    # collectives go through host memory) -- for checking the sharded search end to end, not for timing
    backend = os.environ.get("MEVI_BENCH_BACKEND", "nccl")
    if backend != "nccl":
        local_rank %= torch.cuda.device_count()
    torch.cuda.set_device(local_rank)
    device = torch.device("cuda", local_rank)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=device)
        else:
            dist.init_process_group(backend)

    n_docs, nq = args.docs, args.queries
    start, end = dense.shard_range(n_docs, rank, world)
    docs = gen_shard(start, end, device, n_docs)
    query = gen_queries(nq, device, n_docs)
    torch.cuda.synchronize()

    L = hip.lib()

    # index build (= faiss index.add): split image of the shard, untimed like the corpus upload
    index = docs if args.exact_f32_path else dense.DenseIndex(docs)
    torch.cuda.synchronize()

    def step():
        return dense.sharded_ip_topk(query, index, TOPK, id_offset=start)

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        s, i = step()
    barrier()
    L.mevi_ip_topk_set_profiling(1)
    filt_ms = filt_flops = comp_ms = 0.0
    launches = n_unproven = 0
    t0 = time.perf_counter()
    for _ in range(args.steps):
        s, i = step()
        st = hip.IpTopkStats()
        L.mevi_ip_topk_get_stats(st)
        filt_ms += st.filter_ms
        comp_ms += st.compact_ms
        filt_flops += st.filter_flops
        launches += st.n_chunks
        n_unproven += st.n_failed_queries
    barrier()
    elapsed = time.perf_counter() - t0
    L.mevi_ip_topk_set_profiling(0)
    if world > 1:
        tmax = torch.tensor([elapsed], dtype=torch.float64, device=device if backend == "nccl" else "cpu")
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        elapsed = float(tmax.item())

    # sanity (untimed): every query's planted neighbour must be its rank-1 hit
    top1 = i[:, 0].cpu().numpy()
    planted_ok = float((top1 == planted_ids(nq, n_docs)).mean())

    if rank == 0:
        ms_per_step = elapsed / args.steps * 1e3
        achieved = filt_flops / (filt_ms * 1e-3) / 1e12 if filt_ms > 0 else None
        if args.exact_f32_path:
            kernel, peak, peak_note = "ip_filter_kernel", PEAK_F32_MFMA_TFLOPS, "f32 MFMA dense peak"
        else:  # one f16 MFMA per product
            kernel, peak = "ip_filter_h1_kernel", PEAK_F16_MFMA_TFLOPS
            peak_note = "f16 MFMA dense peak"
        # HBM-side bytes per filter launch: PMC (FETCH_SIZE x 2 on gfx950) of THIS command, recorded by
        # tools/prof_traffic.sh + tools/traffic_summary.py; PMC passes cannot run inside the timed bench.
        traffic, traffic_note = None, "PMC not collected for this configuration"
        tfile = os.path.join(ROOT, "profiles", "r01_filter_h1_traffic.json")
        if (not args.exact_f32_path and world == 1 and n_docs == N_DOCS and nq == N_QUERIES and os.path.exists(tfile)):
            with open(tfile) as f:
                tj = json.load(f)
            traffic = tj["hbm_side_bytes_per_launch"]
            traffic_note = ("bytes per launch, recorded PMC pass (profiles/r01_filter_h1_traffic.json): L2 memory-side "
                            "requests incl. Infinity Cache hits; L2 hit rate %.2f; the corpus image itself is %.2f GB per "
                            "launch" % (tj["l2_hit_rate"], n_docs * DIM * 2 / 1e9 / (launches / args.steps)))
        out = {
            "metric": "queries/sec @ MRR@10-match, MSMARCO dev, 1/2/4/8 MI355X",
            "value": nq * args.steps / elapsed,
            "unit": "queries/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": ms_per_step,
            "higher_is_better": True,
            "scaling": "strong",
            "vs_baseline": None,
            "dtype": "f32" if args.exact_f32_path else "f16+f32",
            "dtype_note": "f32 MFMA chains only" if args.exact_f32_path else
            "f16 MFMA pre-filter (selection only) + exact f32 fmaf-chain re-score with a per-query completeness proof: "
            "results are the f32 results, bit for bit",
            "data": "synthetic",
            "config": {
                "workload": "C2 dense arm (faiss_search.py Flat): %d x %d f32 queries x %d x %d f32 docs, "
                            "exact inner-product top-%d" % (nq, DIM, n_docs, DIM, TOPK),
                "queries": nq, "docs": n_docs, "dim": DIM, "topk": TOPK,
                "parallelism": ("corpus row-sharded x%d, RCCL all-gather of per-shard top-k" % world
                                + ("" if backend == "nccl" else " [REHEARSAL over %s, ranks share devices: not a timing]" % backend))
                if world > 1 else "single GPU",
                "planted_top1_ok": planted_ok,
            },
            "roofline": {
                "kernel": kernel,
                "bound": "mfma",
                "achieved": achieved,
                "peak": peak,
                "peak_note": peak_note,
                "unit": "TFLOP/s",
                "frac": achieved / peak if achieved else None,
                "queries_sent_to_exact_fallback": n_unproven,
                "traffic": traffic,
                "traffic_note": traffic_note,
                "launches": launches,
                "avg_launch_ms": filt_ms / launches if launches else None,
                "algorithmic_flops_per_launch": filt_flops / launches if launches else None,
                "other_kernels_ms_per_step": {"compact_kernel": comp_ms / args.steps},
            },
        }
        if not args.no_cpu_baseline and world == 1:
            out["cpu_baseline"] = cpu_baseline(n_docs, nq)
        if not args.no_seq2seq_legs and world == 1:
            del index, docs
            torch.cuda.empty_cache()
            try:
                out.update(seq2seq_legs(device, nq, ms_per_step, with_cpu=not args.no_cpu_baseline))
            except Exception as e:      # the extras must never cost the headline line
                out["seq2seq_legs_error"] = f"{type(e).__name__}: {e}"
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
