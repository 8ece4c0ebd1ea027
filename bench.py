#!/usr/bin/env python3
"""bench.py -- headline benchmark of the MEVI hot path on MI355X.

Workload (BASELINE.json configs[1], "C2"): the dense arm of MEVI, i.e. what
`faiss_search.py --param Flat` computes for MSMARCO dev: 6980 x 768 f32 queries
against the 8,841,823 x 768 f32 passage-embedding matrix (27.16 GB, resident in
HBM), exact inner-product top-1000.  One "step" = one full search of all queries.

  python bench.py [--gpus N] [--steps K] [--warmup W]                            (N > 1: starts its own N ranks)
  python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N ...     (the driver's launch line)

Plain `python bench.py --gpus N` with N > 1 and no WORLD_SIZE in the environment starts the N ranks itself -- a CHILD
`python -m torch.distributed.run` on a free 127.0.0.1 port, before this process has touched the GPU, as the reference's
entry points spawn theirs (MEVI/main.py:286-298, MEVI/generate.py:215-219) -- relays rank 0's line and exits with the
child's return code.

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

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

np = torch = dist = dense = hip = None      # bound by late_imports(): the self-launching parent never imports them


def late_imports():
    global np, torch, dist, dense, hip
    import numpy
    import torch as _torch
    import torch.distributed as _dist

    from mevi_amd import dense as _dense
    from mevi_amd import hip as _hip

    np, torch, dist, dense, hip = numpy, _torch, _dist, _dense, _hip


def self_launch(n, argv):
    """`python bench.py --gpus N` (N > 1) outside a launcher: start the N ranks as a CHILD process -- the same command
    line the driver uses -- on a free localhost port; stdout / stderr are inherited, so rank 0's one JSON line is this
    process's one JSON line.  Nothing here touches the GPU, and nothing is exec'ed.  Returns the child's return code."""
    import socket
    import subprocess

    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")     # dmabuf IPC only on this pool (RCCL needs it)
    env.setdefault("OMP_NUM_THREADS", str(max(1, (os.cpu_count() or 8) // n)))
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(n), "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__), *argv]
    return subprocess.run(cmd, env=env).returncode

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


def cpu_baseline(n_docs, nq_full, target_s=15.0):
    """The dense arm on the host cores (tools/bench_cpu.py): faiss itself when importable, else the faiss-Flat-style port
    (blocked sgemm + per-query heaps) with backend / threads / block swept and the best used for a bounded sample."""
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import bench_cpu

    return bench_cpu.dense_baseline_subprocess(n_docs, nq_full, DIM, TOPK, target_s=target_s)


# ---- per-query arithmetic of the seq2seq arm (KV-cached, last-position, valid-column formulation) -----
def seq2seq_flops(M, K, R, real_tokens_per_query, d=768, dff=3072, enc_layers=12, dec_layers=6, adaptor_layers=4, S=32):
    """(padded, executed) FLOP per query of NCI generate.  `padded` is SURVEY 8(d)'s budget (every query 32 tokens, the
    adaptor and the head per beam and step); `executed` counts what this build runs: real tokens only in the encoder
    and the cross-attention K|V, the adaptor / head matrices per PREFIX (amortised to ~0 over a pass, PrefixTables)
    except on the last position, where the (K+1)-column head GEMM runs per beam."""
    lin_enc = 2 * (4 * d * d + 2 * d * dff)                    # q,k,v,o + wi,wo per token and layer
    enc_pad = enc_layers * S * (lin_enc + 4 * S * d)
    t = real_tokens_per_query
    enc_real = enc_layers * t * (lin_enc + 4 * t * d)
    lin_dec = 2 * (4 * d * d + 2 * d * d + 2 * d * dff)        # self q,k,v,o + cross q,o + wi,wo per beam-step
    steps = 1 + R * M                                           # position 0 has one beam, positions 1..M have R
    dec = dec_layers * steps * (lin_dec + 4 * S * d)
    xkv_pad, xkv_real = dec_layers * S * 4 * d * d, dec_layers * t * 4 * d * d
    adaptor = adaptor_layers * steps * 2 * (4 * d * d + 2 * d * 2048)
    head = steps * 2 * (K + 1) * d * d
    head_exec = R * 2 * (K + 1) * d * d                         # last position only
    return enc_pad + dec + xkv_pad + adaptor + head, enc_real + dec + xkv_real + head_exec


def tower_flops(lens, d=768, dff=3072, layers=12):
    """Executed FLOP of one T5-ANCE tower pass over sequences of `lens` real tokens (padding-free encoder incl. attention,
    one-token decoder: self-attention = o(v(x)), cross q / o, FFN; cross K|V projections of the real tokens)."""
    lens = np.asarray(lens, np.float64)
    n, real = len(lens), float(lens.sum())
    lin = 2 * (4 * d * d + 2 * d * dff)
    return layers * (real * lin + 4 * d * float((lens ** 2).sum())) + layers * n * (lin + 2 * 2 * d * d) \
        + layers * real * 4 * d * d + layers * 4 * d * real


def gemm_roofline(device, rows):
    """HIP-event time of the linear layers of one encoder block at the arm's pass size: the split-precision GEMM
    (three f16 MFMAs per product).  achieved = algorithmic 2MNK / time; executed = 3x that on the f16 matrix cores."""
    from mevi_amd import ops

    g = torch.Generator(device=device).manual_seed(3)
    shapes = [(rows, 2304, 768, "q|k|v"), (rows, 768, 768, "o"), (rows, 3072, 768, "wi"), (rows, 768, 3072, "wo")]
    per, flops, ms_total = [], 0.0, 0.0
    for M_, N_, K_, name in shapes:
        x = ops.split_rows(torch.randn((M_, K_), device=device, generator=g))
        w = ops.weight_split(torch.randn((N_, K_), device=device, generator=g) * K_ ** -0.5)
        ops.linear(x, w)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(5):
            ops.linear(x, w)
        e1.record()
        torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / 5
        f = 2.0 * M_ * N_ * K_
        per.append({"layer": name, "m": M_, "n": N_, "k": K_, "ms": ms, "algorithmic_tflops": f / ms / 1e9})
        flops, ms_total = flops + f, ms_total + ms
        del x, w
    return {"kernel": "gemm_split_kernel", "bound": "mfma", "achieved": flops / ms_total / 1e9, "unit": "TFLOP/s",
            "executed_f16_tflops": 3 * flops / ms_total / 1e9, "peak": PEAK_F16_MFMA_TFLOPS,
            "peak_note": "f16 MFMA dense peak; the kernel issues three f16 MFMAs per f32 product, so frac = 3 x achieved / peak",
            "frac": 3 * flops / ms_total / 1e9 / PEAK_F16_MFMA_TFLOPS,
            "exact_f32_kernel_note": "MEVI_GEMM=exact runs gemm_nt_kernel (sequential f32 chains) at 110-126 TFLOP/s = 0.70-0.80 "
                                     "of the 157.3 TFLOP/s f32 matrix peak (profiles/r01_gemm_clock_util.txt)",
            "layers": per}


def cli_inclusive(device, docs, index_build_s, search_ms, nq, n_docs):
    """What one `faiss_search.py` process pays around the resident-corpus search: corpus upload (file in the page cache ->
    HBM: mevi_amd.io.upload_rows, a ring of pinned staging buffers filled by reader threads ahead of the copy engine; measured
    on a 2 GB sample file by method and scaled), index build, search."""
    import shutil
    import tempfile

    from mevi_amd import io as mio

    rows = min(docs.shape[0], (2 << 30) // (4 * DIM))
    need = rows * DIM * 4 + (64 << 20)
    tmpdir = tempfile.gettempdir()
    if os.path.isdir("/dev/shm") and os.access("/dev/shm", os.W_OK) and shutil.disk_usage("/dev/shm").free > 2 * need:
        tmpdir = "/dev/shm"
    elif shutil.disk_usage(tmpdir).free < 2 * need:       # a small box: sample what fits (at least 64 MB)
        rows = max((64 << 20) // (4 * DIM), min(rows, shutil.disk_usage(tmpdir).free // (8 * DIM * 4)))
    path = os.path.join(tmpdir, "mevi_bench_upload_%d.bin" % os.getpid())
    by_method = {}
    try:
        docs[:rows].cpu().numpy().tofile(path)
        for method in ("pread", "copy"):
            os.environ["MEVI_UPLOAD"] = method
            m = mio.map_rows(path, DIM)
            mio.upload_rows(m[: rows // 8], device)
            best = 0.0
            for _ in range(2):
                t = time.perf_counter()
                mio.upload_rows(m, device)
                torch.cuda.synchronize()
                best = max(best, rows * DIM * 4 / (time.perf_counter() - t) / 1e9)
            by_method[method] = round(best, 2)
            del m
    finally:
        os.environ.pop("MEVI_UPLOAD", None)
        mio.free_staging()
        if os.path.exists(path):
            os.remove(path)
    gbs = by_method.get("pread") or max(by_method.values())
    upload_s = n_docs * DIM * 4 / 1e9 / gbs
    total = upload_s + index_build_s + search_ms / 1e3
    return {"upload_gb_per_s": gbs, "upload_gb_per_s_by_method": by_method, "upload_frac_of_pcie_gen5_x16": gbs / 63.0, "upload_s": upload_s,
            "upload_sample": f"{rows} rows ({rows * DIM * 4 / 1e9:.2f} GB) of a mapped file in the page cache ({tmpdir}); pread = straight "
                             "into the pinned ring (the default), copy = memcpy out of the mapping; best of two",
            "host_cpus_granted": len(os.sched_getaffinity(0)),
            "index_build_s": index_build_s, "search_s": search_ms / 1e3, "queries_per_s": nq / total,
            "note": "PCIe-inclusive rate of ONE faiss_search.py invocation (corpus file in the page cache); every further "
                    "search on the resident corpus runs at `value`"}


SPLIT_DTYPE = "f16x3 split-precision MFMA GEMM, f32 accumulate (22-bit operand images, f32-equivalent); attention / norms f32"


def dense_small_batch(device, index, query, n_docs):
    """The reference's own timing hook for the dense arm is small batches (faiss_search.profile, MEVI/faiss_search.py:32-68:
    search at batch 1 / 2 / 4 / 8 x 10 batches).  With few queries the filter is bound by streaming the corpus' f16 image
    ("at query micro-batch <= 64 it flips to HBM"): ms per search on resident tensors, HBM rate of the image."""
    from mevi_amd import hip

    t = time.perf_counter()
    index.prepare_small()              # the 8-bit image of the <= 32-query searches (faiss_search.profile builds it inside `add`)
    torch.cuda.synchronize()
    index8_ms = (time.perf_counter() - t) * 1e3
    per = []
    for bs in (1, 2, 4, 8, 32, 64, 128, 255):
        row = {"batch": bs}
        for kk in (TOPK, 100):
            q = query[:bs].contiguous()
            index.search(q, kk)
            torch.cuda.synchronize()
            reps = 10
            t = time.perf_counter()
            unproven = 0
            for r in range(reps):
                index.search(query[r * bs:(r + 1) * bs].contiguous(), kk)
                st = hip.IpTopkStats()
                hip.lib().mevi_ip_topk_get_stats(st)
                unproven += int(st.n_i8_unproven > 0)
            torch.cuda.synchronize()
            ms = (time.perf_counter() - t) / reps * 1e3
            i8 = st.n_i8_queries > 0
            image_bytes = float(n_docs) * (DIM + 4) if i8 else float(n_docs) * DIM * 2   # int8 rows + their scales | f16 rows
            row["top%d" % kk] = {"ms_per_search": round(ms, 3), "queries_per_s": round(bs / ms * 1e3, 1), "image": "int8" if i8 else "f16",
                                 "image_gb_per_s": round(image_bytes / ms / 1e6, 1),
                                 "frac_of_hbm_peak": round(image_bytes / ms / 1e6 / 8000.0, 3)}
            if i8:
                row["top%d" % kk]["searches_repeated_through_f16"] = unproven
        per.append(row)
    best = per[0]["top%d" % TOPK]
    return {"what": "faiss_search.profile's regime: ONE index.search call of <batch> queries over the resident corpus (top-%d as the "
                    "scripts ask, and top-100), mean of 10 calls, inputs and outputs on the device" % TOPK,
            "dtype": "batch <= 32: int8 pre-filter (exact int32 sums, upper-bound keys); above: f16 pre-filter; results: exact f32 chains",
            "kernel": "ip_filter_i8_small_kernel for batch <= 32 (8-bit image, stationary query tile in two int8 digits, every wave streaming "
                      "its own corpus rows; MEVI_IP_I8=0: ip_filter_h1_small_kernel on the f16 image), "
                      "ip_filter_h16_kernel<2> / <4> (query tiles of 64 / 128) up to 128 queries, the 256-query tile above; the last launch of "
                      "a pass takes all remaining rows under a threshold estimated from the rows seen (rank_tau_kernel; exact by sample_check_kernel)",
            "roofline": {"bound": "hbm", "unit": "GB/s", "peak": 8000.0,
                         "algorithmic_bytes_per_search": float(n_docs) * (DIM + 4) if best["image"] == "int8" else float(n_docs) * DIM * 2,
                         "note": "bytes = the image the search reads once (int8: N x (768 + 4); f16: N x 768 x 2 -- per row in per_batch); the f32 "
                                 "rows of the re-scored survivors (K' x 3 KB per query) are not counted; the guide's achievable copy rate is 6.3 TB/s",
                         "achieved": best["image_gb_per_s"], "frac": best["frac_of_hbm_peak"]},
            "index8_build_ms": round(index8_ms, 1), "per_batch": per}


def index_build_leg(device, docs, rn, n_docs):
    """SURVEY 8(f).1 -- the offline index build (marco_generate_embedding_n_rq.sh): the passage tower and the RQ encode of the
    corpus, each with its roofline.  Passage tower: 2048 synthetic passages (lengths ~ N(70, 30) in [8, 128]) through
    TwinTower.encode_passage; RQ encode: the resident corpus at (4, 32) and (3, 256)."""
    from mevi_amd import rq, t5
    import synth

    out = {}
    rq.KEEP_ENCODE_WORKSPACE = True            # the record counters of `stats` below
    try:
        _rq_legs(out, device, docs, n_docs)
    finally:
        rq.KEEP_ENCODE_WORKSPACE = False
        rq._LAST_ENCODE.clear()
    _passage_leg(out, device, n_docs)
    return out


def _rq_legs(out, device, docs, n_docs):
    from mevi_amd import rq

    g = torch.Generator(device=device).manual_seed(5)
    for M_, K_ in ((4, 32), (3, 256)):
        cb = torch.stack([torch.randn((K_, DIM), device=device, generator=g) * (0.05 / (1 + j)) for j in range(M_)])
        rq.rq_encode(docs[:1 << 16], cb)
        ms_all = []
        for _ in range(3):     # the first whole-corpus call also sizes the workspace (a device allocation of ~0.4 GB: 28 vs 128 ms seen)
            torch.cuda.synchronize()
            t = time.perf_counter()
            codes = rq.rq_encode(docs, cb)
            torch.cuda.synchronize()
            ms_all.append((time.perf_counter() - t) * 1e3)
        ms = sorted(ms_all)[1]
        byts = 4.0 * n_docs * DIM + 4.0 * n_docs * M_
        out["rq_encode_%dx%d" % (M_, K_)] = {
            "ms": round(ms, 2), "ms_all": [round(x, 2) for x in ms_all], "rows_per_s": round(n_docs / ms * 1e3), "dtype": "f16 MFMA shortlist + exact f32 (r - c)^2 fmaf chains wherever the shortlist holds more than one centroid: codes are the f32 codes",
            "kernel": "rq_fast_kernel (f16 MFMA shortlist of every level from ONE product against all M*K centroids) + rf_fixup_kernel "
                      "(exact chains of the ambiguous row-levels' candidates) + rq_level_kernel on the rows the speculation got wrong",
            "stats": rq.last_encode_stats(),
            "roofline": {"bound": "hbm", "unit": "GB/s", "peak": 8000.0, "algorithmic_bytes": byts,
                         "bytes_note": "SURVEY 8(d): 4 N d (corpus once) + 4 N M (codes)",
                         "achieved": round(byts / ms / 1e6, 1), "frac": round(byts / ms / 1e6 / 8000.0, 4)}}
        # the matrix side of the same encode (2 N (M K) d products): (4, 32) takes every product in split precision
        # (three f16 MFMAs), (3, 256) as one -- neither is what binds: the x stream is (128-byte pieces of the f32 rows at
        # ~3.9 TB/s, once at (4, 32), once PER LEVEL at K = 256)
        mf = 2.0 * n_docs * (M_ * K_) * DIM * (3 if M_ * K_ <= 128 else 1)
        out["rq_encode_%dx%d" % (M_, K_)]["roofline_mfma"] = {
            "bound": "mfma", "unit": "TFLOP/s", "peak": PEAK_F16_MFMA_TFLOPS, "executed_f16_mfma_flop": mf,
            "achieved": round(mf / ms / 1e9, 1), "frac": round(mf / ms / 1e9 / PEAK_F16_MFMA_TFLOPS, 4),
            "x_stream_gb_per_s": round(4.0 * n_docs * DIM * (1 if K_ <= 32 else M_) / ms / 1e6, 1)}
        del codes, cb


def _passage_leg(out, device, n_docs):
    from mevi_amd import t5
    import synth

    TW = synth.tower_weights(device) if hasattr(synth, "tower_weights") else None
    if TW is not None:
        tower = t5.TwinTower(TW, device=device, num_layers=12, num_decoder_layers=12)     # device passes of t5.DEVICE_PASS_TOKENS
        rng = np.random.default_rng(0)
        n = 8192
        ids = np.zeros((n, 128), np.int64)
        mask = np.zeros((n, 128), np.int64)
        for i in range(n):
            L = int(np.clip(rng.normal(70, 30), 8, 128))
            ids[i, :L - 1] = rng.integers(3, 32100, size=L - 1)
            ids[i, L - 1] = 1
            mask[i, :L] = 1
        psg = {"input_ids": torch.from_numpy(ids).to(device), "attention_mask": torch.from_numpy(mask).to(device)}
        tower.encode_passage(psg)
        torch.cuda.synchronize()
        t = time.perf_counter()
        tower.encode_passage(psg)
        torch.cuda.synchronize()
        dt = time.perf_counter() - t
        real = int(mask.sum())
        d, ff = DIM, 3072
        lin = 2 * (4 * d * d + 2 * d * ff)
        lens = mask.sum(1)
        exe = 12 * (real * lin + 4 * d * float((lens.astype(np.float64) ** 2).sum())) + 12 * n * (lin + 2 * 2 * d * d) \
            + 12 * real * 4 * d * d + 12 * 4 * d * float(lens.sum())
        peak3 = PEAK_F16_MFMA_TFLOPS / 3
        out["passage_tower"] = {
            "passages": n, "tokens": 128, "real_tokens": real, "ms": round(dt * 1e3, 1), "passages_per_s": round(n / dt, 1),
            "corpus_hours_on_one_gpu": round(n_docs / (n / dt) / 3600, 2), "dtype": SPLIT_DTYPE,
            "roofline": {"bound": "mfma", "unit": "TFLOP/s", "flop_executed": exe, "achieved": round(exe / dt / 1e12, 1),
                         "peak": peak3, "frac": round(exe / dt / 1e12 / peak3, 4),
                         "peak_note": "f16 MFMA dense peak / 3 (three f16 MFMAs per f32 product); executed = real tokens only "
                                      "(12 encoder layers incl. attention, 12 one-token decoder layers, cross K|V of real tokens)"}}
        del tower, TW


def extras(device, docs, nq, n_docs, search_ms, index_build_s, with_cpu, query=None, dense_index=None):
    """Untimed extras (N = 1), measured after the timed region and not part of `value`: the path around the dense search at
    t5-base shapes (synthetic weights, MS MARCO-like query lengths, tools/synth.py) --
      * dense_small_batch: the reference's own profile() regime (batch 1..64), HBM roofline of the f16 image,
      * query tower and NCI generate alone, with the arm's roofline entries (GEMM kernel, per-query FLOP budget),
      * config C4 timed directly: tower -> dense search -> beam search -> tower again -> fine stage (+ the ensemble),
      * mrr10_match: CPU oracle vs HIP path on the same inputs (dense slice of the bench's own data; 8 queries of the chain),
      * NCI generate at BASELINE.json's configs[2] code shape (3 levels x 256 codes), with its own oracle agreement,
      * index_build: passage tower + RQ encode rooflines,
      * the CLI-inclusive rate of faiss_search.py (upload + index build + search)."""
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import bench_certify
    import chain_c4
    import synth
    from mevi_amd import nci, ops, t5

    M, K, R, gen_batch = 4, 32, 10, 8192
    out = {"gemm_mode": ops.GEMM_MODE}

    def guarded(name, fn):                      # one failing leg must not cost the others (nor the headline line)
        try:
            out[name] = fn()
        except Exception as e:
            import traceback

            out[name + "_error"] = f"{type(e).__name__}: {e} | {traceback.format_exc()[-500:]}"

    if dense_index is not None and query is not None:
        guarded("dense_small_batch", lambda: dense_small_batch(device, dense_index, query, n_docs))
        guarded("mrr10_match", lambda: {"dense": bench_certify.dense_certificate(query, docs, planted_ids(nq, n_docs), TOPK)})
    del dense_index
    torch.cuda.empty_cache()

    W, TW, _, rn = synth.weights(device, M, K)
    cpu_w = ({k: v.cpu() for k, v in W.items()}, {k: v.cpu() for k, v in TW.items()}) if with_cpu else None
    model = nci.NCIModel(W, device=device, M=M, K=K, adaptor_layer_num=4, num_layers=12, num_decoder_layers=6)
    tower = t5.TwinTower(TW, device=device, num_layers=12, num_decoder_layers=12)
    del W, TW
    rng = np.random.default_rng(0)
    ids, mask = synth.query_ids(nq, device, rng)
    real_tokens = int(mask.sum().item())

    def timed(fn, reps):
        fn()
        torch.cuda.synchronize()
        t = time.perf_counter()
        for _ in range(reps):
            o = fn()
        torch.cuda.synchronize()
        return (time.perf_counter() - t) / reps * 1e3, o

    def gen_all(mdl):
        return [mdl.generate(ids[a:a + gen_batch], mask[a:a + gen_batch], num_beams=R) for a in range(0, nq, gen_batch)]

    def nci_policies(mdl):
        """The beam search under the two prefix-table policies: `one_shot` = tables sized for THIS run's nq
        queries (what main.py --mode eval does: EvalRun.run -> NCIModel.expect_queries), `steady_state` = a long-lived model
        (every table that fits the byte budget).  Per policy: table build ms, the first pass (build + search) and the steady
        pass; break_even_queries = the workload above which the larger tables repay their build."""
        mdl.prefix_table_bytes, mdl._tables = 0, None
        gen_all(mdl)                                              # kernels warm and the allocator grown to a full pass, no table built yet
        mdl.prefix_table_bytes = None
        res, gen = {}, None
        for policy, q in (("one_shot", nq), ("steady_state", None)):
            mdl.prefix_table_queries, mdl._tables = q, None
            torch.cuda.synchronize()
            t = time.perf_counter()
            gen_all(mdl)
            torch.cuda.synchronize()
            first = (time.perf_counter() - t) * 1e3
            steady, gen = timed(lambda: gen_all(mdl), 2)
            res[policy] = dict(mdl.tables().describe(), first_pass_ms=round(first, 2), steady_ms=round(steady, 2))
        o, s_ = res["one_shot"], res["steady_state"]
        gain = (o["steady_ms"] - s_["steady_ms"]) / nq                       # ms per query the larger tables save
        extra = s_["build_ms"] - o["build_ms"]
        res["break_even_queries"] = int(extra / gain) if gain > 1e-9 and extra > 0 else (0 if extra <= 0 else None)
        res["note"] = ("one_shot is what a drop-in `main.py --mode eval` over these %d queries pays (first_pass_ms includes build_ms); "
                       "steady_state is a resident model; break_even_queries = extra build / per-query saving" % nq)
        return res, gen

    tower_ms, qemb = timed(lambda: tower.encode_query({"input_ids": ids, "attention_mask": mask}), 3)
    tabs, gen = nci_policies(model)
    nci_ms = tabs["steady_state"]["steady_ms"]
    pad, exe = seq2seq_flops(M, K, R, real_tokens / nq)
    peak3 = PEAK_F16_MFMA_TFLOPS / 3
    out["dense_arm_with_tower"] = {
        "tower_ms": tower_ms, "tower_queries_per_s": nq / tower_ms * 1e3, "search_ms": search_ms,
        "queries_per_s": nq / (tower_ms + search_ms) * 1e3, "pass_tokens": t5.DEVICE_PASS_TOKENS,
        "dtype": "tower: " + SPLIT_DTYPE + "; search: f16 pre-filter + exact f32 chains",
        "note": "generate.py --gen_query (T5-ANCE tower, t5-base shapes, synthetic weights) + faiss_search.py"}
    out["seq2seq_arm"] = {
        "nci_generate_ms": nci_ms, "nci_generate_queries_per_s": nq / nci_ms * 1e3, "beams": R, "rq": [M, K],
        "queries_per_pass": gen_batch, "dtype": SPLIT_DTYPE, "prefix_table_policies": tabs,
        "one_shot_queries_per_s": nq / tabs["one_shot"]["first_pass_ms"] * 1e3,
        "roofline": {
            "bound": "mfma", "unit": "TFLOP/s", "flop_per_query_survey_8d_padded": pad, "flop_per_query_executed": exe,
            "achieved": exe * nq / nci_ms / 1e9, "achieved_padded_equivalent": pad * nq / nci_ms / 1e9,
            "peak": peak3, "frac": exe * nq / nci_ms / 1e9 / peak3,
            "peak_note": "f16 MFMA dense peak / 3 (three f16 MFMAs per f32 product in the split-precision GEMM); `achieved` "
                         "counts the arithmetic executed (real tokens, per-prefix adaptor tables), not the padded budget"},
        "note": "main.py --mode eval beam search (t5-base NCI model, synthetic weights), all queries resident"}
    guarded("gemm_roofline", lambda: dict(gemm_roofline(device, real_tokens), dtype=SPLIT_DTYPE))

    def sweep_leg():
        """Tower and beam search by queries per call (tools/batch_sweep.py): 873 = what each rank of the 8-GPU ensemble runs
        (6980 / 8, MEVI/main.py:318-322); 128 / 8 = the reference's own batch regimes (generate.py:289, marco_eval_nci_rq.sh:12)."""
        import batch_sweep

        rows = batch_sweep.sweep(model, tower, ids, mask, M, K, R, seq2seq_flops=seq2seq_flops, tower_flops=tower_flops)
        return {"what": "one tower.encode_query / model.generate call of <queries> queries (median of 3-5 after a warm-up call; resident "
                        "model, steady-state prefix tables); frac = executed f32 products x 3 / time / f16 MFMA peak",
                "at_873": next((r for r in rows if r["queries"] == 873), None),
                "at_128": next((r for r in rows if r["queries"] == 128), None), "per_size": rows}

    guarded("seq2seq_batch_sweep", sweep_leg)

    # ---- C4, timed directly on the resident corpus ---------------------------------------------------------------------
    chain, dindex = chain_c4.run(model, tower, docs, ids, mask, planted_ids(nq, n_docs), rn, M, K, R, TOPK, gen_batch, rng,
                                 with_latency=True)
    out["latency_b1"] = chain.pop("latency_b1", None)         # VERDICT r5 #4a: ONE query through the chain, stage by stage
    chain["dtype"] = "tower / NCI: " + SPLIT_DTYPE + "; dense: f16 pre-filter + exact f32 chains; fine stage: f32 chains; ensemble: f64"
    # the chain's own roofline: f16 matrix-core FLOP executed by its stages (three per product of the split GEMMs of both tower
    # passes and the beam search, one per product of the dense pre-filter) over the chain's wall clock
    tw = tower_flops(mask.sum(1).cpu().numpy())
    mfma = 3.0 * (2 * tw + exe * nq) + 2.0 * nq * float(n_docs) * DIM
    chain["roofline"] = {"bound": "mfma", "unit": "TFLOP/s", "peak": PEAK_F16_MFMA_TFLOPS,
                         "executed_f16_mfma_flop": mfma, "achieved": mfma / chain["chain_ms"] / 1e9,
                         "frac": mfma / chain["chain_ms"] / 1e9 / PEAK_F16_MFMA_TFLOPS,
                         "flop_split": {"tower_x2_f32_products": 2 * tw, "nci_f32_products": exe * nq,
                                        "dense_filter_f16_products": 2.0 * nq * float(n_docs) * DIM},
                         "peak_note": "f16 MFMA dense peak; f32 products of the split GEMMs counted three times (three f16 MFMAs each)"}
    chain["prefix_tables"] = dict(model.tables().describe(), policy="steady_state (resident model; the 3 timed repeats never build)",
                                  one_shot_chain_ms=round(chain["chain_ms"] - nci_ms + tabs["one_shot"]["first_pass_ms"], 2),
                                  one_shot_queries_per_s=round(nq / (chain["chain_ms"] - nci_ms + tabs["one_shot"]["first_pass_ms"]) * 1e3, 1),
                                  one_shot_note="chain_ms with the beam-search stage replaced by the one_shot policy's first pass (table build "
                                                "included): what ONE MS MARCO dev run of main.py pays")
    out["chain_c4"] = chain
    guarded("faiss_search_cli_inclusive", lambda: cli_inclusive(device, docs, index_build_s, search_ms, nq, n_docs))
    del dindex

    def chain_trained():
        """The same chain ONCE more on a TRAINED codebook (the random centroids above leave skewed cells -- the
        fine stage's candidates per query move toward SURVEY 8 a18's ~84 with balanced ones).  One timed pass after a warm-up."""
        c, di2 = chain_c4.run(model, tower, docs, ids, mask, planted_ids(nq, n_docs), rn, M, K, R, TOPK, gen_batch, rng, repeats=1,
                              codebook="trained", plant=False)
        del di2
        return {k_: c[k_] for k_ in ("codebook", "chain_ms", "queries_per_s", "fine_candidates_per_query", "fine_candidates_max",
                                     "stage_ms_detail", "mrr10_detail", "setup_untimed_ms") if k_ in c} | {
            "stage_ms_detail": c["ms"], "mrr10_detail": c["mrr10"], "planted_top1_ok": c["planted_top1_ok"]}


    if with_cpu:
        from oracle import t5 as ot5

        ncfg, tcfg = synth.oracle_cfgs(M, K)
        n_t, n_g = min(nq, 64), min(nq, int(os.environ.get("MEVI_BENCH_CERT_QUERIES", "32")))     # 8 -> 32 chain queries
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
        # the chain on the same inputs, CPU vs GPU (codebook / codes: the chain's own, re-derived with its generator state)
        def chain_cert():
            cb = chain_c4.LAST.get("codebook")
            return bench_certify.chain_certificate(model, tower, cpu_w, (ncfg, tcfg), docs, chain_c4.LAST["codes_h"], cb, ids, mask,
                                                   planted_ids(nq, n_docs), M, K, R, TOPK, n_q=n_g, oracle_generate=(ref_dec, ref_sc))

        try:
            out.setdefault("mrr10_match", {})["chain"] = chain_cert()
        except Exception as e:
            import traceback

            out.setdefault("mrr10_match", {})["chain_error"] = f"{type(e).__name__}: {e} | {traceback.format_exc()[-500:]}"
        del cpu_w
    chain_c4.LAST.clear()
    guarded("chain_c4_trained_codebook", chain_trained)      # after the certificate: it re-writes chain_c4.LAST's index artefacts
    chain_c4.LAST.clear()

    # ---- BASELINE.json configs[2]: 3 levels x 256 codes ------------------------------------------------------------------
    del model, gen
    torch.cuda.empty_cache()
    M2, K2 = 3, 256
    W2, _, _, _ = synth.weights(device, M2, K2, tower=False)
    cpu_w2 = {k: v.cpu() for k, v in W2.items()} if with_cpu else None
    model2 = nci.NCIModel(W2, device=device, M=M2, K=K2, adaptor_layer_num=4, num_layers=12, num_decoder_layers=6)
    del W2
    tabs2, gen2 = nci_policies(model2)
    nci2_ms = tabs2["steady_state"]["steady_ms"]
    pad2, exe2 = seq2seq_flops(M2, K2, R, real_tokens / nq)
    out["seq2seq_arm_rq_3x256"] = {
        "nci_generate_ms": nci2_ms, "nci_generate_queries_per_s": nq / nci2_ms * 1e3, "beams": R, "rq": [M2, K2],
        "queries_per_pass": gen_batch, "dtype": SPLIT_DTYPE, "prefix_table_policies": tabs2,
        "one_shot_queries_per_s": nq / tabs2["one_shot"]["first_pass_ms"] * 1e3,
        "flop_per_query_survey_8d_padded": pad2,
        "note": "BASELINE.json configs[2] code shape (3-level RQ-256): prefix tables sized from the device "
                "(nci.default_table_bytes: half the free memory, <= 128 GiB) hold the head matrices of the 65 536 two-code "
                "prefixes and, memory permitting, the adaptor outputs of the 16.7 M prefixes of the final position; the "
                "257-column head GEMM runs per beam at the final position only"}
    if with_cpu:
        def agree2():
            from oracle import t5 as ot5

            ncfg2, _ = synth.oracle_cfgs(M2, K2)
            n2 = min(nq, 16)
            with torch.no_grad():
                t = time.time()
                rd, rs, _ = ot5.nci_generate(cpu_w2, ncfg2, ids[:n2].cpu(), mask[:n2].cpu(), R)
                dt = time.time() - t
            return {"queries": n2, "oracle_seconds": round(dt, 1),
                    "beams_identical": bool(torch.equal(gen2[0][0][:n2 * R].cpu(), rd)),
                    "beam_score_max_abs_diff": float(np.abs(np.asarray(gen2[0][1][:n2 * R]) - rs.numpy()).max())}

        try:
            out["seq2seq_arm_rq_3x256"]["agreement_on_sample"] = agree2()
        except Exception as e:
            out["seq2seq_arm_rq_3x256"]["agreement_error"] = f"{type(e).__name__}: {e}"
    del model2, gen2, cpu_w2, tower
    torch.cuda.empty_cache()
    guarded("index_build", lambda: index_build_leg(device, docs, rn, n_docs))

    def sensitivity():
        """the headline's two data-dependent fast paths on corpora shaped like a dense retriever's output, at
        full C2 size (tools/data_sensitivity.py; per-kind records in the detail file, min / max here)."""
        import data_sensitivity

        return data_sensitivity.sweep(device, n_docs, nq)

    if n_docs >= 1_000_000 or os.environ.get("MEVI_BENCH_SENSITIVITY") == "1":
        guarded("data_sensitivity", sensitivity)

    def cli_chain():
        """the four scripts as FRESH processes at MS MARCO size.  tools/e2e_cli.py writes a 27 GB corpus file and
        takes ~40 s, so -- like the PMC traffic figure -- the bench line carries the RECORDED run (profiles/r06_e2e_cli.json),
        not a live one; MEVI_BENCH_CLI_CHAIN=1 runs it live (scratch under $TMPDIR)."""
        rec_path = os.path.join(ROOT, "profiles", "r06_e2e_cli.json")
        live = os.environ.get("MEVI_BENCH_CLI_CHAIN") == "1"
        if live:
            import subprocess
            import tempfile

            scratch = tempfile.mkdtemp(prefix="mevi_e2e_cli_")
            rec_path = os.path.join(scratch, "e2e_cli.json")
            subprocess.run([sys.executable, os.path.join(ROOT, "tools", "e2e_cli.py"), os.path.join(scratch, "data"), str(n_docs), rec_path],
                           check=True, capture_output=True, timeout=1200)
        with open(rec_path) as f:
            rec = json.load(f)
        return {"cli_chain_wall_s": rec["cli_chain_wall_s"], "cli_chain_wall_s_first_use": rec.get("cli_chain_wall_s_first_use"),
                "queries_per_s_cli_chain": rec.get("queries_per_s_cli_chain"),
                "live": live, "process_wall_s": {p["name"]: p["wall_s"] for p in rec["processes"]},
                "measured": "live in this run" if live else "recorded run of `python tools/e2e_cli.py` (profiles/r06_e2e_cli.json; baseline of the "
                                                            "round: profiles/r06_e2e_cli_baseline.json, 16.82 s)",
                "note": "generate.py --gen_query + faiss_search.py --param Flat + main.py --mode eval (cluster pickles present) + "
                        "ensemble_marco.py as fresh processes on a 27 GB docemb.bin, t5-base-shaped checkpoints, a real SentencePiece "
                        "tokenizer, every flag of marco_eval_nci_rq.sh / marco_ensemble.sh; per-phase lines in profiles/r06_e2e_cli.txt"}

    guarded("cli_chain", cli_chain)
    return out


# ---- the ONE line ---------------------------------------------------------------------------------------------------------
LINE_BUDGET = 7600      # bytes: the driver keeps the last 8 KB of stdout beside `parsed`; the whole line must fit in it
_DROP = ("note", "measured", "process_wall_s", "sample_detail", "per_size", "per_kind", "by_kind", "kinds", "level_ms", "bytes", "expected_queries", "ms_all", "chain_ms_all",
         "chain_ms_min", "device_batch", "rows_per_s", "policy", "queries_per_pass", "real_tokens", "corpus_hours_on_one_gpu", "what", "slice", "per_rank", "per_batch", "layers", "checksums", "statistic", "stats",
         "flop_split", "stage_ms_detail", "mrr10_detail", "head_matrices_at", "adaptor_vectors_only_at", "one_shot_note", "setup_untimed_ms", "recall1000", "loadavg", "cgroup_cpu_max", "backend", "faiss",
         "upload_sample", "seconds", "cpu_seconds", "oracle_seconds", "algorithmic_bytes", "flop_executed",
         "flop_per_query_survey_8d_padded", "flop_per_query_executed", "executed_f16_mfma_flop", "algorithmic_bytes_per_search")
_DROP_ORDER = ("seq2seq_cpu_sample", "gemm_roofline", "dense_small_batch", "faiss_search_cli_inclusive", "dense_arm_with_tower", "chain_c4_trained_codebook", "seq2seq_batch_sweep",
               "seq2seq_arm_rq_3x256", "index_build", "multi_gpu")      # least important first, should the line still be long


_KEEP_PATHS = {("dtype",), ("config", "workload"), ("roofline", "kernel")}


def _compact(v, path=()):
    if isinstance(v, dict):
        return {k: _compact(x, path + (k,)) for k, x in v.items()
                if path + (k,) in _KEEP_PATHS
                or not (k in _DROP or k.endswith("_note") or k in ("dtype", "workload", "cpu", "kernel", "sweep"))}
    if isinstance(v, (list, tuple)):
        return [_compact(x, path) for x in v]
    if isinstance(v, float):
        return float("%.5g" % v)
    if isinstance(v, str) and len(v) > 120:
        return v[:117] + "..."
    return v


def summaries_into_config(out):
    """What the metric is ABOUT goes where the driver's record keeps it (`config`): the MRR@10-match certificate and the
    BASELINE.md section 3 number (the whole chain, timed over >= 3 repeats)."""
    cfg = out["config"]
    mm = out.get("mrr10_match") or {}
    cert = {}
    if "dense" in mm:
        cert["dense_lists_identical"] = mm["dense"].get("lists_identical")
        cert["dense_mrr10_abs_diff"] = mm["dense"].get("abs_diff")
    ch = mm.get("chain")
    if ch:
        diffs = [v["abs_diff"] for v in ch.values() if isinstance(v, dict) and "abs_diff" in v]
        same = [v["top10_identical"] for v in ch.values() if isinstance(v, dict) and "top10_identical" in v]
        cert.update(chain_queries=ch.get("queries"), beams_identical=ch.get("beams_identical"),
                    beams_differing=ch.get("beams_differing_from_oracle"), chain_mrr10_abs_diff=max(diffs) if diffs else None,
                    chain_top10_identical_frac=min(same) if same else None, tower_max_abs_diff=ch.get("tower_max_abs_diff"),
                    rq_codes_identical_on_sample=ch.get("rq_codes_identical_on_sample"),
                    within_1e_4=ch.get("mrr10_within_1e-4"))
    if "chain_error" in mm:
        cert["chain_error"] = mm["chain_error"][:120]
    if cert:
        cfg["mrr10_match"] = cert
    c4 = out.get("chain_c4")
    if c4:
        cfg["chain_c4"] = {"queries_per_s": c4["queries_per_s"], "queries_per_s_best": c4.get("queries_per_s_best"),
                           "repeats": c4.get("repeats"), "chain_ms": c4["chain_ms"], "stage_ms": c4["ms"],
                           "roofline_frac": round(c4["roofline"]["frac"], 4) if "roofline" in c4 else None,
                           "mrr10": c4.get("mrr10")}
    c5 = out.get("chain_c5")
    if c5:
        cfg["chain_c5"] = {"queries_per_s": c5["queries_per_s"], "chain_ms": c5["chain_ms"], "mrr10": c5.get("mrr10"),
                           "dense_lists_identical_on_all_ranks": c5.get("dense_lists_identical_on_all_ranks"),
                           "stage_ms_max_over_ranks": c5.get("stage_ms_max_over_ranks"),
                           "seq2seq_frac_per_rank": c5.get("seq2seq_frac_per_rank")}


def print_line(out):
    """Full record -> $MEVI_BENCH_DETAIL (default gpurun_out/bench_detail_n<N>.json, best effort) and stderr-free; the ONE
    stdout line is its compact form: every number, no prose, under LINE_BUDGET bytes."""
    summaries_into_config(out)
    path = os.environ.get("MEVI_BENCH_DETAIL", os.path.join(ROOT, "gpurun_out", "bench_detail_n%d.json" % out["n_gpus"]))
    try:
        os.makedirs(os.path.dirname(path), exist_ok=True)
        with open(path, "w") as f:
            json.dump(out, f, indent=1)
        out["detail_file"] = os.path.relpath(path, ROOT)
    except OSError:
        pass
    line = _compact(out)
    if "mrr10_match" in line and "mrr10_match" in line.get("config", {}):
        line["mrr10_match"] = "see config.mrr10_match"            # the certificate's summary travels in `config`, its record in the detail file
    for leg in _DROP_ORDER:
        if len(json.dumps(line)) <= LINE_BUDGET:
            break
        if leg in line:
            line[leg] = "see detail_file"
    print(json.dumps(line), flush=True)


def chain_c5(device, rank, world, backend, n_docs, nq, start, end, limit_s):
    """N > 1: BASELINE.json configs[4] is the FULL ensemble on the sharded corpus, so after the timed dense
    steps every rank runs the C5 chain (tools/chain_c4.run_sharded) -- full f32 corpus per rank, dense arm sharded, seq2seq
    arm as replicas over the rank's query slice -- and rank 0 reports `chain_c5`.  A watchdog bounds the leg: should a
    collective hang on the 8-GPU node, rank 0 still prints the headline line (the caller passes `emit`)."""
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import chain_c4
    import synth
    from mevi_amd import nci, t5

    M, K, R, gen_batch = 4, 32, 10, 8192
    docs = gen_shard(0, n_docs, device, n_docs)                  # the whole corpus on every rank (27 GB of 288 GB)
    W, TW, _, rn = synth.weights(device, M, K)
    model = nci.NCIModel(W, device=device, M=M, K=K, adaptor_layer_num=4, num_layers=12, num_decoder_layers=6)
    tower = t5.TwinTower(TW, device=device, num_layers=12, num_decoder_layers=12)
    del W, TW
    rng = np.random.default_rng(0)
    ids, mask = synth.query_ids(nq, device, rng)
    rec = chain_c4.run_sharded(model, tower, docs, start, end, ids, mask, planted_ids(nq, n_docs), rn, M, K, R, TOPK, gen_batch,
                               rng, rank, world, backend)
    allr = [None] * world
    dist.all_gather_object(allr, rec)
    if rank == 0:
        out = chain_c4.aggregate_sharded(allr, nq, world, TOPK, R, M, K, backend)
        try:        # each rank's share of the seq2seq arm against the f16 matrix peak / 3 (the 873-query regime)
            lens = mask.sum(1).cpu().numpy()
            for r in out["per_rank"]:
                mine = lens[r["rank"]::world]
                sm = r["stage_ms"]
                if sm.get("nci_beam_search"):
                    r["nci_frac"] = round(seq2seq_flops(M, K, R, float(mine.mean()))[1] * len(mine) / sm["nci_beam_search"] / 1e9 / (2500.0 / 3), 4)
                if sm.get("tower"):
                    r["tower_frac"] = round(tower_flops(mine) / sm["tower"] / 1e9 / (2500.0 / 3), 4)
            out["seq2seq_frac_per_rank"] = {"nci": [r.get("nci_frac") for r in out["per_rank"]],
                                            "tower": [r.get("tower_frac") for r in out["per_rank"]]}
        except Exception as e:
            out["seq2seq_frac_per_rank"] = f"{type(e).__name__}: {e}"
        return out
    return None


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--docs", type=int, default=N_DOCS, help="corpus rows (default: MSMARCO 8,841,823)")
    ap.add_argument("--queries", type=int, default=N_QUERIES)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-seq2seq-legs", action="store_true", help="skip the untimed tower / beam-search extras")
    ap.add_argument("--chain-limit-s", type=float, default=420.0, help="N > 1: watchdog of the untimed C5 chain leg")
    ap.add_argument("--exact-f32-path", action="store_true",
                    help="search with the f32-MFMA kernel only (no f16 pre-filter); same results")
    args = ap.parse_args()

    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        sys.exit(self_launch(args.gpus, sys.argv[1:]))          # before any GPU call; the child's rc is ours
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}: launch one process per GPU")
    late_imports()
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
    t_index = time.perf_counter()
    index = docs if args.exact_f32_path else dense.DenseIndex(docs)
    torch.cuda.synchronize()
    index_build_s = time.perf_counter() - t_index

    def step(trace=None):
        return dense.sharded_ip_topk(query, index, TOPK, id_offset=start, trace=trace)

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        s, i = step()
    barrier()
    L.mevi_ip_topk_set_profiling(1)
    trace = dense.SearchTrace()                 # kernel counters of every local search (first AND second round)
    t0 = time.perf_counter()
    for _ in range(args.steps):
        s, i = step(trace)
    barrier()
    elapsed = time.perf_counter() - t0
    L.mevi_ip_topk_set_profiling(0)
    filt_ms, comp_ms, filt_flops = trace.stats["filter_ms"], trace.stats["compact_ms"], trace.stats["filter_flops"]
    launches, n_unproven = int(trace.stats["n_chunks"]), int(trace.stats["n_failed_queries"])
    if world > 1:
        tmax = torch.tensor([elapsed], dtype=torch.float64, device=device if backend == "nccl" else "cpu")
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        elapsed = float(tmax.item())

    # N > 1, untimed: one more search with a device synchronisation after every phase, so that a sub-linear point of the
    # scaling curve can be attributed (local search vs all-gather vs merge vs second round) without another round
    multi_gpu = None
    if world > 1:
        barrier()
        tr = dense.SearchTrace(timed=True)
        step(tr)
        barrier()
        mine = {"rank": rank, "shard_rows": end - start, **{k + "_ms": round(v, 3) for k, v in tr.ms.items()},
                "rounds": tr.rounds, "second_round_queries": tr.second_round_queries,
                "filter_kernel_ms": round(tr.stats["filter_ms"], 3), "filter_launches": int(tr.stats["n_chunks"])}
        allr = [None] * world
        dist.all_gather_object(allr, mine)
        if rank == 0:
            ls = [r["local_search_ms"] for r in allr]
            kl = dense.truncated_list_len(TOPK, world)
            multi_gpu = {
                "local_search_ms": {"max": max(ls), "min": min(ls)},
                "all_gather_ms": max(r["all_gather_ms"] for r in allr), "merge_ms": max(r["merge_ms"] for r in allr),
                "rounds": allr[0]["rounds"], "second_round_queries": allr[0]["second_round_queries"],
                "first_round_list_len": kl, "all_gather_bytes_per_rank": nq * kl * 8,
                "note": "one extra, untimed search with a device synchronisation after every phase (phases therefore do "
                        "not overlap as they do in the timed steps); per_rank holds every rank's own numbers",
                "per_rank": allr}

    # sanity (untimed): every query's planted neighbour must be its rank-1 hit
    top1 = i[:, 0].cpu().numpy()
    planted_ok = float((top1 == planted_ids(nq, n_docs)).mean())

    if rank == 0:
        ms_per_step = elapsed / args.steps * 1e3
        achieved = filt_flops / (filt_ms * 1e-3) / 1e12 if filt_ms > 0 else None
        if args.exact_f32_path:
            kernel, peak, peak_note = "ip_filter_kernel", PEAK_F32_MFMA_TFLOPS, "f32 MFMA dense peak"
        else:  # one f16 MFMA per product: v_mfma_f32_16x16x32_f16 (ip_filter_h16_kernel) unless switched back or dim % 64
            h16 = os.environ.get("MEVI_IP_FILTER_MFMA", "") != "32" and DIM % 64 == 0
            kernel, peak = ("ip_filter_h16_kernel" if h16 else "ip_filter_h1_kernel"), PEAK_F16_MFMA_TFLOPS
            peak_note = "f16 MFMA dense peak"
        # HBM-side bytes per filter launch: PMC (FETCH_SIZE x 2 on gfx950) of THIS command, recorded by
        # tools/prof_traffic.sh + tools/traffic_summary.py; PMC passes cannot run inside the timed bench.
        traffic, traffic_note, traffic_current = None, "PMC not collected for this configuration", None
        import glob

        tfiles = sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_filter_h1_traffic.json")))
        tfile = tfiles[-1] if tfiles else ""
        if (not args.exact_f32_path and world == 1 and n_docs == N_DOCS and nq == N_QUERIES and os.path.exists(tfile)):
            with open(tfile) as f:
                tj = json.load(f)
            traffic = tj["hbm_side_bytes_per_launch"]
            traffic_note = ("bytes per launch, recorded PMC pass (profiles/%s): L2 memory-side " % os.path.basename(tfile) +
                            "requests incl. Infinity Cache hits; L2 hit rate %.2f; the corpus image itself is %.2f GB per "
                            "launch" % (tj["l2_hit_rate"], n_docs * DIM * 2 / 1e9 / (launches / args.steps)))
            sys.path.insert(0, os.path.join(ROOT, "tools"))
            import traffic_summary

            traffic_current = tj.get("filter_source_sha256") == traffic_summary.filter_source_sha()
            if not traffic_current:
                traffic_note += "; NOTE: the kernel sources changed after this PMC pass (re-run tools/prof_traffic.sh)"
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
                "traffic_pmc_of_these_sources": traffic_current,     # the recorded PMC pass hashed the filter sources it ran
                "traffic_note": traffic_note,
                "launches": launches,
                "avg_launch_ms": filt_ms / launches if launches else None,
                "algorithmic_flops_per_launch": filt_flops / launches if launches else None,
                "other_kernels_ms_per_step": {"compact_kernel": comp_ms / args.steps},
            },
        }
        if multi_gpu is not None:
            out["multi_gpu"] = multi_gpu
        if not args.no_cpu_baseline and world == 1:
            try:
                out["cpu_baseline"] = cpu_baseline(n_docs, nq)
            except Exception as e:
                out["cpu_baseline_error"] = f"{type(e).__name__}: {e}"
        if not args.no_seq2seq_legs and world == 1:
            try:
                out.update(extras(device, docs, nq, n_docs, ms_per_step, index_build_s, with_cpu=not args.no_cpu_baseline,
                                  query=query, dense_index=None if args.exact_f32_path else index))
            except Exception as e:      # the extras must never cost the headline line
                import traceback

                out["extras_error"] = f"{type(e).__name__}: {e} | {traceback.format_exc()[-600:]}"
    else:
        out = None

    # ---- N > 1: the C5 chain on every rank (untimed extra; `value` stays the sharded dense search) ---------------------------
    if world > 1 and not args.no_seq2seq_legs:
        import threading

        printed = threading.Event()

        def emit(extra):
            if printed.is_set():
                return
            printed.set()
            if rank == 0:
                out.update(extra)
                print_line(out)

        def give_up():      # a hung collective / a dead rank: rank 0 still prints the headline, then every rank leaves
            emit({"chain_c5_error": f"not finished within {args.chain_limit_s}s (watchdog)"})
            sys.stdout.flush()
            os._exit(3)     # non-zero: the launcher (and a self-launching parent) must see that the C5 leg did not finish

        dog = threading.Timer(args.chain_limit_s, give_up)
        dog.daemon = True
        dog.start()
        try:
            del index, docs
            torch.cuda.empty_cache()
            c5 = chain_c5(device, rank, world, backend, n_docs, nq, start, end, args.chain_limit_s)
            extra = {"chain_c5": c5} if rank == 0 else {}
        except Exception as e:
            import traceback

            extra = {"chain_c5_error": f"rank {rank}: {type(e).__name__}: {e} | {traceback.format_exc()[-600:]}"}
            if rank != 0:       # rank 0 may be waiting in a collective this rank will never join: let its watchdog end the wait
                print(json.dumps({"rank": rank, **extra}), file=sys.stderr, flush=True)
        dog.cancel()
        emit(extra)
    elif rank == 0:
        print_line(out)
    if world > 1:
        try:
            dist.barrier()
            dist.destroy_process_group()
        except Exception:
            pass


if __name__ == "__main__":
    main()
else:
    late_imports()      # imported (tests, tools): the helpers above need numpy / torch / mevi_amd bound
