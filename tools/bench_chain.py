"""Config C4 of SURVEY.md §8(d) on one MI355X: the whole inference chain at MS MARCO sizes, synthetic data.

    generate.py --gen_query   query tower, 6980 queries                    -> query embeddings
    faiss_search.py           exact top-1000 over 8,841,823 x 768          -> dense lists
    main.py --mode eval       NCI beam search (beams 10, RQ (4,32)) + query tower + fine stage on the beam clusters
    ensemble_marco.py         alpha .6 beta .03 gamma .02                   -> MRR@10 (host, untimed: it is file parsing)

queries/s = queries / (encode + search + beam + tower-again + fine), I/O excluded, inputs resident in HBM.
python tools/bench_chain.py [nq] [n_docs] [main.py --device_batch_size]"""
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import bench  # noqa: E402
import synth  # noqa: E402
from mevi_amd import dense, fine, metrics, nci, rq  # noqa: E402

nq = int(sys.argv[1]) if len(sys.argv) > 1 else bench.N_QUERIES
N = int(sys.argv[2]) if len(sys.argv) > 2 else bench.N_DOCS
batch = int(sys.argv[3]) if len(sys.argv) > 3 else 8192
M, K, R, k, d = 4, 32, 10, bench.TOPK, 768
dev = torch.device("cuda:0")
model, tower, g, rn = synth.build(dev, M, K, batch)
tower.batch_size = None          # the product's default pass size (mevi_amd.t5.DEVICE_PASS_TOKENS)
rng = np.random.default_rng(0)
ids, mask = synth.query_ids(nq, dev, rng)


def sync_time(fn):
    torch.cuda.synchronize()
    t = time.perf_counter()
    out = fn()
    torch.cuda.synchronize()
    return time.perf_counter() - t, out


def encode():
    return tower.encode_query({"input_ids": ids, "attention_mask": mask})


def main_py():
    """main.py --mode eval: per device batch, beam search -> query tower -> fine stage (EvalRun.infer)."""
    t = {"NCI beam search (main.py)": 0.0, "tower again (main.py fine stage)": 0.0, "fine stage: gather-dot + sort": 0.0}
    codes, ranked, ndoc = [], [], []
    for a in range(0, nq, batch):
        i, m = ids[a:a + batch], mask[a:a + batch]
        dt, o = sync_time(lambda: model.generate(i, m, num_beams=R))
        t["NCI beam search (main.py)"] += dt
        bc = nci.decode_token(o[0], K).view(-1, R, M).cpu().numpy()
        dt, q2 = sync_time(lambda: tower.encode_query({"input_ids": i, "attention_mask": m}))
        t["tower again (main.py fine stage)"] += dt
        dt, (rk, nd) = sync_time(lambda: fs.rerank(q2, bc))
        t["fine stage: gather-dot + sort"] += dt
        codes.append(bc)
        ranked += rk
        ndoc.append(nd)
    return t, np.concatenate(codes), ranked, np.concatenate(ndoc)


# ---- untimed set-up: corpus of C2, one planted neighbour per query at a controlled margin, RQ clusters --------------
encode()
_, qemb = sync_time(encode)
docs = bench.gen_shard(0, N, dev, N)
gt0 = bench.planted_ids(nq, N)
c = qemb - qemb.mean(0, keepdim=True)
c = c / c.norm(dim=1, keepdim=True)
z = torch.from_numpy(rng.uniform(3.8, 6.5, nq).astype(np.float32)).to(dev)     # margin in sigmas of the corpus noise
strength = z * 0.05 * qemb.norm(dim=1) / (qemb * c).sum(1).clamp_min(1e-6)
docs[torch.from_numpy(gt0).to(dev)] += strength[:, None] * c
codebook = torch.stack([rn(K, d, s=0.05 / (1 + j)) for j in range(M)])
codes = rq.rq_encode(docs, codebook)
codes_h = codes.cpu().numpy()
index = rq.ClusterIndex.from_codes(codes_h, K)
fs = fine.FineStage(docs, index)
dindex = dense.DenseIndex(docs)
torch.cuda.synchronize()

# ---- the timed chain (second pass of each stage; the first is the warm-up) ------------------------------------------
stages = {}
for rep in range(2):
    stages["tower (generate.py)"], qemb = sync_time(encode)
    stages["dense top-1000 (faiss_search.py)"], (ds, di) = sync_time(lambda: dindex.search(qemb, k))
    t, bcodes, ranked, ndoc = main_py()
    stages.update(t)
total = sum(stages.values())
for n_, t in stages.items():
    print(f"{n_:36s} {t*1e3:8.1f} ms   {nq/t:9.0f} q/s", flush=True)
print(f"{'chain':36s} {total*1e3:8.1f} ms   {nq/total:9.0f} q/s   ({ndoc.mean():.1f} fine candidates/query, max {ndoc.max()})", flush=True)
reuse = total - stages["tower again (main.py fine stage)"]
print(f"{'chain, main.py --query_embedding_path':36s} {reuse*1e3:7.1f} ms   {nq/reuse:9.0f} q/s   (fine stage reads generate.py's query_emb.bin)", flush=True)

# ---- metrics as marco_ensemble.sh computes them (host side, untimed) -------------------------------------------------
di_h, ds_h = di.cpu().numpy(), ds.cpu().numpy()
gts, dense_p, dense_s, fine_p, fine_s, clusters = {}, {}, {}, {}, {}, {}
for i in range(nq):
    q = f"q{i}"
    gt = [int(gt0[i])]
    fd = ranked[i][0]
    if i % 2 == 0 and len(fd):                  # a second relevant document that only the seq2seq arm can reach
        gt.append(int(fd[min(len(fd) - 1, int(rng.geometric(0.3)) - 1)]))
    gts[q] = gt
    dense_p[q], dense_s[q] = di_h[i].tolist(), ds_h[i].astype(np.float64).tolist()
    fine_p[q], fine_s[q] = fd.tolist(), ranked[i][1].astype(np.float64).tolist()
    clusters[q] = bcodes[i].tolist()


class CodeMap:                                   # rqmapping: doc id -> code tuple, without an 8.8 M-entry dict
    def __getitem__(self, p):
        return tuple(codes_h[p].tolist())


cranks, n_clusters = metrics.cluster_ranks(dense_p, clusters, CodeMap())
res = {"ANCE Pred": metrics.evaluate_ranked("ANCE Pred", [10, 50, 1000], gts, dense_p),
       "Fine Pred": metrics.evaluate_ranked("Fine Pred", [10, 50, 1000], gts, fine_p)}
ens = {q: metrics.ensemble_scores(dense_p[q], dense_s[q], cranks[q], fine_p[q], fine_s[q], n_clusters, 0.6, 0.03, 0.02)
       for q in gts}
res["ensemble"] = metrics.evaluate_ranked("score + 0.6 / (0.03 * crank + 1); punishment (1 - 0.02 * 0.6)", [10, 50, 1000], gts, ens)
import zlib  # noqa: E402
print("checksums:", {n_: zlib.crc32(np.ascontiguousarray(a_).tobytes()) for n_, a_ in
                     (("qemb", qemb.cpu().numpy()), ("codebook", codebook.cpu().numpy()), ("doc_codes", codes_h),
                      ("beam_codes", bcodes), ("dense_ids", di_h), ("ndoc", ndoc))})
line = {"workload": f"C4: {nq} queries, corpus {N} x {d}, beams {R}, RQ ({M},{K}), top-{k}",
        "device_batch": batch, "queries_per_s": round(nq / total, 1),
        "queries_per_s_reusing_query_embeddings": round(nq / reuse, 1), "ms": {n_: round(t * 1e3, 2) for n_, t in stages.items()},
        "fine_candidates_per_query": float(ndoc.mean())}
print(json.dumps(line))
