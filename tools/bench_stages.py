"""Stage timings of the seq2seq arm at t5-base shapes (config C3/C4 of BASELINE.md), synthetic weights/ids.
python tools/bench_stages.py [nq] [batch] [M] [bits]"""
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import os as _os
from mevi_amd import hip as _hip
if _os.environ.get("MEVI_PROBE_LIB"):  # A/B timing of another build of the library on the same device
    _hip.LIB = _os.path.abspath(_os.environ["MEVI_PROBE_LIB"])
from mevi_amd import fine, rq  # noqa: E402

nq = int(sys.argv[1]) if len(sys.argv) > 1 else 2048
batch = int(sys.argv[2]) if len(sys.argv) > 2 else 512
M = int(sys.argv[3]) if len(sys.argv) > 3 else 4
bits = int(sys.argv[4]) if len(sys.argv) > 4 else 5
K, R, d = 2 ** bits, 10, 768
dev = torch.device("cuda:0")
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import synth  # noqa: E402

model, tower, g, rn = synth.build(dev, M, K, batch)
rng = np.random.default_rng(0)
ids, mask = synth.query_ids(nq, dev, rng)


def timed(fn, reps=2):
    fn()
    torch.cuda.synchronize()
    t = time.perf_counter()
    for _ in range(reps):
        out = fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t) / reps, out


dt, reps_ = timed(lambda: tower.encode_query({"input_ids": ids, "attention_mask": mask}))
print(f"tower  : {nq} queries in {dt*1e3:.1f} ms  -> {nq/dt:.0f} q/s   ({6.51e9*nq/dt/1e12:.1f} TFLOP/s padded-equivalent)", flush=True)

# passage side of the same tower (gen_doc_embedding): 128-token windows, lengths ~ clip(N(70, 30), 8, 128)
npsg = 4096
pids = np.zeros((npsg, 128), np.int64)
pmask = np.zeros((npsg, 128), np.int64)
for i in range(npsg):
    L = int(np.clip(rng.normal(70, 30), 8, 128))
    pids[i, :L - 1] = rng.integers(3, 32100, size=L - 1)
    pids[i, L - 1] = 1
    pmask[i, :L] = 1
pids, pmask = torch.from_numpy(pids).to(dev), torch.from_numpy(pmask).to(dev)
dt, _ = timed(lambda: tower.encode_passage({"input_ids": pids, "attention_mask": pmask}))
# encoder 12 layers x 128 tokens: 2*(4*768*768 + 2*768*3072) MACs per token and layer + attention 2*2*128*768
enc_flop = 12 * 128 * (2 * (4 * 768 * 768 + 2 * 768 * 3072) + 4 * 128 * 768)
print(f"passage: {npsg} passages x 128 tokens in {dt*1e3:.1f} ms -> {npsg/dt:.0f} passages/s   "
      f"({enc_flop*npsg/dt/1e12:.1f} TFLOP/s padded-equivalent, encoder only)", flush=True)


def gen_all():
    out = []
    for a in range(0, nq, batch):
        out.append(model.generate(ids[a:a + batch], mask[a:a + batch], num_beams=R)[0])
    return torch.cat(out)


dt, dec = timed(gen_all, reps=1)
flop = (5.44e9 + 50 * 9.91e7 + 4.53e8 + 50 * 4.40e7 + 50 * 3.89e7) * nq
print(f"nci gen: {nq} queries (batch {batch}, M={M}, K={K}) in {dt*1e3:.1f} ms -> {nq/dt:.0f} q/s   ({flop/dt/1e12:.1f} TFLOP/s by the per-beam flop count of SURVEY 8(d); the prefix tables execute less)", flush=True)

# RQ encode + fine stage on a synthetic corpus
N = 1_000_000
emb = torch.empty((N, d), device=dev)
for a in range(0, N, 1 << 18):
    emb[a:a + (1 << 18)] = 0.05 * torch.randn((min(1 << 18, N - a), d), device=dev, generator=g) + 0.02
cb = rn(M, K, d, s=0.05)
dt, codes = timed(lambda: rq.rq_encode(emb, cb))
print(f"rq enc : {N} rows (M={M}, K={K}) in {dt*1e3:.1f} ms -> {N/dt/1e6:.1f} M rows/s; {N*d*4*M/dt/1e9:.0f} GB/s of X, "
      f"{2.0*N*M*K*d/dt/1e12:.1f} T lane-op/s", flush=True)
index = rq.ClusterIndex.from_codes(codes.cpu().numpy(), K)
fs = fine.FineStage(emb, index)
beam = codes[torch.randint(0, N, (nq * R,), device=dev, generator=g)].view(nq, R, M).cpu().numpy()
dt, (out, ndoc) = timed(lambda: fs.rerank(reps_, beam))
print(f"fine   : {nq} queries, {ndoc.mean():.1f} docs/query avg (max {ndoc.max()}) in {dt*1e3:.1f} ms -> {nq/dt:.0f} q/s", flush=True)
