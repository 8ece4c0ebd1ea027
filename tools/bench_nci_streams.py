"""Experiment: NCI beam search of two half batches on two HIP streams (two host threads) against one batch on one stream.
The persistent GEMM leaves CUs idle in its last round of tiles (819 tiles on 256 CUs = 3.2 rounds) and the glue kernels
between GEMMs are bandwidth-bound: a second, independent chain could fill both.  python tools/bench_nci_streams.py [nq]"""
import os
import sys
import threading
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import synth  # noqa: E402

nq = int(sys.argv[1]) if len(sys.argv) > 1 else 6980
dev = torch.device("cuda:0")
model, tower, g, rn = synth.build(dev, 4, 32, 8192)
ids, mask = synth.query_ids(nq, dev, np.random.default_rng(0))


def one():
    return model.generate(ids, mask, num_beams=10)[0]


def two(parts=2):
    outs = [None] * parts
    streams = [torch.cuda.Stream(device=dev) for _ in range(parts)]
    cut = [nq * i // parts for i in range(parts + 1)]
    cur = torch.cuda.current_stream(dev)

    def work(i):
        with torch.cuda.device(dev), torch.cuda.stream(streams[i]):
            streams[i].wait_stream(cur)
            outs[i] = model.generate(ids[cut[i]:cut[i + 1]], mask[cut[i]:cut[i + 1]], num_beams=10)[0]
    th = [threading.Thread(target=work, args=(i,)) for i in range(parts)]
    for t in th:
        t.start()
    for t in th:
        t.join()
    for s in streams:
        cur.wait_stream(s)
    return torch.cat(outs)


def timed(fn, reps=3):
    fn()
    torch.cuda.synchronize()
    t = time.perf_counter()
    for _ in range(reps):
        out = fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t) / reps * 1e3, out


ms1, ref = timed(one)
print(f"one stream : {ms1:8.1f} ms  {nq / ms1 * 1e3:8.0f} q/s", flush=True)
for parts in (2, 3):
    ms2, got = timed(lambda: two(parts))
    print(f"{parts} streams  : {ms2:8.1f} ms  {nq / ms2 * 1e3:8.0f} q/s   same beams: {bool(torch.equal(ref, got))}", flush=True)
