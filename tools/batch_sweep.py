"""The seq2seq arm by batch size: query tower and NCI generate at 8 .. 6980 queries per call -- the batch
regimes the reference itself runs (`--eval_batch_size 2` MEVI/marco_eval_nci_rq.sh:12, `--batch_size 128` MEVI/generate.py:289)
and the 873 queries each of the 8 replicas of BASELINE.json configs[4] gets (DistributedSampler, MEVI/main.py:318-322).

    python tools/batch_sweep.py [sizes, comma separated] [M] [K]

`sweep()` is what bench.py's `seq2seq_batch_sweep` leg calls; ms = median of `reps` calls after one warm-up call of the same
size (prefix tables exist before the first size is timed); frac = executed f32 products x 3 / time / the f16 matrix peak."""
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))

SIZES = (8, 64, 128, 512, 873, 1745, 3490, 6980)
PEAK3 = 2500.0 / 3


def _median_ms(fn, reps):
    fn()
    torch.cuda.synchronize()
    ts = []
    for _ in range(reps):
        t = time.perf_counter()
        fn()
        torch.cuda.synchronize()
        ts.append((time.perf_counter() - t) * 1e3)
    return float(np.median(ts))


def sweep(model, tower, ids, mask, M, K, R, sizes=SIZES, seq2seq_flops=None, tower_flops=None):
    """Per size n: the first n queries of (ids, mask) through tower.encode_query and model.generate (one call each)."""
    out = []
    for n in sizes:
        if n > ids.shape[0]:
            continue
        i, m = ids[:n].contiguous(), mask[:n].contiguous()
        reps = 5 if n <= 1024 else 3
        t_ms = _median_ms(lambda: tower.encode_query({"input_ids": i, "attention_mask": m}), reps) if tower is not None else None
        g_ms = _median_ms(lambda: model.generate(i, m, num_beams=R), reps) if model is not None else None
        row = {"queries": n}
        lens = m.sum(1).cpu().numpy()
        if t_ms is not None:
            row["tower_ms"] = round(t_ms, 3)
            row["tower_queries_per_s"] = round(n / t_ms * 1e3, 1)
            if tower_flops is not None:
                row["tower_frac"] = round(tower_flops(lens) / t_ms / 1e9 / PEAK3, 4)
        if g_ms is not None:
            row["nci_ms"] = round(g_ms, 3)
            row["nci_queries_per_s"] = round(n / g_ms * 1e3, 1)
            if seq2seq_flops is not None:
                _, exe = seq2seq_flops(M, K, R, float(lens.mean()))
                row["nci_frac"] = round(exe * n / g_ms / 1e9 / PEAK3, 4)
        out.append(row)
    return out


if __name__ == "__main__":
    from mevi_amd import hip as _hip
    if os.environ.get("MEVI_PROBE_LIB"):
        _hip.LIB = os.path.abspath(os.environ["MEVI_PROBE_LIB"])
    import synth
    import bench

    sizes = tuple(int(x) for x in sys.argv[1].split(",")) if len(sys.argv) > 1 else SIZES
    M = int(sys.argv[2]) if len(sys.argv) > 2 else 4
    K = int(sys.argv[3]) if len(sys.argv) > 3 else 32
    dev = torch.device("cuda:0")
    model, tower, _, _ = synth.build(dev, M, K, None)
    ids, mask = synth.query_ids(max(sizes), dev, np.random.default_rng(0))
    model.generate(ids[:max(sizes)], mask[:max(sizes)], num_beams=10)        # prefix tables built before anything is timed
    torch.cuda.synchronize()
    for row in sweep(model, tower, ids, mask, M, K, 10, sizes, bench.seq2seq_flops, bench.tower_flops):
        print(row, flush=True)
