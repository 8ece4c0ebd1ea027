#!/usr/bin/env python3
"""Randomised cross-check of the NCI model (mevi_amd.nci.NCIModel.generate: T5 encoder + KV-cached decoder + adaptor + PAWA
head + constrained beam search) and of the twin tower against the torch-fp32 oracle (oracle/t5.py, pinned to the reference by
goldens G1 / G2) on small random models: widths 64..384, 1-3 layers, 1-6 heads of 64, (M, K, R) incl. K < R, ragged query
lengths 1..32, batches 1..12, every prefix-table regime.  Tolerance-based (f32 summation orders differ): scores within 2e-4,
tower within 2e-4 relative, beams identical except swaps inside near-ties of the ORACLE's own scores (< 4e-4):
  python tools/stress_t5.py [seconds] [seed]"""
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import test_t5_gpu as T  # noqa: E402  (the seeded weight generator)
from mevi_amd import nci, t5  # noqa: E402
from oracle import t5 as ot5  # noqa: E402

budget = float(sys.argv[1]) if len(sys.argv) > 1 else 180.0
seed = int(sys.argv[2]) if len(sys.argv) > 2 else 0
rng = np.random.default_rng(seed)
dev = torch.device("cuda", 0)
t0, cases, swaps, beams_total, worst_sc, worst_tw, passages, trees = time.time(), 0, 0, 0, 0.0, 0.0, 0, 0
while time.time() - t0 < budget:
    heads = int(rng.integers(1, 7))
    d = int(rng.choice([64, 128, 256, 384]))
    d_ff = int(rng.choice([64, 256, 1024]))
    M, K = int(rng.integers(1, 5)), int(rng.choice([2, 4, 12, 32, 256]))
    if K ** M > 1 << 22:
        M = 2
    R = int(rng.choice([1, 2, 4, 10]))
    if R > K ** M:
        R = K ** M
    layers = dict(enc_layers=int(rng.integers(1, 4)), dec_layers=int(rng.integers(1, 4)), adaptor_layers=int(rng.integers(1, 3)))
    torch.manual_seed(int(rng.integers(1 << 30)))
    W, cfg = T._seeded_nci_weights(M, K, d, d_ff, heads, **layers)
    B, S = int(rng.integers(1, 13)), 32
    ids = np.zeros((B, S), np.int64)
    mask = np.zeros((B, S), np.int64)
    for i in range(B):
        L = int(rng.choice([1, 2, 32, int(np.clip(rng.poisson(9) + 2, 3, S))]))
        ids[i, :L - 1] = rng.integers(3, 1000, size=L - 1)
        ids[i, L - 1] = 1
        mask[i, :L] = 1
    ids, mask = torch.from_numpy(ids), torch.from_numpy(mask)
    table_bytes = int(rng.choice([0, 200 << 10, 64 << 20, 6 << 30]))
    tag = dict(heads=heads, d=d, d_ff=d_ff, M=M, K=K, R=R, B=B, table_bytes=table_bytes, **layers)
    with torch.no_grad():
        odec, osc, oenc = ot5.nci_generate(W, cfg, ids, mask, R)
        oreps = ot5.tower_encode(W, dict(cfg), ids, mask)
    model = nci.NCIModel(W, device=dev, prefix_table_bytes=table_bytes, **cfg)
    dec, sc, enc, _ = model.generate(ids, mask, num_beams=R)
    sc = np.asarray(sc)
    diff = float(np.abs(sc - osc.numpy()).max())
    worst_sc = max(worst_sc, diff)
    if diff > 2e-4:
        print("BAD scores", tag, diff)
        sys.exit(1)
    got, want, oscq = dec.cpu().numpy().reshape(B, R, -1), odec.numpy().reshape(B, R, -1), osc.numpy().reshape(B, R)
    for i in range(B):
        for j in range(R):
            if (got[i, j] == want[i, j]).all():
                continue
            twins = [jj for jj in range(R) if (got[i, j] == want[i, jj]).all()]
            if not twins or abs(oscq[i, twins[0]] - oscq[i, j]) >= 4e-4:
                print("BAD beams", tag, i, j, twins, oscq[i].tolist())
                sys.exit(1)
            swaps += 1
    beams_total += B * R
    tower = t5.TwinTower(W, device=dev, **{k_: v for k_, v in cfg.items() if k_ not in ("M", "K", "adaptor_layer_num")})
    reps = tower.encode_query({"input_ids": ids, "attention_mask": mask}).cpu()
    tw = float((reps - oreps).abs().max() / oreps.abs().max())
    worst_tw = max(worst_tw, tw)
    if tw > 2e-4:
        print("BAD tower", tag, tw)
        sys.exit(1)
    if K ** M <= 1024 and rng.random() < 0.5:     # the two ablation searches: ALL code paths (_generate_all) and a generic prefix tree
        with torch.no_grad():
            oall, _ = ot5.nci_generate_all(W, cfg, ids, mask)
        gall, _ = model.generate_all(ids, mask, max_rows=int(rng.choice([5, 64, 1 << 16])))
        da = float((gall.cpu() - oall).abs().max())
        worst_sc = max(worst_sc, da)
        if da > 2e-4:
            print("BAD generate_all", tag, da)
            sys.exit(1)
        allp = np.stack(np.meshgrid(*[np.arange(K)] * M, indexing="ij"), -1).reshape(-1, M)
        paths = allp[rng.random(len(allp)) < rng.choice([0.1, 0.5, 1.0])]
        if len(paths) == 0:
            paths = allp[:1]
        Rt = int(rng.choice([1, 2, 4]))
        with torch.no_grad():
            tdec, tsc, _ = ot5.nci_generate_tree(W, cfg, ids, mask, Rt, paths)
        gd, gs, _, _ = model.generate(ids, mask, num_beams=Rt, decode_tree=nci.PrefixTree(paths, M, K, dev))
        gs, tsn = np.asarray(gs), tsc.numpy()
        real = tsn > -1e8                                    # (-1e9-seeded placeholder beams when the trie has fewer paths than beams)
        if np.abs(gs[real] - tsn[real]).max(initial=0.0) > 2e-4 or not np.array_equal(gs > -1e8, real):
            print("BAD tree scores", tag, len(paths), Rt)
            sys.exit(1)
        gtok, ttok, tq = gd.cpu().numpy().reshape(B, Rt, -1), tdec.numpy().reshape(B, Rt, -1), tsn.reshape(B, Rt)
        for i in range(B):
            for j in range(Rt):
                if not real.reshape(B, Rt)[i, j] or (gtok[i, j] == ttok[i, j]).all():
                    continue
                twins = [jj for jj in range(Rt) if (gtok[i, j] == ttok[i, jj]).all()]
                if not twins or abs(tq[i, twins[0]] - tq[i, j]) >= 4e-4:
                    print("BAD tree beams", tag, i, j)
                    sys.exit(1)
                swaps += 1
        trees += 1
    if rng.random() < 0.4:      # the passage side of the tower: 128-token sequences (other attention kernels), ragged lengths 1..128
        Bp, Sp = int(rng.integers(1, 9)), 128
        pid = np.zeros((Bp, Sp), np.int64)
        pm = np.zeros((Bp, Sp), np.int64)
        for i in range(Bp):
            L = int(rng.choice([1, 33, 128, int(np.clip(rng.normal(70, 30), 2, 128))]))
            pid[i, :L - 1] = rng.integers(3, 1000, size=L - 1)
            pid[i, L - 1] = 1
            pm[i, :L] = 1
        pid, pm = torch.from_numpy(pid), torch.from_numpy(pm)
        with torch.no_grad():
            opr = ot5.tower_encode(W, dict(cfg), pid, pm)
        pr = tower.encode_passage({"input_ids": pid, "attention_mask": pm}).cpu()
        tp = float((pr - opr).abs().max() / opr.abs().max())
        worst_tw = max(worst_tw, tp)
        passages += Bp
        if tp > 2e-4:
            print("BAD passage tower", tag, tp)
            sys.exit(1)
    cases += 1
    if cases % 20 == 0:
        print(f"{cases} models ok ... last {tag}", flush=True)
    del model, tower
print(f"{cases} random models, {beams_total} beams: all within tolerance; beams in a near-tie swap {swaps}; largest beam-score diff {worst_sc:.2e}, "
      f"largest relative tower diff {worst_tw:.2e} (incl. {passages} 128-token passages); {trees} models also through generate_all and a random prefix tree")
