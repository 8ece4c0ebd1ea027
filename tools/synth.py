"""Synthetic t5-base-shaped NCI model + T5-ANCE-shaped tower and MS MARCO-like query ids (config C3 of BASELINE.md),
shared by tools/bench_stages.py and tools/bench_chain.py."""
import os

import numpy as np
import torch

from mevi_amd import nci, t5

d = 768


def _t5_weights(W, nl, ndl, rn, dev, prefix_dec="decoder"):
    for st, n, dec in (("encoder", nl, False), (prefix_dec, ndl, True)):
        for l in range(n):
            p = f"{st}.block.{l}.layer"
            W[f"{p}.0.SelfAttention.q.weight"] = rn(d, d, s=(d * 64) ** -0.5)
            for nme in "kvo":
                W[f"{p}.0.SelfAttention.{nme}.weight"] = rn(d, d, s=d ** -0.5)
            W[f"{p}.0.layer_norm.weight"] = torch.ones(d, device=dev)
            ff = 1
            if dec:
                W[f"{p}.1.EncDecAttention.q.weight"] = rn(d, d, s=(d * 64) ** -0.5)
                for nme in "kvo":
                    W[f"{p}.1.EncDecAttention.{nme}.weight"] = rn(d, d, s=d ** -0.5)
                W[f"{p}.1.layer_norm.weight"] = torch.ones(d, device=dev)
                ff = 2
            W[f"{p}.{ff}.DenseReluDense.wi.weight"] = rn(3072, d, s=d ** -0.5)
            W[f"{p}.{ff}.DenseReluDense.wo.weight"] = rn(d, 3072, s=3072 ** -0.5)
            W[f"{p}.{ff}.layer_norm.weight"] = torch.ones(d, device=dev)
        W[f"{st}.block.0.layer.0.SelfAttention.relative_attention_bias.weight"] = rn(32, 12, s=0.5)
        W[f"{st}.final_layer_norm.weight"] = torch.ones(d, device=dev)


def weights(dev, M, K, seed=0, tower=True):
    """(NCI state dict, tower state dict, generator, rn) on `dev`: t5-base widths, 12/6 layers + 4 adaptor layers for the
    NCI model, 12/12 for the tower (which shares the NCI model's token embedding), the reference's initialiser scales."""
    g = torch.Generator(device=dev).manual_seed(seed)

    def rn(*shape, s=1.0):
        return torch.randn(shape, device=dev, generator=g) * s

    V = K * (M + 2) + 2
    W = {"shared.weight": rn(32128, d), "decode_embeddings.weight": rn(V, d),
         "adaptor_embeddings": torch.rand((1, 1, d), device=dev, generator=g),
         "adaptor_linear.weight": rn(d * V, d, s=d ** -0.5 * 0.3)}
    W["lm_head.weight"] = W["decode_embeddings.weight"]
    _t5_weights(W, 12, 6, rn, dev)
    for l in range(4):
        p = f"adaptor.layers.{l}"
        for a in ("self_attn", "multihead_attn"):
            W[f"{p}.{a}.in_proj_weight"], W[f"{p}.{a}.in_proj_bias"] = rn(3 * d, d, s=d ** -0.5), rn(3 * d, s=0.02)
            W[f"{p}.{a}.out_proj.weight"], W[f"{p}.{a}.out_proj.bias"] = rn(d, d, s=d ** -0.5), rn(d, s=0.02)
        W[f"{p}.linear1.weight"], W[f"{p}.linear1.bias"] = rn(2048, d, s=d ** -0.5), rn(2048, s=0.02)
        W[f"{p}.linear2.weight"], W[f"{p}.linear2.bias"] = rn(d, 2048, s=2048 ** -0.5), rn(d, s=0.02)
        for n_ in (1, 2, 3):
            W[f"{p}.norm{n_}.weight"], W[f"{p}.norm{n_}.bias"] = torch.ones(d, device=dev), torch.zeros(d, device=dev)
    TW = None
    if tower:
        TW = {"shared.weight": W["shared.weight"]}
        _t5_weights(TW, 12, 12, rn, dev)
    return W, TW, g, rn


def oracle_cfgs(M, K):
    """The config dicts oracle/t5.py reads for these shapes (NCI model, tower)."""
    base = dict(d_model=d, d_ff=3072, num_heads=12, d_kv=64, layer_norm_epsilon=1e-6, relative_attention_num_buckets=32)
    return (dict(base, M=M, K=K, num_layers=12, num_decoder_layers=6, adaptor_layer_num=4),
            dict(base, num_layers=12, num_decoder_layers=12))


def build(dev, M, K, batch, seed=0):
    W, TW, g, rn = weights(dev, M, K, seed)
    model = nci.NCIModel(W, device=dev, M=M, K=K, adaptor_layer_num=4, num_layers=12, num_decoder_layers=6)
    tower = t5.TwinTower(TW, device=dev, num_layers=12, num_decoder_layers=12, batch_size=batch)
    return model, tower, g, rn


def tower_weights(dev, seed=0):
    """State dict of the T5-ANCE-shaped tower alone (12 + 12 layers), same initialiser scales as weights()."""
    g = torch.Generator(device=dev).manual_seed(seed)

    def rn(*shape, s=1.0):
        return torch.randn(shape, device=dev, generator=g) * s

    TW = {"shared.weight": rn(32128, d)}
    _t5_weights(TW, 12, 12, rn, dev)
    return TW


def build_tower(dev, seed=0):
    return t5.TwinTower(tower_weights(dev, seed), device=dev, num_layers=12, num_decoder_layers=12)


def query_ids(nq, dev, rng):
    """lengths ~ clip(Poisson(9)+2, 3, 32), tokens uniform in [3, 32100), eos = 1 then pad 0."""
    ids = np.zeros((nq, 32), np.int64)
    mask = np.zeros((nq, 32), np.int64)
    for i in range(nq):
        L = int(np.clip(rng.poisson(9) + 2, 3, int(os.environ.get("SYNTH_MAX_QUERY_LEN", "32"))))     # env: experiments on the longest query
        ids[i, :L - 1] = rng.integers(3, 32100, size=L - 1)
        ids[i, L - 1] = 1
        mask[i, :L] = 1
    return torch.from_numpy(ids).to(dev), torch.from_numpy(mask).to(dev)


# ---- corpus distributions for the data-sensitivity sweep -------------------------------------------------------
# The headline corpus (bench.gen_block) is i.i.d. 0.05 N(0,1) + 0.02.  The two data-dependent fast paths -- the f16 pre-filter
# of the dense search (how many rows pass the running threshold, how many queries the proof sends to the second pass) and the
# matrix-core RQ shortlist (how many row-levels are ambiguous) -- are also timed on corpora shaped like what the reference's
# towers emit: clustered, un-normalised with a large common component (MEVI/document_encoder.py:118,139 returns
# last_hidden_state[:, 0, :] as is), and with duplicated passages (MS MARCO has exact and near duplicates).  Every generator
# is seeded per 65536-row block, on the device, like bench.gen_block.
CORPUS_KINDS = ("iid", "clustered", "ance_scale", "duplicates")
CORPUS_BLOCK = 65536
N_CLUSTERS = 10_000


def _cluster_tables(dev, dim, n_clusters, sigma_between, seed):
    g = torch.Generator(device=dev).manual_seed(seed)
    centres = sigma_between * torch.randn((n_clusters, dim), device=dev, generator=g)
    # cluster popularity ~ 1 / (rank + 10)^0.7: the largest clusters hold several thousand rows of an 8.8 M corpus, the median
    # a few hundred -- a query planted in a large cluster has its whole top-1000 inside ONE cluster (dense scores at rank k)
    w = 1.0 / (torch.arange(n_clusters, device=dev, dtype=torch.float32) + 10.0) ** 0.7
    return centres, w / w.sum()


def _assign_runs(rows, weights, g, mean_run=8):
    """Cluster of every row of a block: constant over runs of geometric length (mean `mean_run`) -- the passages of one web
    document sit next to each other in MS MARCO's collection -- each run's cluster drawn by popularity."""
    dev = weights.device
    new_run = torch.rand((rows,), device=dev, generator=g) < 1.0 / mean_run
    new_run[0] = True
    run_id = torch.cumsum(new_run.to(torch.int64), 0) - 1
    n_runs = int(run_id[-1].item()) + 1
    cl = torch.multinomial(weights, n_runs, replacement=True, generator=g)
    return cl[run_id]


def corpus_spec(kind, dev, dim=d, n_clusters=N_CLUSTERS):
    """What every block of corpus `kind` shares (cluster centres, popularity, the common component, per-dimension scales)."""
    if kind not in CORPUS_KINDS:
        raise ValueError(f"unknown corpus kind {kind!r} (one of {CORPUS_KINDS})")
    spec = {"kind": kind, "dim": dim}
    if kind == "clustered":
        spec["centres"], spec["weights"] = _cluster_tables(dev, dim, n_clusters, 0.05, 77)
        spec.update(sigma_within=0.012, shift=0.02)
    elif kind == "ance_scale":
        # row norms 10-14: a common component of norm ~11 (+-0.4 per dimension), a clustered residual of norm ~3.7, a per-row
        # scale jitter of 8 %, and four outlier dimensions with 10x the spread (dense retrievers' embeddings have such)
        spec["centres"], spec["weights"] = _cluster_tables(dev, dim, n_clusters, 0.12, 78)
        g = torch.Generator(device=dev).manual_seed(79)
        spec["common"] = 0.4 * (2.0 * (torch.rand((dim,), device=dev, generator=g) < 0.5).float() - 1.0)
        dscale = torch.ones((dim,), device=dev)
        dscale[torch.randperm(dim, device=dev, generator=g)[:4]] = 10.0
        spec.update(sigma_within=0.06, dim_scale=dscale, row_jitter=0.08)
    return spec


def corpus_block(spec, b, n_docs, block=CORPUS_BLOCK):
    """Rows [b * block, ...) of the corpus described by `spec` (f32 [rows, dim] on the spec's device)."""
    kind, dim = spec["kind"], spec["dim"]
    rows = min(block, n_docs - b * block)
    if kind in ("iid", "duplicates"):
        dev = spec.get("device")
        g = torch.Generator(device=dev).manual_seed(10_000 + b)           # bench.gen_block's stream
        return 0.05 * torch.randn((rows, dim), device=dev, generator=g) + 0.02
    dev = spec["centres"].device
    g = torch.Generator(device=dev).manual_seed(20_000 + b + (1_000_000 if kind == "ance_scale" else 0))
    cl = _assign_runs(rows, spec["weights"], g)
    x = spec["centres"][cl] + spec["sigma_within"] * torch.randn((rows, dim), device=dev, generator=g)
    if kind == "clustered":
        return x + spec["shift"]
    x = x * spec["dim_scale"] + spec["common"]
    return x * (1.0 + spec["row_jitter"] * torch.randn((rows, 1), device=dev, generator=g))


def corpus(kind, dev, n_docs, dim=d, block=CORPUS_BLOCK, n_clusters=N_CLUSTERS):
    """(docs f32 [n_docs, dim] on `dev`, info) -- the whole corpus of `kind`.  'duplicates': the i.i.d. corpus with 1 % of the
    rows overwritten by exact copies of other rows and another 1 % by near copies (relative noise 1e-3)."""
    spec = corpus_spec(kind, dev, dim, n_clusters)
    spec["device"] = dev
    out = torch.empty((n_docs, dim), dtype=torch.float32, device=dev)
    for b in range((n_docs + block - 1) // block):
        blk = corpus_block(spec, b, n_docs, block)
        out[b * block:b * block + blk.shape[0]] = blk
    info = {"kind": kind, "rows": n_docs, "dim": dim}
    if kind == "duplicates" and n_docs >= 8:
        g = torch.Generator(device=dev).manual_seed(4242)
        n_dup = max(1, n_docs // 100)
        perm = torch.randperm(n_docs, device=dev, generator=g)
        dst_exact, dst_near, src = perm[:n_dup], perm[n_dup:2 * n_dup], perm[2 * n_dup:4 * n_dup]
        out[dst_exact] = out[src[:n_dup]]
        out[dst_near] = out[src[n_dup:]] * (1.0 + 1e-3 * torch.randn((n_dup, dim), device=dev, generator=g))
        info.update(exact_duplicates=n_dup, near_duplicates=n_dup, dup_src=src[:n_dup], dup_dst=dst_exact)
    if kind in ("clustered", "ance_scale"):
        info.update(clusters=n_clusters, sigma_within=spec["sigma_within"])
    nr = out[:: max(1, n_docs // 4096)].norm(dim=1)
    info["row_norm"] = {"mean": float(nr.mean()), "min": float(nr.min()), "max": float(nr.max())}
    return out, info


def corpus_queries(kind, docs, nq, info=None, seed=1234):
    """(queries f32 [nq, dim], planted row of every query): a query is a corpus row spread over the WHOLE corpus plus noise
    of half the within-cluster spread -- inside its cluster, not on top of its row.  'duplicates': every other query is planted
    on a row that has an exact copy (its top ranks then hold a tie run: order by ascending id)."""
    n_docs, dim = docs.shape
    dev = docs.device
    g = torch.Generator(device=dev).manual_seed(seed)
    ids = (torch.arange(nq, device=dev, dtype=torch.int64) * max(1, n_docs // max(nq, 1)) + 17) % n_docs
    if kind == "duplicates" and info is not None and "dup_src" in info:
        src = info["dup_src"]
        pick = src[(torch.arange((nq + 1) // 2, device=dev) * 7919) % src.shape[0]]
        ids[::2] = pick[: ids[::2].shape[0]]
    noise = {"iid": 0.005, "duplicates": 0.005, "clustered": 0.006, "ance_scale": 0.03}[kind]
    q = docs[ids] + noise * torch.randn((nq, dim), device=dev, generator=g)
    return q.contiguous(), ids
