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
