"""Oracle for the T5 side of MEVI (TEST INFRASTRUCTURE ONLY -- see oracle/__init__.py).

A plain PyTorch fp32 (CPU) restatement of the arithmetic of the MEVI-modified T5
(MEVI/transformers/modeling_t5.py) and of the constrained beam search
(MEVI/transformers/generation_utils.py), written functionally over a flat weight dict whose
keys are the reference's state_dict names.  Pinned against the reference's own outputs in
tests/golden/g1_*.npz, g2_*.npz, g3_*.npz (tests/test_t5_oracle_cpu.py).

    relative_position_bucket   modeling_t5.py:241-289
    rmsnorm                    T5LayerNorm, modeling_t5.py:155-171
    attention                  T5Attention.forward, modeling_t5.py:322-418 (no 1/sqrt(d) scaling)
    encoder / decoder          T5Stack + T5Block, modeling_t5.py:494-580, 657-813
    tower_encode               DocumentEncoder.encode, document_encoder.py:104-120
    adaptor / adaptive_logits  modeling_t5.py:1647-1689 (nn.TransformerDecoder, post-LN)
    nci_generate               the validated restatement of generate + _generate_beam_search
                               (SURVEY 8(a'), generation_utils.py:116-577, 709-1011)
"""
import math

import numpy as np
import torch
import torch.nn.functional as F

NEG = -1e9


def relative_position_bucket(rel, bidirectional, num_buckets=32, max_distance=128):
    """rel = memory_position - query_position (integer ndarray)."""
    rel = np.asarray(rel, dtype=np.int64)
    ret = np.zeros_like(rel)
    n = num_buckets
    if bidirectional:
        n //= 2
        ret = ret + (rel > 0).astype(np.int64) * n
        rel = np.abs(rel)
    else:
        rel = -np.minimum(rel, 0)
    max_exact = n // 2
    small = rel < max_exact
    # the reference evaluates the log in float32 (torch.log of a float tensor)
    safe = np.maximum(rel, 1)  # rel = 0 is always "small"; avoid log(0)
    if True:
        large = max_exact + (np.log(safe.astype(np.float32) / np.float32(max_exact))
                             / np.float32(math.log(max_distance / max_exact)) * np.float32(n - max_exact)).astype(np.int64)
    large = np.minimum(large, n - 1)
    return ret + np.where(small, rel, large)


def rmsnorm(x, w, eps):
    var = x.float().pow(2).mean(-1, keepdim=True)
    return w * (x / torch.sqrt(var + eps))


def _heads(x, H):
    b, s, _ = x.shape
    return x.view(b, s, H, -1).transpose(1, 2)


def attention(W, pre, x, kv, bias, H):
    """T5 attention sub-layer (without norm / residual).  bias: additive [b or 1, H, q, k]."""
    q = _heads(x @ W[pre + ".q.weight"].T, H)
    src = x if kv is None else kv
    k = _heads(src @ W[pre + ".k.weight"].T, H)
    v = _heads(src @ W[pre + ".v.weight"].T, H)
    scores = q @ k.transpose(-1, -2) + bias
    p = F.softmax(scores.float(), dim=-1)
    ctx = (p @ v).transpose(1, 2).reshape(x.shape[0], x.shape[1], -1)
    return ctx @ W[pre + ".o.weight"].T


def position_bias(W, pre, qlen, klen, bidirectional, num_buckets=32):
    ctx = np.arange(qlen)[:, None]
    mem = np.arange(klen)[None, :]
    b = relative_position_bucket(mem - ctx, bidirectional, num_buckets)
    table = W[pre + ".relative_attention_bias.weight"]            # [buckets, H]
    return table[torch.from_numpy(b)].permute(2, 0, 1).unsqueeze(0)  # [1, H, q, k]


def encoder(W, cfg, ids, mask, prefix="encoder", emb="shared.weight", return_all=False):
    H, eps = cfg["num_heads"], cfg["layer_norm_epsilon"]
    x = W[emb][ids]
    S = ids.shape[1]
    bias = position_bias(W, f"{prefix}.block.0.layer.0.SelfAttention", S, S, True, cfg["relative_attention_num_buckets"])
    bias = bias + (1.0 - mask[:, None, None, :].float()) * NEG
    hs = [x]
    for l in range(cfg["num_layers"]):
        p = f"{prefix}.block.{l}.layer"
        x = x + attention(W, f"{p}.0.SelfAttention", rmsnorm(x, W[f"{p}.0.layer_norm.weight"], eps), None, bias, H)
        h = rmsnorm(x, W[f"{p}.1.layer_norm.weight"], eps)
        x = x + F.relu(h @ W[f"{p}.1.DenseReluDense.wi.weight"].T) @ W[f"{p}.1.DenseReluDense.wo.weight"].T
        hs.append(x)
    x = rmsnorm(x, W[f"{prefix}.final_layer_norm.weight"], eps)
    hs[-1] = x  # HF reports the normed state as the last hidden state
    return (x, hs) if return_all else x


def decoder(W, cfg, dec_ids, enc, enc_mask, prefix="decoder", emb="decode_embeddings.weight", n_layers=None,
            return_all=False):
    """Full-prefix (no cache) decoder stack, causal self-attention + cross-attention."""
    H, eps = cfg["num_heads"], cfg["layer_norm_epsilon"]
    n_layers = cfg["num_decoder_layers"] if n_layers is None else n_layers
    x = W[emb][dec_ids]
    T = dec_ids.shape[1]
    causal = torch.tril(torch.ones(T, T))
    sbias = position_bias(W, f"{prefix}.block.0.layer.0.SelfAttention", T, T, False,
                          cfg["relative_attention_num_buckets"]) + (1.0 - causal)[None, None] * NEG
    xbias = (1.0 - enc_mask[:, None, None, :].float()) * NEG   # zeros + inverted mask (modeling_t5.py:386-389)
    hs = [x]
    for l in range(n_layers):
        p = f"{prefix}.block.{l}.layer"
        x = x + attention(W, f"{p}.0.SelfAttention", rmsnorm(x, W[f"{p}.0.layer_norm.weight"], eps), None, sbias, H)
        x = x + attention(W, f"{p}.1.EncDecAttention", rmsnorm(x, W[f"{p}.1.layer_norm.weight"], eps), enc, xbias, H)
        h = rmsnorm(x, W[f"{p}.2.layer_norm.weight"], eps)
        x = x + F.relu(h @ W[f"{p}.2.DenseReluDense.wi.weight"].T) @ W[f"{p}.2.DenseReluDense.wo.weight"].T
        hs.append(x)
    x = rmsnorm(x, W[f"{prefix}.final_layer_norm.weight"], eps)
    hs[-1] = x
    return (x, hs) if return_all else x


def tower_encode(W, cfg, ids, mask):
    """Twin-tower query embedding: T5Model encoder + ONE decoder step on token 0, hidden[:, 0, :]."""
    enc = encoder(W, cfg, ids, mask)
    dec_ids = torch.zeros((ids.shape[0], 1), dtype=torch.long)
    return decoder(W, cfg, dec_ids, enc, mask, emb="shared.weight")[:, 0, :]


# ---- PAWA adaptive decoder head ---------------------------------------------------------
def _mha(W, pre, x, mem, nhead, mask=None):
    """torch.nn.MultiheadAttention forward, batch-first [b, t, d] here (the module is seq-first)."""
    d = x.shape[-1]
    w, b = W[pre + ".in_proj_weight"], W[pre + ".in_proj_bias"]
    q = x @ w[:d].T + b[:d]
    k = mem @ w[d:2 * d].T + b[d:2 * d]
    v = mem @ w[2 * d:].T + b[2 * d:]
    hd = d // nhead
    q = _heads(q, nhead) * (hd ** -0.5)
    k, v = _heads(k, nhead), _heads(v, nhead)
    s = q @ k.transpose(-1, -2)
    if mask is not None:
        s = s + mask
    ctx = (F.softmax(s, dim=-1) @ v).transpose(1, 2).reshape(x.shape)
    return ctx @ W[pre + ".out_proj.weight"].T + W[pre + ".out_proj.bias"]


def adaptor(W, cfg, tok_emb):
    """nn.TransformerDecoder(TransformerDecoderLayer(d, nhead=8), L) on the decode-token embeddings with
    the single learned memory vector adaptor_embeddings (modeling_t5.py:1252-1255, 1650-1665)."""
    b, t, d = tok_emb.shape
    mem = W["adaptor_embeddings"].reshape(1, 1, d).expand(b, 1, d)
    causal = torch.full((t, t), float("-inf")).triu(1)
    x = tok_emb
    for l in range(cfg["adaptor_layer_num"]):
        p = f"adaptor.layers.{l}"
        x = F.layer_norm(x + _mha(W, p + ".self_attn", x, x, 8, causal), (d,), W[p + ".norm1.weight"], W[p + ".norm1.bias"], 1e-5)
        x = F.layer_norm(x + _mha(W, p + ".multihead_attn", x, mem, 8), (d,), W[p + ".norm2.weight"], W[p + ".norm2.bias"], 1e-5)
        ff = F.relu(x @ W[p + ".linear1.weight"].T + W[p + ".linear1.bias"]) @ W[p + ".linear2.weight"].T + W[p + ".linear2.bias"]
        x = F.layer_norm(x + ff, (d,), W[p + ".norm3.weight"], W[p + ".norm3.bias"], 1e-5)
    return x


def position_valid_mask(pos, K, V):
    """select_valid_embedding (modeling_t5.py:1578-1603): 0 on {1} U [2+pos*K, 2+(pos+1)*K), -1e9 elsewhere."""
    m = torch.full((V,), NEG)
    m[1] = 0.0
    m[2 + pos * K: 2 + (pos + 1) * K] = 0.0
    return m


def nci_last_logits(W, cfg, dec_ids, enc, enc_mask):
    """Masked logits of the LAST decoder position, f32[n, V] (what generation consumes, generation_utils.py:764)."""
    d, K = cfg["d_model"], cfg["K"]
    V = W["lm_head.weight"].shape[0]
    seq = decoder(W, cfg, dec_ids, enc, enc_mask)[:, -1, :] * (d ** -0.5)        # modeling_t5.py:1607
    a = adaptor(W, cfg, W["decode_embeddings.weight"][dec_ids])[:, -1, :]         # [n, d]
    w_adapt = (a @ W["adaptor_linear.weight"].T).reshape(-1, d, V)                # out index = d_idx * V + v
    head = w_adapt + W["lm_head.weight"].T[None]
    logits = torch.bmm(seq[:, None, :], head)[:, 0, :]
    return logits + position_valid_mask(dec_ids.shape[1] - 1, K, V)[None]


def nci_generate(W, cfg, ids, mask, beams, length_penalty=0.8, return_steps=False):
    """Constrained beam search over the shared-layer RQ tree.  Returns (decoded i64[B*R, M+2],
    scores f64[B*R] descending per query, enc f32[B,S,d][, step logits])."""
    M, K = cfg["M"], cfg["K"]
    B = ids.shape[0]
    enc = encoder(W, cfg, ids, mask)
    out_tok, out_sc, steps = [], [], []
    for b in range(B):
        e, m = enc[b:b + 1], mask[b:b + 1]
        prefix = torch.zeros((1, 1), dtype=torch.long)
        score = torch.zeros(1)
        for p in range(M):
            n = prefix.shape[0]
            logits = nci_last_logits(W, cfg, prefix, e.expand(n, -1, -1), m.expand(n, -1))
            if return_steps:
                steps.append((b, p, logits))
            lsm = F.log_softmax(logits, dim=-1)[:, 2 + p * K: 2 + (p + 1) * K]
            cand = (score[:, None] + lsm).reshape(-1)
            top = torch.topk(cand, min(beams, cand.numel()))
            parent, code = top.indices // K, top.indices % K
            prefix = torch.cat([prefix[parent], (2 + p * K + code)[:, None]], 1)
            score = top.values
        n = prefix.shape[0]
        logits = nci_last_logits(W, cfg, prefix, e.expand(n, -1, -1), m.expand(n, -1))
        if return_steps:
            steps.append((b, M, logits))
        final = score + F.log_softmax(logits, dim=-1)[:, 1]
        res = final.double() / (M + 1) ** length_penalty      # f32 sum widened to double, then divided
        order = torch.argsort(-res, stable=True)
        out_tok.append(torch.cat([prefix[order], torch.ones((n, 1), dtype=torch.long)], 1))
        out_sc.append(res[order])
    dec, sc = torch.cat(out_tok), torch.cat(out_sc)
    return (dec, sc, enc, steps) if return_steps else (dec, sc, enc)


def nci_generate_tree(W, cfg, ids, mask, beams, paths, length_penalty=0.8):
    """The same search under a GENERIC prefix tree: `paths` i[n, M] = the existing code paths (TreeBuilder(share_sons=False)
    .add per path, MEVI/main_models.py:50-63; trie walk MEVI/transformers/generation_utils.py:803-818).  Restated as the
    reference runs it: all R beams from the first step, beams 1..R-1 seeded with -1e9 (generation_utils.py:752-756), every
    beam continued along the children of its trie node only (the log-softmax spans eos and all K level codes: the tree mask
    is added afterwards), the R best of the candidates kept (ties: lower beam, lower code).  Every beam sits on a path of
    the trie, so the reference's "path not in tree -> eos" branch is never taken; the -1e9 beams surface in the result only
    when the trie holds fewer than R paths.  Returns (decoded i64[B*R, M+2], scores f64[B*R])."""
    M, K = cfg["M"], cfg["K"]
    pset = [set() for _ in range(M + 1)]
    for pth in np.asarray(paths).tolist():
        for p in range(M + 1):
            pset[p].add(tuple(pth[:p]))
    B, R = ids.shape[0], beams
    enc = encoder(W, cfg, ids, mask)
    out_tok, out_sc = [], []
    for b in range(B):
        e, m = enc[b:b + 1], mask[b:b + 1]
        prefix = torch.zeros((R, 1), dtype=torch.long)
        score = torch.full((R,), -1e9)
        score[0] = 0.0
        codes = [() for _ in range(R)]
        for p in range(M):
            logits = nci_last_logits(W, cfg, prefix, e.expand(R, -1, -1), m.expand(R, -1))
            lsm = F.log_softmax(logits, dim=-1)[:, 2 + p * K: 2 + (p + 1) * K]
            cand = score[:, None] + lsm
            allowed = torch.tensor([[codes[r] + (c,) in pset[p + 1] for c in range(K)] for r in range(R)])
            cand = torch.where(allowed, cand, torch.full_like(cand, -float("inf"))).reshape(-1)
            order = torch.argsort(-cand, stable=True)[:R]          # (score desc, flat index asc)
            parent, code = order // K, order % K
            prefix = torch.cat([prefix[parent], (2 + p * K + code)[:, None]], 1)
            codes = [codes[int(r)] + (int(c),) for r, c in zip(parent, code)]
            score = cand[order]
        logits = nci_last_logits(W, cfg, prefix, e.expand(R, -1, -1), m.expand(R, -1))
        final = score + F.log_softmax(logits, dim=-1)[:, 1]
        res = final.double() / (M + 1) ** length_penalty
        order = torch.argsort(-res, stable=True)
        out_tok.append(torch.cat([prefix[order], torch.ones((R, 1), dtype=torch.long)], 1))
        out_sc.append(res[order])
    return torch.cat(out_tok), torch.cat(out_sc), enc


def nci_generate_all(W, cfg, ids, mask, length_penalty=0.8):
    """_generate_all (MEVI/transformers/generation_utils.py:1013-1136; generate(..., eval_all_documents=True), the
    `use_topic_model` ablation): the score of EVERY code path, f32 [B, K**M] with path index sum_p c_p K**(M-1-p):
    sum of the per-level log-softmax terms (over the position's valid columns) plus the final eos term, divided by
    (max_length - 1) ** length_penalty = (M + 1) ** length_penalty -- all in f32 tensors, as the reference."""
    M, K = cfg["M"], cfg["K"]
    enc = encoder(W, cfg, ids, mask)
    out = []
    for b in range(ids.shape[0]):
        e, m = enc[b:b + 1], mask[b:b + 1]
        prefix = torch.zeros((1, 1), dtype=torch.long)
        score = torch.zeros(1)
        for p in range(M + 1):
            n = prefix.shape[0]
            rows = []
            for a in range(0, n, 128):                       # local_batch_size (generation_utils.py:1029)
                rows.append(nci_last_logits(W, cfg, prefix[a:a + 128], e.expand(min(128, n - a), -1, -1),
                                            m.expand(min(128, n - a), -1)))
            lsm = F.log_softmax(torch.cat(rows), dim=-1)
            if p == M:
                score = (lsm[:, 1:2] + score[:, None]).view(-1)
                break
            score = (lsm[:, 2 + p * K: 2 + (p + 1) * K] + score[:, None]).view(-1)
            new = torch.arange(2 + p * K, 2 + (p + 1) * K).repeat(n)[:, None]
            prefix = torch.cat([prefix[:, None, :].repeat(1, K, 1).view(-1, p + 1), new], dim=-1)
        out.append(score / (M + 1) ** length_penalty)
    return torch.stack(out), enc


def decode_token(decoded, K):
    """main_models.decode_token for codebook models (main_models.py:117-136): strip bos/eos, undo the
    position offset, clamp negatives to 0 -> codes i64[n, M]."""
    seqs = decoded[:, 1:-1] - 2
    seqs = seqs - torch.arange(seqs.shape[1]) * K
    return seqs.clamp(min=0)


def load_weights(npz):
    return {k[2:]: torch.from_numpy(np.asarray(npz[k])) for k in npz.files if k.startswith("w.")}
