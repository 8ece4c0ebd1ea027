"""TEST INFRASTRUCTURE ONLY -- torch-fp32 restatement of the vendored BertModel forward the reference's 'bert' towers
run (MEVI/transformers/modeling_bert.py: BertEmbeddings :166-216, BertSelfAttention :219-301, BertSelfOutput,
BertIntermediate (erf gelu), BertOutput, BertLayer :388-449; extended mask -10000, modeling_utils.py:213-270).
Pinned against the reference's own outputs (tests/golden/g8_bert_tower.npz)."""
import math

import torch
import torch.nn.functional as F


def load_weights(npz, prefix="w."):
    return {k[len(prefix):]: torch.from_numpy(npz[k]) for k in npz.files if k.startswith(prefix)}


def encoder(W, cfg, input_ids, attention_mask):
    B, S = input_ids.shape
    H, eps = cfg["num_attention_heads"], cfg["layer_norm_eps"]
    x = W["embeddings.word_embeddings.weight"][input_ids] + W["embeddings.position_embeddings.weight"][:S][None] \
        + W["embeddings.token_type_embeddings.weight"][0][None, None]
    d = x.shape[-1]
    x = F.layer_norm(x, (d,), W["embeddings.LayerNorm.weight"], W["embeddings.LayerNorm.bias"], eps)
    ext = (1.0 - attention_mask[:, None, None, :].float()) * -10000.0
    dh = d // H
    for l in range(cfg["num_hidden_layers"]):
        p = f"encoder.layer.{l}."
        lin = lambda t, n: F.linear(t, W[p + n + ".weight"], W[p + n + ".bias"])  # noqa: E731
        q = lin(x, "attention.self.query").view(B, S, H, dh).transpose(1, 2)
        k = lin(x, "attention.self.key").view(B, S, H, dh).transpose(1, 2)
        v = lin(x, "attention.self.value").view(B, S, H, dh).transpose(1, 2)
        s = q @ k.transpose(-1, -2) / math.sqrt(dh) + ext
        ctx = (F.softmax(s, -1) @ v).transpose(1, 2).reshape(B, S, d)
        x = F.layer_norm(lin(ctx, "attention.output.dense") + x, (d,), W[p + "attention.output.LayerNorm.weight"],
                         W[p + "attention.output.LayerNorm.bias"], eps)
        h = lin(x, "intermediate.dense")
        h = h * 0.5 * (1.0 + torch.erf(h / math.sqrt(2.0)))
        x = F.layer_norm(lin(h, "output.dense") + x, (d,), W[p + "output.LayerNorm.weight"],
                         W[p + "output.LayerNorm.bias"], eps)
    return x


def tower_encode(W, cfg, input_ids, attention_mask):
    return encoder(W, cfg, input_ids, attention_mask)[:, 0, :]
