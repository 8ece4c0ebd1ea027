"""BERT-family towers of the dense arm (coCondenser / AR2 / ERNIE-2.0: `DocumentEncoder` with mtype 'bert',
MEVI/document_encoder.py:43-44,104-123; model = the vendored `transformers.modeling_bert.BertModel`):
reps = last_hidden_state[:, 0, :] of a post-LN encoder with absolute position embeddings, erf-GELU and biased
linear layers.  Everything runs on the HIP kernels (GEMM with bias / GELU / residual epilogues, add+LayerNorm,
attention with the 1/sqrt(d) scale and the key mask)."""
import torch

from . import ops
from .t5 import DEVICE_PASS_TOKENS, packed_offsets, varlen_ok


def _dev(w, name, device):
    import numpy as np

    t = w[name]
    t = torch.from_numpy(np.ascontiguousarray(t)) if not torch.is_tensor(t) else t
    return t.detach().to(device=device, dtype=torch.float32).contiguous()


class BertEncoder:
    """BertModel (embeddings + encoder, no pooler).  Weights: the reference's state_dict names (optionally under
    `prefix`, e.g. 'bert.')."""

    def __init__(self, w, num_layers, num_heads, eps=1e-12, device=None, prefix=""):
        self.dev = torch.device(device if device is not None else "cuda")
        self.H, self.eps = num_heads, eps
        g = lambda n: _dev(w, prefix + n, self.dev)  # noqa: E731
        self.word = g("embeddings.word_embeddings.weight")
        self.pos = g("embeddings.position_embeddings.weight")
        self.type0 = g("embeddings.token_type_embeddings.weight")[0].contiguous()
        task = prefix + "embeddings.task_type_embeddings.weight"       # ERNIE with use_task_id: task id 0
        if task in w:
            self.type0 = (self.type0 + _dev(w, task, self.dev)[0]).contiguous()
        self.emb_ln = (g("embeddings.LayerNorm.weight"), g("embeddings.LayerNorm.bias"))
        self.layers = []
        for l in range(num_layers):
            p = f"encoder.layer.{l}."
            self.layers.append(dict(
                wqkv=torch.cat([g(p + f"attention.self.{n}.weight") for n in ("query", "key", "value")]).contiguous(),
                bqkv=torch.cat([g(p + f"attention.self.{n}.bias") for n in ("query", "key", "value")]).contiguous(),
                wo=g(p + "attention.output.dense.weight"), bo=g(p + "attention.output.dense.bias"),
                ln1=(g(p + "attention.output.LayerNorm.weight"), g(p + "attention.output.LayerNorm.bias")),
                wi=g(p + "intermediate.dense.weight"), bi=g(p + "intermediate.dense.bias"),
                wo2=g(p + "output.dense.weight"), bo2=g(p + "output.dense.bias"),
                ln2=(g(p + "output.LayerNorm.weight"), g(p + "output.LayerNorm.bias"))))
        ops.prepare_weights(self.layers, ("wqkv", "wo", "wi", "wo2"))
        self.d = self.word.shape[1]
        for l, L in enumerate(self.layers):   # |V| bound of a layer: its input is the previous LayerNorm's output (ops.ctx_bound)
            ln_w, ln_b = self.emb_ln if l == 0 else self.layers[l - 1]["ln2"]
            L["vb"] = ops.ctx_bound(ops.norm_out_bound(ln_w, self.d, ln_b), L["wqkv"], v_bias=L["bqkv"])
        self.pack = True   # padding-free per-token operators (see forward)
        self.dh = self.d // num_heads

    def forward(self, input_ids, attention_mask, pack=None):
        """input_ids / attention_mask i64[B, S] (S <= 256, token types all 0) -> last hidden state f32[B, S, d].

        pack (default: self.pack): per-token operators run on the real tokens only, the padded layout is used for
        attention alone (as mevi_amd.t5.EncoderStack.forward); padded positions of the result are 0."""
        B, S = input_ids.shape
        d = self.d
        pack = self.pack if pack is None else pack
        idx = None
        if pack:
            idx = torch.nonzero(attention_mask.reshape(-1) != 0).view(-1)
            if idx.numel() == 0 or idx.numel() > 0.9 * B * S:
                idx = None
        scale = float(self.dh) ** -0.5
        if idx is None:
            tok, pos = input_ids.reshape(-1), self.pos[:S].repeat(B, 1)
        else:
            tok, pos = input_ids.reshape(-1)[idx], ops.gather_rows(self.pos, idx % S)
        x = ops.gather_rows(self.word, tok)
        x = ops.add_layernorm(x, pos, self.emb_ln[0], self.emb_ln[1], eps=self.eps, cvec=self.type0)
        seq_off, longest = packed_offsets(attention_mask) if idx is not None else (None, 0)
        varlen = seq_off is not None and varlen_ok(longest, self.dh)    # right-padded: attend on the packed rows
        qkv = None if idx is None or varlen else torch.zeros((B * S, 3 * d), dtype=torch.float32, device=x.device)
        for L in self.layers:
            xg = ops.gemm_input(x)
            if idx is None:
                q3 = ops.linear(xg, L["wqkv"], bias=L["bqkv"]).view(B, S, 3 * d)
            elif varlen:
                q2 = ops.linear(xg, L["wqkv"], bias=L["bqkv"])
                ctx = ops.attention_varlen(q2[:, :d], q2[:, d:2 * d], q2[:, 2 * d:], seq_off, longest, self.H, scale=scale,
                                           split_bound=L["vb"])
            else:
                q3 = ops.scatter_rows(ops.linear(xg, L["wqkv"], bias=L["bqkv"]), idx, qkv).view(B, S, 3 * d)
            if not varlen:
                vb = L["vb"] if idx is None else None      # the scattered form gathers its context rows: f32
                ctx = ops.attention(q3[:, :, :d], q3[:, :, d:2 * d], q3[:, :, 2 * d:], self.H, key_mask=attention_mask,
                                    scale=scale, split_bound=vb)
                if vb is None:
                    ctx = ctx.view(B * S, d)
                if idx is not None:
                    ctx = ops.gather_rows(ctx, idx)
            a = ops.linear(ctx, L["wo"], bias=L["bo"])
            x = ops.add_layernorm(a, x, L["ln1"][0], L["ln1"][1], eps=self.eps)
            h = ops.linear(ops.gemm_input(x), L["wi"], bias=L["bi"], gelu=True, for_gemm=True)
            o = ops.linear(h, L["wo2"], bias=L["bo2"])
            x = ops.add_layernorm(o, x, L["ln2"][0], L["ln2"][1], eps=self.eps)
        if idx is None:
            return x.view(B, S, d)
        out = torch.zeros((B * S, d), dtype=torch.float32, device=x.device)
        return ops.scatter_rows(x, idx, out).view(B, S, d)


class BertTower:
    """`DocumentEncoder` of mtype 'bert': reps = last_hidden_state[:, 0, :] (normalize=False).  `weights_p` gives the
    passage model its own weights (AR2: ctx_model / question_model); tied otherwise."""

    def __init__(self, weights_q, num_layers, num_heads, weights_p=None, eps=1e-12, device=None, batch_size=None, prefix=""):
        self.dev = torch.device(device if device is not None else "cuda")
        self.lm_q = BertEncoder(weights_q, num_layers, num_heads, eps, self.dev, prefix)
        self.lm_p = self.lm_q if weights_p is None else BertEncoder(weights_p, num_layers, num_heads, eps, self.dev, prefix)
        self.batch_size = batch_size
        self.dim = self.lm_q.d         # width of the embeddings this tower emits

    def _encode(self, model, items):
        ids = items["input_ids"].to(self.dev, torch.int64)
        mask = items["attention_mask"].to(self.dev, torch.int64)
        outs = []
        step = self.batch_size or max(1, DEVICE_PASS_TOKENS // max(1, ids.shape[1]))
        for a in range(0, ids.shape[0], step):
            i, m = ids[a:a + step].contiguous(), mask[a:a + step].contiguous()
            outs.append(model.forward(i, m)[:, 0, :].contiguous())
        return torch.cat(outs) if outs else torch.empty((0, model.d), device=self.dev)

    def encode_query(self, qry):
        return self._encode(self.lm_q, qry)

    def encode_passage(self, psg):
        return self._encode(self.lm_p, psg)

    encode = encode_query
