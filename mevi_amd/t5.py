"""T5 stacks on MI355X: encoder, KV-cached decoder step, twin-tower query encoder.

Arithmetic follows the MEVI-modified T5 (MEVI/transformers/modeling_t5.py): pre-RMSNorm residual
blocks (T5Block :494-580), attention without 1/sqrt(d) scaling and with the layer-0 relative
position bias shared by all layers (T5Attention :203-418, T5Stack :781-785), bias-free
relu(x Wi^T) Wo^T feed-forward (:174-186), final RMSNorm (:806).  Every matmul is the f32-MFMA
GEMM (ops.linear), everything else the wave-per-row kernels of csrc/t5_ops.hip.

Reference surfaces mirrored:
  DocumentEncoder.encode / encode_query   MEVI/document_encoder.py:104-123  -> TwinTower
"""
import math
import os

import numpy as np
import torch

from . import ops


def relative_position_bucket(rel, bidirectional, num_buckets=32, max_distance=128):
    """T5 bucket of rel = memory_pos - query_pos (modeling_t5.py:241-289); the log runs in float32."""
    rel = np.asarray(rel, dtype=np.int64)
    n = num_buckets
    ret = np.zeros_like(rel)
    if bidirectional:
        n //= 2
        ret = ret + (rel > 0) * n
        rel = np.abs(rel)
    else:
        rel = np.maximum(-rel, 0)
    max_exact = n // 2
    scale = np.float32(n - max_exact) / np.float32(math.log(max_distance / max_exact))
    big = max_exact + (np.log(np.maximum(rel, 1).astype(np.float32) / np.float32(max_exact))
                       / np.float32(math.log(max_distance / max_exact)) * np.float32(n - max_exact)).astype(np.int64)
    del scale
    return ret + np.where(rel < max_exact, rel, np.minimum(big, n - 1))


def bias_table(rel_weight, qlen, klen, bidirectional, num_buckets=32):
    """[H, qlen, klen] additive bias from relative_attention_bias.weight [buckets, H] (compute_bias :291-304)."""
    b = relative_position_bucket(np.arange(klen)[None, :] - np.arange(qlen)[:, None], bidirectional, num_buckets)
    idx = torch.from_numpy(b).to(rel_weight.device)
    return rel_weight[idx].permute(2, 0, 1).contiguous()


class T5Dims:
    def __init__(self, d_model=768, d_ff=3072, num_heads=12, d_kv=64, num_layers=12, num_decoder_layers=12,
                 layer_norm_epsilon=1e-6, relative_attention_num_buckets=32, **unused):
        self.d_model, self.d_ff, self.num_heads, self.d_kv = d_model, d_ff, num_heads, d_kv
        self.num_layers, self.num_decoder_layers = num_layers, num_decoder_layers
        self.eps, self.buckets = layer_norm_epsilon, relative_attention_num_buckets
        self.inner = num_heads * d_kv


def _dev(w, key, device):
    t = w[key]
    t = t if torch.is_tensor(t) else torch.from_numpy(np.asarray(t))
    return t.to(device=device, dtype=torch.float32).contiguous()


class _PreNormBlocks:
    """The linear layers of T5's pre-norm residual blocks in their two forms: with the T5LayerNorm folded into the projection
    it feeds (self.fold: the residual stream is an ops.ResidualRows) or as its own pass (an f32 tensor)."""

    def _nl(self, x, L, ln, wk, **kw):
        """act(rmsnorm(x) W^T)"""
        if self.fold:
            return ops.linear_normed(x, L[wk], self.d.eps, **kw)
        return ops.linear(ops.rmsnorm(x, L[ln], self.d.eps, for_gemm=True), L[wk], **kw)

    def _rl(self, a, L, wk, x):
        """x + a W^T"""
        if self.fold:
            return ops.linear_residual(a, L[wk], x)
        return ops.linear(a, L[wk], residual=x)

    def _start(self, x):
        return ops.residual_start(x) if self.fold else x

    def _final(self, x):
        return ops.rmsnorm(x.x if self.fold else x, self.final_ln, self.d.eps)


class EncoderStack(_PreNormBlocks):
    """T5Stack(is_decoder=False).  Weights: the reference's state_dict names under `prefix`."""

    def __init__(self, w, dims, device, prefix="encoder", max_len=64):
        self.d, self.dev = dims, device
        self.layers = []
        for l in range(dims.num_layers):
            p = f"{prefix}.block.{l}.layer"
            sa = f"{p}.0.SelfAttention"
            self.layers.append(dict(
                ln0=_dev(w, f"{p}.0.layer_norm.weight", device),
                wqkv=torch.cat([_dev(w, f"{sa}.{n}.weight", device) for n in "qkv"]).contiguous(),
                wo=_dev(w, f"{sa}.o.weight", device),
                ln1=_dev(w, f"{p}.1.layer_norm.weight", device),
                wi=_dev(w, f"{p}.1.DenseReluDense.wi.weight", device),
                wo2=_dev(w, f"{p}.1.DenseReluDense.wo.weight", device)))
        # round 5: the T5LayerNorms folded into the projections they feed (ops.ResidualRows; MEVI_FOLD_NORM=0: separate passes)
        self.fold = ops.fold_norm_ok(dims.d_model)
        if self.fold:
            for L in self.layers:
                L["wqkv"], L["wi"] = ops.fold_weight(L["wqkv"], L["ln0"]), ops.fold_weight(L["wi"], L["ln1"])
        ops.prepare_weights(self.layers, ("wqkv", "wo", "wi", "wo2"))
        for L in self.layers:   # |V| bound: the self-attention context goes to `o` as its split image (ops.ctx_bound)
            L["vb"] = ops.ctx_bound(math.sqrt(dims.d_model) * 1.0001 if self.fold else ops.norm_out_bound(L["ln0"], dims.d_model), L["wqkv"])
        self.final_ln = _dev(w, f"{prefix}.final_layer_norm.weight", device)
        self.out_norm = ops.norm_out_bound(self.final_ln, dims.d_model)     # l2 bound of a returned state row
        self.rel = _dev(w, f"{prefix}.block.0.layer.0.SelfAttention.relative_attention_bias.weight", device)
        self._bias = {}
        self.pack = True   # padding-free per-token operators (see forward)

    def bias(self, S):
        if S not in self._bias:
            self._bias[S] = bias_table(self.rel, S, S, True, self.d.buckets)
        return self._bias[S]

    def forward(self, embeddings, input_ids, attention_mask, pack=None):
        """input_ids/attention_mask i64[B, S] (S <= 256) -> last hidden state f32[B, S, d_model].

        pack (default: self.pack): run every per-token operator (norms, the five linear layers of a block) on the REAL
        tokens of the batch only and use the padded [B, S] layout just for attention.  The reference pads every
        query to 32 and every passage to 128 tokens and runs them all (modeling_t5.py:969-1069); with MSMARCO queries
        of ~10 tokens that is 3x the arithmetic.  Real positions get bit-identical values (a GEMM row, a norm, an
        attention row never look at padded rows: their keys are masked, and 0 * finite is exactly 0); PADDED positions
        of the returned states are 0 instead of the reference's don't-care values -- every consumer masks them."""
        d = self.d
        B, S = input_ids.shape
        pack = self.pack if pack is None else pack
        idx = None
        if pack:
            flat = attention_mask.reshape(-1)
            idx = torch.nonzero(flat != 0).view(-1)
            if idx.numel() == 0 or idx.numel() > 0.9 * B * S:   # nothing to gain (or nothing to do)
                idx = None
        bias = self.bias(S)
        if idx is None:
            x = self._start(ops.gather_rows(embeddings, input_ids.reshape(-1)))
            for L in self.layers:
                qkv = self._nl(x, L, "ln0", "wqkv").view(B, S, 3 * d.inner)
                ctx = ops.attention(qkv[:, :, :d.inner], qkv[:, :, d.inner:2 * d.inner], qkv[:, :, 2 * d.inner:],
                                    d.num_heads, bias=bias, key_mask=attention_mask, split_bound=L["vb"])
                x = self._rl(ctx if L["vb"] is not None else ctx.view(B * S, d.inner), L, "wo", x)
                x = self._rl(self._nl(x, L, "ln1", "wi", relu=True, for_gemm=True), L, "wo2", x)
            return self._final(x).view(B, S, d.d_model)
        x = self._start(ops.gather_rows(embeddings, input_ids.reshape(-1)[idx]))            # [T, d_model], T real tokens
        seq_off, longest = packed_offsets(attention_mask)
        if seq_off is not None and varlen_ok(longest, d.d_kv):
            # right-padded sequences (what the tokenizers produce): attention runs on the packed rows too
            for L in self.layers:
                qkv = self._nl(x, L, "ln0", "wqkv")
                ctx = ops.attention_varlen(qkv[:, :d.inner], qkv[:, d.inner:2 * d.inner], qkv[:, 2 * d.inner:], seq_off,
                                           longest, d.num_heads, bias=bias, split_bound=L["vb"])
                x = self._rl(ctx, L, "wo", x)
                x = self._rl(self._nl(x, L, "ln1", "wi", relu=True, for_gemm=True), L, "wo2", x)
        else:
            qkv = torch.zeros((B * S, 3 * d.inner), dtype=torch.float32, device=x.device)   # padded rows stay 0 (finite)
            for L in self.layers:
                ops.scatter_rows(self._nl(x, L, "ln0", "wqkv"), idx, qkv)
                q3 = qkv.view(B, S, 3 * d.inner)
                ctx = ops.attention(q3[:, :, :d.inner], q3[:, :, d.inner:2 * d.inner], q3[:, :, 2 * d.inner:],
                                    d.num_heads, bias=bias, key_mask=attention_mask)
                x = self._rl(ops.gather_rows(ctx.view(B * S, d.inner), idx), L, "wo", x)
                x = self._rl(self._nl(x, L, "ln1", "wi", relu=True, for_gemm=True), L, "wo2", x)
        out = torch.zeros((B * S, d.d_model), dtype=torch.float32, device=x.device)
        ops.scatter_rows(self._final(x), idx, out)
        return out.view(B, S, d.d_model)


# packed attention: sequences up to 64 tokens run one wave per (sequence, head); 65..128 tokens with 64-wide heads (the
# passages of t5-base / bert-base towers) use the matrix-core kernel on the real rows and keys; anything else keeps the
# padded layout
VARLEN_MAX_KEYS = 64
VARLEN_MFMA_KEYS, VARLEN_MFMA_DH = 128, 64


def varlen_ok(longest, dh):
    return longest <= VARLEN_MAX_KEYS or (longest <= VARLEN_MFMA_KEYS and dh == VARLEN_MFMA_DH)


def packed_offsets(attention_mask):
    """(seq_off i64[B+1] on the device, longest) when every row of the mask is 1...10...0 (right padding, empty rows
    allowed), else (None, 0): packed attention numbers positions 0..len-1, which needs the real tokens to lead."""
    m = attention_mask != 0
    lens = m.sum(1)
    S = m.shape[1]
    prefix = torch.arange(S, device=m.device)[None, :] < lens[:, None]
    seq_off = torch.zeros(m.shape[0] + 1, dtype=torch.int64, device=m.device)
    torch.cumsum(lens, 0, out=seq_off[1:])
    ok, longest = torch.stack([(prefix == m).all().to(torch.int64), lens.max()]).tolist()
    return (seq_off, int(longest)) if ok else (None, 0)


# A/B switch of the one-position decoder's pre-multiplied o(v(.)) projection (DecoderStack.__init__): "0" keeps the two
# GEMMs of the reference; either way the tower stays within the goldens' 5e-5 (tests/test_t5_gpu.py)
WOV_FUSE = os.environ.get("MEVI_TOWER_WOV", "1") != "0"

# batches up to this many rows may run as one replayed HIP graph (graph=True); larger ones always run eagerly, packed.
# A graph needs fixed shapes, i.e. the PADDED layout (32 tokens per query instead of the real ~11): it trades the host's ~10 us per
# launch for 3x the encoder rows.  tools/probe_graph_batches.py (profiles/r06_latency.txt), eager -> graph, ms per call:
# tower 8 queries 2.73 -> 1.74, 16: 2.69 -> 2.33, 32: 3.03 -> 2.77, 64: 3.55 -> 4.12; NCI generate 8: 6.11 -> 3.80, 16: 6.18 -> 4.76,
# 32: 6.50 -> 5.68, 64: 7.95 -> 8.58 -- same bits either way.  The crossover is between 32 and 64 queries.
GRAPH_MAX_ROWS = 32


class GraphCache:
    """Captured HIP graphs of fixed-shape device functions, keyed by shape: `run(key, fn, *inputs)` copies the inputs
    into the graph's static buffers, replays, and returns clones of the outputs.  Capture happens on the second call
    of a key (the first runs eagerly: it warms kernels, caches and lazily built tables, none of which may happen
    under capture).  The kernels are issued through the C ABI on torch's current stream, which is the capturing
    stream inside `torch.cuda.graph`."""

    def __init__(self):
        self.seen, self.graphs = set(), {}

    def run(self, key, fn, *inputs):
        if key not in self.seen:
            self.seen.add(key)
            return fn(*inputs)
        if key not in self.graphs:
            static_in = [t.clone() for t in inputs]
            torch.cuda.synchronize()
            g = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g):
                static_out = fn(*static_in)
            self.graphs[key] = (g, static_in, static_out)
        g, static_in, static_out = self.graphs[key]
        for s, t in zip(static_in, inputs):
            s.copy_(t)
        g.replay()
        if isinstance(static_out, (tuple, list)):
            return type(static_out)(o.clone() for o in static_out)
        return static_out.clone()


class CrossKV:
    """Cross-attention K|V of a batch: per layer either padded f32[B, S, 2*inner] + the key mask, or packed
    f32[T_real, 2*inner] + the sequences' row offsets (i64 [B+1]) and the longest length."""

    def __init__(self, layers, mask, kv_off=None, longest=0):
        self.layers, self.mask, self.kv_off, self.longest = layers, mask, kv_off, longest


class DecoderStack(_PreNormBlocks):
    """T5Stack(is_decoder=True) evaluated one position at a time with a KV cache.

    The reference runs use_cache=False and recomputes the whole prefix for every beam at every
    step (main_models.py:3615); under the causal mask the cached form is the same function
    (SURVEY 8(a') note iv).  Cross-attention K/V are projected once per QUERY, not per beam."""

    def __init__(self, w, dims, device, prefix="decoder", n_layers=None, max_len=8):
        self.d, self.dev = dims, device
        n_layers = dims.num_decoder_layers if n_layers is None else n_layers
        self.layers = []
        for l in range(n_layers):
            p = f"{prefix}.block.{l}.layer"
            sa, xa = f"{p}.0.SelfAttention", f"{p}.1.EncDecAttention"
            self.layers.append(dict(
                ln0=_dev(w, f"{p}.0.layer_norm.weight", device),
                # q | k | v as ONE projection: a decode step writes all three into the cache row of its position (q is read back
                # from there), one GEMM of 9 column tiles instead of 3 + 6 -- one tile round less per layer and step on 256 CUs
                wqkv=torch.cat([_dev(w, f"{sa}.{n}.weight", device) for n in "qkv"]).contiguous(),
                wo=_dev(w, f"{sa}.o.weight", device),
                ln1=_dev(w, f"{p}.1.layer_norm.weight", device),
                xq=_dev(w, f"{xa}.q.weight", device),
                xkv=torch.cat([_dev(w, f"{xa}.k.weight", device), _dev(w, f"{xa}.v.weight", device)]),
                xo=_dev(w, f"{xa}.o.weight", device),
                ln2=_dev(w, f"{p}.2.layer_norm.weight", device),
                wi=_dev(w, f"{p}.2.DenseReluDense.wi.weight", device),
                wo2=_dev(w, f"{p}.2.DenseReluDense.wo.weight", device)))
        keys = ("wqkv", "wo", "xq", "xo", "wi", "wo2")
        self.fold = ops.fold_norm_ok(dims.d_model)      # the T5LayerNorms folded into wqkv / (wov) / xq / wi (ops.ResidualRows)
        if max_len == 1 and ops.GEMM_MODE == "split" and WOV_FUSE:
            # A single-position decoder (the towers) attends to ONE key: the softmax weight is exactly 1, the context is v, and
            # o(v(h)) = h (Wo Wv)^T -- one projection instead of two (the product is taken once, in f64).  MEVI_GEMM=exact keeps
            # the two sequential-chain GEMMs of the reference.
            for L in self.layers:
                L["wov"] = (L["wo"].double() @ L["wqkv"][2 * dims.inner:].double()).float().contiguous()
                if self.fold:
                    L["wov"] = ops.fold_weight(L["wov"], L["ln0"])
            keys += ("wov",)
        if self.fold:
            for L in self.layers:
                L["wqkv"], L["xq"], L["wi"] = (ops.fold_weight(L["wqkv"], L["ln0"]), ops.fold_weight(L["xq"], L["ln1"]),
                                               ops.fold_weight(L["wi"], L["ln2"]))
        # the cross-attention K|V of ALL layers read the same encoder states: one GEMM of n_layers * 6 column tiles (12 layers:
        # 85 tile rounds on 256 CUs instead of 12 x 8)
        self.xkv_all = ops.weight(torch.cat([L["xkv"] for L in self.layers]).contiguous(), keep_norm=True)
        for l, L in enumerate(self.layers):      # a layer's own K|V weight = its rows of the one image (held once on the device)
            L["xkv"] = ops.weight_rows(self.xkv_all, l * 2 * dims.inner, (l + 1) * 2 * dims.inner)
        if isinstance(self.xkv_all, ops.SplitRows):
            self.xkv_all.norm = None
        ops.prepare_weights(self.layers, keys)
        for L in self.layers:
            L["vb"] = ops.ctx_bound(math.sqrt(dims.d_model) * 1.0001 if self.fold else ops.norm_out_bound(L["ln0"], dims.d_model), L["wqkv"])
            L["xvb"] = None          # cross-attention: needs the encoder's output norm (set_encoder_norm)
        self.final_ln = _dev(w, f"{prefix}.final_layer_norm.weight", device)
        rel = _dev(w, f"{prefix}.block.0.layer.0.SelfAttention.relative_attention_bias.weight", device)
        self.self_bias = bias_table(rel, max_len, max_len, False, dims.buckets)   # [H, T, T]
        self.max_len = max_len

    def set_encoder_norm(self, enc_norm):
        """l2 bound of the rows handed to cross_kv (EncoderStack.out_norm): lets the cross-attention contexts be written
        as the split image of `EncDecAttention.o` (ops.ctx_bound).  Without it they stay f32."""
        for L in self.layers:
            L["xvb"] = ops.ctx_bound(enc_norm, L["xkv"])

    def cross_kv(self, enc, enc_mask=None, pack=True):
        """Per-layer cross-attention K|V of the encoder states, as a CrossKV.  With `enc_mask` (and pack) only the real
        positions are projected; for right-padded masks the K|V rows stay PACKED (the attention kernels read a
        sequence's real keys through its row offsets), otherwise they are scattered into a zeroed [B, S, 2*inner]
        buffer.  pack=False projects every position and keeps the mask: fixed shapes, no host synchronisation (the
        graph-replay path of small batches)."""
        B, S, dm = enc.shape
        flat = enc.reshape(B * S, dm)
        idx = None
        if enc_mask is not None and pack:
            idx = torch.nonzero(enc_mask.reshape(-1) != 0).view(-1)
            if idx.numel() == 0 or idx.numel() > 0.9 * B * S:
                idx = None
        w2, nl = 2 * self.d.inner, len(self.layers)
        if idx is None:
            kv = ops.linear(flat, self.xkv_all).view(B, S, nl * w2)      # layer l = columns [l w2, (l + 1) w2)
            return CrossKV([kv[:, :, l * w2:(l + 1) * w2] for l in range(nl)], enc_mask)
        real = ops.gather_rows(flat, idx)
        seq_off, longest = packed_offsets(enc_mask)
        if seq_off is not None and 0 < longest <= 256:
            kv = ops.linear(real, self.xkv_all)
            return CrossKV([kv[:, l * w2:(l + 1) * w2] for l in range(nl)], None, seq_off, longest)
        real = ops.gemm_input(real)        # one operand image for all layers
        out = []
        for L in self.layers:
            kv = torch.zeros((B * S, 2 * self.d.inner), dtype=torch.float32, device=enc.device)
            out.append(ops.scatter_rows(ops.linear(real, L["xkv"]), idx, kv).view(B, S, 2 * self.d.inner))
        return CrossKV(out, enc_mask)

    def new_cache(self, rows):
        """Per layer f32 [rows, T, 3 * inner]: q | k | v of every position (q of position t is only read at step t)."""
        return [torch.empty((rows, self.max_len, 3 * self.d.inner), dtype=torch.float32, device=self.dev)
                for _ in self.layers]

    def step(self, x, t, cache, xkv, enc_mask, kv_div, key_rows=None):
        """x f32[n, d_model]: embeddings of the token at position t of every row; cache[l] f32[n, T, 3*inner]
        holds self-attention q|K|V of positions < t (position t is written here); rows r attend to the
        encoder states of query r // kv_div (`xkv`: the CrossKV of cross_kv, which carries its own mask / offsets;
        `enc_mask` is kept for callers of the older signature).  Returns the final-normed hidden state f32[n, d_model].

        key_rows (i32 [n, t + 1], beam search): the caches are NOT re-ordered between steps -- cache[l] has room for the
        largest row count, position t of row r is written to cache row r, and key_rows[r, j] names the cache row that
        holds position j of row r's prefix (its ancestor at step j; key_rows[r, t] = r)."""
        d = self.d
        n = x.shape[0]
        if key_rows is not None:
            cache = [kvc[:n] if t == 0 else kvc for kvc in cache]
        x = self._start(x)
        for L, kvc, xc in zip(self.layers, cache, xkv.layers):
            ctx = None
            if t == 0 and "wov" in L:
                # o(v(norm x)) in one projection (see __init__)
                x = ops.linear_residual(x, L["wov"], x, normed_eps=d.eps) if self.fold else \
                    ops.linear(ops.rmsnorm(x, L["ln0"], d.eps, for_gemm=True), L["wov"], residual=x)
            elif t == 0:
                # one key: its softmax weight is exp(0) / exp(0) = 1 exactly, so the attention output IS v -- no query
                # projection, no attention kernel; a single-position decoder (the towers) never needs k either
                if self.max_len == 1:
                    ctx = self._nl(x, dict(L, wv=L["wqkv"][2 * d.inner:]), "ln0", "wv")
                else:
                    self._nl(x, dict(L, wkv=L["wqkv"][d.inner:]), "ln0", "wkv", out=kvc[:, 0, d.inner:])
                    ctx = kvc[:, 0, 2 * d.inner:]
            else:
                self._nl(x, L, "ln0", "wqkv", out=kvc[:n, t, :])
                q = kvc[:n, t, :d.inner]
                if key_rows is not None:
                    ctx = ops.attention_cached(q, kvc[:, :, d.inner:2 * d.inner], kvc[:, :, 2 * d.inner:], key_rows, d.num_heads,
                                               bias=self.self_bias, q_pos0=t, causal=True, split_bound=L["vb"])
                else:
                    ctx = ops.attention(q.unsqueeze(1), kvc[:, :t + 1, d.inner:2 * d.inner], kvc[:, :t + 1, 2 * d.inner:],
                                        d.num_heads, bias=self.self_bias, q_pos0=t, causal=True, split_bound=L["vb"])
                    if L["vb"] is None:
                        ctx = ctx.view(n, d.inner)
            if ctx is not None:
                x = self._rl(ctx, L, "wo", x)
            q = self._nl(x, L, "ln1", "xq")
            if xkv.kv_off is None:
                ctx = ops.attention(q.view(n, 1, d.inner), xc[:, :, :d.inner], xc[:, :, d.inner:], d.num_heads,
                                    kv_div=kv_div, key_mask=xkv.mask, split_bound=L["xvb"])
            else:
                ctx = ops.attention(q.view(n, 1, d.inner), xc[:, :d.inner], xc[:, d.inner:], d.num_heads, kv_div=kv_div,
                                    kv_off=xkv.kv_off, kv_longest=xkv.longest, split_bound=L["xvb"])
            x = self._rl(ctx if L["xvb"] is not None else ctx.view(n, d.inner), L, "xo", x)
            x = self._rl(self._nl(x, L, "ln2", "wi", relu=True, for_gemm=True), L, "wo2", x)
        return self._final(x)


# Padded tokens per device pass of a tower (8192 queries x 32, 2048 passages x 128): results do not depend on the
# grouping (row-wise operators, tested), larger passes fill the GEMM grids (t5-base tower: 19.5 k queries/s at 512 rows
# per pass, 30.9 k at 2048, 35.4 k at 8192; passages 5.7 k/s at 512, 6.4 k/s at 2048).
# Round 4: 1 M tokens (8192 passages x 128): 21.0 k passages/s against 19.4 k at 2048 per pass and 17.8 k at 512 (the GEMMs' last,
# partly filled round of tiles weighs less; 7 GB of FFN image per pass on a 288 GB device).
DEVICE_PASS_TOKENS = 1048576


class TwinTower:
    """The (tied) tower of the dense arm: T5Model encoder + one decoder step on token 0,
    reps = last_hidden_state[:, 0, :], normalize=False (DocumentEncoder.encode, document_encoder.py:104-120).

    `encode_query(qry)` takes the reference's {'input_ids', 'attention_mask'} mapping."""

    def __init__(self, weights, dims=None, device=None, batch_size=None, **cfg):
        self.dev = torch.device(device if device is not None else "cuda")
        self.d = dims if dims is not None else T5Dims(**cfg)
        self.shared = _dev(weights, "shared.weight", self.dev)
        self.encoder = EncoderStack(weights, self.d, self.dev)
        self.decoder = DecoderStack(weights, self.d, self.dev, max_len=1)
        self.decoder.set_encoder_norm(self.encoder.out_norm)
        self.batch_size = batch_size   # rows per device pass; None: DEVICE_PASS_TOKENS // sequence length
        self.dim = self.d.d_model      # width of the embeddings this tower emits
        self._graphs = GraphCache()

    def _encode_fixed(self, ids, mask):
        """One pass at fixed shapes (padded layout, no host synchronisation): what a HIP graph can capture."""
        enc = self.encoder.forward(self.shared, ids, mask, pack=False)
        B = ids.shape[0]
        x = ops.gather_rows(self.shared, torch.zeros(B, dtype=torch.int64, device=self.dev))
        return self.decoder.step(x, 0, self.decoder.new_cache(B), self.decoder.cross_kv(enc, mask, pack=False), mask, 1)

    def encode_query(self, qry, graph=False):
        """graph=True (batches of at most GRAPH_MAX_ROWS rows): replay a captured HIP graph of the whole forward at
        fixed, padded shapes.  Same kernels, same bits as the eager pass; measured (tools/bench_latency.py) it does not
        lower the median latency -- the ~300 kernels of a single query are bound by their own run time (3.8 ms), the
        asynchronous eager launches already hide the host -- but it removes the host-side jitter (p90 = median)."""
        ids = qry["input_ids"].to(self.dev, torch.int64)
        mask = qry["attention_mask"].to(self.dev, torch.int64)
        if graph and 0 < ids.shape[0] <= GRAPH_MAX_ROWS:
            return self._graphs.run(("tower",) + tuple(ids.shape), self._encode_fixed, ids.contiguous(), mask.contiguous())
        outs = []
        step = self.batch_size or max(1, DEVICE_PASS_TOKENS // max(1, ids.shape[1]))
        for a in range(0, ids.shape[0], step):
            i, m = ids[a:a + step].contiguous(), mask[a:a + step].contiguous()
            enc = self.encoder.forward(self.shared, i, m)
            B = i.shape[0]
            x = ops.gather_rows(self.shared, torch.zeros(B, dtype=torch.int64, device=self.dev))
            outs.append(self.decoder.step(x, 0, self.decoder.new_cache(B), self.decoder.cross_kv(enc, m), m, 1))
        return torch.cat(outs) if outs else torch.empty((0, self.d.d_model), device=self.dev)

    encode = encode_query
    # tied towers (T5-ANCE, DocumentEncoder.build(tied=True)): lm_p is lm_q, a passage is a 128-token "query"
    # (document_encoder.py:122-123)
    encode_passage = encode_query
