"""NCI-RQ seq2seq arm: T5 encoder + 6-layer decoder + PAWA adaptive head + constrained beam search.

Mirrors `T5ForConditionalGeneration.generate(...)` as infer() calls it (MEVI/main_models.py:3612-3663;
MEVI/transformers/generation_utils.py:116-577, 709-1011; modeling_t5.py:1561-1689) for the configuration
every script uses: decode_embedding=2, adaptor_decode=adaptor_efficient=1, the shared-sons codebook tree,
num_beams = num_return_sequences.  The search is the validated restatement of SURVEY 8(a'):

  enc = Encoder(input_ids, mask)                      once per query
  for level p in 0..M-1:  logits (last position, valid columns) -> beam_step -> top-R prefixes
  final = score + log_softmax(level-M logits)[eos];   result = final / (M+1)**length_penalty

Differences from the reference that do not change the function computed: KV-cached decoder and
adaptor (reference: use_cache=False), only the last position and only the K+1 valid columns of the
adaptive head are evaluated (reference materialises [B*R, t, d, V]), cross K/V once per query, and the
adaptor side of the head -- which sees nothing but a beam's code prefix -- is evaluated once per PREFIX
(PrefixTables) instead of once per beam per step.
"""
import os

import numpy as np
import torch

from . import ops
from .t5 import GRAPH_MAX_ROWS, DecoderStack, EncoderStack, GraphCache, T5Dims, _dev


class NCIConfig(T5Dims):
    def __init__(self, M=4, K=32, adaptor_layer_num=4, num_decoder_layers=6, **kw):
        super().__init__(num_decoder_layers=num_decoder_layers, **kw)
        self.M, self.K, self.adaptor_layers = M, K, adaptor_layer_num
        self.V = K * (M + 2) + 2            # decode_vocab_size (main_models.py:1337-1339)
        self.T = M + 1                      # decoder positions actually evaluated (tokens 0..M)


def config_from_weights(w, M, K):
    """Read the architecture off the checkpoint's tensor shapes (the reference takes it from
    --model_info, MEVI/main.py:755-773; the shapes are authoritative for a given checkpoint)."""
    def count(prefix):
        n = 0
        while f"{prefix}.{n}.layer.0.layer_norm.weight" in w:
            n += 1
        return n

    heads = w["encoder.block.0.layer.0.SelfAttention.relative_attention_bias.weight"].shape[1]
    inner = w["encoder.block.0.layer.0.SelfAttention.q.weight"].shape[0]
    na = 0
    while f"adaptor.layers.{na}.norm1.weight" in w:
        na += 1
    return NCIConfig(M=M, K=K, adaptor_layer_num=na, d_model=w["shared.weight"].shape[1],
                     d_ff=w["encoder.block.0.layer.1.DenseReluDense.wi.weight"].shape[0], num_heads=heads,
                     d_kv=inner // heads, num_layers=count("encoder.block"), num_decoder_layers=count("decoder.block"),
                     relative_attention_num_buckets=w["encoder.block.0.layer.0.SelfAttention.relative_attention_bias.weight"].shape[0])


MODEL_INFO = {   # --model_info (MEVI/main.py:755-773): layers, decoder layers, d_ff, d_model, heads (d_kv = 64)
    "small": (6, 3, 2048, 512, 8), "base": (12, 6, 3072, 768, 12), "large": (24, 12, 4096, 1024, 16)}


def expected_shapes(c):
    """name -> shape of every tensor NCIModel reads (the reference's T5ForConditionalGeneration state_dict names);
    a leading None = any size (the token vocabulary)."""
    d, inner, ff, H = c.d_model, c.inner, c.d_ff, c.num_heads
    out = {"shared.weight": (None, d), "decode_embeddings.weight": (c.V, d), "lm_head.weight": (c.V, d),
           "adaptor_embeddings": (1, 1, d), "adaptor_linear.weight": (d * c.V, d)}
    for st, n, dec in (("encoder", c.num_layers, False), ("decoder", c.num_decoder_layers, True)):
        for l in range(n):
            p = f"{st}.block.{l}.layer"
            atts = [f"{p}.0.SelfAttention"] + ([f"{p}.1.EncDecAttention"] if dec else [])
            for a in atts:
                for w in "qkv":
                    out[f"{a}.{w}.weight"] = (inner, d)
                out[f"{a}.o.weight"] = (d, inner)
            ffl = 2 if dec else 1
            for j in range(ffl + 1):
                out[f"{p}.{j}.layer_norm.weight"] = (d,)
            out[f"{p}.{ffl}.DenseReluDense.wi.weight"] = (ff, d)
            out[f"{p}.{ffl}.DenseReluDense.wo.weight"] = (d, ff)
        out[f"{st}.block.0.layer.0.SelfAttention.relative_attention_bias.weight"] = (c.buckets, H)
        out[f"{st}.final_layer_norm.weight"] = (d,)
    for l in range(c.adaptor_layers):
        p = f"adaptor.layers.{l}"
        for a in ("self_attn", "multihead_attn"):
            out[f"{p}.{a}.in_proj_weight"], out[f"{p}.{a}.in_proj_bias"] = (3 * d, d), (3 * d,)
            out[f"{p}.{a}.out_proj.weight"], out[f"{p}.{a}.out_proj.bias"] = (d, d), (d,)
        ffa = None   # dim_feedforward of nn.TransformerDecoderLayer: read off the checkpoint (torch default 2048)
        out[f"{p}.linear1.weight"], out[f"{p}.linear1.bias"] = (ffa, d), (ffa,)
        out[f"{p}.linear2.weight"], out[f"{p}.linear2.bias"] = (d, ffa), (d,)
        for n_ in (1, 2, 3):
            out[f"{p}.norm{n_}.weight"] = out[f"{p}.norm{n_}.bias"] = (d,)
    return out


def check_weights(w, c, report=print):
    """The rule of try_load_ckpt's NCI branch (MEVI/main.py:243-246) for the tensors the inference path reads: a tensor
    that is absent or whose shape differs from the model's is reported as `Bad parameter <name>.` -- the reference then
    CONTINUES with that parameter at its random initialisation, which no build can reproduce, so here the list is
    returned for the caller to refuse."""
    bad = []
    for name, shape in expected_shapes(c).items():
        t = w.get(name)
        ok = t is not None and len(t.shape) == len(shape) and all(e is None or e == g for e, g in zip(shape, t.shape))
        if not ok:
            report(f"Bad parameter {name}.")
            bad.append(name)
    return bad


class Adaptor:
    """nn.TransformerDecoder(TransformerDecoderLayer(d, nhead=8), L) over the decode-token embeddings
    (modeling_t5.py:1252-1255, 1650-1665), one position at a time with cached K|V.  Its memory is the
    single learned vector adaptor_embeddings, so each layer's cross-attention output is a constant:
    softmax over one key is 1  =>  c_l = out_proj(v_proj(memory))."""

    NHEAD = 8

    def __init__(self, w, cfg, device):
        self.cfg, self.dev = cfg, device
        d = cfg.d_model
        mem = _dev(w, "adaptor_embeddings", device).reshape(1, d)
        self.layers = []
        for l in range(cfg.adaptor_layers):
            p = f"adaptor.layers.{l}"
            L = {k: _dev(w, f"{p}.{k}", device) for k in (
                "self_attn.in_proj_weight", "self_attn.in_proj_bias", "self_attn.out_proj.weight",
                "self_attn.out_proj.bias", "linear1.weight", "linear1.bias", "linear2.weight", "linear2.bias",
                "norm1.weight", "norm1.bias", "norm2.weight", "norm2.bias", "norm3.weight", "norm3.bias")}
            xw, xb = _dev(w, f"{p}.multihead_attn.in_proj_weight", device), _dev(w, f"{p}.multihead_attn.in_proj_bias", device)
            v = ops.linear(mem, xw[2 * d:].contiguous(), bias=xb[2 * d:].contiguous())
            L["cross_const"] = ops.linear(v, _dev(w, f"{p}.multihead_attn.out_proj.weight", device),
                                          bias=_dev(w, f"{p}.multihead_attn.out_proj.bias", device)).reshape(d).contiguous()
            L["wq"], L["bq"] = L["self_attn.in_proj_weight"][:d].contiguous(), L["self_attn.in_proj_bias"][:d].contiguous()
            L["wkv"], L["bkv"] = L["self_attn.in_proj_weight"][d:].contiguous(), L["self_attn.in_proj_bias"][d:].contiguous()
            del L["self_attn.in_proj_weight"]
            self.layers.append(L)
        ops.prepare_weights(self.layers, ("wq", "wkv", "self_attn.out_proj.weight", "linear1.weight", "linear2.weight"))

    def new_cache(self, rows):
        return [torch.empty((rows, self.cfg.T, 2 * self.cfg.d_model), dtype=torch.float32, device=self.dev)
                for _ in self.layers]

    def step(self, x, t, cache):
        """cache: new_cache()'s list of [rows, T, 2d] tensors, or an IndexedPrefixCache (the K|V of positions < t read from
        the prefix tables in place, position t written behind them)."""
        d = self.cfg.d_model
        n = x.shape[0]
        indexed = isinstance(cache, IndexedPrefixCache)
        for li, L in enumerate(self.layers):
            xg = ops.gemm_input(x)
            q = ops.linear(xg, L["wq"], bias=L["bq"])
            if indexed:
                buf = cache.bufs[li]
                ops.linear(xg, L["wkv"], bias=L["bkv"], out=buf[cache.tail:cache.tail + n])
                rows = buf.shape[0]
                ctx = ops.attention_cached(q, buf.as_strided((rows, t + 1, d), (2 * d, 0, 1)),
                                           buf.as_strided((rows, t + 1, d), (2 * d, 0, 1), buf.storage_offset() + d),
                                           cache.key_rows, self.NHEAD, q_pos0=t, causal=True, scale=(d // self.NHEAD) ** -0.5)
            else:
                kvc = cache[li]
                ops.linear(xg, L["wkv"], bias=L["bkv"], out=kvc[:, t, :])
                ctx = ops.attention(q.view(n, 1, d), kvc[:, :t + 1, :d], kvc[:, :t + 1, d:], self.NHEAD,
                                    q_pos0=t, causal=True, scale=(d // self.NHEAD) ** -0.5)
            sa = ops.linear(ctx.view(n, d), L["self_attn.out_proj.weight"], bias=L["self_attn.out_proj.bias"])
            x = ops.add_layernorm(x, sa, L["norm1.weight"], L["norm1.bias"])
            x = ops.add_layernorm(x, None, L["norm2.weight"], L["norm2.bias"], cvec=L["cross_const"])
            ff = ops.linear(ops.linear(ops.gemm_input(x), L["linear1.weight"], bias=L["linear1.bias"], relu=True, for_gemm=True),
                            L["linear2.weight"], bias=L["linear2.bias"])
            x = ops.add_layernorm(x, ff, L["norm3.weight"], L["norm3.bias"])
        return x


class IndexedPrefixCache:
    """The adaptor's K|V cache of n beams at position p WITHOUT assembling it: bufs[l] f32 [tail + n, 2d] holds the prefix
    tables of positions 0 .. p-1 one after the other (PrefixTables.kv_cat) and, from row `tail`, the K|V the beams write at
    position p; key_rows i32 [n, p + 1] names the row of every position (mevi_attention_cached_*: same kernel, same bits as
    on the gathered copy PrefixTables.cache_rows makes)."""

    __slots__ = ("bufs", "tail", "key_rows")

    def __init__(self, bufs, tail, key_rows):
        self.bufs, self.tail, self.key_rows = bufs, tail, key_rows


class PrefixTables:
    """The adaptor half of the PAWA head as tables over code prefixes.

    `adaptor(decode_embeddings[prefix])` and the head matrix `adaptor_linear(.)` it yields depend on the decoded prefix
    (0, c1 .. cp) only -- not on the query (modeling_t5.py:1650-1665: the adaptor's memory is one learned vector).
    At position p there are K**p prefixes, shared by every beam of every query: 1, 32, 1024, 32768 for the scripts'
    (M, K) = (4, 32) against 10 beams x thousands of queries per step.  So per level p < `levels`:
        tmat[p]  f32 [K**p, (K+1)*d]   head matrices, lm_head's rows added (row = sum_i c_i K**(p-i)), when within the byte budget,
        avec[p]  f32 [K**p, d]         adaptor outputs otherwise (the head GEMM then runs per beam as before),
        kv[l][p] f32 [K**p, 2d]        the adaptor layers' self-attention K|V of position p, so that the first position
                                       beyond the tables continues from a cache assembled by lookup.
    Every table row is produced by the same Adaptor.step / GEMM a beam would have run (all row-wise operators whose
    results do not depend on what else is in the batch -- tested), so the beams' logits keep their bits."""

    MAX_PREFIXES = 1 << 17
    MAX_FINAL_PREFIXES = 1 << 25     # the final position: adaptor vectors only, built in chunks (no K|V tables behind it)
    FINAL_CHUNK = 1 << 17
    # A position is tabled only when its K**p prefixes, with this margin, are fewer than the beam rows the caller is about to
    # run at it (`expected_queries` x live beams): building a row and running it for a beam cost the same adaptor step / head
    # GEMM, so below that the table is pure overhead -- the (3, 256) final position is 16.7 M prefixes (a 3 s build) of which
    # one MS MARCO dev run touches 69 800.  expected_queries None = a long-lived model (every table that fits the budget).
    WORTH_MARGIN = 1.25

    def __init__(self, model, table_bytes, expected_queries=None, beams=None):
        import time

        c, dev = model.cfg, model.dev
        d, K = c.d_model, c.K
        self.levels = 0
        self._cat, self._retired = None, []
        self.tmat, self.avec, self.kv = [], [], [[] for _ in model.adaptor.layers]
        self.expected_queries, self.beams = expected_queries, beams
        cache, spent = None, 0
        torch.cuda.synchronize(dev)
        t0 = time.perf_counter()
        self.level_ms = []                      # build time per position (a synchronisation per level: the build is rare)

        def lap():
            torch.cuda.synchronize(dev)
            self.level_ms.append(round((time.perf_counter() - t0) * 1e3 - sum(self.level_ms), 2))

        def worth(n, p):
            if expected_queries is None:
                return True
            live = n if beams is None else min(beams, n)                  # beams alive at position p (1, then min(R, K**p))
            return n * self.WORTH_MARGIN <= expected_queries * live or n == 1

        for p in range(c.T):
            n = K ** p
            kv_bytes = n * 2 * d * 4 * len(model.adaptor.layers)
            if not worth(n, p):
                break
            if n > self.MAX_PREFIXES or spent + kv_bytes + n * d * 4 > table_bytes:
                # The final position (eos) of a larger code space -- 32**4 = 1 M prefixes for the scripts' (4, 32): nothing
                # continues from it, so its adaptor K|V need no table, and its adaptor outputs (3 KB per prefix, 3.2 GB) fit
                # where its head matrices (101 KB per prefix) do not.  Built in chunks through the indexed cache (the K|V of the
                # earlier positions read from their tables in place): the adaptor's four layers then run ONCE PER PREFIX instead
                # of once per beam at the position where every query has its full set of beams (12 ms of a 182 ms pass).
                if (p == c.M and p == self.levels and 0 < p <= 7 and INDEXED_ADAPTOR_CACHE and n <= self.MAX_FINAL_PREFIXES
                        and spent + n * d * 4 <= table_bytes):
                    a = torch.empty((n, d), dtype=torch.float32, device=dev)
                    for lo in range(0, n, self.FINAL_CHUNK):
                        rows = torch.arange(lo, min(n, lo + self.FINAL_CHUNK), device=dev)
                        tokens = 2 + (p - 1) * K + rows % K
                        a[lo:lo + rows.numel()] = model.adaptor.step(ops.gather_rows(model.dec_emb, tokens), p,
                                                                     self.indexed_cache(model.adaptor, rows, p))
                    self.tmat.append(None)
                    self.avec.append(a)
                    spent += n * d * 4
                    self.levels = p + 1
                    for kl in self.kv:                        # the build's buffers (tables + a chunk's tail) are not needed again
                        kl[:] = [k.clone() for k in kl]
                    self._cat, self._retired = None, []
                    lap()
                break
            if p == 0:
                tokens = torch.zeros(1, dtype=torch.int64, device=dev)                    # decoder_start_token_id
                cache = model.adaptor.new_cache(1)
            else:
                rows = torch.arange(n, device=dev)
                tokens = 2 + (p - 1) * K + rows % K
                parent = rows // K
                cache = [ops.gather_rows(k.view(k.shape[0], -1), parent).view(n, c.T, -1) for k in cache]
            a = model.adaptor.step(ops.gather_rows(model.dec_emb, tokens), p, cache)
            for l, k in enumerate(cache):
                self.kv[l].append(k[:, p, :].contiguous())
            spent += kv_bytes
            t_bytes = n * (K + 1) * d * 4
            if spent + t_bytes <= table_bytes:
                self.tmat.append(ops.linear(a, model.head_w[p], bias=model.head_e[p] if ROW_LOGITS else None))
                self.avec.append(None)
                spent += t_bytes
            else:
                self.tmat.append(None)
                self.avec.append(a)
                spent += n * d * 4
            self.levels = p + 1
            lap()
        self.bytes = spent
        torch.cuda.synchronize(dev)
        self.build_ms = (time.perf_counter() - t0) * 1e3

    def describe(self):
        """What was tabled, for the bench line / logs."""
        return {"levels": self.levels, "head_matrices_at": [p for p in range(self.levels) if self.tmat[p] is not None],
                "adaptor_vectors_only_at": [p for p in range(self.levels) if self.tmat[p] is None], "bytes": self.bytes,
                "build_ms": round(self.build_ms, 2), "level_ms": self.level_ms, "expected_queries": self.expected_queries, "beams": self.beams}

    def indexed_cache(self, adaptor, pidx, p):
        """IndexedPrefixCache of beams whose prefix index at position p = self.levels is `pidx` -- for a position whose cache
        no later position needs (the final one): 4 layers x p gathers of [n, 2d] rows and the [n, T, 2d] caches not made."""
        assert p == self.levels and p <= 7
        K, n = adaptor.cfg.K, pidx.numel()
        d2 = 2 * adaptor.cfg.d_model
        offs = [0]
        for q in range(p):
            offs.append(offs[-1] + K ** q)
        tail = offs[p]
        if self._cat is None or self._cat[0].shape[0] < tail + n:
            cap = tail + max(n, 0 if self._cat is None else 2 * (self._cat[0].shape[0] - tail))
            cat = []
            for l in range(len(self.kv)):
                buf = torch.empty((cap, d2), dtype=torch.float32, device=pidx.device)
                for q in range(p):
                    buf[offs[q]:offs[q + 1]].copy_(self.kv[l][q])
                    self.kv[l][q] = buf[offs[q]:offs[q + 1]]          # the tables live in the buffer from now on
                cat.append(buf)
            if self._cat is not None:
                self._retired.append(self._cat)     # captured graphs (GraphCache) may still read the old buffers
            self._cat = cat
        cols = [(offs[q] + pidx // (K ** (p - q))) for q in range(p)]
        cols.append(tail + torch.arange(n, device=pidx.device))
        key_rows = torch.stack(cols, 1).to(torch.int32).contiguous()
        return IndexedPrefixCache([b[:tail + n] for b in self._cat], tail, key_rows)

    def cache_rows(self, adaptor, pidx, p):
        """The adaptor's K|V cache (positions < p filled) of beams whose prefix index at position p is `pidx`."""
        K = adaptor.cfg.K
        cache = adaptor.new_cache(pidx.numel())
        for q in range(p):
            idx = pidx // (K ** (p - q))
            for l, k in enumerate(cache):
                ops.gather_rows(self.kv[l][q], idx, out=k[:, q, :])      # straight into the cache slot (row stride T * 2d)
        return cache


# MEVI_HEAD_LOGITS=columns: the head as before round 4 (mevi_scale_f32, lm_head's rows added per column inside
# mevi_adaptive_logits_f32); three times the cache traffic -- kept for A/B timing.  Same bits at widths other than 768; at 768 the
# rows kernel follows the fused head's summation order, so the two differ in the last ulps there
ROW_LOGITS = os.environ.get("MEVI_HEAD_LOGITS", "rows") != "columns"
# MEVI_ADAPTOR_CACHE=copy: assemble the adaptor's cache of the final position by gathers (the form before round 4; same bits)
INDEXED_ADAPTOR_CACHE = os.environ.get("MEVI_ADAPTOR_CACHE", "indexed") != "copy"


def default_table_bytes(dev):
    """Byte budget of the prefix tables when the caller names none: half of the device memory that is free now, at most
    128 GiB.  On a 288 GB MI355X that tables, for a 3 x 256 codebook, the head matrices of the 65 536 two-code prefixes (52 GB:
    the 257-column head GEMM of position 2 becomes a lookup, 315 -> 248 ms per 6980 queries) and the adaptor outputs of the
    16.7 M three-code prefixes of the final position (51 GB); the scripts' 4 x 32 codebook needs 7.4 GB whatever the budget.
    The beams' bits do not depend on it (PrefixTables)."""
    free, _ = torch.cuda.mem_get_info(dev)
    return int(min(128 << 30, free // 2))


class PrefixTree:
    """The generic decode tree of the reference -- TreeBuilder(share_sons=False).add(tokens of one code path) for every
    existing path (MEVI/main_models.py:50-63, built from the doc -> code mapping at :1707-1728) -- level by level as
    arrays: level p holds the distinct prefixes of length p in lexicographic order (level 0 = the root), `mask[p]` i32
    [n_p, ceil(K/32)] = the codes of a node's children, `base[p]` i32 [n_p] = the index in level p + 1 of its first child
    (children are contiguous there, in code order).  All paths have M codes (the RQ setting); eos follows every leaf."""

    def __init__(self, paths, M, K, device):
        paths = np.unique(np.asarray(paths, dtype=np.int64).reshape(-1, M), axis=0)      # lexicographic
        assert paths.shape[0] > 0 and paths.min() >= 0 and paths.max() < K
        self.M, self.K, self.n_paths = M, K, paths.shape[0]
        W = (K + 31) // 32
        self.mask, self.base = [], []
        n = paths.shape[0]
        for p in range(M):
            first_child = np.ones(n, bool)                                 # first path of every distinct (p + 1)-prefix
            first_child[1:] = (paths[1:, :p + 1] != paths[:-1, :p + 1]).any(1)
            first_par = np.zeros(n, bool)                                  # first path of every distinct p-prefix
            first_par[0] = True
            if p > 0:
                first_par[1:] = (paths[1:, :p] != paths[:-1, :p]).any(1)
            par_of_row = np.cumsum(first_par) - 1
            child_of_row = np.cumsum(first_child) - 1
            rows = np.flatnonzero(first_child)
            codes, par = paths[rows, p], par_of_row[rows]
            mask = np.zeros((int(par_of_row[-1]) + 1, W), np.uint32)
            np.bitwise_or.at(mask, (par, codes >> 5), np.left_shift(np.uint32(1), (codes & 31).astype(np.uint32)))
            self.mask.append(torch.from_numpy(mask.view(np.int32)).to(device))
            self.base.append(torch.from_numpy(child_of_row[first_par].astype(np.int32)).to(device))


class NCIModel:
    """`generate()` mirrors the reference call; weights use the reference's state_dict names
    (T5ForConditionalGeneration: shared, encoder.*, decoder.*, decode_embeddings, adaptor*, lm_head)."""

    def __init__(self, weights, cfg=None, device=None, prefix_table_bytes=None, prefix_table_queries=None, **kw):
        self.dev = torch.device(device if device is not None else "cuda")
        self.cfg = cfg if cfg is not None else NCIConfig(**kw)
        # 0: evaluate the adaptor per beam per step; None: sized from the device when the tables are first built
        self.prefix_table_bytes = prefix_table_bytes
        # how many queries this model is about to serve (EvalRun passes its run's count): positions whose prefix count exceeds
        # the beam rows of that workload are not tabled (PrefixTables.WORTH_MARGIN); None = long-lived, table what fits
        self.prefix_table_queries = prefix_table_queries
        self._tables = None
        self._graphs = GraphCache()
        c = self.cfg
        self.shared = _dev(weights, "shared.weight", self.dev)
        self.dec_emb = _dev(weights, "decode_embeddings.weight", self.dev)
        self.encoder = EncoderStack(weights, c, self.dev)
        self.decoder = DecoderStack(weights, c, self.dev, max_len=c.T)
        self.decoder.set_encoder_norm(self.encoder.out_norm)
        self.adaptor = Adaptor(weights, c, self.dev)
        # adaptive head restricted to the valid columns of every position:
        # column 0 = eos (token 1), columns 1..K = tokens 2 + p*K + (c-1)     (modeling_t5.py:1578-1603)
        aw = _dev(weights, "adaptor_linear.weight", self.dev).view(c.d_model, c.V, c.d_model)   # [d, V, e]
        lm = _dev(weights, "lm_head.weight", self.dev)
        self.head_w, self.head_e = [], []
        for p in range(c.M + 1):
            cols = torch.tensor([1] + list(range(2 + p * c.K, 2 + (p + 1) * c.K)), device=self.dev)
            # rows ordered (column, d): W_p[c*d_model + d, e] = adaptor_linear.weight[d*V + v_c, e]
            self.head_w.append(ops.weight(aw[:, cols, :].permute(1, 0, 2).reshape((c.K + 1) * c.d_model, c.d_model).contiguous()))
            self.head_e.append(lm[cols].reshape(-1).contiguous())     # [(K+1)*d]: the bias of the head GEMM (lm_head_weight + adaptor_weight, modeling_t5.py:1683)
        del aw

    def tables(self, beams=None):
        # the policy is sized for the widest search seen: a later generate() with more beams rebuilds (and drops the graphs
        # captured over the old tables); beams=None (describe / tools) reads whatever is there
        if self._tables is not None and beams is not None and self._tables.beams is not None and beams > self._tables.beams:
            self._tables = None
            self._graphs = GraphCache()
        if self._tables is None:
            if self.prefix_table_bytes is None:
                self.prefix_table_bytes = default_table_bytes(self.dev)
            self._tables = PrefixTables(self, self.prefix_table_bytes, self.prefix_table_queries, beams)
        return self._tables

    def expect_queries(self, n):
        """The caller is about to run `n` queries through generate() (None: no limit known).  Tables already built for a
        smaller or equal workload stay; a larger one rebuilds them on the next call (they may now pay for more positions)."""
        old = self.prefix_table_queries
        self.prefix_table_queries = n
        if self._tables is not None and old is not None and (n is None or n > old):
            self._tables = None
            self._graphs = GraphCache()     # captured generate() graphs hold the old tables' addresses and level structure

    # -- one decoding position for all live beams -----------------------------------------------
    def _logits(self, tokens, t, dcache, acache, xkv, mask, kv_div, pidx=None, key_rows=None):
        """pidx: the beams' prefix indices at position t when the prefix tables cover it (acache unused then).
        key_rows: ancestor-indexed decoder caches (DecoderStack.step)."""
        c = self.cfg
        tok = ops.gather_rows(self.dec_emb, tokens)
        seq = self.decoder.step(tok, t, dcache, xkv, mask, kv_div, key_rows=key_rows)
        alpha = c.d_model ** -0.5                                  # modeling_t5.py:1607, applied inside the logits kernel
        if not ROW_LOGITS:
            seq = ops.scale(seq, alpha)
        tmat = None
        if pidx is not None:
            tab = self.tables()
            if tab.tmat[t] is not None:
                tmat = tab.tmat[t]
            else:
                a = ops.gather_rows(tab.avec[t], pidx)
        else:
            a = self.adaptor.step(tok, t, acache)
        if tmat is None:
            pidx = None
            if ROW_LOGITS:     # no table row: the head GEMM and the product with the hidden states in one pass where the shape allows
                return ops.head_logits(a, self.head_w[t], self.head_e[t], seq, alpha, c.K + 1)
            tmat = ops.linear(a, self.head_w[t])                   # [n, (K+1)*d]: adaptor_weight
        if ROW_LOGITS:
            return ops.adaptive_logits_rows(seq, alpha, tmat, c.K + 1, t_index=pidx)            # [n, K+1]
        return ops.adaptive_logits(seq, tmat, self.head_e[t].view(c.K + 1, c.d_model), t_index=pidx)

    @torch.no_grad()
    def generate(self, input_ids, attention_mask, num_beams=10, num_return_sequences=None, length_penalty=0.8,
                 max_length=None, graph=False, **reference_kwargs):
        """Returns (decoded i64[B*R, M+2], scores list[float] (descending per query),
        enc_last_hidden_state f32[B*R, S, d] view, None) like the reference's 4-tuple; the last slot
        (dec_hidden) is only read when query_encoder='nci' and is not produced (SURVEY 8(a') note iii).
        graph=True (batches of at most GRAPH_MAX_ROWS queries): replay the whole search as one captured HIP graph
        (fixed shapes: padded encoder, no host synchronisation inside) -- same kernels, same results, no host jitter;
        the median latency is the kernels' own time either way (tools/bench_latency.py)."""
        c = self.cfg
        if reference_kwargs.get("eval_all_documents"):      # generation_utils.py:507-521 -> _generate_all
            assert num_beams == 1 and num_return_sequences in (None, 1)
            scores, enc = self.generate_all(input_ids, attention_mask, length_penalty)
            return None, scores, enc, None
        R = num_beams
        assert num_return_sequences in (None, R), "needs num_beams == num_return_sequences"
        # decode_tree: None = the shared-sons tree of every script (TreeBuilder(share_sons=True): all K codes at every level),
        # or a PrefixTree = the reference's generic trie of the existing code paths (TreeBuilder(share_sons=False))
        tree = reference_kwargs.get("decode_tree")
        assert tree is None or (isinstance(tree, PrefixTree) and tree.M == c.M and tree.K == c.K), "decode_tree: a PrefixTree of this model's (M, K)"
        # K < R (SURVEY 8(a') note ii): the reference carries -1e9 placeholder beams until K**p real prefixes exist; they
        # never win against a real candidate, so the search keeps min(R, live * K) beams per level (golden G1 (3,8,10), (2,4,10))
        assert tree is not None or R <= c.K ** c.M, "fewer code paths than beams: the reference would return -1e9 placeholder hypotheses"
        assert max_length in (None, c.M + 2)
        ids = input_ids.to(self.dev, torch.int64).contiguous()
        mask = attention_mask.to(self.dev, torch.int64).contiguous()
        if graph and 0 < ids.shape[0] <= GRAPH_MAX_ROWS:
            if self.prefix_table_bytes:
                self.tables(R)
            decoded, hyp, enc = self._graphs.run(("generate", R, float(length_penalty), id(tree)) + tuple(ids.shape),
                                                 lambda i, m: self._search(i, m, R, length_penalty, pack=False, tree=tree), ids, mask)
        else:
            decoded, hyp, enc = self._search(ids, mask, R, length_penalty, pack=True, tree=tree)
        return decoded, hyp.reshape(-1).tolist(), enc, None

    @torch.no_grad()
    def generate_all(self, input_ids, attention_mask, length_penalty=0.8, max_rows=1 << 16):
        """`_generate_all` (MEVI/transformers/generation_utils.py:1013-1136; the `use_topic_model` ablation): the score
        of EVERY code path of every query, f32 [B, K**M] with path index sum_p c_p K**(M-1-p) -- per level the
        log-softmax over the position's valid columns added to the running score, the eos term at the end, divided by
        (M + 1) ** length_penalty in f32.  The reference re-runs the whole decoder on every prefix (use_cache=False, 128
        rows at a time); here the tree is walked depth-first in blocks of at most `max_rows` prefixes with the K|V caches
        of a block's ancestors kept, the adaptor side from the prefix tables where they reach.
        Returns (scores, encoder states)."""
        c = self.cfg
        M, K = c.M, c.K
        ids = input_ids.to(self.dev, torch.int64).contiguous()
        mask = attention_mask.to(self.dev, torch.int64).contiguous()
        B = ids.shape[0]
        enc = self.encoder.forward(self.shared, ids, mask)
        out = torch.empty((B, K ** M), dtype=torch.float32, device=self.dev)
        levels = self.tables().levels if self.prefix_table_bytes != 0 else 0
        for a in range(0, B, max_rows):                      # position 0: one row per query
            m_ = mask[a:a + max_rows]
            xkv = self.decoder.cross_kv(enc[a:a + max_rows], m_, pack=True)
            self._all_paths(out[a:a + m_.shape[0]], xkv, levels, max_rows, length_penalty)
        return out, enc

    def _all_paths(self, out, xkv, levels, max_rows, length_penalty):
        """Expansion of the code tree for the queries of one CrossKV.  A block holds rows in query-major order, the same
        number (kv_div) per query, so row r reads the encoder states of the block's query r // kv_div.  Levels are
        expanded whole while they fit `max_rows`; beyond that a block is split by query, then inside a query by prefix."""
        c = self.cfg
        M, K = c.M, c.K
        scale = (M + 1) ** length_penalty

        def expand(p, xkv, kv_div, q0, pidx, tokens, score, dcache, acache):
            n = tokens.numel()
            if p >= levels and acache is None:               # the adaptor runs per row from here on
                acache = self.adaptor.new_cache(n) if p == 0 else self.tables().cache_rows(self.adaptor, pidx, p)
            logits = self._logits(tokens, p, dcache, acache, xkv, None, kv_div, pidx if p < levels else None)
            lsm = ops.row_softmax(logits, log=True)          # columns: eos, then the K codes of position p
            qrow = q0 + torch.arange(n, device=self.dev) // kv_div
            if p == M:
                out[qrow, pidx] = (lsm[:, 0] + score) / scale
                return
            child = lsm[:, 1:] + score[:, None]              # [n, K]: running scores of the children

            def descend(xkv_, kv_div_, q0_, lo, hi):         # children of parent rows [lo, hi)
                rows = torch.arange(lo, hi, device=self.dev).repeat_interleave(K)
                code = torch.arange(K, device=self.dev).repeat(hi - lo)
                expand(p + 1, xkv_, kv_div_, q0_, pidx[rows] * K + code, 2 + p * K + code, child[lo:hi].reshape(-1),
                       _reorder_cache(dcache, rows, p + 1), _reorder_cache(acache, rows, p + 1) if p >= levels else None)

            nq = n // kv_div
            if n * K <= max_rows:
                descend(xkv, kv_div * K, q0, 0, n)
            elif nq > 1:                                      # split by query
                g = max(1, max_rows // (kv_div * K))
                for a in range(0, nq, g):
                    b = min(nq, a + g)
                    descend(_slice_cross_kv(xkv, a, b), kv_div * K, q0 + a, a * kv_div, b * kv_div)
            else:                                             # one query, too many prefixes: split by prefix
                per = max(1, max_rows // K)
                for lo in range(0, n, per):
                    hi = min(n, lo + per)
                    descend(xkv, (hi - lo) * K, q0, lo, hi)

        nq = out.shape[0]
        zeros = torch.zeros(nq, dtype=torch.int64, device=self.dev)
        expand(0, xkv, 1, 0, zeros, zeros, torch.zeros(nq, dtype=torch.float32, device=self.dev), self.decoder.new_cache(nq), None)

    def _search(self, ids, mask, R, length_penalty, pack, tree=None):
        """The device part of generate(): (decoded i64[B*R, M+2], hypothesis scores f64[B, R], encoder states).
        `tree` (a PrefixTree): beams continue along the trie's children only (mevi_beam_step_tree_f32).  The reference then
        runs all R beams from the first step, beams 1..R-1 seeded with -1e9 (generation_utils.py:752-756): they follow the
        same trie and fill the slots real candidates cannot whenever fewer than R exist -- so does this search (every beam
        sits on a trie node, each node has a child, hence R candidates at every level)."""
        c = self.cfg
        B = ids.shape[0]
        enc = self.encoder.forward(self.shared, ids, mask, pack=None if pack else False)
        xkv = self.decoder.cross_kv(enc, mask, pack=pack)

        nb = 1 if tree is None else R
        tokens = torch.zeros(B * nb, dtype=torch.int64, device=self.dev)     # decoder_start_token_id = 0
        scores = torch.zeros((B, nb), dtype=torch.float32, device=self.dev)
        if tree is not None:
            scores[:, 1:] = -1e9
            node = torch.zeros((B, nb), dtype=torch.int32, device=self.dev)  # every beam starts at the root
        codes = torch.zeros((B, nb, 0), dtype=torch.int64, device=self.dev)
        levels = self.tables(R).levels if self.prefix_table_bytes != 0 else 0     # positions the prefix tables cover
        pidx = torch.zeros(B * nb, dtype=torch.int64, device=self.dev)       # prefix index of every live beam
        # The decoder's K|V caches are never re-ordered (the reference index_selects every layer's cache by the surviving
        # beams' parents after each step, generation_utils.py:927-934): position p of step-p row r stays in cache row r and
        # every live beam carries the cache rows of its ancestors, `anc` i32 [rows, p].  Up to 8 positions (the few-keys
        # attention kernel); longer codes keep the copying form.
        indexed = c.M + 1 <= 8
        dcache = self.decoder.new_cache(B * (min(R, c.K ** c.M) if tree is None else R) if indexed else B * nb)
        anc = torch.zeros((B * nb, 0), dtype=torch.int32, device=self.dev)
        acache = self.adaptor.new_cache(B * nb) if levels == 0 else None
        base = torch.arange(B, device=self.dev)[:, None]
        for p in range(c.M + 1):
            if p == levels and p > 0:       # first position beyond the tables: its cache comes from them
                if p == c.M and p <= 7 and INDEXED_ADAPTOR_CACHE:   # ... in place, when no later position continues from it (<= 8 keys)
                    acache = self.tables().indexed_cache(self.adaptor, pidx, p)
                else:
                    acache = self.tables().cache_rows(self.adaptor, pidx, p)
            key_rows = None
            if indexed:
                key_rows = torch.cat([anc, torch.arange(anc.shape[0], dtype=torch.int32, device=self.dev)[:, None]], 1).contiguous()
            logits = self._logits(tokens, p, dcache, acache, xkv, mask, nb, pidx if p < levels else None, key_rows=key_rows)
            if p == c.M:
                break
            Rp = min(R, nb * c.K)                                             # beams alive after this level
            if tree is None:
                scores, parent, code = ops.beam_step(logits, scores, c.K, Rp)
            else:
                scores, parent, code, node = ops.beam_step_tree(logits, scores, c.K, Rp, node, tree.mask[p], tree.base[p])
            parent, code = parent.long(), code.long()
            rows = (base * nb + parent).reshape(-1)                           # surviving parents, [B*Rp]
            if indexed:
                anc = key_rows[rows]
            else:
                dcache = _reorder_cache(dcache, rows, p + 1)
            if p >= levels:
                acache = _reorder_cache(acache, rows, p + 1)
            pidx = pidx[rows] * c.K + code.reshape(-1)
            codes = torch.cat([torch.gather(codes, 1, parent[:, :, None].expand(-1, -1, codes.shape[2])),
                               code[:, :, None]], dim=2)
            tokens = (2 + p * c.K + code).reshape(-1)
            nb = Rp
        final = ops.beam_step(logits, scores, c.K, R, final_step=True)        # [B, R]
        hyp = final.double() / (c.M + 1) ** length_penalty                    # BeamHypotheses.add: len = M+1
        # descending by score, equal scores in beam order (a stable sort of the reference's hypotheses): hyp is a monotone
        # image of `final`, so the order comes from the tested segment-sort kernel on the f32 values (ties -> lower beam index)
        beam_ids = torch.arange(R, dtype=torch.int64, device=self.dev).repeat(B)
        seg = torch.arange(B + 1, dtype=torch.int64, device=self.dev) * R
        order = ops.segment_sort_desc(final.reshape(-1), beam_ids, seg, R)[1].view(B, R)
        hyp = torch.gather(hyp, 1, order)
        codes = torch.gather(codes, 1, order[:, :, None].expand(-1, -1, c.M))
        toks = 2 + torch.arange(c.M, device=self.dev) * c.K + codes
        decoded = torch.cat([torch.zeros((B, R, 1), dtype=torch.int64, device=self.dev), toks,
                             torch.ones((B, R, 1), dtype=torch.int64, device=self.dev)], dim=2).view(B * R, c.M + 2)
        return decoded, hyp, enc


def _slice_cross_kv(xkv, a, b):
    """The CrossKV of queries [a, b) of `xkv` (views)."""
    from .t5 import CrossKV

    if xkv.kv_off is None:
        return CrossKV([l[a:b] for l in xkv.layers], None if xkv.mask is None else xkv.mask[a:b])
    lo, hi = int(xkv.kv_off[a].item()), int(xkv.kv_off[b].item())
    return CrossKV([l[lo:hi] for l in xkv.layers], None, (xkv.kv_off[a:b + 1] - lo).contiguous(), xkv.longest)


def _reorder_cache(cache, rows, filled):
    """Beam re-ordering of the K|V caches [n, T, w] (generation_utils.py:927-934 `_reorder_cache`): row r of the result is
    row rows[r]; only the `filled` positions written so far are copied."""
    out = []
    for k in cache:
        n, T, w = k.shape
        new = torch.empty((rows.numel(), T, w), dtype=torch.float32, device=k.device)
        ops.gather_rows(k.view(n, T * w)[:, :filled * w], rows, out=new.view(-1, T * w)[:, :filled * w])
        out.append(new)
    return out


def decode_token(decoded, K):
    """main_models.decode_token for codebook models (MEVI/main_models.py:117-136): strip bos/eos, undo
    the per-position offset, clamp negatives to 0 -> codes i64[n, M]."""
    seqs = decoded[:, 1:-1] - 2
    seqs = seqs - torch.arange(seqs.shape[1], device=seqs.device) * K
    return seqs.clamp(min=0)


def dec_2d(dec, size):
    """main_utils.dec_2d (MEVI/main_utils.py:38-47)."""
    if torch.is_tensor(dec):
        return dec.reshape(-1, size, dec.shape[-1])
    return [dec[i:i + size] for i in range(0, len(dec), size)]


def load_npz_weights(npz, prefix="w."):
    return {k[len(prefix):]: torch.from_numpy(np.asarray(npz[k])) for k in npz.files if k.startswith(prefix)}
