"""GPU unit parity of the T5 / beam / fine-stage operators (through the C ABI) against plain
torch fp32 on the CPU.  Tolerances are stated per test; GEMM, pair_dot and the sorts are exact
(sequential fmaf chains / integer keys) and compared bit-for-bit with the C oracle."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

from mevi_amd import ops
from mevi_amd.hip import lib as hip_lib
from oracle import dense as odense

pytestmark = pytest.mark.gpu


def _chain_gemm(a, w):
    """C[m, n] = sequential fmaf chain over k (the oracle's dot)."""
    L = odense.lib()
    import ctypes
    L.oracle_dot_f32.restype = ctypes.c_float
    L.oracle_dot_f32.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int64]
    out = np.empty((a.shape[0], w.shape[0]), np.float32)
    for i in range(a.shape[0]):
        for j in range(w.shape[0]):
            out[i, j] = L.oracle_dot_f32(a[i].ctypes.data, w[j].ctypes.data, a.shape[1])
    return out


@pytest.mark.parametrize("M,N,K", [(70, 50, 64), (300, 130, 100), (5, 768, 768), (257, 129, 36)])
def test_linear_bit_exact_chain(cuda, M, N, K):
    rng = np.random.default_rng(M + N + K)
    a = rng.standard_normal((M, K)).astype(np.float32)
    w = rng.standard_normal((N, K)).astype(np.float32)
    out = ops.linear(torch.from_numpy(a).to(cuda), torch.from_numpy(w).to(cuda)).cpu().numpy()
    if M * N <= 40000:
        assert np.array_equal(out.view(np.uint32), _chain_gemm(a, w).view(np.uint32))
    assert np.abs(out - a @ w.T).max() <= 2e-4


def test_linear_epilogues_and_strides(cuda):
    rng = np.random.default_rng(0)
    a = torch.from_numpy(rng.standard_normal((200, 96)).astype(np.float32))
    w = torch.from_numpy(rng.standard_normal((160, 96)).astype(np.float32))
    b = torch.from_numpy(rng.standard_normal(160).astype(np.float32))
    r = torch.from_numpy(rng.standard_normal((200, 160)).astype(np.float32))
    ref = F.relu(a @ w.T + b) + r
    out = ops.linear(a.to(cuda), w.to(cuda), bias=b.to(cuda), residual=r.to(cuda), relu=True)
    assert (out.cpu() - ref).abs().max() <= 1e-4
    gel = ops.linear(a.to(cuda), w.to(cuda), bias=b.to(cuda), gelu=True)
    assert (gel.cpu() - F.gelu(a @ w.T + b)).abs().max() <= 1e-4
    # strided views: A is a column slice, output goes into a slice of a wider buffer
    big = torch.zeros((200, 400), device=cuda)
    wide = torch.from_numpy(rng.standard_normal((200, 192)).astype(np.float32)).to(cuda)
    ops.linear(wide[:, 96:], w.to(cuda), out=big[:, 40:200])
    assert (big[:, 40:200].cpu() - wide[:, 96:].cpu() @ w.T).abs().max() <= 1e-4
    assert big[:, :40].abs().max() == 0 and big[:, 200:].abs().max() == 0


def test_rmsnorm_layernorm_gather_scale(cuda):
    rng = np.random.default_rng(1)
    x = torch.from_numpy(rng.standard_normal((37, 768)).astype(np.float32) * 3)
    w = torch.from_numpy(rng.standard_normal(768).astype(np.float32))
    ref = w * (x / torch.sqrt(x.pow(2).mean(-1, keepdim=True) + 1e-6))
    assert (ops.rmsnorm(x.to(cuda), w.to(cuda), 1e-6).cpu() - ref).abs().max() <= 2e-6 * ref.abs().max()
    y = torch.from_numpy(rng.standard_normal((37, 768)).astype(np.float32))
    c = torch.from_numpy(rng.standard_normal(768).astype(np.float32))
    b = torch.from_numpy(rng.standard_normal(768).astype(np.float32))
    ref = F.layer_norm(x + y + c, (768,), w, b, 1e-5)
    got = ops.add_layernorm(x.to(cuda), y.to(cuda), w.to(cuda), b.to(cuda), 1e-5, cvec=c.to(cuda)).cpu()
    assert (got - ref).abs().max() <= 1e-5
    ref = F.layer_norm(x, (768,), w, b, 1e-5)
    assert (ops.add_layernorm(x.to(cuda), None, w.to(cuda), b.to(cuda)).cpu() - ref).abs().max() <= 1e-5
    idx = torch.from_numpy(rng.integers(0, 37, size=100))
    assert torch.equal(ops.gather_rows(x.to(cuda), idx.to(cuda)).cpu(), x[idx])
    assert torch.equal(ops.scale(x.to(cuda), 768 ** -0.5).cpu(), x * (768 ** -0.5))


@pytest.mark.parametrize("nb,tq,tk,H,dh,kv_div,causal,scale", [
    (6, 32, 32, 12, 64, 1, False, 1.0),     # encoder self-attention
    (20, 1, 32, 12, 64, 10, False, 1.0),    # decoder cross-attention: 10 beams share a query's K/V
    (20, 1, 5, 12, 64, 1, True, 1.0),       # decoder self-attention with cache (q at position 4)
    (8, 1, 3, 8, 96, 1, True, 96 ** -0.5),  # adaptor nn.MultiheadAttention, head dim 96
    (8, 32, 32, 4, 8, 1, False, 1.0),       # tiny heads (fixture models): fewer output dims than keys
    (6, 1, 6, 8, 4, 1, True, 0.5),
    (3, 128, 128, 12, 64, 1, False, 1.0),   # passage encoder self-attention (two keys per lane, 97 KiB LDS tile)
    (5, 1, 128, 12, 64, 1, False, 1.0),     # decoder cross-attention over a 128-token passage
    (2, 100, 100, 12, 64, 1, False, 1.0),   # matrix-core kernel, ragged sequence (padded keys and query rows)
    (2, 128, 128, 4, 64, 1, True, 0.5),     # matrix-core kernel, causal + scale
    (1, 65, 65, 3, 64, 1, False, 1.0),
    (2, 7, 200, 4, 32, 1, False, 1.0),      # four keys per lane, ragged tail
    (2, 130, 130, 2, 128, 1, True, 1.0),    # causal, tile kernel refused (LDS) -> wave-per-query kernel
    (3, 17, 17, 12, 64, 1, False, 1.0),     # query-length sequences: one wave per (sequence, head), 16 x 16 matrix-core blocks
    (2, 32, 32, 4, 64, 1, True, 0.5),       # the same, causal + scale, 2 x 2 blocks
    (5, 3, 3, 2, 64, 1, False, 1.0),        # the same, one block
    (30, 1, 21, 12, 64, 10, False, 1.0),    # decode-step cross-attention: ten beams per query, 21 keys (two key blocks)
    (7, 1, 12, 4, 64, 1, False, 0.5),       # the towers' single decoder position against 12 keys
    (64, 1, 9, 2, 64, 32, True, 1.0),       # 32 rows per group (two query blocks), causal at q_pos0
])
def test_attention(cuda, nb, tq, tk, H, dh, kv_div, causal, scale):
    rng = np.random.default_rng(nb * tk)
    q = torch.from_numpy(rng.standard_normal((nb, tq, H * dh)).astype(np.float32))
    k = torch.from_numpy(rng.standard_normal((nb // kv_div, tk, H * dh)).astype(np.float32)) * 0.3
    v = torch.from_numpy(rng.standard_normal((nb // kv_div, tk, H * dh)).astype(np.float32))
    q_pos0 = tk - tq if causal else 0
    bias = torch.from_numpy(rng.standard_normal((H, q_pos0 + tq, tk)).astype(np.float32))
    mask = torch.ones((nb // kv_div, tk), dtype=torch.int64)
    if not causal:
        for b in range(mask.shape[0]):
            mask[b, rng.integers(3, tk + 1):] = 0
    qh = q.view(nb, tq, H, dh).transpose(1, 2) * scale
    kh = k.repeat_interleave(kv_div, 0).view(nb, tk, H, dh).transpose(1, 2)
    vh = v.repeat_interleave(kv_div, 0).view(nb, tk, H, dh).transpose(1, 2)
    s = qh @ kh.transpose(-1, -2) + bias[None, :, q_pos0:q_pos0 + tq, :]
    s = s + (1.0 - mask.repeat_interleave(kv_div, 0)[:, None, None, :].float()) * -1e9
    if causal:
        s = s + (1.0 - torch.tril(torch.ones(tk, tk))[q_pos0:q_pos0 + tq])[None, None] * -1e9
    ref = (F.softmax(s, -1) @ vh).transpose(1, 2).reshape(nb, tq, H * dh)
    got = ops.attention(q.to(cuda), k.to(cuda), v.to(cuda), H, kv_div=kv_div, bias=bias.to(cuda), q_pos0=q_pos0,
                        key_mask=mask.to(cuda), causal=causal, scale=scale).cpu()
    assert (got - ref).abs().max() <= 2e-5


def test_adaptive_logits(cuda):
    rng = np.random.default_rng(3)
    rows, ncol, dim = 23, 33, 768
    s = torch.from_numpy(rng.standard_normal((rows, dim)).astype(np.float32)) * 0.05
    t = torch.from_numpy(rng.standard_normal((rows, ncol * dim)).astype(np.float32))
    e = torch.from_numpy(rng.standard_normal((ncol, dim)).astype(np.float32))
    ref = torch.einsum("rd,rcd->rc", s, t.view(rows, ncol, dim) + e[None])
    got = ops.adaptive_logits(s.to(cuda), t.to(cuda), e.to(cuda)).cpu()
    assert (got - ref).abs().max() <= 5e-5


@pytest.mark.parametrize("rows,ncol,dim,prefixes", [(23, 33, 768, 0), (301, 33, 768, 32), (70, 257, 768, 9), (19, 9, 64, 4),
                                                     (11, 65, 1000, 0), (5, 128, 256, 3)])
def test_adaptive_logits_rows_has_the_bits_of_scale_plus_adaptive_logits(cuda, rows, ncol, dim, prefixes):
    """mevi_adaptive_logits_rows_f32 (lm_head's rows inside the head matrices = the GEMM's bias; the d_model^-0.5 inside the
    kernel; the hidden state read once per 64 columns) against mevi_scale_f32 + mevi_adaptive_logits_f32: identical bits (at
    dim 768: the same products in the fused head's summation order, so within rounding), with and without a per-row table
    index, for column counts around the 64-column chunks and dims around the 256-float pieces."""
    rng = np.random.default_rng(rows + ncol)
    nt = prefixes if prefixes else rows
    s = torch.from_numpy(rng.standard_normal((rows, dim)).astype(np.float32)).to(cuda)
    t = torch.from_numpy(rng.standard_normal((nt, ncol * dim)).astype(np.float32)).to(cuda)
    e = torch.from_numpy(rng.standard_normal((ncol, dim)).astype(np.float32)).to(cuda)
    idx = torch.from_numpy(rng.integers(0, nt, size=rows)).to(cuda) if prefixes else None
    alpha = dim ** -0.5
    ref = ops.adaptive_logits(ops.scale(s, alpha), t, e, t_index=idx)
    te = t + e.reshape(1, -1)
    got = ops.adaptive_logits_rows(s, alpha, te, ncol, t_index=idx)
    if dim == 768:      # the summation order of the fused head (test_fused_head_... below): same products, another tree
        assert (got - ref).abs().max() <= 2e-5 * max(1.0, float(ref.abs().max()))
    else:
        assert torch.equal(got, ref)
    f64 = torch.einsum("rd,rcd->rc", s.double().cpu() * alpha,
                       (t.double().cpu()[idx.cpu()] if prefixes else t.double().cpu()).view(rows, ncol, dim) + e.double().cpu()[None])
    assert (got.double().cpu() - f64).abs().max() <= 1e-4 * max(1.0, dim / 768)


@pytest.mark.parametrize("rows,ncol", [(5000, 33), (777, 9), (300, 257)])
def test_fused_head_has_the_bits_of_head_gemm_plus_row_logits(cuda, rows, ncol, monkeypatch):
    """mevi_gemm_nt_split_head_f32 + mevi_logits_finish_f32 (the head matrices multiplied with the hidden states inside the GEMM's
    epilogue, never written) against mevi_gemm_nt_split_f32 with bias + mevi_adaptive_logits_rows_f32: identical bits -- a beam's
    logits do not depend on whether its head matrix came from a prefix table or from the fused kernel -- and both within rounding
    of float64; rows not a multiple of the 256-row tile."""
    dim = 768
    g = torch.Generator(device=cuda).manual_seed(rows + ncol)
    a = torch.randn((rows, dim), device=cuda, generator=g)
    w = torch.randn((ncol * dim, dim), device=cuda, generator=g) * dim ** -0.5
    e = torch.randn((ncol * dim,), device=cuda, generator=g)
    s = torch.randn((rows, dim), device=cuda, generator=g) * 3
    alpha = dim ** -0.5
    old = ops.GEMM_MODE
    try:
        ops.GEMM_MODE = "split"
        ws = ops.weight_split(w)
        assert hip_lib().mevi_gemm_nt_split_head_supported(rows, ncol * dim, dim, dim)
        fused = ops.head_logits(a, ws, e, s, alpha, ncol)
        monkeypatch.setattr(ops, "FUSED_HEAD", False)
        two = ops.head_logits(a, ws, e, s, alpha, ncol)
    finally:
        ops.GEMM_MODE = old
    assert torch.equal(fused, two)
    ref = torch.einsum("rd,rcd->rc", s.double() * alpha, (a.double() @ w.double().T + e.double()).view(rows, ncol, dim))
    assert (fused.double() - ref).abs().max() <= 1e-4 * float(ref.abs().max())


@pytest.mark.parametrize("nq,nb,K,R", [(7, 1, 32, 10), (7, 10, 32, 10), (3, 10, 256, 10), (5, 4, 16, 4)])
def test_beam_step(cuda, nq, nb, K, R):
    rng = np.random.default_rng(nq + nb + K)
    logits = torch.from_numpy(rng.standard_normal((nq * nb, K + 1)).astype(np.float32) * 3)
    bs = torch.from_numpy(-rng.random((nq, nb)).astype(np.float32) * 5)
    lsm = F.log_softmax(logits, -1).view(nq, nb, K + 1)
    cand = (bs[:, :, None] + lsm[:, :, 1:]).reshape(nq, nb * K)
    sc, parent, code = ops.beam_step(logits.to(cuda), bs.to(cuda), K, R)
    top = torch.topk(cand, R, dim=1)
    assert (sc.cpu() - top.values).abs().max() <= 2e-6
    flat = parent.cpu().long() * K + code.cpu().long()
    picked = torch.gather(cand, 1, flat)                       # same candidates up to f32-rounding ties
    assert (picked - top.values).abs().max() <= 2e-6
    fin = ops.beam_step(logits.to(cuda), bs.to(cuda), K, R, final_step=True).cpu()
    assert (fin - (bs + lsm[:, :, 0])).abs().max() <= 2e-6


def test_pair_dot_and_segment_sort(cuda):
    import ctypes
    rng = np.random.default_rng(4)
    A = rng.standard_normal((50, 768)).astype(np.float32)
    B = rng.standard_normal((400, 768)).astype(np.float32)
    ia = rng.integers(0, 50, size=1000)
    ib = rng.integers(0, 400, size=1000)
    got = ops.pair_dot(torch.from_numpy(A).to(cuda), torch.from_numpy(ia).to(cuda), torch.from_numpy(B).to(cuda),
                       torch.from_numpy(ib).to(cuda)).cpu().numpy()
    L = odense.lib()
    L.oracle_dot_f32.restype = ctypes.c_float
    L.oracle_dot_f32.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int64]
    ref = np.array([L.oracle_dot_f32(A[i].ctypes.data, B[j].ctypes.data, 768) for i, j in zip(ia, ib)], np.float32)
    assert np.array_equal(got.view(np.uint32), ref.view(np.uint32))          # bit-exact chain
    # segments incl. empty ones, duplicates (ties -> ascending id)
    lens = [0, 5, 1, 300, 0, 64, 630]
    seg = np.concatenate([[0], np.cumsum(lens)])
    sc = np.round(rng.standard_normal(seg[-1]).astype(np.float32), 1)
    ids = rng.permutation(100000)[: seg[-1]].astype(np.int64)
    os_, oi_ = ops.segment_sort_desc(torch.from_numpy(sc).to(cuda), torch.from_numpy(ids).to(cuda),
                                     torch.from_numpy(seg).to(cuda), max(lens))
    os_, oi_ = os_.cpu().numpy(), oi_.cpu().numpy()
    for a, b in zip(seg[:-1], seg[1:]):
        order = np.lexsort((ids[a:b], -sc[a:b]))
        assert np.array_equal(oi_[a:b], ids[a:b][order]) and np.array_equal(os_[a:b], sc[a:b][order])


def test_fine_stage_matches_reference_procedure(cuda):
    """FineStage.rerank vs a literal restatement of the reference loop (dict lookup per beam cluster,
    scores concatenated in beam order, sorted descending) with the oracle's fmaf-chain scores."""
    import ctypes
    from mevi_amd.fine import FineStage, coarse_ranks, fine_ranks, f32_repr
    from mevi_amd.rq import ClusterIndex
    from oracle import rq as orq

    rng = np.random.default_rng(12)
    N, dim, M, K, B, R = 3000, 64, 3, 6, 9, 5
    emb = rng.standard_normal((N, dim)).astype(np.float32)
    codes = rng.integers(0, K, size=(N, M)).astype(np.int32)
    cluster, mapping = orq.cluster_dict(codes)
    q = rng.standard_normal((B, dim)).astype(np.float32)
    beams = rng.integers(0, K, size=(B, R, M))
    beams[0] = codes[rng.integers(0, N, size=R)]                # populated clusters
    beams[1, :, :] = K - 1                                       # likely rare/empty clusters repeated
    fs = FineStage(torch.from_numpy(emb).to(cuda), ClusterIndex.from_codes(codes, K))
    out, ndoc = fs.rerank(torch.from_numpy(q).to(cuda), beams)
    L = odense.lib()
    L.oracle_dot_f32.restype = ctypes.c_float
    L.oracle_dot_f32.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int64]
    for b in range(B):
        docs, sc = [], []
        seen = set()
        for r in range(R):
            key = tuple(int(x) for x in beams[b, r])
            cur = cluster.get(key)
            if cur is None:
                continue
            # NOTE: a cluster repeated in the beam list is scored again by the reference as well
            docs += cur
            sc += [L.oracle_dot_f32(q[b].ctypes.data, emb[d].ctypes.data, dim) for d in cur]
            seen.add(key)
        assert ndoc[b] == len(docs)
        order = np.lexsort((np.array(docs, dtype=np.int64), -np.array(sc, dtype=np.float32))) if docs else []
        assert out[b][0].tolist() == [docs[i] for i in order]
        assert np.array_equal(out[b][1].view(np.uint32), np.array([sc[i] for i in order], np.float32).view(np.uint32))
    gts = [[int(x) for x in rng.integers(0, N, size=1 + b % 2)] for b in range(B)]
    gs = fs.gt_scores(torch.from_numpy(q).to(cuda), gts)
    for b in range(B):
        ref = np.array([L.oracle_dot_f32(q[b].ctypes.data, emb[d].ctypes.data, dim) for d in gts[b]], np.float32)
        assert np.array_equal(gs[b].view(np.uint32), ref.view(np.uint32))
    assert coarse_ranks(beams[0], [beams[0][2], [K, K, K]]) == (list(map(list, beams[0].tolist())).index(beams[0][2].tolist()), None)
    assert fine_ranks([5, 3, 9, 3], [9, 3, 7]) == (2, 1, None)
    assert f32_repr(np.array([0.1, 100.0], np.float32)) == "0.10000000149011612,100.0"


@pytest.mark.parametrize("S,H,dh,causal,scale,with_bias", [(32, 12, 64, False, 1.0, True), (40, 4, 16, False, 0.25, False),
                                                          (64, 2, 8, True, 1.0, True), (200, 3, 32, False, 1.0, True),
                                                          (128, 12, 64, False, 1.0, True), (100, 2, 64, False, 0.125, False),
                                                          (24, 12, 64, True, 1.0, True), (12, 3, 64, False, 0.125, False)])
def test_packed_attention_equals_padded_attention_bit_for_bit(cuda, S, H, dh, causal, scale, with_bias):
    """attention_varlen on packed rows (ragged lengths incl. 0, 1 and S) against attention on the zero-padded [B, S]
    layout with the key mask: identical bits on every real row; and against a torch fp32 softmax (5e-5)."""
    g = torch.Generator(device=cuda).manual_seed(S)
    lens = torch.tensor([S, 1, 0, 7, S // 2, 3, S - 1, 11, 0, 2], device=cuda)
    B, hd = len(lens), H * dh
    mask = (torch.arange(S, device=cuda)[None, :] < lens[:, None]).to(torch.int64)
    idx = torch.nonzero(mask.reshape(-1)).view(-1)
    packed = torch.randn((int(lens.sum()), 3 * hd), device=cuda, generator=g)
    bias = torch.randn((H, S, S), device=cuda, generator=g) if with_bias else None
    padded = torch.zeros((B * S, 3 * hd), device=cuda)
    padded[idx] = packed
    p3 = padded.view(B, S, 3 * hd)
    ref = ops.attention(p3[:, :, :hd], p3[:, :, hd:2 * hd], p3[:, :, 2 * hd:], H, bias=bias, key_mask=mask, causal=causal,
                        scale=scale).view(B * S, hd)[idx]
    off = torch.zeros(B + 1, dtype=torch.int64, device=cuda)
    off[1:] = torch.cumsum(lens, 0)
    got = ops.attention_varlen(packed[:, :hd], packed[:, hd:2 * hd], packed[:, 2 * hd:], off, S, H, bias=bias, causal=causal,
                               scale=scale)
    assert torch.equal(got, ref)
    q, k, v = (p3[:, :, i * hd:(i + 1) * hd].view(B, S, H, dh).transpose(1, 2) for i in range(3))
    sc = (q * scale) @ k.transpose(-1, -2)
    if bias is not None:
        sc = sc + bias[None]
    sc = sc + (1 - mask[:, None, None, :].float()) * -1e9
    if causal:
        sc = sc + torch.triu(torch.full((S, S), -1e9, device=cuda), 1)
    want = (torch.softmax(sc, -1) @ v).transpose(1, 2).reshape(B * S, hd)[idx]
    assert (got - want).abs().max().item() <= 5e-5


@pytest.mark.parametrize("kv_div,H,dh,S", [(1, 12, 64, 32), (10, 12, 64, 32), (4, 4, 16, 40), (1, 2, 8, 200)])
def test_cross_attention_over_packed_keys_equals_the_padded_masked_form(cuda, kv_div, H, dh, S):
    """Decoder cross-attention (tq = 1) reading a query's real encoder rows through kv_off against the zero-padded K|V
    with the key mask: identical bits (group kernel for kv_div > 1, generic kernel otherwise)."""
    g = torch.Generator(device=cuda).manual_seed(S + kv_div)
    lens = torch.tensor([S, 1, 7, S // 2, 3, S - 1, 11, 2], device=cuda)
    B, hd = len(lens), H * dh
    mask = (torch.arange(S, device=cuda)[None, :] < lens[:, None]).to(torch.int64)
    idx = torch.nonzero(mask.reshape(-1)).view(-1)
    packed = torch.randn((int(lens.sum()), 2 * hd), device=cuda, generator=g)
    padded = torch.zeros((B * S, 2 * hd), device=cuda)
    padded[idx] = packed
    p3 = padded.view(B, S, 2 * hd)
    q = torch.randn((B * kv_div, 1, hd), device=cuda, generator=g)
    ref = ops.attention(q, p3[:, :, :hd], p3[:, :, hd:], H, kv_div=kv_div, key_mask=mask)
    off = torch.zeros(B + 1, dtype=torch.int64, device=cuda)
    off[1:] = torch.cumsum(lens, 0)
    got = ops.attention(q, packed[:, :hd], packed[:, hd:], H, kv_div=kv_div, kv_off=off, kv_longest=S)
    assert torch.equal(got, ref)


@pytest.mark.parametrize("K,N", [(768, 768), (768, 3072), (3072, 768), (100, 130), (36, 64), (772, 25344)])
def test_few_row_gemm_gives_the_rows_of_the_large_gemm(cuda, K, N):
    """Small GEMMs (m * n <= 0.5 M outputs) take gemm_skinny_kernel; the same rows inside a 3000-row GEMM take the MFMA
    tiles: identical bits, with bias / ReLU / GELU / residual epilogues."""
    g = torch.Generator(device=cuda).manual_seed(K + N)
    x = torch.randn((3000, K), device=cuda, generator=g)
    w = torch.randn((N, K), device=cuda, generator=g) * K ** -0.5
    b = torch.randn((N,), device=cuda, generator=g)
    r = torch.randn((3000, N), device=cuda, generator=g)
    for kw in (dict(), dict(bias=b, relu=True), dict(bias=b, gelu=True, residual=r), dict(residual=r)):
        full = ops.linear(x, w, **kw)
        for M in (1, 2, 5, 31, 32, 255, 256, 257):
            kw_m = dict(kw)
            if "residual" in kw_m:
                kw_m["residual"] = r[:M]
            assert torch.equal(ops.linear(x[:M].contiguous(), w, **kw_m), full[:M]), (M, list(kw))


@pytest.mark.parametrize("tk,H,dh", [(1, 12, 64), (5, 12, 64), (8, 4, 32), (3, 2, 128)])
def test_attention_over_ancestor_indexed_caches_equals_the_reordered_copy(cuda, tk, H, dh):
    """mevi_attention_cached_f32: every row reads position j of its prefix from cache row key_rows[b, j]; the reference
    (and `attention` here) first copy the caches into beam order (generation_utils.py:927-934).  Identical bits."""
    g = torch.Generator(device=cuda).manual_seed(tk * 7 + H)
    rows, n, T = 37, 53, 9
    cache = torch.randn((rows, T, 2 * H * dh), device=cuda, generator=g)
    q = torch.randn((n, H * dh), device=cuda, generator=g)
    bias = torch.randn((H, T, T), device=cuda, generator=g)
    key_rows = torch.randint(0, rows, (n, tk), device=cuda, generator=g).to(torch.int32)
    ar = torch.arange(tk, device=cuda)
    gathered = cache[key_rows.long(), ar[None, :], :]                       # [n, tk, 2*H*dh]: the re-ordered copy
    want = ops.attention(q.view(n, 1, -1), gathered[:, :, :H * dh], gathered[:, :, H * dh:], H, bias=bias, q_pos0=tk - 1,
                         causal=True).view(n, -1)
    got = ops.attention_cached(q, cache[:, :, :H * dh], cache[:, :, H * dh:], key_rows, H, bias=bias, q_pos0=tk - 1, causal=True)
    assert torch.equal(got, want)



def test_staged_few_keys_attention_has_the_bits_of_the_direct_form(cuda):
    """attention_few_keys64_kernel (key rows transposed through LDS) against attention_few_keys_kernel (per-lane row walk,
    MEVI_ATTN_FEW_KEYS=direct; the switch is read once per process, hence the child processes): tk = 1..8, 12 x 64 and 8 x 96 heads, a
    row count that leaves the last wave ragged, beam-like ancestor tables."""
    import os
    import subprocess
    import sys

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    for nrows, heads, dh in (("371", "12", "64"), ("8", "12", "64"), ("205", "8", "96")):
        r = subprocess.run([sys.executable, os.path.join(root, "tools", "bench_attn_cached.py")], capture_output=True, text=True,
                           env=dict(os.environ, NROWS=nrows, HEADS=heads, DH=dh), timeout=600)
        assert r.returncode == 0 and r.stdout.count("same bits: True") == 8, r.stdout[-1500:] + r.stderr[-1500:]


def _image_to_f32(sr):
    """(hi + lo) * 2^-e of a SplitRows image, as f32 [rows, k]."""
    kp = sr.img.shape[1] // 2
    h = sr.img.view(torch.float16).to(torch.float64)
    x = (h[:, :sr.k] + h[:, kp:kp + sr.k]) * torch.pow(2.0, -sr.exp.to(torch.float64))[:, None]
    return x.to(torch.float32)


def test_split_precision_passage_attention_scales_its_operands(cuda):
    """attention_h16_kernel holds Q, K, V as f16 pairs under one power-of-two scale per operand and (sequence, head): values
    far beyond the f16 range (1e5) and far below it (1e-6) go through -- finite, and as close to the float64 result as the
    f32-MFMA kernel is; packed sequences (every length 65..128) equal the padded form bit for bit."""
    g = torch.Generator(device=cuda).manual_seed(12)
    H, dh = 12, 64
    hd = H * dh
    nb, S = 3, 97
    for qs, ks, vs in ((1.0, 1.0, 1.0), (300.0, 0.003, 1e5), (1e-3, 1e-3, 1e-6)):
        q = torch.randn((nb, S, hd), device=cuda, generator=g) * qs
        k = torch.randn((nb, S, hd), device=cuda, generator=g) * ks
        v = torch.randn((nb, S, hd), device=cuda, generator=g) * vs
        qd, kd, vd = (x.double().view(nb, S, H, dh).transpose(1, 2) for x in (q, k, v))
        ref = (torch.softmax(qd @ kd.transpose(-1, -2), -1) @ vd).transpose(1, 2).reshape(nb * S, hd)
        f32 = ops.attention(q, k, v, H).reshape(nb * S, hd)                          # f32-MFMA kernel (f32 output)
        img = _image_to_f32(ops.attention(q, k, v, H, split_bound=float(v.abs().max())))
        assert torch.isfinite(img).all()
        e32, e16 = (f32.double() - ref).abs().max().item(), (img.double() - ref).abs().max().item()
        assert e16 <= 1.5 * e32 + 2.0 ** -20 * float(v.abs().max()), (qs, ks, vs, e16, e32)
    lens = torch.tensor([65, 128, 90, 77], device=cuda)
    off = torch.zeros(5, dtype=torch.int64, device=cuda)
    off[1:] = lens.cumsum(0)
    T = int(off[-1])
    q, k, v = (torch.randn((T, hd), device=cuda, generator=g) for _ in range(3))
    bias = torch.randn((H, 128, 128), device=cuda, generator=g)
    vb = float(v.abs().max())
    packed = ops.attention_varlen(q, k, v, off, 128, H, bias=bias, split_bound=vb)
    for i, L in enumerate(lens.tolist()):
        a, b = int(off[i]), int(off[i + 1])
        pad = lambda x: torch.cat([x[a:b], torch.zeros((128 - L, hd), device=cuda)])[None]     # noqa: E731
        mask = (torch.arange(128, device=cuda) < L).long()[None]
        one = ops.attention(pad(q), pad(k), pad(v), H, bias=bias, key_mask=mask, split_bound=vb)
        assert torch.equal(one.img[:L], packed.img[a:b])


@pytest.mark.parametrize("form", ["padded", "cross_group", "cross_packed", "varlen", "varlen_mfma16", "cached", "passage_mfma"])
def test_attention_context_written_as_split_image(cuda, form):
    """mevi_attention*_split_f16: the context goes straight into the o-projection's (hi, lo) f16 image with ONE exponent from
    a bound on |V| (ops.ctx_bound).  The image must decode to the f32 kernel's context to 2^-21 of the bound's binade (22
    significant bits below the exponent's 2^15), and bound-sized values must not overflow."""
    g = torch.Generator(device=cuda).manual_seed(11)
    H, dh = 12, 64
    hd = H * dh
    rnd = lambda *s: torch.randn(s, device=cuda, generator=g)      # noqa: E731

    if form == "padded":
        nb, S = 3, 17
        q, k, v = rnd(nb, S, hd), rnd(nb, S, hd), rnd(nb, S, hd)
        mask = torch.ones((nb, S), dtype=torch.int64, device=cuda)
        mask[1, 9:] = 0
        kw = dict(bias=rnd(H, S, S), key_mask=mask)
        run = lambda **e: ops.attention(q, k, v, H, **kw, **e)     # noqa: E731  (tile kernel)
    elif form == "cross_group":
        nq, R, S = 5, 10, 21
        q, k, v = rnd(nq * R, 1, hd), rnd(nq, S, hd), rnd(nq, S, hd)
        run = lambda **e: ops.attention(q, k, v, H, kv_div=R, **e)     # noqa: E731
    elif form == "cross_packed":
        lens = torch.tensor([4, 9, 1, 30], device=cuda)
        off = torch.zeros(5, dtype=torch.int64, device=cuda)
        off[1:] = lens.cumsum(0)
        T = int(off[-1])
        q, k, v = rnd(4 * 10, 1, hd), rnd(T, hd), rnd(T, hd)
        run = lambda **e: ops.attention(q, k, v, H, kv_div=10, kv_off=off, kv_longest=30, **e)     # noqa: E731
    elif form == "varlen_mfma16":
        lens = torch.tensor([7, 12, 1, 30, 5, 16, 17, 8, 9], device=cuda)
        off = torch.zeros(10, dtype=torch.int64, device=cuda)
        off[1:] = lens.cumsum(0)
        T = int(off[-1])
        q, k, v, bias = rnd(T, hd), rnd(T, hd), rnd(T, hd), rnd(H, 32, 32)
        run = lambda **e: ops.attention_varlen(q, k, v, off, 30, H, bias=bias, **e)     # noqa: E731
    elif form == "varlen":
        lens = torch.tensor([7, 12, 1, 33, 5], device=cuda)
        off = torch.zeros(6, dtype=torch.int64, device=cuda)
        off[1:] = lens.cumsum(0)
        T = int(off[-1])
        q, k, v, bias = rnd(T, hd), rnd(T, hd), rnd(T, hd), rnd(H, 33, 33)
        run = lambda **e: ops.attention_varlen(q, k, v, off, 33, H, bias=bias, **e)     # noqa: E731
    elif form == "cached":
        n, T, tk = 53, 6, 4
        cache, q = rnd(n, T, 2 * hd), rnd(n, hd)
        kr = torch.randint(0, n, (n, tk), device=cuda, generator=g).to(torch.int32)
        b = rnd(H, T, T)
        run = lambda **e: ops.attention_cached(q, cache[:, :, :hd], cache[:, :, hd:], kr, H, bias=b, q_pos0=tk - 1, **e)     # noqa: E731
    else:
        nb, S = 2, 100
        q, k, v = rnd(nb, S, hd), rnd(nb, S, hd), rnd(nb, S, hd)
        run = lambda **e: ops.attention(q, k, v, H, **e)     # noqa: E731
    want = run()
    vmax = float(v.abs().max()) if form != "cached" else float(cache[:, :, hd:].abs().max())
    slack = 0.0
    if form == "passage_mfma":
        # with an image output the passage kernel is the split-precision f16 one (attention_h16_kernel): its arithmetic is not
        # the f32 kernel's, so both are held to a float64 reference -- the image kernel may be off by the image's own
        # quantisation plus what the f32 kernel itself is off by (its error is the yardstick of "f32-equivalent")
        qd, kd, vd = (x.double().view(nb, S, H, dh).transpose(1, 2) for x in (q, k, v))
        ref = (torch.softmax(qd @ kd.transpose(-1, -2), -1) @ vd).transpose(1, 2).reshape(nb, S, hd)
        slack = 1.5 * max((want.double() - ref).abs().max().item(), 1e-6)
        want = ref.float()
    for bound in (vmax, 37.0 * vmax):          # tight, and a few binades loose (what Cauchy-Schwarz gives)
        sr = run(split_bound=bound)
        assert isinstance(sr, ops.SplitRows) and sr.shape == (want.numel() // hd, hd)
        got = _image_to_f32(sr)
        e = int(sr.exp[0])
        assert (sr.exp == e).all() and 2.0 ** 14 <= bound * 1.001 * 2.0 ** e < 2.0 ** 15
        err = (got - want.reshape(-1, hd)).abs().max().item()
        assert err <= 2.0 ** (15 - e) * 2.0 ** -21 + slack, (form, bound, err)
    # and the o-projection takes it like any SplitRows
    w = ops.weight_split(rnd(hd, hd) * 0.05)
    y_img = ops.linear(run(split_bound=vmax), w)
    y_f32 = ops.linear(want.reshape(-1, hd).contiguous(), w)
    assert (y_img - y_f32).abs().max().item() <= 1e-5 * float(y_f32.abs().max())
