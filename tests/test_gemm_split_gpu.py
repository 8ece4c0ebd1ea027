"""Split-precision GEMM (csrc/gemm_split.hip: three f16 MFMAs per product on (hi, lo) float16 row images) against a
float64 reference.  Bar (stated per assertion): |error| <= 2e-6 * sum_k |a_k w_k| per output -- the f32 chain's own
worst case at these K is ~4e-7 of that sum, the split form measures ~1.5e-7 --, every epilogue, both kernels (tile
stream / few-rows), ragged shapes, strided outputs; and a row gives the same BITS alone or inside a large batch."""
import numpy as np
import pytest
import torch

from mevi_amd import ops

pytestmark = pytest.mark.gpu
TOL = 2e-6


def _ref(x, w, bias, act, residual):
    y = x.double() @ w.double().T
    den = x.double().abs() @ w.double().abs().T
    if bias is not None:
        y, den = y + bias.double(), den + bias.double().abs()
    if act == "relu":
        y = torch.relu(y)
    elif act == "gelu":
        y = torch.nn.functional.gelu(y)
    if residual is not None:
        y, den = y + residual.double(), den + residual.double().abs()
    return y, den


def _image_value(s):
    """f64 value of a SplitRows image: 2^-e (hi + lo)."""
    kp = s.img.shape[1] // 2
    h = s.img.view(torch.float16).double()
    return ((h[:, :kp] + h[:, kp:]) * torch.exp2(-s.exp.double())[:, None])[:, :s.k]


@pytest.mark.parametrize("M,N,K", [(3000, 768, 768), (517, 100, 36), (256, 96, 32), (1030, 2304, 768), (7, 3072, 768),
                                   (70000, 256, 64), (300, 36, 64), (1, 768, 3072), (2100, 260, 2048),
                                   (5, 100, 96), (300, 36, 152), (40, 64, 220), (3000, 800, 96)])   # Kp = 96, 160, 224: not whole 64-k chunks
def test_split_gemm_matches_float64(cuda, M, N, K):
    g = torch.Generator(device=cuda).manual_seed(M + N + K)
    x = torch.randn((M, K), device=cuda, generator=g) * torch.exp(2 * torch.randn((M, 1), device=cuda, generator=g))
    w = torch.randn((N, K), device=cuda, generator=g) * K ** -0.5 * torch.exp(torch.randn((N, 1), device=cuda, generator=g))
    b = torch.randn((N,), device=cuda, generator=g)
    r = torch.randn((M, N), device=cuda, generator=g)
    ws = ops.split_rows(w)
    xs = ops.split_rows(x)
    assert (_image_value(xs) - x.double()).abs().max() <= 2.0 ** -21 * x.abs().max()      # 22-bit operand images
    for kw, act in ((dict(), None), (dict(bias=b, relu=True), "relu"), (dict(bias=b, gelu=True), "gelu"),
                    (dict(residual=r), None), (dict(bias=b, residual=r), None)):
        got = ops.linear(xs, ws, **kw)
        ref, den = _ref(x, w, kw.get("bias"), act, kw.get("residual"))
        assert torch.isfinite(got).all()
        err = ((got.double() - ref).abs() / den.clamp_min(1e-30)).max().item()
        assert err <= TOL, (sorted(kw), err)
        assert torch.equal(ops.linear(x, ws, **kw), got)                                     # f32 input: split on the way in


@pytest.mark.parametrize("M,N,K", [(3000, 3072, 768), (300, 64, 32), (5, 2048, 768), (1100, 100, 768)])
@pytest.mark.parametrize("act", ["relu", "gelu", None])
def test_gemm_writing_a_split_image(cuda, M, N, K, act):
    """linear(..., for_gemm=True): the result as the next GEMM's operand image (exponent from the Cauchy-Schwarz
    bound): its value equals the f32 result to 22 bits of the row BOUND, and feeding it on gives the second layer."""
    g = torch.Generator(device=cuda).manual_seed(M + N)
    x = torch.randn((M, K), device=cuda, generator=g) * torch.exp(torch.randn((M, 1), device=cuda, generator=g))
    w1 = torch.randn((N, K), device=cuda, generator=g) * K ** -0.5
    b1 = torch.randn((N,), device=cuda, generator=g) * 0.1
    w2 = torch.randn((40, N), device=cuda, generator=g) * N ** -0.5
    xs, w1s, w2s = ops.split_rows(x), ops.weight_split(w1), ops.weight_split(w2)
    kw = dict(bias=b1, relu=act == "relu", gelu=act == "gelu")
    h32 = ops.linear(xs, w1s, **kw)
    hs = ops.linear(xs, w1s, for_gemm=True, **kw)
    assert isinstance(hs, ops.SplitRows) and hs.shape == (M, N)
    hv = _image_value(hs)
    bound = torch.exp2(15 - hs.exp.double())[:, None]                      # the row's exponent puts the bound below 2^15
    assert (h32.double().abs() <= bound).all()
    assert ((hv - h32.double()).abs() <= bound * 2.0 ** -36 + h32.double().abs() * 2.0 ** -21).all()
    assert (hs.norm.double() >= torch.linalg.vector_norm(h32.double(), dim=1)).all()
    y = ops.linear(hs, w2s)
    ref, den = _ref(h32, w2, None, None, None)
    assert ((y.double() - ref).abs() / den.clamp_min(1e-30)).max().item() <= TOL


def test_rows_keep_their_bits_in_any_batch(cuda):
    """Few rows (gemm_split_skinny_kernel) and the same rows inside a large batch (the tile stream): identical bits,
    for the f32 and the image output, with every epilogue."""
    g = torch.Generator(device=cuda).manual_seed(5)
    x = torch.randn((4000, 768), device=cuda, generator=g)
    b = torch.randn((3072,), device=cuda, generator=g)
    r = torch.randn((4000, 768), device=cuda, generator=g)
    wi, wo = ops.weight_split(torch.randn((3072, 768), device=cuda, generator=g) * 768 ** -0.5), \
        ops.weight_split(torch.randn((768, 3072), device=cuda, generator=g) * 3072 ** -0.5)
    xs = ops.split_rows(x)
    h = ops.linear(xs, wi, bias=b, relu=True, for_gemm=True)
    y = ops.linear(h, wo, residual=r)
    for m in (1, 5, 33, 100):
        hm = ops.linear(xs[:m], wi, bias=b, relu=True, for_gemm=True)
        assert torch.equal(hm.img, h.img[:m]) and torch.equal(hm.exp, h.exp[:m]) and torch.equal(hm.norm, h.norm[:m])
        assert torch.equal(ops.linear(hm, wo, residual=r[:m]), y[:m])
        assert torch.equal(ops.linear(xs[:m], wi, bias=b, relu=True), ops.linear(xs, wi, bias=b, relu=True)[:m])


def test_rows_keep_their_bits_across_tile_heights_and_the_row_split(cuda):
    """Round 5: the tile stream runs 256 / 128 / 64 activation rows per tile (chosen per GEMM by its tile rounds) and cuts GEMMs of
    several rounds into whole rounds + a remainder launch.  Whatever batch a row travels in -- 600 ... 20 000 rows from the head
    or the middle of a 70 000-row batch -- it has the bits it has in the big batch: f32 + residual, plain f32, and the image."""
    g = torch.Generator(device=cuda).manual_seed(11)
    M = 70000
    x = torch.randn((M, 768), device=cuda, generator=g)
    r = torch.randn((M, 768), device=cuda, generator=g)
    b = torch.randn((2304,), device=cuda, generator=g)
    wo = ops.weight_split(torch.randn((768, 768), device=cuda, generator=g) * 768 ** -0.5)
    wq = ops.weight_split(torch.randn((2304, 768), device=cuda, generator=g) * 768 ** -0.5)
    xs = ops.split_rows(x)
    y = ops.linear(xs, wo, residual=r)
    q = ops.linear(xs, wq, bias=b)
    h = ops.linear(xs, wq, bias=b, relu=True, for_gemm=True)
    for lo, m in ((0, 600), (0, 873), (0, 1418), (0, 2048), (0, 4096), (0, 9600), (0, 20000), (30000, 873), (41111, 5000), (69000, 1000)):
        sub = xs[lo:lo + m]
        assert torch.equal(ops.linear(sub, wo, residual=r[lo:lo + m]), y[lo:lo + m]), (lo, m)
        assert torch.equal(ops.linear(sub, wq, bias=b), q[lo:lo + m]), (lo, m)
        hm = ops.linear(sub, wq, bias=b, relu=True, for_gemm=True)
        assert torch.equal(hm.img, h.img[lo:lo + m]) and torch.equal(hm.exp, h.exp[lo:lo + m]), (lo, m)


def test_strided_output_and_fused_rmsnorm(cuda):
    g = torch.Generator(device=cuda).manual_seed(9)
    x = torch.randn((1500, 768), device=cuda, generator=g) * 30
    lnw = 1 + 0.1 * torch.randn((768,), device=cuda, generator=g)
    w = ops.weight_split(torch.randn((1536, 768), device=cuda, generator=g) * 768 ** -0.5)
    h = ops.rmsnorm(x, lnw, 1e-6)
    old = ops.GEMM_MODE
    try:
        ops.GEMM_MODE = "split"
        hs = ops.rmsnorm(x, lnw, 1e-6, for_gemm=True)                       # the norm written as an operand image
    finally:
        ops.GEMM_MODE = old
    ref = ops.split_rows(h)
    assert torch.equal(hs.img, ref.img) and torch.equal(hs.exp, ref.exp) and torch.equal(hs.norm, ref.norm)
    cache = torch.full((1500, 5, 1536), float("nan"), device=cuda)
    ops.linear(hs, w, out=cache[:, 2, :])                                   # the decoder's K|V cache slot
    assert torch.equal(cache[:, 2, :], ops.linear(hs, w)) and torch.isnan(cache[:, 1, :]).all() and torch.isnan(cache[:, 3, :]).all()


def test_extreme_rows(cuda):
    """Row scaling across the f32 range: all-zero rows, rows of magnitude 1e-30 and 1e+30 (exponents at the int8 clamp
    region), one huge element beside tiny ones -- same 2e-6 bar relative to sum |a||w|; non-finite inputs stay non-finite
    in their own rows only."""
    g = torch.Generator(device=cuda).manual_seed(11)
    M, N, K = 700, 256, 512
    x = torch.randn((M, K), device=cuda, generator=g)
    x[0] = 0
    x[1] *= 1e-30
    x[2] *= 1e30
    x[3, 1:] *= 1e-6
    x[3, 0] = 5e4
    w = torch.randn((N, K), device=cuda, generator=g) * K ** -0.5
    w[7] = 0
    w[8] *= 1e-5                    # (1e-30 rows x this column stay above the f32 underflow any GEMM would hit)
    got = ops.linear(ops.split_rows(x), ops.split_rows(w))
    ref, den = _ref(x, w, None, None, None)
    assert torch.isfinite(got).all() and (got[0] == 0).all() and (got[:, 7] == 0).all()
    err = ((got.double() - ref).abs() / den.clamp_min(1e-300))
    err[0], err[:, 7] = 0, 0
    assert err.max().item() <= TOL
    x[5, 9], x[6, 3] = float("inf"), float("nan")
    got = ops.linear(ops.split_rows(x), ops.split_rows(w))
    bad = ~torch.isfinite(got)
    assert bad[5].any() and bad[6].any() and not bad[[0, 1, 2, 3, 4, 7, 8]].any()


def test_norm_and_projection_in_one_kernel_keep_the_two_kernel_bits(cuda, monkeypatch):
    """mevi_gemm_nt_rmsnorm_split_* (the latency path: every workgroup normalises its rows itself) against
    mevi_rmsnorm_split_f16 + mevi_gemm_nt_split_*: identical outputs -- f32 with bias / relu / residual / strided output, and
    the image-writing form (image, exponents, norms) -- for 1 .. 40 rows; unsupported shapes fall back silently."""
    monkeypatch.setattr(ops, "FUSE_NORM", True)       # off by default (measured no faster): MEVI_FUSE_NORM=1
    g = torch.Generator(device=cuda).manual_seed(21)
    rnd = lambda *s_: torch.randn(s_, device=cuda, generator=g)      # noqa: E731
    lnw = 1 + 0.1 * rnd(768)
    for M, N in ((1, 768), (10, 2304), (11, 3072), (32, 768), (33, 1536), (40, 1536)):
        x = rnd(M, 768) * torch.exp(rnd(M, 1))
        w = ops.weight_split(rnd(N, 768) * 768 ** -0.5)
        b, r = rnd(N) * 0.1, rnd(M, N)
        assert ops.FUSE_NORM and hip_lib().mevi_gemm_rmsnorm_supported(M, N, 768)
        two = lambda **kw: ops.linear(ops._rmsnorm_split(x, lnw, 1e-6), w, **kw)         # noqa: E731
        one = lambda **kw: ops.linear(ops.rmsnorm(x, lnw, 1e-6, for_gemm=True), w, **kw)  # noqa: E731
        assert isinstance(ops.rmsnorm(x, lnw, 1e-6, for_gemm=True), ops.NormedRows)
        for kw in ({}, {"bias": b}, {"bias": b, "relu": True}, {"residual": r}, {"bias": b, "residual": r, "gelu": True}):
            assert torch.equal(one(**kw), two(**kw)), (M, N, sorted(kw))
        big = torch.full((M, 3, N), float("nan"), device=cuda)
        one(out=big[:, 1, :])
        assert torch.equal(big[:, 1, :], two()) and torch.isnan(big[:, 0, :]).all()
        h1, h2 = one(bias=b, relu=True, for_gemm=True), two(bias=b, relu=True, for_gemm=True)
        assert torch.equal(h1.img, h2.img) and torch.equal(h1.exp, h2.exp) and torch.equal(h1.norm, h2.norm)
    # not on the fused path: many rows (the tile stream), another width -- the lazy rows materialise, same results
    x = rnd(3000, 768)
    w = ops.weight_split(rnd(768, 768) * 0.03)
    nr = ops.rmsnorm(x[:300], lnw, 1e-6, for_gemm=True)
    assert not isinstance(nr, ops.NormedRows) or torch.equal(ops.linear(nr, w), ops.linear(ops._rmsnorm_split(x[:300], lnw, 1e-6), w))
    assert isinstance(ops.rmsnorm(x, lnw, 1e-6, for_gemm=True), ops.SplitRows)


def hip_lib():
    from mevi_amd import hip

    return hip.lib()
