"""Thin tensor-level wrappers over the C ABI (include/mevi_hip.h).  torch supplies device memory
and the current stream only; every function launches hand-written HIP kernels and raises
MeviHipError when the extension or the GPU is missing."""
import math
import os

import torch

from . import hip

# How the linear layers of the T5 / BERT / adaptor stacks multiply (read when a model prepares its weights):
#   "split" (default) -- split-precision GEMM, three f16 MFMAs per product on (hi, lo) f16 images (csrc/gemm_split.hip)
#   "exact"           -- the sequential f32 fmaf-chain GEMM on the f32 matrix cores (csrc/gemm.hip)
# The dense arm, the fine stage and the RQ kernels always use their exact f32 chains.
GEMM_MODE = os.environ.get("MEVI_GEMM", "split")


def _f32(t):
    assert t.is_cuda and t.dtype == torch.float32, (t.device, t.dtype)
    return t


def _rows2d(t):
    """(pointer tensor, rows, cols, row stride) of a 2-D view whose last dim is contiguous."""
    assert t.dim() == 2 and t.stride(1) == 1, (t.shape, t.stride())
    return t, t.shape[0], t.shape[1], t.stride(0)


class SplitRows:
    """f32 rows held as (hi | lo) float16 images with one power-of-two exponent per row (csrc/gemm_split.hip):
    the operand format of the split-precision GEMM.  `img` i16 [rows, 2 * kp], `exp` i8 [rows]; `norm` f32 [rows]
    (activations: l2 norm of every row, rounded up) and `norm_max` / `bias_abs_max` (weights: the largest row norm)
    feed the exponent bound of a GEMM that writes its result as an image (linear(..., for_gemm=True))."""

    __slots__ = ("img", "exp", "k", "norm", "norm_max")

    def __init__(self, img, exp, k, norm=None, norm_max=None):
        self.img, self.exp, self.k, self.norm, self.norm_max = img, exp, k, norm, norm_max

    @property
    def shape(self):
        return (self.img.shape[0], self.k)

    @property
    def device(self):
        return self.img.device

    def __getitem__(self, rows):
        assert isinstance(rows, slice) and rows.step in (None, 1)
        return SplitRows(self.img[rows], self.exp[rows], self.k, None if self.norm is None else self.norm[rows], self.norm_max)

    def contiguous(self):
        return self


def weight_split(w, keep_norm=False):
    """Split image of a static [out, in] weight, with the largest row norm (the bound of image-writing GEMMs)."""
    ws = split_rows(w.contiguous())
    ws.norm_max = float(ws.norm.max().item()) if ws.norm.numel() else 0.0
    if not keep_norm:
        ws.norm = None
    return ws


def weight(w, keep_norm=False):
    """A linear layer's [out, in] weight in the operand format of the current GEMM_MODE (static: converted once).
    keep_norm: the row norms stay on a split image until weight_rows() has cut it into per-layer weights."""
    if GEMM_MODE == "split":
        return weight_split(w, keep_norm)
    assert GEMM_MODE == "exact", f"MEVI_GEMM must be 'split' or 'exact', not {GEMM_MODE!r}"
    return w


def weight_rows(w, a, b):
    """Output rows [a, b) of a prepared weight as a weight of their own (views; the slice's own largest row norm)."""
    if isinstance(w, SplitRows):
        s = w[a:b]
        if s.norm is not None and s.norm.numel():
            s.norm_max, s.norm = float(s.norm.max().item()), None
        return s
    return w[a:b]


def prepare_weights(layers, keys):
    for L in layers:
        for k in keys:
            L[k] = weight(L[k])


def _abs_max(t):
    """max |t| of a static tensor (a bias), computed once and kept ON the tensor (no host synchronisation per call; a
    cache keyed by address could hand a new model's bias the value of a freed one that lived at the same address)."""
    v = getattr(t, "_mevi_abs_max", None)
    if v is None:
        v = float(t.abs().max().item())
        t._mevi_abs_max = v
    return v


def gemm_input(x):
    """x f32 [m, k] as the A operand of linear() under the current GEMM_MODE: itself, or its split image (made once
    for all the layers that read it)."""
    return split_rows(x) if GEMM_MODE == "split" and not isinstance(x, SplitRows) else x


def split_kp(k):
    """mevi_split_kp (csrc/gemm_split.hip) without the foreign call: whole 32-wide slabs, at least two."""
    return 64 if k <= 64 else (k + 31) // 32 * 32


def _split_buffers(M, K, dev, zero=False):
    kp = split_kp(K)
    img = (torch.zeros if zero and kp != K else torch.empty)((M, 2 * kp), dtype=torch.int16, device=dev)
    return img, torch.empty((M,), dtype=torch.int8, device=dev), torch.empty((M,), dtype=torch.float32, device=dev)


CTX_IMAGE = os.environ.get("MEVI_ATTN_CTX", "image") != "f32"      # A/B switch: attention contexts as f32 + split_rows
_EXP_FILL = {}
_EXP_FILL_RETIRED = []     # see _ctx_image


def norm_out_bound(ln_weight, d_model, ln_bias=None):
    """l2 norm bound of a row of rmsnorm(x) * w (T5LayerNorm) or layernorm(x) * w + b: sqrt(d) max|w| (+ sqrt(d) max|b|)."""
    xn = math.sqrt(d_model) * _abs_max(ln_weight)
    if ln_bias is not None:
        xn += math.sqrt(d_model) * _abs_max(ln_bias)
    return xn


def ctx_bound(x_norm, w, v_bias=None):
    """Bound on |V| = |x W^T (+ b)| over rows x with ||x|| <= x_norm (norm_out_bound), for an attention whose context only
    feeds the o-projection: the context is a convex combination of V rows, so with this bound it can be written as that
    GEMM's split image directly (one exponent for all rows; `split_bound` of the attention wrappers).  Cauchy-Schwarz with
    the largest weight row norm.  None unless the split GEMM is in use (then the f32 form runs)."""
    if x_norm is None or GEMM_MODE != "split" or not CTX_IMAGE or not isinstance(w, SplitRows) or w.norm_max is None:
        return None
    b = x_norm * w.norm_max * 1.001
    if v_bias is not None:
        b += _abs_max(v_bias)
    return b


def _pow2_exp(m):
    """e with m * 2^e in [2^14, 2^15), clamped to +-100 (pow2_exp of gemm_split.hip)."""
    if not (m > 0.0) or math.isinf(m):
        return 0
    return max(-100, min(100, 15 - math.frexp(m)[1]))


def _ctx_image(rows, k, bound, dev):
    """SplitRows for `rows` context rows of width k, all with the exponent of `bound`; returns (image, np, exponent)."""
    kp = split_kp(k)
    img = (torch.zeros if kp != k else torch.empty)((rows, 2 * kp), dtype=torch.int16, device=dev)
    e = _pow2_exp(bound * 1.001)
    fill = _EXP_FILL.get((dev, e))
    if fill is None or fill.numel() < rows:
        if fill is not None:
            # a HIP graph captured earlier (t5.GraphCache: the <= 8-row latency path) holds this buffer's raw address as the
            # exponent array of its o-projection GEMM and does not keep it alive: a superseded fill is retired, never freed
            _EXP_FILL_RETIRED.append(fill)
        fill = _EXP_FILL[(dev, e)] = torch.full((max(rows, 1 << 16),), e, dtype=torch.int8, device=dev)
    # (norm_max of an ACTIVATION image: one bound on every row's l2 norm -- sqrt(k) |element| <= sqrt(k) bound -- for the GEMM
    # that adds this context to the residual stream, linear_residual)
    return SplitRows(img, fill[:rows], k, None, math.sqrt(k) * bound * 1.001), kp, e


def split_rows(x):
    """SplitRows image of x f32 [m, k] (last dim contiguous)."""
    x, M, K, ldx = _rows2d(_f32(x))
    with hip.device_guard(x.device):
        img, exp, norm = _split_buffers(M, K, x.device)
        st = hip.lib().mevi_split_rows_f16(hip.ptr(x), ldx, M, K, hip.ptr(img), hip.ptr(exp), hip.ptr(norm), hip.stream_ptr())
    hip.check(st, "mevi_split_rows_f16")
    return SplitRows(img, exp, K, norm)


def _linear_normed(nr, weight, bias, residual, act, out, for_gemm):
    """rmsnorm + projection in one launch where the C ABI has the shape on its fused path; else None."""
    x = nr.x
    M, K = x.shape
    N = weight.shape[0]
    L = hip.lib()
    if nr._rows is not None or not L.mevi_gemm_rmsnorm_supported(M, N, K):
        return None
    if x.stride(1) != 1 or x.stride(0) % 4 or x.data_ptr() % 16 or nr.weight.data_ptr() % 16 or \
            (bias is not None and bias.data_ptr() % 16):
        return None
    dev = weight.device
    with hip.device_guard(dev):
        if for_gemm:
            if residual is not None or out is not None or weight.norm_max is None:
                return None
            babs = _abs_max(bias) if bias is not None else 0.0
            img, exp, norm = _split_buffers(M, N, dev, zero=True)
            st = L.mevi_gemm_nt_rmsnorm_split_to_split(hip.ptr(x), x.stride(0), hip.ptr(nr.weight), nr.eps, hip.ptr(weight.img),
                                                       hip.ptr(weight.exp), weight.norm_max, M, N, K,
                                                       hip.ptr(bias) if bias is not None else None, babs, act, hip.ptr(img),
                                                       hip.ptr(exp), hip.ptr(norm), hip.stream_ptr())
            hip.check(st, "mevi_gemm_nt_rmsnorm_split_to_split")
            return SplitRows(img, exp, N, norm)
        if out is None:
            out = torch.empty((M, N), dtype=torch.float32, device=dev)
        if out.shape != (M, N) or out.stride(1) != 1 or out.stride(0) % 4 or out.data_ptr() % 16:
            return None
        ldr = 0
        if residual is not None:
            if residual.shape != (M, N) or residual.stride(1) != 1 or residual.stride(0) % 4 or residual.data_ptr() % 16:
                return None
            ldr = residual.stride(0)
        st = L.mevi_gemm_nt_rmsnorm_split_f32(hip.ptr(x), x.stride(0), hip.ptr(nr.weight), nr.eps, hip.ptr(weight.img),
                                              hip.ptr(weight.exp), hip.ptr(out), out.stride(0), M, N, K,
                                              hip.ptr(bias) if bias is not None else None,
                                              hip.ptr(residual) if residual is not None else None, ldr, act, hip.stream_ptr())
        hip.check(st, "mevi_gemm_nt_rmsnorm_split_f32")
        return out


def _linear_split(x, weight, bias, residual, act, out, for_gemm):
    if isinstance(x, NormedRows):
        y = _linear_normed(x, weight, bias, residual, act, out, for_gemm)
        if y is not None:
            return y
        x = x.rows()
    xs = x if isinstance(x, SplitRows) else split_rows(x)
    (M, K), (N, K2) = xs.shape, weight.shape
    assert K == K2, (xs.shape, weight.shape)
    dev = weight.device
    L = hip.lib()
    if for_gemm:       # act(x W^T + b) as the next GEMM's operand
        assert residual is None and out is None and xs.norm is not None and weight.norm_max is not None
        babs = _abs_max(bias) if bias is not None else 0.0
        with hip.device_guard(dev):
            img, exp, norm = _split_buffers(M, N, dev, zero=True)
            st = L.mevi_gemm_nt_split_to_split(hip.ptr(xs.img), hip.ptr(xs.exp), hip.ptr(xs.norm), hip.ptr(weight.img),
                                               hip.ptr(weight.exp), weight.norm_max, M, N, K,
                                               hip.ptr(bias) if bias is not None else None, babs, act, hip.ptr(img),
                                               hip.ptr(exp), hip.ptr(norm), hip.stream_ptr())
        hip.check(st, "mevi_gemm_nt_split_to_split")
        return SplitRows(img, exp, N, norm)
    if out is None:
        out = torch.empty((M, N), dtype=torch.float32, device=dev)
    assert out.shape == (M, N) and out.stride(1) == 1
    ldr = 0
    if residual is not None:
        assert residual.shape == (M, N) and residual.stride(1) == 1
        ldr = residual.stride(0)
    with hip.device_guard(dev):
        st = L.mevi_gemm_nt_split_f32(hip.ptr(xs.img), hip.ptr(xs.exp), hip.ptr(weight.img), hip.ptr(weight.exp),
                                      hip.ptr(out), out.stride(0), M, N, K,
                                      hip.ptr(bias) if bias is not None else None,
                                      hip.ptr(residual) if residual is not None else None, ldr, act, hip.stream_ptr())
    hip.check(st, "mevi_gemm_nt_split_f32")
    return out


@hip.on_device
def linear(x, weight, bias=None, residual=None, relu=False, out=None, gelu=False, for_gemm=False):
    """out = act(x @ weight.T + bias) + residual   (x [M,K], weight [N,K] = nn.Linear layout; act = relu, erf-gelu
    or none).  f32 tensors: the exact sequential-chain GEMM; `weight` a SplitRows (and x f32 or SplitRows): the
    split-precision GEMM (three f16 MFMAs per product).  for_gemm: the result only feeds another linear() -- with
    split weights it is written as a SplitRows image directly (no f32 round trip)."""
    assert not (relu and gelu)
    act = 1 if relu else (2 if gelu else 0)
    if isinstance(weight, SplitRows):
        return _linear_split(x, weight, bias, residual, act, out, for_gemm)
    x, M, K, lda = _rows2d(_f32(x))
    w, N, K2, ldw = _rows2d(_f32(weight))
    assert K == K2, (x.shape, weight.shape)
    if out is None:
        out = torch.empty((M, N), dtype=torch.float32, device=x.device)
    assert out.shape == (M, N) and out.stride(1) == 1
    ldr = 0
    if residual is not None:
        assert residual.shape == (M, N) and residual.stride(1) == 1
        ldr = residual.stride(0)
    st = hip.lib().mevi_gemm_nt_f32(hip.ptr(x), lda, hip.ptr(w), ldw, hip.ptr(out), out.stride(0), M, N, K,
                                    hip.ptr(bias) if bias is not None else None,
                                    hip.ptr(residual) if residual is not None else None, ldr, act,
                                    hip.stream_ptr())
    hip.check(st, "mevi_gemm_nt_f32")
    return out


# ---- T5LayerNorm folded into the linear layers around it (include/mevi_hip.h, "T5LayerNorm folded ...") ---------------------------
# Built in round 5, every golden and oracle comparison holds with it, and it is OFF by default because it measured
# SLOWER where it was meant to pay: NCI generate 166.9 -> 174.7 ms per 6980 queries, tower 60.0 -> 62.0 ms, 873 queries 27.2 -> 28.3 ms
# (one box, profiles/r05_fold_norm.txt).  The separate pass streams at 6 TB/s (60 us per 76.9 k rows); writing the stream's image in
# the producing GEMM's epilogue -- 64-byte store segments, behind the f32 rows the next residual add still needs -- costs as much,
# and the block sums (48 scattered 4-byte stores per row) and the scale launch come on top.  Only the batch-1 graphs gain (launches:
# tower 1.33 -> 1.29 ms, NCI 3.11 -> 3.05 ms), and a row must have the same bits in any batch, so there is no folding small batches
# only.  MEVI_FOLD_NORM=1 turns it on (A/B; tests/test_t5_gpu.py runs the goldens both ways).
FOLD_NORM = os.environ.get("MEVI_FOLD_NORM", "0") == "1"


def fold_norm_ok(d_model):
    """Whether the T5 stacks of width d_model run with their norms folded: split GEMM, a width the stream kernels take."""
    return bool(FOLD_NORM and GEMM_MODE == "split" and hip.lib().mevi_gemm_norm_fold_supported(int(d_model)))


def fold_weight(w, ln):
    """W (.) w_ln: the [out, in] weight of the projection a T5LayerNorm with weight `ln` feeds, with the norm's weight folded in."""
    return (w * ln[None, :]).contiguous()


class ResidualRows:
    """A T5 stack's residual stream with its norms folded away: x f32 [M, D], the split image of x (img, exp), bound f32 [M]
    >= max |row| and ssq f32 [M, D / 16] = sums of squares of the rows' 16-column blocks.  The projections that follow a
    T5LayerNorm read the image and scale their product per row (linear_normed); the projections that add to the stream write
    all five next to each other (linear_residual)."""

    __slots__ = ("x", "img", "exp", "bound", "ssq")

    def __init__(self, x, img, exp, bound, ssq):
        self.x, self.img, self.exp, self.bound, self.ssq = x, img, exp, bound, ssq

    @property
    def shape(self):
        return tuple(self.x.shape)

    @property
    def device(self):
        return self.x.device


def _stream_buffers(M, D, dev):
    kp = split_kp(D)
    img = (torch.zeros if kp != D else torch.empty)((M, 2 * kp), dtype=torch.int16, device=dev)
    return (img, torch.empty((M,), dtype=torch.int8, device=dev), torch.empty((M,), dtype=torch.float32, device=dev),
            torch.empty((M, D // 16), dtype=torch.float32, device=dev))


@hip.on_device
def residual_start(x):
    """Where a stack's residual stream starts (the token embeddings): x f32 [M, D] -> ResidualRows."""
    x, M, D, ldx = _rows2d(_f32(x))
    img, exp, bound, ssq = _stream_buffers(M, D, x.device)
    st = hip.lib().mevi_split_rows_ssq_f16(hip.ptr(x), ldx, M, D, hip.ptr(img), hip.ptr(exp), hip.ptr(bound), hip.ptr(ssq), hip.stream_ptr())
    hip.check(st, "mevi_split_rows_ssq_f16")
    return ResidualRows(x, img, exp, bound, ssq)


def linear_normed(r, weight, eps, bias=None, relu=False, out=None, for_gemm=False):
    """act(rmsnorm(x) W^T + b) for the stream r, `weight` = split image of W (.) w_ln: the image of the RAW rows is multiplied and
    every row of the product scaled by rsqrt(mean x^2 + eps) (mevi_gemm_nt_split_normed_*)."""
    M, K = r.shape
    N = weight.shape[0]
    assert isinstance(weight, SplitRows) and weight.shape[1] == K
    dev = weight.device
    L = hip.lib()
    act = 1 if relu else 0
    with hip.device_guard(dev):
        ws = torch.empty((M,), dtype=torch.float32, device=dev)
        np_ = r.ssq.shape[1]
        if for_gemm:
            assert out is None and weight.norm_max is not None
            babs = _abs_max(bias) if bias is not None else 0.0
            img, exp, norm = _split_buffers(M, N, dev, zero=True)
            st = L.mevi_gemm_nt_split_normed_to_split(hip.ptr(r.img), hip.ptr(r.exp), hip.ptr(r.ssq), np_, float(K), eps, hip.ptr(ws),
                                                      math.sqrt(K) * 1.0001, hip.ptr(weight.img), hip.ptr(weight.exp), weight.norm_max,
                                                      M, N, K, hip.ptr(bias) if bias is not None else None, babs, act, hip.ptr(img),
                                                      hip.ptr(exp), hip.ptr(norm), hip.stream_ptr())
            hip.check(st, "mevi_gemm_nt_split_normed_to_split")
            return SplitRows(img, exp, N, norm)
        if out is None:
            out = torch.empty((M, N), dtype=torch.float32, device=dev)
        assert out.shape == (M, N) and out.stride(1) == 1
        st = L.mevi_gemm_nt_split_normed_f32(hip.ptr(r.img), hip.ptr(r.exp), hip.ptr(r.ssq), np_, float(K), eps, hip.ptr(ws),
                                             hip.ptr(weight.img), hip.ptr(weight.exp), hip.ptr(out), out.stride(0), M, N, K,
                                             hip.ptr(bias) if bias is not None else None, None, 0, act, hip.stream_ptr())
    hip.check(st, "mevi_gemm_nt_split_normed_f32")
    return out


def linear_residual(a, weight, r, normed_eps=None):
    """The stream's next state x + a W^T (T5's `hidden_states + dropout(y)`), as ResidualRows: f32 rows, image, bound and block sums
    from the GEMM's epilogue.  `a`: a SplitRows with per-row norms (`norm`) or one bound for all rows (`norm_max`: attention
    contexts) -- or, with `normed_eps`, the stream itself (x + rmsnorm(x) W^T: the one-position decoder's folded o(v(.)))."""
    assert isinstance(weight, SplitRows) and weight.norm_max is not None
    if torch.is_tensor(a):        # an f32 operand (attention contexts with MEVI_ATTN_CTX=f32): its image, with the rows' norms
        a = split_rows(a)
    M, D = r.shape
    N, K = weight.shape
    assert N == D
    dev = weight.device
    L = hip.lib()
    with hip.device_guard(dev):
        x = torch.empty((M, D), dtype=torch.float32, device=dev)
        img, exp, bound, ssq = _stream_buffers(M, D, dev)
        if normed_eps is not None:
            assert a is r and K == D
            ws = torch.empty((M,), dtype=torch.float32, device=dev)
            a_img, a_exp, a_norm, a_const = r.img, r.exp, None, math.sqrt(K) * 1.0001
            parts, np_, rd, re, wsp = hip.ptr(r.ssq), r.ssq.shape[1], float(K), normed_eps, hip.ptr(ws)
        else:
            assert isinstance(a, SplitRows) and a.shape == (M, K)
            a_img, a_exp = a.img, a.exp
            a_norm, a_const = (a.norm, 0.0) if a.norm is not None else (None, float(a.norm_max))
            assert a_norm is not None or a.norm_max is not None
            parts, np_, rd, re, wsp = None, 0, 0.0, 0.0, None
        st = L.mevi_gemm_nt_split_residual_stream(hip.ptr(a_img), hip.ptr(a_exp), hip.ptr(a_norm) if a_norm is not None else None, a_const,
                                                  parts, np_, rd, re, wsp, hip.ptr(weight.img), hip.ptr(weight.exp), weight.norm_max,
                                                  hip.ptr(x), D, M, N, K, hip.ptr(r.x), r.x.stride(0), hip.ptr(r.bound), hip.ptr(img),
                                                  hip.ptr(exp), hip.ptr(bound), hip.ptr(ssq), hip.stream_ptr())
    hip.check(st, "mevi_gemm_nt_split_residual_stream")
    return ResidualRows(x, img, exp, bound, ssq)


class NormedRows:
    """rmsnorm(x) that has not been computed yet: T5 feeds every T5LayerNorm into exactly ONE projection, and for the few rows
    of the latency path linear() runs norm + GEMM as one kernel (mevi_gemm_nt_rmsnorm_split_*: same bits as the two-kernel
    form).  Anything else asks for `.rows()` -- the SplitRows image mevi_rmsnorm_split_f16 writes."""

    __slots__ = ("x", "weight", "eps", "_rows")

    def __init__(self, x, weight, eps):
        self.x, self.weight, self.eps, self._rows = x, weight, eps, None

    @property
    def shape(self):
        return tuple(self.x.shape)

    @property
    def device(self):
        return self.x.device

    def rows(self):
        if self._rows is None:
            self._rows = _rmsnorm_split(self.x, self.weight, self.eps)
        return self._rows


# Norm + projection as ONE kernel on the latency path (mevi_gemm_nt_rmsnorm_split_*): built, bit-identical to the two-kernel form
# (tests/test_gemm_split_gpu.py), and measured NO faster -- batch-1 tower 1.43 vs 1.40 ms, NCI generate 3.42 vs 3.40 ms: the
# workgroup's own norm (x round trip + one wave per row) serialises in front of its K loop what the separate 3 us kernel + 1 us
# boundary cost.  Off by default; MEVI_FUSE_NORM=1 turns it on.
FUSE_NORM = os.environ.get("MEVI_FUSE_NORM", "0") == "1"


def _rmsnorm_split(x, weight, eps):
    M, D = x.shape
    img, exp, norm = _split_buffers(M, D, x.device)
    st = hip.lib().mevi_rmsnorm_split_f16(hip.ptr(x), x.stride(0), hip.ptr(weight), eps, M, D, hip.ptr(img), hip.ptr(exp),
                                          hip.ptr(norm), hip.stream_ptr())
    hip.check(st, "mevi_rmsnorm_split_f16")
    return SplitRows(img, exp, D, norm)


@hip.on_device
def rmsnorm(x, weight, eps, out=None, for_gemm=False):
    """T5LayerNorm.  for_gemm: the normed rows only feed linear() -- under GEMM_MODE 'split' they are written as a
    SplitRows image directly, or (few rows, width 768) left to the projection's own kernel (NormedRows)."""
    x, M, D, ldx = _rows2d(_f32(x))
    if for_gemm and GEMM_MODE == "split" and FUSE_NORM and D == 768 and 0 < M <= 256 and weight.is_contiguous():
        return NormedRows(x, weight, eps)
    if for_gemm and GEMM_MODE == "split":
        img, exp, norm = _split_buffers(M, D, x.device)
        st = hip.lib().mevi_rmsnorm_split_f16(hip.ptr(x), ldx, hip.ptr(weight), eps, M, D, hip.ptr(img), hip.ptr(exp),
                                              hip.ptr(norm), hip.stream_ptr())
        hip.check(st, "mevi_rmsnorm_split_f16")
        return SplitRows(img, exp, D, norm)
    if out is None:
        out = torch.empty((M, D), dtype=torch.float32, device=x.device)
    st = hip.lib().mevi_rmsnorm_f32(hip.ptr(x), ldx, hip.ptr(weight), eps, M, D, hip.ptr(out), out.stride(0),
                                    hip.stream_ptr())
    hip.check(st, "mevi_rmsnorm_f32")
    return out


@hip.on_device
def add_layernorm(x, y, weight, bias, eps=1e-5, cvec=None, out=None):
    x, M, D, ldx = _rows2d(_f32(x))
    if out is None:
        out = torch.empty((M, D), dtype=torch.float32, device=x.device)
    ldy = 0
    if y is not None:
        assert y.shape == x.shape and y.stride(1) == 1
        ldy = y.stride(0)
    st = hip.lib().mevi_add_layernorm_f32(hip.ptr(x), ldx, hip.ptr(y) if y is not None else None, ldy,
                                          hip.ptr(cvec) if cvec is not None else None, hip.ptr(weight), hip.ptr(bias),
                                          eps, M, D, hip.ptr(out), out.stride(0), hip.stream_ptr())
    hip.check(st, "mevi_add_layernorm_f32")
    return out


@hip.on_device
def gather_rows(table, idx, out=None):
    table, _, D, ldt = _rows2d(_f32(table))
    idx = idx.to(device=table.device, dtype=torch.int64).contiguous().view(-1)
    n = idx.numel()
    if out is None:
        out = torch.empty((n, D), dtype=torch.float32, device=table.device)
    st = hip.lib().mevi_gather_rows_f32(hip.ptr(table), ldt, hip.ptr(idx), n, D, hip.ptr(out), out.stride(0),
                                        hip.stream_ptr())
    hip.check(st, "mevi_gather_rows_f32")
    return out


@hip.on_device
def scatter_rows(src, idx, out):
    """out[idx[r]] = src[r] (distinct rows); returns `out`."""
    src, n, D, lds_ = _rows2d(_f32(src))
    idx = idx.to(device=src.device, dtype=torch.int64).contiguous().view(-1)
    assert idx.numel() == n and out.dim() == 2 and out.shape[1] == D and out.stride(1) == 1
    st = hip.lib().mevi_scatter_rows_f32(hip.ptr(src), lds_, hip.ptr(idx), n, D, hip.ptr(out), out.stride(0),
                                         hip.stream_ptr())
    hip.check(st, "mevi_scatter_rows_f32")
    return out


@hip.on_device
def scale(x, alpha):
    x = _f32(x).contiguous()
    out = torch.empty_like(x)
    st = hip.lib().mevi_scale_f32(hip.ptr(x), alpha, x.numel(), hip.ptr(out), hip.stream_ptr())
    hip.check(st, "mevi_scale_f32")
    return out


@hip.on_device
def attention(q, k, v, heads, out=None, kv_div=1, bias=None, q_pos0=0, key_mask=None, causal=False, scale=1.0,
              kv_off=None, kv_longest=0, split_bound=None):
    """q [nb, tq, H*dh], k/v [nb/kv_div, tk, H*dh] (any batch/token strides, last dim contiguous) -- or, with
    kv_off (i64 [nb/kv_div + 1], device), PACKED k/v [rows, H*dh]: kv batch c owns rows kv_off[c] .. kv_off[c+1]-1, all
    real, kv_longest = the longest of them.  split_bound (ctx_bound): return the context as the SplitRows image
    [nb * tq, H*dh] of the o-projection instead of f32 [nb, tq, H*dh]."""
    assert q.dim() == 3 and q.stride(2) == 1 and q.is_cuda and q.dtype == torch.float32
    nb, tq, hd = q.shape
    dh = hd // heads
    if kv_off is None:
        for t in (k, v):
            assert t.dim() == 3 and t.stride(2) == 1 and t.is_cuda and t.dtype == torch.float32
        tk = k.shape[1]
        assert k.shape[0] * kv_div == nb and v.shape[:2] == k.shape[:2]
        kst, vst = (k.stride(0), k.stride(1)), (v.stride(0), v.stride(1))
    else:
        for t in (k, v):
            assert t.dim() == 2 and t.stride(1) == 1 and t.is_cuda and t.dtype == torch.float32
        assert kv_off.dtype == torch.int64 and kv_off.is_cuda and kv_off.numel() * kv_div == nb + kv_div and key_mask is None
        tk = int(kv_longest)
        kst, vst = (0, k.stride(0)), (0, v.stride(0))
    brows = bld = 0
    if bias is not None:
        assert bias.dim() == 3 and bias.is_contiguous() and bias.shape[0] == heads
        brows, bld = bias.shape[1], bias.shape[2]
    if key_mask is not None:
        key_mask = key_mask.to(device=q.device, dtype=torch.int64).contiguous()
        assert key_mask.shape == (nb // kv_div, tk)
    if split_bound is not None:
        assert out is None
        ctx, kp, e = _ctx_image(nb * tq, hd, split_bound, q.device)
        st = hip.lib().mevi_attention_split_f16(
            hip.ptr(q), q.stride(0), q.stride(1), hip.ptr(k), kst[0], kst[1], hip.ptr(v), vst[0], vst[1],
            hip.ptr(ctx.img), kp, e, tq * 2 * kp, 2 * kp, nb, tq, tk, heads, dh, kv_div,
            hip.ptr(bias) if bias is not None else None, brows, bld, q_pos0,
            hip.ptr(key_mask) if key_mask is not None else None, 1 if causal else 0, scale,
            hip.ptr(kv_off) if kv_off is not None else None, hip.stream_ptr())
        hip.check(st, "mevi_attention_split_f16")
        return ctx
    if out is None:
        out = torch.empty((nb, tq, hd), dtype=torch.float32, device=q.device)
    st = hip.lib().mevi_attention_f32(
        hip.ptr(q), q.stride(0), q.stride(1), hip.ptr(k), kst[0], kst[1], hip.ptr(v), vst[0], vst[1],
        hip.ptr(out), out.stride(0), out.stride(1), nb, tq, tk, heads, dh, kv_div,
        hip.ptr(bias) if bias is not None else None, brows, bld, q_pos0,
        hip.ptr(key_mask) if key_mask is not None else None, 1 if causal else 0, scale,
        hip.ptr(kv_off) if kv_off is not None else None, hip.stream_ptr())
    hip.check(st, "mevi_attention_f32")
    return out


@hip.on_device
def attention_cached(q, k, v, key_rows, heads, bias=None, q_pos0=0, causal=True, scale=1.0, out=None, split_bound=None):
    """One decode step over ancestor-indexed caches: q [n, H*dh]; k / v [rows, T, H*dh] views of the caches (any row / token
    strides); key_rows i32 [n, tk] (tk <= 8): position j of row b is cache row key_rows[b, j].  Same bits as `attention` on the
    caches physically re-ordered by the beams' parents (generation_utils.py:927-934)."""
    assert q.dim() == 2 and q.stride(1) == 1 and q.is_cuda and q.dtype == torch.float32
    for t in (k, v):
        assert t.dim() == 3 and t.stride(2) == 1 and t.is_cuda and t.dtype == torch.float32
    n, hd = q.shape
    assert key_rows.dtype == torch.int32 and key_rows.is_cuda and key_rows.is_contiguous() and key_rows.shape[0] == n
    tk = key_rows.shape[1]
    brows = bld = 0
    if bias is not None:
        assert bias.dim() == 3 and bias.is_contiguous() and bias.shape[0] == heads
        brows, bld = bias.shape[1], bias.shape[2]
    if split_bound is not None:      # the context as the o-projection's SplitRows image (see attention)
        assert out is None
        ctx, kp, e = _ctx_image(n, hd, split_bound, q.device)
        st = hip.lib().mevi_attention_cached_split_f16(
            hip.ptr(q), q.stride(0), hip.ptr(k), k.stride(0), k.stride(1), hip.ptr(v), v.stride(0), v.stride(1),
            hip.ptr(ctx.img), kp, e, 2 * kp, n, tk, heads, hd // heads, hip.ptr(key_rows),
            hip.ptr(bias) if bias is not None else None, brows, bld, q_pos0, 1 if causal else 0, scale, hip.stream_ptr())
        hip.check(st, "mevi_attention_cached_split_f16")
        return ctx
    if out is None:
        out = torch.empty((n, hd), dtype=torch.float32, device=q.device)
    st = hip.lib().mevi_attention_cached_f32(
        hip.ptr(q), q.stride(0), hip.ptr(k), k.stride(0), k.stride(1), hip.ptr(v), v.stride(0), v.stride(1), hip.ptr(out),
        out.stride(0), n, tk, heads, hd // heads, hip.ptr(key_rows), hip.ptr(bias) if bias is not None else None, brows, bld,
        q_pos0, 1 if causal else 0, scale, hip.stream_ptr())
    hip.check(st, "mevi_attention_cached_f32")
    return out


@hip.on_device
def attention_varlen(q, k, v, seq_off, max_len, heads, bias=None, causal=False, scale=1.0, out=None, split_bound=None):
    """Self-attention over packed sequences: q / k / v [T, H*dh] row-strided views, sequence b = rows
    seq_off[b] .. seq_off[b+1]-1 (i64 [nseq+1] on the device), max_len = longest sequence (<= 256)."""
    for t in (q, k, v):
        assert t.dim() == 2 and t.stride(1) == 1 and t.is_cuda and t.dtype == torch.float32
    T, hd = q.shape
    assert seq_off.dtype == torch.int64 and seq_off.is_cuda and seq_off.is_contiguous()
    brows = bld = 0
    if bias is not None:
        assert bias.dim() == 3 and bias.is_contiguous() and bias.shape[0] == heads
        brows, bld = bias.shape[1], bias.shape[2]
    if split_bound is not None:      # the context as the o-projection's SplitRows image (see attention)
        assert out is None
        ctx, kp, e = _ctx_image(T, hd, split_bound, q.device)
        st = hip.lib().mevi_attention_varlen_split_f16(
            hip.ptr(q), q.stride(0), hip.ptr(k), k.stride(0), hip.ptr(v), v.stride(0), hip.ptr(ctx.img), kp, e, 2 * kp,
            hip.ptr(seq_off), seq_off.numel() - 1, int(max_len), heads, hd // heads,
            hip.ptr(bias) if bias is not None else None, brows, bld, 1 if causal else 0, scale, hip.stream_ptr())
        hip.check(st, "mevi_attention_varlen_split_f16")
        return ctx
    if out is None:
        out = torch.empty((T, hd), dtype=torch.float32, device=q.device)
    st = hip.lib().mevi_attention_varlen_f32(
        hip.ptr(q), q.stride(0), hip.ptr(k), k.stride(0), hip.ptr(v), v.stride(0), hip.ptr(out), out.stride(0),
        hip.ptr(seq_off), seq_off.numel() - 1, int(max_len), heads, hd // heads,
        hip.ptr(bias) if bias is not None else None, brows, bld, 1 if causal else 0, scale, hip.stream_ptr())
    hip.check(st, "mevi_attention_varlen_f32")
    return out


@hip.on_device
def adaptive_logits(s, t, e, t_index=None):
    """logits[row, c] = sum_d s[row, d] * (t[trow, c*dim + d] + e[c, d]), trow = row or t_index[row]."""
    s, rows, dim, lds = _rows2d(_f32(s))
    ncol = e.shape[0]
    assert t.dim() == 2 and t.shape[1] == ncol * dim and t.stride(1) == 1 and e.is_contiguous()
    if t_index is None:
        assert t.shape[0] == rows
    else:
        assert t_index.dtype == torch.int64 and t_index.is_cuda and t_index.is_contiguous() and t_index.numel() == rows
    out = torch.empty((rows, ncol), dtype=torch.float32, device=s.device)
    st = hip.lib().mevi_adaptive_logits_f32(hip.ptr(s), lds, hip.ptr(t), t.stride(0),
                                            hip.ptr(t_index) if t_index is not None else None, hip.ptr(e), rows, ncol, dim,
                                            hip.ptr(out), hip.stream_ptr())
    hip.check(st, "mevi_adaptive_logits_f32")
    return out


@hip.on_device
def adaptive_logits_rows(s, alpha, te, ncol, t_index=None):
    """logits[row, c] = sum_d (s[row, d] * alpha) * te[trow, c*dim + d], trow = row or t_index[row]: adaptive_logits with
    lm_head's rows already added into `te` (the head GEMM's bias) and the hidden state's scale applied here -- a third of the
    cache traffic of scale() + adaptive_logits(); the same bits as that pair at widths other than 768 (at 768 the kernel sums in the
    fused head's order, mevi_gemm_nt_split_head_f32, so that table rows and fused rows agree)."""
    s, rows, dim, lds = _rows2d(_f32(s))
    assert te.dim() == 2 and te.shape[1] == ncol * dim and te.stride(1) == 1 and te.dtype == torch.float32
    if t_index is None:
        assert te.shape[0] == rows
    else:
        assert t_index.dtype == torch.int64 and t_index.is_cuda and t_index.is_contiguous() and t_index.numel() == rows
    out = torch.empty((rows, ncol), dtype=torch.float32, device=s.device)
    st = hip.lib().mevi_adaptive_logits_rows_f32(hip.ptr(s), lds, float(alpha), hip.ptr(te), te.stride(0),
                                                 hip.ptr(t_index) if t_index is not None else None, rows, ncol, dim,
                                                 hip.ptr(out), hip.stream_ptr())
    hip.check(st, "mevi_adaptive_logits_rows_f32")
    return out


FUSED_HEAD = os.environ.get("MEVI_HEAD_FUSED", "1") != "0"     # A/B switch: head GEMM + logits as two kernels (same bits)


@hip.on_device
def head_logits(a, weight, bias, s, alpha, ncol):
    """The PAWA head of rows without a table: logits[m, c] = sum_d (s[m, d] * alpha) * (a[m] . weight[c*dim + d] + bias[c*dim + d]).
    With split weights at dim 768 and more rows than the latency kernels take: ONE GEMM whose epilogue multiplies the head
    matrices with the hidden states instead of storing them (mevi_gemm_nt_split_head_f32) + the 12-partial finish; otherwise
    linear(bias=) + adaptive_logits_rows -- the same bits either way."""
    s2, M, dim, lds = _rows2d(_f32(s))
    N = weight.shape[0]
    assert N == ncol * dim
    L = hip.lib()
    if (FUSED_HEAD and isinstance(weight, SplitRows) and bias is not None and
            L.mevi_gemm_nt_split_head_supported(M, N, weight.shape[1], dim) and s2.data_ptr() % 16 == 0 and lds % 4 == 0):
        xs = a if isinstance(a, SplitRows) else split_rows(a)
        assert xs.shape == (M, weight.shape[1])
        dev = weight.device
        part = torch.empty((N // 256, M, 4), dtype=torch.float32, device=dev)
        st = L.mevi_gemm_nt_split_head_f32(hip.ptr(xs.img), hip.ptr(xs.exp), hip.ptr(weight.img), hip.ptr(weight.exp), M, N,
                                           weight.shape[1], hip.ptr(bias), hip.ptr(s2), lds, float(alpha), dim, hip.ptr(part),
                                           hip.stream_ptr())
        hip.check(st, "mevi_gemm_nt_split_head_f32")
        out = torch.empty((M, ncol), dtype=torch.float32, device=dev)
        st = L.mevi_logits_finish_f32(hip.ptr(part), M, ncol, hip.ptr(out), hip.stream_ptr())
        hip.check(st, "mevi_logits_finish_f32")
        return out
    return adaptive_logits_rows(s2, alpha, linear(a, weight, bias=bias), ncol)


@hip.on_device
def beam_step(logits, beam_scores, K, R, final_step=False):
    """logits [nq*nb, K+1] (col 0 eos), beam_scores [nq, nb] -> (scores, parent, code) [nq, R],
    or final scores [nq, nb] when final_step."""
    logits = _f32(logits).contiguous()
    beam_scores = _f32(beam_scores).contiguous()
    nq, nb = beam_scores.shape
    assert logits.shape == (nq * nb, K + 1)
    dev = logits.device
    if final_step:
        out = torch.empty((nq, nb), dtype=torch.float32, device=dev)
        st = hip.lib().mevi_beam_step_f32(hip.ptr(logits), hip.ptr(beam_scores), nq, nb, K, R, 1, hip.ptr(out),
                                          None, None, hip.stream_ptr())
        hip.check(st, "mevi_beam_step_f32")
        return out
    sc = torch.empty((nq, R), dtype=torch.float32, device=dev)
    parent = torch.empty((nq, R), dtype=torch.int32, device=dev)
    code = torch.empty((nq, R), dtype=torch.int32, device=dev)
    st = hip.lib().mevi_beam_step_f32(hip.ptr(logits), hip.ptr(beam_scores), nq, nb, K, R, 0, hip.ptr(sc),
                                      hip.ptr(parent), hip.ptr(code), hip.stream_ptr())
    hip.check(st, "mevi_beam_step_f32")
    return sc, parent, code


@hip.on_device
def beam_step_tree(logits, beam_scores, K, R, node, tree_mask, tree_base):
    """The beam step under a generic prefix tree (mevi_beam_step_tree_f32): logits [nq*nb, K+1], beam_scores [nq, nb],
    node i32 [nq, nb] (trie node of every beam at this level), tree_mask i32/u32 [n_nodes, ceil(K/32)], tree_base i32
    [n_nodes] -> (scores, parent, code, child node) [nq, R]."""
    logits = _f32(logits).contiguous()
    beam_scores = _f32(beam_scores).contiguous()
    nq, nb = beam_scores.shape
    assert logits.shape == (nq * nb, K + 1) and node.shape == (nq, nb) and node.dtype == torch.int32
    assert tree_mask.dtype == torch.int32 and tree_base.dtype == torch.int32 and tree_mask.shape == (tree_base.numel(), (K + 31) // 32)
    dev = logits.device
    sc = torch.empty((nq, R), dtype=torch.float32, device=dev)
    parent, code, child = (torch.empty((nq, R), dtype=torch.int32, device=dev) for _ in range(3))
    st = hip.lib().mevi_beam_step_tree_f32(hip.ptr(logits), hip.ptr(beam_scores), nq, nb, K, R, hip.ptr(node.contiguous()),
                                           hip.ptr(tree_mask.contiguous()), hip.ptr(tree_base.contiguous()), tree_base.numel(),
                                           hip.ptr(sc), hip.ptr(parent), hip.ptr(code), hip.ptr(child), hip.stream_ptr())
    hip.check(st, "mevi_beam_step_tree_f32")
    return sc, parent, code, child


@hip.on_device
def row_softmax(x, log=True, scale=None):
    """Row-wise log-softmax (log=True) or scale[row] * softmax of x f32 [rows, cols], with the beam step's arithmetic
    (mevi_row_softmax_f32): for the branches that keep every candidate (_generate_all, pq.beam_search below R candidates)."""
    x = _f32(x).contiguous()
    rows, cols = x.shape
    out = torch.empty_like(x)
    sc = None if scale is None else _f32(scale).contiguous().view(-1)
    assert sc is None or sc.numel() == rows
    st = hip.lib().mevi_row_softmax_f32(hip.ptr(x), rows, cols, 0 if log else 1, None if sc is None else hip.ptr(sc), hip.ptr(out),
                                        hip.stream_ptr())
    hip.check(st, "mevi_row_softmax_f32")
    return out


@hip.on_device
def pair_dot(a, ia, b, ib):
    a, _, dim, lda = _rows2d(_f32(a))
    b, _, dim2, ldb = _rows2d(_f32(b))
    assert dim == dim2
    ia = ia.to(device=a.device, dtype=torch.int64).contiguous()
    ib = ib.to(device=a.device, dtype=torch.int64).contiguous()
    n = ia.numel()
    assert ib.numel() == n
    out = torch.empty((n,), dtype=torch.float32, device=a.device)
    st = hip.lib().mevi_pair_dot_f32(hip.ptr(a), lda, hip.ptr(ia), hip.ptr(b), ldb, hip.ptr(ib), n, dim,
                                     hip.ptr(out), hip.stream_ptr())
    hip.check(st, "mevi_pair_dot_f32")
    return out


@hip.on_device
def segment_sort_desc(scores, ids, seg_offsets, max_seg_len):
    scores = _f32(scores).contiguous()
    ids = ids.to(device=scores.device, dtype=torch.int64).contiguous()
    seg = seg_offsets.to(device=scores.device, dtype=torch.int64).contiguous()
    out_s = torch.empty_like(scores)
    out_i = torch.empty_like(ids)
    st = hip.lib().mevi_segment_sort_desc_f32(hip.ptr(scores), hip.ptr(ids), hip.ptr(seg), seg.numel() - 1,
                                              int(max_seg_len), hip.ptr(out_s), hip.ptr(out_i), hip.stream_ptr())
    hip.check(st, "mevi_segment_sort_desc_f32")
    return out_s, out_i


@hip.on_device
def segment_aggregate_sort(scores, ids, seg_offsets, max_seg_len, mode):
    """Per segment: entries of one id merged ('add': sequential f32 adds from 0 in list order, 'max'), unique entries
    sorted by (score desc, id asc) at the segment's start; returns (scores, ids, counts i32 [nseg])."""
    scores = _f32(scores).contiguous()
    ids = ids.to(device=scores.device, dtype=torch.int64).contiguous()
    seg = seg_offsets.to(device=scores.device, dtype=torch.int64).contiguous()
    out_s = torch.empty_like(scores)
    out_i = torch.empty_like(ids)
    counts = torch.empty(seg.numel() - 1, dtype=torch.int32, device=scores.device)
    st = hip.lib().mevi_segment_aggregate_sort_f32(hip.ptr(scores), hip.ptr(ids), hip.ptr(seg), seg.numel() - 1, int(max_seg_len),
                                                   {"add": 0, "max": 1}[mode], hip.ptr(out_s), hip.ptr(out_i), hip.ptr(counts),
                                                   hip.stream_ptr())
    hip.check(st, "mevi_segment_aggregate_sort_f32")
    return out_s, out_i, counts
