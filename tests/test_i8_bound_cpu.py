"""CPU restatement (numpy, float64) of the 8-bit image's quantisation and of the UPPER-BOUND keys its filter ranks rows by
(csrc/ip_topk.hip: split_docs_i8_kernel, split_queries_i8_kernel, ip_filter_i8_small_kernel; DESIGN 4.1c'): the claim the proof of
mevi_ip_topk_indexed8_f32 rests on is checked here on inputs chosen to strain it -- for EVERY (query, row):
    exact q.d  <=  t_q * key + q.mu + eps,      key = fl((fl(N) + G_q) * s_r),   G_q = (rho ||w|| + eta ||w - w^||) / t_q
-- and the host-side pieces of the entry points (sizes, eligibility) without a GPU."""
import numpy as np
import pytest

F32 = np.float32


def quantise_docs(d, mu, c):
    y = (d.astype(np.float64) - mu.astype(np.float64)) / c.astype(np.float64)
    y32 = ((d - mu) / c).astype(F32)                                   # the kernel's f32 form decides the rounding
    m = np.abs(y32).max(1)
    s = (m / F32(127.0)).astype(F32)
    inv = np.where(m > 0, F32(127.0) / np.where(m > 0, m, 1), 0).astype(F32)
    integer = np.clip(np.rint(y32 * inv[:, None]), -127, 127).astype(np.int64)
    yhat = s[:, None].astype(np.float64) * integer
    err = np.sqrt(((y - yhat) ** 2).sum(1))
    ylen = np.sqrt((yhat ** 2).sum(1))
    pos = s > 0
    rho = float((err[pos] / s[pos]).max() * 1.0001) if pos.any() else 0.0
    eta = float((ylen[pos] / s[pos]).max() * 1.0001) if pos.any() else 0.0
    return integer, s, rho, eta, float(ylen.max() * 1.0001), float(s.max())


def quantise_query(q, c):
    w = q.astype(np.float64) * c.astype(np.float64)
    t = F32(np.abs((q * c).astype(F32)).max() / F32(16256.0))
    if not (1e-37 < t < 1e37):
        t = F32(1.0)
    big = np.clip(np.rint(w / np.float64(t)), -16256, 16256).astype(np.int64)
    lo = ((big + 64) & 127) - 64
    hi = (big - lo) >> 7
    assert (np.abs(hi) <= 127).all() and (lo >= -64).all() and (lo <= 63).all() and (128 * hi + lo == big).all()
    dq = np.sqrt(((w - np.float64(t) * big) ** 2).sum())
    return hi, lo, t, float(np.sqrt((w ** 2).sum())), float(dq)


def upper_keys(d, q, scale_rows=None):
    mu = d.mean(0, dtype=np.float64).astype(F32)
    c = np.sqrt(((d.astype(np.float64) - mu) ** 2).mean(0)).astype(F32)
    c = np.where((c > 1e-18) & (c < 1e18), c, F32(1.0)).astype(F32)
    integer, s, rho, eta, ymax, smax = quantise_docs(d, mu, c)
    out = []
    for qi in q:
        hi, lo, t, wlen, dq = quantise_query(qi, c)
        n_hi, n_lo = integer @ hi, integer @ lo                              # the matrix cores' exact int32 sums
        assert np.abs(n_hi).max() < 2 ** 24 and np.abs(n_lo).max() < 2 ** 24
        g = F32((rho * wlen + eta * dq) / np.float64(t) * 1.0001)
        n_f = (n_hi.astype(F32) * F32(128.0) + n_lo.astype(F32)).astype(F32)   # one rounding (fmaf in the kernel: never more)
        key = ((n_f + g).astype(F32) * s).astype(F32)
        qn16 = (wlen + dq) * 1.0001
        ddmax = rho * smax * 1.0001 / 1048576.0
        c_acc = 1.001 / 4194304.0
        eps = qn16 * ddmax + c_acc * qn16 * (ymax + ddmax)
        exact = d.astype(np.float64) @ qi.astype(np.float64)
        ub = np.float64(t) * key.astype(np.float64) + float(qi.astype(np.float64) @ mu.astype(np.float64)) + eps
        out.append((exact, ub, wlen))
    return out


@pytest.mark.parametrize("case", ["gaussian", "common component", "outlier columns and long rows", "sparse", "integers", "tiny and huge"])
def test_upper_bound_keys_dominate_the_exact_scores(case):
    rng = np.random.default_rng(len(case))
    nd, dim, nq = 4000, 256, 6
    d = rng.standard_normal((nd, dim)).astype(F32)
    q = rng.standard_normal((nq, dim)).astype(F32)
    if case == "common component":
        d = (0.05 * d + 0.02).astype(F32)
        q = (d[:nq] + 0.005 * rng.standard_normal((nq, dim))).astype(F32)
    elif case == "outlier columns and long rows":
        d[:, 3] *= 300
        d[:, 200] *= 1e-4
        d[rng.integers(0, nd, 20)] *= 50
        d[rng.integers(0, nd, 20), rng.integers(0, dim, 20)] = 900
    elif case == "sparse":
        d *= rng.random((nd, dim)) < 0.05
        q *= rng.random((nq, dim)) < 0.3
    elif case == "integers":
        d = rng.integers(-3, 4, (nd, dim)).astype(F32)
        q = rng.integers(-2, 3, (nq, dim)).astype(F32)
    elif case == "tiny and huge":
        d *= np.exp(rng.normal(0, 3, (nd, 1))).astype(F32)
        q *= np.exp(rng.normal(0, 3, (nq, 1))).astype(F32)
        d[::9] = 0
        q[1] = 0
    slack = []
    for exact, ub, wlen in upper_keys(d, q):
        assert (ub >= exact).all(), (case, float((exact - ub).max()))
        if wlen > 0:
            slack.append(np.median(ub - exact) / wlen)
    if case == "gaussian":                                # the typical row's term: ~0.2 of the score deviation (= ||w|| on unit-variance y)
        assert 0.1 < np.median(slack) < 0.35, slack


def test_8_bit_entry_points_host_side():
    from mevi_amd import hip

    L = hip.lib()
    nd, dim = 8_841_823, 768
    rows = (nd + 255) // 256 * 256
    assert L.mevi_ip_index8_bytes(nd, dim) >= rows * dim + 2 * rows * 4 + dim * 4
    assert L.mevi_ip_index8_bytes(nd, dim) < 1.02 * (rows * dim + 2 * rows * 4)      # ~ one byte per element + 8 per row
    inner = L.mevi_ip_topk_indexed_workspace_bytes(32, dim, 100)
    assert L.mevi_ip_topk_indexed8_workspace_bytes(32, dim, 100) > inner               # the 8-bit pass's own state in front
    assert L.mevi_ip_topk_indexed8_workspace_bytes(33, dim, 100) == L.mevi_ip_topk_indexed_workspace_bytes(33, dim, 100)
    assert L.mevi_ip_topk_indexed8_workspace_bytes(8, 64, 100) == L.mevi_ip_topk_indexed_workspace_bytes(8, 64, 100)   # padded dim < 256
    assert L.mevi_ip_topk_indexed8_workspace_bytes(8, dim, 1400) == L.mevi_ip_topk_indexed_workspace_bytes(8, dim, 1400)   # 3 k + 64 > 4096
    assert L.mevi_ip_topk_indexed8_workspace_bytes(8, dim, 5000) == 0
    # no device: the entry point refuses bad arguments before touching one
    assert L.mevi_ip_topk_indexed8_f32(None, 4, None, None, None, 10, dim, 0, 0, None, None, None, 0, None) != 0
