"""Comparison of ranked lists with the reference's own outputs (goldens G9).

The reference leaves two things unspecified: the order of EXACTLY equal scores (torch.sort / torch.topk /
faiss's heap) and the f32 summation order of a dot product (BLAS).  `same_ranking` therefore compares
  * the score sequences -- bit for bit when `tol == 0` (integer-valued inputs: every order of summation gives the
    same bits), else element-wise within `tol`;
  * the ids as multisets inside every RUN of positions whose neighbouring reference scores are no further apart
    than 2*tol (tol == 0: runs of exactly equal scores) -- a swap can only happen inside such a run; positions
    outside runs must hold the identical document;
  * for a truncated list (top-k), members of the last run may be any rows scoring inside that run's range:
    `full_scores` (exact scores of every row) decides membership there.
Returns the fraction of positions that had to match one to one.
"""
import numpy as np


def same_ranking(got_s, got_i, ref_s, ref_i, tol=0.0, full_scores=None):
    got_s, ref_s = np.asarray(got_s, np.float32), np.asarray(ref_s, np.float32)
    got_i, ref_i = np.asarray(got_i, np.int64), np.asarray(ref_i, np.int64)
    assert got_s.shape == ref_s.shape and got_i.shape == ref_i.shape, (got_s.shape, ref_s.shape)
    n = len(ref_s)
    if n == 0:
        return 1.0
    if tol == 0.0:
        assert np.array_equal(got_s.view(np.uint32), ref_s.view(np.uint32)), "scores differ in bits"
    else:
        assert np.abs(got_s - ref_s).max() <= tol, float(np.abs(got_s - ref_s).max())
    assert np.all(got_s[:-1] >= got_s[1:]), "list not in descending order"
    joined = (ref_s[:-1] - ref_s[1:]) <= 2 * tol                   # position p and p+1 belong to one run
    single = 0
    start = 0
    while start < n:
        end = start + 1
        while end < n and joined[end - 1]:
            end += 1
        if end == n and full_scores is not None:
            lo, hi = ref_s[n - 1] - 2 * tol, ref_s[start] + 2 * tol
            fs = full_scores[got_i[start:end]]
            assert np.all((fs >= lo) & (fs <= hi)), "row outside the last run's score range"
            assert len(np.unique(got_i[start:end])) == len(np.unique(ref_i[start:end]))
        else:
            assert np.array_equal(np.sort(got_i[start:end]), np.sort(ref_i[start:end])), \
                f"run [{start},{end}) holds different documents"
            if end - start == 1 or len(np.unique(ref_i[start:end])) == 1:    # one document (possibly listed twice)
                single += end - start
        start = end
    return single / n
