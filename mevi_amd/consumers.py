"""evaluate.py / ensemble_marco.py on flat arrays and the device (MEVI/evaluate.py:27-157, MEVI/ensemble_marco.py:28-238).

The reference's consumers eval() every TSV field into dicts of Python lists and then walk them: at MS MARCO size
(6980 queries x 1000 dense + fine entries) that is 14 M numbers and 14 M dict operations per (alpha, beta, gamma) point --
4.6 s here with every list operation already in numpy (profiles/r02_e2e_fullsize.txt), against 0.1 s for the search that
wrote the file.  This module keeps the lists as flat arrays from the parser on (one native pass over the file:
mevi_parse_tsv_columns), does cluster ranks, score combination, ranking and gt look-up on the GPU (csrc/consumers.hip) and
hands per-query first-hit ranks to the SAME accumulate()/report() as the dict path, so the printed bytes cannot differ.

Every entry point returns None when its input is not of the plain shape (bracketed or ragged fields, duplicate queries,
a pickle-only mapping, > 8192 entries per query, ...): the caller then runs the dict path of metrics.py, which restates the
reference line by line and raises what the reference raises.  MEVI_CONSUMERS=host forces that path, =device refuses to
fall back.
"""
import os

import numpy as np

from . import io as mio

MAX_ENTRIES = 8192


def mode():
    m = os.environ.get("MEVI_CONSUMERS", "auto")
    if m not in ("auto", "host", "device"):
        raise ValueError(f"MEVI_CONSUMERS={m!r}: expected auto, host or device")
    return m


def device_ready():
    """True when the array path can run (library built, a GPU visible)."""
    if mode() == "host":
        return False
    try:
        import torch

        from . import hip

        ok = torch.cuda.is_available() and os.path.exists(hip.LIB)
    except Exception:
        ok = False
    if not ok and mode() == "device":
        raise RuntimeError("MEVI_CONSUMERS=device: no GPU or libmevi_hip.so not built")
    return ok


class RankedArrays:
    """One ranked TSV: queries in file order, per-query offsets, flat ids (and scores)."""

    def __init__(self, queries, seg, docs, scores=None):
        self.queries, self.seg, self.docs, self.scores = queries, seg, docs, scores
        self.row = {q: i for i, q in enumerate(queries)}

    def __len__(self):
        return len(self.queries)


PARSE_THREADS = 8       # the 172 MB dense TSV of MS MARCO dev: one native pass took 1.1 s of ensemble_marco.py's 2.0 s (tools/e2e_cli.py)


def _parse_chunk(base, a, b, lines, n_commas, template):
    """Bytes [a, b) of the file (whole lines) through the native parser: (span, seg_i, vals_i, seg_f, vals_f) or None."""
    from . import hip

    ci, cf = template.get("pred"), template.get("score")
    cap = n_commas + lines + 1
    span = np.empty((max(lines, 1), 2), np.int64)
    seg_i, vals_i = np.empty(lines + 1, np.int64), np.empty(cap, np.int64)
    seg_f, vals_f = (np.empty(lines + 1, np.int64), np.empty(cap, np.float64)) if cf is not None else (None, None)
    n = hip.lib().mevi_parse_tsv_columns(base + a, b - a, template["query"], ci, -1 if cf is None else cf, span.ctypes.data,
                                         seg_i.ctypes.data, vals_i.ctypes.data, cap,
                                         None if cf is None else seg_f.ctypes.data, None if cf is None else vals_f.ctypes.data,
                                         cap, lines)
    if n != lines:
        return None
    if cf is not None and not np.array_equal(seg_i, seg_f):      # a line whose ids and scores differ in number
        return None
    span[:, 0] += a                                              # spans relative to the whole file
    return span[:lines], seg_i, vals_i[:seg_i[-1]], None if cf is None else vals_f[:seg_f[-1]]


def parse_ranked(path, template):
    """`path` parsed natively -> RankedArrays, or None when the file is not of the plain shape (then io.parse_file gives it
    its Python meaning).  Large files are cut at line ends into PARSE_THREADS pieces parsed side by side (the parser is a C
    function: ctypes drops the GIL) and stitched -- the same arrays as one pass.  A duplicate query makes the dict path keep the
    LAST line at the FIRST position; rare enough to leave to it."""
    import ctypes

    if path.endswith(".pkl"):
        return None
    with open(path, "rb") as f:
        buf = f.read()
    ci, cf = template.get("pred"), template.get("score")
    if ci is None or ci < 0 or template["query"] < 0 or (cf is not None and cf < 0):
        return None
    keep = ctypes.c_char_p(buf)                                  # points into `buf` (no copy); both stay alive to the end
    base = ctypes.cast(keep, ctypes.c_void_p).value or 0
    cuts = [0]
    nthreads = PARSE_THREADS if len(buf) >= (8 << 20) else 1
    for j in range(1, nthreads):
        at = buf.find(b"\n", len(buf) * j // nthreads)
        if at < 0:
            break
        if at + 1 > cuts[-1]:
            cuts.append(at + 1)
    cuts.append(len(buf))
    cuts = sorted(set(cuts))

    def work(j):
        a, b = cuts[j], cuts[j + 1]
        lines = buf.count(b"\n", a, b) + (0 if b == a or buf[b - 1:b] == b"\n" else 1)
        return _parse_chunk(base, a, b, lines, buf.count(b",", a, b), template)

    if len(cuts) > 2:
        from concurrent.futures import ThreadPoolExecutor

        with ThreadPoolExecutor(len(cuts) - 1) as pool:
            parts = list(pool.map(work, range(len(cuts) - 1)))
    else:
        parts = [work(0)] if buf else [_parse_chunk(base, 0, 0, 0, 0, template)]
    if any(p is None for p in parts):
        return None
    span = np.concatenate([p[0] for p in parts])
    seg = [np.zeros(1, np.int64)]
    total = 0
    for p in parts:
        seg.append(p[1][1:] + total)
        total += int(p[1][-1])
    seg_i = np.concatenate(seg)
    vals_i = np.concatenate([p[2] for p in parts])
    vals_f = None if cf is None else np.concatenate([p[3] for p in parts])
    try:
        queries = [buf[a:a + b].decode("utf-8") for a, b in span.tolist()]
    except UnicodeDecodeError:
        return None
    if len(set(queries)) != len(queries):
        return None
    del keep
    return RankedArrays(queries, seg_i, vals_i, vals_f)


# ---- device primitives ------------------------------------------------------------------------------------------------
def _dev():
    import torch

    return torch.device("cuda", torch.cuda.current_device())


def _t(a, dtype):
    import torch

    if torch.is_tensor(a):
        return a.to(device=_dev(), dtype=dtype).contiguous()
    a = np.ascontiguousarray(a)
    if not a.flags.writeable:                 # a read-only memory map (the mapping's array form): torch wants to own writable memory
        a = a.copy()
    return torch.from_numpy(a).to(device=_dev(), dtype=dtype)


def cluster_ranks(codes_t, docs_t, seg_t, beam_t, n_clusters):
    """i32 ranks [total] of every dense entry's cluster among its query's beam clusters; KeyError(doc id) as the dict path."""
    import torch

    from . import hip

    nq, R, M = beam_t.shape
    out = torch.empty(docs_t.numel(), dtype=torch.int32, device=docs_t.device)
    if docs_t.numel() == 0:                   # every dense list empty: nothing to rank (and no device pointers to hand over)
        return out
    bad = torch.empty(1, dtype=torch.int64, device=docs_t.device)
    st = hip.lib().mevi_cluster_ranks_i32(hip.ptr(codes_t), codes_t.shape[0], M, hip.ptr(docs_t), hip.ptr(seg_t), nq,
                                          hip.ptr(beam_t), R, int(n_clusters), hip.ptr(out), hip.ptr(bad), hip.stream_ptr())
    hip.check(st, "mevi_cluster_ranks_i32")
    b = int(bad.item())
    if b != -1:
        raise KeyError(int(docs_t[b].item()))
    return out


def ensemble_rank(seg_d, docs_d, sc_d, cr_d, fine, n_clusters, alpha, beta, gamma, max_entries, out_seg, n_ranks=None):
    """Ranked ids of every query's ensemble: (out_docs i64[out_seg[-1]], out_n i32[nq]) on the device, or None when the
    kernel declined.  fine = (fine_row, seg_f, docs_f, sc_f) or None."""
    import torch

    from . import hip

    nq = seg_d.numel() - 1
    # cluster_ranks_kernel returns the LAST matching beam index (0..R-1) or n_clusters; with repeated beam clusters
    # R - 1 > n_clusters, so the table spans both.  A rank whose denominator is zero raises only if some entry has it
    # (the reference divides per entry, ensemble_marco.py:203).
    n_terms = max(int(n_ranks or 0), int(n_clusters) + 1)
    zero_den = [c for c in range(n_terms) if beta * c + 1 == 0]
    if zero_den and bool(torch.isin(cr_d, torch.tensor(zero_den, dtype=cr_d.dtype, device=cr_d.device)).any()):
        raise ZeroDivisionError("float division by zero")
    term = torch.tensor([float("nan") if beta * c + 1 == 0 else alpha / (beta * c + 1) for c in range(n_terms)],
                        dtype=torch.float64).to(seg_d.device)
    punish = 1 - gamma * alpha
    total = int(out_seg[-1].item()) if nq else 0
    out_docs = torch.empty(total, dtype=torch.int64, device=seg_d.device)
    out_n = torch.empty(nq, dtype=torch.int32, device=seg_d.device)
    err = torch.empty(1, dtype=torch.int32, device=seg_d.device)
    fr, sf, df, scf = fine if fine is not None else (None, None, None, None)
    # an EMPTY tensor has no device pointer, and the C ABI refuses null list pointers (it cannot see that every list is empty):
    # hand it one unused element instead (found by tools/stress_consumers.py: a file whose dense -- or fine -- lists are all empty)
    some = lambda x: x if x is None or x.numel() else torch.zeros(1, dtype=x.dtype, device=x.device)      # noqa: E731
    docs_d, sc_d, cr_d, out_docs_arg, df, scf = some(docs_d), some(sc_d), some(cr_d), some(out_docs), some(df), some(scf)
    p = lambda x: None if x is None else hip.ptr(x)          # noqa: E731
    st = hip.lib().mevi_ensemble_rank_f64(hip.ptr(seg_d), hip.ptr(docs_d), hip.ptr(sc_d), hip.ptr(cr_d), p(fr), p(sf), p(df),
                                          p(scf), nq, int(max_entries), int(n_clusters), hip.ptr(term), float(punish),
                                          hip.ptr(out_seg), hip.ptr(out_docs_arg), hip.ptr(out_n), hip.ptr(err), hip.stream_ptr())
    hip.check(st, "mevi_ensemble_rank_f64")
    if int(err.item()) != 0:
        return None
    return out_docs, out_n


def first_hits(lists_t, seg_t, list_n_t, pair_row, pair_doc):
    """np.int32 rank of every (list row, doc) pair, -1 when absent."""
    import torch

    from . import hip

    n = len(pair_row)
    out = torch.empty(n, dtype=torch.int32, device=seg_t.device)
    if n:
        pr, pd = _t(np.asarray(pair_row, np.int64), torch.int64), _t(np.asarray(pair_doc, np.int64), torch.int64)
        if lists_t.numel() == 0:              # every list empty: one unused element stands in for the (null) list pointer
            lists_t = torch.zeros(1, dtype=lists_t.dtype, device=lists_t.device)
        st = hip.lib().mevi_first_hits_i64(hip.ptr(lists_t), hip.ptr(seg_t), None if list_n_t is None else hip.ptr(list_n_t),
                                           hip.ptr(pr), hip.ptr(pd), n, hip.ptr(out), hip.stream_ptr())
        hip.check(st, "mevi_first_hits_i64")
    return out.cpu().numpy()


# ---- evaluate() on arrays ----------------------------------------------------------------------------------------------
def _gt_pairs(gts, row_of, missing_ok):
    """(pair_row, pair_doc, per-query counts) for metrics.evaluate_ranked's walk over gts.items(); None when a gt is not a
    flat list of ints or -- unless missing_ok -- a gt query has no list (the dict path raises KeyError there)."""
    rows, docs, counts = [], [], []
    for q, gt in gts.items():
        r = row_of.get(q, -1)
        if r < 0 and not missing_ok:
            return None
        if not isinstance(gt, list) or not all(type(g) is int and -(1 << 62) < g < (1 << 62) for g in gt):
            return None
        rows.extend([r] * len(gt))
        docs.extend(gt)
        counts.append(len(gt))
    return rows, docs, counts


def evaluate_lists(title, cutoffs, gts, pairs, lists_t, seg_t, list_n_t, ofile=None, stdout_prefix=""):
    """metrics.evaluate_ranked with the per-gt ranks looked up on the device; same accumulation, same printed bytes."""
    from . import metrics

    rows, docs, counts = pairs
    hits = first_hits(lists_t, seg_t, list_n_t, rows, docs).tolist()
    recall = {k: 0 for k in cutoffs}
    mrr = {k: 0 for k in cutoffs}
    at = 0
    for c in counts:
        metrics.accumulate([h if h >= 0 else None for h in hits[at:at + c]], cutoffs, recall, mrr)
        at += c
    n = len(gts)
    metrics.report(title, cutoffs, recall, mrr, n, ofile, stdout_prefix)
    return {k: v / n for k, v in recall.items()}, {k: v / n for k, v in mrr.items()}


def _or_refuse(fn):
    """MEVI_CONSUMERS=device: a declined input is an error instead of a quiet hand-over to the dict path."""
    import functools

    @functools.wraps(fn)
    def run(*a, **kw):
        out = fn(*a, **kw)
        if out is None and mode() == "device":
            raise RuntimeError(f"MEVI_CONSUMERS=device: {fn.__name__} declined this input (not of the plain shape)")
        return out

    return run


@_or_refuse
def evaluate_main(gts, ance_path, cutoffs, ofile):
    """evaluate.py's body after the gt file is read; None -> dict path."""
    import torch

    if not device_ready():
        return None
    dense = parse_ranked(ance_path, {"query": 0, "pred": 2})
    if dense is None:
        return None
    pairs = _gt_pairs(gts, dense.row, missing_ok=False)
    if pairs is None:
        return None
    if ofile is not None:
        open(ofile, "w").close()
    return evaluate_lists("ANCE Pred", cutoffs, gts, pairs, _t(dense.docs, torch.int64), _t(dense.seg, torch.int64), None,
                          ofile, stdout_prefix="Scoring ")


# ---- ensemble_marco.combine_main on arrays ------------------------------------------------------------------------------
class EnsembleInputs:
    """Everything combine_main reads, as device tensors (built from files by ensemble_main, from a search's own results by
    tools/chain_c4.py)."""

    def __init__(self, queries, seg_d, docs_d, sc_d, beam, codes, fine=None):
        import torch

        self.queries = list(queries)
        self.row = {q: i for i, q in enumerate(self.queries)}
        self.seg_d, self.docs_d, self.sc_d = _t(seg_d, torch.int64), _t(docs_d, torch.int64), _t(sc_d, torch.float64)
        self.beam = _t(beam, torch.int32)                       # [nq, R, M]
        self.codes = _t(codes, torch.int32)                     # [N, M]
        self.fine = None
        if fine is not None:
            fine_queries, fine_row, seg_f, docs_f, sc_f = fine
            self.fine_row_of = {q: i for i, q in enumerate(fine_queries)}
            self.fine = (_t(fine_row, torch.int64), _t(seg_f, torch.int64), _t(docs_f, torch.int64), _t(sc_f, torch.float64))
        nd = np.diff(np.asarray(seg_d.cpu() if torch.is_tensor(seg_d) else seg_d, dtype=np.int64))
        if fine is not None:
            sf = np.asarray(seg_f.cpu() if torch.is_tensor(seg_f) else seg_f, dtype=np.int64)
            fr = np.asarray(fine_row.cpu() if torch.is_tensor(fine_row) else fine_row, dtype=np.int64)
            nf = sf[fr + 1] - sf[fr]
            n = np.minimum(nd + nf, 2 * nd)
        else:
            n = nd
        self.max_entries = int(n.max()) if len(n) else 0
        self.out_seg = _t(np.concatenate([[0], np.cumsum(n)]).astype(np.int64), torch.int64)
        # number of distinct beam clusters: every query must agree (ensemble_marco.py:185-187)
        b = self.beam.cpu().numpy()
        distinct = {len({tuple(c) for c in q}) for q in b.tolist()}
        if len(distinct) > 1:
            raise AssertionError("queries disagree on the number of beam clusters")
        self.n_clusters = distinct.pop() if distinct else None

    def ranks(self):
        return cluster_ranks(self.codes, self.docs_d, self.seg_d, self.beam, self.n_clusters)

    def ensemble(self, cr, alpha, beta, gamma):
        return ensemble_rank(self.seg_d, self.docs_d, self.sc_d, cr, self.fine, self.n_clusters, alpha, beta, gamma,
                             self.max_entries, self.out_seg, n_ranks=self.beam.shape[1])


@_or_refuse
def ensemble_main(gts, ance_path, fine_path, clusters, mapping, alphas, betas, gammas, cutoffs, ofile):
    """ensemble_marco.combine_main after the small files are read (gts, clusters: dicts; mapping: metrics.ArrayMapping).
    Returns the results dict, or None before anything is printed or written when the dict path must run."""
    import torch

    from . import metrics

    if not device_ready() or not isinstance(mapping, metrics.ArrayMapping):
        return None
    dense = parse_ranked(ance_path, metrics.RANKED_TEMPLATE)
    if dense is None or len(dense) == 0:
        return None
    fine = None
    if fine_path is not None:
        fine = parse_ranked(fine_path, metrics.RANKED_TEMPLATE)
        if fine is None or any(q not in fine.row for q in dense.queries):
            return None
    try:
        beam = np.asarray([clusters[q] for q in dense.queries])
    except (KeyError, ValueError):
        return None
    codes = np.asarray(mapping.codes)
    if beam.ndim != 3 or beam.dtype.kind not in "iu" or beam.shape[1] == 0 or beam.shape[2] != codes.shape[1] \
            or np.abs(beam).max(initial=0) >= 1 << 31:
        return None
    pairs_d = _gt_pairs(gts, dense.row, missing_ok=False)
    pairs_f = _gt_pairs(gts, fine.row, missing_ok=False) if fine is not None else True
    pairs_e = _gt_pairs(gts, dense.row, missing_ok=True)
    if pairs_d is None or pairs_f is None or pairs_e is None:
        return None
    fine_in = None
    if fine is not None:
        fine_in = (fine.queries, np.asarray([fine.row[q] for q in dense.queries], np.int64), fine.seg, fine.docs, fine.scores)
    inp = EnsembleInputs(dense.queries, dense.seg, dense.docs, dense.scores, beam, codes, fine_in)
    if inp.max_entries > MAX_ENTRIES:
        return None
    cr = inp.ranks()                                           # KeyError for an id without a code row, as the dict path
    if inp.ensemble(cr, 0.0, 0.0, 0.0) is None:                # ids outside the kernel's range: decided before any output
        return None
    if ofile is not None:
        open(ofile, "w").close()
    results = {"ANCE Pred": evaluate_lists("ANCE Pred", cutoffs, gts, pairs_d, inp.docs_d, inp.seg_d, None, ofile)}
    if fine is not None:
        results["Fine Pred"] = evaluate_lists("Fine Pred", cutoffs, gts, pairs_f, _t(fine.docs, torch.int64),
                                              _t(fine.seg, torch.int64), None, ofile)
    for a in alphas:
        for b in betas:
            for g in gammas:
                out_docs, out_n = inp.ensemble(cr, a, b, g)
                title = f"score + {a} / ({b} * crank + 1); punishment (1 - {g} * {a})"
                results[title] = evaluate_lists(title, cutoffs, gts, pairs_e, out_docs, inp.out_seg, out_n, ofile)
    return results
