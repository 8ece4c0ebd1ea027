"""On-disk formats of the MEVI hot path.

* raw little-endian f32 matrices (`docemb.bin`, `query_emb.bin`): faiss_search.read
  (MEVI/faiss_search.py:9-10)
* dense ranked TSV `query\\t\\tid,id,...\\tscore,score,...`: faiss_search.to_file
  (MEVI/faiss_search.py:71-77)
* per-rank result logs merged by rank 0 (`*_coarse.tsv`, `*_fine.tsv`, `*_hn<N>.tsv`):
  main_models.LogTxtFile (MEVI/main_models.py:244-273)
* readers for the consumers (evaluate.py / ensemble_marco.py parse_file, :91-110)
"""
import ast
import os
import pickle

import numpy as np


def upload_rows(src, device, chunk_rows=1 << 18, threads=8):
    """f32 [rows, dim] host array or memmap -> CUDA tensor through two pinned staging buffers (page-locked copies run
    at PCIe rate and overlap the next chunk's read; a pageable 27 GB `tensor.to(device)` is several times slower).
    The host-side copy into the staging buffer is split over `threads` threads (numpy releases the GIL while copying;
    one thread moves ~10 GB/s, which would otherwise bound the upload)."""
    from concurrent.futures import ThreadPoolExecutor

    import torch

    rows, dim = src.shape
    out = torch.empty((rows, dim), dtype=torch.float32, device=device)
    if rows == 0:
        return out
    chunk_rows = max(1, min(chunk_rows, rows))
    stage = [torch.empty((chunk_rows, dim), dtype=torch.float32).pin_memory() for _ in range(2)]
    views = [b.numpy() for b in stage]
    done = [torch.cuda.Event() for _ in range(2)]
    stream = torch.cuda.current_stream(device)
    threads = max(1, min(threads, os.cpu_count() or 1))

    # a file-backed memmap (faiss_search.read / map_rows): the bytes go from the page cache STRAIGHT into the pinned buffer
    # with pread (one kernel copy, no page-table population of a 27 GB mapping, no intermediate user-space copy); anything
    # else (an array in memory) is copied
    fd = None
    if isinstance(src, np.memmap) and getattr(src, "filename", None) and src.flags.c_contiguous and src.dtype == np.float32 \
            and os.environ.get("MEVI_UPLOAD", "pread") != "copy":      # MEVI_UPLOAD=copy: memcpy out of the mapping (A/B)
        try:
            fd = os.open(src.filename, os.O_RDONLY)
            base = int(src.offset)
        except OSError:
            fd = None

    def fill(dst, a, b):
        if b <= a:
            return
        if fd is None:
            np.copyto(dst, src[a:b])
            return
        mv = memoryview(dst).cast("B")
        off, got = base + a * dim * 4, 0
        while got < len(mv):
            n = os.preadv(fd, [mv[got:]], off + got)
            if n <= 0:
                raise OSError(f"short read of {src.filename} at byte {off + got}")
            got += n

    with ThreadPoolExecutor(threads) as pool:
        for i, a in enumerate(range(0, rows, chunk_rows)):
            b = min(a + chunk_rows, rows)
            if i >= 2:
                done[i & 1].synchronize()              # the copy that last used this buffer has finished
            n = b - a
            cuts = [a + n * j // threads for j in range(threads + 1)]
            list(pool.map(lambda j: fill(views[i & 1][cuts[j] - a:cuts[j + 1] - a], cuts[j], cuts[j + 1]), range(threads)))
            out[a:b].copy_(stage[i & 1][:n], non_blocking=True)
            done[i & 1].record(stream)
    stream.synchronize()
    if fd is not None:
        os.close(fd)
    return out


def map_rows(path, dim, mode="r", rows=None, first_row=0):
    """np.memmap of rows [first_row, first_row + rows) of a raw f32 [*, dim] file (mode "r"), or a new file of `rows`
    rows (mode "w+").  A zero-row range -- a rank beyond the data when there are fewer rows than ranks, an empty part
    file -- is an empty array (np.memmap refuses to map zero bytes); in "w+" mode the (empty) file is still created."""
    if mode == "r":
        if rows is None:
            rows = os.path.getsize(path) // (4 * dim) - first_row
        if rows <= 0:
            return np.empty((0, dim), np.float32)
        return np.memmap(path, dtype=np.float32, mode="r", offset=first_row * dim * 4, shape=(rows, dim))
    if rows == 0:
        open(path, "wb").close()
        return np.empty((0, dim), np.float32)
    return np.memmap(path, dtype=np.float32, mode=mode, shape=(rows, dim))


def flush_rows(a):
    if isinstance(a, np.memmap):
        a.flush()


def encode_batch(tokenizer, texts, max_length, **kw):
    """`tokenizer.batch_encode_plus(texts, max_length=..., padding="max_length", truncation=True, return_tensors="pt")`
    as the reference calls its vendored transformers 3.4 tokenizers (main_models.py:445-455, generate.py:85-87).
    Current `transformers` releases removed `batch_encode_plus` in favour of `__call__` (same arguments, same result);
    either spelling is used, whichever the tokenizer object has."""
    fn = getattr(tokenizer, "batch_encode_plus", None)
    if fn is None:
        fn = tokenizer.__call__
    return fn(list(texts), max_length=max_length, padding="max_length", truncation=True, return_tensors="pt", **kw)


def read(path, dim):
    """Raw f32 file -> [rows, dim] (raises like numpy if the size does not divide).  Large files (the 27 GB corpus) come
    back as a read-only memory map -- an ndarray too -- so that the only copy made is the one into the upload's staging
    buffers; np.fromfile would first copy the whole file into anonymous memory."""
    if os.path.getsize(path) >= (1 << 30):
        return np.memmap(path, dtype=np.float32, mode="r").reshape(-1, dim)
    return np.fromfile(path, dtype=np.float32).reshape(-1, dim)


class _NativeText:
    """`','.join(str(x) ...)` of a row through libmevi_hip.so's host-side formatters (textio.hip): the same bytes as
    Python's, ~15x faster -- the dense TSV of MS MARCO dev is 14 M numbers."""

    def __init__(self, width):
        import ctypes

        from . import hip

        self.lib = hip.lib()
        self.cap = 26 * max(width, 1)
        self.buf = ctypes.create_string_buffer(self.cap)
        self.ctypes = ctypes

    def f32(self, row):
        row = np.ascontiguousarray(row, dtype=np.float32)
        n = self.lib.mevi_format_f32_list(row.ctypes.data, row.size, self.buf, self.cap)
        if n < 0:
            raise RuntimeError(self.lib.mevi_last_error().decode())
        return self.ctypes.string_at(self.buf, n).decode("ascii")

    def i64(self, row):
        row = np.ascontiguousarray(row, dtype=np.int64)
        n = self.lib.mevi_format_i64_list(row.ctypes.data, row.size, self.buf, self.cap)
        if n < 0:
            raise RuntimeError(self.lib.mevi_last_error().decode())
        return self.ctypes.string_at(self.buf, n).decode("ascii")


_text = None


def join_f32(values):
    """','.join(str(float(x)) for x in values) for f32 values (the reference prints each widened to double)."""
    global _text
    values = np.ascontiguousarray(values, dtype=np.float32).reshape(-1)
    if _text is None or _text.cap < 26 * values.size:
        _text = _NativeText(max(values.size, 4096))
    return _text.f32(values)


def join_i64(values):
    global _text
    values = np.ascontiguousarray(values, dtype=np.int64).reshape(-1)
    if _text is None or _text.cap < 26 * values.size:
        _text = _NativeText(max(values.size, 4096))
    return _text.i64(values)


def to_file(query_path, output_path, dists, indices):
    """Dense ranked TSV, one line per line of `query_path` (query text = field 0)."""
    dists = np.asarray(dists)
    indices = np.asarray(indices)
    text = _NativeText(dists.shape[1] if dists.ndim == 2 else 1)
    with open(query_path, "r") as fr, open(output_path, "w") as fw:
        for i, line in enumerate(fr):
            query = line.split("\t")[0]
            fw.write(f"{query}\t\t{text.i64(indices[i])}\t{text.f32(dists[i])}\n")


class RankLog:
    """Per-rank append log merged by rank 0 -- the role of main_models.LogTxtFile.

    Every rank appends tab-joined tuples to `<tmpdir>/<basename>_<rank>`; `merge()`
    (after a barrier supplied by the caller) concatenates the rank files in rank order
    into the final path and removes them.  Tuple fields are rendered with str(), so
    lists appear as Python reprs exactly like `print(*line, sep='\\t')` does.
    """

    def __init__(self, path, rank=0, nrank=1, barrier=None, tmpdir="/tmp", flush_every=50000):
        self.path = path
        self.rank, self.nrank = rank, nrank
        self.barrier = barrier or (lambda: None)
        self.prefix = os.path.join(tmpdir, os.path.basename(path))
        self.tmp = f"{self.prefix}_{rank}"
        if os.path.exists(self.tmp):
            raise FileExistsError(f"stale rank log {self.tmp}: remove it (a previous run crashed)")
        self.flush_every = flush_every
        self.lines = []

    def add(self, fields):
        self.lines.append("\t".join(str(f) for f in fields))
        if len(self.lines) >= self.flush_every:
            self.flush()

    def flush(self):
        if self.lines:
            with open(self.tmp, "a") as f:
                f.write("\n".join(self.lines) + "\n")
            self.lines = []

    def merge(self):
        self.flush()
        if not os.path.exists(self.tmp):
            open(self.tmp, "a").close()
        self.barrier()
        if self.rank == 0:
            with open(self.path, "w") as out:
                for r in range(self.nrank):
                    part = f"{self.prefix}_{r}"
                    with open(part, "r") as f:
                        for chunk in iter(lambda: f.read(1 << 24), ""):
                            out.write(chunk)
            for r in range(self.nrank):
                part = f"{self.prefix}_{r}"
                if os.path.isfile(part):
                    os.remove(part)
        self.barrier()


# ---- readers ---------------------------------------------------------------
_parse_buf = None


def _native_numbers(field):
    """Flat comma list through the library's host-side parsers: list of int, else list of float, else None."""
    global _parse_buf
    try:
        from . import hip

        L = hip.lib()
    except Exception:          # consumers must keep working where the library is not built
        return None
    raw = field.encode("ascii", "ignore") if isinstance(field, str) else field
    cap = raw.count(b",") + 1
    if _parse_buf is None or len(_parse_buf[0]) < cap:
        _parse_buf = (np.empty(max(cap, 4096), np.int64), np.empty(max(cap, 4096), np.float64))
    ib, fb = _parse_buf
    n = L.mevi_parse_i64_list(raw, len(raw), ib.ctypes.data, cap)
    if n >= 0:
        return ib[:n].tolist()
    n = L.mevi_parse_f64_list(raw, len(raw), fb.ctypes.data, cap)
    if n >= 0:
        return fb[:n].tolist()
    return None


def parse_list(field):
    """One TSV field -> Python list, as the reference's eval_list does (ensemble_marco.py:85-89):
    a bare comma list gets brackets added.  Flat numeric lists take a fast path (native for long ones)."""
    if field[0] != "[":
        if len(field) > 256:
            out = _native_numbers(field)
            if out is not None:
                return out
        try:
            return [int(x) for x in field.split(",")]
        except ValueError:
            try:
                return [float(x) for x in field.split(",")]
            except ValueError:
                field = f"[{field}]"
    return ast.literal_eval(field)


def parse_file(path, template):
    """(pred, score, cluster) dicts keyed by query text; `template` maps role -> column."""
    qi = template["query"]
    cols = [template.get(k) for k in ("pred", "score", "cluster")]
    out = ({}, {}, {})
    with open(path, "r") as f:
        for line in f:
            items = line.rstrip("\n").split("\t")
            q = items[qi]
            for c, d in zip(cols, out):
                if c is not None:
                    d[q] = parse_list(items[c])
    return out


def load_parsed(path, template):
    """parse_file with the reference's `.pkl` side cache (ensemble_marco.py:130-139),
    except that a cache older than its TSV is ignored (the reference serves it stale)."""
    if path.endswith(".pkl"):
        with open(path, "rb") as f:
            return pickle.load(f)
    cache = path[: -(len(path.split(".")[-1]) + 1)] + ".pkl"
    if os.path.exists(cache) and os.path.getmtime(cache) >= os.path.getmtime(path):
        with open(cache, "rb") as f:
            return pickle.load(f)
    res = parse_file(path, template)
    try:
        with open(cache, "wb") as f:
            pickle.dump(res, f)
    except OSError:
        pass
    return res
