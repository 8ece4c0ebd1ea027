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


import threading

_STAGING = {}      # (buffers, bytes each) -> pinned uint8 tensors, kept between calls (hipHostMalloc is slow)
_STAGING_LOCK = threading.Lock()       # one upload at a time owns the ring: two concurrent callers would fill the same buffers
_STAGING_KEEP_BYTES = 512 << 20        # rings larger than this (an explicit, large chunk_rows) are released after the call


def _staging(nbuf, nbytes):
    """The ring for this shape; any other cached ring is dropped first (one set of pinned buffers per process)."""
    import torch

    key = (nbuf, nbytes)
    if key not in _STAGING:
        _STAGING.clear()
        _STAGING[key] = [torch.empty(nbytes, dtype=torch.uint8).pin_memory() for _ in range(nbuf)]
    return _STAGING[key]


def free_staging():
    """Release the pinned staging buffers upload_rows keeps between calls."""
    with _STAGING_LOCK:
        _STAGING.clear()


def file_range(src):
    """(path, byte offset of src[0, 0] in the file) when `src` is a C-contiguous f32 view of a file-backed np.memmap, else None.
    The offset comes from ADDRESSES -- a sliced memmap keeps the root's `.offset` (numpy does not adjust it), so
    `root.offset + (address of the view - address of the root)` is the only reading that is right for `m[a:b]` too."""
    if not isinstance(src, np.memmap) or not getattr(src, "filename", None) or not src.flags.c_contiguous or src.dtype != np.float32:
        return None
    root = src
    while isinstance(root.base, np.memmap):
        root = root.base
    if not isinstance(root, np.memmap) or root.size == 0:
        return None
    delta = src.__array_interface__["data"][0] - root.__array_interface__["data"][0]
    if delta < 0 or delta + src.nbytes > root.nbytes:
        return None
    return str(src.filename), int(root.offset) + int(delta)


def upload_rows(src, device, chunk_rows=None, threads=8, buffers=4):
    """f32 [rows, dim] host array or memmap -> CUDA tensor through a ring of pinned staging buffers (64 MiB each unless
    `chunk_rows` says otherwise): reader threads fill the buffers AHEAD of the copy engine -- a buffer is refilled as soon as
    its own copy has finished, while the copies of the other buffers are queued -- so host reads and DMA overlap instead of
    alternating.  A file-backed memmap (faiss_search.read / map_rows) is pread straight from the page cache into the pinned
    buffer (one kernel copy, no page-table population of a 27 GB mapping); anything else is copied out of memory.
    Measured on the pool's boxes (16-CPU cgroup, tools/probe_upload2.py): fills 77-100 GB/s with 8-16 threads, pinned -> device
    DMA 57 GB/s; the round-4 form (two 800 MB buffers allocated per call, fill and copy in turns) delivered 5.8 GB/s.
    MEVI_UPLOAD=copy: memcpy out of the mapping instead of pread (A/B)."""
    import torch

    if src.dtype != np.float32:
        raise TypeError(f"upload_rows: f32 rows expected, got {src.dtype} (the staging ring moves raw bytes)")
    rows, dim = src.shape
    out = torch.empty((rows, dim), dtype=torch.float32, device=device)
    if rows == 0:
        return out
    with _STAGING_LOCK:          # the ring is shared state: concurrent uploads (threads, devices) take turns
        try:
            return _upload_rows_locked(src, out, chunk_rows, threads, buffers)
        finally:
            if sum(k[0] * k[1] for k in _STAGING) > _STAGING_KEEP_BYTES:
                _STAGING.clear()


def _upload_rows_locked(src, out, chunk_rows, threads, buffers):
    from collections import deque
    from concurrent.futures import ThreadPoolExecutor

    import torch

    rows, dim = src.shape
    row_bytes = dim * 4
    if chunk_rows is None:
        chunk_rows = max(1, (64 << 20) // row_bytes)
    chunk_rows = max(1, min(chunk_rows, rows))
    nchunks = (rows + chunk_rows - 1) // chunk_rows
    buffers = max(2, min(buffers, nchunks + 1))
    stage = _staging(buffers, chunk_rows * row_bytes)
    views = [b.numpy() for b in stage]
    done = [torch.cuda.Event() for _ in range(buffers)]
    threads = max(1, min(threads, os.cpu_count() or 1))
    fr = file_range(src) if os.environ.get("MEVI_UPLOAD", "pread") != "copy" else None
    fd = None
    if fr is not None:
        try:
            fd = os.open(fr[0], os.O_RDONLY)
        except OSError:
            fd = None
    src_bytes = None if fd is not None else np.ascontiguousarray(src).view(np.uint8).reshape(rows, row_bytes) \
        if not src.flags.c_contiguous else src.view(np.uint8).reshape(rows, row_bytes)
    piece = max(1 << 20, (chunk_rows * row_bytes + threads - 1) // threads)      # bytes per reader task

    def fill(b, byte0, lo, hi):
        """bytes [lo, hi) of the chunk that starts at byte `byte0` of the matrix -> staging buffer b."""
        dst = memoryview(views[b])[lo:hi]
        if fd is None:
            flat = src_bytes.reshape(-1)
            np.copyto(np.frombuffer(dst, np.uint8), flat[byte0 + lo:byte0 + hi])
            return
        off, got = fr[1] + byte0 + lo, 0
        while got < len(dst):
            n = os.preadv(fd, [dst[got:]], off + got)
            if n <= 0:
                raise OSError(f"short read of {fr[0]} at byte {off + got}")
            got += n

    out_bytes = out.view(torch.uint8).view(-1)
    try:
        with torch.cuda.device(out.device), ThreadPoolExecutor(threads) as pool:
            stream = torch.cuda.current_stream()
            fills = deque()

            def submit(i):
                a = i * chunk_rows
                n = (min(a + chunk_rows, rows) - a) * row_bytes
                fills.append((i, n, [pool.submit(fill, i % buffers, a * row_bytes, lo, min(lo + piece, n)) for lo in range(0, n, piece)]))

            nxt = 0
            while nxt < min(nchunks, buffers - 1):
                submit(nxt)
                nxt += 1
            while fills:
                i, n, futs = fills.popleft()
                for f in futs:
                    f.result()
                a = i * chunk_rows * row_bytes
                out_bytes[a:a + n].copy_(stage[i % buffers][:n], non_blocking=True)
                done[i % buffers].record(stream)
                if nxt < nchunks:        # the buffer chunk `nxt` takes was last read by the copy of chunk nxt - buffers
                    if nxt >= buffers:
                        done[nxt % buffers].synchronize()
                    submit(nxt)
                    nxt += 1
            stream.synchronize()
    finally:
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


def load_checkpoint(path):
    """`torch.load(path, map_location='cpu')` of a checkpoint the user trusts, as the reference loads its own (MEVI/main.py:
    203-248, MEVI/generate.py:200-211) -- memory-mapped when the file is in torch's zip format (the tensors are then pages of
    the file until they are uploaded: no 1-2 GB copy into anonymous memory first), and with the full unpickler when the file
    holds more than tensors (a Lightning checkpoint carries its hyper-parameters as an argparse.Namespace, which
    `weights_only=True`, the default of current torch releases, refuses)."""
    import pickle as _pickle

    import torch

    for kw in ({"mmap": True, "weights_only": True}, {"weights_only": True}, {"mmap": True, "weights_only": False},
               {"weights_only": False}):
        try:
            return torch.load(path, map_location="cpu", **kw)
        except (RuntimeError, ValueError, _pickle.UnpicklingError, TypeError):
            continue
    return torch.load(path, map_location="cpu", weights_only=False)      # raises what torch raises


class SpmT5Tokenizer:
    """The T5 SentencePiece tokenizer of a checkpoint directory WITHOUT importing `transformers` (whose import costs a CLI
    process about a second -- tools/e2e_cli.py: generate.py and main.py each paid it for 6980 short strings).

    What the reference's tokenizer does to a plain string (vendored transformers 3.4 `T5Tokenizer`: `_tokenize` =
    `sp_model.EncodeAsPieces(text)`, `build_inputs_with_special_tokens` appends `</s>`; MEVI/generate.py:85-87,
    MEVI/main_models.py:445-455 call it with `max_length=L, padding='max_length', truncation=True`) is: SentencePiece ids,
    cut to L - 1, eos, pad to L; attention mask 1 on the real tokens.  That is computed here with the `sentencepiece` package
    itself.  A text that contains a special-token spelling (`</s>`, `<pad>`, `<unk>`, `<extra_id_N>`: split out by the HF
    tokenizer before SentencePiece sees the rest) is handed, together with its whole batch, to the HF tokenizer, imported at
    that moment.  `MEVI_TOKENIZER=hf` always uses the HF object."""

    SPECIAL = ("</s>", "<pad>", "<unk>", "<extra_id_")

    def __init__(self, path):
        import sentencepiece

        self.path = path
        self.sp = sentencepiece.SentencePieceProcessor(model_file=os.path.join(path, "spiece.model"))
        self.eos_id, self.pad_id = self.sp.piece_to_id("</s>"), self.sp.piece_to_id("<pad>")
        if self.sp.id_to_piece(self.eos_id) != "</s>" or self.sp.id_to_piece(self.pad_id) != "<pad>":
            raise ValueError("spiece.model without </s> / <pad> pieces")
        self._hf = None

    def hf(self):
        if self._hf is None:
            from transformers import AutoTokenizer

            self._hf = AutoTokenizer.from_pretrained(self.path)
        return self._hf

    def __call__(self, texts, max_length=None, padding=False, truncation=False, return_tensors=None, **kw):
        import torch

        texts = [texts] if isinstance(texts, str) else list(texts)
        plain = (padding == "max_length" and truncation is True and return_tensors == "pt" and max_length and not kw
                 and all(isinstance(t, str) and not any(sp in t for sp in self.SPECIAL) for t in texts))
        if not plain:
            return self.hf()(texts, max_length=max_length, padding=padding, truncation=truncation, return_tensors=return_tensors, **kw)
        ids = np.full((len(texts), max_length), self.pad_id, np.int64)
        mask = np.zeros((len(texts), max_length), np.int64)
        for i, row in enumerate(self.sp.encode(texts)):
            row = row[:max_length - 1] + [self.eos_id]
            ids[i, :len(row)] = row
            mask[i, :len(row)] = 1
        return {"input_ids": torch.from_numpy(ids), "attention_mask": torch.from_numpy(mask)}

    batch_encode_plus = __call__


def load_tokenizer(path):
    """The tokenizer of a checkpoint directory: `AutoTokenizer.from_pretrained(path)` (host-side tokenisation stays an HF call,
    SURVEY 8b) -- except for T5 SentencePiece directories, which get SpmT5Tokenizer (same ids, no `transformers` import)."""
    if os.environ.get("MEVI_TOKENIZER", "") != "hf" and os.path.isfile(os.path.join(path, "spiece.model")):
        cls = None
        cfg = os.path.join(path, "tokenizer_config.json")
        if os.path.isfile(cfg):
            import json

            with open(cfg) as f:
                cls = json.load(f).get("tokenizer_class")
        if cls in (None, "T5Tokenizer", "T5TokenizerFast"):
            try:
                return SpmT5Tokenizer(path)
            except Exception:           # no sentencepiece package, a model without the T5 special pieces, ...
                pass
    from transformers import AutoTokenizer

    return AutoTokenizer.from_pretrained(path)


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
    with open(query_path, "r") as fr:
        queries = [line.split("\t")[0] for line in fr]
    n = len(queries)
    if dists.ndim != 2 or indices.shape != dists.shape or n > dists.shape[0] or n * dists.shape[1] < (1 << 16):
        text = _NativeText(dists.shape[1] if dists.ndim == 2 else 1)         # small or ragged: row by row
        with open(output_path, "w") as fw:
            for i, query in enumerate(queries):
                fw.write(f"{query}\t\t{text.i64(indices[i])}\t{text.f32(dists[i])}\n")
        return
    # MS MARCO dev is 6980 x 1000 -> 14 M renderings (0.32 s row by row): both list columns of every row in ONE native call
    # whose host threads share the rows (mevi_format_ranked_rows); Python only joins query text and rows
    from . import hip

    k = dists.shape[1]
    ids = np.ascontiguousarray(indices[:n], dtype=np.int64)
    sc = np.ascontiguousarray(dists[:n], dtype=np.float32)
    row_cap = 47 * k + 1
    out = np.empty(n * row_cap, dtype=np.uint8)
    lens = np.empty(n, dtype=np.int64)
    hip.check(hip.lib().mevi_format_ranked_rows(ids.ctypes.data, sc.ctypes.data, n, k, out.ctypes.data, row_cap, lens.ctypes.data,
                                                min(8, os.cpu_count() or 1)), "mevi_format_ranked_rows")
    mv = memoryview(out)
    with open(output_path, "w") as fw:
        for i, query in enumerate(queries):
            fw.write(query)
            fw.write("\t\t")
            fw.write(str(mv[i * row_cap:i * row_cap + int(lens[i])], "ascii"))
            fw.write("\n")


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
