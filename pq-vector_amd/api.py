"""Host-side mirror of the reference's builder API over the C ABI (include/pqv.h).

Names, argument meaning and error texts follow src/ivf/parquet.rs:23-103 (IndexBuilder),
src/ivf/search.rs:41-81 (TopkBuilder, SearchResult) and src/ivf/index.rs:9-14 (IvfIndex).
"""
import ctypes as C
from dataclasses import dataclass

import numpy as np

from . import _ffi
from ._ffi import f32p, f64p, u8p, u32p, u64p, vp


class PqvError(Exception):
    """Carries the library's status code and the reference's message text."""

    def __init__(self, code, message):
        super().__init__(message)
        self.code = code
        self.message = message


def _check(rc):
    if rc != _ffi.PQV_OK:
        raise PqvError(rc, _ffi.lib().pqv_last_error().decode())


def _f32(a):
    return np.ascontiguousarray(a, dtype=np.float32)


def device_count():
    return _ffi.lib().pqv_device_count()


# ---------------------------------------------------------------------------------------
class Corpus:
    """The embedding column resident in one GPU's HBM (row-major [n, dim] f32)."""

    def __init__(self, handle, keepalive=None):
        self._h = handle
        self._keepalive = keepalive

    @classmethod
    def upload(cls, rows, device=0):
        rows = np.asarray(rows)
        if rows.ndim != 2:
            raise PqvError(_ffi.PQV_ERR_INVALID, "Embedding data length must be a multiple of dimension")
        n, dim = rows.shape
        h = vp()
        if rows.dtype == np.float64:  # narrowed like src/ivf/parquet.rs:246-256
            _check(_ffi.lib().pqv_corpus_create(device, n, dim, C.byref(h)))
            c = cls(h)
            r64 = np.ascontiguousarray(rows)
            _check(_ffi.lib().pqv_corpus_append_f64(h, r64.ctypes.data_as(f64p), n))
            return c
        r = _f32(rows)
        _check(_ffi.lib().pqv_corpus_upload(device, r.ctypes.data_as(f32p), n, dim, C.byref(h)))
        return cls(h)

    @classmethod
    def create(cls, capacity_rows, dim, device=0):
        h = vp()
        _check(_ffi.lib().pqv_corpus_create(device, capacity_rows, dim, C.byref(h)))
        return cls(h)

    def append(self, rows):
        rows = np.asarray(rows)
        if rows.dtype == np.float64:
            r = np.ascontiguousarray(rows)
            _check(_ffi.lib().pqv_corpus_append_f64(self._h, r.ctypes.data_as(f64p), r.shape[0]))
        else:
            r = _f32(rows)
            _check(_ffi.lib().pqv_corpus_append(self._h, r.ctypes.data_as(f32p), r.shape[0]))

    def write_rows(self, row_offset, rows):
        """Streaming upload (pqv_corpus_write_rows): rows [row_offset, row_offset + len(rows)) from a [m, dim] f32 / f64 array,
        staged in pinned memory and DMA'd asynchronously.  Thread-safe; batches may arrive in any order.  finish() completes."""
        rows = np.asarray(rows)
        if rows.dtype == np.float64:
            r = np.ascontiguousarray(rows)
            _check(_ffi.lib().pqv_corpus_write_rows_f64(self._h, row_offset, r.ctypes.data_as(f64p), r.shape[0]))
        else:
            r = _f32(rows)
            _check(_ffi.lib().pqv_corpus_write_rows(self._h, row_offset, r.ctypes.data_as(f32p), r.shape[0]))

    def write_rows_ptr(self, row_offset, address, n_rows, f64=False):
        """write_rows from a raw host address (a page inside a memory-mapped file: no intermediate array)."""
        fn = _ffi.lib().pqv_corpus_write_rows_f64 if f64 else _ffi.lib().pqv_corpus_write_rows
        _check(fn(self._h, row_offset, C.cast(C.c_void_p(address), f64p if f64 else f32p), n_rows))

    def write_plain_pages(self, file_base, body_off, body_len, first_value, n_values, dim, max_def, f64=False):
        """A run of uncompressed PLAIN data pages from a mapped file (pqv_corpus_write_plain_pages): level runs checked and values
        uploaded natively.  Returns None, or the index of the first page that is not what the page plan assumed."""
        bad = C.c_uint32(0)
        rc = _ffi.lib().pqv_corpus_write_plain_pages(self._h, C.cast(C.c_void_p(file_base), _ffi.u8p), body_off.ctypes.data_as(_ffi.u64p),
                                                     body_len.ctypes.data_as(_ffi.u32p), first_value.ctypes.data_as(_ffi.u64p),
                                                     n_values.ctypes.data_as(_ffi.u32p), len(body_off), dim, max_def, 1 if f64 else 0, C.byref(bad))
        if rc == 1:
            return int(bad.value)
        _check(rc)
        return None

    def finish(self, n_rows):
        _check(_ffi.lib().pqv_corpus_finish(self._h, n_rows))

    @classmethod
    def from_device_ptr(cls, ptr, n, dim, device=0, keepalive=None):
        """Adopt a device buffer (e.g. a torch tensor's data_ptr()); `keepalive` pins its owner."""
        h = vp()
        _check(_ffi.lib().pqv_corpus_from_device(device, vp(ptr), n, dim, C.byref(h)))
        return cls(h, keepalive)

    @property
    def rows(self):
        return _ffi.lib().pqv_corpus_rows(self._h)

    @property
    def dim(self):
        return _ffi.lib().pqv_corpus_dim(self._h)

    @property
    def device(self):
        return _ffi.lib().pqv_corpus_device(self._h)

    def fetch_rows(self, row_ids):
        ids = np.ascontiguousarray(row_ids, dtype=np.uint32)
        out = np.empty((ids.size, self.dim), dtype=np.float32)
        _check(_ffi.lib().pqv_corpus_fetch_rows(self._h, ids.ctypes.data_as(u32p), ids.size,
                                                out.ctypes.data_as(f32p)))
        return out

    def brute_topk(self, queries, k, metric=_ffi.PQV_COSINE):
        """Exhaustive batched top-k over every resident row on the matrix cores (extension:
        cosine / norm-expansion L2).  Returns (row_idx [nq,k], dist [nq,k], n_found [nq])."""
        q = _f32(queries)
        if q.ndim == 1:
            q = q.reshape(1, -1)
        nq, qlen = q.shape
        rows = np.full((nq, max(k, 1)), 0xFFFFFFFF, dtype=np.uint32)
        dist = np.full((nq, max(k, 1)), np.inf, dtype=np.float32)
        nf = np.zeros(nq, dtype=np.uint32)
        _check(_ffi.lib().pqv_brute_topk(self._h, q.ctypes.data_as(f32p), nq, qlen, k, metric,
                                         rows.ctypes.data_as(u32p), dist.ctypes.data_as(f32p),
                                         nf.ctypes.data_as(u32p)))
        return rows, dist, nf

    def close(self):
        if self._h:
            _ffi.lib().pqv_corpus_free(self._h)
            self._h = None
        self._keepalive = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


# ---------------------------------------------------------------------------------------
class Index:
    """IvfIndex{dim, n_clusters, centroids, inverted_lists} (src/ivf/index.rs:9-14)."""

    def __init__(self, handle):
        self._h = handle

    @classmethod
    def from_bytes(cls, blob):
        h = vp()
        blob = bytes(blob)
        _check(_ffi.lib().pqv_index_from_bytes(blob, len(blob), C.byref(h)))
        return cls(h)

    @classmethod
    def from_parts(cls, dim, centroids, lists):
        cent = _f32(centroids).reshape(-1)
        k = len(lists)
        off = np.zeros(k + 1, dtype=np.uint64)
        for i, l in enumerate(lists):
            off[i + 1] = off[i] + len(l)
        rows = (np.concatenate([np.asarray(l, dtype=np.uint32) for l in lists])
                if k and off[-1] else np.zeros(0, dtype=np.uint32))
        rows = np.ascontiguousarray(rows, dtype=np.uint32)
        h = vp()
        _check(_ffi.lib().pqv_index_from_parts(dim, k, cent.ctypes.data_as(f32p),
                                               off.ctypes.data_as(u64p), rows.ctypes.data_as(u32p),
                                               C.byref(h)))
        return cls(h)

    def to_bytes(self):
        buf = u8p()
        n = C.c_size_t(0)
        _check(_ffi.lib().pqv_index_to_bytes(self._h, C.byref(buf), C.byref(n)))
        try:
            return C.string_at(buf, n.value)
        finally:
            _ffi.lib().pqv_bytes_free(buf)

    @property
    def dim(self):
        return _ffi.lib().pqv_index_dim(self._h)

    @property
    def n_clusters(self):
        return _ffi.lib().pqv_index_n_clusters(self._h)

    @property
    def n_rows(self):
        return _ffi.lib().pqv_index_n_rows(self._h)

    @property
    def centroids(self):
        p = _ffi.lib().pqv_index_centroids(self._h)
        return np.ctypeslib.as_array(p, shape=(self.n_clusters, self.dim)).copy()

    @property
    def list_offsets(self):
        p = _ffi.lib().pqv_index_list_offsets(self._h)
        return np.ctypeslib.as_array(p, shape=(self.n_clusters + 1,)).copy()

    @property
    def list_rows(self):
        n = self.n_rows
        if n == 0:
            return np.zeros(0, dtype=np.uint32)
        p = _ffi.lib().pqv_index_list_rows(self._h)
        return np.ctypeslib.as_array(p, shape=(n,)).copy()

    def inverted_lists(self):
        off, rows = self.list_offsets, self.list_rows
        return [rows[int(off[i]):int(off[i + 1])] for i in range(self.n_clusters)]

    def close(self):
        if self._h:
            _ffi.lib().pqv_index_free(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


# ---------------------------------------------------------------------------------------
class IndexBuilder:
    """src/ivf/parquet.rs:23-103.  `source` is a Parquet path (with `embedding_column`, as in the
    reference), a Corpus or a [n, dim] array (the in-memory form of the embedding column);
    defaults n_clusters=None -> ceil(sqrt(n)), max_iters=20, seed=42 (:32-39)."""

    def __init__(self, source, embedding_column=None, device=0):
        self._source = source
        self._embedding_column = embedding_column
        self._device = device
        self._n_clusters = None
        self._max_iters = 20
        self._seed = 42
        self._workers = 0

    def n_clusters(self, n_clusters):
        self._n_clusters = n_clusters
        return self

    def max_iters(self, max_iters):
        self._max_iters = max_iters
        return self

    def seed(self, seed):
        self._seed = seed
        return self

    def workers(self, workers):
        """The available_parallelism() to reproduce; 0 = this host's CPU count."""
        self._workers = workers
        return self

    def _config(self):
        # build_config, src/ivf/parquet.rs:88-102
        if self._max_iters == 0:
            raise PqvError(_ffi.PQV_ERR_INVALID, "max_iters must be > 0")
        if self._n_clusters is not None and self._n_clusters == 0:
            raise PqvError(_ffi.PQV_ERR_INVALID, "n_clusters must be > 0")
        if self._embedding_column is not None and not str(self._embedding_column).strip():
            raise PqvError(_ffi.PQV_ERR_INVALID, "Embedding column name cannot be empty")
        return (self._n_clusters or 0), self._max_iters, self._seed, self._workers

    def _is_path(self):
        import os
        return isinstance(self._source, (str, bytes, os.PathLike))

    def _require_column(self):
        if not self._is_path():
            raise PqvError(_ffi.PQV_ERR_INVALID, "build_inplace/build_new need a Parquet path as source")
        if self._embedding_column is None or not str(self._embedding_column).strip():
            raise PqvError(_ffi.PQV_ERR_INVALID, "Embedding column name cannot be empty")   # mod.rs:25

    def _build_on(self, corpus):
        nc, mi, seed, workers = self._config()
        h = vp()
        _check(_ffi.lib().pqv_index_build(corpus._h, nc, mi, seed, workers, C.byref(h)))
        return Index(h)

    def _load_parquet(self):
        from . import parquet_io
        self._config()                       # build_config() runs first (parquet.rs:58,72)
        self._require_column()
        self.last_stats = {"load": {}}
        return parquet_io.load_embedding_column(self._source, self._embedding_column, self._device, stats=self.last_stats["load"])

    def _timed_build(self, write):
        """load -> build -> write, with the wall time of each in self.last_stats (the reference's benches/index_build.rs times
        the three together)."""
        import time
        t0 = time.perf_counter()
        corpus = self._load_parquet()
        t1 = time.perf_counter()
        index = self._build_on(corpus)
        t2 = time.perf_counter()
        write(index)
        t3 = time.perf_counter()
        self.last_stats.update({"load_s": t1 - t0, "build_s": t2 - t1, "write_s": t3 - t2, "total_s": t3 - t0})
        corpus.close()
        return index

    def build_inplace(self):
        """Build and append the index to the source file (src/ivf/parquet.rs:57-69)."""
        from . import parquet_io
        return self._timed_build(lambda index: parquet_io.append_index_inplace(self._source, index, self._embedding_column))

    def build_new(self, output):
        """Build and write a new file that carries the index (src/ivf/parquet.rs:71-86)."""
        from . import parquet_io
        return self._timed_build(lambda index: parquet_io.write_parquet_with_index(self._source, output, index, self._embedding_column))

    def build(self):
        """In-memory form: returns the Index without touching any file."""
        nc, mi, seed, workers = self._config()
        if self._is_path():
            return self._build_on(self._load_parquet())
        if isinstance(self._source, Corpus):
            return self._build_on(self._source)
        data = np.asarray(self._source)
        if data.ndim != 2:
            raise PqvError(_ffi.PQV_ERR_INVALID, "Embedding data length must be a multiple of dimension")
        dim = data.shape[1]
        flat = _f32(data).reshape(-1)
        h = vp()
        _check(_ffi.lib().pqv_index_build_host(self._device, flat.ctypes.data_as(f32p), flat.size,
                                               dim, nc, mi, seed, workers, C.byref(h)))
        return Index(h)


# ---------------------------------------------------------------------------------------
class CandidateCursor:
    """src/df_vector/access.rs:193-243: round-robin over per-file candidate lists until a cap."""

    def __init__(self, file_count):
        h = vp()
        _check(_ffi.lib().pqv_candidate_cursor_new(file_count, C.byref(h)))
        self._h = h
        self.file_count = file_count

    def add_candidates(self, idx, rows):
        r = np.ascontiguousarray(rows, dtype=np.uint32)
        _check(_ffi.lib().pqv_candidate_cursor_add(self._h, idx, r.ctypes.data_as(u32p), r.size))

    def next_batch(self, batch_size):
        """Returns ([(file_idx, row), ...], cumulative rows taken per file)."""
        of = np.zeros(max(1, batch_size), dtype=np.uint32)
        orow = np.zeros(max(1, batch_size), dtype=np.uint32)
        n = C.c_uint64(0)
        taken = np.zeros(max(1, self.file_count), dtype=np.uint64)
        _check(_ffi.lib().pqv_candidate_cursor_next_batch(self._h, batch_size, of.ctypes.data_as(u32p), orow.ctypes.data_as(u32p),
                                                         C.byref(n), taken.ctypes.data_as(u64p)))
        return list(zip(of[:n.value].tolist(), orow[:n.value].tolist())), taken[:self.file_count].copy()

    def __del__(self):
        try:
            if self._h:
                _ffi.lib().pqv_candidate_cursor_free(self._h)
                self._h = None
        except Exception:
            pass


# ---------------------------------------------------------------------------------------
@dataclass
class SearchResult:
    """src/ivf/search.rs:41-45"""
    row_idx: int
    distance: float


class Searcher:
    """An index bound to a resident corpus on that corpus' GPU."""

    def __init__(self, index, corpus, flags=_ffi.PQV_LAYOUT_IVF_ORDERED):
        h = vp()
        _check(_ffi.lib().pqv_searcher_create(index._h, corpus._h, flags, C.byref(h)))
        self._h = h
        self._corpus = corpus
        self.dim = index.dim
        self.n_clusters = index.n_clusters

    def probe(self, query, nprobe):
        q = _f32(query).reshape(-1)
        out = np.zeros(max(1, min(nprobe, self.n_clusters)), dtype=np.uint32)
        n = C.c_uint32(0)
        _check(_ffi.lib().pqv_probe(self._h, q.ctypes.data_as(f32p), q.size, nprobe,
                                    out.ctypes.data_as(u32p), C.byref(n)))
        return out[:n.value].copy()

    def candidate_rows(self, query, nprobe):
        q = _f32(query).reshape(-1)
        rows = u32p()
        n = C.c_uint64(0)
        _check(_ffi.lib().pqv_candidate_rows(self._h, q.ctypes.data_as(f32p), q.size, nprobe,
                                             C.byref(rows), C.byref(n)))
        try:
            return (np.ctypeslib.as_array(rows, shape=(n.value,)).copy() if n.value
                    else np.zeros(0, dtype=np.uint32))
        finally:
            _ffi.lib().pqv_rows_free(rows)

    def topk(self, queries, k, nprobe, max_candidates=0, metric=_ffi.PQV_L2SQ_REF4, sqrt_out=True):
        """Batched topk(); returns (row_idx [nq,k] u32, dist [nq,k] f32, n_found [nq], n_candidates [nq])."""
        q = _f32(queries)
        if q.ndim == 1:
            q = q.reshape(1, -1)
        nq, qlen = q.shape
        rows = np.full((nq, max(k, 1)), 0xFFFFFFFF, dtype=np.uint32)
        dist = np.full((nq, max(k, 1)), np.inf, dtype=np.float32)
        nf = np.zeros(nq, dtype=np.uint32)
        nc = np.zeros(nq, dtype=np.uint64)
        _check(_ffi.lib().pqv_topk(self._h, q.ctypes.data_as(f32p), nq, qlen, k, nprobe, max_candidates,
                                   metric, 1 if sqrt_out else 0, rows.ctypes.data_as(u32p),
                                   dist.ctypes.data_as(f32p), nf.ctypes.data_as(u32p),
                                   nc.ctypes.data_as(u64p)))
        return rows, dist, nf, nc

    def topk_device(self, d_queries, nq, k, nprobe, d_row_idx, d_dist, d_n_found=0, d_n_candidates=0,
                    max_candidates=0, metric=_ffi.PQV_L2SQ_REF4, sqrt_out=True, stream=0, d_tie_flags=0):
        """Device-pointer form (ints from tensor.data_ptr()); asynchronous on `stream` -- a hipStream_t handle; 0 means the
        searcher's OWN non-blocking stream, not HIP's / torch's default stream (whose handle is 0 too): work that must follow
        the call on the default stream is NOT ordered behind it, so pass an explicit stream (1 = hipStreamLegacy names the default
        stream itself).  d_tie_flags (u32 [nq]):
        also flag the queries whose answer depends on the reference's heap history (re-submit those to topk())."""
        if d_tie_flags:
            _check(_ffi.lib().pqv_topk_device_flags(self._h, vp(d_queries), nq, k, nprobe, max_candidates, metric,
                                                    1 if sqrt_out else 0, vp(d_row_idx), vp(d_dist),
                                                    vp(d_n_found or None), vp(d_n_candidates or None), vp(d_tie_flags),
                                                    vp(stream or None)))
            return
        _check(_ffi.lib().pqv_topk_device(self._h, vp(d_queries), nq, k, nprobe, max_candidates, metric,
                                          1 if sqrt_out else 0, vp(d_row_idx), vp(d_dist),
                                          vp(d_n_found or None), vp(d_n_candidates or None),
                                          vp(stream or None)))

    def counters(self):
        c = _ffi.Counters()
        _check(_ffi.lib().pqv_counters(self._h, C.byref(c)))
        return {"queries": c.queries, "candidate_rows": c.candidate_rows,
                "embeddings_fetched": c.embeddings_fetched, "kernel_launches": c.kernel_launches,
                "exact_replays": c.exact_replays, "screened_pairs": c.screened_pairs,
                "screen_survivors": c.screen_survivors}

    def set_option(self, name, value):
        """Force a dispatch choice (pqv.h: pqv_searcher_set_option); results never change."""
        _check(_ffi.lib().pqv_searcher_set_option(self._h, str(name).encode(), int(value)))
        return self

    def describe(self, nq, k, nprobe, metric=_ffi.PQV_L2SQ_REF4):
        buf = C.create_string_buffer(2048)
        _check(_ffi.lib().pqv_searcher_describe(self._h, nq, k, nprobe, metric, buf, len(buf)))
        return buf.value.decode()

    def footprint(self):
        v = [C.c_uint64(0) for _ in range(4)]
        _check(_ffi.lib().pqv_searcher_footprint(self._h, *[C.byref(x) for x in v]))
        return {"row_order_bytes": v[0].value, "ivf_rows_bytes": v[1].value, "blocked_bytes": v[2].value,
                "other_bytes": v[3].value, "total_bytes": sum(x.value for x in v)}

    def set_timing(self, enabled):
        _check(_ffi.lib().pqv_set_timing(self._h, 1 if enabled else 0))

    def timing_read(self):
        rr, tot, n = C.c_double(0), C.c_double(0), C.c_uint32(0)
        _check(_ffi.lib().pqv_timing_read(self._h, C.byref(rr), C.byref(tot), C.byref(n)))
        return rr.value, tot.value, n.value

    def close(self):
        if self._h:
            _ffi.lib().pqv_searcher_free(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


_PATH_SEARCHERS = {}
_REALPATHS = {}


def searcher_for_parquet(path, device=0):
    """Index + embedding column of an indexed Parquet file, resident on `device`.  Cached per
    (path, size, mtime): the reference re-opens the file, re-parses the blob and re-reads the
    candidate rows on every query (src/ivf/search.rs:89,102-110); here they stay in HBM."""
    import os
    from . import parquet_io
    st = os.stat(path)
    real = _REALPATHS.get(path)              # (realpath is a handful of system calls: once per spelling of the path)
    if real is None:
        real = _REALPATHS.setdefault(path, os.path.realpath(path))
    key = (real, st.st_size, st.st_mtime_ns, device)
    hit = _PATH_SEARCHERS.get(key)
    if hit is None:
        index, column = parquet_io.read_index_from_parquet(path)
        corpus = parquet_io.load_embedding_column(path, column, device)
        # ONE f32 copy of the column stays resident either way: the images-only IVF layout keeps reading the column as loaded;
        # where the searcher falls back to a list-ordered f32 copy of its own (dim % 64 != 0, short lists, PQV_IVF_COPY=1) the
        # loaded row-order rows are released once that copy exists
        hit = Searcher(index, corpus, _ffi.PQV_LAYOUT_IVF_ORDERED | _ffi.PQV_RELEASE_IF_COPIED)
        _PATH_SEARCHERS.clear()          # one resident file at a time by default
        _PATH_SEARCHERS[key] = hit
    return hit


class TopkBuilder:
    """src/ivf/search.rs:49-81: k and nprobe must be set and > 0.  `source` is an indexed
    Parquet path (as in the reference) or an existing Searcher."""

    def __init__(self, source, query, device=0):
        import os
        if isinstance(source, (str, bytes, os.PathLike)):
            self._path, self._searcher = source, None
        else:
            self._path, self._searcher = None, source
        self._device = device
        self._query = query
        self._k = None
        self._nprobe = None

    def k(self, k):
        if k == 0:
            raise PqvError(_ffi.PQV_ERR_INVALID, "k must be > 0")
        self._k = k
        return self

    def nprobe(self, nprobe):
        if nprobe == 0:
            raise PqvError(_ffi.PQV_ERR_INVALID, "nprobe must be > 0")
        self._nprobe = nprobe
        return self

    def search(self):
        if self._k is None:
            raise PqvError(_ffi.PQV_ERR_INVALID, "k must be set")
        if self._nprobe is None:
            raise PqvError(_ffi.PQV_ERR_INVALID, "nprobe must be set")
        if self._searcher is None:
            self._searcher = searcher_for_parquet(self._path, self._device)
        rows, dist, nf, _ = self._searcher.topk(_f32(self._query).reshape(1, -1), self._k, self._nprobe)
        n = int(nf[0])
        return [SearchResult(r, d) for r, d in zip(rows[0, :n].tolist(), dist[0, :n].tolist())]


# ---------------------------------------------------------------------------------------
def rerank_batch(query, cand, k, state=None, ids=None, valid=None, metric=_ffi.PQV_L2SQ_SEQ, device=0):
    """update_topk_heap for one RecordBatch (src/df_vector/exec.rs:457-484).

    state = (rows u32[<=k], d2 f32[<=k]): the reference heap's backing array after the previous batch (None for the
    first); returns the new state.  A float64 `cand` is narrowed `as f32` by the library (exec.rs:538-545).
    rerank_finish(state) gives the rows in output order."""
    q = _f32(query).reshape(-1)
    cand = np.asarray(cand)
    f64 = cand.dtype == np.float64
    cand = np.ascontiguousarray(cand, dtype=np.float64 if f64 else np.float32)
    m, dim = cand.shape if cand.ndim == 2 else (0, q.size)
    io_rows = np.zeros(max(k, 1), dtype=np.uint32)
    io_d2 = np.zeros(max(k, 1), dtype=np.float32)
    cnt = C.c_uint32(0)
    if state is not None:
        r, d = state
        cnt.value = len(r)
        io_rows[:len(r)] = r
        io_d2[:len(r)] = d
    ids_a = None if ids is None else np.ascontiguousarray(ids, dtype=np.uint32)
    valid_a = None if valid is None else np.ascontiguousarray(valid, dtype=np.uint8)
    fn = _ffi.lib().pqv_rerank_f64 if f64 else _ffi.lib().pqv_rerank
    _check(fn(device, q.ctypes.data_as(f32p), cand.ctypes.data_as(_ffi.f64p if f64 else f32p),
              None if ids_a is None else ids_a.ctypes.data_as(u32p),
              None if valid_a is None else valid_a.ctypes.data_as(u8p),
              m, dim, k, metric, io_rows.ctypes.data_as(u32p),
              io_d2.ctypes.data_as(f32p), C.byref(cnt)))
    return io_rows[:cnt.value].copy(), io_d2[:cnt.value].copy()


def rerank_finish(state):
    """heap.into_iter() + the stable sort by distance (src/df_vector/exec.rs:269-274): (rows, d2) ascending."""
    r = np.ascontiguousarray(state[0], dtype=np.uint32)
    d = np.ascontiguousarray(state[1], dtype=np.float32)
    orow = np.empty(max(len(r), 1), dtype=np.uint32)
    od = np.empty(max(len(r), 1), dtype=np.float32)
    _check(_ffi.lib().pqv_rerank_finish(r.ctypes.data_as(u32p), d.ctypes.data_as(f32p), len(r),
                                        orow.ctypes.data_as(u32p), od.ctypes.data_as(f32p)))
    return orow[:len(r)], od[:len(r)]


def merge_topk(dist, rows, counts):
    """Merge per-shard lists [n_lists, nq, k] -> (dist [nq,k], rows [nq,k], list [nq,k], count [nq])."""
    dist = _f32(dist)
    rows = np.ascontiguousarray(rows, dtype=np.uint32)
    counts = np.ascontiguousarray(counts, dtype=np.uint32)
    n_lists, nq, k = dist.shape
    od = np.empty((nq, k), dtype=np.float32)
    orow = np.empty((nq, k), dtype=np.uint32)
    ol = np.empty((nq, k), dtype=np.uint32)
    oc = np.empty(nq, dtype=np.uint32)
    _check(_ffi.lib().pqv_merge_topk(dist.ctypes.data_as(f32p), rows.ctypes.data_as(u32p),
                                     counts.ctypes.data_as(u32p), n_lists, nq, k,
                                     od.ctypes.data_as(f32p), orow.ctypes.data_as(u32p),
                                     ol.ctypes.data_as(u32p), oc.ctypes.data_as(u32p)))
    return od, orow, ol, oc
