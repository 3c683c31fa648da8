"""ctypes binding of the CPU oracle (oracle/libpqv_oracle.so).

TEST INFRASTRUCTURE ONLY: importable from tests/, __graft_entry__.smoke() and bench.py's
cpu_baseline leg.  The product package (pq_vector_amd) never imports this module.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
ORACLE_DIR = os.path.join(os.path.dirname(_HERE), "oracle")

u8p = C.POINTER(C.c_uint8)
u32p = C.POINTER(C.c_uint32)
u64p = C.POINTER(C.c_uint64)
f32p = C.POINTER(C.c_float)
f64p = C.POINTER(C.c_double)


class PqoRng(C.Structure):
    _fields_ = [("key", C.c_uint32 * 8), ("counter", C.c_uint64),
                ("buf", C.c_uint32 * 64), ("index", C.c_uint32)]


class PqoIndex(C.Structure):
    _fields_ = [("dim", C.c_uint32), ("n_clusters", C.c_uint32), ("centroids", f32p),
                ("list_off", u64p), ("list_rows", u32p)]


def build_oracle(target="all"):
    subprocess.check_call(["make", "-s", "-C", ORACLE_DIR, target])


def load(native=False, fast=False):
    name = "libpqv_oracle_fast.so" if fast else "libpqv_oracle_native.so" if native else "libpqv_oracle.so"
    path = os.path.join(ORACLE_DIR, name)
    if not os.path.exists(path):
        build_oracle("fast" if fast else "native" if native else "all")
    lib = C.CDLL(path)
    P = C.POINTER
    lib.pqo_squared_l2_ref4.restype = C.c_float
    lib.pqo_squared_l2_ref4.argtypes = [f32p, f32p, C.c_size_t]
    lib.pqo_squared_l2_seq.restype = C.c_float
    lib.pqo_squared_l2_seq.argtypes = [f32p, f32p, C.c_size_t]
    lib.pqo_squared_l2_seq_f64.restype = C.c_float
    lib.pqo_squared_l2_seq_f64.argtypes = [f64p, f32p, C.c_size_t]
    lib.pqo_chacha_block.argtypes = [u32p, C.c_uint64, C.c_uint64, C.c_int, u32p]
    lib.pqo_rng_seed_from_u64.argtypes = [P(PqoRng), C.c_uint64]
    lib.pqo_rng_next_u32.restype = C.c_uint32
    lib.pqo_rng_next_u32.argtypes = [P(PqoRng)]
    lib.pqo_rng_next_u64.restype = C.c_uint64
    lib.pqo_rng_next_u64.argtypes = [P(PqoRng)]
    lib.pqo_rng_gen_range_usize.restype = C.c_uint64
    lib.pqo_rng_gen_range_usize.argtypes = [P(PqoRng), C.c_uint64, C.c_uint64]
    lib.pqo_rng_gen_range_u32_incl.restype = C.c_uint32
    lib.pqo_rng_gen_range_u32_incl.argtypes = [P(PqoRng), C.c_uint32, C.c_uint32]
    lib.pqo_rng_gen_range_f32_unit.restype = C.c_float
    lib.pqo_rng_gen_range_f32_unit.argtypes = [P(PqoRng)]
    lib.pqo_rng_gen_f32.restype = C.c_float
    lib.pqo_rng_gen_f32.argtypes = [P(PqoRng)]
    lib.pqo_index_sample.restype = C.c_int
    lib.pqo_index_sample.argtypes = [P(PqoRng), C.c_uint64, C.c_uint64, u64p, P(C.c_int)]
    lib.pqo_index_free.argtypes = [P(PqoIndex)]
    lib.pqo_build_ivf_index.restype = C.c_int
    lib.pqo_build_ivf_index.argtypes = [f32p, C.c_uint64, C.c_uint32, C.c_uint32, C.c_uint32,
                                        C.c_uint64, C.c_uint32, P(P(PqoIndex)), C.c_char_p]
    lib.pqo_kmeans.restype = C.c_int
    lib.pqo_kmeans.argtypes = [f32p, C.c_uint64, C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint64,
                               C.c_uint32, f32p, u64p, u32p]
    lib.pqo_index_to_bytes.restype = C.c_int
    lib.pqo_index_to_bytes.argtypes = [P(PqoIndex), P(u8p), P(C.c_size_t)]
    lib.pqo_index_from_bytes.restype = C.c_int
    lib.pqo_index_from_bytes.argtypes = [C.c_char_p, C.c_size_t, P(P(PqoIndex)), C.c_char_p]
    lib.pqo_find_closest_centroids.restype = C.c_uint32
    lib.pqo_find_closest_centroids.argtypes = [P(PqoIndex), f32p, C.c_uint32, u32p]
    lib.pqo_candidate_rows.restype = C.c_int
    lib.pqo_candidate_rows.argtypes = [P(PqoIndex), f32p, C.c_uint32, P(u32p), u64p]
    lib.pqo_topk_ivf.restype = C.c_int
    lib.pqo_topk_ivf.argtypes = [P(PqoIndex), f32p, f32p, C.c_uint32, C.c_uint32, C.c_uint32,
                                 u32p, f32p, u32p, u64p, C.c_char_p]
    lib.pqo_topk_ivf_batch.restype = C.c_int
    lib.pqo_topk_ivf_batch.argtypes = [P(PqoIndex), f32p, f32p, C.c_uint32, C.c_uint32,
                                       C.c_uint32, u32p, f32p, u32p, u64p]
    lib.pqo_topk_df.restype = C.c_int
    lib.pqo_topk_df.argtypes = [f32p, C.c_uint32, u32p, C.c_uint64, f32p, C.c_uint32, u32p,
                                f32p, u32p]
    lib.pqo_candidate_cursor_take.restype = C.c_uint64
    lib.pqo_candidate_cursor_take.argtypes = [P(u32p), u64p, C.c_uint32, C.c_uint64, u32p, u32p]
    return lib


_libc = C.CDLL(None)
_libc.free.argtypes = [C.c_void_p]


def _f32(a):
    a = np.ascontiguousarray(a, dtype=np.float32)
    return a, a.ctypes.data_as(f32p)


class OracleError(Exception):
    pass


class Oracle:
    """Thin object wrapper: numpy in, numpy out."""

    def __init__(self, native=False, fast=False):
        self.lib = load(native, fast)

    # -- distances ------------------------------------------------------------------
    def l2_ref4(self, a, b):
        a, pa = _f32(a)
        b, pb = _f32(b)
        assert a.size == b.size
        return np.float32(self.lib.pqo_squared_l2_ref4(pa, pb, a.size))

    def l2_seq(self, values, query):
        v, pv = _f32(values)
        q, pq = _f32(query)
        return np.float32(self.lib.pqo_squared_l2_seq(pv, pq, v.size))

    def l2_seq_f64(self, values, query):
        v = np.ascontiguousarray(values, dtype=np.float64)
        q, pq = _f32(query)
        return np.float32(self.lib.pqo_squared_l2_seq_f64(v.ctypes.data_as(f64p), pq, v.size))

    # -- rng ------------------------------------------------------------------------
    def chacha_block(self, key_words, counter, rounds):
        key = (C.c_uint32 * 8)(*key_words)
        out = (C.c_uint32 * 16)()
        self.lib.pqo_chacha_block(key, counter, 0, rounds, out)
        return np.array(out, dtype=np.uint32)

    def rng(self, seed):
        r = PqoRng()
        self.lib.pqo_rng_seed_from_u64(C.byref(r), seed)
        return r

    def index_sample(self, rng, length, amount):
        out = np.zeros(max(amount, 1), dtype=np.uint64)
        br = C.c_int(-1)
        rc = self.lib.pqo_index_sample(C.byref(rng), length, amount, out.ctypes.data_as(u64p),
                                       C.byref(br))
        if rc:
            raise OracleError("sample failed")
        return out[:amount], br.value

    # -- index ----------------------------------------------------------------------
    def build_index(self, data, n_clusters=0, max_iters=20, seed=42, workers=1):
        data = np.ascontiguousarray(data, dtype=np.float32)
        n, dim = data.shape if data.ndim == 2 else (0, 0)
        out = C.POINTER(PqoIndex)()
        err = C.create_string_buffer(128)
        rc = self.lib.pqo_build_ivf_index(data.ctypes.data_as(f32p), n, dim, n_clusters,
                                          max_iters, seed, workers, C.byref(out), err)
        if rc:
            raise OracleError(err.value.decode())
        return OracleIndex(self, out)

    def build_index_raw(self, data_flat, n, dim, **kw):
        """For validation-error cases (n == 0, dim == 0 ...)."""
        data_flat = np.ascontiguousarray(data_flat, dtype=np.float32)
        out = C.POINTER(PqoIndex)()
        err = C.create_string_buffer(128)
        rc = self.lib.pqo_build_ivf_index(data_flat.ctypes.data_as(f32p), n, dim,
                                          kw.get("n_clusters", 0), kw.get("max_iters", 20),
                                          kw.get("seed", 42), kw.get("workers", 1),
                                          C.byref(out), err)
        if rc:
            raise OracleError(err.value.decode())
        return OracleIndex(self, out)

    def kmeans(self, data, k, max_iters=20, seed=42, workers=1):
        data = np.ascontiguousarray(data, dtype=np.float32)
        n, dim = data.shape
        cent = np.zeros((k, dim), dtype=np.float32)
        assign = np.zeros(n, dtype=np.uint64)
        iters = C.c_uint32(0)
        rc = self.lib.pqo_kmeans(data.ctypes.data_as(f32p), n, dim, k, max_iters, seed, workers,
                                 cent.ctypes.data_as(f32p), assign.ctypes.data_as(u64p),
                                 C.byref(iters))
        if rc:
            raise OracleError("kmeans failed")
        return cent, assign, iters.value

    def index_from_bytes(self, blob):
        out = C.POINTER(PqoIndex)()
        err = C.create_string_buffer(128)
        rc = self.lib.pqo_index_from_bytes(bytes(blob), len(blob), C.byref(out), err)
        if rc:
            raise OracleError(err.value.decode())
        return OracleIndex(self, out)

    def index_from_parts(self, dim, centroids, lists):
        """Assemble an index from explicit centroids + inverted lists via the blob format."""
        import struct
        centroids = np.ascontiguousarray(centroids, dtype=np.float32).reshape(-1)
        k = len(lists)
        blob = struct.pack("<II", dim, k) + centroids.tobytes()
        for l in lists:
            l = np.asarray(l, dtype=np.uint32)
            blob += struct.pack("<I", len(l)) + l.tobytes()
        return self.index_from_bytes(blob)

    def topk_df(self, embeddings, rows, query, k):
        emb = np.ascontiguousarray(embeddings, dtype=np.float32)
        rows = np.ascontiguousarray(rows, dtype=np.uint32)
        q, pq = _f32(query)
        out_rows = np.zeros(max(k, 1), dtype=np.uint32)
        out_d2 = np.zeros(max(k, 1), dtype=np.float32)
        nf = C.c_uint32(0)
        self.lib.pqo_topk_df(emb.ctypes.data_as(f32p), emb.shape[1], rows.ctypes.data_as(u32p),
                             rows.size, pq, k, out_rows.ctypes.data_as(u32p),
                             out_d2.ctypes.data_as(f32p), C.byref(nf))
        return out_rows[:nf.value].copy(), out_d2[:nf.value].copy()

    def candidate_cursor_take(self, lists, batch_size):
        arrs = [np.ascontiguousarray(l, dtype=np.uint32) for l in lists]
        ptrs = (u32p * len(arrs))(*[a.ctypes.data_as(u32p) for a in arrs])
        lens = np.array([a.size for a in arrs], dtype=np.uint64)
        of = np.zeros(max(batch_size, 1), dtype=np.uint32)
        orow = np.zeros(max(batch_size, 1), dtype=np.uint32)
        n = self.lib.pqo_candidate_cursor_take(ptrs, lens.ctypes.data_as(u64p), len(arrs),
                                               batch_size, of.ctypes.data_as(u32p),
                                               orow.ctypes.data_as(u32p))
        return list(zip(of[:n].tolist(), orow[:n].tolist()))


class OracleIndex:
    def __init__(self, oracle, ptr):
        self.o = oracle
        self.ptr = ptr

    def __del__(self):
        try:
            if self.ptr:
                self.o.lib.pqo_index_free(self.ptr)
                self.ptr = None
        except Exception:
            pass

    @property
    def dim(self):
        return self.ptr.contents.dim

    @property
    def n_clusters(self):
        return self.ptr.contents.n_clusters

    @property
    def centroids(self):
        c = self.ptr.contents
        return np.ctypeslib.as_array(c.centroids, shape=(c.n_clusters, c.dim)).copy()

    @property
    def list_off(self):
        c = self.ptr.contents
        return np.ctypeslib.as_array(c.list_off, shape=(c.n_clusters + 1,)).copy()

    @property
    def list_rows(self):
        c = self.ptr.contents
        total = int(self.list_off[-1])
        if total == 0:
            return np.zeros(0, dtype=np.uint32)
        return np.ctypeslib.as_array(c.list_rows, shape=(total,)).copy()

    def lists(self):
        off, rows = self.list_off, self.list_rows
        return [rows[int(off[i]):int(off[i + 1])] for i in range(self.n_clusters)]

    def to_bytes(self):
        buf = u8p()
        n = C.c_size_t(0)
        rc = self.o.lib.pqo_index_to_bytes(self.ptr, C.byref(buf), C.byref(n))
        if rc:
            raise OracleError("to_bytes failed")
        out = C.string_at(buf, n.value)
        _libc.free(buf)
        return out

    def find_closest_centroids(self, query, nprobe):
        q, pq = _f32(query)
        out = np.zeros(max(min(nprobe, self.n_clusters), 1), dtype=np.uint32)
        n = self.o.lib.pqo_find_closest_centroids(self.ptr, pq, nprobe, out.ctypes.data_as(u32p))
        return out[:n].copy()

    def candidate_rows(self, query, nprobe):
        q, pq = _f32(query)
        rows = u32p()
        n = C.c_uint64(0)
        self.o.lib.pqo_candidate_rows(self.ptr, pq, nprobe, C.byref(rows), C.byref(n))
        out = np.ctypeslib.as_array(rows, shape=(n.value,)).copy() if n.value else np.zeros(0, np.uint32)
        _libc.free(rows)
        return out

    def topk(self, embeddings, query, k, nprobe):
        emb = np.ascontiguousarray(embeddings, dtype=np.float32)
        q, pq = _f32(query)
        rows = np.zeros(max(k, 1), dtype=np.uint32)
        dist = np.zeros(max(k, 1), dtype=np.float32)
        nf = C.c_uint32(0)
        nc = C.c_uint64(0)
        err = C.create_string_buffer(128)
        rc = self.o.lib.pqo_topk_ivf(self.ptr, emb.ctypes.data_as(f32p), pq, q.size, k, nprobe,
                                     rows.ctypes.data_as(u32p), dist.ctypes.data_as(f32p),
                                     C.byref(nf), C.byref(nc), err)
        if rc:
            raise OracleError(err.value.decode())
        return rows[:nf.value].copy(), dist[:nf.value].copy(), nc.value

    def topk_batch(self, embeddings, queries, k, nprobe):
        emb = np.ascontiguousarray(embeddings, dtype=np.float32)
        qs = np.ascontiguousarray(queries, dtype=np.float32)
        nq = qs.shape[0]
        rows = np.zeros((nq, k), dtype=np.uint32)
        dist = np.zeros((nq, k), dtype=np.float32)
        nf = np.zeros(nq, dtype=np.uint32)
        nc = np.zeros(nq, dtype=np.uint64)
        rc = self.o.lib.pqo_topk_ivf_batch(self.ptr, emb.ctypes.data_as(f32p),
                                           qs.ctypes.data_as(f32p), nq, k, nprobe,
                                           rows.ctypes.data_as(u32p), dist.ctypes.data_as(f32p),
                                           nf.ctypes.data_as(u32p), nc.ctypes.data_as(u64p))
        if rc:
            raise OracleError("topk batch failed")
        return rows, dist, nf, nc
