"""CPU tests: the oracle against every known answer the reference's own tests hold for this
path, the public ChaCha vectors, and the committed golden fixtures."""
import glob
import os
import struct

import numpy as np
import pytest

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


# ---- reference known answers ---------------------------------------------------------------
def test_squared_l2_known_answer(oracle):
    """src/ivf/index.rs:487-493"""
    assert abs(float(oracle.l2_ref4([1, 2, 3], [4, 5, 6])) - 27.0) < 1e-6
    assert float(oracle.l2_seq([4, 5, 6], [1, 2, 3])) == 27.0


def test_index_serialization_round_trip_and_bytes(oracle):
    """src/ivf/index.rs:495-511 + the byte image derivable from to_bytes (:65-83)."""
    idx = oracle.index_from_parts(3, [1, 2, 3, 4, 5, 6], [[0, 2, 4], [1, 3]])
    blob = idx.to_bytes()
    expect = (struct.pack("<II", 3, 2) + struct.pack("<6f", 1, 2, 3, 4, 5, 6)
              + struct.pack("<IIII", 3, 0, 2, 4) + struct.pack("<III", 2, 1, 3))
    assert blob == expect and len(blob) == 60
    back = oracle.index_from_bytes(blob)
    assert back.dim == 3 and back.n_clusters == 2
    assert (back.centroids.reshape(-1) == np.array([1, 2, 3, 4, 5, 6], np.float32)).all()
    assert [l.tolist() for l in back.lists()] == [[0, 2, 4], [1, 3]]
    with pytest.raises(Exception, match="IVF index buffer too small"):
        oracle.index_from_bytes(b"\x00" * 7)


FIX_A = np.array([(0, 0), (1, 0), (0, 2), (5, 5), (2, 2), (0.1, 0.1)], np.float32)
FIX_B = np.array([(0, 0), (.05, .05), (.2, .2), (1, 1), (1.1, 1.1), (1.4, 1.4)], np.float32)


@pytest.mark.parametrize("vecs,min_id,expect,fetched", [(FIX_A, 2, [5, 2], 4), (FIX_B, 3, [3, 4], 3)])
def test_sql_fixtures(oracle, vecs, min_id, expect, fetched):
    """src/df_vector/tests.rs:16-104 and :151-241 with the snapshot counters: default
    n_clusters = ceil(sqrt(6)) = 3 < nprobe 64 => every row is a candidate (RNG-independent);
    the filter runs inside the scan, so rows reach the heap in file order."""
    idx = oracle.build_index(vecs)                       # IndexBuilder defaults
    assert idx.n_clusters == 3
    cand = idx.candidate_rows([0, 0], 64)
    assert len(cand) == 6 and sorted(cand.tolist()) == list(range(6))   # candidate_rows: 6
    scan = sorted(r for r in cand.tolist() if r >= min_id)
    assert len(scan) == fetched                                         # embeddings_fetched
    rows, d2 = oracle.topk_df(vecs, scan, [0, 0], 2)
    assert rows.tolist() == expect


def test_fixture_a_distances(oracle):
    """d2 = 0.02, 4, 8, 50 for ids 5, 2, 4, 3."""
    rows, d2 = oracle.topk_df(FIX_A, [2, 3, 4, 5], [0, 0], 4)
    assert rows.tolist() == [5, 2, 4, 3]
    assert np.allclose(d2, [0.02, 4, 8, 50])


def test_inplace_fixture(oracle):
    """src/ivf/parquet.rs:638-659: 3 rows x 2-D => dim 2 (column name is host-side glue)."""
    idx = oracle.build_index(np.array([(0, 0), (1, 0), (0, 2)], np.float32))
    assert idx.dim == 2 and idx.n_clusters == 2 and int(idx.list_off[-1]) == 3


def test_validation_texts(oracle):
    with pytest.raises(Exception, match="Cannot build IVF index with zero vectors"):
        oracle.build_index_raw(np.zeros(0, np.float32), 0, 4)
    with pytest.raises(Exception, match="n_clusters cannot exceed number of vectors"):
        oracle.build_index(np.zeros((3, 2), np.float32), n_clusters=4)
    with pytest.raises(Exception, match="Embedding dimension must be > 0"):
        oracle.build_index_raw(np.zeros(4, np.float32), 4, 0)
    idx = oracle.build_index(FIX_A)
    with pytest.raises(Exception, match="Query dimension mismatch: expected 2, got 3"):
        idx.topk(FIX_A, [0, 0, 0], 1, 1)
    with pytest.raises(Exception, match="k must be > 0"):
        idx.topk(FIX_A, [0, 0], 0, 1)
    with pytest.raises(Exception, match="nprobe must be > 0"):
        idx.topk(FIX_A, [0, 0], 1, 0)


# ---- RNG (third-party crates; public vectors only) ------------------------------------------
def test_chacha_public_vectors(oracle):
    c20 = oracle.chacha_block([0] * 8, 0, 20).tobytes().hex()
    assert c20.startswith("76b8e0ada0f13d90405d6ae55386bd28bdd219b8a08ded1aa836efcc8b770dc7")
    c12 = oracle.chacha_block([0] * 8, 0, 12).tobytes().hex()
    assert c12 == ("9bf49a6a0755f953811fce125f2683d50429c3bb49e074147e0089a52eae155f"
                   "0564f879d27ae3c02ce82834acfa8c793a629f2ca0de6919610be82f411326be")


def test_seed_expansion_and_stream(oracle):
    """seed 42 => key a48fa17b..., first words 0x222724a2 ... (derived from the
    published PCG32 + ChaCha12 definitions; not checkable against Rust here)."""
    r = oracle.rng(42)
    assert bytes(np.array(r.key, np.uint32).tobytes()).hex() == \
        "a48fa17b58323d0aeab8a1cc690114b82b8cc87518b4f7548d446ea1e4df20f2"
    got = [oracle.lib.pqo_rng_next_u32(r) for _ in range(4)]
    assert got == [0x222724A2, 0x86CC7763, 0x3FAD517D, 0x8AF00A13]
    # next_u64 straddling a refill: low half = last word, high half = first word of next buffer
    r = oracle.rng(1)
    words = [oracle.lib.pqo_rng_next_u32(r) for _ in range(64)]
    nxt = [oracle.lib.pqo_rng_next_u32(r) for _ in range(2)]
    r2 = oracle.rng(1)
    for _ in range(63):
        oracle.lib.pqo_rng_next_u32(r2)
    v = oracle.lib.pqo_rng_next_u64(r2)
    assert v == (nxt[0] << 32) | words[63]
    assert oracle.lib.pqo_rng_next_u32(r2) == nxt[1]


@pytest.mark.parametrize("length,amount,branch", [
    (6, 3, 0), (3, 2, 0),                   # reference test sizes => floyd
    (1_000_000, 50_000, 1),                 # C2 sample => inplace
    (10_000_000, 100_000, 2),               # C3/C4/C5 sample => rejection (u32)
    (100_000, 50_000, 1),                   # k-means++ subset of a 100k sample => inplace
    (1000, 20, 0), (1000, 60, 1), (600_000, 100, 0), (600_000, 162, 0), (600_000, 163, 2), (40_000, 163, 1),
    (10_000_000, 163, 2),
])
def test_index_sample_branches(oracle, length, amount, branch):
    s, br = oracle.index_sample(oracle.rng(42), length, amount)
    assert br == branch
    assert len(set(s.tolist())) == amount and int(s.max()) < length
    s2, _ = oracle.index_sample(oracle.rng(42), length, amount)
    assert (s == s2).all()


def test_uniform_float_ranges(oracle):
    r = oracle.rng(7)
    v = np.array([oracle.lib.pqo_rng_gen_f32(r) for _ in range(2000)], np.float32)
    assert v.min() >= 0 and v.max() < 1
    assert ((v * (1 << 24)) == np.round(v * (1 << 24))).all()      # 24-bit resolution
    t = np.array([oracle.lib.pqo_rng_gen_range_f32_unit(r) for _ in range(2000)], np.float32)
    assert t.min() >= 0 and t.max() < 1
    assert ((t * (1 << 23)) == np.round(t * (1 << 23))).all()      # 23-bit resolution


# ---- std::collections::BinaryHeap emulation: independent pure-Python restatement ------------
class _PyBinaryHeap:
    """Rust std BinaryHeap<T> (max-heap) with T ordered by `key` only; sift_up stops on <=,
    pop = swap_remove(0) + sift_down_to_bottom + sift_up (library/alloc binary_heap)."""

    def __init__(self):
        self.d = []

    def push(self, item):
        self.d.append(item)
        self._sift_up(0, len(self.d) - 1)

    def pop(self):
        item = self.d.pop()
        if self.d:
            item, self.d[0] = self.d[0], item
            self._sift_down_to_bottom(0)
        return item

    def _sift_up(self, start, pos):
        elt = self.d[pos]
        while pos > start:
            parent = (pos - 1) // 2
            if elt[0] <= self.d[parent][0]:
                break
            self.d[pos] = self.d[parent]
            pos = parent
        self.d[pos] = elt
        return pos

    def _sift_down_to_bottom(self, pos):
        end, start = len(self.d), pos
        elt = self.d[pos]
        child = 2 * pos + 1
        while child <= max(end - 2, 0) and end >= 2:
            if self.d[child][0] <= self.d[child + 1][0]:
                child += 1
            self.d[pos] = self.d[child]
            pos = child
            child = 2 * pos + 1
        if child == end - 1:
            self.d[pos] = self.d[child]
            pos = child
        self.d[pos] = elt
        self._sift_up(start, pos)


def _py_topk(dists, ids, k):
    h = _PyBinaryHeap()
    for d, i in zip(dists, ids):
        if len(h.d) < k:
            h.push((d, i))
        elif d < h.d[0][0]:
            h.pop()
            h.push((d, i))
    out = list(h.d)                                   # into_iter(): backing-array order
    out.sort(key=lambda t: t[0])                      # stable
    return [i for _, i in out], [d for d, _ in out]


@pytest.mark.parametrize("seed", range(6))
def test_heap_emulation_matches_python_restatement(oracle, seed):
    rng = np.random.default_rng(seed)
    n, dim, k = 400, 4, 7
    emb = rng.integers(0, 3, size=(n, dim)).astype(np.float32)     # tie-heavy
    q = rng.integers(0, 3, size=dim).astype(np.float32)
    rows = rng.permutation(n).astype(np.uint32)
    got_rows, got_d2 = oracle.topk_df(emb, rows, q, k)
    d = [float(oracle.l2_seq(emb[r], q)) for r in rows]
    want_rows, want_d = _py_topk(d, rows.tolist(), k)
    assert got_rows.tolist() == want_rows
    assert got_d2.tolist() == want_d


def test_candidate_cursor_round_robin(oracle):
    """src/df_vector/access.rs:214-242: round-robin over files until the cap."""
    got = oracle.candidate_cursor_take([[10, 11, 12], [20], [30, 31]], 5)
    assert got == [(0, 10), (1, 20), (2, 30), (0, 11), (2, 31)]
    assert oracle.candidate_cursor_take([[1, 2, 3]], 2) == [(0, 1), (0, 2)]
    assert oracle.candidate_cursor_take([[1], []], 10) == [(0, 1)]
    assert oracle.candidate_cursor_take([[1]], 0) == []


def test_kmeans_properties(oracle):
    """Empty clusters stay all-zero (index.rs:436,446-453); lists are ascending and cover
    every row once; `workers` only moves the k-means++ f32 total (F8)."""
    rng = np.random.default_rng(3)
    data = np.tile(rng.random((2, 4), dtype=np.float32), (40, 1))      # 2 distinct points
    idx = oracle.build_index(data, n_clusters=5, workers=2)
    sizes = np.diff(idx.list_off)
    assert (sizes > 0).sum() <= 2
    assert (idx.centroids[sizes == 0] == 0).all()
    rows = idx.list_rows
    assert sorted(rows.tolist()) == list(range(80))
    for l in idx.lists():
        assert (np.diff(l.astype(np.int64)) > 0).all()


# ---- committed golden fixtures: the oracle must reproduce them -----------------------------
@pytest.mark.parametrize("path", sorted(glob.glob(os.path.join(GOLDEN, "*.npz"))),
                         ids=lambda p: os.path.basename(p))
def test_oracle_reproduces_golden(oracle, path):
    g = np.load(path)
    data, queries = g["data"], g["queries"]
    k, nprobe = int(g["k"]), int(g["nprobe"])
    for w in g["workers_list"].tolist():
        idx = oracle.build_index(data, n_clusters=int(g["n_clusters"]), max_iters=int(g["max_iters"]),
                                 seed=int(g["seed"]), workers=w)
        assert idx.to_bytes() == g[f"w{w}_blob"].tobytes()
        rows, dist, nf, nc = idx.topk_batch(data, queries, k, nprobe)
        assert (rows == g[f"w{w}_topk_rows"]).all()
        assert (dist.view(np.uint32) == g[f"w{w}_topk_dist_bits"]).all()
        assert (nf == g[f"w{w}_n_found"]).all() and (nc == g[f"w{w}_n_candidates"]).all()


# ---------------------------------------------------------------------------------------
# rand 0.8.5 value-stability vectors [recalled from rand's own test-suite, rngs/std.rs::test_stdrng_construction]:
#   StdRng::from_seed([1,0,0,0, 23,0,0,0, 200,1,0,0, 210,30,0,0, 0 x 16]).next_u64() == 10719222850664546238
#   StdRng::from_rng(that rng).next_u64()                                            == 14064965282130556830
# (from_rng fills a fresh 32-byte seed from the parent's word stream).  The second value was recalled independently of
# the implementation and reproduced by it at first try, which pins ChaCha12, the seed layout, the u32 / u64 word order
# and the buffer walk together.  Both restatements -- the oracle's C and the product's csrc/rng.hpp -- must give them.
# ---------------------------------------------------------------------------------------
_STD_SEED = bytes([1, 0, 0, 0, 23, 0, 0, 0, 200, 1, 0, 0, 210, 30, 0, 0] + [0] * 16)
_STD_TARGET = (10719222850664546238, 14064965282130556830)


def _oracle_rng_from_seed(oracle, seed):
    from oracle_binding import PqoRng
    import ctypes as C
    oracle.lib.pqo_rng_from_seed.argtypes = [C.POINTER(PqoRng), C.POINTER(C.c_uint8)]
    r = PqoRng()
    oracle.lib.pqo_rng_from_seed(C.byref(r), (C.c_uint8 * 32)(*seed))
    return r


def test_stdrng_value_stability_oracle(oracle):
    r = _oracle_rng_from_seed(oracle, _STD_SEED)
    assert oracle.lib.pqo_rng_next_u64(r) == _STD_TARGET[0]
    words = [oracle.lib.pqo_rng_next_u32(r) for _ in range(8)]
    child = b"".join(int(w).to_bytes(4, "little") for w in words)
    assert oracle.lib.pqo_rng_next_u64(_oracle_rng_from_seed(oracle, child)) == _STD_TARGET[1]


def _diag(lib, seed32, seed64, mode, arg, n):
    import ctypes as C
    out = (C.c_uint64 * max(1, n))()
    sp = (C.c_uint8 * 32)(*seed32) if seed32 is not None else None
    rc = lib.pqv_diag_rng(sp, seed64, mode, arg, out, n)
    assert rc == 0, lib.pqv_last_error()
    return [int(v) for v in out[:n]]


def test_stdrng_value_stability_product_and_cross_check(oracle):
    """csrc/rng.hpp (the product's C++ restatement) against rand's vectors and, draw for draw, against the oracle's C
    restatement: next_u64 / next_u32 streams, gen_range(usize), gen_range(0.0..1.0) f32 and index::sample in all three
    of its branches (floyd, inplace, rejection).  No GPU involved."""
    from pq_vector_amd import _ffi
    lib = _ffi.lib()
    assert _diag(lib, _STD_SEED, 0, 0, 0, 1)[0] == _STD_TARGET[0]
    w = _diag(lib, _STD_SEED, 0, 1, 0, 10)[2:]                     # words 2..9 follow the first next_u64
    child = b"".join(int(x).to_bytes(4, "little") for x in w)
    assert _diag(lib, child, 0, 0, 0, 1)[0] == _STD_TARGET[1]
    for seed in (0, 1, 42, 1234, 2 ** 64 - 1):
        r = oracle.rng(seed)
        assert _diag(lib, None, seed, 0, 0, 100) == [oracle.lib.pqo_rng_next_u64(r) for _ in range(100)]
        r = oracle.rng(seed)
        assert _diag(lib, None, seed, 1, 0, 131) == [oracle.lib.pqo_rng_next_u32(r) for _ in range(131)]
        for span in (1, 2, 3, 50_000, 10_000_000, 2 ** 40 + 17):
            r = oracle.rng(seed)
            assert _diag(lib, None, seed, 2, span, 64) == [oracle.lib.pqo_rng_gen_range_usize(r, 0, span) for _ in range(64)]
        r = oracle.rng(seed)
        want = np.array([oracle.lib.pqo_rng_gen_range_f32_unit(r) for _ in range(64)], np.float32).view(np.uint32)
        assert _diag(lib, None, seed, 3, 0, 64) == want.tolist()
    for length, amount in ((6, 3), (1000, 20), (1000, 60), (1_000_000, 50_000), (10_000_000, 100_000), (600_000, 163), (100_000, 50_000)):
        s, _ = oracle.index_sample(oracle.rng(42), length, amount)
        assert _diag(lib, None, 42, 4, length, amount) == s.astype(np.uint64).tolist()
