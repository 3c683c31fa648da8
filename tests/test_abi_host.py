"""CPU tests of the product library's host side: it loads, exports every symbol
include/pqv.h declares, refuses to compute without a GPU, and its host-only logic (blob
format, shard merge) is correct.  No compute entry point is exercised here."""
import ctypes as C
import os
import re
import struct
import subprocess

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def lib():
    so = os.path.join(ROOT, "pq-vector_amd", "libpqv_hip.so")
    if not os.path.exists(so):
        subprocess.check_call(["make", "-s", "-C", os.path.join(ROOT, "pq-vector_amd", "csrc")])
    from pq_vector_amd import _ffi
    return _ffi.lib()


def _declared_symbols():
    text = open(os.path.join(ROOT, "include", "pqv.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(pqv_[a-z0-9_]+)\s*\(", text)))


def test_every_declared_symbol_is_exported_and_bound(lib):
    from pq_vector_amd import _ffi
    declared = _declared_symbols()
    assert len(declared) >= 35
    raw = C.CDLL(_ffi.LIB_PATH)
    for name in declared:
        assert hasattr(raw, name), f"{name} declared in include/pqv.h but not exported"
        assert name in _ffi.SIGNATURES, f"{name} has no ctypes signature"
    assert sorted(_ffi.SIGNATURES) == declared


def test_abi_version(lib):
    assert lib.pqv_abi_version() == 101


def test_no_cpu_fallback(lib):
    """Without a HIP device every compute entry point fails loudly (never a CPU path)."""
    import pq_vector_amd as pqv
    if pqv.device_count() > 0:
        pytest.skip("a GPU is present")
    with pytest.raises(pqv.PqvError) as e:
        pqv.Corpus.upload(np.zeros((4, 4), np.float32))
    assert e.value.code == -2 and "no CPU fallback" in e.value.message
    with pytest.raises(pqv.PqvError) as e:
        pqv.IndexBuilder(np.zeros((4, 4), np.float32)).build()
    assert e.value.code == -2
    with pytest.raises(pqv.PqvError) as e:
        pqv.rerank_batch([0, 0], np.zeros((1, 2), np.float32), 1)
    assert e.value.code == -2


def test_validation_happens_before_device_use(lib):
    """The reference's argument errors surface even on a machine without a GPU."""
    import pq_vector_amd as pqv
    with pytest.raises(pqv.PqvError, match="max_iters must be > 0"):
        pqv.IndexBuilder(np.zeros((4, 4), np.float32)).max_iters(0).build()
    with pytest.raises(pqv.PqvError, match="n_clusters must be > 0"):
        pqv.IndexBuilder(np.zeros((4, 4), np.float32)).n_clusters(0).build()
    with pytest.raises(pqv.PqvError, match="Cannot build IVF index with zero vectors"):
        pqv.IndexBuilder(np.zeros((0, 4), np.float32)).build()
    with pytest.raises(pqv.PqvError, match="Embedding column name cannot be empty"):
        pqv.IndexBuilder(np.zeros((4, 4), np.float32), "  ").build()


def test_blob_format_host_side(lib, oracle):
    """IvfIndex::to_bytes/from_bytes (src/ivf/index.rs:65-128): the product's byte image is
    the reference test's (index.rs:496-511) and equals the oracle's on random indexes."""
    import pq_vector_amd as pqv
    idx = pqv.Index.from_parts(3, [1, 2, 3, 4, 5, 6], [[0, 2, 4], [1, 3]])
    expect = (struct.pack("<II", 3, 2) + struct.pack("<6f", 1, 2, 3, 4, 5, 6)
              + struct.pack("<IIII", 3, 0, 2, 4) + struct.pack("<III", 2, 1, 3))
    assert idx.to_bytes() == expect
    back = pqv.Index.from_bytes(expect)
    assert back.dim == 3 and back.n_clusters == 2 and back.n_rows == 5
    assert [l.tolist() for l in back.inverted_lists()] == [[0, 2, 4], [1, 3]]
    assert (back.centroids.reshape(-1) == np.arange(1, 7, dtype=np.float32)).all()

    rng = np.random.default_rng(0)
    oidx = oracle.build_index(rng.random((500, 12), dtype=np.float32), n_clusters=7)
    blob = oidx.to_bytes()
    p = pqv.Index.from_bytes(blob)
    assert p.to_bytes() == blob
    assert (p.list_offsets == oidx.list_off).all() and (p.list_rows == oidx.list_rows).all()

    with pytest.raises(pqv.PqvError, match="IVF index buffer too small"):
        pqv.Index.from_bytes(b"\x01\x02\x03")
    with pytest.raises(pqv.PqvError, match="Embedding dimension must be > 0"):
        pqv.Index.from_bytes(struct.pack("<II", 0, 2))
    with pytest.raises(pqv.PqvError, match="Cluster count must be > 0"):
        pqv.Index.from_bytes(struct.pack("<II", 2, 0))
    with pytest.raises(pqv.PqvError, match="truncated"):
        pqv.Index.from_bytes(blob[:-3])
    with pytest.raises(pqv.PqvError, match="truncated"):
        pqv.Index.from_bytes(blob[:40])


def test_merge_topk_host(lib):
    """Multi-file merge (exec.rs:264-267 semantics): ascending by (distance, list, position)."""
    import pq_vector_amd as pqv
    inf = np.inf
    dist = np.array([[[1.0, 3.0, 5.0]], [[1.0, 2.0, inf]]], np.float32)       # [2 lists, 1 q, k 3]
    rows = np.array([[[10, 11, 12]], [[20, 21, 0xFFFFFFFF]]], np.uint32)
    counts = np.array([[3], [2]], np.uint32)
    d, r, l, c = pqv.merge_topk(dist, rows, counts)
    assert d[0].tolist() == [1.0, 1.0, 2.0] and r[0].tolist() == [10, 20, 21]
    assert l[0].tolist() == [0, 1, 1] and c[0] == 3
    d, r, l, c = pqv.merge_topk(dist[:, :, :3], rows, np.array([[1], [0]], np.uint32))
    assert c[0] == 1 and r[0, 0] == 10 and r[0, 1] == 0xFFFFFFFF and np.isinf(d[0, 1])


def test_product_does_not_link_or_import_the_oracle():
    """The oracle is test infrastructure: nothing under pq-vector_amd/ or include/ may
    reference it (a product path routed through it would void every parity claim)."""
    bad = []
    for base in ("pq-vector_amd", "include"):
        for dp, _, files in os.walk(os.path.join(ROOT, base)):
            for f in files:
                if f.endswith((".py", ".cpp", ".hip", ".h", ".hpp", "Makefile")):
                    txt = open(os.path.join(dp, f), errors="ignore").read()
                    if re.search(r"pqv_oracle|oracle_binding|libpqv_oracle|pqo_", txt):
                        bad.append(os.path.join(dp, f))
    assert not bad, bad
    so = os.path.join(ROOT, "pq-vector_amd", "libpqv_hip.so")
    if os.path.exists(so):
        out = subprocess.run(["ldd", so], capture_output=True, text=True).stdout
        assert "oracle" not in out


def test_candidate_cursor_matches_reference_semantics_and_oracle(oracle):
    """pqv_candidate_cursor_* (host-side, no device): src/df_vector/access.rs:214-242 -- the reference-derived cases of
    tests/test_oracle_golden.py plus random lists against the oracle's restatement, including the round-robin position
    carried across batches (the oracle takes one batch, so successive product batches must concatenate to it)."""
    import numpy as np
    import pq_vector_amd as pqv
    c = pqv.CandidateCursor(3)
    for i, l in enumerate([[10, 11, 12], [20], [30, 31]]):
        c.add_candidates(i, l)
    got, taken = c.next_batch(5)
    assert got == [(0, 10), (1, 20), (2, 30), (0, 11), (2, 31)] and taken.tolist() == [2, 1, 2]
    got, taken = c.next_batch(5)
    assert got == [(0, 12)] and taken.tolist() == [3, 1, 2]
    assert c.next_batch(0)[0] == [] and pqv.CandidateCursor(0).next_batch(4)[0] == []
    c = pqv.CandidateCursor(2)
    c.add_candidates(0, [1]); c.add_candidates(5, [9, 9])          # out-of-range index: ignored (access.rs:210)
    assert c.next_batch(10)[0] == [(0, 1)]
    rng = np.random.default_rng(4)
    for _ in range(50):
        nf = int(rng.integers(1, 7))
        lists = [rng.integers(0, 1 << 32, size=int(rng.integers(0, 40)), dtype=np.uint64).astype(np.uint32).tolist() for _ in range(nf)]
        total = sum(len(l) for l in lists)
        cap = int(rng.integers(0, total + 5))
        want = oracle.candidate_cursor_take(lists, cap)
        c = pqv.CandidateCursor(nf)
        for i, l in enumerate(lists):
            c.add_candidates(i, l)
        got, taken = c.next_batch(cap)
        assert got == want
        # a file's share is a prefix of its list
        for f in range(nf):
            assert [r for (ff, r) in got if ff == f] == lists[f][:int(taken[f])]
        # split into two batches: same sequence
        c2 = pqv.CandidateCursor(nf)
        for i, l in enumerate(lists):
            c2.add_candidates(i, l)
        a = int(rng.integers(0, cap + 1))
        g1, _ = c2.next_batch(a)
        g2, _ = c2.next_batch(cap - a)
        assert len(g1) + len(g2) == len(want) and sorted(g1 + g2) == sorted(want)


def test_shard_row_groups_abi_contract():
    """pqv_shard_row_groups (host only): ranges tile the file, row bases are prefix sums, invalid arguments are refused with a text."""
    import ctypes as C
    import numpy as np
    from pq_vector_amd import _ffi
    L = _ffi.lib()
    rows = np.array([1000] * 10 + [37], dtype=np.uint64)
    out = [C.c_uint32(0), C.c_uint32(0), C.c_uint64(0), C.c_uint64(0)]

    def call(rank, world, arr=rows):
        rc = L.pqv_shard_row_groups(arr.ctypes.data_as(_ffi.u64p) if len(arr) else None, len(arr), rank, world, *[C.byref(o) for o in out])
        return rc, tuple(o.value for o in out)
    assert call(0, 1) == (0, (0, 11, 0, 10037))
    assert [call(r, 2)[1] for r in range(2)] == [(0, 5, 0, 5000), (5, 11, 5000, 5037)]
    got = [call(r, 16)[1] for r in range(16)]
    assert got[0][0] == 0 and got[-1][1] == 11 and all(got[i][1] == got[i + 1][0] for i in range(15)) and sum(g[3] for g in got) == 10037
    assert any(g[3] == 0 for g in got)                       # more shards than row groups: empty ranges
    assert call(0, 3, np.zeros(0, np.uint64)) == (0, (0, 0, 0, 0))
    assert call(2, 2)[0] < 0 and b"rank" in L.pqv_last_error()
    assert call(0, 0)[0] < 0
    assert L.pqv_shard_row_groups(rows.ctypes.data_as(_ffi.u64p), len(rows), 0, 1, None, None, None, None) < 0
