"""GPU parity tests proper: the HIP path (through the C ABI) against the CPU oracle on the
same seeded inputs.  Bar: bit-exact row ids, inverted lists, centroids and distances
(integer/byte work and f32 in the reference's summation order alike)."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _bits(a):
    return np.ascontiguousarray(a, dtype=np.float32).view(np.uint32)


def _assert_topk_equal(got, want, k, boundary_ok=None):
    """Distances must match bit for bit, position by position.  Row ids must match position
    by position too, except inside groups of exactly equal output distance, where the
    reference's order depends on BinaryHeap history: there the id
    sets must match.  One further exception, only when `boundary_ok` is given (tie-heavy
    inputs): if the LAST group is tied with candidates that did not make it, Rust's heap
    keeps whichever tied elements its sift history left off the root -- any tied candidate
    is then acceptable (boundary_ok(q, row, dist) says whether `row` is one)."""
    rows, dist, nf = got
    orows, odist, onf = want
    assert (nf == onf).all()
    for q in range(rows.shape[0]):
        m = int(nf[q])
        assert (_bits(dist[q, :m]) == _bits(odist[q, :m])).all(), f"query {q}: distances differ"
        i = 0
        while i < m:
            j = i
            while j + 1 < m and _bits(odist[q, j + 1:j + 2])[0] == _bits(odist[q, i:i + 1])[0]:
                j += 1
            same = sorted(rows[q, i:j + 1].tolist()) == sorted(orows[q, i:j + 1].tolist())
            if not same and boundary_ok is not None and j == m - 1:
                assert len(set(rows[q, i:j + 1].tolist())) == j + 1 - i
                for r in rows[q, i:j + 1]:
                    assert boundary_ok(q, int(r), dist[q, i]), f"query {q}: row {r} is not a tied candidate"
            else:
                assert same, f"query {q}: ids differ in positions {i}..{j}"
            i = j + 1
        assert (rows[q, m:] == 0xFFFFFFFF).all()


def _random_index(oracle, rng, n, dim, kc, workers=1):
    data = rng.random((n, dim), dtype=np.float32)
    oidx = oracle.build_index(data, n_clusters=kc, workers=workers, max_iters=5)
    return data, oidx


# ---------------------------------------------------------------------------------------
# reference known answers through the GPU path
# ---------------------------------------------------------------------------------------
def test_reference_fixture_ids_5_2(pqv, oracle):
    """src/df_vector/tests.rs:31-39,77-80,99: filter id>=2, k=2 => [5, 2] (exec.rs order)."""
    vecs = np.array([(0, 0), (1, 0), (0, 2), (5, 5), (2, 2), (0.1, 0.1)], np.float32)
    corpus = pqv.Corpus.upload(vecs)
    index = pqv.IndexBuilder(corpus).workers(1).build()
    assert index.n_clusters == 3
    s = pqv.Searcher(index, corpus)
    cand = s.candidate_rows([0, 0], 64)
    assert len(cand) == 6                      # candidate_rows: 6 (snapshot)
    fetched = sorted(r for r in cand.tolist() if r >= 2)
    assert len(fetched) == 4                   # embeddings_fetched: 4 (snapshot)
    rows, d2 = pqv.rerank_finish(pqv.rerank_batch([0, 0], vecs[fetched], 2, ids=fetched))
    assert rows.tolist() == [5, 2]
    assert np.allclose(d2, [0.02, 4.0])


def test_reference_fixture_ids_3_4(pqv):
    """src/df_vector/tests.rs:166-174,212-215,235."""
    vecs = np.array([(0, 0), (.05, .05), (.2, .2), (1, 1), (1.1, 1.1), (1.4, 1.4)], np.float32)
    corpus = pqv.Corpus.upload(vecs)
    index = pqv.IndexBuilder(corpus).workers(1).build()
    s = pqv.Searcher(index, corpus)
    cand = s.candidate_rows([0, 0], 64)
    fetched = sorted(r for r in cand.tolist() if r >= 3)
    assert len(cand) == 6 and len(fetched) == 3
    rows, _ = pqv.rerank_finish(pqv.rerank_batch([0, 0], vecs[fetched], 2, ids=fetched))
    assert rows.tolist() == [3, 4]


def test_reference_l2_known_answer(pqv):
    """src/ivf/index.rs:488-493: d2([1,2,3],[4,5,6]) == 27 through the re-rank kernel."""
    rows, d2 = pqv.rerank_batch([1, 2, 3], np.array([[4, 5, 6]], np.float32), 1,
                                metric=pqv.PQV_L2SQ_REF4)
    assert rows.tolist() == [0] and d2.tolist() == [27.0]


def test_inplace_fixture_dim2(pqv):
    """src/ivf/parquet.rs:638-659: 3 rows x 2-D build => dim 2."""
    vecs = np.array([(0, 0), (1, 0), (0, 2)], np.float32)
    index = pqv.IndexBuilder(vecs, "vec").build()
    assert index.dim == 2 and index.n_rows == 3


# ---------------------------------------------------------------------------------------
# top-k vs oracle
# ---------------------------------------------------------------------------------------
@pytest.mark.parametrize("n,dim,kc,k,nprobe", [
    (3000, 128, 12, 10, 4),     # C2-shaped, CG=32 single chunk
    (2000, 768, 10, 10, 3),     # C3-shaped, CG=64 x 3 chunks
    (1500, 1536, 8, 10, 3),     # C5-shaped
    (600, 4096, 5, 10, 5),      # C1-shaped (vldb stand-in dims)
    (2500, 100, 9, 7, 2),       # G=25: partial chunk
    (2500, 72, 9, 5, 9),        # G=18, nprobe == n_clusters
    (1200, 30, 6, 10, 3),       # dim % 4 == 2: unaligned rows + scalar tail
    (1200, 3, 6, 3, 2),         # dim < 4: tail only
    (900, 1, 4, 2, 4),
    (5000, 64, 20, 64, 6),      # k = 64: a full wave of slots
    (5000, 64, 20, 100, 6),     # reference bench K=100: 4 slots per lane
    (4000, 32, 8, 300, 8),      # 16 slots per lane
    (300, 16, 4, 10, 64),       # nprobe > n_clusters: clamped
    (50, 8, 7, 60, 7),          # k > n: fewer than k results
])
@pytest.mark.parametrize("layout", ["ivf", "row"])
def test_topk_matches_oracle(pqv, oracle, n, dim, kc, k, nprobe, layout):
    rng = np.random.default_rng(n * 31 + dim)
    data, oidx = _random_index(oracle, rng, n, dim, kc)
    queries = rng.random((9, dim), dtype=np.float32)
    corpus = pqv.Corpus.upload(data)
    index = pqv.Index.from_bytes(oidx.to_bytes())
    flags = pqv.PQV_LAYOUT_ROW_ORDER if layout == "row" else pqv.PQV_LAYOUT_IVF_ORDERED
    s = pqv.Searcher(index, corpus, flags)
    rows, dist, nf, nc = s.topk(queries, k, nprobe)
    orows, odist, onf, onc = oidx.topk_batch(data, queries, k, nprobe)
    assert (nc == onc).all()
    _assert_topk_equal((rows, dist, nf), (orows, odist, onf), k)
    # probe order and candidate rows, integer work: exact
    for q in range(3):
        assert (s.probe(queries[q], nprobe) == oidx.find_closest_centroids(queries[q], nprobe)).all()
        assert (s.candidate_rows(queries[q], nprobe) == oidx.candidate_rows(queries[q], nprobe)).all()


def test_topk_ties_integer_vectors(pqv, oracle):
    """Tie-heavy integer-valued vectors.  With tied output distances, which rows survive and in
    what order is an artefact of Rust's BinaryHeap sift history (the oracle emulates it).
    pqv_topk flags such queries on the device and replays them through the same heap mechanics:
    its answer must equal the oracle's position by position, ties included."""
    rng = np.random.default_rng(5)
    data = rng.integers(0, 3, size=(4000, 8)).astype(np.float32)
    oidx = oracle.build_index(data, n_clusters=6, workers=1, max_iters=4)
    corpus = pqv.Corpus.upload(data)
    queries = rng.integers(0, 3, size=(80, 8)).astype(np.float32)
    for mode in ("stream", "tile"):
        import os
        os.environ["PQV_RERANK_MODE"] = mode
        try:
            s = pqv.Searcher(pqv.Index.from_bytes(oidx.to_bytes()), corpus)
        finally:
            del os.environ["PQV_RERANK_MODE"]
        for k in (1, 5, 10, 63, 64, 70):
            for cap in (0, 900):
                rows, dist, nf, _ = s.topk(queries, k, 3, max_candidates=cap)
                if cap:
                    for q in range(len(queries)):
                        cand = oidx.candidate_rows(queries[q], 3)[:cap]
                        orow, od2 = oracle_topk_ref4(oracle, data, cand, queries[q], k)
                        assert (rows[q, :len(orow)] == orow).all() and nf[q] == len(orow)
                        assert (_bits(dist[q, :len(orow)]) == _bits(od2)).all()
                else:
                    orows, odist, onf, _ = oidx.topk_batch(data, queries, k, 3)
                    assert (nf == onf).all()
                    assert (rows == orows).all(), f"mode {mode} k {k}: ids differ from the reference heap order"
                    assert (_bits(dist) == _bits(odist)).all()
        assert s.counters()["exact_replays"] > 0


def oracle_topk_ref4(oracle, data, cand, query, k):
    """search.rs:112-141 over an explicit candidate list (cap applied), REF4 order, sqrt, via a
    one-cluster oracle index whose single list is `cand` in that order."""
    sub = np.ascontiguousarray(data[cand])
    idx = oracle.index_from_parts(data.shape[1], np.zeros((1, data.shape[1]), np.float32),
                                  [np.arange(len(cand), dtype=np.uint32)])
    rows, dist, _ = idx.topk(sub, query, k, 1)
    return cand[rows], dist


def test_topk_seq_metric_and_cap(pqv, oracle):
    """PQV_L2SQ_SEQ (exec.rs:529-533 order, no sqrt) and the max_candidates cap
    (access.rs:214-242: keep the first max_candidates in probe-rank order)."""
    rng = np.random.default_rng(11)
    data, oidx = _random_index(oracle, rng, 3000, 96, 10)
    corpus = pqv.Corpus.upload(data)
    s = pqv.Searcher(pqv.Index.from_bytes(oidx.to_bytes()), corpus)
    queries = rng.random((6, 96), dtype=np.float32)
    for cap in (0, 500, 37):
        rows, d2, nf, nc = s.topk(queries, 10, 4, max_candidates=cap, metric=pqv.PQV_L2SQ_SEQ,
                                  sqrt_out=False)
        for q in range(len(queries)):
            cand = oidx.candidate_rows(queries[q], 4)
            assert nc[q] == len(cand)
            if cap:
                cand = cand[:cap]
            orow, od2 = oracle.topk_df(data, cand, queries[q], 10)
            m = len(orow)
            assert nf[q] == m
            assert (rows[q, :m] == orow).all()
            assert (_bits(d2[q, :m]) == _bits(od2)).all()


def test_rerank_batches_match_update_topk_heap(pqv, oracle):
    """Folding RecordBatches one at a time (exec.rs:264-267) == one heap over all rows."""
    rng = np.random.default_rng(3)
    emb = rng.random((5000, 48), dtype=np.float32)
    q = rng.random(48, dtype=np.float32)
    order = rng.permutation(5000).astype(np.uint32)
    state = None
    for b in range(0, 5000, 2048):      # BATCH_ROWS = 2048 (benches/query.rs:29)
        ids = order[b:b + 2048]
        valid = (rng.random(len(ids)) > 0.1).astype(np.uint8)
        state = pqv.rerank_batch(q, emb[ids], 10, state=state, ids=ids, valid=valid)
        if b == 0:
            kept = [ids[valid == 1]]
        else:
            kept.append(ids[valid == 1])
    orow, od2 = oracle.topk_df(emb, np.concatenate(kept), q, 10)
    got = pqv.rerank_finish(state)
    assert (got[0] == orow).all()
    assert (_bits(got[1]) == _bits(od2)).all()


@pytest.mark.parametrize("k", [1, 3, 10, 37, 100, 1500])
def test_rerank_replays_the_reference_heap_under_ties(pqv, oracle, k):
    """exec.rs:474-481 with tied distances: which rows survive at a tied k-th distance and the order inside groups of
    equal distance are artefacts of BinaryHeap's sift history.  Integer-valued vectors on a coarse grid give hundreds of
    exact ties; the batch-at-a-time fold must equal one reference heap over all rows, id for id -- for any k (the
    selection is the heap's own, so k is not bounded by a kernel-side list), with null rows and a Float64 batch
    (narrowed `as f32`, exec.rs:538-545) in between."""
    rng = np.random.default_rng(100 + k)
    n, dim = 6000, 12
    emb = rng.integers(0, 3, size=(n, dim)).astype(np.float32)
    q = rng.integers(0, 3, size=dim).astype(np.float32)
    order = rng.permutation(n).astype(np.uint32)
    state, kept = None, []
    for bi, b in enumerate(range(0, n, 1024)):
        ids = order[b:b + 1024]
        valid = (rng.random(len(ids)) > 0.05).astype(np.uint8)
        batch = emb[ids].astype(np.float64) if bi % 2 else emb[ids]
        state = pqv.rerank_batch(q, batch, k, state=state, ids=ids, valid=valid)
        kept.append(ids[valid == 1])
    orow, od2 = oracle.topk_df(emb, np.concatenate(kept), q, k)
    got = pqv.rerank_finish(state)
    assert len(got[0]) == len(orow) == min(k, sum(len(x) for x in kept))
    assert (_bits(got[1]) == _bits(od2)).all()
    assert (got[0] == orow).all()
    if k >= 3:
        assert len(np.unique(od2)) < len(od2)          # the case does hold ties


def test_rerank_f64_batches_are_narrowed_like_the_reference(pqv, oracle):
    """exec.rs:538-545: a Float64 values buffer is narrowed value by value (`as f32`) before the subtraction."""
    rng = np.random.default_rng(8)
    emb64 = rng.random((3000, 40)) * 3.0 - 1.0
    q = rng.random(40, dtype=np.float32)
    ids = np.arange(3000, dtype=np.uint32)
    state = pqv.rerank_batch(q, emb64[:2048], 10, ids=ids[:2048])
    state = pqv.rerank_batch(q, emb64[2048:], 10, state=state, ids=ids[2048:])
    orow, od2 = oracle.topk_df(emb64.astype(np.float32), ids, q, 10)
    got = pqv.rerank_finish(state)
    assert (got[0] == orow).all() and (_bits(got[1]) == _bits(od2)).all()


# ---------------------------------------------------------------------------------------
# index build vs oracle
# ---------------------------------------------------------------------------------------
@pytest.mark.parametrize("n,dim,kc,workers", [
    (6, 2, 0, 1),            # ceil(sqrt(6)) = 3, sample == n, floyd never runs
    (400, 16, 0, 8),         # default n_clusters = 20; sample_size clamps to k
    (4000, 128, 16, 8),      # sampled: 200 rows, floyd/inplace branch
    (4000, 128, 16, 1),
    (3000, 30, 12, 3),       # unaligned dim
    (60000, 8, 25, 8),       # sample 3000, init subset == sample
    (1200, 768, 6, 5),
    (40000, 128, 24, 8),     # sample 2000 rows of a multiple of 64 dims: the k-means++ rounds take the f16-screened min-update
    (30000, 192, 12, 3),
    (40000, 128, 24, 40),    # 40 chunk chains: the chunk-transposed mirror of the minima (vector lanes = chunks)
    (50000, 64, 30, 256),    # the GPU box's own worker count: chunks of 10 minima
    (24000, 256, 140, 8),    # ... and >= 128 centroids: the 256 x 256-tile f16 assignment + resolve pass
])
def test_index_build_matches_oracle(pqv, oracle, n, dim, kc, workers):
    rng = np.random.default_rng(n + dim)
    data = rng.random((n, dim), dtype=np.float32)
    oidx = oracle.build_index(data, n_clusters=kc, workers=workers)
    corpus = pqv.Corpus.upload(data)
    b = pqv.IndexBuilder(corpus).workers(workers)
    if kc:
        b = b.n_clusters(kc)
    index = b.build()
    assert index.n_clusters == oidx.n_clusters
    assert (_bits(index.centroids) == _bits(oidx.centroids)).all(), "centroids differ"
    assert (index.list_offsets == oidx.list_off).all()
    assert (index.list_rows == oidx.list_rows).all()
    assert index.to_bytes() == oidx.to_bytes()


@pytest.mark.parametrize("kind", ["offset", "scales", "duplicates"])
def test_kmeans_pp_screen_and_wide_assignment_on_hostile_data(pqv, oracle, kind):
    """The f16 screens of the build (k-means++ min-update, 256 x 256-tile assignment) are bounds, never answers: data far from
    the origin (the centring must carry the bound), rows of wildly different norms, and exact duplicates (zero distances, ties
    at the minimum) must give the oracle's blob."""
    rng = np.random.default_rng(31)
    n, dim, kc = 26000, 128, 130
    if kind == "offset":
        data = (np.float32(-1000.0) + np.float32(0.01) * rng.standard_normal((n, dim), dtype=np.float32)).astype(np.float32)
    elif kind == "scales":
        data = rng.standard_normal((n, dim), dtype=np.float32)
        data[::7] *= np.float32(1.0e4)
        data[3::11] *= np.float32(1.0e-4)
    else:
        base = rng.random((n // 40, dim), dtype=np.float32)
        data = np.repeat(base, 40, axis=0)[:n].copy()
        data[::3] += np.float32(1.0e-3) * rng.random((len(data[::3]), dim), dtype=np.float32)
    oidx = oracle.build_index(data, n_clusters=kc, workers=8)
    index = pqv.IndexBuilder(pqv.Corpus.upload(data)).n_clusters(kc).workers(8).build()
    assert (_bits(index.centroids) == _bits(oidx.centroids)).all(), "centroids differ"
    assert index.to_bytes() == oidx.to_bytes()


def test_kmeans_matches_oracle_with_subset_sampling(pqv, oracle):
    """n > 50_000 rows into k_means: the k-means++ subset is drawn by index::sample
    (index.rs:332-338) and Lloyd runs over all rows."""
    rng = np.random.default_rng(99)
    data = rng.random((52000, 4), dtype=np.float32)
    ocent, oassign, oiters = oracle.kmeans(data, 7, max_iters=6, seed=42, workers=4)
    corpus = pqv.Corpus.upload(data)
    import ctypes as C
    from pq_vector_amd import _ffi
    cent = np.zeros((7, 4), np.float32)
    assign = np.zeros(52000, np.uint32)
    iters = C.c_uint32(0)
    rc = _ffi.lib().pqv_kmeans(corpus._h, 7, 6, 42, 4, cent.ctypes.data_as(_ffi.f32p),
                               assign.ctypes.data_as(_ffi.u32p), C.byref(iters))
    assert rc == 0, _ffi.lib().pqv_last_error()
    assert iters.value == oiters
    assert (_bits(cent) == _bits(ocent)).all()
    assert (assign == oassign.astype(np.uint32)).all()


def test_empty_cluster_and_duplicate_rows(pqv, oracle):
    """Duplicate vectors: k-means++ total reaches 0 -> uniform pick (index.rs:384-389);
    clusters that end up empty keep all-zero centroids (index.rs:436,446-453)."""
    data = np.tile(np.array([[1, 2, 3, 4], [5, 6, 7, 8]], np.float32), (50, 1))
    oidx = oracle.build_index(data, n_clusters=5, workers=2)
    index = pqv.IndexBuilder(data).n_clusters(5).workers(2).build()
    assert index.to_bytes() == oidx.to_bytes()


def test_f64_column_is_narrowed(pqv, oracle):
    """src/ivf/parquet.rs:246-256: Float64 embeddings are cast to f32 element-wise."""
    rng = np.random.default_rng(1)
    d64 = rng.random((700, 24))
    corpus = pqv.Corpus.upload(d64)
    got = corpus.fetch_rows(np.arange(700, dtype=np.uint32))
    assert (_bits(got) == _bits(d64.astype(np.float32))).all()


# ---------------------------------------------------------------------------------------
# validation texts (reference strings)
# ---------------------------------------------------------------------------------------
def test_error_texts(pqv):
    rng = np.random.default_rng(0)
    data = rng.random((10, 4), dtype=np.float32)
    with pytest.raises(pqv.PqvError, match="n_clusters cannot exceed number of vectors"):
        pqv.IndexBuilder(data).n_clusters(11).build()
    with pytest.raises(pqv.PqvError, match="max_iters must be > 0"):
        pqv.IndexBuilder(data).max_iters(0).build()
    with pytest.raises(pqv.PqvError, match="n_clusters must be > 0"):
        pqv.IndexBuilder(data).n_clusters(0).build()
    with pytest.raises(pqv.PqvError, match="Cannot build IVF index with zero vectors"):
        pqv.IndexBuilder(np.zeros((0, 4), np.float32)).build()
    corpus = pqv.Corpus.upload(data)
    s = pqv.Searcher(pqv.IndexBuilder(corpus).n_clusters(2).build(), corpus)
    with pytest.raises(pqv.PqvError, match="Query dimension mismatch: expected 4, got 3"):
        s.topk(np.zeros((1, 3), np.float32), 1, 1)
    with pytest.raises(pqv.PqvError, match="k must be > 0"):
        pqv.TopkBuilder(s, data[0]).k(0)
    with pytest.raises(pqv.PqvError, match="nprobe must be > 0"):
        pqv.TopkBuilder(s, data[0]).nprobe(0)
    with pytest.raises(pqv.PqvError, match="k must be set"):
        pqv.TopkBuilder(s, data[0]).nprobe(1).search()
    with pytest.raises(pqv.PqvError, match="nprobe must be set"):
        pqv.TopkBuilder(s, data[0]).k(1).search()
    res = pqv.TopkBuilder(s, data[3]).k(2).nprobe(2).search()
    assert res[0].row_idx == 3 and res[0].distance == 0.0


# ---------------------------------------------------------------------------------------
# committed golden fixtures (tests/golden/*.npz, generated by tests/golden/make_golden.py)
# ---------------------------------------------------------------------------------------
import glob as _glob
import os as _os

_GOLDEN = _os.path.join(_os.path.dirname(_os.path.abspath(__file__)), "golden")


@pytest.mark.parametrize("path", sorted(_glob.glob(_os.path.join(_GOLDEN, "*.npz"))),
                         ids=lambda p: _os.path.basename(p))
def test_gpu_reproduces_golden(pqv, oracle, path):
    g = np.load(path)
    data, queries = g["data"], g["queries"]
    k, nprobe = int(g["k"]), int(g["nprobe"])
    ties = "ties" in _os.path.basename(path)
    corpus = pqv.Corpus.upload(data)
    for w in g["workers_list"].tolist():
        b = pqv.IndexBuilder(corpus).max_iters(int(g["max_iters"])).seed(int(g["seed"])).workers(w)
        if int(g["n_clusters"]):
            b = b.n_clusters(int(g["n_clusters"]))
        index = b.build()
        assert index.to_bytes() == g[f"w{w}_blob"].tobytes(), "index blob differs from the golden blob"
        s = pqv.Searcher(index, corpus)
        rows, dist, nf, nc = s.topk(queries, k, nprobe)
        assert (nc == g[f"w{w}_n_candidates"]).all()
        for q in range(len(queries)):
            assert (s.probe(queries[q], nprobe) == g[f"w{w}_probe"][q]).all()
        want = (g[f"w{w}_topk_rows"], g[f"w{w}_topk_dist_bits"].view(np.float32), g[f"w{w}_n_found"])
        if ties:      # exact replay: identical to the reference's heap order, ties included
            assert (nf == want[2]).all() and (rows == want[0]).all()
            assert (_bits(dist) == _bits(want[1])).all()
        else:
            _assert_topk_equal((rows, dist, nf), want, k)


# ---------------------------------------------------------------------------------------
# batched cluster-major tile path (tile_rerank_kernel) vs the streaming path vs the oracle
# ---------------------------------------------------------------------------------------
@pytest.mark.parametrize("n,dim,kc,k,nprobe,nq", [
    (6000, 128, 10, 10, 4, 70),      # C2-shaped: ~28 queries per cluster, partial groups
    (3000, 768, 6, 10, 3, 40),       # C3-shaped rows (3 KB)
    (4000, 100, 8, 7, 8, 33),        # G = 25: 3 full 128-B steps + 1 remainder group
    (2500, 30, 5, 10, 2, 50),        # unaligned rows + scalar tail
    (1500, 3, 4, 3, 4, 20),          # tail only
    (5000, 64, 7, 64, 3, 45),        # k = 64
    (6000, 32, 6, 100, 3, 40),       # reference bench K = 100: 4 list slots per lane
    (3000, 16, 4, 255, 4, 30),       # k + 1 = 256: the widest list the tile path takes
    (300, 16, 3, 10, 3, 300),        # many queries, tiny lists (< 64 rows per wave)
])
@pytest.mark.parametrize("layout", ["ivf", "row"])
def test_tile_path_matches_oracle(pqv, oracle, monkeypatch, n, dim, kc, k, nprobe, nq, layout):
    rng = np.random.default_rng(n + 7 * dim + nq)
    data, oidx = _random_index(oracle, rng, n, dim, kc)
    queries = rng.random((nq, dim), dtype=np.float32)
    corpus = pqv.Corpus.upload(data)
    index = pqv.Index.from_bytes(oidx.to_bytes())
    flags = pqv.PQV_LAYOUT_ROW_ORDER if layout == "row" else pqv.PQV_LAYOUT_IVF_ORDERED
    orows, odist, onf, onc = oidx.topk_batch(data, queries, k, nprobe)
    results = {}
    for mode in ("tile", "screen", "stream"):
        monkeypatch.setenv("PQV_RERANK_MODE", "tile" if mode == "screen" else mode)
        monkeypatch.setenv("PQV_TILE_FILTER", "2" if mode == "screen" else "0")   # 2 = force the MFMA screen
        s = pqv.Searcher(index, corpus, flags)
        rows, dist, nf, nc = s.topk(queries, k, nprobe)
        assert (nc == onc).all()
        _assert_topk_equal((rows, dist, nf), (orows, odist, onf), k)
        results[mode] = (rows, dist)
    for m in ("tile", "screen"):
        assert (results[m][0] == results["stream"][0]).all()
        assert (_bits(results[m][1]) == _bits(results["stream"][1])).all()
    # the cap applies to the tile path too (screened form)
    monkeypatch.setenv("PQV_RERANK_MODE", "tile")
    monkeypatch.setenv("PQV_TILE_FILTER", "2")
    s = pqv.Searcher(index, corpus, flags)
    rows, d2, nf, nc = s.topk(queries[:5], k, nprobe, max_candidates=97, sqrt_out=False)
    for q in range(5):
        cand = oidx.candidate_rows(queries[q], nprobe)[:97]
        d = np.array([oracle.l2_ref4(queries[q], data[r]) for r in cand], np.float32)
        order = np.lexsort((np.arange(len(cand)), d.view(np.uint32)))[:k]
        assert (rows[q, :len(order)] == cand[order]).all() and nf[q] == len(order)


# ---------------------------------------------------------------------------------------
# reference API on real Parquet files
# ---------------------------------------------------------------------------------------
def test_parquet_build_inplace_and_topk(pqv, oracle, tmp_path):
    """README quick-start flow of the reference: IndexBuilder::new(path, col).build_inplace(),
    then TopkBuilder::new(path, &query).k(..)?.nprobe(..)?.search() -- checked against the
    oracle run on the same column values (f64 column: narrowed like parquet.rs:246-256)."""
    import os
    import pyarrow as pa
    import pyarrow.parquet as pq
    rng = np.random.default_rng(21)
    n, dim = 3000, 48
    vecs64 = rng.random((n, dim))
    t = pa.table({"id": pa.array(np.arange(n, dtype=np.int32)),
                  "embedding": pa.array(vecs64.tolist(), type=pa.list_(pa.field("item", pa.float64())))})
    path = str(tmp_path / "data.parquet")
    pq.write_table(t, path, row_group_size=1024)
    size0 = os.path.getsize(path)
    index = pqv.IndexBuilder(path, "embedding").n_clusters(12).workers(4).build_inplace()
    assert os.path.getsize(path) > size0 and pqv.has_pq_vector_index(path)
    data = vecs64.astype(np.float32)
    oidx = oracle.build_index(data, n_clusters=12, workers=4)
    assert index.to_bytes() == oidx.to_bytes()
    stored, col = pqv.read_index_from_parquet(path)
    assert col == "embedding" and stored.to_bytes() == oidx.to_bytes()
    assert pq.read_table(path).num_rows == n                       # still a valid Parquet file

    q = rng.random(dim, dtype=np.float32)
    hits = pqv.TopkBuilder(path, q).k(10).nprobe(4).search()
    orows, odist, _ = oidx.topk(data, q, 10, 4)
    assert [h.row_idx for h in hits] == orows.tolist()
    assert [np.float32(h.distance) for h in hits] == odist.tolist()
    # second query reuses the resident searcher
    hits2 = pqv.TopkBuilder(path, data[17]).k(3).nprobe(12).search()
    assert hits2[0].row_idx == 17 and hits2[0].distance == 0.0


@pytest.mark.parametrize("dim,kc,images_only", [(64, 8, True), (48, 8, False), (64, 64, False)])
def test_path_searcher_keeps_one_f32_copy_of_the_column_resident(pqv, oracle, tmp_path, dim, kc, images_only):
    """searcher_for_parquet (TopkBuilder(path)): the images-only IVF layout keeps reading the column as loaded; where the searcher
    falls back to its own list-ordered f32 copy (dim % 64 != 0, lists under 192 rows) the loaded row-order rows are released
    (PQV_RELEASE_IF_COPIED) -- one f32 copy of the column in HBM either way, and the answers are the oracle's."""
    import pyarrow as pa
    import pyarrow.parquet as pq
    from pq_vector_amd import api as _api
    rng = np.random.default_rng(31)
    n = 4000
    data = rng.random((n, dim), dtype=np.float32)
    col = pa.ListArray.from_arrays(pa.array(np.arange(0, (n + 1) * dim, dim, dtype=np.int32)), pa.array(data.reshape(-1)))
    path = str(tmp_path / "c.parquet")
    pq.write_table(pa.table({"embedding": col}), path, row_group_size=1500)
    pqv.IndexBuilder(path, "embedding").n_clusters(kc).workers(4).build_inplace()
    oidx = oracle.build_index(data, n_clusters=kc, workers=4)
    _api._PATH_SEARCHERS.clear()
    try:
        s = pqv.searcher_for_parquet(path)
        fp = s.footprint()
        column_bytes = n * dim * 4
        if images_only:
            assert fp["row_order_bytes"] == column_bytes and fp["ivf_rows_bytes"] == 0
        else:
            assert fp["row_order_bytes"] == 0 and fp["ivf_rows_bytes"] >= column_bytes
        for q in (data[5], rng.random(dim, dtype=np.float32)):
            hits = pqv.TopkBuilder(path, q).k(7).nprobe(3).search()
            orows, odist, _ = oidx.topk(data, q, 7, 3)
            assert [h.row_idx for h in hits] == orows.tolist()
            assert [np.float32(h.distance) for h in hits] == odist.tolist()
    finally:
        _api._PATH_SEARCHERS.clear()


@pytest.mark.parametrize("value_type", ["f32", "f64"])
def test_page_runs_from_the_mapped_file_and_what_they_refuse(pqv, tmp_path, value_type):
    """The page-level loader on an uncompressed PLAIN file with hundreds of small pages: runs of pages go through
    pqv_corpus_write_plain_pages (levels checked, values uploaded natively) and the resident matrix is the column; one null
    value, one null row or one ragged list deep inside the file makes its page's check fail, the loader falls back to the Arrow
    reader and that raises the reference's message (parquet.rs:231-280)."""
    import pyarrow as pa
    import pyarrow.parquet as pq
    from pq_vector_amd import parquet_io
    rng = np.random.default_rng(31)
    n, dim = 60000, 40
    dt = np.float64 if value_type == "f64" else np.float32
    vecs = rng.standard_normal((n, dim)).astype(dt)

    def write(path, rows):
        pq.write_table(pa.table({"emb": pa.array(rows, type=pa.list_(pa.float64() if value_type == "f64" else pa.float32()))}),
                       path, compression="NONE", use_dictionary=False, data_page_size=32 * 1024, row_group_size=25000)

    good = str(tmp_path / "good.parquet")
    col = pa.ListArray.from_arrays(pa.array(np.arange(0, (n + 1) * dim, dim, dtype=np.int32)), pa.array(vecs.reshape(-1)))
    pq.write_table(pa.table({"emb": col}), good, compression="NONE", use_dictionary=False, data_page_size=32 * 1024, row_group_size=25000)
    stats = {}
    corpus = parquet_io.load_embedding_column(good, "emb", stats=stats)
    assert stats["path"].startswith("data pages") and stats["pages"] > 3 * parquet_io._PAGE_RUN
    full = corpus.fetch_rows(np.arange(n, dtype=np.uint32))
    assert np.array_equal(full.view(np.uint32), vecs.astype(np.float32).view(np.uint32))
    corpus.close()
    small = vecs[:9000].tolist()
    for name, row, text in (("null_value", small[5000][:-1] + [None], "null"), ("null_row", None, "null"), ("ragged", small[5000][:-1], "inconsistent")):
        rows = list(small)
        rows[5000] = row
        bad = str(tmp_path / f"{name}.parquet")
        write(bad, rows)
        with pytest.raises(pqv.PqvError, match=text):
            parquet_io.load_embedding_column(bad, "emb")


def test_loader_reports_the_first_error_in_file_order(pqv, tmp_path):
    """The reference walks the batches in file order and returns the FIRST error in that order (parquet.rs:231-280).  The threaded
    Arrow fallback (one reader per row group) must too: a null row in row group 1 and ragged lists in row groups 3 and 5 -- whichever
    thread fails first in TIME, the message is row group 1's, every time; with the defects swapped, the other one.  A shard that
    starts behind the first defect (row_groups=(2, 6)) reports ITS first one."""
    import pyarrow as pa
    import pyarrow.parquet as pq
    from pq_vector_amd import parquet_io
    rng = np.random.default_rng(9)
    dim, per = 24, 3000
    base = rng.standard_normal((6 * per, dim)).astype(np.float32).tolist()

    def write(name, defects):
        rows = list(base)
        for rg, kind in defects.items():
            rows[rg * per + 17] = None if kind == "null" else rows[rg * per + 17][:-1]
        path = str(tmp_path / name)
        pq.write_table(pa.table({"emb": pa.array(rows, type=pa.list_(pa.float32()))}), path, compression="NONE", use_dictionary=False,
                       row_group_size=per)
        return path
    a = write("a.parquet", {1: "null", 3: "ragged", 5: "ragged"})
    b = write("b.parquet", {1: "ragged", 3: "null", 5: "null"})
    for _ in range(4):
        with pytest.raises(pqv.PqvError, match="contains null rows"):
            parquet_io.load_embedding_column(a, "emb", readers=6)
        with pytest.raises(pqv.PqvError, match="inconsistent dimensions"):
            parquet_io.load_embedding_column(b, "emb", readers=6)
    with pytest.raises(pqv.PqvError, match="inconsistent dimensions"):
        parquet_io.load_embedding_column(a, "emb", readers=4, row_groups=(2, 6))
    ok = parquet_io.load_embedding_column(a, "emb", row_groups=(2, 3))          # a clean row group of the same file loads
    assert ok.rows == per and np.array_equal(ok.fetch_rows(np.array([0, per - 1], np.uint32)), np.array([base[2 * per], base[3 * per - 1]], np.float32))
    ok.close()


@pytest.mark.parametrize("value_type,list_kind", [("f32", "list"), ("f64", "list"), ("f32", "fixed")])
def test_streamed_parquet_loader_places_every_row_group(pqv, tmp_path, value_type, list_kind):
    """N1 (src/ivf/parquet.rs:216-305): row groups of unequal sizes decoded by several reader threads, every batch uploaded
    through the pinned staging buffers at its row offset (pqv_corpus_write_rows, any order) -- the resident matrix must be the
    column, row for row, and a float64 column narrowed `as f32` (:246-256).  Batches larger than a staging buffer are split."""
    import pyarrow as pa
    import pyarrow.parquet as pq
    from pq_vector_amd import parquet_io
    rng = np.random.default_rng(3)
    dim = 96
    sizes = [70000, 1, 33333, 5000, 90000, 12]
    n = sum(sizes)
    vecs = rng.standard_normal((n, dim)).astype(np.float64 if value_type == "f64" else np.float32)
    vt = pa.float64() if value_type == "f64" else pa.float32()
    typ = pa.list_(pa.field("item", vt), dim) if list_kind == "fixed" else pa.list_(pa.field("item", vt))
    path = str(tmp_path / "col.parquet")
    writer = None
    at = 0
    for m in sizes:
        flat = pa.array(vecs[at:at + m].reshape(-1), type=vt)
        if list_kind == "fixed":
            col = pa.FixedSizeListArray.from_arrays(flat, dim)
        else:
            col = pa.ListArray.from_arrays(pa.array(np.arange(0, (m + 1) * dim, dim, dtype=np.int32)), flat)
        t = pa.table({"id": pa.array(np.arange(at, at + m, dtype=np.int32)), "emb": col.cast(typ)})
        if writer is None:
            writer = pq.ParquetWriter(path, t.schema)
        writer.write_table(t, row_group_size=m)
        at += m
    writer.close()
    assert pq.ParquetFile(path).metadata.num_row_groups == len(sizes)
    stats = {}
    corpus = parquet_io.load_embedding_column(path, "emb", readers=3, stats=stats)
    assert corpus.rows == n and corpus.dim == dim and stats["rows"] == n and stats["reader_threads"] == 3
    want = vecs.astype(np.float32)
    sel = np.concatenate([np.arange(0, n, 997), np.cumsum(sizes)[:-1], np.cumsum(sizes)[:-1] - 1, [0, n - 1]]).astype(np.uint32)
    got = corpus.fetch_rows(sel)
    assert np.array_equal(got.view(np.uint32), want[sel].view(np.uint32))
    # the whole matrix, through the device
    import torch
    full = corpus.fetch_rows(np.arange(n, dtype=np.uint32))
    assert np.array_equal(full.view(np.uint32), want.view(np.uint32))


def test_parquet_build_new_reference_fixture(pqv, tmp_path):
    """src/df_vector/tests.rs:16-104 data through build_new: 6 rows x 2-D, default clusters."""
    import pyarrow as pa
    import pyarrow.parquet as pq
    vecs = [(0, 0), (1, 0), (0, 2), (5, 5), (2, 2), (0.1, 0.1)]
    t = pa.table({"id": pa.array(range(6), type=pa.int32()),
                  "vec": pa.array([list(v) for v in vecs], type=pa.list_(pa.field("item", pa.float32())))})
    src, out = str(tmp_path / "source.parquet"), str(tmp_path / "indexed.parquet")
    pq.write_table(t, src)
    pqv.IndexBuilder(src, "vec").build_new(out)
    assert pqv.has_pq_vector_index(out) and not pqv.has_pq_vector_index(src)
    index, col = pqv.read_index_from_parquet(out)
    assert col == "vec" and index.dim == 2 and index.n_clusters == 3
    hits = pqv.TopkBuilder(out, [0.0, 0.0]).k(2).nprobe(64).search()
    assert [h.row_idx for h in hits] == [0, 5]
    with pytest.raises(pqv.PqvError, match="Embedding column name cannot be empty"):
        pqv.IndexBuilder(src, " ").build_inplace()
    with pytest.raises(pqv.PqvError, match="Column 'nope' not found"):
        pqv.IndexBuilder(src, "nope").build_inplace()


# ---------------------------------------------------------------------------------------
# BASELINE.json configs[1] at FULL size (1 M x 128, n_clusters 100, nprobe 8, k 10):
# size-independent properties + an oracle spot check
# ---------------------------------------------------------------------------------------
def test_full_size_c2_properties(pqv, oracle, monkeypatch):
    n, dim, kc, k, nprobe, nq = 1_000_000, 128, 100, 10, 8, 512
    rng = np.random.default_rng(1234)
    data = (rng.integers(0, 1 << 24, size=(n, dim), dtype=np.int32).astype(np.float32)
            * np.float32(1.0 / (1 << 24)))                       # the bench recipe: 24-bit uniform [0,1)
    corpus = pqv.Corpus.upload(data)
    index = pqv.IndexBuilder(corpus).n_clusters(kc).workers(8).build()
    # index invariants: every row listed exactly once, lists ascending
    off, rows = index.list_offsets, index.list_rows
    assert int(off[-1]) == n and np.array_equal(np.sort(rows), np.arange(n, dtype=np.uint32))
    for c in range(kc):
        l = rows[int(off[c]):int(off[c + 1])].astype(np.int64)
        assert (np.diff(l) > 0).all()
    # determinism / idempotence: a second build gives the identical blob
    blob = index.to_bytes()
    assert pqv.IndexBuilder(corpus).n_clusters(kc).workers(8).build().to_bytes() == blob
    assert pqv.Index.from_bytes(blob).to_bytes() == blob

    queries = np.ascontiguousarray(data[rng.choice(n, nq, replace=False)])   # self-queries
    queries[nq // 2:] = rng.random((nq - nq // 2, dim), dtype=np.float32)
    out = {}
    for mode in ("tile", "screen", "stream"):
        monkeypatch.setenv("PQV_RERANK_MODE", "tile" if mode == "screen" else mode)
        monkeypatch.setenv("PQV_TILE_FILTER", "2" if mode == "screen" else "0")
        s = pqv.Searcher(index, corpus)
        out[mode] = s.topk(queries, k, nprobe)
    # three independent paths (lane-per-row SGPR tiles, the same behind the MFMA screen, and the
    # coalesced stream + LDS transpose) must agree bit for bit
    for m in ("tile", "screen"):
        for a, b in zip(out[m], out["stream"]):
            assert np.array_equal(a.view(np.uint32) if a.dtype == np.float32 else a,
                                  b.view(np.uint32) if b.dtype == np.float32 else b)
    rows_t, dist_t, nf, nc = out["tile"]
    assert (nf == k).all()
    assert (np.diff(dist_t.astype(np.float64), axis=1) >= 0).all()          # sorted ascending
    # a row queried with itself comes back first at distance 0 (its own cluster is probed first)
    self_q = queries[:nq // 2]
    assert (dist_t[:nq // 2, 0] == 0).all()
    assert (np.abs(data[rows_t[:nq // 2, 0]] - self_q).max(axis=1) == 0).all()
    # distances are what they claim: recompute in f64 for every returned row
    for q in range(0, nq, 37):
        d = np.sqrt(((data[rows_t[q]].astype(np.float64) - queries[q]) ** 2).sum(axis=1))
        assert np.allclose(d, dist_t[q], rtol=1e-5, atol=1e-6)
    # oracle spot check on the same index (bit-exact)
    oidx = oracle.index_from_bytes(blob)
    sel = np.arange(0, nq, 16)
    orows, odist, onf, onc = oidx.topk_batch(data, queries[sel], k, nprobe)
    assert (rows_t[sel] == orows).all() and (_bits(dist_t[sel]) == _bits(odist)).all()
    assert (nc[sel] == onc).all()


# ---------------------------------------------------------------------------------------
# BASELINE config 5 (extension): batched brute-force cosine / L2 on the matrix cores vs an
# f64 brute-force oracle.  Tolerance: distances within 1e-4 relative (north star), ids equal
# except where the oracle's own neighbouring distances are closer than that tolerance.
# ---------------------------------------------------------------------------------------
def _f64_brute(data, queries, k, metric):
    d64, q64 = data.astype(np.float64), queries.astype(np.float64)
    s = q64 @ d64.T
    if metric == "cos":
        nq_, nv = np.linalg.norm(q64, axis=1), np.linalg.norm(d64, axis=1)
        with np.errstate(divide="ignore", invalid="ignore"):
            dist = 1.0 - s / (nq_[:, None] * nv[None, :])
        dist[~np.isfinite(dist)] = 1.0
    else:
        dist = (q64 ** 2).sum(1)[:, None] + (d64 ** 2).sum(1)[None, :] - 2 * s
    order = np.argsort(dist, axis=1, kind="stable")[:, :k]
    return order, np.take_along_axis(dist, order, axis=1), dist


@pytest.mark.parametrize("n,dim,nq,k", [
    (20000, 1536, 130, 10),     # C5-shaped rows (ada-002 dim), > 1 query tile, partial tiles
    (9000, 96, 33, 10),
    (70000, 64, 260, 25),       # several progressive ranges (8k, 64k)
    (40000, 100, 140, 10),      # screened ranges with rows padded to the image's K granule (100 -> 128 values), one partial 256-query tile
    (500, 50, 5, 10),           # dim % 16 != 0, dim % 4 != 0, fewer rows than a tile
    (300, 7, 3, 400),           # k > n
])
@pytest.mark.parametrize("metric", ["cos", "l2"])
@pytest.mark.parametrize("op", ["i8", "f16"])
def test_brute_mfma_matches_f64_oracle(pqv, monkeypatch, n, dim, nq, k, metric, op):
    monkeypatch.setenv("PQV_BRUTE_OP", op)          # operand form of the screen beyond the first row range (read per call)
    rng = np.random.default_rng(n + dim)
    data = rng.standard_normal((n, dim)).astype(np.float32)
    queries = rng.standard_normal((nq, dim)).astype(np.float32)
    if metric == "cos" and n > 400:
        data[123] = 0                      # a zero-norm row: distance 1 by definition
    corpus = pqv.Corpus.upload(data)
    m = pqv.PQV_COSINE if metric == "cos" else pqv.PQV_L2SQ_MFMA
    rows, dist, nf = corpus.brute_topk(queries, k, m)
    order, odist, full = _f64_brute(data, queries, k, metric)
    kk = min(k, n)
    assert (nf == kk).all()
    scale = np.maximum(np.abs(odist[:, :kk]), 1e-3)
    assert (np.abs(dist[:, :kk] - odist[:, :kk]) <= 1e-4 * scale + 1e-6).all(), "distance beyond 1e-4 relative"
    for q in range(nq):
        got = rows[q, :kk]
        assert len(set(got.tolist())) == kk and (got < n).all()
        # every returned row must truly belong: its exact distance is within tolerance of the
        # oracle's k-th distance, and every oracle row clearly inside the k-th must be present
        kth = odist[q, kk - 1]
        tol = 1e-4 * max(abs(kth), 1e-3) + 1e-6
        assert (full[q, got] <= kth + tol).all()
        sure = order[q, :kk][odist[q, :kk] < kth - tol]
        assert set(sure.tolist()) <= set(got.tolist())
        assert (np.diff(dist[q, :kk]) >= 0).all()
    assert (rows[:, kk:] == 0xFFFFFFFF).all()


@pytest.mark.parametrize("metric", ["cos", "l2"])
@pytest.mark.parametrize("op", ["i8", "f16", "i8-ring", "f16-ring"])
def test_brute_screen_on_hostile_data(pqv, monkeypatch, metric, op):
    """The int8 / f16 screens of pqv_brute_topk are bounds, not approximations: rows with one dominant component (a coarse
    int8 grid for everything else), near-duplicates of the queries (results decided inside the image's slack), tiny and
    huge norms, a zero row -- the result must still be the exact f32 one."""
    monkeypatch.setenv("PQV_BRUTE_OP", op.split("-")[0])
    monkeypatch.setenv("PQV_BRUTE_RING", "1" if op.endswith("ring") else "0")      # direct-to-LDS form of the 256 x 256 tile
    n, dim, nq, k = 120_000, 192, 170, 10                # > 128 queries: the 256 x 256 tile
    rng = np.random.default_rng(99)
    data = rng.standard_normal((n, dim)).astype(np.float32)
    queries = rng.standard_normal((nq, dim)).astype(np.float32)
    data[40_000:40_500, 7] += 60.0                       # one component carries the norm: S = 127 / ~1, the rest falls to 0 / +-1
    data[50_000:50_200] *= np.float32(1e-12)             # tiny norms
    data[60_000:60_200] *= np.float32(1e12)              # huge norms
    data[70_000] = 0
    for q in range(nq):                                  # 30 near-duplicates of every query, far beyond the first exact range
        data[80_000 + 30 * q:80_000 + 30 * q + 30] = queries[q] + 1e-3 * rng.standard_normal((30, dim)).astype(np.float32)
    queries[3, 11] += 80.0                               # a query with a dominant component
    corpus = pqv.Corpus.upload(data)
    m = pqv.PQV_COSINE if metric == "cos" else pqv.PQV_L2SQ_MFMA
    rows, dist, nf = corpus.brute_topk(queries, k, m)
    order, odist, full = _f64_brute(data, queries, k, metric)
    assert (nf == k).all()
    scale = np.maximum(np.abs(odist), 1e-3)
    # (the norm-expansion form |q|^2 + |v|^2 - 2 q.v cancels for near-duplicates: its f32 error scales with the norms)
    q2 = (queries.astype(np.float64) ** 2).sum(1)
    v2 = (data.astype(np.float64) ** 2).sum(1)
    canc = 4e-6 * (q2[:, None] + v2[order]) if metric == "l2" else 0.0
    assert (np.abs(dist - odist) <= 1e-4 * scale + 1e-6 + canc).all()
    for q in range(nq):
        kth = odist[q, -1]
        tol = 1e-4 * max(abs(kth), 1e-3) + 1e-6 + (4e-6 * (q2[q] + v2[order[q]].max()) if metric == "l2" else 0.0)
        assert (full[q, rows[q]] <= kth + tol).all(), q
        sure = order[q][odist[q] < kth - tol]
        assert set(sure.tolist()) <= set(rows[q].tolist()), q


def test_brute_rejects_bad_arguments(pqv):
    corpus = pqv.Corpus.upload(np.ones((10, 4), np.float32))
    with pytest.raises(pqv.PqvError, match="Query dimension mismatch: expected 4, got 3"):
        corpus.brute_topk(np.ones((1, 3), np.float32), 1)
    with pytest.raises(pqv.PqvError, match="k must be > 0"):
        corpus.brute_topk(np.ones((1, 4), np.float32), 0)
    with pytest.raises(pqv.PqvError, match="metric must be"):
        corpus.brute_topk(np.ones((1, 4), np.float32), 1, metric=pqv.PQV_L2SQ_REF4)


@pytest.mark.parametrize("n,dim,kc,k,nprobe,nq", [
    (20000, 128, 8, 10, 4, 200),     # wide kernel, 4 query groups per block (64 queries in LDS)
    (12000, 256, 5, 10, 3, 150),     # wide kernel, 2 query groups per block
    (16000, 64, 6, 32, 3, 130),      # k = 32: the largest k the screen takes by default
    (9000, 192, 4, 5, 2, 77),        # dim % 64 == 0 but not a power of two; partial quads
    (12000, 96, 5, 10, 3, 100),      # dim % 64 != 0: one 16-query group per block (tile_filter_kernel)
    (9000, 768, 4, 10, 3, 90),       # long rows: wide kernel with the queries in a blocked global copy
    (8000, 320, 4, 7, 2, 70),        # same, dim / 64 odd
    (30000, 256, 6, 100, 3, 140),    # K = 100 (the reference's bench): 4 list slots per lane, deferred evaluation by default
    (24000, 1024, 5, 100, 2, 70),    # ... on 1024-dim rows, f16 images, one 8-wave block per CU
])
@pytest.mark.parametrize("variant", ["default", "tiny_buffer", "narrow", "deferred", "deferred_tiny_buffer"])
def test_screened_paths_match_oracle(pqv, oracle, monkeypatch, n, dim, kc, k, nprobe, nq, variant):
    """Long lists so the MFMA screen really runs: the wide kernel (queries staged in LDS, rows from the
    blocked copy, survivors appended to per-query buffers), the same with a 16-entry buffer (every
    query overflows into the per-wave sorted lists) and the one-group-per-block kernel must all
    reproduce the oracle bit for bit.  "deferred": survivors are appended with the bounds their screen score
    gives and the exact evaluations happen after the filter, for the entries the k-th smallest upper bound
    leaves (what the library does for k > 64: these variants ask for 70 neighbours where the case has fewer);
    with a 16-entry buffer (raised to k by the library) most of them overflow and are evaluated by the
    streaming wave after all."""
    rng = np.random.default_rng(3 * n + dim + nq)
    data, oidx = _random_index(oracle, rng, n, dim, kc)
    queries = rng.random((nq, dim), dtype=np.float32)
    queries[::7] = data[rng.integers(0, n, size=len(queries[::7]))]      # exact hits: distance 0
    corpus = pqv.Corpus.upload(data)
    index = pqv.Index.from_bytes(oidx.to_bytes())
    monkeypatch.setenv("PQV_RERANK_MODE", "tile")
    monkeypatch.setenv("PQV_TILE_FILTER", "2")
    if variant.endswith("tiny_buffer"):
        monkeypatch.setenv("PQV_CAND_CAP", "16")
    if variant == "narrow":
        monkeypatch.setenv("PQV_FILTER_VARIANT", "1")
    if variant.startswith("deferred"):
        k = max(k, 70)
    elif variant != "default":               # ("default": the library's own rule -- deferred for k > 64)
        monkeypatch.setenv("PQV_DEFER", "0")
    s = pqv.Searcher(index, corpus)
    orows, odist, onf, onc = oidx.topk_batch(data, queries, k, nprobe)
    rows, dist, nf, nc = s.topk(queries, k, nprobe)
    assert (nc == onc).all()
    _assert_topk_equal((rows, dist, nf), (orows, odist, onf), k)
    if variant.startswith("deferred"):       # a handful of queries per call (once resolved by the merge's own block)
        for q0, m in ((0, 3), (5, 3), (2, 1), (7, 1)):         # (one query: select + thresholds in the seed kernel's tail)
            r1, d1, n1, _ = s.topk(queries[q0:q0 + m], k, nprobe)
            _assert_topk_equal((r1, d1, n1), (orows[q0:q0 + m], odist[q0:q0 + m], onf[q0:q0 + m]), k)
    c = s.counters()
    # (K = 100 of the ~10 k candidates a query has here is 1 % of them before any margin: the screen cannot drop as much)
    assert c["screened_pairs"] > 0 and 0 < c["screen_survivors"] < (0.2 if k <= 32 else 0.6) * c["screened_pairs"]
    # max_candidates cuts inside the screened window
    cap = 2 * (n // kc) // 3 + 300
    rows, d2, nf, nc = s.topk(queries[:6], k, nprobe, max_candidates=cap, sqrt_out=False)
    for q in range(6):
        cand = oidx.candidate_rows(queries[q], nprobe)[:cap]
        d = np.array([oracle.l2_ref4(queries[q], data[r]) for r in cand], np.float32)
        order = np.lexsort((np.arange(len(cand)), d.view(np.uint32)))[:k]
        assert nf[q] == len(order)
        assert (_bits(d2[q, :len(order)]) == _bits(d[order])).all()


def test_deferred_evaluation_rule_for_short_lists_of_long_rows(pqv, oracle, monkeypatch):
    """k <= 64: batches on short lists of long rows (int8 images, >= 512 dims, mean list <= 3072 rows) defer their exact
    evaluations behind the screen; one-query calls and PQV_DEFER=0 do not.  All three give the oracle's answer bit for bit."""
    rng = np.random.default_rng(99)
    n, dim, kc, k, nprobe, nq = 30000, 768, 24, 10, 6, 200
    data, oidx = _random_index(oracle, rng, n, dim, kc)
    queries = rng.random((nq, dim), dtype=np.float32)
    queries[::9] = data[rng.integers(0, n, size=len(queries[::9]))]
    corpus = pqv.Corpus.upload(data)
    index = pqv.Index.from_bytes(oidx.to_bytes())
    orows, odist, onf, onc = oidx.topk_batch(data, queries, k, nprobe)
    for env in ("1", "0"):
        monkeypatch.setenv("PQV_DEFER", env)
        s = pqv.Searcher(index, corpus)
        text = s.describe(nq, k + 1, nprobe)
        assert ("exact evaluations deferred" in text) == (env == "1"), text
        assert "deferred" not in s.describe(1, k + 1, nprobe)
        if env == "1":
            assert "wide_filter_kernel<6, 4, 1, true, 2, false, true, 4, true>" in text
        rows, dist, nf, nc = s.topk(queries, k, nprobe)
        assert (nc == onc).all()
        _assert_topk_equal((rows, dist, nf), (orows, odist, onf), k)
        r1, d1, n1, _ = s.topk(queries[3:4], k, nprobe)
        _assert_topk_equal((r1, d1, n1), (orows[3:4], odist[3:4], onf[3:4]), k)
        c = s.counters()
        assert c["screen_survivors"] > 0
        s.close() if hasattr(s, "close") else None


def test_mfma_screen_under_cancellation(pqv, oracle, monkeypatch):
    """Rows = large common offset + tiny noise: |q|^2 + |x|^2 - 2 q.x cancels catastrophically,
    the screen's margin dwarfs every distance, so (nearly) all pairs must survive it and be
    evaluated exactly -- results stay bit-identical to the oracle; also exercises the per-wave
    pending queue at its worst-case fill.  A second corpus mixes magnitudes over 6 decades."""
    rng = np.random.default_rng(77)
    n, dim, kc, k, nprobe, nq = 12000, 64, 6, 10, 3, 96
    for variant in ("offset", "decades"):
        if variant == "offset":
            data = (100.0 + 0.01 * rng.random((n, dim))).astype(np.float32)
            queries = (100.0 + 0.01 * rng.random((nq, dim))).astype(np.float32)
        else:
            scale = (10.0 ** rng.integers(-3, 4, size=(n, 1))).astype(np.float32)
            data = (rng.standard_normal((n, dim)).astype(np.float32) * scale)
            queries = rng.standard_normal((nq, dim)).astype(np.float32) * np.float32(10.0)
        oidx = oracle.build_index(data, n_clusters=kc, workers=1, max_iters=5)
        corpus = pqv.Corpus.upload(data)
        monkeypatch.setenv("PQV_RERANK_MODE", "tile")
        monkeypatch.setenv("PQV_TILE_FILTER", "2")      # force the screen regardless of the dispatch rule
        s = pqv.Searcher(pqv.Index.from_bytes(oidx.to_bytes()), corpus)
        rows, dist, nf, nc = s.topk(queries, k, nprobe)
        orows, odist, onf, onc = oidx.topk_batch(data, queries, k, nprobe)
        assert (nc == onc).all() and (nf == onf).all()
        assert (rows == orows).all(), variant
        assert (_bits(dist) == _bits(odist)).all(), variant
        c = s.counters()
        assert c["screened_pairs"] > 0
        if variant == "offset":
            assert c["screen_survivors"] > 0.9 * c["screened_pairs"]   # the bound cannot prune here


def test_merge_topk_device_matches_stable_sort(pqv):
    """pqv_merge_topk_device (the multi-GPU exchange merge) == shard-major stable sort, incl. ties,
    short lists and row bases beyond 2^32."""
    import torch
    from pq_vector_amd import _ffi
    from pq_vector_amd.sharding import merge_gathered
    lib = _ffi.lib()
    torch.manual_seed(5)
    for (w, nq, k) in [(1, 3, 10), (2, 17, 10), (8, 64, 100), (4, 5, 300), (3, 2, 1)]:
        d = torch.randint(0, 50, (w, nq, k), device="cuda").float().sort(dim=2).values  # many ties
        r = torch.randint(0, 1 << 31, (w, nq, k), device="cuda", dtype=torch.int64)
        short = torch.rand((w, nq, 1), device="cuda") < 0.3          # some lists are short
        tail = torch.arange(k, device="cuda").view(1, 1, k) >= max(1, k // 2)
        empty = short & tail
        d = torch.where(empty, torch.full_like(d, float("inf")), d).contiguous()
        bases = torch.tensor([i * ((1 << 32) + 12345) for i in range(w)], dtype=torch.int64, device="cuda")
        grow = torch.where(empty, torch.full_like(r, -1), r + bases.view(w, 1, 1))
        want_d, want_r = merge_gathered(d, grow, k)
        r_u32 = torch.where(empty, torch.full_like(r, -1), r).to(torch.int32).contiguous()  # -1 == 0xFFFFFFFF
        out_d = torch.empty((nq, k), device="cuda")
        out_r = torch.empty((nq, k), dtype=torch.int64, device="cuda")
        rc = lib.pqv_merge_topk_device(0, _ffi.vp(d.data_ptr()), _ffi.vp(r_u32.data_ptr()),
                                       _ffi.vp(bases.data_ptr()), w, nq, k, _ffi.vp(out_d.data_ptr()),
                                       _ffi.vp(out_r.data_ptr()), _ffi.vp(torch.cuda.current_stream().cuda_stream))
        assert rc == 0, lib.pqv_last_error()
        torch.cuda.synchronize()
        assert torch.equal(out_d, want_d), (w, nq, k)
        assert torch.equal(out_r, want_r), (w, nq, k)


def test_calls_on_two_streams_use_independent_scratch(pqv, oracle):
    """pqv_topk_device calls enqueued on two streams overlap on the GPU (one scratch lane per stream);
    interleaved calls with DIFFERENT query batches must each return their own exact result."""
    import torch
    rng = np.random.default_rng(99)
    n, dim, kc, k, nprobe, nq = 40000, 128, 16, 10, 4, 256
    data, oidx = _random_index(oracle, rng, n, dim, kc)
    corpus = pqv.Corpus.upload(data)
    s = pqv.Searcher(pqv.Index.from_bytes(oidx.to_bytes()), corpus)
    batches = [rng.random((nq, dim), dtype=np.float32) for _ in range(4)]
    want = [oidx.topk_batch(data, q, k, nprobe) for q in batches]
    streams = [torch.cuda.Stream(), torch.cuda.Stream()]
    q_t = [torch.from_numpy(q).cuda() for q in batches]
    outs = []
    torch.cuda.synchronize()
    for rep in range(3):                       # 12 calls in flight, alternating streams
        for i, q in enumerate(q_t):
            st = streams[i % 2]
            rows = torch.empty((nq, k), dtype=torch.int32, device="cuda")
            dist = torch.empty((nq, k), dtype=torch.float32, device="cuda")
            nf = torch.empty((nq,), dtype=torch.int32, device="cuda")
            with torch.cuda.stream(st):
                s.topk_device(q.data_ptr(), nq, k, nprobe, rows.data_ptr(), dist.data_ptr(), nf.data_ptr(),
                              stream=st.cuda_stream)
            outs.append((i, rows, dist, nf))
    torch.cuda.synchronize()
    for i, rows, dist, nf in outs:
        orows, odist, onf, _ = want[i]
        got = (rows.cpu().numpy().view(np.uint32), dist.cpu().numpy(), nf.cpu().numpy().view(np.uint32))
        _assert_topk_equal(got, (orows, odist, onf), k)


@pytest.mark.parametrize("dim", [64, 128, 320])
def test_screened_path_unbalanced_lists(pqv, oracle, monkeypatch, dim):
    """One huge dense cluster plus tiny and (possibly) empty ones, every list probed: the wide screened
    kernels must cope with lists shorter than a tile / than the seed window next to 20 k-row lists."""
    rng = np.random.default_rng(1234 + dim)
    n_dense, n_sparse, kc, k, nq = 24000, 400, 12, 10, 96
    dense = (0.5 + 0.02 * rng.standard_normal((n_dense, dim))).astype(np.float32)
    sparse = (rng.random((n_sparse, dim)) * 4.0 - 1.5).astype(np.float32)
    data = np.concatenate([dense, sparse])[rng.permutation(n_dense + n_sparse)]
    oidx = oracle.build_index(data, n_clusters=kc, workers=1, max_iters=4)
    lens = np.diff(oidx.list_off.astype(np.int64))
    assert lens.max() > 5000 and lens.min() < 200, lens
    queries = np.concatenate([(0.5 + 0.02 * rng.standard_normal((nq // 2, dim))).astype(np.float32),
                              (rng.random((nq // 2, dim)) * 4.0 - 1.5).astype(np.float32)])
    corpus = pqv.Corpus.upload(data)
    monkeypatch.setenv("PQV_RERANK_MODE", "tile")
    monkeypatch.setenv("PQV_TILE_FILTER", "2")
    s = pqv.Searcher(pqv.Index.from_bytes(oidx.to_bytes()), corpus)
    for nprobe in (kc, 3):
        orows, odist, onf, onc = oidx.topk_batch(data, queries, k, nprobe)
        rows, dist, nf, nc = s.topk(queries, k, nprobe)
        assert (nc == onc).all()
        _assert_topk_equal((rows, dist, nf), (orows, odist, onf), k)
    assert s.counters()["screened_pairs"] > 0


@pytest.mark.parametrize("k,copies", [(10, 700), (32, 300), (2, 1500)])
def test_running_thresholds_with_hundreds_of_tied_candidates(pqv, oracle, monkeypatch, k, copies):
    """The running thresholds count appended pairs in 8-bit per-bin counters that are allowed to wrap: here every
    query has `copies` (> 255) candidates at EXACTLY the same small distance (duplicated rows placed late in the
    lists, behind the seed window), so the counters wrap several times while blocks of one query run concurrently.
    Rows, distances and counts must still equal the oracle (whose heap replay decides which tied rows survive)."""
    rng = np.random.default_rng(900 + k)
    dim, kc, nq, nprobe = 128, 4, 48, 4
    n_bg = 20000
    bg = rng.random((n_bg, dim), dtype=np.float32)
    protos = rng.random((nq // 4, dim), dtype=np.float32)                 # 12 prototype rows ...
    dup = np.repeat(protos, copies, axis=0)                               # ... each `copies` times
    data = np.concatenate([bg, dup]).astype(np.float32)                   # duplicates have the highest row ids
    queries = (protos[rng.integers(0, len(protos), nq)] + np.float32(1e-3) * rng.standard_normal((nq, dim))).astype(np.float32)
    oidx = oracle.build_index(data, n_clusters=kc, workers=2, max_iters=3, seed=5)
    monkeypatch.setenv("PQV_RERANK_MODE", "tile")
    monkeypatch.setenv("PQV_TILE_FILTER", "2")
    for rows_env in ("", "256"):                                          # default blocks / many concurrent blocks per list
        if rows_env:
            monkeypatch.setenv("PQV_WIDE_ROWS", rows_env)
        s = pqv.Searcher(pqv.Index.from_bytes(oidx.to_bytes()), pqv.Corpus.upload(data))
        rows, dist, nf, nc = s.topk(queries, k, nprobe)
        orows, odist, onf, onc = oidx.topk_batch(data, queries, k, nprobe)
        assert (nc == onc).all() and (nf == onf).all()
        assert (_bits(dist) == _bits(odist)).all()
        assert (rows == orows).all()
        c = s.counters()
        assert c["screened_pairs"] > 0 and c["screen_survivors"] >= nq * min(copies, 255)


@pytest.mark.parametrize("n,dim,kc,min_k", [
    (24000, 64, 600, None),      # natural dispatch (>= 512 centroids), queries staged in LDS
    (9000, 128, 96, 2),          # forced at a small centroid count: fewer seeds than the seed window
    (6000, 320, 70, 2),          # long rows: blocked global query copy
    (5000, 768, 520, None),      # C3-shaped rows
    (7000, 100, 300, 2),         # dim % 32 != 0: the f16 images are zero-padded (the f32 screen does not apply: exact kernel)
    (40000, 16, 257, 2),         # one centroid past a 256-tile; short rows
])
def test_screened_assignment_builds_identical_index(pqv, oracle, monkeypatch, n, dim, kc, min_k):
    """Index build with the f16 contraction + exact re-scoring (round 3: assign_f16_kernel), with the f32 MFMA-screened
    assignment and with the exact VALU assignment (Lloyd iterations and final assignment alike) must produce the
    oracle's blob byte for byte."""
    rng = np.random.default_rng(n + dim + kc)
    data = rng.random((n, dim), dtype=np.float32)
    data[::5] = np.round(data[::5] * 4) / 4          # coarse values: exact distance ties between centroids occur
    data[7::11] = data[3::11][:len(data[7::11])]      # duplicated rows
    want = oracle.build_index(data, n_clusters=kc, workers=3, max_iters=4, seed=11).to_bytes()
    corpus = pqv.Corpus.upload(data)
    blobs = {}
    for mode in ("gemm", "screen", "exact"):
        monkeypatch.setenv("PQV_ASSIGN_GEMM", (str(min_k) if min_k else "1") if mode == "gemm" else "0")
        monkeypatch.setenv("PQV_ASSIGN_SCREEN", "0" if mode == "exact" else (str(min_k) if min_k else "1"))
        blobs[mode] = pqv.IndexBuilder(corpus).n_clusters(kc).max_iters(4).seed(11).workers(3).build().to_bytes()
    assert blobs["exact"] == want
    assert blobs["screen"] == want
    assert blobs["gemm"] == want


def test_gemm_assignment_with_hostile_geometry(pqv, oracle, monkeypatch):
    """The f16 screen of the assignment images rows and centroids as unit vectors about the centroids' mean.  Data far
    from the origin with a tiny spread (the centring carries it), centroids that coincide (k-means++ on duplicated rows:
    exact ties, lowest index must win), one cluster 10^4 times wider than the others (every candidate list of its rows
    overflows: those rows are compared with every centroid) and rows equal to the mean (zero image): the blob must stay
    the oracle's."""
    rng = np.random.default_rng(404)
    dim, kc = 64, 200
    tight = (rng.random((6000, dim), dtype=np.float32) * np.float32(0.01) + np.float32(500.0)).astype(np.float32)
    dup = np.repeat(tight[:50], 20, axis=0)
    wide = (rng.standard_normal((3000, dim)) * 100.0 + 500.0).astype(np.float32)
    data = np.ascontiguousarray(np.concatenate([tight, dup, wide]).astype(np.float32))
    data = data[rng.permutation(len(data))]
    monkeypatch.setenv("PQV_ASSIGN_GEMM", "2")
    for seed in (1, 2):
        want = oracle.build_index(data, n_clusters=kc, workers=2, max_iters=5, seed=seed).to_bytes()
        got = pqv.IndexBuilder(pqv.Corpus.upload(data)).n_clusters(kc).max_iters(5).seed(seed).workers(2).build().to_bytes()
        assert got == want, seed


def _fuzz_case(pqv, oracle, seed):
    rng = np.random.default_rng(seed)
    dim = int(rng.choice([64, 128, 192, 256, 320, 512, 1024])) if seed % 7 else 1024     # 1024: the widest f16-screened rows
    kc = int(rng.integers(2, 9))
    n = int(rng.integers(1200, 6000)) * kc
    k = int(rng.integers(1, 33)) if seed % 5 else int(rng.integers(33, 129))     # every fifth case: 32 < k <= 128
    nprobe = int(rng.integers(1, kc + 1))
    nq = int(rng.integers(1, 180))
    style = seed % 4
    if style == 0:
        data = rng.random((n, dim), dtype=np.float32)
    elif style == 1:                                   # coarse grid: many exactly equal distances
        data = (rng.integers(0, 3, size=(n, dim)) * 0.5).astype(np.float32)
    elif style == 2:                                   # clustered, different scales per cluster
        cen = rng.standard_normal((kc, dim)).astype(np.float32) * 3
        data = (cen[rng.integers(0, kc, n)] + rng.standard_normal((n, dim)).astype(np.float32)
                * rng.choice([0.01, 0.3, 2.0], size=(n, 1)).astype(np.float32)).astype(np.float32)
    else:                                              # duplicated rows
        base = rng.random((n // 4 + 1, dim), dtype=np.float32)
        data = base[rng.integers(0, len(base), n)]
    queries = data[rng.integers(0, n, nq)] + (rng.standard_normal((nq, dim)) * rng.choice([0.0, 0.05])).astype(np.float32)
    queries = queries.astype(np.float32)
    oidx = oracle.build_index(data, n_clusters=kc, workers=2, max_iters=3, seed=seed)
    s = pqv.Searcher(pqv.Index.from_bytes(oidx.to_bytes()), pqv.Corpus.upload(data))
    rows, dist, nf, nc = s.topk(queries, k, nprobe)
    orows, odist, onf, onc = oidx.topk_batch(data, queries, k, nprobe)
    assert (nc == onc).all(), seed
    assert (nf == onf).all(), seed
    # pqv_topk replays the reference heap when output distances tie: ids must match position by position -- over the n_found
    # entries of every query; the slots behind them are the ABI's padding (0xFFFFFFFF / +inf, include/pqv.h), the oracle binding's are
    # zeros (a query with fewer candidates than k: seed 7355, k = 104 on a 60-row list)
    live = np.arange(k)[None, :] < np.asarray(onf)[:, None]
    assert (_bits(dist)[live] == _bits(odist)[live]).all(), seed
    assert (rows[live] == orows[live]).all(), seed
    assert (rows[~live] == 0xFFFFFFFF).all() and np.isinf(dist[~live]).all(), seed
    return s.counters()["screened_pairs"]


@pytest.mark.parametrize("block", range(4))
def test_fuzz_screened_search_against_oracle(pqv, oracle, monkeypatch, block):
    """Random shapes / data styles (uniform, coarse grid with massive ties, multi-scale clusters,
    duplicated rows), random k <= 32, nprobe and batch size, with the MFMA screen forced
    wherever the lists are long enough: rows, distances and counts must equal the oracle exactly."""
    monkeypatch.setenv("PQV_RERANK_MODE", "tile")
    monkeypatch.setenv("PQV_TILE_FILTER", "2")
    # exact refinement of the seed thresholds: by rule (rows of >= 256 dims) in the even
    # blocks, for every shape in the odd ones (ties, duplicated rows and k up to 16 all pass through it there)
    monkeypatch.setenv("PQV_SEED_REFINE", "2" if block % 2 else "1")
    if block == 2:
        monkeypatch.setenv("PQV_WIDE_WAVES", "8")       # one 8-wave block per CU (the default where 96 queries do not fit twice)
    screened = 0
    for seed in range(1000 + 6 * block, 1000 + 6 * block + 6):
        screened += _fuzz_case(pqv, oracle, seed)
    assert screened > 0


@pytest.mark.parametrize("outlier", [1e15, 3e-3])
def test_f16_screen_with_extreme_value_ranges(pqv, oracle, monkeypatch, outlier):
    """f16 operands are scaled by the corpus maximum: one huge row pushes every other value to zero in f16
    (1e15), a corpus of tiny values is scaled up (3e-3); rows / queries the f16 image cannot represent must be
    evaluated exactly instead of being screened, and the results must stay those of the oracle."""
    rng = np.random.default_rng(321)
    n, dim, kc, k, nprobe, nq = 16000, 128, 6, 10, 3, 120
    data = (rng.random((n, dim), dtype=np.float32) * np.float32(min(1.0, outlier * 300 if outlier < 1 else 1.0))).astype(np.float32)
    if outlier > 1:
        data[rng.integers(0, n, 3)] = np.float32(outlier)                   # three enormous rows
    queries = data[rng.integers(0, n, nq)] * np.float32(1.001)
    queries[:4] *= np.float32(1e6)                                         # queries far outside the corpus range
    oidx = oracle.build_index(data, n_clusters=kc, workers=1, max_iters=3)
    monkeypatch.setenv("PQV_RERANK_MODE", "tile")
    monkeypatch.setenv("PQV_TILE_FILTER", "2")
    s = pqv.Searcher(pqv.Index.from_bytes(oidx.to_bytes()), pqv.Corpus.upload(data))
    rows, dist, nf, nc = s.topk(queries, k, nprobe)
    orows, odist, onf, onc = oidx.topk_batch(data, queries, k, nprobe)
    assert (nc == onc).all() and (nf == onf).all()
    assert (_bits(dist) == _bits(odist)).all()
    assert (rows == orows).all()


@pytest.mark.parametrize("case", ["huge_rows", "tiny_values", "constant", "signed_offset"])
@pytest.mark.parametrize("dim,waves", [(256, 4), (512, 4), (1024, 4), (512, 8), (1280, 8), (2048, 8)])
def test_i8_screen_with_extreme_value_ranges(pqv, oracle, monkeypatch, dim, waves, case):
    """int8 operands are per-LIST images of (x - centre_c) * S_c (the list's mid-range centre and scale): a few enormous
    rows squeeze every other row of their list into the same few levels (the residual bounds then make the screen useless
    there and everything is evaluated exactly, overflowing the candidate buffers), tiny values are scaled up, a constant
    corpus has no range at all, and data far from the origin relies on the centring.  Queries far outside the corpus range
    are clamped images (the clamp is monotone: the lower bound stays valid with the rounding residual alone).  Results
    must stay those of the oracle bit for bit."""
    rng = np.random.default_rng(77 + dim)
    n, kc, k, nprobe, nq = 14000, 5, 10, 3, 100
    data = rng.random((n, dim), dtype=np.float32)
    if case == "huge_rows":
        data[rng.integers(0, n, 3)] = np.float32(1e15)
    elif case == "tiny_values":
        data *= np.float32(1e-3)
    elif case == "constant":
        data[:] = np.float32(0.37)
        data[::2, 0] = np.float32(0.38)                  # two distinct rows so that k-means has something to split
    else:
        data = (data * np.float32(0.01) - np.float32(1000.0)).astype(np.float32)      # |x| ~ 1000, spread 0.01
    queries = (data[rng.integers(0, n, nq)] * np.float32(1.0005)).astype(np.float32)
    queries[:4] = queries[:4] * np.float32(1e6) + np.float32(3.0)                    # far outside the corpus range
    oidx = oracle.build_index(data, n_clusters=kc, workers=1, max_iters=3)
    monkeypatch.setenv("PQV_RERANK_MODE", "tile")
    monkeypatch.setenv("PQV_TILE_FILTER", "2")
    monkeypatch.setenv("PQV_WIDE_WAVES", str(waves))        # two 4-wave blocks per CU (up to 1024 dims) / one 8-wave block
    monkeypatch.setenv("PQV_I8_FORM", "2" if waves == 8 or dim == 1024 else "1")      # per-list residual images / one centre: both forms see every case
    if waves == 4 and dim == 512:
        monkeypatch.setenv("PQV_QUAD_WIDTH", "64")          # (the two-block form takes 96-query quads up to 768 dims)
    # grid forms: 4-wave blocks default to the work-item grid (quad x existing row chunk), 8-wave blocks to the 2-D grid
    if waves == 4 and dim == 256:
        monkeypatch.setenv("PQV_ITEM_GRID", "0")
    if waves == 8 and dim == 512:
        monkeypatch.setenv("PQV_ITEM_GRID", "2")
    s = pqv.Searcher(pqv.Index.from_bytes(oidx.to_bytes()), pqv.Corpus.upload(data))
    plan = s.describe(nq, k, nprobe)
    assert "int8 screen operands" in plan and f"{waves} waves per block" in plan, plan
    rows, dist, nf, nc = s.topk(queries, k, nprobe)
    orows, odist, onf, onc = oidx.topk_batch(data, queries, k, nprobe)
    assert (nc == onc).all() and (nf == onf).all()
    assert (_bits(dist) == _bits(odist)).all()
    _assert_topk_equal((rows, dist, nf), (orows, odist, onf), k)
    if case != "constant":
        assert (rows == orows).all()


@pytest.mark.parametrize("dim", [256, 768])
def test_i8_residual_images_clamped_queries_and_pair_pruning(pqv, oracle, dim):
    """Round 3: the int8 screen works on the IVF residual -- per list the image of x - centre_c at the list's own scale, per
    (query, probed list) pair the image of q - centre_c CLAMPED into the list's box -- and drops whole (query, list) pairs
    whose centre-distance bound (|q - centre_c| - radius_c)^2 exceeds the query's threshold.  Tight, well separated
    clusters of very different spreads make all of it bite: most probed pairs are pruned, every query is clamped in
    every list but its own; degenerate lists (one row; identical rows: half range 0) sit among them.  Queries inside a
    cluster, exactly on a row, on a centre, half way between two clusters and 100x outside everything: ids and distance
    bits must equal the oracle's with the pruning on and off, and the pruning must really remove work."""
    rng = np.random.default_rng(5 + dim)
    kc, per, k, nprobe = 12, 1500, 10, 6
    cen = (rng.standard_normal((kc, dim)) * 4.0).astype(np.float32)
    spread = rng.choice([0.02, 0.1, 0.5], size=kc).astype(np.float32)
    parts = [cen[c] + spread[c] * rng.standard_normal((per, dim)).astype(np.float32) for c in range(kc - 2)]
    parts.append(cen[kc - 2][None, :].repeat(40, axis=0))                       # identical rows: a list without any range
    parts.append((cen[kc - 1] + 50.0)[None, :].astype(np.float32))             # a single far row: a list of one
    data = np.ascontiguousarray(np.concatenate(parts).astype(np.float32))
    data = data[rng.permutation(len(data))]
    n = len(data)
    nq = 160
    queries = (cen[rng.integers(0, kc - 2, nq)] + 0.1 * rng.standard_normal((nq, dim))).astype(np.float32)
    queries[0:8] = data[rng.integers(0, n, 8)]                                  # exact hits
    queries[8:16] = cen[:8]                                                     # on a centre
    queries[16:24] = 0.5 * (cen[:8] + cen[1:9])                                 # between two clusters
    queries[24:28] = queries[24:28] * np.float32(100.0)                         # far outside everything
    queries[28] = cen[kc - 2]                                                   # the constant list's own point
    oidx = oracle.build_index(data, n_clusters=kc, workers=2, max_iters=10)
    index = pqv.Index.from_bytes(oidx.to_bytes())
    corpus = pqv.Corpus.upload(data)
    orows, odist, onf, onc = oidx.topk_batch(data, queries, k, nprobe)
    screened = {}
    for prune in (1, 0):
        for width in (0, 64):
            s = pqv.Searcher(index, corpus)
            s.set_option("rerank_mode", 2); s.set_option("tile_filter", 2); s.set_option("pair_prune", prune)
            if width == 64 and prune == 0:
                s.set_option("i8_form", 1)                # the one-centre form on the same data (by rule: residual here)
            if width:
                s.set_option("quad_width", width)
            assert "int8 screen operands" in s.describe(nq, k, nprobe)
            rows, dist, nf, nc = s.topk(queries, k, nprobe)
            assert ("about one centre" if (width == 64 and prune == 0) else "per-list residual") in s.describe(nq, k, nprobe)
            assert (nc == onc).all() and (nf == onf).all()
            assert (_bits(dist) == _bits(odist)).all(), (prune, width)
            _assert_topk_equal((rows, dist, nf), (orows, odist, onf), k)
            # single queries take the fused probe + their own bucketing: same answers
            r1, d1, _, _ = s.topk(queries[17:18], k, nprobe)
            assert (_bits(d1) == _bits(odist[17:18])).all()
            screened[(prune, width)] = s.counters()["screened_pairs"]
    assert screened[(1, 0)] < 0.6 * screened[(0, 0)], screened        # the pruning removes most of the far pairs' rows


@pytest.mark.parametrize("dim,sdim,op", [(96, 128, "f16"), (100, 128, "f16"), (52, 64, "f32"), (200, 256, "int8"), (300, 384, "f16"),
                                         (1000, 1024, "int8"), (36, 36, None), (130, 130, None)])
def test_dims_without_an_mfma_tiling_are_screened_on_zero_padded_rows(pqv, oracle, dim, sdim, op):
    """dim % 64 != 0 used to fall to the exact VALU kernels.  For dim % 4 == 0 the IVF-ordered rows (and per batch the
    queries) are stored zero-padded to the nearest tiling whose padding stays within a third of the row: the reference's
    distance walks 4 elements per step (index.rs:461-473), a padded group adds +0.0, so every distance is bit-identical
    and the screened path applies.  dim % 4 != 0 (the reference's scalar tail) and tiny rows keep the exact kernels.
    Both metrics, a candidate cap, a single query and k = 40 against the oracle."""
    rng = np.random.default_rng(900 + dim)
    n, kc, nprobe, nq = 9000, 6, 3, 70
    data = rng.random((n, dim), dtype=np.float32)
    queries = rng.random((nq, dim), dtype=np.float32)
    queries[::9] = data[rng.integers(0, n, len(queries[::9]))]
    oidx = oracle.build_index(data, n_clusters=kc, workers=2, max_iters=4)
    corpus = pqv.Corpus.upload(data)
    s = pqv.Searcher(pqv.Index.from_bytes(oidx.to_bytes()), corpus)
    plan = s.describe(nq, 10, nprobe)
    if op:
        assert f"zero-padded from {dim} to {sdim} dims" in plan and f"{op} screen operands" in plan and "wide_filter_kernel" in plan, plan
    else:
        assert "zero-padded" not in plan, plan
    for k in (10, 40):
        rows, dist, nf, nc = s.topk(queries, k, nprobe)
        orows, odist, onf, onc = oidx.topk_batch(data, queries, k, nprobe)
        assert (nc == onc).all() and (nf == onf).all()
        assert (_bits(dist) == _bits(odist)).all() and (rows == orows).all(), (dim, k)
    r1, d1, _, _ = s.topk(queries[3:4], 10, nprobe)
    o1 = oidx.topk_batch(data, queries[3:4], 10, nprobe)
    assert (r1 == o1[0]).all() and (_bits(d1) == _bits(o1[1])).all()
    # the DataFusion order (exec.rs:529-533) and a candidate cap in the middle of a list, against a direct evaluation
    cap = n // kc + 137
    rows, d2, nf, nc = s.topk(queries[:8], 10, nprobe, max_candidates=cap, metric=pqv.PQV_L2SQ_SEQ, sqrt_out=False)
    for q in range(8):
        cand = oidx.candidate_rows(queries[q], nprobe)[:cap]
        d = np.array([oracle.l2_seq(data[r], queries[q]) for r in cand], np.float32)
        order = np.lexsort((np.arange(len(cand)), d.view(np.uint32)))[:10]
        assert (_bits(d2[q, :len(order)]) == _bits(d[order])).all() and (rows[q, :len(order)] == cand[order]).all()
    # the padded copy is the searcher's own: the corpus still answers in its real dimension
    got = corpus.fetch_rows(np.array([5, 17], np.uint32)) if hasattr(corpus, "fetch_rows") else data[[5, 17]]
    assert np.array_equal(got, data[[5, 17]])


def test_topk_device_is_ordered_on_the_stream_it_is_given(pqv, oracle):
    """include/pqv.h: `hip_stream` NULL is the searcher's OWN non-blocking stream -- not the default stream, whose handle is 0 too.
    A consumer that must follow the call WITHOUT a synchronise passes the stream it runs on: an explicit one, or hipStreamLegacy
    (handle 1) for the default stream.  Each of the two: the search, then a dependent copy on that stream into buffers preset with
    garbage, with only a final device synchronise -- the copies must hold the oracle's answer (round 5: bench.py's lane 0 passed
    handle 0 and its exchange ran unordered behind the search)."""
    import torch
    dev = torch.device("cuda", 0)
    rng = np.random.default_rng(5)
    n, dim, kc, nprobe, nq, k = 200_000, 128, 64, 16, 512, 10
    data = rng.random((n, dim), dtype=np.float32)
    queries = rng.random((nq, dim), dtype=np.float32)
    oidx = oracle.build_index(data, n_clusters=kc, workers=2, max_iters=4)
    s = pqv.Searcher(pqv.Index.from_bytes(oidx.to_bytes()), pqv.Corpus.upload(data))
    orows, odist, onf, _ = oidx.topk_batch(data, queries, k, nprobe)
    q_t = torch.from_numpy(queries).to(dev)
    side = torch.cuda.Stream(device=dev)
    for stream_obj, handle in ((side, side.cuda_stream), (torch.cuda.default_stream(dev), 1)):
        r_t = torch.full((nq, k), -7, dtype=torch.int32, device=dev)
        d_t = torch.full((nq, k), -7.0, dtype=torch.float32, device=dev)
        torch.cuda.synchronize()
        with torch.cuda.stream(stream_obj):
            for _ in range(3):                      # (a few calls back to back: the copy must wait for the LAST one's kernels)
                s.topk_device(q_t.data_ptr(), nq, k, nprobe, r_t.data_ptr(), d_t.data_ptr(), stream=handle)
            r_c, d_c = r_t.clone(), d_t.clone()
        torch.cuda.synchronize()
        assert (r_c.cpu().numpy().view(np.uint32) == orows).all(), handle
        assert (_bits(d_c.cpu().numpy()) == _bits(odist)).all(), handle


def test_topk_device_flags_mark_every_query_that_needs_the_heap_replay(pqv, oracle):
    """pqv_topk_device_flags: the asynchronous device path + a per-query tie flag.  On tie-heavy data (a coarse grid,
    hundreds of equal distances) and on float data: every UNFLAGGED query must equal the reference position by
    position straight from the device; flagged ones are re-submitted to pqv_topk, whose heap replay must then give the
    reference's answer -- together: exact under ties without leaving the GPU for the queries that do not need it."""
    import torch
    dev = torch.device("cuda", 0)
    rng = np.random.default_rng(11)
    for style in ("grid", "float"):
        n, dim, kc, nprobe, nq = 6000, 16, 5, 3, 150
        data = (rng.integers(0, 3, size=(n, dim)).astype(np.float32) if style == "grid"
                else rng.random((n, dim), dtype=np.float32))
        queries = (rng.integers(0, 3, size=(nq, dim)).astype(np.float32) if style == "grid"
                   else rng.random((nq, dim), dtype=np.float32))
        oidx = oracle.build_index(data, n_clusters=kc, workers=1, max_iters=4)
        s = pqv.Searcher(pqv.Index.from_bytes(oidx.to_bytes()), pqv.Corpus.upload(data))
        q_t = torch.from_numpy(queries).to(dev)
        for k in (1, 10, 40):
            r_t = torch.empty((nq, k), dtype=torch.int32, device=dev)
            d_t = torch.empty((nq, k), dtype=torch.float32, device=dev)
            nf_t = torch.empty((nq,), dtype=torch.int32, device=dev)
            fl_t = torch.full((nq,), 7, dtype=torch.int32, device=dev)
            s.topk_device(q_t.data_ptr(), nq, k, nprobe, r_t.data_ptr(), d_t.data_ptr(), nf_t.data_ptr(),
                          stream=torch.cuda.current_stream().cuda_stream, d_tie_flags=fl_t.data_ptr())
            torch.cuda.synchronize()
            rows, dist, flags = r_t.cpu().numpy().view(np.uint32), d_t.cpu().numpy(), fl_t.cpu().numpy()
            orows, odist, onf, _ = oidx.topk_batch(data, queries, k, nprobe)
            assert set(np.unique(flags).tolist()) <= {0, 1}
            assert (_bits(dist) == _bits(odist)).all()                  # the distance multiset never depends on ties
            clean = flags == 0
            assert (rows[clean] == orows[clean]).all(), f"{style} k={k}: an unflagged query differs from the reference"
            # the flag is exactly "two adjacent entries among the first k + 1 merged results have equal output distance"
            o1r, o1d, o1n, _ = oidx.topk_batch(data, queries, k + 1, nprobe)
            want = np.array([bool((np.diff(o1d[q, :o1n[q]]) == 0).any()) for q in range(nq)])
            assert (want == ~clean).all(), (style, k, int(want.sum()), int((~clean).sum()))
            if style == "grid":
                assert (~clean).any()
            if (~clean).any():
                r2, d2, _, _ = s.topk(queries[~clean], k, nprobe)
                assert (r2 == orows[~clean]).all() and (_bits(d2) == _bits(odist[~clean])).all()


@pytest.mark.parametrize("metric_name", ["seq", "ref4"])
def test_rerank_device_matches_update_topk_heap(pqv, oracle, metric_name):
    """pqv_rerank_device: RecordBatches already resident on the device, the running top-k state never leaving it;
    the reference bench's batch shape (2048 rows, benches/query.rs:29) at dim 1024, k = 10 and k = 100, many batches
    through the pooled context (no allocation in steady state), against one heap over all rows (exec.rs:264-267)."""
    import torch
    from pq_vector_amd import _ffi
    dev = torch.device("cuda", 0)
    rng = np.random.default_rng(17)
    n, dim = 20480, 1024
    emb = rng.random((n, dim), dtype=np.float32)
    emb[5000:5040] = emb[100:140]                         # exact duplicates: ties resolve by arrival
    q = rng.random(dim, dtype=np.float32)
    order = rng.permutation(n).astype(np.uint32)
    metric = pqv.PQV_L2SQ_SEQ if metric_name == "seq" else pqv.PQV_L2SQ_REF4
    emb_t = torch.from_numpy(emb[order]).to(dev)          # arrival order
    ids_t = torch.from_numpy(order.astype(np.int32)).to(dev)
    q_t = torch.from_numpy(q).to(dev)
    for k in (10, 100):
        io_rows = torch.zeros((k,), dtype=torch.int32, device=dev)
        io_d2 = torch.zeros((k,), dtype=torch.float32, device=dev)
        io_cnt = torch.zeros((1,), dtype=torch.int32, device=dev)
        for b in range(0, n, 2048):
            rc = _ffi.lib().pqv_rerank_device(0, _ffi.vp(q_t.data_ptr()), _ffi.vp(emb_t[b:b + 2048].data_ptr()),
                                              _ffi.vp(ids_t[b:b + 2048].data_ptr()), min(2048, n - b), dim, k, metric,
                                              _ffi.vp(io_rows.data_ptr()), _ffi.vp(io_d2.data_ptr()), _ffi.vp(io_cnt.data_ptr()),
                                              _ffi.vp(torch.cuda.current_stream().cuda_stream))
            assert rc == 0, _ffi.lib().pqv_last_error()
        torch.cuda.synchronize()
        assert int(io_cnt.item()) == k
        got_r, got_d = io_rows.cpu().numpy().view(np.uint32), io_d2.cpu().numpy()
        if metric_name == "seq":
            orow, od2 = oracle.topk_df(emb, order, q, k)
        else:
            d2 = np.array([oracle.l2_ref4(q, emb[r]) for r in order], np.float32)
            sel = np.lexsort((np.arange(n), d2.view(np.uint32)))[:k]
            orow, od2 = order[sel], d2[sel]
        assert (_bits(got_d) == _bits(od2)).all()
        _assert_topk_equal((got_r[None, :], got_d[None, :], np.array([k])), (np.asarray(orow)[None, :], np.asarray(od2)[None, :], np.array([k])), k)


@pytest.mark.parametrize("dim,kc,nprobe", [(8, 5, 3), (64, 70, 70), (128, 316, 20), (96, 1000, 100), (768, 1024, 32), (130, 40, 7)])
def test_batched_centroid_probe_matches_find_closest_centroids(pqv, oracle, dim, kc, nprobe):
    """probe_rows_kernel (a lane per centroid, up to 8 queries per row read; dim % 4 == 0) against the oracle's
    find_closest_centroids (src/ivf/index.rs:130-149): the probe ORDER is pinned through a candidate cap in the middle
    of a probed list (the cap cuts the concatenation in probe order), the probed SET through n_candidates, and the
    single-query entry point returns the order itself.  Batch sizes cover every queries-per-lane variant; kc values
    that are not multiples of 64 / 256 exercise the padded transpose; dim 130 takes the stream_kernel fallback."""
    rng = np.random.default_rng(dim * 1000 + kc)
    n = max(4 * kc, 3000)
    data = rng.random((n, dim), dtype=np.float32)
    if kc >= 64:
        data[: kc // 2] = data[kc // 2: 2 * (kc // 2)]                  # duplicate rows: equal centroid distances can occur
    oidx = oracle.build_index(data, n_clusters=kc, workers=1, max_iters=2)
    s = pqv.Searcher(pqv.Index.from_bytes(oidx.to_bytes()), pqv.Corpus.upload(data))
    plan = s.describe(1024, 5, nprobe)
    assert ("probe_rows_kernel" in plan) == (dim % 4 == 0), plan
    assert "centroid probe: stream_kernel" in s.describe(1, 5, nprobe)      # a handful of queries keep the per-query stream
    s.set_option("probe_rows", 2)                                            # ... unless asked: every batch size below
    s0 = pqv.Searcher(pqv.Index.from_bytes(oidx.to_bytes()), pqv.Corpus.upload(data))
    s0.set_option("probe_rows", 0)
    assert "centroid probe: stream_kernel" in s0.describe(1024, 5, nprobe)
    k = 5
    for nq in (1, 3, 40, 130, 700, 2100):
        queries = rng.random((nq, dim), dtype=np.float32)
        queries[0] = data[0]
        for q in range(0, nq, max(1, nq // 5)):
            want = oidx.find_closest_centroids(queries[q], nprobe)
            got = s.probe(queries[q], nprobe)
            assert np.array_equal(np.asarray(got, np.uint32), np.asarray(want, np.uint32)), (nq, q)
        rows, dist, nf, nc = s.topk(queries, k, nprobe)
        r0, d0, nf0, nc0 = s0.topk(queries, k, nprobe)
        assert np.array_equal(rows, r0) and np.array_equal(_bits(dist), _bits(d0)) and np.array_equal(nc, nc0)
        cap = int(nc.min()) // 2 + 1
        rows_c, dist_c, nf_c, nc_c = s.topk(queries, k, nprobe, max_candidates=cap)
        for q in range(0, nq, max(1, nq // 7)):
            cand = oidx.candidate_rows(queries[q], nprobe)
            assert nc[q] == len(cand)
            cand = cand[:cap]
            d2 = np.array([oracle.l2_ref4(queries[q], data[r]) for r in cand], np.float32)
            order = np.lexsort((np.arange(len(cand)), d2.view(np.uint32)))[:k]
            assert (_bits(dist_c[q, :len(order)]) == _bits(np.sqrt(d2[order]))).all(), (nq, q)
            if len(set(d2[order].tolist())) == len(order) and (len(d2) <= k or np.sort(d2)[k] != np.sort(d2)[k - 1]):
                assert (rows_c[q, :len(order)] == cand[order]).all(), (nq, q)


@pytest.mark.parametrize("dim,waves", [(384, 4), (384, 8), (128, 4), (640, 8)])
def test_f16_block_forms_match_oracle(pqv, oracle, monkeypatch, dim, waves):
    """f16 screen operands in both block forms: two 4-wave blocks of 96-query quads per CU (rows up to 128 dims with the
    f32 originals staged next to the images, up to 384 dims without) and one 8-wave block (longer rows; forced for the
    short ones).  dim 384 and 640 are multiples of 128 but not of 256, so they never take the int8 form."""
    rng = np.random.default_rng(dim + waves)
    n, kc, k, nprobe, nq = 24000, 5, 10, 3, 230
    data = rng.random((n, dim), dtype=np.float32)
    queries = (data[rng.integers(0, n, nq)] + rng.standard_normal((nq, dim)).astype(np.float32) * 0.02).astype(np.float32)
    oidx = oracle.build_index(data, n_clusters=kc, workers=1, max_iters=3)
    monkeypatch.setenv("PQV_RERANK_MODE", "tile")
    monkeypatch.setenv("PQV_TILE_FILTER", "2")
    monkeypatch.setenv("PQV_WIDE_WAVES", str(waves))
    s = pqv.Searcher(pqv.Index.from_bytes(oidx.to_bytes()), pqv.Corpus.upload(data))
    plan = s.describe(nq, k, nprobe)
    assert "f16 screen operands" in plan and f"{waves} waves per block" in plan, plan
    if waves == 4:
        assert "quads of 96 queries" in plan, plan
    rows, dist, nf, nc = s.topk(queries, k, nprobe)
    orows, odist, onf, onc = oidx.topk_batch(data, queries, k, nprobe)
    assert (nc == onc).all() and (nf == onf).all()
    assert (_bits(dist) == _bits(odist)).all()
    assert (rows == orows).all()


def _popular_form(d):
    """How the dispatch description says the lists probed by more than 96 queries of the batch are read: 'list' = round 6's
    list_filter_kernel (rows stationary, ALL the list's pairs streamed past them: every such list read once), 'wide' = round 3's
    wide-quad instance (quads of 97..160 pairs on 32-row tiles), None = regular 96-query quads only."""
    if "list_filter_kernel" in d:
        assert "lists probed by more than 96 queries" in d, d
        return "list"
    return "wide" if "lists probed by 97..160 queries" in d else None


@pytest.mark.parametrize("dim,k,wide_rows", [(256, 10, 0), (768, 10, 512), (768, 1, 0), (512, 32, 1024)])
def test_wide_quads_stream_popular_lists_once(pqv, oracle, dim, k, wide_rows):
    """Round 3: a list that more than 96 queries of the batch probe used to be streamed once per 96-query quad.  With
    `wide_quads` (default) such a list's pairs are cut into quads of up to 160 and every quad of 97..160 pairs runs in the
    wide-quad instance of the filter kernel (8 waves on 32-row tiles, its own work-item table), the remainder (<= 96) in
    the regular one.  Few unequal lists and many queries give every case at once: lists with 0 .. 96, 97 .. 160,
    161 .. 256 (a wide quad + a regular one) and > 320 pairs (two wide quads), lists whose length is no multiple of 32,
    one- and multi-chunk lists.  Ids and distance bits must equal the oracle's with the instance on and off, the screened
    pair count must be the same, and the dispatch must say which form ran."""
    rng = np.random.default_rng(11 + dim + k)
    kc, nprobe, nq = 9, 3, 700
    sizes = [4001, 2977, 1500, 833, 700, 650, 517, 300, 45]
    cen = (rng.standard_normal((kc, dim)) * 2.0).astype(np.float32)
    data = np.concatenate([cen[c] + 0.3 * rng.standard_normal((m, dim)).astype(np.float32) for c, m in enumerate(sizes)])
    data = np.ascontiguousarray(data[rng.permutation(len(data))].astype(np.float32))
    # queries drawn near the centres with very unequal popularity
    pop = np.array([0.45, 0.2, 0.12, 0.08, 0.06, 0.04, 0.03, 0.015, 0.005])
    queries = (cen[rng.choice(kc, size=nq, p=pop)] + 0.4 * rng.standard_normal((nq, dim))).astype(np.float32)
    oidx = oracle.build_index(data, n_clusters=kc, workers=2, max_iters=10)
    index = pqv.Index.from_bytes(oidx.to_bytes())
    corpus = pqv.Corpus.upload(data)
    orows, odist, onf, onc = oidx.topk_batch(data, queries, k, nprobe)
    probes = np.stack([np.asarray(oidx.find_closest_centroids(q, nprobe)) for q in queries])
    per_list = np.bincount(probes.ravel(), minlength=kc)
    assert per_list.max() > 320 and ((per_list % 160) > 96).any() and ((per_list > 0) & (per_list <= 96)).any(), per_list
    # Round 6: with `list_once` = 1 such lists go to list_filter_kernel -- one quad of ALL the list's pairs (here up to 700 x 0.45 +
    # what the other centres send), rows stationary in registers: read once whatever the pair count (built, bit-exact, and measured
    # slower than the wide-quad instance: an option, not the default).  Not together with the deferred evaluation (short lists of
    # >= 512-dim rows, k > 64), which keeps the wide-quad instance: the list form is checked with it off.
    screened = {}
    for form in ("list", "wide", None):
        s = pqv.Searcher(index, corpus)
        s.set_option("rerank_mode", 2); s.set_option("tile_filter", 2)
        s.set_option("wide_quads", 0 if form is None else 1); s.set_option("list_once", 1 if form == "list" else 0)
        if form == "list":
            if k > 64:
                continue                # (k > 64 always defers)
            s.set_option("defer", 0)
        if wide_rows:
            s.set_option("wide_quad_rows", wide_rows)
        d = s.describe(nq, k, nprobe)
        assert "int8 screen operands" in d and "quads of 96 queries" in d, d
        assert _popular_form(d) == form, (form, d)
        for _ in range(2):             # (twice: the second call meets the first one's scratch)
            rows, dist, nf, nc = s.topk(queries, k, nprobe)
            assert (nc == onc).all() and (nf == onf).all()
            assert (_bits(dist) == _bits(odist)).all(), form
            _assert_topk_equal((rows, dist, nf), (orows, odist, onf), k)
        screened[form] = s.counters()["screened_pairs"] // 2
    # (pairs whose centre-distance bound exceeds the query's threshold AT THE TIME a block looks are dropped, so the count
    #  depends on the order the blocks run in: the forms agree closely, not exactly)
    for form in screened:
        assert abs(screened[form] - screened[None]) <= 0.02 * screened[None], screened


@pytest.mark.parametrize("opts", [{"wide_quads": 2}, {"wide_quads": 0, "xcd_items": 1}, {"wide_quads": 0, "xcd_items": 0}, {"xcd_items": 3},
                                  {"xcd_items": 0}, {"fork_wide": 1}, {"drain_min": 8}, {"fork_wide": 1, "drain_min": 16, "xcd_items": 3},
                                  {"list_once": 1}, {"list_once": 1, "chunk_major": 0}, {"list_once": 1, "wide_quad_rows": 256, "xcd_items": 0}])
def test_round5_quad_scheduling_options_never_change_a_result(pqv, oracle, opts):
    """Round 5: how a batch's quads are cut and placed -- wide-quad instance or regular quads only, a level's work items filled
    column by column so that the quads of one list run back to back on one XCD, the wide-quad launch forked onto a side stream,
    exact evaluations started before a wave's queue is full -- is scheduling: ids, distance bits and counters must equal the
    oracle's under every combination, on a batch with lists of 0..96, 97..160, 161..320 and > 320 pairs."""
    rng = np.random.default_rng(1234)
    dim, k, kc, nprobe, nq = 256, 10, 9, 3, 700
    sizes = [4001, 2977, 1500, 833, 700, 650, 517, 300, 45]
    cen = (rng.standard_normal((kc, dim)) * 2.0).astype(np.float32)
    data = np.concatenate([cen[c] + 0.3 * rng.standard_normal((m, dim)).astype(np.float32) for c, m in enumerate(sizes)])
    data = np.ascontiguousarray(data[rng.permutation(len(data))].astype(np.float32))
    pop = np.array([0.45, 0.2, 0.12, 0.08, 0.06, 0.04, 0.03, 0.015, 0.005])
    queries = (cen[rng.choice(kc, size=nq, p=pop)] + 0.4 * rng.standard_normal((nq, dim))).astype(np.float32)
    oidx = oracle.build_index(data, n_clusters=kc, workers=2, max_iters=10)
    index = pqv.Index.from_bytes(oidx.to_bytes())
    corpus = pqv.Corpus.upload(data)
    orows, odist, onf, onc = oidx.topk_batch(data, queries, k, nprobe)
    s = pqv.Searcher(index, corpus)
    s.set_option("rerank_mode", 2); s.set_option("tile_filter", 2); s.set_option("wide_rows", 512); s.set_option("wide_quad_rows", 512)
    for name, value in opts.items():
        s.set_option(name, value)
    d = s.describe(nq, k, nprobe)
    want = None if opts.get("wide_quads", 1) == 0 else "wide" if (opts.get("wide_quads", 1) == 2 or opts.get("list_once", 0) == 0) else "list"
    assert "int8 screen operands" in d and _popular_form(d) == want, d
    for _ in range(2):
        rows, dist, nf, nc = s.topk(queries, k, nprobe)
        assert (nc == onc).all() and (nf == onf).all()
        assert (_bits(dist) == _bits(odist)).all(), opts
        _assert_topk_equal((rows, dist, nf), (orows, odist, onf), k)


def test_quads_follow_the_previous_batch_shape(pqv, oracle):
    """Round 5: with the default `wide_quads` = 1 the NEXT batch takes regular quads only once a batch has shown that most rows of
    its popular lists (> 96 pairs) sit in lists of > 160 pairs (several quads either way: they share the rows through one XCD's
    L2), and goes back when a batch shows the opposite.  The hint is read without synchronisation and never changes a result."""
    import torch
    rng = np.random.default_rng(4321)
    dim, k, kc, nprobe = 256, 10, 8, 2
    sizes = [3000, 2500, 2000, 1500, 900, 700, 500, 300]
    cen = (rng.standard_normal((kc, dim)) * 2.0).astype(np.float32)
    data = np.concatenate([cen[c] + 0.3 * rng.standard_normal((m, dim)).astype(np.float32) for c, m in enumerate(sizes)])
    data = np.ascontiguousarray(data[rng.permutation(len(data))].astype(np.float32))
    oidx = oracle.build_index(data, n_clusters=kc, workers=2, max_iters=10)
    index = pqv.Index.from_bytes(oidx.to_bytes())
    corpus = pqv.Corpus.upload(data)
    s = pqv.Searcher(index, corpus)
    s.set_option("rerank_mode", 2); s.set_option("tile_filter", 2)
    # batch A: 1200 queries piled on two centres -> lists of several hundred pairs each; batch B: 8 x 130 queries -> 97..160 pairs per list
    qa = (cen[rng.choice(2, size=1200)] + 0.4 * rng.standard_normal((1200, dim))).astype(np.float32)
    qb = (cen[np.repeat(np.arange(kc), 40)] + 0.05 * rng.standard_normal((kc * 40, dim))).astype(np.float32)
    lens = np.diff(index.list_offsets.astype(np.int64))

    def regular_next(queries):          # the rule of prefer_regular (api.cpp) on this batch's pairs per list
        per = np.bincount(np.concatenate([np.asarray(oidx.find_closest_centroids(q, nprobe)) for q in queries]), minlength=kc)
        multi, pop = int(lens[per > 160].sum()), int(lens[per > 96].sum())
        return pop > 0 and 2 * multi > pop
    assert regular_next(qa) and not regular_next(qb)
    # batch C (round 6, the advisor's middle regime): the lists of 97..160 pairs hold most of the popular rows while one list of > 160
    # pairs holds >= n / 32 rows.  While the wide-quad instance is active the scan's quad width is the 160-wide cut; counted against
    # THAT width the statistic read "every popular row sits in a list of > 160 pairs" and the plan flipped on every call.
    cnts = [60, 60, 60, 60, 0, 0, 100, 0]
    qc = (cen[np.repeat(np.arange(kc), cnts)] + 0.05 * rng.standard_normal((sum(cnts), dim))).astype(np.float32)
    per_c = np.bincount(np.concatenate([np.asarray(oidx.find_closest_centroids(q, nprobe)) for q in qc]), minlength=kc)
    assert not regular_next(qc) and int(lens[per_c > 160].sum()) * 32 >= len(data) and ((per_c > 96) & (per_c <= 160)).any()
    for queries in (qa, qb, qc, qc, qa):
        nq = len(queries)
        want_wide_next = not regular_next(queries)
        orows, odist, onf, onc = oidx.topk_batch(data, queries, k, nprobe)
        rows, dist, nf, nc = s.topk(queries, k, nprobe)
        assert (nc == onc).all() and (nf == onf).all() and (_bits(dist) == _bits(odist)).all()
        _assert_topk_equal((rows, dist, nf), (orows, odist, onf), k)
        torch.cuda.synchronize()
        assert ("lists probed by 97..160 queries" in s.describe(nq, k, nprobe)) == want_wide_next, (nq, s.describe(nq, k, nprobe))


@pytest.mark.parametrize("case", ["levels", "clusters"])
def test_chunk_major_work_items(pqv, oracle, case):
    """Round 3: the filter kernel's work items are numbered chunk-major (row chunk 0 of every quad first) by pair_scan_kernel.
    `levels`: lists of more row chunks than the table has levels (64) in both instances -- the last level then holds the rest;
    `clusters`: more clusters than the scan's 1024 threads (several rounds).  Ids, distance bits and the screened pair count
    must equal the list-major numbering's and the oracle's."""
    rng = np.random.default_rng(77)
    if case == "levels":
        dim, kc, nprobe, nq, k = 256, 5, 2, 500, 10
        sizes = [36000, 17000, 3000, 900, 400]
        cen = (rng.standard_normal((kc, dim)) * 2.0).astype(np.float32)
        data = np.concatenate([cen[c] + 0.1 * rng.standard_normal((m, dim)).astype(np.float32) for c, m in enumerate(sizes)])
        data = np.ascontiguousarray(data[rng.permutation(len(data))].astype(np.float32))
        pop = np.array([0.5, 0.3, 0.1, 0.06, 0.04])
        queries = (cen[rng.choice(kc, size=nq, p=pop)] + 0.4 * rng.standard_normal((nq, dim))).astype(np.float32)
        oidx = oracle.build_index(data, n_clusters=kc, workers=2, max_iters=10)
        opts = {"wide_rows": 256, "wide_quad_rows": 512}
    else:
        dim, kc, nprobe, nq, k = 64, 1100, 8, 512, 10
        data = rng.random((kc * 210, dim), dtype=np.float32)
        queries = rng.random((nq, dim), dtype=np.float32)
        oidx = oracle.build_index(data, n_clusters=kc, workers=4, max_iters=1)
        opts = {}
    lens = np.diff(oidx.list_off.astype(np.int64))
    if case == "levels":
        assert lens.max() > 64 * 512, lens
    index = pqv.Index.from_bytes(oidx.to_bytes())
    corpus = pqv.Corpus.upload(data)
    orows, odist, onf, onc = oidx.topk_batch(data, queries, k, nprobe)
    screened = {}
    for cm in (1, 0):
        s = pqv.Searcher(index, corpus)
        s.set_option("rerank_mode", 2); s.set_option("tile_filter", 2); s.set_option("chunk_major", cm)
        for name, v in opts.items():
            s.set_option(name, v)
        d = s.describe(nq, k, nprobe)
        assert "wide_filter_kernel" in d, d
        if case == "levels":
            assert _popular_form(d) == "wide", d
        rows, dist, nf, nc = s.topk(queries, k, nprobe)
        assert (nc == onc).all() and (nf == onf).all()
        assert (_bits(dist) == _bits(odist)).all(), cm
        _assert_topk_equal((rows, dist, nf), (orows, odist, onf), k)
        screened[cm] = s.counters()["screened_pairs"]
    assert abs(screened[1] - screened[0]) <= 0.02 * screened[0], screened      # (order-dependent pair pruning, as above)


@pytest.mark.parametrize("case", ["finite", "an_infinity", "norms_overflow", "eager"])
def test_creation_builds_the_int8_copy_first_and_takes_the_norms_on_demand(pqv, oracle, monkeypatch, case):
    """Creation order of round 6: where the int8 copy is built at creation, its own first pass (the per-dimension extremes) vouches for
    the data and the pass over the rows' norms waits for a call that reads them -- here the K = 100 call on ~1100-row lists, which
    screens with f16 operands.  Data that pass cannot vouch for (an infinity, values beyond 2^74) must take
    the old order: norms first, no int8 copy.  PQV_EAGER_NORMS=1 is the old order by request.  The answers never change."""
    rng = np.random.default_rng(4242)
    n, dim, kc, nprobe, nq = 72000, 256, 64, 6, 256
    data = rng.random((n, dim), dtype=np.float32)
    if case == "an_infinity":
        data[123, 7] = np.inf
    elif case == "norms_overflow":
        data[789] = np.float32(1e25)                     # finite, but beyond what the f16 / int8 images are scaled for
    if case == "eager":
        monkeypatch.setenv("PQV_EAGER_NORMS", "1")
    queries = (data[rng.integers(1000, n, nq)] * np.float32(1.001)).astype(np.float32)
    oidx = oracle.build_index(rng.random((n, dim), dtype=np.float32) if case == "an_infinity" else data, n_clusters=kc, workers=1, max_iters=2)
    s = pqv.Searcher(pqv.Index.from_bytes(oidx.to_bytes()), pqv.Corpus.upload(data))
    plan10 = s.describe(nq, 10, nprobe)
    if case in ("finite", "eager"):
        assert "int8 screen operands" in plan10, plan10
    else:
        assert "int8 screen operands" not in plan10, plan10
    for k in (10, 100, 10):                              # int8 (or its fallback) -> f16 operands (the norms arrive now) -> int8 again
        rows, dist, nf, nc = s.topk(queries, k, nprobe)
        orows, odist, onf, onc = oidx.topk_batch(data, queries, k, nprobe)
        assert (nc == onc).all() and (nf == onf).all()
        live = ~np.isnan(odist)
        assert (_bits(dist)[live] == _bits(odist)[live]).all()
        _assert_topk_equal((rows, dist, nf), (orows, odist, onf), k)
    if case == "finite":
        assert "f16 screen operands" in s.describe(nq, 100, nprobe) or "f16" in s.describe(nq, 100, nprobe)
