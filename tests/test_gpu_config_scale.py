"""GPU parity at BASELINE.json's configuration PARAMETERS with the library's NATURAL dispatch (no path forcing):

  C3 / C4 shape: dim 768, n_clusters 1024, nprobe 32, k 10, 1024-query batches -- on 1 M rows (the oracle's
                 brute-force cross-check of a 10 M-row corpus would not finish in a test; the 10 M / 12.5 M row
                 runs are checked inside bench.py's cpu_baseline leg on every bench run);
  C5 shape:      1 M x 1536, cosine, 1024-query batch, against an f64 brute force.

Bar as everywhere: bit-identical row ids and distances for the IVF path (oracle built from the GPU index blob,
exactly as test_full_size_c2_properties does), 1e-4 relative for the cosine MFMA extension."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _bits(a):
    return np.ascontiguousarray(a, dtype=np.float32).view(np.uint32)


def _uniform24(rng, n, dim):
    # the bench recipe (benches/bench_util.rs:40): 24-bit uniform [0, 1); generated in slabs to bound memory
    out = np.empty((n, dim), dtype=np.float32)
    step = max(1, (1 << 26) // dim)
    for s in range(0, n, step):
        e = min(n, s + step)
        out[s:e] = rng.integers(0, 1 << 24, size=(e - s, dim), dtype=np.int32).astype(np.float32) * np.float32(1.0 / (1 << 24))
    return out


@pytest.fixture(scope="module")
def c3_shape(pqv):
    n, dim, kc = 1_000_000, 768, 1024
    rng = np.random.default_rng(1234)
    data = _uniform24(rng, n, dim)
    corpus = pqv.Corpus.upload(data)
    index = pqv.IndexBuilder(corpus).n_clusters(kc).max_iters(20).seed(42).workers(8).build()
    return data, corpus, index, rng


@pytest.mark.timeout(1200)
def test_c3_parameters_natural_dispatch(pqv, oracle, c3_shape):
    data, corpus, index, rng = c3_shape
    n, dim = data.shape
    kc, k, nprobe, nq = 1024, 10, 32, 1024
    # index invariants at this size (the >= 512-centroid build takes the MFMA-screened assignment by itself)
    off, rows = index.list_offsets, index.list_rows
    assert index.n_clusters == kc and int(off[-1]) == n
    assert np.array_equal(np.sort(rows), np.arange(n, dtype=np.uint32))
    blob = index.to_bytes()
    assert pqv.IndexBuilder(corpus).n_clusters(kc).max_iters(20).seed(42).workers(8).build().to_bytes() == blob

    queries = _uniform24(np.random.default_rng(7), nq, dim)
    queries[:64] = data[rng.choice(n, 64, replace=False)]            # some self-queries
    s = pqv.Searcher(index, corpus)
    plan = s.describe(nq, k, nprobe)
    assert "wide_filter_kernel" in plan and "int8 screen operands" in plan and "4 waves per block" in plan, plan
    rows_t, dist_t, nf, nc = s.topk(queries, k, nprobe)                 # host API: exact under ties
    assert (nf == k).all()
    assert (np.diff(dist_t.astype(np.float64), axis=1) >= 0).all()
    assert (dist_t[:64, 0] == 0).all() and (np.abs(data[rows_t[:64, 0]] - queries[:64]).max(axis=1) == 0).all()
    for q in range(nq):
        assert len(set(rows_t[q].tolist())) == k
    # the asynchronous device path returns the same thing (no ties in float data)
    import torch
    dev = torch.device("cuda", 0)
    q_t = torch.from_numpy(queries).to(dev)
    r_t = torch.empty((nq, k), dtype=torch.int32, device=dev)
    d_t = torch.empty((nq, k), dtype=torch.float32, device=dev)
    nc_t = torch.empty((nq,), dtype=torch.int64, device=dev)
    before = s.counters()
    s.topk_device(q_t.data_ptr(), nq, k, nprobe, r_t.data_ptr(), d_t.data_ptr(), 0, nc_t.data_ptr(),
                  stream=torch.cuda.current_stream().cuda_stream)
    torch.cuda.synchronize()
    assert np.array_equal(r_t.cpu().numpy().view(np.uint32), rows_t)
    assert np.array_equal(_bits(d_t.cpu().numpy()), _bits(dist_t))
    assert np.array_equal(nc_t.cpu().numpy().astype(np.uint64), nc)
    after = s.counters()
    # plan metrics are kept on the device path too (src/df_vector/index_exec.rs:289-299)
    assert after["candidate_rows"] - before["candidate_rows"] == int(nc.sum())
    assert after["embeddings_fetched"] - before["embeddings_fetched"] == int(nc.sum())
    # the other paths agree bit for bit: int8 with 64-query quads, int8 in one 8-wave block per CU (96- and 128-query quads), f16 operands (8-wave 96-query and 4-wave 32-query quads),
    # f32 operands, the exact tile kernel, the stream kernel
    assert "about one centre" in plan, plan              # uniform data: per-list scales buy nothing, one image per query
    for opts in ({"i8_form": 2}, {"i8_form": 2, "pair_prune": 0}, {"quad_width": 64}, {"item_grid": 0}, {"seed_refine": 0}, {"wide_waves": 8}, {"wide_waves": 8, "item_grid": 2}, {"wide_waves": 8, "quad_width": 128}, {"screen_i8": 0}, {"screen_i8": 0, "wide_waves": 4}, {"screen_i8": 0, "screen_f16": 0},
                 {"tile_filter": 0, "rerank_mode": 2}, {"rerank_mode": 1}):
        s2 = pqv.Searcher(index, corpus)
        for name, v in opts.items():
            s2.set_option(name, v)
        sel = slice(0, 128) if opts.get("rerank_mode") else slice(0, nq)
        r2, d2, nf2, nc2 = s2.topk(queries[sel], k, nprobe)
        assert np.array_equal(r2, rows_t[sel]) and np.array_equal(_bits(d2), _bits(dist_t[sel])), opts
        assert np.array_equal(nc2, nc[sel])
    # oracle spot check on the same index (bit-exact), queries spread over the batch
    oidx = oracle.index_from_bytes(blob)
    sel = np.arange(0, nq, 32)
    orows, odist, onf, onc = oidx.topk_batch(data, queries[sel], k, nprobe)
    assert (rows_t[sel] == orows).all() and (_bits(dist_t[sel]) == _bits(odist)).all()
    assert (nc[sel] == onc).all() and (nf[sel] == onf).all()
    # small batches and single queries take the same kernels (any batch size) and must agree as well
    for b in (1, 3, 40):
        r3, d3, _, _ = s.topk(queries[100:100 + b], k, nprobe)
        assert np.array_equal(r3, rows_t[100:100 + b]) and np.array_equal(_bits(d3), _bits(dist_t[100:100 + b]))
    # (a single query is bucketed and quantised by the probe merge itself; the general three-launch form agrees)
    s3 = pqv.Searcher(index, corpus)
    s3.set_option("single_bucket", 3)          # probe + merge + bucketing in one block (the rule keeps that for small tables)
    r6, d6, _, nc6 = s3.topk(queries[5:6], k, nprobe)
    assert np.array_equal(r6, rows_t[5:6]) and np.array_equal(_bits(d6), _bits(dist_t[5:6])) and nc6[0] == nc[5]
    s1 = pqv.Searcher(index, corpus)
    s1.set_option("single_bucket", 0)
    for q in (0, 100, 777):
        r4, d4, _, nc4 = s1.topk(queries[q:q + 1], k, nprobe)
        r5, d5, _, nc5 = s.topk(queries[q:q + 1], k, nprobe)
        assert np.array_equal(r4, r5) and np.array_equal(_bits(d4), _bits(d5)) and np.array_equal(nc4, nc5)
        assert np.array_equal(r5, rows_t[q:q + 1])


@pytest.mark.timeout(1200)
def test_c3_parameters_candidate_cap_and_k(pqv, oracle, c3_shape):
    """Same corpus: a candidate cap in the middle of a probed list and k up to 100, against the oracle."""
    data, corpus, index, rng = c3_shape
    nprobe = 32
    queries = _uniform24(np.random.default_rng(11), 96, data.shape[1])
    s = pqv.Searcher(index, corpus)
    oidx = oracle.index_from_bytes(index.to_bytes())
    for k, cap in ((10, 0), (10, 12345), (100, 0)):
        rows_t, dist_t, nf, nc = s.topk(queries, k, nprobe, max_candidates=cap)
        for q in range(0, 96, 8):
            cand = oidx.candidate_rows(queries[q], nprobe)
            if cap:
                cand = cand[:cap]
            d2 = np.array([oracle.l2_ref4(queries[q], data[r]) for r in cand], np.float32)
            order = np.lexsort((np.arange(len(cand)), d2.view(np.uint32)))[:k]
            assert (rows_t[q, :len(order)] == cand[order]).all(), (k, cap, q)
            assert (_bits(dist_t[q, :len(order)]) == _bits(np.sqrt(d2[order]))).all()


@pytest.mark.timeout(1800)
def test_c5_parameters_cosine_batch(pqv):
    """BASELINE configs[4] shape: 1 M x 1536, cosine, one 1024-query batch through pqv_brute_topk."""
    n, dim, nq, k = 1_000_000, 1536, 1024, 10
    rng = np.random.default_rng(5)
    data = _uniform24(rng, n, dim)
    data -= np.float32(0.5)                                   # ada-002-like: signed components
    queries = _uniform24(np.random.default_rng(6), nq, dim) - np.float32(0.5)
    corpus = pqv.Corpus.upload(data)
    rows, dist, nf = corpus.brute_topk(queries, k, pqv.PQV_COSINE)
    assert (nf == k).all() and (np.diff(dist.astype(np.float64), axis=1) >= -1e-7).all()
    vn = np.sqrt((data.astype(np.float64) ** 2).sum(axis=1))
    # every returned distance is right (f64 recomputation of the returned rows, all 1024 queries) ...
    for q in range(nq):
        x = data[rows[q]].astype(np.float64)
        qq = queries[q].astype(np.float64)
        d = 1.0 - (x @ qq) / (vn[rows[q]] * np.sqrt((qq ** 2).sum()))
        assert np.allclose(d, dist[q], rtol=1e-4, atol=1e-6), q
        assert len(set(rows[q].tolist())) == k
    # ... and nothing closer was missed: full f64 brute force for a sample of the batch
    sel = np.arange(0, nq, 32)
    qs = queries[sel].astype(np.float64)
    full = np.empty((len(sel), n))
    for s0 in range(0, n, 100_000):
        blk = data[s0:s0 + 100_000].astype(np.float64)
        full[:, s0:s0 + 100_000] = 1.0 - (qs @ blk.T) / (np.sqrt((qs ** 2).sum(axis=1))[:, None] * vn[None, s0:s0 + 100_000])
    order = np.argsort(full, axis=1, kind="stable")[:, :k]
    od = np.take_along_axis(full, order, axis=1)
    for i, q in enumerate(sel):
        assert np.allclose(od[i], dist[q], rtol=1e-4, atol=1e-6)
        kth = od[i, -1]
        tol = 1e-4 * max(abs(kth), 1e-3) + 1e-6
        clearly_in = order[i][od[i] < kth - tol]
        assert set(clearly_in.tolist()) <= set(rows[q].tolist())


# ---------------------------------------------------------------------------------------
# index BUILD parity at configuration scale: blob == oracle.build_index, not only determinism
# ---------------------------------------------------------------------------------------
@pytest.mark.timeout(1800)
def test_c2_full_size_build_blob_equals_oracle(pqv, oracle):
    """BASELINE configs[1] at full size: 1 M x 128, n_clusters 100, max_iters 20, seed 42 (parquet.rs:37-38), workers 8.
    sample_size = 50 000 (index.rs:172-174) drawn by index::sample's INPLACE branch, k-means++ over the whole sample,
    20 Lloyd iterations, final assignment of all 1 M rows -- the blob must equal the CPU oracle's byte for byte."""
    n, dim, kc = 1_000_000, 128, 100
    data = _uniform24(np.random.default_rng(1234), n, dim)
    corpus = pqv.Corpus.upload(data)
    index = pqv.IndexBuilder(corpus).n_clusters(kc).max_iters(20).seed(42).workers(8).build()
    rng_state = oracle.rng(42)
    _, branch = oracle.index_sample(rng_state, n, 50_000)
    assert branch == 1                                           # inplace
    oidx = oracle.build_index(data, n_clusters=kc, max_iters=20, seed=42, workers=8)
    gb, ob = index.to_bytes(), oidx.to_bytes()
    assert len(gb) == len(ob) == 8 + kc * dim * 4 + kc * 4 + n * 4
    assert gb[:8 + kc * dim * 4] == ob[:8 + kc * dim * 4], "centroid bits differ"
    assert gb == ob


@pytest.mark.timeout(1800)
def test_rejection_branch_sample_build_blob_equals_oracle(pqv, oracle):
    """index::sample's REJECTION branch (index.rs:222-242 -> rand's sample_rejection): taken when the 100 000-row sample
    is small against n -- C3 / C4 / C5 all build through it.  4.2 M x 4-dim rows keep the oracle build in seconds; the
    drawn rows decide every centroid, so blob equality pins the branch, its u32 draws and the k-means++ subset after it."""
    n, dim, kc = 4_200_000, 4, 24
    rng = np.random.default_rng(77)
    data = rng.random((n, dim), dtype=np.float32)
    data[::7] += np.float32(2.0)                                  # two blobs, so the lists are not all alike
    _, branch = oracle.index_sample(oracle.rng(42), n, 100_000)
    assert branch == 2                                           # rejection
    corpus = pqv.Corpus.upload(data)
    for workers in (8, 3):
        index = pqv.IndexBuilder(corpus).n_clusters(kc).max_iters(20).seed(42).workers(workers).build()
        oidx = oracle.build_index(data, n_clusters=kc, max_iters=20, seed=42, workers=workers)
        assert index.to_bytes() == oidx.to_bytes(), workers


@pytest.mark.timeout(1800)
def test_c3_shape_screened_final_assignment_matches_oracle_on_a_slice(pqv, oracle, c3_shape):
    """C3's parameters (dim 768, 1024 centroids): the final assignment of the build runs through the MFMA screen
    (ScreenedAssign) -- bounds, survivors, exact re-evaluation.  For 100 000 rows spread over the corpus the cluster the
    GPU put a row into must be the oracle's nearest_centroid (index.rs:244-257: strict '<', lowest index wins) under
    the GPU-built centroids."""
    from concurrent.futures import ThreadPoolExecutor
    data, corpus, index, _ = c3_shape
    n = data.shape[0]
    off, rows = index.list_offsets.astype(np.int64), index.list_rows
    cluster_of = np.empty(n, dtype=np.uint32)
    for c in range(index.n_clusters):
        cluster_of[rows[off[c]:off[c + 1]]] = c
    oidx = oracle.index_from_bytes(index.to_bytes())
    sel = np.arange(0, n, 10)[:100_000]

    def nearest(lo):
        return [int(oidx.find_closest_centroids(data[r], 1)[0]) for r in sel[lo:lo + 500]]
    with ThreadPoolExecutor(max_workers=32) as ex:
        want = np.array([c for part in ex.map(nearest, range(0, len(sel), 500)) for c in part], dtype=np.uint32)
    bad = np.nonzero(cluster_of[sel] != want)[0]
    assert len(bad) == 0, (len(bad), sel[bad[:5]], cluster_of[sel][bad[:5]], want[bad[:5]])


def test_c1_vldb_standin_through_the_path_builders(pqv, tmp_path):
    """BASELINE configs[0] as BASELINE.md defines its stand-in: 1 024 x 4096 `embedding` List<f32> + Utf8 `title` in a Parquet
    file; IndexBuilder(path, "embedding").build_inplace() with the default n_clusters (ceil(sqrt(n)) = 32, index.rs:161-167);
    TopkBuilder(path, q).k(10).nprobe(5).search() per query (search.rs:49-81; queries = rows of the file as in
    src/df_vector/tests.rs:106-149, and fresh vectors).  bench.c1_config is the very leg the default bench line runs as
    configs.c1: blob and answers (row ids + distance bits) == oracle, the file still reads as Parquet and carries the blob."""
    import argparse
    import torch
    import bench
    dev = torch.device("cuda", 0)
    rec = bench.c1_config(argparse.Namespace(no_cpu=False), pqv, torch, dev, 0, n_queries=64, keep_dir=str(tmp_path))
    assert rec["n_clusters"] == 32
    par = rec["parity"]
    assert par["index_blob_identical"] and par["topk_rows_and_distance_bits_identical"] and par["file_round_trip"] and par["ok"]
    assert par["queries_checked"] == 64 and rec["value"] > 0
    # a query that IS a row of the file finds itself first, at distance 0 (the reference's vldb query is row 0 of the file)
    import pyarrow.parquet as pq
    path = str(tmp_path / "vldb_standin.parquet")
    row0 = np.asarray(pq.read_table(path, columns=["embedding"]).column("embedding")[0].as_py(), dtype=np.float32)
    hits = pqv.TopkBuilder(path, row0).k(10).nprobe(5).search()
    assert len(hits) == 10 and hits[0].row_idx == 0 and hits[0].distance == 0.0
    summary = bench.summarize_config(rec)
    assert summary["parity_ok"] is True and summary["n_clusters"] == 32 and "config" not in summary


_LISTS_SCRIPT = r"""
import sys, os, hashlib
sys.path.insert(0, os.getcwd())
import numpy as np
import pq_vector_amd as pqv
n, dim, kc = (int(x) for x in sys.argv[1:4])
rng = np.random.default_rng(n + dim)
data = rng.integers(0, 1 << 24, size=(n, dim), dtype=np.int32).astype(np.float32) * np.float32(1.0 / (1 << 24))
idx = pqv.IndexBuilder(pqv.Corpus.upload(data)).n_clusters(kc).max_iters(3).seed(5).workers(8).build()
off = np.asarray(idx.list_offsets); rows = np.asarray(idx.list_rows)
assert off[0] == 0 and off[-1] == n and np.all(np.diff(off.astype(np.int64)) >= 0)
assert np.array_equal(np.sort(rows), np.arange(n, dtype=rows.dtype))             # every row in exactly one list
asc = np.diff(rows.astype(np.int64)) > 0
asc[off[1:-1][(off[1:-1] > 0) & (off[1:-1] < n)].astype(np.int64) - 1] = True    # (list boundaries may step down)
assert asc.all()                                                                 # ascending row ids inside every list
print("BLOB", hashlib.sha256(idx.to_bytes()).hexdigest())
"""


@pytest.mark.timeout(900)
@pytest.mark.parametrize("n,dim,kc", [(5_000, 64, 37), (200_000, 64, 700), (1_300_000, 32, 4096), (1_100_003, 48, 1)])
def test_device_sorted_lists_equal_the_host_counting_sort(n, dim, kc):
    """index.rs:193-206: list c = the rows assigned to c, ascending.  The device's stable counting sort (list_*_kernel; block sizes
    256 / 1024 / 4096 rows by corpus size) against the host threads' sort (PQV_DEVICE_LISTS=0) on the same build: same blob."""
    import os, subprocess, sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    got = {}
    for mode in ("1", "0"):
        env = dict(os.environ, PQV_DEVICE_LISTS=mode)
        p = subprocess.run([sys.executable, "-c", _LISTS_SCRIPT, str(n), str(dim), str(kc)], cwd=root, env=env, capture_output=True,
                           text=True, timeout=600)
        assert p.returncode == 0, p.stderr[-2000:]
        got[mode] = [l for l in p.stdout.splitlines() if l.startswith("BLOB")][0]
    assert got["1"] == got["0"]


_KPP_SCRIPT = r"""
import sys, os, hashlib
sys.path.insert(0, os.getcwd())
import numpy as np
import pq_vector_amd as pqv
n, dim, kc, style, workers = (int(x) for x in sys.argv[1:6])
rng = np.random.default_rng(n + dim + style)
if style == 0:      # the bench recipe: 24-bit uniform [0, 1)
    data = rng.integers(0, 1 << 24, size=(n, dim), dtype=np.int32).astype(np.float32) * np.float32(1.0 / (1 << 24))
elif style == 1:    # integer-valued features (SIFT-like): integer squared distances, a tie of round-to-nearest-even in almost every add
    data = rng.integers(0, 256, size=(n, dim)).astype(np.float32)
else:               # clustered, with a few rows far outside
    cen = rng.standard_normal((32, dim)).astype(np.float32) * 4
    data = (cen[rng.integers(0, 32, n)] + rng.standard_normal((n, dim)).astype(np.float32) * 0.2).astype(np.float32)
    data[rng.integers(0, n, 5)] *= np.float32(1000.0)
idx = pqv.IndexBuilder(pqv.Corpus.upload(data)).n_clusters(kc).max_iters(2).seed(11).workers(workers).build()
print("BLOB", hashlib.sha256(idx.to_bytes()).hexdigest())
"""


@pytest.mark.timeout(900)
@pytest.mark.parametrize("n,dim,kc,style,workers", [(200_000, 128, 300, 0, 8), (200_000, 128, 300, 1, 8), (120_000, 128, 1000, 2, 256),
                                                    (1_050_000, 128, 1024, 0, 3), (60_000, 192, 64, 1, 1), (3_000, 256, 50, 0, 8)])
def test_device_kmeanspp_rounds_equal_the_host_walk(n, dim, kc, style, workers):
    """index.rs:354-390: the k-means++ rounds enqueued ahead with the pick taken on the device (kernels_kpp.hip) against the host's
    sequential walk (PQV_KPP_DEVICE=0) on the same build -- worker chunkings 1 / 3 / 8 / 256, uniform, integer-valued and clustered data:
    the same blob, i.e. the same 'first slot whose sequential f32 cumulative sum reaches the threshold' in every round."""
    import os, subprocess, sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    got = {}
    for mode in ("1", "0"):
        env = dict(os.environ, PQV_KPP_DEVICE=mode)
        p = subprocess.run([sys.executable, "-c", _KPP_SCRIPT, str(n), str(dim), str(kc), str(style), str(workers)], cwd=root, env=env,
                           capture_output=True, text=True, timeout=600)
        assert p.returncode == 0, p.stderr[-2000:]
        got[mode] = [l for l in p.stdout.splitlines() if l.startswith("BLOB")][0]
    assert got["1"] == got["0"]


_ENV_SCRIPT = r"""
import sys, os, hashlib
sys.path.insert(0, os.getcwd())
import numpy as np
import pq_vector_amd as pqv
rng = np.random.default_rng(99)
n, dim, kc = 120_000, 256, 128
data = rng.random((n, dim), dtype=np.float32)
queries = rng.random((64, dim), dtype=np.float32)
corpus = pqv.Corpus.upload(data)
idx = pqv.IndexBuilder(corpus).n_clusters(kc).max_iters(3).seed(3).workers(8).build()
s = pqv.Searcher(idx, corpus)                                  # made BEFORE anything read the host lists
rows, dist, nf, nc = s.topk(queries, 10, 8)
off = np.asarray(idx.list_offsets); lr = np.asarray(idx.list_rows)                # (the host copy is made here at the latest)
assert off[-1] == n and len(lr) == n and np.array_equal(np.sort(lr), np.arange(n, dtype=lr.dtype))
blob = idx.to_bytes()
s2 = pqv.Searcher(pqv.Index.from_bytes(blob), corpus)          # a loaded index: host lists only, validated row by row
rows2, dist2, nf2, nc2 = s2.topk(queries, 10, 8)
assert np.array_equal(rows, rows2) and np.array_equal(dist.view(np.uint32), dist2.view(np.uint32))
cr = s.candidate_rows(queries[0], 8) if hasattr(s, "candidate_rows") else None   # a host-side call that reads the shared lists
h = hashlib.sha256(blob); h.update(rows.tobytes()); h.update(dist.tobytes()); h.update(nc.tobytes())
if cr is not None: h.update(np.asarray(cr).tobytes())
print("HASH", h.hexdigest())
"""


@pytest.mark.timeout(900)
def test_round6_build_and_creation_switches_never_change_a_result():
    """Round 6 moved work around -- the k-means++ pick and the list sort onto the device, the lists' host copy to its first reader, the
    runtime's one-time set-up to the library's first call, the rows' norms behind the int8 copy, the Lloyd iterations' images out of
    the loop.  Every switch that restores the old place (and all of them together) must give the same blob, the same answers and the
    same candidate rows; a searcher made before the host lists exist and one made from the serialised blob must agree."""
    import os, subprocess, sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    switches = [{}, {"PQV_KPP_DEVICE": "0"}, {"PQV_DEVICE_LISTS": "0"}, {"PQV_KEEP_DEVICE_LISTS": "0"}, {"PQV_LAZY_INIT": "1"},
                {"PQV_EAGER_NORMS": "1"}, {"PQV_LLOYD_KEEP_IMAGES": "0"}, {"PQV_ASSIGN_SHAPE": "256"},
                {"PQV_KPP_DEVICE": "0", "PQV_DEVICE_LISTS": "0", "PQV_KEEP_DEVICE_LISTS": "0", "PQV_LAZY_INIT": "1", "PQV_EAGER_NORMS": "1",
                 "PQV_LLOYD_KEEP_IMAGES": "0", "PQV_ASSIGN_SHAPE": "256"}]
    got = []
    for sw in switches:
        p = subprocess.run([sys.executable, "-c", _ENV_SCRIPT], cwd=root, env=dict(os.environ, **sw), capture_output=True, text=True, timeout=600)
        assert p.returncode == 0, (sw, p.stderr[-2000:])
        got.append([l for l in p.stdout.splitlines() if l.startswith("HASH")][0])
    assert len(set(got)) == 1, list(zip(switches, got))
