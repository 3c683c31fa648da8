"""N > 1 path on CPU: world_size-2 gloo processes run the shard split, the all-gather and
the deterministic merge of pq_vector_amd/sharding.py.  The per-shard search itself needs a
GPU, so here each rank gets its shard's top-k from the CPU oracle (test infrastructure);
what is under test is the exchange: it must equal a single-process merge of the same
per-shard lists, and the global brute-force top-k when every cluster is probed."""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
N, DIM, KC, K, NPROBE, NQ = 3001, 24, 5, 10, 5, 7


def _inputs():
    rng = np.random.default_rng(123)
    return rng.random((N, DIM), dtype=np.float32), rng.random((NQ, DIM), dtype=np.float32)


def _shard_topk(oracle, data, queries, lo, hi):
    shard = np.ascontiguousarray(data[lo:hi])
    idx = oracle.build_index(shard, n_clusters=KC, workers=2)
    rows, distv, nf, _ = idx.topk_batch(shard, queries, K, NPROBE)
    rows = rows.astype(np.int64)
    for q in range(len(queries)):
        rows[q, nf[q]:] = -1
        distv[q, nf[q]:] = np.inf
    return distv, rows, nf


def _worker(rank, world, port, out_path):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from oracle_binding import Oracle
    from pq_vector_amd.sharding import ShardExchange, shard_range
    data, queries = _inputs()
    lo, hi = shard_range(rank, world, N)
    d, r, _ = _shard_topk(Oracle(), data, queries, lo, hi)
    x = ShardExchange(world, NQ, K, torch.device("cpu"))
    md, mr = x.exchange(torch.from_numpy(d), torch.from_numpy(r), lo)
    # every rank must hold the identical answer
    gathered = [torch.empty_like(mr) for _ in range(world)]
    dist.all_gather(gathered, mr)
    assert all(torch.equal(g, mr) for g in gathered)
    if rank == 0:
        np.savez(out_path, dist=md.numpy(), rows=mr.numpy())
    dist.barrier()
    dist.destroy_process_group()


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


@pytest.mark.timeout(300)
def test_two_rank_exchange_matches_single_process_merge(oracle, tmp_path):
    import pq_vector_amd as pqv
    from pq_vector_amd.sharding import shard_range
    world = 2
    out = str(tmp_path / "merged.npz")
    mp.spawn(_worker, args=(world, _free_port(), out), nprocs=world, join=True)
    got = np.load(out)

    data, queries = _inputs()
    ranges = [shard_range(r, world, N) for r in range(world)]
    assert ranges[0][0] == 0 and ranges[-1][1] == N and ranges[0][1] == ranges[1][0]
    per = [_shard_topk(oracle, data, queries, lo, hi) for lo, hi in ranges]
    dist_l = np.stack([p[0] for p in per])
    rows_l = np.stack([np.where(p[1] >= 0, p[1] + lo, 0xFFFFFFFF).astype(np.uint32)
                       for p, (lo, _) in zip(per, ranges)])
    cnt_l = np.stack([p[2] for p in per]).astype(np.uint32)
    md, mr, ml, mc = pqv.merge_topk(dist_l, rows_l, cnt_l)          # product's host merge
    assert (got["dist"].view(np.uint32) == md.view(np.uint32)).all()
    assert (got["rows"] == mr.astype(np.int64)).all()

    # nprobe == n_clusters => the merged answer is the global exact top-k
    for q in range(NQ):
        d2 = np.array([oracle.l2_ref4(queries[q], data[r]) for r in range(N)], np.float32)
        order = np.lexsort((np.arange(N), d2.view(np.uint32)))[:K]
        assert (got["rows"][q] == order).all()
        assert (got["dist"][q] == np.sqrt(d2[order])).all()


def test_shard_ranges_tile_exactly():
    from pq_vector_amd.sharding import shard_range
    for n in (0, 1, 7, 1000, 1_000_000, 100_000_000):
        for world in (1, 2, 3, 4, 8):
            edges = [shard_range(r, world, n) for r in range(world)]
            assert edges[0][0] == 0 and edges[-1][1] == n
            assert all(edges[i][1] == edges[i + 1][0] for i in range(world - 1))
            sizes = [hi - lo for lo, hi in edges]
            assert max(sizes) - min(sizes) <= 1


# ---- ONE Parquet file, one row-group range per rank (BASELINE config 4's partition unit) ----------------------------------
def _write_shared_file(path):
    import pyarrow as pa
    import pyarrow.parquet as pq
    data, _ = _inputs()
    col = pa.ListArray.from_arrays(pa.array(np.arange(0, (N + 1) * DIM, DIM, dtype=np.int32)), pa.array(data.reshape(-1)))
    pq.write_table(pa.table({"id": pa.array(np.arange(N, dtype=np.int32)), "emb": col}), path, row_group_size=431,
                   compression="NONE", use_dictionary=False, data_page_size=8 * 1024)


class _HostRows:
    """Stand-in for the device corpus on a CPU-only box: collects what the page walker would upload."""

    def __init__(self, n, dim):
        self.a = np.full((n, dim), np.nan, np.float32)

    def write_rows_ptr(self, row, addr, m, f64=False):
        import ctypes
        dim = self.a.shape[1]
        self.a[row:row + m] = np.frombuffer(ctypes.string_at(addr, m * dim * 4), dtype=np.float32).reshape(m, dim)


def _file_worker(rank, world, port, path, out_path):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from oracle_binding import Oracle
    from pq_vector_amd import parquet_io
    from pq_vector_amd.sharding import ShardExchange, shard_row_groups
    _, queries = _inputs()
    from pq_vector_amd.sharding import check_shard_dims
    lo, hi, base, rows = shard_row_groups(rank, world, path)          # from the footer alone, the same cuts on every rank
    if rows:
        shard = _HostRows(rows, DIM)
        assert parquet_io._load_pages(path, "emb", shard, DIM, None, 2, [0, 0], row_groups=(lo, hi)) is True
        idx = Oracle().build_index(shard.a, n_clusters=min(KC, rows), workers=2)      # the shard's own index, shard-local row ids
        r, d, nf, _ = idx.topk_batch(shard.a, queries, K, KC)          # every list probed
        r = r.astype(np.int64)
        for q in range(NQ):
            r[q, nf[q]:] = -1
            d[q, nf[q]:] = np.inf
    else:       # fewer row groups than ranks: a surplus rank holds an empty range and answers with empty lists
        r, d = np.full((NQ, K), -1, np.int64), np.full((NQ, K), np.inf, np.float32)
    assert check_shard_dims(DIM if rows else 0) == DIM                 # every shard found the same list length
    if world > 2:                                                      # ... and a shard that found another one stops every rank
        from pq_vector_amd import PqvError
        try:
            check_shard_dims(DIM + 1 if rank == 1 else (DIM if rows else 0))
            raise AssertionError("inconsistent shard dimensions went unnoticed")
        except PqvError as e:
            assert "Embedding vectors have inconsistent dimensions" in str(e)
    x = ShardExchange(world, NQ, K, torch.device("cpu"))
    md, mr = x.exchange(torch.from_numpy(d), torch.from_numpy(r), base)
    meta = torch.tensor([lo, hi, base, rows], dtype=torch.int64)
    allmeta = [torch.empty_like(meta) for _ in range(world)]
    dist.all_gather(allmeta, meta)
    if rank == 0:
        np.savez(out_path, dist=md.numpy(), rows=mr.numpy(), meta=torch.stack(allmeta).numpy())
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.timeout(300)
def test_two_ranks_share_one_parquet_file_by_row_group_ranges(oracle, tmp_path):
    """Two shards built from ONE file: the ranges are cut at row-group boundaries, tile the file, and the merged answer carries
    the same FILE-GLOBAL row ids as a single-shard search with every list probed (= the exact top-k of the whole column)."""
    import pyarrow.parquet as pq
    path, out = str(tmp_path / "shared.parquet"), str(tmp_path / "merged_file.npz")
    _write_shared_file(path)
    n_rg = pq.ParquetFile(path).metadata.num_row_groups
    assert n_rg == 7
    world = 2
    mp.spawn(_file_worker, args=(world, _free_port(), path, out), nprocs=world, join=True)
    got = np.load(out)
    meta = got["meta"]
    assert meta[0][0] == 0 and meta[-1][1] == n_rg and meta[0][1] == meta[1][0]          # row-group ranges tile the file
    assert meta[0][2] == 0 and meta[1][2] == meta[0][3] and meta[:, 3].sum() == N          # row bases = prefix sums
    assert meta[0][3] % 431 == 0                                                           # cut AT a row-group boundary
    data, queries = _inputs()
    whole = oracle.build_index(data, n_clusters=KC, workers=2)
    r1, d1, nf1, _ = whole.topk_batch(data, queries, K, KC)
    assert (nf1 == K).all()
    assert (got["rows"] == r1.astype(np.int64)).all()
    assert (got["dist"].view(np.uint32) == d1.view(np.uint32)).all()


@pytest.mark.timeout(600)
def test_eight_ranks_share_one_parquet_file_with_fewer_row_groups_than_ranks(oracle, tmp_path):
    """The --gpus 8 job shape on CPU (world-8 gloo): EIGHT cuts from one footer of SEVEN row groups -- one rank holds an empty
    range and answers with empty lists --, eight shard indexes, the 8-way deterministic merge; with every list probed the merged
    answer is the exact top-k of the whole column with file-global row ids.  Also: check_shard_dims agrees across the ranks and
    raises the reference's message everywhere when one shard found another list length (src/ivf/parquet.rs:231-280)."""
    import pyarrow.parquet as pq
    path, out = str(tmp_path / "shared8.parquet"), str(tmp_path / "merged_file8.npz")
    _write_shared_file(path)
    n_rg = pq.ParquetFile(path).metadata.num_row_groups
    assert n_rg == 7
    world = 8
    mp.spawn(_file_worker, args=(world, _free_port(), path, out), nprocs=world, join=True)
    got = np.load(out)
    meta = got["meta"]
    assert meta.shape == (8, 4) and meta[0][0] == 0 and meta[-1][1] == n_rg
    assert all(meta[i][1] == meta[i + 1][0] for i in range(7))                             # the ranges tile the file in rank order
    assert (meta[:, 3] == 0).sum() == 1 and meta[:, 3].sum() == N                          # exactly one empty range
    assert all(meta[i][2] == meta[:i, 3].sum() for i in range(8))                          # row bases = prefix sums
    data, queries = _inputs()
    whole = oracle.build_index(data, n_clusters=KC, workers=2)
    r1, d1, nf1, _ = whole.topk_batch(data, queries, K, KC)
    assert (got["rows"] == r1.astype(np.int64)).all()
    assert (got["dist"].view(np.uint32) == d1.view(np.uint32)).all()
