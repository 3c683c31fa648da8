"""CPU tests of the Parquet side: the PQ_VECTOR1 embed/read format, the
Thrift-compact footer splice and the column validation.  Host-only: an Index assembled
from parts needs no GPU."""
import os
import struct

import numpy as np
import pyarrow as pa
import pyarrow.parquet as pq
import pytest


def _table(n=50, dim=4, seed=0, value_type=pa.float32(), extra_meta=None):
    rng = np.random.default_rng(seed)
    vecs = rng.random((n, dim)).astype(np.float64 if value_type == pa.float64() else np.float32)
    ids = pa.array(np.arange(n, dtype=np.int32))
    col = pa.array(vecs.tolist(), type=pa.list_(pa.field("item", value_type)))
    t = pa.table({"id": ids, "vec": col, "title": pa.array([f"t{i}" for i in range(n)])})
    if extra_meta:
        t = t.replace_schema_metadata(extra_meta)
    return t, vecs


def _index(pqv, dim=4, k=3, n=50):
    lists = [np.arange(i, n, k, dtype=np.uint32) for i in range(k)]
    cent = np.arange(k * dim, dtype=np.float32).reshape(k, dim)
    return pqv.Index.from_parts(dim, cent, lists)


@pytest.fixture(scope="module")
def pqv():
    import pq_vector_amd
    from pq_vector_amd import _ffi
    _ffi.lib()
    return pq_vector_amd


@pytest.mark.parametrize("row_group_size,n_meta", [(None, 0), (16, 3), (7, 20)])
def test_inplace_append_round_trip(pqv, tmp_path, row_group_size, n_meta):
    """src/ivf/parquet.rs:623-660: the file grows, stays a valid Parquet file with identical
    data, carries both KV keys, and the blob reads back identical."""
    from pq_vector_amd import parquet_io as pio
    meta = {f"user_key_{i}": "v" * (i + 1) for i in range(n_meta)} or None   # >= 15 entries: long list header
    t, vecs = _table(extra_meta=meta)
    path = str(tmp_path / "data.parquet")
    pq.write_table(t, path, row_group_size=row_group_size)
    before = os.path.getsize(path)
    assert not pqv.has_pq_vector_index(path)
    idx = _index(pqv)
    off = pio.append_index_inplace(path, idx, "vec")
    assert off == before - 8                                            # :565-566
    assert os.path.getsize(path) > before
    assert pqv.has_pq_vector_index(path)

    back = pq.read_table(path)                                          # standard reader ignores the blob
    assert back.column("id").to_pylist() == list(range(50))
    assert np.allclose(np.array(back.column("vec").to_pylist(), np.float32), vecs)
    assert back.column("title").to_pylist() == [f"t{i}" for i in range(50)]
    md = pq.read_metadata(path).metadata
    assert md[b"pq_vector_index_offset"] == str(off).encode()
    assert md[b"pq_vector_embedding_column"] == b"vec"
    for i in range(n_meta):
        assert md[f"user_key_{i}".encode()] == b"v" * (i + 1)          # other KV entries survive
    assert pq.read_metadata(path).num_row_groups == pq.ParquetFile(path).metadata.num_row_groups

    got, col = pqv.read_index_from_parquet(path)
    assert col == "vec" and got.dim == 4                                # :658-659
    assert got.to_bytes() == idx.to_bytes()
    with open(path, "rb") as f:
        f.seek(off)
        assert f.read(10) == b"PQ_VECTOR1"
        assert struct.unpack("<Q", f.read(8))[0] == len(idx.to_bytes())


def test_rebuild_replaces_stale_keys(pqv, tmp_path):
    """src/ivf/parquet.rs:573-575: a second in-place build replaces the KV entries (the old
    blob stays behind as dead bytes)."""
    from pq_vector_amd import parquet_io as pio
    t, _ = _table()
    path = str(tmp_path / "d.parquet")
    pq.write_table(t, path)
    off1 = pio.append_index_inplace(path, _index(pqv, k=3), "vec")
    off2 = pio.append_index_inplace(path, _index(pqv, k=5), "vec")
    assert off2 > off1
    md = pq.read_metadata(path).metadata
    assert md[b"pq_vector_index_offset"] == str(off2).encode()
    kv = pio._footer_kv(pio._read_footer(path)[1])
    assert sum(1 for k, _ in kv if k == b"pq_vector_index_offset") == 1
    got, _ = pqv.read_index_from_parquet(path)
    assert got.n_clusters == 5
    assert pq.read_table(path).num_rows == 50


def test_build_new_copy(pqv, tmp_path):
    from pq_vector_amd import parquet_io as pio
    t, vecs = _table(n=40, dim=6)
    src, out = str(tmp_path / "s.parquet"), str(tmp_path / "o.parquet")
    pq.write_table(t, src, row_group_size=16)
    idx = _index(pqv, dim=6, n=40)
    pio.write_parquet_with_index(src, out, idx, "vec")
    assert not pqv.has_pq_vector_index(src) and pqv.has_pq_vector_index(out)
    back = pq.read_table(out)
    assert np.allclose(np.array(back.column("vec").to_pylist(), np.float32), vecs)
    got, col = pqv.read_index_from_parquet(out)
    assert got.to_bytes() == idx.to_bytes() and col == "vec"
    # one vector per data page: the embedding column chunk has ~rows pages worth of bytes
    rg = pq.ParquetFile(out).metadata.row_group(0)
    names = [rg.column(i).path_in_schema for i in range(rg.num_columns)]
    assert any(n.startswith("vec") for n in names)


def test_payload_errors(pqv, tmp_path):
    from pq_vector_amd import parquet_io as pio
    t, _ = _table()
    path = str(tmp_path / "d.parquet")
    pq.write_table(t, path)
    with pytest.raises(pqv.PqvError, match="Missing pq-vector index metadata in parquet footer"):
        pqv.read_index_from_parquet(path)
    with pytest.raises(pqv.PqvError, match="pq-vector index payload is truncated"):
        pio.read_index_payload(b"PQ_VEC")
    with pytest.raises(pqv.PqvError, match="Invalid pq-vector index magic"):
        pio.read_index_payload(b"XX_VECTOR1" + b"\0" * 8)
    with pytest.raises(pqv.PqvError, match="pq-vector index bytes are truncated"):
        pio.read_index_payload(b"PQ_VECTOR1" + struct.pack("<Q", 100) + b"\0" * 10)
    tiny = str(tmp_path / "tiny")
    open(tiny, "wb").write(b"PAR1")
    with pytest.raises(pqv.PqvError, match="Parquet file too small to contain a footer"):
        pio.append_index_inplace(tiny, _index(pqv), "vec")
    enc = str(tmp_path / "enc")
    open(enc, "wb").write(b"PAR1" + b"\0" * 20 + struct.pack("<I", 4) + b"PARE")
    with pytest.raises(pqv.PqvError, match="Encrypted parquet footers are not supported"):
        pio.append_index_inplace(enc, _index(pqv), "vec")


def test_column_validation_texts(pqv, tmp_path):
    """src/ivf/parquet.rs:231-296 messages, checked on the host-side chunk reader."""
    from pq_vector_amd import parquet_io as pio
    t, vecs = _table(n=30, dim=5)
    p = str(tmp_path / "ok.parquet")
    pq.write_table(t, p, row_group_size=8)
    got = np.concatenate(list(pio._column_chunks(p, "vec", batch_rows=7)))
    assert got.dtype == np.float32 and np.array_equal(got, vecs)
    with pytest.raises(pqv.PqvError, match="Column 'nope' not found"):
        list(pio._column_chunks(p, "nope"))
    with pytest.raises(pqv.PqvError, match="Embedding column is not a list array"):
        list(pio._column_chunks(p, "id"))

    def write(col, name):
        path = str(tmp_path / name)
        pq.write_table(pa.table({"vec": col}), path)
        return path
    f32l = pa.list_(pa.float32())
    with pytest.raises(pqv.PqvError, match="Embedding column contains null rows"):
        list(pio._column_chunks(write(pa.array([[1.0, 2.0], None], type=f32l), "n1"), "vec"))
    with pytest.raises(pqv.PqvError, match="Embedding values contain nulls"):
        list(pio._column_chunks(write(pa.array([[1.0, None]], type=f32l), "n2"), "vec"))
    with pytest.raises(pqv.PqvError, match="Embedding row has zero length"):
        list(pio._column_chunks(write(pa.array([[1.0], []], type=f32l), "n3"), "vec"))
    with pytest.raises(pqv.PqvError, match="Embedding vectors have inconsistent dimensions"):
        list(pio._column_chunks(write(pa.array([[1.0, 2.0], [1.0]], type=f32l), "n4"), "vec"))
    with pytest.raises(pqv.PqvError, match="Embedding values are not float32/float64"):
        list(pio._column_chunks(write(pa.array([[1, 2]], type=pa.list_(pa.int32())), "n5"), "vec"))
    with pytest.raises(pqv.PqvError, match="Embedding column has no rows"):
        list(pio._column_chunks(write(pa.array([], type=f32l), "n6"), "vec"))
    # Float64 and FixedSizeList columns are accepted
    t64, v64 = _table(n=9, dim=3, value_type=pa.float64())
    p64 = str(tmp_path / "f64.parquet")
    pq.write_table(t64, p64)
    got64 = np.concatenate(list(pio._column_chunks(p64, "vec")))
    assert got64.dtype == np.float64 and np.array_equal(got64, v64)
    fsl = pa.FixedSizeListArray.from_arrays(pa.array(np.arange(12, dtype=np.float32)), 4)
    gotf = np.concatenate(list(pio._column_chunks(write(fsl, "fsl"), "vec")))
    assert gotf.shape == (3, 4)


def test_thrift_splice_is_byte_stable(pqv, tmp_path):
    """Re-emitting the untouched footer reproduces it byte for byte (every field is copied
    raw; only field headers are re-encoded)."""
    from pq_vector_amd import parquet_io as pio
    t, _ = _table(extra_meta={"a": "b"})
    path = str(tmp_path / "d.parquet")
    pq.write_table(t, path, row_group_size=10)
    _, meta = pio._read_footer(path)
    assert pio._emit_struct(pio._struct_fields(meta)) == meta


def _data_pages(path, column_root):
    """(data pages, rows) of the column's chunk in row group 0, by walking the Thrift PageHeaders of the chunk."""
    from pq_vector_amd import parquet_io as pio
    md = pq.ParquetFile(path).metadata
    rg = md.row_group(0)
    col = next(rg.column(i) for i in range(rg.num_columns) if rg.column(i).path_in_schema.split(".")[0] == column_root)
    start = col.dictionary_page_offset if col.has_dictionary_page else col.data_page_offset
    buf = open(path, "rb").read()[start:start + col.total_compressed_size]
    pos, pages = 0, 0
    while pos < len(buf):
        end = pio._skip(buf, pos, pio.T_STRUCT)                 # the PageHeader struct
        fields = dict((fid, raw) for fid, _, raw in pio._struct_fields(buf[pos:end]))
        ptype = pio._zigzag_decode(pio._varint(fields[1], 0)[0])
        comp = pio._zigzag_decode(pio._varint(fields[3], 0)[0])
        pages += ptype in (0, 3)                                # DATA_PAGE, DATA_PAGE_V2
        pos = end + comp
    return pages, rg.num_rows


@pytest.mark.parametrize("codec", ["NONE", "SNAPPY"])
def test_build_new_writes_one_vector_per_page_and_keeps_column_codecs(pqv, tmp_path, codec):
    """set_data_page_row_count_limit(1) (src/ivf/parquet.rs:324-326): the embedding column of a build_new copy must
    hold ONE vector per data page, or the reference's single-row page-index reads degrade to whole-chunk reads; the
    source's per-column compression is kept (parquet.rs:328-331)."""
    from pq_vector_amd import parquet_io as pio
    t, _ = _table(n=300, dim=16)
    src, out = str(tmp_path / "s.parquet"), str(tmp_path / "o.parquet")
    pq.write_table(t, src, row_group_size=300, compression={"id": codec, "vec": "NONE", "title": codec})
    pio.write_parquet_with_index(src, out, _index(pqv, dim=16, n=300), "vec")
    pages, rows = _data_pages(out, "vec")
    assert rows == 300 and pages == 300
    rg = pq.ParquetFile(out).metadata.row_group(0)
    got = {rg.column(i).path_in_schema.split(".")[0]: rg.column(i).compression for i in range(rg.num_columns)}
    want = "UNCOMPRESSED" if codec == "NONE" else codec
    assert got == {"id": want, "vec": "UNCOMPRESSED", "title": want}


@pytest.mark.parametrize("codec", ["SNAPPY", "ZSTD"])
def test_build_new_keeps_the_codec_of_the_nested_embedding_leaf(pqv, tmp_path, codec):
    """parquet.rs:322-336 keeps the codec PER LEAF.  parquet-cpp resolves column properties by the full dotted leaf path
    (vec.list.element), so a dict keyed by the top-level name leaves the embedding column -- a nested leaf -- uncompressed
    in the copy; with a compressed source leaf the copy's leaf must carry the same codec (and no dictionary, :352)."""
    from pq_vector_amd import parquet_io as pio
    if not pa.Codec.is_available(codec.lower()):
        pytest.skip(f"{codec} not built into this pyarrow")
    t, _ = _table(n=200, dim=16)
    src, out = str(tmp_path / "s.parquet"), str(tmp_path / "o.parquet")
    pq.write_table(t, src, row_group_size=200, compression={"id": "NONE", "vec.list.element": codec, "title": codec})
    rg_s = pq.ParquetFile(src).metadata.row_group(0)
    src_codecs = {rg_s.column(i).path_in_schema: rg_s.column(i).compression for i in range(rg_s.num_columns)}
    assert src_codecs["vec.list.element"] == codec              # the source really has a compressed nested leaf
    pio.write_parquet_with_index(src, out, _index(pqv, dim=16, n=200), "vec")
    rg = pq.ParquetFile(out).metadata.row_group(0)
    got = {rg.column(i).path_in_schema: rg.column(i).compression for i in range(rg.num_columns)}
    assert got == src_codecs
    vec = [rg.column(i) for i in range(rg.num_columns) if rg.column(i).path_in_schema.startswith("vec.")][0]
    assert not any("DICTIONARY" in str(e) for e in vec.encodings)
    assert pq.read_table(out).equals(pq.read_table(src))


def test_footer_offset_is_parsed_like_rust_u64_and_corrupt_footers_raise_pqv_errors(pqv, tmp_path):
    from pq_vector_amd import parquet_io as pio
    t, _ = _table()
    for bad in (" 5", "5_0", "-5", "0x10", "", "18446744073709551616", "99999999999"):
        path = str(tmp_path / "b.parquet")
        pq.write_table(t.replace_schema_metadata({pio.OFFSET_KEY: bad, pio.COLUMN_KEY: "vec"}), path)
        with pytest.raises(pqv.PqvError):
            pqv.read_index_from_parquet(path)
    path = str(tmp_path / "ok.parquet")
    pq.write_table(t, path)
    pio.append_index_inplace(path, _index(pqv), "vec")
    raw = bytearray(open(path, "rb").read())
    flen = struct.unpack("<I", raw[-8:-4])[0]
    cut = str(tmp_path / "cut.parquet")
    # a footer whose length field points into garbage: the Thrift walk must fail as a PqvError, not an IndexError
    open(cut, "wb").write(bytes(raw[:len(raw) - 8 - flen]) + b"\x19\xfc\xff\xff" + struct.pack("<I", 4) + b"PAR1")
    with pytest.raises(pqv.PqvError):
        pqv.read_index_from_parquet(cut)


# ---------------------------------------------------------------------------------------
# N1 fast path: the page walker and its two host helpers (no GPU: a stand-in corpus collects what would be uploaded)
# ---------------------------------------------------------------------------------------
class _CollectingCorpus:
    def __init__(self, n, dim):
        self.a = np.full((n, dim), np.nan, np.float32)

    def write_rows_ptr(self, row, addr, m, f64=False):
        import ctypes
        dim = self.a.shape[1]
        src = np.frombuffer(ctypes.string_at(addr, m * dim * (8 if f64 else 4)), dtype=np.float64 if f64 else np.float32)
        self.a[row:row + m] = src.reshape(m, dim).astype(np.float32)


def _rg_offsets(path):
    meta = pq.ParquetFile(path).metadata
    off = np.zeros(meta.num_row_groups + 1, dtype=np.int64)
    for i in range(meta.num_row_groups):
        off[i + 1] = off[i] + meta.row_group(i).num_rows
    return off


@pytest.mark.parametrize("value_type,codec,dictionary", [("f32", "NONE", False), ("f32", "SNAPPY", True), ("f64", "ZSTD", False),
                                                         ("f32", "NONE", True), ("f64", "GZIP", True)])
def test_page_walker_uploads_exactly_the_column(pqv, tmp_path, value_type, codec, dictionary):
    """src/ivf/parquet.rs:216-305 through the data pages themselves: PLAIN and dictionary-encoded pages, compressed or not,
    f32 and f64, several row groups and many pages per chunk -- what reaches the corpus is the column, row for row."""
    from pq_vector_amd import parquet_io
    rng = np.random.default_rng(5)
    n, dim = 30000, 24
    dt = np.float64 if value_type == "f64" else np.float32
    vec = rng.integers(0, 40, (n, dim)).astype(dt) if dictionary else rng.standard_normal((n, dim)).astype(dt)
    col = pa.ListArray.from_arrays(pa.array(np.arange(0, (n + 1) * dim, dim, dtype=np.int32)), pa.array(vec.reshape(-1)))
    path = str(tmp_path / "c.parquet")
    pq.write_table(pa.table({"id": pa.array(np.arange(n, dtype=np.int32)), "emb": col}), path, row_group_size=11000,
                   compression=codec, use_dictionary=dictionary, data_page_size=64 * 1024)
    c, counters = _CollectingCorpus(n, dim), [0, 0]
    assert parquet_io._load_pages(path, "emb", c, dim, _rg_offsets(path), 4, counters) is True
    assert counters[1] > 3 and counters[0] == n * dim * (8 if value_type == "f64" else 4)
    assert np.array_equal(c.a.view(np.uint32), vec.astype(np.float32).view(np.uint32))


@pytest.mark.parametrize("codec,dictionary", [("NONE", False), ("SNAPPY", True)])
def test_page_walker_loads_a_row_group_range_as_a_shard(pqv, tmp_path, codec, dictionary):
    """One file shared by several GPUs (BASELINE config 4: 8 Parquet row-group ranges): shard_row_groups cuts at row-group
    boundaries, row_base = the prefix sum of the row groups before the range (src/df_vector/access.rs:128-144), and the walker
    uploads exactly the range's rows, numbered from the range's first row.  The ranges tile the file."""
    from pq_vector_amd import parquet_io
    from pq_vector_amd.sharding import shard_row_groups
    rng = np.random.default_rng(15)
    n, dim = 25000, 16
    vec = rng.integers(0, 40, (n, dim)).astype(np.float32) if dictionary else rng.standard_normal((n, dim)).astype(np.float32)
    col = pa.ListArray.from_arrays(pa.array(np.arange(0, (n + 1) * dim, dim, dtype=np.int32)), pa.array(vec.reshape(-1)))
    path = str(tmp_path / "s.parquet")
    pq.write_table(pa.table({"id": pa.array(np.arange(n, dtype=np.int32)), "emb": col}), path, row_group_size=3000,
                   compression=codec, use_dictionary=dictionary, data_page_size=16 * 1024)
    meta = pq.ParquetFile(path).metadata
    assert meta.num_row_groups == 9
    for world in (1, 2, 3, 8, 12):
        covered = 0
        for rank in range(world):
            lo, hi, base, rows = shard_row_groups(rank, world, meta)
            assert base == covered and rows == sum(meta.row_group(i).num_rows for i in range(lo, hi))
            assert shard_row_groups(rank, world, path) == (lo, hi, base, rows)
            covered += rows
            if rows == 0:
                assert world > meta.num_row_groups
                assert parquet_io._plan_pages(path, "emb", 2, None, (lo, hi)) is None
                continue
            c, counters = _CollectingCorpus(rows, dim), [0, 0]
            assert parquet_io._load_pages(path, "emb", c, dim, None, 3, counters, row_groups=(lo, hi)) is True
            assert counters[0] == rows * dim * 4
            assert np.array_equal(c.a.view(np.uint32), vec[base:base + rows].view(np.uint32))
            # the Arrow fallback reads the same range
            got = np.concatenate(list(parquet_io._column_chunks(path, "emb", row_groups=list(range(lo, hi)))))
            assert np.array_equal(got, vec[base:base + rows])
        assert covered == n
        if world <= meta.num_row_groups:          # balanced to within one row group
            sizes = [shard_row_groups(r, world, meta)[3] for r in range(world)]
            assert max(sizes) - min(sizes) <= 3000
    with pytest.raises(pqv.PqvError, match="row-group range"):
        parquet_io._plan_pages(path, "emb", 2, None, (3, 99))


@pytest.mark.parametrize("lists", [[[1.0, 2.0], [3.0], [4.0, 5.0]], [[1.0, None], [3.0, 4.0]], [[1.0, 2.0], None, [3.0, 4.0]],
                                   [[1.0, 2.0], [], [3.0, 4.0]]])
def test_page_walker_refuses_what_the_reference_rejects(pqv, tmp_path, lists):
    """Ragged lists, null values, null rows, empty lists (parquet.rs:231-280): the level runs give them away, the walker
    returns False and load_embedding_column's Arrow path raises the reference's message."""
    from pq_vector_amd import parquet_io
    path = str(tmp_path / "bad.parquet")
    pq.write_table(pa.table({"emb": pa.array(lists, type=pa.list_(pa.float32()))}), path)
    assert parquet_io._load_pages(path, "emb", _CollectingCorpus(len(lists), 2), 2, _rg_offsets(path), 2, [0, 0]) is False
    assert parquet_io._plan_pages(path, "emb", 2) is None               # ... and the plan says so before any device is touched


def test_page_plan_discovers_the_list_length_without_a_device(pqv, tmp_path):
    from pq_vector_amd import parquet_io
    rng = np.random.default_rng(6)
    for dim, n in ((1, 700), (7, 9000), (768, 300)):
        vec = rng.standard_normal((n, dim)).astype(np.float32)
        col = pa.ListArray.from_arrays(pa.array(np.arange(0, (n + 1) * dim, dim, dtype=np.int32)), pa.array(vec.reshape(-1)))
        path = str(tmp_path / f"d{dim}.parquet")
        pq.write_table(pa.table({"emb": col}), path, compression="NONE", use_dictionary=False, data_page_size=32 * 1024)
        plan = parquet_io._plan_pages(path, "emb", 2)
        assert plan is not None and plan.dim == dim and plan.n_rows == n and not plan.f64
        assert sum(t[4] for t in plan.tasks) == n * dim and all(t[8] is not None for t in plan.tasks)


def test_level_run_and_dictionary_helpers(pqv):
    import ctypes as C
    from pq_vector_amd import _ffi
    L = _ffi.lib()

    def check(buf, bw, n, mode, expect):
        b = (C.c_uint8 * len(buf)).from_buffer_copy(bytes(buf))
        return L.pqv_parquet_levels_check(b, len(buf), bw, n, mode, expect, None)
    # definition levels: one RLE run of 1000 x 2
    assert check([2000 & 0x7F | 0x80, 2000 >> 7, 2], 2, 1000, 0, 2) == 0
    assert check([2000 & 0x7F | 0x80, 2000 >> 7, 1], 2, 1000, 0, 2) == 1
    # repetition levels of lists of 4: bit-packed groups 0111 0111 (lsb first: 0xEE)
    assert check([(2 << 1) | 1, 0xEE, 0xEE], 1, 16, 1, 4) == 0
    assert check([(2 << 1) | 1, 0xEE, 0xEE], 1, 16, 1, 8) == 1
    assert check([(2 << 1) | 1, 0xEE, 0xEF], 1, 16, 1, 4) == 1          # a list of length 5 ... 3
    assert check([(2 << 1) | 1, 0xEE], 1, 16, 1, 4) < 0                  # truncated
    # lists of 12: a bit-packed group holding the 0, then an RLE run of ones that must not cross the next row start
    assert check([(1 << 1) | 1, 0xFE, (4 << 1), 1], 1, 12, 1, 12) == 0
    assert check([(1 << 1) | 1, 0xFE, (5 << 1), 1], 1, 13, 1, 12) == 1
    # discovery (expect 0 + an out pointer): the second level 0 gives the list length, then the run is checked against it
    period = C.c_uint64(0)

    def discover(buf, n):
        b = (C.c_uint8 * len(buf)).from_buffer_copy(bytes(buf))
        period.value = 0
        return L.pqv_parquet_levels_check(b, len(buf), 1, n, 1, 0, C.byref(period)), period.value
    assert discover([(2 << 1) | 1, 0xEE, 0xEE], 16) == (0, 4)
    assert discover([(1 << 1) | 1, 0xFE, (4 << 1), 1], 12) == (0, 12)    # one row in the page: the whole page is the list
    assert discover([(8 << 1), 0], 8) == (0, 1)                          # RLE run of zeros: lists of one value
    assert discover([(2 << 1) | 1, 0xEE, 0xEF], 16)[0] == 1
    assert discover([(2 << 1) | 1, 0xEF, 0xEE], 16)[0] == 1              # the page starts inside a row
    assert check([(2 << 1) | 1, 0xEE, 0xEE], 1, 16, 1, 0) < 0            # expect 0 without the out pointer
    # dictionary indices, bit width 2: 0 1 2 3 | RLE 3 x index 1
    dict32 = np.array([10.0, 11.0, 12.0, 13.0], np.float32)
    out = np.zeros(11, np.float32)
    enc = bytes([2, (1 << 1) | 1, 0b11100100, 0b11100100, (3 << 1), 1])
    b = (C.c_uint8 * len(enc)).from_buffer_copy(enc)
    assert L.pqv_parquet_dict_decode(b, len(enc), dict32.ctypes.data_as(_ffi.vp), 4, 4, 11, out.ctypes.data_as(_ffi.vp)) == 0
    assert out.tolist() == [10, 11, 12, 13, 10, 11, 12, 13, 11, 11, 11]
    enc_bad = bytes([3, (1 << 1) | 1, 0xFF, 0xFF, 0xFF])                  # index 7 of a 4-entry dictionary
    b = (C.c_uint8 * len(enc_bad)).from_buffer_copy(enc_bad)
    assert L.pqv_parquet_dict_decode(b, len(enc_bad), dict32.ctypes.data_as(_ffi.vp), 4, 4, 8, out.ctypes.data_as(_ffi.vp)) < 0
    # a crafted bit-packed header whose byte count wraps to 0 in 64 bits ((1 << 59) groups x 32 bits): refused, nothing past the
    # declared length is read (the guard page of a mapped file would be next)
    def varint(h):
        o = []
        while True:
            o.append((h & 0x7F) | (0x80 if h >> 7 else 0))
            h >>= 7
            if not h:
                return o
    crafted = bytes([32] + varint(((1 << 59) << 1) | 1) + [0])
    assert len(crafted) == 11
    big = np.full(64, -1.0, np.float32)
    b = (C.c_uint8 * len(crafted)).from_buffer_copy(crafted)
    assert L.pqv_parquet_dict_decode(b, len(crafted), dict32.ctypes.data_as(_ffi.vp), 4, 4, 16, big.ctypes.data_as(_ffi.vp)) < 0
    assert (big == -1.0).all()
    assert check(varint(((1 << 61) << 1) | 1) + [0xEE], 8, 16, 0, 2) < 0     # the same through the level checker (bit width 8)
    assert check(varint(((1 << 60) << 1) | 1) + [0xEE], 1, 16, 0, 2) < 0     # a count no buffer can hold


def test_native_page_header_walk_agrees_with_the_file(pqv, tmp_path):
    """pqv_parquet_page_headers: every page of a column chunk, in file order -- sizes chain up to the chunk's compressed size,
    the data pages' value counts add up to the chunk's, encodings are what the writer was asked for; garbage is refused."""
    import ctypes as C
    from pq_vector_amd import _ffi
    L = _ffi.lib()
    rng = np.random.default_rng(8)
    n, dim = 20000, 16
    vec = rng.integers(0, 50, (n, dim)).astype(np.float32)
    col = pa.ListArray.from_arrays(pa.array(np.arange(0, (n + 1) * dim, dim, dtype=np.int32)), pa.array(vec.reshape(-1)))
    for use_dict, codec in ((False, "NONE"), (True, "SNAPPY")):
        path = str(tmp_path / f"h{int(use_dict)}.parquet")
        pq.write_table(pa.table({"emb": col}), path, compression=codec, use_dictionary=use_dict, data_page_size=16 * 1024,
                       write_statistics=True)
        meta = pq.ParquetFile(path).metadata
        cm = meta.row_group(0).column(0)
        start = cm.dictionary_page_offset if cm.has_dictionary_page and cm.dictionary_page_offset else cm.data_page_offset
        raw = open(path, "rb").read()[start:start + cm.total_compressed_size]
        buf = (C.c_uint8 * len(raw)).from_buffer_copy(raw)
        out = np.full((4096, 8), -7, np.int32)
        npg = C.c_uint32(0)
        assert L.pqv_parquet_page_headers(buf, len(raw), 0, 4096, out.ctypes.data_as(C.POINTER(C.c_int32)), C.byref(npg)) == 0
        h = out[:npg.value]
        assert npg.value > 3 and int((h[:, 1] + h[:, 2]).sum()) == len(raw)
        data = h[h[:, 0] == 0]
        assert int(data[:, 4].sum()) == cm.num_values and (data[:, 6] == 3).all() and (data[:, 7] == 3).all()
        if use_dict:
            assert h[0, 0] == 2 and h[0, 5] in (0, 2) and set(data[:, 5].tolist()) <= {2, 8}
        else:
            assert (h[:, 0] == 0).all() and (data[:, 5] == 0).all()
        # stop_values: only as many pages as hold that many values
        assert L.pqv_parquet_page_headers(buf, len(raw), int(data[0, 4]), 4096, out.ctypes.data_as(C.POINTER(C.c_int32)), C.byref(npg)) == 0
        assert npg.value == (2 if use_dict else 1)
        # a truncated chunk and garbage are errors, not crashes
        assert L.pqv_parquet_page_headers(buf, len(raw) - 5, 0, 4096, out.ctypes.data_as(C.POINTER(C.c_int32)), C.byref(npg)) != 0
        junk = (C.c_uint8 * 64)(*([0xFF] * 64))
        assert L.pqv_parquet_page_headers(junk, 64, 0, 16, out.ctypes.data_as(C.POINTER(C.c_int32)), C.byref(npg)) != 0
