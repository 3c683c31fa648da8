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
