"""Parquet side of the hot path (SURVEY.md 8f N1 + N2): stream the embedding column into HBM,
embed / read the IVF index blob in the file the way pq-vector does, so files written here are
readable by the reference and vice versa.

  N1  load_embedding_column    read_parquet_with_embeddings   src/ivf/parquet.rs:216-305
  N2  append_index_inplace     append_index_inplace           src/ivf/parquet.rs:542-611
      write_parquet_with_index write_parquet_with_index       src/ivf/parquet.rs:316-377
      read_index_from_parquet  read_index_from_parquet        src/ivf/parquet.rs:191-208
      has_pq_vector_index      has_pq_vector_index            src/ivf/parquet.rs:187-189

On-disk format:
    [PAR1][row groups ...][old Thrift footer, now dead bytes]
    ["PQ_VECTOR1"][u64 LE blob_len][blob (IvfIndex::to_bytes)]        <- pq_vector_index_offset
    [Thrift FileMetaData + KV pq_vector_index_offset / pq_vector_embedding_column][u32 len]["PAR1"]

The footer is rewritten with a small generic Thrift-compact re-serialiser: every top-level
FileMetaData field is copied byte for byte except key_value_metadata (field 5), so the schema,
row groups, column orders and created_by survive untouched whatever writer produced the file.
This is host I/O glue (pyarrow + bytes); no distance arithmetic happens here.
"""
import os
import struct

import numpy as np

from . import _ffi
from .api import Corpus, Index, PqvError

MAGIC = b"PQ_VECTOR1"                                   # src/ivf/parquet.rs:106
OFFSET_KEY = "pq_vector_index_offset"                   # :109
COLUMN_KEY = "pq_vector_embedding_column"               # :112
FOOTER_SIZE = 8


def _err(msg):
    return PqvError(_ffi.PQV_ERR_INVALID, msg)


# ---------------------------------------------------------------------------------------
# Thrift compact protocol: just enough to splice FileMetaData.key_value_metadata
# ---------------------------------------------------------------------------------------
T_STOP, T_TRUE, T_FALSE, T_BYTE, T_I16, T_I32, T_I64, T_DOUBLE, T_BINARY, T_LIST, T_SET, T_MAP, \
    T_STRUCT, T_UUID = range(14)


def _varint(buf, pos):
    shift = val = 0
    while True:
        b = buf[pos]
        pos += 1
        val |= (b & 0x7F) << shift
        if not b & 0x80:
            return val, pos
        shift += 7


def _put_varint(v):
    out = bytearray()
    while True:
        b = v & 0x7F
        v >>= 7
        if v:
            out.append(b | 0x80)
        else:
            out.append(b)
            return bytes(out)


def _zigzag_decode(v):
    return (v >> 1) ^ -(v & 1)


def _zigzag_encode(v):
    return (v << 1) ^ (v >> 63)


def _skip(buf, pos, ttype):
    """Return the position just past one value of `ttype` starting at pos."""
    if ttype in (T_TRUE, T_FALSE):
        return pos
    if ttype == T_BYTE:
        return pos + 1
    if ttype in (T_I16, T_I32, T_I64):
        return _varint(buf, pos)[1]
    if ttype == T_DOUBLE:
        return pos + 8
    if ttype == T_UUID:
        return pos + 16
    if ttype == T_BINARY:
        n, pos = _varint(buf, pos)
        return pos + n
    if ttype in (T_LIST, T_SET):
        h = buf[pos]
        pos += 1
        n, et = h >> 4, h & 0x0F
        if n == 15:
            n, pos = _varint(buf, pos)
        for _ in range(n):
            pos = pos + 1 if et in (T_TRUE, T_FALSE) else _skip(buf, pos, et)
        return pos
    if ttype == T_MAP:
        n, pos = _varint(buf, pos)
        if n:
            kv = buf[pos]
            pos += 1
            for _ in range(n):
                pos = _skip(buf, pos, kv >> 4)
                pos = _skip(buf, pos, kv & 0x0F)
        return pos
    if ttype == T_STRUCT:
        while True:
            h = buf[pos]
            pos += 1
            if h == T_STOP:
                return pos
            if not h >> 4:
                pos = _varint(buf, pos)[1]
            pos = _skip(buf, pos, h & 0x0F)
    raise _err(f"unsupported Thrift type {ttype} in parquet footer")


def _struct_fields(buf):
    """Top-level fields of a struct: [(field_id, type, raw value bytes)]."""
    fields, pos, last = [], 0, 0
    while True:
        h = buf[pos]
        pos += 1
        if h == T_STOP:
            return fields
        delta, ttype = h >> 4, h & 0x0F
        if delta:
            fid = last + delta
        else:
            z, pos = _varint(buf, pos)
            fid = _zigzag_decode(z)
        end = _skip(buf, pos, ttype)
        fields.append((fid, ttype, bytes(buf[pos:end])))
        pos, last = end, fid


def _emit_struct(fields):
    out, last = bytearray(), 0
    for fid, ttype, raw in sorted(fields, key=lambda f: f[0]):
        delta = fid - last
        if 0 < delta <= 15:
            out.append((delta << 4) | ttype)
        else:
            out.append(ttype)
            out += _put_varint(_zigzag_encode(fid))
        out += raw
        last = fid
    out.append(T_STOP)
    return bytes(out)


def _decode_kv_list(raw):
    h = raw[0]
    pos = 1
    n = h >> 4
    if n == 15:
        n, pos = _varint(raw, pos)
    out = []
    for _ in range(n):
        end = _skip(raw, pos, T_STRUCT)
        key = val = None
        for fid, ttype, v in _struct_fields(raw[pos:end]):
            if ttype == T_BINARY:
                ln, p = _varint(v, 0)
                s = bytes(v[p:p + ln])
                if fid == 1:
                    key = s
                elif fid == 2:
                    val = s
        out.append((key, val))
        pos = end
    return out


def _encode_kv_list(kvs):
    out = bytearray()
    n = len(kvs)
    if n < 15:
        out.append((n << 4) | T_STRUCT)
    else:
        out.append(0xF0 | T_STRUCT)
        out += _put_varint(n)
    for key, val in kvs:
        fields = [(1, T_BINARY, _put_varint(len(key)) + key)]
        if val is not None:
            fields.append((2, T_BINARY, _put_varint(len(val)) + val))
        out += _emit_struct(fields)
    return bytes(out)


def _read_footer(path):
    size = os.path.getsize(path)
    if size < FOOTER_SIZE:
        raise _err("Parquet file too small to contain a footer")               # parquet.rs:549
    with open(path, "rb") as f:
        f.seek(size - FOOTER_SIZE)
        tail = f.read(FOOTER_SIZE)
        if tail[4:] == b"PARE":
            raise _err("Encrypted parquet footers are not supported for in-place indexing")  # :557
        if tail[4:] != b"PAR1":
            raise _err("Invalid Parquet file: corrupt footer")
        meta_len = struct.unpack("<I", tail[:4])[0]
        if meta_len + FOOTER_SIZE > size:
            raise _err("Parquet footer length exceeds file size")              # :562
        f.seek(size - FOOTER_SIZE - meta_len)
        meta = f.read(meta_len)
    return size, meta


def _footer_kv(meta):
    for fid, ttype, raw in _struct_fields(meta):
        if fid == 5 and ttype == T_LIST:
            return _decode_kv_list(raw)
    return []


# ---------------------------------------------------------------------------------------
# N2: index blob in the file
# ---------------------------------------------------------------------------------------
def read_index_metadata(path):
    """(offset, embedding_column) from the footer KV, or None (parse_index_metadata :120-143)."""
    try:
        _, meta = _read_footer(path)
        kv = dict((k, v) for k, v in _footer_kv(meta) if k is not None)
    except (IndexError, struct.error, UnicodeDecodeError) as e:      # truncated / corrupt Thrift footer
        raise PqvError(_ffi.PQV_ERR_FORMAT, f"malformed parquet footer: {e}")
    off, col = kv.get(OFFSET_KEY.encode()), kv.get(COLUMN_KEY.encode())
    if off is None or col is None:
        return None
    col = col.decode()
    if not col.strip():
        raise _err("Embedding column name cannot be empty")
    # Rust's `str::parse::<u64>` (parquet.rs:133): ASCII digits only (an optional leading '+'), no blanks, no '_', no sign
    txt = off.decode("ascii", errors="replace")
    digits = txt[1:] if txt.startswith("+") else txt
    if not digits or not all("0" <= ch <= "9" for ch in digits) or int(digits) >= 1 << 64:
        raise PqvError(_ffi.PQV_ERR_FORMAT, f"invalid pq-vector index offset in parquet footer: {txt!r}")
    offset = int(digits)
    import os
    if offset > os.path.getsize(path):
        raise PqvError(_ffi.PQV_ERR_FORMAT, "pq-vector index offset lies beyond the end of the file")
    return offset, col


def has_pq_vector_index(path):
    return read_index_metadata(path) is not None


def read_index_payload(payload):
    """read_index_from_payload (src/ivf/parquet.rs:151-174)."""
    header = len(MAGIC) + 8
    if len(payload) < header:
        raise _err("pq-vector index payload is truncated")
    if payload[:len(MAGIC)] != MAGIC:
        raise _err("Invalid pq-vector index magic")
    n = struct.unpack("<Q", payload[len(MAGIC):header])[0]
    if len(payload) < header + n:
        raise _err("pq-vector index bytes are truncated")
    return Index.from_bytes(payload[header:header + n])


def read_index_from_parquet(path):
    meta = read_index_metadata(path)
    if meta is None:
        raise _err("Missing pq-vector index metadata in parquet footer")       # :197
    offset, column = meta
    with open(path, "rb") as f:
        f.seek(offset)
        payload = f.read()                                                     # read_to_end :204
    try:
        return read_index_payload(payload), column
    except PqvError as e:
        raise PqvError(e.code, f"Failed to decode pq-vector index payload at offset {offset}: {e.message}")


def append_index_inplace(path, index, embedding_column):
    """Overwrite the 8-byte tail with MAGIC | len | blob, then a new footer whose KV carries the
    offset and the column name; the old Thrift footer stays in the body as dead bytes and
    stale pq_vector_* entries are replaced (src/ivf/parquet.rs:542-611)."""
    size, meta = _read_footer(path)
    index_offset = size - FOOTER_SIZE                                          # :565-566
    fields = _struct_fields(meta)
    kvs = []
    for fid, ttype, raw in fields:
        if fid == 5 and ttype == T_LIST:
            kvs = _decode_kv_list(raw)
    kvs = [(k, v) for k, v in kvs if k not in (OFFSET_KEY.encode(), COLUMN_KEY.encode())]  # :573-575
    kvs.append((OFFSET_KEY.encode(), str(index_offset).encode()))
    kvs.append((COLUMN_KEY.encode(), embedding_column.encode()))
    fields = [f for f in fields if f[0] != 5] + [(5, T_LIST, _encode_kv_list(kvs))]
    new_meta = _emit_struct(fields)
    blob = index.to_bytes()
    with open(path, "r+b") as f:
        f.seek(index_offset)
        f.write(MAGIC)
        f.write(struct.pack("<Q", len(blob)))
        f.write(blob)
        f.write(new_meta)
        f.write(struct.pack("<I", len(new_meta)))
        f.write(b"PAR1")
        f.truncate()
    return index_offset


def write_parquet_with_index(source, output, index, embedding_column):
    """build_new (src/ivf/parquet.rs:316-377): a copy of `source` whose embedding column is
    stored one vector per data page (page limit dim*4 bytes, no dictionary, chunk-level
    statistics) so single rows can be fetched by page index, followed by the index blob.
    The blob sits after the copy's footer exactly as in the in-place layout (standard readers
    ignore it either way)."""
    import pyarrow.parquet as pq
    src = pq.ParquetFile(source)
    names = src.schema_arrow.names
    other = [n for n in names if n != embedding_column]
    rg_rows = src.metadata.row_group(0).num_rows if src.metadata.num_row_groups else 1 << 20
    # per-LEAF codec and dictionary use of the source (collect_column_write_options, parquet.rs:322-336).  parquet-cpp
    # looks column properties up by the full dotted leaf path ("emb.list.element"), so the dicts are keyed by
    # path_in_schema; a top-level name would silently leave every nested leaf -- the embedding column itself -- uncompressed.
    compression, dict_cols = {}, []
    emb_prefix = embedding_column + "."
    if src.metadata.num_row_groups:
        rg0 = src.metadata.row_group(0)
        for j in range(rg0.num_columns):
            col = rg0.column(j)
            path = col.path_in_schema
            codec = "NONE" if col.compression == "UNCOMPRESSED" else col.compression
            # (writers name a list's child "element" or "item"; the copy is written with this pyarrow's spelling, so
            #  both are registered -- the reference pairs source and output leaves by position, parquet.rs:328)
            for alt in {path, path.replace(".list.item", ".list.element"), path.replace(".list.element", ".list.item")}:
                compression.setdefault(alt, codec)
            is_emb = path == embedding_column or path.startswith(emb_prefix)
            uses_dict = any("DICTIONARY" in str(e) for e in col.encodings)
            if uses_dict and not is_emb:           # the embedding column is written without a dictionary (parquet.rs:352)
                dict_cols.append(path)
    else:
        dict_cols = list(other)
    # The reference closes a data page after ONE row (set_data_page_row_count_limit(1), page size limit = one vector).
    # parquet-cpp only looks at the page size every `write_batch_size` leaf values, so that must be one vector's worth:
    # the embedding column then gets exactly one vector per page (page-index reads of single rows stay single-page).
    writer = pq.ParquetWriter(output, src.schema_arrow, data_page_size=max(1, index.dim * 4),
                              write_batch_size=max(1, index.dim),
                              use_dictionary=dict_cols, write_statistics=True, write_page_index=True,
                              compression=compression)
    try:
        for i in range(src.metadata.num_row_groups):
            writer.write_table(src.read_row_group(i), row_group_size=rg_rows)
    finally:
        writer.close()
    return append_index_inplace(output, index, embedding_column)


# ---------------------------------------------------------------------------------------
# N1: embedding column -> HBM
# ---------------------------------------------------------------------------------------
def _check_column_type(pf, column):
    import pyarrow as pa
    if column not in pf.schema_arrow.names:
        raise _err(f"Column '{column}' not found")
    typ = pf.schema_arrow.field(column).type
    if not (pa.types.is_list(typ) or pa.types.is_large_list(typ) or pa.types.is_fixed_size_list(typ)):
        raise _err("Embedding column is not a list array")
    if not (pa.types.is_float32(typ.value_type) or pa.types.is_float64(typ.value_type)):
        raise _err("Embedding values are not float32/float64")
    return typ


def _batch_rows(arr, typ, dim):
    """One decoded batch of the column -> ([rows, dim] float array, dim), with the reference's checks and messages
    (src/ivf/parquet.rs:231-280); `dim` is None until the first batch has fixed it."""
    import pyarrow as pa
    if arr.null_count > 0:
        raise _err("Embedding column contains null rows")
    n = len(arr)
    flat = arr.flatten()                      # honours the slice offsets
    if flat.null_count > 0:
        raise _err("Embedding values contain nulls")
    if pa.types.is_fixed_size_list(typ):
        lens = np.full(n, typ.list_size, dtype=np.int64)
    else:
        lens = np.diff(arr.offsets.to_numpy())
    if (lens == 0).any():
        raise _err("Embedding row has zero length")
    if dim is None:
        dim = int(lens[0])
    if (lens != dim).any():
        raise _err("Embedding vectors have inconsistent dimensions")
    return flat.to_numpy(zero_copy_only=False).reshape(n, dim), dim


def _column_chunks(path, column, batch_rows=1 << 16, row_groups=None):
    """Yields validated [rows, dim] float arrays (f32 or f64) of the column, batch by batch."""
    import pyarrow.parquet as pq
    pf = pq.ParquetFile(path)
    typ = _check_column_type(pf, column)
    dim = None
    for batch in pf.iter_batches(batch_size=batch_rows, columns=[column], row_groups=row_groups):
        arr = batch.column(0)
        if len(arr) == 0:
            continue
        vals, dim = _batch_rows(arr, typ, dim)
        yield vals
    if dim is None and row_groups is None:
        raise _err("Embedding column has no rows")


# ---------------------------------------------------------------------------------------
# N1 fast path: the data pages of the embedding leaf, walked directly
# ---------------------------------------------------------------------------------------
# (Parquet's deprecated 'LZ4' is Hadoop-framed blocks -- pyarrow's 'lz4' codec is the FRAME format and cannot read them: such a
#  chunk is "not mine" up front and goes through the Arrow reader; LZ4_RAW is the plain block format)
_CODECS = {"UNCOMPRESSED": None, "SNAPPY": "snappy", "ZSTD": "zstd", "LZ4_RAW": "lz4_raw", "GZIP": "gzip", "BROTLI": "brotli"}


def _i32_fields(raw):
    """{field id: value} of a Thrift struct's integer fields, {field id: raw bytes} of its struct fields."""
    ints, structs = {}, {}
    for fid, ttype, val in _struct_fields(raw):
        if ttype in (T_I16, T_I32, T_I64):
            ints[fid] = _zigzag_decode(_varint(val, 0)[0])
        elif ttype == T_STRUCT:
            structs[fid] = val
    return ints, structs


class _PagePlan:
    """What _plan_pages found: the list length, the value type and one task per data page."""
    __slots__ = ("dim", "f64", "esz", "max_def", "def_bw", "tasks", "mm", "n_rows")


def _page_levels(L, addr, blen, nv, dim, max_def, def_bw):
    """Check the two level runs at the head of a v1 data page body; returns the offset of the values, or None."""
    import ctypes
    p = 0
    for bw, mode, expect in ((1, 1, dim), (def_bw, 0, max_def)):
        if p + 4 > blen:
            return None
        n = int.from_bytes(ctypes.string_at(addr + p, 4), "little")
        if p + 4 + n > blen:
            return None
        if L.pqv_parquet_levels_check(ctypes.cast(ctypes.c_void_p(addr + p + 4), _ffi.u8p), n, bw, nv, mode, expect, None) != 0:
            return None
        p += 4 + n
    return p


def _rg_range(meta, row_groups):
    """(lo, hi, rows before lo, rows in [lo, hi)) of a half-open row-group range (None = the whole file)."""
    n_rg = meta.num_row_groups
    lo, hi = (0, n_rg) if row_groups is None else (int(row_groups[0]), int(row_groups[1]))
    if not (0 <= lo <= hi <= n_rg):
        raise _err(f"row-group range [{lo}, {hi}) outside the file's {n_rg} row groups")
    rows = [meta.row_group(i).num_rows for i in range(hi)]
    return lo, hi, int(sum(rows[:lo])), int(sum(rows[lo:hi]))


def _plan_pages(path, column, n_threads, on_page=None, row_groups=None):
    """Walk the data pages of the embedding leaf in the memory-mapped file WITHOUT touching a device: a no-null `List<f32|f64>`
    leaf stores its values contiguously behind the page's two level runs.  The level runs are CHECKED, not trusted
    (pqv_parquet_levels_check: every definition level at its maximum -- no null row, no null value, no empty list -- and a
    repetition level 0 exactly every `dim` values, `dim` discovered from the first page); uncompressed pages are all checked
    here, of a compressed chunk the first page (the others when they are decompressed for the upload).  Returns None -- the
    caller takes the Arrow path, which owns the reference's error messages -- for anything it does not handle: v2 pages, other
    encodings, pages that end inside a row, levels that differ.  `row_groups` = (lo, hi): only the column chunks of that
    half-open row-group range (a shard of the file; value positions count from the range's first row -- pages never cross a
    column chunk and rows never cross a row group, src/df_vector/access.rs:128-144)."""
    import ctypes
    from concurrent.futures import ThreadPoolExecutor
    import pyarrow as pa
    import pyarrow.parquet as pq
    pf = pq.ParquetFile(path)
    meta, schema = pf.metadata, pf.schema
    leaves = [j for j in range(meta.num_columns) if schema.column(j).path.split(".")[0] == column]
    if len(leaves) != 1 or meta.num_rows == 0:
        return None
    leaf = leaves[0]
    cs = schema.column(leaf)
    if cs.physical_type not in ("FLOAT", "DOUBLE") or cs.max_repetition_level != 1 or cs.max_definition_level < 1:
        return None
    plan = _PagePlan()
    plan.f64 = cs.physical_type == "DOUBLE"
    plan.esz = 8 if plan.f64 else 4
    plan.max_def, plan.def_bw = cs.max_definition_level, int(cs.max_definition_level).bit_length()
    plan.mm = mm = np.memmap(path, dtype=np.uint8, mode="r")
    rg_lo, rg_hi, _, plan.n_rows = _rg_range(meta, row_groups)
    if plan.n_rows == 0:
        return None
    base = mm.ctypes.data
    L = _ffi.lib()
    plan.dim = None
    tasks = []            # [first value position in values, file offset of the page body, body bytes, uncompressed bytes, n_values, encoding, dictionary, codec, value offset | None]
    values_before = 0
    for rg in range(rg_lo, rg_hi):
        cm = meta.row_group(rg).column(leaf)
        codec = _CODECS.get(cm.compression, "?")
        if codec == "?":
            return None
        start = cm.dictionary_page_offset if cm.has_dictionary_page and cm.dictionary_page_offset else cm.data_page_offset
        pos, end = int(start), int(start) + int(cm.total_compressed_size)
        left = int(cm.num_values)
        dictionary = None
        # the chunk's page headers in one native pass (Thrift compact; ~4 k pages of the reference's 4 GB bench file in a few ms)
        chunk = ctypes.cast(ctypes.c_void_p(base + pos), _ffi.u8p)
        cap = 4096
        while True:
            hdrs = np.empty((cap, 8), dtype=np.int32)
            n_pg = ctypes.c_uint32(0)
            if L.pqv_parquet_page_headers(chunk, end - pos, left, cap, hdrs.ctypes.data_as(ctypes.POINTER(ctypes.c_int32)), ctypes.byref(n_pg)) != 0:
                return None
            if n_pg.value < cap:
                break
            cap *= 8
        for ptype, hlen, csize, usize, nv, enc, denc, renc in hdrs[:n_pg.value].tolist():
            if left <= 0:
                break
            hend = pos + hlen
            if ptype == 2:                       # dictionary page: PLAIN values
                if enc not in (0, 2) or nv < 0:
                    return None
                body = bytes(mm[hend:hend + csize])
                if codec:
                    body = pa.Codec(codec).decompress(body, decompressed_size=usize).to_pybytes()
                dictionary = np.frombuffer(body, dtype="<f8" if plan.f64 else "<f4", count=nv).copy()
            elif ptype == 0:                     # data page v1
                if nv <= 0 or denc != 3 or renc != 3 or enc not in (0, 2, 8) or (enc != 0 and dictionary is None):
                    return None
                voff = None
                if plan.dim is None:             # the first data page fixes the list length (and is checked right here)
                    if codec:
                        buf = pa.Codec(codec).decompress(pa.py_buffer(mm[hend:hend + csize]), decompressed_size=usize)
                        addr, blen = buf.address, buf.size
                    else:
                        addr, blen = base + hend, csize
                    period = ctypes.c_uint64(0)
                    n = int.from_bytes(ctypes.string_at(addr, 4), "little") if blen >= 4 else -1
                    if n < 0 or 4 + n > blen:
                        return None
                    if L.pqv_parquet_levels_check(ctypes.cast(ctypes.c_void_p(addr + 4), _ffi.u8p), n, 1, nv, 1, 0, ctypes.byref(period)) != 0:
                        return None
                    plan.dim = int(period.value)
                    voff = _page_levels(L, addr, blen, nv, plan.dim, plan.max_def, plan.def_bw)
                    if voff is None:
                        return None
                if nv % plan.dim:
                    return None
                tasks.append([values_before, hend, csize, usize, nv, enc, dictionary, codec, voff])
                if on_page is not None:          # pipelined: the caller uploads (and checks) this page while the walk goes on
                    on_page(plan, tasks[-1])
                values_before += nv
                left -= nv
            else:                                # v2 data pages, index pages: the Arrow path
                return None
            pos = hend + csize
        if left != 0:
            return None
    if plan.dim is None or values_before != plan.n_rows * plan.dim:
        return None

    def check(t):                                # uncompressed pages: their level runs now
        if t[7] or t[8] is not None:
            return True
        t[8] = _page_levels(L, base + t[1], t[2], t[4], plan.dim, plan.max_def, plan.def_bw)
        if t[8] is None or (t[5] == 0 and t[2] - t[8] != t[4] * plan.esz):
            return False
        return True

    if on_page is None:
        with ThreadPoolExecutor(max_workers=max(1, n_threads)) as ex:
            if not all(ex.map(check, tasks)):
                return None
    plan.tasks = tasks
    return plan


def _page_uploader(plan, corpus):
    """One planned page -> the corpus: a PLAIN page goes from the page cache straight into the pinned staging buffers
    (pqv_corpus_write_rows from the mapped address), dictionary pages through pqv_parquet_dict_decode, compressed pages through
    pyarrow's codecs; level runs not checked by the plan are checked here.  Returns the payload bytes, or False: something did
    not check out, the Arrow path re-reads the column."""
    import ctypes
    import pyarrow as pa
    L = _ffi.lib()
    mm, base, esz, f64 = plan.mm, plan.mm.ctypes.data, plan.esz, plan.f64

    def do_page(t):
        first, off, csize, usize, nv, enc, dictionary, codec, voff = t
        dim = plan.dim
        buf = None
        if codec:
            buf = pa.Codec(codec).decompress(pa.py_buffer(mm[off:off + csize]), decompressed_size=usize)
            addr, blen = buf.address, buf.size
            voff = None
        else:
            addr, blen = base + off, csize
        if voff is None:
            voff = _page_levels(L, addr, blen, nv, dim, plan.max_def, plan.def_bw)
            if voff is None:
                return False
        row = first // dim
        if enc == 0:
            if blen - voff != nv * esz:
                return False
            corpus.write_rows_ptr(row, addr + voff, nv // dim, f64)
        else:
            out = np.empty(nv, dtype=np.float64 if f64 else np.float32)
            rc = L.pqv_parquet_dict_decode(ctypes.cast(ctypes.c_void_p(addr + voff), _ffi.u8p), blen - voff, dictionary.ctypes.data_as(_ffi.vp),
                                           dictionary.size, esz, nv, out.ctypes.data_as(_ffi.vp))
            if rc != 0:
                return False
            corpus.write_rows_ptr(row, out.ctypes.data, nv // dim, f64)
        del buf
        return nv * esz

    return do_page


_PAGE_RUN = 48          # uncompressed PLAIN pages handed to the library per call (the interpreter lock stays out of the per-page work)


def _plain_run_uploader(plan, corpus):
    """A run of uncompressed PLAIN pages -> Corpus.write_plain_pages; the payload bytes, or False."""
    base = plan.mm.ctypes.data

    def do_run(tasks):
        off = np.array([t[1] for t in tasks], dtype=np.uint64)
        ln = np.array([t[2] for t in tasks], dtype=np.uint32)
        first = np.array([t[0] for t in tasks], dtype=np.uint64)
        nv = np.array([t[4] for t in tasks], dtype=np.uint32)
        if corpus.write_plain_pages(base, off, ln, first, nv, plan.dim, plan.max_def, plan.f64) is not None:
            return False
        return int(nv.sum(dtype=np.uint64)) * plan.esz

    return do_run


def _is_plain_run_page(t):
    return t[5] == 0 and not t[7] and t[2] < (1 << 32) and t[4] < (1 << 32)


def _upload_pages(plan, corpus, n_threads, counters):
    """Every planned page through _page_uploader on `n_threads` threads."""
    from concurrent.futures import ThreadPoolExecutor
    with ThreadPoolExecutor(max_workers=max(1, n_threads)) as ex:
        for r in ex.map(_page_uploader(plan, corpus), plan.tasks):
            if r is False:
                return False
            counters[0] += r
    counters[1] = len(plan.tasks)
    return True


def _load_pages(path, column, corpus, dim, rg_off, n_threads, counters, row_groups=None):
    """plan + upload (tests drive the walker through this with a stand-in corpus)."""
    import pyarrow as pa
    try:
        plan = _plan_pages(path, column, n_threads, None, row_groups)
        if plan is None or plan.dim != dim:
            return False
        return _upload_pages(plan, corpus, n_threads, counters)
    except PqvError:
        raise
    except (pa.ArrowException, OSError, ValueError, IndexError, EOFError, struct.error):
        # a codec / mapping / page-framing problem: "not mine", as load_embedding_column treats it.  Anything else (a TypeError or
        # AttributeError out of the walker itself) is a bug and must surface, not turn into a silent fall-back to the Arrow path.
        return False


def load_embedding_column(path, column, device=0, readers=None, stats=None, row_groups=None):
    """The column -> one resident [n, dim] f32 matrix (src/ivf/parquet.rs:216-305; Float64 values are narrowed on the device,
    :246-256).  Row groups are decoded by `readers` threads (pyarrow releases the GIL while it decodes), each with its own file
    handle; every decoded batch goes through the corpus' pinned staging buffers as an asynchronous DMA (pqv_corpus_write_rows)
    at its row offset, so decoding batch i + 1 overlaps the upload of batch i and nothing larger than a batch is ever
    materialised on the host -- unlike the reference's Vec<f32> of the whole column (:226).  `stats` (a dict) receives rows,
    bytes, seconds and GB/s.

    `row_groups` = (lo, hi): only that half-open range of the file's row groups -- ONE shard of a file that several GPUs share
    (sharding.shard_row_groups cuts at row-group boundaries; the shard's row 0 is the range's first row, its file-global row
    base the prefix sum of the row groups before it, src/df_vector/access.rs:128-144)."""
    import os
    import threading
    import time
    import pyarrow.parquet as pq
    t0 = time.perf_counter()
    pf = pq.ParquetFile(path)
    typ = _check_column_type(pf, column)
    meta = pf.metadata
    rg_lo, rg_hi, _, n_rows = _rg_range(meta, row_groups)
    n_rg = rg_hi - rg_lo
    rg_off = np.zeros(n_rg + 1, dtype=np.int64)           # shard-local row offset of every row group of the range
    for i in range(n_rg):
        rg_off[i + 1] = rg_off[i] + meta.row_group(rg_lo + i).num_rows
    if n_rows == 0:
        raise _err("Embedding column has no rows")
    nthr_pages = max(1, readers or int(os.environ.get("PQV_LOADER_THREADS", "0")) or min(8, os.cpu_count() or 1))
    # the page-level walk first (PQV_PARQUET_PAGES=0: the Arrow reader only).  Its first data page fixes the dimension and is
    # checked before any device is touched; from then on every page is handed to the upload threads while the walk goes on.
    # Whatever the walk cannot take -- or does not like -- goes through pyarrow below, which validates its first batch (before a
    # device is touched, if the walk stopped at the first page) and raises the reference's messages.
    from concurrent.futures import ThreadPoolExecutor
    corpus, plan, futures, ex, t_first, t_create, t_walk = None, None, [], None, None, 0.0, 0.0
    pages = [0, 0]
    fast = False
    if os.environ.get("PQV_PARQUET_PAGES", "1") != "0":
        state = {}

        run = []

        def flush_run():
            if run:
                futures.append(ex.submit(state["run"], list(run)))
                run.clear()

        def on_page(pl, task):
            nonlocal corpus, ex, t_first, t_create
            if corpus is None:
                t_first = time.perf_counter() - t0
                corpus = Corpus.create(n_rows, pl.dim, device)
                t_create = time.perf_counter() - t0 - t_first
                state["do"] = _page_uploader(pl, corpus)
                state["run"] = _plain_run_uploader(pl, corpus)
                ex = ThreadPoolExecutor(max_workers=nthr_pages)
            if _is_plain_run_page(task) and task[8] is None:       # (a page the walk already checked goes alone: its value offset is known)
                run.append(task)
                if len(run) >= _PAGE_RUN:
                    flush_run()
            else:
                flush_run()
                futures.append(ex.submit(state["do"], task))

        try:
            try:
                plan = _plan_pages(path, column, nthr_pages, on_page, row_groups)
                if ex is not None:
                    flush_run()
                t_walk = time.perf_counter() - t0
            except PqvError:
                raise
            except Exception:
                plan = None
            fast = plan is not None
            for f in futures:
                try:
                    r = f.result()
                except PqvError:
                    raise
                except Exception:
                    r = False
                if r is False:
                    fast = False
                else:
                    pages[0] += r
            pages[1] = len(plan.tasks) if plan is not None else 0
        except PqvError:                       # a device error: nothing to fall back to
            if ex is not None:
                ex.shutdown(wait=True)
            if corpus is not None:
                corpus.close()
            raise
        if ex is not None:
            ex.shutdown(wait=True)
    if not fast:
        try:
            first = next(_column_chunks(path, column, batch_rows=4096, row_groups=list(range(rg_lo, rg_hi))), None)      # the first batch fixes the dimension
            if first is None:
                raise _err("Embedding column has no rows")
        except Exception:
            if corpus is not None:
                corpus.close()
            raise
        dim = first.shape[1]
        if corpus is None or plan is None or plan.dim != dim:                      # (else the walk's corpus is overwritten)
            if corpus is not None:
                corpus.close()
            t_first = time.perf_counter() - t0
            corpus = Corpus.create(n_rows, dim, device)
            t_create = time.perf_counter() - t0 - t_first
    else:
        dim = plan.dim
    nthr = max(1, min(nthr_pages, n_rg))
    errors, lock = [], threading.Lock()
    nbytes = [0]
    if fast:
        t_up = time.perf_counter() - t0
        corpus.finish(n_rows)
        if stats is not None:
            el = time.perf_counter() - t0
            stats.update({"walk_done_s": t_walk, "uploads_done_s": t_up})
            stats.update({"rows": int(n_rows), "dim": int(dim), "bytes": int(pages[0]), "seconds": el, "GBps": pages[0] / el / 1e9,
                          "row_groups": int(n_rg), "reader_threads": nthr_pages, "path": "data pages walked in the mapped file", "pages": pages[1],
                          "first_page_s": t_first, "corpus_create_s": t_create})
        return corpus

    def work(t):
        rg = t
        try:
            for rg in range(t, n_rg, nthr):
                at = int(rg_off[rg])
                for vals in _column_chunks(path, column, row_groups=[rg_lo + rg]):
                    if vals.shape[1] != dim:
                        raise _err("Embedding vectors have inconsistent dimensions")
                    corpus.write_rows(at, vals)
                    at += vals.shape[0]
                    with lock:
                        nbytes[0] += vals.nbytes
                    if errors and min(e[0] for e in errors) < rg:     # an EARLIER row group already failed: its error is the answer
                        return
                if at != int(rg_off[rg + 1]):
                    raise _err("Embedding column row count does not match the file metadata")
        except Exception as e:
            # the reference walks the batches in file order and returns the FIRST error in that order (parquet.rs:231-280): keep
            # (row group, error) and raise the lowest row group's after the join -- not whichever thread failed first in time
            with lock:
                errors.append((rg, e))

    threads = [threading.Thread(target=work, args=(t,), daemon=True) for t in range(nthr)]
    for th in threads:
        th.start()
    for th in threads:
        th.join()
    if errors:
        corpus.close()
        raise min(errors, key=lambda e: e[0])[1]
    corpus.finish(n_rows)
    if stats is not None:
        el = time.perf_counter() - t0
        stats.update({"rows": int(n_rows), "dim": int(dim), "bytes": int(nbytes[0]), "seconds": el, "GBps": nbytes[0] / el / 1e9,
                      "row_groups": int(n_rg), "reader_threads": nthr, "path": "pyarrow record batches"})
    return corpus
