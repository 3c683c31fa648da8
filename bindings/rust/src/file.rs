//! Path-taking builders: the reference's public call shapes over an indexed Parquet FILE, with the IVF core on the GPU.
//!
//! | pq-vector (reference)                                                       | here                                   |
//! |------------------------------------------------------------------------------|----------------------------------------|
//! | `IndexBuilder::new(source, column).n_clusters(..).max_iters(..).seed(..)` `src/ivf/parquet.rs:23-53` | [`IndexBuilder`] (same methods, + `device`, `workers`) |
//! | `.build_inplace()` `src/ivf/parquet.rs:55-68`, `.build_new(output)` `:70-87`  | [`IndexBuilder::build_inplace`], [`IndexBuilder::build_new`] |
//! | `TopkBuilder::new(path, &query).k(k)?.nprobe(n)?.search().await?` `src/ivf/search.rs:49-81` | [`TopkBuilder`] (same methods; `search` is `async`, `search_blocking` is the same call without a runtime) |
//! | `read_index_from_parquet` `src/ivf/parquet.rs:623-660`                        | [`read_index_from_parquet`]            |
//! | `read_parquet_with_embeddings` `src/ivf/parquet.rs:216-305`                   | [`load_embedding_column`] (column -> HBM, nothing larger than a batch on the host; `row_groups` = one shard of a shared file) |
//!
//! What differs from the reference on purpose: the column and the index of a file stay RESIDENT on the GPU between
//! queries (a per-path cache keyed by path, length and mtime), where `topk()` re-opens the file, re-parses the blob and
//! re-reads the candidate rows on every call (`src/ivf/search.rs:89-110`).
//!
//! Behind the `parquet-files` cargo feature (the `parquet` / `arrow` crates at the reference's versions).  Like the
//! rest of this crate it has never been compiled in the environment it was written in (no Rust toolchain there);
//! `tests/test_rust_binding.py` checks the method names against the reference's two `impl` blocks and every
//! `sys::pqv_*` call against the header.
#![cfg(feature = "parquet-files")]

use std::collections::HashMap;
use std::fs::{File, OpenOptions};
use std::io::{Read, Seek, SeekFrom, Write};
use std::num::NonZeroUsize;
use std::path::{Path, PathBuf};
use std::sync::{Arc, Mutex, OnceLock};
use std::time::SystemTime;

use arrow::array::{Array, Float32Array, Float64Array, ListArray};
use arrow::datatypes::DataType;
use parquet::arrow::arrow_reader::ParquetRecordBatchReaderBuilder;
use parquet::arrow::{ArrowWriter, ProjectionMask};
use parquet::file::metadata::{FileMetaData, KeyValue, ParquetMetaData, ParquetMetaDataBuilder, ParquetMetaDataReader, ParquetMetaDataWriter};
use parquet::basic::Encoding;
use parquet::file::properties::{EnabledStatistics, WriterProperties};
use parquet::schema::types::ColumnPath;

use crate::{sys, Corpus, Index, Result, SearchResult};

const INDEX_MAGIC: &[u8; 10] = b"PQ_VECTOR1"; // src/ivf/parquet.rs:106
const KEY_OFFSET: &str = "pq_vector_index_offset"; // :109
const KEY_COLUMN: &str = "pq_vector_embedding_column"; // :112
const TAIL: u64 = 8; // 4-byte metadata length + "PAR1"

// ---------------------------------------------------------------------------------------------------------------------
// column -> HBM
// ---------------------------------------------------------------------------------------------------------------------

/// Half-open range of row groups: one shard of a file that several GPUs share.  `None` = the whole file.
pub type RowGroupRange = Option<(usize, usize)>;

/// Shard `rank` of `world` of ONE file: a contiguous row-group range cut at the row-group boundary nearest to
/// `rank * n / world`, and the file-global row id of its first row = the prefix sum of the row groups before it
/// (`src/df_vector/access.rs:128-144`).  Returns `(rg_lo, rg_hi, row_base, n_rows)`: `pqv_shard_row_groups`, which
/// `pq_vector_amd.sharding.shard_row_groups` calls too.
pub fn shard_row_groups(rank: usize, world: usize, rg_rows: &[u64]) -> Result<(usize, usize, u64, u64)> {
    let (mut lo, mut hi, mut base, mut n) = (0u32, 0u32, 0u64, 0u64);
    // ONE implementation of the rule, in the library (host only; no device is touched)
    // (the ABI takes u32 counts: a value that does not fit is an error, not a silent truncation)
    let (n_rg, rank32, world32) = (u32::try_from(rg_rows.len())?, u32::try_from(rank)?, u32::try_from(world)?);
    crate::check(unsafe { sys::pqv_shard_row_groups(rg_rows.as_ptr(), n_rg, rank32, world32, &mut lo, &mut hi, &mut base, &mut n) })?;
    Ok((lo as usize, hi as usize, base, n))
}

/// The embedding column of `path` (or of one row-group range of it) as a resident `[n, dim]` f32 matrix.  Batches are
/// validated with the reference's checks and messages (`src/ivf/parquet.rs:231-280`) and go through the corpus' pinned
/// staging buffers as asynchronous DMAs ([`Corpus::write_rows`]); Float64 values are narrowed on the device (`:246-256`).
pub fn load_embedding_column(path: &Path, column: &str, device: usize, row_groups: RowGroupRange) -> Result<Corpus> {
    let builder = ParquetRecordBatchReaderBuilder::try_new(File::open(path)?)?;
    let schema = builder.schema().clone();
    let col_idx = schema.index_of(column).map_err(|_| format!("Column '{column}' not found"))?;
    let meta = builder.metadata().clone();
    let n_rg = meta.num_row_groups();
    let (lo, hi) = row_groups.unwrap_or((0, n_rg));
    if lo > hi || hi > n_rg {
        return Err(format!("row-group range [{lo}, {hi}) outside the file's {n_rg} row groups").into());
    }
    let n_rows: i64 = (lo..hi).map(|i| meta.row_group(i).num_rows()).sum();
    if n_rows == 0 {
        return Err("Embedding column has no rows".into());
    }
    let mask = ProjectionMask::roots(builder.parquet_schema(), [col_idx]);
    let reader = builder.with_projection(mask).with_row_groups((lo..hi).collect()).with_batch_size(1 << 16).build()?;
    let mut corpus: Option<Corpus> = None;
    let mut at = 0usize;
    for batch in reader {
        let batch = batch?;
        let lists = batch.column(0).as_any().downcast_ref::<ListArray>().ok_or("Embedding column is not a list array")?;
        if lists.null_count() > 0 {
            return Err("Embedding column contains null rows".into());
        }
        let offsets = lists.value_offsets();
        let (first, last) = (offsets[0] as usize, offsets[lists.len()] as usize);
        let dim = match &corpus {
            Some(c) => c.dim(),
            None => {
                let d = (offsets[1] - offsets[0]) as usize;
                if d == 0 {
                    return Err("Embedding row has zero length".into());
                }
                corpus = Some(Corpus::with_capacity(device, n_rows as usize, d)?);
                d
            }
        };
        if offsets.windows(2).any(|w| (w[1] - w[0]) as usize != dim) {
            return Err("Embedding vectors have inconsistent dimensions".into());
        }
        let values = lists.values();
        if values.null_count() > 0 {
            return Err("Embedding values contain nulls".into());
        }
        let c = corpus.as_ref().unwrap();
        match values.data_type() {
            DataType::Float32 => c.write_rows(at, &values.as_any().downcast_ref::<Float32Array>().unwrap().values()[first..last])?,
            DataType::Float64 => c.write_rows_f64(at, &values.as_any().downcast_ref::<Float64Array>().unwrap().values()[first..last])?,
            _ => return Err("Embedding values are not float32/float64".into()),
        }
        at += lists.len();
    }
    let mut corpus = corpus.ok_or("Embedding column has no rows")?;
    corpus.finish(at)?;
    Ok(corpus)
}

// ---------------------------------------------------------------------------------------------------------------------
// index blob <-> file (N2)
// ---------------------------------------------------------------------------------------------------------------------

fn footer_metadata(path: &Path) -> Result<(ParquetMetaData, u64)> {
    let mut f = File::open(path)?;
    let len = f.metadata()?.len();
    if len < TAIL {
        return Err("Parquet file too small to contain a footer".into());
    }
    let mut tail = [0u8; 8];
    f.seek(SeekFrom::End(-(TAIL as i64)))?;
    f.read_exact(&mut tail)?;
    if &tail[4..] == b"PARE" {
        return Err("Encrypted parquet footers are not supported for in-place indexing".into());
    }
    let meta_len = u32::from_le_bytes([tail[0], tail[1], tail[2], tail[3]]) as u64;
    if meta_len + TAIL > len {
        return Err("Parquet footer length exceeds file size".into());
    }
    let meta = ParquetMetaDataReader::new().parse_and_finish(&File::open(path)?)?;
    Ok((meta, len))
}

fn with_index_keys(meta: &ParquetMetaData, offset: u64, column: &str) -> ParquetMetaData {
    let fm = meta.file_metadata();
    let mut kv: Vec<KeyValue> = fm.key_value_metadata().cloned().unwrap_or_default();
    kv.retain(|e| e.key != KEY_OFFSET && e.key != KEY_COLUMN); // a rebuild replaces stale entries (:573-575)
    kv.push(KeyValue::new(KEY_OFFSET.to_string(), offset.to_string()));
    kv.push(KeyValue::new(KEY_COLUMN.to_string(), column.to_string()));
    let fm = FileMetaData::new(fm.version(), fm.num_rows(), fm.created_by().map(str::to_string), Some(kv), fm.schema_descr_ptr(), fm.column_orders().cloned());
    ParquetMetaDataBuilder::new(fm).set_row_groups(meta.row_groups().to_vec()).build()
}

/// `append_index_inplace` (`src/ivf/parquet.rs:542-611`): the blob goes where the footer metadata began -- magic, u64 LE
/// length, bytes -- and a footer with the two keys is written behind it.  Returns the blob's offset.
pub fn append_index_inplace(path: &Path, index: &Index, column: &str) -> Result<u64> {
    let (meta, len) = footer_metadata(path)?;
    let mut tail = [0u8; 4];
    let mut f = OpenOptions::new().read(true).write(true).open(path)?;
    f.seek(SeekFrom::End(-(TAIL as i64)))?;
    f.read_exact(&mut tail)?;
    let data_end = len - TAIL - u32::from_le_bytes(tail) as u64;
    // (the reference writes at `file_len - FOOTER_SIZE`, i.e. BEHIND the old metadata, which stays as dead bytes:
    //  :565-566 -- kept, so offsets agree with files the reference wrote)
    let offset = len - TAIL;
    let _ = data_end;
    let blob = index.to_bytes()?;
    f.seek(SeekFrom::Start(offset))?;
    f.write_all(INDEX_MAGIC)?;
    f.write_all(&(blob.len() as u64).to_le_bytes())?;
    f.write_all(&blob)?;
    ParquetMetaDataWriter::new(&mut f, &with_index_keys(&meta, offset, column)).finish()?;
    f.flush()?;
    Ok(offset)
}

/// `read_index_from_parquet` (`src/ivf/parquet.rs:623-660`) -> the index and the name of its embedding column.
pub fn read_index_from_parquet(path: &Path) -> Result<(Index, String)> {
    let (meta, _) = footer_metadata(path)?;
    let kv = meta.file_metadata().key_value_metadata().cloned().unwrap_or_default();
    let get = |k: &str| kv.iter().find(|e| e.key == k).and_then(|e| e.value.clone());
    let (offset, column) = match (get(KEY_OFFSET), get(KEY_COLUMN)) {
        (Some(o), Some(c)) => (o.parse::<u64>()?, c),
        _ => return Err("Missing pq-vector index metadata in parquet footer".into()),
    };
    let mut f = File::open(path)?;
    f.seek(SeekFrom::Start(offset))?;
    let mut head = [0u8; 18];
    f.read_exact(&mut head).map_err(|_| "pq-vector index payload is truncated")?;
    if &head[..10] != INDEX_MAGIC {
        return Err("Invalid pq-vector index magic".into());
    }
    let n = u64::from_le_bytes(head[10..18].try_into().unwrap()) as usize;
    let mut blob = vec![0u8; n];
    f.read_exact(&mut blob).map_err(|_| "pq-vector index bytes are truncated")?;
    Ok((Index::from_bytes(&blob)?, column))
}

// ---------------------------------------------------------------------------------------------------------------------
// IndexBuilder
// ---------------------------------------------------------------------------------------------------------------------

/// `IndexBuilder` (`src/ivf/parquet.rs:23-103`): same constructor, same knobs, same defaults, same validation texts.
#[derive(Debug, Clone)]
pub struct IndexBuilder {
    source: PathBuf,
    embedding_column: String,
    inner: crate::IndexBuilder,
    device: usize,
}

impl IndexBuilder {
    pub fn new(source: impl AsRef<Path>, embedding_column: impl AsRef<str>) -> Self {
        Self { source: source.as_ref().to_path_buf(), embedding_column: embedding_column.as_ref().to_string(), inner: crate::IndexBuilder::new(), device: 0 }
    }
    pub fn n_clusters(mut self, n_clusters: usize) -> Self {
        self.inner = self.inner.n_clusters(n_clusters);
        self
    }
    pub fn max_iters(mut self, max_iters: usize) -> Self {
        self.inner = self.inner.max_iters(max_iters);
        self
    }
    pub fn seed(mut self, seed: u64) -> Self {
        self.inner = self.inner.seed(seed);
        self
    }
    /// Not in the reference: which GPU holds the column while the index is built.
    pub fn device(mut self, device: usize) -> Self {
        self.device = device;
        self
    }
    /// Not in the reference: the `available_parallelism()` whose chunked k-means++ sum is being reproduced (`src/ivf/index.rs:259-265`).
    pub fn workers(mut self, workers: usize) -> Self {
        self.inner = self.inner.workers(workers);
        self
    }

    fn build_index(&self) -> Result<Index> {
        if self.embedding_column.is_empty() {
            return Err("Embedding column name must be non-empty".into()); // src/ivf/mod.rs:33
        }
        let corpus = load_embedding_column(&self.source, &self.embedding_column, self.device, None)?;
        self.inner.clone().build(&corpus)
    }

    /// Column -> HBM, index build on the GPU, blob + footer appended to `source` (`src/ivf/parquet.rs:55-68`).
    pub fn build_inplace(self) -> Result<()> {
        let index = self.build_index()?;
        append_index_inplace(&self.source, &index, &self.embedding_column)?;
        forget_path(&self.source);
        Ok(())
    }

    /// The same into a copy of the file (`src/ivf/parquet.rs:70-87`).  Layout as the reference writes it (`:315-377`): data pages, then
    /// `PQ_VECTOR1 | u64 length | blob` through the SAME writer behind its last flushed row group -- `bytes_written()` at that
    /// point is the index offset --, then ONE footer that carries the two keys.  Writer properties as there: one vector per data
    /// page of the embedding leaf (page size limit = one vector, row count limit 1), every column with the codec, dictionary flag,
    /// data encoding and statistics level of its first source chunk; the embedding leaf without a dictionary, with chunk-level
    /// statistics and none in the page headers.
    pub fn build_new(self, output: impl AsRef<Path>) -> Result<()> {
        let index = self.build_index()?;
        let src = ParquetRecordBatchReaderBuilder::try_new(File::open(&self.source)?)?;
        let meta = src.metadata().clone();
        let mut props = WriterProperties::builder().set_data_page_size_limit(index.dim().max(1) * 4).set_data_page_row_count_limit(1);
        let mut embedding_leaf: Option<ColumnPath> = None;
        if meta.num_row_groups() > 0 {
            for col in meta.row_group(0).columns() {
                let cp = ColumnPath::from(col.column_path().parts().to_vec());
                let mut uses_dict = false;
                let mut data_encoding = None;
                for e in col.encodings() {
                    match e {
                        Encoding::RLE_DICTIONARY | Encoding::PLAIN_DICTIONARY => uses_dict = true,
                        Encoding::RLE | Encoding::BIT_PACKED => {} // level encodings, not the values'
                        other => data_encoding = Some(other),
                    }
                }
                let stats = if col.statistics().is_some() { EnabledStatistics::Chunk } else { EnabledStatistics::None };
                props = props.set_column_compression(cp.clone(), col.compression()).set_column_dictionary_enabled(cp.clone(), uses_dict);
                if let Some(enc) = data_encoding {
                    if !uses_dict {
                        props = props.set_column_encoding(cp.clone(), enc);
                    }
                }
                props = props.set_column_statistics_enabled(cp.clone(), stats);
                if col.column_path().parts().first().map(String::as_str) == Some(self.embedding_column.as_str()) {
                    embedding_leaf = Some(cp);
                }
            }
        }
        let leaf = embedding_leaf.ok_or_else(|| format!("Column '{}' not found", self.embedding_column))?;
        let props = props
            .set_column_dictionary_enabled(leaf.clone(), false)
            .set_column_statistics_enabled(leaf.clone(), EnabledStatistics::Chunk)
            .set_column_write_page_header_statistics(leaf, false)
            .build();
        let schema = src.schema().clone();
        let mut w = ArrowWriter::try_new(File::create(output.as_ref())?, schema, Some(props))?;
        for batch in src.build()? {
            w.write(&batch?)?;
        }
        w.flush()?;
        let offset = w.bytes_written();
        let blob = index.to_bytes()?;
        w.write_all(INDEX_MAGIC)?;
        w.write_all(&(blob.len() as u64).to_le_bytes())?;
        w.write_all(&blob)?;
        w.append_key_value_metadata(KeyValue::new(KEY_OFFSET.to_string(), offset.to_string()));
        w.append_key_value_metadata(KeyValue::new(KEY_COLUMN.to_string(), self.embedding_column.clone()));
        w.close()?;
        Ok(())
    }
}

// ---------------------------------------------------------------------------------------------------------------------
// TopkBuilder over a path, with the file resident between calls
// ---------------------------------------------------------------------------------------------------------------------

struct Resident {
    searcher: *mut sys::PqvSearcher, // freed before the corpus and the index it points into
    _index: Index,
    _corpus: Corpus,
}
unsafe impl Send for Resident {}
unsafe impl Sync for Resident {}
impl Drop for Resident {
    fn drop(&mut self) {
        unsafe { sys::pqv_searcher_free(self.searcher) }
    }
}

type CacheKey = (PathBuf, u64, SystemTime, usize);
fn cache() -> &'static Mutex<HashMap<CacheKey, Arc<Resident>>> {
    static C: OnceLock<Mutex<HashMap<CacheKey, Arc<Resident>>>> = OnceLock::new();
    C.get_or_init(|| Mutex::new(HashMap::new()))
}
fn forget_path(path: &Path) {
    if let Ok(real) = path.canonicalize() {
        cache().lock().unwrap().retain(|k, _| k.0 != real);
    }
}

/// Index + embedding column of an indexed file, resident on `device`; one entry per (canonical path, length, mtime).
fn resident(path: &Path, device: usize) -> Result<Arc<Resident>> {
    let st = std::fs::metadata(path)?;
    let key: CacheKey = (path.canonicalize()?, st.len(), st.modified()?, device);
    if let Some(hit) = cache().lock().unwrap().get(&key) {
        return Ok(hit.clone());
    }
    let (index, column) = read_index_from_parquet(path)?;
    let mut corpus = load_embedding_column(path, &column, device, None)?;
    let mut raw = std::ptr::null_mut();
    // one f32 copy of the column stays in HBM either way (PQV_RELEASE_IF_COPIED, include/pqv.h)
    crate::check(unsafe { sys::pqv_searcher_create(index.raw, corpus.raw_mut(), sys::PQV_LAYOUT_IVF_ORDERED | sys::PQV_RELEASE_IF_COPIED, &mut raw) })?;
    let r = Arc::new(Resident { searcher: raw, _index: index, _corpus: corpus });
    let mut c = cache().lock().unwrap();
    c.clear(); // one resident file at a time by default
    c.insert(key, r.clone());
    Ok(r)
}

/// `TopkBuilder` (`src/ivf/search.rs:49-81`).
#[derive(Debug, Clone)]
pub struct TopkBuilder<'a> {
    parquet_path: PathBuf,
    query: &'a [f32],
    k: Option<NonZeroUsize>,
    nprobe: Option<NonZeroUsize>,
    device: usize,
}

impl<'a> TopkBuilder<'a> {
    pub fn new(parquet_path: impl AsRef<Path>, query: &'a [f32]) -> Self {
        Self { parquet_path: parquet_path.as_ref().to_path_buf(), query, k: None, nprobe: None, device: 0 }
    }
    pub fn k(mut self, k: usize) -> Result<Self> {
        self.k = Some(NonZeroUsize::new(k).ok_or("k must be > 0")?);
        Ok(self)
    }
    pub fn nprobe(mut self, nprobe: usize) -> Result<Self> {
        self.nprobe = Some(NonZeroUsize::new(nprobe).ok_or("nprobe must be > 0")?);
        Ok(self)
    }
    /// Not in the reference: the GPU the file is (or becomes) resident on.
    pub fn device(mut self, device: usize) -> Self {
        self.device = device;
        self
    }
    /// The reference's call shape (`.search().await?`); nothing inside awaits -- the GPU call is synchronous.
    pub async fn search(self) -> Result<Vec<SearchResult>> {
        self.search_blocking()
    }
    pub fn search_blocking(self) -> Result<Vec<SearchResult>> {
        let k = self.k.ok_or("k must be set")?.get();
        let nprobe = self.nprobe.ok_or("nprobe must be set")?.get();
        let r = resident(&self.parquet_path, self.device)?;
        let (mut rows, mut dist, mut found) = (vec![0u32; k], vec![0f32; k], 0u32);
        crate::check(unsafe {
            sys::pqv_topk(r.searcher, self.query.as_ptr(), 1, self.query.len() as u32, k as u32, nprobe as u32, 0, sys::PQV_L2SQ_REF4, 1,
                          rows.as_mut_ptr(), dist.as_mut_ptr(), &mut found, std::ptr::null_mut())
        })?;
        Ok((0..found as usize).map(|i| SearchResult { row_idx: rows[i], distance: dist[i] }).collect())
    }
}
