//! Safe wrappers over `libpqv_hip.so`, mirroring the types pq-vector's callers use today so that the crate's
//! public API (`IndexBuilder`, `TopkBuilder`, `SearchResult`) keeps its shape while the IVF core runs on the GPU:
//!
//! | pq-vector (reference)                                          | here                                   |
//! |-----------------------------------------------------------------|----------------------------------------|
//! | `Embeddings { data: Vec<f32>, dim }` `src/ivf/mod.rs:72-102`    | [`Corpus`] (the column, resident in HBM) |
//! | `build_ivf_index(&Embeddings, IvfBuildConfig)` `src/ivf/index.rs:152-214` | [`IndexBuilder::build`]        |
//! | `IvfIndex::{to_bytes, from_bytes, dim}` `src/ivf/index.rs:53-128` | [`Index`]                            |
//! | `IvfIndex::candidate_rows` `:57-63`                             | [`Searcher::candidate_rows`]           |
//! | `topk()` / `TopkBuilder` / `SearchResult` `src/ivf/search.rs:41-142` | [`TopkBuilder`], [`SearchResult`], [`Searcher::topk`] |
//! | `update_topk_heap` `src/df_vector/exec.rs:457-484`              | [`RerankState::fold_batch`]            |
//! | `CandidateCursor` `src/df_vector/access.rs:193-243`             | [`CandidateCursor`]                    |
//! | one heap over all files `src/df_vector/exec.rs:264-267`         | [`ShardComm::exchange`] (RCCL all-gather + merge) |
//!
//! Not compiled in the environment this repository was built in (no Rust toolchain in the image);
//! `tests/test_rust_binding.py` keeps `sys.rs` in lock-step with `include/pqv.h`, and every `sys::pqv_*` call below is
//! checked by that test to exist with the right number of arguments.  Error texts are the library's, which are the
//! reference's own (`"k must be > 0"`, `"Query dimension mismatch: expected {}, got {}"`, ...).
pub mod sys;
/// The reference's path-taking call shapes -- `IndexBuilder::new(source, column).build_inplace()`, `TopkBuilder::new(path, &query)`
/// -- over an indexed Parquet file (cargo feature `parquet-files`).
#[cfg(feature = "parquet-files")]
pub mod file;

use std::ffi::CStr;
use std::num::NonZeroUsize;
use std::os::raw::{c_int, c_void};
use std::ptr;

pub type Error = Box<dyn std::error::Error + Send + Sync>;
pub type Result<T> = std::result::Result<T, Error>;

pub(crate) fn check(rc: c_int) -> Result<()> {
    if rc == sys::PQV_OK {
        return Ok(());
    }
    // thread-local message of the failing call on THIS thread
    let msg = unsafe { CStr::from_ptr(sys::pqv_last_error()) }.to_string_lossy().into_owned();
    Err(msg.into())
}

/// Number of usable HIP devices (0 without a GPU; the library has no CPU fallback).
pub fn device_count() -> usize {
    unsafe { sys::pqv_device_count() }.max(0) as usize
}

/// The embedding column in one GPU's HBM (replaces the materialised `Embeddings`).
pub struct Corpus {
    raw: *mut sys::PqvCorpus,
}
unsafe impl Send for Corpus {}
unsafe impl Sync for Corpus {}

impl Corpus {
    /// `Embeddings::new(data, dim)` + upload: `rows` is row-major `[n, dim]`.
    pub fn upload(device: usize, rows: &[f32], dim: usize) -> Result<Self> {
        if dim == 0 {
            return Err("Embedding dimension must be > 0".into()); // src/ivf/mod.rs:59
        }
        if rows.len() % dim != 0 {
            return Err("Embedding data length must be a multiple of dimension".into()); // src/ivf/mod.rs:86
        }
        let mut raw = ptr::null_mut();
        check(unsafe {
            sys::pqv_corpus_upload(device as c_int, rows.as_ptr(), (rows.len() / dim) as u64, dim as u32, &mut raw)
        })?;
        Ok(Self { raw })
    }

    /// Streaming form for `read_parquet_with_embeddings` (`src/ivf/parquet.rs:216-305`): reserve, then append
    /// one row group's values buffer at a time -- no `Vec<f32>` of the whole column.
    pub fn with_capacity(device: usize, capacity_rows: usize, dim: usize) -> Result<Self> {
        let mut raw = ptr::null_mut();
        check(unsafe { sys::pqv_corpus_create(device as c_int, capacity_rows as u64, dim as u32, &mut raw) })?;
        Ok(Self { raw })
    }

    pub fn append(&mut self, rows: &[f32]) -> Result<()> {
        let dim = self.dim();
        if rows.len() % dim != 0 {
            return Err("Embedding data length must be a multiple of dimension".into());
        }
        check(unsafe { sys::pqv_corpus_append(self.raw, rows.as_ptr(), (rows.len() / dim) as u64) })
    }

    /// Float64 columns are narrowed with `as f32` (`src/ivf/parquet.rs:246-256`).
    pub fn append_f64(&mut self, rows: &[f64]) -> Result<()> {
        let dim = self.dim();
        if rows.len() % dim != 0 {
            return Err("Embedding data length must be a multiple of dimension".into());
        }
        check(unsafe { sys::pqv_corpus_append_f64(self.raw, rows.as_ptr(), (rows.len() / dim) as u64) })
    }

    /// Streaming upload (`src/ivf/parquet.rs:262-286` batch by batch): rows `[row_offset, row_offset + rows.len() / dim)` are
    /// staged in pinned memory and DMA'd asynchronously; batches may come in any order, from several threads (`&self`).
    pub fn write_rows(&self, row_offset: usize, rows: &[f32]) -> Result<()> {
        let dim = self.dim();
        if rows.len() % dim != 0 {
            return Err("Embedding data length must be a multiple of dimension".into());
        }
        check(unsafe { sys::pqv_corpus_write_rows(self.raw, row_offset as u64, rows.as_ptr(), (rows.len() / dim) as u64) })
    }

    pub fn write_rows_f64(&self, row_offset: usize, rows: &[f64]) -> Result<()> {
        let dim = self.dim();
        if rows.len() % dim != 0 {
            return Err("Embedding data length must be a multiple of dimension".into());
        }
        check(unsafe { sys::pqv_corpus_write_rows_f64(self.raw, row_offset as u64, rows.as_ptr(), (rows.len() / dim) as u64) })
    }

    /// A run of uncompressed PLAIN v1 data pages of the embedding leaf, straight from the memory-mapped file `file`: page `i`'s
    /// body is `file[body_off[i] .. body_off[i] + body_len[i]]`.  Per page the level runs are verified -- a row starts exactly
    /// every `dim` values, every definition level is `max_def`: the checks of `src/ivf/parquet.rs:231-280`, made on the levels --
    /// and the values behind them go to rows `first_value[i] / dim ..`.  `Ok(None)`: all uploaded; `Ok(Some(i))`: page `i` is not
    /// such a page and nothing of the run was uploaded (read the column through the Arrow reader, which owns the error texts).
    pub fn write_plain_pages(&self, file: &[u8], body_off: &[u64], body_len: &[u32], first_value: &[u64], n_values: &[u32],
                             max_def: u32, f64_values: bool) -> Result<Option<usize>> {
        let n = body_off.len();
        if body_len.len() != n || first_value.len() != n || n_values.len() != n {
            return Err("page tables must have one entry per page".into());
        }
        for i in 0..n {
            if body_off[i].checked_add(body_len[i] as u64).map_or(true, |e| e > file.len() as u64) {
                return Err("a page body lies outside the mapped file".into());
            }
        }
        let mut bad = 0u32;
        let rc = unsafe {
            sys::pqv_corpus_write_plain_pages(self.raw, file.as_ptr(), body_off.as_ptr(), body_len.as_ptr(), first_value.as_ptr(),
                                              n_values.as_ptr(), n as u32, self.dim() as u32, max_def, f64_values as c_int, &mut bad)
        };
        if rc == 1 {
            return Ok(Some(bad as usize));
        }
        check(rc).map(|_| None)
    }

    /// Waits for every upload and sets the row count.
    pub fn finish(&mut self, n_rows: usize) -> Result<()> {
        check(unsafe { sys::pqv_corpus_finish(self.raw, n_rows as u64) })
    }

    /// (crate-internal: the handle, for entry points that take `pqv_corpus *`)
    #[allow(dead_code)]
    pub(crate) fn raw_mut(&mut self) -> *mut sys::PqvCorpus {
        self.raw
    }

    pub fn rows(&self) -> usize {
        unsafe { sys::pqv_corpus_rows(self.raw) as usize }
    }
    pub fn dim(&self) -> usize {
        unsafe { sys::pqv_corpus_dim(self.raw) as usize }
    }
}

impl Drop for Corpus {
    fn drop(&mut self) {
        unsafe { sys::pqv_corpus_free(self.raw) }
    }
}

/// `IvfIndex` (`src/ivf/index.rs:9-14`): dim, n_clusters, centroids, inverted lists.
pub struct Index {
    raw: *mut sys::PqvIndex,
}
unsafe impl Send for Index {}
unsafe impl Sync for Index {}

impl Index {
    /// `IvfIndex::from_bytes` (`src/ivf/index.rs:85-128`).
    pub fn from_bytes(bytes: &[u8]) -> Result<Self> {
        let mut raw = ptr::null_mut();
        check(unsafe { sys::pqv_index_from_bytes(bytes.as_ptr(), bytes.len(), &mut raw) })?;
        Ok(Self { raw })
    }

    /// `IvfIndex::to_bytes` (`src/ivf/index.rs:65-83`); byte-identical layout, so the existing
    /// `append_index_inplace` / `write_parquet_with_index` embed it unchanged.
    pub fn to_bytes(&self) -> Result<Vec<u8>> {
        let (mut buf, mut len) = (ptr::null_mut(), 0usize);
        check(unsafe { sys::pqv_index_to_bytes(self.raw, &mut buf, &mut len) })?;
        let out = if buf.is_null() || len == 0 { Vec::new() } else { unsafe { std::slice::from_raw_parts(buf, len) }.to_vec() };
        unsafe { sys::pqv_bytes_free(buf) };
        Ok(out)
    }

    pub fn dim(&self) -> usize {
        unsafe { sys::pqv_index_dim(self.raw) as usize }
    }
    pub fn n_clusters(&self) -> usize {
        unsafe { sys::pqv_index_n_clusters(self.raw) as usize }
    }
    pub fn centroids(&self) -> &[f32] {
        let n = self.dim() * self.n_clusters();
        unsafe { std::slice::from_raw_parts(sys::pqv_index_centroids(self.raw), n) }
    }
    /// `inverted_lists[c]`, ascending row ids.
    pub fn inverted_list(&self, cluster: usize) -> &[u32] {
        let off = unsafe { std::slice::from_raw_parts(sys::pqv_index_list_offsets(self.raw), self.n_clusters() + 1) };
        let rows = unsafe { std::slice::from_raw_parts(sys::pqv_index_list_rows(self.raw), sys::pqv_index_n_rows(self.raw) as usize) };
        &rows[off[cluster] as usize..off[cluster + 1] as usize]
    }
}

impl Drop for Index {
    fn drop(&mut self) {
        unsafe { sys::pqv_index_free(self.raw) }
    }
}

/// `IndexBuilder` (`src/ivf/parquet.rs:23-103`) over a resident column: same knobs, same defaults
/// (`max_iters` 20, `seed` 42, `n_clusters` = ceil(sqrt(n)) when unset), same validation texts.
#[derive(Debug, Clone)]
pub struct IndexBuilder {
    n_clusters: Option<usize>,
    max_iters: usize,
    seed: u64,
    workers: usize,
}

impl Default for IndexBuilder {
    fn default() -> Self {
        Self::new()
    }
}

impl IndexBuilder {
    pub fn new() -> Self {
        // `workers` is what the reference's worker_count() would see (src/ivf/index.rs:259-265): it fixes the
        // chunking of one f32 sum in k-means++, so it is part of the reproducible configuration.
        let workers = std::thread::available_parallelism().map(|n| n.get()).unwrap_or(1);
        Self { n_clusters: None, max_iters: 20, seed: 42, workers }
    }
    pub fn n_clusters(mut self, n_clusters: usize) -> Self {
        self.n_clusters = Some(n_clusters);
        self
    }
    pub fn max_iters(mut self, max_iters: usize) -> Self {
        self.max_iters = max_iters;
        self
    }
    pub fn seed(mut self, seed: u64) -> Self {
        self.seed = seed;
        self
    }
    pub fn workers(mut self, workers: usize) -> Self {
        self.workers = workers;
        self
    }

    /// `build_config()` + `build_ivf_index()`; the caller embeds `index.to_bytes()` exactly as
    /// `build_inplace` / `build_new` do today.
    pub fn build(self, corpus: &Corpus) -> Result<Index> {
        if self.max_iters == 0 {
            return Err("max_iters must be > 0".into()); // parquet.rs:90
        }
        let n_clusters = match self.n_clusters {
            Some(0) => return Err("n_clusters must be > 0".into()), // parquet.rs:93
            Some(v) => u32::try_from(v)?,
            None => 0, // the library applies ceil(sqrt(n)) (index.rs:161-167)
        };
        let mut raw = ptr::null_mut();
        check(unsafe {
            sys::pqv_index_build(corpus.raw, n_clusters, u32::try_from(self.max_iters)?, self.seed, self.workers as u32, &mut raw)
        })?;
        Ok(Index { raw })
    }
}

/// `SearchResult` (`src/ivf/search.rs:41-45`).
#[derive(Debug, Clone, PartialEq)]
pub struct SearchResult {
    pub row_idx: u32,
    pub distance: f32,
}

/// An index bound to its column on one GPU.  What `topk()` re-creates per query from the file
/// (`read_index_from_parquet` + `read_embeddings_for_rows`, `src/ivf/search.rs:89-110`) is kept resident here:
/// cache one per indexed Parquet file.
pub struct Searcher<'c> {
    raw: *mut sys::PqvSearcher,
    _corpus: std::marker::PhantomData<&'c Corpus>,
}
unsafe impl Send for Searcher<'_> {}
unsafe impl Sync for Searcher<'_> {}

impl<'c> Searcher<'c> {
    pub fn new(index: &Index, corpus: &'c mut Corpus) -> Result<Self> {
        let mut raw = ptr::null_mut();
        check(unsafe { sys::pqv_searcher_create(index.raw, corpus.raw, sys::PQV_LAYOUT_IVF_ORDERED, &mut raw) })?;
        Ok(Self { raw, _corpus: std::marker::PhantomData })
    }

    /// `IvfIndex::candidate_rows` (`src/ivf/index.rs:57-63`): probe-rank major, ascending ids inside.
    pub fn candidate_rows(&self, query: &[f32], nprobe: NonZeroUsize) -> Result<Vec<u32>> {
        let (mut rows, mut n) = (ptr::null_mut(), 0u64);
        check(unsafe {
            sys::pqv_candidate_rows(self.raw, query.as_ptr(), query.len() as u32, nprobe.get() as u32, &mut rows, &mut n)
        })?;
        let out = if rows.is_null() || n == 0 { Vec::new() } else { unsafe { std::slice::from_raw_parts(rows, n as usize) }.to_vec() };
        unsafe { sys::pqv_rows_free(rows) };
        Ok(out)
    }

    /// `topk()` (`src/ivf/search.rs:83-142`) for a batch of queries (`queries.len() == nq * dim`): identical
    /// results to the reference, ties included (flagged queries are replayed through the reference's heap).
    pub fn topk(&self, queries: &[f32], dim: usize, k: NonZeroUsize, nprobe: NonZeroUsize) -> Result<Vec<Vec<SearchResult>>> {
        let nq = if dim == 0 { 0 } else { queries.len() / dim };
        let (k, np) = (k.get(), nprobe.get());
        let mut rows = vec![0u32; nq * k];
        let mut dist = vec![0f32; nq * k];
        let mut found = vec![0u32; nq];
        check(unsafe {
            sys::pqv_topk(self.raw, queries.as_ptr(), nq as u32, dim as u32, k as u32, np as u32, 0, sys::PQV_L2SQ_REF4, 1,
                          rows.as_mut_ptr(), dist.as_mut_ptr(), found.as_mut_ptr(), ptr::null_mut())
        })?;
        Ok((0..nq)
            .map(|q| (0..found[q] as usize).map(|i| SearchResult { row_idx: rows[q * k + i], distance: dist[q * k + i] }).collect())
            .collect())
    }

    /// Plan metrics (`src/df_vector/index_exec.rs:289-299`, `exec.rs:411-427`).
    pub fn counters(&self) -> Result<sys::PqvCounters> {
        let mut c = sys::PqvCounters::default();
        check(unsafe { sys::pqv_counters(self.raw, &mut c) })?;
        Ok(c)
    }
}

impl Drop for Searcher<'_> {
    fn drop(&mut self) {
        unsafe { sys::pqv_searcher_free(self.raw) }
    }
}

/// `TopkBuilder` (`src/ivf/search.rs:49-81`) over a cached [`Searcher`] instead of a path.
pub struct TopkBuilder<'a, 'c> {
    searcher: &'a Searcher<'c>,
    query: &'a [f32],
    k: Option<NonZeroUsize>,
    nprobe: Option<NonZeroUsize>,
}

impl<'a, 'c> TopkBuilder<'a, 'c> {
    pub fn new(searcher: &'a Searcher<'c>, query: &'a [f32]) -> Self {
        Self { searcher, query, k: None, nprobe: None }
    }
    pub fn k(mut self, k: usize) -> Result<Self> {
        self.k = Some(NonZeroUsize::new(k).ok_or("k must be > 0")?); // search.rs:67
        Ok(self)
    }
    pub fn nprobe(mut self, nprobe: usize) -> Result<Self> {
        self.nprobe = Some(NonZeroUsize::new(nprobe).ok_or("nprobe must be > 0")?); // search.rs:72
        Ok(self)
    }
    /// Blocking (the reference's `async fn search` runs its CPU loop inline as well, `search.rs:112-127`).
    pub fn search(self) -> Result<Vec<SearchResult>> {
        let k = self.k.ok_or("k must be set")?; // search.rs:77
        let nprobe = self.nprobe.ok_or("nprobe must be set")?; // search.rs:78
        Ok(self.searcher.topk(self.query, self.query.len(), k, nprobe)?.pop().unwrap_or_default())
    }
}

/// The running top-k of `VectorTopKExec::topk_from_batches` (`src/df_vector/exec.rs:257-277`): fold one
/// `RecordBatch` at a time, materialise `ScalarValue`s only for the <= k survivors at the end.  `rows` / `d2` hold
/// the reference `BinaryHeap`'s backing array between batches (the library replays `push` / `peek` / `pop` exactly,
/// so ties come out as in the reference).
pub struct RerankState {
    device: usize,
    k: usize,
    rows: Vec<u32>,
    d2: Vec<f32>,
    count: u32,
}

impl RerankState {
    pub fn new(device: usize, k: usize) -> Self {
        Self { device, k, rows: vec![0; k.max(1)], d2: vec![0.0; k.max(1)], count: 0 }
    }

    fn check_lens(m: usize, ids: Option<&[u32]>, valid: Option<&[u8]>) -> Result<()> {
        if ids.map_or(false, |v| v.len() != m) { return Err("ids length must equal the number of rows".into()); }
        if valid.map_or(false, |v| v.len() != m) { return Err("valid length must equal the number of rows".into()); }
        Ok(())
    }

    /// `values`: the batch's list-values buffer `[m, dim]`; `ids`: the payload per row (e.g. batch-global row
    /// numbers; `None` = 0..m); `valid`: 0 for null rows / wrong-length lists (`exec.rs:496-498,526-528`).
    /// Distances use the element-by-element order of `compute_distance_values` (`exec.rs:529-533`).
    pub fn fold_batch(&mut self, query: &[f32], values: &[f32], ids: Option<&[u32]>, valid: Option<&[u8]>) -> Result<()> {
        let dim = query.len();
        let m = if dim == 0 { 0 } else { values.len() / dim };
        Self::check_lens(m, ids, valid)?;
        check(unsafe {
            sys::pqv_rerank(self.device as c_int, query.as_ptr(), values.as_ptr(), ids.map_or(ptr::null(), |v| v.as_ptr()),
                            valid.map_or(ptr::null(), |v| v.as_ptr()), m as u64, dim as u32, self.k as u32,
                            sys::PQV_L2SQ_SEQ, self.rows.as_mut_ptr(), self.d2.as_mut_ptr(), &mut self.count)
        })
    }

    /// The same for a `Float64Array` values buffer: each value is narrowed `as f32` first (`exec.rs:538-545`).
    pub fn fold_batch_f64(&mut self, query: &[f32], values: &[f64], ids: Option<&[u32]>, valid: Option<&[u8]>) -> Result<()> {
        let dim = query.len();
        let m = if dim == 0 { 0 } else { values.len() / dim };
        Self::check_lens(m, ids, valid)?;
        check(unsafe {
            sys::pqv_rerank_f64(self.device as c_int, query.as_ptr(), values.as_ptr(), ids.map_or(ptr::null(), |v| v.as_ptr()),
                                valid.map_or(ptr::null(), |v| v.as_ptr()), m as u64, dim as u32, self.k as u32,
                                sys::PQV_L2SQ_SEQ, self.rows.as_mut_ptr(), self.d2.as_mut_ptr(), &mut self.count)
        })
    }

    /// (payload, squared distance) in output order: `heap.into_iter()` + the stable sort of `exec.rs:269-274`
    /// (no distance column, no sqrt on this path).
    pub fn finish(self) -> Result<Vec<(u32, f32)>> {
        let n = self.count as usize;
        let mut rows = vec![0u32; n.max(1)];
        let mut d2 = vec![0f32; n.max(1)];
        check(unsafe { sys::pqv_rerank_finish(self.rows.as_ptr(), self.d2.as_ptr(), self.count, rows.as_mut_ptr(), d2.as_mut_ptr()) })?;
        Ok((0..n).map(|i| (rows[i], d2[i])).collect())
    }
}

/// One rank of the sharded search (`src/df_vector/index_exec.rs:85-164` probes every file's own index;
/// `src/df_vector/exec.rs:264-267` merges them in one heap): one process per GPU, each holding a row range and its
/// index; `exchange` is ONE RCCL all-gather of the per-shard top-k + the deterministic merge, no torch involved.
pub struct ShardComm {
    raw: *mut sys::PqvShardComm,
}
unsafe impl Send for ShardComm {}

impl ShardComm {
    /// Rank 0 draws the rendezvous id and distributes the 128 bytes over the host's own channel.
    pub fn unique_id() -> Result<[u8; 128]> {
        let mut id = [0u8; 128];
        check(unsafe { sys::pqv_shard_unique_id(id.as_mut_ptr()) })?;
        Ok(id)
    }
    /// Collective over all ranks.
    pub fn new(device: usize, rank: u32, world: u32, id: &[u8; 128]) -> Result<Self> {
        let mut raw = ptr::null_mut();
        check(unsafe { sys::pqv_shard_comm_create(device as c_int, rank, world, id.as_ptr(), &mut raw) })?;
        Ok(Self { raw })
    }
    pub fn rank(&self) -> u32 { unsafe { sys::pqv_shard_comm_rank(self.raw) } }
    pub fn world(&self) -> u32 { unsafe { sys::pqv_shard_comm_world(self.raw) } }
    /// Device pointers: this rank's `pqv_topk_device` outputs `[nq, k]`, the shards' first global rows `i64[world]`,
    /// outputs `f32 / i64 [nq, k]` -- identical on every rank.  Asynchronous on `hip_stream`.
    ///
    /// # Safety
    /// All pointers must be valid device allocations of the stated sizes on this communicator's GPU.
    pub unsafe fn exchange(&mut self, d_dist: *const c_void, d_rows: *const c_void, d_row_base: *const c_void, nq: u32, k: u32,
                           d_out_dist: *mut c_void, d_out_rows: *mut c_void, hip_stream: *mut c_void) -> Result<()> {
        check(sys::pqv_shard_exchange(self.raw, d_dist, d_rows, d_row_base, nq, k, d_out_dist, d_out_rows, hip_stream))
    }
}

impl Drop for ShardComm {
    fn drop(&mut self) {
        unsafe { sys::pqv_shard_comm_free(self.raw) }
    }
}

/// `CandidateCursor` (`src/df_vector/access.rs:193-243`).
pub struct CandidateCursor {
    raw: *mut sys::PqvCandidateCursor,
    files: usize,
}

impl CandidateCursor {
    pub fn new(file_count: usize) -> Result<Self> {
        let mut raw = ptr::null_mut();
        check(unsafe { sys::pqv_candidate_cursor_new(file_count as u32, &mut raw) })?;
        Ok(Self { raw, files: file_count })
    }
    pub fn add_candidates(&mut self, idx: usize, candidates: &[u32]) -> Result<()> {
        check(unsafe { sys::pqv_candidate_cursor_add(self.raw, idx as u32, candidates.as_ptr(), candidates.len() as u64) })
    }
    pub fn next_batch(&mut self, batch_size: usize) -> Result<Vec<(usize, u32)>> {
        let mut files = vec![0u32; batch_size.max(1)];
        let mut rows = vec![0u32; batch_size.max(1)];
        let mut n = 0u64;
        let mut taken = vec![0u64; self.files.max(1)];
        check(unsafe {
            sys::pqv_candidate_cursor_next_batch(self.raw, batch_size as u64, files.as_mut_ptr(), rows.as_mut_ptr(), &mut n, taken.as_mut_ptr())
        })?;
        Ok((0..n as usize).map(|i| (files[i] as usize, rows[i])).collect())
    }
}

impl Drop for CandidateCursor {
    fn drop(&mut self) {
        unsafe { sys::pqv_candidate_cursor_free(self.raw) }
    }
}
