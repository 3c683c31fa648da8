/*
 * pqv.h -- C ABI of the MI355X-native pq-vector hot path (libpqv_hip.so).
 *
 * This is the drop-in boundary.  The reference (pure Rust, no FFI of its own) would bind
 * these symbols from a `extern "C"` block where its L1 "IVF core" meets its callers; each
 * entry point cites the reference interface it replaces (paths relative to the reference
 * repo).  INTEGRATION.md shows the Rust-side binding.
 *
 * Conventions
 *   - plain pointers and sizes only; no C++/torch types.
 *   - return 0 (PQV_OK) or a negative pqv_status; the message for the calling thread is
 *     returned by pqv_last_error().  Validation messages are the reference's own strings
 *     (src/ivf/index.rs:24,89,158,169; src/ivf/mod.rs:59,86; src/ivf/parquet.rs:90,93;
 *     src/ivf/search.rs:67,72,92-97) so a Rust shim can surface identical errors.
 *   - inputs are borrowed for the duration of the call (mirrors `&[f32]`); outputs are
 *     either caller-allocated fixed-size arrays or library buffers released with the
 *     matching pqv_*_free.  Opaque handles own host + device memory.
 *   - every entry point is blocking and callable from any thread.  A handle may be shared
 *     across threads; calls on one pqv_searcher serialise on its internal scratch.
 *   - there is NO CPU fallback: every compute entry point fails with PQV_ERR_NO_DEVICE
 *     when no gfx950 device is usable.
 *   - row ids are file-global u32 row ordinals, as in the reference (src/ivf/index.rs:13).
 */
#ifndef PQV_H
#define PQV_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef enum pqv_status {
    PQV_OK              = 0,
    PQV_ERR_INVALID     = -1,  /* argument validation (reference's message texts)      */
    PQV_ERR_NO_DEVICE   = -2,  /* no usable HIP device / device index out of range      */
    PQV_ERR_HIP         = -3,  /* a HIP runtime call failed                              */
    PQV_ERR_OOM         = -4,  /* host or device allocation failed                       */
    PQV_ERR_UNSUPPORTED = -5,  /* outside the implemented envelope (e.g. k > 1024)       */
    PQV_ERR_FORMAT      = -6   /* malformed index blob                                   */
} pqv_status;

/* Summation order of the squared-L2 distance. */
typedef enum pqv_metric {
    PQV_L2SQ_REF4 = 0, /* src/ivf/index.rs:461-480: sum += ((d0^2+d1^2)+d2^2)+d3^2 per 4 */
    PQV_L2SQ_SEQ  = 1  /* src/df_vector/exec.rs:529-533: dist += d^2, element by element */
} pqv_metric;

/* Metrics of pqv_brute_topk (an EXTENSION: the reference has neither cosine distance nor a
 * batched brute-force path -- SURVEY F5; BASELINE.json configs[4] asks for it).  Every returned
 * distance comes from an f32 dot product (f32 matrix cores for the first rows, an exact f32
 * re-scoring behind the int8 / f16 matrix-core screen of the rest: the screen is a rigorous
 * bound, it never drops a row of the result), so distances agree with an f64 oracle to ~1e-6
 * relative (well inside the north star's 1e-4), not bit for bit.  The first call builds the
 * screen's image of the corpus (+ 1 byte per value; PQV_BRUTE_OP=f16: + 2). */
#define PQV_COSINE      2   /* 1 - q.v / (|q| |v|); zero-norm vectors get distance 1            */
#define PQV_L2SQ_MFMA   3   /* |q|^2 + |v|^2 - 2 q.v (norm-expansion form, clamped at 0)        */

/* pqv_searcher_create flags */
#define PQV_LAYOUT_IVF_ORDERED   0x0u /* copy rows into cluster-contiguous order in HBM (default) */
#define PQV_LAYOUT_ROW_ORDER     0x1u /* keep file row order; re-rank gathers rows by id          */
#define PQV_RELEASE_ROW_ORDER    0x2u /* with IVF_ORDERED: let the corpus drop its row-order copy */
#define PQV_RELEASE_IF_COPIED    0x4u /* with IVF_ORDERED: drop the row-order copy only where the searcher made a list-ordered
                                         f32 copy of its own (the images-only layout keeps reading the caller's rows): one f32
                                         copy of the column resident either way                                              */

typedef struct pqv_index    pqv_index;    /* IvfIndex: dim, n_clusters, centroids, inverted lists */
typedef struct pqv_corpus   pqv_corpus;   /* the embedding column, resident in one GPU's HBM     */
typedef struct pqv_searcher pqv_searcher; /* (index, corpus) bound for querying on one GPU       */

/* Thread-local message of the last failing call on this thread ("" if none). */
const char *pqv_last_error(void);
/* Number of usable HIP devices (0 on a machine without a GPU; never an error). */
int pqv_device_count(void);
/* Library ABI version (major*100 + minor). */
int pqv_abi_version(void);

/* ---- embedding column -> HBM ---------------------------------------------------------
 * Replaces the materialised `Embeddings{data: Vec<f32>, dim}` that
 * read_parquet_with_embeddings hands to build_ivf_index (src/ivf/parquet.rs:216-305,
 * src/ivf/mod.rs:72-102) and the per-query read_embeddings_for_rows gather
 * (src/ivf/search.rs:155-244): the column is uploaded once and stays resident. */

/* rows: host [n, dim] row-major f32.  n may be 0 (build then fails like the reference). */
int pqv_corpus_upload(int device, const float *rows, uint64_t n, uint32_t dim,
                      pqv_corpus **out);
/* Append-style upload for streaming row groups: create empty with capacity, then append. */
int pqv_corpus_create(int device, uint64_t capacity_rows, uint32_t dim, pqv_corpus **out);
int pqv_corpus_append(pqv_corpus *corpus, const float *rows, uint64_t n_rows);
/* Same, narrowing a Float64 column to f32 first (src/ivf/parquet.rs:246-256). */
int pqv_corpus_append_f64(pqv_corpus *corpus, const double *rows, uint64_t n_rows);
/* Streaming upload for a loader that decodes row groups on several threads (N1; src/ivf/parquet.rs:216-305 materialises the
 * column batch by batch, :262-286): rows [row_offset, row_offset + n_rows) of a corpus made by pqv_corpus_create are written from
 * `rows`.  The call copies them into one of the corpus' PINNED staging buffers and enqueues the DMA (hipMemcpyAsync); it returns
 * as soon as the caller's buffer may be reused, so decoding the next batch overlaps the upload of this one.  Batches may arrive
 * in any order and from several threads.  _f64 narrows `as f32` on the device (:246-256).  pqv_corpus_finish waits for every
 * DMA and sets the row count to n_rows (<= capacity); until then the corpus must not be used by anything else. */
int pqv_corpus_write_rows(pqv_corpus *corpus, uint64_t row_offset, const float *rows, uint64_t n_rows);
int pqv_corpus_write_rows_f64(pqv_corpus *corpus, uint64_t row_offset, const double *rows, uint64_t n_rows);
int pqv_corpus_finish(pqv_corpus *corpus, uint64_t n_rows);
/* Host-side helpers of the page-level Parquet reader (N1: parquet_io.py walks the data pages of the embedding leaf itself where
 * it can -- uncompressed or codec-decompressed v1 pages, PLAIN or dictionary-encoded -- instead of materialising Arrow lists):
 *   pqv_parquet_levels_check  decodes an RLE / bit-packed hybrid level run (Parquet "RLE" encoding without the length prefix)
 *                             of n_values levels and checks it WITHOUT storing it: mode 0 -- every level == expect (definition
 *                             levels of a column without nulls); mode 1 -- repetition levels of equal-length lists: level 0 at
 *                             every multiple of `expect` (the list length), 1 elsewhere, the page starting a row.
 *                             0 = as expected, 1 = something else (the caller falls back to the Arrow reader and its messages),
 *                             PQV_ERR_INVALID = malformed run.
 *   pqv_parquet_dict_decode   hybrid-encoded dictionary indices (bit width in the first byte, as in a RLE_DICTIONARY data page)
 *                             -> out[i] = dict[index_i], elem_size 4 or 8 bytes per value.
 * No device is touched. */
int pqv_parquet_levels_check(const uint8_t *buf, uint64_t len, uint32_t bit_width, uint64_t n_values, int mode, uint64_t expect,
                             uint64_t *period_out /* mode 1 with expect == 0: the list length is DISCOVERED (the position of the
                                                     second level 0, or n_values if there is none), written here, then checked */);
int pqv_parquet_dict_decode(const uint8_t *buf, uint64_t len, const void *dict, uint64_t dict_n, uint32_t elem_size,
                            uint64_t n_values, void *out);
/* The page headers of one column chunk, walked from its first byte (Thrift compact PageHeader, parquet.thrift): per page 8
 * ints {type, header bytes, compressed_page_size, uncompressed_page_size, num_values, encoding, definition_level_encoding,
 * repetition_level_encoding}, -1 where the header has no such field.  Stops at len, at max_pages, or once stop_values (> 0)
 * values have been seen in data pages.  (The reference reads pages through the parquet crate: src/ivf/parquet.rs:216-230.) */
/* A run of uncompressed PLAIN v1 data pages of the embedding leaf, from the mapped file to the corpus: per page body (at
 * file_base + body_off[i], body_len[i] bytes) the repetition levels must start a row exactly every `dim` values and every
 * definition level must be max_def (src/ivf/parquet.rs:231-280's checks, on the levels); the n_values[i] values behind the level
 * runs go to rows first_value[i] / dim .. through the pinned staging buffers like pqv_corpus_write_rows (f64 != 0: Float64
 * values, narrowed on the device).  Returns 0, a negative error, or 1 with *bad_page set when page i is not such a page. */
int pqv_corpus_write_plain_pages(pqv_corpus *corpus, const uint8_t *file_base, const uint64_t *body_off, const uint32_t *body_len,
                                 const uint64_t *first_value, const uint32_t *n_values, uint32_t n_pages, uint32_t dim,
                                 uint32_t max_def, int f64, uint32_t *bad_page);
int pqv_parquet_page_headers(const uint8_t *buf, uint64_t len, uint64_t stop_values, uint32_t max_pages, int32_t *out, uint32_t *n_pages);
/* Adopt an existing device buffer [n, dim] f32 on `device` (borrowed; caller keeps it
 * alive and frees it). */
int pqv_corpus_from_device(int device, const void *d_rows, uint64_t n, uint32_t dim,
                           pqv_corpus **out);
uint64_t pqv_corpus_rows(const pqv_corpus *corpus);
uint32_t pqv_corpus_dim(const pqv_corpus *corpus);
int      pqv_corpus_device(const pqv_corpus *corpus);
/* Gather rows (file row ordinals) back to the host: out [m, dim]. */
int pqv_corpus_fetch_rows(const pqv_corpus *corpus, const uint32_t *rows, uint64_t m,
                          float *out);
void pqv_corpus_free(pqv_corpus *corpus);

/* ---- index build ---------------------------------------------------------------------
 * Replaces build_ivf_index(&Embeddings, IvfBuildConfig{n_clusters, max_iters, seed})
 * (src/ivf/index.rs:152-214) including k_means (:323-457), sample_embeddings (:222-242)
 * and the final assignment (:189-206).
 *   n_clusters == 0  => ceil(sqrt(n))                     (:161-167)
 *   workers          => the `available_parallelism()` the reference would see; it fixes the
 *                       f32 partial-sum chunking of k-means++ (:259-265,:356-370).
 *                       0 => this host's online CPU count. */
int pqv_index_build(const pqv_corpus *corpus, uint32_t n_clusters, uint32_t max_iters,
                    uint64_t seed, uint32_t workers, pqv_index **out);
/* Phase wall times of the calling thread's last pqv_index_build / pqv_index_build_host / pqv_kmeans (bench records):
 * out[0] k-means++ seconds, [1] Lloyd seconds, [2] Lloyd iterations run, [3] final assignment seconds (device work +
 * download), [4] host inverted-list build seconds, [5] 1 if the final assignment ran through the MFMA screen, [6] same for
 * the Lloyd assignments, [7] sample rows, [8] summed HIP-event seconds of the assign_wide_kernel launches of the final
 * assignment (0 where another form ran), [9] their count; entries beyond n are not written. */
int pqv_index_build_stats(double *out, uint32_t n);
/* ONE k-means++ pick (src/ivf/index.rs:354-390) over given minima, taken the way the build's device rounds take it (kernels_kpp.hip):
 * *total = the `workers` chunks' sequential f32 sums joined in ascending order (:356-370), *pick = the first slot whose sequential f32
 * cumulative sum reaches draw * total (:373-383).  A test entry point: *status 0 = decided; otherwise the reason the build hands the
 * round to its host walk (1 total not positive or not finite, 2 a value that is not a finite non-negative number, 3 no slot reaches the
 * threshold) and *pick is not written.  n in [1, 57344]; workers 0 => this host's online CPU count. */
int pqv_kpp_pick(int device, const float *minima, uint32_t n, uint32_t workers, float draw, uint64_t *pick, float *total,
                 uint32_t *status);
/* Host-pointer form with the reference's exact argument shape: uploads, builds, frees. */
int pqv_index_build_host(int device, const float *data, uint64_t data_len, uint32_t dim,
                         uint32_t n_clusters, uint32_t max_iters, uint64_t seed,
                         uint32_t workers, pqv_index **out);
/* k_means alone (src/ivf/index.rs:323-457) over a resident matrix; centroids [k*dim] and
 * assignments [n] are host outputs; iters_run may be NULL. */
int pqv_kmeans(const pqv_corpus *sample, uint32_t k, uint32_t max_iters, uint64_t seed,
               uint32_t workers, float *centroids, uint32_t *assignments,
               uint32_t *iters_run);

/* ---- index blob ----------------------------------------------------------------------
 * IvfIndex::to_bytes / from_bytes (src/ivf/index.rs:65-128); byte-identical layout. */
int  pqv_index_from_bytes(const uint8_t *bytes, size_t len, pqv_index **out);
int  pqv_index_to_bytes(const pqv_index *index, uint8_t **buf, size_t *len);
void pqv_bytes_free(uint8_t *buf);
/* Assemble from parts (what IvfIndex{..} literal construction does in index.rs:497-502). */
int  pqv_index_from_parts(uint32_t dim, uint32_t n_clusters, const float *centroids,
                          const uint64_t *list_off /*[n_clusters+1]*/,
                          const uint32_t *list_rows, pqv_index **out);
uint32_t pqv_index_dim(const pqv_index *index);              /* IvfIndex::dim() :53 */
uint32_t pqv_index_n_clusters(const pqv_index *index);
uint64_t pqv_index_n_rows(const pqv_index *index);           /* sum of list lengths   */
const float    *pqv_index_centroids(const pqv_index *index); /* [n_clusters*dim]      */
const uint64_t *pqv_index_list_offsets(const pqv_index *index); /* [n_clusters+1]     */
const uint32_t *pqv_index_list_rows(const pqv_index *index); /* lists, concatenated   */
void pqv_index_free(pqv_index *index);

/* ---- searching -----------------------------------------------------------------------
 * Binds an index to the resident column on the corpus' GPU: uploads centroids + lists and
 * (by default) lays the rows out cluster-contiguously so that a probed inverted list is
 * one sequential HBM range. */
int  pqv_searcher_create(const pqv_index *index, pqv_corpus *corpus, uint32_t flags,
                         pqv_searcher **out);
void pqv_searcher_free(pqv_searcher *searcher);

/* Tunables of one searcher (all optional; the defaults are the measured dispatch rules of DESIGN.md 5 / DESIGN_HISTORY.md 5.1c-f).
 * The reference has no such knobs -- its topk() is one fixed loop (src/ivf/search.rs:112-127) -- so nothing here
 * changes results, only which kernels produce them; tests use it to force every path through the same oracle.
 *   "rerank_mode"   0 by rule, 1 streaming kernel, 2 batched tile path
 *   "tile_filter"   MFMA lower-bound screen in the batched path: 0 off, 1 by rule, 2 forced
 *   "filter_variant" 1 = one 16-query group per block instead of the wide kernel
 *   "cand_cap"      candidate-buffer entries per query of the wide screened path (0 = by rule: 2048, 8192 for k > 32)
 *   "screen_f16"    f16 screen operands where the data allows (default 1)
 *   "seed_rows", "wide_rows", "tile_rows"   rows sampled for thresholds / per block (0 = by rule)
 *   "running_thr"   running thresholds of the wide kernel (default 1)
 *   "defer"         k > 64 on the wide screened path: survivors of the screen are appended with the distance bounds their
 *                   screen score gives and evaluated after the filter, only where the k-th smallest upper bound leaves them
 *                   (default 1; 0 = every survivor is evaluated by the streaming wave)
 *   "quad_xcd"      quad-to-XCD affinity of the wide kernels (-1 by rule)
 *   "wide_waves"    waves per block of the wide kernel: 0 by rule, 4 or 8
 *   "quad_width"    queries per quad of the wide kernel (0 by rule; a multiple of 32)
 *   "screen_i8"     int8 screen operands for rows of a multiple of 256 dims (default 1)
 *   "min_blocks"    workgroups the wide kernel's rows-per-block rule aims for on small batches (0 by rule)
 *   "pair_prune"    int8 path: skip (query, list) pairs whose centre-distance bound already exceeds the query's threshold
 *                   (default 1)
 *   "i8_form"       int8 images: 0 by rule -- the per-list residual (one query image per probed pair) where the lists' own
 *                   scales are >= 1.3x the scale one centre for the whole corpus would get, else the one-centre form (one
 *                   image per query) --, 1 one centre, 2 residual; the int8 copy is rebuilt on the next call
 *   "single_bucket" a single-query call is bucketed by the probe merge itself (two launches less).  1 (default) and 3: the
 *                   probe joins that launch too wherever there are at most 4096 centroids (probe_single_kernel); 2: the
 *                   bucketing only, the probe keeps its own launch; 0: the general three-launch pair sort
 *   "seed_refine"   exact distances of the rows behind the k selected seed bounds replace the k-th bound as the first
 *                   threshold (k <= 16; default 1)
 *   "item_grid"     wide kernel grid: 1 = one workgroup per (quad, existing row chunk) for the 4-wave blocks (default),
 *                   2 = for the 8-wave blocks too, 0 = (chunks of the longest list) x quads
 *   "chunk_major"   order of those work items: 1 (default) = row chunk 0 of every quad, then chunk 1, ... -- a query's
 *                   thresholds have seen a piece of each of its lists before the bulk is screened; 0 = list by list
 *   "probe_rows"    batched centroid probe (a lane per centroid): 1 for batches of >= 8 queries (default), 2 always,
 *                   0 = the per-query stream over the centroid table
 *   "wide_quads"    int8 path, batches: lists probed by 97..160 queries in ONE quad of the wide-quad instance (32-row tiles, one
 *                   8-wave block per CU) instead of two regular quads.  1 (default): by the PREVIOUS batch's shape -- regular quads
 *                   only while most rows of the popular lists sit in lists of > 160 pairs (clustered query loads); 2 always; 0 never
 *   "wide_quad_rows" rows per block of that instance (0 by rule: 8192; 2048 for the list form below)
 *   "list_once"     int8 path, batches, dims 256 / 512 / 768, no deferred evaluation: 1 = lists probed by more than 96 queries run in
 *                   list_filter_kernel -- 32 rows per wave stationary in registers, ALL the list's pairs (quads of up to 1024) streamed
 *                   past them: every such list is read once.  0 (default): measured slower than the wide-quad instance
 *   "xcd_items"     a level's work items are filled column by column of an 8-column layout, so that the quads of one list run
 *                   back to back on one XCD and share its rows through that L2: 1 (default) = in the table that has several
 *                   quads per list, 3 = in both tables, 0 = off
 *   "fork_wide"     the wide-quad launch on a side stream of the call's lane: 0 off (default), 1 regular instance first, 2 wide first
 *   "pf96"          f16 96-query form on rows of <= 128 dims: all of the next tile's operands are requested behind the current tile's
 *                   MFMAs (1 = for lists of <= 2048 rows on average (default), 2 always, 0 never)
 *   "drain_min"     queued survivors that start a batch of exact evaluations before a wave's last tile (0 = 64)
 * The same names, upper-cased with a PQV_ prefix, are read from the environment ONCE when a searcher is created
 * (profiling scripts). */
int pqv_searcher_set_option(pqv_searcher *searcher, const char *name, int64_t value);
/* Which kernels a pqv_topk_device call of this shape would run on this searcher (one line of text, for bench
 * records): written NUL-terminated into buf (truncated to len). */
int pqv_searcher_describe(const pqv_searcher *searcher, uint32_t nq, uint32_t k, uint32_t nprobe, int metric,
                          char *buf, size_t len);
/* Device memory held for this searcher, in bytes: the corpus' row-order copy (0 once released), the IVF-ordered
 * f32 rows, the blocked screen-operand copy, and everything else (centroids, lists, norms, scratch lanes). */
int pqv_searcher_footprint(const pqv_searcher *searcher, uint64_t *row_order_bytes, uint64_t *ivf_rows_bytes,
                           uint64_t *blocked_bytes, uint64_t *other_bytes);

/* IvfIndex::find_closest_centroids (src/ivf/index.rs:130-149): clusters_out has room for
 * min(nprobe, n_clusters); *n_out receives the count. */
int pqv_probe(const pqv_searcher *searcher, const float *query, uint32_t query_len,
              uint32_t nprobe, uint32_t *clusters_out, uint32_t *n_out);
/* IvfIndex::candidate_rows (src/ivf/index.rs:57-63): library buffer, probe-rank major. */
int  pqv_candidate_rows(const pqv_searcher *searcher, const float *query, uint32_t query_len,
                        uint32_t nprobe, uint32_t **rows, uint64_t *n_rows);
void pqv_rows_free(uint32_t *rows);

/* topk() (src/ivf/search.rs:83-142) for nq queries at once: probe, re-rank every
 * candidate, keep the k smallest by (d2, candidate position), order ascending.
 *   queries      host [nq, query_len]; query_len must equal the index dim (:91-98)
 *   max_candidates  0 => none; else the CandidateCursor cap (src/df_vector/access.rs:
 *                   214-242, single file): only the first max_candidates candidates in
 *                   probe-rank order are considered
 *   metric       PQV_L2SQ_REF4 (TopkBuilder) or PQV_L2SQ_SEQ (VectorTopKExec)
 *   sqrt_out     nonzero => dist = sqrt(d2) as TopkBuilder returns (:133); 0 => d2
 *   row_idx/dist host [nq*k]; entries past n_found[q] are 0xFFFFFFFF / +inf
 * Ties: when two of a query's k results (or the k-th and the runner-up) have EQUAL output
 * distance, which rows survive and in what order is an artefact of Rust's BinaryHeap sift
 * history (search.rs:113-140).  pqv_topk detects that on the device and replays exactly those
 * queries through the same heap mechanics on the host (distances still computed on the GPU),
 * so its results equal the reference's in every non-NaN case.  pqv_topk_device never leaves
 * the GPU: it returns the k smallest by (d2, candidate position), identical to the reference
 * whenever no such tie exists.  (k == 1024, the largest supported, has no runner-up slot: a tie between the
 * 1024th result and the first excluded candidate is then not detected.)
 *   n_found      host [nq] (may be NULL)
 *   n_candidates host [nq] (may be NULL): sum of the probed lists' lengths, before the cap */
int pqv_topk(const pqv_searcher *searcher, const float *queries, uint32_t nq,
             uint32_t query_len, uint32_t k, uint32_t nprobe, uint64_t max_candidates,
             int metric, int sqrt_out, uint32_t *row_idx, float *dist, uint32_t *n_found,
             uint64_t *n_candidates);

/* Device-resident form: queries and outputs are device pointers on the searcher's GPU,
 * work is enqueued on `hip_stream` (a hipStream_t passed as void*; NULL = the searcher's
 * own NON-BLOCKING stream -- NOT HIP's legacy default stream, whose handle is also 0: a caller
 * whose next operation runs on the default stream (torch.cuda.current_stream() outside a stream
 * context) must pass an explicit stream -- hipStreamLegacy, ((hipStream_t)1), names the default stream itself -- or
 * synchronise) and the call returns without
 * synchronising.  d_n_found / d_n_candidates may be NULL.  This is what bench.py times. */
int pqv_topk_device(const pqv_searcher *searcher, const void *d_queries, uint32_t nq,
                    uint32_t k, uint32_t nprobe, uint64_t max_candidates, int metric,
                    int sqrt_out, void *d_row_idx, void *d_dist, void *d_n_found,
                    void *d_n_candidates, void *hip_stream);

/* The same, plus d_tie_flags (device u32 [nq]): 1 for every query two of whose k results (or the k-th and the
 * runner-up) have EQUAL output distance -- exactly the queries for which the reference's survivors / order depend on
 * BinaryHeap history (src/ivf/search.rs:113-140) and pqv_topk_device's (d2, position) order may differ from it.
 * Everything stays asynchronous; a caller that needs the reference's answer under ties re-submits the flagged
 * queries to pqv_topk, which replays the heap exactly.  (The kernels then work with k + 1 entries: k <= 1023.) */
int pqv_topk_device_flags(const pqv_searcher *searcher, const void *d_queries, uint32_t nq, uint32_t k,
                          uint32_t nprobe, uint64_t max_candidates, int metric, int sqrt_out, void *d_row_idx,
                          void *d_dist, void *d_n_found, void *d_n_candidates, void *d_tie_flags, void *hip_stream);

/* Exhaustive top-k of nq queries over EVERY row of the resident column (no index), batched
 * on the matrix cores: what DataFusion's brute-force `ORDER BY array_distance(..) LIMIT k`
 * baseline does row by row (benches/query.rs:76-98), for the metrics above.  Results are
 * ordered ascending by (distance, row id).  Host arrays as in pqv_topk. */
int pqv_brute_topk(const pqv_corpus *corpus, const float *queries, uint32_t nq, uint32_t query_len,
                   uint32_t k, int metric, uint32_t *row_idx, float *dist, uint32_t *n_found);

/* update_topk_heap / compute_distance_values (src/df_vector/exec.rs:457-550) for one RecordBatch worth of rows:
 * cand host [m, dim] values buffer, ids[m] the payload to return (e.g. batch row numbers; NULL => 0..m), valid[m]
 * optional bytes (0 = null row or length mismatch => skipped, exec.rs:496-498,526-528).  The batch's distances are
 * computed on the GPU in compute_distance_values' order; the rows then pass through the reference's heap policy --
 * std's BinaryHeap push / peek / pop, restated exactly -- in arrival order, so results equal the reference's in every
 * non-NaN case, ties included.  The running state io_rows / io_d2 / io_count (caller-owned, capacity k) IS that heap's
 * backing array between batches (NOT sorted); pass *io_count = 0 for the first batch and call pqv_rerank_finish after
 * the last.  Any k > 0 (the selection is the heap's; no kernel-side list).
 * pqv_rerank_f64: the same for a Float64 values buffer, each value narrowed `as f32` first (exec.rs:538-545).
 * pqv_rerank_finish: heap.into_iter() + the stable sort by distance of exec.rs:269-274 -> out_rows / out_d2 [count]
 * ascending (host-only; the DataFusion path emits no distance column and takes no sqrt). */
int pqv_rerank(int device, const float *query, const float *cand, const uint32_t *ids,
               const uint8_t *valid, uint64_t m, uint32_t dim, uint32_t k, int metric,
               uint32_t *io_rows, float *io_d2, uint32_t *io_count);
int pqv_rerank_f64(int device, const float *query, const double *cand, const uint32_t *ids,
                   const uint8_t *valid, uint64_t m, uint32_t dim, uint32_t k, int metric,
                   uint32_t *io_rows, float *io_d2, uint32_t *io_count);
int pqv_rerank_finish(const uint32_t *io_rows, const float *io_d2, uint32_t count, uint32_t *out_rows, float *out_d2);

/* The same fold for a batch that is ALREADY resident on `device` (e.g. a decoded Arrow values buffer uploaded by the
 * scan): d_cand [m, dim] f32, d_ids u32[m] or NULL (=> 0..m), the running state d_io_rows u32[k] / d_io_d2 f32[k] /
 * d_io_count u32[1] lives on the device between batches, kept SORTED ascending by (d2, arrival) -- a different state
 * form from pqv_rerank's heap array; do not mix the two on one state (a device count above k is read as k).  No null
 * mask (compact before the call).  k <= 1024.  Enqueued on hip_stream (NULL: a library stream) and completed before
 * the call returns.  Everything stays on the GPU, so the heap's sift history is not replayed: the result is the k
 * smallest by (d2, arrival), equal to the reference's whenever no two of the k results (nor the k-th and the first
 * excluded row) have the same distance.
 * pqv_rerank_device_flags also maintains d_tie_flag (device u32[1], zeroed by the caller before the first batch,
 * sticky): set when such a tie was seen in any fold so far -- then, and only then, the reference's survivors / order
 * may differ and the caller re-runs those batches through pqv_rerank (k <= 1023: the lists carry a runner-up). */
int pqv_rerank_device(int device, const void *d_query, const void *d_cand, const void *d_ids, uint64_t m, uint32_t dim,
                      uint32_t k, int metric, void *d_io_rows, void *d_io_d2, void *d_io_count, void *hip_stream);
int pqv_rerank_device_flags(int device, const void *d_query, const void *d_cand, const void *d_ids, uint64_t m, uint32_t dim,
                            uint32_t k, int metric, void *d_io_rows, void *d_io_d2, void *d_io_count, void *d_tie_flag,
                            void *hip_stream);

/* Merge per-shard top-k lists (one list per file/shard, as topk_from_batches does for
 * multi-file tables, src/df_vector/exec.rs:264-267): lists [n_lists, nq, k] of d2 (or
 * distance) and row ids, counts [n_lists, nq]; ties resolve by (value, list index,
 * position in list).  Host arrays. out_list receives the source list of each result. */
int pqv_merge_topk(const float *dist, const uint32_t *rows, const uint32_t *counts,
                   uint32_t n_lists, uint32_t nq, uint32_t k, float *out_dist,
                   uint32_t *out_rows, uint32_t *out_list, uint32_t *out_count);

/* Device-resident form of the merge for the multi-GPU exchange: d_dist f32 / d_rows u32
 * [n_lists, nq, k] as delivered by one all-gather of every rank's pqv_topk_device outputs
 * (empty slots: 0xFFFFFFFF rows), d_row_base i64[n_lists] the first global row of each shard.
 * Writes d_out_dist f32[nq, k] and d_out_rows i64[nq, k] (global row ids, -1 = none), ordered
 * by (distance, list, position).  Asynchronous on hip_stream (NULL = the default stream). */
int pqv_merge_topk_device(int device, const void *d_dist, const void *d_rows, const void *d_row_base,
                          uint32_t n_lists, uint32_t nq, uint32_t k, void *d_out_dist,
                          void *d_out_rows, void *hip_stream);

/* The same merge over PACKED lists: d_pairs [n_lists, nq, k] of {f32 distance, u32 row} (8 bytes per result), the
 * form in which ONE all-gather delivers every rank's top-k (pq_vector_amd/sharding.py). */
int pqv_merge_topk_packed_device(int device, const void *d_pairs, const void *d_row_base, uint32_t n_lists,
                                 uint32_t nq, uint32_t k, void *d_out_dist, void *d_out_rows, void *hip_stream);

/* ---- the exchange itself, without torch --------------------------------------------------------------------------
 * Multi-file tables are probed file by file and merged in ONE heap (src/df_vector/index_exec.rs:85-164,
 * src/df_vector/exec.rs:264-267).  With one shard (file / row-group range) per GPU that heap is one RCCL all-gather of
 * every rank's k packed {distance, row} results per query over xGMI followed by the merge above on every rank.  These
 * entry points let the Rust host do that with nothing but this library: librccl is resolved at run time (a copy the
 * process already loaded -- same SONAME -- else the loader path, else /opt/rocm/lib; PQV_RCCL_LIB overrides) and is
 * only needed by these calls.
 *   unique_id     rank 0 draws the 128-byte rendezvous id (ncclGetUniqueId) and hands it to every rank over the host's
 *                 own channel (a file, a socket, the DataFusion coordinator)
 *   comm_create   collective over all `world` ranks (ncclCommInitRank), one process per GPU
 *   comm_adopt    wraps an existing ncclComm_t (passed as void*) of THE SAME librccl (pqv_shard_rccl_path says which
 *                 one this library bound); the caller keeps ownership of it
 *   exchange      d_dist f32 / d_rows u32 [nq, k]: this rank's pqv_topk_device outputs (empty slots 0xFFFFFFFF);
 *                 d_row_base i64[world]: first global row of every shard; writes d_out_dist f32 / d_out_rows i64 [nq, k]
 *                 ordered by (distance, shard, position) -- identical on every rank.  Asynchronous on hip_stream; calls
 *                 on one communicator must not overlap (one stream, or the caller's own ordering). */
#define PQV_SHARD_ID_BYTES 128
typedef struct pqv_shard_comm pqv_shard_comm;
const char *pqv_shard_rccl_path(void);
int  pqv_shard_unique_id(uint8_t *id /* [PQV_SHARD_ID_BYTES] */);
int  pqv_shard_comm_create(int device, uint32_t rank, uint32_t world, const uint8_t *id, pqv_shard_comm **out);
int  pqv_shard_comm_adopt(int device, void *nccl_comm, pqv_shard_comm **out);
/* ONE Parquet file shared by `world` GPUs (host only; no device is touched): the half-open row-group range [*rg_lo, *rg_hi) of
 * shard `rank`, the file-global row id of its first row (*row_base = rows of the row groups before it: how the reference maps
 * file-global row ids to row groups, src/df_vector/access.rs:128-144) and its row count.  rg_rows[i] = rows of row group i in
 * file order.  Cut r (between shards r - 1 and r) is the row-group boundary whose prefix sum is nearest to r n / world -- the lower
 * one on a tie, never before the previous cut; every rank computes every cut from the footer alone, the ranges tile the file, a row
 * group is never split, and with fewer row groups than shards the surplus shards are empty (*n_rows = 0). */
int  pqv_shard_row_groups(const uint64_t *rg_rows, uint32_t n_row_groups, uint32_t rank, uint32_t world,
                          uint32_t *rg_lo, uint32_t *rg_hi, uint64_t *row_base, uint64_t *n_rows);
uint32_t pqv_shard_comm_rank(const pqv_shard_comm *comm);
uint32_t pqv_shard_comm_world(const pqv_shard_comm *comm);
int  pqv_shard_exchange(pqv_shard_comm *comm, const void *d_dist, const void *d_rows, const void *d_row_base,
                        uint32_t nq, uint32_t k, void *d_out_dist, void *d_out_rows, void *hip_stream);
void pqv_shard_comm_free(pqv_shard_comm *comm);

/* CandidateCursor (src/df_vector/access.rs:193-243; used at src/df_vector/exec.rs:224-231): when a query over a
 * multi-file table carries max_candidates, candidates are taken round-robin across the files -- one per file and
 * turn, each file's own list in probe-rank order -- until the cap; the round-robin position persists between
 * batches.  Host-side integer logic (no device needed).  A file's share of a batch is always the next PREFIX of its
 * list, so per_file_taken (optional, [file_count], cumulative) is what to pass as pqv_topk's max_candidates for
 * that file's searcher.
 *   add:        replaces file idx's candidate list (rows are copied); idx >= file_count is ignored like the reference
 *   next_batch: out_file / out_row have room for batch_size entries; *n_out receives the count */
typedef struct pqv_candidate_cursor pqv_candidate_cursor;
int  pqv_candidate_cursor_new(uint32_t file_count, pqv_candidate_cursor **out);
int  pqv_candidate_cursor_add(pqv_candidate_cursor *cursor, uint32_t idx, const uint32_t *rows, uint64_t n_rows);
int  pqv_candidate_cursor_next_batch(pqv_candidate_cursor *cursor, uint64_t batch_size, uint32_t *out_file,
                                     uint32_t *out_row, uint64_t *n_out, uint64_t *per_file_taken);
void pqv_candidate_cursor_free(pqv_candidate_cursor *cursor);

/* Counters mirroring the reference's plan metrics (src/df_vector/index_exec.rs:289-299,
 * src/df_vector/exec.rs:411-427), accumulated per searcher since creation. */
typedef struct pqv_counters_t {
    uint64_t queries;            /* top-k queries served                                 */
    uint64_t candidate_rows;     /* sum over queries of probed list lengths              */
    uint64_t embeddings_fetched; /* rows whose distance was computed (after the cap)     */
    uint64_t kernel_launches;    /* device kernels enqueued                              */
    uint64_t exact_replays;      /* pqv_topk queries replayed through the exact heap     */
    uint64_t screened_pairs;     /* (row, query) pairs put through the MFMA lower-bound   */
    uint64_t screen_survivors;   /* ... of which were evaluated exactly                   */
} pqv_counters_t;
int pqv_counters(const pqv_searcher *searcher, pqv_counters_t *out);

/* Diagnostic: the library's own restatement of rand 0.8.5's StdRng (ChaCha12) and seq::index::sample, which draw
 * every seeded choice of the index build (src/ivf/index.rs:231-232,327,337,340,373,385).  Exposed so that tests can
 * pin it to rand's published value-stability vectors and cross-check it against the CPU oracle's independent C
 * restatement.  seed32 != NULL: StdRng::from_seed(seed32), else StdRng::seed_from_u64(seed64).
 *   mode 0: out[i] = next_u64()            mode 1: out[i] = next_u32()
 *   mode 2: out[i] = gen_range(0..arg) usize                mode 3: out[i] = bits of gen_range(0.0f32..1.0)
 *   mode 4: out[0..n) = index::sample(rng, length = arg, amount = n) */
int pqv_diag_rng(const uint8_t *seed32, uint64_t seed64, int mode, uint64_t arg, uint64_t *out, uint64_t n);

/* Kernel timing for bench.py's roofline line.  While enabled, every pqv_topk /
 * pqv_topk_device call records HIP events on the stream its kernels run on: around the
 * whole call and around the re-rank kernel alone.  pqv_timing_read synchronises those
 * events, returns the summed milliseconds and the number of calls since the last read,
 * and clears them. */
int pqv_set_timing(pqv_searcher *searcher, int enabled);
int pqv_timing_read(const pqv_searcher *searcher, double *rerank_ms, double *total_ms,
                    uint32_t *n_calls);

#ifdef __cplusplus
}
#endif
#endif /* PQV_H */
