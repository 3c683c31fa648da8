/*
 * pqv_oracle.h -- CPU ORACLE (TEST INFRASTRUCTURE, NOT PRODUCT CODE).
 *
 * A plain-C restatement of pq-vector's IVF build + top-k hot path, following the
 * reference line by line.  Only tests/, __graft_entry__.smoke() and bench.py's
 * cpu_baseline leg may load this library; the product (libpqv_hip.so) never does.
 *
 * Reference files restated (paths relative to the reference repo):
 *   src/ivf/index.rs   :57-63 (candidate_rows) :65-128 (blob) :130-149 (probe)
 *                      :152-214 (build_ivf_index) :222-257 (sample, nearest_centroid)
 *                      :259-320 (worker chunking) :323-457 (k_means) :461-480 (squared_l2)
 *   src/ivf/search.rs  :12-38 (HeapItem order) :83-142 (topk)
 *   src/df_vector/exec.rs :429-550 (TopKRow order, update_topk_heap, compute_distance_values)
 *   src/df_vector/access.rs :193-243 (CandidateCursor)
 *   src/ivf/mod.rs     :16-101 (validation texts)
 *
 * PINNING STATUS
 *   pinned by the reference's own known answers (tests/test_oracle_golden.py):
 *     d2([1,2,3],[4,5,6]) = 27 (index.rs:488-493); blob round trip + hand-derived 60-byte
 *     image (index.rs:496-511); SQL fixtures ids [5,2] and [3,4] and counters
 *     candidate_rows=6 / embeddings_fetched=4,3 (df_vector/tests.rs:31-39,99,166-174,235 and
 *     snapshots); in-place build dim=2 (parquet.rs:652-659).
 *   PARITY UNPINNED: every seeded choice.  The RNG is third-party (`rand 0.8.5`,
 *     `rand_chacha 0.3.1`, `rand_core 0.6.4`; Cargo.lock:2233-2275), absent from the
 *     reference tree, and no reference test fixes a seeded result.  The published
 *     algorithms (ChaCha12 block function, PCG32 seed expansion, widening-multiply range
 *     sampling, index::sample's floyd / inplace / rejection) are restated here and the
 *     ChaCha core is checked against the public zero-key ChaCha20/ChaCha12 vectors, but
 *     which rows the real crate would sample for a given seed cannot be confirmed in this
 *     environment (no Rust toolchain).  Rust std's BinaryHeap / stable sort tie behaviour
 *     is likewise restated from the published std source, pinned by no reference test.
 */
#ifndef PQV_ORACLE_H
#define PQV_ORACLE_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* ---- distances ------------------------------------------------------------------- */
/* src/ivf/index.rs:461-480: 4-wide grouping, sum += ((d0^2+d1^2)+d2^2)+d3^2, scalar tail. */
float pqo_squared_l2_ref4(const float *a, const float *b, size_t len);
/* src/df_vector/exec.rs:529-533: plain sequential dist += (v-q)^2. */
float pqo_squared_l2_seq(const float *values, const float *query, size_t len);
/* src/df_vector/exec.rs:538-545: Float64 column values narrowed to f32 first. */
float pqo_squared_l2_seq_f64(const double *values, const float *query, size_t len);

/* ---- rand 0.8.5 restatement ---------------------------------------- */
typedef struct pqo_rng {
    uint32_t key[8];
    uint64_t counter;   /* next block counter */
    uint32_t buf[64];   /* 4 consecutive ChaCha12 blocks */
    uint32_t index;     /* next unread word; 64 == empty */
} pqo_rng;

void     pqo_chacha_block(const uint32_t key[8], uint64_t counter, uint64_t stream,
                          int rounds, uint32_t out[16]);
void     pqo_rng_from_seed(pqo_rng *rng, const uint8_t seed[32]);
void     pqo_rng_seed_from_u64(pqo_rng *rng, uint64_t state);      /* rand_core 0.6.4 */
uint32_t pqo_rng_next_u32(pqo_rng *rng);
uint64_t pqo_rng_next_u64(pqo_rng *rng);
uint64_t pqo_rng_gen_range_usize(pqo_rng *rng, uint64_t low, uint64_t high);       /* low..high  */
uint32_t pqo_rng_gen_range_u32_incl(pqo_rng *rng, uint32_t low, uint32_t high);    /* low..=high */
float    pqo_rng_gen_range_f32_unit(pqo_rng *rng);                                 /* 0.0..1.0   */
float    pqo_rng_gen_f32(pqo_rng *rng);                                            /* gen::<f32> */
/* rand::seq::index::sample(rng, length, amount); out has room for `amount` entries.
 * *branch (optional) receives 0 floyd, 1 inplace, 2 rejection. Returns 0 / -1. */
int      pqo_index_sample(pqo_rng *rng, uint64_t length, uint64_t amount, uint64_t *out,
                          int *branch);

/* ---- IVF index --------------------------------------------------------------------- */
typedef struct pqo_index {
    uint32_t  dim;
    uint32_t  n_clusters;
    float    *centroids;   /* [n_clusters * dim] */
    uint64_t *list_off;    /* [n_clusters + 1]   */
    uint32_t *list_rows;   /* [list_off[n_clusters]] inverted lists, concatenated */
} pqo_index;

void pqo_index_free(pqo_index *idx);

/* src/ivf/index.rs:152-214.  n_clusters == 0 => ceil(sqrt(n)).  workers == 0 => online
 * CPUs; `workers` fixes the f32 partial-sum chunking of k-means++.
 * On error returns negative and copies the reference's message into err (>= 128 bytes). */
int pqo_build_ivf_index(const float *data, uint64_t n, uint32_t dim, uint32_t n_clusters,
                        uint32_t max_iters, uint64_t seed, uint32_t workers,
                        pqo_index **out, char *err);

/* src/ivf/index.rs:323-457.  centroids [k*dim] out, assignments [n] out (u64 = usize).
 * iters_run (optional) = Lloyd iterations whose assign step ran. */
int pqo_kmeans(const float *data, uint64_t n, uint32_t dim, uint32_t k, uint32_t max_iters,
               uint64_t seed, uint32_t workers, float *centroids, uint64_t *assignments,
               uint32_t *iters_run);

/* src/ivf/index.rs:65-83 / :85-128.  to_bytes mallocs *buf. */
int pqo_index_to_bytes(const pqo_index *idx, uint8_t **buf, size_t *len);
int pqo_index_from_bytes(const uint8_t *bytes, size_t len, pqo_index **out, char *err);

/* src/ivf/index.rs:130-149: all distances, stable sort, first min(nprobe,k_c).
 * out has room for min(nprobe, n_clusters) entries; returns the count written. */
uint32_t pqo_find_closest_centroids(const pqo_index *idx, const float *query,
                                    uint32_t nprobe, uint32_t *out);
/* src/ivf/index.rs:57-63: mallocs *rows (probe-rank major, ascending row ids inside). */
int pqo_candidate_rows(const pqo_index *idx, const float *query, uint32_t nprobe,
                       uint32_t **rows, uint64_t *n_rows);

/* ---- top-k ------------------------------------------------------------------------- */
/* src/ivf/search.rs:83-142 with the embedding column resident in memory (`embeddings`
 * is the full [n, dim] matrix in file row order; read_embeddings_for_rows :155-244 is a
 * gather).  Emulates std::collections::BinaryHeap exactly, then sqrt and
 * a stable sort by distance.  row_idx/dist have room for k; *n_found = results written.
 * Returns 0, or negative with the reference's message in err. */
int pqo_topk_ivf(const pqo_index *idx, const float *embeddings, const float *query,
                 uint32_t query_len, uint32_t k, uint32_t nprobe, uint32_t *row_idx,
                 float *dist, uint32_t *n_found, uint64_t *n_candidates, char *err);

/* src/df_vector/exec.rs:257-277,457-484: heap over the given rows in the given order
 * (the scan's output order, post-filter), plain sequential distance, rows sorted by d2,
 * no sqrt.  `rows` are indices into `embeddings`; out_rows/out_d2 have room for k. */
int pqo_topk_df(const float *embeddings, uint32_t dim, const uint32_t *rows, uint64_t n_rows,
                const float *query, uint32_t k, uint32_t *out_rows, float *out_d2,
                uint32_t *n_found);

/* src/df_vector/access.rs:193-243 CandidateCursor::next_batch for file_count lists.
 * cand[f] has cand_len[f] entries.  Writes up to batch_size (file,row) pairs. */
uint64_t pqo_candidate_cursor_take(const uint32_t *const *cand, const uint64_t *cand_len,
                                   uint32_t file_count, uint64_t batch_size,
                                   uint32_t *out_file, uint32_t *out_row);

/* Batched convenience for the CPU baseline: runs pqo_topk_ivf for nq queries on one
 * thread (faithful to search.rs:115).  Outputs [nq*k]. */
int pqo_topk_ivf_batch(const pqo_index *idx, const float *embeddings, const float *queries,
                       uint32_t nq, uint32_t k, uint32_t nprobe, uint32_t *row_idx,
                       float *dist, uint32_t *n_found, uint64_t *n_candidates);

#ifdef __cplusplus
}
#endif
#endif
