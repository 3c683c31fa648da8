// kernels.h -- host-callable launchers of the gfx950 kernels (internal; the public
// boundary is include/pqv.h).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace pqv {

// Sentinel key: sorts after every real (d2, position) key.
constexpr uint64_t KEY_EMPTY = ~0ull;

// What a streaming distance pass does with each row's exact d2.
enum StreamMode : int {
    STREAM_TOPK   = 0,  // fold into per-wave top-k lists
    STREAM_DIST   = 1,  // out_f32[pos] = d2
    STREAM_MINUPD = 2,  // out_f32[pos] = min(out_f32[pos], d2)   (k-means++ round)
};

struct StreamArgs {
    // row storage [*, dim] f32; a list position p maps to storage row (row_of ? row_of[p] : p)
    const float    *mat;
    const uint32_t *row_of;
    // inverted-list offsets in list positions; probe == nullptr => a single list
    // [single_begin, single_end) is scanned for every query
    const uint64_t *list_off;
    const uint32_t *probe;       // [nq, nprobe] cluster ids in probe-rank order
    const uint64_t *cand_base;   // [nq, nprobe] candidate position of each probed list's first row
    uint64_t        single_begin, single_end;
    const float    *queries;     // [nq, dim]
    uint32_t        nq, nprobe, dim, k;
    uint32_t        rows_per_block;   // multiple of 256
    uint32_t        blocks_per_list;  // gridDim.x
    uint64_t        max_pos;          // candidates at position >= max_pos are ignored
    int             metric;           // pqv_metric
    // TOPK outputs: [nq][nprobe*blocks_per_list*4][k]
    uint64_t       *part_keys;
    uint32_t       *part_vals;
    // DIST / MINUPD output, indexed by candidate position (single-list mode)
    float          *out_f32;
    // MINUPD, optional: every value this pass writes to out_f32 is also written here (pinned host memory that starts as a
    // copy of out_f32: the k-means++ host scan reads it after the stream is drained, with no copy command between the
    // rounds' kernels, and only the minima that changed cross the bus)
    float          *mirror_f32;
    // MINUPD, optional second mirror in CHUNK-TRANSPOSED order: entry pos goes to mirror_t[(pos % mirror_chunk) * mirror_stride +
    // pos / mirror_chunk], so the host adds the w chunk chains of index.rs:356-370 with vector instructions (lane = chunk)
    float          *mirror_t;
    uint32_t        mirror_chunk, mirror_stride;
    // optional: zero_u32[0 .. zero_n) = 0 (scratch of the kernels that follow in the stream)
    uint32_t       *zero_u32;
    uint32_t        zero_n;
};

// Streaming exact-order squared-L2 pass (the re-rank kernel).  Returns hipError_t.
hipError_t launch_stream(const StreamArgs &a, StreamMode mode, hipStream_t s);

// Batched centroid probe (probe_rows_kernel): every query against every centroid, exact reference order.
struct ProbeRowsArgs {
    const float4   *cent_t;      // [dim/4][kc_pad] float4 transpose of the centroid table (launch_transpose_rows4)
    const float    *queries;     // [nq, dim]
    uint32_t        nq, kc, kc_pad, dim;
    uint64_t       *part_keys;   // [nq][4 * ceil(kc/256)][64], unsorted
    uint32_t       *part_vals;
    uint32_t       *zero_u32;    // optional scratch to zero (as StreamArgs::zero_u32)
    uint32_t        zero_n;
};
hipError_t launch_probe_rows(const ProbeRowsArgs &a, hipStream_t s);
struct MergeArgs;
// one query: probe + probe merge in one launch (kc_pad <= 4096, nprobe <= 64); the last block to finish merges
// (pr.part_keys: kc_pad keys of scratch; ticket: one zero-initialised u32 that the kernel leaves at zero)
struct PairQuantArgs;
hipError_t launch_probe_single(const ProbeRowsArgs &pr, const MergeArgs &a, uint32_t *ticket, const PairQuantArgs *quant, hipStream_t s);
hipError_t launch_transpose_rows4(const float *rows, uint32_t kc, uint32_t kc_pad, uint32_t dim, void *out, hipStream_t s);

// int8 images of (query, probed list) pairs (kernels_probe.hip: quantize_pair_i8_wave)
struct PairQuantArgs {
    const float    *queries;     // [nq, dim]
    const uint32_t *probe;       // [nq * nprobe] cluster of pair p
    const float    *center;      // [n_clusters, dim]
    const float    *scale;       // [n_clusters]
    const float    *half;        // [n_clusters] largest |x - centre| component of the list
    const float    *radius;      // [n_clusters] upper bound of |x - centre| over the list's rows
    uint32_t        n_pairs, nprobe, dim;
    int8_t         *q_i8;        // [n_pairs, dim]
    int            *q_n2i;       // [n_pairs]
    float          *q_res, *q_resu, *pair_lb;
};

struct MergeArgs {
    const uint64_t *part_keys;   // [nq][n_part][k_part]
    const uint32_t *part_vals;
    uint32_t        nq, n_part, k_part, k;
    // final mode outputs
    const uint32_t *ids;         // storage row -> file row id (nullptr => identity)
    uint32_t       *row_idx;     // [nq, k]
    float          *dist;        // [nq, k]
    uint32_t       *n_found;     // [nq] or nullptr
    int             sqrt_out;
    uint32_t        k_out;       // results written per query (<= k); 0 => k
    // optional extra source (wide screened path): per-query candidate buffers; with `spilled` the partial
    // lists of a query are read only if spilled[q] != 0
    const uint64_t *cand_keys;   // [nq][cand_cap] or nullptr
    const uint32_t *cand_vals;
    const uint32_t *cand_cnt;    // [nq]
    uint32_t        cand_cap;
    const uint32_t *spilled;     // [nq] or nullptr
    // deferred exact evaluation (TileArgs::cand_lb): the merge first RESOLVES the query's buffer -- T = the k-th smallest upper
    // bound; entries with lower bound <= T are evaluated in the reference's order (index.rs:461-480) against `mat`, the others
    // dropped -- and then merges exact keys as before.  256 threads per query.
    float          *cand_lb;     // [nq][cand_cap] or nullptr
    uint64_t       *cand_keys_rw;
    const float    *mat;         // row-major f32 rows, `dim` values each; cand_vals are row numbers in it
    unsigned long long *resolve_stats;   // optional: the statistics block (slot q % STATS_SLOTS, word 1 += exact evaluations)
    uint32_t       *zero_after;          // final merge: this word is cleared (launch_resolve's work counter, ready for the next call:
                                         // no memset in front of every resolve)
    int             resolve_two_cuts;    // resolve_select_kernel evaluates the entries that define the first cut itself and cuts again (batches:
                                         // the chip is full of such blocks; for a few queries one block per query would be the whole latency)
    uint32_t       *tie_flag;    // [nq] or nullptr: 1 iff two of the first k_out+1 merged entries
                                 // (k_out entries + the runner-up) have equal OUTPUT distance --
                                 // then the reference's order/survivors depend on heap history
    // probe mode outputs (k == nprobe)
    const uint64_t *list_off;
    uint32_t       *probe;       // [nq, k]
    uint64_t       *cand_base;   // [nq, k]
    uint64_t       *n_cand;      // [nq] or nullptr
    uint64_t        max_pos;     // cap applied to n_cand
    // probe mode, optional: the searcher's statistics block -- every query adds its candidate count to the
    // candidate_rows / embeddings_fetched counters (slot q % STATS_SLOTS, words 2 and 3: the plan metrics of
    // src/df_vector/index_exec.rs:289-299 and exec.rs:411-427, kept on the device so that the asynchronous
    // pqv_topk_device path counts too)
    unsigned long long *stats;
    // probe mode extras for the batched path (all optional): cluster histogram of the (query, probe
    // rank) pairs (zeroed beforehand), reset of the per-query admission thresholds, |q|^2
    // one query, nprobe <= 64 (probe mode): the wave also writes the bucketing itself -- every probed cluster is one quad
    // of one pair, in probe-rank order -- and launch_pair_sort is skipped (two launches less per single-query call)
    uint4          *sq_quads;     // nullptr = off
    uint32_t       *sq_pairs, *sq_n_quads;
    uint32_t       *sq_item_quad, *sq_n_items;     // optional work-item table (wide_filter_kernel)
    uint32_t        sq_item_rows, sq_max_items;
    uint32_t       *hist;
    uint32_t        hist_stride;  // > 0: HIST_REPLICAS copies hist[r * hist_stride + c], query q adds to copy q % HIST_REPLICAS
    // probe mode, optional: the query's per-wave partial lists of the re-rank start EMPTY (all-ones keys / values);
    // preset_n entries per query, written by the query's wave (saves a launch per step)
    uint64_t       *preset_keys;
    uint32_t       *preset_vals;
    uint32_t        preset_n;
    // ... or, for the wide screened path (whose waves touch their lists only when a candidate buffer overflows), one
    // "written" byte per list: probe mode clears preset_flag_n bytes per query, wide_filter_kernel sets a list's byte
    // with its first fold (which then writes the whole list), and the final merge reads only flagged lists -- 1 byte per
    // list per step instead of 12 k
    uint8_t        *preset_flags;
    uint32_t        preset_flag_n;
    const uint8_t  *part_flags;  // final mode: [nq][n_part] or nullptr (every list is valid)
    unsigned long long *gthr_init;
    float          *qnorm_out;
    float          *qmax_out;    // max |q_i| per query (f16 screen)
    const float    *queries;
    uint32_t        dim;
};
hipError_t launch_merge_final(const MergeArgs &a, hipStream_t s);
// deferred evaluations of a BATCH, ahead of launch_merge_final (which then gets cand_lb = nullptr): work = nq * cand_cap uint2 of scratch
hipError_t launch_resolve(const MergeArgs &a, void *work, uint32_t *n_work, bool n_work_is_zero, hipStream_t s);
hipError_t launch_merge_probe(const MergeArgs &a, hipStream_t s);

// ---- batched re-rank: cluster-major tiles ------------------------------------------------
// The (query, probe-rank) pairs of a batch are bucketed by cluster; a "group" is up to
// TILE_QB pairs of one cluster.  tile_rerank streams each row tile of that cluster ONCE for
// the whole group (queries as wave-uniform scalar operands), instead of once per query.
constexpr int TILE_QB = 16;

constexpr uint32_t HIST_REPLICAS = 16;
// statistics counters live in STATS_SLOTS copies one cache line apart: stats[8 + 16 i + {0, 1}] screened pairs /
// survivors (wide_filter_kernel), + {2, 3} candidate rows / embeddings fetched (probe merge)
constexpr uint32_t STATS_SLOTS = 64;

constexpr uint32_t ITEM_LEVELS = 64;
struct PairSortArgs {
    const uint32_t *probe;     // [nq * nprobe] cluster of pair p = q*nprobe + j
    uint32_t        n_pairs, n_clusters;
    int             hist_done; // 1: hist was already filled (launch_merge_probe with MergeArgs::hist)
    uint32_t       *hist;      // [n_clusters]      (zeroed by the caller)
    uint32_t        hist_stride; // > 0: the histogram arrives as HIST_REPLICAS partial copies, stride apart; the scan sums
                                 //      them into copy 0 (8 k atomics on the four lines of one copy serialise: 20 -> 9 us on C2)
    uint32_t       *cursor;    // [n_clusters] zeroed by the caller; with hist_stride > 0: [HIST_REPLICAS][hist_stride], set by the
                               // scan to each copy's first index in its cluster's bucket (the scatter's atomics spread likewise)
    uint32_t        nprobe;    // pairs per query (pair p belongs to query p / nprobe: selects the copy)
    uint32_t       *pair_off;  // [n_clusters + 1]
    uint32_t       *group_off; // [n_clusters + 1]
    uint32_t       *pairs;     // [n_pairs] pair ids bucketed by cluster
    uint4          *groups;    // [max_groups] {cluster, first slot, count, 0}
    uint32_t       *n_groups;  // [1]
    // quads: up to quad_width (32 or 64) pairs of one cluster (the unit of wide_filter_kernel)
    uint32_t        quad_width;
    uint32_t       *quad_off;  // [n_clusters + 1]
    uint4          *quads;     // [max_quads] {cluster, first slot, count, 0}
    uint32_t       *n_quads;   // [1]
    // optional work-item table of the wide filter kernel: one item per (quad, row chunk that exists in the quad's
    // list), quads in order, so that the kernel's 1-D grid holds no empty workgroups between real ones
    const uint64_t *list_off;  // [n_clusters + 1]
    uint32_t        item_rows; // rows per chunk (the kernel's rows_per_block); 0 = no table
    uint32_t       *item_off;  // [n_clusters + 1] first item of each cluster's first quad
    uint32_t       *n_items;   // [1]
    uint32_t       *item_quad; // [max_items] quad of each item; quads[q].w = the quad's first item
    uint32_t        max_items;
    // chunk-major item order (item_chunk != nullptr): items are numbered level by level -- row chunk 0 of every quad, then
    // chunk 1, ... (levels >= ITEM_LEVELS - 1 share the last one) -- so a query's thresholds have seen the first chunk of
    // all its lists before any later chunk is screened.  pair_scan_kernel writes the tables (order inside a level: by cluster).
    uint32_t       *item_chunk;      // [max_items] row chunk of each item
    uint32_t       *wide_item_chunk; // [wide_max_items]
    uint32_t       *wide_stats;      // optional [2]: items of the wide table, and how many of them belong to lists with no other quad
    uint32_t       *shape_stats;     // optional [2] (pinned host memory): rows of the lists that more than shape_wide pairs of the batch
    uint32_t        shape_wide;      // probe, and of those that more than shape_narrow do -- how the next batch cuts its quads
    uint32_t        shape_narrow;    // (its own width: quad_width is the 160-wide cut while the wide-quad instance is active)
    uint32_t        wide_list_major; // the WIDE table keeps list-major order (a list's chunks are consecutive items: pair_scatter_kernel writes
                                     // them, wide_item_chunk stays unused) while the regular one is chunk-major -- list_filter_kernel maps
                                     // consecutive items to ONE XCD, whose L2 then holds the list's pair images for all of its chunks
    uint32_t        xcd_items;       // bit 0: the regular table, bit 1: the wide table -- a level's slots filled column by column of an
                                     // 8-column layout (the quads of one list -> the same XCD, back to back)
    // optional second class of quads (the wide-quad instance of the filter kernel): with wide_min > 0 the quads are cut
    // quad_width (160) pairs wide, and a quad of >= wide_min (97) pairs is WIDE -- its items (chunks of wide_item_rows rows)
    // go to a table of their own, quads[q].w = its first item THERE; the others (<= 96 pairs: one per list at most, the
    // last) stay in item_quad
    uint32_t        wide_min;
    uint32_t        wide_item_rows;
    uint32_t       *wide_item_off;  // [n_clusters + 1]
    uint32_t       *wide_n_items;   // [1]
    uint32_t       *wide_item_quad; // [wide_max_items]
    uint32_t        wide_max_items;
};
// hist -> scan -> scatter; three tiny launches
hipError_t launch_pair_sort(const PairSortArgs &a, hipStream_t s);

// Optional refinement of the seed threshold (k <= 16, dim % 32 == 0): the k sampled bounds a query selects belong to 4 k
// rows (a bound is a lane's minimum over the 4 sub-tile rows it saw); their EXACT reference distances are evaluated
// and the k-th smallest of those -- still k distinct candidates, so still an upper bound of the final k-th distance --
// replaces the k-th bound: the threshold loses the slack of the operand form (int8: ~1 % of d2) before the screen starts.
struct SeedRefine {
    const float    *mat;        // f32 rows [*, dim] (nullptr = no refinement): list position p is storage row row_of ? row_of[p] : p
    const uint32_t *row_of;
    const float    *queries;    // [nq, dim]
    const uint64_t *list_off;
    const uint32_t *probe;      // [nq, nprobe]
    const uint64_t *cand_base;  // [nq, nprobe]
    uint32_t        dim, nprobe, seed_sw, seed_rows;
    uint64_t        max_pos;
};
// One-query calls: wide_seed_kernel's LAST block to finish (a ticket counter) runs seed_select_kernel's body itself.
struct SeedTail {
    int                 enable;
    uint32_t            n_vals, k;
    unsigned long long *gthr;
    uint32_t           *cand_cnt, *spilled, *thr_hist;
    float4             *thr_bins;
    uint32_t           *ticket;      // zero-initialised, left at zero
    uint32_t            lds_floats;  // dynamic LDS of the launch, in floats (set by launch_wide_seed): the refinement's term table
    SeedRefine          rf;
};
struct TileArgs {
    const float    *mat;
    const uint32_t *row_of;
    const uint64_t *list_off;
    const float    *queries;
    const uint64_t *cand_base;   // [nq * nprobe]
    const uint32_t *pairs;
    const uint4    *groups;
    const uint32_t *n_groups;
    uint32_t        max_groups;  // gridDim.y
    const uint4    *quads;       // wide_filter_kernel: gridDim.y = max_quads
    const uint32_t *n_quads;
    uint32_t        max_quads, quad_width;
    uint32_t        block_waves; // wide_filter_kernel: waves per block, 4 (default when 0) or 8 (one block per CU sharing a staged quad of up to 128 queries)
    uint32_t        nq, nprobe, dim, k;
    uint32_t        rows_per_block, blocks_per_list;
    uint64_t        max_pos;
    unsigned long long *gthr;    // [nq] per-query global admission threshold, preset to KEY_EMPTY
    // row window of this launch inside every list: rows [row_offset, row_offset + gridDim.x *
    // rows_per_block).  Partial-list slots: a (query, probe rank j) pair owns slots_per_pair
    // lists; block bx of this launch writes slots slot_base + bx * 4 + wave (one list per wave)
    uint64_t        row_offset;
    uint64_t        row_end;     // 0 = none: rows at list offsets >= row_end belong to a later launch
    uint32_t        slots_per_pair, slot_base;
    uint32_t        n_part;      // partial lists per query in part_keys (>= nprobe * slots_per_pair)
    uint32_t        grid_x;      // gridDim.x of this launch
    int             filter_variant;   // launch_tile_filter: 0 = wide_filter_kernel (2 or 4 query groups per block,
                                      // queries staged in LDS), 1 = tile_filter_kernel (one group per block)
    // MFMA filter only: squared norms of the storage rows / of the queries
    const float4   *mat_blk;     // wide_filter_kernel: blocked copy of the IVF-ordered lists (launch_block_rows)
    const uint64_t *blk_off;     // [n_clusters + 1] first 16-row tile of every list in mat_blk
    // f16 operands (wide kernels, dim % 128 == 0): mat_blk is the launch_block_rows_f16 copy, the queries are
    // converted while staged; scale is a power of two, scale2 = scale^2
    int             f16;
    int             q32_lds;      // set by the launcher: the f16 kernel also stages the exact f32 queries in LDS
    float           scale, scale2;
    const float    *query_maxabs;   // [nq] max |q_i| (merge probe)
    // int8 operands (wide kernels, dim % 256 == 0): mat_blk is the launch_block_rows_i8 copy -- per list the residual
    // against the list's centre at the list's scale; the query side is one image per (query, probed list) PAIR
    // (launch_quantize_pairs_i8), indexed by pair = query * nprobe + probe rank
    int             i8;
    const int8_t   *q_i8;        // [nq * nprobe][dim]
    const int      *q_n2i;       // [nq * nprobe] |vi|^2
    const float    *q_res;       // [nq * nprobe] rounding residual of the clamped query residual (lower bounds; +inf: never skip)
    int             i8_pair_images;   // 1: images per pair (the residual form); 0: ONE image per query, indexed by query (every
                                      // list shares centre and scale: the tables hold the same values for all lists)
    const float    *q_resu;      // [nq * nprobe] ... plus what the clamp cut off (upper bounds: wide_seed_kernel)
    const float    *pair_lb;     // [nq * nprobe] or nullptr: lower bound of d2(query, any row of the pair's list)
    const float    *list_scale;  // [n_clusters] S_c
    const int      *row_n2i;     // per storage row: |xi|^2
    const float    *row_res;     // per storage row: upper bound of |x - c - xi / S|
    const float4   *q_blk;       // wide kernels without LDS staging (long rows): blocked queries per quad
                                 // (launch_pack_queries), [max_quads][quad_width / 16][dim / 4][16] float4
    const float    *row_norm2;   // indexed like mat rows (norm_by_pos: by LIST POSITION -- images in list order over row-order f32 rows)
    int             norm_by_pos;
    const float    *query_norm2; // [nq]
    int             xcd_swizzle; // 1: XCD-aware workgroup remap (speed only)
    const uint32_t *item_quad;   // wide_filter_kernel: work-item table (PairSortArgs::item_quad); nullptr = 2-D grid
    const uint32_t *n_items;
    uint32_t        max_items;
    // second launch for the WIDE quads (PairSortArgs::wide_*): 8-wave blocks, quads of wide_width queries on 32-row tiles
    uint32_t        wide_width;  // 0 = none
    const uint32_t *wide_item_quad;
    const uint32_t *item_chunk, *wide_item_chunk;   // chunk-major tables (PairSortArgs::item_chunk); nullptr = chunk = item - quads[q].w
    uint32_t        wide_nt;         // the wide-quad instance streams its rows with the nt policy (most of its lists have one quad)
    uint32_t        list_once;       // the wide table's quads (up to wide_width <= 1024 pairs of one list) run in list_filter_kernel (kernels_list.hip)
    // host side only (launch_tile_filter): run the wide-quad instance on `side_stream`, forked from and joined to the call's stream
    // by the two events, so that the two instances' tails fill each other (they cannot share a CU -- 2 x 81 KB against 142 KB
    // of LDS -- but they can share the chip); NULL: one after the other on the call's stream
    uint32_t        opt_pf96;        // f16 96-query form on rows of <= 128 dims: whole-tile operand prefetch (PQV_PF96)
    uint32_t        drain_min;       // wide_filter_kernel: queue entries that start a batch of exact evaluations before the wave's last tile (0 = 64)
    hipStream_t     side_stream;
    hipEvent_t      ev_fork, ev_join;
    uint32_t        fork_wide_first; // the wide-quad launch is enqueued first, on the call's stream; the regular one on the side stream
    const uint32_t *wide_n_items;
    uint32_t        wide_max_items, wide_rows_per_block;
    // wide_filter_kernel: per-query append buffers of exact-verified candidates
    uint64_t       *cand_keys;   // [nq][cand_cap]
    uint32_t       *cand_vals;
    uint32_t       *cand_cnt;    // [nq] appended so far (may exceed cand_cap: the excess went to the wave lists)
    uint32_t        cand_cap;
    uint32_t       *spilled;     // [nq] set to 1 when a query overflowed its buffer
    // DEFERRED exact evaluation (optional): a survivor of the screen is appended with the BOUNDS its screen score gives --
    // key = (upper bound of the reference distance << 32 | position), cand_lb = lower bound -- instead of being evaluated by
    // the streaming wave; the final merge evaluates only the entries whose lower bound does not exceed the k-th smallest upper
    // bound of the query (MergeArgs::cand_lb).  cand_lb < 0 marks an entry whose key already holds the exact distance (pairs
    // without usable bounds: non-finite data, f16 range overflow, a tile with too many survivors for one expansion pass).
    float          *cand_lb;     // [nq][cand_cap] or nullptr (evaluate in the filter)
    uint32_t       *pendv;       // scratch [blocks][waves][wide_filter_pend()]: raw screen scores of the queued survivors
    uint32_t       *pendv_wide;  // ... of the wide-quad instance's launch
    // wide_seed_kernel: upper bounds [nq][nprobe][seed_sw][16], seed_sw = 4 * gridDim.x of the seed launch
    float          *seed_ub;
    SeedTail        seed_tail;   // wide_seed_kernel, one query: select + refine in the same launch
    uint32_t        seed_sw;
    // running thresholds (wide_filter_kernel, optional): per query two 64-bit words of 8-bit counters --
    // counter b = appended pairs at distance < thr0 - b w -- and the bin parameters {thr0, w, 1 / w, pad}
    // (seed_select_kernel); an appending lane tightens gthr[q] to the upper edge of the nearest bin whose
    // counter reached k
    uint32_t       *thr_hist;
    const float4   *thr_bins;
    unsigned long long *stats;   // [2] optional: += (row, query) pairs screened, += pairs evaluated exactly
    uint64_t       *part_keys;   // [nq][nprobe * slots_per_pair][k]
    uint32_t       *part_vals;
    uint8_t        *part_flags;  // wide_filter_kernel, optional: [nq][n_part] "this list has been written" (see MergeArgs)
};
// PQV_L2SQ_REF4 only, k <= 256.  The caller presets the whole partial-list buffer to EMPTY (0xFF bytes:
// KEY_EMPTY keys, 0xFFFFFFFF values); a wave touches its slots only when a candidate is admitted.
hipError_t launch_tile_rerank(const TileArgs &a, hipStream_t s);
// entries of a wave's survivor queue in wide_filter_kernel<NG = quad_width / 16, NW = waves> (TileArgs::pendv is sized with it)
constexpr uint32_t wide_filter_pend(uint32_t quad_width, uint32_t waves, bool f16) {
    return (waves == 8 ? (quad_width == 64 && f16 ? 512u : 256u) : quad_width == 96 ? 128u : 512u) + 64u;
}
// Same contract, but every (row, query) pair is first screened with an MFMA lower bound of its
// distance; only pairs that could still beat the query's admission threshold are evaluated in
// the reference's exact order.  Needs thresholds seeded by a prior launch_tile_rerank window.
hipError_t launch_tile_filter(const TileArgs &a, hipStream_t s);
// The row-stationary int8 screen for the lists that many queries of the batch probe (kernels_list.hip): `a` is the wide table's
// view of the step -- item_quad / item_chunk / n_items / max_items / rows_per_block of THAT table, quads of up to 1024 pairs.
hipError_t launch_list_filter(const TileArgs &a, hipStream_t s);
// After the exact seed window: gthr[q] = min(gthr[q], k-th smallest key over ALL of q's seed lists)
// (slots 0..3 of every (query, probe rank) pair) -- the k-th of the union, far tighter than the min
// of the per-wave k-th keys the fold publishes.
hipError_t launch_seed_threshold(const uint64_t *part_keys, uint32_t nq, uint32_t nprobe, uint32_t slots_per_pair,
                                 uint32_t k, unsigned long long *gthr, hipStream_t s);

// Thresholds of the wide screened pass from MFMA upper bounds (no exact seed pass): launch_wide_seed over
// the first rows of every list fills TileArgs::seed_ub, launch_seed_select turns a query's n_vals =
// nprobe * seed_sw * 16 minima into gthr[q] and resets its candidate buffer / overflow flag.
hipError_t launch_wide_seed(const TileArgs &a, hipStream_t s);
hipError_t launch_seed_select(const float *seed_ub, uint32_t nq, uint32_t n_vals, uint32_t k, unsigned long long *gthr,
                              uint32_t *cand_cnt, uint32_t *spilled, hipStream_t s,
                              uint32_t *thr_hist = nullptr, float4 *thr_bins = nullptr, const SeedRefine *refine = nullptr);

// ---- batched brute force as a dense Q.V^T contraction on f32 MFMA (BASELINE config 5) -------
// score s[i][j] = q_i . v_j over ALL rows j of a row range; distance by `metric`:
//   BRUTE_COSINE : d = 1 - s * rq[i] * rv[j]     (rq, rv = reciprocal L2 norms)
//   BRUTE_L2SQ   : d = max(0, qn[i] + vn[j] - 2 s)   (norm-expansion form: NOT the reference's
//                  summation order -- tolerance-level agreement only)
// Candidates whose key beats the query's current threshold are appended to a per-query buffer.
enum BruteMetric : int { BRUTE_COSINE = 0, BRUTE_L2SQ = 1 };

struct BruteArgs {
    const float *rows;        // [n, dim]
    const float *queries;     // [nq, dim]
    const float *row_aux;     // [n]  rv (cosine) or vn (l2)
    const float *query_aux;   // [nq] rq or qn
    uint64_t     row_begin, row_end;   // row range of this launch
    uint32_t     nq, dim;
    int          metric;
    const unsigned long long *thr;     // [nq] admission threshold (sortable key), KEY_EMPTY = none
    unsigned long long *cand;          // [nq][cap] appended keys
    uint32_t    *cand_cnt;             // [nq]
    uint32_t     cap;
    // dense != 0: no thresholds yet and empty buffers (the first row range): EVERY pair is kept, in slot row - row_begin of
    // its query -- no atomic per pair; the caller sets cand_cnt to the range's length (<= cap) afterwards
    uint32_t     dense;
};
hipError_t launch_brute_mfma(const BruteArgs &a, hipStream_t s);
// The f16 screen of the same search (kernels_brute.hip: brute_f16_kernel): L2-normalised f16 images x 2^8 of rows / queries
// ([*, dim_p], dim_p a multiple of 32, launch_normalize_f16), lower-bound keys against the exact thresholds; appended
// entries are then rewritten with their exact f32 keys by launch_brute_rescore (entries [first[q], cand_cnt[q]) of query q).
struct BruteF16Args {
    const uint16_t *v16;       // [n, dim_p]
    const uint16_t *q16;       // [nq, dim_p]
    // int8 form (v8 != nullptr; dim_p a multiple of 64): images of (unit vector - its mid-range) and per vector
    // {1 / S, residual norm, mid-range, component sum} + |unit vector - mid-range| (launch_normalize_i8); eps then carries only
    // the f32 roundings (eps_sum: of the component sums, per unit of |mid-range|) -- the image bound is per pair
    const int8_t   *v8, *q8;
    const float4   *row_sr, *query_sr;
    const float    *row_n, *query_n;
    const float    *row_max;  // [9] corpus-wide maxima of |sum - mid-range dim|, norm + 3 residual, |mid-range|, residual, 1 / S, and -- as
                              // sortable_bits keys -- of C, -C, A, -A (C = mid-range, A = sum - mid-range dim): the epilogue's screens
    float           dim_f, eps_sum;
    const float *row_aux;      // [n]  |v|^2 (l2 only)
    const float *query_aux;    // [nq] |q|^2 (l2 only)
    uint64_t     row_begin, row_end;
    uint32_t     nq, dim_p;
    int          metric;
    float        eps;          // bound of |s~ - s^| on the cosine scale
    const unsigned long long *thr;
    unsigned long long *cand;
    uint32_t    *cand_cnt;
    uint32_t     cap;
};
hipError_t launch_brute_f16(const BruteF16Args &a, hipStream_t s);
// The k-means assignment on the f16 matrix pipe (kernels_build.hip: assign_f16_kernel): images of (row - mu) / |row - mu| * 2^8
// for rows and centroids (launch_center_normalize_f16; the centroid table padded to kc_pad rows, a multiple of 256),
// candidate lists per row, then launch_assign_rescore picks the exact argmin among them.
struct AssignF16Args {
    const uint16_t *x16;     // [m, dim_p]
    const uint16_t *c16;     // [kc_pad, dim_p]
    const float    *xn2;     // [m]  |row - mu|^2
    const float    *cn2;     // [kc] |centroid - mu|^2
    uint64_t        m;
    uint32_t        kc, kc_pad, dim_p;
    float           eps, cm;  // image dot-product bound (cosine scale); the reference summation margin
    uint32_t       *cand;     // [m][cap] candidate centroid ids
    uint32_t       *cand_cnt; // [m] (zeroed by the caller; may exceed cap: the row is then compared with every centroid)
    uint32_t        cap;
};
hipError_t launch_assign_f16(const AssignF16Args &a, hipStream_t s);
// Round 4 form of the same screen (kernels_build.hip: assign_wide_kernel): block tile 256 rows x 256 centroids (8 waves, 128-byte K
// stages), the data rows on the LANE-owned side of the MFMA so that a row's running best needs no cross-lane reduction, centroid
// images at ONE global scale (cscale = 2^8 / max |c - mu|) so that the whole epilogue of a pair is  t = cn2[c] - kA xs acc  (one
// FMA + one min; d~ = |x - mu|^2 + t).  Candidates carry their t; launch_assign_resolve filters them against the row's FINAL
// best (the kernel's own test only sees the best so far), returns the survivor where one is left and evaluates exactly --
// reference order, argmin by (distance bits, index) -- only the rows that keep two or more.
struct AssignWideArgs {
    const uint16_t *x16;     // [m, dim_p] images of (row - mu) / |row - mu| * 2^8, dim_p a multiple of 64
    const uint16_t *c16;     // [kc_pad, dim_p] images of (centroid - mu) * cscale; rows >= kc are zero
    const float    *xn2;     // [m]  |row - mu|^2
    const float    *cn2;     // [kc] |centroid - mu|^2   (c16 / cn2 in SORTED order: ascending norm; perm[s] = the centroid in slot s)
    const uint32_t *perm;    // [kc]
    const float    *grp_cs;  // [kc_pad / 32] per group of 32 sorted centroids: >= max |c - mu| ...
    const float    *grp_cn;  // [kc_pad / 32] ... and >= max cn2 (a group's own maxima scale its error bound, not the table's: the
                             // few far-out centroids -- clusters of one or two sample points -- loosen only their own group)
    uint64_t        m;
    uint32_t        kc, kc_pad, dim_p;
    float           kA;      // 2 / (2^8 cscale): (row - mu).(centroid - mu) = acc |row - mu| kA / 2
    float           eps, cm;
    uint32_t       *cand;     // [m][cap] candidate centroid ids
    float          *cand_t;   // [m][cap] their lower-bound scores t - E
    uint32_t       *cand_cnt; // [m] zeroed by the caller; may exceed cap (the row is then compared with every centroid)
    float          *best_t;   // [m] smallest upper-bound score t + E of the row over all centroids
    uint32_t        cap;
};
hipError_t launch_assign_wide(const AssignWideArgs &a, hipStream_t s);
hipError_t launch_assign_resolve(const AssignWideArgs &a, const float *rows, const float *centroids, uint32_t dim, uint32_t *cluster,
                                 unsigned long long *stats /* optional [2]: += rows evaluated exactly, += exact evaluations */, hipStream_t s);
hipError_t launch_assign_rescore(const float *rows, const float *centroids, uint64_t m, uint32_t dim, uint32_t kc, const uint32_t *cand,
                                 const uint32_t *cand_cnt, uint32_t cap, uint32_t *cluster, hipStream_t s);
// fixed_scale > 0: every row is multiplied by that one scale instead of 2^8 / its own norm (the centroid images of the round-4 form)
// idx != nullptr: output row r is made from rows[idx[r]]
hipError_t launch_center_normalize_f16(const float *rows, const float *mu, uint64_t n, uint64_t n_pad, uint32_t dim, uint32_t dim_p,
                                       float *out_n2, void *out, hipStream_t s, float fixed_scale = 0.0f, const uint32_t *idx = nullptr);
hipError_t launch_col_mean(const float *m, uint32_t k, uint32_t dim, float *mu, hipStream_t s);
hipError_t launch_normalize_f16(const float *rows, const float *rnorm, uint64_t n, uint32_t dim, uint32_t dim_p, void *out, hipStream_t s);
hipError_t launch_normalize_i8(const float *rows, const float *rnorm, uint64_t n, uint32_t dim, uint32_t dim_p, void *out, void *sr, float *nrm,
                               float *maxima /* [9], zeroed by the caller; nullptr: not wanted */, hipStream_t s);
hipError_t launch_brute_rescore(const BruteArgs &a, const uint32_t *first, hipStream_t s);
// per-row auxiliary values: mode 0 = 1/sqrt(sum x^2) (0 for a zero row), mode 1 = sum x^2, mode 2 = max |x_i|
hipError_t launch_row_norms(const float *rows, uint64_t n, uint32_t dim, int mode, float *out, hipStream_t s);
// fold a query's appended candidates to its k best (kept at the front of the buffer), set
// thr[q] to its k-th key once k candidates exist, report overflow (count > cap) in *overflow
hipError_t launch_brute_select(unsigned long long *cand, uint32_t *cand_cnt, uint32_t cap, uint32_t nq,
                               uint32_t k, unsigned long long *thr, uint32_t *overflow, hipStream_t s);
// *overflow |= 1 if any query's appended count exceeds cap (run BEFORE the select pass, which
// rewrites the buffer fronts)
hipError_t launch_brute_overflow_check(const uint32_t *cand_cnt, uint32_t nq, uint32_t cap, uint32_t *overflow,
                                       hipStream_t s);
// decode the k best keys of every query: row ids + distances
hipError_t launch_brute_finish(const unsigned long long *cand, const uint32_t *cand_cnt, uint32_t cap,
                               uint32_t nq, uint32_t k, uint32_t *row_idx, float *dist, uint32_t *n_found,
                               hipStream_t s);

// Merge of per-shard top-k lists gathered from all ranks: dist/rows [n_shards, nq, k] (unused
// slots: +inf / 0xFFFFFFFF), row_base[n_shards] the shards' first global row.  One wave per
// query; order = ascending (distance, shard, position in the shard's list).
// `stride` = elements between consecutive results in dist / rows (1: two dense arrays; 2: one packed array of
// {f32 distance, u32 row} pairs, dist = base, rows = base + 1)
hipError_t launch_shard_merge(const float *dist, const uint32_t *rows, const long long *row_base,
                              uint32_t n_shards, uint32_t nq, uint32_t k, float *out_dist,
                              long long *out_rows, hipStream_t s, uint32_t stride = 1);

// out[i] = {bits of dist[i], rows[i]} (8 bytes per result): the packed form pqv_merge_topk_packed_device reads
hipError_t launch_pack_pairs(const float *dist, const uint32_t *rows, uint64_t n, void *out, hipStream_t s);

// MFMA-operand copy of the IVF-ordered lists: 16-row tiles, tile T column ch row j at float4 index
// (T * dim/4 + ch) * 16 + j; blk_off[c] = first tile of list c (lists are padded to 16 rows with zeros)
// (row_of, here and below: list position p reads source row row_of[p]; nullptr = the source is already in list order)
hipError_t launch_block_rows(const float *src, const uint64_t *list_off, const uint64_t *blk_off, uint32_t n_clusters,
                             uint64_t max_tiles, uint32_t dim, void *out, hipStream_t s, const uint32_t *row_of = nullptr);

// blocked copy of every quad's queries (see TileArgs::q_blk); ngrp = quad_width / 16
hipError_t launch_pack_queries(const float *queries, const uint32_t *pairs, const uint4 *quads, const uint32_t *n_quads,
                               uint32_t max_quads, uint32_t nprobe, uint32_t dim, uint32_t ngrp, void *q_blk, hipStream_t s);

// MFMA-screened assignment helpers (api.cpp: assign_screened).  assign_setup: pairs[i] = i, quads of `width`
// consecutive queries of cluster 0, cand_base = 0, gthr = EMPTY, *n_quads.  nonfinite_flag: *flag |= 1 if any
// v[i] is inf / NaN.  count_changed: *changed += #{i : cur[i] != prev[i]}.
hipError_t launch_assign_setup(uint32_t *pairs, uint4 *quads, uint32_t *n_quads, uint64_t *cand_base, unsigned long long *gthr,
                               uint32_t nq, uint32_t width, hipStream_t s);
hipError_t launch_nonfinite_flag(const float *v, uint64_t n, uint32_t *flag, hipStream_t s);
// The inverted lists on the device (kernels_build.hip: list_*_kernel): a stable counting sort of the rows by cluster -- list c holds
// the row ids assigned to c in ascending order (index.rs:193-206).  cnt: scratch [k][ceil(n / rows_per_block)] u32, tot: scratch [k] u64,
// list_off: [k + 1] u64, list_rows: [n] u32, *bad: set to 1 if an assignment is >= k (preset 0).  k <= 4096, n < 2^32,
// rows_per_block a multiple of 256.
hipError_t launch_list_sort(const uint32_t *assign, uint64_t n, uint32_t k, uint32_t *cnt, uint32_t rows_per_block,
                            unsigned long long *tot, uint64_t *list_off, uint32_t *list_rows, uint32_t *bad, hipStream_t s);
hipError_t launch_count_changed(const uint32_t *cur, const uint32_t *prev, uint64_t n, unsigned long long *changed, hipStream_t s);

// pqv_rerank's running state <-> merge lists (see kernels_layout.hip)
hipError_t launch_rerank_state_in(const uint32_t *io_rows, const float *io_d2, const uint32_t *io_count, uint32_t k, uint32_t k_list,
                                  uint64_t *keys, uint32_t *vals, uint32_t *rows_saved, hipStream_t s);
hipError_t launch_rerank_state_out(const uint32_t *m_vals, const float *m_d2, const uint32_t *m_found, const uint32_t *rows_saved,
                                   const uint32_t *ids, uint32_t k, uint32_t *io_rows, float *io_d2, uint32_t *io_count,
                                   const uint32_t *m_tie, uint32_t *io_tie, hipStream_t s);

// a[0 .. a_bytes) and b[0 .. b_bytes) = 0xFF bytes in one launch (byte counts: multiples of 16)
hipError_t launch_fill_ones2(void *a, uint64_t a_bytes, void *b, uint64_t b_bytes, hipStream_t s);

// f16 form of the blocked copy (values * scale rounded to nearest f16, 8 dims per 16-byte column, tile T column
// cc row j at 16-byte index (T * dim/8 + cc) * 16 + j) and the corpus maximum it is scaled by
hipError_t launch_block_rows_f16(const float *src, const uint64_t *list_off, const uint64_t *blk_off, uint32_t n_clusters,
                                 uint64_t max_tiles, uint32_t dim, float scale, void *out, hipStream_t s, const uint32_t *row_of = nullptr);
hipError_t launch_maxabs(const float *v, uint64_t n, uint32_t *out_bits, hipStream_t s);
// one pass over n rows (row r = source row row_of ? row_of[r] : r): out_norm2[r] = sum x^2 and *max_bits = max(*max_bits, bits(max |x|))
hipError_t launch_row_norms_max(const float *rows, const uint32_t *row_of, uint64_t n, uint32_t dim, float *out_norm2, uint32_t *max_bits,
                                hipStream_t s);
// int8 form (see kernels_layout.hip): per-list, per-dimension min / max keys (kmin preset to 0xFF bytes, kmax to 0; [n_clusters, dim]),
// the per-list mid-range centre, half range, scale and (zeroed) radius, the blocked int8 copy of the residuals + per-row
// |xi|^2 and residual bound + the per-list radius, and the per-batch images of the (query, probed list) pairs
hipError_t launch_list_minmax(const float *rows, const uint64_t *list_off, uint32_t n_clusters, uint64_t max_list_len, uint32_t dim,
                              uint32_t *kmin, uint32_t *kmax, hipStream_t s, const uint32_t *row_of = nullptr);
hipError_t launch_list_center(const uint32_t *kmin, const uint32_t *kmax, uint32_t n_clusters, uint32_t dim, const uint64_t *list_off,
                              float *center, float *half, float *scale, float *radius, hipStream_t s);
hipError_t launch_global_center(const uint32_t *kmin, const uint32_t *kmax, uint32_t n_clusters, uint32_t dim, const uint64_t *list_off,
                                float *g_center, float *g_half_scale, hipStream_t s);
hipError_t launch_broadcast_center(const float *g_center, const float *g_half_scale, uint32_t n_clusters, uint32_t dim, float *center,
                                   float *half, float *scale, hipStream_t s);
hipError_t launch_block_rows_i8(const float *src, const uint64_t *list_off, const uint64_t *blk_off, uint32_t n_clusters,
                                uint64_t max_tiles, uint32_t dim, const float *center, const float *list_scale, const float *list_half,
                                float *list_radius, void *out, int *row_n2i, float *row_res, hipStream_t s, const uint32_t *row_of = nullptr);
hipError_t launch_quantize_pairs_i8(const float *queries, const uint32_t *probe, const float *center, const float *scale, const float *half,
                                    const float *radius, uint32_t n_pairs, uint32_t nprobe, uint32_t dim, void *q_i8, int *q_n2i,
                                    float *q_res, float *q_resu, float *pair_lb, hipStream_t s);

// k-means++ round with an int8 screen (kernels_build.hip: minupd_screen_kernel; index.rs:354-369): min_d[r] = min(min_d[r],
// d2(row r, row `pick`)) in the reference's order -- but a row's exact distance is evaluated only if a rigorous lower bound
// of it does not exceed min_d[r]: |x - c| >= |xi - ci| / S - rx - rc on int8 images of (row - centre) S (the searcher's residual
// images with ONE list holding the whole subset: launch_list_minmax / launch_list_center / launch_block_rows_i8), |xi - ci|^2 exact
// in int32 from one v_mfma_i32_16x16x64_i8 chain.  After the first dozen rounds a few rows in a hundred are evaluated, and the
// pass reads one byte per value instead of four -- 38 MB for C3's 50 000 x 768 subset, which the L2s keep between rounds.
struct MinUpdScreenArgs {
    const float  *rows;      // [n, dim] f32 (exact evaluations), dim % 64 == 0
    const float4 *img;       // blocked int8 images: 16-row tiles, tile T column cc (16 dims) row j at 16-byte index (T dim/16 + cc) 16 + j
    const int    *n2i;       // [n] |xi|^2
    const float  *res;       // [n] >= |x - centre - xi / S|
    uint64_t      n, pick;
    uint32_t      dim;
    float         inv_s;     // 1 / S
    float         cm;
    float        *min_d;     // [n]
    float        *mirror;    // optional pinned mirror of min_d (see StreamArgs::mirror_f32)
    float        *mirror_t;  // optional chunk-transposed mirror (StreamArgs::mirror_t)
    uint32_t      mirror_chunk, mirror_stride;
    const unsigned long long *pick_dev;   // optional: the pick is read from device memory (the rounds enqueued ahead, kernels_kpp.hip)
    const uint32_t *stop;                 // optional: a non-zero word makes the launch return at once (KppPickArgs::state)
};
hipError_t launch_minupd_screen(const MinUpdScreenArgs &a, hipStream_t s);

// The k-means++ pick of one round on the device (kernels_kpp.hip; index.rs:354-390): total = the worker chunks' sequential f32 sums joined in
// order, threshold = u[round] * total, picks[round] = the first slot whose sequential f32 cumulative sum reaches it -- bit for bit the
// host walk's result.  n <= 57 344, n_chunks <= 1024.  All scratch words start zeroed and carry the round number as a tag.
struct KppPickArgs {
    const float *md;                 // [n] the current minima
    uint32_t n;
    uint32_t chunk, n_chunks;        // worker chunks: chunk c = [c chunk, min(n, (c + 1) chunk))
    unsigned long long *chunk_sum;   // [n_chunks]
    unsigned long long *blk_sum;     // [64]
    unsigned long long *run_sum;     // [4096] pairs of the quarter runs
    unsigned long long *blk_done;    // [64]
    unsigned long long *head;        // [72] the head block's run starts [64] and its last value
    const float *u;                  // [k] gen_range(0.0..1.0) draws by round
    uint32_t round;                  // 1 .. k - 1
    unsigned long long *picks;       // [k] picks[round] is written
    unsigned long long *stamps;      // optional [32]: diagnostics (pqv_kpp_pick with PQV_KPP_STAMPS=1)
    uint32_t *state;                 // [0] != 0: stopped; [1] the round the host has to decide; [2] why (1 total, 2 value, 3 no slot, 4 wait, 5 slot); [3] bits of the last total
};
hipError_t launch_kpp_pick(const KppPickArgs &a, hipStream_t s);

// out[i, :] = src[idx[i], :]  (sampling gather and the IVF-order re-layout)
hipError_t launch_gather_rows(const float *src, const uint32_t *idx32, const uint64_t *idx64,
                              uint64_t m, uint32_t dim, float *out, hipStream_t s);
// out[i, :dim_p] = src[idx32 ? idx32[i] : i, :dim] zero-padded (dim, dim_p multiples of 4): see pad_rows_kernel
hipError_t launch_pad_rows(const float *src, const uint32_t *idx32, uint64_t m, uint32_t dim, uint32_t dim_p, float *out, hipStream_t s);
// f64 -> f32 narrowing of a staged column chunk
hipError_t launch_narrow_f64(const double *src, uint64_t count, float *out, hipStream_t s);

// Lloyd / final assignment: cluster[r] = argmin_j d2(row r, centroid j), strict '<' in
// ascending j.  prev (optional) -> *changed += (prev[r] != cluster[r]); sizes (optional,
// zeroed by the caller) += histogram.
hipError_t launch_assign(const float *rows, uint64_t n, uint32_t dim, const float *centroids,
                         uint32_t k, uint32_t *cluster, const uint32_t *prev,
                         unsigned long long *changed, unsigned long long *sizes,
                         hipStream_t s);

// Lloyd update in reference order: centroid[c][j] = (sum over the cluster's rows in
// ascending row order of x[r][j]) / size, empty clusters stay 0.
hipError_t launch_lloyd_update(const float *rows, uint32_t dim, const uint32_t *list_rows,
                               const uint64_t *list_off, uint32_t k, float *centroids,
                               hipStream_t s);

// number of per-wave partial lists per (query, probed list)
inline uint32_t waves_per_block() { return 4; }

#ifdef PQV_STAMPS
hipError_t stamps_io(unsigned long long *out, int reset);
#endif
// one per kernel unit: an empty launch -- the runtime loads the unit's code object (api.cpp: use_device, once per device) and sets up
// the stream's hardware queue (a corpus' own stream, when it is made)
hipError_t touch_probe(hipStream_t s);
hipError_t touch_screen(hipStream_t s);
hipError_t touch_brute(hipStream_t s);
hipError_t touch_build(hipStream_t s);
hipError_t touch_layout(hipStream_t s);
hipError_t touch_list(hipStream_t s);
hipError_t touch_kpp(hipStream_t s);

}  // namespace pqv
