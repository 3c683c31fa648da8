// api.cpp -- the C ABI of include/pqv.h: host orchestration of the gfx950 kernels.
//
// Host logic mirrored here (reference paths):
//   build_ivf_index / k_means / sample_embeddings   src/ivf/index.rs:152-214,222-242,323-457
//   IvfIndex::{to_bytes,from_bytes,candidate_rows}    src/ivf/index.rs:57-128
//   topk()                                            src/ivf/search.rs:83-142
// There is no CPU compute fallback: distances, argmins and top-k selection only ever run
// in the HIP kernels of kernels_*.hip; the host draws the seeded choices, walks the two
// order-sensitive f32 scalar scans of k-means++ (index.rs:370-383) and moves bytes.
#include "../../include/pqv.h"

#include <hip/hip_runtime.h>
#include <time.h>
#include <unistd.h>

#include <algorithm>
#include <atomic>
#include <functional>
#include <system_error>
#include <thread>
#include <cmath>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <condition_variable>
#include <memory>
#include <mutex>
#include <new>
#include <string>
#include <type_traits>
#include <vector>

#include "internal.h"
#include "kernels.h"
#include "rng.hpp"

namespace {

thread_local std::string g_last_error;

int fail(int code, const std::string &msg) {
    g_last_error = msg;
    return code;
}

#define HIP_TRY(expr)                                                                     \
    do {                                                                                  \
        hipError_t _e = (expr);                                                           \
        if (_e != hipSuccess) {                                                           \
            (void)hipGetLastError();   /* the runtime's sticky last-error must not leak into the next call */ \
            return fail(_e == hipErrorOutOfMemory ? PQV_ERR_OOM : PQV_ERR_HIP,            \
                        std::string(#expr) + ": " + hipGetErrorString(_e));              \
        }                                                                                 \
    } while (0)

// No C++ exception may cross the C ABI: every entry point runs its body through guard().
template <class F>
int guard(F &&body) noexcept {
    try {
        return body();
    } catch (const std::bad_alloc &) {
        return fail(PQV_ERR_OOM, "host allocation failed");
    } catch (const std::exception &e) {
        return fail(PQV_ERR_INVALID, std::string("internal error: ") + e.what());
    } catch (...) {
        return fail(PQV_ERR_INVALID, "internal error");
    }
}

// A few streams per device made (and their hardware queues set up by an empty launch) at the library's first call for the device;
// a searcher takes its stream from here and gives it back -- made right after a build, a stream costs 10+ ms (the driver is still
// unmapping what the build freed).
struct StreamPool {
    std::mutex mu;
    std::vector<hipStream_t> idle[64];
};
static StreamPool &stream_pool() { static StreamPool *p = new StreamPool(); return *p; }     // (never destroyed: streams outlive static teardown order)
static hipError_t pool_get(int device, hipStream_t *out) {
    if (device >= 0 && device < 64) {
        StreamPool &sp = stream_pool();
        std::lock_guard<std::mutex> lock(sp.mu);
        if (!sp.idle[device].empty()) { *out = sp.idle[device].back(); sp.idle[device].pop_back(); return hipSuccess; }
    }
    return hipStreamCreateWithFlags(out, hipStreamNonBlocking);
}
static void pool_put(int device, hipStream_t s) {
    if (!s) return;
    if (device >= 0 && device < 64) {
        (void)hipStreamSynchronize(s);
        StreamPool &sp = stream_pool();
        std::lock_guard<std::mutex> lock(sp.mu);
        if (sp.idle[device].size() < 8) { sp.idle[device].push_back(s); return; }
    }
    (void)hipStreamDestroy(s);
}

int use_device(int device) {
    int count = 0;
    if (hipGetDeviceCount(&count) != hipSuccess || count <= 0)
        return fail(PQV_ERR_NO_DEVICE,
                    "no HIP device available: libpqv_hip has no CPU fallback (gfx950 required)");
    if (device < 0 || device >= count)
        return fail(PQV_ERR_NO_DEVICE, "device index " + std::to_string(device) +
                                           " out of range (" + std::to_string(count) + " devices)");
    HIP_TRY(hipSetDevice(device));
    // The first call for a device loads the kernels' code objects (the runtime would otherwise do it inside whatever launches a unit's
    // first kernel: 7-8 ms of the first index build).  PQV_LAZY_INIT=1 leaves it to the runtime.
    static std::once_flag warmed[64];
    static const bool lazy = [] { const char *e = std::getenv("PQV_LAZY_INIT"); return e && *e == '1'; }();
    if (!lazy && device < 64)
        std::call_once(warmed[device], [] {
            (void)pqv::touch_probe(nullptr); (void)pqv::touch_screen(nullptr); (void)pqv::touch_brute(nullptr); (void)pqv::touch_build(nullptr);
            (void)pqv::touch_layout(nullptr); (void)pqv::touch_list(nullptr); (void)pqv::touch_kpp(nullptr);
            (void)hipStreamSynchronize(nullptr);
            // ... and the runtime's staging buffers for copies from / to pageable host memory are made by the first such copies
            void *d = nullptr;
            if (hipMalloc(&d, 1 << 20) == hipSuccess) {
                std::vector<char> h(1 << 20, 0);
                (void)hipMemcpy(d, h.data(), h.size(), hipMemcpyHostToDevice);
                (void)hipMemcpy(h.data(), d, h.size(), hipMemcpyDeviceToHost);
                (void)hipFree(d);
            }
            for (int i = 0; i < 2; ++i) {
                hipStream_t st = nullptr;
                if (hipStreamCreateWithFlags(&st, hipStreamNonBlocking) != hipSuccess) break;
                (void)pqv::touch_build(st);
                int dev = -1;
                (void)hipGetDevice(&dev);
                pool_put(dev, st);
            }
            (void)hipGetLastError();
        });
    return PQV_OK;
}

// RAII device buffer
bool verbose();
double now_s();
// PQV_VERBOSE: seconds and calls spent in hipMalloc / hipFree since the last report (the build prints them)
static thread_local double g_alloc_s = 0.0, g_free_s = 0.0;
static thread_local unsigned g_alloc_n = 0, g_free_n = 0;
struct DevBuf {
    void *p = nullptr;
    size_t bytes = 0;
    DevBuf() = default;
    DevBuf(const DevBuf &) = delete;
    DevBuf &operator=(const DevBuf &) = delete;
    ~DevBuf() { release(); }
    void release() {
        if (!p) return;
        const bool vb = verbose();
        const double t0 = vb ? now_s() : 0.0;
        (void)hipFree(p); p = nullptr; bytes = 0;
        if (vb) { g_free_s += now_s() - t0; ++g_free_n; }
    }
    hipError_t alloc(size_t n) {
        release();
        if (n == 0) n = 16;
        const bool vb = verbose();
        const double t0 = vb ? now_s() : 0.0;
        hipError_t e = hipMalloc(&p, n);
        if (vb) { g_alloc_s += now_s() - t0; ++g_alloc_n; }
        if (e == hipSuccess) bytes = n; else p = nullptr;
        return e;
    }
    // grow-only
    hipError_t ensure(size_t n) { return (n <= bytes && p) ? hipSuccess : alloc(n + n / 4); }
    template <class T> T *as() const { return static_cast<T *>(p); }
};

struct PinnedBuf {
    void *p = nullptr;
    size_t bytes = 0;
    ~PinnedBuf() { if (p) (void)hipHostFree(p); }
    hipError_t ensure(size_t n) {
        if (n <= bytes && p) return hipSuccess;
        if (p) { (void)hipHostFree(p); p = nullptr; bytes = 0; }
        hipError_t e = hipHostMalloc(&p, n ? n : 16, hipHostMallocDefault);
        if (e == hipSuccess) bytes = n; else p = nullptr;
        return e;
    }
    template <class T> T *as() const { return static_cast<T *>(p); }
};

// phase wall times of this thread's last index build (pqv_index_build_stats)
thread_local double g_build_stats[10] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0};     // [8]: assign_wide_kernel seconds of the final assignment (HIP events), [9]: its launches

// PQV_VERBOSE=1: phase timings of the build on stderr
bool verbose() {
    static const bool v = [] { const char *e = std::getenv("PQV_VERBOSE"); return e && *e && *e != '0'; }();
    return v;
}
double now_s() {
    timespec ts;
    clock_gettime(CLOCK_MONOTONIC, &ts);
    return ts.tv_sec + ts.tv_nsec * 1e-9;
}

uint32_t host_workers() {
    long n = sysconf(_SC_NPROCESSORS_ONLN);
    return n > 0 ? static_cast<uint32_t>(n) : 1u;
}

}  // namespace

namespace pqv_internal {
int fail(int code, const std::string &msg) { return ::fail(code, msg); }
int use_device(int device) { return ::use_device(device); }
}  // namespace pqv_internal

// ---------------------------------------------------------------------------------------
// handles
// ---------------------------------------------------------------------------------------
// the device copy of the inverted lists a build leaves behind (40 MB per 10 M rows, freed with the index): a searcher made on the
// same device takes its row ids from it instead of uploading them again
struct DevRows {
    void *p = nullptr;
    int device = -1;
    uint64_t n = 0;
    ~DevRows() { if (p) { int cur = -1; (void)hipGetDevice(&cur); if (hipSetDevice(device) == hipSuccess) (void)hipFree(p); if (cur >= 0) (void)hipSetDevice(cur); } }
};
// The index' lists as its callers see them: the host vector, plus -- after a build -- the device copy it has NOT been downloaded
// from yet.  A build leaves the lists where it sorted them (4 bytes per row of HBM, freed with the last owner); the host copy is
// made by the first call that reads it (the blob writer, pqv_index_list_rows, a searcher's host-side calls, a searcher on another
// device).  Shared by the index and its searchers.
struct ListRows {
    std::mutex mu;
    std::vector<uint32_t> host;
    std::shared_ptr<DevRows> dev;
    uint64_t n = 0;
    bool pending = false;             // host is still empty: dev holds the lists
    // the host vector, downloaded now if it has to be (nullptr + fail() if that goes wrong)
    const std::vector<uint32_t> *get() {
        std::lock_guard<std::mutex> lock(mu);
        if (pending) {
            int cur = -1;
            (void)hipGetDevice(&cur);
            hipError_t e = hipSetDevice(dev->device);
            if (e == hipSuccess) {
                try { host.resize(n); } catch (const std::bad_alloc &) { if (cur >= 0) (void)hipSetDevice(cur); (void)fail(PQV_ERR_OOM, "host allocation failed"); return nullptr; }
                e = hipMemcpy(host.data(), dev->p, n * sizeof(uint32_t), hipMemcpyDeviceToHost);
            }
            if (cur >= 0) (void)hipSetDevice(cur);
            if (e != hipSuccess) { host.clear(); (void)fail(PQV_ERR_HIP, std::string("download of the inverted lists: ") + hipGetErrorString(e)); return nullptr; }
            pending = false;
        }
        return &host;
    }
};
struct pqv_index {
    uint32_t dim = 0;
    uint32_t n_clusters = 0;
    std::vector<float> centroids;     // [n_clusters * dim]
    std::vector<uint64_t> list_off;   // [n_clusters + 1]
    // concatenated inverted lists (see ListRows); `list_rows` is the host vector for the code that FILLS it -- readers go through rows->get()
    std::shared_ptr<ListRows> rows{std::make_shared<ListRows>()};
    std::vector<uint32_t> &list_rows{rows->host};
    uint64_t n_rows() const { return rows->pending ? rows->n : rows->host.size(); }
    uint64_t permutation_of = 0;      // != 0: the lists are a permutation of [0, permutation_of) by construction (a build's result)
    pqv_index() = default;
    pqv_index(const pqv_index &) = delete;
    pqv_index &operator=(const pqv_index &) = delete;
};

// pinned staging buffers of the streaming upload (pqv_corpus_write_rows): a buffer is free, being filled by a caller, or in
// flight behind its event
struct UploadStage {
    static constexpr int N = 16;                  // (a page-level reader hands over ~1 MB pages from eight threads: many small buffers)
    static constexpr size_t BYTES = 8u << 20;
    std::mutex mu;
    std::condition_variable cv;
    void *pin[N] = {};
    void *dev64[N] = {};                          // f64 batches: device-side landing area of the same size (made on first use)
    hipEvent_t ev[N] = {};
    int state[N] = {};                            // 0 free, 1 claimed, 2 in flight
    static constexpr int NS = 4;                  // copy streams, slot i uses stream i % NS: several DMA operations in flight (1 MB pages: the
                                                  // per-operation overhead of ONE engine capped the loader at ~15 GB/s)
    hipStream_t copy_stream[NS] = {};
    ~UploadStage() {
        for (int i = 0; i < N; ++i) {
            if (ev[i]) (void)hipEventDestroy(ev[i]);
            if (pin[i]) (void)hipHostFree(pin[i]);
            if (dev64[i]) (void)hipFree(dev64[i]);
        }
        for (int i = 0; i < NS; ++i) if (copy_stream[i]) (void)hipStreamDestroy(copy_stream[i]);
    }
};

struct pqv_corpus {
    int device = 0;
    uint32_t dim = 0;
    uint64_t n = 0;
    uint64_t capacity = 0;
    float *d_rows = nullptr;  // [capacity, dim]
    bool owned = true;
    hipStream_t stream = nullptr;
    std::unique_ptr<UploadStage> upload;       // made by the first pqv_corpus_write_rows
    std::mutex upload_mu;
    // lazily computed per-row auxiliaries of pqv_brute_topk: 1/|v| and |v|^2
    mutable std::mutex aux_mu;
    mutable DevBuf aux_rnorm, aux_norm2, aux_v16;      // aux_v16: L2-normalised f16 images [n, dim_p] (the f16 screen of pqv_brute_topk)
    mutable uint64_t aux_rnorm_rows = 0, aux_norm2_rows = 0, aux_v16_rows = 0;
    mutable DevBuf aux_v8_max;                         // [12, nine used] maxima over the rows (kernels.h: BruteF16Args::row_max)
    mutable DevBuf aux_v8, aux_v8_sr, aux_v8_n;        // int8 images [n, dim_p8] + {1 / S, residual, mid-range, sum} + norm per row (the int8 screen of pqv_brute_topk)
    mutable uint64_t aux_v8_rows = 0;
    ~pqv_corpus() {
        if (d_rows && owned) (void)hipFree(d_rows);
        if (stream) (void)hipStreamDestroy(stream);
    }
};

// Per-call device scratch of a searcher (see pqv_searcher::lanes).
constexpr int PQV_LANES = 4;
struct Scratch {
    DevBuf s_probe_keys, s_probe_vals, s_probe, s_cand_base, s_ncand, s_part_keys, s_part_vals, s_queries, s_rows,
        s_dist, s_nfound, s_pair_u32, s_pairs, s_groups, s_quads, s_items, s_ticket, s_ticket2, s_cand_keys, s_cand_vals, s_cand_cnt, s_spilled,
        s_seed_ub, s_qblk, s_gthr, s_tie, s_replay, s_qnorm, s_qmax, s_thr_hist, s_thr_bins, s_qi8, s_qn2i, s_qres, s_qresu, s_pair_lb, s_part_flags, s_qpad, s_cand_lb, s_pendv, s_work, s_nwork, s_out;
    PinnedBuf h_io;                 // small host calls: queries in, one block of results out, through pinned memory
    hipEvent_t done = nullptr;      // recorded after the last kernel of the call that used this lane
    hipStream_t stream = nullptr;   // the stream of that call
    bool used = false;
    // the lane's side stream for the wide-quad launch of a batch (TileArgs::side_stream), created on first use
    hipStream_t side = nullptr;
    hipEvent_t ev_fork = nullptr, ev_join = nullptr;
    ~Scratch() {
        if (done) (void)hipEventDestroy(done);
        if (ev_fork) (void)hipEventDestroy(ev_fork);
        if (ev_join) (void)hipEventDestroy(ev_join);
        if (side) (void)hipStreamDestroy(side);
    }
};

struct pqv_searcher {
    int device = 0;
    uint32_t dim = 0, n_clusters = 0;
    // storage dimension of the IVF-ordered rows: dim, or dim zero-padded to a multiple of 64 / 128 / 256 where the MFMA
    // screen has no tiling for dim itself (dim % 4 == 0; kernels_layout.hip: pad_rows_kernel -- distances stay bit-identical).
    // Everything downstream of the probe (blocked copies, screen, exact evaluation, replays) works on sdim-wide rows
    // and sdim-wide copies of the batch's queries.
    uint32_t sdim = 0;
    uint64_t n = 0;
    uint64_t max_list_len = 0;
    pqv_corpus *corpus = nullptr;          // borrowed
    std::vector<uint64_t> h_list_off;      // host copy for candidate_rows
    std::shared_ptr<ListRows> h_rows;      // the index' lists (shared, not copied; the host copy is made by the first call that reads it)
    DevBuf d_cent_t;                   // [dim/4][kc_pad] float4 transpose of the centroids (probe_rows_kernel), dim % 4 == 0
    uint32_t kc_pad = 0;
    DevBuf d_centroids, d_list_off, d_ids, d_mat_ivf, d_stats;
    mutable DevBuf d_row_norm2;            // |x|^2 per storage row (f32 / f16 MFMA screens; see norms_done)
    const float *d_mat = nullptr;          // row storage the re-rank reads
    const uint32_t *d_row_of = nullptr;    // list position -> storage row (ROW_ORDER layout, and the images-only IVF layout)
    const uint32_t *d_final_ids = nullptr; // storage row -> file row id (IVF layout with its own f32 copy)
    // IVF-ordered layout WITHOUT a second f32 copy of the corpus (round 4): everything the screen streams -- the blocked operand
    // images, row terms and norms -- is in list order, while the exact evaluations (0.03 % of the pairs, the seed refinement, tie
    // replays) read the caller's row-order matrix through d_row_of.  Footprint 2.28 x -> 1.28 x the corpus on C3, and searcher
    // creation no longer allocates and writes n x dim floats.
    bool images_only = false;
    hipStream_t stream = nullptr;
    // scratch (guarded by mu): PQV_LANES independent sets, one per stream in use, so calls enqueued on
    // different streams overlap on the GPU (the tail of one batch's screen kernel is filled by the next
    // batch's probe / bucketing / seed kernels) instead of serialising on shared buffers
    mutable std::mutex mu;
    mutable Scratch lanes[PQV_LANES];
    // {wide items, of which in single-quad lists} of a recent batch, written by pair_scan_kernel straight into this pinned
    // buffer (a hint for the NEXT batch's cache policy, read without synchronisation -- wide_rows_nt)
    mutable PinnedBuf h_wide_stats;
    mutable uint32_t lane_rr = 0;
    // blocked MFMA-operand copies of the lists, one per operand form in use (0 f32, 1 f16, 2 int8); built at creation
    // for the form the dispatch rule picks, a second form only if a later call asks for it (e.g. k > 32 on short lists)
    mutable DevBuf d_mat_blk_op[3], d_blk_off;
    // f16 operands for the wide screened path: values * f16_scale rounded to f16; possible when the stored rows
    // are finite and the power-of-two scale and its square are representable
    mutable bool f16_ok = false;
    mutable float f16_scale = 1.0f;
    // The rows' squared norms and their maximum (-> f16_ok / f16_scale) are one pass over the corpus that the int8 screen never
    // reads: where the int8 copy is built at creation they are left to the first call that needs them (ensure_row_norms).
    mutable bool norms_done = false;
    mutable std::mutex norms_mu;
    uint64_t n_storage = 0;
    // int8 operands (rows of a multiple of 256 dims): per LIST the images of (x - centre_c) * S_c -- the IVF residual:
    // centre_c the per-dimension mid-range of the list's rows, S_c mapping the list's largest |x - centre_c| component
    // to 127 -- per row |xi|^2 and an upper bound of the residual norm, per list a radius >= |x - centre_c| (see
    // kernels_layout.hip: block_rows_i8_kernel); built with the blocked copy (ensure_blocked_copy)
    mutable bool i8_ok = false;
    // i8_residual: the per-list (residual) form pays where the lists are much tighter than the corpus (clustered data:
    // a per-list scale twice the global one halves the bound's slack); where they are not (uniform data) the one-centre
    // form gives the same bound with ONE image per query instead of one per probed pair -- 0.45 GB less traffic per
    // C3 step and an L2-hot staging source.  Decided when the int8 copy is built (ensure_blocked_copy).
    mutable bool i8_residual = true;
    mutable DevBuf d_center, d_list_scale, d_list_half, d_list_radius;
    mutable DevBuf d_row_n2i, d_row_res;
    // Tunables.  Defaults are what the dispatch rules below were measured with; every one can be set per
    // searcher through pqv_searcher_set_option (tests and benches use that to force a path) and, for the
    // profiling scripts, through a PQV_<NAME> environment variable read ONCE, when the searcher is created.
    struct Opts {
        int rerank_mode = 0;               // 0 auto, 1 stream_kernel, 2 tile path
        int tile_filter = 1;               // MFMA lower-bound screen in the batched path: 0 off, 1 by rule, 2 forced
        int filter_variant = 0;            // 1: one 16-query group per block (tile_filter_kernel)
        uint32_t cand_cap = 0;             // candidate-buffer entries per query of the wide screened path (0 = by rule: 2048, 8192 for k > 32)
        int screen_f16 = 1;                // f16 operands where possible
        int screen_i8 = 1;                 // int8 operands where possible (dim % 256 == 0)
        uint32_t seed_rows = 0;            // rows per list sampled for the thresholds (0 = by rule)
        uint32_t wide_rows = 0;            // rows per block of the wide kernel (0 = by rule)
        uint32_t tile_rows = 0;            // rows per block of the exact tile kernel (0 = 1536)
        int running_thr = 1;               // running thresholds of the wide kernel
        int defer = 1;                     // wide kernel, k > 64: survivors are appended with their bounds, evaluated after the filter (0: in it)
        int quad_xcd = -1;                 // quad-to-XCD affinity of the wide kernels (-1 = by rule)
        int single_bucket = 1;             // one query: the probe merge writes the bucketing, no pair-sort launches
        int seed_refine = 1;               // exact distances behind the k selected seed bounds tighten the first threshold (k <= 16)
        int item_grid = 1;                 // wide filter kernel: 1-D grid over (quad, existing row chunk) items
        int chunk_major = 1;               // ... numbered chunk-major (row chunk 0 of every quad first)
        int wide_waves = 0;                // waves per block of the wide kernel: 0 by rule, 4 or 8
        int probe_rows = 1;                // batched centroid probe (probe_rows_kernel) when dim % 4 == 0; 0 = stream_kernel
        uint32_t quad_width = 0;           // queries per quad of the wide kernel (0 = by rule)
        uint32_t min_blocks = 0;           // wide kernel: blocks a launch should at least have before rows per block shrink (0 = by rule)
        int pair_prune = 1;                // int8 path: drop (query, list) pairs whose centre-distance bound exceeds the query's threshold
        int i8_form = 0;                   // int8 images: 0 by rule (per-list residual where the lists are tight), 1 one centre, 2 residual
        int xcd_items = 1;                 // PairSortArgs::xcd_items (PQV_XCD_ITEMS)
        int pf96 = 1;                      // TileArgs::opt_pf96 (PQV_PF96)
        int drain_min = 0;                 // TileArgs::drain_min (PQV_DRAIN_MIN)
        int fork_wide = 0;                 // the wide-quad launch on the lane's side stream, beside the regular instance (PQV_FORK_WIDE)
        int wide_quads = 1;                // int8, 96-query quads: lists probed by 97..160 queries of the batch take ONE 160-query quad
                                           // (8-wave blocks on 32-row tiles) instead of two passes; 0 = off
        uint32_t wide_quad_rows = 0;       // rows per block of that instance (0 = by rule: 6144)
        int list_once = 0;                 // round 6: 1 = lists probed by more than 96 queries of the batch go to list_filter_kernel (rows stationary in
                                           // registers, ALL the list's pairs streamed past them: every such list is read once -- and measured SLOWER than
                                           // the wide-quad instance / regular quads sharing an L2: C3 2.18 against 1.87 ms of kernels, the mixture 3.7
                                           // against 2.1; DESIGN 5.4d has the counters); 0 (default) = the wide-quad instance
    };
    mutable Opts opt;
    mutable pqv_counters_t counters{};
    // timing
    mutable bool timing = false;
    mutable std::vector<hipEvent_t> ev;    // triples: probe-start, rerank-start, rerank-stop, end
    ~pqv_searcher();
};
pqv_searcher::~pqv_searcher() {
    for (auto e : ev) (void)hipEventDestroy(e);
    pool_put(device, stream);          // (back to the device's pool: see StreamPool)
}

// The scratch lane of a call on `stream`: the lane that stream used last, else a free one, else the least
// recently assigned one.  Resolved ONCE per public entry point (under s->mu) and handed down: a lane taken
// over from another stream is first ordered behind that stream's last call (its `done` event), and
// lane_release() records the new owner when the call's last kernel has been enqueued.
static int lane_acquire(const pqv_searcher *s, hipStream_t stream, Scratch **out) {
    Scratch *sc = nullptr;
    for (auto &l : s->lanes) if (l.used && l.stream == stream) { sc = &l; break; }
    if (!sc) for (auto &l : s->lanes) if (!l.used) { sc = &l; break; }
    if (!sc) sc = &s->lanes[s->lane_rr++ % PQV_LANES];     // only an eviction advances the cursor
    if (!sc->done) HIP_TRY(hipEventCreateWithFlags(&sc->done, hipEventDisableTiming));
    if (sc->used && sc->stream != stream) HIP_TRY(hipStreamWaitEvent(stream, sc->done, 0));
    // owned by `stream` from here on, also if the call fails half-way: whatever it enqueued runs on `stream`
    sc->stream = stream; sc->used = true;
    *out = sc;
    return PQV_OK;
}
static int lane_release(Scratch &sc, hipStream_t stream) {
    HIP_TRY(hipEventRecord(sc.done, stream));
    return PQV_OK;
}
// Records the lane's `done` event on EVERY way out of an entry point (an error return half-way through a call has enqueued
// work on `stream` that the lane's next owner must still be ordered behind); recording again after a successful
// lane_release() is harmless.
struct LaneGuard {
    Scratch &sc; hipStream_t stream;
    ~LaneGuard() { if (sc.done) (void)hipEventRecord(sc.done, stream); }
};

// ---------------------------------------------------------------------------------------
// misc
// ---------------------------------------------------------------------------------------
extern "C" const char *pqv_last_error(void) { return g_last_error.c_str(); }

extern "C" int pqv_device_count(void) {
    int count = 0;
    if (hipGetDeviceCount(&count) != hipSuccess) return 0;
    return count < 0 ? 0 : count;
}

extern "C" int pqv_abi_version(void) { return 101; }

extern "C" int pqv_diag_rng(const uint8_t *seed32, uint64_t seed64, int mode, uint64_t arg, uint64_t *out, uint64_t n) {
    return guard([&] {
        if (!out && n) return fail(PQV_ERR_INVALID, "out must not be NULL");
        pqv::StdRng rng = seed32 ? pqv::StdRng::from_seed(seed32) : pqv::StdRng::seed_from_u64(seed64);
        if (mode == 4) {
            if (n > arg) return fail(PQV_ERR_INVALID, "amount exceeds length");
            const std::vector<uint64_t> s = pqv::index_sample(rng, arg, n);
            for (uint64_t i = 0; i < n; ++i) out[i] = s[i];
            return static_cast<int>(PQV_OK);
        }
        for (uint64_t i = 0; i < n; ++i) {
            if (mode == 0) out[i] = rng.next_u64();
            else if (mode == 1) out[i] = rng.next_u32();
            else if (mode == 2) out[i] = rng.range_usize(0, arg);
            else if (mode == 3) { const float f = rng.unit_f32(); uint32_t b; std::memcpy(&b, &f, 4); out[i] = b; }
            else return fail(PQV_ERR_INVALID, "unknown mode");
        }
        return static_cast<int>(PQV_OK);
    });
}

// ---------------------------------------------------------------------------------------
// corpus
// ---------------------------------------------------------------------------------------
static int pqv_corpus_create_impl(int device, uint64_t capacity_rows, uint32_t dim,
                                 pqv_corpus **out) {
    if (!out) return fail(PQV_ERR_INVALID, "out must not be NULL");
    *out = nullptr;
    if (dim == 0) return fail(PQV_ERR_INVALID, "Embedding dimension must be > 0");  // mod.rs:59
    if (capacity_rows > 0xFFFFFFFFull)
        return fail(PQV_ERR_UNSUPPORTED, "row ids are u32: at most 4294967295 rows per corpus");
    if (int rc = use_device(device)) return rc;
    pqv_corpus *c = new (std::nothrow) pqv_corpus();
    if (!c) return fail(PQV_ERR_OOM, "host allocation failed");
    c->device = device;
    c->dim = dim;
    c->capacity = capacity_rows;
    const size_t bytes = std::max<size_t>(16, static_cast<size_t>(capacity_rows) * dim * sizeof(float));
    hipError_t e = hipMalloc(reinterpret_cast<void **>(&c->d_rows), bytes);
    if (e != hipSuccess) {
        delete c;
        return fail(PQV_ERR_OOM, std::string("hipMalloc(corpus): ") + hipGetErrorString(e));
    }
    e = hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking);
    if (e != hipSuccess) {
        delete c;
        return fail(PQV_ERR_HIP, std::string("hipStreamCreate: ") + hipGetErrorString(e));
    }
    (void)pqv::touch_build(c->stream);            // (the stream's hardware queue is set up by its first launch: here, not in the build)
    (void)hipStreamSynchronize(c->stream);
    *out = c;
    return PQV_OK;
}
extern "C" int pqv_corpus_create(int device, uint64_t capacity_rows, uint32_t dim,
                                 pqv_corpus **out) {
    return guard([&] { return pqv_corpus_create_impl(device, capacity_rows, dim, out); });
}

static int pqv_corpus_append_impl(pqv_corpus *c, const float *rows, uint64_t n_rows) {
    if (!c) return fail(PQV_ERR_INVALID, "corpus must not be NULL");
    if (!c->owned) return fail(PQV_ERR_INVALID, "cannot append to a borrowed device buffer");
    if (n_rows == 0) return PQV_OK;
    if (!rows) return fail(PQV_ERR_INVALID, "rows must not be NULL");
    if (c->n + n_rows > c->capacity) return fail(PQV_ERR_INVALID, "corpus capacity exceeded");
    if (int rc = use_device(c->device)) return rc;
    HIP_TRY(hipMemcpy(c->d_rows + c->n * c->dim, rows, static_cast<size_t>(n_rows) * c->dim * sizeof(float),
                      hipMemcpyHostToDevice));
    c->n += n_rows;
    return PQV_OK;
}
extern "C" int pqv_corpus_append(pqv_corpus *c, const float *rows, uint64_t n_rows) {
    return guard([&] { return pqv_corpus_append_impl(c, rows, n_rows); });
}

static int pqv_corpus_append_f64_impl(pqv_corpus *c, const double *rows, uint64_t n_rows) {
    if (!c) return fail(PQV_ERR_INVALID, "corpus must not be NULL");
    if (!c->owned) return fail(PQV_ERR_INVALID, "cannot append to a borrowed device buffer");
    if (n_rows == 0) return PQV_OK;
    if (!rows) return fail(PQV_ERR_INVALID, "rows must not be NULL");
    if (c->n + n_rows > c->capacity) return fail(PQV_ERR_INVALID, "corpus capacity exceeded");
    if (int rc = use_device(c->device)) return rc;
    const uint64_t count = n_rows * c->dim;
    DevBuf stage;
    HIP_TRY(stage.alloc(count * sizeof(double)));
    HIP_TRY(hipMemcpy(stage.p, rows, count * sizeof(double), hipMemcpyHostToDevice));
    HIP_TRY(pqv::launch_narrow_f64(stage.as<double>(), count, c->d_rows + c->n * c->dim, c->stream));
    HIP_TRY(hipStreamSynchronize(c->stream));
    c->n += n_rows;
    return PQV_OK;
}
extern "C" int pqv_corpus_append_f64(pqv_corpus *c, const double *rows, uint64_t n_rows) {
    return guard([&] { return pqv_corpus_append_f64_impl(c, rows, n_rows); });
}

// ---- streaming upload (N1) ----------------------------------------------------------------
namespace {
int upload_stage_of(pqv_corpus *c, UploadStage **out) {
    std::lock_guard<std::mutex> lock(c->upload_mu);
    if (!c->upload) {
        std::unique_ptr<UploadStage> u(new UploadStage());
        for (int i = 0; i < UploadStage::NS; ++i) HIP_TRY(hipStreamCreateWithFlags(&u->copy_stream[i], hipStreamNonBlocking));
        for (int i = 0; i < UploadStage::N; ++i) {
            HIP_TRY(hipHostMalloc(&u->pin[i], UploadStage::BYTES, hipHostMallocDefault));
            HIP_TRY(hipEventCreateWithFlags(&u->ev[i], hipEventDisableTiming));
        }
        c->upload = std::move(u);
    }
    *out = c->upload.get();
    return PQV_OK;
}
// a staging buffer nobody else is filling: a free one, else the in-flight one whose DMA ends first (waited for outside the lock)
int upload_claim(UploadStage *u, int *slot) {
    std::unique_lock<std::mutex> lock(u->mu);
    for (;;) {
        for (int i = 0; i < UploadStage::N; ++i) if (u->state[i] == 0) { u->state[i] = 1; *slot = i; return PQV_OK; }
        for (int i = 0; i < UploadStage::N; ++i)
            if (u->state[i] == 2) {
                u->state[i] = 1;                       // ours; its previous DMA still has to finish
                lock.unlock();
                const hipError_t e = hipEventSynchronize(u->ev[i]);
                if (e != hipSuccess) { lock.lock(); u->state[i] = 0; u->cv.notify_one(); lock.unlock(); HIP_TRY(e); }
                *slot = i;
                return PQV_OK;
            }
        u->cv.wait(lock);                              // every buffer is being filled by another thread
    }
}
void upload_release(UploadStage *u, int slot, bool in_flight) {
    { std::lock_guard<std::mutex> lock(u->mu); u->state[slot] = in_flight ? 2 : 0; }
    u->cv.notify_one();
}
template <class T>
int corpus_write_rows(pqv_corpus *c, uint64_t row_offset, const T *rows, uint64_t n_rows) {
    if (!c) return fail(PQV_ERR_INVALID, "corpus must not be NULL");
    if (!c->owned) return fail(PQV_ERR_INVALID, "cannot write into a borrowed device buffer");
    if (n_rows == 0) return PQV_OK;
    if (!rows) return fail(PQV_ERR_INVALID, "rows must not be NULL");
    if (row_offset > c->capacity || n_rows > c->capacity - row_offset) return fail(PQV_ERR_INVALID, "corpus capacity exceeded");
    if (int rc = use_device(c->device)) return rc;
    UploadStage *u = nullptr;
    if (int rc = upload_stage_of(c, &u)) return rc;
    const uint64_t row_bytes = static_cast<uint64_t>(c->dim) * sizeof(T);
    const uint64_t rows_per_piece = std::max<uint64_t>(1, UploadStage::BYTES / row_bytes);
    if (row_bytes > UploadStage::BYTES) return fail(PQV_ERR_INVALID, "a row exceeds the staging buffer");
    for (uint64_t r0 = 0; r0 < n_rows; r0 += rows_per_piece) {
        const uint64_t m = std::min<uint64_t>(rows_per_piece, n_rows - r0);
        int slot = -1;
        if (int rc = upload_claim(u, &slot)) return rc;
        std::memcpy(u->pin[slot], rows + r0 * c->dim, m * row_bytes);
        hipStream_t cs = u->copy_stream[slot % UploadStage::NS];
        float *dst = c->d_rows + (row_offset + r0) * c->dim;
        hipError_t e = hipSuccess;
        if (std::is_same<T, float>::value) {
            e = hipMemcpyAsync(dst, u->pin[slot], m * row_bytes, hipMemcpyHostToDevice, cs);
        } else {
            if (!u->dev64[slot]) e = hipMalloc(&u->dev64[slot], UploadStage::BYTES);
            if (e == hipSuccess) e = hipMemcpyAsync(u->dev64[slot], u->pin[slot], m * row_bytes, hipMemcpyHostToDevice, cs);
            if (e == hipSuccess) e = pqv::launch_narrow_f64(static_cast<const double *>(u->dev64[slot]), m * c->dim, dst, cs);
        }
        if (e == hipSuccess) e = hipEventRecord(u->ev[slot], cs);
        upload_release(u, slot, e == hipSuccess);
        HIP_TRY(e);
    }
    return PQV_OK;
}
}  // namespace
extern "C" int pqv_corpus_write_rows(pqv_corpus *c, uint64_t row_offset, const float *rows, uint64_t n_rows) {
    return guard([&] { return corpus_write_rows<float>(c, row_offset, rows, n_rows); });
}
extern "C" int pqv_corpus_write_rows_f64(pqv_corpus *c, uint64_t row_offset, const double *rows, uint64_t n_rows) {
    return guard([&] { return corpus_write_rows<double>(c, row_offset, rows, n_rows); });
}
extern "C" int pqv_corpus_finish(pqv_corpus *c, uint64_t n_rows) {
    return guard([&]() -> int {
        if (!c) return fail(PQV_ERR_INVALID, "corpus must not be NULL");
        if (n_rows > c->capacity) return fail(PQV_ERR_INVALID, "corpus capacity exceeded");
        if (int rc = use_device(c->device)) return rc;
        if (c->upload) {
            for (int i = 0; i < UploadStage::NS; ++i) HIP_TRY(hipStreamSynchronize(c->upload->copy_stream[i]));
            std::lock_guard<std::mutex> lock(c->upload_mu);
            c->upload.reset();                     // the pinned buffers go back: a resident corpus does not keep 128 MB of them
        }
        c->n = n_rows;
        return PQV_OK;
    });
}

// ---- Parquet page helpers (N1; host only) ---------------------------------------------------
namespace {
// One run of the RLE / bit-packed hybrid at a time: fn(value, count) for an RLE run, fn(values...) one by one for a bit-packed
// group.  Returns false on a malformed stream.  `want` values are consumed (a final bit-packed group may hold padding).
template <class F>
bool hybrid_runs(const uint8_t *p, uint64_t len, uint32_t bw, uint64_t want, F &&fn) {
    if (bw > 32) return false;
    const uint8_t *end = p + len;
    const uint32_t vbytes = (bw + 7) / 8;
    const uint64_t mask = bw == 32 ? 0xFFFFFFFFull : ((1ull << bw) - 1ull);
    uint64_t got = 0;
    while (got < want) {
        uint64_t h = 0;
        for (uint32_t shift = 0;; shift += 7) {
            if (p >= end || shift > 56) return false;
            const uint8_t b = *p++;
            h |= static_cast<uint64_t>(b & 0x7F) << shift;
            if (!(b & 0x80)) break;
        }
        if (h & 1) {                                     // bit-packed: (h >> 1) groups of 8 values
            // (group count checked BEFORE the multiplications: a crafted 9-byte varint would wrap nbytes to 0 and pass)
            const uint64_t groups = h >> 1;
            if (groups == 0 || groups > (1ull << 56) || (bw != 0 && groups > static_cast<uint64_t>(end - p) / bw)) return false;
            const uint64_t nvals = groups * 8, nbytes = groups * bw;
            const uint64_t use = std::min<uint64_t>(nvals, want - got);
            uint64_t acc = 0; uint32_t nb = 0; const uint8_t *q = p;
            for (uint64_t i = 0; i < use; ++i) {
                while (nb < bw) { acc |= static_cast<uint64_t>(*q++) << nb; nb += 8; }
                if (!fn(static_cast<uint32_t>(acc & mask), 1ull)) return true;       // the callback has its answer
                acc >>= bw; nb -= bw;
            }
            p += nbytes; got += use;
        } else {                                         // RLE: count, value in ceil(bw / 8) bytes
            const uint64_t cnt = h >> 1;
            if (cnt == 0 || static_cast<uint64_t>(end - p) < vbytes) return false;
            uint32_t v = 0;
            for (uint32_t i = 0; i < vbytes; ++i) v |= static_cast<uint32_t>(p[i]) << (8 * i);
            p += vbytes;
            const uint64_t use = std::min<uint64_t>(cnt, want - got);
            if (!fn(v, use)) return true;
            got += use;
        }
    }
    return true;
}
}  // namespace
extern "C" int pqv_parquet_levels_check(const uint8_t *buf, uint64_t len, uint32_t bit_width, uint64_t n_values, int mode, uint64_t expect,
                                        uint64_t *period_out) {
    return guard([&]() -> int {
        if (n_values == 0) return 0;
        if (!buf) return fail(PQV_ERR_INVALID, "buf must not be NULL");
        if (mode == 1 && expect == 0) {
            // discover the list length: the position of the second level 0 (the first must be at position 0)
            if (!period_out) return fail(PQV_ERR_INVALID, "list length must be > 0");
            uint64_t at = 0, zeros = 0, second = 0;
            const bool w0 = hybrid_runs(buf, len, bit_width, n_values, [&](uint32_t v, uint64_t cnt) {
                if (v == 0) {
                    if (zeros == 0 && at != 0) { zeros = 99; return false; }          // the page starts inside a row
                    if (zeros == 0 && cnt >= 2) { second = 1; zeros = 2; return false; }
                    if (zeros == 1) { second = at; zeros = 2; return false; }
                    zeros = 1;
                }
                at += cnt;
                return true;
            });
            if (!w0) return fail(PQV_ERR_INVALID, "malformed RLE / bit-packed level run");
            if (zeros == 99 || zeros == 0) return 1;
            expect = zeros == 2 ? second : n_values;
            *period_out = expect;
            if (expect == 0) return 1;
        }
        bool ok = true;
        uint64_t pos = 0;                                // mode 1: position inside the page, in values
        const bool well = hybrid_runs(buf, len, bit_width, n_values, [&](uint32_t v, uint64_t cnt) {
            if (mode == 0) { ok = v == expect; return ok; }
            // repetition levels: 0 exactly at the multiples of `expect`
            const uint64_t ph = pos % expect;
            if (v == 0) ok = ph == 0 && (cnt == 1 || expect == 1);
            else ok = v == 1 && ph != 0 && ph + cnt <= expect;
            pos += cnt;
            return ok;
        });
        if (!well) return fail(PQV_ERR_INVALID, "malformed RLE / bit-packed level run");
        if (ok && mode == 1 && pos % expect != 0) ok = false;     // the page ends inside a row
        return ok ? 0 : 1;
    });
}
namespace {
// (levels as above, without the error text: 0 as expected, 1 different, -1 malformed)
int levels_plain(const uint8_t *buf, uint64_t len, uint32_t bw, uint64_t n_values, int mode, uint64_t expect) {
    bool ok = true;
    uint64_t pos = 0;
    const bool well = hybrid_runs(buf, len, bw, n_values, [&](uint32_t v, uint64_t cnt) {
        if (mode == 0) { ok = v == expect; return ok; }
        const uint64_t ph = pos % expect;
        if (v == 0) ok = ph == 0 && (cnt == 1 || expect == 1);
        else ok = v == 1 && ph != 0 && ph + cnt <= expect;
        pos += cnt;
        return ok;
    });
    if (!well) return -1;
    if (ok && mode == 1 && pos % expect != 0) ok = false;
    return ok ? 0 : 1;
}
}  // namespace

// A run of uncompressed PLAIN v1 data pages of a `List<f32|f64>` leaf, straight from the mapped file: per page the two level
// runs are checked (repetition level 0 exactly every `dim` values, every definition level == max_def) and the values behind
// them are uploaded like pqv_corpus_write_rows at row first_value / dim.  One call per dozen pages keeps a Python caller's
// interpreter lock out of the loop.  Returns 0, or 1 with *bad_page = the first page that is not what the plan assumed
// (nothing of it is uploaded; the caller re-reads the column through its general reader).
extern "C" int pqv_corpus_write_plain_pages(pqv_corpus *c, const uint8_t *file_base, const uint64_t *body_off, const uint32_t *body_len,
                                            const uint64_t *first_value, const uint32_t *n_values, uint32_t n_pages, uint32_t dim,
                                            uint32_t max_def, int f64, uint32_t *bad_page) {
    return guard([&]() -> int {
        if (!c) return fail(PQV_ERR_INVALID, "corpus must not be NULL");
        if (!file_base || !body_off || !body_len || !first_value || !n_values) return fail(PQV_ERR_INVALID, "page tables must not be NULL");
        if (dim == 0 || dim != c->dim) return fail(PQV_ERR_INVALID, "list length does not match the corpus");
        uint32_t def_bw = 0;
        for (uint32_t v = max_def; v; v >>= 1) ++def_bw;
        const uint64_t esz = f64 ? 8 : 4;
        for (uint32_t i = 0; i < n_pages; ++i) {
            const uint8_t *b = file_base + body_off[i];
            const uint64_t blen = body_len[i], nv = n_values[i];
            uint64_t p = 0;
            bool good = nv > 0 && nv % dim == 0 && first_value[i] % dim == 0;
            for (int run = 0; run < 2 && good; ++run) {
                if (p + 4 > blen) { good = false; break; }
                uint32_t n;
                std::memcpy(&n, b + p, 4);
                if (p + 4 + n > blen) { good = false; break; }
                good = levels_plain(b + p + 4, n, run == 0 ? 1u : def_bw, nv, run == 0 ? 1 : 0, run == 0 ? dim : max_def) == 0;
                p += 4 + static_cast<uint64_t>(n);
            }
            if (good && blen - p != nv * esz) good = false;
            if (!good) { if (bad_page) *bad_page = i; return 1; }
            const uint64_t row = first_value[i] / dim, rows = nv / dim;
            int rc = f64 ? corpus_write_rows<double>(c, row, reinterpret_cast<const double *>(b + p), rows)
                         : corpus_write_rows<float>(c, row, reinterpret_cast<const float *>(b + p), rows);
            if (rc != PQV_OK) return rc;
        }
        return PQV_OK;
    });
}

// ---- Thrift compact protocol, as far as a Parquet PageHeader needs it ----
namespace {
struct TcReader {
    const uint8_t *p, *end;
    bool ok = true;
    uint64_t varint() {
        uint64_t v = 0;
        for (int sh = 0; sh < 70; sh += 7) {
            if (p >= end) { ok = false; return 0; }
            const uint8_t b = *p++;
            v |= static_cast<uint64_t>(b & 0x7F) << sh;
            if (!(b & 0x80)) return v;
        }
        ok = false;
        return 0;
    }
    int64_t zigzag() { const uint64_t v = varint(); return static_cast<int64_t>(v >> 1) ^ -static_cast<int64_t>(v & 1); }
    void skip_bytes(uint64_t n) { if (n > static_cast<uint64_t>(end - p)) { ok = false; p = end; } else p += n; }
    void skip(int type, int depth) {
        if (!ok || depth > 16) { ok = false; return; }
        switch (type) {
            case 1: case 2: return;                       // bool in the field header
            case 3: skip_bytes(1); return;
            case 4: case 5: case 6: (void)varint(); return;
            case 7: skip_bytes(8); return;
            case 8: skip_bytes(varint()); return;
            case 9: case 10: {
                if (p >= end) { ok = false; return; }
                const uint8_t h = *p++;
                uint64_t n = h >> 4;
                const int et = h & 15;
                if (n == 15) n = varint();
                for (uint64_t i = 0; i < n && ok; ++i) { if (et == 1 || et == 2) skip_bytes(1); else skip(et, depth + 1); }
                return;
            }
            case 11: {
                const uint64_t n = varint();
                if (n == 0) return;
                if (p >= end) { ok = false; return; }
                const uint8_t kv = *p++;
                for (uint64_t i = 0; i < n && ok; ++i) {
                    const int kt = kv >> 4, vt = kv & 15;
                    if (kt == 1 || kt == 2) skip_bytes(1); else skip(kt, depth + 1);
                    if (vt == 1 || vt == 2) skip_bytes(1); else skip(vt, depth + 1);
                }
                return;
            }
            case 12: skip_struct(depth + 1); return;
            default: ok = false; return;
        }
    }
    // fields of a struct: f(field id, type) consumes the value and returns true, or returns false to have it skipped
    template <class F> void fields(int depth, F f) {
        int16_t last = 0;
        while (ok) {
            if (p >= end) { ok = false; return; }
            const uint8_t h = *p++;
            if (h == 0) return;
            const int type = h & 15, delta = h >> 4;
            int16_t id = delta ? static_cast<int16_t>(last + delta) : static_cast<int16_t>(zigzag());
            last = id;
            if (!f(id, type)) skip(type, depth);
        }
    }
    void skip_struct(int depth) { fields(depth, [](int16_t, int) { return false; }); }
};
}  // namespace

// Page headers of one column chunk (format/PageHeader of parquet.thrift), one after the other from `buf`: per page 8 ints
// {type, header bytes, compressed_page_size, uncompressed_page_size, num_values, encoding, definition_level_encoding,
// repetition_level_encoding} (the last four from data_page_header / dictionary_page_header; -1 where absent; v2 data pages and
// index pages report their type only).  Stops at `len`, after `max_pages`, or once `stop_values` values have been seen in data
// pages; *n_pages = pages written.  An error for a header that does not parse or a page that runs past `len`.
extern "C" int pqv_parquet_page_headers(const uint8_t *buf, uint64_t len, uint64_t stop_values, uint32_t max_pages, int32_t *out, uint32_t *n_pages) {
    return guard([&]() -> int {
        if (!buf || !out || !n_pages) return fail(PQV_ERR_INVALID, "buf, out and n_pages must not be NULL");
        uint64_t pos = 0, seen = 0;
        uint32_t n = 0;
        while (pos < len && n < max_pages && (stop_values == 0 || seen < stop_values)) {
            TcReader r{buf + pos, buf + len};
            int32_t *o = out + static_cast<size_t>(n) * 8;
            for (int i = 0; i < 8; ++i) o[i] = -1;
            r.fields(0, [&](int16_t id, int type) {
                const bool i32 = type == 4 || type == 5 || type == 6;
                if (id == 1 && i32) { o[0] = static_cast<int32_t>(r.zigzag()); return true; }
                if (id == 2 && i32) { o[3] = static_cast<int32_t>(r.zigzag()); return true; }
                if (id == 3 && i32) { o[2] = static_cast<int32_t>(r.zigzag()); return true; }
                if ((id == 5 || id == 7) && type == 12) {
                    r.fields(1, [&](int16_t id2, int type2) {
                        const bool j32 = type2 == 4 || type2 == 5 || type2 == 6;
                        if (id2 >= 1 && id2 <= (id == 5 ? 4 : 2) && j32) { o[3 + id2] = static_cast<int32_t>(r.zigzag()); return true; }
                        return false;
                    });
                    return true;
                }
                return false;
            });
            if (!r.ok) return fail(PQV_ERR_INVALID, "malformed Parquet page header");
            const uint64_t hlen = static_cast<uint64_t>(r.p - (buf + pos));
            o[1] = static_cast<int32_t>(hlen);
            if (o[0] < 0 || o[2] < 0 || o[3] < 0 || hlen + static_cast<uint64_t>(o[2]) > len - pos)
                return fail(PQV_ERR_INVALID, "Parquet page runs past its column chunk");
            if (o[0] == 0 && o[4] > 0) seen += static_cast<uint64_t>(o[4]);
            pos += hlen + static_cast<uint64_t>(o[2]);
            ++n;
        }
        *n_pages = n;
        return PQV_OK;
    });
}

extern "C" int pqv_parquet_dict_decode(const uint8_t *buf, uint64_t len, const void *dict, uint64_t dict_n, uint32_t elem_size,
                                       uint64_t n_values, void *out) {
    return guard([&]() -> int {
        if (n_values == 0) return 0;
        if (!buf || !dict || !out || len < 1 || (elem_size != 4 && elem_size != 8)) return fail(PQV_ERR_INVALID, "bad arguments");
        const uint32_t bw = buf[0];
        uint64_t at = 0;
        bool in_range = true;
        auto put = [&](uint32_t idx, uint64_t cnt) {
            if (idx >= dict_n) { in_range = false; return false; }
            if (elem_size == 4) { const uint32_t v = static_cast<const uint32_t *>(dict)[idx]; uint32_t *o = static_cast<uint32_t *>(out) + at; for (uint64_t i = 0; i < cnt; ++i) o[i] = v; }
            else { const uint64_t v = static_cast<const uint64_t *>(dict)[idx]; uint64_t *o = static_cast<uint64_t *>(out) + at; for (uint64_t i = 0; i < cnt; ++i) o[i] = v; }
            at += cnt;
            return true;
        };
        if (bw == 0) { put(0, n_values); return in_range ? 0 : fail(PQV_ERR_INVALID, "dictionary index out of range"); }
        const bool well = hybrid_runs(buf + 1, len - 1, bw, n_values, put);
        if (!well || !in_range || at != n_values) return fail(PQV_ERR_INVALID, "malformed dictionary-encoded page");
        return 0;
    });
}

static int pqv_corpus_upload_impl(int device, const float *rows, uint64_t n, uint32_t dim,
                                 pqv_corpus **out) {
    if (int rc = pqv_corpus_create(device, n, dim, out)) return rc;
    if (int rc = pqv_corpus_append(*out, rows, n)) {
        delete *out;
        *out = nullptr;
        return rc;
    }
    return PQV_OK;
}
extern "C" int pqv_corpus_upload(int device, const float *rows, uint64_t n, uint32_t dim,
                                 pqv_corpus **out) {
    return guard([&] { return pqv_corpus_upload_impl(device, rows, n, dim, out); });
}

static int pqv_corpus_from_device_impl(int device, const void *d_rows, uint64_t n, uint32_t dim,
                                      pqv_corpus **out) {
    if (!out) return fail(PQV_ERR_INVALID, "out must not be NULL");
    *out = nullptr;
    if (dim == 0) return fail(PQV_ERR_INVALID, "Embedding dimension must be > 0");
    if (n > 0xFFFFFFFFull)
        return fail(PQV_ERR_UNSUPPORTED, "row ids are u32: at most 4294967295 rows per corpus");
    if (n && !d_rows) return fail(PQV_ERR_INVALID, "d_rows must not be NULL");
    if (int rc = use_device(device)) return rc;
    pqv_corpus *c = new (std::nothrow) pqv_corpus();
    if (!c) return fail(PQV_ERR_OOM, "host allocation failed");
    c->device = device; c->dim = dim; c->n = n; c->capacity = n;
    c->d_rows = const_cast<float *>(static_cast<const float *>(d_rows));
    c->owned = false;
    hipError_t e = hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking);
    if (e != hipSuccess) {
        delete c;
        return fail(PQV_ERR_HIP, std::string("hipStreamCreate: ") + hipGetErrorString(e));
    }
    (void)pqv::touch_build(c->stream);            // (the stream's hardware queue is set up by its first launch: here, not in the build)
    (void)hipStreamSynchronize(c->stream);
    *out = c;
    return PQV_OK;
}
extern "C" int pqv_corpus_from_device(int device, const void *d_rows, uint64_t n, uint32_t dim,
                                      pqv_corpus **out) {
    return guard([&] { return pqv_corpus_from_device_impl(device, d_rows, n, dim, out); });
}

extern "C" uint64_t pqv_corpus_rows(const pqv_corpus *c) { return c ? c->n : 0; }
extern "C" uint32_t pqv_corpus_dim(const pqv_corpus *c) { return c ? c->dim : 0; }
extern "C" int pqv_corpus_device(const pqv_corpus *c) { return c ? c->device : -1; }

static int pqv_corpus_fetch_rows_impl(const pqv_corpus *c, const uint32_t *rows, uint64_t m,
                                     float *out) {
    if (!c) return fail(PQV_ERR_INVALID, "corpus must not be NULL");
    if (m == 0) return PQV_OK;
    if (!rows || !out) return fail(PQV_ERR_INVALID, "rows/out must not be NULL");
    if (!c->d_rows) return fail(PQV_ERR_INVALID, "corpus row-order copy was released");
    for (uint64_t i = 0; i < m; ++i)
        if (rows[i] >= c->n) return fail(PQV_ERR_INVALID, "row id out of range");
    if (int rc = use_device(c->device)) return rc;
    DevBuf d_idx, d_out;
    HIP_TRY(d_idx.alloc(m * sizeof(uint32_t)));
    HIP_TRY(d_out.alloc(m * c->dim * sizeof(float)));
    HIP_TRY(hipMemcpy(d_idx.p, rows, m * sizeof(uint32_t), hipMemcpyHostToDevice));
    HIP_TRY(pqv::launch_gather_rows(c->d_rows, d_idx.as<uint32_t>(), nullptr, m, c->dim,
                                    d_out.as<float>(), c->stream));
    HIP_TRY(hipStreamSynchronize(c->stream));
    HIP_TRY(hipMemcpy(out, d_out.p, m * c->dim * sizeof(float), hipMemcpyDeviceToHost));
    return PQV_OK;
}
extern "C" int pqv_corpus_fetch_rows(const pqv_corpus *c, const uint32_t *rows, uint64_t m,
                                     float *out) {
    return guard([&] { return pqv_corpus_fetch_rows_impl(c, rows, m, out); });
}

extern "C" void pqv_corpus_free(pqv_corpus *c) {
    if (!c) return;
    (void)hipSetDevice(c->device);
    delete c;
}

// ---------------------------------------------------------------------------------------
// index blob + accessors (src/ivf/index.rs:65-128)
// ---------------------------------------------------------------------------------------
namespace {
inline uint32_t rd_u32(const uint8_t *p) {
    return static_cast<uint32_t>(p[0]) | (static_cast<uint32_t>(p[1]) << 8) |
           (static_cast<uint32_t>(p[2]) << 16) | (static_cast<uint32_t>(p[3]) << 24);
}
inline void wr_u32(uint8_t *p, uint32_t v) {
    p[0] = static_cast<uint8_t>(v); p[1] = static_cast<uint8_t>(v >> 8);
    p[2] = static_cast<uint8_t>(v >> 16); p[3] = static_cast<uint8_t>(v >> 24);
}
}  // namespace

static int pqv_index_from_bytes_impl(const uint8_t *bytes, size_t len, pqv_index **out) {
    if (!out) return fail(PQV_ERR_INVALID, "out must not be NULL");
    *out = nullptr;
    if (!bytes || len < 8) return fail(PQV_ERR_FORMAT, "IVF index buffer too small");  // index.rs:89
    const uint32_t dim = rd_u32(bytes), k = rd_u32(bytes + 4);
    if (dim == 0) return fail(PQV_ERR_INVALID, "Embedding dimension must be > 0");     // mod.rs:59
    if (k == 0) return fail(PQV_ERR_INVALID, "Cluster count must be > 0");             // index.rs:24
    size_t off = 8;
    const uint64_t clen = static_cast<uint64_t>(k) * dim;
    if ((len - off) / 4 < clen) return fail(PQV_ERR_FORMAT, "IVF index buffer truncated (centroids)");
    pqv_index *idx = new (std::nothrow) pqv_index();
    if (!idx) return fail(PQV_ERR_OOM, "host allocation failed");
    idx->dim = dim; idx->n_clusters = k;
    idx->centroids.resize(clen);
    for (uint64_t i = 0; i < clen; ++i, off += 4) {
        const uint32_t bits = rd_u32(bytes + off);
        std::memcpy(&idx->centroids[i], &bits, 4);
    }
    idx->list_off.assign(static_cast<size_t>(k) + 1, 0);
    size_t scan = off;
    for (uint32_t c = 0; c < k; ++c) {
        if (len - scan < 4) { delete idx; return fail(PQV_ERR_FORMAT, "IVF index buffer truncated (list length)"); }
        const uint32_t ll = rd_u32(bytes + scan);
        scan += 4;
        if ((len - scan) / 4 < ll) { delete idx; return fail(PQV_ERR_FORMAT, "IVF index buffer truncated (list rows)"); }
        scan += static_cast<size_t>(ll) * 4;
        idx->list_off[c + 1] = idx->list_off[c] + ll;
    }
    idx->list_rows.resize(idx->list_off[k]);
    for (uint32_t c = 0; c < k; ++c) {
        const uint32_t ll = rd_u32(bytes + off);
        off += 4;
        uint32_t *dst = idx->list_rows.data() + idx->list_off[c];
        for (uint32_t i = 0; i < ll; ++i, off += 4) dst[i] = rd_u32(bytes + off);
    }
    *out = idx;
    return PQV_OK;
}
extern "C" int pqv_index_from_bytes(const uint8_t *bytes, size_t len, pqv_index **out) {
    return guard([&] { return pqv_index_from_bytes_impl(bytes, len, out); });
}

static int pqv_index_to_bytes_impl(const pqv_index *idx, uint8_t **buf, size_t *len) {
    if (!idx || !buf || !len) return fail(PQV_ERR_INVALID, "index/buf/len must not be NULL");
    const uint64_t k = idx->n_clusters;
    const std::vector<uint32_t> *rows_p = idx->rows->get();
    if (!rows_p) return PQV_ERR_HIP;
    const std::vector<uint32_t> &rows_v = *rows_p;
    const size_t sz = 8 + idx->centroids.size() * 4 + static_cast<size_t>(k) * 4 + rows_v.size() * 4;
    uint8_t *b = static_cast<uint8_t *>(std::malloc(sz));
    if (!b) return fail(PQV_ERR_OOM, "host allocation failed");
    size_t off = 0;
    wr_u32(b, idx->dim); wr_u32(b + 4, idx->n_clusters); off = 8;
    for (float v : idx->centroids) {
        uint32_t bits;
        std::memcpy(&bits, &v, 4);
        wr_u32(b + off, bits);
        off += 4;
    }
    for (uint64_t c = 0; c < k; ++c) {
        const uint64_t s = idx->list_off[c], e = idx->list_off[c + 1];
        wr_u32(b + off, static_cast<uint32_t>(e - s));
        off += 4;
        for (uint64_t i = s; i < e; ++i, off += 4) wr_u32(b + off, rows_v[i]);
    }
    *buf = b;
    *len = sz;
    return PQV_OK;
}
extern "C" int pqv_index_to_bytes(const pqv_index *idx, uint8_t **buf, size_t *len) {
    return guard([&] { return pqv_index_to_bytes_impl(idx, buf, len); });
}

extern "C" void pqv_bytes_free(uint8_t *buf) { std::free(buf); }

static int pqv_index_from_parts_impl(uint32_t dim, uint32_t n_clusters, const float *centroids,
                                    const uint64_t *list_off, const uint32_t *list_rows,
                                    pqv_index **out) {
    if (!out) return fail(PQV_ERR_INVALID, "out must not be NULL");
    *out = nullptr;
    if (dim == 0) return fail(PQV_ERR_INVALID, "Embedding dimension must be > 0");
    if (n_clusters == 0) return fail(PQV_ERR_INVALID, "Cluster count must be > 0");
    if (!centroids || !list_off) return fail(PQV_ERR_INVALID, "centroids/list_off must not be NULL");
    for (uint32_t c = 0; c < n_clusters; ++c)
        if (list_off[c + 1] < list_off[c]) return fail(PQV_ERR_INVALID, "list_off must be non-decreasing");
    if (list_off[0] != 0) return fail(PQV_ERR_INVALID, "list_off[0] must be 0");
    const uint64_t total = list_off[n_clusters];
    if (total && !list_rows) return fail(PQV_ERR_INVALID, "list_rows must not be NULL");
    pqv_index *idx = new (std::nothrow) pqv_index();
    if (!idx) return fail(PQV_ERR_OOM, "host allocation failed");
    idx->dim = dim; idx->n_clusters = n_clusters;
    idx->centroids.assign(centroids, centroids + static_cast<uint64_t>(n_clusters) * dim);
    idx->list_off.assign(list_off, list_off + n_clusters + 1);
    idx->list_rows.assign(list_rows, list_rows + total);
    *out = idx;
    return PQV_OK;
}
extern "C" int pqv_index_from_parts(uint32_t dim, uint32_t n_clusters, const float *centroids,
                                    const uint64_t *list_off, const uint32_t *list_rows,
                                    pqv_index **out) {
    return guard([&] { return pqv_index_from_parts_impl(dim, n_clusters, centroids, list_off, list_rows, out); });
}

extern "C" uint32_t pqv_index_dim(const pqv_index *i) { return i ? i->dim : 0; }
extern "C" uint32_t pqv_index_n_clusters(const pqv_index *i) { return i ? i->n_clusters : 0; }
extern "C" uint64_t pqv_index_n_rows(const pqv_index *i) { return i ? i->n_rows() : 0; }
extern "C" const float *pqv_index_centroids(const pqv_index *i) { return i ? i->centroids.data() : nullptr; }
extern "C" const uint64_t *pqv_index_list_offsets(const pqv_index *i) { return i ? i->list_off.data() : nullptr; }
extern "C" const uint32_t *pqv_index_list_rows(const pqv_index *i) {
    if (!i) return nullptr;
    const std::vector<uint32_t> *v = i->rows->get();         // (a build's lists are downloaded by the first reader)
    return v ? v->data() : nullptr;
}
extern "C" void pqv_index_free(pqv_index *i) { delete i; }

// ---------------------------------------------------------------------------------------
// k-means on the device (src/ivf/index.rs:323-457)
// ---------------------------------------------------------------------------------------
namespace {

// Stable counting sort of rows by cluster == the reference's "ascending row ids per
// cluster" (index.rs:193-206).
// false if an assignment is out of range (a kernel bug must surface as an error, not as host memory corruption)
bool lists_from_assignment(const uint32_t *cluster_of, uint64_t n, uint32_t k,
                           std::vector<uint64_t> &off, std::vector<uint32_t> &rows) {
    // Large inputs: a stable counting sort over T contiguous row ranges on T host threads.  Range t's rows of a cluster
    // follow those of the ranges before it, so every list is in ascending row order -- the sequential scan's result.
    // PQV_LIST_THREADS=1 keeps the sequential scan (A/B)
    static const uint32_t t_max = [] { const char *e = std::getenv("PQV_LIST_THREADS"); const long v = e ? std::strtol(e, nullptr, 10) : 8;
                                       return static_cast<uint32_t>(std::min<long>(64, std::max<long>(1, v))); }();
    const uint32_t T = n >= (1u << 20) ? std::min<uint32_t>(t_max, std::max<uint32_t>(1, host_workers())) : 1;
    if (T > 1) {
        try {
            std::vector<uint64_t> cnt(static_cast<size_t>(T) * k, 0);
            std::atomic<bool> bad{false};
            auto run_ranges = [&](auto &&body) {
                std::vector<std::thread> th;
                th.reserve(T);
                for (uint32_t t = 0; t < T; ++t) th.emplace_back([&, t] { body(t, n * t / T, n * (t + 1) / T); });
                for (auto &x : th) x.join();
            };
            run_ranges([&](uint32_t t, uint64_t lo, uint64_t hi) {
                uint64_t *c = cnt.data() + static_cast<size_t>(t) * k;
                for (uint64_t r = lo; r < hi; ++r) {
                    const uint32_t v = cluster_of[r];
                    if (v >= k) { bad.store(true); return; }
                    c[v]++;
                }
            });
            if (bad.load()) return false;
            off.assign(static_cast<size_t>(k) + 1, 0);
            uint64_t run = 0;
            for (uint32_t c = 0; c < k; ++c) {
                off[c] = run;
                for (uint32_t t = 0; t < T; ++t) {
                    uint64_t &slot = cnt[static_cast<size_t>(t) * k + c];
                    const uint64_t x = slot;
                    slot = run;
                    run += x;
                }
            }
            off[k] = run;
            rows.resize(n);
            uint32_t *out = rows.data();
            run_ranges([&](uint32_t t, uint64_t lo, uint64_t hi) {
                uint64_t *cur = cnt.data() + static_cast<size_t>(t) * k;
                for (uint64_t r = lo; r < hi; ++r) out[cur[cluster_of[r]]++] = static_cast<uint32_t>(r);
            });
            return true;
        } catch (const std::system_error &) {
            // no threads to be had: the sequential scan below
        }
    }
    for (uint64_t r = 0; r < n; ++r) if (cluster_of[r] >= k) return false;
    off.assign(static_cast<size_t>(k) + 1, 0);
    for (uint64_t r = 0; r < n; ++r) off[cluster_of[r] + 1]++;
    for (uint32_t c = 0; c < k; ++c) off[c + 1] += off[c];
    rows.resize(n);
    std::vector<uint64_t> cur(off.begin(), off.end() - 1);
    for (uint64_t r = 0; r < n; ++r) rows[cur[cluster_of[r]]++] = static_cast<uint32_t>(r);
    return true;
}

// ---------------------------------------------------------------------------------------
// MFMA-screened assignment (index.rs:395-430 Lloyd assign, :189-206 + :244-257 final assignment).
// cluster[r] = argmin_j d2(x_r, c_j), strict '<' in ascending j, is the top-1 of row r among the
// centroids under the key (d2 bits, j) -- exactly what the wide screened search computes with the
// rows as queries and one list holding all centroids: thresholds from MFMA upper bounds of the first
// centroids (wide_seed_kernel), MFMA lower-bound screen of all of them, exact re-evaluation of the few
// survivors in the reference's summation order (wide_filter_kernel), top-1 merge.  The exact VALU
// kernel (assign_kernel) remains for dim % 64 != 0, for fewer than 512 centroids and for rows or
// centroids with a non-finite norm (NaN ordering is the reference's `<`, not the key order).
// ---------------------------------------------------------------------------------------
struct ScreenedAssign {
    uint32_t dim = 0, kc = 0, width = 0, chunk_q = 0, rpb = 0, bpl = 0, max_quads = 0, ccap = 32;
    const float *d_centroids = nullptr;
    // f16 operands (PQV_ASSIGN_F16=1; dim % 128 == 0, dim <= 1024): the 8-wave kernel with the chunk's rows staged in LDS
    bool f16 = false;
    float f16_scale = 1.0f;
    DevBuf qmax, dmax;
    DevBuf cblk, cnorm, list_off, blk_off, flag;
    DevBuf qnorm, pairs, quads, nq_u32, cand_base, gthr, part_keys, part_vals, cand_keys, cand_vals, cand_cnt, spilled,
        seed_ub, qblk, dist_out, nfound;

    static bool applicable(uint32_t dim, uint32_t kc) {
        // PQV_ASSIGN_SCREEN=0 disables, =N (N > 1) sets the smallest centroid count that takes this path
        const char *e = std::getenv("PQV_ASSIGN_SCREEN");
        const long v = e ? std::strtol(e, nullptr, 10) : 512;
        return v != 0 && (dim % 64) == 0 && kc >= static_cast<uint32_t>(v > 1 ? v : 512);
    }
    // (re)load the centroids: blocked copy + norms; returns 1 in *nonfinite if a centroid norm is inf / NaN
    int set_centroids(const float *d_c, uint32_t k, uint32_t d, hipStream_t stream, bool *nonfinite) {
        using namespace pqv;
        dim = d; kc = k; d_centroids = d_c;
        static const uint32_t wide_env = [] { const char *e = std::getenv("PQV_ASSIGN_WIDTH"); return e ? static_cast<uint32_t>(std::strtoul(e, nullptr, 10)) : 0u; }();
        width = wide_env == 32 && dim > 128 ? 32 : 64;     // 64-row quads halve the re-streaming of the centroids
        static const bool f16_env = [] { const char *e = std::getenv("PQV_ASSIGN_F16"); return e && *e == '1'; }();
        f16 = f16_env && (dim % 128) == 0 && dim <= 1024;
        if (f16) width = static_cast<uint32_t>(std::min<uint64_t>(128, 147456ull / (static_cast<uint64_t>(dim) * (dim <= 128 ? 6 : 2)) / 32 * 32));
        chunk_q = 65536;          // (131072 / 262144 rows per chunk measured the same build time: the per-chunk launches are not what bounds it)
        rpb = f16 ? 2048 : 1024; bpl = (kc + rpb - 1) / rpb;
        max_quads = (chunk_q / width + 7) / 8 * 8;
        const uint64_t tiles = (static_cast<uint64_t>(kc) + 15) / 16;
        const uint64_t h_off[2] = {0, kc}, h_blk[2] = {0, tiles};
        HIP_TRY(list_off.ensure(sizeof h_off)); HIP_TRY(blk_off.ensure(sizeof h_blk)); HIP_TRY(flag.ensure(sizeof(uint32_t)));
        HIP_TRY(hipMemcpyAsync(list_off.p, h_off, sizeof h_off, hipMemcpyHostToDevice, stream));
        HIP_TRY(hipMemcpyAsync(blk_off.p, h_blk, sizeof h_blk, hipMemcpyHostToDevice, stream));
        HIP_TRY(hipStreamSynchronize(stream));          // the two host arrays above are stack-allocated
        HIP_TRY(cblk.ensure(tiles * 16 * dim * sizeof(float)));
        HIP_TRY(cnorm.ensure(static_cast<size_t>(kc) * sizeof(float)));
        HIP_TRY(launch_row_norms(d_c, kc, dim, 1, cnorm.as<float>(), stream));
        if (int rc = check_finite(cnorm.as<float>(), kc, stream, nonfinite)) return rc;
        if (f16 && !*nonfinite) {
            // scale: the centroids' maximum lands below 2^14 (rows beyond the f16 range are never skipped by the kernel)
            uint32_t bits = 0;
            HIP_TRY(dmax.ensure(sizeof(uint32_t)));
            HIP_TRY(hipMemsetAsync(dmax.p, 0, sizeof(uint32_t), stream));
            HIP_TRY(launch_maxabs(d_c, static_cast<uint64_t>(kc) * dim, dmax.as<uint32_t>(), stream));
            HIP_TRY(hipMemcpyAsync(&bits, dmax.p, sizeof bits, hipMemcpyDeviceToHost, stream));
            HIP_TRY(hipStreamSynchronize(stream));
            float m; std::memcpy(&m, &bits, sizeof m);
            int e = 0;
            if (m > 0.0f) (void)std::frexp(m, &e);
            const int se = m > 0.0f ? 14 - e : 0;
            if (se < -60 || se > 60) f16 = false; else f16_scale = std::ldexp(1.0f, se);
        }
        if (f16 && !*nonfinite)
            HIP_TRY(launch_block_rows_f16(d_c, list_off.as<uint64_t>(), blk_off.as<uint64_t>(), 1, tiles, dim, f16_scale, cblk.p, stream));
        else
            HIP_TRY(launch_block_rows(d_c, list_off.as<uint64_t>(), blk_off.as<uint64_t>(), 1, tiles, dim, cblk.p, stream));
        return PQV_OK;
    }
    int check_finite(const float *v, uint64_t n, hipStream_t stream, bool *nonfinite) {
        uint32_t h = 0;
        HIP_TRY(hipMemsetAsync(flag.p, 0, sizeof(uint32_t), stream));
        HIP_TRY(pqv::launch_nonfinite_flag(v, n, flag.as<uint32_t>(), stream));
        HIP_TRY(hipMemcpyAsync(&h, flag.p, sizeof h, hipMemcpyDeviceToHost, stream));
        HIP_TRY(hipStreamSynchronize(stream));
        *nonfinite = h != 0;
        return PQV_OK;
    }
    // cluster[0 .. n) for rows d_rows[0 .. n); *fallback = true if a row norm is not finite (nothing written)
    int run(const float *d_rows, uint64_t n, uint32_t *d_cluster, hipStream_t stream, bool *fallback) {
        using namespace pqv;
        *fallback = false;
        HIP_TRY(qnorm.ensure(static_cast<size_t>(n) * sizeof(float)));
        HIP_TRY(launch_row_norms(d_rows, n, dim, 1, qnorm.as<float>(), stream));
        bool bad = false;
        if (int rc = check_finite(qnorm.as<float>(), n, stream, &bad)) return rc;
        if (bad) { *fallback = true; return PQV_OK; }
        static const uint32_t seed_env = [] { const char *e = std::getenv("PQV_ASSIGN_SEED"); return e ? static_cast<uint32_t>(std::strtoul(e, nullptr, 10)) : 256u; }();
        const uint32_t k = 1, slots = (f16 ? 8u : 4u) * bpl, seed_rows = std::max<uint32_t>(64, seed_env / 64 * 64), seed_sw = 4;
        HIP_TRY(pairs.ensure(static_cast<size_t>(chunk_q) * 4)); HIP_TRY(quads.ensure(static_cast<size_t>(max_quads) * sizeof(uint4)));
        HIP_TRY(nq_u32.ensure(16)); HIP_TRY(cand_base.ensure(static_cast<size_t>(chunk_q) * 8));
        HIP_TRY(gthr.ensure(static_cast<size_t>(chunk_q) * 8));
        const uint64_t entries = (static_cast<uint64_t>(chunk_q) * slots * k + 3) / 4 * 4;
        HIP_TRY(part_keys.ensure((entries + 4) * 8)); HIP_TRY(part_vals.ensure((entries + 4) * 4));
        HIP_TRY(cand_keys.ensure(static_cast<size_t>(chunk_q) * ccap * 8)); HIP_TRY(cand_vals.ensure(static_cast<size_t>(chunk_q) * ccap * 4));
        HIP_TRY(cand_cnt.ensure(static_cast<size_t>(chunk_q) * 4)); HIP_TRY(spilled.ensure(static_cast<size_t>(chunk_q) * 4));
        HIP_TRY(seed_ub.ensure(static_cast<size_t>(chunk_q) * seed_sw * 16 * 4));
        HIP_TRY(dist_out.ensure(static_cast<size_t>(chunk_q) * 4)); HIP_TRY(nfound.ensure(static_cast<size_t>(chunk_q) * 4));
        const bool qlds = f16 || static_cast<uint64_t>(width) * dim * sizeof(float) <= 32768;
        if (!qlds) HIP_TRY(qblk.ensure(static_cast<size_t>(max_quads) * width * dim * sizeof(float)));
        if (f16) {
            HIP_TRY(qmax.ensure(static_cast<size_t>(n) * sizeof(float)));
            HIP_TRY(launch_row_norms(d_rows, n, dim, 2, qmax.as<float>(), stream));
        }
        for (uint64_t r0 = 0; r0 < n; r0 += chunk_q) {
            const uint32_t nq = static_cast<uint32_t>(std::min<uint64_t>(chunk_q, n - r0));
            const float *q = d_rows + r0 * dim;
            HIP_TRY(launch_assign_setup(pairs.as<uint32_t>(), quads.as<uint4>(), nq_u32.as<uint32_t>(), cand_base.as<uint64_t>(),
                                        gthr.as<unsigned long long>(), nq, width, stream));
            HIP_TRY(launch_fill_ones2(part_keys.p, entries * 8, part_vals.p, entries * 4, stream));
            TileArgs ta{};
            ta.mat = d_centroids; ta.row_of = nullptr; ta.list_off = list_off.as<uint64_t>();
            ta.queries = q; ta.cand_base = cand_base.as<uint64_t>(); ta.pairs = pairs.as<uint32_t>();
            ta.quads = quads.as<uint4>(); ta.n_quads = nq_u32.as<uint32_t>(); ta.max_quads = max_quads; ta.quad_width = width;
            ta.max_groups = max_quads;          // only its being non-zero matters to the launcher
            ta.nq = nq; ta.nprobe = 1; ta.dim = dim; ta.k = k;
            ta.rows_per_block = rpb; ta.blocks_per_list = bpl; ta.max_pos = ~0ull;
            ta.slots_per_pair = slots; ta.slot_base = 0; ta.n_part = slots;
            ta.gthr = gthr.as<unsigned long long>();
            ta.part_keys = part_keys.as<uint64_t>(); ta.part_vals = part_vals.as<uint32_t>();
            ta.mat_blk = static_cast<const float4 *>(cblk.p); ta.blk_off = blk_off.as<uint64_t>();
            ta.row_norm2 = cnorm.as<float>(); ta.query_norm2 = qnorm.as<float>() + r0;
            ta.cand_keys = cand_keys.as<uint64_t>(); ta.cand_vals = cand_vals.as<uint32_t>();
            ta.cand_cnt = cand_cnt.as<uint32_t>(); ta.cand_cap = ccap; ta.spilled = spilled.as<uint32_t>();
            ta.xcd_swizzle = qlds ? 0 : 1;
            if (f16) {
                ta.f16 = 1; ta.scale = f16_scale; ta.scale2 = f16_scale * f16_scale; ta.block_waves = 8;
                ta.query_maxabs = qmax.as<float>() + r0;
            }
            if (!qlds) {
                HIP_TRY(launch_pack_queries(q, ta.pairs, ta.quads, ta.n_quads, max_quads, 1, dim, width / 16, qblk.p, stream));
                ta.q_blk = static_cast<const float4 *>(qblk.p);
            }
            TileArgs seed = ta;
            seed.row_offset = 0; seed.row_end = seed_rows; seed.rows_per_block = 256; seed.grid_x = 1;
            seed.seed_sw = seed_sw; seed.seed_ub = seed_ub.as<float>();
            HIP_TRY(launch_wide_seed(seed, stream));
            HIP_TRY(launch_seed_select(seed.seed_ub, nq, seed_sw * 16, k, ta.gthr, ta.cand_cnt, ta.spilled, stream));
            ta.row_offset = 0; ta.grid_x = bpl; ta.filter_variant = 0;
            HIP_TRY(launch_tile_filter(ta, stream));
            MergeArgs fm{};
            fm.part_keys = ta.part_keys; fm.part_vals = ta.part_vals;
            fm.nq = nq; fm.n_part = slots; fm.k_part = k; fm.k = k; fm.k_out = k;
            fm.ids = nullptr; fm.row_idx = d_cluster + r0; fm.dist = dist_out.as<float>(); fm.n_found = nfound.as<uint32_t>();
            fm.sqrt_out = 0;
            fm.cand_keys = ta.cand_keys; fm.cand_vals = ta.cand_vals; fm.cand_cnt = ta.cand_cnt; fm.cand_cap = ccap;
            fm.spilled = ta.spilled;
            HIP_TRY(launch_merge_final(fm, stream));
        }
        return PQV_OK;
    }
};

// ---------------------------------------------------------------------------------------
// Round 3: the assignment as a dense f16 contraction + exact re-scoring (kernels_build.hip: assign_f16_kernel,
// assign_rescore_kernel).  Rows and centroids are imaged as unit vectors about a common centre mu (the column mean of
// the centroids: the distance is translation invariant, and |x - mu| |c - mu| is what scales the bound's slack); the
// screen leaves a handful of candidate centroids per row, the exact pass evaluates those in the reference's order and
// takes the argmin by (distance bits, index) -- index.rs:408-415's strict '<' in ascending order.  Needs dim % 4 == 0
// and finite norms (NaN ordering is the reference's `<`, not the key order): otherwise the callers keep their old path.
// ---------------------------------------------------------------------------------------
struct GemmAssign {
    uint32_t dim = 0, dim_p = 0, kc = 0, kc_pad = 0, cap = 32;
    uint64_t chunk = [] { const char *e = std::getenv("PQV_ASSIGN_CHUNK"); const long v = e ? std::atol(e) : 0; return v >= 4096 ? static_cast<uint64_t>(v) : (1ull << 18); }();
    const float *d_centroids = nullptr;
    DevBuf mu, c16, cn2, x16, xn2, cand, cnt, flag, cand_t, best_t, rstats, d_perm, d_grp;
    // round 4: 256 x 256 tiles with the rows on the lane-owned side and centroid images at one global scale (assign_wide_kernel +
    // assign_resolve_kernel); PQV_ASSIGN_TILE=128 keeps round 3's 128 x 256 kernel with its own exact pass (A/B, tests)
    bool wide = true;
    // The Lloyd iterations assign the SAME rows again and again: with the centring vector kept from the first iteration (any fixed
    // vector serves the bounds; the first centroid mean is as good as the current one) the rows' images and norms are made once.
    std::vector<uint32_t> h_perm; std::vector<float> h_grp;
    bool keep_mu = false, mu_set = false;
    const float *img_rows = nullptr; uint64_t img_n = 0; uint32_t img_dim_p = 0;       // what x16 / xn2 currently hold (single-chunk runs)
    float cmaxs = 0.0f, cn_max = 0.0f, kA = 0.0f;
    std::vector<float> h_cn2;
    unsigned long long exact_rows = 0, exact_evals = 0, rows_total = 0;

    static bool applicable(uint32_t dim, uint32_t kc) {
        const char *e = std::getenv("PQV_ASSIGN_GEMM");          // 0 = off, n > 1 = smallest centroid count that takes this path
        const long v = e ? std::strtol(e, nullptr, 10) : 128L;
        return v != 0 && (dim % 4) == 0 && kc >= static_cast<uint32_t>(v > 1 ? v : 128) && static_cast<uint64_t>((dim + 31) / 32 * 32) * 2 * 320 < 0x7FFFFFFFull;
    }
    int finite(const float *v, uint64_t n, hipStream_t stream, bool *bad) {
        uint32_t h = 0;
        HIP_TRY(flag.ensure(sizeof(uint32_t)));
        HIP_TRY(hipMemsetAsync(flag.p, 0, sizeof(uint32_t), stream));
        HIP_TRY(pqv::launch_nonfinite_flag(v, n, flag.as<uint32_t>(), stream));
        HIP_TRY(hipMemcpyAsync(&h, flag.p, sizeof h, hipMemcpyDeviceToHost, stream));
        HIP_TRY(hipStreamSynchronize(stream));
        *bad = h != 0;
        return PQV_OK;
    }
    int set_centroids(const float *d_c, uint32_t k, uint32_t d, hipStream_t stream, bool *nonfinite) {
        using namespace pqv;
        {
            const char *e = std::getenv("PQV_ASSIGN_TILE");
            wide = !(e && std::atoi(e) == 128) && static_cast<uint64_t>((d + 63) / 64 * 64) * 2 * 512 < 0x7FFFFFFFull;
        }
        dim = d; dim_p = wide ? (d + 63) / 64 * 64 : (d + 31) / 32 * 32; kc = k; kc_pad = (k + 255) / 256 * 256; d_centroids = d_c;
        HIP_TRY(mu.ensure(static_cast<size_t>(dim) * sizeof(float)));
        HIP_TRY(c16.ensure(static_cast<size_t>(kc_pad) * dim_p * sizeof(uint16_t)));
        HIP_TRY(cn2.ensure(static_cast<size_t>(kc_pad) * sizeof(float)));
        if (!(keep_mu && mu_set)) { HIP_TRY(launch_col_mean(d_c, kc, dim, mu.as<float>(), stream)); mu_set = true; img_rows = nullptr; }
        HIP_TRY(launch_center_normalize_f16(d_c, mu.as<float>(), kc, kc_pad, dim, dim_p, cn2.as<float>(), c16.p, stream));
        if (!wide) return finite(cn2.as<float>(), kc, stream, nonfinite);
        // one global scale for the centroid images: 2^8 / max |c - mu| (the norms come back anyway: this is the call's one host check)
        h_cn2.resize(kc);
        HIP_TRY(hipMemcpyAsync(h_cn2.data(), cn2.p, static_cast<size_t>(kc) * sizeof(float), hipMemcpyDeviceToHost, stream));
        HIP_TRY(hipStreamSynchronize(stream));
        float mx = 0.0f;
        bool bad = false;
        for (float v : h_cn2) { if (!(v <= 3.0e38f)) bad = true; else mx = std::max(mx, v); }
        *nonfinite = bad;
        if (bad) return PQV_OK;
        cn_max = mx * 1.000001f;
        cmaxs = std::sqrt(mx) * 1.000001f;
        const float cscale = cmaxs > 0.0f ? 256.0f / cmaxs : 0.0f;
        if (!(cscale > 0.0f && cscale < 3.0e38f)) { wide = false; return PQV_OK; }      // all centroids at mu / degenerate scale: round 3's kernel (images already written)
        kA = 2.0f / (256.0f * cscale);
        // images in ascending-norm order: a 32-centroid group's own largest norm scales its error bound (a table trained on a
        // sample has a few far-out centroids -- clusters of one or two points -- and the table-wide maximum would loosen every bound)
        std::vector<uint32_t> &perm = h_perm;          // (members: the uploads below need no synchronisation to outlive)
        std::vector<float> &grp = h_grp;
        perm.resize(kc);
        for (uint32_t c = 0; c < kc; ++c) perm[c] = c;
        std::stable_sort(perm.begin(), perm.end(), [&](uint32_t x, uint32_t y) { return h_cn2[x] < h_cn2[y]; });
        const uint32_t ngrp = kc_pad / 32;
        grp.assign(2 * static_cast<size_t>(ngrp), 0.0f);
        for (uint32_t g = 0; g < ngrp; ++g) {
            float gm = 0.0f;
            for (uint32_t s = g * 32; s < std::min(kc, (g + 1) * 32); ++s) gm = std::max(gm, h_cn2[perm[s]]);
            grp[g] = std::sqrt(gm) * 1.000001f; grp[ngrp + g] = gm * 1.000001f;
        }
        HIP_TRY(d_perm.ensure(static_cast<size_t>(kc) * sizeof(uint32_t)));
        HIP_TRY(d_grp.ensure(grp.size() * sizeof(float)));
        HIP_TRY(hipMemcpyAsync(d_perm.p, perm.data(), static_cast<size_t>(kc) * sizeof(uint32_t), hipMemcpyHostToDevice, stream));
        HIP_TRY(hipMemcpyAsync(d_grp.p, grp.data(), grp.size() * sizeof(float), hipMemcpyHostToDevice, stream));
        HIP_TRY(launch_center_normalize_f16(d_c, mu.as<float>(), kc, kc_pad, dim, dim_p, cn2.as<float>(), c16.p, stream, cscale, d_perm.as<uint32_t>()));
        return PQV_OK;          // (perm / grp are members: a later call rewrites them only after its own synchronisation on the norms)
    }
    // cluster[0 .. n) for rows d_rows[0 .. n); *fallback = true if a row norm is not finite (the caller re-runs its old path)
    // h_out (optional): the assignment is also copied to this host array, chunk by chunk on a second stream, each copy behind
    // the NEXT chunk's kernels (a pageable destination blocks the calling thread, not the GPU: the copies hide behind the compute)
    int run(const float *d_rows, uint64_t n, uint32_t *d_cluster, hipStream_t stream, bool *fallback, uint32_t *h_out = nullptr) {
        using namespace pqv;
        *fallback = false;
        const double t_run0 = now_s();
        const uint64_t ch = std::min<uint64_t>(chunk, std::max<uint64_t>(1, n));
        struct CopyLane {
            hipStream_t s = nullptr; hipEvent_t ev[2] = {nullptr, nullptr};
            ~CopyLane() { if (s) (void)hipStreamDestroy(s); for (auto e : ev) if (e) (void)hipEventDestroy(e); }
        } cl;
        if (h_out) {
            HIP_TRY(hipStreamCreateWithFlags(&cl.s, hipStreamNonBlocking));
            for (auto &e : cl.ev) HIP_TRY(hipEventCreateWithFlags(&e, hipEventDisableTiming));
        }
        uint64_t pend_r0 = 0, pend_m = 0; int pend_ev = -1, n_chunk = 0;
        auto flush_copy = [&]() -> int {
            if (pend_ev < 0) return PQV_OK;
            HIP_TRY(hipStreamWaitEvent(cl.s, cl.ev[pend_ev], 0));
            HIP_TRY(hipMemcpyAsync(h_out + pend_r0, d_cluster + pend_r0, pend_m * sizeof(uint32_t), hipMemcpyDeviceToHost, cl.s));
            pend_ev = -1;
            return PQV_OK;
        };
        HIP_TRY(x16.ensure(ch * dim_p * sizeof(uint16_t)));
        HIP_TRY(xn2.ensure(ch * sizeof(float)));
        HIP_TRY(cand.ensure(ch * cap * sizeof(uint32_t)));
        HIP_TRY(cnt.ensure(ch * sizeof(uint32_t)));
        if (wide) {
            HIP_TRY(cand_t.ensure(ch * cap * sizeof(float)));
            HIP_TRY(best_t.ensure(ch * sizeof(float)));
            HIP_TRY(rstats.ensure(2 * sizeof(unsigned long long)));
            HIP_TRY(hipMemsetAsync(rstats.p, 0, 2 * sizeof(unsigned long long), stream));
        }
        const float eps = 1.01f * 9.765625e-04f + 2.0f * static_cast<float>(dim_p) * 5.9604645e-08f + 2.0e-6f;
        const float cm = static_cast<float>(dim + 16) * 2.384185791015625e-07f;
        // (one host check of the row norms after all chunks: a non-finite one sends the whole call to the caller's old path)
        HIP_TRY(flag.ensure(sizeof(uint32_t)));
        HIP_TRY(hipMemsetAsync(flag.p, 0, sizeof(uint32_t), stream));
        for (uint64_t r0 = 0; r0 < n; r0 += ch) {
            const uint64_t m = std::min<uint64_t>(ch, n - r0);
            const float *rows = d_rows + r0 * dim;
            const bool have_img = keep_mu && n <= ch && img_rows == d_rows && img_n == n && img_dim_p == dim_p;
            if (!have_img) HIP_TRY(launch_center_normalize_f16(rows, mu.as<float>(), m, m, dim, dim_p, xn2.as<float>(), x16.p, stream));
            HIP_TRY(launch_nonfinite_flag(xn2.as<float>(), m, flag.as<uint32_t>(), stream));
            img_rows = n <= ch ? d_rows : nullptr; img_n = n; img_dim_p = dim_p;
            HIP_TRY(hipMemsetAsync(cnt.p, 0, m * sizeof(uint32_t), stream));
            if (wide) {
                AssignWideArgs w{};
                w.x16 = x16.as<uint16_t>(); w.c16 = c16.as<uint16_t>(); w.xn2 = xn2.as<float>(); w.cn2 = cn2.as<float>();
                w.perm = d_perm.as<uint32_t>(); w.grp_cs = d_grp.as<float>(); w.grp_cn = d_grp.as<float>() + kc_pad / 32;
                w.m = m; w.kc = kc; w.kc_pad = kc_pad; w.dim_p = dim_p; w.kA = kA; w.eps = eps;
                // the reference's summation margin, tight: a squared difference carries <= 3 roundings, the 4-group <= 3 more, the
                // running sum one per group (index.rs:461-480; all terms non-negative) -- (dim / 4 + 6) 2^-24, taken twice over
                w.cm = static_cast<float>(dim / 4 + 16) * 1.1920928955078125e-07f;
                w.cand = cand.as<uint32_t>(); w.cand_t = cand_t.as<float>(); w.cand_cnt = cnt.as<uint32_t>(); w.best_t = best_t.as<float>(); w.cap = cap;
                // (the final assignment brackets the contraction kernel with HIP events -- time_kernels: the roofline of bench.py's
                //  index_build record divides by KERNEL time, not by the phase's wall time)
                hipEvent_t e0 = nullptr, e1 = nullptr;
                if (time_kernels) {
                    HIP_TRY(hipEventCreate(&e0)); HIP_TRY(hipEventCreate(&e1));
                    kernel_ev.push_back(e0); kernel_ev.push_back(e1);
                    HIP_TRY(hipEventRecord(e0, stream));
                }
                HIP_TRY(launch_assign_wide(w, stream));
                if (time_kernels) HIP_TRY(hipEventRecord(e1, stream));
                // (the two counters are same-address atomics from every wave: diagnostic runs only)
                HIP_TRY(launch_assign_resolve(w, rows, d_centroids, dim, d_cluster + r0, verbose() ? rstats.as<unsigned long long>() : nullptr, stream));
            } else {
            AssignF16Args a{};
            a.x16 = x16.as<uint16_t>(); a.c16 = c16.as<uint16_t>(); a.xn2 = xn2.as<float>(); a.cn2 = cn2.as<float>();
            a.m = m; a.kc = kc; a.kc_pad = kc_pad; a.dim_p = dim_p; a.eps = eps; a.cm = cm;
            a.cand = cand.as<uint32_t>(); a.cand_cnt = cnt.as<uint32_t>(); a.cap = cap;
            HIP_TRY(launch_assign_f16(a, stream));
            HIP_TRY(launch_assign_rescore(rows, d_centroids, m, dim, kc, cand.as<uint32_t>(), cnt.as<uint32_t>(), cap, d_cluster + r0, stream));
            }
            if (h_out) {
                const int evi = n_chunk++ & 1;
                if (int rc = flush_copy()) return rc;                   // the previous chunk: its kernels are done or running, this chunk's are queued
                HIP_TRY(hipEventRecord(cl.ev[evi], stream));
                pend_r0 = r0; pend_m = m; pend_ev = evi;
            }
            if (verbose() && r0 == 0 && n >= 65536) {      // candidates the screen left per row (first chunk)
                std::vector<uint32_t> hc(m);
                HIP_TRY(hipMemcpyAsync(hc.data(), cnt.p, m * sizeof(uint32_t), hipMemcpyDeviceToHost, stream));
                HIP_TRY(hipStreamSynchronize(stream));
                uint64_t sum = 0, over = 0; uint32_t mx = 0;
                for (uint32_t c : hc) { sum += c; mx = std::max(mx, c); over += c > cap; }
                std::fprintf(stderr, "[pqv] f16 assignment: %.2f candidates per row (max %u, %llu rows over the %u-entry list) of %u centroids\n",
                             static_cast<double>(sum) / static_cast<double>(m), mx, (unsigned long long)over, cap, kc);
            }
        }
        if (h_out) { if (int rc = flush_copy()) return rc; HIP_TRY(hipStreamSynchronize(cl.s)); }
        const double t_run1 = now_s();
        uint32_t h = 0;
        HIP_TRY(hipMemcpyAsync(&h, flag.p, sizeof h, hipMemcpyDeviceToHost, stream));
        if (verbose() && n >= (1u << 20)) std::fprintf(stderr, "[pqv] f16 assignment: %llu rows enqueued%s in %.1f ms\n", (unsigned long long)n,
                                                       h_out ? " and downloaded" : "", (t_run1 - t_run0) * 1e3);
        if (wide && verbose()) {
            unsigned long long hs[2] = {0, 0};
            HIP_TRY(hipMemcpyAsync(hs, rstats.p, sizeof hs, hipMemcpyDeviceToHost, stream));
            HIP_TRY(hipStreamSynchronize(stream));
            exact_rows += hs[0]; exact_evals += hs[1]; rows_total += n;
            if (n >= 65536) std::fprintf(stderr, "[pqv] f16 assignment (256 x 256 tiles): %.1f %% of %llu rows evaluated exactly, %.2f evaluations per such row\n",
                                         100.0 * static_cast<double>(hs[0]) / static_cast<double>(n), (unsigned long long)n,
                                         hs[0] ? static_cast<double>(hs[1]) / static_cast<double>(hs[0]) : 0.0);
        }
        HIP_TRY(hipStreamSynchronize(stream));
        if (!kernel_ev.empty()) {
            double ms_sum = 0.0;
            for (size_t i = 0; i + 1 < kernel_ev.size(); i += 2) {
                float ms = 0.0f;
                if (hipEventElapsedTime(&ms, kernel_ev[i], kernel_ev[i + 1]) == hipSuccess) ms_sum += ms;
            }
            g_build_stats[8] = ms_sum * 1e-3; g_build_stats[9] = static_cast<double>(kernel_ev.size() / 2);
            for (hipEvent_t e : kernel_ev) (void)hipEventDestroy(e);
            kernel_ev.clear();
        }
        *fallback = h != 0;
        return PQV_OK;
    }
    std::vector<hipEvent_t> kernel_ev;
    bool time_kernels = false;
    // (an early error return of run() leaves its events here)
    ~GemmAssign() { for (hipEvent_t e : kernel_ev) (void)hipEventDestroy(e); }
};

// acc[c] = ((0 + m[0][c]) + m[1][c]) + ... + m[rows - 1][c] for c in [0, width), width a multiple of 16: `width` independent f32
// chains, each in its own sequential order (element-wise vector adds do not re-associate anything).  The AVX2 body is chosen at
// run time; both bodies add the same operands in the same order, so the sums are identical bit for bit.
typedef float v4f_t __attribute__((vector_size(16), aligned(4)));
typedef float v8f_t __attribute__((vector_size(32), aligned(4)));
// k-means++ pick (index.rs:372-383): the first slot whose SEQUENTIAL f32 cumulative sum reaches `gen_range(0.0..1.0) * total`.  The
// cumulative sum does not depend on the threshold, and `total` (the chunk sums of :356-370, ~6 us per round) does not depend on the
// cumulative sum: a second host thread starts the walk the moment a round's minima are final, keeping the prefix it has passed; when
// the main thread publishes the threshold it looks back through that prefix (linearly: the reference's "first slot", whatever the
// values -- a NaN makes the prefix non-monotone) or walks on with the reference's own compare.  The same adds in the same order on
// the same values: the pick is the reference's.
struct PrefixScout {
    const float *md = nullptr;
    uint64_t n = 0;
    std::vector<float> prefix;
    alignas(64) std::atomic<uint32_t> req{0};          // main -> scout: walk round r (0: none yet; ~0u: exit)
    alignas(64) std::atomic<uint32_t> thr_round{0};    // main -> scout: round whose threshold (or cancellation) is published
    float thr_value = 0.0f;
    bool cancel = false;
    alignas(64) std::atomic<uint32_t> res_round{0};    // scout -> main: round whose result is published
    uint64_t res_slot = ~0ull;
    std::thread th;
    static void relax() {
#if defined(__x86_64__)
        __builtin_ia32_pause();
#endif
    }
    void run() {
        uint32_t seen = 0;
        for (;;) {
            uint32_t r;
            while ((r = req.load(std::memory_order_acquire)) == seen) relax();
            if (r == ~0u) return;
            seen = r;
            float cumsum = 0.0f;
            uint64_t pos = 0, found = ~0ull;
            bool have_thr = false, stop = false;
            float thr = 0.0f;
            while (!have_thr && pos < n) {                     // the threshold is not known yet: walk and remember
                const uint64_t e = pos + 64 < n ? pos + 64 : n;
                for (; pos < e; ++pos) { cumsum = cumsum + md[pos]; prefix[pos] = cumsum; }
                if (thr_round.load(std::memory_order_acquire) == r) have_thr = true;
                else if (req.load(std::memory_order_acquire) == ~0u) return;          // (the build failed half-way through a round)
            }
            while (!have_thr) {                                // (walked to the end before the total was there)
                if (thr_round.load(std::memory_order_acquire) == r) have_thr = true;
                else if (req.load(std::memory_order_acquire) == ~0u) return;
                else relax();
            }
            thr = thr_value; stop = cancel;
            if (!stop) {
                for (uint64_t i = 0; i < pos; ++i)             // what has been passed already: the FIRST slot at or above the threshold
                    if (prefix[i] >= thr) { found = i; break; }
                if (found == ~0ull)
                    for (; pos < n; ++pos) {                   // index.rs:375-383 from here on
                        cumsum = cumsum + md[pos];
                        if (cumsum >= thr) { found = pos; break; }
                    }
            }
            res_slot = found;
            res_round.store(r, std::memory_order_release);
        }
    }
    void start(const float *minima, uint64_t count) {
        md = minima; n = count; prefix.assign(count, 0.0f);
        th = std::thread([this] { run(); });
    }
    void begin(uint32_t r) { req.store(r, std::memory_order_release); }
    uint64_t finish(uint32_t r, float threshold, bool cancelled) {
        thr_value = threshold; cancel = cancelled;
        thr_round.store(r, std::memory_order_release);
        while (res_round.load(std::memory_order_acquire) != r) relax();
        return res_slot;
    }
    ~PrefixScout() {
        if (th.joinable()) { req.store(~0u, std::memory_order_release); th.join(); }
    }
};

__attribute__((target("avx2"))) static void chain_sums_avx2(const float *m, uint64_t rows, uint64_t width, float *acc) {
    for (uint64_t c0 = 0; c0 < width; c0 += 32) {          // four 8-lane accumulators: 32 chains per pass over the rows
        const uint64_t nb = std::min<uint64_t>(4, (width - c0) / 8);
        v8f_t s[4] = {{0, 0, 0, 0, 0, 0, 0, 0}, {0, 0, 0, 0, 0, 0, 0, 0}, {0, 0, 0, 0, 0, 0, 0, 0}, {0, 0, 0, 0, 0, 0, 0, 0}};
        for (uint64_t t = 0; t < rows; ++t) {
            const float *r = m + t * width + c0;
            for (uint64_t b = 0; b < nb; ++b) s[b] = s[b] + *reinterpret_cast<const v8f_t *>(r + 8 * b);
        }
        for (uint64_t b = 0; b < nb; ++b) *reinterpret_cast<v8f_t *>(acc + c0 + 8 * b) = s[b];
    }
}
static void chain_sums(const float *m, uint64_t rows, uint64_t width, float *acc) {
    static const bool avx2 = __builtin_cpu_supports("avx2");
    if (avx2) { chain_sums_avx2(m, rows, width, acc); return; }
    for (uint64_t c0 = 0; c0 < width; c0 += 16) {
        v4f_t s[4] = {{0, 0, 0, 0}, {0, 0, 0, 0}, {0, 0, 0, 0}, {0, 0, 0, 0}};
        for (uint64_t t = 0; t < rows; ++t) {
            const float *r = m + t * width + c0;
            for (int b = 0; b < 4; ++b) s[b] = s[b] + *reinterpret_cast<const v4f_t *>(r + 4 * b);
        }
        for (int b = 0; b < 4; ++b) *reinterpret_cast<v4f_t *>(acc + c0 + 4 * b) = s[b];
    }
}

// The inverted lists built on the device (kernels_build.hip: launch_list_sort) -- the same ascending-row-id lists as
// lists_from_assignment above, without the assignment's trip to the host.  PQV_DEVICE_LISTS=0 keeps the host sort (A/B).
struct DeviceLists {
    DevBuf cnt, tot, bad;
    uint32_t rpb = 4096;
    static bool applicable(uint64_t n, uint32_t k) {
        static const bool on = [] { const char *e = std::getenv("PQV_DEVICE_LISTS"); return !(e && e[0] == '0'); }();
        return on && n > 0 && n <= 0xFFFFFFFFull && k > 0 && k <= 4096;
    }
    int prepare(uint64_t n, uint32_t k) {
        // blocks of 4096 rows; below 2^20 rows smaller blocks keep a few hundred of them in flight
        rpb = n >= (1u << 20) ? 4096 : n >= (1u << 17) ? 1024 : 256;
        const uint64_t nblk = (n + rpb - 1) / rpb;
        HIP_TRY(cnt.alloc(static_cast<size_t>(k) * nblk * sizeof(uint32_t)));
        HIP_TRY(tot.alloc(static_cast<size_t>(k) * sizeof(unsigned long long)));
        HIP_TRY(bad.alloc(sizeof(uint32_t)));
        return PQV_OK;
    }
    // enqueues the sort; *bad stays 0 unless an assignment is out of range (check() after the stream drained)
    int run(const uint32_t *d_assign, uint64_t n, uint32_t k, uint64_t *d_list_off, uint32_t *d_list_rows, hipStream_t stream) {
        HIP_TRY(hipMemsetAsync(bad.p, 0, sizeof(uint32_t), stream));
        HIP_TRY(pqv::launch_list_sort(d_assign, n, k, cnt.as<uint32_t>(), rpb, tot.as<unsigned long long>(), d_list_off, d_list_rows,
                                 bad.as<uint32_t>(), stream));
        return PQV_OK;
    }
};

// d_data [n, dim] resident; writes d_centroids [k, dim] (device) and optionally the final
// assignment (host).
int kmeans_device(const float *d_data, uint64_t n, uint32_t dim, uint32_t k, uint32_t max_iters,
                  uint64_t seed, uint32_t workers, hipStream_t stream, float *d_centroids,
                  std::vector<uint32_t> *assign_out, uint32_t *iters_run) {
    using namespace pqv;
    if (workers == 0) workers = host_workers();
    StdRng rng = StdRng::seed_from_u64(seed);                                          // :327
    HIP_TRY(hipMemsetAsync(d_centroids, 0, static_cast<size_t>(k) * dim * sizeof(float), stream)); // :330

    // k-means++ subset (:332-338)
    uint64_t init_n = std::max<uint64_t>(std::min<uint64_t>(n, 50000), k);
    DevBuf d_init_own, d_idx;
    const float *d_init = d_data;
    std::vector<uint64_t> init_indices;
    if (init_n != n) {
        init_indices = index_sample(rng, n, init_n);
        HIP_TRY(d_idx.alloc(init_n * sizeof(uint64_t)));
        HIP_TRY(d_init_own.alloc(init_n * dim * sizeof(float)));
        HIP_TRY(hipMemcpyAsync(d_idx.p, init_indices.data(), init_n * sizeof(uint64_t),
                               hipMemcpyHostToDevice, stream));
        HIP_TRY(launch_gather_rows(d_data, nullptr, d_idx.as<uint64_t>(), init_n, dim,
                                   d_init_own.as<float>(), stream));
        d_init = d_init_own.as<float>();
    }
    // Centroid i is row picks[i] of the subset (~0: none, the centroid keeps its zero fill).  A round measures against that
    // row where it lies and the rows are copied into the table once, after the last round, so that a round is ONE command
    // on the stream: no device-to-device copy before its kernel, no download after it (the kernel mirrors the minima into
    // pinned host memory): 83 -> 64 us per round on C3's 50 000 x 768 subset (kernel 26 us, the host scans about 20).
    std::vector<uint64_t> picks(k, ~0ull);
    picks[0] = rng.range_usize(0, init_n);                                             // :340-342
    auto centroid_row = [&](uint32_t j) -> const float * {
        return picks[j] != ~0ull ? d_init + picks[j] * dim : d_centroids + static_cast<uint64_t>(j) * dim;
    };

    // min_distances (:344-352), then one streaming pass per round (:354-369)
    DevBuf d_min;
    HIP_TRY(d_min.alloc(init_n * sizeof(float)));
    {
        std::vector<float> inf(init_n, INFINITY);
        HIP_TRY(hipMemcpyAsync(d_min.p, inf.data(), init_n * sizeof(float), hipMemcpyHostToDevice, stream));
        HIP_TRY(hipStreamSynchronize(stream));
    }
    PinnedBuf h_min;
    HIP_TRY(h_min.ensure(init_n * sizeof(float)));
    for (uint64_t t = 0; t < init_n; ++t) h_min.as<float>()[t] = INFINITY;     // the mirror starts equal to d_min
    StreamArgs sa{};
    sa.mat = d_init; sa.row_of = nullptr; sa.list_off = nullptr; sa.probe = nullptr; sa.cand_base = nullptr;
    sa.single_begin = 0; sa.single_end = init_n;
    sa.nq = 1; sa.nprobe = 1; sa.dim = dim; sa.k = 1;
    sa.rows_per_block = 256;
    sa.blocks_per_list = static_cast<uint32_t>((init_n + 255) / 256);
    sa.max_pos = ~0ull; sa.metric = PQV_L2SQ_REF4;
    sa.out_f32 = d_min.as<float>();
    sa.mirror_f32 = h_min.as<float>();

    // chunking of the partial sums (:259-265,:305-306)
    const uint64_t w = std::max<uint64_t>(1, std::min<uint64_t>(workers, init_n));
    const uint64_t chunk = (init_n + w - 1) / w;
    // With many chunks (a 256-core host: 256 chunks of 196 minima) the w independent chunk chains are added as vector lanes:
    // the kernels keep a second mirror in chunk-transposed order [position in chunk][chunk] (padding stays +0.0: x + 0.0 == x
    // for the non-negative sums), and step t adds row t of it to the w running sums -- every chain still in its own order.
    const uint64_t n_chunks = (init_n + chunk - 1) / chunk;
    const uint64_t wpad = (n_chunks + 15) / 16 * 16;
    const bool use_t = n_chunks >= 16 && chunk * wpad <= (64ull << 20);
    PinnedBuf h_min_t;
    std::vector<float> chain_acc;
    if (use_t) {
        HIP_TRY(h_min_t.ensure(chunk * wpad * sizeof(float)));
        float *mt = h_min_t.as<float>();
        for (uint64_t t = 0; t < chunk * wpad; ++t) mt[t] = 0.0f;
        for (uint64_t pos = 0; pos < init_n; ++pos) mt[(pos % chunk) * wpad + pos / chunk] = INFINITY;
        sa.mirror_t = mt; sa.mirror_chunk = static_cast<uint32_t>(chunk); sa.mirror_stride = static_cast<uint32_t>(wpad);
        chain_acc.resize(wpad);
    }

    const double t_pp0 = now_s();
    // Round 4: from the second measured centroid on, a round first screens the rows with int8 images (minupd_screen_kernel): a row
    // whose distance to the new centroid provably is not below its current minimum is not evaluated (after a dozen rounds: most).
    // Rows of a multiple of 64 dims, finite data; PQV_KPP_SCREEN=0 keeps the plain streaming pass (A/B, tests).
    DevBuf d_kimg, d_kn2i, d_kres, d_kaux, d_kmm;
    MinUpdScreenArgs ms{};
    bool kpp_screen = false;
    {
        const char *e = std::getenv("PQV_KPP_SCREEN");
        if (!(e && *e == '0') && (dim % 64) == 0 && dim >= 128 && dim <= 8192 && init_n >= 1024 && k > 8) {
            const uint64_t n_tiles = (init_n + 15) / 16;
            // one "list" holding the whole subset: {list_off[2], blk_off[2]} u64, then centre[dim], half, scale, radius floats
            HIP_TRY(d_kaux.alloc(4 * sizeof(uint64_t) + (static_cast<size_t>(dim) + 3) * sizeof(float)));
            const uint64_t h_off[4] = {0, init_n, 0, n_tiles};
            HIP_TRY(hipMemcpyAsync(d_kaux.p, h_off, sizeof h_off, hipMemcpyHostToDevice, stream));
            const uint64_t *d_loff = d_kaux.as<uint64_t>(), *d_boff = d_loff + 2;
            float *d_ctr = reinterpret_cast<float *>(d_kaux.as<uint64_t>() + 4), *d_half = d_ctr + dim, *d_scale = d_half + 1, *d_rad = d_scale + 1;
            HIP_TRY(d_kmm.alloc(2 * static_cast<size_t>(dim) * sizeof(uint32_t)));
            uint32_t *kmin = d_kmm.as<uint32_t>(), *kmax = kmin + dim;
            HIP_TRY(hipMemsetAsync(kmin, 0xFF, static_cast<size_t>(dim) * sizeof(uint32_t), stream));
            HIP_TRY(hipMemsetAsync(kmax, 0, static_cast<size_t>(dim) * sizeof(uint32_t), stream));
            HIP_TRY(d_kimg.alloc(n_tiles * 16 * dim));
            HIP_TRY(d_kn2i.alloc(init_n * sizeof(int)));
            HIP_TRY(d_kres.alloc(init_n * sizeof(float)));
            HIP_TRY(launch_list_minmax(d_init, d_loff, 1, init_n, dim, kmin, kmax, stream));
            HIP_TRY(launch_list_center(kmin, kmax, 1, dim, d_loff, d_ctr, d_half, d_scale, d_rad, stream));
            HIP_TRY(launch_block_rows_i8(d_init, d_loff, d_boff, 1, n_tiles, dim, d_ctr, d_scale, d_half, d_rad, d_kimg.p, d_kn2i.as<int>(),
                                         d_kres.as<float>(), stream));
            float h_hs[2] = {0.0f, 0.0f};       // {half range, scale}
            HIP_TRY(hipMemcpyAsync(h_hs, d_half, sizeof h_hs, hipMemcpyDeviceToHost, stream));
            HIP_TRY(hipStreamSynchronize(stream));
            // (a non-finite value anywhere in the subset shows in the half range; a degenerate scale -- all rows equal -- has nothing to screen)
            if (h_hs[0] > 0.0f && h_hs[0] < 1.0e30f && h_hs[1] > 1.0e-30f && h_hs[1] < 1.0e15f) {
                ms.rows = d_init; ms.img = static_cast<const float4 *>(d_kimg.p); ms.n2i = d_kn2i.as<int>(); ms.res = d_kres.as<float>();
                ms.n = init_n; ms.dim = dim; ms.inv_s = 1.0f / h_hs[1];
                ms.cm = static_cast<float>(dim + 16) * 2.384185791015625e-07f;
                ms.min_d = d_min.as<float>(); ms.mirror = h_min.as<float>();
                ms.mirror_t = sa.mirror_t; ms.mirror_chunk = sa.mirror_chunk; ms.mirror_stride = sa.mirror_stride;
                kpp_screen = true;
            }
        }
    }
    // (Measured and dropped: the screened rounds as ONE resident kernel fed through pinned memory -- commands polled with relaxed
    //  system-scope loads, minima mirrored with system-scope stores, a ticket per block -- 45 us a round against 24 for a launch per
    //  round: the PCIe round trips of the hand-shake cost more than a kernel launch and its completion.  Likewise a completion flag
    //  written by the last block of a per-round launch: the per-block system-scope release costs the kernel 20 us.)
    // Round 6: the rounds enqueued ahead with the pick on the device (kernels_kpp.hip: the reference's sequential f32 sums evaluated
    // exactly by composing the additions' integer images) -- no host round trip per centroid.  The draws of :373 are taken from a copy
    // of the generator (one per round while every total is positive); a round the device cannot decide the reference's way stops the
    // chain and the host walk below takes over from that round.  PQV_KPP_DEVICE=0 keeps the host walk for every round (A/B, tests).
    uint32_t i_start = 1;
    bool resume_after_update = false;
    sa.queries = centroid_row(0);  // distances to centroid 0
    bool kpp_dev = false;
    {
        const char *e = std::getenv("PQV_KPP_DEVICE");
        kpp_dev = !(e && *e == '0') && kpp_screen && k >= 3 && init_n <= 57344 && n_chunks <= 1024 && chunk <= 0xFFFFFFFFull;
    }
    if (kpp_dev) {
        const double t_d0 = now_s();
        // picks [k] | chunk sums [n_chunks] | block sums [64] | quarter-run pairs [4096] | block flags [64] | head [72] (u64 each), draws [k] f32, state [4] u32
        const size_t n64 = static_cast<size_t>(k) + n_chunks + 64 + 4096 + 64 + 72;
        const size_t bytes = n64 * 8 + (static_cast<size_t>(k) + 4) * 4;
        DevBuf d_kpp;
        HIP_TRY(d_kpp.alloc(bytes));
        HIP_TRY(hipMemsetAsync(d_kpp.p, 0, bytes, stream));
        unsigned long long *d_picks = d_kpp.as<unsigned long long>();
        float *d_u = reinterpret_cast<float *>(d_picks + n64);
        uint32_t *d_state = reinterpret_cast<uint32_t *>(d_u + k);
        StdRng ahead = rng;
        std::vector<float> draws(k, 0.0f);
        for (uint32_t i = 1; i < k; ++i) draws[i] = ahead.unit_f32();
        const unsigned long long pick0 = picks[0];
        HIP_TRY(hipMemcpyAsync(d_picks, &pick0, sizeof pick0, hipMemcpyHostToDevice, stream));
        HIP_TRY(hipMemcpyAsync(d_u, draws.data(), static_cast<size_t>(k) * sizeof(float), hipMemcpyHostToDevice, stream));
        StreamArgs sd = sa;
        sd.mirror_f32 = nullptr; sd.mirror_t = nullptr;
        HIP_TRY(launch_stream(sd, STREAM_MINUPD, stream));
        MinUpdScreenArgs md_args = ms;
        md_args.mirror = nullptr; md_args.mirror_t = nullptr; md_args.stop = d_state;
        KppPickArgs pa{};
        pa.md = d_min.as<float>(); pa.n = static_cast<uint32_t>(init_n);
        pa.chunk = static_cast<uint32_t>(chunk); pa.n_chunks = static_cast<uint32_t>(n_chunks);
        pa.chunk_sum = d_picks + k; pa.blk_sum = pa.chunk_sum + n_chunks; pa.run_sum = pa.blk_sum + 64; pa.blk_done = pa.run_sum + 4096; pa.head = pa.blk_done + 64;
        pa.u = d_u; pa.picks = d_picks; pa.state = d_state;
        for (uint32_t i = 1; i < k; ++i) {
            if (i > 1) {                                   // (round 1 would re-measure centroid 0)
                md_args.pick_dev = d_picks + (i - 1);
                HIP_TRY(launch_minupd_screen(md_args, stream));
            }
            pa.round = i;
            HIP_TRY(launch_kpp_pick(pa, stream));
        }
        uint32_t h_state[4] = {0, 0, 0, 0};
        std::vector<unsigned long long> h_picks(k, ~0ull);
        HIP_TRY(hipMemcpyAsync(h_picks.data(), d_picks, static_cast<size_t>(k) * sizeof(unsigned long long), hipMemcpyDeviceToHost, stream));
        HIP_TRY(hipMemcpyAsync(h_state, d_state, sizeof h_state, hipMemcpyDeviceToHost, stream));
        HIP_TRY(hipStreamSynchronize(stream));
        const uint32_t decided = h_state[0] ? std::min<uint32_t>(std::max<uint32_t>(h_state[1], 1), k) : k;   // rounds 1 .. decided - 1
        for (uint32_t i = 1; i < decided; ++i) picks[i] = h_picks[i];
        for (uint32_t i = 1; i < decided; ++i) (void)rng.unit_f32();     // the generator follows the rounds that were decided
        if (verbose()) std::fprintf(stderr, "[pqv] k-means++ on the device: rounds 1..%u of %u in %.1f ms%s\n", decided - 1, k - 1,
                                    (now_s() - t_d0) * 1e3, h_state[0] ? " (the host walk takes over)" : "");
        if (decided < k) {
            if (verbose()) std::fprintf(stderr, "[pqv] k-means++: round %u back to the host (reason %u)\n", decided, h_state[2]);
            // the host walk needs its mirrors of the minima as the device left them (round `decided`'s update has run)
            HIP_TRY(hipMemcpyAsync(h_min.as<float>(), d_min.p, init_n * sizeof(float), hipMemcpyDeviceToHost, stream));
            HIP_TRY(hipStreamSynchronize(stream));
            if (use_t) {
                float *mt = h_min_t.as<float>();
                const float *hm = h_min.as<float>();
                for (uint64_t pos = 0; pos < init_n; ++pos) mt[(pos % chunk) * wpad + pos / chunk] = hm[pos];
            }
            resume_after_update = decided > 1;
        }
        i_start = decided;
    } else {
        HIP_TRY(launch_stream(sa, STREAM_MINUPD, stream));
    }
    double tt_gpu = 0.0, tt_sum = 0.0, tt_pick = 0.0;       // PQV_VERBOSE: where a round's time goes
    const bool vb = verbose();
    // round 6: the pick's walk starts on a second host thread while this one adds the chunk sums (PrefixScout; PQV_KPP_SCOUT=0: one thread)
    std::unique_ptr<PrefixScout> scout;
    {
        const char *e = std::getenv("PQV_KPP_SCOUT");
        if (!(e && *e == '0') && std::thread::hardware_concurrency() >= 2 && k > 2 && init_n >= 4096 && i_start < k) {
            scout.reset(new (std::nothrow) PrefixScout());
            if (scout) scout->start(h_min.as<float>(), init_n);
        }
    }
    for (uint32_t i = i_start; i < k; ++i) {
        const double tr0 = vb ? now_s() : 0.0;
        if (i > 1 && !(resume_after_update && i == i_start)) {  // round 1 would re-measure centroid 0: min-update is the identity
            if (kpp_screen && picks[i - 1] != ~0ull) {
                ms.pick = picks[i - 1];
                HIP_TRY(launch_minupd_screen(ms, stream));
            } else {                                                // (a centroid that is no subset row -- the zero fill -- takes the plain pass)
                sa.queries = centroid_row(i - 1);
                HIP_TRY(launch_stream(sa, STREAM_MINUPD, stream));
            }
        }
        {
            // the round's only command: poll for its completion (the blocking wait's wake-up costs several microseconds a round)
            hipError_t qe;
            uint32_t polls = 0;
            while ((qe = hipStreamQuery(stream)) == hipErrorNotReady && ++polls < 200000u) { }
            if (qe != hipSuccess) { (void)hipGetLastError(); HIP_TRY(hipStreamSynchronize(stream)); }
        }
        const double tr1 = vb ? now_s() : 0.0;
        if (scout) scout->begin(i);
        const float *md = h_min.as<float>();
        // total = sum over worker chunks of the chunk's sequential f32 sum (:356-370)
        // Each chunk's sum is its own sequential f32 chain and the chains are independent of each other, so eight
        // full chunks run side by side; the chunk sums still join `total` in ascending chunk order.
        float total = 0.0f;
        uint64_t s = 0;
        if (use_t) {
            const float *mt = h_min_t.as<float>();
            float *acc = chain_acc.data();
            for (uint64_t c = 0; c < wpad; ++c) acc[c] = 0.0f;
            chain_sums(mt, chunk, wpad, acc);                                          // w independent chains, one per vector lane
            for (uint64_t c = 0; c < n_chunks; ++c) total = total + acc[c];            // joined in ascending chunk order
            s = init_n;
        } else if (chunk < init_n) {
            for (; s + 8 * chunk <= init_n; s += 8 * chunk) {
                const float *p = md + s;
                float a0 = 0.0f, a1 = 0.0f, a2 = 0.0f, a3 = 0.0f, a4 = 0.0f, a5 = 0.0f, a6 = 0.0f, a7 = 0.0f;
                for (uint64_t t = 0; t < chunk; ++t) {
                    a0 = a0 + p[t];             a1 = a1 + p[chunk + t];     a2 = a2 + p[2 * chunk + t]; a3 = a3 + p[3 * chunk + t];
                    a4 = a4 + p[4 * chunk + t]; a5 = a5 + p[5 * chunk + t]; a6 = a6 + p[6 * chunk + t]; a7 = a7 + p[7 * chunk + t];
                }
                total = total + a0; total = total + a1; total = total + a2; total = total + a3;
                total = total + a4; total = total + a5; total = total + a6; total = total + a7;
            }
        }
        for (; s < init_n; s += chunk) {
            const uint64_t e = std::min(init_n, s + chunk);
            float local = 0.0f;
            for (uint64_t t = s; t < e; ++t) local = local + md[t];
            total = total + local;
        }
        const double tr2 = vb ? now_s() : 0.0;
        if (total > 0.0f) {
            const float threshold = rng.unit_f32() * total;                            // :373
            if (scout) {
                picks[i] = scout->finish(i, threshold, false);                         // (~0: no slot reached it, as below)
            } else {
                float cumsum = 0.0f;
                for (uint64_t slot = 0; slot < init_n; ++slot) {                       // :375-383
                    cumsum = cumsum + md[slot];
                    if (cumsum >= threshold) { picks[i] = slot; break; }
                }
            }
        } else {
            if (scout) (void)scout->finish(i, 0.0f, true);
            picks[i] = rng.range_usize(0, init_n);                                     // :385
        }
        if (vb) { const double tr3 = now_s(); tt_gpu += tr1 - tr0; tt_sum += tr2 - tr1; tt_pick += tr3 - tr2; }
    }
    if (vb) std::fprintf(stderr, "[pqv] k-means++ rounds: launch + wait %.1f ms, chunk sums %.1f ms, pick scan %.1f ms\n", tt_gpu * 1e3, tt_sum * 1e3, tt_pick * 1e3);
    // the chosen rows -> the centroid table (a centroid without a pick keeps its zero fill, as in the reference)
    {
        bool all = true;
        for (uint32_t i = 0; i < k; ++i) all = all && picks[i] != ~0ull;
        if (all) {
            DevBuf d_picks;
            HIP_TRY(d_picks.alloc(static_cast<size_t>(k) * sizeof(uint64_t)));
            HIP_TRY(hipMemcpyAsync(d_picks.p, picks.data(), static_cast<size_t>(k) * sizeof(uint64_t), hipMemcpyHostToDevice, stream));
            HIP_TRY(launch_gather_rows(d_init, nullptr, d_picks.as<uint64_t>(), k, dim, d_centroids, stream));
            HIP_TRY(hipStreamSynchronize(stream));
        } else {
            for (uint32_t i = 0; i < k; ++i)
                if (picks[i] != ~0ull)
                    HIP_TRY(hipMemcpyAsync(d_centroids + static_cast<uint64_t>(i) * dim, d_init + picks[i] * dim,
                                           dim * sizeof(float), hipMemcpyDeviceToDevice, stream));
            HIP_TRY(hipStreamSynchronize(stream));
        }
    }
    HIP_TRY(hipStreamSynchronize(stream));
    d_min.release(); d_init_own.release(); d_idx.release(); d_kimg.release(); d_kn2i.release(); d_kres.release(); d_kaux.release(); d_kmm.release();
    HIP_TRY(hipStreamSynchronize(stream));
    const double t_pp1 = now_s();
    g_build_stats[0] = t_pp1 - t_pp0; g_build_stats[8] = 0.0; g_build_stats[9] = 0.0;
    if (verbose()) std::fprintf(stderr, "[pqv] k-means++: %u rounds over %llu rows in %.3f s\n", k,
                                (unsigned long long)init_n, t_pp1 - t_pp0);

    // Lloyd iterations (:392-454)
    DevBuf d_assign_a, d_assign_b, d_counts, d_list_rows, d_list_off;
    HIP_TRY(d_assign_a.alloc(n * sizeof(uint32_t)));
    HIP_TRY(d_assign_b.alloc(n * sizeof(uint32_t)));
    HIP_TRY(d_counts.alloc((static_cast<size_t>(k) + 1) * sizeof(unsigned long long)));
    HIP_TRY(d_list_rows.alloc(n * sizeof(uint32_t)));
    HIP_TRY(d_list_off.alloc((static_cast<size_t>(k) + 1) * sizeof(uint64_t)));
    HIP_TRY(hipMemsetAsync(d_assign_a.p, 0, n * sizeof(uint32_t), stream));            // :392
    uint32_t *d_prev = d_assign_a.as<uint32_t>(), *d_cur = d_assign_b.as<uint32_t>();
    std::vector<uint32_t> h_assign(n, 0);
    std::vector<unsigned long long> h_counts(static_cast<size_t>(k) + 1);
    std::vector<uint64_t> off;
    std::vector<uint32_t> rows;
    uint32_t iters = 0;
    ScreenedAssign screen;
    GemmAssign gemm;
    const bool use_gemm = GemmAssign::applicable(dim, k);
    const bool use_screen = !use_gemm && ScreenedAssign::applicable(dim, k);
    {
        const char *e = std::getenv("PQV_LLOYD_KEEP_IMAGES");
        gemm.keep_mu = !(e && *e == '0');
    }
    const bool dev_lists = DeviceLists::applicable(n, k);
    DeviceLists dlists;
    uint32_t h_bad = 0;
    if (dev_lists) {
        if (int rc = dlists.prepare(n, k)) return rc;
        HIP_TRY(hipMemsetAsync(dlists.bad.p, 0, sizeof(uint32_t), stream));
    }
    for (uint32_t iter = 0; iter < max_iters; ++iter) {
        HIP_TRY(hipMemsetAsync(d_counts.p, 0, (static_cast<size_t>(k) + 1) * sizeof(unsigned long long), stream));
        unsigned long long *d_changed = d_counts.as<unsigned long long>() + k;
        bool exact_assign = !use_screen;
        if (use_gemm) {
            bool bad_c = false, bad_r = false;
            if (int rc = gemm.set_centroids(d_centroids, k, dim, stream, &bad_c)) return rc;
            if (!bad_c) {
                if (int rc = gemm.run(d_data, n, d_cur, stream, &bad_r)) return rc;
                if (!bad_r) HIP_TRY(launch_count_changed(d_cur, d_prev, n, d_changed, stream));
            }
            exact_assign = bad_c || bad_r;
        }
        if (use_screen) {
            bool bad_c = false, bad_r = false;
            if (int rc = screen.set_centroids(d_centroids, k, dim, stream, &bad_c)) return rc;
            if (!bad_c) {
                if (int rc = screen.run(d_data, n, d_cur, stream, &bad_r)) return rc;
                if (!bad_r) HIP_TRY(launch_count_changed(d_cur, d_prev, n, d_changed, stream));
            }
            exact_assign = bad_c || bad_r;
        }
        if (exact_assign)
            HIP_TRY(launch_assign(d_data, n, dim, d_centroids, k, d_cur, d_prev, d_changed, nullptr, stream));
        HIP_TRY(hipMemcpyAsync(h_counts.data(), d_counts.p, h_counts.size() * sizeof(unsigned long long),
                               hipMemcpyDeviceToHost, stream));
        if (!dev_lists) HIP_TRY(hipMemcpyAsync(h_assign.data(), d_cur, n * sizeof(uint32_t), hipMemcpyDeviceToHost, stream));
        if (dev_lists && iter > 0)                 // (the previous iteration's range check rides on this synchronisation)
            HIP_TRY(hipMemcpyAsync(&h_bad, dlists.bad.p, sizeof(uint32_t), hipMemcpyDeviceToHost, stream));
        HIP_TRY(hipStreamSynchronize(stream));
        if (h_bad) return fail(PQV_ERR_HIP, "internal error: Lloyd assignment out of range");
        iters++;
        std::swap(d_prev, d_cur);
        if (h_counts[k] == 0) break;                                                   // :432
        if (dev_lists) {
            if (int rc = dlists.run(d_prev, n, k, d_list_off.as<uint64_t>(), d_list_rows.as<uint32_t>(), stream)) return rc;
        } else {
            if (!lists_from_assignment(h_assign.data(), n, k, off, rows))
                return fail(PQV_ERR_HIP, "internal error: Lloyd assignment out of range");
            HIP_TRY(hipMemcpyAsync(d_list_off.p, off.data(), off.size() * sizeof(uint64_t), hipMemcpyHostToDevice, stream));
            HIP_TRY(hipMemcpyAsync(d_list_rows.p, rows.data(), n * sizeof(uint32_t), hipMemcpyHostToDevice, stream));
        }
        HIP_TRY(launch_lloyd_update(d_data, dim, d_list_rows.as<uint32_t>(), d_list_off.as<uint64_t>(),
                                    k, d_centroids, stream));                          // :436-453
    }
    if (dev_lists && iters > 0) {
        HIP_TRY(hipMemcpyAsync(&h_bad, dlists.bad.p, sizeof(uint32_t), hipMemcpyDeviceToHost, stream));
        if (assign_out) HIP_TRY(hipMemcpyAsync(h_assign.data(), d_prev, n * sizeof(uint32_t), hipMemcpyDeviceToHost, stream));
    }
    HIP_TRY(hipStreamSynchronize(stream));
    if (h_bad) return fail(PQV_ERR_HIP, "internal error: Lloyd assignment out of range");
    g_build_stats[1] = now_s() - t_pp1; g_build_stats[2] = iters; g_build_stats[6] = use_gemm ? 2.0 : use_screen ? 1.0 : 0.0;
    if (verbose()) std::fprintf(stderr, "[pqv] Lloyd: %u iterations over %llu rows in %.3f s\n", iters,
                                (unsigned long long)n, now_s() - t_pp1);
    if (iters_run) *iters_run = iters;
    if (assign_out) *assign_out = std::move(h_assign);
    return PQV_OK;
}

int build_index_impl(const pqv_corpus *corpus, uint32_t n_clusters, uint32_t max_iters,
                     uint64_t seed, uint32_t workers, pqv_index **out) {
    using namespace pqv;
    const uint64_t n = corpus->n;
    const uint32_t dim = corpus->dim;
    if (max_iters == 0) return fail(PQV_ERR_INVALID, "max_iters must be > 0");         // parquet.rs:90
    if (n == 0) return fail(PQV_ERR_INVALID, "Cannot build IVF index with zero vectors"); // index.rs:158
    uint64_t k = n_clusters;
    if (k == 0) k = static_cast<uint64_t>(std::ceil(std::sqrt(static_cast<double>(n)))); // :164
    if (k > n) return fail(PQV_ERR_INVALID, "n_clusters cannot exceed number of vectors"); // :169
    if (k > 0xFFFFFFFFull) return fail(PQV_ERR_INVALID, "Cluster count must fit in u32");
    if (!corpus->d_rows) return fail(PQV_ERR_INVALID, "corpus row-order copy was released");
    if (int rc = use_device(corpus->device)) return rc;
    hipStream_t stream = corpus->stream;
    g_build_stats[8] = 0.0; g_build_stats[9] = 0.0;      // (kernel seconds / launches of THIS build's final assignment, whichever path it takes)

    uint64_t sample_size = std::max<uint64_t>(n / 20, 1);                              // :172
    sample_size = std::min<uint64_t>(sample_size, 100000);                             // :173
    sample_size = std::min<uint64_t>(std::max<uint64_t>(sample_size, k), n);           // :174

    const double t_b0 = now_s();
    DevBuf d_centroids, d_sample, d_idx;
    HIP_TRY(d_centroids.alloc(k * dim * sizeof(float)));
    const float *d_train = corpus->d_rows;
    if (sample_size != n) {                                                            // :182-187
        StdRng rng = StdRng::seed_from_u64(seed);                                      // :231
        std::vector<uint64_t> idx = index_sample(rng, n, sample_size);                 // :232
        if (verbose()) std::fprintf(stderr, "[pqv] build: sample indices drawn at %.1f ms\n", (now_s() - t_b0) * 1e3);
        HIP_TRY(d_idx.alloc(sample_size * sizeof(uint64_t)));
        HIP_TRY(d_sample.alloc(sample_size * dim * sizeof(float)));
        HIP_TRY(hipMemcpyAsync(d_idx.p, idx.data(), sample_size * sizeof(uint64_t), hipMemcpyHostToDevice, stream));
        HIP_TRY(launch_gather_rows(corpus->d_rows, nullptr, d_idx.as<uint64_t>(), sample_size, dim,
                                   d_sample.as<float>(), stream));
        HIP_TRY(hipStreamSynchronize(stream));
        d_train = d_sample.as<float>();
    }
    const double t_b1 = now_s();
    if (int rc = kmeans_device(d_train, sample_size, dim, static_cast<uint32_t>(k), max_iters, seed,
                               workers, stream, d_centroids.as<float>(), nullptr, nullptr))
        return rc;
    const double t_b2 = now_s();
    d_sample.release(); d_idx.release();
    if (verbose()) std::fprintf(stderr, "[pqv] build: training sample %.3f s, k-means call %.3f s (k-means++ %.3f + Lloyd %.3f inside)\n",
                                t_b1 - t_b0, t_b2 - t_b1, g_build_stats[0], g_build_stats[1]);

    // final assignment of every row (:189-206)
    const double t_fa0 = now_s();
    DevBuf d_cluster;
    HIP_TRY(d_cluster.alloc(n * sizeof(uint32_t)));
    // The two host arrays of n row ids (the downloaded assignment and the lists) are allocated and first-touched by a
    // helper thread while the device assigns: 2 x 40 MB of page faults on C3 that used to follow the kernels.
    std::vector<uint32_t> cluster_of, rows_buf;
    const bool dev_lists = DeviceLists::applicable(n, static_cast<uint32_t>(k));      // (the lists sorted on the device: no assignment download)
    struct Joiner { std::thread t; ~Joiner() { if (t.joinable()) t.join(); } } warm;
    if (n >= (1u << 20)) {
        try {
            warm.t = std::thread([&cluster_of, &rows_buf, n, dev_lists] {
                try { if (!dev_lists) { cluster_of.resize(n); rows_buf.resize(n); } } catch (...) { }
            });
        } catch (const std::system_error &) { }
    }
    bool exact_assign = true, downloaded = false;
    double assign_form = 0.0;
    if (GemmAssign::applicable(dim, static_cast<uint32_t>(k))) {
        GemmAssign gemm;
        bool bad_c = false, bad_r = false;
        if (int rc = gemm.set_centroids(d_centroids.as<float>(), static_cast<uint32_t>(k), dim, stream, &bad_c)) return rc;
        if (verbose()) std::fprintf(stderr, "[pqv] final assignment: centroids set at %.1f ms\n", (now_s() - t_fa0) * 1e3);
        if (!bad_c) {
            // (chunk-wise downloads behind the next chunk's kernels were measured and dropped: a device-to-pageable copy on a
            //  second stream stalls the compute stream's queue -- 53 ms for the loop against 34 ms of kernels on C3)
            gemm.time_kernels = true;
            if (int rc = gemm.run(corpus->d_rows, n, d_cluster.as<uint32_t>(), stream, &bad_r, nullptr)) return rc;
            exact_assign = bad_r;
            downloaded = false;
        }
        HIP_TRY(hipStreamSynchronize(stream));     // the context's buffers are released at scope exit
        if (!exact_assign) assign_form = 2.0;
        if (verbose()) std::fprintf(stderr, "[pqv] final assignment: f16 path done at %.1f ms\n", (now_s() - t_fa0) * 1e3);
    }
    if (exact_assign && ScreenedAssign::applicable(dim, static_cast<uint32_t>(k))) {
        ScreenedAssign screen;
        bool bad_c = false, bad_r = false;
        if (int rc = screen.set_centroids(d_centroids.as<float>(), static_cast<uint32_t>(k), dim, stream, &bad_c)) return rc;
        if (!bad_c) {
            if (int rc = screen.run(corpus->d_rows, n, d_cluster.as<uint32_t>(), stream, &bad_r)) return rc;
            exact_assign = bad_r;
        }
        HIP_TRY(hipStreamSynchronize(stream));     // the context's buffers are released at scope exit
    }
    if (exact_assign)
        HIP_TRY(launch_assign(corpus->d_rows, n, dim, d_centroids.as<float>(), static_cast<uint32_t>(k),
                              d_cluster.as<uint32_t>(), nullptr, nullptr, nullptr, stream));
    // the lists: sorted on the device (the rows of a list ascending, index.rs:193-206), then ONE download of the sorted row ids
    // in place of the assignment's download + the host threads' counting sort
    DeviceLists dlists;
    DevBuf d_rows_sorted, d_off;
    if (dev_lists) {
        if (int rc = dlists.prepare(n, static_cast<uint32_t>(k))) return rc;
        HIP_TRY(d_rows_sorted.alloc(n * sizeof(uint32_t)));
        HIP_TRY(d_off.alloc((k + 1) * sizeof(uint64_t)));
        if (int rc = dlists.run(d_cluster.as<uint32_t>(), n, static_cast<uint32_t>(k), d_off.as<uint64_t>(), d_rows_sorted.as<uint32_t>(), stream))
            return rc;
    }
    if (verbose()) { HIP_TRY(hipStreamSynchronize(stream)); std::fprintf(stderr, "[pqv] final assignment: lists sorted at %.1f ms\n", (now_s() - t_fa0) * 1e3); }
    if (warm.t.joinable()) warm.t.join();
    if (verbose()) std::fprintf(stderr, "[pqv] final assignment: host arrays ready at %.1f ms\n", (now_s() - t_fa0) * 1e3);
    if (!dev_lists) cluster_of.resize(n);
    pqv_index *idx = new (std::nothrow) pqv_index();
    if (!idx) return fail(PQV_ERR_OOM, "host allocation failed");
    idx->list_rows = std::move(rows_buf);
    idx->dim = dim; idx->n_clusters = static_cast<uint32_t>(k);
    idx->centroids.resize(k * dim);
    uint32_t h_bad = 0;
    hipError_t e = hipSuccess;
    static const bool keep_dev = [] { const char *e = std::getenv("PQV_KEEP_DEVICE_LISTS"); return !(e && *e == '0'); }();
    if (dev_lists) {
        idx->list_off.resize(k + 1);
        if (!keep_dev) {                            // (the lists are not left on the device: downloaded here)
            idx->list_rows.resize(n);
            e = hipMemcpyAsync(idx->list_rows.data(), d_rows_sorted.p, n * sizeof(uint32_t), hipMemcpyDeviceToHost, stream);
        }
        if (e == hipSuccess) e = hipMemcpyAsync(idx->list_off.data(), d_off.p, (k + 1) * sizeof(uint64_t), hipMemcpyDeviceToHost, stream);
        if (e == hipSuccess) e = hipMemcpyAsync(&h_bad, dlists.bad.p, sizeof(uint32_t), hipMemcpyDeviceToHost, stream);
    } else if (!downloaded) {
        e = hipMemcpyAsync(cluster_of.data(), d_cluster.p, n * sizeof(uint32_t), hipMemcpyDeviceToHost, stream);
    }
    if (e == hipSuccess)
        e = hipMemcpyAsync(idx->centroids.data(), d_centroids.p, k * dim * sizeof(float), hipMemcpyDeviceToHost, stream);
    if (e == hipSuccess) e = hipStreamSynchronize(stream);
    if (e != hipSuccess) {
        delete idx;
        return fail(PQV_ERR_HIP, std::string("final assignment: ") + hipGetErrorString(e));
    }
    const double t_fa1 = now_s();
    if (dev_lists ? h_bad != 0 : !lists_from_assignment(cluster_of.data(), n, idx->n_clusters, idx->list_off, idx->list_rows)) {
        delete idx;
        return fail(PQV_ERR_HIP, "internal error: final assignment out of range");
    }
    if (dev_lists && keep_dev && d_rows_sorted.p) {
        // The sorted row ids stay where they are: a searcher on this device copies them device to device, and the host copy (40 MB per
        // 10 M rows: 3.5 ms of download) is made by the first call that reads it (ListRows::get).  PQV_KEEP_DEVICE_LISTS=0: downloaded above.
        auto dr = std::make_shared<DevRows>();
        dr->p = d_rows_sorted.p; dr->device = corpus->device; dr->n = n;
        d_rows_sorted.p = nullptr; d_rows_sorted.bytes = 0;
        idx->rows->dev = std::move(dr);
        idx->rows->n = n;
        idx->rows->host.clear();
        idx->rows->pending = true;
    }
    idx->permutation_of = n;                        // every row was assigned exactly once
    g_build_stats[3] = t_fa1 - t_fa0; g_build_stats[4] = now_s() - t_fa1; g_build_stats[5] = assign_form > 0.0 ? assign_form : exact_assign ? 0.0 : 1.0;
    g_build_stats[7] = static_cast<double>(sample_size);
    if (verbose()) std::fprintf(stderr, "[pqv] final assignment: %llu rows in %.3f s (+ %.3f s host list build)\n",
                                (unsigned long long)n, t_fa1 - t_fa0, now_s() - t_fa1);
    if (verbose()) {
        std::fprintf(stderr, "[pqv] build: %.3f s in all; hipMalloc %u calls %.1f ms, hipFree %u calls %.1f ms (this thread, since the last report)\n",
                     now_s() - t_b0, g_alloc_n, g_alloc_s * 1e3, g_free_n, g_free_s * 1e3);
        g_alloc_s = g_free_s = 0.0; g_alloc_n = g_free_n = 0;
    }
    *out = idx;
    return PQV_OK;
}

}  // namespace

static int pqv_index_build_impl(const pqv_corpus *corpus, uint32_t n_clusters, uint32_t max_iters,
                               uint64_t seed, uint32_t workers, pqv_index **out) {
    if (!out) return fail(PQV_ERR_INVALID, "out must not be NULL");
    *out = nullptr;
    if (!corpus) return fail(PQV_ERR_INVALID, "corpus must not be NULL");
    return build_index_impl(corpus, n_clusters, max_iters, seed, workers, out);
}
extern "C" int pqv_index_build(const pqv_corpus *corpus, uint32_t n_clusters, uint32_t max_iters,
                               uint64_t seed, uint32_t workers, pqv_index **out) {
    return guard([&] { return pqv_index_build_impl(corpus, n_clusters, max_iters, seed, workers, out); });
}

static int pqv_kpp_pick_impl(int device, const float *minima, uint32_t n, uint32_t workers, float draw, uint64_t *pick, float *total,
                             uint32_t *status) {
    using namespace pqv;
    if (!minima || !pick || !total || !status) return fail(PQV_ERR_INVALID, "minima, pick, total and status must not be NULL");
    if (n == 0 || n > 57344) return fail(PQV_ERR_INVALID, "n must be in [1, 57344]");
    if (int rc = use_device(device)) return rc;
    if (workers == 0) workers = host_workers();
    const uint64_t w = std::max<uint64_t>(1, std::min<uint64_t>(workers, n));          // index.rs:259-265
    const uint64_t chunk = (n + w - 1) / w, n_chunks = (n + chunk - 1) / chunk;
    if (n_chunks > 1024) return fail(PQV_ERR_INVALID, "more than 1024 worker chunks");
    const size_t n64 = 2 + n_chunks + 64 + 4096 + 64 + 72;
    const size_t bytes = n64 * 8 + 4 * 4 + 4 * 4 + 16 + static_cast<size_t>(n) * 4;
    DevBuf buf;
    HIP_TRY(buf.alloc(bytes));
    HIP_TRY(hipMemset(buf.p, 0, bytes));
    unsigned long long *d_picks = buf.as<unsigned long long>();
    float *d_u = reinterpret_cast<float *>(d_picks + n64);
    uint32_t *d_state = reinterpret_cast<uint32_t *>(d_u + 4);
    float *d_md = reinterpret_cast<float *>((reinterpret_cast<uintptr_t>(d_state + 4) + 15) & ~static_cast<uintptr_t>(15));   // 16-byte aligned, as the build's minima are
    const float u2[2] = {0.0f, draw};
    HIP_TRY(hipMemcpy(d_u, u2, sizeof u2, hipMemcpyHostToDevice));
    HIP_TRY(hipMemcpy(d_md, minima, static_cast<size_t>(n) * 4, hipMemcpyHostToDevice));
    KppPickArgs pa{};
    pa.md = d_md; pa.n = n; pa.chunk = static_cast<uint32_t>(chunk); pa.n_chunks = static_cast<uint32_t>(n_chunks);
    pa.chunk_sum = d_picks + 2; pa.blk_sum = pa.chunk_sum + n_chunks; pa.run_sum = pa.blk_sum + 64; pa.blk_done = pa.run_sum + 4096; pa.head = pa.blk_done + 64;
    pa.u = d_u; pa.round = 1; pa.picks = d_picks; pa.state = d_state;
    DevBuf d_stamps;
    const char *want_stamps = std::getenv("PQV_KPP_STAMPS");
    if (want_stamps && *want_stamps == '1') {
        HIP_TRY(d_stamps.alloc(48 * 8));
        HIP_TRY(hipMemset(d_stamps.p, 0, 48 * 8));
        pa.stamps = d_stamps.as<unsigned long long>();
    }
    HIP_TRY(launch_kpp_pick(pa, nullptr));
    HIP_TRY(hipDeviceSynchronize());
    if (pa.stamps) {
        unsigned long long h[48];
        HIP_TRY(hipMemcpy(h, d_stamps.p, sizeof h, hipMemcpyDeviceToHost));
        unsigned long long t0 = ~0ull;
        for (int i = 0; i < 40; ++i) if (h[i] && h[i] < t0) t0 = h[i];
        std::fprintf(stderr, "[pqv] kpp stamps (us after the first; chain block 0-7: start, loaded, summaries there, pairs read, chain done, total, found, picked;"
                             " last summary block 8-12: start, sum out, prefix in, pairs, flag out; chunk block 0 16-17; end of wave w's turn 20+w):");
        for (int i = 0; i < 40; ++i) if (h[i]) std::fprintf(stderr, " %d:%.2f", i, (h[i] - t0) * 0.01);
        std::fprintf(stderr, "\n");
    }
    uint32_t h_state[4];
    unsigned long long h_pick[2];
    HIP_TRY(hipMemcpy(h_state, d_state, sizeof h_state, hipMemcpyDeviceToHost));
    HIP_TRY(hipMemcpy(h_pick, d_picks, sizeof h_pick, hipMemcpyDeviceToHost));
    *status = h_state[0] ? h_state[2] : 0;
    std::memcpy(total, &h_state[3], 4);
    if (!h_state[0]) *pick = h_pick[1];
    return PQV_OK;
}
extern "C" int pqv_kpp_pick(int device, const float *minima, uint32_t n, uint32_t workers, float draw, uint64_t *pick, float *total,
                            uint32_t *status) {
    return guard([&] { return pqv_kpp_pick_impl(device, minima, n, workers, draw, pick, total, status); });
}

extern "C" int pqv_index_build_stats(double *out, uint32_t n) {
    if (!out) return fail(PQV_ERR_INVALID, "out must not be NULL");
    for (uint32_t i = 0; i < n; ++i) out[i] = i < 10 ? g_build_stats[i] : 0.0;
    return PQV_OK;
}

static int pqv_index_build_host_impl(int device, const float *data, uint64_t data_len, uint32_t dim,
                                    uint32_t n_clusters, uint32_t max_iters, uint64_t seed,
                                    uint32_t workers, pqv_index **out) {
    if (!out) return fail(PQV_ERR_INVALID, "out must not be NULL");
    *out = nullptr;
    if (dim == 0) return fail(PQV_ERR_INVALID, "Embedding dimension must be > 0");     // mod.rs:59
    if (data_len % dim != 0)
        return fail(PQV_ERR_INVALID, "Embedding data length must be a multiple of dimension"); // mod.rs:86
    if (max_iters == 0) return fail(PQV_ERR_INVALID, "max_iters must be > 0");
    const uint64_t n = data_len / dim;
    if (n == 0) return fail(PQV_ERR_INVALID, "Cannot build IVF index with zero vectors");
    pqv_corpus *c = nullptr;
    if (int rc = pqv_corpus_upload(device, data, n, dim, &c)) return rc;
    const int rc = build_index_impl(c, n_clusters, max_iters, seed, workers, out);
    pqv_corpus_free(c);
    return rc;
}
extern "C" int pqv_index_build_host(int device, const float *data, uint64_t data_len, uint32_t dim,
                                    uint32_t n_clusters, uint32_t max_iters, uint64_t seed,
                                    uint32_t workers, pqv_index **out) {
    return guard([&] { return pqv_index_build_host_impl(device, data, data_len, dim, n_clusters, max_iters, seed, workers, out); });
}

static int pqv_kmeans_impl(const pqv_corpus *sample, uint32_t k, uint32_t max_iters, uint64_t seed,
                          uint32_t workers, float *centroids, uint32_t *assignments,
                          uint32_t *iters_run) {
    if (!sample || !centroids) return fail(PQV_ERR_INVALID, "sample/centroids must not be NULL");
    if (k == 0) return fail(PQV_ERR_INVALID, "Cluster count must be > 0");
    if (sample->n == 0 || k > sample->n) return fail(PQV_ERR_INVALID, "n_clusters cannot exceed number of vectors");
    if (!sample->d_rows) return fail(PQV_ERR_INVALID, "corpus row-order copy was released");
    if (int rc = use_device(sample->device)) return rc;
    DevBuf d_centroids;
    HIP_TRY(d_centroids.alloc(static_cast<size_t>(k) * sample->dim * sizeof(float)));
    std::vector<uint32_t> assign;
    if (int rc = kmeans_device(sample->d_rows, sample->n, sample->dim, k, max_iters, seed, workers,
                               sample->stream, d_centroids.as<float>(), &assign, iters_run))
        return rc;
    HIP_TRY(hipMemcpy(centroids, d_centroids.p, static_cast<size_t>(k) * sample->dim * sizeof(float),
                      hipMemcpyDeviceToHost));
    if (assignments) std::memcpy(assignments, assign.data(), assign.size() * sizeof(uint32_t));
    return PQV_OK;
}
extern "C" int pqv_kmeans(const pqv_corpus *sample, uint32_t k, uint32_t max_iters, uint64_t seed,
                          uint32_t workers, float *centroids, uint32_t *assignments,
                          uint32_t *iters_run) {
    return guard([&] { return pqv_kmeans_impl(sample, k, max_iters, seed, workers, centroids, assignments, iters_run); });
}

// ---------------------------------------------------------------------------------------
// searcher
// ---------------------------------------------------------------------------------------
namespace {

void opts_from_env(pqv_searcher::Opts &o) {
    auto num = [](const char *name, long long dflt) { const char *e = std::getenv(name); return e && *e ? std::strtoll(e, nullptr, 10) : dflt; };
    if (const char *m = std::getenv("PQV_RERANK_MODE")) o.rerank_mode = !std::strcmp(m, "stream") ? 1 : !std::strcmp(m, "tile") ? 2 : 0;
    o.tile_filter = static_cast<int>(std::min<long long>(2, std::max<long long>(0, num("PQV_TILE_FILTER", o.tile_filter))));
    o.filter_variant = static_cast<int>(num("PQV_FILTER_VARIANT", o.filter_variant));
    o.cand_cap = static_cast<uint32_t>(std::max<long long>(0, num("PQV_CAND_CAP", o.cand_cap)));
    o.screen_f16 = num("PQV_SCREEN_F16", o.screen_f16) != 0;
    o.screen_i8 = static_cast<int>(num("PQV_SCREEN_I8", o.screen_i8));
    o.seed_rows = static_cast<uint32_t>(num("PQV_SEED_ROWS", o.seed_rows));
    o.wide_rows = static_cast<uint32_t>(num("PQV_WIDE_ROWS", o.wide_rows));
    o.tile_rows = static_cast<uint32_t>(num("PQV_TILE_ROWS", o.tile_rows));
    o.running_thr = num("PQV_RUNNING_THR", o.running_thr) != 0;
    o.defer = static_cast<int>(num("PQV_DEFER", o.defer));
    o.quad_xcd = static_cast<int>(num("PQV_QUAD_XCD", o.quad_xcd));
    o.wide_waves = static_cast<int>(num("PQV_WIDE_WAVES", o.wide_waves));
    o.item_grid = static_cast<int>(num("PQV_ITEM_GRID", o.item_grid));
    o.chunk_major = static_cast<int>(num("PQV_CHUNK_MAJOR", o.chunk_major));
    o.seed_refine = static_cast<int>(num("PQV_SEED_REFINE", o.seed_refine));
    o.single_bucket = static_cast<int>(num("PQV_SINGLE_BUCKET", o.single_bucket));
    o.probe_rows = static_cast<int>(num("PQV_PROBE_ROWS", o.probe_rows));
    o.quad_width = static_cast<uint32_t>(num("PQV_QUAD_WIDTH", o.quad_width));
    o.min_blocks = static_cast<uint32_t>(num("PQV_MIN_BLOCKS", o.min_blocks));
    o.wide_quads = static_cast<int>(num("PQV_WIDE_QUADS", o.wide_quads));
    o.fork_wide = static_cast<int>(num("PQV_FORK_WIDE", o.fork_wide));
    o.drain_min = static_cast<int>(num("PQV_DRAIN_MIN", o.drain_min));
    o.pf96 = static_cast<int>(num("PQV_PF96", o.pf96));
    o.xcd_items = static_cast<int>(num("PQV_XCD_ITEMS", o.xcd_items));
    o.wide_quad_rows = static_cast<uint32_t>(num("PQV_WIDE_QUAD_ROWS", o.wide_quad_rows));
    o.list_once = static_cast<int>(num("PQV_LIST_ONCE", o.list_once));
    o.pair_prune = static_cast<int>(num("PQV_PAIR_PRUNE", o.pair_prune));
    o.i8_form = static_cast<int>(num("PQV_I8_FORM", o.i8_form));
}

// Cache policy of the wide-quad instance's row stream (kernels_screen.hip, launch_filter_s): nt unless a recent batch had more than a
// quarter of its wide work items in lists of several quads (those share their rows through the caches).  A hint only: it
// is read without synchronisation and never changes a result; before the first batch has reported: nt.
bool wide_rows_nt(const pqv_searcher *s) {
    if (!s->h_wide_stats.p) return true;
    const volatile uint32_t *h = s->h_wide_stats.as<uint32_t>();
    const uint32_t total = h[0], lone = h[1];
    return total == 0 || static_cast<uint64_t>(lone) * 4 >= static_cast<uint64_t>(total) * 3;
}

// Wide-quad instance or regular quads only?  A list probed by 97..160 queries is read ONCE by a wide quad (32-row tiles, one
// 8-wave block per CU, 4.7 TB/s) instead of twice by two regular quads -- uniform C3: 585 against 549 k q/s.  A list probed by
// HUNDREDS of queries (clustered queries) is read by several quads either way; the regular ones then share its rows through one
// XCD's L2 (PairSortArgs::xcd_items) at the regular instance's rate -- mixture: kernels 2.13 against 2.34 ms.  The previous
// batch's shape decides (pair_scan_kernel -> pinned memory; an unsynchronised hint that never changes a result): regular only
// when more than half of the rows in lists of > 96 pairs sit in lists of > 160 (and such lists hold >= 1 / 32 of the corpus).
bool prefer_regular(const pqv_searcher *s) {
    if (!s->h_wide_stats.p) return false;
    const volatile uint32_t *h = s->h_wide_stats.as<uint32_t>();
    const uint32_t multi = h[2], pop = h[3];
    // (... and those popular lists are a real part of the corpus -- 1 / 32 of its rows: a uniform batch on short lists has a
    //  handful of them, and their split says nothing about the query load)
    return static_cast<uint64_t>(pop) * 32 >= s->n && static_cast<uint64_t>(multi) * 2 > pop;
}

// the wide screened kernels need IVF-ordered rows of a multiple of 64 dims
bool wide_path_possible(const pqv_searcher *s) { return (s->sdim % 64) == 0 && (!s->d_row_of || s->images_only) && s->n > 0; }

// Operand form of the screen for this searcher (pqv::ScreenOp numbering: 0 f32, 1 f16, 2 int8)
// int8 images halve the bytes per streamed row but widen the bound (residual norms): where lists are short and k is
// large the extra exact evaluations cost more than the stream saves -- measured on the reference's bench shape (1 M x
// 1024, 1000-row lists): K = 100 int8 2.93 against f16 2.31 ms per step, K = 10 equal; C3 (9766-row lists) K = 10
// 2.44 against 3.85, K = 100 4.21 against 5.10; 1 M x 768 (976-row lists) K = 10 1.07 against 1.16.
// squared norms of the rows for the f32 / f16 MFMA screens (indexed by storage row; by list position in the images-only layout) and, in
// the same pass, the corpus maximum -> power-of-two scale that maps it below 2^14 (f16 operand copy).  Once per searcher.
int ensure_row_norms(const pqv_searcher *s, hipStream_t stream) {
    std::lock_guard<std::mutex> lock(s->norms_mu);
    if (s->norms_done) return PQV_OK;
    DevBuf d_max;
    HIP_TRY(d_max.alloc(sizeof(uint32_t)));
    HIP_TRY(hipMemsetAsync(d_max.p, 0, sizeof(uint32_t), stream));
    HIP_TRY(s->d_row_norm2.alloc(std::max<uint64_t>(1, s->n_storage) * sizeof(float)));
    HIP_TRY(pqv::launch_row_norms_max(s->d_mat, s->images_only ? s->d_row_of : nullptr, s->n_storage, s->sdim, s->d_row_norm2.as<float>(),
                                      d_max.as<uint32_t>(), stream));
    uint32_t bits = 0;
    HIP_TRY(hipMemcpyAsync(&bits, d_max.p, sizeof bits, hipMemcpyDeviceToHost, stream));
    HIP_TRY(hipStreamSynchronize(stream));
    float m;
    std::memcpy(&m, &bits, sizeof m);
    if (bits < 0x7F800000u) {
        int e = 0;
        if (m > 0.0f) (void)std::frexp(m, &e);           // m = f * 2^e, f in [0.5, 1)  =>  m < 2^e
        const int se = m > 0.0f ? 14 - e : 0;            // scale = 2^se: m * scale < 2^14
        if (se >= -60 && se <= 60) { s->f16_ok = true; s->f16_scale = std::ldexp(1.0f, se); }
    }
    s->norms_done = true;
    return PQV_OK;
}

constexpr int kNotFinite = -9001;      // ensure_blocked_copy(op 2) at creation: the rows are not all finite and moderate (the caller falls back)

int screen_op(const pqv_searcher *s, uint32_t k = 1) {
    const uint64_t mean_len = s->n / std::max<uint32_t>(1, s->n_clusters);
    const bool i8_pays = !(k > 32 && mean_len < 4096) || s->opt.screen_i8 > 1;
    if (s->opt.screen_i8 && i8_pays && s->i8_ok && (s->sdim % 256) == 0 && s->sdim >= 256 && static_cast<uint64_t>(64) * s->sdim <= 147456) return 2;
    if (!s->norms_done && ensure_row_norms(s, s->stream) != PQV_OK) return 0;      // (f16_ok is known only after that pass)
    if (s->opt.screen_f16 && s->f16_ok && (s->sdim % 128) == 0 && s->sdim <= 1024) return 1;
    return 0;
}

// One-off: the blocked MFMA-operand copy of the lists (a second copy of the corpus in HBM: f16 half its size,
// int8 a quarter)
int ensure_blocked_copy(const pqv_searcher *s, int op, hipStream_t stream) {
    using namespace pqv;
    DevBuf &blk = s->d_mat_blk_op[op];
    if (blk.p) return PQV_OK;
    // A copy counts as built only once its LAST step has succeeded: any failure on the way (an allocation, a launch, a
    // synchronisation) releases it, so that the next call rebuilds instead of screening against half-written images, row
    // terms or list scales (the bound would no longer hold and true neighbours could be pruned silently).
    struct Undo {
        const pqv_searcher *s; DevBuf &blk; int op; bool done = false;
        ~Undo() {
            if (done) return;
            (void)hipDeviceSynchronize();       // nothing still in flight may write into what is released
            blk.release();
            if (op == 2) {
                s->d_row_n2i.release(); s->d_row_res.release(); s->d_center.release(); s->d_list_scale.release();
                s->d_list_half.release(); s->d_list_radius.release(); s->i8_residual = true;
            }
        }
    } undo{s, blk, op};
    const uint32_t kc = s->n_clusters;
    std::vector<uint64_t> boff(static_cast<size_t>(kc) + 1, 0);
    for (uint32_t c = 0; c < kc; ++c) boff[c + 1] = boff[c] + (s->h_list_off[c + 1] - s->h_list_off[c] + 15) / 16;
    if (!s->d_blk_off.p) {
        HIP_TRY(s->d_blk_off.alloc(boff.size() * sizeof(uint64_t)));
        const hipError_t ce = hipMemcpy(s->d_blk_off.p, boff.data(), boff.size() * sizeof(uint64_t), hipMemcpyHostToDevice);
        if (ce != hipSuccess) { s->d_blk_off.release(); HIP_TRY(ce); }
    }
    const uint64_t tiles = std::max<uint64_t>(1, boff[kc]);
    if (op != 2) { if (int rc = ensure_row_norms(s, stream)) return rc; }
    if (op == 2) {
        HIP_TRY(blk.alloc(tiles * 16 * s->sdim));
        HIP_TRY(s->d_row_n2i.ensure(std::max<uint64_t>(1, s->n) * sizeof(int)));
        HIP_TRY(s->d_row_res.ensure(std::max<uint64_t>(1, s->n) * sizeof(float)));
        // per-list centre (mid-range per dimension), half range, scale; then the images, row terms and list radii
        const size_t cd = static_cast<size_t>(std::max<uint32_t>(1, kc)) * s->sdim;
        DevBuf d_mm;
        HIP_TRY(d_mm.alloc(2 * cd * sizeof(uint32_t)));
        uint32_t *kmin = d_mm.as<uint32_t>(), *kmax = kmin + cd;
        HIP_TRY(hipMemsetAsync(kmin, 0xFF, cd * sizeof(uint32_t), stream));
        HIP_TRY(hipMemsetAsync(kmax, 0, cd * sizeof(uint32_t), stream));
        HIP_TRY(s->d_center.alloc(cd * sizeof(float)));
        HIP_TRY(s->d_list_scale.alloc(std::max<uint32_t>(1, kc) * sizeof(float)));
        HIP_TRY(s->d_list_half.alloc(std::max<uint32_t>(1, kc) * sizeof(float)));
        HIP_TRY(s->d_list_radius.alloc(std::max<uint32_t>(1, kc) * sizeof(float)));
        HIP_TRY(launch_list_minmax(s->d_mat, s->d_list_off.as<uint64_t>(), kc, s->max_list_len, s->sdim, kmin, kmax, stream, s->d_row_of));
        HIP_TRY(launch_list_center(kmin, kmax, kc, s->sdim, s->d_list_off.as<uint64_t>(), s->d_center.as<float>(), s->d_list_half.as<float>(),
                                   s->d_list_scale.as<float>(), s->d_list_radius.as<float>(), stream));
        {   // residual or one-centre form: compare the lists' scales with the scale ONE centre for the whole corpus would get
            DevBuf d_g;
            HIP_TRY(d_g.alloc((static_cast<size_t>(s->sdim) + 2) * sizeof(float)));
            float *g_center = d_g.as<float>(), *g_hs = g_center + s->sdim;
            HIP_TRY(launch_global_center(kmin, kmax, kc, s->sdim, s->d_list_off.as<uint64_t>(), g_center, g_hs, stream));
            std::vector<float> h_scale(std::max<uint32_t>(1, kc));
            float h_g[2] = {0.0f, 1.0f};
            std::vector<float> h_gc(s->norms_done ? 0 : s->sdim);
            HIP_TRY(hipMemcpyAsync(h_scale.data(), s->d_list_scale.p, static_cast<size_t>(kc) * sizeof(float), hipMemcpyDeviceToHost, stream));
            HIP_TRY(hipMemcpyAsync(h_g, g_hs, sizeof h_g, hipMemcpyDeviceToHost, stream));
            if (!h_gc.empty()) HIP_TRY(hipMemcpyAsync(h_gc.data(), g_center, h_gc.size() * sizeof(float), hipMemcpyDeviceToHost, stream));
            HIP_TRY(hipStreamSynchronize(stream));
            if (!s->norms_done) {
                // No pass over the rows' norms has vouched for the data (creation builds this copy first): the per-dimension extremes
                // do -- every |x_d| <= |mid-range_d| + half range =: m, at most twice the true maximum.  ensure_row_norms' rule is
                // "finite, and 2^-46 <= max |x_d| < 2^74"; anything within a factor of two of those ends, an infinity or a NaN (it
                // shows in the extremes) sends the caller to that pass, which decides exactly.
                float m = 0.0f;
                bool fin = h_g[0] == h_g[0] && h_g[0] < 1.0e30f;
                for (float c : h_gc) { fin = fin && c == c; m = std::max(m, std::fabs(c)); }
                m += h_g[0];
                if (verbose()) std::fprintf(stderr, "[pqv] int8 copy first: half range %.4g, largest |value| <= %.4g, finite %d\n", h_g[0], m, fin ? 1 : 0);
                if (!fin || !(m < 9.0e21f) || !(m > 3.0e-14f)) { s->i8_ok = false; return kNotFinite; }
            }
            // row-weighted median of the list scales (an empty or one-point list says nothing about the data)
            std::vector<std::pair<float, uint64_t>> sc;
            uint64_t rows_total = 0;
            for (uint32_t c = 0; c < kc; ++c) {
                const uint64_t len = s->h_list_off[c + 1] - s->h_list_off[c];
                if (len >= 2) { sc.emplace_back(h_scale[c], len); rows_total += len; }
            }
            std::sort(sc.begin(), sc.end());
            float med = h_g[1];
            uint64_t acc = 0;
            for (const auto &e : sc) { acc += e.second; if (2 * acc >= rows_total) { med = e.first; break; } }
            const int form = s->opt.i8_form;
            s->i8_residual = form == 2 || (form != 1 && med >= 1.3f * h_g[1]);
            if (!s->i8_residual)
                HIP_TRY(launch_broadcast_center(g_center, g_hs, kc, s->sdim, s->d_center.as<float>(), s->d_list_half.as<float>(),
                                                s->d_list_scale.as<float>(), stream));
            if (verbose()) std::fprintf(stderr, "[pqv] int8 images: median list scale %.4g, one-centre scale %.4g -> %s form\n", med, h_g[1],
                                        s->i8_residual ? "per-list residual" : "one-centre");
            HIP_TRY(hipStreamSynchronize(stream));      // d_g is released at scope exit
        }
        HIP_TRY(launch_block_rows_i8(s->d_mat, s->d_list_off.as<uint64_t>(), s->d_blk_off.as<uint64_t>(), kc,
                                     (s->max_list_len + 15) / 16, s->sdim, s->d_center.as<float>(), s->d_list_scale.as<float>(),
                                     s->d_list_half.as<float>(), s->d_list_radius.as<float>(), blk.p, s->d_row_n2i.as<int>(),
                                     s->d_row_res.as<float>(), stream, s->d_row_of));
        HIP_TRY(hipStreamSynchronize(stream));      // d_mm is released at scope exit
    } else if (op == 1) {
        HIP_TRY(blk.alloc(tiles * 16 * s->sdim * 2));
        HIP_TRY(launch_block_rows_f16(s->d_mat, s->d_list_off.as<uint64_t>(), s->d_blk_off.as<uint64_t>(), kc,
                                      (s->max_list_len + 15) / 16, s->sdim, s->f16_scale, blk.p, stream, s->d_row_of));
    } else {
        HIP_TRY(blk.alloc(tiles * 16 * s->sdim * sizeof(float)));
        HIP_TRY(launch_block_rows(s->d_mat, s->d_list_off.as<uint64_t>(), s->d_blk_off.as<uint64_t>(), kc,
                                  (s->max_list_len + 15) / 16, s->sdim, blk.p, stream, s->d_row_of));
    }
    HIP_TRY(hipStreamSynchronize(stream));      // one-off; calls on other streams may follow at once
    undo.done = true;
    return PQV_OK;
}

}  // namespace

static int pqv_searcher_create_impl(const pqv_index *index, pqv_corpus *corpus, uint32_t flags,
                                   pqv_searcher **out) {
    if (!out) return fail(PQV_ERR_INVALID, "out must not be NULL");
    *out = nullptr;
    const auto t_enter = std::chrono::steady_clock::now();
    if (!index || !corpus) return fail(PQV_ERR_INVALID, "index/corpus must not be NULL");
    if (index->dim != corpus->dim)
        return fail(PQV_ERR_INVALID, "index dimension " + std::to_string(index->dim) +
                                         " does not match corpus dimension " + std::to_string(corpus->dim));
    if (index->permutation_of != 0 ? index->permutation_of != corpus->n : false)
        return fail(PQV_ERR_INVALID, "index row id out of range for this corpus");
    if (index->permutation_of == 0) {     // (a build's lists are a permutation of the corpus' rows; anything else is checked)
        const std::vector<uint32_t> *rv = index->rows->get();
        if (!rv) return PQV_ERR_HIP;
        for (uint32_t r : *rv)
            if (r >= corpus->n) return fail(PQV_ERR_INVALID, "index row id out of range for this corpus");
    }
    if (!corpus->d_rows) return fail(PQV_ERR_INVALID, "corpus row-order copy was released");
    if (int rc = use_device(corpus->device)) return rc;
    pqv_searcher *s = new (std::nothrow) pqv_searcher();
    if (!s) return fail(PQV_ERR_OOM, "host allocation failed");
    s->device = corpus->device; s->dim = index->dim; s->sdim = index->dim; s->n_clusters = index->n_clusters;
    if (!(flags & PQV_LAYOUT_ROW_ORDER) && (s->dim % 4) == 0 && (s->dim % 64) != 0) {
        // zero-padded storage: the cheapest tiling whose padding stays within a third of the row -- int8 images (a multiple
        // of 256 dims), else f16 (128), else f32 (64); rows of fewer than 48 dims stay as they are (exact kernels)
        auto up = [&](uint32_t m) { return (s->dim + m - 1) / m * m; };
        const uint64_t lim = static_cast<uint64_t>(s->dim) * 4 / 3;
        if (s->dim >= 192 && up(256) <= lim) s->sdim = up(256);
        else if (up(128) <= lim) s->sdim = up(128);
        else if (up(64) <= lim) s->sdim = up(64);
    }
    s->n = index->n_rows(); s->corpus = corpus;
    s->h_list_off = index->list_off; s->h_rows = index->rows;
    for (uint32_t c = 0; c < index->n_clusters; ++c)
        s->max_list_len = std::max<uint64_t>(s->max_list_len, index->list_off[c + 1] - index->list_off[c]);
    auto cleanup = [&](int code, const std::string &msg) { delete s; return fail(code, msg); };
#define S_TRY(expr)                                                                       \
    do {                                                                                  \
        hipError_t _e = (expr);                                                           \
        if (_e != hipSuccess)                                                             \
            return cleanup(_e == hipErrorOutOfMemory ? PQV_ERR_OOM : PQV_ERR_HIP,         \
                           std::string(#expr) + ": " + hipGetErrorString(_e));           \
    } while (0)
    opts_from_env(s->opt);
    // PQV_CREATE_STATS=1: wall time of the phases below on stderr (diagnostic; synchronises the stream at every mark)
    const bool create_stats = std::getenv("PQV_CREATE_STATS") != nullptr;
    auto t_last = t_enter;
    auto mark = [&](const char *what) {
        if (!create_stats) return;
        if (s->stream) (void)hipStreamSynchronize(s->stream);
        const auto now = std::chrono::steady_clock::now();
        std::fprintf(stderr, "pqv_searcher_create: %-28s %8.2f ms\n", what, std::chrono::duration<double, std::milli>(now - t_last).count());
        t_last = now;
    };
    mark("validation + host copies");
    S_TRY(pool_get(s->device, &s->stream));
    mark("stream");
    S_TRY(s->d_centroids.alloc(index->centroids.size() * sizeof(float)));
    S_TRY(s->d_list_off.alloc(index->list_off.size() * sizeof(uint64_t)));
    S_TRY(s->d_ids.alloc(std::max<size_t>(1, s->n) * sizeof(uint32_t)));
    mark("allocations");
    S_TRY(hipMemcpyAsync(s->d_centroids.p, index->centroids.data(), index->centroids.size() * sizeof(float),
                         hipMemcpyHostToDevice, s->stream));
    S_TRY(hipMemcpyAsync(s->d_list_off.p, index->list_off.data(), index->list_off.size() * sizeof(uint64_t),
                         hipMemcpyHostToDevice, s->stream));
    if ((s->dim % 4) == 0 && s->n_clusters) {
        s->kc_pad = (s->n_clusters + 255) / 256 * 256;
        S_TRY(s->d_cent_t.alloc(static_cast<size_t>(s->kc_pad) * s->dim * sizeof(float)));
        S_TRY(pqv::launch_transpose_rows4(s->d_centroids.as<float>(), s->n_clusters, s->kc_pad, s->dim, s->d_cent_t.p, s->stream));
    }
    mark("centroid tables");
    if (s->n) {
        const DevRows *dr = index->rows->dev.get();
        if (dr && dr->p && dr->device == s->device && dr->n == s->n) {      // the build's device copy
            S_TRY(hipMemcpyAsync(s->d_ids.p, dr->p, s->n * sizeof(uint32_t), hipMemcpyDeviceToDevice, s->stream));
        } else {
            const std::vector<uint32_t> *rv = index->rows->get();
            if (!rv) { delete s; return PQV_ERR_HIP; }
            S_TRY(hipMemcpyAsync(s->d_ids.p, rv->data(), s->n * sizeof(uint32_t), hipMemcpyHostToDevice, s->stream));
        }
    }
    mark("tables + ids upload");
    // images-only IVF layout: where the wide screened path will serve this searcher (rows of a multiple of 64 dims, lists of
    // >= 192 rows on average) and the caller's matrix stays (no PQV_RELEASE_ROW_ORDER); PQV_IVF_COPY=1 keeps the second copy (A/B)
    {
        const char *e = std::getenv("PQV_IVF_COPY");
        const bool want_copy = e && *e && *e != '0';
        s->images_only = !(flags & PQV_LAYOUT_ROW_ORDER) && !((flags & PQV_RELEASE_ROW_ORDER) && corpus->owned) && !want_copy && s->sdim == s->dim &&
                         (s->dim % 64) == 0 && s->n > 0 && s->n / std::max<uint32_t>(1, s->n_clusters) >= 192;
    }
    if ((flags & PQV_LAYOUT_ROW_ORDER) || s->images_only) {
        s->d_mat = corpus->d_rows;
        s->d_row_of = s->d_ids.as<uint32_t>();
        s->d_final_ids = nullptr;
    } else {
        S_TRY(s->d_mat_ivf.alloc(std::max<size_t>(1, s->n) * s->sdim * sizeof(float)));
        mark("allocate the IVF-ordered copy");
        if (s->sdim != s->dim)
            S_TRY(pqv::launch_pad_rows(corpus->d_rows, s->d_ids.as<uint32_t>(), s->n, s->dim, s->sdim, s->d_mat_ivf.as<float>(), s->stream));
        else
        S_TRY(pqv::launch_gather_rows(corpus->d_rows, s->d_ids.as<uint32_t>(), nullptr, s->n, s->dim,
                                      s->d_mat_ivf.as<float>(), s->stream));
        s->d_mat = s->d_mat_ivf.as<float>();
        s->d_row_of = nullptr;
        s->d_final_ids = s->d_ids.as<uint32_t>();
    }
    mark("gather rows into list order");
#ifdef PQV_PROFILE_PHASES
    S_TRY(s->d_stats.alloc((8 + 8 * 65536) * sizeof(unsigned long long)));
    S_TRY(hipMemsetAsync(s->d_stats.p, 0, (8 + 8 * 65536) * sizeof(unsigned long long), s->stream));
#else
    S_TRY(s->d_stats.alloc((8 + 16 * pqv::STATS_SLOTS) * sizeof(unsigned long long)));
    S_TRY(hipMemsetAsync(s->d_stats.p, 0, (8 + 16 * pqv::STATS_SLOTS) * sizeof(unsigned long long), s->stream));
#endif
    s->n_storage = (flags & PQV_LAYOUT_ROW_ORDER) ? corpus->n : s->n;
    // The blocked MFMA-operand copy of the lists is built here, not inside the first query, whenever the wide screened path can apply
    // to this searcher (ensure_blocked_copy rebuilds it if an option changes its form).  int8 form: rows of a multiple of 256 dims,
    // IVF-ordered rows, finite data.  Where that copy is built right away, ITS first pass (the per-dimension extremes) vouches for the
    // data and the rows' norms -- 5.5 ms of C3's creation, never read by the int8 screen -- wait for a call that needs them.
    const bool build_now = wide_path_possible(s) && s->n / std::max<uint32_t>(1, s->n_clusters) >= 192;
    const bool i8_shape = (s->sdim % 256) == 0 && !(flags & PQV_LAYOUT_ROW_ORDER) && s->n > 0;
    bool built = false;
    {
        static const bool eager_norms = [] { const char *e = std::getenv("PQV_EAGER_NORMS"); return e && *e == '1'; }();
        if (build_now && i8_shape && !eager_norms) {
            s->i8_ok = true;                                   // (tentatively: screen_op must see it)
            if (screen_op(s) == 2) {
                const int rc = ensure_blocked_copy(s, 2, s->stream);
                if (rc == PQV_OK) built = true;
                else if (rc != kNotFinite) { delete s; return rc; }
            }
            if (!built) s->i8_ok = false;
        }
    }
    mark("int8 copy first");
    if (!built) {
        if (int rc = ensure_row_norms(s, s->stream)) { delete s; return rc; }
        mark("row norms + maximum");
        s->i8_ok = s->f16_ok && i8_shape;
        if (build_now) {
            if (int rc = ensure_blocked_copy(s, screen_op(s), s->stream)) { delete s; return rc; }
        }
    }
    S_TRY(hipStreamSynchronize(s->stream));
    mark("blocked operand copy");
    if (!(flags & PQV_LAYOUT_ROW_ORDER) && corpus->owned &&
        ((flags & PQV_RELEASE_ROW_ORDER) || ((flags & PQV_RELEASE_IF_COPIED) && !s->images_only))) {
        (void)hipFree(corpus->d_rows);
        corpus->d_rows = nullptr;
    }
#undef S_TRY
    mark("release / epilogue");
    *out = s;
    return PQV_OK;
}
extern "C" int pqv_searcher_create(const pqv_index *index, pqv_corpus *corpus, uint32_t flags,
                                   pqv_searcher **out) {
    return guard([&] { return pqv_searcher_create_impl(index, corpus, flags, out); });
}

extern "C" void pqv_searcher_free(pqv_searcher *s) {
    if (!s) return;
    (void)hipSetDevice(s->device);
    delete s;
}

namespace {

struct TopkPlan {
    uint32_t np;            // effective nprobe
    uint32_t probe_bpl;     // blocks over the centroid matrix
    uint32_t probe_kpart;   // entries per partial list of the probe pass (np, or 64 unsorted from probe_rows_kernel)
    bool probe_rows;        // batched probe_rows_kernel instead of the per-query stream_kernel
    uint32_t rr_rows_per_block, rr_bpl;
    uint32_t n_part_probe, n_part_rr;
    bool tile;              // batched cluster-major tiles instead of one stream per (query, list)
    uint32_t max_groups;
    bool filter;            // tile path with the MFMA lower-bound screen
    uint32_t seed_rows, filter_bpl;
    bool f16;               // wide path with f16 operands (dim % 128 == 0, queries staged in LDS)
    bool i8;                // wide path with int8 operands (dim % 256 == 0, 8-wave blocks)
    bool quad;              // filter: wide_filter_kernel (quad_width queries per block); else tile_filter_kernel
    uint32_t filter_rows_per_block, max_quads, quad_width;
    uint32_t block_waves;   // wide kernel: waves per block (4, or 8 sharing one staged quad on a whole CU)
    uint32_t slots_per_pair;   // partial lists per (query, probe rank)
    uint32_t wide_width;    // > 0: quads of 97..wide_width pairs go to the wide-quad instance (its own item table)
    uint32_t wide_rows_per_block, wide_bpl;
    bool list_once;         // the wide table's quads hold up to wide_width (<= 1024) pairs of ONE list and run in list_filter_kernel
};

// Exact refinement of the seed threshold (SeedRefine): where survivors are expensive (rows of >= 256 dims; C2, 128 dims:
// 7.18 -> 7.04 M QPS with it).  Any batch size: one uniform C3 query pays 7 us for it (256 -> 263), but one query on the
// Gaussian-mixture set overflows its candidate buffer without it (282 -> 500 us).
// candidate-buffer entries per query: survivors of the screen grow with k (C3: 240 at k 10 with the refined first
// threshold, 810 at k 32, 2100 at k 100 -- and a query that overflows its buffer falls back to the per-wave sorted lists)
static uint32_t cand_cap_for(const pqv_searcher *s, uint32_t k) {
    return std::max<uint32_t>(s->opt.cand_cap ? s->opt.cand_cap : (k > 32 ? 8192u : 2048u), k);
}
// Deferred exact evaluation (TileArgs::cand_lb): pays where the exact evaluations are most of the screen kernel's time -- long
// lists of survivors, i.e. large k (the reference's bench shape, 1 M x 1024 at K = 100: 1226 evaluations per query in the
// filter -> 633 after it, screen kernel 2.14 -> 0.95 ms).  At k = 10 the streaming waves hide their evaluations behind each
// other and the extra launches cost more than they save (C3: 1.08 + 0.73 ms of screen either way), so the deferred form is
// compiled into the k > 64 instances of the kernel only (S > 1) and those always use it (PQV_DEFER=0: never).
// k <= 64: only where a wave's own evaluations are a large part of its short life -- lists of a few tiles per wave of long
// rows (int8 images of >= 512 dims; the two-instance form of batches).  With 1024 lists of 768-dim rows and k = 10, q/s
// deferred against in-filter: 488 rows per list 1.79 against 1.49 M, 976: 1.63 / 1.50 M, 2441: 1.29 / 1.20 M, 4883:
// 0.83 / 0.91 M, 9766 (C3): equal at best -- hence mean list <= 3072 rows.  128-dim rows cost less to evaluate than to
// defer (C2 7.8 -> 6.8 M, its 1000-list variant 5.9 -> 5.5 M).
static bool defer_on(const pqv_searcher *s, uint32_t nq, uint32_t k, const TopkPlan &p) {
    if (s->opt.defer == 0) return false;
    if (k > 64) return true;                    // (the kernels carry the deferred form in their k > 64 instances ...)
    const uint64_t mean_len = s->n / std::max<uint32_t>(1, s->n_clusters);
    // (... and in the DEFP instances of the int8 two-blocks-per-CU form: launch_filter_s)
    // (round 5, 768-dim rows, q/s deferred against in-filter on the final kernels -- uniform: 488-row lists 2.14 / 1.68 M, 976: 2.03 /
    //  1.72 M, 2441: 1.45 / 1.27 M; Gaussian mixture: 488: 2.25 / 2.30 M, 976: 1.87 / 1.78 M, 2441: 1.14 / 1.22 M -- on clustered
    //  query loads the upper-bound counting loosens the running thresholds and the gain is gone from ~1500-row lists on: there the
    //  previous batch's shape (prefer_regular: most popular lists probed by hundreds of queries) switches the form off)
    if (s->opt.defer != 2 && mean_len > 1536 && prefer_regular(s)) return false;
    return nq >= 8 && p.i8 && p.block_waves == 4 && p.quad_width == 96 && s->sdim >= 512 && (s->opt.defer == 2 || mean_len <= 3072);
}
static bool seed_refine_on(const pqv_searcher *s, uint32_t nq, uint32_t k) {
    (void)nq;
    return s->opt.seed_refine && (!s->d_row_of || s->images_only) && (s->sdim % 32) == 0 && k <= 16 && (s->opt.seed_refine > 1 || s->sdim >= 256);
}
TopkPlan plan_topk(const pqv_searcher *s, uint32_t nq, uint32_t nprobe, uint32_t k = 1, int metric = 0) {
    TopkPlan p{};
    const pqv_searcher::Opts &o = s->opt;
    p.np = std::min<uint32_t>(nprobe, s->n_clusters);
    // probe pass: every block scans 256 centroids (64 per wave)
    p.probe_bpl = (s->n_clusters + 255) / 256;
    p.n_part_probe = p.probe_bpl * pqv::waves_per_block();
    // (a handful of queries: the per-query stream over the table is the shorter chain -- 26 against 31 us for one query)
    // (it writes every centroid key of every query: beyond 2 GiB of that scratch -- hundreds of thousands of centroids
    //  times tens of thousands of queries in one device call -- the per-query stream probe takes over)
    p.probe_rows = s->opt.probe_rows && s->kc_pad != 0 && (nq >= 8 || s->opt.probe_rows > 1) &&
                   static_cast<uint64_t>(nq) * s->kc_pad * 12 <= (2ull << 30);
    p.probe_kpart = p.probe_rows ? 64u : p.np;
    // re-rank: enough blocks to fill 256 CUs several times over, few enough partial lists
    const uint64_t max_len = std::max<uint64_t>(1, s->max_list_len);
    const uint64_t max_bpl = (max_len + 255) / 256;
    const uint64_t pairs = std::max<uint64_t>(1, static_cast<uint64_t>(nq) * p.np);
    // Tile path: worth it once several queries share a cluster (each streamed row is then
    // reused by up to TILE_QB queries).  k <= 256; REF4 order only.
    p.tile = metric == PQV_L2SQ_REF4 && k <= 256 && pairs >= 4ull * s->n_clusters;
    // ... except where the wide screened kernels apply (5.1c): they read the f16 copy of a list instead of its f32
    // rows and skip the exact arithmetic, which beats the streaming kernel at EVERY batch size -- one query on C2
    // 123 -> 78 us, on C3 2.0 -> 0.34 ms; 32 queries on C3 15.8 -> 2.2 ms
    const uint64_t mean_len = s->n / std::max<uint32_t>(1, s->n_clusters);
    const bool wide_ok = o.filter_variant == 0 && wide_path_possible(s);
    // (round 3: lists down to 192 rows -- the reference's default n_clusters = ceil(sqrt(n)) gives lists of sqrt(n) rows,
    //  index.rs:161-167 -- with a threshold sample that shrinks with them)
    const bool wide_any_batch = metric == PQV_L2SQ_REF4 && o.tile_filter && k <= 128 && wide_ok && mean_len >= 192;
    if (wide_any_batch) p.tile = true;
    if (o.rerank_mode == 1) p.tile = false;
    if (o.rerank_mode == 2) p.tile = metric == PQV_L2SQ_REF4 && k <= 256;
    if (p.tile) {
        const uint64_t est_groups = std::max<uint64_t>(1, pairs / pqv::TILE_QB);
        // Rows per block: long enough to amortise a block's setup and top-k warm-up, short
        // enough that imbalanced lists do not leave a tail (measured optimum ~1.5 k on C2/C3);
        // shrink only when that would leave the GPU with too few blocks.
        const uint64_t tile_rows = o.tile_rows ? o.tile_rows : 1536;
        uint64_t rpb = std::min<uint64_t>(tile_rows, (max_len + 255) / 256 * 256);
        while (rpb > 256 && est_groups * ((max_len + rpb - 1) / rpb) < 2048) rpb -= 256;
        (void)max_bpl;
        p.rr_rows_per_block = static_cast<uint32_t>(rpb);
        // threshold sample: 256 rows per probed list, 512 for lists of >= 4096 rows (survivors per query halve,
        // the sampling pass doubles: C2 0.246 -> 0.226 ms, C3 7.99 -> 7.58 ms)
        // short lists: a third of the mean list, in 64-row tiles
        p.seed_rows = o.seed_rows ? std::max<uint32_t>(64, o.seed_rows / 64 * 64)
                                  : (mean_len >= 4096 ? 512 : mean_len >= 768 ? 256 : static_cast<uint32_t>(std::max<uint64_t>(64, mean_len / 3 / 64 * 64)));
        // k <= 128 (the running-threshold counters are 8 bits wide, the candidate buffers 2048 entries); the wide
        // kernel pays from lists of three seed windows on -- measured on the reference's own bench shape, 1000-row
        // lists at K = 100: 99 k -> 243 k QPS; the one-group kernel (row-order layout, dim % 64 != 0) keeps k <= 32
        // and wins from 4 pairs per cluster on 10 k-row lists: 0.45 -> 0.25 ms at 64 queries, 0.52 -> 0.34 at 256
        p.filter = o.tile_filter && k <= (wide_ok ? 128u : 32u) && pairs >= (wide_ok ? 0ull : 4ull) * s->n_clusters &&
                   mean_len >= (wide_ok ? 3ull : 16ull) * p.seed_rows;
        if (o.tile_filter == 2) p.filter = max_len > 4ull * p.seed_rows && k <= 256;   // forced
        if (p.filter) {
            p.quad = wide_ok;
            const int op = p.quad ? screen_op(s, k) : 0;
            p.i8 = op == 2;
            p.f16 = op == 1;
            // Quad width = queries that share one pass over a list.  f32 operands: 64 (dim <= 128) or 32 staged in
            // LDS, longer rows read a blocked per-quad copy from global memory.  f16 operands: everything the LDS of
            // a CU holds next to the queues -- 144 KB: 128 queries up to 512 dims, 96 at 768, 64 at 1024 -- in ONE
            // 8-wave block per CU, because how often a list is streamed is what bounds long rows (C3, PMC: the
            // 32-query form moves 41 GB per step through the fabric at 6.8 TB/s for 15 GB of distinct rows).
            p.block_waves = 4;
            if (p.i8) {             // int8 images: a byte per value -- 128 queries up to 1024 dims, 96 at 1536, 64 at 2048
                // Up to 1024 dims a 64-query quad is <= 64 KB (96 queries <= 72 KB up to 768 dims), so TWO 4-wave blocks
                // share a CU: the blocks are at different phases (staging / MFMA screen / exact evaluations) and fill
                // each other's stalls, which one 8-wave block in lockstep cannot.  C3, kernel time in one run: one
                // 8-wave block of 96 queries 2.45 ms (12.2 GB through the fabric), two 4-wave blocks of 64 queries 2.36 ms
                // (15.3 GB at 6.5 TB/s: bandwidth-bound), two of 96 queries 2.21 ms.  Longer rows: one 8-wave block.
                p.block_waves = (o.wide_waves != 8 && 64ull * s->sdim <= 65536) ? 4 : 8;
                // (the 128-query form keeps 128 accumulator registers per lane and spills inside the K loop: 96 by default)
                const uint32_t fit = static_cast<uint32_t>(std::min<uint64_t>(128, 147456ull / s->sdim / 32 * 32));
                p.quad_width = std::min<uint32_t>(96, fit);
                if (o.quad_width && (o.quad_width % 32) == 0 && o.quad_width >= 64 && o.quad_width <= fit) p.quad_width = o.quad_width;
                // (a batch that leaves most quads with a few queries -- a single query above all -- takes the 64-query
                //  form: no spills, 250 against 290 us for one C3 query)
                if (p.block_waves == 4)
                    p.quad_width = (o.quad_width != 64 && 96ull * s->sdim <= 73728 && (o.quad_width == 96 || pairs >= 16ull * s->n_clusters)) ? 96 : 64;
            } else if (p.f16) {
                const uint64_t per_q = static_cast<uint64_t>(s->sdim) * (s->sdim <= 128 ? 6 : 2);     // f16 image (+ f32 original)
                const uint32_t fit8 = static_cast<uint32_t>(std::min<uint64_t>(128, 147456 / per_q / 32 * 32));
                // two 4-wave blocks of 96 queries per CU where 96 queries fit 72 KB (up to 128 dims with the f32 originals,
                // up to 384 without): C2 7.15 -> 8.02 M QPS against one 8-wave block of 128 (0.151 -> 0.133 ms kernel
                // time; 5.1f's overlap of two blocks' phases).  Longer rows: one 8-wave block.
                const bool two96 = 96ull * per_q <= 73728;
                const uint32_t fit4 = two96 ? 96 : s->sdim <= 256 ? 64 : 32;
                int waves = o.wide_waves == 4 || o.wide_waves == 8 ? o.wide_waves : two96 ? 4 : 8;
                if (fit8 < 64) waves = 4;
                // (a call of a few queries has no quad of more: the 4-wave form -- 32-query quads of long rows, 64 KB, two blocks per
                //  CU -- loses nothing to a second pass and its blocks are not mostly fixed cost; one query at K = 100 on 1 M x 1024:
                //  330 -> 277 us per call against the 8-wave block of 64; from 8 queries on the 8-wave form is as good or better)
                if (o.wide_waves != 8 && nq <= 4 && !two96 && fit4 >= 32) waves = 4;
                p.block_waves = static_cast<uint32_t>(waves);
                p.quad_width = waves == 8 ? fit8 : fit4;
                // (a requested width the chosen form has no instantiation for is ignored, never an error at launch: the
                //  whole-tile-prefetch form of rows <= 128 dims exists for 64 and 96 queries only)
                const bool pf_form = waves == 4 && s->sdim <= 128;
                if (o.quad_width && (o.quad_width % 32) == 0 && o.quad_width <= p.quad_width && (waves == 4 || o.quad_width >= 64) &&
                    !(pf_form && o.quad_width == 32))
                    p.quad_width = o.quad_width;
            } else {
                p.quad_width = s->sdim <= 128 ? 64 : 32;
            }
            uint64_t r = rpb;
            if (p.quad) {
                // a wave's fixed cost (staging the quad's queries, the last partial batch of exact evaluations) is
                // about half its time at 1536 rows per 4-wave block, and the lists are cut into equal pieces, so
                // longer blocks pay: measured optimum 2304 on C2 (0.221 -> 0.193 ms) and C3 (6.63 -> 6.35 ms);
                // the 8-wave blocks (one per CU) measured best at 3072 on C3 (2.60 -> 2.51 ms against 4608)
                // the two-block int8 form is flat between 1280 and 5120 on the item grid (C3: 2.19 / 2.20 / 2.16 ms at 1280 /
                // 1792 / 3072; C4 shard: 2.77 / 2.69 / 2.71)
                const uint64_t wide_rows = o.wide_rows >= 256 ? std::min<uint64_t>(o.wide_rows, 1u << 22) / 256 * 256
                                           : (p.block_waves == 8 || p.i8 || p.quad_width == 96) ? 3072ull : 2304ull;
                const uint64_t est_quads = std::max<uint64_t>(1, pairs / p.quad_width);
                const uint64_t min_blocks = o.min_blocks ? o.min_blocks : (p.block_waves == 8 ? 1024 : 2048);
                r = std::min<uint64_t>(wide_rows, (max_len + 255) / 256 * 256);
                while (r > 256 * (p.block_waves / 4) && est_quads * (p.quad_width / 16) * ((max_len + r - 1) / r) < min_blocks) r -= 256;
            }
            p.filter_rows_per_block = static_cast<uint32_t>(r);
            if (p.quad) {           // the screened pass covers the whole list; the seed rows are only sampled
                p.filter_bpl = static_cast<uint32_t>((max_len + r - 1) / r);
                p.rr_bpl = p.filter_bpl;
                p.slots_per_pair = p.block_waves * p.filter_bpl;
                // Lists that more than 96 queries of the batch probe (the long, popular ones: 36 % of C3's distinct probed
                // rows) would be streamed twice; a quad of up to 160 queries on 32-row tiles (same accumulator registers) in
                // one 8-wave block per CU reads them once.  Batches only (pairs >= 16 per list on average), work-item grid.
                // Round 6: list_filter_kernel instead -- the list's rows stationary in registers and ALL its pairs (one quad of up to 1024)
                // streamed past them, so a list is read ONCE however many queries probe it (clustered query loads: hundreds).  Not with
                // the deferred form (its survivors carry the wide_filter_kernel queue's raw scores).
                if (p.i8 && p.block_waves == 4 && p.quad_width == 96 && o.wide_quads && o.item_grid >= 1 && pairs >= 16ull * s->n_clusters) {
                    const bool deferred = defer_on(s, nq, k, p) && cand_cap_for(s, k) <= 8192;
                    if (o.list_once && o.wide_quads != 2 && !deferred && (s->sdim % 256) == 0 && s->sdim <= 768 && k <= 256) {
                        p.list_once = true;
                        p.wide_width = std::min<uint32_t>(1024u, std::max<uint32_t>(192u, (nq + 63u) / 64u * 64u));
                        const uint64_t wr = o.wide_quad_rows >= 256 ? std::min<uint64_t>(o.wide_quad_rows, 1u << 22) / 256 * 256 : 2048ull;
                        p.wide_rows_per_block = static_cast<uint32_t>(std::min<uint64_t>(wr, (max_len + 255) / 256 * 256));
                        p.wide_bpl = static_cast<uint32_t>((max_len + p.wide_rows_per_block - 1) / p.wide_rows_per_block);
                        p.slots_per_pair = std::max(p.slots_per_pair, 4 * p.wide_bpl);
                    } else if (!(o.wide_quads == 1 && prefer_regular(s)) && 160ull * s->sdim <= 122880) {
                        p.wide_width = 160;
                        const uint64_t wr = o.wide_quad_rows >= 512 ? std::min<uint64_t>(o.wide_quad_rows, 1u << 22) / 256 * 256 : 8192ull;
                        p.wide_rows_per_block = static_cast<uint32_t>(std::min<uint64_t>(wr, (max_len + 255) / 256 * 256));
                        p.wide_bpl = static_cast<uint32_t>((max_len + p.wide_rows_per_block - 1) / p.wide_rows_per_block);
                        p.slots_per_pair = std::max(p.slots_per_pair, 8 * p.wide_bpl);
                    }
                }
            } else {                // narrow kernel: exact seed window (slot chunk 0), screened remainder
                p.filter_bpl = static_cast<uint32_t>((max_len - p.seed_rows + r - 1) / r);
                p.rr_bpl = 1 + p.filter_bpl;
                p.slots_per_pair = 4 * (1 + p.filter_bpl);
            }
        } else {
            p.filter_bpl = 0;
            p.rr_bpl = static_cast<uint32_t>((max_len + rpb - 1) / rpb);
            p.slots_per_pair = 4 * p.rr_bpl;
        }
        p.n_part_rr = p.np * p.slots_per_pair;
        p.max_quads = static_cast<uint32_t>(pairs / std::max<uint32_t>(16, p.quad_width) + std::min<uint64_t>(s->n_clusters, pairs));
        p.max_quads = (p.max_quads + 7) / 8 * 8;       // quad_xcd_remap
        p.max_groups = static_cast<uint32_t>(pairs / pqv::TILE_QB + std::min<uint64_t>(s->n_clusters, pairs));
        return p;
    }
    uint64_t bpl = (8192 + pairs - 1) / pairs;
    bpl = std::max<uint64_t>(1, std::min<uint64_t>(bpl, max_bpl));
    uint64_t rpb = (max_len + bpl - 1) / bpl;
    rpb = (rpb + 255) / 256 * 256;
    p.rr_rows_per_block = static_cast<uint32_t>(rpb);
    p.rr_bpl = static_cast<uint32_t>((max_len + rpb - 1) / rpb);
    p.n_part_rr = p.np * p.rr_bpl * pqv::waves_per_block();
    return p;
}

// Enqueue probe -> probe-merge -> re-rank -> final merge for one batch on `stream`.
// `k` is the list length the kernels work with; the first k_out entries are written out.
// With k == k_out + 1 the merge can also flag queries whose output distances tie (d_tie).
int enqueue_topk(const pqv_searcher *s, const float *d_queries, uint32_t nq, uint32_t k,
                 uint32_t k_out, uint32_t nprobe, uint64_t max_candidates, int metric, int sqrt_out,
                 uint32_t *d_row_idx, float *d_dist, uint32_t *d_n_found, uint64_t *d_n_cand,
                 uint32_t *d_tie, hipStream_t stream, Scratch &sc) {
    using namespace pqv;
    const TopkPlan p = plan_topk(s, nq, nprobe, k, metric);
    const uint64_t max_pos = max_candidates ? max_candidates : ~0ull;
    // the probe works on the queries as given; everything that meets the (possibly zero-padded) stored rows gets the
    // batch's queries padded the same way
    const float *d_queries_s = d_queries;
    if (s->sdim != s->dim) {
        HIP_TRY(sc.s_qpad.ensure(static_cast<size_t>(nq) * s->sdim * sizeof(float)));
        HIP_TRY(launch_pad_rows(d_queries, nullptr, nq, s->dim, s->sdim, sc.s_qpad.as<float>(), stream));
        d_queries_s = sc.s_qpad.as<float>();
    }

    HIP_TRY(sc.s_probe_keys.ensure(static_cast<size_t>(nq) * p.n_part_probe * p.probe_kpart * sizeof(uint64_t)));
    HIP_TRY(sc.s_probe_vals.ensure(static_cast<size_t>(nq) * p.n_part_probe * p.probe_kpart * sizeof(uint32_t)));
    HIP_TRY(sc.s_probe.ensure(static_cast<size_t>(nq) * p.np * sizeof(uint32_t)));
    HIP_TRY(sc.s_cand_base.ensure(static_cast<size_t>(nq) * p.np * sizeof(uint64_t)));
    HIP_TRY(sc.s_ncand.ensure(static_cast<size_t>(nq) * sizeof(uint64_t)));
    HIP_TRY(sc.s_part_keys.ensure((static_cast<size_t>(nq) * p.n_part_rr * k + 4) * sizeof(uint64_t)));
    HIP_TRY(sc.s_part_vals.ensure((static_cast<size_t>(nq) * p.n_part_rr * k + 4) * sizeof(uint32_t)));

    const bool timing = s->timing;
    hipEvent_t e0 = nullptr, e1 = nullptr, e2 = nullptr, e3 = nullptr;
    if (timing) {
        HIP_TRY(hipEventCreate(&e0)); HIP_TRY(hipEventCreate(&e1));
        HIP_TRY(hipEventCreate(&e2)); HIP_TRY(hipEventCreate(&e3));
        s->ev.push_back(e0); s->ev.push_back(e1); s->ev.push_back(e2); s->ev.push_back(e3);
        HIP_TRY(hipEventRecord(e0, stream));
    }

    // 1. centroid probe: the re-rank kernel over the centroid matrix, k = nprobe
    StreamArgs pa{};
    pa.mat = s->d_centroids.as<float>(); pa.row_of = nullptr; pa.list_off = nullptr;
    pa.probe = nullptr; pa.cand_base = nullptr;
    pa.single_begin = 0; pa.single_end = s->n_clusters;
    pa.queries = d_queries; pa.nq = nq; pa.nprobe = 1; pa.dim = s->dim; pa.k = p.np;
    pa.rows_per_block = 256; pa.blocks_per_list = p.probe_bpl;
    pa.max_pos = ~0ull; pa.metric = PQV_L2SQ_REF4;   // find_closest_centroids always uses index.rs:461
    pa.part_keys = sc.s_probe_keys.as<uint64_t>(); pa.part_vals = sc.s_probe_vals.as<uint32_t>();
    const uint32_t kc_pairs = s->n_clusters;
    uint32_t *pair_u32 = nullptr;
    if (p.tile) {
        // u32 scratch: hist[R][kc] cursor[R][kc] pair_off[kc+1] group_off[kc+1] n_groups[1] quad_off[kc+1] n_quads[1]
        // (R = HIST_REPLICAS partial copies); the probe kernel zeroes hist, the probe merge fills it, the scan sets cursor
        constexpr uint64_t R = pqv::HIST_REPLICAS;
        HIP_TRY(sc.s_pair_u32.ensure(((2 * R + 5) * kc_pairs + 10) * sizeof(uint32_t)));      // + item_off[kc+1] n_items[1] wide_item_off[kc+1] wide_n_items[1]
        pair_u32 = sc.s_pair_u32.as<uint32_t>();
        pa.zero_u32 = pair_u32; pa.zero_n = static_cast<uint32_t>(R * kc_pairs);
        HIP_TRY(sc.s_gthr.ensure(static_cast<size_t>(nq) * sizeof(unsigned long long)));
        if (p.filter) { HIP_TRY(sc.s_qnorm.ensure(static_cast<size_t>(nq) * sizeof(float))); HIP_TRY(sc.s_qmax.ensure(static_cast<size_t>(nq) * sizeof(float))); }
    }
    // one query on the wide screened path: probe, probe merge, bucketing and quantisation in ONE block (probe_single_kernel)
    const bool fused_probe = nq == 1 && p.np <= 64 && p.tile && p.filter && p.quad && s->opt.single_bucket > 0 &&
                             s->opt.single_bucket != 2 && s->kc_pad != 0 && s->kc_pad <= 4096;
    if (fused_probe) {
        // (launched below, once the merge arguments are complete)
    } else if (p.probe_rows) {     // a batch: a lane per centroid, the chains of up to 8 queries in registers
        pqv::ProbeRowsArgs pr{};
        pr.cent_t = s->d_cent_t.as<float4>(); pr.queries = d_queries;
        pr.nq = nq; pr.kc = s->n_clusters; pr.kc_pad = s->kc_pad; pr.dim = s->dim;
        pr.part_keys = pa.part_keys; pr.part_vals = pa.part_vals;
        pr.zero_u32 = pa.zero_u32; pr.zero_n = pa.zero_n;
        HIP_TRY(pqv::launch_probe_rows(pr, stream));
    } else {
        HIP_TRY(launch_stream(pa, STREAM_TOPK, stream));
    }

    MergeArgs pm{};
    pm.part_keys = pa.part_keys; pm.part_vals = pa.part_vals;
    pm.nq = nq; pm.n_part = p.n_part_probe; pm.k_part = p.probe_kpart; pm.k = p.np;
    pm.list_off = s->d_list_off.as<uint64_t>();
    pm.probe = sc.s_probe.as<uint32_t>(); pm.cand_base = sc.s_cand_base.as<uint64_t>();
    pm.n_cand = d_n_cand ? d_n_cand : sc.s_ncand.as<uint64_t>();
    pm.max_pos = max_pos;
    pm.stats = s->d_stats.as<unsigned long long>();
    if (p.tile) {
        pm.hist = pair_u32; pm.hist_stride = kc_pairs; pm.gthr_init = sc.s_gthr.as<unsigned long long>();
        // every partial list of the re-rank starts EMPTY: preset by the probe merge (one wave per query)
        if (p.filter && p.quad) {       // wide path: one "written" byte per list (see MergeArgs::preset_flags)
            const uint32_t nflag = (p.n_part_rr + 3) / 4 * 4;
            HIP_TRY(sc.s_part_flags.ensure(static_cast<size_t>(nq) * nflag));
            pm.preset_flags = sc.s_part_flags.as<uint8_t>(); pm.preset_flag_n = nflag;
        } else {
            pm.preset_keys = sc.s_part_keys.as<uint64_t>(); pm.preset_vals = sc.s_part_vals.as<uint32_t>();
            pm.preset_n = p.n_part_rr * k;
        }
        if (p.filter) { pm.qnorm_out = sc.s_qnorm.as<float>(); pm.qmax_out = sc.s_qmax.as<float>(); pm.queries = d_queries; pm.dim = s->dim; }
    }
    // one query: the probe merge writes the bucketing itself (MergeArgs::sq_*), no pair sort
    const bool single_bucket = nq == 1 && p.np <= 64 && p.tile && p.filter && p.quad && s->opt.single_bucket;
    // (a work-item table of more than 2^24 entries -- batches of hundreds of thousands of queries -- keeps the 2-D grid)
    const bool items = p.tile && p.filter && p.quad && (s->opt.item_grid > 1 || (s->opt.item_grid == 1 && p.block_waves == 4)) &&
                       static_cast<uint64_t>(p.max_quads) * p.filter_bpl <= (1ull << 24);
    const uint32_t max_items = items ? p.max_quads * p.filter_bpl : 0;
    const bool wide = items && p.wide_width != 0 && !single_bucket;
    const uint32_t wide_max_items = wide ? p.max_quads * p.wide_bpl : 0;
    if (p.tile) {
        HIP_TRY(sc.s_quads.ensure(static_cast<size_t>(p.max_quads) * sizeof(uint4)));
        HIP_TRY(sc.s_pairs.ensure(static_cast<size_t>(nq) * p.np * sizeof(uint32_t)));
        HIP_TRY(sc.s_groups.ensure(static_cast<size_t>(p.max_groups) * sizeof(uint4)));
        if (items) HIP_TRY(sc.s_items.ensure(2 * (static_cast<size_t>(max_items) + wide_max_items) * sizeof(uint32_t)));
    }
    if (single_bucket) {
        uint32_t *v = pair_u32 + 2 * (pqv::HIST_REPLICAS - 1) * static_cast<uint64_t>(kc_pairs);
        pm.sq_quads = sc.s_quads.as<uint4>(); pm.sq_pairs = sc.s_pairs.as<uint32_t>(); pm.sq_n_quads = v + 5ull * kc_pairs + 4;
        if (items) {
            pm.sq_item_quad = sc.s_items.as<uint32_t>(); pm.sq_n_items = v + 6ull * kc_pairs + 6;
            pm.sq_item_rows = p.filter_rows_per_block; pm.sq_max_items = max_items;
        }
    }
    // int8 screen: the query images (one per query, or one per probed pair in the residual form) -- see step 2
    pqv::PairQuantArgs pq_args{};
    bool quant_done = false;
    if (p.tile && p.filter && p.quad && p.i8) {
        if (int rc = ensure_blocked_copy(s, 2, stream)) return rc;     // centres and scales (built at creation; here only after an option change)
        const bool per_pair = s->i8_residual;
        const size_t n_img = per_pair ? static_cast<size_t>(nq) * p.np : nq;
        HIP_TRY(sc.s_qi8.ensure(n_img * s->sdim));
        HIP_TRY(sc.s_qn2i.ensure(n_img * sizeof(int)));
        HIP_TRY(sc.s_qres.ensure(n_img * sizeof(float)));
        HIP_TRY(sc.s_qresu.ensure(n_img * sizeof(float)));
        HIP_TRY(sc.s_pair_lb.ensure(n_img * sizeof(float)));
        pq_args = pqv::PairQuantArgs{d_queries_s, per_pair ? sc.s_probe.as<uint32_t>() : nullptr, s->d_center.as<float>(),
                                     s->d_list_scale.as<float>(), s->d_list_half.as<float>(), s->d_list_radius.as<float>(),
                                     static_cast<uint32_t>(n_img), per_pair ? p.np : 1u, s->sdim, static_cast<int8_t *>(sc.s_qi8.p),
                                     sc.s_qn2i.as<int>(), sc.s_qres.as<float>(), sc.s_qresu.as<float>(), sc.s_pair_lb.as<float>()};
    }
    if (fused_probe) {
        pqv::ProbeRowsArgs pr{};
        pr.cent_t = s->d_cent_t.as<float4>(); pr.queries = d_queries;
        pr.nq = 1; pr.kc = s->n_clusters; pr.kc_pad = s->kc_pad; pr.dim = s->dim;
        HIP_TRY(sc.s_probe_keys.ensure(static_cast<size_t>(s->kc_pad) * sizeof(uint64_t)));
        pr.part_keys = sc.s_probe_keys.as<uint64_t>();
        if (!sc.s_ticket.p) {
            HIP_TRY(sc.s_ticket.ensure(sizeof(uint32_t)));
            HIP_TRY(hipMemsetAsync(sc.s_ticket.p, 0, sizeof(uint32_t), stream));
        }
        // (the image(s) of a one-query call are made inside the probe launch)
        quant_done = pq_args.n_pairs != 0;
        HIP_TRY(pqv::launch_probe_single(pr, pm, sc.s_ticket.as<uint32_t>(), quant_done ? &pq_args : nullptr, stream));
    } else {
        HIP_TRY(launch_merge_probe(pm, stream));
    }

    // 2. candidate re-rank + per-wave top-k
    bool use_cand = false;     // wide screened path: the final merge also reads the candidate buffers
    bool use_defer = false;    // ... whose entries carried bounds (deferred evaluation), resolved by launch_resolve
    if (p.tile) {
        const uint32_t n_pairs = nq * p.np, kc = s->n_clusters;
        uint32_t *u = pair_u32;
        PairSortArgs ps{};
        ps.probe = sc.s_probe.as<uint32_t>(); ps.n_pairs = n_pairs; ps.n_clusters = kc; ps.hist_done = 1;
        uint32_t *v = u + 2 * (pqv::HIST_REPLICAS - 1) * static_cast<uint64_t>(kc);    // past the extra histogram / cursor copies
        ps.hist = u; ps.hist_stride = kc; ps.nprobe = p.np; ps.cursor = u + pqv::HIST_REPLICAS * static_cast<uint64_t>(kc);
        ps.pair_off = v + 2ull * kc; ps.group_off = v + 3ull * kc + 1;
        ps.n_groups = v + 4ull * kc + 2;
        ps.quad_off = v + 4ull * kc + 3; ps.n_quads = v + 5ull * kc + 4; ps.quad_width = wide ? p.wide_width : p.quad_width ? p.quad_width : 64;
        ps.pairs = sc.s_pairs.as<uint32_t>(); ps.groups = sc.s_groups.as<uint4>(); ps.quads = sc.s_quads.as<uint4>();
        // work items of the wide filter kernel (quad x row chunk that exists): its grid then has no holes
        // (the 8-wave blocks keep the 2-D grid with its quad-to-XCD affinity: C2 7.25 against 7.13 M QPS)
        if (items) {
            ps.list_off = s->d_list_off.as<uint64_t>(); ps.item_rows = p.filter_rows_per_block;
            ps.item_off = v + 5ull * kc + 5; ps.n_items = v + 6ull * kc + 6;
            ps.item_quad = sc.s_items.as<uint32_t>(); ps.max_items = max_items;
            if (s->opt.chunk_major && !single_bucket) {
                ps.item_chunk = ps.item_quad + max_items + wide_max_items; ps.wide_item_chunk = ps.item_chunk + max_items;
            }
            if (wide && ps.item_chunk) {         // pair_scan_kernel writes the two counts straight into pinned host memory
                if (!s->h_wide_stats.p) {
                    HIP_TRY(s->h_wide_stats.ensure(4 * sizeof(uint32_t)));
                    std::memset(s->h_wide_stats.p, 0, 4 * sizeof(uint32_t));
                }
                ps.wide_stats = s->h_wide_stats.as<uint32_t>();
            }
            if (p.i8 && p.block_waves == 4 && p.quad_width == 96 && ps.item_chunk && s->opt.wide_quads == 1) {
                // the batch's shape for the NEXT batch's choice between the wide-quad instance and regular quads only (prefer_regular)
                if (!s->h_wide_stats.p) {
                    HIP_TRY(s->h_wide_stats.ensure(4 * sizeof(uint32_t)));
                    std::memset(s->h_wide_stats.p, 0, 4 * sizeof(uint32_t));
                }
                ps.shape_stats = s->h_wide_stats.as<uint32_t>() + 2; ps.shape_wide = 160; ps.shape_narrow = p.quad_width;
            }
            // XCD-aware slots: in the wide table always (lists of several wide quads); in the regular table where it has lists of
            // several quads too -- no wide-quad instance, or option 3 (every table)
            ps.xcd_items = !s->opt.xcd_items ? 0u : s->opt.xcd_items >= 3 ? 3u : wide ? 2u : 1u;
            if (wide) {
                ps.wide_list_major = p.list_once ? 1u : 0u;
                ps.wide_min = p.quad_width + 1; ps.wide_item_rows = p.wide_rows_per_block;
                ps.wide_item_off = v + 6ull * kc + 7; ps.wide_n_items = v + 7ull * kc + 8;
                ps.wide_item_quad = sc.s_items.as<uint32_t>() + max_items; ps.wide_max_items = wide_max_items;
            }
        }
        if (!single_bucket) HIP_TRY(launch_pair_sort(ps, stream));
        TileArgs ta{};
        ta.mat = s->d_mat; ta.row_of = s->d_row_of; ta.list_off = s->d_list_off.as<uint64_t>();
        ta.queries = d_queries_s; ta.cand_base = sc.s_cand_base.as<uint64_t>();
        ta.pairs = ps.pairs; ta.groups = ps.groups; ta.n_groups = ps.n_groups; ta.max_groups = p.max_groups;
        ta.nq = nq; ta.nprobe = p.np; ta.dim = s->sdim; ta.k = k;
        ta.quads = ps.quads; ta.n_quads = ps.n_quads; ta.max_quads = p.max_quads; ta.quad_width = p.quad_width;
        ta.rows_per_block = p.rr_rows_per_block; ta.blocks_per_list = p.rr_bpl; ta.max_pos = max_pos;
        ta.slots_per_pair = p.slots_per_pair; ta.slot_base = 0; ta.n_part = p.n_part_rr;
        ta.gthr = sc.s_gthr.as<unsigned long long>();
        ta.part_keys = sc.s_part_keys.as<uint64_t>(); ta.part_vals = sc.s_part_vals.as<uint32_t>();
        if (p.filter && !(p.quad && p.i8)) { if (int rc = ensure_row_norms(s, stream)) return rc; }    // (the f32 / f16 screens read the rows' norms)
        ta.row_norm2 = s->d_row_norm2.as<float>(); ta.norm_by_pos = s->images_only ? 1 : 0;
        ta.stats = s->d_stats.as<unsigned long long>();
        ta.xcd_swizzle = 0;
        if (p.filter && p.quad) {
            if (int rc = ensure_blocked_copy(s, p.i8 ? 2 : p.f16 ? 1 : 0, stream)) return rc;     // built at creation; here only after an option change
            ta.mat_blk = static_cast<const float4 *>(s->d_mat_blk_op[p.i8 ? 2 : p.f16 ? 1 : 0].p);
            ta.blk_off = s->d_blk_off.as<uint64_t>();
            ta.block_waves = p.block_waves;
            if (p.f16) {
                ta.f16 = 1; ta.scale = s->f16_scale; ta.scale2 = s->f16_scale * s->f16_scale;
                ta.query_maxabs = sc.s_qmax.as<float>();
            }
            if (p.i8) {
                // residual form: one image per (query, probed list) pair -- the query's residual against that list's centre at
                // that list's scale -- + the pair's lower bound from the triangle inequality on the centre (pair_lb);
                // one-centre form: one image per query (every list shares centre and scale)
                const bool per_pair = s->i8_residual;
                if (!quant_done) {
                    HIP_TRY(launch_quantize_pairs_i8(pq_args.queries, pq_args.probe, pq_args.center, pq_args.scale, pq_args.half, pq_args.radius,
                                                     pq_args.n_pairs, pq_args.nprobe, pq_args.dim, pq_args.q_i8, pq_args.q_n2i, pq_args.q_res,
                                                     pq_args.q_resu, pq_args.pair_lb, stream));
                    s->counters.kernel_launches += 1;
                }
                ta.i8 = 1; ta.i8_pair_images = per_pair ? 1 : 0;
                ta.q_i8 = static_cast<const int8_t *>(sc.s_qi8.p); ta.q_n2i = sc.s_qn2i.as<int>(); ta.q_res = sc.s_qres.as<float>();
                ta.q_resu = sc.s_qresu.as<float>(); ta.pair_lb = (per_pair && s->opt.pair_prune) ? sc.s_pair_lb.as<float>() : nullptr;
                ta.list_scale = s->d_list_scale.as<float>();
                ta.row_n2i = s->d_row_n2i.as<int>(); ta.row_res = s->d_row_res.as<float>();
            }
            // quad-to-XCD affinity: on by default for the global-query variant, whose per-quad operand copies
            // must stay L2-resident
            const bool q_global = !p.f16 && !p.i8 && static_cast<uint64_t>(p.quad_width) * s->sdim * sizeof(float) > 32768;
            // (8-wave blocks: the quads of one cluster on one XCD, so a list's second pass finds rows in that L2: C3 2.51 -> 2.44 ms)
            ta.xcd_swizzle = s->opt.quad_xcd >= 0 ? s->opt.quad_xcd : (q_global ? 1 : p.block_waves == 8 ? 2 : 0);
            if (q_global) {
                // rows too long to stage a quad's queries in LDS: blocked copy per quad in global memory
                HIP_TRY(sc.s_qblk.ensure(static_cast<size_t>(p.max_quads) * p.quad_width * s->sdim * sizeof(float)));
                HIP_TRY(launch_pack_queries(d_queries_s, ps.pairs, ps.quads, ps.n_quads, p.max_quads, p.np, s->sdim,
                                            p.quad_width / 16, sc.s_qblk.p, stream));
                ta.q_blk = static_cast<const float4 *>(sc.s_qblk.p);
            }
        }
        if (p.filter) ta.query_norm2 = sc.s_qnorm.as<float>();     // filled by the probe merge
        if (timing) HIP_TRY(hipEventRecord(e1, stream));
        if (p.filter && p.quad) {
            const uint32_t ccap = cand_cap_for(s, k);
            HIP_TRY(sc.s_cand_keys.ensure(static_cast<size_t>(nq) * ccap * sizeof(uint64_t)));
            HIP_TRY(sc.s_cand_vals.ensure(static_cast<size_t>(nq) * ccap * sizeof(uint32_t)));
            HIP_TRY(sc.s_cand_cnt.ensure(static_cast<size_t>(nq) * sizeof(uint32_t)));
            HIP_TRY(sc.s_spilled.ensure(static_cast<size_t>(nq) * sizeof(uint32_t)));
            ta.cand_keys = sc.s_cand_keys.as<uint64_t>(); ta.cand_vals = sc.s_cand_vals.as<uint32_t>();
            ta.cand_cnt = sc.s_cand_cnt.as<uint32_t>(); ta.cand_cap = ccap; ta.spilled = sc.s_spilled.as<uint32_t>();
            if (defer_on(s, nq, k, p) && ccap <= 8192) {
                // deferred exact evaluation: bounds per appended pair, and the queues' raw scores (a strip per wave of every block)
                HIP_TRY(sc.s_cand_lb.ensure(static_cast<size_t>(nq) * ccap * sizeof(float)));
                const size_t blocks_r = items ? max_items : static_cast<size_t>(p.filter_bpl | 1u) * p.max_quads;
                const size_t words_r = blocks_r * p.block_waves * pqv::wide_filter_pend(p.quad_width, p.block_waves, p.f16);
                const size_t words_w = wide ? static_cast<size_t>(wide_max_items) * 8 * pqv::wide_filter_pend(p.wide_width, 8, false) : 0;
                HIP_TRY(sc.s_pendv.ensure((words_r + words_w) * sizeof(uint32_t)));
                ta.cand_lb = sc.s_cand_lb.as<float>();
                ta.pendv = sc.s_pendv.as<uint32_t>(); ta.pendv_wide = ta.pendv + words_r;
                use_defer = true;
            }
            // thresholds: upper bounds of the first seed_rows rows of every probed list
            TileArgs seed = ta;
            if (wide) seed.quad_width = p.wide_width;      // (the seed kernel samples a quad in slices of its own 64 queries)
            seed.row_offset = 0; seed.row_end = p.seed_rows; seed.rows_per_block = 256;
            seed.grid_x = (p.seed_rows + 255) / 256; seed.seed_sw = 4 * seed.grid_x;
            const uint32_t n_vals = p.np * seed.seed_sw * 16;
            HIP_TRY(sc.s_seed_ub.ensure(static_cast<size_t>(nq) * n_vals * sizeof(float)));
            seed.seed_ub = sc.s_seed_ub.as<float>();
            // running thresholds: see TileArgs::thr_hist
            if (s->opt.running_thr && k > 1) {
                HIP_TRY(sc.s_thr_hist.ensure(static_cast<size_t>(nq) * 16 * sizeof(uint32_t)));
                HIP_TRY(sc.s_thr_bins.ensure(static_cast<size_t>(nq) * sizeof(float4)));
                ta.thr_hist = sc.s_thr_hist.as<uint32_t>(); ta.thr_bins = static_cast<const float4 *>(sc.s_thr_bins.p);
            }
            pqv::SeedRefine rf{};
            if (seed_refine_on(s, nq, k)) {
                rf.mat = s->d_mat; rf.row_of = s->d_row_of; rf.queries = d_queries_s; rf.list_off = s->d_list_off.as<uint64_t>();
                rf.probe = sc.s_probe.as<uint32_t>(); rf.cand_base = sc.s_cand_base.as<uint64_t>();
                rf.dim = s->sdim; rf.nprobe = p.np; rf.seed_sw = seed.seed_sw; rf.seed_rows = p.seed_rows; rf.max_pos = max_pos;
            }
            // one query: the seed kernel's last block selects (SeedTail); otherwise a launch of its own
            const bool seed_tail = nq == 1 && k <= 256 && s->opt.single_bucket > 0;
            if (seed_tail) {
                if (!sc.s_ticket2.p) {
                    HIP_TRY(sc.s_ticket2.ensure(sizeof(uint32_t)));
                    HIP_TRY(hipMemsetAsync(sc.s_ticket2.p, 0, sizeof(uint32_t), stream));
                }
                seed.seed_tail.enable = 1; seed.seed_tail.n_vals = n_vals; seed.seed_tail.k = k;
                seed.seed_tail.gthr = ta.gthr; seed.seed_tail.cand_cnt = ta.cand_cnt; seed.seed_tail.spilled = ta.spilled;
                seed.seed_tail.thr_hist = ta.thr_hist; seed.seed_tail.thr_bins = static_cast<float4 *>(sc.s_thr_bins.p);
                seed.seed_tail.ticket = sc.s_ticket2.as<uint32_t>();
                seed.seed_tail.rf = rf;
            }
            HIP_TRY(launch_wide_seed(seed, stream));
            if (!seed_tail)
                HIP_TRY(launch_seed_select(seed.seed_ub, nq, n_vals, k, ta.gthr, ta.cand_cnt, ta.spilled, stream,
                                           ta.thr_hist, static_cast<float4 *>(sc.s_thr_bins.p), &rf));
            ta.row_offset = 0; ta.slot_base = 0; ta.grid_x = p.filter_bpl;
            ta.rows_per_block = p.filter_rows_per_block; ta.filter_variant = 0;
            ta.part_flags = sc.s_part_flags.as<uint8_t>();
            if (items) { ta.item_quad = ps.item_quad; ta.item_chunk = ps.item_chunk; ta.wide_item_chunk = ps.wide_item_chunk; ta.n_items = ps.n_items; ta.max_items = max_items; }
            ta.drain_min = static_cast<uint32_t>(std::max(0, s->opt.drain_min)); // (whole-tile prefetch in the 96-query f16 form: 1000-row lists of 128 dims 5.96 -> 6.14 M q/s, C2's 10 k-row lists 7.68 -> 7.14 M)
            ta.opt_pf96 = (s->opt.pf96 >= 2 || (s->opt.pf96 == 1 && s->n / std::max<uint32_t>(1, s->n_clusters) <= 2048)) ? 1u : 0u;
            if (wide) {
                ta.wide_width = p.wide_width; ta.wide_item_quad = ps.wide_item_quad; ta.wide_n_items = ps.wide_n_items;
                ta.wide_max_items = wide_max_items; ta.wide_rows_per_block = p.wide_rows_per_block;
                ta.wide_nt = wide_rows_nt(s) ? 1u : 0u;
                ta.list_once = p.list_once ? 1u : 0u;
                if (s->opt.fork_wide) {
                    if (!sc.side) {
                        HIP_TRY(hipStreamCreateWithFlags(&sc.side, hipStreamNonBlocking));
                        HIP_TRY(hipEventCreateWithFlags(&sc.ev_fork, hipEventDisableTiming));
                        HIP_TRY(hipEventCreateWithFlags(&sc.ev_join, hipEventDisableTiming));
                    }
                    ta.side_stream = sc.side; ta.ev_fork = sc.ev_fork; ta.ev_join = sc.ev_join; ta.fork_wide_first = s->opt.fork_wide >= 2 ? 1u : 0u;
                }
                s->counters.kernel_launches += 1;
            }
            HIP_TRY(launch_tile_filter(ta, stream));
            use_cand = true;
            s->counters.kernel_launches += 3;
            if (use_defer) {
                // the k-th smallest upper bound per query, then every band entry of the call on the whole chip (also for ONE query:
                // a single block evaluating K = 100 rows of 4 KB took 224 us inside the merge, 490 -> 390 us per call); the merge
                // then sees exact keys only.  (Part of the re-rank group: inside its timing events.)
                pqv::MergeArgs rm{};
                rm.nq = nq; rm.k = k; rm.cand_keys = ta.cand_keys; rm.cand_keys_rw = ta.cand_keys; rm.cand_vals = ta.cand_vals;
                rm.cand_cnt = ta.cand_cnt; rm.cand_cap = ccap; rm.cand_lb = ta.cand_lb;
                rm.mat = s->d_mat; rm.queries = d_queries_s; rm.dim = s->sdim; rm.resolve_stats = s->d_stats.as<unsigned long long>();
                // (one query, K = 100: the block evaluating its 100 defining rows alone is 50 of resolve_select's 81 us; the whole
                //  chip evaluates the 380-row band of the single cut in 13)
                rm.resolve_two_cuts = nq >= 32 ? 1 : 0;
                HIP_TRY(sc.s_work.ensure(static_cast<size_t>(nq) * ccap * sizeof(uint32_t) * 2));
                const bool counter_clean = sc.s_nwork.p != nullptr;       // (cleared by the previous call's final merge on this lane)
                HIP_TRY(sc.s_nwork.ensure(sizeof(uint32_t)));
                HIP_TRY(launch_resolve(rm, sc.s_work.p, sc.s_nwork.as<uint32_t>(), counter_clean, stream));
                s->counters.kernel_launches += 2;
            }
        } else if (p.filter) {
            TileArgs seed = ta;          // exact on rows [0, seed_rows) of every list: slot chunk 0
            seed.row_offset = 0; seed.slot_base = 0; seed.grid_x = 1; seed.row_end = p.seed_rows;
            seed.rows_per_block = std::max<uint32_t>(256, p.seed_rows);     // one 64-row tile per wave at least
            HIP_TRY(launch_tile_rerank(seed, stream));
            ta.row_offset = p.seed_rows; ta.slot_base = 4; ta.grid_x = p.filter_bpl;
            ta.rows_per_block = p.filter_rows_per_block; ta.filter_variant = 1;
            HIP_TRY(launch_seed_threshold(ta.part_keys, nq, p.np, p.slots_per_pair, k, ta.gthr, stream));
            HIP_TRY(launch_tile_filter(ta, stream));
            s->counters.kernel_launches += 2;
        } else {
            ta.row_offset = 0; ta.slot_base = 0; ta.grid_x = p.rr_bpl;
            HIP_TRY(launch_tile_rerank(ta, stream));
        }
        if (timing) HIP_TRY(hipEventRecord(e2, stream));
        s->counters.kernel_launches += 3;
    }
    StreamArgs ra{};
    ra.mat = s->d_mat; ra.row_of = s->d_row_of; ra.list_off = s->d_list_off.as<uint64_t>();
    ra.probe = sc.s_probe.as<uint32_t>(); ra.cand_base = sc.s_cand_base.as<uint64_t>();
    ra.queries = d_queries_s; ra.nq = nq; ra.nprobe = p.np; ra.dim = s->sdim; ra.k = k;
    ra.rows_per_block = p.rr_rows_per_block; ra.blocks_per_list = p.rr_bpl;
    ra.max_pos = max_pos; ra.metric = metric;
    ra.part_keys = sc.s_part_keys.as<uint64_t>(); ra.part_vals = sc.s_part_vals.as<uint32_t>();
    if (!p.tile) {
        if (timing) HIP_TRY(hipEventRecord(e1, stream));
        HIP_TRY(launch_stream(ra, STREAM_TOPK, stream));
        if (timing) HIP_TRY(hipEventRecord(e2, stream));
    }

    // 3. fold the per-wave lists
    MergeArgs fm{};
    fm.part_keys = ra.part_keys; fm.part_vals = ra.part_vals;
    fm.nq = nq; fm.n_part = p.n_part_rr; fm.k_part = k; fm.k = k;
    fm.ids = s->d_final_ids; fm.row_idx = d_row_idx; fm.dist = d_dist; fm.n_found = d_n_found;
    fm.sqrt_out = sqrt_out; fm.k_out = k_out; fm.tie_flag = d_tie;
    if (use_cand) {
        fm.cand_keys = sc.s_cand_keys.as<uint64_t>(); fm.cand_vals = sc.s_cand_vals.as<uint32_t>();
        fm.cand_cnt = sc.s_cand_cnt.as<uint32_t>(); fm.cand_cap = cand_cap_for(s, k);
        fm.spilled = sc.s_spilled.as<uint32_t>();
        fm.part_flags = sc.s_part_flags.as<uint8_t>();       // row stride: (n_part + 3) / 4 * 4 == n_part (a multiple of 4 waves)
    }
    if (use_defer) fm.zero_after = sc.s_nwork.as<uint32_t>();
    HIP_TRY(launch_merge_final(fm, stream));
    if (timing) HIP_TRY(hipEventRecord(e3, stream));
    if (int rc = lane_release(sc, stream)) return rc;
    s->counters.kernel_launches += 4;
    return PQV_OK;
}

int validate_topk(const pqv_searcher *s, uint32_t k, uint32_t nprobe, int metric) {
    if (!s) return fail(PQV_ERR_INVALID, "searcher must not be NULL");
    if (k == 0) return fail(PQV_ERR_INVALID, "k must be > 0");                         // search.rs:67
    if (nprobe == 0) return fail(PQV_ERR_INVALID, "nprobe must be > 0");               // search.rs:72
    if (metric != PQV_L2SQ_REF4 && metric != PQV_L2SQ_SEQ) return fail(PQV_ERR_INVALID, "unknown metric");
    return PQV_OK;
}
// the kernels' sorted lists hold up to 1024 entries (k, and the probe's min(nprobe, n_clusters)); pqv_topk goes around
// that limit (topk_unbounded), the asynchronous device entry points report it
bool beyond_kernel_lists(const pqv_searcher *s, uint32_t k_lists, uint32_t nprobe) {
    return k_lists > 1024 || std::min<uint32_t>(nprobe, s->n_clusters) > 1024;
}

}  // namespace

namespace {

// ---- exact replay of the reference's heap for one query ----------------------------------
// std::collections::BinaryHeap<HeapItem> as src/ivf/search.rs:113-127 / exec.rs:264,474-481
// drive it (max-heap on distance; Ord = partial_cmp().unwrap_or(Equal)), restated from the
// published std source: push = append + sift_up (stops on <=), pop = swap_remove(0) +
// sift_down_to_bottom (right child on ties) + sift_up.  Only used for queries whose output
// distances tie, where survivors and order depend on this exact mechanics.
struct HeapEnt { float d; uint32_t row; };

inline bool ent_le(const HeapEnt &a, const HeapEnt &b) { return !(a.d > b.d); }

inline void heap_sift_up(std::vector<HeapEnt> &h, size_t start, size_t pos) {
    const HeapEnt e = h[pos];
    while (pos > start) {
        const size_t parent = (pos - 1) / 2;
        if (ent_le(e, h[parent])) break;
        h[pos] = h[parent];
        pos = parent;
    }
    h[pos] = e;
}

inline void heap_push(std::vector<HeapEnt> &h, HeapEnt e) {
    h.push_back(e);
    heap_sift_up(h, 0, h.size() - 1);
}

inline void heap_pop(std::vector<HeapEnt> &h) {
    HeapEnt item = h.back();
    h.pop_back();
    if (h.empty()) return;
    std::swap(item, h[0]);
    const size_t end = h.size();
    size_t pos = 0;
    const HeapEnt e = h[0];
    size_t child = 1;
    while (end >= 2 && child <= end - 2) {
        if (ent_le(h[child], h[child + 1])) child += 1;
        h[pos] = h[child];
        pos = child;
        child = 2 * pos + 1;
    }
    if (child == end - 1) { h[pos] = h[child]; pos = child; }
    h[pos] = e;
    heap_sift_up(h, 0, pos);
}

// Recompute one query's candidate distances on the device (STREAM_DIST), then replay them
// through the heap in candidate order.  d_probe / d_cand_base: the query's probe list on the device; `clusters` the
// same list on the host.  Any k (the selection is the heap's), any nprobe (the grid is cut into slices of probed lists).
int replay_with_clusters(const pqv_searcher *s, Scratch &sc, const float *d_query, const uint32_t *d_probe,
                         const uint64_t *d_cand_base, const std::vector<uint32_t> &clusters, uint32_t k,
                         uint64_t max_candidates, int metric, int sqrt_out, uint32_t *row_idx, float *dist,
                         uint32_t *n_found) {
    using namespace pqv;
    const uint32_t np = static_cast<uint32_t>(clusters.size());
    uint64_t total = 0;
    for (uint32_t c : clusters) total += s->h_list_off[c + 1] - s->h_list_off[c];
    const uint64_t use = max_candidates ? std::min<uint64_t>(total, max_candidates) : total;
    std::vector<float> d(std::max<uint64_t>(1, total));
    if (total) {
        HIP_TRY(sc.s_replay.ensure(total * sizeof(float)));
        for (uint32_t j0 = 0; j0 < np; j0 += 32768) {                   // gridDim.y <= 65535
            StreamArgs ra{};
            ra.mat = s->d_mat; ra.row_of = s->d_row_of; ra.list_off = s->d_list_off.as<uint64_t>();
            ra.probe = d_probe + j0; ra.cand_base = d_cand_base + j0;
            ra.queries = d_query; ra.nq = 1; ra.nprobe = std::min<uint32_t>(32768, np - j0); ra.dim = s->sdim; ra.k = 1;     // d_query: sdim wide
            ra.rows_per_block = 1024;
            ra.blocks_per_list = static_cast<uint32_t>((std::max<uint64_t>(1, s->max_list_len) + 1023) / 1024);
            ra.max_pos = ~0ull; ra.metric = metric; ra.out_f32 = sc.s_replay.as<float>();
            HIP_TRY(launch_stream(ra, STREAM_DIST, s->stream));
            s->counters.kernel_launches += 1;
        }
        HIP_TRY(hipMemcpyAsync(d.data(), sc.s_replay.p, total * sizeof(float), hipMemcpyDeviceToHost, s->stream));
        HIP_TRY(hipStreamSynchronize(s->stream));
    }
    std::vector<HeapEnt> heap;
    heap.reserve(static_cast<size_t>(std::min<uint64_t>(k, use)) + 1);
    uint64_t pos = 0;
    const std::vector<uint32_t> *h_rows = s->h_rows->get();             // (the index' host lists: downloaded by the first call that reads them)
    if (!h_rows) return PQV_ERR_HIP;
    for (uint32_t c : clusters) {                                        // candidate_rows order
        const uint64_t b = s->h_list_off[c], e = s->h_list_off[c + 1];
        for (uint64_t i = b; i < e && pos < use; ++i, ++pos) {
            const HeapEnt ent{d[pos], (*h_rows)[i]};
            if (heap.size() < k) heap_push(heap, ent);                   // search.rs:119-120
            else if (ent.d < heap[0].d) { heap_pop(heap); heap_push(heap, ent); }   // :121-125
        }
    }
    if (sqrt_out) for (auto &h : heap) h.d = std::sqrt(h.d);             // :133 (IEEE sqrtf)
    std::stable_sort(heap.begin(), heap.end(), [](const HeapEnt &a, const HeapEnt &b) { return a.d < b.d; });
    for (uint32_t i = 0; i < k; ++i) {
        if (i < heap.size()) { row_idx[i] = heap[i].row; dist[i] = heap[i].d; }
        else { row_idx[i] = 0xFFFFFFFFu; dist[i] = INFINITY; }
    }
    if (n_found) *n_found = static_cast<uint32_t>(heap.size());
    return PQV_OK;
}

// qi indexes the current sub-batch's probe scratch (written by the probe merge).
int replay_query_exact(const pqv_searcher *s, Scratch &sc, const float *d_query, uint32_t qi, uint32_t np, uint32_t k,
                       uint64_t max_candidates, int metric, int sqrt_out, uint32_t *row_idx, float *dist,
                       uint32_t *n_found) {
    std::vector<uint32_t> clusters(np);
    HIP_TRY(hipMemcpyAsync(clusters.data(), sc.s_probe.as<uint32_t>() + static_cast<size_t>(qi) * np,
                           np * sizeof(uint32_t), hipMemcpyDeviceToHost, s->stream));
    HIP_TRY(hipStreamSynchronize(s->stream));
    return replay_with_clusters(s, sc, d_query, sc.s_probe.as<uint32_t>() + static_cast<size_t>(qi) * np,
                                sc.s_cand_base.as<uint64_t>() + static_cast<size_t>(qi) * np, clusters, k, max_candidates,
                                metric, sqrt_out, row_idx, dist, n_found);
}

// topk() beyond the kernels' list capacity (k >= 1024: no runner-up slot left; min(nprobe, n_clusters) > 1024): the
// reference accepts any NonZeroUsize (search.rs:56-81).  Per query: every centroid distance on the GPU (STREAM_DIST over
// the centroid table), find_closest_centroids' stable sort + take(nprobe) on the host (index.rs:143-148), every
// candidate distance on the GPU, the reference's heap on the host.  Correct for any k / nprobe; not a fast path.
// find_closest_centroids (index.rs:130-149) without the kernels' list limit: every centroid distance on the GPU
// (STREAM_DIST over the centroid table), the stable sort on the host.  d_query: device [dim]; order: all clusters, nearest first.
int centroid_order_host(const pqv_searcher *s, Scratch &sc, const float *d_query, std::vector<uint32_t> &order) {
    using namespace pqv;
    const uint32_t kc = s->n_clusters;
    HIP_TRY(sc.s_replay.ensure(std::max<size_t>(1, kc) * sizeof(float)));
    std::vector<float> cd(kc);
    StreamArgs pa{};
    pa.mat = s->d_centroids.as<float>(); pa.row_of = nullptr; pa.list_off = nullptr; pa.probe = nullptr; pa.cand_base = nullptr;
    pa.single_begin = 0; pa.single_end = kc;
    pa.queries = d_query; pa.nq = 1; pa.nprobe = 1; pa.dim = s->dim; pa.k = 1;
    pa.rows_per_block = 256; pa.blocks_per_list = (kc + 255) / 256;
    pa.max_pos = ~0ull; pa.metric = PQV_L2SQ_REF4;        // find_closest_centroids always uses index.rs:461
    pa.out_f32 = sc.s_replay.as<float>();
    HIP_TRY(launch_stream(pa, STREAM_DIST, s->stream));
    HIP_TRY(hipMemcpyAsync(cd.data(), sc.s_replay.p, static_cast<size_t>(kc) * sizeof(float), hipMemcpyDeviceToHost, s->stream));
    HIP_TRY(hipStreamSynchronize(s->stream));
    order.resize(kc);
    for (uint32_t c = 0; c < kc; ++c) order[c] = c;
    std::stable_sort(order.begin(), order.end(), [&](uint32_t a, uint32_t b) { return cd[a] < cd[b]; });   // index.rs:143-147
    s->counters.kernel_launches += 1;
    return PQV_OK;
}

int topk_unbounded(const pqv_searcher *s, Scratch &sc, const float *queries, uint32_t nq, uint32_t k, uint32_t nprobe,
                   uint64_t max_candidates, int metric, int sqrt_out, uint32_t *row_idx, float *dist, uint32_t *n_found,
                   uint64_t *n_candidates) {
    using namespace pqv;
    const uint32_t kc = s->n_clusters, np = std::min<uint32_t>(nprobe, kc);
    HIP_TRY(sc.s_queries.ensure(static_cast<size_t>(s->dim) * sizeof(float)));
    HIP_TRY(sc.s_probe.ensure(std::max<size_t>(1, np) * sizeof(uint32_t)));
    HIP_TRY(sc.s_cand_base.ensure(std::max<size_t>(1, np) * sizeof(uint64_t)));
    std::vector<uint32_t> order, clusters(np);
    std::vector<uint64_t> base(np);
    for (uint32_t q = 0; q < nq; ++q) {
        HIP_TRY(hipMemcpyAsync(sc.s_queries.p, queries + static_cast<uint64_t>(q) * s->dim, static_cast<size_t>(s->dim) * sizeof(float),
                               hipMemcpyHostToDevice, s->stream));
        if (int rc = centroid_order_host(s, sc, sc.s_queries.as<float>(), order)) return rc;
        const float *d_q_s = sc.s_queries.as<float>();
        if (s->sdim != s->dim) {
            HIP_TRY(sc.s_qpad.ensure(static_cast<size_t>(s->sdim) * sizeof(float)));
            HIP_TRY(launch_pad_rows(sc.s_queries.as<float>(), nullptr, 1, s->dim, s->sdim, sc.s_qpad.as<float>(), s->stream));
            d_q_s = sc.s_qpad.as<float>();
        }
        uint64_t total = 0;
        for (uint32_t j = 0; j < np; ++j) {
            clusters[j] = order[j]; base[j] = total;
            total += s->h_list_off[order[j] + 1] - s->h_list_off[order[j]];
        }
        HIP_TRY(hipMemcpyAsync(sc.s_probe.p, clusters.data(), static_cast<size_t>(np) * sizeof(uint32_t), hipMemcpyHostToDevice, s->stream));
        HIP_TRY(hipMemcpyAsync(sc.s_cand_base.p, base.data(), static_cast<size_t>(np) * sizeof(uint64_t), hipMemcpyHostToDevice, s->stream));
        uint32_t nf = 0;
        if (int rc = replay_with_clusters(s, sc, d_q_s, sc.s_probe.as<uint32_t>(), sc.s_cand_base.as<uint64_t>(),
                                          clusters, k, max_candidates, metric, sqrt_out, row_idx + static_cast<uint64_t>(q) * k,
                                          dist + static_cast<uint64_t>(q) * k, &nf))
            return rc;
        if (n_found) n_found[q] = nf;
        if (n_candidates) n_candidates[q] = total;
        s->counters.candidate_rows += total;                             // index_exec.rs:289-299
        s->counters.embeddings_fetched += max_candidates ? std::min<uint64_t>(total, max_candidates) : total;   // exec.rs:411-427
        s->counters.queries += 1;
        s->counters.exact_replays += 1;
    }
    return PQV_OK;
}

}  // namespace

static int pqv_topk_device_impl(const pqv_searcher *s, const void *d_queries, uint32_t nq, uint32_t k,
                               uint32_t nprobe, uint64_t max_candidates, int metric, int sqrt_out,
                               void *d_row_idx, void *d_dist, void *d_n_found, void *d_n_candidates,
                               void *d_tie_flags, void *hip_stream) {
    if (int rc = validate_topk(s, k, nprobe, metric)) return rc;
    if (d_tie_flags && k > 1023) return fail(PQV_ERR_UNSUPPORTED, "tie flags need a runner-up entry: k <= 1023");
    if (beyond_kernel_lists(s, k, nprobe))
        return fail(PQV_ERR_UNSUPPORTED, "the device entry points take k <= 1024 and min(nprobe, n_clusters) <= 1024 (pqv_topk has no such limit)");
    if (nq == 0) return PQV_OK;
    if (!d_queries || !d_row_idx || !d_dist) return fail(PQV_ERR_INVALID, "device pointers must not be NULL");
    if (int rc = use_device(s->device)) return rc;
    std::lock_guard<std::mutex> lock(s->mu);
    hipStream_t stream = hip_stream ? static_cast<hipStream_t>(hip_stream) : s->stream;
    Scratch *lane = nullptr;
    if (int rc = lane_acquire(s, stream, &lane)) return rc;
    LaneGuard lane_guard{*lane, stream};
    // with flags the kernels carry one extra merged entry (the runner-up), exactly as pqv_topk does
    const int rc = enqueue_topk(s, static_cast<const float *>(d_queries), nq, d_tie_flags ? k + 1 : k, k, nprobe, max_candidates,
                                metric, sqrt_out, static_cast<uint32_t *>(d_row_idx),
                                static_cast<float *>(d_dist), static_cast<uint32_t *>(d_n_found),
                                static_cast<uint64_t *>(d_n_candidates), static_cast<uint32_t *>(d_tie_flags), stream, *lane);
    if (rc == PQV_OK) s->counters.queries += nq;
    return rc;
}
extern "C" int pqv_topk_device(const pqv_searcher *s, const void *d_queries, uint32_t nq, uint32_t k,
                               uint32_t nprobe, uint64_t max_candidates, int metric, int sqrt_out,
                               void *d_row_idx, void *d_dist, void *d_n_found, void *d_n_candidates,
                               void *hip_stream) {
    return guard([&] { return pqv_topk_device_impl(s, d_queries, nq, k, nprobe, max_candidates, metric, sqrt_out, d_row_idx, d_dist, d_n_found, d_n_candidates, nullptr, hip_stream); });
}
extern "C" int pqv_topk_device_flags(const pqv_searcher *s, const void *d_queries, uint32_t nq, uint32_t k,
                                     uint32_t nprobe, uint64_t max_candidates, int metric, int sqrt_out,
                                     void *d_row_idx, void *d_dist, void *d_n_found, void *d_n_candidates,
                                     void *d_tie_flags, void *hip_stream) {
    if (!d_tie_flags) return fail(PQV_ERR_INVALID, "d_tie_flags must not be NULL");
    return guard([&] { return pqv_topk_device_impl(s, d_queries, nq, k, nprobe, max_candidates, metric, sqrt_out, d_row_idx, d_dist, d_n_found, d_n_candidates, d_tie_flags, hip_stream); });
}

static int pqv_topk_impl(const pqv_searcher *s, const float *queries, uint32_t nq, uint32_t query_len,
                        uint32_t k, uint32_t nprobe, uint64_t max_candidates, int metric, int sqrt_out,
                        uint32_t *row_idx, float *dist, uint32_t *n_found, uint64_t *n_candidates) {
    if (int rc = validate_topk(s, k, nprobe, metric)) return rc;
    if (query_len != s->dim)                                                           // search.rs:91-98
        return fail(PQV_ERR_INVALID, "Query dimension mismatch: expected " + std::to_string(s->dim) +
                                         ", got " + std::to_string(query_len));
    if (nq == 0) return PQV_OK;
    if (!queries || !row_idx || !dist) return fail(PQV_ERR_INVALID, "queries/row_idx/dist must not be NULL");
    if (int rc = use_device(s->device)) return rc;
    std::lock_guard<std::mutex> lock(s->mu);
    // One extra merged entry (the runner-up) lets the merge kernel see ties at the k-th
    // distance; queries it flags are replayed through the exact heap (replay_query_exact).
    // (k == UINT32_MAX -- "keep everything": the reference takes any NonZeroUsize -- must not wrap to a plan with k = 0)
    const uint32_t k_int = k >= 1024u ? k : k + 1;
    Scratch *lane = nullptr;
    if (int rc = lane_acquire(s, s->stream, &lane)) return rc;
    Scratch &sc = *lane;
    LaneGuard lane_guard{sc, s->stream};
    if (k >= 1024u || beyond_kernel_lists(s, k_int, nprobe)) {
        const int rc = topk_unbounded(s, sc, queries, nq, k, nprobe, max_candidates, metric, sqrt_out, row_idx, dist, n_found, n_candidates);
        const int rc2 = lane_release(sc, s->stream);
        return rc ? rc : rc2;
    }
    // bound the scratch: sub-batch so the per-wave partial lists stay under ~1 GiB
    // (per query: partial lists, probe partial lists, candidate buffer, the int8 images of its probed pairs)
    const TopkPlan p1 = plan_topk(s, std::min<uint32_t>(nq, 1024), nprobe, k_int, metric);
    const uint64_t per_query = static_cast<uint64_t>(p1.n_part_rr) * k_int * 12 + static_cast<uint64_t>(p1.n_part_probe) * p1.probe_kpart * 12 +
                               static_cast<uint64_t>(cand_cap_for(s, k_int)) * 12 + static_cast<uint64_t>(p1.np) * (s->sdim + 32) +
                               static_cast<uint64_t>(s->sdim) * 4 + 1;
    uint32_t batch = static_cast<uint32_t>(std::max<uint64_t>(1, std::min<uint64_t>(nq, (1ull << 30) / per_query)));
    HIP_TRY(sc.s_queries.ensure(static_cast<size_t>(batch) * s->dim * sizeof(float)));
    HIP_TRY(sc.s_rows.ensure(static_cast<size_t>(batch) * k * sizeof(uint32_t)));
    HIP_TRY(sc.s_dist.ensure(static_cast<size_t>(batch) * k * sizeof(float)));
    HIP_TRY(sc.s_nfound.ensure(static_cast<size_t>(batch) * sizeof(uint32_t)));
    HIP_TRY(sc.s_ncand.ensure(static_cast<size_t>(batch) * sizeof(uint64_t)));
    HIP_TRY(sc.s_tie.ensure(static_cast<size_t>(batch) * sizeof(uint32_t)));
    std::vector<uint64_t> h_ncand(batch);
    std::vector<uint32_t> h_tie(batch), h_nf(batch);
    const uint32_t np = std::min<uint32_t>(nprobe, s->n_clusters);
    // A call of a few queries (TopkBuilder::search is ONE) is six small pageable copies otherwise -- the query in, rows, distances,
    // counts and tie flags out, each staged and waited for by the runtime: 60-80 us around 180 us of kernels.  Small calls go
    // through ONE pinned buffer instead: the results are laid out as one device block {candidates u64 | rows | dist | found |
    // tie} and come back in one copy.
    const size_t q_bytes = static_cast<size_t>(batch) * s->dim * sizeof(float);
    const size_t out_bytes = static_cast<size_t>(batch) * (16 + 8 * static_cast<size_t>(k));
    // (the result block starts with the u64 candidate counts: its offset behind the queries is rounded up to 64 bytes -- batch * dim
    //  may be odd, and the kernels store 64-bit values there)
    const size_t out_off = (q_bytes + 63) & ~static_cast<size_t>(63);
    const bool small_io = out_off + out_bytes <= (256u << 10);
    uint32_t *o_rows = sc.s_rows.as<uint32_t>(), *o_nf = sc.s_nfound.as<uint32_t>(), *o_tie = sc.s_tie.as<uint32_t>();
    float *o_dist = sc.s_dist.as<float>();
    if (small_io) {
        HIP_TRY(sc.s_out.ensure(out_bytes));
        HIP_TRY(sc.h_io.ensure(out_off + out_bytes));
    }
    for (uint32_t q0 = 0; q0 < nq; q0 += batch) {
        const uint32_t b = std::min<uint32_t>(batch, nq - q0);
        // (round 5: the result block of a small call is WRITTEN by the kernels straight into the pinned buffer -- host memory the
        //  device can address -- so there is no device-to-host copy behind them, only the synchronise: PQV_SMALL_IO_DIRECT=0 keeps the copy)
        static const bool direct_out = [] { const char *e = std::getenv("PQV_SMALL_IO_DIRECT"); return !(e && *e == '0'); }();
        if (small_io) {
            char *ob = direct_out ? static_cast<char *>(sc.h_io.p) + out_off : static_cast<char *>(sc.s_out.p);           // (laid out for THIS sub-batch's b)
            o_rows = reinterpret_cast<uint32_t *>(ob + 8ull * b);
            o_dist = reinterpret_cast<float *>(ob + 8ull * b + 4ull * b * k);
            o_nf = reinterpret_cast<uint32_t *>(ob + 8ull * b + 8ull * b * k);
            o_tie = o_nf + b;
            std::memcpy(sc.h_io.p, queries + static_cast<uint64_t>(q0) * s->dim, static_cast<size_t>(b) * s->dim * sizeof(float));
            HIP_TRY(hipMemcpyAsync(sc.s_queries.p, sc.h_io.p, static_cast<size_t>(b) * s->dim * sizeof(float), hipMemcpyHostToDevice, s->stream));
        } else {
            HIP_TRY(hipMemcpyAsync(sc.s_queries.p, queries + static_cast<uint64_t>(q0) * s->dim,
                                   static_cast<size_t>(b) * s->dim * sizeof(float), hipMemcpyHostToDevice, s->stream));
        }
        if (int rc = enqueue_topk(s, sc.s_queries.as<float>(), b, k_int, k, nprobe, max_candidates, metric,
                                  sqrt_out, o_rows, o_dist, o_nf,
                                  small_io ? reinterpret_cast<uint64_t *>(direct_out ? static_cast<char *>(sc.h_io.p) + out_off : static_cast<char *>(sc.s_out.p)) : nullptr,
                                  o_tie, s->stream, sc))
            return rc;
        if (small_io) {
            char *hb = static_cast<char *>(sc.h_io.p) + out_off;
            const size_t ob_bytes = static_cast<size_t>(b) * (16 + 8 * static_cast<size_t>(k));
            if (!direct_out) HIP_TRY(hipMemcpyAsync(hb, sc.s_out.p, ob_bytes, hipMemcpyDeviceToHost, s->stream));
            HIP_TRY(hipStreamSynchronize(s->stream));
            std::memcpy(h_ncand.data(), hb, static_cast<size_t>(b) * sizeof(uint64_t));
            std::memcpy(row_idx + static_cast<uint64_t>(q0) * k, hb + 8ull * b, static_cast<size_t>(b) * k * sizeof(uint32_t));
            std::memcpy(dist + static_cast<uint64_t>(q0) * k, hb + 8ull * b + 4ull * b * k, static_cast<size_t>(b) * k * sizeof(float));
            std::memcpy(h_nf.data(), hb + 8ull * b + 8ull * b * k, static_cast<size_t>(b) * sizeof(uint32_t));
            std::memcpy(h_tie.data(), hb + 8ull * b + 8ull * b * k + 4ull * b, static_cast<size_t>(b) * sizeof(uint32_t));
        } else {
        HIP_TRY(hipMemcpyAsync(row_idx + static_cast<uint64_t>(q0) * k, sc.s_rows.p,
                               static_cast<size_t>(b) * k * sizeof(uint32_t), hipMemcpyDeviceToHost, s->stream));
        HIP_TRY(hipMemcpyAsync(dist + static_cast<uint64_t>(q0) * k, sc.s_dist.p,
                               static_cast<size_t>(b) * k * sizeof(float), hipMemcpyDeviceToHost, s->stream));
        HIP_TRY(hipMemcpyAsync(h_nf.data(), sc.s_nfound.p, static_cast<size_t>(b) * sizeof(uint32_t),
                               hipMemcpyDeviceToHost, s->stream));
        HIP_TRY(hipMemcpyAsync(h_tie.data(), sc.s_tie.p, static_cast<size_t>(b) * sizeof(uint32_t),
                               hipMemcpyDeviceToHost, s->stream));
        HIP_TRY(hipMemcpyAsync(h_ncand.data(), sc.s_ncand.p, static_cast<size_t>(b) * sizeof(uint64_t),
                               hipMemcpyDeviceToHost, s->stream));
        HIP_TRY(hipStreamSynchronize(s->stream));
        }
        for (uint32_t i = 0; i < b; ++i) {      // (candidate_rows / embeddings_fetched are counted on the device)
            if (n_candidates) n_candidates[q0 + i] = h_ncand[i];
            if (h_tie[i]) {
                // tied output distances: survivors / order follow Rust's heap mechanics exactly
                if (int rc = replay_query_exact(s, sc, s->sdim != s->dim ? sc.s_qpad.as<float>() + static_cast<size_t>(i) * s->sdim
                                                                          : sc.s_queries.as<float>() + static_cast<size_t>(i) * s->dim, i, np,
                                                k, max_candidates, metric, sqrt_out,
                                                row_idx + static_cast<uint64_t>(q0 + i) * k,
                                                dist + static_cast<uint64_t>(q0 + i) * k, &h_nf[i]))
                    return rc;
                s->counters.exact_replays++;
            }
            if (n_found) n_found[q0 + i] = h_nf[i];
        }
        s->counters.queries += b;
    }
    return PQV_OK;
}
extern "C" int pqv_topk(const pqv_searcher *s, const float *queries, uint32_t nq, uint32_t query_len,
                        uint32_t k, uint32_t nprobe, uint64_t max_candidates, int metric, int sqrt_out,
                        uint32_t *row_idx, float *dist, uint32_t *n_found, uint64_t *n_candidates) {
    return guard([&] { return pqv_topk_impl(s, queries, nq, query_len, k, nprobe, max_candidates, metric, sqrt_out, row_idx, dist, n_found, n_candidates); });
}

static int pqv_searcher_set_option_impl(pqv_searcher *s, const char *name, int64_t value) {
    if (!s || !name) return fail(PQV_ERR_INVALID, "searcher/name must not be NULL");
    std::lock_guard<std::mutex> lock(s->mu);
    pqv_searcher::Opts &o = s->opt;
    const std::string n(name);
    if (n == "rerank_mode") o.rerank_mode = static_cast<int>(value);
    else if (n == "tile_filter") o.tile_filter = static_cast<int>(std::min<int64_t>(2, std::max<int64_t>(0, value)));
    else if (n == "filter_variant") o.filter_variant = static_cast<int>(value);
    else if (n == "cand_cap") o.cand_cap = static_cast<uint32_t>(std::max<int64_t>(0, value));
    else if (n == "screen_f16") o.screen_f16 = value != 0;
    else if (n == "screen_i8") o.screen_i8 = static_cast<int>(value);
    else if (n == "seed_rows") o.seed_rows = static_cast<uint32_t>(std::max<int64_t>(0, value));
    else if (n == "wide_rows") o.wide_rows = static_cast<uint32_t>(std::max<int64_t>(0, value));
    else if (n == "tile_rows") o.tile_rows = static_cast<uint32_t>(std::max<int64_t>(0, value));
    else if (n == "running_thr") o.running_thr = value != 0;
    else if (n == "defer") o.defer = static_cast<int>(value);
    else if (n == "quad_xcd") o.quad_xcd = static_cast<int>(value);
    else if (n == "wide_waves") o.wide_waves = static_cast<int>(value);
    else if (n == "single_bucket") o.single_bucket = static_cast<int>(value);      // 2 = bucketing in the merge, separate probe launch
    else if (n == "seed_refine") o.seed_refine = static_cast<int>(value);       // 2 = any dim / batch size
    else if (n == "item_grid") o.item_grid = static_cast<int>(value);          // 2 = also for the 8-wave blocks
    else if (n == "chunk_major") o.chunk_major = value != 0;
    else if (n == "probe_rows") o.probe_rows = static_cast<int>(value);       // 2 = for any batch size
    else if (n == "quad_width") o.quad_width = static_cast<uint32_t>(std::max<int64_t>(0, value));
    else if (n == "min_blocks") o.min_blocks = static_cast<uint32_t>(std::max<int64_t>(0, value));
    else if (n == "pair_prune") o.pair_prune = value != 0;
    else if (n == "wide_quads") o.wide_quads = value <= 0 ? 0 : value >= 2 ? 2 : 1;       // 1: by the previous batch's shape; 2: always
    else if (n == "fork_wide") o.fork_wide = value <= 0 ? 0 : value >= 2 ? 2 : 1;
    else if (n == "pf96") o.pf96 = value <= 0 ? 0 : value >= 2 ? 2 : 1;
    else if (n == "drain_min") o.drain_min = static_cast<int>(std::max<int64_t>(0, std::min<int64_t>(64, value)));
    else if (n == "xcd_items") o.xcd_items = static_cast<int>(std::max<int64_t>(0, std::min<int64_t>(3, value)));
    else if (n == "wide_quad_rows") o.wide_quad_rows = static_cast<uint32_t>(std::max<int64_t>(0, value));
    else if (n == "list_once") o.list_once = static_cast<int>(value);
    else if (n == "i8_form") {        // takes effect when the int8 copy is (re)built
        o.i8_form = static_cast<int>(std::min<int64_t>(2, std::max<int64_t>(0, value)));
        (void)hipSetDevice(s->device);
        (void)hipDeviceSynchronize();          // nothing in flight may still read the copy
        s->d_mat_blk_op[2].release();
    }
    else return fail(PQV_ERR_INVALID, "unknown searcher option: " + n);
    return PQV_OK;
}
extern "C" int pqv_searcher_set_option(pqv_searcher *s, const char *name, int64_t value) {
    return guard([&] { return pqv_searcher_set_option_impl(s, name, value); });
}

static int pqv_searcher_describe_impl(const pqv_searcher *s, uint32_t nq, uint32_t k, uint32_t nprobe, int metric,
                                      char *buf, size_t len) {
    if (!s || !buf || !len) return fail(PQV_ERR_INVALID, "searcher/buf must not be NULL");
    if (int rc = validate_topk(s, k, nprobe, metric)) return rc;
    std::lock_guard<std::mutex> lock(s->mu);
    if (beyond_kernel_lists(s, k, nprobe)) {
        std::snprintf(buf, len, "pqv_topk only: stream_kernel (STREAM_DIST) for every centroid and candidate distance, selection by the "
                                "reference's heap on the host (k > 1024 or min(nprobe, n_clusters) > 1024)");
        return PQV_OK;
    }
    const TopkPlan p = plan_topk(s, std::max<uint32_t>(1, nq), nprobe, k, metric);
    char t[768];
    if (p.tile && p.filter && p.quad)
        std::snprintf(t, sizeof t, "wide_seed_kernel + seed_select_kernel + wide_filter_kernel: %s screen operands, quads of %u queries "
                      "staged %s, %u waves per block, %u rows per block, threshold sample %u rows per list%s%s",
                      p.i8 ? "int8" : p.f16 ? "f16" : "f32", p.quad_width,
                      (p.i8 || p.f16 || static_cast<uint64_t>(p.quad_width) * s->sdim * 4 <= 32768) ? "in LDS" : "as a blocked copy in global memory",
                      p.block_waves, p.filter_rows_per_block, p.seed_rows,
                      seed_refine_on(s, std::max<uint32_t>(1, nq), k) ? " + exact refinement" : "",
                      !p.i8 ? "" : !s->d_mat_blk_op[2].p ? "" : s->i8_residual ? "; int8 images of the per-list residual (one per probed pair)" : "; int8 images about one centre (one per query)");
    else if (p.tile && p.filter)
        std::snprintf(t, sizeof t, "tile_rerank_kernel (exact seed window of %u rows) + tile_filter_kernel: 16-query groups, f32 screen operands, "
                      "%u rows per block", p.seed_rows, p.filter_rows_per_block);
    else if (p.tile)
        std::snprintf(t, sizeof t, "tile_rerank_kernel: exact arithmetic, 16-query groups, %u rows per block", p.rr_rows_per_block);
    else
        std::snprintf(t, sizeof t, "stream_kernel: one candidate stream per (query, probed list), %u rows per block", p.rr_rows_per_block);
    if (p.tile && p.filter && p.quad && p.wide_width) {
        const size_t l = std::strlen(t);
        if (p.list_once)
            std::snprintf(t + l, sizeof t - l, "; lists probed by more than %u queries: list_filter_kernel -- 32 rows per wave stationary in registers, all the list's pairs "
                          "(quads of up to %u) streamed past them in chunks of 32: every such list read once, %u rows per block", p.quad_width, p.wide_width, p.wide_rows_per_block);
        else
        std::snprintf(t + l, sizeof t - l, "; lists probed by %u..%u queries: one quad, 8 waves per block on 32-row tiles, %u rows per block",
                      p.quad_width + 1, p.wide_width, p.wide_rows_per_block);
    }
    if (p.tile && p.filter && p.quad && defer_on(s, std::max<uint32_t>(1, nq), k, p) && cand_cap_for(s, k) <= 8192) {
        const size_t l = std::strlen(t);
        std::snprintf(t + l, sizeof t - l, "; exact evaluations deferred: survivors appended with their bounds, resolve_select_kernel + "
                                           "resolve_exact_kernel evaluate what the k-th smallest upper bound leaves");
    }
    if (s->sdim != s->dim) {
        const size_t l = std::strlen(t);
        std::snprintf(t + l, sizeof t - l, "; rows stored zero-padded from %u to %u dims", s->dim, s->sdim);
    }
    // exact instantiations (as rocprofv3 prints them), so that a profile line can be matched to this dispatch
    char kn[384] = "";
    if (p.tile && p.filter && p.quad) {
        const int S = k <= 64 ? 1 : 4;
        const bool qlds = p.i8 || p.f16 || static_cast<uint64_t>(p.quad_width) * s->sdim * 4 <= 32768;
        const bool pf = p.f16 && s->sdim <= 128 && p.block_waves == 4 && p.quad_width != 96;
        const int op = p.i8 ? 2 : p.f16 ? 1 : 0;
        const int seed_ng = p.i8 ? (64ull * s->sdim <= 49152 ? 4 : 2) : p.f16 ? ((64ull * s->sdim * 2 <= 32768 && p.quad_width % 64 == 0) ? 4 : 2) : static_cast<int>(p.quad_width / 16);
        const bool defp = S == 1 && defer_on(s, std::max<uint32_t>(1, nq), k, p) && cand_cap_for(s, k) <= 8192;
        std::snprintf(kn, sizeof kn, " | kernels: wide_filter_kernel<%u, %u, %d, %s, %d, %s, %s, 4, %s>; wide_seed_kernel<%d, %s, %d, %d>; seed_select_kernel<%d>@%u",
                      p.quad_width / 16, p.block_waves, S, qlds ? "true" : "false", op, pf ? "true" : "false",
                      ((p.i8 && p.block_waves == 4 && p.quad_width == 64 && std::max<uint32_t>(1, nq) <= 64u) || p.wide_width) ? "true" : "false",
                      defp ? "true" : "false", seed_ng, qlds ? "true" : "false", op, (std::max<uint32_t>(1, nq) == 1 && k <= 256 && s->opt.single_bucket > 0) ? 12 : 1, S,
                      std::max<uint32_t>(1, nq) * (seed_refine_on(s, std::max<uint32_t>(1, nq), k) ? 256u : 64u));
        if (p.wide_width) {
            const size_t l = std::strlen(kn);
            if (p.list_once) std::snprintf(kn + l, sizeof kn - l, "; list_filter_kernel<%u, %d>", s->sdim / 64, S);
            else
            std::snprintf(kn + l, sizeof kn - l, "; wide_filter_kernel<%u, 8, %d, true, 2, false, %s, 2, %s>", p.wide_width / 16, S, wide_rows_nt(s) ? "true" : "false",
                          defp ? "true" : "false");
        }
    }
    std::snprintf(buf, len, "%s; centroid probe: %s%s", t, p.probe_rows ? "probe_rows_kernel (a lane per centroid)" : "stream_kernel", kn);
    return PQV_OK;
}
extern "C" int pqv_searcher_describe(const pqv_searcher *s, uint32_t nq, uint32_t k, uint32_t nprobe, int metric,
                                     char *buf, size_t len) {
    return guard([&] { return pqv_searcher_describe_impl(s, nq, k, nprobe, metric, buf, len); });
}

static int pqv_searcher_footprint_impl(const pqv_searcher *s, uint64_t *row_order_bytes, uint64_t *ivf_rows_bytes,
                                       uint64_t *blocked_bytes, uint64_t *other_bytes) {
    if (!s) return fail(PQV_ERR_INVALID, "searcher must not be NULL");
    std::lock_guard<std::mutex> lock(s->mu);
    if (row_order_bytes) *row_order_bytes = s->corpus && s->corpus->d_rows ? s->corpus->capacity * s->corpus->dim * sizeof(float) : 0;
    if (ivf_rows_bytes) *ivf_rows_bytes = s->d_mat_ivf.p ? s->d_mat_ivf.bytes : 0;
    if (blocked_bytes) *blocked_bytes = (s->d_mat_blk_op[0].p ? s->d_mat_blk_op[0].bytes : 0) + (s->d_mat_blk_op[1].p ? s->d_mat_blk_op[1].bytes : 0) +
                                        (s->d_mat_blk_op[2].p ? s->d_mat_blk_op[2].bytes : 0);
    uint64_t other = s->d_centroids.bytes + s->d_cent_t.bytes + s->d_list_off.bytes + s->d_ids.bytes + s->d_stats.bytes + s->d_row_norm2.bytes + s->d_blk_off.bytes +
                     s->d_center.bytes + s->d_list_scale.bytes + s->d_list_half.bytes + s->d_list_radius.bytes + s->d_row_n2i.bytes + s->d_row_res.bytes;
    for (const Scratch &l : s->lanes)
        for (const DevBuf *b : {&l.s_probe_keys, &l.s_probe_vals, &l.s_probe, &l.s_cand_base, &l.s_ncand, &l.s_part_keys, &l.s_part_vals,
                                &l.s_queries, &l.s_rows, &l.s_dist, &l.s_nfound, &l.s_pair_u32, &l.s_pairs, &l.s_groups, &l.s_quads, &l.s_items, &l.s_ticket, &l.s_ticket2,
                                &l.s_cand_keys, &l.s_cand_vals, &l.s_cand_cnt, &l.s_spilled, &l.s_seed_ub, &l.s_qblk, &l.s_gthr, &l.s_tie,
                                &l.s_replay, &l.s_qnorm, &l.s_qmax, &l.s_thr_hist, &l.s_thr_bins, &l.s_qi8, &l.s_qn2i, &l.s_qres, &l.s_part_flags,
                                &l.s_qresu, &l.s_pair_lb, &l.s_qpad, &l.s_cand_lb, &l.s_pendv, &l.s_work, &l.s_nwork, &l.s_out})
            other += b->p ? b->bytes : 0;
    if (other_bytes) *other_bytes = other;
    return PQV_OK;
}
extern "C" int pqv_searcher_footprint(const pqv_searcher *s, uint64_t *row_order_bytes, uint64_t *ivf_rows_bytes,
                                      uint64_t *blocked_bytes, uint64_t *other_bytes) {
    return guard([&] { return pqv_searcher_footprint_impl(s, row_order_bytes, ivf_rows_bytes, blocked_bytes, other_bytes); });
}

static int pqv_probe_impl(const pqv_searcher *s, const float *query, uint32_t query_len, uint32_t nprobe,
                         uint32_t *clusters_out, uint32_t *n_out) {
    if (!s) return fail(PQV_ERR_INVALID, "searcher must not be NULL");
    if (nprobe == 0) return fail(PQV_ERR_INVALID, "nprobe must be > 0");
    if (query_len != s->dim)
        return fail(PQV_ERR_INVALID, "Query dimension mismatch: expected " + std::to_string(s->dim) +
                                         ", got " + std::to_string(query_len));
    if (!query || !clusters_out) return fail(PQV_ERR_INVALID, "query/clusters_out must not be NULL");
    const uint32_t np = std::min<uint32_t>(nprobe, s->n_clusters);
    if (int rc = use_device(s->device)) return rc;
    std::lock_guard<std::mutex> lock(s->mu);
    using namespace pqv;
    Scratch *lane = nullptr;
    if (int rc = lane_acquire(s, s->stream, &lane)) return rc;
    Scratch &sc = *lane;
    LaneGuard lane_guard{sc, s->stream};
    if (np > 1024) {            // beyond the kernels' sorted lists: distances on the GPU, the stable sort on the host
        HIP_TRY(sc.s_queries.ensure(static_cast<size_t>(s->dim) * sizeof(float)));
        HIP_TRY(hipMemcpyAsync(sc.s_queries.p, query, static_cast<size_t>(s->dim) * sizeof(float), hipMemcpyHostToDevice, s->stream));
        std::vector<uint32_t> order;
        if (int rc = centroid_order_host(s, sc, sc.s_queries.as<float>(), order)) return rc;
        std::memcpy(clusters_out, order.data(), static_cast<size_t>(np) * sizeof(uint32_t));
        if (n_out) *n_out = np;
        return lane_release(sc, s->stream);
    }
    const TopkPlan p = plan_topk(s, 1, nprobe);
    HIP_TRY(sc.s_queries.ensure(static_cast<size_t>(s->dim) * sizeof(float)));
    HIP_TRY(sc.s_probe_keys.ensure(static_cast<size_t>(p.n_part_probe) * p.probe_kpart * sizeof(uint64_t)));
    HIP_TRY(sc.s_probe_vals.ensure(static_cast<size_t>(p.n_part_probe) * p.probe_kpart * sizeof(uint32_t)));
    HIP_TRY(sc.s_probe.ensure(static_cast<size_t>(np) * sizeof(uint32_t)));
    HIP_TRY(sc.s_cand_base.ensure(static_cast<size_t>(np) * sizeof(uint64_t)));
    HIP_TRY(sc.s_ncand.ensure(sizeof(uint64_t)));
    HIP_TRY(hipMemcpyAsync(sc.s_queries.p, query, static_cast<size_t>(s->dim) * sizeof(float),
                           hipMemcpyHostToDevice, s->stream));
    StreamArgs pa{};
    pa.mat = s->d_centroids.as<float>(); pa.single_begin = 0; pa.single_end = s->n_clusters;
    pa.queries = sc.s_queries.as<float>(); pa.nq = 1; pa.nprobe = 1; pa.dim = s->dim; pa.k = np;
    pa.rows_per_block = 256; pa.blocks_per_list = p.probe_bpl; pa.max_pos = ~0ull; pa.metric = PQV_L2SQ_REF4;
    pa.part_keys = sc.s_probe_keys.as<uint64_t>(); pa.part_vals = sc.s_probe_vals.as<uint32_t>();
    if (p.probe_rows) {
        pqv::ProbeRowsArgs pr{};
        pr.cent_t = s->d_cent_t.as<float4>(); pr.queries = pa.queries;
        pr.nq = 1; pr.kc = s->n_clusters; pr.kc_pad = s->kc_pad; pr.dim = s->dim;
        pr.part_keys = pa.part_keys; pr.part_vals = pa.part_vals;
        HIP_TRY(pqv::launch_probe_rows(pr, s->stream));
    } else {
        HIP_TRY(launch_stream(pa, STREAM_TOPK, s->stream));
    }
    MergeArgs pm{};
    pm.part_keys = pa.part_keys; pm.part_vals = pa.part_vals; pm.nq = 1; pm.n_part = p.n_part_probe;
    pm.k_part = p.probe_kpart; pm.k = np; pm.list_off = s->d_list_off.as<uint64_t>();
    pm.probe = sc.s_probe.as<uint32_t>(); pm.cand_base = sc.s_cand_base.as<uint64_t>();
    pm.n_cand = sc.s_ncand.as<uint64_t>(); pm.max_pos = ~0ull;
    HIP_TRY(launch_merge_probe(pm, s->stream));
    HIP_TRY(hipMemcpyAsync(clusters_out, sc.s_probe.p, static_cast<size_t>(np) * sizeof(uint32_t),
                           hipMemcpyDeviceToHost, s->stream));
    if (int rc = lane_release(sc, s->stream)) return rc;
    HIP_TRY(hipStreamSynchronize(s->stream));
    if (n_out) *n_out = np;
    s->counters.kernel_launches += 2;
    return PQV_OK;
}
extern "C" int pqv_probe(const pqv_searcher *s, const float *query, uint32_t query_len, uint32_t nprobe,
                         uint32_t *clusters_out, uint32_t *n_out) {
    return guard([&] { return pqv_probe_impl(s, query, query_len, nprobe, clusters_out, n_out); });
}

static int pqv_candidate_rows_impl(const pqv_searcher *s, const float *query, uint32_t query_len,
                                  uint32_t nprobe, uint32_t **rows, uint64_t *n_rows) {
    if (!rows || !n_rows) return fail(PQV_ERR_INVALID, "rows/n_rows must not be NULL");
    *rows = nullptr; *n_rows = 0;
    if (!s) return fail(PQV_ERR_INVALID, "searcher must not be NULL");
    const uint32_t np = std::min<uint32_t>(nprobe, s->n_clusters);
    std::vector<uint32_t> clusters(std::max<uint32_t>(np, 1));
    uint32_t got = 0;
    if (int rc = pqv_probe(s, query, query_len, nprobe, clusters.data(), &got)) return rc;
    uint64_t total = 0;
    for (uint32_t i = 0; i < got; ++i) total += s->h_list_off[clusters[i] + 1] - s->h_list_off[clusters[i]];
    const std::vector<uint32_t> *h_rows = s->h_rows->get();             // (downloaded by the first call that reads them)
    if (!h_rows) return PQV_ERR_HIP;
    uint32_t *buf = static_cast<uint32_t *>(std::malloc(std::max<uint64_t>(1, total) * sizeof(uint32_t)));
    if (!buf) return fail(PQV_ERR_OOM, "host allocation failed");
    uint64_t o = 0;
    for (uint32_t i = 0; i < got; ++i) {  // probe-rank major, ascending ids inside (index.rs:59-62)
        const uint64_t b = s->h_list_off[clusters[i]], e = s->h_list_off[clusters[i] + 1];
        std::memcpy(buf + o, h_rows->data() + b, (e - b) * sizeof(uint32_t));
        o += e - b;
    }
    *rows = buf; *n_rows = total;
    s->counters.candidate_rows += total;
    return PQV_OK;
}
extern "C" int pqv_candidate_rows(const pqv_searcher *s, const float *query, uint32_t query_len,
                                  uint32_t nprobe, uint32_t **rows, uint64_t *n_rows) {
    return guard([&] { return pqv_candidate_rows_impl(s, query, query_len, nprobe, rows, n_rows); });
}

extern "C" void pqv_rows_free(uint32_t *rows) { std::free(rows); }

static int pqv_counters_impl(const pqv_searcher *s, pqv_counters_t *out) {
    if (!s || !out) return fail(PQV_ERR_INVALID, "searcher/out must not be NULL");
    if (int rc = use_device(s->device)) return rc;
    std::lock_guard<std::mutex> lock(s->mu);
    unsigned long long st[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    HIP_TRY(hipMemcpy(st, s->d_stats.p, sizeof st, hipMemcpyDeviceToHost));   // synchronises the device
#ifndef PQV_PROFILE_PHASES
    {   // wide_filter_kernel spreads its two counters over STATS_SLOTS lines
        std::vector<unsigned long long> slots(16 * pqv::STATS_SLOTS);
        HIP_TRY(hipMemcpy(slots.data(), static_cast<const unsigned long long *>(s->d_stats.p) + 8,
                          slots.size() * sizeof(unsigned long long), hipMemcpyDeviceToHost));
        for (uint32_t i = 0; i < pqv::STATS_SLOTS; ++i) {
            st[0] += slots[16 * i]; st[1] += slots[16 * i + 1]; st[2] += slots[16 * i + 2]; st[3] += slots[16 * i + 3];
        }
    }
#endif
#ifdef PQV_PROFILE_PHASES
    {
        std::vector<unsigned long long> rec(8 + 8 * 65536);
        HIP_TRY(hipMemcpy(rec.data(), s->d_stats.p, rec.size() * sizeof(unsigned long long), hipMemcpyDeviceToHost));
        if (const char *path = std::getenv("PQV_PHASES_OUT")) {
            if (FILE *f = std::fopen(path, "wb")) { std::fwrite(rec.data(), sizeof(unsigned long long), rec.size(), f); std::fclose(f); }
        }
    }
#endif
    *out = s->counters;
    out->screened_pairs = st[0];
    out->screen_survivors = st[1];
    out->candidate_rows += st[2];         // top-k calls (device counters) + pqv_candidate_rows (host counter)
    out->embeddings_fetched += st[3];
    return PQV_OK;
}
extern "C" int pqv_counters(const pqv_searcher *s, pqv_counters_t *out) {
    return guard([&] { return pqv_counters_impl(s, out); });
}

static int pqv_set_timing_impl(pqv_searcher *s, int enabled) {
    if (!s) return fail(PQV_ERR_INVALID, "searcher must not be NULL");
    std::lock_guard<std::mutex> lock(s->mu);
    s->timing = enabled != 0;
    return PQV_OK;
}
extern "C" int pqv_set_timing(pqv_searcher *s, int enabled) {
    return guard([&] { return pqv_set_timing_impl(s, enabled); });
}

static int pqv_timing_read_impl(const pqv_searcher *s, double *rerank_ms, double *total_ms,
                               uint32_t *n_calls) {
    if (!s) return fail(PQV_ERR_INVALID, "searcher must not be NULL");
    if (int rc = use_device(s->device)) return rc;
    std::lock_guard<std::mutex> lock(s->mu);
    double rr = 0.0, tot = 0.0;
    uint32_t calls = 0;
    for (size_t i = 0; i + 3 < s->ev.size(); i += 4) {
        HIP_TRY(hipEventSynchronize(s->ev[i + 3]));
        float a = 0.f, b = 0.f;
        HIP_TRY(hipEventElapsedTime(&a, s->ev[i + 1], s->ev[i + 2]));
        HIP_TRY(hipEventElapsedTime(&b, s->ev[i], s->ev[i + 3]));
        rr += a; tot += b; calls++;
    }
    for (auto e : s->ev) (void)hipEventDestroy(e);
    s->ev.clear();
    if (rerank_ms) *rerank_ms = rr;
    if (total_ms) *total_ms = tot;
    if (n_calls) *n_calls = calls;
    return PQV_OK;
}
extern "C" int pqv_timing_read(const pqv_searcher *s, double *rerank_ms, double *total_ms,
                               uint32_t *n_calls) {
    return guard([&] { return pqv_timing_read_impl(s, rerank_ms, total_ms, n_calls); });
}

// ---------------------------------------------------------------------------------------
// batched brute force on the matrix cores (BASELINE config 5; extension, see pqv.h)
// ---------------------------------------------------------------------------------------
static int pqv_brute_topk_impl(const pqv_corpus *c, const float *queries, uint32_t nq, uint32_t query_len,
                              uint32_t k, int metric, uint32_t *row_idx, float *dist, uint32_t *n_found) {
    using namespace pqv;
    if (!c) return fail(PQV_ERR_INVALID, "corpus must not be NULL");
    if (k == 0) return fail(PQV_ERR_INVALID, "k must be > 0");
    if (k > 1024) return fail(PQV_ERR_UNSUPPORTED, "k > 1024 is not supported");
    if (metric != PQV_COSINE && metric != PQV_L2SQ_MFMA)
        return fail(PQV_ERR_INVALID, "pqv_brute_topk: metric must be PQV_COSINE or PQV_L2SQ_MFMA");
    if (query_len != c->dim)
        return fail(PQV_ERR_INVALID, "Query dimension mismatch: expected " + std::to_string(c->dim) +
                                         ", got " + std::to_string(query_len));
    if (nq == 0) return PQV_OK;
    if (!queries || !row_idx || !dist) return fail(PQV_ERR_INVALID, "queries/row_idx/dist must not be NULL");
    if (!c->d_rows) return fail(PQV_ERR_INVALID, "corpus row-order copy was released");
    if (int rc = use_device(c->device)) return rc;
    hipStream_t stream = c->stream;
    const uint64_t n = c->n;
    const uint32_t dim = c->dim;
    const int mode = metric == PQV_COSINE ? 0 : 1;

    // per-row auxiliary (cached in the corpus, recomputed if rows were appended since)
    const float *d_row_aux = nullptr;
    {
        std::lock_guard<std::mutex> lock(c->aux_mu);
        DevBuf &buf = mode == 0 ? c->aux_rnorm : c->aux_norm2;
        uint64_t &have = mode == 0 ? c->aux_rnorm_rows : c->aux_norm2_rows;
        if (have != n || !buf.p) {
            HIP_TRY(buf.alloc(std::max<uint64_t>(1, n) * sizeof(float)));
            HIP_TRY(launch_row_norms(c->d_rows, n, dim, mode, buf.as<float>(), stream));
            HIP_TRY(hipStreamSynchronize(stream));
            have = n;
        }
        d_row_aux = buf.as<float>();
    }
    // Round 3: beyond the first row range the contraction runs on the f16 matrix pipe as a screen (kernels_brute.hip:
    // brute_f16_kernel) and only what it lets through is scored in f32.  Needs the normalised f16 image of the
    // corpus (+ 1 / |v| whatever the metric); PQV_BRUTE_F16=0 keeps the f32 contraction everywhere.
    // The screen's operand form: int8 images (twice the matrix rate, half the staged bytes; PQV_BRUTE_OP=f16 keeps the f16 images).
    static const bool f16_env = [] { const char *e = std::getenv("PQV_BRUTE_F16"); return !(e && *e == '0'); }();
    const bool i8_env = [] { const char *e = std::getenv("PQV_BRUTE_OP"); return !(e && !std::strcmp(e, "f16")); }();      // (read per call: tests run both forms)
    const bool use_i8 = i8_env;
    const uint32_t dim_p = use_i8 ? (dim + 63) / 64 * 64 : (dim + 31) / 32 * 32;
    const bool use_f16 = f16_env && n >= 32768 && static_cast<uint64_t>(dim_p) * 2 * 320 < 0x7FFFFFFFull;
    const uint16_t *d_v16 = nullptr;
    const int8_t *d_v8 = nullptr;
    const float4 *d_v8_sr = nullptr;
    const float *d_v8_n = nullptr;
    const float *d_vn2 = nullptr;
    if (use_f16 && use_i8) {
        std::lock_guard<std::mutex> lock(c->aux_mu);
        if (c->aux_rnorm_rows != n || !c->aux_rnorm.p) {
            HIP_TRY(c->aux_rnorm.alloc(std::max<uint64_t>(1, n) * sizeof(float)));
            HIP_TRY(launch_row_norms(c->d_rows, n, dim, 0, c->aux_rnorm.as<float>(), stream));
            c->aux_rnorm_rows = n;
        }
        if (mode == 1) d_vn2 = d_row_aux;
        if (c->aux_v8_rows != n || !c->aux_v8.p) {
            HIP_TRY(c->aux_v8.alloc(std::max<uint64_t>(1, n) * dim_p));
            HIP_TRY(c->aux_v8_sr.alloc(std::max<uint64_t>(1, n) * sizeof(float4)));
            HIP_TRY(c->aux_v8_n.alloc(std::max<uint64_t>(1, n) * sizeof(float)));
            HIP_TRY(c->aux_v8_max.alloc(12 * sizeof(float)));
            HIP_TRY(hipMemsetAsync(c->aux_v8_max.p, 0, 12 * sizeof(float), stream));
            HIP_TRY(launch_normalize_i8(c->d_rows, c->aux_rnorm.as<float>(), n, dim, dim_p, c->aux_v8.p, c->aux_v8_sr.p, c->aux_v8_n.as<float>(),
                                        c->aux_v8_max.as<float>(), stream));
            HIP_TRY(hipStreamSynchronize(stream));
            c->aux_v8_rows = n;
        }
        d_v8 = c->aux_v8.as<int8_t>(); d_v8_sr = c->aux_v8_sr.as<float4>(); d_v8_n = c->aux_v8_n.as<float>();
    } else if (use_f16) {
        std::lock_guard<std::mutex> lock(c->aux_mu);
        if (c->aux_rnorm_rows != n || !c->aux_rnorm.p) {
            HIP_TRY(c->aux_rnorm.alloc(std::max<uint64_t>(1, n) * sizeof(float)));
            HIP_TRY(launch_row_norms(c->d_rows, n, dim, 0, c->aux_rnorm.as<float>(), stream));
            c->aux_rnorm_rows = n;
        }
        if (mode == 1) d_vn2 = d_row_aux;
        if (c->aux_v16_rows != n || !c->aux_v16.p) {
            HIP_TRY(c->aux_v16.alloc(std::max<uint64_t>(1, n) * dim_p * sizeof(uint16_t)));
            HIP_TRY(launch_normalize_f16(c->d_rows, c->aux_rnorm.as<float>(), n, dim, dim_p, c->aux_v16.p, stream));
            HIP_TRY(hipStreamSynchronize(stream));
            c->aux_v16_rows = n;
        }
        d_v16 = c->aux_v16.as<uint16_t>();
    }
    // (int8: the image bound is per pair, inside the kernel; eps carries the f32 roundings of the normalisation and of both sums)
    const float f16_eps = use_i8 ? 2.0f * static_cast<float>(dim_p) * 5.9604645e-08f + 4.0e-6f
                                 : 1.01f * 9.765625e-04f + 2.0f * static_cast<float>(dim_p) * 5.9604645e-08f + 2.0e-6f;

    const uint32_t cap = std::max<uint32_t>(16384, 4 * k);
    const uint32_t qbatch = std::min<uint32_t>(nq, 4096);
    DevBuf d_q, d_qaux, d_cand, d_cnt, d_cnt_saved, d_thr, d_flag, d_rows_out, d_dist_out, d_nf;
    HIP_TRY(d_q.alloc(static_cast<size_t>(qbatch) * dim * sizeof(float)));
    HIP_TRY(d_qaux.alloc(static_cast<size_t>(qbatch) * sizeof(float)));
    HIP_TRY(d_cand.alloc(static_cast<size_t>(qbatch) * cap * sizeof(unsigned long long)));
    HIP_TRY(d_cnt.alloc(static_cast<size_t>(qbatch) * sizeof(uint32_t)));
    HIP_TRY(d_cnt_saved.alloc(static_cast<size_t>(qbatch) * sizeof(uint32_t)));
    HIP_TRY(d_thr.alloc(static_cast<size_t>(qbatch) * sizeof(unsigned long long)));
    HIP_TRY(d_flag.alloc(sizeof(uint32_t)));
    HIP_TRY(d_rows_out.alloc(static_cast<size_t>(qbatch) * k * sizeof(uint32_t)));
    HIP_TRY(d_dist_out.alloc(static_cast<size_t>(qbatch) * k * sizeof(float)));
    HIP_TRY(d_nf.alloc(static_cast<size_t>(qbatch) * sizeof(uint32_t)));
    DevBuf d_q16, d_qrn, d_qsr, d_qn;
    if (use_f16) {
        HIP_TRY(d_q16.alloc(static_cast<size_t>(qbatch) * dim_p * sizeof(uint16_t)));
        HIP_TRY(d_qrn.alloc(static_cast<size_t>(qbatch) * sizeof(float)));
        if (use_i8) { HIP_TRY(d_qsr.alloc(static_cast<size_t>(qbatch) * sizeof(float4))); HIP_TRY(d_qn.alloc(static_cast<size_t>(qbatch) * sizeof(float))); }
    }

    for (uint32_t q0 = 0; q0 < nq; q0 += qbatch) {
        const uint32_t b = std::min<uint32_t>(qbatch, nq - q0);
        HIP_TRY(hipMemcpyAsync(d_q.p, queries + static_cast<uint64_t>(q0) * dim, static_cast<size_t>(b) * dim * sizeof(float),
                               hipMemcpyHostToDevice, stream));
        HIP_TRY(launch_row_norms(d_q.as<float>(), b, dim, mode, d_qaux.as<float>(), stream));
        if (use_f16) {
            HIP_TRY(launch_row_norms(d_q.as<float>(), b, dim, 0, d_qrn.as<float>(), stream));
            if (use_i8) HIP_TRY(launch_normalize_i8(d_q.as<float>(), d_qrn.as<float>(), b, dim, dim_p, d_q16.p, d_qsr.p, d_qn.as<float>(), nullptr, stream));
            else HIP_TRY(launch_normalize_f16(d_q.as<float>(), d_qrn.as<float>(), b, dim, dim_p, d_q16.p, stream));
        }
        HIP_TRY(hipMemsetAsync(d_cnt.p, 0, static_cast<size_t>(b) * sizeof(uint32_t), stream));
        HIP_TRY(hipMemsetAsync(d_thr.p, 0xFF, static_cast<size_t>(b) * sizeof(unsigned long long), stream));
        HIP_TRY(hipMemsetAsync(d_flag.p, 0, sizeof(uint32_t), stream));

        // Progressive thresholds: row ranges grow 8x; after each range the per-query k-th key
        // becomes the admission threshold of the next, so only ~8k candidates per query
        // survive each pass.  A range whose survivors overflow a buffer (adversarially ordered
        // rows) is rolled back and split.
        std::vector<std::pair<uint64_t, uint64_t>> todo;
        {
            std::vector<std::pair<uint64_t, uint64_t>> fwd;
            // (the first range has no threshold yet, so EVERY pair of it is appended -- one atomic each: 8192 rows cost 3.8 ms on
            //  C5, a tenth of the step; 2048 rows seed thresholds that the next range tightens anyway)
            uint64_t lo = 0, len = std::min<uint64_t>(n, std::max<uint64_t>(use_f16 ? 2048 : 8192, 8ull * k));
            while (lo < n) {
                const uint64_t hi = std::min<uint64_t>(n, lo + len);
                fwd.emplace_back(lo, hi);
                lo = hi;
                len *= 8;
            }
            todo.assign(fwd.rbegin(), fwd.rend());   // stack: pop_back() yields ascending ranges
        }
        bool first_range = true;
        while (!todo.empty()) {
            const auto range = todo.back();
            todo.pop_back();
            HIP_TRY(hipMemcpyAsync(d_cnt_saved.p, d_cnt.p, static_cast<size_t>(b) * sizeof(uint32_t),
                                   hipMemcpyDeviceToDevice, stream));
            BruteArgs ba{};
            ba.rows = c->d_rows; ba.queries = d_q.as<float>(); ba.row_aux = d_row_aux; ba.query_aux = d_qaux.as<float>();
            ba.row_begin = range.first; ba.row_end = range.second; ba.nq = b; ba.dim = dim;
            ba.metric = mode == 0 ? BRUTE_COSINE : BRUTE_L2SQ;
            ba.thr = d_thr.as<unsigned long long>(); ba.cand = d_cand.as<unsigned long long>();
            ba.cand_cnt = d_cnt.as<uint32_t>(); ba.cap = cap;
            // the first range seeds the thresholds exactly; every later one is screened on the f16 pipe
            const bool screen = use_f16 && range.first > 0;
            if (screen) {
                BruteF16Args fa{};
                fa.v16 = d_v16; fa.q16 = d_q16.as<uint16_t>(); fa.row_aux = d_vn2; fa.query_aux = d_qaux.as<float>();
                if (use_i8) {
                    fa.v8 = d_v8; fa.q8 = d_q16.as<int8_t>(); fa.row_sr = d_v8_sr; fa.query_sr = d_qsr.as<float4>();
                    fa.row_n = d_v8_n; fa.query_n = d_qn.as<float>(); fa.dim_f = static_cast<float>(dim); fa.row_max = c->aux_v8_max.as<float>();
                    // a wave sums dim / 64 components per lane + 6 exchange steps: error <= (dim / 64 + 7) 2^-24 sum |v^_i| <= .. sqrt(dim)
                    fa.eps_sum = (static_cast<float>(dim) / 64.0f + 8.0f) * 5.9604645e-08f * std::sqrt(static_cast<float>(dim)) * 1.01f;
                }
                fa.row_begin = range.first; fa.row_end = range.second; fa.nq = b; fa.dim_p = dim_p;
                fa.metric = ba.metric; fa.eps = f16_eps;
                fa.thr = ba.thr; fa.cand = ba.cand; fa.cand_cnt = ba.cand_cnt; fa.cap = cap;
                HIP_TRY(launch_brute_f16(fa, stream));
            } else {
                // the very first range: nothing to compare with yet, every pair is kept -- written to its own slot instead of one
                // atomic append per pair (2048 rows x 1024 queries: 1.07 -> 0.1 ms on C5)
                const bool dense = first_range && range.second - range.first <= cap;
                ba.dense = dense ? 1u : 0u;
                HIP_TRY(launch_brute_mfma(ba, stream));
                if (dense) HIP_TRY(hipMemsetD32Async(static_cast<hipDeviceptr_t>(d_cnt.p), static_cast<int>(range.second - range.first), b, stream));
            }
            first_range = false;
            // overflow is checked BEFORE the select pass touches the buffer fronts, so a
            // rollback only has to restore the counts
            HIP_TRY(launch_brute_overflow_check(d_cnt.as<uint32_t>(), b, cap, d_flag.as<uint32_t>(), stream));
            uint32_t flag = 0;
            HIP_TRY(hipMemcpyAsync(&flag, d_flag.p, sizeof flag, hipMemcpyDeviceToHost, stream));
            HIP_TRY(hipStreamSynchronize(stream));
            if (flag) {
                const uint64_t len = range.second - range.first;
                if (len <= cap / 2)
                    return fail(PQV_ERR_HIP, "pqv_brute_topk: candidate buffer overflow on a minimal range");
                HIP_TRY(hipMemcpyAsync(d_cnt.p, d_cnt_saved.p, static_cast<size_t>(b) * sizeof(uint32_t),
                                       hipMemcpyDeviceToDevice, stream));
                HIP_TRY(hipMemsetAsync(d_flag.p, 0, sizeof(uint32_t), stream));
                const uint64_t mid = range.first + len / 2;
                todo.emplace_back(mid, range.second);
                todo.emplace_back(range.first, mid);
                continue;
            }
            if (screen && std::getenv("PQV_BRUTE_STATS")) {        // diagnostic: pairs the screen let through in this range
                std::vector<uint32_t> h1(b), h0(b);
                HIP_TRY(hipMemcpy(h1.data(), d_cnt.p, b * sizeof(uint32_t), hipMemcpyDeviceToHost));
                HIP_TRY(hipMemcpy(h0.data(), d_cnt_saved.p, b * sizeof(uint32_t), hipMemcpyDeviceToHost));
                uint64_t t = 0;
                for (uint32_t i = 0; i < b; ++i) t += h1[i] - h0[i];
                std::fprintf(stderr, "pqv_brute_topk: rows [%llu, %llu): %.1f screen survivors per query\n",
                             (unsigned long long)range.first, (unsigned long long)range.second, (double)t / b);
            }
            if (screen) HIP_TRY(launch_brute_rescore(ba, d_cnt_saved.as<uint32_t>(), stream));     // exact f32 keys for what the screen let through
            HIP_TRY(launch_brute_select(d_cand.as<unsigned long long>(), d_cnt.as<uint32_t>(), cap, b, k,
                                        d_thr.as<unsigned long long>(), d_flag.as<uint32_t>(), stream));
        }
        HIP_TRY(launch_brute_finish(d_cand.as<unsigned long long>(), d_cnt.as<uint32_t>(), cap, b, k,
                                    d_rows_out.as<uint32_t>(), d_dist_out.as<float>(), d_nf.as<uint32_t>(), stream));
        HIP_TRY(hipMemcpyAsync(row_idx + static_cast<uint64_t>(q0) * k, d_rows_out.p, static_cast<size_t>(b) * k * sizeof(uint32_t),
                               hipMemcpyDeviceToHost, stream));
        HIP_TRY(hipMemcpyAsync(dist + static_cast<uint64_t>(q0) * k, d_dist_out.p, static_cast<size_t>(b) * k * sizeof(float),
                               hipMemcpyDeviceToHost, stream));
        if (n_found)
            HIP_TRY(hipMemcpyAsync(n_found + q0, d_nf.p, static_cast<size_t>(b) * sizeof(uint32_t), hipMemcpyDeviceToHost, stream));
        HIP_TRY(hipStreamSynchronize(stream));
    }
    return PQV_OK;
}
extern "C" int pqv_brute_topk(const pqv_corpus *c, const float *queries, uint32_t nq, uint32_t query_len,
                              uint32_t k, int metric, uint32_t *row_idx, float *dist, uint32_t *n_found) {
    return guard([&] { return pqv_brute_topk_impl(c, queries, nq, query_len, k, metric, row_idx, dist, n_found); });
}

// ---------------------------------------------------------------------------------------
// host-side merge of per-shard lists (multi-file / multi-GPU)
// ---------------------------------------------------------------------------------------
static int pqv_merge_topk_impl(const float *dist, const uint32_t *rows, const uint32_t *counts,
                              uint32_t n_lists, uint32_t nq, uint32_t k, float *out_dist,
                              uint32_t *out_rows, uint32_t *out_list, uint32_t *out_count) {
    if (!dist || !rows || !counts || !out_dist || !out_rows)
        return fail(PQV_ERR_INVALID, "merge arrays must not be NULL");
    if (k == 0) return fail(PQV_ERR_INVALID, "k must be > 0");
    struct Item { float d; uint32_t list, pos, row; };
    std::vector<Item> items;
    for (uint32_t q = 0; q < nq; ++q) {
        items.clear();
        for (uint32_t l = 0; l < n_lists; ++l) {
            const uint32_t cnt = std::min<uint32_t>(counts[static_cast<uint64_t>(l) * nq + q], k);
            const uint64_t base = (static_cast<uint64_t>(l) * nq + q) * k;
            for (uint32_t i = 0; i < cnt; ++i) items.push_back({dist[base + i], l, i, rows[base + i]});
        }
        std::stable_sort(items.begin(), items.end(), [](const Item &a, const Item &b) {
            if (a.d < b.d) return true;
            if (b.d < a.d) return false;
            if (a.list != b.list) return a.list < b.list;
            return a.pos < b.pos;
        });
        const uint32_t take = static_cast<uint32_t>(std::min<size_t>(k, items.size()));
        for (uint32_t i = 0; i < k; ++i) {
            const uint64_t o = static_cast<uint64_t>(q) * k + i;
            if (i < take) {
                out_dist[o] = items[i].d; out_rows[o] = items[i].row;
                if (out_list) out_list[o] = items[i].list;
            } else {
                out_dist[o] = INFINITY; out_rows[o] = 0xFFFFFFFFu;
                if (out_list) out_list[o] = 0xFFFFFFFFu;
            }
        }
        if (out_count) out_count[q] = take;
    }
    return PQV_OK;
}
extern "C" int pqv_merge_topk(const float *dist, const uint32_t *rows, const uint32_t *counts,
                              uint32_t n_lists, uint32_t nq, uint32_t k, float *out_dist,
                              uint32_t *out_rows, uint32_t *out_list, uint32_t *out_count) {
    return guard([&] { return pqv_merge_topk_impl(dist, rows, counts, n_lists, nq, k, out_dist, out_rows, out_list, out_count); });
}

static int pqv_merge_topk_device_impl(int device, const void *d_dist, const void *d_rows, const void *d_row_base,
                                      uint32_t n_lists, uint32_t nq, uint32_t k, void *d_out_dist,
                                      void *d_out_rows, void *hip_stream) {
    if (!d_dist || !d_rows || !d_row_base || !d_out_dist || !d_out_rows)
        return fail(PQV_ERR_INVALID, "device pointers must not be NULL");
    if (k == 0) return fail(PQV_ERR_INVALID, "k must be > 0");
    if (k > 1024) return fail(PQV_ERR_UNSUPPORTED, "k > 1024 is not supported");
    if (n_lists == 0 || nq == 0) return PQV_OK;
    if (int rc = use_device(device)) return rc;
    HIP_TRY(pqv::launch_shard_merge(static_cast<const float *>(d_dist), static_cast<const uint32_t *>(d_rows),
                                    static_cast<const long long *>(d_row_base), n_lists, nq, k,
                                    static_cast<float *>(d_out_dist), static_cast<long long *>(d_out_rows),
                                    static_cast<hipStream_t>(hip_stream)));
    return PQV_OK;
}
extern "C" int pqv_merge_topk_device(int device, const void *d_dist, const void *d_rows, const void *d_row_base,
                                     uint32_t n_lists, uint32_t nq, uint32_t k, void *d_out_dist,
                                     void *d_out_rows, void *hip_stream) {
    return guard([&] { return pqv_merge_topk_device_impl(device, d_dist, d_rows, d_row_base, n_lists, nq, k,
                                                         d_out_dist, d_out_rows, hip_stream); });
}

static int pqv_merge_topk_packed_device_impl(int device, const void *d_pairs, const void *d_row_base, uint32_t n_lists,
                                             uint32_t nq, uint32_t k, void *d_out_dist, void *d_out_rows, void *hip_stream) {
    if (!d_pairs || !d_row_base || !d_out_dist || !d_out_rows)
        return fail(PQV_ERR_INVALID, "device pointers must not be NULL");
    if (k == 0) return fail(PQV_ERR_INVALID, "k must be > 0");
    if (k > 1024) return fail(PQV_ERR_UNSUPPORTED, "k > 1024 is not supported");
    if (n_lists == 0 || nq == 0) return PQV_OK;
    if (int rc = use_device(device)) return rc;
    const float *base = static_cast<const float *>(d_pairs);
    HIP_TRY(pqv::launch_shard_merge(base, reinterpret_cast<const uint32_t *>(base) + 1,
                                    static_cast<const long long *>(d_row_base), n_lists, nq, k,
                                    static_cast<float *>(d_out_dist), static_cast<long long *>(d_out_rows),
                                    static_cast<hipStream_t>(hip_stream), 2));
    return PQV_OK;
}
extern "C" int pqv_merge_topk_packed_device(int device, const void *d_pairs, const void *d_row_base, uint32_t n_lists,
                                            uint32_t nq, uint32_t k, void *d_out_dist, void *d_out_rows, void *hip_stream) {
    return guard([&] { return pqv_merge_topk_packed_device_impl(device, d_pairs, d_row_base, n_lists, nq, k, d_out_dist,
                                                                d_out_rows, hip_stream); });
}

// ---------------------------------------------------------------------------------------
// CandidateCursor (src/df_vector/access.rs:193-243)
// ---------------------------------------------------------------------------------------
struct pqv_candidate_cursor {
    std::vector<std::vector<uint32_t>> candidates;
    std::vector<uint64_t> positions;
    uint64_t round_robin = 0;
};
extern "C" int pqv_candidate_cursor_new(uint32_t file_count, pqv_candidate_cursor **out) {
    return guard([&] {
        if (!out) return fail(PQV_ERR_INVALID, "out must not be NULL");
        pqv_candidate_cursor *c = new pqv_candidate_cursor();
        c->candidates.resize(file_count);                                              // :202-206
        c->positions.assign(file_count, 0);
        *out = c;
        return static_cast<int>(PQV_OK);
    });
}
extern "C" int pqv_candidate_cursor_add(pqv_candidate_cursor *c, uint32_t idx, const uint32_t *rows, uint64_t n_rows) {
    return guard([&] {
        if (!c || (n_rows && !rows)) return fail(PQV_ERR_INVALID, "cursor/rows must not be NULL");
        if (idx < c->candidates.size()) c->candidates[idx].assign(rows, rows + n_rows);  // :209-213 (out of range: ignored)
        return static_cast<int>(PQV_OK);
    });
}
extern "C" int pqv_candidate_cursor_next_batch(pqv_candidate_cursor *c, uint64_t batch_size, uint32_t *out_file,
                                               uint32_t *out_row, uint64_t *n_out, uint64_t *per_file_taken) {
    return guard([&] {
        if (!c || !n_out || (batch_size && (!out_file || !out_row))) return fail(PQV_ERR_INVALID, "cursor/outputs must not be NULL");
        *n_out = 0;
        const uint64_t file_count = c->candidates.size();
        uint64_t n = 0;
        if (batch_size != 0 && file_count != 0) {                                      // :216-218
            uint64_t idx = c->round_robin;
            while (n < batch_size) {                                                   // :222-238
                bool progressed = false;
                for (uint64_t t = 0; t < file_count; ++t) {
                    const uint64_t f = idx % file_count;
                    idx += 1;
                    if (c->positions[f] < c->candidates[f].size()) {
                        out_file[n] = static_cast<uint32_t>(f);
                        out_row[n] = c->candidates[f][c->positions[f]++];
                        ++n;
                        progressed = true;
                        if (n >= batch_size) break;
                    }
                }
                if (!progressed) break;
            }
            c->round_robin = idx % file_count;                                         // :240
        }
        *n_out = n;
        if (per_file_taken) for (uint64_t f = 0; f < file_count; ++f) per_file_taken[f] = c->positions[f];
        return static_cast<int>(PQV_OK);
    });
}
extern "C" void pqv_candidate_cursor_free(pqv_candidate_cursor *c) { delete c; }

// ---------------------------------------------------------------------------------------
// batch-granular re-rank (update_topk_heap, src/df_vector/exec.rs:457-484)
// ---------------------------------------------------------------------------------------
// Scratch of the batch-granular re-rank, pooled per device: steady-state calls allocate nothing (the first version
// paid ten hipMallocs and six synchronous copies per 2048-row RecordBatch -- the allocation-bound pattern SURVEY D1
// criticises in row_to_scalar_values).  A context owns a private stream, device buffers grown on demand and pinned
// staging for the host-buffer entry point.
namespace {
struct RerankCtx {
    int device = 0;
    hipStream_t stream = nullptr;
    DevBuf d_cand, d_q, d_keys, d_vals, d_mvals, d_mdist, d_nf, d_saved, d_base, d_probe0, d_off, d_io_rows, d_io_d2, d_io_cnt, d_ids;
    PinnedBuf h_stage;
    uint64_t off_m = ~0ull, base_k = ~0ull;
    ~RerankCtx() { if (stream) (void)hipStreamDestroy(stream); }
};
std::mutex g_rerank_mu;
std::vector<RerankCtx *> g_rerank_pool;

int rerank_ctx_acquire(int device, RerankCtx **out) {
    {
        std::lock_guard<std::mutex> lock(g_rerank_mu);
        for (size_t i = 0; i < g_rerank_pool.size(); ++i)
            if (g_rerank_pool[i]->device == device) {
                *out = g_rerank_pool[i];
                g_rerank_pool.erase(g_rerank_pool.begin() + static_cast<long>(i));
                return PQV_OK;
            }
    }
    RerankCtx *c = new RerankCtx();
    c->device = device;
    hipError_t e = hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking);
    if (e != hipSuccess) { delete c; return fail(PQV_ERR_HIP, std::string("hipStreamCreate: ") + hipGetErrorString(e)); }
    *out = c;
    return PQV_OK;
}
void rerank_ctx_release(RerankCtx *c) {
    std::lock_guard<std::mutex> lock(g_rerank_mu);
    if (g_rerank_pool.size() < 16) g_rerank_pool.push_back(c); else delete c;
}

// The device-resident fold: d_cand [m, dim] and the running state all on the device; enqueued on `stream`.
int rerank_enqueue(RerankCtx &c, const float *d_query, const float *d_cand, const uint32_t *d_ids, uint64_t m, uint32_t dim,
                   uint32_t k_out, int metric, uint32_t *d_io_rows, float *d_io_d2, uint32_t *d_io_count, uint32_t *d_tie,
                   hipStream_t stream) {
    using namespace pqv;
    // with a tie flag the lists carry one extra entry (the runner-up), as pqv_topk_device_flags does
    const uint32_t k = d_tie ? k_out + 1 : k_out;
    const uint32_t bpl = static_cast<uint32_t>((m + 255) / 256);
    const uint32_t n_part = bpl * waves_per_block() + 1;  // +1: the running state
    HIP_TRY(c.d_keys.ensure(static_cast<size_t>(n_part) * k * sizeof(uint64_t)));
    HIP_TRY(c.d_vals.ensure(static_cast<size_t>(n_part) * k * sizeof(uint32_t)));
    HIP_TRY(c.d_mvals.ensure(k * sizeof(uint32_t)));
    HIP_TRY(c.d_mdist.ensure(k * sizeof(float)));
    HIP_TRY(c.d_nf.ensure(2 * sizeof(uint32_t)));
    HIP_TRY(c.d_saved.ensure(k * sizeof(uint32_t)));
    HIP_TRY(c.d_base.ensure(sizeof(uint64_t))); HIP_TRY(c.d_probe0.ensure(sizeof(uint32_t))); HIP_TRY(c.d_off.ensure(2 * sizeof(uint64_t)));
    if (c.off_m != m || c.base_k != k_out) {      // the one-list descriptor of the stream kernel (changes with the batch size only)
        const uint64_t h_base = k_out, h_off[2] = {0, m};
        const uint32_t h_probe0 = 0;
        c.off_m = ~0ull; c.base_k = ~0ull;        // nothing cached until all three copies have landed
        HIP_TRY(hipMemcpyAsync(c.d_base.p, &h_base, sizeof h_base, hipMemcpyHostToDevice, stream));
        HIP_TRY(hipMemcpyAsync(c.d_probe0.p, &h_probe0, sizeof h_probe0, hipMemcpyHostToDevice, stream));
        HIP_TRY(hipMemcpyAsync(c.d_off.p, h_off, sizeof h_off, hipMemcpyHostToDevice, stream));
        HIP_TRY(hipStreamSynchronize(stream));        // the sources are stack variables
        c.off_m = m; c.base_k = k_out;
    }
    // running state as the first partial list: positions 0..count-1 (earlier arrivals win ties)
    HIP_TRY(launch_rerank_state_in(d_io_rows, d_io_d2, d_io_count, k_out, k, c.d_keys.as<uint64_t>(), c.d_vals.as<uint32_t>(),
                                   c.d_saved.as<uint32_t>(), stream));
    // batch candidates take positions k, k+1, ... via the cand_base entry
    StreamArgs ra{};
    ra.mat = d_cand; ra.row_of = nullptr; ra.list_off = c.d_off.as<uint64_t>();
    ra.probe = c.d_probe0.as<uint32_t>(); ra.cand_base = c.d_base.as<uint64_t>();
    ra.queries = d_query; ra.nq = 1; ra.nprobe = 1; ra.dim = dim; ra.k = k;
    ra.rows_per_block = 256; ra.blocks_per_list = bpl; ra.max_pos = ~0ull; ra.metric = metric;
    ra.part_keys = c.d_keys.as<uint64_t>() + k; ra.part_vals = c.d_vals.as<uint32_t>() + k;
    HIP_TRY(launch_stream(ra, STREAM_TOPK, stream));
    MergeArgs fm{};
    fm.part_keys = c.d_keys.as<uint64_t>(); fm.part_vals = c.d_vals.as<uint32_t>();
    fm.nq = 1; fm.n_part = n_part; fm.k_part = k; fm.k = k; fm.ids = nullptr;
    fm.row_idx = c.d_mvals.as<uint32_t>(); fm.dist = c.d_mdist.as<float>(); fm.n_found = c.d_nf.as<uint32_t>();
    fm.sqrt_out = 0; fm.k_out = k_out; fm.tie_flag = d_tie ? c.d_nf.as<uint32_t>() + 1 : nullptr;
    HIP_TRY(launch_merge_final(fm, stream));
    HIP_TRY(launch_rerank_state_out(c.d_mvals.as<uint32_t>(), c.d_mdist.as<float>(), c.d_nf.as<uint32_t>(), c.d_saved.as<uint32_t>(),
                                    d_ids, k_out, d_io_rows, d_io_d2, d_io_count, d_tie ? c.d_nf.as<uint32_t>() + 1 : nullptr, d_tie, stream));
    return PQV_OK;
}

int rerank_validate(const void *query, const void *io_rows, const void *io_d2, const void *io_count, const void *cand,
                    uint64_t m, uint32_t dim, uint32_t k, int metric) {
    if (!query || !io_rows || !io_d2 || !io_count) return fail(PQV_ERR_INVALID, "query/io arrays must not be NULL");
    if (dim == 0) return fail(PQV_ERR_INVALID, "Embedding dimension must be > 0");
    if (k > 1024) return fail(PQV_ERR_UNSUPPORTED, "k > 1024 is not supported");
    if (metric != PQV_L2SQ_REF4 && metric != PQV_L2SQ_SEQ) return fail(PQV_ERR_INVALID, "unknown metric");
    if (m && !cand) return fail(PQV_ERR_INVALID, "cand must not be NULL");
    if (m > 0x7FFFFFFFull) return fail(PQV_ERR_UNSUPPORTED, "batch larger than 2^31 rows");
    return PQV_OK;
}
}  // namespace

static int pqv_rerank_device_impl(int device, const void *d_query, const void *d_cand, const void *d_ids, uint64_t m,
                                  uint32_t dim, uint32_t k, int metric, void *d_io_rows, void *d_io_d2, void *d_io_count,
                                  void *d_tie_flag, void *hip_stream) {
    if (int rc = rerank_validate(d_query, d_io_rows, d_io_d2, d_io_count, d_cand, m, dim, k, metric)) return rc;
    if (d_tie_flag && k > 1023) return fail(PQV_ERR_UNSUPPORTED, "tie flags need a runner-up entry: k <= 1023");
    if (k == 0 || m == 0) return PQV_OK;
    if (int rc = use_device(device)) return rc;
    RerankCtx *c = nullptr;
    if (int rc = rerank_ctx_acquire(device, &c)) return rc;
    hipStream_t stream = hip_stream ? static_cast<hipStream_t>(hip_stream) : c->stream;
    int rc = rerank_enqueue(*c, static_cast<const float *>(d_query), static_cast<const float *>(d_cand),
                            static_cast<const uint32_t *>(d_ids), m, dim, k, metric, static_cast<uint32_t *>(d_io_rows),
                            static_cast<float *>(d_io_d2), static_cast<uint32_t *>(d_io_count), static_cast<uint32_t *>(d_tie_flag), stream);
    // the context's merge scratch is in use until the stream drains -- also when the enqueue failed half-way: hand it
    // back only then
    if (hipStreamSynchronize(stream) != hipSuccess && rc == PQV_OK) rc = fail(PQV_ERR_HIP, "hipStreamSynchronize failed");
    rerank_ctx_release(c);
    return rc;
}
extern "C" int pqv_rerank_device(int device, const void *d_query, const void *d_cand, const void *d_ids, uint64_t m,
                                 uint32_t dim, uint32_t k, int metric, void *d_io_rows, void *d_io_d2, void *d_io_count,
                                 void *hip_stream) {
    return guard([&] { return pqv_rerank_device_impl(device, d_query, d_cand, d_ids, m, dim, k, metric, d_io_rows, d_io_d2, d_io_count, nullptr, hip_stream); });
}
extern "C" int pqv_rerank_device_flags(int device, const void *d_query, const void *d_cand, const void *d_ids, uint64_t m,
                                       uint32_t dim, uint32_t k, int metric, void *d_io_rows, void *d_io_d2, void *d_io_count,
                                       void *d_tie_flag, void *hip_stream) {
    if (!d_tie_flag) return fail(PQV_ERR_INVALID, "d_tie_flag must not be NULL");
    return guard([&] { return pqv_rerank_device_impl(device, d_query, d_cand, d_ids, m, dim, k, metric, d_io_rows, d_io_d2, d_io_count, d_tie_flag, hip_stream); });
}

// Host-buffer fold with the reference's EXACT heap mechanics (exec.rs:457-484): the distances of the batch are
// computed on the GPU (stream_kernel, STREAM_DIST, in compute_distance_values' order), then the rows walk through
// std's BinaryHeap push / peek / pop on the host in arrival order -- an O(m) compare loop plus O(log k) per admitted
// row.  io_rows / io_d2 ARE the heap's backing array between batches, so ties (which make survivors and order depend
// on sift history) come out as in Rust; pqv_rerank_finish is heap.into_iter() + the stable sort of exec.rs:269-274.
template <class T>
static int rerank_host(int device, const float *query, const T *cand, const uint32_t *ids,
                       const uint8_t *valid, uint64_t m, uint32_t dim, uint32_t k, int metric,
                       uint32_t *io_rows, float *io_d2, uint32_t *io_count) {
    if (!query || !io_rows || !io_d2 || !io_count) return fail(PQV_ERR_INVALID, "query/io arrays must not be NULL");
    if (dim == 0) return fail(PQV_ERR_INVALID, "Embedding dimension must be > 0");
    if (metric != PQV_L2SQ_REF4 && metric != PQV_L2SQ_SEQ) return fail(PQV_ERR_INVALID, "unknown metric");
    if (m && !cand) return fail(PQV_ERR_INVALID, "cand must not be NULL");
    if (m > 0x7FFFFFFFull) return fail(PQV_ERR_UNSUPPORTED, "batch larger than 2^31 rows");
    if (k == 0) return PQV_OK;  // heap.len() < 0 is never true and peek() is None: nothing is kept
    if (*io_count > k) return fail(PQV_ERR_INVALID, "io_count exceeds k");
    if (m == 0) return PQV_OK;
    if (int rc = use_device(device)) return rc;
    uint64_t mv = m;
    if (valid) { mv = 0; for (uint64_t i = 0; i < m; ++i) mv += valid[i] != 0; }
    if (mv == 0) return PQV_OK;
    RerankCtx *c = nullptr;
    if (int rc = rerank_ctx_acquire(device, &c)) return rc;
    // the context goes back to the pool only with its stream drained (a failed call may have left work enqueued)
    struct Release { RerankCtx *c; ~Release() { (void)hipStreamSynchronize(c->stream); rerank_ctx_release(c); } } release{c};
    hipStream_t stream = c->stream;

    // pinned staging: [query | valid rows compacted in arrival order (null / wrong-length rows dropped, exec.rs:496-498,
    // 526-528), a Float64 column narrowed `as f32` on the way (exec.rs:542) | distances back]
    const size_t q_bytes = static_cast<size_t>(dim) * sizeof(float), c_bytes = static_cast<size_t>(mv) * dim * sizeof(float),
                 d_bytes = static_cast<size_t>(mv) * sizeof(float);
    HIP_TRY(c->h_stage.ensure(q_bytes + c_bytes + d_bytes + 64));
    uint8_t *hs = c->h_stage.as<uint8_t>();
    float *h_q = reinterpret_cast<float *>(hs), *h_c = reinterpret_cast<float *>(hs + q_bytes),
          *h_d = reinterpret_cast<float *>(hs + q_bytes + c_bytes);
    std::memcpy(h_q, query, q_bytes);
    {
        uint64_t o = 0;
        for (uint64_t i = 0; i < m; ++i) {
            if (valid && !valid[i]) continue;
            const T *src = cand + i * dim;
            float *dst = h_c + o * dim;
            if constexpr (std::is_same<T, float>::value) std::memcpy(dst, src, static_cast<size_t>(dim) * sizeof(float));
            else for (uint32_t j = 0; j < dim; ++j) dst[j] = static_cast<float>(src[j]);
            ++o;
        }
    }
    HIP_TRY(c->d_q.ensure(q_bytes)); HIP_TRY(c->d_cand.ensure(c_bytes)); HIP_TRY(c->d_mdist.ensure(d_bytes));
    HIP_TRY(c->d_base.ensure(sizeof(uint64_t))); HIP_TRY(c->d_probe0.ensure(sizeof(uint32_t))); HIP_TRY(c->d_off.ensure(2 * sizeof(uint64_t)));
    if (c->off_m != mv || c->base_k != 0) {       // the one-list descriptor of the stream kernel
        const uint64_t h_base = 0, h_off[2] = {0, mv};
        const uint32_t h_probe0 = 0;
        c->off_m = ~0ull; c->base_k = ~0ull;      // nothing cached until all three copies have landed
        HIP_TRY(hipMemcpyAsync(c->d_base.p, &h_base, sizeof h_base, hipMemcpyHostToDevice, stream));
        HIP_TRY(hipMemcpyAsync(c->d_probe0.p, &h_probe0, sizeof h_probe0, hipMemcpyHostToDevice, stream));
        HIP_TRY(hipMemcpyAsync(c->d_off.p, h_off, sizeof h_off, hipMemcpyHostToDevice, stream));
        HIP_TRY(hipStreamSynchronize(stream));        // the sources are stack variables
        c->off_m = mv; c->base_k = 0;
    }
    HIP_TRY(hipMemcpyAsync(c->d_q.p, h_q, q_bytes, hipMemcpyHostToDevice, stream));
    HIP_TRY(hipMemcpyAsync(c->d_cand.p, h_c, c_bytes, hipMemcpyHostToDevice, stream));
    pqv::StreamArgs ra{};
    ra.mat = c->d_cand.as<float>(); ra.row_of = nullptr; ra.list_off = c->d_off.as<uint64_t>();
    ra.probe = c->d_probe0.as<uint32_t>(); ra.cand_base = c->d_base.as<uint64_t>();
    ra.queries = c->d_q.as<float>(); ra.nq = 1; ra.nprobe = 1; ra.dim = dim; ra.k = 1;
    ra.rows_per_block = 256; ra.blocks_per_list = static_cast<uint32_t>((mv + 255) / 256);
    ra.max_pos = ~0ull; ra.metric = metric; ra.out_f32 = c->d_mdist.as<float>();
    HIP_TRY(pqv::launch_stream(ra, pqv::STREAM_DIST, stream));
    HIP_TRY(hipMemcpyAsync(h_d, c->d_mdist.p, d_bytes, hipMemcpyDeviceToHost, stream));
    HIP_TRY(hipStreamSynchronize(stream));

    std::vector<HeapEnt> heap;
    heap.reserve(static_cast<size_t>(std::min<uint64_t>(k, static_cast<uint64_t>(*io_count) + mv)) + 1);     // sized by the data, not by a huge "keep everything" k
    for (uint32_t i = 0; i < *io_count; ++i) heap.push_back(HeapEnt{io_d2[i], io_rows[i]});     // the array IS the heap
    uint64_t o = 0;
    for (uint64_t i = 0; i < m; ++i) {
        if (valid && !valid[i]) continue;
        const HeapEnt ent{h_d[o++], ids ? ids[i] : static_cast<uint32_t>(i)};
        if (heap.size() < k) heap_push(heap, ent);                                     // exec.rs:474-475
        else if (ent.d < heap[0].d) { heap_pop(heap); heap_push(heap, ent); }          // :476-481
    }
    for (size_t i = 0; i < heap.size(); ++i) { io_rows[i] = heap[i].row; io_d2[i] = heap[i].d; }
    *io_count = static_cast<uint32_t>(heap.size());
    return PQV_OK;
}
extern "C" int pqv_rerank(int device, const float *query, const float *cand, const uint32_t *ids,
                          const uint8_t *valid, uint64_t m, uint32_t dim, uint32_t k, int metric,
                          uint32_t *io_rows, float *io_d2, uint32_t *io_count) {
    return guard([&] { return rerank_host<float>(device, query, cand, ids, valid, m, dim, k, metric, io_rows, io_d2, io_count); });
}
extern "C" int pqv_rerank_f64(int device, const float *query, const double *cand, const uint32_t *ids,
                              const uint8_t *valid, uint64_t m, uint32_t dim, uint32_t k, int metric,
                              uint32_t *io_rows, float *io_d2, uint32_t *io_count) {
    return guard([&] { return rerank_host<double>(device, query, cand, ids, valid, m, dim, k, metric, io_rows, io_d2, io_count); });
}
extern "C" int pqv_rerank_finish(const uint32_t *io_rows, const float *io_d2, uint32_t count, uint32_t *out_rows, float *out_d2) {
    return guard([&] {
        if (count && (!io_rows || !io_d2 || !out_rows || !out_d2)) return fail(PQV_ERR_INVALID, "state/output arrays must not be NULL");
        std::vector<HeapEnt> v(count);
        for (uint32_t i = 0; i < count; ++i) v[i] = HeapEnt{io_d2[i], io_rows[i]};     // heap.into_iter(): backing-array order
        std::stable_sort(v.begin(), v.end(), [](const HeapEnt &a, const HeapEnt &b) { return a.d < b.d; });   // exec.rs:270-274
        for (uint32_t i = 0; i < count; ++i) { out_rows[i] = v[i].row; out_d2[i] = v[i].d; }
        return static_cast<int>(PQV_OK);
    });
}

#ifdef PQV_STAMPS
// diagnostic build only (make stamps): device wall-clock stamps of the one-query launch sequence
extern "C" int pqv_debug_stamps(unsigned long long *out, int reset) {
    return pqv::stamps_io(out, reset) == hipSuccess ? 0 : -1;
}
#endif
