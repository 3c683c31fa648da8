// exchange.cpp -- the one exchange step of the sharded search behind the C ABI (include/pqv.h, pqv_shard_*).
//
// The reference merges a multi-file table in ONE heap (src/df_vector/index_exec.rs:85-164 probes every file's own
// index, src/df_vector/exec.rs:264-267 folds all their batches).  Here a file / row-group range is a shard on its
// own GPU and that heap becomes: one RCCL all-gather of every rank's k packed {distance, row} results per query over
// xGMI + the deterministic merge kernel (distance, shard, position) on every rank.  The host of the reference is
// Rust, so the collective must not need torch: RCCL is bound here through its C API.
//
// librccl is resolved at run time (dlopen), never at link time: a single-GPU user of libpqv_hip.so does not need it,
// and a process that already carries a copy (PyTorch-ROCm ships one with the same SONAME librccl.so.1) must keep
// using THAT copy -- a communicator is only valid inside the library that created it.
#include "../../include/pqv.h"

#include <dlfcn.h>

#include <cstdlib>
#include <cstring>
#include <mutex>
#include <string>

#include "internal.h"
#include "kernels.h"

using pqv_internal::fail;
using pqv_internal::guard;
using pqv_internal::use_device;

namespace {

// the slice of rccl.h this file needs (rccl.h:40-43,187,220,260,339,459-462,678)
struct NcclUniqueId { char internal[PQV_SHARD_ID_BYTES]; };
typedef void *NcclComm;
typedef int NcclResult;                       // ncclSuccess == 0
constexpr int kNcclUint8 = 1;

struct Rccl {
    void *handle = nullptr;
    std::string path, error;
    NcclResult (*GetUniqueId)(NcclUniqueId *) = nullptr;
    NcclResult (*CommInitRank)(NcclComm *, int, NcclUniqueId, int) = nullptr;
    NcclResult (*CommDestroy)(NcclComm) = nullptr;
    NcclResult (*CommCount)(const NcclComm, int *) = nullptr;
    NcclResult (*CommUserRank)(const NcclComm, int *) = nullptr;
    NcclResult (*AllGather)(const void *, void *, size_t, int, NcclComm, hipStream_t) = nullptr;
    const char *(*GetErrorString)(NcclResult) = nullptr;
};

Rccl *rccl() {
    static Rccl r;
    static std::once_flag once;
    std::call_once(once, [] {
        // 1. PQV_RCCL_LIB names the library; 2. a copy this process already loaded; 3. the loader path; 4. /opt/rocm
        const char *env = std::getenv("PQV_RCCL_LIB");
        const char *names[] = {"librccl.so.1", "librccl.so"};
        if (env && *env) r.handle = dlopen(env, RTLD_NOW | RTLD_GLOBAL);
        for (const char *n : names) if (!r.handle) r.handle = dlopen(n, RTLD_NOW | RTLD_NOLOAD);
        for (const char *n : names) if (!r.handle) r.handle = dlopen(n, RTLD_NOW | RTLD_GLOBAL);
        if (!r.handle) r.handle = dlopen("/opt/rocm/lib/librccl.so.1", RTLD_NOW | RTLD_GLOBAL);
        if (!r.handle) { const char *e = dlerror(); r.error = std::string("librccl not found: ") + (e ? e : "dlopen failed"); return; }
        auto sym = [&](const char *name) {
            void *p = dlsym(r.handle, name);
            if (!p && r.error.empty()) r.error = std::string("librccl lacks ") + name;
            return p;
        };
        r.GetUniqueId = reinterpret_cast<decltype(r.GetUniqueId)>(sym("ncclGetUniqueId"));
        r.CommInitRank = reinterpret_cast<decltype(r.CommInitRank)>(sym("ncclCommInitRank"));
        r.CommDestroy = reinterpret_cast<decltype(r.CommDestroy)>(sym("ncclCommDestroy"));
        r.CommCount = reinterpret_cast<decltype(r.CommCount)>(sym("ncclCommCount"));
        r.CommUserRank = reinterpret_cast<decltype(r.CommUserRank)>(sym("ncclCommUserRank"));
        r.AllGather = reinterpret_cast<decltype(r.AllGather)>(sym("ncclAllGather"));
        r.GetErrorString = reinterpret_cast<decltype(r.GetErrorString)>(sym("ncclGetErrorString"));
        Dl_info info;
        if (r.GetUniqueId && dladdr(reinterpret_cast<void *>(r.GetUniqueId), &info) && info.dli_fname) r.path = info.dli_fname;
    });
    return &r;
}

int rccl_ready(Rccl **out) {
    Rccl *r = rccl();
    if (!r->error.empty()) return fail(PQV_ERR_UNSUPPORTED, r->error);
    *out = r;
    return PQV_OK;
}

int nccl_fail(Rccl *r, const char *what, NcclResult e) {
    return fail(PQV_ERR_HIP, std::string(what) + ": " + (r->GetErrorString ? r->GetErrorString(e) : "RCCL error") +
                                 " (" + std::to_string(e) + ")");
}

#define HIP_TRY(expr)                                                                                            \
    do {                                                                                                         \
        hipError_t _e = (expr);                                                                                  \
        if (_e != hipSuccess) {                                                                                  \
            (void)hipGetLastError();                                                                             \
            return fail(_e == hipErrorOutOfMemory ? PQV_ERR_OOM : PQV_ERR_HIP, std::string(#expr) + ": " + hipGetErrorString(_e)); \
        }                                                                                                        \
    } while (0)

}  // namespace

struct pqv_shard_comm {
    int device = 0;
    uint32_t rank = 0, world = 1;
    NcclComm comm = nullptr;
    bool owned = false;
    // send [nq, k] and receive [world, nq, k] buffers of packed pairs, grown on demand; calls on one communicator are
    // serialised by the caller (as RCCL requires of a communicator anyway)
    void *d_send = nullptr, *d_gath = nullptr;
    size_t send_bytes = 0, gath_bytes = 0;
};

extern "C" const char *pqv_shard_rccl_path(void) {
    Rccl *r = rccl();
    return r->error.empty() ? r->path.c_str() : "";
}

extern "C" int pqv_shard_unique_id(uint8_t *id) {
    return guard([&] {
        if (!id) return fail(PQV_ERR_INVALID, "id must not be NULL");
        Rccl *r = nullptr;
        if (int rc = rccl_ready(&r)) return rc;
        NcclUniqueId u;
        if (NcclResult e = r->GetUniqueId(&u)) return nccl_fail(r, "ncclGetUniqueId", e);
        std::memcpy(id, u.internal, PQV_SHARD_ID_BYTES);
        return static_cast<int>(PQV_OK);
    });
}

extern "C" int pqv_shard_comm_create(int device, uint32_t rank, uint32_t world, const uint8_t *id, pqv_shard_comm **out) {
    return guard([&] {
        if (!out) return fail(PQV_ERR_INVALID, "out must not be NULL");
        *out = nullptr;
        if (!id) return fail(PQV_ERR_INVALID, "id must not be NULL");
        if (world == 0 || rank >= world) return fail(PQV_ERR_INVALID, "rank must be < world");
        Rccl *r = nullptr;
        if (int rc = rccl_ready(&r)) return rc;
        if (int rc = use_device(device)) return rc;
        NcclUniqueId u;
        std::memcpy(u.internal, id, PQV_SHARD_ID_BYTES);
        pqv_shard_comm *c = new pqv_shard_comm();
        c->device = device; c->rank = rank; c->world = world; c->owned = true;
        if (NcclResult e = r->CommInitRank(&c->comm, static_cast<int>(world), u, static_cast<int>(rank))) {
            delete c;
            return nccl_fail(r, "ncclCommInitRank", e);
        }
        *out = c;
        return static_cast<int>(PQV_OK);
    });
}

extern "C" int pqv_shard_comm_adopt(int device, void *nccl_comm, pqv_shard_comm **out) {
    return guard([&] {
        if (!out) return fail(PQV_ERR_INVALID, "out must not be NULL");
        *out = nullptr;
        if (!nccl_comm) return fail(PQV_ERR_INVALID, "nccl_comm must not be NULL");
        Rccl *r = nullptr;
        if (int rc = rccl_ready(&r)) return rc;
        if (int rc = use_device(device)) return rc;
        int n = 0, me = 0;
        if (NcclResult e = r->CommCount(nccl_comm, &n)) return nccl_fail(r, "ncclCommCount", e);
        if (NcclResult e = r->CommUserRank(nccl_comm, &me)) return nccl_fail(r, "ncclCommUserRank", e);
        pqv_shard_comm *c = new pqv_shard_comm();
        c->device = device; c->rank = static_cast<uint32_t>(me); c->world = static_cast<uint32_t>(n);
        c->comm = nccl_comm; c->owned = false;
        *out = c;
        return static_cast<int>(PQV_OK);
    });
}

extern "C" uint32_t pqv_shard_comm_rank(const pqv_shard_comm *c) { return c ? c->rank : 0; }
extern "C" uint32_t pqv_shard_comm_world(const pqv_shard_comm *c) { return c ? c->world : 0; }

extern "C" void pqv_shard_comm_free(pqv_shard_comm *c) {
    if (!c) return;
    (void)hipSetDevice(c->device);
    if (c->d_send) (void)hipFree(c->d_send);
    if (c->d_gath) (void)hipFree(c->d_gath);
    if (c->owned && c->comm) { Rccl *r = rccl(); if (r->CommDestroy) (void)r->CommDestroy(c->comm); }
    delete c;
}

extern "C" int pqv_shard_exchange(pqv_shard_comm *c, const void *d_dist, const void *d_rows, const void *d_row_base,
                                  uint32_t nq, uint32_t k, void *d_out_dist, void *d_out_rows, void *hip_stream) {
    return guard([&] {
        if (!c) return fail(PQV_ERR_INVALID, "comm must not be NULL");
        if (!d_dist || !d_rows || !d_row_base || !d_out_dist || !d_out_rows)
            return fail(PQV_ERR_INVALID, "device pointers must not be NULL");
        if (k == 0) return fail(PQV_ERR_INVALID, "k must be > 0");
        if (k > 1024) return fail(PQV_ERR_UNSUPPORTED, "k > 1024 is not supported");
        if (nq == 0) return static_cast<int>(PQV_OK);
        Rccl *r = nullptr;
        if (int rc = rccl_ready(&r)) return rc;
        if (int rc = use_device(c->device)) return rc;
        hipStream_t stream = static_cast<hipStream_t>(hip_stream);
        const size_t per_rank = static_cast<size_t>(nq) * k * 8;
        if (per_rank > c->send_bytes) {
            HIP_TRY(hipStreamSynchronize(stream));      // the old buffers may still be in flight on this stream
            if (c->d_send) (void)hipFree(c->d_send);
            if (c->d_gath) (void)hipFree(c->d_gath);
            c->d_send = c->d_gath = nullptr; c->send_bytes = c->gath_bytes = 0;
            HIP_TRY(hipMalloc(&c->d_send, per_rank));
            HIP_TRY(hipMalloc(&c->d_gath, per_rank * c->world));
            c->send_bytes = per_rank; c->gath_bytes = per_rank * c->world;
        }
        HIP_TRY(pqv::launch_pack_pairs(static_cast<const float *>(d_dist), static_cast<const uint32_t *>(d_rows),
                                       static_cast<uint64_t>(nq) * k, c->d_send, stream));
        // ONE collective per step: rank r's [nq, k] pairs land at gath[r]
        if (NcclResult e = r->AllGather(c->d_send, c->d_gath, per_rank, kNcclUint8, c->comm, stream))
            return nccl_fail(r, "ncclAllGather", e);
        const float *base = static_cast<const float *>(c->d_gath);
        HIP_TRY(pqv::launch_shard_merge(base, reinterpret_cast<const uint32_t *>(base) + 1,
                                        static_cast<const long long *>(d_row_base), c->world, nq, k,
                                        static_cast<float *>(d_out_dist), static_cast<long long *>(d_out_rows), stream, 2));
        return static_cast<int>(PQV_OK);
    });
}

// One Parquet file shared by `world` GPUs: the row-group range of shard `rank` (host only; sharding.shard_row_groups and
// bindings/rust/src/file.rs call this).  Cut r is the row-group boundary whose prefix sum of rows is nearest to r n / world
// (the lower one on a tie, never before the previous cut), compared exactly in integers: |pre[b] world - r n|.
extern "C" int pqv_shard_row_groups(const uint64_t *rg_rows, uint32_t n_row_groups, uint32_t rank, uint32_t world,
                                    uint32_t *rg_lo, uint32_t *rg_hi, uint64_t *row_base, uint64_t *n_rows) {
    return guard([&]() -> int {
        if (world == 0 || rank >= world) return fail(PQV_ERR_INVALID, "rank must be < world, world > 0");
        if (n_row_groups && !rg_rows) return fail(PQV_ERR_INVALID, "rg_rows must not be NULL");
        if (!rg_lo || !rg_hi || !row_base || !n_rows) return fail(PQV_ERR_INVALID, "output pointers must not be NULL");
        std::vector<unsigned __int128> pre(static_cast<size_t>(n_row_groups) + 1, 0);
        for (uint32_t i = 0; i < n_row_groups; ++i) pre[i + 1] = pre[i] + rg_rows[i];
        const unsigned __int128 n = pre[n_row_groups];
        if (n > static_cast<unsigned __int128>(0xFFFFFFFFFFFFFFFFull)) return fail(PQV_ERR_INVALID, "row count exceeds 64 bits");
        uint32_t cut_prev = 0, lo = 0, hi = n_row_groups;
        for (uint32_t r = 1; r <= rank + 1 && r < world; ++r) {
            const unsigned __int128 target = static_cast<unsigned __int128>(r) * n;          // x world
            uint32_t best = cut_prev;
            unsigned __int128 best_d = ~static_cast<unsigned __int128>(0);
            for (uint32_t b = cut_prev; b <= n_row_groups; ++b) {
                const unsigned __int128 v = pre[b] * world, d = v > target ? v - target : target - v;
                if (d < best_d) { best_d = d; best = b; }
            }
            if (r == rank) lo = best;
            if (r == rank + 1) hi = best;
            cut_prev = best;
        }
        if (rank == 0) lo = 0;
        if (rank + 1 == world) hi = n_row_groups;
        *rg_lo = lo; *rg_hi = hi;
        *row_base = static_cast<uint64_t>(pre[lo]); *n_rows = static_cast<uint64_t>(pre[hi] - pre[lo]);
        return static_cast<int>(PQV_OK);
    });
}
