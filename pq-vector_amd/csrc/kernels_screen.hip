// kernels_screen.hip -- gfx950 kernels of the batched candidate re-rank (src/ivf/search.rs:112-127 for a whole batch):
// tile_rerank_kernel (exact, rows shared by 16 queries), tile_filter_kernel, and the MFMA-screened path: wide_seed_kernel,
// seed_select_kernel, wide_filter_kernel (int8 / f16 / f32 operand images, exact evaluation of the survivors in index.rs:461-480 order).
#include "device_common.hpp"

namespace pqv {

// ------------------------------------------------------------------------------------
// tile_rerank_kernel: the batched candidate re-rank.
//
// grid = (blocks_per_list, max_groups); block = 4 independent waves.  A block takes one
// group (<= QB queries that all probe cluster c) and one row chunk of c's inverted list;
// each wave walks its rows lane-per-row in 64-row tiles.  Per tile the lane's row is
// loaded 128 B at a time (a full cache line per lane) and every query of the group is
// applied to it from SGPRs (wave-uniform scalar loads) -- each streamed row is used up to QB
// times.  Every (row, query) chain is the reference's serial
//   sum += ((d0^2 + d1^2) + d2^2) + d3^2   in ascending group order (index.rs:461-480).
//
// Top-k: a candidate is admitted iff its key beats min(this wave's k-th key, the query's
// GLOBAL threshold).  The global threshold is the minimum over all waves of their k-th
// keys (device-scope atomic min): each is an upper bound of the final k-th key, so nothing
// that belongs to the final top-k is ever rejected, and the merged result is independent
// of timing.  It collapses the work of the ~nprobe*blocks*4 independent lists per query to
// roughly one list's worth of inserts.  Per-query state is held lane-parallel (lane q of a
// wave holds query q's row index / candidate base / thresholds / local k-th key) and the
// epilogue is a rolled loop over the group's queries reading the tile's sums back from LDS,
// so the fold code exists once.  LDS holds only the running sums (QB x 64 floats per wave).
// ------------------------------------------------------------------------------------
template <int QB, int S, bool ALIGNED>
__global__ __launch_bounds__(256) void tile_rerank_kernel(const TileArgs a) {
    uint32_t bx, gi;
    xcd_remap(bx, gi, a.xcd_swizzle);
    if (gi >= *a.n_groups) return;
    const uint4 grp = a.groups[gi];
    const uint32_t c = grp.x, p0 = grp.y, cnt = grp.z;
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const uint32_t k = a.k;

    __shared__ float lsums_all[4 * QB * 64];
    float *lsums = lsums_all + wave * (QB * 64);

    const uint64_t lbeg = a.list_off[c], lend = a.list_off[c + 1];
    const uint64_t len = lend - lbeg;
    const uint64_t wrows = a.rows_per_block / 4;
    uint64_t r0 = a.row_offset + (uint64_t)bx * a.rows_per_block + (uint64_t)wave * wrows;
    uint64_t r1 = r0 + wrows;
    if (r1 > len) r1 = len;
    if (a.row_end && r1 > a.row_end) r1 = a.row_end;     // window of this launch
    if (r0 > len) r0 = len;

    const uint32_t dim = a.dim;
    const uint32_t G = dim >> 2, tail = dim & 3u;

    // lane-parallel per-query state: lane q (< QB) owns query q of the group
    const uint32_t my_slot = p0 + ((uint32_t)lane < cnt ? (uint32_t)lane : cnt - 1);
    const uint32_t my_pair = a.pairs[my_slot];
    const uint32_t my_qrow = my_pair / a.nprobe;
    const uint64_t my_cbase = a.cand_base[my_pair];
    uint64_t my_lkth = KEY_EMPTY;          // k-th key of this wave's list of query `lane`
    bool my_touched = false;               // this wave has folded into its list of query `lane`
    // this wave's list of query `lane`: slot (q, j, chunk, wave) of the partial-list buffer
    const uint32_t n_part = a.n_part;
    const uint64_t my_base =
        ((uint64_t)my_qrow * n_part + (my_pair % a.nprobe) * a.slots_per_pair + a.slot_base + bx * 4 + wave) * k;

    // (the partial-list buffer was preset to EMPTY by the caller: one memset instead of k-entry stores per
    //  wave and query)

    for (uint64_t t0 = r0; t0 < r1; t0 += 64) {
        const uint32_t nvalid = (r1 - t0 < 64) ? (uint32_t)(r1 - t0) : 64u;
        const uint32_t lrow = (uint32_t)lane < nvalid ? (uint32_t)lane : nvalid - 1;
        const uint64_t lpos = lbeg + t0 + lrow;
        const uint32_t srow = a.row_of ? a.row_of[lpos] : (uint32_t)lpos;
        const float *x = a.mat + (uint64_t)srow * dim;
        // this tile's view of the global thresholds (relaxed device-scope load; a stale
        // value is only a looser bound).  Issued now, consumed after the distance loop.
        const uint64_t my_gthr =
            __hip_atomic_load(a.gthr + my_qrow, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);

        // Distance loop.  The running sums of the group's queries live in LDS
        // (lsums[query][lane]); per 128-B step of the lane's row, a ROLLED loop over exactly
        // the group's `cnt` queries applies each query chunk (wave-uniform, SGPRs) and does a
        // read-modify-write of that query's sum.  No padded work for partial groups, a small
        // loop body, few VGPRs => enough resident waves to hide the load latencies.
        uint32_t g0 = 0;
        for (; g0 + 8 <= G; g0 += 8) {
            float4 xv[8];
#pragma unroll
            for (int g = 0; g < 8; ++g) xv[g] = load4<ALIGNED>(x + (g0 + g) * 4);
            // Scalar (SMEM) loads return out of order, so the only usable wait is
            // lgkmcnt(0).  Software pipeline, two chunk register sets ping-ponging:
            //   wait(chunk of query q) -> issue loads of query q+1 -> math on query q
            // keeps one 128-B chunk in flight behind ~200 cycles of VALU; sched_barrier
            // pins that order (otherwise hipcc issues every load right before its use).
            float4 qa[8], qb[8];
            {
                const float *qp = a.queries + (uint64_t)readlane_u32(my_qrow, 0) * dim + g0 * 4;
#pragma unroll
                for (int g = 0; g < 8; ++g) qa[g] = load4_uniform<ALIGNED>(qp + g * 4);
            }
            uint32_t qq = 0;
#pragma unroll 1
            for (; qq + 2 <= cnt; qq += 2) {
                float acc0 = g0 ? lsums[qq * 64 + lane] : 0.0f;
                float acc1 = g0 ? lsums[(qq + 1) * 64 + lane] : 0.0f;
                __builtin_amdgcn_s_waitcnt(0xC07F);   // lgkmcnt(0): qa (and the sums) landed
                {
                    const float *qp = a.queries + (uint64_t)readlane_u32(my_qrow, (int)qq + 1) * dim + g0 * 4;
#pragma unroll
                    for (int g = 0; g < 8; ++g) qb[g] = load4_uniform<ALIGNED>(qp + g * 4);
                }
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int g = 0; g < 8; ++g) {
                    const float d0 = qa[g].x - xv[g].x, d1 = qa[g].y - xv[g].y;
                    const float d2 = qa[g].z - xv[g].z, d3 = qa[g].w - xv[g].w;
                    float t = d0 * d0 + d1 * d1;
                    t = t + d2 * d2;
                    t = t + d3 * d3;
                    acc0 = acc0 + t;
                }
                lsums[qq * 64 + lane] = acc0;
                __builtin_amdgcn_sched_barrier(0);
                __builtin_amdgcn_s_waitcnt(0xC07F);   // qb landed
                {
                    const uint32_t nq2 = qq + 2 < cnt ? qq + 2 : cnt - 1;
                    const float *qp = a.queries + (uint64_t)readlane_u32(my_qrow, (int)nq2) * dim + g0 * 4;
#pragma unroll
                    for (int g = 0; g < 8; ++g) qa[g] = load4_uniform<ALIGNED>(qp + g * 4);
                }
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int g = 0; g < 8; ++g) {
                    const float d0 = qb[g].x - xv[g].x, d1 = qb[g].y - xv[g].y;
                    const float d2 = qb[g].z - xv[g].z, d3 = qb[g].w - xv[g].w;
                    float t = d0 * d0 + d1 * d1;
                    t = t + d2 * d2;
                    t = t + d3 * d3;
                    acc1 = acc1 + t;
                }
                lsums[(qq + 1) * 64 + lane] = acc1;
                __builtin_amdgcn_sched_barrier(0);
            }
            if (qq < cnt) {   // odd count: qa holds the last query's chunk
                float acc = g0 ? lsums[qq * 64 + lane] : 0.0f;
#pragma unroll
                for (int g = 0; g < 8; ++g) {
                    const float d0 = qa[g].x - xv[g].x, d1 = qa[g].y - xv[g].y;
                    const float d2 = qa[g].z - xv[g].z, d3 = qa[g].w - xv[g].w;
                    float t = d0 * d0 + d1 * d1;
                    t = t + d2 * d2;
                    t = t + d3 * d3;
                    acc = acc + t;
                }
                lsums[qq * 64 + lane] = acc;
            }
        }
        for (; g0 < G; ++g0) {
            const float4 xg = load4<ALIGNED>(x + g0 * 4);
#pragma unroll 1
            for (uint32_t qq = 0; qq < cnt; ++qq) {
                const float4 qv = load4_uniform<ALIGNED>(a.queries + (uint64_t)readlane_u32(my_qrow, (int)qq) * dim + g0 * 4);
                const float d0 = qv.x - xg.x, d1 = qv.y - xg.y;
                const float d2 = qv.z - xg.z, d3 = qv.w - xg.w;
                float t = d0 * d0 + d1 * d1;
                t = t + d2 * d2;
                t = t + d3 * d3;
                const float acc = g0 ? lsums[qq * 64 + lane] : 0.0f;
                lsums[qq * 64 + lane] = acc + t;
            }
        }
        for (uint32_t e = 0; e < tail; ++e) {
            const float xe = x[G * 4 + e];
#pragma unroll 1
            for (uint32_t qq = 0; qq < cnt; ++qq) {
                const float d = load1_uniform(a.queries + (uint64_t)readlane_u32(my_qrow, (int)qq) * dim + G * 4 + e) - xe;
                const float acc = (G || e) ? lsums[qq * 64 + lane] : 0.0f;
                lsums[qq * 64 + lane] = acc + d * d;
            }
        }

        // ---- top-k epilogue: rolled over the group's queries --------------------------
        wave_lds_fence();
        const uint64_t my_thr = my_lkth < my_gthr ? my_lkth : my_gthr;
        const uint64_t posl = t0 + (uint64_t)lane;
#pragma unroll 1
        for (uint32_t qq = 0; qq < cnt; ++qq) {
            const uint64_t thr = readlane_u64(my_thr, (int)qq);
            const uint64_t pos = readlane_u64(my_cbase, (int)qq) + posl;
            const bool valid = (uint32_t)lane < nvalid && pos < a.max_pos;
            const float sv = lsums[qq * 64 + lane];
            const uint64_t mykey =
                valid ? (((uint64_t)__float_as_uint(sv) << 32) | (uint64_t)(uint32_t)pos) : KEY_EMPTY;
            if (__ballot(mykey < thr) != 0ull) {
                const uint64_t base = readlane_u64(my_base, (int)qq);
                const uint64_t nk = tile_fold<S>(a.part_keys + base, a.part_vals + base,
                                                 a.gthr + readlane_u32(my_qrow, (int)qq),
                                                 readlane_u64(my_gthr, (int)qq), readlane_u64(my_lkth, (int)qq),
                                                 mykey, srow, k, lane,
                                                 ((__ballot(my_touched) >> qq) & 1ull) == 0ull);
                if ((uint32_t)lane == qq) { my_lkth = nk; my_touched = true; }
            }
        }
        wave_lds_fence();
    }
}


// ------------------------------------------------------------------------------------
// tile_filter_kernel: the batched re-rank with an MFMA lower-bound screen.
//
// Same work decomposition as tile_rerank_kernel (group of <= 16 queries x row chunk of one
// list; a wave walks 64-row tiles).  Per tile the 16 x 64 score block s = q.x is computed on
// the matrix cores (v_mfma_f32_16x16x4_f32: four 16 x 16 tiles, exact f32 products), then
//     d~ = |q|^2 + |x|^2 - 2 s,     lb = d~ - c (2 (|q|^2 + |x|^2) + |d~|),  c = (dim + 16) 2^-22
// lb is a rigorous lower bound of the reference's d2: both d2 (index.rs:461-480 order) and d~
// approximate the real sum within first-order bounds (dim/4 + 5) u D and (dim + 4) u (|q| + |x|)^2,
// u = 2^-24, and c carries a 4x safety factor.  A pair is skipped iff lb > the query's
// threshold distance -- then its exact key cannot beat the threshold key.  Survivors (a
// fraction of a percent once thresholds are seeded) are queued per wave and evaluated 64 at a
// time, lane-per-pair, in the reference's exact summation order, then folded exactly like in
// tile_rerank_kernel.  Results are therefore identical to the unscreened kernel.
// The k-order of the MFMA contraction is permuted (lane kk owns 4 consecutive dims of each
// 16-dim step) so that every operand fetch is one 16-byte load; the bound is order-free.
// ------------------------------------------------------------------------------------

template <int S, bool ALIGNED, bool PREFETCH>
__global__ __launch_bounds__(256) void tile_filter_kernel(const TileArgs a) {
    static_assert(TILE_QB == 16, "the 16x16x4 MFMA tile fixes the group size");
    constexpr int PEND = 1024 + 64;        // pending (query, row) pairs per wave: a whole tile fits
    uint32_t bx, gi;
    xcd_remap(bx, gi, a.xcd_swizzle);
    if (gi >= *a.n_groups) return;
    const uint4 grp = a.groups[gi];
    const uint32_t c = grp.x, p0 = grp.y, cnt = grp.z;
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const uint32_t k = a.k;

    __shared__ uint32_t pend_all[4 * PEND];     // (query slot << 28) | row-in-list offset from r0
    uint32_t *pend = pend_all + wave * PEND;

    const uint64_t lbeg = a.list_off[c], lend = a.list_off[c + 1];
    const uint64_t len = lend - lbeg;
    const uint64_t wrows = a.rows_per_block / 4;
    uint64_t r0 = a.row_offset + (uint64_t)bx * a.rows_per_block + (uint64_t)wave * wrows;
    uint64_t r1 = r0 + wrows;
    if (r1 > len) r1 = len;
    if (a.row_end && r1 > a.row_end) r1 = a.row_end;     // window of this launch
    if (r0 > len) r0 = len;

    const uint32_t dim = a.dim;
    const uint32_t G = dim >> 2, tail = dim & 3u;
    const float cmargin = (float)(dim + 16) * 2.384185791015625e-07f;   // (dim + 16) * 2^-22

    // lane-parallel per-query state: lane q (< 16) owns query q of the group
    const uint32_t my_slot = p0 + ((uint32_t)lane < cnt ? (uint32_t)lane : cnt - 1);
    const uint32_t my_pair = a.pairs[my_slot];
    const uint32_t my_qrow = my_pair / a.nprobe;
    const uint64_t my_cbase = a.cand_base[my_pair];
    const float my_qn = a.query_norm2[my_qrow];
    uint64_t my_lkth = KEY_EMPTY;
    const uint32_t n_part = a.n_part;
    const uint64_t my_base =
        ((uint64_t)my_qrow * n_part + (my_pair % a.nprobe) * a.slots_per_pair + a.slot_base + bx * 4 + wave) * k;

    // MFMA operand roles of this lane: query / row index inside a 16-tile, and its k slice
    const int l15 = lane & 15, kk = lane >> 4;
    const float *qrow_ptr = a.queries + (uint64_t)__shfl((int)my_qrow, l15, 64) * dim;
    float qn4[4];      // |q|^2 of the 4 queries whose scores this lane receives: i = kk*4 + r
#pragma unroll
    for (int r = 0; r < 4; ++r) qn4[r] = __shfl(my_qn, kk * 4 + r, 64);

    uint32_t npend = 0;
    uint32_t n_exact = 0;

    // exact evaluation of queued pairs [start, start + count), count <= 64, lane-per-pair in the
    // reference's summation order, then one fold per query of the group
    auto eval = [&](uint32_t start, uint32_t count) {
        wave_lds_fence();
        const bool have = (uint32_t)lane < count;
        const uint32_t pe = pend[start + (have ? lane : 0)];
        const uint32_t qs = pe >> 28;                       // query slot in the group
        const uint64_t roff = r0 + (pe & 0x0FFFFFFFu);       // row offset in the list
        const uint64_t lpos = lbeg + roff;
        const uint32_t srow = a.row_of ? a.row_of[lpos] : (uint32_t)lpos;
        const float *x = a.mat + (uint64_t)srow * dim;
        const float *q = a.queries + (uint64_t)__shfl((int)my_qrow, (int)qs, 64) * dim;
        float sum = 0.0f;
        uint32_t g = 0;
        for (; g + 2 <= G; g += 2) {       // 4 loads in flight per lane, then the ordered chain
            float4 xv[2], qv[2];
#pragma unroll
            for (int u = 0; u < 2; ++u) { xv[u] = load4<ALIGNED>(x + (g + u) * 4); qv[u] = load4<ALIGNED>(q + (g + u) * 4); }
#pragma unroll
            for (int u = 0; u < 2; ++u) {
                const float d0 = qv[u].x - xv[u].x, d1 = qv[u].y - xv[u].y;
                const float d2 = qv[u].z - xv[u].z, d3 = qv[u].w - xv[u].w;
                float t = d0 * d0 + d1 * d1;
                t = t + d2 * d2;
                t = t + d3 * d3;
                sum = sum + t;
            }
        }
        for (; g < G; ++g) {
            const float4 xv = load4<ALIGNED>(x + g * 4), qv = load4<ALIGNED>(q + g * 4);
            const float d0 = qv.x - xv.x, d1 = qv.y - xv.y, d2 = qv.z - xv.z, d3 = qv.w - xv.w;
            float t = d0 * d0 + d1 * d1;
            t = t + d2 * d2;
            t = t + d3 * d3;
            sum = sum + t;
        }
        for (uint32_t e = 0; e < tail; ++e) {
            const float d = q[G * 4 + e] - x[G * 4 + e];
            sum = sum + d * d;
        }
        const uint64_t my_gthr =
            __hip_atomic_load(a.gthr + my_qrow, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        const uint64_t my_thr = my_lkth < my_gthr ? my_lkth : my_gthr;
#pragma unroll 1
        for (uint32_t qq = 0; qq < cnt; ++qq) {
            const uint64_t pos = readlane_u64(my_cbase, (int)qq) + roff;
            const bool mine = have && qs == qq && pos < a.max_pos;
            const uint64_t mykey =
                mine ? (((uint64_t)__float_as_uint(sum) << 32) | (uint64_t)(uint32_t)pos) : KEY_EMPTY;
            const uint64_t thr = readlane_u64(my_thr, (int)qq);
            if (__ballot(mykey < thr) != 0ull) {
                const uint64_t base = readlane_u64(my_base, (int)qq);
                const uint64_t nk = tile_fold<S>(a.part_keys + base, a.part_vals + base,
                                                 a.gthr + readlane_u32(my_qrow, (int)qq),
                                                 readlane_u64(my_gthr, (int)qq), readlane_u64(my_lkth, (int)qq),
                                                 mykey, srow, k, lane);
                if ((uint32_t)lane == qq) my_lkth = nk;
            }
        }
    };
    // evaluate tail batches until fewer than `keep_below` entries remain
    auto drain = [&](uint32_t keep_below) {
        while (npend >= keep_below && npend > 0) {
            const uint32_t take = npend < 64 ? npend : 64;
            eval(npend - take, take);
            n_exact += take;
            npend -= take;
        }
        wave_lds_fence();
    };

    const bool fast_k = ALIGNED && (dim & 15u) == 0;       // wave-uniform
    for (uint64_t t0 = r0;; t0 += 64) {
        const bool last = t0 >= r1;
        if (!last) {
            const uint32_t nvalid = (r1 - t0 < 64) ? (uint32_t)(r1 - t0) : 64u;
            // rows of the four 16-row tiles this lane feeds (B operand) / receives (C columns)
            const float *xrow[4];
            float xn[4];
#pragma unroll
            for (int t = 0; t < 4; ++t) {
                uint32_t rr = (uint32_t)(16 * t + l15);
                if (rr >= nvalid) rr = nvalid - 1;
                const uint64_t lpos = lbeg + t0 + rr;
                const uint32_t srow = a.row_of ? a.row_of[lpos] : (uint32_t)lpos;
                xrow[t] = a.mat + (uint64_t)srow * dim;
                xn[t] = a.row_norm2[a.norm_by_pos ? lpos : (uint64_t)srow];
            }
            // thresholds of this tile (a stale value is only a looser bound)
            const uint64_t my_gthr =
                __hip_atomic_load(a.gthr + my_qrow, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            const uint64_t my_thr = my_lkth < my_gthr ? my_lkth : my_gthr;
            // threshold DISTANCE (upper 32 bits of the key) of the lane's 4 queries; KEY_EMPTY
            // gives the NaN pattern 0xFFFFFFFF, which compares false below: "cannot skip"
            const float my_thr_d = __uint_as_float((uint32_t)(my_thr >> 32));
            float thr4[4];
#pragma unroll
            for (int r = 0; r < 4; ++r) thr4[r] = __shfl(my_thr_d, kk * 4 + r, 64);

            f32x4_acc acc[4];
#pragma unroll
            for (int t = 0; t < 4; ++t) acc[t] = (f32x4_acc){0.f, 0.f, 0.f, 0.f};

            if (fast_k) {
                if constexpr (PREFETCH) {
                    // long rows (many K steps per tile): operands of step k0 + 16 are fetched
                    // behind the 16 MFMAs of step k0 (costs ~20 VGPRs = one wave of occupancy)
                    float4 qc = load4<true>(qrow_ptr + 4 * kk), qnx;
                    float4 xc[4], xnx[4];
#pragma unroll
                    for (int t = 0; t < 4; ++t) xc[t] = load4<true>(xrow[t] + 4 * kk);
                    for (uint32_t k0 = 0; k0 < dim; k0 += 16) {
                        const bool more = k0 + 16 < dim;
                        if (more) {
                            qnx = load4<true>(qrow_ptr + k0 + 16 + 4 * kk);
#pragma unroll
                            for (int t = 0; t < 4; ++t) xnx[t] = load4<true>(xrow[t] + k0 + 16 + 4 * kk);
                        }
#pragma unroll
                        for (int t = 0; t < 4; ++t) {
                            acc[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(qc.x, xc[t].x, acc[t], 0, 0, 0);
                            acc[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(qc.y, xc[t].y, acc[t], 0, 0, 0);
                            acc[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(qc.z, xc[t].z, acc[t], 0, 0, 0);
                            acc[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(qc.w, xc[t].w, acc[t], 0, 0, 0);
                        }
                        if (more) {
                            qc = qnx;
#pragma unroll
                            for (int t = 0; t < 4; ++t) xc[t] = xnx[t];
                        }
                    }
                } else {
                    // short rows: no explicit prefetch, the registers are worth more as occupancy
                    for (uint32_t k0 = 0; k0 < dim; k0 += 16) {
                        const float4 qc = load4<true>(qrow_ptr + k0 + 4 * kk);
                        float4 xc[4];
#pragma unroll
                        for (int t = 0; t < 4; ++t) xc[t] = load4<true>(xrow[t] + k0 + 4 * kk);
#pragma unroll
                        for (int t = 0; t < 4; ++t) {
                            acc[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(qc.x, xc[t].x, acc[t], 0, 0, 0);
                            acc[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(qc.y, xc[t].y, acc[t], 0, 0, 0);
                            acc[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(qc.z, xc[t].z, acc[t], 0, 0, 0);
                            acc[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(qc.w, xc[t].w, acc[t], 0, 0, 0);
                        }
                    }
                }
            } else {
                for (uint32_t k0 = 0; k0 < dim; k0 += 16) {
                    const uint32_t kb = k0 + 4 * kk;      // this lane's 4 dims of the step
                    float qf[4] = {0.f, 0.f, 0.f, 0.f}, xf[4][4];
#pragma unroll
                    for (int t = 0; t < 4; ++t)
#pragma unroll
                        for (int e = 0; e < 4; ++e) xf[t][e] = 0.f;
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        if (kb + e < dim) {
                            qf[e] = qrow_ptr[kb + e];
#pragma unroll
                            for (int t = 0; t < 4; ++t) xf[t][e] = xrow[t][kb + e];
                        }
                    }
#pragma unroll
                    for (int t = 0; t < 4; ++t)
#pragma unroll
                        for (int e = 0; e < 4; ++e)
                            acc[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(qf[e], xf[t][e], acc[t], 0, 0, 0);
                }
            }

            // screen: C/D layout col j = lane & 15 (row 16 t + j of the tile), row i = kk * 4 + r (query)
#pragma unroll
            for (int t = 0; t < 4; ++t) {
                const bool jvalid = (uint32_t)(16 * t + l15) < nvalid;
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const uint32_t qi = (uint32_t)(kk * 4 + r);
                    const float nn = qn4[r] + xn[t];
                    const float dt = nn - 2.0f * acc[t][r];
                    const float lb = dt - cmargin * (2.0f * nn + fabsf(dt));
                    const bool skip = lb > thr4[r];            // false when the threshold is EMPTY (NaN)
                    const bool keep = jvalid && qi < cnt && !skip;
                    const unsigned long long m = __ballot(keep);
                    if (m) {
                        const uint32_t before = (uint32_t)__popcll(m & ((1ull << lane) - 1ull));
                        if (keep) pend[npend + before] = (qi << 28) | (uint32_t)(t0 - r0 + 16 * t + l15);
                        npend += (uint32_t)__popcll(m);
                    }
                }
            }
        }
        // evaluate full batches (everything at the end), keep a partial batch queued otherwise
        drain(last ? 1u : 64u);
        if (last) break;
    }
    if (a.stats && lane == 0) {      // counter pairs spread over STATS_SLOTS lines, see wide_filter_kernel
#ifdef PQV_PROFILE_PHASES
        unsigned long long *st = a.stats;
#else
        unsigned long long *st = a.stats + 8 + 16 * ((blockIdx.y * gridDim.x + blockIdx.x + (uint32_t)wave * 17u) % STATS_SLOTS);
#endif
        atomicAdd(&st[0], (unsigned long long)(r1 - r0) * cnt);
        atomicAdd(&st[1], (unsigned long long)n_exact);
    }
}

// ------------------------------------------------------------------------------------
// Per-query candidate buffers of the wide screened path: cand_keys/vals [nq][cap], cand_cnt[nq] -- reset by
// seed_select_kernel, appended to by wide_filter_kernel (one atomic per verified pair), folded by merge_kernel.
// ------------------------------------------------------------------------------------
// ------------------------------------------------------------------------------------
// wide_seed_kernel<NG>: admission thresholds for the wide screened pass WITHOUT an exact pass.
//
// For the first rows of every probed list the 16 NG x 64 score blocks are computed exactly like in
// wide_filter_kernel; d~ + c (2 nn + |d~|) is then a rigorous UPPER bound of the reference's d2 (the
// mirror image of the screen's lower bound).  Every lane keeps the minimum upper bound of the rows
// it sees for each of its queries: lanes (and waves, lists) see DISJOINT rows, so the k-th smallest
// of a query's minima (seed_select_kernel) is the upper bound of k distinct candidates' distances
// -- a valid admission threshold, within the margin of the k-th smallest exact distance of the
// sample.  The sample rows themselves are screened and evaluated by the main pass like all others.
// Output: seed_ub[((qrow * nprobe + j) * seed_sw + blockIdx.x * 4 + wave) * 16 + (lane & 15)].
// ------------------------------------------------------------------------------------
template <int S>
__device__ __forceinline__ void seed_select_body(const uint32_t q, const float *seed_ub, uint32_t n_vals, uint32_t k,
                                                 unsigned long long *gthr, uint32_t *cand_cnt, uint32_t *spilled,
                                                 uint32_t *thr_hist, float4 *thr_bins, const SeedRefine &rf,
                                                 float *lds_terms = nullptr, uint32_t lds_floats = 0);     // (defined below)
// one-query calls (SeedTail): every block of wide_seed_kernel takes a ticket when it is done -- also the ones with nothing
// to sample -- and the last one runs the select / refinement for query 0
__device__ __forceinline__ void seed_tail_finish(const TileArgs &a) {
    __shared__ uint32_t s_seed_last;
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // the bounds went out as agent-scope atomic stores (see probe_single_kernel)
    __syncthreads();
    if (threadIdx.x == 0) {
        const uint32_t t = atomicAdd(a.seed_tail.ticket, 1u);
        s_seed_last = t == gridDim.x * gridDim.y * gridDim.z - 1u ? 1u : 0u;
        if (s_seed_last) *a.seed_tail.ticket = 0u;
    }
    __syncthreads();
    if (!s_seed_last) return;
    PQV_STAMP_MAX(12);
    extern __shared__ float4 qs_tail[];       // the staged queries are no longer needed: the refinement's term table
    if (a.seed_tail.k <= 64u)
        seed_select_body<1>(0u, a.seed_ub, a.seed_tail.n_vals, a.seed_tail.k, a.seed_tail.gthr, a.seed_tail.cand_cnt, a.seed_tail.spilled,
                            a.seed_tail.thr_hist, a.seed_tail.thr_bins, a.seed_tail.rf, reinterpret_cast<float *>(qs_tail), a.seed_tail.lds_floats);
    else        // k <= 256: the k-th bound by the block's radix select (one query at K = 100: a launch and its gap less)
        seed_select_body<4>(0u, a.seed_ub, a.seed_tail.n_vals, a.seed_tail.k, a.seed_tail.gthr, a.seed_tail.cand_cnt, a.seed_tail.spilled,
                            a.seed_tail.thr_hist, a.seed_tail.thr_bins, a.seed_tail.rf, reinterpret_cast<float *>(qs_tail), a.seed_tail.lds_floats);
}
// U: operand stages a wave keeps in flight.  1 for batches (other waves fill the stalls); a one-query call has ONE 64-row
// tile per wave and nothing else on the CU, so its 48 KB are requested 12 stages at a time.  Only the U > 1 instances
// carry the one-query tail (select + refinement by the last block): its register needs (16 row chunks + 16 query chunks
// in flight per lane) would otherwise set the allocation -- and halve the occupancy -- of the batched instances.
template <int NG, bool QLDS, int OP, int U = 1>
__global__ __launch_bounds__(256) void wide_seed_kernel(const TileArgs a) {
    constexpr bool F16 = OP == OP_F16, I8 = OP == OP_I8;
    static_assert(!I8 || QLDS, "int8 operands: queries staged in LDS");
    constexpr uint32_t NQ = 16 * NG;
    // One-query instance (U > 1; launched only for nq == 1, so a quad holds ONE query): a block takes one 64-row tile and
    // each of its waves ONE 16-row sub-tile of it (TS = 1) -- four times the blocks, because a CU takes in ~25-40 GB/s
    // and the sample's 12 MB are cold; the waves' bounds are combined through LDS, so seed_ub looks exactly as when one
    // wave walks the whole tile.  Only group 0 exists (NGE = 1).
    constexpr bool ONE = U > 1;
    constexpr int TS = ONE ? 1 : 4, NGE = ONE ? 1 : NG;
    PQV_STAMP_MIN(8);
    uint32_t bx, by;
    quad_xcd_remap(bx, by, a.xcd_swizzle, *a.n_quads);
    // (one pass of a loop, so that a block with nothing to sample leaves through `break` and the one-query tail below is
    //  instantiated ONCE: inlined at three exits it made this kernel 143 KB of code, more than the instruction cache)
    for (int once = 0; once < 1; ++once) {
    if (by >= *a.n_quads) break;
    const uint4 quad = a.quads[by];
    // a quad wider than this kernel's 16 NG queries (the 8-wave filter kernel takes up to 128) is sampled in
    // slices of 16 NG: blockIdx.z
    const uint32_t sub = blockIdx.z * NQ;
    if (sub >= quad.z) break;
    const uint32_t c = quad.x, p0 = quad.y + sub, cnt = quad.z - sub < NQ ? quad.z - sub : NQ;
    const uint32_t ng = ONE ? 1u : (cnt + 15) >> 4;
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    // (requested before the list bounds are waited for: one round trip for both)
    const uint32_t my_slot = p0 + ((uint32_t)lane < cnt ? (uint32_t)lane : cnt - 1);
    const uint32_t my_pair = a.pairs[my_slot];

    extern __shared__ float4 qs[];
    __shared__ __attribute__((aligned(16))) float qn_all[4 * 64];
    __shared__ __attribute__((aligned(16))) uint32_t lim_all[4 * 64];
    float *qnl = qn_all + wave * 64;
    uint32_t *liml = lim_all + wave * 64;

    const uint64_t lbeg = a.list_off[c], lend = a.list_off[c + 1];
    const uint64_t len = lend - lbeg;
    const uint64_t wrows = ONE ? 64 : a.rows_per_block / 4;
    uint64_t r0 = ONE ? a.row_offset + (uint64_t)bx * 64 : a.row_offset + (uint64_t)bx * a.rows_per_block + (uint64_t)wave * wrows;
    uint64_t r1 = r0 + wrows;
    if (r1 > len) r1 = len;
    if (a.row_end && r1 > a.row_end) r1 = a.row_end;
    if (r0 > r1) r0 = r1;

    const uint32_t dim = a.dim;
    const uint32_t G = I8 ? dim >> 4 : F16 ? dim >> 3 : dim >> 2;    // 16-byte operand columns per row
    const float cmargin = (float)(dim + 16) * 2.384185791015625e-07f;   // (dim + 16) * 2^-22
    const float c16 = F16 ? 1.25f * 9.765625e-04f : 0.0f;               // f16 operands: see wide_filter_kernel
    const float isc2 = F16 ? 1.0f / a.scale2 : 1.0f;                    // scores are contracted at scale^2

    const int l15 = lane & 15, kk = lane >> 4;
    const uint64_t blk0 = a.blk_off[c], blk_last = a.blk_off[c + 1] - 1;
    const uint32_t lane_off = (uint32_t)kk * 16 + (uint32_t)l15;
    // one-query instance: the wave's operand stages are requested NOW, next to the chain pair -> bases -> query image that
    // follows (every step of either chain is a cold round trip, and nothing else runs on the CU to hide it)
    [[maybe_unused]] float4 xpre[ONE ? U : 1];
    if constexpr (ONE) {
        if (r1 > r0) {
            uint64_t T = blk0 + ((r0 + 16u * (uint32_t)wave) >> 4);
            if (T > blk_last) T = blk_last;
            const __amdgpu_buffer_rsrc_t xr0 = operand_rsrc(a.mat_blk + T * G * 16);
            const uint32_t nks0 = G >> 2;
#pragma unroll
            for (int u = 0; u < U; ++u) xpre[u] = buf_ld16(xr0, lane_off * 16u, ((uint32_t)u < nks0 ? (uint32_t)u : nks0 - 1) * 1024);
        }
    }
    const uint32_t my_qrow = my_pair / a.nprobe;
    const uint64_t my_cbase = a.cand_base[my_pair];
    // list offsets below my_lim are candidates of this query (max_candidates cap)
    const uint64_t room = a.max_pos > my_cbase ? a.max_pos - my_cbase : 0;
    qnl[lane] = a.query_norm2[my_qrow];
    const float my_qn0 = a.query_norm2[my_qrow];
    bool my_bad16 = F16 && (a.query_maxabs[my_qrow] * a.scale > 32768.0f || my_qn0 * a.scale2 < 1.0f);   // no valid f16 bound
    [[maybe_unused]] const uint32_t my_img = a.i8_pair_images ? my_pair : my_qrow;     // int8: image per pair or per query
    if constexpr (I8) my_bad16 = !(a.q_resu[my_img] <= 3.0e38f);         // non-finite query: no bound
    liml[lane] = ((uint32_t)lane < cnt && !my_bad16) ? (room > 0xFFFFFFFFull ? 0xFFFFFFFFu : (uint32_t)room) : 0u;
    if constexpr (QLDS) {
        constexpr uint32_t TPQ = 256 / NQ;
        const uint32_t q = threadIdx.x / TPQ, c0 = threadIdx.x % TPQ;
        const float4 *src = reinterpret_cast<const float4 *>(a.queries + (uint64_t)__shfl((int)my_qrow, (int)q, 64) * dim);
        float4 *dst = qs + q * G;
        const uint32_t sw = q & 15u;
        if constexpr (I8) {        // the image of the (query, this list) PAIR: the residual against the list's centre
            const float4 *s8 = reinterpret_cast<const float4 *>(a.q_i8 + (uint64_t)__shfl((int)my_img, (int)q, 64) * dim);
            if constexpr (U > 1) {
#pragma unroll 16
                for (uint32_t ch = c0; ch < G; ch += TPQ) dst[ch ^ sw] = s8[ch];
            } else {
#pragma unroll PQV_STAGE_UNROLL
                for (uint32_t ch = c0; ch < G; ch += TPQ) dst[ch ^ sw] = s8[ch];
            }
        } else if constexpr (F16) {
#pragma unroll 4
            for (uint32_t ch = c0; ch < G; ch += TPQ) dst[ch ^ sw] = pack_f16x8(src[2 * ch], src[2 * ch + 1], a.scale);
        } else {
#pragma unroll 8
            for (uint32_t ch = c0; ch < G; ch += TPQ) dst[ch ^ sw] = src[ch];
        }
    }
    __syncthreads();      // also orders the qnl / liml writes above
    PQV_STAMP_MAX(9);
    const float4 *qblk = a.q_blk + (uint64_t)by * NG * G * 16;
    const __amdgpu_buffer_rsrc_t qr = operand_rsrc(QLDS ? (const void *)a.queries : (const void *)qblk);

    float mins[NGE][4];
    // I8: the largest dot - ceil(Nx / 2) a lane sees per query bounds the smallest |qi - xi|^2 from above
    [[maybe_unused]] int maxs[NGE][4];
    [[maybe_unused]] float rmax = 0.0f;          // largest residual bound among the rows this lane saw
#pragma unroll
    for (int g = 0; g < NGE; ++g)
#pragma unroll
        for (int r = 0; r < 4; ++r) { mins[g][r] = INFINITY; maxs[g][r] = -(1 << 30); }

    for (uint64_t t0 = r0; t0 < r1; t0 += 64) {
        const uint32_t nvalid = (r1 - t0 < 64) ? (uint32_t)(r1 - t0) : 64u;
        const float4 *xbase[TS];
        float xn[TS];
        [[maybe_unused]] int xn2i[TS];
        const int tb = ONE ? wave : 0;               // first 16-row sub-tile of this wave
#pragma unroll
        for (int tt = 0; tt < TS; ++tt) {
            const int t = tb + tt;
            uint32_t rr = (uint32_t)(16 * t + l15);
            if (rr >= nvalid) rr = nvalid - 1;
            if constexpr (I8) {
                xn[tt] = 0.0f;
                xn2i[tt] = a.row_n2i[lbeg + t0 + rr];
                if ((uint32_t)(16 * t + l15) < nvalid) rmax = fmaxf(rmax, a.row_res[lbeg + t0 + rr]);
            } else
            xn[tt] = a.row_norm2[lbeg + t0 + rr];
            uint64_t T = blk0 + ((t0 + 16 * t) >> 4);
            if (T > blk_last) T = blk_last;
            xbase[tt] = a.mat_blk + T * G * 16;
        }
        const __amdgpu_buffer_rsrc_t xr = operand_rsrc(xbase[0]);
        uint32_t xso[TS];
#pragma unroll
        for (int t = 0; t < TS; ++t) xso[t] = (uint32_t)((xbase[t] - xbase[0]) * 16);
        using acc_t = std::conditional_t<I8, i32x4_acc, f32x4_acc>;
        acc_t acc[NGE][TS];
#pragma unroll
        for (int g = 0; g < NGE; ++g)
#pragma unroll
            for (int t = 0; t < TS; ++t) {
                if constexpr (I8) { const int init = -((xn2i[t] + 1) >> 1); acc[g][t] = (i32x4_acc){init, init, init, init}; }
                else acc[g][t] = (f32x4_acc){0.f, 0.f, 0.f, 0.f};
            }
        const uint32_t nks = G >> 2;
        for (uint32_t ks0 = 0; ks0 < nks; ks0 += U) {
            float4 x[U][TS];
            if (ONE && ks0 == 0) {          // (one branch around the whole stage set, not a select per load)
#pragma unroll
                for (int u = 0; u < U; ++u) x[u][0] = xpre[ONE ? u : 0];
            } else {
#pragma unroll
                for (int u = 0; u < U; ++u) {
                    const uint32_t ks = ks0 + u < nks ? ks0 + u : nks - 1;
#pragma unroll
                    for (int t = 0; t < TS; ++t) x[u][t] = buf_ld16(xr, lane_off * 16u, xso[t] + ks * 1024);
                }
            }
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const uint32_t ks = ks0 + u;
                if (U > 1 && ks >= nks) break;
                const uint32_t chq = ks * 4 + (uint32_t)kk;
#pragma unroll
                for (int g = 0; g < NGE; ++g) {
                    if ((uint32_t)g < ng) {
                        float4 qc;
                        if constexpr (QLDS) qc = qs[(16 * g + l15) * G + (chq ^ (uint32_t)l15)];
                        else qc = buf_ld16(qr, lane_off * 16u, (uint32_t)g * G * 256 + ks * 1024);
#pragma unroll
                        for (int t = 0; t < TS; ++t) mfma_step<OP>(acc[g][t], qc, x[u][t]);
                    }
                }
            }
        }
#pragma unroll
        for (int g = 0; g < NGE; ++g) {
            const float4 q4 = *reinterpret_cast<const float4 *>(qnl + 16 * g + 4 * kk);
            const uint4 l4 = *reinterpret_cast<const uint4 *>(liml + 16 * g + 4 * kk);
            const float qn[4] = {q4.x, q4.y, q4.z, q4.w};
            const uint32_t lim[4] = {l4.x, l4.y, l4.z, l4.w};
#pragma unroll
            for (int r = 0; r < 4; ++r) {
#pragma unroll
                for (int tt = 0; tt < TS; ++tt) {
                    const int t = tb + tt;
                    const uint32_t roff = (uint32_t)t0 + (uint32_t)(16 * t + l15);     // list offset (< 2^32 rows per list)
                    const bool valid = (uint32_t)(16 * t + l15) < nvalid && roff < lim[r];
                    if constexpr (I8) {
                        if (valid) maxs[g][r] = max(maxs[g][r], acc[g][tt][r]);
                    } else {
                        const float nn = qn[r] + xn[tt];
                        const float dt = nn - 2.0f * (acc[g][tt][r] * isc2);
                        const float ub = dt + cmargin * (2.0f * nn + fabsf(dt)) + c16 * nn;
                        if (valid) mins[g][r] = fminf(mins[g][r], ub);                     // NaN bounds are ignored
                    }
                }
            }
        }
    }
    PQV_STAMP_MAX(10);
    if constexpr (ONE) {
        // the four sub-tiles' bounds: wave 0 takes the best of each lane position (int8: the largest dot and the largest
        // residual bound, which enter the bound together below)
        __shared__ float s_red[4][5][64];
        float *mine = &s_red[wave][0][lane];
#pragma unroll
        for (int r = 0; r < 4; ++r) mine[r * 64] = I8 ? __int_as_float(maxs[0][r]) : mins[0][r];
        mine[4 * 64] = rmax;
        __syncthreads();
        if (wave == 0)
#pragma unroll
        for (int w = 1; w < 4; ++w) {
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const float o = s_red[w][r][lane];
                if constexpr (I8) maxs[0][r] = max(maxs[0][r], __float_as_int(o));
                else mins[0][r] = fminf(mins[0][r], o);
            }
            rmax = fmaxf(rmax, s_red[w][4][lane]);
        }
    }
    // publish: one value per (query, this wave, lane & 15)
    const uint32_t my_j = my_pair % a.nprobe;
    if (!ONE || wave == 0)
#pragma unroll
    for (int g = 0; g < NGE; ++g) {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const uint32_t qi = (uint32_t)(16 * g + kk * 4 + r);
            const uint32_t qrow = (uint32_t)__shfl((int)my_qrow, (int)qi, 64);
            const uint32_t j = (uint32_t)__shfl((int)my_j, (int)qi, 64);
            if constexpr (I8) {
                // |q - x| <= |vi - xi| / S + rq' + rx (rq' includes what the clamp cut off the query residual),
                // |vi - xi|^2 = Nq + Nx - 2 dot <= Nq - 2 (dot - ceil(Nx / 2)); the reference's computed d2 exceeds the real
                // one by at most the summation margin
                const uint32_t pr = (uint32_t)__shfl((int)my_img, (int)qi, 64);
                if (qi < cnt && maxs[g][r] > -(1 << 30)) {
                    const float n_ub = fmaxf((float)(a.q_n2i[pr] - 2 * maxs[g][r]) * 1.000001f + 2.0f, 0.0f);
                    const float d = sqrtf(n_ub) * 1.000001f / a.list_scale[c] + a.q_resu[pr] + rmax;
                    mins[g][r] = d * d * (1.0f + 4.0f * cmargin) * 1.000002f;
                }
            }
            if (qi < cnt) {
                float *dst = a.seed_ub + (((uint64_t)qrow * a.nprobe + j) * a.seed_sw + (ONE ? bx : bx * 4 + wave)) * 16 + l15;
                if constexpr (U > 1) __hip_atomic_store(dst, fmaxf(mins[g][r], 0.0f), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                else *dst = fmaxf(mins[g][r], 0.0f);
            }
        }
    }
    PQV_STAMP_MAX(11);
    }
    if constexpr (U > 1) { if (a.seed_tail.enable) seed_tail_finish(a); }
}

// gthr[q] = key of the k-th smallest of q's n_vals upper bounds (none if fewer than k are finite); also
// resets the query's candidate buffer and overflow flag.  One wave per query.
template <int S>
__device__ __forceinline__ void seed_select_body(const uint32_t q, const float *seed_ub, uint32_t n_vals, uint32_t k,
                                                 unsigned long long *gthr, uint32_t *cand_cnt, uint32_t *spilled,
                                                 uint32_t *thr_hist, float4 *thr_bins, const SeedRefine &rf,
                                                 float *lds_terms, uint32_t lds_floats) {
    // one wave selects; with the refinement (256 threads) all four waves share the exact evaluations
    __shared__ uint64_t s_ent[16];           // the k selected bounds
    __shared__ uint64_t s_exact[64];         // exact keys of their 4 k rows
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const bool refine = rf.mat && k <= 16u && blockDim.x == 256;
    WaveTopk<S> tk;
    tk.init();
    // with the refinement the four waves select from a quarter of the bounds each and wave 0 merges the four lists
    __shared__ uint64_t s_loc[64];
    const uint32_t quarter = ((n_vals + 3) / 4 + 63) / 64 * 64;
    const uint32_t v_lo = refine ? (uint32_t)wave * quarter : 0u;
    const uint32_t v_hi = refine ? (v_lo + quarter < n_vals ? v_lo + quarter : n_vals) : n_vals;
    // (the one-query tail -- the only caller with LDS for the terms -- reads bounds this very launch published: agent-scope loads)
    const bool same_launch = lds_terms != nullptr;
    // the refinement needs first row, end and candidate base of the probed lists: fetched now (two dependent round trips that
    // hide behind the selection) instead of after it
    __shared__ uint64_t s_lbeg[64], s_lend[64], s_cbase[64];
    const bool pre_lists = refine && same_launch && rf.nprobe <= 64u;
    if (pre_lists && wave == 3 && (uint32_t)lane < rf.nprobe) {
        const uint32_t c = rf.probe[(uint64_t)q * rf.nprobe + lane];
        s_lbeg[lane] = rf.list_off[c];
        s_lend[lane] = rf.list_off[c + 1];
        s_cbase[lane] = rf.cand_base[(uint64_t)q * rf.nprobe + lane];
    }
    auto ld_ub = [&](uint32_t idx) {
        const float *p = seed_ub + (uint64_t)q * n_vals + idx;
        return same_launch ? __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : *p;
    };
    // k > 64 (no refinement): only the k-th smallest bound and the smallest one are needed, not the sorted list -- a radix
    // select by the whole block instead of ~3 k serial insertions into a wave-distributed list (K = 100 on 1024 bounds:
    // 139 -> 12 us per launch)
    uint64_t sel_kth = KEY_EMPTY, sel_m1 = KEY_EMPTY;
    bool sel_done = false;
    if constexpr (S > 1) {
        if (!refine && blockDim.x >= 64) {
            __shared__ uint32_t s_rh[256];
            __shared__ uint32_t s_rs[4];
            if (threadIdx.x == 0) { s_rs[2] = 0u; s_rs[3] = 0xFFFFFFFFu; }
            __syncthreads();
            uint32_t nfin = 0, mn = 0xFFFFFFFFu;
            for (uint32_t i = threadIdx.x; i < n_vals; i += blockDim.x) {
                const float v = ld_ub(i);                        // published as fmaxf(bound, 0): bits order like integers
                if (v < INFINITY) { ++nfin; mn = min(mn, __float_as_uint(v)); }
            }
            if (nfin) { atomicAdd(&s_rs[2], nfin); atomicMin(&s_rs[3], mn); }
            __syncthreads();
            const uint32_t tot = s_rs[2], mnb = s_rs[3];
            if (tot >= k) {
                const uint32_t kb = block_kth_u32([&](uint32_t i) { const float v = ld_ub(i); return v < INFINITY ? __float_as_uint(v) : 0xFFFFFFFFu; },
                                                  n_vals, k, s_rh, s_rs);
                sel_kth = ((uint64_t)kb << 32) | 0xFFFFFFFFull;
                sel_m1 = (uint64_t)mnb << 32;
            }
            sel_done = true;
            if (wave != 0) return;
        }
    }
    if ((wave == 0 || refine) && !sel_done) {
        // pre-filter (k <= 64): the k-th smallest of the 64 lane minima bounds the k-th smallest overall, so only
        // values at or below it are offered to the serial insertion (a few dozen instead of all n_vals)
        uint64_t cut = KEY_EMPTY;
        bool done = false;
        if constexpr (S == 1) {
            if (k <= 64u && v_hi <= v_lo + 1024u) {
                // the wave's bounds fit 16 per lane: one round trip, kept in registers for both passes, and the few that pass the
                // cut are sorted instead of inserted one by one
                uint64_t kreg[16];
#pragma unroll
                for (int u = 0; u < 16; ++u) {
                    const uint32_t idx = v_lo + 64 * u + lane;
                    const float v = idx < v_hi ? ld_ub(idx) : INFINITY;
                    kreg[u] = v < INFINITY ? (((uint64_t)__float_as_uint(v) << 32) | idx) : KEY_EMPTY;
                }
                uint64_t lmin = KEY_EMPTY;
#pragma unroll
                for (int u = 0; u < 16; ++u) lmin = kreg[u] < lmin ? kreg[u] : lmin;
                __shared__ uint64_t s_sel[4 * 128];
                cut = wave_kth_by_rank(lmin, k, lane, s_sel + (wave & 3) * 128);
                uint64_t sorted = KEY_EMPTY;
                if (wave < 4 && wave_select_by_sort<16>(kreg, cut, lane, s_sel + wave * 128, sorted)) {
                    tk.key[0] = (uint32_t)lane < k ? sorted : KEY_EMPTY;
                } else {
#pragma unroll
                    for (int u = 0; u < 16; ++u) {
                        uint64_t key = kreg[u];
                        if (key > cut) key = KEY_EMPTY;
                        if (__ballot(key != KEY_EMPTY) != 0ull) tk.offer(key, 0u, k, lane);
                    }
                }
                done = true;
            }
        }
        if (!done) {
        if (k <= 64u) {
            uint64_t lmin = KEY_EMPTY;
            for (uint32_t i0 = v_lo; i0 < v_hi; i0 += 1024) {         // sixteen loads in flight per lane
                float v[16];
#pragma unroll
                for (int u = 0; u < 16; ++u) {
                    const uint32_t idx = i0 + 64 * u + lane;
                    v[u] = idx < v_hi ? ld_ub(idx) : INFINITY;
                }
#pragma unroll
                for (int u = 0; u < 16; ++u) {
                    if (v[u] < INFINITY) {
                        const uint64_t key = ((uint64_t)__float_as_uint(v[u]) << 32) | (i0 + 64 * u + lane);
                        lmin = key < lmin ? key : lmin;
                    }
                }
            }
            uint32_t dummy = 0;
            bitonic_sort64(lmin, dummy, lane);
            cut = readlane_u64(lmin, (int)k - 1);
        }
        for (uint32_t i0 = v_lo; i0 < v_hi; i0 += 1024) {             // sixteen loads in flight per lane again
            float v[16];
#pragma unroll
            for (int u = 0; u < 16; ++u) {
                const uint32_t idx = i0 + 64 * u + lane;
                v[u] = idx < v_hi ? ld_ub(idx) : INFINITY;
            }
#pragma unroll
            for (int u = 0; u < 16; ++u) {
                uint64_t key = KEY_EMPTY;
                if (v[u] < INFINITY) key = ((uint64_t)__float_as_uint(v[u]) << 32) | (i0 + 64 * u + lane);
                if (key > cut) key = KEY_EMPTY;
                if (__ballot(key != KEY_EMPTY) != 0ull) tk.offer(key, 0u, k, lane);
            }
        }
        }
    }
    if (refine) {           // merge: 4 x (k <= 16) sorted keys -> one 64-lane sort in wave 0; lanes 0 .. k-1 then hold the k smallest
        if (lane < 16) s_loc[wave * 16 + lane] = (uint32_t)lane < k ? tk.key[0] : KEY_EMPTY;
        __syncthreads();
        if (wave == 0) {
            __shared__ uint64_t s_mrg[128];
            const uint64_t mk = wave_sort_by_rank(s_loc[lane], 64u, lane, s_mrg);
            tk.key[0] = (uint32_t)lane < k ? mk : KEY_EMPTY;
        }
    }
    uint64_t kth = sel_done ? sel_kth : tk.kth(k);
    uint64_t m1key = sel_done ? sel_m1 : readlane_u64(tk.key[0], 0);
    PQV_STAMP_MAX(13);
    if (refine) {
        // exact distances of the 4 k rows behind the k selected bounds (SeedRefine): wave w takes entries
        // [w k / 4 ..) -- pair p = 4 e + t is entry e's sub-tile row t -- and L = 4 or 8 lanes share a pair's chain exactly
        // as in wide_filter_kernel's evaluation (the reference's order, bit for bit)
        if (wave == 0 && lane < 16) s_ent[lane] = (uint32_t)lane < k ? tk.key[0] : KEY_EMPTY;
        __syncthreads();
        const uint32_t Gx = rf.dim >> 2;
        const uint32_t NP = 4u * k + 1u <= 64u ? 4u * k + 1u : 64u;        // row stride of the term table (odd: no bank conflicts)
        if (lds_terms && (uint64_t)Gx * NP <= lds_floats) {
            // The tail of a one-query call runs alone on the chip: every dependent round trip costs its full latency, and the
            // 4 k rows sit on 4 k cold pages.  All 256 threads fetch the rows' 16-byte chunks at once (coalesced along a row),
            // leave the per-chunk terms ((d0^2 + d1^2) + d2^2) + d3^2 in LDS, and lane p of wave 0 then adds row p's terms in
            // the reference's order -- the same bits as the lane chains below, in one round trip instead of three.
            __shared__ uint64_t s_rowoff[64];
            const uint32_t np = 4u * k;                                    // <= 64
            bool valid = false;
            uint64_t pos = 0;
            if (wave == 0) {
                const uint32_t pi = (uint32_t)lane;
                const uint32_t e = pi >> 2, t = pi & 3u;
                const uint64_t ekey = s_ent[e < 16u ? e : 0u];
                valid = pi < np && ekey != KEY_EMPTY;
                const uint32_t idx = (uint32_t)ekey;
                const uint32_t l15 = idx & 15u, slot = (idx >> 4) % rf.seed_sw, j = (idx >> 4) / rf.seed_sw;
                const uint32_t row = (slot >> 2) * 256u + (slot & 3u) * 64u + 16u * t + l15;
                uint64_t lbeg = 0;
                if (valid && pre_lists) {
                    lbeg = s_lbeg[j];
                    pos = s_cbase[j] + row;
                    valid = row < rf.seed_rows && lbeg + row < s_lend[j] && pos < rf.max_pos;
                } else if (valid) {
                    const uint32_t c = rf.probe[(uint64_t)q * rf.nprobe + j];
                    lbeg = rf.list_off[c];
                    pos = rf.cand_base[(uint64_t)q * rf.nprobe + j] + row;
                    valid = row < rf.seed_rows && lbeg + row < rf.list_off[c + 1] && pos < rf.max_pos;
                }
                s_rowoff[lane] = valid ? (uint64_t)(rf.row_of ? rf.row_of[lbeg + row] : lbeg + row) * rf.dim : 0ull;
            }
            __syncthreads();
            const float4 *qg4 = reinterpret_cast<const float4 *>(rf.queries + (uint64_t)q * rf.dim);
            if (Gx <= 256u) {
                // thread (rg, g): chunk g of rows rg, rg + RG, ... -- one query chunk per thread, up to 40 row chunks in flight
                // (k = 10 on a 768-dim row: every byte of the 40 rows is requested in ONE round trip)
                const uint32_t RG = 256u / Gx, rg = threadIdx.x / Gx, g = threadIdx.x - rg * Gx;
                if (rg < RG) {
                    const float4 qc = qg4[g];
                    constexpr int B = 40;
                    for (uint32_t p0 = rg; p0 < np; p0 += RG * B) {
                        float4 xv[B];
#pragma unroll
                        for (int u = 0; u < B; ++u) {
                            const uint32_t pr2 = p0 + RG * (uint32_t)u;
                            xv[u] = load4<true>(rf.mat + s_rowoff[pr2 < np ? pr2 : p0] + g * 4u);
                        }
#pragma unroll
                        for (int u = 0; u < B; ++u) {
                            const uint32_t pr2 = p0 + RG * (uint32_t)u;
                            if (pr2 < np) {
                                const float d0 = qc.x - xv[u].x, d1 = qc.y - xv[u].y, d2 = qc.z - xv[u].z, d3 = qc.w - xv[u].w;
                                float w = d0 * d0 + d1 * d1;
                                w = w + d2 * d2;
                                lds_terms[g * NP + pr2] = w + d3 * d3;
                            }
                        }
                    }
                }
            } else {
            const uint32_t items = np * Gx;
            for (uint32_t i0 = 0; i0 < items; i0 += 256u * 16u) {
                float4 xv[16], qv[16];
#pragma unroll
                for (int u = 0; u < 16; ++u) {
                    uint32_t i = i0 + 256u * (uint32_t)u + threadIdx.x;
                    i = i < items ? i : items - 1u;
                    const uint32_t pr2 = i / Gx, g = i - pr2 * Gx;
                    xv[u] = load4<true>(rf.mat + s_rowoff[pr2] + g * 4u);
                    qv[u] = qg4[g];
                }
#pragma unroll
                for (int u = 0; u < 16; ++u) {
                    const uint32_t i = i0 + 256u * (uint32_t)u + threadIdx.x;
                    if (i < items) {
                        const uint32_t pr2 = i / Gx, g = i - pr2 * Gx;
                        const float d0 = qv[u].x - xv[u].x, d1 = qv[u].y - xv[u].y, d2 = qv[u].z - xv[u].z, d3 = qv[u].w - xv[u].w;
                        float w = d0 * d0 + d1 * d1;
                        w = w + d2 * d2;
                        lds_terms[g * NP + pr2] = w + d3 * d3;
                    }
                }
            }
            }
            __syncthreads();
            PQV_STAMP_MAX(14);
            if (wave != 0) return;
            float sum = 0.0f;
            const uint32_t pcol = (uint32_t)lane < np ? (uint32_t)lane : 0u;
            for (uint32_t g = 0; g < Gx; g += 16) {                        // Gx % 16 == 0 (dim % 64 == 0 on this path)
                float t16[16];
#pragma unroll
                for (int u = 0; u < 16; ++u) t16[u] = lds_terms[(g + u) * NP + pcol];
#pragma unroll
                for (int u = 0; u < 16; ++u) sum = sum + t16[u];
            }
            uint64_t xkey = valid ? (((uint64_t)__float_as_uint(sum) << 32) | (uint64_t)(uint32_t)pos) : KEY_EMPTY;
            __shared__ uint64_t s_fin[64];
            const uint64_t kth2 = wave_kth_by_rank(xkey, k, lane, s_fin);
            if (kth != KEY_EMPTY && kth2 < kth) { kth = kth2; m1key = wave_kth_by_rank(xkey, 1u, lane, s_fin); }
        } else {
        constexpr int NB = 16;                 // row chunks a lane has in flight (the tail of a one-query call runs alone)
        uint32_t lg = 0;                       // k pairs per wave
        while (lg < 3 && (k << (lg + 1)) <= 64u && (Gx % ((uint32_t)(2 * NB) << lg)) == 0u) ++lg;
        const uint32_t L = 1u << lg;
        const uint32_t pl = (uint32_t)lane >> lg, pj = (uint32_t)lane & (L - 1u);      // pair within the wave, lane within the pair
        const uint32_t pi = (uint32_t)wave * k + pl;                                     // pair of the query: 0 .. 4 k - 1
        const uint32_t e = pi >> 2, t = pi & 3u;
        const uint64_t ekey = s_ent[e < 16u ? e : 0u];
        bool valid = pl < k && ekey != KEY_EMPTY;
        const uint32_t idx = (uint32_t)ekey;                               // index into the query's n_vals bounds
        const uint32_t l15 = idx & 15u, slot = (idx >> 4) % rf.seed_sw, j = (idx >> 4) / rf.seed_sw;
        const uint32_t row = (slot >> 2) * 256u + (slot & 3u) * 64u + 16u * t + l15;      // position in the list (wide_seed_kernel's tiling)
        uint64_t lbeg = 0, pos = 0;
        if (valid) {
            const uint32_t c = rf.probe[(uint64_t)q * rf.nprobe + j];
            lbeg = rf.list_off[c];
            pos = rf.cand_base[(uint64_t)q * rf.nprobe + j] + row;
            valid = row < rf.seed_rows && lbeg + row < rf.list_off[c + 1] && pos < rf.max_pos;
        }
        const float *x = rf.mat + (valid ? (uint64_t)(rf.row_of ? rf.row_of[lbeg + row] : lbeg + row) : 0ull) * rf.dim;
        const float4 *qg = reinterpret_cast<const float4 *>(rf.queries + (uint64_t)q * rf.dim);
        float sum = 0.0f;
        const uint32_t first = (uint32_t)lane & ~(L - 1u);
        for (uint32_t g0 = 0; g0 < Gx; g0 += NB * L) {
            const uint32_t g = g0 + NB * pj;
            float4 xv[NB], qv[NB];
#pragma unroll
            for (int u = 0; u < NB; ++u) {
                const uint32_t gu = g + u < Gx ? g + u : Gx - 1;          // (L == 1: Gx need not be a multiple of NB)
                xv[u] = load4<true>(x + gu * 4); qv[u] = qg[gu];
            }
            float tt[NB];
#pragma unroll
            for (int u = 0; u < NB; ++u) {
                const float d0 = qv[u].x - xv[u].x, d1 = qv[u].y - xv[u].y, d2 = qv[u].z - xv[u].z, d3 = qv[u].w - xv[u].w;
                float w = d0 * d0 + d1 * d1;
                w = w + d2 * d2;
                tt[u] = w + d3 * d3;
            }
            if (L == 1u) {
#pragma unroll
                for (int u = 0; u < NB; ++u) sum = sum + tt[u];
            } else {
                for (uint32_t sl = 0; sl < L; ++sl) {
                    float sn = sum;
#pragma unroll
                    for (int u = 0; u < NB; ++u) sn = sn + tt[u];
                    sum = __shfl(pj == sl ? sn : sum, (int)(first + sl), 64);
                }
            }
        }
        if (pl < k && pj == 0u && pi < 64u)
            s_exact[pi] = valid ? (((uint64_t)__float_as_uint(sum) << 32) | (uint64_t)(uint32_t)pos) : KEY_EMPTY;
        __syncthreads();
        PQV_STAMP_MAX(14);
        if (wave != 0) return;
        uint64_t xkey = (uint32_t)lane < 4u * k ? s_exact[lane] : KEY_EMPTY;
        uint32_t dummy2 = 0;
        bitonic_sort64(xkey, dummy2, lane);
        const uint64_t kth2 = readlane_u64(xkey, (int)k - 1);
        if (kth != KEY_EMPTY && kth2 < kth) { kth = kth2; m1key = readlane_u64(xkey, 0); }
        }
    } else if (wave != 0) {
        return;
    }
    if (lane == 0) {
        cand_cnt[q] = 0u;
        spilled[q] = 0u;
        // every candidate whose distance is <= the bound must pass (key compare is on (d2, position))
        if (kth != KEY_EMPTY) atomicMin(&gthr[q], (unsigned long long)(kth | 0xFFFFFFFFull));
    }
    if (thr_hist) {
        // running thresholds: 16 bins below thr0 = the k-th bound.  The final k-th distance of a query
        // usually lies a little below the SMALLEST sampled bound m1, so the bins span twice thr0 - m1.
        if (lane < 4) thr_hist[(uint64_t)q * 4 + lane] = 0u;        // two 64-bit words of 8-bit counters
        if (lane == 0) {
            float4 hb = make_float4(0.f, 0.f, 0.f, 0.f);            // 1 / w == 0: no running threshold
            if (kth != KEY_EMPTY) {
                const float thr0 = __uint_as_float((uint32_t)(kth >> 32));
                const float m1 = __uint_as_float((uint32_t)(m1key >> 32));
                float w = (thr0 - m1) * 0.125f;
                if (!(w > thr0 * 1.0e-6f)) w = thr0 * 0.00390625f;      // degenerate sample: 2^-8 of the bound
                if (w > 0.0f && w < INFINITY && thr0 < INFINITY)
                    hb = make_float4(thr0, w, 1.0f / w, thr0 * 9.5367431640625e-07f + w * 1.52587890625e-05f);
            }
            thr_bins[q] = hb;
        }
    }
    PQV_STAMP_MAX(15);
}
template <int S>
__global__ __launch_bounds__(256) void seed_select_kernel(const float *seed_ub, uint32_t n_vals, uint32_t k,
                                                        unsigned long long *gthr, uint32_t *cand_cnt, uint32_t *spilled,
                                                        uint32_t *thr_hist, float4 *thr_bins, const SeedRefine rf) {
    seed_select_body<S>(blockIdx.x, seed_ub, n_vals, k, gthr, cand_cnt, spilled, thr_hist, thr_bins, rf);
}
// QLDS forms: a one-query call (seed_tail) takes the deep-prefetch instance and tells the tail how much dynamic LDS it has
#define SEED_LAUNCH(NG_, OP_, GRID_, LDS_)                                                                              \
    {                                                                                                                   \
        TileArgs b = a;                                                                                                 \
        b.seed_tail.lds_floats = (uint32_t)((size_t)(LDS_) / 4);                                                        \
        if (deep) { dim3 g4 = GRID_; g4.x *= 4; hipLaunchKernelGGL((wide_seed_kernel<NG_, true, OP_, 12>), g4, dim3(256), (LDS_), s, b); } \
        else hipLaunchKernelGGL((wide_seed_kernel<NG_, true, OP_>), GRID_, dim3(256), (LDS_), s, b);                    \
    }
hipError_t launch_wide_seed(const TileArgs &a, hipStream_t s) {
    if (a.max_quads == 0 || a.grid_x == 0) return hipSuccess;
    if ((a.dim % 64) != 0 || !a.mat_blk || (a.row_of && !a.norm_by_pos) || !a.seed_ub) return hipErrorInvalidValue;
    const size_t lds4 = 64ull * a.dim * 4, lds2 = 32ull * a.dim * 4;
    const bool deep = a.seed_tail.enable != 0 && a.nq == 1;       // a one-query call: twelve operand stages in flight per wave
    if (a.i8) {       // int8 images: 32 queries x dim bytes per block
        if ((a.dim % 256) != 0 || !a.q_i8 || !a.q_n2i || !a.q_resu || !a.list_scale || !a.row_n2i || !a.row_res || (a.quad_width % 32) != 0 ||
            32ull * a.dim > 65536) return hipErrorInvalidValue;
        // 64 queries per pass where they fit 48 KB (a 96-query quad is then sampled in two slices instead of three:
        // the sample rows are re-read once per slice)
        if (64ull * a.dim <= 49152)
            SEED_LAUNCH(4, OP_I8, dim3(a.grid_x, a.max_quads, (a.quad_width + 63) / 64), 64ull * a.dim)
        else
            SEED_LAUNCH(2, OP_I8, dim3(a.grid_x, a.max_quads, a.quad_width / 32), 32ull * a.dim)
        return hipGetLastError();
    }
    if (a.f16) {      // f16 operands: the staged queries take half the LDS; 64 queries per block up to 256 dims, else 32
        if ((a.dim % 128) != 0 || !a.query_maxabs || a.dim > 1024) return hipErrorInvalidValue;
        if (lds4 / 2 <= 32768 && (a.quad_width % 64) == 0)
            SEED_LAUNCH(4, OP_F16, dim3(a.grid_x, a.max_quads, a.quad_width / 64), lds4 / 2)
        else if ((a.quad_width % 32) == 0)
            SEED_LAUNCH(2, OP_F16, dim3(a.grid_x, a.max_quads, a.quad_width / 32), lds2 / 2)
        else return hipErrorInvalidValue;
        return hipGetLastError();
    }
    if (a.quad_width == 64 && lds4 <= 32768)
        SEED_LAUNCH(4, OP_F32, dim3(a.grid_x, a.max_quads), lds4)
    else if (a.quad_width == 32 && lds2 <= 32768)
        SEED_LAUNCH(2, OP_F32, dim3(a.grid_x, a.max_quads), lds2)
    else if (a.quad_width == 32 && a.q_blk) {
        if (deep) hipLaunchKernelGGL((wide_seed_kernel<2, false, OP_F32, 12>), dim3(a.grid_x * 4, a.max_quads), dim3(256), 0, s, a);
        else hipLaunchKernelGGL((wide_seed_kernel<2, false, OP_F32>), dim3(a.grid_x, a.max_quads), dim3(256), 0, s, a);
    } else if (a.quad_width == 64 && a.q_blk) {
        if (deep) hipLaunchKernelGGL((wide_seed_kernel<4, false, OP_F32, 12>), dim3(a.grid_x * 4, a.max_quads), dim3(256), 0, s, a);
        else hipLaunchKernelGGL((wide_seed_kernel<4, false, OP_F32>), dim3(a.grid_x, a.max_quads), dim3(256), 0, s, a);
    }
    else return hipErrorInvalidValue;
    return hipGetLastError();
}
#undef SEED_LAUNCH
hipError_t launch_seed_select(const float *seed_ub, uint32_t nq, uint32_t n_vals, uint32_t k, unsigned long long *gthr,
                              uint32_t *cand_cnt, uint32_t *spilled, hipStream_t s,
                              uint32_t *thr_hist, float4 *thr_bins, const SeedRefine *refine) {
    if (nq == 0) return hipSuccess;
    SeedRefine rf{};
    if (refine && refine->mat && (refine->dim % 32) == 0 && k <= 16) rf = *refine;
    const dim3 block(rf.mat ? 256 : 64);
    if (k <= 64) hipLaunchKernelGGL(seed_select_kernel<1>, dim3(nq), block, 0, s, seed_ub, n_vals, k, gthr, cand_cnt, spilled, thr_hist, thr_bins, rf);
    else if (k <= 256) hipLaunchKernelGGL(seed_select_kernel<4>, dim3(nq), dim3(256), 0, s, seed_ub, n_vals, k, gthr, cand_cnt, spilled, thr_hist, thr_bins, rf);
    else return hipErrorInvalidValue;
    return hipGetLastError();
}

// ------------------------------------------------------------------------------------
// wide_filter_kernel<NG, NW, S>: the MFMA-screened re-rank with NG 16-query groups (a "quad" of up
// to 16 NG queries of one cluster) per block of NW waves.
//
// tile_filter_kernel fetches 5 operand vectors from global memory per 16 MFMAs and re-reads a
// cluster's rows once per 16-query group (8 flop per byte).  Here the block's queries are staged
// ONCE in LDS ([16 NG][dim], 16-byte columns XOR-swizzled by the query index so the A-operand
// ds_read_b128 is bank-conflict free); each wave walks its own rows exactly as before, but every
// 64-row x 16-dim B operand it loads is contracted against all NG query groups: 4 global loads per
// 16 NG MFMAs, rows re-read once per 16 NG queries, and -- unlike staging the ROWS in LDS, which
// was tried first and lost to barrier skew -- the waves never synchronise after the prologue.
//
// How often a list is streamed is what bounds the kernel on long rows: PMC on C3 (10 M x 768, 32-query
// quads, round 2) shows 53 GB of operand loads per 1024-query step, 82 % of them L2 misses, 41 GB at the
// fabric = 6.8 TB/s -- the kernel sits at the memory system's ceiling while reading every probed list
// 3x (sum over clusters of len * ceil(pairs / 32) = 30 M rows against 10 M distinct ones).  Hence the
// widest quad the LDS can hold: NW = 8 waves share ONE staged quad of up to 128 queries (96 at 768 dims:
// 144 KB of f16 images), one block per CU -- the same eight waves per CU as two 4-wave blocks, but a
// list is streamed ceil(pairs / 96) times instead of ceil(pairs / 32): 13.6 M rows on C3.
//
// Screening bound, pending queue (drained after every tile's screen, so one tile's worth of
// capacity still suffices), exact re-evaluation and fold are those of tile_filter_kernel.
// Per-query state is lane-parallel: query i of the quad lives in lane i % 64 of state slot i / 64.
// Requires dim % 64 == 0 (swizzle closure), the IVF-ordered layout (row_of == nullptr) and its blocked
// copy (mat_blk / blk_off).
// ------------------------------------------------------------------------------------
// lane-parallel per-query state of a quad: value of query qi (per-lane index / wave-uniform index)
template <int QS>
__device__ __forceinline__ uint32_t qsel_u32(const uint32_t (&v)[QS], uint32_t qi) {
    uint32_t r = (uint32_t)__shfl((int)v[0], (int)(qi & 63u), 64);
#pragma unroll
    for (int s = 1; s < QS; ++s) { const uint32_t rs = (uint32_t)__shfl((int)v[s], (int)(qi & 63u), 64); r = (qi >> 6) == (uint32_t)s ? rs : r; }
    return r;
}
template <int QS>
__device__ __forceinline__ uint64_t qsel_u64(const uint64_t (&v)[QS], uint32_t qi) {
    uint64_t r = shfl_u64(v[0], (int)(qi & 63u));
#pragma unroll
    for (int s = 1; s < QS; ++s) { const uint64_t rs = shfl_u64(v[s], (int)(qi & 63u)); r = (qi >> 6) == (uint32_t)s ? rs : r; }
    return r;
}
template <int QS>
__device__ __forceinline__ uint32_t qread_u32(const uint32_t (&v)[QS], uint32_t qq) {      // qq wave-uniform
    // (the lane is read from EVERY slot and the scalars are selected: a select between the array's elements on a uniform
    //  condition is turned into ONE load through a selected address -- the array then lives in scratch memory and every
    //  access to it, the screen's static ones included, is a VMEM load: seen with QS = 3, the wide-quad instance)
    uint32_t x = readlane_u32(v[0], (int)(qq & 63u));
#pragma unroll
    for (int s = 1; s < QS; ++s) { const uint32_t xs = readlane_u32(v[s], (int)(qq & 63u)); x = (qq >> 6) == (uint32_t)s ? xs : x; }
    return x;
}
template <int QS>
__device__ __forceinline__ uint64_t qread_u64(const uint64_t (&v)[QS], uint32_t qq) {
    uint64_t x = readlane_u64(v[0], (int)(qq & 63u));
#pragma unroll
    for (int s = 1; s < QS; ++s) { const uint64_t xs = readlane_u64(v[s], (int)(qq & 63u)); x = (qq >> 6) == (uint32_t)s ? xs : x; }
    return x;
}

// ONCE: every row of the launch is read by exactly one block (a batch that fits one quad, a one-query call above all): the
// operand stream carries the nt policy, so it does not displace the queries' images and thresholds from L2 / the Infinity
// Cache (C3 single query 174 -> 166 us; on batches whose long lists are streamed twice the same hint costs 3.5 %).
// TS: 16-row sub-tiles per wave tile.  4 (64-row tiles) everywhere but the WIDE-QUAD instance <10, 8, .., TS = 2>: 32-row tiles
// halve the accumulator registers per query group, so ONE block holds a quad of 160 queries (120 KB of int8 images) and a
// list that 97..160 queries of the batch probe is streamed once instead of twice (launch_tile_filter, TileArgs::wide_*).
// DEFP: the deferred-evaluation form (5.4b) also in a k <= 64 instance (the k > 64 instances always carry it)
template <int NG, int NW, int S, bool QLDS, int OP, bool PF, bool ONCE, int TS, bool DEFP>
__global__ __launch_bounds__(64 * NW, (NW == 8 || (NG == 4 && !QLDS) || (QLDS && OP != OP_F32)) ? 2 : 3) void wide_filter_kernel(const TileArgs a) {
    constexpr int ROW_AUX = ONCE ? 2 : PQV_ROW_AUX;
    constexpr bool F16 = OP == OP_F16, I8 = OP == OP_I8;
    static_assert(!PF || (QLDS && F16), "whole-tile operand prefetch: f16 rows of <= 128 dims");
    static_assert(!I8 || QLDS, "int8 operands: queries staged in LDS");
    static_assert(TILE_QB == 16 && NG >= 2 && NG <= 12 && (NG % 2) == 0 && (NW == 4 || NW == 8), "16-row MFMA tiles, 2..12 groups, 4 or 8 waves");
    static_assert(TS == 4 || (TS == 2 && QLDS && !PF && OP != OP_F32), "32-row tiles: staged queries, int8 / f16 operands");
    constexpr uint32_t TROWS = 16 * TS;            // rows per wave tile
    constexpr uint32_t FW = 4 * TS, GPW = 32 / FW; // keep-bits per lane and group; groups per 32-bit word
    constexpr int NWD = (NG + (int)GPW - 1) / (int)GPW;
    constexpr uint32_t NQ = 16 * NG;
    constexpr uint32_t QSH = NQ > 128 ? 24 : 25;   // queue entry = (query index << QSH) | row offset from the wave's r0
    constexpr uint32_t QCNT = 1u << (QSH - 1);     // ... | QCNT: "already counted in the running-threshold bins" (deferred pass -> exact pass);
                                                   // row offsets stay below 2^23 (launch_wide checks rows_per_block)
    constexpr int QS = (NQ + 63) / 64;        // state slots per lane
    // survivors are expanded into the wave's queue a PASS at a time when a tile's do not fit at once: half a
    // group's pairs (queries r < 2 / r >= 2 of every lane: <= 512 entries) for the 4-wave blocks, a quarter
    // (<= 256) for the 8-wave blocks, whose LDS belongs to the staged queries
    // (64 f16 queries on 8 waves -- K = 100 on 1024-dim rows: a tile often has more than 192 survivors, and only what is expanded
    //  in one pass can be deferred; its quad takes <= 128 KB, so the queues may have 18 KB)
    constexpr int PASS = (int)wide_filter_pend(NQ, NW, OP == OP_F16) - 64;      // (96 int8 queries, two blocks per CU: 80 KB each)
    constexpr int PEND = PASS + 64;        // one pass + a partial batch
    constexpr int NT = 64 * NW;
#ifdef PQV_PROFILE_PHASES
    const uint64_t ph_t0 = __builtin_amdgcn_s_memtime();
#endif
    PQV_STAMP_MIN(16);
    uint32_t bx, by;
    uint4 quad;                              // {cluster, first pair slot, pair count <= NQ, first work item}
    if (a.item_quad) {
        // 1-D grid over the work items (quad, existing row chunk): the lists are very unequal, and a (chunks of the
        // longest list) x quads grid is mostly workgroups that exit at once, in a pattern that decides which XCD gets
        // the real ones
        const uint32_t item = blockIdx.x;
        if (item >= *a.n_items) return;
        by = a.item_quad[item];
        quad = a.quads[by];
        bx = a.item_chunk ? a.item_chunk[item] : item - quad.w;
    } else {
        quad_xcd_remap(bx, by, a.xcd_swizzle, *a.n_quads);
        if (by >= *a.n_quads) return;
        quad = a.quads[by];
    }
    const uint32_t c = quad.x, p0 = quad.y;
    uint32_t cnt = quad.z;
    int lane = threadIdx.x & 63;                 // (re-defined opaquely after every K loop: PQV_RELANE below)
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const uint32_t k = a.k;
    // Pairs that cannot contribute: pair_lb[p] is a lower bound of the reference distance between the pair's query and
    // EVERY row of this list (triangle inequality on the list's centre, quantize_pairs_i8_kernel); once it exceeds the
    // query's admission threshold none of the list's rows can enter that query's top-k.  Wave 0 compacts the quad's
    // live pairs to the front (s_perm); a quad without live pairs reads nothing at all.  (The thresholds only ever
    // tighten, so a pair found dead here stays dead; results do not depend on when a block looks.)
    __shared__ uint32_t s_perm[NQ];
    __shared__ uint32_t s_live;
    const bool prune = I8 && a.pair_lb != nullptr;
    if (prune) {
        if (wave == 0) {
            uint32_t nlive = 0;
#pragma unroll
            for (int s = 0; s < QS; ++s) {
                const uint32_t qi = 64u * (uint32_t)s + (uint32_t)lane;
                bool live = false;
                if (qi < cnt) {
                    const uint32_t pair = a.pairs[p0 + qi];
                    const unsigned long long thr = __hip_atomic_load(a.gthr + pair / a.nprobe, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    live = thr == KEY_EMPTY || !(a.pair_lb[pair] > __uint_as_float((uint32_t)(thr >> 32)));
                }
                const unsigned long long m = __ballot(live);
                if (live) s_perm[nlive + (uint32_t)__popcll(m & ((1ull << lane) - 1ull))] = qi;
                nlive += (uint32_t)__popcll(m);
            }
            if (lane == 0) s_live = nlive;
        }
        __syncthreads();
        cnt = s_live;
        if (cnt == 0) return;
    }
    const uint32_t ng = (cnt + 15) >> 4;             // active groups (wave-uniform)

    extern __shared__ float4 qs[];                   // [NQ][dim / 4], column ch of query q at ch ^ (q & 15)
    __shared__ uint32_t pend_all[NW * PEND];         // (query index << QSH) | row offset from the wave's r0
    uint32_t *pend = pend_all + wave * PEND;
    // deferred evaluation (TileArgs::cand_lb): the queue entry's raw screen score lives in the wave's strip of a global
    // scratch (written at expansion, read back when the entry is appended: one wave, program order; no LDS to spare)
    // (compiled into the S > 1 instances -- k > 64 -- and the DEFP ones only, so that the others keep their registers: with a
    //  run-time switch alone C3 lost 5 %)
    const bool defer = (S > 1 || DEFP) && a.cand_lb != nullptr;
    uint32_t *pv = defer ? a.pendv + ((uint64_t)(blockIdx.y * gridDim.x + blockIdx.x) * NW + (uint32_t)wave) * PEND : nullptr;
    constexpr uint32_t NOVAL = I8 ? 0x80000000u : 0x7FC00000u, NOVAL_NOHIST = I8 ? 0x80000001u : 0x7FC00001u;
    __shared__ __attribute__((aligned(16))) float aq_all[NW * NQ];   // per-wave, per-query screen terms
    float *aq = aq_all + wave * NQ;

    const uint64_t lbeg = a.list_off[c], lend = a.list_off[c + 1];
    const uint64_t len = lend - lbeg;
    uint64_t wrows = a.rows_per_block / NW;
    uint64_t r0 = a.row_offset + (uint64_t)bx * a.rows_per_block + (uint64_t)wave * wrows;
    if (a.row_offset == 0 && a.row_end == 0) {
        // the whole list in this launch: ceil(len / rows_per_block) blocks share it in EQUAL wave pieces (a
        // multiple of the 64-row tile), so no block is left with a sliver of a last chunk -- its fixed cost
        // (staging the quad's queries, the final partial batch of exact evaluations) would be wasted
        const uint64_t nch = (len + a.rows_per_block - 1) / a.rows_per_block;
        if (bx >= nch) return;
        wrows = ((len + NW * nch - 1) / (NW * nch) + TROWS - 1) / TROWS * TROWS;
        r0 = ((uint64_t)bx * NW + (uint64_t)wave) * wrows;
    }
    uint64_t r1 = r0 + wrows;
    if (r1 > len) r1 = len;
    if (a.row_end && r1 > a.row_end) r1 = a.row_end;     // window of this launch
    if (r0 > len) r0 = len;

    const uint32_t dim = a.dim;
    // 16-byte operand columns per row: 4 f32, 8 f16 or 16 int8 values each; a K step is 4 columns
    const uint32_t G = I8 ? dim >> 4 : F16 ? dim >> 3 : dim >> 2;
    const uint32_t Gx = dim >> 2;                    // 16-byte chunks of a row-major f32 row (exact evaluation)
    const float cmargin = (float)(dim + 16) * 2.384185791015625e-07f;   // (dim + 16) * 2^-22
    // F16: operands are round-to-nearest f16 images of scale * value (|scale * x| <= 2^14: no overflow), the
    // products are exact in f32, so the score carries an extra error <= 2^-11 (1 + 2^-12) nn (relative
    // 2^-11 per operand, sum |q x| <= nn / 2) and d~ an extra 2^-10 nn; sub-normal images (absolute error
    // 2^-25) add < 3 % of that once scale^2 |q|^2 >= 1 -- queries below that, and queries whose image
    // overflows, are never skipped (see aq below).  c16 = 1.25 * 2^-10 carries both with a margin.
    const float c16 = F16 ? 1.25f * 9.765625e-04f : 0.0f;
    const float inv1c = 1.0f / (1.0f - cmargin);
    [[maybe_unused]] float lscale = 1.0f;            // int8: this list's scale
    if constexpr (I8) lscale = a.list_scale[c];
    // scores are contracted at scale^2: alpha and beta absorb it (powers of two: exact)
    const float sc2 = F16 ? a.scale2 : 1.0f;
    const float alpha = sc2 * 0.5f * (1.0f - (2.0f * cmargin + c16) * inv1c), beta = sc2 * 0.5f * inv1c;

    // lane-parallel per-query state: slot s, lane l own query 64 s + l of the quad (queries past cnt alias the
    // last one; they are masked wherever it matters)
    // The 8-wave blocks keep the per-query CONSTANTS (candidate base, norms, int8 terms) in LDS, written once by
    // every wave with the same values: in registers they were the first thing the allocator spilled, and a spill
    // reload is a VMEM load -- consuming it drains the wave's whole queue of prefetched operands (vmcnt(0)) at the
    // top of every tile.  LDS reads count on lgkmcnt and leave the operand stream alone.
    constexpr bool LST = NW == 8 || I8 || NG == 6;
    __shared__ uint64_t qst_cbase[LST ? NQ : 1];
    __shared__ uint32_t qst_pair[LST ? NQ : 1];
    __shared__ float qst_qn[LST && !I8 ? NQ : 1];     // |q|^2; NaN = never skip this query (float operand forms)
    // (the two the screen reads by LANE index are padded to 64 QS entries: slot s of lane l reads entry 64 s + l without a clamp --
    //  a clamped index is a computed address in a register of its own, which the allocator spilled and reloaded right behind
    //  the next tile's operand prefetch: a wait for the whole queue; entries past NQ are never used)
    __shared__ float qst_res[LST ? 64 * QS : 1];      // int8: residual bound (+inf = never skip)
    __shared__ int qst_n2i[LST ? 64 * QS : 1];        // int8: |qi|^2
    uint32_t my_qrow[QS];
    [[maybe_unused]] uint32_t my_pairi[I8 ? QS : 1];         // int8: the pair (its image is per (query, list))
    [[maybe_unused]] uint32_t my_pair[LST ? 1 : QS];
    [[maybe_unused]] uint64_t my_cbase[LST ? 1 : QS], my_base[LST ? 1 : QS];
    [[maybe_unused]] float my_qn[LST ? 1 : QS];
    [[maybe_unused]] bool my_noskip[LST ? 1 : QS];
    const uint32_t n_part = a.n_part;
#pragma unroll
    for (int s = 0; s < QS; ++s) {
        const uint32_t qi = 64u * (uint32_t)s + (uint32_t)lane;
        const uint32_t qic = qi < cnt ? qi : cnt - 1;
        const uint32_t pair = a.pairs[p0 + (prune ? s_perm[qic] : qic)];
        my_qrow[s] = pair / a.nprobe;
        if constexpr (I8) my_pairi[s] = a.i8_pair_images ? pair : my_qrow[s];      // the image: per pair or per query
        const float qn = a.query_norm2[my_qrow[s]];
        // F16: a query whose scaled image overflows f16 or whose scaled norm is below 1 is never skipped
        const bool noskip = F16 && (!(a.query_maxabs[my_qrow[s]] * a.scale <= 32768.0f) || !(qn * a.scale2 >= 1.0f) || !(qn <= 3.0e38f));
        if constexpr (LST) {
            if (qi < NQ) {
                qst_pair[qi] = pair;
                qst_cbase[qi] = a.cand_base[pair];
                if constexpr (!I8) qst_qn[qi] = noskip ? __uint_as_float(0x7FC00000u) : qn;
                if constexpr (I8) { qst_n2i[qi] = a.q_n2i[my_pairi[s]]; qst_res[qi] = a.q_res[my_pairi[s]]; }   // +inf: non-finite query
            }
        } else {
            my_pair[s] = pair;
            my_cbase[s] = a.cand_base[pair];
            my_qn[s] = qn;
            my_base[s] = ((uint64_t)my_qrow[s] * n_part + (pair % a.nprobe) * a.slots_per_pair + a.slot_base + bx * NW + wave) * k;
            my_noskip[s] = noskip;
        }
    }

    // QLDS: stage the quad's queries: TPQ threads per query, 16-byte columns interleaved between them;
    // the row pointer comes from the lane-parallel state.
    // !QLDS (rows too long for LDS): the A operands come from the quad's BLOCKED query copy in global
    // memory (pack_queries_kernel; L2-resident), fetched like the B operands -- 1 KiB per load.
    if constexpr (QLDS) {
        constexpr uint32_t NQP2 = NQ <= 32 ? 32 : NQ <= 64 ? 64 : NQ <= 128 ? 128 : 256;
        constexpr uint32_t TPQ = NT / NQP2;
        const uint32_t q = threadIdx.x / TPQ, c0 = threadIdx.x % TPQ;
        const uint32_t q_src = qsel_u32<QS>(my_qrow, q < NQ ? q : NQ - 1);
        // (the shuffles must run with every lane active: a lane that skips the staging still SERVES its state to others)
        [[maybe_unused]] uint32_t p_src = 0;
        if constexpr (I8) p_src = qsel_u32<QS>(my_pairi, q < NQ ? q : NQ - 1);
        if (q < 16u * ng) {        // only the active groups are ever read
            const float4 *src = reinterpret_cast<const float4 *>(a.queries + (uint64_t)q_src * dim);
            float4 *dst = qs + q * G;
            const uint32_t sw = q & 15u;
            if constexpr (I8) {        // the int8 images were made once per batch and pair (quantize_pairs_i8_kernel)
                const float4 *s8 = reinterpret_cast<const float4 *>(a.q_i8 + (uint64_t)p_src * dim);
#pragma unroll PQV_STAGE_UNROLL
                for (uint32_t ch = c0; ch < G; ch += TPQ) dst[ch ^ sw] = s8[ch];
            } else if constexpr (F16) {
                if (a.q32_lds) {          // short rows: the exact f32 queries too, for the exact evaluation of survivors
                    float4 *d32 = qs + NQ * G + q * Gx;
#pragma unroll 4
                    for (uint32_t ch = c0; ch < G; ch += TPQ) {
                        const float4 lo = src[2 * ch], hi = src[2 * ch + 1];
                        dst[ch ^ sw] = pack_f16x8_clamped(lo, hi, a.scale);
                        d32[(2 * ch) ^ sw] = lo;
                        d32[(2 * ch + 1) ^ sw] = hi;
                    }
                } else {
#pragma unroll 4
                    for (uint32_t ch = c0; ch < G; ch += TPQ) dst[ch ^ sw] = pack_f16x8_clamped(src[2 * ch], src[2 * ch + 1], a.scale);
                }
            } else {
#pragma unroll 8
                for (uint32_t ch = c0; ch < G; ch += TPQ) dst[ch ^ sw] = src[ch];
            }
        }
        __syncthreads();
    }
    const float4 *qblk = a.q_blk + (uint64_t)by * NG * G * 16;   // + (g G + ch) 16 + query-in-group
    const __amdgpu_buffer_rsrc_t qr = operand_rsrc(QLDS ? (const void *)a.queries : (const void *)qblk);

    int l15 = lane & 15, kk = lane >> 4;
    const uint64_t blk0 = a.blk_off[c], blk_last = a.blk_off[c + 1] - 1;   // the list's 16-row tiles
    uint32_t npend = 0;
    uint32_t n_exact = 0;
#ifdef PQV_PROFILE_PHASES
    uint64_t ph_k = 0, ph_s = 0, ph_e = 0, ph_em = 0, ph_top_sum = 0, ph_xt = 0; const uint64_t ph_pro = __builtin_amdgcn_s_memtime() - ph_t0;
#endif

    // The queries' running thresholds as the wave sees them: the DISTANCE half of the 64-bit threshold key only (one register per
    // slot).  A pair is kept / appended whenever its distance does not exceed it -- the position half of the key only decides
    // between equal distances, and a lenient test merely appends a pair the merge drops again.  Read through a wave-uniform
    // buffer descriptor (no per-lane 64-bit pointer registers), agent scope (sc1): thresholds tighten while the kernel runs.
    // (the k-th key of a wave's own overflow list is not kept: tile_fold publishes it to the query's global threshold)
    const __amdgpu_buffer_rsrc_t rt_thr = operand_rsrc(a.gthr);
    auto load_thr = [&](int s) -> uint32_t { return buf_ld4<16>(rt_thr, my_qrow[s] * 8u + 4u, 0u); };
    auto widen = [](uint32_t h) -> uint64_t { return ((uint64_t)h << 32) | 0xFFFFFFFFull; };
    uint32_t cur_gthr[QS];
#pragma unroll
    for (int s = 0; s < QS; ++s) cur_gthr[s] = load_thr(s);
    // Running threshold + append of one lane's pair (deferred: the UPPER bound stands for the distance -- a pair counted in a
    // bin is truly at or below the bin's edge -- and lbv >= 0 is its lower bound; exact: lbv = -1).  Returns false when the
    // query's buffer is full.
    auto append_pair = [&](bool pass, uint32_t qrow, float dval, uint64_t key, uint32_t srow, float lbv, bool count_hist) -> bool {
        // k == 1: the distance itself.  Otherwise the query has 12 bins below its
        // seed threshold thr0 (bin b = [thr0 - (b + 1) w, thr0 - b w), the last one open-ended) and one 8-bit
        // counter per bin b >= 1 holding the number of appended pairs in bin b OR NEARER: word 0 = bins 8..1,
        // word 1 = bins 12..9, the NEARER bin in the LOWER byte.  An append adds 1 to the counters of bins
        // 1..b with one returning atomic per word -- issued together with the append's own counter, one round
        // trip in all -- and the returned word says whether this add took some counter to k: then that bin's
        // upper edge (+ the rounding pad of the bin arithmetic) bounds the final k-th distance, and exactly one
        // lane publishes it.  A counter that wraps (> 255 pairs) carries into the next byte, the counter of a
        // FARTHER bin, which truly holds at least as many pairs (>= 256 > k): every value the bytes can show
        // is either an under-count or the count of a bin that does hold k pairs.  No look-up, no extra loads.
        int hb_bin = 0;
        float4 hb = make_float4(0.f, 0.f, 0.f, 0.f);
        if (pass && count_hist && k > 1u && a.thr_hist) {
            hb = a.thr_bins[qrow];
            hb_bin = hb.z > 0.0f ? (int)fminf(fmaxf((hb.x - dval) * hb.z, 0.0f), 12.0f) : 0;
        }
        bool full = false;
        if (pass) {
            const unsigned long long ones = 0x0101010101010101ull;
            unsigned long long w0 = 0ull, w1 = 0ull;
            unsigned long long *h2 = reinterpret_cast<unsigned long long *>(a.thr_hist) + (uint64_t)qrow * 2;
            const int b = hb_bin;
            // byte j of word 0 = bin 8 - j: bins <= b are bytes j >= 8 - b;  byte j of word 1 = bin 12 - j
            const unsigned long long add0 = b >= 8 ? ones : b >= 1 ? ones << (8 * (8 - b)) : 0ull;
            const unsigned long long add1 = b >= 9 ? (ones & 0xFFFFFFFFull) << (8 * (12 - b)) & 0xFFFFFFFFull : 0ull;
            if (add1) w1 = atomicAdd(h2 + 1, add1) + add1;
            if (add0) w0 = atomicAdd(h2, add0) + add0;
            const uint32_t idx = atomicAdd(a.cand_cnt + qrow, 1u);
            if (idx < a.cand_cap) {
                a.cand_keys[(uint64_t)qrow * a.cand_cap + idx] = key;
                a.cand_vals[(uint64_t)qrow * a.cand_cap + idx] = srow;
                if (defer) a.cand_lb[(uint64_t)qrow * a.cand_cap + idx] = lbv;
            } else {
                full = true;           // buffer full: fall back to this wave's sorted list (slow, exact)
                a.spilled[qrow] = 1u;
            }
            if (k == 1u) {
                atomicMin(a.gthr + qrow, (unsigned long long)(key | 0xFFFFFFFFull));
            } else if (b > 0) {
                // the nearest bin <= b whose counter shows exactly k after this add
                int bsel = 0;
#pragma unroll
                for (int jj = 7; jj >= 0; --jj) {            // far -> near: the last match is the nearest
                    const int bin = 8 - jj;
                    if (bin <= b && (uint32_t)((w0 >> (8 * jj)) & 0xFFu) == k) bsel = bin;
                }
#pragma unroll
                for (int jj = 3; jj >= 0; --jj) {
                    const int bin = 12 - jj;
                    if (bin <= b && (uint32_t)((w1 >> (8 * jj)) & 0xFFu) == k) bsel = bin;
                }
                if (bsel > 0) {
                    const float e = hb.x - (float)bsel * hb.y + hb.w;
                    if (e < hb.x && e >= 0.0f)
                        atomicMin(a.gthr + qrow, ((unsigned long long)__float_as_uint(e) << 32) | 0xFFFFFFFFull);
                }
            }
        }
        return !full;
    };
    uint32_t n_defer = 0;
    // Deferred pass over queue entries [start, start + count), a lane per entry: the entry's screen score gives a lower and an
    // upper bound of the reference distance (the screen's own inequality, and wide_seed_kernel's); the pair is appended with
    // both and the streaming wave is done with it.  Entries without usable bounds -- or whose query's buffer is full -- are
    // compacted to the front of the segment for the exact evaluation below; returns how many.
    auto defer_pass = [&](uint32_t start, uint32_t count) -> uint32_t {
        wave_lds_fence();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        const bool valid = (uint32_t)lane < count;
        const uint32_t pe = pend[start + (valid ? (uint32_t)lane : 0u)];
        const uint32_t raw = valid ? pv[start + (uint32_t)lane] : NOVAL;
        const uint32_t qsl = pe >> QSH;
        const uint64_t roff = r0 + (pe & (QCNT - 1u));
        const uint64_t lpos = lbeg + roff;
        const uint32_t srow = a.row_of ? a.row_of[lpos] : (uint32_t)lpos;
        const uint32_t qrow = qsel_u32<QS>(my_qrow, qsl);
        const float thr_d = __uint_as_float(qsel_u32<QS>(cur_gthr, qsl));
        uint64_t pos;
        if constexpr (LST) pos = qst_cbase[qsl] + roff; else pos = qsel_u64<QS>(my_cbase, qsl) + roff;
        float lb = 0.0f, ub = INFINITY;
        bool ok = valid && raw != NOVAL && raw != NOVAL_NOHIST;
        if constexpr (I8) {
            // |vi - xi|^2 = Nq + Nx - 2 dot EXACTLY; the accumulator started at -(Nx >> 1)
            const uint32_t img = qsel_u32<QS>(my_pairi, qsl);
            const int xn2 = a.row_n2i[lpos];
            const float rx = a.row_res[lpos];
            const int n = qst_n2i[qsl] + (xn2 & 1) - 2 * (int)raw;
            const float rq = qst_res[qsl], rqu = a.q_resu[img];
            const float nf = fmaxf((float)n, 0.0f), inv_s = 1.0f / lscale;
            const float du = sqrtf(nf * 1.000001f + 2.0f) * 1.000001f * inv_s + rqu + rx;          // wide_seed_kernel's bound
            ub = du * du * (1.0f + 4.0f * cmargin) * 1.000002f;
            const float dl = sqrtf(nf) * 0.999998f * inv_s - rq - rx;                              // the screen's bound, this row's residual
            lb = dl > 0.0f ? dl * dl * (1.0f / (1.0f + 4.0f * cmargin)) * 0.999997f : 0.0f;
            ok = ok && rq <= 3.0e38f && rqu <= 3.0e38f && rx <= 3.0e38f;
        } else {
            float qn;
            bool noskip;
            if constexpr (LST) { qn = qst_qn[qsl]; noskip = !(qn == qn); }
            else {
                uint32_t qnb[QS], nsk[QS];
#pragma unroll
                for (int s = 0; s < QS; ++s) { qnb[s] = __float_as_uint(my_qn[s]); nsk[s] = my_noskip[s] ? 1u : 0u; }
                qn = __uint_as_float(qsel_u32<QS>(qnb, qsl));
                noskip = qsel_u32<QS>(nsk, qsl) != 0u;
            }
            const float sv = __uint_as_float(raw);
            const float nn = qn + a.row_norm2[lpos];
            const float dt = nn - 2.0f * (sv * (1.0f / sc2));
            ub = dt + cmargin * (2.0f * nn + fabsf(dt)) + c16 * nn;                                 // wide_seed_kernel's bound
            lb = fmaxf(((1.0f - cmargin) * dt - (2.0f * cmargin + c16) * nn) * 0.999999f - nn * 1.0e-7f, 0.0f);   // skip <=> lb > thr, with a pad
            ok = ok && !noskip && sv == sv && nn <= 3.0e38f;
        }
        ok = ok && ub >= 0.0f && ub <= 3.0e38f && lb <= ub;
        const bool in_cap = valid && pos < a.max_pos;
        // (the threshold may have tightened since the screen; an unset one is NaN: never dropped)
        const bool dpass = ok && in_cap && !(lb > thr_d);
        const uint64_t dkey = ((uint64_t)__float_as_uint(ub) << 32) | (uint64_t)(uint32_t)pos;
        const bool appended = append_pair(dpass, qrow, ub, dkey, srow, lb, true);
        n_defer += (uint32_t)__popcll(__ballot(dpass && appended));
        // what is left for the exact evaluation: no bounds, or no room
        const bool need = in_cap && (!ok || (dpass && !appended));
        const unsigned long long nm = __ballot(need);
        const uint32_t rank = (uint32_t)__popcll(nm & ((1ull << lane) - 1ull));
        wave_lds_fence();
        if (need) pend[start + rank] = pe | ((dpass && !appended) ? QCNT : 0u);       // (counted in the bins already)
        wave_lds_fence();
        return (uint32_t)__popcll(nm);
    };
    auto eval = [&](uint32_t start, uint32_t count) -> uint32_t {
#ifdef PQV_PROFILE_PHASES
        const uint64_t ph_e0 = __builtin_amdgcn_s_memtime();
#endif
        if (defer) {
            count = defer_pass(start, count);
            if (count == 0) {
#ifdef PQV_PROFILE_PHASES
                ph_em += (__builtin_amdgcn_s_memtime() - ph_e0) | (1ull << 48);
#endif
                return 0u;
            }
        }
        wave_lds_fence();
        // A batch that does not fill the wave (the last one of every wave; most batches of long rows) gives each
        // pair L = 2, 4 or 8 lanes: per round the L lanes of a pair fetch L x 8 consecutive row chunks -- the
        // scattered reads are latency, and a 768-dim row is 24 round trips for one lane, 3 for eight -- and the
        // reference's chain passes through them in chunk order (lane 0's eight adds, then lane 1's, ...), so the
        // sum is bit-identical.  The pair's first lane carries on with the result.
        // (32-row tiles: 16 row chunks in flight per lane -- its batches are larger (a lane per pair: 12 round trips per
        //  768-dim row instead of 24), and the accumulators, dead here, leave the registers)
        constexpr int NB = TS == 2 ? PQV_EVAL_NB_TS2 : (I8 && NW == 4 && NG == 6) ? PQV_EVAL_NB : 8;
        uint32_t lg = 0;
        while (lg < 3 && (count << (lg + 1)) <= 64u && (Gx % ((2u * NB) << lg)) == 0u) ++lg;      // wave-uniform
        const uint32_t L = 1u << lg;
        const uint32_t pi = (uint32_t)lane >> lg, pj = (uint32_t)lane & (L - 1u);
        const bool valid = pi < count;
        const bool have = valid && pj == 0u;
        const uint32_t pe = pend[start + (valid ? pi : 0)];
        const uint32_t qsl = pe >> QSH;                       // query index in the quad
        const uint64_t roff = r0 + (pe & (QCNT - 1u));         // row offset in the list
        const uint64_t lpos = lbeg + roff;
        const uint32_t srow = a.row_of ? a.row_of[lpos] : (uint32_t)lpos;
        const float *x = a.mat + (uint64_t)srow * dim;
        const float4 *ql = F16 ? qs + NQ * G + qsl * Gx : qs + qsl * G;   // the pair's f32 query, staged (swizzled) in LDS
        const uint32_t qsw = qsl & 15u;
        const uint32_t qrow = qsel_u32<QS>(my_qrow, qsl);
        const float4 *qg = reinterpret_cast<const float4 *>(a.queries + (uint64_t)qrow * dim);
        float sum = 0.0f;
        // 8 row chunks in flight per lane, then the reference's ordered chain over them (16 in flight -- two
        // round trips per 128-dim row instead of four -- measured no faster and costs the last free registers)
        const bool q_global = !QLDS || I8 || (F16 && !a.q32_lds);      // wave-uniform
        auto chain = [&](auto qg_c) {
            constexpr bool QG = decltype(qg_c)::value;
            const uint32_t first = (uint32_t)lane & ~(L - 1u);          // first lane of this pair's group
            for (uint32_t g0 = 0; g0 < Gx; g0 += NB * L) {
                const uint32_t g = g0 + NB * pj;
                float4 xv[NB];
#pragma unroll
                for (int u = 0; u < NB; ++u) xv[u] = load4<true>(x + (g + u) * 4);
                float4 qvv[QG ? NB : 1];
                if constexpr (QG) {
#pragma unroll
                    for (int u = 0; u < NB; ++u) qvv[u] = qg[g + u];
                }
                float t[NB];
#pragma unroll
                for (int u = 0; u < NB; ++u) {
                    float4 qv;
                    if constexpr (QG) qv = qvv[u]; else qv = ql[(g + u) ^ qsw];
                    const float d0 = qv.x - xv[u].x, d1 = qv.y - xv[u].y;
                    const float d2 = qv.z - xv[u].z, d3 = qv.w - xv[u].w;
                    float tt = d0 * d0 + d1 * d1;
                    tt = tt + d2 * d2;
                    t[u] = tt + d3 * d3;
                }
                if (L == 1u) {
#pragma unroll
                    for (int u = 0; u < NB; ++u) sum = sum + t[u];
                } else {
                    for (uint32_t sl = 0; sl < L; ++sl) {                // the chain visits the group's lanes in order
                        float sn = sum;
#pragma unroll
                        for (int u = 0; u < NB; ++u) sn = sn + t[u];
                        sum = __shfl(pj == sl ? sn : sum, (int)(first + sl), 64);
                    }
                }
            }
        };
        if (q_global) chain(std::true_type{});
        else chain(std::false_type{});
#ifdef PQV_PROFILE_PHASES
        if (__float_as_uint(sum) == 0x7FC12345u) __builtin_trap();      // consume the sum before the timestamp
        ph_em += (__builtin_amdgcn_s_memtime() - ph_e0) | (1ull << 48);
#endif
        // the wave's view of the thresholds (refreshed every tile)
        uint64_t pos;
        if constexpr (LST) pos = qst_cbase[qsl] + roff; else pos = qsel_u64<QS>(my_cbase, qsl) + roff;
        const uint64_t mykey_all =
            (have && pos < a.max_pos) ? (((uint64_t)__float_as_uint(sum) << 32) | (uint64_t)(uint32_t)pos) : KEY_EMPTY;
        // A pair that beats its query's threshold is APPENDED to the query's candidate buffer: one
        // atomic per lane, all lanes in parallel (a sorted per-wave list would cost one global
        // read-modify-write round trip per query, serially -- measured: half of the kernel).
        const uint64_t pair_thr = widen(qsel_u32<QS>(cur_gthr, qsl));
        const bool pass = mykey_all < pair_thr;
        const bool counted = (pe & QCNT) != 0u;               // (deferred pass: in the bins already)
        const bool spill = !append_pair(pass, qrow, sum, mykey_all, srow, -1.0f, !counted);
        unsigned long long todo = __ballot(spill);
        while (todo) {
            const uint32_t qq = readlane_u32(qsl, __builtin_ctzll(todo));
            const bool mine = spill && qsl == qq;
            todo &= ~__ballot(mine);
            const uint64_t mykey = mine ? mykey_all : KEY_EMPTY;
            const uint64_t thr = widen(qread_u32<QS>(cur_gthr, qq));
            if (__ballot(mykey < thr) != 0ull) {
                uint32_t pr;
                if constexpr (LST) pr = qst_pair[qq]; else pr = qread_u32<QS>(my_pair, qq);
                const uint64_t li = (uint64_t)qread_u32<QS>(my_qrow, qq) * n_part + (pr % a.nprobe) * a.slots_per_pair + a.slot_base + bx * NW + wave;
                const uint64_t base = li * k;
                // first fold into this list: it still holds whatever an earlier batch left there
                bool fresh = false;
                if (a.part_flags) {
                    fresh = a.part_flags[li] == 0;
                    if (fresh && lane == 0) a.part_flags[li] = 1;
                }
                const uint64_t nk = tile_fold<S>(a.part_keys + base, a.part_vals + base,
                                                 a.gthr + qread_u32<QS>(my_qrow, qq),
                                                 thr, KEY_EMPTY, mykey, srow, k, lane, fresh);
                (void)nk;
            }
        }
        return count;
    };
    auto drain = [&](uint32_t keep_below) -> bool {        // true: exact evaluations ran (the caller re-issues the operand prefetch)
        bool ran = false;
        while (npend >= keep_below && npend > 0) {
            const uint32_t take = npend < 64 ? npend : 64;
            n_exact += eval(npend - take, take);
            npend -= take;
            ran = true;
        }
        wave_lds_fence();
        return ran;
    };

    // A wave evaluates its queue when this many survivors wait (and behind its last tile).  64 fills the wave; fewer -- several
    // lanes per pair -- makes the results (and with them the queries' running thresholds) arrive during the scan instead of at its end
    const uint32_t drain_min = a.drain_min ? (a.drain_min < 64u ? a.drain_min : 64u) : 64u;
    const uint32_t lane_off = (uint32_t)kk * 16 + (uint32_t)l15;   // this lane's float4 inside a 1 KiB operand block
    const uint32_t lane_b = lane_off * 16u;                       // ... in bytes
    // Short f16 rows (<= 4 K steps = 128 dims): with the MFMA time gone the tile is latency-bound, so ALL of
    // the next tile's operands (16 loads = 64 registers) and its row norms are requested right after the
    // current tile's MFMAs and fly during its screen / expansion / exact evaluation.
    constexpr bool CAN_PF = PF;
    constexpr bool pf = PF;                                      // (the launcher picks PF iff G / 4 <= 4)
    float4 xt[CAN_PF ? 4 : 1][4];
    float xn_pf[4] = {0.f, 0.f, 0.f, 0.f};
    auto issue_tile = [&](uint64_t tn) {
        const uint32_t nv = (r1 - tn < 64) ? (uint32_t)(r1 - tn) : 64u;
        const float4 *xb0 = nullptr;
        uint32_t so[4];
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            uint32_t rr = (uint32_t)(16 * t + l15);
            if (rr >= nv) rr = nv - 1;
            xn_pf[t] = a.row_norm2[lbeg + tn + rr];
            uint64_t T = blk0 + ((tn + 16 * t) >> 4);
            if (T > blk_last) T = blk_last;
            const float4 *xb = a.mat_blk + T * G * 16;
            if (t == 0) xb0 = xb;
            so[t] = (uint32_t)((xb - xb0) * 16);
        }
        const __amdgpu_buffer_rsrc_t r = operand_rsrc(xb0);
#pragma unroll
        for (int ks = 0; ks < (CAN_PF ? 4 : 1); ++ks)
            if ((uint32_t)ks < (G >> 2)) {
#pragma unroll
                for (int t = 0; t < 4; ++t) xt[ks][t] = buf_ld16(r, lane_b, so[t] + ks * 1024);
            }
    };
    if (pf && r0 < r1) issue_tile(r0);
    // the query thresholds are read one tile ahead (they tighten while the kernel runs, and a freshly
    // modified line costs a fabric round trip that must not sit in front of the operand waits)
    uint32_t gthr_next[QS];
#pragma unroll
    for (int s = 0; s < QS; ++s) gthr_next[s] = cur_gthr[s];
    // B-operand registers of the K loop (two ping-pong stages).  They persist across tiles: the loads of a tile's
    // first two K steps are issued behind the LAST MFMAs of the previous tile, so they fly during its screen /
    // expansion / exact evaluations instead of opening the K loop with a full memory round trip.
#ifndef PQV_XT
#define PQV_XT 1
#endif
#ifndef PQV_XPF
#define PQV_XPF 1
#endif
    constexpr int NS = TS == 2 ? PQV_NS_TS2 : (OP != OP_F32 && QLDS && !PF && NG <= 6 && NW == 8) ? PQV_NS_WIDE
                       : (OP == OP_I8 && QLDS && !PF && NW == 4) ? PQV_NS_REG : 2;      // operand stages in flight
    constexpr bool XPF = PQV_XPF;          // the next tile's first stages are requested before this tile's screen
    float4 xs[NS][TS];
    auto tile_desc = [&](uint64_t tn, uint32_t (&so)[TS]) {
        const float4 *b0 = nullptr;
#pragma unroll
        for (int t = 0; t < TS; ++t) {
            uint64_t T = blk0 + ((tn + 16 * t) >> 4);
            if (T > blk_last) T = blk_last;             // tiles past the list's end: masked by the screen
            const float4 *b = a.mat_blk + T * G * 16;
            if (t == 0) b0 = b;
            so[t] = (uint32_t)((b - b0) * 16);
        }
        return operand_rsrc(b0);
    };
    // (f32 operands keep the per-tile form: their kernels are built for three waves per SIMD and have no registers
    //  to carry two operand stages through the exact evaluations)
    constexpr bool XT = PQV_XT && QLDS && !PF && OP != OP_F32;
    auto prefetch_tile = [&](uint64_t tn) {          // the first NS operand stages of the tile at row tn
        uint32_t so[TS];
        const __amdgpu_buffer_rsrc_t r = tile_desc(tn, so);
#pragma unroll
        for (int j = 0; j < NS; ++j)
#pragma unroll
            for (int t = 0; t < TS; ++t) xs[j][t] = buf_ld16<ROW_AUX>(r, lane_b, so[t] + j * 1024);
    };
    // After exact evaluations in the middle of a wave's rows (rare: nearly all run behind the last tile) the prefetch is issued
    // AGAIN: the operand registers are then dead across the evaluation code -- its 8 + 8 row / query chunks per lane were the
    // kernel's register peak with the stages live through it, and what did not fit was spilled, with the reloads in the screen.
    constexpr bool REPF = XT && XPF && PQV_REPF;
    if constexpr (XT && XPF) {
        if (r0 < r1) prefetch_tile(r0);
    }
    [[maybe_unused]] int xn2i_next[TS] = {};
    [[maybe_unused]] float xres_next[TS] = {};
    // XTA: nothing is waited for between two K loops -- the thresholds of tile i + 1's screen and the row terms of tile i + 2
    // are requested behind K loop i, AHEAD of tile i + 1's operand stages in the (in-order) load queue, so they have landed
    // whenever an operand has.  A threshold is then one tile old when it screens (it only ever tightens: a few more
    // survivors), and the fabric round trip of an agent-scope load (~3 us under load, 19 % of a wave's time on 32-row
    // tiles) leaves the critical path.
    constexpr bool XTA = XT && I8 && (TS == 2 ? PQV_XTA_TS2 : PQV_XTA);
    [[maybe_unused]] int xn2i_nn[XTA ? TS : 1] = {};
    [[maybe_unused]] float xres_nn[XTA ? TS : 1] = {};
    // XTC (the 64-row-tile int8 instances): the same without the extra row-term registers -- after a K loop the next tile's
    // operand stages are issued FIRST, the thresholds (for the next tile's screen) and the next tile's row terms behind them in
    // the in-order queue; they are consumed at the next tile's top / after its K loop, when everything issued here has long
    // landed.  The plain form waited out a fabric round trip (thresholds + row terms, vmcnt(0)) between every two K loops
    // with nothing of the wave in flight: 13 % of a wave's life on C3 (tools/phases_now.sh).
    constexpr bool XTC = XT && I8 && !XTA && PQV_XTC;
    [[maybe_unused]] uint32_t gthr_pf[(XTA || XTC) ? QS : 1];
    // the wave's row terms (int8: |xi|^2 and the residual bound, from its first row r0) through wave-uniform descriptors: a
    // per-lane 64-bit pointer is a register pair the allocator spills, and its reload -- a VMEM load -- waits for every operand
    // stage issued before it
    [[maybe_unused]] __amdgpu_buffer_rsrc_t rt_n2i = operand_rsrc(a.mat_blk), rt_res = rt_n2i;
    if constexpr (I8) { rt_n2i = operand_rsrc(a.row_n2i + lbeg + r0); rt_res = operand_rsrc(a.row_res + lbeg + r0); }
    auto row_terms = [&](uint64_t tn, int (&n2)[TS], float (&rs)[TS]) {       // tile starting at row tn < r1 of the list
        const uint32_t nv = (r1 - tn < TROWS) ? (uint32_t)(r1 - tn) : TROWS;
        const uint32_t ub = (uint32_t)(tn - r0) * 4u;
#pragma unroll
        for (int t = 0; t < TS; ++t) {
            uint32_t rr = (uint32_t)(16 * t + l15);
            if (rr >= nv) rr = nv - 1;
            n2[t] = (int)buf_ld4(rt_n2i, rr * 4u, ub);
            rs[t] = __uint_as_float(buf_ld4(rt_res, rr * 4u, ub));
        }
    };
    if constexpr (XTA || XTC) {
#pragma unroll
        for (int s = 0; s < QS; ++s) gthr_pf[s] = cur_gthr[s];
    }
    if constexpr (I8) {
        if (r0 < r1) {
            row_terms(r0, xn2i_next, xres_next);
            if constexpr (XTA) {
                if (r0 + TROWS < r1) row_terms(r0 + TROWS, xn2i_nn, xres_nn);
            }
        }
    }
    for (uint64_t t0 = r0; t0 < r1; t0 += TROWS) {
#ifdef PQV_PROFILE_PHASES
        const uint64_t ph_top = __builtin_amdgcn_s_memtime();
#endif
        const uint32_t nvalid = (r1 - t0 < TROWS) ? (uint32_t)(r1 - t0) : TROWS;
        // B operands come from the BLOCKED copy of the lists (launch_block_rows): 16-row tile T,
        // 16-byte column ch, row j of the tile at float4 index (T * G + ch) * 16 + j -- the 64 lanes
        // (j = lane & 15, ch = k0 / 4 + lane >> 4) of one load read 1 KiB contiguous.  Tile bases are
        // wave-uniform (scalar registers); the lane offset is shared by all loads.
        const float4 *xbase[TS];
        float xn[TS];
#pragma unroll
        for (int t = 0; t < TS; ++t) {
            uint32_t rr = (uint32_t)(16 * t + l15);
            if (rr >= nvalid) rr = nvalid - 1;
            if constexpr (I8) xn[t] = 0.0f; else xn[t] = pf ? xn_pf[t] : a.row_norm2[lbeg + t0 + rr];
            uint64_t T = blk0 + ((t0 + 16 * t) >> 4);
            if (T > blk_last) T = blk_last;             // tiles past the list's end: masked below
            xbase[t] = a.mat_blk + T * G * 16;
        }
        // I8: the rows' integer norms and residual bounds feed the accumulators' start values, so they are fetched
        // one tile ahead (their latency would otherwise sit in front of the K loop)
        [[maybe_unused]] int xn2i[TS];
        [[maybe_unused]] float xres[TS];
        if constexpr (I8) {
#pragma unroll
            for (int t = 0; t < TS; ++t) { xn2i[t] = xn2i_next[t]; xres[t] = xres_next[t]; }
        }
        // one descriptor per tile (base = its first 16-row sub-tile); the other sub-tiles and the K steps
        // are scalar byte offsets (< 1 MiB)
        const __amdgpu_buffer_rsrc_t xr = operand_rsrc(xbase[0]);
        uint32_t xso[TS];
#pragma unroll
        for (int t = 0; t < TS; ++t) xso[t] = (uint32_t)((xbase[t] - xbase[0]) * 16);
        if constexpr (!XT) {
#pragma unroll
            for (int s = 0; s < QS; ++s) {
                cur_gthr[s] = gthr_next[s];
                gthr_next[s] = load_thr(s);
            }
        }

        using acc_t = std::conditional_t<I8, i32x4_acc, f32x4_acc>;
        acc_t acc[NG][TS];
        if constexpr (I8) {
            // int8 operands.  x = c + xi / S + e_x and q = c + qi / S + e_q (c = per-dimension mid-range, S one global
            // scale, xi / qi the int8 images, |e_x| <= rx and |e_q| <= rq stored upper bounds of the residual norms), so
            //     |q - x| >= |qi - xi| / S - rq - rx        (triangle inequality; |qi - xi|^2 = Nq + Nx - 2 qi.xi EXACTLY)
            // and a pair whose reference distance could still pass the threshold thr has
            //     |qi - xi| <= S (sqrt(thr (1 + c)) + rq + R),  R = the largest rx of the tile's rows
            //     =>  Nq + Nx - 2 dot <= T2 = S^2 (sqrt(thr (1 + c)) + rq + R)^2.
            // skip  <=>  dot + ceil(-Nx / 2) + ceil((T2 - Nq) / 2) < 0: the row term is the accumulator's START value
            // (known before the K loop, no threshold in it), the query term one integer add per pair after the loop,
            // the sign bit the answer.  All roundings go up (never skip wrongly); the contraction itself is exact.
#pragma unroll
            for (int t = 0; t < TS; ++t) {
                const int init = -(xn2i[t] >> 1);
#pragma unroll
                for (int g = 0; g < NG; ++g) acc[g][t] = (i32x4_acc){init, init, init, init};
            }
        } else {
#pragma unroll
            for (int g = 0; g < NG; ++g)
#pragma unroll
                for (int t = 0; t < TS; ++t) acc[g][t] = (f32x4_acc){0.f, 0.f, 0.f, 0.f};
        }

#ifdef PQV_PROFILE_PHASES
        const uint64_t ph_a = __builtin_amdgcn_s_memtime(); ph_top_sum += ph_a - ph_top;
#endif
        // K loop, two 16-dim steps per iteration with ping-pong operand registers: the loads of the
        // next step are in flight behind the 16 NG MFMAs of the current one.  Full quads (all NG
        // groups active) run a branch-free body, so the compiler's wait counts stay exact (with the
        // per-group branches it falls back to vmcnt(0) in front of every MFMA group, which serialises
        // the prefetch).
        const uint32_t nks = G >> 2;          // K steps (4 operand columns = 1 KiB per 16-row sub-tile each): a multiple of 4
        // A operands: one ds_read_b128 per group and K step.  Left to itself the compiler keeps ONE register quad for them
        // -- read, wait out the LDS latency, TS MFMAs, read ... -- which costs little behind four MFMAs but is most of a
        // K step behind two (the wide-quad instance: 1500 cycles per K step for 256 cycles of MFMA).  APD > 0: the read of
        // group g + APD is issued before the MFMAs of group g (rotating register quads; group indices past the quad's last
        // group are clamped to it, the staged part of the LDS).
        constexpr int APD = TS == 2 ? PQV_APD_TS2 : PQV_APD;
        // `full`: std::integral_constant<int, NA> -- NA > 0: exactly the first NA groups, branch-free (NA = NG: a full quad);
        // NA = 0: the quad's ng groups behind per-group branches
        auto mma = [&](const float4 (&x)[TS], uint32_t ks, auto full) {
            constexpr int NA = decltype(full)::value, NGL = NA ? NA : NG;
            const uint32_t chq = ks * 4 + (uint32_t)kk;
            if constexpr (APD == 0) {
#pragma unroll
                for (int g = 0; g < NGL; ++g) {
                    if (NA || (uint32_t)g < ng) {
                        const float4 qc = qs[(16 * g + l15) * G + (chq ^ (uint32_t)l15)];
#pragma unroll
                        for (int t = 0; t < TS; ++t) mfma_step<OP>(acc[g][t], qc, x[t]);
                    }
                }
            } else {
                const float4 *qb0 = qs + (uint32_t)l15 * G + (chq ^ (uint32_t)l15);
                const uint32_t gstride = 16u * G;
                float4 qb[APD + 1];
#pragma unroll
                for (int g = 0; g < APD && g < NGL; ++g) {
                    const uint32_t gi = NA || (uint32_t)g < ng ? (uint32_t)g : ng - 1u;
                    qb[g] = qb0[gi * gstride];
                }
#pragma unroll
                for (int g = 0; g < NGL; ++g) {
                    if (NA || (uint32_t)g < ng) {
                        if (g + APD < NGL) {
                            const uint32_t gi = NA || (uint32_t)(g + APD) < ng ? (uint32_t)(g + APD) : ng - 1u;
                            qb[(g + APD) % (APD + 1)] = qb0[gi * gstride];
                        }
#pragma unroll
                        for (int t = 0; t < TS; ++t) mfma_step<OP>(acc[g][t], qb[g % (APD + 1)], x[t]);
                        // (the machine scheduler otherwise sinks every read back in front of its use to save the registers)
                        __builtin_amdgcn_sched_barrier(0);
                    }
                }
            }
        };
        auto kloop = [&](auto full) {
            if constexpr (!(XT && XPF)) {
#pragma unroll
                for (int j = 0; j < NS; ++j)
#pragma unroll
                    for (int t = 0; t < TS; ++t) xs[j][t] = buf_ld16<ROW_AUX>(xr, lane_b, xso[t] + j * 1024);
            }
            // on entry xs[j] holds (or awaits) K step j of this tile; nks is a multiple of 4 >= NS
            uint32_t ks = 0;
            for (; ks + NS < nks; ks += NS) {
#pragma unroll
                for (int j = 0; j < NS; ++j) {
                    mma(xs[j], ks + j, full);
#pragma unroll
                    for (int t = 0; t < TS; ++t) xs[j][t] = buf_ld16<ROW_AUX>(xr, lane_b, xso[t] + (ks + j + NS) * 1024);
                }
            }
#pragma unroll
            for (int j = 0; j < NS; ++j) mma(xs[j], ks + j, full);
        };
        // !QLDS: both operands stream from global memory through THREE rotating register stages, so the
        // loads of K step s + 2 are issued before the MFMAs of step s (HBM latency is ~2 K steps of a
        // wave that shares the matrix pipe).  Indices past the last step are clamped: branch-free, the
        // compiler's wait counts stay exact.
        auto kloop_gq = [&]() {
            float4 x0[4], x1[4], x2[4], q0[NG], q1[NG], q2[NG];
            auto ld = [&](float4 (&x)[4], float4 (&q)[NG], uint32_t ks) {
                const uint32_t kc = ks < nks ? ks : nks - 1;
                const uint32_t off = kc * 1024;                      // 16 dims = 4 columns = 1 KiB (uniform)
#pragma unroll
                for (int t = 0; t < 4; ++t) x[t] = buf_ld16(xr, lane_b, xso[t] + off);
#pragma unroll
                for (int g = 0; g < NG; ++g) q[g] = buf_ld16(qr, lane_b, (uint32_t)g * G * 256 + off);
            };
            auto mmag = [&](const float4 (&x)[4], const float4 (&q)[NG]) {
#pragma unroll
                for (int g = 0; g < NG; ++g) {
                    if ((uint32_t)g < ng) {
#pragma unroll
                        for (int t = 0; t < 4; ++t) mfma_step<OP>(acc[g][t], q[g], x[t]);
                    }
                }
            };
            ld(x0, q0, 0);
            ld(x1, q1, 1);
            uint32_t ks = 0;
            for (; ks + 3 <= nks; ks += 3) {
                ld(x2, q2, ks + 2); mmag(x0, q0);
                ld(x0, q0, ks + 3); mmag(x1, q1);
                ld(x1, q1, ks + 4); mmag(x2, q2);
            }
            if (ks < nks) mmag(x0, q0);
            if (ks + 1 < nks) mmag(x1, q1);
        };
        if constexpr (!QLDS) { if constexpr (TS == 4) kloop_gq(); }
        else if constexpr (PF) {
#pragma unroll
            for (int ks = 0; ks < 4; ++ks)
                if ((uint32_t)ks < nks) mma(xt[ks], (uint32_t)ks, std::integral_constant<int, 0>{});
        }
        // (32-row tiles: a branch-free body per group count -- a wide quad has 7..10 of its 10 groups; per-group branches would
        //  cost the exact wait counts of the A-operand pipeline, and running all ten always (round 3 / 4) costs up to 30 % of the
        //  K loop's MFMAs where two waves share a SIMD's matrix pipe and full quads keep it ~75 % busy inside the loops)
        else if constexpr (TS == 2 && NG == 10) {
            if (ng >= 10u) kloop(std::integral_constant<int, 10>{});
            else if (ng == 9u) kloop(std::integral_constant<int, 9>{});
            else if (ng == 8u) kloop(std::integral_constant<int, 8>{});
            else kloop(std::integral_constant<int, 7>{});
        }
        else if constexpr (PQV_NA_REG && TS == 4 && NG == 6 && I8 && NW == 4) {
            // (the regular int8 instance: most quads of a batch have 3..5 of their 6 groups -- C3: 51 queries on average)
            if (ng >= 6u) kloop(std::integral_constant<int, 6>{});
            else if (ng == 5u) kloop(std::integral_constant<int, 5>{});
            else if (ng == 4u) kloop(std::integral_constant<int, 4>{});
            else if (ng == 3u) kloop(std::integral_constant<int, 3>{});
            else kloop(std::integral_constant<int, 0>{});
        }
        else if (ng == (uint32_t)NG || TS == 2) kloop(std::integral_constant<int, NG>{});
        else kloop(std::integral_constant<int, 0>{});
#ifdef PQV_PROFILE_PHASES
        const uint64_t ph_x0 = __builtin_amdgcn_s_memtime();
#endif
#if PQV_RELANE
        // The lane index is handed back through an opaque (empty) asm after every K loop: everything the screen, the expansion
        // and the exact evaluations derive from it -- validity masks, LDS addresses, prefix masks: loop invariants the compiler
        // otherwise hoists out of the tile loop and keeps in ~80 registers ACROSS the K loop, spilling what does not fit -- is
        // recomputed per tile (a few dozen VALU operations against ~35 k cycles) and the K loop keeps its registers.
        asm volatile("" : "+v"(lane));
        l15 = lane & 15; kk = lane >> 4;
#endif
        if constexpr (XT) {
            // Between the K loops NOTHING this wave loads may be consumed while operand prefetches are in flight: loads
            // return in order, so waiting for a fresh one drains the whole queue (and a register the allocator spills
            // right after its load does exactly that).  Hence, in this order: (1) the thresholds of this tile's screen
            // and the next tile's row terms are requested and waited for while the queue is empty anyway -- an L2 round
            // trip, and the thresholds are as fresh as they can be; (2) only then the next tile's first operand stages
            // go out, to fly during the screen, the expansion and the exact evaluations.
            const uint64_t tn = t0 + TROWS;
            if constexpr (XTA) {
#pragma unroll
                for (int s = 0; s < QS; ++s) {
                    cur_gthr[s] = gthr_pf[s];
                    gthr_pf[s] = load_thr(s);
                }
#pragma unroll
                for (int t = 0; t < TS; ++t) { xn2i_next[t] = xn2i_nn[t]; xres_next[t] = xres_nn[t]; }
                if (tn + TROWS < r1) row_terms(tn + TROWS, xn2i_nn, xres_nn);
            } else if constexpr (XTC) {
#pragma unroll
                for (int s = 0; s < QS; ++s) cur_gthr[s] = gthr_pf[s];
            } else {
                // (PQV_THR_EVERY = n: the thresholds are re-read behind every n-th tile only -- a scattered agent-scope load
                //  fetches a 32-byte sector per 4-byte threshold: 3 KB per tile and wave against its 48 KB of rows)
                if (PQV_THR_EVERY <= 1 || (((uint32_t)((t0 - r0) / TROWS)) % (uint32_t)PQV_THR_EVERY) == 0u || tn >= r1) {
#pragma unroll
                    for (int s = 0; s < QS; ++s) cur_gthr[s] = load_thr(s);
                }
            }
            if constexpr (I8 && !XTA && !XTC) {
                if (tn < r1) row_terms(tn, xn2i_next, xres_next);
            }
            if constexpr (!XTA && !XTC) __builtin_amdgcn_s_waitcnt(0x0F70);         // vmcnt(0): (1) has landed before (2) is issued
            __builtin_amdgcn_sched_barrier(0);
            if (XPF && tn < r1) prefetch_tile(tn);
            __builtin_amdgcn_sched_barrier(0);
            if constexpr (XTC) {
#pragma unroll
                for (int s = 0; s < QS; ++s) gthr_pf[s] = load_thr(s);
                if (tn < r1) row_terms(tn, xn2i_next, xres_next);
                __builtin_amdgcn_sched_barrier(0);
            }
        }

#ifdef PQV_PROFILE_PHASES
        const uint64_t ph_b = __builtin_amdgcn_s_memtime(); ph_k += ph_b - ph_a; ph_xt += ph_b - ph_x0;
#endif
        // Screen.  skip  <=>  lb > thr  <=>  d~ > (thr + 2 c nn) / (1 - c)  <=>  s < smin, with
        //     smin = (nn - (thr + 2 c nn) / (1 - c)) / 2 = (alpha |q|^2 - beta thr) + alpha |x|^2
        // (d~ = nn - 2 s; for d~ < 0 the bound is negative and never skips either way): one add and one
        // compare of the raw accumulator per pair.  The roundings of smin (a few u nn) come out of the
        // 4x safety factor of c (>= 64 u).  Invalid rows / queries get +inf (always skipped), an EMPTY
        // threshold is NaN (never skipped).  C/D layout: col j = lane & 15 (row 16 t + j of the tile),
        // row i = kk * 4 + r (query i of group g).
        // Every lane collects the keep-bits of ITS 16 NG pairs (g, r, t) in one register per two groups:
        // bits = 2 bits + keep (v_addc with the compare's carry) -- three VALU ops per pair, no
        // branches, and the accumulators die here: the expansion of the bits into queue entries and the
        // exact evaluation below do not have to share registers with them.
        float bt[TS];
#pragma unroll
        for (int t = 0; t < TS; ++t) bt[t] = (uint32_t)(16 * t + l15) < nvalid ? alpha * xn[t] : INFINITY;
        // per-query terms through LDS: lane q publishes a_q, then every lane reads the four values of its
        // kk for each group as one 16-byte load
        if constexpr (I8) {
            float R = 0.0f;
#pragma unroll
            for (int t = 0; t < TS; ++t) R = fmaxf(R, (uint32_t)(16 * t + l15) < nvalid ? xres[t] : 0.0f);
#pragma unroll
            for (int off = 8; off > 0; off >>= 1) R = fmaxf(R, __shfl_xor(R, off, 64));
            wave_lds_fence();
#pragma unroll
            for (int s = 0; s < QS; ++s) {
                const uint32_t qi = 64u * (uint32_t)s + (uint32_t)lane;
                const float thr_d = __uint_as_float(cur_gthr[s]);
                const float qres = qst_res[qi];
                const bool open = !(qres <= 3.0e38f) || cur_gthr[s] == 0xFFFFFFFFu || !(thr_d <= 3.0e38f);
                int a2 = 1 << 29;                              // never skip
                if (qi >= cnt) a2 = -(1 << 30);                // not a query of this quad: always "skipped"
                else if (!open) {
                    const float v = lscale * (sqrtf(thr_d * (1.0f + 4.0f * cmargin)) * 1.000002f + qres + R);
                    const float v2 = fminf(v * v * 1.000002f, 1.0e9f);
                    a2 = ((int)ceilf(v2) + 1 - qst_n2i[qi] + 1) >> 1;
                }
                if (qi < NQ) reinterpret_cast<int *>(aq)[qi] = a2;
            }
            wave_lds_fence();
        } else {
            wave_lds_fence();
#pragma unroll
            for (int s = 0; s < QS; ++s) {
                const uint32_t qi = 64u * (uint32_t)s + (uint32_t)lane;
                // threshold DISTANCE of this lane's query; KEY_EMPTY: "cannot skip"
                const float thr_d = __uint_as_float(cur_gthr[s]);
                float qn;
                bool noskip;
                if constexpr (LST) { qn = qst_qn[qi < NQ ? qi : 0]; noskip = !(qn == qn); }
                else { qn = my_qn[s]; noskip = my_noskip[s]; }
                if (qi < NQ) aq[qi] = qi >= cnt ? INFINITY : (noskip || cur_gthr[s] == 0xFFFFFFFFu) ? -3.0e38f : alpha * qn - beta * thr_d;
            }
            wave_lds_fence();
        }
        // keep-bits of this lane's pairs: FW = 4 TS bits per group -- (r, t) at bit FW - 1 - (TS r + t) of the group's field --
        // GPW groups per word, the first group of a word in its highest field
        uint32_t bits[NWD];
#pragma unroll
        for (int w = 0; w < NWD; ++w) bits[w] = 0;
        if constexpr (I8) {
            // skip <=> acc + A2 < 0: one integer add per pair, the sign bit shifted into the lane's mask
#pragma unroll
            for (int g = 0; g < NG; ++g) {
                const int4 a4 = *reinterpret_cast<const int4 *>(reinterpret_cast<const int *>(aq) + 16 * g + 4 * kk);
                const int ar[4] = {a4.x, a4.y, a4.z, a4.w};
#pragma unroll
                for (int r = 0; r < 4; ++r) {
#pragma unroll
                    for (int t = 0; t < TS; ++t)
                        bits[g / (int)GPW] = __builtin_amdgcn_alignbit(bits[g / (int)GPW], (uint32_t)(acc[g][t][r] + ar[r]), 31);
                }
            }
#pragma unroll
            for (int w = 0; w < NWD; ++w) bits[w] = ~bits[w];      // sign bits say "skip"
        } else if constexpr (F16) {
            // f16 operands: every term is finite by construction (rows scaled below 2^14, query images clamped
            // to the f16 range, never-skip / unset thresholds carry -3e38, invalid ones +inf), so
            // skip <=> acc - smin < 0 <=> its sign bit: two packed adds per TWO pairs and one v_alignbit
            // per pair shift the sign into the lane's mask -- no compare, no scalar mask, no select.
            typedef float f32x2_t __attribute__((ext_vector_type(2)));
#pragma unroll
            for (int g = 0; g < NG; ++g) {
                const float4 a4 = *reinterpret_cast<const float4 *>(aq + 16 * g + 4 * kk);
#pragma unroll
                for (int rp = 0; rp < 2; ++rp) {
                    const f32x2_t ar2 = rp ? f32x2_t{a4.z, a4.w} : f32x2_t{a4.x, a4.y};
                    f32x2_t dd[TS];
#pragma unroll
                    for (int t = 0; t < TS; ++t) {
                        const f32x2_t smin = ar2 + f32x2_t{bt[t], bt[t]};
                        const f32x2_t av = {acc[g][t][2 * rp], acc[g][t][2 * rp + 1]};
                        dd[t] = av - smin;
                    }
#pragma unroll
                    for (int rr = 0; rr < 2; ++rr) {
#pragma unroll
                        for (int t = 0; t < TS; ++t)
                            bits[g / (int)GPW] = __builtin_amdgcn_alignbit(bits[g / (int)GPW], __float_as_uint(dd[t][rr]), 31);
                    }
                }
            }
#pragma unroll
            for (int w = 0; w < NWD; ++w) bits[w] = ~bits[w];      // sign bits say "skip"
        } else {
#pragma unroll
            for (int g = 0; g < NG; ++g) {
                const float4 a4 = *reinterpret_cast<const float4 *>(aq + 16 * g + 4 * kk);
                const float ar[4] = {a4.x, a4.y, a4.z, a4.w};
#pragma unroll
                for (int r = 0; r < 4; ++r) {
#pragma unroll
                    for (int t = 0; t < TS; ++t) {
                        const bool keep = !(acc[g][t][r] < ar[r] + bt[t]);
                        bits[g / (int)GPW] = bits[g / (int)GPW] + bits[g / (int)GPW] + (keep ? 1u : 0u);
                    }
                }
            }
        }
        // (a last word that holds fewer than GPW groups: its fields move up to where a full word has them)
        if constexpr ((NG % (int)GPW) != 0) bits[NWD - 1] <<= FW * (GPW - (uint32_t)(NG % (int)GPW));
        uint32_t rowmask = 0;                          // bit TS - 1 - t: row 16 t + l15 of the tile belongs to this wave
#pragma unroll
        for (int t = 0; t < TS; ++t) rowmask |= ((uint32_t)(16 * t + l15) < nvalid) ? ((1u << (TS - 1)) >> t) : 0u;
        // the accumulators are dead from here on: the next tile's operands can take their registers
        if constexpr (PF) { if (pf && t0 + 64 < r1) issue_tile(t0 + 64); }
        const uint32_t rowbase = (uint32_t)(t0 - r0) + (uint32_t)l15;
        // Fast path (almost every tile): all survivors of the tile fit the queue at once -- ONE prefix scan and
        // one drain per tile.  Otherwise the bits are expanded a pass (BP of a lane's FW pair bits per group) at a
        // time with the drain in between; after the last tile one extra pass flushes the queue.
        // Explicit validity (rows past the wave's range, queries past the quad's count): the compare above keeps
        // a pair whenever its operands are NaN -- an unset threshold, non-finite data -- and an out-of-range row
        // must never reach the exact evaluation.
        uint32_t vw[NWD];
        uint32_t tot = 0;
#pragma unroll
        for (int ww = 0; ww < NWD; ++ww) {
            uint32_t vmw = 0;
#pragma unroll
            for (int gl = 0; gl < (int)GPW; ++gl) {
                const uint32_t g = GPW * (uint32_t)ww + (uint32_t)gl;
                if (g < (uint32_t)NG) {
                    uint32_t vm = 0;
#pragma unroll
                    for (int r = 0; r < 4; ++r) vm |= (16 * g + 4 * (uint32_t)kk + (uint32_t)r < cnt) ? rowmask << (FW - TS * (r + 1)) : 0u;
                    vmw |= g < ng ? vm << (FW * (GPW - 1u - (uint32_t)gl)) : 0u;
                }
            }
            vw[ww] = bits[ww] & vmw;
            tot += (uint32_t)__popc(vw[ww]);
        }
        const uint32_t incl_all = wave_incl_scan_u32(tot);
        const bool one_pass = readlane_u32(incl_all, 63) <= (uint32_t)PASS - 64u;
        const bool last_tile = t0 + TROWS >= r1;
        if (one_pass) {
            uint32_t at = npend + incl_all - tot;
            if (defer) {
                // the raw scores of the survivors, in the order the expansion below numbers them (word, group, r, t): a static
                // sweep over the accumulators, a group at a time and only where some lane has a survivor
                uint32_t atv = at;
#pragma unroll
                for (int g = 0; g < NG; ++g) {
                    const uint32_t field = (vw[g / (int)GPW] >> (FW * (GPW - 1u - (uint32_t)(g % (int)GPW)))) & ((1u << FW) - 1u);
                    if (__ballot(field != 0u) != 0ull) {
#pragma unroll
                        for (int r = 0; r < 4; ++r) {
#pragma unroll
                            for (int t = 0; t < TS; ++t) {
                                if (field & (1u << (FW - 1u - (uint32_t)(TS * r + t)))) {
                                    uint32_t rawv;
                                    if constexpr (I8) rawv = (uint32_t)acc[g][t][r]; else rawv = __float_as_uint(acc[g][t][r]);
                                    if constexpr (!I8) { if (rawv == NOVAL_NOHIST) rawv = NOVAL; }
                                    pv[atv++] = rawv;
                                }
                            }
                        }
                    }
                }
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            }
#pragma unroll
            for (int ww = 0; ww < NWD; ++ww) {
                uint32_t mm = vw[ww];
                while (mm) {
                    const uint32_t b = 31u - (uint32_t)__clz(mm);
                    mm &= ~(1u << b);
                    const uint32_t cc = 31u - b;                         // cc = FW gl + TS r + t
                    const uint32_t gl = cc / FW, rt = cc % FW;
                    const uint32_t qslot = 16u * (GPW * (uint32_t)ww + gl) + 4u * (uint32_t)kk + rt / TS;
                    pend[at++] = (qslot << QSH) + rowbase + 16u * (rt % TS);
                }
            }
            npend += readlane_u32(incl_all, 63);
#ifdef PQV_PROFILE_PHASES
            const uint64_t ph_c = __builtin_amdgcn_s_memtime();
#endif
            const bool ran = drain(last_tile ? 1u : drain_min);
            if constexpr (REPF) { if (ran && !last_tile) prefetch_tile(t0 + TROWS); }
#ifdef PQV_PROFILE_PHASES
            ph_e += __builtin_amdgcn_s_memtime() - ph_c;
#endif
        }
        // slow path: passes of BP of a lane's FW pair bits (cc = TS r + t) per group
        constexpr uint32_t BP = (uint32_t)PASS / 64u < FW ? (uint32_t)PASS / 64u : FW;
        constexpr uint32_t PPG = FW / BP;                    // passes per group
        const uint32_t hend = one_pass ? 0u : PPG * ng + (last_tile ? 1u : 0u);
        [[maybe_unused]] bool ran_slow = false;
#pragma unroll 1
        for (uint32_t hg = 0; hg < hend; ++hg) {
            const uint32_t g = hg / PPG, ps = hg % PPG;
            if (g < ng) {
                uint32_t w = vw[0];
#pragma unroll
                for (int ww = 1; ww < NWD; ++ww) w = (g / GPW) == (uint32_t)ww ? vw[ww] : w;
                uint32_t mm = (w >> (FW * (GPW - 1u - g % GPW))) & ((1u << FW) - 1u);
                // cc = TS r + t lives in bit FW - 1 - cc: pass ps takes cc in [ps BP, ps BP + BP)
                mm &= ((((1u << BP) - 1u) << (FW - BP)) & ((1u << FW) - 1u)) >> (BP * ps);
                const uint32_t cntl = (uint32_t)__popc(mm);
                const uint32_t incl = wave_incl_scan_u32(cntl);
                uint32_t at = npend + incl - cntl;
                const uint32_t qb = (16 * g + 4 * (uint32_t)kk) << QSH;
                while (mm) {
                    const uint32_t b = 31u - (uint32_t)__clz(mm);        // highest set bit first
                    mm &= ~(1u << b);
                    const uint32_t cc = FW - 1u - b;                     // cc = TS r + t
                    if (defer) pv[at] = NOVAL;                           // (a tile with this many survivors: evaluated here)
                    pend[at++] = qb + ((cc / TS) << QSH) + rowbase + 16u * (cc % TS);
                }
                npend += readlane_u32(incl, 63);
            }
#ifdef PQV_PROFILE_PHASES
            const uint64_t ph_c = __builtin_amdgcn_s_memtime();
#endif
            ran_slow = drain(g == ng ? 1u : 64u) || ran_slow;   // g == ng only in the flush pass
#ifdef PQV_PROFILE_PHASES
            ph_e += __builtin_amdgcn_s_memtime() - ph_c;
#endif
        }
        if constexpr (REPF) { if (ran_slow && !last_tile) prefetch_tile(t0 + TROWS); }
#ifdef PQV_PROFILE_PHASES
        ph_s += __builtin_amdgcn_s_memtime() - ph_b;
#endif
    }
    if (a.stats && lane == 0) {
#ifdef PQV_PROFILE_PHASES
        atomicAdd(&a.stats[0], (unsigned long long)(r1 - r0) * cnt);
        atomicAdd(&a.stats[1], (unsigned long long)n_exact);
#else
        // 64 counter pairs, one cache line apart (STATS_SLOTS; the host sums them): thousands of waves adding to
        // ONE line serialise at the memory side
        unsigned long long *st = a.stats + 8 + 16 * ((blockIdx.y * gridDim.x + blockIdx.x + (uint32_t)wave * 17u) % STATS_SLOTS);
        atomicAdd(&st[0], (unsigned long long)(r1 - r0) * cnt);
        atomicAdd(&st[1], (unsigned long long)n_exact);
        if (n_defer) atomicAdd(&st[4], (unsigned long long)n_defer);
#endif
#ifdef PQV_PROFILE_PHASES
        {   // per-wave record: [start, prologue, kloop, screen, drain, end, rows, cnt] at stats[8 + 8 * wave id]
            const unsigned long long wid = atomicAdd(&a.stats[6], 1ull);
            if (wid < 65536ull) {
                unsigned long long *rec = a.stats + 8 + 8 * wid;
                rec[0] = ph_t0; rec[1] = ph_pro | (ph_top_sum << 24); rec[2] = ph_k | (ph_xt << 32); rec[3] = ph_s - ph_e; rec[4] = ph_e;
                rec[5] = __builtin_amdgcn_s_memtime(); rec[6] = ph_em; rec[7] = cnt | ((unsigned long long)n_exact << 32);
            }
        }
#endif
    }
    PQV_STAMP_MAX(17);
}

template <int S>
static hipError_t launch_tile_s(const TileArgs &a, hipStream_t s) {
    dim3 grid(a.grid_x, a.max_groups), block(256);
    if ((a.dim % 4) == 0) hipLaunchKernelGGL((tile_rerank_kernel<TILE_QB, S, true>), grid, block, 0, s, a);
    else hipLaunchKernelGGL((tile_rerank_kernel<TILE_QB, S, false>), grid, block, 0, s, a);
    return hipGetLastError();
}

// gthr[q] = min(gthr[q], k-th smallest key over the seed lists of q)
template <int S>
__global__ __launch_bounds__(64) void seed_threshold_kernel(const uint64_t *part_keys, uint32_t nprobe,
                                                           uint32_t slots_per_pair, uint32_t k,
                                                           unsigned long long *gthr) {
    const int lane = threadIdx.x;
    const uint32_t q = blockIdx.x;
    const uint32_t n_part = nprobe * slots_per_pair;
    WaveTopk<S> tk;
    tk.init();
    const uint32_t total = nprobe * 4 * k;      // entries of q's seed lists
    for (uint32_t i = 0; i < total; i += 64) {
        const uint32_t idx = i + lane;
        uint64_t key = KEY_EMPTY;
        if (idx < total) {
            const uint32_t list = idx / k, e = idx % k;          // list = j * 4 + wave
            const uint32_t j = list >> 2, w = list & 3;
            key = part_keys[((uint64_t)q * n_part + j * slots_per_pair + w) * k + e];
        }
        tk.offer(key, 0u, k, lane);
    }
    const uint64_t kth = tk.kth(k);
    if (lane == 0 && kth != KEY_EMPTY) atomicMin(&gthr[q], (unsigned long long)kth);
}
hipError_t launch_seed_threshold(const uint64_t *part_keys, uint32_t nq, uint32_t nprobe, uint32_t slots_per_pair,
                                 uint32_t k, unsigned long long *gthr, hipStream_t s) {
    if (nq == 0) return hipSuccess;
    if (k <= 64) hipLaunchKernelGGL(seed_threshold_kernel<1>, dim3(nq), dim3(64), 0, s, part_keys, nprobe, slots_per_pair, k, gthr);
    else if (k <= 256) hipLaunchKernelGGL(seed_threshold_kernel<4>, dim3(nq), dim3(64), 0, s, part_keys, nprobe, slots_per_pair, k, gthr);
    else return hipErrorInvalidValue;
    return hipGetLastError();
}

// dynamic LDS beyond 64 KB has to be allowed per kernel once
template <int NG, int NW, int S, bool QLDS, int OP, bool PF = false, bool ONCE = false, int TS = 4, bool DEFP = false>
static hipError_t launch_wide(const TileArgs &a, size_t lds, hipStream_t s) {
    auto kern = wide_filter_kernel<NG, NW, S, QLDS, OP, PF, ONCE, TS, DEFP>;
    if (a.cand_lb && !(S > 1 || DEFP)) return hipErrorInvalidValue;      // this instance would ignore the request
    if (a.rows_per_block / NW + 64u >= (1u << 23)) return hipErrorInvalidValue;       // queue entries: 23 bits of row offset per wave
    if (lds > 65536) {          // raise the kernel's dynamic-LDS ceiling to what this launch needs (static + dynamic <= 160 KB)
        static std::atomic<size_t> allowed{65536};
        if (lds > allowed.load(std::memory_order_relaxed)) {
            const hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
            if (e != hipSuccess) { (void)hipGetLastError(); return e; }
            allowed.store(lds, std::memory_order_relaxed);
        }
    }
    // Workgroups go to the 8 XCDs round-robin in linear order, and the lists are very unequal (C3: 1 .. 44 k rows around
    // a mean of 9.8 k), so most quads use only the first few of the grid's row chunks: with an even grid width the
    // chunk index decides the XCD and some XCDs get most of the work (C4 at 32 chunks: 4.29 ms against 2.76 ms at 19).
    // An odd width makes consecutive quads start on different XCDs; the extra column exits at once.
    if (a.item_quad) hipLaunchKernelGGL(kern, dim3(a.max_items), dim3(64 * NW), lds, s, a);
    else hipLaunchKernelGGL(kern, dim3(a.grid_x | 1u, a.max_quads), dim3(64 * NW), lds, s, a);
    return hipGetLastError();
}

// LDS the wide kernel needs for a quad of `width` queries (f16 images, + the f32 originals for rows of <= 128 dims)
static size_t wide_lds_bytes(uint32_t width, uint32_t dim, bool f16, bool *q32) {
    const size_t q16 = (size_t)width * dim * 2, q32b = (size_t)width * dim * 4;
    if (!f16) { *q32 = false; return q32b; }
    *q32 = dim <= 128;
    return *q32 ? q16 + q32b : q16;
}

template <int S>
static hipError_t launch_filter_s(const TileArgs &a, hipStream_t s) {
#ifdef PQV_DEV_C3_ONLY      // tools/regs_c3.sh: only C3's two instances are instantiated (register / ISA checks in seconds, never shipped)
#ifndef PQV_DEV_WIDE_TS
#define PQV_DEV_WIDE_TS 2
#endif
#ifndef PQV_DEV_WIDE_NW
#define PQV_DEV_WIDE_NW 8
#endif
#ifndef PQV_DEV_WIDE_NG
#define PQV_DEV_WIDE_NG 10
#endif
#ifdef PQV_DEV_C2
    if (a.dim) { hipError_t e = launch_wide<6, 4, 1, true, OP_F16, true>(a, 0, s); return e != hipSuccess ? e : launch_wide<6, 4, 1, true, OP_F16, false>(a, 0, s); }
#endif
    if (a.dim) { hipError_t e = launch_wide<6, 4, 1, true, OP_I8, false, true>(a, 0, s); return e != hipSuccess ? e : launch_wide<PQV_DEV_WIDE_NG, PQV_DEV_WIDE_NW, 1, true, OP_I8, false, true, PQV_DEV_WIDE_TS>(a, 0, s); }
#else
    if (a.filter_variant == 0) {
        if ((a.dim % 64) != 0 || a.max_quads == 0 || !a.mat_blk || (a.row_of && !a.norm_by_pos) || !a.cand_keys) return hipErrorInvalidValue;
        const uint32_t nw = a.block_waves ? a.block_waves : 4;
        if (a.i8) {           // int8 images: 8-wave blocks, up to 128 queries x dim bytes of LDS
            if ((a.dim % 256) != 0 || !a.q_i8 || !a.q_n2i || !a.q_res || !a.list_scale || !a.row_n2i || !a.row_res) return hipErrorInvalidValue;
            // (only the groups of 16 queries a quad really has are staged or read -- a batch of <= 16 queries asks for a
            // quarter of the LDS and twice as many blocks fit a CU: a one-query call streams its lists with 16 waves per CU)
            const uint32_t live_w = a.nq < a.quad_width ? (a.nq + 15u) / 16u * 16u : a.quad_width;
            const size_t lds = (size_t)live_w * a.dim;
            if ((size_t)a.quad_width * a.dim > 147456) return hipErrorInvalidValue;
            if (nw == 4) {        // two 4-wave blocks per CU
                if (a.quad_width == 64 && 64ull * a.dim <= 65536)
                    return a.nq <= 64u ? launch_wide<4, 4, S, true, OP_I8, false, true>(a, lds, s)      // one quad per list: rows read once
                                       : launch_wide<4, 4, S, true, OP_I8>(a, lds, s);
                if (a.quad_width == 96 && 96ull * a.dim <= 73728) {
                    if (a.wide_width) {
                        // lists that 97..160 queries of the batch probe: ONE quad on 32-row tiles, one 8-wave block per CU --
                        // every row of such a list is read once instead of twice.
                        if ((a.wide_width != 160 && !a.list_once) || !a.item_quad || !a.wide_item_quad || !a.wide_max_items || !a.wide_rows_per_block)
                            return hipErrorInvalidValue;
                        TileArgs w = a;
                        w.quad_width = a.wide_width; w.block_waves = 8; w.wide_width = 0; w.pendv = a.pendv_wide;
                        w.item_quad = a.wide_item_quad; w.item_chunk = a.wide_item_chunk; w.n_items = a.wide_n_items; w.max_items = a.wide_max_items;
                        w.rows_per_block = a.wide_rows_per_block;
                        if (a.list_once) {
                            // round 6: the popular lists' rows stationary, their pairs streamed (kernels_list.hip) -- the regular instance
                            // first, as below: every query's thresholds have met most of its lists before the popular ones are read
                            if (a.cand_lb) return hipErrorInvalidValue;
                            w.block_waves = 4; w.item_chunk = nullptr;       // (its table is list-major: PairSortArgs::wide_list_major)
                            hipError_t e = launch_wide<6, 4, S, true, OP_I8, false, true>(a, lds, s);
                            return e != hipSuccess ? e : launch_list_filter(w, s);
                        }
                        // the regular instance first: most lists are its, so every query's thresholds have met most of its lists'
                        // first chunks before the popular lists are read (C3: 257 -> 225 exact evaluations per query, the serial
                        // step's kernels 2.09 -> 2.065 ms against the wide instance first)
                        // Cache policy of the row streams.  A list has ONE quad in the regular table: its rows are read once by that
                        // launch and stream with the nt policy (they do not displace query images, thresholds and survivors' rows
                        // from L2 / the Infinity Cache).  In the wide table a list of more than 160 pairs has several quads, which run
                        // side by side and share its rows through the caches -- nt pays there only when such lists are rare: the
                        // caller decides from the previous batch's counts (TileArgs::wide_nt).  Measured, nt on regular / wide /
                        // both: C3 +0.5 / +3.5 / +4 %, mixture +2.5 / -6 / -4 %.
                        if constexpr (S == 1) if (a.cand_lb) {       // short lists of long rows: the deferred form (defer_on decides)
                            hipError_t e = launch_wide<6, 4, S, true, OP_I8, false, true, 4, true>(a, lds, s);
                            if (e != hipSuccess) return e;
                            return a.wide_nt ? launch_wide<10, 8, S, true, OP_I8, false, true, 2, true>(w, (size_t)a.wide_width * a.dim, s)
                                             : launch_wide<10, 8, S, true, OP_I8, false, false, 2, true>(w, (size_t)a.wide_width * a.dim, s);
                        }
                        if (a.side_stream) {
                            // fork: the wide-quad launch on the side stream, behind everything the call has enqueued so far
                            hipError_t e = hipEventRecord(a.ev_fork, s);
                            if (e == hipSuccess) e = hipStreamWaitEvent(a.side_stream, a.ev_fork, 0);
                            hipStream_t s_reg = a.fork_wide_first ? a.side_stream : s, s_wide = a.fork_wide_first ? s : a.side_stream;
                            if (e == hipSuccess && a.fork_wide_first)
                                e = a.wide_nt ? launch_wide<10, 8, S, true, OP_I8, false, true, 2>(w, (size_t)a.wide_width * a.dim, s_wide)
                                              : launch_wide<10, 8, S, true, OP_I8, false, false, 2>(w, (size_t)a.wide_width * a.dim, s_wide);
                            if (e == hipSuccess) e = launch_wide<6, 4, S, true, OP_I8, false, true>(a, lds, s_reg);
                            if (e == hipSuccess && !a.fork_wide_first)
                                e = a.wide_nt ? launch_wide<10, 8, S, true, OP_I8, false, true, 2>(w, (size_t)a.wide_width * a.dim, s_wide)
                                              : launch_wide<10, 8, S, true, OP_I8, false, false, 2>(w, (size_t)a.wide_width * a.dim, s_wide);
                            if (e == hipSuccess) e = hipEventRecord(a.ev_join, a.side_stream);
                            if (e == hipSuccess) e = hipStreamWaitEvent(s, a.ev_join, 0);
                            return e;
                        }
                        hipError_t e = launch_wide<6, 4, S, true, OP_I8, false, true>(a, lds, s);
                        if (e != hipSuccess) return e;
                        return a.wide_nt ? launch_wide<10, 8, S, true, OP_I8, false, true, 2>(w, (size_t)a.wide_width * a.dim, s)
                                         : launch_wide<10, 8, S, true, OP_I8, false, false, 2>(w, (size_t)a.wide_width * a.dim, s);
                    }
                    if constexpr (S == 1) if (a.cand_lb) return launch_wide<6, 4, S, true, OP_I8, false, false, 4, true>(a, lds, s);
                    return launch_wide<6, 4, S, true, OP_I8>(a, lds, s);
                }
                return hipErrorInvalidValue;
            }
            if (a.quad_width == 128) return launch_wide<8, 8, S, true, OP_I8>(a, lds, s);
            if (a.quad_width == 96) return launch_wide<6, 8, S, true, OP_I8>(a, lds, s);
            if (a.quad_width == 64) return launch_wide<4, 8, S, true, OP_I8>(a, lds, s);
            return hipErrorInvalidValue;
        }
        if (a.f16) {
            if ((a.dim % 128) != 0 || !a.query_maxabs) return hipErrorInvalidValue;
            TileArgs b = a;
            bool q32 = false;
            const size_t lds = wide_lds_bytes(a.quad_width, a.dim, true, &q32);
            b.q32_lds = q32 ? 1 : 0;
            // <= 4 K steps per tile: whole-tile operand prefetch -- 4-wave blocks only (with 128 accumulator registers the
            // 8-wave form spills under it: C2 0.317 against 0.204 ms per serial step without)
            const bool pf = a.dim <= 128 && nw == 4;
            if (nw == 8) {            // one block per CU: up to 144 KB of staged queries + 14 KB of queues
                if (lds > 147456) return hipErrorInvalidValue;
                if (a.quad_width == 128) return launch_wide<8, 8, S, true, OP_F16>(b, lds, s);
                if (a.quad_width == 96) return launch_wide<6, 8, S, true, OP_F16>(b, lds, s);
                if (a.quad_width == 64) return launch_wide<4, 8, S, true, OP_F16>(b, lds, s);
                return hipErrorInvalidValue;
            }
            // (rows of <= 128 dims, k <= 64: the whole-tile operand prefetch in the 96-query form too -- 233 registers, no scratch since round 5)
            if constexpr (S == 1) { if (a.quad_width == 96 && lds <= 73728 && pf && a.opt_pf96) return launch_wide<6, 4, S, true, OP_F16, true>(b, lds, s); }
            if (a.quad_width == 96 && lds <= 73728) return launch_wide<6, 4, S, true, OP_F16>(b, lds, s);     // 80 KB per block: two per CU
            if (lds > 65536) return hipErrorInvalidValue;           // + 10 KB of static LDS: two blocks per CU
            if (a.quad_width == 64 && pf) return launch_wide<4, 4, S, true, OP_F16, true>(b, lds, s);
            if (pf) return hipErrorInvalidValue;
            if (a.quad_width == 64) return launch_wide<4, 4, S, true, OP_F16>(b, lds, s);
            if (a.quad_width == 32) return launch_wide<2, 4, S, true, OP_F16>(b, lds, s);
            return hipErrorInvalidValue;
        }
        if (nw != 4) return hipErrorInvalidValue;
        const size_t lds4 = 64ull * a.dim * 4, lds2 = 32ull * a.dim * 4;
        if (a.quad_width == 64 && lds4 <= 32768) return launch_wide<4, 4, S, true, OP_F32>(a, lds4, s);
        if (a.quad_width == 32 && lds2 <= 32768) return launch_wide<2, 4, S, true, OP_F32>(a, lds2, s);
        if (a.quad_width == 32 && a.q_blk) return launch_wide<2, 4, S, false, OP_F32>(a, 0, s);
        if (a.quad_width == 64 && a.q_blk) return launch_wide<4, 4, S, false, OP_F32>(a, 0, s);
        return hipErrorInvalidValue;
    }
    dim3 grid(a.grid_x, a.max_groups), block(256);
    if ((a.dim % 4) == 0) {
        if (a.dim > 256) hipLaunchKernelGGL((tile_filter_kernel<S, true, true>), grid, block, 0, s, a);
        else hipLaunchKernelGGL((tile_filter_kernel<S, true, false>), grid, block, 0, s, a);
    } else {
        hipLaunchKernelGGL((tile_filter_kernel<S, false, false>), grid, block, 0, s, a);
    }
#endif
    return hipGetLastError();
}

hipError_t launch_tile_filter(const TileArgs &a, hipStream_t s) {
    if ((a.filter_variant == 0 ? a.max_quads : a.max_groups) == 0 || a.grid_x == 0) return hipSuccess;
    if (a.k <= 64) return launch_filter_s<1>(a, s);
#ifndef PQV_DEV_C3_ONLY
    if (a.k <= 256) return launch_filter_s<4>(a, s);
#endif
    return hipErrorInvalidValue;
}

hipError_t launch_tile_rerank(const TileArgs &a, hipStream_t s) {
    if (a.max_groups == 0 || a.grid_x == 0) return hipSuccess;
#ifndef PQV_DEV_C3_ONLY
    if (a.k <= 64) return launch_tile_s<1>(a, s);
    if (a.k <= 256) return launch_tile_s<4>(a, s);
#endif
    return hipErrorInvalidValue;   // larger k uses stream_kernel
}



// an empty kernel the library launches at the first call for a device (and on a new stream): the runtime loads this unit's code
// object and sets up the stream's hardware queue then, not inside the first build or the first query
__global__ void touch_screen_kernel() {}
hipError_t touch_screen(hipStream_t s) {
    hipLaunchKernelGGL(touch_screen_kernel, dim3(1), dim3(64), 0, s);
    return hipGetLastError();
}

}  // namespace pqv
