// kernels_list.hip -- list_filter_kernel: the int8 MFMA screen of the batched candidate re-rank (src/ivf/search.rs:112-127 for a
// whole batch) in its ROW-STATIONARY form, for the lists that MANY queries of the batch probe (src/ivf/index.rs:57-63: a probed
// list's rows are candidates of every query that probes it).
//
// wide_filter_kernel keeps a quad of <= 96 (160) query images stationary in LDS and streams the list's rows past them: a list
// that P queries probe is streamed ceil(P / 96) times.  On clustered data the popular lists are probed by hundreds of queries
// (round 5, Gaussian mixture: the filter launch fetched 9.66 GB for 4.55 GB of minimum bytes -- 2.2 x).  Here the roles are
// swapped: a wave keeps the int8 images of 32 of its rows in REGISTERS for the whole K extent (dim bytes per row: 96 VGPRs at
// 768 dims) and the list's pairs -- ALL of them, up to 1024 -- stream past them through LDS in chunks of 32 query images,
// double-buffered, read from L2 (a pair's image is 768 bytes; the list's images are shared by every block of the list).  Every
// row of such a list leaves HBM ONCE per batch whatever P is; the contraction work is unchanged (v_mfma_i32_16x16x64_i8, exact
// int32 accumulation).  Bound, survivor queue, exact evaluation in the reference's order (index.rs:461-480) and the append /
// running-threshold protocol are wide_filter_kernel's: a pair is only ever dropped when the rigorous lower bound of its
// reference distance exceeds its query's admission threshold, so results are bit-identical.
//
// Block = 4 waves (two blocks per CU), one work item = (list's quad of up to 1024 pairs, row chunk).  Per PASS a wave holds the
// operand images of its next 32 rows (KS x 2 x 1 KiB from the blocked copy, nt: read once) and for every chunk of 32 pairs:
//   loads of chunk n + 2 -> staging registers | 4 KS MFMAs against chunk n (LDS ring slot n % 3) | screen |
//   staging registers -> ring slot (n + 2) % 3 | barrier
// -- a chunk's images are requested a whole iteration before they are stored and two before they are read, and nothing of the
// iteration waits for a load it has just issued.  (First form: the pairs' screen terms rode in four more staging registers, which
// the allocator spilled right behind their loads -- a wait for the whole queue per iteration: C3 1.07 ms against the wide-quad
// instance's 0.64.  Second form: direct-to-LDS loads, which hipcc answers with s_waitcnt vmcnt(0) in front of every use of an
// ordinary load while one is in flight.  Third form: two staging register sets selected by the iteration's parity -- the
// selection became register copies behind every load, i.e. the same wait.)  The NEXT pass' row images are requested inside the
// last chunk's K loop, each K step's registers right behind their last MFMAs.
// Thresholds and the pairs' screen terms are read by every wave one chunk ahead (agent scope), i.e. every 32 rows of every wave.
#include "device_common.hpp"

namespace pqv {

template <int KS, int S>
__global__ __launch_bounds__(256, 2) void list_filter_kernel(const TileArgs a) {
    constexpr int NW = 4, TS = 2, NGC = 2;
    constexpr uint32_t CQ = 16 * NGC;             // pairs per chunk
    constexpr uint32_t PROWS = 16 * TS;           // rows per wave and pass
    constexpr uint32_t G = 4 * KS;                // 16-byte columns per image row
    constexpr uint32_t DIM = 64 * KS;
    constexpr uint32_t MAXP = 1024;               // pairs per quad
    constexpr int PEND = 192;                     // survivor queue entries per wave (3 x 24 KB of ring + 7.5 KB: two blocks per CU at 768 dims)
    constexpr uint32_t QSH = 22;                  // queue entry = (pair slot in the quad << 22) | row offset from the wave's r0
    constexpr int SL = KS / 2;                    // 16-byte columns a thread stages per chunk (8 threads per pair)
    constexpr uint32_t BUF = CQ * G * 16u;        // bytes of a chunk buffer
    static_assert(KS % 4 == 0 && KS >= 4 && KS <= 12, "dim = 64 KS: 256, 512, 768 (1024-dim images run 64-query quads: no popular-list table)");

    // Workgroup b of a 1-D grid runs on XCD b % 8.  The table is list-major (a list's row chunks are consecutive items), and XCD x
    // takes the items [x M, (x + 1) M) in order: the blocks of ONE list run on one XCD, next to each other in time, so the list's
    // pair images -- re-read by every block for every 128 rows -- come out of that XCD's L2.  (With the chunks spread over the XCDs
    // the images of ~64 lists competed for each 4 MB L2 and came through the fabric again and again: as many bytes as the rows
    // at 130 pairs per list, three times as many at 400 -- C3 1.1 ms against the wide-quad instance's 0.64, the mixture 3.9 against 2.1.)
    const uint32_t nit = *a.n_items;
    const uint32_t per_xcd = (nit + 7u) / 8u;
    const uint32_t item = (blockIdx.x & 7u) * per_xcd + (blockIdx.x >> 3);
    if ((blockIdx.x >> 3) >= per_xcd || item >= nit) return;
    const uint32_t by = a.item_quad[item];
    const uint4 quad = a.quads[by];               // {cluster, first pair slot, pair count, first work item}
    const uint32_t bx = a.item_chunk ? a.item_chunk[item] : item - quad.w;
    const uint32_t c = quad.x, p0 = quad.y;
    const uint32_t cnt = quad.z < MAXP ? quad.z : MAXP;
    uint32_t tid = threadIdx.x;                   // (re-defined opaquely at the top of every chunk iteration: see the main loop)
    int lane = (int)(tid & 63u);
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    int l15 = lane & 15, kk = lane >> 4;
    const uint32_t k = a.k;

    const uint64_t lbeg = a.list_off[c], lend = a.list_off[c + 1];
    const uint64_t len = lend - lbeg;
    // the whole list in this launch: ceil(len / rows_per_block) blocks share it in equal wave pieces (a multiple of the pass)
    const uint64_t nch = (len + a.rows_per_block - 1) / a.rows_per_block;
    if (bx >= nch || cnt == 0) return;
    const uint64_t wrows = ((len + NW * nch - 1) / (NW * nch) + PROWS - 1) / PROWS * PROWS;
    uint64_t r0 = ((uint64_t)bx * NW + (uint64_t)wave) * wrows;
    uint64_t r1 = r0 + wrows;
    if (r1 > len) r1 = len;
    if (r0 > len) r0 = len;

    extern __shared__ float4 lq[];                // [3][CQ][G]: the chunk ring, column ch of pair q at ch ^ (q & 15)
    __shared__ uint32_t s_pair[MAXP];             // the quad's pairs as (query << 10) | probe rank (the ring loads and the waves' screens index them by slot)
    __shared__ __attribute__((aligned(16))) int aq_all[NW * CQ];
    __shared__ uint32_t pend_all[NW * PEND];
    int *aq = aq_all + wave * CQ;
    uint32_t *pend = pend_all + wave * PEND;

    // (stored as (query << 10) | probe rank: the one division by nprobe <= 1024 happens here, not per chunk and lane)
    for (uint32_t i = threadIdx.x; i < cnt; i += 256) {
        const uint32_t pair = a.pairs[p0 + i];
        s_pair[i] = ((pair / a.nprobe) << 10) | (pair % a.nprobe);
    }
    __syncthreads();

    const uint32_t nchunks = (cnt + CQ - 1) / CQ;
    const uint32_t npass = (uint32_t)(wrows / PROWS);
    const uint32_t total = npass * nchunks;       // block-uniform: every wave runs every iteration (barriers), rows or not
    const float cmargin = (float)(DIM + 16) * 2.384185791015625e-07f;   // (dim + 16) * 2^-22
    const float lscale = a.list_scale[c];
    const uint32_t nprobe = a.nprobe;
    const uint32_t n_part = a.n_part;
    const uint64_t blk0 = a.blk_off[c], blk_last = a.blk_off[c + 1] - 1;
    uint32_t lane_b = ((uint32_t)kk * 16 + (uint32_t)l15) * 16u;
    const __amdgpu_buffer_rsrc_t rt_thr = operand_rsrc(a.gthr);

    // ---- chunk staging: thread (sq, sc0) carries SL 16-byte columns of pair sq of a chunk -----------------------------------------
    uint32_t sq = tid >> 3, sc0 = tid & 7u;
    f32x4_raw st[SL];                             // (ext-vector elements: an array of the HIP float4 class stayed in scratch memory)
    auto stage_load = [&](uint32_t n) {           // chunk sequence number n (chunk n % nchunks of pass n / nchunks)
        const uint32_t slot = (n % nchunks) * CQ + sq;
        const uint32_t pk = s_pair[slot < cnt ? slot : cnt - 1];
        const uint32_t img = a.i8_pair_images ? (pk >> 10) * nprobe + (pk & 1023u) : pk >> 10;
        const float4 *s8 = reinterpret_cast<const float4 *>(a.q_i8 + (uint64_t)img * DIM);
#pragma unroll
        for (int h = 0; h < SL; ++h) st[h] = *reinterpret_cast<const f32x4_raw *>(s8 + sc0 + 8u * (uint32_t)h);
    };
    // (column (sc0 + 8 h) ^ (sq & 15) = (sc0 ^ (sq & 7)) + 8 ((h & 1) ^ (sq >> 3 & 1)) + 16 (h >> 1): two per-lane addresses and
    //  immediate offsets -- six precomputed addresses were six more registers)
    uint32_t st_off[2] = {0u, 0u};
    auto stage_store = [&](uint32_t n) {
        char *dst = reinterpret_cast<char *>(lq) + (n % 3u) * BUF;
#pragma unroll
        for (int h = 0; h < SL; ++h) *reinterpret_cast<f32x4_raw *>(dst + st_off[h & 1] + 256u * (uint32_t)(h >> 1)) = st[h];
    };
    auto relane = [&]() {
        lane = (int)(tid & 63u); l15 = lane & 15; kk = lane >> 4;
        lane_b = ((uint32_t)kk * 16 + (uint32_t)l15) * 16u;
        sq = tid >> 3; sc0 = tid & 7u;
        st_off[0] = (sq * G + ((sc0 ^ (sq & 7u)) + 8u * (0u ^ ((sq >> 3) & 1u)))) * 16u;
        st_off[1] = (sq * G + ((sc0 ^ (sq & 7u)) + 8u * (1u ^ ((sq >> 3) & 1u)))) * 16u;
    };
    relane();

    // ---- the wave's rows of a pass: operand images in registers, integer norms, residual bounds -------------------------------------
    i32x4_acc xs[KS][TS];                         // (16 int8 values per lane and K step: the MFMA's B operand as it is)
    int xn2i[TS] = {};
    float xres[TS] = {};
    auto ld_img = [&](__amdgpu_buffer_rsrc_t r, uint32_t uniform_bytes) -> i32x4_acc {
        return __builtin_bit_cast(i32x4_acc, __builtin_amdgcn_raw_buffer_load_b128(r, (int)lane_b, (int)uniform_bytes, 2));      // nt: read once
    };
    auto rows_desc = [&](uint64_t t0, uint32_t (&so)[TS]) {
        const float4 *b0 = nullptr;
#pragma unroll
        for (int t = 0; t < TS; ++t) {
            uint64_t T = blk0 + ((t0 + 16 * t) >> 4);
            if (T > blk_last) T = blk_last;       // tiles past the list's end: masked by the screen
            const float4 *b = a.mat_blk + T * G * 16;
            if (t == 0) b0 = b;
            so[t] = (uint32_t)((b - b0) * 16);
        }
        return operand_rsrc(b0);
    };
    auto load_terms = [&](uint64_t t0) {          // rows [t0, t0 + 32) of the list (t0 < r1)
        const uint32_t nv = (r1 - t0 < PROWS) ? (uint32_t)(r1 - t0) : PROWS;
#pragma unroll
        for (int t = 0; t < TS; ++t) {
            uint32_t rr = (uint32_t)(16 * t + l15);
            if (rr >= nv) rr = nv - 1;
            xn2i[t] = a.row_n2i[lbeg + t0 + rr];
            xres[t] = a.row_res[lbeg + t0 + rr];
        }
    };
    auto load_rows = [&](uint64_t t0) {
        uint32_t so[TS];
        const __amdgpu_buffer_rsrc_t r = rows_desc(t0, so);
#pragma unroll
        for (int ks = 0; ks < KS; ++ks)
#pragma unroll
            for (int t = 0; t < TS; ++t) xs[ks][t] = ld_img(r, so[t] + (uint32_t)ks * 1024u);
    };

    uint32_t npend = 0, n_exact = 0;
    auto widen = [](uint32_t h) -> uint64_t { return ((uint64_t)h << 32) | 0xFFFFFFFFull; };
    // Running threshold + append of one lane's pair: wide_filter_kernel's protocol (12 distance bins of 8-bit counters per query,
    // one returning atomic per word; the lane whose add takes a bin's counter to k publishes that bin's upper edge)
    auto append_pair = [&](bool pass, uint32_t qrow, float dval, uint64_t key, uint32_t srow) -> bool {
        int hb_bin = 0;
        float4 hb = make_float4(0.f, 0.f, 0.f, 0.f);
        if (pass && k > 1u && a.thr_hist) {
            hb = a.thr_bins[qrow];
            hb_bin = hb.z > 0.0f ? (int)fminf(fmaxf((hb.x - dval) * hb.z, 0.0f), 12.0f) : 0;
        }
        bool full = false;
        if (pass) {
            const unsigned long long ones = 0x0101010101010101ull;
            unsigned long long w0 = 0ull, w1 = 0ull;
            unsigned long long *h2 = reinterpret_cast<unsigned long long *>(a.thr_hist) + (uint64_t)qrow * 2;
            const int b = hb_bin;
            const unsigned long long add0 = b >= 8 ? ones : b >= 1 ? ones << (8 * (8 - b)) : 0ull;
            const unsigned long long add1 = b >= 9 ? (ones & 0xFFFFFFFFull) << (8 * (12 - b)) & 0xFFFFFFFFull : 0ull;
            if (add1) w1 = atomicAdd(h2 + 1, add1) + add1;
            if (add0) w0 = atomicAdd(h2, add0) + add0;
            const uint32_t idx = atomicAdd(a.cand_cnt + qrow, 1u);
            if (idx < a.cand_cap) {
                a.cand_keys[(uint64_t)qrow * a.cand_cap + idx] = key;
                a.cand_vals[(uint64_t)qrow * a.cand_cap + idx] = srow;
            } else {
                full = true;           // buffer full: this wave's sorted list (slow, exact)
                a.spilled[qrow] = 1u;
            }
            if (k == 1u) {
                atomicMin(a.gthr + qrow, (unsigned long long)(key | 0xFFFFFFFFull));
            } else if (b > 0) {
                int bsel = 0;
#pragma unroll
                for (int jj = 7; jj >= 0; --jj) {
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
    // Exact evaluation of queue entries [start, start + count), count <= 64: L = 1, 2, 4 or 8 lanes per pair fetch L x NB
    // consecutive 16-byte row chunks per round and the reference's chain (index.rs:461-480) passes through them in chunk order.
    auto eval = [&](uint32_t start, uint32_t count) -> uint32_t {
        wave_lds_fence();
        constexpr int NB = 8;
        constexpr uint32_t Gx = DIM / 4;
        uint32_t lg = 0;
        while (lg < 3 && (count << (lg + 1)) <= 64u && (Gx % ((2u * NB) << lg)) == 0u) ++lg;      // wave-uniform
        const uint32_t L = 1u << lg;
        const uint32_t pi = (uint32_t)lane >> lg, pj = (uint32_t)lane & (L - 1u);
        const bool valid = pi < count;
        const bool have = valid && pj == 0u;
        const uint32_t pe = pend[start + (valid ? pi : 0)];
        const uint32_t slot = pe >> QSH;
        const uint64_t roff = r0 + (pe & ((1u << QSH) - 1u));
        const uint64_t lpos = lbeg + roff;
        const uint32_t srow = a.row_of ? a.row_of[lpos] : (uint32_t)lpos;
        const uint32_t pk = s_pair[slot < cnt ? slot : 0u];
        const uint32_t qrow = pk >> 10, pair = qrow * nprobe + (pk & 1023u);
        const float *x = a.mat + (uint64_t)srow * DIM;
        const float4 *qg = reinterpret_cast<const float4 *>(a.queries + (uint64_t)qrow * DIM);
        const uint64_t cbase = a.cand_base[pair];
        const uint32_t thr_now = buf_ld4<16>(rt_thr, qrow * 8u + 4u, 0u);
        float sum = 0.0f;
        const uint32_t first = (uint32_t)lane & ~(L - 1u);
        for (uint32_t g0 = 0; g0 < Gx; g0 += NB * L) {
            const uint32_t g = g0 + NB * pj;
            float4 xv[NB], qv[NB];
#pragma unroll
            for (int u = 0; u < NB; ++u) xv[u] = load4<true>(x + (g + u) * 4);
#pragma unroll
            for (int u = 0; u < NB; ++u) qv[u] = qg[g + u];
            float t[NB];
#pragma unroll
            for (int u = 0; u < NB; ++u) {
                const float d0 = qv[u].x - xv[u].x, d1 = qv[u].y - xv[u].y;
                const float d2 = qv[u].z - xv[u].z, d3 = qv[u].w - xv[u].w;
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
        const uint64_t pos = cbase + roff;
        const uint64_t mykey_all = (have && pos < a.max_pos) ? (((uint64_t)__float_as_uint(sum) << 32) | (uint64_t)(uint32_t)pos) : KEY_EMPTY;
        const bool pass = mykey_all < widen(thr_now);
        const bool spill = !append_pair(pass, qrow, sum, mykey_all, srow);
        unsigned long long todo = __ballot(spill);
        while (todo) {
            const uint32_t qq = readlane_u32(slot, __builtin_ctzll(todo));
            const bool mine = spill && slot == qq;
            todo &= ~__ballot(mine);
            const uint64_t mykey = mine ? mykey_all : KEY_EMPTY;
            const uint32_t pk2 = s_pair[qq];
            const uint32_t qr = pk2 >> 10;
            const uint64_t thr = widen(buf_ld4<16>(rt_thr, qr * 8u + 4u, 0u));
            if (__ballot(mykey < thr) != 0ull) {
                const uint64_t li = (uint64_t)qr * n_part + (pk2 & 1023u) * a.slots_per_pair + a.slot_base + bx * NW + wave;
                const uint64_t base = li * k;
                bool fresh = false;
                if (a.part_flags) {
                    fresh = a.part_flags[li] == 0;
                    if (fresh && lane == 0) a.part_flags[li] = 1;
                }
                (void)tile_fold<S>(a.part_keys + base, a.part_vals + base, a.gthr + qr, thr, KEY_EMPTY, mykey, srow, k, lane, fresh);
            }
        }
        return count;
    };
    auto drain = [&](uint32_t keep_below) -> bool {
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

    // ---- prologue: chunks 0 and 1 into the ring; the first pass' rows; the first chunk's screen terms --------------------------------
    stage_load(0);
    stage_store(0);
    if (total > 1) { stage_load(1); stage_store(1); }
    if (r0 < r1) load_rows(r0);
    // screen terms of a chunk's pairs, read by lanes 0 .. 31 (mirrored in 32 .. 63) of every wave at the top of the chunk's iteration
    // and used behind its K loop: the DISTANCE half of the query's threshold key (agent scope: it tightens while the kernel runs),
    // the residual bound and |vi|^2 of the pair's image, the lower bound of d2(query, any row of the list)
    uint32_t m_thr = 0xFFFFFFFFu, m_res = 0u, m_n2i = 0u, m_lb = 0u;
    auto load_meta = [&](uint32_t n) {
        const uint32_t slot = (n % nchunks) * CQ + ((uint32_t)lane & (CQ - 1u));
        const uint32_t pk = s_pair[slot < cnt ? slot : cnt - 1];
        const uint32_t qrow = pk >> 10, pair = qrow * nprobe + (pk & 1023u);
        const uint32_t img = a.i8_pair_images ? pair : qrow;
        m_thr = buf_ld4<16>(rt_thr, qrow * 8u + 4u, 0u);
        m_res = __float_as_uint(a.q_res[img]);
        m_n2i = (uint32_t)a.q_n2i[img];
        m_lb = a.pair_lb ? __float_as_uint(a.pair_lb[pair]) : 0u;
    };
    __syncthreads();

    uint64_t screened = 0;
    for (uint32_t n = 0; n < total; ++n) {
        const uint32_t chunk = n % nchunks, pass_i = n / nchunks;
        // The thread index is handed back through an opaque (empty) asm: everything derived from it -- lane fields, ring positions,
        // LDS and operand offsets: loop invariants the compiler otherwise carries in registers through the whole loop, spilling what
        // does not fit beside the 8 KS registers of row images -- is recomputed per chunk (a few dozen VALU operations against 4 KS MFMAs)
        asm volatile("" : "+v"(tid));
        relane();
        // (on entry: every wave has passed the barrier behind iteration n - 1 -- chunks n and n + 1 are complete in their ring slots,
        //  and nobody reads slot (n + 2) % 3 = (n - 1) % 3 any more)
        // Loads and their first use sit in the SAME iteration with the K loop between them -- staging registers, the chunk's screen
        // terms, the pass' row terms -- except the next pass' row images, requested inside the previous chunk's K loop: those are
        // touched HERE, at the top of EVERY iteration (an empty asm statement: free), while nothing else is in flight.  hipcc waits for
        // a loop-carried load with s_waitcnt vmcnt(0) wherever some path first reads it -- left to the K loop that was a wait behind
        // this iteration's fresh requests, a round trip per chunk; a touch at the top of a pass only leaves the other path pending.
#pragma unroll
        for (int ks = 0; ks < KS; ++ks)
#pragma unroll
            for (int t = 0; t < TS; ++t) asm volatile("" : "+v"(xs[ks][t]));
        __builtin_amdgcn_sched_barrier(0);
        if (n + 2 < total) stage_load(n + 2);
        const uint64_t t0 = r0 + (uint64_t)pass_i * PROWS;
        if (t0 < r1) {
            const uint32_t nvalid = (r1 - t0 < PROWS) ? (uint32_t)(r1 - t0) : PROWS;
            load_meta(n);
            if (chunk == 0u) load_terms(t0);
            i32x4_acc acc[NGC][TS];
#pragma unroll
            for (int t = 0; t < TS; ++t)
#pragma unroll
                for (int g = 0; g < NGC; ++g) acc[g][t] = (i32x4_acc){0, 0, 0, 0};
            // A operands: column (4 ks + kk) ^ l15 of pair 16 g + l15 = 16 (ks >> 2) + 4 ((ks & 3) ^ (l15 >> 2)) + (kk ^ (l15 & 3)): FOUR
            // per-lane addresses (ks & 3) and immediate offsets for ks >> 2 and g -- one address register per K step was 2 KS
            // registers, spilled and reloaded in front of every ds_read
            const char *qb = reinterpret_cast<const char *>(lq) + (n % 3u) * BUF;
            const char *qa[4];
#pragma unroll
            for (int j = 0; j < 4; ++j)
                qa[j] = qb + ((uint32_t)l15 * G + ((((uint32_t)j ^ ((uint32_t)l15 >> 2)) << 2) | ((uint32_t)kk ^ ((uint32_t)l15 & 3u)))) * 16u;
            // (two rotating operand sets: the reads of K step ks + 1 go out before the MFMAs of step ks, and a scheduling barrier per
            //  step keeps the machine scheduler from hoisting all 2 KS reads -- 8 KS registers -- to the top of the chunk)
            i32x4_acc qc[2][NGC];
            auto a_read = [&](int ks, int set) {
#pragma unroll
                for (int g = 0; g < NGC; ++g)
                    qc[set][g] = *reinterpret_cast<const i32x4_acc *>(qa[ks & 3] + 256u * (uint32_t)(ks >> 2) + (uint32_t)g * (16u * G * 16u));
            };
            // `next`: the last chunk of a pass that is not the wave's last -- every K step's row registers are requested for the NEXT
            // pass right behind their last MFMAs, so the images fly during the rest of the chunk, its screen and the barrier
            auto kloop = [&](auto next) {
                constexpr bool NEXT = decltype(next)::value;
                [[maybe_unused]] uint32_t so[TS];
                [[maybe_unused]] __amdgpu_buffer_rsrc_t rn = rt_thr;
                if constexpr (NEXT) rn = rows_desc(t0 + PROWS, so);
                a_read(0, 0);
#pragma unroll
                for (int ks = 0; ks < KS; ++ks) {
                    if (ks + 1 < KS) a_read(ks + 1, (ks + 1) & 1);
                    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                    for (int g = 0; g < NGC; ++g)
#pragma unroll
                        for (int t = 0; t < TS; ++t) acc[g][t] = __builtin_amdgcn_mfma_i32_16x16x64_i8(qc[ks & 1][g], xs[ks][t], acc[g][t], 0, 0, 0);
                    __builtin_amdgcn_sched_barrier(0);
                    if constexpr (NEXT) {
#pragma unroll
                        for (int t = 0; t < TS; ++t) xs[ks][t] = ld_img(rn, so[t] + (uint32_t)ks * 1024u);
                        __builtin_amdgcn_sched_barrier(0);
                    }
                }
            };
            const bool pass_end = chunk + 1 == nchunks;
            const bool next_rows = pass_end && t0 + PROWS < r1;
            // per-pair screen terms of this wave: skip <=> (dot - (Nx >> 1)) + A2 < 0 (wide_filter_kernel: the int8 bound on the residual);
            // the row term -(Nx >> 1) is added here, not preset in the accumulators: nothing in front of the K loop waits for a load
            uint32_t bits = 0, n_live = 0;
            auto screen = [&]() {
                float R = 0.0f;
#pragma unroll
                for (int t = 0; t < TS; ++t) R = fmaxf(R, (uint32_t)(16 * t + l15) < nvalid ? xres[t] : 0.0f);
#pragma unroll
                for (int off = 8; off > 0; off >>= 1) R = fmaxf(R, __shfl_xor(R, off, 64));
                wave_lds_fence();
                {
                    const uint32_t slot = chunk * CQ + ((uint32_t)lane & (CQ - 1u));
                    const float thr_d = __uint_as_float(m_thr), c_res = __uint_as_float(m_res), c_lb = __uint_as_float(m_lb);
                    const bool open = !(c_res <= 3.0e38f) || m_thr == 0xFFFFFFFFu || !(thr_d <= 3.0e38f);
                    int a2 = 1 << 29;                                   // never skip
                    const bool dead = slot >= cnt || c_lb > thr_d;      // not a pair of this quad / no row of the list can enter this query's top-k
                    if (dead) a2 = -(1 << 30);
                    else if (!open) {
                        const float v = lscale * (sqrtf(thr_d * (1.0f + 4.0f * cmargin)) * 1.000002f + c_res + R);
                        const float v2 = fminf(v * v * 1.000002f, 1.0e9f);
                        a2 = ((int)ceilf(v2) + 1 - (int)m_n2i + 1) >> 1;
                    }
                    if ((uint32_t)lane < CQ) aq[lane] = a2;
                    n_live = (uint32_t)__popcll(__ballot(!dead) & 0xFFFFFFFFull);
                }
                wave_lds_fence();
                int rt[TS];
#pragma unroll
                for (int t = 0; t < TS; ++t) rt[t] = -(xn2i[t] >> 1);
#pragma unroll
                for (int g = 0; g < NGC; ++g) {
                    const int4 a4 = *reinterpret_cast<const int4 *>(aq + 16 * g + 4 * kk);
                    const int ar[4] = {a4.x, a4.y, a4.z, a4.w};
#pragma unroll
                    for (int r = 0; r < 4; ++r)
#pragma unroll
                        for (int t = 0; t < TS; ++t) bits = __builtin_amdgcn_alignbit(bits, (uint32_t)(acc[g][t][r] + (ar[r] + rt[t])), 31);
                }
            };
            // (the next pass' row TERMS are read at the top of its first chunk: the screen still needs this pass')
            // What the iteration loaded at its top is opaque up to the touch below: whatever of the screen does not depend on the
            // accumulators would otherwise be hoisted above the K loop -- with the wait for those loads in front of it.  The wait sits
            // BEHIND the K loop (the loads had its whole length) -- except in a pass' last chunk, where it sits in front: behind, it
            // would also wait for the next pass' row images requested inside that K loop (hipcc: vmcnt(0)), an HBM round trip
            // instead of an L2 one.
            auto touch = [&]() {
                asm volatile("" : "+v"(m_thr), "+v"(m_res), "+v"(m_n2i), "+v"(m_lb));
                asm volatile("" : "+v"(xres[0]), "+v"(xres[1]), "+v"(xn2i[0]), "+v"(xn2i[1]));
            };
            if (next_rows) { touch(); kloop(std::true_type{}); }
            else { kloop(std::false_type{}); touch(); }
            screen();
            uint32_t vm = 0;
#pragma unroll
            for (int g = 0; g < NGC; ++g)
#pragma unroll
                for (int r = 0; r < 4; ++r)
#pragma unroll
                    for (int t = 0; t < TS; ++t) {
                        const bool ok = chunk * CQ + 16u * (uint32_t)g + 4u * (uint32_t)kk + (uint32_t)r < cnt && (uint32_t)(16 * t + l15) < nvalid;
                        vm |= ok ? 1u << (15 - (8 * g + 2 * r + t)) : 0u;
                    }
            uint32_t keep = ~bits & vm;
            screened += (uint64_t)nvalid * n_live;
            const uint32_t tot = (uint32_t)__popc(keep);
            const uint32_t incl = wave_incl_scan_u32(tot);
            const uint32_t wtot = readlane_u32(incl, 63);
            const uint32_t rowbase = (uint32_t)(t0 - r0) + (uint32_t)l15;
            const uint32_t slot0 = chunk * CQ + 4u * (uint32_t)kk;
            if (wtot != 0u) {
                if (npend + wtot <= (uint32_t)PEND) {
                    uint32_t at = npend + incl - tot;
                    while (keep) {
                        const uint32_t b = 31u - (uint32_t)__clz(keep);
                        keep &= ~(1u << b);
                        const uint32_t idx = 15u - b;
                        pend[at++] = ((slot0 + 16u * (idx >> 3) + ((idx >> 1) & 3u)) << QSH) + rowbase + 16u * (idx & 1u);
                    }
                    npend += wtot;
                } else {
                    // more survivors than the queue holds (a query without a threshold yet): a bit position at a time (<= 64
                    // entries each), evaluating in between; the row images are fetched again behind the evaluations (L2) -- of
                    // this pass, or of the next one if they had been requested already
                    for (uint32_t b = 16; b-- > 0;) {
                        const bool mine = (keep >> b) & 1u;
                        const unsigned long long m = __ballot(mine);
                        const uint32_t nb = (uint32_t)__popcll(m);
                        if (nb == 0u) continue;
                        if (npend + nb > (uint32_t)PEND) drain(1u);
                        const uint32_t idx = 15u - b;
                        if (mine) pend[npend + (uint32_t)__popcll(m & ((1ull << lane) - 1ull))] =
                            ((slot0 + 16u * (idx >> 3) + ((idx >> 1) & 3u)) << QSH) + rowbase + 16u * (idx & 1u);
                        npend += nb;
                        wave_lds_fence();
                    }
                    drain(64u);
                    load_rows(next_rows ? t0 + PROWS : t0);      // (the row TERMS in registers stay this pass': the next pass reads its own at its top)
#pragma unroll
                    for (int ks = 0; ks < KS; ++ks)
#pragma unroll
                        for (int t = 0; t < TS; ++t) asm volatile("" : "+v"(xs[ks][t]));       // (waited for here, on the rare path)
                }
            }
            if (pass_end) drain(t0 + PROWS >= r1 ? 1u : 64u);
        }
        if (n + 2 < total) stage_store(n + 2);
        __syncthreads();
    }
    if (a.stats && lane == 0) {
        unsigned long long *stt = a.stats + 8 + 16 * ((blockIdx.x + (uint32_t)wave * 17u) % STATS_SLOTS);
        atomicAdd(&stt[0], (unsigned long long)screened);
        atomicAdd(&stt[1], (unsigned long long)n_exact);
    }
}


template <int KS, int S>
static hipError_t launch_list(const TileArgs &a, hipStream_t s) {
    auto kern = list_filter_kernel<KS, S>;
    const size_t lds = 3ull * 32 * 64 * KS;
    static std::atomic<bool> raised{false};
    if (lds > 49152 && !raised.load(std::memory_order_relaxed)) {
        const hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) { (void)hipGetLastError(); return e; }
        raised.store(true, std::memory_order_relaxed);
    }
    hipLaunchKernelGGL(kern, dim3((a.max_items + 7u) / 8u * 8u + 8u), dim3(256), lds, s, a);
    return hipGetLastError();
}

// `a`: the wide table's view of the step (item_quad / item_chunk / n_items / max_items / rows_per_block of THAT table, quads of up
// to 1024 pairs).  int8 residual images, IVF-ordered blocked copy, the whole list in one launch.
hipError_t launch_list_filter(const TileArgs &a, hipStream_t s) {
    if (!a.i8 || (a.dim % 256) != 0 || a.dim < 256 || a.dim > 768 || !a.item_quad || !a.n_items || a.max_items == 0 || !a.mat_blk || !a.q_i8 || !a.q_n2i ||
        !a.q_res || !a.list_scale || !a.row_n2i || !a.row_res || !a.cand_keys || a.cand_lb || a.row_offset != 0 || a.row_end != 0 ||
        a.quad_width > 1024 || a.rows_per_block == 0 || a.rows_per_block / 4 + 64u >= (1u << 22) || a.k > 256)
        return hipErrorInvalidValue;
#define PQV_LIST_CASE(KS_)                                                                                               \
    if (a.dim == 64u * (KS_)) return a.k <= 64 ? launch_list<KS_, 1>(a, s) : launch_list<KS_, 4>(a, s);
    PQV_LIST_CASE(4) PQV_LIST_CASE(8) PQV_LIST_CASE(12)
#undef PQV_LIST_CASE
    return hipErrorInvalidValue;
}


// an empty kernel the library launches at the first call for a device (and on a new stream): the runtime loads this unit's code
// object and sets up the stream's hardware queue then, not inside the first build or the first query
__global__ void touch_list_kernel() {}
hipError_t touch_list(hipStream_t s) {
    hipLaunchKernelGGL(touch_list_kernel, dim3(1), dim3(64), 0, s);
    return hipGetLastError();
}

}  // namespace pqv
