// kernels_probe.hip -- gfx950 kernels in front of the re-rank: stream_kernel (the exact streaming distance pass: candidate
// re-rank src/ivf/search.rs:112-127, centroid probe index.rs:130-149, k-means++ rounds index.rs:344-369), probe_rows_kernel /
// probe_single_kernel (find_closest_centroids for a batch / one query), quantize_pairs_i8_kernel, merge_kernel (heap semantics of
// search.rs:119-126 == k smallest by (d2, candidate position)) and the pair bucketing (pair_scan / pair_scatter).
#include "device_common.hpp"

namespace pqv {

// ------------------------------------------------------------------------------------
// stream_kernel
//
// grid = (blocks_per_list, nprobe | 1, nq); block = 256 threads = 4 independent waves.
// A wave owns a contiguous run of rows of one inverted list and walks it in 64-row tiles.
// For a tile and a chunk of CG float4 groups of the dimension:
//   1. every load instruction reads 64 x 16 B, fully coalesced (CG = 64: one 1 KiB row
//      segment; CG = 32: two 512 B segments), NB instructions in flight;
//   2. each lane turns its float4 into the reference's per-group partial
//      t = ((d0^2 + d1^2) + d2^2) + d3^2  (PQV_L2SQ_REF4) or four squares (PQV_L2SQ_SEQ)
//      and parks it in a [group][row] LDS tile (XOR-swizzled: conflict-free both ways);
//   3. lane r then replays row r's serial chain  sum += t_g  in ascending g -- the one
//      part of the reference arithmetic that cannot be re-associated.
// LDS traffic is 1/4 of the streamed bytes (REF4), VALU ~12 ops per 16 B: the kernel is
// bound by the HBM/L2 stream.
// ------------------------------------------------------------------------------------
template <int CG, int S, int MODE, bool SEQ, bool ALIGNED>
__global__ __launch_bounds__(256) void stream_kernel(const StreamArgs a) {
    if (a.zero_u32 && blockIdx.x == 0 && blockIdx.y == 0) {     // scratch the NEXT kernels expect zeroed
        for (uint32_t i = blockIdx.z * 256 + threadIdx.x; i < a.zero_n; i += gridDim.z * 256) a.zero_u32[i] = 0u;
    }
    constexpr int RPI = 64 / CG;        // rows per load instruction
    constexpr int NI = CG;              // load instructions per 64-row tile
    constexpr int EPL = SEQ ? 4 : 1;    // LDS values per lane item
    constexpr int LROWS = CG * EPL;     // chain length per chunk
    constexpr int NB = 8;               // loads in flight per lane
    static_assert(NI % NB == 0, "NI must be a multiple of NB");

    // [chain element e][row r] tile per wave, XOR-swizzled (column r ^ (e & 63)) so that
    // both the group-major writes and the row-major chain reads are bank-conflict-free
    // without padding: CG = 64 uses exactly 64 KiB per block.
    __shared__ float lds_all[4 * LROWS * 64];
    const int lane = threadIdx.x & 63;
    const int wave = threadIdx.x >> 6;
    float *lds = lds_all + wave * (LROWS * 64);
#define LDS_AT(e, r) lds[(e) * 64 + ((r) ^ ((e) & 63))]

    const uint32_t q = blockIdx.z, j = blockIdx.y;
    uint64_t lbeg, lend, cbase;
    if (a.probe) {
        const uint32_t c = a.probe[(uint64_t)q * a.nprobe + j];
        lbeg = a.list_off[c];
        lend = a.list_off[c + 1];
        cbase = a.cand_base[(uint64_t)q * a.nprobe + j];
    } else {
        lbeg = a.single_begin;
        lend = a.single_end;
        cbase = 0;
    }
    const uint64_t len = lend - lbeg;
    const uint64_t wrows = a.rows_per_block / 4;
    const uint64_t r0 = (uint64_t)blockIdx.x * a.rows_per_block + (uint64_t)wave * wrows;
    uint64_t r1 = r0 + wrows;
    if (r1 > len) r1 = len;

    const uint32_t dim = a.dim;
    const uint32_t G = dim >> 2;
    const uint32_t tail = dim & 3u;
    const float *qv = a.queries + (uint64_t)q * dim;
    const int g_in = lane % CG;      // my float4 group inside a chunk
    const int row_in = lane / CG;    // my row inside a load instruction

    WaveTopk<S> tk;
    if constexpr (MODE == STREAM_TOPK) tk.init();

    for (uint64_t t0 = r0; t0 < r1; t0 += 64) {
        const uint32_t nvalid = (r1 - t0 < 64) ? (uint32_t)(r1 - t0) : 64u;
        // storage row of tile row `lane` (clamped so every address is in range)
        const uint32_t lrow = (uint32_t)lane < nvalid ? (uint32_t)lane : nvalid - 1;
        const uint64_t lpos = lbeg + t0 + lrow;
        const uint32_t my_srow = a.row_of ? a.row_of[lpos] : (uint32_t)lpos;

        float sum = 0.0f;
        for (uint32_t c0 = 0; c0 < G; c0 += CG) {
            const uint32_t ng = (G - c0 < (uint32_t)CG) ? (G - c0) : (uint32_t)CG;
            const bool gvalid = (uint32_t)g_in < ng;
            const uint32_t goff = (c0 + (gvalid ? g_in : 0)) * 4;
            const float4 qq = load4<ALIGNED>(qv + goff);

#pragma unroll 1
            for (int ib = 0; ib < NI; ib += NB) {
                float4 x[NB];
#pragma unroll
                for (int u = 0; u < NB; ++u) {
                    uint32_t rr = (uint32_t)((ib + u) * RPI + row_in);
                    if (rr >= nvalid) rr = nvalid - 1;
                    const uint32_t srow = (uint32_t)__shfl((int)my_srow, (int)rr, 64);
                    x[u] = load4<ALIGNED>(a.mat + (uint64_t)srow * dim + goff);
                }
#pragma unroll
                for (int u = 0; u < NB; ++u) {
                    const int rr = (ib + u) * RPI + row_in;
                    const float d0 = qq.x - x[u].x, d1 = qq.y - x[u].y;
                    const float d2 = qq.z - x[u].z, d3 = qq.w - x[u].w;
                    if constexpr (SEQ) {
                        if (gvalid) {
                            LDS_AT(g_in * 4 + 0, rr) = d0 * d0;
                            LDS_AT(g_in * 4 + 1, rr) = d1 * d1;
                            LDS_AT(g_in * 4 + 2, rr) = d2 * d2;
                            LDS_AT(g_in * 4 + 3, rr) = d3 * d3;
                        }
                    } else {
                        float t = d0 * d0 + d1 * d1;
                        t = t + d2 * d2;
                        t = t + d3 * d3;
                        if (gvalid) LDS_AT(g_in, rr) = t;
                    }
                }
            }
            wave_lds_fence();
            const uint32_t nchain = ng * EPL;
            uint32_t e = 0;
            for (; e + 8 <= nchain; e += 8) {
                float v[8];
#pragma unroll
                for (int u = 0; u < 8; ++u) v[u] = LDS_AT(e + u, lane);
#pragma unroll
                for (int u = 0; u < 8; ++u) sum = sum + v[u];
            }
            for (; e < nchain; ++e) sum = sum + LDS_AT(e, lane);
            wave_lds_fence();
        }
        if (tail) {  // scalar tail of squared_l2_distance (index.rs:474-478)
            const float *xr = a.mat + (uint64_t)my_srow * dim + (uint64_t)G * 4;
            const float *qt = qv + (uint64_t)G * 4;
            for (uint32_t e = 0; e < tail; ++e) {
                const float d = qt[e] - xr[e];
                sum = sum + d * d;
            }
        }

        const uint64_t pos = cbase + t0 + (uint64_t)lane;
        const bool valid = (uint32_t)lane < nvalid && pos < a.max_pos;
        if constexpr (MODE == STREAM_TOPK) {
            const uint64_t mykey =
                valid ? (((uint64_t)__float_as_uint(sum) << 32) | (uint64_t)(uint32_t)pos)
                      : KEY_EMPTY;
            tk.offer(mykey, my_srow, a.k, lane);
        } else if constexpr (MODE == STREAM_MINUPD) {
            if (valid) {
                const float old = a.out_f32[pos];
                if (sum < old) {                       // index.rs:363-365
                    a.out_f32[pos] = sum;
                    if (a.mirror_f32) a.mirror_f32[pos] = sum;
                    if (a.mirror_t) a.mirror_t[(pos % a.mirror_chunk) * a.mirror_stride + pos / a.mirror_chunk] = sum;
                }
            }
        } else {
            if (valid) a.out_f32[pos] = sum;
        }
    }

    if constexpr (MODE == STREAM_TOPK) {
        const uint32_t n_part = a.nprobe * a.blocks_per_list * 4;
        const uint32_t pi = (j * a.blocks_per_list + blockIdx.x) * 4 + wave;
        const uint64_t base = ((uint64_t)q * n_part + pi) * a.k;
#pragma unroll
        for (int s = 0; s < S; ++s) {
            const uint32_t e = s * 64 + lane;
            if (e < a.k) {
                a.part_keys[base + e] = tk.key[s];
                a.part_vals[base + e] = tk.val[s];
            }
        }
    }
}

#undef LDS_AT

template <int CG, int S, int MODE, bool SEQ, bool ALIGNED>
static hipError_t launch_stream_t(const StreamArgs &a, hipStream_t s) {
    dim3 grid(a.blocks_per_list, a.probe ? a.nprobe : 1, a.nq);
    hipLaunchKernelGGL((stream_kernel<CG, S, MODE, SEQ, ALIGNED>), grid, dim3(256), 0, s, a);
    return hipGetLastError();
}

template <int S>
static hipError_t launch_stream_topk_s(const StreamArgs &a, hipStream_t s) {
    const bool aligned = (a.dim % 4) == 0;
    const uint32_t G = a.dim / 4;
    if (a.metric == 1) {
        return aligned ? launch_stream_t<16, S, STREAM_TOPK, true, true>(a, s)
                       : launch_stream_t<16, S, STREAM_TOPK, true, false>(a, s);
    }
    if (!aligned) return launch_stream_t<32, S, STREAM_TOPK, false, false>(a, s);
    if (G >= 64 && G % 64 == 0) return launch_stream_t<64, S, STREAM_TOPK, false, true>(a, s);
    return launch_stream_t<32, S, STREAM_TOPK, false, true>(a, s);
}

hipError_t launch_stream(const StreamArgs &a, StreamMode mode, hipStream_t s) {
    if (a.nq == 0 || a.blocks_per_list == 0) return hipSuccess;
    if (mode == STREAM_TOPK) {
        if (a.k <= 64) return launch_stream_topk_s<1>(a, s);
        if (a.k <= 256) return launch_stream_topk_s<4>(a, s);
        if (a.k <= 1024) return launch_stream_topk_s<16>(a, s);
        return hipErrorInvalidValue;
    }
    const bool aligned = (a.dim % 4) == 0;
    const uint32_t G = a.dim / 4;
    if (a.metric == 1) {
        if (mode == STREAM_MINUPD)
            return aligned ? launch_stream_t<16, 1, STREAM_MINUPD, true, true>(a, s)
                           : launch_stream_t<16, 1, STREAM_MINUPD, true, false>(a, s);
        if (mode == STREAM_DIST)
            return aligned ? launch_stream_t<16, 1, STREAM_DIST, true, true>(a, s)
                           : launch_stream_t<16, 1, STREAM_DIST, true, false>(a, s);
        return hipErrorInvalidValue;
    }
    if (mode == STREAM_MINUPD) {
        if (!aligned) return launch_stream_t<32, 1, STREAM_MINUPD, false, false>(a, s);
        if (G >= 64 && G % 64 == 0) return launch_stream_t<64, 1, STREAM_MINUPD, false, true>(a, s);
        return launch_stream_t<32, 1, STREAM_MINUPD, false, true>(a, s);
    }
    if (mode == STREAM_DIST)
        return aligned ? launch_stream_t<32, 1, STREAM_DIST, false, true>(a, s)
                       : launch_stream_t<32, 1, STREAM_DIST, false, false>(a, s);
    return hipErrorInvalidValue;
}

// ------------------------------------------------------------------------------------
// probe_rows_kernel: the centroid probe of a BATCH (src/ivf/index.rs:130-149 for every query), dim % 4 == 0.
//
// stream_kernel reads the whole centroid table once per query (C3: 1024 queries x 3 MB through L2, 240 us) and
// re-creates each row's serial chain through an LDS transpose.  Here a LANE owns a centroid and keeps the chains of QB
// queries in registers: the table is read from a [dim/4][kc_pad] float4 transpose (64 consecutive centroids = one 1 KiB
// load), the queries are wave-uniform and arrive as scalar operands, so a float4 of a row serves QB queries and the
// kernel is bound by the 12 VALU operations per (query, centroid, float4) of the reference arithmetic
//   t = ((d0^2 + d1^2) + d2^2) + d3^2;  sum = sum + t        (index.rs:461-472, no FMA)
// grid = (ceil(kc / 256), ceil(nq / QB)); the 4 waves of a block take 4 runs of 64 centroids for the same QB queries.
// Output: UNSORTED partial lists [nq][4 * gridDim.x][64] of (distance bits << 32 | centroid, centroid) for merge_kernel.
// ------------------------------------------------------------------------------------
template <int QB>
__global__ __launch_bounds__(256) void probe_rows_kernel(const ProbeRowsArgs a) {
    if (a.zero_u32 && blockIdx.x == 0) {
        for (uint32_t i = blockIdx.y * 256 + threadIdx.x; i < a.zero_n; i += gridDim.y * 256) a.zero_u32[i] = 0u;
    }
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const uint32_t part = blockIdx.x * 4 + wave;
    const uint32_t c = part * 64 + lane;                     // < kc_pad (the transpose is padded with zero rows)
    const uint32_t q0 = blockIdx.y * QB;
    const uint32_t G = a.dim >> 2;
    const float4 *xt = a.cent_t + c;
    const float *qv[QB];
#pragma unroll
    for (int u = 0; u < QB; ++u) qv[u] = a.queries + (uint64_t)(q0 + u < a.nq ? q0 + u : a.nq - 1) * a.dim;
    float sum[QB];
#pragma unroll
    for (int u = 0; u < QB; ++u) sum[u] = 0.0f;
    if (part * 64 < a.kc_pad) {
#pragma unroll 2
        for (uint32_t g = 0; g < G; ++g) {
            const float4 x = xt[(uint64_t)g * a.kc_pad];
#pragma unroll
            for (int u = 0; u < QB; ++u) {
                const float4 qq = load4_uniform<true>(qv[u] + g * 4);
                const float d0 = qq.x - x.x, d1 = qq.y - x.y, d2 = qq.z - x.z, d3 = qq.w - x.w;
                float t = d0 * d0 + d1 * d1;
                t = t + d2 * d2;
                t = t + d3 * d3;
                sum[u] = sum[u] + t;
            }
        }
    }
    const uint32_t n_part = gridDim.x * 4;
#pragma unroll
    for (int u = 0; u < QB; ++u) {
        if (q0 + u < a.nq) {
            const uint64_t o = ((uint64_t)(q0 + u) * n_part + part) * 64 + lane;
            a.part_keys[o] = c < a.kc ? (((uint64_t)__float_as_uint(sum[u]) << 32) | c) : KEY_EMPTY;
            a.part_vals[o] = c < a.kc ? c : 0xFFFFFFFFu;
        }
    }
}
hipError_t launch_probe_rows(const ProbeRowsArgs &a, hipStream_t s) {
    if (a.nq == 0 || a.kc == 0) return hipSuccess;
    if ((a.dim % 4) != 0 || (a.kc_pad % 64) != 0 || a.kc_pad < a.kc) return hipErrorInvalidValue;
    const uint32_t gx = (a.kc + 255) / 256;
    // enough waves for the chip first (four per SIMD: the loop waits for every row chunk it loads -- C3, 1024 queries:
    // 83 us with 4 queries per lane and 4096 waves, 106 us with 8 and 2048; an explicit prefetch of the next chunk
    // measured slower), then as many queries per row read as the batch allows
    const int qb = (uint64_t)gx * (a.nq / 8) >= 1024 ? 8 : (uint64_t)gx * (a.nq / 4) >= 512 ? 4 : (uint64_t)gx * (a.nq / 2) >= 256 ? 2 : 1;
    const dim3 grid(gx, (a.nq + qb - 1) / qb);
    switch (qb) {
    case 8: hipLaunchKernelGGL(probe_rows_kernel<8>, grid, dim3(256), 0, s, a); break;
    case 4: hipLaunchKernelGGL(probe_rows_kernel<4>, grid, dim3(256), 0, s, a); break;
    case 2: hipLaunchKernelGGL(probe_rows_kernel<2>, grid, dim3(256), 0, s, a); break;
    default: hipLaunchKernelGGL(probe_rows_kernel<1>, grid, dim3(256), 0, s, a); break;
    }
    return hipGetLastError();
}
// cent_t[g * kc_pad + c] = float4 g of centroid c (zero rows for c >= kc)
__global__ __launch_bounds__(256) void transpose_rows4_kernel(const float *__restrict__ rows, uint32_t kc, uint32_t kc_pad, uint32_t dim,
                                                            float4 *__restrict__ out) {
    const uint32_t G = dim >> 2;
    const uint64_t total = (uint64_t)G * kc_pad;
    for (uint64_t i = (uint64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (uint64_t)gridDim.x * 256) {
        const uint32_t g = (uint32_t)(i / kc_pad), c = (uint32_t)(i % kc_pad);
        out[i] = c < kc ? *reinterpret_cast<const float4 *>(rows + (uint64_t)c * dim + g * 4) : make_float4(0.f, 0.f, 0.f, 0.f);
    }
}
hipError_t launch_transpose_rows4(const float *rows, uint32_t kc, uint32_t kc_pad, uint32_t dim, void *out, hipStream_t s) {
    if (kc == 0 || (dim % 4) != 0) return hipErrorInvalidValue;
    const uint64_t total = (uint64_t)(dim / 4) * kc_pad;
    const uint32_t blocks = (uint32_t)std::min<uint64_t>((total + 255) / 256, 4096);
    hipLaunchKernelGGL(transpose_rows4_kernel, dim3(blocks), dim3(256), 0, s, rows, kc, kc_pad, dim, static_cast<float4 *>(out));
    return hipGetLastError();
}

// int8 image of one (query, probed list) pair by one wave (quantize_pairs_i8_kernel): the RESIDUAL v = q - centre of the
// pair's list at that list's scale (see block_rows_i8_kernel for the row side and the bound).
//   vi      = clamp(rint(v S), -127, 127)
//   q_res   >= |clamp_box(v) - vi / S|   (box = [-127 / S, 127 / S]^dim, where every row image lives): clamping a query
//              component towards the box can only SHRINK its distance to a point inside the box, so the LOWER bound
//              |q - x| >= |vi - xi| / S - q_res - rx stays rigorous with the rounding residual alone -- a far-away query keeps
//              a tight bound instead of being "never skipped"
//   q_resu  >= |v - vi / S|               (rounding + what the clamp cut off): the residual of the UPPER bounds (thresholds)
//   pair_lb <= every reference d2(q, x), x in the list: (|v| - radius)^2 by the triangle inequality on the list's centre,
//              with the summation margin of the reference order taken off; 0 = no information
__device__ __forceinline__ void quantize_pair_i8_wave(const PairQuantArgs &a, uint32_t p, uint32_t c, uint32_t q, int lane);
__device__ __forceinline__ void quantize_pair_i8_wave(const PairQuantArgs &a, uint32_t p, int lane) {
    // (probe == nullptr: ONE image per query -- every list then shares centre and scale, entry 0 of the tables)
    quantize_pair_i8_wave(a, p, a.probe ? a.probe[p] : 0u, a.probe ? p / a.nprobe : p, lane);
}
__device__ __forceinline__ void quantize_pair_i8_wave(const PairQuantArgs &a, uint32_t p, uint32_t c, uint32_t q, int lane) {
    const float scale = a.scale[c], inv = 1.0f / scale, box = 127.0f * inv;
    const float *qv = a.queries + (uint64_t)q * a.dim, *cv = a.center + (uint64_t)c * a.dim;
    int n2 = 0;
    float e2 = 0.0f, u2 = 0.0f, v2 = 0.0f, big = 0.0f;
    bool bad = false;
    for (uint32_t d0 = lane * 4; d0 < a.dim; d0 += 256) {
        const float4 x = *reinterpret_cast<const float4 *>(qv + d0);
        const float4 cx = *reinterpret_cast<const float4 *>(cv + d0);
        const float t[4] = {x.x - cx.x, x.y - cx.y, x.z - cx.z, x.w - cx.w};
        uint32_t w = 0;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            bad |= !(fabsf(t[e]) < INFINITY);
            const int v = quant_i8(t[e], scale);
            const float rec = (float)v * inv;
            const float res = fminf(fmaxf(t[e], -box), box) - rec, resu = t[e] - rec;
            n2 += v * v;
            e2 = fmaf(res, res, e2);
            u2 = fmaf(resu, resu, u2);
            v2 = fmaf(t[e], t[e], v2);
            big = fmaxf(big, fabsf(t[e]));
            w |= (uint32_t)(v & 0xFF) << (8 * e);
        }
        *reinterpret_cast<uint32_t *>(a.q_i8 + (uint64_t)p * a.dim + d0) = w;
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
        n2 += __shfl_xor(n2, off, 64);
        e2 += __shfl_xor(e2, off, 64);
        u2 += __shfl_xor(u2, off, 64);
        v2 += __shfl_xor(v2, off, 64);
        big = fmaxf(big, __shfl_xor(big, off, 64));
    }
    bad = __ballot(bad) != 0ull;
    if (lane == 0) {
        a.q_n2i[p] = n2;
        // + the roundings of (q - c) and vi / S inside every residual
        const float pad = 4.0f * 5.9604645e-08f * sqrtf((float)a.dim) * (big + a.half[c] + box);
        const float r = sqrtf(e2) * 1.001f + pad, ru = sqrtf(u2) * 1.001f + pad;
        a.q_res[p] = (bad || !(r < INFINITY)) ? INFINITY : r;
        a.q_resu[p] = (bad || !(ru < INFINITY)) ? INFINITY : ru;
        const float cmargin = (float)(a.dim + 16) * 2.384185791015625e-07f;
        if (a.probe) {
            const float lbd = sqrtf(v2) * 0.99998f - a.radius[c];
            const float lb = lbd > 0.0f ? lbd * lbd * (1.0f - 2.0f * cmargin) * 0.99999f : 0.0f;
            a.pair_lb[p] = (bad || !(lb < INFINITY)) ? 0.0f : lb;
        }
    }
}
__global__ __launch_bounds__(256) void quantize_pairs_i8_kernel(const PairQuantArgs a) {
    const uint32_t p = blockIdx.x * 4u + (threadIdx.x >> 6);
    if (p < a.n_pairs) quantize_pair_i8_wave(a, p, (int)(threadIdx.x & 63));
}
hipError_t launch_quantize_pairs_i8(const float *queries, const uint32_t *probe, const float *center, const float *scale, const float *half,
                                    const float *radius, uint32_t n_pairs, uint32_t nprobe, uint32_t dim, void *q_i8, int *q_n2i,
                                    float *q_res, float *q_resu, float *pair_lb, hipStream_t s) {
    if (n_pairs == 0) return hipSuccess;
    if (dim % 4 || nprobe == 0 || (probe && !pair_lb)) return hipErrorInvalidValue;
    PairQuantArgs a{queries, probe, center, scale, half, radius, n_pairs, nprobe, dim, static_cast<int8_t *>(q_i8), q_n2i, q_res, q_resu, pair_lb};
    hipLaunchKernelGGL(quantize_pairs_i8_kernel, dim3((n_pairs + 3) / 4), dim3(256), 0, s, a);
    return hipGetLastError();
}

// ------------------------------------------------------------------------------------
// merge_kernel: one wave per query folds all partial lists.
// PROBE == false: final results (row ids via ids[], sqrt optional, search.rs:129-141).
// PROBE == true : the k "rows" are centroids; emits the probe order and the candidate
//                 position base of each probed list (index.rs:57-63's concatenation).
// ------------------------------------------------------------------------------------
// probe merge, waves 1 .. (threads >= 64): the query's partial lists of the re-rank start EMPTY; a single-query call also
// gets the int8 image of its query (last wave of the block)
__device__ __forceinline__ void probe_merge_helpers(const MergeArgs &a, uint32_t q) {
    if (a.preset_keys) {
        uint64_t *pk = a.preset_keys + (uint64_t)q * a.preset_n;
        uint32_t *pv = a.preset_vals + (uint64_t)q * a.preset_n;
        for (uint32_t i = threadIdx.x - 64; i < a.preset_n; i += blockDim.x - 64) { pk[i] = KEY_EMPTY; pv[i] = 0xFFFFFFFFu; }
    }
    if (a.preset_flags) {
        uint32_t *pf = reinterpret_cast<uint32_t *>(a.preset_flags + (uint64_t)q * a.preset_flag_n);      // preset_flag_n % 4 == 0
        for (uint32_t i = threadIdx.x - 64; i < a.preset_flag_n / 4; i += blockDim.x - 64) pf[i] = 0u;
    }
}
// |q|^2 for the MFMA screen (any order) and max |q_i| (f16 operand range check): one wave
__device__ __forceinline__ void probe_query_norms(const MergeArgs &a, uint32_t q, int lane) {
    if (a.qnorm_out || a.qmax_out) {      // |q|^2 for the MFMA screen (any order) and max |q_i| (f16 operand range check):
        float acc = 0.0f, m = 0.0f;       // one pass, all of a lane's loads in flight together (up to 8 x 16 bytes)
        const float *qp = a.queries + (uint64_t)q * a.dim;
        if ((a.dim % 4u) == 0u) {
            for (uint32_t d0 = (uint32_t)lane * 4u; d0 < a.dim; d0 += 2048u) {
                float4 v[8];
#pragma unroll
                for (int u = 0; u < 8; ++u) {
                    const uint32_t d = d0 + 256u * (uint32_t)u;
                    v[u] = d < a.dim ? *reinterpret_cast<const float4 *>(qp + d) : make_float4(0.f, 0.f, 0.f, 0.f);
                }
#pragma unroll
                for (int u = 0; u < 8; ++u) {
                    acc += v[u].x * v[u].x; acc += v[u].y * v[u].y; acc += v[u].z * v[u].z; acc += v[u].w * v[u].w;
                    m = fmaxf(fmaxf(m, fmaxf(fabsf(v[u].x), fabsf(v[u].y))), fmaxf(fabsf(v[u].z), fabsf(v[u].w)));
                }
            }
        } else {
            for (uint32_t d = lane; d < a.dim; d += 64) { const float v = qp[d]; acc += v * v; m = fmaxf(m, fabsf(v)); }
        }
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) { acc += __shfl_down(acc, off, 64); m = fmaxf(m, __shfl_down(m, off, 64)); }
        if (lane == 0 && a.qnorm_out) a.qnorm_out[q] = acc;
        if (lane == 0 && a.qmax_out) a.qmax_out[q] = m;
    }
}
// probe merge, wave 0 after the selection: probe order, candidate bases, histogram / single-query bucketing, norms
template <int S>
__device__ __forceinline__ void probe_merge_tail(const MergeArgs &a, uint32_t q, int lane, WaveTopk<S> &tk, bool norms = true) {
    uint64_t carry = 0;
#pragma unroll
    for (int s = 0; s < S; ++s) {
        const uint32_t e = s * 64 + lane;
        const bool have = e < a.k && tk.key[s] != KEY_EMPTY;
        const uint32_t c = have ? tk.val[s] : 0;
        const uint64_t len = have ? (a.list_off[c + 1] - a.list_off[c]) : 0;
        // inclusive wave scan of len
        uint64_t incl = len;
#pragma unroll
        for (int off = 1; off < 64; off <<= 1) {
            const uint32_t lo = (uint32_t)__shfl_up((int)(uint32_t)incl, off, 64);
            const uint32_t hi = (uint32_t)__shfl_up((int)(uint32_t)(incl >> 32), off, 64);
            const uint64_t o = ((uint64_t)hi << 32) | lo;
            if (lane >= off) incl += o;
        }
        if (e < a.k) {
            a.probe[(uint64_t)q * a.k + e] = c;
            a.cand_base[(uint64_t)q * a.k + e] = carry + incl - len;
            if (a.hist && have && !a.sq_quads) atomicAdd(&a.hist[(uint64_t)(q % HIST_REPLICAS) * a.hist_stride + c], 1u);   // pair bucketing: cluster histogram
        }
        if (a.sq_quads && s == 0) {          // single query (q == 0, a.k <= 64): quad e = pair e = probe rank e
            const uint32_t nch = (have && a.sq_item_rows) ? (uint32_t)((len + a.sq_item_rows - 1) / a.sq_item_rows) : 0u;
            uint32_t ii = nch;
#pragma unroll
            for (int off = 1; off < 64; off <<= 1) {
                const uint32_t o = (uint32_t)__shfl_up((int)ii, off, 64);
                if (lane >= off) ii += o;
            }
            const uint32_t first = ii - nch;
            if (have) {
                a.sq_quads[e] = make_uint4(c, e, 1u, first);
                a.sq_pairs[e] = e;
                for (uint32_t t = 0; t < nch && first + t < a.sq_max_items; ++t) a.sq_item_quad[first + t] = e;
            }
            const uint32_t nqd = (uint32_t)__popcll(__ballot(have));
            const uint32_t nit = readlane_u32(ii, 63);
            if (lane == 0) {
                *a.sq_n_quads = nqd;
                if (a.sq_n_items) *a.sq_n_items = nit < a.sq_max_items ? nit : a.sq_max_items;
            }
        }
        carry += readlane_u64(incl, 63);
    }
    if (a.n_cand && lane == 0) a.n_cand[q] = carry;   // uncapped: candidate_rows metric
    if (a.stats && lane == 0) {                       // plan metrics, spread over STATS_SLOTS lines
#ifdef PQV_PROFILE_PHASES
        unsigned long long *st = a.stats;
#else
        unsigned long long *st = a.stats + 8 + 16 * (q % STATS_SLOTS);
#endif
        atomicAdd(&st[2], (unsigned long long)carry);
        atomicAdd(&st[3], (unsigned long long)(carry < a.max_pos ? carry : a.max_pos));
    }
    if (a.gthr_init && lane == 0) a.gthr_init[q] = ~0ull;          // per-query admission threshold: none yet
    if (norms) probe_query_norms(a, q, lane);
}

// ------------------------------------------------------------------------------------
// Deferred exact evaluation (TileArgs::cand_lb, wide_filter_kernel): the query's candidate buffer holds survivors of the screen
// with their distance BOUNDS -- key = (upper bound << 32 | position), cand_lb = lower bound, or cand_lb < 0 and an exact key.
// resolve_select_kernel, a block of 256 threads per query, ahead of the final merge:
//   1. T = the k-th smallest upper bound (4-pass radix select over the bound bits; +inf with fewer than k entries).  k rows
//      have a reference distance <= T, so the k-th smallest reference distance is <= T.
//   1b. the entries that define T (upper bound <= T) are evaluated by the block -- in the reference's order: chunks of 4
//      values, ((d0^2 + d1^2) + d2^2) + d3^2 added to ONE running sum in chunk order (index.rs:461-480); L lanes share a row,
//      the sum passes through them in order -- and T is taken again over exact distances where they exist.
//   2. an entry whose lower bound exceeds T cannot be among the k nearest (not even tied with the k-th): dropped; the others go
//      to the call's work list, resolve_exact_kernel gives them exact keys on the whole chip, and the merge then sees what the
//      in-filter evaluation would have left, minus rows that cannot matter.
// ------------------------------------------------------------------------------------
template <int NB>
__device__ __forceinline__ void resolve_exact(const MergeArgs &a, uint32_t q, const uint16_t *band, uint32_t m, uint64_t *ck, const uint32_t *cv,
                                              int lane, int wave, uint32_t lg, float *cl = nullptr /* marks the entry exact */) {
    const uint32_t Gx = a.dim >> 2;
    const uint32_t L = 1u << lg, per_wave = 64u >> lg;
    const uint32_t pl = (uint32_t)lane >> lg, pj = (uint32_t)lane & (L - 1u);
    const uint32_t first = (uint32_t)lane & ~(L - 1u);
    const float4 *qg = reinterpret_cast<const float4 *>(a.queries + (uint64_t)q * a.dim);
    for (uint32_t p0 = (uint32_t)wave * per_wave; p0 < m; p0 += 4u * per_wave) {
        const uint32_t pi = p0 + pl;
        const bool valid = pi < m;
        const uint32_t idx = band[valid ? pi : p0];
        const float *x = a.mat + (uint64_t)cv[idx] * a.dim;
        float sum = 0.0f;
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
                for (int u = 0; u < NB; ++u) if (g + u < Gx) sum = sum + tt[u];
            } else {
                for (uint32_t sl = 0; sl < L; ++sl) {
                    float sn = sum;
#pragma unroll
                    for (int u = 0; u < NB; ++u) sn = sn + tt[u];
                    sum = __shfl(pj == sl ? sn : sum, (int)(first + sl), 64);
                }
            }
        }
        if (valid && pj == 0u) {
            ck[idx] = ((uint64_t)__float_as_uint(sum) << 32) | (uint64_t)(uint32_t)ck[idx];
            if (cl) cl[idx] = -1.0f;
        }
    }
}
// (the band goes to a call-wide work list {query, entry} for resolve_exact_kernel)
__device__ __forceinline__ void resolve_candidates(const MergeArgs &a, uint32_t q, uint2 *work, uint32_t *n_work) {
    __shared__ uint32_t s_hist[256];
    __shared__ uint32_t s_sel[4];            // {prefix, remaining rank, band count}
    __shared__ uint16_t s_band[8192];
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    uint32_t n = a.cand_cnt[q];
    if (n > a.cand_cap) n = a.cand_cap;
    uint64_t *ck = a.cand_keys_rw + (uint64_t)q * a.cand_cap;
    const uint32_t *cv = a.cand_vals + (uint64_t)q * a.cand_cap;
    float *cl = a.cand_lb + (uint64_t)q * a.cand_cap;
    if (n == 0) return;
    // 1. the k-th smallest upper bound (bounds are >= 0: their bits order like unsigned integers)
    uint32_t T = 0x7F800000u;
    if (n >= a.k) T = block_kth_u32([&](uint32_t i) { return (uint32_t)(ck[i] >> 32); }, n, a.k, s_hist, s_sel);
    auto exact_list = [&](uint32_t m, float *mark) {
        const uint32_t Gx = a.dim >> 2;
        const uint32_t per = (m + 3u) / 4u;                    // pairs per wave
        uint32_t lg = 0;
        if ((Gx % 64u) == 0u) {
            while (lg < 3 && (per << (lg + 1)) <= 64u) ++lg;
            resolve_exact<8>(a, q, s_band, m, ck, cv, lane, wave, lg, mark);
        } else if ((Gx % 16u) == 0u) {
            while (lg < 3 && (per << (lg + 1)) <= 64u) ++lg;
            resolve_exact<2>(a, q, s_band, m, ck, cv, lane, wave, lg, mark);
        } else {
            resolve_exact<8>(a, q, s_band, m, ck, cv, lane, wave, 0u, mark);
        }
        if (a.resolve_stats && threadIdx.x == 0) atomicAdd(&a.resolve_stats[8 + 16 * (q % STATS_SLOTS) + 1], (unsigned long long)m);
    };
    // 1b. (batches) the entries that DEFINE T -- the k smallest upper bounds, nearly the k nearest rows -- are evaluated right
    //     here, and T is taken again over exact distances where they exist: it drops from "k-th distance + the bound's whole
    //     width" to "+ what the k-th nearest row's own bound leaves", and the band below loses the entries whose lower bound
    //     lies in between.  Skipped when a tie group makes that list long.
    if (a.resolve_two_cuts && n >= a.k) {
        if (threadIdx.x == 0) s_sel[2] = 0;
        __syncthreads();
        for (uint32_t i = threadIdx.x; i < n; i += 256)
            if (cl[i] >= 0.0f && (uint32_t)(ck[i] >> 32) <= T) {
                const uint32_t slot = atomicAdd(&s_sel[2], 1u);
                if (slot < 8192u) s_band[slot] = (uint16_t)i;
            }
        __syncthreads();
        const uint32_t mA = s_sel[2];
        __syncthreads();
        if (mA && mA <= 2u * a.k + 64u) {
            exact_list(mA, cl);
            __syncthreads();
            T = block_kth_u32([&](uint32_t i) { return (uint32_t)(ck[i] >> 32); }, n, a.k, s_hist, s_sel);
        }
    }
    // 2. the band: deferred entries whose lower bound does not exceed T; exact entries beyond T and deferred ones outside the
    //    band leave (KEY_EMPTY)
    if (threadIdx.x == 0) s_sel[2] = 0;
    __syncthreads();
    const float Tf = __uint_as_float(T);
    for (uint32_t i = threadIdx.x; i < n; i += 256) {
        const float lb = cl[i];
        const uint32_t ub = (uint32_t)(ck[i] >> 32);
        if (lb < 0.0f) {
            if (ub > T) ck[i] = KEY_EMPTY;
        } else if (lb <= Tf) {
            s_band[atomicAdd(&s_sel[2], 1u)] = (uint16_t)i;
        } else {
            ck[i] = KEY_EMPTY;
        }
    }
    __syncthreads();
    const uint32_t m = s_sel[2];
    if (m) {
        if (threadIdx.x == 0) s_sel[3] = atomicAdd(n_work, m);
        __syncthreads();
        const uint32_t base = s_sel[3];
        for (uint32_t i = threadIdx.x; i < m; i += 256) work[base + i] = make_uint2(q, (uint32_t)s_band[i]);
        if (a.resolve_stats && threadIdx.x == 0) atomicAdd(&a.resolve_stats[8 + 16 * (q % STATS_SLOTS) + 1], (unsigned long long)m);
    }
}

__global__ __launch_bounds__(256) void resolve_select_kernel(const MergeArgs a, uint2 *work, uint32_t *n_work) {
    resolve_candidates(a, blockIdx.x, work, n_work);
}
// the batch's band entries, 8 lanes per pair (Gx % 16 == 0: every path that defers has dim % 64 == 0), grid-stride
template <int NB>
__global__ __launch_bounds__(256) void resolve_exact_kernel(const MergeArgs a, const uint2 *work, const uint32_t *n_work) {
    const int lane = threadIdx.x & 63;
    const uint32_t gw = blockIdx.x * 4u + (threadIdx.x >> 6), nw = gridDim.x * 4u;
    const uint32_t m = *n_work;
    const uint32_t Gx = a.dim >> 2;
    const uint32_t pl = (uint32_t)lane >> 3, pj = (uint32_t)lane & 7u, first = (uint32_t)lane & ~7u;
    for (uint32_t p0 = gw * 8u; p0 < m; p0 += nw * 8u) {
        const uint32_t pi = p0 + pl;
        const bool valid = pi < m;
        const uint2 w = work[valid ? pi : p0];
        uint64_t *ck = a.cand_keys_rw + (uint64_t)w.x * a.cand_cap + w.y;
        const float *x = a.mat + (uint64_t)a.cand_vals[(uint64_t)w.x * a.cand_cap + w.y] * a.dim;
        const float4 *qg = reinterpret_cast<const float4 *>(a.queries + (uint64_t)w.x * a.dim);
        float sum = 0.0f;
        for (uint32_t g0 = 0; g0 < Gx; g0 += NB * 8) {
            const uint32_t g = g0 + NB * pj;
            float4 xv[NB], qv[NB];
#pragma unroll
            for (int u = 0; u < NB; ++u) { xv[u] = load4<true>(x + (g + u) * 4); qv[u] = qg[g + u]; }
            float tt[NB];
#pragma unroll
            for (int u = 0; u < NB; ++u) {
                const float d0 = qv[u].x - xv[u].x, d1 = qv[u].y - xv[u].y, d2 = qv[u].z - xv[u].z, d3 = qv[u].w - xv[u].w;
                float t = d0 * d0 + d1 * d1;
                t = t + d2 * d2;
                tt[u] = t + d3 * d3;
            }
            for (uint32_t sl = 0; sl < 8u; ++sl) {            // the reference's one running sum passes through the pair's lanes in order
                float sn = sum;
#pragma unroll
                for (int u = 0; u < NB; ++u) sn = sn + tt[u];
                sum = __shfl(pj == sl ? sn : sum, (int)(first + sl), 64);
            }
        }
        if (valid && pj == 0u) *ck = ((uint64_t)__float_as_uint(sum) << 32) | (uint64_t)(uint32_t)*ck;
    }
}
// Resolve a batch's deferred evaluations in two launches (selection per query, then every band entry of the batch on the
// whole chip); the final merge then runs without a.cand_lb.  n_work: one u32, zeroed here.
hipError_t launch_resolve(const MergeArgs &a, void *work, uint32_t *n_work, bool n_work_is_zero, hipStream_t s) {
    if (a.nq == 0) return hipSuccess;
    if (!a.cand_lb || a.cand_cap > 8192 || !a.cand_keys_rw || !a.mat || !a.queries || (a.dim % 64) != 0 || !work || !n_work) return hipErrorInvalidValue;
    if (!n_work_is_zero) {              // (otherwise the previous call's final merge left it cleared: MergeArgs::zero_after)
        hipError_t e = hipMemsetAsync(n_work, 0, sizeof(uint32_t), s);
        if (e != hipSuccess) return e;
    }
    hipLaunchKernelGGL(resolve_select_kernel, dim3(a.nq), dim3(256), 0, s, a, static_cast<uint2 *>(work), n_work);
    const uint32_t blocks = (uint32_t)std::min<uint64_t>(4096, ((uint64_t)a.nq * 64 + 31) / 32 + 256);
    if ((a.dim % 256) == 0) hipLaunchKernelGGL(resolve_exact_kernel<8>, dim3(blocks), dim3(256), 0, s, a, static_cast<const uint2 *>(work), n_work);
    else hipLaunchKernelGGL(resolve_exact_kernel<2>, dim3(blocks), dim3(256), 0, s, a, static_cast<const uint2 *>(work), n_work);
    return hipGetLastError();
}

// Final merge, k > 64, nothing spilled: the k smallest of a buffer of a few thousand exact keys.  Serial insertion into the
// wave-distributed list costs ~3 k inserts of S ballots each (K = 100: 138 us); instead the block radix-selects the k-th
// smallest DISTANCE, compacts the keys at or below it (k + ties) into LDS, ranks them against each other and wave 0 picks
// them up in order.  false: more than 64 S keys at or below the k-th distance (a huge tie group) -- the caller inserts.
template <int S>
__device__ __forceinline__ bool merge_select_large(const MergeArgs &a, uint32_t q, WaveTopk<S> &tk) {
    constexpr uint32_t CAPK = 64u * S;
    __shared__ uint32_t s_h[256];
    __shared__ uint32_t s_s[4];
    __shared__ uint64_t s_k[CAPK], s_k2[CAPK];
    __shared__ uint32_t s_v[CAPK], s_v2[CAPK];
    const int lane = threadIdx.x & 63;
    uint32_t n = a.cand_cnt[q];
    if (n > a.cand_cap) n = a.cand_cap;
    const uint64_t *ck = a.cand_keys + (uint64_t)q * a.cand_cap;
    const uint32_t *cv = a.cand_vals + (uint64_t)q * a.cand_cap;
    if (threadIdx.x == 0) { s_s[2] = 0u; s_s[3] = 0u; }
    __syncthreads();
    uint32_t nv = 0;
    for (uint32_t i = threadIdx.x; i < n; i += blockDim.x) nv += ck[i] != KEY_EMPTY ? 1u : 0u;
    if (nv) atomicAdd(&s_s[3], nv);
    __syncthreads();
    const uint32_t nvalid = s_s[3];
    uint32_t m = 0;
    if (nvalid) {
        const uint32_t kk = a.k < nvalid ? a.k : nvalid;
        const uint32_t Td = block_kth_u32([&](uint32_t i) { return (uint32_t)(ck[i] >> 32); }, n, kk, s_h, s_s);     // (KEY_EMPTY sorts last)
        for (uint32_t i = threadIdx.x; i < n; i += blockDim.x) {
            const uint64_t key = ck[i];
            if (key != KEY_EMPTY && (uint32_t)(key >> 32) <= Td) {
                const uint32_t slot = atomicAdd(&s_s[2], 1u);
                if (slot < CAPK) { s_k[slot] = key; s_v[slot] = cv[i]; }
            }
        }
        __syncthreads();
        m = s_s[2];
        if (m > CAPK) return false;
        for (uint32_t e = threadIdx.x; e < m; e += blockDim.x) {          // keys are distinct (the position is part of them)
            const uint64_t key = s_k[e];
            uint32_t rank = 0;
            for (uint32_t j = 0; j < m; ++j) rank += s_k[j] < key ? 1u : 0u;
            s_k2[rank] = key; s_v2[rank] = s_v[e];
        }
        __syncthreads();
    }
    if (threadIdx.x < 64) {
#pragma unroll
        for (int s = 0; s < S; ++s) {
            const uint32_t e = (uint32_t)(s * 64 + lane);
            tk.key[s] = e < m ? s_k2[e] : KEY_EMPTY;
            tk.val[s] = e < m ? s_v2[e] : 0xFFFFFFFFu;
        }
    }
    return true;
}

template <int S, bool PROBE>
__global__ __launch_bounds__(256) void merge_kernel(const MergeArgs a) {
    const int lane = threadIdx.x & 63;
    const uint32_t q = blockIdx.x;
    if constexpr (!PROBE) PQV_STAMP_MIN(24);
    if constexpr (!PROBE) { if (a.zero_after && blockIdx.x == 0 && threadIdx.x == 0) *a.zero_after = 0u; }
    WaveTopk<S> tk;
    tk.init();
    [[maybe_unused]] bool preselected = false;
    if constexpr (!PROBE && S > 1 && S <= 4) {
        if (a.cand_keys && blockDim.x == 256 && a.spilled && a.spilled[q] == 0) preselected = merge_select_large<S>(a, q, tk);
    }
    if (threadIdx.x >= 64) {
        // helper waves (probe mode with a preset only): the query's partial lists of the re-rank start EMPTY
        if constexpr (PROBE) probe_merge_helpers(a, q);
        return;
    }
    const uint64_t total = (uint64_t)a.n_part * a.k_part;
    const uint64_t *pk = a.part_keys + (uint64_t)q * total;
    const uint32_t *pv = a.part_vals + (uint64_t)q * total;
    // with a candidate buffer the partial lists hold something only if the query overflowed it
    const uint64_t scan = (a.cand_keys && a.spilled && a.spilled[q] == 0) ? 0 : total;
    // pre-filter (k <= 64, plain scans): the k-th smallest of the 64 lane minima bounds the k-th smallest overall, so only
    // keys at or below it reach the serial insertion (a few dozen instead of a few hundred); the keys are read twice
    // (L2) for that
    uint64_t cut = KEY_EMPTY;
    uint32_t ncand = 0;
    if (a.cand_keys) { ncand = a.cand_cnt[q]; if (ncand > a.cand_cap) ncand = a.cand_cap; }
    bool folded = preselected;
    if constexpr (S == 1 && !PROBE) {
        // the usual final merge of the wide screened path: nothing spilled, up to 1024 candidates, k <= 64 -- keys and values
        // in ONE round trip (16 per lane), the k-th lane minimum as a cut, and the handful that pass it ordered by rank
        // counting instead of being inserted one by one
        if (a.cand_keys && scan == 0 && ncand <= 1024u && a.k <= 64u) {
            __shared__ uint64_t s_mk[128];
            __shared__ uint32_t s_mv[128];
            const uint64_t *ck = a.cand_keys + (uint64_t)q * a.cand_cap;
            const uint32_t *cv = a.cand_vals + (uint64_t)q * a.cand_cap;
            uint64_t kreg[16];
            uint32_t vreg[16];
#pragma unroll
            for (int u = 0; u < 16; ++u) {
                const uint32_t idx = 64u * (uint32_t)u + (uint32_t)lane;
                kreg[u] = idx < ncand ? ck[idx] : KEY_EMPTY;
                vreg[u] = idx < ncand ? cv[idx] : 0xFFFFFFFFu;
            }
            uint64_t lmin = KEY_EMPTY;
#pragma unroll
            for (int u = 0; u < 16; ++u) lmin = kreg[u] < lmin ? kreg[u] : lmin;
            const uint64_t cut2 = wave_kth_by_rank(lmin, a.k, lane, s_mk);
            uint64_t sk = KEY_EMPTY;
            uint32_t sv = 0xFFFFFFFFu;
            if (wave_select_by_sort_kv<16>(kreg, vreg, cut2, lane, s_mk, s_mv, sk, sv)) {
                tk.key[0] = (uint32_t)lane < a.k ? sk : KEY_EMPTY;
                tk.val[0] = (uint32_t)lane < a.k ? sv : 0xFFFFFFFFu;
            } else {
#pragma unroll
                for (int u = 0; u < 16; ++u) {
                    uint64_t key = kreg[u];
                    if (key > cut2) key = KEY_EMPTY;
                    if (__ballot(key != KEY_EMPTY) != 0ull) tk.offer(key, vreg[u], a.k, lane);
                }
            }
            folded = true;
        }
    }
    if (folded) {
    } else {
    if (S == 1 && !(a.part_flags && scan) && scan + ncand > 128) {
        uint64_t lmin = KEY_EMPTY;
        for (uint64_t i = lane; i < scan; i += 64) { const uint64_t key = pk[i]; lmin = key < lmin ? key : lmin; }
        const uint64_t *ck = a.cand_keys + (uint64_t)q * a.cand_cap;
        for (uint32_t i = lane; i < ncand; i += 64) { const uint64_t key = ck[i]; lmin = key < lmin ? key : lmin; }
        uint32_t dummy = 0;
        bitonic_sort64(lmin, dummy, lane);
        cut = a.k <= 64u ? readlane_u64(lmin, (int)a.k - 1) : KEY_EMPTY;
    }
    if (a.part_flags && scan) {
        // only the lists some wave has written (a handful, and only for a query whose candidate buffer overflowed)
        const uint8_t *fl = a.part_flags + (uint64_t)q * a.n_part;
        for (uint32_t l0 = 0; l0 < a.n_part; l0 += 64) {
            unsigned long long m = __ballot(l0 + lane < a.n_part && fl[l0 + lane] != 0);
            while (m) {
                const uint32_t li = l0 + (uint32_t)__builtin_ctzll(m);
                m &= m - 1;
                for (uint32_t e0 = 0; e0 < a.k_part; e0 += 64) {
                    const uint32_t e = e0 + lane;
                    uint64_t key = KEY_EMPTY;
                    uint32_t val = 0xFFFFFFFFu;
                    if (e < a.k_part) { key = pk[(uint64_t)li * a.k_part + e]; val = pv[(uint64_t)li * a.k_part + e]; }
                    tk.offer(key, val, a.k, lane);
                }
            }
        }
    } else
    for (uint64_t i = 0; i < scan; i += 64) {
        const uint64_t idx = i + lane;
        uint64_t key = KEY_EMPTY;
        uint32_t val = 0xFFFFFFFFu;
        if (idx < scan) { key = pk[idx]; val = pv[idx]; }
        if (key > cut) key = KEY_EMPTY;
        if (__ballot(key != KEY_EMPTY) != 0ull) tk.offer(key, val, a.k, lane);
    }
    if (a.cand_keys) {
        uint32_t n = a.cand_cnt[q];
        if (n > a.cand_cap) n = a.cand_cap;
        const uint64_t *ck = a.cand_keys + (uint64_t)q * a.cand_cap;
        const uint32_t *cv = a.cand_vals + (uint64_t)q * a.cand_cap;
        for (uint32_t i0 = 0; i0 < n; i0 += 256) {               // four key / value loads in flight per lane
            uint64_t kv[4];
            uint32_t vv[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const uint32_t idx = i0 + 64 * u + lane;
                kv[u] = idx < n ? ck[idx] : KEY_EMPTY;
                vv[u] = idx < n ? cv[idx] : 0xFFFFFFFFu;
                if (kv[u] > cut) kv[u] = KEY_EMPTY;
            }
#pragma unroll
            for (int u = 0; u < 4; ++u)
                if (i0 + 64 * u < n && __ballot(kv[u] != KEY_EMPTY) != 0ull) tk.offer(kv[u], vv[u], a.k, lane);
        }
    }
    }
    if constexpr (!PROBE) {
        const uint32_t k_out = a.k_out ? a.k_out : a.k;
        uint32_t found = 0;
        bool tie = false;
        float outd[S];
#pragma unroll
        for (int s = 0; s < S; ++s) {
            const bool have = tk.key[s] != KEY_EMPTY;
            const float d2 = __uint_as_float((uint32_t)(tk.key[s] >> 32));
            outd[s] = have ? (a.sqrt_out ? sqrt_f32_ieee(d2) : d2) : INFINITY;
        }
#pragma unroll
        for (int s = 0; s < S; ++s) {
            const uint32_t e = s * 64 + lane;
            const bool have = e < a.k && tk.key[s] != KEY_EMPTY;
            found += (uint32_t)__popcll(__ballot(have && e < k_out));
            // neighbour e+1 (next lane, or lane 0 of the next slot)
            float nd = __shfl_down(outd[s], 1, 64);
            uint64_t nkey = ((uint64_t)(uint32_t)__shfl_down((int)(uint32_t)(tk.key[s] >> 32), 1, 64) << 32) |
                            (uint32_t)__shfl_down((int)(uint32_t)tk.key[s], 1, 64);
            if (lane == 63) {
                if (s + 1 < S) {
                    nd = __builtin_bit_cast(float, readlane_u32(__builtin_bit_cast(uint32_t, outd[s + 1 < S ? s + 1 : s]), 0));
                    nkey = readlane_u64(tk.key[s + 1 < S ? s + 1 : s], 0);
                } else {
                    nkey = KEY_EMPTY;
                }
            }
            // pairs (e, e+1) with e < k_out and e+1 < k (the runner-up is entry k_out)
            if (have && e < k_out && e + 1 < a.k && nkey != KEY_EMPTY && nd == outd[s]) tie = true;
            if (e < k_out) {
                uint32_t row = 0xFFFFFFFFu;
                float d = INFINITY;
                if (have) {
                    row = a.ids ? a.ids[tk.val[s]] : tk.val[s];
                    d = outd[s];
                }
                a.row_idx[(uint64_t)q * k_out + e] = row;
                a.dist[(uint64_t)q * k_out + e] = d;
            }
        }
        if (a.n_found && lane == 0) a.n_found[q] = found;
        const bool any_tie = __ballot(tie) != 0ull;
        if (a.tie_flag && lane == 0) a.tie_flag[q] = any_tie ? 1u : 0u;
        PQV_STAMP_MAX(25);
    } else {
        probe_merge_tail<S>(a, q, lane, tk);
    }
}

// ------------------------------------------------------------------------------------
// probe_single_kernel: the whole centroid probe of ONE query in one launch.  A block takes 16 centroids (kc / 16 blocks:
// a CU takes in ~40 GB/s, so the 3 MB table wants 64+ of them); its 16 waves split the row's 4-value groups, compute the reference's per-group terms ((d0^2 + d1^2) + d2^2) + d3^2 with
// all of their loads in flight at once and leave them in LDS; wave 0 then adds the terms in the reference's order
// (index.rs:461-480: one running sum over the groups) -- the same bits as probe_rows_kernel<1>, but the 3 MB centroid
// table is read by kc / 64 blocks x 16 waves instead of kc / 256 blocks walking it 16 groups at a time (round 3: 43 -> 
// µs on C3).  The keys go to scratch, and the block that finishes LAST (a ticket counter) selects the nprobe nearest and
// runs the probe merge's tail (probe order, candidate bases, single-query bucketing, norms) with its other waves doing
// the merge's helper work: one launch instead of stream_kernel + merge_kernel.
// ------------------------------------------------------------------------------------
constexpr uint32_t PS_SLAB = 768;      // groups per LDS slab (48 KB of terms)
constexpr uint32_t PS_CPB = 16;        // centroids per block
__global__ __launch_bounds__(1024) void probe_single_kernel(const ProbeRowsArgs pr, const MergeArgs a, uint32_t *ticket, const PairQuantArgs qa) {
    __shared__ uint32_t s_last;
    __shared__ float ts[PS_SLAB * PS_CPB];
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const uint32_t cl = (uint32_t)lane & 15u, gs = (uint32_t)lane >> 4;
    const uint32_t c = blockIdx.x * PS_CPB + cl;                  // < kc_pad (a multiple of 256)
    uint64_t key = KEY_EMPTY;
    PQV_STAMP_MIN(0);
    // the int8 image of the query (wide screened path) rides along.  One-centre form: it does not depend on the probe -- an
    // extra block makes it while the others read the centroids.  Residual form (one image per probed list): the last
    // block's helper waves make them once wave 0 has the probe order (below).
    const bool extra = blockIdx.x * PS_CPB >= pr.kc_pad;
    if (extra) {
        if (wave == 0 && qa.n_pairs && !qa.probe) quantize_pair_i8_wave(qa, 0u, 0u, 0u, lane);
    } else {
        const uint32_t G = pr.dim >> 2;
        const float4 *xt = pr.cent_t + c;
        const float4 *qv = reinterpret_cast<const float4 *>(pr.queries);
        float sum = 0.0f;
        for (uint32_t s0 = 0; s0 < G; s0 += PS_SLAB) {
            const uint32_t sl = G - s0 < PS_SLAB ? G - s0 : PS_SLAB;
            // a load instruction covers 4 groups x 16 centroids (256-byte runs); thread (wave, gs): groups 4 wave + gs + 64 u
            for (uint32_t gb = (uint32_t)wave * 4u + gs; gb < sl; gb += 256) {
                float4 x[4], qq[4];
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    const uint32_t g = gb + 64u * (uint32_t)u;
                    x[u] = xt[(uint64_t)(s0 + (g < sl ? g : gb)) * pr.kc_pad];
                    qq[u] = qv[s0 + (g < sl ? g : gb)];
                }
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    const uint32_t g = gb + 64u * (uint32_t)u;
                    if (g < sl) {
                        const float d0 = qq[u].x - x[u].x, d1 = qq[u].y - x[u].y, d2 = qq[u].z - x[u].z, d3 = qq[u].w - x[u].w;
                        float t = d0 * d0 + d1 * d1;
                        t = t + d2 * d2;
                        t = t + d3 * d3;
                        ts[g * PS_CPB + cl] = t;
                    }
                }
            }
            PQV_STAMP_MAX(1);
            __syncthreads();
            if (wave == 0 && lane < (int)PS_CPB) {
                uint32_t g = 0;
                for (; g + 16 <= sl; g += 16) {
                    float t[16];
#pragma unroll
                    for (int u = 0; u < 16; ++u) t[u] = ts[(g + u) * PS_CPB + cl];
#pragma unroll
                    for (int u = 0; u < 16; ++u) sum = sum + t[u];
                }
                for (; g < sl; ++g) sum = sum + ts[g * PS_CPB + cl];
            }
            __syncthreads();
        }
        if (c < pr.kc) key = ((uint64_t)__float_as_uint(sum) << 32) | c;
    }
    PQV_STAMP_MAX(2);
    if (wave == 0 && !extra && lane < (int)PS_CPB)
    __hip_atomic_store(pr.part_keys + c, key, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    // The keys are published by agent-scope atomic stores (write-through to the memory side) and read back by agent-scope
    // atomic loads: all the ticket needs is that the stores have completed -- a release fence would also write the L2 back,
    // and the matching acquire would invalidate it under the tail's other loads (~1.5 us each way in a tail that runs alone).
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (threadIdx.x == 0) {
        const uint32_t t = atomicAdd(ticket, 1u);
        s_last = t == gridDim.x - 1u ? 1u : 0u;
        if (s_last) *ticket = 0u;                                 // ready for the next call
    }
    __syncthreads();
    if (!s_last) return;
    PQV_STAMP_MAX(3);
    __shared__ uint32_t s_probe_c[64];
    if (wave != 0) {
        if (wave == 1) probe_query_norms(a, 0u, lane);      // (beside wave 0's selection instead of after it)
        probe_merge_helpers(a, 0u);
        if (qa.n_pairs && qa.probe) {
            __syncthreads();                   // wave 0 has the probe order
            const uint32_t nw = blockDim.x / 64u - 1u;
            for (uint32_t p = (uint32_t)wave - 1u; p < qa.n_pairs; p += nw) {
                const uint32_t c = s_probe_c[p];
                if (c != 0xFFFFFFFFu) quantize_pair_i8_wave(qa, p, c, 0u, lane);
            }
        }
        return;
    }
    WaveTopk<1> tk;
    tk.init();
    // the k-th smallest of the 64 lane minima bounds the k-th smallest key: only keys at or below it are inserted.
    // (16 key loads in flight per lane: the tail runs alone on the chip, every dependent round trip is its full latency)
    uint64_t lmin = KEY_EMPTY;
    uint64_t kreg[16];
    for (uint32_t i0 = 0; i0 < pr.kc_pad; i0 += 1024) {
#pragma unroll
        for (int u = 0; u < 16; ++u) {
            const uint32_t i = i0 + 64u * (uint32_t)u + (uint32_t)lane;
            kreg[u] = i < pr.kc_pad ? __hip_atomic_load(pr.part_keys + i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : KEY_EMPTY;
        }
#pragma unroll
        for (int u = 0; u < 16; ++u) lmin = kreg[u] < lmin ? kreg[u] : lmin;
    }
    __shared__ uint64_t s_sel[128];
    const uint64_t cut = wave_kth_by_rank(lmin, a.k, lane, s_sel);       // a.k <= 64
    PQV_STAMP_MAX(4);
    uint64_t sorted = KEY_EMPTY;
    if (pr.kc_pad <= 1024 && wave_select_by_sort<16>(kreg, cut, lane, s_sel, sorted)) {
        tk.key[0] = (uint32_t)lane < a.k ? sorted : KEY_EMPTY;
        tk.val[0] = (uint32_t)tk.key[0];
    } else
    for (uint32_t i0 = 0; i0 < pr.kc_pad; i0 += 1024) {
        if (pr.kc_pad > 1024) {        // (up to 1024 centroids the keys are still in registers)
#pragma unroll
            for (int u = 0; u < 16; ++u) {
                const uint32_t i = i0 + 64u * (uint32_t)u + (uint32_t)lane;
                kreg[u] = i < pr.kc_pad ? __hip_atomic_load(pr.part_keys + i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : KEY_EMPTY;
            }
        }
#pragma unroll
        for (int u = 0; u < 16; ++u) {
            uint64_t k2 = kreg[u];
            if (k2 > cut) k2 = KEY_EMPTY;
            if (__ballot(k2 != KEY_EMPTY) != 0ull) tk.offer(k2, (uint32_t)k2, a.k, lane);
        }
    }
    PQV_STAMP_MAX(5);
    if (qa.n_pairs && qa.probe) {
        s_probe_c[lane] = ((uint32_t)lane < a.k && tk.key[0] != KEY_EMPTY) ? tk.val[0] : 0xFFFFFFFFu;
        __syncthreads();
    }
    probe_merge_tail<1>(a, 0u, lane, tk, false);
    PQV_STAMP_MAX(6);
}
hipError_t launch_probe_single(const ProbeRowsArgs &pr, const MergeArgs &a, uint32_t *ticket, const PairQuantArgs *quant, hipStream_t s) {
    if (pr.nq != 1 || a.nq != 1 || a.k == 0 || a.k > 64 || (pr.kc_pad % 256) != 0 || pr.kc_pad > 4096 || (pr.dim % 4) != 0 ||
        pr.kc == 0 || !pr.part_keys || !ticket) return hipErrorInvalidValue;
    PairQuantArgs qa{};
    if (quant) {
        qa = *quant;
        if (qa.dim % 4 || qa.nprobe == 0 || (qa.probe ? (qa.n_pairs != a.k || !qa.pair_lb) : qa.n_pairs != 1)) return hipErrorInvalidValue;
    }
    const uint32_t extra = (qa.n_pairs && !qa.probe) ? 1u : 0u;
    hipLaunchKernelGGL(probe_single_kernel, dim3(pr.kc_pad / PS_CPB + extra), dim3(1024), 0, s, pr, a, ticket, qa);
    return hipGetLastError();
}

template <bool PROBE>
static hipError_t launch_merge_t(const MergeArgs &a, hipStream_t s) {
    if (a.nq == 0) return hipSuccess;
    // (probe merge with a preset: three helper waves; final merge with deferred evaluation: the block resolves the buffer)
    //  final merge of candidate buffers at k > 64: the block selects, merge_select_large)
    dim3 grid(a.nq), block(((PROBE && (a.preset_keys || a.preset_flags)) || (!PROBE && a.cand_keys && a.k > 64 && a.k <= 256)) ? 256 : 64);
    if (!PROBE && a.cand_lb) return hipErrorInvalidValue;          // (deferred entries are resolved by launch_resolve, before the merge)
    if (a.k <= 64) hipLaunchKernelGGL((merge_kernel<1, PROBE>), grid, block, 0, s, a);
    else if (a.k <= 256) hipLaunchKernelGGL((merge_kernel<4, PROBE>), grid, block, 0, s, a);
    else if (a.k <= 1024) hipLaunchKernelGGL((merge_kernel<16, PROBE>), grid, block, 0, s, a);
    else return hipErrorInvalidValue;
    return hipGetLastError();
}
hipError_t launch_merge_final(const MergeArgs &a, hipStream_t s) { return launch_merge_t<false>(a, s); }
hipError_t launch_merge_probe(const MergeArgs &a, hipStream_t s) { return launch_merge_t<true>(a, s); }

// ------------------------------------------------------------------------------------
// pair bucketing for the batched re-rank: counting sort of (query, probe-rank) pairs by
// cluster + the group table.  Order inside a bucket is arbitrary (atomics) and does not
// matter: every pair writes its partial lists to slots fixed by (q, j, chunk, wave).
// ------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void pair_hist_kernel(const PairSortArgs a) {
    const uint32_t p = blockIdx.x * 256 + threadIdx.x;
    if (p < a.n_pairs) atomicAdd(&a.hist[(uint64_t)(a.hist_stride ? (p / a.nprobe) % HIST_REPLICAS : 0u) * a.hist_stride + a.probe[p]], 1u);
}

__global__ __launch_bounds__(1024) void pair_scan_kernel(const PairSortArgs a) {
    // single block: exclusive scans of hist and of ceil(hist / TILE_QB)
    __shared__ uint32_t s_pair[1024], s_grp[1024], s_quad[1024], s_item[1024], s_witem[1024];
    __shared__ uint32_t carry_pair, carry_grp, carry_quad, carry_item, carry_witem;
    __shared__ uint32_t s_lvl[2 * ITEM_LEVELS];        // chunk-major tables: items per level (this table, the wide one)
    const uint32_t tid = threadIdx.x;
    const bool levels = a.item_rows && a.item_chunk;
    __shared__ uint32_t s_ext[2];                      // ... items beyond the last level
    __shared__ uint32_t s_lbase[2 * ITEM_LEVELS], s_lcnt[2 * ITEM_LEVELS];     // first slot / slots of every level of the two tables (xcd_items)
    __shared__ uint32_t s_lone;                        // wide items of lists whose only quad is that wide one
    __shared__ uint32_t s_shape[2];                    // shape_stats
    if (tid == 0) { s_lone = 0; s_shape[0] = 0; s_shape[1] = 0; }
    if (levels && tid < 2 * ITEM_LEVELS) s_lvl[tid] = 0;
    if (levels && tid < 2) s_ext[tid] = 0;
    if (tid == 0) { carry_pair = 0; carry_grp = 0; carry_quad = 0; carry_item = 0; carry_witem = 0; }
    __syncthreads();
    for (uint32_t base = 0; base < a.n_clusters; base += 1024) {
        const uint32_t c = base + tid;
        uint32_t h = c < a.n_clusters ? a.hist[c] : 0;
        if (a.hist_stride && c < a.n_clusters) {          // partial copies -> total in copy 0 (pair_scatter_kernel reads it),
            uint32_t hv[HIST_REPLICAS];                   // and each copy's first index in the cluster's bucket
#pragma unroll
            for (uint32_t r = 1; r < HIST_REPLICAS; ++r) hv[r] = a.hist[(uint64_t)r * a.hist_stride + c];   // loads in flight together
            a.cursor[c] = 0u;
#pragma unroll
            for (uint32_t r = 1; r < HIST_REPLICAS; ++r) {
                a.cursor[(uint64_t)r * a.hist_stride + c] = h;
                h += hv[r];
            }
            a.hist[c] = h;
        }
        if (a.shape_stats && c < a.n_clusters && h > a.shape_narrow) {
            const uint64_t len = a.list_off[c + 1] - a.list_off[c];
            const uint32_t l32 = len < 0x00FFFFFFull ? (uint32_t)len : 0x00FFFFFFu;
            atomicAdd(&s_shape[1], l32);
            if (h > a.shape_wide) atomicAdd(&s_shape[0], l32);
        }
        const uint32_t g = (h + TILE_QB - 1) / TILE_QB;
        const uint32_t qd = (h + a.quad_width - 1) / a.quad_width;
        uint32_t ni = 0, nwi = 0;            // work items of the cluster: quads x row chunks of its list
        if (a.item_rows && c < a.n_clusters) {
            const uint64_t len = a.list_off[c + 1] - a.list_off[c];
            uint32_t nwq = 0;                // wide quads: the full ones + a remainder of >= wide_min pairs
            if (a.wide_min) {
                nwq = h / a.quad_width + ((h % a.quad_width) >= a.wide_min ? 1u : 0u);
                nwi = nwq * (uint32_t)((len + a.wide_item_rows - 1) / a.wide_item_rows);
            }
            ni = (qd - nwq) * (uint32_t)((len + a.item_rows - 1) / a.item_rows);
        }
        s_pair[tid] = h; s_grp[tid] = g; s_quad[tid] = qd; s_item[tid] = ni; s_witem[tid] = nwi;
        __syncthreads();
        for (uint32_t off = 1; off < 1024; off <<= 1) {
            uint32_t vp = 0, vg = 0, vq = 0, vi = 0, vw = 0;
            if (tid >= off) { vp = s_pair[tid - off]; vg = s_grp[tid - off]; vq = s_quad[tid - off]; vi = s_item[tid - off]; vw = s_witem[tid - off]; }
            __syncthreads();
            s_pair[tid] += vp; s_grp[tid] += vg; s_quad[tid] += vq; s_item[tid] += vi; s_witem[tid] += vw;
            __syncthreads();
        }
        if (c < a.n_clusters) {
            a.pair_off[c] = carry_pair + s_pair[tid] - h;
            a.group_off[c] = carry_grp + s_grp[tid] - g;
            a.quad_off[c] = carry_quad + s_quad[tid] - qd;
            if (a.item_rows) a.item_off[c] = carry_item + s_item[tid] - ni;
            if (a.item_rows && a.wide_min) a.wide_item_off[c] = carry_witem + s_witem[tid] - nwi;
        }
        __syncthreads();
        if (tid == 1023) {
            carry_pair += s_pair[1023]; carry_grp += s_grp[1023]; carry_quad += s_quad[1023]; carry_item += s_item[1023];
            carry_witem += s_witem[1023];
        }
        __syncthreads();
    }
    if (tid == 0) {
        if (a.shape_stats) { a.shape_stats[0] = s_shape[0]; a.shape_stats[1] = s_shape[1]; }
        a.pair_off[a.n_clusters] = carry_pair;
        a.group_off[a.n_clusters] = carry_grp;
        a.quad_off[a.n_clusters] = carry_quad;
        *a.n_groups = carry_grp;
        *a.n_quads = carry_quad;
        if (a.item_rows) { a.item_off[a.n_clusters] = carry_item; *a.n_items = carry_item < a.max_items ? carry_item : a.max_items; }
        if (a.item_rows && a.wide_min) {
            a.wide_item_off[a.n_clusters] = carry_witem;
            *a.wide_n_items = carry_witem < a.wide_max_items ? carry_witem : a.wide_max_items;
        }
    }
    // Chunk-major item tables (both tables here, pair_scatter_kernel then leaves them alone): count the items of every level,
    // scan the levels, and hand out the slots of a level wave by wave -- a wave prefix sum over its 64 clusters and ONE LDS
    // atomic per wave and level (one global atomic per item measured + 27 us on the 5.5 k items of a C3 step).
    if (!levels) return;
    __syncthreads();
    const int lane = tid & 63;
    constexpr uint32_t LL = ITEM_LEVELS - 1;
    uint32_t mn = 0, mw = 0, nch = 0, wnch = 0, q0 = 0;      // normal / wide quads of the cluster, their row chunks, first quad
    auto cluster_shape = [&](uint32_t c) {
        mn = mw = nch = wnch = q0 = 0;
        if (c < a.n_clusters) {
            const uint32_t h = a.hist[c];
            const uint32_t qd = (h + a.quad_width - 1) / a.quad_width;
            const uint64_t len = a.list_off[c + 1] - a.list_off[c];
            if (a.wide_min) mw = h / a.quad_width + ((h % a.quad_width) >= a.wide_min ? 1u : 0u);
            mn = qd - mw;
            nch = mn ? (uint32_t)((len + a.item_rows - 1) / a.item_rows) : 0u;
            wnch = mw ? (uint32_t)((len + a.wide_item_rows - 1) / a.wide_item_rows) : 0u;
            q0 = a.quad_off[c];
        }
    };
    const bool one_round = a.n_clusters <= 1024;       // every thread keeps its cluster's shape in registers for both passes
    if (one_round) cluster_shape(tid);
    for (int pass = 0; pass < 2; ++pass) {
        for (uint32_t base = 0; base < a.n_clusters; base += 1024) {
            if (!one_round) cluster_shape(base + tid);
#pragma unroll
            for (int tb = 0; tb < 2; ++tb) {          // table 0: the quads of <= quad_width pairs; table 1: the wide quads (the cluster's first mw quads)
                if (tb == 1 && a.wide_list_major) continue;      // (list-major wide table: pair_scatter_kernel writes it)
                const uint32_t m = tb ? mw : mn, n = tb ? wnch : nch, qf = tb ? q0 : q0 + mw;
                if (pass == 0) {
                    // counts per level as a difference array: + m at level 0, - m behind the cluster's last level; what lies
                    // beyond the table's levels is added to the last one
                    if (m && n) {
                        if (tb == 1 && mw == 1 && mn == 0) atomicAdd(&s_lone, n);
                        atomicAdd(&s_lvl[tb * ITEM_LEVELS], m);
                        atomicAdd(&s_lvl[tb * ITEM_LEVELS + (n < LL ? n : LL)], 0u - m);
                        if (n > LL) atomicAdd(&s_ext[tb], m * (n - LL));
                    }
                    continue;
                }
                uint32_t nmax = n;
#pragma unroll
                for (int off = 32; off > 0; off >>= 1) { const uint32_t o = (uint32_t)__shfl_xor((int)nmax, off, 64); nmax = o > nmax ? o : nmax; }
                for (uint32_t t = 0; t < nmax; ++t) {
                    const uint32_t lv = t < LL ? t : LL;
                    const uint32_t mine = t < n ? m : 0u;
                    const uint32_t incl = wave_incl_scan_u32(mine);
                    const uint32_t total = readlane_u32(incl, 63);
                    if (total == 0) continue;
                    {
                        uint32_t first = 0;
                        if (lane == 0) first = atomicAdd(&s_lvl[tb * ITEM_LEVELS + lv], total);
                        first = readlane_u32(first, 0) + incl - mine;
                        uint32_t *iq = tb ? a.wide_item_quad : a.item_quad, *ic = tb ? a.wide_item_chunk : a.item_chunk;
                        const uint32_t lim = tb ? a.wide_max_items : a.max_items;
                        if (a.xcd_items & (1u << tb)) {
                            // XCD-aware slots: the quads of ONE list at the same chunk level stream the same rows, so they go to
                            // the same XCD back to back (workgroup i of a 1-D grid runs on XCD i % 8): the level's slots are
                            // filled column by column of an 8-column layout -- logical neighbours are 8 slots apart -- and the
                            // second reader of a row finds it in that XCD's L2 instead of fetching it through the fabric again
                            // (clustered queries: hundreds of pairs per popular list, 3.3 x the distinct rows fetched before)
                            const uint32_t lb = s_lbase[tb * ITEM_LEVELS + lv], n_l = s_lcnt[tb * ITEM_LEVELS + lv];
                            const uint32_t rf = n_l >> 3, rem = n_l & 7u;
                            for (uint32_t k = 0; k < mine; ++k) {
                                const uint32_t j = first + k - lb;
                                uint32_t x, r;
                                if (j < rem * (rf + 1u)) { x = j / (rf + 1u); r = j % (rf + 1u); }
                                else { const uint32_t j2 = j - rem * (rf + 1u); x = rem + j2 / (rf ? rf : 1u); r = j2 % (rf ? rf : 1u); }
                                const uint32_t at = lb + r * 8u + x;
                                if (at < lim && j < n_l) { iq[at] = qf + k; ic[at] = t; }
                            }
                        } else {
                            for (uint32_t k = 0; k < mine; ++k)
                                if (first + k < lim) { iq[first + k] = qf + k; ic[first + k] = t; }
                        }
                    }
                }
            }
        }
        __syncthreads();
        if (pass == 0) {               // differences -> counts -> first slot of every level (the running cursors of pass 1)
            if (tid < 2) {
                uint32_t b = 0, cnt = 0;
                for (uint32_t t = 0; t < ITEM_LEVELS; ++t) {
                    cnt += s_lvl[tid * ITEM_LEVELS + t];
                    s_lvl[tid * ITEM_LEVELS + t] = b;
                    const uint32_t n_l = cnt + (t == LL ? s_ext[tid] : 0u);
                    s_lbase[tid * ITEM_LEVELS + t] = b; s_lcnt[tid * ITEM_LEVELS + t] = n_l;
                    b += n_l;
                }
                if (tid == 1 && a.wide_stats) { a.wide_stats[0] = b; a.wide_stats[1] = s_lone; }
            }
            __syncthreads();
        }
    }
}

__global__ __launch_bounds__(256) void pair_scatter_kernel(const PairSortArgs a) {
    const uint32_t p = blockIdx.x * 256 + threadIdx.x;
    if (p >= a.n_pairs) return;
    const uint32_t c = a.probe[p];
    const uint32_t i = atomicAdd(&a.cursor[(uint64_t)(a.hist_stride ? (p / a.nprobe) % HIST_REPLICAS : 0u) * a.hist_stride + c], 1u);
    const uint32_t slot = a.pair_off[c] + i;
    a.pairs[slot] = p;
    if (i % TILE_QB == 0) {
        const uint32_t h = a.hist[c];
        const uint32_t cnt = (h - i < (uint32_t)TILE_QB) ? (h - i) : (uint32_t)TILE_QB;
        a.groups[a.group_off[c] + i / TILE_QB] = make_uint4(c, slot, cnt, 0u);
    }
    if (i % a.quad_width == 0) {
        const uint32_t h = a.hist[c];
        const uint32_t qi = a.quad_off[c] + i / a.quad_width;
        uint32_t first = 0;
        const uint32_t qcnt = h - i < a.quad_width ? h - i : a.quad_width;
        if (a.item_rows) {
            const uint64_t len = a.list_off[c + 1] - a.list_off[c];
            const bool wq = a.wide_min && qcnt >= a.wide_min;      // a wide quad: the list's quads before it are wide too
            if (a.item_chunk && !(wq && a.wide_list_major)) {                // chunk-major: pair_scan_kernel wrote the table(s)
            } else if (wq) {
                const uint32_t nch = (uint32_t)((len + a.wide_item_rows - 1) / a.wide_item_rows);
                first = a.wide_item_off[c] + (i / a.quad_width) * nch;
                for (uint32_t t = 0; t < nch && first + t < a.wide_max_items; ++t) a.wide_item_quad[first + t] = qi;
            } else {
                const uint32_t nch = (uint32_t)((len + a.item_rows - 1) / a.item_rows);
                // (with wide quads about, a quad of < wide_min pairs is the list's last and its only one in this table)
                first = a.item_off[c] + (a.wide_min ? 0u : (i / a.quad_width) * nch);
                for (uint32_t t = 0; t < nch && first + t < a.max_items; ++t) a.item_quad[first + t] = qi;
            }
        }
        a.quads[qi] = make_uint4(c, slot, qcnt, first);
    }
}

hipError_t launch_pair_sort(const PairSortArgs &a, hipStream_t s) {
    if (a.n_pairs == 0) return hipSuccess;
    const uint32_t blocks = (a.n_pairs + 255) / 256;
    if (!a.hist_done) hipLaunchKernelGGL(pair_hist_kernel, dim3(blocks), dim3(256), 0, s, a);
    hipLaunchKernelGGL(pair_scan_kernel, dim3(1), dim3(1024), 0, s, a);
    hipLaunchKernelGGL(pair_scatter_kernel, dim3(blocks), dim3(256), 0, s, a);
    return hipGetLastError();
}



// an empty kernel the library launches at the first call for a device (and on a new stream): the runtime loads this unit's code
// object and sets up the stream's hardware queue then, not inside the first build or the first query
__global__ void touch_probe_kernel() {}
hipError_t touch_probe(hipStream_t s) {
    hipLaunchKernelGGL(touch_probe_kernel, dim3(1), dim3(64), 0, s);
    return hipGetLastError();
}

}  // namespace pqv
