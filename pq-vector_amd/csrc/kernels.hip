// kernels.hip -- hand-written gfx950 (CDNA4, wave64) kernels of the pq-vector hot path.
//
// Everything here computes the reference's squared-L2 in the reference's exact f32
// summation order (no FMA contraction: built with -ffp-contract=off), so distances are
// bit-identical to the CPU path and every argmin / top-k decision is too.
//
//   stream_kernel   the candidate re-rank (src/ivf/search.rs:112-127, exec.rs:457-484),
//                   the centroid probe (src/ivf/index.rs:130-149) and the k-means++
//                   min-distance rounds (index.rs:344-369): HBM-streaming, coalesced 16 B
//                   per lane, per-row serial chains re-created through an LDS transpose.
//   merge_kernel    folds the per-wave top-k lists of one query (heap semantics of
//                   search.rs:119-126 == k smallest by (d2, candidate position)).
//   assign_kernel   Lloyd assign + final assignment (index.rs:395-424, :189-201,:244-257):
//                   VALU-bound, lane-per-row with the centroid tile in SGPRs.
//   lloyd_update    index.rs:436-453 with the reference's ascending-row f32 add order.
//   gather_rows     sample_embeddings (index.rs:234-239) and the IVF-order re-layout.
#include "kernels.h"

#include <hip/hip_runtime.h>
#include <atomic>
#include <type_traits>
#include <utility>
#include <math.h>
#include <stdlib.h>

namespace pqv {

// diagnostic build (make stamps): wall-clock stamps (100 MHz) of the one-query launch sequence, read by tools/stamps_single.py
#ifdef PQV_STAMPS
__device__ unsigned long long g_stamps[64];
#define PQV_STAMP_MAX(i) do { if (threadIdx.x == 0) atomicMax(&g_stamps[i], (unsigned long long)wall_clock64()); } while (0)
#define PQV_STAMP_MIN(i) do { if (threadIdx.x == 0) atomicMin(&g_stamps[i], (unsigned long long)wall_clock64()); } while (0)
hipError_t stamps_io(unsigned long long *out, int reset) {
    if (out) { hipError_t e = hipMemcpyFromSymbol(out, HIP_SYMBOL(g_stamps), sizeof(unsigned long long) * 64); if (e != hipSuccess) return e; }
    if (reset) {
        unsigned long long init[64];
        for (int k = 0; k < 64; ++k) init[k] = (k % 8 == 0) ? ~0ull : 0ull;       // slots 0, 8, 16, ...: earliest start
        return hipMemcpyToSymbol(HIP_SYMBOL(g_stamps), init, sizeof init);
    }
    return hipSuccess;
}
#else
#define PQV_STAMP_MAX(i) do { } while (0)
#define PQV_STAMP_MIN(i) do { } while (0)
#endif

// ------------------------------------------------------------------------------------
// small wave64 helpers
// ------------------------------------------------------------------------------------
__device__ __forceinline__ uint32_t readlane_u32(uint32_t v, int l) {
    return (uint32_t)__builtin_amdgcn_readlane((int)v, l);
}
// v[lane L] = wave-uniform value s (v_writelane_b32 with an immediate lane; this hipcc has no builtin)
template <int L>
__device__ __forceinline__ void writelane_imm(uint32_t &v, uint32_t s) {
    asm("v_writelane_b32 %0, %1, %2" : "+v"(v) : "s"(s), "n"(L));
}
// inclusive prefix sum over the 64 lanes on the DPP network (no LDS): row_shr 1/2/4/8 inside each
// 16-lane row, then row_bcast:15 / row_bcast:31 carry the row totals across rows
__device__ __forceinline__ uint32_t wave_incl_scan_u32(uint32_t v) {
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x111, 0xF, 0xF, false);
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x112, 0xF, 0xF, false);
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x114, 0xF, 0xF, false);
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x118, 0xF, 0xF, false);
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x142, 0xA, 0xF, false);
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x143, 0xC, 0xF, false);
    return v;
}
// 16-byte operand load through a buffer resource: scalar base (the descriptor) + scalar byte offset + a
// per-lane byte offset that never changes -- no vector ALU address arithmetic per load.  On gfx950 VALU
// work does not overlap the f32 MFMAs, not even across waves (tools/mfma_mix_ubench.hip), so every
// address instruction in the K loop costs matrix throughput.
typedef float f32x4_raw __attribute__((ext_vector_type(4)));
__device__ __forceinline__ __amdgpu_buffer_rsrc_t operand_rsrc(const void *base) {
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<void *>(base), 0, 0x7FFFFFFF, 0x00020000);
}
template <int AUX = 0>      // cache policy bits of the load (2 = nt: a stream that is read once)
__device__ __forceinline__ float4 buf_ld16(__amdgpu_buffer_rsrc_t r, uint32_t lane_bytes, uint32_t uniform_bytes) {
    const f32x4_raw v = __builtin_bit_cast(f32x4_raw, __builtin_amdgcn_raw_buffer_load_b128(r, (int)lane_bytes, (int)uniform_bytes, AUX));
    return make_float4(v.x, v.y, v.z, v.w);
}
#ifndef PQV_ROW_AUX
#define PQV_ROW_AUX 0
#endif
#ifndef PQV_STAGE_UNROLL
#define PQV_STAGE_UNROLL 12
#endif
// build-time knobs of wide_filter_kernel (tools/variant.sh builds A/B libraries with other values; the defaults are the measured ones)
#ifndef PQV_NS_WIDE
#define PQV_NS_WIDE 2          // operand stages in flight, 8-wave blocks of <= 96 queries
#endif
#ifndef PQV_NS_TS2
#define PQV_NS_TS2 2           // ... of the wide-quad instance (32-row tiles): 3 / 4 / 6 measured no faster, 4 and 6 spill
#endif
#ifndef PQV_APD_TS2
#define PQV_APD_TS2 2          // A operands read this many groups ahead in the wide-quad instance (0: 1066, 2: 966, 4: 1004 us on C3)
#endif
#ifndef PQV_APD
#define PQV_APD 0              // ... in the 64-row-tile instances (1 and 2 measured no faster: four MFMAs hide the read)
#endif
#ifndef PQV_XTA_TS2
#define PQV_XTA_TS2 1          // thresholds / row terms requested a tile early in the wide-quad instance (1050 -> 966 us on C3)
#endif
#ifndef PQV_XTA
#define PQV_XTA 0              // ... in the 64-row-tile instances (measured 3 % slower: 8 more spills)
#endif
#ifndef PQV_EVAL_NB_TS2
#define PQV_EVAL_NB_TS2 16     // row chunks in flight per lane in the wide-quad instance's exact evaluations
#endif
typedef float f32x4_acc __attribute__((ext_vector_type(4)));
// One K step of the score contraction for a 16 x 16 tile.  f32 operands: 16 dims, four 16x16x4 MFMAs;
// f16 operands (8 halves per lane, see launch_block_rows_f16): 32 dims, one 16x16x32 MFMA.
typedef _Float16 f16x8_t __attribute__((ext_vector_type(8)));
// Operand forms of the screen contraction: f32 (exact products), f16 images, int8 images (see wide_filter_kernel)
enum ScreenOp : int { OP_F32 = 0, OP_F16 = 1, OP_I8 = 2 };
typedef int i32x4_acc __attribute__((ext_vector_type(4)));
template <int OP>
__device__ __forceinline__ void mfma_step(f32x4_acc &acc, const float4 q, const float4 x) {
    if constexpr (OP == OP_F16) {
        acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8_t, q), __builtin_bit_cast(f16x8_t, x), acc, 0, 0, 0);
    } else {
        acc = __builtin_amdgcn_mfma_f32_16x16x4f32(q.x, x.x, acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_16x16x4f32(q.y, x.y, acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_16x16x4f32(q.z, x.z, acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_16x16x4f32(q.w, x.w, acc, 0, 0, 0);
    }
}
// int8 operands: 16 bytes per lane = 64 dims per step, exact int32 accumulation (v_mfma_i32_16x16x64_i8)
template <int OP>
__device__ __forceinline__ void mfma_step(i32x4_acc &acc, const float4 q, const float4 x) {
    static_assert(OP == OP_I8, "integer accumulators belong to the int8 form");
    acc = __builtin_amdgcn_mfma_i32_16x16x64_i8(__builtin_bit_cast(i32x4_acc, q), __builtin_bit_cast(i32x4_acc, x), acc, 0, 0, 0);
}
// eight f32 values scaled by a power of two and rounded to f16 (round to nearest even), packed as 16 bytes
__device__ __forceinline__ float4 pack_f16x8(const float4 lo, const float4 hi, float scale) {
    f16x8_t h;
    // (callers clamp where the image must stay finite: see pack_f16x8_clamped)
    h[0] = (_Float16)(lo.x * scale); h[1] = (_Float16)(lo.y * scale); h[2] = (_Float16)(lo.z * scale); h[3] = (_Float16)(lo.w * scale);
    h[4] = (_Float16)(hi.x * scale); h[5] = (_Float16)(hi.y * scale); h[6] = (_Float16)(hi.z * scale); h[7] = (_Float16)(hi.w * scale);
    return __builtin_bit_cast(float4, h);
}
// the same with the scaled values clamped to the finite f16 range (NaN -> -65504): the screen of the
// f16 kernels reads sign bits and must never see a NaN score
__device__ __forceinline__ float4 pack_f16x8_clamped(const float4 lo, const float4 hi, float scale) {
    auto c = [&](float v) { return (_Float16)fminf(fmaxf(v * scale, -65504.0f), 65504.0f); };
    f16x8_t h;
    h[0] = c(lo.x); h[1] = c(lo.y); h[2] = c(lo.z); h[3] = c(lo.w);
    h[4] = c(hi.x); h[5] = c(hi.y); h[6] = c(hi.z); h[7] = c(hi.w);
    return __builtin_bit_cast(float4, h);
}
// compile-time loop: f(std::integral_constant<int, 0>{}) ... f(std::integral_constant<int, N - 1>{})
template <class F, int... I>
__device__ __forceinline__ void static_for_impl(F &&f, std::integer_sequence<int, I...>) {
    (f(std::integral_constant<int, I>{}), ...);
}
template <int N, class F>
__device__ __forceinline__ void static_for(F &&f) {
    static_for_impl(f, std::make_integer_sequence<int, N>{});
}
__device__ __forceinline__ uint64_t readlane_u64(uint64_t v, int l) {
    uint32_t lo = readlane_u32((uint32_t)v, l), hi = readlane_u32((uint32_t)(v >> 32), l);
    return ((uint64_t)hi << 32) | lo;
}
__device__ __forceinline__ uint64_t shfl_up1_u64(uint64_t v) {
    uint32_t lo = (uint32_t)__shfl_up((int)(uint32_t)v, 1, 64);
    uint32_t hi = (uint32_t)__shfl_up((int)(uint32_t)(v >> 32), 1, 64);
    return ((uint64_t)hi << 32) | lo;
}
// Compiler-level ordering point between a wave's LDS writes and its own cross-lane LDS
// reads.  LDS executes one wave's DS instructions in issue order, so no s_barrier is
// needed for data that never leaves the wave.
__device__ __forceinline__ void wave_lds_fence() {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

// IEEE-754 correctly rounded f32 sqrt and divide (search.rs:133, index.rs:450), taken
// through f64: 53 >= 2*24+2 bits makes the second rounding innocuous, and it does not
// depend on how the compiler lowers f32 sqrt/div (v_sqrt_f32 / v_rcp_f32 are ~1 ulp).
__device__ __forceinline__ float sqrt_f32_ieee(float x) { return (float)sqrt((double)x); }
__device__ __forceinline__ float div_f32_ieee(float a, float b) { return (float)((double)a / (double)b); }

// ------------------------------------------------------------------------------------
// Wave-distributed sorted top-k list: element e lives in slot e/64, lane e%64; ascending.
// Keys are (f32 bits of d2 << 32) | candidate position: unique, and ordered exactly like
// the reference's heap admission rule (strict '<' keeps the earlier candidate on ties).
// ------------------------------------------------------------------------------------
template <int S>
struct WaveTopk {
    uint64_t key[S];
    uint32_t val[S];

    __device__ __forceinline__ void init() {
#pragma unroll
        for (int s = 0; s < S; ++s) { key[s] = KEY_EMPTY; val[s] = 0xFFFFFFFFu; }
    }
    // key of element k-1 (the admission threshold); k is wave-uniform
    __device__ __forceinline__ uint64_t kth(uint32_t k) const {
        const uint32_t e = k - 1;
        uint64_t r = KEY_EMPTY;
#pragma unroll
        for (int s = 0; s < S; ++s)
            if ((int)(e >> 6) == s) r = readlane_u64(key[s], (int)(e & 63));
        return r;
    }
    // insert (x, xv), wave-uniform, dropping the largest element
    __device__ __forceinline__ void insert(uint64_t x, uint32_t xv, int lane) {
        int p = 0;
#pragma unroll
        for (int s = 0; s < S; ++s) p += __popcll(__ballot(key[s] < x));
#pragma unroll
        for (int s = S - 1; s >= 0; --s) {
            uint64_t up = shfl_up1_u64(key[s]);
            uint32_t upv = (uint32_t)__shfl_up((int)val[s], 1, 64);
            if (s > 0) {
                const uint64_t pk = readlane_u64(key[s - 1], 63);
                const uint32_t pv = readlane_u32(val[s - 1], 63);
                if (lane == 0) { up = pk; upv = pv; }
            }
            const int e = s * 64 + lane;
            if (e > p) { key[s] = up; val[s] = upv; }
            else if (e == p) { key[s] = x; val[s] = xv; }
        }
    }
    // offer one candidate per lane (mykey == KEY_EMPTY for lanes with none)
    __device__ __forceinline__ void offer(uint64_t mykey, uint32_t myval, uint32_t k, int lane) {
        uint64_t thr = kth(k);
        unsigned long long m = __ballot(mykey < thr);
        while (m) {
            const int L = __builtin_ctzll(m);
            const uint64_t x = readlane_u64(mykey, L);
            const uint32_t xv = readlane_u32(myval, L);
            insert(x, xv, lane);
            thr = kth(k);
            m &= m - 1;
            m &= __ballot(mykey < thr);
        }
    }
};

// ------------------------------------------------------------------------------------
// 16-byte row loads.  ALIGNED: dim % 4 == 0 so every row starts 16-B aligned.
// ------------------------------------------------------------------------------------
template <bool ALIGNED>
__device__ __forceinline__ float4 load4(const float *p) {
    if constexpr (ALIGNED) {
        return *reinterpret_cast<const float4 *>(p);
    } else {
        float4 v;
        v.x = p[0]; v.y = p[1]; v.z = p[2]; v.w = p[3];
        return v;
    }
}

// Wave-uniform 16-byte operand (a query / centroid chunk shared by all 64 lanes): read
// through the constant address space so the backend always selects scalar loads
// (s_load_dwordx4..x16 into SGPRs, consumed as free scalar VALU operands).  Without this the
// compiler falls back to per-lane vector loads as soon as the kernel also contains atomics
// or stores it cannot prove disjoint from the operand matrix.
typedef const __attribute__((address_space(4))) float cfloat_as4;
template <bool ALIGNED>
__device__ __forceinline__ float4 load4_uniform(const float *p) {
    cfloat_as4 *c = (cfloat_as4 *)(uintptr_t)p;
    if constexpr (ALIGNED) {
        typedef float f32x4_t __attribute__((ext_vector_type(4)));
        typedef const __attribute__((address_space(4))) f32x4_t cf32x4_as4;
        const f32x4_t v = *(cf32x4_as4 *)c;
        return make_float4(v.x, v.y, v.z, v.w);
    } else {
        float4 v;
        v.x = c[0]; v.y = c[1]; v.z = c[2]; v.w = c[3];
        return v;
    }
}
__device__ __forceinline__ float load1_uniform(const float *p) {
    return *(cfloat_as4 *)(uintptr_t)p;
}

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

__device__ __forceinline__ int quant_i8(float t, float scale) {
    const float v = rintf(t * scale);
    return (int)fminf(fmaxf(v, -127.0f), 127.0f);      // NaN -> -127 (callers flag non-finite inputs separately)
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

__device__ __forceinline__ void bitonic_sort64(uint64_t &key, uint32_t &val, int lane);     // (defined below)
// The k smallest of a wave's NK keys per lane, given a cut that at least k of them do not exceed (the k-th smallest of
// the lane minima): when at most 64 keys pass the cut -- the usual case, k .. a few dozen -- they are compacted through
// `buf` (64 entries of this wave's LDS) and ONE bitonic sort replaces their serial insertion (0.1 us each in a tail that
// runs alone on the chip).  Returns false, leaving `sorted` alone, when more than 64 pass (the caller inserts them).
// k-th smallest (k >= 1) of the wave's 64 keys by rank counting: every lane compares its key with all 64 (broadcast LDS
// reads, ~0.2 us) -- a third of a bitonic sort.  Keys other than KEY_EMPTY are distinct.  `buf`: 64 entries.
__device__ __forceinline__ uint64_t wave_kth_by_rank(uint64_t key, uint32_t k, int lane, uint64_t *buf) {
    buf[lane] = key;
    wave_lds_fence();
    uint32_t rank = 0;
#pragma unroll 16
    for (int j = 0; j < 64; ++j) rank += buf[j] < key ? 1u : 0u;
    const unsigned long long m = __ballot(key != KEY_EMPTY && rank == k - 1u);
    wave_lds_fence();
    return m ? readlane_u64(key, __builtin_ctzll(m)) : KEY_EMPTY;
}
// ascending order of the wave's distinct keys (KEY_EMPTY = none; only lanes < span hold keys) by rank counting
__device__ __forceinline__ uint64_t wave_sort_by_rank(uint64_t key, uint32_t span, int lane, uint64_t *buf /* 128 entries */) {
    buf[lane] = key;
    wave_lds_fence();
    uint32_t rank = 0;
    for (uint32_t j = 0; j < span; ++j) rank += buf[j] < key ? 1u : 0u;
    const bool have = key != KEY_EMPTY;
    const uint32_t total = (uint32_t)__popcll(__ballot(have));
    if (have) buf[64 + rank] = key;
    wave_lds_fence();
    const uint64_t r = (uint32_t)lane < total ? buf[64 + lane] : KEY_EMPTY;
    wave_lds_fence();
    return r;
}
template <int NK>
__device__ __forceinline__ bool wave_select_by_sort(const uint64_t (&keys)[NK], uint64_t cut, int lane, uint64_t *buf /* 128 entries */, uint64_t &sorted) {
    uint32_t mine = 0;
#pragma unroll
    for (int u = 0; u < NK; ++u) mine += (keys[u] != KEY_EMPTY && keys[u] <= cut) ? 1u : 0u;
    uint32_t incl = mine;
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
        const uint32_t o = (uint32_t)__shfl_up((int)incl, off, 64);
        if (lane >= off) incl += o;
    }
    const uint32_t total = readlane_u32(incl, 63);
    if (total > 64u) return false;
    uint32_t at = incl - mine;
#pragma unroll
    for (int u = 0; u < NK; ++u)
        if (keys[u] != KEY_EMPTY && keys[u] <= cut) buf[64 + at++] = keys[u];
    wave_lds_fence();
    const uint64_t mykey = (uint32_t)lane < total ? buf[64 + lane] : KEY_EMPTY;
    wave_lds_fence();
    sorted = wave_sort_by_rank(mykey, total, lane, buf);
    return true;
}
// the same with a 32-bit payload per key
template <int NK>
__device__ __forceinline__ bool wave_select_by_sort_kv(const uint64_t (&keys)[NK], const uint32_t (&vals)[NK], uint64_t cut, int lane,
                                                       uint64_t *buf /* 128 */, uint32_t *vbuf /* 128 */, uint64_t &sorted, uint32_t &sval) {
    uint32_t mine = 0;
#pragma unroll
    for (int u = 0; u < NK; ++u) mine += (keys[u] != KEY_EMPTY && keys[u] <= cut) ? 1u : 0u;
    uint32_t incl = mine;
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
        const uint32_t o = (uint32_t)__shfl_up((int)incl, off, 64);
        if (lane >= off) incl += o;
    }
    const uint32_t total = readlane_u32(incl, 63);
    if (total > 64u) return false;
    uint32_t at = incl - mine;
#pragma unroll
    for (int u = 0; u < NK; ++u)
        if (keys[u] != KEY_EMPTY && keys[u] <= cut) { buf[at] = keys[u]; vbuf[at] = vals[u]; ++at; }
    wave_lds_fence();
    const bool have = (uint32_t)lane < total;
    const uint64_t key = have ? buf[lane] : KEY_EMPTY;
    const uint32_t val = have ? vbuf[lane] : 0xFFFFFFFFu;
    uint32_t rank = 0;
    for (uint32_t j = 0; j < total; ++j) rank += buf[j] < key ? 1u : 0u;
    if (have) { buf[64 + rank] = key; vbuf[64 + rank] = val; }
    wave_lds_fence();
    sorted = have ? buf[64 + lane] : KEY_EMPTY;
    sval = have ? vbuf[64 + lane] : 0xFFFFFFFFu;
    wave_lds_fence();
    return true;
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

template <int S, bool PROBE>
__global__ __launch_bounds__(256) void merge_kernel(const MergeArgs a) {
    const int lane = threadIdx.x & 63;
    const uint32_t q = blockIdx.x;
    if constexpr (!PROBE) PQV_STAMP_MIN(24);
    if (threadIdx.x >= 64) {
        // helper waves (probe mode with a preset only): the query's partial lists of the re-rank start EMPTY
        if constexpr (PROBE) probe_merge_helpers(a, q);
        return;
    }
    WaveTopk<S> tk;
    tk.init();
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
    bool folded = false;
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
    dim3 grid(a.nq), block(PROBE && (a.preset_keys || a.preset_flags) ? 256 : 64);       // probe merge with a preset: three helper waves
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
    __shared__ uint32_t s_lone;                        // wide items of lists whose only quad is that wide one
    if (tid == 0) s_lone = 0;
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
                        for (uint32_t k = 0; k < mine; ++k)
                            if (first + k < lim) { iq[first + k] = qf + k; ic[first + k] = t; }
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
                    b += cnt + (t == LL ? s_ext[tid] : 0u);
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
            if (a.item_chunk) {                // chunk-major: pair_scan_kernel wrote both tables
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

// ------------------------------------------------------------------------------------
// tile_rerank_kernel: the batched candidate re-rank.
//
// grid = (blocks_per_list, max_groups); block = 4 independent waves.  A block takes one
// group (<= QB queries that all probe cluster c) and one row chunk of c's inverted list;
// each wave walks its rows lane-per-row in 64-row tiles.  Per tile the lane's row is
// loaded 128 B at a time (a full cache line per lane) and every query of the group is
// applied to it from SGPRs (wave-uniform scalar loads) -- each streamed row is used QB
// times, with zero LDS traffic.  Every (row, query) chain is the reference's serial
//   sum += ((d0^2 + d1^2) + d2^2) + d3^2   in ascending group order (index.rs:461-480).
// Top-k: one wave-distributed sorted list per query of the group (registers); the first
// tile seeds it with a 64-lane bitonic sort, later tiles insert past the k-th key.
// ------------------------------------------------------------------------------------
// XCD-aware workgroup remap (bijective form).  Hardware places workgroup L on XCD L % 8; giving
// XCD i the i-th CONTIGUOUS slice of the (chunk-fastest) block space puts the blocks of
// consecutive query groups -- the groups of one cluster -- on one XCD, so a row chunk fetched
// for one group is an L2 hit for the next.  Placement only affects speed, never results.
// Quad-to-XCD affinity for the wide kernels: workgroup L runs on XCD L % 8; give XCD i the quads
// i, i + 8, i + 16, ... with all their row chunks, so the blocks that share a quad's operands (its
// blocked queries, its cluster's rows) also share an L2.  gridDim.y must be a multiple of 8.
// Mode 2 (n_quads known): XCD i takes the CONTIGUOUS quad range [i * per, (i + 1) * per), per =
// ceil(n_quads / 8).  The quads of one cluster are adjacent, so they run on one XCD at about the same
// time and the second one finds the cluster's rows in that XCD's L2 instead of fetching them again.
__device__ __forceinline__ void quad_xcd_remap(uint32_t &bx, uint32_t &by, int enable, uint32_t n_quads = 0) {
    if (!enable) { bx = blockIdx.x; by = blockIdx.y; return; }
    const uint32_t L = blockIdx.y * gridDim.x + blockIdx.x;
    const uint32_t xcd = L & 7u, i = L >> 3;
    bx = i % gridDim.x;
    if (enable == 2) {
        const uint32_t per = (n_quads + 7u) >> 3, j = i / gridDim.x;
        by = j < per ? xcd * per + j : 0xFFFFFFFFu;
        return;
    }
    by = (i / gridDim.x) * 8 + xcd;
}
__device__ __forceinline__ void xcd_remap(uint32_t &bx, uint32_t &by, int enable) {
    if (!enable) { bx = blockIdx.x; by = blockIdx.y; return; }
    const uint32_t nwg = gridDim.x * gridDim.y;
    const uint32_t L = blockIdx.y * gridDim.x + blockIdx.x;
    const uint32_t q = nwg >> 3, r = nwg & 7u, xcd = L & 7u;
    const uint32_t V = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (L >> 3);
    bx = V % gridDim.x;
    by = V / gridDim.x;
}

__device__ __forceinline__ uint64_t shfl_u64(uint64_t v, int src) {
    const uint32_t lo = (uint32_t)__shfl((int)(uint32_t)v, src, 64);
    const uint32_t hi = (uint32_t)__shfl((int)(uint32_t)(v >> 32), src, 64);
    return ((uint64_t)hi << 32) | lo;
}

// 64-lane bitonic sort of (key, val), ascending
__device__ __forceinline__ void bitonic_sort64(uint64_t &key, uint32_t &val, int lane) {
#pragma unroll
    for (int k2 = 2; k2 <= 64; k2 <<= 1) {
#pragma unroll
        for (int j = k2 >> 1; j > 0; j >>= 1) {
            const uint32_t plo = (uint32_t)__shfl_xor((int)(uint32_t)key, j, 64);
            const uint32_t phi = (uint32_t)__shfl_xor((int)(uint32_t)(key >> 32), j, 64);
            const uint32_t pv = (uint32_t)__shfl_xor((int)val, j, 64);
            const uint64_t pk = ((uint64_t)phi << 32) | plo;
            const bool up = (lane & k2) == 0;
            const bool lower = (lane & j) == 0;
            const bool take_min = lower == up;
            const bool sw = take_min ? (pk < key) : (pk > key);
            if (sw) { key = pk; val = pv; }
        }
    }
}

// Fold this tile's candidates of one query into the wave's list, which lives in its final
// global slot (part_keys/part_vals[base .. base+k)): the list is touched only when a
// candidate beats the admission threshold, which is rare once the per-query global threshold
// has tightened, so it costs neither registers nor LDS in the distance loop.
// Returns the list's new k-th key.
template <int S>
__device__ __forceinline__ uint64_t tile_fold(uint64_t *gkeys, uint32_t *gvals, unsigned long long *gthr,
                                              uint64_t gseen, uint64_t local_kth, uint64_t mykey,
                                              uint32_t myval, uint32_t k, int lane, bool fresh = false) {
    WaveTopk<S> tk;
    tk.init();
    if (!fresh) {       // fresh (wave-uniform): this wave has not written the list yet -- it is still the
                        // caller's EMPTY preset, no need to wait for a load to learn that
#pragma unroll
        for (int s = 0; s < S; ++s) {
            const uint32_t e = s * 64 + lane;
            tk.key[s] = e < k ? gkeys[e] : KEY_EMPTY;
            tk.val[s] = e < k ? gvals[e] : 0xFFFFFFFFu;
        }
    }
    if (readlane_u64(tk.key[0], 0) == KEY_EMPTY) {
        // empty list: sort the whole tile once instead of up to 64 single inserts
        uint64_t key = mykey < gseen ? mykey : KEY_EMPTY;
        uint32_t val = myval;
        bitonic_sort64(key, val, lane);
        tk.key[0] = key; tk.val[0] = val;
    } else {
        uint64_t thr = local_kth < gseen ? local_kth : gseen;
        unsigned long long m = __ballot(mykey < thr);
        while (m) {
            const int L = __builtin_ctzll(m);
            const uint64_t x = readlane_u64(mykey, L);
            const uint32_t xv = readlane_u32(myval, L);
            tk.insert(x, xv, lane);
            const uint64_t nk = tk.kth(k);
            thr = nk < thr ? nk : thr;
            m &= m - 1;
            m &= __ballot(mykey < thr);
        }
    }
#pragma unroll
    for (int s = 0; s < S; ++s) {
        const uint32_t e = s * 64 + lane;
        if (e < k) { gkeys[e] = tk.key[s]; gvals[e] = tk.val[s]; }
    }
    const uint64_t nk = tk.kth(k);
    // any wave's k-th key bounds the final k-th key from above: publish it
    if (lane == 0 && nk < gseen) atomicMin(gthr, (unsigned long long)nk);
    return nk;
}

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
                xn[t] = a.row_norm2[srow];
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
    seed_select_body<1>(0u, a.seed_ub, a.seed_tail.n_vals, a.seed_tail.k, a.seed_tail.gthr, a.seed_tail.cand_cnt, a.seed_tail.spilled,
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
    if (by >= *a.n_quads) { if constexpr (U > 1) { if (a.seed_tail.enable) seed_tail_finish(a); } return; }
    const uint4 quad = a.quads[by];
    // a quad wider than this kernel's 16 NG queries (the 8-wave filter kernel takes up to 128) is sampled in
    // slices of 16 NG: blockIdx.z
    const uint32_t sub = blockIdx.z * NQ;
    if (sub >= quad.z) { if constexpr (U > 1) { if (a.seed_tail.enable) seed_tail_finish(a); } return; }
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
    if (wave == 0 || refine) {
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
    uint64_t kth = tk.kth(k);
    uint64_t m1key = readlane_u64(tk.key[0], 0);
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
                s_rowoff[lane] = valid ? (lbeg + row) * rf.dim : 0ull;
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
        const float *x = rf.mat + (valid ? (lbeg + row) : 0ull) * rf.dim;
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
    if ((a.dim % 64) != 0 || !a.mat_blk || a.row_of || !a.seed_ub) return hipErrorInvalidValue;
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
    else if (k <= 256) hipLaunchKernelGGL(seed_select_kernel<4>, dim3(nq), dim3(64), 0, s, seed_ub, n_vals, k, gthr, cand_cnt, spilled, thr_hist, thr_bins, rf);
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
    uint32_t x = v[0];
#pragma unroll
    for (int s = 1; s < QS; ++s) x = (qq >> 6) == (uint32_t)s ? v[s] : x;
    return readlane_u32(x, (int)(qq & 63u));
}
template <int QS>
__device__ __forceinline__ uint64_t qread_u64(const uint64_t (&v)[QS], uint32_t qq) {
    uint64_t x = v[0];
#pragma unroll
    for (int s = 1; s < QS; ++s) x = (qq >> 6) == (uint32_t)s ? v[s] : x;
    return readlane_u64(x, (int)(qq & 63u));
}

// ONCE: every row of the launch is read by exactly one block (a batch that fits one quad, a one-query call above all): the
// operand stream carries the nt policy, so it does not displace the queries' images and thresholds from L2 / the Infinity
// Cache (C3 single query 174 -> 166 us; on batches whose long lists are streamed twice the same hint costs 3.5 %).
// TS: 16-row sub-tiles per wave tile.  4 (64-row tiles) everywhere but the WIDE-QUAD instance <10, 8, .., TS = 2>: 32-row tiles
// halve the accumulator registers per query group, so ONE block holds a quad of 160 queries (120 KB of int8 images) and a
// list that 97..160 queries of the batch probe is streamed once instead of twice (launch_tile_filter, TileArgs::wide_*).
template <int NG, int NW, int S, bool QLDS, int OP, bool PF, bool ONCE, int TS>
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
    constexpr int QS = (NQ + 63) / 64;        // state slots per lane
    // survivors are expanded into the wave's queue a PASS at a time when a tile's do not fit at once: half a
    // group's pairs (queries r < 2 / r >= 2 of every lane: <= 512 entries) for the 4-wave blocks, a quarter
    // (<= 256) for the 8-wave blocks, whose LDS belongs to the staged queries
    constexpr int PASS = NW == 8 ? 256 : NG == 6 ? 128 : 512;      // (96 int8 queries, two blocks per CU: 80 KB each)
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
    const int lane = threadIdx.x & 63;
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
    __shared__ float qst_res[LST ? NQ : 1];           // int8: residual bound (+inf = never skip)
    __shared__ int qst_n2i[LST ? NQ : 1];             // int8: |qi|^2
    uint32_t my_qrow[QS];
    uint64_t my_lkth[QS];
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
        my_lkth[s] = KEY_EMPTY;
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

    const int l15 = lane & 15, kk = lane >> 4;
    const uint64_t blk0 = a.blk_off[c], blk_last = a.blk_off[c + 1] - 1;   // the list's 16-row tiles
    uint32_t npend = 0;
    uint32_t n_exact = 0;
#ifdef PQV_PROFILE_PHASES
    uint64_t ph_k = 0, ph_s = 0, ph_e = 0, ph_em = 0, ph_top_sum = 0, ph_xt = 0; const uint64_t ph_pro = __builtin_amdgcn_s_memtime() - ph_t0;
#endif

    uint64_t cur_gthr[QS];
#pragma unroll
    for (int s = 0; s < QS; ++s) cur_gthr[s] = __hip_atomic_load(a.gthr + my_qrow[s], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    auto eval = [&](uint32_t start, uint32_t count) {
#ifdef PQV_PROFILE_PHASES
        const uint64_t ph_e0 = __builtin_amdgcn_s_memtime();
#endif
        wave_lds_fence();
        // A batch that does not fill the wave (the last one of every wave; most batches of long rows) gives each
        // pair L = 2, 4 or 8 lanes: per round the L lanes of a pair fetch L x 8 consecutive row chunks -- the
        // scattered reads are latency, and a 768-dim row is 24 round trips for one lane, 3 for eight -- and the
        // reference's chain passes through them in chunk order (lane 0's eight adds, then lane 1's, ...), so the
        // sum is bit-identical.  The pair's first lane carries on with the result.
        // (32-row tiles: 16 row chunks in flight per lane -- its batches are larger (a lane per pair: 12 round trips per
        //  768-dim row instead of 24), and the accumulators, dead here, leave the registers)
        constexpr int NB = TS == 2 ? PQV_EVAL_NB_TS2 : 8;
        uint32_t lg = 0;
        while (lg < 3 && (count << (lg + 1)) <= 64u && (Gx % ((2u * NB) << lg)) == 0u) ++lg;      // wave-uniform
        const uint32_t L = 1u << lg;
        const uint32_t pi = (uint32_t)lane >> lg, pj = (uint32_t)lane & (L - 1u);
        const bool valid = pi < count;
        const bool have = valid && pj == 0u;
        const uint32_t pe = pend[start + (valid ? pi : 0)];
        const uint32_t qsl = pe >> QSH;                       // query index in the quad
        const uint64_t roff = r0 + (pe & ((1u << QSH) - 1u));  // row offset in the list
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
        uint64_t my_thr[QS];
#pragma unroll
        for (int s = 0; s < QS; ++s) my_thr[s] = my_lkth[s] < cur_gthr[s] ? my_lkth[s] : cur_gthr[s];
        uint64_t pos;
        if constexpr (LST) pos = qst_cbase[qsl] + roff; else pos = qsel_u64<QS>(my_cbase, qsl) + roff;
        const uint64_t mykey_all =
            (have && pos < a.max_pos) ? (((uint64_t)__float_as_uint(sum) << 32) | (uint64_t)(uint32_t)pos) : KEY_EMPTY;
        // A pair that beats its query's threshold is APPENDED to the query's candidate buffer: one
        // atomic per lane, all lanes in parallel (a sorted per-wave list would cost one global
        // read-modify-write round trip per query, serially -- measured: half of the kernel).
        const uint64_t pair_thr = qsel_u64<QS>(my_thr, qsl);
        const bool pass = mykey_all < pair_thr;
        bool spill = false;
        // Running threshold.  k == 1: the exact distance itself.  Otherwise the query has 12 bins below its
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
        if (pass && k > 1u && a.thr_hist) {
            hb = a.thr_bins[qrow];
            hb_bin = hb.z > 0.0f ? (int)fminf(fmaxf((hb.x - sum) * hb.z, 0.0f), 12.0f) : 0;
        }
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
                a.cand_keys[(uint64_t)qrow * a.cand_cap + idx] = mykey_all;
                a.cand_vals[(uint64_t)qrow * a.cand_cap + idx] = srow;
            } else {
                spill = true;          // buffer full: fall back to this wave's sorted list (slow, exact)
                a.spilled[qrow] = 1u;
            }
            if (k == 1u) {
                atomicMin(a.gthr + qrow, (unsigned long long)(mykey_all | 0xFFFFFFFFull));
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
        unsigned long long todo = __ballot(spill);
        while (todo) {
            const uint32_t qq = readlane_u32(qsl, __builtin_ctzll(todo));
            const bool mine = spill && qsl == qq;
            todo &= ~__ballot(mine);
            const uint64_t mykey = mine ? mykey_all : KEY_EMPTY;
            const uint64_t thr = qread_u64<QS>(my_thr, qq);
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
                                                 qread_u64<QS>(cur_gthr, qq), qread_u64<QS>(my_lkth, qq),
                                                 mykey, srow, k, lane, fresh);
#pragma unroll
                for (int s = 0; s < QS; ++s)
                    if ((uint32_t)(64 * s + lane) == qq) my_lkth[s] = nk;
            }
        }
    };
    auto drain = [&](uint32_t keep_below) {
        while (npend >= keep_below && npend > 0) {
            const uint32_t take = npend < 64 ? npend : 64;
            eval(npend - take, take);
            n_exact += take;
            npend -= take;
        }
        wave_lds_fence();
    };

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
    uint64_t gthr_next[QS];
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
    constexpr int NS = TS == 2 ? PQV_NS_TS2 : (OP != OP_F32 && QLDS && !PF && NG <= 6 && NW == 8) ? PQV_NS_WIDE : 2;      // operand stages in flight
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
    if constexpr (XT && XPF) {
        if (r0 < r1) {
            uint32_t so[TS];
            const __amdgpu_buffer_rsrc_t r = tile_desc(r0, so);
#pragma unroll
            for (int j = 0; j < NS; ++j)
#pragma unroll
                for (int t = 0; t < TS; ++t) xs[j][t] = buf_ld16<ROW_AUX>(r, lane_b, so[t] + j * 1024);
        }
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
    [[maybe_unused]] uint64_t gthr_pf[XTA ? QS : 1];
    if constexpr (XTA) {
#pragma unroll
        for (int s = 0; s < QS; ++s) gthr_pf[s] = cur_gthr[s];
    }
    if constexpr (I8) {
        if (r0 < r1) {
            const uint32_t nv = (r1 - r0 < TROWS) ? (uint32_t)(r1 - r0) : TROWS;
#pragma unroll
            for (int t = 0; t < TS; ++t) {
                uint32_t rr = (uint32_t)(16 * t + l15);
                if (rr >= nv) rr = nv - 1;
                xn2i_next[t] = a.row_n2i[lbeg + r0 + rr];
                xres_next[t] = a.row_res[lbeg + r0 + rr];
            }
            if constexpr (XTA) {
                const uint64_t t2 = r0 + TROWS;
                if (t2 < r1) {
                    const uint32_t nv2 = (r1 - t2 < TROWS) ? (uint32_t)(r1 - t2) : TROWS;
#pragma unroll
                    for (int t = 0; t < TS; ++t) {
                        uint32_t rr = (uint32_t)(16 * t + l15);
                        if (rr >= nv2) rr = nv2 - 1;
                        xn2i_nn[t] = a.row_n2i[lbeg + t2 + rr];
                        xres_nn[t] = a.row_res[lbeg + t2 + rr];
                    }
                }
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
        uint64_t my_thr[QS];
        if constexpr (!XT) {
#pragma unroll
            for (int s = 0; s < QS; ++s) {
                cur_gthr[s] = gthr_next[s];
                gthr_next[s] = __hip_atomic_load(a.gthr + my_qrow[s], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                my_thr[s] = my_lkth[s] < cur_gthr[s] ? my_lkth[s] : cur_gthr[s];
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
        auto mma = [&](const float4 (&x)[TS], uint32_t ks, auto full) {
            const uint32_t chq = ks * 4 + (uint32_t)kk;
            if constexpr (APD == 0) {
#pragma unroll
                for (int g = 0; g < NG; ++g) {
                    if (decltype(full)::value || (uint32_t)g < ng) {
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
                for (int g = 0; g < APD && g < NG; ++g) {
                    const uint32_t gi = decltype(full)::value || (uint32_t)g < ng ? (uint32_t)g : ng - 1u;
                    qb[g] = qb0[gi * gstride];
                }
#pragma unroll
                for (int g = 0; g < NG; ++g) {
                    if (decltype(full)::value || (uint32_t)g < ng) {
                        if (g + APD < NG) {
                            const uint32_t gi = decltype(full)::value || (uint32_t)(g + APD) < ng ? (uint32_t)(g + APD) : ng - 1u;
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
                if ((uint32_t)ks < nks) mma(xt[ks], (uint32_t)ks, std::false_type{});
        }
        // (32-row tiles: always the branch-free body -- a wide quad has 7..10 of its 10 groups, the matrix pipe has room for
        //  the idle ones, whose garbage scores are masked with the queries past cnt, and the per-group branches would cost
        //  the exact wait counts of the A-operand pipeline)
        else if (ng == (uint32_t)NG || TS == 2) kloop(std::true_type{});
        else kloop(std::false_type{});
#ifdef PQV_PROFILE_PHASES
        const uint64_t ph_x0 = __builtin_amdgcn_s_memtime();
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
                    gthr_pf[s] = __hip_atomic_load(a.gthr + my_qrow[s], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                }
#pragma unroll
                for (int t = 0; t < TS; ++t) { xn2i_next[t] = xn2i_nn[t]; xres_next[t] = xres_nn[t]; }
                const uint64_t t2 = tn + TROWS;
                if (t2 < r1) {
                    const uint32_t nv2 = (r1 - t2 < TROWS) ? (uint32_t)(r1 - t2) : TROWS;
#pragma unroll
                    for (int t = 0; t < TS; ++t) {
                        uint32_t rr = (uint32_t)(16 * t + l15);
                        if (rr >= nv2) rr = nv2 - 1;
                        xn2i_nn[t] = a.row_n2i[lbeg + t2 + rr];
                        xres_nn[t] = a.row_res[lbeg + t2 + rr];
                    }
                }
            } else {
#pragma unroll
                for (int s = 0; s < QS; ++s) cur_gthr[s] = __hip_atomic_load(a.gthr + my_qrow[s], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
            if constexpr (I8 && !XTA) {
                if (tn < r1) {
                    const uint32_t nv = (r1 - tn < TROWS) ? (uint32_t)(r1 - tn) : TROWS;
#pragma unroll
                    for (int t = 0; t < TS; ++t) {
                        uint32_t rr = (uint32_t)(16 * t + l15);
                        if (rr >= nv) rr = nv - 1;
                        xn2i_next[t] = a.row_n2i[lbeg + tn + rr];
                        xres_next[t] = a.row_res[lbeg + tn + rr];
                    }
                }
            }
#pragma unroll
            for (int s = 0; s < QS; ++s) my_thr[s] = my_lkth[s] < cur_gthr[s] ? my_lkth[s] : cur_gthr[s];
            if constexpr (!XTA) __builtin_amdgcn_s_waitcnt(0x0F70);         // vmcnt(0): (1) has landed before (2) is issued
            __builtin_amdgcn_sched_barrier(0);
            if (XPF && tn < r1) {
                uint32_t nso[TS];
                const __amdgpu_buffer_rsrc_t nxr = tile_desc(tn, nso);
#pragma unroll
                for (int j = 0; j < NS; ++j)
#pragma unroll
                    for (int t = 0; t < TS; ++t) xs[j][t] = buf_ld16<ROW_AUX>(nxr, lane_b, nso[t] + j * 1024);
            }
            __builtin_amdgcn_sched_barrier(0);
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
                const float thr_d = __uint_as_float((uint32_t)(my_thr[s] >> 32));
                const float qres = qst_res[qi < NQ ? qi : 0];
                const bool open = !(qres <= 3.0e38f) || my_thr[s] == KEY_EMPTY || !(thr_d <= 3.0e38f);
                int a2 = 1 << 29;                              // never skip
                if (qi >= cnt) a2 = -(1 << 30);                // not a query of this quad: always "skipped"
                else if (!open) {
                    const float v = lscale * (sqrtf(thr_d * (1.0f + 4.0f * cmargin)) * 1.000002f + qres + R);
                    const float v2 = fminf(v * v * 1.000002f, 1.0e9f);
                    a2 = ((int)ceilf(v2) + 1 - qst_n2i[qi < NQ ? qi : 0] + 1) >> 1;
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
                const float thr_d = __uint_as_float((uint32_t)(my_thr[s] >> 32));
                float qn;
                bool noskip;
                if constexpr (LST) { qn = qst_qn[qi < NQ ? qi : 0]; noskip = !(qn == qn); }
                else { qn = my_qn[s]; noskip = my_noskip[s]; }
                if (qi < NQ) aq[qi] = qi >= cnt ? INFINITY : (noskip || my_thr[s] == KEY_EMPTY) ? -3.0e38f : alpha * qn - beta * thr_d;
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
            drain(last_tile ? 1u : 64u);
#ifdef PQV_PROFILE_PHASES
            ph_e += __builtin_amdgcn_s_memtime() - ph_c;
#endif
        }
        // slow path: passes of BP of a lane's FW pair bits (cc = TS r + t) per group
        constexpr uint32_t BP = (uint32_t)PASS / 64u < FW ? (uint32_t)PASS / 64u : FW;
        constexpr uint32_t PPG = FW / BP;                    // passes per group
        const uint32_t hend = one_pass ? 0u : PPG * ng + (last_tile ? 1u : 0u);
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
                    pend[at++] = qb + ((cc / TS) << QSH) + rowbase + 16u * (cc % TS);
                }
                npend += readlane_u32(incl, 63);
            }
#ifdef PQV_PROFILE_PHASES
            const uint64_t ph_c = __builtin_amdgcn_s_memtime();
#endif
            drain(g == ng ? 1u : 64u);   // g == ng only in the flush pass
#ifdef PQV_PROFILE_PHASES
            ph_e += __builtin_amdgcn_s_memtime() - ph_c;
#endif
        }
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
template <int NG, int NW, int S, bool QLDS, int OP, bool PF = false, bool ONCE = false, int TS = 4>
static hipError_t launch_wide(const TileArgs &a, size_t lds, hipStream_t s) {
    auto kern = wide_filter_kernel<NG, NW, S, QLDS, OP, PF, ONCE, TS>;
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
    if (a.filter_variant == 0) {
        if ((a.dim % 64) != 0 || a.max_quads == 0 || !a.mat_blk || a.row_of || !a.cand_keys) return hipErrorInvalidValue;
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
                        if (a.wide_width != 160 || !a.item_quad || !a.wide_item_quad || !a.wide_max_items || !a.wide_rows_per_block)
                            return hipErrorInvalidValue;
                        TileArgs w = a;
                        w.quad_width = a.wide_width; w.block_waves = 8; w.wide_width = 0;
                        w.item_quad = a.wide_item_quad; w.item_chunk = a.wide_item_chunk; w.n_items = a.wide_n_items; w.max_items = a.wide_max_items;
                        w.rows_per_block = a.wide_rows_per_block;
                        // the regular instance first: most lists are its, so every query's thresholds have met most of its lists'
                        // first chunks before the popular lists are read (C3: 257 -> 225 exact evaluations per query, the serial
                        // step's kernels 2.09 -> 2.065 ms against the wide instance first)
                        // Cache policy of the row streams.  A list has ONE quad in the regular table: its rows are read once by that
                        // launch and stream with the nt policy (they do not displace query images, thresholds and survivors' rows
                        // from L2 / the Infinity Cache).  In the wide table a list of more than 160 pairs has several quads, which run
                        // side by side and share its rows through the caches -- nt pays there only when such lists are rare: the
                        // caller decides from the previous batch's counts (TileArgs::wide_nt).  Measured, nt on regular / wide /
                        // both: C3 +0.5 / +3.5 / +4 %, mixture +2.5 / -6 / -4 %.
                        hipError_t e = launch_wide<6, 4, S, true, OP_I8, false, true>(a, lds, s);
                        if (e != hipSuccess) return e;
                        return a.wide_nt ? launch_wide<10, 8, S, true, OP_I8, false, true, 2>(w, (size_t)a.wide_width * a.dim, s)
                                         : launch_wide<10, 8, S, true, OP_I8, false, false, 2>(w, (size_t)a.wide_width * a.dim, s);
                    }
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
    return hipGetLastError();
}

hipError_t launch_tile_filter(const TileArgs &a, hipStream_t s) {
    if ((a.filter_variant == 0 ? a.max_quads : a.max_groups) == 0 || a.grid_x == 0) return hipSuccess;
    if (a.k <= 64) return launch_filter_s<1>(a, s);
    if (a.k <= 256) return launch_filter_s<4>(a, s);
    return hipErrorInvalidValue;
}

hipError_t launch_tile_rerank(const TileArgs &a, hipStream_t s) {
    if (a.max_groups == 0 || a.grid_x == 0) return hipSuccess;
    if (a.k <= 64) return launch_tile_s<1>(a, s);
    if (a.k <= 256) return launch_tile_s<4>(a, s);
    return hipErrorInvalidValue;   // larger k uses stream_kernel
}

// ------------------------------------------------------------------------------------
// Batched brute force on the matrix cores (BASELINE config 5: cosine, 1024-query batches).
//
// brute_mfma_kernel: block tile 128 queries x 128 rows, 4 waves as 2 x 2, each wave a
// 64 x 64 sub-tile = 2 x 2 MFMA tiles of v_mfma_f32_32x32x2_f32 (f32 in, f32 accumulate:
// a k-ordered fmaf chain, exact f32 products -- bf16 would miss the 1e-4 tolerance).
// K is walked in 16-float stages: each thread fetches 2+2 float4 (queries / rows) into
// registers one stage ahead, the tiles sit in LDS as [row][k] with a 17-dword row stride so
// the MFMA operand reads (32 rows x same k per half-wave) are bank-conflict-free.
// Epilogue: distance from the score, a sortable key (distance bits | row id) per (query, row),
// compared with the query's admission threshold staged in LDS; the rare survivors are
// appended to the query's candidate buffer with one atomic each.  Nothing of the
// nq x n score matrix is ever written.
// ------------------------------------------------------------------------------------
typedef float f32x16_t __attribute__((ext_vector_type(16)));

__device__ __forceinline__ uint32_t sortable_bits(float d) {
    const uint32_t b = __float_as_uint(d);
    return (b & 0x80000000u) ? ~b : (b | 0x80000000u);
}
__device__ __forceinline__ float unsortable_bits(uint32_t u) {
    return __uint_as_float((u & 0x80000000u) ? (u & 0x7FFFFFFFu) : ~u);
}

constexpr int BR_BM = 128, BR_BN = 128, BR_BK = 16;

template <bool FAST>      // FAST: dim % 16 == 0 -- operand fetches through buffer resources (no bounds or address VALU)
__global__ __launch_bounds__(256) void brute_mfma_kernel(const BruteArgs a) {
    // two stages of [row][16 k] tiles as 16-byte chunks; chunk c of row r sits at position c ^ ((r >> 2) & 3),
    // so the staging writes (one chunk per thread) and the operand reads (two chunks per lane, 16 lanes x 16
    // distinct 16-byte slots of a 256-byte bank window) are both bank-conflict free without padding
    __shared__ float4 As4[2][BR_BM * 4];
    __shared__ float4 Bs4[2][BR_BN * 4];
    __shared__ unsigned long long thr_s[BR_BM];
    __shared__ float qaux_s[BR_BM];

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1;
    const uint64_t n0 = a.row_begin + (uint64_t)blockIdx.x * BR_BN;
    const uint32_t m0 = blockIdx.y * BR_BM;
    const uint32_t dim = a.dim;
    const bool al4 = (dim & 3u) == 0;

    // staging: thread -> (row ld_r / ld_r + 64, chunk ld_ch = 4 consecutive k)
    const int ld_r = tid >> 2, ld_ch = tid & 3;
    float4 ra[2], rb[2];
    // FAST: descriptors at the tile's first query / row; rows past nq / row_end are out of range and read as 0
    const uint64_t qleft = a.nq > m0 ? (uint64_t)(a.nq - m0) * dim * 4 : 0, vleft = a.row_end > n0 ? (a.row_end - n0) * dim * 4 : 0;
    const __amdgpu_buffer_rsrc_t qres = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<float *>(a.queries + (uint64_t)m0 * dim), 0, (int)(qleft > 0xFFFFFFFFull ? 0xFFFFFFFFu : (uint32_t)qleft), 0x00020000);
    const __amdgpu_buffer_rsrc_t vres = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<float *>(a.rows + n0 * dim), 0, (int)(vleft > 0xFFFFFFFFull ? 0xFFFFFFFFu : (uint32_t)vleft), 0x00020000);
    const uint32_t lane_b = ((uint32_t)ld_r * dim + (uint32_t)ld_ch * 4) * 4, half_b = 64u * dim * 4;
    auto fetch = [&](uint32_t k0) {
        if constexpr (FAST) {
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                ra[h] = buf_ld16(qres, lane_b, k0 * 4 + h * half_b);
                rb[h] = buf_ld16(vres, lane_b, k0 * 4 + h * half_b);
            }
            return;
        }
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            const uint32_t qi = m0 + ld_r + 64 * h;
            const uint64_t vj = n0 + ld_r + 64 * h;
            const uint32_t kk = k0 + ld_ch * 4;
            float4 va = make_float4(0.f, 0.f, 0.f, 0.f), vb = va;
            if (qi < a.nq) {
                const float *p = a.queries + (uint64_t)qi * dim + kk;
                if (al4 && kk + 4 <= dim) va = *reinterpret_cast<const float4 *>(p);
                else {
                    if (kk < dim) va.x = p[0];
                    if (kk + 1 < dim) va.y = p[1];
                    if (kk + 2 < dim) va.z = p[2];
                    if (kk + 3 < dim) va.w = p[3];
                }
            }
            if (vj < a.row_end) {
                const float *p = a.rows + vj * dim + kk;
                if (al4 && kk + 4 <= dim) vb = *reinterpret_cast<const float4 *>(p);
                else {
                    if (kk < dim) vb.x = p[0];
                    if (kk + 1 < dim) vb.y = p[1];
                    if (kk + 2 < dim) vb.z = p[2];
                    if (kk + 3 < dim) vb.w = p[3];
                }
            }
            ra[h] = va; rb[h] = vb;
        }
    };
    auto stash = [&](int buf) {
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            const int r = ld_r + 64 * h;
            const int pos = r * 4 + (ld_ch ^ ((r >> 2) & 3));
            As4[buf][pos] = ra[h];
            Bs4[buf][pos] = rb[h];
        }
    };

    f32x16_t acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.0f;

    if (tid < BR_BM) {
        const uint32_t qi = m0 + tid;
        thr_s[tid] = qi < a.nq ? a.thr[qi] : 0ull;
        qaux_s[tid] = qi < a.nq ? a.query_aux[qi] : 0.0f;
    }

    const uint32_t nk = (dim + BR_BK - 1) / BR_BK;
    fetch(0);
    stash(0);
    __syncthreads();
    // MFMA operand roles: lane (l31, lk) owns row l31 of a 32-row tile and, per stage, the 8 consecutive k
    // values 8 lk .. 8 lk + 7 (instruction j of the stage contracts k = 8 lk + j; the order of a dot
    // product's terms is free here) -- two 16-byte LDS reads per tile and stage
    const int l31 = lane & 31, lk = lane >> 5;
    int rowa[2], rowb[2], swa[2], swb[2];
#pragma unroll
    for (int t = 0; t < 2; ++t) {
        rowa[t] = wm * 64 + t * 32 + l31; swa[t] = (rowa[t] >> 2) & 3;
        rowb[t] = wn * 64 + t * 32 + l31; swb[t] = (rowb[t] >> 2) & 3;
    }
    for (uint32_t kt = 0; kt < nk; ++kt) {
        const int buf = (int)(kt & 1u);
        if (kt + 1 < nk) fetch((kt + 1) * BR_BK);
        float av[2][8], bv[2][8];
#pragma unroll
        for (int t = 0; t < 2; ++t) {
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                const float4 x = As4[buf][rowa[t] * 4 + ((2 * lk + h) ^ swa[t])];
                const float4 y = Bs4[buf][rowb[t] * 4 + ((2 * lk + h) ^ swb[t])];
                av[t][4 * h] = x.x; av[t][4 * h + 1] = x.y; av[t][4 * h + 2] = x.z; av[t][4 * h + 3] = x.w;
                bv[t][4 * h] = y.x; bv[t][4 * h + 1] = y.y; bv[t][4 * h + 2] = y.z; bv[t][4 * h + 3] = y.w;
            }
        }
#pragma unroll
        for (int j = 0; j < 8; ++j)
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int jj = 0; jj < 2; ++jj)
                    acc[i][jj] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[i][j], bv[jj][j], acc[i][jj], 0, 0, 0);
        if (kt + 1 < nk) stash(buf ^ 1);      // the other stage: last read before the previous barrier
        __syncthreads();
    }

    // ---- epilogue: C/D layout col = lane & 31, row = (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5)
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const uint64_t vj = n0 + wn * 64 + j * 32 + l31;
        const bool jv = vj < a.row_end;
        const float vaux = jv ? a.row_aux[vj] : 0.0f;
#pragma unroll
        for (int i = 0; i < 2; ++i) {
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int ml = wm * 64 + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * lk;
                const uint32_t qi = m0 + ml;
                const float sc = acc[i][j][r];
                float d;
                if (a.metric == BRUTE_COSINE) d = 1.0f - sc * qaux_s[ml] * vaux;
                else { d = qaux_s[ml] + vaux - 2.0f * sc; d = d < 0.0f ? 0.0f : d; }
                const unsigned long long key =
                    ((unsigned long long)sortable_bits(d) << 32) | (unsigned long long)(uint32_t)vj;
                if (a.dense) {
                    if (jv && qi < a.nq) a.cand[(uint64_t)qi * a.cap + (uint32_t)(vj - a.row_begin)] = key;
                } else if (jv && qi < a.nq && key < thr_s[ml]) {
                    const uint32_t slot = atomicAdd(&a.cand_cnt[qi], 1u);
                    if (slot < a.cap) a.cand[(uint64_t)qi * a.cap + slot] = key;
                }
            }
        }
    }
}

hipError_t launch_brute_mfma(const BruteArgs &a, hipStream_t s) {
    if (a.row_end <= a.row_begin || a.nq == 0) return hipSuccess;
    if (a.dense && a.row_end - a.row_begin > a.cap) return hipErrorInvalidValue;
    const uint64_t nb = (a.row_end - a.row_begin + BR_BN - 1) / BR_BN;
    if (nb > 0x7FFFFFFFull) return hipErrorInvalidValue;
    dim3 grid((uint32_t)nb, (a.nq + BR_BM - 1) / BR_BM);
    // the fast variant needs 16-dim stages that never cross a row end and 32-bit byte offsets inside a tile
    if ((a.dim % 16) == 0 && (uint64_t)a.dim * 4 * 192 < 0x7FFFFFFFull) hipLaunchKernelGGL(brute_mfma_kernel<true>, grid, dim3(256), 0, s, a);
    else hipLaunchKernelGGL(brute_mfma_kernel<false>, grid, dim3(256), 0, s, a);
    return hipGetLastError();
}

// ------------------------------------------------------------------------------------
// Round 3: the same batched brute force on the f16 matrix pipe (16x the f32 rate) as a SCREEN, exact f32 re-scoring
// of what it lets through -- the design of the IVF path applied to BASELINE config 5.
//
// Images: every row and query is L2-normalised, scaled by 2^8 and rounded to f16 (unit vectors: components <= 1, so
// nothing overflows and sub-normals are below 2^-22 of the vector), rows padded with zeros to a multiple of 32 dims.
// With s~ = (image dot product) / 2^16 and s^ the true cosine:
//     |s~ - s^| <= (2^-10 + 2^-22) sum |q^_i v^_i| + (accumulation) <= eps = 1.01 * 2^-10 + 2 dim 2^-24 + 2e-6
// (relative 2^-11 per f16 operand, Cauchy-Schwarz on unit vectors; f32 accumulation of dim exact products; the
// roundings of the normalisation itself).  brute_f16_kernel appends every (query, row) whose LOWER bound
//     cosine:  (1 - s~) - eps            l2:  (|q|^2 + |v|^2 - 2 |q||v| s~) - 2 |q||v| eps
// does not exceed the query's threshold (the k-th smallest EXACT distance so far), brute_rescore_kernel replaces each
// appended entry by its exact f32 key (the arithmetic of the f32 path's epilogue), and the select pass goes on as before:
// no candidate of the final top-k can be lost, and every returned distance is an f32 one.
//
// Block tile 128 queries x 256 rows, 4 waves as 2 x 2, each a 64 x 128 sub-tile = 2 x 4 tiles of
// v_mfma_f32_32x32x16_f16; K in 32-value stages through double-buffered LDS (64 bytes per row and stage, 16-byte chunks
// XOR-swizzled like the f32 kernel's).  An LDS operand read feeds 2 (row side) or 4 (query side) MFMAs: 6 reads per 8
// MFMAs, 96 B/clk/CU at the full matrix rate.  The grid is 1-D and XCD-aware: the 8 query tiles of one row tile run
// back to back on ONE XCD, so a row tile leaves HBM once and serves the other seven from that XCD's L2.
// ------------------------------------------------------------------------------------
constexpr int BH_BM = 128, BH_BN = 256, BH_BK = 32;       // assign_f16_kernel's tile (and the 4-wave form of brute_f16_kernel)

// NWM x NWN waves, each a (32 TM) x (32 TN) sub-tile: block tile BM = 32 TM NWM queries x BN = 32 TN NWN rows.
//   <2, 2, 2, 4>: 128 x 256, 256 threads, two blocks per CU (round 3's first form: 0.29 of the f16 peak on C5 -- PMC: the
//                 matrix pipe busy 29 % of the time, 8.4 TB/s of L2 reads with 92 % hits: bound by the L2 -> LDS traffic
//                 of a tile that does 85 flops per staged byte)
//   <2, 4, 4, 2>: 256 x 256, 512 threads, one block per CU: 128 flops per staged byte
// I8: the same tiles on int8 images (v_mfma_i32_32x32x32_i8: twice the f16 rate, half the staged bytes per flop -- and the
// staging traffic is what bounds the f16 form).  Images: the L2-normalised vector times S = 127 / max |component| (per
// vector), rounded to int8 -- of the vector MINUS its own mid-range b along (1, .., 1), so one-sided data (the bench's
// uniform [0, 1) rows) uses the whole grid:  q^.v^ = (q^ - a 1).(v^ - b 1) + b sum(q^) + a sum(v^) - a b dim.  Per vector
// {1 / S, r >= |(v^ - b 1) - image / S|, b, sum(v^)} and n >= |v^ - b 1| (normalize_i8_kernel).  With D = the image dot
// product (exact in int32) and s~ = D / (S_q S_v) + b sum(q^) + a sum(v^) - a b dim:
//     |s~ - s^| <= n_q r_v + n_v r_q + 3 r_q r_v      (Cauchy-Schwarz on the residuals; |image / S| <= n + r)
// -- a bound per PAIR, wider than the f16 one, so more pairs reach the exact re-scoring; that is still far cheaper than the
// contraction time the int8 pipe saves.
typedef int i32x16_t __attribute__((ext_vector_type(16)));
// ST: 16-byte chunks per row and K stage -- 4 (64-byte stages) or 8 (128-byte stages: half the barriers and twice the MFMAs
// between them; 128 KB of LDS for the 256 x 256 tile, chunks swizzled by the row's low three bits).
// RING (0 or 4): the K stages arrive by direct-to-LDS buffer loads into a ring of RING 64-byte stages instead of through
// registers: three stages are in flight while one is contracted, the loads stay outstanding ACROSS the per-stage barrier
// (counted s_waitcnt vmcnt + a raw s_barrier; the operand reads are inline asm so that the compiler does not drain the
// load queue in front of them) -- with one 8-wave block per CU nothing else hides the L2 / HBM latency of a stage.
typedef float f32x4_raw_t __attribute__((ext_vector_type(4)));
template <int NWM, int NWN, int TM, int TN, bool I8, int ST, int RING = 0>
__global__ __launch_bounds__(64 * NWM * NWN, NWM * NWN == 4 ? 2 : 1) void brute_f16_kernel(const BruteF16Args a) {
    constexpr int BM = 32 * TM * NWM, BN = 32 * TN * NWN, NT = 64 * NWM * NWN;
    static_assert(RING == 0 || (RING == 4 && ST == 4 && BM % (NT / 4) == 0 && BN % (NT / 4) == 0 && TM == 4 && TN == 2), "ring form: 64-byte stages, whole 16-row blocks per wave");
    constexpr int CA = BM * ST / NT, CB = BN * ST / NT;        // 16-byte chunks a thread stages per K stage
    static_assert(BM * ST % NT == 0 && BN * ST % NT == 0 && (ST == 4 || ST == 8), "staging split");
    extern __shared__ float4 brute_lds[];                      // [2][BM * ST] query stages, [2][BN * ST] row stages
    float4 *const As4 = brute_lds, *const Bs4 = brute_lds + 2 * BM * ST;
    // chunk swizzle by row: a ds_read_b128 is served in four groups of 16 lanes -- {0-3, 12-15, 20-27}, {4-11, 16-19, 28-31} and
    // the same + 32 (MI355X_MICROARCH.md #LDS) -- and a group is conflict-free when its 16 chunks cover the 64 banks once.  64-byte
    // rows: chunk ^ bits 2-3 of the row; 128-byte rows: chunk ^ (bit 1 of the row | bits 2-3 << 1) (r & 7 leaves two-way
    // conflicts: rows 0 / 24 and 2 / 26 of a group collide)
    auto sw = [](int r) { return ST == 4 ? (r >> 2) & 3 : ((r >> 1) & 1) | (((r >> 2) & 3) << 1); };
    __shared__ unsigned long long thr_s[BM];
    __shared__ float qaux_s[BM];
    __shared__ float4 qsr_s[I8 ? BM : 1];            // int8 form: {1 / S, r, a, sum} of the tile's queries
    __shared__ float qn_s[I8 ? BM : 1];              //            |q^ - a 1|
    __shared__ float4 qk_s[BM];                      // quick-screen constants of the tile's queries (cosine)

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave / NWN, wn = wave % NWN;
    // id -> (XCD, slot): XCD x takes the row tiles = x (mod 8), each followed by all of its query tiles
    const uint32_t ny = (a.nq + BM - 1) / BM;
    const uint32_t xcd = blockIdx.x & 7u, slot = blockIdx.x >> 3;
    const uint64_t vt = (uint64_t)(slot / ny) * 8u + xcd;
    const uint32_t qt = slot % ny;
    const uint64_t n0 = a.row_begin + vt * BN;
    if (n0 >= a.row_end) return;
    const uint32_t m0 = qt * BM;
    const uint32_t dp = a.dim_p;                     // padded dims (a multiple of 32, 2 bytes each; int8: of 64, 1 byte each)
    const uint32_t rbytes = I8 ? dp : dp * 2;        // bytes per image row; a K stage is 64 of them

    constexpr int RPS = NT / ST;                     // rows one staging step of the block covers
    const int ld_r = tid / ST, ld_ch = tid % ST;     // staging: row ld_r (+ RPS h), 16-byte chunk ld_ch of the stage
    float4 ra[CA], rb[CB];
    const uint64_t qleft = a.nq > m0 ? (uint64_t)(a.nq - m0) * rbytes : 0, vleft = (a.row_end - n0) * rbytes;
    const char *qbase = I8 ? reinterpret_cast<const char *>(a.q8) : reinterpret_cast<const char *>(a.q16);
    const char *vbase = I8 ? reinterpret_cast<const char *>(a.v8) : reinterpret_cast<const char *>(a.v16);
    const __amdgpu_buffer_rsrc_t qres = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<char *>(qbase + (uint64_t)m0 * rbytes), 0, (int)(qleft > 0xFFFFFFFFull ? 0xFFFFFFFFu : (uint32_t)qleft), 0x00020000);
    const __amdgpu_buffer_rsrc_t vres = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<char *>(vbase + n0 * rbytes), 0, (int)(vleft > 0xFFFFFFFFull ? 0xFFFFFFFFu : (uint32_t)vleft), 0x00020000);
    const uint32_t lane_b = (uint32_t)ld_r * rbytes + (uint32_t)ld_ch * 16, step_b = (uint32_t)RPS * rbytes;
    auto fetch = [&](uint32_t kb) {                  // kb: byte offset of the stage inside a row
#pragma unroll
        for (int h = 0; h < CA; ++h) ra[h] = buf_ld16(qres, lane_b, kb + h * step_b);
#pragma unroll
        for (int h = 0; h < CB; ++h) rb[h] = buf_ld16(vres, lane_b, kb + h * step_b);
    };
    auto stash = [&](int buf) {
#pragma unroll
        for (int h = 0; h < CA; ++h) { const int r = ld_r + RPS * h; As4[buf * BM * ST + r * ST + (ld_ch ^ sw(r))] = ra[h]; }
#pragma unroll
        for (int h = 0; h < CB; ++h) { const int r = ld_r + RPS * h; Bs4[buf * BN * ST + r * ST + (ld_ch ^ sw(r))] = rb[h]; }
    };

    using acc_t = std::conditional_t<I8, i32x16_t, f32x16_t>;
    acc_t acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0;

    if (tid < BM) {
        const uint32_t qi = m0 + tid;
        thr_s[tid] = qi < a.nq ? a.thr[qi] : 0ull;
        qaux_s[tid] = qi < a.nq ? a.query_aux[qi] : 0.0f;
        if constexpr (I8) {
            qsr_s[tid] = qi < a.nq ? a.query_sr[qi] : make_float4(0.0f, 0.0f, 0.0f, 0.0f);
            qn_s[tid] = qi < a.nq ? a.query_n[qi] : 0.0f;
        }
        // Quick screen of the cosine epilogue (one compare, or three FMAs and a compare, per pair instead of the full bound --
        // the epilogue's VALU work took longer than the MFMAs): per query the constants of a test that every pair the exact
        // test keeps also passes; the exact test (below) runs only on what passes.
        //   f16:  keep  =>  acc 2^-16 >= (1 - T) - eps - slack                                      qk = {that / 2^-16}
        //   int8: keep  =>  D iv_j + (a / iq) A_j + (sum_q / iq) C_j >= ((1 - T) - K - slack) / iq,   qk = {a / iq, sum_q / iq, -rhs}
        //         K = the per-pair eps with the row's terms replaced by their corpus-wide maxima (a.row_max), A_j = sum_v - b_j dim, C_j = b_j
        float4 qk = make_float4(0.0f, 0.0f, __builtin_inff(), 0.0f);          // int8: always passes (rhs = -inf); f16: x = -inf
        if (!I8) qk.x = -__builtin_inff();
        if (qi < a.nq && a.metric == BRUTE_COSINE) {
            const uint32_t th = (uint32_t)(thr_s[tid] >> 32);
            if (th < 0xFF800000u) {                  // a finite threshold distance T (else: no threshold yet, everything passes)
                const float T = unsortable_bits(th);
                if constexpr (I8) {
                    const float4 q = qsr_s[tid];
                    const float mA = a.row_max[0], mB = a.row_max[1], mC = a.row_max[2], mE = a.row_max[3];
                    const float K = (q.y * mB + qn_s[tid] * mE) * 1.00002f + a.eps + a.eps_sum * (fabsf(q.z) + mC);
                    const float slack = 1.0e-5f * (2.0f + fabsf(q.z) * mA + fabsf(q.w) * mC);
                    const float rhs = (1.0f - T) - K - slack;
                    qk = make_float4(q.z / q.x, q.w / q.x, -(rhs / q.x), 0.0f);      // (iq = 0: NaN / inf -- the pair passes)
                } else {
                    qk.x = ((1.0f - T) - a.eps - 4.0e-6f) * 65536.0f;
                }
            }
        }
        qk_s[tid] = qk;
    }

    const uint32_t nk = rbytes / (16 * ST);
    if constexpr (RING == 0) {
        fetch(0);
        stash(0);
    }
    __syncthreads();
    // operand roles of v_mfma_f32_32x32x16_f16: lane (l31, lk) owns row l31 of a 32-row tile and the 8 consecutive k
    // values 8 lk .. 8 lk + 7 of the instruction's 16; MFMA j of a stage takes chunk 2 j + lk (term order is free)
    const int l31 = lane & 31, lk = lane >> 5;
    int rowa[TM], rowb[TN], swa[TM], swb[TN];
#pragma unroll
    for (int t = 0; t < TM; ++t) { rowa[t] = wm * 32 * TM + t * 32 + l31; swa[t] = sw(rowa[t]); }
#pragma unroll
    for (int t = 0; t < TN; ++t) { rowb[t] = wn * 32 * TN + t * 32 + l31; swb[t] = sw(rowb[t]); }
    if constexpr (RING != 0) {
        constexpr int NW = NT / 64, RA = BM / 16 / NW, RB = BN / 16 / NW;      // 16-row blocks (1 KiB of a stage) per wave
        constexpr uint32_t STG = (uint32_t)(BM + BN) * 4;                      // float4s of a ring slot: BM query rows, BN corpus rows
        typedef __attribute__((address_space(3))) void lds_void;
        const uint32_t lds0 = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) char *)brute_lds;
        // lane l of a load instruction fills 16-byte slot l of its 1 KiB: row l / 4 of the block, stored chunk l % 4 -- which holds the
        // row's chunk (l % 4) ^ sw(row) (the block's 16 rows start at a multiple of 16, so sw depends on l only)
        const uint32_t voff = (uint32_t)(lane >> 2) * rbytes + (uint32_t)(((lane & 3) ^ ((lane >> 4) & 3)) << 4);
        const uint32_t wave_u = (uint32_t)__builtin_amdgcn_readfirstlane(wave);      // (scalar: LDS base and buffer offset of a load are wave-uniform)
        auto issue = [&](uint32_t st) {
            float4 *dst = brute_lds + (st & (RING - 1)) * STG;
#pragma unroll
            for (int h = 0; h < RA; ++h) {
                const uint32_t blk = wave_u * RA + h;
                __builtin_amdgcn_raw_ptr_buffer_load_lds(qres, (lds_void *)(dst + blk * 64), 16, (int)voff, (int)(blk * 16 * rbytes + st * 64), 0, 0);
            }
#pragma unroll
            for (int h = 0; h < RB; ++h) {
                const uint32_t blk = wave_u * RB + h;
                __builtin_amdgcn_raw_ptr_buffer_load_lds(vres, (lds_void *)(dst + BM * 4 + blk * 64), 16, (int)voff, (int)(blk * 16 * rbytes + st * 64), 0, 0);
            }
        };
        uint32_t offa[2][TM], offb[2][TN];          // byte offsets of this lane's operands inside a ring slot, K steps 0 / 1
#pragma unroll
        for (int j = 0; j < 2; ++j) {
#pragma unroll
            for (int t = 0; t < TM; ++t) offa[j][t] = lds0 + (uint32_t)(rowa[t] * 4 + ((2 * j + lk) ^ swa[t])) * 16;
#pragma unroll
            for (int t = 0; t < TN; ++t) offb[j][t] = lds0 + (uint32_t)(BM * 4 + rowb[t] * 4 + ((2 * j + lk) ^ swb[t])) * 16;
        }
        for (uint32_t st = 0; st < 3 && st < nk; ++st) issue(st);
        for (uint32_t kt = 0; kt < nk; ++kt) {
            // this wave's part of stage kt has landed (later stages stay in flight: RA + RB loads each); after the barrier
            // everybody's has, and everybody is done with stage kt - 1, whose slot stage kt + 3 takes
            const uint32_t ahead = nk - 1 - kt;
            if (ahead >= 2) asm volatile("s_waitcnt vmcnt(%0)\n\ts_barrier" :: "n"(2 * (RA + RB)) : "memory");
            else if (ahead == 1) asm volatile("s_waitcnt vmcnt(%0)\n\ts_barrier" :: "n"(RA + RB) : "memory");
            else asm volatile("s_waitcnt vmcnt(0)\n\ts_barrier" ::: "memory");
            if (kt + 3 < nk) issue(kt + 3);
            const uint32_t sb = (kt & (RING - 1)) * STG * 16;
            f32x4_raw_t oa[2][TM], ob[2][TN];
#pragma unroll
            for (int j = 0; j < 2; ++j) {
#pragma unroll
                for (int t = 0; t < TM; ++t) asm volatile("ds_read_b128 %0, %1" : "=v"(oa[j][t]) : "v"(offa[j][t] + sb));
#pragma unroll
                for (int t = 0; t < TN; ++t) asm volatile("ds_read_b128 %0, %1" : "=v"(ob[j][t]) : "v"(offb[j][t] + sb));
            }
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                // LDS reads return in order: K step 0's six operands are there once six reads remain outstanding
                if (j == 0) asm volatile("s_waitcnt lgkmcnt(%6)" : "+v"(oa[0][0]), "+v"(oa[0][1]), "+v"(oa[0][2]), "+v"(oa[0][3]), "+v"(ob[0][0]), "+v"(ob[0][1]) : "n"(TM + TN));
                else { __builtin_amdgcn_sched_barrier(0); }
                if (j == 1) asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(oa[1][0]), "+v"(oa[1][1]), "+v"(oa[1][2]), "+v"(oa[1][3]), "+v"(ob[1][0]), "+v"(ob[1][1]));
#pragma unroll
                for (int i = 0; i < TM; ++i)
#pragma unroll
                    for (int jj = 0; jj < TN; ++jj) {
                        if constexpr (I8)
                            acc[i][jj] = __builtin_amdgcn_mfma_i32_32x32x32_i8(__builtin_bit_cast(i32x4_acc, oa[j][i]), __builtin_bit_cast(i32x4_acc, ob[j][jj]), acc[i][jj], 0, 0, 0);
                        else
                            acc[i][jj] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8_t, oa[j][i]), __builtin_bit_cast(f16x8_t, ob[j][jj]), acc[i][jj], 0, 0, 0);
                    }
            }
        }
        __syncthreads();
    } else
    for (uint32_t kt = 0; kt < nk; ++kt) {
        const int buf = (int)(kt & 1u);
        if (kt + 1 < nk) fetch((kt + 1) * 16 * ST);
        // (int8: v_mfma_i32_32x32x32_i8 takes 16 bytes per lane as well -- lane group lk owns one half of the instruction's 32
        //  k values; which half is immaterial, both operands read the same chunk)
        // operands of K step j + 1 are read while the MFMAs of step j run (two register sets; the compiler on its own reuses
        // one set and waits for every read right in front of its MFMA)
        float4 av[2][TM], bv[2][TN];
        auto lds_read = [&](int j, int set) {
#pragma unroll
            for (int t = 0; t < TM; ++t) av[set][t] = As4[buf * BM * ST + rowa[t] * ST + ((2 * j + lk) ^ swa[t])];
#pragma unroll
            for (int t = 0; t < TN; ++t) bv[set][t] = Bs4[buf * BN * ST + rowb[t] * ST + ((2 * j + lk) ^ swb[t])];
        };
        lds_read(0, 0);
#pragma unroll
        for (int j = 0; j < ST / 2; ++j) {
            const int set = j & 1;
            if (j + 1 < ST / 2) lds_read(j + 1, set ^ 1);
            __builtin_amdgcn_sched_barrier(0);          // the reads are issued before this step's MFMAs, which cover their latency
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int jj = 0; jj < TN; ++jj) {
                    if constexpr (I8)
                        acc[i][jj] = __builtin_amdgcn_mfma_i32_32x32x32_i8(__builtin_bit_cast(i32x4_acc, av[set][i]), __builtin_bit_cast(i32x4_acc, bv[set][jj]), acc[i][jj], 0, 0, 0);
                    else
                        acc[i][jj] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8_t, av[set][i]), __builtin_bit_cast(f16x8_t, bv[set][jj]), acc[i][jj], 0, 0, 0);
                }
            __builtin_amdgcn_sched_barrier(0);
        }
        if (kt + 1 < nk) stash(buf ^ 1);      // the other stage: last read before the previous barrier
        __syncthreads();
    }

    // ---- epilogue: C/D layout col = lane & 31, row = (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5)
    const float inv = 1.52587890625e-05f;            // 2^-16: the two images carry 2^8 each
    uint64_t vjs[TN];
    bool jvs[TN];
    float vauxs[TN];
    [[maybe_unused]] float4 vsrs[TN];
    [[maybe_unused]] float vns[TN];
#pragma unroll
    for (int j = 0; j < TN; ++j) {
        vjs[j] = n0 + wn * 32 * TN + j * 32 + l31;
        jvs[j] = vjs[j] < a.row_end;
        vauxs[j] = (jvs[j] && a.metric != BRUTE_COSINE) ? a.row_aux[vjs[j]] : 0.0f;          // l2: |v|^2 (cosine: no row term)
        vsrs[j] = make_float4(0.0f, 0.0f, 0.0f, 0.0f); vns[j] = 0.0f;
        if constexpr (I8) { if (jvs[j]) { vsrs[j] = a.row_sr[vjs[j]]; vns[j] = a.row_n[vjs[j]]; } }
    }
    // the exact test of one pair: lower bound of its distance against the query's threshold key
    auto exact = [&](int ml, int j, auto accv) {
        const uint32_t qi = m0 + ml;
        const float vaux = vauxs[j];
        float sc, eps;
        if constexpr (I8) {
            const float4 qsr = qsr_s[ml], vsr = vsrs[j];
            sc = ((float)accv * qsr.x) * vsr.x + (vsr.z * qsr.w + qsr.z * (vsr.w - vsr.z * a.dim_f));
            // n_q r_v + n_v r_q + 3 r_q r_v, rounded up, + the f32 roundings of both sides (a.eps) and of the two
            // component sums (a.eps_sum per unit of |a| + |b|)
            eps = (qn_s[ml] * vsr.y + qsr.y * (vns[j] + 3.0f * vsr.y)) * 1.00001f + a.eps + a.eps_sum * (fabsf(qsr.z) + fabsf(vsr.z));
        } else {
            sc = accv * inv; eps = a.eps;
        }
        float lb;
        if (a.metric == BRUTE_COSINE) lb = (1.0f - sc) - eps;
        else {
            const float qv = sqrtf(qaux_s[ml] * vaux) * 1.000001f;     // |q| |v|
            lb = (qaux_s[ml] + vaux - 2.0f * qv * sc) - 2.0f * qv * eps - 1.0e-6f * (qaux_s[ml] + vaux);
            lb = lb < 0.0f ? 0.0f : lb;
        }
        // (a NaN bound sorts last, like a NaN distance in the f32 kernel)
        const bool keep = !((unsigned long long)sortable_bits(lb) > (thr_s[ml] >> 32));
        if (jvs[j] && qi < a.nq && keep) {
            const uint32_t slot2 = atomicAdd(&a.cand_cnt[qi], 1u);
            if (slot2 < a.cap) a.cand[(uint64_t)qi * a.cap + slot2] = ((unsigned long long)sortable_bits(lb) << 32) | (uint32_t)vjs[j];
        }
    };
    const bool quick = a.metric == BRUTE_COSINE;
    [[maybe_unused]] float rowA[TN], rowC[TN], rowI[TN];       // int8 quick screen: A_j, C_j, iv_j
    if constexpr (I8) {
#pragma unroll
        for (int j = 0; j < TN; ++j) { rowI[j] = vsrs[j].x; rowC[j] = vsrs[j].z; rowA[j] = vsrs[j].w - vsrs[j].z * a.dim_f; }
    }
#pragma unroll
    for (int i = 0; i < TM; ++i) {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int ml = wm * 32 * TM + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * lk;
            const float4 qk = qk_s[ml];
#pragma unroll
            for (int j = 0; j < TN; ++j) {
                bool pass = true;
                if (quick) {
                    if constexpr (I8) pass = !(__builtin_fmaf((float)acc[i][j][r], rowI[j], __builtin_fmaf(qk.x, rowA[j], __builtin_fmaf(qk.y, rowC[j], qk.z))) < 0.0f);
                    else pass = !(acc[i][j][r] < qk.x);
                }
                if (pass) exact(ml, j, acc[i][j][r]);
            }
        }
    }
}
hipError_t launch_brute_f16(const BruteF16Args &a, hipStream_t s) {
    if (a.row_end <= a.row_begin || a.nq == 0) return hipSuccess;
    const bool i8 = a.v8 != nullptr;
    if (i8 ? ((a.dim_p % 64) != 0 || !a.q8 || !a.row_sr || !a.query_sr || !a.row_n || !a.query_n || !a.row_max) : (a.dim_p % BH_BK) != 0) return hipErrorInvalidValue;
    if ((uint64_t)a.dim_p * 2 * 512 >= 0x7FFFFFFFull) return hipErrorInvalidValue;
    // 256 x 256 tiles from 256 queries on (PQV_BRUTE_TILE=128 keeps the 128 x 256 form for comparison)
    static const int tile_env = [] { const char *e = std::getenv("PQV_BRUTE_TILE"); return e ? std::atoi(e) : 0; }();
    const bool big = tile_env == 256 || (tile_env != 128 && a.nq > 128);
    const uint64_t bm = big ? 256 : 128, bn = 256;
    const uint64_t nb = (a.row_end - a.row_begin + bn - 1) / bn, ny = (a.nq + bm - 1) / bm;
    const uint64_t blocks = (nb + 7) / 8 * 8 * ny;
    if (blocks > 0x7FFFFFFFull) return hipErrorInvalidValue;
    // 128-byte K stages for the 256 x 256 tile where the image rows are a multiple of them (PQV_BRUTE_STAGE=64 keeps 64)
    static const int stage_env = [] { const char *e = std::getenv("PQV_BRUTE_STAGE"); return e ? std::atoi(e) : 0; }();
    const uint64_t rbytes = i8 ? a.dim_p : (uint64_t)a.dim_p * 2;
    const bool st8 = big && stage_env != 64 && (rbytes % 128) == 0;
    auto launch = [&](auto kern, uint32_t threads, size_t lds) -> hipError_t {
        if (lds > 65536) {
            const hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
            if (e != hipSuccess) { (void)hipGetLastError(); return e; }
        }
        hipLaunchKernelGGL(kern, dim3((uint32_t)blocks), dim3(threads), lds, s, a);
        return hipGetLastError();
    };
    // the ring form (direct-to-LDS stages, loads in flight across the barriers) of the 256 x 256 tile measures the same as the
    // register-staged one with 128-byte stages (C5, int8: 19.7 against 19.5 ms) -- opt-in: PQV_BRUTE_RING=1 (read per launch: tests)
    const bool ring_env = [] { const char *e = std::getenv("PQV_BRUTE_RING"); return e && *e == '1'; }();
    if (big && ring_env && (rbytes % 64) == 0) {
        if (i8) return launch(brute_f16_kernel<2, 4, 4, 2, true, 4, 4>, 512, 4 * 512 * 4 * 16);
        return launch(brute_f16_kernel<2, 4, 4, 2, false, 4, 4>, 512, 4 * 512 * 4 * 16);
    }
    // (measured and dropped: four waves of 128 x 128 -- 16 accumulator tiles per wave in AGPRs, one wave per SIMD, half the LDS
    //  reads per MFMA: hipcc keeps 1 KB of scratch per lane for it and the launch takes 136 ms against 18.5)
    if (i8) {
        if (st8) return launch(brute_f16_kernel<2, 4, 4, 2, true, 8>, 512, 2 * 512 * 8 * 16);
        if (big) return launch(brute_f16_kernel<2, 4, 4, 2, true, 4>, 512, 2 * 512 * 4 * 16);
        return launch(brute_f16_kernel<2, 2, 2, 4, true, 4>, 256, 2 * 384 * 4 * 16);
    }
    if (st8) return launch(brute_f16_kernel<2, 4, 4, 2, false, 8>, 512, 2 * 512 * 8 * 16);
    if (big) return launch(brute_f16_kernel<2, 4, 4, 2, false, 4>, 512, 2 * 512 * 4 * 16);
    return launch(brute_f16_kernel<2, 2, 2, 4, false, 4>, 256, 2 * 384 * 4 * 16);
}

// int8 images of the L2-normalised rows (brute_f16_kernel<.., I8>): one wave per row.  v^ = row * rnorm (f32); b = the mid-range
// of its components; S = 127 / max |v^_i - b|; image_i = rint((v^_i - b) S) (padding: 0); sr[r] = {1 / S, r_v, b, sum(v^)},
// nrm[r] = n_v with r_v >= |(v^ - b 1) - image / S| (against the stored 1 / S) and n_v >= |v^ - b 1|, both rounded up.
// A zero row has image 0 and r_v = n_v = 0 (its cosine is what the exact pass says); a row with a non-finite value gets
// r_v = +inf and is never skipped.
__global__ __launch_bounds__(256) void normalize_i8_kernel(const float *__restrict__ rows, const float *__restrict__ rnorm,
                                                          uint64_t n, uint32_t dim, uint32_t dim_p, int8_t *__restrict__ out,
                                                          float4 *__restrict__ sr, float *__restrict__ nrm, uint32_t *__restrict__ maxima) {
    const int lane = threadIdx.x & 63;
    const uint64_t w = (uint64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    const uint64_t nw = (uint64_t)gridDim.x * 4;
    float mxA = 0.0f, mxB = 0.0f, mxC = 0.0f, mxE = 0.0f;       // corpus-wide maxima of |sum - b dim|, n + 3 r, |b|, r (brute_f16_kernel's quick screen)
    for (uint64_t r = w; r < n; r += nw) {
        const float *p = rows + r * dim;
        const float rn = rnorm[r];
        float mx = -3.0e38f, mn = 3.0e38f, sum = 0.0f;
        bool bad = !(rn == rn) || rn > 3.0e38f;
        for (uint32_t e = lane; e < dim; e += 64) {
            const float v = p[e] * rn;
            bad |= !(fabsf(v) <= 3.0e38f);
            mx = fmaxf(mx, v); mn = fminf(mn, v); sum += v;
        }
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) {
            mx = fmaxf(mx, __shfl_xor(mx, off, 64)); mn = fminf(mn, __shfl_xor(mn, off, 64)); sum += __shfl_xor(sum, off, 64);
        }
        bad = __any(bad);
        const float b = bad ? 0.0f : 0.5f * (mx + mn);
        const float half = bad ? 0.0f : fmaxf(mx - b, b - mn);
        const float S = half > 0.0f ? 127.0f / half : 0.0f;
        const float invS = S > 0.0f ? 1.0f / S : 0.0f;
        float res2 = 0.0f, n2 = 0.0f;
        for (uint32_t e = lane; e < dim_p; e += 64) {
            const float v = (e < dim && !bad) ? p[e] * rn - b : 0.0f;
            float qf = rintf(v * S);
            qf = fminf(fmaxf(qf, -127.0f), 127.0f);
            out[r * dim_p + e] = (int8_t)(int)qf;
            const float d = v - qf * invS;
            res2 += d * d; n2 += v * v;
        }
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) { res2 += __shfl_xor(res2, off, 64); n2 += __shfl_xor(n2, off, 64); }
        const float rr = bad ? __builtin_inff() : sqrtf(res2) * 1.0001f + 1.0e-6f, nn = bad ? __builtin_inff() : sqrtf(n2) * 1.0001f + 1.0e-6f;
        if (lane == 0) {
            sr[r] = make_float4(invS, rr, b, bad ? 0.0f : sum);
            nrm[r] = nn;
        }
        mxA = fmaxf(mxA, bad ? __builtin_inff() : fabsf(sum - b * (float)dim) * 1.00001f);
        mxB = fmaxf(mxB, (nn + 3.0f * rr) * 1.00001f); mxC = fmaxf(mxC, fabsf(b)); mxE = fmaxf(mxE, rr * 1.00001f);
    }
    if (maxima && lane == 0) {        // non-negative floats (or +inf): their bit patterns order like the values
        atomicMax(&maxima[0], __float_as_uint(mxA)); atomicMax(&maxima[1], __float_as_uint(mxB));
        atomicMax(&maxima[2], __float_as_uint(mxC)); atomicMax(&maxima[3], __float_as_uint(mxE));
    }
}
hipError_t launch_normalize_i8(const float *rows, const float *rnorm, uint64_t n, uint32_t dim, uint32_t dim_p, void *out, void *sr, float *nrm, float *maxima, hipStream_t s) {
    if (n == 0) return hipSuccess;
    uint64_t blocks = (n + 3) / 4;
    if (blocks > 65536) blocks = 65536;
    hipLaunchKernelGGL(normalize_i8_kernel, dim3((uint32_t)blocks), dim3(256), 0, s, rows, rnorm, n, dim, dim_p, static_cast<int8_t *>(out), static_cast<float4 *>(sr), nrm, reinterpret_cast<uint32_t *>(maxima));
    return hipGetLastError();
}

// L2-normalised f16 images (x 2^8), zero-padded to dim_p: one wave per row.  rnorm: 1 / |row| (0 for a zero row).
__global__ __launch_bounds__(256) void normalize_f16_kernel(const float *__restrict__ rows, const float *__restrict__ rnorm,
                                                           uint64_t n, uint32_t dim, uint32_t dim_p, uint16_t *__restrict__ out) {
    const int lane = threadIdx.x & 63;
    const uint64_t w = (uint64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    const uint64_t nw = (uint64_t)gridDim.x * 4;
    for (uint64_t r = w; r < n; r += nw) {
        const float *p = rows + r * dim;
        const float sc = rnorm[r] * 256.0f;
        for (uint32_t e = lane; e < dim_p; e += 64) {
            float v = e < dim ? p[e] * sc : 0.0f;
            v = fminf(fmaxf(v, -65504.0f), 65504.0f);          // (a non-finite row: the exact pass decides)
            const _Float16 h = (_Float16)v;
            out[r * dim_p + e] = __builtin_bit_cast(uint16_t, h);
        }
    }
}
hipError_t launch_normalize_f16(const float *rows, const float *rnorm, uint64_t n, uint32_t dim, uint32_t dim_p, void *out, hipStream_t s) {
    if (n == 0) return hipSuccess;
    uint64_t blocks = (n + 3) / 4;
    if (blocks > 65536) blocks = 65536;
    hipLaunchKernelGGL(normalize_f16_kernel, dim3((uint32_t)blocks), dim3(256), 0, s, rows, rnorm, n, dim, dim_p, static_cast<uint16_t *>(out));
    return hipGetLastError();
}

// exact f32 keys for the entries the f16 screen appended to the candidate buffers: entry slots [first[q], min(cnt[q], cap)),
// one wave per entry (the row id sits in the entry's low word).  Same arithmetic as brute_mfma_kernel's epilogue on an
// f32 dot product.
__global__ __launch_bounds__(256) void brute_rescore_kernel(const BruteArgs a, const uint32_t *__restrict__ first) {
    const int lane = threadIdx.x & 63;
    const uint32_t q = blockIdx.y;
    uint32_t cnt = a.cand_cnt[q];
    if (cnt > a.cap) cnt = a.cap;
    const float *qp = a.queries + (uint64_t)q * a.dim;
    const float qaux = a.query_aux[q];
    for (uint32_t e = first[q] + blockIdx.x * 4u + (threadIdx.x >> 6); e < cnt; e += gridDim.x * 4u) {
        unsigned long long *ent = a.cand + (uint64_t)q * a.cap + e;
        const uint32_t row = (uint32_t)*ent;
        const float *vp = a.rows + (uint64_t)row * a.dim;
        float s = 0.0f;
        if ((a.dim & 3u) == 0) {
            for (uint32_t d = lane * 4; d < a.dim; d += 256) {
                const float4 x = *reinterpret_cast<const float4 *>(qp + d), y = *reinterpret_cast<const float4 *>(vp + d);
                s = fmaf(x.x, y.x, s); s = fmaf(x.y, y.y, s); s = fmaf(x.z, y.z, s); s = fmaf(x.w, y.w, s);
            }
        } else {
            for (uint32_t d = lane; d < a.dim; d += 64) s = fmaf(qp[d], vp[d], s);
        }
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) s += __shfl_xor(s, off, 64);
        float d;
        const float vaux = a.row_aux[row];
        if (a.metric == BRUTE_COSINE) d = 1.0f - s * qaux * vaux;
        else { d = qaux + vaux - 2.0f * s; d = d < 0.0f ? 0.0f : d; }
        if (lane == 0) *ent = ((unsigned long long)sortable_bits(d) << 32) | row;
    }
}
hipError_t launch_brute_rescore(const BruteArgs &a, const uint32_t *first, hipStream_t s) {
    if (a.nq == 0) return hipSuccess;
    hipLaunchKernelGGL(brute_rescore_kernel, dim3(32, a.nq), dim3(256), 0, s, a, first);
    return hipGetLastError();
}

// ------------------------------------------------------------------------------------
// Round 3: the k-means assignment (index.rs:395-430 Lloyd, :189-206 + :244-257 final) as a dense contraction on the f16
// matrix pipe + exact re-scoring -- the brute-force design with the roles swapped: every ROW keeps a threshold and a
// candidate list, the CENTROIDS are the streamed side.
//
//   images      x^ = (x - mu) / |x - mu| * 2^8 in f16 for rows and centroids alike (mu: any fixed vector -- the distance is
//               translation invariant; centring shrinks |x - mu| |c - mu| and with it the bound's slack)
//   assign_f16_kernel   one block = 128 rows against ALL centroids, 256 at a time (4 waves as 2 x 2, 64 rows x 128
//               centroids each, v_mfma_f32_32x32x16_f16, K in 32-value stages through double-buffered LDS).  With s~ the image
//               dot product / 2^16:  d~ = |a|^2 + |b|^2 - 2 |a||b| s~,  |d~ - d| <= err = 2 |a||b| eps + 4e-6 (|a|^2 + |b|^2)
//               (eps as in brute_f16_kernel), and the reference's computed distance lies within (1 +- cm) of d.  Per row the
//               smallest UPPER bound seen so far is a running threshold (LDS); every centroid whose LOWER bound does not
//               exceed it is appended to the row's candidate list.  The true argmin is never dropped: its lower bound is
//               below its own upper bound, which is below every threshold the row ever had.
//   assign_rescore_kernel   exact reference-order distances (index.rs:461-480) of a row's candidates, four lanes per row,
//               argmin by (distance bits, centroid index) = strict '<' in ascending index order; a row whose list
//               overflowed is compared with every centroid.
// ------------------------------------------------------------------------------------
__global__ __launch_bounds__(256, 2) void assign_f16_kernel(const AssignF16Args a) {
    __shared__ float4 As4[2][BH_BM * 4];
    __shared__ float4 Bs4[2][BH_BN * 4];
    __shared__ uint32_t thr_s[BH_BM];        // running threshold per row: bits of a non-negative float
    __shared__ float xn2_s[BH_BM];

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1;
    const uint64_t m0 = (uint64_t)blockIdx.x * BH_BM;
    const uint32_t dp = a.dim_p;
    const int ld_r = tid >> 2, ld_ch = tid & 3;
    float4 ra[2], rb[4];
    const uint64_t qleft = (a.m - m0) * dp * 2;
    const __amdgpu_buffer_rsrc_t qres = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<uint16_t *>(a.x16 + m0 * dp), 0, (int)(qleft > 0xFFFFFFFFull ? 0xFFFFFFFFu : (uint32_t)qleft), 0x00020000);
    const uint32_t lane_b = (uint32_t)ld_r * dp * 2 + (uint32_t)ld_ch * 16, r64_b = 64u * dp * 2;
    __shared__ float xs_s[BH_BM];            // |row - mu| (the square roots are taken once per row / centroid, not per pair)
    if (tid < BH_BM) {
        thr_s[tid] = 0x7F800000u;            // +inf
        const float xn = m0 + tid < a.m ? a.xn2[m0 + tid] : 0.0f;
        xn2_s[tid] = xn;
        xs_s[tid] = sqrtf(xn) * 1.000001f;
    }
    const int l31 = lane & 31, lk = lane >> 5;
    int rowa[2], rowb[4], swa[2], swb[4];
#pragma unroll
    for (int t = 0; t < 2; ++t) { rowa[t] = wm * 64 + t * 32 + l31; swa[t] = (rowa[t] >> 2) & 3; }
#pragma unroll
    for (int t = 0; t < 4; ++t) { rowb[t] = wn * 128 + t * 32 + l31; swb[t] = (rowb[t] >> 2) & 3; }
    const uint32_t nk = dp / BH_BK;
    const float inv = 1.52587890625e-05f;    // 2^-16

    for (uint32_t c0 = 0; c0 < a.kc; c0 += BH_BN) {
        const uint64_t vleft = (uint64_t)(a.kc_pad - c0) * dp * 2;
        const __amdgpu_buffer_rsrc_t vres = __builtin_amdgcn_make_buffer_rsrc(
            const_cast<uint16_t *>(a.c16 + (uint64_t)c0 * dp), 0, (int)(vleft > 0xFFFFFFFFull ? 0xFFFFFFFFu : (uint32_t)vleft), 0x00020000);
        auto fetch = [&](uint32_t k0) {
#pragma unroll
            for (int h = 0; h < 2; ++h) ra[h] = buf_ld16(qres, lane_b, k0 * 2 + h * r64_b);
#pragma unroll
            for (int h = 0; h < 4; ++h) rb[h] = buf_ld16(vres, lane_b, k0 * 2 + h * r64_b);
        };
        auto stash = [&](int buf) {
#pragma unroll
            for (int h = 0; h < 2; ++h) { const int r = ld_r + 64 * h; As4[buf][r * 4 + (ld_ch ^ ((r >> 2) & 3))] = ra[h]; }
#pragma unroll
            for (int h = 0; h < 4; ++h) { const int r = ld_r + 64 * h; Bs4[buf][r * 4 + (ld_ch ^ ((r >> 2) & 3))] = rb[h]; }
        };
        f32x16_t acc[2][4];
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.0f;
        fetch(0);
        stash(0);
        __syncthreads();
        for (uint32_t kt = 0; kt < nk; ++kt) {
            const int buf = (int)(kt & 1u);
            if (kt + 1 < nk) fetch((kt + 1) * BH_BK);
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                f16x8_t av[2], bv[4];
#pragma unroll
                for (int t = 0; t < 2; ++t) av[t] = __builtin_bit_cast(f16x8_t, As4[buf][rowa[t] * 4 + ((2 * j + lk) ^ swa[t])]);
#pragma unroll
                for (int t = 0; t < 4; ++t) bv[t] = __builtin_bit_cast(f16x8_t, Bs4[buf][rowb[t] * 4 + ((2 * j + lk) ^ swb[t])]);
#pragma unroll
                for (int i = 0; i < 2; ++i)
#pragma unroll
                    for (int jj = 0; jj < 4; ++jj)
                        acc[i][jj] = __builtin_amdgcn_mfma_f32_32x32x16_f16(av[i], bv[jj], acc[i][jj], 0, 0, 0);
            }
            if (kt + 1 < nk) stash(buf ^ 1);
            __syncthreads();
        }
        // ---- bounds of this centroid tile.  C/D layout: col = lane & 31 (centroid), row = (r & 3) + 8 (r >> 2) + 4 lk
        float cn2[4], cs[4];
        bool cv[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const uint32_t vj = c0 + wn * 128 + j * 32 + l31;
            cv[j] = vj < a.kc;
            cn2[j] = cv[j] ? a.cn2[vj] : 0.0f;
            cs[j] = sqrtf(cn2[j]) * 1.000001f;
        }
        auto bounds = [&](int i, int j, int r, float &lb, float &ub) {
            const int ml = wm * 64 + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * lk;
            const float xn = xn2_s[ml], nn = xn + cn2[j];
            const float qv = xs_s[ml] * cs[j];                      // >= |a| |b|
            const float dt = nn - 2.0f * qv * (acc[i][j][r] * inv);
            const float err = 2.0f * qv * a.eps + 4.0e-6f * nn;
            ub = fmaxf(dt + err, 0.0f) * (1.0f + a.cm) + 1.0e-30f;
            lb = fmaxf(dt - err, 0.0f) * (1.0f - a.cm);
        };
        // (1) the rows' running thresholds: smallest upper bound of this tile, reduced over the 32 centroid lanes
#pragma unroll
        for (int i = 0; i < 2; ++i) {
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                float mn = INFINITY;
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    float lb, ub;
                    bounds(i, j, r, lb, ub);
                    if (cv[j] && ub < mn) mn = ub;                  // (a NaN bound never lowers a threshold)
                }
#pragma unroll
                for (int off = 16; off > 0; off >>= 1) mn = fminf(mn, __shfl_xor(mn, off, 64));
                if (l31 == 0 && mn < INFINITY)
                    atomicMin(&thr_s[wm * 64 + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * lk], __float_as_uint(mn));
            }
        }
        __syncthreads();
        // (2) candidates: every centroid whose lower bound does not exceed its row's threshold (NaN bounds are kept)
#pragma unroll
        for (int i = 0; i < 2; ++i) {
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int ml = wm * 64 + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * lk;
                const float thr = __uint_as_float(thr_s[ml]);
                const uint64_t row = m0 + ml;
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    float lb, ub;
                    bounds(i, j, r, lb, ub);
                    if (cv[j] && row < a.m && !(lb > thr)) {
                        const uint32_t slot = atomicAdd(&a.cand_cnt[row], 1u);
                        if (slot < a.cap) a.cand[row * a.cap + slot] = c0 + wn * 128 + j * 32 + l31;
                    }
                }
            }
        }
        __syncthreads();          // the LDS stages are reused by the next centroid tile
    }
}
hipError_t launch_assign_f16(const AssignF16Args &a, hipStream_t s) {
    if (a.m == 0 || a.kc == 0) return hipSuccess;
    if ((a.dim_p % BH_BK) != 0 || (uint64_t)a.dim_p * 2 * 320 >= 0x7FFFFFFFull || (a.kc_pad % BH_BN) != 0 || a.kc_pad < a.kc)
        return hipErrorInvalidValue;
    const uint64_t blocks = (a.m + BH_BM - 1) / BH_BM;
    if (blocks > 0x7FFFFFFFull) return hipErrorInvalidValue;
    hipLaunchKernelGGL(assign_f16_kernel, dim3((uint32_t)blocks), dim3(256), 0, s, a);
    return hipGetLastError();
}

// exact pass: four lanes per row, each walking candidates 4 t + (lane & 3) of its row in the reference's order
__global__ __launch_bounds__(256) void assign_rescore_kernel(const float *__restrict__ rows, const float *__restrict__ centroids, uint64_t m,
                                                            uint32_t dim, uint32_t kc, const uint32_t *__restrict__ cand,
                                                            const uint32_t *__restrict__ cand_cnt, uint32_t cap, uint32_t *__restrict__ cluster) {
    const int lane = threadIdx.x & 63;
    const uint64_t row = ((uint64_t)blockIdx.x * 4 + (threadIdx.x >> 6)) * 16 + (uint32_t)(lane >> 2);
    const uint32_t cl = (uint32_t)lane & 3u;
    const bool live = row < m;
    uint32_t cnt = live ? cand_cnt[row] : 0u;
    const bool all = cnt > cap;                       // list overflowed: every centroid is a candidate
    if (all) cnt = kc;
    const float *x = rows + (live ? row : 0) * dim;
    const uint32_t G = dim >> 2;
    uint64_t best = KEY_EMPTY;
    uint32_t rounds = (cnt + 3) >> 2;
    // (the loop count differs per lane: no cross-lane operation inside)
    for (uint32_t t = 0; t < rounds; ++t) {
        const uint32_t ci = 4 * t + cl;
        if (ci >= cnt) break;
        const uint32_t j = all ? ci : cand[row * cap + ci];
        const float *c = centroids + (uint64_t)j * dim;
        float sum = 0.0f;
        uint32_t g = 0;
        for (; g + 8 <= G; g += 8) {
            float4 xv[8], cvv[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) { xv[u] = load4<true>(x + (g + u) * 4); cvv[u] = load4<true>(c + (g + u) * 4); }
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const float d0 = xv[u].x - cvv[u].x, d1 = xv[u].y - cvv[u].y, d2 = xv[u].z - cvv[u].z, d3 = xv[u].w - cvv[u].w;
                float tt = d0 * d0 + d1 * d1;
                tt = tt + d2 * d2;
                sum = sum + (tt + d3 * d3);
            }
        }
        for (; g < G; ++g) {
            const float4 xv = load4<true>(x + g * 4), cv = load4<true>(c + g * 4);
            const float d0 = xv.x - cv.x, d1 = xv.y - cv.y, d2 = xv.z - cv.z, d3 = xv.w - cv.w;
            float tt = d0 * d0 + d1 * d1;
            tt = tt + d2 * d2;
            sum = sum + (tt + d3 * d3);
        }
        const uint64_t key = ((uint64_t)__float_as_uint(sum) << 32) | j;
        best = key < best ? key : best;
    }
    // the row's four lanes: smallest (distance bits, index) = strict '<' in ascending centroid order (index.rs:408-415)
#pragma unroll
    for (int off = 1; off < 4; off <<= 1) {
        const uint64_t o = shfl_u64(best, lane ^ off);
        best = o < best ? o : best;
    }
    if (live && cl == 0) cluster[row] = best == KEY_EMPTY ? 0u : (uint32_t)best;
}
// The same result with coalesced reads: a wave still owns 16 rows x 4 candidate slots per round, but the 64 lanes read one
// row (and each of its candidates' centroids) 1 KB at a time -- lane g computes the 4-group term of group g -- and park the
// terms in LDS [group][slot] (stride 65: conflict-free both ways); lane `slot` then adds its 64 terms in ascending group order,
// which is the reference's chain (index.rs:461-480).  A row is read once for its four slots.
__global__ __launch_bounds__(128) void assign_rescore_wave_kernel(const float *__restrict__ rows, const float *__restrict__ centroids, uint64_t m,
                                                                 uint32_t dim, uint32_t kc, const uint32_t *__restrict__ cand,
                                                                 const uint32_t *__restrict__ cand_cnt, uint32_t cap, uint32_t *__restrict__ cluster) {
    __shared__ float lds_all[2][64 * 65];
    const int lane = threadIdx.x & 63;
    const int wv = threadIdx.x >> 6;
    float *lds = lds_all[wv];
    const uint64_t wave_row0 = ((uint64_t)blockIdx.x * 2 + (uint32_t)wv) * 16;
    const uint64_t row = wave_row0 + (uint32_t)(lane >> 2);
    const uint32_t cl = (uint32_t)lane & 3u;
    const bool live = row < m;
    uint32_t cnt = live ? cand_cnt[row] : 0u;
    const bool all = cnt > cap;                       // list overflowed: every centroid is a candidate
    if (all) cnt = kc;
    const uint32_t G = dim >> 2;
    uint64_t best = KEY_EMPTY;
    for (uint32_t t = 0;; ++t) {
        const uint32_t ci = 4 * t + cl;
        const bool act = ci < cnt;
        const uint64_t mask = __ballot(act);
        if (mask == 0) break;
        const uint32_t j = act ? (all ? ci : cand[row * cap + ci]) : 0u;
        float sum = 0.0f;
        for (uint32_t g0 = 0; g0 < G; g0 += 64) {
            const uint32_t ng = (G - g0 < 64u) ? (G - g0) : 64u;
            const bool gv = (uint32_t)lane < ng;
            const uint32_t goff = (g0 + (gv ? (uint32_t)lane : 0u)) * 4;
#pragma unroll 1
            for (int r = 0; r < 16; ++r) {
                const uint32_t m4 = (uint32_t)(mask >> (4 * r)) & 0xFu;     // wave-uniform
                if (m4 == 0) continue;
                const float4 xv = load4<true>(rows + (wave_row0 + (uint32_t)r) * dim + goff);
                float4 cv[4];
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    const uint32_t jp = readlane_u32(j, 4 * r + u);
                    cv[u] = load4<true>(centroids + (uint64_t)jp * dim + goff);
                }
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    const float d0 = xv.x - cv[u].x, d1 = xv.y - cv[u].y, d2 = xv.z - cv[u].z, d3 = xv.w - cv[u].w;
                    float tt = d0 * d0 + d1 * d1;
                    tt = tt + d2 * d2;
                    tt = tt + d3 * d3;
                    if (gv) lds[lane * 65 + 4 * r + u] = tt;
                }
            }
            wave_lds_fence();
            uint32_t e = 0;
            for (; e + 8 <= ng; e += 8) {
                float v[8];
#pragma unroll
                for (int u = 0; u < 8; ++u) v[u] = lds[(e + u) * 65 + lane];
#pragma unroll
                for (int u = 0; u < 8; ++u) sum = sum + v[u];
            }
            for (; e < ng; ++e) sum = sum + lds[e * 65 + lane];
            wave_lds_fence();
        }
        if (act) {
            const uint64_t key = ((uint64_t)__float_as_uint(sum) << 32) | j;
            best = key < best ? key : best;
        }
    }
    // the row's four lanes: smallest (distance bits, index) = strict '<' in ascending centroid order (index.rs:408-415)
#pragma unroll
    for (int off = 1; off < 4; off <<= 1) {
        const uint64_t o = shfl_u64(best, lane ^ off);
        best = o < best ? o : best;
    }
    if (live && cl == 0) cluster[row] = best == KEY_EMPTY ? 0u : (uint32_t)best;
}
hipError_t launch_assign_rescore(const float *rows, const float *centroids, uint64_t m, uint32_t dim, uint32_t kc, const uint32_t *cand,
                                 const uint32_t *cand_cnt, uint32_t cap, uint32_t *cluster, hipStream_t s) {
    if (m == 0) return hipSuccess;
    if ((dim % 4) != 0) return hipErrorInvalidValue;
    // PQV_RESCORE_WAVE=0: the lane-per-(row, candidate) form above (A/B)
    static const bool wave_form = [] { const char *e = getenv("PQV_RESCORE_WAVE"); return !(e && *e == '0'); }();
    if (wave_form && dim >= 64) {
        const uint64_t wblocks = (m + 31) / 32;
        if (wblocks > 0x7FFFFFFFull) return hipErrorInvalidValue;
        hipLaunchKernelGGL(assign_rescore_wave_kernel, dim3((uint32_t)wblocks), dim3(128), 0, s, rows, centroids, m, dim, kc, cand, cand_cnt, cap, cluster);
        return hipGetLastError();
    }
    const uint64_t blocks = (m + 63) / 64;
    if (blocks > 0x7FFFFFFFull) return hipErrorInvalidValue;
    hipLaunchKernelGGL(assign_rescore_kernel, dim3((uint32_t)blocks), dim3(256), 0, s, rows, centroids, m, dim, kc, cand, cand_cnt, cap, cluster);
    return hipGetLastError();
}

// |row - mu|^2 and 1 / |row - mu| per row, and the f16 image of (row - mu) / |row - mu| * 2^8 zero-padded to dim_p:
// one wave per row (mu == nullptr: no centring).  pad_to rows beyond n are written as zero rows (the centroid table is
// padded to a multiple of 256 rows).
__global__ __launch_bounds__(256) void center_normalize_f16_kernel(const float *__restrict__ rows, const float *__restrict__ mu, uint64_t n,
                                                                  uint64_t n_pad, uint32_t dim, uint32_t dim_p, float *__restrict__ out_n2,
                                                                  uint16_t *__restrict__ out) {
    const int lane = threadIdx.x & 63;
    const uint64_t w = (uint64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    const uint64_t nw = (uint64_t)gridDim.x * 4;
    for (uint64_t r = w; r < n_pad; r += nw) {
        if (r >= n) {
            for (uint32_t e = lane; e < dim_p; e += 64) out[r * dim_p + e] = 0;
            continue;
        }
        const float *p = rows + r * dim;
        float acc = 0.0f;
        if ((dim & 7u) == 0 && dim <= 2048) {
            // 8 values (two 16-byte loads) per lane and step, kept in registers between the norm and the scaling pass
            float4 v[4][2];
            const uint32_t G8 = dim >> 3;
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const uint32_t g = (uint32_t)lane + 64u * (uint32_t)u;
                if (g < G8) {
                    const float4 *pp = reinterpret_cast<const float4 *>(p + g * 8);
                    float4 a0 = pp[0], a1 = pp[1];
                    if (mu) {
                        const float4 *mm = reinterpret_cast<const float4 *>(mu + g * 8);
                        const float4 m0 = mm[0], m1 = mm[1];
                        a0.x -= m0.x; a0.y -= m0.y; a0.z -= m0.z; a0.w -= m0.w; a1.x -= m1.x; a1.y -= m1.y; a1.z -= m1.z; a1.w -= m1.w;
                    }
                    v[u][0] = a0; v[u][1] = a1;
                    acc = fmaf(a0.x, a0.x, acc); acc = fmaf(a0.y, a0.y, acc); acc = fmaf(a0.z, a0.z, acc); acc = fmaf(a0.w, a0.w, acc);
                    acc = fmaf(a1.x, a1.x, acc); acc = fmaf(a1.y, a1.y, acc); acc = fmaf(a1.z, a1.z, acc); acc = fmaf(a1.w, a1.w, acc);
                }
            }
#pragma unroll
            for (int off = 32; off > 0; off >>= 1) acc += __shfl_xor(acc, off, 64);
            if (lane == 0) out_n2[r] = acc;
            const float sc = acc > 0.0f ? 256.0f / sqrtf(acc) : 0.0f;
            auto cl = [&](float x) { return fminf(fmaxf(x * sc, -65504.0f), 65504.0f); };
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const uint32_t g = (uint32_t)lane + 64u * (uint32_t)u;
                if (g < (dim_p >> 3)) {
                    f16x8_t h = {0, 0, 0, 0, 0, 0, 0, 0};
                    if (g < G8) {
                        h[0] = (_Float16)cl(v[u][0].x); h[1] = (_Float16)cl(v[u][0].y); h[2] = (_Float16)cl(v[u][0].z); h[3] = (_Float16)cl(v[u][0].w);
                        h[4] = (_Float16)cl(v[u][1].x); h[5] = (_Float16)cl(v[u][1].y); h[6] = (_Float16)cl(v[u][1].z); h[7] = (_Float16)cl(v[u][1].w);
                    }
                    *reinterpret_cast<float4 *>(out + r * dim_p + g * 8) = __builtin_bit_cast(float4, h);
                }
            }
            continue;
        }
        for (uint32_t e = lane; e < dim; e += 64) { const float v = p[e] - (mu ? mu[e] : 0.0f); acc = fmaf(v, v, acc); }
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) acc += __shfl_xor(acc, off, 64);
        if (lane == 0) out_n2[r] = acc;
        const float sc = acc > 0.0f ? 256.0f / sqrtf(acc) : 0.0f;
        for (uint32_t e = lane; e < dim_p; e += 64) {
            float v = e < dim ? (p[e] - (mu ? mu[e] : 0.0f)) * sc : 0.0f;
            v = fminf(fmaxf(v, -65504.0f), 65504.0f);
            const _Float16 h = (_Float16)v;
            out[r * dim_p + e] = __builtin_bit_cast(uint16_t, h);
        }
    }
}
hipError_t launch_center_normalize_f16(const float *rows, const float *mu, uint64_t n, uint64_t n_pad, uint32_t dim, uint32_t dim_p,
                                       float *out_n2, void *out, hipStream_t s) {
    if (n_pad == 0) return hipSuccess;
    uint64_t blocks = (n_pad + 3) / 4;
    if (blocks > 65536) blocks = 65536;
    hipLaunchKernelGGL(center_normalize_f16_kernel, dim3((uint32_t)blocks), dim3(256), 0, s, rows, mu, n, n_pad, dim, dim_p, out_n2,
                       static_cast<uint16_t *>(out));
    return hipGetLastError();
}
// mu[d] = mean over the k rows of m[., d]: one block per 64 columns, four row slices per column reduced through LDS
__global__ __launch_bounds__(256) void col_mean_kernel(const float *__restrict__ m, uint32_t k, uint32_t dim, float *__restrict__ mu) {
    __shared__ float part[4][64];
    const uint32_t d = blockIdx.x * 64u + (threadIdx.x & 63u), sl = threadIdx.x >> 6;
    float acc = 0.0f;
    if (d < dim)
        for (uint32_t r = sl; r < k; r += 4) acc += m[(uint64_t)r * dim + d];
    part[sl][threadIdx.x & 63u] = acc;
    __syncthreads();
    if (sl == 0 && d < dim) {
        const float t = (part[0][threadIdx.x] + part[1][threadIdx.x]) + (part[2][threadIdx.x] + part[3][threadIdx.x]);
        mu[d] = k ? t / (float)k : 0.0f;
    }
}
hipError_t launch_col_mean(const float *m, uint32_t k, uint32_t dim, float *mu, hipStream_t s) {
    if (dim == 0) return hipSuccess;
    hipLaunchKernelGGL(col_mean_kernel, dim3((dim + 63) / 64), dim3(256), 0, s, m, k, dim, mu);
    return hipGetLastError();
}

// one wave per row; f32 partial sums, wave-reduced
__global__ __launch_bounds__(256) void row_norms_kernel(const float *__restrict__ rows, uint64_t n,
                                                       uint32_t dim, int mode, float *__restrict__ out) {
    const int lane = threadIdx.x & 63;
    const uint64_t w = (uint64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    const uint64_t nw = (uint64_t)gridDim.x * 4;
    for (uint64_t r = w; r < n; r += nw) {
        const float *p = rows + r * dim;
        float acc = 0.0f;
        if (mode == 2) {                 // max |x_i| (NaN propagates as NaN-free max; non-finite rows are caught by their norm)
            for (uint32_t e = lane; e < dim; e += 64) acc = fmaxf(acc, fabsf(p[e]));
#pragma unroll
            for (int off = 32; off > 0; off >>= 1) acc = fmaxf(acc, __shfl_xor(acc, off, 64));
            if (lane == 0) out[r] = acc;
            continue;
        }
        for (uint32_t e = lane; e < dim; e += 64) acc = fmaf(p[e], p[e], acc);
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) acc += __shfl_xor(acc, off, 64);
        if (lane == 0) out[r] = mode == 0 ? (acc > 0.0f ? 1.0f / sqrtf(acc) : 0.0f) : acc;
    }
}
hipError_t launch_row_norms(const float *rows, uint64_t n, uint32_t dim, int mode, float *out, hipStream_t s) {
    if (n == 0) return hipSuccess;
    uint64_t blocks = (n + 3) / 4;
    if (blocks > 65536) blocks = 65536;
    hipLaunchKernelGGL(row_norms_kernel, dim3((uint32_t)blocks), dim3(256), 0, s, rows, n, dim, mode, out);
    return hipGetLastError();
}

template <int S>
__global__ __launch_bounds__(64) void brute_select_kernel(unsigned long long *cand, uint32_t *cand_cnt,
                                                         uint32_t cap, uint32_t k,
                                                         unsigned long long *thr, uint32_t *overflow) {
    const int lane = threadIdx.x;
    const uint32_t q = blockIdx.x;
    const uint32_t cnt = cand_cnt[q];
    if (cnt > cap) { if (lane == 0) atomicOr(overflow, 1u); return; }
    unsigned long long *c = cand + (uint64_t)q * cap;
    WaveTopk<S> tk;
    tk.init();
    for (uint32_t i = 0; i < cnt; i += 64) {
        const uint64_t key = (i + lane < cnt) ? c[i + lane] : KEY_EMPTY;
        tk.offer(key, 0u, k, lane);
    }
    uint32_t kept = 0;
#pragma unroll
    for (int s = 0; s < S; ++s) {
        const uint32_t e = s * 64 + lane;
        const bool have = e < k && tk.key[s] != KEY_EMPTY;
        kept += (uint32_t)__popcll(__ballot(have));
        if (e < k) c[e] = tk.key[s];
    }
    if (lane == 0) {
        cand_cnt[q] = kept;
        thr[q] = kept >= k ? tk.kth(k) : KEY_EMPTY;
    }
}
__global__ __launch_bounds__(256) void brute_overflow_kernel(const uint32_t *cand_cnt, uint32_t nq,
                                                            uint32_t cap, uint32_t *overflow) {
    const uint32_t q = blockIdx.x * 256 + threadIdx.x;
    if (q < nq && cand_cnt[q] > cap) atomicOr(overflow, 1u);
}
hipError_t launch_brute_overflow_check(const uint32_t *cand_cnt, uint32_t nq, uint32_t cap, uint32_t *overflow,
                                       hipStream_t s) {
    if (nq == 0) return hipSuccess;
    hipLaunchKernelGGL(brute_overflow_kernel, dim3((nq + 255) / 256), dim3(256), 0, s, cand_cnt, nq, cap, overflow);
    return hipGetLastError();
}

hipError_t launch_brute_select(unsigned long long *cand, uint32_t *cand_cnt, uint32_t cap, uint32_t nq,
                               uint32_t k, unsigned long long *thr, uint32_t *overflow, hipStream_t s) {
    if (nq == 0) return hipSuccess;
    if (k <= 64) hipLaunchKernelGGL(brute_select_kernel<1>, dim3(nq), dim3(64), 0, s, cand, cand_cnt, cap, k, thr, overflow);
    else if (k <= 256) hipLaunchKernelGGL(brute_select_kernel<4>, dim3(nq), dim3(64), 0, s, cand, cand_cnt, cap, k, thr, overflow);
    else if (k <= 1024) hipLaunchKernelGGL(brute_select_kernel<16>, dim3(nq), dim3(64), 0, s, cand, cand_cnt, cap, k, thr, overflow);
    else return hipErrorInvalidValue;
    return hipGetLastError();
}

__global__ __launch_bounds__(256) void brute_finish_kernel(const unsigned long long *cand, const uint32_t *cand_cnt,
                                                          uint32_t cap, uint32_t nq, uint32_t k,
                                                          uint32_t *row_idx, float *dist, uint32_t *n_found) {
    const uint64_t i = (uint64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= (uint64_t)nq * k) return;
    const uint32_t q = (uint32_t)(i / k), e = (uint32_t)(i % k);
    const uint32_t cnt = cand_cnt[q] < k ? cand_cnt[q] : k;
    if (e < cnt) {
        const unsigned long long key = cand[(uint64_t)q * cap + e];
        row_idx[i] = (uint32_t)key;
        dist[i] = unsortable_bits((uint32_t)(key >> 32));
    } else {
        row_idx[i] = 0xFFFFFFFFu;
        dist[i] = INFINITY;
    }
    if (e == 0 && n_found) n_found[q] = cnt;
}
hipError_t launch_brute_finish(const unsigned long long *cand, const uint32_t *cand_cnt, uint32_t cap,
                               uint32_t nq, uint32_t k, uint32_t *row_idx, float *dist, uint32_t *n_found,
                               hipStream_t s) {
    if (nq == 0) return hipSuccess;
    const uint64_t total = (uint64_t)nq * k;
    hipLaunchKernelGGL(brute_finish_kernel, dim3((uint32_t)((total + 255) / 256)), dim3(256), 0, s, cand,
                       cand_cnt, cap, nq, k, row_idx, dist, n_found);
    return hipGetLastError();
}

// ------------------------------------------------------------------------------------
// shard_merge_kernel: the exchange step of the sharded search (one list per GPU/file, merged
// like the reference's single heap over all files, src/df_vector/exec.rs:264-267).
// ------------------------------------------------------------------------------------
template <int S>
__global__ __launch_bounds__(64) void shard_merge_kernel(const float *__restrict__ dist,
                                                        const uint32_t *__restrict__ rows,
                                                        const long long *__restrict__ row_base,
                                                        uint32_t n_shards, uint32_t nq, uint32_t k,
                                                        float *__restrict__ out_dist,
                                                        long long *__restrict__ out_rows, uint32_t stride) {
    const int lane = threadIdx.x;
    const uint32_t q = blockIdx.x;
    WaveTopk<S> tk;
    tk.init();
    const uint32_t total = n_shards * k;
    for (uint32_t i = 0; i < total; i += 64) {
        const uint32_t idx = i + lane;
        uint64_t key = KEY_EMPTY;
        uint32_t val = 0xFFFFFFFFu;
        if (idx < total) {
            const uint32_t sh = idx / k, e = idx % k;
            const uint64_t src = (((uint64_t)sh * nq + q) * k + e) * stride;
            const uint32_t r = rows[src];
            if (r != 0xFFFFFFFFu) {
                key = ((uint64_t)sortable_bits(dist[src]) << 32) | (uint64_t)idx;   // idx = shard * k + position
                val = idx;
            }
        }
        tk.offer(key, val, k, lane);
    }
#pragma unroll
    for (int s = 0; s < S; ++s) {
        const uint32_t e = s * 64 + lane;
        if (e < k) {
            float d = INFINITY;
            long long gr = -1;
            if (tk.key[s] != KEY_EMPTY) {
                const uint32_t idx = tk.val[s];
                const uint64_t src = (((uint64_t)(idx / k) * nq + q) * k + (idx % k)) * stride;
                d = dist[src];
                gr = row_base[idx / k] + (long long)rows[src];
            }
            out_dist[(uint64_t)q * k + e] = d;
            out_rows[(uint64_t)q * k + e] = gr;
        }
    }
}

hipError_t launch_shard_merge(const float *dist, const uint32_t *rows, const long long *row_base,
                              uint32_t n_shards, uint32_t nq, uint32_t k, float *out_dist,
                              long long *out_rows, hipStream_t s, uint32_t stride) {
    if (nq == 0) return hipSuccess;
    if (k <= 64) hipLaunchKernelGGL(shard_merge_kernel<1>, dim3(nq), dim3(64), 0, s, dist, rows, row_base, n_shards, nq, k, out_dist, out_rows, stride);
    else if (k <= 256) hipLaunchKernelGGL(shard_merge_kernel<4>, dim3(nq), dim3(64), 0, s, dist, rows, row_base, n_shards, nq, k, out_dist, out_rows, stride);
    else if (k <= 1024) hipLaunchKernelGGL(shard_merge_kernel<16>, dim3(nq), dim3(64), 0, s, dist, rows, row_base, n_shards, nq, k, out_dist, out_rows, stride);
    else return hipErrorInvalidValue;
    return hipGetLastError();
}

// {f32 distance, u32 row} of every result as ONE 8-byte element: the send buffer of the shard exchange's single all-gather
__global__ __launch_bounds__(256) void pack_pairs_kernel(const float *__restrict__ dist, const uint32_t *__restrict__ rows,
                                                        uint64_t n, uint2 *__restrict__ out) {
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) out[i] = make_uint2(__float_as_uint(dist[i]), rows[i]);
}
hipError_t launch_pack_pairs(const float *dist, const uint32_t *rows, uint64_t n, void *out, hipStream_t s) {
    if (n == 0) return hipSuccess;
    hipLaunchKernelGGL(pack_pairs_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, dist, rows, n, static_cast<uint2 *>(out));
    return hipGetLastError();
}

// ------------------------------------------------------------------------------------
// block_rows_kernel: the MFMA-operand copy of the IVF-ordered lists.  Every list is cut into
// 16-row tiles (the last one zero-padded); tile T stores 16-byte column ch of its row j at float4
// index (T * G + ch) * 16 + j, so a 16x16x4 MFMA operand fetch (16 rows x 4 columns) is one
// contiguous 1 KiB read.  grid = (tiles of the longest list, n_clusters).
// ------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void block_rows_kernel(const float *__restrict__ src, const uint64_t *__restrict__ list_off,
                                                        const uint64_t *__restrict__ blk_off, uint32_t dim,
                                                        float4 *__restrict__ out) {
    const uint32_t c = blockIdx.y;
    const uint64_t lbeg = list_off[c], len = list_off[c + 1] - lbeg;
    const uint64_t ntile = blk_off[c + 1] - blk_off[c];
    const uint32_t G = dim >> 2;
    for (uint64_t tl = blockIdx.x; tl < ntile; tl += gridDim.x) {
        float4 *dst = out + (blk_off[c] + tl) * G * 16;
        for (uint32_t e = threadIdx.x; e < G * 16; e += 256) {
            const uint32_t ch = e >> 4, j = e & 15;
            const uint64_t p = tl * 16 + j;
            float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
            if (p < len) v = *reinterpret_cast<const float4 *>(src + (lbeg + p) * dim + ch * 4);
            dst[e] = v;
        }
    }
}
hipError_t launch_block_rows(const float *src, const uint64_t *list_off, const uint64_t *blk_off, uint32_t n_clusters,
                             uint64_t max_tiles, uint32_t dim, void *out, hipStream_t s) {
    if (n_clusters == 0 || max_tiles == 0) return hipSuccess;
    const uint32_t gx = (uint32_t)(max_tiles < 4096 ? max_tiles : 4096);
    hipLaunchKernelGGL(block_rows_kernel, dim3(gx, n_clusters), dim3(256), 0, s, src, list_off, blk_off, dim,
                       static_cast<float4 *>(out));
    return hipGetLastError();
}

// ------------------------------------------------------------------------------------
// pack_queries_kernel: the blocked (MFMA A-operand) copy of every quad's queries, for rows too long
// for LDS staging: q_blk[((quad * NG + g) * G + ch) * 16 + i] = 16-byte column ch of query 16 g + i
// of the quad (queries past the quad's count alias its last one; they are masked by the kernels).
// grid = max_quads blocks.
// ------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void pack_queries_kernel(const float *__restrict__ queries, const uint32_t *__restrict__ pairs,
                                                          const uint4 *__restrict__ quads, const uint32_t *__restrict__ n_quads,
                                                          uint32_t nprobe, uint32_t dim, uint32_t ngrp,
                                                          float4 *__restrict__ q_blk) {
    if (blockIdx.x >= *n_quads) return;
    const uint4 quad = quads[blockIdx.x];
    const uint32_t p0 = quad.y, cnt = quad.z, G = dim >> 2;
    float4 *dst = q_blk + (uint64_t)blockIdx.x * ngrp * G * 16;
    const uint32_t total = ngrp * G * 16;
    for (uint32_t e = threadIdx.x; e < total; e += 256) {
        const uint32_t i = e & 15u, ch = (e >> 4) % G, g = (e >> 4) / G;
        const uint32_t q = 16 * g + i;
        const uint32_t qrow = pairs[p0 + (q < cnt ? q : cnt - 1)] / nprobe;
        dst[e] = *reinterpret_cast<const float4 *>(queries + (uint64_t)qrow * dim + ch * 4);
    }
}
hipError_t launch_pack_queries(const float *queries, const uint32_t *pairs, const uint4 *quads, const uint32_t *n_quads,
                               uint32_t max_quads, uint32_t nprobe, uint32_t dim, uint32_t ngrp, void *q_blk, hipStream_t s) {
    if (max_quads == 0) return hipSuccess;
    hipLaunchKernelGGL(pack_queries_kernel, dim3(max_quads), dim3(256), 0, s, queries, pairs, quads, n_quads, nprobe, dim, ngrp,
                       static_cast<float4 *>(q_blk));
    return hipGetLastError();
}

// ------------------------------------------------------------------------------------
// block_rows_f16_kernel: the f16 form of the blocked operand copy.  Values are multiplied by `scale` (a
// power of two chosen so that the corpus maximum lands below 2^14: exact, no overflow) and rounded to
// nearest f16; a 16-byte column holds 8 consecutive dims, tile T stores column cc of its row j at 16-byte
// index (T * dim/8 + cc) * 16 + j -- a 16x16x32 MFMA operand fetch (16 rows x 4 columns) is one 1 KiB read.
// maxabs_kernel: *out = max(*out, bits(max |v|)) (non-negative floats order like their bit patterns).
// ------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void block_rows_f16_kernel(const float *__restrict__ src, const uint64_t *__restrict__ list_off,
                                                            const uint64_t *__restrict__ blk_off, uint32_t dim, float scale,
                                                            float4 *__restrict__ out) {
    const uint32_t c = blockIdx.y;
    const uint64_t lbeg = list_off[c], len = list_off[c + 1] - lbeg;
    const uint64_t ntile = blk_off[c + 1] - blk_off[c];
    const uint32_t G = dim >> 3;
    for (uint64_t tl = blockIdx.x; tl < ntile; tl += gridDim.x) {
        float4 *dst = out + (blk_off[c] + tl) * G * 16;
        for (uint32_t e = threadIdx.x; e < G * 16; e += 256) {
            const uint32_t cc = e >> 4, j = e & 15;
            const uint64_t p = tl * 16 + j;
            float4 lo = make_float4(0.f, 0.f, 0.f, 0.f), hi = lo;
            if (p < len) {
                const float4 *r = reinterpret_cast<const float4 *>(src + (lbeg + p) * dim + cc * 8);
                lo = r[0]; hi = r[1];
            }
            dst[e] = pack_f16x8(lo, hi, scale);
        }
    }
}
hipError_t launch_block_rows_f16(const float *src, const uint64_t *list_off, const uint64_t *blk_off, uint32_t n_clusters,
                                 uint64_t max_tiles, uint32_t dim, float scale, void *out, hipStream_t s) {
    if (n_clusters == 0 || max_tiles == 0) return hipSuccess;
    if (dim % 8) return hipErrorInvalidValue;
    const uint32_t gx = (uint32_t)(max_tiles < 4096 ? max_tiles : 4096);
    hipLaunchKernelGGL(block_rows_f16_kernel, dim3(gx, n_clusters), dim3(256), 0, s, src, list_off, blk_off, dim, scale,
                       static_cast<float4 *>(out));
    return hipGetLastError();
}
__global__ __launch_bounds__(256) void maxabs_kernel(const float *v, uint64_t n, uint32_t *out) {
    uint32_t m = 0;
    for (uint64_t i = (uint64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (uint64_t)gridDim.x * 256) {
        const uint32_t b = __float_as_uint(v[i]) & 0x7FFFFFFFu;
        m = b > m ? b : m;
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) { const uint32_t o = (uint32_t)__shfl_down((int)m, off, 64); m = o > m ? o : m; }
    __shared__ uint32_t wm[4];                       // one atomic per block, not per wave: a single address
    if ((threadIdx.x & 63) == 0) wm[threadIdx.x >> 6] = m;
    __syncthreads();
    if (threadIdx.x == 0) {
        const uint32_t a01 = wm[0] > wm[1] ? wm[0] : wm[1], a23 = wm[2] > wm[3] ? wm[2] : wm[3];
        const uint32_t mm = a01 > a23 ? a01 : a23;
        if (mm) atomicMax(out, mm);
    }
}
hipError_t launch_maxabs(const float *v, uint64_t n, uint32_t *out, hipStream_t s) {
    if (n == 0) return hipSuccess;
    const uint64_t blocks = (n + 255) / 256;
    hipLaunchKernelGGL(maxabs_kernel, dim3((uint32_t)(blocks < 2048 ? blocks : 2048)), dim3(256), 0, s, v, n, out);
    return hipGetLastError();
}

// ------------------------------------------------------------------------------------
// int8 form of the blocked operand copy (wide_filter_kernel<.., OP_I8>).
//   col_minmax_kernel      per-dimension minimum and maximum of the stored rows, as order-preserving uint keys
//                          (sortable_bits) so plain atomicMin / atomicMax work for any sign
//   block_rows_i8_kernel   xi = clamp(rint((x - c) S), -127, 127) with c the per-dimension mid-range and S one global
//                          scale; a 16-byte column holds 16 consecutive dims, tile T stores column cc of its row j at
//                          16-byte index (T * dim/16 + cc) * 16 + j -- a 16x16x64 MFMA operand fetch (16 rows x 4
//                          columns) is one 1 KiB read.  Per row also Nx = |xi|^2 (exact) and an UPPER bound of the
//                          residual norm |x - c - xi / S| (f32 sum + 0.1 % + the rounding of the residuals themselves).
//   quantize_queries_i8_kernel   the same image of every query of a batch (row-major [nq, dim] int8), |qi|^2 and the
//                          residual bound; a query with a non-finite component gets +inf (never skipped).
// ------------------------------------------------------------------------------------
// per-list, per-dimension minimum / maximum of the stored rows: grid (row chunks, lists), a thread per dimension
__global__ __launch_bounds__(256) void list_minmax_kernel(const float *__restrict__ rows, const uint64_t *__restrict__ list_off,
                                                         uint32_t dim, uint32_t chunk_rows, uint32_t *__restrict__ kmin,
                                                         uint32_t *__restrict__ kmax) {
    const uint32_t c = blockIdx.y;
    const uint64_t lbeg = list_off[c], lend = list_off[c + 1];
    for (uint64_t r0 = lbeg + (uint64_t)blockIdx.x * chunk_rows; r0 < lend; r0 += (uint64_t)gridDim.x * chunk_rows) {
        const uint64_t r1 = r0 + chunk_rows < lend ? r0 + chunk_rows : lend;
        for (uint32_t d = threadIdx.x; d < dim; d += 256) {
            uint32_t lo = 0xFFFFFFFFu, hi = 0u;
            for (uint64_t r = r0; r < r1; ++r) {
                const uint32_t kb = sortable_bits(rows[r * dim + d]);
                lo = kb < lo ? kb : lo;
                hi = kb > hi ? kb : hi;
            }
            atomicMin(&kmin[(uint64_t)c * dim + d], lo);
            atomicMax(&kmax[(uint64_t)c * dim + d], hi);
        }
    }
}
hipError_t launch_list_minmax(const float *rows, const uint64_t *list_off, uint32_t n_clusters, uint64_t max_list_len, uint32_t dim,
                              uint32_t *kmin, uint32_t *kmax, hipStream_t s) {
    if (n_clusters == 0 || max_list_len == 0) return hipSuccess;
    const uint32_t chunk = 1024;
    const uint64_t gx = (max_list_len + chunk - 1) / chunk;
    hipLaunchKernelGGL(list_minmax_kernel, dim3((uint32_t)(gx < 64 ? gx : 64), n_clusters), dim3(256), 0, s, rows, list_off, dim, chunk, kmin, kmax);
    return hipGetLastError();
}
// centre[c][d] = (min + max) / 2 of list c, half[c] = its largest |x - centre| component, scale[c] = 127 / half (a list
// whose rows all equal the centre, or an empty one, gets scale 1; the scale is capped so that its square stays finite);
// radius[c] = 0 (block_rows_i8_kernel raises it).  One block per list.
__global__ __launch_bounds__(256) void list_center_kernel(const uint32_t *__restrict__ kmin, const uint32_t *__restrict__ kmax, uint32_t dim,
                                                         const uint64_t *__restrict__ list_off, float *__restrict__ center,
                                                         float *__restrict__ half, float *__restrict__ scale, float *__restrict__ radius) {
    __shared__ float wm[4];
    const uint32_t c = blockIdx.x;
    const bool empty = list_off[c + 1] == list_off[c];
    float h = 0.0f;
    for (uint32_t d = threadIdx.x; d < dim; d += 256) {
        float ctr = 0.0f;
        if (!empty) {
            const float lo = unsortable_bits(kmin[(uint64_t)c * dim + d]), hi = unsortable_bits(kmax[(uint64_t)c * dim + d]);
            ctr = 0.5f * lo + 0.5f * hi;
            h = fmaxf(h, fabsf(fmaxf(hi - ctr, ctr - lo)));
        }
        center[(uint64_t)c * dim + d] = ctr;
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) h = fmaxf(h, __shfl_xor(h, off, 64));
    if ((threadIdx.x & 63) == 0) wm[threadIdx.x >> 6] = h;
    __syncthreads();
    if (threadIdx.x == 0) {
        h = fmaxf(fmaxf(wm[0], wm[1]), fmaxf(wm[2], wm[3]));
        half[c] = h;
        float sc = h > 0.0f ? 127.0f / (h * 1.000001f) : 1.0f;
        if (!(sc < 1.0e15f)) sc = 1.0e15f;
        if (!(sc > 1.0e-30f)) sc = 1.0e-30f;
        scale[c] = sc;
        radius[c] = 0.0f;
    }
}
hipError_t launch_list_center(const uint32_t *kmin, const uint32_t *kmax, uint32_t n_clusters, uint32_t dim, const uint64_t *list_off,
                              float *center, float *half, float *scale, float *radius, hipStream_t s) {
    if (n_clusters == 0) return hipSuccess;
    hipLaunchKernelGGL(list_center_kernel, dim3(n_clusters), dim3(256), 0, s, kmin, kmax, dim, list_off, center, half, scale, radius);
    return hipGetLastError();
}

// The one-centre form (round 2's): per-dimension min / max over ALL lists -> centre and scale in entry 0 of scratch
// tables (global_center_kernel), and, if that form is chosen, every list's entry overwritten with them
// (broadcast_center_kernel) -- block_rows_i8_kernel and the screen kernels then need no second code path.
__global__ __launch_bounds__(256) void global_center_kernel(const uint32_t *__restrict__ kmin, const uint32_t *__restrict__ kmax,
                                                           uint32_t n_clusters, uint32_t dim, const uint64_t *__restrict__ list_off,
                                                           float *__restrict__ g_center, float *__restrict__ g_half_scale) {
    __shared__ float wm[4];
    float h = 0.0f;
    for (uint32_t d = threadIdx.x; d < dim; d += 256) {
        uint32_t lo = 0xFFFFFFFFu, hi = 0u;
        for (uint32_t c = 0; c < n_clusters; ++c) {
            if (list_off[c + 1] == list_off[c]) continue;
            const uint32_t a = kmin[(uint64_t)c * dim + d], b = kmax[(uint64_t)c * dim + d];
            lo = a < lo ? a : lo; hi = b > hi ? b : hi;
        }
        float ctr = 0.0f;
        if (lo <= hi) {
            const float flo = unsortable_bits(lo), fhi = unsortable_bits(hi);
            ctr = 0.5f * flo + 0.5f * fhi;
            h = fmaxf(h, fabsf(fmaxf(fhi - ctr, ctr - flo)));
        }
        g_center[d] = ctr;
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) h = fmaxf(h, __shfl_xor(h, off, 64));
    if ((threadIdx.x & 63) == 0) wm[threadIdx.x >> 6] = h;
    __syncthreads();
    if (threadIdx.x == 0) {
        h = fmaxf(fmaxf(wm[0], wm[1]), fmaxf(wm[2], wm[3]));
        float sc = h > 0.0f ? 127.0f / (h * 1.000001f) : 1.0f;
        if (!(sc < 1.0e15f)) sc = 1.0e15f;
        if (!(sc > 1.0e-30f)) sc = 1.0e-30f;
        g_half_scale[0] = h; g_half_scale[1] = sc;
    }
}
__global__ __launch_bounds__(256) void broadcast_center_kernel(const float *__restrict__ g_center, const float *__restrict__ g_half_scale,
                                                              uint32_t dim, float *__restrict__ center, float *__restrict__ half,
                                                              float *__restrict__ scale) {
    const uint32_t c = blockIdx.x;
    for (uint32_t d = threadIdx.x; d < dim; d += 256) center[(uint64_t)c * dim + d] = g_center[d];
    if (threadIdx.x == 0) { half[c] = g_half_scale[0]; scale[c] = g_half_scale[1]; }
}
hipError_t launch_global_center(const uint32_t *kmin, const uint32_t *kmax, uint32_t n_clusters, uint32_t dim, const uint64_t *list_off,
                                float *g_center, float *g_half_scale, hipStream_t s) {
    hipLaunchKernelGGL(global_center_kernel, dim3(1), dim3(256), 0, s, kmin, kmax, n_clusters, dim, list_off, g_center, g_half_scale);
    return hipGetLastError();
}
hipError_t launch_broadcast_center(const float *g_center, const float *g_half_scale, uint32_t n_clusters, uint32_t dim, float *center,
                                   float *half, float *scale, hipStream_t s) {
    if (n_clusters == 0) return hipSuccess;
    hipLaunchKernelGGL(broadcast_center_kernel, dim3(n_clusters), dim3(256), 0, s, g_center, g_half_scale, dim, center, half, scale);
    return hipGetLastError();
}

// xi = clamp(rint((x - centre_c) S_c), -127, 127) for the rows of list c -- the RESIDUAL against the list's own
// per-dimension mid-range centre at the list's own scale (the IVF residual: a cluster of tight rows gets a fine grid,
// wherever in space it sits).  Per row: Nx = |xi|^2 (exact), rx >= |x - centre - xi / S| (f32 sum + 0.1 % + the roundings
// of the residuals themselves); per list: radius >= |x - centre| of every row (atomic max of non-negative float bits).
__global__ __launch_bounds__(256) void block_rows_i8_kernel(const float *__restrict__ src, const uint64_t *__restrict__ list_off,
                                                           const uint64_t *__restrict__ blk_off, uint32_t dim,
                                                           const float *__restrict__ center, const float *__restrict__ list_scale,
                                                           const float *__restrict__ list_half, float *__restrict__ list_radius,
                                                           uint4 *__restrict__ out, int *__restrict__ row_n2i, float *__restrict__ row_res) {
    __shared__ int s_n2[16][17];
    __shared__ float s_e2[16][17], s_v2[16][17];
    const uint32_t c = blockIdx.y;
    const uint64_t lbeg = list_off[c], len = list_off[c + 1] - lbeg;
    const uint64_t ntile = blk_off[c + 1] - blk_off[c];
    const uint32_t G = dim >> 4;
    const uint32_t j = threadIdx.x & 15, cg = threadIdx.x >> 4;
    const float scale = list_scale[c], maxabs = list_half[c];
    const float inv = 1.0f / scale;
    const float *ctr = center + (uint64_t)c * dim;
    float rad = 0.0f;
    for (uint64_t tl = blockIdx.x; tl < ntile; tl += gridDim.x) {
        uint4 *dst = out + (blk_off[c] + tl) * G * 16;
        const uint64_t p = tl * 16 + j;
        int n2 = 0;
        float e2 = 0.0f, v2 = 0.0f;
        for (uint32_t cc = cg; cc < G; cc += 16) {
            uint32_t w[4] = {0u, 0u, 0u, 0u};
            if (p < len) {
                const float4 *r = reinterpret_cast<const float4 *>(src + (lbeg + p) * dim + cc * 16);
                const float4 *cv = reinterpret_cast<const float4 *>(ctr + cc * 16);
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    const float4 x = r[u], cx = cv[u];
                    const float t[4] = {x.x - cx.x, x.y - cx.y, x.z - cx.z, x.w - cx.w};
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        const int q = quant_i8(t[e], scale);
                        const float res = t[e] - (float)q * inv;
                        n2 += q * q;
                        e2 = fmaf(res, res, e2);
                        v2 = fmaf(t[e], t[e], v2);
                        w[u] |= (uint32_t)(q & 0xFF) << (8 * e);
                    }
                }
            }
            dst[cc * 16 + j] = make_uint4(w[0], w[1], w[2], w[3]);
        }
        s_n2[j][cg] = n2; s_e2[j][cg] = e2; s_v2[j][cg] = v2;
        __syncthreads();
        if (threadIdx.x < 16 && tl * 16 + threadIdx.x < len) {
            int tn = 0; float te = 0.0f, tv = 0.0f;
#pragma unroll
            for (int g = 0; g < 16; ++g) { tn += s_n2[threadIdx.x][g]; te += s_e2[threadIdx.x][g]; tv += s_v2[threadIdx.x][g]; }
            row_n2i[lbeg + tl * 16 + threadIdx.x] = tn;
            // upper bound: the f32 sum (+ 0.1 %), plus the roundings of (x - c) and q / S in every residual
            const float pad = 4.0f * 5.9604645e-08f * sqrtf((float)dim) * (maxabs + 127.0f * inv);
            row_res[lbeg + tl * 16 + threadIdx.x] = sqrtf(te) * 1.001f + pad;
            rad = fmaxf(rad, sqrtf(tv) * 1.001f + pad);
        }
        __syncthreads();
    }
    if (threadIdx.x < 16) {
#pragma unroll
        for (int off = 8; off > 0; off >>= 1) rad = fmaxf(rad, __shfl_xor(rad, off, 64));
        if (threadIdx.x == 0 && rad > 0.0f) atomicMax(reinterpret_cast<uint32_t *>(list_radius + c), __float_as_uint(rad));
    }
}
hipError_t launch_block_rows_i8(const float *src, const uint64_t *list_off, const uint64_t *blk_off, uint32_t n_clusters,
                                uint64_t max_tiles, uint32_t dim, const float *center, const float *list_scale, const float *list_half,
                                float *list_radius, void *out, int *row_n2i, float *row_res, hipStream_t s) {
    if (n_clusters == 0 || max_tiles == 0) return hipSuccess;
    if (dim % 16) return hipErrorInvalidValue;
    const uint32_t gx = (uint32_t)(max_tiles < 64 ? max_tiles : 64);
    hipLaunchKernelGGL(block_rows_i8_kernel, dim3(gx, n_clusters), dim3(256), 0, s, src, list_off, blk_off, dim, center, list_scale,
                       list_half, list_radius, static_cast<uint4 *>(out), row_n2i, row_res);
    return hipGetLastError();
}

// ------------------------------------------------------------------------------------
// Helpers of the MFMA-screened assignment (api.cpp: assign_screened): the k-means assignment of a
// chunk of rows is the top-1 search of every row among the centroids, i.e. the wide screened path
// with the rows as queries and ONE list holding all centroids.
//   assign_setup_kernel : identity bucketing (pair i = query i, quads of `width` consecutive queries of
//                         cluster 0), candidate bases 0, thresholds EMPTY
//   nonfinite_flag_kernel / count_changed_kernel : see the launchers' comments in kernels.h
// ------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void assign_setup_kernel(uint32_t *pairs, uint4 *quads, uint32_t *n_quads, uint64_t *cand_base,
                                                          unsigned long long *gthr, uint32_t nq, uint32_t width) {
    const uint32_t i = blockIdx.x * 256 + threadIdx.x;
    if (i == 0) *n_quads = (nq + width - 1) / width;
    if (i >= nq) return;
    pairs[i] = i;
    cand_base[i] = 0;
    gthr[i] = ~0ull;
    if (i % width == 0) quads[i / width] = make_uint4(0u, i, nq - i < width ? nq - i : width, 0u);
}
hipError_t launch_assign_setup(uint32_t *pairs, uint4 *quads, uint32_t *n_quads, uint64_t *cand_base, unsigned long long *gthr,
                               uint32_t nq, uint32_t width, hipStream_t s) {
    if (nq == 0) return hipSuccess;
    hipLaunchKernelGGL(assign_setup_kernel, dim3((nq + 255) / 256), dim3(256), 0, s, pairs, quads, n_quads, cand_base, gthr, nq, width);
    return hipGetLastError();
}
__global__ __launch_bounds__(256) void nonfinite_flag_kernel(const float *v, uint64_t n, uint32_t *flag) {
    bool bad = false;
    for (uint64_t i = (uint64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (uint64_t)gridDim.x * 256) bad |= !(fabsf(v[i]) < INFINITY);
    if (__ballot(bad) != 0ull && (threadIdx.x & 63) == 0) atomicOr(flag, 1u);
}
hipError_t launch_nonfinite_flag(const float *v, uint64_t n, uint32_t *flag, hipStream_t s) {
    if (n == 0) return hipSuccess;
    const uint64_t blocks = (n + 255) / 256;
    hipLaunchKernelGGL(nonfinite_flag_kernel, dim3((uint32_t)(blocks < 2048 ? blocks : 2048)), dim3(256), 0, s, v, n, flag);
    return hipGetLastError();
}
__global__ __launch_bounds__(256) void count_changed_kernel(const uint32_t *cur, const uint32_t *prev, uint64_t n,
                                                           unsigned long long *changed) {
    uint32_t c = 0;
    for (uint64_t i = (uint64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (uint64_t)gridDim.x * 256) c += cur[i] != prev[i];
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) c += (uint32_t)__shfl_down((int)c, off, 64);
    if ((threadIdx.x & 63) == 0 && c) atomicAdd(changed, (unsigned long long)c);
}
hipError_t launch_count_changed(const uint32_t *cur, const uint32_t *prev, uint64_t n, unsigned long long *changed, hipStream_t s) {
    if (n == 0) return hipSuccess;
    const uint64_t blocks = (n + 255) / 256;
    hipLaunchKernelGGL(count_changed_kernel, dim3((uint32_t)(blocks < 2048 ? blocks : 2048)), dim3(256), 0, s, cur, prev, n, changed);
    return hipGetLastError();
}

// ------------------------------------------------------------------------------------
// pqv_rerank's device-side state handling (update_topk_heap, src/df_vector/exec.rs:457-484): the running top-k of a
// query is (rows, d2, count), sorted by (d2, arrival).  rerank_state_in turns it into partial list 0 of a merge --
// keys (d2 bits, position 0..count-1: earlier arrivals win ties), values tagged with bit 31 -- and rerank_state_out
// maps the merged values back to payloads: a tagged value is an old entry, anything else a position in this batch.
// One wave each; k <= 1024.
// ------------------------------------------------------------------------------------
__global__ __launch_bounds__(64) void rerank_state_in_kernel(const uint32_t *io_rows, const float *io_d2, const uint32_t *io_count,
                                                            uint32_t k, uint32_t k_list, uint64_t *keys, uint32_t *vals, uint32_t *rows_saved) {
    const uint32_t cnt = *io_count < k ? *io_count : k;      // (a count beyond the state's capacity is clamped)
    for (uint32_t i = threadIdx.x; i < k_list; i += 64) {
        if (i < cnt) {
            keys[i] = ((uint64_t)__float_as_uint(io_d2[i]) << 32) | i;
            vals[i] = 0x80000000u | i;
            rows_saved[i] = io_rows[i];
        } else {
            keys[i] = KEY_EMPTY;
            vals[i] = 0xFFFFFFFFu;
        }
    }
}
__global__ __launch_bounds__(64) void rerank_state_out_kernel(const uint32_t *m_vals, const float *m_d2, const uint32_t *m_found,
                                                             const uint32_t *rows_saved, const uint32_t *ids, uint32_t k,
                                                             uint32_t *io_rows, float *io_d2, uint32_t *io_count,
                                                             const uint32_t *m_tie, uint32_t *io_tie) {
    const uint32_t nf = *m_found < k ? *m_found : k;
    for (uint32_t i = threadIdx.x; i < nf; i += 64) {
        const uint32_t v = m_vals[i];
        io_rows[i] = (v & 0x80000000u) ? rows_saved[v & 0x7FFFFFFFu] : (ids ? ids[v] : v);
        io_d2[i] = m_d2[i];
    }
    if (threadIdx.x == 0) {
        *io_count = nf;
        if (io_tie && m_tie && *m_tie) *io_tie = 1u;          // sticky across batches
    }
}
hipError_t launch_rerank_state_in(const uint32_t *io_rows, const float *io_d2, const uint32_t *io_count, uint32_t k, uint32_t k_list,
                                  uint64_t *keys, uint32_t *vals, uint32_t *rows_saved, hipStream_t s) {
    hipLaunchKernelGGL(rerank_state_in_kernel, dim3(1), dim3(64), 0, s, io_rows, io_d2, io_count, k, k_list, keys, vals, rows_saved);
    return hipGetLastError();
}
hipError_t launch_rerank_state_out(const uint32_t *m_vals, const float *m_d2, const uint32_t *m_found, const uint32_t *rows_saved,
                                   const uint32_t *ids, uint32_t k, uint32_t *io_rows, float *io_d2, uint32_t *io_count,
                                   const uint32_t *m_tie, uint32_t *io_tie, hipStream_t s) {
    hipLaunchKernelGGL(rerank_state_out_kernel, dim3(1), dim3(64), 0, s, m_vals, m_d2, m_found, rows_saved, ids, k, io_rows, io_d2, io_count, m_tie, io_tie);
    return hipGetLastError();
}

// ------------------------------------------------------------------------------------
// fill_ones2_kernel: two buffers set to all-ones bytes in ONE launch (the EMPTY preset of the
// partial-list keys and values; hipMemsetAsync costs 2-3 launches per buffer).  16 B per lane.
// ------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void fill_ones2_kernel(uint4 *a, uint64_t na16, uint4 *b, uint64_t nb16) {
    const uint4 ones = make_uint4(~0u, ~0u, ~0u, ~0u);
    const uint64_t stride = (uint64_t)gridDim.x * 256;
    for (uint64_t i = (uint64_t)blockIdx.x * 256 + threadIdx.x; i < na16 + nb16; i += stride) {
        if (i < na16) a[i] = ones; else b[i - na16] = ones;
    }
}
hipError_t launch_fill_ones2(void *a, uint64_t a_bytes, void *b, uint64_t b_bytes, hipStream_t s) {
    if ((a_bytes | b_bytes) & 15u) return hipErrorInvalidValue;
    const uint64_t n16 = (a_bytes + b_bytes) / 16;
    if (n16 == 0) return hipSuccess;
    const uint64_t blocks = (n16 + 255) / 256;
    hipLaunchKernelGGL(fill_ones2_kernel, dim3((uint32_t)(blocks < 4096 ? blocks : 4096)), dim3(256), 0, s,
                       static_cast<uint4 *>(a), a_bytes / 16, static_cast<uint4 *>(b), b_bytes / 16);
    return hipGetLastError();
}

// ------------------------------------------------------------------------------------
// pad_rows: out[i, 0 .. dim_p) = {src[idx ? idx[i] : i, 0 .. dim), 0 ...} for dim % 4 == 0: rows (and per batch the queries)
// of a dimension the MFMA screen has no tiling for are stored zero-padded to one it has.  The reference's distance
// takes 4 elements per step (index.rs:461-473), so a padded group adds ((0 + 0) + 0) + 0 = +0.0 to a non-negative sum:
// every distance over the padded rows is bit-identical to the one over the originals.
// ------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void pad_rows_kernel(const float *__restrict__ src, const uint32_t *__restrict__ idx32, uint64_t m,
                                                      uint32_t dim, uint32_t dim_p, float *__restrict__ out) {
    const int lane = threadIdx.x & 63;
    const uint64_t wave = (uint64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    const uint64_t nwaves = (uint64_t)gridDim.x * 4;
    const uint32_t G = dim >> 2, Gp = dim_p >> 2;
    for (uint64_t i = wave; i < m; i += nwaves) {
        const uint64_t r = idx32 ? (uint64_t)idx32[i] : i;
        const float4 *s = reinterpret_cast<const float4 *>(src + r * dim);
        float4 *d = reinterpret_cast<float4 *>(out + i * dim_p);
        for (uint32_t g = lane; g < Gp; g += 64) d[g] = g < G ? s[g] : make_float4(0.f, 0.f, 0.f, 0.f);
    }
}
hipError_t launch_pad_rows(const float *src, const uint32_t *idx32, uint64_t m, uint32_t dim, uint32_t dim_p, float *out, hipStream_t s) {
    if (m == 0) return hipSuccess;
    if ((dim % 4) != 0 || (dim_p % 4) != 0 || dim_p < dim) return hipErrorInvalidValue;
    uint64_t blocks = (m + 3) / 4;
    if (blocks > 65536) blocks = 65536;
    hipLaunchKernelGGL(pad_rows_kernel, dim3((uint32_t)blocks), dim3(256), 0, s, src, idx32, m, dim, dim_p, out);
    return hipGetLastError();
}

// ------------------------------------------------------------------------------------
// gather_rows: out[i,:] = src[idx[i],:]; one wave per output row, 16 B per lane.
// ------------------------------------------------------------------------------------
template <bool ALIGNED>
__global__ __launch_bounds__(256) void gather_rows_kernel(const float *__restrict__ src,
                                                         const uint32_t *__restrict__ idx32,
                                                         const uint64_t *__restrict__ idx64,
                                                         uint64_t m, uint32_t dim,
                                                         float *__restrict__ out) {
    const int lane = threadIdx.x & 63;
    const uint64_t wave = (uint64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    const uint64_t nwaves = (uint64_t)gridDim.x * 4;
    for (uint64_t i = wave; i < m; i += nwaves) {
        const uint64_t r = idx64 ? idx64[i] : (uint64_t)idx32[i];
        const float *s = src + r * dim;
        float *d = out + i * dim;
        if constexpr (ALIGNED) {
            const uint32_t G = dim >> 2;
            for (uint32_t g = lane; g < G; g += 64)
                reinterpret_cast<float4 *>(d)[g] = reinterpret_cast<const float4 *>(s)[g];
        } else {
            for (uint32_t e = lane; e < dim; e += 64) d[e] = s[e];
        }
    }
}

hipError_t launch_gather_rows(const float *src, const uint32_t *idx32, const uint64_t *idx64,
                              uint64_t m, uint32_t dim, float *out, hipStream_t s) {
    if (m == 0) return hipSuccess;
    uint64_t blocks = (m + 3) / 4;
    if (blocks > 65536) blocks = 65536;
    if (dim % 4 == 0)
        hipLaunchKernelGGL(gather_rows_kernel<true>, dim3((uint32_t)blocks), dim3(256), 0, s, src,
                           idx32, idx64, m, dim, out);
    else
        hipLaunchKernelGGL(gather_rows_kernel<false>, dim3((uint32_t)blocks), dim3(256), 0, s, src,
                           idx32, idx64, m, dim, out);
    return hipGetLastError();
}

__global__ __launch_bounds__(256) void narrow_f64_kernel(const double *__restrict__ src,
                                                        uint64_t count, float *__restrict__ out) {
    const uint64_t stride = (uint64_t)gridDim.x * 256;
    for (uint64_t i = (uint64_t)blockIdx.x * 256 + threadIdx.x; i < count; i += stride)
        out[i] = (float)src[i];   // `as f32`: round to nearest even (parquet.rs:253)
}
hipError_t launch_narrow_f64(const double *src, uint64_t count, float *out, hipStream_t s) {
    if (count == 0) return hipSuccess;
    uint64_t blocks = (count + 255) / 256;
    if (blocks > 65536) blocks = 65536;
    hipLaunchKernelGGL(narrow_f64_kernel, dim3((uint32_t)blocks), dim3(256), 0, s, src, count, out);
    return hipGetLastError();
}

// ------------------------------------------------------------------------------------
// assign_kernel: Lloyd assign + final assignment (index.rs:395-424, :189-201, :244-257).
// Same skeleton as the tile re-rank: lane-per-row, 128 B of the lane's own row per step, a
// tile of CT centroids applied to it as wave-uniform scalar operands through the rolled,
// software-pipelined loop (two chunk register sets ping-ponging behind lgkmcnt(0) waits),
// running sums in LDS (lsums[centroid][lane]).  Every (row, centroid) chain is summed in
// ascending group order exactly as squared_l2_distance does; the argmin uses strict '<' in
// ascending centroid order.  Exact-order f32 VALU-bound.
// ------------------------------------------------------------------------------------
template <int CT, bool ALIGNED>
__global__ __launch_bounds__(256) void assign_kernel(const float *__restrict__ rows, uint64_t n,
                                                    uint32_t dim,
                                                    const float *__restrict__ cent, uint32_t k,
                                                    uint32_t *__restrict__ cluster,
                                                    const uint32_t *__restrict__ prev,
                                                    unsigned long long *__restrict__ changed,
                                                    unsigned long long *__restrict__ sizes) {
    __shared__ float lsums_all[4 * CT * 64];
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    float *lsums = lsums_all + wave * (CT * 64);

    const uint64_t r = (uint64_t)blockIdx.x * 256 + threadIdx.x;
    const bool valid = r < n;
    const float *x = rows + (valid ? r : (n - 1)) * dim;
    const uint32_t G = dim >> 2, tail = dim & 3u;
    float best = INFINITY;
    uint32_t bestc = 0;

    for (uint32_t c0 = 0; c0 < k; c0 += CT) {
        const uint32_t cnt = (k - c0 < (uint32_t)CT) ? (k - c0) : (uint32_t)CT;
        uint32_t g0 = 0;
        for (; g0 + 8 <= G; g0 += 8) {
            float4 xv[8];
#pragma unroll
            for (int g = 0; g < 8; ++g) xv[g] = load4<ALIGNED>(x + (g0 + g) * 4);
            float4 qa[8], qb[8];
            {
                const float *cp = cent + (uint64_t)c0 * dim + g0 * 4;
#pragma unroll
                for (int g = 0; g < 8; ++g) qa[g] = load4_uniform<ALIGNED>(cp + g * 4);
            }
            uint32_t cc = 0;
#pragma unroll 1
            for (; cc + 2 <= cnt; cc += 2) {
                float acc0 = g0 ? lsums[cc * 64 + lane] : 0.0f;
                float acc1 = g0 ? lsums[(cc + 1) * 64 + lane] : 0.0f;
                __builtin_amdgcn_s_waitcnt(0xC07F);   // lgkmcnt(0)
                {
                    const float *cp = cent + (uint64_t)(c0 + cc + 1) * dim + g0 * 4;
#pragma unroll
                    for (int g = 0; g < 8; ++g) qb[g] = load4_uniform<ALIGNED>(cp + g * 4);
                }
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int g = 0; g < 8; ++g) {
                    const float d0 = xv[g].x - qa[g].x, d1 = xv[g].y - qa[g].y;
                    const float d2 = xv[g].z - qa[g].z, d3 = xv[g].w - qa[g].w;
                    float t = d0 * d0 + d1 * d1;
                    t = t + d2 * d2;
                    t = t + d3 * d3;
                    acc0 = acc0 + t;
                }
                lsums[cc * 64 + lane] = acc0;
                __builtin_amdgcn_sched_barrier(0);
                __builtin_amdgcn_s_waitcnt(0xC07F);
                {
                    const uint32_t nc2 = cc + 2 < cnt ? cc + 2 : cnt - 1;
                    const float *cp = cent + (uint64_t)(c0 + nc2) * dim + g0 * 4;
#pragma unroll
                    for (int g = 0; g < 8; ++g) qa[g] = load4_uniform<ALIGNED>(cp + g * 4);
                }
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int g = 0; g < 8; ++g) {
                    const float d0 = xv[g].x - qb[g].x, d1 = xv[g].y - qb[g].y;
                    const float d2 = xv[g].z - qb[g].z, d3 = xv[g].w - qb[g].w;
                    float t = d0 * d0 + d1 * d1;
                    t = t + d2 * d2;
                    t = t + d3 * d3;
                    acc1 = acc1 + t;
                }
                lsums[(cc + 1) * 64 + lane] = acc1;
                __builtin_amdgcn_sched_barrier(0);
            }
            if (cc < cnt) {
                float acc = g0 ? lsums[cc * 64 + lane] : 0.0f;
#pragma unroll
                for (int g = 0; g < 8; ++g) {
                    const float d0 = xv[g].x - qa[g].x, d1 = xv[g].y - qa[g].y;
                    const float d2 = xv[g].z - qa[g].z, d3 = xv[g].w - qa[g].w;
                    float t = d0 * d0 + d1 * d1;
                    t = t + d2 * d2;
                    t = t + d3 * d3;
                    acc = acc + t;
                }
                lsums[cc * 64 + lane] = acc;
            }
        }
        for (; g0 < G; ++g0) {
            const float4 xg = load4<ALIGNED>(x + g0 * 4);
#pragma unroll 1
            for (uint32_t cc = 0; cc < cnt; ++cc) {
                const float4 cv = load4_uniform<ALIGNED>(cent + (uint64_t)(c0 + cc) * dim + g0 * 4);
                const float d0 = xg.x - cv.x, d1 = xg.y - cv.y;
                const float d2 = xg.z - cv.z, d3 = xg.w - cv.w;
                float t = d0 * d0 + d1 * d1;
                t = t + d2 * d2;
                t = t + d3 * d3;
                const float acc = g0 ? lsums[cc * 64 + lane] : 0.0f;
                lsums[cc * 64 + lane] = acc + t;
            }
        }
        for (uint32_t e = 0; e < tail; ++e) {
            const float xe = x[G * 4 + e];
#pragma unroll 1
            for (uint32_t cc = 0; cc < cnt; ++cc) {
                const float d = xe - load1_uniform(cent + (uint64_t)(c0 + cc) * dim + G * 4 + e);
                const float acc = (G || e) ? lsums[cc * 64 + lane] : 0.0f;
                lsums[cc * 64 + lane] = acc + d * d;
            }
        }
        wave_lds_fence();
#pragma unroll 1
        for (uint32_t cc = 0; cc < cnt; ++cc) {
            const float v = lsums[cc * 64 + lane];
            if (v < best) { best = v; bestc = c0 + cc; }   // strict '<': lowest centroid wins ties
        }
        wave_lds_fence();
    }

    if (valid) cluster[r] = bestc;
    // counters: one atomic per wave and distinct cluster / per wave (every row adding to `changed` and to a
    // hundred cluster sizes on four cache lines serialises at the memory side)
    if (sizes) {
        unsigned long long m = __ballot(valid);
        while (m) {
            const uint32_t c = readlane_u32(bestc, __builtin_ctzll(m));
            const unsigned long long same = __ballot(valid && bestc == c);
            if (lane == __builtin_ctzll(m)) atomicAdd(&sizes[c], (unsigned long long)__popcll(same));
            m &= ~same;
        }
    }
    if (prev && changed) {
        const unsigned long long ch = __ballot(valid && prev[r] != bestc);
        if (lane == 0 && ch) atomicAdd(changed, (unsigned long long)__popcll(ch));
    }
}

hipError_t launch_assign(const float *rows, uint64_t n, uint32_t dim, const float *centroids,
                         uint32_t k, uint32_t *cluster, const uint32_t *prev,
                         unsigned long long *changed, unsigned long long *sizes, hipStream_t s) {
    if (n == 0) return hipSuccess;
    const uint64_t blocks = (n + 255) / 256;
    if (blocks > 0x7FFFFFFFull) return hipErrorInvalidValue;
    if (dim % 4 == 0)
        hipLaunchKernelGGL((assign_kernel<32, true>), dim3((uint32_t)blocks), dim3(256), 0, s, rows,
                           n, dim, centroids, k, cluster, prev, changed, sizes);
    else
        hipLaunchKernelGGL((assign_kernel<32, false>), dim3((uint32_t)blocks), dim3(256), 0, s, rows,
                           n, dim, centroids, k, cluster, prev, changed, sizes);
    return hipGetLastError();
}

// ------------------------------------------------------------------------------------
// lloyd_update: thread (c, j) adds x[r][j] over cluster c's rows in ascending r -- the
// same per-element add order as the reference's single-threaded loop (index.rs:438-444)
// -- then divides by the size (index.rs:446-453); empty clusters stay all-zero.
// ------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void lloyd_update_kernel(const float *__restrict__ rows,
                                                          uint32_t dim,
                                                          const uint32_t *__restrict__ list_rows,
                                                          const uint64_t *__restrict__ list_off,
                                                          float *__restrict__ centroids) {
    const uint32_t c = blockIdx.x;
    const uint32_t jj = blockIdx.y * 256 + threadIdx.x;
    if (jj >= dim) return;
    const uint64_t b = list_off[c], e = list_off[c + 1];
    float acc = 0.0f;
    uint64_t i = b;
    for (; i + 4 <= e; i += 4) {
        const float v0 = rows[(uint64_t)list_rows[i] * dim + jj];
        const float v1 = rows[(uint64_t)list_rows[i + 1] * dim + jj];
        const float v2 = rows[(uint64_t)list_rows[i + 2] * dim + jj];
        const float v3 = rows[(uint64_t)list_rows[i + 3] * dim + jj];
        acc = acc + v0; acc = acc + v1; acc = acc + v2; acc = acc + v3;
    }
    for (; i < e; ++i) acc = acc + rows[(uint64_t)list_rows[i] * dim + jj];
    if (e > b) acc = div_f32_ieee(acc, (float)(e - b));
    centroids[(uint64_t)c * dim + jj] = acc;
}

hipError_t launch_lloyd_update(const float *rows, uint32_t dim, const uint32_t *list_rows,
                               const uint64_t *list_off, uint32_t k, float *centroids,
                               hipStream_t s) {
    if (k == 0) return hipSuccess;
    dim3 grid(k, (dim + 255) / 256);
    hipLaunchKernelGGL(lloyd_update_kernel, grid, dim3(256), 0, s, rows, dim, list_rows, list_off,
                       centroids);
    return hipGetLastError();
}

}  // namespace pqv
