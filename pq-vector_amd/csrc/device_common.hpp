// device_common.hpp -- device-side helpers shared by the kernel translation units (kernels_*.hip): wave-level
// primitives (readlane, DPP scans, bitonic sort, rank selection), buffer loads, the MFMA step wrappers, WaveTopk (the
// wave-distributed sorted list every merge uses), tile_fold, the int8 quantiser and the sortable float keys.
// Everything here computes the reference's squared-L2 in the reference's exact f32 summation order (no FMA contraction:
// built with -ffp-contract=off), so distances are bit-identical to the CPU path and every argmin / top-k decision is too.
#pragma once
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
// one dword per lane through a (wave-uniform) buffer descriptor: no 64-bit per-lane address arithmetic, no pointer registers
// (AUX: cache policy bits of the load -- 16 = sc1: agent scope, what a relaxed agent-scope atomic load compiles to on gfx942 / gfx950)
template <int AUX = 0>
__device__ __forceinline__ uint32_t buf_ld4(__amdgpu_buffer_rsrc_t r, uint32_t lane_bytes, uint32_t uniform_bytes) {
    return (uint32_t)__builtin_amdgcn_raw_buffer_load_b32(r, (int)lane_bytes, (int)uniform_bytes, AUX);
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
#ifndef PQV_NS_REG
#define PQV_NS_REG 2           // ... of the 4-wave int8 instances (two blocks per CU)
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
#ifndef PQV_RELANE
#define PQV_RELANE 1           // wide_filter_kernel: the lane index is re-defined opaquely after every K loop (no hoisted lane-derived invariants)
#endif
#ifndef PQV_REPF
#define PQV_REPF 1             // wide_filter_kernel: the operand prefetch is issued again behind exact evaluations in mid-wave (stages dead across them)
#endif
#ifndef PQV_NA_REG
#define PQV_NA_REG 0           // regular int8 instance: branch-free K loop bodies for quads of 3 / 4 / 5 groups too (124 bytes of spills: off)
#endif
#ifndef PQV_THR_EVERY
#define PQV_THR_EVERY 1        // 64-row-tile instances: thresholds re-read behind every n-th tile
#endif
#ifndef PQV_XTC
#define PQV_XTC 0              // 64-row-tile int8 instances: the next tile's operand stages go out FIRST after a K loop, the fresh thresholds
#endif                         // and the next tile's row terms behind them -- nothing is waited for between two K loops (thresholds one tile old)
#ifndef PQV_EVAL_NB
#define PQV_EVAL_NB 16         // ... in the regular int8 instance (8: C3 kernels + 0.8 %, mixture + 2.4 %; its accumulators are dead there too)
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
// k-th smallest of n 32-bit values (1 <= k <= n), by the whole block: four 8-bit radix passes over a 256-bin LDS histogram;
// get(i) returns value i (read four times).  Every thread of the block must call it; s_hist: 256 words, s_sel: 2 words.
// ------------------------------------------------------------------------------------
template <class F>
__device__ __forceinline__ uint32_t block_kth_u32(F get, uint32_t n, uint32_t k, uint32_t *s_hist, uint32_t *s_sel) {
    const int lane = threadIdx.x & 63;
    uint32_t prefix = 0, mask = 0, rank = k;
    for (int shift = 24; shift >= 0; shift -= 8) {
        for (uint32_t i = threadIdx.x; i < 256; i += blockDim.x) s_hist[i] = 0;
        __syncthreads();
        for (uint32_t i = threadIdx.x; i < n; i += blockDim.x) {
            const uint32_t u = get(i);
            if ((u & mask) == prefix) atomicAdd(&s_hist[(u >> shift) & 255u], 1u);
        }
        __syncthreads();
        if (threadIdx.x < 64) {
            // 4 bins per lane, inclusive scan over the lanes, the digit whose cumulative count reaches `rank`
            uint32_t c4[4], tot = 0;
#pragma unroll
            for (int j = 0; j < 4; ++j) { c4[j] = s_hist[4 * lane + j]; tot += c4[j]; }
            const uint32_t incl = wave_incl_scan_u32(tot);
            uint32_t before = incl - tot;
            int digit = -1;
            uint32_t below = 0;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                if (digit < 0 && before < rank && rank <= before + c4[j]) { digit = 4 * lane + j; below = before; }
                before += c4[j];
            }
            const unsigned long long hit = __ballot(digit >= 0);
            const int src = __builtin_ctzll(hit ? hit : 1ull);
            const uint32_t d = (uint32_t)__shfl(digit, src, 64), bl = (uint32_t)__shfl((int)below, src, 64);
            if (lane == 0) { s_sel[0] = prefix | (d << shift); s_sel[1] = rank - bl; }
        }
        __syncthreads();
        prefix = s_sel[0]; rank = s_sel[1];
        mask |= 0xFFu << shift;
    }
    return prefix;
}

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


__device__ __forceinline__ int quant_i8(float t, float scale) {
    const float v = rintf(t * scale);
    return (int)fminf(fmaxf(v, -127.0f), 127.0f);      // NaN -> -127 (callers flag non-finite inputs separately)
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

typedef float f32x16_t __attribute__((ext_vector_type(16)));

__device__ __forceinline__ uint32_t sortable_bits(float d) {
    const uint32_t b = __float_as_uint(d);
    return (b & 0x80000000u) ? ~b : (b | 0x80000000u);
}
__device__ __forceinline__ float unsortable_bits(uint32_t u) {
    return __uint_as_float((u & 0x80000000u) ? (u & 0x7FFFFFFFu) : ~u);
}

constexpr int BH_BM = 128, BH_BN = 256, BH_BK = 32;       // assign_f16_kernel's tile (and the 4-wave form of brute_f16_kernel)
typedef int i32x16_t __attribute__((ext_vector_type(16)));
typedef float f32x4_raw_t __attribute__((ext_vector_type(4)));

}  // namespace pqv
