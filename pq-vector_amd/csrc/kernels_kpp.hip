// kernels_kpp.hip -- the k-means++ pick on the device (src/ivf/index.rs:354-390), so that the rounds run back to back on the stream
// without a host round trip per centroid.
//
// What the reference computes per round, on the current minima md[0..n):
//     total     = sum over the worker chunks (ascending) of each chunk's SEQUENTIAL f32 sum                 (:356-370, :259-265)
//     threshold = gen_range(0.0..1.0) * total                                                              (:373)
//     pick      = the first slot whose SEQUENTIAL f32 cumulative sum (from slot 0) is >= threshold         (:374-383)
// Both are chains c <- fl(c + x_i): every partial sum's rounding feeds the next, so the order is fixed -- but the chain can still be
// evaluated in parallel, EXACTLY.  While c stays inside one binade [2^e, 2^(e+1)) its ulp u = 2^(e-23) is fixed, C = c / u is an integer
// in [2^23, 2^24), and for 0 <= x < 2^(e+1)
//     fl(c + x) = u * (C + a + g + (tie & odd(C + a))),   t = x / u (exact),  a = floor(t),  f = t - a,  g = [f > 1/2],  tie = [f == 1/2]
// (round to nearest even) as long as the result is below 2^24 u.  So an element is a map "parity of C -> increment", a pair (d0, d1) of
// integers; maps compose associatively ((A then B)(p) = A(p) + B(p xor odd(A(p)))), hence a RUN of consecutive elements collapses to one
// pair and a wave scans 64 runs in six steps.  What the pair needs is the binade: it is predicted from an approximate (any-order) prefix
// sum and CHECKED against the exact c when the chain arrives there; a run whose prediction is wrong, or inside which c crosses into the
// next binade (about log2(n) runs of a chain), is added element by element with ordinary f32 adds.  Nothing is approximated: the run
// summaries are integer identities of the IEEE additions they stand for, and the fallback IS those additions.  (tests: blob identity of
// the build against the host walk -- PQV_KPP_DEVICE=0 -- and against the oracle; tools/fuzz_build.sh.)
//
// One launch per round, blocks of 1024 threads, three roles by block index:
//     summary blocks  [0, SB)             wave 0 of block s = the pairs of QUARTER runs 64 s .. 64 s + 63 of the PICK chain (14 elements per
//                                         lane; the chain block composes four of them into a run of 56); its binade prediction needs the
//                                         approximate sum of everything before: the summary blocks publish their own sums and read those
//                                         of the blocks before them (lower-numbered, dispatched earlier: the wait cannot deadlock)
//     head block      SB                  one wave: the chain's first 64 runs (where c doubles every few runs and most of the element-wise
//                                         additions are) worked through while the summary blocks are still summarising; needs nobody
//     chain block     SB + 1              thread t keeps run t's 56 elements in registers; the 16 waves take their 64 runs in turn
//                                         (wave_turn; a wave whose runs were composed in advance and agree with the arriving value is
//                                         passed over by the wave before it), leaving the exact c at every run start in LDS; then total,
//                                         threshold, the first run that reaches it, and the walk inside that run.  It waits for the
//                                         summary blocks (lower-numbered) and, at the very end, for the chunk sums (ONE waiting block:
//                                         the chunk blocks get their slots whatever the device's size)
//     chunk blocks    (SB + 1, ..]        the whole chain of a worker chunk -> chunk_sum[c]: one block per chunk (run length EC), or one
//                                         WAVE per chunk where chunks have at most 512 elements (a many-core host: 256 chunks of 196)
// Cross-block values are single 64-bit words tagged with the round number (agent-scope relaxed atomics: no cache-wide fences), so
// nothing has to be reset between rounds.  A round the device cannot decide the reference's way -- total not positive or not finite, a
// value that is not a finite non-negative number, no slot reaching the threshold, a wait that ran out -- sets state[0]: the later
// launches return at once and the host takes over from state[1] with its own walk.
#include "device_common.hpp"

namespace pqv {

namespace {

constexpr int KPP_E = 56;                   // run length of the pick chain: 1024 runs x 56 >= 57 344 elements
constexpr int KPP_Q = 14;                   // ... summarised in quarters by the summary blocks
constexpr uint32_t KPP_SAT = 1u << 25;      // increments saturate here (anything >= 2^24 is a crossing and is redone exactly)

__device__ __forceinline__ uint32_t kpp_bits(float x) { return __builtin_bit_cast(uint32_t, x); }
__device__ __forceinline__ float kpp_float(uint32_t b) { return __builtin_bit_cast(float, b); }
__device__ __forceinline__ unsigned long long kpp_load(const unsigned long long *p) {
    return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ void kpp_store(unsigned long long *p, unsigned long long v) {
    __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
// the word at p once it carries this round's tag; false when the wait ran out (the caller gives the round back to the host)
__device__ __forceinline__ bool kpp_wait(const unsigned long long *p, uint32_t round, unsigned long long &v) {
    for (uint32_t spin = 0; spin < (1u << 22); ++spin) {
        v = kpp_load(p);
        if ((uint32_t)(v >> 32) == round) return true;
        __builtin_amdgcn_s_sleep(1);
    }
    return false;
}
// diagnostics (pqv_kpp_pick with PQV_KPP_STAMPS=1): the 100 MHz clock at a few places of a launch
__device__ __forceinline__ void kpp_stamp(unsigned long long *stamps, int i) {
    if (stamps) stamps[i] = wall_clock64();
}

// (A then B): the pair of two consecutive stretches
__device__ __forceinline__ void pair_then(uint32_t p0, uint32_t p1, uint32_t &a0, uint32_t &a1) {
    const uint32_t n0 = p0 + ((p0 & 1u) ? a1 : a0);
    const uint32_t n1 = p1 + ((p1 & 1u) ? a0 : a1);
    a0 = n0; a1 = n1;
}
// inclusive composition over the lanes on the DPP network (row_shr 1/2/4/8, row_bcast 15/31: device_common.hpp's prefix sum with the
// pair product in place of '+'); a lane without a source receives (0, 0), the identity.  64 pairs of <= 2^25 each stay below 2^32.
__device__ __forceinline__ void pair_scan(uint32_t &a0, uint32_t &a1) {
#ifdef KPP_SHFL_SCAN
    const int lane = (int)(threadIdx.x & 63u);
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
        const uint32_t p0 = (uint32_t)__shfl_up((int)a0, off, 64), p1 = (uint32_t)__shfl_up((int)a1, off, 64);
        if (lane >= off) pair_then(p0, p1, a0, a1);
    }
    return;
#endif
#define KPP_SCAN_STEP(CTRL, ROWS)                                                                              \
    {                                                                                                          \
        const uint32_t p0 = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)a0, CTRL, ROWS, 0xF, false);         \
        const uint32_t p1 = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)a1, CTRL, ROWS, 0xF, false);         \
        pair_then(p0, p1, a0, a1);                                                                             \
    }
    KPP_SCAN_STEP(0x111, 0xF) KPP_SCAN_STEP(0x112, 0xF) KPP_SCAN_STEP(0x114, 0xF) KPP_SCAN_STEP(0x118, 0xF)
    KPP_SCAN_STEP(0x142, 0xA) KPP_SCAN_STEP(0x143, 0xC)
#undef KPP_SCAN_STEP
}
// the value of the lane below (lane 0: 0)
// (call it with every lane active: a lane that is masked off delivers nothing)
__device__ __forceinline__ uint32_t lane_below(uint32_t v) {
#ifdef KPP_SHFL_BELOW
    return (uint32_t)__shfl_up((int)v, 1, 64);
#else
    return (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x138, 0xF, 0xF, false);      // wave_shr:1
#endif
}
// inclusive f32 prefix over the lanes (an approximation is all that is asked of it: the order of these adds is free)
__device__ __forceinline__ float approx_scan(float v) {
#define KPP_F_STEP(CTRL, ROWS) v += kpp_float((uint32_t)__builtin_amdgcn_update_dpp(0, (int)kpp_bits(v), CTRL, ROWS, 0xF, false));
    KPP_F_STEP(0x111, 0xF) KPP_F_STEP(0x112, 0xF) KPP_F_STEP(0x114, 0xF) KPP_F_STEP(0x118, 0xF) KPP_F_STEP(0x142, 0xA) KPP_F_STEP(0x143, 0xC)
#undef KPP_F_STEP
    return v;
}

// (d0, d1) of E consecutive elements under the ulp of biased exponent e (23 <= e < 254)
template <int E>
__device__ __forceinline__ void run_summary(const float (&xv)[E], uint32_t e, uint32_t &D0, uint32_t &D1) {
    const float scale = kpp_float((277u - e) << 23);          // 2^(23 - (e - 127)): x * scale = x / ulp, exact
    const float limit = kpp_float((e + 1u) << 23);            // 2^(e + 1 - 127)
    uint32_t d0 = 0, d1 = 0;
#pragma unroll
    for (int j = 0; j < E; ++j) {
        const float x = xv[j];
        const float t = x * scale;
        const float fl = floorf(t);
        const float fr = t - fl;
        uint32_t a = (uint32_t)fl;
        uint32_t g = fr > 0.5f ? 1u : 0u;
        uint32_t tie = fr == 0.5f ? 1u : 0u;
        if (!(x < limit)) { a = 1u << 24; g = 0; tie = 0; }   // leaves the binade by itself (also NaN: the chain block flags those)
        const uint32_t ag = a + g;
        d0 += ag + (tie & (a ^ d0) & 1u);                     // parity before the element: d0 & 1 (the stretch started even)
        d1 += ag + (tie & (a ^ ~d1) & 1u);                    // ... (the stretch started odd)
    }
    D0 = d0 < KPP_SAT ? d0 : KPP_SAT;
    D1 = d1 < KPP_SAT ? d1 : KPP_SAT;
}

// c + run f's elements, one f32 add after the other.  Every lane walks its OWN run from the same c (56 dependent vector adds, no
// cross-lane traffic inside the chain); lane f's result is the one that counts.  (Measured and dropped: walking by quarters with the
// quarters' own pairs -- one integer addition for a quarter that agrees -- 20.1 against 17.8 us for the launch.)
template <int E>
__device__ __forceinline__ float run_adds(float c, const float (&xv)[E], int f) {
    float cl = c;
#pragma unroll
    for (int j = 0; j < E; ++j) cl = cl + xv[j];
    return kpp_float(readlane_u32(kpp_bits(cl), f));
}

// The wave's nl runs (lane l = run l: elements xv, predicted exponent e, usable, pair D0 / D1) appended to the chain that arrives with
// the exact value c, from run `cur` on.  Returns the exact value after the last run; table (if any) receives the exact value at each
// run's start.
template <int E>
__device__ __forceinline__ float wave_chain(float c, int cur, const float (&xv)[E], uint32_t e, bool usable, uint32_t D0, uint32_t D1, int nl,
                                            float *table, int lane) {
    while (cur < nl) {
        const uint32_t cb = (uint32_t)__builtin_amdgcn_readfirstlane((int)kpp_bits(c)), ec = cb >> 23;   // (c is the same in every lane: scalar code from here)
        const bool okl = usable && e == ec && lane >= cur && lane < nl;
        const unsigned long long okm = __ballot(okl);
        const unsigned long long from = ~0ull << cur;
        const unsigned long long stopm = ~okm & from;
        const int hi = stopm ? __builtin_ctzll(stopm) : 64;   // (lanes >= nl are never ok: hi <= nl)
        if (hi == cur) {                                      // this run is not covered by its summary: the additions themselves
            if (table && lane == cur) table[cur] = c;
            c = run_adds<E>(c, xv, cur);
            ++cur;
            continue;
        }
        const uint32_t C = (cb & 0x7FFFFFu) | 0x800000u;
        const bool in = lane >= cur && lane < hi;
        uint32_t a0 = in ? D0 : 0u, a1 = in ? D1 : 0u;        // inclusive composition over [cur, lane]
        pair_scan(a0, a1);
        const uint32_t Iv = (C & 1u) ? a1 : a0;
        const bool cross = in && C + Iv >= (1u << 24);
        const unsigned long long cm = __ballot(cross);
        const int f = cm ? __builtin_ctzll(cm) : hi;          // first run that is not simply "C + increment": crossing, or the run at hi
        const uint32_t below = lane_below(Iv);
        const uint32_t Cst = lane == cur ? C : C + below;     // exact integer at the start of run `lane`, for lanes in [cur, f]
        if (table && lane >= cur && lane <= f && lane < nl) table[lane] = kpp_float((ec << 23) | (Cst & 0x7FFFFFu));
        if (f >= nl) {                                        // every remaining run was covered
            const uint32_t Cend = C + readlane_u32(Iv, nl - 1);
            c = kpp_float((ec << 23) | (Cend & 0x7FFFFFu));
            cur = nl;
            break;
        }
        c = kpp_float((ec << 23) | (readlane_u32(Cst, f) & 0x7FFFFFu));
        c = run_adds<E>(c, xv, f);
        cur = f + 1;
    }
    return c;
}

// A wave's turn.  Before the turns start every wave has composed its runs in SEGMENTS -- maximal stretches of usable runs predicted
// into the same binade (a segmented scan: a head flag stops the composition) -- so that a turn needs no scan: for the segment at `cur`
// the arriving c either agrees with the prediction (then the first crossing, if any, is one ballot away) or the run is walked.  A wave
// that is one segment and is not crossed can even be passed over by the wave before it (WavePre::ok, e0 and the last lane's pair).
struct WavePre { bool ok; uint32_t e0; uint32_t I0, I1; unsigned long long H, U; };
__device__ __forceinline__ WavePre wave_pre(uint32_t e, bool usable, uint32_t D0, uint32_t D1, int nl, int lane) {
    WavePre p;
    const uint32_t e_below = lane_below(e), us_below = lane_below(usable ? 1u : 0u);
    bool h = lane == 0 || !usable || !us_below || e != e_below;
    p.H = __ballot(h); p.U = __ballot(usable);
    uint32_t a0 = usable ? D0 : 0u, a1 = usable ? D1 : 0u;
#define KPP_SEG_STEP(CTRL, ROWS)                                                                               \
    {                                                                                                          \
        const uint32_t p0 = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)a0, CTRL, ROWS, 0xF, false);         \
        const uint32_t p1 = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)a1, CTRL, ROWS, 0xF, false);         \
        const uint32_t ph = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)(h ? 1u : 0u), CTRL, ROWS, 0xF, false); \
        if (!h) { pair_then(p0, p1, a0, a1); h = ph != 0; }                                                    \
    }
    // (a lane without a source composes with the identity and keeps looking: the row broadcasts bring the rows before it)
    KPP_SEG_STEP(0x111, 0xF) KPP_SEG_STEP(0x112, 0xF) KPP_SEG_STEP(0x114, 0xF) KPP_SEG_STEP(0x118, 0xF)
    KPP_SEG_STEP(0x142, 0xA) KPP_SEG_STEP(0x143, 0xC)
#undef KPP_SEG_STEP
    p.I0 = a0; p.I1 = a1;
    const unsigned long long lanes = nl >= 64 ? ~0ull : ((1ull << nl) - 1ull);
    p.e0 = readlane_u32(e, 0);
    p.ok = nl > 0 && (p.H & lanes) == 1ull && (p.U & lanes) == lanes;
    return p;
}
template <int E>
__device__ __forceinline__ float wave_turn(float c, const WavePre &pre, const float (&xv)[E], uint32_t e, bool usable, uint32_t D0, uint32_t D1,
                                           int nl, float *table, int lane) {
    const unsigned long long lanes = nl >= 64 ? ~0ull : ((1ull << nl) - 1ull);
    int cur = 0;
    while (cur < nl) {
        const uint32_t cb = (uint32_t)__builtin_amdgcn_readfirstlane((int)kpp_bits(c)), ec = cb >> 23;
        const bool us_cur = (pre.U >> cur) & 1ull, head_cur = (pre.H >> cur) & 1ull;
        const bool agree = us_cur && readlane_u32(e, cur) == ec;
        if (agree && !head_cur) return wave_chain<E>(c, cur, xv, e, usable, D0, D1, nl, table, lane);   // inside a segment: the scanning walk
        if (!agree) {                                         // not covered by its summary: the additions themselves
            if (table && lane == cur) table[cur] = c;
            c = run_adds<E>(c, xv, cur);
            ++cur;
            continue;
        }
        const unsigned long long later = pre.H & lanes & ((~0ull << cur) << 1);
        const int se = later ? __builtin_ctzll(later) : nl;   // the segment is [cur, se)
        const uint32_t C = (cb & 0x7FFFFFu) | 0x800000u;
        const uint32_t Iv = (C & 1u) ? pre.I1 : pre.I0;
        const bool in = lane >= cur && lane < se;
        const unsigned long long cm = __ballot(in && C + Iv >= (1u << 24));
        const int f = cm ? __builtin_ctzll(cm) : se;          // the first run that crosses into the next binade, or the segment's end
        const uint32_t below = lane_below(Iv);
        const uint32_t Cst = lane == cur ? C : C + below;     // exact integer at the start of run `lane`, for lanes in [cur, f]
        if (table && lane >= cur && lane <= f && lane < se) table[lane] = kpp_float((ec << 23) | (Cst & 0x7FFFFFu));
        if (f == se) {
            const uint32_t Cend = C + readlane_u32(Iv, se - 1);
            c = kpp_float((ec << 23) | (Cend & 0x7FFFFFu));
            cur = se;
            continue;
        }
        c = kpp_float((ec << 23) | (readlane_u32(Cst, f) & 0x7FFFFFu));
        c = run_adds<E>(c, xv, f);
        cur = f + 1;
    }
    return c;
}

// E elements x[off .. off + E) (zero beyond len: c + 0 == c), V floats per load where the address allows
template <int E>
__device__ __forceinline__ void load_run(const float *x, uint32_t off, uint32_t len, float (&xv)[E]) {
    const uintptr_t addr = reinterpret_cast<uintptr_t>(x + off);
    if (off + E <= len && (E % 4) == 0 && (addr & 15u) == 0) {
#pragma unroll
        for (int j = 0; j + 3 < E; j += 4) {
            const float4 v = *reinterpret_cast<const float4 *>(x + off + j);
            xv[j] = v.x; xv[j + 1] = v.y; xv[j + 2] = v.z; xv[j + 3] = v.w;
        }
    } else if (off + E <= len && (E % 2) == 0 && (addr & 7u) == 0) {
#pragma unroll
        for (int j = 0; j + 1 < E; j += 2) {
            const float2 v = *reinterpret_cast<const float2 *>(x + off + j);
            xv[j] = v.x; xv[j + 1] = v.y;
        }
    } else {
#pragma unroll
        for (int j = 0; j < E; ++j) xv[j] = off + j < len ? x[off + j] : 0.0f;
    }
}

// One block = one whole chain from c = 0 over x[0 .. len), len <= 1024 E (the worker chunks): approximate prefix, pairs, the waves in turn.
template <int E>
__device__ float block_chain(const float *x, uint32_t len, float *sh_wsum, float *sh_c) {
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int n_runs = (int)((len + E - 1) / E), n_waves = (n_runs + 63) >> 6;
    if (tid == 0) *sh_c = 0.0f;
    if (wave >= n_waves) {                                   // (no runs here; the barriers below are counted by n_waves)
        for (int w = 0; w <= n_waves; ++w) __syncthreads();
        return *sh_c;
    }
    float xv[E];
    load_run<E>(x, (uint32_t)tid * E, len, xv);
    float s = 0.0f;
#pragma unroll
    for (int j = 0; j < E; ++j) s += xv[j];
    const float inc = approx_scan(s);                        // approximate inclusive prefix inside the wave (any order: a prediction)
    if (lane == 63) sh_wsum[wave] = inc;
    __syncthreads();
    float base = 0.0f;
    for (int w = 0; w < wave; ++w) base += sh_wsum[w];
    const float A = base + (inc - s);
    const uint32_t e = kpp_bits(A) >> 23;
    const bool usable = e >= 23u && e < 254u;                // (a negative or NaN prediction has e >= 256 or is caught by the exact check)
    uint32_t D0 = 0, D1 = 0;
    if (usable) run_summary<E>(xv, e, D0, D1);
    const int nl = n_runs - 64 * wave < 64 ? n_runs - 64 * wave : 64;
    const WavePre pre = wave_pre(e, usable, D0, D1, nl, lane);
    for (int w = 0; w < n_waves; ++w) {
        if (w == wave) {
            const float c = wave_turn<E>(*sh_c, pre, xv, e, usable, D0, D1, nl, nullptr, lane);
            if (lane == 0) *sh_c = c;
        }
        __syncthreads();
    }
    return *sh_c;
}

// One wave = one whole chain from c = 0 over x[0 .. len), len <= 512 (small worker chunks): no LDS, no barriers.
__device__ __forceinline__ float wave_chain_alone(const float *x, uint32_t len, int lane) {
    constexpr int E = 8;
    float xv[E];
    load_run<E>(x, (uint32_t)lane * E, len, xv);
    float s = 0.0f;
#pragma unroll
    for (int j = 0; j < E; ++j) s += xv[j];
    const float inc = approx_scan(s);
    const uint32_t e = kpp_bits(inc - s) >> 23;
    const bool usable = e >= 23u && e < 254u;
    uint32_t D0 = 0, D1 = 0;
    if (usable) run_summary<E>(xv, e, D0, D1);
    const int nl = (int)((len + E - 1) / E);
    const WavePre pre = wave_pre(e, usable, D0, D1, nl, lane);
    return wave_turn<E>(0.0f, pre, xv, e, usable, D0, D1, nl, nullptr, lane);
}

// total (:356-370): the chunk sums (LDS) joined in ascending chunk order from 0, by one wave -- 64 at a time in registers, each added
// from a scalar register (the adds are the chain; the lane reads run ahead of it)
__device__ __forceinline__ float join_chunk_sums(const float *cs, uint32_t jobs, int lane) {
    float c = 0.0f;
    for (uint32_t base = 0; base < jobs; base += 64) {
        const uint32_t sv = base + (uint32_t)lane < jobs ? kpp_bits(cs[base + lane]) : 0u;   // (+0.0 beyond the last chunk: c + 0 == c)
#pragma unroll
        for (int l = 0; l < 64; ++l) c = c + kpp_float(readlane_u32(sv, l));
    }
    return c;
}

}  // namespace

template <int EC>
__global__ __launch_bounds__(1024) void kpp_pick_kernel(const KppPickArgs a) {
    __shared__ float sh_wsum[16];
    __shared__ float sh_c;
    __shared__ float sh_table[1025];
    __shared__ float sh_cs[1024];
    __shared__ uint32_t sh_first, sh_fail, sh_next[2];
    __shared__ uint32_t sh_wok[16], sh_we0[16], sh_wT0[16], sh_wT1[16], sh_wskip[16];
    __shared__ float sh_wstart[16], sh_total;
    __shared__ uint32_t sh_cs_done, sh_total_done;
    if (a.state[0] != 0) return;                              // an earlier round went back to the host
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const uint32_t round = a.round;
    const unsigned long long tag = (unsigned long long)round << 32;
    const uint32_t jobs = a.n_chunks > 1 ? a.n_chunks : 0;    // (one chunk: its sum is the pick chain's last value)
    const uint32_t n_runs = (a.n + KPP_E - 1) / KPP_E, n_q = 4 * n_runs, SB = (n_q + 63) / 64;

    if (blockIdx.x == SB) {                                   // ---- the head of the pick chain: runs 0 .. 63 from c = 0
        if (wave != 0) return;
        float xv[KPP_E];
        const int nl = n_runs < 64u ? (int)n_runs : 64;
        load_run<KPP_E>(a.md, (uint32_t)lane * KPP_E, lane < nl ? a.n : 0u, xv);
        float s = 0.0f;
#pragma unroll
        for (int j = 0; j < KPP_E; ++j) s += xv[j];
        const float inc = approx_scan(s);
        const uint32_t e = kpp_bits(inc - s) >> 23;
        const bool usable = e >= 23u && e < 254u;
        uint32_t D0 = 0, D1 = 0;
        if (usable) run_summary<KPP_E>(xv, e, D0, D1);
        const WavePre pre = wave_pre(e, usable, D0, D1, nl, lane);
        const float c = wave_turn<KPP_E>(0.0f, pre, xv, e, usable, D0, D1, nl, sh_table, lane);
        wave_lds_fence();
        if (lane < nl) kpp_store(a.head + lane, tag | kpp_bits(sh_table[lane]));
        if (lane == 0) kpp_store(a.head + 64, tag | kpp_bits(c));
        if (lane == 0) kpp_stamp(a.stamps, 18);
        return;
    }
    if (blockIdx.x > SB + 1) {                                // ---- worker chunks
        if (EC == 0) {                                        // one wave per chunk
            const uint32_t ch = (blockIdx.x - SB - 2) * 16 + (uint32_t)wave;
            if (ch >= jobs) return;
            const uint32_t s0 = ch * a.chunk;
            const uint32_t len = a.n - s0 < a.chunk ? a.n - s0 : a.chunk;
            const float c = wave_chain_alone(a.md + s0, len, lane);
            if (lane == 0) kpp_store(a.chunk_sum + ch, tag | kpp_bits(c));
            return;
        }
        const uint32_t ch = blockIdx.x - SB - 2;
        const uint32_t s0 = ch * a.chunk;
        const uint32_t len = a.n - s0 < a.chunk ? a.n - s0 : a.chunk;
        if (tid == 0 && ch == 0) kpp_stamp(a.stamps, 16);
        const float c = block_chain<(EC ? EC : 8)>(a.md + s0, len, sh_wsum, &sh_c);
        if (tid == 0) kpp_store(a.chunk_sum + ch, tag | kpp_bits(c));
        if (tid == 0 && ch == 0) kpp_stamp(a.stamps, 17);
        return;
    }
    if (blockIdx.x < SB) {                                    // ---- the pairs of 64 quarter runs of the pick chain
        if (wave != 0) return;
        const uint32_t sb = blockIdx.x, q = sb * 64 + (uint32_t)lane;
        const bool st = lane == 0 && sb == SB - 1;
        if (st) kpp_stamp(a.stamps, 8);
        float xv[KPP_Q];
        load_run<KPP_Q>(a.md, q * KPP_Q, a.n, xv);
        float s = 0.0f;
#pragma unroll
        for (int j = 0; j < KPP_Q; ++j) s += xv[j];
        const float inc = approx_scan(s);
        if (lane == 63) kpp_store(a.blk_sum + sb, tag | kpp_bits(inc));
        if (st) kpp_stamp(a.stamps, 9);
        float before = 0.0f;                                  // the sums of the summary blocks before this one (SB <= 64: one lane each)
        bool ok = true;
        if ((uint32_t)lane < sb) {
            unsigned long long v = 0;
            ok = kpp_wait(a.blk_sum + lane, round, v);
            before = kpp_float((uint32_t)v);
        }
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) before += __shfl_xor(before, off, 64);
        const float A = before + (inc - s);
        if (st) kpp_stamp(a.stamps, 10);
        uint32_t e = kpp_bits(A) >> 23;
        if (!(e >= 23u && e < 254u) || __ballot(!ok)) e = 0;  // 0: the chain block adds this run element by element
        uint32_t D0 = 0, D1 = 0;
        if (e) run_summary<KPP_Q>(xv, e, D0, D1);
        if (st) kpp_stamp(a.stamps, 11);
        if (q < n_q) kpp_store(a.run_sum + q, (unsigned long long)D0 | ((unsigned long long)D1 << 26) | ((unsigned long long)e << 52));
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // the wave's stores have landed before its flag goes out
        if (lane == 0) kpp_store(a.blk_done + sb, tag);
        if (st) kpp_stamp(a.stamps, 12);
        return;
    }

    // ---- the chain block
    if (tid == 0) kpp_stamp(a.stamps, 0);
    float xv[KPP_E];
    const uint32_t t = (uint32_t)tid;
    // (measured and dropped: the wave's 3584 floats read as coalesced rows and handed to the lanes through LDS -- 22.1 against 20.5 us)
    load_run<KPP_E>(a.md, t * KPP_E, t < n_runs ? a.n : 0u, xv);
    bool badv = false;
#pragma unroll
    for (int j = 0; j < KPP_E; ++j) badv = badv || !(xv[j] >= 0.0f && xv[j] < INFINITY);
    if (tid == 0) { sh_c = 0.0f; sh_first = 0xFFFFFFFFu; sh_fail = 0; sh_cs_done = 0; sh_total_done = 0; }
    __syncthreads();
    if (badv) sh_fail = 2;
    if (tid == 0) kpp_stamp(a.stamps, 1);
    if (t < SB) {
        unsigned long long v;
        if (!kpp_wait(a.blk_done + t, round, v)) sh_fail = 4;
    }
    __syncthreads();
    __atomic_signal_fence(__ATOMIC_SEQ_CST);
    if (tid == 0) kpp_stamp(a.stamps, 2);
    uint32_t e = 0, D0 = 0, D1 = 0;
    if (t < n_runs) {                                         // the run's pair from its four quarters (one binade, or the run is walked)
        bool same = true;
#pragma unroll
        for (int h = 0; h < 4; ++h) {
            const unsigned long long v = kpp_load(a.run_sum + 4 * t + h);
            uint32_t q0 = (uint32_t)v & 0x3FFFFFFu, q1 = (uint32_t)(v >> 26) & 0x3FFFFFFu;
            const uint32_t qe = (uint32_t)(v >> 52) & 0xFFu;
            if (h == 0) { e = qe; D0 = q0; D1 = q1; }
            else { same = same && qe == e; pair_then(D0, D1, q0, q1); D0 = q0; D1 = q1; }
        }
        if (!same) e = 0;
        D0 = D0 < KPP_SAT ? D0 : KPP_SAT; D1 = D1 < KPP_SAT ? D1 : KPP_SAT;
    }
    const bool usable = e >= 23u && e < 254u;
    const int n_waves = (int)((n_runs + 63) >> 6);
    const int nl = (int)n_runs - 64 * wave < 64 ? (int)n_runs - 64 * wave : 64;
    const WavePre pre = wave_pre(e, usable, D0, D1, nl > 0 ? nl : 0, lane);
    if (lane == 0 && wave < n_waves) {                        // what the wave before needs to pass over this one
        sh_wok[wave] = pre.ok ? 1u : 0u; sh_we0[wave] = pre.e0; sh_wskip[wave] = 0u;
    }
    if (pre.ok && lane == nl - 1) { sh_wT0[wave] = pre.I0; sh_wT1[wave] = pre.I1; }
    if (tid == 0) { sh_next[0] = 0; kpp_stamp(a.stamps, 3); }
    __syncthreads();
    // lane v of every wave: what is needed to pass over wave v
    const uint32_t r_ok = lane < n_waves ? sh_wok[lane] : 0u, r_e0 = lane < n_waves ? sh_we0[lane] : 0u;
    const uint32_t r_T0 = lane < n_waves ? sh_wT0[lane] : 0u, r_T1 = lane < n_waves ? sh_wT1[lane] : 0u;
    const int fetch_it = EC == 0 ? 1 : 2;                     // (a block per chunk -- 6250-element chunks on an 8-core host -- takes ~10 us)
    for (int it = 0;; ++it) {                                 // (sh_next is double-buffered: a turn writes the word the NEXT iteration reads)
        const int nx = (int)sh_next[it & 1];
        if (nx >= n_waves) break;
        if (nx == wave) {
            float c;
            if (wave == 0) {                                  // the first 64 runs come from the head block
                unsigned long long v = 0, vc = 0;
                bool okw = true;
                if (lane < nl) { okw = kpp_wait(a.head + lane, round, v); sh_table[lane] = kpp_float((uint32_t)v); }
                if (lane == 0) okw = kpp_wait(a.head + 64, round, vc) && okw;
                if (__ballot(!okw)) sh_fail = 4;
                c = kpp_float(readlane_u32((uint32_t)vc, 0));
            } else {
                c = wave_turn<KPP_E>(sh_c, pre, xv, e, usable, D0, D1, nl, sh_table + 64 * wave, lane);
            }
            uint32_t cb = (uint32_t)__builtin_amdgcn_readfirstlane((int)kpp_bits(c));    // scalar from here on
            int v = __builtin_amdgcn_readfirstlane(wave) + 1;
            for (; v < n_waves; ++v) {                        // the waves after this one that the arriving value agrees with
                const uint32_t ec = cb >> 23;
                if (!(readlane_u32(r_ok, v) && readlane_u32(r_e0, v) == ec)) break;
                const uint32_t C = (cb & 0x7FFFFFu) | 0x800000u;
                const uint32_t T0 = readlane_u32(r_T0, v), T1 = readlane_u32(r_T1, v);
                const uint32_t Cend = C + ((C & 1u) ? T1 : T0);
                if (Cend >= (1u << 24)) break;
                if (lane == 0) { sh_wstart[v] = kpp_float(cb); sh_wskip[v] = 1u; }
                cb = (ec << 23) | (Cend & 0x7FFFFFu);
            }
            if (lane == 0) { sh_c = kpp_float(cb); sh_next[(it + 1) & 1] = (uint32_t)v; kpp_stamp(a.stamps, 20 + wave); }
        } else if (it == fetch_it) {
            // The waves that wait for this turn fetch the worker chunks' sums meanwhile (their blocks have finished by now; the turn
            // takes about as long as the fetch): the 960 waiting threads share the chunks
            const uint32_t rank = (int)wave < nx ? t : t - 64u;                 // 0 .. 959 over the waiting threads
            for (uint32_t j = rank; j < jobs; j += 960) {
                unsigned long long v = 0;
                if (!kpp_wait(a.chunk_sum + j, round, v)) sh_fail = 4;
                sh_cs[j] = kpp_float((uint32_t)v);
            }
            if (rank == 0) sh_cs_done = 1u;
        } else if (it == fetch_it + 1 && sh_cs_done && jobs && wave == (nx == 15 ? 14 : 15)) {
            const float c = join_chunk_sums(sh_cs, jobs, lane);     // ... and the last wave joins them
            if (lane == 0) { sh_total = c; sh_total_done = 1u; }
        }
        __syncthreads();
    }
    if (wave < n_waves && sh_wskip[wave]) {                   // a wave that was passed over: its run starts from the value it was passed with
        const uint32_t cb = kpp_bits(sh_wstart[wave]), ec = cb >> 23;
        const uint32_t C = (cb & 0x7FFFFFu) | 0x800000u;
        const uint32_t Iv = (C & 1u) ? pre.I1 : pre.I0;
        const uint32_t below = lane_below(Iv);
        if (lane < nl) sh_table[64 * wave + lane] = kpp_float((ec << 23) | ((lane == 0 ? C : C + below) & 0x7FFFFFu));
    }
    __syncthreads();
    if (tid == 0) sh_table[n_runs] = sh_c;
    if (tid == 0) kpp_stamp(a.stamps, 4);
    // total (:356-370): the chunk sums joined in ascending chunk order (one chunk: its sum is the chain's last value)
    if (!sh_cs_done && t < jobs) {                            // (a chain of fewer than two turns)
        unsigned long long v = 0;
        if (!kpp_wait(a.chunk_sum + t, round, v)) sh_fail = 4;
        sh_cs[t] = kpp_float((uint32_t)v);
    }
    __syncthreads();
    float total = sh_table[n_runs];
    if (jobs && sh_total_done) {
        total = sh_total;
    } else if (jobs) {
        if (wave == 0) {
            const float c = join_chunk_sums(sh_cs, jobs, lane);
            if (lane == 0) sh_c = c;
        }
        __syncthreads();
        total = sh_c;
    }
    if (tid == 0) kpp_stamp(a.stamps, 5);
    uint32_t fail = sh_fail;
    if (!fail && !(total > 0.0f && total < INFINITY)) fail = 1;   // :372 / :385 (the host draws range_usize), or a non-finite sum
    const float thr = a.u[round] * total;                          // :373
    if (tid == 0) a.state[3] = kpp_bits(total);                    // (read by pqv_kpp_pick only)
    // the run-start values never decrease: exactly one run's END is the first at or above the threshold
    if (!fail && t < n_runs && sh_table[t + 1] >= thr && (t == 0 || !(sh_table[t] >= thr))) sh_first = t;
    __syncthreads();
    const uint32_t first = sh_first;
    if (tid == 0) kpp_stamp(a.stamps, 6);
    if (!fail && first == 0xFFFFFFFFu) fail = 3;                   // no slot reaches the threshold: the host leaves the centroid unset
    if (fail) {
        if (tid == 0) { a.state[1] = round; a.state[2] = fail; a.state[0] = 1; }
        return;
    }
    if (t == first) {                                              // the walk inside that run (:375-383): its own lane has the elements
        float c = sh_table[first];
        uint32_t below_thr = 0;                                    // the partial sums never decrease: the slot = the number of them below
#pragma unroll
        for (int j = 0; j < KPP_E; ++j) {
            c = c + xv[j];
            below_thr += c >= thr ? 0u : 1u;
        }
        const uint32_t slot = first * KPP_E + below_thr;
        if (below_thr < (uint32_t)KPP_E && slot < a.n) a.picks[round] = slot;
        else { a.state[1] = round; a.state[2] = 5; a.state[0] = 1; }
        kpp_stamp(a.stamps, 7);
    }
}

hipError_t launch_kpp_pick(const KppPickArgs &a, hipStream_t s) {
    if (a.n == 0 || a.n > 1024u * KPP_E || a.n_chunks == 0 || a.n_chunks > 1024 || a.chunk == 0 || a.round == 0) return hipErrorInvalidValue;
    const uint32_t jobs = a.n_chunks > 1 ? a.n_chunks : 0;
    const uint32_t n_runs = (a.n + KPP_E - 1) / KPP_E, SB = (4 * n_runs + 63) / 64;
    const dim3 block(1024);
    if (jobs && a.chunk <= 512) hipLaunchKernelGGL((kpp_pick_kernel<0>), dim3(SB + 2 + (jobs + 15) / 16), block, 0, s, a);
    else if (a.chunk <= 1024u * 8) hipLaunchKernelGGL((kpp_pick_kernel<8>), dim3(SB + 2 + jobs), block, 0, s, a);
    else if (a.chunk <= 1024u * 16) hipLaunchKernelGGL((kpp_pick_kernel<16>), dim3(SB + 2 + jobs), block, 0, s, a);
    else hipLaunchKernelGGL((kpp_pick_kernel<KPP_E>), dim3(SB + 2 + jobs), block, 0, s, a);
    return hipGetLastError();
}


// an empty kernel the library launches at the first call for a device (and on a new stream): the runtime loads this unit's code
// object and sets up the stream's hardware queue then, not inside the first build or the first query
__global__ void touch_kpp_kernel() {}
hipError_t touch_kpp(hipStream_t s) {
    hipLaunchKernelGGL(touch_kpp_kernel, dim3(1), dim3(64), 0, s);
    return hipGetLastError();
}

}  // namespace pqv
