// rng.hpp -- the seeded choices of the index build, reproducing what the reference draws
// from `rand 0.8.5` (StdRng = ChaCha12, rand_chacha 0.3.1; seed_from_u64, rand_core 0.6.4):
//   src/ivf/index.rs:231-232  sample_embeddings: StdRng::seed_from_u64(seed) + index::sample
//   src/ivf/index.rs:327,337  k_means: a fresh StdRng(seed) + index::sample for the init subset
//   src/ivf/index.rs:340,385  gen_range(0..len)     (usize)
//   src/ivf/index.rs:373      gen_range(0.0..1.0)   (f32)
// The crates are not vendored in the reference; this follows their published algorithms
// (SURVEY.md App. A).  Host-side only: the draws are a few hundred KB of integer work.
#pragma once
#include <cstdint>
#include <cstring>
#include <unordered_set>
#include <vector>

namespace pqv {

class StdRng {
public:
    static StdRng seed_from_u64(uint64_t state) {
        // PCG32 expands the u64 into the 32-byte ChaCha key
        StdRng r;
        for (int w = 0; w < 8; ++w) {
            state = state * 6364136223846793005ULL + 11634580027462260723ULL;
            const uint32_t xorshifted = static_cast<uint32_t>(((state >> 18) ^ state) >> 27);
            const uint32_t rot = static_cast<uint32_t>(state >> 59);
            r.key_[w] = (xorshifted >> rot) | (xorshifted << ((32 - rot) & 31));
        }
        return r;
    }

    // SeedableRng::from_seed: the 32 bytes are the ChaCha key, little-endian words
    static StdRng from_seed(const uint8_t seed[32]) {
        StdRng r;
        for (int w = 0; w < 8; ++w)
            r.key_[w] = static_cast<uint32_t>(seed[4 * w]) | (static_cast<uint32_t>(seed[4 * w + 1]) << 8) |
                        (static_cast<uint32_t>(seed[4 * w + 2]) << 16) | (static_cast<uint32_t>(seed[4 * w + 3]) << 24);
        return r;
    }

    uint32_t next_u32() {
        if (pos_ >= kBuf) { refill(); pos_ = 0; }
        return buf_[pos_++];
    }

    uint64_t next_u64() {
        // BlockRng: two consecutive words, low first; a lone last word pairs with word 0
        // of the next buffer
        if (pos_ < kBuf - 1) {
            const uint64_t v = (static_cast<uint64_t>(buf_[pos_ + 1]) << 32) | buf_[pos_];
            pos_ += 2;
            return v;
        }
        if (pos_ >= kBuf) {
            refill();
            pos_ = 2;
            return (static_cast<uint64_t>(buf_[1]) << 32) | buf_[0];
        }
        const uint64_t lo = buf_[kBuf - 1];
        refill();
        pos_ = 1;
        return (static_cast<uint64_t>(buf_[0]) << 32) | lo;
    }

    // gen_range(low..high) for usize
    uint64_t range_usize(uint64_t low, uint64_t high) {
        const uint64_t span = high - low;
        if (span == 0) return next_u64();
        const uint64_t zone = (span << __builtin_clzll(span)) - 1;
        for (;;) {
            const unsigned __int128 wide = static_cast<unsigned __int128>(next_u64()) * span;
            if (static_cast<uint64_t>(wide) <= zone) return low + static_cast<uint64_t>(wide >> 64);
        }
    }

    // gen_range(low..=high) for u32
    uint32_t range_u32_inclusive(uint32_t low, uint32_t high) {
        const uint32_t span = high - low + 1u;
        if (span == 0) return next_u32();
        const uint32_t zone = (span << __builtin_clz(span)) - 1u;
        for (;;) {
            const uint64_t wide = static_cast<uint64_t>(next_u32()) * span;
            if (static_cast<uint32_t>(wide) <= zone) return low + static_cast<uint32_t>(wide >> 32);
        }
    }

    // gen_range(0.0..1.0) as f32
    float unit_f32() {
        for (;;) {
            const uint32_t bits = (next_u32() >> 9) | 0x3F800000u;
            float one_two;
            std::memcpy(&one_two, &bits, 4);
            const float res = (one_two - 1.0f) * 1.0f + 0.0f;
            if (res < 1.0f) return res;
        }
    }

private:
    static constexpr uint32_t kBuf = 64;  // four ChaCha blocks per refill
    uint32_t key_[8] = {0};
    uint64_t counter_ = 0;
    uint32_t buf_[kBuf];
    uint32_t pos_ = kBuf;

    static inline uint32_t rol(uint32_t v, int n) { return (v << n) | (v >> (32 - n)); }
    static inline void quarter(uint32_t *x, int a, int b, int c, int d) {
        x[a] += x[b]; x[d] = rol(x[d] ^ x[a], 16);
        x[c] += x[d]; x[b] = rol(x[b] ^ x[c], 12);
        x[a] += x[b]; x[d] = rol(x[d] ^ x[a], 8);
        x[c] += x[d]; x[b] = rol(x[b] ^ x[c], 7);
    }
    void block(uint64_t ctr, uint32_t *out) const {
        uint32_t in[16] = {0x61707865u, 0x3320646eu, 0x79622d32u, 0x6b206574u};
        for (int i = 0; i < 8; ++i) in[4 + i] = key_[i];
        in[12] = static_cast<uint32_t>(ctr);
        in[13] = static_cast<uint32_t>(ctr >> 32);
        in[14] = 0; in[15] = 0;
        uint32_t x[16];
        std::memcpy(x, in, sizeof x);
        for (int dr = 0; dr < 6; ++dr) {  // 12 rounds
            quarter(x, 0, 4, 8, 12); quarter(x, 1, 5, 9, 13);
            quarter(x, 2, 6, 10, 14); quarter(x, 3, 7, 11, 15);
            quarter(x, 0, 5, 10, 15); quarter(x, 1, 6, 11, 12);
            quarter(x, 2, 7, 8, 13); quarter(x, 3, 4, 9, 14);
        }
        for (int i = 0; i < 16; ++i) out[i] = x[i] + in[i];
    }
    void refill() {
        for (int b = 0; b < 4; ++b) block(counter_ + b, buf_ + 16 * b);
        counter_ += 4;
    }
};

// rand::seq::index::sample(rng, length, amount) -> indices in draw order
inline std::vector<uint64_t> index_sample(StdRng &rng, uint64_t length, uint64_t amount) {
    std::vector<uint64_t> out;
    out.reserve(amount);
    if (length > 0xFFFFFFFFull) {  // sample_rejection::<usize>
        std::unordered_set<uint64_t> seen;
        seen.reserve(amount * 2);
        const uint64_t reject = (~0ull - length + 1) % length;
        const uint64_t zone = ~0ull - reject;
        auto draw = [&]() {
            for (;;) {
                const unsigned __int128 wide = static_cast<unsigned __int128>(rng.next_u64()) * length;
                if (static_cast<uint64_t>(wide) <= zone) return static_cast<uint64_t>(wide >> 64);
            }
        };
        for (uint64_t i = 0; i < amount; ++i) {
            uint64_t p = draw();
            while (!seen.insert(p).second) p = draw();
            out.push_back(p);
        }
        return out;
    }
    const uint32_t len = static_cast<uint32_t>(length), amt = static_cast<uint32_t>(amount);
    const int big = len < 500000u ? 0 : 1;
    enum { FLOYD, INPLACE, REJECTION } algo;
    if (amt < 163) {
        const float c0[2] = {1.6f, 8.0f / 45.0f}, c1[2] = {10.0f, 70.0f / 9.0f};
        const float a = static_cast<float>(amt);
        const float m4 = c0[big] * a;
        algo = (amt > 11 && static_cast<float>(len) < (c1[big] + m4) * a) ? INPLACE : FLOYD;
    } else {
        const float c[2] = {270.0f, 330.0f / 9.0f};
        algo = (static_cast<float>(len) < c[big] * static_cast<float>(amt)) ? INPLACE : REJECTION;
    }
    if (algo == INPLACE) {
        std::vector<uint32_t> idx(len);
        for (uint32_t i = 0; i < len; ++i) idx[i] = i;
        for (uint32_t i = 0; i < amt; ++i) std::swap(idx[i], idx[rng.range_u32_inclusive(i, len - 1)]);
        for (uint32_t i = 0; i < amt; ++i) out.push_back(idx[i]);
    } else if (algo == FLOYD) {
        const bool insert_in_place = amt < 50;
        std::vector<uint32_t> idx;
        idx.reserve(amt);
        for (uint32_t j = len - amt; j < len; ++j) {
            const uint32_t t = rng.range_u32_inclusive(0, j);
            size_t at = idx.size();
            for (size_t p = 0; p < idx.size(); ++p) if (idx[p] == t) { at = p; break; }
            if (at < idx.size()) {
                if (insert_in_place) idx.insert(idx.begin() + at, j); else idx.push_back(j);
            } else {
                idx.push_back(t);
            }
        }
        if (!insert_in_place)
            for (uint32_t i = amt - 1; i >= 1; --i) std::swap(idx[i], idx[rng.range_u32_inclusive(0, i)]);
        for (uint32_t v : idx) out.push_back(v);
    } else {
        std::unordered_set<uint32_t> seen;
        const uint32_t reject = (0xFFFFFFFFu - len + 1u) % len;
        const uint32_t zone = 0xFFFFFFFFu - reject;
        auto draw = [&]() {
            for (;;) {
                const uint64_t wide = static_cast<uint64_t>(rng.next_u32()) * len;
                if (static_cast<uint32_t>(wide) <= zone) return static_cast<uint32_t>(wide >> 32);
            }
        };
        if (len <= (1u << 28)) {
            // membership as a bit per index (<= 32 MB): the draws and their order are the hash set's, the test is one load
            std::vector<uint64_t> bits((static_cast<size_t>(len) + 63) / 64, 0);
            auto test_and_set = [&](uint32_t p) {
                uint64_t &w = bits[p >> 6];
                const uint64_t b = 1ull << (p & 63);
                const bool fresh = (w & b) == 0;
                w |= b;
                return fresh;
            };
            for (uint32_t i = 0; i < amt; ++i) {
                uint32_t p = draw();
                while (!test_and_set(p)) p = draw();
                out.push_back(p);
            }
            return out;
        }
        seen.reserve(static_cast<size_t>(amt) * 2);
        for (uint32_t i = 0; i < amt; ++i) {
            uint32_t p = draw();
            while (!seen.insert(p).second) p = draw();
            out.push_back(p);
        }
    }
    return out;
}

}  // namespace pqv
