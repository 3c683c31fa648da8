/*
 * pqv_oracle.c -- CPU ORACLE (TEST INFRASTRUCTURE, NOT PRODUCT CODE).  See pqv_oracle.h
 * for scope, the reference files restated, and the pinning status ("parity unpinned" for
 * every seeded choice: the RNG crates are not in the reference tree).
 *
 * Build: gcc -O2 -ffp-contract=off (never -ffast-math): rustc does not contract
 * mul+add into FMA and evaluates f32 expressions left to right, so must we.
 */
#define _GNU_SOURCE
#include "pqv_oracle.h"

#include <math.h>
#include <pthread.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <unistd.h>

#define PQO_ERRLEN 128

static void set_err(char *err, const char *msg) {
    if (err) {
        strncpy(err, msg, PQO_ERRLEN - 1);
        err[PQO_ERRLEN - 1] = 0;
    }
}

/* ===================================================================================
 * Distances
 * =================================================================================== */

/* src/ivf/index.rs:461-480 */
float pqo_squared_l2_ref4(const float *a, const float *b, size_t len) {
    float sum = 0.0f;
    size_t i = 0;
    while (i + 4 <= len) {
        float d0 = a[i] - b[i];
        float d1 = a[i + 1] - b[i + 1];
        float d2 = a[i + 2] - b[i + 2];
        float d3 = a[i + 3] - b[i + 3];
        /* Rust: sum += d0*d0 + d1*d1 + d2*d2 + d3*d3  (left-assoc, then added to sum) */
        float t = d0 * d0 + d1 * d1;
        t = t + d2 * d2;
        t = t + d3 * d3;
        sum = sum + t;
        i += 4;
    }
    while (i < len) {
        float d = a[i] - b[i];
        sum = sum + d * d;
        i += 1;
    }
    return sum;
}

/* src/df_vector/exec.rs:529-533 */
float pqo_squared_l2_seq(const float *values, const float *query, size_t len) {
    float dist = 0.0f;
    for (size_t i = 0; i < len; ++i) {
        float diff = values[i] - query[i];
        dist = dist + diff * diff;
    }
    return dist;
}

/* src/df_vector/exec.rs:538-545 */
float pqo_squared_l2_seq_f64(const double *values, const float *query, size_t len) {
    float dist = 0.0f;
    for (size_t i = 0; i < len; ++i) {
        float diff = (float)values[i] - query[i];
        dist = dist + diff * diff;
    }
    return dist;
}

/* ===================================================================================
 * rand 0.8.5 / rand_chacha 0.3.1 / rand_core 0.6.4 restatement
 * =================================================================================== */

static inline uint32_t rotl32(uint32_t x, int n) { return (x << n) | (x >> (32 - n)); }
static inline uint32_t rotr32(uint32_t x, unsigned n) {
    n &= 31u;
    return n ? ((x >> n) | (x << (32 - n))) : x;
}

#define QR(a, b, c, d)                                                                  \
    do {                                                                                \
        a += b; d ^= a; d = rotl32(d, 16);                                              \
        c += d; b ^= c; b = rotl32(b, 12);                                              \
        a += b; d ^= a; d = rotl32(d, 8);                                               \
        c += d; b ^= c; b = rotl32(b, 7);                                               \
    } while (0)

void pqo_chacha_block(const uint32_t key[8], uint64_t counter, uint64_t stream, int rounds,
                      uint32_t out[16]) {
    uint32_t s[16], x[16];
    s[0] = 0x61707865u; s[1] = 0x3320646eu; s[2] = 0x79622d32u; s[3] = 0x6b206574u;
    for (int i = 0; i < 8; ++i) s[4 + i] = key[i];
    s[12] = (uint32_t)counter; s[13] = (uint32_t)(counter >> 32);
    s[14] = (uint32_t)stream;  s[15] = (uint32_t)(stream >> 32);
    memcpy(x, s, sizeof x);
    for (int r = 0; r < rounds; r += 2) {
        QR(x[0], x[4], x[8], x[12]);  QR(x[1], x[5], x[9], x[13]);
        QR(x[2], x[6], x[10], x[14]); QR(x[3], x[7], x[11], x[15]);
        QR(x[0], x[5], x[10], x[15]); QR(x[1], x[6], x[11], x[12]);
        QR(x[2], x[7], x[8], x[13]);  QR(x[3], x[4], x[9], x[14]);
    }
    for (int i = 0; i < 16; ++i) out[i] = x[i] + s[i];
}

void pqo_rng_from_seed(pqo_rng *rng, const uint8_t seed[32]) {
    for (int i = 0; i < 8; ++i)
        rng->key[i] = (uint32_t)seed[4 * i] | ((uint32_t)seed[4 * i + 1] << 8) |
                      ((uint32_t)seed[4 * i + 2] << 16) | ((uint32_t)seed[4 * i + 3] << 24);
    rng->counter = 0;
    rng->index = 64; /* BlockRng starts empty */
}

/* rand_core 0.6.4 SeedableRng::seed_from_u64: PCG32 fills the 32-byte seed. */
void pqo_rng_seed_from_u64(pqo_rng *rng, uint64_t state) {
    const uint64_t MUL = 6364136223846793005ULL, INC = 11634580027462260723ULL;
    uint8_t seed[32];
    for (int c = 0; c < 8; ++c) {
        state = state * MUL + INC;
        uint32_t xorshifted = (uint32_t)(((state >> 18) ^ state) >> 27);
        uint32_t rot = (uint32_t)(state >> 59);
        uint32_t x = rotr32(xorshifted, rot);
        seed[4 * c] = (uint8_t)x; seed[4 * c + 1] = (uint8_t)(x >> 8);
        seed[4 * c + 2] = (uint8_t)(x >> 16); seed[4 * c + 3] = (uint8_t)(x >> 24);
    }
    pqo_rng_from_seed(rng, seed);
}

/* StdRng = ChaCha12; one refill = 4 consecutive blocks = 64 words. */
static void rng_refill(pqo_rng *rng) {
    for (int b = 0; b < 4; ++b)
        pqo_chacha_block(rng->key, rng->counter + (uint64_t)b, 0, 12, rng->buf + 16 * b);
    rng->counter += 4;
}

uint32_t pqo_rng_next_u32(pqo_rng *rng) {
    if (rng->index >= 64) { rng_refill(rng); rng->index = 0; }
    return rng->buf[rng->index++];
}

/* rand_core BlockRng::next_u64 */
uint64_t pqo_rng_next_u64(pqo_rng *rng) {
    uint32_t index = rng->index;
    if (index < 63) {
        rng->index += 2;
        return ((uint64_t)rng->buf[index + 1] << 32) | rng->buf[index];
    } else if (index >= 64) {
        rng_refill(rng);
        rng->index = 2;
        return ((uint64_t)rng->buf[1] << 32) | rng->buf[0];
    } else {
        uint64_t x = rng->buf[63];
        rng_refill(rng);
        rng->index = 1;
        uint64_t y = rng->buf[0];
        return (y << 32) | x;
    }
}

/* UniformInt<usize>::sample_single -> sample_single_inclusive(low, high-1) */
uint64_t pqo_rng_gen_range_usize(pqo_rng *rng, uint64_t low, uint64_t high) {
    uint64_t range = high - low; /* (high-1) - low + 1 */
    if (range == 0) return pqo_rng_next_u64(rng);
    uint64_t zone = (range << __builtin_clzll(range)) - 1;
    for (;;) {
        uint64_t v = pqo_rng_next_u64(rng);
        unsigned __int128 m = (unsigned __int128)v * range;
        uint64_t hi = (uint64_t)(m >> 64), lo = (uint64_t)m;
        if (lo <= zone) return low + hi;
    }
}

/* UniformInt<u32>::sample_single_inclusive */
uint32_t pqo_rng_gen_range_u32_incl(pqo_rng *rng, uint32_t low, uint32_t high) {
    uint32_t range = high - low + 1u;
    if (range == 0) return pqo_rng_next_u32(rng);
    uint32_t zone = (range << __builtin_clz(range)) - 1u;
    for (;;) {
        uint32_t v = pqo_rng_next_u32(rng);
        uint64_t m = (uint64_t)v * range;
        uint32_t hi = (uint32_t)(m >> 32), lo = (uint32_t)m;
        if (lo <= zone) return low + hi;
    }
}

/* UniformFloat<f32>::sample_single(0.0, 1.0) */
float pqo_rng_gen_range_f32_unit(pqo_rng *rng) {
    const float low = 0.0f, high = 1.0f;
    float scale = high - low;
    for (;;) {
        uint32_t bits = (pqo_rng_next_u32(rng) >> 9) | 0x3F800000u;
        float value1_2;
        memcpy(&value1_2, &bits, 4);
        float value0_1 = value1_2 - 1.0f;
        float res = value0_1 * scale + low;
        if (res < high) return res;
    }
}

/* Standard: Distribution<f32> */
float pqo_rng_gen_f32(pqo_rng *rng) {
    uint32_t v = pqo_rng_next_u32(rng) >> 8;
    return (float)v * (1.0f / 16777216.0f);
}

/* Uniform::new(0u32, length).sample */
typedef struct { uint32_t range, zone; } uniform_u32;
static uniform_u32 uniform_u32_new(uint32_t length) {
    uniform_u32 u;
    u.range = length;
    uint32_t ints_to_reject = length ? (uint32_t)((0xFFFFFFFFu - length + 1u) % length) : 0;
    u.zone = 0xFFFFFFFFu - ints_to_reject;
    return u;
}
static uint32_t uniform_u32_sample(const uniform_u32 *u, pqo_rng *rng) {
    if (u->range == 0) return pqo_rng_next_u32(rng);
    for (;;) {
        uint32_t v = pqo_rng_next_u32(rng);
        uint64_t m = (uint64_t)v * u->range;
        if ((uint32_t)m <= u->zone) return (uint32_t)(m >> 32);
    }
}
typedef struct { uint64_t range, zone; } uniform_u64;
static uniform_u64 uniform_u64_new(uint64_t length) {
    uniform_u64 u;
    u.range = length;
    uint64_t rej = length ? ((UINT64_MAX - length + 1u) % length) : 0;
    u.zone = UINT64_MAX - rej;
    return u;
}
static uint64_t uniform_u64_sample(const uniform_u64 *u, pqo_rng *rng) {
    if (u->range == 0) return pqo_rng_next_u64(rng);
    for (;;) {
        uint64_t v = pqo_rng_next_u64(rng);
        unsigned __int128 m = (unsigned __int128)v * u->range;
        if ((uint64_t)m <= u->zone) return (uint64_t)(m >> 64);
    }
}

/* open-addressing set of u64 (membership only) for sample_rejection's HashSet */
typedef struct { uint64_t *slot; uint64_t mask; } u64set;
static int u64set_init(u64set *s, uint64_t expect) {
    uint64_t cap = 16;
    while (cap < expect * 2 + 2) cap <<= 1;
    s->slot = (uint64_t *)malloc(cap * sizeof(uint64_t));
    if (!s->slot) return -1;
    memset(s->slot, 0xFF, cap * sizeof(uint64_t));
    s->mask = cap - 1;
    return 0;
}
/* returns 1 if newly inserted, 0 if present */
static int u64set_insert(u64set *s, uint64_t v) {
    uint64_t h = v * 0x9E3779B97F4A7C15ULL;
    uint64_t i = (h >> 17) & s->mask;
    for (;;) {
        if (s->slot[i] == UINT64_MAX) { s->slot[i] = v; return 1; }
        if (s->slot[i] == v) return 0;
        i = (i + 1) & s->mask;
    }
}

/* rand::seq::index::sample */
int pqo_index_sample(pqo_rng *rng, uint64_t length, uint64_t amount, uint64_t *out,
                     int *branch) {
    if (amount > length) return -1;
    if (length > 0xFFFFFFFFull) {
        /* sample_rejection::<usize> */
        if (branch) *branch = 2;
        u64set set;
        if (u64set_init(&set, amount)) return -1;
        uniform_u64 distr = uniform_u64_new(length);
        for (uint64_t i = 0; i < amount; ++i) {
            uint64_t pos = uniform_u64_sample(&distr, rng);
            while (!u64set_insert(&set, pos)) pos = uniform_u64_sample(&distr, rng);
            out[i] = pos;
        }
        free(set.slot);
        return 0;
    }
    uint32_t amt = (uint32_t)amount, len = (uint32_t)length;
    int algo; /* 0 floyd, 1 inplace, 2 rejection */
    if (amt < 163) {
        static const float C[2][2] = {{1.6f, 8.0f / 45.0f}, {10.0f, 70.0f / 9.0f}};
        int j = (len < 500000u) ? 0 : 1;
        float amount_fp = (float)amt;
        float m4 = C[0][j] * amount_fp;
        if (amt > 11 && (float)len < (C[1][j] + m4) * amount_fp) algo = 1; else algo = 0;
    } else {
        static const float C[2] = {270.0f, 330.0f / 9.0f};
        int j = (len < 500000u) ? 0 : 1;
        if ((float)len < C[j] * (float)amt) algo = 1; else algo = 2;
    }
    if (branch) *branch = algo;

    if (algo == 1) { /* sample_inplace */
        uint32_t *indices = (uint32_t *)malloc((size_t)len * sizeof(uint32_t));
        if (!indices && len) return -1;
        for (uint32_t i = 0; i < len; ++i) indices[i] = i;
        for (uint32_t i = 0; i < amt; ++i) {
            uint32_t j = pqo_rng_gen_range_u32_incl(rng, i, len - 1u); /* gen_range(i..length) */
            uint32_t t = indices[i]; indices[i] = indices[j]; indices[j] = t;
        }
        for (uint32_t i = 0; i < amt; ++i) out[i] = indices[i];
        free(indices);
        return 0;
    }
    if (algo == 0) { /* sample_floyd */
        int floyd_shuffle = amt < 50;
        uint32_t *indices = (uint32_t *)malloc(((size_t)amt + 1) * sizeof(uint32_t));
        if (!indices) return -1;
        uint32_t cnt = 0;
        for (uint32_t j = len - amt; j < len; ++j) {
            uint32_t t = pqo_rng_gen_range_u32_incl(rng, 0, j);
            uint32_t pos = cnt;
            for (uint32_t p = 0; p < cnt; ++p) if (indices[p] == t) { pos = p; break; }
            if (floyd_shuffle) {
                if (pos < cnt) { /* indices.insert(pos, j) */
                    memmove(indices + pos + 1, indices + pos, (cnt - pos) * sizeof(uint32_t));
                    indices[pos] = j; cnt++;
                    continue;
                }
            } else if (pos < cnt) {
                indices[cnt++] = j;
                continue;
            }
            indices[cnt++] = t;
        }
        if (!floyd_shuffle) {
            for (uint32_t i = amt; i-- > 1;) { /* (1..amount).rev() */
                uint32_t r = pqo_rng_gen_range_u32_incl(rng, 0, i);
                uint32_t t = indices[i]; indices[i] = indices[r]; indices[r] = t;
            }
        }
        for (uint32_t i = 0; i < amt; ++i) out[i] = indices[i];
        free(indices);
        return 0;
    }
    /* sample_rejection::<u32> */
    {
        u64set set;
        if (u64set_init(&set, amt)) return -1;
        uniform_u32 distr = uniform_u32_new(len);
        for (uint32_t i = 0; i < amt; ++i) {
            uint32_t pos = uniform_u32_sample(&distr, rng);
            while (!u64set_insert(&set, pos)) pos = uniform_u32_sample(&distr, rng);
            out[i] = pos;
        }
        free(set.slot);
    }
    return 0;
}

/* ===================================================================================
 * Scoped-thread chunking (src/ivf/index.rs:259-320)
 * =================================================================================== */

static uint32_t hw_workers(void) {
    long n = sysconf(_SC_NPROCESSORS_ONLN);
    return n > 0 ? (uint32_t)n : 1u;
}

/* worker_count(len) with an explicit parallelism */
static uint64_t worker_count(uint64_t len, uint32_t workers) {
    uint64_t w = workers;
    if (w > len) w = len;
    if (w < 1) w = 1;
    return w;
}

typedef void (*chunk_fn)(uint64_t chunk_idx, uint64_t start, uint64_t end, void *ctx);
typedef struct { chunk_fn fn; uint64_t idx, start, end; void *ctx; } chunk_job;
static void *chunk_tramp(void *p) {
    chunk_job *j = (chunk_job *)p;
    j->fn(j->idx, j->start, j->end, j->ctx);
    return NULL;
}

/* Runs fn over chunks [start,end) of size ceil(len/workers), one OS thread per chunk
 * (as thread::scope does).  Returns the number of chunks. */
static uint64_t parallel_chunks(uint64_t len, uint32_t workers, chunk_fn fn, void *ctx) {
    if (len == 0) return 0;
    uint64_t w = worker_count(len, workers);
    uint64_t chunk = (len + w - 1) / w;
    uint64_t nchunks = (len + chunk - 1) / chunk;
    if (nchunks == 1) { fn(0, 0, len, ctx); return 1; }
    chunk_job *jobs = (chunk_job *)malloc(nchunks * sizeof(chunk_job));
    pthread_t *th = (pthread_t *)malloc(nchunks * sizeof(pthread_t));
    for (uint64_t c = 0; c < nchunks; ++c) {
        uint64_t s = c * chunk, e = s + chunk < len ? s + chunk : len;
        jobs[c].fn = fn; jobs[c].idx = c; jobs[c].start = s; jobs[c].end = e; jobs[c].ctx = ctx;
        if (pthread_create(&th[c], NULL, chunk_tramp, &jobs[c]) != 0) {
            chunk_tramp(&jobs[c]);
            th[c] = 0;
        }
    }
    for (uint64_t c = 0; c < nchunks; ++c) if (th[c]) pthread_join(th[c], NULL);
    free(jobs); free(th);
    return nchunks;
}

/* ===================================================================================
 * k-means (src/ivf/index.rs:323-457)
 * =================================================================================== */

typedef struct {
    const float *data; uint32_t dim; const uint64_t *init_indices;
    const float *centroid; float *min_d; float *partial; int first;
} pp_ctx;

static void pp_chunk(uint64_t cidx, uint64_t start, uint64_t end, void *p) {
    pp_ctx *c = (pp_ctx *)p;
    float local_sum = 0.0f;
    for (uint64_t s = start; s < end; ++s) {
        const float *vec = c->data + c->init_indices[s] * (uint64_t)c->dim;
        float dist = pqo_squared_l2_ref4(vec, c->centroid, c->dim);
        if (c->first) {
            c->min_d[s] = dist;                       /* :350 */
        } else {
            if (dist < c->min_d[s]) c->min_d[s] = dist; /* :363-365 */
            local_sum = local_sum + c->min_d[s];        /* :366 */
        }
    }
    if (!c->first) c->partial[cidx] = local_sum;
}

typedef struct {
    const float *data; uint32_t dim; uint32_t k; const float *centroids;
    uint64_t *assign; uint64_t *changed; uint64_t *sizes; /* per chunk: [nchunks][k] */
} assign_ctx;

static void assign_chunk(uint64_t cidx, uint64_t start, uint64_t end, void *p) {
    assign_ctx *c = (assign_ctx *)p;
    uint64_t local_changed = 0;
    uint64_t *local_sizes = c->sizes + cidx * (uint64_t)c->k;
    for (uint64_t row = start; row < end; ++row) {
        const float *vec = c->data + row * (uint64_t)c->dim;
        uint64_t best = 0;
        float best_dist = INFINITY;
        for (uint32_t j = 0; j < c->k; ++j) {
            float dist = pqo_squared_l2_ref4(vec, c->centroids + (uint64_t)j * c->dim, c->dim);
            if (dist < best_dist) { best_dist = dist; best = j; }
        }
        if (c->assign[row] != best) local_changed++;
        c->assign[row] = best;
        local_sizes[best]++;
    }
    c->changed[cidx] = local_changed;
}

int pqo_kmeans(const float *data, uint64_t n, uint32_t dim, uint32_t k, uint32_t max_iters,
               uint64_t seed, uint32_t workers, float *centroids, uint64_t *assignments,
               uint32_t *iters_run) {
    if (workers == 0) workers = hw_workers();
    pqo_rng rng;
    pqo_rng_seed_from_u64(&rng, seed);                                   /* :327 */
    memset(centroids, 0, (size_t)k * dim * sizeof(float));               /* :330 */

    uint64_t init_n = n < 50000 ? n : 50000;                             /* :332 */
    if (init_n < k) init_n = k;
    uint64_t *init_indices = (uint64_t *)malloc(init_n * sizeof(uint64_t));
    if (init_n == n) {
        for (uint64_t i = 0; i < n; ++i) init_indices[i] = i;            /* :334 */
    } else {
        if (pqo_index_sample(&rng, n, init_n, init_indices, NULL)) { free(init_indices); return -1; }
    }

    uint64_t first_choice = pqo_rng_gen_range_usize(&rng, 0, init_n);    /* :340 */
    uint64_t first_idx = init_indices[first_choice];
    memcpy(centroids, data + first_idx * dim, dim * sizeof(float));      /* :342 */

    float *min_d = (float *)calloc(init_n, sizeof(float));               /* :344 */
    uint64_t wmax = worker_count(init_n, workers);
    float *partial = (float *)calloc(wmax, sizeof(float));
    pp_ctx pc = {data, dim, init_indices, centroids, min_d, partial, 1};
    parallel_chunks(init_n, workers, pp_chunk, &pc);                     /* :345-352 */

    for (uint32_t i = 1; i < k; ++i) {                                   /* :354 */
        pc.centroid = centroids + (uint64_t)(i - 1) * dim;
        pc.first = 0;
        uint64_t nch = parallel_chunks(init_n, workers, pp_chunk, &pc);  /* :356-369 */
        float total = 0.0f;
        for (uint64_t c = 0; c < nch; ++c) total = total + partial[c];   /* :370 */
        if (total > 0.0f) {
            float threshold = pqo_rng_gen_range_f32_unit(&rng) * total;  /* :373 */
            float cumsum = 0.0f;
            for (uint64_t slot = 0; slot < init_n; ++slot) {             /* :375-383 */
                cumsum = cumsum + min_d[slot];
                if (cumsum >= threshold) {
                    memcpy(centroids + (uint64_t)i * dim, data + init_indices[slot] * dim,
                           dim * sizeof(float));
                    break;
                }
            }
            /* no slot reached the threshold => centroid i stays all-zero (as in Rust) */
        } else {
            uint64_t choice = pqo_rng_gen_range_usize(&rng, 0, init_n);  /* :385 */
            memcpy(centroids + (uint64_t)i * dim, data + init_indices[choice] * dim,
                   dim * sizeof(float));
        }
    }
    free(min_d); free(partial); free(init_indices);

    for (uint64_t i = 0; i < n; ++i) assignments[i] = 0;                 /* :392 */
    uint64_t *cluster_sizes = (uint64_t *)calloc(k, sizeof(uint64_t));
    uint64_t aw = worker_count(n, workers);
    uint64_t *chunk_sizes = (uint64_t *)malloc(aw * (uint64_t)k * sizeof(uint64_t));
    uint64_t *chunk_changed = (uint64_t *)malloc(aw * sizeof(uint64_t));
    uint32_t iters = 0;
    for (uint32_t iter = 0; iter < max_iters; ++iter) {                  /* :395 */
        memset(cluster_sizes, 0, k * sizeof(uint64_t));
        memset(chunk_sizes, 0, aw * (uint64_t)k * sizeof(uint64_t));
        assign_ctx ac = {data, dim, k, centroids, assignments, chunk_changed, chunk_sizes};
        uint64_t nch = parallel_chunks(n, workers, assign_chunk, &ac);   /* :398-424 */
        iters++;
        uint64_t changed = 0;
        for (uint64_t c = 0; c < nch; ++c) {
            changed += chunk_changed[c];
            for (uint32_t j = 0; j < k; ++j) cluster_sizes[j] += chunk_sizes[c * k + j];
        }
        if (changed == 0) break;                                         /* :432 */

        memset(centroids, 0, (size_t)k * dim * sizeof(float));           /* :436 */
        for (uint64_t i = 0; i < n; ++i) {                               /* :438-444 */
            float *cdst = centroids + assignments[i] * dim;
            const float *vec = data + i * dim;
            for (uint32_t j = 0; j < dim; ++j) cdst[j] = cdst[j] + vec[j];
        }
        for (uint32_t j = 0; j < k; ++j) {                               /* :446-453 */
            if (cluster_sizes[j] > 0) {
                float size = (float)cluster_sizes[j];
                for (uint32_t d = 0; d < dim; ++d)
                    centroids[(uint64_t)j * dim + d] = centroids[(uint64_t)j * dim + d] / size;
            }
        }
    }
    if (iters_run) *iters_run = iters;
    free(cluster_sizes); free(chunk_sizes); free(chunk_changed);
    return 0;
}

/* ===================================================================================
 * build_ivf_index (src/ivf/index.rs:152-214)
 * =================================================================================== */

typedef struct { const float *data; uint32_t dim, k; const float *centroids; uint32_t *cluster_of; } fa_ctx;

static void final_assign_chunk(uint64_t cidx, uint64_t start, uint64_t end, void *p) {
    (void)cidx;
    fa_ctx *c = (fa_ctx *)p;
    for (uint64_t row = start; row < end; ++row) {
        /* nearest_centroid :244-257 */
        const float *vec = c->data + row * (uint64_t)c->dim;
        uint32_t best = 0;
        float best_dist = INFINITY;
        for (uint32_t i = 0; i < c->k; ++i) {
            float dist = pqo_squared_l2_ref4(vec, c->centroids + (uint64_t)i * c->dim, c->dim);
            if (dist < best_dist) { best_dist = dist; best = i; }
        }
        c->cluster_of[row] = best;
    }
}

void pqo_index_free(pqo_index *idx) {
    if (!idx) return;
    free(idx->centroids); free(idx->list_off); free(idx->list_rows); free(idx);
}

int pqo_build_ivf_index(const float *data, uint64_t n, uint32_t dim, uint32_t n_clusters,
                        uint32_t max_iters, uint64_t seed, uint32_t workers,
                        pqo_index **out, char *err) {
    *out = NULL;
    if (dim == 0) { set_err(err, "Embedding dimension must be > 0"); return -1; }   /* mod.rs:59 */
    if (max_iters == 0) { set_err(err, "max_iters must be > 0"); return -1; }       /* parquet.rs:90 */
    if (n == 0) { set_err(err, "Cannot build IVF index with zero vectors"); return -1; } /* :158 */
    if (workers == 0) workers = hw_workers();
    uint64_t k = n_clusters;
    if (k == 0) k = (uint64_t)ceil(sqrt((double)n));                                /* :164 */
    if (k > n) { set_err(err, "n_clusters cannot exceed number of vectors"); return -1; } /* :169 */

    uint64_t sample_size = n / 20; if (sample_size < 1) sample_size = 1;            /* :172 */
    if (sample_size > 100000) sample_size = 100000;                                 /* :173 */
    if (sample_size < k) sample_size = k;                                           /* :174 */
    if (sample_size > n) sample_size = n;

    float *centroids = (float *)malloc((size_t)k * dim * sizeof(float));
    if (sample_size == n) {                                                         /* :182 */
        uint64_t *assign = (uint64_t *)malloc(n * sizeof(uint64_t));
        pqo_kmeans(data, n, dim, (uint32_t)k, max_iters, seed, workers, centroids, assign, NULL);
        free(assign);
    } else {
        /* sample_embeddings :222-242 */
        pqo_rng rng;
        pqo_rng_seed_from_u64(&rng, seed);
        uint64_t *indices = (uint64_t *)malloc(sample_size * sizeof(uint64_t));
        if (pqo_index_sample(&rng, n, sample_size, indices, NULL)) {
            free(indices); free(centroids); set_err(err, "sample failed"); return -1;
        }
        float *sample = (float *)malloc((size_t)sample_size * dim * sizeof(float));
        for (uint64_t i = 0; i < sample_size; ++i)
            memcpy(sample + i * dim, data + indices[i] * dim, dim * sizeof(float));
        free(indices);
        uint64_t *assign = (uint64_t *)malloc(sample_size * sizeof(uint64_t));
        pqo_kmeans(sample, sample_size, dim, (uint32_t)k, max_iters, seed, workers, centroids,
                   assign, NULL);
        free(assign); free(sample);
    }

    /* final assignment :189-206: per-chunk local lists appended in chunk order ==
     * ascending row ids per cluster == a stable counting sort by cluster. */
    uint32_t *cluster_of = (uint32_t *)malloc(n * sizeof(uint32_t));
    fa_ctx fc = {data, dim, (uint32_t)k, centroids, cluster_of};
    parallel_chunks(n, workers, final_assign_chunk, &fc);

    pqo_index *idx = (pqo_index *)calloc(1, sizeof(pqo_index));
    idx->dim = dim; idx->n_clusters = (uint32_t)k; idx->centroids = centroids;
    idx->list_off = (uint64_t *)calloc(k + 1, sizeof(uint64_t));
    idx->list_rows = (uint32_t *)malloc((n ? n : 1) * sizeof(uint32_t));
    for (uint64_t r = 0; r < n; ++r) idx->list_off[cluster_of[r] + 1]++;
    for (uint64_t c = 0; c < k; ++c) idx->list_off[c + 1] += idx->list_off[c];
    uint64_t *cursor = (uint64_t *)malloc(k * sizeof(uint64_t));
    memcpy(cursor, idx->list_off, k * sizeof(uint64_t));
    for (uint64_t r = 0; r < n; ++r) idx->list_rows[cursor[cluster_of[r]]++] = (uint32_t)r;
    free(cursor); free(cluster_of);
    *out = idx;
    return 0;
}

/* ===================================================================================
 * Blob (src/ivf/index.rs:65-128)
 * =================================================================================== */

static void put_u32(uint8_t *p, uint32_t v) {
    p[0] = (uint8_t)v; p[1] = (uint8_t)(v >> 8); p[2] = (uint8_t)(v >> 16); p[3] = (uint8_t)(v >> 24);
}
static uint32_t get_u32(const uint8_t *p) {
    return (uint32_t)p[0] | ((uint32_t)p[1] << 8) | ((uint32_t)p[2] << 16) | ((uint32_t)p[3] << 24);
}

int pqo_index_to_bytes(const pqo_index *idx, uint8_t **buf, size_t *len) {
    uint64_t k = idx->n_clusters, total = idx->list_off[k];
    size_t sz = 8 + (size_t)k * idx->dim * 4 + (size_t)k * 4 + (size_t)total * 4;
    uint8_t *b = (uint8_t *)malloc(sz ? sz : 1);
    if (!b) return -1;
    size_t off = 0;
    put_u32(b + off, idx->dim); off += 4;
    put_u32(b + off, idx->n_clusters); off += 4;
    for (uint64_t i = 0; i < k * idx->dim; ++i) {
        uint32_t bits; memcpy(&bits, &idx->centroids[i], 4);
        put_u32(b + off, bits); off += 4;
    }
    for (uint64_t c = 0; c < k; ++c) {
        uint64_t s = idx->list_off[c], e = idx->list_off[c + 1];
        put_u32(b + off, (uint32_t)(e - s)); off += 4;
        for (uint64_t i = s; i < e; ++i) { put_u32(b + off, idx->list_rows[i]); off += 4; }
    }
    *buf = b; *len = sz;
    return 0;
}

int pqo_index_from_bytes(const uint8_t *bytes, size_t len, pqo_index **out, char *err) {
    *out = NULL;
    if (len < 8) { set_err(err, "IVF index buffer too small"); return -1; }          /* :89 */
    size_t off = 0;
    uint32_t dim = get_u32(bytes); off += 4;
    uint32_t k = get_u32(bytes + off); off += 4;
    if (dim == 0) { set_err(err, "Embedding dimension must be > 0"); return -1; }    /* mod.rs:59 */
    if (k == 0) { set_err(err, "Cluster count must be > 0"); return -1; }            /* :24 */
    uint64_t clen = (uint64_t)k * dim;
    /* Rust would panic on an out-of-range slice; the oracle reports it instead. */
    if (len - off < clen * 4) { set_err(err, "IVF index buffer truncated"); return -2; }
    pqo_index *idx = (pqo_index *)calloc(1, sizeof(pqo_index));
    idx->dim = dim; idx->n_clusters = k;
    idx->centroids = (float *)malloc((clen ? clen : 1) * sizeof(float));
    for (uint64_t i = 0; i < clen; ++i) {
        uint32_t bits = get_u32(bytes + off); off += 4;
        memcpy(&idx->centroids[i], &bits, 4);
    }
    idx->list_off = (uint64_t *)calloc((uint64_t)k + 1, sizeof(uint64_t));
    /* first pass: lengths */
    size_t scan = off;
    uint64_t total = 0;
    for (uint32_t c = 0; c < k; ++c) {
        if (len - scan < 4) { pqo_index_free(idx); set_err(err, "IVF index buffer truncated"); return -2; }
        uint32_t ll = get_u32(bytes + scan); scan += 4;
        if ((len - scan) / 4 < ll) { pqo_index_free(idx); set_err(err, "IVF index buffer truncated"); return -2; }
        scan += (size_t)ll * 4;
        total += ll;
        idx->list_off[c + 1] = total;
    }
    idx->list_rows = (uint32_t *)malloc((total ? total : 1) * sizeof(uint32_t));
    for (uint32_t c = 0; c < k; ++c) {
        uint32_t ll = get_u32(bytes + off); off += 4;
        for (uint32_t i = 0; i < ll; ++i) {
            idx->list_rows[idx->list_off[c] + i] = get_u32(bytes + off); off += 4;
        }
    }
    *out = idx;
    return 0;
}

/* ===================================================================================
 * Probe (src/ivf/index.rs:130-149, :57-63)
 * =================================================================================== */

typedef struct { uint32_t idx; float dist; } cd_pair;

/* partial_cmp(...).unwrap_or(Equal): returns <0, 0, >0 */
static int f32_cmp(float a, float b) {
    if (a < b) return -1;
    if (a > b) return 1;
    return 0; /* equal or unordered */
}

/* Stable merge sort (Rust's slice::sort_by is stable; for a total preorder every stable
 * sort yields the same permutation). */
static void stable_sort_pairs(cd_pair *a, cd_pair *tmp, size_t n) {
    if (n < 2) return;
    size_t mid = n / 2;
    stable_sort_pairs(a, tmp, mid);
    stable_sort_pairs(a + mid, tmp, n - mid);
    size_t i = 0, j = mid, o = 0;
    while (i < mid && j < n) {
        if (f32_cmp(a[j].dist, a[i].dist) < 0) tmp[o++] = a[j++]; else tmp[o++] = a[i++];
    }
    while (i < mid) tmp[o++] = a[i++];
    while (j < n) tmp[o++] = a[j++];
    memcpy(a, tmp, n * sizeof(cd_pair));
}

uint32_t pqo_find_closest_centroids(const pqo_index *idx, const float *query, uint32_t nprobe,
                                    uint32_t *out) {
    uint32_t k = idx->n_clusters;
    if (nprobe > k) nprobe = k;                                                      /* :131 */
    cd_pair *cd = (cd_pair *)malloc((size_t)k * sizeof(cd_pair));
    cd_pair *tmp = (cd_pair *)malloc((size_t)k * sizeof(cd_pair));
    for (uint32_t i = 0; i < k; ++i) {
        cd[i].idx = i;
        cd[i].dist = pqo_squared_l2_ref4(query, idx->centroids + (uint64_t)i * idx->dim, idx->dim);
    }
    stable_sort_pairs(cd, tmp, k);                                                   /* :143 */
    for (uint32_t i = 0; i < nprobe; ++i) out[i] = cd[i].idx;
    free(cd); free(tmp);
    return nprobe;
}

int pqo_candidate_rows(const pqo_index *idx, const float *query, uint32_t nprobe,
                       uint32_t **rows, uint64_t *n_rows) {
    uint32_t k = idx->n_clusters;
    uint32_t np = nprobe > k ? k : nprobe;
    uint32_t *cl = (uint32_t *)malloc(((size_t)np + 1) * sizeof(uint32_t));
    np = pqo_find_closest_centroids(idx, query, nprobe, cl);
    uint64_t total = 0;
    for (uint32_t i = 0; i < np; ++i) total += idx->list_off[cl[i] + 1] - idx->list_off[cl[i]];
    uint32_t *r = (uint32_t *)malloc((total ? total : 1) * sizeof(uint32_t));
    uint64_t o = 0;
    for (uint32_t i = 0; i < np; ++i) {
        uint64_t s = idx->list_off[cl[i]], e = idx->list_off[cl[i] + 1];
        memcpy(r + o, idx->list_rows + s, (e - s) * sizeof(uint32_t));
        o += e - s;
    }
    free(cl);
    *rows = r; *n_rows = total;
    return 0;
}

/* ===================================================================================
 * std::collections::BinaryHeap emulation over {payload, distance}
 * =================================================================================== */

typedef struct { uint32_t row; float distance; } heap_item;
typedef struct { heap_item *data; size_t len; } bheap;

/* a <= b under Ord::cmp built from partial_cmp().unwrap_or(Equal) */
static int item_le(const heap_item *a, const heap_item *b) { return f32_cmp(a->distance, b->distance) <= 0; }

static size_t heap_sift_up(bheap *h, size_t start, size_t pos) {
    heap_item elt = h->data[pos];
    while (pos > start) {
        size_t parent = (pos - 1) / 2;
        if (item_le(&elt, &h->data[parent])) break;
        h->data[pos] = h->data[parent];
        pos = parent;
    }
    h->data[pos] = elt;
    return pos;
}

static void heap_sift_down_to_bottom(bheap *h, size_t pos) {
    size_t end = h->len, start = pos;
    heap_item elt = h->data[pos];
    size_t child = 2 * pos + 1;
    size_t lim = end >= 2 ? end - 2 : 0; /* end.saturating_sub(2) */
    while (child <= lim && end >= 2) {
        child += item_le(&h->data[child], &h->data[child + 1]) ? 1 : 0;
        h->data[pos] = h->data[child];
        pos = child;
        child = 2 * pos + 1;
    }
    if (child == end - 1) {
        h->data[pos] = h->data[child];
        pos = child;
    }
    h->data[pos] = elt;
    heap_sift_up(h, start, pos);
}

static void heap_push(bheap *h, heap_item it) {
    size_t old = h->len;
    h->data[h->len++] = it;
    heap_sift_up(h, 0, old);
}

static void heap_pop(bheap *h) {
    heap_item item = h->data[--h->len];
    if (h->len > 0) {
        heap_item t = h->data[0]; h->data[0] = item; item = t;
        heap_sift_down_to_bottom(h, 0);
    }
    (void)item;
}

/* stable insertion sort by distance (k is small); Rust's sort_by is stable */
static void stable_sort_items(heap_item *a, size_t n) {
    for (size_t i = 1; i < n; ++i) {
        heap_item x = a[i];
        size_t j = i;
        while (j > 0 && f32_cmp(x.distance, a[j - 1].distance) < 0) { a[j] = a[j - 1]; --j; }
        a[j] = x;
    }
}

/* heap policy shared by search.rs:119-126 and exec.rs:474-481 */
static void heap_offer(bheap *h, uint32_t k, heap_item it) {
    if (h->len < k) {
        heap_push(h, it);
    } else if (h->len > 0 && it.distance < h->data[0].distance) {
        heap_pop(h);
        heap_push(h, it);
    }
}

int pqo_topk_ivf(const pqo_index *idx, const float *embeddings, const float *query,
                 uint32_t query_len, uint32_t k, uint32_t nprobe, uint32_t *row_idx,
                 float *dist, uint32_t *n_found, uint64_t *n_candidates, char *err) {
    if (n_found) *n_found = 0;
    if (k == 0) { set_err(err, "k must be > 0"); return -1; }                        /* search.rs:67 */
    if (nprobe == 0) { set_err(err, "nprobe must be > 0"); return -1; }              /* :72 */
    if (query_len != idx->dim) {                                                     /* :91-98 */
        if (err) snprintf(err, PQO_ERRLEN, "Query dimension mismatch: expected %u, got %u",
                          idx->dim, query_len);
        return -1;
    }
    uint32_t *rows; uint64_t nrows;
    pqo_candidate_rows(idx, query, nprobe, &rows, &nrows);                           /* :100 */
    if (n_candidates) *n_candidates = nrows;
    bheap h; h.data = (heap_item *)malloc(((size_t)k + 1) * sizeof(heap_item)); h.len = 0;
    uint32_t dim = idx->dim;
    for (uint64_t i = 0; i < nrows; ++i) {                                           /* :115-127 */
        const float *vec = embeddings + (uint64_t)rows[i] * dim;
        heap_item it; it.row = rows[i];
        it.distance = pqo_squared_l2_ref4(query, vec, dim);                          /* query is arg a */
        heap_offer(&h, k, it);
    }
    for (size_t i = 0; i < h.len; ++i) h.data[i].distance = sqrtf(h.data[i].distance); /* :133 */
    stable_sort_items(h.data, h.len);                                                /* :136-140 */
    for (size_t i = 0; i < h.len; ++i) { row_idx[i] = h.data[i].row; dist[i] = h.data[i].distance; }
    if (n_found) *n_found = (uint32_t)h.len;
    free(h.data); free(rows);
    return 0;
}

int pqo_topk_ivf_batch(const pqo_index *idx, const float *embeddings, const float *queries,
                       uint32_t nq, uint32_t k, uint32_t nprobe, uint32_t *row_idx,
                       float *dist, uint32_t *n_found, uint64_t *n_candidates) {
    for (uint32_t q = 0; q < nq; ++q) {
        uint32_t nf = 0; uint64_t nc = 0;
        int rc = pqo_topk_ivf(idx, embeddings, queries + (uint64_t)q * idx->dim, idx->dim, k, nprobe,
                              row_idx + (uint64_t)q * k, dist + (uint64_t)q * k, &nf, &nc, NULL);
        if (rc) return rc;
        if (n_found) n_found[q] = nf;
        if (n_candidates) n_candidates[q] = nc;
    }
    return 0;
}

int pqo_topk_df(const float *embeddings, uint32_t dim, const uint32_t *rows, uint64_t n_rows,
                const float *query, uint32_t k, uint32_t *out_rows, float *out_d2,
                uint32_t *n_found) {
    if (n_found) *n_found = 0;
    if (k == 0) return 0; /* heap.len() < 0 never true; peek() is None => nothing kept */
    bheap h; h.data = (heap_item *)malloc(((size_t)k + 1) * sizeof(heap_item)); h.len = 0;
    for (uint64_t i = 0; i < n_rows; ++i) {                                          /* exec.rs:467-482 */
        heap_item it; it.row = rows[i];
        it.distance = pqo_squared_l2_seq(embeddings + (uint64_t)rows[i] * dim, query, dim);
        heap_offer(&h, k, it);
    }
    stable_sort_items(h.data, h.len);                                                /* exec.rs:270-274 */
    for (size_t i = 0; i < h.len; ++i) { out_rows[i] = h.data[i].row; out_d2[i] = h.data[i].distance; }
    if (n_found) *n_found = (uint32_t)h.len;
    free(h.data);
    return 0;
}

/* src/df_vector/access.rs:214-242 (one call on a fresh cursor) */
uint64_t pqo_candidate_cursor_take(const uint32_t *const *cand, const uint64_t *cand_len,
                                   uint32_t file_count, uint64_t batch_size,
                                   uint32_t *out_file, uint32_t *out_row) {
    if (batch_size == 0 || file_count == 0) return 0;
    uint64_t *pos = (uint64_t *)calloc(file_count, sizeof(uint64_t));
    uint64_t n = 0, idx = 0;
    while (n < batch_size) {
        int progressed = 0;
        for (uint32_t t = 0; t < file_count; ++t) {
            uint32_t f = (uint32_t)(idx % file_count);
            idx++;
            if (pos[f] < cand_len[f]) {
                out_file[n] = f; out_row[n] = cand[f][pos[f]++];
                n++; progressed = 1;
                if (n >= batch_size) break;
            }
        }
        if (!progressed) break;
    }
    free(pos);
    return n;
}
