// pqv.hpp -- C++ host-side mirror of the reference's Rust API over the C ABI (include/pqv.h):
//   pqv::IndexBuilder   src/ivf/parquet.rs:23-103   (n_clusters / max_iters / seed builder)
//   pqv::TopkBuilder    src/ivf/search.rs:49-81     (k and nprobe must be set and > 0)
//   pqv::SearchResult   src/ivf/search.rs:41-45
// Header-only RAII wrappers; errors become pqv::Error carrying the reference's message text.
#pragma once
#include <cstdint>
#include <memory>
#include <optional>
#include <stdexcept>
#include <string>
#include <vector>

#include "../../include/pqv.h"

namespace pqv {

struct Error : std::runtime_error {
    int code;
    Error(int c, const std::string &m) : std::runtime_error(m), code(c) {}
};
inline void check(int rc) { if (rc != PQV_OK) throw Error(rc, pqv_last_error()); }

struct SearchResult { uint32_t row_idx; float distance; };

class Corpus {
public:
    Corpus(int device, const float *rows, uint64_t n, uint32_t dim) {
        pqv_corpus *h = nullptr;
        check(pqv_corpus_upload(device, rows, n, dim, &h));
        h_.reset(h);
    }
    pqv_corpus *get() const { return h_.get(); }
    uint64_t rows() const { return pqv_corpus_rows(h_.get()); }
    uint32_t dim() const { return pqv_corpus_dim(h_.get()); }
private:
    struct Del { void operator()(pqv_corpus *p) const { pqv_corpus_free(p); } };
    std::unique_ptr<pqv_corpus, Del> h_;
};

class Index {
public:
    explicit Index(pqv_index *h) : h_(h) {}
    static Index from_bytes(const std::vector<uint8_t> &blob) {
        pqv_index *h = nullptr;
        check(pqv_index_from_bytes(blob.data(), blob.size(), &h));
        return Index(h);
    }
    std::vector<uint8_t> to_bytes() const {
        uint8_t *buf = nullptr; size_t len = 0;
        check(pqv_index_to_bytes(h_.get(), &buf, &len));
        std::vector<uint8_t> out(buf, buf + len);
        pqv_bytes_free(buf);
        return out;
    }
    uint32_t dim() const { return pqv_index_dim(h_.get()); }
    uint32_t n_clusters() const { return pqv_index_n_clusters(h_.get()); }
    pqv_index *get() const { return h_.get(); }
private:
    struct Del { void operator()(pqv_index *p) const { pqv_index_free(p); } };
    std::unique_ptr<pqv_index, Del> h_;
};

class IndexBuilder {
public:
    explicit IndexBuilder(const Corpus &corpus) : corpus_(corpus) {}
    IndexBuilder &n_clusters(uint32_t v) { n_clusters_ = v; return *this; }
    IndexBuilder &max_iters(uint32_t v) { max_iters_ = v; return *this; }
    IndexBuilder &seed(uint64_t v) { seed_ = v; return *this; }
    IndexBuilder &workers(uint32_t v) { workers_ = v; return *this; }
    Index build() const {
        if (max_iters_ == 0) throw Error(PQV_ERR_INVALID, "max_iters must be > 0");      // parquet.rs:90
        if (n_clusters_ && *n_clusters_ == 0) throw Error(PQV_ERR_INVALID, "n_clusters must be > 0"); // :93
        pqv_index *h = nullptr;
        check(pqv_index_build(corpus_.get(), n_clusters_.value_or(0), max_iters_, seed_, workers_, &h));
        return Index(h);
    }
private:
    const Corpus &corpus_;
    std::optional<uint32_t> n_clusters_;
    uint32_t max_iters_ = 20;      // parquet.rs:37
    uint64_t seed_ = 42;           // parquet.rs:38
    uint32_t workers_ = 0;
};

class Searcher {
public:
    Searcher(const Index &index, Corpus &corpus, uint32_t flags = PQV_LAYOUT_IVF_ORDERED) {
        pqv_searcher *h = nullptr;
        check(pqv_searcher_create(index.get(), corpus.get(), flags, &h));
        h_.reset(h);
        dim_ = index.dim();
    }
    pqv_searcher *get() const { return h_.get(); }
    uint32_t dim() const { return dim_; }
private:
    struct Del { void operator()(pqv_searcher *p) const { pqv_searcher_free(p); } };
    std::unique_ptr<pqv_searcher, Del> h_;
    uint32_t dim_ = 0;
};

class TopkBuilder {
public:
    TopkBuilder(const Searcher &s, const std::vector<float> &query) : s_(s), query_(query) {}
    TopkBuilder &k(uint32_t v) { if (!v) throw Error(PQV_ERR_INVALID, "k must be > 0"); k_ = v; return *this; }
    TopkBuilder &nprobe(uint32_t v) { if (!v) throw Error(PQV_ERR_INVALID, "nprobe must be > 0"); nprobe_ = v; return *this; }
    std::vector<SearchResult> search() const {
        if (!k_) throw Error(PQV_ERR_INVALID, "k must be set");              // search.rs:77
        if (!nprobe_) throw Error(PQV_ERR_INVALID, "nprobe must be set");    // search.rs:78
        std::vector<uint32_t> rows(*k_);
        std::vector<float> dist(*k_);
        uint32_t found = 0;
        check(pqv_topk(s_.get(), query_.data(), 1, static_cast<uint32_t>(query_.size()), *k_, *nprobe_, 0,
                       PQV_L2SQ_REF4, 1, rows.data(), dist.data(), &found, nullptr));
        std::vector<SearchResult> out;
        for (uint32_t i = 0; i < found; ++i) out.push_back({rows[i], dist[i]});
        return out;
    }
private:
    const Searcher &s_;
    const std::vector<float> &query_;
    std::optional<uint32_t> k_, nprobe_;
};

}  // namespace pqv
