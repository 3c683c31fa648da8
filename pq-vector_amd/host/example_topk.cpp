// example_topk.cpp -- the reference's examples/build_index.rs + topk_search.rs flow through the
// C++ host mirror.  Built by `make -C pq-vector_amd/csrc host_example`; runs on a GPU box.
#include <cstdio>
#include <random>

#include "pqv.hpp"

int main(int argc, char **argv) {
    try {
        const uint64_t n = 20000; const uint32_t dim = 64;
        std::mt19937 gen(1234);
        std::uniform_real_distribution<float> u(0.f, 1.f);
        std::vector<float> data(n * dim), query(dim);
        for (auto &v : data) v = u(gen);
        for (auto &v : query) v = u(gen);
        pqv::Corpus corpus(0, data.data(), n, dim);
        pqv::Index index = pqv::IndexBuilder(corpus).n_clusters(16).build();
        pqv::Searcher searcher(index, corpus);
        auto hits = pqv::TopkBuilder(searcher, query).k(5).nprobe(4).search();
        std::printf("index: dim %u, %u clusters, blob %zu bytes\n", index.dim(), index.n_clusters(),
                    index.to_bytes().size());
        for (auto &h : hits) std::printf("row %u distance %.6f\n", h.row_idx, h.distance);
        if (argc > 1) {
            // tests/test_gpu_exchange_and_limits.py: inputs, index blob and answers as raw little-endian arrays, so that the CPU
            // oracle can be run on exactly this data
            std::FILE *f = std::fopen(argv[1], "wb");
            if (!f) return 3;
            const uint64_t head[4] = {n, dim, index.to_bytes().size(), hits.size()};
            std::fwrite(head, sizeof head, 1, f);
            std::fwrite(data.data(), sizeof(float), data.size(), f);
            std::fwrite(query.data(), sizeof(float), query.size(), f);
            const auto blob = index.to_bytes();
            std::fwrite(blob.data(), 1, blob.size(), f);
            for (auto &h : hits) { std::fwrite(&h.row_idx, 4, 1, f); std::fwrite(&h.distance, 4, 1, f); }
            std::fclose(f);
        }
        return hits.size() == 5 ? 0 : 1;
    } catch (const pqv::Error &e) {
        std::fprintf(stderr, "pqv error %d: %s\n", e.code, e.what());
        return 2;
    }
}
