"""pq_vector_amd -- MI355X-native IVF index build + top-k search (pq-vector's hot path).

Host-side mirror of the reference's Rust API over the C ABI of include/pqv.h:

    reference (Rust)                         here
    ---------------------------------------  -------------------------------------------
    IndexBuilder::new(src, col)              IndexBuilder(source[, embedding_column])
      .n_clusters(n).max_iters(m).seed(s)      .n_clusters(n).max_iters(m).seed(s)
      .build_inplace() / .build_new(out)       .build() -> Index   (in-memory form)
    TopkBuilder::new(path, &query)           TopkBuilder(searcher, query)
      .k(k)?.nprobe(n)?.search().await?        .k(k).nprobe(n).search() -> [SearchResult]
    SearchResult{row_idx, distance}          SearchResult(row_idx, distance)

(src/ivf/parquet.rs:23-103, src/ivf/search.rs:41-81).  All compute runs in the HIP kernels
behind libpqv_hip.so; importing this package without the built library fails loudly.
"""
from .api import (CandidateCursor, Corpus, Index, IndexBuilder, PqvError, Searcher, SearchResult, TopkBuilder,
                  device_count, merge_topk, rerank_batch, rerank_finish, searcher_for_parquet)
from .parquet_io import has_pq_vector_index, read_index_from_parquet
from ._ffi import (PQV_L2SQ_REF4, PQV_L2SQ_SEQ, PQV_COSINE, PQV_L2SQ_MFMA, PQV_LAYOUT_IVF_ORDERED, PQV_LAYOUT_ROW_ORDER,
                   PQV_RELEASE_ROW_ORDER, PQV_RELEASE_IF_COPIED, LIB_PATH)

__all__ = ["CandidateCursor", "Corpus", "Index", "IndexBuilder", "PqvError", "Searcher", "SearchResult",
           "TopkBuilder", "device_count", "merge_topk", "rerank_batch", "rerank_finish", "searcher_for_parquet", "PQV_COSINE", "PQV_L2SQ_MFMA",
           "has_pq_vector_index", "read_index_from_parquet", "PQV_L2SQ_REF4",
           "PQV_L2SQ_SEQ", "PQV_LAYOUT_IVF_ORDERED", "PQV_LAYOUT_ROW_ORDER",
           "PQV_RELEASE_ROW_ORDER", "PQV_RELEASE_IF_COPIED", "LIB_PATH"]
