"""ctypes binding of libpqv_hip.so (the C ABI declared in include/pqv.h).

The library is built in-tree by `make -C pq-vector_amd/csrc` (see __graft_entry__.build)
and must be present: there is no CPU or pure-Python fallback for any compute entry point.
"""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
# PQV_LIB_PATH: diagnostic builds of the same library (tools/ only; e.g. the phase-timestamp build)
LIB_PATH = os.environ.get("PQV_LIB_PATH") or os.path.join(_HERE, "libpqv_hip.so")

u8p = C.POINTER(C.c_uint8)
u32p = C.POINTER(C.c_uint32)
u64p = C.POINTER(C.c_uint64)
f32p = C.POINTER(C.c_float)
f64p = C.POINTER(C.c_double)
vp = C.c_void_p

PQV_OK = 0
PQV_ERR_INVALID = -1
PQV_ERR_NO_DEVICE = -2
PQV_ERR_HIP = -3
PQV_ERR_OOM = -4
PQV_ERR_UNSUPPORTED = -5
PQV_ERR_FORMAT = -6

PQV_L2SQ_REF4 = 0
PQV_L2SQ_SEQ = 1
PQV_COSINE = 2
PQV_L2SQ_MFMA = 3

PQV_LAYOUT_IVF_ORDERED = 0x0
PQV_LAYOUT_ROW_ORDER = 0x1
PQV_RELEASE_ROW_ORDER = 0x2
PQV_RELEASE_IF_COPIED = 0x4


class Counters(C.Structure):
    _fields_ = [("queries", C.c_uint64), ("candidate_rows", C.c_uint64),
                ("embeddings_fetched", C.c_uint64), ("kernel_launches", C.c_uint64),
                ("exact_replays", C.c_uint64), ("screened_pairs", C.c_uint64),
                ("screen_survivors", C.c_uint64)]


# name -> (restype, argtypes); every symbol include/pqv.h declares
SIGNATURES = {
    "pqv_last_error": (C.c_char_p, []),
    "pqv_device_count": (C.c_int, []),
    "pqv_abi_version": (C.c_int, []),
    "pqv_merge_topk_packed_device": (C.c_int, [C.c_int, vp, vp, C.c_uint32, C.c_uint32, C.c_uint32, vp, vp, vp]),
    "pqv_shard_rccl_path": (C.c_char_p, []),
    "pqv_shard_unique_id": (C.c_int, [u8p]),
    "pqv_shard_comm_create": (C.c_int, [C.c_int, C.c_uint32, C.c_uint32, u8p, C.POINTER(vp)]),
    "pqv_shard_comm_adopt": (C.c_int, [C.c_int, vp, C.POINTER(vp)]),
    "pqv_shard_row_groups": (C.c_int, [u64p, C.c_uint32, C.c_uint32, C.c_uint32, u32p, u32p, u64p, u64p]),
    "pqv_shard_comm_rank": (C.c_uint32, [vp]),
    "pqv_shard_comm_world": (C.c_uint32, [vp]),
    "pqv_shard_exchange": (C.c_int, [vp, vp, vp, vp, C.c_uint32, C.c_uint32, vp, vp, vp]),
    "pqv_shard_comm_free": (None, [vp]),
    "pqv_candidate_cursor_new": (C.c_int, [C.c_uint32, C.POINTER(vp)]),
    "pqv_candidate_cursor_add": (C.c_int, [vp, C.c_uint32, u32p, C.c_uint64]),
    "pqv_candidate_cursor_next_batch": (C.c_int, [vp, C.c_uint64, u32p, u32p, u64p, u64p]),
    "pqv_candidate_cursor_free": (None, [vp]),
    "pqv_topk_device_flags": (C.c_int, [vp, vp, C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint64, C.c_int, C.c_int, vp, vp, vp, vp, vp, vp]),
    "pqv_rerank_device": (C.c_int, [C.c_int, vp, vp, vp, C.c_uint64, C.c_uint32, C.c_uint32, C.c_int, vp, vp, vp, vp]),
    "pqv_diag_rng": (C.c_int, [u8p, C.c_uint64, C.c_int, C.c_uint64, u64p, C.c_uint64]),
    "pqv_searcher_set_option": (C.c_int, [vp, C.c_char_p, C.c_int64]),
    "pqv_searcher_describe": (C.c_int, [vp, C.c_uint32, C.c_uint32, C.c_uint32, C.c_int, C.c_char_p, C.c_size_t]),
    "pqv_searcher_footprint": (C.c_int, [vp, u64p, u64p, u64p, u64p]),
    "pqv_corpus_upload": (C.c_int, [C.c_int, f32p, C.c_uint64, C.c_uint32, C.POINTER(vp)]),
    "pqv_corpus_create": (C.c_int, [C.c_int, C.c_uint64, C.c_uint32, C.POINTER(vp)]),
    "pqv_corpus_append": (C.c_int, [vp, f32p, C.c_uint64]),
    "pqv_corpus_append_f64": (C.c_int, [vp, f64p, C.c_uint64]),
    "pqv_corpus_write_rows": (C.c_int, [vp, C.c_uint64, f32p, C.c_uint64]),
    "pqv_corpus_write_rows_f64": (C.c_int, [vp, C.c_uint64, f64p, C.c_uint64]),
    "pqv_corpus_finish": (C.c_int, [vp, C.c_uint64]),
    "pqv_parquet_levels_check": (C.c_int, [u8p, C.c_uint64, C.c_uint32, C.c_uint64, C.c_int, C.c_uint64, u64p]),
    "pqv_parquet_dict_decode": (C.c_int, [u8p, C.c_uint64, vp, C.c_uint64, C.c_uint32, C.c_uint64, vp]),
    "pqv_corpus_write_plain_pages": (C.c_int, [vp, u8p, u64p, u32p, u64p, u32p, C.c_uint32, C.c_uint32, C.c_uint32, C.c_int, u32p]),
    "pqv_parquet_page_headers": (C.c_int, [u8p, C.c_uint64, C.c_uint64, C.c_uint32, C.POINTER(C.c_int32), u32p]),
    "pqv_corpus_from_device": (C.c_int, [C.c_int, vp, C.c_uint64, C.c_uint32, C.POINTER(vp)]),
    "pqv_corpus_rows": (C.c_uint64, [vp]),
    "pqv_corpus_dim": (C.c_uint32, [vp]),
    "pqv_corpus_device": (C.c_int, [vp]),
    "pqv_corpus_fetch_rows": (C.c_int, [vp, u32p, C.c_uint64, f32p]),
    "pqv_corpus_free": (None, [vp]),
    "pqv_index_build": (C.c_int, [vp, C.c_uint32, C.c_uint32, C.c_uint64, C.c_uint32, C.POINTER(vp)]),
    "pqv_index_build_stats": (C.c_int, [C.POINTER(C.c_double), C.c_uint32]),
    "pqv_kpp_pick": (C.c_int, [C.c_int, C.POINTER(C.c_float), C.c_uint32, C.c_uint32, C.c_float, C.POINTER(C.c_uint64),
                               C.POINTER(C.c_float), C.POINTER(C.c_uint32)]),
    "pqv_index_build_host": (C.c_int, [C.c_int, f32p, C.c_uint64, C.c_uint32, C.c_uint32, C.c_uint32,
                                       C.c_uint64, C.c_uint32, C.POINTER(vp)]),
    "pqv_kmeans": (C.c_int, [vp, C.c_uint32, C.c_uint32, C.c_uint64, C.c_uint32, f32p, u32p, u32p]),
    "pqv_index_from_bytes": (C.c_int, [C.c_char_p, C.c_size_t, C.POINTER(vp)]),
    "pqv_index_to_bytes": (C.c_int, [vp, C.POINTER(u8p), C.POINTER(C.c_size_t)]),
    "pqv_bytes_free": (None, [u8p]),
    "pqv_index_from_parts": (C.c_int, [C.c_uint32, C.c_uint32, f32p, u64p, u32p, C.POINTER(vp)]),
    "pqv_index_dim": (C.c_uint32, [vp]),
    "pqv_index_n_clusters": (C.c_uint32, [vp]),
    "pqv_index_n_rows": (C.c_uint64, [vp]),
    "pqv_index_centroids": (f32p, [vp]),
    "pqv_index_list_offsets": (u64p, [vp]),
    "pqv_index_list_rows": (u32p, [vp]),
    "pqv_index_free": (None, [vp]),
    "pqv_searcher_create": (C.c_int, [vp, vp, C.c_uint32, C.POINTER(vp)]),
    "pqv_searcher_free": (None, [vp]),
    "pqv_probe": (C.c_int, [vp, f32p, C.c_uint32, C.c_uint32, u32p, u32p]),
    "pqv_candidate_rows": (C.c_int, [vp, f32p, C.c_uint32, C.c_uint32, C.POINTER(u32p), u64p]),
    "pqv_rows_free": (None, [u32p]),
    "pqv_topk": (C.c_int, [vp, f32p, C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint64,
                           C.c_int, C.c_int, u32p, f32p, u32p, u64p]),
    "pqv_topk_device": (C.c_int, [vp, vp, C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint64, C.c_int,
                                  C.c_int, vp, vp, vp, vp, vp]),
    "pqv_brute_topk": (C.c_int, [vp, f32p, C.c_uint32, C.c_uint32, C.c_uint32, C.c_int, u32p, f32p, u32p]),
    "pqv_rerank": (C.c_int, [C.c_int, f32p, f32p, u32p, u8p, C.c_uint64, C.c_uint32, C.c_uint32,
                             C.c_int, u32p, f32p, u32p]),
    "pqv_rerank_f64": (C.c_int, [C.c_int, f32p, f64p, u32p, u8p, C.c_uint64, C.c_uint32, C.c_uint32,
                                 C.c_int, u32p, f32p, u32p]),
    "pqv_rerank_finish": (C.c_int, [u32p, f32p, C.c_uint32, u32p, f32p]),
    "pqv_rerank_device_flags": (C.c_int, [C.c_int, vp, vp, vp, C.c_uint64, C.c_uint32, C.c_uint32, C.c_int, vp, vp, vp, vp, vp]),
    "pqv_merge_topk": (C.c_int, [f32p, u32p, u32p, C.c_uint32, C.c_uint32, C.c_uint32, f32p, u32p,
                                 u32p, u32p]),
    "pqv_merge_topk_device": (C.c_int, [C.c_int, vp, vp, vp, C.c_uint32, C.c_uint32, C.c_uint32, vp, vp, vp]),
    "pqv_counters": (C.c_int, [vp, C.POINTER(Counters)]),
    "pqv_set_timing": (C.c_int, [vp, C.c_int]),
    "pqv_timing_read": (C.c_int, [vp, C.POINTER(C.c_double), C.POINTER(C.c_double), u32p]),
}

_lib = None


def _share_hip_runtime_with_torch():
    """One HIP runtime per process.  PyTorch-ROCm wheels carry their own libamdhip64.so.7; if this library is
    loaded first it binds /opt/rocm's copy, and torch's later initialisation in the same process then finds
    "No HIP GPUs" (two runtimes on one device).  With torch imported first the dynamic linker hands us the copy
    torch already loaded (same SONAME) and everything shares one runtime -- so when torch is installed but not
    yet loaded, map its copy before ours.  Without torch (a Rust host) nothing happens."""
    import importlib.util
    import sys
    if "torch" in sys.modules:
        return
    try:
        spec = importlib.util.find_spec("torch")
    except (ImportError, ValueError):
        spec = None
    if spec is None or not spec.origin:
        return
    cand = os.path.join(os.path.dirname(spec.origin), "lib", "libamdhip64.so")
    if os.path.exists(cand):
        try:
            C.CDLL(cand, mode=C.RTLD_GLOBAL)
        except OSError:
            pass


def lib():
    """Load libpqv_hip.so; raises loudly if the HIP extension has not been built."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise ImportError(
                f"{LIB_PATH} is missing: build the HIP extension first "
                "(python -c 'import __graft_entry__ as g; g.build()' or make -C pq-vector_amd/csrc). "
                "pq_vector_amd has no CPU fallback.")
        _share_hip_runtime_with_torch()
        l = C.CDLL(LIB_PATH)
        for name, (res, args) in SIGNATURES.items():
            fn = getattr(l, name)  # AttributeError if the library lacks a declared symbol
            fn.restype = res
            fn.argtypes = args
        _lib = l
    return _lib
