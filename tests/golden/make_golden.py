#!/usr/bin/env python3
"""Generates tests/golden/*.npz.

The reference is Rust and cannot be built or run in this environment (no cargo/rustc), so
these vectors are produced by the repo's own CPU oracle (oracle/pqv_oracle.c) -- itself
pinned to every known answer the reference's tests hold (tests/test_oracle_golden.py).
They freeze inputs + expected outputs so that (a) the oracle cannot drift silently and
(b) the GPU path is checked against committed data on the GPU box.

    python tests/golden/make_golden.py
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
from oracle_binding import Oracle  # noqa: E402


def case(o, name, data, n_clusters, queries, k, nprobe, workers_list=(1, 8), max_iters=20, seed=42):
    out = {"data": data, "queries": queries, "n_clusters": np.int64(n_clusters), "k": np.int64(k),
           "nprobe": np.int64(nprobe), "max_iters": np.int64(max_iters), "seed": np.int64(seed),
           "workers_list": np.array(workers_list, dtype=np.int64)}
    for w in workers_list:
        idx = o.build_index(data, n_clusters=n_clusters, max_iters=max_iters, seed=seed, workers=w)
        out[f"w{w}_blob"] = np.frombuffer(idx.to_bytes(), dtype=np.uint8)
        rows, dist, nf, nc = idx.topk_batch(data, queries, k, nprobe)
        out[f"w{w}_topk_rows"] = rows
        out[f"w{w}_topk_dist_bits"] = dist.view(np.uint32)
        out[f"w{w}_n_found"] = nf
        out[f"w{w}_n_candidates"] = nc
        out[f"w{w}_probe"] = np.stack([idx.find_closest_centroids(q, nprobe) for q in queries])
    np.savez_compressed(os.path.join(HERE, name + ".npz"), **out)
    print(name, {k_: (v.shape if hasattr(v, "shape") else v) for k_, v in out.items() if k_.startswith("w1")})


def main():
    o = Oracle()
    rng = np.random.default_rng(20260928)
    # (b) random corpora: 1k x 32 and 4k x 128, quantised to 16 bits so the files compress
    d1 = (rng.integers(0, 1 << 16, size=(1000, 32)).astype(np.float32) / np.float32(1 << 16))
    q1 = (rng.integers(0, 1 << 16, size=(12, 32)).astype(np.float32) / np.float32(1 << 16))
    case(o, "rand_1k_x32", d1, 0, q1, 10, 5)          # default n_clusters = ceil(sqrt(1000)) = 32
    d2 = (rng.integers(0, 1 << 16, size=(4000, 128)).astype(np.float32) / np.float32(1 << 16))
    q2 = (rng.integers(0, 1 << 16, size=(12, 128)).astype(np.float32) / np.float32(1 << 16))
    case(o, "rand_4k_x128", d2, 16, q2, 10, 4)
    # (c) tie-heavy integer-valued vectors; k chosen so boundary ties matter
    d3 = rng.integers(0, 3, size=(2000, 8)).astype(np.float32)
    q3 = rng.integers(0, 3, size=(12, 8)).astype(np.float32)
    case(o, "ties_2k_x8", d3, 6, q3, 5, 3, workers_list=(1,), max_iters=6)
    # unaligned dimension (dim % 4 != 0): scalar tail of squared_l2_distance
    d4 = (rng.integers(0, 1 << 12, size=(1500, 30)).astype(np.float32) / np.float32(1 << 12))
    q4 = (rng.integers(0, 1 << 12, size=(8, 30)).astype(np.float32) / np.float32(1 << 12))
    case(o, "rand_1500_x30", d4, 9, q4, 7, 3, workers_list=(3,))


if __name__ == "__main__":
    main()
