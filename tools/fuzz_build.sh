# One-off fuzz of the index build (screened assignment forced) against the oracle blob on the GPU box: FZ_LO / FZ_HI = seed range.
python - <<'PY'
import sys, os, time
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), "tests"))
import numpy as np
import conftest
from oracle_binding import Oracle
import pq_vector_amd as pqv
oracle = Oracle()
os.environ.setdefault("PQV_ASSIGN_SCREEN", "1")          # PQV_ASSIGN_GEMM=2 (the round-3 f16 contraction, forced) / 0 (the f32 screen)
bad = 0; t = time.time()
for seed in range(int(os.environ.get("FZ_LO", 100)), int(os.environ.get("FZ_HI", 130))):
    rng = np.random.default_rng(seed)
    dim = int(rng.choice([128, 256, 384, 768, 64, 192, 100, 36]))
    kc = int(rng.integers(16, 200))
    n = int(rng.integers(kc * 3, kc * 40))
    style = seed % 4
    if style == 0: data = rng.random((n, dim), dtype=np.float32)
    elif style == 1: data = (rng.integers(0, 3, size=(n, dim)) * 0.5).astype(np.float32)
    elif style == 2:
        cen = rng.standard_normal((8, dim)).astype(np.float32) * 3
        data = (cen[rng.integers(0, 8, n)] + rng.standard_normal((n, dim)).astype(np.float32) * rng.choice([0.01, 0.3, 2.0], size=(n, 1)).astype(np.float32)).astype(np.float32)
    else:
        base = rng.random((n // 4 + 1, dim), dtype=np.float32); data = base[rng.integers(0, len(base), n)]
        data[rng.integers(0, n, 3)] *= np.float32(rng.choice([1.0, 50.0, 1e6]))       # rows far outside the centroid range
    workers = int(os.environ.get("FZ_WORKERS", 2)) or int(rng.choice([1, 2, 3, 8, 64, 256]))     # FZ_WORKERS=0: a different chunking per case
    want = oracle.build_index(data, n_clusters=kc, workers=workers, max_iters=3, seed=seed).to_bytes()
    got = pqv.IndexBuilder(pqv.Corpus.upload(data)).n_clusters(kc).max_iters(3).seed(seed).workers(workers).build().to_bytes()
    if got != want:
        bad += 1; print("FAIL seed", seed, dim, kc, n, style, workers)
print("build fuzz done:", bad, "failures", round(time.time() - t, 1), "s")
PY
