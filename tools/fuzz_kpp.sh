# tools/fuzz_kpp.sh: the k-means++ rounds on the device (kernels_kpp.hip) against the host walk on the SAME build -- blob identity, no
# oracle needed, so the cases are big enough for the 50 000-row subset: FZ_LO / FZ_HI = seed range.  Styles: uniform, integer-valued
# (a tie in almost every add), clustered with far-out rows, FEW DISTINCT rows (the total reaches 0 after that many rounds: the device
# hands over to the host's range_usize draws), tiny values, one huge row.
python - <<'PY'
import sys, os, time
sys.path.insert(0, os.getcwd())
import numpy as np
import pq_vector_amd as pqv
bad = 0; t = time.time(); cases = 0; rows = 0
for seed in range(int(os.environ.get("FZ_LO", 0)), int(os.environ.get("FZ_HI", 60))):
    rng = np.random.default_rng(seed)
    dim = int(rng.choice([128, 192, 256, 384, 768]))
    kc = int(rng.integers(9, 400))
    n = int(rng.integers(max(1100, kc * 3), 140_000)) if dim <= 256 else int(rng.integers(max(1100, kc * 3), 60_000))
    workers = int(rng.choice([1, 2, 3, 7, 8, 16, 64, 256, 1000]))
    style = seed % 6
    if style == 0: data = rng.random((n, dim), dtype=np.float32)
    elif style == 1: data = rng.integers(0, 256, size=(n, dim)).astype(np.float32)
    elif style == 2:
        cen = rng.standard_normal((16, dim)).astype(np.float32) * 3
        data = (cen[rng.integers(0, 16, n)] + rng.standard_normal((n, dim)).astype(np.float32) * np.float32(0.3)).astype(np.float32)
        data[rng.integers(0, n, 3)] *= np.float32(rng.choice([50.0, 1e4]))
    elif style == 3:
        base = rng.random((int(rng.integers(kc // 2 + 1, kc + 40)), dim), dtype=np.float32); data = base[rng.integers(0, len(base), n)]
    elif style == 4: data = (rng.random((n, dim), dtype=np.float32) * np.float32(1e-18)).astype(np.float32)
    else:
        data = rng.random((n, dim), dtype=np.float32); data[int(rng.integers(0, n))] = np.float32(1e12)
    corpus = pqv.Corpus.upload(data)
    blobs = {}
    for mode in ("1", "0"):
        os.environ["PQV_KPP_DEVICE"] = mode
        blobs[mode] = pqv.IndexBuilder(corpus).n_clusters(kc).max_iters(2).seed(seed).workers(workers).build().to_bytes()
    cases += 1; rows += n
    if blobs["1"] != blobs["0"]:
        bad += 1; print("FAIL seed", seed, dim, kc, n, style, workers)
print("k-means++ fuzz done:", cases, "cases,", rows, "rows,", bad, "failures", round(time.time() - t, 1), "s")
PY
