# One-off fuzz of the screened search against the oracle on the GPU box: FZ_LO / FZ_HI = seed range (tests/test_gpu_parity.py::_fuzz_case).
# Variants used: default; PQV_SEED_ROWS=64 PQV_WIDE_ROWS=256 (loose seeds, many small blocks); PQV_CAND_CAP=8 (spills); PQV_PAIR_PRUNE=0; PQV_QUAD_WIDTH=64.
python - <<'PY'
import sys, os, time
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), "tests"))
os.environ["PQV_RERANK_MODE"] = "tile"; os.environ["PQV_TILE_FILTER"] = "2"
import conftest
import test_gpu_parity as T
from oracle_binding import Oracle
import pq_vector_amd as pqv
oracle = Oracle()
t = time.time(); bad = 0
for seed in range(int(os.environ.get("FZ_LO", 2000)), int(os.environ.get("FZ_HI", 2100))):
    try:
        T._fuzz_case(pqv, oracle, seed)
    except AssertionError as e:
        bad += 1; print("FAIL seed", seed, e)
print("fuzz", os.environ.get("FZ_TAG", ""), "done:", bad, "failures", round(time.time() - t, 1), "s")
PY
