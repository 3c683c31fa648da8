import numpy as np, pyarrow as pa, pyarrow.parquet as pq, time, os, sys, ctypes
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import pq_vector_amd as pqv
from pq_vector_amd import parquet_io
n, dim = 1000000, 1024
path = "/tmp/probe.parquet"
if not os.path.exists(path):
    rng = np.random.default_rng(1)
    vec = rng.random((n, dim), dtype=np.float32)
    col = pa.ListArray.from_arrays(pa.array(np.arange(0, (n + 1) * dim, dim, dtype=np.int32)), pa.array(vec.reshape(-1)))
    pq.write_table(pa.table({"embedding": col}), path, compression="NONE", use_dictionary=False, row_group_size=n)
    del vec, col
class Sink:
    def write_rows_ptr(self, row, addr, m, f64=False): pass
class Touch:
    def __init__(self): self.buf = [np.empty(2 << 20, np.uint8) for _ in range(64)]; self.i = 0
    def write_rows_ptr(self, row, addr, m, f64=False):
        b = self.buf[(row // 256) % 64]
        ctypes.memmove(b.ctypes.data, addr, m * dim * 4)
for rep in range(2):
    for thr in (8,):
        t0 = time.perf_counter(); plan = parquet_io._plan_pages(path, "embedding", thr); t1 = time.perf_counter()
        c = [0, 0]; parquet_io._upload_pages(plan, Sink(), thr, c); t2 = time.perf_counter()
        parquet_io._upload_pages(plan, Touch(), thr, c); t3 = time.perf_counter()
        corpus = pqv.Corpus.create(n, dim, 0); t4 = time.perf_counter()
        parquet_io._upload_pages(plan, corpus, thr, c); corpus.finish(n); t5 = time.perf_counter()
        print(f"rep {rep} thr {thr}: plan+levels {t1-t0:.3f}  dispatch only {t2-t1:.3f}  memmove to pageable {t3-t2:.3f}  corpus create {t4-t3:.3f}  upload {t5-t4:.3f}")
        corpus.close()
    st = {}
    t0 = time.perf_counter(); cc = parquet_io.load_embedding_column(path, "embedding", 0, stats=st); t1 = time.perf_counter()
    print("load_embedding_column", round(t1 - t0, 3), st.get("GBps"), st.get("first_page_s")); cc.close()
