import os, sys, time
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import pq_vector_amd as pqv
from pq_vector_amd import parquet_io
path = "/tmp/probe.parquet"
for rep in range(4):
    st = {}
    t0 = time.perf_counter(); cc = parquet_io.load_embedding_column(path, "embedding", 0, stats=st); t1 = time.perf_counter()
    print(round(t1 - t0, 3), {k: (round(v, 3) if isinstance(v, float) else v) for k, v in st.items() if k in ("seconds", "GBps", "first_page_s", "corpus_create_s", "walk_done_s", "uploads_done_s")})
    t2 = time.perf_counter(); cc.close(); print("   close", round(time.perf_counter() - t2, 3))
