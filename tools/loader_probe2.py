# probe: Corpus.write_plain_pages over the reference-shaped file, runs of 48 pages, by thread count and PQV_LOADER_RAW
import numpy as np, pyarrow as pa, pyarrow.parquet as pq, time, os, sys
from concurrent.futures import ThreadPoolExecutor
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import pq_vector_amd as pqv
from pq_vector_amd import parquet_io
n, dim = 1000000, 1024
path = "/tmp/probe.parquet"
if not os.path.exists(path):
    vec = np.random.default_rng(1).random((n, dim), dtype=np.float32)
    col = pa.ListArray.from_arrays(pa.array(np.arange(0, (n + 1) * dim, dim, dtype=np.int32)), pa.array(vec.reshape(-1)))
    pq.write_table(pa.table({"embedding": col}), path, compression="NONE", use_dictionary=False, row_group_size=n)
    del vec, col
plan = parquet_io._plan_pages(path, "embedding", 8)
tasks = [t for t in plan.tasks]
for t in tasks: t[8] = None
for run_len in (16, 48):
    runs = [tasks[i:i + run_len] for i in range(0, len(tasks), run_len)]
    for thr in (1, 2, 4, 8):
        corpus = pqv.Corpus.create(n, dim, 0)
        do = parquet_io._plain_run_uploader(plan, corpus)
        t0 = time.perf_counter()
        with ThreadPoolExecutor(max_workers=thr) as ex:
            res = list(ex.map(do, runs))
        corpus.finish(n)
        dt = time.perf_counter() - t0
        print(f"raw={os.environ.get('PQV_LOADER_RAW','1')} run {run_len} threads {thr}: {dt:.3f} s  {4.096 / dt:.1f} GB/s  ok {all(r is not False for r in res)}")
        corpus.close()
