# where a one-query K = 100 call on the reference's bench shape spends its time (1 M x 1024, default n_clusters, nprobe 16)
import os, sys, time, numpy as np, torch
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import pq_vector_amd as pqv
n, dim, k, nprobe = 1_000_000, 1024, 100, 16
g = torch.Generator(device="cuda").manual_seed(42)
data = torch.rand((n, dim), generator=g, device="cuda", dtype=torch.float32)
corpus = pqv.Corpus.from_device_ptr(data.data_ptr(), n, dim, 0, keepalive=data)
index = pqv.IndexBuilder(corpus).build()
s = pqv.Searcher(index, corpus)
qs = torch.rand((64, dim), generator=g, device="cuda", dtype=torch.float32).cpu().numpy()
print(s.describe(1, k + 1, nprobe)[:400])
for rep in range(3):
    lat = []
    for i in range(64):
        t0 = time.perf_counter(); s.topk(qs[i:i + 1], k, nprobe); lat.append(time.perf_counter() - t0)
    print("Searcher.topk one query K=100: p50 %.1f us  min %.1f" % (np.median(lat) * 1e6, min(lat) * 1e6))
lat = []
for i in range(64):
    t0 = time.perf_counter(); s.topk(qs[i:i + 1], 10, nprobe); lat.append(time.perf_counter() - t0)
print("Searcher.topk one query K=10: p50 %.1f us" % (np.median(lat) * 1e6))
