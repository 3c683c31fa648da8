#!/usr/bin/env python3
"""How many (query, probe) pairs hit each cluster of a bench workload, and what that means for the rows a batched
re-rank must stream: sum_c L_c * ceil(P_c / W) for quad widths W.   usage: tools/pair_stats.py c3 [nq]"""
import os, sys, json
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import pq_vector_amd as pqv
from bench import WORKLOADS

wl = sys.argv[1] if len(sys.argv) > 1 else "c3"
n, dim, kc, nprobe, nq = WORKLOADS[wl]
if len(sys.argv) > 2:
    nq = int(sys.argv[2])
dev = torch.device("cuda", 0)
g = torch.Generator(device=dev); g.manual_seed(1234)
corpus_t = torch.empty((n, dim), dtype=torch.float32, device=dev)
step = max(1, (1 << 28) // (dim * 4))
for s in range(0, n, step):
    e = min(n, s + step)
    corpus_t[s:e] = torch.randint(0, 1 << 24, (e - s, dim), generator=g, device=dev, dtype=torch.int32).to(torch.float32) * (1.0 / (1 << 24))
gq = torch.Generator(device=dev); gq.manual_seed(7)
q = (torch.randint(0, 1 << 24, (nq, dim), generator=gq, device=dev, dtype=torch.int32).to(torch.float32) * (1.0 / (1 << 24))).cpu().numpy()
corpus = pqv.Corpus.from_device_ptr(corpus_t.data_ptr(), n, dim, device=0, keepalive=corpus_t)
index = pqv.IndexBuilder(corpus).n_clusters(kc).max_iters(20).seed(42).workers(os.cpu_count() or 1).build()
off = index.list_offsets.astype(np.int64)
L = np.diff(off)
s = pqv.Searcher(index, corpus, pqv.PQV_LAYOUT_ROW_ORDER)
P = np.zeros(len(L), dtype=np.int64)
for i in range(nq):
    P[s.probe(q[i], nprobe)] += 1
out = {"workload": wl, "nq": nq, "clusters": int(len(L)), "list_len": {"min": int(L.min()), "mean": float(L.mean()), "max": int(L.max())},
       "pairs_per_cluster": {"max": int(P.max()), "mean": float(P.mean()), "p50": float(np.percentile(P, 50)), "p90": float(np.percentile(P, 90)),
                             "weighted_by_len": float((P * L).sum() / max(1, L[P > 0].sum()))},
       "rows_probed_once": int(L[P > 0].sum()), "pair_rows": int((P * L).sum()), "row_visits": {}}
for W in (16, 32, 64, 96, 128, 192, 256, 1 << 20):
    out["row_visits"][str(W)] = int((L * ((P + W - 1) // W)).sum())
print(json.dumps(out))
order = np.argsort(-L)[:12]
print("largest lists (len, pairs):", [(int(L[i]), int(P[i])) for i in order], file=sys.stderr)
