#!/usr/bin/env python3
"""Per-wave phase records of ONE single-query call on a bench workload (phases build of the library):
   PQV_LIB_PATH=.../libpqv_hip_phases.so PQV_PHASES_OUT=out.bin python tools/phase_single.py c3 [nq]"""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import pq_vector_amd as pqv
from bench import WORKLOADS

wl = sys.argv[1] if len(sys.argv) > 1 else "c3"
nq = int(sys.argv[2]) if len(sys.argv) > 2 else 1
n, dim, kc, nprobe, _ = WORKLOADS[wl]
dev = torch.device("cuda", 0)
g = torch.Generator(device=dev); g.manual_seed(1234)
corpus_t = torch.empty((n, dim), dtype=torch.float32, device=dev)
step = max(1, (1 << 28) // (dim * 4))
for s in range(0, n, step):
    e = min(n, s + step)
    corpus_t[s:e] = torch.randint(0, 1 << 24, (e - s, dim), generator=g, device=dev, dtype=torch.int32).to(torch.float32) * (1.0 / (1 << 24))
gq = torch.Generator(device=dev); gq.manual_seed(7)
q_t = torch.randint(0, 1 << 24, (nq, dim), generator=gq, device=dev, dtype=torch.int32).to(torch.float32) * (1.0 / (1 << 24))
corpus = pqv.Corpus.from_device_ptr(corpus_t.data_ptr(), n, dim, device=0, keepalive=corpus_t)
index = pqv.IndexBuilder(corpus).n_clusters(kc).max_iters(20).seed(42).workers(os.cpu_count() or 1).build()
s = pqv.Searcher(index, corpus)
k = 10
r_t = torch.empty((nq, k), dtype=torch.int32, device=dev)
d_t = torch.empty((nq, k), dtype=torch.float32, device=dev)
s.topk_device(q_t.data_ptr(), nq, k, nprobe, r_t.data_ptr(), d_t.data_ptr(), 0, 0, stream=torch.cuda.current_stream().cuda_stream)
torch.cuda.synchronize()
print(s.describe(nq, k, nprobe))
print(s.counters())
