#!/usr/bin/env python3
"""pqv_rerank / pqv_rerank_device at the reference bench's RecordBatch shape (benches/query.rs:27-31: 2048-row batches of
1024-dim f32, K = 100) against the CPU oracle's update_topk_heap loop (oracle.topk_df, one thread like exec.rs:467).
Prints one JSON object.   usage: python tools/bench_rerank.py [n_batches]"""
import json, os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
import pq_vector_amd as pqv
from pq_vector_amd import _ffi
from oracle_binding import Oracle, build_oracle

nb = int(sys.argv[1]) if len(sys.argv) > 1 else 64
rows, dim, k = 2048, 1024, 100
rng = np.random.default_rng(1234)
emb = rng.random((nb * rows, dim), dtype=np.float32)
q = rng.random(dim, dtype=np.float32)
ids = np.arange(nb * rows, dtype=np.uint32)
out = {"shape": f"{nb} RecordBatches x {rows} rows x {dim} f32, k {k}, metric L2SQ_SEQ (exec.rs:529-533)"}
# host-buffer entry point (PCIe inclusive)
state = None
pqv.rerank_batch(q, emb[:rows], k, ids=ids[:rows])          # warm the pooled context
t0 = time.perf_counter()
for b in range(nb):
    state = pqv.rerank_batch(q, emb[b * rows:(b + 1) * rows], k, state=state, ids=ids[b * rows:(b + 1) * rows])
host_s = time.perf_counter() - t0
out["pqv_rerank_host_buffers"] = {"ms_per_batch": host_s / nb * 1e3, "rows_per_s": nb * rows / host_s,
                                   "GBps_incl_pcie": nb * rows * dim * 4 / host_s / 1e9}
# device-resident batches
dev = torch.device("cuda", 0)
emb_t = torch.from_numpy(emb).to(dev); ids_t = torch.from_numpy(ids.astype(np.int32)).to(dev); q_t = torch.from_numpy(q).to(dev)
io_r = torch.zeros((k,), dtype=torch.int32, device=dev); io_d = torch.zeros((k,), dtype=torch.float32, device=dev)
io_c = torch.zeros((1,), dtype=torch.int32, device=dev)
st = torch.cuda.current_stream().cuda_stream
def run():
    io_c.zero_()
    for b in range(nb):
        rc = _ffi.lib().pqv_rerank_device(0, _ffi.vp(q_t.data_ptr()), _ffi.vp(emb_t[b * rows:(b + 1) * rows].data_ptr()),
                                          _ffi.vp(ids_t[b * rows:(b + 1) * rows].data_ptr()), rows, dim, k, pqv.PQV_L2SQ_SEQ,
                                          _ffi.vp(io_r.data_ptr()), _ffi.vp(io_d.data_ptr()), _ffi.vp(io_c.data_ptr()), _ffi.vp(st))
        assert rc == 0
    torch.cuda.synchronize()
run()
t0 = time.perf_counter(); run(); dev_s = time.perf_counter() - t0
out["pqv_rerank_device"] = {"ms_per_batch": dev_s / nb * 1e3, "rows_per_s": nb * rows / dev_s,
                            "GBps": nb * rows * dim * 4 / dev_s / 1e9,
                            "note": "one call per RecordBatch, each host-synchronised (launch-latency bound at 8 MB per batch)"}
# CPU oracle, one thread
build_oracle("native")
o = Oracle(native=True)
t0 = time.perf_counter(); orow, od2 = o.topk_df(emb, ids, q, k); cpu_s = time.perf_counter() - t0
out["cpu_oracle_update_topk_heap"] = {"ms_per_batch": cpu_s / nb * 1e3, "rows_per_s": nb * rows / cpu_s, "threads": 1,
                                      "host_cpus": os.cpu_count()}
out["parity"] = {"host_path_identical": bool((state[0] == orow).all() and (state[1].view(np.uint32) == od2.view(np.uint32)).all()),
                 "device_path_identical": bool((io_r.cpu().numpy().view(np.uint32) == orow).all()
                                               and (io_d.cpu().numpy().view(np.uint32) == od2.view(np.uint32)).all())}
print(json.dumps(out))
