#!/usr/bin/env python3
"""bench.py -- top-k QPS (k=10) of the MI355X IVF hot path on BASELINE.json's configs.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--workload c3|c2|c4|c5|...] [--nq Q]

A "step" is one pass of the hot path (centroid probe -> candidate re-rank -> top-k merge)
over one batch of Q synthetic queries, with corpus, index and queries already resident in
HBM.

N = 1 default workload = BASELINE.json configs[2] (C3), the largest single-GPU configuration:
10 M x 768 uniform f32, n_clusters 1024, k 10, nprobe 32, 1024 queries per step.  Without
--steps the step count is calibrated so that the timed region lasts >= 1 s; with --steps K
exactly K steps are timed.

N > 1 (launched by torch.distributed.run, one rank per GPU) default workload = configs[3]
(C4): every rank holds one 12.5 M x 768 shard of the corpus and its own IVF index (the
reference's per-file index, src/df_vector/index_exec.rs:85-164); every rank searches the
whole query batch on its shard, one RCCL all-gather per step moves k x {distance, row} per
query, and every rank runs the same deterministic merge by (distance, shard, position).
Per-rank work is fixed as N grows ("scaling": "weak"); the job then answers the same
queries/s over an N times larger corpus.  `--multi replica` instead replicates a corpus that
fits one GPU and shards the QUERIES (no data-path collective); it is also timed as a
secondary object on every N > 1 run.

Prints ONE JSON line on rank 0 (contract in the task brief) with `roofline` and
`cpu_baseline` objects; exits non-zero if the parity check inside the CPU leg fails.
torch is plumbing here: device tensors, streams, RCCL.
"""
import argparse
import json
import math
import os
import re
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
# the stream lanes of a step must land on different hardware queues to overlap (new_streams); the ROCm runtime reads this once
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

WORKLOADS = {
    # name: (rows, dim, n_clusters, nprobe, default queries per step)
    "c2": (1_000_000, 128, 100, 8, 1024),
    "c3": (10_000_000, 768, 1024, 32, 1024),
    "c4": (12_500_000, 768, 1024, 32, 1024),   # PER-RANK shard of the 100M x 768 corpus (weak scaling)
    "c5": (10_000_000, 1536, 0, 0, 1024),      # brute-force cosine on the matrix cores (no index)
    "c5s": (1_000_000, 1536, 0, 0, 1024),      # same, 1 M rows (quick check)
    "refbench": (1_000_000, 1024, 0, 16, 1024),  # the reference's benches/query.rs:27-31 shape; use --k 100
    "c1": (1024, 4096, 0, 5, 64),          # vldb stand-in: n_clusters = ceil(sqrt(n)) = 32
    "c2s2": (500_000, 128, 100, 8, 1024),   # C2-sized shards (tuning aid)
    "c2s4": (250_000, 128, 100, 8, 1024),
    "c2s8": (125_000, 128, 100, 8, 1024),
    "c3s": (1_000_000, 768, 1024, 32, 1024),   # C3's parameters on 1 M rows (quick check)
    "c3m": (2_500_000, 768, 1024, 32, 1024),   # C3's parameters on 2.5 M rows: 2441 rows per list
    "c3l": (5_000_000, 768, 1024, 32, 1024),   # ... on 5 M rows: 4883 rows per list
    "c3h": (500_000, 768, 1024, 32, 1024),     # short lists: 488 rows per list (the reference's default n_clusters = ceil(sqrt(n)) regime)
    "c2d": (1_000_000, 128, 0, 8, 1024),       # C2's rows with the DEFAULT n_clusters = ceil(sqrt(n)) = 1000 (index.rs:161-167)
    "tiny": (20_000, 64, 16, 4, 64),       # plumbing check
}
K = 10
# Stream lanes the steps alternate between (--streams): the library keeps one scratch set per stream (four of them), so step i + 1's
# probe / bucketing / seed kernels and the head of its screen run in the tails of the steps before it.  Round 5, 8 hardware queues:
# 2 / 3 / 4 lanes = C3 583 / 601 / 610 k q/s, C2 7.8 / - / 9.6 M, refbench 848 / - / 962 k, mixture 502 / - / 531 k; 5, 6, 8 are
# slower again (a fifth stream evicts a scratch set, which orders it behind that set's last call).
N_LANES = 4
HBM_PEAK_GBS = 8000.0                      # MI355X_MICROARCH.md: HBM3E 8 TB/s spec (6.29 TB/s measured copy ceiling)
PROFILE_ROUND = "r06"


def log(*a):
    print(*a, file=sys.stderr, flush=True)


def synth(torch, dev, seed, rows, dim):
    """The reference's bench recipe (benches/bench_util.rs:12-64): i.i.d. uniform [0,1) f32 with 24-bit
    resolution, generated on the device."""
    g = torch.Generator(device=dev)
    g.manual_seed(seed)
    out = torch.empty((rows, dim), dtype=torch.float32, device=dev)
    step = max(1, (1 << 28) // (dim * 4))
    for s in range(0, rows, step):
        e = min(rows, s + step)
        out[s:e] = torch.randint(0, 1 << 24, (e - s, dim), generator=g, device=dev, dtype=torch.int32).to(torch.float32) * (1.0 / (1 << 24))
    return out


def synth_mixture(torch, dev, seed, rows, dim, n_centres, sigma=0.1, centre_seed=99):
    """SURVEY 8(d)'s secondary data set: a Gaussian mixture -- n_centres component centres uniform in [0,1)^dim (the
    same for corpus and queries), every row a uniformly chosen centre + N(0, sigma^2) noise -- so that the inverted
    lists follow real structure and IVF recall means something.  Generated on the device in slabs."""
    gc = torch.Generator(device=dev)
    gc.manual_seed(centre_seed)
    centres = torch.rand((n_centres, dim), generator=gc, device=dev, dtype=torch.float32)
    g = torch.Generator(device=dev)
    g.manual_seed(seed)
    out = torch.empty((rows, dim), dtype=torch.float32, device=dev)
    step = max(1, (1 << 27) // (dim * 4))
    for s in range(0, rows, step):
        e = min(rows, s + step)
        comp = torch.randint(0, n_centres, (e - s,), generator=g, device=dev)
        out[s:e] = centres[comp] + sigma * torch.randn((e - s, dim), generator=g, device=dev, dtype=torch.float32)
    return out


def pmc_traffic(workload, kernels):
    """HBM-side bytes per launch of the dominant kernels from the committed rocprofv3 PMC summaries of THIS
    workload (profiles/<round>_<workload>_pmc_{FETCH,WRITE}_SIZE.txt, written by tools/profile_round.sh from
    separate --pmc passes): FETCH_SIZE x 2 (the gfx950 half-reporting of 16-byte coalesced reads,
    MI355X_MICROARCH.md #HBM) + WRITE_SIZE, KiB -> bytes, summed over the named kernels.  `kernels`: exact
    instantiation names as the library's dispatch description gives them, optionally "name@grid_x_threads"
    (the index build launches some of the same templates with other grids)."""
    for rnd in (PROFILE_ROUND, "r05"):          # this round's summaries; the previous round's while the kernels keep their names
        got = _pmc_traffic_round(rnd, workload, kernels)
        if got[0] is not None:
            return got
    return None, None


def _pmc_traffic_round(rnd, workload, kernels):
    total, srcs = 0.0, []
    for ctr, mult in (("FETCH_SIZE", 2.0), ("WRITE_SIZE", 1.0)):
        path = os.path.join(ROOT, "profiles", f"{rnd}_{workload}_pmc_{ctr}.txt")
        if not os.path.exists(path):
            return None, None
        got = set()
        for line in open(path):
            f = line.split()
            if ctr not in f:
                continue
            i = f.index(ctr)
            for kname in kernels:
                name, _, grid = kname.partition("@")
                if line.startswith("pqv::" + name + " ") and (not grid or f[i - 1].split("x")[0] == grid):
                    total += mult * float(f[i + 2]) * 1024.0          # counter, dispatches, avg_value, avg_us
                    got.add(kname)
        if len(got) != len(kernels):
            return None, None
        srcs.append(os.path.relpath(path, ROOT))
    return total, srcs


class OracleParity:
    """Bit-for-bit comparison of GPU answers with the CPU oracle (oracle/: test infrastructure -- the checker, never the
    thing measured).  The oracle index is parsed from the GPU-built blob (IvfIndex::from_bytes), the rows are a host copy
    of the very matrix the GPU searched; one query per host thread (results are independent per query)."""

    def __init__(self, blob, host_rows, native=True):
        from oracle_binding import Oracle, build_oracle
        if native:
            build_oracle("native")
        self.oracle = Oracle(native=native)
        self.oidx = self.oracle.index_from_bytes(blob)
        self.host = host_rows
        self.st = {"ids": True, "dist": True, "tie": True, "ncand": True, "groups": 0, "replayed": 0, "replay_ok": True, "checked": 0}

    def one(self, q, k, nprobe):
        return self.oidx.topk_batch(self.host, q.reshape(1, -1), k, nprobe)

    def compare(self, qs, sel, grows, gdist, k, nprobe, searcher=None, gncand=None, threads=None):
        """qs [nq, dim] host queries; sel: indices into the batch to check; grows u32 / gdist f32 [nq, k] GPU answers."""
        from concurrent.futures import ThreadPoolExecutor
        sel = list(sel)
        if not sel:
            return
        nthr = max(1, min(len(sel), threads or (os.cpu_count() or 1), 256))
        with ThreadPoolExecutor(max_workers=nthr) as ex:
            res = list(ex.map(lambda q: self.one(qs[q], k, nprobe), sel))
        st = self.st
        for q, (orows, odist, onf, onc) in zip(sel, res):
            st["checked"] += 1
            g, gd = grows[q], gdist[q]
            st["dist"] &= bool((odist[0].view(np.uint32) == gd.view(np.uint32)).all())
            if gncand is not None:
                st["ncand"] &= int(onc[0]) == int(gncand[q])
            if (orows[0] == g).all():
                continue
            st["ids"] = False
            # pqv_topk_device orders equal output distances by (d2, position); Rust orders them by heap history (the
            # host API pqv_topk replays that exactly).  Inside a group of equal distance the id SETS must still agree.
            if searcher is not None:
                hr, hd, _, _ = searcher.topk(qs[q:q + 1], k, nprobe)
                st["replayed"] += 1
                st["replay_ok"] &= bool((hr[0] == orows[0]).all()) and bool((hd.view(np.uint32)[0] == odist.view(np.uint32)[0]).all())
            j = 0
            while j < k:
                e = j
                while e + 1 < k and odist[0, e + 1] == odist[0, j]:
                    e += 1
                if e > j:
                    st["groups"] += 1
                st["tie"] &= sorted(orows[0, j:e + 1].tolist()) == sorted(g[j:e + 1].tolist())
                j = e + 1

    def ok(self):
        st = self.st
        return bool(st["dist"] and st["tie"] and st["ncand"] and (st["ids"] or (st["replayed"] > 0 and st["replay_ok"])))

    def record(self):
        st = self.st
        return {"checker": "CPU oracle (oracle/pqv_oracle.c), index parsed from the GPU-built blob, rows = host copy of the searched matrix",
                "queries_checked": st["checked"], "row_idx_identical": st["ids"], "dist_bit_identical": st["dist"],
                "n_candidates_identical": st["ncand"],
                "row_idx_identical_up_to_order_inside_equal_distance_groups": st["tie"],
                "equal_distance_groups_seen": st["groups"], "queries_replayed_through_pqv_topk": st["replayed"],
                "row_idx_identical_after_replay": bool(st["ids"] or (st["replayed"] > 0 and st["replay_ok"])),
                "ok": self.ok()}


_HIP = None
_LANES = {}


def new_streams(torch, dev, n):
    """n HIP streams of their own (hipStreamCreateWithFlags, non-blocking -- what the library creates for itself), wrapped for
    torch.  Explicit handles: torch's DEFAULT stream has handle 0, which the ABI reads as "the searcher's own stream", so the
    torch ops that follow a call on the default stream (the exchange's pack above all) would not be ordered behind it.
    Whether two streams overlap at all depends on the hardware queues the runtime maps them to: with its default of 4 queues
    (GPU_MAX_HW_QUEUES) the same two lanes measured 465 k or 515 k q/s on C3 depending on how many other streams the process
    had created before them; bench.py therefore asks for 8 queues (set below, before the runtime starts; recorded in the line)."""
    global _HIP
    import ctypes
    if _HIP is None:
        _HIP = ctypes.CDLL("libamdhip64.so")
    # ONE set of lanes per process and device: every configuration of the run times on the same streams (a pair created later in
    # the process -- after the loaders', the exchange's and torch's own streams -- measured 5.7 M q/s on C2 where the first
    # pair gives 7.6 M: which hardware queue a stream lands on depends on how many were created before it)
    cache = _LANES.setdefault(str(dev), [])
    if len(cache) >= n:
        return cache[:n]
    cand = list(cache)
    with torch.cuda.device(dev):
        for _ in range(max(6, n + 2) - len(cache)):
            h = ctypes.c_void_p()
            if _HIP.hipStreamCreateWithFlags(ctypes.byref(h), 1) != 0 or not h.value:     # hipStreamNonBlocking
                cand.append(torch.cuda.Stream(device=dev))
            else:
                cand.append(torch.cuda.ExternalStream(h.value, device=dev))
        # keep streams that really run side by side: a one-block spin kernel on each of two streams takes one kernel's time if
        # they sit on different hardware queues and two if they share one
        def together(a, b):
            best = 1e9
            for _ in range(3):
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                for s_ in (a, b):
                    with torch.cuda.stream(s_):
                        torch.cuda._sleep(400_000)
                torch.cuda.synchronize()
                best = min(best, time.perf_counter() - t0)
            return best
        alone = together(cand[0], cand[0]) / 2.0
        out = list(cache) if cache else [cand[0]]
        for c in cand:
            if len(out) >= n:
                break
            if any(c is o for o in out):
                continue
            if all(together(o, c) < 1.5 * alone for o in out):
                out.append(c)
        for c in cand:                       # (not enough independent ones: take what there is)
            if len(out) >= n:
                break
            if not any(c is o for o in out):
                out.append(c)
    _LANES[str(dev)] = out
    return out


def spread(nq, m):
    """m query indices spread over a batch of nq (first, last and evenly between)."""
    m = min(m, nq)
    return sorted(set(int(round(i * (nq - 1) / max(1, m - 1))) for i in range(m)))


def ivf_measure(pqv, torch, dev, searcher, index, queries_t, nq, k, nprobe, dim, min_time=0.4, steps=20, exchange=None):
    """Timed blocks of `steps` steps over N_LANES stream lanes (median block), then a serial pass with HIP events around the
    re-rank kernels; min_bytes of the step (see roofline.min_bytes_definition).  Returns (record, lane-0 outputs)."""
    # (explicit streams only: new_streams)
    torch.cuda.synchronize()
    NL = N_LANES
    streams = new_streams(torch, dev, NL)
    rows = [torch.empty((nq, k), dtype=torch.int32, device=dev) for _ in range(NL)]
    dd = [torch.empty((nq, k), dtype=torch.float32, device=dev) for _ in range(NL)]
    nc = [torch.empty((nq,), dtype=torch.int64, device=dev) for _ in range(NL)]

    def st(i):
        s_ = streams[i % NL]
        with torch.cuda.stream(s_):
            searcher.topk_device(queries_t.data_ptr(), nq, k, nprobe, rows[i % NL].data_ptr(), dd[i % NL].data_ptr(), 0, nc[i % NL].data_ptr(),
                                 stream=s_.cuda_stream)
            if exchange is not None:
                exchange[i % NL].exchange_u32(dd[i % NL], rows[i % NL])

    for i in range(2 * NL):
        st(i)
    torch.cuda.synchronize()
    c0 = searcher.counters()
    blocks = []
    while sum(blocks) < min_time and len(blocks) < 200:
        t0 = time.perf_counter()
        for i in range(steps):
            st(i)
        torch.cuda.synchronize()
        blocks.append(time.perf_counter() - t0)
    c1 = searcher.counters()
    el = float(np.median(blocks))
    # serial pass: isolated kernel durations (HIP events recorded by the library on the call's stream)
    searcher.set_timing(True)
    torch.cuda.synchronize()
    ns = 10
    t1 = time.perf_counter()
    for _ in range(ns):
        st(0)
        torch.cuda.synchronize()
    serial_ms = (time.perf_counter() - t1) / ns * 1e3
    searcher.set_timing(False)
    rr, tot, ncalls = searcher.timing_read()
    k_ms = rr / max(1, ncalls)
    plan_text = searcher.describe(nq, k, nprobe)
    # min_bytes: the operand image of every DISTINCT probed row once + 8 bytes per row + the f32 row of every survivor
    qs_host = queries_t.cpu().numpy()
    lens = np.diff(index.list_offsets.astype(np.int64))
    probed = np.zeros(len(lens), dtype=bool)
    for i in (range(nq) if nq <= 256 else range(0, nq, max(1, nq // 256))):
        probed[searcher.probe(qs_host[i], nprobe)] = True
    distinct_rows = int(lens[probed].sum())
    nqs = max(1, c1["queries"] - c0["queries"])
    surv = (c1["screen_survivors"] - c0["screen_survivors"]) / nqs * nq
    wide = "wide_filter_kernel" in plan_text
    opb = 1 if "int8 screen operands" in plan_text else 2 if "f16 screen operands" in plan_text else 4
    cand_rows = int(nc[0].sum().item())
    if wide:
        min_bytes = distinct_rows * (opb * dim + 8) + surv * 4 * dim
    else:
        min_bytes = cand_rows * (4 * dim + 4) if "stream_kernel" in plan_text else distinct_rows * (4 * dim + 4)
    rec = {"value": nq * steps / el, "unit": "queries/s", "ms_per_step": el / steps * 1e3, "steps": steps, "repeats": len(blocks),
           "ms_per_step_serial": serial_ms,
           "roofline": {"bound": "hbm", "kernel_ms": k_ms, "min_bytes": min_bytes,
                        "min_bytes_frac": (min_bytes / (k_ms * 1e-3) / 1e9 / HBM_PEAK_GBS) if k_ms > 0 else None,
                        "achieved": (min_bytes / (k_ms * 1e-3) / 1e9) if k_ms > 0 else None, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                        "basis": "min_bytes (operand image of every distinct probed row once + 8 B per row + 4 dim per survivor) / HIP-event "
                                 "time of the re-rank kernels of a serially issued step"},
           "candidates_per_query": cand_rows / nq,
           "screen_survivors_per_query": (c1["screen_survivors"] - c0["screen_survivors"]) / nqs,
           "screened_rows_per_query": (c1["screened_pairs"] - c0["screened_pairs"]) / nqs,
           "dispatch": plan_text}
    st(0)
    torch.cuda.synchronize()
    return rec, (rows[0], dd[0], nc[0], qs_host)


def ivf_config(args, pqv, torch, dev, local_rank, name, k, data="uniform", parity_queries=64, rccl=False, mixture_centres=0, recall=0):
    """One IVF configuration end to end for the `configs` object of the default line: synthetic corpus on the device, index
    build, searcher, timed steps, roofline on min_bytes, and the oracle check of `parity_queries` queries of the timed
    batch (host copy of the corpus downloaded once)."""
    n, dim, kc, nprobe, nq = WORKLOADS[name]
    t_all = time.perf_counter()
    if data == "mixture":
        corpus_t = synth_mixture(torch, dev, 1234, n, dim, mixture_centres or kc or 1024)
        q_t = synth_mixture(torch, dev, 7, nq, dim, mixture_centres or kc or 1024)
    else:
        corpus_t = synth(torch, dev, 1234, n, dim)
        q_t = synth(torch, dev, 7, nq, dim)
    torch.cuda.synchronize()
    corpus = pqv.Corpus.from_device_ptr(corpus_t.data_ptr(), n, dim, device=local_rank, keepalive=corpus_t)
    t0 = time.perf_counter()
    b = pqv.IndexBuilder(corpus).max_iters(20).seed(42).workers(os.cpu_count() or 1)
    index = b.n_clusters(kc).build() if kc else b.build()
    build_s = time.perf_counter() - t0
    t0 = time.perf_counter()
    srch = pqv.Searcher(index, corpus)
    create_s = time.perf_counter() - t0
    xchg = None
    if rccl:
        import torch.distributed as dist
        from pq_vector_amd.sharding import ShardExchange
        if not dist.is_initialized():
            os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
            os.environ.setdefault("MASTER_PORT", "29531")
            dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
        xchg = [ShardExchange(1, nq, k, dev, always_collective=True, row_bases=[0]) for _ in range(N_LANES)]
    rec, (rows_t, dist_t, nc_t, qs_host) = ivf_measure(pqv, torch, dev, srch, index, q_t, nq, k, nprobe, dim, exchange=xchg)
    rec.update({"config": f"{name}: {n}x{dim} {data} f32, n_clusters {index.n_clusters}, k {k}, nprobe {nprobe}, {nq} queries/step"
                          + (", one shard on one rank: RCCL all-gather (1 rank) + device merge inside every step" if rccl else ""),
                "index_build_s": build_s, "index_build_vectors_per_s": n / build_s, "searcher_create_s": create_s,
                "time_to_first_query_s": build_s + create_s})
    if rccl:
        out_d, out_r = xchg[0].exchange_u32(dist_t, rows_t)
        torch.cuda.synchronize()
        rec["exchange_check"] = bool(torch.equal(out_d, dist_t) and torch.equal(out_r, rows_t.to(torch.int64) & 0xFFFFFFFF))
    if recall:
        m = min(recall, nq)
        br, _, _ = corpus.brute_topk(qs_host[:m], k, pqv.PQV_L2SQ_MFMA)
        got = rows_t[:m].cpu().numpy().view(np.uint32)
        rec["recall_at_k"] = sum(len(set(got[i].tolist()) & set(br[i].tolist())) for i in range(m)) / float(m * k)
    if parity_queries and not args.no_cpu:
        t0 = time.perf_counter()
        host = corpus_t.cpu().numpy()
        par = OracleParity(index.to_bytes(), host)
        par.compare(qs_host, spread(nq, parity_queries), rows_t.cpu().numpy().view(np.uint32), dist_t.cpu().numpy(), k, nprobe,
                    searcher=srch, gncand=nc_t.cpu().numpy())
        rec["parity"] = par.record()
        rec["parity"]["seconds"] = time.perf_counter() - t0
        del host, par
    rec["seconds"] = time.perf_counter() - t_all
    srch.close(); corpus.close()
    del corpus_t
    torch.cuda.empty_cache()
    return rec


def from_parquet_config(args, pqv, torch, dev, local_rank, name="refbench", k=100, n_queries=20):
    """The reference's own end-to-end benches on its own shape (benches/index_build.rs:43-50: IndexBuilder::new(path, "embedding")
    .build_inplace(); benches/query.rs: TopkBuilder over the indexed file): the synthetic column is written as List<f32> with an
    Int32 id column in ONE row group (what ArrowWriter's defaults give benches/bench_util.rs:12-58 for 1 M rows), then
      build_inplace   = Parquet column -> HBM (N1: reader threads + pinned staging + async DMA) + index build + blob appended to the file
      first search    = read the blob back, load the column, create the searcher, answer one query
      warm searches   = the cached resident searcher (the reference re-reads blob and candidate rows on every call, search.rs:89-110)
    next to the CPU oracle on the same host (its k-means / assignment threads = the reference's worker chunks), whose blob and
    answers the GPU's must equal."""
    import tempfile
    import pyarrow as pa
    import pyarrow.parquet as pq
    n, dim, kc, nprobe, _ = WORKLOADS[name]
    t_all = time.perf_counter()
    corpus_t = synth(torch, dev, 1234, n, dim)
    q_t = synth(torch, dev, 7, n_queries, dim)
    host = corpus_t.cpu().numpy()
    qs = q_t.cpu().numpy()
    del corpus_t
    torch.cuda.empty_cache()
    tmp = tempfile.mkdtemp(prefix="pqv_bench_")
    path = os.path.join(tmp, "bench.parquet")
    t0 = time.perf_counter()
    col = pa.ListArray.from_arrays(pa.array(np.arange(0, (n + 1) * dim, dim, dtype=np.int64 if (n + 1) * dim > 2**31 - 1 else np.int32)),
                                   pa.array(host.reshape(-1)))
    if pa.types.is_large_list(col.type):
        col = col.cast(pa.list_(pa.field("item", pa.float32())))
    table = pa.table({"id": pa.array(np.arange(n, dtype=np.int32)), "embedding": col})
    pq.write_table(table, path, row_group_size=max(n, 1 << 20), compression="NONE", use_dictionary=False)
    write_s = time.perf_counter() - t0
    del table, col
    file_bytes = os.path.getsize(path)
    rec = {"config": f"{name}: {n}x{dim} uniform f32 as List<f32> 'embedding' + Int32 'id', {pq.ParquetFile(path).metadata.num_row_groups} row group(s), "
                     f"uncompressed; default n_clusters, max_iters 20, seed 42; k {k}, nprobe {nprobe}",
           "file_bytes": file_bytes, "synthetic_file_write_s": write_s}
    try:
        b = pqv.IndexBuilder(path, "embedding", device=local_rank).workers(os.cpu_count() or 1)
        t0 = time.perf_counter()
        index = b.build_inplace()
        total = time.perf_counter() - t0
        st = b.last_stats
        rec["build_inplace"] = {"seconds": total, "vectors_per_s": n / total, "load_s": st["load_s"], "build_s": st["build_s"], "append_s": st["write_s"],
                                "loader": st["load"], "pcie_peak_GBps": 63.0,
                                "loader_note": "data pages walked in the memory-mapped file (level runs checked, PLAIN values copied from the page cache) -> pinned staging -> hipMemcpyAsync, 8 threads; anything else through pyarrow record batches; GB/s counts the f32 payload"}
        blob = index.to_bytes()
        t0 = time.perf_counter()
        first = pqv.TopkBuilder(path, qs[0], device=local_rank).k(k).nprobe(nprobe).search()
        first_s = time.perf_counter() - t0
        lat, answers = [], [first]
        for i in range(1, n_queries):
            t1 = time.perf_counter()
            answers.append(pqv.TopkBuilder(path, qs[i], device=local_rank).k(k).nprobe(nprobe).search())
            lat.append(time.perf_counter() - t1)
        rec["topk_builder"] = {"first_search_s": first_s, "warm_p50_us": float(np.percentile(np.array(lat) * 1e6, 50)),
                               "warm_calls": len(lat), "k": k, "nprobe": nprobe,
                               "note": "first = read_index_from_parquet + column load + searcher creation + the query; warm = the resident searcher, one query per call"}
        if not args.no_cpu:
            from oracle_binding import Oracle, build_oracle
            build_oracle("native")
            o = Oracle(native=True)
            t0 = time.perf_counter()
            oidx = o.build_index(host, n_clusters=0, max_iters=20, seed=42, workers=os.cpu_count() or 1)
            cpu_build = time.perf_counter() - t0
            same_blob = oidx.to_bytes() == blob
            t0 = time.perf_counter()
            ok = True
            for i in range(n_queries):
                orows, odist, onf, _ = oidx.topk_batch(host, qs[i:i + 1], k, nprobe)
                got = answers[i]
                ok &= len(got) == int(onf[0]) and all(got[j].row_idx == int(orows[0, j]) for j in range(len(got))) and \
                    all(np.float32(got[j].distance).view(np.uint32) == odist[0, j].view(np.uint32) for j in range(len(got)))
            cpu_q = (time.perf_counter() - t0) / n_queries
            rec["cpu_oracle"] = {"build_s_in_memory": cpu_build, "build_threads": os.cpu_count(), "query_s_in_memory_1_thread": cpu_q,
                                 "note": "the oracle on the in-memory column (no Parquet I/O on its side): build with the reference's worker-chunk threads, "
                                         "one query on one thread (search.rs:115)"}
            rec["parity"] = {"checker": "CPU oracle at full size", "index_blob_identical": bool(same_blob), "queries_checked": n_queries,
                             "topk_rows_and_distance_bits_identical": bool(ok), "ok": bool(same_blob and ok)}
    finally:
        try:
            os.remove(path); os.rmdir(tmp)
        except OSError:
            pass
        from pq_vector_amd import api as _api
        _api._PATH_SEARCHERS.clear()
    rec["seconds"] = time.perf_counter() - t_all
    return rec


def c1_config(args, pqv, torch, dev, local_rank, n_queries=64, k=10, keep_dir=None):
    """BASELINE configs[0] as BASELINE.md defines its stand-in (data/vldb_2025.parquet is absent): 1 024 x 4096 `embedding`
    List<f32> + Utf8 `title` in a Parquet file, IndexBuilder(path, "embedding").build_inplace() with the DEFAULT n_clusters
    (ceil(sqrt(1024)) = 32, src/ivf/index.rs:161-167), then TopkBuilder(path, q).k(10).nprobe(5).search() per query
    (src/ivf/search.rs:49-81; the vldb test's query = a row of the file, src/df_vector/tests.rs:128).  Blob and every answer
    (row ids + distance bits) must equal the CPU oracle's."""
    import tempfile
    import pyarrow as pa
    import pyarrow.parquet as pq
    n, dim, _, nprobe, _ = WORKLOADS["c1"]
    t_all = time.perf_counter()
    host = synth(torch, dev, 1234, n, dim).cpu().numpy()
    # queries: rows of the file (the reference's vldb test) and fresh vectors, alternating
    fresh = synth(torch, dev, 7, n_queries, dim).cpu().numpy()
    qs = np.ascontiguousarray(np.where((np.arange(n_queries) % 2 == 0)[:, None], host[(np.arange(n_queries) * 13) % n], fresh), dtype=np.float32)
    tmp = keep_dir or tempfile.mkdtemp(prefix="pqv_bench_c1_")
    path = os.path.join(tmp, "vldb_standin.parquet")
    col = pa.ListArray.from_arrays(pa.array(np.arange(0, (n + 1) * dim, dim, dtype=np.int32)), pa.array(host.reshape(-1)))
    table = pa.table({"title": pa.array([f"paper {i}" for i in range(n)], type=pa.utf8()), "embedding": col})
    pq.write_table(table, path)                       # writer defaults: snappy, dictionary pages where they pay
    rec = {"config": f"c1: {n}x{dim} uniform f32 as List<f32> 'embedding' + Utf8 'title' (vldb stand-in), default n_clusters, k {k}, nprobe {nprobe}, "
                     f"IndexBuilder(path).build_inplace() + TopkBuilder(path, q) per query"}
    try:
        b = pqv.IndexBuilder(path, "embedding", device=local_rank).workers(os.cpu_count() or 1)
        t0 = time.perf_counter()
        index = b.build_inplace()
        rec["build_inplace_s"] = time.perf_counter() - t0
        rec["n_clusters"] = int(index.n_clusters)
        blob = index.to_bytes()
        back, colname = pqv.read_index_from_parquet(path)
        answers, lat = [], []
        for i in range(n_queries):
            t1 = time.perf_counter()
            answers.append(pqv.TopkBuilder(path, qs[i], device=local_rank).k(k).nprobe(nprobe).search())
            lat.append(time.perf_counter() - t1)
        warm = np.array(lat[1:]) * 1e6
        rec.update({"first_search_s": lat[0], "topk_builder_warm_p50_us": float(np.percentile(warm, 50)),
                    "value": float(1e6 / warm.mean()), "unit": "queries/s", "ms_per_step": float(warm.mean() * 1e-3)})
        titles = pq.read_table(path, columns=["title"]).column("title").to_pylist()      # the file still reads as Parquet
        ok_file = len(titles) == n and titles[5] == "paper 5" and colname == "embedding" and back.to_bytes() == blob
        if not args.no_cpu:
            from oracle_binding import Oracle
            o = Oracle()
            oidx = o.build_index(host, n_clusters=0, max_iters=20, seed=42, workers=os.cpu_count() or 1)
            same_blob = oidx.to_bytes() == blob
            ok = True
            for i in range(n_queries):
                orows, odist, onf, _ = oidx.topk_batch(host, qs[i:i + 1], k, nprobe)
                got = answers[i]
                ok &= len(got) == int(onf[0]) and all(got[j].row_idx == int(orows[0, j]) for j in range(len(got))) and \
                    all(np.float32(got[j].distance).view(np.uint32) == odist[0, j].view(np.uint32) for j in range(len(got)))
            rec["parity"] = {"checker": "CPU oracle at full size", "index_blob_identical": bool(same_blob), "queries_checked": n_queries,
                             "topk_rows_and_distance_bits_identical": bool(ok), "file_round_trip": bool(ok_file),
                             "ok": bool(same_blob and ok and ok_file)}
    finally:
        if keep_dir is None:
            try:
                os.remove(path); os.rmdir(tmp)
            except OSError:
                pass
        from pq_vector_amd import api as _api
        _api._PATH_SEARCHERS.clear()
    rec["seconds"] = time.perf_counter() - t_all
    return rec


def from_parquet_sharded(args, pqv, torch, dist, dev, local_rank, rank, world, real_stdout):
    """BASELINE configs[3]'s partition unit end to end (`--gpus N --from-parquet`): ONE Parquet file on the node, one row-group
    RANGE per rank.  Rank 0 writes the synthetic file (uneven row groups, so the cuts are not trivially equal); every rank
    computes every cut from the footer alone (sharding.shard_row_groups: row-group boundaries, row_base = prefix sum of the row
    groups before the range -- src/df_vector/access.rs:128-144), loads ITS range through the page walker
    (parquet_io.load_embedding_column(row_groups=...)), builds its own index (the reference's per-file index,
    src/df_vector/index_exec.rs:85-164) and searches the whole batch; one all-gather + the deterministic merge per step.
    Parity leg: with every list probed the merged answer must carry the same FILE-GLOBAL row ids and distance bits as a
    single-shard search of the whole file on rank 0 (files up to 48 GB of f32; skipped and said so beyond)."""
    import shutil
    import tempfile
    import pyarrow as pa
    import pyarrow.parquet as pq
    from pq_vector_amd.sharding import ShardExchange, load_parquet_shard, shard_row_groups
    _, dim, kc, nprobe, nq_default = WORKLOADS["c4"]
    rpr = args.rows_per_rank or 1_000_000
    n_total = rpr * world
    nq = args.nq or nq_default
    k = K
    rg_rows = max(1, int(n_total / (4 * world + 0.5)))     # 4 full row groups per rank + a short last one: the cuts are not at n / world
    t_all = time.perf_counter()
    path_box = [None]
    tmp = None
    if rank == 0:
        tmp = tempfile.mkdtemp(prefix="pqv_bench_shards_")
        path_box[0] = os.path.join(tmp, "shared.parquet")
        t0 = time.perf_counter()
        schema = pa.schema([("id", pa.int32()), ("embedding", pa.list_(pa.field("item", pa.float32())))])
        w = pq.ParquetWriter(path_box[0], schema, compression="NONE", use_dictionary=False)
        at, g = 0, 0
        while at < n_total:
            m = min(rg_rows, n_total - at)
            host = synth(torch, dev, 1234 + g, m, dim).cpu().numpy()
            col = pa.ListArray.from_arrays(pa.array(np.arange(0, (m + 1) * dim, dim, dtype=np.int64 if (m + 1) * dim > 2**31 - 1 else np.int32)),
                                           pa.array(host.reshape(-1)))
            if pa.types.is_large_list(col.type):
                col = col.cast(schema.field("embedding").type)
            w.write_table(pa.table({"id": pa.array(np.arange(at, at + m, dtype=np.int32)), "embedding": col}, schema=schema), row_group_size=m)
            at += m
            g += 1
        w.close()
        write_s = time.perf_counter() - t0
        torch.cuda.empty_cache()
    if world > 1:
        dist.broadcast_object_list(path_box, src=0)
    path = path_box[0]
    try:
        meta = pq.ParquetFile(path).metadata
        cuts = [shard_row_groups(r, world, meta) for r in range(world)]
        lo, hi, base, n_shard = cuts[rank]
        stats = {}
        t0 = time.perf_counter()
        corpus, base2, n2, _ = load_parquet_shard(path, "embedding", rank, world, device=local_rank, stats=stats)
        load_s = time.perf_counter() - t0
        assert (base2, n2) == (base, n_shard)
        from pq_vector_amd.sharding import check_shard_dims
        check_shard_dims(corpus.dim if corpus is not None else 0)       # parquet.rs:231-280 across the shards of one file
        if corpus is None:
            raise SystemExit(f"rank {rank}: empty row-group range (the file has {meta.num_row_groups} row groups for {world} ranks)")
        t0 = time.perf_counter()
        index = pqv.IndexBuilder(corpus).max_iters(20).seed(42).workers(os.cpu_count() or 1).n_clusters(kc).build()
        build_s = time.perf_counter() - t0
        searcher = pqv.Searcher(index, corpus, pqv.PQV_LAYOUT_IVF_ORDERED)
        queries_t = synth(torch, dev, 7, nq, dim)
        rows_t = torch.empty((nq, k), dtype=torch.int32, device=dev)
        dist_t = torch.empty((nq, k), dtype=torch.float32, device=dev)
        bases = [c[2] for c in cuts]
        fast = args.backend == "nccl"
        xchg = ShardExchange(world, nq, k, dev, always_collective=args.force_dist, row_bases=bases if fast else None)

        torch.cuda.synchronize()
        st = new_streams(torch, dev, 1)[0]         # (explicit: handle 0 would be the searcher's own stream, see new_streams)

        def step(npr=nprobe):
            with torch.cuda.stream(st):
                searcher.topk_device(queries_t.data_ptr(), nq, k, npr, rows_t.data_ptr(), dist_t.data_ptr(), 0, 0, stream=st.cuda_stream)
                if world == 1 and not args.force_dist:
                    return dist_t, rows_t.to(torch.int64) & 0xFFFFFFFF
                if xchg.fast:
                    return xchg.exchange_u32(dist_t, rows_t)
                return xchg.exchange(dist_t, rows_t.to(torch.int64) & 0xFFFFFFFF, base)

        def barrier():
            if world > 1:
                dist.barrier()
            torch.cuda.synchronize()

        for _ in range(max(1, args.warmup)):
            step()
        barrier()
        steps = args.steps if args.steps > 0 else 20
        t0 = time.perf_counter()
        for _ in range(steps):
            step()
        barrier()
        el = torch.tensor([time.perf_counter() - t0], dtype=torch.float64, device=dev)
        loads = torch.tensor([load_s, build_s, float(stats.get("GBps", 0.0))], dtype=torch.float64, device=dev)
        every = [torch.zeros_like(loads) for _ in range(world)]
        if world > 1:
            dist.all_reduce(el, op=dist.ReduceOp.MAX)
            dist.all_gather(every, loads)
        else:
            every = [loads]
        elapsed = float(el.item())
        # ---- parity: every list probed -> the exact top-k of the whole file, whichever way it was cut --------------------------
        parity = {"checker": "single-shard search of the whole file on rank 0, every list probed", "ok": True}
        npq = min(nq, max(0, args.parity_queries))
        if npq and n_total * dim * 4 <= 48e9:
            md, mr = step(kc)
            torch.cuda.synchronize()
            if os.environ.get("PQV_DBG"):
                import pq_vector_amd.parquet_io as _pio
                hostrows = np.concatenate(list(_pio._column_chunks(path, "embedding", row_groups=list(range(lo, hi)))))
                sh = torch.from_numpy(hostrows).to(dev)
                bd, br = torch.topk(torch.cdist(queries_t[:npq], sh), k, dim=1, largest=False)
                lr = (rows_t[:npq].to(torch.int64) & 0xFFFFFFFF)
                log(f"[dbg rank {rank}] local device rows == brute on shard: {bool((lr == br).all())}; base {base}; merged rows[0] {mr[0].tolist()} local[0] {lr[0].tolist()} dist[0] {dist_t[0].tolist()[:3]} md[0] {md[0].tolist()[:3]}")
            md, mr = md[:npq].cpu().numpy().copy(), mr[:npq].cpu().numpy().astype(np.int64)
            barrier()
            if rank == 0:
                from pq_vector_amd import parquet_io
                whole = parquet_io.load_embedding_column(path, "embedding", local_rank)
                widx = pqv.IndexBuilder(whole).max_iters(20).seed(42).workers(os.cpu_count() or 1).n_clusters(kc).build()
                ws = pqv.Searcher(widx, whole, pqv.PQV_LAYOUT_IVF_ORDERED)
                wr, wd, wnf, _ = ws.topk(queries_t[:npq].cpu().numpy(), k, kc)
                if os.environ.get("PQV_DBG"):
                    log(f"[dbg whole] rows {whole.rows} wr[0] {wr[0].tolist()} wd[0] {wd[0].tolist()[:3]} nf {wnf[:4].tolist()} mr.shape {mr.shape} wr.shape {wr.shape} nrows_equal {(wr.astype(np.int64) == mr).mean()}")
                    bad = np.nonzero((wr.astype(np.int64) != mr).any(axis=1))[0]
                    log(f"[dbg whole] bad queries {bad.tolist()}")
                    for q in bad[:2]:
                        log(f"[dbg whole] q{q} merged {mr[q].tolist()} {md[q].tolist()}\n              whole {wr[q].tolist()} {wd[q].tolist()}")
                same_rows = bool((wr.astype(np.int64) == mr).all())
                same_dist = bool((wd.view(np.uint32) == md.view(np.uint32)).all())
                parity.update({"queries_checked": int(npq), "global_row_ids_identical": same_rows, "dist_bit_identical": same_dist,
                               "ok": same_rows and same_dist})
                ws.close(); whole.close()
            barrier()
        else:
            parity.update({"skipped": "no parity queries asked for" if not npq else "the whole file does not fit beside a shard on one GPU"})
        if rank == 0:
            line = {"metric": "topk_queries_per_s_k10" if k == 10 else f"topk_queries_per_s_k{k}", "value": nq * steps / elapsed, "unit": "queries/s",
                    "n_gpus": world, "steps": steps, "warmup": args.warmup, "ms_per_step": elapsed / steps * 1e3, "higher_is_better": True,
                    "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
                    "config": {"workload": f"c4 from ONE Parquet file: {n_total}x{dim} uniform f32 as List<f32> in {meta.num_row_groups} row groups "
                                           f"(uncompressed), one row-group range per rank, n_clusters {kc} per shard, k {k}, nprobe {nprobe}, {nq} queries/step",
                               "shards": world, "rows_per_gpu": rpr, "row_groups": int(meta.num_row_groups),
                               "row_group_ranges": [[c[0], c[1]] for c in cuts], "row_bases": bases, "shard_rows": [c[3] for c in cuts]},
                    "file": {"bytes": os.path.getsize(path), "synthetic_file_write_s": write_s},
                    "per_rank": {"load_s": [float(x[0]) for x in every], "build_s": [float(x[1]) for x in every],
                                 "loader_GBps": [float(x[2]) for x in every], "loader_path": stats.get("path")},
                    "exchange": {"ranks": world, "backend": args.backend, "form": "packed u32 all-gather + device merge" if xchg.fast else "generic (i64 rows)"},
                    "parity": parity, "seconds": time.perf_counter() - t_all}
            emit(real_stdout, line)
        barrier()
        searcher.close(); corpus.close()
        ok = torch.tensor([1 if parity.get("ok", True) else 0], dtype=torch.int32, device=dev)
        if world > 1:
            dist.broadcast(ok, src=0)
        return 0 if int(ok.item()) == 1 else 3
    finally:
        if world > 1:
            dist.barrier()
        if rank == 0 and tmp:
            shutil.rmtree(tmp, ignore_errors=True)


def build_record(n, dim, kc, build_s):
    """Phases of the last index build (pqv_index_build_stats) and the roofline of its dominant step, the final
    assignment of every row (src/ivf/index.rs:189-206): a dense n x n_clusters x dim contraction, SURVEY 8(d):
    2 n k_c dim flops in GEMM form against the f32 MFMA peak (the screen runs on v_mfma_f32_16x16x4_f32)."""
    import ctypes as C
    from pq_vector_amd import _ffi
    st = (C.c_double * 10)()
    _ffi.lib().pqv_index_build_stats(st, 10)
    kpp, lloyd, iters, fa, host, fa_screen, lloyd_screen, sample, aw_s, aw_n = list(st)
    flops = 2.0 * n * kc * dim
    tf_phase = flops / fa / 1e12 if fa > 0 else 0.0
    gemm = fa_screen >= 2
    # the contraction kernel's own time (HIP events around every assign_wide_kernel launch of the final assignment) where the
    # f16 form ran; else the phase's wall time
    tf = flops / aw_s / 1e12 if (gemm and aw_s > 0) else tf_phase
    peak = 2500.0 if gemm else 157.3
    return {"seconds": build_s, "vectors_per_s": n / build_s,
            "phases_s": {"kmeans_pp": kpp, "lloyd": lloyd, "lloyd_iterations": int(iters), "final_assignment": fa,
                         "host_list_build": host, "sample_rows": int(sample),
                         "kmeans_pp_us_per_round": kpp / max(1, kc - 1) * 1e6,
                         "note": "k-means++: the rounds run back to back on the device (csrc/kernels_kpp.hip: the pick's sequential f32 sums "
                                 "evaluated exactly in parallel) where the int8 round screen applies, else a host round trip per centroid; "
                                 "the inverted lists are sorted on the device and their host copy is made by the first call that reads it "
                                 "(host_list_build is the host counting sort: 0 unless PQV_DEVICE_LISTS=0)"},
            "roofline": {"bound": "mfma",
                         "kernel": ("final assignment: center_normalize_f16_kernel + assign_wide_kernel (f16 contraction against all centroids, "
                                    "256 x 256 tiles) + assign_resolve_kernel (exact order where two or more candidates remain)") if gemm else
                                   "final assignment: wide_seed_kernel + seed_select_kernel + wide_filter_kernel<f32 operands> + merge_kernel"
                                   if fa_screen else "final assignment: assign_kernel (exact-order VALU)",
                         "achieved": tf, "peak": peak, "unit": "TFLOP/s", "frac": tf / peak,
                         "kernel_s": aw_s if gemm else None, "kernel_launches": int(aw_n) if gemm else None,
                         "whole_phase": {"achieved": tf_phase, "frac": tf_phase / peak, "seconds": fa},
                         "algo_flops": flops,
                         "note": "2 n k_c dim flops / the summed HIP-event time of the assign_wide_kernel launches of the final assignment "
                                 "(`kernel_s`); `whole_phase` divides by the phase's wall time instead (images, contraction, exact "
                                 "re-scoring of the candidates, download of the assignment); peak = dense f16 MFMA for the f16 contraction, "
                                 "f32 MFMA for the f32 screen"}}



# ---- the ONE stdout line ----------------------------------------------------------------------------------------------------
# The driver parses the LAST stdout line; round 5's line was 19.8 KB (notes, dispatch texts, whole per-config records) and did
# not parse.  The line is now a summary with a hard size cap; the full record -- every note, label, dispatch text and
# sub-record -- goes to stderr and to FULL_RECORD beside this file.
LINE_TARGET = 4096
LINE_CAP = 8192
FULL_RECORD = os.path.join(ROOT, "bench_full.json")


def _num(x, sig=6):
    """Finite numbers rounded to `sig` significant digits (the line is a summary; the full record keeps every digit);
    NaN / inf become null: the line must be strict JSON."""
    if isinstance(x, bool) or x is None:
        return x
    if isinstance(x, (int, np.integer)):
        return int(x)
    if isinstance(x, (float, np.floating)):
        x = float(x)
        if not math.isfinite(x):
            return None
        return float(f"{x:.{sig}g}")
    return x


def _clean(o, maxstr=160):
    if isinstance(o, dict):
        return {str(k): _clean(v, maxstr) for k, v in o.items() if v is not None or k in ("vs_baseline", "traffic")}
    if isinstance(o, (list, tuple)):
        return [_clean(v, maxstr) for v in o]
    if isinstance(o, str):
        return o if len(o) <= maxstr else o[:maxstr - 3] + "..."
    if isinstance(o, np.ndarray):
        return _clean(o.tolist(), maxstr)
    return _num(o)


def _parity_ok(rec):
    par = rec.get("parity") if isinstance(rec, dict) else None
    if not isinstance(par, dict):
        return None
    if "ok" in par:
        return bool(par["ok"])
    return bool(par.get("dist_bit_identical") and par.get("row_idx_identical_up_to_order_inside_equal_distance_groups"))


def _short_kernel(name):
    """'wide_filter_kernel<6, 4, 1, true, ...> + ...' -> 'wide_filter_kernel<6,4> + ...' (the full names are in the full record)."""
    out = []
    for part in str(name).split(" + "):
        m = re.match(r"\s*([A-Za-z_0-9:]+)\s*<\s*([^,>]+)\s*,\s*([^,>]+)", part)
        out.append(f"{m.group(1)}<{m.group(2).strip()},{m.group(3).strip()}>" if m else part.strip().split("(")[0].strip()[:60])
    return " + ".join(out)


def summarize_config(rec):
    """value / ms_per_step / roofline fraction / parity of one secondary configuration (summaries only on the line)."""
    if not isinstance(rec, dict):
        return None
    if "error" in rec and "value" not in rec and "build_inplace" not in rec:
        return {"error": str(rec["error"])[:120]}
    out = {}
    for key in ("value", "unit", "ms_per_step", "ms_per_step_serial", "recall_at_k", "exchange_check", "build_inplace_s", "n_clusters",
                "first_search_s", "topk_builder_warm_p50_us"):
        if key in rec:
            out[key] = rec[key]
    rl = rec.get("roofline")
    if isinstance(rl, dict):
        frac = rl.get("frac", rl.get("min_bytes_frac"))
        out["roofline"] = {"bound": rl.get("bound"), "frac": frac, "kernel_ms": rl.get("kernel_ms"), "min_bytes": rl.get("min_bytes")}
    if "screen_survivors_per_query" in rec:
        out["survivors_per_query"] = rec["screen_survivors_per_query"]
    bi = rec.get("build_inplace")
    if isinstance(bi, dict):        # the from-Parquet record
        out.update({"build_inplace_s": bi.get("seconds"), "vectors_per_s": bi.get("vectors_per_s"), "load_s": bi.get("load_s"),
                    "build_s": bi.get("build_s"), "loader_GBps": (bi.get("loader") or {}).get("GBps")})
        tb = rec.get("topk_builder") or {}
        out.update({"first_search_s": tb.get("first_search_s"), "topk_builder_warm_p50_us": tb.get("warm_p50_us")})
        co = rec.get("cpu_oracle") or {}
        if co:
            out["cpu_oracle_build_s"] = co.get("build_s_in_memory")
    ok = _parity_ok(rec)
    if ok is not None:
        out["parity_ok"] = ok
    return out


def compact_line(result):
    """The summary the driver parses: contract keys, `roofline`, `cpu_baseline`, and per-configuration summaries."""
    keep = ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline",
            "dtype", "data", "repeats", "timed_region_s", "ms_per_step_serial", "time_to_first_query_s", "candidates_per_query",
            "exchange_check")
    line = {k: result[k] for k in keep if k in result}
    line.setdefault("vs_baseline", None)
    cfg = result.get("config") or {}
    if isinstance(cfg, str):
        cfg = {"workload": cfg}
    line["config"] = {k: cfg[k] for k in ("workload", "rows", "rows_per_gpu", "dim", "n_clusters", "k", "nprobe", "queries_per_step",
                                          "shards", "parallelism", "row_groups", "row_group_ranges", "row_bases", "shard_rows") if k in cfg}
    if "pipelining" in result:
        line["streams"] = result["pipelining"].get("streams")
    rl = result.get("roofline")
    if isinstance(rl, dict):
        r = {k: rl.get(k) for k in ("bound", "achieved", "peak", "unit", "frac", "traffic", "min_bytes", "traffic_over_min", "kernel_ms",
                                    "fabric_frac") if k in rl}
        r.setdefault("traffic", None)
        if "kernel" in rl:
            r["kernel"] = _short_kernel(rl["kernel"])
        pv = rl.get("pipelined_step_view")
        if isinstance(pv, dict):
            r["pipelined_frac"] = pv.get("min_bytes_frac_of_peak")
        mv = rl.get("mfma_view")
        if isinstance(mv, dict):
            r["mfma_frac"] = mv.get("frac")
        line["roofline"] = r
    cb = result.get("cpu_baseline")
    if isinstance(cb, dict):
        c = {k: cb.get(k) for k in ("value", "unit", "cores", "kind", "sample", "host_cpus") if k in cb}
        ok = _parity_ok(cb)
        if ok is not None:
            c["parity_ok"] = ok
            c["parity_queries"] = (cb.get("parity") or {}).get("queries_checked")
        g = cb.get("generous")
        if isinstance(g, dict) and "value" in g:
            c["generous"] = {"value": g["value"], "cores": g.get("cores")}
        line["cpu_baseline"] = c
    ib = result.get("index_build")
    if isinstance(ib, dict):
        ph = ib.get("phases_s") or {}
        irl = ib.get("roofline") or {}
        line["index_build"] = {"seconds": ib.get("seconds"), "vectors_per_s": ib.get("vectors_per_s"), "kmeans_pp_s": ph.get("kmeans_pp"),
                               "lloyd_s": ph.get("lloyd"), "final_assignment_s": ph.get("final_assignment"),
                               "roofline": {"bound": irl.get("bound"), "frac": irl.get("frac"), "achieved": irl.get("achieved"),
                                            "peak": irl.get("peak"), "unit": irl.get("unit"), "kernel_s": irl.get("kernel_s")}}
    sq = result.get("single_query")
    if isinstance(sq, dict):
        line["single_query"] = {"p50_us": sq.get("p50_us"), "p99_us": sq.get("p99_us"), "host_api_p50_us": sq.get("host_api_p50_us"),
                                "frac": (sq.get("roofline") or {}).get("frac"), "kernels": sq.get("kernels")}
    ra = result.get("recall_at_k")
    if isinstance(ra, dict):
        line["recall_at_k"] = ra.get("recall")
    ctr = result.get("counters")
    if isinstance(ctr, dict) and ctr.get("queries"):
        line["survivors_per_query"] = ctr.get("screen_survivors", 0) / ctr["queries"]
    if "secondary_mixture" in result:
        line["secondary_mixture"] = summarize_config(result["secondary_mixture"])
    if isinstance(result.get("configs"), dict):
        line["configs"] = {name: summarize_config(r) for name, r in result["configs"].items()}
    ex = result.get("exchange")
    if isinstance(ex, dict):
        line["exchange"] = {k: ex.get(k) for k in ("ranks", "backend", "ms_per_step", "share_of_step", "bytes_per_rank_per_step", "form") if k in ex}
    eca = result.get("exchange_c_abi")
    if isinstance(eca, dict):
        line["exchange_c_abi"] = {k: eca.get(k) for k in ("check", "ms_per_step", "error") if k in eca}
    pr = result.get("per_rank_ms_per_step")
    if isinstance(pr, dict):
        line["per_rank_ms_per_step"] = {"min": pr.get("min"), "max": pr.get("max"), "ranks": pr.get("ranks")}
    rp = result.get("replicas")
    if isinstance(rp, dict):
        line["replicas"] = {k: rp.get(k) for k in ("value", "unit", "ms_per_step", "error") if k in rp}
    for k in ("parity", "per_rank", "file"):       # the from-Parquet / c5 lines
        if isinstance(result.get(k), dict):
            line[k] = {kk: vv for kk, vv in result[k].items() if not isinstance(vv, (dict, str)) or kk in ("skipped", "loader_path")}
    for k in ("build_inplace", "topk_builder"):
        if isinstance(result.get(k), dict):
            line.update({kk: vv for kk, vv in (summarize_config(result) or {}).items() if kk not in line})
            break
    line["full_record"] = os.path.basename(FULL_RECORD)
    line = _clean(line)
    # size guard: shed the optional sections, largest first, until the line fits
    for drop in ("per_rank", "file", "replicas", "exchange_c_abi", "per_rank_ms_per_step", "configs", "secondary_mixture", "single_query",
                 "index_build"):
        if len(json.dumps(line, allow_nan=False, separators=(",", ":"))) <= LINE_CAP - 64:
            break
        if drop in line:
            line[drop] = {"dropped": "line size cap; see full_record"}
    return line


def line_text(result):
    """The stdout line as text: compact_line(), and -- should that ever fail or still exceed the cap -- the contract keys with
    `roofline` and `cpu_baseline` only.  A run must never end without a parseable line."""
    try:
        text = json.dumps(compact_line(result), allow_nan=False, separators=(",", ":"))
        if len(text) < LINE_CAP:
            json.loads(text)
            return text
        why = f"compact line of {len(text)} bytes"
    except Exception as e:
        why = f"compact_line failed: {e!r}"
    log(f"[bench] {why}: falling back to the minimal line")
    keep = ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype", "data")
    line = {k: result.get(k) for k in keep}
    cfg = result.get("config")
    line["config"] = {"workload": str(cfg.get("workload") if isinstance(cfg, dict) else cfg)[:160]}
    for sec, keys in (("roofline", ("bound", "achieved", "peak", "unit", "frac", "traffic", "kernel_ms")), ("cpu_baseline", ("value", "unit", "cores", "kind"))):
        if isinstance(result.get(sec), dict):
            line[sec] = {k: result[sec].get(k) for k in keys}
    line["full_record"] = os.path.basename(FULL_RECORD)
    return json.dumps(_clean(line), allow_nan=False, separators=(",", ":"))


def emit(real_stdout, result, fd_is_file=False):
    """Full record -> stderr + FULL_RECORD; compact strict-JSON summary -> the ONE stdout line."""
    try:
        full = json.dumps(result, default=lambda o: _clean(o, 1 << 20) if isinstance(o, (np.ndarray, np.generic)) else str(o))
        log("[bench] full record: " + full)
        with open(FULL_RECORD, "w") as f:
            f.write(full + "\n")
    except Exception as e:             # the side file must not cost the line
        log(f"[bench] full record not written: {e}")
    text = line_text(result)
    sys.stdout.flush()
    os.write(real_stdout, (text + "\n").encode())
    return text


def self_launch(n):
    """`python bench.py --gpus N` with no launcher environment: re-run this command line under
    torch.distributed.run, one rank per GPU, rendezvous on 127.0.0.1 (the container hostname may not resolve)."""
    import socket
    import subprocess
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    env.setdefault("OMP_NUM_THREADS", str(max(1, (os.cpu_count() or 1) // n)))
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n}",
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    log("[bench] self-launch:", " ".join(cmd))
    return subprocess.call(cmd, env=env)


def main():
    # Libraries (RCCL prints a version banner) may write to fd 1; the contract is ONE JSON line
    # on stdout.  Point fd 1 at stderr for the run and keep the real stdout for the result.
    sys.stdout.flush()
    real_stdout = os.dup(1)
    os.dup2(2, 1)

    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=0, help="timed steps (default: calibrated so the timed region is >= 1 s)")
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--workload", default=None, choices=sorted(WORKLOADS),
                    help="default: c3 on one GPU, c4 (one 12.5 M-row shard per rank) on several")
    ap.add_argument("--nq", type=int, default=0, help="queries per step (default per workload)")
    ap.add_argument("--k", type=int, default=10, help="neighbours per query (BASELINE metric: 10)")
    ap.add_argument("--nprobe", type=int, default=0, help="probed lists per query (default per workload; a diagnostic sweep changes the workload's meaning)")
    ap.add_argument("--cpu-seconds", type=float, default=10.0, help="CPU baseline budget per column")
    ap.add_argument("--no-cpu", action="store_true")
    ap.add_argument("--layout", default="ivf", choices=["ivf", "row"])
    ap.add_argument("--data", default="uniform", choices=["uniform", "mixture"],
                    help="uniform: the reference's bench recipe (default); mixture: Gaussian mixture with n_clusters "
                         "components, sigma 0.1 (SURVEY 8d's secondary set: IVF recall is meaningful there)")
    ap.add_argument("--force-dist", action="store_true",
                    help="initialise RCCL and run the shard exchange even with one rank (path check)")
    ap.add_argument("--backend", default="nccl", choices=["nccl", "gloo"],
                    help="gloo lets several ranks share one GPU (path check only; the real run uses RCCL)")
    ap.add_argument("--streams", type=int, default=N_LANES,
                    help="HIP streams the steps alternate between (1 = strictly serial steps)")
    ap.add_argument("--no-timing", action="store_true",
                    help="do not record HIP events around the kernels (roofline.kernel_ms is then 0)")
    ap.add_argument("--multi", default="auto", choices=["auto", "replica", "shard"],
                    help="N > 1: 'shard' = every rank holds one shard, searches the same batch, top-k lists are exchanged over "
                         "RCCL and merged (default; c4 = 12.5 M rows per rank); 'replica' = every rank holds the whole corpus "
                         "and searches its OWN query batch (throughput mode, no data-path collective)")
    ap.add_argument("--single", type=int, default=100,
                    help="also time this many single-query calls (latency mode, p50 / p99); 0 disables")
    ap.add_argument("--cabi-check", action="store_true",
                    help="multi-rank runs: also run the exchange through the C ABI (pqv_shard_*: RCCL bound by the library) and compare; "
                         "always on for --force-dist with one rank, opt-in beyond (a second communicator that has never met real "
                         "multi-GPU hardware must not be able to cost a scaling run its line)")
    ap.add_argument("--no-secondary", action="store_true", help="skip the Gaussian-mixture pass that follows the default C3 run")
    ap.add_argument("--rows-per-rank", type=int, default=0,
                    help="c4 only: rows of every rank's shard (default 12 500 000; a smaller shard makes a quick multi-rank path check)")
    ap.add_argument("--from-parquet", action="store_true",
                    help="only the reference's end-to-end benches (IndexBuilder(path).build_inplace(), TopkBuilder(path).search()) on a synthetic "
                         "Parquet file of its own bench shape; prints that record as the line.  With --gpus N > 1 (or --force-dist): ONE file, "
                         "one row-group range per rank (--rows-per-rank, default 1 M), loaded, indexed and searched per shard, lists exchanged")
    ap.add_argument("--no-configs", action="store_true", help="skip the c2 / refbench / c4-shard / c5 passes that follow the default C3 run")
    ap.add_argument("--parity-queries", type=int, default=64, help="queries of the step checked bit for bit against the CPU oracle")
    ap.add_argument("--recall", type=int, default=32, help="queries checked against an exact brute force (0 disables)")
    args = ap.parse_args()
    global K
    K = args.k

    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        # Called as `python bench.py --gpus N` without a launcher: start the N ranks ourselves (one process per
        # GPU, torch.distributed.run on 127.0.0.1 with a free port).  Rank 0 of the children inherits the real
        # stdout and prints the ONE JSON line; this process only forwards the exit status.
        os.dup2(real_stdout, 1)
        sys.exit(self_launch(args.gpus))

    import torch
    import torch.distributed as dist
    import pq_vector_amd as pqv

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit(f"WORLD_SIZE={world} does not match --gpus {args.gpus}")
    if args.backend == "nccl" and world > torch.cuda.device_count():
        raise SystemExit(f"--gpus {world} over RCCL needs {world} devices, {torch.cuda.device_count()} visible "
                         "(--backend gloo lets ranks share a device: path check only)")
    if not torch.cuda.is_available() or pqv.device_count() < 1:
        raise SystemExit("bench.py needs a HIP device: pq_vector_amd has no CPU fallback")
    if args.backend == "gloo":          # test hook: ranks may share a device
        local_rank = local_rank % torch.cuda.device_count()
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    use_dist = world > 1 or args.force_dist
    if use_dist:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29517")
        if args.backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
        else:
            dist.init_process_group("gloo", rank=rank, world_size=world)

    if args.from_parquet and use_dist:
        rc = from_parquet_sharded(args, pqv, torch, dist, dev, local_rank, rank, world, real_stdout)
        if dist.is_initialized():
            dist.destroy_process_group()
        sys.exit(rc)
    if args.from_parquet:
        rec = from_parquet_config(args, pqv, torch, dev, local_rank)
        bi = rec.get("build_inplace", {})
        line = {"metric": "index_build_vectors_per_s_from_parquet", "value": bi.get("vectors_per_s"), "unit": "vectors/s", "n_gpus": 1,
                "steps": 1, "warmup": 0, "ms_per_step": (bi.get("seconds") or 0.0) * 1e3, "higher_is_better": True, "scaling": "weak",
                "vs_baseline": None, "dtype": "f32", "data": "synthetic", "config": {"workload": rec["config"]}}
        line.update(rec)
        emit(real_stdout, line)
        sys.exit(0 if rec.get("parity", {}).get("ok", True) else 3)
    if args.workload is None:
        args.workload = "c3" if world == 1 else "c4"
    n_total, dim, n_clusters, nprobe, nq_default = WORKLOADS[args.workload]
    if args.nprobe > 0:
        nprobe = args.nprobe
    if args.workload == "c4" and args.rows_per_rank > 0:
        n_total = args.rows_per_rank
    nq = args.nq or nq_default
    from pq_vector_amd.sharding import ShardExchange, shard_range
    weak = args.workload == "c4"           # per-rank shard size fixed: N ranks hold N x rows
    replica = world > 1 and args.multi == "replica" and not weak and not args.force_dist
    if weak:
        lo, hi = rank * n_total, (rank + 1) * n_total
        n_total = n_total * world
    elif replica:
        lo, hi = 0, n_total                # the whole corpus on every rank; the QUERIES are what is sharded
    else:
        lo, hi = shard_range(rank, world, n_total)
    n_shard = hi - lo

    # ---- synthetic data: corpus seed 1234 (+ rank for a shard), query seed 7 ----------------------------
    if args.data == "mixture":
        corpus_t = synth_mixture(torch, dev, 1234 if replica else 1234 + rank, n_shard, dim, n_clusters or 1024)
        queries_t = synth_mixture(torch, dev, 7 + rank if replica else 7, nq, dim, n_clusters or 1024)
    else:
        corpus_t = synth(torch, dev, 1234 if replica else 1234 + rank, n_shard, dim)
        queries_t = synth(torch, dev, 7 + rank if replica else 7, nq, dim)          # replicas search different batches
    torch.cuda.synchronize()

    t0 = time.perf_counter()
    corpus = pqv.Corpus.from_device_ptr(corpus_t.data_ptr(), n_shard, dim, device=local_rank,
                                        keepalive=corpus_t)
    attach_s = time.perf_counter() - t0     # the library's first call for the device: code objects, the stream's queue, the runtime's copy staging
    if args.workload in ("c5", "c5s"):
        return bench_brute(args, pqv, torch, corpus, corpus_t, queries_t, n_shard, dim, nq, rank, world,
                           real_stdout)

    # ---- index build on the GPU (max_iters 20, seed 42: src/ivf/parquet.rs:37-38) --------
    workers = os.cpu_count() or 1
    t0 = time.perf_counter()
    b = pqv.IndexBuilder(corpus).max_iters(20).seed(42).workers(workers)
    index = b.n_clusters(n_clusters).build() if n_clusters else b.build()
    build_s = time.perf_counter() - t0
    build_info = build_record(n_shard, dim, int(index.n_clusters), build_s)
    flags = pqv.PQV_LAYOUT_ROW_ORDER if args.layout == "row" else pqv.PQV_LAYOUT_IVF_ORDERED
    t0 = time.perf_counter()
    searcher = pqv.Searcher(index, corpus, flags)
    layout_s = time.perf_counter() - t0
    plan_text = searcher.describe(nq, K, nprobe)
    if rank == 0:
        log(f"[bench] shard rows={n_shard} dim={dim} n_clusters={index.n_clusters} build={build_s:.3f}s "
            f"searcher={layout_s:.3f}s\n[bench] {plan_text}")

    # ---- device outputs: one set per stream lane ------------------------------------------------
    # Steps alternate between `--streams` HIP streams (default 2): the library keeps one scratch lane
    # per stream, so step i + 1's probe / bucketing / seed kernels and the head of its screen kernel run
    # in the tail of step i's screen kernel.  Each step is still one complete pass over one batch.
    n_lanes = max(1, args.streams)
    # (explicit streams only: new_streams)
    torch.cuda.synchronize()
    lane_streams = new_streams(torch, dev, n_lanes)
    rows_l = [torch.empty((nq, K), dtype=torch.int32, device=dev) for _ in range(n_lanes)]
    dist_l = [torch.empty((nq, K), dtype=torch.float32, device=dev) for _ in range(n_lanes)]
    nf_l = [torch.empty((nq,), dtype=torch.int32, device=dev) for _ in range(n_lanes)]
    nc_l = [torch.empty((nq,), dtype=torch.int64, device=dev) for _ in range(n_lanes)]
    if weak:
        bases = [r * (n_total // world) for r in range(world)]
    else:
        bases = [shard_range(r, world, n_total)[0] for r in range(world)]
    xchg_l = [ShardExchange(world, nq, K, dev, always_collective=args.force_dist,
                            row_bases=bases if args.backend == "nccl" else None) for _ in range(n_lanes)]
    xchg = xchg_l[0]
    step_no = [0]
    exchange = use_dist and not replica

    def step(lane=None):
        if lane is None:
            lane = step_no[0] % n_lanes
            step_no[0] += 1
        st = lane_streams[lane]
        with torch.cuda.stream(st):
            # hot path on this rank's shard; asynchronous on the lane's stream
            searcher.topk_device(queries_t.data_ptr(), nq, K, nprobe, rows_l[lane].data_ptr(), dist_l[lane].data_ptr(),
                                 nf_l[lane].data_ptr(), nc_l[lane].data_ptr(), stream=st.cuda_stream)
            if not exchange:
                return dist_l[lane], rows_l[lane]
            # exchange: ONE all-gather of k x {dist, row} per query, then the merge keyed (dist, shard,
            # position) -- pq_vector_amd/sharding.py
            if xchg_l[lane].fast:
                return xchg_l[lane].exchange_u32(dist_l[lane], rows_l[lane])
            return xchg_l[lane].exchange(dist_l[lane], rows_l[lane].to(torch.int64) & 0xFFFFFFFF, lo)

    def barrier():
        if use_dist:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(max(1, args.warmup)):
        step()
    barrier()
    steps = args.steps
    if steps <= 0:
        # calibration: enough steps for a timed region of >= 1.2 s (every rank must agree on the count)
        t0 = time.perf_counter()
        for _ in range(8):
            step()
        barrier()
        per = (time.perf_counter() - t0) / 8
        cal = torch.tensor([per], dtype=torch.float64, device=dev)
        if use_dist:
            dist.all_reduce(cal, op=dist.ReduceOp.MAX)
        steps = int(max(20, math.ceil(1.2 / max(float(cal.item()), 1e-6))))
    barrier()
    # The timed block is EXACTLY `steps` steps between barrier + synchronize on both sides.  A short block (the driver's
    # --steps 20 is 40 ms on C3) is repeated until the blocks add up to >= 1 s, and the MEDIAN block is reported
    # (`repeats`; every rank runs the same count: the decision is taken on the all-reduced time of the first block).
    searcher.set_timing(not args.no_timing)
    block_s = []
    rank_block_s = []          # per block: every rank's own wall time (the line reports the spread)
    repeats = 1
    while len(block_s) < repeats:
        t0 = time.perf_counter()
        for _ in range(steps):
            out_d, out_r = step()
        barrier()
        el_b = torch.tensor([time.perf_counter() - t0], dtype=torch.float64, device=dev)
        if use_dist:
            mine = el_b.clone()
            every = [torch.zeros_like(mine) for _ in range(world)]
            dist.all_gather(every, mine)
            rank_block_s.append([float(x.item()) for x in every])
            dist.all_reduce(el_b, op=dist.ReduceOp.MAX)
        block_s.append(float(el_b.item()))
        if len(block_s) == 1 and args.steps > 0:
            repeats = int(min(200, max(1, math.ceil(1.0 / max(block_s[0], 1e-6)))))
    elapsed = float(np.median(block_s))
    searcher.set_timing(False)
    rerank_ms, total_ms, ncalls = searcher.timing_read()
    # A serial pass (outside the timed region): steps issued one at a time on one stream -- the step time
    # without the overlap between consecutive steps, and the isolated per-launch kernel durations.
    serial_ms, serial_rr_ms, serial_hot_ms = None, None, None
    if not args.no_timing:
        ns = min(20, max(3, steps))
        searcher.set_timing(True)
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        for _ in range(ns):
            step(0)
            torch.cuda.synchronize()
        serial_ms = (time.perf_counter() - t1) / ns * 1e3
        searcher.set_timing(False)
        s_rr, s_tot, s_n = searcher.timing_read()
        serial_rr_ms = s_rr / max(1, s_n)
        serial_hot_ms = s_tot / max(1, s_n)


    # ---- bytes of the dominant (re-rank) kernels ------------------------------------------------------
    ncand = nc_l[0].cpu().numpy().astype(np.int64)
    cand_rows = int(ncand.sum())
    ref_algo_bytes = cand_rows * (4 * dim + 4)      # every candidate row + its id, once per query that probes it
    rr_ms = rerank_ms / max(1, ncalls)
    ctr = searcher.counters()
    fp = searcher.footprint()
    # what the LIBRARY holds next to the caller's column (here a torch tensor: row_order_bytes is the caller's, and a library-owned
    # corpus can drop its row-order copy with PQV_RELEASE_ROW_ORDER once the searcher exists)
    fp["library_bytes"] = fp["ivf_rows_bytes"] + fp["blocked_bytes"] + fp["other_bytes"]
    fp["library_over_corpus"] = fp["library_bytes"] / max(1, n_shard * dim * 4)
    screened = "wide_filter_kernel" in plan_text or "tile_filter_kernel" in plan_text
    wide = "wide_filter_kernel" in plan_text
    f16 = "f16 screen operands" in plan_text
    i8 = "int8 screen operands" in plan_text
    # distinct probed rows of one step (every list that at least one query of the batch probes, once)
    off = index.list_offsets.astype(np.int64)
    lens = np.diff(off)
    probed = np.zeros(len(lens), dtype=bool)
    # (the probe order is not returned by the device API: recompute the batch's probed set from n_candidates is
    #  not possible, so ask the library for a sample of the queries and scale -- exact when nq <= 256)
    qs_host = queries_t.cpu().numpy()
    sample_q = range(nq) if nq <= 256 else range(0, nq, max(1, nq // 256))
    for i in sample_q:
        probed[searcher.probe(qs_host[i], nprobe)] = True
    distinct_rows = int(lens[probed].sum())
    survivors_per_step = ctr["screen_survivors"] / max(1, ctr["queries"]) * nq if ctr["queries"] else 0.0
    if wide:
        # what the screened path must move as designed: the operand image of every probed list ONCE
        # (2 or 4 bytes per value) + 8 bytes per row (norm, id) + the f32 row of every survivor
        opb = 1 if i8 else 2 if f16 else 4
        min_bytes = distinct_rows * (opb * dim + 8) + survivors_per_step * 4 * dim
    else:
        min_bytes = ref_algo_bytes if "stream_kernel" in plan_text else distinct_rows * (4 * dim + 4)
    if " | kernels: " in plan_text:
        kernels = [x.strip() for x in plan_text.split(" | kernels: ")[1].split(";")]
    else:
        kernels = (["tile_filter_kernel", "tile_rerank_kernel", "seed_threshold_kernel"] if screened else
                   ["tile_rerank_kernel"] if "tile_rerank_kernel" in plan_text else ["stream_kernel"])
    traffic, traffic_src = (None, None)
    if world == 1 and nq == nq_default and K == 10:
        traffic, traffic_src = (pmc_traffic(args.workload, kernels)
                                if " | kernels: " in plan_text and args.data == "uniform" and nq == nq_default else (None, None))

    k_ms = serial_rr_ms if serial_rr_ms else rr_ms          # isolated launches: what rocprofv3 --kernel-trace reports
    # headline: the bytes the design has to move (min_bytes, recomputed from THIS run's probe sets and survivor counters) over
    # the kernel time measured in THIS run; the committed counter bytes only enter `traffic` / `fabric_frac`
    achieved = min_bytes / (k_ms * 1e-3) / 1e9 if k_ms and k_ms > 0 else 0.0
    fabric = traffic / (k_ms * 1e-3) / 1e9 if traffic and k_ms and k_ms > 0 else None
    mf = 2.0 * dim * cand_rows                         # the Q.X^T contraction of the screen
    result = {
        "metric": f"topk_queries_per_s_k{K}",
        "value": (world if replica else 1) * nq * steps / elapsed,
        "unit": "queries/s",
        "n_gpus": world,
        "steps": steps,
        "warmup": args.warmup,
        "ms_per_step": elapsed / steps * 1e3,
        "higher_is_better": True,
        "scaling": "weak" if (weak or replica or world == 1) else "strong",
        "vs_baseline": None,
        "dtype": "f32",
        "data": "synthetic" if args.data == "uniform" else "synthetic (gaussian mixture, sigma 0.1)",
        "config": {"workload": f"{args.workload}: {n_total}x{dim} {'uniform' if args.data == 'uniform' else 'gaussian-mixture'} f32, n_clusters {index.n_clusters}"
                               f"{' per shard' if world > 1 and not replica else ''}, k {K}, nprobe {nprobe}, "
                               f"{nq} queries/step, layout {args.layout}",
                   "rows": n_total, "rows_per_gpu": n_shard, "dim": dim, "n_clusters": int(index.n_clusters), "k": K,
                   "nprobe": nprobe, "queries_per_step": nq, "shards": 1 if replica else world,
                   "parallelism": ("replicas x%d: corpus replicated, query batches sharded (%d queries per step over "
                                   "the job), no data-path collective" % (world, world * nq)) if replica
                   else ("corpus sharded x%d (%d rows per rank, one IVF index per shard), every rank searches the same batch, one RCCL "
                         "all-gather of the per-shard top-k + device merge per step" % (world, n_shard)) if world > 1
                   else "single GPU"},
        "timed_region_s": float(sum(block_s)),
        "repeats": len(block_s),
        "repeats_note": "blocks of exactly `steps` steps (barrier + synchronize on both sides, max over ranks); ms_per_step and value "
                        "are the MEDIAN block; min / max block ms_per_step: %.4f / %.4f" % (min(block_s) / steps * 1e3, max(block_s) / steps * 1e3),
        "pipelining": {"streams": n_lanes, "GPU_MAX_HW_QUEUES": os.environ.get("GPU_MAX_HW_QUEUES"),
                       "note": "steps alternate between %d HIP streams, so consecutive steps overlap on the GPU and ms_per_step "
                               "(throughput) is below one step's own latency; ms_per_step_serial is the same step issued "
                               "alone and host-synchronised" % n_lanes if n_lanes > 1 else "steps are serial on one stream"},
        "ms_per_step_serial": serial_ms,
        "hot_path_ms_per_step_serial": serial_hot_ms,
        "index_build_vectors_per_s": n_shard / build_s,
        "index_build_s": build_s,
        "library_init_and_corpus_attach_s": attach_s,
        "index_build": build_info,
        "searcher_create_s": layout_s,
        "time_to_first_query_s": build_s + layout_s,
        "candidates_per_query": cand_rows / nq,
        "corpus_row_scans_per_s": n_total * nq * steps / elapsed if not replica else world * n_total * nq * steps / elapsed,
        "hbm_footprint": fp,
        "dispatch": plan_text,
    }
    result["roofline"] = {
        "bound": "hbm", "kernel": " + ".join(k.split("@")[0] for k in kernels),
        "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS,
        "traffic": traffic, "traffic_source": traffic_src,
        "traffic_label": "fabric bytes per step from the committed rocprofv3 PMC summaries of this workload (FETCH_SIZE x 2 + WRITE_SIZE): "
                         "L2 misses served by HBM OR the 256 MB Infinity Cache (FETCH_SIZE counts MALL hits too); fabric_frac = traffic / "
                         "kernel_ms / peak is a memory-system utilisation, an upper bound of the HBM-only figure",
        "fabric_frac": (fabric / HBM_PEAK_GBS) if fabric else None,
        "min_bytes": min_bytes,
        "min_bytes_frac": (min_bytes / (k_ms * 1e-3) / 1e9 / HBM_PEAK_GBS) if k_ms and k_ms > 0 else None,
        "traffic_over_min": (traffic / min_bytes) if traffic and min_bytes else None,
        "kernel_ms": k_ms, "kernel_ms_in_timed_region": rr_ms,
        # the WHOLE step of the pipelined (timed) run -- not a kernel's roofline: consecutive steps overlap on the stream lanes, so a
        # step's bytes pass in ms_per_step although its own kernels, issued alone, take kernel_ms
        "pipelined_step_view": {"min_bytes_GBps": min_bytes / (elapsed / steps) / 1e9,
                                "min_bytes_frac_of_peak": min_bytes / (elapsed / steps) / 1e9 / HBM_PEAK_GBS,
                                "traffic_GBps": (traffic / (elapsed / steps) / 1e9) if traffic else None,
                                "traffic_frac_of_copy_ceiling": (traffic / (elapsed / steps) / 1e9 / 6290.0) if traffic else None,
                                "note": "bytes of ONE step over ms_per_step of the timed, pipelined run (probe, bucketing, seed, screen and merge of "
                                        "neighbouring steps overlap); copy ceiling = 6.29 TB/s (MI355X_MICROARCH.md)"},
        "achieved_basis": "min_bytes of this run / kernel_ms of this run (frac == min_bytes_frac)",
        "min_bytes_definition": ("screened path: (operand bytes per value x dim + 8) per DISTINCT probed row of the step + 4 dim per "
                                 "survivor of the screen" if wide else "SURVEY 8d bytes"),
        "note": "kernel_ms = HIP events around the re-rank kernels (threshold sample + select + screen/exact evaluation) with steps "
                "issued one at a time; kernel_ms_in_timed_region = the same events inside the timed region, where launches of "
                "consecutive steps share the GPU",
        "reference_algorithm_view": {
            "algo_bytes_per_launch": ref_algo_bytes,
            "equivalent_GBps": ref_algo_bytes / (k_ms * 1e-3) / 1e9 if k_ms else 0.0,
            "note": "SURVEY 8d bytes of the reference loop (every candidate row + id once per query that probes it) / kernel_ms: "
                    "what the per-query streaming kernel would have to sustain for this step time -- the batched screened kernel "
                    "serves up to 128 queries from one pass over a list and reads 1- or 2-byte operand images, so this is a speed-up "
                    "figure, not a utilisation"},
        "mfma_view": {"flops_per_launch": mf, "achieved_tflops": mf / (k_ms * 1e-3) / 1e12 if k_ms else 0.0,
                      "peak_tflops": 3944.0 if i8 else 2500.0 if f16 else 157.3,
                      "frac": (mf / (k_ms * 1e-3) / 1e12 / (3944.0 if i8 else 2500.0 if f16 else 157.3)) if k_ms else 0.0} if screened else None,
    }
    result["counters"] = ctr
    if rank_block_s:
        med = int(np.argsort(block_s)[len(block_s) // 2])
        per = [t / steps * 1e3 for t in rank_block_s[med]]
        result["per_rank_ms_per_step"] = {"min": min(per), "max": max(per), "ranks": per,
                                          "note": "every rank's own wall time of the median block / steps; the line's ms_per_step is the maximum"}
    if exchange and xchg.fast:
        # the library merge kernel against the torch stable-sort merge of the same gathered lists
        lane = (step_no[0] - 1) % n_lanes if step_no[0] else 0
        out_d, out_r = step(0)
        ref_d, ref_r = xchg_l[0].exchange(dist_l[0], rows_l[0].to(torch.int64) & 0xFFFFFFFF, lo)
        ok = torch.tensor([int(torch.equal(ref_d, out_d) and torch.equal(ref_r, out_r))], device=dev)
        dist.all_reduce(ok, op=dist.ReduceOp.MIN)
        result["exchange_check"] = bool(ok.item())
        # cost of the exchange alone (all-gather + merge kernel), host-synchronised
        torch.cuda.synchronize()
        if use_dist:
            dist.barrier()
        t1 = time.perf_counter()
        for _ in range(10):
            xchg_l[0].exchange_u32(dist_l[0], rows_l[0])
        torch.cuda.synchronize()
        xms = (time.perf_counter() - t1) / 10 * 1e3
        result["exchange"] = {"ranks": world, "backend": "RCCL" if args.backend == "nccl" else args.backend,
                              "ms_per_step": xms, "share_of_step": xms / (elapsed / steps * 1e3),
                              "bytes_per_rank_per_step": nq * K * 8,
                              "collective": "one all_gather_into_tensor of packed {f32 distance, u32 row} pairs + shard_merge_kernel"}
        if args.backend == "nccl" and (args.cabi_check or world == 1):
            # the same exchange through the C ABI (pqv_shard_*: RCCL bound by the library, what a Rust host calls).  A
            # cross-check only: every rank takes the same collectives whatever fails, and nothing here can cost the line.
            from pq_vector_amd.sharding import RcclShardComm
            comm, why = RcclShardComm.create_collective(rank, world, local_rank)
            if comm is None:
                result["exchange_c_abi"] = {"error": why[:300]}
            else:
                cd = torch.empty((nq, K), dtype=torch.float32, device=dev)
                cr = torch.empty((nq, K), dtype=torch.int64, device=dev)
                bases_t = torch.tensor(bases, dtype=torch.int64, device=dev)
                good, ms_c = 1, None
                try:
                    comm.exchange(dist_l[0], rows_l[0], bases_t, cd, cr)
                    torch.cuda.synchronize()
                    good = int(torch.equal(cd, ref_d) and torch.equal(cr, ref_r))
                    t2 = time.perf_counter()
                    for _ in range(10):
                        comm.exchange(dist_l[0], rows_l[0], bases_t, cd, cr)
                    torch.cuda.synchronize()
                    ms_c = (time.perf_counter() - t2) / 10 * 1e3
                except Exception as e:
                    good = 0
                    result["exchange_c_abi_error"] = str(e)[:300]
                okc = torch.tensor([good], device=dev)
                dist.all_reduce(okc, op=dist.ReduceOp.MIN)
                result["exchange_c_abi"] = {"check": bool(okc.item()), "ms_per_step": ms_c,
                                            "entry_points": "pqv_shard_unique_id / pqv_shard_comm_create / pqv_shard_exchange"}
                comm.close()

    elif exchange:
        # generic exchange (gloo path check: ranks may share a device): host-staged all-gather + the torch stable-sort merge
        step(0)
        torch.cuda.synchronize()
        dist.barrier()
        t1 = time.perf_counter()
        for _ in range(10):
            xchg_l[0].exchange(dist_l[0], rows_l[0].to(torch.int64) & 0xFFFFFFFF, lo)
        torch.cuda.synchronize()
        xms = (time.perf_counter() - t1) / 10 * 1e3
        result["exchange"] = {"ranks": world, "backend": args.backend, "ms_per_step": xms, "share_of_step": xms / (elapsed / steps * 1e3),
                              "bytes_per_rank_per_step": nq * K * 16,
                              "collective": "one all_gather_into_tensor of packed {f32 distance bits, i64 global row} pairs + stable-sort merge"}

    # ---- N > 1: the other multi-GPU mode on the same hardware, as a secondary object -------------------------
    if world > 1 and not replica and not args.force_dist and args.multi == "auto":
        try:        # a secondary object: whatever goes wrong in it (on every rank alike: an allocation, a build) must not cost the headline its line
            result["replicas"] = replica_pass(args, pqv, torch, dist, dev, local_rank, rank, world, nq, steps=max(20, min(steps, 200)))
        except Exception as e:
            result["replicas"] = {"error": str(e)[:300]}

    # ---- latency mode: one query per call through the same device API ------------
    if args.single and rank == 0 and world == 1:
        rows1 = torch.empty((1, K), dtype=torch.int32, device=dev)
        dist1 = torch.empty((1, K), dtype=torch.float32, device=dev)
        nf1 = torch.empty((1,), dtype=torch.int32, device=dev)
        nc1 = torch.empty((1,), dtype=torch.int64, device=dev)
        st0 = lane_streams[0].cuda_stream
        lat = []
        cand1 = []           # candidate rows of every timed one-query call (its min_bytes)
        # diagnostic build only (make -C pq-vector_amd/csrc stamps; PQV_LIB_PATH): device wall-clock stamps inside the kernels
        import ctypes
        from pq_vector_amd import _ffi
        stamps_fn = getattr(_ffi.lib(), "pqv_debug_stamps", None) if os.environ.get("PQV_LIB_PATH") else None
        stamp_rows = []
        for i in range(args.single + 5):
            q1 = queries_t[i % nq:i % nq + 1]
            torch.cuda.synchronize()
            if stamps_fn is not None:
                stamps_fn(None, 1)
            t1 = time.perf_counter()
            searcher.topk_device(q1.data_ptr(), 1, K, nprobe, rows1.data_ptr(), dist1.data_ptr(),
                                 nf1.data_ptr(), nc1.data_ptr(), stream=st0)
            torch.cuda.synchronize()
            lat.append(time.perf_counter() - t1)
            cand1.append(int(nc1.item()))
            if stamps_fn is not None:
                buf = (ctypes.c_ulonglong * 64)()
                stamps_fn(buf, 0)
                stamp_rows.append(np.array(buf[:], dtype=np.uint64))
        lat = np.array(lat[5:]) * 1e6
        if stamp_rows:
            st = np.stack(stamp_rows[5:]).astype(np.int64)
            rel = (st - st[:, :1]) * 0.01            # 100 MHz ticks -> microseconds after the probe kernel's first block started
            med = np.median(rel, axis=0)
            sys.stderr.write("stamps (us after probe start, median): " + " ".join(
                f"[{j}]={med[j]:.1f}" for j in range(64) if (st[:, j] != 0).all() and (st[:, j] != -1).all()) + "\n")
        # the same through the host entry point (pqv_topk: query from host memory, results into host arrays -- what
        # TopkBuilder::search does per call)
        hlat = []
        for i in range(args.single + 5):
            qh = qs_host[i % nq:i % nq + 1]
            t1 = time.perf_counter()
            searcher.topk(qh, K, nprobe)
            hlat.append(time.perf_counter() - t1)
        hlat = np.array(hlat[5:]) * 1e6
        result["single_query"] = {"calls": int(lat.size), "p50_us": float(np.percentile(lat, 50)),
                                  "p99_us": float(np.percentile(lat, 99)), "mean_us": float(lat.mean()),
                                  "qps": float(1e6 / lat.mean()), "dispatch": searcher.describe(1, K, nprobe),
                                  "host_api_p50_us": float(np.percentile(hlat, 50)), "host_api_p99_us": float(np.percentile(hlat, 99)),
                                  "note": "one query per call, host-synchronised after each; p50_us: device pointers in and out "
                                          "(pqv_topk_device + a device synchronise), host_api: pqv_topk with host arrays (copies included)"}
        # roofline of the CALL (not of a kernel): the operand image + 8 bytes of every candidate row of the query, once, over the
        # whole call's p50 -- probe, seed, screen, merge and the launch boundaries between them included
        sq_plan = result["single_query"]["dispatch"]
        sq_opb = 1 if "int8 screen operands" in sq_plan else 2 if "f16 screen operands" in sq_plan else 4
        sq_bytes = float(np.mean(cand1[5:])) * (sq_opb * dim + 8) if "wide_filter_kernel" in sq_plan else float(np.mean(cand1[5:])) * (4 * dim + 4)
        sq_p50 = result["single_query"]["p50_us"] * 1e-6
        result["single_query"]["roofline"] = {"bound": "hbm", "min_bytes": sq_bytes, "achieved": sq_bytes / sq_p50 / 1e9, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                                              "frac": sq_bytes / sq_p50 / 1e9 / HBM_PEAK_GBS,
                                              "basis": "candidate rows of the query x (operand image bytes per row + 8) / p50 of the whole call"}
    # ---- recall of the IVF answers against an exact brute force (benches/query.rs prints it too) --------
    if args.recall and rank == 0 and world == 1:
        m = min(args.recall, nq)
        out_d, out_r = step(0)
        torch.cuda.synchronize()
        got = rows_l[0][:m].cpu().numpy().view(np.uint32)
        br, bd, _ = corpus.brute_topk(qs_host[:m], K, pqv.PQV_L2SQ_MFMA)
        hits = sum(len(set(got[i].tolist()) & set(br[i].tolist())) for i in range(m))
        result["recall_at_k"] = {"queries": m, "recall": hits / float(m * K),
                                 "note": "fraction of the exact top-k (pqv_brute_topk, L2) found by the IVF search at this nprobe; "
                                         "on uniform random data IVF recall is low by nature (benches/query.rs prints the same figure)"}
    # ---- CPU baseline (rank 0, N = 1): the oracle, natively compiled -----------
    rc = 0
    if rank == 0 and world == 1 and not args.no_cpu:
        out_d, out_r = step(0)
        torch.cuda.synchronize()
        result["cpu_baseline"] = cpu_baseline(args, index, corpus_t, qs_host, rows_l[0], dist_l[0], nprobe, nq, searcher)
        par = result["cpu_baseline"]["parity"]
        if not (par["dist_bit_identical"] and par["row_idx_identical_up_to_order_inside_equal_distance_groups"]):
            rc = 3
    # ---- the default line also carries the other BASELINE configurations (each with its own oracle check) -------------
    if rank == 0 and world == 1 and args.workload == "c3" and args.data == "uniform" and not use_dist and K == 10 and \
            not (args.no_secondary and args.no_configs):
        # release the headline's corpus and searcher first: every configuration below builds its own
        searcher.close(); corpus.close()
        del corpus_t, rows_l, dist_l, nf_l, nc_l
        torch.cuda.empty_cache()

        def guarded(fn):
            try:
                return fn()
            except Exception as e:           # the headline must not depend on a secondary configuration
                import traceback
                log(traceback.format_exc())
                return {"error": str(e)[:300]}
        if not args.no_secondary:
            # secondary data set where IVF works: the same shape as a Gaussian mixture, oracle-checked like the headline
            result["secondary_mixture"] = guarded(lambda: ivf_config(args, pqv, torch, dev, local_rank, "c3", K, data="mixture",
                                                                     parity_queries=args.parity_queries, recall=32))
        if not args.no_configs:
            cfg = {}
            cfg["c1"] = guarded(lambda: c1_config(args, pqv, torch, dev, local_rank))
            cfg["c2"] = guarded(lambda: ivf_config(args, pqv, torch, dev, local_rank, "c2", 10, parity_queries=64))
            cfg["refbench"] = guarded(lambda: ivf_config(args, pqv, torch, dev, local_rank, "refbench", 100, parity_queries=64))
            cfg["c4_shard_1rank_rccl"] = guarded(lambda: ivf_config(args, pqv, torch, dev, local_rank, "c4", 10, parity_queries=64, rccl=True))
            cfg["c5"] = guarded(lambda: brute_measure(args, pqv, torch, dev, local_rank, "c5", 10, steps=5))
            cfg["refbench_from_parquet"] = guarded(lambda: from_parquet_config(args, pqv, torch, dev, local_rank))
            result["configs"] = cfg
            result["configs_note"] = ("BASELINE.json configs[1] (c2), the reference's own bench shape benches/query.rs:27-31 (refbench: 1 M x 1024, "
                                      "default n_clusters, K 100, nprobe 16), one configs[3] shard on one rank with the RCCL exchange in every step "
                                      "(c4_shard_1rank_rccl) and configs[4] (c5), each generated, built, timed and checked inside this run")
        for name, r in [("secondary_mixture", result.get("secondary_mixture"))] + list((result.get("configs") or {}).items()):
            if isinstance(r, dict) and isinstance(r.get("parity"), dict) and r["parity"].get("ok") is False:
                log(f"[bench] PARITY FAILURE in {name}")
                rc = 3
        import torch.distributed as _d
        if _d.is_initialized() and not use_dist:
            _d.destroy_process_group()
    if rank == 0:
        emit(real_stdout, result)
    if use_dist:
        dist.destroy_process_group()
    if rc:
        log("[bench] PARITY FAILURE against the CPU oracle")
        sys.exit(rc)


def replica_pass(args, pqv, torch, dist, dev, local_rank, rank, world, nq, steps):
    """Throughput mode for a corpus that fits one GPU (C2: 1 M x 128): every rank holds the whole corpus and its own
    index and searches its OWN query batch; no data-path collective.  Reported next to the sharded headline."""
    n, dim, kc, nprobe, _ = WORKLOADS["c2"]
    corpus_t = synth(torch, dev, 1234, n, dim)
    q = synth(torch, dev, 7 + rank, nq, dim)
    c = pqv.Corpus.from_device_ptr(corpus_t.data_ptr(), n, dim, device=local_rank, keepalive=corpus_t)
    idx = pqv.IndexBuilder(c).n_clusters(kc).max_iters(20).seed(42).workers(os.cpu_count() or 1).build()
    srch = pqv.Searcher(idx, c)
    streams = [torch.cuda.current_stream(), torch.cuda.Stream(device=dev)]
    rows = [torch.empty((nq, K), dtype=torch.int32, device=dev) for _ in range(2)]
    dd = [torch.empty((nq, K), dtype=torch.float32, device=dev) for _ in range(2)]

    def rstep(i):
        st = streams[i % 2]
        with torch.cuda.stream(st):
            srch.topk_device(q.data_ptr(), nq, K, nprobe, rows[i % 2].data_ptr(), dd[i % 2].data_ptr(), stream=st.cuda_stream)

    for i in range(4):
        rstep(i)
    dist.barrier(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(steps * 10):
        rstep(i)
    dist.barrier(); torch.cuda.synchronize()
    el = torch.tensor([time.perf_counter() - t0], dtype=torch.float64, device=dev)
    dist.all_reduce(el, op=dist.ReduceOp.MAX)
    return {"value": world * nq * steps * 10 / float(el.item()), "unit": "queries/s", "scaling": "weak",
            "ms_per_step": float(el.item()) / (steps * 10) * 1e3,
            "config": "c2 (1000000x128, n_clusters 100, nprobe 8) replicated on every rank, %d queries per rank and step" % nq,
            "note": "query-parallel replicas: no data-path collective, per-GPU work fixed"}


def brute_f64_check(torch, corpus_t, queries_t, rows, dist, sel, k):
    """f64 brute force ON THE SAME resident corpus for the queries `sel` (torch f64 matmul on the device, in row slabs --
    an independent checker: nothing of the library is involved): max relative error of the returned cosine distances
    against the true k smallest, and whether every row clearly inside the true top-k was returned."""
    n, dim = corpus_t.shape
    q = queries_t[sel].double()
    qn = q.norm(dim=1)
    best_d = torch.full((len(sel), k), float("inf"), dtype=torch.float64, device=corpus_t.device)
    best_r = torch.zeros((len(sel), k), dtype=torch.int64, device=corpus_t.device)
    slab = max(1, (1 << 28) // dim)
    for s0 in range(0, n, slab):
        x = corpus_t[s0:s0 + slab].double()
        d = 1.0 - (q @ x.T) / (qn[:, None] * x.norm(dim=1)[None, :])
        cd = torch.cat((best_d, d), dim=1)
        cr = torch.cat((best_r, torch.arange(s0, s0 + x.shape[0], device=x.device).expand(len(sel), -1)), dim=1)
        best_d, idx = torch.topk(cd, k, dim=1, largest=False, sorted=True)
        best_r = torch.gather(cr, 1, idx)
        del x, d, cd, cr
    ref_d, ref_r = best_d.cpu().numpy(), best_r.cpu().numpy()
    got_d, got_r = dist[sel].astype(np.float64), rows[sel].astype(np.int64)
    err = float(np.max(np.abs(got_d - ref_d) / np.maximum(np.abs(ref_d), 1e-3)))
    missed = 0
    for i in range(len(sel)):
        kth = ref_d[i, -1]
        tol = 1e-4 * max(abs(kth), 1e-3) + 1e-6
        clearly_in = ref_r[i][ref_d[i] < kth - tol]
        missed += len(set(clearly_in.tolist()) - set(got_r[i].tolist()))
    return {"checker": "f64 brute force of the SAME resident corpus (torch f64 matmul on the device, independent of the library)",
            "queries_checked": len(sel), "rows": n, "max_rel_dist_err": err, "tolerance": 1e-4,
            "ids_equal_fraction": float((got_r == ref_r).mean()), "clearly_closer_rows_missed": missed,
            "ok": bool(err <= 1e-4 and missed == 0)}


def brute_measure(args, pqv, torch, dev, local_rank, name, k, steps=5, corpus_t=None, queries_t=None):
    """BASELINE config 5: exhaustive cosine top-k of nq queries per step as a Q.V^T contraction on the matrix cores
    (pqv_brute_topk; an extension -- the reference has no cosine).  The 32 checked queries are answers of the timed run."""
    n, dim, _, _, nq = WORKLOADS[name]
    t_all = time.perf_counter()
    if corpus_t is None:
        corpus_t = synth(torch, dev, 1234, n, dim)
        queries_t = synth(torch, dev, 7, nq, dim)
        torch.cuda.synchronize()
    corpus = pqv.Corpus.from_device_ptr(corpus_t.data_ptr(), n, dim, device=local_rank, keepalive=corpus_t)
    q_host = queries_t.cpu().numpy()
    t0 = time.perf_counter()
    for _ in range(max(1, args.warmup)):
        rows, dist, nf = corpus.brute_topk(q_host, k, pqv.PQV_COSINE)
    torch.cuda.synchronize()
    first_s = time.perf_counter() - t0
    t0 = time.perf_counter()
    for _ in range(steps):
        rows, dist, nf = corpus.brute_topk(q_host, k, pqv.PQV_COSINE)
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    ms = elapsed / steps * 1e3
    flops = 2.0 * nq * n * dim
    f32_only = os.environ.get("PQV_BRUTE_F16", "1") == "0"
    i8 = not f32_only and os.environ.get("PQV_BRUTE_OP", "i8") != "f16"
    # dense MFMA peaks (MI355X_MICROARCH.md): f32 157.3 TF, f16 2.5 PF, int8 16x16x64 / 32x32x32 >= 3944 TOPS
    peak_tf = 157.3 if f32_only else 3944.0 if i8 else 2500.0
    op_name = "int8" if i8 else "f16"
    rec = {"value": nq * steps / elapsed, "unit": "queries/s", "ms_per_step": ms, "steps": steps,
           "config": f"{name}: brute-force cosine top-{k} over {n}x{dim} uniform f32, {nq} queries/step, Q.V^T on "
                     + ("v_mfma_f32_32x32x2_f32" if f32_only else ("v_mfma_i32_32x32x32_i8" if i8 else "v_mfma_f32_32x32x16_f16") + " (screen) + f32 re-scoring"),
           "dtype": "f32" if f32_only else f"{op_name} screen ({'int32' if i8 else 'f32'} accumulate) + f32 exact re-scoring of what it lets through",
           "warmup_s_including_image_build": first_s,
           "roofline": {"bound": "mfma",
                        "kernel": "brute_mfma_kernel (128x128 tiles, f32 in / f32 accumulate)" if f32_only else
                                  f"brute_f16_kernel (256x256 tiles of normalised {op_name} images, 128-byte K stages, XCD-aware grid) + brute_rescore_kernel; "
                                  "the first 2048 rows through brute_mfma_kernel (f32)",
                        "achieved": flops / (ms * 1e-3) / 1e12, "peak": peak_tf, "unit": "TFLOP/s",
                        "frac": flops / (ms * 1e-3) / 1e12 / peak_tf, "traffic": None,
                        "algo_flops_per_step": flops,
                        "frac_of_f32_mfma_peak": flops / (ms * 1e-3) / 1e12 / 157.3,
                        "note": "achieved = 2*nq*n*dim / whole-step wall time (queries uploaded, 5 progressive "
                                "row ranges with a host check each, select passes, results downloaded): a lower bound for the kernel; "
                                "peak = the dense MFMA rate of the screen's operand form: int8 3944 TOPS (default), f16 2.5 PF "
                                "(PQV_BRUTE_OP=f16), f32 157.3 TF (PQV_BRUTE_F16=0)",
                        "unit_note": "TFLOP/s reads Tera-op/s for the int8 form"}}
    rec["parity"] = brute_f64_check(torch, corpus_t, queries_t, rows, dist, spread(nq, 32), k)
    rec["seconds"] = time.perf_counter() - t_all
    corpus.close()
    return rec


def bench_brute(args, pqv, torch, corpus, corpus_t, queries_t, n, dim, nq, rank, world, real_stdout):
    """`--workload c5 / c5s` as the line of its own."""
    if world != 1:
        raise SystemExit("the c5 workload is single-GPU")
    steps = args.steps if args.steps > 0 else 5
    rec = brute_measure(args, pqv, torch, corpus_t.device, corpus_t.device.index or 0, args.workload, K, steps=steps,
                        corpus_t=corpus_t, queries_t=queries_t)
    result = {
        "metric": "topk_queries_per_s_k10", "value": rec["value"], "unit": "queries/s",
        "n_gpus": 1, "steps": steps, "warmup": args.warmup, "ms_per_step": rec["ms_per_step"],
        "higher_is_better": True, "scaling": "strong", "vs_baseline": None,
        "dtype": rec["dtype"], "data": "synthetic",
        "config": {"workload": rec["config"], "rows": n, "dim": dim, "k": K, "queries_per_step": nq, "shards": 1},
        "timed_region_s": rec["ms_per_step"] * steps * 1e-3,
        "roofline": rec["roofline"], "parity": rec["parity"],
    }
    rc = 0 if rec["parity"]["ok"] else 3
    if not args.no_cpu:
        # CPU column: f64 numpy brute force on a row slice (multithreaded BLAS), extrapolated linearly in rows
        q_host = queries_t.cpu().numpy()
        m = min(n, 200_000)
        sub = corpus_t[:m].cpu().numpy().astype(np.float64)
        qs = q_host[:32].astype(np.float64)
        t1 = time.perf_counter()
        s = qs @ sub.T
        d = 1.0 - s / (np.linalg.norm(qs, axis=1)[:, None] * np.linalg.norm(sub, axis=1)[None, :])
        np.argsort(d, axis=1, kind="stable")[:, :K]
        spent = time.perf_counter() - t1
        result["cpu_baseline"] = {
            "value": 32 / spent * (m / n), "unit": "queries/s", "cores": os.cpu_count(), "kind": "port",
            "sample": f"numpy f64 matmul + argsort, 32 queries x {m} rows in {spent:.2f} s, scaled by {m}/{n} rows"}
    emit(real_stdout, result)
    if rc:
        sys.exit(rc)


def cpu_baseline(args, index, corpus_t, qs, rows_t, dist_t, nprobe, nq, searcher=None):
    """Two columns:
    faithful  -- the CPU oracle (a port of src/ivf/search.rs:83-142, -O3 -march=native -ffp-contract=off on THIS host), ONE
                 thread as the reference's query loop is (search.rs:115), on a bounded sample of the step's queries; the GPU
                 results are checked against it bit for bit.
    generous  -- the same loops built -ffast-math (the compiler may vectorise the reduction like a tuned SIMD path) and
                 `nproc` threads, each answering whole queries: an optimistic CPU, timing only."""
    from concurrent.futures import ThreadPoolExecutor
    from oracle_binding import Oracle, build_oracle
    build_oracle("native")
    o = Oracle(native=True)
    host = corpus_t.cpu().numpy()
    blob = index.to_bytes()
    oidx = o.index_from_bytes(blob)
    grows = rows_t.cpu().numpy().view(np.uint32)
    gdist = dist_t.cpu().numpy()
    done, spent = 0, 0.0
    st = {"ids": True, "dist": True, "tie": True, "groups": 0, "replayed": 0, "replay_ok": True}

    def check(q0, orows, odist):
        b = len(orows)
        g = grows[q0:q0 + b]
        st["ids"] &= bool((orows == g).all())
        st["dist"] &= bool((odist.view(np.uint32) == gdist[q0:q0 + b].view(np.uint32)).all())
        # pqv_topk_device orders equal output distances by (d2, position); Rust orders them by heap
        # history (the host API pqv_topk replays that exactly).  Inside a group of equal distance the
        # id SETS must still agree.
        for i in range(b):
            if (orows[i] == g[i]).all():
                continue
            if searcher is not None:        # the host API replays the reference's heap for exactly these queries
                hr, hd, _, _ = searcher.topk(qs[q0 + i:q0 + i + 1], K, nprobe)
                st["replayed"] += 1
                st["replay_ok"] &= bool((hr[0] == orows[i]).all()) and bool((hd.view(np.uint32)[0] == odist.view(np.uint32)[i]).all())
            j = 0
            while j < K:
                e = j
                while e + 1 < K and odist[i, e + 1] == odist[i, j]:
                    e += 1
                if e > j:
                    st["groups"] += 1
                st["tie"] &= sorted(orows[i, j:e + 1].tolist()) == sorted(g[i, j:e + 1].tolist())
                j = e + 1

    # faithful column: ONE thread, timed
    chunk = 4
    while done < nq and spent < args.cpu_seconds:
        b = min(chunk, nq - done)
        t0 = time.perf_counter()
        orows, odist, onf, _ = oidx.topk_batch(host, qs[done:done + b], K, nprobe)
        spent += time.perf_counter() - t0
        check(done, orows, odist)
        done += b
        chunk = min(64, chunk * 2)
    timed = done
    # parity only (not timed): the sample is widened to >= 64 queries spread over the batch, one query per host thread
    # (results are independent per query)
    extra = [q for q in range(nq - 1, timed - 1, -max(1, (nq - timed) // 64))][:max(0, min(args.parity_queries, nq) - timed)]
    if extra:
        nthr_p = max(1, min(len(extra), os.cpu_count() or 1, 64))
        with ThreadPoolExecutor(max_workers=nthr_p) as ex:
            for q, (orows, odist, _, _) in zip(extra, ex.map(lambda q: oidx.topk_batch(host, qs[q:q + 1], K, nprobe), extra)):
                check(q, orows, odist)
        done += len(extra)
    ids_ok, dist_ok, ids_tie_ok, tie_groups, replayed, replay_ok = st["ids"], st["dist"], st["tie"], st["groups"], st["replayed"], st["replay_ok"]
    out = {"value": timed / spent, "unit": "queries/s", "cores": 1, "kind": "port",
           "threads_note": "one thread, as the reference's query loop (search.rs:115)",
           "sample": f"first {timed} of the step's {nq} queries, in-memory corpus, oracle -O3 -march=native "
                     f"-ffp-contract=off, {spent:.1f} s (+ {done - timed} more queries checked for parity only, one per host thread)",
           "host_cpus": os.cpu_count(),
           "parity": {"queries_checked": done, "row_idx_identical": ids_ok, "dist_bit_identical": dist_ok,
                      "row_idx_identical_up_to_order_inside_equal_distance_groups": ids_tie_ok,
                      "equal_distance_groups_seen": tie_groups,
                      "queries_replayed_through_pqv_topk": replayed,
                      "row_idx_identical_after_replay": bool(ids_ok or (replayed > 0 and replay_ok))}}
    # generous column
    try:
        build_oracle("fast")
        of = Oracle(fast=True)
        fidx = of.index_from_bytes(blob)
        nthr = os.cpu_count() or 1
        # one whole query per thread: with every core streaming its own ~|C| x dim x 4 bytes the host is memory-bound,
        # so more queries per thread only lengthen the run without changing the rate
        per = 1
        total = min(nq, nthr)
        def work(t):
            a = t * per
            if a >= total:
                return 0
            fidx.topk_batch(host, qs[a:min(total, a + per)], K, nprobe)
            return min(total, a + per) - a
        t0 = time.perf_counter()
        with ThreadPoolExecutor(max_workers=nthr) as ex:
            n_done = sum(ex.map(work, range(nthr)))
        fs = time.perf_counter() - t0
        out["generous"] = {"value": n_done / fs, "unit": "queries/s", "cores": nthr, "kind": "port",
                           "sample": f"{n_done} queries over {nthr} threads (whole queries per thread), oracle -O3 -march=native "
                                     f"-ffast-math, {fs:.1f} s",
                           "note": "optimistic CPU column of BASELINE.md: timing only, its sums are not the reference's"}
    except Exception as e:          # the faithful column is the contract; this one is best effort
        out["generous"] = {"error": str(e)}
    return out


if __name__ == "__main__":
    main()
