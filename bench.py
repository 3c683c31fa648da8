#!/usr/bin/env python3
"""bench.py -- top-k QPS (k=10) of the MI355X IVF hot path on BASELINE.json's configs.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--workload c2|c3|c1|tiny] [--nq Q]

A "step" is one pass of the hot path (centroid probe -> candidate re-rank -> top-k merge)
over one batch of Q synthetic queries, with corpus, index and queries already resident in
HBM.  Default workload = BASELINE.json configs[1] (C2): 1 M x 128 uniform f32, n_clusters
100, k 10, nprobe 8.

`--workload c4` is the sharded 100 M x 768 configuration: every rank holds one 12.5 M-row shard
(weak scaling; at N = 8 the job searches the whole 100 M corpus).

N > 1 (launched by torch.distributed.run, one rank per GPU): the corpus is cut into N
contiguous row ranges, one shard + its own IVF index per GPU (the reference's per-file
index, src/df_vector/index_exec.rs:85-164); every rank searches the whole query batch on
its shard and the per-shard top-k lists are exchanged with one RCCL all-gather per step
and merged by (distance, shard, position).  Total work is fixed => "scaling": "strong".

Prints ONE JSON line on rank 0 (contract in the task brief) with `roofline` and
`cpu_baseline` objects.  torch is plumbing here: device tensors, streams, RCCL.
"""
import argparse
import ctypes as C
import json
import os
import subprocess
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
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
    "c2s2": (500_000, 128, 100, 8, 1024),   # what one rank of `c2 --gpus 2/4/8` searches (tuning aid)
    "c2s4": (250_000, 128, 100, 8, 1024),
    "c2s8": (125_000, 128, 100, 8, 1024),
    "tiny": (20_000, 64, 16, 4, 64),       # plumbing check
}
K = 10
HBM_PEAK_GBS = 8000.0                      # MI355X_MICROARCH.md: HBM3E 8 TB/s spec


def log(*a):
    print(*a, file=sys.stderr, flush=True)


def main():
    # Libraries (RCCL prints a version banner) may write to fd 1; the contract is ONE JSON line
    # on stdout.  Point fd 1 at stderr for the run and keep the real stdout for the result.
    sys.stdout.flush()
    real_stdout = os.dup(1)
    os.dup2(2, 1)

    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--workload", default="c2", choices=sorted(WORKLOADS))
    ap.add_argument("--nq", type=int, default=0, help="queries per step (default per workload)")
    ap.add_argument("--k", type=int, default=10, help="neighbours per query (BASELINE metric: 10)")
    ap.add_argument("--cpu-seconds", type=float, default=12.0, help="CPU baseline budget")
    ap.add_argument("--no-cpu", action="store_true")
    ap.add_argument("--layout", default="ivf", choices=["ivf", "row"])
    ap.add_argument("--force-dist", action="store_true",
                    help="initialise RCCL and run the shard exchange even with one rank (path check)")
    ap.add_argument("--backend", default="nccl", choices=["nccl", "gloo"],
                    help="gloo lets several ranks share one GPU (path check only; the real run uses RCCL)")
    ap.add_argument("--streams", type=int, default=2,
                    help="HIP streams the steps alternate between (1 = strictly serial steps)")
    ap.add_argument("--no-timing", action="store_true",
                    help="do not record HIP events around the kernels (roofline.kernel_ms is then 0)")
    ap.add_argument("--multi", default="auto", choices=["auto", "replica", "shard"],
                    help="N > 1: 'replica' = every rank holds the whole corpus and searches its OWN query batch "
                         "(throughput mode, no data-path collective; the sharded-corpus path is then timed as well and "
                         "reported under 'sharded'); 'shard' = the corpus is split N ways, every rank searches the same "
                         "batch, top-k lists are exchanged over RCCL and merged.  auto: replica, except c4 (per-rank "
                         "shards of a corpus that does not fit one GPU) and --force-dist")
    ap.add_argument("--single", type=int, default=0,
                    help="also time this many single-query calls (latency mode) and report them")
    args = ap.parse_args()
    global K
    K = args.k

    import torch
    import torch.distributed as dist
    import pq_vector_amd as pqv

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            raise SystemExit("--gpus N > 1 must be launched with torch.distributed.run (one rank per GPU)")
        raise SystemExit(f"WORLD_SIZE={world} does not match --gpus {args.gpus}")
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

    n_total, dim, n_clusters, nprobe, nq_default = WORKLOADS[args.workload]
    nq = args.nq or nq_default
    from pq_vector_amd.sharding import ShardExchange, shard_range
    weak = args.workload == "c4"           # per-rank shard size fixed: N ranks hold N x rows
    replica = world > 1 and not weak and not args.force_dist and args.multi != "shard"
    if args.multi == "replica" and (weak or world == 1):
        replica = False
    if weak:
        lo, hi = rank * n_total, (rank + 1) * n_total
        n_total = n_total * world
    elif replica:
        lo, hi = 0, n_total                # the whole corpus on every rank; the QUERIES are what is sharded
    else:
        lo, hi = shard_range(rank, world, n_total)
    n_shard = hi - lo

    # ---- synthetic data: the reference's bench recipe (benches/bench_util.rs:12-64) ------
    # i.i.d. uniform [0,1) f32 with 24-bit resolution, corpus seed 1234, query seed 7.
    g = torch.Generator(device=dev)
    g.manual_seed(1234 if replica else 1234 + rank)
    corpus_t = torch.empty((n_shard, dim), dtype=torch.float32, device=dev)
    step_rows = max(1, (1 << 28) // (dim * 4))
    for s in range(0, n_shard, step_rows):
        e = min(n_shard, s + step_rows)
        u = torch.randint(0, 1 << 24, (e - s, dim), generator=g, device=dev, dtype=torch.int32)
        corpus_t[s:e] = u.to(torch.float32) * (1.0 / (1 << 24))
        del u
    gq = torch.Generator(device=dev)
    gq.manual_seed(7 + rank if replica else 7)          # replicas search different batches
    queries_t = (torch.randint(0, 1 << 24, (nq, dim), generator=gq, device=dev, dtype=torch.int32)
                 .to(torch.float32) * (1.0 / (1 << 24)))
    torch.cuda.synchronize()

    corpus = pqv.Corpus.from_device_ptr(corpus_t.data_ptr(), n_shard, dim, device=local_rank,
                                        keepalive=corpus_t)
    if args.workload in ("c5", "c5s"):
        return bench_brute(args, pqv, torch, corpus, corpus_t, queries_t, n_shard, dim, nq, rank, world,
                           real_stdout)

    # ---- index build on the GPU (max_iters 20, seed 42: src/ivf/parquet.rs:37-38) --------
    workers = os.cpu_count() or 1
    t0 = time.perf_counter()
    index = pqv.IndexBuilder(corpus).n_clusters(n_clusters).max_iters(20).seed(42).workers(workers).build() \
        if n_clusters else pqv.IndexBuilder(corpus).max_iters(20).seed(42).workers(workers).build()
    build_s = time.perf_counter() - t0
    flags = pqv.PQV_LAYOUT_ROW_ORDER if args.layout == "row" else pqv.PQV_LAYOUT_IVF_ORDERED
    t0 = time.perf_counter()
    searcher = pqv.Searcher(index, corpus, flags)
    layout_s = time.perf_counter() - t0
    if rank == 0:
        log(f"[bench] shard rows={n_shard} dim={dim} n_clusters={index.n_clusters} build={build_s:.3f}s "
            f"relayout={layout_s:.3f}s")

    # ---- device outputs: one set per stream lane ------------------------------------------------
    # Steps alternate between `--streams` HIP streams (default 2): the library keeps one scratch lane
    # per stream, so step i + 1's probe / bucketing / seed kernels and the head of its screen kernel run
    # in the tail of step i's screen kernel.  Each step is still one complete pass over one batch.
    n_lanes = max(1, args.streams)
    lane_streams = [torch.cuda.current_stream()] + [torch.cuda.Stream(device=dev) for _ in range(n_lanes - 1)]
    rows_l = [torch.empty((nq, K), dtype=torch.int32, device=dev) for _ in range(n_lanes)]
    dist_l = [torch.empty((nq, K), dtype=torch.float32, device=dev) for _ in range(n_lanes)]
    nf_l = [torch.empty((nq,), dtype=torch.int32, device=dev) for _ in range(n_lanes)]
    nc_l = [torch.empty((nq,), dtype=torch.int64, device=dev) for _ in range(n_lanes)]
    rows_t, dist_t, nf_t, nc_t = rows_l[0], dist_l[0], nf_l[0], nc_l[0]
    if weak:
        bases = [r * (n_total // world) for r in range(world)]
    else:
        bases = [shard_range(r, world, n_total)[0] for r in range(world)]
    xchg_l = [ShardExchange(world, nq, K, dev, always_collective=args.force_dist,
                            row_bases=bases if args.backend == "nccl" else None) for _ in range(n_lanes)]
    xchg = xchg_l[0]
    step_no = [0]
    stream = lane_streams[0].cuda_stream

    def step():
        lane = step_no[0] % n_lanes
        step_no[0] += 1
        st = lane_streams[lane]
        with torch.cuda.stream(st):
            # hot path on this rank's shard; asynchronous on the lane's stream
            searcher.topk_device(queries_t.data_ptr(), nq, K, nprobe, rows_l[lane].data_ptr(), dist_l[lane].data_ptr(),
                                 nf_l[lane].data_ptr(), nc_l[lane].data_ptr(), stream=st.cuda_stream)
            if not use_dist or replica:
                return dist_l[lane], rows_l[lane]
            # exchange: one all-gather of k x {dist, row} per query, then the merge keyed (dist, shard,
            # position) -- pq_vector_amd/sharding.py
            if xchg_l[lane].fast:
                return xchg_l[lane].exchange_u32(dist_l[lane], rows_l[lane])
            return xchg_l[lane].exchange(dist_l[lane], rows_l[lane].to(torch.int64) & 0xFFFFFFFF, lo)

    def barrier():
        if use_dist:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        step()
    barrier()
    searcher.set_timing(not args.no_timing)
    t0 = time.perf_counter()
    for _ in range(args.steps):
        out_d, out_r = step()
    barrier()
    elapsed = time.perf_counter() - t0
    searcher.set_timing(False)
    rerank_ms, total_ms, ncalls = searcher.timing_read()
    # With several stream lanes the kernels of consecutive steps share the GPU, so the per-launch
    # durations above include that sharing.  A short serial pass (outside the timed region, one
    # lane, same kernels) gives the isolated per-launch duration for comparison.
    serial_rr_ms = None
    if n_lanes > 1 and not args.no_timing:
        searcher.set_timing(True)
        for _ in range(min(10, max(2, args.steps))):
            step_no[0] = 0
            step()
            torch.cuda.synchronize()
        searcher.set_timing(False)
        s_rr, s_tot, s_n = searcher.timing_read()
        serial_rr_ms = s_rr / max(1, s_n)

    el = torch.tensor([elapsed], dtype=torch.float64, device=dev)
    if use_dist:
        dist.all_reduce(el, op=dist.ReduceOp.MAX)
    elapsed = float(el.item())

    # ---- algorithmic bytes of the dominant (re-rank) kernel --------------------------------
    ncand = nc_t.cpu().numpy().astype(np.int64)
    cand_rows = int(ncand.sum())
    algo_bytes = cand_rows * (4 * dim + 4)          # embedding row + its u32 id
    rr_ms = rerank_ms / max(1, ncalls)
    achieved = algo_bytes / (rr_ms * 1e-3) / 1e9 if rr_ms > 0 else 0.0

    mode = os.environ.get("PQV_RERANK_MODE", "auto")
    pairs_per_cluster = nq * nprobe / max(1, int(index.n_clusters))
    mean_len = n_shard / max(1, int(index.n_clusters))
    wide = dim % 64 == 0 and args.layout == "ivf" and os.environ.get("PQV_FILTER_VARIANT", "0") == "0"
    wide_any = (wide and os.environ.get("PQV_TILE_FILTER", "1") != "0" and K <= 128
                and mean_len >= 3 * (512 if mean_len >= 4096 else 256))           # api.cpp: wide_any_batch
    tile = mode == "tile" or (mode != "stream" and (pairs_per_cluster >= 4 or wide_any))
    screened = tile and (os.environ.get("PQV_TILE_FILTER", "1") == "2" or (
        os.environ.get("PQV_TILE_FILTER", "1") != "0" and K <= (128 if wide else 32) and pairs_per_cluster >= (0 if wide else 4)
        and mean_len >= (3 * (512 if mean_len >= 4096 else 256) if wide else 4096)))
    kernel = ("wide_seed_kernel (MFMA upper-bound thresholds) + wide_filter_kernel (batched cluster-major re-rank, "
              + ("64 queries staged in LDS" if dim <= 128 else "32 queries per quad") + " per streamed row tile from the blocked "
              "copy, MFMA lower-bound screen, exact re-evaluation of the survivors)" if screened and wide
              else "tile_rerank_kernel seed window + tile_filter_kernel (batched cluster-major re-rank, 16 queries per "
              "streamed row tile, MFMA lower-bound screen, exact re-evaluation of the survivors)" if screened
              else "tile_rerank_kernel (batched cluster-major re-rank, 16 queries per streamed row tile)" if tile
              else "stream_kernel (one candidate stream per (query, probed list))")
    traffic = None
    tpath = os.path.join(ROOT, "profiles", f"r01_pmc_traffic_{args.workload}_{'screen' if screened else 'tile' if tile else 'stream'}.json")
    if os.path.exists(tpath):
        try:
            tj = json.load(open(tpath))
            if tj.get("nq") == nq and tj.get("n_gpus", 1) == world:
                traffic = tj.get("hbm_bytes_per_launch")
        except Exception:
            traffic = None

    flops = 3 * dim * cand_rows                       # sub, mul, add per element
    result = {
        "metric": f"topk_queries_per_s_k{K}",
        "value": (world if replica else 1) * nq * args.steps / elapsed,
        "unit": "queries/s",
        "n_gpus": world,
        "steps": args.steps,
        "warmup": args.warmup,
        "ms_per_step": elapsed / args.steps * 1e3,
        "higher_is_better": True,
        # N = 1 is the first point of the default (replica) series: per-GPU work fixed as N grows
        "scaling": "weak" if (weak or replica or (world == 1 and args.multi != "shard" and not args.force_dist)) else "strong",
        "vs_baseline": None,
        "dtype": "f32",
        "data": "synthetic",
        "config": {"workload": f"{args.workload}: {n_total}x{dim} uniform f32, n_clusters {index.n_clusters}"
                               f"{' per shard' if world > 1 and not replica else ''}, k {K}, nprobe {nprobe}, "
                               f"{nq} queries/step, layout {args.layout}",
                   "rows": n_total, "dim": dim, "n_clusters": int(index.n_clusters), "k": K,
                   "nprobe": nprobe, "queries_per_step": nq, "shards": 1 if replica else world,
                   "parallelism": ("replicas x%d: corpus replicated, query batches sharded (%d queries per step over "
                                   "the job), no data-path collective" % (world, world * nq)) if replica
                   else ("corpus sharded x%d, RCCL all-gather of the per-shard top-k + device merge" % world) if world > 1
                   else "single GPU"},
        "index_build_vectors_per_s": n_shard / build_s,
        "index_build_s": build_s,
        "candidates_per_query": cand_rows / nq,
        # filled in below
    }
    hbm_view = {"achieved_GBps": achieved, "peak_GBps": HBM_PEAK_GBS, "frac": achieved / HBM_PEAK_GBS,
                "algo_bytes_per_launch": algo_bytes,
                "note": "ALGORITHMIC bytes (every candidate row + its id, once per query that probes it) / kernel "
                        "time; the batched kernels serve 16 queries from one streamed row tile, so this exceeds the "
                        "HBM peak by design -- `traffic` (PMC) is the real memory-side volume"}
    valu_view = {"ref_ops_per_launch": flops, "achieved_tops": flops / (rr_ms * 1e-3) / 1e12 if rr_ms > 0 else 0.0,
                 "peak_tops_measured": 63.0,
                 "note": "reference arithmetic = 3 non-fusable f32 ops per element; ~63 T such ops/s measured "
                         "(tools/valu_ubench.hip).  The screened kernel skips most of them, so this is a "
                         "speed-up figure there, not a utilisation"}
    if screened:
        mf = 2.0 * dim * cand_rows          # the Q.X^T contraction of the screen: 2 flops per (row, query, dim)
        # f16 operands (default where dim % 128 == 0 and dim <= 1024): the contraction runs on v_mfma_f32_16x16x32_f16,
        # whose dense peak is ~2.5 PFLOP/s (MI355X_MICROARCH.md); f32 operands: 157.3 TFLOP/s
        f16 = wide and dim % 128 == 0 and dim <= 1024 and os.environ.get("PQV_SCREEN_F16", "1") != "0"
        peak = 2500.0 if f16 else 157.3
        ach = mf / (rr_ms * 1e-3) / 1e12 if rr_ms > 0 else 0.0
        result["roofline"] = {"bound": "mfma", "kernel": kernel + (" [f16 operands]" if f16 else " [f32 operands]"),
                              "achieved": ach, "peak": peak, "unit": "TFLOP/s", "frac": ach / peak,
                              "traffic": traffic, "algo_flops_per_launch": mf, "kernel_ms": rr_ms,
                              "hot_path_ms_per_step": total_ms / max(1, ncalls),
                              "note": ("dominant work = the MFMA contraction of the lower-bound screen (threshold seed, select and "
                                       "exact re-evaluation of survivors are inside kernel_ms)"
                                       + ("; with f16 operands the matrix pipe is busy only a few percent of the time -- the "
                                          "kernel is bound by vector-ALU issue (which does not overlap MFMAs on gfx950) and "
                                          "latency, see f32_mfma_view / valu_view / DESIGN.md 5.1c" if f16 else "")),
                              "hbm_view": hbm_view, "valu_view": valu_view}
        if f16:
            result["roofline"]["f32_mfma_view"] = {
                "achieved_tflops": ach, "peak": 157.3, "frac": ach / 157.3,
                "note": "the same flops against the f32 MFMA peak: what the f32-operand form of the kernel "
                        "(PQV_SCREEN_F16=0) is bounded by; it measured 0.37 isolated on C2"}
    else:
        result["roofline"] = {"bound": "hbm", "kernel": kernel, "achieved": achieved, "peak": HBM_PEAK_GBS,
                              "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS, "traffic": traffic,
                              "algo_bytes_per_launch": algo_bytes, "kernel_ms": rr_ms,
                              "hot_path_ms_per_step": total_ms / max(1, ncalls),
                              "note": hbm_view["note"] if tile else "one candidate stream per (query, probed list)",
                              "valu_view": valu_view}

    r = result["roofline"]
    r["streams"] = n_lanes
    if serial_rr_ms:
        peak = r["peak"]
        work = mf / 1e12 if screened else algo_bytes / 1e9
        r["isolated"] = {"kernel_ms": serial_rr_ms, "achieved": work / (serial_rr_ms * 1e-3),
                         "frac": work / (serial_rr_ms * 1e-3) / peak,
                         "note": "same kernels, steps issued one at a time on one stream (not in the timed region)"}
        step_s = elapsed / args.steps
        r["aggregate"] = {"achieved": work / step_s, "frac": work / step_s / peak,
                          "note": "algorithmic work of one step / step period of the timed region (all kernels of "
                                  "the hot path and the overlap between consecutive steps included)"}
        r["note"] = (r.get("note", "") + "; steps alternate between %d streams, so kernel_ms (HIP events in the timed "
                     "region, matches rocprofv3) is the duration of a launch that shares the GPU with the "
                     "neighbouring step's kernels" % n_lanes).lstrip("; ")
    result["counters"] = searcher.counters()
    if use_dist and not replica and xchg.fast:
        # the library merge kernel against the torch stable-sort merge of the same gathered lists
        ref_d, ref_r = xchg.exchange(dist_t, rows_t.to(torch.int64) & 0xFFFFFFFF, lo)
        ok = torch.tensor([int(torch.equal(ref_d, out_d) and torch.equal(ref_r, out_r))], device=dev)
        dist.all_reduce(ok, op=dist.ReduceOp.MIN)
        result["exchange_check"] = bool(ok.item())

    # ---- N > 1, replica mode: the sharded-corpus path (north_star's row-group shards + RCCL merge), timed too ----
    if replica:
        result["sharded"] = sharded_pass(args, pqv, torch, dist, corpus_t, n_total, dim, n_clusters, nprobe, nq, rank, world,
                                         local_rank, dev, flags, n_lanes, lane_streams)

    # ---- optional latency mode: one query per call through the same device API ------------
    if args.single and rank == 0:
        lat = []
        for i in range(min(args.single, nq) + 5):
            q1 = queries_t[i % nq:i % nq + 1]
            torch.cuda.synchronize()
            t1 = time.perf_counter()
            searcher.topk_device(q1.data_ptr(), 1, K, nprobe, rows_t.data_ptr(), dist_t.data_ptr(),
                                 nf_t.data_ptr(), nc_t.data_ptr(), stream=stream)
            torch.cuda.synchronize()
            lat.append(time.perf_counter() - t1)
        lat = np.array(lat[5:]) * 1e6
        result["single_query"] = {"calls": int(lat.size), "p50_us": float(np.percentile(lat, 50)),
                                  "p99_us": float(np.percentile(lat, 99)), "mean_us": float(lat.mean()),
                                  "qps": float(1e6 / lat.mean()),
                                  "note": "one query per call, host-synchronised after each (8 launches on the screened path, 4 on the streaming path)"}
    # ---- CPU baseline (rank 0, N = 1): the oracle, natively compiled, one thread -----------
    if rank == 0 and world == 1 and not args.no_cpu:
        result["cpu_baseline"] = cpu_baseline(args, index, corpus_t, queries_t, rows_t, dist_t, nprobe, nq)
    if rank == 0:
        sys.stdout.flush()
        os.write(real_stdout, (json.dumps(result) + "\n").encode())
    if use_dist:
        dist.destroy_process_group()


def sharded_pass(args, pqv, torch, dist, corpus_t, n_total, dim, n_clusters, nprobe, nq, rank, world, local_rank, dev,
                 flags, n_lanes, lane_streams):
    """The same corpus split N ways by contiguous row ranges (one index per shard, as the reference has one per
    file): every rank searches the SAME batch on its shard, two all-gathers move k x {distance, row} per query and
    every rank merges (pq_vector_amd/sharding.py).  Total work is fixed, so this is the strong-scaling number."""
    from pq_vector_amd.sharding import ShardExchange, shard_range
    lo, hi = shard_range(rank, world, n_total)
    sub = corpus_t[lo:hi]
    c_s = pqv.Corpus.from_device_ptr(sub.data_ptr(), hi - lo, dim, device=local_rank, keepalive=sub)
    b = pqv.IndexBuilder(c_s).max_iters(20).seed(42).workers(os.cpu_count() or 1)
    idx_s = b.n_clusters(n_clusters).build() if n_clusters else b.build()
    srch = pqv.Searcher(idx_s, c_s, flags)
    gq = torch.Generator(device=dev)
    gq.manual_seed(7)                                     # one batch, identical on every rank
    q = (torch.randint(0, 1 << 24, (nq, dim), generator=gq, device=dev, dtype=torch.int32).to(torch.float32) * (1.0 / (1 << 24)))
    bases = [shard_range(r, world, n_total)[0] for r in range(world)]
    xs = [ShardExchange(world, nq, K, dev, row_bases=bases if args.backend == "nccl" else None) for _ in range(n_lanes)]
    rows_l = [torch.empty((nq, K), dtype=torch.int32, device=dev) for _ in range(n_lanes)]
    dist_l = [torch.empty((nq, K), dtype=torch.float32, device=dev) for _ in range(n_lanes)]
    nf_l = [torch.empty((nq,), dtype=torch.int32, device=dev) for _ in range(n_lanes)]
    nc_l = [torch.empty((nq,), dtype=torch.int64, device=dev) for _ in range(n_lanes)]

    def sstep(i):
        lane = i % n_lanes
        st = lane_streams[lane]
        with torch.cuda.stream(st):
            srch.topk_device(q.data_ptr(), nq, K, nprobe, rows_l[lane].data_ptr(), dist_l[lane].data_ptr(),
                             nf_l[lane].data_ptr(), nc_l[lane].data_ptr(), stream=st.cuda_stream)
            if xs[lane].fast:
                return xs[lane].exchange_u32(dist_l[lane], rows_l[lane])
            return xs[lane].exchange(dist_l[lane], rows_l[lane].to(torch.int64) & 0xFFFFFFFF, lo)

    for i in range(args.warmup):
        sstep(i)
    dist.barrier(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(args.steps):
        out_d, out_r = sstep(i)
    dist.barrier(); torch.cuda.synchronize()
    el = torch.tensor([time.perf_counter() - t0], dtype=torch.float64, device=dev)
    dist.all_reduce(el, op=dist.ReduceOp.MAX)
    elapsed = float(el.item())
    out = {"value": nq * args.steps / elapsed, "unit": "queries/s", "ms_per_step": elapsed / args.steps * 1e3,
           "scaling": "strong", "rows_per_rank": hi - lo,
           "note": "same corpus split %d ways (one index per shard), every rank searches the same %d-query batch, "
                   "RCCL all-gather of k x {distance, row} per query + device merge on every rank" % (world, nq)}
    if xs[0].fast:
        lane = (args.steps - 1) % n_lanes
        ref_d, ref_r = xs[lane].exchange(dist_l[lane], rows_l[lane].to(torch.int64) & 0xFFFFFFFF, lo)
        ok = torch.tensor([int(torch.equal(ref_d, out_d) and torch.equal(ref_r, out_r))], device=dev)
        dist.all_reduce(ok, op=dist.ReduceOp.MIN)
        out["exchange_check"] = bool(ok.item())
    return out


def bench_brute(args, pqv, torch, corpus, corpus_t, queries_t, n, dim, nq, rank, world, real_stdout):
    """BASELINE config 5: exhaustive cosine top-k of nq queries per step as a Q.V^T contraction on
    the f32 matrix cores (pqv_brute_topk; an extension -- the reference has no cosine)."""
    if world != 1:
        raise SystemExit("the c5 workload is single-GPU")
    q_host = queries_t.cpu().numpy()
    for _ in range(max(1, args.warmup)):
        rows, dist, nf = corpus.brute_topk(q_host, K, pqv.PQV_COSINE)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        rows, dist, nf = corpus.brute_topk(q_host, K, pqv.PQV_COSINE)
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    ms = elapsed / args.steps * 1e3
    flops = 2.0 * nq * n * dim
    result = {
        "metric": "topk_queries_per_s_k10", "value": nq * args.steps / elapsed, "unit": "queries/s",
        "n_gpus": 1, "steps": args.steps, "warmup": args.warmup, "ms_per_step": ms,
        "higher_is_better": True, "scaling": "strong", "vs_baseline": None, "dtype": "f32",
        "data": "synthetic",
        "config": {"workload": f"{args.workload}: brute-force cosine top-{K} over {n}x{dim} uniform f32, "
                               f"{nq} queries/step, Q.V^T on v_mfma_f32_32x32x2_f32",
                   "rows": n, "dim": dim, "k": K, "queries_per_step": nq, "shards": 1},
        "roofline": {"bound": "mfma", "kernel": "brute_mfma_kernel (128x128 tiles, f32 in / f32 accumulate)",
                     "achieved": flops / (ms * 1e-3) / 1e12, "peak": 157.3, "unit": "TFLOP/s",
                     "frac": flops / (ms * 1e-3) / 1e12 / 157.3, "traffic": None,
                     "algo_flops_per_step": flops,
                     "note": "achieved = 2*nq*n*dim / whole-step wall time (queries uploaded, 5 progressive "
                             "row ranges, select passes, results downloaded): a lower bound for the kernel"},
    }
    if not args.no_cpu:
        # f64 numpy brute force on a row slice (multithreaded BLAS), extrapolated linearly in rows
        m = min(n, 200_000)
        sub = corpus_t[:m].cpu().numpy().astype(np.float64)
        qs = q_host[:32].astype(np.float64)
        t1 = time.perf_counter()
        s = qs @ sub.T
        d = 1.0 - s / (np.linalg.norm(qs, axis=1)[:, None] * np.linalg.norm(sub, axis=1)[None, :])
        ref = np.argsort(d, axis=1, kind="stable")[:, :K]
        spent = time.perf_counter() - t1
        got_r, got_d, _ = pqv.Corpus.upload(corpus_t[:m].cpu().numpy()).brute_topk(q_host[:32], K, pqv.PQV_COSINE)
        refd = np.take_along_axis(d, ref, axis=1)
        result["cpu_baseline"] = {
            "value": 32 / spent * (m / n), "unit": "queries/s", "cores": os.cpu_count(), "kind": "port",
            "sample": f"numpy f64 matmul + argsort, 32 queries x {m} rows in {spent:.2f} s, scaled by {m}/{n} rows",
            "parity": {"queries_checked": 32, "max_rel_dist_err": float(np.max(np.abs(got_d - refd) / np.maximum(np.abs(refd), 1e-3))),
                       "ids_equal_fraction": float((got_r == ref).mean())}}
    sys.stdout.flush()
    os.write(real_stdout, (json.dumps(result) + "\n").encode())


def cpu_baseline(args, index, corpus_t, queries_t, rows_t, dist_t, nprobe, nq):
    """Times the CPU oracle (a port of src/ivf/search.rs:83-142, compiled -O3 -march=native
    -ffp-contract=off on THIS host) on a bounded sample of the same queries, one thread as
    the reference's query loop is (search.rs:115), and checks the GPU results against it."""
    from oracle_binding import Oracle, build_oracle
    build_oracle("native")
    o = Oracle(native=True)
    host = corpus_t.cpu().numpy()
    qs = queries_t.cpu().numpy()
    oidx = o.index_from_bytes(index.to_bytes())
    grows = rows_t.cpu().numpy().view(np.uint32)
    gdist = dist_t.cpu().numpy()
    done, spent = 0, 0.0
    ids_ok, dist_ok, ids_tie_ok, tie_groups = True, True, True, 0
    chunk = 4
    while done < nq and spent < args.cpu_seconds:
        b = min(chunk, nq - done)
        t0 = time.perf_counter()
        orows, odist, onf, _ = oidx.topk_batch(host, qs[done:done + b], K, nprobe)
        spent += time.perf_counter() - t0
        g = grows[done:done + b]
        ids_ok &= bool((orows == g).all())
        dist_ok &= bool((odist.view(np.uint32) == gdist[done:done + b].view(np.uint32)).all())
        # pqv_topk_device orders equal output distances by (d2, position); Rust orders them by heap
        # history (the host API pqv_topk replays that exactly).  Inside a group of equal distance the
        # id SETS must still agree.
        for i in range(b):
            if (orows[i] == g[i]).all():
                continue
            j = 0
            while j < K:
                e = j
                while e + 1 < K and odist[i, e + 1] == odist[i, j]:
                    e += 1
                if e > j:
                    tie_groups += 1
                ids_tie_ok &= sorted(orows[i, j:e + 1].tolist()) == sorted(g[i, j:e + 1].tolist())
                j = e + 1
        done += b
        chunk = min(64, chunk * 2)
    return {"value": done / spent, "unit": "queries/s", "cores": 1, "kind": "port",
            "threads_note": "one thread, as the reference's query loop (search.rs:115)",
            "sample": f"first {done} of the step's {nq} queries, in-memory corpus, oracle -O3 -march=native "
                      f"-ffp-contract=off, {spent:.1f} s",
            "host_cpus": os.cpu_count(),
            "parity": {"queries_checked": done, "row_idx_identical": ids_ok, "dist_bit_identical": dist_ok,
                       "row_idx_identical_up_to_order_inside_equal_distance_groups": ids_tie_ok,
                       "equal_distance_groups_seen": tie_groups}}


if __name__ == "__main__":
    main()
