"""GPU parity at BASELINE.json's FULL configuration sizes, through the C ABI, against the CPU oracle.

  C3         10 M x 768, n_clusters 1024, nprobe 32, k 10, one 1024-query batch -- uniform data (the reference's bench
             recipe) AND the Gaussian mixture (SURVEY 8d's secondary set; the per-list residual images, clamped pair images
             and centre-distance pruning only run there): >= 256 queries of the batch oracle-checked bit for bit, 16 of
             them again as one-query calls, the host API (exact under ties) on 64;
  C4 shard   12.5 M x 768, one rank's share of configs[3]: 64 queries;
  C5         10 M x 1536 cosine, 1024-query batch: 32 queries against an f64 brute force of the same resident corpus.

The corpus is generated on the device (bench.py's generators: the data the bench line is measured on), searched there,
downloaded ONCE, and the oracle (index parsed from the GPU-built blob) answers one query per host thread.  10 M-row
lists are 9 766 rows long: the 512-row threshold sample, the wide-quad instance and the chunk-major item tables only
take their production shapes at this size."""
import gc
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _bits(a):
    return np.ascontiguousarray(a, dtype=np.float32).view(np.uint32)


def _ivf_full(pqv, name, data_kind, n_check, n_single, n_host):
    import torch
    import bench
    n, dim, kc, nprobe, nq = bench.WORKLOADS[name]
    k = 10
    dev = torch.device("cuda", 0)
    if data_kind == "mixture":
        corpus_t = bench.synth_mixture(torch, dev, 1234, n, dim, kc)
        q_t = bench.synth_mixture(torch, dev, 7, nq, dim, kc)
    else:
        corpus_t = bench.synth(torch, dev, 1234, n, dim)
        q_t = bench.synth(torch, dev, 7, nq, dim)
    torch.cuda.synchronize()
    corpus = pqv.Corpus.from_device_ptr(corpus_t.data_ptr(), n, dim, device=0, keepalive=corpus_t)
    index = pqv.IndexBuilder(corpus).n_clusters(kc).max_iters(20).seed(42).workers(os.cpu_count() or 1).build()
    off, lrows = index.list_offsets, index.list_rows
    assert index.n_clusters == kc and int(off[-1]) == n
    seen = np.zeros(n, dtype=bool)
    seen[lrows] = True
    assert seen.all()                                        # every row in exactly one list (n entries, all distinct)
    del seen
    s = pqv.Searcher(index, corpus)
    plan = s.describe(nq, k, nprobe)
    assert "wide_filter_kernel" in plan and "int8 screen operands" in plan, plan
    if data_kind == "mixture":
        assert "per-list residual" in plan or "residual" in plan, plan
    # the asynchronous device path on the whole batch: what bench.py times
    r_t = torch.empty((nq, k), dtype=torch.int32, device=dev)
    d_t = torch.empty((nq, k), dtype=torch.float32, device=dev)
    nf_t = torch.empty((nq,), dtype=torch.int32, device=dev)
    nc_t = torch.empty((nq,), dtype=torch.int64, device=dev)
    flags_t = torch.zeros((nq,), dtype=torch.int32, device=dev)
    st = torch.cuda.current_stream().cuda_stream
    s.topk_device(q_t.data_ptr(), nq, k, nprobe, r_t.data_ptr(), d_t.data_ptr(), nf_t.data_ptr(), nc_t.data_ptr(), stream=st,
                  d_tie_flags=flags_t.data_ptr())
    torch.cuda.synchronize()
    grows, gdist = r_t.cpu().numpy().view(np.uint32), d_t.cpu().numpy()
    gnc = nc_t.cpu().numpy()
    assert (nf_t.cpu().numpy() == k).all()
    assert (np.diff(gdist.astype(np.float64), axis=1) >= 0).all()
    # 24-bit uniform data can tie exactly; a flagged query's answer depends on the reference's heap history: the host API
    # replays it, and every flagged query joins the oracle-checked set below
    flagged = np.nonzero(flags_t.cpu().numpy())[0].tolist()
    assert len(flagged) <= nq // 50, len(flagged)
    # a second submission (running thresholds, atomics, two instances in flight) returns the same bits
    r2 = torch.empty_like(r_t); d2 = torch.empty_like(d_t)
    s.topk_device(q_t.data_ptr(), nq, k, nprobe, r2.data_ptr(), d2.data_ptr(), stream=st)
    torch.cuda.synchronize()
    assert torch.equal(r2, r_t) and torch.equal(d2.view(torch.int32), d_t.view(torch.int32))
    # one-query calls (probe_single_kernel + the seed tail: different launches) agree with the batch
    qs = q_t.cpu().numpy()
    sel = sorted(set(bench.spread(nq, n_check)) | set(flagged))
    r1 = torch.empty((1, k), dtype=torch.int32, device=dev); d1 = torch.empty((1, k), dtype=torch.float32, device=dev)
    nc1 = torch.empty((1,), dtype=torch.int64, device=dev)
    for q in [q for q in sel if q not in set(flagged)][::max(1, len(sel) // n_single)][:n_single]:
        s.topk_device(q_t[q:q + 1].data_ptr(), 1, k, nprobe, r1.data_ptr(), d1.data_ptr(), 0, nc1.data_ptr(), stream=st)
        torch.cuda.synchronize()
        assert np.array_equal(r1.cpu().numpy().view(np.uint32)[0], grows[q]), q
        assert np.array_equal(_bits(d1.cpu().numpy())[0], _bits(gdist[q])), q
        assert int(nc1.item()) == int(gnc[q])
    # the host API (exact under ties: replays the reference heap where distances tie)
    hs = [q for q in sel if q not in set(flagged)][:n_host]
    hr, hd, hnf, hnc = s.topk(qs[hs], k, nprobe)
    assert np.array_equal(hr, grows[hs]) and np.array_equal(_bits(hd), _bits(gdist[hs])) and np.array_equal(hnc.astype(np.int64), gnc[hs])
    # the oracle: one query per host thread on a host copy of the searched matrix
    host = corpus_t.cpu().numpy()
    par = bench.OracleParity(index.to_bytes(), host, native=False)
    par.compare(qs, sel, grows, gdist, k, nprobe, searcher=s, gncand=gnc)
    rec = par.record()
    del host, par
    s.close(); corpus.close()
    del corpus_t
    gc.collect()
    torch.cuda.empty_cache()
    assert rec["queries_checked"] == len(sel) and len(sel) >= min(n_check, nq)
    # distances bit-identical everywhere; ids identical, or (flagged queries only) identical once replayed through pqv_topk
    assert rec["dist_bit_identical"] and rec["n_candidates_identical"] and rec["row_idx_identical_after_replay"], rec
    assert rec["row_idx_identical"] or (flagged and rec["queries_replayed_through_pqv_topk"] <= len(flagged)), rec
    return rec


@pytest.mark.timeout(1500)
def test_c3_full_size_uniform_against_oracle(pqv):
    _ivf_full(pqv, "c3", "uniform", n_check=256, n_single=16, n_host=64)


@pytest.mark.timeout(1500)
def test_c3_full_size_mixture_against_oracle(pqv):
    _ivf_full(pqv, "c3", "mixture", n_check=256, n_single=16, n_host=64)


@pytest.mark.timeout(1500)
def test_c4_shard_full_size_against_oracle(pqv):
    _ivf_full(pqv, "c4", "uniform", n_check=64, n_single=4, n_host=16)


@pytest.mark.timeout(1500)
def test_c5_full_size_cosine_against_f64_brute_force(pqv):
    import torch
    import bench
    n, dim, _, _, nq = bench.WORKLOADS["c5"]
    k = 10
    dev = torch.device("cuda", 0)
    corpus_t = bench.synth(torch, dev, 1234, n, dim)
    corpus_t -= 0.5                                           # ada-002-like: signed components
    q_t = bench.synth(torch, dev, 7, nq, dim) - 0.5
    torch.cuda.synchronize()
    corpus = pqv.Corpus.from_device_ptr(corpus_t.data_ptr(), n, dim, device=0, keepalive=corpus_t)
    rows, dist, nf = corpus.brute_topk(q_t.cpu().numpy(), k, pqv.PQV_COSINE)
    assert (nf == k).all() and (np.diff(dist.astype(np.float64), axis=1) >= -1e-7).all()
    for q in range(nq):
        assert len(set(rows[q].tolist())) == k
    rec = bench.brute_f64_check(torch, corpus_t, q_t, rows, dist, bench.spread(nq, 32), k)
    corpus.close()
    del corpus_t
    gc.collect()
    torch.cuda.empty_cache()
    assert rec["ok"] and rec["queries_checked"] == 32, rec
    # the bench's own data (one-sided uniform [0, 1) rows: every cosine within 0.02 of each other) as well
    corpus_t = bench.synth(torch, dev, 1234, 2_000_000, dim)
    q_t = bench.synth(torch, dev, 7, nq, dim)
    corpus = pqv.Corpus.from_device_ptr(corpus_t.data_ptr(), 2_000_000, dim, device=0, keepalive=corpus_t)
    rows, dist, nf = corpus.brute_topk(q_t.cpu().numpy(), k, pqv.PQV_COSINE)
    rec = bench.brute_f64_check(torch, corpus_t, q_t, rows, dist, bench.spread(nq, 32), k)
    corpus.close()
    assert rec["ok"], rec
