"""Row-range sharding of a corpus across ranks and the top-k exchange between them.

The reference handles a multi-file table as independent per-file IVF indexes probed one by
one and merged in a single heap (src/df_vector/index_exec.rs:85-164,
src/df_vector/exec.rs:264-267).  Here a shard = one contiguous row range + its own index on
one GPU; the only exchange is ONE all-gather of k packed {distance, row id} pairs per query
(RCCL when the tensors live on GPUs, gloo on CPU), followed by the same deterministic merge
on every rank: ascending by (distance, shard, position in the shard's list).

torch is plumbing here (collectives + tensor ops); nothing in this module computes
distances.
"""
import torch
import torch.distributed as dist


def shard_range(rank, world, n_rows):
    """Contiguous row range [lo, hi) of `rank`; ranges tile [0, n_rows) exactly."""
    return rank * n_rows // world, (rank + 1) * n_rows // world


def shard_row_groups(rank, world, rg_rows):
    """Shard `rank` of ONE Parquet file that `world` GPUs share: a contiguous range of ROW GROUPS.

    `rg_rows`: the rows of every row group in file order (or a pyarrow FileMetaData / a path).  Returns
    (rg_lo, rg_hi, row_base, n_rows): the half-open row-group range, the file-global row id of the shard's row 0 -- the prefix
    sum of the row groups before it, exactly how the reference maps file-global row ids to row groups
    (src/df_vector/access.rs:128-144) -- and the shard's row count.  Cut r (between ranks r - 1 and r) is the row-group
    boundary whose prefix sum is nearest to r * n / world (the lower one on a tie): a row group is never split (a data page never
    crosses a column chunk, so a shard is a set of whole column chunks), every rank computes every cut from the footer alone, the
    ranges tile the file, and with fewer row groups than ranks the surplus ranks get an EMPTY range (n_rows == 0: they answer
    with empty lists).  parquet_io.load_embedding_column(path, column, device, row_groups=(rg_lo, rg_hi)) loads the range."""
    import ctypes as C
    import numpy as np
    from . import _ffi
    rows = np.ascontiguousarray(_rg_rows(rg_rows), dtype=np.uint64)
    lo, hi, base, n = C.c_uint32(0), C.c_uint32(0), C.c_uint64(0), C.c_uint64(0)
    # ONE implementation of the rule, behind the C ABI (pqv_shard_row_groups, host only): a Rust host cuts the same way
    _rc(_ffi.lib().pqv_shard_row_groups(rows.ctypes.data_as(_ffi.u64p), len(rows), rank, world, C.byref(lo), C.byref(hi), C.byref(base), C.byref(n)))
    return lo.value, hi.value, base.value, n.value


def _shard_row_groups_py(rank, world, rg_rows):
    """The same rule in plain Python (tests cross-check the library against it): exact integer comparison |pre[b] world - r n|."""
    rows = _rg_rows(rg_rows)
    pre = [0]
    for r in rows:
        pre.append(pre[-1] + int(r))
    n = pre[-1]
    cuts = [0]
    for r in range(1, world):
        best = min(range(cuts[-1], len(pre)), key=lambda b: (abs(pre[b] * world - r * n), b))
        cuts.append(best)
    cuts.append(len(rows))
    lo, hi = cuts[rank], cuts[rank + 1]
    return lo, hi, pre[lo], pre[hi] - pre[lo]


def _rg_rows(src):
    if isinstance(src, (list, tuple)):
        return [int(x) for x in src]
    if hasattr(src, "num_row_groups"):
        meta = src
    else:
        import pyarrow.parquet as pq
        meta = pq.ParquetFile(src).metadata
    return [meta.row_group(i).num_rows for i in range(meta.num_row_groups)]


def load_parquet_shard(path, column, rank, world, device=0, readers=None, stats=None):
    """This rank's row-group range of `path` -> a resident corpus.  Returns (corpus or None for an empty range, row_base,
    n_rows, (rg_lo, rg_hi))."""
    from . import parquet_io
    lo, hi, base, n = shard_row_groups(rank, world, path)
    if n == 0:
        return None, base, 0, (lo, hi)
    corpus = parquet_io.load_embedding_column(path, column, device, readers=readers, stats=stats, row_groups=(lo, hi))
    return corpus, base, n, (lo, hi)


def check_shard_dims(dim, group=None):
    """Every shard discovers the list length from ITS row-group range; the reference reads the whole column and rejects a file
    whose vectors differ in length ("Embedding vectors have inconsistent dimensions", src/ivf/parquet.rs:231-280).  All ranks
    compare (`dim` = this rank's, 0 / None for an empty range): raises the reference's message on every rank on a mismatch."""
    from .api import PqvError
    from . import _ffi
    mine = int(dim or 0)
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size(group) == 1:
        return mine
    dims = [None] * dist.get_world_size(group)
    dist.all_gather_object(dims, mine, group=group)
    seen = sorted({int(d) for d in dims if d})
    if len(seen) > 1:
        raise PqvError(_ffi.PQV_ERR_INVALID, "Embedding vectors have inconsistent dimensions")
    return seen[0] if seen else 0


def merge_gathered(gath_dist, gath_rows, k):
    """gath_dist/gath_rows: [world, nq, k] (unused slots: +inf / -1).  Returns ([nq,k],[nq,k]).

    Shard-major concatenation + a stable sort == ordering by (distance, shard, position)."""
    world, nq, kk = gath_dist.shape
    d = gath_dist.permute(1, 0, 2).reshape(nq, world * kk)
    r = gath_rows.permute(1, 0, 2).reshape(nq, world * kk)
    order = torch.sort(d, dim=1, stable=True).indices[:, :k]
    return torch.gather(d, 1, order), torch.gather(r, 1, order)


class ShardExchange:
    """Pre-allocated all-gather buffers for a fixed (nq, k); ONE collective per step: distance and row of
    every result travel as one packed element.

    On GPUs the merge is one library kernel (pqv_merge_topk_packed_device, ordered by (distance, shard,
    position)); on CPU tensors (gloo tests) the same order comes from a stable torch sort."""

    def __init__(self, world, nq, k, device, always_collective=False, row_bases=None):
        self.world, self.nq, self.k = world, nq, k
        self.always_collective = always_collective
        self.device = torch.device(device)
        # generic path: {f32 distance bits, i64 global row} as two int64 words per result
        self.gath = torch.empty((world, nq, k, 2), dtype=torch.int64, device=device)
        self.fast = self.device.type == "cuda" and row_bases is not None
        if self.fast:
            # GPU path: {f32 distance, u32 shard-local row} as two int32 words per result (8 bytes)
            self.gath32 = torch.empty((world, nq, k, 2), dtype=torch.int32, device=device)
            self.bases = torch.tensor(list(row_bases), dtype=torch.int64, device=device)
            self.out_d = torch.empty((nq, k), dtype=torch.float32, device=device)
            self.out_r = torch.empty((nq, k), dtype=torch.int64, device=device)

    def exchange_u32(self, local_dist, local_rows_i32):
        """GPU fast path: local_rows_i32 [nq,k] is the searcher's raw u32 output viewed as int32
        (0xFFFFFFFF = empty).  One pack, ONE all-gather, one merge kernel on the current stream."""
        from . import _ffi
        packed = torch.stack((local_dist.view(torch.int32), local_rows_i32), dim=-1)      # [nq, k, 2]
        if self.world == 1 and not self.always_collective:
            self.gath32[0].copy_(packed)
        else:
            dist.all_gather_into_tensor(self.gath32.view(self.world * self.nq, self.k * 2), packed.view(self.nq, self.k * 2))
        rc = _ffi.lib().pqv_merge_topk_packed_device(
            self.device.index or 0, _ffi.vp(self.gath32.data_ptr()), _ffi.vp(self.bases.data_ptr()), self.world, self.nq,
            self.k, _ffi.vp(self.out_d.data_ptr()), _ffi.vp(self.out_r.data_ptr()),
            _ffi.vp(torch.cuda.current_stream().cuda_stream))
        if rc != 0:
            raise RuntimeError(_ffi.lib().pqv_last_error().decode())
        return self.out_d, self.out_r

    def exchange(self, local_dist, local_rows_i64, row_base):
        """local_dist [nq,k] f32, local_rows_i64 [nq,k] shard-local ids (-1 / 0xFFFFFFFF = empty)."""
        empty = (local_rows_i64 < 0) | (local_rows_i64 == 0xFFFFFFFF)
        grow = torch.where(empty, torch.full_like(local_rows_i64, -1), local_rows_i64 + row_base)
        if self.world == 1 and not self.always_collective:
            return local_dist, grow
        packed = torch.stack((local_dist.contiguous().view(torch.int32).to(torch.int64), grow), dim=-1)   # [nq, k, 2]
        # rank-major concatenation along dim 0: the layout both RCCL and gloo accept
        dist.all_gather_into_tensor(self.gath.view(self.world * self.nq, self.k * 2), packed.view(self.nq, self.k * 2))
        gd = self.gath[..., 0].to(torch.int32).view(torch.float32)
        return merge_gathered(gd, self.gath[..., 1], self.k)


class RcclShardComm:
    """The exchange behind the C ABI (pqv_shard_*, include/pqv.h): RCCL bound by the library itself, so the same calls
    serve a Rust host.  torch only carries the 128-byte rendezvous id from rank 0 to the others here (any channel
    would do); the all-gather and the merge run inside libpqv_hip.so on the caller's stream."""

    def __init__(self, rank, world, device_index, id_bytes=None):
        import ctypes as C
        from . import _ffi
        L = _ffi.lib()
        if id_bytes is None:
            buf = (C.c_uint8 * 128)()
            if rank == 0:
                _rc(L.pqv_shard_unique_id(buf))
            t = torch.tensor(list(buf), dtype=torch.uint8)
            if world > 1:
                # the process group may be RCCL-only (device tensors) or gloo (host tensors)
                if dist.get_backend() == "nccl":
                    t = t.to(torch.device("cuda", device_index))
                dist.broadcast(t, src=0)
            id_bytes = bytes(t.cpu().tolist())
        idb = (C.c_uint8 * 128).from_buffer_copy(id_bytes)
        h = C.c_void_p()
        _rc(L.pqv_shard_comm_create(device_index, rank, world, idb, C.byref(h)))
        self._h, self.rank, self.world, self.device_index = h, rank, world, device_index

    @classmethod
    def create_collective(cls, rank, world, device_index):
        """Every rank of the torch process group calls this together.  Returns (comm or None, reason).

        A rank that cannot even reach ncclCommInitRank (librccl not resolvable, bad device index) would leave the others
        blocked inside it, so a NON-collective pre-flight runs first on every rank (the library resolves RCCL and selects
        the device) and its result is all-reduced: the collective initialisation starts only if every rank passed.
        Failures before and after the initialisation take the same number of torch collectives on every rank; a failure
        INSIDE ncclCommInitRank itself (a rank dying mid-rendezvous) is RCCL's to time out."""
        import ctypes as C
        from . import _ffi
        on_gpu = world > 1 and dist.get_backend() == "nccl"
        # the torch collectives of this hand-shake run on the device the PROCESS GROUP already uses (the current device), never on
        # `device_index`: a rank whose index is out of range must still be able to join them and report 0
        dev = torch.device("cuda", torch.cuda.current_device()) if on_gpu else torch.device("cpu")
        pre_ok, pre_why = 1, ""
        try:
            if not _ffi.lib().pqv_shard_rccl_path():
                pre_ok, pre_why = 0, "librccl could not be resolved on this rank: " + _ffi.lib().pqv_last_error().decode()
            elif not (0 <= device_index < max(1, _ffi.lib().pqv_device_count())):
                pre_ok, pre_why = 0, f"device index {device_index} out of range on this rank"
        except Exception as e:
            pre_ok, pre_why = 0, str(e)
        if world > 1:
            okt = torch.tensor([pre_ok], dtype=torch.int32, device=dev)
            dist.all_reduce(okt, op=dist.ReduceOp.MIN)
            if int(okt.item()) != 1:
                return None, pre_why or "the RCCL pre-flight failed on another rank"
        elif not pre_ok:
            return None, pre_why
        msg = torch.zeros(129, dtype=torch.uint8)
        reason = ""
        if rank == 0:
            buf = (C.c_uint8 * 128)()
            if _ffi.lib().pqv_shard_unique_id(buf) == 0:
                msg[0] = 1
                msg[1:] = torch.tensor(list(buf), dtype=torch.uint8)
            else:
                reason = _ffi.lib().pqv_last_error().decode()
        if world > 1:
            t = msg.to(dev) if on_gpu else msg
            dist.broadcast(t, src=0)
            msg = t.cpu()
        if int(msg[0]) != 1:
            return None, reason or "rank 0 could not draw a rendezvous id"
        comm = None
        try:
            comm = cls(rank, world, device_index, id_bytes=bytes(msg[1:].tolist()))
        except Exception as e:
            reason = str(e)
        ok = torch.tensor([1 if comm is not None else 0], dtype=torch.int32, device=dev)
        if world > 1:
            dist.all_reduce(ok, op=dist.ReduceOp.MIN)
        if int(ok.item()) != 1:
            if comm is not None:
                comm.close()
            return None, reason or "communicator creation failed on another rank"
        return comm, ""

    def exchange(self, local_dist, local_rows_i32, row_bases_i64, out_d, out_r, stream=None):
        """local_dist f32 / local_rows_i32 [nq, k] device tensors (raw searcher outputs), row_bases_i64 [world];
        writes out_d f32 / out_r i64 [nq, k] on `stream` (default: torch's current stream)."""
        from . import _ffi
        nq, k = local_dist.shape
        # raw pointers cross the ABI below: a sliced view, a wrong dtype or a host tensor would be read as garbage, silently
        for name, t, dt, shape in (("local_dist", local_dist, torch.float32, (nq, k)), ("local_rows_i32", local_rows_i32, torch.int32, (nq, k)),
                                   ("row_bases_i64", row_bases_i64, torch.int64, (self.world,)), ("out_d", out_d, torch.float32, (nq, k)),
                                   ("out_r", out_r, torch.int64, (nq, k))):
            if not (t.is_cuda and t.device.index == self.device_index and t.is_contiguous() and t.dtype == dt and tuple(t.shape) == shape):
                raise ValueError(f"{name}: expected a contiguous {dt} tensor of shape {shape} on cuda:{self.device_index}, got "
                                 f"{t.dtype} {tuple(t.shape)} on {t.device}{'' if t.is_contiguous() else ' (non-contiguous)'}")
        st = torch.cuda.current_stream().cuda_stream if stream is None else stream
        _rc(_ffi.lib().pqv_shard_exchange(self._h, _ffi.vp(local_dist.data_ptr()), _ffi.vp(local_rows_i32.data_ptr()),
                                          _ffi.vp(row_bases_i64.data_ptr()), nq, k, _ffi.vp(out_d.data_ptr()),
                                          _ffi.vp(out_r.data_ptr()), _ffi.vp(st)))
        return out_d, out_r

    def close(self):
        if self._h:
            from . import _ffi
            _ffi.lib().pqv_shard_comm_free(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def _rc(rc):
    if rc != 0:
        from . import _ffi
        raise RuntimeError(_ffi.lib().pqv_last_error().decode())
