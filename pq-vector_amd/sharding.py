"""Row-range sharding of a corpus across ranks and the top-k exchange between them.

The reference handles a multi-file table as independent per-file IVF indexes probed one by
one and merged in a single heap (src/df_vector/index_exec.rs:85-164,
src/df_vector/exec.rs:264-267).  Here a shard = one contiguous row range + its own index on
one GPU; the only exchange is one all-gather of k x {distance, global row id} per query
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


def merge_gathered(gath_dist, gath_rows, k):
    """gath_dist/gath_rows: [world, nq, k] (unused slots: +inf / -1).  Returns ([nq,k],[nq,k]).

    Shard-major concatenation + a stable sort == ordering by (distance, shard, position)."""
    world, nq, kk = gath_dist.shape
    d = gath_dist.permute(1, 0, 2).reshape(nq, world * kk)
    r = gath_rows.permute(1, 0, 2).reshape(nq, world * kk)
    order = torch.sort(d, dim=1, stable=True).indices[:, :k]
    return torch.gather(d, 1, order), torch.gather(r, 1, order)


class ShardExchange:
    """Pre-allocated all-gather buffers for a fixed (nq, k); one collective per tensor."""

    def __init__(self, world, nq, k, device, always_collective=False):
        self.world, self.nq, self.k = world, nq, k
        self.always_collective = always_collective
        self.gath_d = torch.empty((world, nq, k), dtype=torch.float32, device=device)
        self.gath_r = torch.empty((world, nq, k), dtype=torch.int64, device=device)

    def exchange(self, local_dist, local_rows_i64, row_base):
        """local_dist [nq,k] f32, local_rows_i64 [nq,k] shard-local ids (-1 / 0xFFFFFFFF = empty)."""
        empty = (local_rows_i64 < 0) | (local_rows_i64 == 0xFFFFFFFF)
        grow = torch.where(empty, torch.full_like(local_rows_i64, -1), local_rows_i64 + row_base)
        if self.world == 1 and not self.always_collective:
            return local_dist, grow
        # rank-major concatenation along dim 0: the layout both RCCL and gloo accept
        dist.all_gather_into_tensor(self.gath_d.view(self.world * self.nq, self.k), local_dist.contiguous())
        dist.all_gather_into_tensor(self.gath_r.view(self.world * self.nq, self.k), grow.contiguous())
        return merge_gathered(self.gath_d, self.gath_r, self.k)
