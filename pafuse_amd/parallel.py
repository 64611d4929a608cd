"""Hypothesis-axis sharding across the GPUs of one node (one process per GPU, torch.distributed over RCCL).

DDIM trajectories are independent per hypothesis (nothing in common/diffusionpose.py:192-316 or
common/mixste.py reduces over P), so rank r owns hypotheses [lo, hi) for all clips and all steps, weights are
replicated, and the only exchange is ONE all-gather of the per-rank predictions before the aggregation
protocols (J-Agg & co.) - 8.7 MB per rank at P_local=20, T=10, B=1.  Every rank draws the full-P noise tensor
from the same seed and keeps its slice, so the sharded run reproduces the single-GPU hypotheses.
"""
import torch
import torch.distributed as dist


def shard_range(num_proposals, rank, world):
    """Contiguous, balanced [lo, hi) of the hypothesis axis for `rank` (first P % world ranks get one more)."""
    base, rem = divmod(num_proposals, world)
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


def gather_hypotheses(local, num_proposals=None, group=None):
    """all-gather per-rank predictions [B,T,P_r,F,J,3] along the hypothesis axis -> [B,T,P,F,J,3] on every rank.

    Ragged shards (P % world != 0) are padded to the largest shard for the collective and trimmed after."""
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size(group) == 1:
        return local
    world = dist.get_world_size(group)
    P_r = local.shape[2]
    if num_proposals is None:
        sizes = [torch.zeros(1, dtype=torch.int64, device=local.device) for _ in range(world)]
        dist.all_gather(sizes, torch.tensor([P_r], dtype=torch.int64, device=local.device), group=group)
        counts = [int(s.item()) for s in sizes]
    else:
        counts = [shard_range(num_proposals, r, world)[1] - shard_range(num_proposals, r, world)[0] for r in range(world)]
    pmax = max(counts)
    send = local.transpose(0, 2).contiguous()                      # [P_r, T, B, F, J, 3]: hypothesis-major
    if P_r < pmax:
        send = torch.cat([send, send.new_zeros((pmax - P_r,) + tuple(send.shape[1:]))])
    recv = send.new_empty((world * pmax,) + tuple(send.shape[1:]))   # concatenated along dim 0
    dist.all_gather_into_tensor(recv, send, group=group)
    parts = [recv[r * pmax:r * pmax + counts[r]] for r in range(world)]
    return torch.cat(parts).transpose(0, 2).contiguous()


class ShardedSampler:
    """Wrap a pafuse_amd.D3DP so that ``__call__`` runs this rank's hypotheses and returns the gathered result."""

    def __init__(self, model, group=None):
        self.model, self.group = model, group
        self.rank = dist.get_rank(group) if dist.is_initialized() else 0
        self.world = dist.get_world_size(group) if dist.is_initialized() else 1

    def __call__(self, input_2d, input_3d, input_2d_flip=None):
        m = self.model
        m.proposal_shard = shard_range(m.num_proposals, self.rank, self.world) if self.world > 1 else None
        try:
            local = m(input_2d, input_3d, input_2d_flip=input_2d_flip)
        finally:
            m.proposal_shard = None
        return gather_hypotheses(local, m.num_proposals, self.group)
