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


def shard_counts(num_proposals, world):
    return [shard_range(num_proposals, r, world)[1] - shard_range(num_proposals, r, world)[0] for r in range(world)]


def gather_hypotheses(local, num_proposals=None, group=None):
    """all-gather per-rank predictions [B,T,P_r,F,J,3] along the hypothesis axis -> [B,T,P,F,J,3] on every rank.

    ONE collective (RCCL all-gather into a [world, B, T, P_max, F, J, 3] buffer, rank-major as the collective lays it
    out) and ONE pass over the result (the permute that moves the rank axis next to the hypothesis axis).  Ragged shards
    (P % world != 0) are padded to the largest shard for the collective and dropped in that same pass."""
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size(group) == 1:
        return local
    if local.is_cuda and dist.get_backend(group) == "gloo":          # rehearsal on a box with fewer GPUs than ranks
        return gather_hypotheses(local.cpu(), num_proposals, group).to(local.device)
    world = dist.get_world_size(group)
    B, T, P_r = local.shape[:3]
    tail = tuple(local.shape[3:])
    if num_proposals is None:
        sizes = [torch.zeros(1, dtype=torch.int64, device=local.device) for _ in range(world)]
        dist.all_gather(sizes, torch.tensor([P_r], dtype=torch.int64, device=local.device), group=group)
        counts = [int(s.item()) for s in sizes]
    else:
        counts = shard_counts(num_proposals, world)
    pmax = max(counts)
    send = local.contiguous()
    if P_r < pmax:
        send = torch.cat([send, send.new_zeros((B, T, pmax - P_r) + tail)], dim=2)
    recv = send.new_empty((world * B, T, pmax) + tail)               # concatenation along dim 0 = rank-major
    dist.all_gather_into_tensor(recv, send, group=group)
    recv = recv.view((world, B, T, pmax) + tail)
    out = recv.permute(1, 2, 0, 3, *range(4, recv.dim()))            # [B, T, world, P_max, ...] view
    if all(c == pmax for c in counts):
        return out.reshape((B, T, world * pmax) + tail)              # the one copy
    keep = torch.tensor([r * pmax + i for r in range(world) for i in range(counts[r])], device=local.device)
    return out.reshape((B, T, world * pmax) + tail).index_select(2, keep)


def rank_census(p_local, group=None):
    """What the collective layer actually sees: every rank contributes (rank, local hypothesis count) to one all-gather.
    Returns {"ranks_seen": n, "P_local": [count of rank 0, 1, ...]} on every rank (bench.py prints it, so a run whose
    ranks did not all join the job cannot report an N-GPU number)."""
    if not (dist.is_available() and dist.is_initialized()):
        return {"ranks_seen": 1, "P_local": [int(p_local)]}
    world = dist.get_world_size(group)
    dev = torch.device("cuda", torch.cuda.current_device()) if dist.get_backend(group) == "nccl" else torch.device("cpu")
    mine = torch.tensor([dist.get_rank(group), int(p_local)], dtype=torch.int64, device=dev)
    got = torch.empty(world * 2, dtype=torch.int64, device=dev)
    dist.all_gather_into_tensor(got, mine, group=group)
    got = got.view(world, 2).cpu()
    ranks = sorted(set(int(r) for r in got[:, 0]))
    by_rank = {int(r): int(c) for r, c in got.tolist()}
    return {"ranks_seen": len(ranks), "P_local": [by_rank[r] for r in ranks]}


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
