"""gloo tests (CPU) of the N>1 path at world sizes 2 and 8: shard ranges, the single all-gather, ragged shards - the
gathered tensor must EQUAL the unsharded one, element for element (values, not shapes)."""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from pafuse_amd.parallel import gather_hypotheses, rank_census, shard_range


def test_shard_ranges_cover_axis():
    for P in (1, 5, 20, 160, 7):
        for world in (1, 2, 3, 8):
            rs = [shard_range(P, r, world) for r in range(world)]
            assert rs[0][0] == 0 and rs[-1][1] == P
            assert all(rs[i][1] == rs[i + 1][0] for i in range(world - 1))
            assert max(h - l for l, h in rs) - min(h - l for l, h in rs) <= 1


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, world, port, P, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        full = torch.arange(2 * 3 * P * 4 * 5 * 3, dtype=torch.float32).reshape(2, 3, P, 4, 5, 3)   # [B,T,P,F,J,3]
        lo, hi = shard_range(P, rank, world)
        got = gather_hypotheses(full[:, :, lo:hi].contiguous(), P)
        q.put((rank, bool(torch.equal(got, full)), tuple(got.shape)))
        got2 = gather_hypotheses(full[:, :, lo:hi].contiguous())          # sizes discovered by a tiny all-gather
        q.put((rank, bool(torch.equal(got2, full)), tuple(got2.shape)))
        census = rank_census(hi - lo)
        want = {"ranks_seen": world, "P_local": [shard_range(P, r, world)[1] - shard_range(P, r, world)[0] for r in range(world)]}
        q.put((rank, census == want, tuple(got.shape)))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("P", [4, 5])
def test_all_gather_world2(P):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, P, q)) for r in range(2)]
    [p.start() for p in procs]
    res = [q.get(timeout=120) for _ in range(6)]
    [p.join(timeout=60) for p in procs]
    assert all(ok for _, ok, _ in res), res
    assert all(shape == (2, 3, P, 4, 5, 3) for _, _, shape in res)


@pytest.mark.parametrize("P", [16, 21])
def test_all_gather_world8(P):
    """the node's own world size (8 ranks): even (P=16: 2 per rank) and ragged (P=21: five ranks carry 3) shards"""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 8, port, P, q)) for r in range(8)]
    [p.start() for p in procs]
    res = [q.get(timeout=300) for _ in range(24)]
    [p.join(timeout=60) for p in procs]
    assert all(ok for _, ok, _ in res), res
    assert all(shape == (2, 3, P, 4, 5, 3) for _, _, shape in res)
