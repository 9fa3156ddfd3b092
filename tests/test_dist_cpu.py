"""CPU tests (-m "not gpu"): the N>1 path -- batch sharding + end-of-forward gather -- on gloo, world_size 2 and 3.

The compute callable is the oracle (CPU): what is under test is the host-side shard/gather logic, which is the same
code bench.py and the modules use on RCCL.  Property: gather(shards) == unsharded forward, bit for bit.
"""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

import oracle as O


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, world, port, batch, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from mi355attn.dist import forward_sharded, shard_bounds
        torch.manual_seed(1234)
        w1, w2 = torch.randn(4, 32) * 0.2, torch.randn(32, 4) * 0.2
        torch.manual_seed(4321)
        x = torch.randn(batch, 32, 6, 6)
        full = O.se_forward(x, w1, w2)
        got = forward_sharded(lambda xs: O.se_forward(xs, w1, w2), x)
        lo, hi = shard_bounds(batch, rank, world)
        ok = torch.equal(got, full) and got.shape[0] == batch and (hi - lo) in (batch // world, batch // world + 1)
        q.put((rank, bool(ok)))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world,batch", [(2, 8), (2, 7), (3, 10)])
def test_sharded_forward_equals_unsharded(world, batch):
    import sys
    from conftest import PKG
    if PKG not in sys.path:
        sys.path.insert(0, PKG)
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, batch, q)) for r in range(world)]
    for p in procs:
        p.start()
    results = [q.get(timeout=120) for _ in range(world)]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert sorted(results) == [(r, True) for r in range(world)]


def test_shard_bounds_partition():
    from mi355attn.dist import shard_bounds
    for batch in (1, 7, 8, 256, 2048, 2049):
        for world in (1, 2, 3, 4, 8):
            cuts = [shard_bounds(batch, r, world) for r in range(world)]
            assert cuts[0][0] == 0 and cuts[-1][1] == batch
            assert all(a[1] == b[0] for a, b in zip(cuts, cuts[1:]))
            sizes = [hi - lo for lo, hi in cuts]
            assert max(sizes) - min(sizes) <= 1
