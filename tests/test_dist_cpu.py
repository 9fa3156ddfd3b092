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


def _shape_check_worker(rank, world, port, q):
    """RcclComm._check_equal_shards on a SUB-GROUP (ranks 0 and 2 of 3): the host collective must run over exactly the communicator's
    ranks, a ragged call must raise on every member (not deadlock), and the decision to verify must not depend on what an individual
    rank has seen before (a per-rank shape cache turned the ragged case into a hang)."""
    import os
    import torch.distributed as dist
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from mi355attn.dist import RcclComm
        grp = dist.new_group([0, 2])                      # every rank creates it; rank 1 is not a member
        out = []
        if rank in (0, 2):
            c = RcclComm.__new__(RcclComm)                # the check needs no device: build the object without the C communicator
            c.group, c.check, c._ncalls, c._h = grp, "always", 0, None
            c.rank, c.world = dist.get_rank(grp), dist.get_world_size(grp)
            c._check_equal_shards((4, 10))                # equal: passes
            c._check_equal_shards((4, 10))                # seen before on both: still a collective, still passes
            try:                                          # ragged, and rank 0 has "seen" its shape before while rank 2 has not
                c._check_equal_shards((4, 10) if rank == 0 else (3, 10))
                out.append("no error")
            except ValueError:
                out.append("raised")
            c.fix_shape()
            c._check_equal_shards((2, 2))                 # verified (first call after fix_shape)
            c._check_equal_shards((2, 2) if rank == 0 else (9, 9))     # not verified any more: no collective, no error, no hang
            out.append(c._ncalls)
        dist.barrier()
        q.put((rank, out))
    finally:
        dist.destroy_process_group()


def test_rccl_comm_shape_check_is_group_aware_and_collective_safe():
    import sys
    from conftest import PKG
    if PKG not in sys.path:
        sys.path.insert(0, PKG)
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_shape_check_worker, args=(r, 3, port, q)) for r in range(3)]
    for p in procs:
        p.start()
    results = dict(q.get(timeout=120) for _ in range(3))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert results[0] == ["raised", 2] and results[2] == ["raised", 2] and results[1] == []
