"""Batch-axis sharding over the GPUs of one node (the only parallelism of the hot path).

Every op of the forward path is independent per image, so rank g of G simply runs images
[g*B/G, (g+1)*B/G) with replicated weights; no data-path collective exists except the optional end-of-forward
all-gather of the (small) outputs, which is an RCCL all-gather over xGMI when the process group backend is "nccl"
(RCCL on ROCm) and a gloo all-gather in the CPU tests.
"""
import torch
import torch.distributed as dist


def shard_bounds(batch, rank, world):
    """[start, stop) of `rank`'s slice of a batch of `batch` images: sizes differ by at most one, in rank order."""
    if not (0 <= rank < world):
        raise ValueError(f"rank {rank} outside world of {world}")
    base, extra = divmod(batch, world)
    start = rank * base + min(rank, extra)
    return start, start + base + (1 if rank < extra else 0)


def shard_batch(x, rank=None, world=None):
    """This rank's contiguous slice of a globally-known batch tensor (axis 0)."""
    rank = dist.get_rank() if rank is None else rank
    world = dist.get_world_size() if world is None else world
    lo, hi = shard_bounds(x.shape[0], rank, world)
    return x[lo:hi]


def gather_batch(y_local, global_batch=None, group=None):
    """All-gather per-rank outputs (axis 0) into the full batch, in rank order, on every rank.

    Equal shards use one all_gather_into_tensor (a single RCCL all-gather: 1 MB per rank for ViT logits);
    ragged shards fall back to all_gather on padded buffers.
    """
    world = dist.get_world_size(group)
    if world == 1:
        return y_local
    y_local = y_local.contiguous()
    n_local = y_local.shape[0]
    if global_batch is None or global_batch % world == 0:
        out = torch.empty((world * n_local,) + tuple(y_local.shape[1:]), dtype=y_local.dtype, device=y_local.device)
        if dist.get_backend(group) == "gloo":
            parts = list(out.chunk(world, dim=0))
            dist.all_gather(parts, y_local, group=group)
        else:
            dist.all_gather_into_tensor(out, y_local, group=group)
        return out
    sizes = [shard_bounds(global_batch, r, world) for r in range(world)]
    width = max(hi - lo for lo, hi in sizes)
    pad = torch.zeros((width,) + tuple(y_local.shape[1:]), dtype=y_local.dtype, device=y_local.device)
    pad[:n_local] = y_local
    parts = [torch.empty_like(pad) for _ in range(world)]
    dist.all_gather(parts, pad, group=group)
    return torch.cat([p[:hi - lo] for p, (lo, hi) in zip(parts, sizes)], dim=0)


def forward_sharded(fn, x_global, group=None, gather=True):
    """Run `fn` (a module or callable, per-image independent) on this rank's slice of x_global; optionally gather."""
    rank, world = dist.get_rank(group), dist.get_world_size(group)
    lo, hi = shard_bounds(x_global.shape[0], rank, world)
    y = fn(x_global[lo:hi])
    return gather_batch(y, x_global.shape[0], group) if gather else y
