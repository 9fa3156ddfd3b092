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


class RcclComm:
    """Communicator of the C ABI (include/mi355attn.h: mi355_comm_init / mi355_allgather_f32 / mi355_comm_destroy): the all-gather
    is issued by libmi355attn on torch's current stream, RCCL over xGMI underneath.  The 128-byte unique id is created by rank 0 and
    handed to the other ranks through the already initialised torch.distributed group (its store is the bootstrap; any backend)."""

    def __init__(self, group=None, device=None, check="always"):
        """`group`: the torch.distributed group whose ranks form the communicator (default: WORLD); it also carries the bootstrap and
        the equal-shard check.  `check`: "always" = every all_gather verifies on the host that all ranks pass the same shape (safe
        default: a mismatch is a ValueError on every rank instead of an ncclAllGather with unequal counts); "first" = only the first
        all_gather of this communicator is verified and the caller guarantees the shape never changes afterwards (what a timed loop
        wants: no host collective inside it); "never".  The decision depends on a call counter that is the same on every rank --
        never on what an individual rank has seen -- so the ranks always enter the host collective together."""
        import ctypes
        from . import _ffi
        if check not in ("always", "first", "never"):
            raise ValueError("RcclComm: check must be 'always', 'first' or 'never'")
        self._ffi = _ffi
        self.group, self.check, self._ncalls = group, check, 0
        if dist.is_available() and dist.is_initialized():
            self.rank, self.world = dist.get_rank(group), dist.get_world_size(group)
        else:
            self.rank, self.world = 0, 1
        self.device = torch.device("cuda", torch.cuda.current_device()) if device is None else device
        buf = ctypes.create_string_buffer(128)
        err = None
        if self.rank == 0:
            try:
                _ffi.check(_ffi.lib().mi355_comm_unique_id(buf, 128), "mi355_comm_unique_id")
            except Exception as e:                         # noqa: BLE001  (re-raised below, on EVERY rank)
                err = e
        if self.world > 1:
            # rank 0 always enters the broadcast -- with None when it has no id to hand out -- so that its failure is an exception on every
            # rank instead of rank 0 raising while the others wait in the broadcast for ever
            box = [None if err is not None else bytes(buf.raw)]
            dist.broadcast_object_list(box, src=dist.get_global_rank(group, 0) if group is not None else 0, group=group)
            if box[0] is None:
                raise err if err is not None else RuntimeError("RcclComm: rank 0 could not create the RCCL unique id (see its log)")
            buf = ctypes.create_string_buffer(box[0], 128)
        elif err is not None:
            raise err
        self._h = ctypes.c_void_p()
        with torch.cuda.device(self.device):
            _ffi.check(_ffi.lib().mi355_comm_init(buf, 128, self.rank, self.world, ctypes.byref(self._h)), "mi355_comm_init")

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()
        return False

    def __del__(self):                                     # the ncclComm_t and the C handle are not garbage-collected by anybody else
        try:
            self.close()
        except Exception:                                  # noqa: BLE001  (interpreter shutdown: the library may be gone already)
            pass

    def _check_equal_shards(self, shape):
        """ncclAllGather needs the same count on every rank: verified through the communicator's torch.distributed group (a host-side
        object all-gather over exactly the ranks of this communicator).  Whether a call verifies is a function of the call counter
        only (see __init__), so a rank that has seen a shape before can never skip a collective another rank enters."""
        n, self._ncalls = self._ncalls, self._ncalls + 1
        if self.world == 1 or self.check == "never" or (self.check == "first" and n > 0):
            return
        if not (dist.is_available() and dist.is_initialized()):
            return
        shapes = [None] * self.world
        dist.all_gather_object(shapes, tuple(shape), group=self.group)
        if any(s != tuple(shape) for s in shapes):
            raise ValueError(f"RcclComm.all_gather: shard shapes differ across ranks: {shapes} (equal shards only; use gather_batch with "
                             "global_batch for ragged batches)")

    def fix_shape(self):
        """From now on the caller gathers ONE shape: the next all_gather is verified, later ones are not (check = "first" with the
        call counter restarted).  Must be called on every rank at the same point, like the collectives themselves."""
        self.check, self._ncalls = "first", 0

    def all_gather(self, y_local):
        """(n, ...) fp32 on every rank -> (world * n, ...) in rank order on every rank (equal shards)."""
        f = self._ffi
        if not self._h:
            raise RuntimeError("RcclComm.all_gather: communicator already closed")
        y_local = f.require_device_f32(y_local, "y_local")
        self._check_equal_shards(tuple(y_local.shape))
        out = torch.empty((self.world * y_local.shape[0],) + tuple(y_local.shape[1:]), dtype=torch.float32, device=y_local.device)
        f.check(f.lib().mi355_allgather_f32(self._h, f.dptr(y_local), f.dptr(out), y_local.numel(), f.stream_ptr(y_local.device)),
                "mi355_allgather_f32")
        return out

    def close(self):
        h, self._h = getattr(self, "_h", None), None
        if h:
            self._ffi.check(self._ffi.lib().mi355_comm_destroy(h), "mi355_comm_destroy")


def gather_batch(y_local, global_batch=None, group=None, comm=None):
    """All-gather per-rank outputs (axis 0) into the full batch, in rank order, on every rank.

    Equal shards use ONE all-gather (1 MB per rank for ViT logits): through the C ABI when an RcclComm is passed
    (mi355_allgather_f32), else torch.distributed's all_gather_into_tensor (RCCL under the "nccl" backend, gloo in the CPU tests);
    ragged shards fall back to all_gather on padded buffers.
    """
    if comm is not None and (global_batch is None or global_batch % comm.world == 0):
        return y_local if comm.world == 1 else comm.all_gather(y_local)
    world = dist.get_world_size(group)
    if world == 1:
        return y_local
    y_local = y_local.contiguous()
    n_local = y_local.shape[0]
    if global_batch is None or global_batch % world == 0:
        out = torch.empty((world * n_local,) + tuple(y_local.shape[1:]), dtype=y_local.dtype, device=y_local.device)
        if dist.get_backend(group) == "gloo" and y_local.is_cuda:
            # host-side process group with device tensors (bench.py --dist-backend gloo: ranks sharing one GPU): staged through the host
            host = y_local.cpu()
            parts = [torch.empty_like(host) for _ in range(world)]
            dist.all_gather(parts, host, group=group)
            out.copy_(torch.cat(parts, dim=0))
        elif dist.get_backend(group) == "gloo":
            parts = list(out.chunk(world, dim=0))
            dist.all_gather(parts, y_local, group=group)
        else:
            dist.all_gather_into_tensor(out, y_local, group=group)
        return out
    sizes = [shard_bounds(global_batch, r, world) for r in range(world)]
    width = max(hi - lo for lo, hi in sizes)
    # a host-side process group (gloo) cannot take device tensors: pad, gather and concatenate on the host, then copy back
    staged = dist.get_backend(group) == "gloo" and y_local.is_cuda
    src = y_local.cpu() if staged else y_local
    pad = torch.zeros((width,) + tuple(src.shape[1:]), dtype=src.dtype, device=src.device)
    pad[:n_local] = src
    parts = [torch.empty_like(pad) for _ in range(world)]
    dist.all_gather(parts, pad, group=group)
    out = torch.cat([p[:hi - lo] for p, (lo, hi) in zip(parts, sizes)], dim=0)
    return out.to(y_local.device) if staged else out


def forward_sharded(fn, x_global, group=None, gather=True):
    """Run `fn` (a module or callable, per-image independent) on this rank's slice of x_global; optionally gather."""
    rank, world = dist.get_rank(group), dist.get_world_size(group)
    lo, hi = shard_bounds(x_global.shape[0], rank, world)
    y = fn(x_global[lo:hi])
    return gather_batch(y, x_global.shape[0], group) if gather else y
