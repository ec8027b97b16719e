"""One-process-per-GPU helpers on top of torch.distributed (backend "nccl" = RCCL over xGMI on ROCm, "gloo" for the
CPU tests).  The prediction path shards the slices of every direction across ranks and needs exactly ONE exchange:
an elementwise max all-reduce of the packed (prob, direction, label) keys (or a sum of the one-hot votes).  Training
shards the minibatch and all-reduces the flat gradient buffer (engine.VolSegUnet._allreduce_grads)."""
from __future__ import annotations

import os

import torch
import torch.distributed as dist


def world() -> tuple[int, int]:
    """(rank, world_size); (0, 1) when no process group is initialised."""
    if dist.is_available() and dist.is_initialized():
        return dist.get_rank(), dist.get_world_size()
    return 0, 1


def init_from_env(backend: str | None = None) -> tuple[int, int, int]:
    """Initialise the default process group from torchrun's environment; returns (rank, world, local_rank)."""
    ws = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if ws > 1 and not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        if backend is None:
            backend = "nccl" if torch.cuda.is_available() else "gloo"
        kw = {}
        if os.environ.get("VOLSEG_DIST_TIMEOUT_S"):   # rendezvous / collective timeout (tests: fail instead of waiting for ever)
            import datetime
            kw["timeout"] = datetime.timedelta(seconds=float(os.environ["VOLSEG_DIST_TIMEOUT_S"]))
        if backend == "nccl":
            torch.cuda.set_device(local)
            dist.init_process_group(backend, device_id=torch.device("cuda", local), **kw)
        else:
            dist.init_process_group(backend, **kw)
    return rank, ws, local


def shard_range(n: int, rank: int, world_size: int) -> tuple[int, int]:
    """Contiguous [lo, hi) share of n items for this rank; shares differ by at most one item."""
    base, rem = divmod(n, world_size)
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


class VsComm:
    """This rank's RCCL communicator behind the C ABI (include/volseg_hip.h: vs_comm_*): the transport for callers without
    torch.distributed, and - with VOLSEG_COMM=rccl in the environment - the one the key merge and the plain gradient
    all-reduce of this package use.  Rank 0's unique id reaches the other ranks over the default process group (any
    backend); without a process group the communicator has one rank."""

    def __init__(self, device: torch.device | None = None):
        import ctypes
        from . import _lib
        self._lib = _lib
        rank, ws = world()
        if device is not None:
            torch.cuda.set_device(device)
        ident = ctypes.create_string_buffer(128)
        if rank == 0:
            _lib.check(_lib.lib.vs_comm_unique_id(ident))
        if ws > 1:
            box = [bytes(ident.raw)]
            dist.broadcast_object_list(box, src=0)
            ident = ctypes.create_string_buffer(box[0], 128)
        self.handle = ctypes.c_void_p()
        _lib.check(_lib.lib.vs_comm_init(ctypes.byref(self.handle), ws, rank, ident))
        self.rank, self.size = rank, ws

    def allreduce_sum_(self, t: torch.Tensor) -> None:
        assert t.dtype == torch.float32 and t.is_contiguous() and t.is_cuda
        self._lib.check(self._lib.lib.vs_comm_allreduce_sum_f32(self.handle, self._lib.ptr(t), t.numel(), self._lib.stream_ptr()))

    def allreduce_max_keys_(self, keys: torch.Tensor) -> None:
        assert keys.element_size() == 4 and keys.is_contiguous() and keys.is_cuda
        self._lib.check(self._lib.lib.vs_comm_allreduce_max_u32(self.handle, self._lib.ptr(keys), keys.numel(), self._lib.stream_ptr()))

    def reduce_scatter_max_keys(self, keys: torch.Tensor, out: torch.Tensor) -> None:
        assert keys.numel() == out.numel() * self.size and keys.is_contiguous() and out.is_contiguous()
        self._lib.check(self._lib.lib.vs_comm_reduce_scatter_max_u32(self.handle, self._lib.ptr(keys), self._lib.ptr(out), out.numel(),
                                                                    self._lib.stream_ptr()))

    def allgather(self, part: torch.Tensor, out: torch.Tensor) -> None:
        nbytes = part.numel() * part.element_size()
        assert out.numel() * out.element_size() == nbytes * self.size and part.is_contiguous() and out.is_contiguous()
        self._lib.check(self._lib.lib.vs_comm_allgather(self.handle, self._lib.ptr(part), self._lib.ptr(out), nbytes, self._lib.stream_ptr()))

    def broadcast_(self, t: torch.Tensor, root: int = 0) -> None:
        assert t.is_contiguous() and t.is_cuda
        self._lib.check(self._lib.lib.vs_comm_broadcast(self.handle, self._lib.ptr(t), t.numel() * t.element_size(), root,
                                                       self._lib.stream_ptr()))

    def close(self) -> None:
        if self.handle:
            self._lib.lib.vs_comm_destroy(self.handle)
            self.handle = None


_vs_comm: VsComm | None = None


def vs_comm(device: torch.device | None = None) -> VsComm | None:
    """The process's C-ABI communicator when VOLSEG_COMM=rccl selects that transport (created on first use), else None."""
    global _vs_comm
    if os.environ.get("VOLSEG_COMM", "") != "rccl":
        return None
    if _vs_comm is None:
        _vs_comm = VsComm(device)
    return _vs_comm


def allreduce_max_keys(keys: torch.Tensor, group=None) -> None:
    """In-place elementwise max of uint32 packed keys across ranks.  Keys are < 2**31 (fp16 bits of a probability
    <= 1.0 are <= 0x3C00), so they are reduced as int32 - the order is the same."""
    if world()[1] == 1:
        return
    comm = vs_comm(keys.device) if keys.is_cuda and group is None else None
    if comm is not None:
        comm.allreduce_max_keys_(keys)
        return
    dist.all_reduce(keys.view(torch.int32), op=dist.ReduceOp.MAX, group=group)


def allreduce_sum_votes(votes: torch.Tensor, group=None) -> None:
    if world()[1] == 1:
        return
    dist.all_reduce(votes, op=dist.ReduceOp.SUM, group=group)


def _scalar_device(group=None) -> torch.device:
    """Where a tensor must live to be reduced by the group's backend (RCCL reduces device memory only)."""
    if dist.get_backend(group) == "nccl":
        return torch.device("cuda", torch.cuda.current_device())
    return torch.device("cpu")


def allreduce_sums(values, group=None) -> list[float]:
    """Elementwise SUM over ranks of a few host scalars (float64), returned to the host: the trainer's loss / metric
    bookkeeping.  Every rank receives the same numbers, so decisions taken on them (learning rate from the LR finder, early
    stopping) are the same on every rank by construction.  Identity for a single process."""
    vals = [float(v) for v in values]
    if world()[1] == 1:
        return vals
    t = torch.tensor(vals, dtype=torch.float64, device=_scalar_device(group))
    dist.all_reduce(t, op=dist.ReduceOp.SUM, group=group)
    return [float(v) for v in t.cpu()]


def mean_scalar(value: float, group=None) -> float:
    """Mean of one host scalar over the ranks (the LR finder's per-step loss)."""
    w = world()[1]
    return allreduce_sums([value], group)[0] / w if w > 1 else float(value)


def barrier() -> None:
    if world()[1] > 1:
        dist.barrier()


def broadcast_module(module: torch.nn.Module, src: int = 0) -> None:
    """Parameters and buffers of a plain nn.Module from rank ``src`` (the engine broadcasts its two flat buffers instead)."""
    if world()[1] == 1:
        return
    for t in list(module.parameters()) + list(module.buffers()):
        dist.broadcast(t.data, src)


def average_gradients(module: torch.nn.Module) -> None:
    """Mean of every parameter gradient over the ranks, for models that do not all-reduce inside backward (the engine does)."""
    w = world()[1]
    if w == 1:
        return
    grads = [p.grad for p in module.parameters() if p.grad is not None]
    if not grads:
        return
    flat = torch.cat([g.reshape(-1) for g in grads])
    dist.all_reduce(flat)
    flat /= w
    off = 0
    for g in grads:
        g.copy_(flat[off:off + g.numel()].view_as(g))
        off += g.numel()


def padded_len(n: int, world_size: int) -> int:
    """n rounded up to a multiple of the world size (the key volume's length, so that it splits into equal shards)."""
    return (n + world_size - 1) // world_size * world_size


def exchange_keys_sharded(keys: torch.Tensor, unpack, want_probs: bool, all_ranks: bool, group=None):
    """The ONE exchange of a sharded prediction without shipping the merged key volume to every rank:

      1. reduce-scatter(MAX) of the packed keys - every rank receives the merged keys of ITS 1/N of the voxels
         (4 B/voxel through the ring once, instead of twice for an all-reduce);
      2. every rank unpacks its shard (``unpack(shard_int32) -> (labels uint8, probs float16 | None)``): the unpack is sharded;
      3. the labels / probabilities (1 + 2 B/voxel) are gathered on rank 0 only, or all-gathered when every rank is to return
         the volume (``all_ranks``).

    ``keys``: int32 view of the key volume, length a multiple of the world size (padded_len; padding keys are 0).
    Returns (labels, probs) flat tensors of that padded length on the ranks that receive the result, else (None, None).
    gloo has no reduce-scatter: there the shard is cut out of an all-reduce (the CPU tests exercise steps 2 and 3 as written)."""
    rank, w = world()
    if w == 1:
        return unpack(keys)
    n = keys.numel()
    assert n % w == 0, "exchange_keys_sharded: pad the key volume to a multiple of the world size (padded_len)"
    per = n // w
    if dist.get_backend(group) == "nccl":
        shard = torch.empty(per, dtype=keys.dtype, device=keys.device)
        dist.reduce_scatter_tensor(shard, keys, op=dist.ReduceOp.MAX, group=group)
    else:
        dist.all_reduce(keys, op=dist.ReduceOp.MAX, group=group)
        shard = keys[rank * per:(rank + 1) * per].clone()
    lab, prb = unpack(shard)
    out = []
    for t in ((lab, prb) if want_probs else (lab,)):
        if all_ranks:
            full = torch.empty(n, dtype=t.dtype, device=t.device)
            dist.all_gather_into_tensor(full, t.contiguous(), group=group)
            out.append(full)
        else:
            full = torch.empty(n, dtype=t.dtype, device=t.device) if rank == 0 else None
            dist.gather(t.contiguous(), list(full.view(w, per).unbind(0)) if rank == 0 else None, dst=0, group=group)
            out.append(full)
    if not want_probs:
        out.append(None)
    return out[0], out[1]
