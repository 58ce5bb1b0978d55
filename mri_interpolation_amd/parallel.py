"""Data parallelism over the 8 GPUs of one node: one process per GPU, RCCL over xGMI.

The reference has no distributed code.  The path shards naturally:
coordinates are independent, all parameters are shared, so each rank trains on its own
slab of the volume (contiguous range of the slowest axis, equal local batch) and one
gradient reduction per step makes the update identical on every rank (mean of equal-sized
local means = global mean, i.e. DDP semantics).

Everything here is host logic over `torch.distributed`; it works with the `gloo` backend
on CPU tensors too, which is how the N > 1 path is tested without GPUs.
"""
import os
from typing import Optional, Tuple

import torch
import torch.distributed as dist


def env_world() -> Tuple[int, int, int]:
    """(rank, world_size, local_rank) from the torchrun environment (1 process = 1 GPU).
    MRI_SINGLE_DEVICE=1 maps every rank to device 0 (multi-rank rehearsal on a one-GPU box)."""
    local = 0 if os.environ.get("MRI_SINGLE_DEVICE") == "1" else int(os.environ.get("LOCAL_RANK", "0"))
    return int(os.environ.get("RANK", "0")), int(os.environ.get("WORLD_SIZE", "1")), local


def init(backend: Optional[str] = None) -> Tuple[int, int, int]:
    """Join the process group described by the environment; no-op for a single process."""
    rank, world, local = env_world()
    ipc_default(world)
    if world > 1 and not dist.is_initialized():
        if backend is None:  # "nccl" is RCCL on ROCm; MRI_DIST_BACKEND=gloo rehearses on one GPU
            backend = os.environ.get("MRI_DIST_BACKEND",
                                     "nccl" if torch.cuda.is_available() else "gloo")
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if backend == "nccl":
            torch.cuda.set_device(local)
        dist.init_process_group(backend=backend, rank=rank, world_size=world)
    return rank, world, local


def ipc_default(world: Optional[int] = None) -> None:
    """RCCL shares device buffers between the ranks of a node through dmabuf IPC; the legacy IPC mode is
    not supported by every host driver (hipIpcGetMemHandle: invalid argument).  Multi-process runs only,
    and only a default: an explicit setting of the launcher wins.  Read when HIP initialises, so entry
    points call this before anything touches the GPU; a single process never sets it."""
    if world is None:
        world = int(os.environ.get("WORLD_SIZE", "1"))
    if world > 1:
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")


def world_size() -> int:
    return dist.get_world_size() if dist.is_initialized() else 1


def slab_range(n_slow: int, rank: int, world: int) -> Tuple[int, int]:
    """Rows [lo, hi) of the slowest axis owned by `rank`; remainders go to the first ranks."""
    if world > n_slow:
        raise ValueError(f"cannot cut {n_slow} slices into {world} slabs; shard another axis "
                         "or use flat_range()")
    base, rem = divmod(n_slow, world)
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


def flat_range(n_voxels: int, rank: int, world: int) -> Tuple[int, int]:
    """Contiguous flat-index range per rank (for volumes whose slow axis is shorter than
    the world size, e.g. the sample's 6 z-slices on 8 GPUs -- SURVEY.md 8(d) cfg 5)."""
    base, rem = divmod(n_voxels, world)
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


def voxel_range(shape, rank: int, world: int) -> Tuple[int, int]:
    """Flat C-order voxel range of this rank's slab (slowest axis), falling back to a flat
    split when the slowest axis has fewer slices than ranks."""
    n = 1
    for s in shape:
        n *= int(s)
    if world == 1:
        return 0, n
    if shape[0] >= world:
        lo, hi = slab_range(int(shape[0]), rank, world)
        per_slice = n // int(shape[0])
        return lo * per_slice, hi * per_slice
    return flat_range(n, rank, world)


def all_reduce_sum(flat: torch.Tensor) -> torch.Tensor:
    """In-place sum over ranks of one flat buffer: ONE collective per step for all
    gradients (tables + MLP).  Gradients are pre-divided by world_size in the loss kernel."""
    if dist.is_initialized() and dist.get_world_size() > 1:
        dist.all_reduce(flat, op=dist.ReduceOp.SUM)
    return flat


def all_reduce_async(flat: torch.Tensor):
    """Start an in-place sum over ranks of a contiguous gradient slice; returns a handle for
    wait_all().  With the RCCL backend the collective runs on the process group's own stream,
    ordered after the work already queued on the current stream, so later kernels overlap it."""
    if dist.is_initialized() and dist.get_world_size() > 1:
        return dist.all_reduce(flat, op=dist.ReduceOp.SUM, async_op=True)
    return None


def shard_range(numel: int, rank: int, world: int):
    """[lo, hi) of a flat buffer of `numel` elements (a multiple of `world`) owned by `rank`."""
    if numel % world:
        raise ValueError(f"flat buffer of {numel} elements does not divide over {world} ranks")
    per = numel // world
    return rank * per, (rank + 1) * per


_staging = {}  # (device, dtype, numel) -> shard-sized buffer: the collectives below never alias


def _shard_staging(like: torch.Tensor) -> torch.Tensor:
    key = (like.device, like.dtype, like.numel())
    buf = _staging.get(key)
    if buf is None:
        _staging.clear()  # one shard size per process at a time
        buf = _staging[key] = torch.empty_like(like)
    return buf


def reduce_scatter_sum(flat: torch.Tensor, rank: int, world: int) -> torch.Tensor:
    """Sum `flat` over ranks, leaving this rank's shard (shard_range) reduced in `flat` and returning that
    view; the other shards of `flat` hold garbage afterwards.  RCCL: reduce_scatter (each rank receives
    1/world of the bytes an all-reduce moves to it) into a STAGING buffer of one shard (6 MB at config 4,
    8 ranks) that is copied back: input and output of the collective never overlap -- the in-place form
    `reduce_scatter_tensor(flat[lo:hi], flat)` is legal NCCL but has never met a multi-GPU RCCL here.
    gloo has no reduce_scatter, so the CPU rehearsal all-reduces and keeps the shard."""
    lo, hi = shard_range(flat.numel(), rank, world)
    shard = flat[lo:hi]
    if dist.is_initialized() and dist.get_world_size() > 1:
        if dist.get_backend() == "nccl":
            out = _shard_staging(shard)
            dist.reduce_scatter_tensor(out, flat, op=dist.ReduceOp.SUM)
            shard.copy_(out)
        else:
            dist.all_reduce(flat, op=dist.ReduceOp.SUM)
    return shard


def all_gather_shards(flat: torch.Tensor, rank: int, world: int) -> torch.Tensor:
    """Every rank contributes its shard of `flat` and receives all the others (the shard is sent from a
    staging copy: the collective's input never lies inside its output)."""
    lo, hi = shard_range(flat.numel(), rank, world)
    if dist.is_initialized() and dist.get_world_size() > 1:
        if dist.get_backend() == "nccl":
            src = _shard_staging(flat[lo:hi])
            src.copy_(flat[lo:hi])
            dist.all_gather_into_tensor(flat, src)
        else:
            parts = [torch.empty_like(flat[lo:hi]) for _ in range(world)]
            dist.all_gather(parts, flat[lo:hi].clone())
            flat.copy_(torch.cat(parts))
    return flat


def gather_optimizer_state(optimizer, rank: int, world: int) -> None:
    """dp_mode "reduce_scatter" leaves every rank with the Adam moments of its own shard only; before a
    checkpoint is written every rank calls this: the padded moment buffers are all-gathered shard by shard
    (as the parameters are after every step) and the optimiser is marked whole again."""
    flat = optimizer.flatten()
    if getattr(optimizer, "sharded", False) and world > 1:
        all_gather_shards(flat._exp_avg_all, rank, world)
        all_gather_shards(flat._exp_avg_sq_all, rank, world)
    optimizer.sharded = False


def wait_all(handles):
    """Make the current stream (or the host, for gloo) wait for the pending reductions."""
    for h in handles:
        if h is not None:
            h.wait()


def all_reduce_max(value: float, device) -> float:
    if not (dist.is_initialized() and dist.get_world_size() > 1):
        return value
    t = torch.tensor([value], dtype=torch.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())


def barrier():
    if dist.is_initialized() and dist.get_world_size() > 1:
        if dist.get_backend() == "nccl":  # name the device: no guessing, no warning
            dist.barrier(device_ids=[torch.cuda.current_device()])
        else:
            dist.barrier()
