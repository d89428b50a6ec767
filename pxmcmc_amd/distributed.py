"""
Multi-GPU host logic: chains are independent, so they shard over ranks with no collective on the
data path (the reference runs extra chains as extra processes, experiments/earthtopography/main.py:31-36).
One process per GPU; ``torch.distributed`` (RCCL on ROCm, gloo in CPU tests) is used only for the
start/stop barrier and the max-over-ranks elapsed time of a benchmark or a final gather of summaries.
"""
import os

import torch
import torch.distributed as dist


def env_rank_world():
    return int(os.environ.get("RANK", "0")), int(os.environ.get("LOCAL_RANK", "0")), int(os.environ.get("WORLD_SIZE", "1"))


def shard_chains(total_chains, rank, world):
    """(first global chain id, number of chains) owned by ``rank``: contiguous blocks, remainder to the low ranks."""
    if not 0 <= rank < world:
        raise ValueError("rank outside [0, world)")
    base, rem = divmod(int(total_chains), int(world))
    count = base + (1 if rank < rem else 0)
    first = rank * base + min(rank, rem)
    return first, count


def init(backend=None, device_id=None):
    """Initialise the default process group from the torchrun environment (no-op for one process)."""
    rank, local_rank, world = env_rank_world()
    if world > 1 and not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if backend is None:
            backend = "nccl" if torch.cuda.is_available() else "gloo"
        kw = {}
        if backend == "nccl" and device_id is not None:
            kw["device_id"] = device_id
        dist.init_process_group(backend, **kw)
    return rank, local_rank, world


def barrier():
    """device idle on every rank, then the ranks aligned (and the barrier's own work drained)"""
    if torch.cuda.is_available():
        torch.cuda.synchronize()
    if dist.is_initialized():
        dist.barrier()
        if torch.cuda.is_available():
            torch.cuda.synchronize()


def max_over_ranks(value):
    """max of a python float over all ranks (the timing rule of bench.py)."""
    if not dist.is_initialized():
        return float(value)
    dev = "cuda" if dist.get_backend() == "nccl" else "cpu"
    t = torch.tensor([float(value)], dtype=torch.float64, device=dev)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())


def all_gather_float(value):
    """one python float per rank -> list of world_size floats in rank order (a plain fixed-size all-gather: no object
    pickling on the path the 8-GPU benchmark takes)"""
    if not dist.is_initialized():
        return [float(value)]
    dev = "cuda" if dist.get_backend() == "nccl" else "cpu"
    t = torch.tensor([float(value)], dtype=torch.float64, device=dev)
    out = [torch.empty_like(t) for _ in range(dist.get_world_size())]
    dist.all_gather(out, t)
    return [float(o.item()) for o in out]


def count_ranks():
    """number of ranks that take part in the process group, by an all-reduce (SUM of ones) over the group's own
    transport (RCCL for ``nccl``): 1 without a group"""
    if not dist.is_initialized():
        return 1
    dev = "cuda" if dist.get_backend() == "nccl" else "cpu"
    t = torch.ones(1, dtype=torch.int64, device=dev)
    dist.all_reduce(t, op=dist.ReduceOp.SUM)
    return int(t.item())


def gather_summaries(array):
    """end-of-run gather of per-chain summaries ([C_local, ...] numpy or tensor) onto every rank, chain order kept"""
    t = torch.as_tensor(array)
    if not dist.is_initialized():
        return t
    if dist.get_backend() == "nccl":
        t = t.cuda()
    sizes = [None] * dist.get_world_size()
    dist.all_gather_object(sizes, int(t.shape[0]))
    nmax = max(sizes)  # all_gather wants equal shapes: pad the short ranks, trim after
    pad = torch.zeros((nmax,) + tuple(t.shape[1:]), dtype=t.dtype, device=t.device)
    pad[: t.shape[0]] = t
    out = [torch.empty_like(pad) for _ in sizes]
    dist.all_gather(out, pad)
    return torch.cat([o[:n] for o, n in zip(out, sizes)]).cpu()
