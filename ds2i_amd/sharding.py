"""Multi-GPU sharding of a query batch: one process per GPU, index replicated in every GPU's HBM,
queries partitioned into contiguous slices, results concatenated on rank 0. The path has no exchange
step, so there is no data-path collective; torch.distributed (RCCL on GPUs, gloo on CPU)
is only used for rendezvous, barriers, the max-over-ranks timing and the result gather."""
import os

import numpy as np


def init_distributed(backend=None):
    """-> (rank, local_rank, world_size, dist-or-None). Reads RANK / LOCAL_RANK / WORLD_SIZE / MASTER_*."""
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world <= 1:
        return rank, local_rank, world, None
    import torch
    import torch.distributed as dist
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29533")
    if backend is None:
        backend = "nccl" if torch.cuda.is_available() else "gloo"
    if backend == "nccl":
        torch.cuda.set_device(local_rank)  # one process per GPU
    if not dist.is_initialized():
        dist.init_process_group(backend, rank=rank, world_size=world)
    return rank, local_rank, world, dist


def query_slice(nq, rank, world):
    """Contiguous slice [begin, end) of a batch of nq queries owned by `rank` (sizes differ by at most 1)."""
    base, rem = divmod(nq, world)
    begin = rank * base + min(rank, rem)
    return begin, begin + base + (1 if rank < rem else 0)


def _device(dist):
    import torch
    return "cuda" if dist.get_backend() == "nccl" else "cpu"


def share_bytes(dist, rank, blob, path):
    """Rank 0 holds `blob`; every rank returns it. Goes through a file (page cache / tmpfs), not a collective:
    a GOV2-scale index image is gigabytes."""
    if dist is None:
        return blob
    if rank == 0:
        with open(path, "wb") as f:
            f.write(blob)
    dist.barrier()
    if rank != 0:
        with open(path, "rb") as f:
            blob = f.read()
    dist.barrier()
    if rank == 0:
        os.remove(path)
    return blob


def max_over_ranks(dist, value):
    if dist is None:
        return float(value)
    import torch
    t = torch.tensor([float(value)], dtype=torch.float64, device=_device(dist))
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())


def broadcast_int(dist, value, src=0):
    if dist is None:
        return int(value)
    import torch
    t = torch.tensor([int(value)], dtype=torch.int64, device=_device(dist))
    dist.broadcast(t, src)
    return int(t.item())


def gather_concat(dist, rank, world, local):
    """Concatenates per-rank result arrays (first axis) in rank order on every rank."""
    local = np.ascontiguousarray(local)
    if dist is None:
        return local
    import torch
    dev = _device(dist)
    n = torch.tensor([local.shape[0]], dtype=torch.int64, device=dev)
    sizes = [torch.zeros(1, dtype=torch.int64, device=dev) for _ in range(world)]
    dist.all_gather(sizes, n)
    sizes = [int(s.item()) for s in sizes]
    m = max(sizes) if sizes else 0
    pad = np.zeros((m,) + local.shape[1:], dtype=local.dtype)
    pad[:local.shape[0]] = local
    raw = torch.from_numpy(pad.view(np.uint8).reshape(m, -1) if m else np.zeros((0, 1), np.uint8)).to(dev)
    outs = [torch.zeros_like(raw) for _ in range(world)]
    dist.all_gather(outs, raw)
    parts = []
    for r in range(world):
        a = outs[r].cpu().numpy().reshape(-1).view(local.dtype).reshape((m,) + local.shape[1:])
        parts.append(a[:sizes[r]])
    return np.concatenate(parts, axis=0)
