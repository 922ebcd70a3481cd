"""Multi-GPU plumbing: reads shard by ordinal (no data-path collective); the only exchange step is
gathering per-read alignment records (variable-length block arrays) to rank 0 for ordered
emission: all_gather of the per-rank sizes, then one padded gather.
Backend "nccl" is RCCL over xGMI on the GPU box; the same code runs on "gloo" in the CPU tests."""
import torch
import torch.distributed as dist


def shard_of(ordinal, world_size):
    """Rank that owns read `ordinal` (static, deterministic)."""
    return ordinal % world_size


def shard_ordinals(n_total, rank, world_size):
    return list(range(rank, n_total, world_size))


def gather_records(local: torch.Tensor, dst=0):
    """Gather a 1-D tensor of per-rank variable length to `dst`.  Returns the list of per-rank tensors on
    dst (None elsewhere).  Works for world_size 1 without a process group."""
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size() == 1:
        return [local]
    ws, rank = dist.get_world_size(), dist.get_rank()
    size = torch.tensor([local.numel()], dtype=torch.int64, device=local.device)
    sizes = [torch.zeros_like(size) for _ in range(ws)]
    dist.all_gather(sizes, size)
    sizes = [int(s.item()) for s in sizes]
    mx = max(max(sizes), 1)
    padded = torch.zeros(mx, dtype=local.dtype, device=local.device)
    padded[:local.numel()] = local
    if rank == dst:
        bufs = [torch.empty(mx, dtype=local.dtype, device=local.device) for _ in range(ws)]
        dist.gather(padded, bufs, dst=dst)
        return [b[:n] for b, n in zip(bufs, sizes)]
    dist.gather(padded, None, dst=dst)
    return None
