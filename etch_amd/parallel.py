"""Multi-GPU harness for the hot path: one process per GPU, scans sharded contiguously over ranks, no
collective on the data path (scans are independent end to end); ONE all_gather of the per-scan
result rows at the end of a batch (RCCL over xGMI via torch.distributed backend "nccl"; "gloo" in CPU tests)."""
import os

import torch
import torch.distributed as dist


def env_rank_world():
    return int(os.environ.get("RANK", "0")), int(os.environ.get("WORLD_SIZE", "1")), int(os.environ.get("LOCAL_RANK", "0"))


def init(backend=None):
    """Initialise torch.distributed from the torchrun environment (no-op for a single process)."""
    rank, world, local = env_rank_world()
    if world > 1 and not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        backend = backend or os.environ.get("ETCH_DIST_BACKEND") or ("nccl" if torch.cuda.is_available() else "gloo")
        dist.init_process_group(backend, rank=rank, world_size=world)
    return rank, world, local


def _coll(t):
    """Tensor as the active backend can take it: gloo collectives run on host copies (used by CPU tests and by the
    single-GPU smoke of the multi-rank control flow); RCCL takes the device tensor as is."""
    if dist.get_backend() == "gloo" and t.is_cuda:
        return t.cpu()
    return t


def shard_range(total, rank, world):
    """Contiguous split of `total` scans over `world` ranks (first `total % world` ranks get one more)."""
    base, rem = divmod(total, world)
    start = rank * base + min(rank, rem)
    return start, start + base + (1 if rank < rem else 0)


def gather_rows(rows):
    """all_gather of per-scan result rows (n_local, k) -> (n_total, k) on every rank, rank order == scan order.
    Ranks may hold different n_local (ragged shards are padded to the maximum and trimmed)."""
    if not dist.is_initialized() or dist.get_world_size() == 1:
        return rows
    world = dist.get_world_size()
    dev = rows.device
    rows = _coll(rows)
    n = torch.tensor([rows.shape[0]], dtype=torch.int64, device=rows.device)
    counts = [torch.zeros_like(n) for _ in range(world)]
    dist.all_gather(counts, n)
    counts = [int(c.item()) for c in counts]
    mx = max(counts)
    pad = torch.zeros((mx,) + tuple(rows.shape[1:]), dtype=rows.dtype, device=rows.device)
    pad[: rows.shape[0]] = rows
    out = [torch.empty_like(pad) for _ in range(world)]
    dist.all_gather(out, pad)
    return torch.cat([o[:c] for o, c in zip(out, counts)], 0).to(dev)


def barrier():
    if dist.is_initialized() and dist.get_world_size() > 1:
        dist.barrier()


def max_over_ranks(value, device):
    if not dist.is_initialized() or dist.get_world_size() == 1:
        return value
    t = _coll(torch.tensor([value], dtype=torch.float64, device=device))
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())
