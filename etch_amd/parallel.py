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


# ---------------------------------------------------------------------------------------------- host-side placement
def _read_props(path):
    out = {}
    try:
        with open(path) as f:
            for line in f:
                k, _, v = line.strip().partition(" ")
                if v.strip().lstrip("-").isdigit():
                    out[k] = int(v)
    except OSError:
        pass
    return out


def _parse_cpulist(txt):
    cpus = set()
    for part in txt.strip().split(","):
        if not part:
            continue
        lo, _, hi = part.partition("-")
        cpus.update(range(int(lo), int(hi or lo) + 1))
    return cpus


def gpu_numa_nodes(kfd_root="/sys/class/kfd/kfd/topology/nodes"):
    """NUMA node of every GPU in HIP ordinal order, read from the KFD topology in sysfs (NO HIP call: this runs before the
    process touches the GPU).  KFD lists CPU nodes (cpu_cores_count > 0; node id == NUMA node) and GPU nodes (simd_count > 0,
    in the order HIP enumerates them); a GPU's io_link to a CPU node names its NUMA-local socket.  [] when sysfs has no KFD."""
    try:
        ids = sorted(int(d) for d in os.listdir(kfd_root) if d.isdigit())
    except OSError:
        return []
    props = {i: _read_props(os.path.join(kfd_root, str(i), "properties")) for i in ids}
    cpu_nodes = [i for i in ids if props[i].get("cpu_cores_count", 0) > 0]
    out = []
    for i in ids:
        if props[i].get("simd_count", 0) <= 0 or props[i].get("cpu_cores_count", 0) > 0:
            continue
        numa = None
        links = os.path.join(kfd_root, str(i), "io_links")
        try:
            for l in sorted(os.listdir(links)):
                to = _read_props(os.path.join(links, l, "properties")).get("node_to")
                if to in cpu_nodes:
                    numa = cpu_nodes.index(to)
                    break
        except OSError:
            pass
        out.append(numa)
    return out


def local_cpu_set(local, world, kfd_root="/sys/class/kfd/kfd/topology/nodes", node_root="/sys/devices/system/node"):
    """The host cores rank `local` of `world` ranks on this node should run on: the cores of its GPU's NUMA node, split evenly
    among the ranks whose GPUs share that node (the host enqueues ~400 launches per step, so each rank wants its own
    NUMA-local cores).  Falls back to an even contiguous split of the current affinity mask when the topology is unreadable."""
    avail = sorted(os.sched_getaffinity(0))
    numa = gpu_numa_nodes(kfd_root)
    if local < len(numa) and numa[local] is not None:
        try:
            with open(os.path.join(node_root, f"node{numa[local]}", "cpulist")) as f:
                cpus = sorted(_parse_cpulist(f.read()) & set(avail))
        except OSError:
            cpus = []
        sharing = [r for r in range(min(world, len(numa))) if numa[r] == numa[local]]
        if cpus and local in sharing:
            k, n = sharing.index(local), len(sharing)
            per = max(1, len(cpus) // n)
            mine = cpus[k * per:(k + 1) * per] if k < n - 1 else cpus[k * per:]
            if mine:
                return mine
    per = max(1, len(avail) // max(world, 1))
    mine = avail[local * per:(local + 1) * per]
    return mine or avail


def pin_to_local_cores(local, world):
    """sched_setaffinity of the calling process (call BEFORE GPU initialisation so the runtime's helper threads inherit it)."""
    cpus = local_cpu_set(local, world)
    try:
        os.sched_setaffinity(0, cpus)
    except OSError:
        return sorted(os.sched_getaffinity(0))
    return cpus


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
