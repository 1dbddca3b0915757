"""Multi-GPU harness for the hot path: one process per GPU, scans sharded contiguously over ranks, no
collective on the data path (scans are independent end to end); ONE all_gather of the per-scan
result rows at the end of a batch (RCCL over xGMI via torch.distributed backend "nccl"; "gloo" in CPU tests)."""
import os

import torch
import torch.distributed as dist


def env_rank_world():
    return int(os.environ.get("RANK", "0")), int(os.environ.get("WORLD_SIZE", "1")), int(os.environ.get("LOCAL_RANK", "0"))


def forced():
    """ETCH_FORCE_DIST=1: run the collective path even with ONE rank (a world-size-1 RCCL process group), so that RCCL initialisation,
    the device-tensor all_gather and the fp64 all_reduce(MAX) execute on a 1-GPU box exactly as they will on an 8-GPU node."""
    return os.environ.get("ETCH_FORCE_DIST", "") not in ("", "0")


def _active():
    return dist.is_initialized() and (dist.get_world_size() > 1 or forced())


def init(backend=None):
    """Initialise torch.distributed from the torchrun environment (no-op for a single process unless ETCH_FORCE_DIST is set)."""
    rank, world, local = env_rank_world()
    if (world > 1 or forced()) and not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29531")
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
    """NUMA node of every GPU this process can open, in HIP ordinal order, read from the KFD topology in sysfs (NO HIP call:
    this runs before the process touches the GPU).  KFD lists CPU nodes first (cpu_cores_count > 0; the k-th CPU node is NUMA
    node k) and then one node per GPU in the order HIP enumerates them.  The socket of a GPU is taken from the CPU node whose
    io_links name it (the CPU side is always readable), else from the GPU's own links.  In a container that was handed a
    subset of the GPUs the other GPUs' `properties` are unreadable (EPERM, seen on the bench pool): those are skipped, so
    ordinal i is the i-th GPU this process may use.  [] when sysfs has no KFD."""
    try:
        ids = sorted(int(d) for d in os.listdir(kfd_root) if d.isdigit())
    except OSError:
        return []
    props = {i: _read_props(os.path.join(kfd_root, str(i), "properties")) for i in ids}
    cpu_nodes = [i for i in ids if props[i].get("cpu_cores_count", 0) > 0]
    gpus = [i for i in ids if i not in cpu_nodes]
    readable = [i for i in gpus if props[i].get("simd_count", 0) > 0]
    gpus = readable or gpus

    def links(i):
        d = os.path.join(kfd_root, str(i), "io_links")
        try:
            return [_read_props(os.path.join(d, l, "properties")).get("node_to") for l in sorted(os.listdir(d))]
        except OSError:
            return []

    socket = {}
    for k, c in enumerate(cpu_nodes):
        for to in links(c):
            if to in gpus:
                socket.setdefault(to, k)
    for g in gpus:
        if g not in socket:
            for to in links(g):
                if to in cpu_nodes:
                    socket[g] = cpu_nodes.index(to)
                    break
    return [socket.get(g) for g in gpus]


def _split_by_core(cpus, k, n, cpu_root="/sys/devices/system/cpu"):
    """k-th of n even parts of `cpus`, split by PHYSICAL core (SMT siblings stay together on one rank)."""
    core = {}
    for c in cpus:
        try:
            with open(os.path.join(cpu_root, f"cpu{c}", "topology", "thread_siblings_list")) as f:
                core[c] = min(_parse_cpulist(f.read()))
        except (OSError, ValueError):
            core[c] = c
    cores = sorted(set(core.values()))
    per = max(1, len(cores) // n)
    mine = set(cores[k * per:(k + 1) * per] if k < n - 1 else cores[k * per:])
    return [c for c in cpus if core[c] in mine]


def local_cpu_set(local, world, kfd_root="/sys/class/kfd/kfd/topology/nodes", node_root="/sys/devices/system/node"):
    """The host cores rank `local` of `world` ranks on this node should run on: the cores of its GPU's NUMA node, split evenly
    (by physical core) among the ranks whose GPUs share that node (the host enqueues ~400 launches per step, so each
    rank wants its own NUMA-local cores).  Falls back to an even split of the current affinity mask when the topology is unreadable."""
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
            mine = _split_by_core(cpus, sharing.index(local), len(sharing))
            if mine:
                return mine
    return _split_by_core(avail, local % max(world, 1), max(world, 1)) or avail


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
    if not _active():
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
    if _active():
        if dist.get_backend() == "nccl":
            dist.barrier(device_ids=[torch.cuda.current_device()])
        else:
            dist.barrier()


def max_over_ranks(value, device):
    if not _active():
        return value
    t = _coll(torch.tensor([value], dtype=torch.float64, device=device))
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())
