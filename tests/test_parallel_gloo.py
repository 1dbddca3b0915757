"""world_size-2 CPU (gloo) test of the N>1 path: contiguous scan sharding + the single end-of-batch all_gather."""
import os
import socket

import torch
import torch.multiprocessing as mp


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, total, q):
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    from etch_amd import parallel as P
    r, w, _ = P.init("gloo")
    s, e = P.shard_range(total, r, w)
    rows = torch.stack([torch.arange(s, e, dtype=torch.float32), torch.arange(s, e, dtype=torch.float32) ** 2], 1)
    P.barrier()
    allrows = P.gather_rows(rows)
    mx = P.max_over_ranks(float(rank + 1), torch.device("cpu"))
    q.put((rank, (s, e), allrows.tolist(), mx))
    torch.distributed.destroy_process_group()


def test_shard_ranges_cover_everything():
    from etch_amd.parallel import shard_range
    for total in (0, 1, 7, 32, 256):
        for world in (1, 2, 3, 8):
            ranges = [shard_range(total, r, world) for r in range(world)]
            assert ranges[0][0] == 0 and ranges[-1][1] == total
            assert all(ranges[i][1] == ranges[i + 1][0] for i in range(world - 1))
            sizes = [e - s for s, e in ranges]
            assert max(sizes) - min(sizes) <= 1


def test_two_rank_gather_gloo():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port, total = _free_port(), 7          # ragged: 4 + 3
    procs = [ctx.Process(target=_worker, args=(r, 2, port, total, q)) for r in range(2)]
    [p.start() for p in procs]
    res = sorted(q.get(timeout=120) for _ in range(2))
    [p.join(timeout=60) for p in procs]
    assert [r[1] for r in res] == [(0, 4), (4, 7)]
    expect = [[float(i), float(i * i)] for i in range(total)]
    assert res[0][2] == expect and res[1][2] == expect
    assert res[0][3] == 2.0 and res[1][3] == 2.0


def _bench(extra_env, *argv, timeout=600):
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_PORT")}
    env.update(extra_env)
    return subprocess.run([sys.executable, os.path.join(root, "bench.py"), *argv], env=env, capture_output=True, text=True, timeout=timeout)


def test_bench_self_launch_two_ranks_control_flow():
    """`python bench.py --gpus 2` from a plain interpreter (no torchrun): the parent starts the two ranks itself and relays rank 0's
    JSON line.  ETCH_BENCH_DRY replaces the GPU step by a stub so the WHOLE control flow of the N > 1 path runs here on CPU over gloo:
    rank environment, NUMA / core pinning, sharding of distinct batches, barrier + max-over-ranks timing, the end-of-job all_gather."""
    import json
    r = _bench({"ETCH_BENCH_DRY": "1"}, "--gpus", "2", "--steps", "3", "--warmup", "1", "--batch", "4", "--points", "128")
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1                                    # ONE JSON line, from rank 0 only
    out = json.loads(lines[0])
    assert out["n_gpus"] == 2 and out["steps"] == 3 and out["scaling"] == "weak" and out["config"]["global_batch"] == 8
    assert out["config"]["launcher"] == "self" and out["config"]["distinct_batches"] == 4
    assert out["gathered_rows"]["scans_reported"] == 8        # both ranks' rows arrived, in scan order
    assert out["value"] == __import__("pytest").approx(2 * 4 * 3 / (out["ms_per_step"] * 3e-3), rel=1e-3)


def test_bench_self_launch_propagates_a_rank_failure():
    """A rank that dies must fail the job (and must not leave its peers waiting in a barrier)."""
    r = _bench({"ETCH_BENCH_DRY": "1", "ETCH_BENCH_DRY_FAIL_RANK": "1"}, "--gpus", "2", "--steps", "2", "--warmup", "1", "--batch", "2",
               "--points", "64", timeout=300)
    assert r.returncode == 3 and "ranks failed" in r.stderr
    assert not [l for l in r.stdout.splitlines() if l.startswith("{")]


def test_core_pinning_from_kfd_topology(tmp_path):
    """local_cpu_set on a fake sysfs tree shaped like the 8-GPU MI355X node of the bench pool (gpurun_out/topology.txt): KFD nodes
    0, 1 = the two sockets, whose io_links name GPU nodes 2-5 / 6-9; GPU `properties` are only readable for the GPUs the
    container was handed.  Each rank gets its own quarter of its socket's cores."""
    from etch_amd import parallel as P
    avail = sorted(os.sched_getaffinity(0))
    half = max(1, len(avail) // 2)
    lists = [avail[:half], avail[half:] or avail[:half]]

    def tree(name, readable):
        kfd, nodes = tmp_path / name / "kfd", tmp_path / name / "node"
        for i in range(10):
            d = kfd / str(i)
            cpu = i < 2
            peers = ([1 - i] + list(range(2, 6) if i == 0 else range(6, 10))) if cpu else [0 if i < 6 else 1]
            for k, to in enumerate(peers):
                (d / "io_links" / str(k)).mkdir(parents=True)
                (d / "io_links" / str(k) / "properties").write_text(f"type 2\nnode_from {i}\nnode_to {to}\n")
            if cpu or i in readable:
                (d / "properties").write_text(f"cpu_cores_count {64 if cpu else 0}\nsimd_count {0 if cpu else 1024}\n")
        for k in range(2):
            (nodes / f"node{k}").mkdir(parents=True)
            (nodes / f"node{k}" / "cpulist").write_text(",".join(str(c) for c in lists[k]) + "\n")
        return str(kfd), str(nodes)

    kfd, nodes = tree("all8", range(2, 10))
    assert P.gpu_numa_nodes(kfd) == [0, 0, 0, 0, 1, 1, 1, 1]
    sets = [P.local_cpu_set(r, 8, kfd, nodes) for r in range(8)]
    assert all(s for s in sets)
    for r in range(8):
        assert set(sets[r]) <= set(lists[0 if r < 4 else 1])
    if len(lists[0]) >= 4:
        assert all(not (set(sets[a]) & set(sets[b])) for a in range(4) for b in range(a + 1, 4))
    # a container that was handed one GPU of the second socket (KFD node 7): ordinal 0 is that GPU
    kfd1, nodes1 = tree("one", [7])
    assert P.gpu_numa_nodes(kfd1) == [1]
    assert set(P.local_cpu_set(0, 1, kfd1, nodes1)) == set(lists[1])
    # unreadable topology -> even split of the affinity mask
    got = P.local_cpu_set(1, 2, str(tmp_path / "none"), nodes)
    assert got and set(got) <= set(avail) and (len(avail) < 2 or not set(got) & set(P.local_cpu_set(0, 2, str(tmp_path / "none"), nodes)))


def test_bench_preflight_describes_the_8_rank_job_without_a_gpu():
    """`bench.py --gpus 8 --preflight`: one JSON line with every rank's shard, CPU set and the collective plan; no GPU, no child processes."""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "8", "--preflight"], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr[-500:]
    d = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][-1])
    assert d["preflight"] and d["n_gpus"] == 8 and d["global_batch"] == 256 and d["scaling"] == "weak"
    assert [x["scans"] for x in d["ranks"]] == [[32 * k, 32 * k + 32] for k in range(8)]
    assert all(x["device"] == f"cuda:{k}" for k, x in enumerate(d["ranks"])) and "all_gather" in d["collective"]["calls"]
    r4 = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "8", "--config", "4", "--preflight"], capture_output=True, text=True, timeout=300)
    d4 = json.loads([l for l in r4.stdout.splitlines() if l.startswith("{")][-1])
    assert d4["global_batch"] == 64 and d4["points"] == 20000
