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
