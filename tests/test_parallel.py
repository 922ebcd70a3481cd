"""N>1 path on CPU: world_size-2 gloo processes shard read ordinals and gather variable-length records to rank 0."""
import os
import socket

import torch
import torch.distributed as dist
import torch.multiprocessing as mp


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, ws, port, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=ws)
    from lra_amd import parallel
    mine = parallel.shard_ordinals(11, rank, ws)
    # record of read i: i+1 copies of the value i
    local = torch.cat([torch.full((i + 1,), i, dtype=torch.int32) for i in mine]) if mine else torch.zeros(0, dtype=torch.int32)
    got = parallel.gather_records(local, dst=0)
    if rank == 0:
        q.put([g.tolist() for g in got])
    else:
        assert got is None
    dist.barrier()
    dist.destroy_process_group()


def test_shard_and_gather_world2():
    ws, port = 2, _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    ps = [ctx.Process(target=_worker, args=(r, ws, port, q)) for r in range(ws)]
    for p in ps:
        p.start()
    got = q.get(timeout=120)
    for p in ps:
        p.join(timeout=120)
        assert p.exitcode == 0
    from lra_amd import parallel
    assert sorted(parallel.shard_ordinals(11, 0, 2) + parallel.shard_ordinals(11, 1, 2)) == list(range(11))
    for r in range(ws):
        exp = []
        for i in parallel.shard_ordinals(11, r, ws):
            exp += [i] * (i + 1)
        assert got[r] == exp
    assert all(parallel.shard_of(i, 2) == i % 2 for i in range(11))


def test_gather_single_process():
    from lra_amd import parallel
    t = torch.arange(5)
    assert parallel.gather_records(t)[0] is t
