"""CPU, world_size 2, gloo: the N>1 path -- flat-gradient all-reduce, parameter broadcast, episode sharding
and reward gathering used by the data-parallel DAGGER loop and by bench.py."""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    port = s.getsockname()[1]
    s.close()
    return port


def _worker(rank, world, port, q):
    sys.path.insert(0, ROOT)
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank),
                      MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port))
    from multiagent_gnn_policies_amd import parallel
    rk, w, _ = parallel.init_from_env(backend='gloo')
    assert (rk, w) == (rank, world) and parallel.is_distributed()
    sync = parallel.FlatGradSync()
    # parameter broadcast: every rank ends with rank 0's flat buffer
    flat = torch.full((1730,), float(rank + 1))
    sync.broadcast_(flat)
    ok_bcast = bool((flat == 1.0).all())
    # gradient mean
    g = torch.arange(1730, dtype=torch.float32) * (rank + 1)
    sync.all_reduce_mean_(g)
    expect = torch.arange(1730, dtype=torch.float32) * (sum(range(1, world + 1)) / world)
    ok_grad = bool(torch.allclose(g, expect))
    lo, hi = parallel.shard_range(7)
    rewards = parallel.all_gather_floats([float(i) for i in range(lo, hi)])
    import torch.distributed as dist
    dist.barrier()
    q.put((rank, ok_bcast, ok_grad, (lo, hi), rewards))
    dist.destroy_process_group()


@pytest.mark.timeout(120)
def test_flat_grad_sync_world2_gloo():
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=100) for _ in procs)
    for p in procs:
        p.join(timeout=30)
        assert p.exitcode == 0
    assert [r[0] for r in res] == [0, 1]
    assert all(r[1] and r[2] for r in res)
    assert res[0][3] == (0, 4) and res[1][3] == (4, 7)
    assert res[0][4] == res[1][4] == [float(i) for i in range(7)]
