"""The N>1 path on CPU: world_size-2 gloo run of the chain partition + result gather used by bench.py / sample drivers."""
import os
import socket

import numpy as np
import torch
import torch.distributed as dist
import torch.multiprocessing as mp


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, nchains, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import sys
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    import cmblensing_jl_amd as C
    ids = C.partition_chains(nchains, world, rank)
    # each chain's "result": its seed-derived scalars and a small map
    vals = np.array([[C.chain_seed(100, c), c * c, -c] for c in ids], dtype=float).reshape(len(ids), 3)
    maps = np.array([np.full((4, 4), float(c)) for c in ids]).reshape(len(ids), 4, 4)
    g1 = C.gather_chain_values(ids, vals, nchains, dist)
    g2 = C.gather_chain_values(ids, maps, nchains, dist)
    t = torch.tensor([float(rank + 1)])
    dist.all_reduce(t, op=dist.ReduceOp.MAX)             # the max-over-ranks timing reduction bench.py uses
    # MAP_marg's mean field: simulations i mod world == rank contribute locally, one all_reduce gives the sum over all sims
    local = sum(torch.full((1, 1, 4, 3), complex(i, -2 * i), dtype=torch.complex128) for i in range(7) if i % world == rank)
    tot = C.allreduce_sum(local, dist)
    assert torch.allclose(tot, torch.full((1, 1, 4, 3), complex(21, -42), dtype=torch.complex128))
    q.put((rank, ids, g1, g2, float(t.item())))
    dist.barrier()
    dist.destroy_process_group()


def test_partition_and_gather_world2():
    world, nchains = 2, 5                                  # ragged: rank 0 owns 3 chains, rank 1 owns 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    ps = [ctx.Process(target=_worker, args=(r, world, port, nchains, q)) for r in range(world)]
    [p.start() for p in ps]
    res = [q.get(timeout=120) for _ in ps]
    [p.join(timeout=60) for p in ps]
    assert all(p.exitcode == 0 for p in ps)
    owned = sorted(sum((r[1] for r in res), []))
    assert owned == list(range(nchains))                   # every chain exactly once
    for rank, ids, g1, g2, tmax in res:
        assert tmax == world
        np.testing.assert_array_equal(g1[:, 0], 100 + np.arange(nchains))
        np.testing.assert_array_equal(g1[:, 1], np.arange(nchains) ** 2)
        for c in range(nchains):
            assert np.all(g2[c] == c)


def test_single_process_gather():
    import cmblensing_jl_amd as C
    assert C.partition_chains(8, 8, 3) == [3] and C.partition_chains(3, 8, 5) == []
    x = torch.ones(2, dtype=torch.complex64)
    assert C.allreduce_sum(x, None) is x
    out = C.gather_chain_values([0, 1, 2], np.arange(3.0), 3, None)
    np.testing.assert_array_equal(out[:, 0], np.arange(3.0))
