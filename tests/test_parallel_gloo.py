"""world_size-2 CPU test (gloo) of the multi-GPU layer: sequence sharding and the memory-bank all-gather."""
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


def _worker(rank, world, port, ret):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from estdepth_amd import parallel
    assert parallel.shard_sequences(5) == [s for s in range(5) if s % world == rank]
    g = torch.Generator().manual_seed(100 + rank)
    key = torch.randn(1, 16, 4, 3, 5, generator=g)
    value = torch.randn(1, 16, 4, 3, 5, generator=g)
    pose = torch.randn(1, 4, 4, generator=g)
    bank = parallel.allgather_memory_bank({"keys": [key], "values": [value]}, [pose])
    ok = len(bank) == world
    for r in range(world):
        gr = torch.Generator().manual_seed(100 + r)
        k = torch.randn(1, 16, 4, 3, 5, generator=gr)
        v = torch.randn(1, 16, 4, 3, 5, generator=gr)
        p = torch.randn(1, 4, 4, generator=gr)
        ok = ok and torch.equal(bank[r][0]["keys"][0], k) and torch.equal(bank[r][0]["values"][0], v) and torch.equal(bank[r][1][0], p)
    # channels-last (internal kv record) flavour: bit-exact round trip as views
    from estdepth_amd.hybrid_depth_decoder import kv_views
    kv = torch.randn(4, 3, 5, 32, generator=torch.Generator().manual_seed(200 + rank))
    k2, v2 = kv_views(kv)
    bank2 = parallel.allgather_memory_bank({"keys": [k2], "values": [v2]}, [pose])
    for r in range(world):
        ref = torch.randn(4, 3, 5, 32, generator=torch.Generator().manual_seed(200 + r))
        rk, rv = kv_views(ref)
        ok = ok and torch.equal(bank2[r][0]["keys"][0], rk) and torch.equal(bank2[r][0]["values"][0], rv)
        ok = ok and bank2[r][0]["keys"][0].shape == (1, 16, 4, 3, 5)
    # asynchronous variant (what bench.py overlaps with the next step)
    pend = parallel.allgather_memory_bank_async({"keys": [k2], "values": [v2]}, [pose])
    kv.add_(1.0)                                             # the source may be overwritten while the collective is in flight
    bank3 = pend.wait()
    for r in range(world):
        ref = torch.randn(4, 3, 5, 32, generator=torch.Generator().manual_seed(200 + r))
        ok = ok and torch.equal(bank3[r][0]["values"][0], kv_views(ref)[1])
    # no-staging variant (the caller keeps the source untouched until wait(): GraphedForward's fresh memory tensors)
    kv4 = torch.randn(4, 3, 5, 32, generator=torch.Generator().manual_seed(300 + rank))
    k4, v4 = kv_views(kv4)
    pose4 = torch.randn(1, 4, 4, generator=torch.Generator().manual_seed(400 + rank))
    bank4 = parallel.allgather_memory_bank_async({"keys": [k4], "values": [v4]}, [pose4], stage=False).wait()
    ok = ok and len(bank4) == world
    for r in range(world):
        ref = torch.randn(4, 3, 5, 32, generator=torch.Generator().manual_seed(300 + r))
        rk, rv = kv_views(ref)
        rp = torch.randn(1, 4, 4, generator=torch.Generator().manual_seed(400 + r))
        ok = ok and torch.equal(bank4[r][0]["keys"][0], rk) and torch.equal(bank4[r][0]["values"][0], rv) and torch.equal(bank4[r][1][0], rp)
    ret[rank] = bool(ok)
    dist.destroy_process_group()


def test_allgather_memory_bank_world2():
    world = 2
    mgr = mp.Manager()
    ret = mgr.dict()
    mp.spawn(_worker, args=(world, _free_port(), ret), nprocs=world, join=True)
    assert all(ret.get(r, False) for r in range(world)), dict(ret)
