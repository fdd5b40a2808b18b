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
    # the record with the frame's initial logit volume (north_star: "all-gather of per-frame probability volumes"), both exchange
    # algorithms (ONE all-gather per stream | one send + one receive per peer under one group), staged and unstaged: bit-equal
    lg = torch.randn(4, 3, 5, generator=torch.Generator().manual_seed(500 + rank))
    for algo in ("collective", "direct"):
        for stage in (True, False):
            bank5 = parallel.allgather_memory_bank_async({"keys": [k4], "values": [v4]}, [pose4], stage=stage, logits=lg, algo=algo).wait()
            ok = ok and len(bank5) == world
            for r in range(world):
                ref = torch.randn(4, 3, 5, 32, generator=torch.Generator().manual_seed(300 + r))
                rk, rv = kv_views(ref)
                rp = torch.randn(1, 4, 4, generator=torch.Generator().manual_seed(400 + r))
                rl = torch.randn(4, 3, 5, generator=torch.Generator().manual_seed(500 + r))
                ok = ok and torch.equal(bank5[r][0]["keys"][0], rk) and torch.equal(bank5[r][0]["values"][0], rv)
                ok = ok and torch.equal(bank5[r][1][0], rp) and torch.equal(bank5[r][0]["logits"][0], rl)
        # NCDHW (foreign) tensors through the direct exchange as well
        bank6 = parallel.allgather_memory_bank({"keys": [key], "values": [value]}, [pose], logits=lg, algo=algo)
        for r in range(world):
            gr = torch.Generator().manual_seed(100 + r)
            k = torch.randn(1, 16, 4, 3, 5, generator=gr)
            v = torch.randn(1, 16, 4, 3, 5, generator=gr)
            p = torch.randn(1, 4, 4, generator=gr)
            rl = torch.randn(4, 3, 5, generator=torch.Generator().manual_seed(500 + r))
            ok = ok and torch.equal(bank6[r][0]["keys"][0], k) and torch.equal(bank6[r][0]["values"][0], v)
            ok = ok and torch.equal(bank6[r][1][0], p) and torch.equal(bank6[r][0]["logits"][0], rl)
    try:
        parallel.allgather_memory_bank({"keys": [key], "values": [value]}, [pose], algo="ring-of-fire")
        ok = False
    except RuntimeError:
        pass
    ret[rank] = bool(ok)
    dist.destroy_process_group()


def test_allgather_memory_bank_world2():
    world = 2
    mgr = mp.Manager()
    ret = mgr.dict()
    mp.spawn(_worker, args=(world, _free_port(), ret), nprocs=world, join=True)
    assert all(ret.get(r, False) for r in range(world)), dict(ret)


def _worker3(rank, world, port, ret):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from estdepth_amd import parallel
    from estdepth_amd.hybrid_depth_decoder import kv_views
    mk = lambda base, r, *shape: torch.randn(*shape, generator=torch.Generator().manual_seed(base + r))
    k, v = kv_views(mk(300, rank, 4, 3, 5, 32))
    ok = True
    for algo in ("direct", "collective"):
        bank = parallel.allgather_memory_bank_async({"keys": [k], "values": [v]}, [mk(400, rank, 1, 4, 4)], stage=False,
                                                    logits=mk(500, rank, 4, 3, 5), algo=algo).wait()
        for r in range(world):
            rk, rv = kv_views(mk(300, r, 4, 3, 5, 32))
            ok = ok and torch.equal(bank[r][0]["keys"][0], rk) and torch.equal(bank[r][0]["values"][0], rv)
            ok = ok and torch.equal(bank[r][1][0], mk(400, r, 1, 4, 4)) and torch.equal(bank[r][0]["logits"][0], mk(500, r, 4, 3, 5))
    # ESTD_AG_ALGO=auto: both algorithms timed on this communicator, every rank lands on the same choice, and that choice is what
    # an exchange with algo=None runs afterwards
    assert parallel.AG_ALGO == "auto" and parallel.active_algo() == "collective"
    sel = parallel.select_exchange_algo({"keys": [k], "values": [v]}, [mk(400, rank, 1, 4, 4)], logits=mk(500, rank, 4, 3, 5), reps=2)
    got = [None] * world
    dist.all_gather_object(got, (sel["chosen"], sel["ms_collective"], sel["ms_direct"]))
    ok = ok and all(g == got[0] for g in got) and sel["chosen"] in ("collective", "direct") and parallel.active_algo() == sel["chosen"]
    bank = parallel.allgather_memory_bank_async({"keys": [k], "values": [v]}, [mk(400, rank, 1, 4, 4)], stage=False).wait()
    for r in range(world):
        ok = ok and torch.equal(bank[r][0]["values"][0], kv_views(mk(300, r, 4, 3, 5, 32))[1])
    ret[rank] = bool(ok)
    dist.destroy_process_group()


def test_direct_exchange_world3():
    """three ranks: the rotated peer order of the direct exchange (rank + d sends, rank - d receives) lands every shard in its slot"""
    world = 3
    mgr = mp.Manager()
    ret = mgr.dict()
    mp.spawn(_worker3, args=(world, _free_port(), ret), nprocs=world, join=True)
    assert all(ret.get(r, False) for r in range(world)), dict(ret)
