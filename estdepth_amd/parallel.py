"""Multi-GPU layer: one process per GPU, one video stream per process.

The hot path itself shards at sequence granularity with NO data-path collective (the reference
cannot even batch sequences, Q15).  The one exchange step with a reference-side meaning is the
temporal-fusion memory bank: after a window, every rank all-gathers {key, fused value, pose} of
its stream -- and, when the caller hands them over, the frame's initial logit volume (the per-frame
probability volume of `north_star` before its softmax, hybrid_depth_decoder.py:200-204; 4.9 MB beside the
157 MB of K||V at cfg2/3 size) -- so that any rank can continue any stream (and so the collective's
bandwidth over xGMI is exercised and reported).  Backend "nccl" is RCCL on ROCm; CPU tests use gloo.

Algorithm of the exchange (``ESTD_AG_ALGO`` / ``algo=``):
  * ``auto`` (default): ``collective`` until the faster algorithm is known.  ``select_exchange_algo`` times both on the real
    communicator, agrees on the result across the ranks and switches; bench.py times the two itself (the untried one last and under a
    watchdog: the untried algorithm of a machine nobody has run yet must not cost the measurement), agrees across the ranks the same
    way and switches with ``set_active_algo``;
  * ``collective``: ONE ``all_gather_into_tensor`` per stream of the record (RCCL picks ring / direct itself);
  * ``direct``: every rank posts a send to and a receive from every peer under ONE group
    (``batch_isend_irecv`` = ncclGroupStart{ncclSend/ncclRecv to all peers}ncclGroupEnd on RCCL) and copies its own shard
    locally: on the fully connected xGMI mesh of an MI355X node every one of the 7 links of a GPU carries exactly one
    157 MB message (~1 ms at 153 GB/s per link) where a ring all-gather pushes (N-1) x 157 MB through ONE link per GPU
    (~7.2 ms at N = 8).  Same receive layout, bit-identical result.  Exercised on gloo (world 2 and 3) and on a
    world-size-1 RCCL communicator only: experimental until a multi-GPU RCCL run has been through it.
"""
import os

import torch
import torch.distributed as dist

AG_ALGO = os.environ.get("ESTD_AG_ALGO", "auto")
_ACTIVE = {"algo": "collective"}      # what "auto" resolves to (select_exchange_algo)


def set_active_algo(algo):
    """what ``ESTD_AG_ALGO=auto`` / ``algo=None`` resolves to from now on.  Every rank must make the same call (bench.py: after an
    all-reduce of the two exchange times)."""
    if algo not in ("collective", "direct"):
        raise RuntimeError("exchange algorithm must be collective or direct, got %r" % (algo,))
    _ACTIVE["algo"] = algo


def active_algo():
    """the algorithm an exchange with ``algo=None`` runs now"""
    return _ACTIVE["algo"] if AG_ALGO == "auto" else AG_ALGO


def select_exchange_algo(costs, cam_poses, group=None, logits=None, reps=3, margin=0.9, sync=None):
    """Time the memory-bank exchange alone with BOTH algorithms on this communicator (``reps`` exchanges each after one untimed
    one), agree on the result across the ranks (every rank's times are reduced with MAX, so all ranks see the same two numbers) and
    make the faster one what ``algo=None`` / ``ESTD_AG_ALGO=auto`` uses from now on -- ``direct`` only when it wins by more than
    ``1 - margin``.  ``sync``: device synchronisation (``torch.cuda.synchronize``; None on CPU / gloo).  Returns
    {"chosen", "ms_collective", "ms_direct"}.  Collective call: every rank must make it at the same point."""
    import time
    ms = {}
    for algo in ("collective", "direct"):
        allgather_memory_bank_async(costs, cam_poses, group=group, stage=False, logits=logits, algo=algo).wait()
        if sync:
            sync()
        dist.barrier(group)
        t0 = time.perf_counter()
        for _ in range(reps):
            allgather_memory_bank_async(costs, cam_poses, group=group, stage=False, logits=logits, algo=algo).wait()
        if sync:
            sync()
        ms[algo] = 1e3 * (time.perf_counter() - t0) / reps
    t = torch.tensor([ms["collective"], ms["direct"]], dtype=torch.float64, device=costs["keys"][0].device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX, group=group)
    mc, md = float(t[0]), float(t[1])
    set_active_algo("direct" if md < margin * mc else "collective")
    return {"chosen": _ACTIVE["algo"], "ms_collective": round(mc, 3), "ms_direct": round(md, 3)}


def _gather_flat(recv, send, group, algo, async_op=True):
    """recv [world * n] <- every rank's send [n].  Returns a list of work handles (possibly empty)."""
    algo = active_algo() if algo is None else algo
    if algo == "auto":
        algo = _ACTIVE["algo"]
    if algo not in ("collective", "direct"):
        raise RuntimeError("ESTD_AG_ALGO must be auto, collective or direct, got %r" % (algo,))
    world, rank = dist.get_world_size(group), dist.get_rank(group)
    if algo == "collective":
        w = dist.all_gather_into_tensor(recv, send, group=group, async_op=async_op)
        return [w] if async_op else []
    n = send.numel()
    rv = recv.view(world, n)
    ops_ = []
    for d in range(1, world):                          # peer order rotated by rank: at every position of the list the
        dst, src = (rank + d) % world, (rank - d) % world      # world posts a perfect matching (no link carries two messages at once)
        gdst = dist.get_global_rank(group, dst) if group is not None else dst
        gsrc = dist.get_global_rank(group, src) if group is not None else src
        ops_.append(dist.P2POp(dist.isend, send, gdst, group))
        ops_.append(dist.P2POp(dist.irecv, rv[src], gsrc, group))
    works = dist.batch_isend_irecv(ops_) if ops_ else []        # (world size 1: the own-shard copy is the whole exchange)
    rv[rank].copy_(send)                               # own shard: a device-local copy on the compute stream
    if not async_op:
        for w in works:
            w.wait()
        return []
    return list(works)


def shard_sequences(n_sequences, rank=None, world=None):
    """Round-robin assignment of independent sequences to ranks."""
    rank = dist.get_rank() if rank is None else rank
    world = dist.get_world_size() if world is None else world
    return [s for s in range(n_sequences) if s % world == rank]


class _PendingBank:
    """Handle of an in-flight memory-bank all-gather (runs on the communication stream, overlapping compute)."""

    def __init__(self, work, recv, n, meta):
        self.work, self.recv, self.n, self.meta = work, recv, n, meta

    def wait(self):
        if self.work is not None:
            for w in self.work:         # the current stream waits for the exchange; the host does not block
                w.wait()
            self.work = None
        return _unpack_bank(self.recv, self.n, *self.meta)


def _unpack_bank(recv, n, world, kv_shape, key_shape, value_shape, pose_shape, channels_last, logits_shape=None):
    out = []
    npose = 1
    for d in pose_shape:
        npose *= d
    for r in range(world):
        p = recv[r, n:n + npose].reshape(pose_shape)
        if channels_last:
            from .hybrid_depth_decoder import kv_views
            k, v = kv_views(recv[r, :n].reshape(kv_shape))
        else:
            v = recv[r, :n // 2].reshape(value_shape)
            k = recv[r, n // 2:n].reshape(key_shape)
        costs = {"keys": [k], "values": [v]}
        if logits_shape is not None:
            costs["logits"] = [recv[r, n + npose:].reshape(logits_shape)]
        out.append((costs, [p]))
    return out


def allgather_memory_bank_async(costs, cam_poses, group=None, stage=True, logits=None, algo=None):
    """Non-blocking variant: stages {K, V_fused, pose[, init logits]} into one send buffer (so the source may be overwritten by
    the next forward / graph replay) and starts ONE exchange on the communication stream.  ``.wait()`` returns the bank.
    ``stage=False``: the caller guarantees that the memory it hands in is not written again before ``.wait()`` (the fresh
    tensors GraphedForward returns): the 157 MB record stream is sent from where it lies, the pose (and the 4.9 MB logit volume)
    in a second, small exchange -- no staging copy of the records on the compute stream.
    ``logits``: the frame's initial logit volume [D,H,W] (``DepthHybridDecoder.memory_logits`` / ``GraphedForward.memory_logits``):
    every bank entry then carries ``costs["logits"] = [volume]`` as well.  ``algo``: see the module docstring."""
    world = dist.get_world_size(group)
    key, value, pose = costs["keys"][0], costs["values"][0], cam_poses[0]
    kv = getattr(value, "_estd_kv", None)
    channels_last = kv is not None and getattr(key, "_estd_kv", None) is kv
    lshape = tuple(logits.shape) if logits is not None else None
    meta = (world, tuple(kv.shape) if channels_last else None, tuple(key.shape), tuple(value.shape), tuple(pose.shape), channels_last, lshape)
    if not stage and channels_last and kv.is_contiguous():
        flat = kv.reshape(-1)
        small = [pose.reshape(-1).to(flat.dtype)] + ([logits.reshape(-1).to(flat.dtype)] if logits is not None else [])
        psend = torch.cat(small).contiguous() if len(small) > 1 else small[0].contiguous()
        n, m = flat.numel(), psend.numel()
        recv_kv = torch.empty(world * n, device=flat.device, dtype=flat.dtype)
        recv_p = torch.empty(world * m, device=flat.device, dtype=flat.dtype)
        works = _gather_flat(recv_p, psend, group, algo) + _gather_flat(recv_kv, flat, group, algo)
        pend = _PendingBank2(works, recv_kv.view(world, n), recv_p.view(world, m), meta)
        pend._send = (flat, psend, kv)  # keep the source alive until the exchange has run
        return pend
    flat = kv.reshape(-1) if channels_last else torch.cat([value.reshape(-1), key.reshape(-1)])
    send = torch.cat([flat, pose.reshape(-1).to(flat.dtype)] + ([logits.reshape(-1).to(flat.dtype)] if logits is not None else []))
    recv = torch.empty(world * send.numel(), device=send.device, dtype=send.dtype)
    works = _gather_flat(recv, send, group, algo)
    pend = _PendingBank(works, recv.view(world, send.numel()), flat.numel(), meta)
    pend._send = send                   # keep the staging buffer alive until the exchange has run
    return pend


class _PendingBank2:
    """in-flight memory-bank all-gather without a staging copy: records and poses arrive in two receive buffers."""

    def __init__(self, works, recv_kv, recv_p, meta):
        self.work, self.recv_kv, self.recv_p, self.meta = works, recv_kv, recv_p, meta

    def wait(self):
        if self.work is not None:
            # BOTH handles: RCCL completes the collectives of one communicator in order on one stream, but gloo runs async
            # works on several threads with no completion order (and only a work's own wait() orders its copy-back with
            # the current stream).  Waiting on the pose gather as well costs nothing on RCCL and is correct everywhere.
            for w in self.work:
                w.wait()
            self.work = None
        world, kv_shape, key_shape, value_shape, pose_shape, _, logits_shape = self.meta
        from .hybrid_depth_decoder import kv_views
        npose = 1
        for d in pose_shape:
            npose *= d
        out = []
        for r in range(world):
            k, v = kv_views(self.recv_kv[r].reshape(kv_shape))
            costs = {"keys": [k], "values": [v]}
            if logits_shape is not None:
                costs["logits"] = [self.recv_p[r, npose:].reshape(logits_shape)]
            out.append((costs, [self.recv_p[r, :npose].reshape(pose_shape)]))
        return out


def allgather_memory_bank(costs, cam_poses, group=None, logits=None, algo=None):
    """costs = {"keys": [K], "values": [V]} with K, V [1,16,D,H,W]; cam_poses = [pose [1,4,4]]; optional ``logits`` [D,H,W].
    Returns a list (one entry per rank) of (costs, cam_poses) in the same structure (+ ``costs["logits"]`` when logits were
    handed in); entry[rank] aliases nothing of the input.  One fused buffer per rank -> a single exchange (per-link bound on
    xGMI: prefer one large message over three small ones)."""
    return allgather_memory_bank_async(costs, cam_poses, group=group, stage=True, logits=logits, algo=algo).wait()
