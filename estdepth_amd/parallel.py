"""Multi-GPU layer: one process per GPU, one video stream per process.

The hot path itself shards at sequence granularity with NO data-path collective (the reference
cannot even batch sequences, Q15).  The one exchange step with a reference-side meaning is the
temporal-fusion memory bank: after a window, every rank all-gathers {key, fused value, pose} of
its stream so that any rank can continue any stream (and so the collective's bandwidth over xGMI
is exercised and reported).  Backend "nccl" is RCCL on ROCm; CPU tests use gloo.
"""
import torch
import torch.distributed as dist


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
            self.work.wait()            # the current stream waits for the collective; the host does not block
            self.work = None
        return _unpack_bank(self.recv, self.n, *self.meta)


def _unpack_bank(recv, n, world, kv_shape, key_shape, value_shape, pose_shape, channels_last):
    out = []
    for r in range(world):
        p = recv[r, n:].reshape(pose_shape)
        if channels_last:
            from .hybrid_depth_decoder import kv_views
            k, v = kv_views(recv[r, :n].reshape(kv_shape))
        else:
            v = recv[r, :n // 2].reshape(value_shape)
            k = recv[r, n // 2:n].reshape(key_shape)
        out.append(({"keys": [k], "values": [v]}, [p]))
    return out


def allgather_memory_bank_async(costs, cam_poses, group=None, stage=True):
    """Non-blocking variant: stages {K, V_fused, pose} into one send buffer (so the source may be overwritten by the
    next forward / graph replay) and starts ONE all-gather on the communication stream.  ``.wait()`` returns the bank.
    ``stage=False``: the caller guarantees that the memory it hands in is not written again before ``.wait()`` (the fresh
    tensors GraphedForward returns): the 157 MB record stream is sent from where it lies, the pose in a second, tiny
    all-gather -- no staging copy on the compute stream."""
    world = dist.get_world_size(group)
    key, value, pose = costs["keys"][0], costs["values"][0], cam_poses[0]
    kv = getattr(value, "_estd_kv", None)
    channels_last = kv is not None and getattr(key, "_estd_kv", None) is kv
    meta = (world, tuple(kv.shape) if channels_last else None, tuple(key.shape), tuple(value.shape), tuple(pose.shape), channels_last)
    if not stage and channels_last and kv.is_contiguous():
        flat = kv.reshape(-1)
        psend = pose.reshape(-1).to(flat.dtype).contiguous()
        n, m = flat.numel(), psend.numel()
        recv_kv = torch.empty(world * n, device=flat.device, dtype=flat.dtype)
        recv_p = torch.empty(world * m, device=flat.device, dtype=flat.dtype)
        work_p = dist.all_gather_into_tensor(recv_p, psend, group=group, async_op=True)
        work = dist.all_gather_into_tensor(recv_kv, flat, group=group, async_op=True)
        pend = _PendingBank2((work_p, work), recv_kv.view(world, n), recv_p.view(world, m), meta)
        pend._send = (flat, psend, kv)  # keep the source alive until the collective has run
        return pend
    flat = kv.reshape(-1) if channels_last else torch.cat([value.reshape(-1), key.reshape(-1)])
    send = torch.cat([flat, pose.reshape(-1).to(flat.dtype)])
    recv = torch.empty(world * send.numel(), device=send.device, dtype=send.dtype)
    work = dist.all_gather_into_tensor(recv, send, group=group, async_op=True)
    pend = _PendingBank(work, recv.view(world, send.numel()), flat.numel(), meta)
    pend._send = send                   # keep the staging buffer alive until the collective has run
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
        world, kv_shape, key_shape, value_shape, pose_shape, _ = self.meta
        from .hybrid_depth_decoder import kv_views
        out = []
        for r in range(world):
            k, v = kv_views(self.recv_kv[r].reshape(kv_shape))
            out.append(({"keys": [k], "values": [v]}, [self.recv_p[r].reshape(pose_shape)]))
        return out


def allgather_memory_bank(costs, cam_poses, group=None):
    """costs = {"keys": [K], "values": [V]} with K, V [1,16,D,H,W]; cam_poses = [pose [1,4,4]].
    Returns a list (one entry per rank) of (costs, cam_poses) in the same structure; entry[rank] aliases
    nothing of the input.  One fused buffer per rank -> a single all-gather (per-link bound on xGMI:
    prefer one large message over three small ones)."""
    world = dist.get_world_size(group)
    key, value, pose = costs["keys"][0], costs["values"][0], cam_poses[0]
    kv = getattr(value, "_estd_kv", None)
    if kv is not None and getattr(key, "_estd_kv", None) is kv:
        flat = kv.reshape(-1)                      # already one contiguous [D,H,W,32] record stream
        channels_last = True
    else:
        flat = torch.cat([value.reshape(-1), key.reshape(-1)])
        channels_last = False
    send = torch.cat([flat, pose.reshape(-1).to(flat.dtype)])
    recv = torch.empty(world * send.numel(), device=send.device, dtype=send.dtype)
    dist.all_gather_into_tensor(recv, send, group=group)
    recv = recv.view(world, send.numel())
    n = flat.numel()
    out = []
    for r in range(world):
        p = recv[r, n:].reshape(pose.shape)
        if channels_last:
            from .hybrid_depth_decoder import kv_views
            k, v = kv_views(recv[r, :n].reshape(kv.shape))
        else:
            v = recv[r, :n // 2].reshape(value.shape)
            k = recv[r, n // 2:n].reshape(key.shape)
        out.append(({"keys": [k], "values": [v]}, [p]))
    return out
