"""Camera algebra of the hot path -- a few dozen 4x4 / 3x3 matrix inverses and products per forward.

These tiny matrices decide on which side of the ``|normalised coordinate| > 1 -> 2`` masks (homo_utils.py:488-491,
:192-197) every sample of the plane sweep and of the volume warps falls.  The masks are discontinuous: a relative
difference of 1e-5 in a projection matrix (what an fp64 device evaluation differs by from the reference's fp32 LAPACK
composition) flips isolated boundary samples, and one flipped voxel of a cost volume moves ~2000 depth pixels by up to
2e-3 m after the 3D convolutions (measured at BASELINE configs[4] size, tools/parity_diag.py).  So the default evaluates
them ON THE HOST with the very torch-CPU calls the reference makes -- same ATen kernels, same shapes (the batched [1,4,4]
forms: ATen's bmm and mm kernels round differently), same association order -- and hands the kernels bit-identical
matrices; the per-voxel arithmetic in the kernels then reproduces the reference's rounding sequence op for op
(csrc/plane_sweep.hip::sweep_coords, csrc/est_fusion.hip::volume_coords_base).

    model_hybrid.py:74-88       extrinsic = inverse(pose);  proj[:, :3, :4] = K @ extrinsic[:, :3, :4]
    homo_utils.py:469-471       proj = src_proj @ inverse(ref_proj);  rot | trans
    hybrid_depth_decoder.py:235 rel = pose_j @ inverse(pose_i)                       
    homo_utils.py:258, :51      inverse(rel), inverse(K)

Cost: one small D2H copy of the poses, ~40 ATen CPU calls (one C++ operator), one H2D copy.  The copy is queued before the 2D
networks of the forward and awaited after they are launched (begin() / finish()), so the host work overlaps GPU work.  ``mode="device"`` keeps everything on the GPU (estd_cam_* kernels, fp64
Gauss-Jordan, no synchronisation) for latency-critical eager callers that accept the boundary-sample caveat above.
"""
import torch

from . import ops


def _cpu(t):
    return t.detach().to(device="cpu", dtype=torch.float32)


def _sweep_set(poses, K, ref, srcs):
    """rot(9) | trans(3) of the plane-sweep homographies from reference view ``ref`` into the views ``srcs`` -> [len(srcs), 12]."""
    def view_proj(v):
        extrinsic = torch.inverse(poses[:, v, :, :])                            # model_hybrid.py:74,:83
        proj = extrinsic.clone()
        proj[:, :3, :4] = torch.matmul(K, extrinsic[:, :3, :4])                 # :87-88
        return proj
    ref_inv = torch.inverse(view_proj(ref))
    out = torch.empty(len(srcs), 12, dtype=torch.float32)
    for k, s in enumerate(srcs):
        pr = torch.matmul(view_proj(s), ref_inv)                                # homo_utils.py:469
        out[k, :9] = pr[0, :3, :3].reshape(-1)                                  # :470
        out[k, 9:] = pr[0, :3, 3]                                               # :471
    return out


def sweep_projection_set(cam_poses, cam_intr_q, ref, srcs, device):
    """get_costvolume() of one reference view: cam_poses [1,V,4,4], cam_intr_q [1,3,3] (1/4 scale) -> [len(srcs), 12]."""
    return _sweep_set(_cpu(cam_poses), _cpu(cam_intr_q), ref, list(srcs)).to(device)


def sweep_projections(cam_poses, cam_intr_q, device):
    """cam_poses [1,V,4,4] camera-to-world, cam_intr_q [1,3,3] at 1/4 scale  ->  [V-2, 2, 12] on ``device``:
    the homography sweep of target t (= view t+1) from its two sources (views t and t+2), model_hybrid.py:152-156."""
    poses, K = _cpu(cam_poses), _cpu(cam_intr_q)
    return torch.stack([_sweep_set(poses, K, t + 1, (t, t + 2)) for t in range(poses.shape[1] - 2)]).to(device)


def pair_projection(src_proj, ref_proj, device):
    """level-1 homo_warping(): one batch element, src_proj / ref_proj [4,4] -> [12] on ``device``."""
    pr = torch.matmul(_cpu(src_proj)[None], torch.inverse(_cpu(ref_proj)[None]))
    return torch.cat([pr[0, :3, :3].reshape(-1), pr[0, :3, 3]]).to(device)


def volume_matrices(poses, n_targets, cam_intr_q, device):
    """poses: list of [1,4,4] (targets first, then memory poses); -> [n_targets, len(poses)-1, 30] on ``device``:
    per target i and other view j (ascending j, i skipped): inverse(K)(9) | inverse(pose_j @ inverse(pose_i))[:3](12) | K(9)."""
    P = [_cpu(p).reshape(1, 4, 4) for p in poses]
    K = _cpu(cam_intr_q).reshape(1, 3, 3)
    kinv = torch.inverse(K)                                                      # homo_utils.py:51
    n = len(P)
    out = torch.empty(n_targets, n - 1, 30, dtype=torch.float32)
    for i in range(n_targets):
        inv_i = torch.inverse(P[i])
        r = 0
        for j in range(n):
            if j == i:
                continue
            rel = torch.matmul(P[j], inv_i)                                      # hybrid_depth_decoder.py:235 (Q8)
            m = torch.inverse(rel)                                               # homo_utils.py:258
            out[i, r, :9] = kinv[0].reshape(-1)
            out[i, r, 9:21] = m[0, :3, :].reshape(-1)
            out[i, r, 21:] = K[0].reshape(-1)
            r += 1
    return out.to(device)


def relative_volume_matrix(rel_pose, cam_intr_q, device):
    """level-1 warp_volume(): rel_pose [4,4] is already the relative pose -> [30] on ``device``."""
    K = _cpu(cam_intr_q).reshape(1, 3, 3)
    m = torch.inverse(_cpu(rel_pose).reshape(1, 4, 4))
    return torch.cat([torch.inverse(K)[0].reshape(-1), m[0, :3, :].reshape(-1), K[0].reshape(-1)]).to(device)


class _PinnedRing:
    """a few page-locked staging buffers per size, reused round-robin: asynchronous copies need pinned memory, and a buffer
    must not be rewritten while a copy that reads it may still be queued (4 forwards deep is far beyond what can be in flight)."""

    def __init__(self, depth=4):
        self.depth, self.bufs, self.next = depth, {}, {}

    def get(self, n):
        ring = self.bufs.setdefault(n, [])
        if len(ring) < self.depth:
            ring.append(torch.empty(n, dtype=torch.float32).pin_memory())
            return ring[-1]
        i = self.next.get(n, 0)
        self.next[n] = (i + 1) % self.depth
        return ring[i]


_pinned = _PinnedRing()


class _Pending:
    __slots__ = ("flat_cpu", "event", "V", "n_pre", "with_volume", "device")


def begin(cam_poses, cam_intr_q, pre_poses, with_volume, device):
    """Queue ONE asynchronous device-to-host copy of the poses / intrinsics / memory poses of a forward and return a handle
    for finish().  Called BEFORE the 2D networks are launched: the copy only waits for work that was queued earlier, and the
    host meets its completion event while the GPU is busy with the 2D networks (CPU tensors: nothing to wait for)."""
    pre = list(pre_poses) if (with_volume and pre_poses is not None) else []
    parts = [cam_poses.reshape(-1).float(), cam_intr_q.reshape(-1).float()] + [p.reshape(-1).float() for p in pre]
    p = _Pending()
    p.V, p.n_pre, p.with_volume, p.device = cam_poses.shape[1], len(pre), bool(with_volume), device
    if all(t.is_cuda for t in parts):
        flat = torch.cat(parts).detach()
        p.flat_cpu = _pinned.get(flat.numel())
        p.flat_cpu.copy_(flat, non_blocking=True)
        p.event = torch.cuda.Event()
        p.event.record()
    else:
        p.flat_cpu, p.event = torch.cat([t.detach().cpu() for t in parts]), None
    return p


def finish(p):
    """{"sweep": [T,2,12], "vol": [T,n,30] or None} on the device.  One C++ call -- estdepth_hip::camera_matrices_host
    (csrc/torch_ops.cpp) makes the same ATen CPU calls as sweep_projections() / volume_matrices() above (bit-identical,
    tests/test_camera_host.py) without ~40 trips through the Python dispatcher -- and one asynchronous host-to-device copy."""
    if p.event is not None:
        p.event.synchronize()
    flat, V = p.flat_cpu, p.V
    poses = flat[:V * 16].reshape(1, V, 4, 4)
    K = flat[V * 16:V * 16 + 9].reshape(1, 3, 3)
    pre_cpu = [flat[V * 16 + 9 + 16 * i:V * 16 + 25 + 16 * i].reshape(1, 4, 4) for i in range(p.n_pre)]
    if ops.BINDING == "torch":
        sweep, vol = ops.T().camera_matrices_host(poses, K, pre_cpu, p.with_volume)
    else:
        # torch-free binding (ESTD_BINDING=ctypes): the same ATen CPU calls from Python -- bit-identical to the C++ operator
        # (tests/test_camera_host.py), ~40 dispatcher trips slower, and libestd_torch_ops.so is not needed
        sweep = torch.stack([_sweep_set(poses, K, t + 1, (t, t + 2)) for t in range(V - 2)])
        if p.with_volume:
            plist = [poses[:, t + 1] for t in range(V - 2)] + pre_cpu
            vol = volume_matrices(plist, V - 2, K, "cpu")
        else:
            vol = torch.empty(0, dtype=torch.float32)
    n_sweep, n = sweep.numel(), sweep.numel() + vol.numel()
    if torch.device(p.device).type == "cuda":
        stage = _pinned.get(n)
        stage[:n_sweep].copy_(sweep.reshape(-1))
        stage[n_sweep:].copy_(vol.reshape(-1))
        both = stage.to(p.device, non_blocking=True)
    else:
        both = torch.cat([sweep.reshape(-1), vol.reshape(-1)])
    return {"sweep": both[:n_sweep].reshape(sweep.shape), "vol": both[n_sweep:].reshape(vol.shape) if p.with_volume else None}


def forward_matrices(cam_poses, cam_intr_q, pre_poses, with_volume, device):
    """begin() + finish() back to back (callers with nothing to overlap)."""
    return finish(begin(cam_poses, cam_intr_q, pre_poses, with_volume, device))


# ------------------------------------------------------------------------------------------------ device variants
def sweep_projections_device(cam_poses, cam_intr_q):
    poses, K = cam_poses[0].contiguous().float(), cam_intr_q[0].contiguous().float()
    V = poses.shape[0]
    return torch.stack([torch.stack([ops.cam_sweep_proj(poses[t + 1], poses[s], K) for s in (t, t + 2)]) for t in range(V - 2)])


def volume_matrices_device(poses, n_targets, cam_intr_q):
    P = [p.reshape(4, 4).contiguous().float() for p in poses]
    K = cam_intr_q.reshape(3, 3).contiguous().float()
    out = torch.empty((n_targets, len(P) - 1, 30), device=P[0].device, dtype=torch.float32)
    for i in range(n_targets):
        for r, j in enumerate([j for j in range(len(P)) if j != i]):
            ops.cam_volume_mats(P[j], P[i], K, out=out[i, r])
    return out
