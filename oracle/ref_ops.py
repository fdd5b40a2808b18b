"""Numpy front-end of the C oracle, with the reference's operator signatures.

ORACLE = test infrastructure only (see oracle/__init__.py).  Arrays are float32 numpy,
layouts are the reference's (NCHW / NCDHW).  Small camera matrices are computed with
LAPACK in fp32 (numpy.linalg.inv), the same routine family torch.inverse uses on CPU.
"""
import ctypes
import numpy as np

from . import build as _build

_lib = None
_f = ctypes.POINTER(ctypes.c_float)


def lib():
    global _lib
    if _lib is None:
        _lib = ctypes.CDLL(_build.build())
        _lib.orc_num_threads.restype = ctypes.c_int
    return _lib


def num_threads():
    return lib().orc_num_threads()


def set_num_threads(n):
    lib().orc_set_num_threads(int(n))


def _c(a):
    a = np.ascontiguousarray(a, dtype=np.float32)
    return a, a.ctypes.data_as(_f)


try:                      # the reference's own third-party arithmetic for the tiny camera matrices
    import torch as _torch
except ImportError:       # numpy-only environment: same LAPACK family, last-bit differences possible
    _torch = None


def inv(m):
    """torch.inverse on fp32, as called by the reference (model_hybrid.py:74,:83; homo_utils.py:51,:258,:469;
    hybrid_depth_decoder.py:235): ATen's CPU LAPACK path itself when torch is importable -- numpy's LAPACK build gives results
    that differ in the last bits, which is enough to flip samples across the |norm| > 1 masks."""
    m = np.asarray(m, dtype=np.float32)
    if _torch is not None:
        t = _torch.from_numpy(np.ascontiguousarray(m))
        return (_torch.inverse(t[None])[0] if t.dim() == 2 else _torch.inverse(t)).numpy()
    return np.linalg.inv(m).astype(np.float32)


def matmul(a, b):
    """torch.matmul of the small fp32 camera matrices as the reference calls it: on tensors WITH the batch dimension
    ([B,4,4] @ [B,4,4], [B,3,3] @ [B,3,4]), i.e. ATen's bmm path -- its small-matrix kernel and the 2-D mm kernel round
    differently, so 2-D inputs are given a batch axis of 1 here."""
    a, b = np.asarray(a, np.float32), np.asarray(b, np.float32)
    if _torch is not None:
        ta, tb = _torch.from_numpy(np.ascontiguousarray(a)), _torch.from_numpy(np.ascontiguousarray(b))
        if ta.dim() == 2:
            return _torch.matmul(ta[None], tb[None])[0].numpy()
        return _torch.matmul(ta, tb).numpy()
    return (a @ b).astype(np.float32)


def sweep_proj(cam_poses, cam_intr, ref, src):
    """proj = src_proj @ inverse(ref_proj) of get_costvolume + homo_warping (model_hybrid.py:74-88, homo_utils.py:469) for
    reference view ``ref`` and source view ``src``: cam_poses [B,V,4,4], cam_intr [B,3,3] -> [B,4,4].
    The chain is kept in torch tensors end to end when torch is importable: ``clone()`` of an inverse keeps LAPACK's
    column-major strides and torch.inverse rounds differently on the two layouts, so a numpy round trip in the middle
    would change last bits (and with them which samples fall across the |norm| > 1 mask)."""
    cam_poses, cam_intr = np.asarray(cam_poses, np.float32), np.asarray(cam_intr, np.float32)
    if _torch is None:
        out = []
        for b in range(cam_poses.shape[0]):
            ref_e, src_e = inv(cam_poses[b, ref]), inv(cam_poses[b, src])
            sp, rp = src_e.copy(), ref_e.copy()
            sp[:3, :4] = cam_intr[b] @ src_e[:3, :4]
            rp[:3, :4] = cam_intr[b] @ ref_e[:3, :4]
            out.append((sp @ inv(rp)).astype(np.float32))
        return np.stack(out)
    poses, K = _torch.from_numpy(np.ascontiguousarray(cam_poses)), _torch.from_numpy(np.ascontiguousarray(cam_intr))
    ref_extrinsic = _torch.inverse(poses[:, ref, :, :])                          # :74
    src_extrinsic = _torch.inverse(poses[:, src, :, :])                          # :83
    src_proj, ref_proj = src_extrinsic.clone(), ref_extrinsic.clone()            # :85-86
    src_proj[:, :3, :4] = _torch.matmul(K, src_extrinsic[:, :3, :4])             # :87
    ref_proj[:, :3, :4] = _torch.matmul(K, ref_extrinsic[:, :3, :4])             # :88
    return _torch.matmul(src_proj, _torch.inverse(ref_proj)).numpy()             # homo_utils.py:469


# ----------------------------------------------------------------------------- ops
def homo_warping_proj(src_fea, proj, depth_values):
    """utils/homo_utils.py:470-504 given proj = src_proj @ inverse(ref_proj) [B,4,4]; depth_values [B,D], [B,D,1,1] or per-pixel
    hypotheses [B,D,H,W] (:462)."""
    src_fea = np.asarray(src_fea, np.float32)
    B, C, H, W = src_fea.shape
    depth_values = np.asarray(depth_values, np.float32)
    D = depth_values.shape[1]
    dv = depth_values.reshape(B, -1)
    per_pixel = int(dv.shape[1] == D * H * W and H * W > 1)
    out = np.empty((B, C, D, H, W), np.float32)
    for b in range(B):
        rot, rp = _c(proj[b][:3, :3])
        tr, tp = _c(proj[b][:3, 3])
        s, sp = _c(src_fea[b])
        d, dp = _c(dv[b])
        o = out[b]
        lib().orc_homo_warping(sp, rp, tp, dp, per_pixel, C, H, W, D, o.ctypes.data_as(_f))
    return out


def homo_warping(src_fea, src_proj, ref_proj, depth_values):
    """utils/homo_utils.py:458-504.  src_fea [B,C,H,W]; *_proj [B,4,4]; depth_values [B,D,1,1] or [B,D]."""
    proj = matmul(np.asarray(src_proj, np.float32), inv(np.asarray(ref_proj, np.float32)))   # :469 (batched, like the reference)
    return homo_warping_proj(src_fea, proj, depth_values)


def set_id_grid(h, w):
    """utils/homo_utils.py:7-14 -> [1,3,H,W] (x, y, 1)."""
    j = np.broadcast_to(np.arange(w, dtype=np.float32)[None, :], (h, w))
    i = np.broadcast_to(np.arange(h, dtype=np.float32)[:, None], (h, w))
    return np.stack([j, i, np.ones((h, w), np.float32)], 0)[None]


def warp_volume(feat_volume, depth, pose, cam_intr, pixel_coords, depth_min, depth_interval, padding_mode="zeros", padding_value=0.0,
                disp_min=None, disp_interval=None):
    """utils/homo_utils.py:240-279 (trilinear; padding zeros or border + padding value; depth or disparity planes).
    feat_volume [N,C,D,H,W]; depth [N,1,D,H*W] (per voxel); pose [N,4,4]; cam_intr [N,3,3].  pixel_coords is the reference's
    cached (x,y,1) grid (set_id_grid); the restatement regenerates it from indices."""
    assert padding_mode in ("zeros", "border")
    feat_volume = np.asarray(feat_volume, np.float32)
    N, C, D, H, W = feat_volume.shape
    depth = np.asarray(depth, np.float32).reshape(N, D, H * W)
    out = np.empty_like(feat_volume)
    for b in range(N):
        kinv, kinvp = _c(inv(cam_intr[b]))          # :51
        m, mp = _c(inv(pose[b]))                    # :258
        k, kp = _c(cam_intr[b])
        v, vp = _c(feat_volume[b])
        dd, ddp = _c(depth[b])
        lib().orc_warp_volume(vp, ddp, kinvp, mp, kp, ctypes.c_float(depth_min), ctypes.c_float(depth_interval),
                              int(disp_min is not None), ctypes.c_float(disp_min or 0.0), ctypes.c_float(disp_interval or 1.0),
                              int(padding_mode == "border"), ctypes.c_float(padding_value),
                              C, D, H, W, out[b].ctypes.data_as(_f))
    return out


def conv3d(x, weight, bias=None):
    """nn.Conv3d, stride 1, padding k//2.  x [B,Cin,D,H,W]; weight [Cout,Cin,k,k,k]."""
    x = np.asarray(x, np.float32)
    B, Cin, D, H, W = x.shape
    weight = np.asarray(weight, np.float32)
    Cout, _, k = weight.shape[:3]
    assert Cout <= 64
    out = np.empty((B, Cout, D, H, W), np.float32)
    w, wp = _c(weight)
    bp = None
    if bias is not None:
        bb, bp = _c(bias)
    for b in range(B):
        xx, xp = _c(x[b])
        lib().orc_conv3d(xp, wp, bp, Cin, Cout, k, D, H, W, out[b].ctypes.data_as(_f))
    return out


def bn_act(x, bn, act="none", eps=1e-5):
    """BatchNorm3d in eval mode (+ReLU/Tanh).  bn = (weight, bias, running_mean, running_var)."""
    x = np.array(x, np.float32, copy=True, order="C")
    B, C = x.shape[:2]
    N = int(np.prod(x.shape[2:]))
    g, gp = _c(bn[0]); b_, bp = _c(bn[1]); m, mp = _c(bn[2]); v, vp = _c(bn[3])
    code = {"none": 0, "relu": 1, "tanh": 2}[act]
    for b in range(B):
        lib().orc_bn_act(x[b].ctypes.data_as(_f), gp, bp, mp, vp, ctypes.c_float(eps), code, C, ctypes.c_long(N))
    return x


def groupnorm1(x, weight, bias, eps=1e-5):
    """nn.GroupNorm(1, C, eps, affine) on [B,C,...]."""
    x = np.ascontiguousarray(x, np.float32)
    B, C = x.shape[:2]
    N = int(np.prod(x.shape[2:]))
    out = np.empty_like(x)
    g, gp = _c(weight); b_, bp = _c(bias)
    for b in range(B):
        lib().orc_groupnorm1(x[b].ctypes.data_as(_f), gp, bp, ctypes.c_float(eps), C, ctypes.c_long(N),
                             out[b].ctypes.data_as(_f))
    return out


def epipolar_attention(target_key, warped_keys, warped_values):
    """transformer/epipolar_transformer.py:62-73.  Returns h [B,C,D,H,W]."""
    tk = np.ascontiguousarray(target_key, np.float32)
    B, C = tk.shape[:2]
    N = int(np.prod(tk.shape[2:]))
    nv = len(warped_keys)
    wk = np.ascontiguousarray(np.stack(warped_keys, 1), np.float32)   # [B,nv,C,...]
    wv = np.ascontiguousarray(np.stack(warped_values, 1), np.float32)
    out = np.empty_like(tk)
    for b in range(B):
        lib().orc_epipolar_attention(tk[b].ctypes.data_as(_f), wk[b].ctypes.data_as(_f), wv[b].ctypes.data_as(_f),
                                     nv, C, ctypes.c_long(N), out[b].ctypes.data_as(_f))
    return out


def depthlayer_upsampled(logits_lowres, depth_values, scale=4):
    """F.interpolate(logits, scale_factor=scale) (nearest) followed by depthlayer
    (hybrid_depth_decoder.py:33-38,:202-204).  logits [B,D,H,W]; depth_values [B,D,1,1].
    Returns depth, prob each [B,1,scale*H,scale*W]."""
    lg = np.ascontiguousarray(logits_lowres, np.float32)
    B, D, H, W = lg.shape
    dv = np.ascontiguousarray(np.asarray(depth_values, np.float32).reshape(B, D))
    depth = np.empty((B, 1, H * scale, W * scale), np.float32)
    prob = np.empty_like(depth)
    for b in range(B):
        lib().orc_depthlayer_up(lg[b].ctypes.data_as(_f), dv[b].ctypes.data_as(_f), D, H, W, scale,
                                depth[b].ctypes.data_as(_f), prob[b].ctypes.data_as(_f))
    return depth, prob


def sigmoid(x):
    x = np.asarray(x, np.float32)
    return (1.0 / (1.0 + np.exp(-x, dtype=np.float32))).astype(np.float32)
