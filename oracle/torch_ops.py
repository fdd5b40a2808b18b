"""The same operator surface as oracle/ref_ops.py, evaluated with torch's own CPU operators (ATen: oneDNN convolutions,
grid_sampler_2d / grid_sampler_3d, native group_norm / softmax) instead of the C restatement.

ORACLE = test infrastructure only (see oracle/__init__.py).  Purpose: the SECOND cpu_baseline leg of bench.py
(``kind: "torch-ops"``, SURVEY.md section 8(d) "How the reference CPU path is timed": the build's own restatement with the
same torch ops the reference calls, at 8 threads and at all threads) -- the reference's CPU path is oneDNN-bound
(``mkldnn_convolution`` 43 % of its time) and scales with the cores, which the naive C/OpenMP
convolution of estd_oracle.c does not.  Written from the operator semantics recorded in SURVEY.md section 8(a) / Appendix B,
each function citing the reference lines it restates; ``oracle.ref_model.use_ops(torch_ops)`` swaps it in under the
unchanged composition of ref_model.py.  Checked against the C oracle in tests/test_oracle_torch_ops.py.

Arrays in and out are float32 numpy (zero-copy views of the torch tensors), layouts NCHW / NCDHW.
"""
import numpy as np
import torch
import torch.nn.functional as F

from .ref_ops import inv, matmul, sweep_proj, set_id_grid      # the tiny camera matrices: the reference's own ATen calls already

_f32 = np.float32


def num_threads():
    return torch.get_num_threads()


def set_num_threads(n):
    torch.set_num_threads(int(n))


def _t(a):
    return torch.from_numpy(np.ascontiguousarray(a, dtype=_f32))


def homo_warping_proj(src_fea, proj, depth_values):
    """utils/homo_utils.py:470-504 given proj = src_proj @ inverse(ref_proj) [B,4,4]: q = (R [x,y,1]^T) d + t (:479-482),
    p = q.xy / (q.z + 1e-8) (:483), normalised by (size - 1) / 2 (:484-485), |n| > 1 -> 2 (:488-491), bilinear grid_sample with
    zero padding and align_corners=False (:499-501).  depth_values [B,D], [B,D,1,1] or per-pixel [B,D,H,W]."""
    with torch.no_grad():
        x = _t(src_fea)
        B, C, H, W = x.shape
        P = _t(proj)
        dv = _t(depth_values)
        D = dv.shape[1]
        rot, trans = P[:, :3, :3], P[:, :3, 3:4]
        yy, xx = torch.meshgrid(torch.arange(H, dtype=torch.float32), torch.arange(W, dtype=torch.float32), indexing="ij")
        xyz = torch.stack((xx.reshape(-1), yy.reshape(-1), torch.ones(H * W)))[None].repeat(B, 1, 1)       # [B,3,HW]
        rot_xyz = torch.matmul(rot, xyz)
        if dv.numel() == B * D * H * W and H * W > 1:
            d = dv.reshape(B, 1, D, H * W)
        else:
            d = dv.reshape(B, 1, D, 1)
        q = rot_xyz[:, :, None, :] * d + trans.reshape(B, 3, 1, 1)                                         # [B,3,D,HW]
        pxy = q[:, :2] / (q[:, 2:3] + 1e-8)
        xn = pxy[:, 0] / ((W - 1) / 2) - 1
        yn = pxy[:, 1] / ((H - 1) / 2) - 1
        xn = torch.where((xn > 1) | (xn < -1), torch.full_like(xn, 2.0), xn)
        yn = torch.where((yn > 1) | (yn < -1), torch.full_like(yn, 2.0), yn)
        grid = torch.stack((xn, yn), dim=3).reshape(B, D * H, W, 2)
        out = F.grid_sample(x, grid, mode="bilinear", padding_mode="zeros", align_corners=False)
        return out.reshape(B, C, D, H, W).numpy()


def homo_warping(src_fea, src_proj, ref_proj, depth_values):
    """utils/homo_utils.py:458-504."""
    return homo_warping_proj(src_fea, matmul(np.asarray(src_proj, _f32), inv(np.asarray(ref_proj, _f32))), depth_values)


def warp_volume(feat_volume, depth, pose, cam_intr, pixel_coords, depth_min, depth_interval, padding_mode="zeros", padding_value=0.0,
                disp_min=None, disp_interval=None):
    """utils/homo_utils.py:240-279, the branch the hybrid decoder calls (zero padding, depth planes): c = K^-1 [x,y,1]^T depth
    (pixel2cam :51-54), c' = inverse(pose) [c;1] (:258, cam2cam :33-36), q = K c' (:116), X = q.x / (q.z + 1e-10), Y likewise, Z = q.z
    (:117-121), normalised x, y by (size - 1) and z by the plane index (:183-188), |n| > 1 -> 2 (:193-198), trilinear 5-D grid_sample,
    zero padding, align_corners=False (:276-277)."""
    assert padding_mode == "zeros" and disp_min is None, "the torch-ops leg restates the branch the hybrid decoder calls"
    with torch.no_grad():
        v = _t(feat_volume)
        N, C, D, H, W = v.shape
        dep = _t(depth).reshape(N, 1, D, H * W)
        K = _t(cam_intr)
        kinv = torch.from_numpy(np.stack([inv(cam_intr[b]) for b in range(N)]))
        M = torch.from_numpy(np.stack([inv(pose[b]) for b in range(N)]))
        grid = torch.from_numpy(set_id_grid(H, W)).reshape(1, 3, H * W).repeat(N, 1, 1)                      # (x, y, 1)
        cam = torch.matmul(kinv, grid)[:, :, None, :] * dep                                                  # [N,3,D,HW]
        cam = cam.reshape(N, 3, D * H * W)
        cam2 = torch.matmul(M[:, :3, :3], cam) + M[:, :3, 3:4]
        q = torch.matmul(K, cam2)
        X = q[:, 0] / (q[:, 2] + 1e-10)
        Y = q[:, 1] / (q[:, 2] + 1e-10)
        Z = q[:, 2]
        xn = 2 * X / (W - 1) - 1
        yn = 2 * Y / (H - 1) - 1
        zn = 2 * ((Z - depth_min) / depth_interval) / (D - 1) - 1
        two = torch.full_like(xn, 2.0)
        xn = torch.where((xn > 1) | (xn < -1), two, xn)
        yn = torch.where((yn > 1) | (yn < -1), two, yn)
        zn = torch.where((zn > 1) | (zn < -1), two, zn)
        g = torch.stack((xn, yn, zn), dim=2).reshape(N, D, H, W, 3)
        return F.grid_sample(v, g, mode="bilinear", padding_mode="zeros", align_corners=False).numpy()


def conv3d(x, weight, bias=None):
    """nn.Conv3d, stride 1, padding k // 2 (networks/layers_op.py:18,:31,:37; transformer/epipolar_transformer.py:21,:26)."""
    with torch.no_grad():
        w = _t(weight)
        return F.conv3d(_t(x), w, None if bias is None else _t(bias), padding=w.shape[2] // 2).numpy()


def bn_act(x, bn, act="none", eps=1e-5):
    """BatchNorm3d in eval mode (+ ReLU / Tanh): networks/layers_op.py:19,:32-33,:38-39."""
    with torch.no_grad():
        y = F.batch_norm(_t(x), _t(bn[2]), _t(bn[3]), _t(bn[0]), _t(bn[1]), False, 0.0, eps)
        if act == "relu":
            y = torch.relu_(y)
        elif act == "tanh":
            y = torch.tanh_(y)
        return y.numpy()


def groupnorm1(x, weight, bias, eps=1e-5):
    """nn.GroupNorm(1, C, eps, affine): transformer/epipolar_transformer.py:22-27."""
    with torch.no_grad():
        return F.group_norm(_t(x), 1, _t(weight), _t(bias), eps).numpy()


def epipolar_attention(target_key, warped_keys, warped_values):
    """transformer/epipolar_transformer.py:62-73: corr_n = sum_c K_t K_n (:65), softmax over the views (:69), h = mean_n(V_n a_n) (:73)."""
    with torch.no_grad():
        kt = _t(target_key)
        wk = torch.stack([_t(k) for k in warped_keys], 1)                   # [B,n,C,D,H,W]
        wv = torch.stack([_t(v) for v in warped_values], 1)
        corr = (kt[:, None] * wk).sum(2, keepdim=True)
        att = torch.softmax(corr, dim=1)
        return (wv * att).mean(1).numpy()


def depthlayer_upsampled(logits_lowres, depth_values, scale=4):
    """F.interpolate(scale_factor=scale) (nearest) + depthlayer (hybrid_depth_decoder.py:33-38,:202-204)."""
    with torch.no_grad():
        lg = F.interpolate(_t(logits_lowres), scale_factor=scale)
        B, D = lg.shape[:2]
        p = torch.softmax(lg, dim=1)
        depth = (p * _t(depth_values).reshape(B, D, 1, 1)).sum(1, keepdim=True)
        prob = p.max(1, keepdim=True)[0]
        return depth.numpy(), prob.numpy()


def sigmoid(x):
    with torch.no_grad():
        return torch.sigmoid(_t(x)).numpy()
