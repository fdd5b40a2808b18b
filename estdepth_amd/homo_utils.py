"""Level-1 geometric operators with the reference's names and signatures
(utils/homo_utils.py), executed by the HIP kernels of libestd_hip.so.

Semantics kept verbatim (hard parts): align_corners=False un-normalisation although
coordinates are normalised with the (size-1) formula, |norm| > 1 -> 2 masking, no z>0 test,
eps 1e-8 (2D) vs 1e-10 (3D).
"""
import torch

from . import camera, ops


def set_id_grid(h, w):
    """utils/homo_utils.py:7-14 -> [1,3,H,W] rows (x, y, 1), float32 on CPU like the reference."""
    i_range = torch.arange(0, h).view(1, h, 1).expand(1, h, w).to(torch.float32)
    j_range = torch.arange(0, w).view(1, 1, w).expand(1, h, w).to(torch.float32)
    ones = torch.ones(1, h, w, dtype=torch.float32)
    return torch.stack((j_range, i_range, ones), dim=1)


def _per_plane(depth, batch, num_depth):
    """[B,D,...] -> ([B,D] per-plane constants, None) when every plane holds one value, else (None, [B,D,H*W] per pixel).
    No device synchronisation where the layout says it all: a trailing size of 1, or an EXPANDED view (stride 0 over the pixels --
    what ``depth_values.expand(...)`` hands over).  A materialised [B,D,H,W] tensor (``depth_values.repeat(1,1,H,W)``, the hybrid
    callers: model_hybrid.py:95, hybrid_depth_decoder.py:238) has to be looked at: one blocking ``.all()`` -- the level-1 operators
    are the drop-in surface, the fast path (DepthNetHybrid.forward) never comes through here."""
    if depth.dim() >= 3 and depth.shape[0] == batch and depth.shape[1] == num_depth and all(
            sz == 1 or st == 0 for sz, st in zip(depth.shape[2:], depth.stride()[2:])):
        return depth[(slice(None), slice(None)) + (0,) * (depth.dim() - 2)].float().contiguous(), None
    dv = depth.reshape(batch, num_depth, -1).float()
    if dv.shape[2] == 1:
        return dv[:, :, 0].contiguous(), None
    if bool((dv == dv[:, :, :1]).all()):
        return dv[:, :, 0].contiguous(), None
    return None, dv.contiguous()


def homo_warping(src_fea, src_proj, ref_proj, depth_values):
    """utils/homo_utils.py:458-504.  src_fea [B,C,H,W]; src_proj/ref_proj [B,4,4];
    depth_values [B,D], [B,D,1,1] or per-pixel hypotheses [B,D,H,W] (:462) -> [B,C,D,H,W]."""
    batch, channels, height, width = src_fea.shape
    num_depth = depth_values.shape[1]
    planes, per_pixel = _per_plane(depth_values, batch, num_depth)
    outs = []
    for b in range(batch):
        proj = camera.pair_projection(src_proj[b], ref_proj[b], src_fea.device)          # :469-471 with the reference's torch-CPU calls
        if planes is not None:
            outs.append(ops.homo_warping_chw(src_fea[b].contiguous(), proj, planes[b], num_depth))
        else:
            outs.append(ops.homo_warping_px_chw(src_fea[b].contiguous(), proj, per_pixel[b].reshape(num_depth, height, width)))
    return torch.stack(outs, 0)


def warp_volume(feat_volume, depth, pose, cam_intr, pixel_coords, depth_min, depth_interval,
                padding_mode='zeros', padding_value=0., disp_min=None, disp_interval=None, inter_mode='bilinear'):
    """utils/homo_utils.py:240-279.  feat_volume [N,C,D,H,W]; depth [N,1,D,H*W] (per plane or per voxel, :246);
    pose [N,4,4] relative pose (the function applies inverse(pose) like the reference); cam_intr [N,3,3];
    padding_mode 'zeros' | 'border' (+ padding_value, :271-274); disparity planes when disp_min is given (:187-190).
    ``pixel_coords`` (the cached id grid) is accepted for signature parity and regenerated in-kernel.
    Works for any D >= 2 (the reference crashes for D < 63 because of a debug leftover)."""
    if padding_mode not in ('zeros', 'border') or inter_mode != 'bilinear':
        raise RuntimeError("warp_volume: padding_mode 'zeros' or 'border', inter_mode 'bilinear' (got %r, %r)" % (padding_mode, inter_mode))
    if (disp_min is None) != (disp_interval is None):
        raise RuntimeError("warp_volume: disp_min and disp_interval go together")
    N, C, D, H, W = feat_volume.shape
    planes, per_voxel = _per_plane(depth.reshape(N, D, H * W), N, D)
    plain = planes is not None and padding_mode == 'zeros' and disp_min is None
    outs = []
    for b in range(N):
        mats = camera.relative_volume_matrix(pose[b], cam_intr[b], feat_volume.device)  # :51, :258 with the reference's torch-CPU calls
        if plain:                                                                       # the hybrid path's call
            outs.append(ops.warp_volume_cdhw(feat_volume[b].contiguous(), mats, planes[b], depth_min, depth_interval))
        else:
            dep = planes[b] if planes is not None else per_voxel[b].reshape(-1)
            outs.append(ops.warp_volume_ex_cdhw(feat_volume[b].contiguous(), mats, dep, planes is None, depth_min, depth_interval,
                                                disp_min, disp_interval, padding_mode == 'border', padding_value))
    return torch.stack(outs, 0)
