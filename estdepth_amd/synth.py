"""Deterministic synthetic weights and inputs (no datasets / checkpoints are reachable offline).

Used by bench.py, the tests and tools/gen_golden.py so that the reference, the oracle and the
HIP path all see bit-identical parameters without shipping a 145 MB state dict.

Recipe: key-hashed filler; BatchNorm running statistics randomised (the
default init gives a constant 5.05 m depth map that would pass any tolerance vacuously); the two
stereo heads get a gain so the soft-argmin is non-degenerate and a bias shift so that the fused
logits straddle zero (otherwise ``relu(all_fused_logits)`` is identically 0 and the refined
outputs ignore the matching branch).
"""
import math
import re
import zlib

import numpy as np
import torch


_RESIDUAL_BN = re.compile(r"(layer\d+\.\d+\.conv2\.1|layer\d+\.\d+\.bn[23])\.weight$")


def _rng(key, seed):
    return np.random.RandomState((zlib.crc32(key.encode()) + 7919 * seed) & 0x7FFFFFFF)


def fill_state_dict(module, seed=0, head_gain=10.0, disp_gain=0.1):
    """In-place deterministic fill of every parameter/buffer of ``module`` keyed by its state-dict name."""
    sd = module.state_dict()
    new = {}
    for key in sorted(sd.keys()):
        t = sd[key]
        r = _rng(key, seed)
        shape = tuple(t.shape)
        leaf = key.rsplit(".", 1)[-1]
        if leaf == "num_batches_tracked":
            new[key] = torch.zeros_like(t)
            continue
        if leaf == "running_mean":
            v = r.normal(0.0, 0.1, shape)
        elif leaf == "running_var":
            v = r.uniform(0.6, 1.4, shape)
        elif t.dim() <= 1 and leaf == "weight":          # BN / GN scale
            v = r.uniform(0.8, 1.2, shape)
        elif t.dim() <= 1 and leaf == "bias":            # BN / GN shift, conv bias
            v = r.normal(0.0, 0.1, shape)
        else:                                            # conv / linear weight: He-style
            fan_in = int(np.prod(shape[1:])) if len(shape) > 1 else shape[0]
            v = r.normal(0.0, math.sqrt(2.0 / max(fan_in, 1)), shape)
        new[key] = torch.from_numpy(np.asarray(v, np.float32)).reshape(shape).to(t.dtype)
    # stereo heads: gain + small bias (logits = gain * w.relu(f) + b straddle zero because w is signed)
    for key in new:
        if key.endswith("stereo_head0.1.weight") or key.endswith("stereo_head1.1.weight"):
            new[key] = new[key] * head_gain
        if _RESIDUAL_BN.search(key):
            new[key] = new[key] * 0.25           # damp every residual branch: 25+ blocks would otherwise blow up to 1e5
        if key.endswith("lastconv.2.weight"):
            new[key] = new[key] * 0.25           # matching features ~O(1)
        if key.endswith("dispconv_0.weight") or key.endswith("dispconv_1.weight"):
            new[key] = new[key] * disp_gain      # keep depth_max*sigmoid(.) out of saturation
    module.load_state_dict(new)
    return module


# ------------------------------------------------------------------------------------------ inputs
def rot_xyz(ax, ay, az):
    cx, sx, cy, sy, cz, sz = math.cos(ax), math.sin(ax), math.cos(ay), math.sin(ay), math.cos(az), math.sin(az)
    rx = np.array([[1, 0, 0], [0, cx, -sx], [0, sx, cx]])
    ry = np.array([[cy, 0, sy], [0, 1, 0], [-sy, 0, cy]])
    rz = np.array([[cz, -sz, 0], [sz, cz, 0], [0, 0, 1]])
    return rz @ ry @ rx


def camera_pose(v, motion=1.0):
    """Camera-to-world pose of frame v: R = Rz(.01v) Ry(.02v) Rx(-.005v), t = (.05v,.002v,.01v) m.
    Non-commuting on purpose so the relative-pose quirk is observable."""
    p = np.eye(4)
    p[:3, :3] = rot_xyz(-0.005 * v * motion, 0.02 * v * motion, 0.01 * v * motion)
    p[:3, 3] = np.array([0.05, 0.002, 0.01]) * v * motion
    return p.astype(np.float32)


def intrinsics(hi, wi):
    """ScanNet/7-Scenes constants of data/general_eval.py:167-178, rescaled to the image size."""
    return np.array([[577.87 * wi / 640.0, 0.0, 319.5 * wi / 640.0],
                     [0.0, 577.87 * hi / 480.0, 239.5 * hi / 480.0],
                     [0.0, 0.0, 1.0]], np.float32)


def make_sequence(n_views, hi, wi, seed, first_frame=0, batch=1):
    """Synthetic sequence: imgs [B,V,3,Hi,Wi] in 0..255, cam_poses [B,V,4,4], cam_intr [B,3,3], sample dict."""
    g = torch.Generator().manual_seed(int(seed))
    imgs = torch.rand(batch, n_views, 3, hi, wi, generator=g) * 255.0
    poses = torch.from_numpy(np.stack([camera_pose(first_frame + v) for v in range(n_views)]))[None].repeat(batch, 1, 1, 1)
    intr = torch.from_numpy(intrinsics(hi, wi))[None].repeat(batch, 1, 1)
    sample = {"dmaps": torch.rand(batch, n_views, 1, hi, wi, generator=g) * 5.0 + 0.5,
              "dmasks": torch.ones(batch, n_views, 1, hi, wi, dtype=torch.bool)}
    return imgs, poses, intr, sample


def smooth_images(n_views, hi, wi, seed, batch=1):
    """Low-frequency images (white noise makes every plane equally (un)likely); used by fixtures."""
    g = torch.Generator().manual_seed(int(seed))
    low = torch.rand(batch * n_views, 3, hi // 8, wi // 8, generator=g)
    img = torch.nn.functional.interpolate(low, size=(hi, wi), mode="bilinear", align_corners=False)
    img = img + 0.05 * torch.rand(batch * n_views, 3, hi, wi, generator=g)
    return (img.clamp(0, 1) * 255.0).reshape(batch, n_views, 3, hi, wi)
