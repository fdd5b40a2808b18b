"""Deterministic INPUT recipes of the golden fixtures (tests/golden/*.npz hold only expected outputs).

Shared by tools/gen_golden.py (which feeds them to the reference) and by the tests (which feed
them to the oracle and to the HIP path).  Everything is generated on CPU with seeded generators.
"""
import numpy as np
import torch

from estdepth_amd import synth


def _t(seed, *shape, scale=1.0):
    g = torch.Generator().manual_seed(seed)
    return torch.randn(*shape, generator=g) * scale


def poses_list(n, first=0):
    return [torch.from_numpy(synth.camera_pose(first + v))[None] for v in range(n)]


# ---------------------------------------------------------------- G1 homo_warping
def g1_cases():
    cases = []
    for name, (C, H, W, D, seed) in {"a": (8, 24, 32, 16, 11), "b": (4, 12, 16, 64, 12)}.items():
        src = _t(seed, 1, C, H, W)
        K = torch.from_numpy(synth.intrinsics(H * 4, W * 4)).clone()
        K[:2] *= 0.25
        def proj(v):
            e = torch.inverse(torch.from_numpy(synth.camera_pose(v, motion=2.0)))
            p = e.clone()
            p[:3, :4] = K @ e[:3, :4]
            return p[None]
        dv = (torch.arange(D, dtype=torch.float32) * ((10.0 - 0.1) / (D - 1)) + 0.1).view(1, D, 1, 1)
        cases.append((name, src, proj(0), proj(1), dv))
    return cases


# ---------------------------------------------------------------- G3 warp_volume
def g3_case():
    C, D, H, W = 16, 64, 12, 16
    vol = _t(31, 1, C, D, H, W)
    K = torch.from_numpy(synth.intrinsics(H * 4, W * 4)).clone()
    K[:2] *= 0.25
    depth_min, depth_max = 0.1, 10.0
    interval = (depth_max - depth_min) / (D - 1)
    dv = (torch.arange(D, dtype=torch.float32) * interval + depth_min).view(1, D, 1, 1)
    depth = dv.repeat(1, 1, H, W).view(1, 1, D, H * W)
    pi = torch.from_numpy(synth.camera_pose(1, motion=1.5))
    pj = torch.from_numpy(synth.camera_pose(3, motion=1.5))
    rel = (pj @ torch.inverse(pi))[None]
    return vol, depth, rel, K[None], depth_min, interval


# ---------------------------------------------------------------- G12 level-1 signatures the hybrid callers do not use
def g12_homo_case():
    """homo_warping with PER-PIXEL depth hypotheses [B, Ndepth, H, W] (homo_utils.py:462,:480-481)."""
    C, H, W, D = 4, 12, 16, 8
    name, src, sp, rp, dv = g1_cases()[0][0], _t(121, 1, C, H, W), None, None, None
    K = torch.from_numpy(synth.intrinsics(H * 4, W * 4)).clone()
    K[:2] *= 0.25
    def proj(v):
        e = torch.inverse(torch.from_numpy(synth.camera_pose(v, motion=2.0)))
        p = e.clone()
        p[:3, :4] = K @ e[:3, :4]
        return p[None]
    planes = (torch.arange(D, dtype=torch.float32) * ((10.0 - 0.1) / (D - 1)) + 0.1).view(1, D, 1, 1)
    depth = planes * (1.0 + 0.05 * torch.tanh(_t(122, 1, D, H, W)))
    return src, proj(0), proj(1), depth.contiguous()


def g12_volume_cases():
    """warp_volume (homo_utils.py:240-279) beyond the hybrid path's call: per-VOXEL depth (:246,:253), padding_mode='border' with a
    padding value (:271-274, :305-319), disparity planes (:187-190).  D = 64: the reference reads depth[:, 0, 62] (Q6)."""
    C, D, H, W = 4, 64, 12, 16
    vol = _t(131, 1, C, D, H, W)
    K = torch.from_numpy(synth.intrinsics(H * 4, W * 4)).clone()
    K[:2] *= 0.25
    pi = torch.from_numpy(synth.camera_pose(1, motion=1.5))
    pj = torch.from_numpy(synth.camera_pose(3, motion=1.5))
    rel = (pj @ torch.inverse(pi))[None]
    dmin, dmax = 0.1, 10.0
    dint = (dmax - dmin) / (D - 1)
    planes = (torch.arange(D, dtype=torch.float32) * dint + dmin).view(1, D, 1, 1)
    per_plane = planes.repeat(1, 1, H, W).view(1, 1, D, H * W)
    per_voxel = (planes * (1.0 + 0.03 * torch.tanh(_t(132, 1, D, H, W)))).reshape(1, 1, D, H * W).contiguous()
    disp_min, disp_max = 1.0 / 10.0, 1.0 / 0.5
    disp_int = (disp_max - disp_min) / (D - 1)
    disp_depth = (1.0 / (torch.arange(D, dtype=torch.float32) * disp_int + disp_min)).view(1, D, 1, 1).repeat(1, 1, H, W).view(1, 1, D, H * W)
    base = dict(feat_volume=vol, pose=rel, cam_intr=K[None], depth_min=dmin, depth_interval=dint)
    return {"per_voxel": dict(base, depth=per_voxel),
            "border": dict(base, depth=per_plane, padding_mode="border", padding_value=0.5),
            "border_per_voxel": dict(base, depth=per_voxel, padding_mode="border", padding_value=-1.25),
            "disp": dict(base, depth=disp_depth, disp_min=disp_min, disp_interval=disp_int)}


# ---------------------------------------------------------------- G4 EpipolarTransformer
def g4_case(n_views):
    C, D, H, W = 16, 8, 12, 16
    tk = torch.relu(_t(41, 1, C, D, H, W))
    tv = torch.tanh(_t(42, 1, C, D, H, W))
    wk = [torch.relu(_t(43 + i, 1, C, D, H, W)) for i in range(n_views)]
    wv = [torch.tanh(_t(53 + i, 1, C, D, H, W)) for i in range(n_views)]
    return tk, tv, wv, wk


# ---------------------------------------------------------------- G5 depthlayer
def g5_cases():
    D, H, W = 16, 6, 8
    dv = (torch.arange(D, dtype=torch.float32) * 0.66 + 0.1).view(1, D, 1, 1)
    flat = torch.zeros(1, D, H, W)
    peaky = _t(61, 1, D, H, W, scale=25.0)
    tie = torch.zeros(1, D, H, W)
    tie[:, 3] = 5.0
    tie[:, 9] = 5.0
    return dv, {"flat": flat, "peaky": peaky, "tie": tie, "rand": _t(62, 1, D, H, W, scale=3.0)}


# ---------------------------------------------------------------- G6 decoder
def g6_inputs(resnet, T, H=24, W=32, D=64, seed=70):
    ch = [64, 64, 128, 256, 512] if resnet <= 34 else [64, 256, 512, 1024, 2048]
    sem = [torch.relu(_t(seed + s, T, ch[s], (4 * H) >> (s + 1), (4 * W) >> (s + 1))) for s in range(5)]
    cvs = [_t(seed + 10 + t, 1, 32, D, H, W, scale=0.7) for t in range(T)]
    K = torch.from_numpy(synth.intrinsics(H * 4, W * 4)).clone()
    K[:2] *= 0.25
    dmin, dmax = 0.1, 10.0
    interval = (dmax - dmin) / (D - 1)
    dv = (torch.arange(D, dtype=torch.float32) * interval + dmin).view(1, D, 1, 1)
    poses = poses_list(T, first=1)
    return cvs, sem, poses, K[None], dv, dmin, interval


def g6_memory(n, H=24, W=32, D=64, seed=90):
    keys = [torch.relu(_t(seed + i, 1, 16, D, H, W)) for i in range(n)]
    vals = [torch.tanh(_t(seed + 5 + i, 1, 16, D, H, W)) for i in range(n)]
    poses = [torch.from_numpy(synth.camera_pose(-1 - i))[None] for i in range(n)]
    return {"keys": keys, "values": vals}, poses


# ---------------------------------------------------------------- G7-G9 end to end
def e2e_inputs(n_views, hi, wi, seed, first_frame=0):
    imgs = synth.smooth_images(n_views, hi, wi, seed)
    poses = torch.from_numpy(np.stack([synth.camera_pose(first_frame + v) for v in range(n_views)]))[None]
    intr = torch.from_numpy(synth.intrinsics(hi, wi))[None]
    sample = {"dmaps": torch.ones(1, n_views, 1, hi, wi), "dmasks": torch.ones(1, n_views, 1, hi, wi, dtype=torch.bool)}
    return imgs, poses, intr, sample


E2E_HI, E2E_WI = 128, 160


def g10_cases():
    """G10 depth-error suite (metric.py): (name, pred, gt) float64 maps with out-of-window, zero, NaN and inf pixels."""
    out = []
    for i, (h, w, bias) in enumerate(((24, 32, 0.0), (17, 23, 0.4), (8, 8, -0.2))):
        g = np.random.default_rng(100 + i)
        gt = g.uniform(0.1, 6.0, size=(h, w))
        pred = gt * np.exp(g.normal(bias * 0.1, 0.15, size=(h, w))) + bias * 0.05
        gt[0, :3] = 0.0
        gt[1, 0], gt[1, 1] = np.nan, np.inf
        pred[2, 0], pred[2, 1], pred[2, 2] = np.nan, np.inf, -1.0
        out.append(("m%d" % i, pred, gt))
    out.append(("empty", np.full((4, 4), 7.0), np.full((4, 4), 7.0)))
    return out
