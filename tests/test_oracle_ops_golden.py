"""Pin the C oracle's per-op restatements against golden vectors generated from the reference
(tools/gen_golden.py).  CPU only."""
import os

import numpy as np
import torch

import fixtures_spec as S
from oracle import ref_ops as O
from oracle import ref_model as M


def _load(golden_dir, name):
    return np.load(os.path.join(golden_dir, name))


def test_g1_homo_warping(golden_dir):
    g = _load(golden_dir, "g1_homo_warping.npz")
    for name, src, sp, rp, dv in S.g1_cases():
        out = O.homo_warping(src.numpy(), sp.numpy(), rp.numpy(), dv.numpy())
        ref = g[name]
        diff = np.abs(out - ref)
        # mask discontinuity (|xn|>1 -> 2): the oracle composes its coordinates with the reference's own rounding sequence, so NO
        # sample may flip (a flip moves a value by a texel's worth, 0.1 .. 1).  Measured: max 2.3e-5 / 7.1e-6 on |values| <= 3.3
        # (interpolation-weight rounding), zero elements above 5e-5 -- the same bar the GPU path is held to (_vol_close).
        assert diff.max() < 5e-5, (name, float(diff.max()), int((diff > 5e-5).sum()))
        assert np.median(diff) < 1e-6
        assert (ref != 0).mean() > 0.3


def test_g3_warp_volume(golden_dir):
    g = _load(golden_dir, "g3_warp_volume.npz")
    vol, depth, rel, K, dmin, dint = S.g3_case()
    out = O.warp_volume(vol.numpy(), depth.numpy(), rel.numpy(), K.numpy(), None, dmin, dint)
    diff = np.abs(out - g["out"])
    # measured: bit-identical to the reference's output (max 0.0); one ulp of the volume's range is the bar, no flipped sample
    assert diff.max() < 1e-6, (float(diff.max()), int((diff > 1e-6).sum()))
    assert abs((out == 0).mean() - float(g["zero_frac"])) < 1e-3


def test_g12_level1_signatures_the_hybrid_path_does_not_use(golden_dir):
    """per-pixel depth hypotheses in homo_warping (homo_utils.py:462,:480-481); per-voxel depth, padding_mode='border' with a
    padding value and disparity planes in warp_volume (:246,:253,:187-190,:271-274,:305-319) against the reference's outputs."""
    g = _load(golden_dir, "g12_level1_signatures.npz")
    src, sp, rp, depth = S.g12_homo_case()
    out = O.homo_warping(src.numpy(), sp.numpy(), rp.numpy(), depth.numpy())
    d = np.abs(out - g["homo_per_pixel"])
    assert d.max() < 2e-5 and (g["homo_per_pixel"] != 0).mean() > 0.2, float(d.max())
    for name, kw in S.g12_volume_cases().items():
        a = {k: (v.numpy() if isinstance(v, torch.Tensor) else v) for k, v in kw.items()}
        out = O.warp_volume(a.pop("feat_volume"), a.pop("depth"), a.pop("pose"), a.pop("cam_intr"), None, a.pop("depth_min"),
                            a.pop("depth_interval"), **a)
        ref = g["vol_" + name]
        d = np.abs(out - ref)
        assert d.max() < 2e-5, (name, float(d.max()), float((d > 1e-4).mean()))
        assert (ref != 0).mean() > 0.1, name


def test_g4_epipolar_transformer(golden_dir):
    from estdepth_amd import synth
    from estdepth_amd.epipolar_transformer import EpipolarTransformer
    g = _load(golden_dir, "g4_epipolar_transformer.npz")
    tr = EpipolarTransformer(16, 16, 3)
    synth.fill_state_dict(tr, seed=4)
    P = {"t." + k: v.numpy() for k, v in tr.state_dict().items()}
    for n in (1, 2, 3):
        tk, tv, wv, wk = S.g4_case(n)
        out = M.epipolar_transformer(P, "t", tk.numpy(), tv.numpy(), [w.numpy() for w in wv], [w.numpy() for w in wk])
        assert np.abs(out - g["n%d" % n]).max() < 2e-5


def test_g5_depthlayer(golden_dir):
    g = _load(golden_dir, "g5_depthlayer.npz")
    dv, cases = S.g5_cases()
    for name, lg in cases.items():
        d, p = O.depthlayer_upsampled(lg.numpy(), dv.numpy(), 4)
        assert np.abs(d - g[name + "_depth"]).max() < 2e-5, name
        assert np.abs(p - g[name + "_prob"]).max() < 2e-6, name
    # closed-form known answers: uniform logits -> mean depth, prob 1/D
    d, p = O.depthlayer_upsampled(cases["flat"].numpy(), dv.numpy(), 4)
    assert np.allclose(d, dv.numpy().mean(), atol=1e-5) and np.allclose(p, 1.0 / 16, atol=1e-7)


def test_g2_get_costvolume(golden_dir):
    from estdepth_amd import synth
    from estdepth_amd.model_hybrid import DepthNetHybrid
    g = _load(golden_dir, "g2_get_costvolume.npz")
    m = DepthNetHybrid(ndepths=16, depth_min=0.1, depth_max=10.0, resnet=18, IF_EST_transformer=False)
    synth.fill_state_dict(m, seed=1, head_gain=3.0)
    P = {k: v.numpy() for k, v in m.state_dict().items() if k.startswith("pre")}
    feats = [S._t(20 + i, 1, 32, 16, 20).numpy() for i in range(3)]
    poses = np.stack([synth.camera_pose(v) for v in range(3)])[None]
    K = synth.intrinsics(64, 80).copy()
    K[:2] *= 0.25
    dv = m.depth_cands.view(1, 16, 1, 1).numpy()
    out = M.get_costvolume(P, feats, poses, K[None], dv, 16)
    diff = np.abs(out - g["out"])
    # two 3x3x3 convolutions behind the sweep: measured max 6.4e-6 on |values| <= 9.1 (std 1.24), zero flipped samples
    assert diff.max() < 2e-5, (float(diff.max()), int((diff > 2e-5).sum()))
    assert np.median(diff) < 2e-6


def test_identity_pose_known_answer():
    """identity pose => homo_warping resamples src at x*W/(W-1)-0.5 (Q5), not identity."""
    C, H, W, D = 2, 6, 9, 3
    src = np.random.RandomState(0).randn(1, C, H, W).astype(np.float32)
    eye = np.eye(4, dtype=np.float32)[None]
    dv = np.array([1.0, 2.0, 3.0], np.float32).reshape(1, D, 1, 1)
    out = O.homo_warping(src, eye, eye, dv)
    t = torch.from_numpy(src)
    ys, xs = torch.meshgrid(torch.arange(H, dtype=torch.float32), torch.arange(W, dtype=torch.float32), indexing="ij")
    grid = torch.stack([xs / ((W - 1) / 2) - 1, ys / ((H - 1) / 2) - 1], -1)[None]
    exp = torch.nn.functional.grid_sample(t, grid, mode="bilinear", padding_mode="zeros", align_corners=False).numpy()
    for d in range(D):
        assert np.abs(out[:, :, d] - exp).max() < 1e-5


def test_attention_known_answers():
    """SURVEY §4, transformer/epipolar_transformer.py:62-73 (softmax over the views, then the MEAN over views of the weighted values):
    one source => weight 1 => h = V_1; n identical sources => weights 1/n => h = V / n; a source whose key correlates far more
    strongly than the other takes all the weight => h = V_1 / 2."""
    rs = np.random.RandomState(1)
    B, C, D, H, W = 1, 16, 3, 4, 5
    tk = rs.randn(B, C, D, H, W).astype(np.float32)
    k1, v1 = rs.randn(B, C, D, H, W).astype(np.float32), rs.randn(B, C, D, H, W).astype(np.float32)
    h = O.epipolar_attention(tk, [k1], [v1])
    assert np.abs(h - v1).max() < 1e-6
    h = O.epipolar_attention(tk, [k1, k1, k1], [v1, v1, v1])
    assert np.abs(h - v1 / 3.0).max() < 1e-6
    v2 = rs.randn(B, C, D, H, W).astype(np.float32)
    h = O.epipolar_attention(tk, [50.0 * tk, -50.0 * tk], [v1, v2])     # correlation +50|k|^2 vs -50|k|^2
    assert np.abs(h - v1 / 2.0).max() < 1e-5
