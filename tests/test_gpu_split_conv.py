"""estd_conv3d_k3_split (3 x bf16 operand split, six bf16 MFMAs per product block) against
  * an fp64 convolution (how much error does each arithmetic carry?),
  * the fp32 MFMA kernel on every epilogue feature and on ragged shapes,
  * the reference's golden vectors end to end (depth within 1e-4) with the split arithmetic switched on.
"""
import numpy as np
import pytest
import torch

import fixtures_spec as S
from helpers import checksum, checksum_close

pytestmark = [pytest.mark.gpu, pytest.mark.ab]
DEV = torch.device("cuda:0")


@pytest.fixture(scope="module", autouse=True)
def _need_gpu():
    if not torch.cuda.is_available():
        pytest.fail("GPU tests need an MI355X (no CPU path exists)")


@pytest.fixture()
def split_arith(monkeypatch):
    from estdepth_amd import ops
    monkeypatch.setattr(ops, "CONV3D_ARITH", "bf16x3")
    return ops


def _plan(seed, act="relu", bias=True):
    from estdepth_amd import synth
    from estdepth_amd.layers_op import ConvBN3d
    mod = ConvBN3d(32, 32, 3, 1, 1, act).eval()
    synth.fill_state_dict(mod, seed=seed)
    return mod.to(DEV), mod.to(DEV).plan()


def _run(plan, arith, x, dims, **kw):
    from estdepth_amd import ops
    old, old_algo = ops.CONV3D_ARITH, ops.CONV3D_ALGO
    ops.CONV3D_ARITH = arith
    ops.CONV3D_ALGO = "direct"          # the fp32 yardstick of this file is the DIRECT 27-tap MFMA kernel (the split kernel's sibling)
    try:
        out = kw.pop("out", None)
        if out is None:
            out = torch.empty(dims + (kw.get("out_stride", 32),), device=DEV)
        plan.run(x, dims, out=out, **kw)
        torch.cuda.synchronize()
        return out
    finally:
        ops.CONV3D_ARITH, ops.CONV3D_ALGO = old, old_algo


@pytest.mark.parametrize("dims,scale", [((1, 6, 19, 45), 1.0), ((2, 3, 8, 32), 100.0), ((1, 1, 5, 7), 1e-3)])
def test_split_error_is_fp32_level(dims, scale):
    """both kernels vs an fp64 convolution of the same fp32 data: the split arithmetic must not be worse than twice
    the fp32 MFMA kernel's own rounding error (and far inside the 1e-4 depth budget)."""
    mod, plan = _plan(11, act=None)
    N, D, H, W = dims
    g = torch.Generator().manual_seed(5)
    x = torch.randn(N, D, H, W, 32, generator=g) * scale
    w64 = mod[0].weight.detach().double().cpu()
    bn = mod[1]
    sc = (bn.weight / torch.sqrt(bn.running_var + bn.eps)).double().cpu()
    sh = (bn.bias.double().cpu() - bn.running_mean.double().cpu() * sc)
    ref = torch.nn.functional.conv3d(x.permute(0, 4, 1, 2, 3).double(), w64, padding=1)
    ref = (ref * sc[None, :, None, None, None] + sh[None, :, None, None, None]).permute(0, 2, 3, 4, 1)
    xd = x.to(DEV)
    e32 = (_run(plan, "f32", xd, dims).double().cpu() - ref).abs().max().item()
    esp = (_run(plan, "bf16x3", xd, dims).double().cpu() - ref).abs().max().item()
    mag = ref.abs().max().item()
    print("fp64 check dims=%s scale=%g: |ref|max %.3g  err f32 %.3g  err bf16x3 %.3g" % (dims, scale, mag, e32, esp))
    assert esp <= 2.0 * e32 + 1e-7 * mag, (esp, e32)
    assert esp < 2e-6 * mag


@pytest.mark.parametrize("dims", [(1, 4, 8, 32), (3, 5, 13, 50), (1, 2, 120, 160), (2, 1, 9, 33), (1, 70, 8, 32)])
def test_split_matches_fp32_kernel_all_epilogues(dims):
    """residual, second residual, 1/n scale, running accumulation, none|ReLU split, GroupNorm partials, strided I/O."""
    from estdepth_amd import ops
    mod, plan = _plan(21)
    plan.act_a, plan.act_b, plan.act_split = ops.ACT["none"], ops.ACT["relu"], 16
    N, D, H, W = dims
    g = torch.Generator(device=DEV).manual_seed(3)
    x = torch.randn(N, D, H, W, 40, device=DEV, generator=g)           # in_stride 40 > 32
    r1 = torch.randn(N, D, H, W, 36, device=DEV, generator=g)          # out_stride 36
    r2 = torch.randn(N, D, H, W, 36, device=DEV, generator=g)
    base = torch.randn(N, D, H, W, 36, device=DEV, generator=g)
    nblk = ops.conv3d_grid(N, D, H, W)
    res = {}
    for arith in ("f32", "bf16x3"):
        out = base.clone()
        part = torch.full((nblk * 4,), float("nan"), device=DEV, dtype=torch.float64)
        _run(plan, arith, x, dims, in_stride=40, out=out, out_stride=36, residual=r1, residual2=r2, out_scale=0.5,
             accumulate=True, stats_partials=part)
        st = ops.groupnorm_finalize(part, nblk, 16.0 * N * D * H * W, 1e-5)
        res[arith] = (out, st)
    a, b = res["f32"][0], res["bf16x3"][0]
    assert torch.equal(a[..., 32:], base[..., 32:]) and torch.equal(b[..., 32:], base[..., 32:])   # padding channels untouched
    mag = a[..., :32].abs().max().item()
    assert (a - b).abs().max().item() < 3e-6 * max(mag, 1.0)
    np.testing.assert_allclose(res["bf16x3"][1].cpu().numpy(), res["f32"][1].cpu().numpy(), rtol=2e-5, atol=1e-6)


def test_split_rejects_other_shapes():
    from estdepth_amd import _native as N_, ops
    d = N_.Conv3dDesc()
    d.N, d.D, d.H, d.W, d.cin_main, d.in_stride, d.n_tiles = 1, 2, 8, 16, 16, 16, 1
    t = torch.zeros(64, device=DEV)
    d.in_main = d.w_split = d.scale = d.shift = d.out_main = t.data_ptr()
    d.out_stride = 16
    import ctypes
    assert N_.lib().estd_conv3d_k3_split(ctypes.byref(d), None) < 0          # 16-channel shapes stay on the fp32 kernel
    assert N_.lib().estd_conv3d_k3_split(None, None) < 0


def test_split_linearity_full_size():
    """BASELINE cfg2 volume (64x120x160x32, N=3): conv(a x + b y) = a conv(x) + b conv(y) and agreement with the fp32
    kernel on the full tensor."""
    from estdepth_amd.layers_op import ConvBN3d
    mod, plan = _plan(77, act=None)
    with torch.no_grad():
        mod[1].bias.zero_(); mod[1].running_mean.zero_()
    plan = mod.plan()                    # PlanCache repacks after the parameter edit
    dims = (3, 64, 120, 160)
    g = torch.Generator(device=DEV).manual_seed(1)
    x = torch.randn(*dims, 32, device=DEV, generator=g)
    y = torch.randn(*dims, 32, device=DEV, generator=g)
    ox, oy = _run(plan, "bf16x3", x, dims), _run(plan, "bf16x3", y, dims)
    oz = _run(plan, "bf16x3", 0.5 * x - 2.0 * y, dims)
    lin = 0.5 * ox - 2.0 * oy
    assert (oz - lin).abs().max().item() < 5e-4 * lin.abs().max().item()
    o32 = _run(plan, "f32", x, dims)
    assert (o32 - ox).abs().max().item() < 2e-6 * o32.abs().max().item()


def test_split_cfg5_volume_matches_fp32_kernel():
    """BASELINE configs[4] volume (128x240x320x32 = 1.26 GB): the two arithmetics agree on the whole tensor, with a
    residual and the 1/2 scale of the cost-volume mean in the epilogue."""
    mod, plan = _plan(55)
    dims = (1, 128, 240, 320)
    g = torch.Generator(device=DEV).manual_seed(2)
    x = torch.randn(*dims, 32, device=DEV, generator=g)
    a = _run(plan, "f32", x, dims, residual=x, out_scale=0.5)
    b = _run(plan, "bf16x3", x, dims, residual=x, out_scale=0.5)
    assert bool(torch.isfinite(b).all())
    assert (a - b).abs().max().item() < 2e-6 * a.abs().max().item()


def test_joint_carry_golden_with_split_arithmetic(golden_dir, split_arith):
    """configs[1] protocol (two chained Joint calls, EST transformer on the second) vs the reference's golden depth maps."""
    from estdepth_amd import DepthNetHybrid, synth
    g = np.load(golden_dir + "/g9_joint_carry.npz")
    m = DepthNetHybrid(ndepths=64, depth_min=0.1, depth_max=10.0, resnet=18, IF_EST_transformer=True).eval()
    synth.fill_state_dict(m, seed=2, head_gain=1.0)
    m = m.to(DEV)
    imgs, poses, intr, sample = S.e2e_inputs(8, S.E2E_HI, S.E2E_WI, seed=1004)
    imgs, poses, intr = imgs.to(DEV), poses.to(DEV), intr.to(DEV)
    pre_costs, pre_poses = None, None
    worst = 0.0
    for call in range(2):
        sl = slice(3 * call, 3 * call + 5)
        with torch.no_grad():
            outputs, pre_costs, pre_poses = m(imgs[:, sl], poses[:, sl], intr, {k: v[:, sl] for k, v in sample.items()},
                                              pre_costs, pre_poses, mode="val")
        for k, v in outputs.items():
            name = "c%d|" % call + "|".join(map(str, k))
            if name not in g.files:
                continue
            err = float(np.abs(v.cpu().numpy() - g[name]).max())
            worst = max(worst, err)
            assert err < 1e-4, (name, err)                     # north_star tolerance
        assert checksum_close(checksum(pre_costs["values"][0].cpu().numpy()), g["c%d|value_ck" % call])
    print("joint carry with bf16x3 arithmetic: worst |depth - reference| = %.3g" % worst)


@pytest.mark.parametrize("dims", [(1, 4, 8, 32), (2, 5, 13, 50), (1, 66, 9, 33), (3, 1, 24, 32)])
def test_split_extra_input_and_33rd_output_match_fp32_kernel(dims):
    """dres2 (33 -> 33: scalar extra input channel + 33rd output channel) and the fused value|key conv (33 -> 32, tanh|relu)
    on both arithmetics (hybrid_depth_decoder.py:95-97)."""
    from estdepth_amd import ops, synth
    from estdepth_amd.layers_op import ConvBN3d
    N, D, H, W = dims
    g = torch.Generator(device=DEV).manual_seed(9)
    x = torch.randn(N, D, H, W, 32, device=DEV, generator=g)
    xe = torch.randn(N, D, H, W, device=DEV, generator=g)
    d2 = ConvBN3d(33, 33, 3, 1, 1, "relu").eval()
    synth.fill_state_dict(d2, seed=33)
    p2 = d2.to(DEV).plan(main_idx=list(range(1, 33)), extra_idx=0, out_idx=list(range(33)), n_tiles=3)
    vl, kl = ConvBN3d(33, 16, 3, 1, 1, "tanh").eval(), ConvBN3d(33, 16, 3, 1, 1, "relu").eval()
    synth.fill_state_dict(vl, seed=34); synth.fill_state_dict(kl, seed=35)
    vl, kl = vl.to(DEV), kl.to(DEV)
    sv, hv = vl.folded(); sk, hk = kl.folded()
    pkv = ops.Conv3dPlan(torch.cat([vl[0].weight.detach(), kl[0].weight.detach()], 0), list(range(32)), 32, list(range(32)), 2,
                         torch.cat([sv, sk]), torch.cat([hv, hk]), act_a="tanh", act_b="relu", act_split=16, device=DEV)
    assert p2.w_split is not None and pkv.w_split is not None
    res = {}
    for arith in ("f32", "bf16x3"):
        ops.CONV3D_ARITH = arith
        try:
            a = torch.empty_like(x); ex = torch.full_like(xe, float("nan"))
            p2.run(x, dims, in_extra=xe, out=a, out_stride=32, out_extra=ex)
            kv = torch.empty_like(x)
            pkv.run(a, dims, in_extra=ex, out=kv, out_stride=32)
            torch.cuda.synchronize()
        finally:
            ops.CONV3D_ARITH = "f32"
        res[arith] = (a, ex, kv)
    for k, name in enumerate(("dres2 main", "dres2 33rd", "value|key")):
        a, b = res["f32"][k], res["bf16x3"][k]
        assert torch.isfinite(b).all(), name
        assert (a - b).abs().max().item() < 3e-6 * max(1.0, a.abs().max().item()), (name, (a - b).abs().max().item())


def test_decoder_golden_with_split_arithmetic(golden_dir, split_arith):
    """G6 (decoder, R18, one memory volume: transformer branch) with every eligible conv on the split kernel."""
    from estdepth_amd import DepthHybridDecoder, synth
    import os
    g = np.load(os.path.join(golden_dir, "g6_decoder_r18_mem1.npz"))
    ch = np.array([64, 64, 128, 256, 512])
    dec = DepthHybridDecoder(ch, ndepths=64, depth_max=10.0, IF_EST_transformer=True).eval()
    synth.fill_state_dict(dec, seed=6)
    dec = dec.to(DEV)
    cvs, sem, cposes, K, dv, dmin, dint = S.g6_inputs(18, 2)
    pre_costs, pre_poses = S.g6_memory(1)
    to = lambda t: t.to(DEV)
    pre_costs = {k: [to(t) for t in v] for k, v in pre_costs.items()}
    with torch.no_grad():
        outputs, costs, rposes = dec([to(c) for c in cvs], [to(s_) for s_ in sem], [to(p_) for p_ in cposes], to(K), to(dv), dmin, dint,
                                     pre_costs, [to(p_) for p_ in pre_poses], mode="val")
    for k, v in outputs.items():
        name = "|".join(map(str, k))
        assert np.abs(v.cpu().numpy() - g[name]).max() < 1e-4, name


@pytest.mark.parametrize("dims,stats", [((1, 4, 8, 32), True), ((2, 5, 13, 50), False), ((1, 64, 120, 160), True)])
def test_split_32_to_16_matches_fp32_kernel(dims, stats):
    """the GRU output convolution shape (32 -> 16, bias, optional GroupNorm partials; epipolar_transformer.py:26)."""
    from estdepth_amd import ops
    g = torch.Generator().manual_seed(4)
    w = torch.randn(16, 32, 3, 3, 3, generator=g) * 0.05
    bias = torch.randn(16, generator=g)
    plan = ops.Conv3dPlan(w, list(range(32)), None, list(range(16)), 1, torch.ones(16), bias, device=DEV)
    assert plan.w_split is not None
    N, D, H, W = dims
    x = torch.randn(N, D, H, W, 32, generator=torch.Generator().manual_seed(5)).to(DEV)
    nblk = ops.conv3d_grid(N, D, H, W)
    res = {}
    for arith in ("f32", "bf16x3"):
        part = torch.full((nblk * 4,), float("nan"), device=DEV, dtype=torch.float64) if stats else None
        out = _run(plan, arith, x, dims, out=torch.empty(N, D, H, W, 16, device=DEV), out_stride=16, stats_partials=part)
        st = ops.groupnorm_finalize(part, nblk, 16.0 * N * D * H * W, 1e-5) if stats else None
        res[arith] = (out, st)
    a, b = res["f32"][0], res["bf16x3"][0]
    assert (a - b).abs().max().item() < 3e-6 * max(1.0, a.abs().max().item())
    if stats:
        np.testing.assert_allclose(res["bf16x3"][1].cpu().numpy()[:2], res["f32"][1].cpu().numpy()[:2], rtol=2e-5, atol=1e-6)


def test_split_kernels_are_deterministic_at_full_size():
    """race detector: the LDS pipelines (ring refill, weight hand-over, per-tap barriers) must give bit-identical results on
    repeated launches of the cfg2-size problem (a missing barrier shows up as run-to-run differences)."""
    from estdepth_amd import ops, synth
    from estdepth_amd.backbones import conv_bn2d
    mod, plan = _plan(91)
    dims = (3, 64, 120, 160)
    x = torch.randn(*dims, 32, device=DEV, generator=torch.Generator(device=DEV).manual_seed(8))
    nblk = ops.conv3d_grid(*dims)
    ref, ref_st = None, None
    for _ in range(6):
        part = torch.zeros(nblk * 4, device=DEV, dtype=torch.float64)
        out = _run(plan, "bf16x3", x, dims, residual=x, stats_partials=part)
        if ref is None:
            ref, ref_st = out.clone(), part.clone()
        else:
            assert torch.equal(out, ref) and torch.equal(part, ref_st)
    c2 = conv_bn2d(64, 64, 3, 1, 1, 1).eval()
    synth.fill_state_dict(c2, seed=3)
    c2 = c2.to(DEV)
    p2 = ops.Conv2dPlan(c2[0], c2[1], relu_before=True)
    x2 = torch.randn(5, 120, 160, 64, device=DEV, generator=torch.Generator(device=DEV).manual_seed(9))
    ops.CONV2D_ARITH = "bf16x3"
    try:
        outs = [p2.run(x2).clone() for _ in range(6)]
        torch.cuda.synchronize()
    finally:
        ops.CONV2D_ARITH = "f32"
    assert all(torch.equal(outs[0], o) for o in outs[1:])
