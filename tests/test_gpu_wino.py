"""estd_conv3d_k3_wino (32 -> 32, depth axis in Winograd F(2,3) form on fp32 MFMA; csrc/conv3d_wino.hip) against
  * an fp64 convolution of the same fp32 data (how much error does the transform add to the direct kernel's?),
  * the direct fp32 MFMA kernel on every epilogue feature the plain instance has and on ragged shapes (odd D, D = 1, partial
    tiles, batches), including the GroupNorm partial sums,
  * the CPU oracle."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = torch.device("cuda:0")


@pytest.fixture(scope="module", autouse=True)
def _need_gpu():
    if not torch.cuda.is_available():
        pytest.fail("GPU tests need an MI355X (no CPU path exists)")


def _plan(seed, act="relu"):
    from estdepth_amd import synth
    from estdepth_amd.layers_op import ConvBN3d
    mod = ConvBN3d(32, 32, 3, 1, 1, act).eval()
    synth.fill_state_dict(mod, seed=seed)
    mod = mod.to(DEV)
    return mod, mod.plan()


def _run(plan, algo, x, dims, **kw):
    from estdepth_amd import ops
    old, old_x, old_3, old_3x = ops.CONV3D_ALGO, ops.W2X, ops.W3, ops.W3_EXTRA
    ops.W3_EXTRA = algo == "wino3"                      # (the three-axis kernel's scalar-channel instance: the default of the 33 -> 32 launch, ESTD_W3_EXTRA=1)
    # "wino2x" = the two-axis form on the operand-reuse kernel (csrc/conv3d_wino2x.hip: ESTD_BUILD_AB=1 builds only, ESTD_W2X=1) for the plain 32 -> 32 instance;
    # "wino3" = all three axes in Winograd form (csrc/conv3d_wino3.hip, the default, ESTD_W3=1: 32 -> 32 with every read-back epilogue, 32 -> 32 + GroupNorm
    # partials and 33 -> 32 without read-back streams); "wino2" = the same launches on the two-axis kernel (ESTD_W3=0)
    ops.CONV3D_ALGO, ops.W2X, ops.W3 = ("wino2", True, False) if algo == "wino2x" else ("wino2", False, True) if algo == "wino3" else (algo, False, False)
    try:
        out = kw.pop("out", None)
        if out is None:
            out = torch.empty(dims + (kw.get("out_stride", 32),), device=DEV)
        plan.run(x, dims, out=out, **kw)
        torch.cuda.synchronize()
        return out
    finally:
        ops.CONV3D_ALGO, ops.W2X, ops.W3, ops.W3_EXTRA = old, old_x, old_3, old_3x


def _has_ab():
    from estdepth_amd import _native
    try:
        return _native.has_ab()
    except RuntimeError:
        return False


# depth and row axis in Winograd form (default); "wino" = depth axis only (csrc/conv3d_wino.hip: part of the ESTD_BUILD_AB=1 build)
ALGOS = ("wino2",) + (("wino",) if _has_ab() else ())
# instances without a scalar channel: + the three-axis kernel (default), + the operand-reuse kernel (superseded: ESTD_BUILD_AB=1 builds only)
ALGOS_PLAIN = ALGOS + ("wino3",) + (("wino2x",) if _has_ab() else ())


@pytest.mark.parametrize("algo", ALGOS_PLAIN)
@pytest.mark.parametrize("dims,scale", [((1, 6, 19, 45), 1.0), ((2, 3, 8, 32), 100.0), ((1, 1, 5, 7), 1e-3), ((1, 64, 24, 32), 1.0)])
def test_wino_error_vs_fp64_is_at_the_direct_kernels_level(dims, scale, algo):
    mod, plan = _plan(11, act=None)
    N, D, H, W = dims
    x = torch.randn(N, D, H, W, 32, generator=torch.Generator().manual_seed(5)) * scale
    w64 = mod[0].weight.detach().double().cpu()
    bn = mod[1]
    sc = (bn.weight / torch.sqrt(bn.running_var + bn.eps)).double().cpu()
    sh = bn.bias.double().cpu() - bn.running_mean.double().cpu() * sc
    ref = torch.nn.functional.conv3d(x.permute(0, 4, 1, 2, 3).double(), w64, padding=1)
    ref = (ref * sc[None, :, None, None, None] + sh[None, :, None, None, None]).permute(0, 2, 3, 4, 1)
    xd = x.to(DEV)
    e_dir = (_run(plan, "direct", xd, dims).double().cpu() - ref).abs().max().item()
    e_win = (_run(plan, algo, xd, dims).double().cpu() - ref).abs().max().item()
    mag = ref.abs().max().item()
    print("fp64 check dims=%s scale=%g: |ref|max %.3g  err direct %.3g  err %s %.3g" % (dims, scale, mag, e_dir, algo, e_win))
    assert e_win <= 3.0 * e_dir + 1e-7 * mag, (e_win, e_dir)
    assert e_win < 3e-6 * mag


@pytest.mark.parametrize("algo", ALGOS_PLAIN)
@pytest.mark.parametrize("dims", [(1, 4, 8, 32), (3, 5, 13, 50), (1, 2, 120, 160), (2, 1, 9, 33), (1, 7, 8, 16), (1, 70, 8, 32)])
def test_wino_matches_direct_kernel_all_epilogues(dims, algo):
    """ReLU / none, residual, two residuals + scale, running accumulation, strided output, GroupNorm partial sums."""
    from estdepth_amd import ops
    N, D, H, W = dims
    g = torch.Generator().manual_seed(sum(dims))
    x = torch.randn(N, D, H, W, 32, generator=g).to(DEV)
    r1 = torch.randn(N, D, H, W, 32, generator=g).to(DEV)
    r2 = torch.randn(N, D, H, W, 32, generator=g).to(DEV)
    for act in ("relu", None):
        _, plan = _plan(3, act=act)
        cases = [dict(), dict(residual=r1), dict(residual=r1, residual2=r2, out_scale=0.5)]
        for kw in cases:
            a = _run(plan, "direct", x, dims, **kw)
            b = _run(plan, algo, x, dims, **kw)
            tol = 5e-6 * max(1.0, float(a.abs().max()))
            assert float((a - b).abs().max()) < tol, (act, sorted(kw), float((a - b).abs().max()))
        # running accumulation (mean over source views): out += result
        base = torch.randn(N, D, H, W, 32, generator=g).to(DEV)
        a = _run(plan, "direct", x, dims, out=base.clone(), accumulate=True, out_scale=0.5)
        b = _run(plan, algo, x, dims, out=base.clone(), accumulate=True, out_scale=0.5)
        assert float((a - b).abs().max()) < 5e-6 * max(1.0, float(a.abs().max()))
    # GroupNorm partial sums (the GRU gate convolution: bias, no activation, N = 1 volume at a time)
    _, plan = _plan(9, act=None)
    nblk = ops.conv3d_grid(N, D, H, W)
    pa = torch.zeros(nblk * 4, device=DEV, dtype=torch.float64)
    pb = torch.zeros(nblk * 4, device=DEV, dtype=torch.float64)
    a = _run(plan, "direct", x, dims, stats_partials=pa)
    b = _run(plan, algo, x, dims, stats_partials=pb)
    assert float((a - b).abs().max()) < 5e-6 * max(1.0, float(a.abs().max()))
    sa = ops.groupnorm_finalize(pa, nblk, 16.0 * N * D * H * W).cpu()
    sb = ops.groupnorm_finalize(pb, nblk, 16.0 * N * D * H * W).cpu()
    assert float((sa - sb).abs().max()) < 1e-5 * max(1.0, float(sa.abs().max()))
    # every tile wrote its partials (none left at the initial zero in BOTH sum slots)
    assert int((pb.view(-1, 4)[:, 1] == 0).sum()) == 0


@pytest.mark.parametrize("algo", ALGOS_PLAIN)
def test_wino_vs_oracle_ragged(algo):
    from oracle import ref_ops as O
    mod, plan = _plan(21, act="relu")
    dims = (2, 5, 11, 19)
    x = torch.randn(*dims, 32, generator=torch.Generator().manual_seed(2))
    bn = mod[1]
    ref = O.bn_act(O.conv3d(x.permute(0, 4, 1, 2, 3).numpy(), mod[0].weight.detach().cpu().numpy()),
                   (bn.weight.detach().cpu().numpy(), bn.bias.detach().cpu().numpy(), bn.running_mean.cpu().numpy(),
                    bn.running_var.cpu().numpy()), "relu")
    out = _run(plan, algo, x.to(DEV), dims).cpu().numpy()
    assert np.abs(np.moveaxis(out, -1, 1) - ref).max() < 2e-5


@pytest.mark.parametrize("algo", ALGOS_PLAIN)
def test_wino_full_size_linearity_and_match(algo):
    """BASELINE configs[1] size (3 volumes of 64x120x160): against the direct kernel and a linearity property."""
    _, plan = _plan(5, act=None)
    dims = (3, 64, 120, 160)
    x = torch.randn(*dims, 32, device=DEV, generator=torch.Generator(device=DEV).manual_seed(1))
    a = _run(plan, "direct", x, dims)
    b = _run(plan, algo, x, dims)
    assert float((a - b).abs().max()) < 5e-6 * float(a.abs().max())
    del a
    zero = _run(plan, algo, torch.zeros_like(x), dims)              # = folded shift
    c = _run(plan, algo, x * -2.0, dims)
    assert float((c - zero + 2.0 * (b - zero)).abs().max()) < 2e-5 * float(b.abs().max())


@pytest.mark.parametrize("algo", ALGOS + ("wino3",))
@pytest.mark.parametrize("dims", [(1, 4, 8, 32), (3, 5, 13, 50), (2, 1, 9, 33), (1, 64, 24, 32)])
def test_wino_extra_input_channel_matches_direct_kernel_and_fp64(dims, algo):
    """the key || value convolution's shape: 32 channels-last inputs + a scalar 33rd input volume -> 32 outputs (ReLU)."""
    from estdepth_amd import ops
    N, D, H, W = dims
    g = torch.Generator().manual_seed(7 + sum(dims))
    w = torch.randn(32, 33, 3, 3, 3, generator=g) * 0.05
    sc, sh = torch.rand(32, generator=g) + 0.5, torch.randn(32, generator=g) * 0.1
    plan = ops.Conv3dPlan(w, list(range(32)), 32, list(range(32)), 2, sc, sh, act_a="relu", device=DEV)
    assert plan.w_wino2_extra is not None and (plan.w_wino_extra is not None or not _has_ab())
    x = torch.randn(N, D, H, W, 32, generator=g)
    e = torch.randn(N, D, H, W, generator=g)
    a = _run(plan, "direct", x.to(DEV), dims, in_extra=e.to(DEV))
    b = _run(plan, algo, x.to(DEV), dims, in_extra=e.to(DEV))
    full = torch.cat([x.permute(0, 4, 1, 2, 3), e[:, None]], 1).double()
    ref = torch.nn.functional.conv3d(full, w.double(), padding=1) * sc.double()[None, :, None, None, None] + sh.double()[None, :, None, None, None]
    ref = torch.relu(ref).permute(0, 2, 3, 4, 1)
    mag = float(ref.abs().max())
    e_dir, e_win = float((a.double().cpu() - ref).abs().max()), float((b.double().cpu() - ref).abs().max())
    assert e_win <= 3.0 * e_dir + 1e-7 * mag and e_win < 3e-6 * mag, (e_dir, e_win, mag)
    assert float((a - b).abs().max()) < 5e-6 * max(1.0, mag)


@pytest.mark.parametrize("algo", ALGOS + ("wino3",))
@pytest.mark.parametrize("dims", [(1, 4, 8, 32), (3, 5, 13, 50), (2, 1, 9, 33), (1, 64, 24, 32), (1, 3, 120, 160), (2, 17, 37, 21)])
def test_wino_33_to_33_matches_direct_kernel_and_fp64(dims, algo):
    """dres2's shape: input = [scalar channel 0 | 32 channels-last], output = 32 channels-last + a scalar 33rd volume (ReLU).  wino2:
    the XOUT instance of the 2-axis kernel (the 33rd output channel on the VALU from the row-transformed fragments, split over the
    two waves of a SIMD, cross-wave sum through LDS).  wino3 (the default, round 6): the 32 main outputs on the three-axis kernel's 33 -> 32
    instance + output channel 32 as a pass of its own with the taps as matrix rows (csrc/conv3d_xout.hip); ragged tiles, depth segments
    that end inside a 16-plane segment, a single plane."""
    from estdepth_amd import ops
    N, D, H, W = dims
    g = torch.Generator().manual_seed(3 + sum(dims))
    w = torch.randn(33, 33, 3, 3, 3, generator=g) * 0.05
    sc, sh = torch.rand(33, generator=g) + 0.5, torch.randn(33, generator=g) * 0.1
    plan = ops.Conv3dPlan(w, list(range(1, 33)), 0, list(range(33)), 3, sc, sh, act_a="relu", device=DEV)
    assert plan.w_wino2_xout is not None and (plan.w_wino_xout is not None or not _has_ab())
    x = torch.randn(N, D, H, W, 32, generator=g)
    e = torch.randn(N, D, H, W, generator=g)
    outs = {}
    for a_ in ("direct", algo):
        ex = torch.full((N, D, H, W), float("nan"), device=DEV)
        o = _run(plan, a_, x.to(DEV), dims, in_extra=e.to(DEV), out_extra=ex)
        outs["direct" if a_ == "direct" else "wino"] = torch.cat([o, ex[..., None]], -1)
    full = torch.cat([e[:, None], x.permute(0, 4, 1, 2, 3)], 1).double()
    ref = torch.nn.functional.conv3d(full, w.double(), padding=1) * sc.double()[None, :, None, None, None] + sh.double()[None, :, None, None, None]
    ref = torch.relu(ref).permute(0, 2, 3, 4, 1)
    mag = float(ref.abs().max())
    e_dir, e_win = float((outs["direct"].double().cpu() - ref).abs().max()), float((outs["wino"].double().cpu() - ref).abs().max())
    assert e_win <= 3.0 * e_dir + 1e-7 * mag and e_win < 3e-6 * mag, (e_dir, e_win, mag)
    assert float((outs["direct"] - outs["wino"]).abs().max()) < 5e-6 * max(1.0, mag)


def test_reserved_cus_changes_the_partition_not_the_result():
    """estd_set_reserved_cus (persistent grids leave CUs to a concurrent collective, N > 1): 248- and 128-CU grids give
    bit-identical outputs and GroupNorm partial sums on both conv3d kernels and the conv2d kernel; the setting round-trips."""
    from estdepth_amd import ops
    _, plan = _plan(13, act=None)
    dims = (2, 9, 40, 70)
    x = torch.randn(*dims, 32, generator=torch.Generator().manual_seed(4)).to(DEV)
    conv = torch.nn.Conv2d(64, 64, 3, 1, 1, bias=False).to(DEV)
    bn = torch.nn.BatchNorm2d(64).to(DEV).eval()
    p2 = ops.Conv2dPlan(conv, bn, relu_before=True)
    x2 = torch.randn(3, 120, 160, 64, generator=torch.Generator().manual_seed(5)).to(DEV)
    nblk = ops.conv3d_grid(*dims)
    res = {}
    try:
        for r in (0, 8, 5, 128, 1000):
            eff = ops.set_reserved_cus(r)
            assert eff == {0: 0, 8: 8, 5: 8, 128: 128, 1000: 128}[r]
            part = torch.zeros(nblk * 4, device=DEV, dtype=torch.float64)
            res[r] = (_run(plan, "wino2", x, dims, stats_partials=part), part, _run(plan, "direct", x, dims), p2.run(x2))
    finally:
        ops.set_reserved_cus(0)
    for r in (8, 128):
        for a, b in zip(res[0], res[r]):
            assert torch.equal(a, b)


def test_fuzz_winograd_kernels_against_direct_kernels():
    """tools/fuzz_convs.py for 15 s (~400 random cases): ragged shapes, batches, every instance and epilogue flag of the 3D and
    2D Winograd kernels against the direct kernels of the same operator."""
    import os, subprocess, sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, os.path.join(root, "tools", "fuzz_convs.py"), "15", "7"], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0 and "random cases agree" in r.stdout, (r.stdout[-500:], r.stderr[-500:])


@pytest.mark.ab
def test_wino_rejects_other_shapes():
    from estdepth_amd import _native
    d = _native.Conv3dDesc()
    d.N = d.D = d.H = d.W = 4
    assert _native.lib().estd_conv3d_k3_wino(d, None) == -1                 # null pointers
    x = torch.zeros(4, 4, 4, 32, device=DEV)
    d.in_main = d.out_main = d.w_wino = d.scale = d.shift = x.data_ptr()
    d.cin_main, d.n_tiles, d.in_stride, d.out_stride = 16, 1, 32, 32
    assert _native.lib().estd_conv3d_k3_wino(d, None) == -3                 # not the plain 32 -> 32 instance


@pytest.mark.parametrize("dims", [(1, 4, 8, 32), (2, 5, 13, 50), (1, 1, 9, 33), (1, 64, 24, 32), (1, 7, 120, 160)])
def test_wino2_16_output_channels_matches_direct_kernel_and_fp64(dims):
    """the GRU output convolution's shape (32 -> 16, bias, no activation, GroupNorm partial sums; epipolar_transformer.py:26): the
    16-output-channel instance of the 2-axis Winograd kernel (input channels split over the two waves of a SIMD, cross-wave
    reduction through LDS) against the direct kernel and an fp64 convolution, incl. odd D and partial tiles."""
    from estdepth_amd import ops
    N, D, H, W = dims
    g = torch.Generator().manual_seed(11 + sum(dims))
    w = torch.randn(16, 32, 3, 3, 3, generator=g) * 0.05
    sc, sh = torch.rand(16, generator=g) + 0.5, torch.randn(16, generator=g) * 0.1
    plan = ops.Conv3dPlan(w, list(range(32)), None, list(range(16)), 1, sc, sh, device=DEV)
    assert plan.w_wino2_o16 is not None
    x = torch.randn(N, D, H, W, 32, generator=g)
    nblk = ops.conv3d_grid(N, D, H, W)
    outs, parts = {}, {}
    for algo in ("direct", "wino2"):
        for with_stats in (False, True):
            part = torch.zeros(nblk * 4, device=DEV, dtype=torch.float64) if with_stats else None
            o = torch.full((N, D, H, W, 16), float("nan"), device=DEV)
            _run(plan, algo, x.to(DEV), dims, out=o, out_stride=16, stats_partials=part)
            outs[(algo, with_stats)], parts[algo] = o, part
    ref = torch.nn.functional.conv3d(x.permute(0, 4, 1, 2, 3).double(), w.double(), padding=1) * sc.double()[None, :, None, None, None] \
        + sh.double()[None, :, None, None, None]
    ref = ref.permute(0, 2, 3, 4, 1)
    mag = float(ref.abs().max())
    e_dir = float((outs[("direct", False)].double().cpu() - ref).abs().max())
    for with_stats in (False, True):
        e_win = float((outs[("wino2", with_stats)].double().cpu() - ref).abs().max())
        assert e_win <= 3.0 * e_dir + 1e-7 * mag and e_win < 3e-6 * mag, (with_stats, e_dir, e_win, mag)
    if N == 1:      # GroupNorm statistics (one volume per launch, as the ConvGRU calls it)
        sa = ops.groupnorm_finalize(parts["direct"], nblk, 16.0 * N * D * H * W).cpu()
        sb = ops.groupnorm_finalize(parts["wino2"], nblk, 16.0 * N * D * H * W).cpu()
        assert float((sa[:2] - sb[:2]).abs().max()) < 1e-5 * max(1.0, float(sa[:2].abs().max())), (sa, sb)


def _head_plan(seed, act="relu"):
    from estdepth_amd import ops
    g = torch.Generator().manual_seed(seed)
    w = torch.randn(16, 16, 3, 3, 3, generator=g) * 0.08
    sc, sh = torch.rand(16, generator=g) + 0.5, torch.randn(16, generator=g) * 0.3
    hw, hb = torch.randn(16, generator=g), torch.randn(1, generator=g)
    plan = ops.Conv3dPlan(w, list(range(16)), None, list(range(16)), 1, sc, sh, act_a=act, head_w=hw, head_b=hb, device=DEV)
    return plan, (w, sc, sh, hw, hb)


@pytest.mark.parametrize("dims,stride,scale", [((1, 6, 19, 45), 32, 1.0), ((2, 3, 16, 16), 16, 100.0), ((1, 1, 5, 7), 32, 1e-3), ((1, 7, 33, 18), 16, 1.0),
                                              ((3, 5, 17, 31), 32, 1.0), ((1, 64, 30, 40), 32, 1.0)])
@pytest.mark.parametrize("act", ["relu", "none"])
def test_stereo_head_c16_wino2_vs_fp64_and_the_direct_kernel(dims, stride, scale, act):
    """csrc/conv3d_wino2_c16.hip (16 -> 16 + BN + activation + 1x1x1 head, only the logit volume is written; stereo_head0/1,
    hybrid_depth_decoder.py:96-112) against an fp64 evaluation of the same fp32 data and against the direct kernel: ragged tiles
    (H, W not multiples of 16), odd D, D = 1, batches, both record strides (16 and the 32 of the key|value records)."""
    from estdepth_amd import ops
    plan, (w, sc, sh, hw, hb) = _head_plan(sum(dims) + stride, act)
    assert plan.w_wino2_c16 is not None
    N, D, H, W = dims
    x = torch.randn(N, D, H, W, stride, generator=torch.Generator().manual_seed(7)) * scale
    y = torch.nn.functional.conv3d(x[..., :16].permute(0, 4, 1, 2, 3).double(), w.double(), padding=1)
    y = y * sc.double()[None, :, None, None, None] + sh.double()[None, :, None, None, None]
    if act == "relu":
        y = y.clamp_min(0)
    ref = (y * hw.double()[None, :, None, None, None]).sum(1) + hb.double()
    xd = x.to(DEV)
    outs = {}
    old = ops.CONV3D_ALGO
    try:
        for algo in ("direct", "wino2"):
            ops.CONV3D_ALGO = algo
            lg = torch.full((N, D, H, W), float("nan"), device=DEV)
            plan.run(xd, dims, in_stride=stride, out_head=lg)
            torch.cuda.synchronize()
            outs[algo] = lg.double().cpu()
    finally:
        ops.CONV3D_ALGO = old
    mag = ref.abs().max().item()
    e_dir, e_win = (outs["direct"] - ref).abs().max().item(), (outs["wino2"] - ref).abs().max().item()
    print("head fp64 check dims=%s stride=%d scale=%g act=%s: |ref|max %.3g  err direct %.3g  err wino2-c16 %.3g" % (dims, stride, scale, act, mag, e_dir, e_win))
    assert not torch.isnan(outs["wino2"]).any()                      # every voxel written (ragged tiles, odd D)
    assert e_win <= 3.0 * e_dir + 2e-7 * mag, (e_win, e_dir)
    assert e_win < 3e-6 * mag


def test_stereo_head_c16_is_the_default_and_refuses_what_it_cannot_do():
    from estdepth_amd import ops, _native
    plan, _ = _head_plan(5)
    d = _native.Conv3dDesc()
    d.N, d.D, d.H, d.W = 1, 4, 16, 16
    d.cin_main, d.in_stride, d.n_tiles = 16, 16, 1
    x = torch.zeros(1, 4, 16, 16, 16, device=DEV)
    lg = torch.zeros(1, 4, 16, 16, device=DEV)
    o = torch.zeros(1, 4, 16, 16, 16, device=DEV)
    d.in_main, d.w_wino2 = x.data_ptr(), plan.w_wino2_c16.data_ptr()
    d.scale, d.shift = plan.scale.data_ptr(), plan.shift.data_ptr()
    d.act_a = d.act_b = ops.ACT["relu"]
    d.out_scale = 1.0
    lib = _native.lib()
    assert lib.estd_conv3d_k3_wino2(d, None) == -1                               # no head: ESTD_ERR_ARG
    d.head_w, d.head_b, d.out_head = plan.head_w.data_ptr(), plan.head_b.data_ptr(), lg.data_ptr()
    assert lib.estd_conv3d_k3_wino2(d, None) == 0
    d.out_main, d.out_stride, d.out_channels = o.data_ptr(), 16, 16
    assert lib.estd_conv3d_k3_wino2(d, None) == -3                               # a 16-channel main output: ESTD_ERR_UNSUPPORTED (direct kernel)
    torch.cuda.synchronize()
