"""GPU: the MFMA conv2d kernel and the fused PSM path against PyTorch fp32 references of the same ops."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = "cuda"


@pytest.mark.parametrize("cin,cout,dil,dims,relu,res", [(32, 32, 1, (2, 13, 21), True, False), (64, 64, 1, (1, 24, 32), False, True),
                                                        (128, 128, 2, (1, 17, 35), True, False), (96, 64, 1, (1, 8, 16), False, False),
                                                        (320, 128, 1, (1, 9, 20), True, False)])
def test_conv2d_mfma_vs_torch_fp64(cin, cout, dil, dims, relu, res):
    from estdepth_amd import synth, ops
    from estdepth_amd.backbones import conv_bn2d
    N, H, W = dims
    mod = conv_bn2d(cin, cout, 3, 1, dil, dil).eval()
    synth.fill_state_dict(mod, seed=cin + cout + dil)
    g = torch.Generator().manual_seed(cin * 3 + cout)
    x = torch.randn(N, cin, H, W, generator=g)
    r = torch.randn(N, cout, H, W, generator=g) if res else None
    with torch.no_grad():
        ref = mod.double()(x.double())
        if relu:
            ref = torch.relu(ref)
        if res:
            ref = ref + r.double()
    mod = mod.float().to(DEV)
    plan = ops.Conv2dPlan(mod[0], mod[1], relu_before=relu)
    xin = x.to(DEV).permute(0, 2, 3, 1).contiguous()
    rin = r.to(DEV).permute(0, 2, 3, 1).contiguous() if res else None
    out = plan.run(xin, residual=rin).permute(0, 3, 1, 2).cpu().double()
    assert (out - ref).abs().max().item() < 2e-5 * max(1.0, ref.abs().max().item())


@pytest.mark.parametrize("c,dims", [(64, (3, 120, 160)), (128, (3, 60, 80)), (256, (3, 30, 40))])
def test_resnet50_stride1_3x3_on_the_mfma_kernel_vs_torch_fp64(c, dims):
    """the stride-1 3x3 convolutions of ResNet-50 (resnet_encoder.py:43-49) at their cfg2 shapes through
    backbones.conv_bn_act with enable_hip_3x3: Conv + folded BN + ReLU in the MFMA conv2d kernel vs torch fp64; and the
    residual form relu(bn(conv(x)) + r) of the basic block (ReLU AFTER the add)."""
    from estdepth_amd import synth
    from estdepth_amd.backbones import conv_bn_act, enable_hip_3x3, _hip_3x3_plan
    N, H, W = dims
    conv = torch.nn.Conv2d(c, c, 3, 1, 1, bias=False)
    bn = torch.nn.BatchNorm2d(c).eval()
    seq = torch.nn.Sequential(conv, bn).eval()
    synth.fill_state_dict(seq, seed=c)
    g = torch.Generator().manual_seed(c)
    x = torch.randn(N, c, H, W, generator=g)
    r = torch.randn(N, c, H, W, generator=g)
    with torch.no_grad():
        y64 = seq.double()(x.double())
        ref_plain, ref_res = torch.relu(y64), torch.relu(y64 + r.double())
    seq = seq.float().to(DEV).to(memory_format=torch.channels_last)
    enable_hip_3x3(seq)
    xg = x.to(DEV).contiguous(memory_format=torch.channels_last)
    rg = r.to(DEV).contiguous(memory_format=torch.channels_last)
    assert _hip_3x3_plan(seq[0], seq[1], xg, True, False) is not None          # the MFMA kernel is what runs
    with torch.no_grad():
        out_plain = conv_bn_act(seq[0], seq[1], xg, relu=True)
        out_res = conv_bn_act(seq[0], seq[1], xg, relu=True, residual=rg)
    for out, ref in ((out_plain, ref_plain), (out_res, ref_res)):
        assert tuple(out.shape) == (N, c, H, W)
        assert (out.cpu().double() - ref).abs().max().item() < 2e-5 * max(1.0, ref.abs().max().item())
    small = torch.randn(1, c, 15, 20, device=DEV).contiguous(memory_format=torch.channels_last)
    assert _hip_3x3_plan(seq[0], seq[1], small, True, False) is None          # too few tiles for 256 CUs: library convolution


def test_psm_hip_path_matches_torch_path():
    from estdepth_amd import synth
    from estdepth_amd.backbones import PSMFeatures
    m = PSMFeatures().eval()
    synth.fill_state_dict(m, seed=21)
    x = torch.randn(2, 3, 128, 160, generator=torch.Generator().manual_seed(4))
    with torch.no_grad():
        ref = m(x)                                   # CPU fp32 torch path (what the oracle-side tests use)
        out = m.to(DEV).use_hip_convs()(x.to(DEV)).cpu()
    assert out.shape == ref.shape
    assert (out - ref).abs().max().item() < 1e-4 * max(1.0, ref.abs().max().item())


def test_e2e_golden_with_hip_psm(golden_dir):
    """streaming golden (G8) with the PSM 3x3 convs on the MFMA kernel: depth still within 1e-4 of the reference."""
    import os
    import fixtures_spec as S
    from estdepth_amd import synth, DepthNetHybrid
    g = np.load(os.path.join(golden_dir, "g8_estm_stream.npz"))
    m = DepthNetHybrid(ndepths=64, depth_min=0.1, depth_max=10.0, resnet=18, IF_EST_transformer=True).eval()
    synth.fill_state_dict(m, seed=2, head_gain=1.0)
    m = m.to(DEV).use_hip_psm()
    imgs, poses, intr, sample = S.e2e_inputs(6, S.E2E_HI, S.E2E_WI, seed=1003)
    imgs, poses, intr = imgs.to(DEV), poses.to(DEV), intr.to(DEV)
    mem_costs, mem_poses = [], []
    for w in range(3):
        sl = slice(w, w + 3)
        pc = {"keys": [c["keys"][0] for c in mem_costs], "values": [c["values"][0] for c in mem_costs]} if mem_costs else None
        pp = [p[0] for p in mem_poses] if mem_poses else None
        with torch.no_grad():
            outputs, costs, cposes = m(imgs[:, sl], poses[:, sl], intr, {k: v[:, sl] for k, v in sample.items()}, pc, pp, mode="val")
        mem_costs.append(costs); mem_poses.append(cposes)
        mem_costs, mem_poses = mem_costs[-2:], mem_poses[-2:]
        for k, v in outputs.items():
            name = "w%d|" % w + "|".join(map(str, k))
            assert np.abs(v.cpu().numpy() - g[name]).max() < 1e-4, name


def test_overlapped_semantic_branch_and_all_accelerators_match_plain_path():
    """use_hip_psm + fuse_bn_2d + overlap_semantic_branch + hipGraph replay (what bench.py runs) == plain eager forward."""
    import fixtures_spec as S
    from estdepth_amd import synth, DepthNetHybrid
    from estdepth_amd.graph import GraphedForward
    def make():
        m = DepthNetHybrid(ndepths=64, depth_min=0.1, depth_max=10.0, resnet=18, IF_EST_transformer=True).eval()
        synth.fill_state_dict(m, seed=2, head_gain=1.0)
        return m.to(DEV)
    plain, fast = make().plain_path(), make()            # a model on a ROCm device runs every accelerator by default
    assert fast._channels_last_2d and fast._overlap_semantic and not plain._channels_last_2d and not plain._overlap_semantic
    gf = GraphedForward(fast)
    imgs, poses, intr, sample = S.e2e_inputs(8, S.E2E_HI, S.E2E_WI, seed=1004)
    imgs, poses, intr = imgs.to(DEV), poses.to(DEV), intr.to(DEV)
    pc_a = pp_a = pc_b = pp_b = None
    for call in range(2):
        sl = slice(3 * call, 3 * call + 5)
        smp = {k: v[:, sl].to(DEV) for k, v in sample.items()}
        with torch.no_grad():
            a, pc_a, pp_a = plain(imgs[:, sl], poses[:, sl], intr, smp, pc_a, pp_a, mode="val")
            b, cb, pb = gf(imgs[:, sl], poses[:, sl], intr, smp, pc_b, pp_b, mode="val")
            for k in a:
                assert (a[k] - b[k]).abs().max().item() < 5e-5, (call, k)
            # the graph's outputs are static buffers: detach the memory we carry to the next call
            pc_b = {"keys": [cb["keys"][0].contiguous().clone()], "values": [cb["values"][0].contiguous().clone()]}
            pp_b = [pb[0].clone()]


@pytest.mark.parametrize("shape,relu,res", [((2, 64, 13, 21), True, True), ((1, 32, 5, 7), False, False), ((3, 256, 8, 8), True, False)])
def test_bn_act_nhwc_matches_torch(shape, relu, res):
    from estdepth_amd import ops, synth
    from estdepth_amd.backbones import _folded
    bn = torch.nn.BatchNorm2d(shape[1]).eval()
    synth.fill_state_dict(bn, seed=shape[1])
    g = torch.Generator().manual_seed(3)
    x = torch.randn(*shape, generator=g)
    r = torch.randn(*shape, generator=g) if res else None
    with torch.no_grad():
        ref = bn.double()(x.double())
        if res:
            ref = ref + r.double()
        if relu:
            ref = torch.relu(ref)
    bn = bn.float().to(DEV)
    sc, sh = _folded(bn)
    y = x.to(DEV).contiguous(memory_format=torch.channels_last)
    out = ops.bn_act_nhwc_(y, sc, sh, relu, r.to(DEV).contiguous(memory_format=torch.channels_last) if res else None)
    assert out.data_ptr() == y.data_ptr()
    assert (out.double().cpu() - ref).abs().max().item() < 1e-5
    with pytest.raises(RuntimeError, match="channels_last"):
        ops.bn_act_nhwc_(x.to(DEV), sc, sh, relu)


def test_e2e_golden_with_fused_bn(golden_dir):
    """cfg1 golden (R18, EST off) and the Joint carry golden (R18, EST on) with fuse_bn_2d + use_hip_psm: depth within 1e-4."""
    import os
    import fixtures_spec as S
    from estdepth_amd import synth, DepthNetHybrid
    g = np.load(os.path.join(golden_dir, "g9_joint_carry.npz"))
    m = DepthNetHybrid(ndepths=64, depth_min=0.1, depth_max=10.0, resnet=18, IF_EST_transformer=True).eval()
    synth.fill_state_dict(m, seed=2, head_gain=1.0)
    m = m.to(DEV).use_hip_psm().fuse_bn_2d()
    imgs, poses, intr, sample = S.e2e_inputs(8, S.E2E_HI, S.E2E_WI, seed=1004)
    imgs, poses, intr = imgs.to(DEV), poses.to(DEV), intr.to(DEV)
    pc = pp = None
    for call in range(2):
        sl = slice(3 * call, 3 * call + 5)
        with torch.no_grad():
            outputs, pc, pp = m(imgs[:, sl], poses[:, sl], intr, {k: v[:, sl] for k, v in sample.items()}, pc, pp, mode="val")
        for k, v in outputs.items():
            name = "c%d|" % call + "|".join(map(str, k))
            if name in g.files:
                assert np.abs(v.cpu().numpy() - g[name]).max() < 1e-4, name


@pytest.mark.parametrize("cin,cout,dims,relu,res,dil", [(32, 32, (2, 13, 21), True, False, 1), (64, 64, (1, 24, 32), False, True, 1),
                                                        (96, 64, (1, 8, 16), False, False, 1), (320, 128, (1, 9, 40), True, False, 1),
                                                        (64, 64, (5, 120, 160), False, True, 1), (128, 128, (1, 17, 35), True, False, 2),
                                                        (32, 64, (2, 8, 16), False, True, 2), (128, 128, (5, 120, 160), True, False, 2)])
@pytest.mark.ab
def test_conv2d_split_vs_fp64_and_fp32_kernel(cin, cout, dims, relu, res, dil):
    """3xbf16 split conv2d: fp32-level error against an fp64 convolution, agreement with the fp32 MFMA kernel."""
    from estdepth_amd import synth, ops
    from estdepth_amd.backbones import conv_bn2d
    N, H, W = dims
    mod = conv_bn2d(cin, cout, 3, 1, dil, dil).eval()
    synth.fill_state_dict(mod, seed=cin + cout)
    g = torch.Generator().manual_seed(cin * 3 + cout)
    x = torch.randn(N, cin, H, W, generator=g)
    r = torch.randn(N, cout, H, W, generator=g) if res else None
    mod = mod.to(DEV)
    plan = ops.Conv2dPlan(mod[0], mod[1], relu_before=relu, relu_after=res)
    xin = x.to(DEV).permute(0, 2, 3, 1).contiguous()
    rin = r.to(DEV).permute(0, 2, 3, 1).contiguous() if res else None
    outs = {}
    old_algo = ops.CONV2D_ALGO
    for arith in ("f32", "bf16x3"):
        ops.CONV2D_ARITH = arith
        ops.CONV2D_ALGO = "direct"           # the split kernel is a direct-form kernel: its error is held against the direct fp32 MFMA kernel's
        try:                                  # (the Winograd kernels add fewer, larger-magnitude products and land closer to fp64)
            outs[arith] = plan.run(xin, residual=rin)
            torch.cuda.synchronize()
        finally:
            ops.CONV2D_ARITH, ops.CONV2D_ALGO = "f32", old_algo
    a, b = outs["f32"], outs["bf16x3"]
    mag = max(1.0, a.abs().max().item())
    assert (a - b).abs().max().item() < 3e-6 * mag
    if N * H * W <= 4096:
        with torch.no_grad():
            ref = mod.double().cpu()(x.double())
            if relu:
                ref = torch.relu(ref)
            if res:
                ref = torch.relu(ref + r.double())
        ref = ref.permute(0, 2, 3, 1)
        e32 = (a.double().cpu() - ref).abs().max().item()
        esp = (b.double().cpu() - ref).abs().max().item()
        assert esp <= 2.0 * e32 + 1e-7 * mag, (esp, e32)


@pytest.mark.ab
def test_e2e_golden_with_all_split_arithmetic(golden_dir):
    """Joint carry golden with conv3d AND the PSM conv2d kernels on the split arithmetic: depth within 1e-4."""
    import os
    import fixtures_spec as S
    from estdepth_amd import synth, DepthNetHybrid, ops
    g = np.load(os.path.join(golden_dir, "g9_joint_carry.npz"))
    m = DepthNetHybrid(ndepths=64, depth_min=0.1, depth_max=10.0, resnet=18, IF_EST_transformer=True).eval()
    synth.fill_state_dict(m, seed=2, head_gain=1.0)
    m = m.to(DEV).use_hip_psm().fuse_bn_2d()
    imgs, poses, intr, sample = S.e2e_inputs(8, S.E2E_HI, S.E2E_WI, seed=1004)
    imgs, poses, intr = imgs.to(DEV), poses.to(DEV), intr.to(DEV)
    ops.CONV2D_ARITH = ops.CONV3D_ARITH = "bf16x3"
    try:
        pc = pp = None
        for call in range(2):
            sl = slice(3 * call, 3 * call + 5)
            with torch.no_grad():
                outputs, pc, pp = m(imgs[:, sl], poses[:, sl], intr, {k: v[:, sl] for k, v in sample.items()}, pc, pp, mode="val")
            for k, v in outputs.items():
                name = "c%d|" % call + "|".join(map(str, k))
                if name in g.files:
                    assert np.abs(v.cpu().numpy() - g[name]).max() < 1e-4, name
    finally:
        ops.CONV2D_ARITH = ops.CONV3D_ARITH = "f32"


def test_spp_upsample_cat_matches_torch():
    """fused bilinear-upsample x4 + concat of the PSM SPP tail vs F.interpolate(align_corners=False) + torch.cat."""
    from estdepth_amd import ops
    g = torch.Generator().manual_seed(0)
    n, h, w = 2, 120, 160
    raw = torch.randn(n, h, w, 64, generator=g).to(DEV)
    skip = torch.randn(n, h, w, 128, generator=g).to(DEV)
    brs = [torch.randn(n, bh, bw, 32, generator=g).to(DEV) for bh, bw in ((30, 40), (15, 20), (7, 10), (3, 5))]
    out = ops.spp_upsample_cat(raw, skip, brs)
    ups = [torch.nn.functional.interpolate(b.permute(0, 3, 1, 2), size=(h, w), mode="bilinear", align_corners=False).permute(0, 2, 3, 1) for b in brs]
    ref = torch.cat([raw, skip] + ups, 3)
    assert out.shape == ref.shape
    assert torch.equal(out[..., :192], ref[..., :192])
    assert (out - ref).abs().max().item() < 2e-6


@pytest.mark.parametrize("cin,cout,k,s,hw,relu,with_bn", [(32, 64, 3, 2, (120, 160), True, True), (32, 64, 1, 2, (120, 160), False, True),
                                                           (64, 128, 1, 1, (60, 80), False, True), (128, 32, 1, 1, (30, 40), True, True),
                                                           (128, 32, 1, 1, (3, 5), True, True), (128, 32, 1, 1, (60, 80), False, False),
                                                           (32, 64, 3, 2, (37, 51), True, True), (32, 32, 1, 1, (9, 17), False, True)])
def test_small_psm_convolutions_vs_torch_fp64(cin, cout, k, s, hw, relu, with_bn):
    """csrc/refine2d.hip::conv2d_small_kernel -- the PSM extractor's 3x3 stride-2 and 1x1 (stride 1 | 2) convolutions + folded BN
    [+ ReLU] (psm_submodule.py:52,:72-74,:78-83,:100-110), NHWC, ragged sizes -- against torch's float64 convolution."""
    from estdepth_amd.backbones import small_conv_nhwc
    g = torch.Generator().manual_seed(cin + cout + k + s)
    conv = torch.nn.Conv2d(cin, cout, k, s, k // 2, bias=False)
    bn = torch.nn.BatchNorm2d(cout).eval() if with_bn else None
    with torch.no_grad():
        conv.weight.copy_(torch.randn(conv.weight.shape, generator=g) * 0.1)
        if bn is not None:
            bn.weight.copy_(torch.rand(cout, generator=g) + 0.5); bn.bias.copy_(torch.randn(cout, generator=g) * 0.1)
            bn.running_mean.copy_(torch.randn(cout, generator=g) * 0.1); bn.running_var.copy_(torch.rand(cout, generator=g) + 0.5)
    x = torch.randn(2, cin, hw[0], hw[1], generator=g)
    with torch.no_grad():
        ref = conv.double()(x.double())
        if bn is not None:
            ref = bn.double()(ref)
        if relu:
            ref = torch.relu(ref)
    conv, bn = conv.float().to(DEV), (bn.float().to(DEV) if bn is not None else None)
    out = small_conv_nhwc(conv, bn, x.to(DEV).permute(0, 2, 3, 1).contiguous(), relu)
    assert out is not None and tuple(out.shape) == (2, ref.shape[2], ref.shape[3], cout)
    err = float((out.permute(0, 3, 1, 2).double().cpu() - ref).abs().max())
    assert err < 2e-6 * max(1.0, float(ref.abs().max())) * (cin * k * k) ** 0.5, err


def test_psm_extractor_runs_no_library_convolution():
    """the matching branch on a ROCm device launches no MIOpen / hipBLASLt kernel: every convolution of PSMFeatures is one of
    the in-house kernels (only ATen's pooling remains)."""
    from torch.profiler import profile, ProfilerActivity
    from estdepth_amd import synth
    from estdepth_amd.backbones import PSMFeatures, enable_fused_bn
    m = PSMFeatures().eval()
    synth.fill_state_dict(m, seed=5)
    m = m.to(DEV).use_hip_convs()
    enable_fused_bn(m, True)
    x = synth.smooth_images(2, 128, 160, seed=3)[0].to(DEV) / 255.0 * 2 - 1
    x = x.contiguous(memory_format=torch.channels_last)
    with torch.no_grad():
        m(x)
        torch.cuda.synchronize()
        with profile(activities=[ProfilerActivity.CUDA]) as prof:
            y = m(x)
            torch.cuda.synchronize()
    names = [e.key for e in prof.key_averages()]
    lib = [n for n in names if any(t in n for t in ("Cijk_", "igemm", "miopen", "MIOpen", "gemm", "xdl", "naive_conv", "Conv"))]
    assert not lib, lib
    assert tuple(y.shape) == (2, 32, 32, 40)
