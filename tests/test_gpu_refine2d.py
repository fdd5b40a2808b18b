"""csrc/refine2d.hip (glue kernels of the decoder's 2D refinement tail, hybrid_depth_decoder.py:267-290) against plain torch on the
same tensors, through both bindings; and DepthHybridDecoder._refine_hip against _refine."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu
DEV = torch.device("cuda:0")


@pytest.fixture(scope="module", autouse=True)
def _need_gpu():
    if not torch.cuda.is_available():
        pytest.fail("GPU tests need an MI355X (no CPU path exists)")
    torch.backends.cudnn.allow_tf32 = False
    torch.backends.cuda.matmul.allow_tf32 = False


def _both_bindings(fn):
    from estdepth_amd import ops
    outs = []
    old = ops.BINDING
    try:
        for b in ("torch", "ctypes"):
            ops.BINDING = b
            outs.append(fn())
    finally:
        ops.BINDING = old
    assert torch.equal(outs[0], outs[1])
    return outs[0]


@pytest.mark.parametrize("shape", [(3, 64, 64, 120, 160), (1, 5, 3, 7, 9), (2, 16, 48, 1, 130), (1, 128, 128, 60, 81)])
def test_planes_cat_nhwc(shape):
    from estdepth_amd import ops
    n, ca, cb, h, w = shape
    g = torch.Generator().manual_seed(sum(shape))
    a, b = torch.randn(n, ca, h, w, generator=g).to(DEV), torch.randn(n, cb, h, w, generator=g).to(DEV)
    for relu in (False, True):
        got = _both_bindings(lambda: ops.planes_cat_nhwc(a, b, relu_b=relu))
        ref = torch.cat([a, torch.relu(b) if relu else b], 1).permute(0, 2, 3, 1)
        assert tuple(got.shape) == (n, h, w, ca + cb) and torch.equal(got, ref)


@pytest.mark.parametrize("shape", [(3, 32, 64, 240, 320), (1, 4, 8, 2, 6), (2, 16, 4, 10, 14)])
def test_upsample2_cat_nhwc(shape):
    from estdepth_amd import ops
    n, cx, cs, h, w = shape
    g = torch.Generator().manual_seed(sum(shape))
    x, skip = torch.randn(n, h // 2, w // 2, cx, generator=g).to(DEV), torch.randn(n, h, w, cs, generator=g).to(DEV)
    got = _both_bindings(lambda: ops.upsample2_cat_nhwc(x, skip))
    up = F.interpolate(x.permute(0, 3, 1, 2), scale_factor=2, mode="nearest")
    ref = torch.cat([up, skip.permute(0, 3, 1, 2)], 1).permute(0, 2, 3, 1)
    assert torch.equal(got, ref)
    with pytest.raises(RuntimeError):
        ops.upsample2_cat_nhwc(x[:, :, :, :3].contiguous(), skip)          # channel count not a multiple of 4


@pytest.mark.parametrize("shape,up", [((3, 16, 480, 640), 1), ((3, 32, 240, 320), 2), ((1, 16, 5, 7), 1), ((2, 32, 1, 9), 2)])
def test_disp_head_nhwc_vs_torch_fp64(shape, up):
    from estdepth_amd import ops
    n, c, h, w = shape
    g = torch.Generator().manual_seed(sum(shape) + up)
    x = torch.randn(n, h, w, c, generator=g).to(DEV)
    conv = torch.nn.Conv2d(c, 1, 3, 1, 1, 1, bias=True)
    with torch.no_grad():
        conv.weight.copy_(torch.randn(conv.weight.shape, generator=g) * 0.2)
        conv.bias.copy_(torch.randn(1, generator=g))
    conv = conv.to(DEV)
    got = _both_bindings(lambda: ops.disp_head_nhwc(x, conv.weight.detach(), conv.bias.detach(), 10.0, up))
    ref = 10.0 * torch.sigmoid(F.conv2d(x.permute(0, 3, 1, 2).double(), conv.weight.detach().double(), conv.bias.detach().double(), padding=1))
    if up == 2:
        ref = F.interpolate(ref, scale_factor=2)
    assert tuple(got.shape) == (n, 1, up * h, up * w)
    assert float((got.double() - ref).abs().max()) < 2e-5                  # depths of up to 10 m
    with pytest.raises(RuntimeError):
        ops.disp_head_nhwc(x[..., :8].contiguous(), conv.weight.detach()[:, :8].contiguous(), conv.bias.detach(), 10.0, 1)   # C = 8


def test_refine_hip_matches_refine():
    """the decoder's tail with the glue kernels on vs the plain torch tail (same convolutions), cfg2 sizes, ResNet-50 skips."""
    from estdepth_amd import synth
    from estdepth_amd.hybrid_depth_decoder import DepthHybridDecoder
    dec = DepthHybridDecoder(np.array([64, 256, 512, 1024, 2048]), ndepths=64, depth_max=10.0, IF_EST_transformer=False).eval()
    synth.fill_state_dict(dec, seed=12)
    dec = dec.to(DEV)
    for name, child in dec.named_children():                         # as DepthNetHybrid.use_channels_last_2d leaves the 2D layers
        if name.startswith(("upconv", "dispconv")):
            child.to(memory_format=torch.channels_last)
    g = torch.Generator().manual_seed(3)
    T, H, W = 3, 120, 160
    sem = torch.relu(torch.randn(T, 64, H, W, generator=g)).to(DEV)
    logits = torch.randn(T, 64, H, W, generator=g).to(DEV)
    f0 = torch.relu(torch.randn(T, 64, 2 * H, 2 * W, generator=g)).to(DEV).contiguous(memory_format=torch.channels_last)
    feats = [f0, None, None, None, None]
    with torch.no_grad():
        a1, a0 = dec._refine(sem, logits, feats)
        dec._hip_refine = True
        b1, b0 = dec._refine(sem, logits, feats)
    assert tuple(b1.shape) == tuple(a1.shape) == (T, 1, 4 * H, 4 * W) and tuple(b0.shape) == tuple(a0.shape)
    assert float((a1 - b1).abs().max()) < 2e-5 and float((a0 - b0).abs().max()) < 2e-5


@pytest.mark.parametrize("shape", [(5, 480, 640), (1, 7, 9), (2, 33, 64), (1, 1, 1)])
def test_stem3x3s2_nhwc_vs_torch_fp64(shape):
    """first PSM layer: Conv2d(3,32,3,2,1) + BatchNorm2d(eval) + ReLU (networks/psm_submodule.py:47) incl. odd sizes."""
    from estdepth_amd import ops, packing
    n, h, w = shape
    g = torch.Generator().manual_seed(sum(shape))
    x = (torch.rand(n, h, w, 3, generator=g) * 2 - 1).to(DEV)
    conv = torch.nn.Conv2d(3, 32, 3, 2, 1, bias=False)
    bn = torch.nn.BatchNorm2d(32).eval()
    with torch.no_grad():
        conv.weight.copy_(torch.randn(conv.weight.shape, generator=g) * 0.3)
        bn.weight.copy_(torch.rand(32, generator=g) + 0.5); bn.bias.copy_(torch.randn(32, generator=g) * 0.2)
        bn.running_mean.copy_(torch.randn(32, generator=g) * 0.2); bn.running_var.copy_(torch.rand(32, generator=g) + 0.5)
    sc, sh = packing.fold_bn_fp32(bn, list(range(32)))
    wd = conv.weight.detach().to(DEV)
    got = _both_bindings(lambda: ops.stem3x3s2_nhwc(x, wd, sc.to(DEV), sh.to(DEV)))
    with torch.no_grad():
        ref = torch.relu(bn.double()(conv.double()(x.cpu().double().permute(0, 3, 1, 2)))).permute(0, 2, 3, 1)
    assert tuple(got.shape) == tuple(ref.shape) == (n, (h - 1) // 2 + 1, (w - 1) // 2 + 1, 32)
    assert float((got.cpu().double() - ref).abs().max()) < 1e-5 * max(1.0, float(ref.abs().max()))


def test_normalise_nhwc_is_bit_identical_to_the_reference_cpu_ops():
    """2 * (imgs / 255.) - 1. (model_hybrid.py:119) as torch evaluates it on the CPU (the reference path): a true division, then two
    more roundings.  (torch on the GPU multiplies by fl(1/255) instead -- up to 1 ulp away from the reference.)"""
    from estdepth_amd import ops
    g = torch.Generator().manual_seed(9)
    imgs = (torch.rand(5, 3, 480, 640, generator=g) * 255).to(DEV)
    imgs[0, 0, 0, :4] = torch.tensor([0.0, 255.0, 127.5, 1e-3], device=DEV)
    got = _both_bindings(lambda: ops.normalise_nhwc(imgs))
    ref = (2 * (imgs.cpu() / 255.) - 1.).permute(0, 2, 3, 1)
    assert tuple(got.shape) == (5, 480, 640, 3) and torch.equal(got.cpu(), ref)
    odd = (torch.rand(2, 3, 7, 13, generator=g) * 255).to(DEV)
    assert torch.equal(ops.normalise_nhwc(odd).cpu(), (2 * (odd.cpu() / 255.) - 1.).permute(0, 2, 3, 1))


@pytest.mark.parametrize("cin,up,shape", [(32, False, (3, 240, 320)), (16, True, (3, 240, 320)), (16, False, (1, 5, 7)), (32, True, (2, 3, 9)),
                                          (16, True, (1, 1, 1)), (32, False, (1, 17, 33))])
def test_conv2d_k3_to16_vs_torch_fp64(cin, up, shape):
    """ConvBlock (conv3x3 + BN + ReLU) to 16 channels, optionally on the nearest-x2 upsampled input, incl. ragged sizes."""
    from estdepth_amd import ops, packing
    n, h, w = shape
    g = torch.Generator().manual_seed(sum(shape) + cin + int(up))
    x = torch.randn(n, h, w, cin, generator=g).to(DEV)
    conv = torch.nn.Conv2d(cin, 16, 3, 1, 1, bias=False)
    bn = torch.nn.BatchNorm2d(16).eval()
    with torch.no_grad():
        conv.weight.copy_(torch.randn(conv.weight.shape, generator=g) * 0.1)
        bn.weight.copy_(torch.rand(16, generator=g) + 0.5); bn.bias.copy_(torch.randn(16, generator=g) * 0.2)
        bn.running_mean.copy_(torch.randn(16, generator=g) * 0.2); bn.running_var.copy_(torch.rand(16, generator=g) + 0.5)
    sc, sh = packing.fold_bn_fp32(bn, list(range(16)))
    wp = packing.pack_conv2d_to16(conv.weight).to(DEV)
    got = _both_bindings(lambda: ops.conv2d_k3_to16_nhwc(x, wp, sc.to(DEV), sh.to(DEV), up))
    xin = x.cpu().double().permute(0, 3, 1, 2)
    if up:
        xin = F.interpolate(xin, scale_factor=2, mode="nearest")
    with torch.no_grad():
        ref = torch.relu(bn.double()(conv.double()(xin))).permute(0, 2, 3, 1)
    assert tuple(got.shape) == tuple(ref.shape)
    assert float((got.cpu().double() - ref).abs().max()) < 1e-5 * max(1.0, float(ref.abs().max()))


@pytest.mark.parametrize("shape", [(3, 120, 160, 64), (2, 7, 9, 5), (1, 1, 1, 1), (1, 4, 6, 300)])
def test_nhwc_to_planes_is_a_permuted_copy(shape):
    """estd_nhwc_to_planes: [N,H,W,C] records -> [N,C,H,W] planes, bit-identical to permute + contiguous, through both bindings."""
    from estdepth_amd import ops
    x = torch.randn(*shape, generator=torch.Generator().manual_seed(shape[1])).to("cuda")
    ref = x.permute(0, 3, 1, 2).contiguous()
    old = ops.BINDING
    try:
        for b in ("torch", "ctypes"):
            ops.BINDING = b
            out = ops.nhwc_to_planes(x)
            assert out.is_contiguous() and torch.equal(out, ref)
    finally:
        ops.BINDING = old
    with pytest.raises(RuntimeError):
        ops.nhwc_to_planes(torch.zeros(1, 2, 2, 600, device="cuda"))       # more channels than the LDS tile holds
