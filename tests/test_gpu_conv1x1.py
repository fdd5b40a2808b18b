"""GPU: estd_conv1x1_nhwc (csrc/conv1x1.hip) -- 1x1 convolution (stride 1|2) + folded BatchNorm [+ residual] [+ ReLU] of an NHWC
map, the ResNet bottleneck convolutions of the semantic branch (hybrid_models/resnet_encoder.py:40-51) -- against an fp64
evaluation of the same fp32 data, through both bindings, on both wave-tile sizes, ragged pixel counts and every epilogue."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = "cuda"


def _ref(x, w, sc, sh, stride, relu, res):
    xs = x[:, ::stride, ::stride].double()
    y = torch.einsum("nhwc,oc->nhwo", xs, w.double())
    if sc is not None:
        y = y * sc.double()
    if sh is not None:
        y = y + sh.double()
    if res is not None:
        y = y + res.double()
    return y.clamp_min(0) if relu else y


CASES = [  # N, H, W, cin, cout, stride, relu, residual, affine
    (2, 24, 40, 64, 256, 1, True, True, True),        # layer1 conv3 + shortcut (64 x 64 blocks need >= 2048 wave tiles: this one takes 32 x 32)
    (3, 120, 160, 64, 256, 1, True, True, True),      # the full-size one: 64 x 64 blocks
    (3, 120, 160, 256, 64, 1, True, False, True),     # layer1 conv1
    (2, 24, 40, 256, 512, 2, False, False, True),     # downsample, stride 2
    (1, 7, 9, 16, 32, 1, False, False, False),        # smallest channels, ragged pixel count, no affine
    (1, 15, 20, 2048, 512, 1, True, False, True),     # layer4 conv1: long K, few pixels
    (2, 5, 7, 48, 96, 2, True, True, True),           # odd chunk count (cin = 48), odd map with stride 2, residual
    # the LDS-tiled form (round 6; default wherever there are >= ~224 workgroup tiles of 64 x 64):
    (1, 61, 83, 64, 256, 1, True, True, True),        # 16-channel stages, ragged last pixel tile (5063 pixels), residual
    (3, 60, 80, 512, 128, 1, True, False, True),      # 32-channel stages (layer2 conv1)
    (3, 30, 40, 1024, 256, 1, False, False, True),    # 32-pixel tiles, 64-channel stages (layer3 conv1), no ReLU
    (3, 59, 81, 128, 512, 2, True, True, True),       # stride 2 on an odd map through the tiled form, residual
]


@pytest.mark.parametrize("case", CASES)
@pytest.mark.parametrize("binding", ["torch", "ctypes"])
def test_conv1x1_vs_fp64(case, binding):
    from estdepth_amd import ops
    N, H, W, cin, cout, stride, relu, has_res, affine = case
    g = torch.Generator().manual_seed(sum(case[:6]))
    x = torch.randn(N, H, W, cin, generator=g)
    w = torch.randn(cout, cin, generator=g) / np.sqrt(cin)
    sc = (torch.rand(cout, generator=g) + 0.5) if affine else None
    sh = torch.randn(cout, generator=g) if affine else None
    Ho, Wo = (H - 1) // stride + 1, (W - 1) // stride + 1
    res = torch.randn(N, Ho, Wo, cout, generator=g) if has_res else None
    ref = _ref(x, w, sc, sh, stride, relu, res)
    old = ops.BINDING
    ops.BINDING = binding
    try:
        out = ops.conv1x1_nhwc(x.to(DEV), w.to(DEV), sc.to(DEV) if affine else None, sh.to(DEV) if affine else None, stride, relu,
                               res.to(DEV) if has_res else None)
        torch.cuda.synchronize()
    finally:
        ops.BINDING = old
    assert tuple(out.shape) == (N, Ho, Wo, cout)
    err = (out.double().cpu() - ref).abs().max().item()
    mag = ref.abs().max().item()
    # fp32 accumulation over cin products: the error of a length-cin fp32 dot product
    assert err < 2e-7 * np.sqrt(cin) * max(mag, 1.0) + 1e-6, (err, mag)


@pytest.mark.parametrize("cfg", ["1441", "1422", "1242", "1224", "1122"])
def test_conv1x1_forced_tile_configurations(cfg):
    """every other workgroup tile of the LDS-tiled form (ESTD_C1X1_CFG is latched at the first call: one child process per configuration),
    ragged pixel count, 96 output channels short of a whole 128-channel tile is NOT allowed (cout % 64 == 0 is the form's condition) --
    320 channels leave a half-empty last 128-channel tile instead."""
    import os
    import subprocess
    import sys
    code = (
        "import torch, numpy as np\n"
        "from estdepth_amd import ops\n"
        "g = torch.Generator().manual_seed(5)\n"
        "x = torch.randn(2, 37, 53, 128, generator=g); w = torch.randn(320, 128, generator=g) / 128 ** 0.5\n"
        "sc = torch.rand(320, generator=g) + 0.5; sh = torch.randn(320, generator=g); r = torch.randn(2, 37, 53, 320, generator=g)\n"
        "ref = (torch.einsum('nhwc,oc->nhwo', x.double(), w.double()) * sc.double() + sh.double() + r.double()).clamp_min(0)\n"
        "out = ops.conv1x1_nhwc(x.cuda(), w.cuda(), sc.cuda(), sh.cuda(), 1, True, r.cuda())\n"
        "err = float((out.double().cpu() - ref).abs().max()); mag = float(ref.abs().max())\n"
        "assert err < 2e-7 * np.sqrt(128) * mag + 1e-6, (err, mag)\n"
        "print('OK', err)\n")
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, "-c", code], cwd=root, env=dict(os.environ, ESTD_C1X1_CFG=cfg), capture_output=True, text=True, timeout=600)
    assert r.returncode == 0 and "OK" in r.stdout, (cfg, r.stdout[-300:], r.stderr[-600:])


def test_conv1x1_rejects_what_it_has_no_instance_for():
    from estdepth_amd import ops
    x = torch.zeros(1, 4, 4, 24, device=DEV)
    with pytest.raises(RuntimeError):
        ops.conv1x1_nhwc(x, torch.zeros(32, 24, device=DEV), None, None)             # cin not a multiple of 16
    with pytest.raises(RuntimeError):
        ops.conv1x1_nhwc(torch.zeros(1, 4, 4, 32, device=DEV), torch.zeros(32, 32, device=DEV), None, None, stride=3)


@pytest.mark.parametrize("policy", ["all", "auto"])
def test_bottleneck_fused_path(policy):
    """ResNet-50 layer1 (three stride-1 bottlenecks at 120x160) in the fused-BN path: with ESTD_HIP_1X1=all every convolution, the
    residual add and the ReLUs run in csrc/conv1x1.hip / the MFMA conv2d kernel -- no hipBLASLt / rocBLAS / MIOpen kernel and no
    separate BatchNorm pass; with the default policy the convolutions with a residual do.  Both equal the plain module."""
    from estdepth_amd import backbones, synth
    from estdepth_amd.backbones import ResNetTrunk, enable_fused_bn, enable_hip_3x3
    trunk = ResNetTrunk(50).eval()
    synth.fill_state_dict(trunk, seed=6)
    stage = trunk.layer1
    x = torch.randn(2, 64, 120, 160, generator=torch.Generator().manual_seed(3))
    old = backbones.HIP_1X1
    backbones.HIP_1X1 = policy
    try:
        with torch.no_grad():
            ref = stage(x)
            gs = stage.to(DEV).to(memory_format=torch.channels_last)
            enable_fused_bn(gs, True)
            enable_hip_3x3(gs, True)
            xg = x.to(DEV).contiguous(memory_format=torch.channels_last)
            gs(xg)                                                     # plans
            with torch.profiler.profile(activities=[torch.profiler.ProfilerActivity.CUDA]) as prof:
                got = gs(xg)
                torch.cuda.synchronize()
    finally:
        backbones.HIP_1X1 = old
    names = [e.key for e in prof.key_averages()]
    assert sum(("conv1x1_nhwc_kernel" in n) or ("conv1x1_lds_kernel" in n) for n in names) >= 1, names
    if policy == "all":
        assert not any(("Cijk" in n) or ("gemm" in n.lower()) or ("bn_act" in n) or ("miopen" in n.lower()) for n in names), names
    scale = float(ref.abs().max())
    assert float((got.cpu() - ref).abs().max()) < 2e-5 * max(scale, 1.0)


def test_semantic_encoder_launches_no_library_gemm():
    """the whole ResNet-50 semantic branch in the default fused path at the benchmark's size (3 images of 480x640): every 1x1
    convolution (csrc/conv1x1.hip) and every stride-1 3x3 convolution (csrc/conv2d_wino2.hip) is in-house -- no hipBLASLt / rocBLAS
    GEMM; the 7x7 stem and the three stride-2 3x3 convolutions are csrc/conv2d_taps.hip's (tests/test_gpu_conv2d_taps.py asserts that NO library
    kernel is left)."""
    from estdepth_amd import synth
    from estdepth_amd.backbones import SemanticEncoder, enable_fused_bn, enable_hip_3x3
    enc = SemanticEncoder(50, "pretrained").eval()
    synth.fill_state_dict(enc, seed=4)
    g = enc.to(DEV).to(memory_format=torch.channels_last)
    enable_fused_bn(g, True)
    enable_hip_3x3(g, True)
    x = torch.randn(3, 3, 480, 640, device=DEV).contiguous(memory_format=torch.channels_last)
    with torch.no_grad():
        g(x)
        with torch.profiler.profile(activities=[torch.profiler.ProfilerActivity.CUDA]) as prof:
            g(x)
            torch.cuda.synchronize()
    ev = {e.key: e.count for e in prof.key_averages()}
    assert not any("Cijk" in k for k in ev), [k for k in ev if "Cijk" in k]
    # 16 bottlenecks x (conv1, conv3) + 4 downsample convolutions, on the direct or the LDS-tiled form of csrc/conv1x1.hip
    assert sum(c for k, c in ev.items() if "conv1x1_nhwc_kernel" in k or "conv1x1_lds_kernel" in k) == 36
    assert sum(c for k, c in ev.items() if "conv1x1_lds_kernel" in k) >= 12             # (the tiled form is the default of most of them)
    assert sum(c for k, c in ev.items() if "conv2d_wino2_kernel" in k) == 13           # the stride-1 3x3 convolutions
    assert sum(c for k, c in ev.items() if "igemm" in k or "gemm" in k.lower() and "Cijk" not in k) <= 6, ev
