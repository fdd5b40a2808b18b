"""GPU: csrc/conv2d_taps.hip -- the layers of the semantic ResNet outside the 1x1 / tiled stride-1 kernels
(hybrid_models/resnet_encoder.py:40-51 over torchvision's ResNet: conv1 7x7 / stride 2 + bn1 + relu, maxpool, the stride-2 3x3
convolutions of layer2..4; the 3x3 of the 2D decoder on the 1/32 map, hybrid_depth_decoder.py:17-30) and the average pooling of the
PSM SPP branches (networks/psm_submodule.py:56-70) -- against fp64 evaluations of the same fp32 data through both bindings, and the
whole semantic branch free of library convolution / pooling kernels."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu
DEV = "cuda"


def _both_bindings(fn):
    from estdepth_amd import ops
    outs = []
    old = ops.BINDING
    try:
        for b in ("torch", "ctypes"):
            ops.BINDING = b
            outs.append(fn())
            torch.cuda.synchronize()
    finally:
        ops.BINDING = old
    assert torch.equal(torch.nan_to_num(outs[0], nan=3.0), torch.nan_to_num(outs[1], nan=3.0))      # the two bindings launch the same kernel on the same data
    return outs[0]


CASES = [  # N, H, W, cin, cout, k, stride, pad, relu, residual, affine
    (3, 120, 160, 128, 128, 3, 2, 1, True, False, True),     # layer2[0].conv2 at the benchmark's size
    (3, 60, 80, 256, 256, 3, 2, 1, True, False, True),       # layer3[0].conv2
    (3, 30, 40, 512, 512, 3, 2, 1, True, False, True),       # layer4[0].conv2: K split over the four waves
    (3, 15, 20, 2048, 256, 3, 1, 1, True, False, True),      # the 2D decoder's first 3x3 on the 1/32 map: K = 18432
    (2, 13, 17, 64, 128, 3, 2, 1, True, True, True),         # ResNet-18 layer2[0].conv1 shape, odd map, with a residual
    (1, 7, 9, 16, 32, 3, 1, 1, False, False, False),         # smallest channels, ragged pixel count, no affine, no activation
    (1, 9, 11, 48, 96, 5, 2, 2, True, True, True),           # 5x5, odd chunk count (cin = 48)
    (2, 8, 8, 32, 64, 3, 1, 0, False, False, True),          # no padding
    (1, 6, 10, 64, 64, 1, 2, 0, True, False, True),          # k = 1 degenerates to the 1x1 kernel's arithmetic
]


@pytest.mark.parametrize("case", CASES)
def test_conv2d_taps_vs_fp64(case):
    from estdepth_amd import ops, packing
    N, H, W, cin, cout, k, stride, pad, relu, has_res, affine = case
    g = torch.Generator().manual_seed(sum(case[:8]))
    x = torch.randn(N, H, W, cin, generator=g)
    w = torch.randn(cout, cin, k, k, generator=g) / np.sqrt(cin * k * k)
    sc = (torch.rand(cout, generator=g) + 0.5) if affine else None
    sh = torch.randn(cout, generator=g) if affine else None
    ref = F.conv2d(x.permute(0, 3, 1, 2).double(), w.double(), None, stride, pad).permute(0, 2, 3, 1)
    Ho, Wo = ref.shape[1], ref.shape[2]
    res = torch.randn(N, Ho, Wo, cout, generator=g) if has_res else None
    if affine:
        ref = ref * sc.double() + sh.double()
    if has_res:
        ref = ref + res.double()
    if relu:
        ref = ref.clamp_min(0)
    wt = packing.pack_conv2d_taps(w).to(DEV)
    xg = x.to(DEV)
    out = _both_bindings(lambda: ops.conv2d_taps_nhwc(xg, wt, sc.to(DEV) if affine else None, sh.to(DEV) if affine else None, k, stride, pad, relu,
                                                      res.to(DEV) if has_res else None))
    assert tuple(out.shape) == (N, Ho, Wo, cout)
    err = (out.double().cpu() - ref).abs().max().item()
    mag = ref.abs().max().item()
    assert err < 2e-7 * np.sqrt(cin * k * k) * max(mag, 1.0) + 1e-6, (err, mag)


@pytest.mark.parametrize("cfg", ["441", "421", "241", "221", "444", "424", "244", "224", "124", "121"])
def test_conv2d_taps_every_instance(cfg, monkeypatch):
    """every (block, K-split) instance behind the heuristic computes the same convolution (a child process: the override is read once)."""
    import os
    import subprocess
    import sys
    code = r'''
import numpy as np, torch, torch.nn.functional as F
from estdepth_amd import ops, packing
g = torch.Generator().manual_seed(11)
x = torch.randn(2, 11, 13, 64, generator=g); w = torch.randn(64, 64, 3, 3, generator=g) / 24
ref = F.conv2d(x.permute(0, 3, 1, 2).double(), w.double(), None, 2, 1).permute(0, 2, 3, 1).clamp_min(0)
out = ops.conv2d_taps_nhwc(x.cuda(), packing.pack_conv2d_taps(w).cuda(), None, None, 3, 2, 1, True, None)
err = float((out.double().cpu() - ref).abs().max())
assert err < 1e-5, err
print("OK")
'''
    env = dict(os.environ, ESTD_CTAPS_CFG=cfg)
    r = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, cwd=os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    assert r.returncode == 0 and "OK" in r.stdout, r.stderr[-2000:]


def test_conv2d_taps_rejects_what_it_has_no_instance_for():
    from estdepth_amd import ops
    with pytest.raises(RuntimeError):
        ops.conv2d_taps_nhwc(torch.zeros(1, 4, 4, 24, device=DEV), torch.zeros(9, 32, 24, device=DEV), None, None, 3)      # cin % 16
    with pytest.raises(RuntimeError):
        ops.conv2d_taps_nhwc(torch.zeros(1, 4, 4, 32, device=DEV), torch.zeros(9, 32, 32, device=DEV), None, None, 3, stride=3)
    with pytest.raises(RuntimeError):
        ops.conv2d_taps_nhwc(torch.zeros(1, 4, 4, 32, device=DEV), torch.zeros(49, 32, 32, device=DEV), None, None, 7)     # 7x7: no instance


@pytest.mark.parametrize("shape", [(3, 480, 640), (1, 37, 53), (2, 8, 6), (1, 1, 1)])
def test_stem7x7_vs_fp64(shape):
    from estdepth_amd import ops, packing
    N, H, W = shape
    g = torch.Generator().manual_seed(H + W)
    x = torch.randn(N, H, W, 3, generator=g)
    w = torch.randn(64, 3, 7, 7, generator=g) / np.sqrt(147.0)
    sc, sh = torch.rand(64, generator=g) + 0.5, torch.randn(64, generator=g)
    ref = F.conv2d(x.permute(0, 3, 1, 2).double(), w.double(), None, 2, 3).permute(0, 2, 3, 1)
    ref = (ref * sc.double() + sh.double()).clamp_min(0)
    wp = packing.pack_stem7x7(w).to(DEV)
    xg, scg, shg = x.to(DEV), sc.to(DEV), sh.to(DEV)
    out = _both_bindings(lambda: ops.stem7x7s2_nhwc(xg, wp, scg, shg))
    assert tuple(out.shape) == tuple(ref.shape)
    err = (out.double().cpu() - ref).abs().max().item()
    assert err < 2e-7 * np.sqrt(147.0) * max(ref.abs().max().item(), 1.0) + 1e-6, err


@pytest.mark.parametrize("shape", [(3, 240, 320, 64), (1, 7, 9, 4), (2, 6, 8, 12), (1, 1, 1, 8)])
def test_maxpool_is_atens(shape):
    from estdepth_amd import ops
    N, H, W, C = shape
    x = torch.randn(N, H, W, C, generator=torch.Generator().manual_seed(H))
    if H > 4:
        x[0, 3, 2, 1] = float("nan")
    ref = F.max_pool2d(x.permute(0, 3, 1, 2), 3, 2, 1).permute(0, 2, 3, 1)
    xg = x.to(DEV)
    out = _both_bindings(lambda: ops.maxpool3x3s2_nhwc(xg)).cpu()
    assert tuple(out.shape) == tuple(ref.shape)
    assert torch.equal(torch.isnan(out), torch.isnan(ref))
    assert torch.equal(torch.nan_to_num(out, nan=7.0), torch.nan_to_num(ref, nan=7.0))


@pytest.mark.parametrize("case", [(5, 120, 160, 128, 4), (1, 30, 40, 128, 8), (2, 9, 11, 8, 2), (1, 5, 5, 4, 5), (1, 4, 4, 4, 1)])
def test_avgpool_vs_torch(case):
    from estdepth_amd import ops
    N, H, W, C, k = case
    x = torch.randn(N, H, W, C, generator=torch.Generator().manual_seed(k))
    ref = F.avg_pool2d(x.permute(0, 3, 1, 2).double(), k, k).permute(0, 2, 3, 1)
    xg = x.to(DEV)
    out = _both_bindings(lambda: ops.avgpool_nhwc(xg, k)).cpu()
    assert tuple(out.shape) == tuple(ref.shape)
    assert float((out.double() - ref).abs().max()) < 1e-6 * max(1.0, float(ref.abs().max()))
    # same summation order and one division as ATen's kernel on the same device: bit-identical
    assert torch.equal(out, F.avg_pool2d(xg.permute(0, 3, 1, 2), k, k).permute(0, 2, 3, 1).cpu())


def test_pool_argument_errors():
    from estdepth_amd import ops
    with pytest.raises(RuntimeError):
        ops.maxpool3x3s2_nhwc(torch.zeros(1, 4, 4, 6, device=DEV))       # C % 4
    with pytest.raises(RuntimeError):
        ops.avgpool_nhwc(torch.zeros(1, 4, 4, 8, device=DEV), 5)        # window larger than the map


@pytest.mark.parametrize("depth", [50, 18])
def test_semantic_encoder_runs_no_library_kernel(depth):
    """the whole semantic branch (hybrid_models/resnet_encoder.py:40-51) in the default fused path at the benchmark's size: stem, max
    pooling, every 1x1 / 3x3 convolution with its BatchNorm / residual / ReLU in-house -- no MIOpen, hipBLASLt / rocBLAS or ATen
    pooling / BatchNorm kernel -- and equal to the plain module on the CPU."""
    from estdepth_amd import synth
    from estdepth_amd.backbones import SemanticEncoder, enable_fused_bn, enable_hip_3x3
    enc = SemanticEncoder(depth, "pretrained").eval()
    synth.fill_state_dict(enc, seed=4)
    x = torch.randn(3, 3, 480, 640, generator=torch.Generator().manual_seed(2))
    with torch.no_grad():
        ref = enc(x)
        g = enc.to(DEV).to(memory_format=torch.channels_last)
        enable_fused_bn(g, True)
        enable_hip_3x3(g, True)
        xg = x.to(DEV).contiguous(memory_format=torch.channels_last)
        g(xg)
        with torch.profiler.profile(activities=[torch.profiler.ProfilerActivity.CUDA]) as prof:
            got = g(xg)
            torch.cuda.synchronize()
    ev = {e.key: e.count for e in prof.key_averages()}
    lib = [k for k in ev if any(t in k.lower() for t in ("igemm", "miopen", "cijk", "gemm", "max_pool", "avg_pool", "batch_norm", "subtensorop",
                                                         "naive_conv", "winograd", "at::native"))]
    assert not lib, lib
    assert sum(c for k, c in ev.items() if "stem7x7s2" in k) == 1 and sum(c for k, c in ev.items() if "maxpool3x3s2" in k) == 1
    assert sum(c for k, c in ev.items() if "conv2d_taps_kernel" in k) >= 3            # the stride-2 3x3 convolutions
    for a, b in zip(got, ref):
        scale = float(b.abs().max())
        assert float((a.cpu() - b).abs().max()) < 3e-5 * max(scale, 1.0)
