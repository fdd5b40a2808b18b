"""The PSM extractor's stride-2 / 1x1 convolutions at cfg2 size (5 frames): in-house conv2d_small_kernel vs the library path
(MIOpen convolution / hipBLASLt GEMM + the fused BN pass).  python tools/psm_small_bench.py"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from estdepth_amd.microbench import warm
import torch
from estdepth_amd import backbones as B
dev = "cuda"


def t(f, n=30):
    warm(f, 0.15)                                   # sustained clocks (estdepth_amd/microbench.py)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): f()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


for name, cin, cout, k, s, h, w, relu, with_bn in [("layer2[0].conv1 3x3 s2", 32, 64, 3, 2, 240, 320, True, True),
                                                   ("layer2[0].downsample 1x1 s2", 32, 64, 1, 2, 240, 320, False, True),
                                                   ("layer3[0].downsample 1x1", 64, 128, 1, 1, 120, 160, False, True),
                                                   ("SPP branch 1x1 (pool 4)", 128, 32, 1, 1, 30, 40, True, True),
                                                   ("lastconv 1x1", 128, 32, 1, 1, 120, 160, False, False)]:
    conv = torch.nn.Conv2d(cin, cout, k, s, k // 2, bias=False).to(dev).to(memory_format=torch.channels_last)
    bn = torch.nn.BatchNorm2d(cout).to(dev).eval() if with_bn else None
    x = torch.randn(5, cin, h, w, device=dev).contiguous(memory_format=torch.channels_last)
    xn = x.permute(0, 2, 3, 1)
    ours = t(lambda: B.small_conv_nhwc(conv, bn, xn, relu))
    with torch.no_grad():
        if bn is not None:
            lib = t(lambda: B.conv_bn_act(conv, bn, x, relu))
        else:
            lib = t(lambda: B.conv1x1_gemm(conv, x))
    gf = 2.0 * 5 * (h // s) * (w // s) * cin * cout * k * k / 1e9
    mb = 4.0 * 5 * (h * w * cin / (s * s if k == 1 else 1) + (h // s) * (w // s) * cout) / 1e6
    print("%-28s ours %7.1f us (%.1f TF/s, %.2f TB/s)   library %7.1f us" % (name, ours, gf / ours * 1e-3 * 1e3 / 1e3, mb / ours, lib))
