"""csrc/conv2d_taps.hip stand-alone at the benchmark's shapes (3 images of 480x640) beside the library path it replaces (MIOpen convolution
+ estd_bn_act_nhwc; ATen pooling): stride-2 3x3 convolutions of ResNet-50's layer2..4, the decoder's 2048 -> 256 3x3 on the 15x20 map, the
7x7 stem, max pooling, the SPP average pooling.    python tools/taps_bench.py"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch, torch.nn.functional as F
from estdepth_amd import ops, packing
from estdepth_amd.microbench import warm
dev = "cuda"


def t(f, n=30):
    warm(f, 0.2)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): f()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n


N = 3
print("k x k convolutions + BN + ReLU: shape, library ms (conv + bn_act), taps kernel ms (TFLOP/s)")
for (h, w, cin, cout, k, s) in [(120, 160, 128, 128, 3, 2), (60, 80, 256, 256, 3, 2), (30, 40, 512, 512, 3, 2), (15, 20, 2048, 256, 3, 1),
                                (120, 160, 64, 128, 3, 2), (60, 80, 128, 256, 3, 2), (30, 40, 256, 512, 3, 2)]:
    conv = torch.nn.Conv2d(cin, cout, k, s, k // 2, bias=False).to(dev).to(memory_format=torch.channels_last)
    x = torch.randn(N, cin, h, w, device=dev).contiguous(memory_format=torch.channels_last)
    sc, sh = torch.rand(cout, device=dev) + 0.5, torch.randn(cout, device=dev)
    wt = packing.pack_conv2d_taps(conv.weight).to(dev)
    xn = x.permute(0, 2, 3, 1)
    ho, wo = (h - 1) // s + 1, (w - 1) // s + 1
    gf = 2.0 * k * k * N * ho * wo * cin * cout / 1e9
    with torch.no_grad():
        a = t(lambda: ops.bn_act_nhwc_(conv(x), sc, sh, True, None))
        b = t(lambda: ops.conv2d_taps_nhwc(xn, wt, sc, sh, k, s, k // 2, True, None))
    print("%4dx%-4d %5d->%-5d k%d s%d  library %.4f   taps %.4f (%.1f)" % (h, w, cin, cout, k, s, a, b, gf / b))
print("stem 7x7 s2 3->64 @480x640 x3")
conv = torch.nn.Conv2d(3, 64, 7, 2, 3, bias=False).to(dev).to(memory_format=torch.channels_last)
x = torch.randn(N, 3, 480, 640, device=dev).contiguous(memory_format=torch.channels_last)
sc, sh = torch.rand(64, device=dev) + 0.5, torch.randn(64, device=dev)
wp = packing.pack_stem7x7(conv.weight).to(dev)
xn = x.permute(0, 2, 3, 1)
with torch.no_grad():
    a = t(lambda: ops.bn_act_nhwc_(conv(x), sc, sh, True, None))
    b = t(lambda: ops.stem7x7s2_nhwc(xn, wp, sc, sh))
print("  library %.4f   stem7x7s2 %.4f ms (%.2f TB/s of the 59 MB output)" % (a, b, N * 240 * 320 * 64 * 4 / b / 1e9))
f0 = torch.randn(N, 64, 240, 320, device=dev).contiguous(memory_format=torch.channels_last)
fn = f0.permute(0, 2, 3, 1)
a = t(lambda: F.max_pool2d(f0, 3, 2, 1)); b = t(lambda: ops.maxpool3x3s2_nhwc(fn))
print("maxpool 3x3 s2 64ch @240x320 x3: ATen %.4f   in-house %.4f ms" % (a, b))
sk = torch.randn(5, 128, 120, 160, device=dev).contiguous(memory_format=torch.channels_last)
sn = sk.permute(0, 2, 3, 1)
a = t(lambda: F.avg_pool2d(sk, 4, 4)); b = t(lambda: ops.avgpool_nhwc(sn, 4))
print("avgpool 4x4 128ch @120x160 x5: ATen %.4f   in-house %.4f ms" % (a, b))
print("PSM small convolutions (5 images): conv2d_small_kernel vs conv2d_taps / conv1x1")
from estdepth_amd import packing as _pk
for (h, w, cin, cout, k, s) in [(240, 320, 32, 64, 3, 2), (240, 320, 32, 64, 1, 2), (120, 160, 128, 32, 1, 1), (120, 160, 64, 128, 1, 1), (30, 40, 128, 32, 1, 1)]:
    conv = torch.nn.Conv2d(cin, cout, k, s, k // 2, bias=False).to(dev)
    x = torch.randn(5, h, w, cin, device=dev)
    sc, sh = torch.rand(cout, device=dev) + 0.5, torch.randn(cout, device=dev)
    a = None
    if (cin, k, s) in ops.SMALL_CONV_SHAPES:
        wp = _pk.pack_conv2d_small(conv.weight).to(dev)
        a = t(lambda: ops.conv2d_small_nhwc(x, wp, sc, sh, cout, k, s, True))
    if k == 1:
        w2 = conv.weight.detach().reshape(cout, cin).contiguous()
        b = t(lambda: ops.conv1x1_nhwc(x, w2, sc, sh, s, True, None))
    else:
        wt = _pk.pack_conv2d_taps(conv.weight).to(dev)
        b = t(lambda: ops.conv2d_taps_nhwc(x, wt, sc, sh, k, s, k // 2, True, None))
    print("%4dx%-4d %4d->%-4d k%d s%d  small %s   taps/1x1 %.4f" % (h, w, cin, cout, k, s, "%.4f" % a if a else "  -   ", b))
