"""Which library path is fastest for the ResNet-50 convolutions at cfg2 size (3 images, 480x640)?
1x1: MIOpen conv (NHWC) vs a plain GEMM (torch.mm -> rocBLAS/hipBLASLt fp32); 3x3: MIOpen vs the MFMA conv2d kernel."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch, torch.nn.functional as F
from estdepth_amd import ops
dev = "cuda"


def t(f, n=20):
    for _ in range(3): f()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): f()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n


N = 3
print("1x1 convs: shape, MIOpen ms (TF/s), mm ms (TF/s)")
for (h, w, cin, cout) in [(120, 160, 64, 64), (120, 160, 64, 256), (120, 160, 256, 64), (60, 80, 256, 128), (60, 80, 128, 512), (60, 80, 512, 128),
                          (30, 40, 512, 256), (30, 40, 256, 1024), (30, 40, 1024, 256), (15, 20, 1024, 512), (15, 20, 512, 2048), (15, 20, 2048, 512)]:
    x = torch.randn(N, cin, h, w, device=dev).contiguous(memory_format=torch.channels_last)
    wt = torch.randn(cout, cin, 1, 1, device=dev).contiguous(memory_format=torch.channels_last)
    w2 = wt.reshape(cout, cin).t().contiguous()
    xm = x.permute(0, 2, 3, 1).reshape(-1, cin)
    gf = 2.0 * N * h * w * cin * cout / 1e9
    a = t(lambda: F.conv2d(x, wt))
    b = t(lambda: torch.mm(xm, w2))
    print("%4dx%-4d %5d->%-5d  conv %.4f (%.1f)   mm %.4f (%.1f)" % (h, w, cin, cout, a, gf / a, b, gf / b))
print("3x3 convs: MIOpen vs conv2d_mfma")
for (h, w, c) in [(120, 160, 64), (60, 80, 128), (30, 40, 256), (15, 20, 512)]:
    conv = torch.nn.Conv2d(c, c, 3, 1, 1, bias=False).to(dev).to(memory_format=torch.channels_last)
    bn = torch.nn.BatchNorm2d(c).to(dev).eval()
    x = torch.randn(N, c, h, w, device=dev).contiguous(memory_format=torch.channels_last)
    plan = ops.Conv2dPlan(conv, bn, relu_before=True)
    xn = x.permute(0, 2, 3, 1)
    gf = 2.0 * 9 * N * h * w * c * c / 1e9
    with torch.no_grad():
        a = t(lambda: conv(x))
        b = t(lambda: plan.run(xn))
    print("%4dx%-4d %5d  miopen %.4f (%.1f)   mfma %.4f (%.1f)" % (h, w, c, a, gf / a, b, gf / b))
