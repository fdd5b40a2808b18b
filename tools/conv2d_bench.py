"""Time the PSM 3x3 convolutions (5 images, cfg2) on both arithmetics.  python tools/conv2d_bench.py"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from estdepth_amd import ops
from estdepth_amd.microbench import warm
from estdepth_amd import _native
AB = _native.has_ab()          # row-only Winograd / bf16 operand split: ESTD_BUILD_AB=1 builds
dev = "cuda"
for (h, w, cin, cout, dil) in [(240, 320, 32, 32, 1), (120, 160, 64, 64, 1), (120, 160, 128, 128, 1), (120, 160, 320, 128, 1), (120, 160, 128, 128, 2)]:
    conv = torch.nn.Conv2d(cin, cout, 3, 1, dil, dil, bias=False).to(dev)
    bn = torch.nn.BatchNorm2d(cout).to(dev).eval()
    plan = ops.Conv2dPlan(conv, bn, relu_before=True)
    x = torch.randn(5, h, w, cin, device=dev)
    gf = 2.0 * 9 * 5 * h * w * cin * cout / 1e9
    line = "%3dx%-3d %3d->%-3d dil %d" % (h, w, cin, cout, dil)
    for arith in (("f32/wino2", "f32/wino", "f32/direct", "bf16x3") if AB else ("f32/wino2", "f32/direct")):
        ops.CONV2D_ARITH = arith.split("/")[0]
        ops.CONV2D_ALGO = arith.split("/")[1] if "/" in arith else "wino"
        warm(lambda: plan.run(x), 0.1)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20): plan.run(x)
        e1.record(); torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / 20
        line += "   %s %.4f ms (%.1f TF/s)" % (arith, ms, gf / ms)
    print(line)
    if cin == cout:                                         # BasicBlock tail: conv + BN + residual add
        res = torch.randn(5, h, w, cout, device=dev)
        line = "%3dx%-3d %3d->%-3d dil %d + residual" % (h, w, cin, cout, dil)
        for algo in (("wino2", "wino", "direct") if AB else ("wino2", "direct")):
            ops.CONV2D_ARITH, ops.CONV2D_ALGO = "f32", algo
            warm(lambda: plan.run(x, residual=res), 0.1)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(20): plan.run(x, residual=res)
            e1.record(); torch.cuda.synchronize()
            line += "   f32/%s %.4f ms" % (algo, e0.elapsed_time(e1) / 20)
        print(line)
