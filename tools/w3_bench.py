"""A/B of the 32 -> 32 Winograd convolution: the THREE-axis kernel (csrc/conv3d_wino3.hip, ops.W3 = True) against the two-axis 8-wave kernel
(csrc/conv3d_wino2.hip).  Correctness first (every epilogue, ragged sizes, against the 8-wave kernel and against an fp64 convolution), then
timings at sustained clocks.   python tools/w3_bench.py [N] [iters]       (W3_NOCHECK=1: timings only)"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import torch.nn.functional as F
from estdepth_amd import synth, ops
from estdepth_amd.microbench import warm
from estdepth_amd.layers_op import ConvBN3d

N = int(sys.argv[1]) if len(sys.argv) > 1 else 3
iters = int(sys.argv[2]) if len(sys.argv) > 2 else 30
dev = torch.device("cuda:0")
mod = ConvBN3d(32, 32, 3, 1, 1, "relu").eval()
synth.fill_state_dict(mod, seed=1)
plan = mod.to(dev).plan()


def run(x, dims, w2x, **kw):
    ops.W3 = w2x
    y = torch.zeros(dims + (32,), device=dev)
    if kw.get("accumulate"):
        y += 0.25
    plan.run(x, dims, out=y, out_stride=32, **kw)
    torch.cuda.synchronize()
    return y


if not os.environ.get("W3_NOCHECK"):
    g = torch.Generator(device=dev).manual_seed(5)
    for dims in ((1, 4, 8, 16), (2, 5, 13, 21), (1, 7, 24, 40), (1, 64, 120, 160)):
        x = torch.randn(dims + (32,), device=dev, generator=g)
        r1, r2 = torch.randn_like(x), torch.randn_like(x)
        for name, kw in (("plain", {}), ("accumulate", dict(accumulate=True)), ("residual", dict(residual=r1)),
                         ("2 residuals + scale", dict(residual=r1, residual2=r2, out_scale=0.5)),
                         ("residual + accumulate", dict(residual=r1, accumulate=True))):
            a, b = run(x, dims, False, **kw), run(x, dims, True, **kw)
            print("%-18s %-24s max |new - old| = %.3g  (max |old| %.3g)" % (dims, name, float((a - b).abs().max()), float(a.abs().max())))
        if dims[1] * dims[2] * dims[3] <= 7 * 24 * 40:
            conv, bn = mod[0], mod[1]
            xin = x.permute(0, 4, 1, 2, 3).double().cpu()
            ref = F.conv3d(xin, conv.weight.double().cpu(), padding=1)
            sc = (bn.weight.double() / torch.sqrt(bn.running_var.double() + bn.eps)).cpu()
            ref = torch.relu(ref * sc.view(1, -1, 1, 1, 1) + (bn.bias.double().cpu() - bn.running_mean.double().cpu() * sc).view(1, -1, 1, 1, 1))
            ref = ref.permute(0, 2, 3, 4, 1)
            a, b = run(x, dims, False), run(x, dims, True)
            print("%-18s vs fp64: old %.3g  new %.3g  (max |ref| %.3g)" % (dims, float((a.cpu().double() - ref).abs().max()),
                                                                          float((b.cpu().double() - ref).abs().max()), float(ref.abs().max())))

D, H, W = 64, 120, 160
x = torch.randn(N, D, H, W, 32, device=dev)
y = torch.empty_like(x)
r1, r2 = torch.randn_like(x), torch.randn_like(x)
gf = N * 2 * 27 * 32 * 32 * D * H * W / 1e9
cases = (("plain", {}), ("accumulate", dict(accumulate=True)), ("residual", dict(residual=r1)),
         ("2 residuals + scale", dict(residual=r1, residual2=r2, out_scale=0.5)))
for rep in range(2):
    for name, kw in cases:
        line = "%-22s" % name
        for w2x in (False, True):
            ops.W3 = w2x
            warm(lambda: plan.run(x, (N, D, H, W), out=y, out_stride=32, **kw), 0.2)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(iters):
                plan.run(x, (N, D, H, W), out=y, out_stride=32, **kw)
            e1.record()
            torch.cuda.synchronize()
            ms = e0.elapsed_time(e1) / iters
            line += "  %s %.4f ms (%.1f TF alg)" % ("3-axis" if w2x else "8-wave", ms, gf / ms)
        print("N=%d %s" % (N, line))
