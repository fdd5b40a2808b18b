"""Time the dominant kernel (3x3x3 conv 32->32, cfg2 volume) in isolation.  python tools/conv_bench.py [N] [iters]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from estdepth_amd import synth, ops
from estdepth_amd.microbench import warm
from estdepth_amd.layers_op import ConvBN3d

N = int(sys.argv[1]) if len(sys.argv) > 1 else 1
iters = int(sys.argv[2]) if len(sys.argv) > 2 else 20
D = int(os.environ.get('CB_D', '64')); H = int(os.environ.get('CB_H', '120')); W = int(os.environ.get('CB_W', '160'))
dev = torch.device("cuda:0")
mod = ConvBN3d(32, 32, 3, 1, 1, "relu").eval()
synth.fill_state_dict(mod, seed=1)
plan = mod.to(dev).plan()
x = torch.randn(N, D, H, W, 32, device=dev)
y = torch.empty_like(x)
if os.environ.get("CB_COLD"):                             # the old protocol: three warm-up launches on an idle GPU (clocks still ramping)
    for _ in range(3):
        plan.run(x, (N, D, H, W), out=y, out_stride=32)
    torch.cuda.synchronize()
else:
    warm(lambda: plan.run(x, (N, D, H, W), out=y, out_stride=32))          # sustained clocks (estdepth_amd/microbench.py)
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(iters):
    plan.run(x, (N, D, H, W), out=y, out_stride=32)
e1.record()
torch.cuda.synchronize()
ms = e0.elapsed_time(e1) / iters
gf = N * 2 * 27 * 32 * 32 * D * H * W / 1e9
print("conv3d 32->32 N=%d D=%d: %.4f ms  %.1f TFLOP/s  (%.1f%% of 157.3)" % (N, D, ms, gf / ms, gf / ms / 157.3 * 100))
if os.environ.get("CB_EPI"):                               # the read-back epilogues of pre1 (running sum) / pre2 (two residuals)
    r1, r2 = torch.randn_like(x), torch.randn_like(x)
    for name, kw in (("accumulate", dict(accumulate=True)), ("residual", dict(residual=r1)),
                     ("2 residuals + scale", dict(residual=r1, residual2=r2, out_scale=0.5))):
        warm(lambda: plan.run(x, (N, D, H, W), out=y, out_stride=32, **kw), 0.15)
        e0.record()
        for _ in range(iters):
            plan.run(x, (N, D, H, W), out=y, out_stride=32, **kw)
        e1.record()
        torch.cuda.synchronize()
        print("   + %-20s %.4f ms" % (name, e0.elapsed_time(e1) / iters))
