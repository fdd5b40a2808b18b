"""Time the 16->16 head conv (+1x1x1 head) and the 32->16 output conv at cfg2 size.  python tools/head_bench.py"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from estdepth_amd.microbench import warm
import torch
from estdepth_amd import ops
dev = "cuda"
D, H, W = 64, 120, 160
def t(f, n=30):
    warm(f, 0.15)                                   # sustained clocks (estdepth_amd/microbench.py)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): f()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n
g = torch.Generator().manual_seed(0)
w16 = torch.randn(16, 16, 3, 3, 3, generator=g) * 0.05
head = ops.Conv3dPlan(w16, list(range(16)), None, list(range(16)), 1, torch.ones(16), torch.zeros(16), act_a="relu",
                      head_w=torch.randn(16, generator=g), head_b=torch.zeros(1), device=dev)
w32 = torch.randn(16, 32, 3, 3, 3, generator=g) * 0.05
outc = ops.Conv3dPlan(w32, list(range(32)), None, list(range(16)), 1, torch.ones(16), torch.zeros(16), device=dev)
for N in (1, 3):
    kv = torch.randn(N, D, H, W, 32, device=dev)
    lg = torch.empty(N, D, H, W, device=dev)
    o = torch.empty(N, D, H, W, 16, device=dev)
    ops.CONV3D_ALGO = "wino2"
    a2 = t(lambda: head.run(kv, (N, D, H, W), in_stride=32, out_head=lg))
    ops.CONV3D_ALGO = "direct"
    a = t(lambda: head.run(kv, (N, D, H, W), in_stride=32, out_head=lg))
    b = t(lambda: outc.run(kv, (N, D, H, W), out=o, out_stride=16))
    ops.CONV3D_ALGO = "wino2"
    b2 = t(lambda: outc.run(kv, (N, D, H, W), out=o, out_stride=16))
    gfa, gfb = N * 2 * 27 * 16 * 16 * D * H * W / 1e9, N * 2 * 27 * 32 * 16 * D * H * W / 1e9
    print("N=%d  head 16->16 direct %.4f ms (%.1f TF/s)  wino2-c16 %.4f ms (%.1f algorithmic TF/s)" % (N, a, gfa / a, a2, gfa / a2))
    print("N=%d  head 16->16 %.4f ms (%.1f TF/s)   out 32->16 direct %.4f ms (%.1f TF/s)  wino2 %.4f ms (%.1f TF/s)" % (N, a, gfa / a, b, gfb / b, b2, gfb / b2))
