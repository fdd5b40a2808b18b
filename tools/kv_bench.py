"""Time the 33-channel convolutions (key||value 33 -> 32, dres2 33 -> 33; 3 volumes of 64x120x160): direct vs Winograd kernel."""
import sys, os
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import torch
from estdepth_amd import ops
from estdepth_amd.microbench import warm
from estdepth_amd import _native
dev = torch.device("cuda:0")
ALGOS = ("direct", "wino", "wino2", "wino", "wino2") if _native.has_ab() else ("direct", "wino2", "wino2")      # depth-only kernel: ESTD_BUILD_AB=1 builds
N, D, H, W = 3, 64, 120, 160
g = torch.Generator().manual_seed(1)
w = torch.randn(32, 33, 3, 3, 3, generator=g) * 0.05
plan = ops.Conv3dPlan(w, list(range(32)), 32, list(range(32)), 2, torch.ones(32), torch.zeros(32), act_a="relu", device=dev)
x = torch.randn(N, D, H, W, 32, device=dev); e = torch.randn(N, D, H, W, device=dev); y = torch.empty_like(x)
for algo in ALGOS:
    ops.CONV3D_ALGO = algo
    warm(lambda: plan.run(x, (N, D, H, W), in_extra=e, out=y), 0.15)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20): plan.run(x, (N, D, H, W), in_extra=e, out=y)
    e1.record(); torch.cuda.synchronize()
    print(algo, "kv 33->32 N=3: %.4f ms" % (e0.elapsed_time(e1) / 20))

w = torch.randn(33, 33, 3, 3, 3, generator=g) * 0.05
plan = ops.Conv3dPlan(w, list(range(1, 33)), 0, list(range(33)), 3, torch.ones(33), torch.zeros(33), act_a="relu", device=dev)
ex = torch.empty(N, D, H, W, device=dev)
for algo in ALGOS:
    ops.CONV3D_ALGO = algo
    warm(lambda: plan.run(x, (N, D, H, W), in_extra=e, out=y, out_extra=ex), 0.15)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20): plan.run(x, (N, D, H, W), in_extra=e, out=y, out_extra=ex)
    e1.record(); torch.cuda.synchronize()
    print(algo, "dres2 33->33 N=3: %.4f ms" % (e0.elapsed_time(e1) / 20))
