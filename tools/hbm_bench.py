"""Micro-benchmark of the HBM-bound kernels at cfg2 size with their algorithmic bytes."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
from estdepth_amd import synth, ops

dev = torch.device("cuda:0")
D, H, W = 64, 120, 160
vox = D * H * W
g = torch.Generator(device=dev).manual_seed(0)

def timeit(fn, n=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e-3

def report(name, sec, nbytes):
    print("%-28s %8.1f us  %7.1f MB  %6.2f TB/s  (%.0f%% of 8 TB/s)" % (name, sec * 1e6, nbytes / 1e6, nbytes / sec / 1e12, nbytes / sec / 8e12 * 100))

K = torch.from_numpy(synth.intrinsics(480, 640)).clone(); K[:2] *= 0.25
K = K.to(dev)
poses = [torch.from_numpy(synth.camera_pose(v)).to(dev) for v in range(5)]
dv = (torch.arange(D, dtype=torch.float32) * (9.9 / 63) + 0.1).to(dev)
src = torch.randn(H, W, 32, device=dev, generator=g); ref = torch.randn(H, W, 32, device=dev, generator=g)
proj = ops.cam_sweep_proj(poses[1], poses[0], K)
out = torch.empty(D, H, W, 32, device=dev)
report("homo_warp_costvol", timeit(lambda: ops.homo_warp_costvol(src, ref, proj, dv, D, out=out)), 4 * (2 * 32 * H * W + 32 * vox))
kvs = [torch.randn(D, H, W, 32, device=dev, generator=g) for _ in range(4)]
for n in (1, 2, 3):
    mats = torch.stack([ops.cam_volume_mats(poses[j + 1], poses[0], K) for j in range(n)])
    report("warp_attention N=%d" % n, timeit(lambda: ops.warp_attention(kvs[0], kvs[1:1 + n], mats, dv, 0.1, 9.9 / 63)), 4 * 16 * vox * (2 + 2 * n))   # SURVEY §8d algorithmic bytes (K_t, h, K_j, V_j)
xh = kvs[0]; ru = kvs[1]; st = torch.tensor([0.1, 1.1, -0.1, 0.9], device=dev)
gm = torch.ones(16, device=dev); bt = torch.zeros(16, device=dev)
report("gru_reset_apply", timeit(lambda: ops.gru_reset_apply(xh, ru, st, gm, bt)), 4 * vox * (32 + 16 + 32))
o_raw = torch.randn(D, H, W, 16, device=dev, generator=g)
report("gru_blend", timeit(lambda: ops.gru_blend(xh, ru, o_raw, st, st, gm, bt, gm, bt, kvs[2], 32)), 4 * vox * (16 + 16 + 16 + 16))
lg = torch.randn(3, D, H, W, device=dev, generator=g)
report("softargmin_up (T=3)", timeit(lambda: ops.softargmin_up(lg, dv, 4)), 4 * 3 * (vox + 2 * 16 * H * W))
f = torch.randn(32, H, W, device=dev, generator=g); wm = torch.randn(32, 32, device=dev, generator=g)
report("mix1x1", timeit(lambda: ops.mix1x1(f, wm, None)), 4 * 2 * 32 * H * W)
