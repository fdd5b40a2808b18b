"""Per-tile start stamps of the 32->32 conv3d kernels (needs a -DESTD_TIMELINE build passed via ESTD_LIB):
cycles per tile (s_memtime = shader clock), shader clock derived from wall_clock64 (100 MHz), matrix-pipe busy fraction.
    ESTD_LIB=... python tools/tile_timeline.py [N] [f32|bf16x3|wino]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
from estdepth_amd import synth, ops
from estdepth_amd.layers_op import ConvBN3d
N = int(sys.argv[1]) if len(sys.argv) > 1 else 3
arith = sys.argv[2] if len(sys.argv) > 2 else "f32"
D, H, W = 64, 120, 160
ops.CONV3D_ARITH = "f32" if arith == "wino" else arith
ops.CONV3D_ALGO = "wino" if arith == "wino" else "direct"
dev = torch.device("cuda:0")
mod = ConvBN3d(32, 32, 3, 1, 1, "relu").eval(); synth.fill_state_dict(mod, seed=1)
plan = mod.to(dev).plan()
x = torch.randn(N, D, H, W, 32, device=dev); y = torch.empty_like(x)
nb = ops.conv3d_grid(N, D, H, W)
st = torch.zeros(nb * 4, device=dev, dtype=torch.float64)
for _ in range(30):                       # let the power management settle
    plan.run(x, (N, D, H, W), out=y, out_stride=32)
plan.run(x, (N, D, H, W), out=y, out_stride=32, stats_partials=st)
torch.cuda.synchronize()
a = st.cpu().numpy().reshape(-1, 4)
a = a[a[:, 0] > 0]
t, b, wc = a[:, 0], a[:, 1].astype(int), a[:, 2]
per = {}
for k in np.argsort(t):
    per.setdefault(b[k], []).append((t[k], wc[k]))
c, w = [], []
for blk, v in per.items():
    v = np.array(v)
    c.append(np.diff(v[:, 0])); w.append(np.diff(v[:, 1]))
c = np.concatenate(c); w = np.concatenate(w)
# MFMA floor per tile and CU: f32: 2 workgroups x 27 taps x 32 MFMAs x 32 cycles per SIMD = 55296 per pair of tiles -> 27648 per tile
#                             split: 2 waves/SIMD x 27 x 24 x 16 = 20736 per tile
#                             wino (F(2,3) along depth): one workgroup, 2 waves/SIMD x 36 x 16 MFMAs x 32 cycles = 36864 per 2-plane tile
floor = 27648.0 * 2 if arith == "f32" else 36864.0 if arith == "wino" else 20736.0
print("%s N=%d: tiles %d, workgroups %d, cycles per tile mean %.0f (p10 %.0f p90 %.0f) -> pipe busy %.1f %% ; shader clock %.0f MHz ; span %.3f ms" % (
    arith, N, len(t), len(per), c.mean(), np.percentile(c, 10), np.percentile(c, 90), 100 * floor / c.mean(),
    100.0 * c.sum() / w.sum(), (wc.max() - wc.min()) / 1e5))
