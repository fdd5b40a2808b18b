"""Debug: per-tile start stamps of conv3d_k3_split_kernel (needs a -DESTD_TIMELINE build passed via ESTD_LIB).
Prints cycles per tile (s_memtime = shader clock), the shader clock derived from wall_clock64 (100 MHz) and the
matrix-pipe busy fraction (2 waves/SIMD x 27 taps x 24 MFMAs x 16 cycles = 20736 cycles per tile)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
from estdepth_amd import synth, ops
from estdepth_amd.layers_op import ConvBN3d
N = int(sys.argv[1]) if len(sys.argv) > 1 else 3
D, H, W = 64, 120, 160
ops.CONV3D_ARITH = "bf16x3"
dev = torch.device("cuda:0")
mod = ConvBN3d(32, 32, 3, 1, 1, "relu").eval(); synth.fill_state_dict(mod, seed=1)
plan = mod.to(dev).plan()
x = torch.randn(N, D, H, W, 32, device=dev); y = torch.empty_like(x)
nb = ops.conv3d_grid(N, D, H, W)
st = torch.zeros(nb * 4, device=dev, dtype=torch.float64)
for _ in range(20):
    plan.run(x, (N, D, H, W), out=y, out_stride=32)
plan.run(x, (N, D, H, W), out=y, out_stride=32, stats_partials=st)
torch.cuda.synchronize()
a = st.cpu().numpy().reshape(-1, 4)
a = a[a[:, 0] > 0]
t, b, wc = a[:, 0], a[:, 1].astype(int), a[:, 2]
per = {}
for k in np.argsort(t):
    per.setdefault(b[k], []).append((t[k], wc[k]))
d_cyc, d_wall = [], []
for blk, v in per.items():
    v = np.array(v)
    d_cyc.append(np.diff(v[:, 0])); d_wall.append(np.diff(v[:, 1]))
c = np.concatenate(d_cyc); w = np.concatenate(d_wall)
mhz = 100.0 * c.sum() / w.sum()
print("tiles stamped %d, blocks %d" % (len(t), len(per)))
print("cycles per tile: mean %.0f p10 %.0f p50 %.0f p90 %.0f  -> matrix pipe busy %.1f %%" % (c.mean(), np.percentile(c, 10), np.median(c), np.percentile(c, 90), 100 * 20736 / c.mean()))
print("shader clock %.0f MHz ; kernel span %.3f ms (wall_clock64)" % (mhz, (wc.max() - wc.min()) / 1e5))
