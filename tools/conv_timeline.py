"""Debug: per-tile start stamps of the conv kernel (needs a -DESTD_TIMELINE build passed via ESTD_LIB)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
from estdepth_amd import synth, ops
from estdepth_amd.layers_op import ConvBN3d
N = int(sys.argv[1]) if len(sys.argv) > 1 else 1
D, H, W = 64, 120, 160
dev = torch.device("cuda:0")
mod = ConvBN3d(32, 32, 3, 1, 1, "relu").eval(); synth.fill_state_dict(mod, seed=1)
plan = mod.to(dev).plan()
x = torch.randn(N, D, H, W, 32, device=dev); y = torch.empty_like(x)
nb = ops.conv3d_grid(N, D, H, W)
st = torch.zeros(nb * 4, device=dev, dtype=torch.float64)
for _ in range(3):
    plan.run(x, (N, D, H, W), out=y, out_stride=32)
torch.cuda.synchronize()
plan.stats_never = True
import ctypes
# run once with the stamp buffer (the ESTD_TIMELINE build skips the GroupNorm sums)
from estdepth_amd import _native as NV
plan.run(x, (N, D, H, W), out=y, out_stride=32, stats_partials=st)
torch.cuda.synchronize()
a = st.cpu().numpy().reshape(-1, 4)
t, b, wc = a[:, 0], a[:, 1].astype(int), a[:, 2]
t0 = t.min()
per_block = {}
for k in np.argsort(t):
    per_block.setdefault(b[k], []).append(t[k] - t0)
durs = []
for blk, ts in per_block.items():
    d = np.diff(ts)
    durs.append(d)
L = min(len(d) for d in durs)
M = np.stack([d[:L] for d in durs])
print("tiles per block ~", L + 1, "blocks", len(durs))
print("mean tile duration (cycles of s_memtime) by index k:")
print(np.round(M.mean(0)).astype(int))
print("p10:", np.round(np.percentile(M, 10, axis=0)).astype(int))
print("p90:", np.round(np.percentile(M, 90, axis=0)).astype(int))
starts = np.array([ts[0] for ts in per_block.values()])
print("first-tile start spread (cycles): min %d p50 %d max %d" % (starts.min(), np.median(starts), starts.max()))
print("kernel span cycles", t.max() - t0, "wall_clock span (100MHz ticks)", wc.max() - wc.min())
