"""Intra-tile timeline of conv3d_wino2_kernel (needs a -DESTD_W2TIME build passed via ESTD_LIB, ESTD_BINDING=ctypes): s_memtime
stamps of waves 0 and 4 (the two waves of SIMD 0) of workgroup 0 over its first 8 tiles.
Points: 0 tile top, 1 first MFMA, 2/3/4 steps 6/12/18, 5 step 19 (after the in-loop slice rewrite), 6 loop end, 7 after the
post-loop barriers, 8 after the epilogue.   ESTD_LIB=... ESTD_BINDING=ctypes python tools/wino2_timeline.py [N]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
from estdepth_amd import synth, ops
from estdepth_amd.layers_op import ConvBN3d
N = int(sys.argv[1]) if len(sys.argv) > 1 else 3
D, H, W = 64, 120, 160
ops.CONV3D_ALGO = "wino2"
dev = torch.device("cuda:0")
mod = ConvBN3d(32, 32, 3, 1, 1, "relu").eval(); synth.fill_state_dict(mod, seed=1)
plan = mod.to(dev).plan()
x = torch.randn(N, D, H, W, 32, device=dev); y = torch.empty_like(x)
st = torch.zeros(max(ops.conv3d_grid(N, D, H, W) * 4, 4096), device=dev, dtype=torch.float64)
for _ in range(20):
    plan.run(x, (N, D, H, W), out=y, out_stride=32)
plan.run(x, (N, D, H, W), out=y, out_stride=32, stats_partials=st)
torch.cuda.synchronize()
a = st.cpu().numpy()[:2 * 8 * 16].reshape(2, 8, 16)
names = ["top->mfma0", "steps 0-5", "steps 6-11", "steps 12-17", "barrier+slices012", "steps 19-23", "post barriers+slice3", "epilogue", "-> next top"]
for wv in range(2):
    print("wave %d (SIMD 0)%s" % (wv * 4, "" if wv == 0 else " (partner)"))
    for t in range(1, 7):
        row = a[wv, t]
        nxt = a[wv, t + 1, 0]
        d = [row[k + 1] - row[k] for k in range(8)] + [nxt - row[8]]
        print("  tile %d: " % t + "  ".join("%s %5.0f" % (n, v) for n, v in zip(names, d)) + "   | total %6.0f" % (nxt - row[0]))
