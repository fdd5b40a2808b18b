"""Timeline of conv2d_wino_kernel work items (needs a -DESTD_C2TIME build passed via ESTD_LIB, ESTD_BINDING=ctypes): s_memtime stamps of
the four waves of the first workgroups over their first items.  Points: 0 item top, 1 transform written, 2 barrier passed, 3 tap loops
done (MFMAs issued), 4 epilogue issued.      ESTD_LIB=... ESTD_BINDING=ctypes python tools/conv2d_timeline.py [cin] [H] [W]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
from estdepth_amd import ops
cin = int(sys.argv[1]) if len(sys.argv) > 1 else 64
H = int(sys.argv[2]) if len(sys.argv) > 2 else 120
W = int(sys.argv[3]) if len(sys.argv) > 3 else 160
dev = torch.device("cuda:0")
conv = torch.nn.Conv2d(cin, cin, 3, 1, 1, bias=False).to(dev)
bn = torch.nn.BatchNorm2d(cin).eval().to(dev)
ops.CONV2D_ALGO = "wino"
plan = ops.Conv2dPlan(conv, bn, relu_before=True, relu_after=False)
x = torch.randn(5, H, W, cin, device=dev)
res = torch.zeros(5, H, W, cin, device=dev)
for _ in range(10):
    plan.run(x, residual=res)
torch.cuda.synchronize()
res.zero_()
plan.run(x, residual=res)
torch.cuda.synchronize()
a = res.view(-1).cpu().numpy().view(np.uint64)[:16 * 8 * 4 * 8].reshape(16, 8, 4, 8).astype(np.int64)
names = ["transform", "barrier", "taps", "epilogue", "->next"]
for wg in (0, 1, 8):
    print("workgroup %d" % wg)
    for it in range(0, 6):
        for wv in range(4):
            r = a[wg, it, wv]
            if r[0] == 0:
                continue
            nxt = a[wg, it + 1, wv, 0] if a[wg, it + 1, wv, 0] else r[4]
            d = [r[1] - r[0], r[2] - r[1], r[3] - r[2], r[4] - r[3], nxt - r[4]]
            print("  item %d wave %d: " % (it, wv) + "  ".join("%s %6d" % (n, v) for n, v in zip(names, d)) + "  | total %6d  start %+7d" % (nxt - r[0], r[0] - a[wg, 0, 0, 0]))
