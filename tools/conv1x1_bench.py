"""1x1 convolutions of ResNet-50 at the semantic branch's sizes (3 images of 480x640): csrc/conv1x1.hip (conv + BN + residual + ReLU
in one launch) vs the library path (hipBLASLt GEMM with bias/ReLU epilogue, or GEMM + the BN/add/ReLU pass).   python tools/conv1x1_bench.py"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from estdepth_amd import ops
from estdepth_amd.microbench import warm
dev = "cuda"
def t(f, n=30):
    warm(f, 0.1)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): f()
    e1.record(); torch.cuda.synchronize()
    return 1e3 * e0.elapsed_time(e1) / n
N = 3
SHAPES = [  # name, H, W, cin, cout, stride, residual
    ("l1 conv1 64->64", 120, 160, 64, 64, 1, False), ("l1 conv3 64->256 +res", 120, 160, 64, 256, 1, True), ("l1 ds 64->256", 120, 160, 64, 256, 1, False),
    ("l1 conv1 256->64", 120, 160, 256, 64, 1, False), ("l2 conv1 256->128", 120, 160, 256, 128, 1, False), ("l2 conv3 128->512 +res", 60, 80, 128, 512, 1, True),
    ("l2 ds 256->512 s2", 120, 160, 256, 512, 2, False), ("l2 conv1 512->128", 60, 80, 512, 128, 1, False), ("l3 conv1 512->256", 60, 80, 512, 256, 1, False),
    ("l3 conv3 256->1024 +res", 30, 40, 256, 1024, 1, True), ("l3 ds 512->1024 s2", 60, 80, 512, 1024, 2, False), ("l3 conv1 1024->256", 30, 40, 1024, 256, 1, False),
    ("l4 conv1 1024->512", 30, 40, 1024, 512, 1, False), ("l4 conv3 512->2048 +res", 15, 20, 512, 2048, 1, True), ("l4 ds 1024->2048 s2", 30, 40, 1024, 2048, 2, False),
    ("l4 conv1 2048->512", 15, 20, 2048, 512, 1, False)]
tot_h = tot_l = 0.0
for name, H, W, cin, cout, s, res in SHAPES:
    x = torch.randn(N, H, W, cin, device=dev)
    w = torch.randn(cout, cin, device=dev) / cin ** 0.5
    sc, sh = torch.rand(cout, device=dev) + 0.5, torch.randn(cout, device=dev)
    Ho, Wo = (H - 1) // s + 1, (W - 1) // s + 1
    r = torch.randn(N, Ho, Wo, cout, device=dev) if res else None
    a = t(lambda: ops.conv1x1_nhwc(x, w, sc, sh, s, True, r))
    wt = (w * sc[:, None]).t().contiguous()
    def lib():
        xs = x[:, ::s, ::s].contiguous() if s > 1 else x
        x2 = xs.reshape(-1, cin)
        if r is None:
            return torch._addmm_activation(sh, x2, wt, use_gelu=False)
        y = torch.mm(x2, w.t()).view(N, Ho, Wo, cout).permute(0, 3, 1, 2)
        return ops.bn_act_nhwc_(y, sc, sh, True, r.permute(0, 3, 1, 2))
    b = t(lib)
    gf = 2.0 * N * Ho * Wo * cin * cout / 1e9
    mb = 4.0 * N * (Ho * Wo * cin + Ho * Wo * cout * (2 if res else 1)) / 1e6
    tot_h += a; tot_l += b
    print("%-26s hip %7.1f us (%6.1f TF/s, %5.0f GB/s)   library %7.1f us   x%.2f" % (name, a, gf * 1e3 / a, mb * 1e3 / a, b, b / a))
print("sum over these 16 shapes: hip %.1f us, library %.1f us" % (tot_h, tot_l))
