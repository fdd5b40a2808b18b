"""dres2 (33 -> 33, 3 volumes of 64x120x160) as one launch of the two-axis kernel's 33 -> 33 instance (ESTD_W3_XOUT=0) against the 33 -> 32 launch of the
three-axis kernel + output channel 32 as a pass of its own (csrc/conv3d_xout.hip), and that pass alone.   python tools/xout_bench.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from estdepth_amd import ops, _native as N
from estdepth_amd.microbench import warm
dev = torch.device("cuda:0")
Nn, D, H, W = 3, 64, 120, 160
g = torch.Generator().manual_seed(1)
w = torch.randn(33, 33, 3, 3, 3, generator=g) * 0.05
plan = ops.Conv3dPlan(w, list(range(1, 33)), 0, list(range(33)), 3, torch.ones(33), torch.zeros(33), act_a="relu", device=dev)
x = torch.randn(Nn, D, H, W, 32, device=dev); e = torch.randn(Nn, D, H, W, device=dev); y = torch.empty_like(x); ex = torch.empty(Nn, D, H, W, device=dev)
def t(f, n=30):
    warm(f, 0.15)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): f()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n
for xo in (False, True, False, True):
    ops.W3_XOUT = xo
    print("dres2 33->33 N=3, %s: %.4f ms" % ("33->32 on the three-axis kernel + channel 32 alone" if xo else "two-axis kernel's 33->33 instance", t(lambda: plan.run(x, (Nn, D, H, W), in_extra=e, out=y, out_extra=ex))))
import ctypes
d = N.Conv3dDesc()
d.N, d.D, d.H, d.W, d.cin_main, d.in_stride, d.n_tiles = Nn, D, H, W, 32, 32, 3
d.in_main, d.in_extra, d.scale, d.shift = x.data_ptr(), e.data_ptr(), plan.scale.data_ptr(), plan.shift.data_ptr()
d.act_a = d.act_b = 1
d.w_xout, d.out_extra = plan.w_xout_taps.data_ptr(), ex.data_ptr()
st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
ms = t(lambda: N.lib().estd_conv3d_k3_xout(ctypes.byref(d), st))
vox = Nn * D * H * W
print("output channel 32 alone: %.4f ms  (%.0f MB read = %.2f TB/s; %.1f GFLOP = %.1f TFLOP/s algorithmic)" % (ms, vox * 132 / 1e6, vox * 132 / ms / 1e9, 2 * 27 * 33 * vox / 1e9, 2 * 27 * 33 * vox / ms / 1e9))
