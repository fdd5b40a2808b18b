"""The ConvGRU with the reset gate folded into the output convolution's plane loads (ESTD_GATE_IN_CONV=1, default) against the gate as
a pass of its own (estd_gru_reset_apply): agreement of the fused value, and the time of one GRU (transformer/epipolar_transformer.py:31-54,80-83)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from estdepth_amd import epipolar_transformer as ET, synth
from estdepth_amd.microbench import warm
dev = torch.device("cuda:0")
et = ET.EpipolarTransformer(16, 16, 3).eval()
synth.fill_state_dict(et, seed=4)
et = et.to(dev)
for dims in ((5, 13, 21), (8, 24, 40), (64, 120, 160)):
    D, H, W = dims
    g = torch.Generator(device=dev).manual_seed(3)
    xh = torch.randn(D, H, W, 32, device=dev, generator=g)
    outs = {}
    for mode in (False, True):
        ET.GATE_IN_CONV = mode
        o = torch.zeros(D, H, W, 16, device=dev)
        with torch.no_grad():
            et.gru(xh, dims, o, 16)
        torch.cuda.synchronize()
        outs[mode] = o
    d = float((outs[True] - outs[False]).abs().max())
    print("dims %s: max |fold - pass| = %.3g  (max |value| %.3g)" % (dims, d, float(outs[False].abs().max())))
    assert d < 2e-5, d
D, H, W = 64, 120, 160
xh = torch.randn(D, H, W, 32, device=dev)
o = torch.zeros(D, H, W, 16, device=dev)
for rep in range(2):
    for mode in (False, True):
        ET.GATE_IN_CONV = mode
        with torch.no_grad():
            warm(lambda: et.gru(xh, (D, H, W), o, 16), 0.2)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(30):
                et.gru(xh, (D, H, W), o, 16)
            e1.record()
            torch.cuda.synchronize()
        print("one ConvGRU (gate conv + norms + output conv + blend), gate %s: %.4f ms" % ("folded into the output convolution" if mode else "as its own pass", e0.elapsed_time(e1) / 30))
