import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from estdepth_amd import DepthNetHybrid, synth
dev = torch.device("cuda:0")
torch.backends.cudnn.benchmark = True
m = DepthNetHybrid(ndepths=64, depth_min=0.1, depth_max=10.0, resnet=50).eval()
synth.fill_state_dict(m, seed=0, head_gain=1.0)
m = m.to(dev)
x = torch.randn(5, 3, 480, 640, device=dev)
def timeit(fn, n=10):
    for _ in range(3): fn()
    torch.cuda.synchronize(); t = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t) / n * 1e3
with torch.no_grad():
    ref = m.matchingFeature(x)
    print("PSM NCHW ms", timeit(lambda: m.matchingFeature(x)))
    print("R50 NCHW ms", timeit(lambda: m.semanticFeature(x[:3])))
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        y = m.matchingFeature(x)
    print("PSM NCHW graph ms", timeit(lambda: g.replay()))
    m2 = m.matchingFeature.to(memory_format=torch.channels_last)
    xc = x.contiguous(memory_format=torch.channels_last)
    out = m2(xc)
    print("PSM channels_last ms", timeit(lambda: m2(xc)), "max diff vs NCHW", (out - ref).abs().max().item())
    r2 = m.semanticFeature.to(memory_format=torch.channels_last)
    print("R50 channels_last ms", timeit(lambda: r2(xc[:3])))

with torch.no_grad():
    m3 = m.matchingFeature.use_hip_convs()
    out3 = m3(xc)
    print("PSM HIP convs ms", timeit(lambda: m3(xc)), "max diff vs NCHW torch", (out3 - ref).abs().max().item())
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        y = m3(xc)
    print("PSM HIP convs graph ms", timeit(lambda: g.replay()))
