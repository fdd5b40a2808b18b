"""cfg5 stress (960x1280, D=128, ESTM windows) on the GPU: runs, memory, timing, sanity properties."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from estdepth_amd import DepthNetHybrid, synth
dev = torch.device("cuda:0")
torch.backends.cudnn.benchmark = True
m = DepthNetHybrid(ndepths=128, depth_min=0.1, depth_max=10.0, resnet=50, IF_EST_transformer=True).eval()
synth.fill_state_dict(m, seed=0, head_gain=1.0)
m = m.to(dev)
imgs, poses, intr, sample = synth.make_sequence(5, 960, 1280, seed=1005)
imgs, poses, intr = imgs.to(dev), poses.to(dev), intr.to(dev)
sub = lambda sl: {k: v[:, sl].to(dev) for k, v in sample.items()}
mem_c, mem_p = [], []
with torch.no_grad():
    for w in range(3):
        sl = slice(w, w + 3)
        pc = {"keys": [c["keys"][0] for c in mem_c], "values": [c["values"][0] for c in mem_c]} if mem_c else None
        pp = [p[0] for p in mem_p] if mem_p else None
        torch.cuda.synchronize(); t = time.perf_counter()
        out, c, p = m(imgs[:, sl], poses[:, sl], intr, sub(sl), pc, pp, mode="val")
        torch.cuda.synchronize(); dt = time.perf_counter() - t
        mem_c.append(c); mem_p.append(p)
        mem_c, mem_p = mem_c[-2:], mem_p[-2:]
        d2 = out[("depth", 0, 2)]
        print("window %d: %.1f ms  depth2 [%.3f, %.3f] finite=%s prob max %.4f" % (
            w, dt * 1e3, d2.min().item(), d2.max().item(), bool(torch.isfinite(d2).all()), out[("fused_prob", 0)].max().item()))
    # timed steady-state window
    torch.cuda.synchronize(); t = time.perf_counter()
    for _ in range(3):
        out, c, p = m(imgs[:, 2:5], poses[:, 2:5], intr, sub(slice(2, 5)), pc, pp, mode="val")
    torch.cuda.synchronize()
    print("cfg5 steady window: %.1f ms  (%.2f depth frames/s)" % ((time.perf_counter() - t) / 3 * 1e3, 3 / (time.perf_counter() - t)))
print("max mem GB", torch.cuda.max_memory_allocated() / 2**30)
