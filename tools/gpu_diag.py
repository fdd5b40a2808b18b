"""GPU-box diagnostics (not a test): run-to-run determinism of the 2D backbones vs the HIP path, stage timings."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from estdepth_amd import DepthNetHybrid, synth, ops

dev = torch.device("cuda:0")
torch.backends.cudnn.benchmark = True
m = DepthNetHybrid(ndepths=64, depth_min=0.1, depth_max=10.0, resnet=50, IF_EST_transformer=True).eval()
synth.fill_state_dict(m, seed=0, head_gain=1.0)
m = m.to(dev)
imgs, poses, intr, sample = synth.make_sequence(5, 480, 640, seed=1000)
imgs, poses, intr = imgs.to(dev), poses.to(dev), intr.to(dev)
sample = {k: v.to(dev) for k, v in sample.items()}

def sync():
    torch.cuda.synchronize()

with torch.no_grad():
    x = (2 * (imgs / 255.) - 1.).reshape(5, 3, 480, 640)
    f1 = m.matchingFeature(x); f2 = m.matchingFeature(x)
    print("PSM run-to-run max diff", (f1 - f2).abs().max().item(), "mag", f1.abs().mean().item())
    def timeit(fn, n=5):
        fn(); sync(); t = time.perf_counter()
        for _ in range(n): fn()
        sync(); return (time.perf_counter() - t) / n * 1e3
    print("PSM 5 imgs ms", timeit(lambda: m.matchingFeature(x)))
    print("R50 3 imgs ms", timeit(lambda: m.semanticFeature(x[1:4])))
    sem = m.semanticFeature(x[1:4])
    print("dec2d semantic_vs ms", timeit(lambda: m.CostRegNet._semantic_vs(sem)))
    sv = m.CostRegNet._semantic_vs(sem)
    lg = torch.randn(3, 64, 120, 160, device=dev)
    print("dec2d refine ms", timeit(lambda: m.CostRegNet._refine(sv, lg, sem)))
    # hot path pieces
    feats = f1
    P = m._plans()
    srcm = [m._mix(feats[v].contiguous(), "src") for v in range(5)]
    refm = m._mix(feats[1].contiguous(), "ref")
    pz = poses[0].contiguous(); k4 = m.scale_cam_intr(intr, 0.25)[0].contiguous()
    dv = m.depth_cands.view(-1).to(dev).contiguous()
    print("costvolume (1 target) ms", timeit(lambda: m._costvolume(refm, [srcm[0], srcm[2]], pz[1], [pz[0], pz[2]], k4, dv)))
    proj = ops.cam_sweep_proj(pz[1], pz[0], k4)
    print("  homo_warp_costvol ms", timeit(lambda: ops.homo_warp_costvol(srcm[0], refm, proj, dv, 64)))
    xx = ops.homo_warp_costvol(srcm[0], refm, proj, dv, 64); yy = torch.empty_like(xx)
    t = timeit(lambda: P["pre1"].run(xx, (1, 64, 120, 160), out=yy, out_stride=32), 10)
    print("  conv3d 32->32 (N=1) ms", t, "TFLOP/s", 67.95e-3 / t)
    x3 = torch.randn(3, 64, 120, 160, 32, device=dev); y3 = torch.empty_like(x3)
    t = timeit(lambda: P["pre1"].run(x3, (3, 64, 120, 160), out=y3, out_stride=32), 10)
    print("  conv3d 32->32 (N=3) ms", t, "TFLOP/s", 3 * 67.95e-3 / t)
    o1 = torch.empty_like(xx); o2 = torch.empty_like(xx)
    P["pre1"].run(xx, (1, 64, 120, 160), out=o1, out_stride=32); P["pre1"].run(xx, (1, 64, 120, 160), out=o2, out_stride=32)
    print("  conv3d run-to-run bit-exact:", torch.equal(o1, o2))
    out, c, p = m(imgs, poses, intr, sample, None, None, mode="val")
    print("forward call-1 (no EST) ms", timeit(lambda: m(imgs, poses, intr, sample, None, None, mode="val"), 3))
    print("forward call-2 (EST, N=3) ms", timeit(lambda: m(imgs, poses, intr, sample, c, list(p), mode="val"), 3))
    a, _, _ = m(imgs, poses, intr, sample, c, list(p), mode="val")
    b, _, _ = m(imgs, poses, intr, sample, c, list(p), mode="val")
    print("forward run-to-run max diff", max((a[k] - b[k]).abs().max().item() for k in a))
    print("max mem GB", torch.cuda.max_memory_allocated() / 2**30)
