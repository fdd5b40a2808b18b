"""Stage A (the camera-independent 2D networks of a Joint step: PSM matching features of 5 frames on one stream, ResNet-50 + 2D decoder of the
3 targets on another) as hipGraph replays: each branch alone, both in sequence on ONE stream, both forked on two streams (what the step runs).
Says how much of the phase is plain kernel time (sum of the branches alone) and what the two-stream form gains or loses against it.
    python tools/stage_a_bench.py [reps]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import bench

reps = int(sys.argv[1]) if len(sys.argv) > 1 else 30
dev = torch.device("cuda:0")
m = bench.build_model("joint", dev)
imgs, poses, intr, sample = bench.make_inputs("joint", 0, dev)
x = imgs[:, 3:8].contiguous()


def graphed(fn, warm=3):
    s = torch.cuda.Stream()
    s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s), torch.no_grad():
        for _ in range(warm):
            fn()
    torch.cuda.current_stream().wait_stream(s)
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.no_grad(), torch.cuda.graph(g):
        keep = fn()
    return g, keep


def timeit(g, n):
    for _ in range(max(5, n // 3)):
        g.replay()
    torch.cuda.synchronize()
    t = time.perf_counter()
    for _ in range(n):
        g.replay()
    torch.cuda.synchronize()
    return (time.perf_counter() - t) / n * 1e3


def flat_of(x):
    from estdepth_amd import ops
    v = x.shape[1]
    f = ops.normalise_nhwc(x.reshape(v, 3, *x.shape[-2:]).contiguous()).permute(0, 3, 1, 2)
    return f.contiguous(memory_format=torch.channels_last)


flat = flat_of(x)


def psm():
    return m.matchingFeature(flat)


def sem():
    sf = m.semanticFeature(flat[1:4])
    return sf, m.CostRegNet._semantic_vs(sf).contiguous()


def both_serial():
    return psm(), sem()


def stage_a(overlap):
    m.overlap_semantic_branch(overlap)
    return m.forward_2d(x, join=True)


res = {}
for name, fn in (("psm alone", psm), ("semantic alone", sem), ("both, one stream", both_serial),
                 ("forward_2d one stream", lambda: stage_a(False)), ("forward_2d two streams", lambda: stage_a(True))):
    g, keep = graphed(fn)
    res[name] = [timeit(g, reps) for _ in range(3)]
    print("%-26s %s ms" % (name, " ".join("%.3f" % v for v in res[name])), flush=True)
