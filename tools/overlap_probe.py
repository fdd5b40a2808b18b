"""Probe: is there throughput in overlapping DIFFERENT steps on one GPU?  Two independent hipGraph replays of the Joint step (own static
buffers, own output rings) on two streams, enqueued alternately -- stage A of one meets stage B of the other at a drifting phase -- against
the same number of steps on one stream.  If two interleaved streams of steps are not faster than one, pipelining stage A of call k + 1
under stage B of call k (the only dependence between consecutive calls is the memory record stage B hands on) has nothing to win either.
    python tools/overlap_probe.py [steps]"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch

import bench as B
from estdepth_amd.graph import GraphedForward

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 40
dev = torch.device("cuda", 0)
torch.backends.cudnn.allow_tf32 = False
model = B.build_model("joint", dev)
imgs, poses, intr, sample = B.make_inputs("joint", 0, dev)
sl, frames, pre_costs, pre_poses = B.steady_state(model, "joint", imgs, poses, intr, sample)
x_imgs, x_poses = imgs[:, sl].contiguous(), poses[:, sl].contiguous()
x_sample = {k: v[:, sl] for k, v in sample.items()}
fwds = [GraphedForward(model, zero_copy_memory=True) for _ in range(2)]
streams = [torch.cuda.Stream(), torch.cuda.Stream()]


def call(i):
    with torch.cuda.stream(streams[i]), torch.no_grad():
        return fwds[i](x_imgs, x_poses, intr, x_sample, pre_costs, list(pre_poses), mode="val")


for i in (0, 1):
    for _ in range(4):
        call(i)
    torch.cuda.synchronize()
ref = {k: v.clone() for k, v in call(0)[0].items()}
torch.cuda.synchronize()


def timed(pattern, n):
    for i in pattern:
        call(i)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        for i in pattern:
            call(i)
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / (n * len(pattern))


for rep in range(2):
    one = timed([0], steps)
    two = timed([0, 1], steps // 2)
    print("rep %d: one stream of steps %.3f ms/step (%.1f depth frames/s); two interleaved streams %.3f ms/step (%.1f)   x%.3f"
          % (rep, 1e3 * one, 3 / one, 1e3 * two, 3 / two, one / two))
out = call(1)[0]
torch.cuda.synchronize()
print("outputs of the second replay equal the first's: max |diff| = %.3g" % max(float((out[k] - ref[k]).abs().max()) for k in ref))
