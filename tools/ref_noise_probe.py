"""How sharp can an end-to-end golden fixture be?  Runs the REFERENCE (imported read-only from /root/reference, build
container only) on the ESTM fixture recipe with 1 and with 8 torch threads and prints the largest depth difference between
the two runs per output scale, for stereo-head gains 10 / 30 / 100, and the same for the low-resolution logit volumes
(captured with forward hooks on stereo_head0/1) at the gain the committed fixtures use.

Result in this container (torch 2.10 CPU, 128x160, D=64, 3 windows): gain 10 -> 5.1e-4 m, gain 30 -> 1.7e-3 m,
gain 100 -> 4.9e-3 m on ("depth", t, 2): the reference's own fp32 summation-order noise is above the 1e-4 bar for any
end-to-end fixture sharper than gain ~3 (8e-5), which is why G7-G9 use gains 1-3 and the sharp checks live at decoder
level (G6, gain 10 on synthetic cost volumes) and at logit level (G11).

    python tools/ref_noise_probe.py"""
import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools")); sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
import gen_golden as G
import fixtures_spec as S
from estdepth_amd import synth
torch.set_grad_enabled(False)
hu, et, hd, mh = G.import_reference()
def run(threads, gain, seed_model, windows=3, logits=None):
    torch.set_num_threads(threads)
    m = mh.DepthNetHybrid(ndepths=64, depth_min=0.1, depth_max=10.0, resnet=18, IF_EST_transformer=True).eval()
    synth.fill_state_dict(m, seed=seed_model, head_gain=gain)
    if logits is not None:
        m.CostRegNet.stereo_head0.register_forward_hook(lambda mod, i, o: logits.append(("init", o.numpy().copy())))
        m.CostRegNet.stereo_head1.register_forward_hook(lambda mod, i, o: logits.append(("fused", o.numpy().copy())))
    imgs, poses, intr, sample = S.e2e_inputs(windows + 2, S.E2E_HI, S.E2E_WI, seed=1007)
    mem_costs, mem_poses, out = [], [], {}
    for w_ in range(windows):
        sl = slice(w_, w_ + 3)
        smp = {k: v[:, sl] for k, v in sample.items()}
        pc = {"keys": [c["keys"][0] for c in mem_costs], "values": [c["values"][0] for c in mem_costs]} if mem_poses else None
        pp = [p[0] for p in mem_poses] if mem_poses else None
        outputs, costs, cposes = m(imgs[:, sl], poses[:, sl], intr, smp, pc, pp, mode="val")
        mem_costs.append(costs); mem_poses.append(cposes)
        if len(mem_costs) > 2: mem_costs.pop(0); mem_poses.pop(0)
        for k, v in outputs.items(): out[(w_,) + k] = v.numpy().copy()
    return out
la, lb = [], []
run(1, 1.0, 2, logits=la); run(8, 1.0, 2, logits=lb)
print("gain 1 logits: 1-vs-8-thread max |d logit| %.2e, logit range +-%.2f" % (
    max(float(np.abs(x[1] - y[1]).max()) for x, y in zip(la, lb)), max(float(np.abs(x[1]).max()) for x in la)), flush=True)
for gain in (10.0, 30.0, 100.0):
    a = run(1, gain, 5); b = run(8, gain, 5)
    worst = {}
    for k in a:
        if k[1] == "depth":
            worst[k[3]] = max(worst.get(k[3], 0), float(np.abs(a[k]-b[k]).max()))
    pr = max(float(a[k].max()) for k in a if k[1] == "fused_prob")
    dstd = float(np.mean([a[k].std() for k in a if k[1] == "depth" and k[3] == 2]))
    print("gain", gain, "1-vs-8-thread max |d depth| per scale", {s: "%.2e" % w for s, w in sorted(worst.items())}, "max fused_prob %.3f" % pr, "depth2 std %.3f" % dstd, flush=True)
