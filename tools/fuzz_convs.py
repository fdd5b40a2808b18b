"""Randomised A/B of the Winograd kernels against the direct kernels (same operator, same descriptor): random ragged shapes,
batch sizes, epilogue flags, instances; the 2D case runs the row-only AND the two-axis kernel (the default) and, one time in four, a map with
more work items than persistent workgroups (the multi-item path of the in-loop transform).  python tools/fuzz_convs.py [seconds] [seed]     (GPU; prints the first mismatch and exits 1)"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch
from estdepth_amd import ops

from estdepth_amd import _native
DEV = torch.device("cuda:0")
AB = _native.has_ab()                                         # depth-only / row-only Winograd kernels: ESTD_BUILD_AB=1 builds only
WINO3D = (("wino", "wino2") if AB else ("wino2",)) + ("wino3",)     # "wino3": the 32 -> 32 instances on csrc/conv3d_wino3.hip (ops.W3, the default); "wino2": ops.W3 = False
budget = float(sys.argv[1]) if len(sys.argv) > 1 else 60.0
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 0)
g = torch.Generator().manual_seed(int(rng.integers(1 << 30)))


def rnd(*shape, scale=1.0):
    return (torch.randn(*shape, generator=g) * scale).to(DEV)


def run3d(plan, algo, x, dims, **kw):
    ops.CONV3D_ALGO, ops.W3 = ("wino2", True) if algo == "wino3" else (algo, False)
    ops.W3_EXTRA = ops.W3
    out = kw.pop("out")
    plan.run(x, dims, out=out, **kw)
    torch.cuda.synchronize()
    return out


def case_conv3d():
    N, D, H, W = int(rng.integers(1, 4)), int(rng.integers(1, 12)), int(rng.integers(1, 40)), int(rng.integers(1, 70))
    inst = rng.choice(["plain", "extra", "xout"])
    dims = (N, D, H, W)
    x = rnd(N, D, H, W, 32)
    kw_common = {}
    if inst == "plain":
        w = rnd(32, 32, 3, 3, 3, scale=0.06).cpu()
        act = str(rng.choice(["relu", "none", "tanh"]))
        plan = ops.Conv3dPlan(w, list(range(32)), None, list(range(32)), 2, torch.rand(32, generator=g) + 0.5, torch.randn(32, generator=g) * 0.1,
                              act_a=act, device=DEV)
        mode = rng.choice(["none", "res", "res2", "acc", "stats"])
        if mode == "res":
            kw_common = dict(residual=rnd(N, D, H, W, 32))
        elif mode == "res2":
            kw_common = dict(residual=rnd(N, D, H, W, 32), residual2=rnd(N, D, H, W, 32), out_scale=0.5)
        elif mode == "acc":
            kw_common = dict(accumulate=True, out_scale=float(rng.uniform(0.3, 1.0)))
        outs = {}
        base = rnd(N, D, H, W, 32)
        for algo in ("direct",) + WINO3D:
            kw = dict(kw_common)
            if mode == "stats":
                kw["stats_partials"] = torch.zeros(ops.conv3d_grid(*dims) * 4, device=DEV, dtype=torch.float64)
            outs[algo] = (run3d(plan, algo, x, dims, out=base.clone(), **kw), kw.get("stats_partials"))
        a = outs["direct"][0]
        tol = 4e-5 * max(1.0, float(a.abs().max()))
        ok, worst = True, 0.0
        for alg in WINO3D:
            b = outs[alg][0]
            worst = max(worst, float((a - b).abs().max()))
            ok = ok and float((a - b).abs().max()) < tol
            if mode == "stats" and ok:
                sa = ops.groupnorm_finalize(outs["direct"][1], ops.conv3d_grid(*dims), 16.0 * N * D * H * W)
                sb = ops.groupnorm_finalize(outs[alg][1], ops.conv3d_grid(*dims), 16.0 * N * D * H * W)
                ok = float((sa - sb).abs().max()) < 1e-4 * max(1.0, float(sa.abs().max()))
        return ok, ("conv3d", inst, dims, act, mode, worst)
    e = rnd(N, D, H, W)
    if inst == "extra":
        w = rnd(32, 33, 3, 3, 3, scale=0.06).cpu()
        plan = ops.Conv3dPlan(w, list(range(32)), 32, list(range(32)), 2, torch.rand(32, generator=g) + 0.5, torch.randn(32, generator=g) * 0.1,
                              act_a="tanh", act_b="relu", act_split=16, device=DEV)
        a = run3d(plan, "direct", x, dims, in_extra=e, out=torch.empty_like(x))
        b2 = run3d(plan, "wino2", x, dims, in_extra=e, out=torch.empty_like(x))
        b = run3d(plan, "wino", x, dims, in_extra=e, out=torch.empty_like(x)) if AB else b2
        d2 = float((a - b2).abs().max())
        if not ((d2 == d2) and d2 < 4e-5 * max(1.0, float(a.abs().max()))):
            return False, ("conv3d", "extra/wino2", dims, d2)
        b3 = run3d(plan, "wino3", x, dims, in_extra=e, out=torch.empty_like(x))
        d3 = float((a - b3).abs().max())
        if not ((d3 == d3) and d3 < 4e-5 * max(1.0, float(a.abs().max()))):
            return False, ("conv3d", "extra/wino3", dims, d3)
    else:
        w = rnd(33, 33, 3, 3, 3, scale=0.06).cpu()
        plan = ops.Conv3dPlan(w, list(range(1, 33)), 0, list(range(33)), 3, torch.rand(33, generator=g) + 0.5, torch.randn(33, generator=g) * 0.1,
                              act_a="relu", device=DEV)
        ea, eb = torch.full((N, D, H, W), float("nan"), device=DEV), torch.full((N, D, H, W), float("nan"), device=DEV)
        a = run3d(plan, "direct", x, dims, in_extra=e, out=torch.empty_like(x), out_extra=ea)
        ec = torch.full((N, D, H, W), float("nan"), device=DEV)
        b2 = run3d(plan, "wino2", x, dims, in_extra=e, out=torch.empty_like(x), out_extra=ec)
        if AB:
            b = run3d(plan, "wino", x, dims, in_extra=e, out=torch.empty_like(x), out_extra=eb)
        else:
            b, eb = b2, ec
        a, b, b2 = torch.cat([a, ea[..., None]], -1), torch.cat([b, eb[..., None]], -1), torch.cat([b2, ec[..., None]], -1)
        d2 = float((a - b2).abs().max())
        if not ((d2 == d2) and d2 < 4e-5 * max(1.0, float(a.abs().max()))):
            return False, ("conv3d", "xout/wino2", dims, d2)
    d = float((a - b).abs().max())
    return (d == d) and d < 4e-5 * max(1.0, float(a.abs().max())), ("conv3d", inst, dims, d)


def case_conv2d():
    cin, cout = int(rng.choice([32, 64, 96, 128, 320])), int(rng.choice([32, 64, 128]))
    dil = int(rng.choice([1, 2]))
    N, H, W = int(rng.integers(1, 6)), int(rng.integers(1, 60)), int(rng.integers(1, 90))
    if rng.integers(4) == 0:            # a map with more work items than persistent workgroups: every workgroup walks several items
        H, W = int(rng.integers(100, 250)), int(rng.integers(120, 330))
    conv = torch.nn.Conv2d(cin, cout, 3, 1, dil, dil, bias=False)
    with torch.no_grad():
        conv.weight.copy_(torch.randn(conv.weight.shape, generator=g) * 0.05)
    conv = conv.to(DEV)
    bn = torch.nn.BatchNorm2d(cout).eval()
    with torch.no_grad():
        bn.weight.copy_(torch.rand(cout, generator=g) + 0.5); bn.bias.copy_(torch.randn(cout, generator=g) * 0.1)
        bn.running_mean.copy_(torch.randn(cout, generator=g) * 0.1); bn.running_var.copy_(torch.rand(cout, generator=g) + 0.5)
    bn = bn.to(DEV)
    rb, ra = bool(rng.integers(2)), bool(rng.integers(2))
    plan = ops.Conv2dPlan(conv, bn, relu_before=rb, relu_after=ra)
    x = rnd(N, H, W, cin)
    res = rnd(N, H, W, cout) if rng.integers(2) else None
    outs = {}
    for algo in ("direct",) + tuple(a for a in WINO3D if a != "wino3"):            # two-axis Winograd (the default; + row-only in an ESTD_BUILD_AB=1 build) against the direct kernel
        ops.CONV2D_ALGO = algo
        outs[algo] = plan.run(x, residual=res)
        torch.cuda.synchronize()
    a = outs["direct"]
    d = max(float((a - outs[alg]).abs().max()) for alg in WINO3D if alg != "wino3")
    return (d == d) and d < 4e-5 * max(1.0, float(a.abs().max())), ("conv2d", (N, H, W), cin, cout, dil, rb, ra, res is not None, d)


def case_head():
    """the 16-channel instances against a float64 torch convolution: 16 -> 16 + 1x1x1 head (head only / head + volume, ReLU / none)
    on the direct kernel, 32 -> 16 with bias on the two-axis Winograd kernel's 16-output-channel instance"""
    import torch.nn.functional as F
    N, D, H, W = int(rng.integers(1, 3)), int(rng.integers(1, 10)), int(rng.integers(1, 30)), int(rng.integers(1, 50))
    dims = (N, D, H, W)
    x = rnd(N, D, H, W, 32)
    xc = x.double().cpu().permute(0, 4, 1, 2, 3)                       # [N,32,D,H,W]
    scale, shift = torch.rand(16, generator=g) + 0.5, torch.randn(16, generator=g) * 0.1
    if rng.integers(2):
        w = rnd(16, 16, 3, 3, 3, scale=0.08).cpu()
        lo = int(rng.choice([0, 16]))
        act = str(rng.choice(["relu", "none"]))
        hw, hb = torch.randn(16, generator=g) * 0.3, torch.randn(1, generator=g)
        plan = ops.Conv3dPlan(w, list(range(16)), None, list(range(16)), 1, scale, shift, act_a=act, head_w=hw, head_b=hb, device=DEV)
        ref = F.conv3d(xc[:, lo:lo + 16], w.double(), padding=1) * scale.double().view(1, 16, 1, 1, 1) + shift.double().view(1, 16, 1, 1, 1)
        ref = ref.clamp_min(0) if act == "relu" else ref
        ref_head = (ref * hw.double().view(1, 16, 1, 1, 1)).sum(1) + hb.double()
        head = torch.full((N, D, H, W), float("nan"), device=DEV)
        with_out = bool(rng.integers(2))
        out = torch.full((N, D, H, W, 16), float("nan"), device=DEV) if with_out else None
        ops.CONV3D_ALGO = "wino2"
        plan.run(x[..., lo:], dims, in_stride=32, out=out, out_stride=16 if with_out else None, out_head=head)
        torch.cuda.synchronize()
        d = float((head.double().cpu() - ref_head).abs().max())
        if with_out:
            d = max(d, float((out.double().cpu().permute(0, 4, 1, 2, 3) - ref).abs().max()))
        return (d == d) and d < 2e-5 * max(1.0, float(ref_head.abs().max())), ("head16", dims, lo, act, with_out, d)
    w = rnd(16, 32, 3, 3, 3, scale=0.06).cpu()
    plan = ops.Conv3dPlan(w, list(range(32)), None, list(range(16)), 1, torch.ones(16), shift, act_a="none", device=DEV)
    ref = F.conv3d(xc, w.double(), padding=1) + shift.double().view(1, 16, 1, 1, 1)
    worst = 0.0
    for algo in ("wino2", "direct"):
        ops.CONV3D_ALGO = algo
        out = torch.full((N, D, H, W, 16), float("nan"), device=DEV)
        plan.run(x, dims, out=out, out_stride=16)
        torch.cuda.synchronize()
        worst = max(worst, float((out.double().cpu().permute(0, 4, 1, 2, 3) - ref).abs().max()))
    return (worst == worst) and worst < 2e-5 * max(1.0, float(ref.abs().max())), ("out16", dims, worst)


t0, n = time.time(), 0
while time.time() - t0 < budget:
    pick = int(rng.integers(4))
    ok, info = (case_conv2d if pick == 0 else case_head if pick == 1 else case_conv3d)()
    n += 1
    if not ok:
        print("MISMATCH after %d cases:" % n, info)
        sys.exit(1)
print("fuzz_convs: %d random cases agree (direct vs Winograd kernels, 16-channel instances vs float64)" % n)
