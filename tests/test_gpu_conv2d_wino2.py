"""GPU: estd_conv2d_k3_wino2 (csrc/conv2d_wino2.hip: 3x3 / stride 1 / dilation 1 | 2 convolution on NHWC maps with both image axes in
Winograd F(2,3) form + folded BN / ReLU / residual) against an fp64 evaluation of the same fp32 data, against the direct and the
row-only Winograd kernels, through both bindings, on ragged maps, batches, several channel chunks and channel groups."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = "cuda"


def _run(plan, algo, x, res=None):
    from estdepth_amd import ops
    old = ops.CONV2D_ALGO
    ops.CONV2D_ALGO = algo
    try:
        out = plan.run(x, residual=res)
        torch.cuda.synchronize()
        return out
    finally:
        ops.CONV2D_ALGO = old


CASES = [(32, 32, (2, 13, 21), 1), (64, 64, (1, 24, 32), 1), (96, 64, (1, 8, 16), 1), (320, 128, (1, 9, 20), 1), (32, 32, (5, 30, 40), 1),
         (128, 128, (3, 7, 50), 1), (32, 64, (1, 1, 1), 1), (64, 32, (1, 120, 160), 1),
         (128, 128, (1, 17, 35), 2), (32, 64, (2, 8, 16), 2), (64, 64, (1, 3, 5), 2), (128, 128, (2, 60, 80), 2), (32, 32, (1, 1, 1), 2),
         # more work items than the 512 persistent workgroups: every workgroup walks SEVERAL items (the next item's first chunk is requested,
         # transformed and written inside the step loop of the current item's last chunk; 1 / 2 / 4 chunks per item, 1 / 2 / 4 channel groups)
         (32, 32, (3, 240, 320), 1), (64, 64, (4, 120, 160), 1), (128, 64, (2, 120, 160), 1), (128, 128, (2, 120, 160), 2)]


@pytest.mark.parametrize("cin,cout,dims,dil", CASES)
@pytest.mark.parametrize("mode", ["relu", "plain+res", "relu_after_res"])
def test_conv2d_wino2_vs_fp64_and_the_other_kernels(cin, cout, dims, dil, mode):
    from estdepth_amd import synth, ops
    from estdepth_amd.backbones import conv_bn2d
    N, H, W = dims
    mod = conv_bn2d(cin, cout, 3, 1, dil, dil).eval()
    synth.fill_state_dict(mod, seed=cin + cout)
    g = torch.Generator().manual_seed(cin * 3 + cout + H)
    x = torch.randn(N, cin, H, W, generator=g)
    r = torch.randn(N, cout, H, W, generator=g) if mode != "relu" else None
    with torch.no_grad():
        ref = mod.double()(x.double())
    if mode == "relu":
        ref = torch.relu(ref)
    elif mode == "plain+res":
        ref = ref + r.double()
    else:
        ref = torch.relu(ref + r.double())
    mod = mod.float().to(DEV)
    plan = ops.Conv2dPlan(mod[0], mod[1], relu_before=(mode == "relu"), relu_after=(mode == "relu_after_res"))
    assert plan.w_wino2 is not None
    xin = x.to(DEV).permute(0, 2, 3, 1).contiguous()
    rin = r.to(DEV).permute(0, 2, 3, 1).contiguous() if r is not None else None
    errs = {}
    from estdepth_amd import _native
    for algo in ("direct", "wino2") + (("wino",) if _native.has_ab() else ()):      # (row-only Winograd: ESTD_BUILD_AB=1 builds)
        out = _run(plan, algo, xin, rin).permute(0, 3, 1, 2).cpu().double()
        assert tuple(out.shape) == (N, cout, H, W)
        errs[algo] = (out - ref).abs().max().item()
    mag = max(1.0, ref.abs().max().item())
    print("cin %d cout %d dims %s dil %d %s: err direct %.3g  wino %s  wino2 %.3g  (|ref| %.3g)" % (cin, cout, dims, dil, mode, errs["direct"], errs.get("wino"), errs["wino2"], mag))
    assert errs["wino2"] <= 3.0 * errs["direct"] + 2e-7 * mag, errs
    assert errs["wino2"] < 2e-5 * mag


def test_conv2d_wino2_both_bindings_bit_identical_and_argument_checks():
    from estdepth_amd import ops, synth, _native
    from estdepth_amd.backbones import conv_bn2d
    mod = conv_bn2d(64, 64, 3, 1, 1, 1).eval()
    synth.fill_state_dict(mod, seed=5)
    mod = mod.to(DEV)
    plan = ops.Conv2dPlan(mod[0], mod[1], relu_before=True)
    x = torch.randn(2, 19, 33, 64, device=DEV)
    outs = []
    old = ops.BINDING
    try:
        for b in ("torch", "ctypes"):
            ops.BINDING = b
            outs.append(_run(plan, "wino2", x))
    finally:
        ops.BINDING = old
    assert torch.equal(outs[0], outs[1])
    d = _native.Conv2dDesc()
    assert _native.lib().estd_conv2d_k3_wino2(d, None) == -1
