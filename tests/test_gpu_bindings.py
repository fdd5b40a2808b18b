"""GPU: the two bindings of the hot-path operators -- PyTorch custom operators (TORCH_LIBRARY ``estdepth_hip``, the default,
csrc/torch_ops.cpp) and the raw ctypes C ABI -- launch the same kernels: bit-identical results, and both translate bad
arguments into RuntimeError (TORCH_CHECK / estd_status) instead of running a fallback."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = "cuda"


@pytest.fixture()
def both_bindings():
    from estdepth_amd import ops
    saved = ops.BINDING

    def run(fn):
        outs = []
        for b in ("torch", "ctypes"):
            ops.BINDING = b
            outs.append(fn())
        ops.BINDING = saved
        return outs
    yield run
    ops.BINDING = saved


def _t(seed, *shape):
    return torch.randn(*shape, generator=torch.Generator().manual_seed(seed)).to(DEV)


def test_operators_are_registered_with_the_dispatcher():
    from estdepth_amd import ops
    T = ops.T()
    for name in ("homo_warp_costvol", "conv3d_k3", "softargmin_up", "warp_attention", "groupnorm_finalize", "gru_blend",
                 "gru_reset_apply", "homo_warping", "warp_volume", "conv2d_k3", "mix1x1", "cam_sweep_proj"):
        assert hasattr(T, name), name
    schema = str(torch.ops.estdepth_hip.warp_attention.default._schema)
    assert "Tensor[] kv_sources" in schema and schema.startswith("estdepth_hip::warp_attention")


def test_plane_sweep_and_softargmin_through_both_bindings(both_bindings):
    from estdepth_amd import ops, synth
    D, H, W = 8, 12, 20
    K = torch.from_numpy(synth.intrinsics(H * 4, W * 4)).clone()
    K[:2] *= 0.25
    K = K.to(DEV)
    p0, p1 = [torch.from_numpy(synth.camera_pose(v)).to(DEV) for v in range(2)]
    dv = torch.linspace(0.5, 5.0, D).to(DEV)
    src, ref = _t(1, H, W, 32), _t(2, H, W, 32)
    lg = _t(3, 2, D, H, W)

    def run():
        proj = ops.cam_sweep_proj(p0, p1, K)
        vol = ops.homo_warp_costvol(src, ref, proj, dv, D)
        dep, prob = ops.softargmin_up(lg, dv, 4)
        return proj.cpu(), vol.cpu(), dep.cpu(), prob.cpu()
    a, b = both_bindings(run)
    for x, y in zip(a, b):
        assert torch.equal(x, y)


def test_conv3d_and_est_fusion_through_both_bindings(both_bindings):
    from estdepth_amd import ops, synth
    from estdepth_amd.layers_op import ConvBN3d
    from estdepth_amd.epipolar_transformer import EpipolarTransformer
    D, H, W = 6, 11, 19
    mod = ConvBN3d(32, 32, 3, 1, 1, "relu").eval()
    synth.fill_state_dict(mod, seed=5)
    plan = mod.to(DEV).plan()
    est = EpipolarTransformer(16, 16, 3).eval()
    synth.fill_state_dict(est, seed=6)
    est = est.to(DEV)
    x = _t(7, 2, D, H, W, 32)
    kv = [_t(10 + j, D, H, W, 32) for j in range(3)]
    K = torch.from_numpy(synth.intrinsics(H * 4, W * 4)).clone()
    K[:2] *= 0.25
    K = K.to(DEV)
    poses = [torch.from_numpy(synth.camera_pose(v)).to(DEV) for v in range(3)]
    dv = torch.linspace(0.5, 4.0, D).to(DEV)

    def run():
        y = torch.empty_like(x)
        plan.run(x, (2, D, H, W), out=y, out_stride=32)
        mats = torch.stack([ops.cam_volume_mats(poses[j], poses[0], K) for j in (1, 2)])
        tgt = kv[0].clone()
        with torch.no_grad():
            est.fuse_kv(tgt, [kv[1], kv[2]], mats, dv, 0.5, float(dv[1] - dv[0]))
        return y.cpu(), tgt.cpu()
    a, b = both_bindings(run)
    for u, v in zip(a, b):
        assert torch.equal(u, v)


def test_torch_check_translation():
    """bad arguments surface as RuntimeError from the operator layer (no silent fallback, no crash)."""
    from estdepth_amd import ops
    T = ops.T()
    good = torch.zeros(1, 4, 4, 4, device=DEV)
    with pytest.raises(RuntimeError, match="float32"):
        T.softargmin_up(good.double(), torch.ones(4, device=DEV), 4)
    with pytest.raises(RuntimeError):                                   # CPU tensor: no kernel registered for the CPU backend
        T.softargmin_up(torch.zeros(1, 4, 4, 4), torch.ones(4), 4)
    with pytest.raises(RuntimeError, match="contiguous"):
        T.softargmin_up(torch.zeros(1, 4, 4, 8, device=DEV)[..., ::2], torch.ones(4, device=DEV), 4)
    kv = torch.zeros(4, 4, 4, 32, device=DEV)
    with pytest.raises(RuntimeError, match="at most"):
        T.warp_attention(kv, [kv] * 17, torch.zeros(17, 30, device=DEV), torch.ones(4, device=DEV), 0.1, 0.1)
    with pytest.raises(RuntimeError, match="another shape"):
        T.warp_attention(kv, [torch.zeros(4, 4, 5, 32, device=DEV)], torch.zeros(1, 30, device=DEV), torch.ones(4, device=DEV), 0.1, 0.1)
    with pytest.raises(RuntimeError, match="estd_status"):                # the C ABI's own validation, translated
        T.conv3d_k3(kv, None, kv, None, None, None, torch.ones(32, device=DEV), torch.zeros(32, device=DEV), [1, 4, 4, 4], 24, 32, 2,
                    0, 0, 0, torch.empty_like(kv), 32, 32, None, None, 1.0, False, None, None, None, None, None, False)


def test_ops_run_on_the_current_stream():
    """kernels are enqueued on at::hip::getCurrentHIPStream(): work issued inside a torch.cuda.stream block is ordered
    with that stream (the result is complete after synchronising it alone)."""
    from estdepth_amd import ops
    s = torch.cuda.Stream()
    lg = _t(3, 1, 16, 30, 40)
    dv = torch.linspace(0.5, 5.0, 16).to(DEV)
    ref = ops.softargmin_up(lg, dv, 4)[0].cpu()
    torch.cuda.synchronize()
    with torch.cuda.stream(s):
        out = ops.softargmin_up(lg, dv, 4)[0]
    s.synchronize()
    assert torch.equal(out.cpu(), ref)
