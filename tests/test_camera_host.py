"""CPU: the host-side camera algebra of the product (estdepth_amd/camera.py) and the rounding model the kernels and the
C oracle rely on.

  * camera.py composes the matrices with the reference's own torch-CPU calls; the oracle does the same through
    ref_ops.inv / ref_ops.matmul -> bit-identical matrices on both sides of every parity test;
  * ATen's GEMM kernels behind torch.matmul / torch.bmm accumulate k in order with fused multiply-adds: the property that
    csrc/plane_sweep.hip::sweep_coords, csrc/est_fusion.hip::volume_coords_base and oracle/estd_oracle.c spell with
    explicit fmaf() so that the discontinuous |norm| > 1 masks see the reference's coordinates bit for bit."""
import numpy as np
import torch

from estdepth_amd import camera, synth
from oracle import ref_ops as O


def _fma(a, b, c):
    return (a.astype(np.float64) * b.astype(np.float64) + c.astype(np.float64)).astype(np.float32)


def test_gemm_accumulates_k_in_order_with_fma():
    g = torch.Generator().manual_seed(0)
    H, W = 120, 160
    rot = (torch.randn(1, 3, 3, generator=g) * 0.01 + torch.eye(3)).float()
    y, x = torch.meshgrid([torch.arange(0, H, dtype=torch.float32), torch.arange(0, W, dtype=torch.float32)], indexing="ij")
    xyz = torch.stack((x.reshape(-1), y.reshape(-1), torch.ones(H * W)))[None]
    ref = torch.matmul(rot, xyz)[0].numpy()                                   # homo_utils.py:479
    R, X = rot[0].numpy(), xyz[0].numpy()
    for i in range(3):
        a = [np.full(H * W, R[i, k], np.float32) for k in range(3)]
        chain = _fma(a[2], X[2], _fma(a[1], X[1], a[0] * X[0]))
        assert np.array_equal(chain, ref[i])
    M = (torch.randn(1, 4, 4, generator=g) * 0.05 + torch.eye(4)).float()
    c = torch.randn(1, 4, 4096, generator=g) * 3
    c[:, 3] = 1
    ref = torch.bmm(M, c)[0].numpy()                                          # homo_utils.py:35
    Mn, C = M[0].numpy(), c[0].numpy()
    for i in range(4):
        a = [np.full(4096, Mn[i, k], np.float32) for k in range(4)]
        chain = _fma(a[3], C[3], _fma(a[2], C[2], _fma(a[1], C[1], a[0] * C[0])))
        assert np.array_equal(chain, ref[i])


def _poses(n):
    return torch.from_numpy(np.stack([synth.camera_pose(v) for v in range(n)]))[None]


def test_sweep_projections_equal_the_oracle_composition_bitwise():
    poses = _poses(5)
    K = torch.from_numpy(synth.intrinsics(480, 640))[None].clone()
    K[:, :2] *= 0.25
    got = camera.sweep_projections(poses, K, "cpu").numpy()
    for t in range(3):
        for k, s in enumerate((t, t + 2)):
            proj = O.sweep_proj(poses.numpy(), K.numpy(), t + 1, s)[0]
            assert np.array_equal(got[t, k, :9], proj[:3, :3].reshape(-1)) and np.array_equal(got[t, k, 9:], proj[:3, 3])
    one = camera.sweep_projection_set(poses[:, 1:4], K, 1, (0, 2), "cpu").numpy()         # get_costvolume() of views 1..3
    assert np.array_equal(one, got[1])


def test_volume_matrices_equal_the_oracle_composition_bitwise():
    poses = _poses(5)
    K = torch.from_numpy(synth.intrinsics(480, 640))[None].clone()
    K[:, :2] *= 0.25
    plist = [poses[:, v] for v in range(5)]
    got = camera.volume_matrices(plist, 3, K, "cpu").numpy()
    assert got.shape == (3, 4, 30)
    for i in range(3):
        for r, j in enumerate([j for j in range(5) if j != i]):
            rel = O.matmul(plist[j][0].numpy(), O.inv(plist[i][0].numpy()))
            assert np.array_equal(got[i, r, :9], O.inv(K[0].numpy()).reshape(-1))
            assert np.array_equal(got[i, r, 9:21], O.inv(rel)[:3].reshape(-1))
            assert np.array_equal(got[i, r, 21:], K[0].numpy().reshape(-1))
    one = camera.relative_volume_matrix(torch.from_numpy(O.matmul(plist[2][0].numpy(), O.inv(plist[0][0].numpy()))), K[0], "cpu").numpy()
    assert np.array_equal(one, got[0, 1])


def test_cpp_operator_equals_the_python_statement_bitwise():
    """estdepth_hip::camera_matrices_host (one C++ call per forward) == camera.sweep_projections / volume_matrices."""
    poses = _poses(7)
    K = torch.from_numpy(synth.intrinsics(480, 640))[None].clone()
    K[:, :2] *= 0.25
    pre = [poses[:, 5], poses[:, 6]]
    got = camera.forward_matrices(poses[:, :5], K, pre, True, "cpu")
    assert torch.equal(got["sweep"], camera.sweep_projections(poses[:, :5], K, "cpu"))
    plist = [poses[:, t + 1] for t in range(3)] + pre
    assert torch.equal(got["vol"], camera.volume_matrices(plist, 3, K, "cpu"))
    nov = camera.forward_matrices(poses[:, :3], K, None, False, "cpu")
    assert nov["vol"] is None and torch.equal(nov["sweep"], camera.sweep_projections(poses[:, :3], K, "cpu"))
