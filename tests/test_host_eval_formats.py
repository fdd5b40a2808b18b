"""Host-side pieces around the streaming path: depth-error suite vs the reference's own numbers
(G10), on-disk readers/writers, and the window/memory bookkeeping of ESTMStream with a stub model (CPU only)."""
import os

import numpy as np
import pytest
import torch

import fixtures_spec as S
from estdepth_amd import eval_io, metrics
from estdepth_amd.streaming import ESTMStream

GOLD = os.path.join(os.path.dirname(__file__), "golden", "g10_metrics.npz")


@pytest.mark.parametrize("case", [c[0] for c in S.g10_cases()])
def test_metrics_match_reference(case):
    g = np.load(GOLD)
    name, pred, gt = [c for c in S.g10_cases() if c[0] == case][0]
    with np.errstate(all="ignore"):
        e = metrics.compute_errors(pred.copy(), gt.copy())
    keys = [str(k) for k in g[case + "|keys"]]
    assert sorted(e) == keys
    np.testing.assert_allclose([float(e[k]) for k in keys], g[case + "|vals"], rtol=1e-12, atol=0, equal_nan=True)
    m = metrics.compute_valid_depth_mask(pred, gt)
    assert np.array_equal(m, g[case + "|mask"])
    if m.sum():
        p, q = pred[m], gt[m]
        sc = [metrics.compute_depth_scale_factor(p, q, s) for s in ("abs", "log", "inv")]
        np.testing.assert_allclose(sc, g[case + "|scale"], rtol=1e-12)
        e0, e1 = metrics.evaluate_depth(np.array([0.3, 0.1, 0.2]), gt.copy(), pred.copy(), inverse_gt=False,
                                        inverse_pred=False)
        np.testing.assert_allclose([float(e0[k]) for k in keys], g[case + "|eval0"], rtol=1e-12, equal_nan=True)
        np.testing.assert_allclose([float(e1[k]) for k in keys], g[case + "|eval1"], rtol=1e-12, equal_nan=True)
    else:
        assert e["num_valid"] == 0 and np.isnan(e["abs_relative"])


def test_metrics_reject_unprocessed_maps_and_unknown_scaling():
    with pytest.raises(AssertionError):
        metrics.l1(np.array([1.0, -1.0]), np.array([1.0, 1.0]))
    with pytest.raises(Exception, match="Unknown depth scaling"):
        metrics.compute_depth_scale_factor(np.ones(3), np.ones(3), "median")


def _scene(tmp_path, n=5, h=48, w=64):
    g = np.random.default_rng(0)
    imgs = g.integers(0, 256, size=(n, h, w, 3), dtype=np.uint8)
    dm = g.uniform(0.0, 6.0, size=(n, h, w))
    poses = [np.eye(4) + 0.01 * i for i in range(n)]
    poses[3] = np.full((4, 4), np.nan)                              # dropped by check_pose
    scene = str(tmp_path / "scene0")
    eval_io.write_synthetic_scene(scene, imgs, dm, poses)
    return scene, imgs, dm, poses


def test_sequence_reader_sample_dictionary(tmp_path):
    scene, imgs, dm, poses = _scene(tmp_path)
    rd = eval_io.SequenceReader(scene, image_size=(64, 48), depth_min=0.5, depth_max=5.0, frame_interval=1)
    assert len(rd) == 4                                             # NaN pose frame skipped
    assert [os.path.basename(r["img_path"]) for r in rd.seqs] == ["0.png", "10.png", "20.png", "40.png"]   # natural order
    s = rd[1]
    assert s["img"].shape == (1, 3, 48, 64) and s["img"].dtype == torch.float32
    assert torch.equal(s["img"][0].permute(1, 2, 0), torch.from_numpy(imgs[1]).float())     # no resize -> exact
    assert s["img_raw"].shape == (1, 48, 64, 3)
    mm = np.rint(dm[1] * 1000.) / 1000.
    valid = (mm >= 0.5) & (mm <= 5.0)
    assert np.array_equal(s["dmask"][0, 0].numpy(), valid)
    np.testing.assert_allclose(s["dmap"][0, 0].numpy(), np.where(valid, mm, 0).astype(np.float32))
    np.testing.assert_allclose(s["cam_pose"][0].numpy(), poses[1].astype(np.float32))
    k = s["cam_intr"][0].numpy()
    np.testing.assert_allclose(k, [[57.787, 0, 31.95], [0, 57.787, 23.95], [0, 0, 1]], rtol=1e-6)


def test_resize_matches_half_pixel_bilinear(tmp_path):
    g = np.random.default_rng(1)
    img = g.integers(0, 256, size=(96, 128, 3), dtype=np.uint8)
    out = eval_io.resize_bilinear_u8(img, 64, 48)
    ref = torch.nn.functional.interpolate(torch.from_numpy(img).permute(2, 0, 1)[None].double(), size=(48, 64),
                                          mode="bilinear", align_corners=False)[0].permute(1, 2, 0)
    assert out.shape == (48, 64, 3)
    assert np.abs(out.astype(np.int64) - ref.round().numpy().astype(np.int64)).max() <= 1    # .5 rounding ties only
    assert eval_io.resize_bilinear_u8(img, 128, 96) is img


def test_save_window_outputs_layout(tmp_path):
    t = lambda v: torch.full((1, 1, 8, 10), v)
    outputs = {("depth", 0, 2): t(1.5), ("depth", 0, 0): t(2.5), ("init_prob", 0): t(0.25), ("fused_prob", 0): t(0.75)}
    paths = eval_io.save_window_outputs(outputs, str(tmp_path / "out"), "/data/scene/rgb/120.jpg")
    assert sorted(paths) == ["init_depth", "init_prob", "refined_depth", "refined_prob"]
    for kind, val, shape in (("init_depth", 1.5, (1, 8, 10)), ("refined_depth", 2.5, (1, 8, 10)),
                             ("init_prob", 0.25, (8, 10)), ("refined_prob", 0.75, (8, 10))):
        a = np.load(str(tmp_path / "out" / kind / "120.npy"))
        assert a.dtype == np.float16 and a.shape == shape and np.all(a == val)


class _StubModel:
    """records what ESTMStream hands to forward(); returns a fresh 'memory' per window."""

    def __init__(self):
        self.calls = []

    def __call__(self, imgs, poses, intr, sample, pre_costs, pre_poses, mode="val", matching_features=None):
        n = len(self.calls)
        self.calls.append({"imgs": imgs.clone(), "poses": poses.clone(), "n_mem": 0 if pre_costs is None else len(pre_costs["keys"]),
                           "mem_ids": [] if pre_costs is None else [int(k[0, 0]) for k in pre_costs["keys"]],
                           "dmaps": sample["dmaps"].shape, "feats": matching_features})
        costs = {"keys": [torch.full((1, 1), float(n))], "values": [torch.full((1, 1), float(n))]}
        return {("depth", 0, 0): torch.zeros(1, 1, 4, 4)}, costs, [poses[:, 1]]


def test_stream_window_and_memory_protocol():
    m = _StubModel()
    st = ESTMStream(m, lwindow=3, memory_size=2, cache_features=False)
    K = torch.eye(3)
    res = []
    for i in range(7):
        res.append(st.push(torch.full((3, 4, 4), float(i)), torch.eye(4) * (i + 1), K))
    assert [r is None for r in res] == [True, True] + [False] * 5
    assert st.windows == 5 and len(m.calls) == 5
    for w, c in enumerate(m.calls):
        assert c["imgs"].shape == (1, 3, 3, 4, 4) and c["dmaps"] == (1, 3, 1, 4, 4)
        assert [float(c["imgs"][0, v, 0, 0, 0]) for v in range(3)] == [w, w + 1, w + 2]       # slides by one frame
        assert c["feats"] is None
    assert [c["n_mem"] for c in m.calls] == [0, 1, 2, 2, 2]                                   # eval_hybrid_seq.py:191-193
    assert m.calls[4]["mem_ids"] == [2, 3]                                                    # oldest window evicted
    st.reset()
    assert st.push(torch.zeros(3, 4, 4), torch.eye(4), K) is None and st.windows == 0
    with pytest.raises(RuntimeError, match="at least 3 frames"):
        ESTMStream(m, lwindow=2)
