"""GraphedForward(zero_copy_memory=True): the memory record a call returns lies in a ring of output buffers the graph writes
directly, and records handed in are read where they lie (estdepth_amd/graph.py).  Same results as the eager forward under the
reference's two protocols -- eval_hybrid_seq.py:160-193 (sliding windows, the last ``memory_size`` records carried) and the
Joint carry of general_eval.py:52 (one record) -- and a record that is passed back stays what it was.
"""
import pytest
import torch

import fixtures_spec as S

pytestmark = pytest.mark.gpu

DEV = "cuda"


@pytest.fixture(scope="module", autouse=True)
def _need_gpu():
    if not torch.cuda.is_available():
        pytest.fail("GPU tests need a ROCm device")
    from estdepth_amd import _native
    _native.lib()
    torch.backends.cudnn.allow_tf32 = False
    torch.backends.cuda.matmul.allow_tf32 = False
    yield


def _model():
    from estdepth_amd import synth, DepthNetHybrid
    m = DepthNetHybrid(ndepths=64, depth_min=0.1, depth_max=10.0, resnet=18, IF_EST_transformer=True).eval()
    synth.fill_state_dict(m, seed=2, head_gain=1.0)
    return m.to(DEV)


def _inputs(n):
    imgs, poses, intr, sample = S.e2e_inputs(n, S.E2E_HI, S.E2E_WI, seed=1003)
    smp = lambda sl: {k: v[:, sl].to(DEV) for k, v in sample.items()}
    return imgs.to(DEV), poses.to(DEV), intr.to(DEV), smp


def _pc(mem):
    if not mem:
        return None, None
    return ({"keys": [c["keys"][0] for c, _ in mem], "values": [c["values"][0] for c, _ in mem]}, [p[0] for _, p in mem])


@pytest.mark.parametrize("memory_size", [1, 2])
def test_zero_copy_memory_matches_eager_with_carried_records(memory_size):
    from estdepth_amd.graph import GraphedForward
    m = _model()
    gf = GraphedForward(m, zero_copy_memory=True)
    n_win = 7
    imgs, poses, intr, smp = _inputs(n_win + 2)
    mem_e, mem_g, kept = [], [], []
    with torch.no_grad():
        for w in range(n_win):
            sl = slice(w, w + 3)
            pce, ppe = _pc(mem_e)
            e, ec, ep = m(imgs[:, sl], poses[:, sl], intr, smp(sl), pce, ppe, mode="val")
            e = {k: v.clone() for k, v in e.items()}
            pcg, ppg = _pc(mem_g)
            g, gc, gp = gf(imgs[:, sl], poses[:, sl], intr, smp(sl), pcg, ppg, mode="val")
            for k in e:
                assert (e[k] - g[k]).abs().max().item() < 2e-5, (w, k)
            assert torch.equal(ep[0], gp[0])
            assert (ec["values"][0] - gc["values"][0]).abs().max().item() < 2e-5, w
            assert (ec["keys"][0] - gc["keys"][0]).abs().max().item() < 2e-5, w
            # the records the caller still holds (the ones it passes back) are what they were when they were returned
            for rec, snap in kept:
                assert torch.equal(rec["values"][0], snap[0]) and torch.equal(rec["keys"][0], snap[1]), w
            mem_e = (mem_e + [(ec, ep)])[-memory_size:]
            mem_g = (mem_g + [(gc, gp)])[-memory_size:]
            kept = (kept + [(gc, (gc["values"][0].clone(), gc["keys"][0].clone()))])[-memory_size:]
    torch.cuda.synchronize()
    ring = next(iter(gf._ring.values()))
    assert len(ring["bufs"]) == memory_size + 1                 # inputs + the buffer being written: nothing more is ever allocated
    # records live in the ring: no copy out of the graph, and the next call reads them in place
    ptrs = {b[-1].data_ptr() for b in ring["bufs"]}
    assert all(c["values"][0]._estd_kv.data_ptr() in ptrs for c, _ in mem_g)
    # steady state: one capture per ring position (memory_size + 1), plus the shorter memories of the first windows
    assert len(gf._graphs) <= 2 * (memory_size + 1)


def test_zero_copy_memory_with_fixed_foreign_records():
    """what bench.py does: the same eager-produced record handed in at every call, the returned one not carried"""
    from estdepth_amd.graph import GraphedForward
    m = _model()
    imgs, poses, intr, smp = _inputs(5)
    with torch.no_grad():
        _, c0, p0 = m(imgs[:, 0:3], poses[:, 0:3], intr, smp(slice(0, 3)), None, None, mode="val")
        pc = {"keys": [c0["keys"][0]], "values": [c0["values"][0]]}
        before = c0["values"][0].clone()
        sl = slice(1, 4)
        e, ec, _ = m(imgs[:, sl], poses[:, sl], intr, smp(sl), pc, [p0[0]], mode="val")
        e = {k: v.clone() for k, v in e.items()}
        gf = GraphedForward(m, zero_copy_memory=True)
        prev = None
        for it in range(5):
            g, gc, _ = gf(imgs[:, sl], poses[:, sl], intr, smp(sl), pc, [p0[0]], mode="val")
            for k in e:
                assert (e[k] - g[k]).abs().max().item() < 2e-5, (it, k)
            assert (ec["values"][0] - gc["values"][0]).abs().max().item() < 2e-5
            if prev is not None:                                 # the record of the previous call is still intact (e.g. being sent)
                assert prev[0]._estd_kv.data_ptr() != gc["values"][0]._estd_kv.data_ptr()
                assert torch.equal(prev[0], prev[1])
            prev = (gc["values"][0], gc["values"][0].clone())
        assert torch.equal(c0["values"][0], before)             # read in place, never written
        assert len(gf._graphs) == 2 and len(next(iter(gf._ring.values()))["bufs"]) == 2


def test_zero_copy_memory_falls_back_to_the_copy_path_for_ever_new_records():
    """a caller that hands in a NEW tensor at every call gets MAX_FOREIGN address-keyed captures, then the static-copy capture"""
    from estdepth_amd.graph import GraphedForward
    m = _model()
    imgs, poses, intr, smp = _inputs(5)
    with torch.no_grad():
        _, c0, p0 = m(imgs[:, 0:3], poses[:, 0:3], intr, smp(slice(0, 3)), None, None, mode="val")
        sl = slice(1, 4)
        pc0 = {"keys": [c0["keys"][0]], "values": [c0["values"][0]]}
        e, _, _ = m(imgs[:, sl], poses[:, sl], intr, smp(sl), pc0, [p0[0]], mode="val")
        e = {k: v.clone() for k, v in e.items()}
        gf = GraphedForward(m, zero_copy_memory=True)
        hold = []
        for it in range(8):
            from estdepth_amd.hybrid_depth_decoder import kv_views
            k, v = kv_views(c0["values"][0]._estd_kv.clone())   # a new address every time
            hold.append((k, v))
            g, _, _ = gf(imgs[:, sl], poses[:, sl], intr, smp(sl), {"keys": [k], "values": [v]}, [p0[0]], mode="val")
            for kk in e:
                assert (e[kk] - g[kk]).abs().max().item() < 2e-5, (it, kk)
        keyed = {k[-2][0] for k in gf._graphs if k[-2][0] is not None}
        assert len(keyed) == gf.MAX_FOREIGN
        assert any(k[-2][0] is None for k in gf._graphs)


def test_streaming_harness_with_graph_replay_reproduces_estm_golden(golden_dir):
    """ESTMStream(graph=True) -- hipGraph replay with zero-copy memory (a ring of memory_size + 1 record buffers) and the per-frame PSM
    graph, what bench.py's `stream` workload times -- fed frame by frame must reproduce the reference's streaming run (G8), and the
    memory the harness holds must stay what the windows returned."""
    import os
    import numpy as np
    from estdepth_amd.streaming import ESTMStream
    g = np.load(os.path.join(golden_dir, "g8_estm_stream.npz"))
    m = _model()
    imgs, poses, intr, _ = _inputs(6)
    st = ESTMStream(m, lwindow=3, memory_size=2, cache_features=True, graph=True)
    assert st.model.zero_copy_memory
    w, snaps = 0, []
    for f in range(6):
        r = st.push(imgs[0, f], poses[0, f], intr[0])
        if f < 2:
            assert r is None
            continue
        outputs, costs, cposes = r
        for k, v in outputs.items():
            name = "w%d|" % w + "|".join(map(str, k))
            if name in g.files:
                assert float(np.abs(v.cpu().numpy() - g[name]).max()) < 1e-4, name
        assert np.array_equal(cposes[0].cpu().numpy(), g["w%d|pose" % w])
        for rec, snap in snaps:                                  # the records still in the harness's memory are untouched
            assert torch.equal(rec["values"][0], snap)
        snaps = (snaps + [(costs, costs["values"][0].clone())])[-2:]
        w += 1
    assert w == 4
    assert len(next(iter(st.model._ring.values()))["bufs"]) == 3


@pytest.mark.parametrize("zero_copy", [False, True])
def test_new_weights_force_a_recapture(zero_copy):
    """load_state_dict bumps the model's weights epoch: the next call re-captures (packed weights are baked into a graph) and the
    captures of the old weights are dropped -- in both memory modes, with every ring position"""
    from estdepth_amd import synth, DepthNetHybrid
    from estdepth_amd.graph import GraphedForward
    m = _model()
    other = DepthNetHybrid(ndepths=64, depth_min=0.1, depth_max=10.0, resnet=18, IF_EST_transformer=True).eval()
    synth.fill_state_dict(other, seed=7, head_gain=1.0)
    sd_b = {k: v.clone() for k, v in other.state_dict().items()}
    imgs, poses, intr, smp = _inputs(5)
    sl = slice(1, 4)
    gf = GraphedForward(m, zero_copy_memory=zero_copy)
    with torch.no_grad():
        _, c0, p0 = m(imgs[:, 0:3], poses[:, 0:3], intr, smp(slice(0, 3)), None, None, mode="val")
        pc = {"keys": [c0["keys"][0].clone()], "values": [c0["values"][0].clone()]}      # plain tensors: the memory does not change with the weights
        ea, _, _ = m(imgs[:, sl], poses[:, sl], intr, smp(sl), pc, [p0[0]], mode="val")
        ea = {k: v.clone() for k, v in ea.items()}
        for _ in range(3):
            ga, _, _ = gf(imgs[:, sl], poses[:, sl], intr, smp(sl), pc, [p0[0]], mode="val")
        for k in ea:
            assert (ea[k] - ga[k]).abs().max().item() < 2e-5, k
        old_epoch = m._estd_weights_epoch
        m.load_state_dict(sd_b)
        assert m._estd_weights_epoch != old_epoch
        eb, _, _ = m(imgs[:, sl], poses[:, sl], intr, smp(sl), pc, [p0[0]], mode="val")
        eb = {k: v.clone() for k, v in eb.items()}
        assert max((ea[k] - eb[k]).abs().max().item() for k in ea) > 1e-3          # the two weight sets give different depths
        for _ in range(3):
            gb, _, _ = gf(imgs[:, sl], poses[:, sl], intr, smp(sl), pc, [p0[0]], mode="val")
            for k in eb:
                assert (eb[k] - gb[k]).abs().max().item() < 2e-5, k
    assert all(k[-1] == m._estd_weights_epoch for k in gf._graphs)                  # nothing of the old weights is kept


def test_kernel_switches_flipped_at_run_time_force_a_recapture():
    """A capture bakes the kernel choice in.  The module-level switches tests and tools flip at run time (ops.W3: three-axis vs two-axis 3D
    convolution; epipolar_transformer.GATE_IN_CONV: reset gate folded into the output convolution vs its own pass) are part of the capture key
    (round 6, ADVICE): a live GraphedForward re-captures instead of silently replaying the old kernels -- an A/B through one wrapper compares
    two different graphs -- and each graph keeps replaying its own choice."""
    from estdepth_amd import ops
    from estdepth_amd import epipolar_transformer as ET
    from estdepth_amd.graph import GraphedForward
    m = _model()
    imgs, poses, intr, smp = _inputs(5)
    sl = slice(1, 4)
    old_w3, old_gate = ops.W3, ET.GATE_IN_CONV
    try:
        with torch.no_grad():
            _, c0, p0 = m(imgs[:, 0:3], poses[:, 0:3], intr, smp(slice(0, 3)), None, None, mode="val")
            pc = {"keys": [c0["keys"][0].clone()], "values": [c0["values"][0].clone()]}
            gf = GraphedForward(m, clone_outputs=True)
            args = (imgs[:, sl], poses[:, sl], intr, smp(sl), pc, [p0[0]])
            a, _, _ = gf(*args, mode="val")
            assert len(gf._graphs) == 1
            ops.W3 = not old_w3
            b, _, _ = gf(*args, mode="val")
            assert len(gf._graphs) == 2                      # a second capture, the first one is kept
            ET.GATE_IN_CONV = not old_gate
            c, _, _ = gf(*args, mode="val")
            assert len(gf._graphs) == 3
            ops.W3, ET.GATE_IN_CONV = old_w3, old_gate
            a2, _, _ = gf(*args, mode="val")
            assert len(gf._graphs) == 3                      # back on the first capture
        for k in a:
            assert torch.equal(a[k], a2[k]), k
            assert (a[k] - b[k]).abs().max().item() < 2e-5 and (a[k] - c[k]).abs().max().item() < 2e-5, k
        # different kernels round differently: the two-axis and the three-axis convolution do not agree bit for bit on every output
        assert any(not torch.equal(a[k], b[k]) for k in a)
    finally:
        ops.W3, ET.GATE_IN_CONV = old_w3, old_gate
