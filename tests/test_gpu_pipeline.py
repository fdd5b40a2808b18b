"""GPU: GraphedForward(pipeline=True) -- stage A of call k + 1 beside stage B of call k on two lanes of captures -- returns bit for bit what
the serial replay returns, call after call, on the reference's two protocols with CARRIED memory (the one dependence between consecutive
calls): Joint clips at stride seq_len - 2 (eval_hybrid.py:229-243) and ESTM windows with a memory of two (eval_hybrid_seq.py:160-193)."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = "cuda"


def _model():
    from estdepth_amd import DepthNetHybrid, synth
    m = DepthNetHybrid(ndepths=32, depth_min=0.1, depth_max=10.0, resnet=18, IF_EST_transformer=True).eval()
    synth.fill_state_dict(m, seed=2, head_gain=3.0)
    return m.to(DEV)


def _run(fwd, protocol, imgs, poses, intr, n_calls):
    """the outputs (cloned after join) and the memory checksums of every call"""
    res = []
    mem_c, mem_p = None, None
    keep = []
    for c in range(n_calls):
        if protocol == "joint":
            sl = slice(3 * c, 3 * c + 5)
        else:
            sl = slice(c, c + 3)
        sample = {"dmaps": torch.ones(1, sl.stop - sl.start, 1, 128, 160, device=DEV), "dmasks": torch.ones(1, sl.stop - sl.start, 1, 128, 160, dtype=torch.bool, device=DEV)}
        with torch.no_grad():
            out, costs, cposes = fwd(imgs[:, sl].contiguous(), poses[:, sl].contiguous(), intr, sample, mem_c, mem_p, mode="val")
        if protocol == "joint":
            mem_c, mem_p = costs, cposes
        else:                                   # the last two windows' records (memory_size = 2)
            keep = (keep + [(costs, cposes)])[-2:]
            mem_c = {"keys": [k["keys"][0] for k, _ in keep], "values": [k["values"][0] for k, _ in keep]}
            mem_p = [p[0] for _, p in keep]
        fwd.join(fwd.last_event)                # (this call's stage B alone; a no-op for the serial replay)
        res.append(({k: v.clone() for k, v in out.items()}, costs["values"][0].double().sum().item(), costs["keys"][0].double().sum().item(), cposes[0].clone()))
    torch.cuda.synchronize()
    return res


@pytest.mark.parametrize("protocol", ["joint", "estm"])
@pytest.mark.parametrize("zero_copy", [True, False])
def test_pipelined_replay_equals_serial_replay(protocol, zero_copy):
    from estdepth_amd import synth
    from estdepth_amd.graph import GraphedForward
    torch.backends.cudnn.allow_tf32 = False
    n_calls = 6
    frames = 5 + 3 * (n_calls - 1) if protocol == "joint" else n_calls + 2
    imgs, poses, intr, _ = synth.make_sequence(frames, 128, 160, seed=77)
    imgs, poses, intr = imgs.to(DEV), poses.to(DEV), intr.to(DEV)
    model = _model()
    serial = _run(GraphedForward(model, zero_copy_memory=zero_copy), protocol, imgs, poses, intr, n_calls)
    torch.cuda.synchronize()
    piped = _run(GraphedForward(model, zero_copy_memory=zero_copy, pipeline=True), protocol, imgs, poses, intr, n_calls)
    for c, (a, b) in enumerate(zip(serial, piped)):
        assert set(a[0]) == set(b[0])
        for k in a[0]:
            assert torch.equal(a[0][k], b[0][k]), (protocol, c, k, float((a[0][k] - b[0][k]).abs().max()))
        assert a[1] == b[1] and a[2] == b[2] and torch.equal(a[3], b[3]), (protocol, c)


def test_pipelined_calls_without_join_in_between():
    """the bench's pattern: K calls back to back (same arguments, nobody consumes a result in between), ONE join at the end: the last two
    calls' outputs (one per lane) equal the serial replay's."""
    from estdepth_amd import synth
    from estdepth_amd.graph import GraphedForward
    imgs, poses, intr, _ = synth.make_sequence(8, 128, 160, seed=78)
    imgs, poses, intr = imgs.to(DEV), poses.to(DEV), intr.to(DEV)
    model = _model()
    sample = lambda n: {"dmaps": torch.ones(1, n, 1, 128, 160, device=DEV), "dmasks": torch.ones(1, n, 1, 128, 160, dtype=torch.bool, device=DEV)}
    with torch.no_grad():
        _, c0, p0 = model(imgs[:, 0:5], poses[:, 0:5], intr, sample(5), None, None, mode="val")
        x = (imgs[:, 3:8].contiguous(), poses[:, 3:8].contiguous(), intr, sample(5), c0, list(p0))
        ref, _, _ = GraphedForward(model, zero_copy_memory=True, clone_outputs=True)(*x, mode="val")
        torch.cuda.synchronize()
        fwd = GraphedForward(model, zero_copy_memory=True, pipeline=True)
        outs = [fwd(*x, mode="val")[0] for _ in range(9)]
        fwd.join()
        torch.cuda.synchronize()
    for out in outs[-2:]:
        for k in ref:
            assert torch.equal(out[k], ref[k]), (k, float((out[k] - ref[k]).abs().max()))
    assert len({id(outs[-1][k]) for k in ref} & {id(outs[-2][k]) for k in ref}) == 0          # two lanes: two sets of output buffers
