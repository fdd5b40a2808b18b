"""CPU: the product's state dict is key-for-key, shape-for-shape the REFERENCE model's (checkpoints load unchanged).

G13 (tests/golden/g13_state_dict_keys.json) was written by tools/gen_golden.py from ``DepthNetHybrid(...).state_dict()`` of the
reference itself (hybrid_models/model_hybrid.py:15-60), ResNet-18 and ResNet-50, EST transformer on and off: key order, shapes and
dtypes.  The ``semanticFeature.encoder.*`` entries of that fixture come from the torchvision stand-in the generator has to register
(torchvision is absent in the image), i.e. from the product's own trunk; they are therefore ALSO checked against torchvision's
published ResNet layout, enumerated here independently of both (BasicBlock [2,2,2,2] / Bottleneck [3,4,6,3], expansion 4,
``downsample.{0,1}`` on the first block of a stage when stride or width changes, ``fc`` 1000-way).
"""
import json
import os

import pytest
import torch

from estdepth_amd import DepthNetHybrid


def _fixture(golden_dir):
    with open(os.path.join(golden_dir, "g13_state_dict_keys.json")) as f:
        return json.load(f)


@pytest.mark.parametrize("resnet,est", [(18, True), (18, False), (50, True), (50, False)])
def test_state_dict_keys_shapes_and_order_equal_the_reference(golden_dir, resnet, est):
    ref = _fixture(golden_dir)["r%d_est%d" % (resnet, int(est))]
    m = DepthNetHybrid(ndepths=64, depth_min=0.1, depth_max=10.0, resnet=resnet, IF_EST_transformer=est)
    sd = m.state_dict()
    got = [[k, list(v.shape), str(v.dtype).replace("torch.", "")] for k, v in sd.items()]
    want = ref["entries"]
    assert [g[0] for g in got] == [w[0] for w in want]                   # same keys, same ORDER (strict load_state_dict)
    bad = [(g, w) for g, w in zip(got, want) if g != w]
    assert not bad, bad[:5]
    assert sum(p.numel() for p in m.parameters()) == ref["nparams"]
    assert not hasattr(m, "_buffers") or "depth_cands" not in dict(m.named_buffers())      # plain attribute upstream (model_hybrid.py:32-33)


def _bn(prefix, c):
    return [(prefix + ".weight", [c]), (prefix + ".bias", [c]), (prefix + ".running_mean", [c]), (prefix + ".running_var", [c]),
            (prefix + ".num_batches_tracked", [])]


def torchvision_resnet_layout(depth):
    """torchvision.models.resnet{18,50}().state_dict() keys and shapes, written down from the published architecture"""
    basic = depth in (18, 34)
    blocks = {18: [2, 2, 2, 2], 34: [3, 4, 6, 3], 50: [3, 4, 6, 3]}[depth]
    exp = 1 if basic else 4
    out = [("conv1.weight", [64, 3, 7, 7])] + _bn("bn1", 64)
    inpl = 64
    for li, (n, planes) in enumerate(zip(blocks, (64, 128, 256, 512)), start=1):
        for b in range(n):
            stride = 2 if (b == 0 and li > 1) else 1
            p = "layer%d.%d" % (li, b)
            if basic:
                out += [(p + ".conv1.weight", [planes, inpl, 3, 3])] + _bn(p + ".bn1", planes)
                out += [(p + ".conv2.weight", [planes, planes, 3, 3])] + _bn(p + ".bn2", planes)
            else:
                out += [(p + ".conv1.weight", [planes, inpl, 1, 1])] + _bn(p + ".bn1", planes)
                out += [(p + ".conv2.weight", [planes, planes, 3, 3])] + _bn(p + ".bn2", planes)
                out += [(p + ".conv3.weight", [planes * 4, planes, 1, 1])] + _bn(p + ".bn3", planes * 4)
            if b == 0 and (stride != 1 or inpl != planes * exp):
                out += [(p + ".downsample.0.weight", [planes * exp, inpl, 1, 1])] + _bn(p + ".downsample.1", planes * exp)
            inpl = planes * exp
    out += [("fc.weight", [1000, 512 * exp]), ("fc.bias", [1000])]
    return out


@pytest.mark.parametrize("resnet", [18, 50])
def test_semantic_encoder_keys_follow_torchvisions_resnet_layout(golden_dir, resnet):
    m = DepthNetHybrid(ndepths=64, depth_min=0.1, depth_max=10.0, resnet=resnet, IF_EST_transformer=True)
    pre = "semanticFeature.encoder."
    got = [(k[len(pre):], list(v.shape)) for k, v in m.state_dict().items() if k.startswith(pre)]
    want = torchvision_resnet_layout(resnet)
    assert [g[0] for g in got] == [w[0] for w in want]
    assert got == [(k, s) for k, s in want]
    # and the fixture's marked entries are exactly these (the generator's stand-in is this trunk)
    ref = _fixture(golden_dir)["r%d_est1" % resnet]["entries"]
    assert [(k[len(pre):], s) for k, s, _ in ref if k.startswith(pre)] == [(k, s) for k, s in want]


def test_reference_checkpoint_format_loads(tmp_path):
    """train_hybrid.py:138-142: {'epoch', 'model', 'optimizer'}; a state dict with the fixture's keys loads strictly"""
    m = DepthNetHybrid(ndepths=64, depth_min=0.1, depth_max=10.0, resnet=18, IF_EST_transformer=True)
    ck = {"epoch": 3, "model": {k: torch.zeros_like(v) for k, v in m.state_dict().items()}, "optimizer": {}}
    path = os.path.join(tmp_path, "ck.pth")
    torch.save(ck, path)
    missing, unexpected = m.load_state_dict(torch.load(path)["model"], strict=True)
    assert not missing and not unexpected
