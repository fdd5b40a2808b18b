"""Numpy composition of the ESTDepth hot path on top of the C oracle.

ORACLE = test infrastructure only (see oracle/__init__.py).

Restates, with the reference's quirks kept verbatim:
  * DepthNetHybrid.get_costvolume        hybrid_models/model_hybrid.py:62-102
  * EpipolarTransformer.forward          transformer/epipolar_transformer.py:56-83
  * DepthHybridDecoder.forward_*         hybrid_models/hybrid_depth_decoder.py:138-432
  * DepthNetHybrid.forward (mode='val')  hybrid_models/model_hybrid.py:110-184

Weights ``P`` are a dict of float32 numpy arrays keyed by the reference's state-dict names.
The 2D networks (PSM / ResNet / 2D decoder) are not part of the hot path; they are passed in
as callables (``nets``) working on numpy arrays so this package stays independent of torch.
"""
import numpy as np

import contextlib

from . import ref_ops as O


@contextlib.contextmanager
def use_ops(module):
    """Evaluate the composition below with another operator module of the same surface (oracle/torch_ops.py: torch's own CPU
    operators instead of the C restatement -- the "torch-ops" cpu_baseline leg of bench.py)."""
    global O
    old, O = O, module
    try:
        yield
    finally:
        O = old


def _bn(P, key):
    return (P[key + ".weight"], P[key + ".bias"], P[key + ".running_mean"], P[key + ".running_var"])


def convbn3d(P, prefix, x, act):
    """networks/layers_op.py:16-39: Conv3d(bias=False) -> BatchNorm3d -> act; params at prefix.0 / prefix.1"""
    y = O.conv3d(x, P[prefix + ".0.weight"])
    return O.bn_act(y, _bn(P, prefix + ".1"), act)


# ----------------------------------------------------------------------------------------------
def get_costvolume(P, features, cam_poses, cam_intr, depth_values, ndepths):
    """model_hybrid.py:62-102.  features: list of V arrays [B,32,H,W]; cam_poses [B,V,4,4];
    cam_intr [B,3,3] (already scaled to 1/4); depth_values [B,D,1,1]."""
    num_views = len(features)
    ref_feature = features[num_views // 2]                                   # :72
    B = ref_feature.shape[0]
    ref_volume = np.repeat(ref_feature[:, :, None], ndepths, axis=2)        # :76
    cost = np.zeros_like(ref_volume)
    for v in range(num_views):
        if v == num_views // 2:
            continue
        proj = O.sweep_proj(cam_poses, cam_intr, num_views // 2, v)         # :74-88 + homo_utils.py:469 (one torch chain)
        warped = O.homo_warping_proj(features[v], proj, depth_values)       # :90
        x = np.concatenate([ref_volume, warped], 1)                          # :93
        x = convbn3d(P, "pre0", x, "none")                                   # :94
        x = x + convbn3d(P, "pre2", convbn3d(P, "pre1", x, "relu"), "none")  # :95
        cost = cost + x                                                      # :97
    return cost / np.float32(num_views - 1)                                  # :99


# ----------------------------------------------------------------------------------------------
def epipolar_transformer(P, prefix, target_key, target_value, warped_values, warped_keys):
    """transformer/epipolar_transformer.py:56-83 (GRU + attention)."""
    x = target_value
    if warped_values is not None:
        h = O.epipolar_attention(target_key, warped_keys, warped_values)    # :62-73
    else:
        h = np.zeros_like(x)                                                 # :78-79
    f = O.conv3d(np.concatenate([x, h], 1), P[prefix + ".gate_conv.weight"], P[prefix + ".gate_conv.bias"])  # :36-37
    C = f.shape[1]
    r, u = f[:, :C // 2], f[:, C // 2:]                                      # :41-42
    rn = O.groupnorm1(r, P[prefix + ".reset_gate_norm.weight"], P[prefix + ".reset_gate_norm.bias"])    # :44
    un = O.groupnorm1(u, P[prefix + ".update_gate_norm.weight"], P[prefix + ".update_gate_norm.bias"])  # :45
    rns, uns = O.sigmoid(rn), O.sigmoid(un)                                  # :46-47
    o = O.conv3d(np.concatenate([x, rns * h], 1), P[prefix + ".output_conv.weight"], P[prefix + ".output_conv.bias"])  # :51-52
    on = O.groupnorm1(o, P[prefix + ".output_norm.weight"], P[prefix + ".output_norm.bias"])  # :53
    y = np.tanh(on).astype(np.float32)                                       # :82
    return (uns * h + (1 - uns) * y).astype(np.float32)                      # :83


# ----------------------------------------------------------------------------------------------
def _head(P, prefix, x):
    """stereo_head{0,1}: convbnrelu_3d(16,16) then Conv3d(16,1,k=1,bias)  (hybrid_depth_decoder.py:104-112)"""
    y = convbn3d(P, prefix + ".0", x, "relu")
    return O.conv3d(y, P[prefix + ".1.weight"], P[prefix + ".1.bias"])[:, 0]   # squeeze(1)


def decoder_forward(P, costvolumes, semantic_features, cam_poses, cam_intr, depth_values,
                    depth_min, depth_interval, pre_costs, pre_cam_poses, mode, nets,
                    IF_EST_transformer=True, prefix="CostRegNet"):
    """DepthHybridDecoder.forward (hybrid_depth_decoder.py:419-432) and both branches
    (:138-292 transformer, :294-417 no transformer).  ``cam_poses`` is a python list of [B,4,4]
    and IS MUTATED in the transformer branch exactly like the reference (:221, Q7).
    Returns (outputs, {"keys":[..],"values":[..]}, [pose]).  Also returns the low-res logits in
    outputs under ("init_logits",) / ("fused_logits",) for debugging."""
    num = len(costvolumes)
    B, C, D, H, W = costvolumes[0].shape
    flag = IF_EST_transformer and (pre_costs is not None or mode == "train")   # :423
    outputs = {}

    semantic_vs = nets.semantic_vs(semantic_features)                        # :162-184 (2D decoder)
    cv = np.stack(costvolumes, 1).reshape(B * num, C, D, H, W)               # :187-188
    m = convbn3d(P, prefix + ".dres0.1", convbn3d(P, prefix + ".dres0.0", cv, "relu"), "relu")   # :190
    m = convbn3d(P, prefix + ".dres1.1", convbn3d(P, prefix + ".dres1.0", m, "relu"), "relu")    # :191
    x = np.concatenate([semantic_vs[:, None], m], 1)                         # :195  (semantic = channel 0)
    x = convbn3d(P, prefix + ".dres2.0", x, "relu")                          # :196
    value = convbn3d(P, prefix + ".value_layer.0", x, "tanh")                # :198
    key = convbn3d(P, prefix + ".key_layer.0", x, "relu")                    # :199
    init_logits_ = _head(P, prefix + ".stereo_head0", value)                 # :200  [B*num,D,H,W]
    d3, p3 = O.depthlayer_upsampled(init_logits_, np.repeat(depth_values, num, 0), 4)   # :202-204
    d3 = d3.reshape(B, num, 1, 4 * H, 4 * W)
    p3 = p3.reshape(B, num, 1, 4 * H, 4 * W)
    for i in range(num):
        outputs[("depth", i, 3)] = d3[:, i]
        outputs[("init_prob", i)] = p3[:, i]

    value = value.reshape(B, num, 16, D, H, W)
    key = key.reshape(B, num, 16, D, H, W)
    values = [value[:, i] for i in range(num)]
    keys = [key[:, i] for i in range(num)]
    det_values = list(values)
    det_keys = list(keys)

    if flag:
        if pre_costs is not None:                                            # :220-224 (in-place list extend)
            cam_poses += pre_cam_poses
            values += list(pre_costs["values"])
            keys += list(pre_costs["keys"])
            pre_num = len(pre_cam_poses)
        else:
            pre_num = 0
        depth_lowres = np.broadcast_to(depth_values.reshape(B, 1, D, 1), (B, 1, D, H * W))
        all_fused = []
        for i in range(num):                                                 # :229  sequential (Q9)
            wk, wv = [], []
            for j in range(num + pre_num):
                if i == j:
                    continue
                rel = np.stack([O.matmul(cam_poses[j][b], O.inv(cam_poses[i][b])) for b in range(B)])   # :235 (Q8)
                wk.append(O.warp_volume(keys[j], depth_lowres, rel, cam_intr, None, depth_min, depth_interval))    # :237
                wv.append(O.warp_volume(values[j], depth_lowres, rel, cam_intr, None, depth_min, depth_interval))  # :241
            fused = epipolar_transformer(P, prefix + ".epipolar_transformer", keys[i], values[i], wv, wk)   # :248
            values[i] = fused                                                # :253
            det_values[i] = fused
            fl = _head(P, prefix + ".stereo_head1", fused)                   # :256
            all_fused.append(fl)
            d2, p2 = O.depthlayer_upsampled(fl, depth_values, 4)             # :259-260
            outputs[("depth", i, 2)], outputs[("fused_prob", i)] = d2, p2
        all_fused_logits = np.stack(all_fused, 1).reshape(B * num, D, H, W)  # :264-265
    else:
        all_fused_logits = _head(P, prefix + ".stereo_head1", value.reshape(B * num, 16, D, H, W))   # :377
        d2, p2 = O.depthlayer_upsampled(all_fused_logits, np.repeat(depth_values, num, 0), 4)        # :379-381
        d2 = d2.reshape(B, num, 1, 4 * H, 4 * W)
        p2 = p2.reshape(B, num, 1, 4 * H, 4 * W)
        for i in range(num):
            outputs[("depth", i, 2)] = d2[:, i]
            outputs[("fused_prob", i)] = p2[:, i]

    outputs[("init_logits",)] = init_logits_
    outputs[("fused_logits",)] = all_fused_logits
    # depth refinement :268-290 / :392-415 (2D decoder)
    s1, s0 = nets.refine(semantic_vs, all_fused_logits, semantic_features)
    s1 = s1.reshape(B, num, 1, 4 * H, 4 * W)
    s0 = s0.reshape(B, num, 1, 4 * H, 4 * W)
    for i in range(num):
        outputs[("depth", i, 1)] = s1[:, i]
        outputs[("depth", i, 0)] = s0[:, i]
    return outputs, {"keys": det_keys[-1:], "values": det_values[-1:]}, cam_poses[-1:]   # :292 / :417


# ----------------------------------------------------------------------------------------------
def model_forward(P, imgs, cam_poses, cam_intr, pre_costs, pre_cam_poses, nets,
                  ndepths=64, depth_min=0.01, depth_max=10.0, IF_EST_transformer=True, mode="val"):
    """DepthNetHybrid.forward, inference modes (model_hybrid.py:110-184)."""
    imgs = np.asarray(imgs, np.float32)
    cam_poses = np.asarray(cam_poses, np.float32)
    cam_intr = np.asarray(cam_intr, np.float32)
    depth_interval = (depth_max - depth_min) / (ndepths - 1)                 # :29
    depth_cands = (np.arange(ndepths, dtype=np.float32) * np.float32(depth_interval)
                   + np.float32(depth_min)).astype(np.float32)               # :32-33 (fp32 arithmetic)
    imgs = (2 * (imgs / np.float32(255.)) - 1.).astype(np.float32)           # :119
    B, V, _, Hi, Wi = imgs.shape
    H, W = Hi // 4, Wi // 4
    assert V > 2                                                             # :123
    T = V - 2
    mf = nets.matching(imgs.reshape(B * V, 3, Hi, Wi)).reshape(B, V, -1, H, W)   # :128-129
    feats = [np.ascontiguousarray(mf[:, v]) for v in range(V)]               # :130
    sem = nets.semantic(np.ascontiguousarray(imgs[:, 1:1 + T]).reshape(B * T, 3, Hi, Wi))   # :138-139
    k4 = cam_intr.copy()
    k4[:, :2, :] *= np.float32(0.25)                                         # :142 / :104-108
    depth_values = np.broadcast_to(depth_cands.reshape(1, ndepths, 1, 1), (B, ndepths, 1, 1)).copy()   # :144
    cvs, tposes = [], []
    for t in range(T):                                                       # :152-164
        cvs.append(get_costvolume(P, feats[t:t + 3], cam_poses[:, t:t + 3], k4, depth_values, ndepths))
        tposes.append(cam_poses[:, t + 1])
    return decoder_forward(P, cvs, sem, tposes, k4, depth_values, depth_min, depth_interval,
                           pre_costs, pre_cam_poses, mode, nets, IF_EST_transformer)
