"""Where do HIP-vs-oracle depth differences at a FULL-SIZE configuration come from?  (run on the GPU box)

    python tools/parity_diag.py --workload cfg5|joint|estm [--no-oracle]

1. camera matrices: device fp64 Gauss-Jordan (estd_cam_*) vs the oracle's fp32 LAPACK composition, bit level;
2. plane sweep: voxels whose sample flips across the |norm| > 1 mask because of (1) -- HIP kernel with device matrices vs
   the same kernel with the oracle's matrices uploaded vs the C oracle;
3. whole forward: HIP (device matrices), HIP with the oracle's matrices injected, oracle -- max |d depth| and the number
   of pixels beyond 1e-4 per output scale;
4. per-operator wall time of the oracle step (what bench.py's cpu_baseline is made of).
"""
import argparse
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench as B  # noqa: E402


def host_sweep_proj(ref_pose, src_pose, K):
    from oracle import ref_ops as O
    ref_pose, src_pose, K = [np.asarray(a.detach().cpu().numpy(), np.float32) for a in (ref_pose, src_pose, K)]
    ref_extr, src_extr = O.inv(ref_pose), O.inv(src_pose)
    sp, rp = src_extr.copy(), ref_extr.copy()
    sp[:3, :4] = K @ src_extr[:3, :4]
    rp[:3, :4] = K @ ref_extr[:3, :4]
    proj = (sp @ O.inv(rp)).astype(np.float32)
    return np.concatenate([proj[:3, :3].reshape(-1), proj[:3, 3]]).astype(np.float32)


def host_volume_mats(pose_j, pose_i, K):
    from oracle import ref_ops as O
    pose_j, pose_i, K = [np.asarray(a.detach().cpu().numpy(), np.float32) for a in (pose_j, pose_i, K)]
    rel = (pose_j @ O.inv(pose_i)).astype(np.float32)
    m = O.inv(rel)
    return np.concatenate([O.inv(K).reshape(-1), m.reshape(-1)[:12], K.reshape(-1)]).astype(np.float32)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--workload", default="cfg5", choices=["cfg5", "joint", "estm", "cfg1"])
    ap.add_argument("--no-oracle", action="store_true")
    ap.add_argument("--threads", type=int, default=0)
    args = ap.parse_args()
    dev = torch.device("cuda", 0)
    torch.backends.cudnn.allow_tf32 = False
    torch.backends.cuda.matmul.allow_tf32 = False
    from estdepth_amd import ops
    model = B.build_model(args.workload, dev)
    imgs, poses, intr, sample = B.make_inputs(args.workload, 0, dev)
    sl, frames, pre_costs, pre_poses = B.steady_state(model, args.workload, imgs, poses, intr, sample)
    x_imgs, x_poses = imgs[:, sl].contiguous(), poses[:, sl].contiguous()
    x_sample = {k: v[:, sl] for k, v in sample.items()}
    D = B.WORKLOADS[args.workload][3]
    report = {"workload": args.workload}

    # ---- 1. matrices ----
    k4 = intr[0].clone()
    k4[:2] *= 0.25
    p = x_poses[0]
    worst_rel, same = 0.0, True
    for t in range(p.shape[0] - 2):
        for s in (t, t + 2):
            g = ops.cam_sweep_proj(p[t + 1].contiguous(), p[s].contiguous(), k4.contiguous()).cpu().numpy()
            h = host_sweep_proj(p[t + 1], p[s], k4)
            same &= bool(np.array_equal(g, h))
            worst_rel = max(worst_rel, float(np.max(np.abs(g - h) / np.maximum(np.abs(h), 1e-6))))
    report["sweep_proj"] = {"bit_identical": same, "max_rel_diff": worst_rel}

    # ---- 2. flips in the plane sweep of target 0 / source 0 ----
    with torch.no_grad():
        feats = model.matchingFeature(model.normalise_images(x_imgs[0]).contiguous(memory_format=torch.channels_last))
    dv = model.depth_cands.view(-1).to(dev)
    g12 = ops.cam_sweep_proj(p[1].contiguous(), p[0].contiguous(), k4.contiguous())
    h12 = torch.from_numpy(host_sweep_proj(p[1], p[0], k4)).to(dev)
    src = feats[0].contiguous()
    wg = ops.homo_warping_chw(src, g12, dv, D)
    wh = ops.homo_warping_chw(src, h12, dv, D)
    dgh = (wg - wh).abs().amax(0)
    report["plane_sweep_device_vs_host_matrices"] = {"voxels": int(dgh.numel()), "voxels_gt_1e-3": int((dgh > 1e-3).sum()),
                                                     "max": float(dgh.max())}
    del wg, wh, dgh

    def run_gpu():
        with torch.no_grad():
            out, _, _ = model(x_imgs, x_poses, intr, x_sample, pre_costs, list(pre_poses) if pre_poses else None, mode="val")
        torch.cuda.synchronize()
        return {k: v.cpu().numpy() for k, v in out.items()}

    out_dev = run_gpu()
    # ---- inject the oracle's matrices ----
    orig = (ops.cam_sweep_proj, ops.cam_volume_mats)

    def inj_sweep(ref_pose, src_pose, K):
        return torch.from_numpy(host_sweep_proj(ref_pose, src_pose, K)).to(ref_pose.device)

    def inj_vol(pose_j, pose_i, K, out=None):
        m = torch.from_numpy(host_volume_mats(pose_j, pose_i, K)).to(pose_j.device)
        if out is not None:
            out.copy_(m)
            return out
        return m

    ops.cam_sweep_proj, ops.cam_volume_mats = inj_sweep, inj_vol
    out_inj = run_gpu()
    ops.cam_sweep_proj, ops.cam_volume_mats = orig
    dd = {}
    for k in out_dev:
        if k[0] == "depth":
            d = np.abs(out_dev[k] - out_inj[k])
            e = dd.setdefault("scale%d" % k[2], {"max": 0.0, "pixels_gt_1e-4": 0})
            e["max"] = max(e["max"], float(d.max()))
            e["pixels_gt_1e-4"] += int((d > 1e-4).sum())
    report["hip_device_mats_vs_hip_host_mats"] = dd

    if not args.no_oracle:
        from oracle import ref_model as M, ref_ops as O
        from oracle.nets2d import Nets2D, sd_numpy
        n = args.threads or torch.get_num_threads()
        O.set_num_threads(n)
        torch.set_num_threads(n)
        timers = {}

        def timed(mod, name):
            fn = getattr(mod, name)

            def w(*a, **kw):
                t0 = time.time()
                r = fn(*a, **kw)
                timers[name] = timers.get(name, 0.0) + time.time() - t0
                return r
            setattr(mod, name, w)

        for name in ("homo_warping", "warp_volume", "conv3d", "bn_act", "groupnorm1", "epipolar_attention",
                     "depthlayer_upsampled", "sigmoid", "inv"):
            timed(O, name)
        cpu_model = B.build_model(args.workload, "cpu")
        nets = Nets2D(model=cpu_model)
        for name in ("matching", "semantic", "semantic_vs", "refine"):
            timed(nets, name)
        np_ = lambda t: t.detach().float().cpu().contiguous().numpy()
        pc = pp = None
        if pre_costs is not None:
            pc = {"keys": [np_(k) for k in pre_costs["keys"]], "values": [np_(v) for v in pre_costs["values"]]}
            pp = [np_(q) for q in pre_poses]
        t0 = time.time()
        ref, _, _ = M.model_forward(sd_numpy(cpu_model), np_(x_imgs), np_(x_poses), np_(intr), pc, pp, nets, ndepths=D,
                                    depth_min=0.1, depth_max=10.0, IF_EST_transformer=B.WORKLOADS[args.workload][5])
        total = time.time() - t0
        timers["(python glue: concatenate/stack/repeat ...)"] = total - sum(timers.values())
        report["oracle_step_seconds"] = {"total": round(total, 2), "threads": n,
                                         **{k: round(v, 2) for k, v in sorted(timers.items(), key=lambda kv: -kv[1])}}
        for tag, out in (("hip_device_mats_vs_oracle", out_dev), ("hip_host_mats_vs_oracle", out_inj)):
            dd = {}
            for k in out:
                if k[0] == "depth":
                    d = np.abs(out[k] - ref[k])
                    e = dd.setdefault("scale%d" % k[2], {"max": 0.0, "pixels_gt_1e-4": 0, "pixels": 0, "median": 0.0})
                    e["max"] = max(e["max"], float(d.max()))
                    e["pixels_gt_1e-4"] += int((d > 1e-4).sum())
                    e["pixels"] += int(d.size)
                    e["median"] = max(e["median"], float(np.median(d)))
            report[tag] = dd
    print(json.dumps(report, indent=1))


if __name__ == "__main__":
    main()
