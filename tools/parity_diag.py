"""Where do HIP-vs-oracle depth differences at a FULL-SIZE configuration come from?  (run on the GPU box)

    python tools/parity_diag.py --workload cfg5|joint|estm [--no-oracle]

1. camera matrices: device fp64 Gauss-Jordan (camera_algebra="device", estd_cam_*) vs the reference's torch-CPU composition
   (camera_algebra="host", the default; estdepth_amd/camera.py), bit level;
2. plane sweep: voxels whose sample flips across the |norm| > 1 mask because of (1);
3. whole forward: HIP with device matrices, HIP with host matrices, oracle -- max |d depth| and the number of pixels
   beyond 1e-4 per output scale;
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

    # ---- 1. matrices: the reference's torch-CPU composition (camera_algebra="host", the default) vs the fp64 device kernels ----
    k4 = model.scale_cam_intr(intr, 0.25)
    model.camera_algebra = "host"
    host = model.camera_matrices(x_poses, k4, pre_poses)
    model.camera_algebra = "device"
    devm = model.camera_matrices(x_poses, k4, pre_poses)
    report["camera_matrices_device_vs_host"] = {
        name: {"bit_identical": bool(torch.equal(host[name], devm[name])),
               "max_rel_diff": float(((host[name] - devm[name]).abs() / host[name].abs().clamp_min(1e-6)).max())}
        for name in host if host[name] is not None}

    # ---- 2. flips in the plane sweep of target 0 / source 0 ----
    with torch.no_grad():
        feats = model.matchingFeature(model.normalise_images(x_imgs[0]).contiguous(memory_format=torch.channels_last))
    dv = model.depth_cands.view(-1).to(dev)
    src = feats[0].contiguous()
    wg = ops.homo_warping_chw(src, devm["sweep"][0, 0].contiguous(), dv, D)
    wh = ops.homo_warping_chw(src, host["sweep"][0, 0].contiguous(), dv, D)
    dgh = (wg - wh).abs().amax(0)
    report["plane_sweep_device_vs_host_matrices"] = {"voxels": int(dgh.numel()), "voxels_gt_1e-3": int((dgh > 1e-3).sum()),
                                                     "max": float(dgh.max())}
    del wg, wh, dgh

    def run_gpu(mode):
        model.camera_algebra = mode
        with torch.no_grad():
            out, _, _ = model(x_imgs, x_poses, intr, x_sample, pre_costs, list(pre_poses) if pre_poses else None, mode="val")
        torch.cuda.synchronize()
        return {k: v.cpu().numpy() for k, v in out.items()}

    out_dev = run_gpu("device")
    out_inj = run_gpu("host")
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
