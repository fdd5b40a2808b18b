#!/usr/bin/env python
"""Streaming (ESTM) evaluation of one scene directory: the role of the reference's eval_hybrid_seq.py
(test_scannet_seq, :123-258) on top of estdepth_amd.ESTMStream -- read frames, slide the 3-frame window with a
2-window memory, dump float16 .npy depth / confidence maps, report the depth-error suite against the scene's
ground-truth depth.  Needs an MI355X (the model has no CPU path).

    python tools/run_stream.py --scene-dir /data/scannet/scene0707_00 --out /tmp/eval --loadckpt model.ckpt
    python tools/run_stream.py --synthetic 8 --out /tmp/eval          # self-contained demo on a generated scene
"""
import argparse
import json
import os
import sys
import tempfile
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--scene-dir")
    ap.add_argument("--synthetic", type=int, default=0, help="generate a scene of this many frames instead")
    ap.add_argument("--out", required=True)
    ap.add_argument("--loadckpt")
    ap.add_argument("--resnet", type=int, default=50)
    ap.add_argument("--ndepths", type=int, default=64)
    ap.add_argument("--depth_min", type=float, default=0.1)
    ap.add_argument("--depth_max", type=float, default=10.0)
    ap.add_argument("--image-size", type=int, nargs=2, default=(320, 256), metavar=("W", "H"))
    ap.add_argument("--frame-interval", type=int, default=10)
    ap.add_argument("--lwindow", type=int, default=3)
    ap.add_argument("--memory_size", type=int, default=2)
    ap.add_argument("--layout", choices=("scannet", "7scenes"), default="scannet")
    ap.add_argument("--no-feature-cache", action="store_true")
    args = ap.parse_args()

    from estdepth_amd import DepthNetHybrid, synth
    from estdepth_amd.streaming import ESTMStream
    from estdepth_amd.eval_io import SequenceReader, save_window_outputs, write_synthetic_scene
    from estdepth_amd.metrics import RunningErrors

    scene_dir, interval = args.scene_dir, args.frame_interval
    if args.synthetic:
        w, h = args.image_size
        _, poses, _, sample = synth.make_sequence(args.synthetic, h, w, seed=7)
        frames = synth.smooth_images(args.synthetic, h, w, seed=7)
        scene_dir = tempfile.mkdtemp(prefix="estd_scene_")
        imgs = [frames[0, i].permute(1, 2, 0).round().clamp(0, 255).byte().numpy() for i in range(args.synthetic)]
        dmaps = [sample["dmaps"][0, i, 0].numpy() for i in range(args.synthetic)]
        write_synthetic_scene(scene_dir, imgs, dmaps, [poses[0, i].numpy() for i in range(args.synthetic)])
        interval = 1
    if not scene_dir:
        ap.error("--scene-dir or --synthetic is required")

    dev = torch.device("cuda:0")
    model = DepthNetHybrid(ndepths=args.ndepths, depth_min=args.depth_min, depth_max=args.depth_max,
                           resnet=args.resnet, IF_EST_transformer=True)
    if args.loadckpt:
        sd = torch.load(args.loadckpt, map_location="cpu")
        model.load_state_dict(sd.get("model", sd))
    else:
        synth.fill_state_dict(model, seed=2, head_gain=1.0)
    model = model.to(dev).eval()
    model.use_channels_last_2d()
    model.use_hip_psm()

    reader = SequenceReader(scene_dir, image_size=tuple(args.image_size), depth_min=args.depth_min,
                            depth_max=args.depth_max, frame_interval=interval,
                            scannet_layout=args.layout == "scannet")
    stream = ESTMStream(model, lwindow=args.lwindow, memory_size=args.memory_size,
                        cache_features=not args.no_feature_cache)
    errs, times, window, resized = RunningErrors(), [], [], 0
    for idx in range(len(reader)):
        s = reader[idx]
        window.append(s)
        window = window[-args.lwindow:]
        torch.cuda.synchronize()
        t0 = time.time()
        res = stream.push(s["img"].to(dev), s["cam_pose"].to(dev), s["cam_intr"].to(dev),
                          s["dmap"].to(dev), s["dmask"].to(dev))
        torch.cuda.synchronize()
        if res is None:
            continue
        times.append(time.time() - t0)
        outputs = res[0]
        target = window[args.lwindow // 2]                                   # eval_hybrid_seq.py:197
        save_window_outputs(outputs, args.out, target["img_path"])
        pred = outputs[("depth", 0, 0)][0, 0].cpu().numpy().astype(np.float64)
        gt = target["dmap"][0, 0].numpy().astype(np.float64)
        if gt.shape != pred.shape:
            # the ground-truth depth stays at native resolution (general_eval_seq.py:191) while the network runs at
            # --image-size: bring the PREDICTION to the ground-truth grid (nearest neighbour on pixel centres, no new
            # depth values are invented) instead of silently skipping the frame
            ys = np.minimum(((np.arange(gt.shape[0]) + 0.5) * pred.shape[0] / gt.shape[0]).astype(np.int64), pred.shape[0] - 1)
            xs = np.minimum(((np.arange(gt.shape[1]) + 0.5) * pred.shape[1] / gt.shape[1]).astype(np.int64), pred.shape[1] - 1)
            pred = pred[ys][:, xs]
            resized += 1
        errs.add(pred, gt)
    report = {"scene": scene_dir, "frames": len(reader), "windows": stream.windows,
              "mean_window_ms": 1e3 * float(np.mean(times[1:] or times or [0.0])), "errors": errs.mean(),
              "predictions_resized_to_gt_grid": resized}
    os.makedirs(args.out, exist_ok=True)
    with open(os.path.join(args.out, "metrics.json"), "w") as f:
        json.dump(report, f, indent=1)
    print(json.dumps(report))


if __name__ == "__main__":
    main()
