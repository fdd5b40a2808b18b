"""Benchmark of the ESTDepth hot path on MI355X (contract: see the task statement / DESIGN.md §Measurement).

    python bench.py [--gpus N --steps K --warmup W] [--workload joint|estm|cfg1] [--no-cpu-baseline]

A "step" is one ``DepthNetHybrid.forward`` over one synthetic sequence resident in HBM:
  * joint (default, BASELINE.json configs[1]): seq_len=5, 480x640, D=64, ResNet-50, Joint-mode steady state
    (memory carried from the previous call => EST transformer on, 3 depth frames per step);
  * estm  (configs[2]): steady-state ESTM window (3 frames, 2 memory volumes, 1 depth frame per step).
For N > 1 one process per GPU (launched by torch.distributed.run) runs its own sequence (weak scaling);
after every step the ranks all-gather their memory bank {K, V_fused, pose} over RCCL so any
rank could continue any stream; ``value`` = depth frames of all ranks / max-over-ranks time.
Rank 0 prints ONE JSON line.
"""
import argparse
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

PEAK_FP32_MATRIX_TFLOPS = 157.3      # MI355X_MICROARCH.md: v_mfma_f32_16x16x4_f32 dense peak
PEAK_BF16_MATRIX_TFLOPS = 2500.0     # MI355X_MICROARCH.md: dense bf16 MFMA peak; the 3xbf16 split spends 6 bf16 FLOPs per fp32 FLOP
HBM_PEAK_GBS = 8000.0


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--workload", default="joint", choices=["joint", "estm", "cfg1", "stream"])
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-allgather", action="store_true")
    ap.add_argument("--no-graph", action="store_true", help="launch eagerly instead of replaying a captured hipGraph")
    ap.add_argument("--conv3d-arith", default=os.environ.get("ESTD_CONV3D_ARITH", "f32"), choices=["f32", "bf16x3"],
                    help="products of the plain 32->32 3D convolutions: native fp32 MFMA (default) or the exact 3-way bf16 "
                         "operand split with six bf16 MFMAs per product block (fp32-level error, opt-in)")
    ap.add_argument("--no-alt", action="store_true", help="skip the second timed loop with the other convolution arithmetic")
    ap.add_argument("--conv2d-arith", default=os.environ.get("ESTD_CONV2D_ARITH", "f32"), choices=["f32", "bf16x3"],
                    help="same choice for the 3x3 NHWC convolutions of the PSM extractor / 2D decoder (opt-in)")
    return ap.parse_args()


def build_model(workload, device):
    from estdepth_amd import DepthNetHybrid, synth
    if workload == "cfg1":
        m = DepthNetHybrid(ndepths=16, depth_min=0.1, depth_max=10.0, resnet=18, IF_EST_transformer=False)
    else:
        m = DepthNetHybrid(ndepths=64, depth_min=0.1, depth_max=10.0, resnet=50, IF_EST_transformer=True)
    synth.fill_state_dict(m, seed=0, head_gain=1.0)
    m = m.eval().to(device)
    if str(device) != "cpu" and os.environ.get("ESTD_NCHW_2D", "0") != "1":
        m.use_channels_last_2d()                      # NHWC MIOpen kernels for the 2D backbones
        if os.environ.get("ESTD_PSM", "hip") == "hip":
            m.use_hip_psm()                           # PSM 3x3 convs on the MFMA conv2d kernel
        if os.environ.get("ESTD_FUSE_BN", "1") == "1":
            m.fuse_bn_2d()                            # BN(+add)(+ReLU) after the library convs in one NHWC pass
        if os.environ.get("ESTD_OVERLAP", "1") == "1":
            m.overlap_semantic_branch()               # semantic branch on a second stream
    return m


def make_inputs(workload, rank, device):
    from estdepth_amd import synth
    if workload == "cfg1":
        v, hi, wi = 3, 128, 160
    elif workload == "joint":
        v, hi, wi = 8, 480, 640      # two consecutive 5-frame calls with stride seq_len-2 (general_eval.py:52)
    else:
        v, hi, wi = 5, 480, 640      # three sliding windows of 3
    imgs, poses, intr, sample = synth.make_sequence(v, hi, wi, seed=1000 + rank)
    to = lambda t: t.to(device)
    return to(imgs), to(poses), to(intr), {k: to(t) for k, t in sample.items()}


def cpu_baseline(model, workload):
    """Oracle (C port, OpenMP) timed on the host cores on a bounded sample: ONE get_costvolume at the
    workload's size (2 plane sweeps + pre0 + 2x(pre1,pre2): 281.8 GF of the 2496 GF 3D hot path of a
    Joint sequence), scaled by the FLOP ratio, plus the 2D networks timed with torch on the same cores."""
    import numpy as np
    from oracle import ref_model as M, ref_ops as O
    from estdepth_amd import synth
    cores = O.num_threads()
    P = {k: v.detach().cpu().numpy() for k, v in model.state_dict().items() if k.startswith("pre")}
    D = model.ndepths
    H, W = (32, 40) if workload == "cfg1" else (120, 160)
    rng = np.random.RandomState(0)
    feats = [rng.randn(1, 32, H, W).astype(np.float32) for _ in range(3)]
    poses = np.stack([synth.camera_pose(v) for v in range(3)])[None]
    K = synth.intrinsics(H * 4, W * 4).copy()
    K[:2] *= 0.25
    dv = model.depth_cands.view(1, D, 1, 1).numpy()
    t0 = time.time()
    M.get_costvolume(P, feats, poses, K[None], dv, D)
    t_cv = time.time() - t0
    # 2D networks on the same cores (torch CPU): PSM on V images, ResNet + 2D decoder on T images
    cpu_model = build_model(workload, "cpu")
    V, T = (3, 1) if workload != "joint" else (5, 3)
    hi, wi = H * 4, W * 4
    with torch.no_grad():
        x = torch.randn(V, 3, hi, wi)
        t0 = time.time()
        cpu_model.matchingFeature(x)
        sem = cpu_model.semanticFeature(x[:T])
        sv = cpu_model.CostRegNet._semantic_vs(sem)
        cpu_model.CostRegNet._refine(sv, torch.randn(T, D, H, W), sem)
        t_2d = time.time() - t0
    vox = D * H * W
    gf_cv = (2 * 2 * 27 * 32 * 32 * 2 + 2 * 2 * 64 * 32) * vox / 1e9                   # one get_costvolume
    conv = lambda ci, co: 2 * 27 * ci * co * vox / 1e9
    per_target = gf_cv + 4 * conv(32, 32) + conv(33, 33) + 2 * conv(33, 16) + 2 * conv(16, 16)
    est = conv(32, 32) + conv(32, 16) if workload != "cfg1" else 0.0
    gf_3d = T * (per_target + est)
    t_seq = t_cv * gf_3d / gf_cv + t_2d
    return {"value": round(T / t_seq, 4), "unit": "depth frames/s", "cores": cores, "kind": "port",
            "sample": "oracle (C/OpenMP) get_costvolume for 1 target at full size: %.2f s for %.1f GF, scaled x%.2f to the "
                      "3D hot path of one step, + 2D networks on torch-CPU %.2f s" % (t_cv, gf_cv, gf_3d / gf_cv, t_2d)}


def stream_bench(args, device, rank, world):
    """Extra (not the headline): frame-by-frame ESTM streaming at cfg3 size through estdepth_amd.streaming.ESTMStream
    with the per-frame PSM feature cache; one step = one pushed frame = one depth frame."""
    from estdepth_amd import synth
    from estdepth_amd.streaming import ESTMStream
    model = build_model("estm", device)
    n = args.warmup + args.steps + 2
    imgs, poses, intr, _ = synth.make_sequence(n, 480, 640, seed=1003 + rank)
    imgs, poses, intr = imgs.to(device), poses.to(device), intr.to(device)
    st = ESTMStream(model, cache_features=True)
    for f in range(args.warmup + 2):
        st.push(imgs[0, f], poses[0, f], intr[0])
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for f in range(args.warmup + 2, n):
        st.push(imgs[0, f], poses[0, f], intr[0])
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    if rank == 0:
        print(json.dumps({"metric": "depth frames/sec (ESTM streaming, 480x640, D=64, cached matching features)",
                          "value": round(args.steps * world / dt, 3), "unit": "depth frames/s", "n_gpus": world, "steps": args.steps,
                          "warmup": args.warmup, "ms_per_step": round(1e3 * dt / args.steps, 3), "higher_is_better": True,
                          "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
                          "config": {"workload": "ESTM stream: 1 new frame per step, window 3, memory 2, eager launches"}}), flush=True)


def main():
    args = parse()
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    if not torch.cuda.is_available():
        raise RuntimeError("bench.py needs a ROCm GPU (the product path has no CPU fallback)")
    if "ESTD_FORCE_DEVICE" in os.environ:          # code-path smoke test of N > 1 on a single-GPU box (with gloo)
        local_rank = int(os.environ["ESTD_FORCE_DEVICE"])
    torch.cuda.set_device(local_rank)
    device = torch.device("cuda", local_rank)
    dist = None
    if world > 1:
        import torch.distributed as dist
        # The only collective is the per-step memory-bank all-gather (157 MB per rank, ~37 GB/s of ingress at N = 8): a few RCCL
        # channels carry it inside one step, and every channel is a workgroup taken from the convolutions it overlaps with.
        os.environ.setdefault("NCCL_MAX_NCHANNELS", "4")
        backend = os.environ.get("ESTD_DIST_BACKEND", "nccl")      # "nccl" is RCCL on ROCm
        if backend == "nccl":
            dist.init_process_group(backend="nccl", device_id=device)
        else:
            dist.init_process_group(backend=backend)
    torch.backends.cudnn.benchmark = True
    torch.backends.cudnn.allow_tf32 = False
    torch.backends.cuda.matmul.allow_tf32 = False

    from estdepth_amd import ops, parallel
    ops.CONV3D_ARITH = args.conv3d_arith
    ops.CONV2D_ARITH = args.conv2d_arith
    if args.workload == "stream":
        return stream_bench(args, device, rank, world)
    model = build_model(args.workload, device)
    imgs, poses, intr, sample = make_inputs(args.workload, rank, device)
    sub = lambda sl: {k: v[:, sl] for k, v in sample.items()}

    # ---- establish the steady state (untimed): memory bank from the preceding call(s) ----
    with torch.no_grad():
        if args.workload == "joint":
            _, pre_costs, pre_poses = model(imgs[:, 0:5], poses[:, 0:5], intr, sub(slice(0, 5)), None, None, mode="val")
            sl, frames = slice(3, 8), 3
        elif args.workload == "estm":
            _, c0, p0 = model(imgs[:, 0:3], poses[:, 0:3], intr, sub(slice(0, 3)), None, None, mode="val")
            _, c1, p1 = model(imgs[:, 1:4], poses[:, 1:4], intr, sub(slice(1, 4)),
                              {"keys": [c0["keys"][0]], "values": [c0["values"][0]]}, [p0[0]], mode="val")
            pre_costs = {"keys": [c0["keys"][0], c1["keys"][0]], "values": [c0["values"][0], c1["values"][0]]}
            pre_poses = [p0[0], p1[0]]
            sl, frames = slice(2, 5), 1
        else:
            pre_costs, pre_poses, sl, frames = None, None, slice(0, 3), 1
    x_imgs, x_poses, x_sample = imgs[:, sl].contiguous(), poses[:, sl].contiguous(), sub(sl)

    from estdepth_amd.graph import GraphedForward
    fwd = model if args.no_graph else GraphedForward(model)     # hipGraph replay of the same forward (same kernels)

    state = {"pending": None, "fwd": fwd, "allgather": world > 1 and not args.no_allgather, "notes": []}

    def drain():
        if state["pending"] is not None:
            state["pending"].wait()
            state["pending"] = None

    def step(f=None):
        f = state["fwd"] if f is None else f
        with torch.no_grad():
            try:
                out, costs, cposes = f(x_imgs, x_poses, intr, x_sample, pre_costs, list(pre_poses) if pre_poses else None, mode="val")
            except Exception as e:                       # graph capture refused on this stack: keep measuring, eagerly
                if f is model:
                    raise
                state["notes"].append("hipGraph capture failed (%s): eager launches" % type(e).__name__)
                state["fwd"] = model
                torch.cuda.synchronize()
                out, costs, cposes = model(x_imgs, x_poses, intr, x_sample, pre_costs, list(pre_poses) if pre_poses else None, mode="val")
            if state["allgather"]:
                # memory bank {K, V_fused, pose} of this window -> every rank; the collective of step k overlaps step k+1
                try:
                    drain()
                    state["pending"] = parallel.allgather_memory_bank_async(costs, cposes)
                except Exception as e:                   # same failure on every rank (collective): keep the shards running
                    state["notes"].append("memory-bank all-gather failed (%s: %s): disabled" % (type(e).__name__, str(e)[:80]))
                    state["allgather"], state["pending"] = False, None
        return out

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        step()
    barrier()
    ops.profile_mark(0)                    # estd_mark_kernel brackets the timed region in a rocprofv3 trace
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    drain()                                # the last window's all-gather is inside the timed region
    ops.profile_mark(1)
    barrier()
    elapsed = time.perf_counter() - t0
    # Roofline of the dominant kernel: the same steps once more, launched eagerly, with a HIP-event pair around
    # every launch of that kernel on its launch stream (events cannot bracket nodes inside a graph replay).
    ops.PROFILE = []
    state["allgather"] = False
    for _ in range(args.steps):
        step(model)
    barrier()
    prof, ops.PROFILE = ops.PROFILE, None
    # Second opinion, reported beside (never instead of) the headline: the same K steps with the 3x3x3 / 3x3 convolutions
    # on the exact 3-way bf16 operand split (fp32-level error, tests/test_gpu_split_conv.py), N = 1 only.
    alt = None
    if world == 1 and not args.no_alt and (args.conv3d_arith, args.conv2d_arith) == ("f32", "f32"):
        try:
            ops.CONV3D_ARITH = ops.CONV2D_ARITH = "bf16x3"
            state["fwd"] = model if args.no_graph else GraphedForward(model)
            for _ in range(args.warmup):
                step()
            barrier()
            ta = time.perf_counter()
            for _ in range(args.steps):
                step()
            barrier()
            alt_elapsed = time.perf_counter() - ta
            # how far apart are the two arithmetics on this very input?  (eager forwards of both; all depth outputs, metres)
            with torch.no_grad():
                o_alt = model(x_imgs, x_poses, intr, x_sample, pre_costs, list(pre_poses) if pre_poses else None, mode="val")[0]
                ops.CONV3D_ARITH, ops.CONV2D_ARITH = args.conv3d_arith, args.conv2d_arith
                o_ref = model(x_imgs, x_poses, intr, x_sample, pre_costs, list(pre_poses) if pre_poses else None, mode="val")[0]
            ddiff = max(float((o_alt[k] - o_ref[k]).abs().max()) for k in o_ref if k[0] == "depth")
            alt = {"conv3d_arith": "bf16x3", "conv2d_arith": "bf16x3", "value": round(frames * args.steps / alt_elapsed, 3),
                   "ms_per_step": round(1e3 * alt_elapsed / args.steps, 3),
                   "max_abs_depth_diff_vs_headline_arith_m": float("%.3g" % ddiff),
                   "note": "opt-in (--conv3d-arith/--conv2d-arith bf16x3): every fp32 product as six bf16 MFMA products of exactly "
                           "3-way-split operands, fp32 accumulation; same 1e-4 parity tests, conv error vs fp64 equal to the fp32 MFMA kernel's"}
        except Exception as e:
            alt = {"error": "%s: %s" % (type(e).__name__, str(e)[:120])}
        finally:
            ops.CONV3D_ARITH, ops.CONV2D_ARITH = args.conv3d_arith, args.conv2d_arith
    if world > 1:
        tt = torch.tensor([elapsed], device=device, dtype=torch.float64)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        elapsed = tt.item()

    if rank == 0:
        value = frames * world * args.steps / elapsed
        tot_ms = sum(s.elapsed_time(e) for (_, s, e) in prof)
        tot_flop = sum(f for (f, _, _) in prof)
        achieved = tot_flop / (tot_ms * 1e-3) / 1e12 if tot_ms > 0 else 0.0
        peak = PEAK_FP32_MATRIX_TFLOPS if args.conv3d_arith == "f32" else PEAK_BF16_MATRIX_TFLOPS / 6.0
        # HBM bytes per launch from the committed rocprofv3 PMC passes (FETCH_SIZE x2 correction + WRITE_SIZE, see
        # profiles/r1_conv3d_pmc.json), scaled to this run's average volumes per launch; null if the file is absent
        traffic = None
        pmc_file = os.path.join(ROOT, "profiles", "r1_conv3d_pmc.json")
        if os.path.exists(pmc_file) and prof and args.workload != "cfg1" and args.conv3d_arith == "f32":
            per_vol = json.load(open(pmc_file))["hbm_bytes_per_volume"]
            vols = tot_flop / (2.0 * 27 * 32 * 32 * 64 * 120 * 160)
            traffic = round(per_vol * vols / len(prof))
        line = {
            "metric": "depth frames/sec (seq_len=5, 480x640, D=64)" if args.workload == "joint" else "depth frames/sec",
            "value": round(value, 3), "unit": "depth frames/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(1e3 * elapsed / args.steps, 3), "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None,
            "dtype": "f32" if args.conv3d_arith == "f32" else "f32 (32->32 conv3d products as six bf16 MFMAs of exactly 3-way-split operands, f32 accumulate)",
            "data": "synthetic",
            "config": {"workload": {"joint": "cfg2: seq_len=5, 480x640, ndepths=64, ResNet-50, Joint mode steady state (carried memory, EST on)",
                                    "estm": "cfg3: ESTM steady-state window (3 frames, memory 2), 480x640, ndepths=64, ResNet-50",
                                    "cfg1": "cfg1: seq_len=3, 128x160, ndepths=16, ResNet-18, EST off"}[args.workload],
                       "depth_frames_per_step": frames, "input_frames_per_step": x_imgs.shape[1],
                       "input_frames_per_s": round(x_imgs.shape[1] * world * args.steps / elapsed, 3),
                       "launch": "eager" if (args.no_graph or state["fwd"] is model) else "hipGraph replay",
                       "conv3d_arith": args.conv3d_arith, "conv2d_arith": args.conv2d_arith,
                       "notes": state["notes"],
                       "parallelism": "1 sequence per GPU" + ("; RCCL all-gather of {K,V,pose} per step, overlapped with the next step" if world > 1 and not args.no_allgather else "")},
            "roofline": {"bound": "mfma",
                         "kernel": "conv3d_k3_kernel<32,2> (3x3x3 conv 32->32, fp32 MFMA 16x16x4)" if args.conv3d_arith == "f32" else
                                   "conv3d_k3_split_kernel (3x3x3 conv 32->32, 6 x bf16 MFMA 16x16x32 per fp32 product block; peak = bf16 dense / 6)",
                         "achieved": round(achieved, 2), "peak": round(peak, 1), "unit": "TFLOP/s",
                         "frac": round(achieved / peak, 4), "traffic": traffic,
                         "traffic_unit": "HBM bytes per launch (rocprofv3 FETCH_SIZE x2 + WRITE_SIZE, profiles/r1_conv3d_pmc.json)",
                         "launches": len(prof), "avg_launch_ms": round(tot_ms / max(len(prof), 1), 4)},
        }
        if alt is not None:
            line["alt_arith"] = alt
        if world == 1 and not args.no_cpu_baseline:
            line["cpu_baseline"] = cpu_baseline(model.cpu(), args.workload)
        print(json.dumps(line), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
