"""Benchmark of the ESTDepth hot path on MI355X (contract: see the task statement / DESIGN.md §5 Measurement).

    python bench.py [--gpus N --steps K --warmup W] [--workload joint|estm|cfg5|cfg1|stream] [--no-cpu-baseline]

A "step" is one ``DepthNetHybrid.forward`` over one synthetic sequence resident in HBM:
  * joint (default, BASELINE.json configs[1]): seq_len=5, 480x640, D=64, ResNet-50, Joint-mode steady state
    (memory carried from the previous call => EST transformer on, 3 depth frames per step);
  * estm  (configs[2]): steady-state ESTM window (3 frames, 2 memory volumes, 1 depth frame per step);
  * cfg5  (configs[4]): the same ESTM steady-state window at 960x1280, D=128 (HBM-bound stress);
  * cfg1  (configs[0]): seq_len=3, 128x160, D=16, ResNet-18, EST off;
  * stream: frame-by-frame ESTM streaming with the per-frame matching-feature cache (extra, not a BASELINE config).
``--gpus N`` with N > 1 (BASELINE.json configs[3]): if the process was not started by a launcher (no WORLD_SIZE in the
environment) bench.py re-executes ITSELF under ``python -m torch.distributed.run --nproc-per-node N`` -- one rank per
GPU over RCCL -- so ``python bench.py --gpus 8`` and the explicit torchrun command line are the same job.  Every rank
runs its own sequence (weak scaling, no data-path collective); after every step the ranks all-gather their memory bank
{K, V_fused, pose} over RCCL, overlapped with the next step; ``value`` = depth frames of all ranks /
max-over-ranks time.  Rank 0 prints ONE JSON line.
"""
import argparse
import json
import os
import socket
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

PEAK_FP32_MATRIX_TFLOPS = 157.3      # MI355X_MICROARCH.md: v_mfma_f32_16x16x4_f32 dense peak
PEAK_BF16_MATRIX_TFLOPS = 2500.0     # MI355X_MICROARCH.md: dense bf16 MFMA peak; the 3xbf16 split spends 6 bf16 FLOPs per fp32 FLOP
HBM_PEAK_GBS = 8000.0                # MI355X_MICROARCH.md: HBM3E spec peak (about 6.3 TB/s is achievable by a copy)
# fraction of the direct convolution's 27-tap MFMA products a kernel actually issues (the rest is removed by exact Winograd identities)
EXECUTED_FACTOR = {"direct": 1.0, "wino": 18.0 / 27.0, "wino2": 12.0 / 27.0, "wino3": 8.0 / 27.0,
                   "taps": 32.0 * 36.0 / (27.0 * 33.0)}       # conv3d_xout: 32 tap rows x 36 channel steps issued for 27 x 33 useful products

#            name:  (views, Hi,  Wi,   D,  resnet, EST,   description)
WORKLOADS = {
    "cfg1": (3, 128, 160, 16, 18, False, "cfg1: seq_len=3, 128x160, ndepths=16, ResNet-18, EST off"),
    "joint": (8, 480, 640, 64, 50, True, "cfg2: seq_len=5, 480x640, ndepths=64, ResNet-50, Joint mode steady state (carried memory, EST on)"),
    "estm": (5, 480, 640, 64, 50, True, "cfg3: ESTM steady-state window (3 frames, memory 2), 480x640, ndepths=64, ResNet-50"),
    "cfg5": (5, 960, 1280, 128, 50, True, "cfg5: ESTM steady-state window (3 frames, memory 2), 960x1280, ndepths=128, ResNet-50"),
}


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--workload", default="joint", choices=["joint", "estm", "cfg5", "cfg1", "stream"])
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-threads", type=int, default=0,
                    help="host threads of the cpu_baseline leg (0 = torch's default = the physical cores of the box; SMT siblings slow both "
                         "the OpenMP oracle and the oneDNN convolutions of the 2D networks down)")
    ap.add_argument("--cpu-leg-child", default=None, help=argparse.SUPPRESS)          # internal: the pinned child of cpu_leg_pinned()
    ap.add_argument("--cpu-leg-kind", default="torch-ops", help=argparse.SUPPRESS)
    ap.add_argument("--no-allgather", action="store_true")
    ap.add_argument("--no-graph", action="store_true", help="launch eagerly instead of replaying a captured hipGraph")
    ap.add_argument("--graph-memory", default=os.environ.get("ESTD_GRAPH_MEMORY", "zero-copy"), choices=["zero-copy", "copy"],
                    help="hipGraph replay: zero-copy = memory records are read where they lie and returned in a ring of output buffers "
                         "(GraphedForward(zero_copy_memory=True), default); copy = static input buffers + a fresh clone per call "
                         "(two 157 MB device copies per Joint step)")
    ap.add_argument("--conv3d-arith", default=os.environ.get("ESTD_CONV3D_ARITH", "f32"), choices=["f32", "bf16x3"],
                    help="products of the plain 32->32 3D convolutions: native fp32 MFMA (default) or the exact 3-way bf16 "
                         "operand split with six bf16 MFMAs per product block (fp32-level error; ESTD_BUILD_AB=1 builds only)")
    ap.add_argument("--conv3d-algo", default=os.environ.get("ESTD_CONV3D_ALGO", "wino2"), choices=["wino2", "wino", "direct"],
                    help="3D convolutions under f32 arithmetic: wino2 = depth and row axis of the plain 32->32 instance in Winograd F(2,3) "
                         "form (0.444 of the fp32 MFMA products, csrc/conv3d_wino2.hip: every 32/33-channel instance, 32->16 and the 16->16 heads; default), "
                         "wino = depth axis only (2/3 of the products, csrc/conv3d_wino.hip; ESTD_BUILD_AB=1 builds only), direct = 27-tap implicit GEMM (csrc/conv3d_mfma.hip)")
    ap.add_argument("--no-alt", action="store_true", help="skip the second timed loop with the other convolution arithmetic")
    ap.add_argument("--no-replay-profile", action="store_true",
                    help="skip the rocprofv3 kernel trace of a short child run (per-kernel durations INSIDE the hipGraph replay)")
    ap.add_argument("--no-other-workloads", action="store_true",
                    help="skip the short timed loops of the other single-GPU workloads (ESTM window, cfg5, stream) reported beside the headline")
    ap.add_argument("--pipeline", default=os.environ.get("ESTD_PIPELINE", "off"), choices=["on", "off"],
                    help="hipGraph replay: on = stage A (2D networks) of step k + 1 beside the SECOND half of stage B of step k (GraphedForward(pipeline=True): "
                         "two lanes of captures, stage B cut into two graphs behind the key||value convolution; the only dependence between consecutive calls is "
                         "the memory record stage B hands on).  Bit-identical results (tests/test_gpu_pipeline.py) but NOT a robust gain: +0.2 .. +1.2 %% on the "
                         "Joint step in 40-step A/B pairs on four boxes, -2 .. -7 %% in the default 10-step command on a fifth, 200-step buckets between 15.7 and "
                         "16.3 ms where the serial replay holds 15.5 +- 0.05 (profiles/r6_pipeline_ab.txt): what the next stage A meets is left to the hardware "
                         "queues.  Default off; a line with it on carries the serial replay of the same process (config.serial_replay)")
    ap.add_argument("--sustained-s", type=float, default=float(os.environ.get("ESTD_SUSTAINED_S", "20")),
                    help="after the K timed steps: the same step for about this many seconds in buckets of --sustained-bucket steps, shader clock and "
                         "board power sampled beside it by a host thread (config.sustained, config.sustained_ms_per_step, "
                         "config.sustained_over_timed); 0 = skip.  `value` stays the K-step figure of the contract")
    ap.add_argument("--sustained-bucket", type=int, default=200)
    ap.add_argument("--conv2d-arith", default=os.environ.get("ESTD_CONV2D_ARITH", "f32"), choices=["f32", "bf16x3"],
                    help="same choice for the 3x3 NHWC convolutions of the PSM extractor / 2D decoder (opt-in)")
    return ap.parse_args()


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def self_launch_if_needed(args):
    """``python bench.py --gpus N`` (N > 1) without a launcher: become N ranks (one per GPU) under torch.distributed.run.
    On a box with fewer than N GPUs (the 1-GPU test box) the ranks share the devices round-robin and use gloo instead of
    RCCL (RCCL refuses two ranks on one device): a code-path check, flagged as such in the JSON line."""
    if args.gpus <= 1 or "WORLD_SIZE" in os.environ:
        return
    import torch
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    ndev = torch.cuda.device_count()
    if ndev < args.gpus:
        env["ESTD_OVERSUBSCRIBED"] = str(max(ndev, 1))
        env.setdefault("ESTD_DIST_BACKEND", "gloo")
        # several processes on ONE GPU oversubscribe its hardware queues (4 per process by default + gloo's streams): the queue
        # scheduler then time-slices them with millisecond quanta (measured: 5 ms per cfg1 step with 2 queues per process, 25-500 ms
        # with 4-8).  Only this code-path mode; one process per GPU is not affected.
        env.setdefault("GPU_MAX_HW_QUEUES", "2")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(args.gpus),
           "--master-addr", "127.0.0.1", "--master-port", str(_free_port()), os.path.abspath(__file__)] + sys.argv[1:]
    sys.stdout.flush()
    os.execve(sys.executable, cmd, env)


def rccl_debug_summary(path):
    """What RCCL says it set up and chose (NCCL_DEBUG=INFO, subsystems INIT|GRAPH|TUNING|COLL, written to NCCL_DEBUG_FILE of rank 0):
    channel count, ring order per channel, the algorithm / protocol picked for the memory-bank exchange.  Tolerant text scraping --
    the point is that the FIRST multi-GPU run of this line is diagnostic (ring vs direct on the xGMI mesh)."""
    import glob
    import re
    files = sorted(glob.glob(path.replace("%p", "*").replace("%h", "*")))
    if not files:
        return {"note": "no RCCL debug file (%s); NCCL_DEBUG=%s NCCL_DEBUG_SUBSYS=%s" % (path, os.environ.get("NCCL_DEBUG"), os.environ.get("NCCL_DEBUG_SUBSYS"))}
    lines = []
    for f in files[:1]:
        try:
            lines += open(f, errors="replace").read().splitlines()
        except OSError:
            pass
    out = {"debug_lines": len(lines)}
    chan = [l for l in lines if re.search(r"Channel \d+/\d+ *:", l)]
    if chan:
        m = re.search(r"Channel \d+/(\d+) *:", chan[0])
        out["ring_channels"] = int(m.group(1))
        out["ring_order_channel0"] = re.search(r"Channel \d+/\d+ *: *(.*)", chan[0]).group(1).strip()[:120]
    for l in lines:
        if "coll channels" in l:
            out["channels_line"] = l.split("NCCL INFO", 1)[-1].strip()[:200]
            break
    algos = {}
    for l in lines:
        m = re.search(r"(AllGather|SendRecv|Broadcast|AllReduce)[^\n]*?(\d+) Bytes -> Algo (\S+) proto (\S+)", l)
        if m:
            algos["%s %s B" % (m.group(1), m.group(2))] = {"algo": m.group(3), "proto": m.group(4)}
    if algos:
        out["tuning"] = algos
    keep = [l.split("NCCL INFO", 1)[-1].strip()[:160] for l in lines
            if any(k in l for k in ("Connected all", "threadThresholds", "Init COMPLETE", "comm 0x", "NCCL_MAX_NCHANNELS", "P2P", "Using network"))]
    out["sample"] = keep[:12]
    return out


def build_model(workload, device):
    from estdepth_amd import DepthNetHybrid, synth
    _, _, _, D, resnet, est, _ = WORKLOADS[workload]
    m = DepthNetHybrid(ndepths=D, depth_min=0.1, depth_max=10.0, resnet=resnet, IF_EST_transformer=est)
    synth.fill_state_dict(m, seed=0, head_gain=1.0)
    m = m.eval().to(device)       # on a ROCm device the model switches its accelerators on by itself (DepthNetHybrid.accelerate)
    if str(device) != "cpu":
        # A/B switches of the individual accelerators (all on by default = what `DepthNetHybrid(...).cuda().eval()` runs)
        if os.environ.get("ESTD_NCHW_2D", "0") == "1":
            m.plain_path()
        else:
            if os.environ.get("ESTD_PSM", "hip") != "hip":
                m.use_hip_psm(False)                      # PSM 3x3 convs back on the library
            if os.environ.get("ESTD_FUSE_BN", "1") != "1":
                m.fuse_bn_2d(False)                       # separate BN / add / ReLU passes after the library convs
            if os.environ.get("ESTD_OVERLAP", "1") != "1":
                m.overlap_semantic_branch(False)          # semantic branch on the main stream
    return m


def make_inputs(workload, rank, device):
    from estdepth_amd import synth
    v, hi, wi = WORKLOADS[workload][:3]
    imgs, poses, intr, sample = synth.make_sequence(v, hi, wi, seed=1000 + rank)
    to = lambda t: t.to(device)
    return to(imgs), to(poses), to(intr), {k: to(t) for k, t in sample.items()}


def steady_state(model, workload, imgs, poses, intr, sample):
    """Untimed calls that establish the memory bank of the timed step.  Returns (frame slice, depth frames per step,
    pre_costs, pre_poses)."""
    import torch
    sub = lambda sl: {k: v[:, sl] for k, v in sample.items()}
    with torch.no_grad():
        if workload == "joint":       # two consecutive 5-frame calls with stride seq_len-2 (general_eval.py:52); call 2 is timed
            _, pre_costs, pre_poses = model(imgs[:, 0:5], poses[:, 0:5], intr, sub(slice(0, 5)), None, None, mode="val")
            return slice(3, 8), 3, pre_costs, pre_poses
        if workload in ("estm", "cfg5"):   # three sliding windows of 3 (eval_hybrid_seq.py:160-193); window 3 is timed
            _, c0, p0 = model(imgs[:, 0:3], poses[:, 0:3], intr, sub(slice(0, 3)), None, None, mode="val")
            _, c1, p1 = model(imgs[:, 1:4], poses[:, 1:4], intr, sub(slice(1, 4)),
                              {"keys": [c0["keys"][0]], "values": [c0["values"][0]]}, [p0[0]], mode="val")
            pre_costs = {"keys": [c0["keys"][0], c1["keys"][0]], "values": [c0["values"][0], c1["values"][0]]}
            return slice(2, 5), 1, pre_costs, [p0[0], p1[0]]
    return slice(0, 3), 1, None, None


def _cpu_model_name():
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name"):
                return line.split(":", 1)[1].strip()
    except OSError:
        pass
    return "unknown"


def _socket_cpus(package=None):
    """one hardware thread per physical core of ONE socket (the first sibling of every core of ``package``; default: the socket
    with the most cores allowed to this process) from /sys/devices/system/cpu/*/topology; [] if the topology cannot be read."""
    import glob
    allowed = os.sched_getaffinity(0)
    cores = {}
    for d in glob.glob("/sys/devices/system/cpu/cpu[0-9]*"):
        try:
            cpu = int(os.path.basename(d)[3:])
            pkg = int(open(os.path.join(d, "topology", "physical_package_id")).read())
            core = int(open(os.path.join(d, "topology", "core_id")).read())
        except (OSError, ValueError):
            continue
        if cpu in allowed:
            cores.setdefault(pkg, {}).setdefault(core, []).append(cpu)
    if not cores:
        return []
    if package is None:
        package = max(cores, key=lambda p: (len(cores[p]), -p))
    return sorted(min(sib) for sib in cores.get(package, {}).values())


def cpu_leg_pinned(workload, threads, x_imgs, x_poses, intr, pre_costs, pre_poses, frames, kind):
    """One CPU leg in a CHILD process confined to the physical cores of ONE socket: ``threads`` OpenMP / oneDNN workers, one per core,
    affinity set before the child creates its first thread (so every worker inherits it; memory is first-touched on that socket),
    OMP_PROC_BIND=close / OMP_PLACES=cores.  An unpinned run across both sockets is SLOWER at 128 threads than at 8 (round 5: 0.115 vs
    0.206 depth frames/s): the box's CPU path deserves the pinned figure beside it.  The inputs (and carried memory) of the GPU step
    travel through an .npz in the temporary directory; the child times the same model_forward call as the in-process legs."""
    import subprocess
    import tempfile
    import numpy as np
    cpus = _socket_cpus()
    threads = max(1, min(threads, len(cpus)))
    use = cpus[:threads]
    np_ = lambda t: t.detach().float().cpu().contiguous().numpy()
    arrays = {"imgs": np_(x_imgs), "poses": np_(x_poses), "intr": np_(intr)}
    if pre_costs is not None:
        for i, (k, v, p) in enumerate(zip(pre_costs["keys"], pre_costs["values"], pre_poses)):
            arrays["key%d" % i], arrays["value%d" % i], arrays["pose%d" % i] = np_(k), np_(v), np_(p)
    tmp = tempfile.mkdtemp(prefix="estd_cpuleg_")
    path = os.path.join(tmp, "step.npz")
    try:
        np.savez(path, **arrays)
        env = dict(os.environ, ESTD_CPU_LEG_CPUS=",".join(str(c) for c in use), OMP_NUM_THREADS=str(threads), MKL_NUM_THREADS=str(threads),
                   OMP_PROC_BIND="close", OMP_PLACES="cores", HIP_VISIBLE_DEVICES="", ROCR_VISIBLE_DEVICES="")
        for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT", "ESTD_FORCE_DIST", "ESTD_BENCH_CHILD"):
            env.pop(k, None)
        r = subprocess.run([sys.executable, os.path.abspath(__file__), "--cpu-leg-child", path, "--cpu-leg-kind", kind, "--workload", workload,
                            "--cpu-threads", str(threads)], env=env, capture_output=True, text=True, timeout=1800)
        res = None
        for line in reversed(r.stdout.splitlines()):
            if line.startswith("{"):
                res = json.loads(line)
                break
        if r.returncode != 0 or res is None:
            return {"value": 0.0, "unit": "depth frames/s", "cores": threads, "kind": kind, "wall_s": None, "pinned": None,
                    "error": "child failed (rc %d): %s" % (r.returncode, (r.stderr or "")[-300:])}
    finally:
        import shutil
        shutil.rmtree(tmp, ignore_errors=True)
    dt = res["wall_s"]
    how = "C/OpenMP oracle" if kind == "port" else "torch-CPU operators (oneDNN conv3d, ATen grid_sample / group_norm)"
    return {"value": round(frames / dt, 4), "unit": "depth frames/s", "cores": threads, "kind": kind, "wall_s": round(dt, 2),
            "pinned": "child process, %d threads on cpus %d-%d = physical cores of ONE socket (%d on it), affinity set before the first thread, "
                      "OMP_PROC_BIND=close OMP_PLACES=cores; torch reports %d threads" % (threads, use[0], use[-1], len(cpus), res["torch_threads"]),
            "cpu": "%s (%d hardware threads on the box)" % (_cpu_model_name(), os.cpu_count() or 1),
            "sample": "ONE full step of this workload (%d depth frames: 2D networks on torch-CPU + %s for the whole 3D hot path), %.1f s wall, "
                      "%d threads pinned to one socket" % (frames, how, dt, threads)}


def cpu_leg_child(args):
    """the child of cpu_leg_pinned(): affinity first (before numpy / torch create a thread), then the timed forward; prints one JSON line"""
    cpus = [int(c) for c in os.environ.get("ESTD_CPU_LEG_CPUS", "").split(",") if c]
    if cpus:
        os.sched_setaffinity(0, cpus)
    import contextlib
    import numpy as np
    import torch
    from oracle import ref_model as M, ref_ops as O, torch_ops as TO
    from oracle.nets2d import Nets2D, sd_numpy
    torch.set_num_threads(args.cpu_threads)
    O.set_num_threads(args.cpu_threads)
    z = np.load(args.cpu_leg_child)
    n_mem = sum(1 for k in z.files if k.startswith("key"))
    pc = {"keys": [z["key%d" % i] for i in range(n_mem)], "values": [z["value%d" % i] for i in range(n_mem)]} if n_mem else None
    pp = [z["pose%d" % i] for i in range(n_mem)] if n_mem else None
    cpu_model = build_model(args.workload, "cpu")
    P, nets = sd_numpy(cpu_model), Nets2D(model=cpu_model)
    a_imgs, a_poses, a_intr = z["imgs"], z["poses"], z["intr"]
    t0 = time.time()
    with (M.use_ops(TO) if args.cpu_leg_kind == "torch-ops" else contextlib.nullcontext()):
        ref, _, _ = M.model_forward(P, a_imgs, a_poses, a_intr, pc, pp, nets, ndepths=WORKLOADS[args.workload][3], depth_min=0.1, depth_max=10.0,
                                    IF_EST_transformer=WORKLOADS[args.workload][5])
    dt = time.time() - t0
    print(json.dumps({"wall_s": dt, "torch_threads": torch.get_num_threads(), "affinity": len(os.sched_getaffinity(0)),
                      "depth0_checksum": float(np.asarray(ref[("depth", 0, 0)], np.float64).sum())}), flush=True)


_DEFAULT_TORCH_THREADS = [0]        # torch's own default (= the physical cores of the box), recorded before anything changes it


def feature_parity(gpu_feats, nets):
    """Feature-level bar (round 6): the GPU path's PSM matching features [V,32,H/4,W/4] and the five ResNet scales of the timed step against
    the 2D networks the oracle ran (the product's plain nn.Modules on torch-CPU, oracle/nets2d.py) -- max |diff|, the reference's range and
    the two relative figures (max |diff| / range, ||diff||_2 / ||ref||_2).  The end of the pipeline (depth, logits) says little about the
    2D branches behind a flat softmax; any arithmetic change in them (Winograd tile sizes, operand splits) is held to THESE numbers."""
    import numpy as np
    out = {}
    pairs = [("psm_matching", gpu_feats["matching"], nets.last_matching)]
    pairs += [("resnet_scale%d" % i, g, r) for i, (g, r) in enumerate(zip(gpu_feats["semantic_features"], nets.last_semantic))]
    for name, g, r in pairs:
        g = g.detach().float().cpu().contiguous().numpy().astype(np.float64)
        r = np.asarray(r, np.float64).reshape(g.shape)
        d = np.abs(g - r)
        rng = float(np.abs(r).max())
        out[name] = {"max_abs_diff": float("%.3g" % d.max()), "ref_range": float("%.3g" % rng), "rel_to_range": float("%.3g" % (d.max() / max(rng, 1e-30))),
                     "rel_l2": float("%.3g" % (np.sqrt((d * d).sum()) / max(np.sqrt((r * r).sum()), 1e-30))), "elements": int(r.size)}
    return out


# bars of the feature-level comparison (tests/test_gpu_full_config.py reads them from here): max |diff| / range of the map.  Measured with the
# default kernels (round 6, cfg2 step): PSM 1.1e-6, ResNet scales 4.3e-7 .. 1.4e-6 -- the bar leaves ~3x; the emulation of tools/wino2d_f43_error.py puts
# F(4,3) on one axis of the PSM convolutions at 2.2e-6 (inside) and F(4x4) at 6e-6 (outside)
FEATURE_TOL_REL = {"psm_matching": 4e-6, "resnet": 4e-6}


def cpu_baseline(workload, threads, x_imgs, x_poses, intr, pre_costs, pre_poses, gpu_outputs, frames, gpu_logits=None, kind="port", pin_socket=False,
                 gpu_feats=None):
    """A CPU restatement of the reference timed on ONE full step of the same workload on the box's host cores: the whole
    DepthNetHybrid.forward of the timed step -- PSM, ResNet, plane sweeps, every 3D convolution, the 2N volume warps + attention +
    ConvGRU per target, soft-argmin, 2D refinement -- on the very inputs (and carried memory) of the GPU step.  Nothing is extrapolated.
      kind "port"      = the C/OpenMP oracle (oracle/estd_oracle.c) under the numpy composition of oracle/ref_model.py: the parity checker;
      kind "torch-ops" = the same composition on torch's own CPU operators (oracle/torch_ops.py: oneDNN convolutions, ATen grid_sample /
                         group_norm / softmax -- the operators the reference itself runs on a CPU).
    The 2D networks are the product's plain nn.Modules on torch-CPU in both.  Either call doubles as the full-size parity check of the
    benchmarked configuration (max |depth_gpu - depth_cpu|, abs_rel(depth_gpu, depth_cpu))."""
    import numpy as np
    import torch
    from oracle import ref_model as M, ref_ops as O, torch_ops as TO
    from oracle.nets2d import Nets2D, sd_numpy
    ncores = os.cpu_count() or 1
    threads = _DEFAULT_TORCH_THREADS[0] if threads <= 0 else min(threads, ncores)
    if pin_socket:
        return cpu_leg_pinned(workload, threads, x_imgs, x_poses, intr, pre_costs, pre_poses, frames, kind), None
    pinned = None
    O.set_num_threads(threads)
    torch.set_num_threads(threads)
    D = WORKLOADS[workload][3]
    cpu_model = build_model(workload, "cpu")
    P = sd_numpy(cpu_model)
    nets = Nets2D(model=cpu_model)
    np_ = lambda t: t.detach().float().cpu().contiguous().numpy()
    pc, pp = None, None
    if pre_costs is not None:
        pc = {"keys": [np_(k) for k in pre_costs["keys"]], "values": [np_(v) for v in pre_costs["values"]]}
        pp = [np_(p) for p in pre_poses]
    a_imgs, a_poses, a_intr = np_(x_imgs), np_(x_poses), np_(intr)
    import contextlib
    t0 = time.time()
    with (M.use_ops(TO) if kind == "torch-ops" else contextlib.nullcontext()):
        ref, _, _ = M.model_forward(P, a_imgs, a_poses, a_intr, pc, pp, nets, ndepths=D, depth_min=0.1, depth_max=10.0,
                                    IF_EST_transformer=WORKLOADS[workload][5])
    dt = time.time() - t0
    worst, arel = {}, {}
    for k, v in gpu_outputs.items():
        if k[0] == "depth":
            g = np_(v)
            worst[k[2]] = max(worst.get(k[2], 0.0), float(np.abs(g - ref[k]).max()))
            # abs_rel(pred, ref) = mean(|ref - pred| / ref) (metric.py:131-150, model_hybrid.py:306) of the GPU depth against the CPU depth
            arel.setdefault(k[2], []).append(float(np.mean(np.abs(ref[k] - g) / ref[k])))
    how = "C/OpenMP oracle" if kind == "port" else "torch-CPU operators (oneDNN conv3d, ATen grid_sample / group_norm)"
    base = {"value": round(frames / dt, 4), "unit": "depth frames/s", "cores": threads, "kind": kind, "wall_s": round(dt, 2), "pinned": pinned,
            "cpu": "%s (%d hardware threads on the box)" % (_cpu_model_name(), ncores),
            "sample": "ONE full step of this workload (%d depth frames: 2D networks on torch-CPU + %s for the whole 3D "
                      "hot path incl. volume warps, attention, ConvGRU), %.1f s wall, %d threads" % (frames, how, dt, threads)}
    parity = {"max_abs_depth_diff_vs_oracle_m": {"scale%d" % s: float("%.3g" % w) for s, w in sorted(worst.items())},
              "abs_rel_vs_oracle": {"scale%d" % s: float("%.3g" % (sum(a) / len(a))) for s, a in sorted(arel.items())},
              "abs_rel_def": "mean(|depth_oracle - depth_gpu| / depth_oracle) per output scale, averaged over the depth frames of the step "
                             "(metric.py:131-150, model_hybrid.py:306)",
              "oracle_kind": kind,
              "tolerance_m": 1e-4, "within_tolerance": bool(max(worst.values()) <= 1e-4),
              "what": "hipGraph/eager HIP path of THIS benchmark configuration vs the CPU oracle on the same inputs, all depth outputs"}
    if gpu_logits:
        # the raw logit volumes of stereo_head0 / stereo_head1 (hybrid_depth_decoder.py:200-204,:256-260): what the depth maps hide
        # behind a flat softmax -- every 3x3x3 convolution of the step undamped (bar: 6e-5, tests/test_gpu_full_config.py)
        lg = {}
        for name, key in (("init", ("init_logits",)), ("fused", ("fused_logits",))):
            if name in gpu_logits and key in ref:
                a, b = np_(gpu_logits[name]), np.asarray(ref[key])
                lg[name] = {"max_abs_diff": float("%.3g" % np.abs(a - b).max()), "oracle_range": float("%.3g" % np.abs(b).max()),
                            "oracle_std": float("%.3g" % b.std()), "voxels": int(b.size)}
        parity["logit_volumes_vs_oracle"] = lg
        parity["logit_tolerance"] = 6e-5
        parity["logits_within_tolerance"] = bool(lg and all(v["max_abs_diff"] <= 6e-5 for v in lg.values()))
    if gpu_feats is not None and nets.last_matching is not None and nets.last_semantic is not None:
        fp = feature_parity(gpu_feats, nets)
        parity["features_2d_vs_cpu_modules"] = fp
        parity["feature_tolerance_rel_to_range"] = FEATURE_TOL_REL
        parity["features_within_tolerance"] = bool(all(v["rel_to_range"] <= FEATURE_TOL_REL["psm_matching" if k == "psm_matching" else "resnet"]
                                                       for k, v in fp.items()))
    return base, parity


def _prime_until_captured(push, model, at_least, limit):
    """untimed calls until the set of captured hipGraphs has stopped growing for three calls in a row (a streaming harness in zero-copy mode
    walks through start-up signatures -- 0, 1, 2 memory records, no cached features -- and then through the slots of its memory ring: the
    steady-state captures are complete only after that), at least ``at_least`` and at most ``limit`` calls.  Returns the number of calls made."""
    n, stable, last = 0, 0, -1
    while n < limit and (n < at_least or stable < 3):
        push(n)
        n += 1
        now = len(getattr(model, "_graphs", ()))
        stable = stable + 1 if now == last else 0
        last = now
    return n


def stream_bench(args, device, rank, world):
    """Extra (not the headline): frame-by-frame ESTM streaming at cfg3 size through estdepth_amd.streaming.ESTMStream
    with the per-frame PSM feature cache; one step = one pushed frame = one depth frame."""
    import torch
    from estdepth_amd import synth
    from estdepth_amd.streaming import ESTMStream
    model = build_model("estm", device)
    prime_max = args.warmup + 12
    n = prime_max + args.steps
    imgs, poses, intr, _ = synth.make_sequence(n, 480, 640, seed=1003 + rank)
    imgs, poses, intr = imgs.to(device), poses.to(device), intr.to(device)
    st = ESTMStream(model, cache_features=True, graph=not args.no_graph)
    # fills the window and the memory; every capture of the steady state is made here, not inside the timed loop
    f0 = _prime_until_captured(lambda f: st.push(imgs[0, f], poses[0, f], intr[0]), st.model, args.warmup + 4, prime_max)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for f in range(f0, f0 + args.steps):
        st.push(imgs[0, f], poses[0, f], intr[0])
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    return {"metric": "depth frames/sec (ESTM streaming, 480x640, D=64, cached matching features)",
            "value": round(args.steps * world / dt, 3), "unit": "depth frames/s", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": round(1e3 * dt / args.steps, 3), "higher_is_better": True,
            "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": "ESTM stream: 1 new frame per step, window 3, memory 2, %s"
                                   % ("eager launches" if args.no_graph else "hipGraph replay (the new frame's PSM features inside stage A of the window forward)"),
                       "untimed_priming_calls": f0, "captures": len(getattr(st.model, "_graphs", ()))}}


def joint_stream_bench(steps, warmup, device, no_graph=False):
    """Extra (not the headline, which stays the whole forward): the Joint protocol as a stream -- consecutive 5-frame clips at stride 3
    (data/general_eval.py:52) with carried memory (eval_hybrid.py:229-243) through estdepth_amd.streaming.JointStream, which keeps the
    matching features of the two frames a clip shares with its predecessor: 3 of 5 frames go through the PSM extractor per step."""
    import torch
    from estdepth_amd import synth
    from estdepth_amd.streaming import JointStream
    model = build_model("joint", device)
    prime_max = warmup + 8
    nclips = prime_max + steps
    imgs, poses, intr, _ = synth.make_sequence(5 + 3 * (nclips - 1), 480, 640, seed=1002)
    imgs, poses, intr = imgs.to(device), poses.to(device), intr.to(device)
    st = JointStream(model, seq_len=5, cache_features=True, graph=not no_graph)
    clip = lambda c: st.push_clip(imgs[0, 3 * c:3 * c + 5], poses[0, 3 * c:3 * c + 5], intr[0])
    c0 = _prime_until_captured(clip, st.model, warmup + 3, prime_max)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for c in range(c0, c0 + steps):
        clip(c)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    del st, model
    torch.cuda.empty_cache()
    return {"workload": "Joint stream: 5-frame clips at stride 3 with carried memory, matching features of the 2 shared frames cached "
                        "(3 of 5 frames extracted per step), 480x640, D=64, ResNet-50, %s" % ("eager launches" if no_graph else "hipGraph replay"),
            "value": round(3 * steps / dt, 3), "unit": "depth frames/s", "ms_per_step": round(1e3 * dt / steps, 3), "steps": steps, "warmup": warmup,
            "depth_frames_per_step": 3, "untimed_priming_calls": c0}


def sustained_loop(step, drain, barrier, args, elapsed, device, dist, device_index):
    """The timed step once more, for about ``--sustained-s`` seconds: buckets of ``--sustained-bucket`` steps, each between two device
    synchronisations (one host wait per bucket), while a host thread samples shader clock and board power (estdepth_amd.profiling.GpuSampler:
    sysfs reads, nothing on a HIP stream).  The number of buckets is fixed BEFORE the loop from the K-step time (MAX over the ranks), so
    every rank of an N > 1 run makes the same number of calls (the exchange is a collective).  Returns the config.sustained dict."""
    import torch
    from estdepth_amd.profiling import GpuSampler
    t_step = elapsed / max(args.steps, 1)
    if dist is not None:
        tt = torch.tensor([t_step], device=device, dtype=torch.float64)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        t_step = float(tt[0])
    bucket = max(1, args.sustained_bucket)
    nb = max(2, int(round(args.sustained_s / (bucket * t_step))))
    barrier()
    sampler = GpuSampler(device_index).start()
    t0 = time.perf_counter()
    edges, per = [0.0], []
    for _ in range(nb):
        tb = time.perf_counter()
        for _ in range(bucket):
            step()
        drain()
        torch.cuda.synchronize()
        te = time.perf_counter()
        per.append(1e3 * (te - tb) / bucket)
        edges.append(te - t0)
    barrier()
    total = time.perf_counter() - t0
    sampler.stop()
    if dist is not None:
        tt = torch.tensor([total], device=device, dtype=torch.float64)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        total = float(tt[0])
    ms = 1e3 * total / (nb * bucket)
    return {"seconds": round(total, 2), "steps": nb * bucket, "bucket_steps": bucket, "ms_per_step": round(ms, 3),
            "ms_per_step_per_bucket": [round(x, 3) for x in per], "slowest_bucket_ms": round(max(per), 3), "fastest_bucket_ms": round(min(per), 3),
            "timed_ms_per_step": round(1e3 * t_step, 3), "timed_over_sustained": round(1e3 * t_step / ms, 4),
            "clocks": sampler.summary(edges),
            "how": "after the K timed steps: %d buckets of %d steps of the same call, one device synchronisation per bucket, wall clock; "
                   "shader clock / power from a host thread" % (nb, bucket)}


GRAPH_PRIME = 2      # untimed calls in front of the warm-up steps of a hipGraph run: zero-copy memory alternates between two captures


def quick_measure(workload, steps, warmup, device, no_graph=False, zero_copy=True):
    """One of the OTHER single-GPU workloads, timed the same way as the headline (warm-up, K steps between synchronisations, hipGraph
    replay of the whole forward) but without roofline / parity / CPU legs: the numbers beside the headline line."""
    import torch
    from estdepth_amd.graph import GraphedForward
    model = build_model(workload, device)
    imgs, poses, intr, sample = make_inputs(workload, 0, device)
    sl, frames, pre_costs, pre_poses = steady_state(model, workload, imgs, poses, intr, sample)
    x_imgs, x_poses = imgs[:, sl].contiguous(), poses[:, sl].contiguous()
    x_sample = {k: v[:, sl] for k, v in sample.items()}
    fwd = model if no_graph else GraphedForward(model, zero_copy_memory=zero_copy)
    with torch.no_grad():
        for _ in range(warmup + (0 if no_graph else GRAPH_PRIME)):
            fwd(x_imgs, x_poses, intr, x_sample, pre_costs, list(pre_poses) if pre_poses else None, mode="val")
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(steps):
            fwd(x_imgs, x_poses, intr, x_sample, pre_costs, list(pre_poses) if pre_poses else None, mode="val")
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
    del fwd, model
    torch.cuda.empty_cache()
    return {"workload": WORKLOADS[workload][6], "value": round(frames * steps / dt, 3), "unit": "depth frames/s",
            "ms_per_step": round(1e3 * dt / steps, 3), "steps": steps, "warmup": warmup, "depth_frames_per_step": frames}


def replay_profile(args):
    """Per-kernel durations INSIDE the hipGraph replay: a short child run of this very file under ``rocprofv3 --kernel-trace``
    (HIP events cannot bracket graph nodes; the eager event brackets perturb what overlaps with what).  Returns
    (families, info) of estdepth_amd.profiling.replay_families, or (None, {"error": ...})."""
    import shutil
    import subprocess
    import tempfile
    from estdepth_amd import profiling
    rp = shutil.which("rocprofv3") or ("/opt/rocm/bin/rocprofv3" if os.path.exists("/opt/rocm/bin/rocprofv3") else None)
    if rp is None:
        return None, {"error": "rocprofv3 not found"}
    steps = 5
    out = tempfile.mkdtemp(prefix="estd_replay_")
    env = dict(os.environ, ESTD_BENCH_CHILD="1", TMPDIR=tempfile.gettempdir())
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT", "ESTD_FORCE_DIST"):
        env.pop(k, None)
    cmd = [rp, "--kernel-trace", "--output-format", "csv", "-d", out, "-o", "p", "--", sys.executable, os.path.abspath(__file__),
           "--workload", args.workload, "--steps", str(steps), "--warmup", "2", "--no-cpu-baseline", "--no-alt",
           "--conv3d-arith", args.conv3d_arith, "--conv2d-arith", args.conv2d_arith, "--conv3d-algo", args.conv3d_algo,
           "--graph-memory", args.graph_memory, "--pipeline", "off", "--sustained-s", "0"] + (["--no-graph"] if args.no_graph else [])
    t0 = time.time()
    try:
        r = subprocess.run(cmd, cwd=tempfile.gettempdir(), env=env, capture_output=True, text=True, timeout=600)
        trace = None
        for root, _, files in os.walk(out):
            for f in files:
                if f.endswith("kernel_trace.csv"):
                    trace = os.path.join(root, f)
        if r.returncode != 0 or trace is None:
            return None, {"error": "child run failed (rc %d): %s" % (r.returncode, (r.stderr or "")[-200:])}
        fams, info = profiling.replay_families(trace, steps)
        child_ms = None
        for line in reversed(r.stdout.splitlines()):
            if line.startswith("{"):
                child_ms = json.loads(line).get("ms_per_step")
                break
        info.update({"how": "rocprofv3 --kernel-trace of a child run of this command (%d steps, %s), kernels between the two estd_mark_kernel "
                            "launches of its timed loop" % (steps, "eager launches" if args.no_graph else
                                                            "hipGraph replay" + (", SERIAL (--pipeline off: per-kernel durations without the next step's stage A beside them; "
                                                                               "config.serial_replay is this configuration timed in the parent)" if args.pipeline == "on" else "")),
                     "child_ms_per_step_under_the_profiler": child_ms, "wall_s": round(time.time() - t0, 1)})
        return fams, info
    except Exception as e:
        return None, {"error": "%s: %s" % (type(e).__name__, str(e)[:160])}
    finally:
        shutil.rmtree(out, ignore_errors=True)


def conv3d_algo_of(group, algo, arith):
    """which kernel a profiled conv3d group ran on: every instance follows --conv3d-algo; under wino2 the 33->33 instance (dres2) runs on
    the two-axis kernel's XOUT instance (ops.W2_XOUT, the default: 12/27 of the products) and on the depth-only kernel (18/27) only with
    ESTD_W2_XOUT=0; 32->16 and the 16->16 heads have wino2 instances and fall back to the direct kernel under wino."""
    from estdepth_amd import ops
    if arith != "f32":
        return "direct"
    if algo == "wino2" and ops.W3 and (group == "conv3d:32->32" or (group == "conv3d:33->32" and ops.W3_EXTRA)):
        return "wino3"                                   # all three axes in Winograd form (csrc/conv3d_wino3.hip: 8/27 of the products)
    if group in ("conv3d:32->32", "conv3d:33->32"):      # the key || value convolution (33 -> 32) has a wino2 instance as well
        return algo
    if group == "conv3d:33->1":                              # dres2's 33rd output channel: the taps as matrix rows (csrc/conv3d_xout.hip)
        return "taps"
    if group == "conv3d:33->33":
        if algo == "wino2":
            return "wino2" if ops.W2_XOUT else "wino"
        return algo
    if group in ("conv3d:32->16", "conv3d:16->16"):  # the GRU output convolution (16-output-channel instance of the wino2 kernel) and the
        return "wino2" if algo == "wino2" else "direct"      # stereo heads (csrc/conv3d_wino2_c16.hip)
    return "direct"


def summarize(prof, peak_tf, algo="direct", arith="f32", replay=None):
    """ops.PROFILE entries -> per-group averages.  FLOP groups ("conv3d:*") against the MFMA peak, byte groups against HBM.
    `frac` of a convolution = EXECUTED MFMA FLOPs / peak (never above 1); the algorithmic rate (all 27 taps counted, SURVEY
    §8d) is carried beside it."""
    groups = {}
    for group, amount, e0, e1 in prof:
        g = groups.setdefault(group, [0, 0.0, 0.0])
        g[0] += 1
        g[1] += amount
        g[2] += e0.elapsed_time(e1)
    mfma, hbm = {}, {}
    for name, (n, amount, ms) in sorted(groups.items()):
        if ms <= 0:
            continue
        if name.startswith("conv3d:"):
            tf = amount / (ms * 1e-3) / 1e12
            kalgo = conv3d_algo_of(name, algo, arith)
            ex = EXECUTED_FACTOR[kalgo]
            mfma[name] = {"launches": n, "eager_bracket_ms": round(ms / n, 4), "gflop_per_launch": round(amount / n / 1e9, 2),
                          "kernel_algo": kalgo, "executed_factor": round(ex, 4),
                          "eager": {"achieved_tflops": round(tf * ex, 2), "frac": round(tf * ex / peak_tf, 4),
                                    "algorithmic_tflops": round(tf, 2), "algorithmic_frac": round(tf / peak_tf, 4)}}
            rep = (replay or {}).get(name)
            if rep:      # the figures of the hipGraph replay that produced `value` (rocprofv3 trace of the child run)
                rtf = amount / n / (rep["avg_launch_ms"] * 1e-3) / 1e12
                mfma[name].update({"avg_launch_ms": rep["avg_launch_ms"], "launches_per_step": rep["launches_per_step"], "source": "replay",
                                   "achieved_tflops": round(rtf * ex, 2), "frac": round(rtf * ex / peak_tf, 4),
                                   "algorithmic_tflops": round(rtf, 2), "algorithmic_frac": round(rtf / peak_tf, 4),
                                   "eager_over_replay": round(ms / n / rep["avg_launch_ms"], 3)})
            else:
                mfma[name].update({"avg_launch_ms": round(ms / n, 4), "source": "eager brackets (no replay trace)"})
                mfma[name].update(mfma[name]["eager"])
        else:
            gbs = amount / (ms * 1e-3) / 1e9
            hbm[name] = {"launches": n, "eager_bracket_us": round(1e3 * ms / n, 1), "algorithmic_mb_per_launch": round(amount / n / 1e6, 2),
                         "eager": {"achieved_gbs": round(gbs, 1), "frac": round(gbs / HBM_PEAK_GBS, 4)}}
            rep = (replay or {}).get(name)
            if rep:
                rgbs = amount / n / (rep["avg_launch_ms"] * 1e-3) / 1e9
                hbm[name].update({"avg_launch_us": round(1e3 * rep["avg_launch_ms"], 1), "launches_per_step": rep["launches_per_step"], "source": "replay",
                                  "achieved_gbs": round(rgbs, 1), "frac": round(rgbs / HBM_PEAK_GBS, 4),
                                  "eager_over_replay": round(ms / n / rep["avg_launch_ms"], 3)})
            else:
                hbm[name].update({"avg_launch_us": round(1e3 * ms / n, 1), "source": "eager brackets (no replay trace)"})
                hbm[name].update(hbm[name]["eager"])
    return mfma, hbm


def main():
    args = parse()
    if args.cpu_leg_child:
        return cpu_leg_child(args)
    if "bf16x3" in (args.conv3d_arith, args.conv2d_arith) or args.conv3d_algo == "wino":
        # the superseded A/B kernels are not part of the default library: say so here, once, instead of in the middle of the first step
        from estdepth_amd import _native
        if not _native.has_ab():
            raise SystemExit("bench.py: --conv3d-arith/--conv2d-arith bf16x3 and --conv3d-algo wino run on the superseded A/B kernels, which the "
                             "default build does not carry: rebuild with ESTD_BUILD_AB=1 (`ESTD_BUILD_AB=1 python -m estdepth_amd.build`) and "
                             "run with ESTD_BUILD_AB=1 in the environment")
    self_launch_if_needed(args)
    if (int(os.environ.get("WORLD_SIZE", "1")) > 1 or os.environ.get("ESTD_FORCE_DIST", "0") == "1") \
            and os.environ.get("ESTD_DIST_BACKEND", "nccl") == "nccl" and os.environ.get("ESTD_RCCL_DEBUG", "1") == "1":
        # RCCL's own account of its topology / channels / algorithm choice goes to a file per process (stdout keeps the one JSON
        # line); rank 0's file is summarised into config.allgather.rccl.  Set BEFORE torch (and with it librccl) is loaded: the
        # library latches its debug level at its first log call.
        import tempfile
        if os.environ.get("NCCL_DEBUG", "").upper() not in ("INFO", "TRACE"):
            os.environ["NCCL_DEBUG"] = "INFO"
        os.environ.setdefault("NCCL_DEBUG_SUBSYS", "INIT,GRAPH,TUNING,COLL")
        os.environ.setdefault("NCCL_DEBUG_FILE", os.path.join(tempfile.gettempdir(), "estd_rccl_%d_r%s.log" % (os.getpid(), os.environ.get("RANK", "0"))))
    import torch
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    _DEFAULT_TORCH_THREADS[0] = torch.get_num_threads()
    if not torch.cuda.is_available():
        raise RuntimeError("bench.py needs a ROCm GPU (the product path has no CPU fallback)")
    oversub = int(os.environ.get("ESTD_OVERSUBSCRIBED", "0"))
    if oversub:                                     # fewer GPUs than ranks (1-GPU test box): share devices, gloo collectives
        local_rank = local_rank % oversub
    if "ESTD_FORCE_DEVICE" in os.environ:
        local_rank = int(os.environ["ESTD_FORCE_DEVICE"])
    torch.cuda.set_device(local_rank)
    device = torch.device("cuda", local_rank)
    dist = None
    backend = None
    # ESTD_FORCE_DIST=1 with one rank: a world-size-1 RCCL communicator on the one GPU -- every line of the N > 1 code (process
    # group with device_id, channel cap, CU reserve, the asynchronous no-staging all-gather overlapped with the next step) runs
    # against real RCCL on a 1-GPU box.
    force_dist = world == 1 and os.environ.get("ESTD_FORCE_DIST", "0") == "1"
    if force_dist:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", str(_free_port()))
        os.environ.setdefault("RANK", "0")
        os.environ.setdefault("WORLD_SIZE", "1")
    dist_on = world > 1 or force_dist
    if dist_on:
        import torch.distributed as dist
        # The only collective is the per-step memory-bank all-gather (157 MB per rank, ~48 GB/s of ingress at N = 8): a few RCCL
        # channels carry it inside one step.  Every channel is a workgroup that holds a CU: as many channels as CUs the
        # convolution grids leave free (estd_set_reserved_cus(8) below: one per XCD).
        os.environ.setdefault("NCCL_MAX_NCHANNELS", "8")
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
    ops.CONV3D_ALGO = args.conv3d_algo
    if args.workload == "stream":
        line = stream_bench(args, device, rank, world)
        if rank == 0:
            print(json.dumps(line), flush=True)
        if dist_on:
            dist.barrier()
            dist.destroy_process_group()
        return
    model = build_model(args.workload, device)
    model.CostRegNet.keep_logits = True      # keeps references to the two logit volumes of the last forward (no extra work): parity block
    imgs, poses, intr, sample = make_inputs(args.workload, rank, device)
    sl, frames, pre_costs, pre_poses = steady_state(model, args.workload, imgs, poses, intr, sample)
    x_imgs, x_poses = imgs[:, sl].contiguous(), poses[:, sl].contiguous()
    x_sample = {k: v[:, sl] for k, v in sample.items()}

    from estdepth_amd.graph import GraphedForward
    zero_copy = args.graph_memory == "zero-copy"
    state = {"pending": None, "fwd": None, "allgather": dist_on and not args.no_allgather, "notes": [], "bank": None}
    if force_dist:
        state["notes"].append("ESTD_FORCE_DIST=1: world-size-1 RCCL communicator on one GPU (code-path + overlap-cost measurement, not scaling)")
    reserve = None
    if state["allgather"] and not oversub:
        # the RCCL exchange of step k runs on a few CUs WHILE step k+1 computes.  The convolutions launch one or two resident
        # workgroups per CU with static tile ranges: were all 256 CUs claimed, the workgroups displaced by the collective would
        # queue behind the others and double the launch.  Leave one CU per XCD free -- but only where the exchange runs: it starts
        # when stage B of step k ends and overlaps stage A (the 2D networks, ~30 % of a step) of step k + 1; an exchange shorter than
        # that (the direct exchange over xGMI: ~1 ms; the device-local copy of a world-size-1 run) needs no reserve in stage B, which
        # holds the 3D convolutions the reserve costs most (0.6 ms per Joint step for the whole-step reserve).
        n_res = int(os.environ.get("ESTD_RESERVED_CUS", "8"))
        scope = os.environ.get("ESTD_RESERVE_SCOPE", "auto")          # auto | A | AB
        probe = None
        if scope == "auto" and not args.no_graph:
            # one eager forward for a record, then the exchange alone (the algorithm the timed steps will use) against that step
            with torch.no_grad():
                model(x_imgs, x_poses, intr, x_sample, pre_costs, list(pre_poses) if pre_poses else None, mode="val")
                torch.cuda.synchronize()
                te = time.perf_counter()
                _, c_, p_ = model(x_imgs, x_poses, intr, x_sample, pre_costs, list(pre_poses) if pre_poses else None, mode="val")
                torch.cuda.synchronize()
                t_step = time.perf_counter() - te
                lg_ = model.CostRegNet.memory_logits
                parallel.allgather_memory_bank_async(c_, p_, stage=False, logits=lg_).wait()
                torch.cuda.synchronize()
                dist.barrier()
                te = time.perf_counter()
                for _ in range(3):
                    parallel.allgather_memory_bank_async(c_, p_, stage=False, logits=lg_).wait()
                torch.cuda.synchronize()
                t_x = torch.tensor([(time.perf_counter() - te) / 3, t_step], device=device, dtype=torch.float64)
                dist.all_reduce(t_x, op=dist.ReduceOp.MAX)            # every rank decides on the same numbers
                del c_, p_
            probe = {"exchange_alone_ms": round(1e3 * float(t_x[0]), 3), "eager_step_ms": round(1e3 * float(t_x[1]), 3)}
            scope = "A" if float(t_x[0]) < 0.25 * float(t_x[1]) else "AB"
        elif scope == "auto":
            scope = "AB"
        if scope not in ("A", "AB"):
            raise RuntimeError("ESTD_RESERVE_SCOPE must be auto, A or AB, got %r" % (scope,))
        reserve = (n_res, n_res if scope == "AB" else 0)
        state["reserved_cus"], state["reserve_scope"], state["reserve_probe"] = n_res, scope, probe
        if args.no_graph:
            ops.set_reserved_cus(n_res)                                # eager launches: the process-wide setting, every kernel
    pipelined = (not args.no_graph) and args.pipeline == "on"
    if pipelined and state["allgather"]:
        raise SystemExit("bench.py: --pipeline on needs --no-allgather in a distributed run (the exchange consumes every step's record on the caller's stream)")
    fwd = model if args.no_graph else GraphedForward(model, zero_copy_memory=zero_copy, reserve_cus=reserve, pipeline=pipelined)     # hipGraph replay of the same forward (same kernels)
    state["fwd"] = fwd
    if oversub:
        state["notes"].append("%d ranks share %d GPU(s), gloo collectives: code-path check, not a scaling measurement" % (world, oversub))

    def drain():
        if state["pending"] is not None:
            state["bank"] = state["pending"].wait()
            state["pending"] = None

    def step(f=None):
        f = state["fwd"] if f is None else f
        with torch.no_grad():
            try:
                out, costs, cposes = f(x_imgs, x_poses, intr, x_sample, pre_costs, list(pre_poses) if pre_poses else None, mode="val")
            except Exception as e:                       # graph capture refused on this stack: keep measuring, eagerly
                if f is model:
                    raise
                state["notes"].append("hipGraph capture failed (%s: %s): eager launches" % (type(e).__name__, str(e)[:80]))
                state["fwd"] = model
                torch.cuda.synchronize()
                out, costs, cposes = model(x_imgs, x_poses, intr, x_sample, pre_costs, list(pre_poses) if pre_poses else None, mode="val")
            if state["allgather"]:
                # memory bank {K, V_fused, pose} of this window -> every rank; the collective of step k overlaps step k+1
                try:
                    drain()
                    # GraphedForward hands back fresh memory tensors: they are sent from where they lie (no staging copy)
                    # + the frame's initial logit volume (the per-frame probability volume of north_star before its softmax)
                    state["logits"] = state["fwd"].memory_logits if state["fwd"] is not model else model.CostRegNet.memory_logits
                    state["pending"] = parallel.allgather_memory_bank_async(costs, cposes, stage=state["fwd"] is model, logits=state["logits"])
                except Exception as e:                   # same failure on every rank (collective): keep the shards running
                    state["notes"].append("memory-bank all-gather failed (%s: %s): disabled" % (type(e).__name__, str(e)[:80]))
                    state["allgather"], state["pending"] = False, None
        return out, costs, cposes

    def barrier():
        if dist_on:
            dist.barrier()
        torch.cuda.synchronize()

    if not args.no_graph:
        for _ in range(GRAPH_PRIME * (2 if pipelined else 1)):       # the captures (zero-copy memory alternates between two ring buffers = two captures; x 2 lanes when pipelined): untimed set-up
            step()
    for _ in range(args.warmup):
        step()
    barrier()
    ops.profile_mark(0)                    # estd_mark_kernel brackets the timed region in a rocprofv3 trace
    t0 = time.perf_counter()
    for _ in range(args.steps):
        last = step()
    drain()                                # the last window's all-gather is inside the timed region
    if pipelined and state["fwd"] is not model:
        state["fwd"].join()                # (the trace marker below belongs behind the last step's stage B, which runs on a stream of its own)
    ops.profile_mark(1)
    barrier()
    elapsed = time.perf_counter() - t0
    gathered = state["allgather"]
    gpu_outputs = {k: v.clone() for k, v in last[0].items()}        # outputs of the timed configuration (for the parity report)
    gpu_logits = {k: v.clone() for k, v in (getattr(model.CostRegNet, "last_logits", None) or {}).items()}
    f2d = getattr(state["fwd"], "last_features2d", None) or getattr(model, "last_features2d", None)       # stage A's outputs of the last timed step
    gpu_feats = {"matching": f2d["matching"].clone(), "semantic_features": [t.clone() for t in f2d["semantic_features"]]} if f2d else None
    # "gathered bank equals each owner's tensors bit for bit" -- every rank checks the shard it owns
    bank_ok = None
    if gathered and state["bank"] is not None:
        (bc, bp) = state["bank"][rank]
        bank_ok = bool(torch.equal(bc["keys"][0], last[1]["keys"][0]) and torch.equal(bc["values"][0], last[1]["values"][0])
                       and torch.equal(bp[0].to(last[2][0].dtype), last[2][0])
                       and "logits" in bc and torch.equal(bc["logits"][0], state["logits"]))
        if not bank_ok:
            raise RuntimeError("rank %d: the all-gathered memory bank differs from the tensors this rank sent" % rank)

    # ---- is the K-step figure a sustained one?  The same step (same launch path, same overlapped exchange) for tens of seconds ----
    sustained = None
    if args.sustained_s > 0 and os.environ.get("ESTD_BENCH_CHILD") != "1":
        sustained = sustained_loop(step, drain, barrier, args, elapsed, device, dist if world > 1 else None, local_rank)

    # ---- pipelined replay: the SAME K steps once more on the serial replay (stage A and stage B of a step back to back on one stream, no
    #      overlap between steps): the A/B of the pipeline inside every line, and the configuration the per-kernel trace below describes ----
    serial = None
    if pipelined and state["fwd"] is not model and os.environ.get("ESTD_BENCH_CHILD") != "1":
        try:
            f_ser = GraphedForward(model, zero_copy_memory=zero_copy, reserve_cus=reserve, pipeline=False)
            for _ in range(GRAPH_PRIME + args.warmup):
                step(f_ser)
            barrier()
            ts = time.perf_counter()
            for _ in range(args.steps):
                step(f_ser)
            barrier()
            t_ser = time.perf_counter() - ts
            serial = {"ms_per_step": round(1e3 * t_ser / args.steps, 3), "value": round(frames * args.steps / t_ser, 3),
                      "pipelined_over_serial": round(t_ser / elapsed, 4),
                      "what": "GraphedForward(pipeline=False): the round-5 launch path, same kernels, same K steps"}
            del f_ser
        except Exception as e:
            serial = {"error": "%s: %s" % (type(e).__name__, str(e)[:100])}

    # ---- the collective alone (N > 1): bytes per rank and achieved bus bandwidth ----
    ag = None
    if dist_on and gathered:
        with torch.no_grad():
            reps = 5
            def exchange_alone(algo):
                parallel.allgather_memory_bank_async(last[1], last[2], stage=False, logits=state["logits"], algo=algo).wait()
                barrier()
                ta = time.perf_counter()
                for _ in range(reps):
                    parallel.allgather_memory_bank_async(last[1], last[2], stage=False, logits=state["logits"], algo=algo).wait()
                barrier()
                return (time.perf_counter() - ta) / reps
            t_ag = exchange_alone(None)                     # the algorithm the timed steps used (ESTD_AG_ALGO; "auto" starts on the one all-gather)
            used_algo = parallel.active_algo()
            other = "direct" if used_algo == "collective" else "collective"
            # the other one beside it (ring-vs-direct on the xGMI mesh is the open question) is timed LAST, under a watchdog, when the
            # line is complete (other_algo_diagnostic below): it is the one call of a multi-GPU run that no timed step has exercised
            t_other = None
            if world > 1:
                state["other_algo"] = (other, exchange_alone)
        nbytes = 4 * (last[1]["keys"][0].numel() + last[1]["values"][0].numel() + 16 + state["logits"].numel())
        ag = {"record": "K||V_fused (%d B) + pose (64 B) + initial logit volume (%d B)" % (4 * 2 * last[1]["keys"][0].numel(), 4 * state["logits"].numel()),
              "bytes_sent_per_rank": nbytes, "algo": used_algo, "algo_setting": parallel.AG_ALGO, "ms_alone": round(1e3 * t_ag, 3),
              "bus_gbs_per_rank": round((world - 1) * nbytes / t_ag / 1e9, 2),
              "other_algo": {"algo": other, "ms_alone": round(1e3 * t_other, 3), "bus_gbs_per_rank": round((world - 1) * nbytes / t_other / 1e9, 2)} if t_other else None,
              "local_copy_gbs": round(nbytes / t_ag / 1e9, 2) if world == 1 else None,
              "own_shard_bit_equal": bank_ok,
              "backend": "RCCL (nccl)" if backend == "nccl" else backend,
              "nccl_max_nchannels": os.environ.get("NCCL_MAX_NCHANNELS")}
        if backend == "nccl" and rank == 0 and os.environ.get("NCCL_DEBUG_FILE"):
            ag["rccl"] = rccl_debug_summary(os.environ["NCCL_DEBUG_FILE"])
    state["allgather"] = False

    def timed_plain(f=None):
        """the same K steps (after W warm-up steps) with forward ``f`` and whatever state["allgather"] says; max over the ranks, seconds"""
        for _ in range(args.warmup):
            step(f)
        barrier()
        ta = time.perf_counter()
        for _ in range(args.steps):
            step(f)
        drain()
        barrier()
        tl = time.perf_counter() - ta
        if world > 1:
            tt_ = torch.tensor([tl], device=device, dtype=torch.float64)
            dist.all_reduce(tt_, op=dist.ReduceOp.MAX)
            tl = float(tt_[0])
        return tl

    if dist_on and gathered:
        # every N: the same K steps WITHOUT the collective (CU reserve still in place) -- what the overlapped exchange costs a step, and the
        # figure that is comparable with the driver's N = 1 line
        t_nc = timed_plain()
        ag["ms_per_step_without_collective"] = round(1e3 * t_nc / args.steps, 3)
        ag["ms_per_step_with_collective"] = round(1e3 * elapsed / args.steps, 3)
        ag["reserved_cus"], ag["reserve_scope"] = state.get("reserved_cus"), state.get("reserve_scope")
        ag["reserve_scope_probe"] = state.get("reserve_probe")
        if state.get("reserved_cus") and not args.no_graph and not oversub and os.environ.get("ESTD_RESERVE_COST", "1") != "0":
            # ... and without the reserve either (a third set of captures with full grids): what leaving the CUs free costs a step
            try:
                f0 = GraphedForward(model, zero_copy_memory=zero_copy, reserve_cus=(0, 0))
                for _ in range(GRAPH_PRIME):
                    step(f0)
                t_nr = timed_plain(f0)
                ag["ms_per_step_without_collective_without_reserve"] = round(1e3 * t_nr / args.steps, 3)
                ag["reserve_cost_ms"] = round(1e3 * (t_nc - t_nr) / args.steps, 3)
                del f0
            except Exception as e:
                ag["reserve_cost_ms"] = None
                state["notes"].append("reserve-cost loop failed (%s: %s)" % (type(e).__name__, str(e)[:80]))

    child = os.environ.get("ESTD_BENCH_CHILD") == "1"       # the traced child run of replay_profile(): the timed loop is all it is for
    if child:
        if rank == 0:
            print(json.dumps({"value": round(frames * args.steps / elapsed, 3), "ms_per_step": round(1e3 * elapsed / args.steps, 3),
                              "steps": args.steps, "child": True}), flush=True)
        return
    # Rooflines: the same steps once more, launched eagerly, with a HIP-event pair around every hot-path launch on its
    # launch stream (events cannot bracket nodes inside a graph replay).  Rank 0 only.
    prof = []
    if rank == 0:
        ops.PROFILE = []
        for _ in range(args.steps):
            step(model)
        torch.cuda.synchronize()
        prof, ops.PROFILE = ops.PROFILE, None
    # ... and what those kernels take INSIDE the hipGraph replay that produced `value`: kernel trace of a short child run
    replay, replay_info = None, None
    if rank == 0 and world == 1 and not force_dist and not args.no_replay_profile:
        replay, replay_info = replay_profile(args)
    # Second opinion, reported beside (never instead of) the headline: the same K steps with the 3x3x3 / 3x3 convolutions
    # on the exact 3-way bf16 operand split (fp32-level error, tests/test_gpu_split_conv.py), N = 1 only.
    alt = None
    from estdepth_amd import _native
    if world == 1 and not args.no_alt and (args.conv3d_arith, args.conv2d_arith) == ("f32", "f32") and _native.has_ab():     # (A/B build only: ESTD_BUILD_AB=1)
        try:
            ops.CONV3D_ARITH = ops.CONV2D_ARITH = "bf16x3"
            state["fwd"] = model if args.no_graph else GraphedForward(model, zero_copy_memory=zero_copy)
            for _ in range(args.warmup + (0 if args.no_graph else GRAPH_PRIME)):
                step()
            barrier()
            ta = time.perf_counter()
            for _ in range(args.steps):
                o_alt = step()[0]
            barrier()
            alt_elapsed = time.perf_counter() - ta
            ddiff = max(float((o_alt[k] - gpu_outputs[k]).abs().max()) for k in gpu_outputs if k[0] == "depth")
            alt = {"conv3d_arith": "bf16x3", "conv2d_arith": "bf16x3", "value": round(frames * args.steps / alt_elapsed, 3),
                   "ms_per_step": round(1e3 * alt_elapsed / args.steps, 3),
                   "max_abs_depth_diff_vs_headline_arith_m": float("%.3g" % ddiff),
                   "note": "opt-in (--conv3d-arith/--conv2d-arith bf16x3): every fp32 product as six bf16 MFMA products of exactly "
                           "3-way-split operands, fp32 accumulation; same 1e-4 parity tests, conv error vs fp64 equal to the fp32 MFMA kernel's"}
        except Exception as e:
            alt = {"error": "%s: %s" % (type(e).__name__, str(e)[:120])}
        finally:
            ops.CONV3D_ARITH, ops.CONV2D_ARITH = args.conv3d_arith, args.conv2d_arith
    per_rank_ms = [1e3 * elapsed / args.steps]
    if world > 1:
        tt = torch.tensor([elapsed], device=device, dtype=torch.float64)
        allt = [torch.zeros_like(tt) for _ in range(world)]
        dist.all_gather(allt, tt)
        per_rank_ms = [round(1e3 * float(t.item()) / args.steps, 3) for t in allt]
        elapsed = max(float(t.item()) for t in allt)

    if rank == 0:
        value = frames * world * args.steps / elapsed
        peak = PEAK_FP32_MATRIX_TFLOPS if args.conv3d_arith == "f32" else PEAK_BF16_MATRIX_TFLOPS / 6.0
        mfma, hbm = summarize(prof, peak, args.conv3d_algo, args.conv3d_arith, replay)
        kalgo = conv3d_algo_of("conv3d:32->32", args.conv3d_algo, args.conv3d_arith)
        zero = {"achieved_tflops": 0.0, "frac": 0.0, "algorithmic_tflops": 0.0, "algorithmic_frac": 0.0}
        dom = mfma.get("conv3d:32->32", dict(zero, launches=0, eager_bracket_ms=0.0, gflop_per_launch=0.0, eager=zero))
        # per-family agreement of the two measurements (>15 % apart = a kernel whose neighbours differ between the eager bracket pass
        # and the replay: the replay figure is the one that describes `value`)
        perturbed = sorted(k for k, v in list(mfma.items()) + list(hbm.items()) if abs(v.get("eager_over_replay", 1.0) - 1.0) > 0.15)
        # HBM bytes per launch of the dominant kernel from the committed rocprofv3 PMC passes (FETCH_SIZE x2 correction +
        # WRITE_SIZE, tools/pmc_collect.sh; counters cannot be read from inside the timed process), scaled to this run's average
        # volumes per launch; null if the file is absent
        traffic, traffic_src = None, None
        vox = WORKLOADS[args.workload][3] * (WORKLOADS[args.workload][1] // 4) * (WORKLOADS[args.workload][2] // 4)
        for name in ("r6_conv3d_pmc.json", "r5_conv3d_pmc.json", "r4_conv3d_pmc.json", "r3_conv3d_pmc.json", "r2_conv3d_pmc.json", "r1_conv3d_pmc.json"):
            pmc_file = os.path.join(ROOT, "profiles", name)
            if os.path.exists(pmc_file) and dom["launches"] and args.workload in ("joint", "estm") and args.conv3d_arith == "f32":
                rec = json.load(open(pmc_file))
                per_vol = rec.get("hbm_bytes_per_volume_by_algo", {}).get(kalgo, rec.get("hbm_bytes_per_volume"))
                if per_vol is None:
                    continue
                vols = dom["gflop_per_launch"] * 1e9 / (2.0 * 27 * 32 * 32 * vox)
                traffic, traffic_src = round(per_vol * vols), "profiles/" + name
                break
        # the HBM-bound kernels once more, each ALONE on the device (the in-step brackets sit on overlapped streams: they measure
        # contention with the convolutions, not the kernel)
        hbm_alone = None
        if args.workload in ("joint", "estm", "cfg5"):
            try:
                from estdepth_amd.microbench import hbm_kernels_standalone
                hbm_alone = hbm_kernels_standalone(WORKLOADS[args.workload][3], WORKLOADS[args.workload][1] // 4, WORKLOADS[args.workload][2] // 4,
                                                   n=10, device=device, peak_gbs=HBM_PEAK_GBS)
            except Exception as e:
                hbm_alone = {"error": "%s: %s" % (type(e).__name__, str(e)[:100])}
            if args.workload == "joint" and isinstance(hbm_alone, dict) and "error" not in hbm_alone:
                # ... and the fused warp + attention at the HBM stress size (cfg5: 128 x 240 x 320 voxels, 2 memory volumes): the HBM kernel
                # furthest below its roof, at the size where it weighs most
                try:
                    big = hbm_kernels_standalone(128, 240, 320, n=5, device=device, peak_gbs=HBM_PEAK_GBS, only_attention=(2,))
                    hbm_alone["warp_attention N=2 @cfg5 (128x240x320)"] = big["warp_attention N=2"]
                    torch.cuda.empty_cache()
                except Exception as e:
                    hbm_alone["warp_attention N=2 @cfg5 (128x240x320)"] = {"error": "%s: %s" % (type(e).__name__, str(e)[:100])}
        kname = {"wino3": "conv3d_wino3_kernel (3x3x3 conv 32->32, fp32 MFMA 16x16x4, all three axes in Winograd F(2,3) form: 8/27 of the products)",
                 "wino2": "conv3d_wino2_kernel (3x3x3 conv 32->32, fp32 MFMA 16x16x4, depth and row axis in Winograd F(2,3) form: 12/27 of the products)",
                 "wino": "conv3d_wino_kernel (3x3x3 conv 32->32, fp32 MFMA 16x16x4, depth axis in Winograd F(2,3) form: 18/27 of the products)",
                 "direct": "conv3d_k3_kernel<32,2> (3x3x3 conv 32->32, fp32 MFMA 16x16x4)"}[kalgo]
        line = {
            "metric": "depth frames/sec (seq_len=5, 480x640, D=64)" if args.workload == "joint" else "depth frames/sec",
            "value": round(value, 3), "unit": "depth frames/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(1e3 * elapsed / args.steps, 3), "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None,
            "dtype": "f32" if args.conv3d_arith == "f32" else "f32 (32->32 conv3d products as six bf16 MFMAs of exactly 3-way-split operands, f32 accumulate)",
            "data": "synthetic",
            "config": {"workload": WORKLOADS[args.workload][6],
                       "depth_frames_per_step": frames, "input_frames_per_step": x_imgs.shape[1],
                       "input_frames_per_s": round(x_imgs.shape[1] * world * args.steps / elapsed, 3),
                       "launch": "eager" if (args.no_graph or state["fwd"] is model) else
                                 ("hipGraph replay, stage A of step k + 1 beside stage B of step k (two lanes of captures)" if pipelined else "hipGraph replay"),
                       "pipeline": bool(pipelined and not args.no_graph),
                       "serial_replay": serial,
                       "graph_memory": None if (args.no_graph or state["fwd"] is model) else args.graph_memory,
                       "conv3d_arith": args.conv3d_arith, "conv2d_arith": args.conv2d_arith,
                       "conv3d_algo_32to32": kalgo,
                       "notes": state["notes"],
                       # the K-step figure against tens of seconds of the same step (timed / sustained: 1.0 = the headline IS a sustained rate)
                       "sustained_ms_per_step": sustained["ms_per_step"] if sustained else None,
                       "sustained_over_timed": sustained["timed_over_sustained"] if sustained else None,
                       "sustained": sustained,
                       "per_rank_ms_per_step": per_rank_ms,
                       "parallelism": "1 sequence per GPU" + ("; RCCL all-gather of {K,V,pose} per step, overlapped with the next step" if gathered else "")
                                      + (("; %d CUs left free for the collective in stage %s" % (state["reserved_cus"], "A (2D networks)" if state.get("reserve_scope") == "A" else "A and B"))
                                         if state.get("reserved_cus") else "")},
            # `achieved` / `frac` = MFMA FLOPs the dominant kernel actually EXECUTES per second against the fp32 matrix peak (<= 1 by
            # construction: the hardware fraction).  `algorithmic_*` = the direct convolution's 2*27*Cin*Cout FLOPs per voxel (SURVEY
            # §8d) over the same time -- above the peak when exact Winograd identities remove products.
            "roofline": {"bound": "mfma",
                         "kernel": kname if args.conv3d_arith == "f32" else
                                   "conv3d_k3_split_kernel (3x3x3 conv 32->32, 6 x bf16 MFMA 16x16x32 per fp32 product block; peak = bf16 dense / 6)",
                         # dominant kernel, live HIP events on its launch stream (the contract's measurement) ...
                         "achieved": dom["eager"]["achieved_tflops"], "peak": round(peak, 1), "unit": "TFLOP/s",
                         "frac": dom["eager"]["frac"],
                         "executed_factor": round(EXECUTED_FACTOR[kalgo], 4),
                         "algorithmic_tflops": dom["eager"]["algorithmic_tflops"], "algorithmic_frac": dom["eager"]["algorithmic_frac"],
                         "traffic": traffic,
                         "traffic_unit": "HBM bytes per launch (rocprofv3 FETCH_SIZE x2 + WRITE_SIZE, %s)" % traffic_src,
                         "launches": dom["launches"], "avg_launch_ms": dom["eager_bracket_ms"],
                         "launches_per_step": round(dom["launches"] / max(args.steps, 1), 2),
                         "how": "HIP events around every launch of the dominant kernel in %d eager steps after the timed loop, on its launch stream" % args.steps,
                         # ... and the same kernel inside the hipGraph replay (rocprofv3 trace of a child run): the two must agree
                         "replay": ({"avg_launch_ms": dom.get("avg_launch_ms"), "frac": dom.get("frac"), "achieved": dom.get("achieved_tflops"),
                                     "algorithmic_tflops": dom.get("algorithmic_tflops"), "eager_over_replay": dom.get("eager_over_replay")}
                                    if dom.get("source") == "replay" else None),
                         "replay_trace": replay_info,
                         # every family: `avg_launch_*` / `frac` / `achieved_*` describe the REPLAY when source == "replay" (the run that
                         # produced `value`); `eager_bracket_*` / `eager` = the HIP-event brackets of the eager pass, kept for comparison
                         "families_perturbed_by_eager_brackets": perturbed,
                         "mfma_kernels": mfma,                     # every 3x3x3 convolution instance: executed and algorithmic TFLOP/s
                         "hbm_kernels": hbm,                       # in-step (overlapped streams): GB/s of ALGORITHMIC bytes, fraction of 8 TB/s
                         "hbm_kernels_standalone": hbm_alone},     # the same kernels alone on the device
        }
        if ag is not None:
            line["config"]["allgather"] = ag
        if alt is not None:
            line["alt_arith"] = alt
        if world == 1:
            del model, fwd
            state["fwd"] = None
            torch.cuda.empty_cache()
        if world == 1 and not force_dist and not args.no_other_workloads and args.workload == "joint":
            # the other single-GPU configurations of BASELINE.json (configs[2], configs[4]) and the streaming harness, timed the same way
            # in this very run (short loops): driver-visible numbers beside the headline, not instead of it
            others = {}
            for name, (k_, w_) in (("estm", (10, 3)), ("cfg5", (5, 2))):
                try:
                    others[name] = quick_measure(name, k_, w_, device, args.no_graph, zero_copy)
                except Exception as e:
                    others[name] = {"error": "%s: %s" % (type(e).__name__, str(e)[:120])}
            try:
                sa = argparse.Namespace(steps=20, warmup=3, no_graph=args.no_graph)
                sl_ = stream_bench(sa, device, 0, 1)
                others["stream"] = {"workload": sl_["config"]["workload"], "value": sl_["value"], "unit": sl_["unit"], "ms_per_step": sl_["ms_per_step"],
                                    "steps": 20, "warmup": 3, "depth_frames_per_step": 1, "untimed_priming_calls": sl_["config"]["untimed_priming_calls"]}
                torch.cuda.empty_cache()
            except Exception as e:
                others["stream"] = {"error": "%s: %s" % (type(e).__name__, str(e)[:120])}
            try:
                others["joint_stream"] = joint_stream_bench(10, 3, device, args.no_graph)
            except Exception as e:
                others["joint_stream"] = {"error": "%s: %s" % (type(e).__name__, str(e)[:120])}
            line["other_workloads"] = others
        if world == 1 and not args.no_cpu_baseline:
            # two CPU restatements on the box's host cores -- the C/OpenMP port (the parity checker, 8 threads: it
            # does not scale further) and the torch-operator leg (oneDNN / ATen, what the reference itself would run: 8 threads for
            # comparability with BASELINE.md section 2 AND torch's default = all physical cores); headline entry = the fastest
            # round 6: + the torch-operator leg at 32 and 64 threads PINNED to one socket (the unpinned all-core leg crosses sockets and loses to 8 threads)
            legs = [("port", args.cpu_threads, False)] if args.cpu_threads > 0 else \
                [("port", 8, False), ("torch-ops", 8, False), ("torch-ops", 0, False), ("torch-ops", 32, True), ("torch-ops", 64, True)]
            if args.cpu_threads > 0:
                legs.append(("torch-ops", args.cpu_threads, False))
            if args.workload == "cfg5" and args.cpu_threads <= 0:
                legs = legs[:3]                        # (one cfg5 step is ~1 min of CPU work per leg)
            runs, parity, parity_t = [], None, None
            for kind, th, pin in legs:
                if pin and len(_socket_cpus()) < 16:
                    continue                           # (a box without a 16-core socket: the unpinned legs say it all)
                base, par = cpu_baseline(args.workload, th, x_imgs, x_poses, intr, pre_costs, pre_poses, gpu_outputs, frames, gpu_logits, kind=kind, pin_socket=pin,
                                         gpu_feats=gpu_feats)
                runs.append(base)
                if pin:
                    continue
                if kind == "port":
                    parity = par
                else:
                    parity_t = par
            best = max(runs, key=lambda b: b["value"])
            line["cpu_baseline"] = dict(best)
            line["cpu_baseline"]["all_runs"] = [{"kind": b["kind"], "cores": b["cores"], "value": b["value"], "wall_s": b["wall_s"], "pinned": b["pinned"],
                                                 **({"error": b["error"]} if "error" in b else {})} for b in runs]
            if parity_t is not None:      # the torch-operator leg's own view of the GPU depth (a second, independent CPU arithmetic)
                parity["vs_torch_ops"] = {"max_abs_depth_diff_m": parity_t["max_abs_depth_diff_vs_oracle_m"], "abs_rel": parity_t["abs_rel_vs_oracle"]}
            line["parity"] = parity
    else:
        line = None

    def emit():
        try:                               # RCCL's banner sits in the C stdio buffer of a redirected stdout: flush it so that the JSON
            import ctypes                  # line is the LAST line of the output
            ctypes.CDLL(None).fflush(None)
        except Exception:
            pass
        print(json.dumps(line), flush=True)

    def watchdog(seconds, on_timeout):
        """runs ``on_timeout`` and ends the process (exit code 0) unless the returned event is set within ``seconds``"""
        import threading
        done = threading.Event()

        def dog():
            if not done.wait(seconds):
                try:
                    on_timeout()
                finally:
                    os._exit(0)
        threading.Thread(target=dog, daemon=True).start()
        return done

    if dist_on and state.get("other_algo") and os.environ.get("ESTD_AG_DIAG", "1") != "0":
        # every rank: the OTHER exchange algorithm alone.  A hang here (an untested RCCL call pattern at a world size nobody has run)
        # must not cost the line: after ESTD_AG_DIAG_TIMEOUT seconds rank 0 prints it without the figure and every rank leaves.
        other, exchange_alone = state["other_algo"]
        limit = float(os.environ.get("ESTD_AG_DIAG_TIMEOUT", "60"))

        def gave_up():
            if rank == 0:
                line["config"]["allgather"]["other_algo"] = {"algo": other, "error": "no result within %.0f s (watchdog)" % limit}
                emit()
        done = watchdog(limit, gave_up)
        try:
            if os.environ.get("ESTD_AG_DIAG_TEST_HANG") == "1":      # (test hook of the watchdog)
                time.sleep(1e6)
            with torch.no_grad():
                t_other = exchange_alone(other)
            nb = ag["bytes_sent_per_rank"]
            res = {"algo": other, "ms_alone": round(1e3 * t_other, 3), "bus_gbs_per_rank": round((world - 1) * nb / t_other / 1e9, 2)}
        except Exception as e:
            res = {"algo": other, "error": "%s: %s" % (type(e).__name__, str(e)[:80])}
        # ESTD_AG_ALGO=auto (default): every rank has both figures now -- agree on them (MAX over the ranks) and, when the other algorithm
        # wins the exchange alone by more than 10 %, it is the CHOSEN algorithm: the K timed steps run once more on it (still under the
        # watchdog: the line as it stands is printed if this never returns) and THAT pass is the line's value -- whichever way the
        # step time moved (no minimum over the two passes); the first pass is reported beside it
        if parallel.AG_ALGO == "auto" and "ms_alone" in res and world > 1 and os.environ.get("ESTD_AG_AUTO_RETIME", "1") != "0":
            try:
                tx = torch.tensor([ag["ms_alone"], res["ms_alone"]], device=device, dtype=torch.float64)
                dist.all_reduce(tx, op=dist.ReduceOp.MAX)
                switch = float(tx[1]) < 0.9 * float(tx[0])
                auto = {"ms_alone": {used_algo: round(float(tx[0]), 3), other: round(float(tx[1]), 3)}, "chosen": other if switch else used_algo}
                if switch:
                    parallel.set_active_algo(other)
                    state["allgather"] = True
                    t2 = timed_plain()
                    state["allgather"] = False
                    auto["first_pass"] = {"algo": used_algo, "ms_per_step": round(1e3 * elapsed / args.steps, 3)}
                    auto["second_pass"] = {"algo": other, "ms_per_step": round(1e3 * t2 / args.steps, 3)}
                    if rank == 0:
                        line["value"] = round(frames * world * args.steps / t2, 3)
                        line["ms_per_step"] = round(1e3 * t2 / args.steps, 3)
                        line["config"]["input_frames_per_s"] = round(x_imgs.shape[1] * world * args.steps / t2, 3)
                        line["config"]["allgather"]["algo"] = other
                        line["config"]["allgather"]["ms_per_step_with_collective"] = round(1e3 * t2 / args.steps, 3)
                if rank == 0:
                    line["config"]["allgather"]["auto"] = auto
            except Exception as e:
                if rank == 0:
                    line["config"]["allgather"]["auto"] = {"error": "%s: %s" % (type(e).__name__, str(e)[:80])}
        done.set()
        if rank == 0:
            line["config"]["allgather"]["other_algo"] = res
    if rank == 0:
        emit()
    if dist_on:
        done = watchdog(30.0, lambda: None)       # the line is out: a rank that never arrives must not keep the others (and the launcher) waiting
        dist.barrier()
        dist.destroy_process_group()
        done.set()


if __name__ == "__main__":
    main()
