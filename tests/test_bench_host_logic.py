"""CPU: host-side pieces of bench.py that need no GPU -- the socket topology reader, the pinned CPU-baseline leg (a child process confined to
the physical cores of one socket before its first thread exists: bench.cpu_leg_pinned / cpu_leg_child), and the shader-clock / power
sampler's behaviour on a box without an amdgpu device."""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def test_socket_cpus_are_allowed_physical_cores():
    import bench
    cpus = bench._socket_cpus()
    allowed = os.sched_getaffinity(0)
    assert cpus and set(cpus) <= allowed and len(set(cpus)) == len(cpus) and cpus == sorted(cpus)
    # one hardware thread per physical core: no two entries share a core
    seen = set()
    for c in cpus:
        d = "/sys/devices/system/cpu/cpu%d/topology/" % c
        key = (open(d + "physical_package_id").read().strip(), open(d + "core_id").read().strip())
        assert key not in seen
        seen.add(key)
    assert len({k[0] for k in seen}) == 1                       # ONE socket


def test_pinned_cpu_leg_runs_in_a_confined_child_and_matches_the_in_process_oracle():
    """cfg1 size (3 frames, 128x160, D = 16, ResNet-18, no memory): the child reports its wall time, its thread count and affinity; the depth it
    computed equals the in-process oracle's (same composition, same inputs)."""
    import numpy as np
    import bench
    from estdepth_amd import synth
    from oracle import ref_model as M, ref_ops as O
    from oracle.nets2d import Nets2D, sd_numpy
    imgs, poses, intr, _ = synth.make_sequence(3, 128, 160, seed=1000)
    n = min(2, len(bench._socket_cpus()))
    for kind in ("torch-ops", "port"):
        r = bench.cpu_leg_pinned("cfg1", n, imgs, poses, intr, None, None, 1, kind)
        assert "error" not in r, r
        assert r["kind"] == kind and r["cores"] == n and r["value"] > 0 and r["wall_s"] > 0
        assert "child process, %d threads" % n in r["pinned"] and "torch reports %d threads" % n in r["pinned"]
    # the child's arithmetic is the oracle's: run its entry point in this process on the same file and compare with model_forward here
    import argparse
    import io
    import json
    import tempfile
    from contextlib import redirect_stdout
    with tempfile.TemporaryDirectory() as tmp:
        path = os.path.join(tmp, "step.npz")
        np.savez(path, imgs=imgs.numpy(), poses=poses.numpy(), intr=intr.numpy())
        buf = io.StringIO()
        old = os.environ.pop("ESTD_CPU_LEG_CPUS", None)
        try:
            with redirect_stdout(buf):
                bench.cpu_leg_child(argparse.Namespace(cpu_leg_child=path, cpu_leg_kind="port", workload="cfg1", cpu_threads=2))
        finally:
            if old is not None:
                os.environ["ESTD_CPU_LEG_CPUS"] = old
        child = json.loads([l for l in buf.getvalue().splitlines() if l.startswith("{")][-1])
    O.set_num_threads(2)
    cpu_model = bench.build_model("cfg1", "cpu")
    ref, _, _ = M.model_forward(sd_numpy(cpu_model), imgs.numpy(), poses.numpy(), intr.numpy(), None, None, Nets2D(model=cpu_model),
                                ndepths=16, depth_min=0.1, depth_max=10.0, IF_EST_transformer=False)
    want = float(np.asarray(ref[("depth", 0, 0)], np.float64).sum())
    assert abs(child["depth0_checksum"] - want) <= 1e-6 * abs(want), (child, want)
    torch.set_num_threads(max(1, (os.cpu_count() or 2)))


def test_gpu_sampler_without_a_device_reports_an_error_not_a_number():
    import time
    from estdepth_amd.profiling import GpuSampler
    s = GpuSampler(0, period=0.05).start()
    time.sleep(0.2)
    out = s.stop().summary([0.0, 0.1, 0.2])
    assert ("error" in out) or (out["samples"] >= 1 and out["sclk_mhz"] is not None or out["power_w"] is not None)
