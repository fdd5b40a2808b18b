"""Kernel-trace bookkeeping shared by bench.py and tools/prof_*.py: which hot-path family a kernel of a rocprofv3
``--kernel-trace`` CSV belongs to, and the per-family averages of bench.py's timed region (between the two
``estd_mark_kernel`` launches) -- i.e. of the hipGraph REPLAY that produced the benchmark's ``value``.

HIP events cannot bracket nodes inside a graph replay, and an eager pass with an event pair around every launch perturbs what
overlaps with what (the step is host-bound in eager mode: the side-stream kernels meet different neighbours).  The trace of the
replay has neither problem.  Measurement infrastructure only: nothing on the product path imports this module.
"""
import csv
import re
from collections import defaultdict

# family names are ops._Prof's group names (estdepth_amd/ops.py), so that amounts (FLOPs / bytes per launch) recorded there apply
_W2 = re.compile(r"conv3d_wino2_kernel<\s*(\d+),\s*(\d+|true|false),\s*(true|false),\s*(true|false)(?:,\s*(true|false))?(?:,\s*(true|false))?\s*>")
_W2X = re.compile(r"conv3d_wino2x_kernel<")          # the operand-reuse form of the 32 -> 32 instance (csrc/conv3d_wino2x.hip, opt-in)
_W3 = re.compile(r"conv3d_wino3_kernel<\s*(\d+),\s*(true|false),\s*(true|false)\s*>")            # the 32 -> 32 instances with all three axes in Winograd form (csrc/conv3d_wino3.hip, default since round 5)
_W2H = re.compile(r"conv3d_wino2_c16_kernel")
_W1 = re.compile(r"conv3d_wino_kernel<\s*(true|false),\s*(true|false)\s*>")
_K3 = re.compile(r"conv3d_k3_kernel<\s*(\d+),\s*(\d+),\s*(true|false),\s*(true|false)\s*>")


def family_of(kernel_name):
    """hot-path family of a kernel name of the trace, or None (2D networks, library kernels, copies)"""
    n = kernel_name
    m = _W2.search(n)
    if m:
        extra, o16 = m.group(3) == "true", m.group(4) == "true"
        xout = m.group(5) == "true"
        if o16:
            return "conv3d:32->16"
        if extra:
            return "conv3d:33->33" if xout else "conv3d:33->32"
        return "conv3d:32->32"
    m = _W3.search(n)
    if m:
        return "conv3d:33->32" if m.group(3) == "true" else "conv3d:32->32"      # <read-back kind, GroupNorm partials, scalar 33rd input channel>
    if _W2X.search(n):
        return "conv3d:32->32"
    if _W2H.search(n):
        return "conv3d:16->16"
    m = _W1.search(n)
    if m:
        extra, xout = m.group(1) == "true", m.group(2) == "true"
        return "conv3d:33->33" if xout else ("conv3d:33->32" if extra else "conv3d:32->32")
    m = _K3.search(n)
    if m:
        cm, nt, extra, xout = int(m.group(1)), int(m.group(2)), m.group(3) == "true", m.group(4) == "true"
        if cm == 16:
            return "conv3d:16->16"
        if xout:
            return "conv3d:33->33"
        if extra:
            return "conv3d:33->32"
        return "conv3d:32->%d" % (16 * nt)
    if "conv3d_k3_split_kernel" in n:
        return "conv3d:32->32"
    for key, fam in (("homo_warp_costvol_kernel", "homo_warp_costvol"), ("warp_attention_kernel", "warp_attention"),
                     ("softargmin_up_kernel", "softargmin"), ("gru_reset_kernel", "gru_elementwise"), ("gru_blend_kernel", "gru_elementwise")):
        if key in n:
            return fam
    return None


def read_trace(path):
    rows = []
    with open(path) as f:
        for r in csv.DictReader(f):
            rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]))
    rows.sort()
    return rows


def timed_region(rows):
    """kernels between the LAST pair of estd_mark_kernel launches (bench.py brackets its timed loop with them)"""
    marks = [i for i, r in enumerate(rows) if "estd_mark_kernel" in r[2]]
    if len(marks) < 2:
        raise RuntimeError("no estd_mark_kernel pair in the trace")
    lo, hi = marks[-2], marks[-1]
    return rows[lo + 1:hi], rows[hi][0] - rows[lo][1]


def replay_families(trace_csv, steps):
    """-> ({family: {"launches_per_step", "avg_launch_ms", "total_ms_per_step"}}, {"ms_per_step", "gpu_busy_pct", "kernels_per_step"})"""
    region, span = timed_region(read_trace(trace_csv))
    agg = defaultdict(lambda: [0, 0])
    busy = 0
    for s, e, n in region:
        busy += e - s
        fam = family_of(n)
        if fam is not None:
            agg[fam][0] += 1
            agg[fam][1] += e - s
    fams = {k: {"launches_per_step": round(c / steps, 2), "avg_launch_ms": round(t / c / 1e6, 4), "total_ms_per_step": round(t / 1e6 / steps, 4)}
            for k, (c, t) in sorted(agg.items())}
    info = {"ms_per_step": round(span / 1e6 / steps, 3), "gpu_busy_pct": round(100.0 * busy / span, 1), "kernels_per_step": len(region) // max(steps, 1)}
    return fams, info
