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
_XO = re.compile(r"conv3d_xout_kernel")              # output channel 32 of dres2 as a pass of its own (csrc/conv3d_xout.hip, round 6)
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
    if _XO.search(n):
        return "conv3d:33->1"
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


# ------------------------------------------------------------------------------------------------------------------------------
# Shader clock / board power beside a long timed loop (bench.py --sustained-s): a host thread that reads the amdgpu sysfs nodes of the
# device (no process spawn, no HIP call: nothing enters the timed stream), falling back to one `rocm-smi` / `amd-smi` call per sample.
class GpuSampler:
    """samples (t, sclk MHz, power W) every ``period`` seconds between start() and stop(); summary() -> dict or {"error": ...}"""

    def __init__(self, device_index=0, period=0.25):
        import threading
        self.period = period
        self.samples = []
        self._stop = threading.Event()
        self._thread = None
        self.source = None
        self._card = self._find_card(device_index)

    @staticmethod
    def _find_card(index):
        """sysfs node of HIP device ``index``: by its PCI address (a container may see the sysfs nodes of every GPU of the node while HIP
        sees one: the n-th DRM card is then somebody else's GPU), else the n-th amdgpu card"""
        import glob
        import os
        try:
            import torch
            pr = torch.cuda.get_device_properties(index)
            dev = "/sys/bus/pci/devices/%04x:%02x:%02x.0" % (pr.pci_domain_id, pr.pci_bus_id, pr.pci_device_id)
            if os.path.exists(os.path.join(dev, "pp_dpm_sclk")) or glob.glob(os.path.join(dev, "hwmon", "hwmon*")):
                return dev
        except Exception:
            pass
        cards = []
        for c in sorted(glob.glob("/sys/class/drm/card[0-9]*")):
            dev = os.path.join(c, "device")
            if os.path.basename(c).count("-") == 0 and os.path.exists(os.path.join(dev, "pp_dpm_sclk")):
                cards.append(dev)
        return cards[index] if index < len(cards) else (cards[0] if cards else None)

    def _read_sysfs(self):
        import glob
        import os
        if self._card is None:
            return None
        sclk, power = None, None
        try:
            for line in open(os.path.join(self._card, "pp_dpm_sclk")):
                if line.rstrip().endswith("*"):
                    sclk = float(re.search(r"(\d+)\s*Mhz", line, re.I).group(1))
        except (OSError, AttributeError):
            pass
        for hw in glob.glob(os.path.join(self._card, "hwmon", "hwmon*")):
            if sclk is None:
                try:
                    sclk = float(open(os.path.join(hw, "freq1_input")).read()) / 1e6
                except (OSError, ValueError):
                    pass
            for node in ("power1_average", "power1_input"):
                try:
                    power = float(open(os.path.join(hw, node)).read()) / 1e6
                    break
                except (OSError, ValueError):
                    pass
        if sclk is None and power is None:
            return None
        return sclk, power

    def _read_smi(self):
        import shutil
        import subprocess
        exe = shutil.which("rocm-smi") or "/opt/rocm/bin/rocm-smi"
        try:
            r = subprocess.run([exe, "--showclocks", "--showpower"], capture_output=True, text=True, timeout=10).stdout
        except Exception:
            return None
        m = re.search(r"GPU\[\d+\]\s*:\s*sclk clock level[^(]*\((\d+)Mhz\)", r)
        p = re.search(r"GPU\[\d+\]\s*:\s*[^\n]*Power \(W\):\s*([\d.]+)", r)
        if not m and not p:
            return None
        return (float(m.group(1)) if m else None), (float(p.group(1)) if p else None)

    def _run(self):
        import time
        read = self._read_sysfs
        if read() is None:
            read, self.source = self._read_smi, "rocm-smi --showclocks --showpower (one call per sample)"
            self.period = max(self.period, 1.0)
        else:
            self.source = "amdgpu sysfs (%s: pp_dpm_sclk / hwmon freq1_input, power1_average)" % self._card
        t0 = time.perf_counter()
        while not self._stop.is_set():
            s = read()
            if s is not None:
                self.samples.append((time.perf_counter() - t0, s[0], s[1]))
            self._stop.wait(self.period)

    def start(self):
        import threading
        self._thread = threading.Thread(target=self._run, daemon=True)
        self._thread.start()
        return self

    def stop(self):
        self._stop.set()
        if self._thread is not None:
            self._thread.join(timeout=15)
        return self

    def summary(self, bucket_edges=None):
        if not self.samples:
            return {"error": "no shader-clock / power reading (no amdgpu sysfs node and no rocm-smi output)", "source": self.source}

        def stat(vals):
            vals = [v for v in vals if v is not None]
            if not vals:
                return None
            return {"min": round(min(vals), 1), "mean": round(sum(vals) / len(vals), 1), "max": round(max(vals), 1)}
        out = {"source": self.source, "samples": len(self.samples), "period_s": self.period,
               "sclk_mhz": stat([s[1] for s in self.samples]), "power_w": stat([s[2] for s in self.samples])}
        if bucket_edges:
            per = []
            for a, b in zip(bucket_edges[:-1], bucket_edges[1:]):
                inb = [s for s in self.samples if a <= s[0] < b]
                sc = [s[1] for s in inb if s[1] is not None]
                pw = [s[2] for s in inb if s[2] is not None]
                per.append([round(sum(sc) / len(sc)) if sc else None, round(sum(pw) / len(pw)) if pw else None])
            out["per_bucket_sclk_mhz_power_w"] = per
        return out
