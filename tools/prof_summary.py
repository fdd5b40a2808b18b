"""Cut a rocprofv3 --kernel-trace CSV to bench.py's timed region (between the two estd_mark_kernel launches)
and print/save per-kernel statistics.   python tools/prof_summary.py <kernel_trace.csv> <out.csv> [steps]"""
import csv
import sys
from collections import defaultdict

src, dst = sys.argv[1], sys.argv[2]
steps = int(sys.argv[3]) if len(sys.argv) > 3 else 1
rows = []
with open(src) as f:
    for r in csv.DictReader(f):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]))
rows.sort()
marks = [i for i, r in enumerate(rows) if "estd_mark_kernel" in r[2]]
assert len(marks) >= 2, "no estd_mark_kernel pair in trace"
lo, hi = marks[-2], marks[-1]
region = rows[lo + 1:hi]
span = rows[hi][0] - rows[lo][1]
agg = defaultdict(lambda: [0, 0])
busy = 0
for s, e, n in region:
    agg[n][0] += 1
    agg[n][1] += e - s
    busy += e - s
with open(dst, "w") as f:
    w = csv.writer(f)
    w.writerow(["kernel", "calls_per_step", "avg_us", "total_ms_per_step", "pct_of_busy"])
    w.writerow(["#timed_region", steps, "", round(span / 1e6 / steps, 3), "gpu_busy_pct=%.1f" % (100.0 * busy / span)])
    for n, (c, t) in sorted(agg.items(), key=lambda kv: -kv[1][1]):
        w.writerow([n[:160], round(c / steps, 2), round(t / c / 1e3, 2), round(t / 1e6 / steps, 4), round(100.0 * t / busy, 2)])
print("timed region: %.3f ms/step, GPU busy %.1f %%, %d kernels/step" % (span / 1e6 / steps, 100.0 * busy / span, len(region) // steps))
