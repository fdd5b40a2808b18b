"""Per-queue kernel-family totals inside a time window of ONE step of bench.py's timed region (rocprofv3 --kernel-trace CSV).
    python tools/prof_phase.py <kernel_trace.csv> <t0_ms> <t1_ms>      (times relative to the start of the LAST timed step)"""
import csv, sys
from collections import defaultdict
src, t0w, t1w = sys.argv[1], float(sys.argv[2]), float(sys.argv[3])
rows = []
with open(src) as f:
    for r in csv.DictReader(f):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"], r.get("Queue_Id", "")))
rows.sort()
marks = [i for i, r in enumerate(rows) if "estd_mark_kernel" in r[2]]
lo, hi = marks[-2], marks[-1]
region = rows[lo + 1:hi]
steps = int(sys.argv[4]) if len(sys.argv) > 4 else 5
span = (rows[hi][0] - rows[lo][1]) / steps
base = rows[hi][0] - span                      # start of the last step
fam = lambda n: next((k for k in ("conv3d_wino2", "conv3d_wino", "conv2d_wino", "conv2d_small", "conv2d_k3", "conv3d_k3", "bn_act", "warp_attention", "homo_warp", "gru_", "stem3x3", "spp_", "Cijk", "igemm", "xdl", "pool", "elementwise", "copyBuffer", "normalise", "planes_cat", "upsample2", "disp_head") if k in n), n[:30])
agg = defaultdict(lambda: defaultdict(lambda: [0, 0.0]))
ends = defaultdict(float)
for s, e, n, q in region:
    a, b = (s - base) / 1e6, (e - base) / 1e6
    if b < t0w or a > t1w:
        continue
    agg[q][fam(n)][0] += 1
    agg[q][fam(n)][1] += (e - s) / 1e6
    ends[q] = max(ends[q], b)
for q in sorted(agg):
    tot = sum(v[1] for v in agg[q].values())
    print("queue %s: %.2f ms of kernels, last end +%.2f ms" % (q, tot, ends[q]))
    for k, (c, t) in sorted(agg[q].items(), key=lambda kv: -kv[1][1]):
        print("    %-18s %3d launches %6.3f ms" % (k, c, t))
