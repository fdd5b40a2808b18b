"""Kernel sequence (start, end, queue, name) inside a time window of the LAST timed step of bench.py (rocprofv3 --kernel-trace CSV).
    python tools/prof_seq.py <kernel_trace.csv> <t0_ms> <t1_ms> [steps]"""
import csv, sys
src, t0w, t1w = sys.argv[1], float(sys.argv[2]), float(sys.argv[3])
rows = []
with open(src) as f:
    for r in csv.DictReader(f):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"], r.get("Queue_Id", "")))
rows.sort()
marks = [i for i, r in enumerate(rows) if "estd_mark_kernel" in r[2]]
lo, hi = marks[-2], marks[-1]
steps = int(sys.argv[4]) if len(sys.argv) > 4 else 5
span = (rows[hi][0] - rows[lo][1]) / steps
base = rows[hi][0] - span
for s, e, n, q in rows[lo + 1:hi]:
    a, b = (s - base) / 1e6, (e - base) / 1e6
    if b < t0w or a > t1w:
        continue
    n = n.replace("void (anonymous namespace)::", "").replace("void ", "")
    print("q%s %+8.3f %+8.3f %7.3f  %s" % (q, a, b, b - a, n[:70]))
