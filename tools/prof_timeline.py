"""Timeline of ONE step of bench.py's timed region from a rocprofv3 --kernel-trace CSV: union-busy time, idle gaps and,
per kernel family, the time it runs ALONE (nothing else on the GPU) vs overlapped.
    python tools/prof_timeline.py <kernel_trace.csv> [steps]"""
import csv
import sys
from collections import defaultdict

src = sys.argv[1]
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 1
rows = []
with open(src) as f:
    for r in csv.DictReader(f):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"], r.get("Queue_Id", ""), r.get("Stream_Id", "")))
rows.sort()
marks = [i for i, r in enumerate(rows) if "estd_mark_kernel" in r[2]]
lo, hi = marks[-2], marks[-1]
t0, t1 = rows[lo][1], rows[hi][0]
region = rows[lo + 1:hi]
span = t1 - t0


def fam(n):
    for key in ("conv3d_wino", "conv2d_wino", "conv3d_k3_split", "conv3d_k3_kernel", "conv2d_k3", "bn_act_nhwc", "warp_attention", "homo_warp", "gru_", "softargmin", "groupnorm",
                "BatchNorm", "igemm", "xdl", "Cijk", "elementwise", "avg_pool", "upsample", "CatArray", "mix1x1", "cam_"):
        if key in n:
            return key
    return "other"


# sweep: events
ev = []
for s, e, n, q, st in region:
    ev.append((s, 1, n)); ev.append((e, -1, n))
ev.sort()
active = defaultdict(int)
nact = 0
last = t0
idle = 0
alone = defaultdict(int)
shared = defaultdict(int)
for t, d, n in ev:
    dt = t - last
    if dt > 0:
        if nact == 0:
            idle += dt
        elif nact == 1:
            k = [a for a, c in active.items() if c > 0][0]
            alone[fam(k)] += dt
        else:
            for a, c in active.items():
                if c > 0:
                    shared[fam(a)] += dt
    active[n] += d
    nact += d
    last = t
idle += t1 - last
print("span %.3f ms/step ; idle %.3f ms/step (%.1f %%)" % (span / 1e6 / steps, idle / 1e6 / steps, 100.0 * idle / span))
print("%-22s %10s %10s" % ("family", "alone ms", "shared ms"))
for k in sorted(set(alone) | set(shared), key=lambda k: -(alone[k] + shared[k])):
    print("%-22s %10.3f %10.3f" % (k, alone[k] / 1e6 / steps, shared[k] / 1e6 / steps))
qs = defaultdict(int)
for s, e, n, q, st in region:
    qs[(q, st)] += e - s
print("per (queue, stream) busy ms/step:", {k: round(v / 1e6 / steps, 2) for k, v in qs.items()})

# ---- per-queue gaps of the LAST step: where does each stream wait? ----
if "--gaps" in sys.argv:
    step_len = span // steps
    s0 = t1 - step_len
    last = [r for r in region if r[0] >= s0]
    byq = defaultdict(list)
    for s, e, n, q, st in last:
        byq[q].append((s, e, n))
    for q, ks in byq.items():
        ks.sort()
        busy = sum(e - s for s, e, _ in ks)
        print("queue %s: %d kernels, busy %.2f ms, first start +%.2f ms, last end +%.2f ms" % (q, len(ks), busy / 1e6, (ks[0][0] - s0) / 1e6, (ks[-1][1] - s0) / 1e6))
        prev_e, prev_n = ks[0][1], ks[0][2]
        for s, e, n in ks[1:]:
            if s - prev_e > 100000:
                print("    gap %.2f ms at +%.2f ms: after %s -> before %s" % ((s - prev_e) / 1e6, (prev_e - s0) / 1e6, prev_n[:60], n[:60]))
            if e > prev_e:
                prev_e, prev_n = e, n

# ---- every kernel of the LAST step in start order: +start ms, duration us, queue, name ----
if "--seq" in sys.argv:
    import re
    step_len = span // steps
    s0 = t1 - step_len
    for s, e, n, q, st in region:
        if s >= s0:
            short = re.sub(r"\(anonymous namespace\)::|void |at::native::", "", n)[:70]
            print("+%7.3f %8.1f us  q%s  %s" % ((s - s0) / 1e6, (e - s) / 1e3, q, short))
