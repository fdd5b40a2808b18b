#!/bin/bash
# Separate rocprofv3 --pmc passes (never combined with sys/hip traces) over one command; prints per-kernel averages.
#   tools/pmc_collect.sh "<counters...>" <out.csv> -- <command...>
cd /tmp && export TMPDIR=/tmp
CTRS="$1"; OUT="$2"; shift 3
rm -rf /tmp/pmc_run; mkdir -p /tmp/pmc_run
for c in $CTRS; do
  rocprofv3 --kernel-trace --pmc $c --output-format csv -d /tmp/pmc_run/$c -o p -- "$@" > /tmp/pmc_run/$c.log 2>&1
done
python - "$OUT" $CTRS <<'PY'
import csv, glob, sys, collections
out, ctrs = sys.argv[1], sys.argv[2:]
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for c in ctrs:
    for f in glob.glob("/tmp/pmc_run/%s/**/p_counter_collection.csv" % c, recursive=True):
        for r in csv.DictReader(open(f)):
            if r["Counter_Name"] == c:
                agg[r["Kernel_Name"]][c].append(float(r["Counter_Value"]))
with open(out, "w") as f:
    w = csv.writer(f)
    w.writerow(["kernel", "launches"] + ["avg_" + c for c in ctrs])
    for k, d in sorted(agg.items()):
        n = max(len(v) for v in d.values())
        w.writerow([k[:120], n] + [round(sum(d[c]) / max(len(d[c]), 1), 2) if c in d else "" for c in ctrs])
print(open(out).read()[:6000])
PY
