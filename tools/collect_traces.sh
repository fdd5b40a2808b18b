#!/bin/bash
# The rocprofv3 kernel traces of the bench workloads only (the first block of tools/collect_profiles.sh): kernel statistics, timeline, launch sequence of the timed region.
#   bash tools/collect_traces.sh [round prefix, default r5]
set -u
R=$PWD
P=${1:-r5}
OUT=$R/gpurun_out/profiles_$P
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
for wl in joint estm cfg5; do
  rm -rf /tmp/prof_$wl
  rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_$wl -o p -- python $R/bench.py --workload $wl --steps 5 --warmup 2 --no-cpu-baseline --no-alt --no-replay-profile --no-other-workloads > $OUT/${P}_bench_${wl}_profiled.json 2> /dev/null
  T=$(find /tmp/prof_$wl -name "p_kernel_trace.csv" | head -1)
  python $R/tools/prof_summary.py $T $OUT/${P}_bench_${wl}_kernel_stats.csv 5
  python $R/tools/prof_timeline.py $T 5 --gaps > $OUT/${P}_bench_${wl}_timeline.txt 2>&1
  python $R/tools/prof_seq.py $T -1 60 5 > $OUT/${P}_bench_${wl}_seq.txt 2>&1
  S=$(find /tmp/prof_$wl -name "p_kernel_stats.csv" | head -1)
  [ -n "$S" ] && head -41 $S > $OUT/${P}_bench_${wl}_rocprof_stats_top40.csv
done
head -14 $OUT/${P}_bench_joint_timeline.txt
