#!/bin/bash
# One traced run of the headline workload on the GPU box: bench line + per-kernel stats + timeline + kernel sequence of the last step.
#   bash tools/trace_step.sh <tag> [workload] [t0_ms] [t1_ms]     -> gpurun_out/<tag>_{bench.json,kernel_stats.csv,timeline.txt,seq.txt}
set -u
R=$PWD
TAG=${1:-trace}; WL=${2:-joint}; T0=${3:--1.0}; T1=${4:-25.0}
mkdir -p $R/gpurun_out
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/prof_$TAG
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_$TAG -o p -- python $R/bench.py --workload $WL --steps 5 --warmup 2 --no-cpu-baseline --no-alt --no-replay-profile --no-other-workloads > $R/gpurun_out/${TAG}_bench.json 2> $R/gpurun_out/${TAG}_bench.err
T=$(find /tmp/prof_$TAG -name "p_kernel_trace.csv" | head -1)
python $R/tools/prof_summary.py $T $R/gpurun_out/${TAG}_kernel_stats.csv 5
python $R/tools/prof_timeline.py $T 5 --gaps > $R/gpurun_out/${TAG}_timeline.txt 2>&1
python $R/tools/prof_seq.py $T $T0 $T1 5 > $R/gpurun_out/${TAG}_seq.txt 2>&1
cd $R
