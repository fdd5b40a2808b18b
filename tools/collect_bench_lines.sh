#!/bin/bash
# The bench.py JSON lines of a round (default command per workload + the A/B lines), without the traces / PMC passes of tools/collect_profiles.sh.
#   bash tools/collect_bench_lines.sh [round prefix, default r6]
set -u
R=$PWD
P=${1:-r6}
OUT=$R/gpurun_out/profiles_$P
mkdir -p $OUT
last() { grep "^{" | tail -1; }
python bench.py 2>/dev/null | last > $OUT/${P}_bench_joint.json
python bench.py --workload estm 2>/dev/null | last > $OUT/${P}_bench_estm.json
python bench.py --workload cfg1 2>/dev/null | last > $OUT/${P}_bench_cfg1.json
python bench.py --workload cfg5 --steps 5 --warmup 2 2>/dev/null | last > $OUT/${P}_bench_cfg5.json
python bench.py --workload stream --steps 20 2>/dev/null | last > $OUT/${P}_bench_stream.json
AB="--no-cpu-baseline --no-alt --no-replay-profile --no-other-workloads --sustained-s 0 --steps 40 --warmup 5"
for rep in 1 2; do
  python bench.py $AB 2>/dev/null | last > $OUT/${P}_ab_default_$rep.json
  ESTD_W3=0 python bench.py $AB 2>/dev/null | last > $OUT/${P}_ab_two_axis_$rep.json      # 32 -> 32 convolutions on the two-axis kernel (csrc/conv3d_wino2.hip)
  ESTD_GATE_IN_CONV=0 python bench.py $AB 2>/dev/null | last > $OUT/${P}_ab_gate_pass_$rep.json
  python bench.py $AB --pipeline on 2>/dev/null | last > $OUT/${P}_ab_pipeline_$rep.json                 # stage A of step k + 1 beside the second half of stage B of step k (+ the serial A/B of the same process)
  ESTD_C1X1_LDS=0 python bench.py $AB 2>/dev/null | last > $OUT/${P}_ab_conv1x1_direct_$rep.json          # 1x1 convolutions on the direct form only
  ESTD_W3_XOUT=0 python bench.py $AB 2>/dev/null | last > $OUT/${P}_ab_dres2_one_launch_$rep.json          # dres2 on the two-axis kernel's 33 -> 33 instance
done
ESTD_FORCE_DIST=1 python bench.py --no-cpu-baseline --no-alt --no-other-workloads 2>/dev/null | last > $OUT/${P}_bench_joint_rccl_world1.json
ESTD_FORCE_DIST=1 ESTD_RESERVE_SCOPE=AB python bench.py --no-cpu-baseline --no-alt --no-other-workloads 2>/dev/null | last > $OUT/${P}_bench_joint_rccl_world1_reserve_ab.json
ESTD_FORCE_DIST=1 python bench.py --workload estm --no-cpu-baseline --no-alt --no-other-workloads 2>/dev/null | last > $OUT/${P}_bench_estm_rccl_world1.json
python bench.py --gpus 2 --workload cfg1 --steps 5 --warmup 2 2>/dev/null | last > $OUT/${P}_bench_gpus2_codepath.json
timeout 2400 python -m pytest tests/ -q -m gpu > $OUT/${P}_gputests.log 2>&1
ls -la $OUT | tail -20
