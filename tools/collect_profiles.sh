#!/bin/bash
# Round-2 profile collection on the GPU box (writes under gpurun_out/profiles_r2/; copy what is kept into profiles/).
#   bash tools/collect_profiles.sh
set -u
R=$PWD
OUT=$R/gpurun_out/profiles_r2
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
for wl in joint estm cfg5; do
  rm -rf /tmp/prof_$wl
  rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_$wl -o p -- python $R/bench.py --workload $wl --steps 5 --warmup 2 --no-cpu-baseline --no-alt > $OUT/r2_bench_${wl}_profiled.json 2> $OUT/r2_bench_${wl}_profiled.err
  T=$(find /tmp/prof_$wl -name "p_kernel_trace.csv" | head -1)
  python $R/tools/prof_summary.py $T $OUT/r2_bench_${wl}_kernel_stats.csv 5
  python $R/tools/prof_timeline.py $T 5 --gaps > $OUT/r2_bench_${wl}_timeline.txt 2>&1
  S=$(find /tmp/prof_$wl -name "p_kernel_stats.csv" | head -1)
  [ -n "$S" ] && head -41 $S > $OUT/r2_bench_${wl}_rocprof_stats_top40.csv
done
# PMC: dominant kernel (conv_bench: N = 3 volumes) and the HBM-bound kernels, separate --pmc passes
bash $R/tools/pmc_collect.sh "FETCH_SIZE WRITE_SIZE" $OUT/r2_conv3d_wino_pmc.csv -- python $R/tools/conv_bench.py 3 10 > /dev/null 2>&1
ESTD_CONV3D_ALGO=direct bash $R/tools/pmc_collect.sh "FETCH_SIZE WRITE_SIZE" $OUT/r2_conv3d_direct_pmc.csv -- python $R/tools/conv_bench.py 3 10 > /dev/null 2>&1
bash $R/tools/pmc_collect.sh "FETCH_SIZE WRITE_SIZE" $OUT/r2_hbm_kernels_pmc.csv -- python $R/tools/hbm_bench.py > /dev/null 2>&1
# the hardware's own matrix-pipe utilisation counter (and LDS bank conflicts) of every convolution kernel, stand-alone benches
for b in "conv_bench.py 3 10" "kv_bench.py" "head_bench.py" "conv2d_bench.py"; do
  bash $R/tools/pmc_collect.sh "MfmaUtil SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE" /tmp/mfma_one.csv -- python $R/tools/$b > /dev/null 2>&1
  echo "# python tools/$b" >> $OUT/r2_mfma_util_pmc.csv; grep -v "at::native\|rocclr" /tmp/mfma_one.csv >> $OUT/r2_mfma_util_pmc.csv
done
cd $R
python tools/hbm_bench.py > $OUT/r2_hbm_bench.txt 2>&1
CB_EPI=1 python tools/conv_bench.py 3 30 > $OUT/r2_conv_bench.txt 2>&1
python tools/conv_bench.py 1 30 >> $OUT/r2_conv_bench.txt 2>&1
CB_EPI=1 ESTD_CONV3D_ALGO=direct python tools/conv_bench.py 3 30 >> $OUT/r2_conv_bench.txt 2>&1
ESTD_CONV3D_ALGO=direct python tools/conv_bench.py 1 30 >> $OUT/r2_conv_bench.txt 2>&1
python tools/head_bench.py >> $OUT/r2_conv_bench.txt 2>&1
python tools/kv_bench.py >> $OUT/r2_conv_bench.txt 2>&1
python tools/conv2d_bench.py > $OUT/r2_conv2d_bench.txt 2>&1
# eager host overhead: torch custom ops vs raw ctypes binding (same kernels, ~330 launches per step)
for b in torch ctypes; do
  ESTD_BINDING=$b python bench.py --no-graph --steps 10 --warmup 3 --no-cpu-baseline --no-alt 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('eager binding=$b', d['value'], 'depth frames/s', d['ms_per_step'], 'ms/step')" >> $OUT/r2_eager_binding_overhead.txt
done
# default bench lines (with cpu_baseline + parity) of every workload, and the 8-thread CPU baseline
python bench.py > $OUT/r2_bench_joint.json 2>/dev/null
python bench.py --workload estm > $OUT/r2_bench_estm.json 2>/dev/null
python bench.py --workload cfg1 > $OUT/r2_bench_cfg1.json 2>/dev/null
python bench.py --workload cfg5 --steps 5 --warmup 2 > $OUT/r2_bench_cfg5.json 2>/dev/null
python bench.py --workload stream --steps 20 > $OUT/r2_bench_stream.json 2>/dev/null
python bench.py --cpu-threads 8 --steps 5 --warmup 2 --no-alt > $OUT/r2_bench_joint_cpu8threads.json 2>/dev/null
python bench.py --conv3d-algo direct --no-cpu-baseline > $OUT/r2_bench_joint_direct_conv.json 2>/dev/null
python bench.py --gpus 2 --workload cfg1 --steps 5 --warmup 2 > $OUT/r2_bench_gpus2_codepath.json 2>/dev/null
ls -la $OUT
