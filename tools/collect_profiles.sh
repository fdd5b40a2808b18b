#!/bin/bash
# Round-3 profile collection on the GPU box (writes under gpurun_out/profiles_r3/; copy what is kept into profiles/).
#   bash tools/collect_profiles.sh
set -u
R=$PWD
OUT=$R/gpurun_out/profiles_r3
mkdir -p $OUT tools/bin
cd /tmp && export TMPDIR=/tmp
for wl in joint estm cfg5; do
  rm -rf /tmp/prof_$wl
  rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_$wl -o p -- python $R/bench.py --workload $wl --steps 5 --warmup 2 --no-cpu-baseline --no-alt > $OUT/r3_bench_${wl}_profiled.json 2> $OUT/r3_bench_${wl}_profiled.err
  T=$(find /tmp/prof_$wl -name "p_kernel_trace.csv" | head -1)
  python $R/tools/prof_summary.py $T $OUT/r3_bench_${wl}_kernel_stats.csv 5
  python $R/tools/prof_timeline.py $T 5 --gaps > $OUT/r3_bench_${wl}_timeline.txt 2>&1
  S=$(find /tmp/prof_$wl -name "p_kernel_stats.csv" | head -1)
  [ -n "$S" ] && head -41 $S > $OUT/r3_bench_${wl}_rocprof_stats_top40.csv
done
# PMC: dominant kernel (conv_bench: N = 3 volumes) per algorithm, separate --pmc passes
for algo in wino2 wino direct; do
  ESTD_CONV3D_ALGO=$algo bash $R/tools/pmc_collect.sh "FETCH_SIZE WRITE_SIZE" $OUT/r3_conv3d_${algo}_pmc.csv -- python $R/tools/conv_bench.py 3 10 > /dev/null 2>&1
done
python $R/tools/pmc_json.py $OUT          # r3_conv3d_pmc.json: what bench.py reads for roofline.traffic (from profiles/)
bash $R/tools/pmc_collect.sh "FETCH_SIZE WRITE_SIZE" $OUT/r3_hbm_kernels_pmc.csv -- python $R/tools/hbm_bench.py > /dev/null 2>&1
# the hardware's own matrix-pipe utilisation counter of every convolution kernel, stand-alone benches
for b in "conv_bench.py 3 10" "conv_bench.py 1 10" "kv_bench.py" "head_bench.py" "conv2d_bench.py"; do
  bash $R/tools/pmc_collect.sh "MfmaUtil SQ_INSTS_VALU SQ_WAVE_CYCLES SQ_WAIT_ANY" /tmp/mfma_one.csv -- python $R/tools/$b > /dev/null 2>&1
  echo "# python tools/$b" >> $OUT/r3_mfma_util_pmc.csv; grep -v "at::native\|rocclr" /tmp/mfma_one.csv >> $OUT/r3_mfma_util_pmc.csv
done
cd $R
python tools/hbm_bench.py > $OUT/r3_hbm_bench.txt 2>&1
for algo in wino2 wino direct; do
  echo "# ESTD_CONV3D_ALGO=$algo" >> $OUT/r3_conv_bench.txt
  ESTD_CONV3D_ALGO=$algo CB_EPI=1 python tools/conv_bench.py 3 30 2>&1 | grep -v amdgpu >> $OUT/r3_conv_bench.txt
  ESTD_CONV3D_ALGO=$algo python tools/conv_bench.py 1 30 2>&1 | grep -v amdgpu >> $OUT/r3_conv_bench.txt
done
python tools/head_bench.py 2>&1 | grep -v amdgpu >> $OUT/r3_conv_bench.txt
python tools/kv_bench.py 2>&1 | grep -v amdgpu >> $OUT/r3_conv_bench.txt
python tools/conv2d_bench.py 2>&1 | grep -v amdgpu > $OUT/r3_conv2d_bench.txt
python tools/psm_small_bench.py 2>&1 | grep -v amdgpu > $OUT/r3_psm_small_bench.txt
# (built here, before the gpurun call: /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 tools/src/mfma_valu_overlap.hip -o tools/bin/mfma_valu_overlap)
[ -x tools/bin/mfma_valu_overlap ] || /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 tools/src/mfma_valu_overlap.hip -o tools/bin/mfma_valu_overlap
tools/bin/mfma_valu_overlap > $OUT/r3_mfma_valu_overlap.txt 2>&1
# default bench lines (with cpu_baseline + parity) of every workload; algorithm A/B; the world-size-1 RCCL run
last() { grep "^{" | tail -1; }
python bench.py 2>/dev/null | last > $OUT/r3_bench_joint.json
python bench.py --workload estm 2>/dev/null | last > $OUT/r3_bench_estm.json
python bench.py --workload cfg1 2>/dev/null | last > $OUT/r3_bench_cfg1.json
python bench.py --workload cfg5 --steps 5 --warmup 2 2>/dev/null | last > $OUT/r3_bench_cfg5.json
python bench.py --workload stream --steps 20 2>/dev/null | last > $OUT/r3_bench_stream.json
python bench.py --conv3d-algo wino --no-cpu-baseline --no-alt 2>/dev/null | last > $OUT/r3_bench_joint_wino1.json
python bench.py --conv3d-algo direct --no-cpu-baseline --no-alt 2>/dev/null | last > $OUT/r3_bench_joint_direct_conv.json
python bench.py --no-graph --no-cpu-baseline --no-alt 2>/dev/null | last > $OUT/r3_bench_joint_eager.json
ESTD_FORCE_DIST=1 python bench.py --no-cpu-baseline --no-alt 2>/dev/null | last > $OUT/r3_bench_joint_rccl_world1.json
ESTD_FORCE_DIST=1 python bench.py --workload estm --no-cpu-baseline --no-alt 2>/dev/null | last > $OUT/r3_bench_estm_rccl_world1.json
python bench.py --gpus 2 --workload cfg1 --steps 5 --warmup 2 2>/dev/null | last > $OUT/r3_bench_gpus2_codepath.json
ls -la $OUT
