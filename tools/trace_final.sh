set -u
R=$PWD; P=r6f; OUT=$R/gpurun_out/profiles_${P}t; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
for wl in joint estm; do
  rm -rf /tmp/prof_$wl
  rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_$wl -o p -- python $R/bench.py --workload $wl --steps 5 --warmup 2 --no-cpu-baseline --no-alt --no-replay-profile --no-other-workloads --sustained-s 0 > $OUT/${P}_bench_${wl}_profiled.json 2> /dev/null
  T=$(find /tmp/prof_$wl -name "p_kernel_trace.csv" | head -1)
  python $R/tools/prof_summary.py $T $OUT/${P}_bench_${wl}_kernel_stats.csv 5
  python $R/tools/prof_timeline.py $T 5 --gaps > $OUT/${P}_bench_${wl}_timeline.txt 2>&1
  python $R/tools/prof_seq.py $T -1 60 5 > $OUT/${P}_bench_${wl}_seq.txt 2>&1
  S=$(find /tmp/prof_$wl -name "p_kernel_stats.csv" | head -1)
  [ -n "$S" ] && head -41 $S > $OUT/${P}_bench_${wl}_rocprof_stats_top40.csv
done
for b in "conv1x1_bench.py" "xout_bench.py" "kv_bench.py"; do
  bash $R/tools/pmc_collect.sh "MfmaUtil SQ_INSTS_VALU SQ_WAVE_CYCLES SQ_WAIT_ANY" /tmp/mfma_one.csv -- python $R/tools/$b > /dev/null 2>&1
  echo "# python tools/$b" >> $OUT/${P}_mfma_util_pmc.csv; grep -v "at::native\|rocclr" /tmp/mfma_one.csv >> $OUT/${P}_mfma_util_pmc.csv
done
bash $R/tools/pmc_collect.sh "FETCH_SIZE WRITE_SIZE" $OUT/${P}_xout_pmc.csv -- python $R/tools/xout_bench.py > /dev/null 2>&1
cd $R
python tools/conv1x1_bench.py 2>&1 | grep -v amdgpu > $OUT/${P}_conv1x1_bench.txt
ls $OUT
