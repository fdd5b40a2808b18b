P=r5; OUT=gpurun_out/profiles_$P; mkdir -p $OUT
last() { grep "^{" | tail -1; }
python bench.py 2>/dev/null | last > $OUT/${P}_bench_joint.json
python bench.py --workload estm 2>/dev/null | last > $OUT/${P}_bench_estm.json
python bench.py --workload cfg5 --steps 5 --warmup 2 2>/dev/null | last > $OUT/${P}_bench_cfg5.json
timeout 2400 python -m pytest tests/ -q -m gpu > $OUT/${P}_gputests.log 2>&1
tail -3 $OUT/${P}_gputests.log
