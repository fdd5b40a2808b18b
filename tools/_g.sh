mkdir -p gpurun_out/r3
python bench.py --steps 20 --warmup 5 --no-alt --conv3d-algo wino2 > gpurun_out/r3/bench_wino2.json 2> gpurun_out/r3/bench_wino2.err; tail -c 1500 gpurun_out/r3/bench_wino2.json | cut -c1-300
python bench.py --steps 20 --warmup 5 --no-alt --no-cpu-baseline --conv3d-algo wino > gpurun_out/r3/bench_wino.json 2>/dev/null
python - <<'PY'
import json
for f in ("gpurun_out/r3/bench_wino2.json","gpurun_out/r3/bench_wino.json"):
    try:
        l=json.loads(open(f).read().strip().split("\n")[-1])
        print(f, l["value"], l["ms_per_step"], l.get("parity",{}).get("max_abs_depth_diff_vs_oracle_m"), {k:(v["avg_launch_ms"],v["launches"]) for k,v in l["roofline"]["mfma_kernels"].items()})
    except Exception as e: print(f, "ERR", e)
PY
