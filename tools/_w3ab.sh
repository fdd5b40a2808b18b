for rep in 1 2; do for v in 0 1; do
 ESTD_W3_EXTRA=$v python bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-alt --no-replay-profile --no-other-workloads 2>/dev/null | tail -1 | python -c "import json,sys; d=json.load(sys.stdin); print('W3_EXTRA=$v:', d['value'], d['ms_per_step'])"
done; done
timeout 1500 python -m pytest tests/test_gpu_parity.py tests/test_gpu_full_config.py tests/test_gpu_wino.py tests/test_gpu_edge_cases.py -x -q 2>&1 | tail -3
timeout 100 python tools/fuzz_convs.py 60 11 2>&1 | tail -2
