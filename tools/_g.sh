python -m pytest tests/test_gpu_wino.py tests/test_gpu_parity.py tests/test_gpu_edge_cases.py -x -q 2>&1 | tail -3
python bench.py --steps 20 --warmup 5 --no-alt --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
d=json.loads([l for l in sys.stdin.read().splitlines() if l.startswith('{')][-1]); print(d['value'], d['ms_per_step'], d['roofline']['frac'], {k:(v['avg_launch_ms'], v['kernel_algo']) for k,v in d['roofline']['mfma_kernels'].items()})"
python bench.py --workload estm --steps 20 --warmup 5 --no-alt --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
d=json.loads([l for l in sys.stdin.read().splitlines() if l.startswith('{')][-1]); print('estm', d['value'], d['ms_per_step'])"
