python tools/_gt.py 2>&1 | grep -v amdgpu
python -m pytest tests/test_gpu_wino.py tests/test_gpu_parity.py tests/test_gpu_full_config.py -x -q -m gpu 2>&1 | tail -1
for i in 1 2; do python bench.py --steps 20 --warmup 5 --no-alt --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
d=json.loads([l for l in sys.stdin.read().splitlines() if l.startswith('{')][-1]); print(d['value'], d['ms_per_step'], d.get('parity',{}).get('max_abs_depth_diff_vs_oracle_m'))"; done
