python -m pytest tests/test_gpu_wino.py tests/test_gpu_parity.py -x -q -m gpu 2>&1 | tail -1
CB_EPI=1 python tools/conv_bench.py 3 100 2>&1 | grep -v amdgpu
python tools/conv_bench.py 1 100 2>&1 | grep -v amdgpu
python tools/head_bench.py 2>&1 | grep -v amdgpu | tail -2
python tools/kv_bench.py 2>&1 | grep "wino2 kv" | tail -1
for i in 1 2 3; do python bench.py --steps 20 --warmup 5 --no-alt --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
d=json.loads([l for l in sys.stdin.read().splitlines() if l.startswith('{')][-1]); print(d['value'], d['ms_per_step'])"; done
