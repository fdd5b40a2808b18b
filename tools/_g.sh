python -m pytest tests/test_gpu_wino.py -x -q -m gpu 2>&1 | tail -1
CB_EPI=1 python tools/conv_bench.py 3 100 2>&1 | grep -v amdgpu
for i in 1 2; do python bench.py --steps 20 --warmup 5 --no-alt --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
d=json.loads([l for l in sys.stdin.read().splitlines() if l.startswith('{')][-1]); print(d['value'], d['ms_per_step'])"; done
