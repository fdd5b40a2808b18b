python -m pytest tests/test_gpu_psm.py tests/test_gpu_parity.py -x -q 2>&1 | tail -3
for v in 1 0 1 0; do echo "== ESTD_HIP_SMALL_CONVS=$v"; ESTD_HIP_SMALL_CONVS=$v python bench.py --steps 20 --warmup 5 --no-alt --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
d=json.loads([l for l in sys.stdin.read().splitlines() if l.startswith('{')][-1]); print(d['value'], d['ms_per_step'])"; done
