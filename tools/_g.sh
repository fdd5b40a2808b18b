for m in 256 128 64 256 128; do echo "== min items $m"; ESTD_HIP3X3_MIN_ITEMS=$m python bench.py --steps 20 --warmup 5 --no-alt --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
d=json.loads([l for l in sys.stdin.read().splitlines() if l.startswith('{')][-1]); print(d['value'], d['ms_per_step'])"; done
