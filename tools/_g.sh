for v in auto 4 auto 4; do echo "== ESTD_CONV2D_NT=$v"; ESTD_CONV2D_NT=$v python bench.py --steps 20 --warmup 5 --no-alt --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
d=json.loads([l for l in sys.stdin.read().splitlines() if l.startswith('{')][-1]); print(d['value'], d['ms_per_step'])"; done
