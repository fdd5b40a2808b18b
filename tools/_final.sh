P=r5; OUT=gpurun_out/profiles_$P; mkdir -p $OUT
last() { grep "^{" | tail -1; }
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -1
( time python bench.py 2>/dev/null | last > $OUT/${P}_bench_joint.json ) 2>&1 | grep real
timeout 2400 python -m pytest tests/ -q -m gpu > $OUT/${P}_gputests.log 2>&1
tail -2 $OUT/${P}_gputests.log
python -c "
import json; d=json.load(open('$OUT/${P}_bench_joint.json')); print(d['value'], d['ms_per_step'], d['roofline']['frac'], d['roofline']['replay'], d['parity']['within_tolerance'], d['parity']['logits_within_tolerance'] if 'logits_within_tolerance' in d['parity'] else d['parity'].get('logit_volumes_vs_oracle'))"
