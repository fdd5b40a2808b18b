timeout 600 python -m pytest tests/test_gpu_wino.py tests/test_gpu_parity.py -x -q 2>&1 | tail -2
timeout 100 python tools/fuzz_convs.py 60 3 2>&1 | tail -1
for rep in 1 2 3; do for v in main th0; do
  lib="X=1"; [ $v != main ] && lib="ESTD_LIB=$PWD/estdepth_amd/lib/libestd_hip_$v.so"
  env $lib ESTD_BINDING=ctypes python bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-alt --no-replay-profile --no-other-workloads 2>/dev/null | tail -1 | python -c "import json,sys; d=json.load(sys.stdin); print('$v:', d['value'], d['ms_per_step'])"
done; done
