for rep in 1 2; do for v in main f0 fd1; do
  lib="X=1"; [ $v != main ] && lib="ESTD_LIB=$PWD/estdepth_amd/lib/libestd_hip_$v.so"
  echo "== $v"; env $lib ESTD_BINDING=ctypes python tools/conv2d_bench.py 2>&1 | grep -v amdgpu | grep -v residual | cut -c1-60
done; done
for rep in 1 2; do for v in main f0; do
  lib="X=1"; [ $v != main ] && lib="ESTD_LIB=$PWD/estdepth_amd/lib/libestd_hip_$v.so"
  env $lib ESTD_BINDING=ctypes python bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-alt --no-replay-profile --no-other-workloads 2>/dev/null | tail -1 | python -c "import json,sys; d=json.load(sys.stdin); print('$v:', d['value'], d['ms_per_step'])"
done; done
