Q0="ESTD_LIB=$PWD/estdepth_amd/lib/libestd_hip_q0.so"
for rep in 1 2; do
 echo "== QSCHED=1"; ESTD_BINDING=ctypes CB_EPI=1 python tools/conv_bench.py 3 40 2>&1 | grep -v amdgpu; ESTD_BINDING=ctypes python tools/conv_bench.py 1 40 2>&1 | grep -v amdgpu; ESTD_BINDING=ctypes python tools/kv_bench.py 2>&1 | grep wino2 | tail -2
 echo "== QSCHED=0"; env $Q0 ESTD_BINDING=ctypes CB_EPI=1 python tools/conv_bench.py 3 40 2>&1 | grep -v amdgpu; env $Q0 ESTD_BINDING=ctypes python tools/conv_bench.py 1 40 2>&1 | grep -v amdgpu; env $Q0 ESTD_BINDING=ctypes python tools/kv_bench.py 2>&1 | grep wino2 | tail -2
done
for rep in 1 2; do for v in main q0; do L="X=1"; [ $v = q0 ] && L="$Q0"
 env $L ESTD_BINDING=ctypes python bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-alt --no-replay-profile --no-other-workloads 2>/dev/null | tail -1 | python -c "import json,sys; d=json.load(sys.stdin); print('$v:', d['value'], d['ms_per_step'])"
done; done
ESTD_BINDING=ctypes timeout 900 python -m pytest tests/test_gpu_wino.py tests/test_gpu_gate_fold.py tests/test_gpu_parity.py tests/test_gpu_full_config.py -x -q 2>&1 | tail -3
