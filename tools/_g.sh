mkdir -p gpurun_out/r3
timeout 3000 python -m pytest tests -m gpu -q --deselect tests/test_gpu_parity.py --deselect tests/test_gpu_wino.py 2>&1 | tail -15 > gpurun_out/r3/gputests.txt; cat gpurun_out/r3/gputests.txt
for v in "" _abl8 _abl1 _abl2 _abl16 _abl27; do echo "== $v"; ESTD_BINDING=ctypes ESTD_LIB=$PWD/estdepth_amd/lib/libestd_hip$v.so python tools/conv2d_bench.py 2>&1 | grep -v amdgpu.ids | cut -c1-60; done
