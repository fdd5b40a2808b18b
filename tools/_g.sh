CB_EPI=1 python tools/conv_bench.py 3 100 2>&1 | grep -v amdgpu
python tools/conv_bench.py 1 100 2>&1 | grep -v amdgpu
ESTD_CONV3D_ALGO=wino python tools/conv_bench.py 3 100 2>&1 | grep -v amdgpu
ESTD_CONV3D_ALGO=direct python tools/conv_bench.py 3 100 2>&1 | grep -v amdgpu
python tools/head_bench.py 2>&1 | grep -v amdgpu | tail -2
python tools/kv_bench.py 2>&1 | grep -v amdgpu | tail -4
python tools/conv2d_bench.py 2>&1 | grep -v amdgpu | cut -c1-110
