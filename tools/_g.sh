for nw in 8 4; do echo "== NW=$nw"; ESTD_WINO2_WAVES=$nw python -m pytest tests/test_gpu_wino.py -x -q -k "wino2 or fuzz" 2>&1 | tail -2
ESTD_WINO2_WAVES=$nw ESTD_CONV3D_ALGO=wino2 CB_EPI=1 python tools/conv_bench.py 3 30 2>&1 | grep -v amdgpu.ids
ESTD_WINO2_WAVES=$nw ESTD_CONV3D_ALGO=wino2 python tools/conv_bench.py 1 30 2>&1 | grep conv3d; done
