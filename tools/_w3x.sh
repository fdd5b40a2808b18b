timeout 600 python -m pytest tests/test_gpu_wino.py -x -q -k "extra" 2>&1 | tail -2
for v in 1 0 1 0; do echo "ESTD_W3_EXTRA=$v"; ESTD_W3_EXTRA=$v python tools/kv_bench.py 2>&1 | grep "wino2 kv" | tail -1; done
