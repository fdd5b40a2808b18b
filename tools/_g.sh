python tools/fuzz_convs.py 90 11 2>&1 | grep -v amdgpu | tail -3
python tools/fuzz_convs.py 60 12 2>&1 | grep -v amdgpu | tail -3
