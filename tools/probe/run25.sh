timeout 600 python -m pytest tests/test_gpu_split_items.py -m gpu -x -q 2>&1 | tail -5
timeout 300 python tools/hetero_decode.py 2>&1 | grep -v amdgpu | grep "lens\|balanced, max 32\|balanced, max 16"
