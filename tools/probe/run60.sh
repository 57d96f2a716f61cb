timeout 600 python -m pytest tests/test_gpu_split_items.py tests/test_cascade_groups.py -m gpu -x -q 2>&1 | tail -3
RX_SPLIT_OCC3=1 timeout 900 python -m pytest tests/test_gpu_backend.py -m gpu -x -q 2>&1 | tail -2
