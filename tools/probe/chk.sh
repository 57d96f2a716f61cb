timeout 1500 python -m pytest tests/test_gpu_parity.py tests/test_gpu_backend.py tests/test_gpu_baseline_configs.py tests/test_gpu_cascade.py -m gpu -x -q 2>&1 | tail -2
python3 tools/extend_window_bench.py 2>/dev/null | tail -6
