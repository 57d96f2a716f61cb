timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "other_head_dims" 2>&1 | tail -3
RX_EXT_D256_AT64=1 timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_baseline_configs.py -m gpu -x -q -k "extend or config" 2>&1 | tail -2
