export RX_EXT_PW=0
for rep in 1 2; do
for v in "" _e32_nojump; do
  echo -n "== lib$v "; RX_LIB_NAME=libradix_hip$v.so timeout 120 python bench.py --extend-only 2>&1 | tail -1 | grep -o '"tflops": [0-9.]*'
done
done
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_baseline_configs.py tests/test_gpu_random.py tests/test_gpu_backend.py -m gpu -x -q -k "extend or config or random or dense" 2>&1 | tail -3
