export RX_EXT_PW=2
timeout 300 python tools/pw_debug.py 2>&1 | grep -v amdgpu | grep -c "nan 0"
timeout 300 python tools/pw_debug.py 2>&1 | grep -v amdgpu | head -9
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_baseline_configs.py tests/test_gpu_random.py tests/test_foreign_pool.py tests/test_gpu_backend.py -m gpu -x -q -k "extend or config or random or foreign or dense" 2>&1 | tail -4
for v in "" _pw_abl4; do
  echo "== lib$v"; RX_LIB_NAME=libradix_hip$v.so timeout 120 python bench.py --extend-only 2>&1 | tail -1 | grep -o '"tflops": [0-9.]*'
done
RX_EXT_PW=0 timeout 120 python bench.py --extend-only 2>&1 | tail -1 | grep -o '"tflops": [0-9.]*'
