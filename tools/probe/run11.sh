export RX_EXT_PW=2
timeout 300 python tools/pw_debug.py 2>&1 | grep -v amdgpu | awk '{print $NF, $(NF-3), $0}' | cut -c1-200 | sort | uniq -c | sort -rn | head -5
timeout 300 python tools/pw_debug.py 2>&1 | grep -v "nan 0" | head
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_baseline_configs.py tests/test_gpu_random.py tests/test_foreign_pool.py tests/test_gpu_backend.py -m gpu -x -q -k "extend or config or random or foreign or dense" 2>&1 | tail -4
for v in "" _pw_abl4; do
  echo "== lib$v"; RX_LIB_NAME=libradix_hip$v.so timeout 120 python bench.py --extend-only 2>&1 | tail -1 | grep -o '"tflops": [0-9.]*'
done
echo "== stamps"; RX_LIB_NAME=libradix_hip_pwstamp.so timeout 120 python tools/pw_stamps.py 2>&1 | tail -9 | head -6
