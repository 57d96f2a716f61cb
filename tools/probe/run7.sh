python -m pytest tests/test_gpu_score_bias.py tests/test_gpu_parity.py tests/test_gpu_extend_autopack.py tests/test_gpu_adversarial_scores.py tests/test_gpu_deterministic.py tests/test_gpu_backend.py tests/test_gpu_baseline_configs.py tests/test_gpu_fullsize.py tests/test_gpu_random.py tests/test_gpu_cascade.py tests/test_gpu_fp8.py tests/test_foreign_pool.py tests/test_cascade_groups.py -x -q 2>&1 | grep -v "^  File\|^Extension" | tail -30
python -m pytest tests/test_dispatch_coverage.py -x -q -m gpu -k "extend" 2>&1 | tail -4
for i in 1 2 3; do
for LIB in libradix_hip.so libradix_hip_nomp.so; do
RX_LIB_NAME=$LIB python bench.py --extend-only 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$LIB chunk', round(d['kernel_only']['tflops'],1), 'backend', round(d['tflops'],1), 'sclk', round(d['sustained_clock']['mhz']))"
RX_LIB_NAME=$LIB RX_EXTEND_SHAPE=0,2048,8 python bench.py --extend-only --layers 8 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$LIB prefill2k', round(d['kernel_only']['tflops'],1), 'backend', round(d['tflops'],1))"
RX_LIB_NAME=$LIB RX_EXTEND_SHAPE=512,512,32 python bench.py --extend-only --layers 8 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$LIB 512+512', round(d['kernel_only']['tflops'],1), 'backend', round(d['tflops'],1))"
done
done
