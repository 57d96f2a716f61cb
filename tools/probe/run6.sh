python -m pytest tests/test_gpu_score_bias.py tests/test_gpu_parity.py tests/test_gpu_extend_autopack.py tests/test_gpu_adversarial_scores.py tests/test_gpu_deterministic.py tests/test_gpu_backend.py tests/test_gpu_baseline_configs.py tests/test_gpu_fullsize.py tests/test_gpu_random.py tests/test_gpu_cascade.py -x -q 2>&1 | grep -v "^  File\|^Extension" | tail -40
python -m pytest tests/test_dispatch_coverage.py -x -q -m gpu -k "extend" 2>&1 | tail -4
timeout 600 python tools/fuzz_extend_forms.py 2>&1 | head -2 | cut -c1-200
timeout 600 python tools/fuzz_score_bias.py 2>&1 | tail -2
timeout 300 python tools/deterministic_bench.py 2>&1 | grep -B1 -A3 '"ms_per_launch"'
RX_OPT_EXT32_BIAS=0 timeout 300 python tools/deterministic_bench.py 2>&1 | grep -A3 'rel_bias'
python bench.py --extend-only 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('chunk', d['kernel'], d['kernel_only'], 'backend', round(d['tflops'],1))"
RX_EXTEND_SHAPE=0,2048,8 python bench.py --extend-only --layers 8 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('prefill2k', d['kernel'], d['kernel_only'], 'backend', round(d['tflops'],1))"
