python -m pytest tests/test_gpu_parity.py tests/test_gpu_extend_autopack.py tests/test_gpu_adversarial_scores.py tests/test_gpu_score_bias.py tests/test_gpu_deterministic.py tests/test_gpu_backend.py tests/test_gpu_baseline_configs.py tests/test_gpu_fullsize.py tests/test_gpu_random.py tests/test_gpu_cascade.py tests/test_gpu_vs_reference_cpu.py -x -q 2>&1 | tail -8
python -m pytest tests/test_dispatch_coverage.py -x -q -m gpu -k "extend" 2>&1 | tail -4
timeout 600 python tools/fuzz_extend_forms.py 2>&1 | tail -3
timeout 600 python tools/fuzz_score_bias.py 2>&1 | tail -3
N=12 timeout 600 python tools/fuzz_deterministic.py 2>&1 | tail -2
timeout 300 python tools/deterministic_bench.py 2>&1 | grep -B1 -A3 '"ms_per_launch"'
for MP in 1 0; do
RX_OPT_EXT32_MASK_PIPE=$MP python bench.py --extend-only 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('mask_pipe=$MP chunk', d['kernel'], d['kernel_only'], 'backend', round(d['tflops'],1))"
RX_OPT_EXT32_MASK_PIPE=$MP RX_EXTEND_SHAPE=0,2048,8 python bench.py --extend-only --layers 8 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('mask_pipe=$MP prefill2k', d['kernel'], d['kernel_only'], 'backend', round(d['tflops'],1))"
RX_OPT_EXT32_MASK_PIPE=$MP RX_EXTEND_SHAPE=512,512,32 python bench.py --extend-only --layers 8 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('mask_pipe=$MP 512+512', d['kernel'], d['kernel_only'], 'backend', round(d['tflops'],1))"
done
