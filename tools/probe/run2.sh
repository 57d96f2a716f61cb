set -x
python -m pytest tests/test_gpu_deterministic.py tests/test_gpu_vs_reference_cpu.py tests/test_gpu_fp8.py tests/test_gpu_allocator.py tests/test_gpu_score_bias.py tests/test_gpu_radix_flow.py -x -q 2>&1 | tail -15
python -m pytest tests/test_dispatch_coverage.py -x -q -m gpu -k "uni_kernel" 2>&1 | tail -5
python -m pytest tests/test_gpu_allreduce.py -x -q -s 2>&1 | grep -v "^$" | tail -60
N=16 timeout 600 python tools/fuzz_deterministic.py 2>&1 | tail -5
timeout 300 python tools/deterministic_bench.py 2>&1 | tail -40
RX_OPT_EXT32_UNI=0 timeout 300 python tools/deterministic_bench.py 2>&1 | grep -A4 unified_deterministic
SHAPES=128x4096 HQ=8 HKV=1 SPLITS=1,2,4,8 timeout 300 python tools/decode_timeline.py 2>&1 | tail -12
SHAPES=256x4096 HQ=4 HKV=1 SPLITS=1,2,4 timeout 300 python tools/decode_timeline.py 2>&1 | tail -12
SHAPES=64x2176 HQ=32 HKV=8 SPLITS=1,2 timeout 300 python tools/decode_timeline.py 2>&1 | tail -12
