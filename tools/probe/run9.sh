python -m pytest tests/test_gpu_fp8.py -x -q -k "edge_values" 2>&1 | grep -v "^  File\|^Extension" | tail -40
python -m pytest tests/test_gpu_split_items.py tests/test_gpu_backend.py tests/test_gpu_baseline_configs.py tests/test_gpu_radix_flow.py tests/test_gpu_deterministic.py tests/test_gpu_random.py tests/test_gpu_cascade.py tests/test_gpu_draft_decode.py tests/test_gpu_fullsize.py tests/test_gpu_vs_reference_cpu.py tests/test_gpu_fp8.py tests/test_foreign_pool.py tests/test_bench_launch.py -x -q 2>&1 | grep -v "^  File\|^Extension" | tail -25
python -m pytest tests/test_dispatch_coverage.py tests/test_gpu_parity.py -x -q -m gpu -k "decode" 2>&1 | tail -3
leg() { python bench.py "$@" --no-extend --no-radix-hit --no-cpu-baseline --no-extra --full-json --steps 10 --warmup 3 2>/dev/null | tail -1 | python -c "
import json,sys; d=json.loads(sys.stdin.read()); r=d['roofline']; print('$TAG', 'ms_per_step', round(d['ms_per_step'],3), 'evt_frac', round(r['frac'],4), 'b2b', round(r.get('back_to_back',{}).get('frac',0),4), 'b2b_us', round(1e3*r.get('back_to_back',{}).get('avg_launch_ms',0),1))"; }
for i in 1 2; do
for NU in 0 1; do
  if [ $NU = 1 ]; then export RX_NO_DECODE_UNITS=1; else unset RX_NO_DECODE_UNITS; fi
  TAG="config3 nounits=$NU" leg --model llama3-70b --tp-sim 8 --bs 128 --ctx 4096 --layers 80
  TAG="tp8 nounits=$NU" leg --tp-sim 8
  TAG="config1 nounits=$NU" leg --bs 64 --ctx 2176
  TAG="headline nounits=$NU" leg
done
done
