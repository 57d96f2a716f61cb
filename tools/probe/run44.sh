for i in 1 2; do
for v in "RX_GRAPH_OCC3=0" "RX_GRAPH_OCC3=1"; do
for extra in "" "--ragged"; do
env $v timeout 600 python bench.py --no-cpu-baseline --no-extend --no-radix-hit $extra 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); r=d['roofline']
print('$v $extra', round(d['value']), round(d['ms_per_step'],3), round(r['frac'],4), round(r['avg_launch_ms'],4))"
done; done; done
timeout 900 python -m pytest tests/test_gpu_split_items.py tests/test_gpu_backend.py tests/test_gpu_parity.py -m gpu -x -q 2>&1 | tail -2
