for i in 1 2; do
for v in "" "RX_NO_SPLIT_ITEMS=1"; do
env $v timeout 600 python bench.py --no-cpu-baseline --no-extend --no-radix-hit 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); r=d['roofline']
print('$v', round(d['value']), round(d['ms_per_step'],3), round(r['frac'],4), round(r['avg_launch_ms'],4))"
done; done
