for i in 1 2; do
for LIB in libradix_hip.so libradix_hip_d2.so; do
  RX_LIB_NAME=$LIB python bench.py --no-extend --no-radix-hit --no-cpu-baseline --no-extra --full-json --steps 10 --warmup 3 2>/dev/null | tail -1 | python -c "
import json,sys; d=json.loads(sys.stdin.read()); r=d['roofline']; print('$LIB', 'ms_per_step', round(d['ms_per_step'],3), 'evt_ms', round(r['avg_launch_ms'],5), 'frac', round(r['frac'],4), 'b2b', r.get('back_to_back',{}).get('frac'), r['kernel'])"
done
done
for LIB in libradix_hip.so libradix_hip_d2.so; do
  RX_LIB_NAME=$LIB python bench.py --ragged --no-extend --no-radix-hit --no-cpu-baseline --no-extra --full-json --steps 10 --warmup 3 2>/dev/null | tail -1 | python -c "
import json,sys; d=json.loads(sys.stdin.read()); r=d['roofline']; print('ragged $LIB', 'ms_per_step', round(d['ms_per_step'],3), 'evt_ms', round(r['avg_launch_ms'],5), 'frac', round(r['frac'],4), 'b2b', r.get('back_to_back',{}).get('frac'))"
  RX_LIB_NAME=$LIB python bench.py --tp-sim 2 --no-extend --no-radix-hit --no-cpu-baseline --no-extra --full-json --steps 10 --warmup 3 2>/dev/null | tail -1 | python -c "
import json,sys; d=json.loads(sys.stdin.read()); r=d['roofline']; print('tp2 $LIB', 'ms_per_step', round(d['ms_per_step'],3), 'evt_ms', round(r['avg_launch_ms'],5), 'frac', round(r['frac'],4), 'b2b', r.get('back_to_back',{}).get('frac'))"
  RX_LIB_NAME=$LIB python bench.py --tp-sim 4 --no-extend --no-radix-hit --no-cpu-baseline --no-extra --full-json --steps 10 --warmup 3 2>/dev/null | tail -1 | python -c "
import json,sys; d=json.loads(sys.stdin.read()); r=d['roofline']; print('tp4 $LIB', 'ms_per_step', round(d['ms_per_step'],3), 'evt_ms', round(r['avg_launch_ms'],5), 'frac', round(r['frac'],4), 'b2b', r.get('back_to_back',{}).get('frac'))"
  RX_LIB_NAME=$LIB python bench.py --tp-sim 8 --no-extend --no-radix-hit --no-cpu-baseline --no-extra --full-json --steps 10 --warmup 3 2>/dev/null | tail -1 | python -c "
import json,sys; d=json.loads(sys.stdin.read()); r=d['roofline']; print('tp8 $LIB', 'ms_per_step', round(d['ms_per_step'],3), 'evt_ms', round(r['avg_launch_ms'],5), 'frac', round(r['frac'],4), 'b2b', r.get('back_to_back',{}).get('frac'))"
done
