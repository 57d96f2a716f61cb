for lib in libradix_hip.so libradix_hip_dec3.so; do
for extra in "" "--ragged" "--tp-sim 8" "--bs 64 --ctx 2176"; do
echo "== $lib $extra"
RX_LIB_NAME=$lib timeout 600 python bench.py --no-cpu-baseline --no-extend $extra 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); r=d['roofline']
print(round(d['value']), d['ms_per_step'], r['frac'], r['avg_launch_ms'], {k:(round(v['us_per_layer']) if isinstance(v,dict) and 'us_per_layer' in v else None) for k,v in (d.get('heterogeneous_decode') or d.get('hetero_decode') or {}).items()})"
done; done
for lib in libradix_hip.so libradix_hip_dec3.so; do for wg in 640 768 1024; do
echo "== $lib wg $wg"
RX_HETERO_WG=$wg RX_LIB_NAME=$lib timeout 600 python bench.py --no-cpu-baseline --no-extend --bs 32 --ctx 1024 --layers 2 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read())
print({k:(round(v['us_per_layer']),v['splits_of_the_long_request']) for k,v in d['heterogeneous_decode'].items() if isinstance(v,dict)})"
done; done
