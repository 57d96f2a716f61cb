timeout 900 python -m pytest tests/test_gpu_split_items.py tests/test_gpu_backend.py tests/test_cascade_groups.py -m gpu -x -q 2>&1 | tail -3
timeout 600 python bench.py --no-cpu-baseline --no-extend 2>/dev/null | tail -1 > gpurun_out/bench40.json
python - <<'PY'
import json
d=json.load(open('gpurun_out/bench40.json')); r=d['roofline']
print(round(d['value']), d['ms_per_step'], r['frac'], r['avg_launch_ms'])
print({k:(round(v['us_per_layer'],1),v['splits_of_the_long_request'],round(v['frac_of_hbm_peak'],3)) for k,v in d['heterogeneous_decode'].items() if isinstance(v,dict)})
PY
